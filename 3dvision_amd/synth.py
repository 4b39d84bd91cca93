"""Deterministic synthetic workloads for the parity tests and bench.py.

Object: an asymmetric solid so that registration is well conditioned — a cuboid 0.20 x 0.12 x 0.06 m
with a 0.04 m cube boss on its top face at an off-centre corner.  Surface points come from
area-weighted stratified sampling; every random number is a uniform double from numpy's PCG64
bit generator with a fixed seed (Gaussians are Box-Muller on those uniforms), so the clouds are the
same on every machine with this numpy.

Scene = model moved by inverse(T_gt) (so that T_gt maps scene -> model, the direction
Registration::icpRefine / ransacRegistration estimate), plus Gaussian noise and uniform outliers.
"""
import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def rotation_from_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def make_transform(axis, angle_deg, translation):
    T = np.eye(4)
    T[:3, :3] = rotation_from_axis_angle(axis, np.deg2rad(angle_deg))
    T[:3, 3] = translation
    return T.astype(np.float32)


def _faces():
    """List of (origin, edge_u, edge_v, outward normal) rectangles of the object's surface."""
    L, W, H, B = 0.20, 0.12, 0.06, 0.04
    f = []

    def box(o, size, skip_bottom=False, top_hole=None):
        o = np.asarray(o, np.float64); sx, sy, sz = size
        ex, ey, ez = np.array([sx, 0, 0.]), np.array([0, sy, 0.]), np.array([0, 0, sz])
        if not skip_bottom:
            f.append((o, ex, ey, np.array([0, 0, -1.])))
        f.append((o + ez, ex, ey, np.array([0, 0, 1.])))
        f.append((o, ex, ez, np.array([0, -1., 0])))
        f.append((o + ey, ex, ez, np.array([0, 1., 0])))
        f.append((o, ey, ez, np.array([-1., 0, 0])))
        f.append((o + ex, ey, ez, np.array([1., 0, 0])))

    box([-L / 2, -W / 2, -H / 2], (L, W, H))
    # boss sits on the top face near the (+x, +y) corner, inset by 1 cm; its footprint on the top
    # face is also sampled (interior points), which is harmless for registration
    box([L / 2 - 0.01 - B, W / 2 - 0.01 - B, H / 2], (B, B, B), skip_bottom=True)
    return f


def sample_object(n, seed=42):
    """n surface points (float32 [n,3]) with analytic outward normals (float32 [n,3])."""
    rng = _rng(seed)
    faces = _faces()
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v, _ in faces])
    cum = np.cumsum(areas) / areas.sum()
    # stratified choice of face: sorted stratified uniforms mapped through the area CDF
    s = (np.arange(n) + rng.random(n)) / n
    face_id = np.searchsorted(cum, s, side="right").clip(0, len(faces) - 1)
    a = rng.random(n); b = rng.random(n)
    pts = np.empty((n, 3), np.float64); nrm = np.empty((n, 3), np.float64)
    for k, (o, u, v, nn) in enumerate(faces):
        m = face_id == k
        pts[m] = o + a[m, None] * u + b[m, None] * v
        nrm[m] = nn
    perm = _rng(seed + 1).permutation(n)  # break the face ordering
    return pts[perm].astype(np.float32), nrm[perm].astype(np.float32)


def gt_transform(seed=42, angle_deg=20.0, translation=(0.03, -0.02, 0.80)):
    rng = _rng(seed + 7)
    axis = rng.random(3) * 2 - 1
    return make_transform(axis, angle_deg, translation)


def perturb(T, seed=42, angle_deg=3.0, trans=0.005):
    rng = _rng(seed + 11)
    axis = rng.random(3) * 2 - 1
    d = rng.random(3) * 2 - 1
    d = d / np.linalg.norm(d) * trans
    P = make_transform(axis, angle_deg, d)
    return (P.astype(np.float64) @ T.astype(np.float64)).astype(np.float32)


def make_scene(n, seed=42, noise_sigma=0.0002, outlier_frac=0.10, T_gt=None):
    """Scene cloud (float32 [n,3]) such that T_gt maps it onto the model frame."""
    if T_gt is None:
        T_gt = gt_transform(seed)
    rng = _rng(seed + 3)
    n_out = int(round(n * outlier_frac))
    n_in = n - n_out
    pts, _ = sample_object(n_in, seed + 5)
    u1 = np.maximum(rng.random((n_in, 3)), 1e-12); u2 = rng.random((n_in, 3))
    gauss = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)
    pts = pts.astype(np.float64) + noise_sigma * gauss
    lo = np.array([-0.12, -0.08, -0.05]); hi = np.array([0.12, 0.08, 0.09])
    outl = lo + rng.random((n_out, 3)) * (hi - lo)
    allp = np.concatenate([pts, outl], 0)
    Tinv = np.linalg.inv(T_gt.astype(np.float64))
    scene = allp @ Tinv[:3, :3].T + Tinv[:3, 3]
    perm = _rng(seed + 13).permutation(n)
    return scene[perm].astype(np.float32), T_gt


def mean_spacing(n):
    """Approximate mean point spacing (m) of an n-point sampling of the object."""
    area = sum(np.linalg.norm(np.cross(u, v)) for _, u, v, _ in _faces())
    return float(np.sqrt(area / max(n, 1)))


def random_features(n, seed):
    """Descriptor-shaped random data: non-negative, rows sum to 1 (like FPFH), float32 [n,33]."""
    rng = _rng(seed)
    f = rng.random((n, 33)) ** 3
    f /= f.sum(1, keepdims=True)
    return f.astype(np.float32)


def rotation_angle(Ra, Rb):
    """Angle (rad) of Ra^T Rb."""
    M = np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)
    # atan2(|sin|, cos): arccos((tr-1)/2) alone is ill-conditioned near 0 (1e-8 of float32
    # rounding in the trace already reads as 1.4e-4 rad)
    s = 0.5 * np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    c = (np.trace(M) - 1) / 2
    return float(np.arctan2(np.linalg.norm(s), c))
