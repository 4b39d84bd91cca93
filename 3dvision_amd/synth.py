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


# --------------------------------------------------------------------------------------------------
# Relief part: the workload of the FULL chain (voxel -> normals -> FPFH -> match -> RANSAC -> ICP).
#
# The cuboid above is fine for ICP- or scoring-only runs, but its faces are flat: FPFH descriptors are
# identical over most of the surface, the reference's nearest-descriptor matching (src/registration.cpp:216-232)
# returns arbitrary points there and no 3-point hypothesis ever collects inliers.  A part that the reference's own
# chain can register needs local shape everywhere at the scale of the FPFH radius (5 voxels): a plate whose top is
# a sum of anisotropic Gaussian bumps and dents of irregular size, position and orientation.  The reference model
# is a SCAN of that part (rendered at a canonical pose and unprojected), as in a bin-picking cell: the reference
# flips every normal towards the origin (registration.cpp:125-127), so model and scene normals only agree when both
# clouds are seen from a viewpoint at their origin.
class ReliefPart:
    """Height field z = h(x, y) over [-L/2, L/2] x [-W/2, W/2] (metres), bumps drawn from PCG64(seed).
    `feature` is the typical bump radius; keep it at 3-9 voxels of the registration's voxel size."""

    def __init__(self, seed=3, L=0.12, W=0.08, feature=0.011, density=0.09):
        rng = _rng(seed)
        self.L, self.W = float(L), float(W)
        nb = max(4, int(round(density * L * W / (feature * feature))))
        b = np.empty((nb, 6))
        for i in range(nb):  # one draw order, so that (seed, nb) fixes the part
            bx = (rng.random() - 0.5) * L * 0.8
            by = (rng.random() - 0.5) * W * 0.8
            amp = (0.55 + 1.1 * rng.random()) * feature * (1.0 if rng.random() < 0.7 else -0.6)
            sx = (0.55 + 0.9 * rng.random()) * feature
            sy = (0.55 + 0.9 * rng.random()) * feature
            b[i] = (bx, by, amp, sx, sy, rng.random() * np.pi)
        self.bumps = b

    def height_grid(self, step):
        """(xs, ys, Z[len(ys), len(xs)]) on a regular grid; each bump is accumulated inside its 4.5-sigma window."""
        xs = np.arange(-self.L / 2, self.L / 2 + 1e-9, step)
        ys = np.arange(-self.W / 2, self.W / 2 + 1e-9, step)
        Z = np.zeros((len(ys), len(xs)))
        for bx, by, amp, sx, sy, rot in self.bumps:
            r = 4.5 * max(sx, sy)
            i0, i1 = np.searchsorted(xs, [bx - r, bx + r]); j0, j1 = np.searchsorted(ys, [by - r, by + r])
            X, Y = np.meshgrid(xs[i0:i1], ys[j0:j1])
            c, s = np.cos(rot), np.sin(rot)
            u = (X - bx) * c + (Y - by) * s; v = -(X - bx) * s + (Y - by) * c
            Z[j0:j1, i0:i1] += amp * np.exp(-0.5 * ((u / sx) ** 2 + (v / sy) ** 2))
        return xs, ys, Z

    def surface_points(self, step):
        """Dense float64 [n,3] samples of the top surface (render input; `step` well below the pixel footprint)."""
        xs, ys, Z = self.height_grid(step)
        X, Y = np.meshgrid(xs, ys)
        return np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)


def scan_pose(distance=0.5):
    """Pose (part frame -> camera frame) of the reference scan: the part's top faces a camera at the origin."""
    return make_transform([1.0, 0.0, 0.0], 180.0, (0.0, 0.0, distance)).astype(np.float64)


def instance_pose(b, distance=0.5, tilt_deg=30.0, seed=42):
    """Pose of instance b in the camera frame: the scan pose, tilted by up to `tilt_deg` about an in-plane axis, turned
    by any angle about the viewing axis and shifted by a few centimetres.  Deterministic in (b, seed)."""
    rng = _rng(seed * 7919 + b)
    a = rng.random() * 2 * np.pi
    tilt = make_transform([np.cos(a), np.sin(a), 0.0], tilt_deg * rng.random(), (0, 0, 0)).astype(np.float64)
    spin = make_transform([0.0, 0.0, 1.0], 360.0 * rng.random(), (0, 0, 0)).astype(np.float64)
    shift = np.eye(4); shift[:3, 3] = [(rng.random() - 0.5) * 0.04, (rng.random() - 0.5) * 0.04, distance * (1.0 + 0.08 * (rng.random() - 0.5))]
    M0 = make_transform([1.0, 0.0, 0.0], 180.0, (0, 0, 0)).astype(np.float64)
    return shift @ spin @ tilt @ M0


def render_depth(points, pose, fx, fy, cx, cy, width, height, scale_to_meters):
    """z-buffer splat of dense surface samples: uint16 depth (units of 1/scale_to_meters m) and the 0/255 mask of the
    pixels hit.  Numpy; the bench renders the same way with torch on the GPU."""
    p = points @ pose[:3, :3].T + pose[:3, 3]
    u = np.round(p[:, 0] / p[:, 2] * fx + cx).astype(np.int64)
    v = np.round(p[:, 1] / p[:, 2] * fy + cy).astype(np.int64)
    ok = (u >= 0) & (u < width) & (v >= 0) & (v < height) & (p[:, 2] > 0)
    z = np.full(height * width, np.inf)
    np.minimum.at(z, v[ok] * width + u[ok], p[ok, 2])
    z = z.reshape(height, width)
    hit = np.isfinite(z)
    depth = np.zeros((height, width), np.uint16)
    depth[hit] = np.round(z[hit] * scale_to_meters).astype(np.uint16)
    return depth, np.where(hit, 255, 0).astype(np.uint8)


def pose_error(T_est, T_gt):
    """(rotation angle in rad, translation distance in m) between two 4x4 transforms."""
    return rotation_angle(np.asarray(T_gt)[:3, :3], np.asarray(T_est)[:3, :3]), float(
        np.linalg.norm(np.asarray(T_gt, np.float64)[:3, 3] - np.asarray(T_est, np.float64)[:3, 3]))


def render_depth_torch(points_t, pose, fx, fy, cx, cy, width, height, scale_to_meters):
    """render_depth on the GPU (torch, float64): same splat, same rounding; returns (int16-viewed uint16 depth tensor
    [H, W], uint8 mask tensor [H, W]) on the device of points_t.  Used by the benches to build hundreds of frames."""
    import torch
    T = torch.as_tensor(np.asarray(pose, np.float64), device=points_t.device)
    p = points_t @ T[:3, :3].T + T[:3, 3]
    u = torch.round(p[:, 0] / p[:, 2] * fx + cx).to(torch.int64)
    v = torch.round(p[:, 1] / p[:, 2] * fy + cy).to(torch.int64)
    ok = (u >= 0) & (u < width) & (v >= 0) & (v < height) & (p[:, 2] > 0)
    z = torch.full((height * width,), float("inf"), dtype=torch.float64, device=points_t.device)
    z.scatter_reduce_(0, (v[ok] * width + u[ok]), p[ok, 2], reduce="amin")
    hit = torch.isfinite(z)
    d = torch.where(hit, torch.round(z * scale_to_meters), torch.zeros_like(z)).to(torch.int32)
    depth = torch.where(d > 32767, d - 65536, d).to(torch.int16)   # uint16 bits in an int16 tensor (torch has no uint16 arithmetic)
    return depth.reshape(height, width), (hit.to(torch.uint8) * 255).reshape(height, width)


# --------------------------------------------------------------------------------------------------
# Tray scene: the workload of config C5 (BASELINE.json configs[4]: "1024-instance batch ... 500k-pt scene" per GPU).
# ONE depth frame shows a tray of small relief parts, one per grid cell, each at its own pose (any spin about the viewing
# axis, a tilt, a small shift inside its cell); the instances are cut out of the frame by ONE label image (label = instance
# + 1, uint16: more than 255 instances).  1,024 parts of ~490 pixels each make a scene cloud of ~500k points.  The model is
# a scan of the same part at the centre of the image, as Pipeline::run prepares it (src/pipeline.cpp:275-294).
class DiscPart(ReliefPart):
    """A round relief part (diameter D): the bumps of ReliefPart over a disc, so that it fits its cell at any spin."""

    def __init__(self, seed, D, feature, density=0.16):
        ReliefPart.__init__(self, seed, L=D, W=D, feature=feature, density=density)
        self.D = float(D)

    def surface_points(self, step):
        p = ReliefPart.surface_points(self, step)
        return p[p[:, 0] ** 2 + p[:, 1] ** 2 <= (self.D / 2) ** 2]


def tray_cells(n_instances, width, height, cell_w, cell_h):
    """Centres (u, v) in pixels of the first n_instances cells of the grid that fills the frame, row by row."""
    cols, rows = width // cell_w, height // cell_h
    assert cols * rows >= n_instances, "frame %dx%d holds %d cells of %dx%d, %d wanted" % (width, height, cols * rows, cell_w, cell_h, n_instances)
    x0 = (width - cols * cell_w) / 2.0; y0 = (height - rows * cell_h) / 2.0
    b = np.arange(n_instances)
    return np.stack([x0 + (b % cols + 0.5) * cell_w, y0 + (b // cols + 0.5) * cell_h], 1)


def tray_instance_pose(b, centre_uv, fx, fy, cx, cy, distance, tilt_deg, jitter_px, seed=42):
    """Pose (part frame -> camera frame) of the part in cell b: top towards the camera, any spin, a tilt of up to tilt_deg,
    its centre within jitter_px of the cell centre, at distance * (1 +- 1 %).  Deterministic in (b, seed)."""
    rng = _rng(seed * 104729 + b)
    a = rng.random() * 2 * np.pi
    tilt = make_transform([np.cos(a), np.sin(a), 0.0], tilt_deg * rng.random(), (0, 0, 0)).astype(np.float64)
    spin = make_transform([0.0, 0.0, 1.0], 360.0 * rng.random(), (0, 0, 0)).astype(np.float64)
    z = distance * (1.0 + 0.02 * (rng.random() - 0.5))
    u = centre_uv[0] + (rng.random() - 0.5) * 2 * jitter_px; v = centre_uv[1] + (rng.random() - 0.5) * 2 * jitter_px
    shift = np.eye(4); shift[:3, 3] = [(u - cx) * z / fx, (v - cy) * z / fy, z]
    M0 = make_transform([1.0, 0.0, 0.0], 180.0, (0, 0, 0)).astype(np.float64)
    return shift @ spin @ tilt @ M0


def render_tray(points, poses, fx, fy, cx, cy, width, height, scale_to_meters):
    """z-buffer splat of the same dense part samples under every pose into ONE frame: uint16 depth, uint16 label image
    (0 = background, b + 1 = the instance whose surface is nearest at the pixel).  Numpy."""
    z = np.full(height * width, np.inf); lab = np.zeros(height * width, np.uint16)
    idx_all, z_all, b_all = [], [], []
    for b, pose in enumerate(poses):
        p = points @ pose[:3, :3].T + pose[:3, 3]
        u = np.round(p[:, 0] / p[:, 2] * fx + cx).astype(np.int64)
        v = np.round(p[:, 1] / p[:, 2] * fy + cy).astype(np.int64)
        ok = (u >= 0) & (u < width) & (v >= 0) & (v < height) & (p[:, 2] > 0)
        idx_all.append(v[ok] * width + u[ok]); z_all.append(p[ok, 2]); b_all.append(np.full(int(ok.sum()), b + 1, np.uint16))
    idx = np.concatenate(idx_all); pz = np.concatenate(z_all); pb = np.concatenate(b_all)
    np.minimum.at(z, idx, pz)
    first = pz == z[idx]                       # the samples that set their pixel's depth; ties between instances: the lowest label
    order = np.argsort(-pb[first].astype(np.int64), kind="stable")
    lab[idx[first][order]] = pb[first][order]
    hit = np.isfinite(z)
    depth = np.zeros(height * width, np.uint16)
    depth[hit] = np.round(z[hit] * scale_to_meters).astype(np.uint16)
    return depth.reshape(height, width), lab.reshape(height, width)


def tray_scene(n_instances, width=1280, height=720, f=1500.0, distance=0.45, part_px=25, cell_w=31, cell_h=28, voxel_px=1.2,
               tilt_deg=12.0, jitter_px=1.0, scale_to_meters=50000.0, seed=3, feature_voxels=3.5, pose_seed=None):
    """The C5 workload: dict(depth u16 [H,W], label u16 [H,W], T_gt list (scene -> model), model_depth, model_mask, voxel,
    intrinsics).  Defaults: 1,025 cells of 31 x 28 px in a 1280 x 720 frame, parts of 25 px diameter (~490 px)."""
    cx, cy = width / 2.0, height / 2.0
    px = distance / f
    voxel = voxel_px * px
    part = DiscPart(seed, part_px * px, feature_voxels * voxel)
    dense = part.surface_points(px / 2.5)
    M = scan_pose(distance)
    model_depth, model_mask = render_depth(dense, M, f, f, cx, cy, width, height, scale_to_meters)
    cells = tray_cells(n_instances, width, height, cell_w, cell_h)
    # (pose_seed: other poses of the SAME part - the ranks of a sharded job each see their own tray of one part)
    poses = [tray_instance_pose(b, cells[b], f, f, cx, cy, distance, tilt_deg, jitter_px, seed if pose_seed is None else pose_seed) for b in range(n_instances)]
    depth, label = render_tray(dense, poses, f, f, cx, cy, width, height, scale_to_meters)
    return dict(depth=depth, label=label, T_gt=[M @ np.linalg.inv(S) for S in poses], model_depth=model_depth, model_mask=model_mask,
                voxel=float(np.float32(voxel)), fx=f, fy=f, cx=cx, cy=cy, width=width, height=height, scale=scale_to_meters,
                zmax=1.3, bumps=len(part.bumps))
