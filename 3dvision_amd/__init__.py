"""3dvision_amd — MI355X (gfx950) point-cloud registration backend.

Python front end over the C ABI of ``include/tdv_hip.h`` (``lib3dvision_hip.so``, hand-written
HIP).  It mirrors the reference's operator API so that tests read like calls into the reference:

    reference (C++, namespace industry_picking)           here
    ------------------------------------------------      -----------------------------------
    PointCloud / FPFHFeatures / RegistrationResult        PointCloud / ndarray[n,33] / RegistrationResult
    GPUDepth::preprocess, ::isCudaAvailable               GPUDepth.preprocess, .isCudaAvailable
    GPUPointCloud::generate                               GPUPointCloud.generate
    GPURegistration::icpRefine                            GPURegistration.icpRefine
    Registration::voxelDownsample / estimateNormals /     Registration.voxelDownsample / ...
      computeFPFH / ransacRegistration / icpRefine

(include/gpu_depth.hpp:9-22, include/gpu_registration.hpp:8-19, include/registration.hpp:10-60 of the
reference).  The package directory name starts with a digit, so import it with
``importlib.import_module("3dvision_amd")``.

There is NO CPU fallback: every operator calls the HIP library and raises if the library or a GPU
is missing (``isCudaAvailable()`` is the only call that answers without one).  4x4 transforms are
ordinary row-major ``[4,4]`` numpy arrays at this level; the column-major ``float[16]`` of
``Eigen::Matrix4f::data()`` is the ABI's layout and is converted here.
"""
import ctypes as C
import os
import sys
import threading
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TDV_LIB_VARIANT=study (tests marked `study`, tools/studies): the -DTDV_STUDY build, which keeps the A/B variants that lost their
# measurement and the tuning knobs (csrc/tdv_internal.hpp: study_env).  Anything else: the product library.
STUDY_BUILD = os.environ.get("TDV_LIB_VARIANT") == "study"
LIB_PATH = os.path.join(_HERE, "lib3dvision_hip_study.so" if STUDY_BUILD else "lib3dvision_hip.so")

TDV_MASK_THRESHOLD10 = 0
TDV_MASK_NONZERO = 1
TDV_MASK_LABEL_BASE = 256
TDV_VOXEL_ORDER_FIRST = 0
TDV_VOXEL_ORDER_REFERENCE = 1
TIMER_ICP_NN, TIMER_RANSAC_SCORE, TIMER_FEATURE_MATCH, TIMER_KNN, TIMER_RADIUS, TIMER_DEPTH, TIMER_VOXEL, TIMER_FM_INDEX = range(8)
DEPTH_BATCH_MASK_PASSES = 1   # the B masks cross HBM once in tdv_depth_to_cloud_batch_dev (SURVEY.md 8d's minimum): algorithmic bytes of that op

# every symbol include/tdv_hip.h declares (checked by the CPU test-suite against the built library)
ABI_SYMBOLS = [
    "tdv_device_count", "tdv_ctx_create", "tdv_ctx_set_stream", "tdv_ctx_set_icp_search", "tdv_ctx_set_icp_accumulation", "tdv_ctx_last_icp_search", "tdv_ctx_last_batch_lanes", "tdv_ctx_last_voxel_grouping", "tdv_ctx_last_feature_match_path", "tdv_ctx_workspace_bytes", "tdv_ctx_set_ransac_score", "tdv_ctx_last_ransac_rescore", "tdv_ctx_last_ransac_scored", "tdv_ctx_get_stream", "tdv_ctx_synchronize",
    "tdv_ctx_destroy", "tdv_status_string", "tdv_last_error", "tdv_version", "tdv_timing_enable", "tdv_timing_read",
    "tdv_depth_preprocess", "tdv_deproject", "tdv_depth_to_cloud", "tdv_voxel_downsample", "tdv_estimate_normals",
    "tdv_compute_fpfh", "tdv_feature_match", "tdv_ransac", "tdv_icp", "tdv_icp_correspondences",
    "tdv_icp_dev", "tdv_ransac_dev", "tdv_feature_match_dev", "tdv_estimate_normals_dev", "tdv_compute_fpfh_dev", "tdv_normals_fpfh_dev", "tdv_radix_sort_pairs_dev",
    "tdv_depth_to_cloud_dev", "tdv_voxel_downsample_dev", "tdv_sample_triples", "tdv_pose_compose",
    "tdv_register_batch_dev", "tdv_prepare_model_dev", "tdv_bilateral_filter", "tdv_filter_duplicates", "tdv_load_ply_ascii", "tdv_load_mask_png", "tdv_load_masks_from_dir",
    "tdv_depth_to_cloud_batch_dev", "tdv_broadcast_model", "tdv_gather_results", "tdv_mask_resize_nearest", "tdv_mask_resize_nearest_dev", "tdv_voxel_downsample_batch_dev", "tdv_voxel_downsample_batch_pinhole_dev",
]


class TdvError(RuntimeError):
    pass


class RansacResultC(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("fitness", C.c_float), ("rmse", C.c_float), ("inliers", C.c_int),
                ("best_iteration", C.c_int), ("iterations_run", C.c_int)]


class BatchParamsC(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("scale_to_meters", C.c_float), ("mask_mode", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("zmax", C.c_float),
                ("voxel_size", C.c_float), ("normals_k", C.c_int), ("fpfh_radius_factor", C.c_float),
                ("ransac_max_iterations", C.c_int), ("ransac_confidence", C.c_float), ("icp_distance_factor", C.c_float),
                ("icp_max_iterations", C.c_int), ("point_to_plane", C.c_int), ("seed", C.c_uint32),
                ("voxel_order", C.c_int), ("n_frames", C.c_int), ("frame_of_instance", C.c_void_p),
                ("mask_format", C.c_int), ("mask_width", C.c_int), ("mask_height", C.c_int)]


class InstanceResultC(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("fitness", C.c_float), ("rmse", C.c_float), ("coarse_fitness", C.c_float),
                ("coarse_inliers", C.c_int), ("icp_iterations", C.c_int), ("n_points", C.c_int), ("n_voxels", C.c_int),
                ("status", C.c_int)]


class IcpResultC(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("fitness", C.c_float), ("rmse", C.c_float), ("iterations", C.c_int),
                ("n_corr", C.c_int)]


_lib = None
_lib_lock = threading.Lock()


def build():
    """Compile the HIP library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tdv_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise TdvError("lib3dvision_hip.so is not built (%s). Run `python 3dvision_amd/build.py` or "
                               "__graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
            # A process that will also use torch must load torch FIRST: its wheel brings its own copy of the HIP runtime, and once the system's
            # copy (this library's dependency) has initialised the device, torch's reports "No HIP GPUs are available".  The harness around
            # this package (tests, bench.py, tools) always uses torch for device buffers, so it is imported here when it is installed.
            if "torch" not in sys.modules:
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
            l = C.CDLL(LIB_PATH)
            l.tdv_status_string.restype = C.c_char_p
            l.tdv_last_error.restype = C.c_char_p
            l.tdv_version.restype = C.c_char_p
            l.tdv_ctx_get_stream.restype = C.c_void_p
            l.tdv_ctx_workspace_bytes.restype = C.c_ulonglong
            l.tdv_ctx_last_ransac_rescore.restype = C.c_double
            l.tdv_ctx_last_ransac_scored.restype = C.c_double
            _lib = l
    return _lib


def _check(ctx, status, what):
    if status != 0:
        msg = lib().tdv_status_string(status).decode()
        detail = lib().tdv_last_error(ctx).decode() if ctx else ""
        raise TdvError("%s: %s (%d) %s" % (what, msg, status, detail))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def to_colmajor16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def from_colmajor16(t):
    return np.array(t, dtype=np.float32).reshape(4, 4).T.copy()


def device_count():
    n = C.c_int(0)
    _check(None, lib().tdv_device_count(C.byref(n)), "tdv_device_count")
    return n.value


@dataclass
class PointCloud:
    """include/registration.hpp:10-19."""
    points: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    normals: Optional[np.ndarray] = None
    colors: Optional[np.ndarray] = None

    def size(self):
        return len(self.points)

    def empty(self):
        return len(self.points) == 0

    def hasNormals(self):
        return self.normals is not None and len(self.normals) == len(self.points)

    def hasColors(self):
        return self.colors is not None and len(self.colors) == len(self.points)


@dataclass
class RegistrationResult:
    """include/registration.hpp:26-30 (+ diagnostics the ABI also returns)."""
    transformation: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    fitness: float = 0.0
    rmse: float = 0.0
    iterations: int = 0
    inliers: int = 0
    best_iteration: int = -1
    iterations_run: int = 0
    n_corr: int = 0
    trace_inliers: Optional[np.ndarray] = None


class Context:
    """One tdv_ctx (stream + workspace).  Not thread-safe; one per host thread."""

    def __init__(self, device=0, stream=None):
        self.icp_search_name = "auto"
        self._h = C.c_void_p()
        _check(None, lib().tdv_ctx_create(int(device), C.byref(self._h)), "tdv_ctx_create")
        self.device = device
        if stream is not None:
            _check(self._h, lib().tdv_ctx_set_stream(self._h, C.c_void_p(stream)), "tdv_ctx_set_stream")

    ICP_SEARCH = {"auto": 0, "brute": 1, "pruned": 2, "grid": 3}

    def set_icp_search(self, mode):
        """'auto' (by size and cell occupancy), 'brute' (the reference's scan), 'pruned' (exact box-pruned walk) or 'grid' (hash grid
        with cells of 2.2 x the threshold; falls back to 'pruned' when the threshold is large against the spacing); same results."""
        _check(self._h, lib().tdv_ctx_set_icp_search(self._h, self.ICP_SEARCH[mode]), "tdv_ctx_set_icp_search")
        self.icp_search_name = mode

    def set_icp_accumulation(self, mode):
        """'tree' (default: f64 sums in a fixed tree, transform within tolerance of the CPU path) or 'reference' (f32 sums in
        ascending source index as registration.cpp:340-358,374-386: transform, rmse, fitness and iteration count equal the
        CPU path's bit for bit; a serial chain per iteration)."""
        _check(self._h, lib().tdv_ctx_set_icp_accumulation(self._h, {"tree": 0, "reference": 1}[mode]), "tdv_ctx_set_icp_accumulation")

    def set_ransac_score(self, mode):
        """'fast' (default: FMA pass, chunks inside the rounding band re-scored with the reference arithmetic) or 'exact'
        (the reference arithmetic only); same inlier counts."""
        _check(self._h, lib().tdv_ctx_set_ransac_score(self._h, {"fast": 0, "exact": 1, "matrix": 2}[mode]), "tdv_ctx_set_ransac_score")

    def last_ransac_rescore(self):
        """Fraction of the (hypothesis, point) tests the last RANSAC call scored a second time exactly (-1 in 'exact' mode)."""
        return float(lib().tdv_ctx_last_ransac_rescore(self._h))

    def last_ransac_scored(self):
        """Share of the (hypothesis, point) tests the last RANSAC call evaluated (< 1: calls without a trace stop scoring a
        hypothesis that can no longer beat the best count of the earlier batches; same result)."""
        return float(lib().tdv_ctx_last_ransac_scored(self._h))

    def last_feature_match_path(self):
        """'scan', 'leaf_major' or 'walk': the search the last feature_match call on this context ran."""
        return {0: "none", 1: "scan", 2: "leaf_major", 3: "walk"}[int(lib().tdv_ctx_last_feature_match_path(self._h))]

    def last_voxel_grouping(self):
        """'table' or 'pixels': how the last batched voxel stage on this context grouped its points."""
        return {0: "none", 1: "table", 2: "pixels"}[int(lib().tdv_ctx_last_voxel_grouping(self._h))]

    def last_batch_lanes(self):
        """Host lanes the last register_batch_dev call on this context used."""
        return int(lib().tdv_ctx_last_batch_lanes(self._h))

    def last_icp_search(self):
        """Name of the search the last ICP / correspondence call ran ('brute', 'pruned', 'grid'; 'auto' before any)."""
        v = lib().tdv_ctx_last_icp_search(self._h)
        return {n: k for k, n in self.ICP_SEARCH.items()}[v]

    def workspace_high_water(self):
        """Bytes of device memory held by this ctx's workspace arenas (its batch lanes' included): the high-water mark so far."""
        return int(lib().tdv_ctx_workspace_bytes(self._h))

    def close(self):
        if self._h:
            lib().tdv_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().tdv_ctx_get_stream(self._h)

    def synchronize(self):
        _check(self._h, lib().tdv_ctx_synchronize(self._h), "tdv_ctx_synchronize")

    def timing_enable(self, on=True):
        _check(self._h, lib().tdv_timing_enable(self._h, int(on)), "tdv_timing_enable")

    def timing_read(self, slot):
        ms = C.c_double(); n = C.c_int()
        _check(self._h, lib().tdv_timing_read(self._h, slot, C.byref(ms), C.byref(n)), "tdv_timing_read")
        return ms.value, n.value

    # ---------------------------------------------------------------- R1 / R2
    def depth_preprocess(self, raw, mask, scale, mask_mode=TDV_MASK_THRESHOLD10):
        raw = np.ascontiguousarray(raw, np.uint16)
        h, w = raw.shape
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.empty((h, w), np.float32)
        _check(self._h, lib().tdv_depth_preprocess(self._h, _ptr(raw), _ptr(m), w, h, C.c_float(scale), mask_mode, _ptr(out)),
               "tdv_depth_preprocess")
        return out

    def bilateral_filter(self, depth, sigma_spatial, sigma_range):
        d = _f32(depth); h, w = d.shape
        out = np.empty_like(d)
        _check(self._h, lib().tdv_bilateral_filter(self._h, _ptr(d), w, h, C.c_float(sigma_spatial), C.c_float(sigma_range), _ptr(out)),
               "tdv_bilateral_filter")
        return out

    def mask_resize_nearest(self, masks, dst_width, dst_height):
        """cv::resize(mask, ..., INTER_NEAREST) of src/pipeline.cpp:38-41; masks: uint8 [h, w] or [B, h, w]."""
        m = np.ascontiguousarray(masks, np.uint8)
        single = m.ndim == 2
        if single:
            m = m[None]
        B, sh, sw = m.shape
        out = np.empty((B, dst_height, dst_width), np.uint8)
        _check(self._h, lib().tdv_mask_resize_nearest(self._h, _ptr(m), B, sw, sh, dst_width, dst_height, _ptr(out)), "tdv_mask_resize_nearest")
        return out[0] if single else out

    def deproject(self, depth, bgr, fx, fy, cx, cy, zmax, capacity=None):
        depth = _f32(depth)
        h, w = depth.shape
        b = None if bgr is None else np.ascontiguousarray(bgr, np.uint8)
        cap = h * w if capacity is None else capacity
        xyz = np.empty((cap, 3), np.float32)
        rgb = np.empty((cap, 3), np.float32) if b is not None else None
        n = C.c_int()
        _check(self._h, lib().tdv_deproject(self._h, _ptr(depth), _ptr(b), w, h, C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                            C.c_float(cy), C.c_float(zmax), _ptr(xyz), _ptr(rgb), cap, C.byref(n)), "tdv_deproject")
        return xyz[:n.value].copy(), (None if rgb is None else rgb[:n.value].copy())

    def depth_to_cloud(self, raw, mask, bgr, scale, fx, fy, cx, cy, zmax, mask_mode=TDV_MASK_THRESHOLD10):
        raw = np.ascontiguousarray(raw, np.uint16)
        h, w = raw.shape
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        b = None if bgr is None else np.ascontiguousarray(bgr, np.uint8)
        xyz = np.empty((h * w, 3), np.float32)
        rgb = np.empty((h * w, 3), np.float32) if b is not None else None
        n = C.c_int()
        _check(self._h, lib().tdv_depth_to_cloud(self._h, _ptr(raw), _ptr(m), _ptr(b), w, h, C.c_float(scale), mask_mode,
                                                 C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(zmax),
                                                 _ptr(xyz), _ptr(rgb), h * w, C.byref(n)), "tdv_depth_to_cloud")
        return xyz[:n.value].copy(), (None if rgb is None else rgb[:n.value].copy())

    # ---------------------------------------------------------------- R3
    def voxel_downsample(self, xyz, rgb, voxel, order=TDV_VOXEL_ORDER_REFERENCE):
        xyz = _f32(xyz); rgb = _f32(rgb)
        n = len(xyz)
        oxyz = np.empty((max(n, 1), 3), np.float32)
        orgb = np.empty((max(n, 1), 3), np.float32) if rgb is not None else None
        m = C.c_int()
        _check(self._h, lib().tdv_voxel_downsample(self._h, _ptr(xyz), _ptr(rgb), n, C.c_float(voxel), order, _ptr(oxyz), _ptr(orgb),
                                                   n, C.byref(m)), "tdv_voxel_downsample")
        return oxyz[:m.value].copy(), (None if orgb is None else orgb[:m.value].copy())

    # ---------------------------------------------------------------- R4
    def estimate_normals(self, xyz, k=30, want_knn=False):
        xyz = _f32(xyz); n = len(xyz)
        nrm = np.empty((n, 3), np.float32)
        knn = np.empty((n, k), np.int32) if want_knn else None
        _check(self._h, lib().tdv_estimate_normals(self._h, _ptr(xyz), n, k, _ptr(nrm), _ptr(knn)), "tdv_estimate_normals")
        return (nrm, knn) if want_knn else nrm

    def compute_fpfh(self, xyz, normals, radius, want_neighbors=False):
        xyz = _f32(xyz); normals = _f32(normals); n = len(xyz)
        desc = np.empty((n, 33), np.float32)
        nb = np.empty((n, 100), np.int32) if want_neighbors else None
        cnt = np.empty(n, np.int32) if want_neighbors else None
        _check(self._h, lib().tdv_compute_fpfh(self._h, _ptr(xyz), _ptr(normals), n, C.c_float(radius), _ptr(desc), _ptr(nb), _ptr(cnt)),
               "tdv_compute_fpfh")
        return (desc, nb, cnt) if want_neighbors else desc

    # ---------------------------------------------------------------- R5
    def feature_match(self, fs, ft):
        fs = _f32(fs); ft = _f32(ft)
        corr = np.empty(len(fs), np.int32)
        _check(self._h, lib().tdv_feature_match(self._h, _ptr(fs), len(fs), _ptr(ft), len(ft), _ptr(corr)), "tdv_feature_match")
        return corr

    def ransac(self, src, tgt, fs=None, ft=None, corr=None, voxel=0.001, max_iterations=100000, confidence=0.999,
               seed=42, trace=False):
        src = _f32(src); tgt = _f32(tgt); fs = _f32(fs); ft = _f32(ft)
        c = None if corr is None else np.ascontiguousarray(corr, np.int32)
        res = RansacResultC()
        tr = np.full(max(max_iterations, 1), -2, np.int32) if trace else None
        _check(self._h, lib().tdv_ransac(self._h, _ptr(src), len(src), _ptr(tgt), len(tgt), _ptr(fs), _ptr(ft), _ptr(c),
                                         C.c_float(voxel), max_iterations, C.c_float(confidence), C.c_uint32(seed),
                                         C.byref(res), _ptr(tr)), "tdv_ransac")
        return RegistrationResult(transformation=from_colmajor16(res.T), fitness=np.float32(res.fitness), rmse=np.float32(res.rmse),
                                  inliers=res.inliers, best_iteration=res.best_iteration, iterations_run=res.iterations_run,
                                  trace_inliers=tr)

    # ---------------------------------------------------------------- R6
    def icp(self, src, tgt, tgt_normals, T0, thr, max_iterations=200, point_to_plane=True):
        src = _f32(src); tgt = _f32(tgt); tn = _f32(tgt_normals)
        res = IcpResultC()
        t0 = to_colmajor16(T0)
        _check(self._h, lib().tdv_icp(self._h, _ptr(src), len(src), _ptr(tgt), _ptr(tn), len(tgt), _ptr(t0), C.c_float(thr),
                                      max_iterations, int(point_to_plane), C.byref(res)), "tdv_icp")
        return RegistrationResult(transformation=from_colmajor16(res.T), fitness=np.float32(res.fitness), rmse=np.float32(res.rmse),
                                  iterations=res.iterations, n_corr=res.n_corr)

    def icp_correspondences(self, src, tgt, T, thr):
        src = _f32(src); tgt = _f32(tgt)
        ns = len(src)
        corr = np.empty(ns, np.int32); d2 = np.empty(ns, np.float32); acc = np.empty(ns, np.uint8); nc = C.c_int()
        t = to_colmajor16(T)
        _check(self._h, lib().tdv_icp_correspondences(self._h, _ptr(src), ns, _ptr(tgt), len(tgt), _ptr(t), C.c_float(thr),
                                                      _ptr(corr), _ptr(d2), _ptr(acc), C.byref(nc)), "tdv_icp_correspondences")
        return dict(corr=corr, d2=d2, accepted=acc.astype(bool), n_corr=nc.value)

    # ---------------------------------------------------------------- device-resident (pointers are ints)
    def icp_dev(self, d_src, ns, d_tgt, d_tgt_normals, nt, T0, thr, max_iterations, point_to_plane=True, fixed_iterations=False):
        res = IcpResultC()
        t0 = to_colmajor16(T0)
        _check(self._h, lib().tdv_icp_dev(self._h, _ptr(d_src), ns, _ptr(d_tgt), _ptr(d_tgt_normals), nt, _ptr(t0), C.c_float(thr),
                                          max_iterations, int(point_to_plane), int(fixed_iterations), C.byref(res)), "tdv_icp_dev")
        return RegistrationResult(transformation=from_colmajor16(res.T), fitness=np.float32(res.fitness), rmse=np.float32(res.rmse),
                                  iterations=res.iterations, n_corr=res.n_corr)

    def ransac_dev(self, d_src, ns, d_tgt, nt, d_fs, d_ft, d_corr, voxel, max_iterations, confidence=0.999, seed=42, trace=False):
        """trace=True also returns the per-iteration inlier counts (host array) - and thereby makes the call evaluate every
        (hypothesis, point) test: the exact bail-out only runs when no trace is asked for."""
        res = RansacResultC()
        tr = np.full(max_iterations, -2, np.int32) if trace else None
        _check(self._h, lib().tdv_ransac_dev(self._h, _ptr(d_src), ns, _ptr(d_tgt), nt, _ptr(d_fs), _ptr(d_ft), _ptr(d_corr),
                                             C.c_float(voxel), max_iterations, C.c_float(confidence), C.c_uint32(seed),
                                             C.byref(res), _ptr(tr)), "tdv_ransac_dev")
        return RegistrationResult(transformation=from_colmajor16(res.T), fitness=np.float32(res.fitness), rmse=np.float32(res.rmse),
                                  inliers=res.inliers, best_iteration=res.best_iteration, iterations_run=res.iterations_run, trace_inliers=tr)

    def feature_match_dev(self, d_fs, ns, d_ft, nt, d_corr):
        _check(self._h, lib().tdv_feature_match_dev(self._h, _ptr(d_fs), ns, _ptr(d_ft), nt, _ptr(d_corr)), "tdv_feature_match_dev")

    def estimate_normals_dev(self, d_xyz, n, k, d_normals, d_knn=None):
        _check(self._h, lib().tdv_estimate_normals_dev(self._h, _ptr(d_xyz), n, k, _ptr(d_normals), _ptr(d_knn)), "tdv_estimate_normals_dev")

    def compute_fpfh_dev(self, d_xyz, d_normals, n, radius, d_desc, d_nbr=None, d_cnt=None):
        _check(self._h, lib().tdv_compute_fpfh_dev(self._h, _ptr(d_xyz), _ptr(d_normals), n, C.c_float(radius), _ptr(d_desc),
                                                   _ptr(d_nbr), _ptr(d_cnt)), "tdv_compute_fpfh_dev")

    def normals_fpfh_dev(self, d_xyz, n, k, radius, d_normals, d_desc):
        """estimateNormals(k) + computeFPFH(radius) with one neighbour walk (device pointers); same bits as the two calls."""
        _check(self._h, lib().tdv_normals_fpfh_dev(self._h, _ptr(d_xyz), n, k, C.c_float(radius), _ptr(d_normals), _ptr(d_desc)), "tdv_normals_fpfh_dev")

    def radix_sort_pairs_dev(self, d_keys_in, d_keys_out, d_vals_in, d_vals_out, n, end_bit):
        """Stable sort of n (u64 key, u32 value) pairs on the low end_bit bits of the key (device pointers; csrc/sort.hip)."""
        _check(self._h, lib().tdv_radix_sort_pairs_dev(self._h, _ptr(d_keys_in), _ptr(d_keys_out), _ptr(d_vals_in), _ptr(d_vals_out), C.c_size_t(n), end_bit),
               "tdv_radix_sort_pairs_dev")

    def depth_to_cloud_dev(self, d_raw, d_mask, d_bgr, w, h, scale, fx, fy, cx, cy, zmax, d_xyz, d_rgb, capacity,
                           mask_mode=TDV_MASK_THRESHOLD10):
        n = C.c_int()
        _check(self._h, lib().tdv_depth_to_cloud_dev(self._h, _ptr(d_raw), _ptr(d_mask), _ptr(d_bgr), w, h, C.c_float(scale), mask_mode,
                                                     C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(zmax),
                                                     _ptr(d_xyz), _ptr(d_rgb), capacity, C.byref(n)), "tdv_depth_to_cloud_dev")
        return n.value

    def voxel_downsample_dev(self, d_xyz, d_rgb, n, voxel, d_out_xyz, d_out_rgb, capacity, order=TDV_VOXEL_ORDER_FIRST):
        m = C.c_int()
        _check(self._h, lib().tdv_voxel_downsample_dev(self._h, _ptr(d_xyz), _ptr(d_rgb), n, C.c_float(voxel), order, _ptr(d_out_xyz),
                                                       _ptr(d_out_rgb), capacity, C.byref(m)), "tdv_voxel_downsample_dev")
        return m.value


def _voxel_downsample_batch_dev(self, d_xyz, cloud_offsets, voxel, d_out_xyz, pinhole=None):
    """All clouds' voxels (first-occurrence order) in one set of launches; returns the voxel offsets (int32 [n_clouds + 1]).
    pinhole = (fx, fy, cx, cy): the clouds come from depth images with these intrinsics, in row-major pixel order (pixel-window grouping)."""
    off = np.ascontiguousarray(cloud_offsets, np.int32)
    voff = np.zeros(len(off), np.int32)
    if pinhole is None:
        _check(self._h, lib().tdv_voxel_downsample_batch_dev(self._h, _ptr(d_xyz), _ptr(off), len(off) - 1, C.c_float(voxel), _ptr(d_out_xyz), _ptr(voff)),
               "tdv_voxel_downsample_batch_dev")
    else:
        fx, fy, cx, cy = pinhole
        _check(self._h, lib().tdv_voxel_downsample_batch_pinhole_dev(self._h, _ptr(d_xyz), _ptr(off), len(off) - 1, C.c_float(voxel), C.c_float(fx), C.c_float(fy),
                                                                     C.c_float(cx), C.c_float(cy), _ptr(d_out_xyz), _ptr(voff)), "tdv_voxel_downsample_batch_pinhole_dev")
    return voff


def batch_params(width=1280, height=720, scale_to_meters=1000.0, mask_mode=TDV_MASK_THRESHOLD10, fx=900.0, fy=900.0, cx=640.0,
                 cy=360.0, zmax=1.5, voxel_size=0.001, normals_k=30, fpfh_radius_factor=5.0, ransac_max_iterations=100000,
                 ransac_confidence=0.999, icp_distance_factor=0.4, icp_max_iterations=200, point_to_plane=True, seed=42,
                 voxel_order=TDV_VOXEL_ORDER_REFERENCE, n_frames=1, frame_of_instance=None, mask_format=0, mask_width=0, mask_height=0):
    """Defaults = include/pipeline_config.hpp + config/pipeline_config.yaml of the reference; voxel_order defaults to the
    reference's container order (the poses of Pipeline::processInstance).  frame_of_instance: int array or None."""
    p = BatchParamsC(width, height, scale_to_meters, mask_mode, fx, fy, cx, cy, zmax, voxel_size, normals_k, fpfh_radius_factor,
                     ransac_max_iterations, ransac_confidence, icp_distance_factor, icp_max_iterations, int(point_to_plane), seed,
                     voxel_order, n_frames, None, mask_format, mask_width, mask_height)
    if frame_of_instance is not None:
        p._frame_map = np.ascontiguousarray(frame_of_instance, np.int32)   # kept alive by the struct object
        p.frame_of_instance = p._frame_map.ctypes.data
    return p


def _register_batch_dev(self, d_raw, d_bgr, d_masks, n_instances, params, d_model_xyz, d_model_normals, d_model_fpfh, n_model):
    """Batched device-resident Pipeline::processInstance; returns a list of dicts (one per instance)."""
    res = (InstanceResultC * max(n_instances, 1))()
    _check(self._h, lib().tdv_register_batch_dev(self._h, _ptr(d_raw), _ptr(d_bgr), _ptr(d_masks), n_instances, C.byref(params),
                                                 _ptr(d_model_xyz), _ptr(d_model_normals), _ptr(d_model_fpfh), n_model, res),
           "tdv_register_batch_dev")
    # one structured view over the result array instead of a ctypes attribute walk per instance (1,024 instances: 6 ms -> 0.6 ms)
    dt = np.dtype([("T", np.float32, 16), ("fitness", np.float32), ("rmse", np.float32), ("coarse_fitness", np.float32), ("coarse_inliers", np.int32),
                   ("icp_iterations", np.int32), ("n_points", np.int32), ("n_voxels", np.int32), ("status", np.int32)])
    assert dt.itemsize == C.sizeof(InstanceResultC)
    a = np.frombuffer(res, dtype=dt, count=n_instances).copy() if n_instances else np.zeros(0, dt)
    Ts = a["T"].reshape(-1, 4, 4).transpose(0, 2, 1).copy()          # column-major float[16] -> row-major [4, 4]
    out = [dict(T=Ts[i], fitness=a["fitness"][i], rmse=a["rmse"][i], coarse_fitness=a["coarse_fitness"][i], coarse_inliers=int(a["coarse_inliers"][i]),
                icp_iterations=int(a["icp_iterations"][i]), n_points=int(a["n_points"][i]), n_voxels=int(a["n_voxels"][i]), status=int(a["status"][i]))
           for i in range(n_instances)]
    return out


def _prepare_model_dev(self, d_xyz, n, voxel, k, radius_factor, d_out_xyz, d_out_normals, d_out_fpfh, order=TDV_VOXEL_ORDER_REFERENCE):
    m = C.c_int()
    _check(self._h, lib().tdv_prepare_model_dev(self._h, _ptr(d_xyz), n, C.c_float(voxel), order, k, C.c_float(radius_factor), _ptr(d_out_xyz),
                                                _ptr(d_out_normals), _ptr(d_out_fpfh), C.byref(m)), "tdv_prepare_model_dev")
    return m.value


def _depth_to_cloud_batch_dev(self, d_raw, d_masks, d_bgr, n_instances, w, h, scale, fx, fy, cx, cy, zmax, d_xyz, d_rgb, capacity,
                              mask_format=0, mask_mode=TDV_MASK_THRESHOLD10, n_frames=1, frame_of_instance=None):
    """All instances of a scene -> clouds back to back; returns the offsets array (n_instances + 1)."""
    off = np.zeros(n_instances + 1, np.int32)
    fmap = None if frame_of_instance is None else np.ascontiguousarray(frame_of_instance, np.int32)
    st = lib().tdv_depth_to_cloud_batch_dev(self._h, _ptr(d_raw), _ptr(d_masks), _ptr(d_bgr), n_instances, mask_format, n_frames, _ptr(fmap), w, h,
                                            C.c_float(scale), mask_mode, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                            C.c_float(zmax), _ptr(d_xyz), _ptr(d_rgb), C.c_longlong(capacity), _ptr(off))
    _check(self._h, st, "tdv_depth_to_cloud_batch_dev")
    return off


def _broadcast_model(self, comm, root, d_xyz, d_normals, d_fpfh, capacity, n_model):
    """tdv_broadcast_model: comm is an ncclComm_t (int / c_void_p); returns the model's point count on every rank."""
    n = C.c_int(int(n_model))
    _check(self._h, lib().tdv_broadcast_model(self._h, C.c_void_p(comm), int(root), _ptr(d_xyz), _ptr(d_normals), _ptr(d_fpfh), int(capacity), C.byref(n)),
           "tdv_broadcast_model")
    return n.value


def _gather_results(self, comm, local, slots_per_rank, world_size):
    """tdv_gather_results: local = list of InstanceResultC; returns world_size * slots_per_rank InstanceResultC (status -1 = unused slot)."""
    loc = (InstanceResultC * max(len(local), 1))(*local)
    out = (InstanceResultC * max(world_size * slots_per_rank, 1))()
    _check(self._h, lib().tdv_gather_results(self._h, C.c_void_p(comm), loc, len(local), int(slots_per_rank), out), "tdv_gather_results")
    return list(out)[:world_size * slots_per_rank]


Context.depth_to_cloud_batch_dev = _depth_to_cloud_batch_dev
Context.broadcast_model = _broadcast_model
Context.gather_results = _gather_results
Context.register_batch_dev = _register_batch_dev
Context.voxel_downsample_batch_dev = _voxel_downsample_batch_dev
Context.prepare_model_dev = _prepare_model_dev


def sample_triples(n, count, seed=42):
    out = np.empty((count, 3), np.uint64)
    _check(None, lib().tdv_sample_triples(C.c_uint32(seed), C.c_uint64(n), count, _ptr(out)), "tdv_sample_triples")
    return out


def filter_duplicates(poses, min_distance=0.1):
    """Pipeline::filterDuplicates (src/pipeline.cpp:153-180); poses: [n,4,4]."""
    poses = np.asarray(poses, np.float32).reshape(-1, 4, 4)
    cm = np.ascontiguousarray(np.transpose(poses, (0, 2, 1))).reshape(-1, 16)
    out = np.empty_like(cm); m = C.c_int()
    _check(None, lib().tdv_filter_duplicates(_ptr(cm), len(cm), C.c_float(min_distance), _ptr(out), C.byref(m)), "tdv_filter_duplicates")
    return np.transpose(out[:m.value].reshape(-1, 4, 4), (0, 2, 1)).copy()


def load_reference_model(path, capacity=1 << 20):
    """Registration::loadReferenceModel (src/registration.cpp:416-461): returns a PointCloud (empty if the file is missing)."""
    xyz = np.zeros((capacity, 3), np.float32); rgb = np.zeros((capacity, 3), np.float32); n = C.c_int(); hc = C.c_int()
    st = lib().tdv_load_ply_ascii(path.encode(), _ptr(xyz), _ptr(rgb), capacity, C.byref(n), C.byref(hc))
    if st != 0:
        return PointCloud()
    return PointCloud(points=xyz[:n.value].copy(), colors=(rgb[:n.value].copy() if hc.value else None))


def load_mask_png(path):
    """One mask file as Segmentation::loadMasksFromDir reads it (grey PNG, > 10 -> 255): uint8 [h, w], or None if the file
    cannot be decoded here (colour / palette PNG, JPEG, missing)."""
    w = C.c_int(); h = C.c_int()
    if lib().tdv_load_mask_png(path.encode(), None, C.c_longlong(0), C.byref(w), C.byref(h)) != 0:
        return None
    out = np.zeros((h.value, w.value), np.uint8)
    if lib().tdv_load_mask_png(path.encode(), _ptr(out), C.c_longlong(out.size), C.byref(w), C.byref(h)) != 0:
        return None
    return out


def load_masks_from_dir(masks_dir, width, height):
    """Segmentation::loadMasksFromDir (src/segmentation.cpp:12-42): (masks uint8 [n, height, width], n_skipped)."""
    n = C.c_int(); sk = C.c_int()
    _check(None, lib().tdv_load_masks_from_dir(masks_dir.encode(), width, height, None, 0, C.byref(n), C.byref(sk)), "tdv_load_masks_from_dir")
    out = np.zeros((n.value, height, width), np.uint8)
    if n.value:
        _check(None, lib().tdv_load_masks_from_dir(masks_dir.encode(), width, height, _ptr(out), n.value, C.byref(n), C.byref(sk)), "tdv_load_masks_from_dir")
    return out, sk.value


def pose_compose(extrinsics, T):
    e = to_colmajor16(extrinsics); t = to_colmajor16(T); o = np.zeros(16, np.float32)
    _check(None, lib().tdv_pose_compose(_ptr(e), _ptr(t), _ptr(o)), "tdv_pose_compose")
    return from_colmajor16(o)


# --------------------------------------------------------------------------------------------------
# Operator API in the reference's own names.  A thread-local default Context stands in for the
# per-thread state the reference's static methods hide.
_tls = threading.local()


def default_context():
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = Context(0)
        _tls.ctx = ctx
    return ctx


class GPUDepth:
    """include/gpu_depth.hpp:9-13."""

    @staticmethod
    def isCudaAvailable():
        try:
            return device_count() > 0
        except TdvError:
            return False

    @staticmethod
    def preprocess(raw_depth, mask, scale):
        if not GPUDepth.isCudaAvailable():
            raise RuntimeError("CUDA not available")  # src/gpu_impl.cpp:64
        return default_context().depth_preprocess(raw_depth, mask, scale)


class GPUPointCloud:
    """include/gpu_depth.hpp:15-22.  zmax defaults to the reference dispatch's hard-coded 10.0
    (src/gpu_impl.cpp:97); pass config.depth.clipping_max to follow the CPU branch."""

    @staticmethod
    def generate(depth, rgb, fx, fy, cx, cy, zmax=10.0):
        if not GPUDepth.isCudaAvailable():
            return PointCloud()  # src/gpu_impl.cpp:126
        xyz, col = default_context().deproject(depth, rgb, fx, fy, cx, cy, zmax)
        return PointCloud(points=xyz, colors=col)


class GPURegistration:
    """include/gpu_registration.hpp:8-19."""

    @staticmethod
    def isCudaAvailable():
        return GPUDepth.isCudaAvailable()

    @staticmethod
    def icpRefine(source, target, initial_transform, distance_threshold, max_iterations=200):
        if not GPURegistration.isCudaAvailable():
            raise RuntimeError("CUDA not available")  # src/gpu_impl.cpp:258
        tn = target.normals if target.hasNormals() else None
        return default_context().icp(source.points, target.points, tn, initial_transform, distance_threshold, max_iterations, True)


class Registration:
    """include/registration.hpp:32-60, served by the HIP backend."""

    @staticmethod
    def voxelDownsample(cloud, voxel_size, order=TDV_VOXEL_ORDER_REFERENCE):
        col = cloud.colors if cloud.hasColors() else None
        xyz, c = default_context().voxel_downsample(cloud.points, col, voxel_size, order)
        return PointCloud(points=xyz, colors=c)

    @staticmethod
    def estimateNormals(cloud, k=30):
        cloud.normals = default_context().estimate_normals(cloud.points, k)

    @staticmethod
    def computeFPFH(cloud, radius):
        return default_context().compute_fpfh(cloud.points, cloud.normals, radius)

    @staticmethod
    def ransacRegistration(source, target, source_features, target_features, voxel_size, max_iterations=100000, confidence=0.999):
        return default_context().ransac(source.points, target.points, source_features, target_features, None, voxel_size,
                                        max_iterations, confidence)

    @staticmethod
    def icpRefine(source, target, initial_transform, distance_threshold, max_iterations=200, point_to_plane=True):
        tn = target.normals if target.hasNormals() else None
        return default_context().icp(source.points, target.points, tn, initial_transform, distance_threshold, max_iterations,
                                     point_to_plane)
