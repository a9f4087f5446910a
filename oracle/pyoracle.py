"""ctypes loader for the CPU oracle (oracle/libo3d_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
PARITY UNPINNED for the Open3D algorithms, pinned for open3d_slam's own code (see oracle/o3d_oracle.h, oracle/ref_build/README.md).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("O3DS_ORACLE_LIB") or os.path.join(_HERE, "libo3d_oracle.so")  # the override serves A/B runs of oracle builds


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "o3d_oracle.c")
    if os.environ.get("O3DS_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class IcpResult(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("n_corr", C.c_uint64)]


class Crop(C.Structure):
    _fields_ = [("kind", C.c_int32), ("invert", C.c_int32), ("center", C.c_double * 3), ("rmin", C.c_double),
                ("rmax", C.c_double), ("zmin", C.c_double), ("zmax", C.c_double)]


CROP_NONE, CROP_MAX_RADIUS, CROP_MIN_RADIUS, CROP_MIN_MAX_RADIUS, CROP_CYLINDER = range(5)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [_dp, C.c_size_t]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_search_hybrid.restype = C.c_int
        L.orc_kdtree_search_hybrid.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, _ip, _dp]
        L.orc_kdtree_search_knn.restype = C.c_int
        L.orc_kdtree_search_knn.argtypes = [C.c_void_p, _dp, C.c_int, _ip, _dp]
        L.orc_evaluate.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _ip, _dp, _dp, _dp, C.POINTER(C.c_uint64)]
        L.orc_compute_jtj_jtr.argtypes = [_dp, C.c_size_t, _dp, _dp, _ip, _dp, _dp, _dp]
        L.orc_solve_update.restype = C.c_int
        L.orc_solve_update.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_vector6_to_matrix4.argtypes = [_dp, _dp]
        L.orc_transform_points.argtypes = [_dp, C.c_size_t, _dp]
        L.orc_transform_normals.argtypes = [_dp, C.c_size_t, _dp]
        L.orc_icp_point_to_plane.restype = C.c_int
        L.orc_icp_point_to_plane.argtypes = [_dp, C.c_size_t, _dp, _dp, C.c_size_t, C.c_void_p, C.c_double, _dp, C.c_int,
                                             C.c_double, C.c_double, C.POINTER(IcpResult)]
        L.orc_icp_generalized.restype = C.c_int
        L.orc_icp_generalized.argtypes = [_dp, _dp, C.c_size_t, _dp, _dp, C.c_size_t, C.c_void_p, C.c_double, _dp, C.c_int, C.c_double,
                                          C.c_double, C.c_double, C.POINTER(IcpResult)]
        L.orc_icp_point_to_point.restype = C.c_int
        L.orc_icp_point_to_point.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, C.c_void_p, C.c_double, _dp, C.c_int, C.c_double, C.c_double,
                                             C.POINTER(IcpResult)]
        L.orc_umeyama_update.restype = C.c_int
        L.orc_umeyama_update.argtypes = [_dp, C.c_size_t, _dp, _ip, _dp]
        L.orc_svd3.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_information_matrix.restype = C.c_int
        L.orc_information_matrix.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, C.c_void_p, C.c_double, _dp, _dp]
        L.orc_carve_flags.restype = C.c_size_t
        L.orc_carve_flags.argtypes = [_dp, C.c_size_t, _dp, _dp, _dp, C.c_size_t, C.POINTER(C.c_int64), C.c_size_t, C.c_double, C.c_double,
                                      C.c_double, C.c_double, C.POINTER(C.c_uint8)]
        L.orc_overlap_flags.restype = None
        L.orc_overlap_flags.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, _dp, C.c_double, C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.orc_dense_fuse.restype = C.c_size_t
        L.orc_dense_fuse.argtypes = [_dp, _dp, C.c_size_t, C.c_double, _dp, _dp, _ip]
        L.orc_undistort.restype = None
        L.orc_undistort.argtypes = [_dp, C.c_size_t, _dp, _dp, C.c_double, C.c_int]
        L.orc_voxel_key.restype = C.c_int64
        L.orc_voxel_key.argtypes = [_dp, C.c_double]
        L.orc_dense_carve.restype = C.c_size_t
        L.orc_dense_carve.argtypes = [_dp, C.c_size_t, _dp, C.POINTER(C.c_int64), C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.POINTER(C.c_uint8)]
        L.orc_gicp_jtj_jtr.argtypes = [_dp, _dp, C.c_size_t, _dp, _dp, _ip, _dp, _dp]
        L.orc_covariance_from_normal.argtypes = [_dp, C.c_double, _dp]
        L.orc_estimate_normals.argtypes = [_dp, C.c_size_t, C.c_double, C.c_int, _dp]
        L.orc_estimate_normals_ex.argtypes = [_dp, C.c_size_t, C.c_double, C.c_int, C.c_int, _dp]
        L.orc_estimate_normals_ex.restype = None
        L.orc_fast_eigen3x3_min_evec.argtypes = [_dp, _dp]
        L.orc_acos.restype = C.c_double
        L.orc_acos.argtypes = [C.c_double]
        L.orc_cos.restype = C.c_double
        L.orc_cos.argtypes = [C.c_double]
        L.orc_voxel_down_sample.restype = C.c_size_t
        L.orc_voxel_down_sample.argtypes = [_dp, _dp, C.c_size_t, C.c_double, _dp, _dp]
        L.orc_crop_indices.restype = C.c_size_t
        L.orc_crop_indices.argtypes = [_dp, C.c_size_t, C.POINTER(Crop), C.POINTER(C.c_int64)]
        L.orc_voxelize_within_volume.restype = C.c_size_t
        L.orc_voxelize_within_volume.argtypes = [_dp, _dp, C.c_size_t, C.c_double, C.POINTER(Crop), _dp, _dp,
                                                 C.POINTER(C.c_size_t)]
        L.orc_voxelize_within_volume_colors.restype = C.c_size_t
        L.orc_voxelize_within_volume_colors.argtypes = [_dp, _dp, C.c_size_t, C.c_double, C.POINTER(Crop), _dp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def colmajor(T) -> np.ndarray:
    """numpy 4x4 (row-major view) -> 16 doubles column-major."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).ravel()


def from_colmajor(v) -> np.ndarray:
    return np.array(v, dtype=np.float64).reshape(4, 4).T.copy()


def make_crop(kind=CROP_NONE, center=(0, 0, 0), rmin=0.0, rmax=0.0, zmin=0.0, zmax=0.0, invert=False) -> Crop:
    c = Crop()
    c.kind, c.invert = int(kind), int(bool(invert))
    c.center[:] = [float(x) for x in center]
    c.rmin, c.rmax, c.zmin, c.zmax = float(rmin), float(rmax), float(zmin), float(zmax)
    return c


class KDTree:
    def __init__(self, pts):
        self.pts, p = _d(pts)
        self.h = lib().orc_kdtree_build(p, len(self.pts))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h)
            self.h = None

    def search_hybrid(self, q, radius, max_nn):
        q, qp = _d(q)
        idx = np.empty(max_nn, np.int32)
        d2 = np.empty(max_nn, np.float64)
        k = lib().orc_kdtree_search_hybrid(self.h, qp, radius, max_nn, idx.ctypes.data_as(_ip), d2.ctypes.data_as(_dp))
        return idx[:k], d2[:k]

    def search_knn(self, q, k):
        q, qp = _d(q)
        idx = np.empty(k, np.int32)
        d2 = np.empty(k, np.float64)
        kk = lib().orc_kdtree_search_knn(self.h, qp, k, idx.ctypes.data_as(_ip), d2.ctypes.data_as(_dp))
        return idx[:kk], d2[:kk]


def evaluate(tree: KDTree, src, max_corr):
    src, sp = _d(src)
    n = len(src)
    corr = np.empty(n, np.int32)
    d2 = np.empty(n, np.float64)
    fit, rmse, nc = C.c_double(), C.c_double(), C.c_uint64()
    lib().orc_evaluate(tree.h, sp, n, max_corr, corr.ctypes.data_as(_ip), d2.ctypes.data_as(_dp), C.byref(fit), C.byref(rmse),
                       C.byref(nc))
    return corr, d2, fit.value, rmse.value, nc.value


def compute_jtj_jtr(src, tgt, nrm, corr):
    src, sp = _d(src)
    tgt, tp = _d(tgt)
    nrm, npp = _d(nrm)
    corr = np.ascontiguousarray(corr, np.int32)
    JTJ = np.zeros(36)
    JTr = np.zeros(6)
    r2 = C.c_double()
    lib().orc_compute_jtj_jtr(sp, len(src), tp, npp, corr.ctypes.data_as(_ip), JTJ.ctypes.data_as(_dp), JTr.ctypes.data_as(_dp),
                              C.byref(r2))
    return JTJ.reshape(6, 6), JTr, r2.value


def solve_update(JTJ, JTr):
    JTJ, jp = _d(np.asarray(JTJ).reshape(36))
    JTr, rp = _d(JTr)
    U = np.zeros(16)
    x = np.zeros(6)
    lib().orc_solve_update(jp, rp, U.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return from_colmajor(U), x


def vector6_to_matrix4(x):
    x, xp = _d(x)
    U = np.zeros(16)
    lib().orc_vector6_to_matrix4(xp, U.ctypes.data_as(_dp))
    return from_colmajor(U)


def transform_points(pts, T):
    out = np.array(pts, dtype=np.float64, order="C", copy=True)
    Tc, tp = _d(colmajor(T))
    lib().orc_transform_points(out.ctypes.data_as(_dp), len(out), tp)
    return out


def transform_normals(nrm, T):
    out = np.array(nrm, dtype=np.float64, order="C", copy=True)
    Tc, tp = _d(colmajor(T))
    lib().orc_transform_normals(out.ctypes.data_as(_dp), len(out), tp)
    return out


def icp_point_to_plane(src, tgt, nrm, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, tree: KDTree | None = None):
    src, sp = _d(src)
    tgt, tp = _d(tgt)
    nrm, npp = _d(nrm)
    init = np.eye(4) if init is None else init
    Tc, ip = _d(colmajor(init))
    out = IcpResult()
    rc = lib().orc_icp_point_to_plane(sp, len(src), tp, npp, len(tgt), tree.h if tree is not None else None, max_corr, ip,
                                      max_iter, rel_fitness, rel_rmse, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_icp_point_to_plane rc={rc}")
    return dict(transformation=from_colmajor(out.transformation), fitness=out.fitness, inlier_rmse=out.inlier_rmse,
                iterations=out.iterations, converged=bool(out.converged), n_corr=int(out.n_corr))


def icp_point_to_point(src, tgt, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, tree: KDTree | None = None):
    src, sp = _d(src)
    tgt, tp = _d(tgt)
    init = np.eye(4) if init is None else init
    Tc, ip = _d(colmajor(init))
    out = IcpResult()
    rc = lib().orc_icp_point_to_point(sp, len(src), tp, len(tgt), tree.h if tree is not None else None, max_corr, ip, max_iter,
                                      rel_fitness, rel_rmse, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_icp_point_to_point rc={rc}")
    return dict(transformation=from_colmajor(out.transformation), fitness=out.fitness, inlier_rmse=out.inlier_rmse,
                iterations=out.iterations, converged=bool(out.converged), n_corr=int(out.n_corr))


def umeyama_update(P, tgt, corr):
    P, pp = _d(P)
    tgt, tp = _d(tgt)
    corr = np.ascontiguousarray(corr, np.int32)
    U = np.zeros(16)
    lib().orc_umeyama_update(pp, len(P), tp, corr.ctypes.data_as(_ip), U.ctypes.data_as(_dp))
    return from_colmajor(U)


def svd3(A):
    A, ap = _d(np.asarray(A, dtype=np.float64).reshape(9))
    U, d, V = np.zeros(9), np.zeros(3), np.zeros(9)
    lib().orc_svd3(ap, U.ctypes.data_as(_dp), d.ctypes.data_as(_dp), V.ctypes.data_as(_dp))
    return U.reshape(3, 3), d, V.reshape(3, 3)


def information_matrix(src, tgt, max_corr, T=None, tree: KDTree | None = None):
    src, sp = _d(src)
    tgt, tp = _d(tgt)
    Tc, ip = _d(colmajor(np.eye(4) if T is None else T))
    out = np.zeros(36)
    rc = lib().orc_information_matrix(sp, len(src), tp, len(tgt), tree.h if tree is not None else None, max_corr, ip, out.ctypes.data_as(_dp))
    if rc != 0:
        raise RuntimeError(f"orc_information_matrix rc={rc}")
    return out.reshape(6, 6)


def carve_flags(scan, sensor, map_pts, map_nrm, subset, voxel=0.1, max_length=20.0, truncation=0.1, min_dot=0.5):
    """getIdxsOfCarvedPoints (helpers.cpp:235-271): boolean mask over the map of the points a scan carves away."""
    scan, sp = _d(scan)
    sensor, snp = _d(np.asarray(sensor, dtype=np.float64).reshape(3))
    mp, mpp = _d(map_pts)
    mn, mnp = (None, None) if map_nrm is None else _d(map_nrm)
    subset = np.ascontiguousarray(subset, dtype=np.int64)
    flags = np.zeros(len(mp), dtype=np.uint8)
    lib().orc_carve_flags(sp, len(scan), snp, mpp, mnp, len(mp), subset.ctypes.data_as(C.POINTER(C.c_int64)), len(subset), voxel, max_length,
                          truncation, min_dot, flags.ctypes.data_as(C.POINTER(C.c_uint8)))
    return flags.astype(bool)


def overlap_indices(src, tgt, T=None, voxel=0.5, min_points=1):
    """computeIndicesOfOverlappingPoints (helpers.cpp:307-332): (ascending source indices, ascending target indices)."""
    src, sp = _d(src)
    tgt, tp = _d(tgt)
    Tc, ip = _d(colmajor(np.eye(4) if T is None else T))
    fs, ft = np.zeros(len(src), np.uint8), np.zeros(len(tgt), np.uint8)
    lib().orc_overlap_flags(sp, len(src), tp, len(tgt), ip, voxel, int(min_points), fs.ctypes.data_as(C.POINTER(C.c_uint8)),
                            ft.ctypes.data_as(C.POINTER(C.c_uint8)))
    return np.flatnonzero(fs), np.flatnonzero(ft)


def dense_fuse(pts, nrm, voxel):
    """VoxelizedPointCloud insert(s) + toPointCloud (Voxel.cpp:66-114): (voxel means, mean normals or None, counts)."""
    pts, pp = _d(pts)
    nr, npp = (None, None) if nrm is None else _d(nrm)
    op, on, cnt = np.zeros((max(len(pts), 1), 3)), np.zeros((max(len(pts), 1), 3)), np.zeros(max(len(pts), 1), np.int32)
    m = lib().orc_dense_fuse(pp, npp, len(pts), voxel, op.ctypes.data_as(_dp), on.ctypes.data_as(_dp) if nrm is not None else None,
                             cnt.ctypes.data_as(_ip))
    return op[:m].copy(), (on[:m].copy() if nrm is not None else None), cnt[:m].copy()


def undistort(pts, lin_vel, ang_vel_rpy, scan_duration, clockwise=False):
    """ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-139); returns the moved copy."""
    out = np.array(pts, dtype=np.float64, order="C", copy=True).reshape(-1, 3)
    v, vp = _d(np.asarray(lin_vel, dtype=np.float64).reshape(3))
    w, wp = _d(np.asarray(ang_vel_rpy, dtype=np.float64).reshape(3))
    lib().orc_undistort(out.ctypes.data_as(_dp), len(out), vp, wp, float(scan_duration), int(bool(clockwise)))
    return out


def dense_carve(scan, sensor, voxel_points, voxel, radius=0.1, max_length=20.0, truncation=0.1):
    """Submap::carve for the dense map (Submap.cpp:126-136): boolean mask over `voxel_points` (one representative point per occupied
    voxel of the map, e.g. the voxel means) of the voxels a scan removes."""
    scan, sp = _d(scan)
    sensor, snp = _d(np.asarray(sensor, dtype=np.float64).reshape(3))
    vp = np.ascontiguousarray(voxel_points, dtype=np.float64).reshape(-1, 3)
    keys = np.array([lib().orc_voxel_key(vp[i].ctypes.data_as(_dp), voxel) for i in range(len(vp))], dtype=np.int64)
    rem = np.zeros(len(vp), np.uint8)
    rc = lib().orc_dense_carve(sp, len(scan), snp, keys.ctypes.data_as(C.POINTER(C.c_int64)), len(keys), voxel, radius, max_length, truncation,
                               rem.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc == 2 ** 64 - 1:
        raise ValueError("dense_carve: radius and voxel must be > 0 (a zero radius makes the reference's ray step zero: it never terminates)")
    return rem.astype(bool)


def icp_generalized(src, src_nrm, tgt, tgt_nrm, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6, epsilon=1e-3,
                    tree: KDTree | None = None):
    src, sp = _d(src)
    sn, snp = (None, None) if src_nrm is None else _d(src_nrm)
    tgt, tp = _d(tgt)
    tn, tnp = (None, None) if tgt_nrm is None else _d(tgt_nrm)
    init = np.eye(4) if init is None else init
    Tc, ip = _d(colmajor(init))
    out = IcpResult()
    rc = lib().orc_icp_generalized(sp, snp, len(src), tp, tnp, len(tgt), tree.h if tree is not None else None, max_corr, ip, max_iter,
                                   rel_fitness, rel_rmse, epsilon, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_icp_generalized rc={rc}")
    return dict(transformation=from_colmajor(out.transformation), fitness=out.fitness, inlier_rmse=out.inlier_rmse,
                iterations=out.iterations, converged=bool(out.converged), n_corr=int(out.n_corr))


def covariance_from_normal(nrm, epsilon=1e-3):
    n, npp = _d(np.asarray(nrm, dtype=np.float64).reshape(3))
    cov = np.zeros(9)
    lib().orc_covariance_from_normal(npp, epsilon, cov.ctypes.data_as(_dp))
    return cov.reshape(3, 3)


def gicp_jtj_jtr(src, src_cov, tgt, tgt_cov, corr):
    src, sp = _d(src)
    sc, scp = _d(np.asarray(src_cov).reshape(-1, 9))
    tgt, tp = _d(tgt)
    tc, tcp = _d(np.asarray(tgt_cov).reshape(-1, 9))
    corr = np.ascontiguousarray(corr, np.int32)
    JTJ = np.zeros(36)
    JTr = np.zeros(6)
    lib().orc_gicp_jtj_jtr(sp, scp, len(src), tp, tcp, corr.ctypes.data_as(_ip), JTJ.ctypes.data_as(_dp), JTr.ctypes.data_as(_dp))
    return JTJ.reshape(6, 6), JTr


def estimate_normals(pts, radius, max_nn):
    pts, pp = _d(pts)
    out = np.empty_like(pts)
    lib().orc_estimate_normals(pp, len(pts), radius, max_nn, out.ctypes.data_as(_dp))
    return out


def estimate_normals_knn_raw(pts, max_nn=20):
    """[O3D] EstimateNormals(KDTreeSearchParamKNN(max_nn)) alone -- InitializePointCloudForGeneralizedICP on a cloud without normals"""
    pts, pp = _d(pts)
    out = np.empty_like(pts)
    lib().orc_estimate_normals_ex(pp, len(pts), 0.0, max_nn, 1, out.ctypes.data_as(_dp))
    return out


def fast_eigen3x3_min_evec(cov):
    cov, cp = _d(np.asarray(cov).reshape(9))
    out = np.zeros(3)
    lib().orc_fast_eigen3x3_min_evec(cp, out.ctypes.data_as(_dp))
    return out


def voxel_down_sample(pts, voxel, nrm=None):
    pts, pp = _d(pts)
    n = len(pts)
    out = np.empty((max(n, 1), 3))
    if nrm is not None:
        nrm, npp = _d(nrm)
        on = np.empty((max(n, 1), 3))
        m = lib().orc_voxel_down_sample(pp, npp, n, voxel, out.ctypes.data_as(_dp), on.ctypes.data_as(_dp))
        return out[:m].copy(), on[:m].copy()
    m = lib().orc_voxel_down_sample(pp, None, n, voxel, out.ctypes.data_as(_dp), None)
    return out[:m].copy()


def crop_indices(pts, crop: Crop):
    pts, pp = _d(pts)
    idx = np.empty(max(len(pts), 1), np.int64)
    k = lib().orc_crop_indices(pp, len(pts), C.byref(crop), idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return idx[:k].copy()


def voxelize_within_volume(pts, nrm, voxel, crop: Crop):
    pts, pp = _d(pts)
    n = len(pts)
    out = np.empty((max(n, 1), 3))
    npass = C.c_size_t()
    if nrm is not None:
        nrm, npp = _d(nrm)
        on = np.empty((max(n, 1), 3))
        m = lib().orc_voxelize_within_volume(pp, npp, n, voxel, C.byref(crop), out.ctypes.data_as(_dp), on.ctypes.data_as(_dp),
                                             C.byref(npass))
        return out[:m].copy(), on[:m].copy(), int(npass.value)
    m = lib().orc_voxelize_within_volume(pp, None, n, voxel, C.byref(crop), out.ctypes.data_as(_dp), None, C.byref(npass))
    return out[:m].copy(), None, int(npass.value)


def voxelize_within_volume_colors(pts, col, voxel, crop: Crop):
    """colours of voxelize_within_volume's output, in its order (last point of a voxel wins)"""
    pts, pp = _d(pts)
    col, cp = _d(col)
    n = len(pts)
    out = np.empty((max(n, 1), 3))
    m = lib().orc_voxelize_within_volume_colors(pp, cp, n, voxel, C.byref(crop), out.ctypes.data_as(_dp))
    return out[:m].copy()
