"""ctypes binding of oracle/_ref/libo3dslam_ref.so: open3d_slam's OWN hot-path sources (croppers.cpp, helpers.cpp, Voxel.cpp,
VoxelHashMap.cpp, MotionCompensation.cpp, Transform.cpp, math.cpp, ...) compiled unchanged from the reference checkout against the
stand-in headers of oracle/ref_build/shim (oracle/ref_build/Makefile, oracle/ref_build/README.md).

TEST INFRASTRUCTURE ONLY, like everything under oracle/: tests/ check the CPU restatement (oracle/o3d_oracle.c) and the device path
against it; nothing under open3d_slam_amd/ may import it.  What it pins is the reference's own logic (open3d_slam code), not Open3D's
algorithms and not Eigen's rounding -- those are absent from /root/reference and from this image.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("O3DS_REF_LIB") or os.path.join(HERE, "_ref", "libo3dslam_ref.so")  # (the override: the same units built with other flags, tests/test_oracle_vs_reference.py)
# the same units with integration/open3d_slam_o3ds.patch applied, linked against the HIP backend: needs a GPU to run
LIB_PATCHED = os.path.join(HERE, "_ref", "libo3dslam_ref_patched.so")
REFERENCE = os.environ.get("O3DS_REFERENCE_DIR", "/root/reference")

CROP_NONE, CROP_MAX_RADIUS, CROP_MIN_RADIUS, CROP_MIN_MAX_RADIUS, CROP_CYLINDER = 0, 1, 2, 3, 4  # the oracle's numbering

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_lib = None


def sources_present() -> bool:
    return os.path.isfile(os.path.join(REFERENCE, "open3d_slam", "open3d_slam", "src", "croppers.cpp"))


def build(force: bool = False) -> str | None:
    """compile the reference's units where they lie (only when the checkout is present); returns the library path or None"""
    if sources_present():
        cmd = ["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(HERE, "ref_build"), f"REF={REFERENCE}"] + (["-B"] if force else [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle/_ref build failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
    return LIB if os.path.isfile(LIB) else None


def available() -> bool:
    return os.path.isfile(LIB) or sources_present()


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB) and build() is None:
            raise RuntimeError("oracle/_ref/libo3dslam_ref.so is neither built nor buildable here (no reference checkout)")
        L = C.CDLL(LIB)
        L.ref_crop.restype = C.c_size_t
        L.ref_crop.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp, _dp, C.c_size_t, _i64p, _dp, _dp, _dp]
        L.ref_voxelize_within_cropping_volume.restype = C.c_size_t
        L.ref_voxelize_within_cropping_volume.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp, _dp,
                                                          C.c_size_t, _dp, _dp, _dp]
        L.ref_transform.restype = C.c_size_t
        L.ref_transform.argtypes = [_dp, _dp, _dp, _dp, C.c_size_t, _dp, _dp, C.POINTER(C.c_int)]
        L.ref_carved_idxs.restype = C.c_size_t
        L.ref_carved_idxs.argtypes = [_dp, C.c_size_t, _dp, _dp, _dp, C.c_size_t, _i64p, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, _i64p]
        L.ref_overlap.restype = None
        L.ref_overlap.argtypes = [_dp, C.c_size_t, _dp, C.c_size_t, _dp, C.c_double, C.c_size_t, _i64p, C.POINTER(C.c_size_t), _i64p, C.POINTER(C.c_size_t)]
        L.ref_dense_fuse.restype = C.c_size_t
        L.ref_dense_fuse.argtypes = [_dp, _dp, C.c_size_t, C.c_double, C.c_int, _dp, _dp, _dp, _i32p, _i32p]
        L.ref_dense_carve_keys.restype = C.c_size_t
        L.ref_dense_carve_keys.argtypes = [_dp, C.c_size_t, _dp, _dp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _i32p, C.c_size_t]
        L.ref_voxel_idx.restype = None
        L.ref_voxel_idx.argtypes = [_dp, C.c_double, _i32p]
        L.ref_is_valid_color.restype = C.c_int
        L.ref_is_valid_color.argtypes = [_dp]
        L.ref_icp_max_correspondence_distance.restype = C.c_double
        L.ref_icp_max_correspondence_distance.argtypes = [C.c_double]
        L.ref_information_matrix_max_correspondence_distance.restype = C.c_double
        L.ref_information_matrix_max_correspondence_distance.argtypes = [C.c_double]
        L.ref_undistort.restype = None
        L.ref_undistort.argtypes = [_dp, C.c_size_t, _dp, _dp, C.c_double, C.c_double, C.c_int, _dp, _dp]
        _lib = L
    return _lib


def _d(a, shape=(-1, 3)):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(shape)
    return a, a.ctypes.data_as(_dp)


def _opt(a):
    if a is None:
        return None, None
    return _d(a)


def _pose(T):
    T = np.eye(4) if T is None else np.asarray(T, dtype=np.float64).reshape(4, 4)
    v = np.ascontiguousarray(T.T.reshape(-1))  # column-major
    return v, v.ctypes.data_as(_dp)


def crop(pts, kind, rmin=0.0, rmax=0.0, zmin=0.0, zmax=0.0, pose=None, invert=False, nrm=None, col=None):
    """CroppingVolume::getIndicesWithinVolume and ::crop (croppers.cpp:65-106): (indices, points, normals | None, colours | None)"""
    pts, pp = _d(pts)
    nr, np_ = _opt(nrm)
    co, cp = _opt(col)
    n = len(pts)
    Tv, Tp = _pose(pose)
    idx = np.empty(max(n, 1), np.int64)
    op, on, oc = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
    k = lib().ref_crop(kind, rmin, rmax, zmin, zmax, Tp, int(bool(invert)), pp, np_, cp, n, idx.ctypes.data_as(_i64p), op.ctypes.data_as(_dp),
                       on.ctypes.data_as(_dp), oc.ctypes.data_as(_dp))
    assert k != 2 ** 64 - 1, "crop() and getIndicesWithinVolume() disagree"
    return idx[:k].copy(), op[:k].copy(), (on[:k].copy() if nrm is not None else None), (oc[:k].copy() if col is not None else None)


def voxelize_within_cropping_volume(pts, voxel, kind, rmin=0.0, rmax=0.0, zmin=0.0, zmax=0.0, pose=None, invert=False, nrm=None, col=None):
    """voxelizeWithinCroppingVolume (helpers.cpp:115-183) in the reference's own output order: (points, normals | None, colours | None)"""
    pts, pp = _d(pts)
    nr, np_ = _opt(nrm)
    co, cp = _opt(col)
    n = len(pts)
    Tv, Tp = _pose(pose)
    op, on, oc = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
    m = lib().ref_voxelize_within_cropping_volume(voxel, kind, rmin, rmax, zmin, zmax, Tp, int(bool(invert)), pp, np_, cp, n, op.ctypes.data_as(_dp),
                                                  on.ctypes.data_as(_dp), oc.ctypes.data_as(_dp))
    return op[:m].copy(), (on[:m].copy() if nrm is not None else None), (oc[:m].copy() if col is not None else None)


def transform(T, pts, nrm=None, col=None):
    """o3d_slam::transform (helpers.cpp:273-305): (points, normals | None, result still has colours).  For a T within 1e-4 of the identity the
    reference returns 2 n points: the input followed by the transformed copies (and the colours, kept at n entries, no longer count)."""
    pts, pp = _d(pts)
    nr, np_ = _opt(nrm)
    co, cp = _opt(col)
    Tv, Tp = _pose(T)
    n = len(pts)
    op, on = np.empty((2 * max(n, 1), 3)), np.empty((2 * max(n, 1), 3))
    hc = C.c_int()
    m = lib().ref_transform(Tp, pp, np_, cp, n, op.ctypes.data_as(_dp), on.ctypes.data_as(_dp), C.byref(hc))
    return op[:m].copy(), (on[:m].copy() if nrm is not None else None), bool(hc.value)


def carved_idxs(scan, sensor, map_pts, map_nrm, subset=None, voxel=0.1, max_length=20.0, truncation=0.1, min_dot=0.5):
    """getIdxsOfCarvedPoints (helpers.cpp:221-271): sorted indices of the carved map points"""
    scan, sp = _d(scan)
    sen, snp = _d(sensor, (3,))
    mp, mpp = _d(map_pts)
    mn, mnp = _opt(map_nrm)
    out = np.empty(max(len(mp), 1), np.int64)
    if subset is None:
        k = lib().ref_carved_idxs(sp, len(scan), snp, mpp, mnp, len(mp), None, 0, voxel, max_length, truncation, min_dot, out.ctypes.data_as(_i64p))
    else:
        sub = np.ascontiguousarray(subset, dtype=np.int64)
        k = lib().ref_carved_idxs(sp, len(scan), snp, mpp, mnp, len(mp), sub.ctypes.data_as(_i64p), len(sub), voxel, max_length, truncation, min_dot,
                                  out.ctypes.data_as(_i64p))
    return np.sort(out[:k])


def overlap_indices(src, tgt, T=None, voxel=0.5, min_points=1):
    """computeIndicesOfOverlappingPoints (helpers.cpp:307-332): (sorted source indices, sorted target indices)"""
    src, sp = _d(src)
    tgt, tp = _d(tgt)
    Tv, Tp = _pose(T)
    os_, ot = np.empty(max(len(src), 1), np.int64), np.empty(max(len(tgt), 1), np.int64)
    ns, nt = C.c_size_t(), C.c_size_t()
    lib().ref_overlap(sp, len(src), tp, len(tgt), Tp, voxel, int(min_points), os_.ctypes.data_as(_i64p), C.byref(ns), ot.ctypes.data_as(_i64p), C.byref(nt))
    return np.sort(os_[: ns.value]), np.sort(ot[: nt.value])


def dense_fuse(pts, nrm, voxel, batches=1, T_after=None):
    """VoxelizedPointCloud::insert (as `batches` consecutive scans) [+ ::transform(T_after)] + toPointCloud (Voxel.cpp:49-114), in the hash
    map's order: (means, mean normals | None, counts, voxel keys -- which a transform leaves as they were)"""
    pts, pp = _d(pts)
    nr, np_ = _opt(nrm)
    n = len(pts)
    op, on = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
    cnt, keys = np.empty(max(n, 1), np.int32), np.empty((max(n, 1), 3), np.int32)
    Tp = None if T_after is None else _pose(T_after)
    m = lib().ref_dense_fuse(pp, np_, n, voxel, int(batches), None if Tp is None else Tp[1], op.ctypes.data_as(_dp), on.ctypes.data_as(_dp),
                             cnt.ctypes.data_as(_i32p), keys.ctypes.data_as(_i32p))
    assert m != 2 ** 64 - 1
    return op[:m].copy(), (on[:m].copy() if nrm is not None else None), cnt[:m].copy(), keys[:m].copy()


def dense_carve_keys(scan, sensor, map_pts, voxel, radius=0.1, max_length=20.0, truncation=0.1, dedup_scan=True):
    """Submap::carve for the dense map (Submap.cpp:126-136): the voxel keys (k x 3 int32, unordered) getKeysOfCarvedPoints returns"""
    scan, sp = _d(scan)
    sen, snp = _d(sensor, (3,))
    mp, mpp = _d(map_pts)
    cap = 4 * max(len(mp), 1) + 64
    out = np.empty((cap, 3), np.int32)
    k = lib().ref_dense_carve_keys(sp, len(scan), snp, mpp, len(mp), voxel, radius, max_length, truncation, int(bool(dedup_scan)), out.ctypes.data_as(_i32p), cap)
    assert k <= cap
    return out[:k].copy()


def submap_dense(scans, poses, voxel=0.1, crop_rmax=15.0, carve_every=10, radius=0.1, max_length=20.0, truncation=0.1, T_after=None):
    """Submap::insertScanDenseMap over raw scans (equal sizes, sensor frame) at the given poses, carving asked for every time [+ Submap::transform]:
    (voxel means sorted by key, keys, counts, number of voxels after every scan)"""
    L = lib()
    L.ref_submap_dense.restype = C.c_size_t
    L.ref_submap_dense.argtypes = [_dp, C.c_size_t, C.c_int, _dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp, _i32p, _i32p,
                                   C.POINTER(C.c_size_t), C.c_size_t]
    a = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.float64).reshape(-1, 3) for s in scans]))
    f, n = a.shape[0], a.shape[1]
    P = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).reshape(4, 4).T.reshape(-1) for T in poses]))
    cap = f * n + 16
    op, ok, oc = np.empty((cap, 3)), np.empty((cap, 3), np.int32), np.empty(cap, np.int32)
    sizes = (C.c_size_t * f)()
    Tp = None if T_after is None else _pose(T_after)
    m = L.ref_submap_dense(a.ctypes.data_as(_dp), n, f, P.ctypes.data_as(_dp), voxel, crop_rmax, int(carve_every), radius, max_length, truncation,
                           None if Tp is None else Tp[1], op.ctypes.data_as(_dp), ok.ctypes.data_as(_i32p), oc.ctypes.data_as(_i32p), sizes, cap)
    o = np.lexsort((ok[:m, 2], ok[:m, 1], ok[:m, 0]))
    return op[:m][o], ok[:m][o], oc[:m][o], [int(x) for x in sizes]


def voxel_idx(p, voxel):
    p, pp = _d(p, (3,))
    out = np.empty(3, np.int32)
    lib().ref_voxel_idx(pp, voxel, out.ctypes.data_as(_i32p))
    return out


def undistort(pts, finish_xyz, finish_rpy, dt, scan_duration, clockwise=False):
    """ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-139) for a sensor that moved from identity to
    (finish_xyz, finish_rpy) within dt: (moved points, the estimated [linear velocity, angular velocity rpy])"""
    pts, pp = _d(pts)
    x, xp = _d(finish_xyz, (3,))
    r, rp = _d(finish_rpy, (3,))
    out, vel = np.empty_like(pts), np.empty(6)
    lib().ref_undistort(pp, len(pts), xp, rp, float(dt), float(scan_duration), int(bool(clockwise)), out.ctypes.data_as(_dp), vel.ctypes.data_as(_dp))
    return out, vel


# ---- the reference's own frame loop (Odometry.cpp, Mapper.cpp, ScanToMapRegistration.cpp, Submap.cpp, SubmapCollection.cpp, run unchanged;
#      the Open3D algorithms underneath are served by the oracle: oracle/ref_build/open3d_served_by_oracle.cpp)
class SlamParams(C.Structure):
    _fields_ = [("odo_voxel", C.c_double), ("odo_ratio", C.c_double), ("odo_rmin", C.c_double), ("odo_rmax", C.c_double), ("odo_max_corr", C.c_double),
                ("odo_knn_radius", C.c_double), ("odo_knn", C.c_int32), ("odo_max_iter", C.c_int32),
                ("map_voxel", C.c_double), ("map_ratio", C.c_double), ("map_rmin", C.c_double), ("map_rmax", C.c_double), ("map_max_corr", C.c_double),
                ("map_knn_radius", C.c_double), ("min_refinement_fitness", C.c_double), ("min_movement", C.c_double), ("map_knn", C.c_int32),
                ("map_max_iter", C.c_int32),
                ("builder_voxel", C.c_double), ("builder_rmin", C.c_double), ("builder_rmax", C.c_double),
                ("carve_voxel", C.c_double), ("carve_max_length", C.c_double), ("carve_truncation", C.c_double), ("carve_min_dot", C.c_double),
                ("carve_every_n_scans", C.c_int32), ("submap_radius", C.c_double), ("generalized", C.c_int32)]


class ReferenceSlam:
    """LidarOdometry + Mapper of the reference, fed one scan at a time (SlamWrapper's two workers, one after the other)."""

    def __init__(self, mp, op, carve_every_n_scans=10, submap_radius=20.0, min_movement=0.0, carving=(0.1, 20.0, 0.1, 0.5), patched=False, quiet=True):
        """mp / op: open3d_slam_amd.parameters.MapperParameters / OdometryParameters (MinMaxRadius croppers, point-to-plane);
        carving = (voxel, max ray length, truncation, min dot) of SpaceCarvingParameters (Parameters.hpp:85-92 defaults);
        patched: the build with integration/open3d_slam_o3ds.patch applied, whose registration / pre-processing / map fusion run on the
        GPU through libo3ds_backend.so (GPU box only); quiet: open3d_slam's own std::cout reports are dropped"""
        if patched:
            if not os.path.isfile(LIB_PATCHED):
                build()
            L = C.CDLL(LIB_PATCHED)
        else:
            L = lib()
        L.ref_set_quiet.restype = None
        L.ref_set_quiet.argtypes = [C.c_int]
        L.ref_set_quiet(1 if quiet else 0)
        L.ref_slam_run_stream.restype = C.c_int
        L.ref_slam_run_stream.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_double, C.c_int, _dp, _dp, C.POINTER(C.c_size_t), _dp]
        L.ref_slam_create.restype = C.c_void_p
        L.ref_slam_create.argtypes = [C.POINTER(SlamParams)]
        L.ref_slam_free.argtypes = [C.c_void_p]
        L.ref_slam_add_scan.restype = C.c_int
        L.ref_slam_add_scan.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _dp, _dp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ref_slam_map.restype = C.c_size_t
        L.ref_slam_map.argtypes = [C.c_void_p, _dp, _dp]
        L.ref_slam_preprocessed_scan.restype = C.c_size_t
        L.ref_slam_preprocessed_scan.argtypes = [C.c_void_p, _dp, _dp]
        oi, mi = op.scanMatcher_.icp_, mp.scanMatcher_.icp_
        q = SlamParams(op.scanProcessing_.voxelSize_, op.scanProcessing_.downSamplingRatio_, op.scanProcessing_.cropper_.croppingMinRadius_,
                       op.scanProcessing_.cropper_.croppingMaxRadius_, oi.maxCorrespondenceDistance_, oi.maxDistanceKnn_, oi.knn_, oi.maxNumIter_,
                       mp.scanProcessing_.voxelSize_, mp.scanProcessing_.downSamplingRatio_, mp.scanProcessing_.cropper_.croppingMinRadius_,
                       mp.scanProcessing_.cropper_.croppingMaxRadius_, mi.maxCorrespondenceDistance_, mi.maxDistanceKnn_,
                       mp.scanMatcher_.minRefinementFitness_, min_movement, mi.knn_, mi.maxNumIter_,
                       mp.mapBuilder_.mapVoxelSize_, mp.mapBuilder_.cropper_.croppingMinRadius_, mp.mapBuilder_.cropper_.croppingMaxRadius_,
                       carving[0], carving[1], carving[2], carving[3], int(carve_every_n_scans), float(submap_radius),
                       int(getattr(mp.scanMatcher_.scanToMapRegType_, "name", "") == "GeneralizedIcp"))
        self.L = L
        self.h = L.ref_slam_create(C.byref(q))

    def close(self):
        if self.h:
            self.L.ref_slam_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add_scan(self, pts, t_seconds):
        """returns (status, odomToRangeSensor, mapToRangeSensor, points in the active submap, number of submaps)"""
        pts, pp = _d(pts)
        o, m = np.empty(16), np.empty(16)
        n_map, n_sub = C.c_size_t(), C.c_size_t()
        rc = self.L.ref_slam_add_scan(self.h, pp, len(pts), float(t_seconds), o.ctypes.data_as(_dp), m.ctypes.data_as(_dp), C.byref(n_map), C.byref(n_sub))
        return rc, o.reshape(4, 4).T.copy(), m.reshape(4, 4).T.copy(), int(n_map.value), int(n_sub.value)

    def map(self):
        n = self.L.ref_slam_map(self.h, None, None)
        p, nn = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
        self.L.ref_slam_map(self.h, p.ctypes.data_as(_dp), nn.ctypes.data_as(_dp))
        return p[:n].copy(), nn[:n].copy()

    def preprocessed_scan(self):
        n = self.L.ref_slam_preprocessed_scan(self.h, None, None)
        p, nn = np.empty((max(n, 1), 3)), np.empty((max(n, 1), 3))
        self.L.ref_slam_preprocessed_scan(self.h, p.ctypes.data_as(_dp), nn.ctypes.data_as(_dp))
        return p[:n].copy(), nn[:n].copy()

    def run_stream(self, scans32, dt=0.1, threads=False, lead=None):
        """all frames through the reference's two workers, timed inside the library: (accepted frames, per-frame mapToRangeSensor,
        per-frame odomToRangeSensor, total ms, points in the active submap)"""
        a = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.float32).reshape(-1, 3) for s in scans32]))
        f, n = a.shape[0], a.shape[1]
        poses = np.empty((f, 32))
        ms, n_map = C.c_double(), C.c_size_t()
        workers = np.zeros(2)
        ok = self.L.ref_slam_run_stream(self.h, a.ctypes.data_as(C.POINTER(C.c_float)), n, f, float(dt), (int(lead) + 1 if lead else 1) if threads else 0, poses.ctypes.data_as(_dp),
                                        C.byref(ms), C.byref(n_map), workers.ctypes.data_as(_dp))
        self.ms_workers = {"odometry": float(workers[0]), "mapping": float(workers[1])}  # serial mode, frames 1..
        M = poses[:, :16].reshape(f, 4, 4).transpose(0, 2, 1).copy()
        O = poses[:, 16:].reshape(f, 4, 4).transpose(0, 2, 1).copy()
        return ok, M, O, float(ms.value), int(n_map.value)
