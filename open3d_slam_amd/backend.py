"""ctypes binding of the C-ABI in include/o3ds_backend.h (libo3ds_backend.so, HIP/gfx950).

This is plumbing only: every function forwards to the shared library.  There is NO CPU
fallback -- if the library is missing or no GPU is present the calls raise (the product
path must fail loudly rather than silently run something else).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("O3DS_BACKEND_LIB") or os.path.join(_PKG, "lib", "libo3ds_backend.so")  # the override is a development aid (instrumented builds)

OK = 0
ERR_INVALID_ARG, ERR_NO_NORMALS, ERR_OOM, ERR_HIP, ERR_BAD_HANDLE, ERR_EMPTY, ERR_CAPACITY = -1, -2, -3, -4, -5, -6, -7
NO_FIELD = C.c_size_t(-1).value  # O3DS_NO_FIELD
COLOR_FIELD_RGB, COLOR_FIELD_INTENSITY = 0, 1
PRECISION_F32, PRECISION_F64 = 0, 1
ICP_POINT_TO_PLANE, ICP_GENERALIZED, ICP_POINT_TO_POINT = 0, 1, 2
CROP_NONE, CROP_MAX_RADIUS, CROP_MIN_RADIUS, CROP_MIN_MAX_RADIUS, CROP_CYLINDER = range(5)


class Crop(C.Structure):
    _fields_ = [("kind", C.c_int32), ("invert", C.c_int32), ("center", C.c_double * 3), ("rmin", C.c_double),
                ("rmax", C.c_double), ("zmin", C.c_double), ("zmax", C.c_double)]


class IcpResult(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("n_corr", C.c_uint64)]


class CloudView(C.Structure):  # o3ds_cloud_view
    _fields_ = [("pts", C.c_void_p), ("nrm", C.c_void_p), ("col", C.c_void_p), ("n", C.c_size_t), ("precision", C.c_int), ("device", C.c_int),
                ("has_box", C.c_int), ("reserved", C.c_int), ("box_min", C.c_double * 3), ("box_max", C.c_double * 3), ("event", C.c_void_p)]


class IcpParams(C.Structure):
    _fields_ = [("max_correspondence_distance", C.c_double), ("max_iteration", C.c_int32), ("method", C.c_int32),
                ("relative_fitness", C.c_double), ("relative_rmse", C.c_double)]


_dp = C.POINTER(C.c_double)
_H = C.c_void_p
_CL = C.c_uint64

# name -> (restype, argtypes); must list every symbol include/o3ds_backend.h declares
class CarvingParams(C.Structure):  # o3ds_carving_params (SpaceCarvingParameters, Parameters.hpp:85-92)
    _fields_ = [("voxel_size", C.c_double), ("max_raytracing_length", C.c_double), ("truncation_distance", C.c_double),
                ("min_dot_product_with_normal", C.c_double)]


OVERLAP_FN = C.CFUNCTYPE(None, C.c_void_p)  # o3ds_overlap_fn

SIGNATURES = {
    "o3ds_create": (C.c_int, [C.c_int, C.POINTER(_H)]),
    "o3ds_destroy": (C.c_int, [_H]),
    "o3ds_last_error": (C.c_char_p, [_H]),
    "o3ds_set_precision": (C.c_int, [_H, C.c_int]),
    "o3ds_synchronize": (C.c_int, [_H]),
    "o3ds_stream": (C.c_void_p, [_H]),
    "o3ds_version": (C.c_char_p, []),
    "o3ds_set_stream": (C.c_int, [_H, C.c_void_p]),
    "o3ds_profile_enable": (C.c_int, [_H, C.c_int]),
    "o3ds_profile_read": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "o3ds_cloud_export_rows_by_owner": (C.c_int, [_H, _CL, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "o3ds_cloud_import_rows": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(_CL)]),
    "o3ds_icp_nn_keys": (C.c_int, [_H, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    "o3ds_icp_accumulate_keys": (C.c_int, [_H, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "o3ds_profile_span": (C.c_int, [_H, C.c_int, C.c_int]),
    "o3ds_profile_span_read": (C.c_int, [_H, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "o3ds_cloud_upload": (C.c_int, [_H, _dp, _dp, C.c_size_t, C.POINTER(_CL)]),
    "o3ds_cloud_upload_f32": (C.c_int, [_H, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(_CL)]),
    "o3ds_pinned_alloc": (C.c_int, [_H, C.c_size_t, C.POINTER(C.c_void_p)]),
    "o3ds_pinned_free": (C.c_int, [_H, C.c_void_p]),
    "o3ds_cloud_wait_ingest": (C.c_int, [_H, _CL]),
    "o3ds_cloud_size_bound": (C.c_int, [_H, _CL, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "o3ds_cloud_free": (C.c_int, [_H, _CL]),
    "o3ds_cloud_size": (C.c_int, [_H, _CL, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "o3ds_cloud_download": (C.c_int, [_H, _CL, _dp, _dp, C.c_size_t]),
    "o3ds_cloud_set_colors": (C.c_int, [_H, _CL, _dp]),
    "o3ds_cloud_has_colors": (C.c_int, [_H, _CL, C.POINTER(C.c_int)]),
    "o3ds_cloud_get_colors": (C.c_int, [_H, _CL, _dp, C.c_size_t]),
    "o3ds_cloud_download_f32": (C.c_int, [_H, _CL, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                          C.c_size_t, C.c_int]),
    "o3ds_cloud_set_colors_from_records": (C.c_int, [_H, _CL, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "o3ds_cloud_build_index": (C.c_int, [_H, _CL, C.c_double, C.c_double]),
    "o3ds_icp_point_to_plane": (C.c_int, [_H, _dp, C.c_size_t, _dp, _dp, C.c_size_t, _dp, C.POINTER(IcpParams),
                                          C.POINTER(IcpResult)]),
    "o3ds_icp_point_to_plane_dev": (C.c_int, [_H, _CL, _CL, C.POINTER(Crop), _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_icp_register_dev": (C.c_int, [_H, _CL, _CL, C.POINTER(Crop), _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_icp_overlap_next": (C.c_int, [_H, OVERLAP_FN, C.c_void_p]),
    "o3ds_icp_pass": (C.c_int, [_H, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "o3ds_icp_pass_finish": (C.c_int, [_H, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(IcpResult)]),
    "o3ds_cloud_undistort": (C.c_int, [_H, _CL, _dp, _dp, C.c_double, C.c_int]),
    "o3ds_dense_map_create": (C.c_int, [_H, C.c_double, C.POINTER(C.c_uint64)]),
    "o3ds_dense_map_free": (C.c_int, [_H, C.c_uint64]),
    "o3ds_dense_map_insert": (C.c_int, [_H, C.c_uint64, _CL, _dp]),
    "o3ds_dense_map_size": (C.c_int, [_H, C.c_uint64, C.POINTER(C.c_size_t)]),
    "o3ds_dense_map_to_cloud": (C.c_int, [_H, C.c_uint64, C.POINTER(_CL)]),
    "o3ds_dense_map_transform": (C.c_int, [_H, C.c_uint64, _dp]),
    "o3ds_dense_map_carve": (C.c_int, [_H, C.c_uint64, _CL, _dp, _dp, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_size_t)]),
    "o3ds_dense_map_count_occupied": (C.c_int, [_H, C.c_uint64, _CL, _dp, C.POINTER(C.c_size_t)]),
    "o3ds_overlap_indices": (C.c_int, [_H, _CL, _CL, _dp, C.c_double, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]),
    "o3ds_map_carve": (C.c_int, [_H, _CL, _CL, _dp, C.POINTER(Crop), C.POINTER(CarvingParams), C.POINTER(C.c_size_t)]),
    "o3ds_map_carve_removed": (C.c_int, [_H, _CL, _CL, _dp, C.POINTER(Crop), C.POINTER(CarvingParams), C.POINTER(C.c_size_t), C.POINTER(_CL)]),
    "o3ds_information_matrix": (C.c_int, [_H, _dp, C.c_size_t, _dp, C.c_size_t, _dp, C.c_double, _dp]),
    "o3ds_information_matrix_dev": (C.c_int, [_H, _CL, _CL, C.POINTER(Crop), _dp, C.c_double, _dp]),
    "o3ds_icp_point_to_point": (C.c_int, [_H, _dp, C.c_size_t, _dp, C.c_size_t, _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_icp_point_to_point_dev": (C.c_int, [_H, _CL, _CL, C.POINTER(Crop), _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_icp_generalized": (C.c_int, [_H, _dp, _dp, C.c_size_t, _dp, _dp, C.c_size_t, _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_icp_generalized_dev": (C.c_int, [_H, _CL, _CL, C.POINTER(Crop), _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_set_gicp_epsilon": (C.c_int, [_H, C.c_double]),
    "o3ds_icp_begin": (C.c_int, [_H, _CL, _CL, C.POINTER(Crop), _dp, C.POINTER(IcpParams)]),
    "o3ds_icp_accumulate": (C.c_int, [_H, C.c_size_t, C.c_size_t, C.c_void_p]),
    "o3ds_icp_update": (C.c_int, [_H, C.c_void_p, C.c_uint64]),
    "o3ds_icp_finish": (C.c_int, [_H, C.POINTER(IcpResult)]),
    "o3ds_icp_done": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "o3ds_cloud_export_view": (C.c_int, [_H, _CL, C.POINTER(CloudView)]),
    "o3ds_cloud_import_view": (C.c_int, [_H, C.POINTER(CloudView), C.POINTER(_CL)]),
    "o3ds_cloud_view_release": (C.c_int, [C.POINTER(CloudView)]),
    "o3ds_comm_unique_id": (C.c_int, [_H, C.c_char_p]),
    "o3ds_comm_init": (C.c_int, [_H, C.c_char_p, C.c_int, C.c_int]),
    "o3ds_comm_attach": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "o3ds_comm_destroy": (C.c_int, [_H]),
    "o3ds_icp_register_sharded": (C.c_int, [_H, C.c_int, _CL, _CL, C.POINTER(Crop), _dp, C.POINTER(IcpParams), C.POINTER(IcpResult)]),
    "o3ds_crop_cloud": (C.c_int, [_H, _CL, C.POINTER(Crop), C.POINTER(_CL)]),
    "o3ds_voxel_down_sample": (C.c_int, [_H, _CL, C.c_double, C.POINTER(_CL)]),
    "o3ds_crop_voxel_down_sample": (C.c_int, [_H, _CL, C.POINTER(Crop), C.c_double, C.POINTER(_CL)]),
    "o3ds_estimate_normals": (C.c_int, [_H, _CL, C.c_double, C.c_int]),
    "o3ds_select_by_index": (C.c_int, [_H, _CL, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(_CL)]),
    "o3ds_random_down_sample": (C.c_int, [_H, _CL, C.c_double, C.c_uint64, C.POINTER(_CL)]),
    "o3ds_transform_cloud": (C.c_int, [_H, _CL, _dp, C.POINTER(_CL)]),
    "o3ds_cloud_append": (C.c_int, [_H, _CL, _CL]),
    "o3ds_cloud_copy_across": (C.c_int, [_H, _H, _CL, C.POINTER(_CL)]),
    "o3ds_voxelize_within_volume": (C.c_int, [_H, _CL, C.c_double, C.POINTER(Crop)]),
    "o3ds_map_insert_scan": (C.c_int, [_H, _CL, _CL, _dp, C.c_double, C.POINTER(Crop), C.c_double]),
}

_lib = None
_EMPTY = (C.c_double * 3)()  # non-NULL placeholder for zero-length clouds


class BackendError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (assert.hpp:13-64 / Open3D LogError)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"o3ds error {code}: {msg}")
        self.code = code


LIB_AB_PATH = os.path.join(_PKG, "lib", "libo3ds_backend_ab.so")  # built with -DO3DS_AB_SWITCHES: the O3DS_* A/B levers exist only there
_lib_ab = None


def load(ab: bool = False):
    """dlopen the HIP backend; raises if it has not been built (no fallback).  ab=True: the twin library whose A/B switches are compiled
    in (tests and experiment scripts; the shipped library does not read them)."""
    global _lib, _lib_ab
    cur = _lib_ab if ab else _lib
    if cur is None:
        path = LIB_AB_PATH if ab and not os.environ.get("O3DS_BACKEND_LIB") else LIB_PATH
        if not os.path.exists(path):
            raise ImportError(f"{path} not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("O3DS_BACKEND_LIB") and not hasattr(L, name):
                continue  # (an instrumented / older build named by the development override may lack the newest entry points)
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        cur = L
        if ab:
            _lib_ab = L
        else:
            _lib = L
    return cur


def make_crop(kind=CROP_NONE, center=(0.0, 0.0, 0.0), rmin=0.0, rmax=0.0, zmin=0.0, zmax=0.0, invert=False) -> Crop:
    c = Crop()
    c.kind, c.invert = int(kind), int(bool(invert))
    c.center[:] = [float(x) for x in center]
    c.rmin, c.rmax, c.zmin, c.zmax = float(rmin), float(rmax), float(zmin), float(zmax)
    return c


def colmajor(T) -> np.ndarray:
    return np.asarray(T, dtype=np.float64).ravel(order="F")  # (a fresh, contiguous array: the transpose flattened row by row)


def from_colmajor(v) -> np.ndarray:
    return np.ndarray((4, 4), np.float64, v, 0, None, "F").copy()  # a view of the 16 doubles as a column-major matrix, copied row-major


def _d(a):
    """(array kept alive by the caller, what ctypes passes for a `const double*`).  `ndarray.ctypes.data_as` costs ~5 us per call -- more than
    everything else a registration call does on the host -- so small writable arrays are handed over as a ctypes array over their buffer (~1 us)."""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.size == 0:  # numpy may hand out a NULL data pointer for empty arrays; the ABI reads NULL as "absent"
        a = np.zeros((0, 3), dtype=np.float64)
        return a, C.cast(_EMPTY, _dp)
    if a.size <= 64 and a.flags.writeable:  # poses, velocities, 6x6 matrices: where the call's own cost is what counts
        return a, (C.c_double * a.size).from_buffer(a)
    return a, a.ctypes.data_as(_dp)


_IDENTITY16 = (C.c_double * 16)(*np.eye(4).ravel())  # init = None of the registration calls (read-only for the library)


class Backend:
    """One handle = one HIP stream + scratch (not re-entrant; one per thread)."""

    def __init__(self, device_id: int = 0, precision: int = PRECISION_F32, ab: bool = False):
        self.lib = load(ab)
        self.h = _H()
        rc = self.lib.o3ds_create(device_id, C.byref(self.h))
        if rc != OK:
            raise BackendError(rc, (self.lib.o3ds_last_error(None) or b"").decode())
        if precision != PRECISION_F32:
            self._ck(self.lib.o3ds_set_precision(self.h, precision))
        self.device_id = device_id
        self.precision = precision

    def close(self):
        if getattr(self, "h", None):
            self.free_pinned()
            self.lib.o3ds_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int):
        if rc != OK:
            raise BackendError(rc, (self.lib.o3ds_last_error(self.h) or b"").decode())

    # -- clouds
    def upload(self, xyz, normals=None) -> int:
        xyz, xp = _d(np.asarray(xyz).reshape(-1, 3))
        nrm, npp = _d(None if normals is None else np.asarray(normals).reshape(-1, 3))
        cid = _CL()
        self._ck(self.lib.o3ds_cloud_upload(self.h, xp, npp, len(xyz), C.byref(cid)))
        return cid.value

    def upload_f32(self, records, off_x: int = 0, off_y: int = 4, off_z: int = 8) -> int:
        """sensor_msgs/PointCloud2-style ingest: `records` is a C-contiguous (n, k) float32 array (x, y, z in the first three
        columns by default) or any (n,)-shaped structured / (n, step)-byte array with float32 x/y/z at the given byte offsets."""
        a = np.ascontiguousarray(records)
        n = a.shape[0]
        step = a.strides[0] if n else (a.dtype.itemsize * (a.shape[1] if a.ndim > 1 else 1))
        cid = _CL()
        self._ck(self.lib.o3ds_cloud_upload_f32(self.h, a.ctypes.data_as(C.c_void_p), n, step, off_x, off_y, off_z, C.byref(cid)))
        return cid.value

    def pinned_records(self, n: int, point_step: int = 16) -> np.ndarray:
        """(n, point_step) bytes of page-locked host memory (o3ds_pinned_alloc) as a numpy array: a sensor / message buffer that
        upload_f32 DMAs from directly and asynchronously instead of copying it through the handle's pinned ring first.  Freed with the
        handle (or free_pinned)."""
        p = C.c_void_p()
        self._ck(self.lib.o3ds_pinned_alloc(self.h, max(n * point_step, 1), C.byref(p)))
        buf = (C.c_uint8 * (n * point_step)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8).reshape(n, point_step)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        return arr

    def free_pinned(self):
        for p in getattr(self, "_pinned", []):
            self.lib.o3ds_pinned_free(self.h, C.c_void_p(p))
        self._pinned = []

    def wait_ingest(self, cid: int):
        self._ck(self.lib.o3ds_cloud_wait_ingest(self.h, cid))

    def size_bound(self, cid: int):
        """(lower, upper) bounds of the size, never waits (o3ds_cloud_size_bound)"""
        lo, up = C.c_size_t(), C.c_size_t()
        self._ck(self.lib.o3ds_cloud_size_bound(self.h, cid, C.byref(lo), C.byref(up)))
        return lo.value, up.value

    def has_normals(self, cid: int) -> bool:
        """without waiting for a size that is still in flight (o3ds_cloud_size with n = NULL)"""
        hn = C.c_int()
        self._ck(self.lib.o3ds_cloud_size(self.h, cid, None, C.byref(hn)))
        return bool(hn.value)

    def free(self, cid: int):
        self._ck(self.lib.o3ds_cloud_free(self.h, cid))

    def size(self, cid: int):
        n, hn = C.c_size_t(), C.c_int()
        self._ck(self.lib.o3ds_cloud_size(self.h, cid, C.byref(n), C.byref(hn)))
        return n.value, bool(hn.value)

    def download(self, cid: int):
        n, hn = self.size(cid)
        xyz = np.empty((n, 3))
        nrm = np.empty((n, 3)) if hn else None
        self._ck(self.lib.o3ds_cloud_download(self.h, cid, xyz.ctypes.data_as(_dp),
                                              nrm.ctypes.data_as(_dp) if hn else None, n))
        return xyz, nrm

    def download_f32(self, cid: int, point_step: int = 16, off_x: int = 0, off_y: int = 4, off_z: int = 8, off_normal: int | None = None,
                     off_rgb: int | None = None, rgb_rounding: int = 0):
        """The cloud as (n, point_step) bytes of float32 records (o3ds_cloud_download_f32): PointCloud2 'xyz' layout by default
        (point_step 32 with off_rgb 16 for a coloured cloud); point_step 24 with off_normal 12 is the row of a binary PCD with
        normals.  rgb_rounding 0 = (int)(255 c) as open3dToRos, 1 = clamp / round as [O3D] ColorToUint8 (PCD)."""
        n, _ = self.size(cid)
        buf = np.zeros((n, point_step), dtype=np.uint8)
        self._ck(self.lib.o3ds_cloud_download_f32(self.h, cid, buf.ctypes.data_as(C.c_void_p), n, point_step, off_x, off_y, off_z,
                                                  NO_FIELD if off_normal is None else off_normal, NO_FIELD if off_rgb is None else off_rgb,
                                                  rgb_rounding))
        return buf

    def set_colors_from_records(self, cid: int, records, off_field: int, kind: int):
        """colours from the PointCloud2 records the cloud was uploaded from (rosToOpen3d with skip_colors = false):
        kind COLOR_FIELD_RGB or COLOR_FIELD_INTENSITY, the field at byte offset off_field"""
        a = np.ascontiguousarray(records)
        n, _ = self.size(cid)
        if a.shape[0] != n:
            raise ValueError("set_colors_from_records: one record per point")
        step = a.strides[0] if n else 16
        self._ck(self.lib.o3ds_cloud_set_colors_from_records(self.h, cid, a.ctypes.data_as(C.c_void_p), step, off_field, kind))

    def set_colors(self, cid: int, rgb):
        """PointCloud::colors_ of a device cloud ((n, 3) doubles; None clears)."""
        if rgb is None:
            self._ck(self.lib.o3ds_cloud_set_colors(self.h, cid, None))
            return
        n, _ = self.size(cid)
        a, ap = _d(np.asarray(rgb).reshape(-1, 3))
        if len(a) != n:
            raise ValueError("set_colors: one colour per point")
        self._ck(self.lib.o3ds_cloud_set_colors(self.h, cid, ap))

    def has_colors(self, cid: int) -> bool:
        hc = C.c_int()
        self._ck(self.lib.o3ds_cloud_has_colors(self.h, cid, C.byref(hc)))
        return bool(hc.value)

    def get_colors(self, cid: int):
        if not self.has_colors(cid):
            return None
        n, _ = self.size(cid)
        rgb = np.empty((n, 3))
        self._ck(self.lib.o3ds_cloud_get_colors(self.h, cid, rgb.ctypes.data_as(_dp), n))
        return rgb

    def build_index(self, cid: int, max_corr_hint: float, cell_size: float = 0.0):
        self._ck(self.lib.o3ds_cloud_build_index(self.h, cid, max_corr_hint, cell_size))

    # -- ICP
    @staticmethod
    def _params(max_corr, max_iter, rel_fitness, rel_rmse, method=ICP_POINT_TO_PLANE) -> IcpParams:
        p = IcpParams()
        p.method = int(method)
        p.max_correspondence_distance = float(max_corr)
        p.max_iteration = int(max_iter)
        p.relative_fitness = float(rel_fitness)
        p.relative_rmse = float(rel_rmse)
        return p

    @staticmethod
    def _result(r: IcpResult) -> dict:
        return dict(transformation=from_colmajor(r.transformation), fitness=r.fitness, inlier_rmse=r.inlier_rmse,
                    iterations=r.iterations, converged=bool(r.converged), n_corr=int(r.n_corr))

    def icp_point_to_plane(self, src, tgt, tgt_normals, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
        src, sp = _d(np.asarray(src).reshape(-1, 3))
        tgt, tp = _d(np.asarray(tgt).reshape(-1, 3))
        nrm, npp = _d(None if tgt_normals is None else np.asarray(tgt_normals).reshape(-1, 3))
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse)
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_point_to_plane(self.h, sp, len(src), tp, npp, len(tgt), ip, C.byref(p), C.byref(out)))
        return self._result(out)

    # o3ds_icp_overlap_next: `overlap_next` (a callable, set by the caller) is handed to the next device-resident registration and runs
    # once its launches are queued, before the host waits for the result -- the stream driver queues the pre-processing of the next scan
    # there (bench.run_stream).  An exception raised inside it is re-raised when the registration has returned.
    overlap_next = None

    def _hand_over_overlap(self):
        fn, self.overlap_next = self.overlap_next, None
        if fn is None:
            return None
        state = {"called": False, "error": None, "fn": fn}

        def cb(_arg):
            state["called"] = True
            try:
                fn()
            except BaseException as e:  # noqa: BLE001 -- crosses a C frame: kept and re-raised by _after_overlap
                state["error"] = e

        state["cb"] = OVERLAP_FN(cb)  # (kept alive until the registration has returned)
        self._ck(self.lib.o3ds_icp_overlap_next(self.h, state["cb"], None))
        return state

    def _after_overlap(self, state):
        if state is None:
            return
        if not state["called"]:  # the registration did not get that far (an error path, the two-launch form): the work is still due
            state["fn"]()
        if state["error"] is not None:
            raise state["error"]

    def icp_point_to_plane_dev(self, source: int, target: int, max_corr, init=None, max_iter=30, rel_fitness=1e-6,
                               rel_rmse=1e-6, target_crop: Crop | None = None):
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse)
        out = IcpResult()
        held = self._hand_over_overlap()
        try:
            self._ck(self.lib.o3ds_icp_point_to_plane_dev(self.h, source, target, C.byref(target_crop) if target_crop else None, ip, C.byref(p),
                                                         C.byref(out)))
        finally:
            self._after_overlap(held)
        return self._result(out)

    def information_matrix(self, src, tgt, max_corr, T=None) -> np.ndarray:
        """[O3D] GetInformationMatrixFromPointClouds on host buffers."""
        src, sp = _d(np.asarray(src).reshape(-1, 3))
        tgt, tp = _d(np.asarray(tgt).reshape(-1, 3))
        T0, ip = _d(colmajor(np.eye(4) if T is None else T))
        out = np.zeros(36)
        self._ck(self.lib.o3ds_information_matrix(self.h, sp, len(src), tp, len(tgt), ip, float(max_corr), out.ctypes.data_as(_dp)))
        return out.reshape(6, 6)

    def information_matrix_dev(self, source: int, target: int, max_corr, T=None, target_crop: Crop | None = None) -> np.ndarray:
        T0, ip = _d(colmajor(np.eye(4) if T is None else T))
        out = np.zeros(36)
        self._ck(self.lib.o3ds_information_matrix_dev(self.h, source, target, C.byref(target_crop) if target_crop else None, ip,
                                                      float(max_corr), out.ctypes.data_as(_dp)))
        return out.reshape(6, 6)

    def icp_point_to_point(self, src, tgt, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
        src, sp = _d(np.asarray(src).reshape(-1, 3))
        tgt, tp = _d(np.asarray(tgt).reshape(-1, 3))
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, ICP_POINT_TO_POINT)
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_point_to_point(self.h, sp, len(src), tp, len(tgt), ip, C.byref(p), C.byref(out)))
        return self._result(out)

    def icp_point_to_point_dev(self, source: int, target: int, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                               target_crop: Crop | None = None):
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, ICP_POINT_TO_POINT)
        out = IcpResult()
        held = self._hand_over_overlap()
        try:
            self._ck(self.lib.o3ds_icp_point_to_point_dev(self.h, source, target, C.byref(target_crop) if target_crop else None, ip, C.byref(p),
                                                         C.byref(out)))
        finally:
            self._after_overlap(held)
        return self._result(out)

    def icp_generalized(self, src, src_normals, tgt, tgt_normals, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
        src, sp = _d(np.asarray(src).reshape(-1, 3))
        sn, snp = _d(None if src_normals is None else np.asarray(src_normals).reshape(-1, 3))
        tgt, tp = _d(np.asarray(tgt).reshape(-1, 3))
        tn, tnp = _d(None if tgt_normals is None else np.asarray(tgt_normals).reshape(-1, 3))
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, ICP_GENERALIZED)
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_generalized(self.h, sp, snp, len(src), tp, tnp, len(tgt), ip, C.byref(p), C.byref(out)))
        return self._result(out)

    def icp_generalized_dev(self, source: int, target: int, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                            target_crop: Crop | None = None):
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, ICP_GENERALIZED)
        out = IcpResult()
        held = self._hand_over_overlap()
        try:
            self._ck(self.lib.o3ds_icp_generalized_dev(self.h, source, target, C.byref(target_crop) if target_crop else None, ip, C.byref(p),
                                                         C.byref(out)))
        finally:
            self._after_overlap(held)
        return self._result(out)

    ICP_SUMS_DOUBLES = 512
    ICP_PASS_MAX_QUERIES = 262144  # O3DS_ICP_PASS_MAX_QUERIES

    def icp_pass(self, first: int, count: int, n_src_total: int, sums_in_ptr: int | None, sums_out_ptr: int, sums_next_ptr: int):
        """fused step-wise form: one kernel; the caller all-reduces sums_out afterwards (device pointers to 512 doubles each)"""
        self._ck(self.lib.o3ds_icp_pass(self.h, first, count, n_src_total, C.c_void_p(sums_in_ptr or 0), C.c_void_p(sums_out_ptr),
                                        C.c_void_p(sums_next_ptr)))

    def icp_pass_finish(self, n_src_total: int, sums_in_ptr: int, sums_scratch_ptr: int):
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_pass_finish(self.h, n_src_total, C.c_void_p(sums_in_ptr), C.c_void_p(sums_scratch_ptr), C.byref(out)))
        return self._result(out)

    def export_view(self, cid: int) -> "CloudView":
        """o3ds_cloud_export_view: what another handle needs to copy this cloud without touching this handle"""
        v = CloudView()
        self._ck(self.lib.o3ds_cloud_export_view(self.h, cid, C.byref(v)))
        return v

    def import_view(self, view: "CloudView") -> int:
        out = _CL()
        self._ck(self.lib.o3ds_cloud_import_view(self.h, C.byref(view), C.byref(out)))
        return out.value

    def release_view(self, view: "CloudView"):
        self.lib.o3ds_cloud_view_release(C.byref(view))

    def icp_register_dev(self, source: int, target: int, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                         target_crop: Crop | None = None, method=ICP_POINT_TO_PLANE):
        """o3ds_icp_register_dev: the estimator chosen by `method` (what the three per-estimator wrappers above call)"""
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, method)
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_register_dev(self.h, source, target, C.byref(target_crop) if target_crop else None, ip, C.byref(p), C.byref(out)))
        return self._result(out)

    # -- sharded registrations inside the library (RCCL through dlopen; o3ds_backend.h)
    SHARD_SOURCE, SHARD_SUBMAP, SHARD_UNION = 0, 1, 2

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._ck(self.lib.o3ds_comm_unique_id(self.h, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._ck(self.lib.o3ds_comm_init(self.h, C.create_string_buffer(unique_id, 128), int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self.lib.o3ds_comm_destroy(self.h))

    def icp_register_sharded(self, partitioning: int, source: int, target: int, max_corr, init=None, max_iter=30, rel_fitness=1e-6,
                             rel_rmse=1e-6, target_crop: Crop | None = None, method=ICP_POINT_TO_PLANE):
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, method)
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_register_sharded(self.h, int(partitioning), source, target, C.byref(target_crop) if target_crop else None, ip,
                                                    C.byref(p), C.byref(out)))
        return self._result(out)

    def set_gicp_epsilon(self, eps: float):
        self._ck(self.lib.o3ds_set_gicp_epsilon(self.h, float(eps)))

    def icp_begin(self, source: int, target: int, max_corr, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6,
                  target_crop: Crop | None = None, method=ICP_POINT_TO_PLANE):
        T0, ip = (None, _IDENTITY16) if init is None else _d(colmajor(init))
        p = self._params(max_corr, max_iter, rel_fitness, rel_rmse, method)
        self._ck(self.lib.o3ds_icp_begin(self.h, source, target, C.byref(target_crop) if target_crop else None, ip, C.byref(p)))

    def icp_accumulate(self, first: int, count: int, d_record_ptr: int):
        self._ck(self.lib.o3ds_icp_accumulate(self.h, first, count, C.c_void_p(d_record_ptr)))

    def icp_nn_keys(self, first: int, count: int, rank: int, d_keys_ptr: int):
        self._ck(self.lib.o3ds_icp_nn_keys(self.h, first, count, rank, C.c_void_p(d_keys_ptr)))

    def icp_accumulate_keys(self, first: int, count: int, rank: int, d_keys_ptr: int, d_record_ptr: int):
        self._ck(self.lib.o3ds_icp_accumulate_keys(self.h, first, count, rank, C.c_void_p(d_keys_ptr), C.c_void_p(d_record_ptr)))

    def icp_update(self, d_record_ptr: int, n_src_total: int):
        self._ck(self.lib.o3ds_icp_update(self.h, C.c_void_p(d_record_ptr), n_src_total))

    def icp_done(self) -> bool:
        d = C.c_int()
        self._ck(self.lib.o3ds_icp_done(self.h, C.byref(d)))
        return bool(d.value)

    def icp_finish(self) -> dict:
        out = IcpResult()
        self._ck(self.lib.o3ds_icp_finish(self.h, C.byref(out)))
        return self._result(out)

    def synchronize(self):
        self._ck(self.lib.o3ds_synchronize(self.h))

    def set_stream(self, hip_stream: int | None):
        self._ck(self.lib.o3ds_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    def profile_enable(self, on):
        """True / 1: hipEvent brackets around every ICP pass launch + tagged spans; 2: tagged spans only; False: off"""
        self._ck(self.lib.o3ds_profile_enable(self.h, int(on)))

    def profile_read(self):
        n, ms = C.c_uint64(), C.c_double()
        self._ck(self.lib.o3ds_profile_read(self.h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def span(self, tag: int):
        """context manager: a tagged pair of hipEvents on the handle's stream around the calls inside (profiling enabled)"""
        be = self

        class _Span:
            def __enter__(self_inner):
                be._ck(be.lib.o3ds_profile_span(be.h, tag, 0))

            def __exit__(self_inner, *exc):
                be._ck(be.lib.o3ds_profile_span(be.h, tag, 1))
                return False

        return _Span()

    def span_read(self, tag: int):
        n, ms = C.c_uint64(), C.c_double()
        self._ck(self.lib.o3ds_profile_span_read(self.h, tag, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    @property
    def stream(self) -> int:
        return int(self.lib.o3ds_stream(self.h) or 0)

    # -- pre-processing
    def crop_cloud(self, cid: int, crop: Crop) -> int:
        out = _CL()
        self._ck(self.lib.o3ds_crop_cloud(self.h, cid, C.byref(crop), C.byref(out)))
        return out.value

    def voxel_down_sample(self, cid: int, voxel: float) -> int:
        out = _CL()
        self._ck(self.lib.o3ds_voxel_down_sample(self.h, cid, voxel, C.byref(out)))
        return out.value

    def crop_voxel_down_sample(self, cid: int, crop: Crop, voxel: float) -> int:
        """crop_cloud followed by voxel_down_sample, bit for bit, in one call (the first two steps of both preprocess chains)."""
        out = _CL()
        self._ck(self.lib.o3ds_crop_voxel_down_sample(self.h, cid, C.byref(crop), voxel, C.byref(out)))
        return out.value

    def estimate_normals(self, cid: int, radius: float, max_nn: int):
        self._ck(self.lib.o3ds_estimate_normals(self.h, cid, radius, max_nn))

    def select_by_index(self, cid: int, idx) -> int:
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        out = _CL()
        self._ck(self.lib.o3ds_select_by_index(self.h, cid, idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(idx), C.byref(out)))
        return out.value

    def random_down_sample(self, cid: int, ratio: float, seed: int) -> int:
        """[O3D] RandomDownSample drawn on the device (o3ds_random_down_sample): the k = int(ratio * n) points with the smallest keys of the
        counter-based generator seeded with `seed`, in cloud order; the input's size may still be in flight, the result's then is too"""
        out = _CL()
        self._ck(self.lib.o3ds_random_down_sample(self.h, cid, float(ratio), int(seed) & 0xFFFFFFFFFFFFFFFF, C.byref(out)))
        return out.value

    # -- map fusion
    def transform_cloud(self, cid: int, T) -> int:
        Tc, tp = _d(colmajor(T))
        out = _CL()
        self._ck(self.lib.o3ds_transform_cloud(self.h, cid, tp, C.byref(out)))
        return out.value

    def cloud_append(self, map_id: int, add_id: int):
        self._ck(self.lib.o3ds_cloud_append(self.h, map_id, add_id))

    def copy_from(self, other: "Backend", cid: int) -> int:
        """a cloud of another handle on the same device as a cloud of this one (device-to-device copy)"""
        out = _CL()
        self._ck(self.lib.o3ds_cloud_copy_across(self.h, other.h, cid, C.byref(out)))
        return out.value

    def voxelize_within_volume(self, map_id: int, voxel: float, crop: Crop):
        self._ck(self.lib.o3ds_voxelize_within_volume(self.h, map_id, voxel, C.byref(crop)))

    def undistort(self, cid: int, lin_vel, ang_vel_rpy, scan_duration: float, clockwise: bool = False):
        """ConstantVelocityMotionCompensation::undistortInputPointCloud, in place on the device cloud."""
        v, vp = _d(np.asarray(lin_vel, dtype=np.float64).reshape(3))
        w, wp = _d(np.asarray(ang_vel_rpy, dtype=np.float64).reshape(3))
        self._ck(self.lib.o3ds_cloud_undistort(self.h, cid, vp, wp, float(scan_duration), int(bool(clockwise))))

    # -- dense voxel map (VoxelizedPointCloud)
    def dense_map_create(self, voxel: float) -> int:
        d = C.c_uint64(0)
        self._ck(self.lib.o3ds_dense_map_create(self.h, float(voxel), C.byref(d)))
        return d.value

    def dense_map_free(self, dm: int):
        self._ck(self.lib.o3ds_dense_map_free(self.h, dm))

    def dense_map_insert(self, dm: int, cloud: int, T=None):
        if T is None:
            self._ck(self.lib.o3ds_dense_map_insert(self.h, dm, cloud, None))
        else:
            Tc, tp = _d(colmajor(T))
            self._ck(self.lib.o3ds_dense_map_insert(self.h, dm, cloud, tp))

    def export_rows_by_owner(self, cid: int, T, voxel: float, world: int, d_rows_ptr: int, d_counts_ptr: int):
        """rows [x y z nx ny nz] of the (placed) cloud grouped by voxel owner into device memory, group sizes into device memory"""
        Tm = None if T is None else colmajor(T).ctypes.data_as(C.POINTER(C.c_double))
        self._ck(self.lib.o3ds_cloud_export_rows_by_owner(self.h, cid, Tm, float(voxel), int(world), C.c_void_p(d_rows_ptr), C.c_void_p(d_counts_ptr)))

    def import_rows(self, d_rows_ptr: int, n: int, has_normals: bool) -> int:
        out = _CL()
        self._ck(self.lib.o3ds_cloud_import_rows(self.h, C.c_void_p(d_rows_ptr), int(n), int(bool(has_normals)), C.byref(out)))
        return int(out.value)

    def dense_map_size(self, dm: int) -> int:
        n = C.c_size_t(0)
        self._ck(self.lib.o3ds_dense_map_size(self.h, dm, C.byref(n)))
        return int(n.value)

    def dense_map_to_cloud(self, dm: int) -> int:
        cid = _CL()
        self._ck(self.lib.o3ds_dense_map_to_cloud(self.h, dm, C.byref(cid)))
        return cid.value

    def dense_map_carve(self, dm: int, scan: int, sensor_position, scan_pose=None, radius=0.1, max_length=20.0, truncation=0.1) -> int:
        """Submap::carve for the dense map; returns the number of removed voxels."""
        sp, spp = _d(np.asarray(sensor_position, dtype=np.float64).reshape(3))
        n = C.c_size_t(0)
        if scan_pose is None:
            self._ck(self.lib.o3ds_dense_map_carve(self.h, dm, scan, None, spp, radius, max_length, truncation, C.byref(n)))
        else:
            Tc, tp = _d(colmajor(scan_pose))
            self._ck(self.lib.o3ds_dense_map_carve(self.h, dm, scan, tp, spp, radius, max_length, truncation, C.byref(n)))
        return int(n.value)

    def dense_map_count_occupied(self, dm: int, cloud: int, T=None) -> int:
        """points of the placed cloud that fall into an occupied voxel (isSwitchingSubmapsConsistant's numerator)"""
        n = C.c_size_t(0)
        if T is None:
            self._ck(self.lib.o3ds_dense_map_count_occupied(self.h, dm, cloud, None, C.byref(n)))
        else:
            Tc, tp = _d(colmajor(T))
            self._ck(self.lib.o3ds_dense_map_count_occupied(self.h, dm, cloud, tp, C.byref(n)))
        return int(n.value)

    def dense_map_transform(self, dm: int, T):
        Tc, tp = _d(colmajor(T))
        self._ck(self.lib.o3ds_dense_map_transform(self.h, dm, tp))

    def overlap_indices(self, source: int, target: int, T=None, voxel: float = 0.5, min_points: int = 1):
        """computeIndicesOfOverlappingPoints: (ascending source indices, ascending target indices) as uint64 arrays."""
        Tc, tp = _d(colmajor(np.eye(4) if T is None else T))
        ns, nt = self.size(source)[0], self.size(target)[0]
        i_s, i_t = np.zeros(max(ns, 1), np.uint64), np.zeros(max(nt, 1), np.uint64)
        c_s, c_t = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.lib.o3ds_overlap_indices(self.h, source, target, tp, float(voxel), int(min_points), i_s.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               C.byref(c_s), i_t.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(c_t)))
        return i_s[: c_s.value].copy(), i_t[: c_t.value].copy()

    def map_carve(self, map_id: int, raw_scan_id: int, T, crop: Crop | None, voxel=0.1, max_length=20.0, truncation=0.1, min_dot=0.5,
                  want_count: bool = True):
        """Submap::carve on the device-resident sparse map; returns the number of removed points (want_count=False: nothing -- the count is
        what a carve of a submap in its persistent form would have to wait for; Submap::carve itself returns void)."""
        Tc, tp = _d(colmajor(T))
        p = CarvingParams(voxel, max_length, truncation, min_dot)
        n = C.c_size_t(0)
        self._ck(self.lib.o3ds_map_carve(self.h, map_id, raw_scan_id, tp, C.byref(crop) if crop else None, C.byref(p), C.byref(n) if want_count else None))
        return int(n.value) if want_count else None

    def map_carve_removed(self, map_id: int, raw_scan_id: int, T, crop: Crop | None, voxel=0.1, max_length=20.0, truncation=0.1, min_dot=0.5):
        """Submap::carve with its toRemove_ cloud: (number of removed points, device cloud of the removed points in map order)."""
        Tc, tp = _d(colmajor(T))
        p = CarvingParams(voxel, max_length, truncation, min_dot)
        n, gone = C.c_size_t(0), _CL()
        self._ck(self.lib.o3ds_map_carve_removed(self.h, map_id, raw_scan_id, tp, C.byref(crop) if crop else None, C.byref(p), C.byref(n), C.byref(gone)))
        return int(n.value), gone.value

    def map_insert_scan(self, map_id: int, scan_id: int, T, map_voxel: float, crop: Crop, max_corr_hint: float = 0.0):
        Tc, tp = _d(colmajor(T))
        self._ck(self.lib.o3ds_map_insert_scan(self.h, map_id, scan_id, tp, map_voxel, C.byref(crop), max_corr_hint))
