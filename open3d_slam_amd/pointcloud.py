"""Device-resident stand-in for open3d::geometry::PointCloud at the hot-path seams (typedefs.hpp:24).
Holds an o3ds_cloud id; points_/normals_ download lazily.  Method names follow the Open3D members the
reference touches on this path (HasNormals, IsEmpty, points_, normals_)."""
from __future__ import annotations

import os

import numpy as np


class PointCloud:
    def __init__(self, be, cid: int, owns: bool = True):
        self.be, self.id, self._owns = be, cid, owns
        self._refs = 1          # holders of this device cloud (retain / release)
        self._pre_memo = None   # pre-processed versions of this (raw) scan: shared_preprocess
        self._known_nonempty = False  # IsEmpty() has seen points in it (see there)

    def retain(self) -> "PointCloud":
        """One more holder of the same device cloud: release() frees it when the last one lets go.  (open3d_slam hands its clouds
        around as shared_ptr; the mirror's explicit release() needs the count spelt out.)"""
        self._refs += 1
        return self

    @classmethod
    def from_numpy(cls, be, points, normals=None, colors=None) -> "PointCloud":
        cloud = cls(be, be.upload(np.asarray(points, dtype=np.float64).reshape(-1, 3), normals))
        if colors is not None:
            be.set_colors(cloud.id, colors)
        return cloud

    @classmethod
    def from_pointcloud2(cls, be, records, off_x: int = 0, off_y: int = 4, off_z: int = 8, fourth_field: tuple | None = None) -> "PointCloud":
        """open3d_conversions::rosToOpen3d (open3d_conversions.cpp:59-88) without the host-side widening: float32 x/y/z records go
        to the device as they are (o3ds_cloud_upload_f32).  fourth_field = ("rgb" | "intensity", byte offset) reproduces what the
        reference does with a fourth field when skip_colors is false (every caller): colours from the rgb bytes / 255, or the first
        byte of the intensity field three times."""
        cloud = cls(be, be.upload_f32(records, off_x, off_y, off_z))
        if fourth_field is not None:
            name, off = fourth_field
            kind = {"rgb": 0, "intensity": 1}[name]
            be.set_colors_from_records(cloud.id, records, off, kind)
        return cloud

    def to_pointcloud2(self) -> np.ndarray:
        """open3d_conversions::open3dToRos (open3d_conversions.cpp:19-53): the `data` member of the message, narrowed on the device.
        Without colours (n, 16) bytes = float32 x, y, z + 4 bytes of padding (fields "xyz"); with colours (n, 32) bytes, the packed
        rgb field at 16 (fields "xyz", "rgb"), bytes = (int)(255 * colour)."""
        if self.HasColors():
            return self.be.download_f32(self.id, 32, 0, 4, 8, None, 16, 0)
        return self.be.download_f32(self.id, 16, 0, 4, 8, None)

    def __len__(self) -> int:
        return self.be.size(self.id)[0]

    def IsEmpty(self) -> bool:
        """decided by what is known of the size without waiting for it where that suffices (o3ds_cloud_size_bound).  A cloud seen to
        hold points keeps holding them (the mirror's clouds are made once; the map only grows): remembered, not asked again -- except
        where points are taken away in place (Submap.carve forgets it: forget_size)."""
        if self._known_nonempty:
            return False
        lo, up = self.be.size_bound(self.id)
        if up == 0:
            return True
        if lo > 0 or len(self) > 0:
            self._known_nonempty = True
            return False
        return True

    def forget_size(self):
        """after an operation that removes points from this cloud in place"""
        self._known_nonempty = False

    def HasNormals(self) -> bool:
        """[O3D] points_.size() > 0 && normals_.size() == points_.size().  Whether the device cloud carries normals is known at once; its
        size may still be in flight (o3ds_cloud_size), so it is only asked for when there are normals to speak of."""
        return self.be.has_normals(self.id) and not self.IsEmpty()

    def HasColors(self) -> bool:
        return len(self) > 0 and self.be.has_colors(self.id)

    @property
    def colors_(self):
        return self.be.get_colors(self.id)

    @property
    def points_(self) -> np.ndarray:
        return self.be.download(self.id)[0]

    @property
    def normals_(self):
        return self.be.download(self.id)[1]

    def release(self):
        if self._refs > 1:
            self._refs -= 1
            return
        self._refs = 0
        if self._pre_memo:
            memo, self._pre_memo = self._pre_memo, None
            for c in memo.values():
                c.release()
        if self._owns and self.id:
            try:
                self.be.free(self.id)
            except Exception:
                pass
            self.id = 0

    def __del__(self):
        try:
            if getattr(self.be, "h", None):
                self._refs = 1  # (the object is going away: whatever holders forgot to release() cannot use it any more)
                self.release()
        except Exception:
            pass


# [O3D] PointCloud::SelectByIndex(indices) marks the listed indices in a mask and walks the cloud once: the selection comes out in CLOUD
# order whatever the order of the list (PointCloud.cpp of v0.15.1, restated from the upstream source -- unpinned like every Open3D row).
# SURVEY A.7 reads it as "output in shuffled order"; the two readings differ only in the order of the kept points (summation orders
# downstream), and both are kept selectable: False = the selection in the order of the shuffled list.
SELECT_BY_INDEX_KEEPS_CLOUD_ORDER = os.environ.get("O3DS_SELECT_SHUFFLED", "0") == "0"


def random_down_sample(cloud: "PointCloud", ratio: float, rng=None, shuffle_at_full_ratio: bool = False) -> "PointCloud":
    """[O3D] PointCloud::RandomDownSample as open3d_slam calls it (Odometry.cpp:29, ScanToMapRegistration.cpp:39): shuffle the
    indices 0..n-1, keep the first int(ratio * n), SelectByIndex (SELECT_BY_INDEX_KEEPS_CLOUD_ORDER: the kept points in cloud order, or in
    the order of the shuffled list).  Open3D seeds a fresh mt19937 from std::random_device per call, so the reference is not reproducible
    here; `rng` (a numpy Generator) pins the list: under the default reading of SelectByIndex (cloud order) one 64-bit seed per call and the
    draw happens on the device (o3ds_random_down_sample); under the other reading the shuffled list itself.  With ratio >= 1 every point is kept -- in cloud order the cloud itself; under the
    other reading a PERMUTATION of it, which `shuffle_at_full_ratio` reproduces when a test wants it.  Consumes `cloud`."""
    if ratio >= 1.0 and (SELECT_BY_INDEX_KEEPS_CLOUD_ORDER or not shuffle_at_full_ratio):
        return cloud  # (before the size is asked for: it may still be in flight on the device, o3ds_cloud_size)
    if SELECT_BY_INDEX_KEEPS_CLOUD_ORDER:
        # drawn on the device (o3ds_random_down_sample): the size of the cloud is not asked for, nothing is shuffled or uploaded -- the host
        # draw below is a 55 000-element permutation and a wait per call, 0.9 ms of a 1.6 ms frame of the shipped configuration.  One
        # 64-bit seed per call from the caller's generator names the subset (oracle/pipeline.py draw_keep restates it)
        if rng is None:
            rng = np.random.default_rng()
        seed = int(rng.integers(0, 2**64, dtype=np.uint64))
        out = PointCloud(cloud.be, cloud.be.random_down_sample(cloud.id, ratio, seed))
        cloud.release()
        return out
    n = len(cloud)
    if n == 0:
        return cloud
    if rng is None:
        rng = np.random.default_rng()
    keep = rng.permutation(n)[: int(min(ratio, 1.0) * n)]
    if SELECT_BY_INDEX_KEEPS_CLOUD_ORDER:
        keep = np.sort(keep)
    out = PointCloud(cloud.be, cloud.be.select_by_index(cloud.id, keep))
    cloud.release()
    return out


SHARE_PREPROCESS = os.environ.get("O3DS_SHARE_PREPROCESS", "1") != "0"


def shared_preprocess(raw: "PointCloud", crop_abi, voxel_size: float, cloud_registration) -> "PointCloud":
    """crop -> voxelize -> estimateNormalsOrCovariancesIfNeeded, the first three lines of BOTH LidarOdometry::preprocess
    (Odometry.cpp:25-30) and ScanToMapIcp::preprocess (ScanToMapRegistration.cpp:35-40).  The shipped configuration gives the two the
    same cropping volume, voxel size and normal-estimation parameters (parameter_structure_definitions.lua: both `scan_processing`
    blocks and the map builder's `scan_cropping` are copies of one table), so on one raw scan they compute the same cloud twice.
    Here the second caller gets the first caller's cloud: the result is remembered ON the raw scan, keyed by every parameter that
    enters it, and dies with it.  Each caller holds its own reference (retain / release) and applies its own RandomDownSample
    afterwards, as in the reference.  O3DS_SHARE_PREPROCESS=0 computes it twice, as the reference does."""
    key = (bytes(crop_abi) if crop_abi is not None else b"", float(voxel_size), type(cloud_registration).__name__,
           getattr(cloud_registration, "knnNormalEstimation_", None), getattr(cloud_registration, "maxRadiusNormalEstimation_", None))
    if SHARE_PREPROCESS and raw._pre_memo and key in raw._pre_memo:
        return raw._pre_memo[key].retain()
    be = raw.be
    vox = PointCloud(be, be.crop_voxel_down_sample(raw.id, crop_abi, voxel_size))
    cloud_registration.estimateNormalsOrCovariancesIfNeeded(vox)
    if SHARE_PREPROCESS:
        if raw._pre_memo is None:
            raw._pre_memo = {}
        raw._pre_memo[key] = vox.retain()  # (the memo's own reference, released with the raw scan)
    return vox
