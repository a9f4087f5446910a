"""CroppingVolume family of the reference (include/open3d_slam/croppers.hpp:26-47, src/croppers.cpp) on device clouds.
Same class / method names; `crop` runs the HIP compaction kernel (o3ds_crop_cloud)."""
from __future__ import annotations

import enum

import numpy as np

from . import backend as _b
from .parameters import ScanCroppingParameters
from .pointcloud import PointCloud


class CroppingVolumeEnum(enum.IntEnum):  # croppers.hpp:19
    MaxRadius = 0
    MinRadius = 1
    Cylinder = 2
    MinMaxRadius = 3


cropperNames = {e.name: e for e in CroppingVolumeEnum}  # croppers.hpp:21-24


class CroppingVolume:
    """Base class: the whole space (croppers.cpp:49-51)."""

    _kind = _b.CROP_NONE

    def __init__(self):
        self.pose_ = np.eye(4)
        self.isInvertVolume_ = False

    def setScaling(self, scaling: float):  # croppers.cpp:114-116: nothing by default
        pass

    def setIsInvertVolume(self, val: bool):
        self.isInvertVolume_ = bool(val)

    def setPose(self, pose):
        self.pose_ = np.array(pose, dtype=np.float64)

    def _radii(self):
        return dict(rmin=0.0, rmax=0.0, zmin=0.0, zmax=0.0)

    def to_abi(self) -> _b.Crop:
        """o3ds_crop for this volume; only pose_.translation() enters the predicate (croppers.cpp:121-165).  The struct is kept until
        something that enters it changes (a frame of the stream asks for the same four volumes again and again; callers only read it)."""
        pose = self.pose_
        key = (float(pose[0, 3]), float(pose[1, 3]), float(pose[2, 3]), self.isInvertVolume_, tuple(self._radii().values()))
        cached = getattr(self, "_abi_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, _b.make_crop(self._kind, center=pose[:3, 3], invert=self.isInvertVolume_, **self._radii()))
            self._abi_cache = cached
        return cached[1]

    def isWithinVolume(self, p) -> bool:
        """Scalar predicate (host); the bulk path is `crop`."""
        d = np.asarray(p, dtype=np.float64) - self.pose_[:3, 3]
        r = self._radii()
        k = self._kind
        if k == _b.CROP_MAX_RADIUS:
            inside = np.linalg.norm(d) <= r["rmax"]
        elif k == _b.CROP_MIN_RADIUS:
            inside = np.linalg.norm(d) >= r["rmin"]
        elif k == _b.CROP_MIN_MAX_RADIUS:
            inside = r["rmin"] <= np.linalg.norm(d) <= r["rmax"]
        elif k == _b.CROP_CYLINDER:
            inside = r["zmin"] <= float(p[2]) <= r["zmax"] and np.linalg.norm(d[:2]) <= r["rmax"]
        else:
            return True
        return (not inside) if self.isInvertVolume_ else bool(inside)

    def crop(self, cloud: PointCloud) -> PointCloud:
        """CroppingVolume::crop (croppers.cpp:76-106): stable compaction of points (+normals)."""
        return PointCloud(cloud.be, cloud.be.crop_cloud(cloud.id, self.to_abi()))

    def contains(self, other: "CroppingVolume") -> bool:
        """True when every point `other` keeps is kept by this volume too (same kind, same centre, neither inverted, radii nested): then
        cropping `other`'s output with this volume returns it unchanged -- crop(crop(x, V1), V2) = crop(x, V1) for V1 inside V2."""
        if self._kind == _b.CROP_NONE:
            return not self.isInvertVolume_  # the base volume keeps everything (croppers.cpp:49-55), inverted nothing
        if type(self) is not type(other) or self.isInvertVolume_ or other.isInvertVolume_:
            return False
        a_, b_ = self.pose_, other.pose_
        if not (a_[0, 3] == b_[0, 3] and a_[1, 3] == b_[1, 3] and a_[2, 3] == b_[2, 3]):
            return False
        a, b = self._radii(), other._radii()
        if self._kind == _b.CROP_MAX_RADIUS:
            return a["rmax"] >= b["rmax"]
        if self._kind == _b.CROP_MIN_RADIUS:
            return a["rmin"] <= b["rmin"]
        if self._kind == _b.CROP_MIN_MAX_RADIUS:
            return a["rmin"] <= b["rmin"] and a["rmax"] >= b["rmax"]
        if self._kind == _b.CROP_CYLINDER:
            return a["rmax"] >= b["rmax"] and a["zmin"] <= b["zmin"] and a["zmax"] >= b["zmax"]
        return False


class MinMaxRadiusCroppingVolume(CroppingVolume):  # croppers.cpp:119-128
    _kind = _b.CROP_MIN_MAX_RADIUS

    def __init__(self, radiusMin: float = 0.0, radiusMax: float = 1e4):
        super().__init__()
        self.radiusMin_, self.radiusMax_ = radiusMin, radiusMax

    def setParameters(self, radiusMin, radiusMax):
        self.radiusMin_, self.radiusMax_ = radiusMin, radiusMax

    def _radii(self):
        return dict(rmin=self.radiusMin_, rmax=self.radiusMax_, zmin=0.0, zmax=0.0)


class MaxRadiusCroppingVolume(CroppingVolume):  # croppers.cpp:134-141
    _kind = _b.CROP_MAX_RADIUS

    def __init__(self, radius: float = 1e4):
        super().__init__()
        self.radius_ = radius

    def setParameters(self, radius):
        self.radius_ = radius

    def _radii(self):
        return dict(rmin=0.0, rmax=self.radius_, zmin=0.0, zmax=0.0)


class MinRadiusCroppingVolume(CroppingVolume):  # croppers.cpp:147-155
    _kind = _b.CROP_MIN_RADIUS

    def __init__(self, radius: float = 0.0):
        super().__init__()
        self.radius_ = radius

    def setParameters(self, radius):
        self.radius_ = radius

    def _radii(self):
        return dict(rmin=self.radius_, rmax=0.0, zmin=0.0, zmax=0.0)


class CylinderCroppingVolume(CroppingVolume):  # croppers.cpp:161-171
    _kind = _b.CROP_CYLINDER

    def __init__(self, radius: float = 1e4, minZ: float = -1e4, maxZ: float = 1e4):
        super().__init__()
        self.radius_, self.minZ_, self.maxZ_ = radius, minZ, maxZ

    def setParameters(self, radius, minZ, maxZ):
        self.radius_, self.minZ_, self.maxZ_ = radius, minZ, maxZ

    def _radii(self):
        return dict(rmin=0.0, rmax=self.radius_, zmin=self.minZ_, zmax=self.maxZ_)


def croppingVolumeFactory(p_or_type, p: ScanCroppingParameters | None = None) -> CroppingVolume:
    """croppers.cpp:20-47: by ScanCroppingParameters (name lookup) or by (enum, parameters)."""
    if p is None:
        p = p_or_type
        if p.cropperName_ not in cropperNames:
            raise RuntimeError("Unknown cropper type")  # std::map::at would throw std::out_of_range
        kind = cropperNames[p.cropperName_]
    else:
        kind = p_or_type
    if kind == CroppingVolumeEnum.Cylinder:
        return CylinderCroppingVolume(p.croppingMaxRadius_, p.croppingMinZ_, p.croppingMaxZ_)
    if kind == CroppingVolumeEnum.MinRadius:
        return MinRadiusCroppingVolume(p.croppingMinRadius_)
    if kind == CroppingVolumeEnum.MaxRadius:
        return MaxRadiusCroppingVolume(p.croppingMaxRadius_)
    if kind == CroppingVolumeEnum.MinMaxRadius:
        return MinMaxRadiusCroppingVolume(p.croppingMinRadius_, p.croppingMaxRadius_)
    raise RuntimeError("Unknown cropper type")
