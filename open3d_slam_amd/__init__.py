"""open3d_slam_amd -- MI355X (gfx950) scan-matching / map-fusion backend for open3d_slam.

Hot path only: point-to-plane ICP, scan pre-processing, submap voxel merge/crop,
as hand-written HIP kernels behind the C-ABI in include/o3ds_backend.h.  `backend` is the ctypes
binding; `registration`, `croppers`, `submap` mirror the reference's CloudRegistration /
ScanToMapRegistration / CroppingVolume / Submap interfaces on top of it.
"""
__all__ = ["backend", "synthetic"]
