"""Egress of maps: saveToFile (open3d_slam/open3d_slam/src/output.cpp:39-47) and the assembled map
(Mapper::getAssembledMapPointCloud, src/Mapper.cpp:183-208; its voxelisation for display, SlamWrapperRos.cpp:229-231).

saveToFile in the reference copies the cloud and hands it to [O3D] io::WritePointCloudToPCD with default options, which writes
a binary, uncompressed PCD v0.7 whose rows are float32 x y z (+ normal_x normal_y normal_z when the cloud has normals,
and a float-packed rgb field when it has colours).  Here the rows are produced on the device (o3ds_cloud_download_f32) and written as they arrive."""
from __future__ import annotations

import numpy as np

from .pointcloud import PointCloud


def _pcd_header(n: int, has_normals: bool, has_colors: bool = False) -> bytes:
    fields = ["x", "y", "z"] + (["normal_x", "normal_y", "normal_z"] if has_normals else []) + (["rgb"] if has_colors else [])
    k = len(fields)
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
            f"FIELDS {' '.join(fields)}\nSIZE {' '.join(['4'] * k)}\nTYPE {' '.join(['F'] * k)}\nCOUNT {' '.join(['1'] * k)}\n"
            f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n").encode("ascii")


def saveToFile(filename: str, cloud: PointCloud) -> bool:
    """output.cpp:39-47: '.pcd' is appended unless the name already contains it."""
    name = filename if ".pcd" in filename else filename + ".pcd"
    has_normals = cloud.HasNormals()
    has_colors = cloud.HasColors()
    step = 12 + (12 if has_normals else 0) + (4 if has_colors else 0)
    rows = cloud.be.download_f32(cloud.id, step, 0, 4, 8, 12 if has_normals else None, step - 4 if has_colors else None, 1)
    try:
        with open(name, "wb") as f:
            f.write(_pcd_header(len(rows), has_normals, has_colors))
            f.write(rows.tobytes())
    except OSError:
        return False
    return True


def readPcd(filename: str):
    """Reader for the files saveToFile writes (binary, 4-byte fields): returns (points (n,3) f32, normals (n,3) f32 or None,
    rgb bytes (n,3) uint8 [r, g, b] or None)."""
    with open(filename, "rb") as f:
        blob = f.read()
    head, _, body = blob.partition(b"DATA binary\n")
    meta = {ln.split()[0]: ln.split()[1:] for ln in head.decode("ascii").splitlines() if ln and not ln.startswith("#")}
    fields, n = meta["FIELDS"], int(meta["POINTS"][0])
    if set(meta["SIZE"]) != {"4"} or set(meta["TYPE"]) != {"F"}:
        raise ValueError("readPcd: only 4-byte F fields are supported")
    raw = np.frombuffer(body, dtype=np.uint8, count=n * 4 * len(fields)).reshape(n, 4 * len(fields))
    rows = raw.view(np.float32)

    def cols(names):
        return rows[:, [fields.index(c) for c in names]]

    pts = cols(("x", "y", "z"))
    nrm = cols(("normal_x", "normal_y", "normal_z")) if "normal_x" in fields else None
    rgb = None
    if "rgb" in fields:
        o = 4 * fields.index("rgb")
        rgb = raw[:, [o + 2, o + 1, o]]
    return pts, nrm, rgb


def assembleMapPointCloud(be, submaps) -> PointCloud:
    """Mapper::getAssembledMapPointCloud (Mapper.cpp:183-208): the map clouds of all submaps, concatenated in submap order, on the
    device.  The copies of the reference (getMapPointCloudCopy per submap, push_back per point) become one device copy + appends."""
    out = None
    for sm in submaps:
        cloud = sm.getMapPointCloud()
        if out is None:
            out = PointCloud(be, be.transform_cloud(cloud.id, np.eye(4)))
        else:
            be.cloud_append(out.id, cloud.id)
    return out if out is not None else PointCloud.from_numpy(be, np.zeros((0, 3)))


def voxelize(be, voxel_size: float, cloud: PointCloud) -> PointCloud:
    """o3d_slam::voxelize (helpers.cpp:107-113) as publishMaps uses it on the assembled map; voxel_size <= 0 leaves the cloud alone."""
    if voxel_size <= 0:
        return cloud
    return PointCloud(be, be.voxel_down_sample(cloud.id, voxel_size))
