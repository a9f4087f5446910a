"""Is the config-2 stream reproducible to the last bit?  Runs it twice in one process (two fresh handles), hashes what every ABI
call produced (clouds that come back, clouds changed in place, registration results) and reports the first call whose hash differs
between the runs -- the tool for the open item of DESIGN.md section 6 (the final pose varies in its 9th digit from run to run).
    python scripts/repro_stream.py [--frames 12]"""
import argparse, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, parameters as P, synthetic as syn
from open3d_slam_amd.mapper import Mapper
from open3d_slam_amd.odometry import LidarOdometry
from open3d_slam_amd.pointcloud import PointCloud

RETURNS_CLOUD = {"upload", "upload_f32", "crop_cloud", "voxel_down_sample", "crop_voxel_down_sample", "select_by_index", "transform_cloud"}
CHANGES_FIRST_ARG = {"estimate_normals", "map_insert_scan", "cloud_append", "voxelize_within_volume", "map_carve"}
ICP = {"icp_point_to_plane_dev", "icp_generalized_dev", "icp_point_to_point_dev", "icp_point_to_plane"}
raw_download = backend.Backend.download
log = None


def digest(be, cid):
    p, n = raw_download(be, cid)
    h = hashlib.sha1(np.ascontiguousarray(p).tobytes())
    if n is not None:
        h.update(np.ascontiguousarray(n).tobytes())
    return "%d:%s" % (len(p), h.hexdigest()[:10])


def wrap(name):
    fn = getattr(backend.Backend, name)

    def w(self, *a, **k):
        r = fn(self, *a, **k)
        if log is not None:
            if name in RETURNS_CLOUD:
                log.append((name, digest(self, r)))
            elif name in CHANGES_FIRST_ARG:
                log.append((name, digest(self, a[0])))
            elif name in ICP:
                log.append((name, hashlib.sha1(np.ascontiguousarray(r["transformation"]).tobytes()).hexdigest()[:10] + " it=%d" % r["iterations"]))
        return r

    setattr(backend.Backend, name, w)


for nm in RETURNS_CLOUD | CHANGES_FIRST_ARG | ICP:
    if hasattr(backend.Backend, nm):
        wrap(nm)


def run(frames, scans, mp, op):
    global log
    log = []
    be = backend.Backend(0)
    odo = LidarOdometry(be)
    odo.setParameters(op)
    mapper = Mapper(be, odo)
    mapper.setParameters(mp)
    marks = []
    for k in range(frames):
        marks.append(len(log))
        cloud = PointCloud.from_pointcloud2(be, scans[k])
        odo.addRangeScan(cloud, 0.1 * k)
        mapper.addRangeMeasurement(cloud, 0.1 * k)
        cloud.release()
    out, T = log, mapper.getMapToRangeSensor().copy()
    log = None
    be.close()
    return out, marks, T


ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--dirty-pool", action="store_true", help="fill and free a large device buffer before the first run: if the first run then "
                "agrees with the later ones, something reads memory it has not written")
args = ap.parse_args()
mp = P.lua_default_mapper_parameters()
op = P.OdometryParameters()
op.scanMatcher_.icp_ = P.IcpParameters(maxNumIter_=50, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
op.scanProcessing_.voxelSize_ = 0.1
op.scanProcessing_.cropper_ = P.ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
scene = syn.make_scene()
poses = syn.figure_eight_poses(200, 0.1)
scans = [syn.os128_scan(scene, poses[k], frame=k).astype(np.float32) for k in range(args.frames)]
if args.dirty_pool:
    be0 = backend.Backend(0)
    ids = [be0.upload(np.full((2_000_000, 3), np.nan), np.full((2_000_000, 3), np.nan)) for _ in range(3)]
    for i in ids:
        be0.free(i)
    be0.close()
ref, marks, T0 = run(args.frames, scans, mp, op)
print("calls per run:", len(ref))
for r in range(1, args.runs):
    cur, _, T = run(args.frames, scans, mp, op)
    first = next((i for i, (a, b) in enumerate(zip(ref, cur)) if a != b), None)
    if first is None and len(cur) == len(ref):
        print("run %d: identical to run 0 (%d calls), final pose bitwise equal: %s" % (r, len(cur), np.array_equal(T, T0)))
    else:
        frame = max(k for k, m in enumerate(marks) if m <= (first if first is not None else 0))
        print("run %d: first difference at call %s (frame %d): %s vs %s; previous call: %s" %
              (r, first, frame, ref[first] if first is not None else None, cur[first] if first is not None else None, ref[first - 1] if first else None))
