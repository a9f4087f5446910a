"""Is the config-2 stream reproducible to the last bit?  Runs it several times in one process (fresh handles), hashes what every ABI
call produced (clouds that come back, clouds changed in place, registration results) and reports the first call whose hash differs
between the runs (round 1: the first estimate_normals; since round 2 none -- tests/test_repro_gpu.py asserts it).
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


def check(frames=12, runs=3, dirty_pool=False, n_az=1024, verbose=True):
    """runs the stream `runs` times on fresh handles; returns [(run, first differing call or None, frame, pose bitwise equal)]"""
    mp = P.lua_default_mapper_parameters()
    op = P.OdometryParameters()
    op.scanMatcher_.icp_ = P.IcpParameters(maxNumIter_=50, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
    op.scanProcessing_.voxelSize_ = 0.1
    op.scanProcessing_.cropper_ = P.ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)
    scans = [syn.os128_scan(scene, poses[k], frame=k, n_az=n_az).astype(np.float32) for k in range(frames)]
    ref, marks, T0 = run(frames, scans, mp, op)
    if verbose:
        print("calls per run:", len(ref))
    out = []
    for r in range(1, runs):
        if dirty_pool:  # fill and free large device buffers on another handle and on this process's pools: the addresses, and with
            be0 = backend.Backend(0)  # them the arrival order of every atomic scatter, change
            ids = [be0.upload(np.full((500_000 + 333_333 * i, 3), np.nan), np.full((500_000 + 333_333 * i, 3), np.nan)) for i in range(3)]
            for i in ids:
                be0.free(i)
            be0.close()
        cur, _, T = run(frames, scans, mp, op)
        first = next((i for i, (a, b) in enumerate(zip(ref, cur)) if a != b), None)
        if first is None and len(cur) != len(ref):
            first = min(len(cur), len(ref))
        same_pose = bool(np.array_equal(T, T0))
        frame = None if first is None else max(k for k, m in enumerate(marks) if m <= first)
        out.append((r, first, frame, same_pose))
        if verbose:
            if first is None:
                print("run %d: identical to run 0 (%d calls), final pose bitwise equal: %s" % (r, len(cur), same_pose))
            else:
                print("run %d: first difference at call %s (frame %d): %s vs %s; previous call: %s" %
                      (r, first, frame, ref[first] if first < len(ref) else None, cur[first] if first < len(cur) else None, ref[first - 1] if first else None))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--dirty-pool", action="store_true", help="disturb the device allocations between the runs")
    args = ap.parse_args()
    check(args.frames, args.runs, args.dirty_pool)
