"""Per-ABI-call time in the config-2 stream: every Backend method is wrapped with a device sync + timer (so the numbers are
kernel + host time of each call in isolation; the un-instrumented loop overlaps some of it)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, parameters as P, synthetic as syn
from open3d_slam_amd.mapper import Mapper
from open3d_slam_amd.odometry import LidarOdometry
from open3d_slam_amd.pointcloud import PointCloud

_sync = backend.Backend.synchronize
acc = collections.defaultdict(lambda: [0, 0.0])
hist = collections.defaultdict(list)  # per call site: every duration (--outliers prints the frames far above the median)
seq = []  # every call in order (the last frame is printed with --sequence)
phase = ["?"]
def wrap(cls, name):
    fn = getattr(cls, name)
    def w(self, *a, **k):
        t0 = time.perf_counter(); r = fn(self, *a, **k); _sync(self); dt = time.perf_counter() - t0
        e = acc[(phase[0], name)]; e[0] += 1; e[1] += dt
        seq.append((phase[0], name, dt))
        hist[(phase[0], name)].append(dt)
        return r
    setattr(cls, name, w)
for n in dir(backend.Backend):
    if not n.startswith("_") and callable(getattr(backend.Backend, n)) and n not in ("close", "synchronize"):
        wrap(backend.Backend, n)
mp = P.lua_default_mapper_parameters()
op = P.OdometryParameters()
op.scanMatcher_.icp_ = P.IcpParameters(maxNumIter_=50, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
op.scanProcessing_.voxelSize_ = 0.1
op.scanProcessing_.cropper_ = P.ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
scene = syn.make_scene(); poses = syn.figure_eight_poses(200, 0.1)
F = 20
scans = [syn.os128_scan(scene, poses[k], frame=k, n_az=1024) for k in range(F)]
be = backend.Backend(0); odo = LidarOdometry(be); odo.setParameters(op); mapper = Mapper(be, odo); mapper.setParameters(mp)
for k, raw in enumerate(scans):
    if k == 1: acc.clear()
    if k == F - 1: seq.clear()
    phase[0] = "upload"; cloud = PointCloud.from_numpy(be, raw)
    phase[0] = "odom"; odo.addRangeScan(cloud, 0.1 * k)
    phase[0] = "map"; mapper.addRangeMeasurement(cloud, 0.1 * k)
    cloud.release()
tot = sum(v[1] for v in acc.values())
print("per frame: %.3f ms in ABI calls" % (tot / (F - 1) * 1e3))
for (ph, n), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%-7s %-28s calls/frame %5.2f  ms/frame %7.3f  us/call %8.1f" % (ph, n, c / (F - 1), t / (F - 1) * 1e3, t / c * 1e6))
if "--sequence" in sys.argv:
    print("calls of the last frame, in order:")
    for ph, n, dt in seq:
        print("  %-7s %-28s %8.1f us" % (ph, n, dt * 1e6))
if "--outliers" in sys.argv:
    for key, v in sorted(hist.items()):
        a = np.array(v) * 1e6
        med = np.median(a)
        big = [(i, x) for i, x in enumerate(a) if x > 2.0 * med + 50.0]
        if big:
            print("%-7s %-26s median %7.1f us; outliers (call #, us): %s" % (key[0], key[1], med, ", ".join("%d: %.0f" % b for b in big[:12])))
