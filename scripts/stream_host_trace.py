"""Host-side picture of the configs[2] stream: every Backend method timed on the host WITHOUT a device synchronisation (when the call
was made, relative to the frame's start, and how long the host spent in it), frames as bench.py runs them (float32 PointCloud2 ingest,
odometry, mapping).  --pinned: the raw scans live in page-locked buffers (Backend.pinned_records); --frames N; --stage-sync as bench.py's
staged run.  Prints the per-call means over the steady frames and the call sequence of the last frame."""
import argparse, collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend
from open3d_slam_amd.mapper import Mapper
from open3d_slam_amd.odometry import LidarOdometry
from open3d_slam_amd.pointcloud import PointCloud

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--pinned", action="store_true")
ap.add_argument("--stage-sync", action="store_true")
ap.add_argument("--shipped", action="store_true", help="GeneralizedIcp + downsampling_ratio 0.3, as the shipped Lua")
args = ap.parse_args()
scans = bench.make_stream(args.frames)
acc = collections.defaultdict(lambda: [0, 0.0])
seq = []
t_frame = [0.0]
def wrap(cls, name):
    fn = getattr(cls, name)
    def w(self, *a, **k):
        t0 = time.perf_counter(); r = fn(self, *a, **k); t1 = time.perf_counter()
        e = acc[name]; e[0] += 1; e[1] += t1 - t0
        seq.append((name, (t0 - t_frame[0]) * 1e6, (t1 - t0) * 1e6))
        return r
    setattr(cls, name, w)
for n in dir(backend.Backend):
    if not n.startswith("_") and callable(getattr(backend.Backend, n)) and n not in ("close", "pinned_records", "free_pinned"):
        wrap(backend.Backend, n)
be = backend.Backend(0)
if args.pinned:
    pin = []
    for s in scans:
        rec = np.zeros((len(s), 4), dtype=np.float32); rec[:, :3] = s
        p = be.pinned_records(len(s), 16); p[:] = rec.view(np.uint8).reshape(len(s), 16); pin.append(p.view(np.float32).reshape(len(s), 4))
    scans = pin
mp, op = bench.stream_parameters(shipped=args.shipped)
odo = LidarOdometry(be); odo.setParameters(op); mapper = Mapper(be, odo); mapper.setParameters(mp)
if args.shipped:
    odo.setDownSampleSeed(71); mapper.scan2MapReg_.setDownSampleSeed(72)
frame_t = []
for k, raw in enumerate(scans):
    if k == 5: acc.clear()
    if k == len(scans) - 1: seq.clear()
    t_frame[0] = t0 = time.perf_counter()
    cloud = PointCloud.from_pointcloud2(be, raw)
    odo.addRangeScan(cloud, 0.1 * k)
    if args.stage_sync: be.synchronize()
    mapper.addRangeMeasurement(cloud, 0.1 * k)
    if args.stage_sync: be.synchronize()
    cloud.release()
    frame_t.append(time.perf_counter() - t0)
be.synchronize()
n = len(scans) - 5
ft = np.array(frame_t[5:]) * 1e6
print("frames %d  mean %.1f us  median %.1f us  -> %.1f scans/s; host time inside ABI calls %.1f us per frame" %
      (n, ft.mean(), np.median(ft), 1e6 / ft.mean(), sum(v[1] for v in acc.values()) / n * 1e6))
for name, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-28s calls/frame %5.2f  host us/frame %7.1f  us/call %7.1f" % (name, c / n, t / n * 1e6, t / c * 1e6))
print("last frame (call, start us, host us):")
for name, st, dt in seq:
    print("  %-28s %8.1f %8.1f" % (name, st, dt))
