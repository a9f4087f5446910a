"""Where the normal-estimation kernel spends its time: per-wavefront wall-clock (100 MHz) duration, start time, rounds, farthest ring and
chunks from a -DO3DS_NRM_CHECK build (O3DS_BACKEND_LIB=.../libo3ds_check.so), on the voxel-filtered OS-128-like scan (r 3 m, knn 20)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
path = os.path.join(tempfile.gettempdir(), "nrm_stats.bin")
os.environ["O3DS_NRM_STATS_FILE"] = path
from open3d_slam_amd import backend, synthetic as syn
radius, knn = (float(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3.0, 20)
scene = syn.make_scene()
scan = syn.os128_scan(scene, syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0)))
be = backend.Backend(0)
c = be.upload(scan)
v = be.voxel_down_sample(c, 0.1)
for rep in range(3):
    be.estimate_normals(v, radius, knn)
be.synchronize()
raw = np.fromfile(path, dtype=np.uint64)
words = 16 if os.environ.get("O3DS_NRM_PHASES") else 6  # a -DO3DS_NRM_PHASES build also stores shader-clock cycles per phase
w = raw.reshape(-1, words).astype(np.float64)
w = w[w[:, 0] > 0]
clk, start, rounds, ring, chunks = w[:, 0] / 100.0, (w[:, 1] - w[:, 1].min()) / 100.0, w[:, 2], w[:, 3], w[:, 4]  # us
pc = lambda a: "mean %8.1f p50 %8.1f p90 %8.1f p99 %8.1f max %8.1f" % (a.mean(), *np.percentile(a, [50, 90, 99]), a.max())
print(f"r {radius} knn {knn}: {len(w)} wavefronts (16 points each)")
print("duration us  ", pc(clk))
print("rounds       ", pc(rounds))
print("chunks       ", pc(chunks))
print("farthest ring", pc(ring))
end = start + clk
print("kernel span us %.1f; sum of wavefront durations %.0f us = %.2f wavefronts resident on average (1024 SIMDs)" % (end.max(), clk.sum(), clk.sum() / end.max()))
print("start times us", pc(start))
for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 100)):
    a, b = np.percentile(end, [lo, hi]) if False else (end.max() * lo / 100, end.max() * hi / 100)
    live = ((start < b) & (end > a)).sum()
    print(f"  wavefronts alive during {lo:3d}-{hi:3d} % of the span: {live}")
print("corr(duration, rounds) %.2f  corr(duration, chunks) %.2f" % (np.corrcoef(clk, rounds)[0, 1], np.corrcoef(clk, chunks)[0, 1]))
order = np.argsort(start)
for d in range(10):
    sel = order[len(order) * d // 10:len(order) * (d + 1) // 10]
    print("  start decile %d: start %6.1f .. %6.1f us, duration mean %5.1f p90 %5.1f max %5.1f, chunks mean %.2f rounds mean %.2f" % (
        d, start[sel].min(), start[sel].max(), clk[sel].mean(), np.percentile(clk[sel], 90), clk[sel].max(), chunks[sel].mean(), rounds[sel].mean()))
slow = np.argsort(-clk)[:8]
for i in slow:
    print("  slow wavefront %5d: %7.1f us rounds %3d ring %d chunks %3d start %7.1f" % (i, clk[i], rounds[i], ring[i], chunks[i], start[i]))

if words == 16:
    names = ["set-up", "find work (row arithmetic)", "row bounds + segment table", "flat number -> segment", "candidate loads + distances + first test",
             "regula falsi on the count", "compaction", "ranking + new bound", "cumulants + output", "-"]
    ph = w[:, 6:16]
    tot = ph.sum()
    print("shader-clock cycles per wavefront by phase (mean; share of the total):")
    for k, nm in enumerate(names[:9]):
        print("  %-44s %9.0f  %5.1f %%" % (nm, ph[:, k].mean(), 100.0 * ph[:, k].sum() / tot))
    print("  %-44s %9.0f" % ("all phases", ph.sum(1).mean()))
