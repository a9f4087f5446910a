"""Where the normal-estimation kernel spends its time: per-point work counters and wavefront clocks from a -DO3DS_NRM_STATS build
(scripts/gpu_normals_stats.sh builds it as lib/libo3ds_backend_stats.so).  Points are processed in cell order, 64 per wavefront."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
backend.LIB_PATH = os.path.join(os.path.dirname(backend.LIB_PATH), "libo3ds_backend_stats.so")
path = "/tmp/nrm_stats.bin"
os.environ["O3DS_NRM_STATS_FILE"] = path
scene = syn.make_scene()
scan = syn.os128_scan(scene, syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0)))
be = backend.Backend(0, backend.PRECISION_F32)
c = be.upload(scan)
v = be.voxel_down_sample(c, 0.1)
be.estimate_normals(v, 3.0, 20)
be.estimate_normals(v, 3.0, 20)
be.synchronize()
raw = open(path, "rb").read()
cell, n = np.frombuffer(raw[:16], dtype=np.float64)
st = np.frombuffer(raw[16:], dtype=np.uint32).reshape(-1, 8)
n = len(st)
print(f"n {n} cell {cell:.4f}")
names = ["last ring", "rows", "candidates", "accepts", "clk search", "clk cov+eig", "cnt"]
for k, nm in enumerate(names):
    col = st[:, k].astype(np.float64)
    print(f"{nm:12s} mean {col.mean():10.1f}  p50 {np.percentile(col,50):9.0f}  p90 {np.percentile(col,90):9.0f}  p99 {np.percentile(col,99):9.0f}  max {col.max():9.0f}")
w = n // 64
per_wave = st[: w * 64].reshape(w, 64, 8)
clk = per_wave[:, :, 4].max(axis=1).astype(np.float64)  # lanes of a wavefront finish the search together
print(f"waves {w}: search clocks mean {clk.mean():.0f} p50 {np.percentile(clk,50):.0f} p90 {np.percentile(clk,90):.0f} p99 {np.percentile(clk,99):.0f} max {clk.max():.0f}")
for k, nm in ((0, "max ring"), (1, "max rows"), (2, "max cand"), (3, "max acc")):
    mx = per_wave[:, :, k].max(axis=1).astype(np.float64)
    sm = per_wave[:, :, k].mean(axis=1)
    print(f"  per wave {nm}: mean {mx.mean():.1f} p90 {np.percentile(mx,90):.0f} max {mx.max():.0f}   (lane mean {sm.mean():.1f})   corr with clocks {np.corrcoef(mx, clk)[0,1]:.3f}")
t0 = per_wave[:, 0, 7].astype(np.int64)
start = (t0 - t0.min()) & 0xffffffff
end = start + clk
print(f"wave start spread: p50 {np.percentile(start,50):.0f} p90 {np.percentile(start,90):.0f} max {start.max():.0f}; last end {end.max():.0f} clocks")
order = np.argsort(-clk)[:8]
for i in order:
    print(f"  slow wave {i}: clk {clk[i]:.0f} ring max {per_wave[i,:,0].max()} rows max {per_wave[i,:,1].max()} cand max {per_wave[i,:,2].max()} acc max {per_wave[i,:,3].max()} acc sum {per_wave[i,:,3].sum()} cand sum {per_wave[i,:,2].sum()}")
be.close()
