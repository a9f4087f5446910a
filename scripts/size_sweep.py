"""How the fused ICP pass kernel scales with the size of the scan: kernel time per pass, algorithmic GB/s and fraction of the
HBM peak for sources of 64k ... 4M points against 1M / 4M-point maps (same scene, same pose offset, 10 fixed iterations).
The configs[1] headline (64k vs 1M) is a 16-us kernel; this shows what the same kernel sustains once a launch carries
enough queries to fill the chip.  Usage (GPU box): python scripts/size_sweep.py > gpurun_out/size_sweep.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from bench import ALGO_BYTES_PER_POINT, HBM_PEAK_GBS, ICP_ITERS, MAX_CORR

scene = syn.make_scene()
T = syn.ground_truth_pose()
Ti = np.linalg.inv(T)
be = backend.Backend(0, backend.PRECISION_F32)
print("%10s %10s %12s %12s %10s %8s %10s" % ("n_src", "n_map", "ms/registr.", "us/pass", "GB/s", "frac", "fitness"))
maps = {}
for n_map in (1_000_000, 4_000_000):
    if os.environ.get("O3DS_SWEEP_ONLY") and n_map != 1_000_000:
        continue
    tgt, nrm = syn.sample_map(scene, n_map, seed=syn.SEED_MAP)
    t_id = be.upload(tgt, nrm)
    be.build_index(t_id, MAX_CORR, 0.0)
    maps[n_map] = t_id
CASES = ((65536, 1_000_000), (262144, 1_000_000), (1_048_576, 1_000_000), (4_194_304, 1_000_000), (4_194_304, 4_000_000))
if os.environ.get("O3DS_SWEEP_ONLY"):  # one case only (for counter runs)
    CASES = (CASES[int(os.environ["O3DS_SWEEP_ONLY"])],)
for n_src, n_map in CASES:
    if n_src == 65536:
        src = syn.vlp16_scan(scene, T)
    else:  # a scan-like subset of the surfaces, seen from the sensor pose
        p, _ = syn.sample_map(scene, n_src, seed=4242)
        src = p @ Ti[:3, :3].T + Ti[:3, 3]
    s_id = be.upload(src)
    t_id = maps[n_map]
    step = lambda: be.icp_point_to_plane_dev(s_id, t_id, MAX_CORR, max_iter=ICP_ITERS, rel_fitness=0.0, rel_rmse=0.0)
    for _ in range(3):
        res = step()
    be.synchronize()
    reps = 20 if n_src <= 1_048_576 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        res = step()
    be.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    be.profile_enable(True)
    for _ in range(reps):
        step()
    n_launch, kern_ms = be.profile_read()
    be.profile_enable(False)
    us = kern_ms * 1e3 / n_launch
    gbs = n_src * ALGO_BYTES_PER_POINT / (us * 1e-6) / 1e9
    print("%10d %10d %12.3f %12.1f %10.1f %8.4f %10.4f" % (n_src, n_map, ms, us, gbs, gbs / HBM_PEAK_GBS, res["fitness"]))
    be.free(s_id)
be.close()
