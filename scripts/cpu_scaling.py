import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import synthetic as syn
from oracle import pyoracle as po
src, tgt, nrm, _ = syn.config2_inputs()
tree = po.KDTree(tgt)
print("nproc", os.cpu_count(), "omp default", po.lib().orc_num_threads())
for th in (1, 8, 16, 32, 64, 128, 256):
    po.lib().orc_set_num_threads(th)
    po.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=2, rel_fitness=0, rel_rmse=0, tree=tree)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps):
        po.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0, rel_rmse=0, tree=tree)
    dt = (time.perf_counter() - t0) / reps
    print(f"threads {th:4d}: {10/dt:8.1f} icp it/s   ({dt*1e3:.1f} ms / 10 iters)")
scan = syn.os128_scan(syn.make_scene(), np.eye(4))
v = po.voxel_down_sample(scan, 0.1)
for th in (16, 64, 128, 256):
    po.lib().orc_set_num_threads(th)
    t0 = time.perf_counter(); po.estimate_normals(v, 3.0, 20); print(f"normals {len(v)} pts, threads {th}: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); po.voxel_down_sample(scan, 0.1); print("voxel ds ms", (time.perf_counter()-t0)*1e3)
