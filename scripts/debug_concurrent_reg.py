"""Is one registration's result the same bits whatever else the GPU is doing?  The same (scan, map, guess) registered N times on one handle,
alone and while a second handle on another thread runs registrations of its own; then a scan that is the head of a lazy chain
(crop + voxel, count not yet on the host) registered at once and after a synchronize."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open3d_slam_amd import backend, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
scene = syn.make_scene()
src = syn.vlp16_scan(scene, syn.ground_truth_pose())
tgt, nrm = syn.sample_map(scene, bench.N_MAP, seed=syn.SEED_MAP)
be = backend.Backend(0)
s_id, t_id = be.upload(src), be.upload(tgt, nrm)
be.build_index(t_id, bench.MAX_CORR, 0.0)
kw = dict(max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6)
def reg(b, s, t):
    return np.array(b.icp_point_to_plane_dev(s, t, bench.MAX_CORR, **kw)["transformation"])
ref = reg(be, s_id, t_id)
alone = [reg(be, s_id, t_id) for _ in range(n)]
print("alone: differing results", sum(not np.array_equal(ref, r) for r in alone), "of", n, flush=True)
stop = False
def other():
    b2 = backend.Backend(0)
    s2, t2 = b2.upload(src[:30000]), b2.upload(tgt[:300000], nrm[:300000])
    b2.build_index(t2, bench.MAX_CORR, 0.0)
    while not stop:
        reg(b2, s2, t2)
    b2.close()
th = threading.Thread(target=other); th.start()
import time; time.sleep(0.5)
busy = [reg(be, s_id, t_id) for _ in range(n)]
stop = True; th.join()
bad = [r for r in busy if not np.array_equal(ref, r)]
print("beside another handle: differing results", len(bad), "of", n, "| max abs difference", max([float(np.abs(r - ref).max()) for r in bad], default=0.0), flush=True)
if bad:
    print(bad[0] - ref)
be.close()
