import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po
src, tgt, nrm, _ = syn.config2_inputs(n_map=100_000, n_az=512)
for n in (37, 64, 200):
    s = src[:n]
    for k in range(0, 4):
        ref = po.icp_point_to_plane(s, tgt, nrm, 1.0, max_iter=k, rel_fitness=0.0, rel_rmse=0.0)
        be = backend.Backend(0, backend.PRECISION_F64)
        got = be.icp_point_to_plane(s, tgt, nrm, 1.0, max_iter=k, rel_fitness=0.0, rel_rmse=0.0)
        be.close()
        print(os.environ.get("O3DS_ICP_MODE"), n, k, "ref fit %.4f ncorr %d T03 %.5f | got fit %.4f ncorr %d T03 %.5f it %d" % (
            ref["fitness"], ref["n_corr"], ref["transformation"][0, 3], got["fitness"], got["n_corr"], got["transformation"][0, 3], got["iterations"]))
