"""Diagnostic for test_large_coordinates_and_exact_sums: pose error vs the oracle under the current split, without the x64 rank
headroom, and with the split bypassed (plain f64 sums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po
src, tgt, nrm, _ = syn.config2_inputs(n_map=100_000, n_az=512)
off = np.array([1.0e5, -2.0e5, 50.0]); T0 = np.eye(4); T0[:3, 3] = off
ref = po.icp_point_to_plane(src, tgt + off, nrm, 1.0, init=T0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
be = backend.Backend(0, backend.PRECISION_F64)
for rep in range(3):
    got = be.icp_point_to_plane(src, tgt + off, nrm, 1.0, init=T0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
    print(os.environ.get("VARIANT"), rep, "dt %.3e dr %.3e" % syn.se3_error(got["transformation"], ref["transformation"]))
be.close()
