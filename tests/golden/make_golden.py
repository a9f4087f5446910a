"""Generates tests/golden/*.npz.

The reference ships no golden vectors for this path (parity unpinned) and Open3D cannot
be imported here, so these fixtures pin the INDEPENDENT numpy/scipy restatement (oracle/np_oracle.py)
on small seeded inputs.  The C oracle and the HIP backend are both tested against them.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from open3d_slam_amd import synthetic as syn  # noqa: E402
from oracle import np_oracle as no  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # G1: scan-to-map ICP, 2048-pt scan vs 50k-pt map, 10 fixed iterations
    src, tgt, nrm, T_gt = syn.config2_inputs(n_map=50_000, n_az=128)
    r = no.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    rc = no.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)  # default convergence criteria
    np.savez(os.path.join(HERE, "icp_scan_to_map.npz"), n_map=50_000, n_az=128, max_corr=1.0,
             T10=r["transformation"], fitness10=r["fitness"], rmse10=r["inlier_rmse"],
             Tconv=rc["transformation"], fitness_conv=rc["fitness"], rmse_conv=rc["inlier_rmse"], iters_conv=rc["iterations"])
    # G2: scan-to-scan (config 1 shape, reduced): voxel 0.1 -> normals(knn 20, r 3.0) -> ICP 10 iters
    a, b = syn.config1_inputs(n_az=256)
    av, _ = no.voxel_down_sample(a, 0.1)
    bv, _ = no.voxel_down_sample(b, 0.1)
    bn = no.estimate_normals(bv, 3.0, 20)
    r2 = no.icp_point_to_plane(av, bv, bn, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    order = np.lexsort((bv[:, 2], bv[:, 1], bv[:, 0]))
    np.savez(os.path.join(HERE, "scan_pair.npz"), n_az=256, voxel=0.1, knn=20, radius=3.0, n_a=len(av), n_b=len(bv),
             b_sorted_head=bv[order][:64], bn_sorted_head=bn[order][:64], T10=r2["transformation"], fitness10=r2["fitness"],
             rmse10=r2["inlier_rmse"])
    print("wrote golden fixtures:", os.listdir(HERE))


if __name__ == "__main__":
    main()
