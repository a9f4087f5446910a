#!/usr/bin/env python
"""A/B of the two f32 neighbourhood kernels behind o3ds_estimate_normals: the ranking kernel (O3DS_NRM_SELECT=0, verified bit for bit
against the oracle in f64 storage) and the selection kernel (default).  Same neighbour SETS are required: the normals may differ only by
the rounding of the cumulant sums (sequential in rank order vs exact order-independent).  Prints per case: points, exact-equal normals,
worst angle, points beyond 1e-6 rad with a defined direction; the kernels' time per call from the library's event spans (tag 8)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_amd import backend, synthetic as syn  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def cases():
    scene = syn.make_scene()
    raw = syn.os128_scan(scene, np.eye(4))
    vox = po.voxel_down_sample(raw[(np.linalg.norm(raw, axis=1) >= 2.0) & (np.linalg.norm(raw, axis=1) <= 30.0)], 0.1)
    rng = np.random.default_rng(3)
    yield "os128 voxel 0.1 (stream scan)", vox, [(3.0, 20), (1.0, 5), (0.5, 30), (2.0, 48)]
    yield "vlp16 raw", syn.vlp16_scan(scene, syn.ground_truth_pose()), [(3.0, 20), (0.3, 20)]
    yield "uniform cube, dense", rng.uniform(-1, 1, size=(40_000, 3)), [(0.5, 20), (0.2, 10), (1.0, 48)]
    dup = np.repeat(rng.uniform(-1, 1, size=(500, 3)), 40, axis=0)  # 40 copies of every point: every distance tie there is
    yield "duplicates x40", dup[rng.permutation(len(dup))], [(0.5, 20), (3.0, 30)]
    lattice = np.stack(np.meshgrid(np.arange(40), np.arange(40), np.arange(12), indexing="ij"), -1).reshape(-1, 3) * 0.125
    yield "lattice (equal distances everywhere)", lattice.astype(np.float64), [(1.0, 20), (0.3, 20), (3.0, 48)]
    yield "sparse (rings to the radius)", rng.uniform(-30, 30, size=(3000, 3)), [(3.0, 20), (10.0, 20)]
    yield "tiny", rng.uniform(-1, 1, size=(7, 3)), [(3.0, 20)]
    yield "one point", np.array([[1.0, 2.0, 3.0]]), [(3.0, 20)]


def run(be, pts, radius, knn, select):
    os.environ["O3DS_NRM_SELECT"] = "1" if select else "0"
    c = be.upload(pts)
    be.estimate_normals(c, radius, knn)  # warm (cell size heuristic remembered)
    be.profile_enable(True)
    reps = 5
    for _ in range(reps):
        be.estimate_normals(c, radius, knn)
    be.synchronize()
    cnt, ms = be.span_read(8)
    be.profile_enable(False)
    _, nrm = be.download(c)
    stored, _ = be.download(c)
    be.free(c)
    return nrm, stored, 1e3 * ms / max(cnt, 1)


def main():
    be = backend.Backend(0, backend.PRECISION_F32)
    bad_total = 0
    for name, pts, settings in cases():
        for radius, knn in settings:
            old, stored, t_old = run(be, pts, radius, knn, False)
            new, _, t_new = run(be, pts, radius, knn, True)
            again, _, _ = run(be, pts, radius, knn, True)
            same = np.all(old == new, axis=1)
            sin = np.linalg.norm(np.cross(old, new), axis=1)
            flip = np.einsum("ij,ij->i", old, new) < 0
            # where the direction is defined by the data (oracle on the stored values): the two must agree to rounding
            beyond = np.flatnonzero((sin > 1e-6) | flip)
            gap_ok = 0
            if len(beyond):
                ref = po.estimate_normals(stored, radius, knn)
                d_old = np.linalg.norm(np.cross(old[beyond], ref[beyond]), axis=1)
                d_new = np.linalg.norm(np.cross(new[beyond], ref[beyond]), axis=1)
                gap_ok = int(np.sum((d_old > 1e-4) | (d_new > 1e-4)))  # ill-defined directions: both or either far from the oracle too
            repeat = bool(np.array_equal(new, again))
            bad = len(beyond) - gap_ok
            bad_total += bad + (0 if repeat else 1)
            print(f"{name:38s} r {radius:5.2f} knn {knn:2d} n {len(pts):6d}: equal {same.mean():7.4f}  worst sin {sin.max():.2e}  flips {int(flip.sum()):3d}  "
                  f"beyond 1e-6: {len(beyond):4d} (ill-defined {gap_ok:4d}, UNEXPLAINED {bad:3d})  repeat {repeat}  kernels {t_old:7.1f} -> {t_new:7.1f} us", flush=True)
    be.close()
    print("normals_select_check:", "OK" if bad_total == 0 else f"FAILED ({bad_total})")
    return 0 if bad_total == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
