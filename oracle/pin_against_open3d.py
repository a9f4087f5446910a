#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- pins the oracle against the REAL Open3D v0.15.1 (open3d_catkin/CMakeLists.txt:116-118 pins that tag).

    pip install open3d==0.15.1        # on any machine that can; this image has no network and no wheel
    python oracle/pin_against_open3d.py [--write]

Every array of tests/golden/*.npz is regenerated from Open3D's own calls at the sites open3d_slam calls them from:
  registration_icp + TransformationEstimationPointToPlane / PointToPoint   CloudRegistration.cpp:44-48,69-73
  registration_generalized_icp (covariances from normals, epsilon 1e-3)    CloudRegistration.cpp:16-21
  estimate_normals(KDTreeSearchParamHybrid) + normalize_normals + orient_normals_towards_camera_location   CloudRegistration.cpp:49-56
  voxel_down_sample                                                        helpers.cpp:107-113
  get_information_matrix_from_point_clouds                                 constraint_builders.cpp:70-73
and compared with the two restatements this repository tests against (oracle/np_oracle.py, the C oracle through oracle/pyoracle.py).
Exit status 0 = every comparison inside its tolerance (poses 1e-9 m / rad, scalars 1e-9 relative, index sets exact, normals: direction
1e-9 where the neighbourhood's two smallest covariance eigenvalues differ by more than 1e-6 of the largest; the count of points
outside that is reported -- Open3D leaves equal distances to its tree and uses the platform's acos / cos, the oracle fixes both,
DESIGN.md 2).  --write stores the Open3D-generated arrays as tests/golden/open3d_0_15_1.npz, after which tests/test_oracle.py's
test_golden_* read THAT file and DESIGN.md's "parity unpinned" can be struck.

Without the wheel the script says so and exits with status 3; tests/test_oracle.py::test_oracle_pinned_against_open3d is skipped
for the same reason.  Nothing in the product imports this file."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def have_open3d():
    try:
        import open3d as o3d  # noqa: F401

        return True
    except Exception:
        return False


def _pcd(o3d, pts, nrm=None):
    c = o3d.geometry.PointCloud()
    c.points = o3d.utility.Vector3dVector(np.ascontiguousarray(pts, dtype=np.float64))
    if nrm is not None:
        c.normals = o3d.utility.Vector3dVector(np.ascontiguousarray(nrm, dtype=np.float64))
    return c


def _se3_error(A, B):
    D = np.linalg.inv(A) @ B
    w = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    return float(np.linalg.norm(D[:3, 3])), float(np.arcsin(min(1.0, np.linalg.norm(w))))


def open3d_vectors():
    """every golden array, from Open3D itself"""
    import open3d as o3d

    from open3d_slam_amd import synthetic as syn

    reg = o3d.pipelines.registration
    out = {"open3d_version": o3d.__version__}
    # G1: scan-to-map ICP, 2048-pt scan vs 50k-pt map (tests/golden/icp_scan_to_map.npz)
    src, tgt, nrm, _ = syn.config2_inputs(n_map=50_000, n_az=128)
    crit10 = reg.ICPConvergenceCriteria(relative_fitness=0.0, relative_rmse=0.0, max_iteration=10)
    r = reg.registration_icp(_pcd(o3d, src), _pcd(o3d, tgt, nrm), 1.0, np.eye(4), reg.TransformationEstimationPointToPlane(), crit10)
    rc = reg.registration_icp(_pcd(o3d, src), _pcd(o3d, tgt, nrm), 1.0, np.eye(4), reg.TransformationEstimationPointToPlane(),
                              reg.ICPConvergenceCriteria(max_iteration=30))
    out.update(g1_T10=np.asarray(r.transformation), g1_fitness10=r.fitness, g1_rmse10=r.inlier_rmse, g1_Tconv=np.asarray(rc.transformation),
               g1_fitness_conv=rc.fitness, g1_rmse_conv=rc.inlier_rmse)
    rp = reg.registration_icp(_pcd(o3d, src), _pcd(o3d, tgt), 1.0, np.eye(4), reg.TransformationEstimationPointToPoint(False), crit10)
    out.update(g1_p2p_T10=np.asarray(rp.transformation), g1_p2p_fitness10=rp.fitness, g1_p2p_rmse10=rp.inlier_rmse)
    info = reg.get_information_matrix_from_point_clouds(_pcd(o3d, src), _pcd(o3d, tgt), 0.3, np.asarray(rc.transformation))
    out.update(g1_info=np.asarray(info))
    # G2: scan pair (tests/golden/scan_pair.npz): voxel 0.1 -> normals (knn 20, r 3.0) -> ICP 10 iterations; GICP on the same pair
    a, b = syn.config1_inputs(n_az=256)
    av = np.asarray(_pcd(o3d, a).voxel_down_sample(0.1).points)
    bc = _pcd(o3d, b).voxel_down_sample(0.1)
    bv = np.asarray(bc.points)
    bc.estimate_normals(o3d.geometry.KDTreeSearchParamHybrid(radius=3.0, max_nn=20))
    bc.normalize_normals()
    bc.orient_normals_towards_camera_location(np.zeros(3))
    bn = np.asarray(bc.normals)
    r2 = reg.registration_icp(_pcd(o3d, av), bc, 1.0, np.eye(4), reg.TransformationEstimationPointToPlane(), crit10)
    ac = _pcd(o3d, av)
    ac.estimate_normals(o3d.geometry.KDTreeSearchParamHybrid(radius=3.0, max_nn=20))
    ac.normalize_normals()
    ac.orient_normals_towards_camera_location(np.zeros(3))
    rg = reg.registration_generalized_icp(ac, bc, 1.0, np.eye(4), reg.TransformationEstimationForGeneralizedICP(), crit10)
    out.update(g2_av=av, g2_bv=bv, g2_bn=bn, g2_an=np.asarray(ac.normals), g2_T10=np.asarray(r2.transformation), g2_fitness10=r2.fitness,
               g2_rmse10=r2.inlier_rmse, g2_gicp_T10=np.asarray(rg.transformation), g2_gicp_fitness10=rg.fitness, g2_gicp_rmse10=rg.inlier_rmse)
    return out


def compare(vec, verbose=True):
    """Open3D's vectors against the numpy restatement and the C oracle; returns the list of failures"""
    from open3d_slam_amd import synthetic as syn
    from oracle import np_oracle as no
    from oracle import pyoracle as po

    bad = []

    def pose(name, A, B, tol=1e-9):
        dt, dr = _se3_error(np.asarray(A), np.asarray(B))
        ok = dt <= tol and dr <= tol
        if verbose:
            print(f"  {name}: |dt| {dt:.2e} m, angle {dr:.2e} rad {'ok' if ok else 'FAIL'}")
        if not ok:
            bad.append(name)

    def scalar(name, a, b, tol=1e-9):
        ok = abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))
        if verbose:
            print(f"  {name}: {float(a):.12g} vs {float(b):.12g} {'ok' if ok else 'FAIL'}")
        if not ok:
            bad.append(name)

    def same_set(name, A, B, tol=1e-12):
        A, B = np.asarray(A), np.asarray(B)
        oa, ob = np.lexsort(A.T[::-1]), np.lexsort(B.T[::-1])
        ok = A.shape == B.shape and np.allclose(A[oa], B[ob], atol=tol, rtol=0)
        if verbose:
            print(f"  {name}: {A.shape} vs {B.shape} {'ok' if ok else 'FAIL'}")
        if not ok:
            bad.append(name)
        return oa, ob

    src, tgt, nrm, _ = syn.config2_inputs(n_map=50_000, n_az=128)
    for tag, o in (("numpy restatement", no), ("C oracle", po)):
        print(tag)
        r = o.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
        pose("G1 point-to-plane, 10 iterations", r["transformation"], vec["g1_T10"])
        scalar("G1 fitness", r["fitness"], vec["g1_fitness10"])
        scalar("G1 inlier_rmse", r["inlier_rmse"], vec["g1_rmse10"])
        rc = o.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
        pose("G1 default criteria", rc["transformation"], vec["g1_Tconv"])
        if hasattr(o, "icp_point_to_point"):
            rp = o.icp_point_to_point(src, tgt, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
            pose("G1 point-to-point, 10 iterations", rp["transformation"], vec["g1_p2p_T10"])
        if hasattr(o, "information_matrix"):
            info = o.information_matrix(src, tgt, 0.3, np.asarray(vec["g1_Tconv"]))
            scalar("G1 information matrix (Frobenius)", np.linalg.norm(np.asarray(info) - vec["g1_info"]) / np.linalg.norm(vec["g1_info"]), 0.0, 1e-9)
        a, b = syn.config1_inputs(n_az=256)
        av = o.voxel_down_sample(a, 0.1)
        av = av[0] if isinstance(av, tuple) else av
        bv = o.voxel_down_sample(b, 0.1)
        bv = bv[0] if isinstance(bv, tuple) else bv
        same_set("G2 voxel_down_sample(a)", av, vec["g2_av"])
        ob, og = same_set("G2 voxel_down_sample(b)", bv, vec["g2_bv"])
        bn = o.estimate_normals(bv, 3.0, 20)
        dots = np.einsum("ij,ij->i", bn[ob], np.asarray(vec["g2_bn"])[og])
        n_off = int(np.sum(np.abs(dots) < 1 - 1e-9))
        n_flip = int(np.sum(dots < 0))
        print(f"  G2 normals: {n_off} of {len(dots)} directions differ by more than 1e-9, {n_flip} orientations differ "
              "(expected: only neighbourhoods the data do not define)")
        if n_off > 0.01 * len(dots):
            bad.append("G2 normals")
        r2 = o.icp_point_to_plane(np.asarray(vec["g2_av"]), np.asarray(vec["g2_bv"]), np.asarray(vec["g2_bn"]), 1.0, max_iter=10, rel_fitness=0.0,
                                  rel_rmse=0.0)  # on Open3D's own clouds and normals: isolates the registration
        pose("G2 point-to-plane on Open3D's normals", r2["transformation"], vec["g2_T10"])
        if hasattr(o, "icp_generalized"):
            rg = o.icp_generalized(np.asarray(vec["g2_av"]), np.asarray(vec["g2_an"]), np.asarray(vec["g2_bv"]), np.asarray(vec["g2_bn"]), 1.0,
                                   max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
            pose("G2 generalized ICP on Open3D's normals", rg["transformation"], vec["g2_gicp_T10"], 1e-8)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="store the Open3D-generated arrays as tests/golden/open3d_0_15_1.npz")
    args = ap.parse_args()
    if not have_open3d():
        print("open3d is not importable here: nothing was pinned (install open3d==0.15.1 and run this again)")
        return 3
    import open3d as o3d

    if not o3d.__version__.startswith("0.15"):
        print(f"warning: open3d {o3d.__version__} is not the pinned v0.15.1")
    vec = open3d_vectors()
    bad = compare(vec)
    if args.write:
        np.savez(os.path.join(GOLD, "open3d_0_15_1.npz"), **vec)
        print("wrote tests/golden/open3d_0_15_1.npz")
    print("PINNED: every comparison inside its tolerance" if not bad else f"NOT pinned: {bad}")
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main())
