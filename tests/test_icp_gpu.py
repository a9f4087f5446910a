"""HIP point-to-plane ICP vs the CPU oracle, through the C-ABI (needs an MI355X)."""
import os

import numpy as np
import pytest

from open3d_slam_amd import backend, synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

# SURVEY 8c stated tolerances (same fixed number of iterations, same init, same clouds)
TOL_T, TOL_R = 1e-3, 1e-3          # f32 point storage, f64 accumulation
TOL_T64, TOL_R64 = 1e-6, 1e-6      # f64 point storage


def _check(got, ref, n_src, tol_t, tol_r):
    dt, dr = syn.se3_error(got["transformation"], ref["transformation"])
    assert dt <= tol_t and dr <= tol_r, (dt, dr)
    assert abs(got["fitness"] - ref["fitness"]) <= 4.0 / n_src
    assert abs(got["inlier_rmse"] - ref["inlier_rmse"]) <= 1e-3 * max(ref["inlier_rmse"], 1e-9)
    return dt, dr


def test_small_fixed_iterations_f32(backend_f32, oracle, small_c2):
    src, tgt, nrm, _ = small_c2
    got = backend_f32.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    assert got["iterations"] == 10 and not got["converged"]
    dt, dr = _check(got, ref, len(src), TOL_T, TOL_R)
    print("f32 small:", dt, dr)


def test_small_fixed_iterations_f64(backend_f64, oracle, small_c2):
    src, tgt, nrm, _ = small_c2
    got = backend_f64.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    dt, dr = _check(got, ref, len(src), TOL_T64, TOL_R64)
    assert got["n_corr"] == ref["n_corr"]
    assert abs(got["fitness"] - ref["fitness"]) < 1e-12
    print("f64 small:", dt, dr)


def test_every_iteration_count_matches(backend_f64, oracle, small_c2):
    """k = 0..6 fixed iterations: the whole trajectory of poses matches, not only the end point."""
    src, tgt, nrm, _ = small_c2
    tree = oracle.KDTree(tgt)
    s = backend_f64.upload(src)
    t = backend_f64.upload(tgt, nrm)
    backend_f64.build_index(t, 1.0)
    for k in range(0, 7):
        got = backend_f64.icp_point_to_plane_dev(s, t, 1.0, max_iter=k, rel_fitness=0.0, rel_rmse=0.0)
        ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=k, rel_fitness=0.0, rel_rmse=0.0, tree=tree)
        assert got["iterations"] == k
        _check(got, ref, len(src), TOL_T64, TOL_R64)
    backend_f64.free(s)
    backend_f64.free(t)


def test_convergence_criteria_default(backend_f64, backend_f32, oracle, small_c2):
    src, tgt, nrm, _ = small_c2
    ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
    got = backend_f64.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
    assert ref["converged"] and got["converged"]
    assert got["iterations"] == ref["iterations"]
    _check(got, ref, len(src), TOL_T64, TOL_R64)
    got32 = backend_f32.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
    assert got32["converged"] and abs(got32["iterations"] - ref["iterations"]) <= 1  # 1e-6 thresholds vs f32 quantisation
    _check(got32, ref, len(src), TOL_T, TOL_R)


def test_non_identity_init_and_cell_sizes(backend_f32, oracle, small_c2):
    src, tgt, nrm, _ = small_c2
    T0 = syn.make_pose([0.2, -0.1, 0.02], [0.2, 0.1, 1.0])
    ref = oracle.icp_point_to_plane(src, tgt, nrm, 0.7, init=T0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    s = backend_f32.upload(src)
    t = backend_f32.upload(tgt, nrm)
    res = []
    for cell in (0.05, 0.175, 0.4, 1.3):  # the NN search is exact for any cell size
        backend_f32.build_index(t, 0.7, cell)
        got = backend_f32.icp_point_to_plane_dev(s, t, 0.7, init=T0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
        _check(got, ref, len(src), TOL_T, TOL_R)
        res.append(got)
    for r in res[1:]:  # identical correspondences => identical sums up to nothing: bitwise equal results
        np.testing.assert_array_equal(r["transformation"], res[0]["transformation"])
        assert r["n_corr"] == res[0]["n_corr"]
    backend_f32.free(s)
    backend_f32.free(t)


def test_deterministic_bitwise(backend_f32, small_c2):
    src, tgt, nrm, _ = small_c2
    a = backend_f32.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
    b = backend_f32.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
    np.testing.assert_array_equal(a["transformation"], b["transformation"])
    assert a["inlier_rmse"] == b["inlier_rmse"] and a["fitness"] == b["fitness"]


def test_fused_map_crop_equals_cropped_target(backend_f64, oracle, small_c2):
    """scanMatcherCropper_->crop(map) (ScanToMapRegistration.cpp:58-59) fused into the search == ICP on the cropped copy."""
    src, tgt, nrm, _ = small_c2
    crop_o = oracle.make_crop(oracle.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=20.0)
    keep = oracle.crop_indices(tgt, crop_o)
    assert 0 < len(keep) < len(tgt)
    ref = oracle.icp_point_to_plane(src, tgt[keep], nrm[keep], 1.0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
    s = backend_f64.upload(src)
    t = backend_f64.upload(tgt, nrm)
    crop = backend.make_crop(backend.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=20.0)
    got = backend_f64.icp_point_to_plane_dev(s, t, 1.0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0, target_crop=crop)
    _check(got, ref, len(src), TOL_T64, TOL_R64)
    assert got["n_corr"] == ref["n_corr"]
    backend_f64.free(s)
    backend_f64.free(t)


def test_stepwise_equals_oneshot(backend_f32, small_c2):
    import torch

    src, tgt, nrm, _ = small_c2
    s = backend_f32.upload(src)
    t = backend_f32.upload(tgt, nrm)
    backend_f32.build_index(t, 1.0)
    one = backend_f32.icp_point_to_plane_dev(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    rec = torch.zeros(32, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    backend_f32.icp_begin(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    for _ in range(7):
        backend_f32.icp_accumulate(0, len(src), rec.data_ptr())
        backend_f32.icp_update(rec.data_ptr(), len(src))
    assert backend_f32.icp_done()
    step = backend_f32.icp_finish()
    np.testing.assert_array_equal(step["transformation"], one["transformation"])
    assert step["iterations"] == 6 and step["fitness"] == one["fitness"]
    # two shards summed by the caller == one shard (up to fp reassociation of the 2-way split)
    rec2 = torch.zeros(32, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    backend_f32.icp_begin(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    h = len(src) // 2
    for _ in range(7):
        backend_f32.icp_accumulate(0, h, rec.data_ptr())
        backend_f32.icp_accumulate(h, len(src) - h, rec2.data_ptr())
        backend_f32.synchronize()
        tot = rec + rec2
        torch.cuda.synchronize()
        backend_f32.icp_update(tot.data_ptr(), len(src))
        backend_f32.synchronize()
    two = backend_f32.icp_finish()
    np.testing.assert_allclose(two["transformation"], one["transformation"], atol=1e-10)
    assert two["n_corr"] == one["n_corr"]
    backend_f32.free(s)
    backend_f32.free(t)


def test_error_conventions(backend_f32, small_c2):
    src, tgt, nrm, _ = small_c2
    with pytest.raises(backend.BackendError) as e:
        backend_f32.icp_point_to_plane(src, tgt, nrm, 0.0)
    assert e.value.code == backend.ERR_INVALID_ARG and "max_correspondence_distance" in str(e.value)
    with pytest.raises(backend.BackendError) as e:
        backend_f32.icp_point_to_plane(src, tgt, None, 1.0)
    assert e.value.code == backend.ERR_NO_NORMALS
    with pytest.raises(backend.BackendError) as e:
        backend_f32.icp_point_to_plane(src, np.zeros((0, 3)), np.zeros((0, 3)), 1.0)
    assert e.value.code == backend.ERR_EMPTY
    with pytest.raises(backend.BackendError):
        backend_f32.free(123456)


def test_no_correspondences_and_empty_source(backend_f32, oracle, small_c2):
    src, tgt, nrm, _ = small_c2
    far = src + 1000.0
    T0 = syn.make_pose([0.1, 0, 0], [0, 0, 0.3])
    got = backend_f32.icp_point_to_plane(far, tgt, nrm, 1.0, init=T0, max_iter=5)
    ref = oracle.icp_point_to_plane(far, tgt, nrm, 1.0, init=T0, max_iter=5)
    assert got["fitness"] == 0.0 and got["inlier_rmse"] == 0.0 and got["n_corr"] == 0
    np.testing.assert_allclose(got["transformation"], T0, atol=1e-15)  # identity updates only
    assert got["iterations"] == ref["iterations"] == 1 and got["converged"] and ref["converged"]
    got = backend_f32.icp_point_to_plane(np.zeros((0, 3)), tgt, nrm, 1.0, max_iter=3)
    assert got["fitness"] == 0.0 and got["n_corr"] == 0


def test_ragged_sizes(backend_f64, oracle):
    """Sizes that are not multiples of the wavefront/workgroup; targets smaller than one cell row."""
    scene = syn.make_scene()
    tgt, nrm = syn.sample_map(scene, 30_011)
    for n_az in (1, 3, 37):
        src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=n_az)
        got = backend_f64.icp_point_to_plane(src, tgt, nrm, 1.5, max_iter=4, rel_fitness=0.0, rel_rmse=0.0)
        ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.5, max_iter=4, rel_fitness=0.0, rel_rmse=0.0)
        assert got["n_corr"] == ref["n_corr"]
        if not np.isfinite(ref["transformation"]).all():
            # one azimuth column = points in one plane through the sensor: JtJ is singular and, as in Open3D (no PSD /
            # determinant check on this code path), NaNs propagate -- on both sides
            assert n_az == 1 and not np.isfinite(got["transformation"]).all()
            continue
        _check(got, ref, len(src), 1e-5, 1e-5)  # few points => ill-conditioned, still tight in f64
    tiny_t, tiny_n = tgt[:7], nrm[:7]
    got = backend_f64.icp_point_to_plane(tgt[:50], tiny_t, tiny_n, 5.0, max_iter=1, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_point_to_plane(tgt[:50], tiny_t, tiny_n, 5.0, max_iter=1, rel_fitness=0.0, rel_rmse=0.0)
    assert got["n_corr"] == ref["n_corr"] and abs(got["inlier_rmse"] - ref["inlier_rmse"]) < 1e-9


def test_golden_fixture(backend_f64, backend_f32):
    g = np.load(os.path.join(GOLD, "icp_scan_to_map.npz"))
    src, tgt, nrm, _ = syn.config2_inputs(n_map=int(g["n_map"]), n_az=int(g["n_az"]))
    got = backend_f64.icp_point_to_plane(src, tgt, nrm, float(g["max_corr"]), max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    dt, dr = syn.se3_error(got["transformation"], g["T10"])
    assert dt < TOL_T64 and dr < TOL_R64
    assert abs(got["fitness"] - float(g["fitness10"])) < 1e-12
    got = backend_f64.icp_point_to_plane(src, tgt, nrm, float(g["max_corr"]), max_iter=30)
    assert got["iterations"] == int(g["iters_conv"])
    got = backend_f32.icp_point_to_plane(src, tgt, nrm, float(g["max_corr"]), max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    dt, dr = syn.se3_error(got["transformation"], g["T10"])
    assert dt < TOL_T and dr < TOL_R


def test_full_size_config2(backend_f32, oracle):
    """BASELINE.json configs[1]: 65 536-pt scan vs 1 000 000-pt map, 10 iterations, vs the oracle + ground truth."""
    src, tgt, nrm, T_gt = syn.config2_inputs()
    assert len(src) == 65536 and len(tgt) == 1_000_000
    got = backend_f32.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    dt, dr = _check(got, ref, len(src), TOL_T, TOL_R)
    gt_t, gt_r = syn.se3_error(got["transformation"], T_gt)
    print(f"C2 full: vs oracle dt={dt:.3e} dr={dr:.3e}; vs truth dt={gt_t:.3e} dr={gt_r:.3e}; fitness={got['fitness']}")
    assert gt_t < 5e-3 and gt_r < 5e-4


def test_full_size_properties(backend_f32):
    """Size-independent properties at full size: registering a map subset against the map is a fixed point
    (fitness 1, rmse 0, T = I), and a pure translation along a wall normal is recovered."""
    _, tgt, nrm, _ = syn.config2_inputs()
    sub = tgt[::16][:65536]
    s = backend_f32.upload(sub)
    t = backend_f32.upload(tgt, nrm)
    backend_f32.build_index(t, 1.0)
    got = backend_f32.icp_point_to_plane_dev(s, t, 1.0, max_iter=3, rel_fitness=0.0, rel_rmse=0.0)
    assert got["fitness"] == 1.0 and got["inlier_rmse"] == 0.0
    np.testing.assert_allclose(got["transformation"], np.eye(4), atol=1e-12)
    shifted = backend_f32.upload(sub + np.array([0.05, -0.04, 0.03]))
    got = backend_f32.icp_point_to_plane_dev(shifted, t, 1.0, max_iter=15, rel_fitness=0.0, rel_rmse=0.0)
    np.testing.assert_allclose(got["transformation"][:3, 3], [-0.05, 0.04, -0.03], atol=2e-3)
    for c in (s, t, shifted):
        backend_f32.free(c)


def test_sharded_driver_single_rank_on_gpu(backend_f32, small_c2):
    """open3d_slam_amd.sharded.ShardedIcp with world_size 1 (no process group): the step-wise ABI driven on a torch side
    stream (o3ds_set_stream) reproduces the one-shot registration bit for bit, in both partitionings."""
    from open3d_slam_amd import sharded

    src, tgt, nrm, _ = small_c2
    s = backend_f32.upload(src)
    t = backend_f32.upload(tgt, nrm)
    backend_f32.build_index(t, 1.0)
    one = backend_f32.icp_point_to_plane_dev(s, t, 1.0, max_iter=30)
    try:
        for mode in ("source", "submap"):
            drv = sharded.ShardedIcp(backend_f32, mode=mode)
            got = drv.register(s, t, len(src), 1.0, max_iter=30, check_every=3)
            np.testing.assert_array_equal(got["transformation"], one["transformation"])
            assert got["iterations"] == one["iterations"] and got["converged"] == one["converged"]
            assert got["fitness"] == one["fitness"] and got["inlier_rmse"] == one["inlier_rmse"]
    finally:
        backend_f32.set_stream(None)
    again = backend_f32.icp_point_to_plane_dev(s, t, 1.0, max_iter=30)
    np.testing.assert_array_equal(again["transformation"], one["transformation"])
    backend_f32.free(s)
    backend_f32.free(t)


def test_two_launch_mode_is_bitwise_the_default(oracle, small_c2, backend_f32, monkeypatch):
    """O3DS_ICP_MODE=launch (pass kernel + one-workgroup update kernel per iteration) sums the per-workgroup records in the same
    order as the default fused form (slot = row % 32, rows ascending, slots ascending), so the two agree bit for bit."""
    src, tgt, nrm, _ = small_c2
    monkeypatch.setenv("O3DS_ICP_MODE", "launch")
    be = backend.Backend(0, backend.PRECISION_F32, ab=True)
    try:
        for kw in (dict(max_iter=10, rel_fitness=0.0, rel_rmse=0.0), dict(max_iter=30)):
            got = be.icp_point_to_plane(src, tgt, nrm, 1.0, **kw)
            dflt = backend_f32.icp_point_to_plane(src, tgt, nrm, 1.0, **kw)
            np.testing.assert_array_equal(got["transformation"], dflt["transformation"])
            assert got["iterations"] == dflt["iterations"] and got["fitness"] == dflt["fitness"] and got["inlier_rmse"] == dflt["inlier_rmse"]
            ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, **kw)
            assert got["iterations"] == ref["iterations"] and got["converged"] == ref["converged"]
            _check(got, ref, len(src), TOL_T, TOL_R)
    finally:
        be.close()


def test_fused_prologue_kernel_mode(oracle, small_c2, monkeypatch):
    """Default form (O3DS_ICP_MODE=fused): ONE launch per pass -- the previous pass's tail (record fold, convergence test, 6x6 solve, T <- U*T) runs
    in every workgroup's prologue.  Same loop semantics (iterations / converged / early stop / empty set) as the oracle, both
    precisions, point-to-plane and generalized, with and without a crop, and bitwise repeatable."""
    monkeypatch.setenv("O3DS_ICP_MODE", "fused")
    src, tgt, nrm, _ = small_c2
    for prec, tt, tr in ((backend.PRECISION_F64, TOL_T64, TOL_R64), (backend.PRECISION_F32, TOL_T, TOL_R)):
        be = backend.Backend(0, prec, ab=True)
        try:
            for kw in (dict(max_iter=10, rel_fitness=0.0, rel_rmse=0.0), dict(max_iter=30), dict(max_iter=0), dict(max_iter=1)):
                got = be.icp_point_to_plane(src, tgt, nrm, 1.0, **kw)
                ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, **kw)
                assert got["iterations"] == ref["iterations"] and got["converged"] == ref["converged"], kw
                _check(got, ref, len(src), tt, tr)
            a = be.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=7, rel_fitness=0.0, rel_rmse=0.0)
            b = be.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=7, rel_fitness=0.0, rel_rmse=0.0)
            np.testing.assert_array_equal(a["transformation"], b["transformation"])
            far = be.icp_point_to_plane(src + 1000.0, tgt, nrm, 1.0, max_iter=5)
            assert far["fitness"] == 0.0 and far["iterations"] == 1 and far["converged"]
            few = src[:: len(src) // 37][:37]  # one workgroup: 63 of the 64 record slots stay empty
            tiny = be.icp_point_to_plane(few, tgt, nrm, 1.0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
            ref = oracle.icp_point_to_plane(few, tgt, nrm, 1.0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
            _check(tiny, ref, 37, tt, tr)
        finally:
            be.close()
    be = backend.Backend(0, backend.PRECISION_F64, ab=True)
    try:
        sn = oracle.estimate_normals(src, 3.0, 20)
        ref = oracle.icp_generalized(src, sn, tgt, nrm, 1.0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
        got = be.icp_generalized(src, sn, tgt, nrm, 1.0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
        assert got["iterations"] == ref["iterations"]
        _check(got, ref, len(src), TOL_T64, TOL_R64)
    finally:
        be.close()


# ---- generalized ICP (what the shipped Lua configs select) --------------------------------------------
def test_gicp_matches_oracle(backend_f64, backend_f32, oracle, small_c2):
    src, tgt, nrm, T_gt = small_c2
    sn = oracle.estimate_normals(src, 3.0, 20)
    for kw in (dict(max_iter=8, rel_fitness=0.0, rel_rmse=0.0), dict(max_iter=30)):
        ref = oracle.icp_generalized(src, sn, tgt, nrm, 1.0, **kw)
        got = backend_f64.icp_generalized(src, sn, tgt, nrm, 1.0, **kw)
        assert got["iterations"] == ref["iterations"] and got["converged"] == ref["converged"]
        dt, dr = _check(got, ref, len(src), TOL_T64, TOL_R64)
        assert got["n_corr"] == ref["n_corr"]
        got32 = backend_f32.icp_generalized(src, sn, tgt, nrm, 1.0, **kw)
        _check(got32, ref, len(src), TOL_T, TOL_R)
    # non-identity init: the source covariances rotate with the cloud
    T0 = syn.make_pose([0.1, -0.1, 0.0], [1.0, -2.0, 3.0])
    ref = oracle.icp_generalized(src, sn, tgt, nrm, 1.0, init=T0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
    got = backend_f64.icp_generalized(src, sn, tgt, nrm, 1.0, init=T0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
    _check(got, ref, len(src), TOL_T64, TOL_R64)
    # normals close to -e1 take GetRotationFromE1ToX's special case on both sides
    sn2, nrm2 = sn.copy(), nrm.copy()
    sn2[::7] = [-1.0, 0.0, 0.0]
    nrm2[::5] = [-0.999, 0.0447101778, 0.0]
    nrm2[::5] /= np.linalg.norm(nrm2[::5], axis=1, keepdims=True)
    ref = oracle.icp_generalized(src, sn2, tgt, nrm2, 1.0, max_iter=4, rel_fitness=0.0, rel_rmse=0.0)
    got = backend_f64.icp_generalized(src, sn2, tgt, nrm2, 1.0, max_iter=4, rel_fitness=0.0, rel_rmse=0.0)
    _check(got, ref, len(src), TOL_T64, TOL_R64)


def test_gicp_device_forms_and_errors(backend_f32, oracle, small_c2):
    import torch

    src, tgt, nrm, _ = small_c2
    sn = oracle.estimate_normals(src, 3.0, 20)
    s, t = backend_f32.upload(src, sn), backend_f32.upload(tgt, nrm)
    backend_f32.build_index(t, 1.0)
    one = backend_f32.icp_generalized_dev(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    p2p = backend_f32.icp_point_to_plane_dev(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    assert not np.array_equal(one["transformation"], p2p["transformation"])  # different estimator, not a silent alias
    rec = torch.zeros(32, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    backend_f32.icp_begin(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0, method=backend.ICP_GENERALIZED)
    for _ in range(7):
        backend_f32.icp_accumulate(0, len(src), rec.data_ptr())
        backend_f32.icp_update(rec.data_ptr(), len(src))
    step = backend_f32.icp_finish()
    np.testing.assert_array_equal(step["transformation"], one["transformation"])
    # fused map crop
    crop_o = oracle.make_crop(oracle.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=20.0)
    keep = oracle.crop_indices(tgt, crop_o)
    ref = oracle.icp_generalized(src, sn, tgt[keep], nrm[keep], 1.0, max_iter=4, rel_fitness=0.0, rel_rmse=0.0)
    got = backend_f32.icp_generalized_dev(s, t, 1.0, max_iter=4, rel_fitness=0.0, rel_rmse=0.0,
                                          target_crop=backend.make_crop(backend.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=20.0))
    _check(got, ref, len(src), TOL_T, TOL_R)
    with pytest.raises(backend.BackendError):
        backend_f32.set_gicp_epsilon(0.0)
    for c in (s, t):
        backend_f32.free(c)


def test_gicp_without_normals_estimates_them_like_open3d(backend_f64, backend_f32, oracle, small_c2):
    """[O3D] InitializePointCloudForGeneralizedICP (call site CloudRegistration.cpp:16-21): a cloud that has no normals gets
    EstimateNormals(KDTreeSearchParamKNN(20)) -- no NormalizeNormals, no orientation -- on a copy"""
    src, tgt, nrm, _ = small_c2
    sn, tn = oracle.estimate_normals_knn_raw(src, 20), oracle.estimate_normals_knn_raw(tgt, 20)
    kw = dict(max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_generalized(src, sn, tgt, tn, 1.0, **kw)
    got = backend_f64.icp_generalized(src, None, tgt, None, 1.0, **kw)
    assert got["n_corr"] == ref["n_corr"]
    _check(got, ref, len(src), TOL_T64, TOL_R64)
    # one side only: the given normals are used as they are, the other side is estimated
    ref1 = oracle.icp_generalized(src, sn, tgt, nrm, 1.0, **kw)
    got1 = backend_f64.icp_generalized(src, None, tgt, nrm, 1.0, **kw)
    _check(got1, ref1, len(src), TOL_T64, TOL_R64)
    # device form: the caller's clouds stay without normals
    s, t = backend_f32.upload(src), backend_f32.upload(tgt)
    got32 = backend_f32.icp_generalized_dev(s, t, 1.0, **kw)
    _check(got32, ref, len(src), TOL_T, TOL_R)
    assert backend_f32.size(s) == (len(src), False) and backend_f32.size(t) == (len(tgt), False)
    for c in (s, t):
        backend_f32.free(c)


def test_gicp_full_size_config2(backend_f32, oracle):
    src, tgt, nrm, T_gt = syn.config2_inputs()
    sn = oracle.estimate_normals(src, 3.0, 20)
    got = backend_f32.icp_generalized(src, sn, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_generalized(src, sn, tgt, nrm, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    dt, dr = _check(got, ref, len(src), TOL_T, TOL_R)
    gt_t, gt_r = syn.se3_error(got["transformation"], T_gt)
    print(f"GICP C2 full: vs oracle {dt:.2e} {dr:.2e}; vs truth {gt_t:.2e} {gt_r:.2e}")
    assert gt_t < 5e-3 and gt_r < 5e-4


# ---- point-to-point ICP (SURVEY.md 8f rank 1 "trivial variant": CloudRegistration.cpp:69-81) ------------------------------------
def test_point_to_point_matches_oracle(backend_f64, backend_f32, oracle, small_c2):
    """[O3D] RegistrationICP + TransformationEstimationPointToPoint (Eigen::umeyama, no scaling): same loop semantics as the oracle
    for fixed iteration counts and for the default convergence test, in both storage precisions; no normals needed."""
    src, tgt, _, _ = small_c2
    for kw in (dict(max_iter=8, rel_fitness=0.0, rel_rmse=0.0), dict(max_iter=40), dict(max_iter=0), dict(max_iter=1, rel_fitness=0.0, rel_rmse=0.0)):
        ref = oracle.icp_point_to_point(src, tgt, 1.0, **kw)
        got = backend_f64.icp_point_to_point(src, tgt, 1.0, **kw)
        assert got["iterations"] == ref["iterations"] and got["converged"] == ref["converged"], kw
        assert got["n_corr"] == ref["n_corr"]
        _check(got, ref, len(src), TOL_T64, TOL_R64)
        got32 = backend_f32.icp_point_to_point(src, tgt, 1.0, **kw)
        assert got32["iterations"] == ref["iterations"]
        _check(got32, ref, len(src), TOL_T, TOL_R)
    # a non-identity initial guess, and an empty correspondence set (-> identity updates, converges at once)
    T0 = syn.make_pose((0.1, -0.05, 0.02), (0.5, 0.2, -1.0))
    ref = oracle.icp_point_to_point(src, tgt, 1.0, init=T0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
    got = backend_f64.icp_point_to_point(src, tgt, 1.0, init=T0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
    _check(got, ref, len(src), TOL_T64, TOL_R64)
    far = backend_f64.icp_point_to_point(src + 1000.0, tgt, 1.0, max_iter=5)
    assert far["fitness"] == 0.0 and far["iterations"] == 1 and far["converged"]
    np.testing.assert_allclose(far["transformation"], np.eye(4), atol=0)


def test_point_to_point_device_forms(backend_f32, oracle, small_c2, monkeypatch):
    """one-shot fused == two-launch == step-wise (bit for bit), the fused map crop, determinism, and a different estimator than
    point-to-plane (not a silent alias)."""
    import torch

    src, tgt, nrm, _ = small_c2
    s, t = backend_f32.upload(src), backend_f32.upload(tgt)  # target without normals
    backend_f32.build_index(t, 1.0)
    one = backend_f32.icp_point_to_point_dev(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    again = backend_f32.icp_point_to_point_dev(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    np.testing.assert_array_equal(one["transformation"], again["transformation"])
    rec = torch.zeros(32, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    backend_f32.icp_begin(s, t, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0, method=backend.ICP_POINT_TO_POINT)
    for _ in range(7):
        backend_f32.icp_accumulate(0, len(src), rec.data_ptr())
        backend_f32.icp_update(rec.data_ptr(), len(src))
    step = backend_f32.icp_finish()
    np.testing.assert_array_equal(step["transformation"], one["transformation"])
    monkeypatch.setenv("O3DS_ICP_MODE", "launch")
    be2 = backend.Backend(0, backend.PRECISION_F32, ab=True)
    try:
        two = be2.icp_point_to_point(src, tgt, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
        np.testing.assert_array_equal(two["transformation"], one["transformation"])
    finally:
        be2.close()
    with pytest.raises(backend.BackendError):  # point-to-plane on the same normal-less target must still fail loudly
        backend_f32.icp_point_to_plane_dev(s, t, 1.0, max_iter=2)
    tn = backend_f32.upload(tgt, nrm)
    p2plane = backend_f32.icp_point_to_plane_dev(s, tn, 1.0, max_iter=6, rel_fitness=0.0, rel_rmse=0.0)
    assert not np.array_equal(p2plane["transformation"], one["transformation"])
    crop_o = oracle.make_crop(oracle.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=20.0)
    keep = oracle.crop_indices(tgt, crop_o)
    ref = oracle.icp_point_to_point(src, tgt[keep], 1.0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
    crop = backend.make_crop(backend.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=20.0)
    got = backend_f32.icp_point_to_point_dev(s, t, 1.0, max_iter=5, rel_fitness=0.0, rel_rmse=0.0, target_crop=crop)
    _check(got, ref, len(src), TOL_T, TOL_R)
    for c in (s, t, tn):
        backend_f32.free(c)


def test_point_to_point_through_the_reference_named_classes(backend_f32, oracle, small_c2):
    from open3d_slam_amd import parameters as P
    from open3d_slam_amd.cloud_registration import cloudRegistrationFactory
    from open3d_slam_amd.pointcloud import PointCloud

    src, tgt, _, _ = small_c2
    p = P.CloudRegistrationParameters()
    p.regType_ = P.CloudRegistrationType.PointToPointIcp
    p.icp_ = P.IcpParameters(maxNumIter_=25, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
    reg = cloudRegistrationFactory(p)
    a, b = PointCloud.from_numpy(backend_f32, src), PointCloud.from_numpy(backend_f32, tgt)
    reg.estimateNormalsOrCovariancesIfNeeded(b)  # no-op for this estimator
    assert not b.HasNormals()
    r = reg.registerClouds(a, b, np.eye(4))
    ref = oracle.icp_point_to_point(src, tgt, 1.0, max_iter=25)
    dt, dr = syn.se3_error(r.transformation_, ref["transformation"])
    assert dt <= TOL_T and dr <= TOL_R and abs(r.fitness_ - ref["fitness"]) <= 4.0 / len(src)
    a.release()
    b.release()


# ---- information matrix (constraint_builders.cpp:70-73, PlaceRecognition.cpp:148-149) ----------------
def test_information_matrix_matches_oracle(backend_f64, backend_f32, oracle, small_c2):
    src, tgt, nrm, T_gt = small_c2
    for T in (T_gt, np.eye(4)):
        ref = oracle.information_matrix(src, tgt, 1.0, T)
        got = backend_f64.information_matrix(src, tgt, 1.0, T)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-7)  # same correspondences, f64 sums in another order
        assert got[3, 3] == ref[3, 3]  # the correspondence count is exact
        got32 = backend_f32.information_matrix(src, tgt, 1.0, T)
        np.testing.assert_allclose(got32, ref, rtol=0, atol=2e-5 * np.abs(ref).max())  # f32 point storage: relative to the matrix scale
    s, t = backend_f64.upload(src), backend_f64.upload(tgt, nrm)
    backend_f64.build_index(t, 1.0)
    dev = backend_f64.information_matrix_dev(s, t, 1.0, T_gt)
    np.testing.assert_array_equal(dev, backend_f64.information_matrix(src, tgt, 1.0, T_gt))
    # a registration before and after leaves it unchanged (the match cache of the ICP loop is not trusted across sessions)
    backend_f64.icp_point_to_plane_dev(s, t, 1.0, max_iter=3)
    np.testing.assert_array_equal(backend_f64.information_matrix_dev(s, t, 1.0, T_gt), dev)
    assert np.array_equal(backend_f64.information_matrix(src + 1000.0, tgt, 1.0), np.zeros((6, 6)))
    with pytest.raises(backend.BackendError):
        backend_f64.information_matrix(src, tgt, 0.0)
    backend_f64.free(s)
    backend_f64.free(t)


def test_large_coordinates_and_exact_sums(backend_f64, oracle, small_c2):
    """Clouds far from the origin (UTM-like offsets).  The problem is ill-conditioned by construction -- a 2.2e5 m lever arm lets
    rotation and translation trade against each other -- so the yardstick is the reference algorithm's own f64 noise floor on the
    SAME inputs: the disagreement between the C oracle and the independent numpy restatement (measured: 2.9e-5 m / 4.7e-9 rad; the C
    oracle differs from itself by 1.4e-5 m / 1.0e-8 rad between 1 and 16 threads; near the origin the pair agrees to 2e-16 m).  The
    GPU must sit within 3x of that floor.  History: with ONE global quantum for the exact record sums this registration was off by
    4.4e-4 m / 7e-8 rad (plain f64 sums: 1.7e-5 m), which is why the quanta are per record term; this test would have caught it.
    Second half: a different launch geometry (other per-workgroup partial sums) lands within the same floor."""
    from oracle import np_oracle

    src, tgt, nrm, _ = small_c2
    off = np.array([1.0e5, -2.0e5, 50.0])
    T0 = np.eye(4)
    T0[:3, 3] = off  # the source stays near the origin, the initial guess carries it to the map
    kw = dict(init=T0, max_iter=8, rel_fitness=0.0, rel_rmse=0.0)
    ref = oracle.icp_point_to_plane(src, tgt + off, nrm, 1.0, **kw)
    ref2 = np_oracle.icp_point_to_plane(src, tgt + off, nrm, 1.0, **kw)
    floor_t, floor_r = syn.se3_error(ref["transformation"], ref2["transformation"])
    assert 1e-7 < floor_t < 1e-3 and floor_r < 1e-6, (floor_t, floor_r)  # the premise: this input IS ill-conditioned, mildly
    lim_t, lim_r = 3.0 * max(floor_t, 1e-5), 3.0 * max(floor_r, 5e-9)
    got = backend_f64.icp_point_to_plane(src, tgt + off, nrm, 1.0, **kw)
    assert got["iterations"] == ref["iterations"] and got["n_corr"] == ref["n_corr"]
    dt, dr = syn.se3_error(got["transformation"], ref["transformation"])
    assert dt <= lim_t and dr <= lim_r, (dt, dr, lim_t, lim_r)
    again = backend_f64.icp_point_to_plane(src, tgt + off, nrm, 1.0, **kw)
    np.testing.assert_array_equal(again["transformation"], got["transformation"])  # and it is reproducible, unlike the CPU reduction
    os.environ["O3DS_PASS_ROWS"] = "333"
    try:
        be = backend.Backend(0, backend.PRECISION_F64, ab=True)
        other = be.icp_point_to_plane(src, tgt + off, nrm, 1.0, **kw)
        be.close()
    finally:
        del os.environ["O3DS_PASS_ROWS"]
    dt, dr = syn.se3_error(other["transformation"], got["transformation"])
    assert dt <= lim_t and dr <= lim_r, (dt, dr, lim_t, lim_r)


def test_rank_deficient_normal_equations_follow_eigen_ldlt(backend_f64, oracle):
    """A scene of ONE plane leaves x / y translation and yaw unobservable (exactly zero rows of J^T J), and one or two correspondences
    leave most of the six unknowns free.  [O3D] solves with Eigen's LDLT, which returns zero for the components of null pivots; the
    device's solve does the same (icp_kernels.hpp solve6_wave) -- finite poses equal to the oracle's, never inf / NaN."""
    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(np.arange(-8, 8, 0.25), np.arange(-8, 8, 0.25)), -1).reshape(-1, 2)
    tgt = np.c_[g, np.zeros(len(g))]
    nrm = np.tile([0.0, 0.0, 1.0], (len(tgt), 1))
    src = np.c_[rng.uniform(-6, 6, (500, 2)), 0.04 + 0.02 * rng.uniform(-1, 1, 500)]
    for s in (src, src[:2], src[:1]):
        got = backend_f64.icp_point_to_plane(s, tgt, nrm, 0.5, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
        ref = oracle.icp_point_to_plane(s, tgt, nrm, 0.5, max_iter=5, rel_fitness=0.0, rel_rmse=0.0)
        assert np.isfinite(got["transformation"]).all() and np.isfinite(ref["transformation"]).all()
        if len(s) < 3:  # the observable 3x3 block is itself singular: its last pivot is rounding residue, on both sides, and so is
            continue    # the "solution" -- what is guaranteed (and what Eigen guarantees) is a finite one
        np.testing.assert_allclose(got["transformation"], ref["transformation"], atol=1e-9)
        T = got["transformation"]
        # the unobservable motions get no update of their own (what little x / y appears is the tilt acting on the z offset)
        assert abs(T[0, 3]) < 1e-6 and abs(T[1, 3]) < 1e-6 and abs(T[1, 0]) < 1e-6


# ---- candidate sets: a steady pass verifies its matches instead of searching (icp_kernels.hpp, Collect) ----------------------
def _set_variants(monkeypatch, env):
    for k in ("O3DS_ICP_SETS", "O3DS_SET_GAIN", "O3DS_SET_MIN", "O3DS_SET_CAP"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("prec", [backend.PRECISION_F32, backend.PRECISION_F64])
def test_candidate_sets_do_not_change_a_single_bit(small_c2, oracle, monkeypatch, prec):
    """A pass that proves its matches inside the candidate sets the previous pass left (no search) yields the SAME correspondences as
    the search, hence bit-identical sums, poses, fitness and rmse -- for every margin policy: off, default, margins so small that
    nearly every verification fails, margins so large that the lists overflow.  Point-to-plane with and without a map crop,
    generalized, point-to-point; fixed iterations and the default convergence criteria."""
    src, tgt, nrm, _ = small_c2
    sn = oracle.estimate_normals(src, 3.0, 20)
    crop = backend.make_crop(backend.CROP_MAX_RADIUS, center=(1.0, -2.0, 0.0), rmax=22.0)
    policies = [{"O3DS_ICP_SETS": "0"}, {}, {"O3DS_SET_MIN": "1e-6", "O3DS_SET_GAIN": "0.01"},
                {"O3DS_SET_MIN": "0.04", "O3DS_SET_CAP": "10", "O3DS_SET_GAIN": "8"}, {"O3DS_SET_MIN": "0.3", "O3DS_SET_CAP": "10"}]
    results = []
    for env in policies:
        _set_variants(monkeypatch, env)
        be = backend.Backend(0, prec, ab=True)
        try:
            out = []
            s_id, t_id = be.upload(src, sn), be.upload(tgt, nrm)
            for kw in (dict(max_iter=12, rel_fitness=0.0, rel_rmse=0.0), dict(max_iter=30)):
                out.append(be.icp_point_to_plane_dev(s_id, t_id, 1.0, **kw))
                out.append(be.icp_point_to_plane_dev(s_id, t_id, 1.0, target_crop=crop, **kw))
                out.append(be.icp_generalized_dev(s_id, t_id, 1.0, **kw))
                out.append(be.icp_point_to_point_dev(s_id, t_id, 1.0, **kw))
            # a second registration on the same handle from another start: the sets of the first must not leak into it
            T0 = syn.make_pose([0.1, 0.05, -0.02], [0.2, 0.1, -0.5])
            out.append(be.icp_point_to_plane_dev(s_id, t_id, 1.0, init=T0, max_iter=12, rel_fitness=0.0, rel_rmse=0.0))
            results.append(out)
        finally:
            be.close()
    base = results[0]
    assert base[0]["iterations"] == 12 and base[0]["fitness"] > 0.5
    for env, out in zip(policies[1:], results[1:]):
        for a, b in zip(base, out):
            np.testing.assert_array_equal(a["transformation"], b["transformation"], err_msg=str(env))
            assert (a["iterations"], a["converged"], a["n_corr"], a["fitness"], a["inlier_rmse"]) == (
                b["iterations"], b["converged"], b["n_corr"], b["fitness"], b["inlier_rmse"]), env


@pytest.mark.parametrize("method", ["plane", "generalized"])
def test_a_registration_does_not_depend_on_when_the_host_learnt_the_size_of_its_scan(method, small_c2):
    """The head of a crop + VoxelDownSample chain has a size the host has not seen (an upper bound + a device word).  A registration that
    is queued at once deals its queries out over the bound; one that is queued after the size arrived used to deal them out over the
    exact number -- other workgroups summed other queries, and the pose differed in its last place (and the quanta of the exact sums
    followed the same number).  Which of the two a stream got was a matter of timing: every third two-handle run of configs[2] differed
    from the one-handle run by an ulp of one pose entry (round 6, scripts/debug_wobble.py).  Both must give the same bits."""
    be = backend.Backend(0)
    _, tgt, nrm, _ = small_c2
    t = be.upload(tgt, nrm)
    be.build_index(t, 1.0)
    scene = syn.make_scene()
    raw = np.ascontiguousarray(syn.os128_scan(scene, syn.make_pose([0.3, 0.2, 0.0], [0.0, 0.0, 2.0]), frame=3), dtype=np.float32)
    crop = backend.make_crop(backend.CROP_MAX_RADIUS, rmax=25.0)
    reg = be.icp_point_to_plane_dev if method == "plane" else be.icp_generalized_dev

    big = be.upload(np.random.default_rng(5).uniform(-15.0, 15.0, (4_000_000, 3)))
    shift = np.eye(4)

    def chain(learn_size):
        r = be.upload_f32(raw)  # the ingest of the stream: asynchronous, the box of the volume comes with it
        if not learn_size:  # a backlog on the stream (queued, never waited for): what follows cannot have run when the registration is queued
            for _ in range(60):
                be.free(be.transform_cloud(big, shift))
        v = be.crop_voxel_down_sample(r, crop, 0.1)
        be.estimate_normals(v, 2.0, 10)
        be.free(r)
        if learn_size:
            be.size(v)  # (waits for the size: exact from here on)
        lazy = be.size_bound(v)[1] == len(raw)  # still the bound a moment before the registration is queued?
        out = reg(v, t, 1.0, max_iter=12, rel_fitness=0.0, rel_rmse=0.0)
        n = be.size(v)[0]
        be.free(v)
        return out, n, lazy

    for _ in range(3):  # the first chains wait here and there (the pilot of the normal estimation, the first box): not the steady state
        chain(False)
    results = [chain(k % 2 == 1) for k in range(16)]
    assert not any(lazy for _, _, lazy in results[1::2])
    if not any(lazy for _, _, lazy in results[0::2]):
        pytest.skip("the size of the scan reached the host before every registration: nothing to compare on this machine")
    assert (results[0][1] + 63) // 64 < (len(raw) + 63) // 64  # the premise: the bound (the raw scan's size) asks for more workgroups than the voxel count
    for out, n, _ in results[1:]:
        assert n == results[0][1]
        np.testing.assert_array_equal(out["transformation"], results[0][0]["transformation"])
        assert (out["fitness"], out["inlier_rmse"], out["n_corr"]) == (results[0][0]["fitness"], results[0][0]["inlier_rmse"], results[0][0]["n_corr"])
    assert results[0][0]["fitness"] > 0.3
    be.free(t)
    be.free(big)
    be.close()


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_work_queued_behind_a_registration_changes_neither_result(prec, small_c2):
    """o3ds_icp_overlap_next: a callback that runs once the registration's launches are queued, before the host waits (the stream driver
    queues the pre-processing of the next scan there).  The registration's result and the nested calls' clouds must be, bit for bit,
    what they are without it; the callback runs exactly once, belongs to ONE registration, and what it raises reaches the caller."""
    be = backend.Backend(0, backend.PRECISION_F64 if prec == "f64" else backend.PRECISION_F32)
    src, tgt, nrm, _ = small_c2
    s = be.upload(src)
    t = be.upload(tgt, nrm)
    be.build_index(t, 1.0)
    scene = syn.make_scene()
    raw = syn.vlp16_scan(scene, syn.make_pose([1.0, 0.5, 0.0], [0.0, 0.0, 10.0]), frame=3, n_az=512)
    crop = backend.make_crop(backend.CROP_MAX_RADIUS, rmax=25.0)

    def chain():
        r = be.upload(raw)
        v = be.crop_voxel_down_sample(r, crop, 0.1)
        be.estimate_normals(v, 2.0, 10)
        be.free(r)
        return v

    plain = be.icp_point_to_plane_dev(s, t, 1.0, max_iter=12)
    v0 = chain()
    ref_p, ref_n = be.download(v0)
    be.free(v0)

    made, calls = [], []

    def behind():
        calls.append(1)
        made.append(chain())

    be.overlap_next = behind
    got = be.icp_point_to_plane_dev(s, t, 1.0, max_iter=12)
    assert calls == [1] and be.overlap_next is None
    assert np.array_equal(got["transformation"], plain["transformation"]) and got["iterations"] == plain["iterations"]
    assert got["fitness"] == plain["fitness"] and got["inlier_rmse"] == plain["inlier_rmse"]
    p, n = be.download(made[0])
    assert p.tobytes() == ref_p.tobytes() and n.tobytes() == ref_n.tobytes()
    be.free(made.pop())

    again = be.icp_point_to_plane_dev(s, t, 1.0, max_iter=12)  # the next registration inherits nothing
    assert calls == [1] and np.array_equal(again["transformation"], plain["transformation"])

    def raises():
        raise ValueError("from inside the callback")

    be.overlap_next = raises
    with pytest.raises(ValueError, match="from inside the callback"):
        be.icp_point_to_plane_dev(s, t, 1.0, max_iter=12)
    # a registration that fails before it queues anything: the work is still done, by the wrapper, and the error is the registration's
    be.overlap_next = behind
    with pytest.raises(backend.BackendError):
        be.icp_point_to_plane_dev(s, 987654, 1.0, max_iter=12)
    assert calls == [1, 1] and len(made) == 1
    be.free(made.pop())
    be.free(s)
    be.free(t)
    be.close()
