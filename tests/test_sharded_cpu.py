"""The N>1 path on CPU: world_size-2 gloo process group driving open3d_slam_amd.sharded.run_sharded_loop.
No GPU here, so the per-rank correspondence/reduction pass is played by the CPU oracle (the checker standing in
for the kernel); what is under test is the host logic: sharding, the 32-double record all-reduce, lock-step
termination, and that both partitionings reproduce the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from open3d_slam_amd import sharded
from open3d_slam_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, w) for r in range(w)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for f, c in spans:
                assert f == pos
                pos += c
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


class _HostIcpState:
    """Host restatement of the device-side step (icp_kernels.hpp icp_step_from_record) for the CPU stand-in."""

    def __init__(self, init, max_iter, rel_fit, rel_rmse):
        self.T = np.array(init, dtype=np.float64)
        self.max_iter, self.rel_fit, self.rel_rmse = max_iter, rel_fit, rel_rmse
        self.fitness = self.rmse = 0.0
        self.n_corr = 0
        self.passes = self.iterations = 0
        self.done = self.converged = False

    def step(self, rec, n_total, oracle):
        if self.done:
            return
        cnt = rec[28]
        fit = cnt / n_total if cnt > 0 else 0.0
        rmse = float(np.sqrt(rec[29] / cnt)) if cnt > 0 else 0.0
        conv = self.passes > 0 and abs(self.fitness - fit) < self.rel_fit and abs(self.rmse - rmse) < self.rel_rmse
        self.fitness, self.rmse, self.n_corr = fit, rmse, int(cnt + 0.5)
        self.passes += 1
        if conv:
            self.converged = self.done = True
            return
        if self.iterations >= self.max_iter:
            self.done = True
            return
        if cnt > 0:
            A = np.zeros((6, 6))
            A[np.triu_indices(6)] = rec[:21]
            A = A + A.T - np.diag(np.diag(A))
            U, _ = oracle.solve_update(A, rec[21:27])
        else:
            U = np.eye(4)
        self.T = U @ self.T
        self.iterations += 1


def _record(oracle, tree, src, tgt, nrm, T, max_corr):
    """One correspondence + reduction pass of `src` under T, as the 32-double record of include/o3ds_backend.h."""
    P = src @ T[:3, :3].T + T[:3, 3]
    rec = np.zeros(32)
    if len(P) == 0:
        return rec
    corr, d2, _, _, nc = oracle.evaluate(tree, P, max_corr)
    JTJ, JTr, r2 = oracle.compute_jtj_jtr(P, tgt, nrm, corr)
    rec[:21] = JTJ[np.triu_indices(6)]
    rec[21:27] = JTr
    rec[27], rec[28], rec[29] = r2, nc, d2[corr >= 0].sum()
    return rec


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as oracle

    oracle.lib().orc_set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = syn.make_scene()
    src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=128)
    seed = syn.SEED_MAP + (rank if mode == "submap" else 0)
    tgt, nrm = syn.sample_map(scene, 40_000, seed=seed)
    tree = oracle.KDTree(tgt)
    max_iter = 30
    st = _HostIcpState(np.eye(4), max_iter, 1e-6, 1e-6)
    if mode == "source":
        first, count = sharded.shard_range(len(src), rank, world)
        n_total = len(src)
    else:
        first, count = 0, len(src)
        n_total = len(src) * world
    rec_t = torch.zeros(32, dtype=torch.float64)

    def accumulate():
        rec_t.copy_(torch.from_numpy(_record(oracle, tree, src[first:first + count], tgt, nrm, st.T, 1.0)))
        return rec_t

    passes = sharded.run_sharded_loop(accumulate, lambda r: dist.all_reduce(r), lambda r: st.step(r.numpy(), n_total, oracle),
                                      lambda: st.done, max_iter, check_every=1)
    Ts = [torch.zeros(16, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Ts, torch.from_numpy(st.T.ravel().copy()))
    if rank == 0:
        np.savez(out_path, T=st.T, fitness=st.fitness, rmse=st.rmse, iterations=st.iterations, converged=st.converged, passes=passes,
                 all_T=np.stack([t.numpy() for t in Ts]))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["source", "submap"])
def test_world2_gloo_matches_single_process(tmp_path, oracle, mode):
    out = str(tmp_path / f"{mode}.npz")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    r = np.load(out)
    # every rank ended with the identical pose (same reduced record => same update, no broadcast needed)
    np.testing.assert_array_equal(r["all_T"][0], r["all_T"][1])
    scene = syn.make_scene()
    src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=128)
    if mode == "source":
        tgt, nrm = syn.sample_map(scene, 40_000)
        ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
        np.testing.assert_allclose(r["T"], ref["transformation"], atol=1e-9)
        assert int(r["iterations"]) == ref["iterations"] and bool(r["converged"]) == ref["converged"]
        assert abs(float(r["fitness"]) - ref["fitness"]) < 1e-12 and abs(float(r["rmse"]) - ref["inlier_rmse"]) < 1e-9
    else:
        # joint problem over both submaps: single-process reference sums both records itself
        maps = [syn.sample_map(scene, 40_000, seed=syn.SEED_MAP + k) for k in range(2)]
        trees = [oracle.KDTree(m[0]) for m in maps]
        st = _HostIcpState(np.eye(4), 30, 1e-6, 1e-6)
        while not st.done:
            rec = sum(_record(oracle, trees[k], src, maps[k][0], maps[k][1], st.T, 1.0) for k in range(2))
            st.step(rec, 2 * len(src), oracle)
        np.testing.assert_allclose(r["T"], st.T, atol=1e-9)
        assert int(r["iterations"]) == st.iterations
        dt, dr = syn.se3_error(r["T"], syn.ground_truth_pose())
        assert dt < 5e-3 and dr < 1e-3


# ---- Partitioning B ("union"): ONE map split over the ranks, the per-query argmin decided by a MIN all-reduce of 64-bit keys
def _union_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as oracle

    oracle.lib().orc_set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = syn.make_scene()
    src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=128)
    tgt_all, nrm_all = syn.sample_map(scene, 40_000)
    mine = np.flatnonzero((tgt_all[:, 0] < 0.0) == (rank == 0))  # two spatial shards: x < 0 and x >= 0
    tgt, nrm = tgt_all[mine], nrm_all[mine]
    tree = oracle.KDTree(tgt)
    st = _HostIcpState(np.eye(4), 30, 1e-6, 1e-6)
    rec_t = torch.zeros(32, dtype=torch.float64)
    NO_KEY = np.int64(0x7FFFFFFFFFFFFFFF)

    def accumulate():
        P = src @ st.T[:3, :3].T + st.T[:3, 3]
        corr, d2, _, _, _ = oracle.evaluate(tree, P, 1.0)
        # the key of include/o3ds_backend.h (o3ds_icp_nn_keys): float bits of d2 << 32 | rank << 28 | position in this shard
        bits = np.asarray(d2, dtype=np.float32).view(np.uint32).astype(np.int64)
        keys = np.where(corr >= 0, (bits << 32) | (np.int64(rank) << 28) | corr.astype(np.int64), NO_KEY)
        kt = torch.from_numpy(keys.copy())
        dist.all_reduce(kt, op=dist.ReduceOp.MIN)
        win = kt.numpy()
        won = (win != NO_KEY) & (((win >> 28) & 0xF) == rank)
        c2 = np.where(won, win & 0x0FFFFFFF, -1).astype(corr.dtype)
        rec = np.zeros(32)
        if won.any():
            JTJ, JTr, r2 = oracle.compute_jtj_jtr(P, tgt, nrm, c2)
            rec[:21] = JTJ[np.triu_indices(6)]
            rec[21:27] = JTr
            d = P[won] - tgt[c2[won]]
            rec[27], rec[28], rec[29] = r2, won.sum(), np.einsum("ij,ij->", d, d)
        rec_t.copy_(torch.from_numpy(rec))
        return rec_t

    sharded.run_sharded_loop(accumulate, lambda r: dist.all_reduce(r), lambda r: st.step(r.numpy(), len(src), oracle), lambda: st.done, 30,
                             check_every=1)
    Ts = [torch.zeros(16, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Ts, torch.from_numpy(st.T.ravel().copy()))
    if rank == 0:
        np.savez(out_path, T=st.T, fitness=st.fitness, rmse=st.rmse, iterations=st.iterations, converged=st.converged,
                 all_T=np.stack([t.numpy() for t in Ts]))
    dist.destroy_process_group()


def test_world2_union_equals_registration_against_the_whole_map(tmp_path, oracle):
    """the key exchange reproduces the reference's registration against ONE cloud (Mapper.cpp:141): pose, fitness, rmse and iteration
    count of the two-shard run equal the oracle's ICP against the concatenated map"""
    out = str(tmp_path / "union.npz")
    mp.spawn(_union_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    np.testing.assert_array_equal(r["all_T"][0], r["all_T"][1])
    scene = syn.make_scene()
    src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=128)
    tgt, nrm = syn.sample_map(scene, 40_000)
    ref = oracle.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
    np.testing.assert_allclose(r["T"], ref["transformation"], atol=1e-9)
    assert int(r["iterations"]) == ref["iterations"] and bool(r["converged"]) == ref["converged"]
    assert abs(float(r["fitness"]) - ref["fitness"]) < 1e-12 and abs(float(r["rmse"]) - ref["inlier_rmse"]) < 1e-9


def test_fused_loop_control_flow():
    """run_sharded_fused_loop: max_iteration + 1 passes, one all-reduce per pass on the buffer that pass returned, the device's
    `done` polled every check_every passes only, and never after the last pass."""
    from open3d_slam_amd.sharded import run_sharded_fused_loop

    log = []

    def issue(p):
        log.append(("pass", p))
        return ("buf", p % 3)

    def allred(b):
        log.append(("allreduce", b))

    polls = []

    def done():
        polls.append(len([e for e in log if e[0] == "pass"]))
        return False

    assert run_sharded_fused_loop(issue, allred, done, max_iteration=6, check_every=3) == 7
    assert [e for e in log if e[0] == "pass"] == [("pass", p) for p in range(7)]
    assert [e[1] for e in log if e[0] == "allreduce"] == [("buf", p % 3) for p in range(7)]
    assert polls == [3, 6]
    stop_at = iter([False, True])
    log.clear()
    assert run_sharded_fused_loop(issue, allred, lambda: next(stop_at), max_iteration=50, check_every=2) == 4  # stops at the second poll


# ---- one dense voxel map over two ranks: the exchange step of the fusion (sharded.exchange_by_owner) -------------------------------
def _fusion_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = syn.make_scene()
    voxel = 0.1
    means, counts = {}, {}
    received = 0
    for ins in range(3):  # three insertions; every rank contributes its own share of each (different sizes: ragged all-to-all)
        pts, nrm = syn.sample_map(scene, 20_000 + 5_000 * rank + 1_000 * ins, seed=100 + 10 * ins + rank)
        p, n = sharded.exchange_by_owner(pts, nrm, voxel, None, None)
        assert np.all(sharded.voxel_owner(p, voxel, world) == rank)  # only voxels this rank owns arrive here
        received += len(p)
        means.setdefault("p", []).append(p)
        means.setdefault("n", []).append(n)
    allp, alln = np.vstack(means["p"]), np.vstack(means["n"])
    fp, fn, fc = oracle.dense_fuse(allp, alln, voxel)  # the local fusion, played by the oracle (no GPU here)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p=fp, n=fn, c=fc, received=received)
    empty_p, empty_n = sharded.exchange_by_owner(np.zeros((0, 3)), None, voxel, None, None)  # a rank with nothing to send still takes part
    assert empty_p.shape == (0, 3) and empty_n is None
    # the map has normals, but rank 1's share of this scan is empty and it has no normals array to pass; rank 0's share has a NaN return:
    # six columns travel from everybody (no mismatched row sizes), the NaN row goes nowhere
    if rank == 0:
        pts, nrm = syn.sample_map(scene, 3_000, seed=777)
        pts[5] = np.nan
        p, n = sharded.exchange_by_owner(pts, nrm, voxel, None, None, has_normals=True)
    else:
        p, n = sharded.exchange_by_owner(np.zeros((0, 3)), None, voxel, None, None, has_normals=True)
    assert n is not None and n.shape == p.shape and np.isfinite(p).all()
    tot = torch.tensor([len(p)])
    dist.all_reduce(tot)
    assert int(tot.item()) == 2_999
    dist.destroy_process_group()


def test_voxel_owner_is_a_function_of_the_voxel():
    rng = np.random.default_rng(0)
    p = rng.uniform(-30, 30, size=(5000, 3))
    for world in (1, 2, 3, 8):
        o = sharded.voxel_owner(p, 0.1, world)
        assert o.min() >= 0 and o.max() < world
        jitter = p + rng.uniform(0.0, 1e-9, size=p.shape) * (np.floor((p + 1e-9) * 10) == np.floor(p * 10))  # stays inside the voxel
        np.testing.assert_array_equal(sharded.voxel_owner(jitter, 0.1, world), o)
    # the owner is decided on the value the device will store: 0.7 is voxel 7 as a double and voxel 6 once rounded to float32
    edge = np.array([[0.7, 0.05, 0.05]])
    k64 = np.floor(edge * 10.0)[0, 0]
    k32 = np.floor(edge.astype(np.float32).astype(np.float64) * 10.0)[0, 0]
    assert (k64, k32) == (7.0, 6.0)
    assert sharded.voxel_owner(edge, 0.1, 1 << 20, "f64")[0] != sharded.voxel_owner(edge, 0.1, 1 << 20, "f32")[0]
    counts = np.bincount(sharded.voxel_owner(p, 0.1, 8), minlength=8)
    assert counts.min() > 0.6 * counts.mean()  # the reference's hash spreads a room's voxels over 8 owners reasonably evenly


def test_world2_gloo_dense_map_fusion_equals_single_map(tmp_path, oracle):
    """Two ranks, three ragged insertions each: after the exchange every voxel lives on exactly one rank, and the union of the per-rank
    fusions is the fusion of all points on one rank -- same voxel set, same counts, same means."""
    mp.spawn(_fusion_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(str(tmp_path / f"rank{r}.npz")) for r in range(2)]
    scene = syn.make_scene()
    clouds = [syn.sample_map(scene, 20_000 + 5_000 * r + 1_000 * ins, seed=100 + 10 * ins + r) for ins in range(3) for r in range(2)]
    allp, alln = np.vstack([c[0] for c in clouds]), np.vstack([c[1] for c in clouds])
    assert sum(int(p["received"]) for p in parts) == len(allp)  # nothing lost, nothing duplicated
    rp, rn, rc = oracle.dense_fuse(allp, alln, 0.1)
    key = lambda a: np.floor(a * 10.0 + 1e-9 * 0).astype(np.int64)  # means lie inside their voxels
    got_p, got_n, got_c = np.vstack([p["p"] for p in parts]), np.vstack([p["n"] for p in parts]), np.concatenate([p["c"] for p in parts])
    assert len(got_p) == len(rp) and int(got_c.sum()) == int(rc.sum()) == len(allp)
    from scipy.spatial import cKDTree

    d, j = cKDTree(rp).query(got_p)
    assert len(np.unique(j)) == len(rp) and d.max() < 1e-9  # a bijection between the sharded voxels and the single map's
    np.testing.assert_array_equal(got_c, rc[j])
    np.testing.assert_allclose(got_n, rn[j], atol=1e-9)
    # and no voxel on both ranks
    k0, k1 = set(map(tuple, key(parts[0]["p"]))), set(map(tuple, key(parts[1]["p"])))
    assert not (k0 & k1)
