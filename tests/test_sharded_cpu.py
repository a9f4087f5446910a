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
