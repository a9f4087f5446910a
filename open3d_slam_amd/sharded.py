"""Multi-GPU scan-to-map registration: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI)
carrying ONE small all-reduce per ICP iteration -- the only exchange step the path has.

Per iteration the default ("fused") form enqueues ONE kernel (o3ds_icp_pass: the previous iteration's solve/update in its
prologue, this rank's correspondence pass, exact hi/lo sums of the normal equations) and one 4-KB all-reduce; the classic
form (O3DS_SHARDED_FORM=classic: o3ds_icp_accumulate / all-reduce of 32 doubles / o3ds_icp_update) is three kernels plus the
collective.  The sums of the fused form are exact, so its result does not depend on the number of ranks.

Two partitionings, both expressed through the step-wise C-ABI:

* "source"  : every rank holds the same target (+index); rank r accumulates source points
              [r*n/W, (r+1)*n/W).  Sum of the records == the single-GPU record, so the result is the
              single-GPU registration (up to fp64 reassociation of the W-way sum).
* "submap"  : every rank holds ITS OWN target submap and the whole source; the summed record is the joint
              point-to-plane problem over all submaps (north_star: "RCCL all-reduce of the per-submap 6x6
              normal equations").  Fitness is the mean per-submap fitness (n_src_total = W * n).

All ranks apply the identical update to identical state, so no broadcast is needed.
The reference has no multi-device code at all; this is a new design, not a translation.
"""
from __future__ import annotations

from typing import Callable


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced split of n source points: returns (first, count)."""
    base, rem = divmod(n, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def run_sharded_loop(accumulate: Callable[[], object], all_reduce: Callable[[object], None], update: Callable[[object], None],
                     is_done: Callable[[], bool], max_iteration: int, check_every: int = 4) -> int:
    """[O3D] RegistrationICP loop with the reduction distributed: max_iteration updates need max_iteration+1
    correspondence passes; the device decides termination (convergence test inside update), the host only polls
    `is_done` every `check_every` passes so that no per-iteration host sync is needed.  Returns passes issued."""
    passes = 0
    total = max_iteration + 1
    while passes < total:
        rec = accumulate()
        all_reduce(rec)
        update(rec)
        passes += 1
        if passes < total and passes % check_every == 0 and is_done():
            break
    return passes


def run_sharded_fused_loop(issue_pass: Callable[[int], object], all_reduce: Callable[[object], None], is_done: Callable[[], bool],
                           max_iteration: int, check_every: int = 4) -> int:
    """The same loop for the fused step-wise form: `issue_pass(p)` enqueues launch p (update from pass p-1 + pass p) and returns the
    buffer that holds this rank's sums of pass p; the caller-side all-reduce makes them global before launch p+1 folds them."""
    passes = 0
    total = max_iteration + 1
    while passes < total:
        all_reduce(issue_pass(passes))
        passes += 1
        if passes < total and passes % check_every == 0 and is_done():
            break
    return passes


class ShardedIcp:
    """GPU driver of run_sharded_loop over a Backend handle and a torch.distributed process group."""

    def __init__(self, be, mode: str = "source", group=None):
        import torch
        import torch.distributed as dist

        assert mode in ("source", "submap")
        self.be, self.mode, self.group = be, mode, group
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev = torch.device(f"cuda:{be.device_id}")
        # a dedicated torch stream shared by the backend's kernels and the collective: accumulate -> all_reduce ->
        # update are stream-ordered, no host sync per iteration (torch's default stream has handle 0 == "NULL = own
        # stream" in o3ds_set_stream, hence an explicit side stream)
        self.tstream = torch.cuda.Stream(device=dev)
        import os

        self.fused = os.environ.get("O3DS_SHARDED_FORM", "fused") != "classic"
        with torch.cuda.stream(self.tstream):
            self.rec = torch.zeros(32, dtype=torch.float64, device=dev)
            self.sums = [torch.zeros(be.ICP_SUMS_DOUBLES, dtype=torch.float64, device=dev) for _ in range(3)]
        self.tstream.synchronize()
        be.set_stream(self.tstream.cuda_stream)

    def register(self, source: int, target: int, n_src: int, max_corr: float, init=None, max_iter: int = 30,
                 rel_fitness: float = 1e-6, rel_rmse: float = 1e-6, target_crop=None, check_every: int = 4) -> dict:
        be, dist = self.be, self.dist
        if self.mode == "source":
            first, count = shard_range(n_src, self.rank, self.world)
            n_total = n_src
        else:
            first, count = 0, n_src
            n_total = n_src * self.world
        be.icp_begin(source, target, max_corr, init=init, max_iter=max_iter, rel_fitness=rel_fitness, rel_rmse=rel_rmse,
                     target_crop=target_crop)
        ptr = self.rec.data_ptr()

        def accumulate():
            be.icp_accumulate(first, count, ptr)
            return self.rec

        def all_reduce(rec):
            if self.world > 1:
                dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.group)

        def update(rec):
            be.icp_update(ptr, n_total)

        def issue_pass(p: int):
            out, nxt, prev = self.sums[p % 3], self.sums[(p + 1) % 3], self.sums[(p + 2) % 3]
            be.icp_pass(first, count, n_total, prev.data_ptr() if p > 0 else None, out.data_ptr(), nxt.data_ptr())
            return out

        with self.torch.cuda.stream(self.tstream):
            if not self.fused:
                run_sharded_loop(accumulate, all_reduce, update, be.icp_done, max_iter, check_every)
                return be.icp_finish()
            for t in self.sums:
                t.zero_()
            passes = run_sharded_fused_loop(issue_pass, all_reduce, be.icp_done, max_iter, check_every)
            last = self.sums[(passes - 1) % 3]
            return be.icp_pass_finish(n_total, last.data_ptr(), self.sums[passes % 3].data_ptr())


# ---------------------------------------------------------------------------------------------------------------------------------
# Multi-GPU fusion of ONE dense voxel map (BASELINE configs[4]; SURVEY.md 8e "map fusion across GPUs").  The path has one exchange
# step: a voxel's running sums must live on one rank, so every insertion routes each point to the owner of its voxel -- one
# all-to-all of the scan's points (a few MB; direct, not a ring: xGMI is point-to-point) -- and the owner fuses what it receives
# into its local table.  No other collective: the union of the per-rank tables IS the map, a voxel never straddles ranks.


def voxel_owner(points, voxel: float, world: int, storage="f64"):
    """Rank that owns the voxel of every point: the voxel index is the reference's floor(p * (1 / voxel)) (VoxelHashMap.hpp:47-50),
    the owner its hash x + 17191 y + 17191^2 z (VoxelHashMap.hpp:25-35, as a 32-bit unsigned) modulo the world size.  `storage`
    ("f32" / "f64") is the precision the fusing backend stores points in: the owner has to be decided on the value the device will
    bin, or a point that f32 rounding moves across a voxel face would found the same voxel on two ranks."""
    import numpy as np

    p = np.asarray(points, dtype=np.float64)
    if storage == "f32":
        p = p.astype(np.float32).astype(np.float64)
    k = np.floor(p * (1.0 / voxel)).astype(np.int64)
    h = (k[:, 0] + 17191 * k[:, 1] + 17191 * 17191 * k[:, 2]) & 0xFFFFFFFF
    return (h % world).astype(np.int64)


def exchange_by_owner(points, normals, voxel: float, group=None, device=None, storage="f64"):
    """One insertion's exchange step: returns the rows (points, normals or None) whose voxels this rank owns, gathered from all
    ranks.  Counts go first (all_to_all of world ints), then one all_to_all of the rows, sorted by destination.  `device`: where the
    collective's tensors live (None = CPU for gloo; the rank's GPU for nccl / RCCL)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    cols = 3 if normals is None else 6
    rows = points if normals is None else np.hstack([points, np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)])
    if world == 1:
        return points, normals
    owner = voxel_owner(points, voxel, world, storage)
    order = np.argsort(owner, kind="stable")  # rows grouped by destination, original order kept inside a group
    send_counts = np.bincount(owner, minlength=world).astype(np.int64)
    t_send_counts = torch.from_numpy(send_counts).to(device) if device is not None else torch.from_numpy(send_counts)
    t_recv_counts = torch.empty_like(t_send_counts)
    dist.all_to_all_single(t_recv_counts, t_send_counts, group=group)
    recv_counts = t_recv_counts.cpu().numpy()
    t_send = torch.from_numpy(np.ascontiguousarray(rows[order]))
    if device is not None:
        t_send = t_send.to(device)
    t_recv = torch.empty((int(recv_counts.sum()), cols), dtype=torch.float64, device=t_send.device)
    dist.all_to_all_single(t_recv, t_send, output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts],
                           group=group)
    got = t_recv.cpu().numpy()
    return got[:, :3].copy(), (None if normals is None else got[:, 3:].copy())


class ShardedDenseMap:
    """One VoxelizedPointCloud (Voxel.hpp:59-76) spread over the ranks of a process group by voxel owner.  insert() takes THIS rank's
    share of a scan (e.g. the sensors attached to this GPU) already placed in the map frame, exchanges, and fuses the received rows
    into the local device table (o3ds_dense_map_insert); size() is the global voxel count.  The exchange goes through torch tensors
    on `device` (plumbing); the fusion is the backend's.  Host-staged: the rows come from / go to numpy on either side of the
    collective -- an o3ds entry point that exports / imports device rows would remove two PCIe hops per insertion (not built)."""

    def __init__(self, be, voxel: float, group=None, device=None):
        self.be, self.voxel, self.group, self.device = be, float(voxel), group, device
        self.storage = "f64" if getattr(be, "precision", 0) == 1 else "f32"  # backend.PRECISION_F64 == 1
        self.dm = be.dense_map_create(self.voxel)

    def insert(self, points, normals=None):
        p, n = exchange_by_owner(points, normals, self.voxel, self.group, self.device, self.storage)
        if len(p):
            c = self.be.upload(p, n)
            self.be.dense_map_insert(self.dm, c)
            self.be.free(c)
        return len(p)

    def local_size(self) -> int:
        return self.be.dense_map_size(self.dm)

    def size(self) -> int:
        import torch
        import torch.distributed as dist

        t = torch.tensor([self.local_size()], dtype=torch.int64, device=self.device)
        if dist.is_initialized():
            dist.all_reduce(t, group=self.group)
        return int(t.item())

    def close(self):
        self.be.dense_map_free(self.dm)
