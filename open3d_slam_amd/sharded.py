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
* "union"   : ONE map split over the ranks (spatial shards); every rank holds the whole source.  Per iteration every rank
              searches its shard and writes a 64-bit key per query (distance bits | rank | position), one element-wise MIN
              all-reduce (n x 8 B) picks each query's match in the union of the shards, the owning rank contributes its rows, and
              the 32-double records are summed (SURVEY.md 8e Partitioning B).  This is the sharding that reproduces the reference's
              registration against a single cloud (Mapper.cpp:141): equal to the one-GPU registration against the whole map.

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


def init_library_comm(be, group=None) -> None:
    """The backend handle's OWN RCCL communicator (o3ds_comm_init) for the ranks of a torch.distributed group: rank 0 draws the
    ncclUniqueId inside the library, torch.distributed only carries its 128 bytes to the other ranks.  After this,
    Backend.icp_register_sharded queues kernels and ncclAllReduce calls itself -- no Python, no torch between the passes."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        be.comm_init(be.comm_unique_id(), 0, 1)
        return
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device(f"cuda:{be.device_id}") if on_gpu else torch.device("cpu")
    t = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        t = torch.frombuffer(bytearray(be.comm_unique_id()), dtype=torch.uint8).to(dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    be.comm_init(bytes(t.cpu().numpy().tobytes()), rank, world)


class LibraryShardedIcp:
    """ShardedIcp's interface over o3ds_icp_register_sharded: the loop, the kernels and the collectives live in libo3ds_backend.so."""

    MODES = {"source": 0, "submap": 1, "union": 2}

    def __init__(self, be, mode: str = "source", group=None):
        assert mode in self.MODES
        self.be, self.mode = be, mode
        init_library_comm(be, group)

    def register(self, source: int, target: int, n_src: int, max_corr: float, init=None, max_iter: int = 30, rel_fitness: float = 1e-6,
                 rel_rmse: float = 1e-6, target_crop=None, check_every: int = 4, method=None) -> dict:
        kw = {} if method is None else {"method": method}
        return self.be.icp_register_sharded(self.MODES[self.mode], source, target, max_corr, init=init, max_iter=max_iter,
                                            rel_fitness=rel_fitness, rel_rmse=rel_rmse, target_crop=target_crop, **kw)


class ShardedIcp:
    """GPU driver of run_sharded_loop over a Backend handle and a torch.distributed process group."""

    def __init__(self, be, mode: str = "source", group=None, always_collective: bool = False):
        import torch
        import torch.distributed as dist

        assert mode in ("source", "submap", "union")
        self.be, self.mode, self.group = be, mode, group
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # a group of one rank skips its collectives; `always_collective` issues them anyway (they are the identity), so that the
        # RCCL launch path and its ordering with the backend's kernels on the shared stream run on a one-GPU box
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
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
            self.keys = None  # "union": one int64 key per source point, grown on demand
        self.tstream.synchronize()
        be.set_stream(self.tstream.cuda_stream)

    def register(self, source: int, target: int, n_src: int, max_corr: float, init=None, max_iter: int = 30,
                 rel_fitness: float = 1e-6, rel_rmse: float = 1e-6, target_crop=None, check_every: int = 4) -> dict:
        be, dist = self.be, self.dist
        if self.mode == "union":
            return self._register_union(source, target, n_src, max_corr, init, max_iter, rel_fitness, rel_rmse, target_crop, check_every)
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
            if self.collective:
                dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.group)

        def update(rec):
            be.icp_update(ptr, n_total)

        def issue_pass(p: int):
            out, nxt, prev = self.sums[p % 3], self.sums[(p + 1) % 3], self.sums[(p + 2) % 3]
            be.icp_pass(first, count, n_total, prev.data_ptr() if p > 0 else None, out.data_ptr(), nxt.data_ptr())
            return out

        # the fused pass serves at most ICP_PASS_MAX_QUERIES points per call (o3ds_backend.h): larger shards take the classic triple,
        # whose pass kernel loops -- on EVERY rank (count differs by at most one between ranks of a "source" split, but the decision must
        # be the same everywhere: the two forms issue different collectives)
        fused = self.fused and -(-n_src // (self.world if self.mode == "source" else 1)) <= be.ICP_PASS_MAX_QUERIES
        with self.torch.cuda.stream(self.tstream):
            if not fused:
                run_sharded_loop(accumulate, all_reduce, update, be.icp_done, max_iter, check_every)
                return be.icp_finish()
            for t in self.sums:
                t.zero_()
            passes = run_sharded_fused_loop(issue_pass, all_reduce, be.icp_done, max_iter, check_every)
            last = self.sums[(passes - 1) % 3]
            return be.icp_pass_finish(n_total, last.data_ptr(), self.sums[passes % 3].data_ptr())


    def _register_union(self, source, target, n_src, max_corr, init, max_iter, rel_fitness, rel_rmse, target_crop, check_every):
        be, dist, torch = self.be, self.dist, self.torch
        assert self.world <= 16, "the key carries the rank in 4 bits"
        with torch.cuda.stream(self.tstream):
            if self.keys is None or self.keys.numel() < n_src:
                self.keys = torch.empty(n_src, dtype=torch.int64, device=self.rec.device)
            keys = self.keys[:n_src]
            be.icp_begin(source, target, max_corr, init=init, max_iter=max_iter, rel_fitness=rel_fitness, rel_rmse=rel_rmse, target_crop=target_crop)
            kptr, rptr = keys.data_ptr(), self.rec.data_ptr()

            def accumulate():
                be.icp_nn_keys(0, n_src, self.rank, kptr)
                if self.collective:
                    dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=self.group)
                be.icp_accumulate_keys(0, n_src, self.rank, kptr, rptr)
                return self.rec

            def all_reduce(rec):
                if self.collective:
                    dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.group)

            run_sharded_loop(accumulate, all_reduce, lambda rec: be.icp_update(rptr, n_src), be.icp_done, max_iter, check_every)
            return be.icp_finish()


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
    assert len(k) == 0 or np.abs(k).max() < (1 << 20), "voxel indices beyond 2^20: the device table packs 21 bits per axis"
    h = (k[:, 0] + 17191 * k[:, 1] + 17191 * 17191 * k[:, 2]) & 0xFFFFFFFF
    return (h % world).astype(np.int64)


def exchange_by_owner(points, normals, voxel: float, group=None, device=None, storage="f64", has_normals=None):
    """One insertion's exchange step, HOST-STAGED (numpy rows; the CPU / gloo form that tests/test_sharded_cpu.py drives with the oracle
    as the local fusion -- ShardedDenseMap on a GPU uses the device form below): returns the rows (points, normals or None) whose voxels
    this rank owns, gathered from all ranks.  Counts go first (all_to_all of world ints), then one all_to_all of the rows, sorted by
    destination.  Every rank sends SIX columns whatever its own arguments are -- `has_normals` is a property of the map, fixed by the
    caller for all ranks (a rank whose share of a scan is empty has no normals array to infer it from) -- and rows that are not finite
    are dropped before the owner is computed, as the device form does."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if has_normals is None:
        has_normals = normals is not None
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    nrm = np.zeros_like(points) if normals is None else np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
    keep = np.isfinite(points).all(axis=1)
    points, nrm = points[keep], nrm[keep]
    if world == 1:
        return points, (nrm if has_normals else None)
    rows = np.hstack([points, nrm])
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
    t_recv = torch.empty((int(recv_counts.sum()), 6), dtype=torch.float64, device=t_send.device)
    dist.all_to_all_single(t_recv, t_send, output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts],
                           group=group)
    got = t_recv.cpu().numpy()
    return got[:, :3].copy(), (got[:, 3:].copy() if has_normals else None)


class ShardedDenseMap:
    """One VoxelizedPointCloud (Voxel.hpp:59-76) spread over the ranks of a process group by voxel owner.  insert() takes THIS rank's
    share of a scan as a device cloud (or host arrays, uploaded first) and the pose that places it in the map frame.  On the device:
    o3ds_cloud_export_rows_by_owner groups the placed rows by owner into a torch tensor, the group sizes travel first (world ints),
    then ONE all_to_all_single of the rows between the GPUs (direct, not a ring: xGMI is point-to-point), o3ds_cloud_import_rows turns
    what arrived into a cloud and o3ds_dense_map_insert fuses it into the local table.  Nothing goes through host memory except the
    split sizes, which torch.distributed needs as Python ints.  size() is the global voxel count."""

    def __init__(self, be, voxel: float, group=None, has_normals: bool = True, always_collective: bool = False):
        import torch
        import torch.distributed as dist

        self.be, self.voxel, self.group, self.has_normals = be, float(voxel), group, bool(has_normals)
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())  # as in ShardedIcp
        self.device = torch.device(f"cuda:{be.device_id}")
        self.dm = be.dense_map_create(self.voxel)
        # the backend's kernels and the collective share one torch stream, as in ShardedIcp
        self.tstream = torch.cuda.Stream(device=self.device)
        be.set_stream(self.tstream.cuda_stream)

    def insert(self, cloud, T=None, normals=None) -> int:
        """cloud: a device cloud id of this backend, or host points (n x 3) [+ normals].  Returns the number of rows this rank fused."""
        be, torch, dist = self.be, self.torch, self.dist
        own = not isinstance(cloud, int)
        cid = be.upload(cloud, normals) if own else cloud
        n = be.size(cid)[0]
        with torch.cuda.stream(self.tstream):
            rows = torch.empty((max(n, 1), 6), dtype=torch.float64, device=self.device)
            counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
            be.export_rows_by_owner(cid, T, self.voxel, self.world, rows.data_ptr(), counts.data_ptr())
            if self.collective:
                recv_counts = torch.empty_like(counts)
                dist.all_to_all_single(recv_counts, counts, group=self.group)
                send_split = [int(c) for c in counts.cpu()]
                recv_split = [int(c) for c in recv_counts.cpu()]
                got = torch.empty((sum(recv_split), 6), dtype=torch.float64, device=self.device)
                dist.all_to_all_single(got, rows[: sum(send_split)], output_split_sizes=recv_split, input_split_sizes=send_split, group=self.group)
            else:
                got = rows[: int(counts.sum().item())]
            m = int(got.shape[0])
            if m:
                c = be.import_rows(got.data_ptr(), m, self.has_normals)
                be.dense_map_insert(self.dm, c)
                be.free(c)
            self.tstream.synchronize()  # `rows` / `got` go back to torch's allocator when this returns
        if own:
            be.free(cid)
        return m

    def local_size(self) -> int:
        return self.be.dense_map_size(self.dm)

    def size(self) -> int:
        t = self.torch.tensor([self.local_size()], dtype=self.torch.int64, device=self.device)
        if self.dist.is_initialized() and self.world > 1:
            self.dist.all_reduce(t, group=self.group)
        return int(t.item())

    def close(self):
        self.be.dense_map_free(self.dm)
        self.be.set_stream(None)
