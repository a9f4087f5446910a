// sharded.hpp -- registrations sharded over the GPUs of a node, inside the library: kernels and RCCL collectives queued on the handle's
// stream for all passes of a registration (o3ds_backend.h "sharded over the GPUs of a node").  Included at the end of backend.hip: it
// uses the step-wise entry points defined there (o3ds_icp_begin / _pass / _pass_finish / _nn_keys / _accumulate_keys / _update) -- the
// same kernels open3d_slam_amd/sharded.py drives from Python with torch.distributed collectives in between.
//
// RCCL is reached through dlopen: librccl.so is not a link-time dependency of a library most callers use on one GPU, and the
// communicator is created by the copy that was loaded (a process that has imported torch already holds torch's own librccl.so; a
// communicator must not cross copies).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only: no symbol of it is linked

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // O3DS_RCCL_LIB names a file (memory sizing aside, the one other environment variable the shipped library reads: WHICH copy of
    // RCCL a multi-GPU process uses is a deployment decision); otherwise the soname the ROCm packages install
    const char* names[] = {getenv("O3DS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      api.error = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "not found");
      return;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) api.error = "librccl.so lacks an expected symbol";
  });
  return api;
}

int rccl_fail(o3ds_handle h, const char* what, ncclResult_t r) {
  RcclApi& api = rccl();
  return fail(h, O3DS_ERR_HIP, std::string(what) + ": " + (api.GetErrorString ? api.GetErrorString(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
}
#define RCCL_TRY(what, call)                                \
  do {                                                      \
    const ncclResult_t _r = (call);                         \
    if (_r != ncclSuccess) return rccl_fail(h, what, _r);   \
  } while (0)

// element-wise all-reduce in place on the handle's stream; a handle without a communicator is a group of one: nothing to do
int comm_all_reduce(o3ds_handle h, void* buf, size_t count, ncclDataType_t type, ncclRedOp_t op) {
  if (!h->nccl_comm) return O3DS_OK;
  RCCL_TRY("ncclAllReduce", rccl().AllReduce(buf, buf, count, type, op, (ncclComm_t)h->nccl_comm, h->stream));
  return O3DS_OK;
}

}  // namespace

extern "C" {

int o3ds_comm_unique_id(o3ds_handle h, unsigned char id[128]) {
  CHECK_HANDLE(h);
  static_assert(sizeof(ncclUniqueId) == 128, "the id the header promises");
  if (!id) return fail(h, O3DS_ERR_INVALID_ARG, "comm_unique_id: null id");
  RcclApi& api = rccl();
  if (!api.error.empty()) return fail(h, O3DS_ERR_HIP, api.error);
  ncclUniqueId u;
  RCCL_TRY("ncclGetUniqueId", api.GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return O3DS_OK;
}

int o3ds_comm_destroy(o3ds_handle h) {
  CHECK_HANDLE(h);
  if (h->nccl_comm && h->nccl_owned) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)rccl().CommDestroy((ncclComm_t)h->nccl_comm);
  }
  h->nccl_comm = nullptr;
  h->nccl_owned = false;
  h->comm_rank = 0;
  h->comm_world = 1;
  return O3DS_OK;
}

int o3ds_comm_init(o3ds_handle h, const unsigned char id[128], int rank, int world) {
  CHECK_HANDLE(h);
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(h, O3DS_ERR_INVALID_ARG, "comm_init: bad argument");
  RcclApi& api = rccl();
  if (!api.error.empty()) return fail(h, O3DS_ERR_HIP, api.error);
  (void)o3ds_comm_destroy(h);
  HIP_TRY(hipSetDevice(h->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  RCCL_TRY("ncclCommInitRank", api.CommInitRank(&c, world, u, rank));
  h->nccl_comm = c;
  h->nccl_owned = true;
  h->comm_rank = rank;
  h->comm_world = world;
  return O3DS_OK;
}

int o3ds_comm_attach(o3ds_handle h, void* nccl_comm, int rank, int world) {
  CHECK_HANDLE(h);
  if (!nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(h, O3DS_ERR_INVALID_ARG, "comm_attach: bad argument");
  RcclApi& api = rccl();
  if (!api.error.empty()) return fail(h, O3DS_ERR_HIP, api.error);
  (void)o3ds_comm_destroy(h);
  h->nccl_comm = nccl_comm;
  h->nccl_owned = false;
  h->comm_rank = rank;
  h->comm_world = world;
  return O3DS_OK;
}

int o3ds_icp_register_sharded(o3ds_handle h, int partitioning, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop,
                              const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!init || !params || !out) return fail(h, O3DS_ERR_INVALID_ARG, "icp_register_sharded: null argument");
  if (partitioning != O3DS_SHARD_SOURCE && partitioning != O3DS_SHARD_SUBMAP && partitioning != O3DS_SHARD_UNION)
    return fail(h, O3DS_ERR_INVALID_ARG, "icp_register_sharded: unknown partitioning");
  const int world = h->nccl_comm ? h->comm_world : 1, rank = h->nccl_comm ? h->comm_rank : 0;
  if (partitioning == O3DS_SHARD_UNION && world > 16) return fail(h, O3DS_ERR_INVALID_ARG, "icp_register_sharded: the union form carries the rank in 4 bits (<= 16 ranks)");
  HIP_TRY(hipSetDevice(h->device));
  // the exact source size (ranges are addressed by it), then a session as o3ds_icp_begin opens it
  CloudRec* src = find_cloud(h, source);
  if (!src) return fail(h, O3DS_ERR_INVALID_ARG, "icp_register_sharded: unknown source cloud");
  const size_t n = src->n;
  int rc = begin_session(h, source, target, target_crop, init, params);
  if (rc) return rc;
  const int total = params->max_iteration + 1;  // max_iteration updates need max_iteration + 1 correspondence passes
  // every rank holds the identical state after every pass (identical sums, identical arithmetic), so the few looks at `done` below
  // come out the same everywhere and the ranks issue the same collectives
  auto loop_done = [&](int passes, int* done) -> int {
    *done = 0;
    if (passes >= total || passes % 8 != 0) return O3DS_OK;
    return o3ds_icp_done(h, done);
  };
  if (partitioning == O3DS_SHARD_UNION) {
    if (h->shard_keys_cap < n) {
      if (h->d_shard_keys) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_shard_keys);
        h->d_shard_keys = nullptr;
        h->shard_keys_cap = 0;
      }
      const size_t cap = std::max<size_t>(n + n / 4, 1 << 16);
      if (hipMalloc((void**)&h->d_shard_keys, sizeof(unsigned long long) * cap) != hipSuccess) return fail(h, O3DS_ERR_OOM, "icp_register_sharded: key buffer allocation failed");
      h->shard_keys_cap = cap;
    }
    if (!h->d_shard_sums && hipMalloc((void**)&h->d_shard_sums, sizeof(double) * 3 * O3DS_ICP_SUMS_DOUBLES) != hipSuccess)
      return fail(h, O3DS_ERR_OOM, "icp_register_sharded: record buffer allocation failed");
    double* rec = h->d_shard_sums;  // (32 of its doubles)
    for (int p = 0; p < total; ++p) {
      rc = o3ds_icp_nn_keys(h, 0, n, rank, h->d_shard_keys);
      if (!rc) rc = comm_all_reduce(h, h->d_shard_keys, n, ncclUint64, ncclMin);
      if (!rc) rc = o3ds_icp_accumulate_keys(h, 0, n, rank, h->d_shard_keys, rec);
      if (!rc) rc = comm_all_reduce(h, rec, kRec, ncclDouble, ncclSum);
      if (!rc) rc = o3ds_icp_update(h, rec, (uint64_t)n);
      int done = 0;
      if (!rc) rc = loop_done(p + 1, &done);
      if (rc) return rc;
      if (done) break;
    }
    return o3ds_icp_finish(h, out);
  }
  size_t first = 0, count = n;
  if (partitioning == O3DS_SHARD_SOURCE) {  // contiguous, balanced split (open3d_slam_amd/sharded.py shard_range)
    const size_t base = n / (size_t)world, rem = n % (size_t)world;
    first = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
    count = base + ((size_t)rank < rem ? 1 : 0);
  }
  const uint64_t n_total = partitioning == O3DS_SHARD_SOURCE ? (uint64_t)n : (uint64_t)n * (uint64_t)world;
  // the fused pass serves at most kFusedMaxQueries points per call: larger shards take the accumulate / update triple, whose pass kernel
  // loops -- on EVERY rank (the two forms issue different collectives: the decision is taken on the largest share)
  const size_t largest = partitioning == O3DS_SHARD_SOURCE ? (n + (size_t)world - 1) / (size_t)world : n;
  if (!h->d_shard_sums && hipMalloc((void**)&h->d_shard_sums, sizeof(double) * 3 * O3DS_ICP_SUMS_DOUBLES) != hipSuccess)
    return fail(h, O3DS_ERR_OOM, "icp_register_sharded: record buffer allocation failed");
  if (!h->fused || largest > kFusedMaxQueries) {
    double* rec = h->d_shard_sums;
    for (int p = 0; p < total; ++p) {
      rc = o3ds_icp_accumulate(h, first, count, rec);
      if (!rc) rc = comm_all_reduce(h, rec, kRec, ncclDouble, ncclSum);
      if (!rc) rc = o3ds_icp_update(h, rec, n_total);
      int done = 0;
      if (!rc) rc = loop_done(p + 1, &done);
      if (rc) return rc;
      if (done) break;
    }
    return o3ds_icp_finish(h, out);
  }
  double* sums[3] = {h->d_shard_sums, h->d_shard_sums + O3DS_ICP_SUMS_DOUBLES, h->d_shard_sums + 2 * O3DS_ICP_SUMS_DOUBLES};
  HIP_TRY(hipMemsetAsync(h->d_shard_sums, 0, sizeof(double) * 3 * O3DS_ICP_SUMS_DOUBLES, h->stream));
  int passes = 0;
  while (passes < total) {
    // launch `passes`: the update from the all-reduced sums of the previous pass in its prologue, this rank's correspondence pass, its
    // exact hi / lo sums added to sums[passes % 3]; then the one collective of the iteration, in place, behind it on the same stream
    double* o = sums[passes % 3];
    rc = o3ds_icp_pass(h, first, count, (size_t)n_total, passes > 0 ? sums[(passes + 2) % 3] : nullptr, o, sums[(passes + 1) % 3]);
    if (!rc) rc = comm_all_reduce(h, o, O3DS_ICP_SUMS_DOUBLES, ncclDouble, ncclSum);
    ++passes;
    int done = 0;
    if (!rc) rc = loop_done(passes, &done);
    if (rc) return rc;
    if (done) break;
  }
  return o3ds_icp_pass_finish(h, (size_t)n_total, sums[(passes - 1) % 3], sums[passes % 3], out);
}

}  // extern "C"
