// backend.hip -- C-ABI (include/o3ds_backend.h) of the gfx950 scan-matching / map-fusion backend.
// Host-side orchestration only; all arithmetic is in icp_kernels.hpp / cloud_kernels.hpp.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>

#include <rocprim/device/device_radix_sort.hpp>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "cloud_kernels.hpp"
#include "common.hpp"
#include "icp_kernels.hpp"
#include "map_kernels.hpp"
#include "normals_kernel.hpp"

using namespace o3ds;

namespace {

thread_local std::string g_thread_error;

// A/B levers, tuning knobs and debugging aids (O3DS_ICP_MODE, O3DS_ICP_SETS, O3DS_SUM_NO_SPLIT, O3DS_VOXEL_SORT, O3DS_CARVE_SORT, ...) exist
// only in the library built with -DO3DS_AB_SWITCHES (lib/libo3ds_backend_ab.so: what the A/B tests and the experiment scripts load).  The
// shipped library reads three environment variables, two of memory sizing and the file name of RCCL: O3DS_POOL_CAP_MB, O3DS_ARENA_MB, O3DS_RCCL_LIB -- a drop-in library's results
// and code paths must not depend on its host process's environment.
#ifdef O3DS_AB_SWITCHES
inline const char* ab_getenv(const char* name) { return getenv(name); }
#else
inline const char* ab_getenv(const char*) { return nullptr; }
#endif

struct CloudRec {
  size_t n = 0;
  int precision = 0;
  void* pts = nullptr;  // P4[n]
  void* nrm = nullptr;  // P4[n] or null
  void* col = nullptr;  // P4[n] or null: PointCloud::colors_ (carried by the cloud operations, never read by registration)
  size_t cap = 0;       // points the pts / nrm arrays have room for when that is more than n (a map after its merge: the next scan is placed
                        // behind the last point instead of copying the map into larger arrays); 0 = exactly n
  // nearest-neighbour index
  bool has_index = false;
  bool index_byproduct = false;  // the index was left by the normal estimation (its cell size suits THAT search): a registration
                                 // keeps it if the cell is within [r / 8, 0.75 r] of its own radius and rebuilds otherwise
  GridDev grid{};
  int* cell_start = nullptr;
  void* spts = nullptr;
  void* snrm = nullptr;
  // A box that is known to contain every point (not necessarily tight): set where a bounding box has been computed anyway
  // (VoxelDownSample, an index build) and carried to clouds derived from it (subsets, voxel means, rigid placements, unions), so
  // that the next index build of the per-scan pipeline does not pay a reduction kernel + read-back + host sync for it again.
  // layout left by the map merge: [pass-through block | voxel block in ascending key order]; vox_first < 0 = unknown (any other
  // operation that reorders or moves points resets it), see voxel_reduce_t
  long long vox_first = -1;
  size_t vox_count = 0;
  struct PMapRec* pm = nullptr;  // non-null: the cloud is a submap in its PERSISTENT form (map_kernels.hpp): pts / nrm are slot arrays, n an
                                 // upper bound of the slots in use, the index is row-paged and kept up to date by the insertions
  size_t index_positions = 0;    // positions the cell-sorted arrays of the index span (0: n; the row-paged index has gaps)
  // A count the host has not seen yet (common.hpp, CountPub): `n` is then an UPPER bound -- the arrays have room for it, launches are
  // sized by it -- and the exact number is in the device word of record `lazy_slot` once the producing kernel has run (and in the pinned
  // record, stamped `lazy_seq`, for the host: resolve_count).  -1: n is exact.
  int lazy_slot = -1;
  int lazy_seq = 0;
  size_t n_lower = 0;  // ... and a lower bound of it (o3ds_cloud_size_bound: "is it empty" rarely needs the exact number)
  // An ingest that is still in flight on the handle's copy stream (o3ds_cloud_upload_f32): the first use of the cloud on the handle's
  // stream waits for this event (cloud_ready); `ingest_buf` is the ingest buffer the points live in (not a pool block), -1 = none.
  hipEvent_t ingest_ev = nullptr;
  int ingest_buf = -1;
  // ... and the box of the points inside `pre_crop` that the ingest kernel reduced on the way (pinned record `pre_slot`, stamp `pre_seq`)
  int pre_slot = -1;
  int pre_seq = 0;
  CropDev pre_crop{};
  bool has_box = false;
  bool box_padded = false;  // the box already has a margin against the rounding of derived values (never padded twice: a map is
                            // re-voxelised at every insertion and its box must not creep outwards over a long mission)
  double bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};
};

void box_inflate(CloudRec& c) {  // stored values are rounded (f32 storage, f64 -> f32 of means / placements): keep the box conservative
  if (c.box_padded) return;
  c.box_padded = true;
  for (int a = 0; a < 3; ++a) {
    const double pad = 1e-5 * (std::fabs(c.bmn[a]) + std::fabs(c.bmx[a]) + (c.bmx[a] - c.bmn[a])) + 1e-9;
    c.bmn[a] -= pad;
    c.bmx[a] += pad;
  }
}
void box_copy(CloudRec& to, const CloudRec& from) {
  to.has_box = from.has_box;
  to.box_padded = from.box_padded;
  for (int a = 0; a < 3; ++a) to.bmn[a] = from.bmn[a], to.bmx[a] = from.bmx[a];
}
void box_union(CloudRec& to, const CloudRec& a, const CloudRec& b) {  // of two non-empty clouds; an empty one contributes nothing
  if (a.n == 0) return box_copy(to, b);
  if (b.n == 0) return box_copy(to, a);
  to.has_box = a.has_box && b.has_box;
  to.box_padded = a.box_padded && b.box_padded;
  for (int k = 0; k < 3; ++k) to.bmn[k] = std::min(a.bmn[k], b.bmn[k]), to.bmx[k] = std::max(a.bmx[k], b.bmx[k]);
}
void box_transform(CloudRec& to, const CloudRec& from, const double T[16]) {  // box of the 8 placed corners, inflated
  to.has_box = from.has_box;
  to.box_padded = false;  // the placed values are rounded again
  if (!from.has_box) return;
  for (int a = 0; a < 3; ++a) to.bmn[a] = 1e300, to.bmx[a] = -1e300;
  for (int k = 0; k < 8; ++k) {
    const double x = (k & 1) ? from.bmx[0] : from.bmn[0], y = (k & 2) ? from.bmx[1] : from.bmn[1], z = (k & 4) ? from.bmx[2] : from.bmn[2];
    const double w = T[3] * x + T[7] * y + T[11] * z + T[15];
    for (int a = 0; a < 3; ++a) {
      const double v = (T[a] * x + T[4 + a] * y + T[8 + a] * z + T[12 + a]) / w;
      to.bmn[a] = std::min(to.bmn[a], v);
      to.bmx[a] = std::max(to.bmx[a], v);
    }
  }
  for (int a = 0; a < 3; ++a)
    if (!std::isfinite(to.bmn[a]) || !std::isfinite(to.bmx[a])) to.has_box = false;
  if (to.has_box) box_inflate(to);
}

// host side of a submap in its persistent form (map_kernels.hpp)
struct PMapRec {
  PmDev dev{};
  CropDev* d_hist = nullptr;
  int t = 0;              // insertions since the map entered this form (their volumes: d_hist[1..t])
  size_t n_upper = 0;     // slots in use: the host's upper bound (exact value: dev.counters[kPmN])
  size_t live_lower = 0;  // live points: a lower bound
  double clamped = 0;     // slots that entered the index outside its grid, as of the last record
  double multi_total = 0; // entries of the multi list so far (appended to, never compacted), as of the last record
  double pool_top = 0;    // first free position of the index pool: an upper bound (the last record the host saw + the worst case of
                          // every insertion since)
  int rec_slot = -1, rec_seq = 0;  // pinned record the insertions publish {slots, pool top, dead, error} into; stamp of the latest
  size_t rows = 0;
  std::vector<void*> blocks;  // every pool block of the form except the slot arrays and the index arrays (those hang on the CloudRec)
};

constexpr int kMaxPassBlocks = 4096;  // capacity of the partial-record buffer (rows per pass <= pass_rows <= this)
constexpr size_t kMaxCells = (size_t)1 << 27;  // 512 MiB of cell_start at most

}  // namespace

// dense voxel map (VoxelizedPointCloud): a device hash table, see cloud_kernels.hpp
struct DenseRec {
  double voxel = 0.0;
  size_t cap = 0;          // power of two
  o3ds::DenseDev dev{};
  bool has_normals = false;
  bool has_colors = false;
  size_t used_upper = 0;   // upper bound on the OCCUPIED slots = live voxels + tombstones (exact count fetched only when the table
                           // looks too full); the load factor of the open-addressing table is about these, not about live voxels
};

struct o3ds_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int precision = O3DS_PRECISION_F32;
  std::string err, deferred_err;
  std::unordered_map<uint64_t, CloudRec> clouds;
  std::unordered_map<uint64_t, DenseRec> dense_maps;
  uint64_t next_id = 1;
  // ICP scratch
  double* d_partials = nullptr;     // [kMaxPassBlocks][kRec]
  IcpStateDev* d_state = nullptr;
  IcpStateDev* h_state = nullptr;   // pinned, mapped
  char* h_pin = nullptr;            // pinned block the small device -> host read-backs land in (read_back)
  char* h_pin_dev = nullptr;        // the same block as kernels address it: a kernel that produces a size / count / box the host is about
                                    // to wait for stores it here itself (no copy on the stream, the host only synchronises)
  char* h_stage[2] = {nullptr, nullptr};  // pinned ring the large host <-> device copies of caller buffers go through (staged_copy)
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  IcpStateDev* h_state_dev = nullptr;  // the device's view of h_state
  // step-wise ICP session
  bool session = false;
  IcpPassArgs pass{};
  o3ds_icp_params params{};
  size_t session_n_src = 0;
  int session_launches = 0;                     // fused step-wise form: launches issued in this session
  const IcpStateDev* session_state = nullptr;   // ... and the state the latest one wrote (null: d_state)
  int session_precision = 0;
  bool session_crop = false;
  int session_method = O3DS_ICP_POINT_TO_PLANE;
  double gicp_epsilon = 1e-3;  // [O3D] TransformationEstimationForGeneralizedICP default
  hipStream_t own_stream = nullptr;
  // device memory of the handle: a caching allocator over hipMalloc (dev_alloc / dev_free below)
  std::multimap<size_t, void*> pool_free;        // cached blocks by size
  std::unordered_map<void*, size_t> pool_size;   // every block this handle obtained from hipMalloc -> its size
  size_t pool_bytes = 0;   // everything obtained from hipMalloc
  size_t pool_cached = 0;  // of which in the free lists
  // bump arena for the temporaries of one top-level ABI call (stream-ordered reuse: everything runs on one stream)
  std::vector<std::pair<char*, size_t>> arena_blocks;
  size_t arena_cur = 0, arena_off = 0;
  int arena_depth = 0;
  // rows (workgroups) per pass are capped at pass_rows (O3DS_PASS_ROWS, for experiments and the geometry-independence test)
  // ONE launch per pass with the previous pass's tail in its prologue (O3DS_ICP_MODE=fused), see icp_fused_kernel
  // normal estimation: remembered cell size of the ring search (see normals_t)
  double nrm_cell = 0.0, nrm_radius = 0.0;
  int nrm_knn = 0, nrm_age = 0;
  size_t nrm_n = 0;
  bool fused = true;  // O3DS_ICP_MODE=launch selects the two-kernel form (same results bit for bit)
  int* d_nn_cache = nullptr;  // per-query match of the previous pass (bound for the pruned search); grown on demand
  size_t nn_cache_cap = 0;
  // candidate sets of the fused loop (icp_kernels.hpp, Collect): one allocation of nn_cache_cap x (kSetCap ints + {p_ref, L} at f64 width)
  int* d_set_pos = nullptr;
  void* d_set_ref = nullptr;
  bool sets = true;                                      // O3DS_ICP_SETS=0: every pass searches (same results bit for bit)
  float set_gain = 2.0f, set_min = 1e-3f, set_cap = 0.04f;  // O3DS_SET_GAIN / _MIN / _CAP (metres)
  char* d_fused = nullptr;  // [2 states | 3 x kFusedSlots slot records (hi and lo sums)]
  unsigned long long fused_launches = 0;
  unsigned long long fused_seq = 0;  // stamp of the last launch that was asked to write the pinned state (wait_fused_state)
  o3ds_overlap_fn overlap_fn = nullptr;  // o3ds_icp_overlap_next: called once behind the next registration's launches
  void* overlap_arg = nullptr;
  // Scratch that its users leave the way they found it, so that no launch is spent on clearing it: the per-cell counters of an index
  // build (the scatter counts them back down to zero) and the voxel table of VoxelDownSample (vox_mean_kernel empties the slots it
  // used).  `*_clean` is false while an operation is in flight or after one failed: the next user clears the block first.
  int* d_cells = nullptr;
  size_t cells_cap = 0;
  bool cells_clean = false;
  unsigned char* d_voxtab = nullptr;
  o3ds::DrawState* d_draw = nullptr;  // RandomDownSample on the device (cloud_kernels.hpp): histograms zero between calls
  size_t voxtab_cap = 0;  // slots
  bool voxtab_clean = false;
  // ---- values kernels publish for the host without a copy or a wait on the stream: kPinRecs records {count, stamp, box} in pinned
  // memory (h_rec; the device's view h_rec_dev), the counts also in device words (d_cnt) for the kernels that consume a cloud whose size
  // the host has not seen yet.  A record belongs to one cloud at a time (rec_owner: cloud id, 0 = free).
  struct PinRec {
    int cnt, seq;
    double box[6];
    double pad;  // box[6] to the kernels that publish seven values (pm_carve_finish_kernel: the number of carved points)
  };
  PinRec* h_rec = nullptr;
  PinRec* h_rec_dev = nullptr;
  int* d_cnt = nullptr;  // [kPinRecs * 16]: one word per record, 64 bytes apart
  uint64_t rec_owner[64] = {0};
  int rec_next = 0, rec_seq = 0;
  size_t voxel_count_hint = 0;  // the last VoxelDownSample size the host saw: the estimate heuristics use while the current one is in flight
  // ---- ingest (o3ds_cloud_upload_f32): its own stream, so that scan k + 1 crosses PCIe and is unpacked while frame k runs; buffers of
  // its own (not pool blocks: the pool's reuse is ordered by the handle's ONE stream), handed back behind an event on that stream
  struct IngestBuf {
    void* pts = nullptr;
    size_t pts_bytes = 0;
    unsigned long long* box = nullptr;  // device box record (order_bits images), armed
    hipEvent_t freed = nullptr;         // recorded on the handle's stream when the cloud was freed: the next ingest into the buffer waits for it
    bool in_use = false;
  };
  std::vector<IngestBuf> ingest_bufs;
  hipStream_t copy_stream = nullptr;
  unsigned char* d_raw = nullptr;  // the records as they crossed PCIe (copy-stream ordered: one buffer)
  size_t raw_cap = 0;
  std::vector<hipEvent_t> ev_pool;  // disable-timing events for ingest_ev
  bool pre_crop_valid = false;      // the volume of the last crop + VoxelDownSample: what the next ingest reduces its box for
  CropDev pre_crop{};
  // chained scan of vox_order_kernel: tile records (never cleared: they carry the call's number) and the running ticket counter
  unsigned long long* d_tiles = nullptr;
  size_t tiles_cap = 0;
  unsigned int* d_ticket = nullptr;
  unsigned int ticket_base = 0, scan_gen = 0;
  int fused_chunk_hint[2] = {12, 12};  // [registration against a cropped target?]: scan-to-map and scan-to-scan alternate on a handle  // launches queued before the host first looks at the state: what the previous registration needed, plus one
  int debug_update = 0;  // O3DS_DEBUG_UPDATE: timing experiments only
  int pass_rows = 1024;
  // sharded registrations inside the library (sharded.hpp): the RCCL communicator (created by o3ds_comm_init: owned; attached: the
  // caller's), this handle's place in it, and the buffers the collectives run on
  void* nccl_comm = nullptr;
  bool nccl_owned = false;
  int comm_rank = 0, comm_world = 1;
  double* d_shard_sums = nullptr;  // 3 x O3DS_ICP_SUMS_DOUBLES
  unsigned long long* d_shard_keys = nullptr;
  size_t shard_keys_cap = 0;
  // profiling (bench.py roofline): 1 = event pairs around every accumulate launch + tagged spans, 2 = tagged spans only (a span around
  // a whole registration then holds its kernels and nothing else)
  int profiling = 0;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  // ... and tagged spans (o3ds_profile_span): pairs of events around whatever the caller -- or, for the internal tags, a kernel
  // sequence inside a call -- encloses
  std::vector<hipEvent_t> span_ev[16];
  size_t span_used[16] = {0};
};

namespace {

int fail(o3ds_handle h, int code, const std::string& msg) {
  std::string text = msg;
  if (h && !h->deferred_err.empty()) {  // what went wrong inside find_cloud (a size that never arrived, a map that could not be folded)
    text += " [" + h->deferred_err + "]";
    h->deferred_err.clear();
  }
  g_thread_error = text;
  if (h) h->err = text;
  return code;
}

#define HIP_TRY(expr)                                                                                       \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess)                                                                                   \
      return fail(h, _e == hipErrorOutOfMemory ? O3DS_ERR_OOM : O3DS_ERR_HIP,                               \
                  std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

// Every entry point makes the handle's device the calling thread's current device: the current device is per thread and defaults to
// 0, handles are shared across the caller's worker threads (host/o3ds_mapping.hpp) and one process may drive one handle per GPU.
#define CHECK_HANDLE(h)                                                        \
  if (!(h)) return fail(nullptr, O3DS_ERR_BAD_HANDLE, "null handle");          \
  if (hipSetDevice((h)->device) != hipSuccess) return fail((h), O3DS_ERR_HIP, "hipSetDevice failed")

// ---- small read-backs ---------------------------------------------------------------------------------------------
// Sizes, counts and bounding boxes that the host needs go through a pinned block owned by the handle: a D2H copy into pageable
// memory is staged and waited for inside the runtime (tens of microseconds for four bytes -- a dozen of them per lidar frame);
// into pinned memory it is an ordinary stream operation, and one synchronisation serves all items of a call.
constexpr size_t kPinBytes = 64 << 10;
constexpr size_t kPubOff = kPinBytes - 256;  // the last 256 bytes: slots for values kernels publish themselves (pub_slot)
struct D2H {
  void* dst;
  const void* src;
  size_t bytes;
};
int read_back(o3ds_handle h, std::initializer_list<D2H> items) {
  size_t off = 0;
  for (const D2H& it : items) {
    if (off + it.bytes > kPubOff) return fail(h, O3DS_ERR_INVALID_ARG, "read_back: item list exceeds the pinned block");
    HIP_TRY(hipMemcpyAsync(h->h_pin + off, it.src, it.bytes, hipMemcpyDeviceToHost, h->stream));
    off += (it.bytes + 15) & ~(size_t)15;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  off = 0;
  for (const D2H& it : items) {
    memcpy(it.dst, h->h_pin + off, it.bytes);
    off += (it.bytes + 15) & ~(size_t)15;
  }
  return O3DS_OK;
}

// ---- large copies of caller buffers ------------------------------------------------------------------------------
// A caller's cloud is pageable memory (std::vector<Eigen::Vector3d>, a numpy array).  What the runtime does with a multi-megabyte
// pageable copy depends on its mood: seen on the same box, the same 3 MB download took 0.3 ms in one process and 8 ms in the next
// (tests/cpp/stream_mapping.cpp: 0.9 vs 24 ms per scan).  So copies above kStageMin go through two pinned buffers owned by the
// handle: memcpy into / out of the ring on the host, DMA between the ring and the device, one chunk in flight while the other is
// copied.  The host-to-device form returns with the last DMA queued (callers synchronise as before); device-to-host is complete.
constexpr size_t kStageBytes = 1 << 20, kStageMin = 128 << 10;  // 1 MB chunks: a 3 MB cloud is three chunks, DMA and host memcpy overlap
int stage_init(o3ds_handle h) {
  if (h->h_stage[1] && h->stage_ev[1]) return O3DS_OK;
  HIP_TRY(hipSetDevice(h->device));  // the events must belong to the device of the handle's stream
  for (int k = 0; k < 2; ++k) {
    if (!h->h_stage[k]) HIP_TRY(hipHostMalloc((void**)&h->h_stage[k], kStageBytes, hipHostMallocDefault));
    if (!h->stage_ev[k]) HIP_TRY(hipEventCreateWithFlags(&h->stage_ev[k], hipEventDisableTiming));
  }
  return O3DS_OK;
}
int h2d_copy(o3ds_handle h, void* d_dst, const void* h_src, size_t bytes, hipStream_t stream = nullptr /* the handle's */) {
  if (!stream) stream = h->stream;
  if (bytes < kStageMin) {
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, stream));
    return O3DS_OK;
  }
  int rc = stage_init(h);
  if (rc) return rc;
  int k = 0;
  for (size_t off = 0; off < bytes; off += kStageBytes, k ^= 1) {
    const size_t n = std::min(kStageBytes, bytes - off);
    HIP_TRY(hipEventSynchronize(h->stage_ev[k]));  // the DMA that last read this buffer is done (a fresh event is complete)
    memcpy(h->h_stage[k], (const char*)h_src + off, n);
    HIP_TRY(hipMemcpyAsync((char*)d_dst + off, h->h_stage[k], n, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(h->stage_ev[k], stream));
  }
  return O3DS_OK;
}
int d2h_copy(o3ds_handle h, void* h_dst, const void* d_src, size_t bytes) {  // synchronous: h_dst is filled on return
  if (bytes < kStageMin) {
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return O3DS_OK;
  }
  int rc = stage_init(h);
  if (rc) return rc;
  const size_t chunks = (bytes + kStageBytes - 1) / kStageBytes;
  for (size_t c = 0; c <= chunks; ++c) {
    if (c < chunks) {  // queue chunk c into buffer c & 1 (its previous content, chunk c - 2, was copied out one step ago)
      const size_t off = c * kStageBytes, n = std::min(kStageBytes, bytes - off);
      HIP_TRY(hipMemcpyAsync(h->h_stage[c & 1], (const char*)d_src + off, n, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipEventRecord(h->stage_ev[c & 1], h->stream));
    }
    if (c > 0) {  // while it flies, hand chunk c - 1 to the caller
      const size_t off = (c - 1) * kStageBytes, n = std::min(kStageBytes, bytes - off);
      HIP_TRY(hipEventSynchronize(h->stage_ev[(c - 1) & 1]));
      memcpy((char*)h_dst + off, h->h_stage[(c - 1) & 1], n);
    }
  }
  return O3DS_OK;
}

// ---- device memory -------------------------------------------------------------------------------------------------
// Every device buffer of a handle (clouds, indices, arena blocks) comes from this cache over plain hipMalloc.  All work of a handle
// is ordered on ONE stream (o3ds_set_stream synchronises the stream it leaves), so a block handed back by dev_free may be handed out
// again at once: whatever still reads it was enqueued earlier on the same stream -- the semantics of a stream-ordered pool without
// the runtime's.  Round 1 used hipMallocAsync / hipFreeAsync; under alloc / free churn of blocks of tens to hundreds of MB
// (estimate_normals with a fine pilot grid: a 150 MB scratch block, 40 MB cell tables per call) ROCm 7.2's pool intermittently
// handed out memory whose contents were then lost -- cell_start came back zeroed from some cell on, or the GPU faulted on it
// (scripts/stress_normals.py with O3DS_NRM_DEBUG=1 shows the broken table before the kernel that reads it runs; the scratch arena
// had met the same thing in round 1 and stopped returning its blocks).  Sizes are rounded up to 1/8 of their power of two, so the
// slowly varying sizes of a lidar stream hit the same classes; blocks go back to the driver only when the handle is destroyed.
hipError_t dev_alloc(o3ds_handle h, void** out, size_t bytes) {
  if (bytes < 256) bytes = 256;
  size_t gran = 256;
  while (gran * 8 < bytes) gran <<= 1;
  const size_t want = (bytes + gran - 1) / gran * gran;
  auto it = h->pool_free.lower_bound(want);
  if (it != h->pool_free.end() && it->first <= want + want / 4) {
    *out = it->second;
    h->pool_cached -= it->first;
    h->pool_free.erase(it);
    return hipSuccess;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess && !h->pool_free.empty()) {  // out of memory: give the cached blocks back and retry
    (void)hipStreamSynchronize(h->stream);
    for (auto& b : h->pool_free) {
      h->pool_bytes -= b.first;
      h->pool_size.erase(b.second);
      (void)hipFree(b.second);
    }
    h->pool_free.clear();
    h->pool_cached = 0;
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) return e;
  h->pool_size.emplace(p, want);
  h->pool_bytes += want;
  *out = p;
  return hipSuccess;
}
void dev_free(o3ds_handle h, void* p) {
  if (!p) return;
  auto it = h->pool_size.find(p);
  if (it == h->pool_size.end()) return;  // not ours (never happens)
  h->pool_free.emplace(it->second, p);
  h->pool_cached += it->second;
  // The cache is bounded: size classes drift as a map grows, and blocks of outgrown classes are never asked for again.  Beyond the cap
  // (O3DS_POOL_CAP_MB, default 32 GB of the 288) everything cached goes back to the driver -- after a stream synchronisation, because a
  // cached block may still be read by work enqueued before its dev_free.
  static const size_t cap = (getenv("O3DS_POOL_CAP_MB") ? (size_t)atoll(getenv("O3DS_POOL_CAP_MB")) : (size_t)32768) << 20;
  if (h->pool_cached > cap) {
    (void)hipStreamSynchronize(h->stream);
    for (auto& b : h->pool_free) {
      h->pool_bytes -= b.first;
      h->pool_size.erase(b.second);
      (void)hipFree(b.second);
    }
    h->pool_free.clear();
    h->pool_cached = 0;
  }
}
void dev_release_all(o3ds_handle h) {  // o3ds_destroy: the stream has been synchronised
  for (auto& b : h->pool_size) (void)hipFree(b.first);
  h->pool_size.clear();
  h->pool_free.clear();
  h->pool_bytes = 0;
  h->pool_cached = 0;
}

// device blocks obtained in a function that may still fail: handed back to the cache on every early return, kept on release()
struct DevGuard {
  o3ds_handle h;
  std::vector<void**> held;
  explicit DevGuard(o3ds_handle hh) : h(hh) {}
  ~DevGuard() {
    for (void** q : held)
      if (*q) {
        dev_free(h, *q);
        *q = nullptr;
      }
  }
  void add(void** q) { held.push_back(q); }
  void release() { held.clear(); }
};

// ---- scratch arena ------------------------------------------------------------------------------------------------
// Temporaries (scan block sums, flags, sort buffers, staging copies ...) are bump-allocated from blocks that persist
// for the life of the handle: a per-scan pipeline makes ~120 allocations otherwise (4.6 us each + free).  The bump
// pointer is reset at the start of every outermost ABI call; safe without synchronisation because all work of a handle
// is ordered on one stream.
int arena_alloc(o3ds_handle h, void** out, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  static const bool no_reuse = ab_getenv("O3DS_NO_ARENA") != nullptr;  // debugging aid: every temporary is its own pool allocation
  if (no_reuse) {
    char* p = nullptr;
    if (dev_alloc(h, (void**)&p, bytes) != hipSuccess) return fail(h, O3DS_ERR_OOM, "arena: out of memory");
    h->arena_blocks.emplace_back(p, 0);
    *out = p;
    return O3DS_OK;
  }
  for (;;) {
    if (h->arena_cur < h->arena_blocks.size()) {
      auto& b = h->arena_blocks[h->arena_cur];
      if (h->arena_off + bytes <= b.second) {
        *out = b.first + h->arena_off;
        h->arena_off += bytes;
        return O3DS_OK;
      }
      ++h->arena_cur;
      h->arena_off = 0;
      continue;
    }
    static const size_t first_mb = getenv("O3DS_ARENA_MB") ? (size_t)atoi(getenv("O3DS_ARENA_MB")) : 32;
    size_t want = std::max<size_t>(bytes, first_mb << 20);
    if (!h->arena_blocks.empty()) want = std::max(want, 2 * h->arena_blocks.back().second);
    char* p = nullptr;
    hipError_t e = dev_alloc(h, (void**)&p, want);
    if (e != hipSuccess) return fail(h, O3DS_ERR_OOM, std::string("arena: ") + hipGetErrorString(e));
    h->arena_blocks.emplace_back(p, want);
    if (ab_getenv("O3DS_ARENA_LOG"))
      fprintf(stderr, "[arena] new block %zu MB (request %zu B, blocks now %zu, prev off %zu)\n", want >> 20, bytes, h->arena_blocks.size(), h->arena_off);
  }
}
struct ArenaScope {
  o3ds_handle h;
  explicit ArenaScope(o3ds_handle hh) : h(hh) {
    if (h && h->arena_depth++ == 0) {
      if (ab_getenv("O3DS_NO_ARENA")) {
        for (auto& b : h->arena_blocks) dev_free(h, b.first);
        h->arena_blocks.clear();
      }
      // Blocks are never returned or merged while the handle lives: they grow geometrically, so there are at most a
      // handful, and a call simply walks through them.  (Freeing a 32 MB and a 64 MB block and immediately requesting
      // one 96 MB block from the stream-ordered pool corrupted live data on ROCm 7.2 -- timing dependent, reproduced
      // with scripts/debug_stream.py -- so the arena avoids free/alloc churn altogether.)
      h->arena_cur = 0;
      h->arena_off = 0;
    }
  }
  ~ArenaScope() {
    if (h) --h->arena_depth;
  }
};
// debugging aid: O3DS_SYNC_MASK re-inserts a stream synchronisation at the end of selected internal steps
inline void dbg_sync(o3ds_handle h, int bit) {
  static const int mask = ab_getenv("O3DS_SYNC_MASK") ? atoi(ab_getenv("O3DS_SYNC_MASK")) : 0;
  if (mask & bit) (void)hipStreamSynchronize(h->stream);
}
// tagged span marks on the handle's stream (only while profiling is enabled): begin and end alternate per tag
void span_mark(o3ds_handle h, int tag) {
  if (!h->profiling || tag < 0 || tag >= 16) return;
  auto& v = h->span_ev[tag];
  if (h->span_used[tag] >= v.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    v.push_back(e);
  }
  (void)hipEventRecord(v[h->span_used[tag]++], h->stream);
}
constexpr int kSpanNormalsKernels = 8, kSpanIndexBuild = 9;  // internal tags

#define TMP_ALLOC(ptr, bytes)                                   \
  do {                                                          \
    int _rc = arena_alloc(h, (void**)&(ptr), (size_t)(bytes));  \
    if (_rc) return _rc;                                        \
  } while (0)

inline int grid_for(size_t n, int cap = 4096) {
  size_t g = (n + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

size_t p4_size(int precision) { return precision == O3DS_PRECISION_F64 ? sizeof(P4d) : sizeof(P4f); }

void cloud_ready(o3ds_handle h, CloudRec& c);
int resolve_count(o3ds_handle h, CloudRec& c, bool block);
// A cloud for an operation on the handle's stream.  find_cloud: with its exact size (waits for a size a kernel has yet to publish -- the
// default, every operation that sizes anything on the host by the cloud); find_cloud_lazy: for the operations of the per-scan chain that
// take the size as it is (an upper bound + the device word, count_ref): normal estimation, registration, map insertion.
int pm_poll(o3ds_handle h, PMapRec* pm, bool block);
int pm_exit(o3ds_handle h, CloudRec& c);  // a persistent submap back into the reference's array form (defined with the map functions)
CloudRec* find_cloud_lazy(o3ds_handle h, o3ds_cloud id) {
  auto it = h->clouds.find(id);
  if (it == h->clouds.end()) return nullptr;
  cloud_ready(h, it->second);
  (void)resolve_count(h, it->second, false);
  return &it->second;
}
CloudRec* find_cloud(o3ds_handle h, o3ds_cloud id) {
  auto it = h->clouds.find(id);
  if (it == h->clouds.end()) return nullptr;
  cloud_ready(h, it->second);
  if (resolve_count(h, it->second, true) != O3DS_OK || (it->second.pm && pm_exit(h, it->second) != O3DS_OK)) {  // whoever wants the map as an
    h->deferred_err = h->err;                                                                                     // array gets the reference's array
    return nullptr;
  }
  return &it->second;
}

void free_index(o3ds_handle h, CloudRec& c) {
  if (c.cell_start) dev_free(h, c.cell_start);
  if (c.spts) dev_free(h, c.spts);
  if (c.snrm) dev_free(h, c.snrm);
  c.cell_start = nullptr;
  c.spts = c.snrm = nullptr;
  c.has_index = false;
  c.index_byproduct = false;
}
// ---- records kernels publish into, events, ingest buffers ---------------------------------------------------------------------------
constexpr int kPinRecs = 64;
inline int* cnt_word(o3ds_handle h, int slot) { return h->d_cnt + 16 * slot; }
int resolve_count(o3ds_handle h, CloudRec& c, bool block);

// a free record; when all are held (64 clouds with a size or a box in flight) the oldest holder among the CLOUDS is settled first.  A
// record of a persistent map (kRecOwnerMap) or of a cloud still being made (~0) is never taken from its owner; -1 if every record is one
// of those (sixty-odd persistent maps on one handle: the callers fail or do without)
constexpr uint64_t kRecOwnerMap = ~0ull - 1;
int take_rec(o3ds_handle h) {
  for (int tries = 0; tries < kPinRecs; ++tries) {
    const int slot = h->rec_next;
    h->rec_next = (h->rec_next + 1) % kPinRecs;
    if (h->rec_owner[slot] == 0) {
      h->rec_owner[slot] = ~0ull;  // (taken; add_cloud writes the id)
      return slot;
    }
  }
  for (int tries = 0; tries < kPinRecs; ++tries) {
    const int slot = h->rec_next;
    h->rec_next = (h->rec_next + 1) % kPinRecs;
    auto it = h->clouds.find(h->rec_owner[slot]);
    if (it == h->clouds.end()) continue;  // a map's record, or a cloud in the making
    if (it->second.lazy_slot == slot && resolve_count(h, it->second, true) != O3DS_OK) continue;
    if (it->second.pre_slot == slot && it->second.ingest_ev) continue;  // its box_publish_kernel may still be queued on the copy stream: it would write over the new owner
    if (it->second.pre_slot == slot) it->second.pre_slot = -1;  // the box is reduced again when it is asked for
    h->rec_owner[slot] = ~0ull;
    return slot;
  }
  return -1;
}
// the points of a cloud changed in place, or its point array was replaced: the box the asynchronous ingest reduced (pre_slot / pre_crop) is
// the box of points that no longer exist -- VoxelDownSample's grid origin, the 'nothing inside the volume' return and the index's clamping
// must not be fed from it (advisor, round 5).  The callers have used the cloud on the handle's stream, i.e. its ingest is ordered before
// whatever the record is used for next.
void drop_ingest_box(o3ds_handle h, CloudRec& c) {
  if (c.pre_slot >= 0) {
    h->rec_owner[c.pre_slot] = 0;
    c.pre_slot = -1;
  }
}
hipEvent_t take_event(o3ds_handle h) {
  if (!h->ev_pool.empty()) {
    hipEvent_t e = h->ev_pool.back();
    h->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  return e;
}
// the first use of an ingested cloud on the handle's stream: queue the wait for its copy + unpack (no host wait)
void cloud_ready(o3ds_handle h, CloudRec& c) {
  if (!c.ingest_ev) return;
  (void)hipStreamWaitEvent(h->stream, c.ingest_ev, 0);
  h->ev_pool.push_back(c.ingest_ev);
  c.ingest_ev = nullptr;
}
// the size of a cloud whose count a kernel published: taken from the pinned record if its stamp is there; `block`: wait for it
int resolve_count(o3ds_handle h, CloudRec& c, bool block) {
  if (c.lazy_slot < 0) return O3DS_OK;
  volatile o3ds_context::PinRec* r = h->h_rec + c.lazy_slot;
  if (r->seq != c.lazy_seq) {
    if (!block) return O3DS_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (r->seq != c.lazy_seq) return fail(h, O3DS_ERR_HIP, "the size of a cloud was never published by the kernel that decides it");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const int cnt = r->cnt;
  if (cnt < 0 || (size_t)cnt > c.n) return fail(h, O3DS_ERR_HIP, "published cloud size outside its bound");
  c.n = (size_t)cnt;
  h->voxel_count_hint = c.n;
  h->rec_owner[c.lazy_slot] = 0;
  c.lazy_slot = -1;
  return O3DS_OK;
}
// what a consumer kernel is handed for the cloud's size
inline CountRef count_ref(o3ds_handle h, const CloudRec& c) { return CountRef{c.n, c.lazy_slot >= 0 ? cnt_word(h, c.lazy_slot) : nullptr}; }

// the point array of a cloud goes back where it came from: the pool, or the ingest buffers (behind an event on the handle's stream, so
// that the next ingest into the buffer -- on the copy stream -- comes after everything that still reads it here)
void free_points(o3ds_handle h, CloudRec& c) {
  if (c.ingest_ev) cloud_ready(h, c);
  if (c.ingest_buf >= 0) {
    o3ds_context::IngestBuf& b = h->ingest_bufs[(size_t)c.ingest_buf];
    if (!b.freed) (void)hipEventCreateWithFlags(&b.freed, hipEventDisableTiming);
    (void)hipEventRecord(b.freed, h->stream);
    b.in_use = false;
    c.ingest_buf = -1;
  } else if (c.pts) {
    dev_free(h, c.pts);
  }
  c.pts = nullptr;
}
void pm_release(o3ds_handle h, CloudRec& c) {  // the persistent form's own blocks (slot arrays and index arrays stay with the cloud)
  if (!c.pm) return;
  for (void* b : c.pm->blocks) dev_free(h, b);
  if (c.pm->rec_slot >= 0) h->rec_owner[c.pm->rec_slot] = 0;
  delete c.pm;
  c.pm = nullptr;
  c.index_positions = 0;
}
void free_cloud(o3ds_handle h, CloudRec& c) {
  pm_release(h, c);
  free_index(h, c);
  free_points(h, c);
  if (c.nrm) dev_free(h, c.nrm);
  if (c.col) dev_free(h, c.col);
  c.nrm = c.col = nullptr;
  if (c.lazy_slot >= 0) h->rec_owner[c.lazy_slot] = 0;
  if (c.pre_slot >= 0) h->rec_owner[c.pre_slot] = 0;
  c.lazy_slot = c.pre_slot = -1;
  c.n = 0;
  c.cap = 0;
}

// a cloud under construction in a function that may still fail: freed on every early return, handed over by release()
struct CloudGuard {
  o3ds_handle h;
  CloudRec* c;
  CloudGuard(o3ds_handle hh, CloudRec& cc) : h(hh), c(&cc) {}
  ~CloudGuard() {
    if (c) free_cloud(h, *c);
  }
  void release() { c = nullptr; }
};

// exclusive scan of m ints (in -> out) with the hand-written 3-phase scan; in may alias out
// pub: null, or a device-visible address (the handle's pinned block) that also receives out[m - 1] -- the total, when the input ends in a
// zero sentinel -- from the kernel that finishes that element
template <typename T>
int exclusive_scan_t(o3ds_handle h, const T* in, T* out, size_t m, T* pub = nullptr) {
  if (m == 0) return O3DS_OK;
  const int nb = (int)((m + kScanPerBlock - 1) / kScanPerBlock);
  T* sums = nullptr;
  TMP_ALLOC(sums, sizeof(T) * (size_t)nb);
  scan_local_kernel<T><<<nb, kBlock, 0, h->stream>>>(in, out, sums, m, nb == 1 ? pub : nullptr);
  if (nb > 1 && nb <= kScanFusedBlocks) {
    scan_add_fused_kernel<T><<<nb, kBlock, 0, h->stream>>>(out, sums, m, pub);
  } else if (nb > 1) {
    scan_sums_kernel<T><<<1, kBlock, 0, h->stream>>>(sums, nb);
    scan_add_kernel<T><<<nb, kBlock, 0, h->stream>>>(out, sums, m, pub);
  }
  HIP_TRY(hipGetLastError());
  dbg_sync(h, 1);
  return O3DS_OK;  // results are stream-ordered; callers that need a value on the host copy it back and synchronise
}
int exclusive_scan_int(o3ds_handle h, const int* in, int* out, size_t m, int* pub = nullptr) { return exclusive_scan_t<int>(h, in, out, m, pub); }

// slots of the pinned block for values kernels publish themselves (the front of the block belongs to bbox_of / read_back)
template <typename T>
T* pub_slot(o3ds_handle h, int k) { return (T*)(h->h_pin_dev + kPubOff + 16 * (size_t)k); }
template <typename T>
T pub_value(o3ds_handle h, int k) { return *(const volatile T*)(h->h_pin + kPubOff + 16 * (size_t)k); }
int wait_stream(o3ds_handle h) {
  HIP_TRY(hipStreamSynchronize(h->stream));
  return O3DS_OK;
}
// The fused registration loop's wait.  The launch the host waits for writes the state into pinned memory and then, with a system-scope
// release, the stamp `seq` into a pinned word (icp_fused_kernel); every earlier launch of the stream has completed by then, and once the
// loop is done the rest of that launch is workgroups returning.  Watching the word hands the result over as it is written instead of
// after the kernel's completion signal has made its way through the runtime (a few microseconds per registration of ~200).  A stamp
// that does not show up within 50 ms (a faulted queue) falls back to the stream wait, which reports the error.
constexpr int kSeqSlot = 8;
constexpr int kWaitSpinUs = 20;
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
hipError_t wait_fused_state(o3ds_handle h, unsigned long long seq) {
  static const bool watch = !(ab_getenv("O3DS_ICP_WATCH_STATE") && atoi(ab_getenv("O3DS_ICP_WATCH_STATE")) == 0);
  if (watch) {
    const volatile unsigned long long* w = (const volatile unsigned long long*)(h->h_pin + kPubOff + 16 * (size_t)kSeqSlot);
    // spin for ~20 us (a steady pass launch is 10 us: the stamp of a registration that is about to finish shows up inside that), then
    // give the core away between looks: SlamWrapper runs four to seven threads (SlamWrapper.cpp:227-236) and a GPU box grants its
    // container a CPU quota -- a core that spins through a whole registration is a core the other worker does not get
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int spins = 1;; ++spins) {
      if (*w == seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return hipSuccess;
      }
      cpu_relax();
      if ((spins & 0x3fu) == 0) {
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > std::chrono::milliseconds(50)) break;
        if (dt > std::chrono::microseconds(kWaitSpinUs)) sched_yield();
      }
    }
  }
  return hipStreamSynchronize(h->stream);
}

template <typename P4>
int bbox_of(o3ds_handle h, const P4* pts, size_t n, double mn[3], double mx[3], const CropDev* crop = nullptr) {
  const int g = grid_for(n, 1024);  // 1024 x 48 B fit the pinned block below the published slots
  static_assert(1024 * 6 * sizeof(double) <= kPinBytes - 256, "bbox partials fit the pinned block");
  CropDev all{};
  bbox_kernel<P4><<<g, kBlock, 0, h->stream>>>(pts, n, crop ? *crop : all, (double*)h->h_pin_dev);  // every block stores its box there itself
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double* hb = (const double*)h->h_pin;
  for (int a = 0; a < 3; ++a) {
    mn[a] = 1e300;
    mx[a] = -1e300;
  }
  for (int b = 0; b < g; ++b)
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], hb[(size_t)b * 6 + a]);
      mx[a] = std::max(mx[a], hb[(size_t)b * 6 + 3 + a]);
    }
  return O3DS_OK;
}

// Build one uniform grid over `pts`: cell_start (exclusive scan of the per-cell counts) and the points (+ normals when
// given) re-stored in cell order.  Output buffers are allocated here; the caller owns them.
template <typename P4>
int build_grid_t(o3ds_handle h, const P4* pts, const P4* nrm, size_t n, double cell, GridDev* out_grid, int** out_cell_start, void** out_spts,
                 void** out_snrm, CloudRec* box /* in: a known box, if has_box; out: the box used */,
                 const int* n_dev = nullptr /* n is an upper bound, the exact count is here (a cloud whose size the host has not seen: it then brings its box) */) {
  if (n == 0) return fail(h, O3DS_ERR_EMPTY, "build_index: empty cloud");
  double mn[3], mx[3];
  int rc = O3DS_OK;
  static const bool no_box_cache = ab_getenv("O3DS_NO_BOX_CACHE") != nullptr;  // debugging aid: always reduce
  if (box && box->has_box && !no_box_cache) {
    for (int a = 0; a < 3; ++a) mn[a] = box->bmn[a], mx[a] = box->bmx[a];
  } else {
    if (n_dev) return fail(h, O3DS_ERR_INVALID_ARG, "build_index: a cloud whose size is still in flight must bring its box");
    rc = bbox_of<P4>(h, pts, n, mn, mx);
    if (rc) return rc;
    if (box) {
      box->has_box = true;
      box->box_padded = false;  // exact bounds of the stored values
      for (int a = 0; a < 3; ++a) box->bmn[a] = mn[a], box->bmx[a] = mx[a];
    }
  }
  for (int a = 0; a < 3; ++a)
    if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) return fail(h, O3DS_ERR_INVALID_ARG, "build_index: non-finite coordinates");
  size_t nx, ny, nz;
  for (;;) {
    nx = (size_t)std::floor((mx[0] - mn[0]) / cell) + 1;
    ny = (size_t)std::floor((mx[1] - mn[1]) / cell) + 1;
    nz = (size_t)std::floor((mx[2] - mn[2]) / cell) + 1;
    if (nx * ny * nz <= kMaxCells) break;
    cell *= 1.26;  // ~ x2 fewer cells per step
  }
  const size_t ncell = nx * ny * nz;
  GridDev g{};
  g.ox = mn[0];
  g.oy = mn[1];
  g.oz = mn[2];
  g.cell = cell;
  g.inv_cell = 1.0 / cell;
  g.nx = (int)nx;
  g.sx = (int)nx;
  g.ny = (int)ny;
  g.nz = (int)nz;
  int *counts = nullptr, *cell_id = nullptr, *cell_start = nullptr;
  void *spts = nullptr, *snrm = nullptr;
  // per-cell counters: the handle's block of zeros (ncell + 1: the scan's sentinel stays zero); counted up by cell_count_kernel, read
  // by the scan, counted back down to zero by scatter_kernel
  if (h->cells_cap < ncell + 1) {
    if (h->d_cells) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_cells);
      h->d_cells = nullptr;
      h->cells_cap = 0;
    }
    const size_t cap = ncell + 1 + ncell / 4;
    if (hipMalloc((void**)&h->d_cells, sizeof(int) * cap) != hipSuccess) return fail(h, O3DS_ERR_OOM, "build_index: cell counters allocation failed");
    h->cells_cap = cap;
    h->cells_clean = false;
  }
  static const bool always_clear = ab_getenv("O3DS_ALWAYS_CLEAR") != nullptr;  // debugging / A/B aid: do not rely on self-cleaning scratch
  if (!h->cells_clean || always_clear) HIP_TRY(hipMemsetAsync(h->d_cells, 0, sizeof(int) * h->cells_cap, h->stream));
  h->cells_clean = false;
  counts = h->d_cells;
  TMP_ALLOC(cell_id, sizeof(int) * n);
  HIP_TRY(dev_alloc(h, (void**)&cell_start, sizeof(int) * (ncell + 1 + 4)));  // +4: the search reads rows as unaligned 16-B vectors
  HIP_TRY(dev_alloc(h, (void**)&spts, sizeof(P4) * n));
  if (nrm) HIP_TRY(dev_alloc(h, (void**)&snrm, sizeof(P4) * n));
  span_mark(h, kSpanIndexBuild);
  cell_count_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(pts, n, g, counts, cell_id, n_dev);
  rc = exclusive_scan_int(h, counts, cell_start, ncell + 1);
  if (rc) return rc;
  scatter_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(pts, nrm, n, cell_id, cell_start, counts, (P4*)spts, (P4*)snrm, n_dev);
  span_mark(h, kSpanIndexBuild);
  HIP_TRY(hipGetLastError());
  h->cells_clean = true;  // (stream order: whoever counts next runs after the scatter)
  dbg_sync(h, 2);
  g.cell_start = cell_start;
  *out_grid = g;
  *out_cell_start = cell_start;
  *out_spts = spts;
  if (out_snrm) *out_snrm = snrm;
  return O3DS_OK;
}

template <typename P4>
int build_index_t(o3ds_handle h, CloudRec& c, double cell) {
  free_index(h, c);
  if (c.lazy_slot >= 0 && !c.has_box) {
    const int rr = resolve_count(h, c, true);
    if (rr) return rr;
  }
  int rc = build_grid_t<P4>(h, (const P4*)c.pts, (const P4*)c.nrm, c.n, cell, &c.grid, &c.cell_start, &c.spts, &c.snrm, &c,
                            c.lazy_slot >= 0 ? cnt_word(h, c.lazy_slot) : nullptr);
  if (rc) return rc;
  c.has_index = true;
  return O3DS_OK;
}

// grid cell of a registration target = max_correspondence_distance / this (O3DS_INDEX_CELL_DIV: tuning experiments)
double index_cell_div() {
  static const double d = ab_getenv("O3DS_INDEX_CELL_DIV") ? std::max(1.0, atof(ab_getenv("O3DS_INDEX_CELL_DIV"))) : 4.0;
  return d;
}

int build_index(o3ds_handle h, CloudRec& c, double cell) {
  return c.precision == O3DS_PRECISION_F64 ? build_index_t<P4d>(h, c, cell) : build_index_t<P4f>(h, c, cell);
}

// f32 storage: the caller's doubles are narrowed on their way into the pinned ring (the narrowing the device would do: round to nearest
// even) and widened on their way out, so half the bytes cross PCIe and the host touches every value once either way
int h2d_copy_narrow(o3ds_handle h, float* d_dst, const double* h_src, size_t count) {
  int rc = stage_init(h);
  if (rc) return rc;
  const size_t per = kStageBytes / sizeof(float);
  int k = 0;
  for (size_t off = 0; off < count; off += per, k ^= 1) {
    const size_t n = std::min(per, count - off);
    HIP_TRY(hipEventSynchronize(h->stage_ev[k]));
    float* dst = (float*)h->h_stage[k];
    const double* src = h_src + off;
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];
    HIP_TRY(hipMemcpyAsync(d_dst + off, dst, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->stage_ev[k], h->stream));
  }
  return O3DS_OK;
}
int d2h_copy_widen(o3ds_handle h, double* h_dst, const float* d_src, size_t count) {  // synchronous
  int rc = stage_init(h);
  if (rc) return rc;
  const size_t per = kStageBytes / sizeof(float);
  const size_t chunks = (count + per - 1) / per;
  for (size_t c = 0; c <= chunks; ++c) {
    if (c < chunks) {
      const size_t off = c * per, n = std::min(per, count - off);
      HIP_TRY(hipMemcpyAsync(h->h_stage[c & 1], d_src + off, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipEventRecord(h->stage_ev[c & 1], h->stream));
    }
    if (c > 0) {
      const size_t off = (c - 1) * per, n = std::min(per, count - off);
      HIP_TRY(hipEventSynchronize(h->stage_ev[(c - 1) & 1]));
      const float* src = (const float*)h->h_stage[(c - 1) & 1];
      double* dst = h_dst + off;
      for (size_t i = 0; i < n; ++i) dst[i] = (double)src[i];
    }
  }
  return O3DS_OK;
}
static const bool kNarrowOnHost = ab_getenv("O3DS_NO_HOST_NARROW") == nullptr;  // A/B switch

template <typename P4>
int upload_array_t(o3ds_handle h, const double* host, size_t n, P4* d_out) {  // host double[3n] -> device P4[n]
  if (std::is_same<P4, P4f>::value && kNarrowOnHost && sizeof(double) * 3 * n >= kStageMin) {
    float* stage = nullptr;
    TMP_ALLOC(stage, sizeof(float) * 3 * n);
    const int rcc = h2d_copy_narrow(h, stage, host, 3 * n);
    if (rcc) return rcc;
    pack_kernel<P4, float><<<grid_for(n), kBlock, 0, h->stream>>>(stage, n, d_out);
    return O3DS_OK;
  }
  double* stage = nullptr;
  TMP_ALLOC(stage, sizeof(double) * 3 * n);
  const int rcc = h2d_copy(h, stage, host, sizeof(double) * 3 * n);
  if (rcc) return rcc;
  pack_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(stage, n, d_out);
  return O3DS_OK;
}
template <typename P4>
int download_array_t(o3ds_handle h, const P4* d_in, size_t n, double* host) {  // device P4[n] -> host double[3n]; complete on return
  if (std::is_same<P4, P4f>::value && kNarrowOnHost && sizeof(double) * 3 * n >= kStageMin) {
    float* stage = nullptr;
    TMP_ALLOC(stage, sizeof(float) * 3 * n);
    unpack_kernel<P4, float><<<grid_for(n), kBlock, 0, h->stream>>>(d_in, n, stage);
    return d2h_copy_widen(h, host, stage, 3 * n);
  }
  double* stage = nullptr;
  TMP_ALLOC(stage, sizeof(double) * 3 * n);
  unpack_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(d_in, n, stage);
  return d2h_copy(h, host, stage, sizeof(double) * 3 * n);
}

template <typename P4>
int upload_t(o3ds_handle h, const double* xyz, const double* normals, size_t n, CloudRec& c) {
  c.n = n;
  c.precision = h->precision;
  if (n == 0) return O3DS_OK;
  HIP_TRY(dev_alloc(h, (void**)&c.pts, sizeof(P4) * n));
  int rcc = upload_array_t<P4>(h, xyz, n, (P4*)c.pts);
  if (rcc) return rcc;
  if (normals) {
    HIP_TRY(dev_alloc(h, (void**)&c.nrm, sizeof(P4) * n));
    rcc = upload_array_t<P4>(h, normals, n, (P4*)c.nrm);
    if (rcc) return rcc;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));  // the caller may release its host buffers as soon as this returns
  return O3DS_OK;
}

template <typename P4>
int download_t(o3ds_handle h, const CloudRec& c, double* xyz, double* normals) {
  if (c.n == 0) return O3DS_OK;
  if (xyz) {
    const int rcc = download_array_t<P4>(h, (const P4*)c.pts, c.n, xyz);
    if (rcc) return rcc;
  }
  if (normals && c.nrm) {
    const int rcc = download_array_t<P4>(h, (const P4*)c.nrm, c.n, normals);
    if (rcc) return rcc;
  }
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

o3ds_cloud add_cloud(o3ds_handle h, CloudRec&& c) {
  const uint64_t id = h->next_id++;
  if (c.lazy_slot >= 0) h->rec_owner[c.lazy_slot] = id;
  if (c.pre_slot >= 0) h->rec_owner[c.pre_slot] = id;
  h->clouds.emplace(id, std::move(c));
  return id;
}

const double kIdentity16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

// ---- dense voxel map helpers -----------------------------------------------------------------------
void dense_release(o3ds_handle h, DenseRec& d) {
  if (d.dev.keys) dev_free(h, d.dev.keys);
  if (d.dev.cnt) dev_free(h, d.dev.cnt);
  if (d.dev.sp) dev_free(h, d.dev.sp);
  if (d.dev.sn) dev_free(h, d.dev.sn);
  if (d.dev.sc) dev_free(h, d.dev.sc);
  d.dev = o3ds::DenseDev{};
  d.cap = 0;
}

int dense_alloc(o3ds_handle h, size_t cap, o3ds::DenseDev* out) {
  o3ds::DenseDev d{};
  HIP_TRY(dev_alloc(h, (void**)&d.keys, sizeof(unsigned long long) * cap));
  HIP_TRY(dev_alloc(h, (void**)&d.cnt, sizeof(int) * cap));
  HIP_TRY(dev_alloc(h, (void**)&d.sp, sizeof(long long) * 3 * cap));
  HIP_TRY(dev_alloc(h, (void**)&d.sn, sizeof(long long) * 3 * cap));
  HIP_TRY(dev_alloc(h, (void**)&d.sc, sizeof(long long) * 3 * cap));
  HIP_TRY(hipMemsetAsync(d.keys, 0xFF, sizeof(unsigned long long) * cap, h->stream));
  HIP_TRY(hipMemsetAsync(d.cnt, 0, sizeof(int) * cap, h->stream));
  HIP_TRY(hipMemsetAsync(d.sp, 0, sizeof(long long) * 3 * cap, h->stream));
  HIP_TRY(hipMemsetAsync(d.sn, 0, sizeof(long long) * 3 * cap, h->stream));
  HIP_TRY(hipMemsetAsync(d.sc, 0, sizeof(long long) * 3 * cap, h->stream));
  d.mask = (unsigned int)(cap - 1);
  *out = d;
  return O3DS_OK;
}

// number of live voxels and of occupied slots (one scan + two small read backs)
int dense_count(o3ds_handle h, DenseRec& d, size_t* n_live, int** flag_out = nullptr, int** pos_out = nullptr, size_t* n_occupied = nullptr) {
  *n_live = 0;
  if (n_occupied) *n_occupied = 0;
  if (d.cap == 0) return O3DS_OK;
  int *flag = nullptr, *pos = nullptr;
  unsigned long long* d_occ = nullptr;
  TMP_ALLOC(flag, sizeof(int) * (d.cap + 1));
  TMP_ALLOC(pos, sizeof(int) * (d.cap + 1));
  TMP_ALLOC(d_occ, sizeof(unsigned long long));
  HIP_TRY(hipMemsetAsync(flag + d.cap, 0, sizeof(int), h->stream));
  HIP_TRY(hipMemsetAsync(d_occ, 0, sizeof(unsigned long long), h->stream));
  dense_used_flag_kernel<<<grid_for(d.cap), kBlock, 0, h->stream>>>(d.dev, d.cap, flag, d_occ);
  int rc = exclusive_scan_int(h, flag, pos, d.cap + 1);
  if (rc) return rc;
  int total = 0;
  unsigned long long occ = 0;
  rc = read_back(h, {{&total, pos + d.cap, sizeof(int)}, {&occ, d_occ, sizeof(occ)}});
  if (rc) return rc;
  *n_live = (size_t)total;
  if (n_occupied) *n_occupied = (size_t)occ;
  d.used_upper = (size_t)occ;  // exact now
  if (flag_out) *flag_out = flag;
  if (pos_out) *pos_out = pos;
  return O3DS_OK;
}

// make room for up to n_new more voxels: occupied slots (live + tombstones) stay <= cap / 2, so a probe always finds a free slot
int dense_reserve(o3ds_handle h, DenseRec& d, size_t n_new) {
  if (d.cap && (d.used_upper + n_new) * 2 <= d.cap) {
    d.used_upper += n_new;
    return O3DS_OK;
  }
  size_t live = 0, occupied = 0;
  if (d.cap) {
    int rc = dense_count(h, d, &live, nullptr, nullptr, &occupied);  // the bound was pessimistic (most points fall into existing voxels)
    if (rc) return rc;
    if ((occupied + n_new) * 2 <= d.cap) {
      d.used_upper = occupied + n_new;
      return O3DS_OK;
    }
  }
  // rehash: only live voxels move, tombstones are dropped -- so the new table may even be the same size as the old one
  size_t cap = 1024;
  while (cap < 4 * (live + n_new)) cap <<= 1;
  if (cap > ((size_t)1 << 31)) return fail(h, O3DS_ERR_OOM, "dense map: table would exceed 2^31 slots");
  o3ds::DenseDev nd{};
  int rc = dense_alloc(h, cap, &nd);
  if (rc) return rc;
  if (d.cap) {
    dense_rehash_kernel<<<grid_for(d.cap), kBlock, 0, h->stream>>>(d.dev, d.cap, nd);
    HIP_TRY(hipGetLastError());
    DenseRec old = d;
    dense_release(h, old);
  }
  d.dev = nd;
  d.cap = cap;
  d.used_upper = live + n_new;
  return O3DS_OK;
}

template <typename P4>
int dense_insert_t(o3ds_handle h, DenseRec& d, const CloudRec& c, const double T[16]) {
  if (c.n == 0) return O3DS_OK;
  int rc = dense_reserve(h, d, c.n);
  if (rc) return rc;
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) M.m[r * 4 + col] = T ? T[col * 4 + r] : (r == col ? 1.0 : 0.0);
  dense_insert_kernel<P4><<<grid_for(c.n), kBlock, 0, h->stream>>>((const P4*)c.pts, (const P4*)c.nrm, (const P4*)c.col, c.n, M, 1.0 / d.voxel, d.dev);
  HIP_TRY(hipGetLastError());
  if (c.nrm) d.has_normals = true;  // isHasNormals_, Voxel.cpp:80-83
  if (c.col) d.has_colors = true;   // isHasColors_, Voxel.cpp:84-87
  return O3DS_OK;
}

template <typename P4>
int dense_to_cloud_t(o3ds_handle h, DenseRec& d, CloudRec& out) {
  out.precision = h->precision;
  out.n = 0;
  size_t used = 0;
  int *flag = nullptr, *pos = nullptr;
  int rc = dense_count(h, d, &used, &flag, &pos);
  if (rc || used == 0) return rc;
  unsigned long long *k0 = nullptr, *k1 = nullptr;
  uint32_t *s0 = nullptr, *s1 = nullptr;
  TMP_ALLOC(k0, sizeof(unsigned long long) * used);
  TMP_ALLOC(k1, sizeof(unsigned long long) * used);
  TMP_ALLOC(s0, sizeof(uint32_t) * used);
  TMP_ALLOC(s1, sizeof(uint32_t) * used);
  dense_list_kernel<<<grid_for(d.cap), kBlock, 0, h->stream>>>(d.dev, d.cap, flag, pos, k0, s0);
  size_t temp_bytes = 0;  // ascending key order: the table order depends on how insertions interleaved, the output must not
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, s0, s1, used, 0, 64, h->stream));
  void* temp = nullptr;
  TMP_ALLOC(temp, temp_bytes ? temp_bytes : 16);
  HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, s0, s1, used, 0, 64, h->stream));
  out.n = used;
  HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * used));
  if (d.has_normals) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * used));
  if (d.has_colors) HIP_TRY(dev_alloc(h, (void**)&out.col, sizeof(P4) * used));
  dense_emit_kernel<P4><<<grid_for(used), kBlock, 0, h->stream>>>(d.dev, s1, used, (P4*)out.pts, (P4*)out.nrm, (P4*)out.col);
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

inline QuantumTable quantum_table(const IcpPassArgs& a) {
  QuantumTable t;
  for (int k = 0; k < kRec; ++k) t.q[k] = a.q_hi[k];
  return t;
}

// ---- ICP launch helpers -------------------------------------------------------------------------
// threads per workgroup of the pass kernels (4 lanes per query): every form of the pass uses the same number, so that the queries are
// partitioned into the same batches -- a batch's record is one rounded f64 reduction -- and the forms agree bit for bit
#ifndef O3DS_ICP_BLOCK
#define O3DS_ICP_BLOCK 512
#endif
constexpr int kIcpBlock = O3DS_ICP_BLOCK, kIcpQ = kIcpBlock / 4;

template <typename P4>
void launch_accumulate(o3ds_handle h, const IcpPassArgs& a, bool crop, int nblocks) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profiling == 1) {
    if (h->ev_used + 2 > h->ev.size()) {
      hipEvent_t a0, a1;
      if (hipEventCreate(&a0) == hipSuccess && hipEventCreate(&a1) == hipSuccess) {
        h->ev.push_back(a0);
        h->ev.push_back(a1);
      }
    }
    if (h->ev_used + 2 <= h->ev.size()) {
      e0 = h->ev[h->ev_used];
      e1 = h->ev[h->ev_used + 1];
      h->ev_used += 2;
      (void)hipEventRecord(e0, h->stream);
    }
  }
  // one geometry: kIcpBlock threads = kIcpBlock / 4 queries x 4 lanes (G = 2 / 8 were swept and dropped; 512 threads against 256: a launch costs 2.7 ns per
  // workgroup beyond the first 256 -- scripts/ubench/launch_shape.hip -- and a steady pass is 9.7 us instead of 10.2)
  if (a.keys_mode != 0) {  // target-sharded registration (o3ds_icp_nn_keys / o3ds_icp_accumulate_keys): the instantiation with the key code
    if (h->session_method == O3DS_ICP_GENERALIZED) {
      if (crop)
        icp_accumulate_kernel<P4, true, kIcpBlock, 4, true, true><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
      else
        icp_accumulate_kernel<P4, false, kIcpBlock, 4, true, true><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
    } else {
      if (crop)
        icp_accumulate_kernel<P4, true, kIcpBlock, 4, false, true><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
      else
        icp_accumulate_kernel<P4, false, kIcpBlock, 4, false, true><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
    }
  } else if (h->session_method == O3DS_ICP_GENERALIZED) {
    if (crop)
      icp_accumulate_kernel<P4, true, kIcpBlock, 4, true><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
    else
      icp_accumulate_kernel<P4, false, kIcpBlock, 4, true><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
  } else {
    if (crop)
      icp_accumulate_kernel<P4, true, kIcpBlock, 4, false><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
    else
      icp_accumulate_kernel<P4, false, kIcpBlock, 4, false><<<nblocks, kIcpBlock, 0, h->stream>>>(a);
  }
  if (e1) (void)hipEventRecord(e1, h->stream);
}


// fused form: byte offsets inside d_fused
constexpr size_t kFusedStateStride = 256;
constexpr size_t kFusedSlotsOff = 2 * kFusedStateStride;
constexpr size_t kFusedSlotBufBytes = (size_t)kFusedSlots * kSlotDoubles * sizeof(double);
constexpr size_t kFusedBytes = kFusedSlotsOff + 3 * kFusedSlotBufBytes;
static_assert(sizeof(IcpStateDev) <= kFusedStateStride, "state slot too small");

template <typename P4>
void launch_fused(o3ds_handle h, const IcpFusedArgs& fa, bool crop, int nblocks, bool bracket) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profiling == 1 && bracket) {
    if (h->ev_used + 2 > h->ev.size()) {
      hipEvent_t a0, a1;
      if (hipEventCreate(&a0) == hipSuccess && hipEventCreate(&a1) == hipSuccess) {
        h->ev.push_back(a0);
        h->ev.push_back(a1);
      }
    }
    if (h->ev_used + 2 <= h->ev.size()) {
      e0 = h->ev[h->ev_used];
      e1 = h->ev[h->ev_used + 1];
      h->ev_used += 2;
      (void)hipEventRecord(e0, h->stream);
    }
  }
  if (h->session_method == O3DS_ICP_GENERALIZED) {
    if (crop)
      icp_fused_kernel<P4, true, kIcpBlock, 4, true><<<nblocks, kIcpBlock, 0, h->stream>>>(fa.state_in, fa.slots_in, fa.first, fa);
    else
      icp_fused_kernel<P4, false, kIcpBlock, 4, true><<<nblocks, kIcpBlock, 0, h->stream>>>(fa.state_in, fa.slots_in, fa.first, fa);
  } else {
    if (crop)
      icp_fused_kernel<P4, true, kIcpBlock, 4, false><<<nblocks, kIcpBlock, 0, h->stream>>>(fa.state_in, fa.slots_in, fa.first, fa);
    else
      icp_fused_kernel<P4, false, kIcpBlock, 4, false><<<nblocks, kIcpBlock, 0, h->stream>>>(fa.state_in, fa.slots_in, fa.first, fa);
  }
  if (e1) (void)hipEventRecord(e1, h->stream);
}

// the fused kernel serves ONE batch of kIcpQ queries per workgroup (no batch loop: icp_pass_body, kSingle); its records go to slot
// blockIdx % kFusedSlots, so the grid has no capacity to respect (the exact sums hold for 4 096 workgroup records: 2 048 at most here)
constexpr size_t kFusedMaxQueries = (size_t)4096 * 64;
static_assert(kFusedMaxQueries == O3DS_ICP_PASS_MAX_QUERIES, "the limit the header documents");
int fused_blocks(size_t count) { return (int)std::max<size_t>((count + kIcpQ - 1) / kIcpQ, 1); }

int pass_blocks(o3ds_handle h, size_t count) {
  const size_t qpb = kIcpQ;  // one batch of kIcpBlock / 4 queries per workgroup iteration
  size_t g = (count + qpb - 1) / qpb;
  if (g < 1) g = 1;
  if (g > (size_t)h->pass_rows) g = h->pass_rows;
  return (int)g;
}

int validate_icp(o3ds_handle h, const CloudRec* src, const CloudRec* tgt, const o3ds_icp_params* p) {
  if (!src || !tgt) return fail(h, O3DS_ERR_INVALID_ARG, "icp: unknown cloud id");
  if (!p) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null params");
  if (tgt->n == 0) return fail(h, O3DS_ERR_EMPTY, "icp: empty target (map patch size is zero)");  // ScanToMapRegistration.cpp:60
  if (!(p->max_correspondence_distance > 0.0))
    return fail(h, O3DS_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");  // [O3D] RegistrationICP
  if (p->method != O3DS_ICP_POINT_TO_PLANE && p->method != O3DS_ICP_GENERALIZED && p->method != O3DS_ICP_POINT_TO_POINT)
    return fail(h, O3DS_ERR_INVALID_ARG, "icp: unknown method");
  if (!tgt->nrm && p->method != O3DS_ICP_POINT_TO_POINT)
    return fail(h, O3DS_ERR_NO_NORMALS, p->method == O3DS_ICP_GENERALIZED ? "generalized ICP: target has no normals (covariances are built from normals)"
                                                                           : "TransformationEstimationPointToPlane requires target normals");
  if (p->method == O3DS_ICP_GENERALIZED && src->n > 0 && !src->nrm)
    return fail(h, O3DS_ERR_NO_NORMALS, "generalized ICP: source has no normals (covariances are built from normals)");
  if (src->precision != tgt->precision) return fail(h, O3DS_ERR_INVALID_ARG, "icp: source/target precision mismatch");
  if (p->max_iteration < 0) return fail(h, O3DS_ERR_INVALID_ARG, "icp: negative max_iteration");
  return O3DS_OK;
}

int begin_session(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* crop, const double init[16],
                  const o3ds_icp_params* params, bool upload_state = true) {
  // (sizes as they are: a source or target whose size a kernel has yet to publish enters with its upper bound and the device word)
  CloudRec* src = find_cloud_lazy(h, source);
  CloudRec* tgt = find_cloud_lazy(h, target);
  // a persistent submap stays persistent as the TARGET (its row-paged index is what the pass kernel reads); as the SOURCE its slot array
  // -- dead and never-written slots among the live ones -- would become queries and the fitness denominator: it is folded first
  if (src && src->pm) {
    const int re = pm_exit(h, *src);
    if (re) return re;
    tgt = find_cloud_lazy(h, target);  // (source == target: the fold dropped the index)
  }
  int rc = validate_icp(h, src, tgt, params);
  if (rc) return rc;
  const double r_hint = params->max_correspondence_distance;
  static const double reuse_max = ab_getenv("O3DS_INDEX_REUSE_MAX") ? atof(ab_getenv("O3DS_INDEX_REUSE_MAX")) : 0.75;  // tuning experiments
  if (!tgt->has_index || (tgt->nrm && !tgt->snrm) || (tgt->index_byproduct && (tgt->grid.cell < r_hint / 8.0 || tgt->grid.cell > r_hint * reuse_max))) {
    rc = build_index(h, *tgt, params->max_correspondence_distance / index_cell_div());
    if (rc) return rc;
  }
  if (src->n > h->nn_cache_cap) {
    if (h->d_nn_cache) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_nn_cache);
      h->d_nn_cache = nullptr;
      h->nn_cache_cap = 0;
    }
    const size_t cap = (std::max<size_t>(src->n + src->n / 4, 1 << 16) + 7) & ~(size_t)7;
    if (hipMalloc((void**)&h->d_nn_cache, cap * sizeof(int)) != hipSuccess) return fail(h, O3DS_ERR_OOM, "icp: match cache allocation failed");
    if (h->d_set_pos) (void)hipFree(h->d_set_pos);
    h->d_set_pos = nullptr;
    h->d_set_ref = nullptr;
    if (hipMalloc((void**)&h->d_set_pos, cap * (kSetCap * sizeof(int) + 4 * sizeof(double))) != hipSuccess) {
      (void)hipFree(h->d_nn_cache);
      h->d_nn_cache = nullptr;
      return fail(h, O3DS_ERR_OOM, "icp: candidate-set allocation failed");
    }
    h->d_set_ref = (char*)h->d_set_pos + cap * kSetCap * sizeof(int);  // (cap is a multiple of 4: 32-byte aligned)
    h->nn_cache_cap = cap;
  }
  IcpStateDev st{};
  memcpy(st.T, init, sizeof(double) * 16);
  *h->h_state = st;
  if (upload_state) HIP_TRY(hipMemcpyAsync(h->d_state, h->h_state, sizeof(IcpStateDev), hipMemcpyHostToDevice, h->stream));
  IcpPassArgs a{};
  a.src = src->pts;
  a.first = 0;
  a.count = src->n;
  a.tpts = tgt->spts;
  a.tnrm = tgt->snrm;
  a.grid = tgt->grid;
  a.crop = to_dev(crop);
  const double r = params->max_correspondence_distance;
  a.r2max = r * r;
  a.method = params->method;
  {  // per-term bounds on the sums of |record terms| -> per-term quanta of the exact record sums (split_exact): 2^53 q >= 8 B_k
    const GridDev& g = tgt->grid;
    const double ex = std::max(std::fabs(g.ox), std::fabs(g.ox + g.nx * g.cell)), ey = std::max(std::fabs(g.oy), std::fabs(g.oy + g.ny * g.cell)),
                 ez = std::max(std::fabs(g.oz), std::fabs(g.oz + g.nz * g.cell));
    const double P = std::max(1.0, std::sqrt(ex * ex + ey * ey + ez * ez) + r);  // a matched source point lies within r of the target's box
    const double rr = std::max(r, 1e-3);
    // headroom: 64 ranks of a sharded run may add up ("submap" mode: every rank contributes up to n correspondences).  The number of
    // queries enters as a CONSTANT (2^24, or the next power of two above a larger scan), not as src->n: for the head of a lazy chain
    // that is the exact count or the chain's upper bound, whichever the host happens to know when the registration starts, and a
    // quantum that follows it makes the dropped bits -- below 2^-85 of a term's bound, but the whole last place of J^T r once that sum
    // has converged to nothing -- a matter of timing: two-handle runs of the stream differed from one-handle runs by an ulp of a pose entry
    // in every third run (round 6, scripts/debug_wobble.py), the two-kernel form, which waits for the count, never did.  (The other half
    // of the same story is in the kernels: the queries are dealt out over the exact number the DEVICE holds, icp_kernels.hpp deal_count.)
    size_t n_for_quantum = (size_t)1 << 24;
    while (n_for_quantum < src->n) n_for_quantum <<= 1;
    const double nn = 64.0 * (double)n_for_quantum;
    double bound[kRec];
    if (params->method == O3DS_ICP_GENERALIZED) {
      const double gw = 4.0 * std::max(1.0, 0.5 / h->gicp_epsilon);  // |M^-1| <= 1 / (2 eps)
      int k = 0;
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) bound[k++] = gw * (i < 3 ? P : 1.0) * (j < 3 ? P : 1.0);  // A^T M^-1 A, A = [-[p]x | I]
      for (int i = 0; i < 6; ++i) bound[21 + i] = gw * (i < 3 ? P : 1.0) * rr;                 // A^T M^-1 d
      bound[27] = gw * rr * rr;
      bound[28] = 1.0;
      bound[29] = rr * rr;
      bound[30] = bound[31] = 1.0;
    } else {
      // the record terms are products of two per-query slots; slot magnitudes per method (icp_kernels.hpp, kTermA/B tables)
      double slot[10];
      const unsigned char *ta, *tb;
      static const unsigned char A0[kRec] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9};
      static const unsigned char B0[kRec] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7, 7, 9, 9};
      static const unsigned char A1[kRec] = {3, 3, 3, 4, 4, 4, 5, 5, 5, 0, 1, 2, 3, 4, 5, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 8, 9, 9};
      static const unsigned char B1[kRec] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 7, 7, 7, 7, 7, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 7, 9, 9};
      if (params->method == O3DS_ICP_POINT_TO_POINT) {  // also the information-matrix pass (same slots, q q^T terms bounded by p q^T's)
        for (int i = 0; i < 6; ++i) slot[i] = P;  // p, q
        slot[6] = 0.0, slot[7] = 1.0, slot[8] = rr * rr, slot[9] = 0.0;
        ta = A1, tb = B1;
      } else {
        for (int i = 0; i < 3; ++i) slot[i] = P, slot[3 + i] = 1.0;  // J = [p x n ; n], unit normals
        slot[6] = rr, slot[7] = 1.0, slot[8] = rr * rr, slot[9] = 0.0;
        ta = A0, tb = B0;
      }
      for (int k = 0; k < kRec; ++k) bound[k] = std::max(slot[ta[k]] * slot[tb[k]], 1e-30);
      if (params->method == O3DS_ICP_POINT_TO_POINT)
        for (int k = 0; k < 9; ++k) bound[k] = P * P;  // information matrix: terms 0..8 are q_a q_b or q_a (<= P^2 either way)
    }
    const bool no_split = ab_getenv("O3DS_SUM_NO_SPLIT") != nullptr;  // diagnostic: plain f64 sums
    for (int k = 0; k < kRec; ++k) {
      int e = 0;
      (void)std::frexp(nn * bound[k], &e);                      // n * bound < 2^e
      a.q_hi[k] = no_split ? 0.0 : std::ldexp(1.0, e + 3 - 53);  // 2^53 q = 8 * 2^e
    }
  }
  a.kmax = std::max(1, (int)std::ceil(r / tgt->grid.cell));  // a neighbour within r is at most this many cells away
  a.nn_cache = h->d_nn_cache;
  a.set_pos = h->fused && h->sets ? h->d_set_pos : nullptr;
  a.set_ref = h->d_set_ref;
  a.set_gain = h->set_gain;
  a.set_min = h->set_min;
  a.set_cap = h->set_cap;
  a.n_tgt = (int)(tgt->index_positions ? tgt->index_positions : tgt->n);
  a.snrm = src->nrm;
  a.count_dev = src->lazy_slot >= 0 ? cnt_word(h, src->lazy_slot) : nullptr;
  a.gicp_k = 1.0 - h->gicp_epsilon;
  a.state = h->d_state;
  a.partials = h->d_partials;
  a.debug = ab_getenv("O3DS_DEBUG_ACC") ? atoi(ab_getenv("O3DS_DEBUG_ACC")) : 0;
  h->pass = a;
  h->params = *params;
  h->session = true;
  h->session_n_src = src->n;
  h->session_launches = 0;
  h->session_state = nullptr;
  h->session_precision = src->precision;
  h->session_crop = crop && (crop->kind != O3DS_CROP_NONE || crop->invert);
  h->session_method = params->method;
  return O3DS_OK;
}

void copy_result(o3ds_handle h, o3ds_icp_result* out) {
  if (out) {
    memcpy(out->transformation, h->h_state->T, sizeof(double) * 16);
    out->fitness = h->h_state->fitness;
    out->inlier_rmse = h->h_state->rmse;
    out->iterations = h->h_state->iterations;
    out->converged = h->h_state->converged;
    out->n_corr = h->h_state->n_corr;
  }
}

int read_state(o3ds_handle h, o3ds_icp_result* out, const IcpStateDev* d_from = nullptr) {
  HIP_TRY(hipMemcpyAsync(h->h_state, d_from ? d_from : h->d_state, sizeof(IcpStateDev), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (out) {
    memcpy(out->transformation, h->h_state->T, sizeof(double) * 16);
    out->fitness = h->h_state->fitness;
    out->inlier_rmse = h->h_state->rmse;
    out->iterations = h->h_state->iterations;
    out->converged = h->h_state->converged;
    out->n_corr = h->h_state->n_corr;
  }
  return O3DS_OK;
}

}  // namespace

// =================================================================================================

namespace {
int gicp_knn_normals(o3ds_handle h, CloudRec& c);  // defined after normals_t
}

extern "C" {

const char* o3ds_version(void) {
#ifdef O3DS_AB_SWITCHES
  return "o3ds_backend 0.2 (gfx950, hip, f32/f64 storage, f64 accumulate; A/B switches compiled in)";
#else
  return "o3ds_backend 0.2 (gfx950, hip, f32/f64 storage, f64 accumulate)";
#endif
}

const char* o3ds_last_error(o3ds_handle h) { return h ? h->err.c_str() : g_thread_error.c_str(); }

int o3ds_create(int device_id, o3ds_handle* out) {
  if (!out) return fail(nullptr, O3DS_ERR_INVALID_ARG, "o3ds_create: null out");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, O3DS_ERR_HIP, std::string("o3ds_create: no HIP device (") + hipGetErrorString(e) + ")");
  if (device_id < 0 || device_id >= count) return fail(nullptr, O3DS_ERR_INVALID_ARG, "o3ds_create: bad device id");
  o3ds_handle h = new o3ds_context();
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      (h->own_stream = h->stream, false) || hipMalloc(&h->d_partials, sizeof(double) * kRec * kMaxPassBlocks) != hipSuccess ||
      hipMalloc(&h->d_state, sizeof(IcpStateDev)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_state, sizeof(IcpStateDev), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->h_state_dev, h->h_state, 0) != hipSuccess ||
      hipHostMalloc((void**)&h->h_pin, kPinBytes, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->h_pin_dev, h->h_pin, 0) != hipSuccess) {
    o3ds_destroy(h);  // frees whatever part of the context exists (every member is null-checked there)
    return fail(nullptr, O3DS_ERR_HIP, "o3ds_create: device initialisation failed");
  }
  {
    if (hipMalloc((void**)&h->d_fused, kFusedBytes) != hipSuccess || hipMemset(h->d_fused, 0, kFusedBytes) != hipSuccess) {
      o3ds_destroy(h);
      return fail(nullptr, O3DS_ERR_OOM, "o3ds_create: scratch allocation failed");
    }
    if (const char* e = ab_getenv("O3DS_ICP_SETS")) h->sets = atoi(e) != 0;
    if (const char* e = ab_getenv("O3DS_SET_GAIN")) h->set_gain = (float)atof(e);
    if (const char* e = ab_getenv("O3DS_SET_MIN")) h->set_min = (float)atof(e);
    if (const char* e = ab_getenv("O3DS_SET_CAP")) h->set_cap = (float)atof(e);
    if (const char* e = ab_getenv("O3DS_ICP_MODE")) {
      h->fused = std::string(e) != "launch";
    }
  }
  {
    static_assert(sizeof(o3ds_context::PinRec) == 64, "one pinned record per cache line");
    if (hipHostMalloc((void**)&h->h_rec, sizeof(o3ds_context::PinRec) * kPinRecs, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->h_rec_dev, h->h_rec, 0) != hipSuccess ||
        hipMalloc((void**)&h->d_cnt, sizeof(int) * 16 * kPinRecs) != hipSuccess || hipMemset(h->d_cnt, 0, sizeof(int) * 16 * kPinRecs) != hipSuccess ||
        hipMalloc((void**)&h->d_ticket, 64) != hipSuccess || hipMemset(h->d_ticket, 0, 64) != hipSuccess ||
        hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) {
      o3ds_destroy(h);
      return fail(nullptr, O3DS_ERR_OOM, "o3ds_create: record allocation failed");
    }
    memset(h->h_rec, 0, sizeof(o3ds_context::PinRec) * kPinRecs);
  }
  if (const char* e = ab_getenv("O3DS_DEBUG_UPDATE")) h->debug_update = atoi(e);
  if (const char* e = ab_getenv("O3DS_PASS_ROWS")) h->pass_rows = std::min(std::max(atoi(e), 1), kMaxPassBlocks);
  *out = h;
  return O3DS_OK;
}

int o3ds_destroy(o3ds_handle h) {
  CHECK_HANDLE(h);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamSynchronize(h->own_stream);
  if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  for (auto& kv : h->clouds) free_cloud(h, kv.second);
  for (auto& kv : h->dense_maps) dense_release(h, kv.second);
  for (auto& b : h->arena_blocks) dev_free(h, b.first);
  (void)hipStreamSynchronize(h->stream);
  dev_release_all(h);
  for (auto& b : h->ingest_bufs) {
    if (b.pts) (void)hipFree(b.pts);
    if (b.box) (void)hipFree(b.box);
    if (b.freed) (void)hipEventDestroy(b.freed);
  }
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->d_raw) (void)hipFree(h->d_raw);
  if (h->d_tiles) (void)hipFree(h->d_tiles);
  if (h->d_ticket) (void)hipFree(h->d_ticket);
  if (h->d_cnt) (void)hipFree(h->d_cnt);
  if (h->h_rec) (void)hipHostFree(h->h_rec);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  (void)o3ds_comm_destroy(h);
  if (h->d_shard_sums) (void)hipFree(h->d_shard_sums);
  if (h->d_shard_keys) (void)hipFree(h->d_shard_keys);
  if (h->d_fused) (void)hipFree(h->d_fused);
  if (h->d_nn_cache) (void)hipFree(h->d_nn_cache);
  if (h->d_set_pos) (void)hipFree(h->d_set_pos);
  if (h->d_cells) (void)hipFree(h->d_cells);
  if (h->d_voxtab) (void)hipFree(h->d_voxtab);
  if (h->d_draw) (void)hipFree(h->d_draw);
  if (h->d_partials) (void)hipFree(h->d_partials);
  if (h->d_state) (void)hipFree(h->d_state);
  if (h->h_state) (void)hipHostFree(h->h_state);
  if (h->h_pin) (void)hipHostFree(h->h_pin);
  for (int k = 0; k < 2; ++k) {
    if (h->h_stage[k]) (void)hipHostFree(h->h_stage[k]);
    if (h->stage_ev[k]) (void)hipEventDestroy(h->stage_ev[k]);
  }
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  for (auto& v : h->span_ev)
    for (hipEvent_t e : v) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return O3DS_OK;
}

int o3ds_set_precision(o3ds_handle h, int precision) {
  CHECK_HANDLE(h);
  if (precision != O3DS_PRECISION_F32 && precision != O3DS_PRECISION_F64) return fail(h, O3DS_ERR_INVALID_ARG, "bad precision");
  h->precision = precision;
  return O3DS_OK;
}

int o3ds_synchronize(o3ds_handle h) {
  CHECK_HANDLE(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return O3DS_OK;
}

void* o3ds_stream(o3ds_handle h) { return h ? (void*)h->stream : nullptr; }

int o3ds_set_stream(o3ds_handle h, void* hip_stream) {
  CHECK_HANDLE(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return O3DS_OK;
}

int o3ds_profile_enable(o3ds_handle h, int on) {
  CHECK_HANDLE(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->profiling = on == 2 ? 2 : (on != 0);
  h->ev_used = 0;
  for (auto& u : h->span_used) u = 0;
  return O3DS_OK;
}

int o3ds_profile_span(o3ds_handle h, int tag, int end) {
  CHECK_HANDLE(h);
  if (tag < 0 || tag >= 8) return fail(h, O3DS_ERR_INVALID_ARG, "profile_span: caller tags are 0..7");
  if ((int)(h->span_used[tag] & 1) != (end ? 1 : 0)) return fail(h, O3DS_ERR_INVALID_ARG, "profile_span: begin / end out of order");
  span_mark(h, tag);
  return O3DS_OK;
}

int o3ds_profile_span_read(o3ds_handle h, int tag, uint64_t* n_spans, double* total_ms) {
  CHECK_HANDLE(h);
  if (tag < 0 || tag >= 16) return fail(h, O3DS_ERR_INVALID_ARG, "profile_span_read: bad tag");
  HIP_TRY(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  const size_t used = h->span_used[tag] & ~(size_t)1;
  for (size_t i = 0; i + 1 < used; i += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->span_ev[tag][i], h->span_ev[tag][i + 1]));
    tot += (double)ms;
  }
  if (n_spans) *n_spans = used / 2;
  if (total_ms) *total_ms = tot;
  h->span_used[tag] = 0;
  return O3DS_OK;
}

int o3ds_profile_read(o3ds_handle h, uint64_t* n_launches, double* total_ms) {
  CHECK_HANDLE(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    tot += (double)ms;
  }
  if (n_launches) *n_launches = h->ev_used / 2;
  if (total_ms) *total_ms = tot;
  h->ev_used = 0;
  return O3DS_OK;
}

// ---- clouds ------------------------------------------------------------------------------------
int o3ds_cloud_upload(o3ds_handle h, const double* xyz, const double* normals, size_t n, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!out || (n > 0 && !xyz)) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_upload: null argument");
  if (n > 0x7fffffffull) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_upload: more than 2^31-1 points");
  HIP_TRY(hipSetDevice(h->device));
  CloudRec c;
  int rc = h->precision == O3DS_PRECISION_F64 ? upload_t<P4d>(h, xyz, normals, n, c) : upload_t<P4f>(h, xyz, normals, n, c);
  if (rc) {
    free_cloud(h, c);
    return rc;
  }
  *out = add_cloud(h, std::move(c));
  return O3DS_OK;
}

int o3ds_cloud_upload_f32(o3ds_handle h, const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                          o3ds_cloud* out) {
  CHECK_HANDLE(h);
  if (!out || (n > 0 && !data)) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_upload_f32: null argument");
  if (n > 0x7fffffffull) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_upload_f32: more than 2^31-1 points");
  if (point_step < 12 || off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step)
    return fail(h, O3DS_ERR_INVALID_ARG, "cloud_upload_f32: x/y/z fields do not fit the point step");
  HIP_TRY(hipSetDevice(h->device));
  CloudRec c;
  CloudGuard c_guard(h, c);
  c.n = n;
  c.precision = h->precision;
  if (n > 0) {
    // The whole ingest runs on the handle's COPY stream -- DMA of the records, unpack, box -- and the call returns with it queued: the
    // scan crosses PCIe while the frame before it is still being registered and merged on the handle's stream, whose first use of the
    // cloud waits for an event (cloud_ready), not the host.  The points live in an ingest buffer (free_points).
    const size_t psz = p4_size(h->precision), bytes = n * point_step;
    // a buffer nothing on the handle's stream still reads (its "freed" event has passed): the ingest must not queue behind the frame
    // that used the buffer last -- that frame is exactly what it is meant to overlap with.  Up to four buffers before one is waited for.
    int bi = -1, idle = -1;
    for (size_t k = 0; k < h->ingest_bufs.size(); ++k) {
      o3ds_context::IngestBuf& cand = h->ingest_bufs[k];
      if (cand.in_use) continue;
      if (idle < 0) idle = (int)k;
      if (cand.pts_bytes >= psz * n && (!cand.freed || hipEventQuery(cand.freed) == hipSuccess)) {
        bi = (int)k;
        break;
      }
    }
    (void)hipGetLastError();  // (hipErrorNotReady of a pending event is not an error)
    if (bi < 0) {
      if (h->ingest_bufs.size() < 4 || idle < 0) {
        h->ingest_bufs.emplace_back();
        bi = (int)h->ingest_bufs.size() - 1;
      } else {
        bi = idle;  // too small (grown below) or still being read (waited for on the copy stream)
      }
    }
    o3ds_context::IngestBuf& b = h->ingest_bufs[(size_t)bi];
    if (b.pts_bytes < psz * n) {
      if (b.pts) {
        if (b.freed) HIP_TRY(hipEventSynchronize(b.freed));
        HIP_TRY(hipStreamSynchronize(h->copy_stream));
        (void)hipFree(b.pts);
        b.pts = nullptr;
        b.pts_bytes = 0;
      }
      const size_t want = psz * (n + n / 8);
      if (hipMalloc(&b.pts, want) != hipSuccess) return fail(h, O3DS_ERR_OOM, "cloud_upload_f32: ingest buffer allocation failed");
      b.pts_bytes = want;
    }
    if (!b.box) {
      if (hipMalloc((void**)&b.box, 64) != hipSuccess) return fail(h, O3DS_ERR_OOM, "cloud_upload_f32: box record allocation failed");
      unsigned long long arm[8] = {0};
      for (int a = 0; a < 6; ++a) arm[a] = order_bits(a < 3 ? 1e300 : -1e300);
      HIP_TRY(hipMemcpy(b.box, arm, sizeof(arm), hipMemcpyHostToDevice));
    }
    if (h->raw_cap < bytes) {
      HIP_TRY(hipStreamSynchronize(h->copy_stream));
      if (h->d_raw) (void)hipFree(h->d_raw);
      h->d_raw = nullptr;
      h->raw_cap = 0;
      if (hipMalloc((void**)&h->d_raw, bytes + bytes / 8) != hipSuccess) return fail(h, O3DS_ERR_OOM, "cloud_upload_f32: staging allocation failed");
      h->raw_cap = bytes + bytes / 8;
    }
    if (b.freed) HIP_TRY(hipStreamWaitEvent(h->copy_stream, b.freed, 0));  // whatever still read the buffer's last cloud on the handle's stream
    // the records: straight DMA from memory the runtime knows as pinned (o3ds_pinned_alloc, hipHostRegister: read asynchronously, see the
    // header), through the handle's pinned ring otherwise (the caller's buffer is consumed when this returns)
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, data) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();  // (a pageable pointer makes the query fail: not an error of ours)
    if (pinned) {
      HIP_TRY(hipMemcpyAsync(h->d_raw, data, bytes, hipMemcpyHostToDevice, h->copy_stream));
    } else {
      const int rcc = h2d_copy(h, h->d_raw, data, bytes, h->copy_stream);
      if (rcc) return rcc;
    }
    b.in_use = true;
    c.pts = b.pts;
    c.ingest_buf = bi;
    // the box of the points inside the volume the handle last cropped with rides on the unpack (pack_strided_f32_box_kernel)
    const int box_slot = h->pre_crop_valid ? take_rec(h) : -1;  // (-1: every record is held by a map -- the box is reduced when it is asked for)
    if (box_slot >= 0) {
      c.pre_slot = box_slot;
      c.pre_seq = ++h->rec_seq;
      c.pre_crop = h->pre_crop;
      o3ds_context::PinRec* rec = h->h_rec_dev + c.pre_slot;
      if (h->precision == O3DS_PRECISION_F64)
        pack_strided_f32_box_kernel<P4d><<<grid_for(n, 1024), kBlock, 0, h->copy_stream>>>(h->d_raw, n, point_step, off_x, off_y, off_z, (P4d*)c.pts, c.pre_crop, b.box);
      else
        pack_strided_f32_box_kernel<P4f><<<grid_for(n, 1024), kBlock, 0, h->copy_stream>>>(h->d_raw, n, point_step, off_x, off_y, off_z, (P4f*)c.pts, c.pre_crop, b.box);
      box_publish_kernel<<<1, 64, 0, h->copy_stream>>>(b.box, rec->box, &rec->seq, c.pre_seq);
    } else if (h->precision == O3DS_PRECISION_F64) {
      pack_strided_f32_kernel<P4d><<<grid_for(n), kBlock, 0, h->copy_stream>>>(h->d_raw, n, point_step, off_x, off_y, off_z, (P4d*)c.pts);
    } else {
      pack_strided_f32_kernel<P4f><<<grid_for(n), kBlock, 0, h->copy_stream>>>(h->d_raw, n, point_step, off_x, off_y, off_z, (P4f*)c.pts);
    }
    HIP_TRY(hipGetLastError());
    c.ingest_ev = take_event(h);
    if (!c.ingest_ev) return fail(h, O3DS_ERR_HIP, "cloud_upload_f32: event creation failed");
    HIP_TRY(hipEventRecord(c.ingest_ev, h->copy_stream));
  }
  c_guard.release();
  *out = add_cloud(h, std::move(c));
  return O3DS_OK;
}

int o3ds_pinned_alloc(o3ds_handle h, size_t bytes, void** out) {
  CHECK_HANDLE(h);
  if (!out) return fail(h, O3DS_ERR_INVALID_ARG, "pinned_alloc: null out");
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return O3DS_OK;
}
int o3ds_pinned_free(o3ds_handle h, void* p) {
  CHECK_HANDLE(h);
  if (h->copy_stream) HIP_TRY(hipStreamSynchronize(h->copy_stream));  // an ingest may still be reading it
  if (p) HIP_TRY(hipHostFree(p));
  return O3DS_OK;
}
int o3ds_cloud_wait_ingest(o3ds_handle h, o3ds_cloud id) {
  CHECK_HANDLE(h);
  auto it = h->clouds.find(id);
  if (it == h->clouds.end()) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_wait_ingest: unknown cloud id");
  if (it->second.ingest_ev) HIP_TRY(hipEventSynchronize(it->second.ingest_ev));
  return O3DS_OK;
}

int o3ds_cloud_free(o3ds_handle h, o3ds_cloud id) {
  CHECK_HANDLE(h);
  auto it = h->clouds.find(id);
  if (it == h->clouds.end()) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_free: unknown cloud id");
  // no host synchronisation: the buffers go back to the handle's allocator behind whatever still reads them on this stream
  // (o3ds_set_stream synchronises the stream it leaves).  O3DS_SYNC_ON_FREE=1 restores the wait (six of them per lidar frame).
  static const bool sync_on_free = ab_getenv("O3DS_SYNC_ON_FREE") != nullptr;
  if (sync_on_free) (void)hipStreamSynchronize(h->stream);
  free_cloud(h, it->second);
  h->clouds.erase(it);
  return O3DS_OK;
}

int o3ds_cloud_size(o3ds_handle h, o3ds_cloud id, size_t* n, int* has_normals) {
  CHECK_HANDLE(h);
  CloudRec* c = find_cloud_lazy(h, id);  // (whether a cloud has normals is known without waiting for its size)
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_size: unknown cloud id");
  if (n && c->pm) {  // a submap in its persistent form stays in it: its live points = slots in use - dead slots, from the device's counters
    int counters[kPmCounters];
    ArenaScope arena_scope(h);
    const int rb = read_back(h, {{counters, c->pm->dev.counters, sizeof(counters)}});
    if (rb) return rb;
    *n = (size_t)(counters[kPmN] - counters[kPmDeadCnt]);
    if (has_normals) *has_normals = c->nrm != nullptr;
    return O3DS_OK;
  }
  if (n) {
    const int rr = resolve_count(h, *c, true);
    if (rr) return rr;
  }
  if (n) *n = c->n;
  if (has_normals) *has_normals = c->nrm != nullptr;
  return O3DS_OK;
}

int o3ds_cloud_size_bound(o3ds_handle h, o3ds_cloud id, size_t* lower, size_t* upper) {
  CHECK_HANDLE(h);
  CloudRec* c = find_cloud_lazy(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_size_bound: unknown cloud id");
  if (c->pm) {
    (void)pm_poll(h, c->pm, false);
    if (lower) *lower = c->pm->live_lower;
    if (upper) *upper = c->pm->n_upper;
    return O3DS_OK;
  }
  if (lower) *lower = c->lazy_slot >= 0 ? std::min(c->n_lower, c->n) : c->n;
  if (upper) *upper = c->n;
  return O3DS_OK;
}

int o3ds_cloud_download(o3ds_handle h, o3ds_cloud id, double* xyz, double* normals, size_t capacity) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download: unknown cloud id");
  if (capacity < c->n) return fail(h, O3DS_ERR_CAPACITY, "cloud_download: capacity < cloud size");
  return c->precision == O3DS_PRECISION_F64 ? download_t<P4d>(h, *c, xyz, normals) : download_t<P4f>(h, *c, xyz, normals);
}

int o3ds_cloud_download_f32(o3ds_handle h, o3ds_cloud id, void* data, size_t capacity, size_t point_step, size_t off_x, size_t off_y,
                            size_t off_z, size_t off_normal, size_t off_rgb, int rgb_rounding) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download_f32: unknown cloud id");
  if (capacity < c->n) return fail(h, O3DS_ERR_CAPACITY, "cloud_download_f32: capacity < cloud size");
  if (c->n > 0 && !data) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download_f32: null buffer");
  if (point_step < 12 || off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step)
    return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download_f32: x/y/z fields do not fit the point step");
  if (off_normal != O3DS_NO_FIELD) {
    if (off_normal + 12 > point_step) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download_f32: normal fields do not fit the point step");
    if (!c->nrm) return fail(h, O3DS_ERR_NO_NORMALS, "cloud_download_f32: normals requested but the cloud has none");
  }
  if (off_rgb != O3DS_NO_FIELD) {
    if (off_rgb + 4 > point_step) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download_f32: rgb field does not fit the point step");
    if (!c->col) return fail(h, O3DS_ERR_EMPTY, "cloud_download_f32: rgb requested but the cloud has no colours");
    if (rgb_rounding != 0 && rgb_rounding != 1) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_download_f32: rgb_rounding must be 0 or 1");
  }
  if (c->n == 0) return O3DS_OK;
  HIP_TRY(hipSetDevice(h->device));
  unsigned char* d_raw = nullptr;
  const size_t bytes = c->n * point_step;
  TMP_ALLOC(d_raw, bytes);
  HIP_TRY(hipMemsetAsync(d_raw, 0, bytes, h->stream));  // padding and the fields this path does not carry
  const size_t on = off_normal == O3DS_NO_FIELD ? kNoField : off_normal;
  const size_t oc = off_rgb == O3DS_NO_FIELD ? kNoField : off_rgb;
  if (c->precision == O3DS_PRECISION_F64)
    unpack_strided_f32_kernel<P4d><<<grid_for(c->n), kBlock, 0, h->stream>>>((const P4d*)c->pts, (const P4d*)c->nrm, (const P4d*)c->col, c->n,
                                                                            point_step, off_x, off_y, off_z, on, oc, rgb_rounding, d_raw);
  else
    unpack_strided_f32_kernel<P4f><<<grid_for(c->n), kBlock, 0, h->stream>>>((const P4f*)c->pts, (const P4f*)c->nrm, (const P4f*)c->col, c->n,
                                                                            point_step, off_x, off_y, off_z, on, oc, rgb_rounding, d_raw);
  HIP_TRY(hipGetLastError());
  return d2h_copy(h, data, d_raw, bytes);
}

int o3ds_cloud_set_colors_from_records(o3ds_handle h, o3ds_cloud id, const void* data, size_t point_step, size_t off_field, int kind) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_set_colors_from_records: unknown cloud id");
  if (kind != O3DS_COLOR_FIELD_RGB && kind != O3DS_COLOR_FIELD_INTENSITY)
    return fail(h, O3DS_ERR_INVALID_ARG, "cloud_set_colors_from_records: unknown field kind");
  if (off_field + 4 > point_step) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_set_colors_from_records: the field does not fit the point step");
  if (c->n > 0 && !data) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_set_colors_from_records: null buffer");
  HIP_TRY(hipSetDevice(h->device));
  if (c->col) dev_free(h, c->col);
  c->col = nullptr;
  if (c->n == 0) return O3DS_OK;
  unsigned char* d_raw = nullptr;
  TMP_ALLOC(d_raw, c->n * point_step);
  {
    const int rcc = h2d_copy(h, d_raw, data, c->n * point_step);
    if (rcc) return rcc;
  }
  HIP_TRY(dev_alloc(h, (void**)&c->col, p4_size(c->precision) * c->n));
  if (c->precision == O3DS_PRECISION_F64)
    colors_from_records_kernel<P4d><<<grid_for(c->n), kBlock, 0, h->stream>>>(d_raw, c->n, point_step, off_field, kind, (P4d*)c->col);
  else
    colors_from_records_kernel<P4f><<<grid_for(c->n), kBlock, 0, h->stream>>>(d_raw, c->n, point_step, off_field, kind, (P4f*)c->col);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));  // `data` may be reused by the caller
  return O3DS_OK;
}

int o3ds_cloud_set_colors(o3ds_handle h, o3ds_cloud id, const double* rgb) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_set_colors: unknown cloud id");
  HIP_TRY(hipSetDevice(h->device));
  if (c->col) dev_free(h, c->col);
  c->col = nullptr;
  if (!rgb || c->n == 0) return O3DS_OK;  // colors_.clear()
  double* stage = nullptr;
  TMP_ALLOC(stage, sizeof(double) * 3 * c->n);
  {
    const int rcc = h2d_copy(h, stage, rgb, sizeof(double) * 3 * c->n);
    if (rcc) return rcc;
  }
  HIP_TRY(dev_alloc(h, (void**)&c->col, p4_size(c->precision) * c->n));
  if (c->precision == O3DS_PRECISION_F64)
    pack_kernel<P4d><<<grid_for(c->n), kBlock, 0, h->stream>>>(stage, c->n, (P4d*)c->col);
  else
    pack_kernel<P4f><<<grid_for(c->n), kBlock, 0, h->stream>>>(stage, c->n, (P4f*)c->col);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));  // rgb may be released by the caller
  return O3DS_OK;
}

int o3ds_cloud_has_colors(o3ds_handle h, o3ds_cloud id, int* has_colors) {
  CHECK_HANDLE(h);
  CloudRec* c = find_cloud(h, id);
  if (!c || !has_colors) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_has_colors: bad argument");
  *has_colors = c->col != nullptr;
  return O3DS_OK;
}

int o3ds_cloud_get_colors(o3ds_handle h, o3ds_cloud id, double* rgb, size_t capacity) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, id);
  if (!c || !rgb) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_get_colors: bad argument");
  if (!c->col) return fail(h, O3DS_ERR_EMPTY, "cloud_get_colors: the cloud has no colours");
  if (capacity < c->n) return fail(h, O3DS_ERR_CAPACITY, "cloud_get_colors: capacity < cloud size");
  double* stage = nullptr;
  TMP_ALLOC(stage, sizeof(double) * 3 * c->n);
  if (c->precision == O3DS_PRECISION_F64)
    unpack_kernel<P4d><<<grid_for(c->n), kBlock, 0, h->stream>>>((const P4d*)c->col, c->n, stage);
  else
    unpack_kernel<P4f><<<grid_for(c->n), kBlock, 0, h->stream>>>((const P4f*)c->col, c->n, stage);
  HIP_TRY(hipGetLastError());
  return d2h_copy(h, rgb, stage, sizeof(double) * 3 * c->n);
}

int o3ds_cloud_build_index(o3ds_handle h, o3ds_cloud id, double max_corr_hint, double cell_size) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "build_index: unknown cloud id");
  double cell = cell_size > 0.0 ? cell_size : max_corr_hint / index_cell_div();
  if (!(cell > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "build_index: need cell_size > 0 or max_corr_hint > 0");
  int rc = build_index(h, *c, cell);
  return rc;
}

// ---- ICP ---------------------------------------------------------------------------------------
int o3ds_icp_begin(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                   const o3ds_icp_params* params) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!init) return fail(h, O3DS_ERR_INVALID_ARG, "icp_begin: null init");
  (void)find_cloud(h, source);  // the step-wise forms address source RANGES: they take the exact size
  return begin_session(h, source, target, target_crop, init, params);
}

int o3ds_icp_accumulate(o3ds_handle h, size_t first, size_t count, double* d_record) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate: no session (call o3ds_icp_begin)");
  if (!d_record) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate: null record");
  if (first + count > h->session_n_src) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate: range outside source");
  IcpPassArgs a = h->pass;
  a.first = first;
  a.count = count;
  {
    const int nb = pass_blocks(h, count);
    if (h->session_precision == O3DS_PRECISION_F64)
      launch_accumulate<P4d>(h, a, h->session_crop, nb);
    else
      launch_accumulate<P4f>(h, a, h->session_crop, nb);
    icp_reduce_kernel<<<1, kUpdBlock, 0, h->stream>>>(h->d_partials, nb, h->d_state, d_record, quantum_table(a));
  }
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

// Partitioning B (one map split over the GPUs of a node): search pass -> keys; the caller MIN-all-reduces them; accumulate pass
int o3ds_icp_nn_keys(o3ds_handle h, size_t first, size_t count, int rank, unsigned long long* d_keys) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_nn_keys: no session (call o3ds_icp_begin)");
  if (!d_keys) return fail(h, O3DS_ERR_INVALID_ARG, "icp_nn_keys: null keys");
  if (rank < 0 || rank > 15) return fail(h, O3DS_ERR_INVALID_ARG, "icp_nn_keys: rank must be 0..15 (4 bits of the key)");
  if (first + count > h->session_n_src) return fail(h, O3DS_ERR_INVALID_ARG, "icp_nn_keys: range outside source");
  if ((size_t)h->pass.n_tgt > ((size_t)1 << 28)) return fail(h, O3DS_ERR_INVALID_ARG, "icp_nn_keys: more than 2^28 target points per shard");
  IcpPassArgs a = h->pass;
  a.first = first;
  a.count = count;
  a.keys_mode = 1;
  a.keys_rank = rank;
  a.keys = d_keys;
  const int nb = pass_blocks(h, count);
  if (h->session_precision == O3DS_PRECISION_F64)
    launch_accumulate<P4d>(h, a, h->session_crop, nb);
  else
    launch_accumulate<P4f>(h, a, h->session_crop, nb);
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

int o3ds_icp_accumulate_keys(o3ds_handle h, size_t first, size_t count, int rank, const unsigned long long* d_keys, double* d_record) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate_keys: no session (call o3ds_icp_begin)");
  if (!d_keys || !d_record) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate_keys: null argument");
  if (rank < 0 || rank > 15) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate_keys: rank must be 0..15");
  if (first + count > h->session_n_src) return fail(h, O3DS_ERR_INVALID_ARG, "icp_accumulate_keys: range outside source");
  IcpPassArgs a = h->pass;
  a.first = first;
  a.count = count;
  a.keys_mode = 2;
  a.keys_rank = rank;
  a.keys = const_cast<unsigned long long*>(d_keys);
  const int nb = pass_blocks(h, count);
  if (h->session_precision == O3DS_PRECISION_F64)
    launch_accumulate<P4d>(h, a, h->session_crop, nb);
  else
    launch_accumulate<P4f>(h, a, h->session_crop, nb);
  icp_reduce_kernel<<<1, kUpdBlock, 0, h->stream>>>(h->d_partials, nb, h->d_state, d_record, quantum_table(a));
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

int o3ds_icp_update(o3ds_handle h, const double* d_record, uint64_t n_src_total) {
  CHECK_HANDLE(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_update: no session");
  if (!d_record) return fail(h, O3DS_ERR_INVALID_ARG, "icp_update: null record");
  icp_update_kernel<<<1, 128, 0, h->stream>>>(d_record, h->d_state, (unsigned long long)n_src_total, h->params.max_iteration,
                                             h->params.relative_fitness, h->params.relative_rmse, h->session_method);
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

int o3ds_icp_pass(o3ds_handle h, size_t first, size_t count, size_t n_src_total, const double* d_sums_in, double* d_sums_out,
                  double* d_sums_next) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_pass: no session (call o3ds_icp_begin)");
  if (!d_sums_out || !d_sums_next || (h->session_launches > 0 && !d_sums_in)) return fail(h, O3DS_ERR_INVALID_ARG, "icp_pass: null sums buffer");
  if (first + count > h->session_n_src) return fail(h, O3DS_ERR_INVALID_ARG, "icp_pass: range outside source");
  if (count > kFusedMaxQueries)
    return fail(h, O3DS_ERR_CAPACITY, "icp_pass: at most O3DS_ICP_PASS_MAX_QUERIES (262144) source points per call: use o3ds_icp_accumulate / o3ds_icp_update for larger shards");
  IcpFusedArgs fa{};
  fa.pass = h->pass;
  fa.pass.first = first;
  fa.pass.count = count;
  fa.n_src_total = (unsigned long long)n_src_total;
  fa.max_iter = h->params.max_iteration;
  fa.rel_fitness = h->params.relative_fitness;
  fa.rel_rmse = h->params.relative_rmse;
  fa.init = *h->h_state;  // written by begin_session
  const int j = h->session_launches++;
  fa.first = j == 0;
  fa.pass_index = j;
  fa.state_in = h->session_state ? h->session_state : h->d_state;
  fa.state_out = (IcpStateDev*)(h->d_fused + (size_t)(j & 1) * kFusedStateStride);
  fa.state_host = nullptr;
  fa.slots_in = d_sums_in ? d_sums_in : d_sums_next;  // unused by the first launch
  fa.slots_out = d_sums_out;
  fa.slots_clear = d_sums_next;
  fa.trace = nullptr;
  const int nb = fused_blocks(count);
  if (h->session_precision == O3DS_PRECISION_F64)
    launch_fused<P4d>(h, fa, h->session_crop, nb, true);
  else
    launch_fused<P4f>(h, fa, h->session_crop, nb, true);
  h->session_state = fa.state_out;
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

int o3ds_icp_pass_finish(o3ds_handle h, size_t n_src_total, const double* d_sums_in, double* d_sums_scratch, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_pass_finish: no session");
  if (!out || !d_sums_in || !d_sums_scratch) return fail(h, O3DS_ERR_INVALID_ARG, "icp_pass_finish: null argument");
  if (h->session_launches == 0) return fail(h, O3DS_ERR_INVALID_ARG, "icp_pass_finish: no pass was issued");
  IcpFusedArgs fa{};  // the one-workgroup tail launch: folds the last pass, never reaches the body
  fa.pass = h->pass;
  fa.pass.count = 0;
  fa.n_src_total = (unsigned long long)n_src_total;
  fa.max_iter = h->params.max_iteration;
  fa.rel_fitness = h->params.relative_fitness;
  fa.rel_rmse = h->params.relative_rmse;
  const int j = h->session_launches++;
  fa.first = 0;
  fa.state_in = h->session_state;
  fa.state_out = (IcpStateDev*)(h->d_fused + (size_t)(j & 1) * kFusedStateStride);
  fa.state_host = h->h_state_dev;
  fa.seq_host = pub_slot<unsigned long long>(h, kSeqSlot);
  fa.seq = ++h->fused_seq;
  fa.slots_in = d_sums_in;
  fa.slots_out = d_sums_scratch;  // a launch that still has iterations left would add an (empty) pass here
  fa.slots_clear = d_sums_scratch;
  if (h->session_precision == O3DS_PRECISION_F64)
    launch_fused<P4d>(h, fa, h->session_crop, 1, false);
  else
    launch_fused<P4f>(h, fa, h->session_crop, 1, false);
  HIP_TRY(hipGetLastError());
  HIP_TRY(wait_fused_state(h, fa.seq));
  h->session = false;
  h->session_state = nullptr;
  copy_result(h, out);
  return O3DS_OK;
}

int o3ds_icp_done(o3ds_handle h, int* done) {
  CHECK_HANDLE(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_done: no session");
  int rc = read_state(h, nullptr, h->session_state);
  if (rc) return rc;
  if (done) *done = h->h_state->done;
  return O3DS_OK;
}

int o3ds_icp_finish(o3ds_handle h, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  if (!h->session) return fail(h, O3DS_ERR_INVALID_ARG, "icp_finish: no session");
  if (!out) return fail(h, O3DS_ERR_INVALID_ARG, "icp_finish: null out");
  h->session = false;
  return read_state(h, out);
}

int o3ds_set_gicp_epsilon(o3ds_handle h, double epsilon) {
  CHECK_HANDLE(h);
  if (!(epsilon > 0.0 && epsilon <= 1.0)) return fail(h, O3DS_ERR_INVALID_ARG, "gicp epsilon must be in (0, 1]");
  h->gicp_epsilon = epsilon;
  return O3DS_OK;
}

int o3ds_icp_point_to_plane_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop,
                                const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  if (!params) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null params");
  o3ds_icp_params p = *params;
  p.method = O3DS_ICP_POINT_TO_PLANE;
  return o3ds_icp_register_dev(h, source, target, target_crop, init, &p, out);
}

int o3ds_information_matrix_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double T[16],
                                double max_correspondence_distance, double information[36]) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!T || !information) return fail(h, O3DS_ERR_INVALID_ARG, "information_matrix: null argument");
  o3ds_icp_params p{};
  p.max_correspondence_distance = max_correspondence_distance;
  p.max_iteration = 0;
  p.method = O3DS_ICP_POINT_TO_POINT;  // validation as for point-to-point: no normals needed
  (void)find_cloud(h, source);
  int rc = begin_session(h, source, target, target_crop, T, &p);
  if (rc) return rc;
  h->session = false;
  IcpPassArgs a = h->pass;
  a.method = kMethodInformation;
  double* d_record = nullptr;
  TMP_ALLOC(d_record, sizeof(double) * kRec);
  const int nb = pass_blocks(h, a.count);
  if (h->session_precision == O3DS_PRECISION_F64)
    launch_accumulate<P4d>(h, a, h->session_crop, nb);
  else
    launch_accumulate<P4f>(h, a, h->session_crop, nb);
  icp_reduce_kernel<<<1, kUpdBlock, 0, h->stream>>>(h->d_partials, nb, h->d_state, d_record, quantum_table(a));
  HIP_TRY(hipGetLastError());
  double rec[kRec];
  rc = read_back(h, {{rec, d_record, sizeof(rec)}});
  if (rc) return rc;
  // Lambda = [[ |q|^2 I - q q^T , [q]x ], [ [q]x^T , I ]] summed over the matched target points
  const double xx = rec[0], xy = rec[1], xz = rec[2], yy = rec[3], yz = rec[4], zz = rec[5], sx = rec[6], sy = rec[7], sz = rec[8];
  const double m = rec[kRecCount];
  const double L[6][6] = {{yy + zz, -xy, -xz, 0.0, -sz, sy}, {-xy, xx + zz, -yz, sz, 0.0, -sx}, {-xz, -yz, xx + yy, -sy, sx, 0.0},
                          {0.0, sz, -sy, m, 0.0, 0.0},       {-sz, 0.0, sx, 0.0, m, 0.0},      {sy, -sx, 0.0, 0.0, 0.0, m}};
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) information[r * 6 + c] = L[r][c];
  return O3DS_OK;
}

int o3ds_information_matrix(o3ds_handle h, const double* src_xyz, size_t n_src, const double* tgt_xyz, size_t n_tgt, const double T[16],
                            double max_correspondence_distance, double information[36]) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!T || !information) return fail(h, O3DS_ERR_INVALID_ARG, "information_matrix: null argument");
  if (!(max_correspondence_distance > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");
  if (n_tgt == 0) return fail(h, O3DS_ERR_EMPTY, "information_matrix: empty target");
  o3ds_cloud s = 0, t = 0;
  int rc = o3ds_cloud_upload(h, src_xyz, nullptr, n_src, &s);
  if (rc) return rc;
  rc = o3ds_cloud_upload(h, tgt_xyz, nullptr, n_tgt, &t);
  if (!rc) rc = o3ds_information_matrix_dev(h, s, t, nullptr, T, max_correspondence_distance, information);
  const std::string keep = h->err;
  (void)o3ds_cloud_free(h, s);
  if (t) (void)o3ds_cloud_free(h, t);
  if (rc) h->err = keep;
  return rc;
}

int o3ds_icp_point_to_point_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                                const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  if (!params) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null params");
  o3ds_icp_params p = *params;
  p.method = O3DS_ICP_POINT_TO_POINT;
  return o3ds_icp_register_dev(h, source, target, target_crop, init, &p, out);
}

int o3ds_icp_point_to_point(o3ds_handle h, const double* src_xyz, size_t n_src, const double* tgt_xyz, size_t n_tgt, const double init[16],
                            const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!params || !out || !init) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null argument");
  if (!(params->max_correspondence_distance > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");
  if (n_tgt == 0) return fail(h, O3DS_ERR_EMPTY, "icp: empty target (map patch size is zero)");
  o3ds_cloud s = 0, t = 0;
  int rc = o3ds_cloud_upload(h, src_xyz, nullptr, n_src, &s);
  if (rc) return rc;
  rc = o3ds_cloud_upload(h, tgt_xyz, nullptr, n_tgt, &t);
  if (!rc) rc = o3ds_icp_point_to_point_dev(h, s, t, nullptr, init, params, out);
  const std::string keep = h->err;
  (void)o3ds_cloud_free(h, s);
  if (t) (void)o3ds_cloud_free(h, t);
  if (rc) h->err = keep;
  return rc;
}

int o3ds_icp_generalized_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                             const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  if (!params) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null params");
  o3ds_icp_params p = *params;
  p.method = O3DS_ICP_GENERALIZED;
  // [O3D] InitializePointCloudForGeneralizedICP (call site CloudRegistration.cpp:16-21): a cloud that carries no normals gets
  // EstimateNormals(KDTreeSearchParamKNN(20)) on a COPY -- the caller's cloud stays without normals
  o3ds_cloud use[2] = {source, target}, own[2] = {0, 0};
  int rc = O3DS_OK;
  static const double kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int k = 0; k < 2 && !rc; ++k) {
    const CloudRec* c = find_cloud_lazy(h, use[k]);
    if (!c || c->n == 0 || c->nrm) continue;
    rc = o3ds_transform_cloud(h, use[k], kIdentity, &own[k]);  // x * 1 + 0: a bit-exact copy
    if (rc) break;
    use[k] = own[k];
    CloudRec* cc = find_cloud(h, own[k]);
    ArenaScope arena_scope(h);
    rc = gicp_knn_normals(h, *cc);
  }
  if (!rc) rc = o3ds_icp_register_dev(h, use[0], use[1], target_crop, init, &p, out);
  const std::string keep = h->err;
  for (int k = 0; k < 2; ++k)
    if (own[k]) (void)o3ds_cloud_free(h, own[k]);
  if (rc) h->err = keep;
  return rc;
}

int o3ds_icp_generalized(o3ds_handle h, const double* src_xyz, const double* src_normals, size_t n_src, const double* tgt_xyz,
                         const double* tgt_normals, size_t n_tgt, const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!params || !out || !init) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null argument");
  if (!(params->max_correspondence_distance > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");
  if (n_tgt == 0) return fail(h, O3DS_ERR_EMPTY, "icp: empty target (map patch size is zero)");
  o3ds_cloud s = 0, t = 0;
  int rc = o3ds_cloud_upload(h, src_xyz, src_normals, n_src, &s);
  if (rc) return rc;
  rc = o3ds_cloud_upload(h, tgt_xyz, tgt_normals, n_tgt, &t);
  if (!rc) rc = o3ds_icp_generalized_dev(h, s, t, nullptr, init, params, out);
  const std::string keep = h->err;
  (void)o3ds_cloud_free(h, s);
  if (t) (void)o3ds_cloud_free(h, t);
  if (rc) h->err = keep;
  return rc;
}

int o3ds_icp_overlap_next(o3ds_handle h, o3ds_overlap_fn fn, void* arg) {
  CHECK_HANDLE(h);
  h->overlap_fn = fn;
  h->overlap_arg = fn ? arg : nullptr;
  return O3DS_OK;
}

int o3ds_icp_register_dev(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const o3ds_crop* target_crop, const double init[16],
                          const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  struct OverlapOnce {  // the callback belongs to THIS registration: whatever happens, the next one does not inherit it
    o3ds_handle h;
    ~OverlapOnce() { h->overlap_fn = nullptr, h->overlap_arg = nullptr; }
  } overlap_once{h};
  if (!init || !out) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null init/out");
  // the fused loop runs one workgroup per kIcpQ queries; its exact record sums are order-independent for up to 4096 workgroup records
  // (split_exact), so sources beyond kFusedMaxQueries points take the two-launch form (same results, looped pass kernel)
  CloudRec* src_rec = find_cloud_lazy(h, source);
  const bool use_fused = h->fused && src_rec && src_rec->n <= kFusedMaxQueries;
  if (src_rec && !use_fused) {  // the two-launch form hands the source size to its update kernel by value
    const int rr = resolve_count(h, *src_rec, true);
    if (rr) return rr;
  }
  int rc = begin_session(h, source, target, target_crop, init, params, !use_fused);
  if (rc) return rc;
  h->session = false;  // the loop below owns the state
  const IcpPassArgs a = h->pass;
  if (use_fused) {
    // launch j = [tail of pass j-1 in every workgroup's prologue] + pass j; launch max_iter+1 is prologue-only (one workgroup)
    IcpFusedArgs fa{};
    fa.pass = a;
    fa.n_src_total = (unsigned long long)a.count;
    fa.max_iter = params->max_iteration;
    fa.rel_fitness = params->relative_fitness;
    fa.rel_rmse = params->relative_rmse;
    const int nb = fused_blocks(a.count);
    fa.init = *h->h_state;
    const int total = params->max_iteration + 2;
    // O3DS_FUSED_TRACE=<file>: phase timestamps of every workgroup of launch 5 (development aid, see scripts/fused_trace.py)
    const char* trace_path = ab_getenv("O3DS_FUSED_TRACE");
    unsigned long long* d_trace = nullptr;
    const int trace_launch = ab_getenv("O3DS_FUSED_TRACE_LAUNCH") ? atoi(ab_getenv("O3DS_FUSED_TRACE_LAUNCH")) : 5;
    if (trace_path && total > trace_launch + 1) {
      HIP_TRY(hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 16 * nb));
      HIP_TRY(hipMemset(d_trace, 0, sizeof(unsigned long long) * 16 * nb));
    }
    // O3DS_ICP_STATS=1: per-launch counters of how the queries were served (verified from their candidate set / searched / sets left /
    // stage-3 queries), printed to stderr after the registration (development aid)
    static const bool want_stats = ab_getenv("O3DS_ICP_STATS") != nullptr;
    unsigned long long* d_stats = nullptr;
    if (want_stats) {
      HIP_TRY(hipMalloc((void**)&d_stats, sizeof(unsigned long long) * 4 * (size_t)total));
      HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(unsigned long long) * 4 * (size_t)total, h->stream));
    }
    int j = 0;
    const IcpStateDev* last = h->d_state;
    while (j < total) {
      // A launch after the loop has ended costs its launch (~4.6 us each, a dozen of them per registration of a stream whose scans
      // converge in four or five iterations), a look at the state costs a host round trip: queue what the previous registration on this
      // handle needed plus one, then four at a time
      const int chunk = std::min(total - j, j == 0 ? h->fused_chunk_hint[target_crop ? 1 : 0] : 4);
      for (int k = 0; k < chunk; ++k, ++j) {
        const int par = j & 1;
        fa.first = j == 0;
        fa.pass_index = j;
        fa.state_in = last;
        fa.state_out = (IcpStateDev*)(h->d_fused + par * kFusedStateStride);
        // slot buffers rotate with a launch counter that runs across registrations: launch g reads (g-1)%3, adds into g%3, clears (g+1)%3
        const unsigned long long g = h->fused_launches++;
        fa.slots_in = (const double*)(h->d_fused + kFusedSlotsOff + ((g + 2) % 3) * kFusedSlotBufBytes);
        fa.slots_out = (double*)(h->d_fused + kFusedSlotsOff + (g % 3) * kFusedSlotBufBytes);
        fa.slots_clear = (double*)(h->d_fused + kFusedSlotsOff + ((g + 1) % 3) * kFusedSlotBufBytes);
        const bool tail_only = j == total - 1;
        fa.trace = j == trace_launch ? d_trace : nullptr;
        fa.pass.stats = d_stats ? d_stats + 4 * (size_t)j : nullptr;
        fa.state_host = k == chunk - 1 ? h->h_state_dev : nullptr;  // the launch the host waits for also writes the pinned copy
        fa.seq_host = pub_slot<unsigned long long>(h, kSeqSlot);
        if (fa.state_host) fa.seq = ++h->fused_seq;
        if (h->session_precision == O3DS_PRECISION_F64)
          launch_fused<P4d>(h, fa, h->session_crop, tail_only ? 1 : nb, !tail_only);
        else
          launch_fused<P4f>(h, fa, h->session_crop, tail_only ? 1 : nb, !tail_only);
        last = fa.state_out;
      }
      {
        // a failed launch or stream breaks the "slot buffer g % 3 was cleared by launch g - 1" rotation: clear all three and restart
        // the counter, so that the next registration does not add into records that were never cleared
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && h->overlap_fn) {  // o3ds_icp_overlap_next: the caller's work for the time this thread would wait (nested
          const o3ds_overlap_fn fn = h->overlap_fn;  // ABI calls bump-allocate behind this call's temporaries: the arena is reset by the
          void* const arg = h->overlap_arg;          // outermost call only)
          h->overlap_fn = nullptr;
          fn(arg);
          (void)hipSetDevice(h->device);
          e = hipGetLastError();
        }
        if (e == hipSuccess) e = wait_fused_state(h, h->fused_seq);
        if (e != hipSuccess) {
          (void)hipStreamSynchronize(h->stream);
          (void)hipMemset(h->d_fused + kFusedSlotsOff, 0, 3 * kFusedSlotBufBytes);
          h->fused_launches = 0;
          if (d_trace) (void)hipFree(d_trace);
          return fail(h, O3DS_ERR_HIP, std::string("icp (fused loop): ") + hipGetErrorString(e));
        }
      }
      copy_result(h, out);
      if (h->h_state->done) break;
    }
    h->fused_chunk_hint[target_crop ? 1 : 0] = std::min(std::max(h->h_state->iterations + 3, 4), 12);  // iterations + 2 launches were needed
    // (development aids below read device memory with null-stream copies: the handle's stream is non-blocking and the pinned stamp may have
    // been seen while the last launch was still draining)
    if (d_stats || d_trace) (void)hipStreamSynchronize(h->stream);
    if (d_stats) {
      std::vector<unsigned long long> t((size_t)4 * total);
      (void)hipMemcpy(t.data(), d_stats, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipFree(d_stats);
      fprintf(stderr, "icp stats (n_src %zu):", (size_t)a.count);
      for (int k = 0; k < j; ++k) fprintf(stderr, " [%d v%llu s%llu k%llu f%llu]", k, t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
      fprintf(stderr, "\n");
    }
    if (d_trace) {
      std::vector<unsigned long long> t((size_t)16 * nb);
      (void)hipMemcpy(t.data(), d_trace, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipFree(d_trace);
      if (FILE* f = fopen(trace_path, "w")) {
        for (int b = 0; b < nb; ++b) {
          for (int k = 0; k < 16; ++k) fprintf(f, "%llu ", t[(size_t)b * 16 + k]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
    return O3DS_OK;
  }
  const int nb = pass_blocks(h, a.count);
  const int total_passes = params->max_iteration + 1;  // max_iter updates need max_iter+1 correspondence passes
  int launched = 0;
  while (launched < total_passes) {
    // the device loop terminates itself (done flag); the host only checks between chunks of queued passes
    const int chunk = std::min(total_passes - launched, launched == 0 ? 12 : 8);
    for (int k = 0; k < chunk; ++k) {
      if (h->session_precision == O3DS_PRECISION_F64)
        launch_accumulate<P4d>(h, a, h->session_crop, nb);
      else
        launch_accumulate<P4f>(h, a, h->session_crop, nb);
      icp_reduce_update_kernel<<<1, kUpdBlock, 0, h->stream>>>(h->d_partials, nb, h->d_state, (unsigned long long)a.count,
                                                            params->max_iteration, params->relative_fitness, params->relative_rmse,
                                                            h->debug_update, h->session_method, quantum_table(a));
    }
    launched += chunk;
    HIP_TRY(hipGetLastError());
    rc = read_state(h, out);
    if (rc) return rc;
    if (h->h_state->done) break;
  }
  return O3DS_OK;
}

int o3ds_icp_point_to_plane(o3ds_handle h, const double* src_xyz, size_t n_src, const double* tgt_xyz, const double* tgt_normals,
                            size_t n_tgt, const double init[16], const o3ds_icp_params* params, o3ds_icp_result* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!params || !out || !init) return fail(h, O3DS_ERR_INVALID_ARG, "icp: null argument");
  if (!(params->max_correspondence_distance > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "Invalid max_correspondence_distance.");
  if (n_tgt == 0) return fail(h, O3DS_ERR_EMPTY, "icp: empty target (map patch size is zero)");
  if (!tgt_normals) return fail(h, O3DS_ERR_NO_NORMALS, "TransformationEstimationPointToPlane requires target normals");
  o3ds_cloud s = 0, t = 0;
  int rc = o3ds_cloud_upload(h, src_xyz, nullptr, n_src, &s);
  if (rc) return rc;
  rc = o3ds_cloud_upload(h, tgt_xyz, tgt_normals, n_tgt, &t);
  if (!rc) rc = o3ds_icp_point_to_plane_dev(h, s, t, nullptr, init, params, out);
  const std::string keep = h->err;
  (void)o3ds_cloud_free(h, s);
  if (t) (void)o3ds_cloud_free(h, t);
  if (rc) h->err = keep;
  return rc;
}


// ---- pre-processing -----------------------------------------------------------------------------
}  // extern "C" (helpers below are C++)

namespace {

template <typename P4>
__global__ __launch_bounds__(kBlock) void reindex_copy_kernel(const P4* __restrict__ in, size_t n, P4* __restrict__ out, size_t off, int set_index) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    P4 p = in[i];
    if (set_index) p.i = (typename Scalar<P4>::index)(off + i);
    out[off + i] = p;
  }
}

// occupied cells of a grid index = cells whose start differs from the next start
__global__ __launch_bounds__(kBlock) void count_occupied_kernel(const int* __restrict__ cs, size_t m, unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock) c += cs[i + 1] != cs[i];
  for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// first position whose key has the pass-through bit (keys sorted ascending) -> number of in-volume points
__global__ void first_pass_kernel(const unsigned long long* __restrict__ keys, size_t n, unsigned long long* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  size_t lo = 0, hi = n;
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (keys[mid] & kPassBit)
      hi = mid;
    else
      lo = mid + 1;
  }
  *out = lo;
}

template <typename P4>
int crop_t(o3ds_handle h, const CloudRec& in, const CropDev& crop, CloudRec& out) {
  out.precision = in.precision;
  out.n = 0;
  box_copy(out, in);  // a subset
  if (in.n == 0) return O3DS_OK;
  int *flags = nullptr, *pos = nullptr;
  TMP_ALLOC(flags, sizeof(int) * (in.n + 1));
  TMP_ALLOC(pos, sizeof(int) * (in.n + 1));
  crop_flag_kernel<P4><<<grid_for(in.n), kBlock, 0, h->stream>>>((const P4*)in.pts, in.n, crop, flags);  // also writes the sentinel flags[n] = 0
  int rc = exclusive_scan_int(h, flags, pos, in.n + 1, pub_slot<int>(h, 0));
  if (rc) return rc;
  rc = wait_stream(h);
  if (rc) return rc;
  const int total = pub_value<int>(h, 0);
  out.n = (size_t)total;
  if (total > 0) {
    HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * out.n));
    if (in.nrm) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * out.n));
    compact_kernel<P4><<<grid_for(in.n), kBlock, 0, h->stream>>>((const P4*)in.pts, (const P4*)in.nrm, in.n, flags, pos, 1, (P4*)out.pts,
                                                               (P4*)out.nrm);
    if (in.col) {  // colours ride along as a second attribute array through the same kernel
      HIP_TRY(dev_alloc(h, (void**)&out.col, sizeof(P4) * out.n));
      compact_kernel<P4><<<grid_for(in.n), kBlock, 0, h->stream>>>((const P4*)in.col, nullptr, in.n, flags, pos, 1, (P4*)out.col, nullptr);
    }
    HIP_TRY(hipGetLastError());
  }
  dbg_sync(h, 4);
  return O3DS_OK;
}

// shared by VoxelDownSample (mode 0) and voxelizeWithinCroppingVolume (mode 1)
template <typename P4>
int voxel_reduce_t(o3ds_handle h, const CloudRec& in, int mode, double voxel, const CropDev& crop, CloudRec& out, bool filter = false,
                   long long merge_np = -1 /* >= 0: `in` is [pass block: merge_np | voxel block in key order: merge_nv | new points] */,
                   size_t merge_nv = 0) {
  // filter (mode 0 only): CroppingVolume::crop followed by VoxelDownSample in one go -- the points outside `crop` are keyed as
  // pass-through, sort behind the voxels and are not emitted; same grid anchor (the box of the INSIDE points), same keys, same
  // summation order as cropping first, without the compaction, its scan and its size read-back
  out.precision = in.precision;
  out.n = 0;
  const size_t n = in.n;
  if (n == 0) return O3DS_OK;
  double ox = 0, oy = 0, oz = 0;
  // VoxelDownSample proper: no sort (cloud_kernels.hpp, VoxTable); O3DS_VOXEL_SORT=1 keeps the sort-based path below for A/B runs
  static const bool voxel_sort = ab_getenv("O3DS_VOXEL_SORT") != nullptr;
  const bool table_path = mode == 0 && !voxel_sort && n < ((size_t)1 << 30);
  if (mode == 0) {  // [O3D] voxel_min_bound = GetMinBound() - voxel_size * 0.5
    double mn[3], mx[3];
    // the box of the points inside the volume: reduced by the ingest of this very cloud if it assumed the volume asked for now (a lidar
    // stream: always) -- the pinned record is normally there long before anyone asks; reduced now otherwise (one launch + a wait)
    CropDev want{};
    if (filter) want = crop;
    bool have_box = false;
    if (in.pre_slot >= 0 && memcmp(&in.pre_crop, &want, sizeof(CropDev)) == 0) {
      volatile o3ds_context::PinRec* r = h->h_rec + in.pre_slot;
      if (r->seq != in.pre_seq) HIP_TRY(hipStreamSynchronize(h->copy_stream));
      if (r->seq == in.pre_seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int a = 0; a < 3; ++a) mn[a] = r->box[a], mx[a] = r->box[3 + a];
        have_box = true;
      }
    }
    if (!have_box) {
      int rc = bbox_of<P4>(h, (const P4*)in.pts, n, mn, mx, filter ? &crop : nullptr);
      if (rc) return rc;
    }
    if (filter) {  // the next ingest of this handle reduces its box for this volume
      h->pre_crop = crop;
      h->pre_crop_valid = true;
    }
    if (filter && mn[0] > mx[0]) return O3DS_OK;  // nothing inside the volume: an empty cloud
    if (voxel * 2147483647.0 < std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]}) + voxel)
      return fail(h, O3DS_ERR_INVALID_ARG, "[VoxelDownSample] voxel_size is too small.");
    ox = mn[0] - voxel * 0.5;
    oy = mn[1] - voxel * 0.5;
    oz = mn[2] - voxel * 0.5;
    out.has_box = true;  // voxel means lie in the box of the points they average
    out.box_padded = false;
    for (int a = 0; a < 3; ++a) out.bmn[a] = mn[a], out.bmx[a] = mx[a];
    if (!std::isfinite(mn[0] + mn[1] + mn[2] + mx[0] + mx[1] + mx[2])) out.has_box = false;
  } else {
    box_copy(out, in);
  }
  if (out.has_box) box_inflate(out);
  if (table_path) {
    size_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    // the handle's table: all 0xff between calls (vox_mean_kernel empties the slots a call used), grown to the largest cloud seen
    if (h->voxtab_cap < cap) {
      if (h->d_voxtab) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_voxtab);
        h->d_voxtab = nullptr;
        h->voxtab_cap = 0;
      }
      if (hipMalloc((void**)&h->d_voxtab, kVoxSlotBytes * cap + 16) != hipSuccess) return fail(h, O3DS_ERR_OOM, "VoxelDownSample: voxel table allocation failed");
      h->voxtab_cap = cap;
      h->voxtab_clean = false;
    }
    cap = h->voxtab_cap;
    unsigned char* tab = h->d_voxtab;
    int *lead_slot = nullptr, *run_next = nullptr, *run_len = nullptr, *order = nullptr;
    uint32_t* starts = nullptr;
    int2* piece = nullptr;
    TMP_ALLOC(lead_slot, sizeof(int) * n);
    TMP_ALLOC(run_next, sizeof(int) * n);
    TMP_ALLOC(run_len, sizeof(int) * n);
    TMP_ALLOC(order, sizeof(int) * n);
    TMP_ALLOC(starts, sizeof(uint32_t) * n);
    if (in.col) TMP_ALLOC(piece, sizeof(int2) * n);
    VoxTable t{(VoxSlot*)tab, (unsigned int*)(tab + kVoxSlotBytes * cap), (unsigned int)(cap - 1)};
    static const bool always_clear = ab_getenv("O3DS_ALWAYS_CLEAR") != nullptr;
    if (!h->voxtab_clean || always_clear) HIP_TRY(hipMemsetAsync(tab, 0xff, kVoxSlotBytes * cap + 16, h->stream));
    h->voxtab_clean = false;
    // the chained scan's tile records (kept across calls, see vox_order_kernel)
    const size_t n_tiles = (n + kVoxTile - 1) / kVoxTile;
    if (h->tiles_cap < n_tiles) {
      if (h->d_tiles) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_tiles);
        h->d_tiles = nullptr;
        h->tiles_cap = 0;
      }
      const size_t want = std::max<size_t>(n_tiles + n_tiles / 4, 1024);
      if (hipMalloc((void**)&h->d_tiles, sizeof(unsigned long long) * want) != hipSuccess) return fail(h, O3DS_ERR_OOM, "VoxelDownSample: scan state allocation failed");
      HIP_TRY(hipMemsetAsync(h->d_tiles, 0, sizeof(unsigned long long) * want, h->stream));
      h->tiles_cap = want;
    }
    // the size of the result: an upper bound here, the number of voxels in a device word and a pinned record when vox_order_kernel has run
    out.n = n;
    out.n_lower = 1;  // (the box is not empty: at least one point lies inside the volume, hence at least one voxel)
    out.lazy_slot = take_rec(h);
    if (out.lazy_slot < 0) return fail(h, O3DS_ERR_CAPACITY, "VoxelDownSample: no record free for the size of the result (every one is held by a persistent map)");
    out.lazy_seq = ++h->rec_seq;
    CountPub pub{cnt_word(h, out.lazy_slot), &(h->h_rec_dev + out.lazy_slot)->cnt, out.lazy_seq};
    const CountRef m_ref{n, pub.dev};
    HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * n));
    if (in.nrm) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * n));
    if (in.col) HIP_TRY(dev_alloc(h, (void**)&out.col, sizeof(P4) * n));
    if (++h->scan_gen == 0) h->scan_gen = 1;
    vox_insert_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)in.pts, n, ox, oy, oz, voxel, crop, filter ? 1 : 0, t, lead_slot, run_next, run_len);
    vox_order_kernel<<<(unsigned int)n_tiles, kBlock, 0, h->stream>>>(lead_slot, CountRef{n, nullptr}, t, h->d_tiles, h->d_ticket, h->ticket_base, h->scan_gen, order, pub);
    h->ticket_base += (unsigned int)n_tiles;
    vox_mean_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)in.pts, (const P4*)in.nrm, m_ref, order, run_next, run_len, starts, piece, 0,
                                                             (P4*)out.pts, (P4*)out.nrm, t);
    h->voxtab_clean = true;
    if (in.col)  // [O3D] VoxelDownSample: AccumulatedPoint averages the colours
      vox_mean_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)in.col, nullptr, m_ref, order, run_next, run_len, starts, piece, 1, (P4*)out.col,
                                                               nullptr, t);
    HIP_TRY(hipGetLastError());
    dbg_sync(h, 8);
    return O3DS_OK;
  }
  unsigned long long *k0 = nullptr, *k1 = nullptr, *d_scalar = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr;
  size_t merged_n_inside = 0;
  int *head = nullptr, *seg_id = nullptr, *seg_start = nullptr;
  TMP_ALLOC(k0, sizeof(unsigned long long) * n);
  TMP_ALLOC(k1, sizeof(unsigned long long) * n);
  TMP_ALLOC(v0, sizeof(uint32_t) * n);
  TMP_ALLOC(v1, sizeof(uint32_t) * n);
  TMP_ALLOC(head, sizeof(int) * (n + 1));
  TMP_ALLOC(seg_id, sizeof(int) * (n + 1));
  TMP_ALLOC(d_scalar, sizeof(unsigned long long));
  // The sorted (key, index) list: by merging when the input is a map the previous merge left in key order plus new points (only the
  // new keys are sorted), by one radix sort of everything otherwise.  Same arrays either way (cloud_kernels.hpp, merge_class_kernel).
  static const bool no_incremental = ab_getenv("O3DS_NO_INCREMENTAL_MERGE") != nullptr;  // A/B and debugging
  bool merged = false;
  if (mode == 1 && !filter && merge_np >= 0 && !no_incremental && n < ((size_t)1 << 31) && (size_t)merge_np + merge_nv <= n) {
    const size_t np = (size_t)merge_np, nv = merge_nv;
    unsigned long long *cls = nullptr, *rank = nullptr, *vk = nullptr, *xk = nullptr, *xk2 = nullptr;
    uint32_t *vv = nullptr, *xv = nullptr, *xv2 = nullptr;
    int* d_unsorted = nullptr;
    const size_t nx_cap = n - nv;  // the pass block and the new points
    TMP_ALLOC(cls, sizeof(unsigned long long) * (n + 1));
    TMP_ALLOC(rank, sizeof(unsigned long long) * (n + 1));
    TMP_ALLOC(vk, sizeof(unsigned long long) * (nv + 1));
    TMP_ALLOC(vv, sizeof(uint32_t) * (nv + 1));
    TMP_ALLOC(xk, sizeof(unsigned long long) * (nx_cap + 1));
    TMP_ALLOC(xv, sizeof(uint32_t) * (nx_cap + 1));
    // the "not in key order" flag and the three totals are stored into the pinned block by the kernels themselves (slot 2: the host
    // clears it here -- nothing is in flight that writes it, every use ends in a synchronisation)
    d_unsorted = pub_slot<int>(h, 2);
    *(volatile int*)(h->h_pin + kPubOff + 32) = 0;
    merge_class_kernel<P4><<<grid_for(n + 1), kBlock, 0, h->stream>>>((const P4*)in.pts, n, voxel, crop, np, nv, k0, cls, d_unsorted);
    int rcs = exclusive_scan_t<unsigned long long>(h, cls, rank, n + 1, pub_slot<unsigned long long>(h, 1));
    if (rcs) return rcs;
    rcs = wait_stream(h);
    if (rcs) return rcs;
    const unsigned long long tot = pub_value<unsigned long long>(h, 1);
    const int unsorted = pub_value<int>(h, 2);
    if (!unsorted) {
      const size_t npass_ = (size_t)(tot & kCntMask), nvin = (size_t)(tot >> 32), nx = n - npass_ - nvin;
      int *xpos = nullptr, *cnt = nullptr, *before = nullptr;
      TMP_ALLOC(xpos, sizeof(int) * (nx + 1));
      TMP_ALLOC(cnt, sizeof(int) * (nvin + 2));
      TMP_ALLOC(before, sizeof(int) * (nvin + 2));
      merge_split_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k0, rank, n, np, nv, k1, v1, vk, vv, xk, xv, cnt, nvin + 2);
      const unsigned long long* xks = xk;
      const uint32_t* xvs = xv;
      if (nx > 1) {  // the only sort: the points that are new to the volume -- tile sort in LDS, then merge passes (cloud_kernels.hpp)
        TMP_ALLOC(xk2, sizeof(unsigned long long) * nx);
        TMP_ALLOC(xv2, sizeof(uint32_t) * nx);
        static const bool lib_sort = ab_getenv("O3DS_MERGE_LIBRARY_SORT") != nullptr;  // A/B: rocPRIM's radix sort of the same pairs
        if (lib_sort) {
          size_t tb = 0;
          HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, xk, xk2, xv, xv2, nx, 0, 63, h->stream));
          void* tmp = nullptr;
          TMP_ALLOC(tmp, tb ? tb : 16);
          HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, xk, xk2, xv, xv2, nx, 0, 63, h->stream));
          xks = xk2, xvs = xv2;
        } else {
          unsigned long long *ka = xk2, *kb = xk;  // the tile sort reads xk / xv and writes the second pair of buffers; passes ping-pong
          uint32_t *va = xv2, *vb = xv;
          sort_tile_kernel<<<(unsigned int)((nx + kSortTile - 1) / kSortTile), kBlock, 0, h->stream>>>(xk, xv, nx, ka, va);
          static const int ways = [] {  // tuning experiments: 2, 4, 8 or 16 runs merged per pass; anything else is ignored (a width that
            const int w = ab_getenv("O3DS_SORT_WAYS") ? atoi(ab_getenv("O3DS_SORT_WAYS")) : kSortWays;  // grows by 3 under an 8-way kernel sorts wrongly)
            return (w == 2 || w == 4 || w == 8 || w == 16) ? w : kSortWays;
          }();
          for (size_t width = kSortTile; width < nx; width *= (size_t)ways) {  // nx < 2^31 on this path
            const unsigned int gsz = (unsigned int)grid_for(nx);
            if (ways == 2)
              sort_merge_pass_kernel<2><<<gsz, kBlock, 0, h->stream>>>(ka, va, (uint32_t)nx, (uint32_t)width, kb, vb);
            else if (ways == 4)
              sort_merge_pass_kernel<4><<<gsz, kBlock, 0, h->stream>>>(ka, va, (uint32_t)nx, (uint32_t)width, kb, vb);
            else if (ways == 16)
              sort_merge_pass_kernel<16><<<gsz, kBlock, 0, h->stream>>>(ka, va, (uint32_t)nx, (uint32_t)width, kb, vb);
            else
              sort_merge_pass_kernel<8><<<gsz, kBlock, 0, h->stream>>>(ka, va, (uint32_t)nx, (uint32_t)width, kb, vb);
            std::swap(ka, kb);
            std::swap(va, vb);
          }
          xks = ka, xvs = va;
        }
      }
      if (nx) merge_rank_kernel<<<grid_for(nx), kBlock, 0, h->stream>>>(xks, xvs, nx, vk, nvin, np, xpos, cnt);
      rcs = exclusive_scan_int(h, cnt, before, nvin + 2);
      if (rcs) return rcs;
      if (nvin + nx) merge_place_kernel<<<grid_for(nvin + nx), kBlock, 0, h->stream>>>(vk, vv, nvin, before, xks, xvs, xpos, nx, k1, v1);
      (void)npass_;
      merged = true;
      merged_n_inside = nvin + nx;
    }
  }
  if (!merged) {
    voxel_key_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)in.pts, n, mode, ox, oy, oz, voxel, crop, k0, v0, filter ? 1 : 0);
    size_t temp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
    void* temp = nullptr;
    TMP_ALLOC(temp, temp_bytes ? temp_bytes : 16);
    HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
  }
  segment_head_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k1, n, head);  // also writes the sentinel head[n] = 0
  if (!merged) first_pass_kernel<<<1, 64, 0, h->stream>>>(k1, n, pub_slot<unsigned long long>(h, 3));
  int rc = exclusive_scan_int(h, head, seg_id, n + 1, pub_slot<int>(h, 0));
  if (rc) return rc;
  rc = wait_stream(h);
  if (rc) return rc;
  const int n_seg = pub_value<int>(h, 0);
  const unsigned long long n_inside = merged ? (unsigned long long)merged_n_inside : pub_value<unsigned long long>(h, 3);
  (void)d_scalar;
  const size_t n_pass = n - (size_t)n_inside;
  if (mode == 1 && !filter) {  // the layout the next merge can rely on
    out.vox_first = (long long)n_pass;
    out.vox_count = (size_t)n_seg - n_pass;
  }
  TMP_ALLOC(seg_start, sizeof(int) * ((size_t)n_seg + 1));
  segment_start_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(head, seg_id, n, seg_start);
  const int drop = filter ? 1 : 0;
  out.n = filter ? (size_t)n_seg - n_pass : (size_t)n_seg;
  if (out.n == 0) return O3DS_OK;
  // a map (mode 1) gets room for the scans to come: Submap::insertScan appends one before every merge, and with the room in place the
  // append is one kernel that places the scan's points behind the map's instead of two allocations and a copy of the whole map
  const size_t room = (mode == 1 && !filter) ? out.n + std::max<size_t>(out.n / 4, (size_t)1 << 18) : out.n;
  HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * room));
  if (in.nrm) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * room));
  if (room > out.n && !in.col) out.cap = room;
  segment_mean_kernel<P4><<<grid_for((size_t)n_seg), kBlock, 0, h->stream>>>((const P4*)in.pts, (const P4*)in.nrm, k1, v1, seg_start, (size_t)n_seg, n,
                                                                            mode == 1 ? 1 : 0, n_pass, (P4*)out.pts, (P4*)out.nrm, drop);
  if (in.col) {
    HIP_TRY(dev_alloc(h, (void**)&out.col, sizeof(P4) * out.n));
    if (mode == 0)  // [O3D] VoxelDownSample: AccumulatedPoint averages the colours
      segment_mean_kernel<P4><<<grid_for((size_t)n_seg), kBlock, 0, h->stream>>>((const P4*)in.col, nullptr, k1, v1, seg_start, (size_t)n_seg, n, 0, n_pass,
                                                                                (P4*)out.col, nullptr, drop);
    else  // the map merge keeps the colour of the LAST point of a voxel (helpers.cpp:40-42,61-63: `color_ = ...`, not `+=`)
      segment_last_kernel<P4><<<grid_for(out.n), kBlock, 0, h->stream>>>((const P4*)in.col, k1, v1, seg_start, out.n, n, n_pass, (P4*)out.col);
  }
  HIP_TRY(hipGetLastError());
  dbg_sync(h, 8);
  return O3DS_OK;
}

template <typename P4>
int normals_t(o3ds_handle h, CloudRec& c, double radius, int max_nn, bool knn_raw = false) {
  if (c.n == 0) return O3DS_OK;
  if (knn_raw || !c.has_box) {  // (paths that reduce a box on the host need the exact size first)
    const int rr = resolve_count(h, c, true);
    if (rr) return rr;
    if (c.n == 0) return O3DS_OK;
  }
  if (knn_raw) {  // [O3D] EstimateNormals(KDTreeSearchParamKNN(max_nn)) and nothing after it: a radius no pair of points exceeds
    double mn[3], mx[3];
    const int rb = bbox_of<P4>(h, (const P4*)c.pts, c.n, mn, mx);
    if (rb) return rb;
    const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    radius = std::sqrt(dx * dx + dy * dy + dz * dz) * 1.001 + 1e-3;
    if (!std::isfinite(radius)) return fail(h, O3DS_ERR_INVALID_ARG, "estimate_normals: non-finite points");
  }
  // The cell size only steers the cost of the (exact) ring search: aim at max_nn / pi points per occupied cell, so that the
  // max_nn-th neighbour typically lies inside the first 3x3x3 ring.  The density comes from a pilot grid at radius/8 -- one
  // extra index build, one counting kernel and a host round trip -- so it is remembered per (radius, max_nn) and reused
  // while the cloud size stays within 25 % (a lidar stream: every scan), refreshed every 32 calls.
  // (a cloud whose size is still in flight: the heuristics below go by the size of the handle's previous VoxelDownSample result -- a lidar
  // stream's scans are alike --, the kernels by the device word; the density pilot needs the real size and waits for it)
  size_t n_est = c.lazy_slot >= 0 ? std::min(c.n, h->voxel_count_hint) : c.n;
  bool reuse = h->nrm_cell > 0.0 && h->nrm_radius == radius && h->nrm_knn == max_nn && h->nrm_age < 32 && n_est > 0 &&
               (double)n_est <= 1.25 * (double)h->nrm_n && (double)n_est >= 0.75 * (double)h->nrm_n;
  if (!reuse && c.lazy_slot >= 0) {
    const int rr = resolve_count(h, c, true);
    if (rr) return rr;
    if (c.n == 0) return O3DS_OK;
    n_est = c.n;
    reuse = h->nrm_cell > 0.0 && h->nrm_radius == radius && h->nrm_knn == max_nn && h->nrm_age < 32 &&
            (double)n_est <= 1.25 * (double)h->nrm_n && (double)n_est >= 0.75 * (double)h->nrm_n;
  }
  const int* n_dev = c.lazy_slot >= 0 ? cnt_word(h, c.lazy_slot) : nullptr;
  CloudRec tmp;
  tmp.n = c.n;
  tmp.precision = c.precision;
  tmp.pts = c.pts;  // borrowed
  tmp.lazy_slot = c.lazy_slot;  // (borrowed as well: build_index_t hands the device word to its kernels)
  box_copy(tmp, c);
  int rc = O3DS_OK;
  if (reuse) {
    ++h->nrm_age;
    rc = build_index_t<P4>(h, tmp, h->nrm_cell);
    if (rc) {
      tmp.pts = nullptr;
      free_index(h, tmp);
      return rc;
    }
  } else {
    rc = build_index_t<P4>(h, tmp, radius / 8.0);
    if (rc) {
      tmp.pts = nullptr;
      free_index(h, tmp);
      return rc;
    }
    const size_t ncell = (size_t)tmp.grid.nx * tmp.grid.ny * tmp.grid.nz;
    unsigned long long* d_cnt = nullptr;
    TMP_ALLOC(d_cnt, sizeof(unsigned long long));
    HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), h->stream));
    count_occupied_kernel<<<grid_for(ncell), kBlock, 0, h->stream>>>(tmp.cell_start, ncell, d_cnt);
    unsigned long long occ = 0;
    rc = read_back(h, {{&occ, d_cnt, sizeof(occ)}});
    if (rc) {
      tmp.pts = nullptr;
      free_index(h, tmp);
      return rc;
    }
    const double avg = occ ? (double)c.n / (double)occ : 1.0;
    static const double cell_scale = ab_getenv("O3DS_NRM_CELL_SCALE") ? atof(ab_getenv("O3DS_NRM_CELL_SCALE")) : 1.0;  // tuning experiments
    double cell = cell_scale * tmp.grid.cell * std::sqrt(std::max(1.0, (double)max_nn) / (3.14159265358979 * avg));
    cell = std::min(std::max(cell, radius / 64.0), radius);
    rc = build_index_t<P4>(h, tmp, cell);
    if (rc) {
      tmp.pts = nullptr;
      free_index(h, tmp);
      return rc;
    }
    h->nrm_cell = tmp.grid.cell;
    h->nrm_radius = radius;
    h->nrm_knn = max_nn;
    h->nrm_n = c.n;
    h->nrm_age = 0;
  }
  // (a map keeps room behind its points, CloudRec::cap: normals that arrive later get the same room, the in-place append of
  // o3ds_map_insert_scan writes both arrays behind the last point)
  if (!c.nrm) HIP_TRY(dev_alloc(h, (void**)&c.nrm, sizeof(P4) * std::max(c.n, c.cap)));
  const int rmax = std::max(1, (int)std::ceil(radius / tmp.grid.cell));
  static const bool nrm_debug = ab_getenv("O3DS_NRM_DEBUG") != nullptr;  // debugging aid: where a fault happens
  if (nrm_debug) {
    const hipError_t e = hipStreamSynchronize(h->stream);
    fprintf(stderr, "[normals] n %zu radius %g max_nn %d reuse %d grid %d x %d x %d cell %g rmax %d (index built: %s)\n", c.n, radius, max_nn, (int)reuse,
            tmp.grid.nx, tmp.grid.ny, tmp.grid.nz, tmp.grid.cell, rmax, hipGetErrorString(e));
    const size_t ncell = (size_t)tmp.grid.nx * tmp.grid.ny * tmp.grid.nz;
    std::vector<int> hcs(ncell + 1);
    (void)hipMemcpy(hcs.data(), tmp.cell_start, sizeof(int) * (ncell + 1), hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < ncell; ++i)
      if (hcs[i + 1] < hcs[i]) {
        if (!bad) first = i;
        ++bad;
      }
    fprintf(stderr, "[normals] cell_start: first %d last %d (n %zu), descents %zu", hcs[0], hcs[ncell], c.n, bad);
    if (bad) fprintf(stderr, " first at cell %zu (block %zu): %d -> %d", first, (first + 1) / 1024, hcs[first], hcs[first + 1]);
    fprintf(stderr, "\n");
  }
  // 16 lanes per point, 16 points per 256-thread workgroup (normals_kernel.hpp), then a thread per point for the eigen-solve; the instantiation is picked by max_nn
  {
    const P4* p_pts = (const P4*)c.pts;
    const P4* p_sp = (const P4*)tmp.spts;
    P4* p_out = (P4*)c.nrm;
    const unsigned int gsz = (unsigned int)((c.n + o3ds::kNrmPointsPerBlock - 1) / o3ds::kNrmPointsPerBlock);
#ifdef O3DS_NRM_CHECK
    unsigned long long* d_ws = nullptr;
    if (ab_getenv("O3DS_NRM_STATS_FILE")) {
      HIP_TRY(hipMalloc((void**)&d_ws, sizeof(unsigned long long) * o3ds::kNrmStatWords * ((size_t)o3ds::kNrmWaves * gsz)));
      HIP_TRY(hipMemset(d_ws, 0, sizeof(unsigned long long) * o3ds::kNrmStatWords * ((size_t)o3ds::kNrmWaves * gsz)));
    }
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(o3ds::g_nrm_wave_stats), &d_ws, sizeof(d_ws)));
#endif
    double* d_sums = nullptr;
    int* d_cnts = nullptr;
    TMP_ALLOC(d_sums, sizeof(double) * 9 * c.n);
    TMP_ALLOC(d_cnts, sizeof(int) * c.n);
    span_mark(h, kSpanNormalsKernels);
    if (max_nn <= 32) {  // the shipped configs' knn is 20
      if constexpr (sizeof(P4) == 16)
        normals_kernel_occ5<P4, 32><<<gsz, 64 * o3ds::kNrmWaves, 0, h->stream>>>(p_pts, c.n, tmp.grid, p_sp, radius, max_nn, rmax, d_sums, d_cnts, n_dev);
      else
        normals_kernel<P4, 32><<<gsz, 64 * o3ds::kNrmWaves, 0, h->stream>>>(p_pts, c.n, tmp.grid, p_sp, radius, max_nn, rmax, d_sums, d_cnts, n_dev);
    } else
      normals_kernel<P4, 128><<<gsz, 64 * o3ds::kNrmWaves, 0, h->stream>>>(p_pts, c.n, tmp.grid, p_sp, radius, max_nn, rmax, d_sums, d_cnts, n_dev);
    // the grid, the cell-ordered points and (written here) the cell-ordered normals are a complete nearest-neighbour index of the cloud:
    // kept, so that a registration against this cloud (scan-to-scan odometry: the previous scan) does not build another one
    if (dev_alloc(h, (void**)&tmp.snrm, sizeof(P4) * c.n) != hipSuccess) {
      tmp.pts = nullptr;
      free_index(h, tmp);
      return fail(h, O3DS_ERR_OOM, "estimate_normals: out of device memory");
    }
    normals_finish_kernel<P4><<<(unsigned int)((c.n + 255) / 256), 256, 0, h->stream>>>(p_sp, c.n, d_sums, d_cnts, p_out, knn_raw ? 1 : 0, (P4*)tmp.snrm, n_dev);
    span_mark(h, kSpanNormalsKernels);
  }
  HIP_TRY(hipGetLastError());
  dbg_sync(h, 16);
#ifdef O3DS_NRM_CHECK
  if (const char* sf = ab_getenv("O3DS_NRM_STATS_FILE")) {
    const unsigned int gsz = (unsigned int)((c.n + o3ds::kNrmPointsPerBlock - 1) / o3ds::kNrmPointsPerBlock);
    unsigned long long* d_ws = nullptr;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpyFromSymbol(&d_ws, HIP_SYMBOL(o3ds::g_nrm_wave_stats), sizeof(d_ws)));
    std::vector<unsigned long long> ws((size_t)o3ds::kNrmStatWords * ((size_t)o3ds::kNrmWaves * gsz));
    HIP_TRY(hipMemcpy(ws.data(), d_ws, sizeof(unsigned long long) * ws.size(), hipMemcpyDeviceToHost));
    (void)hipFree(d_ws);
    if (FILE* f = fopen(sf, "wb")) {
      fwrite(ws.data(), sizeof(unsigned long long), ws.size(), f);
      fclose(f);
    }
  }
#endif
  if (nrm_debug) {
    fprintf(stderr, "[normals] kernel done: %s\n", hipGetErrorString(hipStreamSynchronize(h->stream)));
#ifdef O3DS_NRM_CHECK
    unsigned int dbg[8] = {0};
    (void)hipMemcpyFromSymbol(dbg, HIP_SYMBOL(o3ds::g_nrm_dbg), sizeof(dbg));
    fprintf(stderr, "[normals] violations: unsorted %u, bad p %u, bad oi %u\n", dbg[0], dbg[1], dbg[2]);
    const unsigned int zero[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(o3ds::g_nrm_dbg), zero, sizeof(zero));
#endif
  }
  tmp.pts = nullptr;
  box_copy(c, tmp);  // the box an index build reduced is kept for the cloud
  free_index(h, c);  // the cloud's own index (if any) no longer matches its normals
  c.grid = tmp.grid, c.cell_start = tmp.cell_start, c.spts = tmp.spts, c.snrm = tmp.snrm;
  c.has_index = true;
  c.index_byproduct = true;
  return O3DS_OK;
}

int gicp_knn_normals(o3ds_handle h, CloudRec& c) {
  return c.precision == O3DS_PRECISION_F64 ? normals_t<P4d>(h, c, 0.0, 20, true) : normals_t<P4f>(h, c, 0.0, 20, true);
}

template <typename P4>
int transform_t(o3ds_handle h, const CloudRec& in, const double T[16], CloudRec& out) {
  out.precision = in.precision;
  out.n = in.n;
  box_transform(out, in, T);
  if (in.n == 0) return O3DS_OK;
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) M.m[r * 4 + c] = T[c * 4 + r];
  HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * in.n));
  if (in.nrm) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * in.n));
  transform_kernel<P4><<<grid_for(in.n), kBlock, 0, h->stream>>>((const P4*)in.pts, (const P4*)in.nrm, in.n, M, T[3], T[7], T[11], T[15],
                                                               (P4*)out.pts, (P4*)out.nrm, 0);
  if (in.col) {
    HIP_TRY(dev_alloc(h, (void**)&out.col, sizeof(P4) * in.n));
    HIP_TRY(hipMemcpyAsync(out.col, in.col, sizeof(P4) * in.n, hipMemcpyDeviceToDevice, h->stream));
  }
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

// Submap::carve (Submap.cpp:109-125): remove from `map` the points that the rays of `scan` (sensor frame, placed by T) see through
template <typename P4>
int carve_t(o3ds_handle h, CloudRec& map, const CloudRec& scan, const double T[16], const CropDev& crop, const o3ds_carving_params& cp,
            size_t* n_removed, CloudRec* removed = nullptr /* the carved points, in map order (Submap::toRemove_) */) {
  *n_removed = 0;
  const size_t n = map.n;
  if (n == 0 || scan.n == 0) return O3DS_OK;
  unsigned long long *k0 = nullptr, *k1 = nullptr, *tkey = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr;
  int *head = nullptr, *seg_id = nullptr, *seg_start = nullptr, *tseg = nullptr, *keep = nullptr, *pos = nullptr;
  TMP_ALLOC(k0, sizeof(unsigned long long) * n);
  TMP_ALLOC(k1, sizeof(unsigned long long) * n);
  TMP_ALLOC(v0, sizeof(uint32_t) * n);
  TMP_ALLOC(v1, sizeof(uint32_t) * n);
  TMP_ALLOC(head, sizeof(int) * (n + 1));
  TMP_ALLOC(seg_id, sizeof(int) * (n + 1));
  // VoxelMap of the map points inside the wide cropping volume (cropper.getIndicesWithinVolume + insertCloud): same key as the merge
  voxel_key_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)map.pts, n, 1, 0.0, 0.0, 0.0, cp.voxel_size, crop, k0, v0);
  // The sorted (key, index) list.  A map that the merge of an insertion left is [outside points | voxel block in ascending key order], and the
  // carving volume is the volume of that insertion (Submap.cpp:56-71: the cropper still holds its pose): the inside entries are then already in
  // key order, and a stable PARTITION by "inside" (one scan, one scatter) gives what the stable sort gives.  Whether they really were in order
  // -- another voxel size, a map assembled otherwise, a mean that rounding put across a voxel boundary -- is checked on the device while the
  // segment heads are formed; if not, the library sort runs as before (O3DS_CARVE_SORT=1 forces it, for A/B runs).
  static const bool carve_sort = ab_getenv("O3DS_CARVE_SORT") != nullptr;
  const bool try_partition = !carve_sort && map.vox_first >= 0 && (size_t)map.vox_first + map.vox_count == n && n < ((size_t)1 << 31);
  int rc = O3DS_OK;
  int n_seg = 0;
  bool sorted_ok = false;
  if (try_partition) {
    int* rank = nullptr;
    TMP_ALLOC(rank, sizeof(int) * (n + 1));
    {
      const size_t ms = n + 1;
      const int nbs = (int)((ms + kScanPerBlock - 1) / kScanPerBlock);
      int* sums = nullptr;
      TMP_ALLOC(sums, sizeof(int) * (size_t)nbs);
      scan_local_fn_kernel<int, KeyInsideFlag><<<nbs, kBlock, 0, h->stream>>>(KeyInsideFlag{k0, n}, rank, sums, ms, nullptr);
      if (nbs > 1 && nbs <= kScanFusedBlocks) {
        scan_add_fused_kernel<int><<<nbs, kBlock, 0, h->stream>>>(rank, sums, ms, nullptr);
      } else if (nbs > 1) {
        scan_sums_kernel<int><<<1, kBlock, 0, h->stream>>>(sums, nbs);
        scan_add_kernel<int><<<nbs, kBlock, 0, h->stream>>>(rank, sums, ms, nullptr);
      }
    }
    partition_keys_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k0, rank, n, k1, v1);
    int* d_unsorted = pub_slot<int>(h, 2);
    *(volatile int*)(h->h_pin + kPubOff + 32) = 0;  // (nothing in flight writes the slot: every use ends in a synchronisation)
    segment_head_check_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k1, n, head, d_unsorted);
    rc = exclusive_scan_int(h, head, seg_id, n + 1);
    if (rc) return rc;
    rc = read_back(h, {{&n_seg, seg_id + n, sizeof(int)}});
    if (rc) return rc;
    sorted_ok = pub_value<int>(h, 2) == 0;
  }
  if (!sorted_ok) {
    size_t temp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
    void* temp = nullptr;
    TMP_ALLOC(temp, temp_bytes ? temp_bytes : 16);
    HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
    segment_head_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k1, n, head);  // also writes the sentinel head[n] = 0
    rc = exclusive_scan_int(h, head, seg_id, n + 1);
    if (rc) return rc;
    rc = read_back(h, {{&n_seg, seg_id + n, sizeof(int)}});
    if (rc) return rc;
  }
  TMP_ALLOC(seg_start, sizeof(int) * ((size_t)n_seg + 1));
  segment_start_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(head, seg_id, n, seg_start);
  size_t tsize = 1024;
  while (tsize < 2 * (size_t)n_seg) tsize <<= 1;
  TMP_ALLOC(tkey, sizeof(unsigned long long) * tsize);
  TMP_ALLOC(tseg, sizeof(int) * tsize);
  HIP_TRY(hipMemsetAsync(tkey, 0xFF, sizeof(unsigned long long) * tsize, h->stream));
  unsigned int* block_bits = nullptr;  // one bit per hashed 4 x 4 x 4 block of voxels (cloud_kernels.hpp, carve_block_bit)
  TMP_ALLOC(block_bits, sizeof(unsigned int) << (kCarveBitsLog2 - 5));
  HIP_TRY(hipMemsetAsync(block_bits, 0, sizeof(unsigned int) << (kCarveBitsLog2 - 5), h->stream));
  carve_table_insert_kernel<<<grid_for((size_t)n_seg), kBlock, 0, h->stream>>>(k1, seg_start, (size_t)n_seg, tkey, tseg, (unsigned int)(tsize - 1),
                                                                               block_bits);
  TMP_ALLOC(keep, sizeof(int) * (n + 1));
  TMP_ALLOC(pos, sizeof(int) * (n + 1));
  fill_int_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(keep, n, 1);
  HIP_TRY(hipMemsetAsync(keep + n, 0, sizeof(int), h->stream));
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) M.m[r * 4 + c] = T[c * 4 + r];
  carve_rays_kernel<P4><<<grid_for(scan.n), kBlock, 0, h->stream>>>((const P4*)scan.pts, scan.n, M, T[12], T[13], T[14], cp.voxel_size,
                                                                   cp.max_raytracing_length, cp.truncation_distance,
                                                                   cp.min_dot_product_with_normal, tkey, tseg, (unsigned int)(tsize - 1), seg_start,
                                                                   (size_t)n_seg, n, v1, (const P4*)map.nrm, keep, block_bits);
  rc = exclusive_scan_int(h, keep, pos, n + 1);
  if (rc) return rc;
  int total = 0, kept_outside = 0;
  // with the merge's layout known, the number of kept points of its first block comes back with the total: both blocks shrink in place, in
  // order, and the next insertion still finds [outside points | voxel block in key order] -- no sort of the whole map after a carve
  const bool layout = map.vox_first >= 0 && (size_t)map.vox_first + map.vox_count == n;
  rc = layout ? read_back(h, {{&total, pos + n, sizeof(int)}, {&kept_outside, pos + (size_t)map.vox_first, sizeof(int)}})
              : read_back(h, {{&total, pos + n, sizeof(int)}});
  if (rc) return rc;
  *n_removed = n - (size_t)total;
  if (*n_removed == 0) return O3DS_OK;  // removeByIds: nothing to do (helpers.cpp:221-223)
  if (removed) {  // map->SelectByIndex(idxsToRemove) (Submap.cpp:119): the same flags and positions, the other side of the compaction
    removed->precision = map.precision;
    removed->n = *n_removed;
    HIP_TRY(dev_alloc(h, (void**)&removed->pts, sizeof(P4) * removed->n));
    if (map.nrm) HIP_TRY(dev_alloc(h, (void**)&removed->nrm, sizeof(P4) * removed->n));
    compact_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)map.pts, (const P4*)map.nrm, n, keep, pos, 0, (P4*)removed->pts, (P4*)removed->nrm);
    if (map.col) {
      HIP_TRY(dev_alloc(h, (void**)&removed->col, sizeof(P4) * removed->n));
      compact_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)map.col, nullptr, n, keep, pos, 0, (P4*)removed->col, nullptr);
    }
    box_copy(*removed, map);
  }
  void *np = nullptr, *nn = nullptr, *nc = nullptr;
  if (total > 0) {
    HIP_TRY(dev_alloc(h, (void**)&np, sizeof(P4) * (size_t)total));
    if (map.nrm) HIP_TRY(dev_alloc(h, (void**)&nn, sizeof(P4) * (size_t)total));
    compact_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)map.pts, (const P4*)map.nrm, n, keep, pos, 1, (P4*)np, (P4*)nn);
    if (map.col) {
      HIP_TRY(dev_alloc(h, (void**)&nc, sizeof(P4) * (size_t)total));
      compact_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>((const P4*)map.col, nullptr, n, keep, pos, 1, (P4*)nc, nullptr);
    }
    HIP_TRY(hipGetLastError());
  }
  free_index(h, map);
  free_points(h, map);
  drop_ingest_box(h, map);
  if (map.nrm) dev_free(h, map.nrm);
  if (map.col) dev_free(h, map.col);
  map.pts = np;
  map.nrm = nn;
  map.col = nc;
  map.n = (size_t)total;
  map.cap = 0;
  if (layout) {
    map.vox_first = (long long)kept_outside;
    map.vox_count = (size_t)total - (size_t)kept_outside;
  } else {
    map.vox_first = -1;
  }
  return O3DS_OK;
}

// computeIndicesOfOverlappingPoints (helpers.cpp:307-332)
template <typename P4>
int overlap_t(o3ds_handle h, const CloudRec& src, const CloudRec& tgt, const double T[16], double voxel, size_t min_points,
              unsigned long long* idx_src, size_t* n_src_out, unsigned long long* idx_tgt, size_t* n_tgt_out) {
  *n_src_out = *n_tgt_out = 0;
  const size_t ns = src.n, nt = tgt.n, n = ns + nt;
  if (ns == 0 || nt == 0) return O3DS_OK;  // no voxel can hold points of both clouds
  CloudRec moved;
  int rc = transform_t<P4>(h, src, T, moved);  // sourceTransformed.Transform(sourceToTarget)
  if (rc) {
    free_cloud(h, moved);
    return rc;
  }
  unsigned long long *k0 = nullptr, *k1 = nullptr, *d_out = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr;
  int *head = nullptr, *seg_id = nullptr, *cnt = nullptr, *flag = nullptr, *pos = nullptr;
  TMP_ALLOC(k0, sizeof(unsigned long long) * n);
  TMP_ALLOC(k1, sizeof(unsigned long long) * n);
  TMP_ALLOC(v0, sizeof(uint32_t) * n);
  TMP_ALLOC(v1, sizeof(uint32_t) * n);
  TMP_ALLOC(head, sizeof(int) * (n + 1));
  TMP_ALLOC(seg_id, sizeof(int) * (n + 1));
  CropDev none{};
  voxel_key_kernel<P4><<<grid_for(nt), kBlock, 0, h->stream>>>((const P4*)tgt.pts, nt, 1, 0.0, 0.0, 0.0, voxel, none, k0, v0);
  voxel_key_kernel<P4><<<grid_for(ns), kBlock, 0, h->stream>>>((const P4*)moved.pts, ns, 1, 0.0, 0.0, 0.0, voxel, none, k0 + nt, v0 + nt);
  overlap_tag_kernel<<<grid_for(ns), kBlock, 0, h->stream>>>(v0 + nt, ns);
  size_t temp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
  void* temp = nullptr;
  TMP_ALLOC(temp, temp_bytes ? temp_bytes : 16);
  HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
  HIP_TRY(hipMemsetAsync(head + n, 0, sizeof(int), h->stream));
  segment_head_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k1, n, head);
  rc = exclusive_scan_int(h, head, seg_id, n + 1);
  if (rc) {
    free_cloud(h, moved);
    return rc;
  }
  TMP_ALLOC(cnt, sizeof(int) * 2 * n);  // at most n segments: [0, n) source counts, [n, 2n) target counts
  HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * 2 * n, h->stream));
  overlap_count_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(v1, head, seg_id, n, cnt, cnt + n);
  TMP_ALLOC(flag, sizeof(int) * (n + 2));  // [0, ns] source flags (+ terminator), then [ns + 1, ns + 1 + nt] target flags
  TMP_ALLOC(pos, sizeof(int) * (n + 2));
  HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int) * (n + 2), h->stream));
  int* flag_s = flag;
  int* flag_t = flag + ns + 1;
  overlap_flag_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(v1, head, seg_id, n, cnt, cnt + n, (int)std::min<size_t>(min_points, 0x7fffffff), flag_s,
                                                             flag_t);
  rc = exclusive_scan_int(h, flag_s, pos, ns + 1);
  if (!rc) rc = exclusive_scan_int(h, flag_t, pos + ns + 1, nt + 1);
  free_cloud(h, moved);
  if (rc) return rc;
  int tot[2] = {0, 0};
  rc = read_back(h, {{&tot[0], pos + ns, sizeof(int)}, {&tot[1], pos + ns + 1 + nt, sizeof(int)}});
  if (rc) return rc;
  TMP_ALLOC(d_out, sizeof(unsigned long long) * (size_t)(tot[0] + tot[1] + 1));
  index_compact_kernel<<<grid_for(ns), kBlock, 0, h->stream>>>(flag_s, pos, ns, d_out);
  index_compact_kernel<<<grid_for(nt), kBlock, 0, h->stream>>>(flag_t, pos + ns + 1, nt, d_out + tot[0]);
  HIP_TRY(hipGetLastError());
  if (tot[0]) HIP_TRY(hipMemcpyAsync(idx_src, d_out, sizeof(unsigned long long) * (size_t)tot[0], hipMemcpyDeviceToHost, h->stream));
  if (tot[1]) HIP_TRY(hipMemcpyAsync(idx_tgt, d_out + tot[0], sizeof(unsigned long long) * (size_t)tot[1], hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *n_src_out = (size_t)tot[0];
  *n_tgt_out = (size_t)tot[1];
  return O3DS_OK;
}

template <typename P4>
int append_t(o3ds_handle h, CloudRec& map, const CloudRec& add) {
  // [O3D] PointCloud::operator+= : normals survive only if (map empty or map has normals) and add has normals
  const bool keep_nrm = (map.n == 0 || map.nrm) && add.nrm;
  const bool keep_col = (map.n == 0 || map.col) && add.col;  // the same rule for colors_
  const size_t n = map.n + add.n;
  CloudRec joined;
  box_union(joined, map, add);
  void *np = nullptr, *nn = nullptr, *nc = nullptr;
  DevGuard guard(h);
  guard.add(&np), guard.add(&nn), guard.add(&nc);
  if (n > 0) HIP_TRY(dev_alloc(h, (void**)&np, sizeof(P4) * n));
  if (keep_nrm && n > 0) HIP_TRY(dev_alloc(h, (void**)&nn, sizeof(P4) * n));
  if (keep_col && n > 0) HIP_TRY(dev_alloc(h, (void**)&nc, sizeof(P4) * n));
  if (map.n) {
    HIP_TRY(hipMemcpyAsync(np, map.pts, sizeof(P4) * map.n, hipMemcpyDeviceToDevice, h->stream));
    if (keep_nrm) HIP_TRY(hipMemcpyAsync(nn, map.nrm, sizeof(P4) * map.n, hipMemcpyDeviceToDevice, h->stream));
    if (keep_col) HIP_TRY(hipMemcpyAsync(nc, map.col, sizeof(P4) * map.n, hipMemcpyDeviceToDevice, h->stream));
  }
  if (add.n) {
    reindex_copy_kernel<P4><<<grid_for(add.n), kBlock, 0, h->stream>>>((const P4*)add.pts, add.n, (P4*)np, map.n, 1);
    if (keep_nrm) reindex_copy_kernel<P4><<<grid_for(add.n), kBlock, 0, h->stream>>>((const P4*)add.nrm, add.n, (P4*)nn, map.n, 0);
    if (keep_col) reindex_copy_kernel<P4><<<grid_for(add.n), kBlock, 0, h->stream>>>((const P4*)add.col, add.n, (P4*)nc, map.n, 0);
  }
  HIP_TRY(hipGetLastError());
  dbg_sync(h, 32);
  guard.release();
  free_index(h, map);
  free_points(h, map);
  drop_ingest_box(h, map);
  if (map.nrm) dev_free(h, map.nrm);
  if (map.col) dev_free(h, map.col);
  map.pts = np;
  map.nrm = nn;
  map.col = nc;
  map.n = n;
  map.cap = 0;
  box_copy(map, joined);
  return O3DS_OK;
}

// [O3D] PointCloud::RandomDownSample(ratio) (Odometry.cpp:29, ScanToMapRegistration.cpp:39) as a draw on the device (cloud_kernels.hpp,
// DrawState): nothing of it -- not the size of the input, not the number kept -- passes through the host
template <typename P4>
int random_down_sample_t(o3ds_handle h, const CloudRec& in, double ratio, unsigned long long seed, CloudRec& out) {
  out.precision = in.precision;
  out.n = 0;
  box_copy(out, in);  // a subset
  if (in.n == 0 || (int)(ratio * (double)in.n) <= 0) return O3DS_OK;  // (k is monotone in the size: nothing is kept of the bound, nothing of the exact size)
  if (in.n >= ((size_t)1 << 31)) return fail(h, O3DS_ERR_INVALID_ARG, "random_down_sample: more than 2^31 points");
  constexpr int kDrawErrSlot = 9;  // pinned word draw_pick_kernel raises when its candidate list overflowed (sticky, like DrawState::error)
  if (!h->d_draw) {
    if (hipMalloc((void**)&h->d_draw, sizeof(o3ds::DrawState)) != hipSuccess) return fail(h, O3DS_ERR_OOM, "random_down_sample: state allocation failed");
    HIP_TRY(hipMemsetAsync(h->d_draw, 0, sizeof(o3ds::DrawState), h->stream));
    *(volatile int*)(h->h_pin + kPubOff + 16 * (size_t)kDrawErrSlot) = 0;
  }
  if (pub_value<int>(h, kDrawErrSlot) != 0)
    return fail(h, O3DS_ERR_CAPACITY, "random_down_sample: an earlier draw on this handle found more than 2048 keys sharing 22 leading bits and kept nothing");
  o3ds::DrawState* st = h->d_draw;
  // The number kept is published like a VoxelDownSample's -- also when the input's size is known and the host could compute it: the record
  // is how a draw that failed on the device (draw_pick_kernel) reaches whoever asks for the result's size.
  // (the record FIRST: with all of them held, take_rec settles the oldest holder -- possibly `in`, whose size is exact from then on and
  // whose device word may be the very record handed out here; everything below reads `in` as it is afterwards)
  out.lazy_slot = take_rec(h);
  if (out.lazy_slot < 0) return fail(h, O3DS_ERR_CAPACITY, "random_down_sample: no record free for the size of the result");
  out.lazy_seq = ++h->rec_seq;
  const CountPub pub{cnt_word(h, out.lazy_slot), &(h->h_rec_dev + out.lazy_slot)->cnt, out.lazy_seq};
  const size_t bound = in.n;  // (an upper bound while the input's size is in flight)
  const size_t k_bound = (size_t)std::max(0, (int)(ratio * (double)bound));  // monotone in the size: at least the exact number
  const CountRef n_ref = count_ref(h, in);
  out.n = k_bound;
  out.n_lower = in.lazy_slot >= 0 ? (size_t)std::max(0, (int)(ratio * (double)in.n_lower)) : k_bound;
  if (k_bound == 0) {  // (the settled size keeps nothing)
    h->rec_owner[out.lazy_slot] = 0, out.lazy_slot = -1;
    return O3DS_OK;
  }
  int *flags = nullptr, *pos = nullptr;
  TMP_ALLOC(flags, sizeof(int) * (bound + 1));
  TMP_ALLOC(pos, sizeof(int) * (bound + 1));
  HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * k_bound));
  if (in.nrm) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * k_bound));
  if (in.col) HIP_TRY(dev_alloc(h, (void**)&out.col, sizeof(P4) * k_bound));
  const unsigned int g = grid_for(bound);
  // (O3DS_DRAW_LIST_CAP in the A/B library: a smaller list, so that the overflow path can be walked by a test)
  // (read at every call, not once per process: the test that sets it shares its process with tests that draw before it)
  const char* cap_env = ab_getenv("O3DS_DRAW_LIST_CAP");
  const unsigned int list_cap = cap_env ? (unsigned int)std::max(0, std::min(atoi(cap_env), o3ds::kDrawListCap)) : (unsigned int)o3ds::kDrawListCap;
  draw_hist1_kernel<<<g, kBlock, 0, h->stream>>>(n_ref, seed, st);
  draw_hist2_kernel<<<g, kBlock, 0, h->stream>>>(n_ref, seed, ratio, st);
  draw_collect_kernel<<<g, kBlock, 0, h->stream>>>(n_ref, seed, st, list_cap);
  draw_pick_kernel<<<1, kBlock, 0, h->stream>>>(st, pub, pub_slot<int>(h, kDrawErrSlot), list_cap);
  draw_flag_kernel<<<grid_for(bound + 1), kBlock, 0, h->stream>>>(n_ref, seed, st, flags);
  int rc = exclusive_scan_int(h, flags, pos, bound + 1);
  if (rc) return rc;
  compact_kernel<P4><<<g, kBlock, 0, h->stream>>>((const P4*)in.pts, (const P4*)in.nrm, bound, flags, pos, 1, (P4*)out.pts, (P4*)out.nrm);
  if (in.col) compact_kernel<P4><<<g, kBlock, 0, h->stream>>>((const P4*)in.col, nullptr, bound, flags, pos, 1, (P4*)out.col, nullptr);
  HIP_TRY(hipGetLastError());
  dbg_sync(h, 4);
  return O3DS_OK;
}

#define DISPATCH(prec, fn, ...) ((prec) == O3DS_PRECISION_F64 ? fn<P4d>(__VA_ARGS__) : fn<P4f>(__VA_ARGS__))

}  // namespace

extern "C" {

int o3ds_crop_cloud(o3ds_handle h, o3ds_cloud in, const o3ds_crop* crop, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, in);
  if (!c || !out) return fail(h, O3DS_ERR_INVALID_ARG, "crop_cloud: bad argument");
  CloudRec o;
  const CropDev cd = to_dev(crop);
  int rc = DISPATCH(c->precision, crop_t, h, *c, cd, o);
  if (rc) {
    free_cloud(h, o);
    return rc;
  }
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_voxel_down_sample(o3ds_handle h, o3ds_cloud in, double voxel_size, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, in);
  if (!c || !out) return fail(h, O3DS_ERR_INVALID_ARG, "voxel_down_sample: bad argument");
  CloudRec o;
  int rc;
  if (voxel_size <= 0.0) {  // o3d_slam::voxelize returns the cloud unchanged (helpers.cpp:108-110)
    o.precision = c->precision;
    rc = DISPATCH(c->precision, append_t, h, o, *c);
  } else {
    CropDev none{};
    rc = DISPATCH(c->precision, voxel_reduce_t, h, *c, 0, voxel_size, none, o);
  }
  if (rc) {
    free_cloud(h, o);
    return rc;
  }
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_crop_voxel_down_sample(o3ds_handle h, o3ds_cloud in, const o3ds_crop* crop, double voxel_size, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  if (!(voxel_size > 0.0)) return o3ds_crop_cloud(h, in, crop, out);  // o3d_slam::voxelize leaves the cropped cloud as it is
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, in);
  if (!c || !out) return fail(h, O3DS_ERR_INVALID_ARG, "crop_voxel_down_sample: bad argument");
  CloudRec o;
  const CropDev cd = to_dev(crop);
  const int rc = DISPATCH(c->precision, voxel_reduce_t, h, *c, 0, voxel_size, cd, o, true);
  if (rc) {
    free_cloud(h, o);
    return rc;
  }
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_estimate_normals(o3ds_handle h, o3ds_cloud id, double radius, int max_nn) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud_lazy(h, id);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "estimate_normals: unknown cloud id");
  if (!(radius > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "maxRadiusNormalEstimation_ must be > 0");  // CloudRegistration.cpp:50
  if (max_nn <= 0) return fail(h, O3DS_ERR_INVALID_ARG, "knnNormalEstimation_ must be > 0");              // CloudRegistration.cpp:51
  if (max_nn > 128) return fail(h, O3DS_ERR_INVALID_ARG, "estimate_normals: max_nn > 128 unsupported");
  if (c->pm) {  // normals_t reads pts[0..n) as an array: a persistent submap is folded into the reference's array first (o3ds_backend.h)
    const int re = pm_exit(h, *c);
    if (re) return re;
  }
  return DISPATCH(c->precision, normals_t, h, *c, radius, max_nn);
}

int o3ds_select_by_index(o3ds_handle h, o3ds_cloud in, const uint32_t* keep_idx, size_t m, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, in);
  if (!c || !out || (m && !keep_idx)) return fail(h, O3DS_ERR_INVALID_ARG, "select_by_index: bad argument");
  for (size_t i = 0; i < m; ++i)
    if (keep_idx[i] >= c->n) return fail(h, O3DS_ERR_INVALID_ARG, "select_by_index: index out of range");
  CloudRec o;
  CloudGuard o_guard(h, o);
  o.precision = c->precision;
  o.n = m;
  box_copy(o, *c);  // a subset
  if (m) {
    uint32_t* d_idx = nullptr;
    const size_t psz = p4_size(c->precision);
    TMP_ALLOC(d_idx, sizeof(uint32_t) * m);
    HIP_TRY(hipMemcpyAsync(d_idx, keep_idx, sizeof(uint32_t) * m, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(dev_alloc(h, (void**)&o.pts, psz * m));
    if (c->nrm) HIP_TRY(dev_alloc(h, (void**)&o.nrm, psz * m));
    if (c->precision == O3DS_PRECISION_F64)
      gather_kernel<P4d><<<grid_for(m), kBlock, 0, h->stream>>>((const P4d*)c->pts, (const P4d*)c->nrm, d_idx, m, (P4d*)o.pts, (P4d*)o.nrm);
    else
      gather_kernel<P4f><<<grid_for(m), kBlock, 0, h->stream>>>((const P4f*)c->pts, (const P4f*)c->nrm, d_idx, m, (P4f*)o.pts, (P4f*)o.nrm);
    if (c->col) {
      HIP_TRY(dev_alloc(h, (void**)&o.col, psz * m));
      if (c->precision == O3DS_PRECISION_F64)
        gather_kernel<P4d><<<grid_for(m), kBlock, 0, h->stream>>>((const P4d*)c->col, nullptr, d_idx, m, (P4d*)o.col, nullptr);
      else
        gather_kernel<P4f><<<grid_for(m), kBlock, 0, h->stream>>>((const P4f*)c->col, nullptr, d_idx, m, (P4f*)o.col, nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));  // keep_idx may be released by the caller
  }
  o_guard.release();
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_random_down_sample(o3ds_handle h, o3ds_cloud in, double ratio, uint64_t seed, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud_lazy(h, in);
  if (!c || !out) return fail(h, O3DS_ERR_INVALID_ARG, "random_down_sample: bad argument");
  if (c->pm) return fail(h, O3DS_ERR_INVALID_ARG, "random_down_sample: the cloud is a map");
  if (!(ratio >= 0.0 && ratio <= 1.0))
    return fail(h, O3DS_ERR_INVALID_ARG, "[RandomDownSample] Illegal sampling_ratio, sampling_ratio must be between 0 and 1.");  // [O3D]
  HIP_TRY(hipSetDevice(h->device));
  CloudRec o;
  CloudGuard o_guard(h, o);
  int rc = DISPATCH(c->precision, random_down_sample_t, h, *c, ratio, (unsigned long long)seed, o);
  if (rc) return rc;
  o_guard.release();
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_transform_cloud(o3ds_handle h, o3ds_cloud in, const double T[16], o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, in);
  if (!c || !out || !T) return fail(h, O3DS_ERR_INVALID_ARG, "transform_cloud: bad argument");
  CloudRec o;
  int rc = DISPATCH(c->precision, transform_t, h, *c, T, o);
  if (rc) {
    free_cloud(h, o);
    return rc;
  }
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_cloud_append(o3ds_handle h, o3ds_cloud map, o3ds_cloud add) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* m = find_cloud(h, map);
  CloudRec* a = find_cloud(h, add);
  if (!m || !a || m == a) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_append: bad cloud id");
  if (m->n && a->n && m->precision != a->precision) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_append: precision mismatch");
  if (m->n == 0) m->precision = a->precision;
  return DISPATCH(m->precision, append_t, h, *m, *a);
}

int o3ds_cloud_copy_across(o3ds_handle dst, o3ds_handle src, o3ds_cloud src_cloud, o3ds_cloud* out) {
  o3ds_handle h = dst;
  CHECK_HANDLE(h);
  if (!src || !out) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_copy_across: bad argument");
  if (src == dst) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_copy_across: source and destination handle are the same (use the cloud itself)");
  if (src->device != dst->device) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_copy_across: the handles live on different devices");
  CloudRec* c = find_cloud(src, src_cloud);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_copy_across: unknown cloud id on the source handle");
  if (c->n && c->precision != dst->precision) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_copy_across: the handles store points in different precisions");
  CloudRec o;
  CloudGuard o_guard(h, o);
  o.precision = c->n ? c->precision : dst->precision;
  o.n = c->n;
  box_copy(o, *c);
  if (c->n) {
    // what the source handle queued for this cloud must have happened before the copy reads it: the destination stream waits for
    // an event on the source stream (no host wait); the source cloud must stay alive until the destination stream passed the copy,
    // which the synchronisation below guarantees before this returns
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, src->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(dst->stream, ev, 0);
    (void)hipEventDestroy(ev);  // released once it has completed
    HIP_TRY(e);
    const size_t bytes = p4_size(c->precision) * c->n;
    HIP_TRY(dev_alloc(h, (void**)&o.pts, bytes));
    HIP_TRY(hipMemcpyAsync(o.pts, c->pts, bytes, hipMemcpyDeviceToDevice, h->stream));
    if (c->nrm) {
      HIP_TRY(dev_alloc(h, (void**)&o.nrm, bytes));
      HIP_TRY(hipMemcpyAsync(o.nrm, c->nrm, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    if (c->col) {
      HIP_TRY(dev_alloc(h, (void**)&o.col, bytes));
      HIP_TRY(hipMemcpyAsync(o.col, c->col, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  o_guard.release();
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_cloud_export_view(o3ds_handle h, o3ds_cloud id, o3ds_cloud_view* out) {
  CHECK_HANDLE(h);
  if (!out) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_export_view: null view");
  memset(out, 0, sizeof(*out));
  CloudRec* c = find_cloud(h, id);  // (the exact size; a persistent map is folded: a view is of arrays)
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_export_view: unknown cloud id");
  HIP_TRY(hipSetDevice(h->device));
  out->pts = c->pts, out->nrm = c->nrm, out->col = c->col;
  out->n = c->n;
  out->precision = c->n ? c->precision : h->precision;
  out->device = h->device;
  out->has_box = c->has_box ? (c->box_padded ? 2 : 1) : 0;
  for (int a = 0; a < 3; ++a) out->box_min[a] = c->bmn[a], out->box_max[a] = c->bmx[a];
  if (c->n) {  // behind everything this handle has queued for the cloud so far
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const hipError_t e = hipEventRecord(ev, h->stream);
    if (e != hipSuccess) {
      (void)hipEventDestroy(ev);
      HIP_TRY(e);
    }
    out->event = ev;
  }
  return O3DS_OK;
}

int o3ds_cloud_view_release(o3ds_cloud_view* view) {
  if (view && view->event) (void)hipEventDestroy((hipEvent_t)view->event);
  if (view) memset(view, 0, sizeof(*view));
  return O3DS_OK;
}

int o3ds_cloud_import_view(o3ds_handle h, const o3ds_cloud_view* v, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  if (!v || !out) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_import_view: null argument");
  if (v->device != h->device) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_import_view: the view's cloud lives on another device");
  if (v->n && v->precision != h->precision) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_import_view: the view's cloud is stored in another precision");
  if (v->n && (!v->pts || !v->event)) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_import_view: a released or incomplete view");
  HIP_TRY(hipSetDevice(h->device));
  CloudRec o;
  CloudGuard o_guard(h, o);
  o.precision = h->precision;
  o.n = v->n;
  o.has_box = v->has_box != 0;
  o.box_padded = v->has_box == 2;
  for (int a = 0; a < 3; ++a) o.bmn[a] = v->box_min[a], o.bmx[a] = v->box_max[a];
  if (v->n) {
    HIP_TRY(hipStreamWaitEvent(h->stream, (hipEvent_t)v->event, 0));  // what the owner queued for the cloud before it exported the view
    const size_t bytes = p4_size(h->precision) * v->n;
    HIP_TRY(dev_alloc(h, (void**)&o.pts, bytes));
    HIP_TRY(hipMemcpyAsync(o.pts, v->pts, bytes, hipMemcpyDeviceToDevice, h->stream));
    if (v->nrm) {
      HIP_TRY(dev_alloc(h, (void**)&o.nrm, bytes));
      HIP_TRY(hipMemcpyAsync(o.nrm, v->nrm, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    if (v->col) {
      HIP_TRY(dev_alloc(h, (void**)&o.col, bytes));
      HIP_TRY(hipMemcpyAsync(o.col, v->col, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));  // the owner may free its cloud as soon as this returns
  }
  o_guard.release();
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

}  // extern "C" (a helper with default arguments follows)
namespace {
// voxelizeWithinCroppingVolume of a device cloud in place; merge_np >= 0: the cloud is [pass block | voxel block in key order | new points]
int voxelize_within_volume_impl(o3ds_handle h, o3ds_cloud map, double voxel_size, const o3ds_crop* crop, long long merge_np = -1, size_t merge_nv = 0) {
  CloudRec* m = find_cloud(h, map);
  if (!m) return fail(h, O3DS_ERR_INVALID_ARG, "voxelize_within_volume: unknown cloud id");
  if (voxel_size <= 0.0 || m->n == 0) return O3DS_OK;  // helpers.cpp:119-123 / Submap.cpp:139: unchanged
  CloudRec o;
  const CropDev cd = to_dev(crop);
  int rc = m->precision == O3DS_PRECISION_F64 ? voxel_reduce_t<P4d>(h, *m, 1, voxel_size, cd, o, false, merge_np, merge_nv)
                                              : voxel_reduce_t<P4f>(h, *m, 1, voxel_size, cd, o, false, merge_np, merge_nv);
  if (rc) {
    free_cloud(h, o);
    return rc;
  }
  free_cloud(h, *m);
  *m = o;
  return O3DS_OK;
}
}  // namespace

// ---- the persistent form of a submap (map_kernels.hpp): enter, insert, leave ------------------------------------------------------------
namespace {

#define PM_ALLOC(ptr, bytes)                                                                               \
  do {                                                                                                     \
    void* _p = nullptr;                                                                                    \
    if (dev_alloc(h, &_p, (size_t)(bytes)) != hipSuccess) return fail(h, O3DS_ERR_OOM, "persistent map: out of device memory"); \
    pm->blocks.push_back(_p);                                                                              \
    (ptr) = (decltype(ptr))_p;                                                                             \
  } while (0)

// cell edge of the row-paged index in voxels: the classic index uses max_corr / 4; here a whole number of voxels
int pm_cell_voxels(double max_corr_hint, double voxel) {
  const double want = max_corr_hint > 0.0 ? max_corr_hint / index_cell_div() : 2.5 * voxel;
  return std::max(1, (int)std::floor(want / voxel + 0.5));
}

// The map `c` -- an array [pass-through block | voxel block in key order] (vox_first / vox_count) with normals or without -- enters the
// persistent form.  O(N): history and voxel hash of every point, the row-paged search index.
template <typename P4>
int pm_enter_t(o3ds_handle h, CloudRec& c, double voxel, double max_corr_hint, size_t scan_upper) {
  const size_t n = c.n;
  PMapRec* pm = new PMapRec();
  c.pm = pm;  // (from here on free_cloud / pm_release own it)
  // room: half the map again and a few scans' worth
  const size_t want_cap = n + std::max<size_t>({n / 2, 16 * scan_upper, (size_t)1 << 18});
  if (c.cap < want_cap) {  // (entering costs a pass over the map anyway; room for half the map again makes the next capacity fold a rare event)
    void *np = nullptr, *nn = nullptr;
    HIP_TRY(dev_alloc(h, &np, sizeof(P4) * want_cap));
    HIP_TRY(hipMemcpyAsync(np, c.pts, sizeof(P4) * n, hipMemcpyDeviceToDevice, h->stream));
    if (c.nrm) {
      if (dev_alloc(h, &nn, sizeof(P4) * want_cap) != hipSuccess) {
        dev_free(h, np);
        return fail(h, O3DS_ERR_OOM, "persistent map: out of device memory");
      }
      HIP_TRY(hipMemcpyAsync(nn, c.nrm, sizeof(P4) * n, hipMemcpyDeviceToDevice, h->stream));
    }
    free_points(h, c);
    if (c.nrm) dev_free(h, c.nrm);
    c.pts = np;
    c.nrm = nn;
    c.cap = want_cap;
  }
  const size_t cap = c.cap;
  PmDev& d = pm->dev;
  d.pts = c.pts;
  d.nrm = c.nrm;
  d.cap = cap;
  PM_ALLOC(d.slot, sizeof(PmSlot) * cap);
  size_t hcap = 1024;
  while (hcap < 2 * cap) hcap <<= 1;
  PM_ALLOC(d.h, sizeof(PmHash) * hcap);
  d.hmask = (unsigned int)(hcap - 1);
  pm_hash_init_kernel<<<grid_for(hcap), kBlock, 0, h->stream>>>(d.h, hcap);
  d.list_cap = (int)std::min<size_t>(cap, 0x7fffffff);
  PM_ALLOC(d.counters, sizeof(int) * kPmCounters);
  for (int k = 0; k < 2; ++k) PM_ALLOC(d.unsettled[k], sizeof(int) * cap);
  PM_ALLOC(d.multi[0], sizeof(unsigned long long) * cap);
  d.multi[1] = nullptr;
  PM_ALLOC(d.complex_groups, sizeof(int) * cap);
  PM_ALLOC(d.outside_pts, sizeof(int) * cap);
  PM_ALLOC(d.relink, sizeof(int) * cap);
  PM_ALLOC(d.touched_rows, sizeof(int) * cap);
  PM_ALLOC(d.new_slots, sizeof(int) * cap);
  PM_ALLOC(pm->d_hist, sizeof(CropDev) * (kPmHistory + 1));
  d.hist = pm->d_hist;
  d.n_base = (int)n;
  d.np_base = (int)c.vox_first;
  d.voxel = voxel;
  d.inv_voxel = 1.0 / voxel;
  // the grid of the index: aligned with the voxel grid, over the map's box with a margin (points beyond it are clamped into the border cells,
  // as build_grid_t's cell_of does: slower to search, never wrong)
  if (!c.has_box) {
    double mn[3], mx[3];
    const int rb = bbox_of<P4>(h, (const P4*)c.pts, n, mn, mx);
    if (rb) return rb;
    c.has_box = true;
    c.box_padded = false;
    for (int a = 0; a < 3; ++a) c.bmn[a] = mn[a], c.bmx[a] = mx[a];
  }
  for (int a = 0; a < 3; ++a)
    if (!std::isfinite(c.bmn[a]) || !std::isfinite(c.bmx[a])) return fail(h, O3DS_ERR_INVALID_ARG, "persistent map: non-finite coordinates");
  d.kc = pm_cell_voxels(max_corr_hint, voxel);
  double cell = d.kc * voxel;
  long long g0[3], gn[3];
  for (;;) {
    for (int a = 0; a < 3; ++a) {
      const double ext = c.bmx[a] - c.bmn[a];
      const double pad = a < 2 ? std::max(5.0, 0.1 * ext) : std::max(2.0, 0.1 * ext);
      const long long lo = (long long)std::floor((c.bmn[a] - pad) / cell), hi = (long long)std::floor((c.bmx[a] + pad) / cell);
      g0[a] = lo * d.kc;
      gn[a] = hi - lo + 1;
    }
    if ((double)gn[0] * (double)gn[1] * (double)gn[2] <= (double)kMaxCells / 2 && gn[0] < 5000) break;  // (pm_rows_kernel keeps three ints per cell of a row in LDS)
    d.kc *= 2;
    cell = d.kc * voxel;
  }
  for (int a = 0; a < 3; ++a)
    if (std::llabs(g0[a]) + gn[a] * d.kc >= (1ll << 20)) return fail(h, O3DS_ERR_INVALID_ARG, "persistent map: voxel coordinates beyond 2^20");
  d.gx0 = g0[0], d.gy0 = g0[1], d.gz0 = g0[2];
  GridDev g{};
  g.ox = (double)g0[0] * voxel;
  g.oy = (double)g0[1] * voxel;
  g.oz = (double)g0[2] * voxel;
  g.cell = cell;
  g.inv_cell = 1.0 / cell;
  g.nx = (int)gn[0], g.ny = (int)gn[1], g.nz = (int)gn[2];
  g.sx = g.nx + 1;
  pm->rows = (size_t)g.ny * g.nz;
  const size_t table = pm->rows * (size_t)g.sx;
  // the arrays of the index hang on the cloud (free_index): the registration uses them like any index
  free_index(h, c);
  int* cs = nullptr;
  HIP_TRY(dev_alloc(h, (void**)&cs, sizeof(int) * (table + 1 + 4)));
  c.cell_start = cs;
  const size_t pool = 6 * n + 16 * scan_upper + ((size_t)1 << 22);
  if (pool >= ((size_t)1 << 31)) return fail(h, O3DS_ERR_OOM, "persistent map: index pool beyond 2^31 positions");
  HIP_TRY(dev_alloc(h, &c.spts, sizeof(P4) * pool));
  if (c.nrm) HIP_TRY(dev_alloc(h, &c.snrm, sizeof(P4) * pool));
  PM_ALLOC(d.row_cap, sizeof(int) * pm->rows);
  PM_ALLOC(d.row_flag, sizeof(int) * pm->rows);
  PM_ALLOC(d.cell_add, sizeof(int) * (table + 1));
  HIP_TRY(hipMemsetAsync(d.cell_add, 0, sizeof(int) * (table + 1), h->stream));
  g.cell_start = cs;
  d.grid = g;
  d.cs = cs;
  d.spts = c.spts;
  d.snrm = c.snrm;
  d.pool_cap = (int)pool;
  // counters: the handle's block of zeros (build_grid_t's: counted up, scanned, counted back down to zero by the scatter)
  if (h->cells_cap < table + 1) {
    if (h->d_cells) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_cells);
      h->d_cells = nullptr;
      h->cells_cap = 0;
    }
    const size_t want = table + 1 + table / 4;
    if (hipMalloc((void**)&h->d_cells, sizeof(int) * want) != hipSuccess) return fail(h, O3DS_ERR_OOM, "persistent map: cell counters allocation failed");
    h->cells_cap = want;
    h->cells_clean = false;
  }
  if (!h->cells_clean) HIP_TRY(hipMemsetAsync(h->d_cells, 0, sizeof(int) * h->cells_cap, h->stream));
  h->cells_clean = false;
  HIP_TRY(hipMemsetAsync(d.counters, 0, sizeof(int) * kPmCounters, h->stream));
  HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(d.counters + kPmN), (int)n, 1, h->stream));
  int* cell_id = nullptr;
  TMP_ALLOC(cell_id, sizeof(int) * n);
  const int ni = (int)n;
  span_mark(h, kSpanIndexBuild);
  pm_enter_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(d, ni);
  pm_cell_count_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(d, ni, h->d_cells, cell_id);
  pm_row_slack_kernel<<<grid_for(pm->rows * 64), kBlock, 0, h->stream>>>(h->d_cells, (int)pm->rows, g.nx, d.row_cap, d.row_flag);
  int rc = exclusive_scan_int(h, h->d_cells, cs, table + 1);
  if (rc) return rc;
  pm_row_finish_kernel<<<grid_for(pm->rows), kBlock, 0, h->stream>>>(h->d_cells, cs, (int)pm->rows, g.nx, d.counters + kPmPoolTop);
  pm_scatter_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(d, ni, cell_id, h->d_cells);
  // (the voxels with several members that pm_enter_kernel listed are what the first insertion looks at)
  HIP_TRY(hipMemcpyAsync(d.counters + kPmMultiIn, d.counters + kPmMultiOut, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  span_mark(h, kSpanIndexBuild);
  HIP_TRY(hipGetLastError());
  h->cells_clean = true;
  c.grid = g;
  c.has_index = true;
  c.index_byproduct = false;
  c.index_positions = pool;
  pm->t = 0;
  pm->n_upper = n;
  pm->live_lower = n;
  pm->pool_top = (double)(n + n / 2 + 8 * pm->rows);  // (an upper bound until the first record arrives: count + slack of every row)
  pm->rec_slot = take_rec(h);
  if (pm->rec_slot < 0) return fail(h, O3DS_ERR_CAPACITY, "persistent map: no record free for its counters");
  h->rec_owner[pm->rec_slot] = kRecOwnerMap;
  pm->rec_seq = 0;
  return O3DS_OK;
}

// what the latest record the insertions published says, if it has arrived (never waits unless `block`)
int pm_poll(o3ds_handle h, PMapRec* pm, bool block) {
  if (pm->rec_seq == 0) return O3DS_OK;
  volatile o3ds_context::PinRec* r = h->h_rec + pm->rec_slot;
  if (r->seq != pm->rec_seq) {
    if (!block) return O3DS_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (r->seq != pm->rec_seq) return fail(h, O3DS_ERR_HIP, "persistent map: an insertion never published its record");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const size_t slots = (size_t)r->cnt, dead = (size_t)r->box[1];
  if ((int)r->box[2] != 0) return fail(h, O3DS_ERR_CAPACITY, "persistent map: an internal capacity was exceeded (error " + std::to_string((int)r->box[2]) + ")");
  static const bool stats = ab_getenv("O3DS_PM_STATS") != nullptr;  // development aid: what the insertions left behind
  if (stats)
    fprintf(stderr, "[pm] t %d slots %zu dead %zu pool top %.0f of %d; multi %.0f complex %.0f outside the grid %.0f\n", pm->t, slots, dead, r->box[0],
            pm->dev.pool_cap, r->box[3], r->box[4], r->box[5]);
  pm->n_upper = slots;
  pm->live_lower = slots - dead;
  pm->pool_top = r->box[0];
  pm->clamped = r->box[5];
  pm->multi_total = std::max(pm->multi_total, (double)r->box[3]);  // (a carve's record repeats the list's count; its removed points travel in PinRec::pad)
  pm->rec_seq = 0;  // consumed
  return O3DS_OK;
}

// Submap::insertScan's map += T * scan; voxelizeWithinCroppingVolume on the persistent form: six launches over the scan, nothing returns
template <typename P4>
int pm_insert_t(o3ds_handle h, CloudRec& c, const CloudRec& scan, const double T[16], const CropDev& crop) {
  PMapRec* pm = c.pm;
  PmDev d = pm->dev;
  const size_t ms = scan.n;  // (an upper bound when the scan's size is still in flight)
  const bool has_nrm = c.nrm != nullptr;
  const int t_now = pm->t + 1;
  // scratch: the voxel table of VoxelDownSample, the chained scan's tiles
  size_t tcap = 1024;
  while (tcap < 2 * ms) tcap <<= 1;
  if (h->voxtab_cap < tcap) {
    if (h->d_voxtab) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_voxtab);
      h->d_voxtab = nullptr;
      h->voxtab_cap = 0;
    }
    if (hipMalloc((void**)&h->d_voxtab, kVoxSlotBytes * tcap + 16) != hipSuccess) return fail(h, O3DS_ERR_OOM, "map_insert_scan: voxel table allocation failed");
    h->voxtab_cap = tcap;
    h->voxtab_clean = false;
  }
  tcap = h->voxtab_cap;
  unsigned char* tab = h->d_voxtab;
  VoxTable t{(VoxSlot*)tab, (unsigned int*)(tab + kVoxSlotBytes * tcap), (unsigned int)(tcap - 1)};
  if (!h->voxtab_clean) HIP_TRY(hipMemsetAsync(tab, 0xff, kVoxSlotBytes * tcap + 16, h->stream));
  h->voxtab_clean = false;
  const size_t n_tiles = (ms + kVoxTile - 1) / kVoxTile;
  if (h->tiles_cap < n_tiles) {
    if (h->d_tiles) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_tiles);
      h->d_tiles = nullptr;
      h->tiles_cap = 0;
    }
    const size_t want = std::max<size_t>(n_tiles + n_tiles / 4, 1024);
    if (hipMalloc((void**)&h->d_tiles, sizeof(unsigned long long) * want) != hipSuccess) return fail(h, O3DS_ERR_OOM, "map_insert_scan: scan state allocation failed");
    HIP_TRY(hipMemsetAsync(h->d_tiles, 0, sizeof(unsigned long long) * want, h->stream));
    h->tiles_cap = want;
  }
  P4 *placed = nullptr, *placed_nrm = nullptr;
  int *lead_slot = nullptr, *run_next = nullptr, *run_len = nullptr, *order = nullptr, *d_groups = nullptr;
  uint32_t* starts = nullptr;
  int2* piece = nullptr;
  unsigned long long* group_key = nullptr;
  TMP_ALLOC(placed, sizeof(P4) * ms);
  if (has_nrm) TMP_ALLOC(placed_nrm, sizeof(P4) * ms);
  TMP_ALLOC(lead_slot, sizeof(int) * ms);
  TMP_ALLOC(run_next, sizeof(int) * ms);
  TMP_ALLOC(run_len, sizeof(int) * ms);
  TMP_ALLOC(order, sizeof(int) * ms);
  TMP_ALLOC(starts, sizeof(uint32_t) * ms);
  TMP_ALLOC(piece, sizeof(int2) * ms);
  TMP_ALLOC(group_key, sizeof(unsigned long long) * ms);
  TMP_ALLOC(d_groups, sizeof(int) * 16);
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 4; ++cc) M.m[r * 4 + cc] = T[cc * 4 + r];
  if (++h->scan_gen == 0) h->scan_gen = 1;
  const CountRef n_scan = count_ref(h, scan);
  const CountPub groups_pub{d_groups, nullptr, 0};
  const CountRef groups{ms, d_groups};
  pm_place_kernel<P4><<<grid_for(ms), kBlock, 0, h->stream>>>((const P4*)scan.pts, has_nrm ? (const P4*)scan.nrm : nullptr, n_scan, M, T[3], T[7], T[11], T[15], crop, d, t,
                                                            placed, placed_nrm, lead_slot, run_next, run_len);
  vox_order_kernel<<<(unsigned int)n_tiles, kBlock, 0, h->stream>>>(lead_slot, n_scan, t, h->d_tiles, h->d_ticket, h->ticket_base, h->scan_gen, order, groups_pub);
  h->ticket_base += (unsigned int)n_tiles;
  pm_group_kernel<P4><<<grid_for(ms), kBlock, 0, h->stream>>>(d, groups, order, run_next, run_len, starts, piece, placed, placed_nrm, t, crop, t_now, group_key);
  h->voxtab_clean = true;
  pm_merge_kernel<P4><<<512, 64, 0, h->stream>>>(d, piece, run_next, run_len, placed, placed_nrm, group_key, crop, t_now);
  pm_misc_kernel<P4><<<256, kBlock, 0, h->stream>>>(d, placed, placed_nrm, crop, t_now);
  const unsigned int row_blocks = (unsigned int)std::min<size_t>(std::max<size_t>(ms / 8, 64), 2048);
  pm_rows_kernel<P4><<<row_blocks, kBlock, sizeof(int) * 3 * (size_t)(d.grid.nx + 1), h->stream>>>(d);
  pm_place_new_kernel<P4><<<grid_for(ms), kBlock, 0, h->stream>>>(d);
  pm->rec_seq = ++h->rec_seq;
  o3ds_context::PinRec* rec = h->h_rec_dev + pm->rec_slot;
  const CountPub pub{cnt_word(h, pm->rec_slot), &rec->cnt, pm->rec_seq};
  pm_turn_kernel<<<1, 64, 0, h->stream>>>(d, pub, pm->d_hist, crop, t_now, rec->box);
  HIP_TRY(hipGetLastError());
  // the lists trade places for the next insertion
  std::swap(pm->dev.unsettled[0], pm->dev.unsettled[1]);
  pm->t = t_now;
  pm->n_upper += ms;
  c.n = pm->n_upper;
  return O3DS_OK;
}

// the reference's array back: [pass-through block in original order | voxel block in key order], then the ordinary cloud it was
template <typename P4>
int pm_exit_t(o3ds_handle h, CloudRec& c) {
  PMapRec* pm = c.pm;
  int rc = pm_poll(h, pm, true);
  if (rc) return rc;
  int counters[kPmCounters];
  rc = read_back(h, {{counters, pm->dev.counters, sizeof(counters)}});
  if (rc) return rc;
  if (counters[kPmError]) return fail(h, O3DS_ERR_CAPACITY, "persistent map: an internal capacity was exceeded (error " + std::to_string(counters[kPmError]) + ")");
  const size_t n = (size_t)counters[kPmN], live = n - (size_t)counters[kPmDeadCnt];
  unsigned long long *hi = nullptr, *lo = nullptr, *k1 = nullptr, *d_np = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr;
  TMP_ALLOC(hi, sizeof(unsigned long long) * n);
  TMP_ALLOC(lo, sizeof(unsigned long long) * n);
  TMP_ALLOC(k1, sizeof(unsigned long long) * n);
  TMP_ALLOC(v0, sizeof(uint32_t) * n);
  TMP_ALLOC(v1, sizeof(uint32_t) * n);
  TMP_ALLOC(d_np, sizeof(unsigned long long));
  HIP_TRY(hipMemsetAsync(d_np, 0, sizeof(unsigned long long), h->stream));
  pm_view_key_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(pm->dev, (int)n, pm->t, hi, lo, v0, d_np);
  // order by (hi, lo): two stable sorts, least significant key first
  size_t tb = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, lo, k1, v0, v1, n, 0, 64, h->stream));
  void* tmp = nullptr;
  TMP_ALLOC(tmp, tb ? tb : 16);
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, lo, k1, v0, v1, n, 0, 64, h->stream));
  pm_gather_u64_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(hi, v1, n, lo);  // lo := hi in the order of the first sort
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tb, lo, k1, v1, v0, n, 0, 64, h->stream));
  unsigned long long n_pass = 0;
  rc = read_back(h, {{&n_pass, d_np, sizeof(n_pass)}});
  if (rc) return rc;
  const size_t room = live + std::max<size_t>(live / 4, (size_t)1 << 18);
  void *np = nullptr, *nn = nullptr;
  if (live > 0) {
    HIP_TRY(dev_alloc(h, &np, sizeof(P4) * room));
    if (c.nrm && dev_alloc(h, &nn, sizeof(P4) * room) != hipSuccess) {
      dev_free(h, np);
      return fail(h, O3DS_ERR_OOM, "persistent map: out of device memory");
    }
    pm_permute_kernel<P4><<<grid_for(live), kBlock, 0, h->stream>>>((const P4*)c.pts, (const P4*)c.nrm, v0, live, (P4*)np, (P4*)nn);
    HIP_TRY(hipGetLastError());
  }
  pm_release(h, c);
  free_index(h, c);
  free_points(h, c);
  if (c.nrm) dev_free(h, c.nrm);
  c.pts = np;
  c.nrm = nn;
  c.n = live;
  c.cap = live > 0 ? room : 0;
  c.vox_first = (long long)n_pass;
  c.vox_count = live - (size_t)n_pass;
  return O3DS_OK;
}

}  // namespace

namespace {
int pm_exit(o3ds_handle h, CloudRec& c) {
  if (!c.pm) return O3DS_OK;
  ArenaScope arena_scope(h);
  return c.precision == O3DS_PRECISION_F64 ? pm_exit_t<P4d>(h, c) : pm_exit_t<P4f>(h, c);
}
}  // namespace

extern "C" {

int o3ds_voxelize_within_volume(o3ds_handle h, o3ds_cloud map, double voxel_size, const o3ds_crop* crop) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  return voxelize_within_volume_impl(h, map, voxel_size, crop);
}

int o3ds_cloud_undistort(o3ds_handle h, o3ds_cloud cloud, const double linear_velocity[3], const double angular_velocity_rpy[3],
                         double scan_duration, int spinning_clockwise) {
  CHECK_HANDLE(h);
  CloudRec* c = find_cloud(h, cloud);
  if (!c || !linear_velocity || !angular_velocity_rpy) return fail(h, O3DS_ERR_INVALID_ARG, "cloud_undistort: bad argument");
  if (!(scan_duration > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "lidar scanDuration_: must be > 0");  // MotionCompensation.cpp:62
  if (c->n == 0) return O3DS_OK;
  const double* v = linear_velocity;
  const double* w = angular_velocity_rpy;
  if (c->precision == O3DS_PRECISION_F64)
    undistort_kernel<P4d><<<grid_for(c->n), kBlock, 0, h->stream>>>((P4d*)c->pts, c->n, v[0], v[1], v[2], w[0], w[1], w[2], scan_duration,
                                                                  spinning_clockwise);
  else
    undistort_kernel<P4f><<<grid_for(c->n), kBlock, 0, h->stream>>>((P4f*)c->pts, c->n, v[0], v[1], v[2], w[0], w[1], w[2], scan_duration,
                                                                  spinning_clockwise);
  HIP_TRY(hipGetLastError());
  c->vox_first = -1;  // the points moved
  if (c->nrm) {
    dev_free(h, c->nrm);
    c->nrm = nullptr;
  }
  free_index(h, *c);
  c->has_box = false;  // the points moved
  drop_ingest_box(h, *c);  // ... and so did the box the ingest reduced
  return O3DS_OK;
}

int o3ds_dense_map_create(o3ds_handle h, double voxel_size, o3ds_dense_map* out) {
  CHECK_HANDLE(h);
  if (!out) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_create: null out");
  if (!(voxel_size > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_create: voxel_size must be > 0");
  DenseRec d;
  d.voxel = voxel_size;
  const uint64_t id = h->next_id++;
  h->dense_maps.emplace(id, d);
  *out = id;
  return O3DS_OK;
}

int o3ds_dense_map_free(o3ds_handle h, o3ds_dense_map id) {
  CHECK_HANDLE(h);
  auto it = h->dense_maps.find(id);
  if (it == h->dense_maps.end()) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_free: unknown id");
  (void)hipStreamSynchronize(h->stream);
  dense_release(h, it->second);
  h->dense_maps.erase(it);
  return O3DS_OK;
}

int o3ds_dense_map_insert(o3ds_handle h, o3ds_dense_map id, o3ds_cloud cloud, const double T[16]) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  auto it = h->dense_maps.find(id);
  CloudRec* c = find_cloud(h, cloud);
  if (it == h->dense_maps.end() || !c) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_insert: unknown id");
  return c->precision == O3DS_PRECISION_F64 ? dense_insert_t<P4d>(h, it->second, *c, T) : dense_insert_t<P4f>(h, it->second, *c, T);
}

extern "C++" {
namespace {
template <typename P4>
int export_rows_t(o3ds_handle h, const CloudRec& c, const double T[16], double voxel, int world, double* d_rows, long long* d_counts) {
  Mat34 M{};
  if (T)
    for (int r = 0; r < 3; ++r)
      for (int col = 0; col < 4; ++col) M.m[r * 4 + col] = T[col * 4 + r];
  HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(long long) * (size_t)world, h->stream));
  if (c.n == 0) return O3DS_OK;
  int* owner = nullptr;
  unsigned long long* offsets = nullptr;
  TMP_ALLOC(owner, sizeof(int) * c.n);
  TMP_ALLOC(offsets, sizeof(unsigned long long) * 2 * (size_t)world);
  owner_count_kernel<P4><<<grid_for(c.n), kBlock, 0, h->stream>>>((const P4*)c.pts, c.n, M, T ? 1 : 0, 1.0 / voxel, world, owner, d_counts);
  owner_offsets_kernel<<<1, 64, 0, h->stream>>>(d_counts, world, offsets);
  owner_scatter_kernel<P4><<<grid_for(c.n), kBlock, 0, h->stream>>>((const P4*)c.pts, (const P4*)c.nrm, c.n, M, T ? 1 : 0, world, owner, offsets, d_rows);
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}
template <typename P4>
int import_rows_t(o3ds_handle h, const double* d_rows, size_t n, int has_normals, CloudRec& out) {
  out.n = n;
  out.precision = h->precision;
  if (n == 0) return O3DS_OK;
  HIP_TRY(dev_alloc(h, (void**)&out.pts, sizeof(P4) * n));
  if (has_normals) HIP_TRY(dev_alloc(h, (void**)&out.nrm, sizeof(P4) * n));
  rows_to_cloud_kernel<P4><<<grid_for(n), kBlock, 0, h->stream>>>(d_rows, n, (P4*)out.pts, (P4*)out.nrm);
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}
}  // namespace
}  // extern "C++"

int o3ds_cloud_export_rows_by_owner(o3ds_handle h, o3ds_cloud cloud, const double T[16], double voxel_size, int world, double* d_rows,
                                    long long* d_counts) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* c = find_cloud(h, cloud);
  if (!c) return fail(h, O3DS_ERR_INVALID_ARG, "export_rows_by_owner: unknown cloud id");
  if (!(voxel_size > 0.0) || world < 1 || world > 64 || !d_counts || (c->n && !d_rows))
    return fail(h, O3DS_ERR_INVALID_ARG, "export_rows_by_owner: voxel_size > 0, 1 <= world <= 64, non-null buffers");
  return c->precision == O3DS_PRECISION_F64 ? export_rows_t<P4d>(h, *c, T, voxel_size, world, d_rows, d_counts)
                                            : export_rows_t<P4f>(h, *c, T, voxel_size, world, d_rows, d_counts);
}

int o3ds_cloud_import_rows(o3ds_handle h, const double* d_rows, size_t n, int has_normals, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  if (!out || (n && !d_rows)) return fail(h, O3DS_ERR_INVALID_ARG, "import_rows: null argument");
  CloudRec c;
  const int rc = h->precision == O3DS_PRECISION_F64 ? import_rows_t<P4d>(h, d_rows, n, has_normals, c) : import_rows_t<P4f>(h, d_rows, n, has_normals, c);
  if (rc) {
    free_cloud(h, c);
    return rc;
  }
  *out = add_cloud(h, std::move(c));
  return O3DS_OK;
}

int o3ds_dense_map_size(o3ds_handle h, o3ds_dense_map id, size_t* n_voxels) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  auto it = h->dense_maps.find(id);
  if (it == h->dense_maps.end() || !n_voxels) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_size: bad argument");
  return dense_count(h, it->second, n_voxels);
}

int o3ds_dense_map_to_cloud(o3ds_handle h, o3ds_dense_map id, o3ds_cloud* out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  auto it = h->dense_maps.find(id);
  if (it == h->dense_maps.end() || !out) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_to_cloud: bad argument");
  CloudRec o;
  const int rc = h->precision == O3DS_PRECISION_F64 ? dense_to_cloud_t<P4d>(h, it->second, o) : dense_to_cloud_t<P4f>(h, it->second, o);
  if (rc) {
    free_cloud(h, o);
    return rc;
  }
  *out = add_cloud(h, std::move(o));
  return O3DS_OK;
}

int o3ds_dense_map_transform(o3ds_handle h, o3ds_dense_map id, const double T[16]) {
  CHECK_HANDLE(h);
  auto it = h->dense_maps.find(id);
  if (it == h->dense_maps.end() || !T) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_transform: bad argument");
  DenseRec& d = it->second;
  if (d.cap == 0) return O3DS_OK;
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) M.m[r * 4 + col] = T[col * 4 + r];
  dense_transform_kernel<<<grid_for(d.cap), kBlock, 0, h->stream>>>(d.dev, d.cap, M);
  HIP_TRY(hipGetLastError());
  return O3DS_OK;
}

int o3ds_dense_map_carve(o3ds_handle h, o3ds_dense_map id, o3ds_cloud scan, const double scan_pose[16], const double sensor_position[3],
                         double neighborhood_radius, double max_raytracing_length, double truncation_distance, size_t* n_removed) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  auto it = h->dense_maps.find(id);
  CloudRec* c = find_cloud(h, scan);
  if (it == h->dense_maps.end() || !c || !sensor_position) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_carve: bad argument");
  if (!(neighborhood_radius > 0.0))
    return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_carve: neighborhood radius must be > 0 (the ray step is 2 * radius)");
  if (n_removed) *n_removed = 0;
  DenseRec& d = it->second;
  const size_t n = c->n;
  if (d.cap == 0 || n == 0) return O3DS_OK;  // cloud->empty() (Submap.cpp:128)
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) M.m[r * 4 + col] = scan_pose ? scan_pose[col * 4 + r] : (r == col ? 1.0 : 0.0);
  // removeDuplicatePointsWithinSameVoxels on the PLACED scan: keys of the placed points, stable sort, heads of the key segments
  CloudRec placed;
  int rc = c->precision == O3DS_PRECISION_F64 ? transform_t<P4d>(h, *c, scan_pose ? scan_pose : kIdentity16, placed)
                                              : transform_t<P4f>(h, *c, scan_pose ? scan_pose : kIdentity16, placed);
  if (rc) {
    free_cloud(h, placed);
    return rc;
  }
  unsigned long long *k0 = nullptr, *k1 = nullptr, *d_removed = nullptr;
  uint32_t *v0 = nullptr, *v1 = nullptr;
  int *first = nullptr, *mark = nullptr;
  TMP_ALLOC(k0, sizeof(unsigned long long) * n);
  TMP_ALLOC(k1, sizeof(unsigned long long) * n);
  TMP_ALLOC(v0, sizeof(uint32_t) * n);
  TMP_ALLOC(v1, sizeof(uint32_t) * n);
  TMP_ALLOC(first, sizeof(int) * n);
  TMP_ALLOC(mark, sizeof(int) * d.cap);
  TMP_ALLOC(d_removed, sizeof(unsigned long long));
  CropDev none{};
  if (c->precision == O3DS_PRECISION_F64)
    voxel_key_kernel<P4d><<<grid_for(n), kBlock, 0, h->stream>>>((const P4d*)placed.pts, n, 1, 0.0, 0.0, 0.0, d.voxel, none, k0, v0);
  else
    voxel_key_kernel<P4f><<<grid_for(n), kBlock, 0, h->stream>>>((const P4f*)placed.pts, n, 1, 0.0, 0.0, 0.0, d.voxel, none, k0, v0);
  size_t temp_bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
  void* temp = nullptr;
  TMP_ALLOC(temp, temp_bytes ? temp_bytes : 16);
  HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, k0, k1, v0, v1, n, 0, 64, h->stream));
  first_of_voxel_kernel<<<grid_for(n), kBlock, 0, h->stream>>>(k1, v1, n, first);
  HIP_TRY(hipMemsetAsync(mark, 0, sizeof(int) * d.cap, h->stream));
  HIP_TRY(hipMemsetAsync(d_removed, 0, sizeof(unsigned long long), h->stream));
  // rays are cast from the ALREADY PLACED points (identity here), so that the de-duplication and the rays see the same coordinates
  Mat34 I;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) I.m[r * 4 + col] = r == col ? 1.0 : 0.0;
  const double* sp = sensor_position;
  if (c->precision == O3DS_PRECISION_F64)
    dense_carve_kernel<P4d><<<grid_for(n), kBlock, 0, h->stream>>>((const P4d*)placed.pts, first, n, I, sp[0], sp[1], sp[2], d.voxel,
                                                                  neighborhood_radius, max_raytracing_length, truncation_distance, d.dev, mark);
  else
    dense_carve_kernel<P4f><<<grid_for(n), kBlock, 0, h->stream>>>((const P4f*)placed.pts, first, n, I, sp[0], sp[1], sp[2], d.voxel,
                                                                  neighborhood_radius, max_raytracing_length, truncation_distance, d.dev, mark);
  dense_erase_marked_kernel<<<grid_for(d.cap), kBlock, 0, h->stream>>>(d.dev, d.cap, mark, d_removed);
  HIP_TRY(hipGetLastError());
  unsigned long long removed = 0;
  {
    const int rb = read_back(h, {{&removed, d_removed, sizeof(removed)}});
    if (rb) {
      free_cloud(h, placed);
      return rb;
    }
  }
  free_cloud(h, placed);
  if (n_removed) *n_removed = (size_t)removed;
  return O3DS_OK;
}

int o3ds_dense_map_count_occupied(o3ds_handle h, o3ds_dense_map id, o3ds_cloud cloud, const double T[16], size_t* n_hits) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  auto it = h->dense_maps.find(id);
  CloudRec* c = find_cloud(h, cloud);
  if (it == h->dense_maps.end() || !c || !n_hits) return fail(h, O3DS_ERR_INVALID_ARG, "dense_map_count_occupied: bad argument");
  *n_hits = 0;
  DenseRec& d = it->second;
  if (d.cap == 0 || c->n == 0) return O3DS_OK;
  Mat34 M;
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) M.m[r * 4 + col] = T ? T[col * 4 + r] : (r == col ? 1.0 : 0.0);
  unsigned long long* d_hits = nullptr;
  TMP_ALLOC(d_hits, sizeof(unsigned long long));
  HIP_TRY(hipMemsetAsync(d_hits, 0, sizeof(unsigned long long), h->stream));
  if (c->precision == O3DS_PRECISION_F64)
    dense_probe_kernel<P4d><<<grid_for(c->n), kBlock, 0, h->stream>>>((const P4d*)c->pts, c->n, M, 1.0 / d.voxel, d.dev, d_hits);
  else
    dense_probe_kernel<P4f><<<grid_for(c->n), kBlock, 0, h->stream>>>((const P4f*)c->pts, c->n, M, 1.0 / d.voxel, d.dev, d_hits);
  HIP_TRY(hipGetLastError());
  unsigned long long hits = 0;
  {
    const int rb = read_back(h, {{&hits, d_hits, sizeof(hits)}});
    if (rb) return rb;
  }
  *n_hits = (size_t)hits;
  return O3DS_OK;
}

int o3ds_overlap_indices(o3ds_handle h, o3ds_cloud source, o3ds_cloud target, const double source_to_target[16], double voxel_size,
                         size_t min_points_per_voxel, uint64_t* idx_source, size_t* n_idx_source, uint64_t* idx_target, size_t* n_idx_target) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* s = find_cloud(h, source);
  CloudRec* t = find_cloud(h, target);
  if (!s || !t || !source_to_target || !n_idx_source || !n_idx_target) return fail(h, O3DS_ERR_INVALID_ARG, "overlap_indices: bad argument");
  if ((s->n && !idx_source) || (t->n && !idx_target)) return fail(h, O3DS_ERR_INVALID_ARG, "overlap_indices: null index buffer");
  if (!(voxel_size > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "overlap_indices: voxel_size must be > 0");
  if (min_points_per_voxel < 1) return fail(h, O3DS_ERR_INVALID_ARG, "minNumPointsPerVoxel must be >= 1");  // assert_ge, helpers.cpp:310
  if (s->n && t->n && s->precision != t->precision) return fail(h, O3DS_ERR_INVALID_ARG, "overlap_indices: precision mismatch");
  static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "index type");
  return s->precision == O3DS_PRECISION_F64
             ? overlap_t<P4d>(h, *s, *t, source_to_target, voxel_size, min_points_per_voxel, (unsigned long long*)idx_source, n_idx_source,
                              (unsigned long long*)idx_target, n_idx_target)
             : overlap_t<P4f>(h, *s, *t, source_to_target, voxel_size, min_points_per_voxel, (unsigned long long*)idx_source, n_idx_source,
                              (unsigned long long*)idx_target, n_idx_target);
}

int o3ds_map_carve(o3ds_handle h, o3ds_cloud map, o3ds_cloud raw_scan, const double map_to_range_sensor[16], const o3ds_crop* map_builder_crop,
                   const o3ds_carving_params* params, size_t* n_removed) {
  return o3ds_map_carve_removed(h, map, raw_scan, map_to_range_sensor, map_builder_crop, params, n_removed, nullptr);
}

int o3ds_map_carve_removed(o3ds_handle h, o3ds_cloud map, o3ds_cloud raw_scan, const double map_to_range_sensor[16], const o3ds_crop* map_builder_crop,
                           const o3ds_carving_params* params, size_t* n_removed, o3ds_cloud* removed_out) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  {  // a submap in its persistent form is carved where it is when the carving voxel is the map's (the voxel hash is the reference's table)
    CloudRec* pmc = find_cloud_lazy(h, map);
    CloudRec* sc = find_cloud(h, raw_scan);
    if (pmc && sc && pmc != sc && pmc->pm && params && map_to_range_sensor && !removed_out && params->voxel_size == pmc->pm->dev.voxel &&
        (pmc->n == 0 || sc->n == 0 || pmc->precision == sc->precision)) {
      if (n_removed) *n_removed = 0;
      if (sc->n == 0) return O3DS_OK;
      PMapRec* pm = pmc->pm;
      const PmDev d = pm->dev;
      const double* T = map_to_range_sensor;
      Mat34 M;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) M.m[r * 4 + c] = T[c * 4 + r];
      const CropDev cd = to_dev(map_builder_crop);
      unsigned int* block_bits = nullptr;
      TMP_ALLOC(block_bits, sizeof(unsigned int) << (kCarveBitsLog2 - 5));
      HIP_TRY(hipMemsetAsync(block_bits, 0, sizeof(unsigned int) << (kCarveBitsLog2 - 5), h->stream));
      pm_carve_bits_kernel<<<grid_for((size_t)d.hmask + 1), kBlock, 0, h->stream>>>(d.h, (size_t)d.hmask + 1, block_bits);
      pm->rec_seq = ++h->rec_seq;
      o3ds_context::PinRec* rec = h->h_rec_dev + pm->rec_slot;
      const CountPub pub{cnt_word(h, pm->rec_slot), &rec->cnt, pm->rec_seq};
      if (pmc->precision == O3DS_PRECISION_F64) {
        pm_carve_rays_kernel<P4d><<<(unsigned int)((sc->n + kBlock - 1) / kBlock), kBlock, 0, h->stream>>>(d, (const P4d*)sc->pts, sc->n, M, T[12], T[13], T[14], params->max_raytracing_length,
                                                                            params->truncation_distance, params->min_dot_product_with_normal, cd, block_bits);
        pm_carve_apply_kernel<P4d><<<256, kBlock, 0, h->stream>>>(d, pub, rec->box);
      } else {
        pm_carve_rays_kernel<P4f><<<(unsigned int)((sc->n + kBlock - 1) / kBlock), kBlock, 0, h->stream>>>(d, (const P4f*)sc->pts, sc->n, M, T[12], T[13], T[14], params->max_raytracing_length,
                                                                            params->truncation_distance, params->min_dot_product_with_normal, cd, block_bits);
        pm_carve_apply_kernel<P4f><<<256, kBlock, 0, h->stream>>>(d, pub, rec->box);
      }
      pm_carve_finish_kernel<<<1, 64, 0, h->stream>>>(d, pub, rec->box);
      HIP_TRY(hipGetLastError());
      if (n_removed) {  // (asking for the count is what waits)
        HIP_TRY(hipStreamSynchronize(h->stream));
        *n_removed = (size_t)(h->h_rec + pm->rec_slot)->pad;  // (the seventh double of the record)
        const int rp = pm_poll(h, pm, true);
        if (rp) return rp;
      }
      return O3DS_OK;
    }
  }
  CloudRec* m = find_cloud(h, map);
  CloudRec* s = find_cloud(h, raw_scan);
  if (!m || !s || !map_to_range_sensor || !params || m == s) return fail(h, O3DS_ERR_INVALID_ARG, "map_carve: bad argument");
  if (!(params->voxel_size > 0.0)) return fail(h, O3DS_ERR_INVALID_ARG, "map_carve: voxel_size must be > 0");
  if (m->n > 0 && s->n > 0 && m->precision != s->precision) return fail(h, O3DS_ERR_INVALID_ARG, "map_carve: precision mismatch");
  size_t removed = 0;
  const CropDev cd = to_dev(map_builder_crop);
  CloudRec gone;
  CloudGuard gone_guard(h, gone);
  gone.precision = m->precision;
  const int rc = m->precision == O3DS_PRECISION_F64 ? carve_t<P4d>(h, *m, *s, map_to_range_sensor, cd, *params, &removed, removed_out ? &gone : nullptr)
                                                    : carve_t<P4f>(h, *m, *s, map_to_range_sensor, cd, *params, &removed, removed_out ? &gone : nullptr);
  if (n_removed) *n_removed = removed;
  if (rc) return rc;
  if (removed_out) {
    gone_guard.release();
    *removed_out = add_cloud(h, std::move(gone));  // an empty cloud when nothing was carved
  }
  return O3DS_OK;
}

int o3ds_map_insert_scan(o3ds_handle h, o3ds_cloud map, o3ds_cloud scan, const double T[16], double map_voxel_size,
                         const o3ds_crop* map_builder_crop, double max_corr_hint) {
  CHECK_HANDLE(h);
  ArenaScope arena_scope(h);
  CloudRec* m = find_cloud_lazy(h, map);  // (a map in its persistent form stays in it)
  CloudRec* s = find_cloud_lazy(h, scan);
  if (!m || !s || !T || m == s) return fail(h, O3DS_ERR_INVALID_ARG, "map_insert_scan: bad argument");
  if (s->pm) return fail(h, O3DS_ERR_INVALID_ARG, "map_insert_scan: the scan is a map");
  if (s->n == 0) return O3DS_OK;  // Submap.cpp:41-43: empty pre-processed scan is a no-op
  if (m->n == 0) m->precision = s->precision;
  if (m->precision != s->precision) return fail(h, O3DS_ERR_INVALID_ARG, "map_insert_scan: precision mismatch");
  int rc = O3DS_OK;
  // ---- the persistent form (map_kernels.hpp): work proportional to the scan.  It takes a map with a known layout [pass-through block | voxel
  // block in key order] -- what the first insertion into a map (below) or a fold leaves --, a search index being wanted, no colours, and map
  // and scan agreeing on normals ([O3D] operator+= drops the map's normals otherwise).
  static const bool no_pm = ab_getenv("O3DS_NO_PERSISTENT_MAP") != nullptr;  // A/B and the bitwise tests: the array form at every insertion
  const bool pm_ok = !no_pm && map_voxel_size > 0.0 && max_corr_hint > 0.0 && !m->col && !s->col && m->n > 0 && (m->nrm != nullptr) == (s->nrm != nullptr) &&
                     m->n + s->n < ((size_t)1 << 30);
  const CropDev cd = to_dev(map_builder_crop);
  CloudRec placed;
  placed.n = s->n;
  box_transform(placed, *s, T);
  if (m->pm) {
    PMapRec* pm = m->pm;
    rc = pm_poll(h, pm, false);
    if (rc) return rc;
    bool refold = !pm_ok || pm->dev.voxel != map_voxel_size || pm->dev.kc != pm_cell_voxels(max_corr_hint, map_voxel_size) || pm->t + 1 > kPmHistory;
    const double growth = 2.0 * (double)(pm->n_upper + s->n) + 8.0 * (double)s->n;  // every touched row moving to the pool's end
    if (!refold && (pm->n_upper + s->n > pm->dev.cap || pm->pool_top + growth > (double)pm->dev.pool_cap)) {
      rc = pm_poll(h, pm, true);  // the bounds were pessimistic? the exact numbers
      if (rc) return rc;
      refold = pm->n_upper + s->n > pm->dev.cap || pm->pool_top + growth > (double)pm->dev.pool_cap;
    }
    if (!refold && pm->clamped > 20000.0) refold = true;  // the map has grown well beyond the index grid: new extents (the fold re-grids)
    if (!refold && pm->multi_total + 4.0 * (double)s->n > 0.5 * (double)pm->dev.list_cap) refold = true;  // the multi list only grows (a fold starts it afresh)
    if (refold) {
      rc = pm_exit(h, *m);
      if (rc) return rc;
    }
  }
  if (!m->pm && pm_ok && m->vox_first >= 0 && (size_t)m->vox_first + m->vox_count == m->n) {
    rc = DISPATCH(m->precision, pm_enter_t, h, *m, map_voxel_size, max_corr_hint, s->n);
    if (rc) {
      pm_release(h, *m);
      m->vox_first = -1;
      return rc;
    }
  }
  if (m->pm) {
    CloudRec joined;
    box_union(joined, *m, placed);
    rc = DISPATCH(m->precision, pm_insert_t, h, *m, *s, T, cd);
    if (rc) return rc;
    box_copy(*m, joined);
    m->pm->pool_top += 2.0 * (double)(m->pm->n_upper) + 8.0 * (double)s->n;
    return O3DS_OK;
  }
  // ---- the array form (the reference's own way: everything is re-binned): the first insertions of a map, colours, a call without index ...
  rc = resolve_count(h, *s, true);
  if (rc) return rc;
  if (s->n == 0) return O3DS_OK;
  // the layout the previous insertion left (pass-through block, then the voxel block in key order): the merge below re-bins by merging
  const bool known = m->vox_first >= 0 && (size_t)m->vox_first + m->vox_count == m->n;
  const long long merge_np = known ? m->vox_first : -1;
  const size_t merge_nv = known ? m->vox_count : 0;
  if (m->n > 0 && m->cap >= m->n + s->n && m->nrm && s->nrm && !m->col && !s->col) {
    // transform (Submap.cpp:54) and operator+= (Submap.cpp:70) in one launch: the placed points go behind the map's last one, where the
    // previous merge left room -- the same kernel with the same arithmetic as transform_t, indices as append_t assigns them
    CloudRec joined;
    placed.n = s->n;
    box_union(joined, *m, placed);
    Mat34 M;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) M.m[r * 4 + c] = T[c * 4 + r];
    if (m->precision == O3DS_PRECISION_F64)
      transform_kernel<P4d><<<grid_for(s->n), kBlock, 0, h->stream>>>((const P4d*)s->pts, (const P4d*)s->nrm, s->n, M, T[3], T[7], T[11], T[15], (P4d*)m->pts,
                                                                      (P4d*)m->nrm, m->n);
    else
      transform_kernel<P4f><<<grid_for(s->n), kBlock, 0, h->stream>>>((const P4f*)s->pts, (const P4f*)s->nrm, s->n, M, T[3], T[7], T[11], T[15], (P4f*)m->pts,
                                                                      (P4f*)m->nrm, m->n);
    HIP_TRY(hipGetLastError());
    free_index(h, *m);
    m->n += s->n;
    box_copy(*m, joined);
  } else {
    CloudRec t;
    rc = DISPATCH(s->precision, transform_t, h, *s, T, t);      // Submap.cpp:54
    if (!rc) rc = DISPATCH(m->precision, append_t, h, *m, t);   // Submap.cpp:70
    free_cloud(h, t);
  }
  if (rc) return rc;
  rc = voxelize_within_volume_impl(h, map, map_voxel_size, map_builder_crop, merge_np, merge_nv);  // Submap.cpp:71-72
  if (rc) return rc;
  m = find_cloud(h, map);
  if (max_corr_hint > 0.0) {
    rc = build_index(h, *m, max_corr_hint / index_cell_div());
  }
  return rc;
}

}  // extern "C"

#include "sharded.hpp"
