// map_kernels.hpp -- Submap::insertScan (Submap.cpp:39-75) in time independent of the map's size: the PERSISTENT form of a submap.
//
// The reference re-bins the whole map at every inserted scan: mapCloud_ += T * scan, then voxelizeWithinCroppingVolume (helpers.cpp:115-183)
// walks all N + m points -- inside the cropping volume each voxel's points are replaced by their mean (normal averaged and re-normalised),
// outside they pass through -- and rounds 1-4 of this backend did the same with sorted key lists (backend.hip voxel_reduce_t) plus a
// rebuild of the search index: ~230 us per scan at 1 M points, linear in N.  But an insertion CHANGES only what the scan touches.  A voxel
// that holds one point and receives none keeps it: mean of one point p is p / 1 = p exactly (only its normal is re-normalised, see below).
// So the map is kept as
//   * slot arrays      pts / nrm [cap]: a point lives in one slot from its creation to its death; new points are appended;
//   * a voxel hash     key floor(p * (1 / v)) (VoxelHashMap.hpp:47-50) -> chain of the slots whose point has that key.  Nearly every
//                      chain has one member; two points share a voxel only if they were never inside a volume together (points outside
//                      pass through unmerged) or a mean was rounded across a voxel face;
//   * the search index a uniform grid like build_grid_t's, but aligned with the voxel grid (cell = k voxels: a mean never leaves its voxel,
//                      hence never its cell, so it is updated in place) and PAGED BY ROW: every (y, z) row of cells owns a region of the
//                      cell-sorted arrays with room to spare, `cell_start` holds nx + 1 absolute positions per row, and new points are
//                      merged into their rows by a kernel that rewrites only those rows (a row that outgrows its region moves to the
//                      end of the pool).  The registration kernels search it unchanged: a row's cells are still one contiguous range;
//   * per-slot history the insertion that last wrote the slot and the voxel key it was written under, and the list of the volumes since
//                      the map entered this form: enough to reproduce the reference's output ORDER (pass-through points first, "in
//                      original order", then the voxel means, here in key order) when somebody looks (pm_view_key_kernel).
// and an insertion is five launches over the m scan points: group the placed scan by voxel (the VoxelDownSample machinery), one thread
// per touched voxel sums [old members inside the volume, in map order] + [scan points, in scan order] exactly as the reference's loop
// does, writes the mean into the old member's slot (or a new one), then the few special cases (voxels with several old members, means
// that crossed a face, scan points outside the volume, normals that are not yet a fixed point of the re-normalisation), then the rows.
// Nothing returns to the host: the number of slots lives in a device word, the host keeps an upper bound.
//
// Exactness notes.  (1) normalized(n) is not idempotent in floating point: re-normalising a unit vector may move its last bit, and the
// reference re-normalises EVERY point inside the volume at EVERY insertion.  A slot whose normal is not yet a fixed point is kept on a list
// and re-normalised at each insertion that has it inside the volume until it is; all others need no work.  (2) A voxel with several old
// members sums them in map order, which for the persistent form means the order of pm_view_key (the same function the view uses).
#pragma once
#include "cloud_kernels.hpp"
#include "common.hpp"

namespace o3ds {

#pragma clang fp contract(off)

constexpr int kPmHistory = 256;                 // insertions between two folds at most (the list of their volumes)
constexpr unsigned long long kPmRaw = 1ull << 63;  // okey flag: a scan point that was inserted OUTSIDE the volume (low bits: its scan index)
constexpr unsigned int kPmDead = 1, kPmUnsettled = 2, kPmFresh = 4, kPmCarved = 8;  // (fresh: created by the running insertion, not in the index yet)
constexpr int kPmMaxOld = 8;  // old members of one voxel the special-case path sorts in LDS (more: pm_merge_many)

struct alignas(32) PmSlot {
  unsigned long long okey;  // voxel key the slot was written under (kPmRaw | scan index for a point inserted outside the volume)
  int stamp;                // insertion (1, 2, ...; 0 = the base the map entered with) that last wrote the slot
  int pos;                  // position in the paged index
  int hnext;                // next slot of the same voxel chain, -1 = end
  unsigned int flags;       // kPmDead | kPmUnsettled | kPmFresh
  int out_tau;              // memo of pm_view_key's walk back through the volumes: the slot left the voxel block with insertion out_tau
  int out_checked;          // (0: it never was in one since its stamp) and has been outside ever since up to insertion out_checked;
                            // valid while out_checked >= stamp (a write of the slot makes it stale by itself); -1 = no memo
};
struct alignas(16) PmHash {
  unsigned long long key;
  int head;           // first slot of the chain, -1 = none
  unsigned int info;  // bit 0: on the multi list; bits 8..31: the insertion whose pm_group_kernel found several old members of the voxel
                      // inside the volume (pm_merge_kernel: "dealt with")
};
// everything a kernel needs of a persistent map (device pointers; a copy travels by value in the kernel arguments)
struct PmDev {
  void* pts;  // P4[cap]
  void* nrm;  // P4[cap] or null
  struct PmSlot* slot;       // [cap] per-slot record (one cache line per slot: as separate arrays an insertion touched five lines per voxel)
  size_t cap;
  // voxel hash (open addressing, never deleted from: a chain may be empty)
  struct PmHash* h;
  unsigned int hmask;
  // lists and counters (device)
  int* counters;        // see PmCounter
  int* unsettled[2];    // double-buffered list of slots whose normal is not a fixed point of the re-normalisation
  unsigned long long* multi[2];  // [0]: list of the voxel keys whose chain holds more than one live slot -- appended to (kPmMultiOut entries so far), never
                                 // compacted: an entry that is settled becomes kEmptyKey where it stands; an insertion looks at the first kPmMultiIn
                                 // ([1] is unused)
  int* complex_groups;  // groups of this insertion with more than one old member inside the volume (by number in `order`)
  int* outside_pts;     // scan points of this insertion that lie outside the volume
  int* relink;          // slots whose mean left its voxel (rounding): re-hashed and re-indexed one by one
  int list_cap;
  // base layout and history
  int n_base, np_base;  // slots [0, np_base): pass-through block of the base, [np_base, n_base): its voxel block in key order
  const CropDev* hist;  // [kPmHistory + 1] volume of insertion t (entry 0 unused)
  double voxel, inv_voxel;
  // paged index
  GridDev grid;         // cell_start = row-paged table: (nx + 1) entries per row, absolute positions
  int* cs;              // the same table, writable
  void* spts;           // P4[pool]
  void* snrm;           // P4[pool] or null
  int* row_cap;         // [rows] capacity of the row's region
  int* row_flag;        // [rows] 1: on the list of touched rows (cleared by pm_rows_kernel)
  int* cell_add;        // [like cs] slots that enter the cell this insertion: counted up by pm_row_push, read by pm_rows_kernel, counted back
                        // down to zero by pm_place_new_kernel (the cursors of the placement)
  int* touched_rows;    // rows with new slots this insertion
  int* new_slots;       // the slots that enter the index this insertion
  int pool_cap;         // positions the cell-sorted arrays have
  int kc;               // cell edge in voxels
  long long gx0, gy0, gz0;  // voxel coordinates of the grid's min corner
};
enum PmCounter {
  kPmN = 0,        // slots in use (dead ones included)
  kPmDeadCnt,      // ... of which dead
  kPmUnsettledIn,  // entries of unsettled[cur]
  kPmUnsettledOut,
  kPmMultiIn,
  kPmMultiOut,
  kPmComplex,
  kPmOutside,
  kPmRelink,
  kPmTouched,
  kPmNew,          // entries of new_slots
  kPmClamped,      // slots that entered the index outside its grid since the map took this form (the host re-grids when they are many)
  kPmCarvedCnt,    // slots the running carve removes (listed in new_slots: no insertion is in flight then)
  kPmPoolTop,      // first free position of the index pool
  kPmError,        // sticky: 1 a list overflowed, 2 the slot arrays, 4 the index pool (the host sizes all three so that none can happen)
  kPmCounters = 16
};

template <typename P4>
__device__ __forceinline__ unsigned long long pm_key(const P4& p, double inv) {
  return pack_key((long long)floor((double)p.x * inv), (long long)floor((double)p.y * inv), (long long)floor((double)p.z * inv));
}
__device__ __forceinline__ unsigned int pm_hash(unsigned long long k) { return (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> 32); }

// hash entry of key k (inserted if absent)
__device__ __forceinline__ unsigned int pm_entry(const PmDev& m, unsigned long long k) {
  unsigned int e = pm_hash(k) & m.hmask;
  while (true) {
    const unsigned long long prev = atomicCAS(&m.h[e].key, kEmptyKey, k);
    if (prev == kEmptyKey || prev == k) return e;
    e = (e + 1) & m.hmask;
  }
}
// ... or ~0u when the key is not in the table (read-only probe)
__device__ __forceinline__ unsigned int pm_find(const PmDev& m, unsigned long long k) {
  unsigned int e = pm_hash(k) & m.hmask;
  while (true) {
    const unsigned long long cur = m.h[e].key;
    if (cur == k) return e;
    if (cur == kEmptyKey) return ~0u;
    e = (e + 1) & m.hmask;
  }
}

// index cell of a voxel key: (row, x) with the clamping build_grid_t's cell_of has
__device__ __forceinline__ void pm_cell(const PmDev& m, unsigned long long key, int* row, int* x) {
  const long long vx = (long long)(key & 0x1FFFFFull) - (1ll << 20), vy = (long long)((key >> 21) & 0x1FFFFFull) - (1ll << 20),
                  vz = (long long)((key >> 42) & 0x1FFFFFull) - (1ll << 20);
  auto cell = [](long long v, long long g0, int kc, int n) {
    long long d = v - g0;
    long long c = d >= 0 ? d / kc : -((-d + kc - 1) / kc);
    return (int)(c < 0 ? 0 : (c >= n ? n - 1 : c));
  };
  const int ix = cell(vx, m.gx0, m.kc, m.grid.nx), iy = cell(vy, m.gy0, m.kc, m.grid.ny), iz = cell(vz, m.gz0, m.kc, m.grid.nz);
  *row = iz * m.grid.ny + iy;
  *x = ix;
}
// whether the voxel lies outside the grid (its points are kept in the border cells: exact, but the search wades through them)
__device__ __forceinline__ bool pm_outside_grid(const PmDev& m, unsigned long long key) {
  const long long vx = (long long)(key & 0x1FFFFFull) - (1ll << 20), vy = (long long)((key >> 21) & 0x1FFFFFull) - (1ll << 20),
                  vz = (long long)((key >> 42) & 0x1FFFFFull) - (1ll << 20);
  return vx < m.gx0 || vx >= m.gx0 + (long long)m.grid.nx * m.kc || vy < m.gy0 || vy >= m.gy0 + (long long)m.grid.ny * m.kc || vz < m.gz0 ||
         vz >= m.gz0 + (long long)m.grid.nz * m.kc;
}

// whether every point a voxel can hold lies outside the cropping volume (false when in doubt)
__device__ __forceinline__ bool pm_voxel_outside(const PmDev& m, unsigned long long key, const CropDev& c) {
  if (c.invert || c.kind == O3DS_CROP_NONE) return false;
  const double vx = (double)((long long)(key & 0x1FFFFFull) - (1ll << 20)), vy = (double)((long long)((key >> 21) & 0x1FFFFFull) - (1ll << 20)),
               vz = (double)((long long)((key >> 42) & 0x1FFFFFull) - (1ll << 20));
  const double v = m.voxel, half = 0.5 * v;
  const double dx = (vx + 0.5) * v - c.cx, dy = (vy + 0.5) * v - c.cy, dz = (vz + 0.5) * v - c.cz;
  const double slack = 0.8661 * v + 1e-6 * (fabs(dx) + fabs(dy) + fabs(dz) + 1.0);  // half the voxel's diagonal, rounded up, and a margin
  if (c.kind == O3DS_CROP_CYLINDER) {
    const double lo = (vz + 0.0) * v, hi = (vz + 1.0) * v;
    if (lo > c.zmax + 1e-9 * fabs(c.zmax) + 1e-12 || hi < c.zmin - 1e-9 * fabs(c.zmin) - 1e-12) return true;
    const double d = sqrt(dx * dx + dy * dy);
    return c.le2 >= 0.0 ? d - slack > sqrt(c.le2) : true;
  }
  (void)half;
  const double d = sqrt(dx * dx + dy * dy + dz * dz);
  const bool beyond = (c.kind == O3DS_CROP_MAX_RADIUS || c.kind == O3DS_CROP_MIN_MAX_RADIUS) && (c.le2 < 0.0 || d - slack > sqrt(c.le2));
  const bool within = (c.kind == O3DS_CROP_MIN_RADIUS || c.kind == O3DS_CROP_MIN_MAX_RADIUS) && c.ge2 > 0.0 && d + slack < sqrt(c.ge2);
  return beyond || within;
}

// Eigen normalized() as segment_mean_kernel spells it; returns whether anything was divided
__device__ __forceinline__ void pm_normalize(double& nx, double& ny, double& nz) {
  const double z2 = (nx * nx + ny * ny) + nz * nz;
  if (z2 > 0.0) {
    const double nn = sqrt(z2);
    nx /= nn;
    ny /= nn;
    nz /= nn;
  }
}
// what the NEXT insertion makes of a slot that stays alone in its voxel: mean of one normal (NaN skipped: the sum stays 0), re-normalised,
// rounded to storage.  A slot is settled when that is the normal it has.
template <typename P4>
__device__ __forceinline__ P4 pm_renormalized(const P4& n) {
  using R = typename Scalar<P4>::type;
  double a = (double)n.x, b = (double)n.y, c = (double)n.z;
  if (isnan(a) || isnan(b) || isnan(c)) a = b = c = 0.0;
  a /= 1.0;  // (the mean of one: spelled out, exact)
  b /= 1.0;
  c /= 1.0;
  pm_normalize(a, b, c);
  P4 o;
  o.x = (R)a;
  o.y = (R)b;
  o.z = (R)c;
  o.i = 0;
  return o;
}
template <typename P4>
__device__ __forceinline__ bool pm_same_bits(const P4& a, const P4& b) {
  using R = typename Scalar<P4>::type;
  if (sizeof(R) == 4) return __float_as_uint((float)a.x) == __float_as_uint((float)b.x) && __float_as_uint((float)a.y) == __float_as_uint((float)b.y) && __float_as_uint((float)a.z) == __float_as_uint((float)b.z);
  return __double_as_longlong((double)a.x) == __double_as_longlong((double)b.x) && __double_as_longlong((double)a.y) == __double_as_longlong((double)b.y) &&
         __double_as_longlong((double)a.z) == __double_as_longlong((double)b.z);
}

// One ticket per ACTIVE lane from a shared counter with ONE atomic per wavefront: a device-scope atomic that returns its value costs ~25 ns
// when many wavefronts aim at the same address, and an insertion draws ~15 k tickets from three counters (new slots,
// slots entering the index, unsettled normals).  The lanes that reach the call together -- whatever branch they are in: __ballot(1) is the
// execution mask -- are served by their lowest lane.
__device__ __forceinline__ int pm_ticket(int* counter) {
  const unsigned long long act = __ballot(1);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int leader = (int)__builtin_ctzll(act);
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, (int)__popcll(act));
  base = __builtin_amdgcn_readfirstlane(base);  // the first active lane is the leader
  return base + (int)__popcll(act & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ void pm_push(int* list, int* counter, int cap, int v, int* err) {
  const int k = pm_ticket(counter);
  if (k < cap)
    list[k] = v;
  else
    atomicOr(err, 1);
}
__device__ __forceinline__ void pm_push64(unsigned long long* list, int* counter, int cap, unsigned long long v, int* err) {
  const int k = pm_ticket(counter);
  if (k < cap)
    list[k] = v;
  else
    atomicOr(err, 1);
}

// "the slot was inside the volume of insertion t, or written by it" -- i.e. it belonged to the voxel block the reference's array had after
// insertion t.  Only asked for t >= the slot's stamp (before that the slot held another point, or none).
template <typename P4>
__device__ __forceinline__ bool pm_in_block(const PmDev& m, int s, const P4& p, int st, unsigned long long ok, int t) {
  if (t == 0) return s >= m.np_base && s < m.n_base;
  if (st == t) return !(ok & kPmRaw);
  const CropDev c = m.hist[t];
  return crop_contains(c, (double)p.x, (double)p.y, (double)p.z);
}
// Position of a live slot in the reference's array after insertion t_ref, as a 128-bit sort key (hi, lo): pass-through block first ("in
// original order": by the insertion after which the point was last outside for good, base pass-through points before everything; points
// that left the voxel block of insertion t' follow in that block's order; scan points inserted outside come after them, in scan order),
// then the voxel block in key order.
template <typename P4>
__device__ __forceinline__ void pm_view_key(const PmDev& m, int s, int t_ref, unsigned long long* hi, unsigned long long* lo) {
  const P4 p = ((const P4*)m.pts)[s];
  const int st = m.slot[s].stamp;
  const unsigned long long ok = m.slot[s].okey;
  const unsigned long long kp = pm_key(p, m.inv_voxel);
  if (pm_in_block(m, s, p, st, ok, t_ref)) {
    *hi = 1ull << 63;
    *lo = t_ref == 0 ? (unsigned long long)s : (st == t_ref ? ok : kp);
    return;
  }
  // the walk back through the volumes, as far as the slot's memo does not already answer it (a listed voxel far from the volume is looked at
  // at every insertion: without the memo each look walks the whole history again, hundreds of steps late in a long run)
  const PmSlot ms = m.slot[s];
  const bool memo = ms.out_checked >= st && ms.out_checked < t_ref;
  int tau = memo ? ms.out_tau : 0;
  const int t_low = max(memo ? ms.out_checked + 1 : st, 0);
  for (int t_hi = t_ref - 1; t_hi >= t_low; t_hi -= 8) {  // eight volumes per round: their loads go out together
    bool in[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) in[u] = t_hi - u >= t_low && pm_in_block(m, s, p, st, ok, t_hi - u);
    int hit = -1;
#pragma unroll
    for (int u = 7; u >= 0; --u)
      if (in[u]) hit = u;
    if (hit >= 0) {
      tau = t_hi - hit + 1;
      break;
    }
  }
  m.slot[s].out_tau = tau;
  m.slot[s].out_checked = t_ref - 1;
  if (tau > 0) {  // it belonged to the voxel block of insertion tau - 1 and left with insertion tau
    const int t = tau - 1;
    *hi = (unsigned long long)tau << 1;
    *lo = t == 0 ? (unsigned long long)s : (st == t ? ok : kp);
    return;
  }
  if ((ok & kPmRaw) && st > 0) {  // inserted outside the volume at insertion st and never inside since
    *hi = ((unsigned long long)st << 1) | 1ull;
    *lo = ok & ~kPmRaw;
    return;
  }
  *hi = 0;  // a pass-through point of the base that has not been inside since
  *lo = (unsigned long long)s;
}

// The same for the lanes of a wavefront that ask (`want`), one after the other, with all 64 lanes walking: 64 volumes per round instead of
// eight.  A point that comes back into the volume after two hundred insertions outside is a walk of two hundred steps -- 25 rounds of
// dependent loads on one lane, and pm_merge_kernel is as long as its longest lane; four rounds this way.  Whole wavefronts only.
template <typename P4>
__device__ __forceinline__ void pm_view_key_wave(const PmDev& m, bool want, int s_mine, int t_ref, unsigned long long* hi, unsigned long long* lo) {
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  unsigned long long pending = __ballot(want);
  while (pending) {
    const int leader = (int)__builtin_ctzll(pending);
    pending &= pending - 1ull;
    const int s = __shfl(s_mine, leader, 64);
    const P4 p = ((const P4*)m.pts)[s];
    const PmSlot ms = m.slot[s];
    const int st = ms.stamp;
    const unsigned long long ok = ms.okey;
    const unsigned long long kp = pm_key(p, m.inv_voxel);
    unsigned long long h, l;
    if (pm_in_block(m, s, p, st, ok, t_ref)) {
      h = 1ull << 63;
      l = t_ref == 0 ? (unsigned long long)s : (st == t_ref ? ok : kp);
    } else {
      const bool memo = ms.out_checked >= st && ms.out_checked < t_ref;
      int tau = memo ? ms.out_tau : 0;
      const int t_low = max(memo ? ms.out_checked + 1 : st, 0);
      for (int base = t_ref - 1; base >= t_low; base -= 64) {
        const int t = base - lane;
        const bool in = t >= t_low && pm_in_block(m, s, p, st, ok, t);
        const unsigned long long hit = __ballot(in);
        if (hit) {  // the lowest lane is the latest volume
          tau = base - (int)__builtin_ctzll(hit) + 1;
          break;
        }
      }
      if (lane == leader) {
        m.slot[s].out_tau = tau;
        m.slot[s].out_checked = t_ref - 1;
      }
      if (tau > 0) {
        const int t = tau - 1;
        h = (unsigned long long)tau << 1;
        l = t == 0 ? (unsigned long long)s : (st == t ? ok : kp);
      } else if ((ok & kPmRaw) && st > 0) {
        h = ((unsigned long long)st << 1) | 1ull;
        l = ok & ~kPmRaw;
      } else {
        h = 0;
        l = (unsigned long long)s;
      }
    }
    if (lane == leader) *hi = h, *lo = l;
  }
}

// ---- entering the persistent form -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pm_hash_init_kernel(PmHash* __restrict__ hh, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    PmHash e;
    e.key = kEmptyKey;
    e.head = -1;
    e.info = 0u;
    hh[i] = e;
  }
}
// per slot of the base [pass block | voxel block in key order]: history, hash chain, settled or not
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_enter_kernel(PmDev m, int n) {
  int* err = m.counters + kPmError;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < n; s += gridDim.x * kBlock) {
    const P4 p = ((const P4*)m.pts)[s];
    const unsigned long long k = pm_key(p, m.inv_voxel);
    m.slot[s].stamp = 0;
    m.slot[s].okey = k;
    m.slot[s].out_checked = -1;
    m.slot[s].out_tau = 0;
    unsigned int fl = 0;
    if (m.nrm) {
      const P4 nv = ((const P4*)m.nrm)[s];
      if (!pm_same_bits(pm_renormalized(nv), nv)) {
        fl |= kPmUnsettled;
        pm_push(m.unsettled[0], m.counters + kPmUnsettledIn, m.list_cap, s, err);
      }
    }
    m.slot[s].flags = fl;
    const unsigned int e = pm_entry(m, k);
    const int old = atomicExch(&m.h[e].head, s);
    m.slot[s].hnext = old;
    if (old != -1 && !(atomicOr(&m.h[e].info, 1u) & 1u)) pm_push64(m.multi[0], m.counters + kPmMultiOut, m.list_cap, k, err);
  }
}

// ---- the paged index: build -----------------------------------------------------------------------------------------------------------
// counts per cell of the (nx + 1)-strided table; the extra entry of every row receives the row's spare room afterwards (pm_row_slack_kernel),
// so that ONE exclusive scan of the table yields absolute cell starts with the rows' regions laid out one after the other
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_cell_count_kernel(PmDev m, int n, int* __restrict__ counts, int* __restrict__ cell_id) {
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < n; s += gridDim.x * kBlock) {
    if (m.slot[s].flags & kPmDead) {
      cell_id[s] = -1;
      continue;
    }
    const P4 p = ((const P4*)m.pts)[s];
    int row, x;
    pm_cell(m, pm_key(p, m.inv_voxel), &row, &x);
    const int c = row * (m.grid.nx + 1) + x;
    cell_id[s] = c;
    atomicAdd(&counts[c], 1);
  }
}
__global__ __launch_bounds__(kBlock) void pm_row_slack_kernel(int* __restrict__ counts, int rows, int nx, int* __restrict__ row_cap, int* __restrict__ row_flag) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); r < rows; r += gridDim.x * (kBlock / 64)) {  // one wavefront per row
    int c = 0;
    for (int x = lane; x < nx; x += 64) c += counts[(size_t)r * (nx + 1) + x];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane == 0) {
      const int slack = c > 0 ? max(8, c / 2) : 0;  // a row that has points gets room for half as many again; an empty one moves to the pool's end with its first point
      counts[(size_t)r * (nx + 1) + nx] = slack;
      row_cap[r] = c + slack;
      row_flag[r] = 0;
    }
  }
}
// the slack entries must not stay in the counters the scatter counts down: cleared after the scan; the pool's first free position = the total
__global__ __launch_bounds__(kBlock) void pm_row_finish_kernel(int* __restrict__ counts, const int* __restrict__ cs, int rows, int nx, int* __restrict__ pool_top) {
  for (int r = blockIdx.x * kBlock + threadIdx.x; r < rows; r += gridDim.x * kBlock) counts[(size_t)r * (nx + 1) + nx] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *pool_top = cs[(size_t)rows * (nx + 1)];
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_scatter_kernel(PmDev m, int n, const int* __restrict__ cell_id, int* __restrict__ counts) {
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < n; s += gridDim.x * kBlock) {
    const int c = cell_id[s];
    if (c < 0) continue;
    const int old = atomicSub(&counts[c], 1);
    const int pos = m.cs[c] + old - 1;
    P4 p = ((const P4*)m.pts)[s];
    p.i = (typename Scalar<P4>::index)s;
    ((P4*)m.spts)[pos] = p;
    if (m.nrm) ((P4*)m.snrm)[pos] = ((const P4*)m.nrm)[s];
    m.slot[s].pos = pos;
  }
}

// ---- an insertion -----------------------------------------------------------------------------------------------------------------------
// (1) place the scan (o3d_slam::transform: the arithmetic of transform_kernel), round to storage, and group the placed points that lie inside
// the volume by voxel: vox_insert_kernel's run lists on the WORLD-anchored key.  Points outside the volume go on a list.
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_place_kernel(const P4* __restrict__ spts, const P4* __restrict__ snrm, CountRef n_in, Mat34 M, double w0,
                                                          double w1, double w2, double w3, CropDev crop, PmDev m, VoxTable t, P4* __restrict__ placed,
                                                          P4* __restrict__ placed_nrm, int* __restrict__ lead_slot, int* __restrict__ run_next,
                                                          int* __restrict__ run_len) {
  using R = typename Scalar<P4>::type;
  const size_t n = count_of(n_in);
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) *t.cursor = 0u;
  for (size_t i0 = (size_t)blockIdx.x * kBlock; i0 < n; i0 += (size_t)gridDim.x * kBlock) {  // whole wavefronts iterate together
    const size_t i = i0 + threadIdx.x;
    unsigned long long k = kEmptyKey;
    if (i < n) {
      const P4 p = spts[i];
      const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
      const double w = w0 * x + w1 * y + w2 * z + w3;
      P4 o;
      o.x = (R)((M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3]) / w);
      o.y = (R)((M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7]) / w);
      o.z = (R)((M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11]) / w);
      o.i = (typename Scalar<P4>::index)i;
      placed[i] = o;
      if (snrm) {
        const P4 q = snrm[i];
        const double a = (double)q.x, b = (double)q.y, c = (double)q.z;
        P4 on;
        on.x = (R)(M.m[0] * a + M.m[1] * b + M.m[2] * c);
        on.y = (R)(M.m[4] * a + M.m[5] * b + M.m[6] * c);
        on.z = (R)(M.m[8] * a + M.m[9] * b + M.m[10] * c);
        on.i = 0;
        placed_nrm[i] = on;
      }
      if (crop_contains(crop, (double)o.x, (double)o.y, (double)o.z))
        k = pm_key(o, m.inv_voxel);
      else
        pm_push(m.outside_pts, m.counters + kPmOutside, m.list_cap, (int)i, m.counters + kPmError);
    }
    const unsigned long long kp = __shfl_up(k, 1, 64);
    const bool lead = k != kEmptyKey && (lane == 0 || kp != k);
    const unsigned long long ends = __ballot(lane == 0 || kp != k);
    int slot = -1;
    if (lead) {
      const unsigned long long above = lane == 63 ? 0ull : (ends >> (lane + 1));
      const int len = above ? (int)__builtin_ctzll(above) + 1 : 64 - lane;
      unsigned int sl = pm_hash(k) & t.mask;
      while (true) {
        const unsigned long long prev = atomicCAS(&t.s[sl].key, kEmptyKey, k);
        if (prev == kEmptyKey || prev == k) break;
        sl = (sl + 1) & t.mask;
      }
      atomicMin(&t.s[sl].first, (unsigned int)i);
      atomicAdd(&t.s[sl].nrun, 1u);
      run_next[i] = atomicExch(&t.s[sl].head, (int)i);
      run_len[i] = len;
      slot = (int)sl;
    }
    if (i < n) lead_slot[i] = slot;
  }
}

// a new slot for a point; -1 (and the error flag) when the arrays are full
__device__ __forceinline__ int pm_new_slot(const PmDev& m) {
  const int s = pm_ticket(m.counters + kPmN);
  if ((size_t)s >= m.cap) {
    atomicOr(m.counters + kPmError, 2);
    return -1;
  }
  return s;
}
// a slot enters the index: counted into its cell, its row marked, the slot listed (pm_rows_kernel makes the room, pm_place_new_kernel fills it)
__device__ __forceinline__ void pm_row_push(const PmDev& m, int s, unsigned long long key) {
  int row, x;
  pm_cell(m, key, &row, &x);
  atomicAdd(&m.cell_add[(size_t)row * (m.grid.nx + 1) + x], 1);
  if (pm_outside_grid(m, key)) atomicAdd(m.counters + kPmClamped, 1);
  // (a look before the exchange: a floor row receives hundreds of slots per scan, and all but the first find the row marked already)
  if (__hip_atomic_load(&m.row_flag[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && atomicExch(&m.row_flag[row], 1) == 0)
    pm_push(m.touched_rows, m.counters + kPmTouched, m.list_cap, row, m.counters + kPmError);
  pm_push(m.new_slots, m.counters + kPmNew, m.list_cap, s, m.counters + kPmError);
}
// the mean of a voxel's members written where it belongs: point, normal, history, search index (in place: the mean of points of one voxel
// lies in that voxel, hence in the same index cell), settled or not; `fresh`: the slot is new (its index entry comes with its row)
template <typename P4>
__device__ __forceinline__ void pm_store(const PmDev& m, int s, const P4& op, const P4& on, bool has_nrm, int t, unsigned long long key, bool fresh,
                                         int known_pos = -1) {
  P4 o = op;
  o.i = (typename Scalar<P4>::index)s;
  ((P4*)m.pts)[s] = o;
  if (has_nrm) ((P4*)m.nrm)[s] = on;
  m.slot[s].stamp = t;
  m.slot[s].okey = key;
  m.slot[s].out_checked = -1;  // (whatever pm_view_key remembered of the slot's history belonged to the point it held before)
  unsigned int fl = fresh ? kPmFresh : 0u;
  if (has_nrm && !pm_same_bits(pm_renormalized(on), on)) {
    fl |= kPmUnsettled;
    pm_push(m.unsettled[1], m.counters + kPmUnsettledOut, m.list_cap, s, m.counters + kPmError);
  }
  m.slot[s].flags = fl;
  if (!fresh) {
    const int pos = known_pos >= 0 ? known_pos : m.slot[s].pos;
    ((P4*)m.spts)[pos] = o;
    if (has_nrm) ((P4*)m.snrm)[pos] = on;
  }
}
template <typename P4>
__device__ __forceinline__ void pm_kill(const PmDev& m, int s) {
  using R = typename Scalar<P4>::type;
  m.slot[s].flags = kPmDead;
  atomicAdd(m.counters + kPmDeadCnt, 1);
  P4 far;  // never the nearest neighbour of anything: its squared distance is +inf
  far.x = far.y = far.z = sizeof(R) == 4 ? (R)3.0e38f : (R)1.0e300;
  far.i = (typename Scalar<P4>::index)0x7fffffff;
  ((P4*)m.spts)[m.slot[s].pos] = far;
}

// AccumulatedPoint (helpers.cpp:30-73) over the members of one voxel in the reference's order
struct PmAcc {
  double sx = 0, sy = 0, sz = 0, nx = 0, ny = 0, nz = 0;
  int cnt = 0;
  template <typename P4>
  __device__ __forceinline__ void add(const P4& p, bool has_n, const P4& nq) {
    sx += (double)p.x;
    sy += (double)p.y;
    sz += (double)p.z;
    if (has_n) {
      const double a = (double)nq.x, b = (double)nq.y, c = (double)nq.z;
      if (!(isnan(a) || isnan(b) || isnan(c))) {  // helpers.cpp:35-38
        nx += a;
        ny += b;
        nz += c;
      }
    }
    ++cnt;
  }
  template <typename P4>
  __device__ __forceinline__ void mean(P4* op, P4* on) {
    using R = typename Scalar<P4>::type;
    const double c = (double)cnt;
    op->x = (R)(sx / c);
    op->y = (R)(sy / c);
    op->z = (R)(sz / c);
    op->i = 0;
    double a = nx / c, b = ny / c, d = nz / c;
    pm_normalize(a, b, d);  // helpers.cpp:172
    on->x = (R)a;
    on->y = (R)b;
    on->z = (R)d;
    on->i = 0;
  }
};

// after a mean was stored under key k: does it still lie in voxel k?  (rounding to storage can put it on the far side of a face.)  If not it
// is re-hashed and re-indexed by the serial step.
template <typename P4>
__device__ __forceinline__ void pm_check_face(const PmDev& m, int s, const P4& op, unsigned long long k) {
  if (pm_key(op, m.inv_voxel) != k) pm_push(m.relink, m.counters + kPmRelink, m.list_cap, s, m.counters + kPmError);
}

// (3) one thread per voxel the scan touched (numbered by vox_order_kernel, their number in a device word): the voxel's runs in ascending
// order (as vox_mean_kernel), its chain in the voxel hash, the sum old member first -- the common cases, at most one old member inside the
// volume, are finished here; a voxel with more goes on the list of pm_special_kernel.
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_group_kernel(PmDev m, CountRef g_in, const int* __restrict__ order, const int* __restrict__ run_next,
                                                          const int* __restrict__ run_len, uint32_t* __restrict__ starts, int2* __restrict__ piece,
                                                          const P4* __restrict__ placed, const P4* __restrict__ placed_nrm, VoxTable t, CropDev crop,
                                                          int t_now, unsigned long long* __restrict__ group_key) {
  const size_t g = count_of(g_in);
  const int lane = threadIdx.x & 63;
  const bool has_nrm = m.nrm != nullptr;
  for (size_t r0 = (size_t)blockIdx.x * kBlock; r0 < g; r0 += (size_t)gridDim.x * kBlock) {  // whole wavefronts iterate together
    const size_t r = r0 + threadIdx.x;
    const bool have = r < g;
    int k = 0, head = -1;
    unsigned long long key = 0;
    if (have) {
      const int s = order[r];
      const VoxSlot v = t.s[s];
      k = (int)v.nrun + 1;
      head = v.head;
      key = v.key;
      VoxSlot e0;  // the scratch table is handed back empty (all 0xff), as vox_mean_kernel does
      e0.key = kEmptyKey, e0.first = ~0u, e0.head = -1, e0.nrun = ~0u, e0.pad[0] = e0.pad[1] = e0.pad[2] = ~0u;
      t.s[s] = e0;
    }
    // (the first probe of the voxel hash goes out now, beside the walks over the runs: two chains of dependent loads, side by side)
    const unsigned int e0 = pm_hash(key) & m.hmask;
    const PmHash h0 = have ? m.h[e0] : PmHash{kEmptyKey, -1, 0u};
    const unsigned long long hk0 = h0.key;
    const int hh0 = h0.head;
    // the runs in ascending order: looked for one by one (next_run), or -- the few voxels with many -- from a sorted list in the scratch array
    const int kk = k > kInlineRuns ? k : 0;
    int b = 0;
    if (__ballot(kk > 0) != 0ull) {
      int incl = kk;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
      }
      const int wave_total = __shfl(incl, 63, 64);
      unsigned int wbase = 0;
      if (lane == 0) wbase = atomicAdd(t.cursor, (unsigned int)wave_total);
      wbase = __shfl(wbase, 0, 64);
      b = (int)wbase + incl - kk;
      int node = head;
      for (int j = 0; j < kk; ++j) {
        starts[b + j] = (uint32_t)node;
        node = run_next[node];
      }
      if (kk > 1) sort_indices(starts + b, kk);
    }
    if (!have) continue;
    piece[r] = make_int2(head, k);  // (for pm_merge_kernel, should the voxel turn out to have several old members)
    group_key[r] = key;
    // old members of the voxel that lie inside the volume (they are in the voxel hash under the same key).  The first member's point and
    // normal are what nearly every voxel needs: fetched in the walk, kept.
    int old_slot = -1, n_old_in = 0, n_live = 0;
    unsigned int e = e0;
    int chain = hh0;
    if (hk0 != key) {  // not at its home entry: probe on (plain loads; the compare-and-swap only for a voxel the map has never seen)
      e = pm_find(m, key);
      if (e == ~0u) e = pm_entry(m, key);
      chain = m.h[e].head;
    }
    P4 old_p{}, old_n{};
    for (int s = chain; s != -1;) {
      const P4 p = ((const P4*)m.pts)[s];
      const P4 q = has_nrm ? ((const P4*)m.nrm)[s] : p;
      const PmSlot ms = m.slot[s];
      const int nxt = ms.hnext;
      if (ms.flags & kPmDead) {  // (carved: a carve leaves its dead in the chains, the next fold drops them)
        s = nxt;
        continue;
      }
      ++n_live;
      if (crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) {
        ++n_old_in;
        old_slot = s;
        old_p = p;
        old_n = q;
      }
      s = nxt;
    }
    if (n_old_in > 1) {
      m.h[e].info = (m.h[e].info & 0xffu) | ((unsigned int)t_now << 8);
      pm_push(m.complex_groups, m.counters + kPmComplex, m.list_cap, (int)r, m.counters + kPmError);
      continue;
    }
    PmAcc acc;
    if (n_old_in == 1) acc.add(old_p, has_nrm, old_n);
    int prev = -1;
    for (int j = 0; j < k; ++j) {
      const int st = kk ? (int)starts[b + j] : next_run(run_next, head, prev);
      prev = st;
      const int len = run_len[st];
      for (int q = 0; q < len; q += 4) {  // four points' loads in flight together (clamped), added in order
        P4 pp[4], nn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = st + min(q + u, len - 1);
          pp[u] = placed[idx];
          nn[u] = has_nrm ? placed_nrm[idx] : pp[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q + u < len) acc.add(pp[u], has_nrm, nn[u]);
      }
    }
    P4 op, on;
    acc.mean(&op, &on);
    int s = old_slot;
    const bool fresh = s < 0;
    if (fresh) {
      s = pm_new_slot(m);
      if (s < 0) continue;
      m.slot[s].hnext = m.h[e].head;  // (this thread is the only one that touches this voxel's chain in this launch)
      m.h[e].head = s;
      if (n_live > 0 && !(atomicOr(&m.h[e].info, 1u) & 1u)) pm_push64(m.multi[0], m.counters + kPmMultiOut, m.list_cap, key, m.counters + kPmError);
    }
    pm_store(m, s, op, on, has_nrm, t_now, key, fresh);
    if (fresh) pm_row_push(m, s, pm_key(op, m.inv_voxel));  // (the cell of where the mean really is: pm_check_face re-hashes it if that is another voxel)
    pm_check_face(m, s, op, key);
  }
}

// unlink slot s from the chain of hash entry e (serial contexts only)
__device__ __forceinline__ void pm_unlink(const PmDev& m, unsigned int e, int s) {
  int prev = -1;
  for (int c = m.h[e].head; c != -1; prev = c, c = m.slot[c].hnext) {
    if (c != s) continue;
    if (prev == -1)
      m.h[e].head = m.slot[c].hnext;
    else
      m.slot[prev].hnext = m.slot[c].hnext;
    return;
  }
}

// (4a) the voxels with several old members: those of the scan (listed by pm_group_kernel) and those on the multi list that the scan did
// not touch but whose members now lie inside the volume together.  One thread per voxel (a voxel's chain belongs to its thread; the lists
// are pushed to with atomics).  The members inside the volume are summed in the order of the reference's array before this insertion
// (pm_view_key at t - 1), then the scan's points of the voxel, into the first member's slot; the others die.
struct PmOld {
  unsigned long long hi[kPmMaxOld], lo[kPmMaxOld];
  int slot[kPmMaxOld];
};
template <typename P4>
__device__ __forceinline__ int pm_gather_old(const PmDev& m, unsigned int e, const CropDev& crop, int t_now, PmOld& o /* LDS */) {
  int n = 0;
  for (int s = m.h[e].head; s != -1; s = m.slot[s].hnext) {
    if (m.slot[s].flags & kPmDead) continue;
    const P4 p = ((const P4*)m.pts)[s];
    if (!crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) continue;
    if (n == kPmMaxOld) return kPmMaxOld + 1;  // more than the sorted list holds: pm_merge_many
    // Its place in the array before this insertion.  The usual pair is one point that passed through last time (it lay outside the
    // volume) and one voxel mean: the pass-through block comes first whatever its history, so the walk back through the volumes
    // (pm_view_key: as many steps as the point has been outside) is only taken when a SECOND pass-through member turns up (below).
    unsigned long long h, l;
    {
      const PmSlot ms = m.slot[s];
      if (pm_in_block(m, s, p, ms.stamp, ms.okey, t_now - 1)) {
        h = 1ull << 63;
        l = t_now - 1 == 0 ? (unsigned long long)s : (ms.stamp == t_now - 1 ? ms.okey : pm_key(p, m.inv_voxel));
      } else {
        h = 0;  // "somewhere in the pass-through block": exact only if it stays the only one
        l = ~0ull;
      }
    }
    int j = n++;
    while (j > 0 && (o.hi[j - 1] > h || (o.hi[j - 1] == h && o.lo[j - 1] > l))) {
      o.hi[j] = o.hi[j - 1], o.lo[j] = o.lo[j - 1], o.slot[j] = o.slot[j - 1];
      --j;
    }
    o.hi[j] = h, o.lo[j] = l, o.slot[j] = s;
  }
  if (n > 1 && o.hi[1] == 0) {  // several pass-through members: their order is their history
    int np = 0;
    while (np < n && o.hi[np] == 0) ++np;
    for (int a = 0; a < np; ++a) pm_view_key<P4>(m, o.slot[a], t_now - 1, &o.hi[a], &o.lo[a]);
    for (int a = 1; a < np; ++a) {  // insertion sort of the first np entries
      const unsigned long long h = o.hi[a], l = o.lo[a];
      const int sl = o.slot[a];
      int j = a;
      while (j > 0 && (o.hi[j - 1] > h || (o.hi[j - 1] == h && o.lo[j - 1] > l))) {
        o.hi[j] = o.hi[j - 1], o.lo[j] = o.lo[j - 1], o.slot[j] = o.slot[j - 1];
        --j;
      }
      o.hi[j] = h, o.lo[j] = l, o.slot[j] = sl;
    }
  }
  return n;
}
template <typename P4>
__device__ __forceinline__ void pm_merge_old(const PmDev& m, unsigned int e, unsigned long long key, const PmOld& o, int n_old, int group /* -1: no scan points */,
                                             const int2* __restrict__ piece, const int* __restrict__ run_next, const int* __restrict__ run_len,
                                             const P4* __restrict__ placed, const P4* __restrict__ placed_nrm, int t_now) {
  const bool has_nrm = m.nrm != nullptr;
  PmAcc acc;
  for (int j = 0; j < n_old; ++j) {
    const P4 p = ((const P4*)m.pts)[o.slot[j]];
    P4 q{};
    if (has_nrm) q = ((const P4*)m.nrm)[o.slot[j]];
    acc.add(p, has_nrm, q);
  }
  if (group >= 0) {
    const int2 pc = piece[group];  // {head of the run list, number of runs}
    int prev = -1;
    for (int j = 0; j < pc.y; ++j) {
      const int st = next_run(run_next, pc.x, prev);
      prev = st;
      const int len = run_len[st];
      for (int q = 0; q < len; ++q) {
        P4 nq{};
        if (has_nrm) nq = placed_nrm[st + q];
        acc.add(placed[st + q], has_nrm, nq);
      }
    }
  }
  P4 op, on;
  acc.mean(&op, &on);
  for (int j = 1; j < n_old; ++j) {
    pm_unlink(m, e, o.slot[j]);
    pm_kill<P4>(m, o.slot[j]);
  }
  pm_store(m, o.slot[0], op, on, has_nrm, t_now, key, false);
  pm_check_face(m, o.slot[0], op, key);
}
// the same for a voxel with more old members inside the volume than the sorted list holds (a map that entered with many raw points per
// voxel): no list -- the member next in order is looked for again for every addend (quadratic in the members; such voxels are few)
template <typename P4>
__device__ __forceinline__ void pm_merge_many(const PmDev& m, unsigned int e, unsigned long long key, const CropDev& crop, int group,
                                              const int2* __restrict__ piece, const int* __restrict__ run_next, const int* __restrict__ run_len,
                                              const P4* __restrict__ placed, const P4* __restrict__ placed_nrm, int t_now) {
  const bool has_nrm = m.nrm != nullptr;
  PmAcc acc;
  unsigned long long last_hi = 0, last_lo = 0;
  int first = -1;
  for (bool any = false;; any = true) {
    int best = -1;
    unsigned long long bh = ~0ull, bl = ~0ull;
    for (int s = m.h[e].head; s != -1; s = m.slot[s].hnext) {
      if (m.slot[s].flags & kPmDead) continue;
      const P4 p = ((const P4*)m.pts)[s];
      if (!crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) continue;
      unsigned long long h, l;
      pm_view_key<P4>(m, s, t_now - 1, &h, &l);
      if (any && !(h > last_hi || (h == last_hi && l > last_lo))) continue;  // summed already
      if (best == -1 || h < bh || (h == bh && l < bl)) best = s, bh = h, bl = l;
    }
    if (best == -1) break;
    const P4 p = ((const P4*)m.pts)[best];
    P4 q{};
    if (has_nrm) q = ((const P4*)m.nrm)[best];
    acc.add(p, has_nrm, q);
    if (first == -1) first = best;
    last_hi = bh, last_lo = bl;
  }
  if (first == -1) return;
  if (group >= 0) {
    const int2 pc = piece[group];  // {head of the run list, number of runs}
    int prev = -1;
    for (int j = 0; j < pc.y; ++j) {
      const int st = next_run(run_next, pc.x, prev);
      prev = st;
      const int len = run_len[st];
      for (int q = 0; q < len; ++q) {
        P4 nq{};
        if (has_nrm) nq = placed_nrm[st + q];
        acc.add(placed[st + q], has_nrm, nq);
      }
    }
  }
  P4 op, on;
  acc.mean(&op, &on);
  for (int s = m.h[e].head; s != -1;) {  // the other members inside the volume die
    const int nxt = m.slot[s].hnext;
    const P4 p = ((const P4*)m.pts)[s];
    if (s != first && !(m.slot[s].flags & kPmDead) && crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) {
      pm_unlink(m, e, s);
      pm_kill<P4>(m, s);
    }
    s = nxt;
  }
  pm_store(m, first, op, on, has_nrm, t_now, key, false);
  pm_check_face(m, first, op, key);
}

// The same in ONE walk over the chain.  A thread that merges a voxel is a chain of dependent loads -- hash entry, member after member, their
// points, their histories, their index positions -- and the kernel is as long as its longest chain: written as above (walk to count, walk to
// gather, walk to unlink each member that goes, slot records loaded again for their positions) that was fifty round trips to memory late
// in a long run, 70 us for a few hundred voxels.  Here every member's slot record and point are fetched together, once, into a small table
// in LDS; the order, the sum (all addends' loads in flight together), the deaths and the new links of the chain come from the table.
// Chains of more than kPmMaxOld members take the functions above.
constexpr int kPmNodeIn = 1 << 8, kPmNodeInBlock = 1 << 9, kPmNodeGone = 1 << 10;
struct PmNode {
  unsigned long long okey, kp;  // history key; voxel key of where the point is
  int s, pos, stamp, flags;     // slot, index position, stamp, slot flags | kPmNode*
};
struct PmNodes {
  PmNode n[kPmMaxOld];
};
// returns the number of chain members (all of them: the dead too), -1 if the chain is longer than the table
template <typename P4>
__device__ __forceinline__ int pm_walk_chain(const PmDev& m, unsigned int e, const CropDev& crop, const CropDev& prev_crop, int t_now, PmNodes& nd /* LDS */,
                                             int* live, int* inside, bool* touched) {
  const int t_prev = t_now - 1;
  int cnt = 0, lv = 0, in = 0;
  bool tch = false;
  for (int s = m.h[e].head; s != -1;) {
    if (cnt == kPmMaxOld) return -1;
    const PmSlot ms = m.slot[s];
    const P4 p = ((const P4*)m.pts)[s];
    int fl = (int)(ms.flags & 0xffu);
    if (!(ms.flags & kPmDead)) {
      ++lv;
      tch |= ms.stamp == t_now && !(ms.okey & kPmRaw);
      if (crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) {
        ++in;
        fl |= kPmNodeIn;
        const bool in_block = t_prev == 0 ? (s >= m.np_base && s < m.n_base)
                                          : (ms.stamp == t_prev ? !(ms.okey & kPmRaw) : crop_contains(prev_crop, (double)p.x, (double)p.y, (double)p.z));
        if (in_block) fl |= kPmNodeInBlock;
      }
    }
    PmNode& n = nd.n[cnt++];
    n.okey = ms.okey;
    n.kp = pm_key(p, m.inv_voxel);
    n.s = s;
    n.pos = ms.pos;
    n.stamp = ms.stamp;
    n.flags = fl;
    s = ms.hnext;
  }
  *live = lv, *inside = in, *touched = tch;
  return cnt;
}
// the members inside the volume (there are several) become one: summed in the order of the reference's array before this insertion, then the
// scan's points of the voxel (group >= 0), into the first member's slot; the others die and leave the chain.  In three steps, because the
// middle one is taken by the whole wavefront together: (1) the members inside the volume in the order of pm_gather_old's keys -- returns
// how many there are, and in *np how many of them need their history looked at (several pass-through members); (2) pm_view_key_wave for
// o.slot[0 .. np); (3) the rest.
template <typename P4>
__device__ __forceinline__ int pm_nodes_order(const PmDev& m, const PmNodes& nd, int cnt, PmOld& o /* LDS */, int t_now, int* np_out) {
  const int t_prev = t_now - 1;
  int n = 0;
  for (int i = 0; i < cnt; ++i) {
    const PmNode nn = nd.n[i];
    if (!(nn.flags & kPmNodeIn)) continue;
    unsigned long long h, l;
    if (nn.flags & kPmNodeInBlock) {
      h = 1ull << 63;
      l = t_prev == 0 ? (unsigned long long)nn.s : (nn.stamp == t_prev ? nn.okey : nn.kp);
    } else {
      h = 0;
      l = ~0ull;
    }
    int j = n++;
    while (j > 0 && (o.hi[j - 1] > h || (o.hi[j - 1] == h && o.lo[j - 1] > l))) {
      o.hi[j] = o.hi[j - 1], o.lo[j] = o.lo[j - 1], o.slot[j] = o.slot[j - 1];
      --j;
    }
    o.hi[j] = h, o.lo[j] = l, o.slot[j] = i;
  }
  int np = 0;
  if (n > 1 && o.hi[1] == 0)  // several pass-through members: their order is their history
    while (np < n && o.hi[np] == 0) ++np;
  *np_out = np;
  return n;
}
template <typename P4>
__device__ __forceinline__ void pm_nodes_finish(const PmDev& m, unsigned int e, unsigned long long key, PmNodes& nd, int cnt, PmOld& o /* LDS */, int n, int np,
                                                int group, const int2* __restrict__ piece, const int* __restrict__ run_next,
                                                const int* __restrict__ run_len, const P4* __restrict__ placed, const P4* __restrict__ placed_nrm, int t_now) {
  using R = typename Scalar<P4>::type;
  const bool has_nrm = m.nrm != nullptr;
  for (int a = 1; a < np; ++a) {  // the pass-through members by their histories (insertion sort of the first np entries)
    const unsigned long long h = o.hi[a], l = o.lo[a];
    const int sl = o.slot[a];
    int j = a;
    while (j > 0 && (o.hi[j - 1] > h || (o.hi[j - 1] == h && o.lo[j - 1] > l))) {
      o.hi[j] = o.hi[j - 1], o.lo[j] = o.lo[j - 1], o.slot[j] = o.slot[j - 1];
      --j;
    }
    o.hi[j] = h, o.lo[j] = l, o.slot[j] = sl;
  }
  PmAcc acc;
  for (int j0 = 0; j0 < n; j0 += 4) {  // four members' loads in flight together (clamped), added in order
    P4 pp[4], qq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sl = nd.n[o.slot[min(j0 + u, n - 1)]].s;
      pp[u] = ((const P4*)m.pts)[sl];
      qq[u] = has_nrm ? ((const P4*)m.nrm)[sl] : pp[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + u < n) acc.add(pp[u], has_nrm, qq[u]);
  }
  if (group >= 0) {
    const int2 pc = piece[group];  // {head of the run list, number of runs}
    int prev = -1;
    for (int j = 0; j < pc.y; ++j) {
      const int st = next_run(run_next, pc.x, prev);
      prev = st;
      const int len = run_len[st];
      for (int q = 0; q < len; ++q) {
        P4 nq{};
        if (has_nrm) nq = placed_nrm[st + q];
        acc.add(placed[st + q], has_nrm, nq);
      }
    }
  }
  P4 op, on;
  acc.mean(&op, &on);
  P4 far;  // (pm_kill's sentinel)
  far.x = far.y = far.z = sizeof(R) == 4 ? (R)3.0e38f : (R)1.0e300;
  far.i = (typename Scalar<P4>::index)0x7fffffff;
  for (int j = 1; j < n; ++j) {
    PmNode& g = nd.n[o.slot[j]];
    g.flags |= kPmNodeGone;
    m.slot[g.s].flags = kPmDead;
    ((P4*)m.spts)[g.pos] = far;
  }
  atomicAdd(m.counters + kPmDeadCnt, n - 1);
  int prev_s = -1;  // the chain without the members that went, in the order it had
  bool gap = false;
  for (int i = 0; i < cnt; ++i) {
    const int fl = nd.n[i].flags, si = nd.n[i].s;
    if (fl & kPmNodeGone) {
      gap = true;
      continue;
    }
    if (gap) {
      if (prev_s == -1)
        m.h[e].head = si;
      else
        m.slot[prev_s].hnext = si;
      gap = false;
    }
    prev_s = si;
  }
  if (gap) m.slot[prev_s].hnext = -1;  // (the member that stays is never the one that went: prev_s is a slot)
  const PmNode keep = nd.n[o.slot[0]];
  pm_store(m, keep.s, op, on, has_nrm, t_now, key, false, keep.pos);
  pm_check_face(m, keep.s, op, key);
}

template <typename P4>
__global__ __launch_bounds__(64) void pm_merge_kernel(PmDev m, const int2* __restrict__ piece, const int* __restrict__ run_next, const int* __restrict__ run_len,
                                                      const P4* __restrict__ placed, const P4* __restrict__ placed_nrm,
                                                      const unsigned long long* __restrict__ group_key /* [groups] key of group r */, CropDev crop, int t_now) {
  __shared__ PmOld s_old[64];
  __shared__ PmNodes s_nodes[64];
  PmOld& o = s_old[threadIdx.x];
  PmNodes& nd = s_nodes[threadIdx.x];
  const int cap = m.list_cap;
  const int n_complex = min(m.counters[kPmComplex], cap), n_multi = min(m.counters[kPmMultiIn], cap);
  const CropDev prev_crop = m.hist[max(t_now - 1, 0)];  // (entry 0 is unused: insertion 0 is the base, told by the slot's number)
  for (int i0 = blockIdx.x * 64; i0 < n_complex + n_multi; i0 += gridDim.x * 64) {  // the wavefront iterates together (pm_view_key_wave)
    const int i = i0 + threadIdx.x;
    // (a) per lane: the voxel, its chain, whether its members inside the volume merge
    unsigned long long key = kEmptyKey;
    unsigned int e = ~0u;
    int group = -1, cnt = 0, live = 0, inside = 0;
    bool merge = false, listed = false;  // listed: an entry of the multi list whose voxel was looked at (its place on the list is decided below)
    if (i < n_complex) {  // a voxel of the scan with several old members inside the volume
      group = m.complex_groups[i];
      key = group_key[group];
      e = pm_entry(m, key);
      bool touched;
      cnt = pm_walk_chain<P4>(m, e, crop, prev_crop, t_now, nd, &live, &inside, &touched);
      if (cnt >= 0) {
        merge = true;
      } else {  // a chain longer than the table
        const int n_old = pm_gather_old<P4>(m, e, crop, t_now, o);
        if (n_old > kPmMaxOld)
          pm_merge_many<P4>(m, e, key, crop, group, piece, run_next, run_len, placed, placed_nrm, t_now);
        else
          pm_merge_old<P4>(m, e, key, o, n_old, group, piece, run_next, run_len, placed, placed_nrm, t_now);
      }
    } else if (i < n_complex + n_multi) {
      // a voxel on the multi list.  One the scan touched has been dealt with: its group saw the whole chain (a complex group of this very
      // launch belongs to another thread, which may be rewriting the chain right now: told apart by the list of this launch, not by the
      // chain).  The list is not rewritten from insertion to insertion (that was an atomic ticket and a store per entry and insertion,
      // 20 000 entries in a map of a million points): an entry stays where it is until its voxel is down to one member, then it is blanked.
      key = m.multi[0][i - n_complex];
      // Most listed voxels lie far from where anything happens: a voxel that straddled the volume's outer boundary when the sensor passed
      // stays listed (two members that were never inside together) for as long as the volume stays away.  A voxel that lies ENTIRELY
      // outside the volume has no member inside it, so nothing merges and the scan cannot have touched it: told from the key alone --
      // no load.  (Conservative by the voxel's half diagonal plus a margin; only for the plain radius / cylinder volumes.)
      if (key != kEmptyKey && !pm_voxel_outside(m, key, crop)) {
        e = pm_find(m, key);
        if (e == ~0u) {
          m.multi[0][i - n_complex] = kEmptyKey;
        } else if ((int)(m.h[e].info >> 8) != t_now) {  // (== t_now: pm_group_kernel handed it to the other branch of this launch; stays listed)
          listed = true;
          bool touched = false;
          cnt = pm_walk_chain<P4>(m, e, crop, prev_crop, t_now, nd, &live, &inside, &touched);
          if (cnt >= 0) {
            merge = !touched && inside > 1;
          } else {  // a chain longer than the table: walk by walk
            live = inside = 0;
            for (int s = m.h[e].head; s != -1; s = m.slot[s].hnext) {
              if (m.slot[s].flags & kPmDead) continue;
              ++live;
              touched |= m.slot[s].stamp == t_now && !(m.slot[s].okey & kPmRaw);
              const P4 p = ((const P4*)m.pts)[s];
              inside += crop_contains(crop, (double)p.x, (double)p.y, (double)p.z) ? 1 : 0;
            }
            if (!touched && inside > 1) {
              const int n_old = pm_gather_old<P4>(m, e, crop, t_now, o);
              if (n_old > kPmMaxOld)
                pm_merge_many<P4>(m, e, key, crop, -1, piece, run_next, run_len, placed, placed_nrm, t_now);
              else if (n_old > 1)
                pm_merge_old<P4>(m, e, key, o, n_old, -1, piece, run_next, run_len, placed, placed_nrm, t_now);
              live -= inside - 1;
            }
          }
        }
      }
    }
    // (b) the members in order; the histories of several pass-through members by the whole wavefront, lane after lane
    int n = 0, np = 0;
    if (merge) n = pm_nodes_order<P4>(m, nd, cnt, o, t_now, &np);
    for (int a = 0; a < kPmMaxOld; ++a) {
      const bool want = a < np;
      if (__ballot(want) == 0ull) break;
      pm_view_key_wave<P4>(m, want, want ? nd.n[o.slot[a]].s : 0, t_now - 1, &o.hi[a], &o.lo[a]);
    }
    // (c) per lane again
    if (merge) {
      pm_nodes_finish<P4>(m, e, key, nd, cnt, o, n, np, group, piece, run_next, run_len, placed, placed_nrm, t_now);
      live -= inside - 1;
    }
    if (listed && live <= 1) {  // settled: off the list
      m.multi[0][i - n_complex] = kEmptyKey;
      m.h[e].info &= ~1u;
    }
  }
}

// (4b) the rest.  In parallel, the normals that are not a fixed point yet: a slot that nothing wrote this time and that lies inside the
// volume is alone in its voxel there -- the reference replaces its normal by the re-normalised one (its point by itself).  And, on ONE
// thread (a handful of items, and what they do to the chains needs no locks that way): the scan points outside the volume (they pass
// through, i.e. join the map as they were placed) and the means that were rounded across a voxel face (chain of the old key -> chain of
// the key they have now; the index entry of an old slot is killed and comes back with the rows, a new slot is on its row's list already).
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_misc_kernel(PmDev m, const P4* __restrict__ placed, const P4* __restrict__ placed_nrm, CropDev crop, int t_now) {
  int* err = m.counters + kPmError;
  const bool has_nrm = m.nrm != nullptr;
  const int cap = m.list_cap;
  const int n_uns = min(m.counters[kPmUnsettledIn], cap);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_uns; i += gridDim.x * kBlock) {
    const int s = m.unsettled[0][i];
    const unsigned int fl = m.slot[s].flags;
    if ((fl & kPmDead) || !(fl & kPmUnsettled) || m.slot[s].stamp == t_now) continue;  // (a slot written now was listed again by pm_store if need be)
    const P4 p = ((const P4*)m.pts)[s];
    if (crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) {
      const P4 nv = ((const P4*)m.nrm)[s];
      const P4 nn = pm_renormalized(nv);
      ((P4*)m.nrm)[s] = nn;
      ((P4*)m.snrm)[m.slot[s].pos] = nn;
      if (pm_same_bits(pm_renormalized(nn), nn)) {
        m.slot[s].flags = fl & ~kPmUnsettled;
        continue;
      }
    }
    pm_push(m.unsettled[1], m.counters + kPmUnsettledOut, cap, s, err);
  }
  if (blockIdx.x != gridDim.x - 1 || threadIdx.x != 0) return;
  const int n_out = min(m.counters[kPmOutside], cap);
  for (int i = 0; i < n_out; ++i) {
    const int si = m.outside_pts[i];
    const int s = pm_new_slot(m);
    if (s < 0) break;
    P4 o = placed[si];
    const unsigned long long key = pm_key(o, m.inv_voxel);
    o.i = (typename Scalar<P4>::index)s;
    ((P4*)m.pts)[s] = o;
    P4 on{};
    if (has_nrm) {
      on = placed_nrm[si];
      ((P4*)m.nrm)[s] = on;
    }
    m.slot[s].stamp = t_now;
    m.slot[s].okey = kPmRaw | (unsigned long long)si;
    m.slot[s].out_checked = -1;
    unsigned int fl = kPmFresh;
    if (has_nrm && !pm_same_bits(pm_renormalized(on), on)) {
      fl |= kPmUnsettled;
      pm_push(m.unsettled[1], m.counters + kPmUnsettledOut, cap, s, err);
    }
    m.slot[s].flags = fl;
    const unsigned int e = pm_entry(m, key);
    m.slot[s].hnext = m.h[e].head;
    const bool had = m.h[e].head != -1;
    m.h[e].head = s;
    if (had && !(m.h[e].info & 1u)) {
      m.h[e].info |= 1u;
      pm_push64(m.multi[0], m.counters + kPmMultiOut, cap, key, err);
    }
    pm_row_push(m, s, key);
  }
  const int n_rel = min(m.counters[kPmRelink], cap);
  for (int i = 0; i < n_rel; ++i) {
    const int s = m.relink[i];
    if (m.slot[s].flags & kPmDead) continue;
    const P4 p = ((const P4*)m.pts)[s];
    const unsigned long long k_old = m.slot[s].okey, k_new = pm_key(p, m.inv_voxel);
    pm_unlink(m, pm_entry(m, k_old), s);
    const unsigned int e = pm_entry(m, k_new);
    const bool had = m.h[e].head != -1;
    m.slot[s].hnext = m.h[e].head;
    m.h[e].head = s;
    if (had && !(m.h[e].info & 1u)) {
      m.h[e].info |= 1u;
      pm_push64(m.multi[0], m.counters + kPmMultiOut, cap, k_new, err);
    }
    if (!(m.slot[s].flags & kPmFresh)) {  // in the index, in the cell of its old voxel (a new slot is listed under the cell it really is in)
      using R = typename Scalar<P4>::type;
      P4 far;
      far.x = far.y = far.z = sizeof(R) == 4 ? (R)3.0e38f : (R)1.0e300;
      far.i = (typename Scalar<P4>::index)0x7fffffff;
      ((P4*)m.spts)[m.slot[s].pos] = far;
      pm_row_push(m, s, k_new);
    }
  }
}

// (5) the rows of the index that receive slots: one workgroup per row makes the room.  The row's cells keep their order; every cell grows by
// the slots counted into it (cell_add), so its old points move up by the number of new ones in the cells before it -- inside the row's
// region if that has the room, into a fresh region at the end of the pool (twice the need) otherwise.  The old points are moved from the
// back in chunks of a wavefront (all of a chunk is read before any of it is written, and a point only ever moves up).  The new slots
// themselves are written by pm_place_new_kernel, one thread each, into the gaps this leaves at the end of every cell.
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_rows_kernel(PmDev m) {
  extern __shared__ int s_cells[];  // [3][nx + 1]: old starts, slots added per cell, new starts
  __shared__ int s_wave[kBlock / 64];
  __shared__ int s_base, s_first;
  const int nx = m.grid.nx, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int* s_old = s_cells;
  int* s_add = s_cells + (nx + 1);
  int* s_new = s_cells + 2 * (nx + 1);
  const int n_rows = min(m.counters[kPmTouched], m.list_cap);
  const bool has_nrm = m.snrm != nullptr;
  for (int ri = blockIdx.x; ri < n_rows; ri += gridDim.x) {  // one workgroup per row (a floor row of a lidar map holds thousands of points)
    const int row = m.touched_rows[ri];
    int* cs = m.cs + (size_t)row * (nx + 1);
    int* add = m.cell_add + (size_t)row * (nx + 1);
    int n_new = 0, first = nx;
    for (int x = tid; x <= nx; x += kBlock) {
      s_old[x] = cs[x];
      const int a = x < nx ? add[x] : 0;
      s_add[x] = a;
      n_new += a;
      if (a > 0) first = min(first, x);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      n_new += __shfl_xor(n_new, d, 64);
      first = min(first, __shfl_xor(first, d, 64));
    }
    if (lane == 0) s_wave[w] = n_new;
    if (tid == 0) s_first = nx;
    __syncthreads();
    if (lane == 0) atomicMin(&s_first, first);
    n_new = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    static_assert(kBlock / 64 == 4, "four wavefronts per workgroup");
    __syncthreads();
    const int old_start = s_old[0], old_end = s_old[nx], old_cnt = old_end - old_start;
    const int need = old_cnt + n_new;
    if (tid == 0) {
      int base = old_start;
      if (need > m.row_cap[row]) {  // the row moves to the end of the pool
        const int cap_new = 2 * need + 8;
        base = atomicAdd(m.counters + kPmPoolTop, cap_new);
        if (base + cap_new > m.pool_cap) {
          atomicOr(m.counters + kPmError, 4);
          base = -1;
        } else {
          m.row_cap[row] = cap_new;
        }
      }
      s_base = base;
    }
    __syncthreads();
    const int base = s_base;
    if (base < 0) {  // (never: the host sizes the pool for the worst case) -- the row's slots stay out of the index
      for (int x = tid; x < nx; x += kBlock) add[x] = 0;
      if (tid == 0) m.row_flag[row] = 0;
      __syncthreads();
      continue;
    }
    if (w == 0) {  // new starts: exclusive scan over (old count + added) per cell, 64 cells at a time
      int run = base;
      for (int x0 = 0; x0 <= nx; x0 += 64) {
        const int x = x0 + lane;
        const int c = x < nx ? (s_old[x + 1] - s_old[x]) + s_add[x] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int y = __shfl_up(incl, d, 64);
          if (lane >= d) incl += y;
        }
        if (x <= nx) s_new[x] = run + incl - c;
        run += __shfl(incl, 63, 64);
      }
    }
    __syncthreads();
    // the old points from the back, a workgroup's worth per round; the cells in front of the first one that grows stay where they are
    // (unless the whole row moved)
    const int stop = base == old_start ? s_old[min(s_first, nx - 1)] : old_start;
    constexpr int kPer = 4;  // points per thread and round: a floor row of a dense map holds thousands of points, and a round is two barriers
    for (int hi = old_end; hi > stop; hi -= kBlock * kPer) {
      P4 p[kPer], q[kPer];
      int j[kPer], dst[kPer];
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        j[u] = hi - kBlock * (u + 1) + tid;
        const int jc = max(j[u], old_start);  // (clamped, not predicated: the loads stay out of the branch)
        p[u] = ((const P4*)m.spts)[jc];
        q[u] = ((const P4*)(has_nrm ? m.snrm : m.spts))[jc];
      }
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        dst[u] = -1;
        if (j[u] >= stop) {
          int lo = 0, hh = nx - 1;  // its cell: the last x with s_old[x] <= j
          while (lo < hh) {
            const int mid = (lo + hh + 1) >> 1;
            if (s_old[mid] <= j[u])
              lo = mid;
            else
              hh = mid - 1;
          }
          dst[u] = s_new[lo] + (j[u] - s_old[lo]);
        }
      }
      __syncthreads();  // all of the round is read before any of it is written (a point only ever moves up: into this round's range or beyond)
#pragma unroll
      for (int u = 0; u < kPer; ++u)
        if (dst[u] >= 0 && dst[u] != j[u]) {
          P4 pp, qq;  // (field by field: an aggregate copy through a conditional branch went through scratch memory)
          pp.x = p[u].x, pp.y = p[u].y, pp.z = p[u].z, pp.i = p[u].i;
          qq.x = q[u].x, qq.y = q[u].y, qq.z = q[u].z, qq.i = q[u].i;
          ((P4*)m.spts)[dst[u]] = pp;
          if (has_nrm) ((P4*)m.snrm)[dst[u]] = qq;
          if ((int)p[u].i != 0x7fffffff) m.slot[(int)p[u].i].pos = dst[u];
        }
      __syncthreads();
    }
    for (int x = tid; x <= nx; x += kBlock) cs[x] = s_new[x];
    if (tid == 0) m.row_flag[row] = 0;
    __syncthreads();
  }
}
// ... and the slots that enter: the k-th arrival of a cell takes the k-th place from the cell's end (the counters run back down to zero: the
// block of counters needs no clearing).  The order inside a cell is the order the atomics are served in; nothing depends on it.
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_place_new_kernel(PmDev m) {
  const int n_new = min(m.counters[kPmNew], m.list_cap);
  const bool has_nrm = m.snrm != nullptr;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_new; i += gridDim.x * kBlock) {
    const int s = m.new_slots[i];
    P4 p = ((const P4*)m.pts)[s];
    int row, x;
    pm_cell(m, pm_key(p, m.inv_voxel), &row, &x);
    const size_t c = (size_t)row * (m.grid.nx + 1) + x;
    const int k = atomicSub(&m.cell_add[c], 1);
    if (k <= 0) {  // (the row found no room: see pm_rows_kernel)
      atomicAdd(&m.cell_add[c], 1);
      continue;
    }
    const int dst = m.cs[c + 1] - k;
    p.i = (typename Scalar<P4>::index)s;
    ((P4*)m.spts)[dst] = p;
    if (has_nrm) ((P4*)m.snrm)[dst] = ((const P4*)m.nrm)[s];
    m.slot[s].pos = dst;
    m.slot[s].flags &= ~kPmFresh;
  }
}

// bookkeeping between two insertions: the "out" lists become the "in" lists, the per-insertion lists are emptied; the number of slots goes
// to the pinned record
__global__ __launch_bounds__(64) void pm_turn_kernel(PmDev m, CountPub pub, CropDev* __restrict__ hist, CropDev crop, int t_now, double* __restrict__ host_vals) {
  if (threadIdx.x != 0) return;
  int* c = m.counters;
  hist[t_now] = crop;  // the volume of this insertion joins the history (pm_view_key)
  if (host_vals) {     // what the host sizes the next insertion by: first free index position, dead slots, sticky error
    host_vals[0] = (double)c[kPmPoolTop];
    host_vals[1] = (double)c[kPmDeadCnt];
    host_vals[2] = (double)c[kPmError];
    host_vals[3] = (double)c[kPmMultiOut];                    // entries of the multi list so far; voxels of this scan with several old
    host_vals[4] = (double)load_then_store(c + kPmComplex);   // members inside)
    host_vals[5] = (double)c[kPmClamped];                     // slots outside the index grid so far
  }
  c[kPmUnsettledIn] = min(load_then_store(c + kPmUnsettledOut), m.list_cap);  // (read and reset by this thread: see common.hpp)
  c[kPmUnsettledOut] = 0;
  c[kPmMultiIn] = min(c[kPmMultiOut], m.list_cap);  // (the multi list is appended to, never rewritten: what it holds now is what the next insertion looks at)
  c[kPmComplex] = c[kPmOutside] = c[kPmRelink] = c[kPmTouched] = c[kPmNew] = 0;
  publish_count(pub, c[kPmN]);
}

// ---- Submap::carve (Submap.cpp:109-125 -> getIdxsOfCarvedPoints, helpers.cpp:235-271) on the persistent form -----------------------------
// The reference bins the map points inside the cropping volume by voxel and lets every scan ray probe that table; here the table exists
// already -- the voxel hash, when the carving voxel is the map's voxel (the shipped configuration: 0.1 m both).  Same rays, same samples
// (carve_rays_kernel's arithmetic), same test per point (inside the volume; |ray . unit normal| > min_dot or no normals); a point that
// goes is marked, listed once, and killed by pm_carve_apply_kernel: dead flag, its index entry replaced by the far sentinel.  It stays
// in its chain (many rays reach one voxel at the same time: no unlinking here), chain walks skip the dead, the next fold drops them.
// The survivors keep their order by construction: the order is a function of the slots' histories (pm_view_key), not of an array.
__global__ __launch_bounds__(kBlock) void pm_carve_bits_kernel(const PmHash* __restrict__ hh, size_t n, unsigned int* __restrict__ block_bits) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const PmHash e = hh[i];
    if (e.key == kEmptyKey || e.head == -1) continue;
    const unsigned int bit = carve_block_bit(e.key);
    atomicOr(&block_bits[bit >> 5], 1u << (bit & 31u));
  }
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_carve_rays_kernel(PmDev m, const P4* __restrict__ scan, size_t n_scan, Mat34 M /* map <- sensor */, double sx,
                                                               double sy, double sz, double max_len, double trunc, double min_dot, CropDev crop,
                                                               const unsigned int* __restrict__ block_bits) {
  const double voxel = m.voxel, inv = 1.0 / voxel;
  const bool has_nrm = m.nrm != nullptr;
  {  // one ray per thread (no loop over rays: the pose and the scan's address are dead after these lines, which is what keeps the kernel's
     // scalar registers within the file)
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_scan) return;
    const P4 q = scan[i];
    const double x = (double)q.x, y = (double)q.y, z = (double)q.z;
    const double px = M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3], py = M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7],
                 pz = M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11];
    const double dx = px - sx, dy = py - sy, dz = pz - sz;
    const double length = sqrt(dx * dx + dy * dy + dz * dz);
    if (!(length > 0.0)) return;
    const double ux = dx / length, uy = dy / length, uz = dz / length;
    const double lim = fmax(voxel, fmin(length - trunc, max_len));
    constexpr int kBatch = 8;
    double dist = 0.0;
    while (dist < lim) {
      unsigned long long ks[kBatch];
      unsigned int bits[kBatch], words[kBatch];
      int nb = 0;
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        if (dist < lim) {
          const double cx = dist * ux + sx, cy = dist * uy + sy, cz = dist * uz + sz;
          ks[u] = pack_key((long long)(int)floor(cx * inv), (long long)(int)floor(cy * inv), (long long)(int)floor(cz * inv));
          bits[u] = carve_block_bit(ks[u]);
          dist += voxel;
          nb = u + 1;
        }
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        if (u < nb) words[u] = block_bits[bits[u] >> 5];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        if (u >= nb) break;
        if (!((words[u] >> (bits[u] & 31u)) & 1u)) continue;  // nothing of the map in this block of voxels
        const unsigned int e = pm_find(m, ks[u]);
        if (e == ~0u) continue;
        for (int s = m.h[e].head; s != -1; s = m.slot[s].hnext) {
          const unsigned int fl = m.slot[s].flags;
          if (fl & (kPmDead | kPmCarved)) continue;
          const P4 p = ((const P4*)m.pts)[s];
          if (!crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) continue;  // (the reference's table holds the points inside the volume)
          bool rem = true;
          if (has_nrm) {
            const P4 nn = ((const P4*)m.nrm)[s];
            const double a = (double)nn.x, bb = (double)nn.y, c = (double)nn.z;
            const double nl = sqrt(a * a + bb * bb + c * c);
            const double dot = nl > 0.0 ? (ux * a + uy * bb + uz * c) / nl : 0.0;  // Eigen normalized(): the zero vector stays zero
            rem = fabs(dot) > min_dot;
          }
          if (rem && !(atomicOr(&m.slot[s].flags, kPmCarved) & kPmCarved)) {
            const int k = atomicAdd(m.counters + kPmCarvedCnt, 1);  // (one ticket per removal: a carve removes a few hundred points)
            if (k < m.list_cap)
              m.new_slots[k] = s;
            else
              atomicOr(m.counters + kPmError, 1);
          }
        }
      }
    }
  }
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_carve_apply_kernel(PmDev m, CountPub pub, double* __restrict__ host_vals) {
  using R = typename Scalar<P4>::type;
  const int n = min(m.counters[kPmCarvedCnt], m.list_cap);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int s = m.new_slots[i];
    m.slot[s].flags = kPmDead;
    P4 far;
    far.x = far.y = far.z = sizeof(R) == 4 ? (R)3.0e38f : (R)1.0e300;
    far.i = (typename Scalar<P4>::index)0x7fffffff;
    ((P4*)m.spts)[m.slot[s].pos] = far;
  }
  // (the last workgroup to get here would be the place to fold the count into the dead counter; done by pm_carve_finish_kernel, one launch on)
  (void)pub;
  (void)host_vals;
}
__global__ __launch_bounds__(64) void pm_carve_finish_kernel(PmDev m, CountPub pub, double* __restrict__ host_vals) {
  if (threadIdx.x != 0) return;
  int* c = m.counters;
  const int n = min(load_then_store(c + kPmCarvedCnt), m.list_cap);  // (see common.hpp: read and reset by the same thread)
  const int dead = load_then_store(c + kPmDeadCnt) + n;
  c[kPmDeadCnt] = dead;
  c[kPmCarvedCnt] = 0;
  host_vals[0] = (double)c[kPmPoolTop];
  host_vals[1] = (double)dead;
  host_vals[2] = (double)c[kPmError];
  host_vals[3] = (double)c[kPmMultiOut];  // (what pm_poll folds into its multi-list total: unchanged by a carve)
  host_vals[4] = 0.0;
  host_vals[5] = (double)c[kPmClamped];
  host_vals[6] = (double)n;  // removed by this carve (PinRec::pad: read by o3ds_map_carve only)
  publish_count(pub, c[kPmN]);
}

// ---- leaving the persistent form: the reference's array --------------------------------------------------------------------------------
// 128-bit position key of every live slot (dead ones sort behind everything and are counted)
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_view_key_kernel(PmDev m, int n, int t_last, unsigned long long* __restrict__ hi, unsigned long long* __restrict__ lo,
                                                             uint32_t* __restrict__ val, unsigned long long* __restrict__ n_pass) {
  unsigned long long cnt = 0;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < n; s += gridDim.x * kBlock) {
    unsigned long long h = ~0ull, l = ~0ull;
    if (!(m.slot[s].flags & kPmDead)) {
      pm_view_key<P4>(m, s, t_last, &h, &l);
      if (!(h >> 63)) ++cnt;
    }
    hi[s] = h;
    lo[s] = l;
    val[s] = (uint32_t)s;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(n_pass, cnt);
}
__global__ __launch_bounds__(kBlock) void pm_gather_u64_kernel(const unsigned long long* __restrict__ in, const uint32_t* __restrict__ idx, size_t n,
                                                               unsigned long long* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) out[i] = in[idx[i]];
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void pm_permute_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, const uint32_t* __restrict__ idx, size_t n,
                                                            P4* __restrict__ out_pts, P4* __restrict__ out_nrm) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    P4 p = pts[idx[i]];
    p.i = (typename Scalar<P4>::index)i;
    out_pts[i] = p;
    if (nrm) out_nrm[i] = nrm[idx[i]];
  }
}

#pragma clang fp contract(fast)

}  // namespace o3ds
