// icp_kernels.hpp -- nearest-neighbour index build and the ICP pass kernels for gfx950.
//
// Replaces, for the reference's registerClouds seams (CloudRegistration.cpp:16-21, 44-48, 69-74), the Open3D v0.15.1 routines
//   KDTreeFlann::SetGeometry              -> grid index build  (bbox -> cell count -> scan -> scatter)
//   GetRegistrationResultAndCorrespondences
//   + PointCloud::Transform
//   + TransformationEstimationPointToPlane / ...ForGeneralizedICP / ...PointToPoint ::ComputeTransformation (the reductions)
//   + SolveJacobianSystemAndObtainExtrinsicMatrix (or Eigen::umeyama) + the convergence test of the PREVIOUS iteration
//                                         -> ONE kernel per iteration: icp_fused_kernel (default form)
//   the same split in two / three kernels -> icp_accumulate_kernel + icp_reduce_update_kernel (O3DS_ICP_MODE=launch),
//                                            icp_accumulate_kernel + icp_reduce_kernel + icp_update_kernel (classic sharded form)
// Record sums are order-independent (split_exact), so all forms agree bit for bit.  DESIGN.md section 4 has the measurements.
#pragma once
#include "common.hpp"

// NO implicit fused multiply-add contraction in this file: whether a*b+c becomes one rounding or two is otherwise the compiler's choice
// per inlined copy, and the same source runs in several kernels (fused / two-launch / step-wise) and, inside one kernel, in several
// copies (a correspondence is served by the search or by its verified candidate set) that must agree bit for bit.  Unfused is also how
// the reference rounds (-O3 without -march: DESIGN.md section 2), so the transform, the residuals and the Jacobian rows are the
// reference's arithmetic; the spots where a fused operation is wanted (the f32 candidate distance, the per-workgroup accumulation, the
// 6x6 solve) say fma() explicitly.
#pragma clang fp contract(off)

namespace o3ds {

constexpr int kBlock = 256;  // 4 wavefronts of 64

// ----------------------------------------------------------------------------------------------
// wave / block reductions (64-wide wavefronts)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// ----------------------------------------------------------------------------------------------
// index build
// ----------------------------------------------------------------------------------------------
// per-block bounding boxes of the points inside `crop` (O3DS_CROP_NONE: all): out[block*6 + {0..2}] = min, {3..5} = max
template <typename P4>
__global__ __launch_bounds__(kBlock) void bbox_kernel(const P4* __restrict__ pts, size_t n, CropDev crop, double* __restrict__ out) {
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    if (!crop_contains(crop, (double)p.x, (double)p.y, (double)p.z)) continue;
    mn[0] = fmin(mn[0], (double)p.x);
    mn[1] = fmin(mn[1], (double)p.y);
    mn[2] = fmin(mn[2], (double)p.z);
    mx[0] = fmax(mx[0], (double)p.x);
    mx[1] = fmax(mx[1], (double)p.y);
    mx[2] = fmax(mx[2], (double)p.z);
  }
  __shared__ double s[kBlock / 64][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      mn[a] = fmin(mn[a], __shfl_xor(mn[a], m, 64));
      mx[a] = fmax(mx[a], __shfl_xor(mx[a], m, 64));
    }
  }
  if (lane == 0) {
    for (int a = 0; a < 3; ++a) {
      s[w][a] = mn[a];
      s[w][3 + a] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = s[0][threadIdx.x];
    for (int k = 1; k < kBlock / 64; ++k) v = threadIdx.x < 3 ? fmin(v, s[k][threadIdx.x]) : fmax(v, s[k][threadIdx.x]);
    out[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__host__ __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename P4>
__device__ __forceinline__ int cell_of(const GridDev& g, const P4& p) {
  const int ix = clampi((int)floor(((double)p.x - g.ox) * g.inv_cell), 0, g.nx - 1);
  const int iy = clampi((int)floor(((double)p.y - g.oy) * g.inv_cell), 0, g.ny - 1);
  const int iz = clampi((int)floor(((double)p.z - g.oz) * g.inv_cell), 0, g.nz - 1);
  return (iz * g.ny + iy) * g.nx + ix;
}

// A run of equal cells in consecutive lanes (clouds come in scan-line or voxel-key order: neighbours in the array are neighbours in space) is
// served by ONE atomic of its first lane.  `c` < 0 marks a lane without a point.  Returns whether this lane leads a run, the run's length
// and the lane that leads the run this lane belongs to.  All 64 lanes call it together.
__device__ __forceinline__ bool cell_run(int c, int lane, int* len, int* lead_lane) {
  const int cp = __shfl_up(c, 1, 64);
  const bool start = lane == 0 || cp != c;
  const unsigned long long starts = __ballot(start);
  const unsigned long long above = lane == 63 ? 0ull : (starts >> (lane + 1));
  *len = above ? (int)__builtin_ctzll(above) + 1 : 64 - lane;
  const unsigned long long below = starts & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  *lead_lane = 63 - __builtin_clzll(below);  // lane 0 always starts a run
  return start && c >= 0;
}

// counts[cell] += (points of the cell) ; cell_id[i] = cell
template <typename P4>
__global__ __launch_bounds__(kBlock) void cell_count_kernel(const P4* __restrict__ pts, size_t n, GridDev g, int* __restrict__ counts,
                                                            int* __restrict__ cell_id, const int* __restrict__ n_dev = nullptr /* the exact count when n is an upper bound */) {
  if (n_dev) n = (size_t)*n_dev;
  const int lane = threadIdx.x & 63;
  for (size_t i0 = (size_t)blockIdx.x * kBlock; i0 < n; i0 += (size_t)gridDim.x * kBlock) {  // whole wavefronts iterate together
    const size_t i = i0 + threadIdx.x;
    int c = -1;
    if (i < n) {
      c = cell_of(g, pts[i]);
      cell_id[i] = c;
    }
    int len, lead_lane;
    if (cell_run(c, lane, &len, &lead_lane)) atomicAdd(&counts[c], len);
  }
}

// 3-phase exclusive scan over `m` values (int, or unsigned long long for packed multi-counter scans), 1024 elements per block
// (4 per thread)
constexpr int kScanPerBlock = 1024;
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_local_kernel(const T* __restrict__ in, T* __restrict__ out, T* __restrict__ block_sums, size_t m,
                                                            T* __restrict__ pub = nullptr /* single-block scans: where out[m - 1] also goes */) {
  __shared__ T s_wave[kBlock / 64];
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 4;
  T v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (base + k < m) ? in[base + k] : (T)0;
  const T tsum = v[0] + v[1] + v[2] + v[3];
  // inclusive scan of tsum across the wave
  T x = tsum;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) s_wave[w] = x;
  __syncthreads();
  T woff = 0;
  for (int k = 0; k < w; ++k) woff += s_wave[k];
  T excl = woff + x - tsum;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < m) out[base + k] = excl;
    if (pub && base + k == m - 1) *pub = excl;
    excl += v[k];
  }
  if (threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = woff + x;
}
// single block: exclusive scan of block sums in place (nb <= 1<<20 handled serially per thread chunk)
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_sums_kernel(T* __restrict__ sums, int nb) {
  __shared__ T s_tot[kBlock];
  const int per = (nb + kBlock - 1) / kBlock;
  const int b = threadIdx.x * per, e = min(nb, b + per);
  T t = 0;
  for (int i = b; i < e; ++i) t += sums[i];
  s_tot[threadIdx.x] = t;
  __syncthreads();
  T off = 0;
  for (int k = 0; k < (int)threadIdx.x; ++k) off += s_tot[k];
  for (int i = b; i < e; ++i) {
    const T v = sums[i];
    sums[i] = off;
    off += v;
  }
}
// phases 2 and 3 in one launch for up to kScanFusedBlocks blocks: every block adds up the block sums in front of it itself (at most
// a few thousand values, read once per block and L2-resident) instead of waiting for a one-block scan of them -- a launch less per
// scan, and the per-scan pipeline of a lidar frame runs ten scans
constexpr int kScanFusedBlocks = 4096;
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_add_fused_kernel(T* __restrict__ out, const T* __restrict__ sums, size_t m, T* __restrict__ pub = nullptr) {
  __shared__ T s_part[kBlock / 64];
  T t = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) t += sums[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
  __syncthreads();
  T off = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; ++k) off += s_part[k];
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < m) {
      const T v = out[base + k] + off;
      out[base + k] = v;
      if (pub && base + k == m - 1) *pub = v;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void scan_add_kernel(T* __restrict__ out, const T* __restrict__ sums, size_t m, T* __restrict__ pub = nullptr) {
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 4;
  const T off = sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < m) {
      const T v = out[base + k] + off;
      out[base + k] = v;
      if (pub && base + k == m - 1) *pub = v;
    }
}

// sorted[pos] = pts[i] (+ normals), pos = cell_start[cell] + --counts[cell]: the counters of cell_count_kernel double as the cursors
// and are back at zero when every point is placed (the block of zeros they came from needs no clearing for the next build).  The order
// of the points inside a cell is the order the atomics are served in; nothing downstream depends on it.
template <typename P4>
__global__ __launch_bounds__(kBlock) void scatter_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, size_t n,
                                                         const int* __restrict__ cell_id, const int* __restrict__ cell_start,
                                                         int* __restrict__ counts, P4* __restrict__ spts, P4* __restrict__ snrm,
                                                         const int* __restrict__ n_dev = nullptr) {
  if (n_dev) n = (size_t)*n_dev;
  const int lane = threadIdx.x & 63;
  for (size_t i0 = (size_t)blockIdx.x * kBlock; i0 < n; i0 += (size_t)gridDim.x * kBlock) {  // whole wavefronts iterate together
    const size_t i = i0 + threadIdx.x;
    const int c = i < n ? cell_id[i] : -1;
    int len, lead_lane, old = 0;
    if (cell_run(c, lane, &len, &lead_lane)) old = atomicSub(&counts[c], len);  // the run takes the places old - len ... old - 1 of its cell
    old = __shfl(old, lead_lane, 64);
    if (i < n) {
      const int pos = cell_start[c] + old - 1 - (lane - lead_lane);
      P4 p = pts[i];
      p.i = (typename Scalar<P4>::index)i;
      spts[pos] = p;
      if (nrm) snrm[pos] = nrm[i];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// exact 1-NN within radius on the grid ([O3D] KDTreeFlann::SearchHybrid(q, r, 1)), G lanes per query
// ----------------------------------------------------------------------------------------------
// Two costs trade against each other (measured with PMC counters on MI355X): with many lanes per query the per-query
// set-up and the per-candidate bookkeeping are replicated per wavefront and the kernel becomes VALU-issue bound; with one
// lane per query the candidate loads of a lane form one long dependent chain and the kernel is latency bound.  So a
// query is served by a small group of G lanes (template parameter, 2/4/8), each lane keeps FOUR independent 16-B loads
// in flight, and the candidate test is branch-free.  Ties go to the smaller original index, so the result does not
// depend on row order, sort order, cell size or G.

template <typename P4>
struct NNBest {
  typename Scalar<P4>::type d2;
  int pos;  // position in the cell-sorted target arrays; -1 = none
  typename Scalar<P4>::index idx;
};

// squared candidate distance: with f32 storage two fused steps (this is the innermost loop of the search); with f64 storage the three
// products and two sums the reference's k-d tree forms (nanoflann's L2 adaptor, built without FMA)
__device__ __forceinline__ float dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }
__device__ __forceinline__ double dist2(double dx, double dy, double dz) { return dx * dx + dy * dy + dz * dz; }

// branch-free candidate test (the crop predicate, when compiled in, is only evaluated for would-be winners)
template <typename P4, bool kCrop>
__device__ __forceinline__ typename Scalar<P4>::type consider(const P4& t, int p, bool valid, typename Scalar<P4>::type qx,
                                                              typename Scalar<P4>::type qy, typename Scalar<P4>::type qz,
                                                              const CropDev& crop, NNBest<P4>& best) {
  using R = typename Scalar<P4>::type;
  const R dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
  const R d2 = dist2(dx, dy, dz);
  // strict d2 < best (initially r^2); ties broken towards the smaller original index
  bool better = valid & ((d2 < best.d2) | ((d2 == best.d2) & (t.i < best.idx)));
  if (kCrop) {
    if (better) better = crop_contains(crop, (double)t.x, (double)t.y, (double)t.z);
  }
  best.d2 = better ? d2 : best.d2;
  best.pos = better ? p : best.pos;
  best.idx = better ? t.i : best.idx;
  return d2;
}

// ---- candidate sets ("the neighbourhood a query's match was proven in") ---------------------------------------------------------
// A search that runs with a margin m > 0 trims every row with the ball of radius (current bound + m) and lists every scanned point
// closer than tau = (distance to the bound it STARTED from) + m in the query's LDS list.  The bound only shrinks, so when the
// search ends with the nearest neighbour at distance d1, every target point that is NOT listed is at least
//     L = min(d1 + m, cell * (k + distance to the nearest cell face))          (k = the block radius the search reached)
// away from the query position p_ref: scanned and not listed => >= tau >= d1 + m; trimmed away => beyond the ball of the bound of
// that moment + m >= d1 + m; outside the block => beyond the block.  The NEXT pass of the registration moves the query by
// delta = |p_new - p_ref| (millimetres near convergence); if the best listed point is closer to p_new than L - delta, it is the
// nearest neighbour of p_new among ALL target points (ties are impossible: everything else is strictly farther), and the pass
// needs no search at all -- the listed points and their normals are fetched BEFORE the pose is known, while the previous pass's
// 6x6 system is being solved (icp_pass_body, "verified match").  A list holds at most kSetCap points (one per lane of the query's
// group); a longer one, or a margin of zero, means "no set": that query searches again next pass, from the bound it has.
constexpr int kSetCap = 4;
template <typename R>
struct Collect {
  R tau2;     // scanned candidates with d2 < tau2 are listed; 0 = off
  int* cnt;   // LDS: listed so far (beyond kSetCap = overflow, the set is dropped)
  int* list;  // LDS: [kSetCap] positions in the cell-sorted target
};
template <typename R>
__device__ __forceinline__ void collect_push(const Collect<R>& c, int p) {
  const int k = atomicAdd(c.cnt, 1);
  if (k < kSetCap) c.list[k] = p;
}

// candidates s+lane, s+lane+stride, ... of [s,e): four loads issued before the first is consumed (two with f64 storage: a
// candidate is eight registers there, and four in flight pushed the kernel over its register budget -- spills that the compiler
// places inside divergent regions, which is not safe: a value stored under a narrow EXEC mask and reloaded under a wider one)
template <typename P4, bool kCrop, bool kCollect>
__device__ __forceinline__ void scan_strided(const P4* __restrict__ tp, int s, int e, int lane, int stride, typename Scalar<P4>::type qx,
                                             typename Scalar<P4>::type qy, typename Scalar<P4>::type qz, const CropDev& crop,
                                             NNBest<P4>& best, const Collect<typename Scalar<P4>::type>& col) {
  using R = typename Scalar<P4>::type;
  constexpr int kInFlight = sizeof(R) == 8 ? 2 : 4;
  for (int p = s + lane; p < e; p += kInFlight * stride) {
    int pk[kInFlight];
    bool vk[kInFlight];
    P4 tk[kInFlight];
    R dk[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
      pk[k] = p + k * stride;
      vk[k] = pk[k] < e;
      tk[k] = tp[vk[k] ? pk[k] : p];  // clamped, not predicated: branches around the loads measured slower
    }
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) dk[k] = consider<P4, kCrop>(tk[k], pk[k], vk[k], qx, qy, qz, crop, best);
    if (kCollect) {
      // the candidate-set list: a handful of points per query lie inside tau, so ONE test per batch of candidates (a clamped load
      // repeats candidate p, so the minimum needs no masks; the flags sort it out inside)
      R dmin = dk[0];
#pragma unroll
      for (int k = 1; k < kInFlight; ++k) dmin = min(dmin, dk[k]);
      if (dmin < col.tau2) {
#pragma unroll
        for (int k = 0; k < kInFlight; ++k)
          if (vk[k] & (dk[k] < col.tau2)) collect_push(col, pk[k]);
      }
    }
  }
}

template <typename P4, int W>
__device__ __forceinline__ void lanes_min(NNBest<P4>& b) {
#pragma unroll
  for (int m = 1; m < W; m <<= 1) {
    const auto od2 = __shfl_xor(b.d2, m, W);
    const int opos = __shfl_xor(b.pos, m, W);
    const auto oidx = __shfl_xor(b.idx, m, W);
    const bool take = (opos != -1) & ((od2 < b.d2) | ((od2 == b.d2) & (oidx < b.idx)));
    b.d2 = take ? od2 : b.d2;
    b.pos = take ? opos : b.pos;
    b.idx = take ? oidx : b.idx;
  }
}

// ---- bound-pruned search -------------------------------------------------------------------------
// The cells of one (y,z) row are ONE contiguous range of the cell-sorted target, so a "segment" (row, xa..xb) costs two
// cell_start values and is scanned with one strided loop.  What a pass costs is the number of dependent memory rounds, and
// the rounds of a wavefront are those of its slowest query group; at the bench's map density (~100 pts/m^2, own cell
// empty for 45 % of the queries) a full 3x3x3 scan is ~58 candidates in ~5 rows = 5-7 rounds, although the converged
// nearest neighbour is 5 cm away.  So every search starts from an UPPER BOUND and only touches cells inside that ball:
//   bound    the query's match of the previous pass (nn_cache, one gathered point): an ICP update moves a point by
//            millimetres near convergence, so the ball usually covers one or two cells.  Any target point is a valid
//            bound, so a stale entry can cost time but never correctness; pass 0 of a registration starts from r;
//   stage 1  the 3x3x3 block, every row trimmed to the x-extent of the ball at that row (rows out of reach vanish).
//            Proven exact if best <= cell * (1 + distance to the nearest face);
//   stage 2  (only if not proven) the 5x5x5 shell, trimmed the same way.  Proven if best <= cell * (2 + face distance);
//   stage 3  (nearest neighbour farther than two cells) the whole wavefront serves the query, nn_search_wave_far.
// All cell_start values stage 1 can need (9 rows x 4) are fetched in ONE batch together with the cached match; stage 2
// fetches its bounds in two batches.  Non-empty segments are compacted into a per-group LDS list, so a wavefront
// iterates max-over-groups(#segments) times and not over the union of the groups' rows.  Pruned cells only hold points
// strictly farther than the current best, so the result (nearest within r, ties to the smaller original index) is the
// one the full scan gives.
constexpr int kFarList = 256;  // stage 3 list, aliased onto the wavefront's groups' lists
constexpr int kSegMax = 32;   // stage 1: <= 9 segments; stage 2: <= 19 per batch of 13 rows; 32 so that a wavefront (>= 8 groups) owns >= kFarList entries

// Workgroup barrier that orders LDS only.  __syncthreads() is a release/acquire fence over GLOBAL memory as well: it waits for every
// outstanding vector-memory operation of the wavefront (s_waitcnt vmcnt(0)), i.e. for the acknowledgement of the cache-entry and
// candidate-set stores a pass has just issued and for any load still in flight -- a memory round trip per barrier on a path that has
// six of them and, with verified matches, nothing else to wait for.  Nothing in the pass kernels hands global data from one wavefront
// of a workgroup to another, so their barriers wait for the LDS queue alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void lds_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum of c over the G lanes of a group; *total = group sum
template <int G>
__device__ __forceinline__ int group_exclusive_sum(int c, int gl, int* total) {
  int incl = c;
#pragma unroll
  for (int d = 1; d < G; d <<= 1) {
    const int o = __shfl_up(incl, d, G);
    if (gl >= d) incl += o;
  }
  *total = __shfl(incl, G - 1, G);
  return incl - c;
}

struct QueryCell {
  float ux, uy, uz;  // position inside the own cell, [0,1]
  float mf;          // distance to the nearest cell face, cell units
  int ix, iy, iz;
};

// Cell coordinates come from the same f64 expression the index build uses; everything the pruning does afterwards is
// relative to the own cell, in cell units, in f32 with margins (1e-4 cells) far above the rounding of either side.
__device__ __forceinline__ QueryCell locate(const GridDev& g, double qx, double qy, double qz) {
  QueryCell c;
  const double fx = (qx - g.ox) * g.inv_cell, fy = (qy - g.oy) * g.inv_cell, fz = (qz - g.oz) * g.inv_cell;
  const double flx = floor(fx), fly = floor(fy), flz = floor(fz);
  const double lim = 1.0e9;  // queries far outside the grid cannot have a neighbour within r: clamp so the int conversion is safe
  c.ix = (int)fmin(fmax(flx, -lim), lim);
  c.iy = (int)fmin(fmax(fly, -lim), lim);
  c.iz = (int)fmin(fmax(flz, -lim), lim);
  c.ux = (float)(fx - flx);
  c.uy = (float)(fy - fly);
  c.uz = (float)(fz - flz);
  c.mf = fminf(fminf(fminf(c.ux, 1.0f - c.ux), fminf(c.uy, 1.0f - c.uy)), fminf(c.uz, 1.0f - c.uz));
  return c;
}

// distance (cell units) from the query to the slab of cells at offset d along one axis; u = position inside the own cell
__device__ __forceinline__ float slab_dist(int d, float u) { return d == 0 ? 0.0f : (d < 0 ? u + (float)(-d - 1) : (1.0f - u) + (float)(d - 1)); }

// x-extent [xa, xb] (cell offsets relative to the own cell) of the ball of squared radius b2 (cell units, already
// inflated) at a row whose squared distance is rowd2; returns false if the row is out of reach
__device__ __forceinline__ bool row_extent(float b2, float rowd2, float ux, int* xa, int* xb) {
  const float w2 = b2 - rowd2;
  if (!(w2 >= 0.0f)) return false;
  const float w = sqrtf(w2) + 1e-4f;
  *xa = (int)floorf(fmaxf(ux - w, -64.0f));
  *xb = (int)floorf(fminf(ux + w, 64.0f));
  return true;
}

// (d + m)^2 for a squared distance d2 and a margin m (metres); m = 0 leaves d2 as it is
template <typename R>
__device__ __forceinline__ float widen2(R d2, R m) {
  float v = (float)d2;
  if (m > (R)0) {
    const float d = sqrtf(v) + (float)m;
    v = d * d;
  }
  return v;
}
// squared bound in cell units, inflated so that rounding never culls a needed cell; m = the candidate-set margin (see Collect)
template <typename R>
__device__ __forceinline__ float bound_cells2(R d2, R m, const GridDev& g) {
  const float ic = (float)g.inv_cell;
  return widen2(d2, m) * ic * ic * (1.0f + 1e-4f) + 1e-6f;
}

// All G lanes of a group call this with the same query and the same starting bound `best` (any eligible target point, or {r^2, -1,
// -1}); every lane returns the same winner.  m / col: the candidate-set margin and list (m = 0, col.tau2 = 0: off); *kdone = the
// block radius (cells) the search covered.
template <typename P4, bool kCrop, int G, bool kCollect>
__device__ __forceinline__ NNBest<P4> nn_search_group(const GridDev& g, const P4* __restrict__ tp, typename Scalar<P4>::type qx,
                                                      typename Scalar<P4>::type qy, typename Scalar<P4>::type qz, int kmax,
                                                      const CropDev& crop, int gl, int2* seg /* this group's kSegMax entries */,
                                                      NNBest<P4> best, typename Scalar<P4>::type m,
                                                      const Collect<typename Scalar<P4>::type>& col, bool* resolved, int* kdone) {
  const QueryCell c = locate(g, (double)qx, (double)qy, (double)qz);
  const int* __restrict__ cs = g.cell_start;
  // ---- one batch, issued before the bound is even computed: the 4 cell_start values of each of this lane's rows of the 3x3
  // cross-section (fetching only the rows in reach, after the bound, measured slower: the ALU chain delays the loads)
  const int xlo = max(c.ix - 1, 0), xhi = min(c.ix + 1, g.nx - 1);
  constexpr int kOwn = (9 + G - 1) / G;
  int v[kOwn][4];
  bool rv[kOwn];
#pragma unroll
  for (int k = 0; k < kOwn; ++k) {
    const int r = gl + k * G;
    const int y = c.iy + (r % 3) - 1, z = c.iz + (r / 3) - 1;
    rv[k] = r < 9 && xlo <= xhi && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz;
    const int row = rv[k] ? (z * g.ny + y) * g.sx : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[k][j] = rv[k] ? cs[row + min(xlo + j, xhi + 1)] : 0;
  }
  // ---- stage 1: the 3x3x3 block, trimmed by the bound
  {
    const float b2 = bound_cells2(best.d2, m, g);
    int ss[kOwn], ee[kOwn];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kOwn; ++k) {
      const int r = gl + k * G;
      const float ddy = slab_dist((r % 3) - 1, c.uy), ddz = slab_dist((r / 3) - 1, c.uz);
      int xa, xb;
      ss[k] = ee[k] = 0;
      if (rv[k] && row_extent(b2, ddy * ddy + ddz * ddz, c.ux, &xa, &xb)) {
        xa = max(c.ix + xa, xlo);
        xb = min(c.ix + xb, xhi);
        if (xa <= xb) {
          const int ja = xa - xlo, jb = xb - xlo + 1;
          ss[k] = ja == 0 ? v[k][0] : (ja == 1 ? v[k][1] : v[k][2]);
          ee[k] = jb == 1 ? v[k][1] : (jb == 2 ? v[k][2] : v[k][3]);
        }
      }
      cnt += ee[k] > ss[k] ? 1 : 0;
    }
    int total;
    int off = group_exclusive_sum<G>(cnt, gl, &total);
#pragma unroll
    for (int k = 0; k < kOwn; ++k)
      if (ee[k] > ss[k]) seg[off++] = make_int2(ss[k], ee[k]);
    lds_wave_sync();
    for (int t = 0; t < total; ++t) {
      const int2 se = seg[t];
      scan_strided<P4, kCrop, kCollect>(tp, se.x, se.y, gl, G, qx, qy, qz, crop, best, col);
    }
    lds_wave_sync();  // the list is rewritten by stage 2
    lanes_min<P4, G>(best);
  }
  // proven exact if nothing outside the scanned block can be nearer: best <= (cell * (k + face distance))^2, tested with margin
  // (with a candidate-set margin: if the ball of best + m lies inside the block)
  const float ic2 = (float)(g.inv_cell * g.inv_cell) * (1.0f + 1e-4f);
  bool proven = kmax <= 1 || widen2(best.d2, m) * ic2 <= (1.0f + c.mf) * (1.0f + c.mf);
  *kdone = 1;
  // ---- stage 2: the 5x5x5 shell, trimmed by the bound (group-uniform branch), rows in two batches
  if (!proven) {
    constexpr int kHalf = 13, kOwn2 = (kHalf + G - 1) / G;
#pragma unroll 1
    for (int r0 = 0; r0 < 25; r0 += kHalf) {
      const float b2 = bound_cells2(best.d2, m, g);
      int s2[2 * kOwn2], e2[2 * kOwn2];
#pragma unroll
      for (int k = 0; k < kOwn2; ++k) {
        const int rl = gl + k * G, r = r0 + rl;
        const int dy = (r % 5) - 2, dz = (r / 5) - 2;
        const int y = c.iy + dy, z = c.iz + dz;
        s2[2 * k] = e2[2 * k] = s2[2 * k + 1] = e2[2 * k + 1] = 0;
        if (rl < kHalf && r < 25 && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz) {
          const float ddy = slab_dist(dy, c.uy), ddz = slab_dist(dz, c.uz);
          int xa, xb;
          if (row_extent(b2, ddy * ddy + ddz * ddz, c.ux, &xa, &xb)) {
            xa = max(max(c.ix + xa, c.ix - 2), 0);
            xb = min(min(c.ix + xb, c.ix + 2), g.nx - 1);
            const bool inner = abs(dy) <= 1 && abs(dz) <= 1;  // cells ix-1..ix+1 of these rows were stage 1
            const int row = (z * g.ny + y) * g.sx;
            const int xb1 = inner ? min(xb, c.ix - 2) : xb;
            if (xa <= xb1) {
              s2[2 * k] = cs[row + xa];
              e2[2 * k] = cs[row + xb1 + 1];
            }
            const int xa2 = max(xa, c.ix + 2);
            if (inner && xa2 <= xb) {
              s2[2 * k + 1] = cs[row + xa2];
              e2[2 * k + 1] = cs[row + xb + 1];
            }
          }
        }
      }
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < 2 * kOwn2; ++k) cnt += e2[k] > s2[k] ? 1 : 0;
      int total;
      int off = group_exclusive_sum<G>(cnt, gl, &total);
#pragma unroll
      for (int k = 0; k < 2 * kOwn2; ++k)
        if (e2[k] > s2[k]) seg[off++] = make_int2(s2[k], e2[k]);
      lds_wave_sync();
      for (int t = 0; t < total; ++t) {
        const int2 se = seg[t];
        scan_strided<P4, kCrop, kCollect>(tp, se.x, se.y, gl, G, qx, qy, qz, crop, best, col);
      }
      lds_wave_sync();
      lanes_min<P4, G>(best);
    }
    proven = kmax <= 2 || widen2(best.d2, m) * ic2 <= (2.0f + c.mf) * (2.0f + c.mf);
    *kdone = 2;
  }
  *resolved = proven;
  return best;
}

// Stage 3 -- queries whose nearest neighbour (if any) is farther than two cells: all 64 lanes of the wavefront serve ONE
// query, still on the fine grid.  The half-rows of the (2K+1)^2 cross-section (K = ceil(r / cell)) that intersect the ball
// of the current bound and were not scanned by stages 0-2 are dealt to the lanes (one cell_start pair each), compacted
// through the wavefront's LDS list (mbcnt rank), and the lanes regroup so that every listed half-row gets
// 64 / pow2(#rows) (>= 4) lanes striding over it.
template <typename P4, bool kCrop, bool kCollect>
__device__ __forceinline__ void nn_search_wave_far(const GridDev& g, const P4* __restrict__ tp, typename Scalar<P4>::type qx,
                                                   typename Scalar<P4>::type qy, typename Scalar<P4>::type qz, int K,
                                                   const CropDev& crop, NNBest<P4>& best, int lane, int2* s_list /* kFarList entries */,
                                                   typename Scalar<P4>::type m, const Collect<typename Scalar<P4>::type>& col) {
  constexpr int kdone = 2;  // cells within offset 2 were scanned by the group stages
  constexpr int kPer = 4;   // half-rows per lane and round: all their bounds are fetched in one batch
  const QueryCell c = locate(g, (double)qx, (double)qy, (double)qz);
  const float b2 = bound_cells2(best.d2, m, g);
  const int* __restrict__ cs = g.cell_start;
  const int side = 2 * K + 1, entries = 2 * side * side;
  NNBest<P4> mine = best;
  for (int base = 0; base < entries; base += 64 * kPer) {  // wave-uniform; one round for K <= 5
    int s_own[kPer], e_own[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int e = base + u * 64 + lane;
      s_own[u] = e_own[u] = 0;
      if (e < entries) {
        const int half = e & 1, rr = e >> 1;
        const int dy = rr % side - K, dz = rr / side - K;
        const int y = c.iy + dy, z = c.iz + dz;
        if ((unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz) {
          const float ddy = slab_dist(dy, c.uy), ddz = slab_dist(dz, c.uz);
          int xa, xb;
          if (row_extent(b2, ddy * ddy + ddz * ddz, c.ux, &xa, &xb)) {
            xa = max(c.ix + xa, c.ix - K);
            xb = min(c.ix + xb, c.ix + K);
            const bool inner = abs(dy) <= kdone && abs(dz) <= kdone;
            if (half == 0)
              xb = min(xb, inner ? c.ix - kdone - 1 : c.ix);
            else
              xa = max(xa, inner ? c.ix + kdone + 1 : c.ix + 1);
            xa = max(xa, 0);
            xb = min(xb, g.nx - 1);
            if (xa <= xb) {
              const int row = (z * g.ny + y) * g.sx;
              s_own[u] = cs[row + xa];
              e_own[u] = cs[row + xb + 1];
            }
          }
        }
      }
    }
    // compact the non-empty half-rows of all kPer slices into the wavefront's list
    int total = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const unsigned long long have = __ballot(e_own[u] > s_own[u]);
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(have >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)have, 0u));
      if (e_own[u] > s_own[u]) s_list[total + rank] = make_int2(s_own[u], e_own[u]);
      total += __popcll(have);
    }
    if (!total) continue;
    lds_wave_sync();
    int groups = 1;  // lanes per listed half-row: 64 / pow2ceil(min(total,16)), i.e. 64,32,16,8,4
    while (groups < total && groups < 16) groups <<= 1;
    const int W = 64 / groups;
    const int grp = lane / W, gl = lane % W;
    for (int t = grp; t < total; t += groups) {
      const int2 se = s_list[t];
      scan_strided<P4, kCrop, kCollect>(tp, se.x, se.y, gl, W, qx, qy, qz, crop, mine, col);
    }
    lds_wave_sync();  // s_list is rewritten by the next round
  }
  lanes_min<P4, 64>(mine);
  best = mine;
}

// ----------------------------------------------------------------------------------------------
// one ICP pass: transform -> 1-NN -> residual/Jacobian -> block-reduced normal equations
// ----------------------------------------------------------------------------------------------
struct QuantumTable {
  double q[kRec];
};

constexpr unsigned long long kNoKey = 0x7fffffffffffffffull;  // "no match" of the target-sharded search (see keys_mode)
struct IcpPassArgs {
  const void* src;   // P4[n_src]
  size_t first, count;
  const void* tpts;  // target points sorted by fine cell
  const void* tnrm;  // target normals, same order
  GridDev grid;      // uniform grid, cell = r/4 by default
  CropDev crop;
  double r2max;
  int kmax;          // ceil(r / cell): how many cells away a neighbour within r can be
  double q_hi[kRec]; // per-term quantum of the exact record sums (see split_exact); powers of two, from per-term bounds
  int method;        // o3ds_icp_method: which record a correspondence contributes (the GICP record is a separate instantiation)
  int* nn_cache;     // [n_src] position (in the sorted target) of each query's match in the previous pass of this registration
  int n_tgt;
  const void* snrm;  // source normals (generalized ICP only), same order as src
  double gicp_k;     // 1 - epsilon of [O3D] TransformationEstimationForGeneralizedICP (covariance = I - k n n^T)
  const IcpStateDev* state;
  double* partials;  // [gridDim.x][kRec]
  int debug;         // timing experiments only (O3DS_DEBUG_ACC): 1 = exit after prologue, 2 = no search, 3 = no winner gather
  // Target-sharded registration against ONE map split over several GPUs (SURVEY.md 8e Partitioning B): keys_mode 1 = search only,
  // this rank's best match of every query goes out as a 64-bit key (float bits of d2 << 32 | rank << 28 | position in this rank's
  // sorted target; 0x7fff...f = none within the radius, so that the keys also order as SIGNED 64-bit integers) -- non-negative float bit patterns order like the floats, so an element-wise MIN
  // over the ranks (one all-reduce) is the argmin over the union of the shards; keys_mode 2 = no search, a query contributes its
  // record iff the winning key names this rank.  0 = the ordinary pass.
  int keys_mode, keys_rank;
  unsigned long long* keys;  // [n_src]
  // Candidate sets (see Collect): per query the <= kSetCap listed positions (set_pos[4 i + k], -1 = unused; without a set entry 0 is
  // the match itself, the next search's bound) and {p_ref.xyz, L} in storage precision (set_ref[4 i + 0..3]; L = 0: no set).
  // Fused kernel only (the margin comes from the update the launch's prologue has just computed); null = off.
  int* set_pos;
  void* set_ref;
  float set_gain, set_min, set_cap;  // margin m = gain * (|R - I|_F |p| + |t|) of the last update, at least set_min; above set_cap: no set
  unsigned long long* stats;  // null, or per-launch counters [launch][4]: verified matches, searches, sets left behind, stage-3 queries (O3DS_ICP_STATS)
  // null, or the exact number of source points when `count` is only an upper bound (a source cloud whose size the host has not seen:
  // common.hpp, CountPub).  Queries are dealt out over `count` as always; the ones at or beyond the exact number are dropped where a query
  // past the end is dropped, and the fitness is taken over the exact number.
  const int* count_dev;
};

// Per-query record staged in LDS: {J0..J5, r, one, d2, 0}.  Every entry of the 32-double normal-equation record is a
// product of two slots: JtJ[a][b] = J[a]*J[b], Jtr[a] = J[a]*r, sum r^2 = r*r, count = one*one, sum d^2 = d2*one.
constexpr int kRecSlots = 10;
constexpr unsigned char kTermA[kRec] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9};
constexpr unsigned char kTermB[kRec] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7, 7, 9, 9};
// Point-to-point ([O3D] TransformationEstimationPointToPoint = Eigen::umeyama without scaling): per-query slots {p[3], q[3], 0, one,
// d2, 0}; record [0..8] = sum q_a p_b (row a, column b), [9..11] = sum p, [12..14] = sum q, [28] = count, [29] = sum d2.
constexpr unsigned char kTermA_p2p[kRec] = {3, 3, 3, 4, 4, 4, 5, 5, 5, 0, 1, 2, 3, 4, 5, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 8, 9, 9};
constexpr unsigned char kTermB_p2p[kRec] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 7, 7, 7, 7, 7, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 7, 9, 9};
// [O3D] GetInformationMatrixFromPointClouds (same slots as point-to-point): record [0..5] = sum of q q^T (xx, xy, xz, yy, yz, zz),
// [6..8] = sum q, [28] = count, [29] = sum d2; the 6x6 is assembled from these ten numbers on the host.
constexpr int kMethodInformation = 3;  // internal value of IcpPassArgs::method, not an o3ds_icp_method
constexpr unsigned char kTermA_inf[kRec] = {3, 3, 3, 4, 4, 5, 3, 4, 5, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 8, 9, 9};
constexpr unsigned char kTermB_inf[kRec] = {3, 4, 5, 4, 5, 5, 7, 7, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 7, 9, 9};

struct TermPack {
  unsigned long long lo, hi;  // terms 0..15, 16..31, four bits each
};
constexpr TermPack pack_terms(const unsigned char (&t)[kRec]) {
  TermPack p{0ull, 0ull};
  for (int k = 0; k < 16; ++k) {
    p.lo |= (unsigned long long)(t[k] & 15) << (4 * k);
    p.hi |= (unsigned long long)(t[16 + k] & 15) << (4 * k);
  }
  return p;
}
constexpr TermPack kPackA = pack_terms(kTermA), kPackB = pack_terms(kTermB), kPackA_p2p = pack_terms(kTermA_p2p),
                   kPackB_p2p = pack_terms(kTermB_p2p), kPackA_inf = pack_terms(kTermA_inf), kPackB_inf = pack_terms(kTermB_inf);
__device__ __forceinline__ int term_slot(unsigned long long lo, unsigned long long hi, int term) {  // (by value: literals, not a table in memory)
  return (int)(((term & 16) ? hi : lo) >> (4 * (term & 15))) & 15;
}

// Workgroup = BLOCK threads = BLOCK/G queries x G lanes for the search, then (32 record terms) x (BLOCK/32 query slices) for the
// accumulation: one f64 accumulator per thread instead of 30, so the kernel stays small in registers and the chip can
// keep enough wavefronts in flight to hide the dependent-load latency of the search.
__device__ __forceinline__ double to_sgpr(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// One correspondence + reduction pass of the workgroup's share of the source (batches wg, wg+nwg, ..) under the
// transformation Tm (column-major, 16 doubles, any address space).  Leaves the workgroup's 32-double partial record
// in s_red[0][0..31]... more precisely returns it in `row_val` of threads 0..31.
// [O3D] GeneralizedICP (reference call site CloudRegistration.cpp:16-21), closed form for covariances built
// from unit normals: C = Rx diag(eps,1,1) Rx^T = I - k n n^T (k = 1 - eps; n := e1 when n.x < -0.99, GetRotationFromE1ToX's
// special case), so M = Ct + R Cs R^T = 2I - k (a a^T + b b^T).  With A = [-[p]x | I] and W = M^-1/2 the three residual rows
// W d and Jacobian rows W A contribute  J^T J = A^T M^-1 A  and  J^T r = A^T M^-1 d : only M^-1 is needed (3x3 cofactors).
// Writes the 21 + 6 + 3 record values of ONE correspondence.
__device__ __forceinline__ void gicp_record(double px, double py, double pz, double dx, double dy, double dz, const double a[3],
                                            const double b[3], double k, double* rec) {
  const double m00 = 2.0 - k * (a[0] * a[0] + b[0] * b[0]), m01 = -k * (a[0] * a[1] + b[0] * b[1]), m02 = -k * (a[0] * a[2] + b[0] * b[2]);
  const double m11 = 2.0 - k * (a[1] * a[1] + b[1] * b[1]), m12 = -k * (a[1] * a[2] + b[1] * b[2]), m22 = 2.0 - k * (a[2] * a[2] + b[2] * b[2]);
  const double c00 = m11 * m22 - m12 * m12, c01 = m02 * m12 - m01 * m22, c02 = m01 * m12 - m02 * m11;
  const double c11 = m00 * m22 - m02 * m02, c12 = m01 * m02 - m00 * m12, c22 = m00 * m11 - m01 * m01;
  const double idet = 1.0 / (m00 * c00 + m01 * c01 + m02 * c02);
  const double B[3][3] = {{c00 * idet, c01 * idet, c02 * idet}, {c01 * idet, c11 * idet, c12 * idet}, {c02 * idet, c12 * idet, c22 * idet}};
  const double S[3][3] = {{0.0, -pz, py}, {pz, 0.0, -px}, {-py, px, 0.0}};  // S[c] = column c of -[p]x
  double BS[3][3];  // BS[c] = B * S[c]
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) BS[c][r] = B[r][0] * S[c][0] + B[r][1] * S[c][1] + B[r][2] * S[c][2];
  const double Bd[3] = {B[0][0] * dx + B[0][1] * dy + B[0][2] * dz, B[1][0] * dx + B[1][1] * dy + B[1][2] * dz,
                        B[2][0] * dx + B[2][1] * dy + B[2][2] * dz};
  int t = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = r; c < 3; ++c) rec[t++] = S[r][0] * BS[c][0] + S[r][1] * BS[c][1] + S[r][2] * BS[c][2];  // S^T B S
#pragma unroll
    for (int c = 0; c < 3; ++c) rec[t++] = BS[r][c];  // S^T B  (row r, col 3+c) = (B S[r])[c]
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = r; c < 3; ++c) rec[t++] = B[r][c];
#pragma unroll
  for (int r = 0; r < 3; ++r) rec[21 + r] = S[r][0] * Bd[0] + S[r][1] * Bd[1] + S[r][2] * Bd[2];
#pragma unroll
  for (int r = 0; r < 3; ++r) rec[24 + r] = Bd[r];
  rec[kRecR2] = dx * Bd[0] + dy * Bd[1] + dz * Bd[2];
  rec[kRecCount] = 1.0;
  rec[kRecD2] = dx * dx + dy * dy + dz * dz;
  rec[30] = 0.0;
  rec[31] = 0.0;
}

// Which query slot ql of batch b serves.  Passes that have a cached match per query take CONSECUTIVE queries (neighbours on a
// scan ring share cells and cache lines).  Pass 0 of a registration has no bound, so queries whose neighbour is far are
// expensive and they cluster (the misalignment grows with range: measured mean 21 us per workgroup, slowest 85 us); it
// takes queries count/16 apart, which spreads every region of the scan over all wavefronts.
// order: 0 = scan order, 1 = single queries spread over the scan, 2 = every wavefront ONE run of kQPW consecutive queries, the runs scattered
// over the scan (pass_order below)
template <int kQPB, int kQPW /* queries per wavefront */>
__device__ __forceinline__ size_t query_index(size_t count, size_t b, int ql, int order) {
  if (order == 0) return b * kQPB + (size_t)ql;
  constexpr unsigned long long kMul = 1000003ull;
  const size_t off = b * (kQPB / kQPW) + (size_t)(ql / kQPW);  // this wavefront's number among all wavefronts of the pass
  const int qw = ql % kQPW;
  if (order == 2) {
    const size_t n_run = (count + kQPW - 1) / kQPW;
    const size_t run = n_run % kMul ? (size_t)(((unsigned long long)off * kMul) % n_run) : off;
    const size_t i = run * kQPW + (size_t)qw;
    return off < n_run && i < count ? i : count;
  }
  // a wavefront takes kQPW / kRun runs of kRun consecutive queries, the runs count/(kQPW/kRun) apart (a wavefront serves its
  // far queries one after the other, so the mix has to hold per wavefront, not just per workgroup; runs keep some of the
  // cache-line sharing of neighbouring queries) ...
  constexpr int kRun = 1, kRuns = kQPW / kRun;  // runs of 4: body mean 37 -> 32 us but slowest workgroup 55 -> 58 us; runs of 16 (a whole wavefront): group search 10 us but slowest workgroup 66 us, 38.9k it/s
  static_assert(kQPW % kRun == 0, "runs tile a wavefront");
  const size_t n_run = (count + kRun - 1) / kRun;         // runs in the scan
  const size_t n_off = (n_run + kRuns - 1) / kRuns;       // offsets: one per wavefront slot
  // ... and the offsets are scattered by a multiplicative bijection (prime multiplier), so that the wavefronts of one
  // workgroup do not all sample the same azimuth sector of the scan
  const size_t offs = n_off % kMul ? (size_t)(((unsigned long long)off * kMul) % n_off) : off;
  const size_t run = (size_t)(qw / kRun) * n_off + offs;
  const size_t i = run * kRun + (size_t)(qw % kRun);
  return off < n_off && run < n_run && i < count ? i : count;
}

// How pass number `pass` of a registration deals its queries out.  Any order gives the same matches; what it changes is time.
//   pass 0   has no bound: its far queries are expensive and cluster (the misalignment grows with range), so single queries are spread over
//            all wavefronts;
//   pass 1   follows the first, largest update.  It has a cached match per query and profits from neighbours sharing cells and cache lines
//            -- and after an update that did not land (generalized ICP on configs[1]: 3 550 queries are still more than two cells from their
//            neighbour) those queries are ONE region of the scan: in scan order they fill a few dozen workgroups of 64, each sixteen rounds
//            of stage 3, while the chip idles (97 us for a pass of 2 395 VALU instructions per wavefront, against 64 us for pass 0's 4 366:
//            profiles/r06_pmc_gicp_per_pass.txt).  Runs of one wavefront keep the sharing inside the wavefront and give the four
//            wavefronts of a workgroup four places of the scan: the workgroup's far pool holds a quarter of such a region (68 us), and a
//            pass 1 without far queries -- point-to-plane's -- costs what it did (22.6 us);
//   later    scan order.
// A function of the pass number alone, so every form of the loop and every rank deals alike, and a workgroup knows its queries before it
// has read anything: the fetch of their points goes out first (a rule on the previous pass's far count, carried as a record term, gave
// the same 68 us and cost every pass the wait for that term: -1 % on configs[1]).
__host__ __device__ __forceinline__ int pass_order(int pass) { return pass == 0 ? 1 : (pass == 1 ? 2 : 0); }

// What a query's search needs that does NOT depend on the pose: its source point, the position of its match in the previous
// pass and that matched target point.  The fused kernel issues these loads for the workgroup's first batch BEFORE it waits
// for the previous pass's records and runs the solve, so two of the body's dependent memory rounds overlap the serial tail.
template <typename P4>
struct QueryPrefetch {
  P4 s, tprev;
  int prev;
  // candidate-set mode: THIS LANE's listed point (prev / tprev above) and its normal, and the set's reference {p_ref, L}
  P4 nprev;
  typename Scalar<P4>::type rx, ry, rz, rL;
};

template <typename P4>
struct alignas(4 * sizeof(typename Scalar<P4>::type)) SetRef {
  typename Scalar<P4>::type x, y, z, L;
};

// The number a pass deals its queries out over: the exact size of the source -- from the device word where the host has only an upper bound
// (IcpPassArgs::count_dev), NOT that bound.  Which workgroup sums which queries decides the last place of the sums (one rounded f64
// reduction per workgroup), and `count` is the bound or the exact number depending on whether the size had reached the host when the
// registration was queued: dealt out over `count`, a stream's poses wobbled by an ulp with the timing of its host threads (round 6,
// scripts/debug_wobble.py; tests/test_icp_gpu.py::test_a_registration_does_not_depend_on_when_the_host_learnt_the_size_of_its_scan).
// The launch is still sized by `count`: the workgroups beyond the exact number find no query.
__device__ __forceinline__ size_t deal_count(const IcpPassArgs& a) { return a.count_dev ? min((size_t)*a.count_dev, a.count) : a.count; }

template <typename P4, int kPassBlock, int kGroup>
__device__ __forceinline__ QueryPrefetch<P4> prefetch_query(const IcpPassArgs& a, size_t batch, bool use_cache, bool use_sets, int order, size_t n_deal) {
  constexpr int kQPB = kPassBlock / kGroup;
  using R = typename Scalar<P4>::type;
  QueryPrefetch<P4> q;
  q.prev = -1;
  q.s = P4{};
  q.nprev = P4{};
  q.tprev = P4{};
  q.rx = q.ry = q.rz = q.rL = (R)0;
  const bool sets = use_cache && use_sets && a.set_pos != nullptr;  // (use_sets: the fused kernel only, see icp_pass_body)
  const size_t i = query_index<kQPB, 64 / kGroup>(n_deal, batch, threadIdx.x / kGroup, order);
  if (a.count == 0) return q;
  // Straight-line loads, no branch per lane and none per mode (a lane past the end reads the last query and drops it; a mode that has no
  // use for a load reads a harmless address): a load inside a conditional block is followed by the moves that merge it with the other
  // arm's value, i.e. by a wait for it right where it was issued -- three memory round trips one after the other in front of the solve
  // instead of two overlapped ones.
  static_assert(sizeof(SetRef<P4>) == sizeof(P4), "a source point stands in for the reference a mode without sets does not read");
  static_assert(kGroup == kSetCap, "one listed point per lane of the group");
  const bool live = i < a.count;
  const size_t ic = a.first + (live ? i : a.count - 1);
  const P4* __restrict__ src = (const P4*)a.src;
  const int* pprev = sets ? a.set_pos + ic * kSetCap + (threadIdx.x & (kGroup - 1)) : a.nn_cache + ic;
  const SetRef<P4>* pref = sets ? (const SetRef<P4>*)a.set_ref + ic : (const SetRef<P4>*)(src + ic);
  const P4 s = src[ic];
  int prev = *pprev;
  const SetRef<P4> r = *pref;
  if (prev >= a.n_tgt || !live || !use_cache) prev = -1;  // never trust the cache with an address
  const int pc = max(prev, 0);
  const P4 t = ((const P4*)a.tpts)[pc];
  const P4 nrm = ((const P4*)(sets && a.tnrm ? a.tnrm : a.tpts))[pc];
  q.s = s;  // (a dropped lane's point is never looked at: every use is behind i < count)
  q.prev = prev;
  q.tprev = t;
  q.nprev = nrm;
  q.rx = r.x, q.ry = r.y, q.rz = r.z, q.rL = sets && live ? r.L : (R)0;
  return q;
}

// lane 0 takes the next ticket of an LDS counter; the result is a scalar (SGPR) value in every lane
__device__ __forceinline__ int wave_pop(int* counter, int lane) {
  int k = 0;
  if (lane == 0) k = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(k);
}

// an unresolved query parked for stage 3
template <typename P4>
struct FarItem {
  typename Scalar<P4>::type x, y, z, d2, m, tau2;
  typename Scalar<P4>::index idx;
  int pos;
};

// What the prologue of a fused launch knows about the update it has just applied (the candidate-set margin is derived from it per
// query); unit = off.
struct SetMargin {
  float w, t;  // |R - I|_F and |t| of the last update U; w < 0: no sets this pass
};

// The record of ONE correspondence (query at p, matched target point q with normal nq) into its 10 (or, generalized ICP, 32) LDS slots.
template <typename P4, bool kGicp>
__device__ __forceinline__ void write_record(const IcpPassArgs& a, double* rec, bool p2p, double px, double py, double pz, const P4& q,
                                             const P4& nq, size_t i, double t00, double t01, double t02, double t10, double t11,
                                             double t12, double t20, double t21, double t22) {
  const double dx = px - (double)q.x, dy = py - (double)q.y, dz = pz - (double)q.z;
  if (!kGicp && p2p) {
    rec[0] = px;
    rec[1] = py;
    rec[2] = pz;
    rec[3] = (double)q.x;
    rec[4] = (double)q.y;
    rec[5] = (double)q.z;
    rec[6] = 0.0;
    rec[7] = 1.0;
    rec[8] = dx * dx + dy * dy + dz * dz;
    rec[9] = 0.0;
    return;
  }
  const double nx = (double)nq.x, ny = (double)nq.y, nz = (double)nq.z;
  if (kGicp) {
    const P4 ns = ((const P4*)a.snrm)[a.first + i];
    double sa[3] = {(double)ns.x, (double)ns.y, (double)ns.z};
    if (sa[0] < -0.99) sa[0] = 1.0, sa[1] = 0.0, sa[2] = 0.0;  // GetRotationFromE1ToX special case (decided on the stored normal)
    const double av[3] = {t00 * sa[0] + t01 * sa[1] + t02 * sa[2], t10 * sa[0] + t11 * sa[1] + t12 * sa[2],
                          t20 * sa[0] + t21 * sa[1] + t22 * sa[2]};  // covariance rotates with the cloud
    double bv[3] = {nx, ny, nz};
    if (bv[0] < -0.99) bv[0] = 1.0, bv[1] = 0.0, bv[2] = 0.0;
    gicp_record(px, py, pz, dx, dy, dz, av, bv, a.gicp_k, rec);
  } else {
    rec[0] = py * nz - pz * ny;  // J = [p x n ; n]
    rec[1] = pz * nx - px * nz;
    rec[2] = px * ny - py * nx;
    rec[3] = nx;
    rec[4] = ny;
    rec[5] = nz;
    rec[6] = dx * nx + dy * ny + dz * nz;  // r = (p - q) . n
    rec[7] = 1.0;
    rec[8] = dx * dx + dy * dy + dz * dz;
    rec[9] = 0.0;
  }
}

template <typename P4, bool kCrop, int kPassBlock, int kGroup, bool kGicp, bool kKeys = false /* the keys_mode code (classic kernel only) */,
          bool kCollect = false /* the searches of this pass list candidate sets (see Collect) */,
          bool kSingle = false /* ONE batch per workgroup (wg < number of batches): without the batch loop the compiler neither hoists a
                                  dozen per-lane constants out of it nor keeps the first batch's prefetch alive through it -- 112
                                  registers instead of 128 + spills */>
__device__ __forceinline__ double icp_pass_body(const IcpPassArgs& a, const double* Tm, int wg, int nwg, double* s_rec_flat,
                                                double (*s_red)[kRec], int2* s_seg /* [kPassBlock / kGroup][kSegMax] */,
                                                bool use_cache, const QueryPrefetch<P4>& first_batch,
                                                unsigned long long* tr = nullptr /* 3 timestamps, development aid */,
                                                SetMargin sm = SetMargin{-1.0f, 0.0f}, int* s_set = nullptr /* [kQPB][1 + kSetCap] */,
                                                size_t n_live = ~(size_t)0 /* queries at or beyond it are dropped (IcpPassArgs::count_dev) */,
                                                int order = 0 /* how the queries are dealt out (pass_order) */) {
  constexpr int kQPB = kPassBlock / kGroup;
  n_live = min(n_live, a.count);
  constexpr int kStride = kGicp ? kRec : kRecSlots;  // doubles per query record
  static_assert((64 / kGroup) * kSegMax >= kFarList, "a wavefront's share of s_seg holds the stage-3 list");
  using R = typename Scalar<P4>::type;
  const P4* __restrict__ tp = (const P4*)a.tpts;  // (the source points arrive through the prefetch)
  const P4* __restrict__ tn = (const P4*)a.tnrm;
  // the pose is wave-uniform: keep it in scalar registers (it arrives through LDS in the persistent kernel)
  const double t00 = to_sgpr(Tm[0]), t10 = to_sgpr(Tm[1]), t20 = to_sgpr(Tm[2]), t01 = to_sgpr(Tm[4]), t11 = to_sgpr(Tm[5]),
               t21 = to_sgpr(Tm[6]), t02 = to_sgpr(Tm[8]), t12 = to_sgpr(Tm[9]), t22 = to_sgpr(Tm[10]), t03 = to_sgpr(Tm[12]),
               t13 = to_sgpr(Tm[13]), t23 = to_sgpr(Tm[14]);
  const int gl = threadIdx.x & (kGroup - 1), ql = threadIdx.x / kGroup;
  const int term = threadIdx.x & 31, qs = threadIdx.x >> 5;
  const bool inf = a.method == kMethodInformation;
  const bool p2p = a.method == O3DS_ICP_POINT_TO_POINT || inf;  // uniform: records built from the points themselves, no normals
  // which two slots a record term multiplies: from the tables above, packed four bits per term into literals (a table in memory would
  // be a load whose latency the verified-match path has nothing to hide behind)
  const int ta = term_slot(inf ? kPackA_inf.lo : (p2p ? kPackA_p2p.lo : kPackA.lo), inf ? kPackA_inf.hi : (p2p ? kPackA_p2p.hi : kPackA.hi), term);
  const int tb = term_slot(inf ? kPackB_inf.lo : (p2p ? kPackB_p2p.lo : kPackB.lo), inf ? kPackB_inf.hi : (p2p ? kPackB_p2p.hi : kPackB.hi), term);
  // candidate sets: read (verified matches) whenever the previous pass left them, written whenever this pass has a margin
  const bool sets = !kKeys && a.set_pos != nullptr && s_set != nullptr;
  const bool sets_in = sets && use_cache;
  const bool sets_out = kCollect && sets;
  const R rmax = (R)sqrt(a.r2max);
  int* my_set = sets ? s_set + ql * (1 + kSetCap) : nullptr;
  double acc = 0.0;
  const size_t n_batches = (n_live + kQPB - 1) / kQPB;  // (n_live is deal_count: see there)
  static_assert(sizeof(FarItem<P4>) <= kStride * sizeof(double), "a parked far query fits its record slot");
  static_assert((2 + kQPB) * sizeof(int) <= (kPassBlock / 32) * kRec * sizeof(double), "the far list fits s_red");
  int* s_far = (int*)&s_red[0][0];  // [0] count, [1] next, [2..] query slots; s_red itself is only used after the loop
  if (threadIdx.x == 0) s_far[0] = s_far[1] = 0;
  lds_barrier();
  for (size_t b = (size_t)wg; b < n_batches; b += kSingle ? n_batches : (size_t)nwg) {
    const size_t i = query_index<kQPB, 64 / kGroup>(n_live, b, ql, order);
    double px = 0, py = 0, pz = 0;
    NNBest<P4> nn;
    nn.pos = -1;
    nn.idx = -1;
    nn.d2 = (R)0;
    bool unresolved = false;
    bool verified = false;   // the match was proven inside the prefetched candidate set: the winner's lane holds point and normal
    int kdone = 0;           // block radius (cells) the search of this query covered; 0 = no search ran
    R m = (R)0;  // candidate-set margin of this query's search (0: the search leaves no set)
    const QueryPrefetch<P4> qp = (kSingle || b == (size_t)wg) ? first_batch : prefetch_query<P4, kPassBlock, kGroup>(a, b, use_cache, sets, order, n_live);
    if (sets && gl == 0) my_set[0] = 0;  // (same wavefront as its readers and writers below; the far stage is behind a barrier)
    if (i < n_live) {  // uniform across the lanes of a group
      const P4 s = qp.s;
      // [O3D] PointCloud::Transform: rigid 4x4 (bottom row 0 0 0 1 for every pose the reference passes)
      px = t00 * (double)s.x + t01 * (double)s.y + t02 * (double)s.z + t03;
      py = t10 * (double)s.x + t11 * (double)s.y + t12 * (double)s.z + t13;
      pz = t20 * (double)s.x + t21 * (double)s.y + t22 * (double)s.z + t23;
      const R qx = (R)px, qy = (R)py, qz = (R)pz;
      bool resolved = true;
      if (a.debug == 2) {
        nn.pos = (int)(i % 1000);
        nn.idx = nn.pos;
      } else if (kKeys && a.keys_mode == 2) {  // the match was decided by the all-reduce: mine iff the key names this rank
        const unsigned long long key = a.keys[a.first + i];
        if (key != kNoKey && (int)((key >> 28) & 0xfu) == a.keys_rank) {
          nn.pos = (int)(key & 0x0fffffffu);
          nn.idx = nn.pos;
          nn.d2 = (R)__uint_as_float((unsigned int)(key >> 32));
        }
      } else {
        // the starting bound: the cached match (every lane the same point) or, with candidate sets, the best of the listed points
        // (every lane its own; a set without a list keeps the match in entry 0)
        NNBest<P4> best;
        best.d2 = (R)a.r2max;
        best.pos = -1;
        best.idx = -1;
        consider<P4, kCrop>(qp.tprev, qp.prev, qp.prev >= 0, qx, qy, qz, a.crop, best);
        if (sets_in) {
          lanes_min<P4, kGroup>(best);
          // verified match: everything that is not listed is at least L away from p_ref, hence at least L - |p - p_ref| from p
          const R ex = qx - qp.rx, ey = qy - qp.ry, ez = qz - qp.rz;
          const R delta = (R)sqrt(ex * ex + ey * ey + ez * ez);
          const R dcmp = best.pos != -1 ? (R)sqrt(best.d2) : rmax;
          verified = (qp.rL - delta) * (R)0.99998 > dcmp;  // (false for L = 0 = no set, and for NaN)
          if (a.debug == 64) verified = false;
        }
        if (verified) {
          // the lane that holds the winner writes the record now (point and normal are in its registers); no match: lane 0, zeros
          nn = best;
          if (nn.pos != -1 ? qp.prev == nn.pos : gl == 0) {
            a.nn_cache[a.first + i] = nn.pos;
            double* rec = s_rec_flat + ql * kStride;
            if (nn.pos != -1) {
              // (the normal is first looked at HERE: left to itself the compiler widens it to f64 right behind its load, i.e. waits
              // for that load in the prologue, in front of the solve)
              P4 nprev = qp.nprev;
              asm volatile("" : "+v"(nprev.x), "+v"(nprev.y), "+v"(nprev.z));
              write_record<P4, kGicp>(a, rec, p2p, px, py, pz, qp.tprev, nprev, i, t00, t01, t02, t10, t11, t12, t20, t21, t22);
            } else {
#pragma unroll
              for (int k = 0; k < kStride; ++k) rec[k] = 0.0;
            }
          }
        } else {
          R tau2 = (R)0;
          if (sets_out) {  // margin of this query's search from the update just applied: the next update is smaller
            const float pn = sqrtf((float)(px * px + py * py + pz * pz));
            const float mm = a.set_gain * (sm.w * pn + sm.t);
            if (mm <= a.set_cap) {
              m = (R)fmaxf(mm, a.set_min);
              const R tau = (R)sqrt(best.d2) + m;
              tau2 = tau * tau;
            }
          }
          Collect<R> col;
          col.tau2 = tau2;
          col.cnt = my_set;
          col.list = my_set + 1;
          // (the lane's row offsets are recomputed per batch: hoisted out of the batch loop they cost a dozen registers, i.e. spills)
          int gl_b = gl;
          asm volatile("" : "+v"(gl_b));
          nn = nn_search_group<P4, kCrop, kGroup, kCollect>(a.grid, tp, qx, qy, qz, a.kmax, a.crop, gl_b, s_seg + ql * kSegMax, best, m, col, &resolved, &kdone);
          if (!resolved && gl == 0) {  // park the query for stage 3 (its record slot is still unused)
            FarItem<P4>* mine_item = (FarItem<P4>*)(s_rec_flat + ql * kStride);
            mine_item->x = qx;
            mine_item->y = qy;
            mine_item->z = qz;
            mine_item->d2 = nn.d2;
            mine_item->m = m;
            mine_item->tau2 = tau2;
            mine_item->idx = nn.idx;
            mine_item->pos = nn.pos;
          }
        }
      }
      unresolved = !resolved;
    }
    if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
    // Stage 3.  A far query takes a whole wavefront for a few memory rounds, and far queries cluster, so they are pooled per
    // WORKGROUP: every unresolved query parks its state in its (still unused) record slot and enters a list; the four
    // wavefronts then pop queries from the list until it is empty and write the winner back for the owner.  The search
    // result does not depend on who computes it, and the records stay in their slots, so the sums are unchanged.
    {
      FarItem<P4>* mine_item = (FarItem<P4>*)(s_rec_flat + ql * kStride);
      if (a.debug == 16) unresolved = false;
      if (unresolved && gl == 0) {
        const int k = atomicAdd(&s_far[0], 1);
        s_far[2 + k] = ql;
      }
      lds_barrier();
      const int n_far = __builtin_amdgcn_readfirstlane(s_far[0]);
      if (n_far > 0) {  // workgroup-uniform
        const int lane = threadIdx.x & 63;
        int2* list = s_seg + (threadIdx.x >> 6) * (64 / kGroup) * kSegMax;
        for (int k = wave_pop(&s_far[1], lane); k < n_far; k = wave_pop(&s_far[1], lane)) {  // k is scalar: a uniform loop
          const int slot = s_far[2 + k];
          FarItem<P4>* it = (FarItem<P4>*)(s_rec_flat + slot * kStride);
          NNBest<P4> bq;
          bq.d2 = it->d2;
          bq.pos = it->pos;
          bq.idx = it->idx;
          Collect<R> col;
          col.tau2 = it->tau2;
          col.cnt = sets ? s_set + slot * (1 + kSetCap) : nullptr;
          col.list = col.cnt + 1;
          if (a.debug != 32) nn_search_wave_far<P4, kCrop, kCollect>(a.grid, tp, it->x, it->y, it->z, a.kmax, a.crop, bq, lane, list, it->m, col);
          // every lane holds the same winner and stores it (same address, same value): no lane-0 branch inside this loop --
          // with one, the structurised code re-ran the body for the other lanes forever (seen on ROCm 7.2)
          it->d2 = bq.d2;
          it->pos = bq.pos;
          it->idx = bq.idx;
        }
        lds_barrier();
        if (unresolved) {  // (all lanes of the group: the set below is written by all of them)
          nn.d2 = mine_item->d2;
          nn.pos = mine_item->pos;
          nn.idx = mine_item->idx;
          kdone = a.kmax;
        }
      }
    }
    if (tr && threadIdx.x == 0) tr[1] = wall_clock64();
    if (a.stats) {  // development aid: how the queries of this launch were served
      const bool q0 = gl == 0 && i < n_live;
      const unsigned long long nv = __popcll(__ballot(q0 && verified)), ns = __popcll(__ballot(q0 && !verified)),
                               nk = __popcll(__ballot(q0 && !verified && m > (R)0 && sets && s_set[ql * (1 + kSetCap)] <= kSetCap)),
                               nf = __popcll(__ballot(q0 && unresolved));
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(a.stats + 0, nv);
        atomicAdd(a.stats + 1, ns);
        atomicAdd(a.stats + 2, nk);
        atomicAdd(a.stats + 3, nf);
      }
    }
    // ---- the candidate set this query's search leaves for the next pass (a verified match keeps the one it has)
    if (sets && i < n_live && !verified && a.debug != 2) {
      const int n_listed = my_set[0];
      const bool have = m > (R)0 && n_listed <= kSetCap;
      int pos_out = have ? (gl < n_listed ? my_set[1 + gl] : -1) : (gl == 0 ? nn.pos : -1);
      a.set_pos[(a.first + i) * kSetCap + gl] = pos_out;
      if (gl == 0) {
        const R qx = (R)px, qy = (R)py, qz = (R)pz;
        SetRef<P4> r;
        r.x = qx, r.y = qy, r.z = qz;
        r.L = (R)0;
        if (have) {
          // L = min(d1 + m, block reach): see Collect.  The block reach is cell * (k + distance to the nearest face), from the
          // same f32 cell-relative quantities the search's own "proven" tests use, shrunk by their margin
          const QueryCell c = locate(a.grid, (double)qx, (double)qy, (double)qz);
          const R d1 = nn.pos != -1 ? (R)sqrt(nn.d2) : rmax;
          const R reach = (R)(a.grid.cell * (double)(((float)kdone + c.mf) * (1.0f - 2e-4f)));
          r.L = (R)fmin((double)(d1 + m), (double)reach);
        }
        ((SetRef<P4>*)a.set_ref)[a.first + i] = r;
      }
    }
    // ---- records
    if (!verified && gl == 0) {
      {
        if (i < n_live && !(kKeys && a.keys_mode == 2)) a.nn_cache[a.first + i] = nn.pos;
        if (kKeys && a.keys_mode == 1 && i < n_live)
          a.keys[a.first + i] = nn.pos == -1 ? kNoKey
                                             : ((unsigned long long)__float_as_uint((float)nn.d2) << 32) |
                                                   ((unsigned long long)(a.keys_rank & 0xf) << 28) | (unsigned long long)(nn.pos & 0x0fffffff);
        double* rec = s_rec_flat + ql * kStride;
        if (nn.pos != -1 && !(kKeys && a.keys_mode == 1)) {
          if (a.debug == 3) nn.pos = (int)(i % 1000);
          const P4 q = tp[nn.pos];
          const P4 nq = (!kGicp && p2p) ? P4{} : tn[nn.pos];
          write_record<P4, kGicp>(a, rec, p2p, px, py, pz, q, nq, i, t00, t01, t02, t10, t11, t12, t20, t21, t22);
        } else {
#pragma unroll
          for (int k = 0; k < kStride; ++k) rec[k] = 0.0;
        }
      }
    }
    lds_barrier();
    if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
    if (threadIdx.x == 0) s_far[0] = s_far[1] = 0;  // everyone is past the far list (ordered before its next use by the barrier below)
#pragma unroll
    for (int qq = 0; qq < kQPB / (kPassBlock / 32); ++qq) {
      const double* rec = s_rec_flat + (qs * (kQPB / (kPassBlock / 32)) + qq) * kStride;
      if (kGicp)
        acc += rec[term];  // the record already holds the 32 terms of this correspondence
      else
        acc = fma(rec[ta], rec[tb], acc);
    }
    lds_barrier();  // s_rec is rewritten by the next batch
  }
  // query slices -> 1, fixed order => bitwise reproducible for a given launch geometry
  s_red[qs][term] = acc;
  lds_barrier();
  double v = 0.0;
  if (threadIdx.x < kRec) {
#pragma unroll
    for (int k = 0; k < kPassBlock / 32; ++k) v += s_red[k][threadIdx.x];
    if (threadIdx.x >= 30) v = 0.0;
  }
  lds_barrier();  // s_red may be reused by the caller
  return v;         // valid in threads 0..31
}

// Workgroup = BLOCK threads = BLOCK/G queries x G lanes for the search, then (32 record terms) x (BLOCK/32 query slices) for the
// accumulation: one f64 accumulator per thread instead of 30, so the kernel stays small in registers and the chip can
// keep enough wavefronts in flight to hide the dependent-load latency of the search.
template <typename P4, bool kCrop, int kPassBlock, int kGroup, bool kGicp, bool kKeys = false>
__global__ __launch_bounds__(kPassBlock) __attribute__((amdgpu_waves_per_eu(4))) void icp_accumulate_kernel(IcpPassArgs a) {
  constexpr int kQPB = kPassBlock / kGroup;
  if (a.state->done) return;  // device-side loop already terminated: keep the previous partials
  __shared__ double s_rec[kQPB * (kGicp ? kRec : kRecSlots)];
  __shared__ double s_red[kPassBlock / 32][kRec];
  __shared__ int2 s_seg[kQPB * kSegMax];
  if (a.debug == 1) {
    if (threadIdx.x < kRec) a.partials[(size_t)blockIdx.x * kRec + threadIdx.x] = a.state->T[0] * 1e-300;
    return;
  }
  const bool use_cache = a.state->pass > 0;
  const int order = pass_order(a.state->pass);
  const size_t n_live = deal_count(a);
  const QueryPrefetch<P4> qp = prefetch_query<P4, kPassBlock, kGroup>(a, blockIdx.x, use_cache, false, order, n_live);
  const double v = icp_pass_body<P4, kCrop, kPassBlock, kGroup, kGicp, kKeys>(a, a.state->T, blockIdx.x, gridDim.x, s_rec, s_red, s_seg, use_cache, qp, nullptr,
                                                                                SetMargin{-1.0f, 0.0f}, nullptr, n_live, order);
  if (threadIdx.x < kRec) a.partials[(size_t)blockIdx.x * kRec + threadIdx.x] = v;
}

// ----------------------------------------------------------------------------------------------
// per-pass serial tail: sum the partial records, convergence test, 6x6 solve, T <- U*T   (one workgroup)
// ----------------------------------------------------------------------------------------------
constexpr int kUpdBlock = 1024;

// ---- order-independent record sums ---------------------------------------------------------------------------------------
// Every workgroup record value v is split into hi + lo, hi a multiple of q_hi and lo a multiple of q_lo = q_hi * 2^-41, with
// q_hi a power of two chosen PER RECORD TERM from a bound B_k on the sum of |v| for that term (2^53 q_hi >= 8 B_k).  Sums of at
// most 4096 such hi (resp. lo) values are exactly representable, so f64 additions of them are EXACT and therefore associative: the
// records can be accumulated with hardware f64 atomics in whatever order the workgroups finish, and the fused, two-launch and
// step-wise forms agree bit for bit for any launch geometry.  What is dropped is below q_lo / 2 per record, i.e. ~B_k * 2^-93
// RELATIVE TO THAT TERM'S OWN BOUND -- which is why the bound has to be per term: the 32 terms of one record span ten orders of
// magnitude when the clouds are far from the origin (rot-rot ~ n |p|^2, translational J^T r ~ n r), and one global quantum cost
// 4.4e-4 m of pose accuracy at |p| = 2e5 m where plain f64 sums give 1.7e-5 (tests/test_icp_gpu.py::test_large_coordinates_*).
// If a caller's data exceed a bound (e.g. normals far from unit length) the split degrades to hi = v, lo = 0 -- ordinary f64 sums,
// still correct, merely no longer order-independent.
__device__ __forceinline__ void split_exact(double v, double q_hi, double* hi, double* lo) {
  const double c_hi = 6755399441055744.0 * q_hi;  // 1.5 * 2^52 * q_hi: (v + c) - c rounds v to a multiple of q_hi (|v| < 2^51 q_hi)
  const double h = (v + c_hi) - c_hi;
  const double r = v - h;  // exact
  const double c_lo = c_hi * 4.547473508864641e-13;  // * 2^-41
  *hi = h;
  *lo = (r + c_lo) - c_lo;
}

// sum the per-block partial records into one 32-double record (order-independent, see above); result in s_out[0..31].
// The rows were written by other CUs/XCDs and come from memory, so each dependent load round costs ~1 us: all (<= 32) loads
// of a thread are issued before the first add (measured: an add-per-load loop took 10 us, 8-deep batches 8 us).
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int nrows, const double* q_hi /* [32] */, double* s_part /* [2][32][32] */,
                                                double* s_out /* [32] */) {
  constexpr int kParts = kUpdBlock / 32;
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;  // 32 parts x 32 columns
  double vh = 0.0, vl = 0.0;
  const double qc = q_hi[col];
  for (int b0 = part; b0 < nrows; b0 += 32 * kParts) {
    double x[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {  // the rows were written by other CUs/XCDs: all loads of a thread in flight before the first add
      const int b = b0 + k * kParts;
      x[k] = b < nrows ? partials[(size_t)b * kRec + col] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      double h, l;
      split_exact(x[k], qc, &h, &l);
      vh += h;  // exact
      vl += l;  // exact
    }
  }
  s_part[part * kRec + col] = vh;
  s_part[(kParts + part) * kRec + col] = vl;
  __syncthreads();
  if (threadIdx.x < kRec) {
    double th = 0.0, tl = 0.0;
#pragma unroll
    for (int k = 0; k < kParts; ++k) {
      th += s_part[k * kRec + threadIdx.x];
      tl += s_part[(kParts + k) * kRec + threadIdx.x];
    }
    s_out[threadIdx.x] = th + tl;  // the one rounding
  }
  __syncthreads();
}
__global__ __launch_bounds__(kUpdBlock) void icp_reduce_kernel(const double* __restrict__ partials, int nrows, const IcpStateDev* state,
                                                               double* __restrict__ record, QuantumTable qt) {
  if (state->done) return;
  __shared__ double s_part[2 * (kUpdBlock / 32) * kRec];
  __shared__ double s_out[kRec];
  reduce_partials(partials, nrows, qt.q, s_part, s_out);
  if (threadIdx.x < kRec) record[threadIdx.x] = s_out[threadIdx.x];
}

// [O3D] TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x[3:6]; column-major
__host__ __device__ inline void vector6_to_matrix4(const double x[6], double U[16]) {
  const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
  U[0] = cg * cb;
  U[1] = sg * cb;
  U[2] = -sb;
  U[3] = 0.0;
  U[4] = cg * sb * sa - sg * ca;
  U[5] = sg * sb * sa + cg * ca;
  U[6] = cb * sa;
  U[7] = 0.0;
  U[8] = cg * sb * ca + sg * sa;
  U[9] = sg * sb * ca - cg * sa;
  U[10] = cb * ca;
  U[11] = 0.0;
  U[12] = x[3];
  U[13] = x[4];
  U[14] = x[5];
  U[15] = 1.0;
}

// [O3D] SolveLinearSystemPSD(JTJ, -JTr) = JTJ.ldlt().solve(-JTr): LDL^T with largest-|diagonal| symmetric pivoting.
// Every loop is fully unrolled and the pivot swap is a uniform branch over static index pairs, so the 6x6 system
// lives in registers (no scratch): the solve sits on the serial critical path of every ICP iteration.
__host__ __device__ inline void solve6_ldlt(const double* rec, double x[6]) {
  double A[6][6], b[6];
  int perm[6];
  {
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 6; ++c) {
        A[r][c] = rec[k];
        A[c][r] = rec[k];
        ++k;
      }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    b[r] = -rec[21 + r];
    perm[r] = r;
  }
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    int piv = s;
    double best = fabs(A[s][s]);
#pragma unroll
    for (int i = s + 1; i < 6; ++i) {
      const double v = fabs(A[i][i]);
      if (v > best) {
        best = v;
        piv = i;
      }
    }
#pragma unroll
    for (int p = s + 1; p < 6; ++p) {
      if (piv == p) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double t = A[s][j];
          A[s][j] = A[p][j];
          A[p][j] = t;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double t = A[i][s];
          A[i][s] = A[i][p];
          A[i][p] = t;
        }
        const double tb = b[s];
        b[s] = b[p];
        b[p] = tb;
        const int tp = perm[s];
        perm[s] = perm[p];
        perm[p] = tp;
      }
    }
    const double d = A[s][s];
#pragma unroll
    for (int i = s + 1; i < 6; ++i) {
      const double l = A[i][s] / d;
#pragma unroll
      for (int j = s + 1; j < 6; ++j) A[i][j] -= l * A[s][j];
      A[i][s] = l;
    }
  }
  double y[6], w[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) s -= A[i][j] * y[j];
    y[i] = s;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i] / A[i][i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) s -= A[j][i] * w[j];
    w[i] = s;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (perm[i] == k) x[k] = w[i];
}

// The same solve spread over the first six lanes of a wavefront: lane i owns row i of the augmented system [JtJ | -Jtr].
// Symmetric pivoting on the largest |diagonal| (first occurrence, as Eigen's LDLT), elimination with the multipliers of
// all remaining rows computed in parallel, back substitution.  Arithmetically this is the L D L^T solve (elimination of the
// right-hand side is L y = b; back substitution on D L^T is D z = y, L^T w = z fused) with multiplications by the pivots'
// reciprocals in place of divisions, so it agrees with solve6_ldlt to rounding; it keeps ~30 VGPRs instead of ~90 and has 6 dependent divisions instead of 21, which matters because this
// tail is on the critical path of every ICP iteration (measured: 3-4 us serial, 2.1 us here; a Gauss-Jordan variant with one
// matrix element per lane -- no swaps, no back substitution, reciprocal by Newton steps -- measured 2.5 us: its pivot search lives
// on the scalar unit and every step ping-pongs between SALU and VALU).
// broadcast of lane SRC (compile-time constant after unrolling) through v_readlane: a few cycles, where ds_bpermute (__shfl)
// costs an LDS round trip -- the solve has ~80 of them on the critical path of every ICP iteration
#define O3DS_BCAST(v, SRC) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (SRC)), __builtin_amdgcn_readlane(__double2loint(v), (SRC)))

__device__ __forceinline__ void solve6_wave(const double* rec, double* x_out, int lane, int* order_out = nullptr /* [6]: row at pivot position t */) {
  double a[6], b = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int r = min(lane, j), c = max(lane, j);
    a[j] = lane < 6 ? rec[r * 6 - (r * (r - 1)) / 2 + (c - r)] : (j == 0 ? 1.0 : 0.0);
  }
  if (lane < 6) b = -rec[21 + lane];
  int perm = lane;
  double rd[6];  // reciprocals of the pivots (wave-uniform): ONE division per elimination step, none in the back substitution
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    double diag = a[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) diag = lane == k ? a[k] : diag;
    const double ad = fabs(diag);
    int piv = s;
    double best = O3DS_BCAST(ad, s);
#pragma unroll
    for (int j = s + 1; j < 6; ++j) {
      const double v = O3DS_BCAST(ad, j);
      if (v > best) {
        best = v;
        piv = j;
      }
    }
    if (piv != s) {  // wave-uniform
      const int srcl = lane == s ? piv : (lane == piv ? s : lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k] = __shfl(a[k], srcl, 64);
      b = __shfl(b, srcl, 64);
      perm = __shfl(perm, srcl, 64);
#pragma unroll
      for (int k = s + 1; k < 6; ++k) {
        if (piv == k) {
          const double t = a[s];
          a[s] = a[k];
          a[k] = t;
        }
      }
    }
    const double d = O3DS_BCAST(a[s], s);
    const double bs = O3DS_BCAST(b, s);
    // Eigen's LDLT, which [O3D] SolveLinearSystemPSD uses: a column under an exactly zero pivot is left undivided (ldlt_inplace:
    // pivot_is_valid), and the solve zeroes the components of pivots not above the smallest normal double (_solve_impl) -- a scene of
    // one plane, or fewer than six independent correspondences, gives a finite update instead of inf / NaN
    rd[s] = fabs(d) > 2.2250738585072014e-308 ? 1.0 / d : 0.0;
    const double l = d != 0.0 ? a[s] * (1.0 / d) : a[s];
    const bool below = lane > s && lane < 6;
#pragma unroll
    for (int j = s + 1; j < 6; ++j) {
      const double rs = O3DS_BCAST(a[j], s);
      if (below) a[j] = fma(-l, rs, a[j]);
    }
    if (below) b = fma(-l, bs, b);
  }
  double xs[6];
  double mine = 0.0;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double num = b;
#pragma unroll
    for (int j = i + 1; j < 6; ++j) num = fma(-a[j], xs[j], num);
    // a[i] of lane i is pivot i; after the elimination a[j] (j > i) of lane i is D_i L_ji (L_ji itself under an invalid pivot), so
    // with z_i = y_i / D_i (0 under an invalid pivot): w_i = z_i - sum_j L_ji w_j
    const double xi = rd[i] != 0.0 ? num * rd[i] : num - b;
    xs[i] = O3DS_BCAST(xi, i);
    if (lane == i) mine = xi;
  }
  if (lane < 6) {
    x_out[perm] = mine;
    if (order_out) order_out[lane] = perm;
  }
}

// The same solve when the pivot order is KNOWN (the order the previous iteration's solve found: the normal equations of consecutive ICP
// iterations are almost the same matrix).  Position t of the order is loaded into lane t with its columns in pivot order, so the
// elimination is solve6_wave's with every "is the pivot already in place" answered yes: no search, no exchange of rows and columns --
// those were two fifths of the instructions on this serial path -- and the same operations on the same values, hence the same bits.
// Each step checks that the pivot it was handed is the STRICT maximum of the remaining diagonal (then Eigen's first-maximum rule picks
// it too); if any step fails the result is discarded and the caller runs the searching solve.  Returns whether the order held.
__device__ __forceinline__ bool solve6_wave_ordered(const double* rec, double* x_out, int lane, unsigned order) {
  const int pt = (int)((order >> (3 * min(lane, 5))) & 7u);
  double a[6], b = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pj = (int)((order >> (3 * j)) & 7u);
    const int r = min(pt, pj), c = max(pt, pj);
    a[j] = lane < 6 ? rec[r * 6 - (r * (r - 1)) / 2 + (c - r)] : (j == 0 ? 1.0 : 0.0);
  }
  if (lane < 6) b = -rec[21 + pt];
  bool bad = false;
  double rd[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    double diag = a[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) diag = lane == k ? a[k] : diag;
    const double d = O3DS_BCAST(a[s], s);
    const double bs = O3DS_BCAST(b, s);
    bad |= lane > s && lane < 6 && fabs(diag) >= fabs(d);
    rd[s] = fabs(d) > 2.2250738585072014e-308 ? 1.0 / d : 0.0;
    const double l = d != 0.0 ? a[s] * (1.0 / d) : a[s];
    const bool below = lane > s && lane < 6;
#pragma unroll
    for (int j = s + 1; j < 6; ++j) {
      const double rs = O3DS_BCAST(a[j], s);
      if (below) a[j] = fma(-l, rs, a[j]);
    }
    if (below) b = fma(-l, bs, b);
  }
  if (__ballot(bad) != 0ull) return false;
  double xs[6];
  double mine = 0.0;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double num = b;
#pragma unroll
    for (int j = i + 1; j < 6; ++j) num = fma(-a[j], xs[j], num);
    const double xi = rd[i] != 0.0 ? num * rd[i] : num - b;
    xs[i] = O3DS_BCAST(xi, i);
    if (lane == i) mine = xi;
  }
  if (lane < 6) x_out[pt] = mine;
  return true;
}

// One-sided Jacobi SVD of a 3x3 (row-major): A = U diag(d) V^T, d descending; columns of U for vanishing singular values are
// completed by cross products.  One lane, f64: the point-to-point step is closed form and this is all it costs.
__device__ inline void svd3_jacobi(const double Ain[9], double U[9], double d[3], double V[9]) {
  double a0[3] = {Ain[0], Ain[3], Ain[6]}, a1[3] = {Ain[1], Ain[4], Ain[7]}, a2[3] = {Ain[2], Ain[5], Ain[8]};  // columns
  double v0[3] = {1, 0, 0}, v1[3] = {0, 1, 0}, v2[3] = {0, 0, 1};
  auto rot = [](double* ap, double* aq, double* vp, double* vq) -> double {
    const double alpha = ap[0] * ap[0] + ap[1] * ap[1] + ap[2] * ap[2], beta = aq[0] * aq[0] + aq[1] * aq[1] + aq[2] * aq[2];
    const double gamma = ap[0] * aq[0] + ap[1] * aq[1] + ap[2] * aq[2];
    if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) return 0.0;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double x = ap[r], y = aq[r];
      ap[r] = c * x - sn * y;
      aq[r] = sn * x + c * y;
      const double vx = vp[r], vy = vq[r];
      vp[r] = c * vx - sn * vy;
      vq[r] = sn * vx + c * vy;
    }
    return fabs(gamma);
  };
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = rot(a0, a1, v0, v1) + rot(a0, a2, v0, v2) + rot(a1, a2, v1, v2);
    if (off == 0.0) break;
  }
  double n0 = sqrt(a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]), n1 = sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]),
         n2 = sqrt(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2]);
  auto swp = [](double* x, double* y, double* vx, double* vy, double& nx, double& ny) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double t = x[r];
      x[r] = y[r];
      y[r] = t;
      t = vx[r];
      vx[r] = vy[r];
      vy[r] = t;
    }
    const double t = nx;
    nx = ny;
    ny = t;
  };
  if (n1 > n0) swp(a0, a1, v0, v1, n0, n1);  // descending
  if (n2 > n0) swp(a0, a2, v0, v2, n0, n2);
  if (n2 > n1) swp(a1, a2, v1, v2, n1, n2);
  const double tiny = 1e-14 * (n0 > 0 ? n0 : 1.0);
  double u0[3], u1[3], u2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    u0[r] = n0 > tiny ? a0[r] / n0 : 0.0;
    u1[r] = n1 > tiny ? a1[r] / n1 : 0.0;
    u2[r] = n2 > tiny ? a2[r] / n2 : 0.0;
  }
  if (!(n0 > tiny)) {
    u0[0] = 1, u0[1] = 0, u0[2] = 0, u1[0] = 0, u1[1] = 1, u1[2] = 0, u2[0] = 0, u2[1] = 0, u2[2] = 1;
  } else {
    if (!(n1 > tiny)) {  // any unit vector orthogonal to u0
      const int k = fabs(u0[0]) <= fabs(u0[1]) ? (fabs(u0[0]) <= fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
      const double dp = k == 0 ? u0[0] : (k == 1 ? u0[1] : u0[2]);
      double w[3] = {(k == 0 ? 1.0 : 0.0) - dp * u0[0], (k == 1 ? 1.0 : 0.0) - dp * u0[1], (k == 2 ? 1.0 : 0.0) - dp * u0[2]};
      const double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      u1[0] = w[0] / wn, u1[1] = w[1] / wn, u1[2] = w[2] / wn;
    }
    if (!(n2 > tiny)) {  // u2 = u0 x u1
      u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
      u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
      u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    }
  }
  d[0] = n0, d[1] = n1, d[2] = n2;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    U[r * 3] = u0[r], U[r * 3 + 1] = u1[r], U[r * 3 + 2] = u2[r];
    V[r * 3] = v0[r], V[r * 3 + 1] = v1[r], V[r * 3 + 2] = v2[r];
  }
}

__device__ inline double det3_rm(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// [O3D] TransformationEstimationPointToPoint::ComputeTransformation = Eigen::umeyama(source, target, with_scaling = false) from the
// point-to-point record (sums instead of demeaned matrices: Sigma = E[q p^T] - E[q] E[p]^T).  Ucm: 4x4 column-major.
__device__ inline void umeyama_from_record(const double* rec, double Ucm[16]) {
  const double m = rec[kRecCount], inv = 1.0 / m;
  const double ms[3] = {rec[9] * inv, rec[10] * inv, rec[11] * inv}, mt[3] = {rec[12] * inv, rec[13] * inv, rec[14] * inv};
  double S[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) S[a * 3 + b] = rec[a * 3 + b] * inv - mt[a] * ms[b];
  double Um[9], dd[3], Vm[9];
  svd3_jacobi(S, Um, dd, Vm);
  const double sgn = det3_rm(Um) * det3_rm(Vm) < 0.0 ? -1.0 : 1.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) Ucm[k] = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double Ra[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      Ra[b] = Um[a * 3] * Vm[b * 3] + Um[a * 3 + 1] * Vm[b * 3 + 1] + sgn * Um[a * 3 + 2] * Vm[b * 3 + 2];
      Ucm[b * 4 + a] = Ra[b];
    }
    Ucm[12 + a] = mt[a] - (Ra[0] * ms[0] + Ra[1] * ms[1] + Ra[2] * ms[2]);
  }
  Ucm[15] = 1.0;
}

// [O3D] RegistrationICP loop body after the correspondence pass: convergence test, solve, T <- U*T.
// Called by every thread of one workgroup (>= 128 threads) with the record in LDS.  This tail is on the critical path of
// every ICP iteration, so its two dependent chains run on two wavefronts (= two SIMDs) at once and meet at ONE workgroup
// barrier: wavefront 0 solves (lane-parallel), takes the three sincos in three lanes (broadcast with v_readlane) and forms
// U*T speculatively; wavefront 1 computes fitness / rmse and the convergence test and decides whether the update is
// applied.  (Measured inside icp_fused_kernel: 4.7 us for the one-thread-plus-barriers form this replaces.)
__device__ __forceinline__ void icp_step_block(const double* s_rec, IcpStateDev* st, unsigned long long n_src_total, int max_iter,
                                               double rel_fitness, double rel_rmse, double* s_x /* [8] */, double* s_sc /* [8]: scratch of the solve (pivot order) */,
                                               double* s_U /* [16] */, double* s_T /* [16] */, int* s_go,
                                               unsigned long long* tr = nullptr /* 8 timestamps, development aid */,
                                               int method = O3DS_ICP_POINT_TO_PLANE,
                                               float* s_margin = nullptr /* [2]: |R - I|_F and |t| of the update (candidate-set margin) */) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#define O3DS_TSTAMP(k)                                       \
  do {                                                       \
    if (tr && threadIdx.x == 0) tr[k] = wall_clock64();      \
  } while (0)
  O3DS_TSTAMP(0);
  const double count = s_rec[kRecCount];
  double tnew = 0.0;
  if (wv == 0) {
    if (lane < 16) s_T[lane] = st->T[lane];
    if (lane < 8) s_x[lane] = 0.0;
    lds_wave_sync();
    if (method == O3DS_ICP_POINT_TO_POINT) {  // closed form: U straight from the record, no 6-vector (uniform branch)
      if (lane == 0) {
        double Ucm[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (count > 0.0) umeyama_from_record(s_rec, Ucm);  // [O3D] corres.empty() -> Identity
#pragma unroll
        for (int k = 0; k < 16; ++k) s_U[k] = Ucm[k];
      }
      lds_wave_sync();
    } else {
    if (count > 0.0) {  // empty correspondence set => identity update (x = 0)
      // the pivot order of the previous iteration's solve travels in the state (bit 31: valid; 3 bits per position)
      const unsigned order = (unsigned)st->pad;
      bool held = false;
      if (order & 0x80000000u) held = solve6_wave_ordered(s_rec, s_x, lane, order);
      if (!held) {
        int* s_ord = (int*)s_sc;
        solve6_wave(s_rec, s_x, lane, s_ord);
        lds_wave_sync();
        if (lane == 0) {
          unsigned o = 0x80000000u;
#pragma unroll
          for (int t = 0; t < 6; ++t) o |= ((unsigned)s_ord[t] & 7u) << (3 * t);
          st->pad = (int)o;
        }
      }
      lds_wave_sync();
    }
    O3DS_TSTAMP(2);
    double sn = 0.0, cs = 1.0;
    const double ang = s_x[lane < 6 ? lane : 0];
    if (lane < 3) sincos(ang, &sn, &cs);
    const double sa = O3DS_BCAST(sn, 0), sb = O3DS_BCAST(sn, 1), sg = O3DS_BCAST(sn, 2);
    const double ca = O3DS_BCAST(cs, 0), cb = O3DS_BCAST(cs, 1), cg = O3DS_BCAST(cs, 2);
    const double tx = O3DS_BCAST(ang, 3), ty = O3DS_BCAST(ang, 4), tz = O3DS_BCAST(ang, 5);
    O3DS_TSTAMP(3);
    if (lane < 16) {  // [O3D] TransformVector6dToMatrix4d: R = Rz(x2) Ry(x1) Rx(x0), t = x[3:6]; column-major
      double v;
      switch (lane) {
        case 0: v = cg * cb; break;
        case 1: v = sg * cb; break;
        case 2: v = -sb; break;
        case 4: v = cg * sb * sa - sg * ca; break;
        case 5: v = sg * sb * sa + cg * ca; break;
        case 6: v = cb * sa; break;
        case 8: v = cg * sb * ca + sg * sa; break;
        case 9: v = sg * sb * ca - cg * sa; break;
        case 10: v = cb * ca; break;
        case 12: v = tx; break;
        case 13: v = ty; break;
        case 14: v = tz; break;
        case 15: v = 1.0; break;
        default: v = 0.0; break;
      }
      s_U[lane] = v;
    }
    lds_wave_sync();
    }
    if (lane < 16) {  // U * T
      const int c = lane >> 2, r = lane & 3;
#pragma unroll
      for (int k = 0; k < 4; ++k) tnew = fma(s_U[k * 4 + r], s_T[c * 4 + k], tnew);
    }
  } else if (wv == 1) {
    const double fitness = count > 0.0 ? count / (double)n_src_total : 0.0;
    const double rmse = count > 0.0 ? sqrt(s_rec[kRecD2] / count) : 0.0;
    const int pass = st->pass;
    const bool conv = pass > 0 && fabs(st->fitness - fitness) < rel_fitness && fabs(st->rmse - rmse) < rel_rmse;
    const int go = (!conv && st->iterations < max_iter) ? 1 : 0;
    lds_wave_sync();  // every lane has read the old state before lane 0 overwrites it
    if (lane == 0) {
      st->fitness = fitness;
      st->rmse = rmse;
      st->n_corr = (unsigned long long)(count + 0.5);
      st->pass = pass + 1;
      if (conv) st->converged = 1;
      if (!go) st->done = 1;
      *s_go = go;
    }
  }
  lds_barrier();
  if (wv == 0 && *s_go) {  // T <- U * T
    if (lane < 16) st->T[lane] = tnew;
    if (lane == 0) st->iterations += 1;
  }
  if (s_margin && wv == 2 && lane == 0) {  // how far the update moves a point p: at most |R - I|_F |p| + |t| (an idle wavefront's work)
    double w2 = 0.0, t2 = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double d = s_U[c * 4 + r] - (r == c ? 1.0 : 0.0);
        w2 += d * d;
      }
      t2 += s_U[12 + c] * s_U[12 + c];
    }
    s_margin[0] = sqrtf((float)w2);
    s_margin[1] = sqrtf((float)t2);
  }
  O3DS_TSTAMP(4);
#undef O3DS_TSTAMP
}

// single-GPU path: reduce the partials and step, one workgroup
__global__ __launch_bounds__(kUpdBlock) void icp_reduce_update_kernel(const double* __restrict__ partials, int nrows, IcpStateDev* state,
                                                                      unsigned long long n_src_total, int max_iter, double rel_fitness,
                                                                      double rel_rmse, int debug_mode, int method, QuantumTable qt) {
  if (state->done) return;
  if (debug_mode == 1) {  // timing experiment: launch + done check only
    if (threadIdx.x == 0) { state->pass += 1; state->iterations += 1; if (state->iterations > max_iter) state->done = 1; }
    return;
  }
  __shared__ double s_part[2 * (kUpdBlock / 32) * kRec];
  __shared__ double s_out[kRec];
  __shared__ double s_x[8], s_sc[8], s_U[16], s_T[16];
  __shared__ int s_go;
  reduce_partials(partials, nrows, qt.q, s_part, s_out);
  if (debug_mode == 2) {  // timing experiment: reduction only
    if (threadIdx.x == 0) { state->pass += 1; state->iterations += 1; state->fitness = s_out[28]; if (state->iterations > max_iter) state->done = 1; }
    return;
  }
  icp_step_block(s_out, state, n_src_total, max_iter, rel_fitness, rel_rmse, s_x, s_sc, s_U, s_T, &s_go, nullptr, method);
}

// sharded path: the record was all-reduced by the caller
__global__ __launch_bounds__(128) void icp_update_kernel(const double* __restrict__ record, IcpStateDev* state, unsigned long long n_src_total,
                                                        int max_iter, double rel_fitness, double rel_rmse, int method) {
  if (state->done) return;
  __shared__ double s_out[kRec];
  __shared__ double s_x[8], s_sc[8], s_U[16], s_T[16];
  __shared__ int s_go;
  if (threadIdx.x < kRec) s_out[threadIdx.x] = record[threadIdx.x];
  __syncthreads();
  icp_step_block(s_out, state, n_src_total, max_iter, rel_fitness, rel_rmse, s_x, s_sc, s_U, s_T, &s_go, nullptr, method);
}

// ----------------------------------------------------------------------------------------------
// fused form: ONE launch per pass, no separate update kernel, no grid rendezvous
// ----------------------------------------------------------------------------------------------
// The serial tail of pass j-1 (sum the records, convergence test, 6x6 solve, T <- U*T) is executed redundantly in the
// PROLOGUE of every workgroup of pass j's launch -- identical inputs, identical instruction stream, identical result in
// every workgroup -- so the dependent chain per iteration is one kernel instead of two (every kernel on the chain costs
// >= 4.6 us on MI355X).  To keep that prologue cheap the 1024 per-workgroup records of a pass are accumulated into
// kFusedSlots slot records with hardware f64 atomics: the addends are split so that the additions are exact (split_exact),
// hence the order in which workgroups finish does not matter, nobody takes a ticket, nobody waits, and the epilogue is
// fire-and-forget.  (The first form -- write-through row stores, a ticket per slot, the last arriver folding the slot in
// ascending order -- was deterministic too but put three dependent memory round trips, 2.4 us, at the end of every pass.)
// The slot buffers rotate over three launches: launch j reads buffer (j-1)%3, accumulates into j%3 and clears (j+1)%3.
constexpr int kFusedSlots = 8;   // atomics of 1024 workgroups spread over 8 addresses per term
constexpr int kSlotDoubles = 2 * kRec;  // [0..31] hi sums, [32..63] lo sums

struct IcpFusedArgs {
  IcpPassArgs pass;               // pass.state / pass.partials are unused here
  const IcpStateDev* state_in;    // written by the previous launch (unused by launch 0: its state is `init`)
  IcpStateDev* state_out;         // written by workgroup 0
  IcpStateDev* state_host;        // null, or pinned host memory that also receives the state (last launch of a chunk): the host
                                  // then only has to wait for the stream, there is no copy on the chain
  IcpStateDev init;               // launch 0: the initial state travels in the kernel arguments (no host-to-device copy)
  const double* slots_in;         // [kFusedSlots][kSlotDoubles] of the previous pass
  double* slots_out;              // [kFusedSlots][kSlotDoubles] of this pass (zero on entry)
  double* slots_clear;            // [kFusedSlots][kSlotDoubles] of the next pass: cleared by workgroup 0
  unsigned long long n_src_total;
  int max_iter;
  double rel_fitness, rel_rmse;
  int first;                      // 1: launch 0 -- there is no previous pass to fold
  int pass_index;                 // which pass of the registration this launch evaluates (= its number in the launch sequence): pass_order
  unsigned long long* trace;      // null, or [gridDim.x][16] phase timestamps (100 MHz wall clock) of thread 0 (O3DS_FUSED_TRACE)
  unsigned long long* seq_host;   // with state_host: a pinned word that receives `seq` AFTER the state (system-scope release), so that the
  unsigned long long seq;         // host can pick the state up the moment it is written instead of after the kernel's completion signal
};

// Kernel arguments live in a kernarg segment the scalar cache has never seen when a wavefront starts; the compiler fetches each
// field where it is first used and waits for it there, so a kernel with ~900 bytes of arguments pays one cold miss per 64-byte line it
// touches, one after the other (seven of them before this kernel's first global load, read off the ISA).  One scalar load per line,
// all in flight together, turns that into ONE miss; everything after it hits.
template <int kBytes>
__device__ __forceinline__ void kernarg_warm() {
  static_assert(kBytes > 0 && kBytes <= 16 * 64, "at most 16 lines");
  constexpr int kLines = (kBytes + 63) / 64;
  const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  int t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;
  // (offsets beyond the last line repeat the last dword of the segment: never past its end)
#define O3DS_KA(k) ((k) < kLines - 1 ? (k) * 64 : kBytes - 4)
  asm volatile(
      "s_load_dword %0, %16, %17\n\ts_load_dword %1, %16, %18\n\ts_load_dword %2, %16, %19\n\ts_load_dword %3, %16, %20\n\t"
      "s_load_dword %4, %16, %21\n\ts_load_dword %5, %16, %22\n\ts_load_dword %6, %16, %23\n\ts_load_dword %7, %16, %24\n\t"
      "s_load_dword %8, %16, %25\n\ts_load_dword %9, %16, %26\n\ts_load_dword %10, %16, %27\n\ts_load_dword %11, %16, %28\n\t"
      "s_load_dword %12, %16, %29\n\ts_load_dword %13, %16, %30\n\ts_load_dword %14, %16, %31\n\ts_load_dword %15, %16, %32\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10),
        "=&s"(t11), "=&s"(t12), "=&s"(t13), "=&s"(t14), "=&s"(t15)
      : "s"(ka), "n"(O3DS_KA(0)), "n"(O3DS_KA(1)), "n"(O3DS_KA(2)), "n"(O3DS_KA(3)), "n"(O3DS_KA(4)), "n"(O3DS_KA(5)), "n"(O3DS_KA(6)),
        "n"(O3DS_KA(7)), "n"(O3DS_KA(8)), "n"(O3DS_KA(9)), "n"(O3DS_KA(10)), "n"(O3DS_KA(11)), "n"(O3DS_KA(12)), "n"(O3DS_KA(13)),
        "n"(O3DS_KA(14)), "n"(O3DS_KA(15))
      : "memory");
#undef O3DS_KA
}

template <typename P4, bool kCrop, int kPassBlock, int kGroup, bool kGicp>
__global__ __launch_bounds__(kPassBlock) __attribute__((amdgpu_waves_per_eu(4))) void icp_fused_kernel(
    const IcpStateDev* state_in, const double* slots_in, int first /* = fa.state_in, fa.slots_in, first: leading scalar arguments are
    PRELOADED into SGPRs at dispatch (-mllvm -amdgpu-kernarg-preload-count), so the loads the serial tail waits for go out before the
    first byte of the argument block has been fetched */,
    IcpFusedArgs fa) {
  constexpr int kQPB = kPassBlock / kGroup;
  constexpr int kParts = kPassBlock / 32;
  __shared__ double s_rec[kQPB * (kGicp ? kRec : kRecSlots)];
  __shared__ double s_red[kParts][kRec];
  __shared__ int2 s_seg[kQPB * kSegMax];
  __shared__ double s_out[kRec];
  __shared__ double s_x[8], s_sc[8], s_U[16], s_T[16];
  __shared__ IcpStateDev s_st;
  __shared__ int s_set[kQPB * (1 + kSetCap)];
  __shared__ float s_margin[2];
  __shared__ double s_qhi[kRec];
  __shared__ int s_go;
  // ---------------- prologue: this pass's state from the previous state and the previous pass's slot records ----------------
  // The loads the serial tail waits for go out FIRST (the memory pipeline returns in order), from preloaded pointers: wavefront 0, lane l
  // holds dword l of the state and slot values l, 64 + l, ..., 448 + l (the hi sums of term l, or the lo sums of term l - 32, of the
  // eight slots)
  constexpr int kStateWords = (int)(sizeof(IcpStateDev) / sizeof(unsigned));
  static_assert(sizeof(IcpStateDev) % sizeof(unsigned) == 0 && kStateWords <= 64, "one dword of the state per lane");
  typedef unsigned __attribute__((may_alias)) word_alias;  // the state's doubles and ints travel as dwords: tell the compiler these accesses alias them
  unsigned st_word = 0u;
  double sv[kFusedSlots];
  if (!first && threadIdx.x < 64) {
    if (threadIdx.x < kStateWords) st_word = ((const word_alias*)state_in)[threadIdx.x];
#pragma unroll
    for (int k = 0; k < kFusedSlots; ++k) sv[k] = slots_in[k * kSlotDoubles + threadIdx.x];
  }
  __builtin_amdgcn_sched_barrier(0);
  kernarg_warm<(int)sizeof(IcpFusedArgs) + 24>();  // (+ the three leading arguments)
#define O3DS_STAMP(k)                                                                                  \
  do {                                                                                                 \
    if (fa.trace && threadIdx.x == 0) fa.trace[(size_t)blockIdx.x * 16 + (k)] = wall_clock64();         \
  } while (0)
  O3DS_STAMP(0);
  const bool use_cache = !first;  // launch j evaluates pass j of the registration
  // (a source whose size the host has not seen: the exact number from the device word, fetched here, used after the serial tail)
  size_t n_live = fa.pass.count;
  unsigned long long n_src_total = fa.n_src_total;
  if (fa.pass.count_dev) {
    n_live = min((size_t)*fa.pass.count_dev, fa.pass.count);
    n_src_total = (unsigned long long)n_live;
  }
  // (the quanta of the epilogue: a per-lane load of a kernel argument, fetched now and parked in LDS -- loaded at the very end it is a
  // memory round trip on the tail of every pass, kept in a register it is live through the whole kernel)
  const double q_hi_mine = threadIdx.x < kRec ? fa.pass.q_hi[threadIdx.x] : 0.0;
  __builtin_amdgcn_sched_barrier(0);
  // pose-independent loads of this workgroup's queries: source point, cached match or candidate set (points AND normals); they land
  // while the tail of the previous pass is computed
  const int order = pass_order(first ? 0 : max(fa.pass_index, 1));
  const QueryPrefetch<P4> qp = prefetch_query<P4, kPassBlock, kGroup>(fa.pass, blockIdx.x, use_cache, true, order, n_live);
  if (blockIdx.x == 0) {  // the buffer of the NEXT pass was last read two launches ago
    for (int j = threadIdx.x; j < kFusedSlots * kSlotDoubles; j += kPassBlock) fa.slots_clear[j] = 0.0;
  }
  if (first) {
    if (threadIdx.x == 0) s_st = fa.init;
  } else if (threadIdx.x < 64) {
    if (threadIdx.x < kStateWords) ((word_alias*)&s_st)[threadIdx.x] = st_word;
    // sums of hi (resp. lo) values are exact, so any order will do
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kFusedSlots; ++k) t += sv[k];
    const double tl = __shfl_down(t, 32, 64);
    if (threadIdx.x < kRec) s_out[threadIdx.x] = t + tl;  // the one rounding, as in reduce_partials
  }
  if (threadIdx.x < 2) s_margin[threadIdx.x] = threadIdx.x == 0 ? -1.0f : 0.0f;
  if (threadIdx.x < kRec) s_qhi[threadIdx.x] = q_hi_mine;
  lds_barrier();
  if (s_st.done) {  // loop already terminated: hand the final state on
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *fa.state_out = s_st;
      if (fa.state_host) {
        *fa.state_host = s_st;
        __hip_atomic_store(fa.seq_host, fa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  O3DS_STAMP(1);
  if (!first) {
    icp_step_block(s_out, &s_st, n_src_total, fa.max_iter, fa.rel_fitness, fa.rel_rmse, s_x, s_sc, s_U, s_T, &s_go,
                   fa.trace ? fa.trace + (size_t)blockIdx.x * 16 + 8 : nullptr, fa.pass.method, s_margin);
    lds_barrier();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *fa.state_out = s_st;
    if (fa.state_host) {
      *fa.state_host = s_st;
      __hip_atomic_store(fa.seq_host, fa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (s_st.done) return;
  O3DS_STAMP(2);
  // ---------------- body: correspondence + reduction pass under the new pose ----------------
  SetMargin sm;
  sm.w = s_margin[0];
  sm.t = s_margin[1];
  // a launch whose update was too large for any query to get a margin under the cap runs the pass without the list code (a second copy
  // of the pass body: one test per four candidates is a quarter of the search's instructions)
  const bool collect = fa.pass.set_pos != nullptr && sm.w >= 0.0f && fa.pass.set_gain * sm.t <= fa.pass.set_cap;
  unsigned long long* const trb = fa.trace ? fa.trace + (size_t)blockIdx.x * 16 + 13 : nullptr;
  const double v = collect ? icp_pass_body<P4, kCrop, kPassBlock, kGroup, kGicp, false, true, true>(fa.pass, s_st.T, blockIdx.x, gridDim.x, s_rec, s_red, s_seg,
                                                                                              use_cache, qp, trb, sm, s_set, n_live, order)
                           : icp_pass_body<P4, kCrop, kPassBlock, kGroup, kGicp, false, false, true>(fa.pass, s_st.T, blockIdx.x, gridDim.x, s_rec, s_red, s_seg,
                                                                                               use_cache, qp, trb, sm, s_set, n_live, order);
  O3DS_STAMP(3);
  // ---------------- epilogue: add the record to this workgroup's slot, exactly ----------------
  if (threadIdx.x < kRec) {
    double hi, lo;
    split_exact(v, s_qhi[threadIdx.x], &hi, &lo);
    double* slot = fa.slots_out + (size_t)(blockIdx.x % kFusedSlots) * kSlotDoubles;
    if (hi != 0.0) (void)__hip_atomic_fetch_add(slot + threadIdx.x, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lo != 0.0) (void)__hip_atomic_fetch_add(slot + kRec + threadIdx.x, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  O3DS_STAMP(4);
  O3DS_STAMP(5);
  O3DS_STAMP(6);
  if (fa.trace && threadIdx.x == 0) fa.trace[(size_t)blockIdx.x * 16 + 7] = 0ull;
#undef O3DS_STAMP
}

}  // namespace o3ds
#pragma clang fp contract(fast)
