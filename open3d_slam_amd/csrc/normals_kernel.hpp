// normals_kernel.hpp -- [O3D] EstimateNormals(KDTreeSearchParamHybrid(radius, max_nn)) + NormalizeNormals +
// OrientNormalsTowardsCameraLocation(0,0,0); call site CloudRegistration.cpp:49-56 (estimateNormals).
//
// SIXTEEN LANES PER POINT (one DPP row), four points per wavefront; a second kernel turns the cumulants into normals.
//   * The neighbourhood kept is a SET defined without reference to any search structure: the max_nn smallest of the points with
//     d2 < r^2, in the total order (d2, original index) -- the order the CPU checker under oracle/ fixes (its knn_accepts / knn_push).  It is found by
//     RANKING, not by insertion: the candidates a group meets are compared against the current max_nn-th key, the survivors of a
//     chunk (<= 64, four per lane) and the kept keys are ranked against each other with broadcast LDS reads (rank = number of
//     smaller keys; keys are unique), and whoever ranks below max_nn stores itself at slot [rank].  The kept list is therefore
//     always sorted, does not depend on the order the candidates arrive in (the order of the points inside a grid cell is the
//     arrival order of the index build's atomic scatter and differs from launch to launch), and there is no per-candidate
//     sweep over the list -- the sweep was 80 % of the one-lane-per-point kernel this replaces (profiles/r01_normals_kernel_work_counters.txt).
//   * The ring walk is cooperative: the (2R+1)^2 rows of ring R are tasks, sixteen per round, one per lane; a lane tests its row
//     against the current max_nn-th distance (slab gaps, as before), fetches the row's cell_start pair(s), and the segments of all
//     sixteen lanes are laid end to end (DPP row scan) so that the candidates are dealt out evenly: lane l takes candidates
//     l, l+16, ... of the round, whichever row they come from (5-step binary search in the LDS segment table).
//   * The nine cumulants are summed by nine lanes, each over the kept list IN ITS SORTED ORDER, in binary64 without contraction:
//     with f64 storage the covariance, the eigenvector (det_math.hpp) and the orientation test are the oracle's bit for bit; with
//     f32 storage the distances are f32 and everything after the selection is the same f64 arithmetic on the stored values.
//   * The closed-form eigen-solve is ~1200 instructions of f64 per point and needs one lane: normals_finish_kernel, a thread per point.
#pragma once
#include "common.hpp"
#include "det_math.hpp"
#include <type_traits>

namespace o3ds {

#ifdef O3DS_NRM_BARRIER
#define O3DS_WAVE_SYNC() __syncthreads()
#else
#define O3DS_WAVE_SYNC()                                  \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
#endif
#ifdef O3DS_NRM_PHASES  // development aid: shader-clock cycles per phase of the kernel, per wavefront (implies O3DS_NRM_CHECK)
#define O3DS_PH(k)                      \
  do {                                  \
    const long long _t = clock64();     \
    st_ph[k] += _t - st_last;           \
    st_last = _t;                       \
  } while (0)
#else
#define O3DS_PH(k) ((void)0)
#endif
#ifdef O3DS_NRM_CHECK  // development aid: invariant violations counted in a device array (o3ds_debug_counters)
__device__ unsigned int g_nrm_dbg[8];
#define O3DS_NRM_BAD(k) atomicAdd(&g_nrm_dbg[k], 1u)
#ifdef O3DS_NRM_PHASES
constexpr int kNrmStatWords = 16;
#else
constexpr int kNrmStatWords = 6;
#endif
__device__ unsigned long long* g_nrm_wave_stats;  // per wavefront: {clocks, start, rounds, max ring, chunks, merges} when set
#else
#define O3DS_NRM_BAD(k) ((void)0)
#endif

// inclusive prefix sum over the 16 lanes of a DPP row
__device__ __forceinline__ int row_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1, zero fill
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  return v;
}
// maximum over the 16 lanes of a DPP row, in every lane (rotations by 1, 2, 4, 8)
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double row_dpp(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename R>
__device__ __forceinline__ R row_max(R v) {
  v = fmax(v, row_dpp<0x121>(v));  // row_ror:1
  v = fmax(v, row_dpp<0x122>(v));
  v = fmax(v, row_dpp<0x124>(v));
  v = fmax(v, row_dpp<0x128>(v));
  return v;
}
__device__ __forceinline__ int row_last(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x15f, 0xf, 0xf, true); }  // row_newbcast:15

template <bool WIDE, int KMAX, typename R>
struct alignas(16) NrmGroupLds {
  unsigned long long Kk[KMAX];  // kept keys, ascending by (key, index)
  int Ki[WIDE ? KMAX : 4];      // original indices (f64 storage; with f32 storage the index is the key's low word)
  union {
    struct {
      unsigned long long Sk[64];  // survivors of the current chunk
      int Si[WIDE ? 64 : 4];
      int seg_off[128];   // first flat candidate number of each segment of the round (8 per lane: up to 4 rows of 2 segments)
      int seg_base[128];  // position in the sorted cloud minus seg_off
    };
    // x y z 1 of the kept points for the cumulants (after the search); widened once where that fits the space of the tables above
    std::conditional_t<(KMAX <= 32), double, R> X[KMAX][4];
  };
};

// squared distance exactly as the oracle's search_rec forms it: ((dx*dx + dy*dy) + dz*dz), every operation rounded on its own
template <typename R>
__device__ __forceinline__ R nrm_d2(R tx, R ty, R tz, R qx, R qy, R qz) {
#pragma clang fp contract(off)
  const R dx = tx - qx, dy = ty - qy, dz = tz - qz;
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ unsigned long long nrm_key(float d2, int idx) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)idx;
}
__device__ __forceinline__ unsigned long long nrm_key(double d2, int) { return (unsigned long long)__double_as_longlong(d2); }

template <bool WIDE>
__device__ __forceinline__ bool nrm_less(unsigned long long ak, int ai, unsigned long long bk, int bi) {
  if constexpr (WIDE)
    return ak < bk || (ak == bk && ai < bi);
  else
    return ak < bk;
}

#ifndef O3DS_NRM_WAVES
#define O3DS_NRM_WAVES 1
#endif
constexpr int kNrmWaves = O3DS_NRM_WAVES;  // wavefronts per workgroup (they share nothing)
constexpr int kNrmPointsPerBlock = 4 * kNrmWaves;

#ifndef O3DS_NRM_REACH_GAIN
#define O3DS_NRM_REACH_GAIN 1.1f
#endif
constexpr float nrm_reach_gain = O3DS_NRM_REACH_GAIN;  // how far beyond the estimate a sweep of a list that is not full yet reaches

template <typename P4, int KMAX>
__device__ __forceinline__ void normals_body(const P4* __restrict__ pts /* original order */, size_t n, const GridDev& g,
                                             const P4* __restrict__ sp /* sorted by cell */, double radius, int max_nn, int rmax_cells,
                                             double* __restrict__ out_sums /* [n][9], cell order */, int* __restrict__ out_cnt /* [n] */) {
  using R = typename Scalar<P4>::type;
  constexpr bool WIDE = sizeof(P4) > 16;
  constexpr int KPL = (KMAX + 15) / 16;  // kept keys per lane
  constexpr int PW = 4;                  // points per wavefront: one group of 16 lanes each
  constexpr int PB = kNrmWaves * PW;     // points per workgroup
  constexpr unsigned long long kNever = ~0ull;  // a key no candidate is smaller than (masks table entries past the end)
  __shared__ NrmGroupLds<WIDE, KMAX, R> s_grp[4 * kNrmWaves];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l = lane & 15, grp = lane >> 4;
  NrmGroupLds<WIDE, KMAX, R>& L = s_grp[wv * 4 + grp];
  const size_t base = (size_t)blockIdx.x * PB + (size_t)wv * PW;
  // Which rows of a ring the current bound still allows is a FILTER: a row let through in vain costs candidates that the exact key test
  // rejects, a row ruled out wrongly would cost a neighbour.  So the row geometry runs in f32 with every rounding pushed to the side of
  // letting through (f64 sqrt alone is ~115 cycles a lane on gfx950, scripts/ubench/op_rates.hip): gaps are LOWER bounds of the distance in
  // cells between the query and a row / cell, `left` and the x-reach UPPER bounds.  Coordinates enter as fractions of the query's own cell
  // (< 1, so f32 holds them to 3e-8) and small integer offsets, never as absolute cell numbers.
  auto gap_lo = [](int d, float f) {
    const float v = d > 0 ? (float)d - f : d < 0 ? f - (float)(d + 1) : 0.0f;
    return fmaxf(v * (1.0f - 1e-6f) - 1e-7f, 0.0f);
  };

#ifdef O3DS_NRM_CHECK
  const unsigned long long st_t0 = wall_clock64();
  unsigned int st_rounds = 0, st_ring = 0, st_chunks = 0, st_cand = 0;
#endif
#ifdef O3DS_NRM_PHASES
  long long st_ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif
  const int* __restrict__ cs = g.cell_start;
  const float cell2_lo = (float)(g.cell * g.cell * (1.0 - 4e-6));
  const float inv_cell_hi = (float)(g.inv_cell * (1.0 + 2e-6));
  // queries are taken in CELL order (sp), so the four points of a wavefront walk the same few cells
  for (int it = 0; it < PW / 4; ++it) {
    const size_t j = base + (size_t)(it * 4 + grp);
    const bool have = j < n;
    const P4 q = sp[have ? j : 0];
    const R qx = q.x, qy = q.y, qz = q.z;
    const double fx = ((double)qx - g.ox) * g.inv_cell, fy = ((double)qy - g.oy) * g.inv_cell, fz = ((double)qz - g.oz) * g.inv_cell;
    const int ix = (int)floor(fx), iy = (int)floor(fy), iz = (int)floor(fz);
    const double frx = fx - floor(fx), fry = fy - floor(fy), frz = fz - floor(fz);
    const double mf = fmin(fmin(fmin(frx, 1.0 - frx), fmin(fry, 1.0 - fry)), fmin(frz, 1.0 - frz));

    // group state (the same value in all 16 lanes) ...
    int cnt = 0;
    const R r2 = (R)(radius * radius);
    unsigned long long tau_k = nrm_key(r2, 0);  // accept iff (key, idx) < (tau_k, tau_i): d2 < r^2 until the list is full
    int tau_i = 0;
    double worst = (double)r2;
    float worst_hi = (float)worst * (1.0f + 1e-6f);
    const float frx_f = (float)frx, fry_f = (float)fry, frz_f = (float)frz;
    // A SWEEP covers the cells of the block of half-width `ring` around the query's cell that are not in the block of half-width `rin`
    // searched before (-1: nothing yet).  The first sweep is the 3x3x3 block.  How far the next one reaches is a guess that costs or saves
    // work, never a neighbour: with a full list it is the reach that makes the max_nn-th distance certain, otherwise the reach at which a
    // surface of the density seen so far holds max_nn points (a walk ring by ring spent a round per ring -- row tests, two dependent memory
    // round trips, a ranking -- on neighbourhoods that need five of them: the sparse far end of a lidar scan, a sixth of its points and
    // almost half of this kernel's time).
    int ring = 1, rin = -1, ntask = 9;
    float inv_w = 1.0f / 3.0f;
    bool active = have;
    // ... and this lane's cursor: it owns the rows t = l, l + 16, ... of the ring
    int tnext = l;

    // ROUND ONE is the same for every point: the 3 x 3 rows of the block around its cell, one per lane l < 9, each one segment of the sorted
    // cloud.  Nothing is known yet that could rule a row out (the radius test is part of the key test every candidate takes anyway), so the
    // f64 row geometry of the general round is not needed, and with one segment per lane the segment table stays in registers: the segment of a
    // flat candidate number comes from eight DPP row broadcasts and compares, not from a search of the LDS table.
    bool round1 = true;  // wavefront-uniform
    int r1_off = 0, r1_base = 0;
    // the nine rows from the middle outwards -- the query's own row, its four neighbours, the four diagonal ones -- so that a second chunk,
    // when the block holds more than 64 points, is made of the rows least likely to hold anything below the bound the first chunk set
    const int r1_dz = (int)((164373u >> (2 * l)) & 3u) - 1, r1_dy = (int)((139617u >> (2 * l)) & 3u) - 1;  // lanes 0 .. 8; the others take no row

    O3DS_PH(0);  // set-up
    for (;;) {
      int T = 0;
      if (round1) {
        int s0 = 0, e0 = 0;
        const int z = iz + r1_dz, y = iy + r1_dy;
        if (active && l < 9 && (unsigned)z < (unsigned)g.nz && (unsigned)y < (unsigned)g.ny) {
          const int row = (z * g.ny + y) * g.nx;
          s0 = cs[row + max(ix - 1, 0)];
          e0 = cs[row + min(ix + 1, g.nx - 1) + 1];
        }
        const int tot = e0 - s0, incl = row_incl_scan(tot);
        T = row_last(incl);
        r1_off = incl - tot;
        r1_base = s0 - r1_off;
        tnext = ntask;  // the nine rows are taken: the general round starts by moving on to ring 2
#ifdef O3DS_NRM_CHECK
        ++st_rounds;
        st_ring = max(st_ring, 1u);
#endif
        O3DS_PH(2);
      } else {
      // ---- find work: every lane moves to its next row that the current bound does not rule out (arithmetic only: at ring 5-6 of a
      // sparse neighbourhood nearly all of the 100-200 rows are ruled out); when no lane of the group has one left the ring is done.
      // A lane takes up to FOUR rows per round (their cell_start loads are in flight together): a walk to ring 6 of a neighbourhood
      // that never fills its list -- 31 rounds of 16 rows, each a chain of two memory round trips -- becomes 10 rounds.
      int ss[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ee[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      auto next_row = [&](int& row, int& dzf, int& dyf, float& left) -> bool {
        while (tnext < ntask) {
          const int w = 2 * ring + 1;
          const int tz = (int)(((float)tnext + 0.5f) * inv_w);  // exact for these sizes
          const int dz = tz - ring, dy = tnext - tz * w - ring;
          const int z = iz + dz, y = iy + dy;
          tnext += 16;
          if ((unsigned)z >= (unsigned)g.nz || (unsigned)y >= (unsigned)g.ny) continue;
          const float gz = gap_lo(dz, frz_f), gy = gap_lo(dy, fry_f);
          left = fmaf(-(gz * gz + gy * gy) * (1.0f - 1e-6f), cell2_lo, worst_hi) + 2e-7f * worst_hi;  // what the x-offset may still use (upper bound)
          if (left <= 0.0f) continue;
          row = (z * g.ny + y) * g.nx;
          dzf = dz, dyf = dy;
          return true;
        }
        return false;
      };
      // up to two segments of the sorted cloud for one row
      auto load_row = [&](int row, int dzf, int dyf, float left, int& s0, int& e0, int& s1, int& e1) {
        const float wx = __builtin_amdgcn_sqrtf(left) * inv_cell_hi * (1.0f + 1e-5f) + 1e-6f;  // in cells (upper bound; the instruction is good to 1 ulp)
        const int xlo = max(-ring, (int)floorf(frx_f - wx)), xhi = min(ring, (int)floorf(frx_f + wx));  // cells of the row in reach, relative to ix
        const bool inner = abs(dzf) <= rin && abs(dyf) <= rin;  // a row through the block searched before: its cells |dx| <= rin are done
        if (!inner) {
          const int x0 = max(ix + xlo, 0), x1 = min(ix + xhi, g.nx - 1);
          if (x0 <= x1) {
            s0 = cs[row + x0];
            e0 = cs[row + x1 + 1];
          }
        } else {
          const int a0 = max(ix + xlo, 0), a1 = min(ix - rin - 1, g.nx - 1), b0 = max(ix + rin + 1, 0), b1 = min(ix + xhi, g.nx - 1);
          const bool okl = a0 <= a1, okr = b0 <= b1;
          const int sl = cs[okl ? row + a0 : 0], el = cs[okl ? row + a1 + 1 : 0], sr = cs[okr ? row + b0 : 0], er = cs[okr ? row + b1 + 1 : 0];
          if (okl) s0 = sl, e0 = el;
          if (okr) s1 = sr, e1 = er;
        }
      };
      bool found = false;
      int row = 0, dzf = 0, dyf = 0;
      float left = 0.0f;
      for (;;) {
        if (active && !found) found = next_row(row, dzf, dyf, left);
        const unsigned int mine = (unsigned int)(__ballot(found) >> (grp * 16)) & 0xffffu;
        const bool next_ring = active && mine == 0u;
        if (__ballot(next_ring) == 0ull) break;
        if (next_ring) {
          rin = ring;
          const double lb = g.cell * ((double)rin + mf) * (1.0 - 1e-6);
          if (rin >= rmax_cells || worst <= lb * lb) {  // the max_nn-th best (or r^2) already lies inside the searched block
            active = false;
          } else {
            const float reach = cnt == max_nn ? __builtin_amdgcn_sqrtf(worst_hi) * inv_cell_hi - (float)mf
                                              : ((float)rin + 0.5f) * __builtin_amdgcn_sqrtf((float)max_nn * __builtin_amdgcn_rcpf((float)max(cnt, 1))) * nrm_reach_gain;
            ring = min(max((int)ceilf(reach), rin + 1), rmax_cells);
            ntask = (2 * ring + 1) * (2 * ring + 1);
            inv_w = 1.0f / (float)(2 * ring + 1);
            tnext = l;
          }
        }
      }
      O3DS_PH(1);  // find work (arithmetic only)
      if (__ballot(found) == 0ull) break;  // no group has anything left
#ifdef O3DS_NRM_CHECK
      ++st_rounds;
      st_ring = max(st_ring, (unsigned int)__builtin_amdgcn_readfirstlane(ring));
#endif
      if (found) load_row(row, dzf, dyf, left, ss[0], ee[0], ss[1], ee[1]);
#pragma unroll
      for (int k = 1; k < 4; ++k) {  // further rows of the same ring for this lane
        if (__ballot(found && tnext < ntask) == 0ull) break;
        if (found) found = next_row(row, dzf, dyf, left);
        if (found) load_row(row, dzf, dyf, left, ss[2 * k], ee[2 * k], ss[2 * k + 1], ee[2 * k + 1]);
      }
      // ---- lay the segments of the 16 lanes end to end
      int tot = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) tot += ee[k] - ss[k];
      const int incl = row_incl_scan(tot);
      T = row_last(incl);
      {
        int run = incl - tot;
        int so[8], sb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          so[k] = run;
          sb[k] = ss[k] - run;
          run += ee[k] - ss[k];
        }
        int4* po = reinterpret_cast<int4*>(&L.seg_off[8 * l]);
        int4* pb = reinterpret_cast<int4*>(&L.seg_base[8 * l]);
        po[0] = make_int4(so[0], so[1], so[2], so[3]);
        po[1] = make_int4(so[4], so[5], so[6], so[7]);
        pb[0] = make_int4(sb[0], sb[1], sb[2], sb[3]);
        pb[1] = make_int4(sb[4], sb[5], sb[6], sb[7]);
      }
      O3DS_WAVE_SYNC();
      O3DS_PH(2);  // row bounds (cell_start loads) + segment table
      }

      for (int f0 = 0; __ballot(f0 < T) != 0ull; f0 += 64) {
        // ---- four candidates per lane (fewer when the round has few left: nslot is wavefront-uniform): flat number -> segment by a
        // 7-step search of the table, the searches of a lane in step
#ifdef O3DS_NRM_CHECK
        ++st_chunks;
#endif
        const int nslot = __ballot(T - f0 > 48) != 0ull ? 4 : __ballot(T - f0 > 32) != 0ull ? 3 : __ballot(T - f0 > 16) != 0ull ? 2 : 1;
        int sb[4] = {0, 0, 0, 0};
        if (round1) {  // the last lane m whose first candidate number is <= f (an empty segment shares its successor's number, which wins)
          const int b0 = __builtin_amdgcn_update_dpp(0, r1_base, 0x150, 0xf, 0xf, true);  // row_newbcast:0
          sb[0] = sb[1] = sb[2] = sb[3] = b0;
#define O3DS_R1_STEP(m)                                                                        \
  {                                                                                            \
    const int o = __builtin_amdgcn_update_dpp(0, r1_off, 0x150 + (m), 0xf, 0xf, true);        \
    const int b = __builtin_amdgcn_update_dpp(0, r1_base, 0x150 + (m), 0xf, 0xf, true);       \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) sb[c] = o <= f0 + l + 16 * c ? b : sb[c];   \
  }
          O3DS_R1_STEP(1) O3DS_R1_STEP(2) O3DS_R1_STEP(3) O3DS_R1_STEP(4) O3DS_R1_STEP(5) O3DS_R1_STEP(6) O3DS_R1_STEP(7) O3DS_R1_STEP(8)
#undef O3DS_R1_STEP
        } else {
          int pos[4] = {0, 0, 0, 0};  // last segment whose first candidate number is <= f (empty segments share their successor's)
#pragma unroll
          for (int step = 64; step >= 1; step >>= 1) {
            int v[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < nslot) v[c] = L.seg_off[pos[c] + step];
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < nslot && v[c] <= f0 + l + 16 * c) pos[c] += step;
          }
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < nslot) sb[c] = L.seg_base[pos[c]];
        }
        O3DS_PH(3);  // flat number -> segment
        P4 cand[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int f = f0 + l + 16 * c;
          int p = f < T ? sb[c] + f : 0;
#ifdef O3DS_NRM_CHECK
          if ((size_t)(unsigned)p >= n) {
            O3DS_NRM_BAD(1);
            p = 0;
          }
#endif
          if (c < nslot)
            cand[c] = sp[p];
          else
            cand[c] = q;
        }
        R cd[4];
        unsigned long long ck[4];
        int ci[4];
        bool sv[4];
        int ns = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          cd[c] = nrm_d2<R>(cand[c].x, cand[c].y, cand[c].z, qx, qy, qz);
          ci[c] = (int)cand[c].i;
          ck[c] = nrm_key(cd[c], ci[c]);
          sv[c] = (f0 + l + 16 * c < T) && nrm_less<WIDE>(ck[c], ci[c], tau_k, tau_i);
          ns += sv[c] ? 1 : 0;
        }
        int sincl = row_incl_scan(ns);
        int stot = row_last(sincl);
        O3DS_PH(4);  // candidate loads, distances, keys, first test
        if (__ballot(stot > 0) == 0ull) continue;
        // ---- a first chunk usually brings 40-60 candidates inside the radius for max_nn places, and ranking costs (survivors)^2.
        // So the bound is tightened first: a distance t with max_nn <= #{d2 < t} <= max_nn + 8 is found by regula falsi on the COUNT
        // (on a surface the count grows like t) -- a few compare-and-count steps -- and only candidates with d2 < t are ranked.  At
        // least max_nn candidates are below t, so the max_nn smallest by (d2, index) all are: the result is the same set.
        {
          const bool refine0 = cnt == 0 && stot > max_nn + 8;
          if (__ballot(refine0) != 0ull) {
            bool refine = refine0;
            // the bracket starts at the largest distance present (just above it: all stot candidates lie below), not at r^2 -- with r = 3 m
            // and a block 1 m across the interpolation spent its first three steps walking down from 9 m^2
            R dmax = (R)0;
#pragma unroll
            for (int c = 0; c < 4; ++c) dmax = sv[c] ? fmax(dmax, cd[c]) : dmax;
            dmax = row_max<R>(dmax);
            R t_lo = (R)0, t_hi = dmax * (R)(1.0 + 1e-6) + (R)1e-30;
            int c_lo = 0, c_hi = stot;
            for (int stepn = 0; stepn < 6 && __ballot(refine) != 0ull; ++stepn) {
              const R t = t_lo + (t_hi - t_lo) * ((R)(max_nn + 4 - c_lo) * (R)__builtin_amdgcn_rcpf((float)(c_hi - c_lo)));  // any t in between will do
              int cl = 0;
#pragma unroll
              for (int c = 0; c < 4; ++c) cl += (sv[c] && cd[c] < t) ? 1 : 0;
              const int ctot = row_last(row_incl_scan(cl));
              if (refine) {
                if (ctot < max_nn) {
                  t_lo = t, c_lo = ctot;
                } else {
                  t_hi = t, c_hi = ctot;
                  if (ctot <= max_nn + 8) refine = false;
                }
              }
            }
            if (refine0) {  // t_hi always has at least max_nn candidates below it
              ns = 0;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                sv[c] = sv[c] && cd[c] < t_hi;
                ns += sv[c] ? 1 : 0;
              }
            }
            sincl = row_incl_scan(ns);
            stot = row_last(sincl);
          }
        }
        O3DS_PH(5);  // regula falsi on the count
        {
          int slot = sincl - ns;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (sv[c]) {
              L.Sk[slot] = ck[c];
              if constexpr (WIDE) L.Si[slot] = ci[c];
              ++slot;
            }
          // the ranking loops run to the longest table of the wavefront's four groups, four entries at a time: fill this group's up to
          // there with a key nothing is smaller than
          const int smax = max(max(__builtin_amdgcn_readlane(stot, 0), __builtin_amdgcn_readlane(stot, 16)),
                               max(__builtin_amdgcn_readlane(stot, 32), __builtin_amdgcn_readlane(stot, 48)));
          const int send = (smax + 3) & ~3;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (stot + l + 16 * k < send) L.Sk[stot + l + 16 * k] = kNever;
        }
        O3DS_WAVE_SYNC();
        O3DS_PH(6);  // compaction
        // ---- rank survivors and kept keys against each other: lane l owns survivors l, l + 16, ... and kept keys l, l + 16, ...
        // Table entries are read four at a time (one wait per four); entries past the end are replaced by a key nothing is
        // smaller than.  The number of owned survivors per lane (1..4) is a wavefront-uniform switch.
        auto merge = [&](auto nu_tag, auto kept_tag) {
          constexpr int NU = decltype(nu_tag)::value;
          constexpr bool KEPT = decltype(kept_tag)::value;
          unsigned long long ok[NU], kk[KPL];
          int oi[NU], ki[KPL], rs[NU], rk[KPL];
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int idx = l + 16 * u;
            ok[u] = idx < stot ? L.Sk[idx] : kNever;
            oi[u] = 0;
            if constexpr (WIDE) oi[u] = L.Si[idx];
            rs[u] = 0;
          }
#pragma unroll
          for (int u = 0; u < KPL; ++u) {
            const int idx = l + 16 * u;
            kk[u] = (KEPT && idx < cnt) ? L.Kk[idx] : kNever;
            ki[u] = 0;
            if constexpr (WIDE && KEPT) ki[u] = L.Ki[idx < KMAX ? idx : 0];
            rk[u] = idx;
          }
          for (int jj = 0; __ballot(jj < stot) != 0ull; jj += 4) {
            const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&L.Sk[jj]), b = *reinterpret_cast<const ulonglong2*>(&L.Sk[jj + 2]);
            unsigned long long sk[4] = {a.x, a.y, b.x, b.y};
            int si[4] = {0, 0, 0, 0};
            if constexpr (WIDE) {
              const int4 t4 = *reinterpret_cast<const int4*>(&L.Si[jj]);
              si[0] = t4.x, si[1] = t4.y, si[2] = t4.z, si[3] = t4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
              for (int u = 0; u < NU; ++u) rs[u] += nrm_less<WIDE>(sk[e], si[e], ok[u], oi[u]) ? 1 : 0;
              if constexpr (KEPT) {
#pragma unroll
                for (int u = 0; u < KPL; ++u) rk[u] += nrm_less<WIDE>(sk[e], si[e], kk[u], ki[u]) ? 1 : 0;
              }
            }
          }
          if constexpr (KEPT) {
            for (int jj = 0; __ballot(jj < cnt) != 0ull; jj += 4) {
              const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&L.Kk[jj]), b = *reinterpret_cast<const ulonglong2*>(&L.Kk[jj + 2]);
              unsigned long long sk[4] = {a.x, a.y, b.x, b.y};
              int si[4] = {0, 0, 0, 0};
              if constexpr (WIDE) {
                const int4 t4 = *reinterpret_cast<const int4*>(&L.Ki[jj]);
                si[0] = t4.x, si[1] = t4.y, si[2] = t4.z, si[3] = t4.w;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int u = 0; u < NU; ++u) rs[u] += nrm_less<WIDE>(sk[e], si[e], ok[u], oi[u]) ? 1 : 0;
              }
            }
          }
          O3DS_WAVE_SYNC();
          if (stot > 0) {
            if constexpr (KEPT) {
#pragma unroll
              for (int u = 0; u < KPL; ++u)
                if (l + 16 * u < cnt && rk[u] < max_nn) {
                  L.Kk[rk[u]] = kk[u];
                  if constexpr (WIDE) L.Ki[rk[u]] = ki[u];
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
              if (l + 16 * u < stot && rs[u] < max_nn) {
                L.Kk[rs[u]] = ok[u];
                if constexpr (WIDE) L.Ki[rs[u]] = oi[u];
              }
          }
        };
        {
          const bool any_kept = __ballot(cnt > 0 && stot > 0) != 0ull;
          const bool n2 = __ballot(stot > 16) != 0ull, n3 = __ballot(stot > 32) != 0ull, n4 = __ballot(stot > 48) != 0ull;
          using T1 = std::integral_constant<int, 1>;
          using T2 = std::integral_constant<int, 2>;
          using T3 = std::integral_constant<int, 3>;
          using T4 = std::integral_constant<int, 4>;
          if (any_kept) {
            if (n4) merge(T4{}, std::true_type{});
            else if (n3) merge(T3{}, std::true_type{});
            else if (n2) merge(T2{}, std::true_type{});
            else merge(T1{}, std::true_type{});
          } else {
            if (n4) merge(T4{}, std::false_type{});
            else if (n3) merge(T3{}, std::false_type{});
            else if (n2) merge(T2{}, std::false_type{});
            else merge(T1{}, std::false_type{});
          }
        }
        O3DS_WAVE_SYNC();
        cnt = min(cnt + stot, max_nn);
        {  // same fill for the kept list
          const int cmax = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                               max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
          const int cend = (cmax + 3) & ~3;
#pragma unroll
          for (int k = 0; k < KPL; ++k)
            if (cnt + l + 16 * k < cend) L.Kk[cnt + l + 16 * k] = kNever;
        }
#ifdef O3DS_NRM_CHECK
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
          const int idx = l + 16 * u;
          if (idx + 1 < cnt) {
            int i0 = 0, i1 = 0;
            if constexpr (WIDE) i0 = L.Ki[idx], i1 = L.Ki[idx + 1];
            if (!nrm_less<WIDE>(L.Kk[idx], i0, L.Kk[idx + 1], i1)) O3DS_NRM_BAD(0);
          }
        }
#endif
        if (cnt == max_nn) {  // list full: the bound becomes the max_nn-th key
          tau_k = L.Kk[max_nn - 1];
          if constexpr (WIDE) {
            tau_i = L.Ki[max_nn - 1];
            worst = __longlong_as_double((long long)tau_k);
            worst_hi = (float)worst * (1.0f + 1e-6f);
          } else {
            tau_i = 0;
            worst_hi = __uint_as_float((unsigned int)(tau_k >> 32));
            worst = (double)worst_hi;
            worst_hi *= 1.0f + 1e-6f;
          }
        }
        O3DS_PH(7);  // ranking + new bound
      }
      O3DS_WAVE_SYNC();  // the segment table is rewritten by the next round
      round1 = false;
    }

    // ---- cumulants over the kept list in its order
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      const int idx = l + 16 * u;
      if (idx < cnt) {
        int oi;
        if constexpr (WIDE)
          oi = L.Ki[idx];
        else
          oi = (int)(unsigned int)(L.Kk[idx] & 0xffffffffull);
#ifdef O3DS_NRM_CHECK
        if ((size_t)(unsigned)oi >= n) {
          O3DS_NRM_BAD(2);
          oi = 0;
        }
#endif
        const P4 tpt = pts[oi];
        using XT = std::remove_reference_t<decltype(L.X[0][0])>;
        L.X[idx][0] = (XT)tpt.x;
        L.X[idx][1] = (XT)tpt.y;
        L.X[idx][2] = (XT)tpt.z;
        L.X[idx][3] = (XT)1;
      }
    }
    O3DS_WAVE_SYNC();
    {
      // lane t < 9 sums term t = {x y z xx xy xz yy yz zz}[t] as X[.][ia] * X[.][ib] (x = x * 1 exactly), one after the other in
      // list order (the additions are a chain by definition; the operands of four steps are fetched together)
      const int ia = l < 3 ? l : (l < 6 ? 0 : (l < 8 ? 1 : 2));
      const int ib = l < 3 ? 3 : (l < 6 ? l - 3 : (l < 8 ? l - 5 : 2));
      double acc = 0.0;
      if (l < 9 && have) {
#pragma clang fp contract(off)
        int jj = 0;
        for (; jj + 4 <= cnt; jj += 4) {
          const double a0 = (double)L.X[jj][ia], b0 = (double)L.X[jj][ib], a1 = (double)L.X[jj + 1][ia], b1 = (double)L.X[jj + 1][ib];
          const double a2 = (double)L.X[jj + 2][ia], b2 = (double)L.X[jj + 2][ib], a3 = (double)L.X[jj + 3][ia], b3 = (double)L.X[jj + 3][ib];
          acc += a0 * b0;
          acc += a1 * b1;
          acc += a2 * b2;
          acc += a3 * b3;
        }
        for (; jj < cnt; ++jj) acc += (double)L.X[jj][ia] * (double)L.X[jj][ib];
        out_sums[9 * j + l] = acc;
      }
      if (l == 0 && have) out_cnt[j] = cnt;
    }
    O3DS_WAVE_SYNC();
    O3DS_PH(8);  // cumulants
  }
#ifdef O3DS_NRM_CHECK
  if (g_nrm_wave_stats && lane == 0) {
    unsigned long long* o = g_nrm_wave_stats + kNrmStatWords * ((size_t)blockIdx.x * kNrmWaves + wv);
    o[0] = wall_clock64() - st_t0, o[1] = st_t0, o[2] = st_rounds, o[3] = st_ring, o[4] = st_chunks, o[5] = 0;
#ifdef O3DS_NRM_PHASES
    for (int k = 0; k < 10; ++k) o[6 + k] = (unsigned long long)st_ph[k];
#endif
  }
#endif
}

template <typename P4, int KMAX>
__global__ __launch_bounds__(64 * kNrmWaves) void normals_kernel(const P4* __restrict__ pts, size_t n, GridDev g, const P4* __restrict__ sp, double radius,
                                                                 int max_nn, int rmax_cells, double* __restrict__ out_sums, int* __restrict__ out_cnt,
                                                                 const int* __restrict__ n_dev = nullptr /* the exact count when n is an upper bound */) {
  if (n_dev) n = (size_t)*n_dev;
  // (no early return for the workgroups past the end: it cost the occ5 instantiation its last registers -- 12 bytes of scratch;
  // their groups find `have` false and fall through)
  normals_body<P4, KMAX>(pts, n, g, sp, radius, max_nn, rmax_cells, out_sums, out_cnt);
}
// The instantiation the lidar stream uses (f32 storage, max_nn <= 32) fits 96 registers without spilling when asked to: five wavefronts per
// SIMD instead of four (their 5 x 4 x 7.3 KB of LDS just fit a CU's 160 KB).  The kernel is a chain of dependent memory round trips per
// point (query -> cell table -> candidates -> kept points), so a fifth wavefront to switch to is worth 7 % (100.5 -> 92.7 us per 108 k points).
template <typename P4, int KMAX>
__global__ __launch_bounds__(64 * kNrmWaves) __attribute__((amdgpu_waves_per_eu(5))) void normals_kernel_occ5(const P4* __restrict__ pts, size_t n, GridDev g,
                                                                                                              const P4* __restrict__ sp, double radius, int max_nn,
                                                                                                              int rmax_cells, double* __restrict__ out_sums,
                                                                                                              int* __restrict__ out_cnt, const int* __restrict__ n_dev = nullptr) {
  if (n_dev) n = (size_t)*n_dev;
  normals_body<P4, KMAX>(pts, n, g, sp, radius, max_nn, rmax_cells, out_sums, out_cnt);
}

// Covariance from the cumulants, eigenvector of the smallest eigenvalue, normalise, orient: ~1200 instructions of f64 per point that
// need one lane each -- run as a kernel of their own, one thread per point, rather than on 4 of the 64 lanes of a search wavefront
template <typename P4>
__global__ __launch_bounds__(256) void normals_finish_kernel(const P4* __restrict__ sp /* sorted by cell */, size_t n, const double* __restrict__ sums,
                                                             const int* __restrict__ cnts, P4* __restrict__ out_nrm, int raw = 0,
                                                             P4* __restrict__ out_sorted = nullptr /* the same normals in cell order */,
                                                             const int* __restrict__ n_dev = nullptr) {
  using R = typename Scalar<P4>::type;
  if (n_dev) n = (size_t)*n_dev;
  const size_t pj = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pj < n) {
    const P4 q = sp[pj];
    const int k = cnts[pj];
    double cov[6] = {1, 0, 0, 1, 0, 1};
    if (k >= 3) {
      double s[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) s[t] = sums[9 * pj + t];
      det::cov_from_cumulants(s, k, cov);
    }
    double nv[3];
    det::fast_eigen3x3_min(cov, nv);
    det::normalize_orient(nv, (double)q.x, (double)q.y, (double)q.z, raw != 0);
    P4 o;
    o.x = (R)nv[0];
    o.y = (R)nv[1];
    o.z = (R)nv[2];
    o.i = 0;
    out_nrm[(size_t)q.i] = o;
    if (out_sorted) out_sorted[pj] = o;
  }
}

}  // namespace o3ds
