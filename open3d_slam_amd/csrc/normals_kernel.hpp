// normals_kernel.hpp -- [O3D] EstimateNormals(KDTreeSearchParamHybrid(radius, max_nn)) + NormalizeNormals +
// OrientNormalsTowardsCameraLocation(0,0,0); call site CloudRegistration.cpp:49-56 (estimateNormals).
//
// SIXTEEN LANES PER POINT (one DPP row), four points per wavefront, sixteen points per wavefront in four rounds.
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
//   * The closed-form eigen-solve is ~1200 instructions of f64 per point; it runs once per wavefront for its 16 points on 16 lanes.
#pragma once
#include "common.hpp"
#include "det_math.hpp"

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
#ifdef O3DS_NRM_CHECK  // development aid: invariant violations counted in a device array (o3ds_debug_counters)
__device__ unsigned int g_nrm_dbg[8];
__device__ int g_nrm_info[16];
__device__ int g_nrm_segs[64 * 4];
__device__ double g_nrm_q[8];
#define O3DS_NRM_BAD(k) atomicAdd(&g_nrm_dbg[k], 1u)
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
__device__ __forceinline__ int row_last(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x15f, 0xf, 0xf, true); }  // row_newbcast:15

template <bool WIDE, int KMAX, typename R>
struct NrmGroupLds {
  unsigned long long Kk[KMAX];  // kept keys, ascending by (key, index)
  unsigned long long Sk[64];    // survivors of the current chunk
  int Ki[WIDE ? KMAX : 1];      // original indices (f64 storage; with f32 storage the index is the key's low word)
  int Si[WIDE ? 64 : 1];
  int seg_off[32];   // first flat candidate number of each segment of the round
  int seg_base[32];  // position in the sorted cloud minus seg_off
  R X[KMAX][4];      // x y z 1 of the kept points, for the cumulants
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

template <typename P4, int KMAX>
__global__ __launch_bounds__(64) void normals_kernel(const P4* __restrict__ pts /* original order */, size_t n, GridDev g,
                                                     const P4* __restrict__ sp /* sorted by cell */, double radius, int max_nn, int rmax_cells,
                                                     P4* __restrict__ out_nrm) {
  using R = typename Scalar<P4>::type;
  constexpr bool WIDE = sizeof(P4) > 16;
  constexpr int KPL = (KMAX + 15) / 16;  // kept keys per lane
  constexpr int PW = 16;                 // points per wavefront
  __shared__ NrmGroupLds<WIDE, KMAX, R> s_grp[4];
  __shared__ double s_sum[PW][9];
  __shared__ int s_cnt[PW];
#ifdef O3DS_NRM_CHECK
  __shared__ int s_dbg[64][4];
#endif
  const int lane = threadIdx.x, l = lane & 15, grp = lane >> 4;
  NrmGroupLds<WIDE, KMAX, R>& L = s_grp[grp];
  const int* __restrict__ cs = g.cell_start;
  const size_t base = (size_t)blockIdx.x * PW;
  const double cell2 = g.cell * g.cell * (1.0 - 2e-6);
  auto gap = [](int d, double f) { return d > 0 ? (double)d - f : d < 0 ? f - (double)(d + 1) : 0.0; };

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

    // group state (the same value in all 16 lanes)
    int cnt = 0;
    const R r2 = (R)(radius * radius);
    unsigned long long tau_k = nrm_key(r2, 0);  // accept iff (key, idx) < (tau_k, tau_i): d2 < r^2 until the list is full
    int tau_i = 0;
    double worst = (double)r2;
    int ring = 1, tbase = 0, ntask = 9;  // ring 1 = the whole 3x3x3 block (rings 0 and 1 of the old walk)
    bool active = have;

    while (__ballot(active) != 0ull) {
      // ---- this lane's row of the round: up to two segments [s0,e0) [s1,e1) of the sorted cloud
      int s0 = 0, e0 = 0, s1 = 0, e1 = 0;
      const int t = tbase + l;
      if (active && t < ntask) {
        const int w = 2 * ring + 1;
        const int tz = t / w;
        const int dz = tz - ring, dy = t - tz * w - ring;
        const int z = iz + dz, y = iy + dy;
        if ((unsigned)z < (unsigned)g.nz && (unsigned)y < (unsigned)g.ny) {
          const double gz = gap(dz, frz), gy = gap(dy, fry);
          const double left = worst - (gz * gz + gy * gy) * cell2;  // what the x-offset may still use; the bound only ever shrinks
          if (left > 0.0) {
            const int row = (z * g.ny + y) * g.nx;
            const bool full = ring == 1 || dz == -ring || dz == ring || dy == -ring || dy == ring;
            if (full) {
              const double wx = sqrt(left) * g.inv_cell * (1.0 + 1e-6);  // in cells
              const int x0 = max(max(ix - ring, 0), (int)floor(fx - wx)), x1 = min(min(ix + ring, g.nx - 1), (int)floor(fx + wx));
              if (x0 <= x1) {
                s0 = cs[row + x0];
                e0 = cs[row + x1 + 1];
              }
            } else {  // interior rows: only the two end cells are new
              const int xl = ix - ring, xr = ix + ring;
              const double w2 = left * g.inv_cell * g.inv_cell * (1.0 + 4e-6);
              const double gl = gap(-ring, frx), gr = gap(ring, frx);
              const bool okl = (unsigned)xl < (unsigned)g.nx && gl * gl < w2, okr = (unsigned)xr < (unsigned)g.nx && gr * gr < w2;
              const int il = okl ? row + xl : 0, ir = okr ? row + xr : 0;
              const int sl = cs[il], el = cs[il + 1], sr = cs[ir], er = cs[ir + 1];
              if (okl) s0 = sl, e0 = el;
              if (okr) s1 = sr, e1 = er;
            }
          }
        }
      }
#ifdef O3DS_NRM_CHECK
      if (g_nrm_dbg[5] == 0u) {  // until the first offence: remember the last round of every lane of this wavefront
        s_dbg[lane][0] = s0, s_dbg[lane][1] = e0, s_dbg[lane][2] = s1, s_dbg[lane][3] = e1;
      }
#endif
      // ---- lay the segments of the 16 lanes end to end
      const int len0 = e0 - s0, len1 = e1 - s1;
      const int incl = row_incl_scan(len0 + len1);
      const int off = incl - (len0 + len1);
      const int T = row_last(incl);
      L.seg_off[2 * l] = off;
      L.seg_base[2 * l] = s0 - off;
      L.seg_off[2 * l + 1] = off + len0;
      L.seg_base[2 * l + 1] = s1 - (off + len0);
      O3DS_WAVE_SYNC();

      for (int f0 = 0; __ballot(f0 < T) != 0ull; f0 += 64) {
        // ---- four candidates per lane
        unsigned long long ck[4];
        int ci[4];
        bool sv[4];
        int ns = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int f = f0 + l + 16 * c;
          const bool valid = f < T;
          int pos = 0;  // last segment whose first candidate number is <= f (empty segments share their successor's number)
#pragma unroll
          for (int step = 16; step >= 1; step >>= 1)
            if (L.seg_off[pos + step] <= f) pos += step;
          int p = valid ? L.seg_base[pos] + f : 0;
#ifdef O3DS_NRM_CHECK
          if ((size_t)(unsigned)p >= n || p < 0) {
            O3DS_NRM_BAD(1);
            p = 0;
          }
          if (valid && !(L.seg_off[pos] <= f && (pos == 31 || L.seg_off[pos + 1] > f))) O3DS_NRM_BAD(3);
#endif
          const P4 cand = sp[p];
          const R d2 = nrm_d2<R>(cand.x, cand.y, cand.z, qx, qy, qz);
          ci[c] = (int)cand.i;
          ck[c] = nrm_key(d2, ci[c]);
          sv[c] = valid && nrm_less<WIDE>(ck[c], ci[c], tau_k, tau_i);
          ns += sv[c] ? 1 : 0;
        }
        const int sincl = row_incl_scan(ns);
        const int stot = row_last(sincl);
        if (__ballot(stot > 0) == 0ull) continue;
        {
          int slot = sincl - ns;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (sv[c]) {
              L.Sk[slot] = ck[c];
              if constexpr (WIDE) L.Si[slot] = ci[c];
              ++slot;
            }
        }
        O3DS_WAVE_SYNC();
        // ---- rank survivors and kept keys against each other
        unsigned long long kk[KPL];
        int ki[KPL], rk[KPL];
        bool kv[KPL];
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
          const int idx = l + 16 * u;
          kv[u] = idx < cnt;
          kk[u] = L.Kk[idx < KMAX ? idx : 0];
          ki[u] = 0;
          if constexpr (WIDE) ki[u] = L.Ki[idx < KMAX ? idx : 0];
          rk[u] = idx;
        }
        int rs[4] = {0, 0, 0, 0};
#ifdef O3DS_NRM_CHECK
        int eqs[4] = {0, 0, 0, 0};
#endif
        for (int jj = 0; __ballot(jj < stot) != 0ull; ++jj) {
          const bool in = jj < stot;
          const unsigned long long sk = L.Sk[jj];
          int si = 0;
          if constexpr (WIDE) si = L.Si[jj];
#pragma unroll
          for (int c = 0; c < 4; ++c) rs[c] += (in && nrm_less<WIDE>(sk, si, ck[c], ci[c])) ? 1 : 0;
#pragma unroll
          for (int u = 0; u < KPL; ++u) rk[u] += (in && nrm_less<WIDE>(sk, si, kk[u], ki[u])) ? 1 : 0;
#ifdef O3DS_NRM_CHECK
#pragma unroll
          for (int c = 0; c < 4; ++c) eqs[c] += (in && sk == ck[c] && si == (WIDE ? ci[c] : 0)) ? 1 : 0;
#pragma unroll
          for (int u = 0; u < KPL; ++u)
            if (in && kv[u] && sk == kk[u] && si == ki[u]) O3DS_NRM_BAD(6);
#endif
        }
#ifdef O3DS_NRM_CHECK
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (sv[c] && eqs[c] != 1) {
            if (atomicAdd(&g_nrm_dbg[5], 1u) == 0u) {
              g_nrm_info[0] = ring, g_nrm_info[1] = tbase, g_nrm_info[2] = f0, g_nrm_info[3] = T, g_nrm_info[4] = stot, g_nrm_info[5] = cnt;
              g_nrm_info[6] = eqs[c], g_nrm_info[7] = (int)(ck[c] & 0xffffffffu), g_nrm_info[8] = l, g_nrm_info[9] = c, g_nrm_info[10] = (int)j;
              g_nrm_info[11] = ix, g_nrm_info[12] = iy, g_nrm_info[13] = iz, g_nrm_info[14] = s0, g_nrm_info[15] = e0;
              g_nrm_info[15] = lane;
              for (int a = 0; a < 64; ++a)
                for (int b = 0; b < 4; ++b) g_nrm_segs[4 * a + b] = s_dbg[a][b];
              g_nrm_q[0] = fx, g_nrm_q[1] = fy, g_nrm_q[2] = fz, g_nrm_q[3] = worst, g_nrm_q[4] = (double)qx, g_nrm_q[5] = (double)qy, g_nrm_q[6] = (double)qz;
            }
          }
#endif
        for (int jj = 0; __ballot(jj < cnt) != 0ull; ++jj) {
          const bool in = jj < cnt;
          const unsigned long long sk = L.Kk[jj];
          int si = 0;
          if constexpr (WIDE) si = L.Ki[jj];
#pragma unroll
          for (int c = 0; c < 4; ++c) rs[c] += (in && nrm_less<WIDE>(sk, si, ck[c], ci[c])) ? 1 : 0;
        }
        O3DS_WAVE_SYNC();
        if (stot > 0) {
#pragma unroll
          for (int u = 0; u < KPL; ++u)
            if (kv[u] && rk[u] < max_nn) {
              L.Kk[rk[u]] = kk[u];
              if constexpr (WIDE) L.Ki[rk[u]] = ki[u];
            }
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (sv[c] && rs[c] < max_nn) {
              L.Kk[rs[c]] = ck[c];
              if constexpr (WIDE) L.Ki[rs[c]] = ci[c];
            }
        }
        O3DS_WAVE_SYNC();
        cnt = min(cnt + stot, max_nn);
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
        if (T != __shfl(incl, (lane & 48) | 15)) O3DS_NRM_BAD(4);
#endif
        if (cnt == max_nn) {  // list full: the bound becomes the max_nn-th key
          tau_k = L.Kk[max_nn - 1];
          if constexpr (WIDE) {
            tau_i = L.Ki[max_nn - 1];
            worst = __longlong_as_double((long long)tau_k);
          } else {
            tau_i = 0;
            worst = (double)__uint_as_float((unsigned int)(tau_k >> 32));
          }
        }
      }
      // ---- next round / next ring / done
      if (active) {
        tbase += 16;
        if (tbase >= ntask) {
          ring += 1;
          tbase = 0;
          ntask = (2 * ring + 1) * (2 * ring + 1);
          if (ring > rmax_cells) {
            active = false;
          } else {
            const double lb = g.cell * ((double)(ring - 1) + mf) * (1.0 - 1e-6);
            if (worst <= lb * lb) active = false;  // the max_nn-th best (or r^2) already lies inside the searched block
          }
        }
      }
      O3DS_WAVE_SYNC();  // the segment table is rewritten by the next round
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
        L.X[idx][0] = tpt.x;
        L.X[idx][1] = tpt.y;
        L.X[idx][2] = tpt.z;
        L.X[idx][3] = (R)1;
      }
    }
    O3DS_WAVE_SYNC();
    {
      // lane t < 9 sums term t = {x y z xx xy xz yy yz zz}[t] as X[.][ia] * X[.][ib] (x = x * 1 exactly)
      const int ia = l < 3 ? l : (l < 6 ? 0 : (l < 8 ? 1 : 2));
      const int ib = l < 3 ? 3 : (l < 6 ? l - 3 : (l < 8 ? l - 5 : 2));
      double acc = 0.0;
      if (l < 9) {
#pragma clang fp contract(off)
        for (int jj = 0; jj < cnt; ++jj) acc += (double)L.X[jj][ia] * (double)L.X[jj][ib];
        s_sum[it * 4 + grp][l] = acc;
      }
      if (l == 0) s_cnt[it * 4 + grp] = cnt;
    }
    O3DS_WAVE_SYNC();
  }

  // ---- covariance, eigenvector of the smallest eigenvalue, normalise, orient: one lane per point
  if (lane < PW && base + (size_t)lane < n) {
    const P4 q = sp[base + (size_t)lane];
    const int k = s_cnt[lane];
    double cov[6] = {1, 0, 0, 1, 0, 1};
    if (k >= 3) {
      double s[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) s[t] = s_sum[lane][t];
      det::cov_from_cumulants(s, k, cov);
    }
    double nv[3];
    det::fast_eigen3x3_min(cov, nv);
    det::normalize_orient(nv, (double)q.x, (double)q.y, (double)q.z);
    P4 o;
    o.x = (R)nv[0];
    o.y = (R)nv[1];
    o.z = (R)nv[2];
    o.i = 0;
    out_nrm[(size_t)q.i] = o;
  }
}

}  // namespace o3ds
