// normals_select_kernel.hpp -- the f32-storage instantiation of the hybrid neighbourhood search behind [O3D] EstimateNormals
// (call site CloudRegistration.cpp:49-56): the SAME neighbour set as normals_kernel (normals_kernel.hpp) -- the max_nn smallest of the
// points with d2 < r^2 in the total order (f32 d2, original index) -- found by SELECTION instead of by keeping a sorted list, and the
// nine cumulants summed order-independently (the hi / lo split of icp_kernels.hpp's split_exact), so the result is still a function of
// the cloud alone.
//
//   * FOUR lanes (one DPP quad) per point, sixteen points per wavefront, one wavefront per workgroup.  Queries are taken in cell order,
//     so the sixteen points of a wavefront walk the same few cells.
//   * Candidates are streamed row by row (a row of the grid = one contiguous range of the cell-sorted cloud).  A lane looks for its next
//     row the current bound does not rule out (arithmetic only) and fetches the row's cell_start pair(s); the quad then walks the (up to
//     eight) segments of its four lanes one after the other, sixteen consecutive candidates per step (four loads in flight per lane).
//     A candidate whose key (d2 bits << 32 | original index) lies below the bound is appended to the point's list in LDS (positions from
//     a quad prefix sum).
//   * SELECT -- when every quad of the wavefront has finished its ring, or when a list has no room for another step: a list is reduced
//     to exactly the max_nn smallest keys.  One pass over the entries (a lane owns entries l, l + 4, ... and holds them in registers)
//     builds a 16-bin histogram of d2 in two packed 64-bit counters per lane; the bin that holds the max_nn-th key follows from the quad's
//     sum; entries in lower bins are in, entries in higher bins are out, and only the few entries of the boundary bin are ranked against
//     each other (full 64-bit keys, so ties in d2 go to the smaller original index).  The bound becomes the max_nn-th key.  No sorted
//     list, no rank matrix, no per-candidate sweep.
//   * The ring walk and its termination rule are normals_kernel's (the searched block must contain the ball of the max_nn-th distance).
//   * Cumulants: every term x * y of f32 coordinates is exact in binary64; the terms are split into hi + lo multiples of per-term quanta
//     so that their sums are exact and hence independent of the order the set is visited in.  Covariance, eigenvector, normalisation
//     and orientation are normals_finish_kernel, unchanged.
// f64 storage keeps normals_kernel (sorted list, sequential sums: the oracle's arithmetic bit for bit).
#pragma once
#include "normals_kernel.hpp"

namespace o3ds {

// ---- quad (4-lane) DPP primitives (quad_perm controls); every lane of the wavefront must execute them -----------------------------------
template <int CTRL>
__device__ __forceinline__ int quad_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int quad_sum(int v) {
  v += quad_dpp<0xB1>(v);  // quad_perm(1,0,3,2): lane ^ 1
  v += quad_dpp<0x4E>(v);  // quad_perm(2,3,0,1): lane ^ 2
  return v;
}
__device__ __forceinline__ int quad_max(int v) {
  v = max(v, quad_dpp<0xB1>(v));
  v = max(v, quad_dpp<0x4E>(v));
  return v;
}
__device__ __forceinline__ int quad_incl_scan(int v, int l) {  // inclusive prefix sum over the quad; l = lane & 3
  const int a = quad_dpp<0x90>(v);  // quad_perm(0,0,1,2): lane - 1 (lane 0 reads itself: masked)
  v += l >= 1 ? a : 0;
  const int b = quad_dpp<0x44>(v);  // quad_perm(0,1,0,1): lane - 2 (lanes 0, 1 masked)
  v += l >= 2 ? b : 0;
  return v;
}
template <int K>
__device__ __forceinline__ int quad_bcast(int v) {  // the value of lane K of the quad
  static_assert(K >= 0 && K < 4, "lane of a quad");
  return quad_dpp<K * 0x55>(v);  // quad_perm(K,K,K,K)
}
__device__ __forceinline__ float quad_maxf(float v) {
  v = fmaxf(v, __int_as_float(quad_dpp<0xB1>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(quad_dpp<0x4E>(__float_as_int(v))));
  return v;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long quad_dpp64(unsigned long long v) {
  const unsigned int lo = (unsigned int)quad_dpp<CTRL>((int)(unsigned int)v), hi = (unsigned int)quad_dpp<CTRL>((int)(unsigned int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long quad_sum64_bytes(unsigned long long v) {  // packed 8-bit counters: no carries while every total < 256
  v += quad_dpp64<0xB1>(v);
  v += quad_dpp64<0x4E>(v);
  return v;
}
__device__ __forceinline__ unsigned long long quad_max64(unsigned long long v) {
  unsigned long long o = quad_dpp64<0xB1>(v);
  v = o > v ? o : v;
  o = quad_dpp64<0x4E>(v);
  return o > v ? o : v;
}
__device__ __forceinline__ double quad_sumd(double v) {  // exact when the addends are multiples of one quantum and the total stays below 2^53 quanta
  v += __longlong_as_double((long long)quad_dpp64<0xB1>((unsigned long long)__double_as_longlong(v)));
  v += __longlong_as_double((long long)quad_dpp64<0x4E>((unsigned long long)__double_as_longlong(v)));
  return v;
}

// v is added to hi as a multiple of the term's quantum q ((v + c) - c with c = 1.5 * 2^52 q rounds to a multiple of q, |v| < 2^51 q) and
// to lo as the exact remainder rounded to a multiple of q * 2^-41 (icp_kernels.hpp split_exact)
__device__ __forceinline__ void nrm_split_add(double v, double c_hi, double c_lo, double& hi, double& lo) {
#pragma clang fp contract(off)
  const double h = (v + c_hi) - c_hi;
  const double r = v - h;
  hi += h;
  lo += (r + c_lo) - c_lo;
}

constexpr int kSelCap = 64;              // list entries per point; a lane owns entries l, l + 4, ...
constexpr int kSelOwn = kSelCap / 4;     // ... at most this many
constexpr int kSelStride = kSelCap + 4;  // in 8-byte slots, = 4 mod 32: the sixteen lists of a wavefront start 4 slots apart modulo the LDS banks
constexpr int kSelMaxNN = kSelCap - 16;  // a step appends up to 16 entries to a list that SELECT left with max_nn
constexpr unsigned long long kSelHole = ~0ull;  // an entry SELECT dropped without moving the others

__global__ __launch_bounds__(64) void normals_select_kernel(const P4f* __restrict__ pts /* original order */, size_t n, GridDev g,
                                                            const P4f* __restrict__ sp /* sorted by cell */, double radius, int max_nn, int rmax_cells,
                                                            double* __restrict__ out_sums /* [n][9], cell order */, int* __restrict__ out_cnt /* [n] */) {
  __shared__ unsigned long long s_list[16 * kSelStride];
  __shared__ int2 s_seg[16][8];  // the segments the quad is walking: [start, end) in the sorted cloud
  const int lane = threadIdx.x, l = lane & 3, quad = lane >> 2;
  unsigned long long* const L = s_list + quad * kSelStride;
  const size_t j = (size_t)blockIdx.x * 16 + quad;
  const bool have = j < n;
  auto gap = [](int d, double f) { return d > 0 ? (double)d - f : d < 0 ? f - (double)(d + 1) : 0.0; };

  const int* __restrict__ cs = g.cell_start;
  const double cell2 = g.cell * g.cell * (1.0 - 2e-6);
  const P4f q = sp[have ? j : 0];
  const float qx = q.x, qy = q.y, qz = q.z;
  const double fx = ((double)qx - g.ox) * g.inv_cell, fy = ((double)qy - g.oy) * g.inv_cell, fz = ((double)qz - g.oz) * g.inv_cell;
  const int ix = (int)floor(fx), iy = (int)floor(fy), iz = (int)floor(fz);
  const double frx = fx - floor(fx), fry = fy - floor(fy), frz = fz - floor(fz);
  const double mf = fmin(fmin(fmin(frx, 1.0 - frx), fmin(fry, 1.0 - fry)), fmin(frz, 1.0 - frz));

  // quad state (the same value in its four lanes)
  const float r2 = (float)(radius * radius);
  unsigned long long tau = nrm_key(r2, 0);  // keep a candidate iff its key < tau: d2 < r^2 until max_nn are known, then key <= the max_nn-th
  double worst = (double)r2;
  int ring = 1, ntask = 9;  // ring 1 = the whole 3x3x3 block
  float inv_w = 1.0f / 3.0f;
  int nlist = 0;            // slots of the list in use (holes included)
  int nvalid = 0;           // ... of which hold an entry
  bool dirty = false;       // entries were appended since the last SELECT
  enum { FIND = 0, WALK = 1, DONE = 2 };
  int state = have ? FIND : DONE;
  int seg = 8, p0 = 0, pe = 0;  // WALK: segment number, next candidate, end of the segment
  // lane state: its cursor over the rows t = l, l + 4, ... of the ring
  int tnext = l;

  for (;;) {
    bool want_select = false;  // quad-uniform
    // ---- FIND: every lane moves to its next row that the current bound does not rule out (arithmetic only).  When no lane of the quad
    // has one left the ring is done: SELECT if anything was appended, then the next ring unless the searched block already contains the
    // ball of `worst`.
    if (__ballot(state == FIND) != 0ull) {
      bool found = false;
      int row = 0, dzf = 0, dyf = 0;
      double left = 0.0;
      if (state == FIND) {
        while (tnext < ntask) {
          const int w = 2 * ring + 1;
          const int tz = (int)(((float)tnext + 0.5f) * inv_w);  // exact for these sizes
          const int dz = tz - ring, dy = tnext - tz * w - ring;
          const int z = iz + dz, y = iy + dy;
          tnext += 4;
          if ((unsigned)z >= (unsigned)g.nz || (unsigned)y >= (unsigned)g.ny) continue;
          const double gz = gap(dz, frz), gy = gap(dy, fry);
          left = worst - (gz * gz + gy * gy) * cell2;  // what the x-offset may still use; the bound only ever shrinks
          if (left <= 0.0) continue;
          row = (z * g.ny + y) * g.nx;
          dzf = dz, dyf = dy;
          found = true;
          break;
        }
      }
      const bool quad_found = quad_sum(found ? 1 : 0) != 0;
      int s0 = 0, e0 = 0, s1 = 0, e1 = 0;
      if (found) {
        const bool full = ring == 1 || dzf == -ring || dzf == ring || dyf == -ring || dyf == ring;
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
      if (state == FIND) {
        if (quad_found) {
          s_seg[quad][2 * l] = make_int2(s0, e0);
          s_seg[quad][2 * l + 1] = make_int2(s1, e1);
          state = WALK;
          seg = -1;
          p0 = pe = 0;
        } else if (dirty) {
          want_select = true;  // the ring is done; the list is reduced first, the ring advances in a later round
        } else {
          ring += 1;
          ntask = (2 * ring + 1) * (2 * ring + 1);
          inv_w = 1.0f / (float)(2 * ring + 1);
          tnext = l;
          if (ring > rmax_cells) {
            state = DONE;
          } else {
            const double lb = g.cell * ((double)(ring - 1) + mf) * (1.0 - 1e-6);
            if (worst <= lb * lb) state = DONE;  // the max_nn-th best (or r^2) already lies inside the searched block
          }
        }
      }
      O3DS_WAVE_SYNC();  // the segment table
    }
    if (__ballot(state != DONE) == 0ull) break;

    // ---- WALK: move to the next non-empty segment; a quad that ran out of segments looks for rows again
    for (;;) {
      const bool adv = state == WALK && p0 >= pe && seg < 8;
      if (__ballot(adv) == 0ull) break;
      if (adv) {
        ++seg;
        if (seg < 8) {
          const int2 se = s_seg[quad][seg];
          p0 = se.x, pe = se.y;
        }
      }
    }
    if (state == WALK && seg >= 8) state = FIND;
    const bool stepping = state == WALK;  // p0 < pe
    const bool no_room = stepping && nlist + 16 > kSelCap;
    // SELECT when a list is out of room, or when every quad that is still at work has finished its ring (so that one pass serves them all)
    if (__ballot(no_room) != 0ull || (__ballot(want_select) != 0ull && __ballot(stepping) == 0ull)) {
      // quad-uniform; every list that can be reduced is -- also one that only has holes to lose (a list whose boundary bin was large may
      // be out of room right after a SELECT: the second one finds every entry of the last bin in the set and just compacts)
      const bool go = nvalid >= max_nn && (dirty || nlist > nvalid);
      unsigned long long key[kSelOwn];
      float dmax = 0.0f;
#pragma unroll
      for (int e = 0; e < kSelOwn; ++e) {
        const int i = 4 * e + l;
        key[e] = (go && i < nlist) ? L[i] : kSelHole;
        if (key[e] != kSelHole) dmax = fmaxf(dmax, __uint_as_float((unsigned int)(key[e] >> 32)));
      }
      O3DS_WAVE_SYNC();  // every entry is in a register before the list is rewritten
      dmax = quad_maxf(dmax);
      const float scale = dmax > 0.0f ? 16.0f / dmax : 0.0f;  // bin = min(15, int(d2 * scale)): monotone in d2
      unsigned long long h0 = 0ull, h1 = 0ull;                // sixteen 8-bit counters
      int bin[kSelOwn];
#pragma unroll
      for (int e = 0; e < kSelOwn; ++e) {
        const float d2 = __uint_as_float((unsigned int)(key[e] >> 32));
        const int b = key[e] != kSelHole ? min(15, (int)(d2 * scale)) : 16;
        bin[e] = b;
        const unsigned long long inc = b < 16 ? 1ull << ((b & 7) * 8) : 0ull;
        h0 += (b & 8) ? 0ull : inc;
        h1 += (b & 8) ? inc : 0ull;
      }
      const unsigned long long t0 = quad_sum64_bytes(h0), t1 = quad_sum64_bytes(h1);  // at most 64 entries: no counter overflows
      // boundary bin: the first whose running total reaches max_nn; `below` = entries in the bins before it
      int bstar = 16, below = 0;
      {
        int run = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          const int nb = (int)(((b < 8 ? t0 : t1) >> ((b & 7) * 8)) & 0xffull);
          if (bstar == 16 && run + nb >= max_nn) {
            bstar = b;
            below = run;
          }
          run += nb;
        }
      }
      const int need = max_nn - below;  // how many of the boundary bin's entries belong to the set (>= 1 when go)
      // rewrite the list: the lower bins' entries first, the boundary bin's behind them, the rest dropped
      int n_in = 0, n_b = 0;
#pragma unroll
      for (int e = 0; e < kSelOwn; ++e) {
        n_in += bin[e] < bstar ? 1 : 0;
        n_b += bin[e] == bstar ? 1 : 0;
      }
      if (!go) n_in = n_b = 0;
      const int in_incl = quad_incl_scan(n_in, l), b_incl = quad_incl_scan(n_b, l);
      const int A = quad_bcast<3>(in_incl), B = quad_bcast<3>(b_incl);
      const int b_first = A + b_incl - n_b;  // where this lane's boundary entries start
      if (go) {
        int w_in = in_incl - n_in, w_b = b_first;
#pragma unroll
        for (int e = 0; e < kSelOwn; ++e) {
          if (bin[e] < bstar) L[w_in++] = key[e];
          if (bin[e] == bstar) L[w_b++] = key[e];
        }
      }
      O3DS_WAVE_SYNC();
      // rank the boundary entries against each other (full keys: ties in d2 go to the smaller original index); B is small unless many
      // candidates share one distance.  An entry that does not make it becomes a hole.
      unsigned int drop = 0u;
      unsigned long long kmax = 0ull;
      const bool ranking = go && B > need;
      for (int e = 0; __ballot(ranking && e < n_b) != 0ull; ++e) {
        const bool mine = ranking && e < n_b;
        const unsigned long long k = mine ? L[b_first + e] : 0ull;
        int rank = 0;
        for (int f = 0; __ballot(mine && f < B) != 0ull; ++f) {
          const unsigned long long o = (mine && f < B) ? L[A + f] : kSelHole;
          rank += o < k ? 1 : 0;
        }
        if (mine && rank >= need) drop |= 1u << e;
        if (mine && rank < need) kmax = k > kmax ? k : kmax;
      }
      if (go && !ranking) {  // every boundary entry belongs to the set
#pragma unroll
        for (int e = 0; e < kSelOwn; ++e)
          if (bin[e] == bstar) kmax = key[e] > kmax ? key[e] : kmax;
      }
      O3DS_WAVE_SYNC();
      for (int e = 0; __ballot(drop >> e) != 0ull; ++e)
        if ((drop >> e) & 1u) L[b_first + e] = kSelHole;
      kmax = quad_max64(kmax);
      if (go) {
        nlist = A + B;
        nvalid = max_nn;
        tau = kmax + 1ull;
        worst = (double)__uint_as_float((unsigned int)(kmax >> 32));
      }
      dirty = false;  // also where nothing could be reduced (fewer than max_nn entries): the ring may advance
      O3DS_WAVE_SYNC();
      continue;  // states are unchanged: the round is repeated with the reduced lists
    }

    // ---- one step of the walk: sixteen consecutive candidates of the segment, four per lane, all four loads in flight
    if (__ballot(stepping) != 0ull) {
      P4f c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 4 * u + l;
        c[u] = sp[(stepping && p < pe) ? p : 0];
      }
      unsigned long long k[4];
      bool acc[4];
      int na = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 4 * u + l;
        const float d2 = nrm_d2<float>(c[u].x, c[u].y, c[u].z, qx, qy, qz);
        k[u] = nrm_key(d2, c[u].i);
        acc[u] = stepping && p < pe && k[u] < tau;
        na += acc[u] ? 1 : 0;
      }
      const int incl = quad_incl_scan(na, l);
      const int tot = quad_bcast<3>(incl);
      int w = nlist + incl - na;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (acc[u]) L[w++] = k[u];
      nlist += tot;
      nvalid += tot;
      dirty = dirty || tot > 0;
      if (stepping) p0 += 16;
      O3DS_WAVE_SYNC();
    }
  }

  // ---- cumulants over the set, order-independently: every product of two f32 coordinates is exact in binary64; added as hi + lo
  // multiples of the term's quantum, the sums are exact
  double hi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    // quanta from a bound that depends on nothing but the point and the radius: every neighbour lies within `radius` of the query, so
    // |coordinate| < 2^ex; a coordinate is at most 2^44 quanta q = 2^(ex - 44), a product 2^44 quanta 2^(2 ex - 44): sums of <= 48 exact
    int ex = 0;
    (void)frexp(fmax(fmax(fabs((double)qx), fabs((double)qy)), fabs((double)qz)) + radius, &ex);
    const double c1h = ldexp(1.5, ex + 8), c1l = c1h * 4.547473508864641e-13;       // 1.5 * 2^52 * q, and that * 2^-41
    const double c2h = ldexp(1.5, 2 * ex + 8), c2l = c2h * 4.547473508864641e-13;
    for (int e0 = 0; __ballot(4 * e0 + l < nlist) != 0ull; e0 += 2) {
      P4f t[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = 4 * (e0 + u) + l;
        const unsigned long long kk = i < nlist ? L[i] : kSelHole;
        ok[u] = kk != kSelHole;
        const unsigned int oi = (unsigned int)(kk & 0xffffffffull);
        t[u] = pts[(ok[u] && (size_t)oi < n) ? oi : 0];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (ok[u]) {
          const double x = (double)t[u].x, y = (double)t[u].y, z = (double)t[u].z;
          nrm_split_add(x, c1h, c1l, hi[0], lo[0]);
          nrm_split_add(y, c1h, c1l, hi[1], lo[1]);
          nrm_split_add(z, c1h, c1l, hi[2], lo[2]);
          nrm_split_add(x * x, c2h, c2l, hi[3], lo[3]);
          nrm_split_add(x * y, c2h, c2l, hi[4], lo[4]);
          nrm_split_add(x * z, c2h, c2l, hi[5], lo[5]);
          nrm_split_add(y * y, c2h, c2l, hi[6], lo[6]);
          nrm_split_add(y * z, c2h, c2l, hi[7], lo[7]);
          nrm_split_add(z * z, c2h, c2l, hi[8], lo[8]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const double H = quad_sumd(hi[t]), Lo = quad_sumd(lo[t]);
    if (have && (t & 3) == l) out_sums[9 * j + t] = H + Lo;
  }
  if (have && l == 0) out_cnt[j] = nvalid;
}

}  // namespace o3ds
