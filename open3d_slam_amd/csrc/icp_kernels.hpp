// icp_kernels.hpp -- nearest-neighbour index build and the fused point-to-plane ICP pass for gfx950.
//
// Replaces, for the reference's registerClouds seam (CloudRegistration.cpp:44-48), the Open3D v0.15.1 routines
//   KDTreeFlann::SetGeometry              -> grid index build  (bbox -> cell count -> scan -> scatter)
//   GetRegistrationResultAndCorrespondences
//   + PointCloud::Transform
//   + TransformationEstimationPointToPlane::ComputeTransformation (ComputeJTJandJTr)
//                                         -> ONE kernel per pass: icp_accumulate_kernel
//   SolveJacobianSystemAndObtainExtrinsicMatrix + convergence test
//                                         -> icp_update_kernel (one workgroup, on device: no host round trip)
#pragma once
#include "common.hpp"

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
// per-block bounding boxes: out[block*6 + {0..2}] = min, {3..5} = max
template <typename P4>
__global__ __launch_bounds__(kBlock) void bbox_kernel(const P4* __restrict__ pts, size_t n, double* __restrict__ out) {
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
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

// counts[cell]++ ; cell_id[i] = cell
template <typename P4>
__global__ __launch_bounds__(kBlock) void cell_count_kernel(const P4* __restrict__ pts, size_t n, GridDev g, int* __restrict__ counts,
                                                            int* __restrict__ cell_id) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int c = cell_of(g, pts[i]);
    cell_id[i] = c;
    atomicAdd(&counts[c], 1);
  }
}

// 3-phase exclusive scan over `m` ints, 1024 elements per block (4 per thread)
constexpr int kScanPerBlock = 1024;
__global__ __launch_bounds__(kBlock) void scan_local_kernel(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ block_sums,
                                                            size_t m) {
  __shared__ int s_wave[kBlock / 64];
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (base + k < m) ? in[base + k] : 0;
  const int tsum = v[0] + v[1] + v[2] + v[3];
  // inclusive scan of tsum across the wave
  int x = tsum;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) s_wave[w] = x;
  __syncthreads();
  int woff = 0;
  for (int k = 0; k < w; ++k) woff += s_wave[k];
  int excl = woff + x - tsum;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < m) out[base + k] = excl;
    excl += v[k];
  }
  if (threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = woff + x;
}
// single block: exclusive scan of block sums in place (nb <= 1<<20 handled serially per thread chunk)
__global__ __launch_bounds__(kBlock) void scan_sums_kernel(int* __restrict__ sums, int nb) {
  __shared__ int s_tot[kBlock];
  const int per = (nb + kBlock - 1) / kBlock;
  const int b = threadIdx.x * per, e = min(nb, b + per);
  int t = 0;
  for (int i = b; i < e; ++i) t += sums[i];
  s_tot[threadIdx.x] = t;
  __syncthreads();
  int off = 0;
  for (int k = 0; k < (int)threadIdx.x; ++k) off += s_tot[k];
  for (int i = b; i < e; ++i) {
    const int v = sums[i];
    sums[i] = off;
    off += v;
  }
}
__global__ __launch_bounds__(kBlock) void scan_add_kernel(int* __restrict__ out, const int* __restrict__ sums, size_t m) {
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 4;
  const int off = sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + k < m) out[base + k] += off;
}

// sorted[pos] = pts[i] (+ normals), pos = cell_start[cell] + cursor[cell]++
template <typename P4>
__global__ __launch_bounds__(kBlock) void scatter_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, size_t n,
                                                         const int* __restrict__ cell_id, const int* __restrict__ cell_start,
                                                         int* __restrict__ cursor, P4* __restrict__ spts, P4* __restrict__ snrm) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int c = cell_id[i];
    const int pos = cell_start[c] + atomicAdd(&cursor[c], 1);
    P4 p = pts[i];
    p.i = (typename Scalar<P4>::index)i;
    spts[pos] = p;
    if (nrm) snrm[pos] = nrm[i];
  }
}

// ----------------------------------------------------------------------------------------------
// exact 1-NN within radius on the grid ([O3D] KDTreeFlann::SearchHybrid(q, r, 1))
// ----------------------------------------------------------------------------------------------
template <typename P4>
struct NNResult {
  typename Scalar<P4>::type d2;
  int pos;  // position in the sorted target arrays, -1 = none
  typename Scalar<P4>::index idx;
};

template <typename P4, bool kCrop>
__device__ __forceinline__ void scan_range(const P4* __restrict__ tp, int s, int e, typename Scalar<P4>::type qx,
                                           typename Scalar<P4>::type qy, typename Scalar<P4>::type qz, const CropDev& crop,
                                           NNResult<P4>& best) {
  using R = typename Scalar<P4>::type;
  for (int p = s; p < e; ++p) {
    const P4 t = tp[p];
    const R dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
    const R d2 = dx * dx + dy * dy + dz * dz;
    // strict d2 < best (initially r^2) ; ties broken towards the smaller original index => order-independent result
    if (d2 < best.d2 || (d2 == best.d2 && t.i < best.idx)) {
      if (!kCrop || crop_contains(crop, (double)t.x, (double)t.y, (double)t.z)) {
        best.d2 = d2;
        best.pos = p;
        best.idx = t.i;
      }
    }
  }
}

template <typename P4, bool kCrop>
__device__ __forceinline__ NNResult<P4> nn_search(const GridDev& g, const P4* __restrict__ tp, typename Scalar<P4>::type qx,
                                                  typename Scalar<P4>::type qy, typename Scalar<P4>::type qz,
                                                  typename Scalar<P4>::type r2max, int rmax_cells, const CropDev& crop) {
  using R = typename Scalar<P4>::type;
  NNResult<P4> best;
  best.d2 = r2max;
  best.pos = -1;
  best.idx = -1;
  const double fx = ((double)qx - g.ox) * g.inv_cell, fy = ((double)qy - g.oy) * g.inv_cell, fz = ((double)qz - g.oz) * g.inv_cell;
  const double flx = floor(fx), fly = floor(fy), flz = floor(fz);
  // queries far outside the grid cannot have a neighbour within r: clamp so the int conversion is safe
  const double lim = 1.0e9;
  const int ix = (int)fmin(fmax(flx, -lim), lim), iy = (int)fmin(fmax(fly, -lim), lim), iz = (int)fmin(fmax(flz, -lim), lim);
  double mf = fmin(fx - flx, 1.0 - (fx - flx));
  mf = fmin(mf, fmin(fy - fly, 1.0 - (fy - fly)));
  mf = fmin(mf, fmin(fz - flz, 1.0 - (fz - flz)));
  const int* __restrict__ cs = g.cell_start;

  // ---- ring 1: the 3x3x3 block = 9 rows of up to 3 contiguous cells; row bounds are loaded up front
  {
    const int x0 = max(ix - 1, 0), x1 = min(ix + 1, g.nx - 1);
    int rs[9], re[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int y = iy + (k % 3) - 1, z = iz + (k / 3) - 1;
      const bool ok = x0 <= x1 && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz;
      const int row = ok ? (z * g.ny + y) * g.nx : 0;
      rs[k] = ok ? cs[row + x0] : 0;
      re[k] = ok ? cs[row + x1 + 1] : 0;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) scan_range<P4, kCrop>(tp, rs[k], re[k], qx, qy, qz, crop, best);
  }
  // ---- rings 2..rmax: only while a closer point could still hide outside the searched block
  for (int ring = 2; ring <= rmax_cells; ++ring) {
    const double lb = g.cell * ((double)(ring - 1) + mf) * (1.0 - 1e-6);
    if ((double)best.d2 <= lb * lb) break;
    for (int dz = -ring; dz <= ring; ++dz) {
      const int z = iz + dz;
      if ((unsigned)z >= (unsigned)g.nz) continue;
      for (int dy = -ring; dy <= ring; ++dy) {
        const int y = iy + dy;
        if ((unsigned)y >= (unsigned)g.ny) continue;
        const int row = (z * g.ny + y) * g.nx;
        const bool shell = (dz == -ring || dz == ring || dy == -ring || dy == ring);
        if (shell) {
          const int x0 = max(ix - ring, 0), x1 = min(ix + ring, g.nx - 1);
          if (x0 <= x1) scan_range<P4, kCrop>(tp, cs[row + x0], cs[row + x1 + 1], qx, qy, qz, crop, best);
        } else {
          const int xl = ix - ring, xr = ix + ring;
          if ((unsigned)xl < (unsigned)g.nx) scan_range<P4, kCrop>(tp, cs[row + xl], cs[row + xl + 1], qx, qy, qz, crop, best);
          if ((unsigned)xr < (unsigned)g.nx) scan_range<P4, kCrop>(tp, cs[row + xr], cs[row + xr + 1], qx, qy, qz, crop, best);
        }
      }
    }
  }
  return best;
}

// ----------------------------------------------------------------------------------------------
// one ICP pass: transform -> 1-NN -> residual/Jacobian -> block-reduced normal equations
// ----------------------------------------------------------------------------------------------
struct IcpPassArgs {
  const void* src;   // P4[n_src]
  size_t first, count;
  const void* tpts;  // sorted target points
  const void* tnrm;  // sorted target normals
  GridDev grid;
  CropDev crop;
  double r2max;
  int rmax_cells;
  const IcpStateDev* state;
  double* partials;  // [gridDim.x][kRec]
};

template <typename P4, bool kCrop>
__global__ __launch_bounds__(kBlock) void icp_accumulate_kernel(IcpPassArgs a) {
  using R = typename Scalar<P4>::type;
  if (a.state->done) return;  // device-side loop already terminated: keep the previous partials
  const P4* __restrict__ src = (const P4*)a.src;
  const P4* __restrict__ tp = (const P4*)a.tpts;
  const P4* __restrict__ tn = (const P4*)a.tnrm;
  const double* T = a.state->T;  // column-major
  const double t00 = T[0], t10 = T[1], t20 = T[2], t01 = T[4], t11 = T[5], t21 = T[6], t02 = T[8], t12 = T[9], t22 = T[10],
               t03 = T[12], t13 = T[13], t23 = T[14];
  double acc[30];
#pragma unroll
  for (int k = 0; k < 30; ++k) acc[k] = 0.0;

  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.count; i += (size_t)gridDim.x * kBlock) {
    const P4 s = src[a.first + i];
    // [O3D] PointCloud::Transform: rigid 4x4 (bottom row 0 0 0 1 for every pose the reference passes)
    const double px = t00 * (double)s.x + t01 * (double)s.y + t02 * (double)s.z + t03;
    const double py = t10 * (double)s.x + t11 * (double)s.y + t12 * (double)s.z + t13;
    const double pz = t20 * (double)s.x + t21 * (double)s.y + t22 * (double)s.z + t23;
    const NNResult<P4> nn = nn_search<P4, kCrop>(a.grid, tp, (R)px, (R)py, (R)pz, (R)a.r2max, a.rmax_cells, a.crop);
    if (nn.pos >= 0) {
      const P4 q = tp[nn.pos];
      const P4 nq = tn[nn.pos];
      const double dx = px - (double)q.x, dy = py - (double)q.y, dz = pz - (double)q.z;
      const double nx = (double)nq.x, ny = (double)nq.y, nz = (double)nq.z;
      const double r = dx * nx + dy * ny + dz * nz;  // (p - q) . n
      double J[6];
      J[0] = py * nz - pz * ny;  // p x n
      J[1] = pz * nx - px * nz;
      J[2] = px * ny - py * nx;
      J[3] = nx;
      J[4] = ny;
      J[5] = nz;
      int k = 0;
#pragma unroll
      for (int r0 = 0; r0 < 6; ++r0)
#pragma unroll
        for (int c0 = r0; c0 < 6; ++c0) acc[k++] += J[r0] * J[c0];
#pragma unroll
      for (int r0 = 0; r0 < 6; ++r0) acc[21 + r0] += J[r0] * r;
      acc[kRecR2] += r * r;
      acc[kRecCount] += 1.0;
      acc[kRecD2] += dx * dx + dy * dy + dz * dz;
    }
  }
  // wavefront reduction, then 4 waves -> 1 through LDS; fixed order => bitwise reproducible per launch geometry
  __shared__ double s_part[kBlock / 64][kRec];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 30; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) s_part[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kRec) {
    double v = 0.0;
    if (threadIdx.x < 30) {
      for (int k = 0; k < kBlock / 64; ++k) v += s_part[k][threadIdx.x];
    }
    a.partials[(size_t)blockIdx.x * kRec + threadIdx.x] = v;
  }
}

// sum the per-block partial records into one 32-double record (fixed order)
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int nblocks, double* s_rec /* LDS [8][32] */,
                                                double* rec_out /* LDS [32] */) {
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;  // 8 parts x 32 columns
  double v = 0.0;
  for (int b = part; b < nblocks; b += kBlock / 32) v += partials[(size_t)b * kRec + col];
  s_rec[part * kRec + col] = v;
  __syncthreads();
  if (threadIdx.x < kRec) {
    double t = 0.0;
    for (int k = 0; k < kBlock / 32; ++k) t += s_rec[k * kRec + threadIdx.x];
    rec_out[threadIdx.x] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void icp_reduce_kernel(const double* __restrict__ partials, int nblocks, const IcpStateDev* state,
                                                            double* __restrict__ record) {
  if (state->done) return;
  __shared__ double s_rec[(kBlock / 32) * kRec];
  __shared__ double s_out[kRec];
  reduce_partials(partials, nblocks, s_rec, s_out);
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

// [O3D] RegistrationICP loop body after the correspondence pass: convergence test, solve, T <- U*T.
__device__ inline void icp_step_from_record(const double* rec, IcpStateDev* st, unsigned long long n_src_total, int max_iter,
                                            double rel_fitness, double rel_rmse) {
  const double count = rec[kRecCount];
  const double fitness = count > 0.0 ? count / (double)n_src_total : 0.0;
  const double rmse = count > 0.0 ? sqrt(rec[kRecD2] / count) : 0.0;
  bool conv = false;
  if (st->pass > 0) conv = fabs(st->fitness - fitness) < rel_fitness && fabs(st->rmse - rmse) < rel_rmse;
  st->fitness = fitness;
  st->rmse = rmse;
  st->n_corr = (unsigned long long)(count + 0.5);
  st->pass += 1;
  if (conv) {
    st->converged = 1;
    st->done = 1;
    return;
  }
  if (st->iterations >= max_iter) {
    st->done = 1;
    return;
  }
  double U[16];
  if (count > 0.0) {
    double x[6];
    solve6_ldlt(rec, x);
    vector6_to_matrix4(x, U);
  } else {  // empty correspondence set => identity update
    for (int i = 0; i < 16; ++i) U[i] = (i % 5 == 0) ? 1.0 : 0.0;
  }
  double Tn[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += U[k * 4 + r] * st->T[c * 4 + k];
      Tn[c * 4 + r] = s;
    }
  for (int i = 0; i < 16; ++i) st->T[i] = Tn[i];
  st->iterations += 1;
}

// single-GPU path: reduce the partials and step, one workgroup
__global__ __launch_bounds__(kBlock) void icp_reduce_update_kernel(const double* __restrict__ partials, int nblocks, IcpStateDev* state,
                                                                   unsigned long long n_src_total, int max_iter, double rel_fitness,
                                                                   double rel_rmse) {
  if (state->done) return;
  __shared__ double s_rec[(kBlock / 32) * kRec];
  __shared__ double s_out[kRec];
  reduce_partials(partials, nblocks, s_rec, s_out);
  if (threadIdx.x == 0) icp_step_from_record(s_out, state, n_src_total, max_iter, rel_fitness, rel_rmse);
}

// sharded path: the record was all-reduced by the caller
__global__ void icp_update_kernel(const double* __restrict__ record, IcpStateDev* state, unsigned long long n_src_total, int max_iter,
                                  double rel_fitness, double rel_rmse) {
  if (state->done) return;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double rec[kRec];
    for (int i = 0; i < kRec; ++i) rec[i] = record[i];
    icp_step_from_record(rec, state, n_src_total, max_iter, rel_fitness, rel_rmse);
  }
}

}  // namespace o3ds
