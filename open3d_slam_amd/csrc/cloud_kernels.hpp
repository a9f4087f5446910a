// cloud_kernels.hpp -- device-cloud plumbing and the pre-processing / map-fusion kernels for gfx950.
//
//   pack/unpack          host fp64 AoS (std::vector<Eigen::Vector3d>) <-> device P4 records
//   crop                 CroppingVolume::crop                 croppers.cpp:76-106
//   transform            o3d_slam::transform                  helpers.cpp:273-305   ([O3D] PointCloud::Transform)
//   voxel keys / reduce  [O3D] VoxelDownSample (helpers.cpp:107-113) and voxelizeWithinCroppingVolume (helpers.cpp:115-183)
//   normals              [O3D] EstimateNormals(Hybrid) + NormalizeNormals + OrientNormalsTowardsCameraLocation
//                        (CloudRegistration.cpp:49-56)
#pragma once
#include "common.hpp"
#include "icp_kernels.hpp"

namespace o3ds {

// Everything in this header is streaming work whose results are compared with the reference's (and the oracle's) plain double
// arithmetic: a*b + c is two roundings there, so it is two roundings here -- no fused multiply-add (hipcc's default for device code is
// -ffp-contract=fast).  That is what makes o3d_slam::transform, the ray samples of the carving and every boundary decision of a cropper
// come out with the reference's bits and not merely close to them.  None of these kernels is arithmetic-bound.
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------- pack / unpack
template <typename P4, typename S = double>  // S: the scalar of the staged host array (float when the host side already narrowed it)
__global__ __launch_bounds__(kBlock) void pack_kernel(const S* __restrict__ xyz, size_t n, P4* __restrict__ out) {
  using R = typename Scalar<P4>::type;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    P4 p;
    p.x = (R)xyz[3 * i];
    p.y = (R)xyz[3 * i + 1];
    p.z = (R)xyz[3 * i + 2];
    p.i = (typename Scalar<P4>::index)i;
    out[i] = p;
  }
}
// sensor_msgs/PointCloud2-style records: n points of `step` bytes, float32 x / y / z at byte offsets ox / oy / oz
// (open3d_conversions.cpp:59-68 reads exactly these three fields and widens them to double)
template <typename P4>
__global__ __launch_bounds__(kBlock) void pack_strided_f32_kernel(const unsigned char* __restrict__ raw, size_t n, size_t step, size_t ox,
                                                                  size_t oy, size_t oz, P4* __restrict__ out) {
  using R = typename Scalar<P4>::type;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned char* rec = raw + i * step;
    float x, y, z;  // offsets are 4-byte aligned by the format; memcpy keeps it legal for packed layouts as well
    __builtin_memcpy(&x, rec + ox, 4);
    __builtin_memcpy(&y, rec + oy, 4);
    __builtin_memcpy(&z, rec + oz, 4);
    P4 p;
    p.x = (R)x;
    p.y = (R)y;
    p.z = (R)z;
    p.i = (typename Scalar<P4>::index)i;
    out[i] = p;
  }
}
// the way back (open3d_conversions.cpp:19-53 open3dToRos, and the float32 rows of a binary PCD): record i gets float32 x / y / z
// at byte offsets ox / oy / oz and, when `on` is a valid offset, the normal at on, on + 4, on + 8; the other bytes of the
// (pre-zeroed) record are left alone.  double -> float is the same round-to-nearest narrowing `*ros_pc2_x = point(0)` does.
constexpr size_t kNoField = ~(size_t)0;
// colour byte of a [0, 1] double: mode 0 = `(int)(255 * c)` stored to a uint8 (open3dToRos, open3d_conversions.cpp:41-43),
// mode 1 = clamp, scale, round ([O3D] utility::ColorToUint8, which the PCD writer uses)
__device__ __forceinline__ unsigned char color_byte(double c, int mode) {
  if (mode == 0) return (unsigned char)(int)(255.0 * c);
  return (unsigned char)rint(fmin(1.0, fmax(0.0, c)) * 255.0);
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void unpack_strided_f32_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm,
                                                                    const P4* __restrict__ col, size_t n, size_t step, size_t ox, size_t oy,
                                                                    size_t oz, size_t on, size_t oc, int color_mode,
                                                                    unsigned char* __restrict__ raw) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    unsigned char* rec = raw + i * step;
    const P4 p = pts[i];
    const float x = (float)p.x, y = (float)p.y, z = (float)p.z;
    __builtin_memcpy(rec + ox, &x, 4);
    __builtin_memcpy(rec + oy, &y, 4);
    __builtin_memcpy(rec + oz, &z, 4);
    if (on != kNoField && nrm) {
      const P4 q = nrm[i];
      const float a = (float)q.x, b = (float)q.y, c = (float)q.z;
      __builtin_memcpy(rec + on, &a, 4);
      __builtin_memcpy(rec + on + 4, &b, 4);
      __builtin_memcpy(rec + on + 8, &c, 4);
    }
    if (oc != kNoField && col) {  // the packed `rgb` field: bytes b, g, r, 0 (PointCloud2 sub-fields b / g / r at +0 / +1 / +2)
      const P4 c = col[i];
      rec[oc] = color_byte((double)c.z, color_mode);
      rec[oc + 1] = color_byte((double)c.y, color_mode);
      rec[oc + 2] = color_byte((double)c.x, color_mode);
      rec[oc + 3] = 0;
    }
  }
}
// colours the way rosToOpen3d reads them (open3d_conversions.cpp:70-86): kind 0 = an `rgb` field at byte offset `off` (r, g, b bytes
// at +2, +1, +0, each / 255.0); kind 1 = an `intensity` field read through a uint8 iterator, i.e. its FIRST byte, unscaled, three times
template <typename P4>
__global__ __launch_bounds__(kBlock) void colors_from_records_kernel(const unsigned char* __restrict__ raw, size_t n, size_t step, size_t off,
                                                                     int kind, P4* __restrict__ out) {
  using R = typename Scalar<P4>::type;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned char* f = raw + i * step + off;
    P4 c;
    if (kind == 0) {
      c.x = (R)((double)(int)f[2] / 255.0);
      c.y = (R)((double)(int)f[1] / 255.0);
      c.z = (R)((double)(int)f[0] / 255.0);
    } else {
      c.x = c.y = c.z = (R)(double)f[0];
    }
    c.i = 0;
    out[i] = c;
  }
}
template <typename P4, typename S = double>
__global__ __launch_bounds__(kBlock) void unpack_kernel(const P4* __restrict__ in, size_t n, S* __restrict__ xyz) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = in[i];
    xyz[3 * i] = (S)p.x;
    xyz[3 * i + 1] = (S)p.y;
    xyz[3 * i + 2] = (S)p.z;
  }
}

// ---------------------------------------------------------------------------------------------- stream compaction
// flags -> exclusive positions: reuse the 3-phase int scan of icp_kernels.hpp.
template <typename P4>
__global__ __launch_bounds__(kBlock) void crop_flag_kernel(const P4* __restrict__ pts, size_t n, CropDev crop, int* __restrict__ flags) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    flags[i] = crop_contains(crop, (double)p.x, (double)p.y, (double)p.z) ? 1 : 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[n] = 0;  // sentinel: the exclusive scan of n + 1 flags ends with the total
}
// stable compaction: out[pos[i]] = in[i] where flags[i]; `want` selects flag value 1 (inside) or 0 (outside, pos = i - pos[i])
template <typename P4>
__global__ __launch_bounds__(kBlock) void compact_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, size_t n,
                                                         const int* __restrict__ flags, const int* __restrict__ pos, int want,
                                                         P4* __restrict__ out_pts, P4* __restrict__ out_nrm) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    if (flags[i] != want) continue;
    const size_t o = want ? (size_t)pos[i] : i - (size_t)pos[i];
    P4 p = pts[i];
    p.i = (typename Scalar<P4>::index)o;
    out_pts[o] = p;
    if (nrm) out_nrm[o] = nrm[i];
  }
}
// ---------------------------------------------------------------------------------------------- [O3D] RandomDownSample on the device
// PointCloud::RandomDownSample(ratio) (call sites Odometry.cpp:29, ScanToMapRegistration.cpp:39) shuffles the indices 0..n-1 with an mt19937
// seeded from std::random_device and keeps the first k = (int)(ratio * n): a uniformly drawn k-subset that no two runs of the reference
// share.  On the host that is a 55 000-element shuffle per call and a wait for n in front of it (0.45 ms, twice per frame of the shipped
// configuration, the GPU idle meanwhile).  Here every index gets a 64-bit key from a counter-based generator -- key(i) = mix(seed + (i + 1) *
// golden), the splitmix64 finaliser: a bijection of distinct arguments, so the keys are DISTINCT -- and the k points with the smallest keys are
// kept, in cloud order ([O3D] SelectByIndex walks the cloud through a mask): the same distribution, reproducible from the seed, restated
// in numpy for the checker (oracle/pipeline.py draw_keys).  The k-th smallest key is found exactly without a sort: an 11-bit histogram of
// the keys' top bits names the bin that holds it, a second one inside that bin the next 11 bits, the handful of keys of that sub-bin are
// listed and ranked by one workgroup; n and k never leave the device (CountRef in, CountPub out).
constexpr int kDrawBits = 11, kDrawBins = 1 << kDrawBits, kDrawListCap = 2048;
struct DrawState {  // one per handle; hist1 / hist2 / list_n are zero between calls (draw_pick_kernel leaves them so)
  unsigned int hist1[kDrawBins], hist2[kDrawBins];
  unsigned long long list[kDrawListCap];
  unsigned int list_n;
  int k;                                   // points to keep
  unsigned int bin1, below1, bin2, below2; // the bin of the k-th smallest key at each level, and the keys below it
  unsigned long long thr;                  // k > 0: keys <= thr are kept
  int error;                               // sticky: the listed sub-bin overflowed (it holds n / 4 M keys on average)
};
__host__ __device__ __forceinline__ unsigned long long draw_key(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + (i + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// the bin that holds the key of rank `target` (1-based) and the number of keys in the bins before it; kBlock threads, result in every thread
__device__ __forceinline__ void draw_find_bin(const unsigned int* __restrict__ hist, unsigned int target, unsigned int* s_scan /* [kBlock + 2] */,
                                              unsigned int* bin, unsigned int* below) {
  constexpr int kPer = kDrawBins / kBlock;
  static_assert(kDrawBins % kBlock == 0, "bins split evenly over the threads");
  unsigned int mine[kPer], sum = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    mine[j] = hist[threadIdx.x * kPer + j];
    sum += mine[j];
  }
  s_scan[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < kBlock; d <<= 1) {  // inclusive scan of the threads' sums
    const unsigned int add = threadIdx.x >= (unsigned)d ? s_scan[threadIdx.x - d] : 0u;
    __syncthreads();
    s_scan[threadIdx.x] += add;
    __syncthreads();
  }
  const unsigned int incl = s_scan[threadIdx.x], excl = incl - sum;
  if (excl < target && target <= incl) {  // exactly one thread
    unsigned int run = excl;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (run < target && target <= run + mine[j]) {
        s_scan[kBlock] = threadIdx.x * kPer + j;
        s_scan[kBlock + 1] = run;
      }
      run += mine[j];
    }
  }
  __syncthreads();
  *bin = s_scan[kBlock];
  *below = s_scan[kBlock + 1];
  __syncthreads();
}
__global__ __launch_bounds__(kBlock) void draw_hist1_kernel(CountRef n_ref, unsigned long long seed, DrawState* __restrict__ st) {
  __shared__ unsigned int s_hist[kDrawBins];
  for (int b = threadIdx.x; b < kDrawBins; b += kBlock) s_hist[b] = 0u;
  __syncthreads();
  const size_t n = count_of(n_ref);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) atomicAdd(&s_hist[draw_key(seed, i) >> (64 - kDrawBits)], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < kDrawBins; b += kBlock)
    if (s_hist[b]) atomicAdd(&st->hist1[b], s_hist[b]);
}
__global__ __launch_bounds__(kBlock) void draw_hist2_kernel(CountRef n_ref, unsigned long long seed, double ratio, DrawState* __restrict__ st) {
  __shared__ unsigned int s_scan[kBlock + 2];
  const size_t n = count_of(n_ref);
  const int k = (int)(ratio * (double)n);  // [O3D] indices.resize((int)(sampling_ratio * points_.size()))
  if (k <= 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->k = 0;
    return;
  }
  unsigned int bin1, below1;
  draw_find_bin(st->hist1, (unsigned int)k, s_scan, &bin1, &below1);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->k = k, st->bin1 = bin1, st->below1 = below1;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned long long key = draw_key(seed, i);
    if ((unsigned int)(key >> (64 - kDrawBits)) == bin1) atomicAdd(&st->hist2[(key >> (64 - 2 * kDrawBits)) & (kDrawBins - 1)], 1u);
  }
}
__global__ __launch_bounds__(kBlock) void draw_collect_kernel(CountRef n_ref, unsigned long long seed, DrawState* __restrict__ st, unsigned int list_cap) {
  __shared__ unsigned int s_scan[kBlock + 2];
  const int k = st->k;
  if (k <= 0) return;
  const unsigned int bin1 = st->bin1, below1 = st->below1;
  unsigned int bin2, below2;
  draw_find_bin(st->hist2, (unsigned int)k - below1, s_scan, &bin2, &below2);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->bin2 = bin2, st->below2 = below2;
  const unsigned long long want = ((unsigned long long)bin1 << kDrawBits) | bin2;
  const size_t n = count_of(n_ref);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned long long key = draw_key(seed, i);
    if ((key >> (64 - 2 * kDrawBits)) == want) {
      const unsigned int at = atomicAdd(&st->list_n, 1u);
      if (at < list_cap) st->list[at] = key;
    }
  }
}
// one workgroup: the key of rank k among the listed ones is the threshold; the count goes where the consumers and the host look for it;
// the histograms are handed back empty
__global__ __launch_bounds__(kBlock) void draw_pick_kernel(DrawState* __restrict__ st, CountPub pub, int* __restrict__ host_error, unsigned int list_cap) {
  const int k = st->k;
  if (k > 0) {
    const unsigned int m = st->list_n, r = (unsigned int)k - st->below1 - st->below2;  // rank inside the list, 1-based
    if (m > list_cap || r == 0 || r > m) {  // more than 2 048 keys share 22 leading bits (their number averages n / 4 M): nothing
      if (threadIdx.x == 0) {                              // is kept, and the handle's next draw -- or whoever asks for this size -- is told
        st->error = 1, st->thr = 0ull, st->k = 0;
        __hip_atomic_store(host_error, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      for (unsigned int j = threadIdx.x; j < m; j += kBlock) {
        const unsigned long long mine = st->list[j];
        unsigned int smaller = 0;
        for (unsigned int x = 0; x < m; ++x) smaller += st->list[x] < mine ? 1u : 0u;
        if (smaller + 1 == r) st->thr = mine;  // (keys are distinct: exactly one)
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kDrawBins; b += kBlock) st->hist1[b] = 0u, st->hist2[b] = 0u;
  if (threadIdx.x == 0) {
    st->list_n = 0u;
    if (st->error) {  // the consumers' word says "no points"; the host's record carries a size no cloud has (resolve_count fails on it)
      *pub.dev = 0;
      if (pub.host) {
        pub.host[0] = -1;
        __hip_atomic_store(pub.host + 1, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      publish_count(pub, st->k);
    }
  }
}
// flags[i] = the point is kept; flags[n_bound] = 0 (sentinel of the scan).  n_bound >= the exact size.
__global__ __launch_bounds__(kBlock) void draw_flag_kernel(CountRef n_ref, unsigned long long seed, const DrawState* __restrict__ st, int* __restrict__ flags) {
  const size_t n = count_of(n_ref);
  const int k = st->k;
  const unsigned long long thr = st->thr;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i <= n_ref.n; i += (size_t)gridDim.x * kBlock)
    flags[i] = (i < n && k > 0 && draw_key(seed, i) <= thr) ? 1 : 0;
}

template <typename P4>
__global__ __launch_bounds__(kBlock) void gather_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, const uint32_t* __restrict__ idx,
                                                        size_t m, P4* __restrict__ out_pts, P4* __restrict__ out_nrm) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock) {
    P4 p = pts[idx[i]];
    p.i = (typename Scalar<P4>::index)i;
    out_pts[i] = p;
    if (nrm) out_nrm[i] = nrm[idx[i]];
  }
}

// ---------------------------------------------------------------------------------------------- rigid transform
struct Mat34 {
  double m[12];  // row-major 3x4
};
template <typename P4>
__global__ __launch_bounds__(kBlock) void transform_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, size_t n, Mat34 M,
                                                            double w0, double w1, double w2, double w3, P4* __restrict__ out_pts,
                                                            P4* __restrict__ out_nrm, size_t out_off) {
  using R = typename Scalar<P4>::type;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    const double w = w0 * x + w1 * y + w2 * z + w3;  // helpers.cpp:289-290: xyz / new_point(3)
    P4 o;
    o.x = (R)((M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3]) / w);
    o.y = (R)((M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7]) / w);
    o.z = (R)((M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11]) / w);
    o.i = (typename Scalar<P4>::index)(out_off + i);
    out_pts[out_off + i] = o;
    if (nrm) {
      const P4 q = nrm[i];
      const double a = (double)q.x, b = (double)q.y, c = (double)q.z;
      P4 on;
      on.x = (R)(M.m[0] * a + M.m[1] * b + M.m[2] * c);  // helpers.cpp:294-296: T * (n, 0)
      on.y = (R)(M.m[4] * a + M.m[5] * b + M.m[6] * c);
      on.z = (R)(M.m[8] * a + M.m[9] * b + M.m[10] * c);
      on.i = 0;
      out_nrm[out_off + i] = on;
    }
  }
}

// ---------------------------------------------------------------------------------------------- constant-velocity de-skew
// ConstantVelocityMotionCompensation::undistortInputPointCloud (MotionCompensation.cpp:64-139), in place: the phase of a point is
// its azimuth / 2 pi (1 - that for a clockwise sensor, 0 at azimuth exactly 0), its motion T(phase * D * v, Rz Ry Rx(phase * D * w)).
template <typename P4>
__global__ __launch_bounds__(kBlock) void undistort_kernel(P4* __restrict__ pts, size_t n, double vx, double vy, double vz, double wr, double wp,
                                                           double wy, double scan_duration, int clockwise) {
  using R = typename Scalar<P4>::type;
  const double kTwoPi = 6.283185307179586476925286766559;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    P4 p = pts[i];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    const double angle = atan2(y, x);
    const double wrapped = angle < 0.0 ? angle + kTwoPi : angle;
    double phase = 0.0;
    if (wrapped != 0.0) phase = clockwise ? 1.0 - wrapped / kTwoPi : wrapped / kTwoPi;
    const double s = phase * scan_duration;
    double sa, ca, sb, cb, sg, cg;
    sincos(s * wr, &sa, &ca);  // roll
    sincos(s * wp, &sb, &cb);  // pitch
    sincos(s * wy, &sg, &cg);  // yaw
    // R = Rz(yaw) Ry(pitch) Rx(roll) (fromRPY, math.cpp:32-37)
    const double r00 = cg * cb, r01 = cg * sb * sa - sg * ca, r02 = cg * sb * ca + sg * sa;
    const double r10 = sg * cb, r11 = sg * sb * sa + cg * ca, r12 = sg * sb * ca - cg * sa;
    const double r20 = -sb, r21 = cb * sa, r22 = cb * ca;
    p.x = (R)(r00 * x + r01 * y + r02 * z + s * vx);
    p.y = (R)(r10 * x + r11 * y + r12 * z + s * vy);
    p.z = (R)(r20 * x + r21 * y + r22 * z + s * vz);
    pts[i] = p;
  }
}

// ---------------------------------------------------------------------------------------------- voxel reduction
// Voxel keys are packed into one sortable u64: 21 bits per axis, biased by 2^20 (|k| < 2^20 voxels per axis;
// at the 0.02 m dense-map voxel that is +-20 km).  Points outside the crop volume get a pass-through key
// (top bit set | original index) so that one sort yields [voxelised inside points by key][pass-through in order]...
// the reference emits pass-through first; the host-visible order is fixed up at emit time.
constexpr unsigned long long kPassBit = 1ull << 63;

__host__ __device__ __forceinline__ unsigned long long pack_key(long long kx, long long ky, long long kz) {
  const unsigned long long bx = (unsigned long long)(kx + (1ll << 20)) & 0x1FFFFFull;
  const unsigned long long by = (unsigned long long)(ky + (1ll << 20)) & 0x1FFFFFull;
  const unsigned long long bz = (unsigned long long)(kz + (1ll << 20)) & 0x1FFFFFull;
  return (bz << 42) | (by << 21) | bx;
}

// mode 0: data-anchored ([O3D] VoxelDownSample: floor((p - origin)/v));  mode 1: world-anchored (floor(p * (1/v)))
// (Packing the voxel coordinates relative to a known box -- the cloud's bounding box, or the box around a bounded cropping volume --
// into ~30 bits and sorting those with Onesweep, 4 passes, instead of letting rocPRIM merge-sort the 63-bit keys (its choice below
// 1 M elements, ~30 launches for the 700 k-point map) was measured and dropped: a 20 us Onesweep pass plus the memset of its
// look-back state per pass made map_insert_scan 0.86 ms instead of 0.82 ms and voxel_down_sample 0.24 ms instead of 0.18 ms.)
template <typename P4>
__global__ __launch_bounds__(kBlock) void voxel_key_kernel(const P4* __restrict__ pts, size_t n, int mode, double ox, double oy, double oz, double v,
                                                           CropDev crop, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                                           int filter = 0 /* mode 0 only: points outside `crop` get pass-through keys too */) {
  const double inv = 1.0 / v;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    unsigned long long k;
    if ((mode == 1 || filter) && !crop_contains(crop, x, y, z)) {
      k = kPassBit | (unsigned long long)i;
    } else if (mode == 0) {
      k = pack_key((long long)floor((x - ox) / v), (long long)floor((y - oy) / v), (long long)floor((z - oz) / v));
    } else {
      k = pack_key((long long)floor(x * inv), (long long)floor(y * inv), (long long)floor(z * inv));
    }
    keys[i] = k;
    vals[i] = (uint32_t)i;
  }
}

// ---- the sorted (key, value) list of a map merge WITHOUT sorting the map ---------------------------------------------------------------
// voxelizeWithinCroppingVolume (helpers.cpp:115-183) of mapCloud_ += scan re-bins the whole map at every insertion.  The map the previous
// insertion left is [pass-through block | voxel block in ascending key order] and its voxel block is still sorted: the stable sort of
// all n = N + m keys is a MERGE of that block (minus what left the volume) with the few keys that are new to the volume -- the scan's
// points and pass-through points that re-entered -- followed by the outside points in index order.  The kernels below produce exactly
// the arrays rocPRIM's radix sort would (same keys, same tie order: ascending index), so everything after them is unchanged and the
// result is bit-identical; only the m new keys are sorted.
// two counters in one 64-bit scan: points outside the volume (low word) and kept voxel-block entries (high word); every entry is one of
// outside / kept / new, so the number of new entries in front of entry i is i minus the other two
constexpr unsigned long long kCntPass = 1ull, kCntV = 1ull << 32, kCntMask = 0xffffffffull;
template <typename P4>
__global__ __launch_bounds__(kBlock) void merge_class_kernel(const P4* __restrict__ pts, size_t n, double v, CropDev crop, size_t np, size_t nv,
                                                             unsigned long long* __restrict__ keys, unsigned long long* __restrict__ cls,
                                                             int* __restrict__ unsorted) {
  const double inv = 1.0 / v;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += (size_t)gridDim.x * kBlock) {
    if (i == n) {
      cls[n] = 0;  // sentinel: the exclusive scan of n + 1 entries ends with the three totals
      continue;
    }
    const P4 p = pts[i];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    const unsigned long long vk = pack_key((long long)floor(x * inv), (long long)floor(y * inv), (long long)floor(z * inv));
    const bool inside = crop_contains(crop, x, y, z);
    const bool in_v = i >= np && i < np + nv;
    keys[i] = inside ? vk : (kPassBit | (unsigned long long)i);
    cls[i] = inside ? (in_v ? kCntV : 0ull) : kCntPass;
    if (in_v && i > np) {  // the voxel block must still be in key order (a mean that rounding put on the far side of a face breaks it)
      const P4 q = pts[i - 1];
      const unsigned long long pk = pack_key((long long)floor((double)q.x * inv), (long long)floor((double)q.y * inv), (long long)floor((double)q.z * inv));
      if (vk < pk) atomicOr(unsorted, 1);
    }
  }
}
// ranks -> the three lists: outside points straight into the tail of the sorted arrays, kept voxel-block entries and new entries apart
__global__ __launch_bounds__(kBlock) void merge_split_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ rank,
                                                             size_t n, size_t np, size_t nv, unsigned long long* __restrict__ k_sorted,
                                                             uint32_t* __restrict__ v_sorted, unsigned long long* __restrict__ vk, uint32_t* __restrict__ vv,
                                                             unsigned long long* __restrict__ xk, uint32_t* __restrict__ xv,
                                                             int* __restrict__ zero /* the gap counters merge_rank_kernel adds to */, size_t n_zero) {
  const unsigned long long tot = rank[n];
  const size_t n_in = n - (size_t)(tot & kCntMask);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_zero; i += (size_t)gridDim.x * kBlock) zero[i] = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned long long k = keys[i], r = rank[i];
    if (k & kPassBit) {
      const size_t o = n_in + (size_t)(r & kCntMask);
      k_sorted[o] = k;
      v_sorted[o] = (uint32_t)i;
    } else if (i >= np && i < np + nv) {
      const size_t o = (size_t)(r >> 32);
      vk[o] = k;
      vv[o] = (uint32_t)i;
    } else {
      const size_t o = i - (size_t)(r & kCntMask) - (size_t)(r >> 32);
      xk[o] = k;
      xv[o] = (uint32_t)i;
    }
  }
}
// a sorted new entry goes in front of the kept entries with the same key if it comes from the pass-through block (smaller index),
// behind them if it comes from the scan: xpos = number of kept entries in front of it; cnt[xpos] counts the insertions per gap
__global__ __launch_bounds__(kBlock) void merge_rank_kernel(const unsigned long long* __restrict__ xk, const uint32_t* __restrict__ xv, size_t nx,
                                                            const unsigned long long* __restrict__ vk, size_t nvin, size_t np,
                                                            int* __restrict__ xpos, int* __restrict__ cnt) {
  for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < nx; j += (size_t)gridDim.x * kBlock) {
    const unsigned long long k = xk[j];
    const bool before = (size_t)xv[j] < np;
    size_t lo = 0, hi = nvin;
    while (lo < hi) {
      const size_t mid = (lo + hi) >> 1;
      const unsigned long long m = vk[mid];
      if (before ? m < k : m <= k)
        lo = mid + 1;
      else
        hi = mid;
    }
    xpos[j] = (int)lo;
    atomicAdd(&cnt[lo], 1);
  }
}
__global__ __launch_bounds__(kBlock) void merge_place_kernel(const unsigned long long* __restrict__ vk, const uint32_t* __restrict__ vv, size_t nvin,
                                                             const int* __restrict__ before /* exclusive scan of cnt, nvin + 2 entries */,
                                                             const unsigned long long* __restrict__ xk, const uint32_t* __restrict__ xv,
                                                             const int* __restrict__ xpos, size_t nx, unsigned long long* __restrict__ k_sorted,
                                                             uint32_t* __restrict__ v_sorted) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nvin + nx; i += (size_t)gridDim.x * kBlock) {
    if (i < nvin) {
      const size_t o = i + (size_t)before[i + 1];  // new entries with xpos <= i lie in front of kept entry i
      k_sorted[o] = vk[i];
      v_sorted[o] = vv[i];
    } else {
      const size_t j = i - nvin;
      const size_t o = j + (size_t)xpos[j];
      k_sorted[o] = xk[j];
      v_sorted[o] = xv[j];
    }
  }
}

// ---- sort of the (key, index) pairs that are NEW to the volume in a map merge (tens of thousands per inserted scan): hand-written, two
// kernels.  The pairs are unique, so ordering by (key, index) is the stable order by key of a list whose indices ascend.
//   sort_tile_kernel        one workgroup per tile of 256 pairs: every pair counts the smaller pairs of its tile (broadcast LDS reads) and
//                           writes itself to that place -- 256 compares per thread, no network, no barrier after the load
//   sort_merge_pass_kernel  4 runs of `width` pairs -> one run: every pair finds its rank in the three other runs by binary search (the
//                           runs sit in L2) and writes itself to its final place of the pass; four passes for up to 65 536 pairs
// Measured on the stream (60 k pairs per call, rocprofv3): tile sort 9.0 us; a pass 6.2 us 2-way (8 passes), 8.2 us 4-way (4), 11.6 us 8-way
// (3), 32 us 16-way (2: 154 VGPRs) -- 42 us per call 4-way, against 50 us for rocPRIM's radix sort of the same pairs (six merge-sort launches
// and a 16 us block sort).  A first version sorted tiles of 2048 with a bitonic network in LDS: 66 barrier-separated stages on 30 workgroups
// took 52 us per call by themselves (profiles/r03_rocprof_kernel_stats_stream_bitonic_tiles.txt).
constexpr int kSortTile = 256;
constexpr int kSortWays = 4;
static_assert(kSortTile == kBlock, "one pair per thread in the tile sort");
__device__ __forceinline__ bool kv_less(unsigned long long ka, uint32_t va, unsigned long long kb, uint32_t vb) {
  return ka < kb || (ka == kb && va < vb);
}
__global__ __launch_bounds__(kBlock) void sort_tile_kernel(const unsigned long long* __restrict__ kin, const uint32_t* __restrict__ vin, size_t n,
                                                           unsigned long long* __restrict__ kout, uint32_t* __restrict__ vout) {
  __shared__ unsigned long long sk[kSortTile];
  __shared__ uint32_t sv[kSortTile];
  const size_t base = (size_t)blockIdx.x * kSortTile, gi = base + threadIdx.x;
  const unsigned long long k = gi < n ? kin[gi] : ~0ull;  // padding sorts behind every real pair
  const uint32_t v = gi < n ? vin[gi] : 0xffffffffu;
  sk[threadIdx.x] = k;
  sv[threadIdx.x] = v;
  __syncthreads();
  int rank = 0;
#pragma unroll 8
  for (int j = 0; j < kSortTile; ++j) rank += kv_less(sk[j], sv[j], k, v) ? 1 : 0;
  if (gi < n) {
    kout[base + rank] = k;
    vout[base + rank] = v;
  }
}
// kWays runs of `width` pairs -> one run: a pair's place is its offset in its own run plus the number of smaller pairs in each of the other
// runs of its group -- kWays - 1 binary searches, advanced together so that their loads are in flight at the same time (the depth of the
// dependent chain is that of ONE search)
template <int kWays>
__global__ __launch_bounds__(kBlock) void sort_merge_pass_kernel(const unsigned long long* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t n,
                                                                 uint32_t width, unsigned long long* __restrict__ kout, uint32_t* __restrict__ vout) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint32_t run = i / width, start = run * width, gstart = (run / kWays) * kWays * width;
    const unsigned long long k = kin[i];
    const uint32_t v = vin[i];
    uint32_t lo[kWays], hi[kWays];
#pragma unroll
    for (int r = 0; r < kWays; ++r) {
      const unsigned long long rs = (unsigned long long)gstart + (unsigned long long)r * width;
      lo[r] = rs < n ? (uint32_t)rs : n;
      hi[r] = rs + width < n ? (uint32_t)(rs + width) : n;
      if (rs == start) hi[r] = lo[r];  // the pair's own run: nothing to search
    }
    for (;;) {
      bool any = false;
      unsigned long long km[kWays];
      uint32_t vm[kWays], mid[kWays];
#pragma unroll
      for (int r = 0; r < kWays; ++r) {
        mid[r] = lo[r] + ((hi[r] - lo[r]) >> 1);
        if (lo[r] < hi[r]) {
          km[r] = kin[mid[r]];
          vm[r] = vin[mid[r]];
          any = true;
        }
      }
      if (!any) break;
#pragma unroll
      for (int r = 0; r < kWays; ++r)
        if (lo[r] < hi[r]) {
          if (kv_less(km[r], vm[r], k, v))
            lo[r] = mid[r] + 1;
          else
            hi[r] = mid[r];
        }
    }
    uint32_t o = gstart + (i - start);
#pragma unroll
    for (int r = 0; r < kWays; ++r) {
      const unsigned long long rs = (unsigned long long)gstart + (unsigned long long)r * width;
      if (rs != start && rs < n) o += lo[r] - (uint32_t)rs;  // pairs of run r in front of this one (no two pairs are equal)
    }
    kout[o] = k;
    vout[o] = v;
  }
}

// after sorting (key, val): head[i] = 1 where a new segment starts
__global__ __launch_bounds__(kBlock) void segment_head_kernel(const unsigned long long* __restrict__ keys, size_t n, int* __restrict__ head) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) head[n] = 0;  // sentinel: the exclusive scan of n + 1 heads ends with the segment count
}
// the same, and a flag when the keys are NOT in ascending order (the caller took them for sorted: carve_t's partition path)
__global__ __launch_bounds__(kBlock) void segment_head_check_kernel(const unsigned long long* __restrict__ keys, size_t n, int* __restrict__ head,
                                                                    int* __restrict__ unsorted) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned long long k = keys[i], kp = i ? keys[i - 1] : 0ull;
    head[i] = (i == 0 || k != kp) ? 1 : 0;
    if (i && k < kp) atomicOr(unsorted, 1);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) head[n] = 0;
}
// "entry i lies inside the volume" (its key carries no pass-through bit) as the input of a scan; entry n is the zero sentinel
struct KeyInsideFlag {
  const unsigned long long* keys;
  size_t n;
  __device__ __forceinline__ int operator()(size_t i) const { return i < n && !(keys[i] & kPassBit); }
};
// stable partition of (key, index) by that flag: inside entries first, in their order, then the others in theirs -- for a map whose inside
// entries already lie in ascending key order (the voxel block a merge left) this IS the sorted list, ties in ascending index as a stable sort has them
__global__ __launch_bounds__(kBlock) void partition_keys_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ rank, size_t n,
                                                                unsigned long long* __restrict__ k_out, uint32_t* __restrict__ v_out) {
  const size_t n_in = (size_t)rank[n];
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned long long k = keys[i];
    const size_t r = (size_t)rank[i];
    const size_t o = (k & kPassBit) ? n_in + (i - r) : r;
    k_out[o] = k;
    v_out[o] = (uint32_t)i;
  }
}
// seg_start[seg_id] = i for every head (seg ids from the exclusive scan of head)
__global__ __launch_bounds__(kBlock) void segment_start_kernel(const int* __restrict__ head, const int* __restrict__ seg_id, size_t n,
                                                               int* __restrict__ seg_start) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    if (head[i]) seg_start[seg_id[i]] = (int)i;
}

// one thread per output segment: mean of points / normals in ascending original-index order (vals are sorted
// within a segment because the radix sort is stable and vals were emitted ascending) => deterministic sums.
// renorm: re-normalise the mean normal (helpers.cpp:172); skip_nan: ignore NaN normals (helpers.cpp:35-38)
template <typename P4>
__global__ __launch_bounds__(kBlock) void segment_mean_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm,
                                                              const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                              const int* __restrict__ seg_start, size_t n_seg, size_t n, int renorm,
                                                              size_t n_pass, P4* __restrict__ out_pts, P4* __restrict__ out_nrm,
                                                              int drop_pass = 0 /* emit the voxel means only (crop + VoxelDownSample in one) */) {
  using R = typename Scalar<P4>::type;
  (void)keys;  // segments are told apart by position (voxel segments sort first)
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < n_seg; s += (size_t)gridDim.x * kBlock) {
    const size_t b = (size_t)seg_start[s], e = (s + 1 < n_seg) ? (size_t)seg_start[s + 1] : n;
    // reference order: pass-through points first, voxel means after (helpers.cpp:151-181).  Sorted order has the
    // voxel segments first (n_seg - n_pass of them), pass-through segments (one point each, by original index) last.
    const size_t n_vox = n_seg - n_pass;
    const bool pass = s >= n_vox;
    if (pass && drop_pass) continue;
    const size_t o = drop_pass ? s : (pass ? (s - n_vox) : (n_pass + s));
    double sx = 0, sy = 0, sz = 0, nx = 0, ny = 0, nz = 0;
    for (size_t j = b; j < e; ++j) {
      const uint32_t id = vals[j];
      const P4 p = pts[id];
      sx += (double)p.x;
      sy += (double)p.y;
      sz += (double)p.z;
      if (nrm) {
        const P4 q = nrm[id];
        const double a = (double)q.x, bb = (double)q.y, c = (double)q.z;
        if (!renorm || !(isnan(a) || isnan(bb) || isnan(c))) {
          nx += a;
          ny += bb;
          nz += c;
        }
      }
    }
    const double cnt = (double)(e - b);
    P4 op;
    if (pass) {
      op = pts[vals[b]];
    } else {
      op.x = (R)(sx / cnt);
      op.y = (R)(sy / cnt);
      op.z = (R)(sz / cnt);
    }
    op.i = (typename Scalar<P4>::index)o;
    out_pts[o] = op;
    if (nrm) {
      P4 on;
      if (pass) {
        on = nrm[vals[b]];
      } else {
        nx /= cnt;
        ny /= cnt;
        nz /= cnt;
        if (renorm) {  // Eigen normalized(): v / sqrt(v . v), unchanged when the squared norm is 0 -- spelled as the oracle spells it
#pragma clang fp contract(off)
          const double z2 = (nx * nx + ny * ny) + nz * nz;
          if (z2 > 0.0) {
            const double nn = sqrt(z2);
            nx /= nn;
            ny /= nn;
            nz /= nn;
          }
        }
        on.x = (R)nx;
        on.y = (R)ny;
        on.z = (R)nz;
      }
      on.i = 0;
      out_nrm[o] = on;
    }
  }
}

// (normal estimation: normals_kernel.hpp + det_math.hpp)
// colours in the map merge: AccumulatedPoint::AddPoint ASSIGNS the colour (helpers.cpp:40-42; isValidColor, helpers.cpp:83-85, is
// true for every value), so a voxel ends up with the colour of its last point in cloud order; same output placement as
// segment_mean_kernel (pass-through points first, their own colour)
template <typename P4>
__global__ __launch_bounds__(kBlock) void segment_last_kernel(const P4* __restrict__ col, const unsigned long long* __restrict__ keys,
                                                              const uint32_t* __restrict__ vals, const int* __restrict__ seg_start, size_t n_seg,
                                                              size_t n, size_t n_pass, P4* __restrict__ out_col) {
  (void)keys;  // segments are told apart by position (voxel segments sort first)
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < n_seg; s += (size_t)gridDim.x * kBlock) {
    const size_t e = (s + 1 < n_seg) ? (size_t)seg_start[s + 1] : n;
    const size_t n_vox = n_seg - n_pass;
    const bool pass = s >= n_vox;  // the pass-through segments (one point each) sort after all voxel segments
    const size_t o = pass ? (s - n_vox) : (n_pass + s);
    out_col[o] = col[vals[e - 1]];  // vals ascend inside a segment (stable sort of ascending indices)
  }
}

// ---------------------------------------------------------------------------------------------- space carving (sparse map)
// Submap::carve -> getIdxsOfCarvedPoints (Submap.cpp:109-125, helpers.cpp:235-271).  The reference keeps a hash map voxel ->
// indices and lets every scan ray probe it every `voxel` metres under an `omp critical`; here the cropped map points are keyed
// (voxel_key_kernel, the same floor(p * (1/v)) key), sorted, their segments entered into an open-addressing table
// (key -> segment) and one thread per ray marches and probes; marking a point is an idempotent store, no atomics.
constexpr unsigned long long kEmptyKey = ~0ull;

// ---- VoxelDownSample without a sort, in three launches ------------------------------------------------------------------------------------
// [O3D] VoxelDownSample walks the points once and keeps an unordered_map voxel -> AccumulatedPoint, so a voxel's sum runs over its points
// in cloud order and (here, as in the oracle) the voxels come out in order of first appearance.  Three kernels, no sort, no size read-back:
//   vox_insert_kernel   every RUN of equal keys in consecutive lanes (a lidar scan lists its points along the scan lines: neighbours in the
//                       array are neighbours in space, two points share a voxel on average and mostly sit next to each other) enters an
//                       open-addressing table once: its first lane claims the slot, lowers the voxel's smallest point index, and pushes
//                       the run {first index, length} onto the voxel's run list (an exchange of the list head).  Which thread claims a slot
//                       or pushes first is a race; what is READ later is not: the set of keys, a voxel's smallest index, its set of runs.
//   vox_order_kernel    one pass over "this point opens its voxel" flags (a chained scan: every tile publishes its sum and looks back):
//                       the voxels numbered in order of first appearance, order[r] = table slot, the number of voxels to the device
//                       word and the pinned record the consumers of the cloud and the host read it from (CountPub).
//   vox_mean_kernel     one thread per voxel: its runs put in ascending order (they are few), the sums taken run by run -- i.e. in cloud
//                       order --, the slot handed back empty (the table is all 0xff between calls, no memset per call).
// Rounds 1-4 ran bbox, bbox-final, insert, two scan kernels, a wait for the size, number, gather, mean: 46 MB of counter traffic for a
// 4.4 MB job; the box now comes with the ingest (pack_strided_f32_box_kernel), the scan is one kernel, the member lists
// are run lists built by the insert itself.
struct alignas(32) VoxSlot {  // one record per slot: the four words an insertion touches share a cache line (as four arrays they were four
  unsigned long long key;     // lines per run: most of the 46 MB this step moved for a 4.4 MB job in round 4)
  unsigned int first;         // smallest point index of the voxel
  int head;                   // last run pushed (index of its first point); -1 = none
  unsigned int nrun;          // (number of runs) - 1
  unsigned int pad[3];
};
struct VoxTable {
  VoxSlot* s;            // [cap], all 0xff when not in use
  unsigned int* cursor;  // one word behind the table: next free entry of vox_mean_kernel's run-start scratch
  unsigned int mask;
};
constexpr size_t kVoxSlotBytes = sizeof(VoxSlot);

// scan_local_kernel (icp_kernels.hpp) with its input computed on the fly
template <typename T, typename Load>
__global__ __launch_bounds__(kBlock) void scan_local_fn_kernel(Load in, T* __restrict__ out, T* __restrict__ block_sums, size_t m, T* __restrict__ pub) {
  __shared__ T s_wave[kBlock / 64];
  const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 4;
  T v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (base + k < m) ? (T)in(base + k) : (T)0;
  const T tsum = v[0] + v[1] + v[2] + v[3];
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

// The g per-block boxes of bbox_kernel folded into one: {min x y z, max x y z} to device memory (the next kernel reads the grid origin from
// there) and to the handle's pinned block (the host reads it after the call's one synchronisation) -- so that a bounding box that only
// anchors a voxel grid costs a 4-us launch on the chain instead of a host round trip
__global__ __launch_bounds__(64) void bbox_final_kernel(const double* __restrict__ blocks, int g, double* __restrict__ dev_out, double* __restrict__ host_out) {
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int b = threadIdx.x; b < g; b += 64)
    for (int a = 0; a < 3; ++a) {
      mn[a] = fmin(mn[a], blocks[(size_t)b * 6 + a]);
      mx[a] = fmax(mx[a], blocks[(size_t)b * 6 + 3 + a]);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      mn[a] = fmin(mn[a], __shfl_xor(mn[a], m, 64));
      mx[a] = fmax(mx[a], __shfl_xor(mx[a], m, 64));
    }
  }
  if (threadIdx.x == 0)
    for (int a = 0; a < 3; ++a) {
      dev_out[a] = host_out[a] = mn[a];
      dev_out[3 + a] = host_out[3 + a] = mx[a];
    }
}

// one workgroup's bounding box into the cloud's box record: 6 atomics on order-preserving integer images of the doubles (min x y z, max x y z);
// all threads call it.
__device__ __forceinline__ void block_box_atomic(double mn[3], double mx[3], unsigned long long* __restrict__ box) {
  __shared__ double s_box[kBlock / 64][6];
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
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      s_box[w][a] = mn[a];
      s_box[w][3 + a] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = s_box[0][threadIdx.x];
    for (int k = 1; k < kBlock / 64; ++k) v = threadIdx.x < 3 ? fmin(v, s_box[k][threadIdx.x]) : fmax(v, s_box[k][threadIdx.x]);
    // (a workgroup without a point inside the volume adds the record's own initial values: 1e300 / -1e300)
    if (threadIdx.x < 3)
      atomicMin(box + threadIdx.x, order_bits(v));
    else
      atomicMax(box + threadIdx.x, order_bits(v));
  }
}
// the ingest of a PointCloud2-style buffer (pack_strided_f32_kernel) that also reduces the bounding box of the points inside `crop` -- the
// volume the handle's last crop + VoxelDownSample used: a lidar stream crops every scan with the same one -- so that the pre-processing
// of the scan starts with its grid anchor known instead of with two launches that read every point again
template <typename P4>
__global__ __launch_bounds__(kBlock) void pack_strided_f32_box_kernel(const unsigned char* __restrict__ raw, size_t n, size_t step, size_t ox,
                                                                      size_t oy, size_t oz, P4* __restrict__ out, CropDev crop,
                                                                      unsigned long long* __restrict__ box) {
  using R = typename Scalar<P4>::type;
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const unsigned char* rec = raw + i * step;
    float x, y, z;
    __builtin_memcpy(&x, rec + ox, 4);
    __builtin_memcpy(&y, rec + oy, 4);
    __builtin_memcpy(&z, rec + oz, 4);
    P4 p;
    p.x = (R)x;
    p.y = (R)y;
    p.z = (R)z;
    p.i = (typename Scalar<P4>::index)i;
    out[i] = p;
    const double dx = (double)p.x, dy = (double)p.y, dz = (double)p.z;
    if (crop_contains(crop, dx, dy, dz)) {  // fmin / fmax pass over NaN like bbox_kernel's
      mn[0] = fmin(mn[0], dx), mn[1] = fmin(mn[1], dy), mn[2] = fmin(mn[2], dz);
      mx[0] = fmax(mx[0], dx), mx[1] = fmax(mx[1], dy), mx[2] = fmax(mx[2], dz);
    }
  }
  block_box_atomic(mn, mx, box);
}
// the same reduction for a cloud that is already on the device (no ingest to ride on, or another volume than the ingest assumed)
template <typename P4>
__global__ __launch_bounds__(kBlock) void bbox_atomic_kernel(const P4* __restrict__ pts, size_t n, CropDev crop, unsigned long long* __restrict__ box) {
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    const double dx = (double)p.x, dy = (double)p.y, dz = (double)p.z;
    if (crop_contains(crop, dx, dy, dz)) {
      mn[0] = fmin(mn[0], dx), mn[1] = fmin(mn[1], dy), mn[2] = fmin(mn[2], dz);
      mx[0] = fmax(mx[0], dx), mx[1] = fmax(mx[1], dy), mx[2] = fmax(mx[2], dz);
    }
  }
  block_box_atomic(mn, mx, box);
}
// box record -> the pinned record the host reads it from: {stamp, pad, min x y z, max x y z} (the stamp last, system-scope release);
// re-arms the device record for its next use
__global__ __launch_bounds__(64) void box_publish_kernel(unsigned long long* __restrict__ box, double* __restrict__ host_box, int* __restrict__ host_seq,
                                                         int seq) {
  if (threadIdx.x < 6) {
    host_box[threadIdx.x] = order_value(box[threadIdx.x]);
    box[threadIdx.x] = order_bits(threadIdx.x < 3 ? 1e300 : -1e300);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// [O3D] voxel_min_bound = GetMinBound() - voxel_size * 0.5 arrives as (ox, oy, oz): the host knows the box before it launches
template <typename P4>
__global__ __launch_bounds__(kBlock) void vox_insert_kernel(const P4* __restrict__ pts, size_t n, double ox, double oy, double oz, double v,
                                                            CropDev crop, int filter, VoxTable t, int* __restrict__ lead_slot,
                                                            int* __restrict__ run_next, int* __restrict__ run_len) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) *t.cursor = 0u;  // (vox_mean_kernel, two launches on, hands out the scratch from zero)
  for (size_t i0 = (size_t)blockIdx.x * kBlock; i0 < n; i0 += (size_t)gridDim.x * kBlock) {  // whole wavefronts iterate together (the runs are found by lane)
    const size_t i = i0 + threadIdx.x;
    unsigned long long k = kEmptyKey;  // no entry: past the end, or outside the volume
    if (i < n) {
      const P4 p = pts[i];
      const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
      if (!filter || crop_contains(crop, x, y, z))
        k = pack_key((long long)floor((x - ox) / v), (long long)floor((y - oy) / v), (long long)floor((z - oz) / v));
    }
    const unsigned long long kp = __shfl_up(k, 1, 64);
    const bool lead = k != kEmptyKey && (lane == 0 || kp != k);
    const unsigned long long ends = __ballot(lane == 0 || kp != k);  // a run also ends where entries without a key begin
    int slot = -1;
    if (lead) {
      const unsigned long long above = lane == 63 ? 0ull : (ends >> (lane + 1));
      const int len = above ? (int)__builtin_ctzll(above) + 1 : 64 - lane;
      unsigned int sl = (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> 32) & t.mask;
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

// Chained scan (one pass, decoupled look-back) over the flags "point i is the smallest index of its voxel": tile b publishes its own sum,
// then the sum of everything up to and including itself; a tile adds up its predecessors' records from b - 1 downwards until it meets an
// inclusive one.  The records carry the number of this call (`gen`), so the buffer is never cleared: a record of an earlier call reads as
// "not there yet".  Tiles take their number from a ticket counter that runs on across calls (`ticket_base` = its value before this launch):
// a tile only ever waits for tiles whose ticket was drawn earlier, i.e. for workgroups that are already running.
constexpr int kVoxTile = 1024;
__device__ __forceinline__ unsigned long long tile_record(unsigned int gen, unsigned int kind /* 1 own sum, 2 inclusive */, unsigned int value) {
  return ((unsigned long long)(gen & 0x3fffffffu) << 34) | ((unsigned long long)kind << 32) | value;
}
__global__ __launch_bounds__(kBlock) void vox_order_kernel(const int* __restrict__ lead_slot, CountRef n_in, VoxTable t, unsigned long long* __restrict__ tiles,
                                                           unsigned int* __restrict__ ticket, unsigned int ticket_base, unsigned int gen,
                                                           int* __restrict__ order, CountPub pub) {
  __shared__ unsigned int s_tile;
  __shared__ int s_wave[kBlock / 64];
  __shared__ int s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
  __syncthreads();
  const unsigned int tile = s_tile, n_tiles = gridDim.x;
  const size_t n = count_of(n_in);
  const size_t base = (size_t)tile * kVoxTile + (size_t)threadIdx.x * 4;
  int sl[4], f[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sl[k] = base + k < n ? lead_slot[base + k] : -1;
    f[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (sl[k] >= 0) f[k] = t.s[sl[k]].first == (unsigned int)(base + k) ? 1 : 0;
  const int tsum = f[0] + f[1] + f[2] + f[3];
  int x = tsum;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) s_wave[w] = x;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; ++k) {
    if (k < w) woff += s_wave[k];
    total += s_wave[k];
  }
  if (w == 0) {  // the look-back: wavefront 0, 64 predecessors per round (a walk by one thread is as many dependent loads as there are tiles)
    if (lane == 0 && tile > 0) __hip_atomic_store(&tiles[tile], tile_record(gen, 1u, (unsigned int)total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int prefix = 0;
    for (int hi = (int)tile - 1; hi >= 0;) {
      const int b = hi - lane;
      unsigned long long r = 0;
      bool there = true;
      if (b >= 0) {
        r = __hip_atomic_load(&tiles[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        there = (unsigned int)(r >> 34) == (gen & 0x3fffffffu);
      }
      if (__ballot(!there) != 0ull) {  // a predecessor in the window has not published yet
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      // the nearest inclusive record in the window ends the walk: only the lanes in front of it (smaller lane = nearer tile) and itself count
      const unsigned long long incl = __ballot(b >= 0 && ((r >> 32) & 3u) == 2u);
      const int stop = incl ? (int)__builtin_ctzll(incl) : 64;
      int v = (b >= 0 && lane <= stop) ? (int)(unsigned int)r : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      prefix += v;
      if (incl) break;
      hi -= 64;
    }
    if (lane == 0) {
      __hip_atomic_store(&tiles[tile], tile_record(gen, 2u, (unsigned int)(prefix + total)), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
      if (tile == n_tiles - 1) publish_count(pub, prefix + total);  // the number of voxels = the size of the cloud being made
    }
  }
  __syncthreads();
  int r = s_prefix + woff + x - tsum;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (f[k]) order[r] = sl[k];
    r += f[k];
  }
}

// ascending order of a short list kept in global memory: insertion sort, heap sort beyond 16 entries (a voxel next to the sensor
// can hold a hundred returns)
__device__ inline void sort_indices(uint32_t* a, int k) {
  if (k <= 16) {
    for (int i = 1; i < k; ++i) {
      const uint32_t x = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > x) {
        a[j + 1] = a[j];
        --j;
      }
      a[j + 1] = x;
    }
    return;
  }
  auto sift = [&](int root, int end) {
    const uint32_t x = a[root];
    while (true) {
      int c = 2 * root + 1;
      if (c >= end) break;
      if (c + 1 < end && a[c + 1] > a[c]) ++c;
      if (a[c] <= x) break;
      a[root] = a[c];
      root = c;
    }
    a[root] = x;
  };
  for (int i = k / 2 - 1; i >= 0; --i) sift(i, k);
  for (int e = k - 1; e > 0; --e) {
    const uint32_t x = a[0];
    a[0] = a[e];
    a[e] = x;
    sift(0, e);
  }
}

// The runs of a voxel in ascending order of their first point WITHOUT a list in memory: the next one is looked for again each time (a voxel
// of a lidar scan has one or two runs; a list per voxel needs a place in a scratch array, i.e. a ticket from ONE counter -- a thousand
// wavefronts drawing tickets from one address took longer than everything else the kernel does).  INT_MAX when there is none.
constexpr int kInlineRuns = 6;  // voxels with more runs than this do get a sorted list in the scratch array (they are a fraction of a per cent)
__device__ __forceinline__ int next_run(const int* __restrict__ run_next, int head, int prev) {
  int best = 0x7fffffff;
  for (int node = head; node != -1; node = run_next[node])
    if (node > prev && node < best) best = node;
  return best;
}

// AccumulatedPoint::GetAveragePoint / GetAverageNormal / GetAverageColor ([O3D] PointCloud.cpp VoxelDownSample): sums in cloud order,
// divided by the count; normals are averaged, not re-normalised.  One thread per voxel r (the number of voxels comes from the device word
// vox_order_kernel left): the voxel's runs in ascending order (next_run; a voxel with many runs: copied into a piece of `starts` and sorted),
// summed run by run.  piece[r] = {head of the run list, number of runs} is left for a second pass over another attribute (colours).
template <typename P4>
__global__ __launch_bounds__(kBlock) void vox_mean_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, CountRef m_in, const int* __restrict__ order,
                                                          const int* __restrict__ run_next, const int* __restrict__ run_len, uint32_t* __restrict__ starts,
                                                          int2* __restrict__ piece, int attr_only, P4* __restrict__ out_pts, P4* __restrict__ out_nrm,
                                                          VoxTable t) {
  using R = typename Scalar<P4>::type;
  const size_t m = count_of(m_in);
  const int lane = threadIdx.x & 63;
  for (size_t r0 = (size_t)blockIdx.x * kBlock; r0 < m; r0 += (size_t)gridDim.x * kBlock) {  // whole wavefronts iterate together
    const size_t r = r0 + threadIdx.x;
    const bool have = r < m;
    int head = -1, k = 0;
    if (have) {
      if (attr_only) {
        head = piece[r].x, k = piece[r].y;
      } else {
        const int s = order[r];
        const VoxSlot v = t.s[s];
        k = (int)v.nrun + 1;
        head = v.head;
        // the table is done with this voxel: leave the slot as the 0xff fill left it, for the next call
        VoxSlot e;
        e.key = kEmptyKey, e.first = ~0u, e.head = -1, e.nrun = ~0u, e.pad[0] = e.pad[1] = e.pad[2] = ~0u;
        t.s[s] = e;
        if (piece) piece[r] = make_int2(head, k);
      }
    }
    // the few voxels with many runs: a sorted list in the scratch array, its place from one cursor atomic per wavefront that has any
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
    double sx = 0, sy = 0, sz = 0, nx = 0, ny = 0, nz = 0;
    int cnt = 0, prev = -1;
    for (int j = 0; j < k; ++j) {
      const int st = kk ? (int)starts[b + j] : next_run(run_next, head, prev);
      prev = st;
      const int len = run_len[st];
      for (int q = 0; q < len; q += 4) {  // four points' loads in flight together (clamped), added in order
        P4 pp[4], nn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = st + min(q + u, len - 1);
          pp[u] = pts[idx];
          nn[u] = nrm ? nrm[idx] : pp[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q + u < len) {
            sx += (double)pp[u].x;
            sy += (double)pp[u].y;
            sz += (double)pp[u].z;
            if (nrm) {
              nx += (double)nn[u].x;
              ny += (double)nn[u].y;
              nz += (double)nn[u].z;
            }
          }
      }
      cnt += len;
    }
    const double c = (double)cnt;
    P4 op;
    op.x = (R)(sx / c);
    op.y = (R)(sy / c);
    op.z = (R)(sz / c);
    op.i = (typename Scalar<P4>::index)r;
    out_pts[r] = op;
    if (nrm) {
      P4 on;
      on.x = (R)(nx / c);
      on.y = (R)(ny / c);
      on.z = (R)(nz / c);
      on.i = 0;
      out_nrm[r] = on;
    }
  }
}

// A ray of the carving takes up to 200 samples, nearly all of them in free space, and each used to cost a random 8-byte probe of the
// 16 MB voxel table: 26 M probes per scan, three quarters of them fetched as whole lines from beyond the L2 -- the kernel was bound by
// that traffic (its time follows the number of samples, putting eight probes in flight together changed nothing).  Free space is now
// answered by a 512 KB bit set that stays in the L2: one bit per hashed 4 x 4 x 4 block of voxels, set where the block holds a voxel of
// the table.  A clear bit proves the voxel absent (no probe); a set bit -- an occupied block, or another block with the same hash --
// sends the sample to the table as before.  The block of a key is the key with the two low bits of its three fields cleared.
constexpr int kCarveBitsLog2 = 22;
constexpr unsigned long long kCarveBlockMask = ~((3ull << 42) | (3ull << 21) | 3ull);
__device__ __forceinline__ unsigned int carve_block_bit(unsigned long long key) {
  return (unsigned int)(((key & kCarveBlockMask) * 0x9E3779B97F4A7C15ull) >> (64 - kCarveBitsLog2));
}

// keys of the map points inside the wide cropping volume, everything else gets the pass-through bit (never probed)
__global__ __launch_bounds__(kBlock) void carve_table_insert_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ seg_start,
                                                                    size_t n_seg, unsigned long long* __restrict__ tkey, int* __restrict__ tseg,
                                                                    unsigned int mask, unsigned int* __restrict__ block_bits) {
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < n_seg; s += (size_t)gridDim.x * kBlock) {
    const unsigned long long k = keys[seg_start[s]];
    if (k & kPassBit) continue;  // segments of points outside the volume
    const unsigned int bit = carve_block_bit(k);
    atomicOr(&block_bits[bit >> 5], 1u << (bit & 31u));
    unsigned int slot = (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> 32) & mask;
    while (true) {
      const unsigned long long prev = atomicCAS(&tkey[slot], kEmptyKey, k);
      if (prev == kEmptyKey || prev == k) {
        tseg[slot] = (int)s;
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

template <typename P4>
__global__ __launch_bounds__(kBlock) void carve_rays_kernel(const P4* __restrict__ scan, size_t n_scan, Mat34 M /* map <- sensor */,
                                                            double sx, double sy, double sz, double voxel, double max_len, double trunc,
                                                            double min_dot, const unsigned long long* __restrict__ tkey,
                                                            const int* __restrict__ tseg, unsigned int mask, const int* __restrict__ seg_start,
                                                            size_t n_seg, size_t n_sorted, const uint32_t* __restrict__ vals,
                                                            const P4* __restrict__ map_nrm, int* __restrict__ flags,
                                                            const unsigned int* __restrict__ block_bits) {
  const double inv = 1.0 / voxel;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_scan; i += (size_t)gridDim.x * kBlock) {
    const P4 q = scan[i];
    // o3d_slam::transform (helpers.cpp:273-305) of the raw scan into the map frame, then the ray from the sensor position
    const double x = (double)q.x, y = (double)q.y, z = (double)q.z;
    const double px = M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3], py = M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7],
                 pz = M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11];
    const double dx = px - sx, dy = py - sy, dz = pz - sz;
    const double length = sqrt(dx * dx + dy * dy + dz * dz);
    if (!(length > 0.0)) continue;
    const double ux = dx / length, uy = dy / length, uz = dz / length;
    const double lim = fmax(voxel, fmin(length - trunc, max_len));
    // The samples of a ray do not depend on each other -- only `dist` does, by the reference's repeated addition -- so kBatch of them look
    // their blocks up together (the bit set answers from the L2, a few hundred cycles each when asked one by one); the few that find their
    // bit set go to the table one after the other (clearing a keep flag is an idempotent store, the order does not matter).
    constexpr int kBatch = 8;
    double dist = 0.0;
    while (dist < lim) {
      unsigned long long ks[kBatch];
      unsigned int bits[kBatch], words[kBatch];
      int m = 0;
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        if (dist < lim) {
          const double cx = dist * ux + sx, cy = dist * uy + sy, cz = dist * uz + sz;
          ks[u] = pack_key((long long)(int)floor(cx * inv), (long long)(int)floor(cy * inv), (long long)(int)floor(cz * inv));
          bits[u] = carve_block_bit(ks[u]);
          dist += voxel;
          m = u + 1;
        }
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
        if (u < m) words[u] = block_bits[bits[u] >> 5];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        if (u >= m) break;
        if (!((words[u] >> (bits[u] & 31u)) & 1u)) continue;  // nothing of the table in this block of voxels
        const unsigned long long k = ks[u];
        unsigned int slot = (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> 32) & mask;
        int seg = -1;
        while (true) {
          const unsigned long long t = tkey[slot];
          if (t == k) {
            seg = tseg[slot];
            break;
          }
          if (t == kEmptyKey) break;
          slot = (slot + 1) & mask;
        }
        if (seg < 0) continue;
        const size_t b = (size_t)seg_start[seg], e = (size_t)seg + 1 < n_seg ? (size_t)seg_start[seg + 1] : n_sorted;
        for (size_t j = b; j < e; ++j) {
          const uint32_t id = vals[j];
          bool rem = true;
          if (map_nrm) {
            const P4 nn = map_nrm[id];
            const double a = (double)nn.x, bb = (double)nn.y, c = (double)nn.z;
            const double nl = sqrt(a * a + bb * bb + c * c);
            const double dot = nl > 0.0 ? (ux * a + uy * bb + uz * c) / nl : 0.0;  // Eigen normalized(): the zero vector stays zero
            rem = fabs(dot) > min_dot;
          }
          if (rem) flags[id] = 0;  // flags are "keep" flags: 1 = stays in the map
        }
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void fill_int_kernel(int* __restrict__ p, size_t n, int v) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) p[i] = v;
}

// ---------------------------------------------------------------------------------------------- overlap on a voxel grid
// computeIndicesOfOverlappingPoints (helpers.cpp:307-332): keys of both clouds sorted together, per-voxel counts per cloud, flags.
constexpr uint32_t kSourceTag = 0x80000000u;

// vals[i] |= tag for the second (source) block of a concatenated key array
__global__ __launch_bounds__(kBlock) void overlap_tag_kernel(uint32_t* __restrict__ vals, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) vals[i] |= kSourceTag;
}
// counts of source / target members per segment (seg ids from the exclusive scan of the heads)
__global__ __launch_bounds__(kBlock) void overlap_count_kernel(const uint32_t* __restrict__ vals, const int* __restrict__ head,
                                                               const int* __restrict__ seg_id, size_t n, int* __restrict__ cnt_s,
                                                               int* __restrict__ cnt_t) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int sgm = seg_id[i] + head[i] - 1;  // exclusive scan of heads: the segment of element i is (#heads up to and including i) - 1
    atomicAdd((vals[i] & kSourceTag) ? &cnt_s[sgm] : &cnt_t[sgm], 1);
  }
}
__global__ __launch_bounds__(kBlock) void overlap_flag_kernel(const uint32_t* __restrict__ vals, const int* __restrict__ head,
                                                              const int* __restrict__ seg_id, size_t n, const int* __restrict__ cnt_s,
                                                              const int* __restrict__ cnt_t, int min_points, int* __restrict__ flag_s,
                                                              int* __restrict__ flag_t) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int sgm = seg_id[i] + head[i] - 1;
    const bool ok = cnt_s[sgm] >= min_points && cnt_t[sgm] >= min_points;
    const uint32_t v = vals[i];
    if (v & kSourceTag)
      flag_s[v & ~kSourceTag] = ok ? 1 : 0;
    else
      flag_t[v] = ok ? 1 : 0;
  }
}
// out[pos[i]] = i where flag[i]
__global__ __launch_bounds__(kBlock) void index_compact_kernel(const int* __restrict__ flag, const int* __restrict__ pos, size_t n,
                                                               unsigned long long* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    if (flag[i]) out[pos[i]] = (unsigned long long)i;
}

// ---------------------------------------------------------------------------------------------- dense voxel map
// VoxelizedPointCloud (Voxel.hpp:38-76, Voxel.cpp:18-114): a persistent hash map voxel -> {count, sum of positions, sum of normals}.
// Device form: open addressing on the packed 64-bit voxel key (atomicCAS), counts and sums updated with integer atomics -- the sums
// are kept in FIXED POINT (2^-30 m for positions, 2^-40 for normals), so they are exact, order-independent and reproducible no matter
// how the insertions interleave; the quantisation (<= 0.5 nm per inserted point) is far below the f32 / f64 noise of the inputs.
constexpr double kDensePosQ = 1.0 / 1073741824.0;        // 2^-30
constexpr double kDenseNrmQ = 1.0 / 1099511627776.0;     // 2^-40
constexpr double kDenseColQ = 1.0 / 16777216.0;          // 2^-24: colours reach 255 when they come from an intensity byte (open3d_conversions.cpp:81-86)

struct DenseDev {
  unsigned long long* keys;  // [cap], kEmptyKey = free
  int* cnt;                  // [cap]
  long long* sp;             // [3 * cap] sum of positions / kDensePosQ
  long long* sn;             // [3 * cap] sum of normals / kDenseNrmQ
  long long* sc;             // [3 * cap] sum of colours / kDenseColQ (AggregatedVoxel::aggregateColor, Voxel.cpp:33-35)
  unsigned int mask;         // cap - 1
};

__device__ __forceinline__ unsigned int dense_slot_of(unsigned long long k, unsigned int mask) {
  return (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}
__device__ __forceinline__ unsigned int dense_find_or_insert(const DenseDev& d, unsigned long long k) {
  unsigned int slot = dense_slot_of(k, d.mask);
  while (true) {
    const unsigned long long prev = atomicCAS(&d.keys[slot], kEmptyKey, k);
    if (prev == kEmptyKey || prev == k) return slot;
    slot = (slot + 1) & d.mask;
  }
}

// VoxelizedPointCloud::insert (Voxel.cpp:66-90) of o3d_slam::transform(T, cloud) (Submap.cpp:81-84)
template <typename P4>
__global__ __launch_bounds__(kBlock) void dense_insert_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, const P4* __restrict__ col,
                                                              size_t n, Mat34 M, double inv, DenseDev d) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    const double px = M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3], py = M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7],
                 pz = M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11];
    const unsigned long long k = pack_key((long long)(int)floor(px * inv), (long long)(int)floor(py * inv), (long long)(int)floor(pz * inv));
    const unsigned int slot = dense_find_or_insert(d, k);
    atomicAdd(&d.cnt[slot], 1);
    atomicAdd((unsigned long long*)&d.sp[3 * (size_t)slot], (unsigned long long)llrint(px / kDensePosQ));
    atomicAdd((unsigned long long*)&d.sp[3 * (size_t)slot + 1], (unsigned long long)llrint(py / kDensePosQ));
    atomicAdd((unsigned long long*)&d.sp[3 * (size_t)slot + 2], (unsigned long long)llrint(pz / kDensePosQ));
    if (nrm) {
      const P4 q = nrm[i];
      const double a = (double)q.x, b = (double)q.y, c = (double)q.z;
      const double nx = M.m[0] * a + M.m[1] * b + M.m[2] * c, ny = M.m[4] * a + M.m[5] * b + M.m[6] * c, nz = M.m[8] * a + M.m[9] * b + M.m[10] * c;
      atomicAdd((unsigned long long*)&d.sn[3 * (size_t)slot], (unsigned long long)llrint(nx / kDenseNrmQ));
      atomicAdd((unsigned long long*)&d.sn[3 * (size_t)slot + 1], (unsigned long long)llrint(ny / kDenseNrmQ));
      atomicAdd((unsigned long long*)&d.sn[3 * (size_t)slot + 2], (unsigned long long)llrint(nz / kDenseNrmQ));
    }
    if (col) {  // colours are not rotated (o3d_slam::transform leaves colors_ alone)
      const P4 c = col[i];
      atomicAdd((unsigned long long*)&d.sc[3 * (size_t)slot], (unsigned long long)llrint((double)c.x / kDenseColQ));
      atomicAdd((unsigned long long*)&d.sc[3 * (size_t)slot + 1], (unsigned long long)llrint((double)c.y / kDenseColQ));
      atomicAdd((unsigned long long*)&d.sc[3 * (size_t)slot + 2], (unsigned long long)llrint((double)c.z / kDenseColQ));
    }
  }
}

// ---- one dense map over several GPUs: rows of a placed scan grouped by the rank that owns their voxel -----------------------------
// owner = the reference's voxel hash (x + 17191 y + 17191^2 z as a 32-bit unsigned, VoxelHashMap.hpp:25-35) of the voxel index the
// importing rank will bin (floor(value * (1 / voxel)) of the value ROUNDED TO THE STORAGE TYPE, as dense_insert_kernel computes it on
// the imported cloud), modulo the number of ranks.  Non-finite points have no owner and are dropped (the reference drops NaN returns in
// its cropper before they reach the map).  Rows are [x y z nx ny nz] doubles; the order inside a destination is the arrival order of
// an atomic -- irrelevant, the fusion sums are fixed point.
template <typename P4>
__device__ __forceinline__ void placed_point(const P4& p, const Mat34& M, bool place, typename Scalar<P4>::type out[3]) {
  using R = typename Scalar<P4>::type;
  const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  out[0] = place ? (R)(M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3]) : p.x;
  out[1] = place ? (R)(M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7]) : p.y;
  out[2] = place ? (R)(M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11]) : p.z;
}
// all lanes of the wavefront that hold the same small integer act through one of them: returns the number of lanes with my value
// (`cnt`), my rank among them (`rank`) and whether I am their first lane -- one atomic per (wavefront, value) instead of one per lane
__device__ __forceinline__ void wave_group_by(int v, bool valid, int* cnt, int* rank, bool* first) {
  *cnt = 0, *rank = 0, *first = false;
  unsigned long long todo = __ballot(valid);
  const int lane = threadIdx.x & 63;
  while (todo) {  // uniform: one round per distinct value present
    const int lead = __ffsll((long long)todo) - 1;
    const int val = __builtin_amdgcn_readlane(v, lead);
    const unsigned long long same = __ballot(valid && v == val);
    if (valid && v == val) {
      *cnt = __popcll(same);
      *rank = __popcll(same & ((1ull << lane) - 1ull));
      *first = lane == lead;
    }
    todo &= ~same;
  }
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void owner_count_kernel(const P4* __restrict__ pts, size_t n, Mat34 M, int place, double inv, int world,
                                                             int* __restrict__ owner, long long* __restrict__ counts) {
  __shared__ unsigned int s_hist[64];
  if (threadIdx.x < 64) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const size_t n_up = (n + 63) / 64 * 64;  // whole wavefronts stay in the loop together (the grouping uses wave-wide ballots)
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_up; i += (size_t)gridDim.x * kBlock) {
    int o = -1;
    if (i < n) {
      typename Scalar<P4>::type q[3];
      placed_point<P4>(pts[i], M, place != 0, q);
      if (isfinite((double)q[0]) && isfinite((double)q[1]) && isfinite((double)q[2])) {
        const long long kx = (long long)(int)floor((double)q[0] * inv), ky = (long long)(int)floor((double)q[1] * inv),
                        kz = (long long)(int)floor((double)q[2] * inv);
        const unsigned long long hsh = (unsigned long long)(kx + 17191ll * ky + 17191ll * 17191ll * kz) & 0xffffffffull;
        o = (int)(hsh % (unsigned long long)world);
      }
      owner[i] = o;
    }
    int cnt, rank;
    bool first;
    wave_group_by(o, o >= 0, &cnt, &rank, &first);
    if (first) atomicAdd(&s_hist[o], (unsigned int)cnt);
  }
  __syncthreads();
  if (threadIdx.x < world && s_hist[threadIdx.x]) atomicAdd((unsigned long long*)&counts[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}
__global__ void owner_offsets_kernel(const long long* __restrict__ counts, int world, unsigned long long* __restrict__ offsets /* [world] then cursors [world] */) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long run = 0;
    for (int r = 0; r < world; ++r) {
      offsets[r] = run;
      offsets[world + r] = 0;
      run += (unsigned long long)counts[r];
    }
  }
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void owner_scatter_kernel(const P4* __restrict__ pts, const P4* __restrict__ nrm, size_t n, Mat34 M, int place,
                                                               int world, const int* __restrict__ owner, unsigned long long* __restrict__ offsets,
                                                               double* __restrict__ rows) {
  const size_t n_up = (n + 63) / 64 * 64;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_up; i += (size_t)gridDim.x * kBlock) {
    const int o = i < n ? owner[i] : -1;
    int cnt, rank;
    bool first;
    wave_group_by(o, o >= 0, &cnt, &rank, &first);
    unsigned long long base = 0;
    if (first) base = atomicAdd(&offsets[world + o], (unsigned long long)cnt);  // one reservation per (wavefront, owner)
    // everyone of the group reads the leader's base: the leader is the group's first lane
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(o >= 0);
    unsigned long long mybase = 0;
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const int val = __builtin_amdgcn_readlane(o, lead);
      const unsigned long long b = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(base >> 32), lead) << 32) |
                                   (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(base & 0xffffffffull), lead);
      const unsigned long long same = __ballot(o == val && o >= 0);
      if (o == val) mybase = b;
      todo &= ~same;
    }
    (void)lane;
    if (o < 0) continue;
    const unsigned long long pos = offsets[o] + mybase + (unsigned long long)rank;
    typename Scalar<P4>::type q[3];
    placed_point<P4>(pts[i], M, place != 0, q);
    double* r = rows + 6 * pos;
    r[0] = (double)q[0], r[1] = (double)q[1], r[2] = (double)q[2];
    double a = 0.0, b = 0.0, c = 0.0;
    if (nrm) {
      const P4 v = nrm[i];
      a = (double)v.x, b = (double)v.y, c = (double)v.z;
      if (place) {  // normals rotate with the cloud (o3d_slam::transform, helpers.cpp:273-305)
        const double x = a, y = b, z = c;
        a = M.m[0] * x + M.m[1] * y + M.m[2] * z, b = M.m[4] * x + M.m[5] * y + M.m[6] * z, c = M.m[8] * x + M.m[9] * y + M.m[10] * z;
      }
    }
    r[3] = a, r[4] = b, r[5] = c;
  }
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void rows_to_cloud_kernel(const double* __restrict__ rows, size_t n, P4* __restrict__ pts, P4* __restrict__ nrm) {
  using R = typename Scalar<P4>::type;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const double* r = rows + 6 * i;
    P4 p;
    p.x = (R)r[0], p.y = (R)r[1], p.z = (R)r[2];
    p.i = (typename Scalar<P4>::index)i;
    pts[i] = p;
    if (nrm) {
      P4 q;
      q.x = (R)r[3], q.y = (R)r[4], q.z = (R)r[5];
      q.i = 0;
      nrm[i] = q;
    }
  }
}

// VoxelHashMap::hasVoxelContainingPoint (VoxelHashMap.hpp:110-114) for every point of a placed cloud: number of hits
// (SubmapCollection::isSwitchingSubmapsConsistant, SubmapCollection.cpp:352-364, divides it by the cloud size)
template <typename P4>
__global__ __launch_bounds__(kBlock) void dense_probe_kernel(const P4* __restrict__ pts, size_t n, Mat34 M, double inv, DenseDev d,
                                                             unsigned long long* __restrict__ hits) {
  unsigned int mine = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const P4 p = pts[i];
    const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
    const double px = M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3], py = M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7],
                 pz = M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11];
    const unsigned long long k = pack_key((long long)(int)floor(px * inv), (long long)(int)floor(py * inv), (long long)(int)floor(pz * inv));
    unsigned int slot = dense_slot_of(k, d.mask);
    while (true) {
      const unsigned long long t = d.keys[slot];
      if (t == k) {
        mine += d.cnt[slot] > 0 ? 1u : 0u;
        break;
      }
      if (t == kEmptyKey) break;
      slot = (slot + 1) & d.mask;
    }
  }
  if (mine) atomicAdd(hits, (unsigned long long)mine);
}

// Carving of the dense map: Submap::carve (Submap.cpp:126-136) = removeDuplicatePointsWithinSameVoxels (Voxel.cpp:162-191; done by the
// caller of this kernel through a key sort: `first[i]` = point i is the first of its voxel) + getKeysOfCarvedPoints
// (helpers.cpp:347-377) + getVoxelsWithinPointNeighborhood (VoxelHashMap.cpp:13-44, keys by DIVISION floor(p / v) as written there).
// One thread per kept scan point; marking a slot is an idempotent store.
__device__ __forceinline__ void dense_mark(const DenseDev& d, unsigned long long k, int* __restrict__ mark) {
  unsigned int slot = dense_slot_of(k, d.mask);
  while (true) {
    const unsigned long long t = d.keys[slot];
    if (t == k) {
      if (d.cnt[slot] > 0) mark[slot] = 1;
      return;
    }
    if (t == kEmptyKey) return;
    slot = (slot + 1) & d.mask;
  }
}
__device__ __forceinline__ unsigned long long key_by_division(double x, double y, double z, double v) {
  return pack_key((long long)(int)floor(x / v), (long long)(int)floor(y / v), (long long)(int)floor(z / v));
}
template <typename P4>
__global__ __launch_bounds__(kBlock) void dense_carve_kernel(const P4* __restrict__ scan, const int* __restrict__ first, size_t n, Mat34 M,
                                                             double sx, double sy, double sz, double v, double radius, double max_len,
                                                             double trunc, DenseDev d, int* __restrict__ mark) {
  const double step = 2.0 * radius;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    if (!first[i]) continue;
    const P4 q = scan[i];
    const double x = (double)q.x, y = (double)q.y, z = (double)q.z;
    const double px = M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3], py = M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7],
                 pz = M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11];
    const double dx0 = px - sx, dy0 = py - sy, dz0 = pz - sz;
    const double length = sqrt(dx0 * dx0 + dy0 * dy0 + dz0 * dz0);
    if (!(length > 0.0)) continue;
    const double ux = dx0 / length, uy = dy0 / length, uz = dz0 / length;
    const double lim = fmax(step, fmin(length - trunc, max_len));
    for (double dist = 0.0; dist < lim; dist += step) {
      const double cx = dist * ux + sx, cy = dist * uy + sy, cz = dist * uz + sz;
      const unsigned long long ck = key_by_division(cx, cy, cz, v);
      bool center_added = false;
      for (double ax = -radius; ax <= radius; ax += v)
        for (double ay = -radius; ay <= radius; ay += v)
          for (double az = -radius; az <= radius; az += v) {
            const double tx = cx + ax, ty = cy + ay, tz = cz + az;
            const double ex = tx - (floor(tx / v) * v + v * 0.5), ey = ty - (floor(ty / v) * v + v * 0.5), ez = tz - (floor(tz / v) * v + v * 0.5);
            if (sqrt(ex * ex + ey * ey + ez * ez) <= radius) {
              const unsigned long long k = key_by_division(tx, ty, tz, v);
              center_added |= k == ck;
              dense_mark(d, k, mark);
            }
          }
      if (!center_added) dense_mark(d, ck, mark);
    }
  }
}
// first[i] = 1 iff point vals[j] heads its key segment in the sorted order (lowest index of its voxel: the sort is stable)
__global__ __launch_bounds__(kBlock) void first_of_voxel_kernel(const unsigned long long* __restrict__ keys_sorted, const uint32_t* __restrict__ vals,
                                                                size_t n, int* __restrict__ first) {
  for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (size_t)gridDim.x * kBlock)
    first[vals[j]] = (j == 0 || keys_sorted[j] != keys_sorted[j - 1]) ? 1 : 0;
}
// VoxelHashMap::removeKey for every marked slot: the slot stays as a tombstone with count 0 (skipped by toPointCloud / transform,
// "not present" for hasVoxelWithKey, re-usable by a later insert); *removed += number of voxels that had points
__global__ __launch_bounds__(kBlock) void dense_erase_marked_kernel(DenseDev d, size_t cap, const int* __restrict__ mark,
                                                                    unsigned long long* __restrict__ removed) {
  unsigned int mine = 0;
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (size_t)gridDim.x * kBlock) {
    if (!mark[s] || d.cnt[s] <= 0) continue;
    d.cnt[s] = 0;
    for (int c = 0; c < 3; ++c) d.sp[3 * s + c] = 0, d.sn[3 * s + c] = 0, d.sc[3 * s + c] = 0;
    ++mine;
  }
  if (mine) atomicAdd(removed, (unsigned long long)mine);
}

// move every used slot of `from` into `to` (a larger, empty table)
__global__ __launch_bounds__(kBlock) void dense_rehash_kernel(DenseDev from, size_t from_cap, DenseDev to) {
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < from_cap; s += (size_t)gridDim.x * kBlock) {
    const unsigned long long k = from.keys[s];
    if (k == kEmptyKey || from.cnt[s] <= 0) continue;    // free slots, and tombstones left by carving: dropped here
    const unsigned int t = dense_find_or_insert(to, k);  // keys are unique: this thread owns slot t
    to.cnt[t] = from.cnt[s];
    for (int c = 0; c < 3; ++c) {
      to.sp[3 * (size_t)t + c] = from.sp[3 * s + c];
      to.sn[3 * (size_t)t + c] = from.sn[3 * s + c];
      to.sc[3 * (size_t)t + c] = from.sc[3 * s + c];
    }
  }
}

// list of used slots: flags for the scan, then (key, slot) pairs.  *occupied counts every slot whose key is taken -- live voxels
// AND the count-0 tombstones carving leaves behind: that, not the number of live voxels, is what the load factor is about
__global__ __launch_bounds__(kBlock) void dense_used_flag_kernel(DenseDev d, size_t cap, int* __restrict__ flag,
                                                                 unsigned long long* __restrict__ occupied) {
  unsigned int mine = 0;
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (size_t)gridDim.x * kBlock) {
    const bool taken = d.keys[s] != kEmptyKey;
    flag[s] = (taken && d.cnt[s] > 0) ? 1 : 0;
    mine += taken ? 1u : 0u;
  }
  if (mine) atomicAdd(occupied, (unsigned long long)mine);
}
__global__ __launch_bounds__(kBlock) void dense_list_kernel(DenseDev d, size_t cap, const int* __restrict__ flag, const int* __restrict__ pos,
                                                            unsigned long long* __restrict__ keys_out, uint32_t* __restrict__ slots_out) {
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (size_t)gridDim.x * kBlock)
    if (flag[s]) {
      keys_out[pos[s]] = d.keys[s];
      slots_out[pos[s]] = (uint32_t)s;
    }
}
// VoxelizedPointCloud::toPointCloud (Voxel.cpp:92-114): sum / count per voxel, in the order of `slots` (sorted by key)
template <typename P4>
__global__ __launch_bounds__(kBlock) void dense_emit_kernel(DenseDev d, const uint32_t* __restrict__ slots, size_t m, P4* __restrict__ out_pts,
                                                            P4* __restrict__ out_nrm, P4* __restrict__ out_col) {
  using R = typename Scalar<P4>::type;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock) {
    const size_t s = slots[i];
    const double c = (double)d.cnt[s];
    P4 p;
    p.x = (R)((double)d.sp[3 * s] * kDensePosQ / c);
    p.y = (R)((double)d.sp[3 * s + 1] * kDensePosQ / c);
    p.z = (R)((double)d.sp[3 * s + 2] * kDensePosQ / c);
    p.i = (typename Scalar<P4>::index)i;
    out_pts[i] = p;
    if (out_nrm) {
      P4 q;
      q.x = (R)((double)d.sn[3 * s] * kDenseNrmQ / c);
      q.y = (R)((double)d.sn[3 * s + 1] * kDenseNrmQ / c);
      q.z = (R)((double)d.sn[3 * s + 2] * kDenseNrmQ / c);
      q.i = 0;
      out_nrm[i] = q;
    }
    if (out_col) {
      P4 q;
      q.x = (R)((double)d.sc[3 * s] * kDenseColQ / c);
      q.y = (R)((double)d.sc[3 * s + 1] * kDenseColQ / c);
      q.z = (R)((double)d.sc[3 * s + 2] * kDenseColQ / c);
      q.i = 0;
      out_col[i] = q;
    }
  }
}
// VoxelizedPointCloud::transform (Voxel.cpp:49-64), quirks included: the keys stay, and the Isometry is applied to the SUMS as if
// they were points (translation added once to the position sum -- and to the normal sum as well)
__global__ __launch_bounds__(kBlock) void dense_transform_kernel(DenseDev d, size_t cap, Mat34 M) {
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += (size_t)gridDim.x * kBlock) {
    if (d.keys[s] == kEmptyKey || d.cnt[s] <= 0) continue;
    const double x = (double)d.sp[3 * s] * kDensePosQ, y = (double)d.sp[3 * s + 1] * kDensePosQ, z = (double)d.sp[3 * s + 2] * kDensePosQ;
    d.sp[3 * s] = llrint((M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3]) / kDensePosQ);
    d.sp[3 * s + 1] = llrint((M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7]) / kDensePosQ);
    d.sp[3 * s + 2] = llrint((M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11]) / kDensePosQ);
    const double a = (double)d.sn[3 * s] * kDenseNrmQ, b = (double)d.sn[3 * s + 1] * kDenseNrmQ, c = (double)d.sn[3 * s + 2] * kDenseNrmQ;
    d.sn[3 * s] = llrint((M.m[0] * a + M.m[1] * b + M.m[2] * c + M.m[3]) / kDenseNrmQ);
    d.sn[3 * s + 1] = llrint((M.m[4] * a + M.m[5] * b + M.m[6] * c + M.m[7]) / kDenseNrmQ);
    d.sn[3 * s + 2] = llrint((M.m[8] * a + M.m[9] * b + M.m[10] * c + M.m[11]) / kDenseNrmQ);
  }
}

#pragma clang fp contract(fast)

}  // namespace o3ds
