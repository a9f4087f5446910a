// common.hpp -- shared device/host types of the gfx950 backend (no torch, no Eigen).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <string>

#include "../../include/o3ds_backend.h"

namespace o3ds {

// Device point records.  xyz + the point's index in its source cloud, one aligned vector load each
// (16 B for f32 storage, 32 B for f64 storage).
struct alignas(16) P4f {
  float x, y, z;
  int32_t i;
};
struct alignas(32) P4d {
  double x, y, z;
  int64_t i;
};

template <typename P4>
struct Scalar;
template <>
struct Scalar<P4f> {
  using type = float;
  using index = int32_t;
};
template <>
struct Scalar<P4d> {
  using type = double;
  using index = int64_t;
};

// Cropping volume in device form (croppers.cpp:121-165).  The reference compares DISTANCES (the square root of a squared norm) with
// its radii; here the squared norm is compared with thresholds that are the radii's exact pre-images under the correctly rounded
// square root -- le2 = the largest x with sqrt(x) <= rmax, ge2 = the smallest x with sqrt(x) >= rmin (to_dev, a few steps of
// nextafter around r * r on the host) -- so the predicate keeps exactly the points the reference keeps, points on the radius included,
// without a double-precision square root per candidate (fifteen instructions and the registers of the search's inner loop).
struct CropDev {
  int kind;    // o3ds_crop_kind
  int invert;
  double cx, cy, cz;
  double le2, ge2;  // thresholds of the squared distance for rmax / rmin (see above)
  double zmin, zmax;
};

#pragma clang fp contract(off)  // the distances below round as the reference's do (see cloud_kernels.hpp)
__host__ __device__ inline bool crop_contains(const CropDev& c, double x, double y, double z) {
  bool in = true;
  const double dx = x - c.cx, dy = y - c.cy, dz = z - c.cz;
  switch (c.kind) {
    case O3DS_CROP_MAX_RADIUS:
      in = dx * dx + dy * dy + dz * dz <= c.le2;
      break;
    case O3DS_CROP_MIN_RADIUS:
      in = dx * dx + dy * dy + dz * dz >= c.ge2;
      break;
    case O3DS_CROP_MIN_MAX_RADIUS: {
      const double d2 = dx * dx + dy * dy + dz * dz;
      in = d2 <= c.le2 && d2 >= c.ge2;
      break;
    }
    case O3DS_CROP_CYLINDER:
      in = z >= c.zmin && z <= c.zmax && dx * dx + dy * dy <= c.le2;
      break;
    default:
      return !c.invert;  // O3DS_CROP_NONE = the base CroppingVolume (croppers.cpp:49-55): everything is inside, so inverted nothing is
  }
  return c.invert ? !in : in;
}
#pragma clang fp contract(fast)

// the largest x with sqrt(x) <= r (r >= 0; -1 for a negative or NaN radius: nothing passes `<=`, as with the reference's comparison)
inline double sqrt_preimage_le(double r) {
  if (!(r >= 0.0)) return -1.0;
  if (std::isinf(r)) return r;
  double x = r * r;
  if (std::isinf(x)) x = 1.7976931348623157e308;
  while (std::sqrt(x) > r) x = std::nextafter(x, -1.0);
  for (;;) {
    const double up = std::nextafter(x, INFINITY);
    if (std::isinf(up) || !(std::sqrt(up) <= r)) break;
    x = up;
  }
  return x;
}
// the smallest x >= 0 with sqrt(x) >= r (0 for r <= 0: every squared distance passes `>=`; +inf for an infinite or NaN radius... a NaN
// radius passes nothing in the reference either: the comparison is false)
inline double sqrt_preimage_ge(double r) {
  if (r != r) return r;  // NaN: `>=` is false for every distance, as in the reference
  if (r <= 0.0) return 0.0;
  if (std::isinf(r)) return r;
  double x = r * r;
  if (std::isinf(x)) return x;
  while (std::sqrt(x) < r) x = std::nextafter(x, INFINITY);
  for (;;) {
    const double dn = std::nextafter(x, -1.0);
    if (dn < 0.0 || !(std::sqrt(dn) >= r)) break;
    x = dn;
  }
  return x;
}

inline CropDev to_dev(const o3ds_crop* c) {
  CropDev d{};
  if (!c) {
    d.kind = O3DS_CROP_NONE;
    return d;
  }
  d.kind = c->kind;
  d.invert = c->invert;
  d.cx = c->center[0];
  d.cy = c->center[1];
  d.cz = c->center[2];
  d.le2 = sqrt_preimage_le(c->rmax);
  d.ge2 = sqrt_preimage_ge(c->rmin);
  d.zmin = c->zmin;
  d.zmax = c->zmax;
  return d;
}

// Uniform grid over the target's bounding box; points are stored sorted by linear cell id
// (x fastest), so the cells x0..x1 of one (y,z) row are ONE contiguous range of the sorted array:
// [cell_start[row+x0], cell_start[row+x1+1]).
struct GridDev {
  double ox, oy, oz;  // min corner
  double cell;        // edge length
  double inv_cell;
  int nx, ny, nz;
  const int* cell_start;  // nx*ny*nz + 1 entries (exclusive scan of per-cell counts)
  int sx;                 // entries per row of cell_start: nx (the classic table), or nx + 1 for the row-paged table of a persistent map
                          // (map_kernels.hpp: every row owns a region with room to spare, the extra entry is the end of its last cell)
};

// ---- counts the host has not seen yet ---------------------------------------------------------------------------------------------
// A kernel that decides how many points a cloud has (VoxelDownSample: the number of voxels) leaves the number in a device word, where
// the kernels that consume the cloud read it, and in a pinned record {count, stamp} the host looks at when it next needs the number:
// the frame's launches are queued from the cloud's UPPER bound (the size of the input) and nothing waits for a size to come back.
struct CountPub {
  int* dev;   // device word
  int* host;  // pinned {count, stamp}; the stamp is stored last, with a system-scope release
  int seq;
};
__device__ __forceinline__ void publish_count(const CountPub& p, int cnt) {
  *p.dev = cnt;
  if (p.host) {
    p.host[0] = cnt;
    __hip_atomic_store(p.host + 1, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// what a consumer is handed: the exact count, or the upper bound together with the device word that holds the exact one
struct CountRef {
  size_t n;
  const int* dev;
};
__device__ __forceinline__ size_t count_of(const CountRef& c) { return c.dev ? (size_t)*c.dev : c.n; }

// A word ONE thread reads and then overwrites (the counters a bookkeeping kernel consumes and resets).  The compiler turns a uniform plain
// load into a scalar-cache load and waits for it only where the value is first USED; the hardware does not order the scalar and the vector
// memory pipelines against each other, so a vector store to the same word issued in between can overtake the load -- the thread reads its
// own reset (pm_carve_finish_kernel did, once in ten cold starts).  An agent-scope atomic load is a vector load: ordered with the wave's
// later stores to the address.  scripts/check_scalar_war.py looks for the pattern in the assembly of every kernel (tests/test_abi.py).
__device__ __forceinline__ int load_then_store(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// doubles as unsigned integers of the same order (atomicMin / atomicMax on bounding boxes)
__host__ __device__ __forceinline__ unsigned long long order_bits(double x) {
  unsigned long long u;
  __builtin_memcpy(&u, &x, 8);
  return (u >> 63) ? ~u : (u | (1ull << 63));
}
__host__ __device__ __forceinline__ double order_value(unsigned long long e) {
  const unsigned long long u = (e >> 63) ? (e & ~(1ull << 63)) : ~e;
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}

// The 32-double normal-equation record (see include/o3ds_backend.h, o3ds_icp_accumulate).
constexpr int kRec = 32;
constexpr int kRecR2 = 27, kRecCount = 28, kRecD2 = 29;

// Device-side ICP loop state ([O3D] RegistrationICP locals).
struct IcpStateDev {
  double T[16];  // column-major current transformation
  double fitness, rmse;
  unsigned long long n_corr;
  int pass;        // correspondence passes evaluated
  int iterations;  // updates applied
  int done;
  int converged;
  int error;  // 1: a workgroup of the persistent loop kernel timed out at the grid rendezvous (never expected)
  int pad;  // the pivot order of the last 6x6 solve (icp_kernels.hpp solve6_wave_ordered): bit 31 valid, 3 bits per position
};

}  // namespace o3ds
