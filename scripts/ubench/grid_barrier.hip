// What does a pass boundary cost?  (a) a device-side barrier over 1024 co-resident workgroups of 256 threads (release / acquire at agent scope,
// one counter), against (b) the same work as back-to-back launches on one stream.  Each "pass" does a token amount of work per workgroup (reads a
// few values another workgroup wrote in the previous pass, adds into a slot with an f64 atomic) so that the fences have something to order.
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kSlots = 8;
template <int kMode>
__device__ __forceinline__ bool grid_barrier(unsigned int* bar, unsigned int pass, unsigned int* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned int target = (pass + 1) * gridDim.x;
    if (kMode == 0) {  // the textbook form: release add, acquire loads
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *fail = 1; ok = false; break; }  // never hang the box
      }
    } else if (kMode == 1) {  // one release fence, relaxed add, relaxed polls, one acquire fence
      __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope in clang's builtin; the HIP one below is agent)
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { *fail = 1; ok = false; break; }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {  // two levels: eight counters (blockIdx % 8: the XCD a workgroup runs on under round-robin dispatch), the last arrival of each adds to the top one
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      unsigned int* mine = bar + 16 * (1 + (blockIdx.x & 7));
      const unsigned int per = (gridDim.x + 7 - (blockIdx.x & 7)) / 8;  // workgroups with this residue
      const unsigned int got = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got + 1 == (pass + 1) * per) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int top = (pass + 1) * (gridDim.x < 8 ? gridDim.x : 8);
      int spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < top) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { *fail = 1; ok = false; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  return ok;
}
template <int kMode>
__global__ __launch_bounds__(256) void persistent(double* slots, unsigned int* bar, unsigned int* fail, int passes, double* out) {
  double acc = 0.0;
  for (int p = 0; p < passes; ++p) {
    const double* in = slots + ((p + 2) % 3) * kSlots * 64;
    double* o = slots + (p % 3) * kSlots * 64;
    double* clr = slots + ((p + 1) % 3) * kSlots * 64;
    if (threadIdx.x < 64) {
      double v = 0.0;
      for (int s = 0; s < kSlots; ++s) v += __hip_atomic_load(in + s * 64 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += v;
      if (blockIdx.x < kSlots) clr[blockIdx.x * 64 + threadIdx.x] = 0.0;
      __hip_atomic_fetch_add(o + (blockIdx.x % kSlots) * 64 + threadIdx.x, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier<kMode>(bar, (unsigned int)p, fail)) break;
  }
  if (threadIdx.x < 64 && blockIdx.x == 0) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void one_pass(double* slots, int p, double* out) {
  const double* in = slots + ((p + 2) % 3) * kSlots * 64;
  double* o = slots + (p % 3) * kSlots * 64;
  double* clr = slots + ((p + 1) % 3) * kSlots * 64;
  if (threadIdx.x < 64) {
    double v = 0.0;
    for (int s = 0; s < kSlots; ++s) v += in[s * 64 + threadIdx.x];
    if (blockIdx.x < kSlots) clr[blockIdx.x * 64 + threadIdx.x] = 0.0;
    __hip_atomic_fetch_add(o + (blockIdx.x % kSlots) * 64 + threadIdx.x, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0) out[threadIdx.x] += v;
  }
}
int main() {
  double *slots, *out; unsigned int *bar, *fail;
  CK(hipMalloc(&slots, sizeof(double) * 3 * kSlots * 64)); CK(hipMalloc(&out, sizeof(double) * 64));
  CK(hipMalloc(&bar, 4 * 16 * 9)); CK(hipMalloc(&fail, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int nblk = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, persistent<0>, 256, 0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("CUs %d, resident workgroups per CU %d -> %d co-resident\n", prop.multiProcessorCount, nblk, nblk * prop.multiProcessorCount);
  for (int grid : {256, 512, 1024}) {
    if (grid > nblk * prop.multiProcessorCount) { printf("grid %d does not fit\n", grid); continue; }
    for (int passes : {12, 120}) {
      float bestm[3]; unsigned int f = 0;
      for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
          CK(hipMemsetAsync(slots, 0, sizeof(double) * 3 * kSlots * 64, st)); CK(hipMemsetAsync(bar, 0, 4 * 16 * 9, st)); CK(hipMemsetAsync(fail, 0, 4, st));
          void* args[] = {&slots, &bar, &fail, &passes, &out};
          CK(hipEventRecord(e0, st));
          const void* fn = mode == 0 ? (const void*)persistent<0> : mode == 1 ? (const void*)persistent<1> : (const void*)persistent<2>;
          CK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(256), args, 0, st));
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        unsigned int ff; CK(hipMemcpy(&ff, fail, 4, hipMemcpyDeviceToHost)); f |= ff;
        bestm[mode] = best;
      }
      float best2 = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemsetAsync(slots, 0, sizeof(double) * 3 * kSlots * 64, st));
        CK(hipEventRecord(e0, st));
        for (int p = 0; p < passes; ++p) one_pass<<<grid, 256, 0, st>>>(slots, p, out);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best2 = ms < best2 ? ms : best2;
      }
      printf("grid %4d x 256, %3d passes: persistent textbook %.2f | fenced once %.2f | two-level %.2f us per pass (fail %u) | launches %.2f us per pass\n", grid, passes, bestm[0] * 1e3 / passes, bestm[1] * 1e3 / passes, bestm[2] * 1e3 / passes, f, best2 * 1e3 / passes);
    }
  }
  return 0;
}
