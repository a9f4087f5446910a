// What a kernel boundary costs as a function of the launch's shape: the same 262 144 threads as 1024 x 256, 512 x 512, 256 x 1024 (and
// half / a quarter of them), a token amount of work, back-to-back launches on one stream.  us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int kThreads>
__global__ __launch_bounds__(kThreads) void token(double* slots, int p) {
  if (threadIdx.x < 64) __hip_atomic_fetch_add(slots + ((p % 3) * 8 + (blockIdx.x % 8)) * 64 + threadIdx.x, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int kThreads>
float run(hipStream_t st, hipEvent_t e0, hipEvent_t e1, double* slots, int grid, int passes) {
  float best = 1e9f;
  for (int rep = 0; rep < 20; ++rep) {
    hipEventRecord(e0, st);
    for (int p = 0; p < passes; ++p) token<kThreads><<<grid, kThreads, 0, st>>>(slots, p);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best * 1e3f / passes;
}
int main() {
  double* slots; CK(hipMalloc(&slots, sizeof(double) * 3 * 8 * 64)); CK(hipMemset(slots, 0, sizeof(double) * 3 * 8 * 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int passes = 120;
  for (int total : {262144, 131072, 65536}) {
    printf("%7d threads: %4d x 256: %.2f us | %4d x 512: %.2f us | %4d x 1024: %.2f us | %4d x 128: %.2f us | %5d x 64: %.2f us\n", total,
           total / 256, run<256>(st, e0, e1, slots, total / 256, passes), total / 512, run<512>(st, e0, e1, slots, total / 512, passes),
           total / 1024, run<1024>(st, e0, e1, slots, total / 1024, passes), total / 128, run<128>(st, e0, e1, slots, total / 128, passes),
           total / 64, run<64>(st, e0, e1, slots, total / 64, passes));
  }
  return 0;
}
