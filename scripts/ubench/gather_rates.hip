// What a round of divergent loads costs on gfx950 when the whole chip does it: 1024 workgroups x 256 threads (= the geometry of
// icp_fused_kernel: 16 wavefronts per CU), every lane a chain of DEP dependent loads of ELEM bytes at pseudo-random places of a
// working set, PAR independent chains per lane in flight together.  Prints time per dependent round and lane-loads per second.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_rates scripts/ubench/gather_rates.hip && /tmp/gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int PAR, typename T, int GROUP /* lanes that read consecutive elements (1: every lane its own place) */>
__global__ __launch_bounds__(256) void chase(const T* __restrict__ data, unsigned mask, int dep, unsigned* out) {
  unsigned idx[PAR];
#pragma unroll
  for (int p = 0; p < PAR; ++p) idx[p] = mix((blockIdx.x * 256 + threadIdx.x) / GROUP * 977 + p * 7919 + 1);
  unsigned acc = 0;
  for (int d = 0; d < dep; ++d) {
    T v[PAR];
#pragma unroll
    for (int p = 0; p < PAR; ++p) v[p] = data[((idx[p] & mask) / GROUP * GROUP) + threadIdx.x % GROUP];
#pragma unroll
    for (int p = 0; p < PAR; ++p) {
      const unsigned w = *(const unsigned*)&v[p];
      acc += w;
      idx[p] = mix(idx[p] + (w & 1) + d);  // the next address depends on the loaded value
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
struct E16 { unsigned a, b, c, d; };
template <int PAR, typename T, int GROUP>
void run(const char* name, size_t bytes, int dep) {
  const size_t n = bytes / sizeof(T);
  T* d; hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
  unsigned* out; hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) chase<PAR, T, GROUP><<<1024, 256>>>(d, (unsigned)(n - 1), dep, out);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) chase<PAR, T, GROUP><<<1024, 256>>>(d, (unsigned)(n - 1), dep, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5;
  printf("%-34s ws %7.1f MB  PAR %d  %7.2f us per launch  %6.3f us per round  %7.1f G lane-loads/s\n", name, bytes / 1e6, PAR, us, us / dep,
         1024.0 * 256 * dep * PAR / us * 1e-3);
  hipFree(d); hipFree(out);
}
int main() {
  const int dep = 32;
  for (size_t ws : {(size_t)256 << 10, (size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30}) {
    run<1, unsigned, 1>("4 B, every lane its own line", ws, dep);
    run<4, unsigned, 1>("4 B, every lane its own line", ws, dep);
    run<4, E16, 1>("16 B, every lane its own line", ws, dep);
    run<4, E16, 4>("16 B, 4 lanes share a 64-B run", ws, dep);
    run<8, unsigned, 1>("4 B, every lane its own line", ws, dep);
  }
  return 0;
}
