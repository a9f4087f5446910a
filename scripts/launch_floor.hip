// launch_floor.hip -- how long does a chain of dependent, nearly empty kernels take on one stream: plain launches against one hipGraph launch of
// the same chain?  (The fused ICP loop is such a chain: 12 launches of 1024 workgroups per registration, >= 4.6 us each even when a launch only
// reads the state.)  Not product code.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ __launch_bounds__(256) void link_kernel(const int* in, int* out) {
  __shared__ int s;
  if (threadIdx.x == 0) s = *in;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) *out = s + 1;
}
int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int* d = nullptr;
  CK(hipMalloc(&d, 64 * sizeof(int)));
  CK(hipMemset(d, 0, 64 * sizeof(int)));
  const int chain = 12, reps = 200;
  for (int blocks : {1, 1024, 4096}) {
    auto run_plain = [&]() { for (int k = 0; k < chain; ++k) link_kernel<<<blocks, 256, 0, st>>>(d + (k & 1), d + ((k + 1) & 1)); };
    for (int w = 0; w < 20; ++w) run_plain();
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run_plain();
    CK(hipStreamSynchronize(st));
    const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    run_plain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
    std::printf("chain of %d dependent kernels, %4d workgroups: %.2f us per kernel as stream launches, %.2f us per kernel as one hipGraph launch\n", chain, blocks, plain, graph);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
