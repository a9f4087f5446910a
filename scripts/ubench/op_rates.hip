// issue cost of a few VALU operations on gfx950, one wavefront per SIMD alone: cycles per instruction from s_memtime around an unrolled chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 256
template <int K>
__global__ void k(unsigned long long* out, unsigned long long a0, unsigned long long b0, double d0) {
  unsigned long long a = a0 + threadIdx.x, b = b0 + threadIdx.x * 3;
  unsigned int r = 0, x = (unsigned)a, y = (unsigned)b;
  double d = d0 + threadIdx.x, e = d0 * 0.5;
  float f = (float)d0 + threadIdx.x, g = 0.5f;
  const long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    if (K == 0) { r += (a < b) ? 1 : 0; b += 0x100000001ull; asm volatile("" : "+v"(b)); }                 // v_cmp_lt_u64 + addc + 64-bit add
    if (K == 1) { r += (x < y) ? 1 : 0; y += 3; asm volatile("" : "+v"(y)); }                             // v_cmp_lt_u32 + addc + add
    if (K == 2) { d = __builtin_fma(d, e, e); asm volatile("" : "+v"(d)); }                               // v_fma_f64
    if (K == 3) { f = __builtin_fmaf(f, g, g); asm volatile("" : "+v"(f)); }                              // v_fma_f32
    if (K == 4) { d = __builtin_sqrt(d + 1.0); asm volatile("" : "+v"(d)); }                              // f64 sqrt
    if (K == 5) { a += b; b ^= a; asm volatile("" : "+v"(a), "+v"(b)); }                                           // 64-bit add + xor
    if (K == 6) { r += (a < b) ? 1 : 0; asm volatile("" : "+v"(a)); }         // cmp_u64 + addc only
    if (K == 7) { r += (x < y) ? 1 : 0; asm volatile("" : "+v"(x)); }         // cmp_u32 + addc only
  }
  const long long t1 = clock64();
  out[threadIdx.x + 64 * blockIdx.x] = (unsigned long long)(t1 - t0) + ((r + (unsigned)a + (unsigned)b + x + y + (unsigned)d + (unsigned)f) & 0);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8 * 64);
  const char* names[] = {"cmp_lt_u64+addc+add_u64", "cmp_lt_u32+addc+add_u32", "fma_f64", "fma_f32", "sqrt_f64(+add)", "add_u64+xor_b64", "cmp_lt_u64+addc", "cmp_lt_u32+addc"};
  auto run = [&](int kk) {
    for (int w = 0; w < 2; ++w) {
      switch (kk) { case 0: k<0><<<1,64>>>(d,1,2,1.5); break; case 1: k<1><<<1,64>>>(d,1,2,1.5); break; case 2: k<2><<<1,64>>>(d,1,2,1.5); break; case 3: k<3><<<1,64>>>(d,1,2,1.5); break;
        case 4: k<4><<<1,64>>>(d,1,2,1.5); break; case 5: k<5><<<1,64>>>(d,1,2,1.5); break; case 6: k<6><<<1,64>>>(d,1,2,1.5); break; case 7: k<7><<<1,64>>>(d,1,2,1.5); break; }
      hipDeviceSynchronize();
    }
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-28s %6.2f clock64 ticks per iteration\n", names[kk], (double)h / REP);
  };
  for (int kk = 0; kk < 8; ++kk) run(kk);
  return 0;
}
