// pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_EA0_* counters on gfx950 against KNOWN byte counts in the
// access patterns of this backend (MI355X_MICROARCH.md "HBM": "calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Not product code.
//
// Every configuration is its own kernel name (template id), launched twice back to back (cold / warm), so that the per-dispatch counter
// rows of a `rocprofv3 --pmc ... --kernel-trace` pass can be matched to the table this program prints:
//   id  pattern   working set   requested bytes   distinct 64-B / 128-B blocks touched
// Patterns (16-B records, one per lane, n_reads records per launch):
//   stream  : record i                                   -- wide coalesced streaming read (the guide's calibrated case: counter x 2)
//   run8    : aligned runs of 8 records (128 B) at random run positions   -- every touched 128-B line is used completely
//   run4    : aligned runs of 4 records (64 B) at random positions        -- half of every 128-B line, all of every 64-B sector pair
//   run1    : single records at random positions                          -- 16 B of every touched line
//   icp     : 16 "queries" per wavefront, 4 lanes each; a query reads 4 runs of 4 consecutive records around a position that moves slowly
//             along the array (neighbouring queries overlap) -- the locality of the ICP pass over a cell-sorted target
//   write16 : record i written (streaming 16-B stores)
// Working sets: 32 MiB (inside L2 x 8 = 32 MiB aggregate, well inside the Infinity Cache), 128 MiB (inside the 256 MiB Infinity Cache),
// 1 GiB (outside everything).  The index stream is generated on the fly from the lane id (no index array in memory).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
      std::exit(1);                                                              \
    }                                                                            \
  } while (0)

struct alignas(16) Rec {
  float x, y, z;
  int i;
};

__device__ __forceinline__ uint64_t mix(uint64_t v) {  // splitmix64 finaliser: a fixed pseudo-random map
  v += 0x9e3779b97f4a7c15ull;
  v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ull;
  v = (v ^ (v >> 27)) * 0x94d049bb133111ebull;
  return v ^ (v >> 31);
}

enum Pattern { STREAM = 0, RUN8 = 1, RUN4 = 2, RUN1 = 3, ICP = 4, WRITE16 = 5 };

template <int ID, int PATTERN>
__global__ __launch_bounds__(256) void calib_kernel(const Rec* __restrict__ src, Rec* __restrict__ dst, size_t n_rec, size_t n_reads, float* sink) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_reads) return;
  float acc = 0.f;
  if (PATTERN == STREAM) {
    const Rec r = src[t % n_rec];
    acc = r.x + r.y + r.z + (float)r.i;
  } else if (PATTERN == RUN8 || PATTERN == RUN4 || PATTERN == RUN1) {
    constexpr size_t RUN = PATTERN == RUN8 ? 8 : (PATTERN == RUN4 ? 4 : 1);
    const size_t run = t / RUN, within = t % RUN;
    const size_t n_runs = n_rec / RUN;
    const size_t pos = (size_t)(mix(run * 0x51ed27ull + ID) % n_runs) * RUN + within;
    const Rec r = src[pos];
    acc = r.x + r.y + r.z + (float)r.i;
  } else if (PATTERN == ICP) {
    // query q = t / 4, lane l = t % 4; queries advance ~6 records each through the array with a jitter of +-64 records; a query reads four
    // runs of 4 records at offsets {0, +row, +2 row, -row} (rows of a 3x3 cell block in a cell-sorted array are `row` records apart)
    const size_t q = t >> 2, l = t & 3;
    const size_t row = 4096;
    const size_t base = (q * 6 + (mix(q + ID) & 127)) % (n_rec - 4 * row - 8) + row;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t off = k == 3 ? base - row : base + (size_t)k * row;
      const Rec r = src[(off & ~(size_t)3) + l];
      acc += r.x + r.y + r.z + (float)r.i;
    }
  } else {  // WRITE16
    Rec r{(float)t, 1.f, 2.f, (int)t};
    dst[t % n_rec] = r;
    return;
  }
  if (acc == 123456.789f) sink[0] = acc;  // never true: keeps the loads alive
}

__global__ void fill_kernel(Rec* p, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = Rec{(float)(t & 1023), 0.5f, 0.25f, (int)(t & 0xffff)};
}

template <int ID, int PATTERN>
static void run(const char* name, const Rec* src, Rec* dst, size_t ws_bytes, size_t n_reads, float* sink, hipStream_t s) {
  const size_t n_rec = ws_bytes / sizeof(Rec);
  const int blocks = (int)((n_reads + 255) / 256);
  // flush: stream 1.5 GiB of another buffer through the caches between configurations is done by the caller; here cold then warm
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  CK(hipEventRecord(e0, s));
  calib_kernel<ID, PATTERN><<<blocks, 256, 0, s>>>(src, dst, n_rec, n_reads, sink);
  CK(hipEventRecord(e1, s));
  calib_kernel<ID, PATTERN><<<blocks, 256, 0, s>>>(src, dst, n_rec, n_reads, sink);
  CK(hipEventRecord(e2, s));
  CK(hipStreamSynchronize(s));
  float ms0 = 0, ms1 = 0;
  CK(hipEventElapsedTime(&ms0, e0, e1));
  CK(hipEventElapsedTime(&ms1, e1, e2));
  const double per = PATTERN == ICP ? 64.0 : 16.0;  // requested bytes per lane
  const double req = per * (double)n_reads;
  // distinct blocks touched (expected value for the random patterns; exact for stream / write / icp is not needed: printed as requested)
  std::printf("%2d %-8s ws_MiB %5zu requested_MB %9.2f cold_ms %8.3f warm_ms %8.3f cold_GBs %8.1f warm_GBs %8.1f\n", ID, name, ws_bytes >> 20, req / 1e6, ms0, ms1,
              req / ms0 / 1e6, req / ms1 / 1e6);
}

static void flush(Rec* big, size_t n, hipStream_t s) {  // evict: rewrite a 1.5 GiB buffer
  fill_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(big, n);
  CK(hipStreamSynchronize(s));
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const size_t GiB = (size_t)1 << 30;
  Rec *buf = nullptr, *big = nullptr;
  float* sink = nullptr;
  CK(hipMalloc(&buf, GiB));
  CK(hipMalloc(&big, GiB + GiB / 2));
  CK(hipMalloc(&sink, 64));
  const size_t nbig = (GiB + GiB / 2) / sizeof(Rec);
  fill_kernel<<<(int)((GiB / sizeof(Rec) + 255) / 256), 256, 0, s>>>(buf, GiB / sizeof(Rec));
  CK(hipStreamSynchronize(s));
  const size_t N = (size_t)16 << 20;  // 16 Mi lanes per launch: 256 MiB requested (1 GiB for the icp pattern)
  std::printf("# id pattern working-set requested cold/warm time; every id is launched twice (cold, warm) as calib_kernel<id, pattern>\n");
#define ROW(ID, PAT, NAME, WS)          \
  flush(big, nbig, s);                  \
  run<ID, PAT>(NAME, buf, buf, (size_t)(WS) << 20, N, sink, s);
  ROW(1, STREAM, "stream", 32)
  ROW(2, STREAM, "stream", 128)
  ROW(3, STREAM, "stream", 1024)
  ROW(4, RUN8, "run8", 32)
  ROW(5, RUN8, "run8", 128)
  ROW(6, RUN8, "run8", 1024)
  ROW(7, RUN4, "run4", 32)
  ROW(8, RUN4, "run4", 128)
  ROW(9, RUN4, "run4", 1024)
  ROW(10, RUN1, "run1", 32)
  ROW(11, RUN1, "run1", 128)
  ROW(12, RUN1, "run1", 1024)
  ROW(13, ICP, "icp", 32)
  ROW(14, ICP, "icp", 128)
  ROW(15, ICP, "icp", 1024)
  ROW(16, WRITE16, "write16", 32)
  ROW(17, WRITE16, "write16", 1024)
  CK(hipFree(buf));
  CK(hipFree(big));
  CK(hipFree(sink));
  return 0;
}
