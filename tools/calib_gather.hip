// calib_gather.hip -- calibration micro-benchmarks for the FM-index access pattern on MI355X.
// Measures (a) the peak rate of independent random 64-B / 128-B block gathers issued exactly the way the
// quad kernels issue them, (b) what rocprofv3's FETCH_SIZE reports for each pattern against the KNOWN
// number of bytes requested, so that the `traffic` figure in bench.py can be corrected as
// MI355X_MICROARCH.md prescribes ("calibrate on a known byte count in your own access pattern").
//   usage: calib_gather [table_MiB=2048] [accesses_M=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

// P0: streaming read, 16 B per lane
__global__ __launch_bounds__(256) void stream_read(const ulonglong2* __restrict__ t, uint64_t n16, uint64_t* sink) {
  uint64_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { ulonglong2 v = t[i]; acc += v.x ^ v.y; }
  if (acc == 0x1234567) sink[0] = acc;
}

// quad gathers: every quad reads `per_quad` random 128-B lines; MODE 1 = first 64-B half only (1 x dwordx4 per lane),
// MODE 2 = both halves (2 x dwordx4 per lane), MODE 3 = one 8-B word, same address in all 4 lanes (seed-probe shape),
// MODE 4 = one 8-B word per lane, every lane its own random line
template <int MODE>
__global__ __launch_bounds__(256) void quad_gather(const uint64_t* __restrict__ t, uint64_t nlines, uint64_t per_quad, uint64_t* sink, uint64_t salt) {
  const int l = threadIdx.x & 3;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t quad = gid >> 2;
  uint64_t acc = 0;
  for (uint64_t j = 0; j < per_quad; j++) {
    const uint64_t id = MODE == 4 ? gid : quad;
    const uint64_t line = mix(id * 0x100000001B3ull + j + salt) % nlines;
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(t + line * 16);
    if (MODE == 1) { ulonglong2 a = p[l]; acc += a.x ^ a.y; }
    if (MODE == 2) { ulonglong2 a = p[l], b = p[4 + l]; acc += a.x ^ a.y ^ b.x ^ b.y; }
    if (MODE == 3) { acc += t[line * 16 + 3]; }
    if (MODE == 4) { acc += t[line * 16 + 5]; }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <class F> float timeit(F&& f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; }
  return best;
}

int main(int argc, char** argv) {
  const uint64_t table_mib = argc > 1 ? strtoull(argv[1], 0, 10) : 2048;
  const uint64_t acc_m = argc > 2 ? strtoull(argv[2], 0, 10) : 256;
  const uint64_t bytes = table_mib << 20, nlines = bytes / 128;
  uint64_t *t, *sink;
  CK(hipMalloc(&t, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(t, 1, bytes));
  const int grid = 256 * 8, block = 256;
  const uint64_t quads = (uint64_t)grid * block / 4, lanes = (uint64_t)grid * block;
  printf("table %llu MiB (%llu lines of 128 B), grid %d x %d\n", (unsigned long long)table_mib, (unsigned long long)nlines, grid, block);
  {
    float ms = timeit([&] { hipLaunchKernelGGL(stream_read, dim3(grid), dim3(block), 0, 0, (const ulonglong2*)t, bytes / 16, sink); });
    printf("P0 stream_read      : %8.3f ms  %8.1f GB/s   requested_bytes=%llu\n", ms, bytes / ms / 1e6, (unsigned long long)bytes);
  }
  const uint64_t per_quad = acc_m * 1000000ull / quads;
  uint64_t salt = 1;
  {
    float ms = timeit([&] { hipLaunchKernelGGL(quad_gather<1>, dim3(grid), dim3(block), 0, 0, t, nlines, per_quad, sink, salt++); });
    double n = (double)per_quad * quads;
    printf("P1 quad 64B half    : %8.3f ms  %8.2f G lines/s  %8.1f GB/s(64B)  requested_bytes=%.0f\n", ms, n / ms / 1e6, n * 64 / ms / 1e6, n * 64);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(quad_gather<2>, dim3(grid), dim3(block), 0, 0, t, nlines, per_quad, sink, salt++); });
    double n = (double)per_quad * quads;
    printf("P2 quad 128B line   : %8.3f ms  %8.2f G lines/s  %8.1f GB/s(128B) requested_bytes=%.0f\n", ms, n / ms / 1e6, n * 128 / ms / 1e6, n * 128);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(quad_gather<3>, dim3(grid), dim3(block), 0, 0, t, nlines, per_quad, sink, salt++); });
    double n = (double)per_quad * quads;
    printf("P3 quad-uniform 8B  : %8.3f ms  %8.2f G probes/s  requested_bytes=%.0f\n", ms, n / ms / 1e6, n * 8);
  }
  {
    const uint64_t per_lane = acc_m * 1000000ull / lanes;
    float ms = timeit([&] { hipLaunchKernelGGL(quad_gather<4>, dim3(grid), dim3(block), 0, 0, t, nlines, per_lane, sink, salt++); });
    double n = (double)per_lane * lanes;
    printf("P4 per-lane 8B      : %8.3f ms  %8.2f G probes/s  requested_bytes=%.0f\n", ms, n / ms / 1e6, n * 8);
  }
  return 0;
}
