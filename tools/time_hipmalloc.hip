// how long hipMalloc / hipFree of seed-table-sized buffers take on this driver (the seed table of a GRCh38-scale index is
// 137 GB + 34 GB of build scratch).  build: hipcc --offload-arch=gfx950 -O2 tools/time_hipmalloc.hip -o tools/bin/time_hipmalloc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(unsigned long long* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i;
}
int main() {
  (void)hipFree(0);
  const size_t gb = 1ull << 30;
  for (int rep = 0; rep < 1; rep++)
    for (size_t s : {1 * gb, 34 * gb, 48 * gb, 64 * gb, 65 * gb, 80 * gb, 96 * gb, 128 * gb, 137 * gb}) {
      void* p = nullptr;
      double t0 = now();
      hipError_t e = hipMalloc(&p, s);
      double t1 = now();
      if (e != hipSuccess) { printf("%zu GB: malloc failed\n", s / gb); continue; }
      touch<<<4096, 256>>>((unsigned long long*)p, s / 8);
      (void)hipDeviceSynchronize();
      double t2 = now();
      touch<<<4096, 256>>>((unsigned long long*)p, s / 8);
      (void)hipDeviceSynchronize();
      double t3 = now();
      (void)hipFree(p);
      double t4 = now();
      printf("rep %d  %3zu GB: hipMalloc %.3f s, first touch (write all) %.3f s, second write %.3f s, hipFree %.3f s\n", rep, s / gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
      fflush(stdout);
    }
  // one virtual range backed by several physical allocations (hipMemCreate / hipMemMap)
  {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    printf("vmm granularity %zu (%s)\n", gran, hipGetErrorString(e));
    const size_t total = 137 * gb, nchunk = 8, chunk = (total / nchunk + gran - 1) / gran * gran;
    double t0 = now();
    void* va = nullptr;
    e = hipMemAddressReserve(&va, chunk * nchunk, 0, nullptr, 0);
    printf("reserve: %s\n", hipGetErrorString(e));
    hipMemGenericAllocationHandle_t h[8];
    bool ok = e == hipSuccess;
    for (size_t c = 0; c < nchunk && ok; c++) {
      e = hipMemCreate(&h[c], chunk, &prop, 0);
      if (e != hipSuccess) { printf("create %zu: %s\n", c, hipGetErrorString(e)); ok = false; break; }
      e = hipMemMap((char*)va + c * chunk, chunk, 0, h[c], 0);
      if (e != hipSuccess) { printf("map %zu: %s\n", c, hipGetErrorString(e)); ok = false; break; }
    }
    if (ok) {
      hipMemAccessDesc acc{};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      e = hipMemSetAccess(va, chunk * nchunk, &acc, 1);
      printf("set access: %s\n", hipGetErrorString(e));
      double t1 = now();
      touch<<<4096, 256>>>((unsigned long long*)va, total / 8);
      e = hipDeviceSynchronize();
      double t2 = now();
      touch<<<4096, 256>>>((unsigned long long*)va, total / 8);
      (void)hipDeviceSynchronize();
      double t3 = now();
      printf("vmm 137 GB in %zu chunks: reserve+create+map+access %.3f s, first write %.3f s (%s), second write %.3f s\n", nchunk, t1 - t0, t2 - t1, hipGetErrorString(e), t3 - t2);
      (void)hipMemUnmap(va, chunk * nchunk);
      for (size_t c = 0; c < nchunk; c++) (void)hipMemRelease(h[c]);
      (void)hipMemAddressFree(va, chunk * nchunk);
      printf("released in %.3f s\n", now() - t3);
    }
  }
  return 0;
}
