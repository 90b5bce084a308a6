// What a fresh result array costs the host path: first-touch page faults of an untouched anonymous mapping, written by T
// threads -- plain 4 KB pages, with MADV_HUGEPAGE, with MADV_POPULATE_WRITE issued per thread range first -- against writing
// into pages that are already there.  build: g++ -O2 -pthread tools/time_first_touch.cpp -o tools/bin/time_first_touch
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void fill(char* p, size_t bytes, int T, int mode) {
  std::vector<std::thread> th;
  const size_t per = (bytes / T + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; t++)
    th.emplace_back([=] {
      const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, per * (t + 1));
      if (hi <= lo) return;
      if (mode == 2) (void)madvise(p + lo, hi - lo, MADV_POPULATE_WRITE);
      memset(p + lo, 1, hi - lo);
    });
  for (auto& x : th) x.join();
}
int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 160) << 20;
  const int T = argc > 2 ? atoi(argv[2]) : 16;
  const char* names[] = {"4 KB pages", "MADV_HUGEPAGE", "MADV_POPULATE_WRITE per thread range", "MADV_HUGEPAGE + POPULATE_WRITE"};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 4; mode++) {
      char* p = (char*)mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      char* a = (char*)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
      if (mode == 1 || mode == 3) (void)madvise(a, bytes, MADV_HUGEPAGE);
      double t0 = now();
      fill(a, bytes, T, mode == 3 ? 2 : mode);
      double t1 = now();
      fill(a, bytes, T, 0);
      double t2 = now();
      munmap(p, bytes + (2u << 20));
      double t3 = now();
      printf("%-40s %zu MB, %d threads: first write %.2f ms, second write %.2f ms, munmap %.2f ms\n", names[mode], bytes >> 20, T, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
  return 0;
}
