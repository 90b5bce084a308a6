// host_pack.cpp -- persistent host worker pool + AVX2 ASCII -> 2-bit packer (see host_pack.h).
#include "host_pack.h"

#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>

namespace awry {

unsigned effective_cpus() {
  if (const char* e = getenv("AWRY_HOST_THREADS")) {
    const int v = atoi(e);
    if (v >= 1) return (unsigned)std::min(v, 256);
  }
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::max(1, CPU_COUNT(&set));
  auto quota = [&](const char* path_quota, const char* path_period) {
    FILE* f = fopen(path_quota, "r");
    if (!f) return;
    char a[64] = {0}, b[64] = {0};
    const int got = fscanf(f, "%63s %63s", a, b);
    fclose(f);
    long long q = -1, p = 0;
    if (path_period) {  // cgroup v1: two files
      q = atoll(a);
      if (FILE* g = fopen(path_period, "r")) {
        if (fscanf(g, "%63s", b) == 1) p = atoll(b);
        fclose(g);
      }
    } else if (got == 2 && strcmp(a, "max") != 0) {  // cgroup v2: "<quota> <period>"
      q = atoll(a);
      p = atoll(b);
    }
    if (q > 0 && p > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, q / p));
  };
  quota("/sys/fs/cgroup/cpu.max", nullptr);
  quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  return std::max(1u, std::min(n, 64u));
}

// Several jobs may be in flight at once -- one per replica thread of a batch call over N replicas, which pack their
// shards concurrently: a job is a counter over [0, n) posted in a slot; the posting thread works on its own job, and
// every pool thread takes items from whichever posted job still has some, so the pool's threads split themselves between
// the replicas by demand (round 2's pool ran one job at a time, the replicas taking turns).
struct HostPool::Impl {
  struct Job {
    const std::function<void(uint64_t)>* fn = nullptr;
    uint64_t n = 0;
    std::atomic<uint64_t> next{0}, finished{0};
    std::atomic<int> refs{0};  // pool threads inside this job
    std::mutex emu;
    std::exception_ptr err;
  };
  static constexpr int SLOTS = 32;
  std::vector<std::thread> workers;
  std::mutex slot_mu;                 // guards slot[] and the taking of a reference
  Job* slot[SLOTS] = {nullptr};
  std::atomic<uint32_t> posted{0};    // bit s: slot s holds a job (a hint read without the lock)
  std::mutex mu;                      // sleeping workers
  std::condition_variable cv;
  std::atomic<uint64_t> epoch{0};     // bumped whenever a job is posted
  bool stop = false;

  static void work(Job& j) {
    for (;;) {
      const uint64_t i = j.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= j.n) break;
      try {
        (*j.fn)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(j.emu);
        if (!j.err) j.err = std::current_exception();
      }
      j.finished.fetch_add(1, std::memory_order_release);
    }
  }

  Job* take(int s) {  // a reference to the job in slot s if it still has items to hand out
    std::lock_guard<std::mutex> lk(slot_mu);
    Job* j = slot[s];
    if (!j || j->next.load(std::memory_order_relaxed) >= j->n) return nullptr;
    j->refs.fetch_add(1, std::memory_order_acq_rel);
    return j;
  }

  void worker() {
    for (;;) {
      const uint64_t e = epoch.load(std::memory_order_acquire);
      bool did = false;
      const uint32_t hint = posted.load(std::memory_order_acquire);
      for (int s = 0; s < SLOTS; s++)
        if (((hint >> s) & 1u) == 0) continue;
        else if (Job* j = take(s)) {
          work(*j);
          j->refs.fetch_sub(1, std::memory_order_acq_rel);
          did = true;
        }
      if (did) continue;
      // a batch call posts jobs back to back: spin briefly before going to sleep on the condition variable
      for (int spin = 0; spin < 4000 && epoch.load(std::memory_order_acquire) == e; spin++) _mm_pause();
      if (epoch.load(std::memory_order_acquire) != e) continue;
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return stop || epoch.load(std::memory_order_acquire) != e; });
      if (stop) return;
    }
  }
};

// a forked child inherits the pool object but none of its threads: it runs its jobs inline
static std::atomic<bool> g_pool_forked{false};

HostPool::HostPool() : impl_(new Impl) {
  const unsigned t = effective_cpus();
  for (unsigned i = 1; i < t; i++) impl_->workers.emplace_back([this] { impl_->worker(); });
  (void)pthread_atfork(nullptr, nullptr, [] { g_pool_forked.store(true); });
}

HostPool::~HostPool() {
  if (g_pool_forked.load()) return;  // (the threads belong to the parent)
  {
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->stop = true;
    impl_->epoch.fetch_add(1, std::memory_order_release);
  }
  impl_->cv.notify_all();
  for (auto& w : impl_->workers) w.join();
  delete impl_;
}

HostPool& HostPool::instance() {
  static HostPool pool;
  return pool;
}

unsigned HostPool::threads() const { return (unsigned)impl_->workers.size() + 1; }

void HostPool::run(uint64_t n, const std::function<void(uint64_t)>& fn) {
  if (n == 0) return;
  Impl& p = *impl_;
  if (n == 1 || p.workers.empty() || g_pool_forked.load(std::memory_order_relaxed)) {
    for (uint64_t i = 0; i < n; i++) fn(i);
    return;
  }
  Impl::Job job;
  job.fn = &fn;
  job.n = n;
  int mine = -1;
  {
    std::lock_guard<std::mutex> lk(p.slot_mu);
    for (int s = 0; s < Impl::SLOTS && mine < 0; s++)
      if (!p.slot[s]) { p.slot[s] = &job; mine = s; p.posted.fetch_or(1u << s, std::memory_order_release); }
  }
  if (mine < 0) {  // every slot taken (more than 32 concurrent callers): inline
    for (uint64_t i = 0; i < n; i++) fn(i);
    return;
  }
  {
    std::lock_guard<std::mutex> lk(p.mu);
    p.epoch.fetch_add(1, std::memory_order_release);
  }
  p.cv.notify_all();
  Impl::work(job);
  {
    std::lock_guard<std::mutex> lk(p.slot_mu);  // no new references after this
    p.slot[mine] = nullptr;
    p.posted.fetch_and(~(1u << mine), std::memory_order_release);
  }
  // the items other threads are still finishing (the tail of the job: short)
  for (unsigned spins = 0; job.finished.load(std::memory_order_acquire) < n || job.refs.load(std::memory_order_acquire) != 0; spins++) {
    if (spins < 4000) _mm_pause();
    else std::this_thread::yield();  // (a helper was descheduled inside its last item)
  }
  if (job.err) std::rethrow_exception(job.err);
}

void HostPool::run_ranges(uint64_t n, uint64_t grain, const std::function<void(uint64_t, uint64_t)>& fn) {
  if (n == 0) return;
  grain = std::max<uint64_t>(1, grain);
  const uint64_t pieces = (n + grain - 1) / grain;
  run(pieces, [&](uint64_t i) { fn(i * grain, std::min(n, (i + 1) * grain)); });
}

void pool_memcpy(void* dst, const void* src, size_t bytes) {
  if (bytes < (1u << 20)) {
    memcpy(dst, src, bytes);
    return;
  }
  HostPool::instance().run_ranges(bytes, 1u << 19, [&](uint64_t lo, uint64_t hi) {
    memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, hi - lo);
  });
}

void pool_widen_u32(uint64_t* dst, const uint32_t* src, uint64_t n) {
  // streaming stores: the result array is written once and not read here, so no line of it is fetched first
  auto piece = [&](uint64_t lo, uint64_t hi) {
    uint64_t i = lo;
    for (; i < hi && (reinterpret_cast<uintptr_t>(dst + i) & 31); i++) dst[i] = src[i];
    for (; i + 4 <= hi; i += 4)
      _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepu32_epi64(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i))));
    for (; i < hi; i++) dst[i] = src[i];
    _mm_sfence();
  };
  if (n < (1u << 17)) { piece(0, n); return; }
  HostPool::instance().run_ranges(n, 1u << 16, piece);
}

namespace {

// up to 32 letters at p -> 64 bits; *ok is cleared when one of the first m bytes is not in ACGTacgt.
// Two 16-entry tables indexed by the low nibble of the case-folded byte (A x1, C x3, T x4, G x7): the letter that
// nibble stands for -- the byte is valid iff it IS that letter (pshufb yields 0 for bytes >= 0x80, which then differ) --
// and its 2-bit code A0 C1 G2 T3.
inline uint64_t pack32(const uint8_t* p, int m, bool* ok) {
  const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
  const __m256i c = _mm256_and_si256(x, _mm256_set1_epi8((char)0xDF));  // upper-case; bytes >= 0x80 stay >= 0x80
  const __m256i letter = _mm256_setr_epi8(-1, 'A', -1, 'C', 'T', -1, -1, 'G', -1, -1, -1, -1, -1, -1, -1, -1,
                                          -1, 'A', -1, 'C', 'T', -1, -1, 'G', -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i code = _mm256_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
  const uint32_t vm = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(letter, c), c));
  const uint32_t need = m >= 32 ? ~0u : ((1u << m) - 1u);
  if ((vm & need) != need) *ok = false;
  const __m256i y = _mm256_shuffle_epi8(code, c);
  const __m256i t = _mm256_maddubs_epi16(y, _mm256_set1_epi16(0x0401));  // b0 + 4 b1: 4 bits per 16-bit lane
  const __m256i u = _mm256_madd_epi16(t, _mm256_set1_epi32(0x00100001));  // w0 + 16 w1: 8 bits per 32-bit lane
  const __m256i s = _mm256_shuffle_epi8(u, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
  const __m128i lo = _mm256_castsi256_si128(s), hi = _mm256_extracti128_si256(s, 1);
  const uint64_t w = (uint64_t)_mm_cvtsi128_si64(_mm_unpacklo_epi32(lo, hi));
  return m >= 32 ? w : (w & ((1ull << (2 * m)) - 1));
}

// Two k-mers per step with AVX-512 (BW + VL; Zen 4 / Zen 5 and recent Xeons): the 32 bytes at p0 and at p1 side by side in
// one register, the same table lookups and multiply-adds at twice the width, the four result dwords compacted into two
// consecutive 64-bit words.  Bit i of the returned mask is set when byte i (0..31: first k-mer, 32..63: second) is valid.
__attribute__((target("avx512f,avx512bw,avx512vl,avx512dq"))) inline uint64_t pack32x2_avx512(const uint8_t* p0, const uint8_t* p1, int m,
                                                                                             uint64_t* out2) {
  const __m512i x = _mm512_inserti64x4(_mm512_castsi256_si512(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(p0))),
                                       _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p1)), 1);
  const __m512i c = _mm512_and_si512(x, _mm512_set1_epi8((char)0xDF));
  const __m512i letter = _mm512_broadcast_i32x4(_mm_setr_epi8(-1, 'A', -1, 'C', 'T', -1, -1, 'G', -1, -1, -1, -1, -1, -1, -1, -1));
  const __m512i code = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0));
  const uint64_t valid = _mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(letter, c), c);
  const __m512i y = _mm512_shuffle_epi8(code, c);
  const __m512i t = _mm512_maddubs_epi16(y, _mm512_set1_epi16(0x0401));
  const __m512i u = _mm512_madd_epi16(t, _mm512_set1_epi32(0x00100001));
  const __m512i s = _mm512_shuffle_epi8(u, _mm512_broadcast_i32x4(_mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1)));
  // dword 0 of each 128-bit lane -> dwords 0..3: first k-mer = lanes 0, 1, second = lanes 2, 3
  const __m512i g = _mm512_permutexvar_epi32(_mm512_setr_epi32(0, 4, 8, 12, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), s);
  __m128i w = _mm512_castsi512_si128(g);
  if (m < 32) w = _mm_and_si128(w, _mm_set1_epi64x((long long)((1ull << (2 * m)) - 1)));
  if ((reinterpret_cast<uintptr_t>(out2) & 15) == 0) _mm_stream_si128(reinterpret_cast<__m128i*>(out2), w);
  else { _mm_stream_si64(reinterpret_cast<long long*>(out2), _mm_cvtsi128_si64(w)); _mm_stream_si64(reinterpret_cast<long long*>(out2 + 1), _mm_extract_epi64(w, 1)); }
  return valid;
}

static const bool g_have_avx512 = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq") &&
                                  !(getenv("AWRY_HOST_AVX512") && !strcmp(getenv("AWRY_HOST_AVX512"), "0"));

// one query of `len` letters at p -> W words; returns false when it holds a byte outside ACGTacgt
inline bool pack_query(const uint8_t* p, uint64_t len, const uint8_t* end, uint64_t* out, uint64_t W) {
  bool ok = true;
  uint64_t k = 0;
  for (uint64_t j = 0; j < len; j += 32, k++) {
    const int m = (int)std::min<uint64_t>(32, len - j);
    if (p + j + 32 <= end) {
      _mm_stream_si64(reinterpret_cast<long long*>(out + k), (long long)pack32(p + j, m, &ok));
    } else {  // the last bytes of the buffer: a padded copy
      alignas(32) uint8_t tmp[32];
      memset(tmp, 'A', 32);
      memcpy(tmp, p + j, (size_t)m);
      _mm_stream_si64(reinterpret_cast<long long*>(out + k), (long long)pack32(tmp, m, &ok));
    }
  }
  for (; k < W; k++) _mm_stream_si64(reinterpret_cast<long long*>(out + k), 0);  // (every word of a line by the same kind of store)
  return ok;
}

}  // namespace

bool pack_nt2_host(const uint8_t* ascii, const uint8_t* ascii_end, const uint64_t* off, uint64_t lo, uint64_t hi, uint64_t L,
                   uint64_t* words, uint32_t* lens, std::vector<uint32_t>& bad, const uint64_t* check_off) {
  const uint64_t n = hi - lo, W = (L + 31) / 32;
  bad.clear();
  if (n == 0) return true;
  std::mutex bad_mu;
  std::atomic<int> uneven{0};
  const uint64_t base = off ? off[lo] : 0;
  // waking the pool costs ~40 us (its workers sleep between jobs): a batch one thread packs faster than that stays here
  const uint64_t grain = n * W <= 16384 ? n : 8192;
  HostPool::instance().run_ranges(n, grain, [&](uint64_t a, uint64_t b) {
    uint32_t local[64];
    int nl = 0;
    auto flush = [&] {
      std::lock_guard<std::mutex> lk(bad_mu);
      bad.insert(bad.end(), local, local + nl);
      nl = 0;
    };
    if (check_off) {  // the assumed length, checked against the batch's offsets (branch-free: vectorises)
      uint64_t diff = 0;
      const uint64_t* o = check_off + lo;
      for (uint64_t q = a; q < b; q++) diff |= (o[q + 1] - o[q]) ^ L;
      if (diff) { uneven.store(1, std::memory_order_relaxed); return; }
    }
    if (!off && L <= 32) {  // k-mers: one load, one word per query
      const uint8_t* p = ascii + a * L;
      const int m = (int)L;
      uint64_t q = a;
      if (g_have_avx512) {  // two k-mers per step
        const uint64_t need1 = m >= 32 ? 0xFFFFFFFFull : ((1ull << m) - 1), need = need1 | (need1 << 32);
        for (; q + 2 <= b && p + L + 32 <= ascii_end; q += 2, p += 2 * L) {
          _mm_prefetch(reinterpret_cast<const char*>(p + 1024), _MM_HINT_T0);
          const uint64_t valid = pack32x2_avx512(p, p + L, m, words + q);
          if ((valid & need) != need) {
            if ((valid & need1) != need1) { local[nl++] = (uint32_t)q; if (nl == 64) flush(); }
            if (((valid >> 32) & need1) != need1) { local[nl++] = (uint32_t)(q + 1); if (nl == 64) flush(); }
          }
        }
      }
      for (; q < b; q++, p += L) {
        _mm_prefetch(reinterpret_cast<const char*>(p + 1024), _MM_HINT_T0);
        bool ok = true;
        if (p + 32 <= ascii_end) _mm_stream_si64(reinterpret_cast<long long*>(words + q), (long long)pack32(p, m, &ok));  // written once, read by the DMA engine
        else ok = pack_query(p, L, ascii_end, words + q, 1);
        if (!ok) { local[nl++] = (uint32_t)q; if (nl == 64) flush(); }
      }
    } else {
      for (uint64_t q = a; q < b; q++) {
        const uint64_t s = off ? off[lo + q] - base : q * L, len = off ? off[lo + q + 1] - off[lo + q] : L;
        if (lens) lens[q] = (uint32_t)len;
        if (!pack_query(ascii + s, len, ascii_end, words + q * W, W)) { local[nl++] = (uint32_t)q; if (nl == 64) flush(); }
      }
    }
    if (nl) flush();
    _mm_sfence();
  });
  if (bad.size() > 1) std::sort(bad.begin(), bad.end());
  return uneven.load() == 0;
}

}  // namespace awry
