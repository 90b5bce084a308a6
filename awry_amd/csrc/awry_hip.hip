// awry_hip.hip -- C ABI (include/awry_hip.h) over the gfx950 kernels: device replicas, batch drivers,
// seed-table construction.  There is deliberately no CPU implementation of count / locate here.
#include "../../include/awry_hip.h"

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "alphabet.h"
#include "host_index.h"
#include "host_pack.h"
#include "kernels.hip.h"
#include "sais.hpp"

using namespace awry;

namespace {

thread_local std::string g_last_error;

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };
struct QueryError : std::runtime_error { using std::runtime_error::runtime_error; };
struct NoDeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIP_CHECK(expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                     std::to_string(__LINE__) + ")");                                                \
  } while (0)

template <class F>
int guarded(F&& fn) {
  try {
    fn();
    return AWRY_OK;
  } catch (const HipError& e) { g_last_error = e.what(); return AWRY_ERR_HIP;
  } catch (const ArgError& e) { g_last_error = e.what(); return AWRY_ERR_ARG;
  } catch (const QueryError& e) { g_last_error = e.what(); return AWRY_ERR_INVALID_QUERY;
  } catch (const NoDeviceError& e) { g_last_error = e.what(); return AWRY_ERR_NO_DEVICE;
  } catch (const std::bad_alloc&) { g_last_error = "out of host memory"; return AWRY_ERR_OOM;
  } catch (const std::invalid_argument& e) { g_last_error = e.what(); return AWRY_ERR_FORMAT;
  } catch (const std::exception& e) { g_last_error = e.what(); return AWRY_ERR_IO;
  } catch (...) { g_last_error = "unknown error"; return AWRY_ERR_IO; }
}

template <class T>
struct DevBuf {  // RAII device allocation on the current device
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept { reset(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; return *this; }
  ~DevBuf() { reset(); }
  void alloc(size_t count) {
    reset();
    n = count;
    if (count) {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
      if (e != hipSuccess) { p = nullptr; n = 0; throw HipError(std::string("hipMalloc failed: ") + hipGetErrorString(e)); }
    }
  }
  void reset() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

template <class T>
struct PinBuf {  // pinned host staging, grows on demand
  T* p = nullptr;
  size_t cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t c = n + n / 4 + 1024;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), c * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; throw HipError(std::string("hipHostMalloc failed: ") + hipGetErrorString(e)); }
    cap = c;
  }
};

// one pipeline lane of the packed host path (count_shard_packed): buffers persist in the replica and only grow
struct PackedLane {
  hipStream_t s = nullptr;  // owned by the replica
  hipEvent_t done = nullptr;
  hipEvent_t ev_in = nullptr, ev_k = nullptr;  // copy-in stream -> lane stream, lane stream -> copy-out stream (Replica::copy_in / copy_out)
  DevBuf<uint8_t> ascii;
  DevBuf<uint64_t> words, counts, off;  // off / lens: batches of unequal lengths
  DevBuf<uint32_t> lens, bad_list;      // bad_list: the chunk's queries with bytes outside ACGT
  DevBuf<uint8_t> status;               // generic kernel: per-query status of the chunk,
  PinBuf<uint8_t> h_status;             //   and where the host reads it
  DevBuf<unsigned long long> bad;       // [0] number of listed queries, [1] first rejected query (index << 8 | status) or ~0
  unsigned long long* h_bad = nullptr;  // pinned copy of both
  // host-packed path (count_shard_packed): persistent pinned staging, so that no caller memory is ever registered --
  // packed words in, counts out, and the compact copy (indices, offsets, bytes) of the chunk's queries with other letters
  PinBuf<uint64_t> h_words, h_boff;
  PinBuf<uint32_t> h_counts32, h_lens, h_bq;  // counts cross PCIe as 32-bit words and are widened into counts_out
  DevBuf<uint32_t> counts32;
  PinBuf<uint8_t> h_bbytes;
  DevBuf<uint8_t> bbytes;
  DevBuf<uint64_t> boff;
  uint64_t nbad = 0;
  uint64_t chunk_lo = 0, chunk_hi = 0;
  bool busy = false;
  ~PackedLane() {
    if (done) (void)hipEventDestroy(done);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_k) (void)hipEventDestroy(ev_k);
    if (h_bad) (void)hipHostFree(h_bad);
  }
};

// one pipeline lane of the packed host locate path (locate_shard_packed); persists in the replica
struct LocateLane {
  hipEvent_t counted = nullptr, located = nullptr;
  hipEvent_t ev_in = nullptr, ev_k = nullptr;  // copy-in stream -> lane stream, lane stream -> copy-out stream (Replica::copy_in / copy_out)
  DevBuf<uint8_t> ascii;
  DevBuf<uint64_t> words, rstart, counts, hit_off, scratch, gpos, pos, off;
  DevBuf<uint32_t> lens, bad_list;
  DevBuf<uint8_t> status;                      // generic kernel: per-query status of the chunk,
  PinBuf<uint8_t> h_status;                    //   and where the host reads it
  DevBuf<unsigned long long> bad;              // [0] reads with bytes outside ACGT, [1] first rejected read (index << 8 | status) or ~0
  PinBuf<uint64_t> h_counts, h_gpos, h_meta;  // h_meta: [0] total hits of the chunk, [1..2] copy of `bad`
  PinBuf<awry_pos_t> h_pos;
  // host-packed reads: pinned staging of the packed words / lengths and the compact copy of the reads with other letters
  PinBuf<uint64_t> h_words, h_boff;
  PinBuf<uint32_t> h_lens, h_bq;
  PinBuf<uint8_t> h_bbytes;
  DevBuf<uint8_t> bbytes;
  DevBuf<uint64_t> boff;
  uint64_t lo = 0, hi = 0, total = 0;
  int stage = 0;                               // 0 idle, 1 count queued, 2 locate queued
  ~LocateLane() {
    if (counted) (void)hipEventDestroy(counted);
    if (located) (void)hipEventDestroy(located);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_k) (void)hipEventDestroy(ev_k);
  }
};

struct Replica {
  int device = -1;
  hipStream_t stream = nullptr;
  static constexpr int NLANES = 3;
  hipStream_t lane_stream[NLANES] = {nullptr, nullptr, nullptr};  // the pipeline lanes of the host paths (locate uses two)
  // All chunk copies of the host-packed count path go through these two, one per direction, tied to the lanes' kernels by
  // events.  With the copies on the lane streams themselves, three streams copied at once, and after an accelerator rebuild
  // (or on a second replica) ONE of them was left on a copy path 2-3x slower (chunk in: 75-150 -> 250-300 us, out: 40-80 ->
  // 160-200 us; rocprofv3 --memory-copy-trace, profiles/r03a1_*), which then set the pace of every call: 1.45 -> 2.25 ms per
  // 5 M 31-mers, for good.  PCIe is the limit either way and one stream per direction sustains it.
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  PackedLane lanes[NLANES];
  LocateLane loc_lanes[2];
  std::mutex lane_mu;  // one packed host call at a time per replica
  // single-query calls (count_string, search_range): a pinned mailbox the generic kernel reads and writes in place --
  // one launch and one stream synchronisation per call, no device allocation, no copies
  struct Mailbox {
    static constexpr size_t QCAP = 1 << 16;
    static constexpr size_t HCAP = 1 << 15;  // hits a single-query locate returns through the mailbox
    uint8_t* q = nullptr;       // [QCAP + 16]
    uint64_t* words = nullptr;  // off[2], count, range[2], status, hit_off[2]
    uint64_t* gpos = nullptr;   // [HCAP]
    uint64_t* pos = nullptr;    // [2 * HCAP]
    ~Mailbox() {
      for (void* p : {(void*)q, (void*)words, (void*)gpos, (void*)pos})
        if (p) (void)hipHostFree(p);
    }
  } mailbox;
  std::mutex mailbox_mu;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBuf<uint64_t> blocks, sa_words, seq_starts;
  DevBuf<uint32_t> seq_bucket;  // DevIndex::seq_bucket (indexes of more records than the locate kernels keep in LDS)
  DevBuf<SeedEntry> seed;
  DevBuf<SeedEntry64> seed64;  // wide-row replicas (bwt_len >= 2^32, or forced): 16-byte entries
  // seed tables for k-mers SHORTER than the main table's k ("rungs": one complete 4^L table per query length L that has
  // been asked for, built on first use; a 12-mer is then answered by its entry instead of 12 LF steps)
  std::map<int, DevBuf<SeedEntry>> rungs;
  std::set<int> rungs_refused;  // lengths whose table did not fit the HBM budget when first asked for
  std::mutex rung_mu;
  bool wide = false;           // 64-bit rows: wide kernels, no 32-bit accelerators
  DevBuf<uint32_t> text4;                     // 4-bit text for seed-and-verify (device-only accelerator)
  DevBuf<uint8_t> text8;                      // the text as symbol indices, for the generic kernel's verify (any alphabet)
  DevBuf<uint32_t> dense_sa;                  // SA[j * dense_ratio] as u32 (device-only accelerator for locate)
  DevBuf<uint32_t> sa_nblock;                 // SA of the rows whose suffix starts with N (kept while locate has to walk)
  DevBuf<uint64_t> lcx_key, lcx_rowpos, lcx_inner;  // left-context index (layout.h, DevIndex::lcx_key); kept with position seeds
  uint32_t dense_ratio = 0;                   // 0 = use the file's bit-packed samples
  bool verify_kmers = false;                  // also use seed-and-verify in the k-mer (L <= 32) kernel
  // survivor lists of the two-phase count schedule, one per stream (launches on one stream are ordered, so reuse is safe)
  struct SurvScratch {
    DevBuf<uint64_t> w, range;
    DevBuf<uint32_t> q, count;
    uint64_t cap = 0, cap_q = 0;
    // what lcx_quad_reads_kernel leaves for the LF pass: one device-wide list (fcount[0] slots)
    DevBuf<uint64_t> fw, frange;
    DevBuf<uint32_t> fq, fcount;
    uint64_t fcap = 0;
    // awry_dev_count_ascii_uniform on a nucleotide index: packed words of the batch, the pack kernel's list of queries with
    // other letters and its counter
    DevBuf<uint64_t> u_words;
    DevBuf<uint32_t> u_list;
    DevBuf<unsigned long long> u_bad;
    // work-queue heads of the chunk / locate kernels launched on this stream: launches on one stream are ordered, so a
    // head is free again by the time the ring comes back to it, however many launches other streams have in flight
    DevBuf<unsigned long long> counters;
    unsigned counter_seq = 0;
  };
  std::mutex scratch_mu;
  std::map<hipStream_t, std::unique_ptr<SurvScratch>> scratch;
  int seed_k = 0;
  int num_cus = 256;
  DevIndex dev{};
  ~Replica() {
    if (device >= 0) {
      (void)hipSetDevice(device);
      if (stream) (void)hipStreamDestroy(stream);
      for (auto& ls : lane_stream) if (ls) (void)hipStreamDestroy(ls);
      if (copy_in) (void)hipStreamDestroy(copy_in);
      if (copy_out) (void)hipStreamDestroy(copy_out);
      if (ev0) (void)hipEventDestroy(ev0);
      if (ev1) (void)hipEventDestroy(ev1);
      blocks.reset(); sa_words.reset(); seq_starts.reset(); seed.reset(); seed64.reset(); rungs.clear(); dense_sa.reset(); text4.reset();
      scratch.clear();
      sa_nblock.reset(); text8.reset();
      lcx_key.reset(); lcx_rowpos.reset(); lcx_inner.reset();
    }
  }
};

}  // namespace

struct awry_index {
  HostIndex host;
  std::vector<std::unique_ptr<Replica>> reps;
  int seed_k_request = -1;      // -1 = default policy
  int dense_ratio_request = 0;  // 0 = locate walks to the file's SA samples
  int verify_request = -2;      // -2: policy; -1: seed-and-verify off; >= 0: LF steps before switching to text comparison
  bool verify_kmers_request = false;
  int lcx_request = -1;         // -1: policy (on when it fits); 0: no left-context index; 1: as -1
};

namespace {

void require(bool ok, const char* msg) { if (!ok) throw ArgError(msg); }

Replica& replica(awry_index* ix, int slot) {
  require(ix != nullptr, "null index");
  if (ix->reps.empty()) throw NoDeviceError("no device replica: call awry_set_devices() first (there is no CPU search path)");
  require(slot >= 0 && slot < (int)ix->reps.size(), "replica slot out of range");
  Replica& r = *ix->reps[slot];
  HIP_CHECK(hipSetDevice(r.device));
  return r;
}

int grid_for(const Replica& r, uint64_t work_items, int per_block, int blocks_per_cu = 8) {
  uint64_t want = (work_items + per_block - 1) / per_block;
  uint64_t cap = (uint64_t)r.num_cus * blocks_per_cu;
  return (int)std::max<uint64_t>(1, std::min(want, cap));
}

// which instantiation serves awry_dev_count_nt2: 0 strided quads, 1 LDS-staged chunks, 2 groups of four
std::atomic<int> g_count_kernel{-1};
// -1: no explicit choice (env / policy)
int count_kernel_override() {
  int m = g_count_kernel.load();
  if (m >= 0) return m;
  const char* e = getenv("AWRY_COUNT_KERNEL");
  if (e && !strcmp(e, "strided")) return 0;
  if (e && !strcmp(e, "chunk")) return 1;
  if (e && !strcmp(e, "quad4")) return 2;
  if (e && !strcmp(e, "twophase")) return 3;
  return -1;
}
// Policy, from measurements on MI355X, GRCh38-scale, 10 M random 31-mers per launch (tools/ab_count.py, G queries/s):
//   seed k   strided  quad4  chunk  twophase
//     14      10.1    10.3    9.9     8.2
//     16      19.4    21.8   14.4    21.3
//     17      25.3    28.1     -     29.9
// quad4 is the general default; once the table is so sparse that most queries are decided by their entry alone
// (4^k >= 3 bwt_len) the per-lane probe pass of the two-phase schedule wins.
int count_kernel_mode(uint64_t bwt_len, int seed_k, bool seeded) {
  const int m = count_kernel_override();
  if (m >= 0) return m;
  if (seeded && seed_k >= 1 && seed_k <= 31 && (1ull << (2 * seed_k)) / 3 >= bwt_len) return 3;
  return 2;
}

// Rows fit 32 bits: the packed kernels, seed entries and accelerators are the 32-bit ones.  awry_debug_force_wide_rows(1)
// makes replicas built afterwards take the wide-row (64-bit) kernels whatever their size -- how those kernels are tested,
// since an index of 2^32 rows takes an hour of host SA-IS to build (the GPU builder stops below 2^32).
std::atomic<int> g_force_wide{0};
bool narrow(const HostIndex& h) { return h.bwt_len < (1ull << 32) - 512 && !g_force_wide.load(); }

// HBM the accelerator policies may plan with on the current device: what is free now, capped by AWRY_HBM_BUDGET_GB (a
// process that shares the GPU, or wants room for its own buffers, sets it; the seed table is sized to 70 % and the
// verify accelerators admitted below 50 % of this figure)
bool hbm_budget(size_t* free_out) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  if (const char* e = getenv("AWRY_HBM_BUDGET_GB")) {
    const double gb = atof(e);
    if (gb > 0) free_b = std::min<size_t>(free_b, (size_t)(gb * 1e9));
  }
  *free_out = free_b;
  return true;
}

int default_seed_k(const HostIndex& h) {
  if (!narrow(h)) {  // wide rows: nucleotide only, 16-byte entries (+ 4 B scratch per entry while building)
    if (h.alphabet != NUCLEOTIDE) return 0;
    if (const char* e = getenv("AWRY_SEED_K")) return std::max(0, std::min(17, atoi(e)));
    int k = std::min(17, (int)std::floor(std::log((double)h.bwt_len) / std::log(4.0)) + 2);
    size_t free_b = 0;
    if (hbm_budget(&free_b))
      while (k > 1 && 20.0 * std::pow(4.0, k) > 0.7 * (double)free_b) k--;
    return std::max(1, k);
  }
  const bool nt = h.alphabet == NUCLEOTIDE;
  if (const char* e = getenv("AWRY_SEED_K")) return std::max(0, std::min(nt ? 17 : 7, atoi(e)));
  if (!nt) {  // amino: 20^k ~ 1..20 x bwt_len (Swiss-Prot 9e7 -> k = 7, 10 GB), same memory rule as below
    int k = (int)std::floor(std::log((double)h.bwt_len) / std::log(20.0)) + 1;
    k = std::max(1, std::min(k, 7));
    size_t free_b = 0;
    if (hbm_budget(&free_b))
      while (k > 1 && 8.5 * std::pow((double)AA_SEED_SIGMA, k) > 0.7 * (double)free_b) k--;
    return k;
  }
  // A dozen table entries per suffix or more: the smallest k with 4^k >= 12 x bwt_len, at most 17 (GRCh38: 17, 137 GB of the
  // 288 GB HBM and 5.5 entries per suffix -- 18 would not fit; chr1: 16, 34 GB; E. coli: 13).  A random k-mer's entry is
  // then empty or a singleton whose BWT symbol rarely matches, so a query costs one probe plus ~0.1 steps instead of ~16
  // steps (27 block reads), and a k-mer from the text rarely shares its seed with another one.  (Round 1 took
  // floor(log4 bwt_len) + 2, i.e. 4..16 entries per suffix; chr1 sat at the low end of that with k = 15: k = 16 counts
  // random 31-mers 8 %, 31-mers from the text 36 % and 101-bp reads 18 % faster.)  The table and its build scratch (1/4 of
  // it) must fit in 70 % of the free HBM, else k drops.
  int k = 1;
  while (k < 17 && (double)(1ull << (2 * k)) < 12.0 * (double)h.bwt_len) k++;
  size_t free_b = 0;
  if (hbm_budget(&free_b))
    while (k > 1 && (double)(10ull << (2 * k)) > 0.7 * (double)free_b) k--;  // 8 B + 2 B scratch per entry
  return k;
}

// the complete sigma^k table of (first row, count + BWT symbol of a singleton) entries for a 32-bit-row replica, built
// level by level on the replica's own stream (see seed_extend_kernel); synchronous
void build_seed_table(Replica& r, bool nt, int k, DevBuf<SeedEntry>& out) {
  const uint64_t sigma = nt ? 4 : AA_SEED_SIGMA;
  uint64_t nfinal = 1;
  for (int j = 0; j < k; j++) nfinal *= sigma;
  DevBuf<SeedEntry> a(nfinal), b(std::max<uint64_t>(sigma, nfinal / sigma));
  // level j lands in `a` when (k - j) is even, so the last level is in `a`
  SeedEntry* cur = ((k - 1) % 2 == 0) ? a.p : b.p;
  if (nt) hipLaunchKernelGGL(seed_level1_kernel, dim3(1), dim3(256), 0, r.stream, r.dev, cur);
  else hipLaunchKernelGGL(aa_seed_level1_kernel, dim3(1), dim3(256), 0, r.stream, r.dev, cur);
  uint64_t nchild = sigma;
  for (int j = 2; j <= k; j++) {
    SeedEntry* nxt = ((k - j) % 2 == 0) ? a.p : b.p;
    nchild *= sigma;
    if (nt) hipLaunchKernelGGL(seed_extend_kernel, dim3(grid_for(r, nchild * 4, 256)), dim3(256), 0, r.stream, r.dev, cur, nxt, nchild);
    else hipLaunchKernelGGL(aa_seed_extend_kernel, dim3(grid_for(r, nchild, 256)), dim3(256), 0, r.stream, r.dev, cur, nxt, nchild);
    cur = nxt;
  }
  if (nt) hipLaunchKernelGGL(seed_finalize_kernel, dim3(grid_for(r, nfinal, 256)), dim3(256), 0, r.stream, r.dev, a.p, nfinal);
  else hipLaunchKernelGGL(aa_seed_finalize_kernel, dim3(grid_for(r, nfinal, 256)), dim3(256), 0, r.stream, r.dev, a.p, nfinal);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(r.stream));
  out = std::move(a);
}

// The table for nucleotide k-mers of L < seed_k letters, built on first use and kept.  nullptr: not available (it does not
// fit the HBM budget, or AWRY_SEED_RUNGS=0) -- the caller falls back to LF steps from the last letter.
constexpr int SEED_RUNG_MIN = 6;  // shorter k-mers: a handful of LF steps over blocks that live in L2
const SeedEntry* seed_rung(Replica& r, int L) {
  static const bool off = getenv("AWRY_SEED_RUNGS") && !strcmp(getenv("AWRY_SEED_RUNGS"), "0");
  if (off || r.wide || r.dev.alphabet != NUCLEOTIDE || L < SEED_RUNG_MIN || L > 16) return nullptr;
  std::lock_guard<std::mutex> lock(r.rung_mu);
  auto it = r.rungs.find(L);
  if (it != r.rungs.end()) return it->second.p;
  if (r.rungs_refused.count(L)) return nullptr;
  // 8 B per entry + a quarter of that while building must fit half of what is free now (AWRY_HBM_BUDGET_GB caps that figure),
  // and all rungs of a replica together stay below AWRY_SEED_RUNG_GB (default 48: every length 6..16 at once would be 46 GB)
  static const double rung_cap = [] { const char* e = getenv("AWRY_SEED_RUNG_GB"); return (e && atof(e) > 0 ? atof(e) : 48.0) * 1e9; }();
  double held = 0;
  for (const auto& kv : r.rungs) held += 8.0 * (double)kv.second.n;
  size_t free_b = 0;
  if (!hbm_budget(&free_b) || (double)(10ull << (2 * L)) > 0.5 * (double)free_b || held + (double)(8ull << (2 * L)) > rung_cap) {
    r.rungs_refused.insert(L);
    return nullptr;
  }
  int cur_dev = 0;
  HIP_CHECK(hipGetDevice(&cur_dev));
  if (cur_dev != r.device) HIP_CHECK(hipSetDevice(r.device));
  struct Restore { int dev, mine; ~Restore() { if (dev != mine) (void)hipSetDevice(dev); } } restore{cur_dev, r.device};  // the caller's device
  DevBuf<SeedEntry> t;
  try {
    build_seed_table(r, true, L, t);
  } catch (const HipError&) {  // (hipMalloc: the budget was an estimate)
    (void)hipGetLastError();
    r.rungs_refused.insert(L);
    return nullptr;
  }
  const SeedEntry* p = t.p;
  r.rungs.emplace(L, std::move(t));
  return p;
}

// level-by-level seed table on the replica's device (see seed_extend_kernel)
void drop_lcx(Replica& r) {
  r.lcx_key.reset(); r.lcx_rowpos.reset(); r.lcx_inner.reset();
  r.dev.lcx_key = r.dev.lcx_rowpos = r.dev.lcx_inner = nullptr;
  for (auto& o : r.dev.lcx_off) o = 0;
}

void build_seed(awry_index* ix, Replica& r, int k) {
  drop_lcx(r);  // its flags live in the table's entries
  r.seed.reset();
  r.seed_k = 0;
  r.dev.seed = nullptr;
  r.dev.seed_k = 0;
  r.dev.seed_pos = 0;
  r.dev.ctx_extra = 0;
  r.seed64.reset();
  r.dev.seed64 = nullptr;
  if (k <= 0) return;
  if (r.wide) {  // 64-bit rows: 16-byte entries, nucleotide only
    require(ix->host.alphabet == NUCLEOTIDE, "a wide-row seed table needs a nucleotide index");
    require(k <= 17, "seed k-mer length must be <= 17");
    const uint64_t nfinal = 1ull << (2 * k);
    DevBuf<SeedEntry64> a(nfinal), b(std::max<uint64_t>(4, nfinal / 4));
    SeedEntry64* cur = ((k - 1) % 2 == 0) ? a.p : b.p;
    hipLaunchKernelGGL(seed64_level1_kernel, dim3(1), dim3(256), 0, r.stream, r.dev, cur);
    uint64_t nchild = 4;
    for (int j = 2; j <= k; j++) {
      SeedEntry64* nxt = ((k - j) % 2 == 0) ? a.p : b.p;
      nchild *= 4;
      hipLaunchKernelGGL(seed64_extend_kernel, dim3(grid_for(r, nchild * 4, 256)), dim3(256), 0, r.stream, r.dev, cur, nxt, nchild);
      cur = nxt;
    }
    hipLaunchKernelGGL(seed64_finalize_kernel, dim3(grid_for(r, nfinal, 256)), dim3(256), 0, r.stream, r.dev, a.p, nfinal);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(r.stream));
    r.seed64 = std::move(a);
    r.seed_k = k;
    r.dev.seed64 = r.seed64.p;
    r.dev.seed_k = k;
    return;
  }
  require(narrow(ix->host), "seed table needs an index with bwt_len < 2^32");
  const bool nt = ix->host.alphabet == NUCLEOTIDE;
  require(k <= (nt ? 17 : 7), "seed k-mer length must be <= 17 (nucleotide) / 7 (amino)");
  DevBuf<SeedEntry> a;
  build_seed_table(r, nt, k, a);
  r.seed = std::move(a);
  r.seed_k = k;
  r.dev.seed = r.seed.p;
  r.dev.seed_k = k;
}

void build_dense_sa(awry_index* ix, Replica& r, int ratio);
void build_verify(awry_index* ix, Replica& r, int after_steps);
void sync_seed_mode(awry_index* ix, Replica& r);
void refresh_nblock(awry_index* ix, Replica& r);

bool lcx_wanted(const awry_index* ix) {
  static const bool off = getenv("AWRY_LCX") && !strcmp(getenv("AWRY_LCX"), "0");
  return !off && ix->lcx_request != 0;
}

// The left-context index of a nucleotide replica (layout.h, lcx.hip.h): 16 B per row for the keys and the (position, row)
// pairs, 0.6 B for the sampled levels.  Built from what is resident anyway -- the final seed table (its 2+ row entries name
// the buckets), the ratio-1 dense SA, the 4-bit text -- in chunks of rows cut at bucket boundaries: per chunk two stable radix
// sorts (context key, then bucket) order the covered rows, which then go back to their buckets' own row slots.
// Skipped (returns false) when the HBM that is free does not hold it and its build scratch with room to spare.
bool build_lcx(awry_index* ix, Replica& r) {
  drop_lcx(r);
  const HostIndex& h = ix->host;
  if (r.wide || h.alphabet != NUCLEOTIDE || !r.seed.p || r.seed_k < 8 || !r.dev.text4 || !r.dense_sa.p || r.dense_ratio != 1) return false;
  const uint64_t N = h.bwt_len, nfinal = 1ull << (2 * r.seed_k);
  uint64_t lev_n[8] = {0}, lev_off[8] = {0}, inner_total = 16;
  for (int t = 1; t <= 7; t++) {
    lev_n[t] = ((((N - 1) >> (4 * t)) + 1 + 15) / 16) * 16 + 16;  // whole nodes, one to spare
    lev_off[t] = inner_total;
    inner_total += lev_n[t];
  }
  const double resident = 16.0 * (double)(N + 32) + 8.0 * (double)inner_total;
  size_t free_b = 0;
  if (!hbm_budget(&free_b)) return false;
  constexpr double PER_ROW = 96.0;  // build scratch per row of a chunk (row info 17 B, covered rows 4 B, two double-buffered pair sorts 48 B, rocPRIM's own)
  if (resident + PER_ROW * (double)(1u << 24) > 0.75 * (double)free_b) return false;
  const uint64_t chunk = (uint64_t)std::max(1.0 * (1u << 24), std::min(1.0 * (1u << 28), (0.75 * (double)free_b - resident) / PER_ROW));
  const uint32_t max_bucket = (uint32_t)std::min<uint64_t>(1u << 24, chunk / 2);
  static const bool verbose = getenv("AWRY_VERBOSE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  DevBuf<uint64_t> key(N + 32), rowpos(N + 32), inner(inner_total);
  hipStream_t s = r.stream;
  HIP_CHECK(hipMemsetAsync(key.p, 0, (N + 32) * 8, s));
  HIP_CHECK(hipMemsetAsync(rowpos.p, 0, (N + 32) * 8, s));
  hipLaunchKernelGGL(lcx_flag_big_kernel, dim3(grid_for(r, nfinal, 256)), dim3(256), 0, s, r.seed.p, nfinal, max_bucket);
  HIP_CHECK(hipGetLastError());
  {
    const uint64_t cap = std::min(chunk + max_bucket, N);
    DevBuf<uint64_t> bkey(cap), ckey(cap), k1a(cap), k1b(cap), b1a(cap), b1b(cap);
    DevBuf<uint8_t> valid(cap);
    DevBuf<uint32_t> slot(cap + 1), p1a(cap), p1b(cap), q2a(cap), q2b(cap), small(4);
    size_t tmp_bytes = 0, need = 0;
    {  // rocPRIM scratch: the largest of the three calls at full capacity
      rocprim::double_buffer<uint64_t> dk(k1a.p, k1b.p);
      rocprim::double_buffer<uint32_t> dv(p1a.p, p1b.p);
      HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, dk, dv, (size_t)cap, 0, 64, s));
      tmp_bytes = need;
      HIP_CHECK(rocprim::select(nullptr, need, rocprim::counting_iterator<uint32_t>(0), valid.p, slot.p, small.p, (size_t)cap, s));
      tmp_bytes = std::max(tmp_bytes, need);
    }
    DevBuf<uint8_t> tmp(tmp_bytes + 256);
    uint32_t h_small[4];
    uint64_t r0 = 0, covered = 0, nchunks = 0;
    while (r0 < N) {
      uint64_t r1 = std::min(N, r0 + chunk);
      if (r1 < N) {  // cut at the first row of the bucket that holds row r1
        hipLaunchKernelGGL(lcx_bucket_start_kernel, dim3(1), dim3(64), 0, s, r.dev, (uint32_t)r1, max_bucket, small.p + 1);
        HIP_CHECK(hipMemcpyAsync(h_small, small.p + 1, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        if (h_small[0] > r0 && h_small[0] <= r1) r1 = h_small[0];
      }
      const uint64_t n = r1 - r0;
      const dim3 g(grid_for(r, n, 256)), b(256);
      hipLaunchKernelGGL(lcx_rowinfo_kernel, g, b, 0, s, r.dev, (uint32_t)r0, (uint32_t)n, max_bucket, bkey.p, ckey.p, valid.p);
      need = tmp_bytes;
      HIP_CHECK(rocprim::select(tmp.p, need, rocprim::counting_iterator<uint32_t>(0), valid.p, slot.p, small.p, (size_t)n, s));
      HIP_CHECK(hipMemcpyAsync(h_small, small.p, 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      const uint64_t nv = h_small[0];
      if (nv) {
        const dim3 gv(grid_for(r, nv, 256));
        hipLaunchKernelGGL(lcx_gather_u64_kernel, gv, b, 0, s, ckey.p, slot.p, nv, k1a.p);
        hipLaunchKernelGGL(lcx_iota_kernel, gv, b, 0, s, p1a.p, nv);
        rocprim::double_buffer<uint64_t> dk(k1a.p, k1b.p);
        rocprim::double_buffer<uint32_t> dv(p1a.p, p1b.p);
        need = tmp_bytes;
        HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, need, dk, dv, (size_t)nv, 0, 64, s));
        // bucket keys in the order of the first sort, then the (stable) sort by bucket
        hipLaunchKernelGGL(lcx_gather2_u64_kernel, gv, b, 0, s, bkey.p, dv.current(), slot.p, nv, b1a.p);
        hipLaunchKernelGGL(lcx_iota_kernel, gv, b, 0, s, q2a.p, nv);
        rocprim::double_buffer<uint64_t> db(b1a.p, b1b.p);
        rocprim::double_buffer<uint32_t> dq(q2a.p, q2b.p);
        need = tmp_bytes;
        HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, need, db, dq, (size_t)nv, 0, 2 * r.seed_k + 1, s));
        hipLaunchKernelGGL(lcx_place_kernel, gv, b, 0, s, r.dev, (uint32_t)r0, nv, dk.current(), dv.current(), dq.current(), slot.p, key.p, rowpos.p);
        hipLaunchKernelGGL(lcx_tail_kernel, gv, b, 0, s, r.dev, (uint32_t)r0, nv, db.current(), slot.p, key.p, r.seed.p);
        HIP_CHECK(hipGetLastError());
      }
      covered += nv;
      nchunks++;
      r0 = r1;
    }
    HIP_CHECK(hipStreamSynchronize(s));
    if (verbose) fprintf(stderr, "[awry replica %d] left-context index: %llu of %llu rows in buckets of 2..%u rows, %llu chunks of <= %llu rows\n", r.device,
                         (unsigned long long)covered, (unsigned long long)N, max_bucket, (unsigned long long)nchunks, (unsigned long long)chunk);
  }
  for (int t = 1; t <= 7; t++)
    hipLaunchKernelGGL(lcx_sample_kernel, dim3(grid_for(r, lev_n[t], 256)), dim3(256), 0, s, key.p, N, t, inner.p + lev_off[t], lev_n[t]);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(s));
  r.lcx_key = std::move(key);
  r.lcx_rowpos = std::move(rowpos);
  r.lcx_inner = std::move(inner);
  r.dev.lcx_key = r.lcx_key.p;
  r.dev.lcx_rowpos = r.lcx_rowpos.p;
  r.dev.lcx_inner = r.lcx_inner.p;
  for (int t = 0; t < 8; t++) r.dev.lcx_off[t] = (uint32_t)lev_off[t];
  if (verbose) fprintf(stderr, "[awry replica %d] left-context index built in %.2f s (%.1f GB)\n", r.device,
                       std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(), resident / 1e9);
  return true;
}

// Position seeds are kept exactly while they pay: nucleotide replica with the verify accelerators resident and a table
// sparse enough for the two-phase schedules (the kernels of those schedules settle a singleton from the text and never
// need its row; the other schedules and the generic kernel would have to start such queries over without the table).
// Called after anything that changes the table or the accelerators.  AWRY_SEED_POS=0 keeps rows.
void sync_seed_mode(awry_index* ix, Replica& r) {
  const HostIndex& h = ix->host;
  static const bool off = getenv("AWRY_SEED_POS") && !strcmp(getenv("AWRY_SEED_POS"), "0");
  if (r.wide) return;  // wide rows: no position seeds (32-bit structures)
  const bool nt = h.alphabet == NUCLEOTIDE;
  // nucleotide: text4 resident and the two-phase schedules are the policy; amino: text8 resident (its only consumer, the
  // generic kernel, then finishes singletons against the text)
  const bool want = !off && narrow(h) && r.seed_k > 0 && r.seed.p && r.dense_ratio == 1 && r.dense_sa.p &&
                    (nt ? (r.dev.text4 && (1ull << (2 * r.seed_k)) / 3 >= h.bwt_len) : r.dev.text8 != nullptr);
  const bool want_lcx = want && nt && lcx_wanted(ix);
  if (want == (r.dev.seed_pos != 0)) {
    if (want && want_lcx != (r.dev.lcx_key != nullptr)) {
      if (want_lcx) build_lcx(ix, r);
      else { build_seed(ix, r, r.seed_k); sync_seed_mode(ix, r); }  // (the table carries the index's flags: a fresh one, then position seeds again)
    }
    return;
  }
  if (!want) {  // rows again: rebuild (the row of a position is not recoverable without an inverse SA)
    build_seed(ix, r, r.seed_k);
    return;
  }
  uint64_t nfinal = 1;
  for (int j = 0; j < r.seed_k; j++) nfinal *= nt ? 4 : AA_SEED_SIGMA;
  // context letters beyond the 14 of the count field ride in the top bits of sp that positions of this text never use
  const int extra = nt ? (int)std::min<uint64_t>(15, (32 - std::min<uint64_t>(32, h.sa_bits)) / 2) : 0;
  hipLaunchKernelGGL(seed_rows_to_positions_kernel, dim3(grid_for(r, nfinal, 256)), dim3(256), 0, r.stream, r.seed.p, nfinal, r.dense_sa.p,
                     nt ? SEED_CNT_SAT : AA_SEED_CNT_SAT, nt ? r.dev.text4 : nullptr, extra, nt ? nullptr : r.dev.text8);
  r.dev.ctx_extra = (uint32_t)extra;
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(r.stream));
  r.dev.seed_pos = 1;
  if (want_lcx) build_lcx(ix, r);
}

std::unique_ptr<Replica> make_replica(awry_index* ix, int device) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) throw NoDeviceError("no HIP device available (there is no CPU search path)");
  require(device >= 0 && device < ndev, "device id out of range");
  HIP_CHECK(hipSetDevice(device));
  // replicas of one GPU are built one after the other (each sizes its seed table and accelerators from the HBM that is
  // free when its turn comes); replicas of different GPUs build concurrently
  static std::mutex build_mu[64];
  std::lock_guard<std::mutex> build_lock(build_mu[device & 63]);
  auto r = std::make_unique<Replica>();
  r->device = device;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  r->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_CHECK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
  for (auto& ls : r->lane_stream) HIP_CHECK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&r->copy_in, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&r->copy_out, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreate(&r->ev0));
  HIP_CHECK(hipEventCreate(&r->ev1));
  const HostIndex& h = ix->host;
  static const bool verbose = getenv("AWRY_VERBOSE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {  // AWRY_VERBOSE: where the seconds of a replica's construction go
    if (!verbose) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[awry replica %d] %s: %.2f s\n", device, what, std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  r->blocks.alloc(h.blocks.size());
  r->sa_words.alloc(h.sa_words.size() + 1);  // +1: the straddle read of the last sample never leaves the buffer
  r->seq_starts.alloc(std::max<size_t>(1, h.seq_starts.size()));
  HIP_CHECK(hipMemcpy(r->blocks.p, h.blocks.data(), h.blocks.size() * 8, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemset(r->sa_words.p, 0, (h.sa_words.size() + 1) * 8));
  if (!h.sa_words.empty()) HIP_CHECK(hipMemcpy(r->sa_words.p, h.sa_words.data(), h.sa_words.size() * 8, hipMemcpyHostToDevice));
  if (!h.seq_starts.empty())
    HIP_CHECK(hipMemcpy(r->seq_starts.p, h.seq_starts.data(), h.seq_starts.size() * 8, hipMemcpyHostToDevice));
  r->wide = !narrow(h);
  DevIndex& d = r->dev;
  d.seed64 = nullptr;
  d.blocks = r->blocks.p;
  d.sa_words = r->sa_words.p;
  d.seed = nullptr;
  d.seq_starts = r->seq_starts.p;
  d.seq_bucket = nullptr;
  d.seq_bucket_shift = d.seq_bucket_pad = 0;
  if (h.seq_starts.size() > (size_t)LOC_SEQ_LDS && h.seq_starts.size() < (1ull << 32)) {
    // about four buckets per record (at most 2^22): a position's record is then one of the one or two its bucket touches
    const uint64_t nseq = h.seq_starts.size();
    uint64_t want = 1;
    while (want < 4 * nseq && want < (1ull << 22)) want <<= 1;
    uint32_t shift = 0;
    while ((((h.bwt_len - 1) >> shift) + 1) > want) shift++;
    const uint64_t nb = ((h.bwt_len - 1) >> shift) + 1;
    std::vector<uint32_t> tab(nb + 1);
    uint64_t rec = 0;
    for (uint64_t b = 0; b < nb; b++) {
      const uint64_t p0 = b << shift;
      while (rec + 1 < nseq && h.seq_starts[rec + 1] <= p0) rec++;
      tab[b] = (uint32_t)rec;
    }
    tab[nb] = (uint32_t)(nseq - 1);
    r->seq_bucket.alloc(nb + 1);
    HIP_CHECK(hipMemcpy(r->seq_bucket.p, tab.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
    d.seq_bucket = r->seq_bucket.p;
    d.seq_bucket_shift = shift;
  }
  d.nblocks = h.nblocks;
  d.bwt_len = h.bwt_len;
  d.sentinel_row = h.sentinel_row;
  d.nseq = h.seq_starts.size();
  for (int i = 0; i < 24; i++) d.prefix_sums[i] = i < (int)h.prefix_sums.size() ? h.prefix_sums[i] : 0;
  d.sa_bits = (uint32_t)h.sa_bits;
  d.sa_ratio = (uint32_t)h.sa_ratio;
  d.alphabet = h.alphabet;
  d.seed_k = 0;
  d.dense_sa = nullptr;
  d.text4 = nullptr;
  d.text8 = nullptr;
  d.dense_ratio = 0;
  d.verify_after = 0;
  d.sa_nblock = nullptr;
  d.seed_pos = 0;
  d.ctx_extra = 0;
  lap("index upload");
  build_seed(ix, *r, ix->seed_k_request < 0 ? default_seed_k(h) : ix->seed_k_request);
  lap("seed table");
  build_dense_sa(ix, *r, ix->dense_ratio_request);
  lap("dense SA");
  int vreq = ix->verify_request;
  if (vreq == -2) {  // policy: keep the accelerators (dense SA 4 B + text 1.5 B / 1 B per symbol) resident when they fit comfortably
    vreq = -1;
    const char* e = getenv("AWRY_VERIFY");
    size_t free_b = 0;
    if (!(e && !strcmp(e, "0")) && !r->wide && narrow(h) && hbm_budget(&free_b) && (double)h.bwt_len * 7.0 < 0.5 * (double)free_b)
      vreq = e && atoi(e) > 0 ? atoi(e) : 2;
  }
  if (vreq >= 0) build_verify(ix, *r, vreq);
  lap("verify accelerators (dense SA at ratio 1, text)");
  r->verify_kmers = ix->verify_kmers_request;
  refresh_nblock(ix, *r);
  sync_seed_mode(ix, *r);
  lap("block-of-sample table, seed mode");
  return r;
}

// ---- kernel launch helpers (all asynchronous on `s`) ------------------------------------------------

// ASCII -> packed 2-bit words.  d_off == nullptr: n queries of L bytes each; else query q = bytes [d_off[q] - base, d_off[q+1] - base)
// of d_ascii (total_bytes in all), W words per query (stride), lengths to d_lens.
void launch_pack_nt2(Replica& r, const uint8_t* d_ascii, const uint64_t* d_off, uint64_t base, uint64_t n, uint64_t total_bytes, int L, int W,
                     uint64_t* d_words, uint32_t* d_lens, unsigned long long* d_bad, hipStream_t s, uint32_t* d_bad_list = nullptr) {
  if (n == 0) return;
  const dim3 g(grid_for(r, (n + 63) / 64 * 64, 256)), b(256);
  if (d_off) hipLaunchKernelGGL(pack_nt2_tile_kernel<true>, g, b, 0, s, d_ascii, d_off, base, n, total_bytes, L, W, d_words, d_lens, d_bad, d_bad_list);
  else hipLaunchKernelGGL(pack_nt2_tile_kernel<false>, g, b, 0, s, d_ascii, d_off, base, n, total_bytes, L, W, d_words, d_lens, d_bad, d_bad_list);
  HIP_CHECK(hipGetLastError());
}

Replica::SurvScratch* surv_scratch(Replica& r, hipStream_t s);

// The amino k-mer schedule: count_aa_kmer_probe_kernel (one query per lane: the seed entry, or the entry plus a text
// window, decides most) and the generic kernel over what it listed, as one pool.  d_off == nullptr: n queries of L residues
// back to back; else query q = d_q[d_off[q], d_off[q + 1]) of any length (k .. 24 residues take the first pass, the rest
// is listed).  d_ranges (optional): RS_* words / row starts for the locate pass, in the generic kernel's layout.
void launch_aa_two_phase(Replica& r, const uint8_t* d_q, const uint64_t* d_off, uint64_t n, int L, uint64_t* d_counts, uint64_t* d_ranges,
                         uint8_t* d_status, hipStream_t s, unsigned long long* d_tally) {
  Replica::SurvScratch* sc = surv_scratch(r, s);
  const unsigned nblk = (unsigned)r.num_cus * 8;
  const uint64_t per_block = ((n + (uint64_t)nblk * 256 - 1) / ((uint64_t)nblk * 256)) * 256;  // queries a block sees
  if (sc->cap_q < per_block * nblk) {
    HIP_CHECK(hipStreamSynchronize(s));
    sc->q.alloc(per_block * nblk);
    sc->cap_q = per_block * nblk;
    sc->cap = 0;  // the nucleotide k-mer path re-allocates its three lists together
  }
  if (!sc->count.p) sc->count.alloc(nblk);
  // the second pass works through all lists as one pool on a grid sized to what is resident at once (3 blocks per CU)
  const bool pooled = nblk <= (unsigned)LIST_MAX_LISTS && !getenv("AWRY_AA_LIST_PER_BLOCK");
  const QueryList ql{sc->q.p, sc->count.p, per_block, nullptr, nullptr, 0, d_tally, pooled ? nblk : 0u};
  const unsigned nblk2 = pooled ? (unsigned)r.num_cus * 3 : nblk;
  // Two queries in flight per lane (one: the same rate; four: 141 VGPRs, 10 % slower).  The second pass is a latency
  // chain over a few per cent of the batch; running it for the first half of a batch on a side stream beside the first
  // pass of the second half (event fork / join) was measured and costs more than it hides (12.7 -> 10.7 G present
  // 12-mers/s, host path 0.83 -> 0.52 G queries/s).
  // queries of more than 24 residues (up to AA_KMER_LONG_MAX): the LONG instantiations -- the same pass over a query's last 24
  // residues plus a comparison of the rest with the text for the candidates that are left
  static const bool no_long = getenv("AWRY_AA_LONG") && !strcmp(getenv("AWRY_AA_LONG"), "0");
  if (d_off && !no_long) hipLaunchKernelGGL((count_aa_kmer_probe_kernel<2, true, true>), dim3(nblk), dim3(256), 0, s, r.dev, d_q, d_off, n, 0, d_counts, d_ranges, d_status, ql);
  else if (d_off) hipLaunchKernelGGL((count_aa_kmer_probe_kernel<2, true>), dim3(nblk), dim3(256), 0, s, r.dev, d_q, d_off, n, 0, d_counts, d_ranges, d_status, ql);
  else if (L > AA_KMER_MAX) hipLaunchKernelGGL((count_aa_kmer_probe_kernel<2, false, true>), dim3(nblk), dim3(256), 0, s, r.dev, d_q, d_off, n, L, d_counts, d_ranges, d_status, ql);
  else hipLaunchKernelGGL((count_aa_kmer_probe_kernel<2, false>), dim3(nblk), dim3(256), 0, s, r.dev, d_q, d_off, n, L, d_counts, d_ranges, d_status, ql);
  hipLaunchKernelGGL((count_scalar_kernel<AMINO, LIST_BLOCK>), dim3(nblk2), dim3(256), 0, s, r.dev, d_q, d_off, n, d_counts, d_ranges, d_status, 1,
                     d_off ? 0 : (uint64_t)L, ql);
  HIP_CHECK(hipGetLastError());
}

// allow_verify: the generic kernel may finish queries against the text (ranges then hold RS_* words for locate, not rows)
// ulen != 0: n queries of ulen bytes each, back to back (d_off is not read)
// ref_kmer_len >= 0: the reference's own step schedule with that lookup_table_kmer_len -- no seed table, kmer_len - 1 steps taken
// unconditionally (src/fm_index.rs:402-438, src/kmer_lookup_table.rs:90-110): what awry_search_range returns, rows of absent queries included
void launch_count_ascii(Replica& r, const uint8_t* d_q, const uint64_t* d_off, uint64_t n, uint64_t* d_counts,
                        uint64_t* d_ranges, uint8_t* d_status, hipStream_t s, bool allow_verify, uint64_t ulen = 0, int ref_kmer_len = -1) {
  if (n == 0) return;
  const int vmode = ref_kmer_len >= 0 ? (2 | (ref_kmer_len << 8)) : (allow_verify ? 1 : 0);
  if (ref_kmer_len >= 0) allow_verify = false;
  static const bool aa_off = getenv("AWRY_AA_KMER") && !strcmp(getenv("AWRY_AA_KMER"), "0");
  if (r.dev.alphabet == AMINO && allow_verify && !ulen && d_off && r.seed_k >= 1 && n >= 4096 && n < (1ull << 32) && !aa_off) {
    // amino batches of any lengths: the k-mer schedule with per-query lengths (queries it does not take are listed)
    launch_aa_two_phase(r, d_q, d_off, n, 0, d_counts, d_ranges, d_status, s, nullptr);
    return;
  }
  const dim3 g(grid_for(r, n, 256)), b(256);
  const QueryList none{};
  if (r.dev.alphabet == NUCLEOTIDE)
    hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_NONE>), g, b, 0, s, r.dev, d_q, d_off, n, d_counts, d_ranges, d_status, vmode, ulen, none);
  else
    hipLaunchKernelGGL((count_scalar_kernel<AMINO, LIST_NONE>), g, b, 0, s, r.dev, d_q, d_off, n, d_counts, d_ranges, d_status, vmode, ulen, none);
  HIP_CHECK(hipGetLastError());
}

uint64_t scan_tiles(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

void launch_scan(Replica& r, const uint64_t* d_counts, uint64_t n, uint64_t* d_hit_off, uint64_t* d_scratch, hipStream_t s) {
  if (n == 0) { HIP_CHECK(hipMemsetAsync(d_hit_off, 0, 8, s)); return; }
  const uint64_t tiles = scan_tiles(n);
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(256), 0, s, d_counts, n, d_scratch);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(256), 0, s, d_scratch, tiles, d_scratch + tiles);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(256), 0, s, d_counts, n, d_scratch, d_hit_off);
  HIP_CHECK(hipGetLastError());
}

Replica::SurvScratch* surv_scratch(Replica& r, hipStream_t s);

unsigned long long* next_counter(Replica& r, hipStream_t s) {
  Replica::SurvScratch* sc = surv_scratch(r, s);
  unsigned long long* ctr;
  {
    std::lock_guard<std::mutex> lock(r.scratch_mu);
    if (!sc->counters.p) sc->counters.alloc(8);
    ctr = sc->counters.p + (sc->counter_seq++ & 7u);
  }
  HIP_CHECK(hipMemsetAsync(ctr, 0, 8, s));
  return ctr;
}

// d_range_start[q * rs_stride] = first BWT row of query q's range
void launch_locate(Replica& r, const uint64_t* d_range_start, int rs_stride, const uint64_t* d_hit_off, uint64_t n, uint64_t total,
                   uint64_t* d_gpos, uint64_t* d_pos, hipStream_t s, unsigned long long* d_tally = nullptr) {
  if (total == 0) return;
  static const bool scalar = getenv("AWRY_LOCATE_KERNEL") && !strcmp(getenv("AWRY_LOCATE_KERNEL"), "scalar");
  if (scalar && rs_stride == 2) {  // round-1 baseline kernel: one hit per lane, global binary search, file samples only
    const dim3 g(grid_for(r, total, 256)), b(256);
    if (r.dev.alphabet == NUCLEOTIDE)
      hipLaunchKernelGGL(locate_scalar_kernel<NUCLEOTIDE>, g, b, 0, s, r.dev, d_range_start, d_hit_off, n, total, d_gpos, d_pos);
    else
      hipLaunchKernelGGL(locate_scalar_kernel<AMINO>, g, b, 0, s, r.dev, d_range_start, d_hit_off, n, total, d_gpos, d_pos);
    HIP_CHECK(hipGetLastError());
    return;
  }
  unsigned long long* ctr = next_counter(r, s);
  const uint64_t tiles = (total + LOC_TILE - 1) / LOC_TILE;
  const dim3 g((unsigned)std::min<uint64_t>(tiles, (uint64_t)r.num_cus * 8)), b(256);
  // consecutive tiles a block draws at a time (one search of the whole offset array per run): long enough to amortise
  // that search, short enough that every block still draws several runs and the launch ends evenly
  const uint32_t run_len = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, tiles / ((uint64_t)g.x * 4)));
  const uint32_t* dense = r.dense_ratio ? r.dense_sa.p : nullptr;
  if (r.dev.alphabet == NUCLEOTIDE)
    hipLaunchKernelGGL(locate_tile_kernel<NUCLEOTIDE>, g, b, 0, s, r.dev, d_range_start, rs_stride, d_hit_off, n, total, dense, r.dense_ratio,
                       d_gpos, d_pos, ctr, run_len);
  else
    hipLaunchKernelGGL(locate_tile_kernel<AMINO>, g, b, 0, s, r.dev, d_range_start, rs_stride, d_hit_off, n, total, dense, r.dense_ratio,
                       d_gpos, d_pos, ctr, run_len);
  HIP_CHECK(hipGetLastError());
  if (r.dense_ratio == 1) return;  // every row is a sampled row: nothing was deferred
  // the hits whose row is not sampled walk in a second pass that is not tied to tiles (locate_walk_kernel)
  unsigned long long* wctr = next_counter(r, s);
  const dim3 gw(grid_for(r, total, 256, 7));
  static const bool generic_walk = getenv("AWRY_LOCATE_WALK") && !strcmp(getenv("AWRY_LOCATE_WALK"), "generic");
  static const bool direct_walk = getenv("AWRY_LOCATE_WALK") && !strcmp(getenv("AWRY_LOCATE_WALK"), "direct");
  if (r.dev.alphabet == NUCLEOTIDE && !generic_walk && !direct_walk && d_tally)
    hipLaunchKernelGGL((locate_walk_nt_lane_kernel<true, true>), dim3(grid_for(r, total, 256, 4)), b, 0, s, r.dev, total, dense, r.dense_ratio, d_gpos, wctr, d_tally);
  else if (r.dev.alphabet == NUCLEOTIDE && !generic_walk && !direct_walk)
    hipLaunchKernelGGL((locate_walk_nt_lane_kernel<true, false>), dim3(grid_for(r, total, 256, 4)), b, 0, s, r.dev, total, dense, r.dense_ratio, d_gpos, wctr, nullptr);
  else if (r.dev.alphabet == NUCLEOTIDE && !generic_walk)
    hipLaunchKernelGGL((locate_walk_nt_lane_kernel<false, false>), dim3(grid_for(r, total, 256, 8)), b, 0, s, r.dev, total, dense, r.dense_ratio, d_gpos, wctr, nullptr);
  else if (r.dev.alphabet == NUCLEOTIDE) hipLaunchKernelGGL(locate_walk_kernel<NUCLEOTIDE>, gw, b, 0, s, r.dev, total, dense, r.dense_ratio, d_gpos, d_pos, wctr);
  else hipLaunchKernelGGL(locate_walk_kernel<AMINO>, gw, b, 0, s, r.dev, total, dense, r.dense_ratio, d_gpos, d_pos, wctr);
  if (d_pos) hipLaunchKernelGGL(localise_walked_kernel, dim3(grid_for(r, total, 256)), b, 0, s, r.dev, total, d_gpos, d_pos);
  HIP_CHECK(hipGetLastError());
}

// dense device SA for locate: ratio 0 = off (walk to the file's samples), r >= 1 = keep SA[j r] for every j as u32
void build_dense_sa(awry_index* ix, Replica& r, int ratio) {
  r.text4.reset();  // the verify shortcut rides on the ratio-1 dense SA; it is re-enabled by build_verify()
  r.text8.reset();
  r.dev.text4 = nullptr;
  r.dev.text8 = nullptr;
  r.dense_sa.reset();
  r.dense_ratio = 0;
  r.dev.dense_sa = nullptr;
  r.dev.dense_ratio = 0;
  if (ratio <= 0) return;
  require(!r.wide && ix->host.bwt_len < (1ull << 32), "a dense device SA needs an index with 32-bit rows (bwt_len < 2^32)");
  const uint64_t nentries = (ix->host.bwt_len + ratio - 1) / ratio;
  const uint64_t nsamples = (ix->host.bwt_len + ix->host.sa_ratio - 1) / ix->host.sa_ratio;  // one chain per file sample
  DevBuf<uint32_t> d(nentries);
  const dim3 g(grid_for(r, nsamples, 256, 64)), b(256);
  if (r.dev.alphabet == NUCLEOTIDE) hipLaunchKernelGGL(densify_sa_kernel<NUCLEOTIDE>, g, b, 0, r.stream, r.dev, (uint32_t)ratio, nsamples, d.p);
  else hipLaunchKernelGGL(densify_sa_kernel<AMINO>, g, b, 0, r.stream, r.dev, (uint32_t)ratio, nsamples, d.p);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(r.stream));
  r.dense_sa = std::move(d);
  r.dense_ratio = (uint32_t)ratio;
  r.dev.dense_sa = r.dense_sa.p;
  r.dev.dense_ratio = r.dense_ratio;
}

// While locate has to walk (no ratio-1 dense SA), a nucleotide replica keeps the SA values of the BWT's N block: a walk
// that runs into an N run stops there instead of following the run (see DevIndex::sa_nblock).  4 B per N of the text.
void refresh_nblock(awry_index* ix, Replica& r) {
  const HostIndex& h = ix->host;
  const uint64_t lo = h.alphabet == NUCLEOTIDE ? h.prefix_sums[4] : 0, hi = h.alphabet == NUCLEOTIDE ? h.prefix_sums[5] : 0;
  const bool want = h.alphabet == NUCLEOTIDE && !r.wide && narrow(h) && r.dense_ratio != 1 && hi - lo >= 64;
  if (!want) {
    r.sa_nblock.reset();
    r.dev.sa_nblock = nullptr;
    return;
  }
  if (r.sa_nblock.p) return;  // depends on the index only
  DevBuf<uint32_t> d(hi - lo);
  const uint64_t nsamples = (h.bwt_len + h.sa_ratio - 1) / h.sa_ratio;
  hipLaunchKernelGGL(nblock_sa_kernel<NUCLEOTIDE>, dim3(grid_for(r, nsamples, 256, 64)), dim3(256), 0, r.stream, r.dev, nsamples,
                     (uint32_t)lo, (uint32_t)hi, d.p);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(r.stream));
  r.sa_nblock = std::move(d);
  r.dev.sa_nblock = r.sa_nblock.p;
}

// seed-and-verify: needs the ratio-1 dense SA and the text, both recovered from the index on the device -- as 4-bit
// codes for the packed nucleotide kernels (text4) and as one symbol index per byte for the generic kernel (text8, any
// alphabet).  after_steps < 0 switches it off.
void build_verify(awry_index* ix, Replica& r, int after_steps) {
  r.text4.reset();
  r.text8.reset();
  r.dev.text4 = nullptr;
  r.dev.text8 = nullptr;
  r.dev.verify_after = 0;
  if (after_steps < 0) return;
  require(!r.wide && narrow(ix->host), "seed-and-verify needs an index with bwt_len < 2^32");
  if (r.dense_ratio != 1) build_dense_sa(ix, r, 1);
  if (r.dense_ratio != 1 || !r.dense_sa.p) throw HipError("seed-and-verify: the ratio-1 dense SA is missing");
  const bool nt = ix->host.alphabet == NUCLEOTIDE;
  const dim3 g(grid_for(r, ix->host.bwt_len, 256)), b(256);
  DevBuf<uint8_t> t8(ix->host.bwt_len + 16);
  if (nt) hipLaunchKernelGGL(text8_scatter_kernel<NUCLEOTIDE>, g, b, 0, r.stream, r.dev, t8.p);
  else hipLaunchKernelGGL(text8_scatter_kernel<AMINO>, g, b, 0, r.stream, r.dev, t8.p);
  HIP_CHECK(hipGetLastError());
  if (nt) {
    const uint64_t nwords = (ix->host.bwt_len + 7) / 8 + 8;  // + slack: a 32-symbol window read touches 5 words
    DevBuf<uint32_t> t(nwords);
    HIP_CHECK(hipMemsetAsync(t.p, 0, nwords * 4, r.stream));
    hipLaunchKernelGGL(text4_scatter_kernel<NUCLEOTIDE>, g, b, 0, r.stream, r.dev, t.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(r.stream));
    r.text4 = std::move(t);
    r.dev.text4 = r.text4.p;
  }
  HIP_CHECK(hipStreamSynchronize(r.stream));
  r.text8 = std::move(t8);
  r.dev.text8 = r.text8.p;
  r.dev.verify_after = (uint32_t)after_steps;
}

// survivor lists of the two-phase schedules, one set per stream
Replica::SurvScratch* surv_scratch(Replica& r, hipStream_t s) {
  std::lock_guard<std::mutex> lock(r.scratch_mu);
  auto& slot = r.scratch[s];
  if (!slot) slot = std::make_unique<Replica::SurvScratch>();
  return slot.get();
}

// reads: phase 2 as a pooled search pass (lcx_quad_reads_kernel) + LF pass whenever the left-context index is resident;
// AWRY_LCX_POOL=0 keeps count_nt2_reads_kernel<.., LIST> (block b works through block b's list) for A/B
bool lcx_lanes(const Replica& r) {
  static const bool off = getenv("AWRY_LCX_POOL") && !strcmp(getenv("AWRY_LCX_POOL"), "0");
  return !off && r.dev.lcx_key != nullptr && r.dev.text4 != nullptr && r.dev.dense_ratio == 1;
}
// wide-row replicas: the two-phase schedule (count_nt2_wide_probe_kernel + the listed quad pass) is an alternative, not the
// policy -- on a GRCh38-scale index forced onto 64-bit rows (k = 16, 69 GB of 16-byte entries) it runs random 31-mers at 16.7
// against 16.5 G/s and reads from the text 10 % slower than the single strided quad kernel: without the 32-bit accelerators
// (dense SA, text, position seeds) the entry settles too few queries for a second launch to pay.  Selected with
// AWRY_COUNT_KERNEL=twophase / awry_debug_set_count_kernel(3).
bool wide_two_phase(uint64_t n) { return count_kernel_override() == 3 && n < (1ull << 32); }
// the survivor lists of a two-phase launch over n queries: `in` (all three arrays) and, with lanes, the fallback lists `out`
void two_phase_lists(Replica& r, hipStream_t s, uint64_t n, bool lanes, Nt2Survivors* in, Nt2Survivors* out, unsigned* nblk_out) {
  Replica::SurvScratch* sc = surv_scratch(r, s);
  const unsigned nblk = (unsigned)r.num_cus * 8;  // lists = blocks of the probe pass
  const uint64_t per_block = ((n + (uint64_t)nblk * 256 - 1) / ((uint64_t)nblk * 256)) * 256, total = per_block * nblk;
  if (sc->cap < total || (lanes && sc->fcap < total)) {
    HIP_CHECK(hipStreamSynchronize(s));
    if (sc->cap < total) {
      sc->w.alloc(total); sc->range.alloc(total); sc->q.alloc(total);
      sc->cap = sc->cap_q = total;
    }
    if (lanes && sc->fcap < total) {  // (+ the chunks the waves of lcx_quad_kernel reserve and do not fill)
      const uint64_t slack = 1ull << 21;
      sc->fw.alloc(total + slack); sc->frange.alloc(total + slack); sc->fq.alloc(total + slack);
      sc->fcap = total;
    }
  }
  if (!sc->count.p) sc->count.alloc(nblk);
  if (lanes && !sc->fcount.p) sc->fcount.alloc(8);  // [0] length of the LF list
  *in = Nt2Survivors{sc->w.p, sc->range.p, sc->q.p, sc->count.p, per_block};
  *out = Nt2Survivors{};
  if (lanes) {
    *out = Nt2Survivors{sc->fw.p, sc->frange.p, sc->fq.p, sc->fcount.p, total};
    in->lf_count = sc->fcount.p;
  }
  *nblk_out = nblk;
}
// blocks of lcx_quad_kernel that are resident at once: its grid (the pool of survivors is shared out dynamically)
template <class K>
unsigned resident_grid(const Replica& r, K kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
  return (unsigned)r.num_cus * (unsigned)per_cu;
}

// d_lens != nullptr: read q has d_lens[q] letters (1..L) in its W = ceil(L / 32) words; else every read has L letters
void launch_count_nt2_long(Replica& r, const uint64_t* d_words, uint64_t n, int L, uint64_t* d_counts, uint64_t* d_range_start,
                           bool use_seed, hipStream_t s, const uint32_t* d_lens = nullptr) {
  require(r.dev.alphabet == NUCLEOTIDE, "packed 2-bit queries need a nucleotide index");
  require(L >= 1 && L <= 1 << 20, "packed read length out of range");
  if (n == 0) return;
  if (r.wide) {  // 64-bit rows
    const bool sdw = use_seed && r.seed_k > 0 && r.dev.seed64 && (d_lens || r.seed_k <= L);
    const dim3 gw(grid_for(r, n * 4, 256)), bw(256);
    if (sdw && wide_two_phase(n)) {  // per-lane probe pass, then the quads on what has to be stepped
      Nt2Survivors sv, fb;
      unsigned nblk = 0;
      two_phase_lists(r, s, n, false, &sv, &fb, &nblk);
      if (d_lens) {
        hipLaunchKernelGGL((count_nt2_wide_probe_kernel<true, false>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, sv, d_lens, (unsigned long long*)nullptr);
        hipLaunchKernelGGL((count_nt2_wide_kernel<true, true, true>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, d_lens, (unsigned long long*)nullptr, sv);
      } else {
        hipLaunchKernelGGL((count_nt2_wide_probe_kernel<false, false>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, sv, d_lens, (unsigned long long*)nullptr);
        hipLaunchKernelGGL((count_nt2_wide_kernel<true, false, true>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, d_lens, (unsigned long long*)nullptr, sv);
      }
      HIP_CHECK(hipGetLastError());
      return;
    }
#define AWRY_LAUNCH_WIDE(S, R) hipLaunchKernelGGL((count_nt2_wide_kernel<S, R>), gw, bw, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, d_lens, (unsigned long long*)nullptr)
    if (d_lens) { if (sdw) AWRY_LAUNCH_WIDE(true, true); else AWRY_LAUNCH_WIDE(false, true); }
    else { if (sdw) AWRY_LAUNCH_WIDE(true, false); else AWRY_LAUNCH_WIDE(false, false); }
#undef AWRY_LAUNCH_WIDE
    HIP_CHECK(hipGetLastError());
    return;
  }
  const bool sd = use_seed && r.seed_k > 0 && (d_lens || r.seed_k <= L);  // ragged reads decide per read
  const bool vfy = r.dev.text4 && r.dev.dense_ratio == 1;
  const dim3 g(grid_for(r, n * 4, 256)), b(256);
  const int om = count_kernel_override();
  if (vfy && sd && L - r.seed_k >= 3 && L <= 512 && n < (1ull << 32) && (om < 0 || om == 3)) {
    // two-phase: a per-lane pass settles the reads their seed entry (plus one SA read and one text window) decides,
    // the quad kernel works through the rest
    const bool lanes = lcx_lanes(r);
    Nt2Survivors sv, fb;
    unsigned nblk = 0;  // both phases of the quad schedule use this grid
    two_phase_lists(r, s, n, lanes, &sv, &fb, &nblk);
    if (!lanes) sv.w = sv.range = nullptr;  // (the probe pass then lists the reads only)
    const dim3 gq((unsigned)r.num_cus * 8);  // the quad code over what the lanes left
    if (d_lens) {
      hipLaunchKernelGGL(count_nt2_reads_probe_kernel<true>, dim3(nblk), b, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, sv, d_lens);
      if (lanes) {
        static const unsigned gl = resident_grid(r, lcx_quad_reads_kernel<true>);
        hipLaunchKernelGGL(lcx_quad_reads_kernel<true>, dim3(gl), b, 0, s, r.dev, d_words, L, d_counts, d_range_start, sv, fb, nblk, d_lens);
        hipLaunchKernelGGL(count_nt2_reads_pool_kernel<true>, gq, b, 0, s, r.dev, d_words, L, d_counts, d_range_start, fb, d_lens);
      } else hipLaunchKernelGGL((count_nt2_reads_kernel<true, true, true, true>), dim3(nblk), b, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, sv, d_lens);
    } else {
      hipLaunchKernelGGL(count_nt2_reads_probe_kernel<false>, dim3(nblk), b, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, sv, d_lens);
      if (lanes) {
        static const unsigned gl = resident_grid(r, lcx_quad_reads_kernel<false>);
        hipLaunchKernelGGL(lcx_quad_reads_kernel<false>, dim3(gl), b, 0, s, r.dev, d_words, L, d_counts, d_range_start, sv, fb, nblk, d_lens);
        hipLaunchKernelGGL(count_nt2_reads_pool_kernel<false>, gq, b, 0, s, r.dev, d_words, L, d_counts, d_range_start, fb, d_lens);
      } else hipLaunchKernelGGL((count_nt2_reads_kernel<true, true, true, false>), dim3(nblk), b, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, sv, d_lens);
    }
    HIP_CHECK(hipGetLastError());
    return;
  }
  const Nt2Survivors none{};
#define AWRY_LAUNCH_READS(S, V)                                                                                                    \
  do {                                                                                                                            \
    if (d_lens) hipLaunchKernelGGL((count_nt2_reads_kernel<S, V, false, true>), g, b, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, none, d_lens); \
    else hipLaunchKernelGGL((count_nt2_reads_kernel<S, V, false, false>), g, b, 0, s, r.dev, d_words, n, L, d_counts, d_range_start, none, d_lens);       \
  } while (0)
  if (vfy) { if (sd) AWRY_LAUNCH_READS(true, true); else AWRY_LAUNCH_READS(false, true); }
  else { if (sd) AWRY_LAUNCH_READS(true, false); else AWRY_LAUNCH_READS(false, false); }
#undef AWRY_LAUNCH_READS
  HIP_CHECK(hipGetLastError());
}

void launch_count_nt2(Replica& r, const uint64_t* d_words, uint64_t n, int L, uint64_t* d_counts, bool use_seed, hipStream_t s,
                      unsigned long long* d_tally = nullptr) {
  require(r.dev.alphabet == NUCLEOTIDE, "packed 2-bit queries need a nucleotide index");
  require(L >= 1 && L <= 32, "packed k-mer length must be in 1..32");
  if (n == 0) return;
  if (r.wide) {  // 64-bit rows: one word per k-mer is the W = 1 case of the wide kernel
    const bool sdw = use_seed && r.seed_k > 0 && r.dev.seed64 && r.seed_k <= L;
    const dim3 gw(grid_for(r, n * 4, 256)), bw(256);
    if (sdw && wide_two_phase(n)) {
      Nt2Survivors sv, fb;
      unsigned nblk = 0;
      two_phase_lists(r, s, n, false, &sv, &fb, &nblk);
      if (d_tally) hipLaunchKernelGGL((count_nt2_wide_probe_kernel<false, true>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, (uint64_t*)nullptr, sv, (const uint32_t*)nullptr, d_tally);
      else hipLaunchKernelGGL((count_nt2_wide_probe_kernel<false, false>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, (uint64_t*)nullptr, sv, (const uint32_t*)nullptr, d_tally);
      hipLaunchKernelGGL((count_nt2_wide_kernel<true, false, true>), dim3(nblk), bw, 0, s, r.dev, d_words, n, L, d_counts, (uint64_t*)nullptr, (const uint32_t*)nullptr, d_tally, sv);
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (sdw) hipLaunchKernelGGL((count_nt2_wide_kernel<true, false>), gw, bw, 0, s, r.dev, d_words, n, L, d_counts, (uint64_t*)nullptr, (const uint32_t*)nullptr, d_tally);
    else hipLaunchKernelGGL((count_nt2_wide_kernel<false, false>), gw, bw, 0, s, r.dev, d_words, n, L, d_counts, (uint64_t*)nullptr, (const uint32_t*)nullptr, d_tally);
    HIP_CHECK(hipGetLastError());
    return;
  }
  // k-mers shorter than the seed table's k: their own complete table ("rung", built on first use) -- the entry IS the
  // answer, where LF steps from the last letter cost L dependent block reads (GRCh38 scale: 12-mers 5.6 -> 30+ G/s)
  DevIndex dv = r.dev;
  bool rung = false;
  if (use_seed && r.seed_k > L && r.seed.p && n >= 4096 && n < (1ull << 32) && count_kernel_override() < 0)
    if (const SeedEntry* t = seed_rung(r, L)) { dv.seed = t; dv.seed_k = L; dv.seed_pos = 0; dv.ctx_extra = 0; rung = true; }
  const bool seeded = rung || (use_seed && r.seed_k > 0 && r.seed_k <= L);
  const dim3 g(grid_for(r, n * 4, 256)), b(256);
  // AWRY_COUNT_KERNEL=chunk selects the LDS-staged variant (count_nt2_chunk_kernel).  Measured on MI355X it is
  // equal at seed k=14 and 23% slower at k=16 (GRCh38-scale): the strided kernel's query words already arrive
  // as L2 hits, so staging only removes the partial-line result writes and pays chunk drain + refill for it.
  const int kmode = rung ? 3 : count_kernel_mode(r.dev.bwt_len, r.seed_k, seeded);
  const bool use_chunk = kmode == 1;
  if (use_chunk) {
    unsigned long long* ctr = next_counter(r, s);
    if (d_tally) {
      if (seeded) hipLaunchKernelGGL((count_nt2_chunk_kernel<true, true>), g, b, 0, s, r.dev, d_words, n, L, d_counts, ctr, d_tally);
      else hipLaunchKernelGGL((count_nt2_chunk_kernel<false, true>), g, b, 0, s, r.dev, d_words, n, L, d_counts, ctr, d_tally);
    } else {
      if (seeded) hipLaunchKernelGGL((count_nt2_chunk_kernel<true, false>), g, b, 0, s, r.dev, d_words, n, L, d_counts, ctr, d_tally);
      else hipLaunchKernelGGL((count_nt2_chunk_kernel<false, false>), g, b, 0, s, r.dev, d_words, n, L, d_counts, ctr, d_tally);
    }
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (kmode == 3 && seeded && n < (1ull << 32)) {
    // two-phase: per-lane seed probes decide most queries, the quad machinery resumes the survivors
    Nt2Survivors sv, fb;
    unsigned nblk = 0;  // both phases use this grid
    two_phase_lists(r, s, n, false, &sv, &fb, &nblk);
    const dim3 gp(nblk);
    // survivors of phase 1 use seed-and-verify whenever its accelerators are resident (cheap: random batches barely
    // reach phase 2); the single-kernel schedules use it only on request (awry_set_verify_kmers)
    const bool vfy = r.dev.text4 != nullptr && r.dev.dense_ratio == 1;
#define AWRY_LAUNCH_TWO_PHASE(T, V)                                                                                 \
  do {                                                                                                             \
    hipLaunchKernelGGL((count_nt2_probe_kernel<T, V>), gp, b, 0, s, dv, d_words, n, L, d_counts, sv, d_tally);     \
    hipLaunchKernelGGL((count_nt2_resume_kernel<T, V>), gp, b, 0, s, dv, sv, L, d_counts, d_tally);                \
  } while (0)
    if (d_tally) { if (vfy) AWRY_LAUNCH_TWO_PHASE(true, true); else AWRY_LAUNCH_TWO_PHASE(true, false); }
    else { if (vfy) AWRY_LAUNCH_TWO_PHASE(false, true); else AWRY_LAUNCH_TWO_PHASE(false, false); }
#undef AWRY_LAUNCH_TWO_PHASE
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (kmode == 2 || kmode == 3) {  // groups of 4 consecutive queries per quad: whole-sector result writes
    const dim3 g4(grid_for(r, n, 256));
    const bool verify = r.verify_kmers && r.dev.text4 != nullptr && r.dev.dense_ratio == 1;
#define AWRY_LAUNCH_QUAD4(S, T, V) hipLaunchKernelGGL((count_nt2_quad4_kernel<S, T, V>), g4, b, 0, s, r.dev, d_words, n, L, d_counts, d_tally)
    if (verify) {
      if (d_tally) { if (seeded) AWRY_LAUNCH_QUAD4(true, true, true); else AWRY_LAUNCH_QUAD4(false, true, true); }
      else { if (seeded) AWRY_LAUNCH_QUAD4(true, false, true); else AWRY_LAUNCH_QUAD4(false, false, true); }
    } else {
      if (d_tally) { if (seeded) AWRY_LAUNCH_QUAD4(true, true, false); else AWRY_LAUNCH_QUAD4(false, true, false); }
      else { if (seeded) AWRY_LAUNCH_QUAD4(true, false, false); else AWRY_LAUNCH_QUAD4(false, false, false); }
    }
#undef AWRY_LAUNCH_QUAD4
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (d_tally) {
    if (seeded) hipLaunchKernelGGL((count_nt2_quad_kernel<true, true>), g, b, 0, s, r.dev, d_words, n, L, d_counts, d_tally);
    else hipLaunchKernelGGL((count_nt2_quad_kernel<false, true>), g, b, 0, s, r.dev, d_words, n, L, d_counts, d_tally);
  } else {
    if (seeded) hipLaunchKernelGGL((count_nt2_quad_kernel<true, false>), g, b, 0, s, r.dev, d_words, n, L, d_counts, d_tally);
    else hipLaunchKernelGGL((count_nt2_quad_kernel<false, false>), g, b, 0, s, r.dev, d_words, n, L, d_counts, d_tally);
  }
  HIP_CHECK(hipGetLastError());
}

// ---- host batch drivers ----------------------------------------------------------------------------

struct Shard { uint64_t lo, hi; };

std::vector<Shard> shard_queries(uint64_t n, size_t parts) {  // query i -> replica floor(i * G / n): contiguous
  std::vector<Shard> out(parts);
  for (size_t g = 0; g < parts; g++) out[g] = Shard{n * g / parts, n * (g + 1) / parts};
  return out;
}

// cut [lo, hi) into chunks bounded in queries and bytes
std::vector<Shard> chunk_queries(const uint64_t* qoff, uint64_t lo, uint64_t hi) {
  const uint64_t MAXQ = 1ull << 24, MAXB = 1ull << 29;
  std::vector<Shard> out;
  uint64_t a = lo;
  while (a < hi) {
    uint64_t b = std::min(hi, a + MAXQ);
    while (b > a + 1 && qoff[b] - qoff[a] > MAXB) b = a + (b - a) / 2;
    out.push_back(Shard{a, b});
    a = b;
  }
  return out;
}

struct ChunkBuffers {
  DevBuf<uint8_t> q, status;
  DevBuf<uint64_t> off, counts, ranges;
  std::vector<uint64_t> h_off;
  std::vector<uint8_t> h_status;
};

// upload one chunk and run the generic count kernel; leaves counts / ranges / status on the device
void run_count_chunk(Replica& r, ChunkBuffers& cb, const uint8_t* qbytes, const uint64_t* qoff, Shard c, bool want_ranges,
                     bool allow_verify = true, int ref_kmer_len = -1) {
  const uint64_t n = c.hi - c.lo, base = qoff[c.lo], nbytes = qoff[c.hi] - base;
  cb.h_off.resize(n + 1);
  for (uint64_t i = 0; i <= n; i++) {
    if (qoff[c.lo + i] < base || (i && qoff[c.lo + i] < qoff[c.lo + i - 1])) throw ArgError("query offsets must be non-decreasing");
    cb.h_off[i] = qoff[c.lo + i] - base;
  }
  if (cb.q.n < nbytes + 16) cb.q.alloc(nbytes + 16);  // the kernel reads whole aligned 8-byte words
  if (cb.off.n < n + 1) cb.off.alloc(n + 1);
  if (cb.counts.n < n) cb.counts.alloc(n);
  if (cb.status.n < n) cb.status.alloc(n);
  if (want_ranges && cb.ranges.n < 2 * n) cb.ranges.alloc(2 * n);
  if (nbytes) HIP_CHECK(hipMemcpyAsync(cb.q.p, qbytes + base, nbytes, hipMemcpyHostToDevice, r.stream));
  HIP_CHECK(hipMemcpyAsync(cb.off.p, cb.h_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, r.stream));
  launch_count_ascii(r, cb.q.p, cb.off.p, n, cb.counts.p, want_ranges ? cb.ranges.p : nullptr, cb.status.p, r.stream, allow_verify, 0, ref_kmer_len);
  cb.h_status.resize(n);
  HIP_CHECK(hipMemcpyAsync(cb.h_status.data(), cb.status.p, n, hipMemcpyDeviceToHost, r.stream));
}

void check_status(const ChunkBuffers& cb, uint64_t first_query) {
  for (size_t i = 0; i < cb.h_status.size(); i++)
    if (cb.h_status[i] != Q_OK) {
      static const char* why[] = {"", "empty query", "query contains '$' or '#'", "query contains a non-ASCII byte"};
      throw QueryError("query " + std::to_string(first_query + i) + ": " + why[cb.h_status[i] & 3] +
                       " (undefined in the reference: src/fm_index.rs:406, src/bwt.rs:126-128)");
    }
}

// n ASCII queries of L bytes each, back to back: counts (and status) only.  Nucleotide: packed on the device and served by
// the packed kernels.  Amino k-mers with a seed table: the two-phase schedule (count_aa_kmer_probe_kernel, then the
// generic kernel on what it listed).  Anything else: the generic kernel reading its queries at q * L.
// d_ranges (optional): (start, end) / RS_* words per query for the locate pass, as launch_count_ascii writes them.
void launch_count_ascii_uniform(Replica& r, const uint8_t* d_q, uint64_t n, uint64_t L, uint64_t* d_counts, uint8_t* d_status, hipStream_t s,
                                uint64_t* d_ranges = nullptr, unsigned long long* d_tally = nullptr) {
  if (n == 0) return;
  require(L >= 1, "query length must be at least 1");
  static const bool off = getenv("AWRY_AA_KMER") && !strcmp(getenv("AWRY_AA_KMER"), "0");
  static const bool no_long = getenv("AWRY_AA_LONG") && !strcmp(getenv("AWRY_AA_LONG"), "0");
  const bool two_phase = !off && r.dev.alphabet == AMINO && L >= (uint64_t)AA_KMER_MIN && L <= (uint64_t)(no_long ? AA_KMER_MAX : AA_KMER_LONG_MAX) &&
                         r.seed_k >= 1 && (uint64_t)r.seed_k <= L && n < (1ull << 32);
  Replica::SurvScratch* sc = surv_scratch(r, s);
  if (r.dev.alphabet == NUCLEOTIDE && !d_ranges && L <= 4096 && n < (1ull << 32)) {
    // the device half of the packed host path: pack 2 bits per letter, packed kernels, and the generic kernel over the
    // pack kernel's list for the queries with letters outside ACGT (it also writes their status)
    const uint64_t W = (L + 31) / 32;
    if (sc->u_words.n < n * W || sc->u_list.n < n || !sc->u_bad.p) {
      HIP_CHECK(hipStreamSynchronize(s));
      if (sc->u_words.n < n * W) sc->u_words.alloc(n * W + n * W / 4);
      if (sc->u_list.n < n) sc->u_list.alloc(n + n / 4);
      if (!sc->u_bad.p) sc->u_bad.alloc(2);
    }
    HIP_CHECK(hipMemsetAsync(sc->u_bad.p, 0, 16, s));
    if (d_status) HIP_CHECK(hipMemsetAsync(d_status, 0, n, s));
    launch_pack_nt2(r, d_q, nullptr, 0, n, n * L, (int)L, (int)W, sc->u_words.p, nullptr, sc->u_bad.p, s, sc->u_list.p);
    if (L <= 32) launch_count_nt2(r, sc->u_words.p, n, (int)L, d_counts, true, s, nullptr);
    else launch_count_nt2_long(r, sc->u_words.p, n, (int)L, d_counts, nullptr, true, s, nullptr);
    const QueryList ql{sc->u_list.p, nullptr, 0, sc->u_bad.p, nullptr, 0};
    hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_GLOBAL>), dim3((unsigned)r.num_cus * 2), dim3(256), 0, s, r.dev, d_q, nullptr, n,
                       d_counts, nullptr, d_status, 1, L, ql);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (!two_phase) {
    require(!d_tally, "the census is kept by the amino k-mer schedule only");
    launch_count_ascii(r, d_q, nullptr, n, d_counts, d_ranges, d_status, s, true, L);
    return;
  }
  launch_aa_two_phase(r, d_q, nullptr, n, (int)L, d_counts, d_ranges, d_status, s, d_tally);
}

// pins a caller-owned host range for the duration of a batch so that H2D/D2H run as real async DMA
struct HostPin {
  void* p = nullptr;
  HostPin(const void* ptr, size_t bytes) {
    if (ptr && bytes >= (8u << 20) && hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) == hipSuccess) p = const_cast<void*>(ptr);
    else (void)hipGetLastError();
  }
  ~HostPin() { if (p) (void)hipHostUnregister(p); }
};

// Fast path of parallel_count for the common k-mer batch: nucleotide index, every query the same length L,
// letters in ACGT.  Chunks flow through two stream lanes (H2D ASCII -> pack on device -> packed quad kernel ->
// D2H counts) so transfers overlap kernels; no per-query offsets cross PCIe.  A chunk that turns out to hold
// other bytes (N, IUPAC codes, '$' ...) is re-run through the generic kernel, so results never depend on the path.
// The lane buffers persist in the replica (PackedLane), so a call costs no device allocation.
// Can a shard of queries take the packed kernels, and how?  uniform: every query has Lmax letters; ragged: lengths in
// [1, Lmax], packed at a stride of W = ceil(Lmax / 32) words (accepted while that stride wastes little: the words of
// a query may take up to ~2x its own bytes).  Whether the letters are all ACGT is found out on the device.
struct PackedPlan {
  bool ok = false, ragged = false;
  uint64_t Lmax = 0;
};
PackedPlan plan_packed(const uint64_t* qoff, Shard sh) {
  PackedPlan plan;
  if (sh.hi <= sh.lo) return plan;
  const uint64_t n = sh.hi - sh.lo;
  auto scan = [&](uint64_t lo, uint64_t hi, uint64_t& mn, uint64_t& mx) {  // branch-free so the loop vectorises
    uint64_t a = ~0ull, b = 0;
    for (uint64_t i = lo; i < hi; i++) {
      const uint64_t d = qoff[i + 1] - qoff[i];
      a = d < a ? d : a;
      b = d > b ? d : b;
    }
    mn = a;
    mx = b;
  };
  uint64_t mn = ~0ull, mx = 0;
  if (n < (1u << 18)) {
    scan(sh.lo, sh.hi, mn, mx);
  } else {  // on the worker pool
    const uint64_t grain = 1u << 16, pieces = (n + grain - 1) / grain;
    std::vector<uint64_t> mns(pieces, ~0ull), mxs(pieces, 0);
    HostPool::instance().run(pieces, [&](uint64_t t) { scan(sh.lo + t * grain, std::min(sh.hi, sh.lo + (t + 1) * grain), mns[t], mxs[t]); });
    for (uint64_t t = 0; t < pieces; t++) { mn = std::min(mn, mns[t]); mx = std::max(mx, mxs[t]); }
  }
  if (mn == 0 || mx > 4096 || qoff[sh.hi] < qoff[sh.lo]) return plan;  // empty queries are the generic path's to reject
  plan.Lmax = mx;
  plan.ragged = mn != mx;
  if (plan.ragged) {
    const uint64_t bytes = qoff[sh.hi] - qoff[sh.lo], W = (mx + 31) / 32;
    if (mx > 512 || W * 8 * n > 2 * bytes + 16 * n) return plan;
  }
  plan.ok = true;
  return plan;
}

// chunks of a packed shard: at most max_q queries and ~max_bytes of ASCII each
std::vector<Shard> packed_chunks(const uint64_t* qoff, Shard sh, uint64_t max_q, uint64_t max_bytes) {
  std::vector<Shard> out;
  uint64_t a = sh.lo;
  while (a < sh.hi) {
    uint64_t b = std::min(sh.hi, a + max_q);
    if (qoff[b] - qoff[a] > max_bytes) {
      b = (uint64_t)(std::upper_bound(qoff + a, qoff + b + 1, qoff[a] + max_bytes) - qoff) - 1;
      b = std::max(b, a + 1);
    }
    out.push_back(Shard{a, b});
    a = b;
  }
  return out;
}

void launch_count_nt2(Replica& r, const uint64_t* d_words, uint64_t n, int L, uint64_t* d_counts, bool use_seed, hipStream_t s,
                      unsigned long long* d_tally);
void count_shard_generic(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, uint64_t* counts_out);

// The packed lanes with HOST packing -- the default path of parallel_count for nucleotide batches.  Per chunk: the pool
// packs the ASCII into the lane's pinned staging (2 bits per letter: 8 B per 31-mer cross PCIe instead of 31 B, and no
// caller memory is ever registered with the driver -- hipHostRegister of a fresh 150 MB batch cost more than its
// transfer), H2D, packed kernels, D2H of the counts into pinned staging, and on retirement a pool memcpy into
// counts_out (the first-touch faults of a fresh result array are taken by all threads).  The few queries with letters
// outside ACGT travel as a compact CSR batch of their own and are redone on the device by the generic kernel
// (LIST_COMPACT), overwriting their packed counts: results never depend on the path.
// words != nullptr: the caller's k-mers are packed already (awry_count_packed_kmers): staged with a pool memcpy.
// A caller that allocates its result array per call (a fresh Vec<u64>) hands over untouched pages: filling 40 MB of them
// costs ~10 000 page faults.  Advising huge pages for the 2 MB-aligned interior makes that ~20 (no effect where the pages
// are already there, or where transparent huge pages are off); the advice is the only thing done to the caller's mapping.
void advise_huge_pages(void* p, size_t bytes) {
  static const bool off = getenv("AWRY_NO_THP_ADVICE") != nullptr;
  if (off || bytes < (8u << 20)) return;
  const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
  const uintptr_t b = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(uintptr_t)((2u << 20) - 1);
  if (b > a) (void)madvise(reinterpret_cast<void*>(a), b - a, MADV_HUGEPAGE);
}

struct NotUniform {};  // thrown by count_shard_hostpacked(assume_uniform) when a query's length differs from the assumed one
void count_shard_hostpacked(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, PackedPlan plan, uint64_t* counts_out,
                            const uint64_t* words = nullptr, bool assume_uniform = false) {
  const uint64_t L = plan.Lmax, W = (L + 31) / 32;
  static const bool trace = getenv("AWRY_TRACE_HOST") != nullptr;
  static const uint64_t chunk_q = [] { const char* e = getenv("AWRY_HOST_CHUNK"); return e && atoll(e) > 0 ? (uint64_t)atoll(e) : (uint64_t)(1u << 20); }();
  // full-size chunks, then a tail that halves down to 128 K queries: what cannot overlap anything is the GPU time of
  // the last chunk, so it is kept small
  std::vector<Shard> chunks;
  for (uint64_t a = sh.lo; a < sh.hi;) {
    const uint64_t rest = sh.hi - a;
    uint64_t m = std::min(chunk_q, std::max<uint64_t>(rest / 2, std::min<uint64_t>(rest, 128u << 10)));
    if (rest - m < (64u << 10)) m = rest;
    if (!words && qoff[a + m] - qoff[a] > (256ull << 20)) {  // long reads: bound the bytes too
      m = (uint64_t)(std::upper_bound(qoff + a, qoff + a + m + 1, qoff[a] + (256ull << 20)) - qoff) - 1 - a;
      m = std::max<uint64_t>(m, 1);
    }
    chunks.push_back(Shard{a, a + m});
    a += m;
  }
  uint64_t cap_q = 0;
  for (Shard c : chunks) cap_q = std::max(cap_q, c.hi - c.lo);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  double t_pack = 0, t_wait = 0, t_out = 0, t_bad = 0;
  double t_enq[5] = {0, 0, 0, 0, 0};  // enqueue by operation: copy in, count kernels, listed reads, narrow + copy out, event
  const bool narrow32 = r.dev.bwt_len < (1ull << 32);  // a count is at most bwt_len
  advise_huge_pages(counts_out + sh.lo, (sh.hi - sh.lo) * 8);
  std::lock_guard<std::mutex> lane_lock(r.lane_mu);
  const auto t0 = now();
  PackedLane* lanes = r.lanes;
  uint64_t redone = 0;
  auto retire = [&](PackedLane& ln) {
    if (!ln.busy) return;
    ln.busy = false;
    auto a = now();
    HIP_CHECK(hipEventSynchronize(ln.done));
    auto b = now();
    if (narrow32) pool_widen_u32(counts_out + ln.chunk_lo, ln.h_counts32.p, ln.chunk_hi - ln.chunk_lo);
    else pool_memcpy(counts_out + ln.chunk_lo, ln.h_words.p, (ln.chunk_hi - ln.chunk_lo) * 8);  // (a count may pass 2^32: 64-bit words, staged where the packed words were)
    if (trace) { t_wait += ms(a, b); t_out += ms(b, now()); }
    redone += ln.nbad;
    if (ln.nbad && ln.h_bad[1] != ~0ull) {  // the lowest query of the chunk that the reference leaves undefined
      ChunkBuffers cb;
      cb.h_status.assign(1, (uint8_t)(ln.h_bad[1] & 0xFF));
      check_status(cb, ln.chunk_lo + (ln.h_bad[1] >> 8));
    }
  };
  struct Drain {  // every exit, normal or not, leaves the lanes idle
    Replica& r;
    ~Drain() {
      bool any = false;
      for (int li = 0; li < Replica::NLANES; li++)
        if (r.lanes[li].busy) { (void)hipStreamSynchronize(r.lane_stream[li]); r.lanes[li].busy = false; any = true; }
      if (any) { (void)hipStreamSynchronize(r.copy_in); (void)hipStreamSynchronize(r.copy_out); }
    }
  } drain{r};
  // AWRY_COPY_STREAMS=0: the chunk copies on the lane streams themselves, as before (A/B)
  static const bool copy_streams = !(getenv("AWRY_COPY_STREAMS") && !strcmp(getenv("AWRY_COPY_STREAMS"), "0"));
  const int nl = (int)std::min<size_t>(Replica::NLANES, chunks.size());
  for (int li = 0; li < nl; li++) {
    PackedLane& ln = lanes[li];
    ln.s = r.lane_stream[li];
    if (!ln.done) HIP_CHECK(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    if (!ln.ev_in) HIP_CHECK(hipEventCreateWithFlags(&ln.ev_in, hipEventDisableTiming));
    if (!ln.ev_k) HIP_CHECK(hipEventCreateWithFlags(&ln.ev_k, hipEventDisableTiming));
    if (!ln.h_bad) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ln.h_bad), 16, hipHostMallocDefault));
    ln.h_words.ensure(cap_q * W);
    ln.h_counts32.ensure(cap_q);
    if (ln.words.n < cap_q * W) ln.words.alloc(cap_q * W);
    if (ln.counts.n < cap_q) ln.counts.alloc(cap_q);
    if (ln.counts32.n < cap_q) ln.counts32.alloc(cap_q);
    if (plan.ragged) { ln.h_lens.ensure(cap_q); if (ln.lens.n < cap_q) ln.lens.alloc(cap_q); }
    if (ln.bad.n < 2) ln.bad.alloc(2);
  }
  const auto t1 = now();
  std::vector<uint32_t> bad;
  int which = 0;
  for (Shard c : chunks) {
    PackedLane& ln = lanes[which];
    which = (which + 1) % nl;
    retire(ln);
    const uint64_t lo = c.lo, hi = c.hi, n = hi - lo;
    ln.chunk_lo = lo;
    ln.chunk_hi = hi;
    auto a = now();
    if (words) { pool_memcpy(ln.h_words.p, words + lo, n * 8); bad.clear(); }
    // (assume_uniform: the chunks before this one have been checked, so qoff[lo] is where query lo starts either way)
    else if (!pack_nt2_host(qbytes + qoff[lo], qbytes + qoff[sh.hi], plan.ragged ? qoff : nullptr, lo, hi, L, ln.h_words.p,
                            plan.ragged ? ln.h_lens.p : nullptr, bad, assume_uniform ? qoff : nullptr))
      throw NotUniform{};  // (the Drain guard leaves the lanes idle; the caller plans the batch again from a full length scan)
    auto b = now();
    ln.nbad = bad.size();
    // (the lane's previous chunk has been retired: its kernels and its copy out are done, ln.words / ln.counts32 are free)
    hipStream_t cin = copy_streams ? r.copy_in : ln.s, cout = copy_streams ? r.copy_out : ln.s;
    HIP_CHECK(hipMemcpyAsync(ln.words.p, ln.h_words.p, n * W * 8, hipMemcpyHostToDevice, cin));
    if (plan.ragged) HIP_CHECK(hipMemcpyAsync(ln.lens.p, ln.h_lens.p, n * 4, hipMemcpyHostToDevice, cin));
    if (copy_streams) {
      HIP_CHECK(hipEventRecord(ln.ev_in, cin));
      HIP_CHECK(hipStreamWaitEvent(ln.s, ln.ev_in, 0));
    }
    auto e1 = now();
    if (L <= 32 && !plan.ragged) launch_count_nt2(r, ln.words.p, n, (int)L, ln.counts.p, true, ln.s, nullptr);
    else launch_count_nt2_long(r, ln.words.p, n, (int)L, ln.counts.p, nullptr, true, ln.s, plan.ragged ? ln.lens.p : nullptr);
    auto e2 = now();
    if (ln.nbad) {  // compact copy of the listed queries: indices, offsets, bytes
      const uint64_t nb = ln.nbad;
      ln.h_bq.ensure(nb);
      ln.h_boff.ensure(nb + 1);
      uint64_t tot = 0;
      for (uint64_t i = 0; i < nb; i++) {
        const uint64_t q = lo + bad[i];
        ln.h_bq.p[i] = bad[i];
        ln.h_boff.p[i] = tot;
        tot += qoff[q + 1] - qoff[q];
      }
      ln.h_boff.p[nb] = tot;
      ln.h_bbytes.ensure(tot + 16);
      HostPool::instance().run_ranges(nb, 4096, [&](uint64_t x, uint64_t y) {
        for (uint64_t i = x; i < y; i++) {
          const uint64_t q = lo + bad[i];
          memcpy(ln.h_bbytes.p + ln.h_boff.p[i], qbytes + qoff[q], qoff[q + 1] - qoff[q]);
        }
      });
      if (ln.bad_list.n < nb) ln.bad_list.alloc(nb + nb / 4 + 1024);
      if (ln.boff.n < nb + 1) ln.boff.alloc(nb + nb / 4 + 1024);
      if (ln.bbytes.n < tot + 16) ln.bbytes.alloc(tot + tot / 4 + 4096);
      HIP_CHECK(hipMemcpyAsync(ln.bad_list.p, ln.h_bq.p, nb * 4, hipMemcpyHostToDevice, ln.s));
      HIP_CHECK(hipMemcpyAsync(ln.boff.p, ln.h_boff.p, (nb + 1) * 8, hipMemcpyHostToDevice, ln.s));
      HIP_CHECK(hipMemcpyAsync(ln.bbytes.p, ln.h_bbytes.p, tot, hipMemcpyHostToDevice, ln.s));
      HIP_CHECK(hipMemsetAsync(ln.bad.p + 1, 0xFF, 8, ln.s));
      const QueryList ql{ln.bad_list.p, nullptr, nb, nullptr, ln.bad.p + 1, 0};
      hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_COMPACT>), dim3(grid_for(r, nb, 256)), dim3(256), 0, ln.s, r.dev, ln.bbytes.p,
                         ln.boff.p, n, ln.counts.p, nullptr, nullptr, 1, 0, ql);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(ln.h_bad + 1, ln.bad.p + 1, 8, hipMemcpyDeviceToHost, ln.s));
    }
    auto e3 = now();
    if (narrow32) {
      hipLaunchKernelGGL(narrow_counts_kernel, dim3(grid_for(r, n, 1024)), dim3(256), 0, ln.s, ln.counts.p, ln.counts32.p, n);
      HIP_CHECK(hipGetLastError());
    }
    if (copy_streams) {  // everything the lane stream holds for this chunk (the listed reads' small copies too) precedes the copy out
      HIP_CHECK(hipEventRecord(ln.ev_k, ln.s));
      HIP_CHECK(hipStreamWaitEvent(cout, ln.ev_k, 0));
    }
    if (narrow32) HIP_CHECK(hipMemcpyAsync(ln.h_counts32.p, ln.counts32.p, n * 4, hipMemcpyDeviceToHost, cout));
    else HIP_CHECK(hipMemcpyAsync(ln.h_words.p, ln.counts.p, n * 8, hipMemcpyDeviceToHost, cout));
    auto e4 = now();
    HIP_CHECK(hipEventRecord(ln.done, cout));
    ln.busy = true;
    if (trace) {
      t_pack += ms(a, b); t_bad += ms(b, now());
      t_enq[0] += ms(b, e1); t_enq[1] += ms(e1, e2); t_enq[2] += ms(e2, e3); t_enq[3] += ms(e3, e4); t_enq[4] += ms(e4, now());
    }
  }
  for (int k = 0; k < nl; k++) { retire(lanes[which]); which = (which + 1) % nl; }  // in chunk order
  if (trace)
    fprintf(stderr, "[awry] host-packed shard %llu queries%s L=%llu, %zu chunks, %u pool threads: lane setup %.2f ms, pipeline %.2f ms (%s %.2f, enqueue %.2f "
            "[copy in %.2f, count kernels %.2f, listed reads %.2f, narrow + copy out %.2f, event %.2f], "
            "waiting for the GPU %.2f, copying counts out %.2f), %llu redone by the generic kernel\n",
            (unsigned long long)(sh.hi - sh.lo), plan.ragged ? " (ragged)" : "", (unsigned long long)L, chunks.size(), HostPool::instance().threads(),
            ms(t0, t1), ms(t1, now()), words ? "staging" : "host pack", t_pack, t_bad, t_enq[0], t_enq[1], t_enq[2], t_enq[3], t_enq[4], t_wait, t_out,
            (unsigned long long)redone);
}

// the packed lanes with DEVICE packing (round 1's path, kept for A/B: AWRY_HOST_PACK=0): the chunk's ASCII crosses PCIe
// from caller memory registered in place and pack_nt2_tile_kernel packs it
void count_shard_packed(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, PackedPlan plan, uint64_t* counts_out) {
  const uint64_t L = plan.Lmax;
  const int W = (int)((L + 31) / 32);
  const std::vector<Shard> chunks = packed_chunks(qoff, sh, 4u << 20, 256ull << 20);
  uint64_t cap_q = 0, cap_b = 0;
  for (Shard c : chunks) { cap_q = std::max(cap_q, c.hi - c.lo); cap_b = std::max(cap_b, qoff[c.hi] - qoff[c.lo]); }
  static const bool trace = getenv("AWRY_TRACE_HOST") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  std::lock_guard<std::mutex> lane_lock(r.lane_mu);
  const auto t0 = now();
  // the output range is pinned by a helper thread (first-touch page faults of a fresh buffer dominate it) while
  // this thread pins the input and starts the first chunk; joined before the first D2H is queued
  std::unique_ptr<HostPin> pin_out;
  std::thread pin_out_thread([&] {
    (void)hipSetDevice(r.device);
    pin_out.reset(new HostPin(counts_out + sh.lo, (sh.hi - sh.lo) * 8));
  });
  struct Joiner {
    std::thread& t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } joiner{pin_out_thread};
  HostPin pin_in(qbytes + qoff[sh.lo], qoff[sh.hi] - qoff[sh.lo]);
  HostPin pin_off(plan.ragged ? qoff + sh.lo : nullptr, (sh.hi - sh.lo + 1) * 8);
  const auto t1 = now();
  PackedLane* lanes = r.lanes;
  uint64_t redone = 0;
  auto retire = [&](PackedLane& ln) {
    if (!ln.busy) return;
    ln.busy = false;
    HIP_CHECK(hipEventSynchronize(ln.done));
    redone += ln.h_bad[0];
    if (ln.h_bad[1] != ~0ull) {  // the lowest query of the chunk that the reference leaves undefined
      ChunkBuffers cb;
      cb.h_status.assign(1, (uint8_t)(ln.h_bad[1] & 0xFF));
      check_status(cb, ln.chunk_lo + (ln.h_bad[1] >> 8));
    }
  };
  // every exit, normal or not, leaves the lanes idle before the host ranges are unpinned
  struct Drain {
    Replica& r;
    ~Drain() {
      for (int li = 0; li < 2; li++)
        if (r.lanes[li].busy) { (void)hipStreamSynchronize(r.lane_stream[li]); r.lanes[li].busy = false; }
    }
  } drain{r};
  for (int li = 0; li < 2; li++) {
    PackedLane& ln = lanes[li];
    ln.s = r.lane_stream[li];
    if (!ln.done) HIP_CHECK(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    if (!ln.h_bad) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ln.h_bad), 16, hipHostMallocDefault));
    if (ln.ascii.n < cap_b + 16) ln.ascii.alloc(cap_b + 16);
    if (ln.words.n < cap_q * W) ln.words.alloc(cap_q * W);
    if (ln.counts.n < cap_q) ln.counts.alloc(cap_q);
    if (plan.ragged && ln.off.n < cap_q + 1) ln.off.alloc(cap_q + 1);
    if (plan.ragged && ln.lens.n < cap_q) ln.lens.alloc(cap_q);
    if (ln.bad_list.n < cap_q) ln.bad_list.alloc(cap_q);
    if (ln.bad.n < 2) ln.bad.alloc(2);
  }
  const auto t2 = now();
  int which = 0;
  for (Shard c : chunks) {
    PackedLane& ln = lanes[which];
    which ^= 1;
    retire(ln);
    const uint64_t lo = c.lo, hi = c.hi, n = hi - lo, nbytes = qoff[hi] - qoff[lo];
    ln.chunk_lo = lo;
    ln.chunk_hi = hi;
    HIP_CHECK(hipMemcpyAsync(ln.ascii.p, qbytes + qoff[lo], nbytes, hipMemcpyHostToDevice, ln.s));
    if (plan.ragged) HIP_CHECK(hipMemcpyAsync(ln.off.p, qoff + lo, (n + 1) * 8, hipMemcpyHostToDevice, ln.s));
    HIP_CHECK(hipMemsetAsync(ln.bad.p, 0, 8, ln.s));
    HIP_CHECK(hipMemsetAsync(ln.bad.p + 1, 0xFF, 8, ln.s));
    launch_pack_nt2(r, ln.ascii.p, plan.ragged ? ln.off.p : nullptr, qoff[lo], n, nbytes, (int)L, W, ln.words.p,
                    plan.ragged ? ln.lens.p : nullptr, ln.bad.p, ln.s, ln.bad_list.p);
    if (L <= 32 && !plan.ragged) launch_count_nt2(r, ln.words.p, n, (int)L, ln.counts.p, true, ln.s, nullptr);
    else launch_count_nt2_long(r, ln.words.p, n, (int)L, ln.counts.p, nullptr, true, ln.s, plan.ragged ? ln.lens.p : nullptr);
    {
      // The queries the pack kernel listed (N, IUPAC codes, 'U', '$' ...) are redone by the generic kernel where they lie,
      // from the chunk's ASCII in HBM, and overwrite their packed counts: results never depend on the path, and a batch
      // of real reads -- a few such reads in every chunk -- pays one small launch per chunk, no trip through the host.
      const QueryList ql{ln.bad_list.p, nullptr, 0, ln.bad.p, ln.bad.p + 1, 0};
      const uint8_t* bytes = plan.ragged ? reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(ln.ascii.p) - qoff[lo]) : ln.ascii.p;
      hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_GLOBAL>), dim3((unsigned)r.num_cus * 2), dim3(256), 0, ln.s, r.dev, bytes,
                         plan.ragged ? ln.off.p : nullptr, n, ln.counts.p, nullptr, nullptr, 1, plan.ragged ? 0 : L, ql);
      HIP_CHECK(hipGetLastError());
    }
    if (pin_out_thread.joinable()) pin_out_thread.join();
    HIP_CHECK(hipMemcpyAsync(counts_out + lo, ln.counts.p, n * 8, hipMemcpyDeviceToHost, ln.s));
    HIP_CHECK(hipMemcpyAsync(ln.h_bad, ln.bad.p, 16, hipMemcpyDeviceToHost, ln.s));
    HIP_CHECK(hipEventRecord(ln.done, ln.s));
    ln.busy = true;
  }
  for (int li = 0; li < 2; li++) retire(lanes[li]);
  if (trace) fprintf(stderr, "[awry] packed shard %llu queries%s: pin %.2f ms, lane setup %.2f ms, pipeline %.2f ms, %llu redone by the generic kernel\n",
                     (unsigned long long)(sh.hi - sh.lo), plan.ragged ? " (ragged)" : "", ms(t0, t1), ms(t1, t2), ms(t2, now()), (unsigned long long)redone);
}

// Generic kernel, pipelined like the packed path: any alphabet, any letters, any lengths (amino batches, long or very
// unequal nucleotide reads).  Pinned input / offsets / output, two stream lanes, persistent lane buffers; the kernel
// reads the chunk's queries through the batch's own offsets (the ASCII pointer is biased by the chunk's first byte).
// ulen != 0: every query of the shard has ulen bytes -- the offsets stay on the host and the kernels address query q at q * ulen.
void count_shard_generic_pipelined(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, uint64_t* counts_out, uint64_t ulen = 0) {
  if (!ulen)  // (one length: the length scan has seen every offset already)
    for (uint64_t i = sh.lo; i < sh.hi; i++)  // (vectorises) non-decreasing offsets
      if (qoff[i + 1] < qoff[i]) throw ArgError("query offsets must be non-decreasing");
  const std::vector<Shard> chunks = packed_chunks(qoff, sh, 1u << 20, 128ull << 20);
  uint64_t cap_q = 0, cap_b = 0;
  for (Shard c : chunks) { cap_q = std::max(cap_q, c.hi - c.lo); cap_b = std::max(cap_b, qoff[c.hi] - qoff[c.lo]); }
  std::lock_guard<std::mutex> lane_lock(r.lane_mu);
  static const bool trace = getenv("AWRY_TRACE_HOST") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // Like the host-packed lanes: the chunk's bytes (and offsets) are copied by the pool into the lane's pinned staging --
  // nothing of the caller's is registered with the driver --, counts come back as 32-bit words when they fit (a count is
  // at most bwt_len) and are widened into counts_out, and instead of one status byte per query the lowest rejected
  // query crosses PCIe as one word.
  const bool narrow32 = r.dev.bwt_len < (1ull << 32);
  PackedLane* lanes = r.lanes;
  double t_stage = 0, t_wait = 0, t_out = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto retire = [&](PackedLane& ln) {
    if (!ln.busy) return;
    ln.busy = false;
    auto a = now();
    HIP_CHECK(hipEventSynchronize(ln.done));
    auto b = now();
    const uint64_t n = ln.chunk_hi - ln.chunk_lo;
    if (narrow32) pool_widen_u32(counts_out + ln.chunk_lo, ln.h_counts32.p, n);
    else pool_memcpy(counts_out + ln.chunk_lo, ln.h_words.p, n * 8);
    if (trace) { t_wait += ms(a, b); t_out += ms(b, now()); }
    if (ln.h_bad[1] != ~0ull) {
      ChunkBuffers cb;
      cb.h_status.assign(1, (uint8_t)(ln.h_bad[1] & 0xFF));
      check_status(cb, ln.chunk_lo + (ln.h_bad[1] >> 8));  // raises INVALID_QUERY naming the first such query
    }
  };
  struct Drain {
    Replica& r;
    ~Drain() {
      for (int li = 0; li < Replica::NLANES; li++)
        if (r.lanes[li].busy) { (void)hipStreamSynchronize(r.lane_stream[li]); r.lanes[li].busy = false; }
    }
  } drain{r};
  const int nl = (int)std::min<size_t>(Replica::NLANES, chunks.size());
  for (int li = 0; li < nl; li++) {
    PackedLane& ln = lanes[li];
    ln.s = r.lane_stream[li];
    if (!ln.done) HIP_CHECK(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    if (!ln.h_bad) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ln.h_bad), 16, hipHostMallocDefault));
    ln.h_bbytes.ensure(cap_b + 16);
    if (ln.ascii.n < cap_b + 16) ln.ascii.alloc(cap_b + 16);
    if (!ulen) { ln.h_boff.ensure(cap_q + 1); if (ln.off.n < cap_q + 1) ln.off.alloc(cap_q + 1); }
    if (ln.counts.n < cap_q) ln.counts.alloc(cap_q);
    if (ln.status.n < cap_q) ln.status.alloc(cap_q);
    if (ln.bad.n < 2) ln.bad.alloc(2);
    if (narrow32) { ln.h_counts32.ensure(cap_q); if (ln.counts32.n < cap_q) ln.counts32.alloc(cap_q); }
    else ln.h_words.ensure(cap_q);  // (pinned staging of the 64-bit counts)
  }
  int which = 0;
  for (Shard c : chunks) {
    PackedLane& ln = lanes[which];
    which = (which + 1) % nl;
    retire(ln);
    const uint64_t lo = c.lo, hi = c.hi, n = hi - lo, base = qoff[lo], nbytes = qoff[hi] - base;
    ln.chunk_lo = lo;
    ln.chunk_hi = hi;
    auto a = now();
    if (nbytes) pool_memcpy(ln.h_bbytes.p, qbytes + base, nbytes);
    if (!ulen) pool_memcpy(ln.h_boff.p, qoff + lo, (n + 1) * 8);
    if (trace) t_stage += ms(a, now());
    if (nbytes) HIP_CHECK(hipMemcpyAsync(ln.ascii.p, ln.h_bbytes.p, nbytes, hipMemcpyHostToDevice, ln.s));
    if (ulen) {
      launch_count_ascii_uniform(r, ln.ascii.p, n, ulen, ln.counts.p, ln.status.p, ln.s);
    } else {
      HIP_CHECK(hipMemcpyAsync(ln.off.p, ln.h_boff.p, (n + 1) * 8, hipMemcpyHostToDevice, ln.s));
      const uint8_t* biased = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(ln.ascii.p) - base);
      launch_count_ascii(r, biased, ln.off.p, n, ln.counts.p, nullptr, ln.status.p, ln.s, true);
    }
    HIP_CHECK(hipMemsetAsync(ln.bad.p + 1, 0xFF, 8, ln.s));
    hipLaunchKernelGGL(status_first_bad_kernel, dim3(grid_for(r, n, 4096)), dim3(256), 0, ln.s, ln.status.p, n, ln.bad.p + 1);
    if (narrow32) {
      hipLaunchKernelGGL(narrow_counts_kernel, dim3(grid_for(r, n, 1024)), dim3(256), 0, ln.s, ln.counts.p, ln.counts32.p, n);
      HIP_CHECK(hipMemcpyAsync(ln.h_counts32.p, ln.counts32.p, n * 4, hipMemcpyDeviceToHost, ln.s));
    } else {
      HIP_CHECK(hipMemcpyAsync(ln.h_words.p, ln.counts.p, n * 8, hipMemcpyDeviceToHost, ln.s));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(ln.h_bad + 1, ln.bad.p + 1, 8, hipMemcpyDeviceToHost, ln.s));
    HIP_CHECK(hipEventRecord(ln.done, ln.s));
    ln.busy = true;
  }
  for (int k2 = 0; k2 < nl; k2++) { retire(lanes[which]); which = (which + 1) % nl; }
  if (trace)
    fprintf(stderr, "[awry] generic shard %llu queries%s, %zu chunks: %.2f ms (staging %.2f, waiting for the GPU %.2f, copying counts out %.2f)\n",
            (unsigned long long)(sh.hi - sh.lo), ulen ? " (one length)" : "", chunks.size(), ms(t0, now()), t_stage, t_wait, t_out);
}

void count_shard(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, uint64_t* counts_out) {
  HIP_CHECK(hipSetDevice(r.device));
  static const bool no_fast = getenv("AWRY_HOST_PATH") && !strcmp(getenv("AWRY_HOST_PATH"), "generic");
  const auto t0 = std::chrono::steady_clock::now();
  static const bool dev_pack = getenv("AWRY_HOST_PACK") && !strcmp(getenv("AWRY_HOST_PACK"), "0");
  PackedPlan plan;
  const bool packable = !no_fast && r.dev.alphabet == NUCLEOTIDE;  // (wide-row replicas take the 64-bit packed kernels)
  if (packable && !dev_pack && sh.hi - sh.lo >= (1u << 16)) {
    // k-mer and read batches are nearly always of one length: assume the first query's, let the packer check the
    // offsets in the pass that reads the bytes anyway (a separate scan of 8 B per query costs 7 % of a 31-mer batch)
    const uint64_t n = sh.hi - sh.lo, L0 = qoff[sh.lo + 1] - qoff[sh.lo];
    if (L0 >= 1 && L0 <= 4096 && qoff[sh.hi] >= qoff[sh.lo] && qoff[sh.hi] - qoff[sh.lo] == n * L0) {
      PackedPlan guess;
      guess.ok = true;
      guess.Lmax = L0;
      try {
        count_shard_hostpacked(r, qbytes, qoff, sh, guess, counts_out, nullptr, true);
        return;
      } catch (const NotUniform&) {  // plan it properly below; what was written to counts_out is overwritten
      }
    }
  }
  if (packable) plan = plan_packed(qoff, sh);
  if (getenv("AWRY_TRACE_HOST"))
    fprintf(stderr, "[awry] length scan %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  if (plan.ok) {
    if (dev_pack) count_shard_packed(r, qbytes, qoff, sh, plan, counts_out);
    else count_shard_hostpacked(r, qbytes, qoff, sh, plan, counts_out);
    return;
  }
  if (!no_fast && sh.hi - sh.lo >= 4096) {
    // amino batches of one length (k-mers): no offsets cross PCIe and the two-phase amino schedule serves them
    uint64_t ulen = 0;
    if (r.dev.alphabet == AMINO) {
      const PackedPlan ap = plan_packed(qoff, sh);
      if (ap.ok && !ap.ragged) ulen = ap.Lmax;
      if (getenv("AWRY_TRACE_HOST"))
        fprintf(stderr, "[awry] amino length scan %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    count_shard_generic_pipelined(r, qbytes, qoff, sh, counts_out, ulen);
  } else count_shard_generic(r, qbytes, qoff, sh, counts_out);
}

void count_shard_generic(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, uint64_t* counts_out) {
  HIP_CHECK(hipSetDevice(r.device));
  ChunkBuffers cb;
  for (Shard c : chunk_queries(qoff, sh.lo, sh.hi)) {
    run_count_chunk(r, cb, qbytes, qoff, c, false);
    HIP_CHECK(hipMemcpyAsync(counts_out + c.lo, cb.counts.p, (c.hi - c.lo) * 8, hipMemcpyDeviceToHost, r.stream));
    HIP_CHECK(hipStreamSynchronize(r.stream));
    check_status(cb, c.lo);
  }
}

// Result arrays of the batch entry points (offsets, positions, (record, offset) pairs) are PINNED host memory, recycled
// through a process-wide pool: the locate kernels' output is copied by the DMA engine straight into the array the caller
// receives -- no pinned staging, no host memcpy, and after the first call no first-touch page faults either (a fresh
// 100 MB array costs more in faults than its bytes cost on PCIe).  awry_free_buffer returns a block to the pool; blocks
// are kept up to AWRY_PINNED_CACHE_GB (default 4) and otherwise released.  Where pinning fails the arrays are plain
// malloc memory and the copies are staged by the runtime.
class PinnedPool {
 public:
  static PinnedPool& instance() {
    static PinnedPool* pool = new PinnedPool;  // never destroyed: the HIP runtime may be gone before static destructors run
    return *pool;
  }
  void* get(size_t bytes) {  // >= bytes of pinned memory, or nullptr
    const size_t want = round_up(bytes);
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = free_.lower_bound(want);
      if (it != free_.end() && it->first <= 2 * want) {
        void* p = it->second;
        cached_ -= it->first;
        live_[p] = it->first;
        free_.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(mu_);
    live_[p] = want;
    return p;
  }
  bool put(void* p) {  // false: not a block of this pool
    size_t bytes = 0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      auto it = live_.find(p);
      if (it == live_.end()) return false;
      bytes = it->second;
      live_.erase(it);
      if (cached_ + bytes <= cap_) {
        free_.emplace(bytes, p);
        cached_ += bytes;
        return true;
      }
    }
    (void)hipHostFree(p);
    return true;
  }
 private:
  PinnedPool() {
    const char* e = getenv("AWRY_PINNED_CACHE_GB");
    cap_ = (size_t)((e && atof(e) >= 0 ? atof(e) : 4.0) * (double)(1ull << 30));
  }
  static size_t round_up(size_t b) {  // 1 MiB, then powers of two up to 64 MiB, then multiples of 64 MiB
    size_t c = 1u << 20;
    while (c < b && c < (64u << 20)) c <<= 1;
    return c >= b ? c : (b + (64u << 20) - 1) / (64u << 20) * (64u << 20);
  }
  std::mutex mu_;
  std::multimap<size_t, void*> free_;
  std::map<void*, size_t> live_;
  size_t cached_ = 0, cap_ = 0;
};

void release_result(void* p) {
  if (p && !PinnedPool::instance().put(p)) free(p);
}

template <class T>
struct MBuf {  // geometrically growing result array whose storage is handed to the caller (released with awry_free_buffer)
  T* p = nullptr;
  size_t cap = 0;
  size_t used_bytes = 0;  // bytes of p[] that hold data (what a re-allocation has to carry over)
  MBuf() = default;
  MBuf(const MBuf&) = delete;
  MBuf& operator=(const MBuf&) = delete;
  MBuf(MBuf&& o) noexcept : p(o.p), cap(o.cap), used_bytes(o.used_bytes) { o.p = nullptr; o.cap = 0; o.used_bytes = 0; }
  ~MBuf() { release_result(p); }
  void grow(size_t need) {
    if (need <= cap) return;
    const size_t c = std::max(need, cap + cap / 2 + 4096), bytes = c * sizeof(T);
    void* q = bytes >= (256u << 10) ? PinnedPool::instance().get(bytes) : nullptr;
    if (!q) q = malloc(bytes);
    if (!q) throw std::bad_alloc();
    if (p && used_bytes) pool_memcpy(q, p, used_bytes);
    release_result(p);
    p = static_cast<T*>(q);
    cap = c;
  }
  T* release() { T* q = p; p = nullptr; cap = 0; used_bytes = 0; return q; }
};

struct LocateResult {  // per shard, in query order
  uint64_t* off = nullptr;  // the shard's slice of the batch's offset array: off[i + 1] - off[i] = hits of query i; the
                            //   shard writes off[1..n] relative to its own first hit, the caller rebases
  uint64_t nq = 0, filled = 0, running = 0;
  MBuf<uint64_t> gpos;
  MBuf<awry_pos_t> pos;
  size_t total = 0;      // hits whose results are in (or on their way into) the arrays
  bool want_pos = true;  // false: the caller passed hits_out == NULL -- (record, offset) pairs are neither computed nor moved
  void add_counts(const uint64_t* counts, uint64_t n) {  // next n queries of the shard
    for (uint64_t i = 0; i < n; i++) { running += counts[i]; off[filled + i + 1] = running; }
    filled += n;
  }
  // next n queries of the shard, whose inclusive hit offsets RELATIVE TO THE CHUNK already sit in off[filled + 1 ...]
  // (copied there from the device scan): rebase them onto the shard's running total
  void rebase_offsets(uint64_t n, uint64_t chunk_total) {
    uint64_t* o = off + filled + 1;
    const uint64_t base = running;
    if (base) HostPool::instance().run_ranges(n, 1u << 16, [&](uint64_t a, uint64_t b) { for (uint64_t i = a; i < b; i++) o[i] += base; });
    running += chunk_total;
    filled += n;
  }
  // room for n more hits; true when an array moved (copies in flight into the old one must have finished: see `quiesce`)
  template <class Quiesce>
  void reserve(size_t n, bool want_gpos, Quiesce&& quiesce) {
    if ((!want_pos || total + n <= pos.cap) && (!want_gpos || total + n <= gpos.cap)) return;
    size_t need = total + n;
    if (filled && filled < nq) need = std::max(need, (size_t)((double)(total + n) / (double)filled * (double)nq * 1.05) + 4096);  // the whole shard, from the hit rate so far
    quiesce();
    if (want_pos) { pos.used_bytes = total * sizeof(awry_pos_t); pos.grow(need); }
    if (want_gpos) { gpos.used_bytes = total * 8; gpos.grow(need); }
  }
  void append(const uint64_t* g, const awry_pos_t* p, size_t n, bool want_gpos) {
    if (!n) return;
    reserve(n, want_gpos, [] {});
    if (want_pos) pool_memcpy(pos.p + total, p, n * sizeof(awry_pos_t));
    if (want_gpos) pool_memcpy(gpos.p + total, g, n * 8);
    total += n;
  }
};

// generic kernels, synchronous: any alphabet, ragged lengths, ambiguity codes
void locate_chunk_generic(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard c, uint64_t* counts_out,
                          std::vector<uint64_t>& gpos, std::vector<awry_pos_t>& pos, bool want_pos) {
  ChunkBuffers cb;
  const uint64_t n = c.hi - c.lo;
  run_count_chunk(r, cb, qbytes, qoff, c, true);
  DevBuf<uint64_t> hit_off(n + 1), scratch(scan_tiles(n) + 1);
  launch_scan(r, cb.counts.p, n, hit_off.p, scratch.p, r.stream);
  uint64_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, hit_off.p + n, 8, hipMemcpyDeviceToHost, r.stream));
  HIP_CHECK(hipMemcpyAsync(counts_out, cb.counts.p, n * 8, hipMemcpyDeviceToHost, r.stream));
  HIP_CHECK(hipStreamSynchronize(r.stream));
  check_status(cb, c.lo);
  gpos.resize(total);
  pos.resize(want_pos ? total : 0);
  if (total == 0) return;
  DevBuf<uint64_t> d_gpos(total), d_pos(want_pos ? 2 * total : 0);
  launch_locate(r, cb.ranges.p, 2, hit_off.p, n, total, d_gpos.p, d_pos.p, r.stream);
  if (want_pos) HIP_CHECK(hipMemcpyAsync(pos.data(), d_pos.p, total * 16, hipMemcpyDeviceToHost, r.stream));
  HIP_CHECK(hipMemcpyAsync(gpos.data(), d_gpos.p, total * 8, hipMemcpyDeviceToHost, r.stream));
  HIP_CHECK(hipStreamSynchronize(r.stream));
}

void locate_shard_generic(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, bool want_gpos, LocateResult& out) {
  std::vector<uint64_t> g, counts;
  std::vector<awry_pos_t> p;
  for (Shard c : chunk_queries(qoff, sh.lo, sh.hi)) {
    counts.resize(c.hi - c.lo);
    locate_chunk_generic(r, qbytes, qoff, c, counts.data(), g, p, out.want_pos);
    out.add_counts(counts.data(), c.hi - c.lo);
    out.append(g.data(), p.data(), g.size(), want_gpos);
  }
}

// Fast path of parallel_locate: nucleotide index, every read the same length L.  Chunks of reads flow through the
// replica's two stream lanes in three stages -- (1) H2D ASCII, pack, packed count with range starts, scan, D2H counts;
// (2) once the host knows the chunk's hit total: locate kernels, D2H of the positions into pinned staging; (3) copy
// into the result arrays -- so that one chunk's transfers and host copies overlap the other chunk's kernels.  A chunk
// that holds bytes outside ACGT is redone by the generic kernels; results never depend on the path.
// plan.ok == false: the same pipeline around the generic kernel (any alphabet, letters and lengths; ranges as two words
// per query, statuses checked in stage 2)
void locate_shard_packed(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, PackedPlan plan, bool want_gpos, LocateResult& out) {
  const bool generic = !plan.ok, want_pos = out.want_pos;
  uint64_t ulen = 0;  // generic, every query of one length (amino k-mers): no offsets travel, the amino k-mer schedule counts
  if (generic) {
    if (r.dev.alphabet == AMINO) {
      const PackedPlan ap = plan_packed(qoff, sh);
      if (ap.ok && !ap.ragged) ulen = ap.Lmax;
    }
    plan.ragged = ulen == 0;  // offsets travel with the chunk
    plan.Lmax = 1;
    if (!ulen)
      for (uint64_t i = sh.lo; i < sh.hi; i++)
        if (qoff[i + 1] < qoff[i]) throw ArgError("query offsets must be non-decreasing");
  }
  const uint64_t L = plan.Lmax, W = (L + 31) / 32;
  // nucleotide reads are packed on the HOST (2 bits per letter cross PCIe, nothing of the caller's is registered with the
  // driver), as in count_shard_hostpacked; AWRY_HOST_PACK=0 keeps round 1's device packing for A/B
  static const bool dev_pack = getenv("AWRY_HOST_PACK") && !strcmp(getenv("AWRY_HOST_PACK"), "0");
  const bool hostpack = !generic && !dev_pack;
  const std::vector<Shard> chunks = packed_chunks(qoff, sh, 1u << 20, 128ull << 20);
  uint64_t cap = 0, cap_b = 0;
  for (Shard c : chunks) { cap = std::max(cap, c.hi - c.lo); cap_b = std::max(cap_b, qoff[c.hi] - qoff[c.lo]); }
  static const bool trace = getenv("AWRY_TRACE_HOST") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> lane_lock(r.lane_mu);
  HostPin pin_in(hostpack ? nullptr : qbytes + qoff[sh.lo], qoff[sh.hi] - qoff[sh.lo]);
  HostPin pin_off(plan.ragged && !hostpack ? qoff + sh.lo : nullptr, (sh.hi - sh.lo + 1) * 8);
  std::vector<uint32_t> bad;
  double t_pack = 0;
  LocateLane* lanes = r.loc_lanes;
  double t_pin = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), t_wait_count = 0, t_wait_locate = 0;
  double t_grow_host = 0, t_grow_dev = 0;  // result arrays (pinned pool) and the lanes' device hit buffers that had to grow
  int n_grow_host = 0, n_grow_dev = 0;
  auto timed = [&](double& acc, auto&& fn) {
    if (!trace) { fn(); return; }
    const auto a = std::chrono::steady_clock::now();
    fn();
    acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
  };
  struct Drain {  // every exit leaves the lanes idle before the input is unpinned
    Replica& r;
    ~Drain() {
      bool any = false;
      for (int li = 0; li < 2; li++) {
        if (r.loc_lanes[li].stage) { (void)hipStreamSynchronize(r.lane_stream[li]); any = true; }
        r.loc_lanes[li].stage = 0;
      }
      if (any) { (void)hipStreamSynchronize(r.copy_in); (void)hipStreamSynchronize(r.copy_out); }
    }
  } drain{r};
  // chunk copies on the replica's copy-in / copy-out streams (see Replica::copy_in); AWRY_COPY_STREAMS=0: on the lane streams
  static const bool copy_streams = !(getenv("AWRY_COPY_STREAMS") && !strcmp(getenv("AWRY_COPY_STREAMS"), "0"));
  for (int li = 0; li < 2; li++) {
    LocateLane& ln = lanes[li];
    if (!ln.counted) HIP_CHECK(hipEventCreateWithFlags(&ln.counted, hipEventDisableTiming));
    if (!ln.located) HIP_CHECK(hipEventCreateWithFlags(&ln.located, hipEventDisableTiming));
    if (!ln.ev_in) HIP_CHECK(hipEventCreateWithFlags(&ln.ev_in, hipEventDisableTiming));
    if (!ln.ev_k) HIP_CHECK(hipEventCreateWithFlags(&ln.ev_k, hipEventDisableTiming));
    if (!hostpack && ln.ascii.n < cap_b + 16) ln.ascii.alloc(cap_b + 16);
    if (ln.words.n < cap * W) ln.words.alloc(cap * W);
    if (plan.ragged && !hostpack && ln.off.n < cap + 1) ln.off.alloc(cap + 1);
    if (plan.ragged && ln.lens.n < cap) ln.lens.alloc(cap);
    if (!hostpack && ln.bad_list.n < cap) ln.bad_list.alloc(cap);
    if (hostpack) { ln.h_words.ensure(cap * W); if (plan.ragged) ln.h_lens.ensure(cap); }
    if (ln.rstart.n < (generic ? 2 : 1) * cap) ln.rstart.alloc((generic ? 2 : 1) * cap);
    if (generic && ln.status.n < cap) ln.status.alloc(cap);
    if (generic) ln.h_status.ensure(cap);
    if (ln.counts.n < cap) ln.counts.alloc(cap);
    if (ln.hit_off.n < cap + 1) ln.hit_off.alloc(cap + 1);
    if (ln.scratch.n < scan_tiles(cap) + 1) ln.scratch.alloc(scan_tiles(cap) + 1);
    if (ln.bad.n < 2) ln.bad.alloc(2);
    ln.h_meta.ensure(3);
    ln.stage = 0;
  }
  auto stage1 = [&](int li, uint64_t lo, uint64_t hi) {  // count
    LocateLane& ln = lanes[li];
    hipStream_t s = r.lane_stream[li];
    const uint64_t n = hi - lo;
    ln.lo = lo;
    ln.hi = hi;
    const uint64_t nbytes = qoff[hi] - qoff[lo];
    if (hostpack) {
      timed(t_pack, [&] {
        pack_nt2_host(qbytes + qoff[lo], qbytes + qoff[sh.hi], plan.ragged ? qoff : nullptr, lo, hi, L, ln.h_words.p, plan.ragged ? ln.h_lens.p : nullptr, bad);
      });
      hipStream_t cin = copy_streams ? r.copy_in : s;
      HIP_CHECK(hipMemcpyAsync(ln.words.p, ln.h_words.p, n * W * 8, hipMemcpyHostToDevice, cin));
      if (plan.ragged) HIP_CHECK(hipMemcpyAsync(ln.lens.p, ln.h_lens.p, n * 4, hipMemcpyHostToDevice, cin));
      if (copy_streams) {
        HIP_CHECK(hipEventRecord(ln.ev_in, cin));
        HIP_CHECK(hipStreamWaitEvent(s, ln.ev_in, 0));
      }
      HIP_CHECK(hipMemsetAsync(ln.bad.p, 0, 8, s));
      HIP_CHECK(hipMemsetAsync(ln.bad.p + 1, 0xFF, 8, s));
      launch_count_nt2_long(r, ln.words.p, n, (int)L, ln.counts.p, ln.rstart.p, true, s, plan.ragged ? ln.lens.p : nullptr);
      if (const uint64_t nb = bad.size()) {
        // reads with other bytes (N, IUPAC codes ...): a compact CSR batch of their own, redone by the generic kernel, which
        // overwrites their counts and range words (starts only, the packed kernels' layout) before the scan
        ln.h_bq.ensure(nb);
        ln.h_boff.ensure(nb + 1);
        uint64_t tot = 0;
        for (uint64_t i = 0; i < nb; i++) {
          ln.h_bq.p[i] = bad[i];
          ln.h_boff.p[i] = tot;
          tot += qoff[lo + bad[i] + 1] - qoff[lo + bad[i]];
        }
        ln.h_boff.p[nb] = tot;
        ln.h_bbytes.ensure(tot + 16);
        HostPool::instance().run_ranges(nb, 4096, [&](uint64_t x, uint64_t y) {
          for (uint64_t i = x; i < y; i++) memcpy(ln.h_bbytes.p + ln.h_boff.p[i], qbytes + qoff[lo + bad[i]], qoff[lo + bad[i] + 1] - qoff[lo + bad[i]]);
        });
        if (ln.bad_list.n < nb) ln.bad_list.alloc(nb + nb / 4 + 1024);
        if (ln.boff.n < nb + 1) ln.boff.alloc(nb + nb / 4 + 1024);
        if (ln.bbytes.n < tot + 16) ln.bbytes.alloc(tot + tot / 4 + 4096);
        HIP_CHECK(hipMemcpyAsync(ln.bad_list.p, ln.h_bq.p, nb * 4, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(ln.boff.p, ln.h_boff.p, (nb + 1) * 8, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(ln.bbytes.p, ln.h_bbytes.p, tot, hipMemcpyHostToDevice, s));
        const QueryList ql{ln.bad_list.p, nullptr, nb, nullptr, ln.bad.p + 1, 1};
        hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_COMPACT>), dim3(grid_for(r, nb, 256)), dim3(256), 0, s, r.dev, ln.bbytes.p, ln.boff.p, n,
                           ln.counts.p, ln.rstart.p, nullptr, 1, 0, ql);
        HIP_CHECK(hipGetLastError());
      }
      launch_scan(r, ln.counts.p, n, ln.hit_off.p, ln.scratch.p, s);
      // (the chunk's hit total and offsets stay on the lane stream: the host needs the total to start stage 2, and on the
      // shared copy-out stream it would queue behind the other lane's result arrays)
      HIP_CHECK(hipMemcpyAsync(ln.h_meta.p, ln.hit_off.p + n, 8, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(ln.h_meta.p + 1, ln.bad.p, 16, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(out.off + (lo - sh.lo) + 1, ln.hit_off.p + 1, n * 8, hipMemcpyDeviceToHost, s));  // chunk-relative; rebased in stage 2
      HIP_CHECK(hipEventRecord(ln.counted, s));
      ln.stage = 1;
      return;
    }
    HIP_CHECK(hipMemcpyAsync(ln.ascii.p, qbytes + qoff[lo], nbytes, hipMemcpyHostToDevice, s));
    if (plan.ragged) HIP_CHECK(hipMemcpyAsync(ln.off.p, qoff + lo, (n + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(ln.bad.p, 0, 8, s));
    HIP_CHECK(hipMemsetAsync(ln.bad.p + 1, 0xFF, 8, s));
    const uint8_t* biased = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(ln.ascii.p) - qoff[lo]);
    if (generic) {
      if (ulen) launch_count_ascii_uniform(r, ln.ascii.p, n, ulen, ln.counts.p, ln.status.p, s, ln.rstart.p);
      else launch_count_ascii(r, biased, ln.off.p, n, ln.counts.p, ln.rstart.p, ln.status.p, s, true);
      HIP_CHECK(hipMemcpyAsync(ln.h_status.p, ln.status.p, n, hipMemcpyDeviceToHost, s));
    } else {
      launch_pack_nt2(r, ln.ascii.p, plan.ragged ? ln.off.p : nullptr, qoff[lo], n, nbytes, (int)L, (int)W, ln.words.p,
                      plan.ragged ? ln.lens.p : nullptr, ln.bad.p, s, ln.bad_list.p);
      launch_count_nt2_long(r, ln.words.p, n, (int)L, ln.counts.p, ln.rstart.p, true, s, plan.ragged ? ln.lens.p : nullptr);
      // reads with other bytes (N, IUPAC codes ...) are redone where they lie by the generic kernel working through the
      // pack kernel's list: it overwrites their counts and range words (starts only, the packed kernels' layout) before
      // the scan, so the locate pass and the result arrays never know the difference
      const QueryList ql{ln.bad_list.p, nullptr, 0, ln.bad.p, ln.bad.p + 1, 1};
      hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_GLOBAL>), dim3((unsigned)r.num_cus * 2), dim3(256), 0, s, r.dev,
                         plan.ragged ? biased : ln.ascii.p, plan.ragged ? ln.off.p : nullptr, n, ln.counts.p, ln.rstart.p, nullptr, 1,
                         plan.ragged ? 0 : L, ql);
      HIP_CHECK(hipGetLastError());
    }
    launch_scan(r, ln.counts.p, n, ln.hit_off.p, ln.scratch.p, s);
    HIP_CHECK(hipMemcpyAsync(ln.h_meta.p, ln.hit_off.p + n, 8, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(ln.h_meta.p + 1, ln.bad.p, 16, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(out.off + (lo - sh.lo) + 1, ln.hit_off.p + 1, n * 8, hipMemcpyDeviceToHost, s));  // chunk-relative; rebased in stage 2
    HIP_CHECK(hipEventRecord(ln.counted, s));
    ln.stage = 1;
  };
  auto stage2 = [&](int li) {  // locate, once the chunk's hit total is known
    LocateLane& ln = lanes[li];
    if (ln.stage != 1) return;
    hipStream_t s = r.lane_stream[li];
    const uint64_t n = ln.hi - ln.lo;
    timed(t_wait_count, [&] { HIP_CHECK(hipEventSynchronize(ln.counted)); });
    if (generic) {
      uint64_t any = 0;
      for (uint64_t i = 0; i < n; i++) any |= ln.h_status.p[i];
      if (any) {
        ChunkBuffers cb;
        cb.h_status.assign(ln.h_status.p, ln.h_status.p + n);
        check_status(cb, ln.lo);  // raises INVALID_QUERY naming the first such query
      }
    }
    if (!generic && ln.h_meta.p[2] != ~0ull) {  // the lowest read of the chunk that the reference leaves undefined
      ChunkBuffers cb;
      cb.h_status.assign(1, (uint8_t)(ln.h_meta.p[2] & 0xFF));
      check_status(cb, ln.lo + (ln.h_meta.p[2] >> 8));
    }
    ln.total = ln.h_meta.p[0];
    out.rebase_offsets(n, ln.total);  // stage 2 runs in chunk order
    if (ln.total) {
      // the positions go from the device straight into the result arrays (pinned, PinnedPool): no staging, no host copy.
      // An array that has to grow first waits for the copies still on their way into it.
      timed(t_grow_host, [&] {
        out.reserve(ln.total, want_gpos, [&] {
          n_grow_host++;
          for (int l2 = 0; l2 < 2; l2++) HIP_CHECK(hipStreamSynchronize(r.lane_stream[l2]));
          HIP_CHECK(hipStreamSynchronize(r.copy_out));  // (copies on their way into the old arrays)
        });
      });
      timed(t_grow_dev, [&] {
        if (ln.gpos.n < ln.total) { ln.gpos.alloc(ln.total + ln.total / 4); n_grow_dev++; }
        if (want_pos && ln.pos.n < 2 * ln.total) { ln.pos.alloc(2 * (ln.total + ln.total / 4)); n_grow_dev++; }
      });
      launch_locate(r, ln.rstart.p, generic ? 2 : 1, ln.hit_off.p, n, ln.total, ln.gpos.p, want_pos ? ln.pos.p : nullptr, s);
      hipStream_t cout = copy_streams ? r.copy_out : s;
      if (copy_streams) {
        HIP_CHECK(hipEventRecord(ln.ev_k, s));
        HIP_CHECK(hipStreamWaitEvent(cout, ln.ev_k, 0));
      }
      if (want_pos) HIP_CHECK(hipMemcpyAsync(out.pos.p + out.total, ln.pos.p, ln.total * 16, hipMemcpyDeviceToHost, cout));
      if (want_gpos) HIP_CHECK(hipMemcpyAsync(out.gpos.p + out.total, ln.gpos.p, ln.total * 8, hipMemcpyDeviceToHost, cout));
      out.total += ln.total;
      HIP_CHECK(hipEventRecord(ln.located, cout));
    } else {
      HIP_CHECK(hipEventRecord(ln.located, s));
    }
    ln.stage = 2;
  };
  auto stage3 = [&](int li) {  // results into the output arrays, in chunk order
    LocateLane& ln = lanes[li];
    if (ln.stage != 2) return;
    timed(t_wait_locate, [&] { HIP_CHECK(hipEventSynchronize(ln.located)); });
    ln.stage = 0;
  };
  uint64_t i = 0;
  for (Shard c : chunks) {
    const int li = (int)(i & 1);
    stage3(li);              // chunk i - 2
    stage1(li, c.lo, c.hi);  // chunk i
    stage2(li ^ 1);          // chunk i - 1
    i++;
  }
  const int last = (int)((i + 1) & 1);            // lane of chunk i - 1
  stage3(last ^ 1);                               // chunk i - 2
  stage2(last);
  stage3(last);
  if (trace)
    fprintf(stderr, "[awry] packed locate shard: %llu reads, %zu hits, %.2f ms (pin %.2f, host pack %.2f, waiting for counts %.2f, for positions %.2f, growing the result arrays %.2f in %d step(s), the lanes' device hit buffers %.2f in %d; results land in the caller's arrays by DMA)\n",
            (unsigned long long)(sh.hi - sh.lo), out.total, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
            t_pin, t_pack, t_wait_count, t_wait_locate, t_grow_host, n_grow_host, t_grow_dev, n_grow_dev);
}

void locate_shard(Replica& r, const uint8_t* qbytes, const uint64_t* qoff, Shard sh, bool want_gpos, LocateResult& out) {
  HIP_CHECK(hipSetDevice(r.device));
  out.nq = sh.hi - sh.lo;
  static const bool no_fast = getenv("AWRY_HOST_PATH") && !strcmp(getenv("AWRY_HOST_PATH"), "generic");
  PackedPlan plan;
  if (!no_fast && r.dev.alphabet == NUCLEOTIDE) plan = plan_packed(qoff, sh);
  if (plan.ok || (!no_fast && sh.hi - sh.lo >= 4096))
    locate_shard_packed(r, qbytes, qoff, sh, plan, want_gpos, out);
  else
    locate_shard_generic(r, qbytes, qoff, sh, want_gpos, out);
}

// run fn(replica, shard, slot) on every replica concurrently; rethrow the first failure
template <class F>
void for_each_replica(awry_index* ix, uint64_t n, F&& fn) {
  if (ix->reps.empty()) throw NoDeviceError("no device replica: call awry_set_devices() first (there is no CPU search path)");
  auto shards = shard_queries(n, ix->reps.size());
  if (ix->reps.size() == 1) { fn(*ix->reps[0], shards[0], 0); return; }
  std::vector<std::exception_ptr> errs(ix->reps.size());
  std::vector<std::thread> pool;
  for (size_t g = 0; g < ix->reps.size(); g++)
    pool.emplace_back([&, g] {
      try { fn(*ix->reps[g], shards[g], (int)g); } catch (...) { errs[g] = std::current_exception(); }
    });
  for (auto& t : pool) t.join();
  for (auto& e : errs) if (e) std::rethrow_exception(e);
}

// small result arrays of the single-query entry points (awry_locate): plain heap memory, which awry_free_buffer tells
// apart from the pinned pool blocks of the batch paths (release_result); large ones 2 MB-aligned and advised for huge pages
template <class T>
T* malloc_array(size_t n) {
  const size_t bytes = std::max<size_t>(1, n) * sizeof(T);
  void* p = nullptr;
  if (bytes >= (8u << 20)) {
    if (posix_memalign(&p, 2u << 20, bytes) != 0) p = nullptr;
    if (p) (void)madvise(p, bytes, MADV_HUGEPAGE);
  } else p = malloc(bytes);
  if (!p) throw std::bad_alloc();
  return static_cast<T*>(p);
}

void fill_ref_kmer_table(awry_index* ix) {
  HostIndex& h = ix->host;
  const uint64_t nslots = ref_kmer_table_entries(h.alphabet, h.kmer_len);
  if (h.ref_kmer_table.size() == 2 * nslots) return;
  if (ix->reps.empty()) {  // saving is host work: no replica, no GPU needed
    fill_ref_kmer_table_host(h);
    return;
  }
  Replica& r = replica(ix, 0);
  DevBuf<uint64_t> tab(2 * nslots);
  const dim3 g(grid_for(r, nslots, 256)), b(256);
  if (h.alphabet == NUCLEOTIDE)
    hipLaunchKernelGGL(ref_kmer_table_kernel<NUCLEOTIDE>, g, b, 0, r.stream, r.dev, (int)h.kmer_len, nslots, tab.p);
  else
    hipLaunchKernelGGL(ref_kmer_table_kernel<AMINO>, g, b, 0, r.stream, r.dev, (int)h.kmer_len, nslots, tab.p);
  HIP_CHECK(hipGetLastError());
  h.ref_kmer_table.resize(2 * nslots);
  HIP_CHECK(hipMemcpyAsync(h.ref_kmer_table.data(), tab.p, 2 * nslots * 8, hipMemcpyDeviceToHost, r.stream));
  HIP_CHECK(hipStreamSynchronize(r.stream));
}

uint64_t scalar_op(awry_index* ix, int op, uint64_t a, uint64_t b, int idx, uint64_t* second = nullptr) {
  Replica& r = replica(ix, 0);
  DevBuf<uint64_t> out(2);
  if (r.dev.alphabet == NUCLEOTIDE)
    hipLaunchKernelGGL(scalar_ops_kernel<NUCLEOTIDE>, dim3(1), dim3(64), 0, r.stream, r.dev, op, a, b, idx, out.p);
  else
    hipLaunchKernelGGL(scalar_ops_kernel<AMINO>, dim3(1), dim3(64), 0, r.stream, r.dev, op, a, b, idx, out.p);
  HIP_CHECK(hipGetLastError());
  uint64_t h[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(h, out.p, 16, hipMemcpyDeviceToHost, r.stream));
  HIP_CHECK(hipStreamSynchronize(r.stream));
  if (second) *second = h[1];
  return h[0];
}

int checked_symbol(const awry_index* ix, uint8_t ascii) {
  if (ascii >= 0x80) throw QueryError("non-ASCII symbol");
  return index_of_ascii(ix->host.alphabet, ascii);
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* awry_last_error(void) { return g_last_error.c_str(); }

// build_device: >= 0 construct on that GPU (sa_builder.hip); AWRY_BUILD_HOST (-1) host SA-IS;
// AWRY_BUILD_AUTO (-2): GPU 0 when one is visible and the text is large enough to pay for it
// The index is built over the CANONICAL text: every byte replaced by the letter of its symbol index (lower case folded,
// U -> T, IUPAC codes and anything else -> N; non-standard residues -> X) -- the map queries go through
// (src/alphabet.rs:109-114,169-248).  Suffixes must be sorted in the order the BWT encodes them: sorted by raw bytes, a
// text with R / Y / K ... (or B / Z / U / O / J in proteins) would put its N- (X-) suffixes in several places while
// prefix_sums assume one block, and LF steps and text comparison would disagree.  Returns true and fills `out` when the
// text had to be rewritten.  An inner '$' / '#' is an argument error (the text model has exactly one sentinel, at the end).
static bool canonical_text(const uint8_t* text, uint64_t bwt_len, int alphabet, std::vector<uint8_t>& out) {
  uint8_t canon[256];
  for (int b = 0; b < 256; b++) canon[b] = ascii_of_index(alphabet, index_of_ascii(alphabet, (uint8_t)b));
  const uint64_t body = bwt_len - 1;
  std::atomic<int> other{0}, sentinel{0};
  HostPool::instance().run_ranges(body, 1u << 22, [&](uint64_t lo, uint64_t hi) {
    unsigned diff = 0, sent = 0;
    for (uint64_t i = lo; i < hi; i++) {  // branch-free: vectorises
      const uint8_t c = canon[text[i]];
      diff |= (unsigned)(c != text[i]);
      sent |= (unsigned)(c == '$');
    }
    if (diff) other.store(1, std::memory_order_relaxed);
    if (sent) sentinel.store(1, std::memory_order_relaxed);
  });
  if (sentinel.load()) throw ArgError("the text holds '$' or '#' before its last byte (the text model has one sentinel, at the end)");
  if (!other.load()) return false;
  out.resize(bwt_len);
  HostPool::instance().run_ranges(body, 1u << 22, [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) out[i] = canon[text[i]];
  });
  out[body] = '$';
  return true;
}

static void construct(awry_index* ix, const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio, uint8_t kmer_len,
                      const uint64_t* seq_starts, const char* const* headers, uint64_t nseq, int build_device) {
  static const uint64_t zero = 0;
  if (nseq == 0 || !seq_starts) { seq_starts = &zero; nseq = 1; headers = nullptr; }
  if (bwt_len == 0 || text[bwt_len - 1] != '$') throw ArgError("text must end with '$'");
  std::vector<uint8_t> canon;
  if (canonical_text(text, bwt_len, alphabet, canon)) text = canon.data();
  const bool automatic = build_device == AWRY_BUILD_AUTO;
  if (build_device == AWRY_BUILD_AUTO) {
    const char* e = getenv("AWRY_BUILD");
    int ndev = 0;
    if (e && !strcmp(e, "host")) build_device = AWRY_BUILD_HOST;
    else if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && bwt_len < (1ull << 32) - 1 && (bwt_len >= (1u << 20) || (e && !strcmp(e, "gpu")))) {
      // the calling thread's current device (one process per GPU sets it to its own), else GPU 0
      if (hipGetDevice(&build_device) != hipSuccess || build_device < 0 || build_device >= ndev) build_device = 0;
    } else build_device = AWRY_BUILD_HOST;
  }
  if (build_device < 0) {
    build_from_text(ix->host, text, bwt_len, alphabet, sa_ratio, kmer_len, seq_starts, headers, nseq);
    return;
  }
  prepare_build(ix->host, text, bwt_len, alphabet, sa_ratio, kmer_len, seq_starts, headers, nseq);
  try {
    gpu_build_index(ix->host, text, bwt_len, build_device, getenv("AWRY_VERBOSE") != nullptr);
  } catch (const std::bad_alloc&) {
    throw;
  } catch (const std::exception& e) {
    // Construction is host work in the reference; the GPU is how it gets fast here, not a requirement.  When the device
    // was chosen automatically and cannot do it (its HBM is taken by replicas, say: the builder needs ~45 B per symbol),
    // the host builder produces the same index, only slower.  An explicit device request fails loudly instead.
    if (!automatic) throw HipError(std::string("GPU index construction: ") + e.what());
    (void)hipGetLastError();
    if (getenv("AWRY_VERBOSE")) fprintf(stderr, "[awry] GPU index construction failed (%s): building on the host\n", e.what());
    ix->host = HostIndex();
    build_from_text(ix->host, text, bwt_len, alphabet, sa_ratio, kmer_len, seq_starts, headers, nseq);
  }
}

int awry_build_from_text_on(const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio, uint8_t kmer_len,
                            const uint64_t* seq_starts, const char* const* headers, uint64_t nseq, int build_device,
                            awry_index_t** out) {
  return guarded([&] {
    require(text && out && bwt_len > 0, "null argument");
    require(alphabet == NUCLEOTIDE || alphabet == AMINO, "bad alphabet id");
    require(build_device >= AWRY_BUILD_AUTO, "bad build device");
    auto ix = std::make_unique<awry_index>();
    construct(ix.get(), text, bwt_len, alphabet, sa_ratio, kmer_len, seq_starts, headers, nseq, build_device);
    *out = ix.release();
  });
}

int awry_build_from_text(const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio, uint8_t kmer_len,
                         const uint64_t* seq_starts, const char* const* headers, uint64_t nseq, awry_index_t** out) {
  return awry_build_from_text_on(text, bwt_len, alphabet, sa_ratio, kmer_len, seq_starts, headers, nseq, AWRY_BUILD_AUTO, out);
}

int awry_build(const awry_build_args_t* args, awry_index_t** out) {
  return guarded([&] {
    require(args && args->input_path && out, "null argument");
    require(args->alphabet <= 1, "bad alphabet id");
    SequenceFile sf = read_sequence_file(args->input_path, args->alphabet);
    std::vector<const char*> hdr;
    for (auto& h : sf.headers) hdr.push_back(h.c_str());
    auto ix = std::make_unique<awry_index>();
    construct(ix.get(), sf.text.data(), sf.text.size(), args->alphabet, args->sa_ratio, args->kmer_len, sf.starts.data(),
              hdr.data(), sf.starts.size(), AWRY_BUILD_AUTO);
    *out = ix.release();
  });
}

int awry_load(const char* path, awry_index_t** out) {
  return guarded([&] {
    require(path && out, "null argument");
    auto ix = std::make_unique<awry_index>();
    load_awry(ix->host, path);
    *out = ix.release();
  });
}

int awry_save(awry_index_t* idx, const char* path) {
  return guarded([&] {
    require(idx && path, "null argument");
    fill_ref_kmer_table(idx);
    save_awry(idx->host, path);
  });
}

void awry_free(awry_index_t* idx) { delete idx; }

namespace {
// The first batch call of a process used to pay for what every later one finds in place: the lanes' pinned staging and device
// buffers, their events, the worker pool's threads and the pinned result arrays of the locate path (4 M 101-bp reads: 30 ms
// for the first awry_locate_batch, 6 ms from the third on).  awry_set_devices sets all of it up for the chunk sizes the
// host paths use (2^20 queries of up to 128 letters), so that call #1 costs what call #3 does.  AWRY_PREWARM=0 skips it.
void prewarm_host_paths(Replica& r) {
  static const bool off = getenv("AWRY_PREWARM") && !strcmp(getenv("AWRY_PREWARM"), "0");
  if (off) return;
  HIP_CHECK(hipSetDevice(r.device));
  (void)HostPool::instance();
  const bool nt = r.dev.alphabet == NUCLEOTIDE;
  const uint64_t cap = 1u << 20, W = 4;  // one chunk of the host paths; W words per query cover reads of up to 128 letters
  std::unique_lock<std::mutex> lane_lock(r.lane_mu);
  size_t free_b = 0, total_b = 0;
  const bool hbm_plenty = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > (32ull << 30);
  if (nt) {
    for (int li = 0; li < Replica::NLANES; li++) {  // count_shard_hostpacked
      PackedLane& ln = r.lanes[li];
      ln.s = r.lane_stream[li];
      if (!ln.done) HIP_CHECK(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
      if (!ln.h_bad) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&ln.h_bad), 16, hipHostMallocDefault));
      ln.h_words.ensure(cap * W);
      ln.h_counts32.ensure(cap);
      if (ln.words.n < cap * W) ln.words.alloc(cap * W);
      if (ln.counts.n < cap) ln.counts.alloc(cap);
      if (ln.counts32.n < cap) ln.counts32.alloc(cap);
      if (ln.bad.n < 2) ln.bad.alloc(2);
    }
    for (int li = 0; li < 2; li++) {  // locate_shard_packed
      LocateLane& ln = r.loc_lanes[li];
      if (!ln.counted) HIP_CHECK(hipEventCreateWithFlags(&ln.counted, hipEventDisableTiming));
      if (!ln.located) HIP_CHECK(hipEventCreateWithFlags(&ln.located, hipEventDisableTiming));
      if (ln.words.n < cap * W) ln.words.alloc(cap * W);
      ln.h_words.ensure(cap * W);
      if (ln.rstart.n < cap) ln.rstart.alloc(cap);
      if (ln.counts.n < cap) ln.counts.alloc(cap);
      if (ln.hit_off.n < cap + 1) ln.hit_off.alloc(cap + 1);
      if (ln.scratch.n < scan_tiles(cap) + 1) ln.scratch.alloc(scan_tiles(cap) + 1);
      if (ln.bad.n < 2) ln.bad.alloc(2);
      ln.h_meta.ensure(3);
      // hit buffers of a chunk: reads from repeat-rich genomes bring several hits each (7.6 on the GRCh38-shaped text), and a
      // lane whose buffer is too small frees and re-allocates it in the middle of the first call (3-4 ms, four times): room for
      // 16 hits per read (400 MB per lane) while that is a small part of the free HBM, 1.25 otherwise
      static const bool big_hits = !(getenv("AWRY_PREWARM_HITS") && !strcmp(getenv("AWRY_PREWARM_HITS"), "0"));
      const uint64_t hits_cap = hbm_plenty && big_hits ? 16 * cap : cap + cap / 4;
      if (ln.gpos.n < hits_cap) ln.gpos.alloc(hits_cap);
      if (ln.pos.n < 2 * hits_cap) ln.pos.alloc(2 * hits_cap);
    }
    // scratch of the two-phase schedules on the lane streams (survivor lists of a full chunk)
    for (int li = 0; li < Replica::NLANES; li++) {
      Replica::SurvScratch* sc = surv_scratch(r, r.lane_stream[li]);
      const unsigned nblk = (unsigned)r.num_cus * 8;
      const uint64_t per_block = ((cap + (uint64_t)nblk * 256 - 1) / ((uint64_t)nblk * 256)) * 256, total = per_block * nblk;
      if (sc->cap < total) { sc->w.alloc(total); sc->range.alloc(total); sc->q.alloc(total); sc->cap = sc->cap_q = total; }
      if (!sc->count.p) sc->count.alloc(nblk);
      if (!sc->counters.p) sc->counters.alloc(8);
    }
  }
  // pinned result arrays of the locate path (offsets, positions, (record, offset) pairs), taken from the process-wide pool and
  // handed back so that the first call finds them cached.  Pinning is what a first call with large results paid for: 55 ms
  // of a 93 ms awry_locate_batch that returned 735 MB (4 M reads, 30.6 M hits, GRCh38-shaped text) went into ONE growth step
  // of the result arrays, i.e. hipHostMalloc at ~13 GB/s.  AWRY_PINNED_PREWARM_MB (default 1024, capped by the pool's
  // AWRY_PINNED_CACHE_GB) is pinned here instead, as blocks of 64, 64, 128, 256 and 512 MB -- the sizes the arrays of
  // results up to ~750 MB round to; a first call with more than that still pins the excess itself, once.
  static std::once_flag once;
  void* warm_block = nullptr;  // one block stays out until the locate warm-up below has copied into it
  std::call_once(once, [&] {
    const char* e = getenv("AWRY_PINNED_PREWARM_MB");
    const size_t budget = (size_t)((e && atof(e) >= 0 ? atof(e) : 1024.0) * (double)(1u << 20));
    const size_t sizes[5] = {64u << 20, 64u << 20, 128u << 20, 256u << 20, 512u << 20};
    void* blocks[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t used = 0;
    for (int i = 0; i < 5 && used + sizes[i] <= budget; i++) { blocks[i] = PinnedPool::instance().get(sizes[i]); used += sizes[i]; }
    warm_block = blocks[0];
    for (int i = 1; i < 5; i++)
      if (blocks[i]) release_result(blocks[i]);
  });
  // the locate path's own kernels (reads probe with range words, the generic pass over the listed reads, scan, tile/walk/
  // localise) and its copies into pool memory, once per locate lane: 8-9 ms of a first awry_locate_batch were first uses
  // AWRY_PREWARM_LOCATE: bit 0 the reads probe + listed pass, bit 1 scan + locate pass, bit 2 the chunk-sized copies into pool
  // memory (default 7; 0 = none) -- for tools/first_call_ab.sh
  static const int lmask = getenv("AWRY_PREWARM_LOCATE") ? atoi(getenv("AWRY_PREWARM_LOCATE")) : 7;
  const bool warm_locate = lmask != 0;
  if (nt && warm_locate && r.dev.bwt_len >= 4)
    for (int li = 0; li < 2; li++) {
      LocateLane& ln = r.loc_lanes[li];
      hipStream_t s = r.lane_stream[li];
      if (lmask & 1) {
        memset(ln.h_words.p, 0, 16 * W * 8);  // 16 reads of 101 A's: whatever they find, the kernels have run
        HIP_CHECK(hipMemcpyAsync(ln.words.p, ln.h_words.p, 16 * W * 8, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(ln.bad.p, 0, 8, s));
        HIP_CHECK(hipMemsetAsync(ln.bad.p + 1, 0xFF, 8, s));
        launch_count_nt2_long(r, ln.words.p, 16, 101, ln.counts.p, ln.rstart.p, true, s, nullptr);
        const QueryList ql{nullptr, nullptr, 0, ln.bad.p, ln.bad.p + 1, 1};  // an empty list: the launch itself is what is warmed
        hipLaunchKernelGGL((count_scalar_kernel<NUCLEOTIDE, LIST_GLOBAL>), dim3(1), dim3(256), 0, s, r.dev, (const uint8_t*)nullptr, (const uint64_t*)nullptr,
                           (uint64_t)0, ln.counts.p, ln.rstart.p, nullptr, 1, (uint64_t)101, ql);
        HIP_CHECK(hipGetLastError());
      }
      if (lmask & 2) {
        // the locate pass on one range that is valid in every index: one hit, the row in the middle of the BWT (RS_PLAIN)
        ln.h_meta.p[0] = r.dev.bwt_len / 2;
        ln.h_meta.p[1] = 1;
        HIP_CHECK(hipMemcpyAsync(ln.rstart.p, ln.h_meta.p, 8, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(ln.counts.p, ln.h_meta.p + 1, 8, hipMemcpyHostToDevice, s));
        launch_scan(r, ln.counts.p, 1, ln.hit_off.p, ln.scratch.p, s);
        launch_locate(r, ln.rstart.p, 1, ln.hit_off.p, 1, 1, ln.gpos.p, ln.pos.p, s);
        HIP_CHECK(hipMemcpyAsync(ln.h_meta.p, ln.hit_off.p + 1, 8, hipMemcpyDeviceToHost, s));
      }
      if (warm_block && (lmask & 4)) {  // chunk-sized copies into pool memory, as the call's results take them
        HIP_CHECK(hipMemcpyAsync(warm_block, ln.gpos.p, cap * 8, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(static_cast<char*>(warm_block) + cap * 8, ln.pos.p, cap * 16, hipMemcpyDeviceToHost, s));
      }
      HIP_CHECK(hipEventRecord(ln.located, s));
      HIP_CHECK(hipEventSynchronize(ln.located));
      HIP_CHECK(hipStreamSynchronize(s));
    }
  if (warm_block) release_result(warm_block);
  // one round trip per lane stream -- a chunk-sized copy in, the count kernels, a chunk-sized copy out.  Measured: without it the
  // first awry_count_batch of a process spent 17.7 ms enqueueing its first chunks, with a round trip of small copies 7 ms, with
  // chunk-sized ones 0.2 ms (what exactly the runtime sets up on a stream's first use I have not looked at)
  if (nt)
    for (int li = 0; li < Replica::NLANES; li++) {
      PackedLane& ln = r.lanes[li];
      hipStream_t s = r.lane_stream[li];
      memset(ln.h_words.p, 0, cap * 8);  // (a whole chunk each way: large pinned copies take the copy engines, small ones do not)
      const auto c0 = std::chrono::steady_clock::now();
      HIP_CHECK(hipMemcpyAsync(ln.words.p, ln.h_words.p, cap * 8, hipMemcpyHostToDevice, s));
      if (getenv("AWRY_TRACE_HOST"))
        fprintf(stderr, "[awry] warm-up, count lane %d: enqueue of the chunk-sized copy in took %.2f ms\n", li,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count());
      launch_count_nt2(r, ln.words.p, 64, r.seed_k > 0 && r.seed_k < 31 ? r.seed_k + 1 : 31, ln.counts.p, true, s, nullptr);
      launch_count_nt2_long(r, ln.words.p, 16, 101, ln.counts.p, nullptr, true, s, nullptr);
      hipLaunchKernelGGL(narrow_counts_kernel, dim3(grid_for(r, cap, 1024)), dim3(256), 0, s, ln.counts.p, ln.counts32.p, cap);
      HIP_CHECK(hipMemcpyAsync(ln.h_counts32.p, ln.counts32.p, cap * 4, hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipEventRecord(ln.done, s));
      HIP_CHECK(hipEventSynchronize(ln.done));
      HIP_CHECK(hipStreamSynchronize(s));
    }
  lane_lock.unlock();
  // Last: three chunks of synthetic packed 31-mers through the REAL pipelined count path.  Measured, not explained
  // (tools/first_call_ab.sh, fresh processes on one box, profiles/r03H_first_call_ab.txt): once the warm-up above had run the
  // reads probe with range words (the part that takes 8 ms off the first awry_locate_batch), the first awry_count_batch of the
  // process blocked 11-34 ms inside its first host-to-device copies -- 16 of 19 processes -- although every single operation
  // of that call had been issued here before, and although chunk-sized copies issued here one at a time, in any order and
  // number, returned in 0.01 ms and absorbed nothing.  The stall is paid once, by whichever pipelined call comes first, and
  // never again (count after locate after count: steady).  So the first pipelined call is made here: first awry_count_batch
  // 2.0-2.5 ms against 1.8-2.25 steady in 8 of 8 processes (r03H setting K, r03J setting L), 2.0-4.0 ms in 6 of 6 (r03M); after
  // the chunk copies moved to the replica's copy streams 1.8-2.1 ms in 4 of 5 and 14.9 ms in one (r03d1): it still gets
  // through now and then.  AWRY_PREWARM_REALCOUNT=0 leaves it out (for the A/B).  Tried and dropped: a real-shaped awry_locate_batch of reads without hits as well -- after the count
  // call it left 1 of 5 first count calls at 15 ms again, before it (once or twice) 10 of 16 first locate calls at 13-17 ms
  // where this arrangement gives 7.9-8.6 (profiles/r03J..r03L_first_call_ab.txt).  The first awry_locate_batch therefore
  // still costs ~2.5 ms more than the ones after it (5.5-6.1 ms).
  if (nt && !r.wide && !(getenv("AWRY_PREWARM_REALCOUNT") && !strcmp(getenv("AWRY_PREWARM_REALCOUNT"), "0"))) {
    const uint64_t n = 3ull << 20;
    std::vector<uint64_t> words(n, 0), counts(n, 0);
    PackedPlan plan;
    plan.ok = true;
    plan.Lmax = 31;
    count_shard_hostpacked(r, nullptr, nullptr, Shard{0, n}, plan, counts.data(), words.data());
  }
}
}  // namespace

int awry_set_devices(awry_index_t* idx, const int* device_ids, int n_devices) {
  return guarded([&] {
    require(idx != nullptr, "null index");
    if (n_devices <= 0 || !device_ids) throw NoDeviceError("awry_set_devices needs at least one GPU: there is no CPU search path");
    // the old replicas go first: the policies below size the seed table and the accelerators from the HBM that is free,
    // and a rebuilt replica on the same GPU must not see half of it (if the build fails the index is left without replicas)
    idx->reps.clear();
    std::vector<std::unique_ptr<Replica>> reps(n_devices);
    if (n_devices == 1) {
      reps[0] = make_replica(idx, device_ids[0]);
    } else {  // one host thread per GPU: upload + seed-table build run concurrently on all of them
      std::vector<std::exception_ptr> errs(n_devices);
      std::vector<std::thread> pool;
      for (int i = 0; i < n_devices; i++)
        pool.emplace_back([&, i] {
          try { reps[i] = make_replica(idx, device_ids[i]); } catch (...) { errs[i] = std::current_exception(); }
        });
      for (auto& t : pool) t.join();
      for (auto& e : errs) if (e) std::rethrow_exception(e);
    }
    idx->reps = std::move(reps);
    for (auto& r : idx->reps) prewarm_host_paths(*r);
  });
}

int awry_set_seed_kmer_len(awry_index_t* idx, int k) {
  return guarded([&] {
    require(idx != nullptr, "null index");
    require(k >= -1 && k <= 17, "seed k-mer length must be in -1..17 (amino: ..7)");
    idx->seed_k_request = k;
    for (size_t s = 0; s < idx->reps.size(); s++) {
      Replica& r = replica(idx, (int)s);
      build_seed(idx, r, k < 0 ? default_seed_k(idx->host) : k);
      sync_seed_mode(idx, r);
    }
  });
}

int awry_debug_set_count_kernel(int mode) { g_count_kernel.store(mode); return AWRY_OK; }
int awry_debug_force_wide_rows(int on) { g_force_wide.store(on ? 1 : 0); return AWRY_OK; }

const char* awry_count_schedule(const awry_index_t* idx, int L) {
  static const char* names[] = {"count_nt2_quad_kernel", "count_nt2_chunk_kernel", "count_nt2_quad4_kernel",
                                "count_nt2_probe_kernel+count_nt2_resume_kernel"};
  if (!idx || idx->reps.empty()) return "";
  const Replica& r = *idx->reps[0];
  if (r.wide) return count_kernel_override() == 3 && r.seed_k > 0 && r.seed_k <= L ? "count_nt2_wide_probe_kernel+count_nt2_wide_kernel" : "count_nt2_wide_kernel";
  const bool seeded = r.seed_k > 0 && r.seed_k <= L;
  if (L > 32) {  // launch_count_nt2_long
    const int om = count_kernel_override();
    const bool two = r.dev.text4 && r.dev.dense_ratio == 1 && seeded && L - r.seed_k >= 3 && L <= 512 && (om < 0 || om == 3);
    return two ? "count_nt2_reads_probe_kernel+count_nt2_reads_kernel" : "count_nt2_reads_kernel";
  }
  static const bool rungs_off = getenv("AWRY_SEED_RUNGS") && !strcmp(getenv("AWRY_SEED_RUNGS"), "0");
  if (!seeded && r.seed_k > L && L >= SEED_RUNG_MIN && !rungs_off && count_kernel_override() < 0)
    return "count_nt2_probe_kernel+count_nt2_resume_kernel (table of its own for this length; batches of 4096 queries and more)";
  int m = count_kernel_mode(r.dev.bwt_len, r.seed_k, seeded);
  if (m == 3 && !seeded) m = 2;
  return names[m & 3];
}

int awry_seed_kmer_len(const awry_index_t* idx) { return idx && !idx->reps.empty() ? idx->reps[0]->seed_k : 0; }
int awry_num_devices(const awry_index_t* idx) { return idx ? (int)idx->reps.size() : 0; }
int awry_replica_device(const awry_index_t* idx, int slot) {
  return idx && slot >= 0 && slot < (int)idx->reps.size() ? idx->reps[slot]->device : -1;
}

int awry_count_batch(awry_index_t* idx, const uint8_t* qbytes, const uint64_t* qoff, uint64_t n, uint64_t* counts_out) {
  return guarded([&] {
    require(idx && qoff && (counts_out || n == 0), "null argument");
    require(qbytes || qoff[n] == qoff[0], "null query bytes");
    const auto t0 = std::chrono::steady_clock::now();
    for_each_replica(idx, n, [&](Replica& r, Shard sh, int) { count_shard(r, qbytes, qoff, sh, counts_out); });
    if (getenv("AWRY_TRACE_HOST"))
      fprintf(stderr, "[awry] awry_count_batch %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  });
}

// k-mers that the caller already holds packed (letter j of a k-mer in bits [2j, 2j + 2) of its word, A0 C1 G2 T3): no
// ASCII crosses PCIe, 16 B per query both ways instead of L + 8.
int awry_count_packed_kmers(awry_index_t* idx, const uint64_t* words, uint64_t n, int L, uint64_t* counts_out) {
  return guarded([&] {
    require(idx && ((words && counts_out) || n == 0), "null argument");
    require(L >= 1 && L <= 32, "packed k-mer length must be in 1..32");
    for_each_replica(idx, n, [&](Replica& r, Shard sh, int) {
      HIP_CHECK(hipSetDevice(r.device));
      require(r.dev.alphabet == NUCLEOTIDE, "packed k-mers need a nucleotide index");
      if (sh.hi <= sh.lo) return;
      PackedPlan plan;
      plan.ok = true;
      plan.Lmax = (uint64_t)L;
      count_shard_hostpacked(r, nullptr, nullptr, sh, plan, counts_out, words);  // staged through the lanes' pinned buffers
    });
  });
}

int awry_locate_batch(awry_index_t* idx, const uint8_t* qbytes, const uint64_t* qoff, uint64_t n, uint64_t** hit_off_out,
                      awry_pos_t** hits_out, uint64_t** global_pos_out) {
  return guarded([&] {
    require(idx && qoff && hit_off_out, "null argument");
    require(qbytes || qoff[n] == qoff[0], "null query bytes");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<LocateResult> res(std::max<size_t>(1, idx->reps.size()));
    for (auto& x : res) x.want_pos = hits_out != nullptr;
    MBuf<uint64_t> off;  // pinned (PinnedPool): the shards' device scans are copied straight into it
    off.grow(n + 1);
    off.p[0] = 0;
    {
      auto shards = shard_queries(n, res.size());  // the same cut for_each_replica makes
      for (size_t g = 0; g < res.size(); g++) res[g].off = off.p + shards[g].lo;
    }
    for_each_replica(idx, n, [&](Replica& r, Shard sh, int g) { locate_shard(r, qbytes, qoff, sh, global_pos_out != nullptr, res[g]); });
    uint64_t total = 0;
    for (size_t g = 0; g < res.size(); g++) {  // shards are contiguous in query order: rebase every shard after the first
      if (g && total)
        for (uint64_t i = 1; i <= res[g].nq; i++) res[g].off[i] += total;
      total += res[g].total;
    }
    if (res.size() == 1 && (!hits_out || res[0].pos.p) && (!global_pos_out || res[0].gpos.p)) {  // one replica: its arrays are the result
      if (hits_out) *hits_out = res[0].pos.release();
      if (global_pos_out) *global_pos_out = res[0].gpos.release();
    } else {
      MBuf<awry_pos_t> hits;
      MBuf<uint64_t> gp;
      if (hits_out) hits.grow(std::max<uint64_t>(1, total));
      if (global_pos_out) gp.grow(std::max<uint64_t>(1, total));
      uint64_t at = 0;
      for (auto& x : res) {
        if (hits_out && x.total) pool_memcpy(hits.p + at, x.pos.p, x.total * sizeof(awry_pos_t));
        if (global_pos_out && x.total) pool_memcpy(gp.p + at, x.gpos.p, x.total * 8);
        at += x.total;
      }
      if (hits_out) *hits_out = hits.release();
      if (global_pos_out) *global_pos_out = gp.release();
    }
    *hit_off_out = off.release();
    if (getenv("AWRY_TRACE_HOST"))
      fprintf(stderr, "[awry] awry_locate_batch %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  });
}

void awry_free_buffer(void* p) { release_result(p); }

namespace {
// one query through the replica's pinned mailbox; want_rows: the range must be a row interval (no text shortcut)
// (the caller holds r.mailbox_mu)
void single_query(Replica& r, const uint8_t* q, uint64_t len, bool want_rows, uint64_t& count, uint64_t& start, uint64_t& end, int ref_kmer_len = -1) {
  Replica::Mailbox& m = r.mailbox;
  if (!m.q) {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m.q), Replica::Mailbox::QCAP + 16, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m.words), 8 * 8, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m.gpos), Replica::Mailbox::HCAP * 8, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m.pos), Replica::Mailbox::HCAP * 16, hipHostMallocDefault));
  }
  if (len) memcpy(m.q, q, len);
  m.words[0] = 0;
  m.words[1] = len;
  uint8_t* status = reinterpret_cast<uint8_t*>(m.words + 5);
  launch_count_ascii(r, m.q, m.words, 1, m.words + 2, m.words + 3, status, r.stream, !want_rows, 0, want_rows ? ref_kmer_len : -1);
  HIP_CHECK(hipStreamSynchronize(r.stream));
  if (*status != Q_OK) {
    ChunkBuffers cb;
    cb.h_status.assign(1, *status);
    check_status(cb, 0);  // raises INVALID_QUERY with the usual message
  }
  count = m.words[2];
  start = m.words[3];
  end = m.words[4];
}
}  // namespace

int awry_count(awry_index_t* idx, const uint8_t* q, uint64_t len, uint64_t* count) {
  if (idx && count && idx->reps.size() == 1 && (q || len == 0) && len <= Replica::Mailbox::QCAP)
    return guarded([&] {
      uint64_t a = 0, b = 0;
      Replica& r = replica(idx, 0);
      std::lock_guard<std::mutex> lock(r.mailbox_mu);
      single_query(r, q, len, false, *count, a, b);
    });
  const uint64_t off[2] = {0, len};
  return awry_count_batch(idx, q, off, 1, count);
}

int awry_search_range(awry_index_t* idx, const uint8_t* q, uint64_t len, awry_range_t* out) {
  return guarded([&] {
    require(idx && out && (q || len == 0), "null argument");
    Replica& r = replica(idx, 0);
    if (len <= Replica::Mailbox::QCAP) {
      uint64_t c = 0;
      std::lock_guard<std::mutex> lock(r.mailbox_mu);
      single_query(r, q, len, true, c, out->start_ptr, out->end_ptr, (int)idx->host.kmer_len);
      return;
    }
    ChunkBuffers cb;
    const uint64_t off[2] = {0, len};
    run_count_chunk(r, cb, q, off, Shard{0, 1}, true, false, (int)idx->host.kmer_len);  // the caller wants rows: the reference's schedule
    uint64_t h[2];
    HIP_CHECK(hipMemcpyAsync(h, cb.ranges.p, 16, hipMemcpyDeviceToHost, r.stream));
    HIP_CHECK(hipStreamSynchronize(r.stream));
    check_status(cb, 0);
    out->start_ptr = h[0];
    out->end_ptr = h[1];
  });
}

int awry_locate(awry_index_t* idx, const uint8_t* q, uint64_t len, awry_pos_t** hits_out, uint64_t** global_pos_out,
                uint64_t* n_hits) {
  if (idx && hits_out && idx->reps.size() == 1 && (q || len == 0) && len <= Replica::Mailbox::QCAP) {
    // one query: count through the mailbox, then -- for a hit list that fits it -- locate straight into pinned memory
    bool done = false;
    int rc = guarded([&] {
      Replica& r = replica(idx, 0);
      std::lock_guard<std::mutex> lock(r.mailbox_mu);
      uint64_t count = 0, rs = 0, unused = 0;
      single_query(r, q, len, false, count, rs, unused);
      if (count > Replica::Mailbox::HCAP) return;  // the batch path sizes its own buffers
      Replica::Mailbox& m = r.mailbox;
      std::unique_ptr<awry_pos_t, decltype(&free)> hits(malloc_array<awry_pos_t>(count), &free);
      std::unique_ptr<uint64_t, decltype(&free)> gp(global_pos_out ? malloc_array<uint64_t>(count) : nullptr, &free);
      if (count) {
        m.words[6] = 0;
        m.words[7] = count;
        launch_locate(r, m.words + 3, 2, m.words + 6, 1, count, m.gpos, m.pos, r.stream);
        HIP_CHECK(hipStreamSynchronize(r.stream));
        memcpy(hits.get(), m.pos, count * sizeof(awry_pos_t));
        if (gp) memcpy(gp.get(), m.gpos, count * 8);
      }
      *hits_out = hits.release();
      if (global_pos_out) *global_pos_out = gp.release();
      if (n_hits) *n_hits = count;
      done = true;
    });
    if (rc != AWRY_OK || done) return rc;
  }
  const uint64_t off[2] = {0, len};
  uint64_t* hit_off = nullptr;
  int rc = awry_locate_batch(idx, q, off, 1, &hit_off, hits_out, global_pos_out);
  if (rc == AWRY_OK) {
    if (n_hits) *n_hits = hit_off[1];
    release_result(hit_off);
  }
  return rc;
}

int awry_initial_range(const awry_index_t* idx, uint8_t symbol_ascii, awry_range_t* out) {
  return guarded([&] {
    require(idx && out, "null argument");
    int s = checked_symbol(idx, symbol_ascii);
    out->start_ptr = idx->host.prefix_sums[s];           // src/search.rs:43-48
    out->end_ptr = idx->host.prefix_sums[s + 1] - 1;
  });
}

int awry_update_range(awry_index_t* idx, awry_range_t in, uint8_t symbol_ascii, awry_range_t* out) {
  return guarded([&] {
    require(idx && out, "null argument");
    int s = checked_symbol(idx, symbol_ascii);
    if (s == 0) throw QueryError("cannot extend a range with the sentinel (src/bwt.rs:126-128 panics)");
    if (in.start_ptr == 0 || in.start_ptr > idx->host.bwt_len || in.end_ptr >= idx->host.bwt_len)
      throw ArgError("range outside the BWT");
    out->start_ptr = scalar_op(idx, 0, in.start_ptr, in.end_ptr, s, &out->end_ptr);
  });
}

int awry_backstep(awry_index_t* idx, uint64_t row, uint64_t* out) {
  return guarded([&] {
    require(idx && out, "null argument");
    require(row < idx->host.bwt_len, "row outside the BWT");
    *out = scalar_op(idx, 1, row, 0, 0);
  });
}

int awry_get_seq_location(const awry_index_t* idx, uint64_t g, awry_pos_t* out) {
  return guarded([&] {
    require(idx && out, "null argument");
    const auto& st = idx->host.seq_starts;
    require(!st.empty(), "index has no sequence records");
    size_t i = (size_t)(std::upper_bound(st.begin(), st.end(), g) - st.begin());
    i = i ? i - 1 : 0;
    out->seq_idx = i;
    out->local_pos = g - st[i];
  });
}

int awry_alphabet(const awry_index_t* idx) { return idx ? idx->host.alphabet : -1; }
uint64_t awry_bwt_len(const awry_index_t* idx) { return idx ? idx->host.bwt_len : 0; }
uint64_t awry_version(const awry_index_t* idx) { return idx ? idx->host.version : 0; }
uint64_t awry_sa_ratio(const awry_index_t* idx) { return idx ? idx->host.sa_ratio : 0; }
uint8_t awry_kmer_len(const awry_index_t* idx) { return idx ? idx->host.kmer_len : 0; }
uint64_t awry_sentinel_row(const awry_index_t* idx) { return idx ? idx->host.sentinel_row : 0; }
const uint64_t* awry_prefix_sums(const awry_index_t* idx, uint64_t* len) {
  if (!idx) return nullptr;
  if (len) *len = idx->host.prefix_sums.size();
  return idx->host.prefix_sums.data();
}
uint64_t awry_num_sequences(const awry_index_t* idx) { return idx ? idx->host.seq_starts.size() : 0; }
uint64_t awry_sequence_start(const awry_index_t* idx, uint64_t i) {
  return idx && i < idx->host.seq_starts.size() ? idx->host.seq_starts[i] : 0;
}
const char* awry_sequence_header(const awry_index_t* idx, uint64_t i) {
  return idx && i < idx->host.headers.size() ? idx->host.headers[i].c_str() : nullptr;
}
const uint64_t* awry_block_words(const awry_index_t* idx, uint64_t* nwords) {
  if (!idx) return nullptr;
  if (nwords) *nwords = idx->host.blocks.size();
  return idx->host.blocks.data();
}
const uint64_t* awry_sa_words(const awry_index_t* idx, uint64_t* nwords) {
  if (!idx) return nullptr;
  if (nwords) *nwords = idx->host.sa_words.size();
  return idx->host.sa_words.data();
}
int awry_block_reference_layout(const awry_index_t* idx, uint64_t block, uint64_t* out, uint64_t out_words) {
  return guarded([&] {
    require(idx && out, "null argument");
    require(block < idx->host.nblocks, "block out of range");
    const uint64_t need = 4 * num_planes(idx->host.alphabet) + (idx->host.alphabet == NUCLEOTIDE ? 8 : 24);
    require(out_words >= need, "output buffer too small");
    block_to_reference(idx->host, block, out);
  });
}

int awry_read_query_file(const char* path, uint8_t** qbytes_out, uint64_t** qoff_out, uint64_t* n_out) {
  return guarded([&] {
    require(path && qbytes_out && qoff_out && n_out, "null argument");
    read_query_file(path, qbytes_out, qoff_out, n_out);
  });
}

int awry_host_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa_out) {
  return guarded([&] {
    require(text && sa_out, "null argument");
    suffix_array_bytes(text, n, sa_out);
  });
}
uint8_t awry_symbol_index(int alphabet, uint8_t ascii) { return (uint8_t)index_of_ascii(alphabet, ascii); }

int awry_host_pack_nt2(const uint8_t* qbytes, const uint64_t* qoff, uint64_t n, uint64_t L, uint64_t* words_out, uint32_t* lens_out,
                       uint32_t* bad_out, uint64_t* nbad_out) {
  return guarded([&] {
    require((qbytes && words_out && nbad_out) || n == 0, "null argument");
    require(L >= 1 && L <= 4096, "packed query length out of range");
    if (nbad_out) *nbad_out = 0;
    if (n == 0) return;
    if (qoff)
      for (uint64_t i = 0; i < n; i++) require(qoff[i + 1] >= qoff[i] && qoff[i + 1] - qoff[i] >= 1 && qoff[i + 1] - qoff[i] <= L, "query length outside 1..L");
    std::vector<uint32_t> bad;
    const uint8_t* first = qbytes + (qoff ? qoff[0] : 0);
    pack_nt2_host(first, qbytes + (qoff ? qoff[n] : n * L), qoff, 0, n, L, words_out, qoff ? lens_out : nullptr, bad);
    *nbad_out = bad.size();
    if (bad_out) std::copy(bad.begin(), bad.end(), bad_out);
  });
}
int awry_host_threads(void) { return (int)HostPool::instance().threads(); }
void awry_host_memcpy(void* dst, const void* src, uint64_t bytes) { pool_memcpy(dst, src, bytes); }

// ---- device-resident API -----------------------------------------------------------------------------

int awry_dev_pack_nt2(awry_index_t* idx, int slot, const void* d_ascii, uint64_t n, int L, void* d_words, void* d_bad, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require(L >= 1 && L <= (1 << 20), "packed read length out of range");
    require(d_ascii && d_words && d_bad, "null device pointer");
    if (n == 0) return;
    launch_pack_nt2(r, (const uint8_t*)d_ascii, nullptr, 0, n, n * (uint64_t)L, L, (L + 31) / 32, (uint64_t*)d_words, nullptr,
                    (unsigned long long*)d_bad, (hipStream_t)stream);
  });
}

int awry_dev_count_nt2(awry_index_t* idx, int slot, const void* d_words, uint64_t n, int L, void* d_counts, int use_seed, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_words && d_counts) || n == 0, "null device pointer");
    launch_count_nt2(r, (const uint64_t*)d_words, n, L, (uint64_t*)d_counts, use_seed != 0, (hipStream_t)stream);
  });
}

int awry_dev_count_nt2_tally(awry_index_t* idx, int slot, const void* d_words, uint64_t n, int L, void* d_counts, int use_seed,
                             void* d_tally, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_words && d_counts && d_tally) || n == 0, "null device pointer");
    launch_count_nt2(r, (const uint64_t*)d_words, n, L, (uint64_t*)d_counts, use_seed != 0, (hipStream_t)stream,
                     (unsigned long long*)d_tally);
  });
}

int awry_dev_count_nt2_long(awry_index_t* idx, int slot, const void* d_words, uint64_t n, int L, void* d_counts, void* d_range_start,
                            int use_seed, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_words && d_counts) || n == 0, "null device pointer");
    launch_count_nt2_long(r, (const uint64_t*)d_words, n, L, (uint64_t*)d_counts, (uint64_t*)d_range_start, use_seed != 0, (hipStream_t)stream);
  });
}

int awry_set_locate_sa_ratio(awry_index_t* idx, int ratio) {
  return guarded([&] {
    require(idx != nullptr, "null index");
    require(ratio >= 0 && ratio <= 1024, "dense SA ratio must be in 0..1024");
    idx->dense_ratio_request = ratio;
    for (size_t s = 0; s < idx->reps.size(); s++) {
      build_dense_sa(idx, replica(idx, (int)s), ratio);
      refresh_nblock(idx, replica(idx, (int)s));
      sync_seed_mode(idx, replica(idx, (int)s));
    }
  });
}
int awry_set_verify(awry_index_t* idx, int after_steps) {
  return guarded([&] {
    require(idx != nullptr, "null index");
    require(after_steps >= -1 && after_steps <= 1000, "verify threshold out of range");
    idx->verify_request = after_steps;
    if (after_steps >= 0) idx->dense_ratio_request = 1;
    for (size_t s = 0; s < idx->reps.size(); s++) {
      build_verify(idx, replica(idx, (int)s), after_steps);
      refresh_nblock(idx, replica(idx, (int)s));
      sync_seed_mode(idx, replica(idx, (int)s));
    }
  });
}
int awry_set_verify_kmers(awry_index_t* idx, int on) {
  return guarded([&] {
    require(idx != nullptr, "null index");
    idx->verify_kmers_request = on != 0;
    for (auto& r : idx->reps) r->verify_kmers = on != 0;
  });
}
int awry_set_lcx(awry_index_t* idx, int on) {
  return guarded([&] {
    require(idx != nullptr, "null index");
    idx->lcx_request = on ? -1 : 0;
    for (size_t s = 0; s < idx->reps.size(); s++) sync_seed_mode(idx, replica(idx, (int)s));
  });
}
int awry_lcx_enabled(const awry_index_t* idx) { return idx && !idx->reps.empty() && idx->reps[0]->dev.lcx_key != nullptr; }
int awry_debug_lcx(const awry_index_t* idx, int slot, const void** d_keys, const void** d_rowpos) {
  if (!idx || slot < 0 || slot >= (int)idx->reps.size() || !d_keys || !d_rowpos) return AWRY_ERR_ARG;
  *d_keys = idx->reps[slot]->lcx_key.p;
  *d_rowpos = idx->reps[slot]->lcx_rowpos.p;
  return AWRY_OK;
}
const void* awry_debug_dense_sa(const awry_index_t* idx, int slot) {
  return idx && slot >= 0 && slot < (int)idx->reps.size() ? (const void*)idx->reps[slot]->dense_sa.p : nullptr;
}
int awry_verify_enabled(const awry_index_t* idx) {
  return idx && !idx->reps.empty() && (idx->reps[0]->dev.text4 != nullptr || idx->reps[0]->dev.text8 != nullptr);
}

int awry_locate_sa_ratio(const awry_index_t* idx) {
  if (!idx || idx->reps.empty()) return 0;
  return idx->reps[0]->dense_ratio ? (int)idx->reps[0]->dense_ratio : (int)idx->host.sa_ratio;
}

int awry_dev_count_ascii(awry_index_t* idx, int slot, const void* d_qbytes, const void* d_qoff, uint64_t n, void* d_counts,
                         void* d_ranges, void* d_status, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_qoff && d_counts) || n == 0, "null device pointer");
    launch_count_ascii(r, (const uint8_t*)d_qbytes, (const uint64_t*)d_qoff, n, (uint64_t*)d_counts, (uint64_t*)d_ranges,
                       (uint8_t*)d_status, (hipStream_t)stream, d_ranges == nullptr);  // ranges requested: they are row intervals
  });
}

int awry_dev_count_ascii_for_locate(awry_index_t* idx, int slot, const void* d_qbytes, const void* d_qoff, uint64_t n, void* d_counts,
                                    void* d_locate_words, void* d_status, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_qoff && d_counts && d_locate_words) || n == 0, "null device pointer");
    launch_count_ascii(r, (const uint8_t*)d_qbytes, (const uint64_t*)d_qoff, n, (uint64_t*)d_counts, (uint64_t*)d_locate_words,
                       (uint8_t*)d_status, (hipStream_t)stream, true);  // RS_* words where the count pass verified against the text
  });
}

int awry_dev_count_ascii_uniform(awry_index_t* idx, int slot, const void* d_qbytes, uint64_t n, uint64_t len, void* d_counts,
                                 void* d_status, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_qbytes && d_counts) || n == 0, "null device pointer");
    launch_count_ascii_uniform(r, (const uint8_t*)d_qbytes, n, len, (uint64_t*)d_counts, (uint8_t*)d_status, (hipStream_t)stream);
  });
}

int awry_dev_count_ascii_uniform_tally(awry_index_t* idx, int slot, const void* d_qbytes, uint64_t n, uint64_t len, void* d_counts,
                                       void* d_tally, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_qbytes && d_counts && d_tally) || n == 0, "null device pointer");
    require(r.dev.alphabet == AMINO, "the census of the uniform entry point is kept by the amino k-mer schedule");
    launch_count_ascii_uniform(r, (const uint8_t*)d_qbytes, n, len, (uint64_t*)d_counts, nullptr, (hipStream_t)stream, nullptr,
                               (unsigned long long*)d_tally);
  });
}

uint64_t awry_dev_scan_scratch_bytes(uint64_t n) { return (scan_tiles(n) + 1) * 8; }

int awry_dev_scan_counts(awry_index_t* idx, int slot, const void* d_counts, uint64_t n, void* d_hit_off, void* d_scratch, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require(d_hit_off && ((d_counts && d_scratch) || n == 0), "null device pointer");
    launch_scan(r, (const uint64_t*)d_counts, n, (uint64_t*)d_hit_off, (uint64_t*)d_scratch, (hipStream_t)stream);
  });
}

int awry_dev_locate(awry_index_t* idx, int slot, const void* d_ranges, int range_stride, const void* d_hit_off, uint64_t n,
                    uint64_t total, void* d_global_pos, void* d_pos, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_ranges && d_hit_off && d_global_pos) || total == 0, "null device pointer");
    require(range_stride == 1 || range_stride == 2, "range_stride must be 1 (starts) or 2 ((start,end) pairs)");
    launch_locate(r, (const uint64_t*)d_ranges, range_stride, (const uint64_t*)d_hit_off, n, total, (uint64_t*)d_global_pos,
                  (uint64_t*)d_pos, (hipStream_t)stream);
  });
}

int awry_dev_locate_tally(awry_index_t* idx, int slot, const void* d_ranges, int range_stride, const void* d_hit_off, uint64_t n,
                          uint64_t total, void* d_global_pos, void* d_pos, void* d_tally, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require((d_ranges && d_hit_off && d_global_pos && d_tally) || total == 0, "null device pointer");
    require(range_stride == 1 || range_stride == 2, "range_stride must be 1 (starts) or 2 ((start,end) pairs)");
    require(r.dev.alphabet == NUCLEOTIDE, "the walk census is kept by the nucleotide walk kernel");
    launch_locate(r, (const uint64_t*)d_ranges, range_stride, (const uint64_t*)d_hit_off, n, total, (uint64_t*)d_global_pos,
                  (uint64_t*)d_pos, (hipStream_t)stream, (unsigned long long*)d_tally);
  });
}

int awry_dev_phase_marker(awry_index_t* idx, int slot, int phase_id, void* stream) {
  return guarded([&] {
    replica(idx, slot);
    require(phase_id >= 1 && phase_id <= 65535, "phase id out of range");
    hipLaunchKernelGGL(phase_marker_kernel, dim3((unsigned)phase_id), dim3(64), 0, (hipStream_t)stream);
    HIP_CHECK(hipGetLastError());
  });
}

int awry_dev_malloc(awry_index_t* idx, int slot, uint64_t bytes, void** d_out) {
  return guarded([&] { replica(idx, slot); require(d_out != nullptr, "null argument"); HIP_CHECK(hipMalloc(d_out, std::max<uint64_t>(bytes, 8))); });
}
int awry_dev_free(awry_index_t* idx, int slot, void* d) {
  return guarded([&] { replica(idx, slot); if (d) HIP_CHECK(hipFree(d)); });
}
int awry_dev_memcpy_h2d(awry_index_t* idx, int slot, void* d_dst, const void* h_src, uint64_t bytes) {
  return guarded([&] { replica(idx, slot); if (bytes) HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); });
}
int awry_dev_memcpy_d2h(awry_index_t* idx, int slot, void* h_dst, const void* d_src, uint64_t bytes) {
  return guarded([&] { replica(idx, slot); if (bytes) HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost)); });
}
int awry_dev_memset(awry_index_t* idx, int slot, void* d_dst, int value, uint64_t bytes) {
  return guarded([&] { replica(idx, slot); if (bytes) HIP_CHECK(hipMemset(d_dst, value, bytes)); });
}
int awry_dev_synchronize(awry_index_t* idx, int slot) {
  return guarded([&] { replica(idx, slot); HIP_CHECK(hipDeviceSynchronize()); });
}
int awry_dev_stream_copy(awry_index_t* idx, int slot, void* d_dst, const void* d_src, uint64_t bytes, void* stream) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require(d_dst && d_src && bytes % 16 == 0, "stream copy needs device pointers and a multiple of 16 bytes");
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)r.num_cus * 16), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_src, (uint4*)d_dst, bytes / 16);
    HIP_CHECK(hipGetLastError());
  });
}
int awry_dev_timer_begin(awry_index_t* idx, int slot, void* stream) {
  return guarded([&] { Replica& r = replica(idx, slot); HIP_CHECK(hipEventRecord(r.ev0, (hipStream_t)stream)); });
}
int awry_dev_timer_end(awry_index_t* idx, int slot, void* stream, float* ms_out) {
  return guarded([&] {
    Replica& r = replica(idx, slot);
    require(ms_out != nullptr, "null argument");
    HIP_CHECK(hipEventRecord(r.ev1, (hipStream_t)stream));
    HIP_CHECK(hipEventSynchronize(r.ev1));
    HIP_CHECK(hipEventElapsedTime(ms_out, r.ev0, r.ev1));
  });
}

}  // extern "C"
