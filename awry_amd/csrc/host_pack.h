// host_pack.h -- host half of the drop-in batch boundary (awry_count_batch / awry_locate_batch): a persistent worker
// pool and the AVX2 packer that turns ASCII nucleotide queries into the 2-bit words the packed kernels read, so that
// 8 B per 31-mer cross PCIe instead of 31 B (hard part 2; caller side of
// /root/reference src/fm_index.rs:455-487).  Same arithmetic as pack_nt2_tile_kernel (kernels.hip.h): case fold,
// membership test against A C G T, bits 1..2 of the ASCII code, swap of the last two codes, squeeze.
// Nothing here searches: queries with any other byte are only LISTED, and the device redoes them with the generic kernel.
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace awry {

// CPUs this process may really use: cgroup quota, else affinity mask (the GPU boxes show 256 logical CPUs and grant 16);
// AWRY_HOST_THREADS overrides
unsigned effective_cpus();

// Persistent pool (created on first use, effective_cpus() threads counting the caller).  run() hands the indices
// [0, n) to the workers and the calling thread and returns when all are done; one job at a time (callers queue).
class HostPool {
 public:
  static HostPool& instance();
  unsigned threads() const;
  void run(uint64_t n, const std::function<void(uint64_t)>& fn);
  // [0, n) cut into pieces of about `grain` items: fn(lo, hi) per piece
  void run_ranges(uint64_t n, uint64_t grain, const std::function<void(uint64_t, uint64_t)>& fn);
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;

 private:
  HostPool();
  ~HostPool();
  struct Impl;
  Impl* impl_;
};

// queries [lo, hi) of a batch -> packed words.  Query q is ascii[off[q] - off[lo] .. off[q + 1] - off[lo]) when off is
// given (ragged), else ascii[(q - lo) * L ..) with L letters; its W = ceil(L / 32) words go to words[(q - lo) * W ..]
// (letter j in word j / 32, bits 2 (j % 32); A0 C1 G2 T3; unused bits and words zero) and, ragged, its length to
// lens[q - lo].  ascii points at the first byte of query lo; ascii_end is the end of the readable buffer (32-byte loads
// never cross it).  Queries holding a byte outside ACGTacgt are appended to `bad` (index relative to lo); their words
// are unspecified.  Runs on the pool.
// check_off (with off == nullptr): the caller only ASSUMES that every query has L letters; the packer reads the batch's
// offsets as it goes and returns false as soon as check_off[q + 1] - check_off[q] != L for some q in [lo, hi) (the
// output is then unspecified) -- this folds the length scan of a batch into the pass that reads its bytes anyway.
bool pack_nt2_host(const uint8_t* ascii, const uint8_t* ascii_end, const uint64_t* off, uint64_t lo, uint64_t hi, uint64_t L,
                   uint64_t* words, uint32_t* lens, std::vector<uint32_t>& bad, const uint64_t* check_off = nullptr);

// memcpy cut over the pool (result-sized copies out of pinned staging)
void pool_memcpy(void* dst, const void* src, size_t bytes);
// dst[i] = src[i], u32 -> u64, over the pool: counts cross PCIe as 32-bit words (a count is < bwt_len < 2^32 on the packed paths)
void pool_widen_u32(uint64_t* dst, const uint32_t* src, uint64_t n);

}  // namespace awry
