// sa_builder.hip -- index construction on the GPU (FmIndex::new without libsufr).
//
// Replaces, for texts that fit one MI355X (n < 2^32; GRCh38 = 3.1e9 needs ~130 GB of the 288 GB HBM):
//   * libsufr's SufrBuilder (/root/reference src/fm_index.rs:156-181) by prefix doubling
//     (Manber-Myers / Larsson-Sadakane with compaction of the unsorted groups) on rocPRIM radix sorts;
//   * the single pass over the suffix array of src/fm_index.rs:203-240 by streaming kernels that emit the
//     device block layout (layout.h), milestones, prefix sums, the sentinel row and the bit-packed
//     row-sampled SA (src/compressed_suffix_array.rs:51-64).
// The result is bit-identical to the host path (sais.hpp + pack_index), which tests/ verify.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "alphabet.h"
#include "host_index.h"
#include "layout.h"

namespace awry {

namespace {

struct GpuBuildError : std::runtime_error { using std::runtime_error::runtime_error; };

#define GB_CHECK(expr)                                                                                \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      throw GpuBuildError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (sa_builder.hip:" + \
                          std::to_string(__LINE__) + ")");                                            \
  } while (0)

template <class T>
struct Buf {
  T* p = nullptr;
  size_t n = 0;
  Buf() = default;
  explicit Buf(size_t count) : n(count) {
    if (count) {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
      if (e != hipSuccess) { p = nullptr; throw GpuBuildError(std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + " B) failed: " + hipGetErrorString(e)); }
    }
  }
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  ~Buf() { if (p) (void)hipFree(p); }
};

constexpr int TPB = 256;
constexpr int TILE = 2048;  // elements per block in the scans

inline unsigned blocks_for(uint64_t n, int per_block = TPB, uint64_t cap = 1u << 20) {
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + per_block - 1) / per_block, cap));
}

// ---------------------------------------------------------------------------------------------- scans
struct OpSum { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } static __device__ uint32_t id() { return 0; } };
struct OpMax { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } static __device__ uint32_t id() { return 0; } };

template <class Op>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* tot, Op op) {
  __shared__ uint32_t wsum[TPB / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v = op(v, o);
  }
  if (lane == 63) wsum[wv] = v;
  __syncthreads();
  uint32_t base = Op::id(), t = Op::id();
#pragma unroll
  for (int i = 0; i < TPB / 64; i++) {
    if (i < wv) base = op(base, wsum[i]);
    t = op(t, wsum[i]);
  }
  __syncthreads();
  *tot = t;
  return op(base, v);
}

template <class Op>
__global__ __launch_bounds__(TPB) void tile_reduce_kernel(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ tiles) {
  Op op;
  const uint64_t base = (uint64_t)blockIdx.x * TILE;
  uint32_t s = Op::id();
  for (int j = 0; j < TILE / TPB; j++) {
    uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
    if (i < n) s = op(s, in[i]);
  }
  uint32_t tot;
  block_incl_scan(s, &tot, op);
  if (threadIdx.x == 0) tiles[blockIdx.x] = tot;
}

// exclusive scan of the tile aggregates in place (single block); total -> tiles[ntiles]
template <class Op>
__global__ __launch_bounds__(TPB) void tile_scan_kernel(uint32_t* __restrict__ tiles, uint64_t ntiles) {
  Op op;
  uint32_t carry = Op::id();
  for (uint64_t b = 0; b < ntiles; b += TPB) {
    uint64_t i = b + threadIdx.x;
    uint32_t v = i < ntiles ? tiles[i] : Op::id(), tot;
    uint32_t inc = block_incl_scan(v, &tot, op);
    // exclusive = carry op (inclusive of predecessors): recompute from the left neighbour
    uint32_t left = __shfl_up(inc, 1, 64);
    __shared__ uint32_t wlast[TPB / 64];
    if ((threadIdx.x & 63) == 63) wlast[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t ex = (threadIdx.x & 63) ? left : (threadIdx.x >> 6 ? wlast[(threadIdx.x >> 6) - 1] : Op::id());
    __syncthreads();
    if (i < ntiles) tiles[i] = op(carry, ex);
    carry = op(carry, tot);
  }
  if (threadIdx.x == 0) tiles[ntiles] = carry;
}

// out[i] = scan up to and including (INCLUSIVE) or excluding element i
template <class Op, bool INCLUSIVE>
__global__ __launch_bounds__(TPB) void tile_apply_kernel(const uint32_t* __restrict__ in, uint64_t n, const uint32_t* __restrict__ tiles,
                                                         uint32_t* __restrict__ out) {
  Op op;
  const uint64_t base = (uint64_t)blockIdx.x * TILE;
  uint32_t carry = tiles[blockIdx.x];
  __shared__ uint32_t wlast[TPB / 64];
  for (int j = 0; j < TILE / TPB; j++) {
    uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
    uint32_t v = i < n ? in[i] : Op::id(), tot;
    uint32_t inc = block_incl_scan(v, &tot, op);
    uint32_t r;
    if (INCLUSIVE) r = op(carry, inc);
    else {
      uint32_t left = __shfl_up(inc, 1, 64);
      if ((threadIdx.x & 63) == 63) wlast[threadIdx.x >> 6] = inc;
      __syncthreads();
      uint32_t ex = (threadIdx.x & 63) ? left : (threadIdx.x >> 6 ? wlast[(threadIdx.x >> 6) - 1] : Op::id());
      __syncthreads();
      r = op(carry, ex);
    }
    if (i < n) out[i] = r;
    carry = op(carry, tot);
  }
}

// scan of n u32 values; `tiles` must hold ceil(n/TILE)+1 values.  Returns nothing; total in tiles[ntiles] (device).
template <class Op, bool INCLUSIVE>
void device_scan(const uint32_t* in, uint64_t n, uint32_t* out, uint32_t* tiles, hipStream_t s) {
  const uint64_t nt = (n + TILE - 1) / TILE;
  hipLaunchKernelGGL((tile_reduce_kernel<Op>), dim3((unsigned)nt), dim3(TPB), 0, s, in, n, tiles);
  hipLaunchKernelGGL((tile_scan_kernel<Op>), dim3(1), dim3(TPB), 0, s, tiles, nt);
  hipLaunchKernelGGL((tile_apply_kernel<Op, INCLUSIVE>), dim3((unsigned)nt), dim3(TPB), 0, s, in, n, tiles, out);
}

// ---------------------------------------------------------------------------------------------- SA kernels

struct CodeLut { uint8_t code[256]; };

// round 0: key = the first P symbols of suffix i as dense order-preserving codes, most significant first
__global__ __launch_bounds__(TPB) void initial_keys_kernel(const uint8_t* __restrict__ text, uint64_t n, CodeLut lut, int bits, int P,
                                                           uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
    uint64_t k = 0;
    for (int j = 0; j < P; j++) {
      uint64_t c = i + j < n ? lut.code[text[i + j]] : 0;
      k = (k << bits) | c;
    }
    keys[i] = k;
    vals[i] = (uint32_t)i;
  }
}

// head[j] = j if position j starts a new group (key differs from its left neighbour), else 0
__global__ __launch_bounds__(TPB) void heads_from_keys_kernel(const uint64_t* __restrict__ keys, uint64_t n, const uint32_t* __restrict__ pos,
                                                              uint32_t* __restrict__ head) {
  for (uint64_t j = (uint64_t)blockIdx.x * TPB + threadIdx.x; j < n; j += (uint64_t)gridDim.x * TPB) {
    bool is_head = j == 0 || keys[j] != keys[j - 1];
    head[j] = is_head ? (pos ? pos[j] : (uint32_t)j) : 0u;
  }
}

// isa[sa[j]] = group start of position j
__global__ __launch_bounds__(TPB) void scatter_rank_kernel(const uint32_t* __restrict__ sa_vals, const uint32_t* __restrict__ grp, uint64_t n,
                                                           uint32_t* __restrict__ isa) {
  for (uint64_t j = (uint64_t)blockIdx.x * TPB + threadIdx.x; j < n; j += (uint64_t)gridDim.x * TPB) isa[sa_vals[j]] = grp[j];
}

// flag[j] = 1 if the element at (compact) index j belongs to a group of more than one element.
// grp[] is non-decreasing, equal inside a group.
__global__ __launch_bounds__(TPB) void active_flags_kernel(const uint32_t* __restrict__ grp, uint64_t n, uint32_t* __restrict__ flag) {
  for (uint64_t j = (uint64_t)blockIdx.x * TPB + threadIdx.x; j < n; j += (uint64_t)gridDim.x * TPB) {
    bool same_left = j > 0 && grp[j - 1] == grp[j];
    bool same_right = j + 1 < n && grp[j + 1] == grp[j];
    flag[j] = (same_left || same_right) ? 1u : 0u;
  }
}

// compaction: where flag, out_pos[excl[j]] = position, out_val[excl[j]] = suffix
__global__ __launch_bounds__(TPB) void compact_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ excl, uint64_t n,
                                                      const uint32_t* __restrict__ pos_in /* nullable: identity */,
                                                      const uint32_t* __restrict__ val_in, uint32_t* __restrict__ pos_out,
                                                      uint32_t* __restrict__ val_out) {
  for (uint64_t j = (uint64_t)blockIdx.x * TPB + threadIdx.x; j < n; j += (uint64_t)gridDim.x * TPB)
    if (flag[j]) {
      pos_out[excl[j]] = pos_in ? pos_in[j] : (uint32_t)j;
      val_out[excl[j]] = val_in[j];
    }
}

// key = (group start of the suffix, rank of the suffix h symbols further + 1)
__global__ __launch_bounds__(TPB) void doubling_keys_kernel(const uint32_t* __restrict__ vals, uint64_t na, const uint32_t* __restrict__ isa,
                                                            uint64_t n, uint64_t h, uint64_t* __restrict__ keys) {
  for (uint64_t a = (uint64_t)blockIdx.x * TPB + threadIdx.x; a < na; a += (uint64_t)gridDim.x * TPB) {
    const uint64_t s = vals[a];
    const uint64_t second = s + h < n ? (uint64_t)isa[s + h] + 1 : 0;
    keys[a] = ((uint64_t)isa[s] << 32) | second;
  }
}

__global__ __launch_bounds__(TPB) void write_back_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ vals, uint64_t na,
                                                         uint32_t* __restrict__ sa) {
  for (uint64_t a = (uint64_t)blockIdx.x * TPB + threadIdx.x; a < na; a += (uint64_t)gridDim.x * TPB) sa[pos[a]] = vals[a];
}

// ---------------------------------------------------------------------------------------------- packing kernels

// one wave per 64 BWT rows: planes by ballot, per-block letter histogram by atomics, sentinel row
template <int A>
__global__ __launch_bounds__(TPB) void bwt_planes_kernel(const uint8_t* __restrict__ text, const uint32_t* __restrict__ sa, uint64_t n,
                                                         uint64_t nblocks, uint64_t* __restrict__ blocks, uint32_t* __restrict__ blockcnt,
                                                         unsigned long long* __restrict__ sentinel_row) {
  constexpr int P = A == NUCLEOTIDE ? 3 : 5;
  constexpr int BW = A == NUCLEOTIDE ? NT_BLOCK_WORDS : AA_BLOCK_WORDS;
  constexpr int NL = A == NUCLEOTIDE ? 4 : 21;
  const int lane = threadIdx.x & 63;
  const uint64_t nslices = nblocks * 4;
  for (uint64_t slice = ((uint64_t)blockIdx.x * TPB + threadIdx.x) >> 6; slice < nslices; slice += ((uint64_t)gridDim.x * TPB) >> 6) {
    const uint64_t r = slice * 64 + lane;
    int idx = -1;  // rows past bwt_len carry no symbol (all-zero code)
    if (r < n) {
      const uint32_t v = sa[r];
      if (v == 0) *sentinel_row = r;
      idx = index_of_ascii(A, v == 0 ? (uint8_t)'$' : text[v - 1]);  // src/fm_index.rs:220-227
    }
    const uint32_t code = idx < 0 ? 0u : (A == NUCLEOTIDE ? nt_code_of_index(idx) : aa_code_of_index(idx));
    const uint64_t b = slice >> 2;
    const int l = (int)(slice & 3);
    uint64_t* blk = blocks + b * BW;
#pragma unroll
    for (int p = 0; p < P; p++) {
      const uint64_t w = __ballot((code >> p) & 1u);
      if (lane == 0) blk[plane_word(A, p, l)] = w;
    }
#pragma unroll
    for (int t = 0; t < NL; t++) {
      const int want = A == NUCLEOTIDE ? nt_index_of_letter(t) : t + 1;
      const uint64_t w = __ballot(idx == want);
      if (lane == 0 && w) atomicAdd(&blockcnt[(uint64_t)t * nblocks + b], (uint32_t)__popcll(w));
    }
  }
}

// exclusive scan results (u32, per letter) -> milestone slots of every block
template <int A>
__global__ __launch_bounds__(TPB) void milestones_kernel(const uint32_t* __restrict__ ms /* [letter][block] */, uint64_t nblocks,
                                                         uint64_t* __restrict__ blocks) {
  constexpr int BW = A == NUCLEOTIDE ? NT_BLOCK_WORDS : AA_BLOCK_WORDS;
  for (uint64_t b = (uint64_t)blockIdx.x * TPB + threadIdx.x; b < nblocks; b += (uint64_t)gridDim.x * TPB) {
    uint64_t* blk = blocks + b * BW;
    if (A == NUCLEOTIDE) {
      for (int l = 0; l < 4; l++) blk[nt_ms_word(l)] = ms[(uint64_t)l * nblocks + b];
    } else {
      uint64_t w[12];
      for (int i = 0; i < 12; i++) w[i] = 0;
      for (int t = 0; t < 21; t++) w[t >> 1] |= (uint64_t)ms[(uint64_t)t * nblocks + b] << (32 * (t & 1));
      for (int t = 0; t < 24; t += 2) blk[aa_ms_word(t)] = w[t >> 1];
    }
  }
}

// bit-packed row-sampled SA, one thread per output word (src/compressed_suffix_array.rs:51-64)
__global__ __launch_bounds__(TPB) void sa_samples_kernel(const uint32_t* __restrict__ sa, uint64_t nsamp, uint64_t ratio, uint64_t bits,
                                                         uint64_t nwords, uint64_t* __restrict__ words) {
  for (uint64_t w = (uint64_t)blockIdx.x * TPB + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * TPB) {
    uint64_t out = 0;
    uint64_t j = (w * 64) / bits;  // first sample overlapping this word
    for (; j < nsamp && j * bits < (w + 1) * 64; j++) {
      const uint64_t v = sa[j * ratio], off = j * bits;
      if (off >= w * 64) out |= v << (off - w * 64);
      else out |= v >> (w * 64 - off);
    }
    words[w] = out;
  }
}

struct SortTemp {  // rocPRIM picks its algorithm (and temporary size) per call from n and the bit range
  void* p = nullptr;
  size_t bytes = 0;
  ~SortTemp() { if (p) (void)hipFree(p); }
  void reserve(size_t need, hipStream_t s) {
    if (need <= bytes) return;
    if (p) { GB_CHECK(hipStreamSynchronize(s)); (void)hipFree(p); p = nullptr; bytes = 0; }
    GB_CHECK(hipMalloc(&p, need + 256));
    bytes = need + 256;
  }
};

void radix_sort(SortTemp& tmp, rocprim::double_buffer<uint64_t>& k, rocprim::double_buffer<uint32_t>& v, uint64_t n,
                unsigned begin_bit, unsigned end_bit, hipStream_t s) {
  size_t need = 0;
  GB_CHECK(rocprim::radix_sort_pairs(nullptr, need, k, v, (size_t)n, begin_bit, end_bit, s));
  tmp.reserve(need, s);
  GB_CHECK(rocprim::radix_sort_pairs(tmp.p, need, k, v, (size_t)n, begin_bit, end_bit, s));
}

unsigned bits_for(uint64_t v) { unsigned b = 0; while (b < 64 && (v >> b)) b++; return b; }

}  // namespace

// Builds the suffix array of `text` (n bytes, ends in the unique smallest byte '$') on `device` and fills
// ix.blocks / sa_words / prefix_sums / sentinel_row exactly as pack_index() does.  n must be < 2^32 - 1.
void gpu_build_index(HostIndex& ix, const uint8_t* text, uint64_t n, int device, bool verbose) {
  if (n >= (1ull << 32) - 1) throw GpuBuildError("GPU index construction needs bwt_len < 2^32");
  if (getenv("AWRY_DEBUG_FAIL_GPU_BUILD")) throw GpuBuildError("forced failure (AWRY_DEBUG_FAIL_GPU_BUILD)");  // tests: the host fallback
  const int A = ix.alphabet;
  if (A == AMINO && n >= (1ull << 32)) throw GpuBuildError("amino bwt_len >= 2^32 unsupported");
  GB_CHECK(hipSetDevice(device));
  hipStream_t s = nullptr;  // default stream: everything below is strictly ordered
  hipEvent_t e0, e1;
  GB_CHECK(hipEventCreate(&e0));
  GB_CHECK(hipEventCreate(&e1));
  GB_CHECK(hipEventRecord(e0, s));

  // dense order-preserving codes of the bytes that occur
  std::vector<uint64_t> hist(256, 0);
  for (uint64_t i = 0; i < n; i++) hist[text[i]]++;
  CodeLut lut{};
  unsigned sigma = 0;
  for (int c = 0; c < 256; c++)
    if (hist[c]) lut.code[c] = (uint8_t)sigma++;
  if (text[n - 1] != '$' || hist['$'] != 1 || lut.code['$'] != 0) throw GpuBuildError("text must end with a unique, smallest '$'");
  const int bits = std::max(1u, bits_for(sigma - 1));
  const int P = 64 / bits;

  Buf<uint8_t> d_text(n);
  GB_CHECK(hipMemcpy(d_text.p, text, n, hipMemcpyHostToDevice));
  Buf<uint32_t> d_sa(n), d_isa(n), d_va(n), d_vb(n), d_pos(n), d_tmp(n), d_pos2(n);
  Buf<uint64_t> d_ka(n), d_kb(n);
  const uint64_t ntiles = (n + TILE - 1) / TILE;
  Buf<uint32_t> d_tiles(ntiles + 2);
  SortTemp sort_tmp;

  // ---- round 0: sort by the first P symbols
  hipLaunchKernelGGL(initial_keys_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, d_text.p, n, lut, bits, P, d_ka.p, d_va.p);
  rocprim::double_buffer<uint64_t> dk(d_ka.p, d_kb.p);
  rocprim::double_buffer<uint32_t> dv(d_va.p, d_vb.p);
  radix_sort(sort_tmp, dk, dv, n, 0, (unsigned)(bits * P), s);
  GB_CHECK(hipMemcpyAsync(d_sa.p, dv.current(), n * 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(heads_from_keys_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, dk.current(), n, (const uint32_t*)nullptr, d_tmp.p);
  device_scan<OpMax, true>(d_tmp.p, n, d_tmp.p, d_tiles.p, s);  // d_tmp = group start of every position
  hipLaunchKernelGGL(scatter_rank_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, d_sa.p, d_tmp.p, n, d_isa.p);
  // active set = positions in groups of size > 1
  uint32_t* flag = reinterpret_cast<uint32_t*>(dk.alternate());  // scratch (n u32 fit in n u64)
  uint32_t* excl = flag + n;
  hipLaunchKernelGGL(active_flags_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, d_tmp.p, n, flag);
  device_scan<OpSum, false>(flag, n, excl, d_tiles.p, s);
  uint32_t na32 = 0;
  GB_CHECK(hipMemcpyAsync(&na32, d_tiles.p + ntiles, 4, hipMemcpyDeviceToHost, s));
  GB_CHECK(hipStreamSynchronize(s));
  uint64_t na = na32;
  uint32_t *pos = d_pos.p, *pos_next = d_pos2.p;
  uint32_t* vals = dv.alternate();  // compact suffix list of the active set
  hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, flag, excl, n, (const uint32_t*)nullptr, d_sa.p, pos, vals);
  uint32_t* vals_other = dv.current();
  GB_CHECK(hipGetLastError());

  // ---- doubling rounds over the shrinking active set
  uint64_t h = (uint64_t)P;
  int round = 0;
  while (na > 0) {
    if (verbose) fprintf(stderr, "[awry gpu build] round %d: h=%llu active=%llu\n", round, (unsigned long long)h, (unsigned long long)na);
    if (h >= 2 * n) throw GpuBuildError("prefix doubling did not converge (duplicate suffixes?)");
    uint64_t* keys = d_ka.p;
    hipLaunchKernelGGL(doubling_keys_kernel, dim3(blocks_for(na)), dim3(TPB), 0, s, vals, na, d_isa.p, n, h, keys);
    rocprim::double_buffer<uint64_t> k2(d_ka.p, d_kb.p);
    rocprim::double_buffer<uint32_t> v2(vals, vals_other);
    radix_sort(sort_tmp, k2, v2, na, 0, 32 + bits_for(n), s);
    const uint32_t* sv = v2.current();
    hipLaunchKernelGGL(write_back_kernel, dim3(blocks_for(na)), dim3(TPB), 0, s, pos, sv, na, d_sa.p);
    // new group starts among the active elements (positions are increasing, so a max-scan carries the head position)
    hipLaunchKernelGGL(heads_from_keys_kernel, dim3(blocks_for(na)), dim3(TPB), 0, s, k2.current(), na, (const uint32_t*)pos, d_tmp.p);
    device_scan<OpMax, true>(d_tmp.p, na, d_tmp.p, d_tiles.p, s);
    hipLaunchKernelGGL(scatter_rank_kernel, dim3(blocks_for(na)), dim3(TPB), 0, s, sv, d_tmp.p, na, d_isa.p);
    flag = reinterpret_cast<uint32_t*>(k2.alternate());
    excl = flag + na;
    hipLaunchKernelGGL(active_flags_kernel, dim3(blocks_for(na)), dim3(TPB), 0, s, d_tmp.p, na, flag);
    device_scan<OpSum, false>(flag, na, excl, d_tiles.p, s);
    const uint64_t nt2 = (na + TILE - 1) / TILE;
    GB_CHECK(hipMemcpyAsync(&na32, d_tiles.p + nt2, 4, hipMemcpyDeviceToHost, s));
    uint32_t* nv = v2.alternate();
    hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(na)), dim3(TPB), 0, s, flag, excl, na, (const uint32_t*)pos, sv, pos_next, nv);
    GB_CHECK(hipGetLastError());
    GB_CHECK(hipStreamSynchronize(s));
    vals_other = const_cast<uint32_t*>(sv);
    vals = nv;
    std::swap(pos, pos_next);
    na = na32;
    h *= 2;
    round++;
  }

  // ---- BWT planes, milestones, prefix sums, sentinel row, sampled SA
  const int BW = block_words(A), NL = A == NUCLEOTIDE ? 4 : 21, card = cardinality(A);
  ix.bwt_len = n;
  ix.nblocks = (n + 255) / 256;
  ix.sa_bits = csa_bits_per_element(n);
  const uint64_t nb = ix.nblocks, nwords = csa_word_len(n, ix.sa_ratio), nsamp = (n + ix.sa_ratio - 1) / ix.sa_ratio;
  Buf<uint64_t> d_blocks(nb * BW), d_words(nwords + 1);
  Buf<uint32_t> d_cnt((uint64_t)NL * nb), d_ms((uint64_t)NL * nb);
  Buf<unsigned long long> d_sent(1);
  GB_CHECK(hipMemsetAsync(d_blocks.p, 0, nb * BW * 8, s));
  GB_CHECK(hipMemsetAsync(d_cnt.p, 0, (uint64_t)NL * nb * 4, s));
  if (A == NUCLEOTIDE)
    hipLaunchKernelGGL(bwt_planes_kernel<NUCLEOTIDE>, dim3(blocks_for(nb * 256)), dim3(TPB), 0, s, d_text.p, d_sa.p, n, nb, d_blocks.p, d_cnt.p, d_sent.p);
  else
    hipLaunchKernelGGL(bwt_planes_kernel<AMINO>, dim3(blocks_for(nb * 256)), dim3(TPB), 0, s, d_text.p, d_sa.p, n, nb, d_blocks.p, d_cnt.p, d_sent.p);
  std::vector<uint64_t> total(24, 0);
  const uint64_t nbt = (nb + TILE - 1) / TILE;
  for (int t = 0; t < NL; t++) {
    device_scan<OpSum, false>(d_cnt.p + (uint64_t)t * nb, nb, d_ms.p + (uint64_t)t * nb, d_tiles.p, s);
    uint32_t tot = 0;
    GB_CHECK(hipMemcpyAsync(&tot, d_tiles.p + nbt, 4, hipMemcpyDeviceToHost, s));
    GB_CHECK(hipStreamSynchronize(s));
    total[A == NUCLEOTIDE ? nt_index_of_letter(t) : t + 1] = tot;
  }
  if (A == NUCLEOTIDE) hipLaunchKernelGGL(milestones_kernel<NUCLEOTIDE>, dim3(blocks_for(nb)), dim3(TPB), 0, s, d_ms.p, nb, d_blocks.p);
  else hipLaunchKernelGGL(milestones_kernel<AMINO>, dim3(blocks_for(nb)), dim3(TPB), 0, s, d_ms.p, nb, d_blocks.p);
  if (ix.sa_bits)
    hipLaunchKernelGGL(sa_samples_kernel, dim3(blocks_for(nwords)), dim3(TPB), 0, s, d_sa.p, nsamp, ix.sa_ratio, ix.sa_bits, nwords, d_words.p);
  GB_CHECK(hipGetLastError());
  total[0] = 1;  // exactly one '$'
  if (A == NUCLEOTIDE) total[4] = n - 1 - total[1] - total[2] - total[3] - total[5];  // everything else is N
  ix.prefix_sums.assign(card + 1, 0);
  uint64_t acc = 0;
  for (int i = 0; i <= card; i++) { ix.prefix_sums[i] = acc; if (i != card) acc += total[i]; }
  if (acc != n) throw GpuBuildError("letter counts do not add up to bwt_len");
  ix.blocks.resize(nb * BW);
  ix.sa_words.resize(nwords);
  GB_CHECK(hipMemcpyAsync(ix.blocks.data(), d_blocks.p, nb * BW * 8, hipMemcpyDeviceToHost, s));
  if (nwords) GB_CHECK(hipMemcpyAsync(ix.sa_words.data(), d_words.p, nwords * 8, hipMemcpyDeviceToHost, s));
  unsigned long long sent = 0;
  GB_CHECK(hipMemcpyAsync(&sent, d_sent.p, 8, hipMemcpyDeviceToHost, s));
  GB_CHECK(hipEventRecord(e1, s));
  GB_CHECK(hipStreamSynchronize(s));
  ix.sentinel_row = sent;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (verbose) fprintf(stderr, "[awry gpu build] n=%llu rounds=%d device time %.1f ms\n", (unsigned long long)n, round, ms);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

}  // namespace awry
