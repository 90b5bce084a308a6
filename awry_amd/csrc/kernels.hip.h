// kernels.hip.h -- gfx950 (CDNA4, wave64) kernels of the FM-index hot path.
//
//   rank / Occ        /root/reference src/bwt.rs:114-135,230-271 + src/simd_instructions.rs:96-121
//   step              src/fm_index.rs:559-582 (update_range_with_symbol)
//   backward search   src/fm_index.rs:402-438 + src/kmer_lookup_table.rs:90-110
//   backtrace/locate  src/fm_index.rs:516-544,585-593 + src/compressed_suffix_array.rs:76-111
//   localisation      src/sequence_index.rs:108-141 (intended semantics)
//
// Integer/bit path only (no MFMA): every kernel is bound by random 128-B line fetches from HBM.
// Two families:
//   * "scalar" kernels: one query (or one hit) per lane, any alphabet, any symbol, any length;
//   * "quad" kernels: the hot count path for packed nucleotide k-mers.  A wavefront holds 16
//     independent queries, one per QUAD of lanes; a quad fetches a 128-B block as 2 x
//     global_load_dwordx4 per lane (4 lanes x 16 B = one 64-B half line per instruction), ranks its
//     64-symbol slice with __popcll and sums partials with two quad_perm DPP adds -- no LDS round trip.
//
// Kernel map (default in CAPS; the others are kept as measured alternatives):
//   count, any query      COUNT_SCALAR_KERNEL<A>             ASCII + offsets, one query per lane, seed probe, verify against text8
//   count, amino k-mers   COUNT_AA_KMER_PROBE_KERNEL         equal-length ASCII residues, one query per lane: entry / text decide most
//                         + COUNT_SCALAR_KERNEL<AMINO, LIST> the generic kernel on the listed rest
//   count, packed k-mers  COUNT_NT2_PROBE_KERNEL             phase 1: one query per lane, entry / context / text decide most
//                         + COUNT_NT2_RESUME_KERNEL          phase 2: quads resume the listed survivors (sparse seed tables)
//                         COUNT_NT2_QUAD4_KERNEL             groups of four queries per quad (dense seed tables)
//                         count_nt2_quad_kernel              one strided query per quad
//                         count_nt2_chunk_kernel             queries/results staged through LDS per wave
//   count, packed reads   COUNT_NT2_READS_PROBE_KERNEL<R>    phase 1 for reads of any (per-read) length
//                         + COUNT_NT2_READS_KERNEL<..LIST>   quads on the listed reads; without LIST: the single-kernel schedule
//   locate                LOCATE_TILE_KERNEL<A>              hit -> row, sampled / verified hits finished
//                         + LOCATE_WALK_NT_LANE_KERNEL       LF walks of the rest, one hit per lane, whole block per step
//                         + LOCALISE_WALKED_KERNEL           record / offset of the walked hits
//                         locate_walk_kernel<A>              generic walk (amino), locate_scalar_kernel<A> (round-1 baseline)
//   accelerators          seed_level1/extend/finalize (+aa_*), seed_rows_to_positions_kernel, densify_sa_kernel,
//                         nblock_sa_kernel, text4_scatter_kernel, text8_scatter_kernel
//   glue                  pack_nt2_tile_kernel<R>, scan_*_kernel, ref_kmer_table_kernel,
//                         scalar_ops_kernel
#pragma once
#include <hip/hip_runtime.h>

#include "alphabet.h"
#include "layout.h"

namespace awry {

// A seed-table probe reads 8 bytes of a line nobody will touch again (the table is 10..140 GB and the probes are random):
// loaded non-temporally, so that the line does not displace the streams that do have locality (query words, counts,
// survivor lists) from L2 / Infinity Cache.  Measured on the headline batch: 33.5 -> 35.9 G queries/s.
__device__ __forceinline__ SeedEntry seed_probe(const SeedEntry* __restrict__ p) {
  const unsigned long long raw = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
  return SeedEntry{(uint32_t)raw, (uint32_t)(raw >> 32)};
}

// ------------------------------------------------------------------------------------------------
// scalar helpers (one lane does a whole rank)
// ------------------------------------------------------------------------------------------------

// inclusive mask of bits 0..=t of a 64-symbol slice; t < 0 -> none, t >= 63 -> all
__device__ __forceinline__ uint64_t slice_mask(int t) {
  uint64_t m = ~0ull >> (63 - (t > 63 ? 63 : (t < 0 ? 0 : t)));
  return t < 0 ? 0ull : m;
}

template <int A>
__device__ __forceinline__ uint64_t slice_pred(const uint64_t* blk, int l, uint32_t code) {
  uint64_t pr = ~0ull;
#pragma unroll
  for (int b = 0; b < (A == NUCLEOTIDE ? 3 : 5); b++) {
    uint64_t x = ((code >> b) & 1u) ? 0ull : ~0ull;
    pr &= blk[plane_word(A, b, l)] ^ x;
  }
  return pr;
}

// milestone of symbol index `idx` at the start of block `b` (exclusive prefix count, src/fm_index.rs:212-217)
template <int A>
__device__ __forceinline__ uint64_t milestone(const DevIndex& ix, const uint64_t* blk, uint64_t b, int idx) {
  if (A == NUCLEOTIDE) {
    int letter = nt_letter_of_index(idx);
    if (letter >= 0) return blk[nt_ms_word(letter)];
    // N is derived: rows before the block that are neither A,C,G,T nor the single '$'
    uint64_t sum = blk[nt_ms_word(0)] + blk[nt_ms_word(1)] + blk[nt_ms_word(2)] + blk[nt_ms_word(3)];
    return 256ull * b - sum - (ix.sentinel_row < 256ull * b ? 1ull : 0ull);
  }
  int t = idx - 1;
  return (blk[aa_ms_word(t)] >> (32 * aa_ms_half(t))) & 0xffffffffull;
}

// Occ(idx, row) inclusive of `row`: src/bwt.rs:338-357
template <int A>
__device__ __forceinline__ uint64_t rank_scalar(const DevIndex& ix, uint64_t row, int idx) {
  const uint64_t b = row >> 8;
  const int p = (int)(row & 255);
  const uint64_t* blk = ix.blocks + b * (A == NUCLEOTIDE ? NT_BLOCK_WORDS : AA_BLOCK_WORDS);
  const uint32_t code = A == NUCLEOTIDE ? nt_code_of_index(idx) : aa_code_of_index(idx);
  uint32_t cnt = 0;
#pragma unroll
  for (int l = 0; l < 4; l++) cnt += (uint32_t)__popcll(slice_pred<A>(blk, l, code) & slice_mask(p - 64 * l));
  return milestone<A>(ix, blk, b, idx) + cnt;
}

// symbol index stored at BWT row `row`: src/bwt.rs:307-325
template <int A>
__device__ __forceinline__ int symbol_at(const DevIndex& ix, uint64_t row) {
  const uint64_t* blk = ix.blocks + (row >> 8) * (A == NUCLEOTIDE ? NT_BLOCK_WORDS : AA_BLOCK_WORDS);
  const int l = (int)((row >> 6) & 3), bit = (int)(row & 63);
  uint32_t code = 0;
#pragma unroll
  for (int b = 0; b < (A == NUCLEOTIDE ? 3 : 5); b++) code |= (uint32_t)((blk[plane_word(A, b, l)] >> bit) & 1ull) << b;
  return A == NUCLEOTIDE ? nt_index_of_code(code) : aa_index_of_code(code);
}

// src/fm_index.rs:559-582
template <int A>
__device__ __forceinline__ void step_scalar(const DevIndex& ix, uint64_t& sp, uint64_t& ep, int idx) {
  const uint64_t c = ix.prefix_sums[idx];
  const uint64_t s2 = c + rank_scalar<A>(ix, sp - 1, idx);
  ep = c + rank_scalar<A>(ix, ep, idx) - 1;
  sp = s2;
}

// src/fm_index.rs:585-593
template <int A>
__device__ __forceinline__ uint64_t backstep_scalar(const DevIndex& ix, uint64_t row) {
  int idx = symbol_at<A>(ix, row);
  if (idx == 0) return 0;
  return ix.prefix_sums[idx] + rank_scalar<A>(ix, row, idx) - 1;
}

// src/compressed_suffix_array.rs:76-106
__device__ __forceinline__ uint64_t sa_sample(const DevIndex& ix, uint64_t sample) {
  const uint64_t bits = ix.sa_bits;
  if (bits == 0) return 0;
  const uint64_t off = sample * bits, w = off >> 6, s = off & 63;
  uint64_t v = ix.sa_words[w] >> s;
  if (s + bits > 64) v |= ix.sa_words[w + 1] << (64 - s);
  return bits >= 64 ? v : (v & ((1ull << bits) - 1));
}

// ------------------------------------------------------------------------------------------------
// generic count: one ASCII query per lane (any alphabet / symbol / length)
// ------------------------------------------------------------------------------------------------

enum : uint8_t { Q_OK = 0, Q_EMPTY = 1, Q_SENTINEL = 2, Q_NON_ASCII = 3 };

// Byte access through aligned 8-byte loads: one memory instruction per 8 consecutive bytes instead of one per byte
// (the lanes of a wave read different queries, so every byte load is a line lookup of its own in the texture path;
// the generic kernel's loops were bound by exactly that).  Reads the aligned word around a byte: the buffer must be
// readable up to the next 8-byte boundary (device allocations are).
struct ByteStream {
  const uint8_t* base;
  uint64_t word = 0;
  uintptr_t at = ~(uintptr_t)0;
  __device__ __forceinline__ explicit ByteStream(const uint8_t* p) : base(p) {}
  __device__ __forceinline__ uint8_t operator[](uint64_t i) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base + i), w = a & ~(uintptr_t)7;
    if (w != at) { word = *reinterpret_cast<const uint64_t*>(w); at = w; }
    return (uint8_t)(word >> (8 * (a & 7)));
  }
};

// Eight ASCII nucleotide letters -> eight symbol indices, word-wise (src/alphabet.rs:109-114,169-248: A 1, C 2, G 3,
// T / U 5, everything else N = 4; '$' / '#' and bytes >= 0x80 never get here, their queries are rejected).
__device__ __forceinline__ uint64_t nt_indices8(uint64_t x) {
  constexpr uint64_t K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
  const uint64_t c = x & 0xDFDFDFDFDFDFDFDFull;  // upper-case
  auto eq = [&](uint64_t pat) { const uint64_t t = c ^ pat; return (((((t & K7F) + K7F) | t) & K80) ^ K80) >> 7; };  // 1 per equal byte
  const uint64_t a = eq(0x4141414141414141ull), cc = eq(0x4343434343434343ull), g = eq(0x4747474747474747ull),
                 t = eq(0x5454545454545454ull) | eq(0x5555555555555555ull);
  return 0x0404040404040404ull - 3 * a - 2 * cc - g + t;  // bytewise, no borrows: at most one of the masks is set per byte
}

// Do the `rem` symbols text8[0 .. rem) equal the query bytes q[0 .. rem) (as symbol indices)?  Nucleotide: eight at a
// time; both buffers are readable 8 bytes past their end.
template <int A>
__device__ __forceinline__ bool text_equals_query(const uint8_t* __restrict__ text8, const uint8_t* __restrict__ q, uint64_t rem,
                                                  const uint8_t* lut) {
  if (A == NUCLEOTIDE) {
    for (uint64_t j = 0; j < rem; j += 8) {
      uint64_t tw, qw;
      __builtin_memcpy(&tw, text8 + j, 8);
      __builtin_memcpy(&qw, q + j, 8);
      uint64_t d = tw ^ nt_indices8(qw);
      if (rem - j < 8) d &= (1ull << (8 * (rem - j))) - 1;
      if (d) return false;
    }
    return true;
  }
  ByteStream t(text8), a(q);
  for (uint64_t j = 0; j < rem; j++)
    if (t[j] != lut[a[j]]) return false;
  return true;
}

// status[q] != 0 marks inputs the reference leaves undefined: empty query, '$'/'#',
// bytes >= 0x80.  ranges (optional) receives the final (start, end) row interval.
// allow_verify (with the dense SA and ix.text8 resident): once the range has shrunk to <= 4 rows, the letters still to
// the left are compared with the text in front of each candidate instead of being stepped one by one; ranges[2q] then
// holds an RS_SINGLE / RS_MULTI word for the locate pass, not a row interval -- callers that need rows pass 0.
// ulen != 0: every query has ulen bytes, back to back (off is not read).
// LIST_BLOCK: only the queries block b of an earlier pass (same grid) listed for itself, ql.q[b * ql.cap ...) -- the
// second phase of count_aa_kmer_probe_kernel.  LIST_GLOBAL: only the *ql.total queries of one device-wide list, in any
// order -- the reads of a packed nucleotide chunk that hold letters outside ACGT, redone in place; the first query
// (lowest index) with a non-zero status is reported through ql.first_bad as (index << 8 | status).
// LIST_COMPACT: the ql.cap listed queries travel as a CSR batch of their own -- entry `it` is bytes [off[it], off[it + 1])
// of ascii and answers for query ql.q[it] -- the form in which the host-packed paths hand over the few queries of a
// chunk that hold letters outside ACGT (only those bytes cross PCIe); first_bad as for LIST_GLOBAL.
enum { LIST_NONE = 0, LIST_BLOCK = 1, LIST_GLOBAL = 2, LIST_COMPACT = 3 };
struct QueryList {
  uint32_t* q;                      // query indices; LIST_BLOCK: block b owns slots [b * cap, (b + 1) * cap)
  uint32_t* count;                  // LIST_BLOCK: listed queries per block
  uint64_t cap;
  const unsigned long long* total;  // LIST_GLOBAL: number of listed queries
  unsigned long long* first_bad;    // LIST_GLOBAL (nullable): min over rejected queries of (index << 8 | status)
  uint32_t range_stride;            // LIST_GLOBAL: 1 = ranges[q] receives the range start / RS_* word only (the layout of
                                    //   the packed read kernels' range_start), otherwise (start, end) pairs
  unsigned long long* tally;        // nullable work census (untimed runs): [0] seed probes, [1] executed steps, [2] distinct
                                    //   blocks ranked, [3] SA reads and [4] text comparisons of seed-and-verify
  uint32_t nlists;                  // LIST_BLOCK: number of per-block lists (the first pass's grid), <= LIST_MAX_LISTS: the
                                    //   lists are then worked through as ONE pool by whatever grid this pass is launched
                                    //   with (0: block b takes list b)
};
constexpr int LIST_MAX_LISTS = 4096;
__device__ __forceinline__ void tally_add(unsigned long long* tally, int slot, unsigned long long v) {
  if (tally && v) atomicAdd(&tally[slot], v);
}

__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* tot);

template <int A, int LIST = LIST_NONE>
__global__ __launch_bounds__(256) void count_scalar_kernel(DevIndex ix, const uint8_t* __restrict__ ascii,
                                                           const uint64_t* __restrict__ off, uint64_t n,
                                                           uint64_t* __restrict__ counts, uint64_t* __restrict__ ranges,
                                                           uint8_t* __restrict__ status, int allow_verify, uint64_t ulen, QueryList ql) {
  __shared__ uint8_t lut[256];
  // LIST_BLOCK with ql.nlists: exclusive prefix sums of the lists' lengths.  The kernel holds ~140 VGPRs (3 waves per
  // SIMD), so a grid of one block per list ran in three rounds, each as long as the longest chain of dependent loads in
  // it -- 66 us for a few hundred thousand queries; as one pool the listed queries spread over every resident thread.
  __shared__ uint32_t s_pref[LIST == LIST_BLOCK ? LIST_MAX_LISTS + 1 : 1];
  lut[threadIdx.x] = (uint8_t)(threadIdx.x >= 128 ? 0xFF : index_of_ascii(A, (uint8_t)threadIdx.x));
  const bool pooled = LIST == LIST_BLOCK && ql.nlists != 0;
  uint64_t pool_total = 0;
  if (LIST == LIST_BLOCK && pooled) {
    const uint32_t per = (ql.nlists + blockDim.x - 1) / blockDim.x;  // consecutive lists per thread
    const uint32_t l0 = threadIdx.x * per;
    uint64_t mine = 0;
    for (uint32_t j = 0; j < per; j++) mine += l0 + j < ql.nlists ? ql.count[l0 + j] : 0u;
    uint64_t tot;
    uint64_t run = block_excl_scan(mine, &tot);
    for (uint32_t j = 0; j < per; j++)
      if (l0 + j < ql.nlists) { s_pref[l0 + j] = (uint32_t)run; run += ql.count[l0 + j]; }
    if (threadIdx.x == 0) s_pref[ql.nlists] = (uint32_t)tot;
    pool_total = tot;
  }
  __syncthreads();
  const uint64_t stride = LIST == LIST_BLOCK && !pooled ? blockDim.x : (uint64_t)gridDim.x * blockDim.x;
  const uint64_t todo = LIST == LIST_BLOCK ? (pooled ? pool_total : (uint64_t)ql.count[blockIdx.x])
                                           : (LIST == LIST_GLOBAL ? (uint64_t)*ql.total : (LIST == LIST_COMPACT ? ql.cap : n));
  const uint8_t* const ascii_bytes = ascii;
  for (uint64_t it = LIST == LIST_BLOCK && !pooled ? threadIdx.x : (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < todo; it += stride) {
    uint64_t q;
    if (LIST == LIST_BLOCK && pooled) {  // item `it` of the pool: list l with s_pref[l] <= it < s_pref[l + 1]
      uint32_t lo = 0, hi = ql.nlists;
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pref[mid] <= (uint32_t)it) lo = mid; else hi = mid; }
      q = ql.q[(uint64_t)lo * ql.cap + (it - s_pref[lo])];
    } else {
      q = LIST == LIST_BLOCK ? ql.q[(uint64_t)blockIdx.x * ql.cap + it] : (LIST == LIST_GLOBAL || LIST == LIST_COMPACT ? ql.q[it] : it);
    }
    const uint64_t b = LIST == LIST_COMPACT ? off[it] : (ulen ? q * ulen : off[q]);
    const uint64_t e = LIST == LIST_COMPACT ? off[it + 1] : (ulen ? b + ulen : off[q + 1]);
    ByteStream ascii(ascii_bytes);  // shadows the pointer: same indexing, 8 bytes per load
    uint8_t st = e > b ? Q_OK : Q_EMPTY;
    if (A == NUCLEOTIDE) {  // eight bytes at a time: any byte >= 0x80, any '$' or '#'
      constexpr uint64_t K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
      uint64_t high = 0, sent = 0;
      for (uint64_t i = b; i < e; i += 8) {
        uint64_t x;
        __builtin_memcpy(&x, ascii_bytes + i, 8);
        if (e - i < 8) x &= (1ull << (8 * (e - i))) - 1;  // bytes past the query read as 0: neither test fires
        high |= x & K80;
        const uint64_t t1 = x ^ 0x2424242424242424ull, t2 = x ^ 0x2323232323232323ull;
        sent |= (((((t1 & K7F) + K7F) | t1) & K80) ^ K80) | (((((t2 & K7F) + K7F) | t2) & K80) ^ K80);
      }
      if (high) st = Q_NON_ASCII;
      else if (sent && st == Q_OK) st = Q_SENTINEL;
    } else {
      for (uint64_t i = b; i < e; i++) {
        uint8_t s = lut[ascii[i]];
        if (s == 0xFF) st = Q_NON_ASCII;
        else if (s == 0 && st == Q_OK) st = Q_SENTINEL;
      }
    }
    uint64_t sp = 1, ep = 0, vcount = 0, vrs = 0;
    bool verified = false;
    if (st == Q_OK) {
      uint64_t i = e - 1;
      bool seeded = false;
      // reference schedule (awry_search_range): no table, and kmer_len - 1 steps taken whether or not the range is empty
      // (src/kmer_lookup_table.rs:90-110), so that the rows of an ABSENT query are the reference's too
      const bool ref_mode = (allow_verify & 2) != 0;
      uint64_t uncond = ref_mode && e - b >= (uint64_t)(allow_verify >> 8) && (allow_verify >> 8) > 0 ? (uint64_t)(allow_verify >> 8) - 1 : 0;
      if (ref_mode) allow_verify = 0;
      if (!ref_mode && A == AMINO && ix.seed && e - b >= (uint64_t)ix.seed_k) {  // last k residues all standard -> one table probe
        const int k = ix.seed_k;
        uint64_t sidx = 0;
        bool std20 = true;
        for (int j = k - 1; j >= 0; j--) {  // leftmost window letter least significant
          const int letter = aa_letter_of_index(lut[ascii[e - k + j]]);
          std20 = std20 && letter >= 0;
          sidx = sidx * AA_SEED_SIGMA + (uint64_t)(letter < 0 ? 0 : letter);
        }
        if (std20) {
          const SeedEntry se = seed_probe(ix.seed + sidx);
          tally_add(ql.tally, 0, 1);
          const uint32_t scnt = aa_seed_cnt(se);
          // BWT[row] is not the next residue / the next residue does not occur in the BWT over the entry's 2..4 rows
          const bool wrong_sym = e - k > b && ((scnt == 1 && (int)aa_seed_sym(se) != (int)lut[ascii[e - k - 1]]) ||
                                               (aa_seed_is_multi(se) && !((aa_seed_mask(se) >> lut[ascii[e - k - 1]]) & 1u)));
          if (ix.seed_pos && scnt == 1 && !wrong_sym) {  // position seed, as in the nucleotide branch below
            const uint64_t rem = e - k - b, p = se.sp;
            if (allow_verify && ix.text8 && rem < 65536) {
              const bool same = p >= rem && text_equals_query<A>(ix.text8 + (p - rem), ascii_bytes + b, rem, lut);
              tally_add(ql.tally, 4, p >= rem ? 1 : 0);
              verified = true;
              vcount = same ? 1 : 0;
              vrs = same ? ((RS_SINGLE << RS_MODE_SHIFT) | (p - rem)) : ((RS_MULTI << RS_MODE_SHIFT) | (rem << 32));
              seeded = true;
              sp = 1; ep = 0;
            }
          } else if (scnt != AA_SEED_CNT_SAT) {
            sp = scnt ? se.sp : 1;
            ep = scnt ? (uint64_t)se.sp + scnt - 1 : 0;
            i = e - k;
            seeded = true;
            if (wrong_sym) { sp = 1; ep = 0; }
          }
        }
      }
      if (!ref_mode && A == NUCLEOTIDE && ix.seed && e - b >= (uint64_t)ix.seed_k) {  // last k symbols all in ACGT -> one table probe
        const int k = ix.seed_k;
        uint64_t sidx = 0;
        bool acgt = true;
        for (int j = 0; j < k; j++) {
          const int letter = nt_letter_of_index(lut[ascii[e - k + j]]);
          acgt = acgt && letter >= 0;
          sidx |= (uint64_t)(letter & 3) << (2 * j);  // leftmost letter of the window least significant
        }
        if (acgt) {
          const SeedEntry se = seed_probe(ix.seed + sidx);
          tally_add(ql.tally, 0, 1);
          const uint32_t scnt = seed_cnt(se);
          const bool wrong_sym = scnt == 1 && e - k > b && seed_sym(se) != (int)lut[ascii[e - k - 1]];  // BWT[row] is not the next symbol
          // position seeds (ix.seed_pos): a singleton entry names a text position, not a row -- good enough to reject
          // the query by its symbol or to finish it against the text, not to continue the search: without the text
          // such a query starts over without the table
          if (ix.seed_pos && scnt == 1 && !wrong_sym) {
            const uint64_t rem = e - k - b, p = seed_position(se, (int)ix.ctx_extra);
            if (allow_verify && ix.text8 && rem < 65536) {
              // (p < rem: the suffix starts too close to the text's beginning)
              const bool same = p >= rem && text_equals_query<A>(ix.text8 + (p - rem), ascii_bytes + b, rem, lut);
              tally_add(ql.tally, 4, p >= rem ? 1 : 0);
              verified = true;
              vcount = same ? 1 : 0;
              vrs = same ? ((RS_SINGLE << RS_MODE_SHIFT) | (p - rem)) : ((RS_MULTI << RS_MODE_SHIFT) | (rem << 32));
              seeded = true;
              sp = 1; ep = 0;  // skips the step loop below
            }
          } else if (scnt != SEED_CNT_SAT) {
            sp = scnt ? se.sp : 1;
            ep = scnt ? (uint64_t)se.sp + scnt - 1 : 0;
            i = e - k;
            seeded = true;
            if (wrong_sym) { sp = 1; ep = 0; }
          }
        }
      }
      if (!seeded) {
        int idx = lut[ascii[i]];
        sp = ix.prefix_sums[idx];          // SearchRange::new, src/search.rs:43-48
        ep = ix.prefix_sums[idx + 1] - 1;
      }
      const bool can_verify = allow_verify && ix.text8 && ix.dense_sa && ix.dense_ratio == 1;
      while (i > b && (sp <= ep || uncond > 0)) {  // emptiness is sticky, so stopping early never changes the count
        if (uncond > 0) uncond--;
        const uint64_t rem = i - b, cnt = ep - sp + 1;
        // (second pass of the amino k-mer schedule: what counts there is the length of the chain of dependent loads, and
        //  SA + text is two of them where every LF step is one more)
        if (can_verify && cnt <= 4 && (3 * cnt <= rem || LIST == LIST_BLOCK) && rem < 65536) {
          uint32_t mask = 0;
          uint64_t g1 = 0;
          if (A == AMINO && rem <= 24) {
            // short rests (k-mers): the candidates' SA entries, then their text words, are fetched TOGETHER -- two or three
            // dependent round trips for up to four candidates instead of two or three per candidate
            uint32_t pc[4];
#pragma unroll
            for (int c = 0; c < 4; c++) pc[c] = (uint64_t)c < cnt ? ix.dense_sa[sp + c] : 0u;
            uint64_t diff[4] = {0, 0, 0, 0};
            for (uint64_t w0 = 0; w0 < rem; w0 += 8) {
              const int nb = rem - w0 < 8 ? (int)(rem - w0) : 8;
              uint64_t qw = 0;
              for (int t = 0; t < nb; t++) qw |= (uint64_t)lut[ascii[b + w0 + t]] << (8 * t);
              const uint64_t m = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1;
              uint64_t tw[4];
#pragma unroll
              for (int c = 0; c < 4; c++) {
                tw[c] = ~qw;
                if ((uint64_t)c < cnt && pc[c] >= rem) __builtin_memcpy(&tw[c], ix.text8 + ((uint64_t)pc[c] - rem) + w0, 8);  // (16 bytes of slack behind the text)
              }
#pragma unroll
              for (int c = 0; c < 4; c++) diff[c] |= (tw[c] ^ qw) & m;
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
              if ((uint64_t)c < cnt) {
                tally_add(ql.tally, 3, 1);
                if (pc[c] >= rem) {
                  tally_add(ql.tally, 4, 1);
                  if (!diff[c]) { mask |= 1u << c; g1 = (uint64_t)pc[c] - rem; }
                }
              }
          } else
          for (uint64_t c = 0; c < cnt; c++) {
            const uint64_t p = ix.dense_sa[sp + c];
            tally_add(ql.tally, 3, 1);
            if (p < rem) continue;  // the suffix starts too close to the text's beginning
            tally_add(ql.tally, 4, 1);
            if (text_equals_query<A>(ix.text8 + (p - rem), ascii_bytes + b, rem, lut)) { mask |= 1u << c; g1 = p - rem; }
          }
          verified = true;
          vcount = (uint64_t)__popc(mask);
          vrs = (cnt == 1 && mask) ? ((RS_SINGLE << RS_MODE_SHIFT) | g1)
                                   : ((RS_MULTI << RS_MODE_SHIFT) | sp | (rem << 32) | ((uint64_t)mask << 48));
          break;
        }
        i--;
        if (ql.tally) { tally_add(ql.tally, 1, 1); tally_add(ql.tally, 2, ((sp - 1) >> 8) == (ep >> 8) ? 1 : 2); }
        step_scalar<A>(ix, sp, ep, lut[ascii[i]]);
      }
    }
    const bool starts_only = (LIST == LIST_GLOBAL || LIST == LIST_COMPACT) && ql.range_stride == 1;
    if (verified) {
      counts[q] = vcount;
      if (ranges) { if (starts_only) ranges[q] = vrs; else { ranges[2 * q] = vrs; ranges[2 * q + 1] = 0; } }
    } else {
      counts[q] = sp > ep ? 0 : ep - sp + 1;  // src/search.rs:66-71
      if (ranges) { if (starts_only) ranges[q] = sp; else { ranges[2 * q] = sp; ranges[2 * q + 1] = ep; } }
    }
    if (status) status[q] = st;
    if ((LIST == LIST_GLOBAL || LIST == LIST_COMPACT) && ql.first_bad && st != Q_OK) atomicMin(ql.first_bad, ((unsigned long long)q << 8) | st);
  }
}

// ------------------------------------------------------------------------------------------------
// Amino k-mer batches: n ASCII queries of the same length L (AA_KMER_MIN..AA_KMER_MAX residues) back to back -- the
// shape of BASELINE configs[3] (10 M 12-mers).  First phase of a two-phase schedule, one query per LANE, NQ in flight:
// the bytes of a query are two or three unaligned 8-byte loads at q * L (no offsets, no byte stream), an LDS table
// turns each byte into its symbol index and its base-20 digit, the last k residues name one seed entry (the query's
// one random line).  Entry empty: absent.  Singleton whose BWT symbol is not the next residue: absent.  Singleton
// otherwise (position seeds, dense SA and byte text resident): the L - k residues in front of the one candidate are
// one <= 24-B window of the text, compared word-wise; an entry of 2..AA_KMER_VMULTI rows likewise, candidate by
// candidate through the dense SA, when enough lanes of the wave hold one.  Everything else -- a non-standard residue
// in the seed window, bytes the reference leaves undefined, entries with more rows, row seeds -- is listed per block
// and redone by count_scalar_kernel<AMINO, LIST_BLOCK> on the same grid.  The generic kernel spends ~1 900 wave instructions
// per 64 such queries, most of them offset and byte-stream bookkeeping; this pass executes 300-400 (counted in the ISA for L = 12).
// ranges (optional): what the locate pass reads for a settled query, in the generic kernel's layout -- ranges[2q] = a row
// interval's start or an RS_SINGLE / RS_MULTI word (verified text position / candidate rows + mask), ranges[2q + 1] = 0.
constexpr int AA_KMER_MIN = 8, AA_KMER_MAX = 24;
// LONG: queries of up to AA_KMER_LONG_MAX residues (peptides, protein fragments).  The pass works on a query's LAST 24
// residues exactly as above -- seed window, the residue in front of it, up to 17 residues compared in registers -- and the
// residues before those (the "far" part) are screened for bytes the reference leaves undefined when the query is loaded and
// compared with the text, eight at a time, only for candidates that passed everything else.  (The generic kernel serves a
// 40-residue batch from the text at 3.9 G queries/s; this pass at the rate of its 24-residue tail plus that comparison.)
constexpr int AA_KMER_LONG_MAX = 1024;
// symbol index of residue j of a query held as three words of one index per byte
__device__ __forceinline__ uint32_t jn_idx(uint64_t i0, uint64_t i1, uint64_t i2, int j) {
  const uint64_t w = j < 8 ? i0 : (j < 16 ? i1 : i2);
  return (uint32_t)((w >> (8 * (j & 7))) & 0xFF);
}
constexpr int AA_KMER_VMULTI = 4;        // seed ranges of up to this many rows are verified candidate by candidate
constexpr int AA_KMER_VMULTI_LANES = 8;  //   when at least this many lanes of the wave hold one (1 and 3 measure no better)

// RAGGED: query q is ascii[off[q], off[q + 1]) with its own length (k .. AA_KMER_MAX residues take this pass, any other
// length is listed for the generic kernel); L is then ignored.  Same per-lane work with the length, the number of
// residues left of the seed window and the byte masks as per-lane values instead of wave constants.
template <int NQ, bool RAGGED = false, bool LONG = false>
__global__ __launch_bounds__(256) void count_aa_kmer_probe_kernel(DevIndex ix, const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ off,
                                                                  uint64_t n, int L, uint64_t* __restrict__ counts, uint64_t* __restrict__ ranges,
                                                                  uint8_t* __restrict__ status, QueryList ql) {
  // per byte: bits 0..4 symbol index, bits 8..12 digit of the seed-table index (the 21 searchable symbols), bit 15
  // undefined in the reference ('$', '#', bytes >= 0x80)
  __shared__ uint16_t lut[256];
  __shared__ unsigned int s_count;
  {
    const int c = threadIdx.x;
    const int idx = c >= 128 ? 0 : index_of_ascii(AMINO, (uint8_t)c);
    const int digit = aa_letter_of_index(idx);
    lut[c] = (uint16_t)(idx <= 0 ? 0x8000 : (idx | (digit < 0 ? 0x4000 : digit << 8)));
  }
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int k = ix.seed_k;
  const SeedEntry* __restrict__ seed = ix.seed;
  const bool pos = ix.seed_pos && ix.text8 && ix.dense_sa && ix.dense_ratio == 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t region = (uint64_t)blockIdx.x * ql.cap;
  auto bytes_mask = [](int m) { return m >= 8 ? ~0ull : (m <= 0 ? 0ull : (1ull << (8 * m)) - 1); };
  auto ld8 = [](const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; };
  // LONG: does a word hold a byte >= 0x80, a '$' or a '#' (the bytes whose symbol index is not positive)?
  auto undefined8 = [](uint64_t x) {
    constexpr uint64_t K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
    const uint64_t t1 = x ^ 0x2424242424242424ull, t2 = x ^ 0x2323232323232323ull;
    return (x & K80) | (((((t1 & K7F) + K7F) | t1) & K80) ^ K80) | (((((t2 & K7F) + K7F) | t2) & K80) ^ K80);
  };
  // LONG: are the `far` residues at qp (ASCII) the symbols at text8[tpos, tpos + far)?
  auto far_equal = [&](uint64_t tpos, const uint8_t* qp, int far) {
    for (int w0 = 0; w0 < far; w0 += 8) {
      const uint64_t qc = ld8(qp + w0), tw = ld8(ix.text8 + tpos + (uint64_t)w0);  // (the query's 24-residue tail follows: in bounds)
      uint64_t iw = 0;
#pragma unroll
      for (int bj = 0; bj < 8; bj++) iw |= (uint64_t)(lut[(qc >> (8 * bj)) & 0xFF] & 0x1Fu) << (8 * bj);
      const int nb = far - w0;
      if ((tw ^ iw) & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1)) return false;
    }
    return true;
  };
  // which of the candidates at text positions p[0 .. nc) have the query's first `rem` residues in front of them (bit c)
  auto candidates = [&](const uint32_t (&p)[AA_KMER_VMULTI], uint32_t nc, uint64_t j0, uint64_t j1, uint64_t j2, int rem) {
    const uint64_t m0 = bytes_mask(rem), m1 = bytes_mask(rem - 8), m2 = bytes_mask(rem - 16);
    uint32_t found = 0;
#pragma unroll
    for (int c = 0; c < AA_KMER_VMULTI; c++) {
      if ((uint32_t)c >= nc || p[c] < (uint32_t)rem) continue;  // (the suffix starts too close to the text's beginning)
      const uint8_t* t = ix.text8 + ((uint64_t)p[c] - (uint64_t)rem);
      uint64_t d = (ld8(t) ^ j0) & m0;
      if (rem > 8) d |= (ld8(t + 8) ^ j1) & m1;
      if (rem > 16) d |= (ld8(t + 16) ^ j2) & m2;
      found |= d ? 0u : 1u << c;
    }
    return found;
  };
  // the trip count is wave-uniform (ballots and the wave-level atomic below need every lane of the wave)
  for (uint64_t wbase = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); wbase < n; wbase += NQ * stride) {
    uint64_t qv[NQ], c0[NQ], c1[NQ], c2[NQ];
    int Lq[NQ];      // residues of the query this pass holds in registers (LONG: its last 24; RAGGED: per query)
    int far[NQ];     // LONG: residues in front of those
    const uint8_t* qp[NQ];  // LONG: the query's first byte
    bool odd[NQ];    // RAGGED: length outside k .. AA_KMER_MAX (LONG: AA_KMER_LONG_MAX)
#pragma unroll
    for (int h = 0; h < NQ; h++) {  // bytes [0, 8), [8, 16), [16, 24) of the query (bytes past its end are ignored below)
      qv[h] = wbase + lane + (uint64_t)h * stride;
      c0[h] = c1[h] = c2[h] = 0;
      Lq[h] = LONG && L > AA_KMER_MAX ? AA_KMER_MAX : L;
      far[h] = 0;
      qp[h] = ascii;
      odd[h] = false;
      if (qv[h] < n) {
        if (RAGGED) {
          const uint64_t b = off[qv[h]], len = off[qv[h] + 1] - b;
          odd[h] = len < (uint64_t)(k > 1 ? k : 1) || len > (uint64_t)(LONG ? AA_KMER_LONG_MAX : AA_KMER_MAX);
          Lq[h] = odd[h] ? AA_KMER_MAX : (len > (uint64_t)AA_KMER_MAX ? AA_KMER_MAX : (int)len);
          if (!odd[h]) {  // never reads a byte past the query's last one (a caller's buffer may end right there)
            if (LONG) { far[h] = (int)len - Lq[h]; qp[h] = ascii + b; }
            const uint64_t first = b + (LONG ? (uint64_t)far[h] : 0ull);
            const uint8_t* p = ascii + first;
            const int Lt = Lq[h];
            if (Lt >= 8) {  // the last word is anchored at the query's end, as in the equal-length branch
              c0[h] = ld8(p);
              const uint64_t last = ld8(p + Lt - 8);
              if (Lt >= 16) { c1[h] = ld8(p + 8); if (Lt > 16) c2[h] = last >> (8 * (24 - Lt)); }
              else if (Lt > 8) c1[h] = last >> (8 * (16 - Lt));
            } else if (first + (uint64_t)Lt >= 8) {  // shorter than a word: the word that ENDS with the query
              c0[h] = ld8(p + Lt - 8) >> (8 * (8 - Lt));
            } else {  // within the buffer's first seven bytes
              for (int t = 0; t < Lt; t++) c0[h] |= (uint64_t)p[t] << (8 * t);
            }
          }
        } else {
          const int Lt = Lq[h];
          if (LONG) { far[h] = L - Lt; qp[h] = ascii + qv[h] * (uint64_t)L; }
          const uint8_t* p = ascii + qv[h] * (uint64_t)L + (LONG ? (uint64_t)far[h] : 0ull);
          c0[h] = ld8(p);
          if (Lt > 8) {
            const uint64_t last = ld8(p + Lt - 8);  // never reads past the query
            if (Lt >= 16) { c1[h] = ld8(p + 8); if (Lt > 16) c2[h] = last >> (8 * (24 - Lt)); }
            else c1[h] = last >> (8 * (16 - Lt));
          }
        }
      }
    }
    uint64_t i0[NQ], i1[NQ], i2[NQ];  // the same bytes as symbol indices
    uint32_t flags[NQ];
    SeedEntry ev[NQ];
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      const int len = Lq[h], rem = len - k;
      uint32_t slot = 0, mul = 1, fl = odd[h] ? 0x4000u : 0u;  // 21^7 < 2^32
      auto word = [&](uint64_t c, int base) {
        uint64_t iw = 0;
#pragma unroll
        for (int bj = 0; bj < 8; bj++) {
          const int j = base + bj;
          if (j < len) {
            const uint32_t t = lut[(c >> (8 * bj)) & 0xFF];
            fl |= t & 0x8000u;
            iw |= (uint64_t)(t & 0x1Fu) << (8 * bj);
            if (j >= rem) {  // seed window: leftmost residue least significant
              fl |= t & 0x4000u;
              slot += ((t >> 8) & 0x1Fu) * mul;
              mul *= (uint32_t)AA_SEED_SIGMA;
            }
          }
        }
        return iw;
      };
      i0[h] = word(c0[h], 0);
      i1[h] = word(c1[h], 8);
      i2[h] = word(c2[h], 16);
      if (LONG && qv[h] < n && !odd[h]) {  // the far residues: any byte the reference leaves undefined sends the query to the generic kernel
        uint64_t und = 0;
        for (int w0 = 0; w0 < far[h]; w0 += 8) {
          uint64_t x = ld8(qp[h] + w0);
          if (far[h] - w0 < 8) x &= (1ull << (8 * (far[h] - w0))) - 1;
          und |= undefined8(x);
        }
        if (und) fl |= 0x8000u;
      }
      flags[h] = fl;
      ev[h] = SeedEntry{1u, 0u};
      if (qv[h] < n && !fl) ev[h] = seed_probe(seed + slot);
      if (ql.tally) { const uint64_t pm = __ballot(qv[h] < n && !fl); if (lane == 0) tally_add(ql.tally, 0, (unsigned long long)__popcll(pm)); }
    }
    bool listed[NQ], vfy[NQ], multi[NQ];
    uint64_t value[NQ], rs[NQ], t0[NQ], t1[NQ], t2[NQ];  // rs: what the locate pass reads for the query (ranges[2q])
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      const SeedEntry e = ev[h];
      const uint32_t scnt = aa_seed_cnt(e);
      const int rem = Lq[h] - k;
      listed[h] = vfy[h] = multi[h] = false;
      value[h] = 0;
      rs[h] = (RS_PLAIN << RS_MODE_SHIFT) | 1ull;  // no hits
      t0[h] = t1[h] = t2[h] = 0;
      if (qv[h] >= n) continue;
      if (flags[h]) listed[h] = true;
      else if (scnt == 0u) value[h] = 0;
      else if (rem == 0) {  // the seed window is the whole query: the entry is the answer
        if (scnt == AA_SEED_CNT_SAT) listed[h] = true;
        else { value[h] = scnt; rs[h] = scnt == 1u && ix.seed_pos ? ((RS_SINGLE << RS_MODE_SHIFT) | e.sp) : ((RS_PLAIN << RS_MODE_SHIFT) | e.sp); }
      }
      else if (scnt == 1u) {
        // the residue in front of the seed window must be BWT[row]
        if (jn_idx(i0[h], i1[h], i2[h], rem - 1) != aa_seed_sym(e)) value[h] = 0;
        else if (aa_seed_is_ctx(e) && rem <= AA_SEED_CTX_LEN) {
          // the entry holds the residues in front of the one occurrence: decided here, no text access
          uint32_t qctx = 0;  // query residues rem-2, rem-3, ... 0 in the entry's order (rem <= 6: all in bytes 0..7)
#pragma unroll
          for (int j = 0; j < AA_SEED_CTX_LEN - 1; j++)
            if (j < rem - 1) qctx |= (uint32_t)((i0[h] >> (8 * (rem - 2 - j))) & 0x1Fu) << (5 * j);
          const uint32_t cmask = rem >= 2 ? (1u << (5 * (rem - 1))) - 1u : 0u;
          if ((aa_seed_ctx(e) & cmask) == qctx) { value[h] = 1; rs[h] = (RS_SINGLE << RS_MODE_SHIFT) | ((uint64_t)e.sp - (uint64_t)rem); }
        }
        else if (pos) {
          if (e.sp >= (uint32_t)(rem + (LONG ? far[h] : 0))) {  // else the suffix starts too close to the text's beginning
            vfy[h] = true;  // the window's loads are issued here, for all NQ queries, and compared below
            const uint8_t* t = ix.text8 + ((uint64_t)e.sp - (uint64_t)rem);
            t0[h] = ld8(t);
            if (rem > 8) t1[h] = ld8(t + 8);
            if (rem > 16) t2[h] = ld8(t + 16);
          }
        } else listed[h] = true;
      } else if (aa_seed_is_multi(e) && !((aa_seed_mask(e) >> (jn_idx(i0[h], i1[h], i2[h], rem - 1) & 0x1Fu)) & 1u)) {
        value[h] = 0;  // the residue in front of the seed window does not occur in the BWT over the entry's rows: absent
      } else if (pos && scnt <= (uint32_t)AA_KMER_VMULTI) multi[h] = true;
      else listed[h] = true;
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      const int rem = Lq[h] - k;
      if (vfy[h]) {
        value[h] = (((t0[h] ^ i0[h]) & bytes_mask(rem)) | ((t1[h] ^ i1[h]) & bytes_mask(rem - 8)) | ((t2[h] ^ i2[h]) & bytes_mask(rem - 16))) ? 0ull : 1ull;
        if (LONG && value[h] && far[h] > 0 && !far_equal((uint64_t)ev[h].sp - (uint64_t)rem - (uint64_t)far[h], qp[h], far[h])) value[h] = 0;
        if (value[h]) rs[h] = (RS_SINGLE << RS_MODE_SHIFT) | ((uint64_t)ev[h].sp - (uint64_t)rem - (uint64_t)(LONG ? far[h] : 0));
      }
      if (ql.tally) { const uint64_t vm = __ballot(vfy[h]); if (lane == 0) tally_add(ql.tally, 4, (unsigned long long)__popcll(vm)); }
      // A handful of candidate rows, neighbours in the dense SA: each is compared with the text -- two dependent loads
      // the whole wave waits for, so a wave does it only when enough of its lanes need it (a batch of k-mers from the
      // text); the odd such lane of a random batch is listed, and the second pass works through those densely.
      // (Queueing them in LDS until a wave-full is pending, as the nucleotide probe does, was measured: the work
      // moves from the second pass into this one and the sum grows by 6 %.)
      const uint64_t mm = __ballot(multi[h]);
      if (__popcll(mm) < AA_KMER_VMULTI_LANES) { listed[h] = listed[h] || multi[h]; multi[h] = false; }
      if (multi[h]) {
        const uint32_t sp = ev[h].sp, nc = aa_seed_cnt(ev[h]);
        uint32_t p[AA_KMER_VMULTI];
#pragma unroll
        for (int c = 0; c < AA_KMER_VMULTI; c++) p[c] = (uint32_t)c < nc ? ix.dense_sa[sp + c] : 0u;
        uint32_t mask = candidates(p, nc, i0[h], i1[h], i2[h], rem);
        if (LONG && far[h] > 0) {
#pragma unroll
          for (int c = 0; c < AA_KMER_VMULTI; c++)
            if ((mask >> c) & 1u)
              if (p[c] < (uint32_t)(rem + far[h]) || !far_equal((uint64_t)p[c] - (uint64_t)rem - (uint64_t)far[h], qp[h], far[h])) mask &= ~(1u << c);
        }
        if (ql.tally) { tally_add(ql.tally, 3, nc); tally_add(ql.tally, 4, nc); }
        value[h] = (uint64_t)__popc(mask);
        rs[h] = (RS_MULTI << RS_MODE_SHIFT) | (uint64_t)sp | ((uint64_t)(rem + (LONG ? far[h] : 0)) << 32) | ((uint64_t)mask << 48);
      }
      if (qv[h] < n && !listed[h]) {
        counts[qv[h]] = value[h];
        if (ranges) { ranges[2 * qv[h]] = rs[h]; ranges[2 * qv[h] + 1] = 0; }
        if (status) status[qv[h]] = Q_OK;
      }
      const uint64_t lm = __ballot(listed[h]);
      if (lm) {
        unsigned int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&s_count, (unsigned int)__popcll(lm));
        slot0 = __shfl(slot0, 0, 64);
        if (listed[h]) ql.q[region + slot0 + (uint64_t)__popcll(lm & lane_lt)] = (uint32_t)qv[h];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) ql.count[blockIdx.x] = s_count;
}

// measurement aid: device-to-device copy, 16 bytes per lane per step (the streaming rate the roofline object prints next to
// the nominal HBM peak)
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const v4u* __restrict__ s4 = reinterpret_cast<const v4u*>(src);
  v4u* __restrict__ d4 = reinterpret_cast<v4u*>(dst);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(&s4[i]), &d4[i]);
}

// profiling aid: an empty kernel whose grid size names a phase of a benchmark run, so that the per-dispatch rows of a
// rocprofv3 counter pass (which cannot be combined with marker tracing on this pool) can be cut into those phases
__global__ void phase_marker_kernel() {}

// counts as 32-bit words for the trip over PCIe (host-packed paths: a count is < bwt_len < 2^32 there)
__global__ __launch_bounds__(256) void narrow_counts_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint32_t)in[i];
}

// the lowest query of a chunk that the generic kernel rejected, as (index << 8 | status), or ~0: eight bytes cross PCIe
// instead of one status byte per query
__global__ __launch_bounds__(256) void status_first_bad_kernel(const uint8_t* __restrict__ status, uint64_t n, unsigned long long* __restrict__ first_bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long best = ~0ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (status[i] != Q_OK) { const unsigned long long v = ((unsigned long long)i << 8) | status[i]; best = v < best ? v : best; }
  if (best != ~0ull) atomicMin(first_bad, best);
}

// one step / one backstep / one initial range for the scalar conveniences of the C ABI
template <int A>
__global__ void scalar_ops_kernel(DevIndex ix, int op, uint64_t a, uint64_t b, int idx, uint64_t* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (op == 0) {  // update_range_with_symbol
    uint64_t sp = a, ep = b;
    step_scalar<A>(ix, sp, ep, idx);
    out[0] = sp; out[1] = ep;
  } else if (op == 1) {  // backstep
    out[0] = backstep_scalar<A>(ix, a);
  } else if (op == 2) {  // global_occurrence
    out[0] = rank_scalar<A>(ix, a, idx);
  } else {  // symbol_at
    out[0] = (uint64_t)symbol_at<A>(ix, a);
  }
}

// ------------------------------------------------------------------------------------------------
// the reference's k-mer table content, for byte-identical .awry files (src/kmer_lookup_table.rs:121-167):
// slot = sum_j s_j * sigma^j with s_0 = LAST symbol, digits restricted to 1..sigma-1; steps are applied
// without any emptiness check; every other slot stays SearchRange::zero() = {1, 0}.
// ------------------------------------------------------------------------------------------------
template <int A>
__global__ __launch_bounds__(256) void ref_kmer_table_kernel(DevIndex ix, int kmer_len, uint64_t nslots,
                                                             uint64_t* __restrict__ table) {
  const uint64_t sigma = A == NUCLEOTIDE ? 4 : 20;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < nslots; slot += stride) {
    uint64_t sp = 1, ep = 0, rem = slot;
    bool populated = kmer_len > 0;
    for (int j = 0; j < kmer_len; j++) {
      if (rem % sigma == 0) populated = false;
      rem /= sigma;
    }
    if (populated) {
      rem = slot;
      int idx = (int)(rem % sigma);
      rem /= sigma;
      sp = ix.prefix_sums[idx];
      ep = ix.prefix_sums[idx + 1] - 1;
      for (int j = 1; j < kmer_len; j++) {
        idx = (int)(rem % sigma);
        rem /= sigma;
        step_scalar<A>(ix, sp, ep, idx);
      }
    }
    table[2 * slot] = sp;
    table[2 * slot + 1] = ep;
  }
}

// ------------------------------------------------------------------------------------------------
// locate: one hit per lane.  hit h belongs to the query q with hit_off[q] <= h < hit_off[q+1]; its BWT
// row is ranges[2q] + (h - hit_off[q]) -- ascending row order inside a query, src/fm_index.rs:521.
// ------------------------------------------------------------------------------------------------
template <int A>
__global__ __launch_bounds__(256) void locate_scalar_kernel(DevIndex ix, const uint64_t* __restrict__ ranges,
                                                            const uint64_t* __restrict__ hit_off, uint64_t n,
                                                            uint64_t total, uint64_t* __restrict__ gpos,
                                                            uint64_t* __restrict__ pos /* (seq_idx, local) pairs */) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < total; h += stride) {
    uint64_t lo = 0, hi = n;  // largest q with hit_off[q] <= h
    while (hi - lo > 1) {
      uint64_t mid = (lo + hi) >> 1;
      if (hit_off[mid] <= h) lo = mid; else hi = mid;
    }
    uint64_t row = ranges[2 * lo] + (h - hit_off[lo]);
    uint64_t steps = 0;
    while (row % ix.sa_ratio != 0) {  // position_is_sampled, src/compressed_suffix_array.rs:109-111
      row = backstep_scalar<A>(ix, row);
      steps++;
    }
    const uint64_t g = (sa_sample(ix, row / ix.sa_ratio) + steps) % ix.bwt_len;  // src/fm_index.rs:534
    gpos[h] = g;
    if (pos) {
      uint64_t a = 0, z = ix.nseq;  // largest i with seq_starts[i] <= g
      while (z - a > 1) {
        uint64_t mid = (a + z) >> 1;
        if (ix.seq_starts[mid] <= g) a = mid; else z = mid;
      }
      pos[2 * h] = a;
      pos[2 * h + 1] = g - (ix.nseq ? ix.seq_starts[a] : 0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of u64 counts (locate's CSR offsets): per-tile sums, scan of tile sums, fix-up
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_TILE = 2048;  // elements per 256-thread block

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, total in *tot
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* tot) {
  __shared__ uint64_t wsum[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint64_t inc = wave_incl_scan(v);
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint64_t base = 0, t = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i < wv) base += wsum[i];
    t += wsum[i];
  }
  __syncthreads();
  *tot = t;
  return base + inc - v;
}

__global__ __launch_bounds__(256) void scan_tile_sums_kernel(const uint64_t* __restrict__ in, uint64_t n,
                                                             uint64_t* __restrict__ tile_sums) {
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
  uint64_t s = 0;
  for (int j = 0; j < SCAN_TILE / 256; j++) {
    uint64_t i = base + (uint64_t)j * 256 + threadIdx.x;
    if (i < n) s += in[i];
  }
  uint64_t tot;
  block_excl_scan(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of tile sums in place; writes the grand total to *total
__global__ __launch_bounds__(256) void scan_tile_offsets_kernel(uint64_t* __restrict__ tile_sums, uint64_t ntiles,
                                                                uint64_t* __restrict__ total) {
  uint64_t carry = 0;
  for (uint64_t b = 0; b < ntiles; b += 256) {
    uint64_t i = b + threadIdx.x;
    uint64_t v = i < ntiles ? tile_sums[i] : 0, tot;
    uint64_t ex = block_excl_scan(v, &tot);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

// out has n + 1 entries; out[n] = grand total
__global__ __launch_bounds__(256) void scan_apply_kernel(const uint64_t* __restrict__ in, uint64_t n,
                                                         const uint64_t* __restrict__ tile_offs,
                                                         uint64_t* __restrict__ out) {
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
  uint64_t carry = tile_offs[blockIdx.x];
  for (int j = 0; j < SCAN_TILE / 256; j++) {
    uint64_t i = base + (uint64_t)j * 256 + threadIdx.x;
    uint64_t v = i < n ? in[i] : 0, tot;
    uint64_t ex = block_excl_scan(v, &tot);
    if (i < n) out[i] = carry + ex;
    if (i == n - 1) out[n] = carry + ex + v;
    carry += tot;
  }
}

// ------------------------------------------------------------------------------------------------
// quad-cooperative nucleotide path (bwt_len < 2^32): packed 2-bit k-mers, seed table, persistent quads
// ------------------------------------------------------------------------------------------------

// sum over the 4 lanes of a quad; every lane receives the total (quad_perm DPP, no LDS)
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  return v;
}

// 2-bit letter (A0 C1 G2 T3) -> per-plane XOR masks of its 3-bit code (A110 C101 G011 T001)
struct NtXor { uint64_t x0, x1, x2; };
__device__ __forceinline__ NtXor nt_xor_of_letter(uint32_t c) {
  const uint32_t code = (0x1356u >> (4 * c)) & 7u;  // nibbles: A=6, C=5, G=3, T=1
  NtXor r;
  r.x0 = (code & 1u) ? 0ull : ~0ull;
  r.x1 = (code & 2u) ? 0ull : ~0ull;
  r.x2 = (code & 4u) ? 0ull : ~0ull;
  return r;
}

struct QuadBlock { ulonglong2 lo, hi; };  // lane l: lo = {plane0[l], plane1[l]}, hi = {plane2[l], milestone[l]}

__device__ __forceinline__ QuadBlock quad_load(const uint64_t* __restrict__ blocks, uint32_t b, int l) {
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(blocks + (uint64_t)b * NT_BLOCK_WORDS);
  QuadBlock q;
  q.lo = p[l];      // bytes [16 l, 16 l + 16) of the first half line
  q.hi = p[4 + l];  // bytes [64 + 16 l, ...) of the second half line
  return q;
}

// this lane's share of C-free rank(row, letter c): popcount of its slice + the milestone if it owns it
__device__ __forceinline__ uint32_t quad_rank_part(const QuadBlock& d, const NtXor& x, uint32_t row, uint32_t c, int l) {
  const uint64_t pred = (d.lo.x ^ x.x0) & (d.lo.y ^ x.x1) & (d.hi.x ^ x.x2);
  const uint32_t cnt = (uint32_t)__popcll(pred & slice_mask((int)(row & 255u) - 64 * l));
  return cnt + ((uint32_t)l == c ? (uint32_t)d.hi.y : 0u);
}

// one backward-search step for the quad's query: [sp, ep] -> [sp', ep'] with letter c (src/fm_index.rs:559-582)
__device__ __forceinline__ void quad_step(const uint64_t* __restrict__ blocks, uint32_t cl, uint32_t& sp, uint32_t& ep,
                                          uint32_t c, int l) {
  const uint32_t r0 = sp - 1, r1 = ep;
  const uint32_t b0 = r0 >> 8, b1 = r1 >> 8;
  QuadBlock d0 = quad_load(blocks, b0, l);
  QuadBlock d1 = d0;
  if (b1 != b0) d1 = quad_load(blocks, b1, l);  // most steps rank both rows in one block
  const NtXor x = nt_xor_of_letter(c);
  const uint32_t v0 = quad_sum(quad_rank_part(d0, x, r0, c, l));
  const uint32_t v1 = quad_sum(quad_rank_part(d1, x, r1, c, l));
  sp = cl + v0;
  ep = cl + v1 - 1;
}

// Count fixed-length packed k-mers.  Query word: letter j (0 = leftmost) in bits [2j, 2j+2).
// Every quad walks its own strided list of queries (q = quad id, += number of quads) as a small state
// machine: one random HBM access group (a seed probe or the block(s) of one step) per loop iteration, so
// quads that finish early immediately start their next query instead of idling behind slower ones.
// TALLY adds the work census the roofline figure is computed from: tally[0] += seed probes,
// tally[1] += executed steps, tally[2] += distinct BWT blocks ranked (1 or 2 per step), SURVEY.md 8(d).
template <bool USE_SEED, bool TALLY>
__global__ __launch_bounds__(256) void count_nt2_quad_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                             uint64_t* __restrict__ counts, unsigned long long* __restrict__ tally) {
  const int l = threadIdx.x & 3;
  const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint64_t* __restrict__ blocks = ix.blocks;
  const SeedEntry* __restrict__ seed = ix.seed;
  const int k = USE_SEED ? ix.seed_k : 1;
  const uint32_t cA = (uint32_t)ix.prefix_sums[1], cC = (uint32_t)ix.prefix_sums[2], cG = (uint32_t)ix.prefix_sums[3],
                 cN = (uint32_t)ix.prefix_sums[4], cT = (uint32_t)ix.prefix_sums[5], cEnd = (uint32_t)ix.prefix_sums[6];

  bool have = q < n;
  uint64_t w = have ? queries[q] : 0;
  bool fresh = true;  // next access is the seed probe / initial range of query q
  uint32_t sp = 1, ep = 0;
  int i = 0;  // symbols still to consume (the next one is letter i-1)
  uint32_t t_probe = 0, t_step = 0, t_blk = 0;

  while (__any(have)) {
    if (have) {
      if (fresh) {
        if (USE_SEED) {
          const uint64_t sidx = ((w >> (2 * (L - k))) & ((1ull << (2 * k)) - 1));
          const SeedEntry e = seed[sidx];
          const uint32_t scnt = seed_cnt(e);
          sp = scnt ? e.sp : 1u;
          ep = scnt ? e.sp + scnt - 1u : 0u;
          i = L - k;
          if (scnt == 1u && i > 0) {  // singleton: it survives the next step only if BWT[sp] is the next letter
            const uint32_t nc = (uint32_t)(w >> (2 * (i - 1))) & 3u;
            if (seed_sym(e) != (int)(nc == 3u ? 5u : nc + 1u)) { sp = 1u; ep = 0u; }
          }
          // count not representable, or a position seed (ix.seed_pos) that would have to be stepped: start without the table
          if (scnt == SEED_CNT_SAT || (ix.seed_pos && scnt == 1u && sp <= ep && i > 0)) i = -1;
          if (TALLY) t_probe++;
        }
        if (!USE_SEED || i < 0) {
          const uint32_t c = (uint32_t)(w >> (2 * (L - 1))) & 3u;  // SearchRange::new(last symbol)
          sp = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
          ep = (c == 0 ? cC : (c == 1 ? cG : (c == 2 ? cN : cEnd))) - 1;
          i = L - 1;
        }
        fresh = false;
      } else {
        i--;
        const uint32_t c = (uint32_t)(w >> (2 * i)) & 3u;
        const uint32_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
        if (TALLY) { t_step++; t_blk += ((sp - 1) >> 8) == (ep >> 8) ? 1u : 2u; }
        quad_step(blocks, cl, sp, ep, c, l);
      }
      if (sp > ep || i == 0) {
        if (l == 0) counts[q] = sp > ep ? 0ull : (uint64_t)(ep - sp) + 1ull;
        q += nquads;
        have = q < n;
        if (have) w = queries[q];
        fresh = true;
      }
    }
  }
  if (TALLY && l == 0) {
    atomicAdd(&tally[0], (unsigned long long)t_probe);
    atomicAdd(&tally[1], (unsigned long long)t_step);
    atomicAdd(&tally[2], (unsigned long long)t_blk);
  }
}

// census of the quad4 kernel: tally[5] += blocks ranked by a query's steps after its first TALLY_DEEP_STEP ones.  The
// blocks of step j of a table-less search are shared by all queries that agree on their last j letters: at most 2 * 4^j
// lines, which stay in the Infinity Cache (256 MiB) up to j = 10 -- only the deeper steps reach HBM.
constexpr int TALLY_DEEP_STEP = 10;

// Seed-and-verify switch: compare the remaining i letters with the text instead of taking i more LF steps?
// A single candidate is verified at once (2 lines: SA + text, against one line per remaining letter); a range of
// 2..8 rows first takes `after` LF steps, which usually thin it out at one line each.
__device__ __forceinline__ bool verify_now(uint32_t cnt, int i, int steps_done, int after) {
  return cnt <= 8u && (int)(3u * cnt) <= i && (cnt == 1u || steps_done >= after);
}

// 16 packed 2-bit letters (low 32 bits of x) -> 16 nibbles holding the same letters
__device__ __forceinline__ uint64_t spread_letters16(uint64_t x) {
  x &= 0xFFFFFFFFull;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  return x;
}

struct Text20 { uint32_t w[5]; };  // 5 consecutive u32 of the 4-bit text: any 32 symbols at any nibble offset

// A window of up to 32 text symbols, fetched now and compared later (so that several can be in flight per lane).
struct TextWin { Text20 t; int m, sh; };  // m symbols starting at nibble sh/4 of t
__device__ __forceinline__ TextWin text_window_load(const uint32_t* __restrict__ text4, uint64_t t0, int m) {
  TextWin w;
  w.m = m < 0 ? 0 : (m > 32 ? 32 : m);
  w.sh = 4 * (int)(t0 & 7);
  if (w.m > 0) w.t = *reinterpret_cast<const Text20*>(text4 + (t0 >> 3));
  else w.t = Text20{{0u, 0u, 0u, 0u, 0u}};
  return w;
}
// 1 = the window differs from the 32 letters of qword (its first m letters)
__device__ __forceinline__ uint32_t text_window_differs(const TextWin& w, uint64_t qword) {
  if (w.m == 0) return 0u;
  const Text20& t = w.t;
  const int sh = w.sh, m = w.m;
  const uint64_t a0 = (uint64_t)t.w[0] | ((uint64_t)t.w[1] << 32), a1 = (uint64_t)t.w[2] | ((uint64_t)t.w[3] << 32), a2 = t.w[4];
  const uint64_t lo = sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;
  const uint64_t hi = sh ? (a1 >> sh) | (a2 << (64 - sh)) : a1;
  const uint64_t qlo = spread_letters16(qword), qhi = spread_letters16(qword >> 32);
  const uint64_t mlo = m >= 16 ? ~0ull : ((1ull << (4 * m)) - 1);
  const uint64_t mhi = m <= 16 ? 0ull : (m >= 32 ? ~0ull : ((1ull << (4 * (m - 16))) - 1));
  return (((lo ^ qlo) & mlo) | ((hi ^ qhi) & mhi)) ? 1u : 0u;
}

// Does text[g + 32 j0' .. ) equal this lane's 32-letter query word?  Lane l of the quad compares window symbols
// [128 c + 32 l, +32) of a window of `len` symbols starting at text position g; returns 1 on a mismatch.
__device__ __forceinline__ uint32_t verify_part(const uint32_t* __restrict__ text4, uint64_t g, int len, int c, int l, uint64_t qword) {
  const int j0 = 128 * c + 32 * l;
  return text_window_differs(text_window_load(text4, g + (uint64_t)j0, len - j0), qword);
}

}  // namespace awry
#include "lcx.hip.h"
namespace awry {

// "quad4" variant of the hot kernel: a quad owns GROUPS of 4 consecutive queries (32 contiguous bytes in and out).
// Lane t of the quad loads query 4m+t and keeps result 4m+t, so the group is read with one 32-B request and
// written back as one whole 32-B sector; the strided kernel above writes every 8-B count on its own, which
// rocprofv3 shows as 5x write amplification (WRITE_SIZE 40 B per query) and ~0.65 extra L2 misses per query.
// VERIFY: seed-and-verify as in count_nt2_reads_kernel (one candidate at a time, the <= 31 remaining letters are
// one 16-B text window checked by lane 0); tally[3] += SA reads, tally[4] += text windows compared.
template <bool USE_SEED, bool TALLY, bool VERIFY>
__global__ __launch_bounds__(256) void count_nt2_quad4_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                              uint64_t* __restrict__ counts, unsigned long long* __restrict__ tally) {
  const int lane = threadIdx.x & 63, l = lane & 3;
  const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  const uint64_t ngroups = (n + 3) >> 2;
  uint64_t m = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;  // group index
  const uint64_t* __restrict__ blocks = ix.blocks;
  const SeedEntry* __restrict__ seed = ix.seed;
  const int k = USE_SEED ? ix.seed_k : 1;
  const int verify_after = (int)ix.verify_after;
  const uint32_t cA = (uint32_t)ix.prefix_sums[1], cC = (uint32_t)ix.prefix_sums[2], cG = (uint32_t)ix.prefix_sums[3],
                 cN = (uint32_t)ix.prefix_sums[4], cT = (uint32_t)ix.prefix_sums[5], cEnd = (uint32_t)ix.prefix_sums[6];
  const int kshift = 2 * (L - k);
  const uint64_t kmask = (1ull << (2 * k)) - 1;
  bool have = m < ngroups;
  uint64_t wq = (have && 4 * m + l < n) ? queries[4 * m + l] : 0;  // this lane's query of the group
  // the four seed probes of a group are issued together, one per lane, as soon as the group's words are there;
  // the next group's words are fetched one group ahead.  A query that the entry alone decides costs no wait.
  uint64_t eq = 0;        // this lane's seed entry (sp | cnt << 32)
  if (USE_SEED && have) { const SeedEntry e0 = seed[((wq >> kshift) & kmask)]; eq = (uint64_t)e0.sp | ((uint64_t)e0.cnt << 32); }
  uint64_t wq_next = (m + nquads < ngroups && 4 * (m + nquads) + l < n) ? queries[4 * (m + nquads) + l] : 0;
  int nvalid = have ? (int)(n - 4 * m < 4 ? n - 4 * m : 4) : 0;
  int t = 0;              // query of the group being searched
  bool fresh = true;
  uint64_t w = 0, res = 0;
  uint32_t sp = 1, ep = 0;
  int i = 0, steps_done = 0;
  int mode = 0, vj = 0;   // verify: 0 = LF steps, 1 = read SA of candidate vj, 2 = compare its text window
  uint32_t vhits = 0, vp = 0;
  uint32_t t_probe = 0, t_step = 0, t_blk = 0, t_vsa = 0, t_vtxt = 0, t_deep = 0;

  while (__any(have)) {
    if (have) {
      bool finished = false;
      uint64_t out_count = 0;
      if (!VERIFY || mode == 0) {
        if (fresh) {
          w = __shfl(wq, (lane & ~3) | t, 64);
          if (USE_SEED) {
            const uint64_t ev = __shfl(eq, (lane & ~3) | t, 64);
            const SeedEntry e{(uint32_t)ev, (uint32_t)(ev >> 32)};
            const uint32_t scnt = seed_cnt(e);
            sp = scnt ? e.sp : 1u;
            ep = scnt ? e.sp + scnt - 1u : 0u;
            i = L - k;
            if (scnt == 1u && i > 0) {  // singleton: it survives the next step only if BWT[sp] is the next letter
              const uint32_t nc = (uint32_t)(w >> (2 * (i - 1))) & 3u;
              if (seed_sym(e) != (int)(nc == 3u ? 5u : nc + 1u)) { sp = 1u; ep = 0u; }
            }
            // count not representable, or a position seed (ix.seed_pos) that would have to be stepped: start without the table
            if (scnt == SEED_CNT_SAT || (ix.seed_pos && scnt == 1u && sp <= ep && i > 0)) i = -1;
            if (TALLY) t_probe++;
          }
          if (!USE_SEED || i < 0) {
            const uint32_t c = (uint32_t)(w >> (2 * (L - 1))) & 3u;
            sp = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
            ep = (c == 0 ? cC : (c == 1 ? cG : (c == 2 ? cN : cEnd))) - 1;
            i = L - 1;
          }
          steps_done = 0;
          fresh = false;
        } else {
          i--;
          const uint32_t c = (uint32_t)(w >> (2 * i)) & 3u;
          const uint32_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
          if (TALLY) {
            const uint32_t nb = ((sp - 1) >> 8) == (ep >> 8) ? 1u : 2u;
            t_step++;
            t_blk += nb;
            if (steps_done >= TALLY_DEEP_STEP) t_deep += nb;
          }
          quad_step(blocks, cl, sp, ep, c, l);
          steps_done++;
        }
        if (sp > ep || i == 0) {
          finished = true;
          out_count = sp > ep ? 0ull : (uint64_t)(ep - sp) + 1ull;
        } else if (VERIFY) {
          const uint32_t cnt = ep - sp + 1u;
          if (verify_now(cnt, i, steps_done, verify_after)) { mode = 1; vj = 0; vhits = 0; }
        }
      } else if (mode == 1) {
        vp = ix.dense_sa[sp + (uint32_t)vj];
        if (TALLY) t_vsa++;
        if (vp >= (uint32_t)i) mode = 2;
        else vj++;
      } else {
        const uint32_t bad = quad_sum(verify_part(ix.text4, (uint64_t)vp - (uint64_t)i, i, 0, l, w));
        if (TALLY) t_vtxt++;
        if (!bad) vhits++;
        vj++;
        mode = 1;
      }
      if (VERIFY && mode == 1 && vj > (int)(ep - sp)) { finished = true; out_count = vhits; }
      if (finished) {
        if (l == t) res = out_count;
        t++;
        fresh = true;
        mode = 0;
        if (t == nvalid) {  // group finished: one 32-B store, then the next group
          if (l < nvalid) counts[4 * m + l] = res;
          m += nquads;
          have = m < ngroups;
          t = 0;
          nvalid = have ? (int)(n - 4 * m < 4 ? n - 4 * m : 4) : 0;
          wq = wq_next;
          if (USE_SEED && have) { const SeedEntry e0 = seed[((wq >> kshift) & kmask)]; eq = (uint64_t)e0.sp | ((uint64_t)e0.cnt << 32); }
          wq_next = (m + nquads < ngroups && 4 * (m + nquads) + l < n) ? queries[4 * (m + nquads) + l] : 0;
        }
      }
    }
  }
  if (TALLY && l == 0) {
    atomicAdd(&tally[0], (unsigned long long)t_probe);
    atomicAdd(&tally[1], (unsigned long long)t_step);
    atomicAdd(&tally[2], (unsigned long long)t_blk);
    atomicAdd(&tally[5], (unsigned long long)t_deep);
    if (VERIFY) {
      atomicAdd(&tally[3], (unsigned long long)t_vsa);
      atomicAdd(&tally[4], (unsigned long long)t_vtxt);
    }
  }
}

// Two-phase schedule for seeded k-mer batches.  With 4^k ~ bwt_len three quarters of all random queries are decided
// by their seed entry alone (absent k-mer, or a singleton whose BWT symbol is not the next letter), so
//   phase 1 (this kernel): one query per LANE, fully coalesced query reads and count writes, 64 independent seed
//     probes per wave instruction and two queries in flight per lane -- no dependent chain beyond query -> entry;
//     queries that need LF steps are appended (wave ballot + one atomic per wave) to a compact survivor list;
//   phase 2 (count_nt2_resume_kernel): the quad machinery on the survivors only, resuming from the probed range.
constexpr int VMULTI = 4;  // seed ranges of up to this many rows are verified candidate by candidate in phase 1
constexpr int LCX_LANE_ROWS = 4;  // buckets of the left-context index with up to this many rows are decided by a lane of phase 1
constexpr int LCX_TAIL_MAX = 16;  // more incomplete entries than this in a bucket: its queries take LF steps

struct Nt2Survivors {
  uint64_t* w;                 // query words
  uint64_t* range;             // sp | cnt << 32 as probed (cnt == SEED_CNT_SAT: restart without the table)
  uint32_t* q;                 // original query index
  uint32_t* count;             // survivors per phase-1 block (block b owns slots [b * cap, (b + 1) * cap))
  uint64_t cap;                // slots per block
  // (nullable) the length of lcx_quad_reads_kernel's LF list: the probe pass of the same launch sequence clears it
  uint32_t* lf_count = nullptr;
};

// VERIFY (dense SA + 4-bit text resident): a probed singleton whose BWT symbol matched is not handed to phase 2 but
// settled here, one candidate per LANE: SA[sp] gives its text position, the L - k letters in front of it are one
// <= 16-B text window.  Such queries wait in a wave-private LDS queue until 64 are pending, so the two dependent loads
// are always issued by full waves (a batch of k-mers that occur in the text takes this path wholesale; a random
// batch fills a queue once in a while and pays nothing otherwise).
template <bool TALLY, bool VERIFY>
__global__ __launch_bounds__(256) void count_nt2_probe_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                              uint64_t* __restrict__ counts, Nt2Survivors sv,
                                                              unsigned long long* __restrict__ tally) {
  __shared__ unsigned int s_count;  // a single device-wide list head would serialise ~150 k wave-level atomics (1.8 ms)
  constexpr int VQ = 192;                          // queue slots per wave: drained 128 at a time, two per lane
  __shared__ uint64_t s_vw[VERIFY ? 4 : 1][VQ];   // per-wave verify queue: query word,
  __shared__ uint32_t s_vsp[VERIFY ? 4 : 1][VQ];  //   candidate row,
  __shared__ uint32_t s_vq[VERIFY ? 4 : 1][VQ];   //   query index,
  __shared__ uint8_t s_vn[VERIFY ? 4 : 1][VQ];    //   number of candidate rows (1..VMULTI)
  if (threadIdx.x == 0) s_count = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && sv.lf_count) *sv.lf_count = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
  const SeedEntry* __restrict__ seed = ix.seed;
  const int k = ix.seed_k, i0 = L - k, kshift = 2 * (L - k);
  const uint64_t kmask = (1ull << (2 * k)) - 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t region = (uint64_t)blockIdx.x * sv.cap;
  const bool pos = VERIFY && ix.seed_pos;   // singleton entries hold SA[row]: no SA read, and no row to step from
  const int cx = (int)ix.ctx_extra, clen = SEED_CTX_LEN + cx;  // letters in front of the occurrence a context entry holds
  const bool verify = VERIFY && (i0 >= 3 || (pos && i0 >= 1));
  // left-context index resident: a seed range of 2+ rows is told apart by the keys of its bucket -- up to LCX_LANE_ROWS rows
  // by this lane (their keys are 32 contiguous bytes), more by the quads of phase 2 -- and takes no LF step
  const bool lcx = VERIFY && ix.lcx_key != nullptr && i0 >= 1;
  int vcount = 0;  // wave-uniform fill of this wave's queue
  uint32_t t_vsa = 0, t_vtxt = 0, t_lcx = 0;
  auto drain = [&](int base, int cnt) {  // entries [base, base + cnt) of the queue, cnt <= 128: two per lane
    uint64_t w[2];
    uint32_t q[2], sp[2], nc[2], vp[2];
    bool on[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int s = base + lane + 64 * h;
      on[h] = lane + 64 * h < cnt;
      w[h] = on[h] ? s_vw[VERIFY ? wv_id : 0][s] : 0;
      q[h] = on[h] ? s_vq[VERIFY ? wv_id : 0][s] : 0;
      sp[h] = on[h] ? s_vsp[VERIFY ? wv_id : 0][s] : 0;
      nc[h] = on[h] ? s_vn[VERIFY ? wv_id : 0][s] : 0;
      vp[h] = sp[h];  // position seed: the entry is the candidate's text position already
      if (on[h] && !(nc[h] & 0x80u) && !(pos && nc[h] == 1u)) { vp[h] = ix.dense_sa[sp[h]]; if (TALLY) t_vsa++; }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (!on[h]) continue;
      uint64_t value = 0;
      if (nc[h] & 0x80u) {  // a bucket of the left-context index: the rows whose key starts with the query's i0 letters
        const uint64_t want = (w[h] & ((1ull << (2 * i0)) - 1)) << (64 - 2 * i0);
        const uint32_t rows = nc[h] & 0x7Fu;
        const uint64_t* __restrict__ kp = ix.lcx_key + sp[h];
        const uint64_t k0 = kp[0], k1 = kp[1], k2 = rows > 2u ? kp[2] : ~want, k3 = rows > 3u ? kp[3] : ~want;  // (2+ rows; one line, mostly)
        const int sh = 64 - 2 * i0;
        value = (((k0 ^ want) >> sh) == 0) + (((k1 ^ want) >> sh) == 0) + (((k2 ^ want) >> sh) == 0) + (((k3 ^ want) >> sh) == 0);
        if (TALLY) t_lcx++;
        counts[q[h]] = value;
        continue;
      }
      for (uint32_t c = 0; c < nc[h]; c++) {  // the rows of a range are neighbours in the dense SA: mostly one line
        const uint32_t p = c ? ix.dense_sa[sp[h] + c] : vp[h];
        if (TALLY && c) t_vsa++;
        if (p >= (uint32_t)i0) {  // else the suffix starts too close to the text's beginning
          if (TALLY) t_vtxt++;
          value += verify_part(ix.text4, (uint64_t)p - (uint64_t)i0, i0, 0, 0, w[h]) ? 0ull : 1ull;
        }
      }
      counts[q[h]] = value;
    }
  };
  // the trip count is wave-uniform (ballots and the wave-level atomic below need every lane of the wave)
  constexpr int NQ = 4;  // queries in flight per lane: all NQ words, then all NQ seed probes, are issued before any is used
  for (uint64_t wbase = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); wbase < n; wbase += NQ * stride) {
    uint64_t qv[NQ], wv[NQ];
    SeedEntry ev[NQ];
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      qv[h] = wbase + lane + (uint64_t)h * stride;
      wv[h] = qv[h] < n ? queries[qv[h]] : 0;
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      ev[h] = SeedEntry{1u, 0u};
      if (qv[h] < n) ev[h] = seed_probe(seed + ((wv[h] >> kshift) & kmask));
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      const bool valid = qv[h] < n;
      const uint64_t q = qv[h], w = wv[h];
      const SeedEntry e = ev[h];
      const uint32_t cnt = seed_cnt(e);
      bool survivor = false, queued = false, lcx_q = false;
      uint64_t value = 0;
      if (valid) {
        if (cnt == SEED_CNT_SAT) survivor = true;
        else if (cnt == 0u) value = 0;
        else if (i0 == 0) value = cnt;
        else if (cnt == 1u && pos && seed_has_ctx(e) && i0 <= clen) {
          // the entry holds the letters in front of the one occurrence: decided here, no text access
          value = (w & ((1ull << (2 * i0)) - 1)) == (seed_full_ctx(e, cx) >> (2 * (clen - i0))) ? 1ull : 0ull;
        } else if (cnt == 1u && pos && seed_has_ctx(e)) {
          // more letters than the entry holds: those nearest the seed window must agree before the text is asked
          queued = ((w >> (2 * (i0 - clen))) & ((1ull << (2 * clen)) - 1)) == seed_full_ctx(e, cx);
        } else if (cnt == 1u) {
          const uint32_t nc = (uint32_t)(w >> (2 * (i0 - 1))) & 3u;
          survivor = seed_sym(e) == (int)(nc == 3u ? 5u : nc + 1u);  // else BWT[sp] is not the next letter: absent
          if (verify && survivor) { queued = true; survivor = false; }
        } else if (lcx && !(e.cnt & (SEED_LCX_NONE | SEED_LCX_TAIL)) && cnt <= (uint32_t)LCX_LANE_ROWS) {
          queued = true;  // the keys of the bucket's few rows decide (one line, no SA, no text)
          lcx_q = true;
        } else if (lcx && !(e.cnt & SEED_LCX_NONE)) {
          survivor = true;  // phase 2 searches the bucket's keys
        } else if (verify && cnt <= (uint32_t)VMULTI && (int)(3u * cnt) <= i0 + 2) {  // (+ 2: two rows with 4 or 5 letters left are cheaper here than as survivors)
          queued = true;  // a handful of candidate rows: each is checked against the text here, none goes to phase 2
        } else survivor = true;
        if (!queued) counts[q] = value;  // coalesced; survivors are overwritten by phase 2
      }
      if (VERIFY) {
        const uint64_t qm = __ballot(queued);
        if (qm) {
          if (queued) {
            const int s = vcount + (int)__popcll(qm & lane_lt);
            s_vw[wv_id][s] = w;
            s_vsp[wv_id][s] = cnt == 1u && pos ? seed_position(e, cx) : e.sp;
            s_vq[wv_id][s] = (uint32_t)q;
            s_vn[wv_id][s] = (uint8_t)(cnt | (lcx_q ? 0x80u : 0u));
          }
          vcount += (int)__popcll(qm);
          __builtin_amdgcn_wave_barrier();
          if (vcount >= 128) {
            vcount -= 128;
            drain(vcount, 128);
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      const uint64_t sm = __ballot(survivor);
      if (sm) {
        unsigned int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&s_count, (unsigned int)__popcll(sm));
        slot0 = __shfl(slot0, 0, 64);
        if (survivor) {
          const uint64_t s = region + slot0 + (uint64_t)__popcll(sm & lane_lt);
          sv.w[s] = w;
          sv.range[s] = (uint64_t)e.sp | ((uint64_t)(cnt | (e.cnt & (SEED_LCX_NONE | SEED_LCX_TAIL))) << 32);  // (flags of a 2+ row entry)
          sv.q[s] = (uint32_t)q;
        }
      }
    }
  }
  if (VERIFY && vcount > 0) drain(0, vcount);
  __syncthreads();
  if (threadIdx.x == 0) sv.count[blockIdx.x] = s_count;
  if (TALLY) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&tally[0], (unsigned long long)n);
    if (VERIFY && (t_vsa | t_vtxt | t_lcx)) {
      atomicAdd(&tally[3], (unsigned long long)t_vsa);
      atomicAdd(&tally[4], (unsigned long long)t_vtxt);
      atomicAdd(&tally[6], (unsigned long long)t_lcx);
    }
  }
}

// same grid as phase 1: block b resumes the survivors block b recorded.  VERIFY: seed-and-verify for the survivors
// (batches of k-mers that really occur in the text survive phase 1 wholesale; comparing their <= 31 remaining letters
// with the text costs ~2 lines per candidate instead of one line per letter).  Random batches barely reach this kernel,
// so the extra state costs them nothing -- which is why the k-mer path can keep verify on by default.
// (the body of count_nt2_resume_kernel as a block-level function, with the set of list positions a quad walks as parameters)
template <bool TALLY, bool VERIFY>
__device__ __forceinline__ void resume_block_list(const DevIndex& ix, const Nt2Survivors& sv, uint64_t region, uint64_t ns, int L,
                                                  uint64_t* __restrict__ counts, unsigned long long* __restrict__ tally, bool allow_lcx,
                                                  uint64_t r_start, uint64_t r_stride) {
  const int l = threadIdx.x & 3;
  uint64_t r = r_start;  // the quads that walk this list: r_start, r_start + r_stride, ...
  const uint64_t* __restrict__ blocks = ix.blocks;
  const int k = ix.seed_k;
  const int verify_after = (int)ix.verify_after;
  const bool lcx = VERIFY && ix.lcx_key != nullptr && allow_lcx;
  const uint32_t cA = (uint32_t)ix.prefix_sums[1], cC = (uint32_t)ix.prefix_sums[2], cG = (uint32_t)ix.prefix_sums[3],
                 cN = (uint32_t)ix.prefix_sums[4], cT = (uint32_t)ix.prefix_sums[5], cEnd = (uint32_t)ix.prefix_sums[6];
  bool have = r < ns, fresh = true;
  uint64_t w = 0;
  uint32_t sp = 1, ep = 0, qidx = 0;
  int i = 0, steps_done = 0;
  // 0 = LF steps, 1 = read SA of candidate vj, 2 = compare its text window (seed-and-verify);
  // left-context index: 3 = read the number of incomplete entries, 4 = search the keys, 5 = text position of incomplete
  // entry vj, 6 = compare it with the text
  int mode = 0, vj = 0;
  uint32_t vhits = 0, vp = 0;
  // The search of a bucket's keys keeps its state in the variables the LF modes do not use meanwhile -- this kernel runs
  // as many waves as its registers allow, and a search is a chain of dependent loads: lower bound [vp, vhits), upper bound
  // [ub_a, ub_b), level vj, incomplete rows of the bucket steps_done; ep stays the bucket's last row.
  uint32_t ub_a = 0, ub_b = 0;
  const LcxRefs lq{vp, vhits, ub_a, ub_b, vj};
  uint32_t t_step = 0, t_blk = 0, t_vsa = 0, t_vtxt = 0, t_lcx = 0, t_rp = 0;
  // the next record of this quad's walk is asked for while the current one is searched (one dependent load less per survivor)
  uint64_t nx_w = 0, nx_rg = 0;
  uint32_t nx_q = 0;
  if (have) { nx_w = sv.w[region + r]; nx_rg = sv.range[region + r]; nx_q = sv.q[region + r]; }
  while (__any(have)) {
    if (have) {
      bool finished = false;
      uint64_t out_count = 0;
      if (mode == 0) {
        if (fresh) {  // the record replaces the seed probe; the first step follows in the same iteration
          w = nx_w;
          const uint64_t rg = nx_rg;
          qidx = nx_q;
          if (r + r_stride < ns) { nx_w = sv.w[region + r + r_stride]; nx_rg = sv.range[region + r + r_stride]; nx_q = sv.q[region + r + r_stride]; }
          const uint32_t cf = (uint32_t)(rg >> 32), cnt = cf & SEED_CNT_SAT;
          steps_done = 0;
          fresh = false;
          if (qidx == 0xFFFFFFFFu) {  // an empty slot of lcx_quad_reads_kernel's list: nothing to do
            sp = 1u; ep = 0u; i = 0;
          } else if (cnt == SEED_CNT_SAT) {
            const uint32_t c = (uint32_t)(w >> (2 * (L - 1))) & 3u;
            sp = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
            ep = (c == 0 ? cC : (c == 1 ? cG : (c == 2 ? cN : cEnd))) - 1;
            i = L - 1;
          } else {
            sp = (uint32_t)rg;
            ep = sp + cnt - 1u;
            i = L - k;
            if (lcx && cnt >= 2u && !(cf & SEED_LCX_NONE) && i > 0) {  // the bucket's keys tell its rows apart: no LF step
              steps_done = 0;  // (incomplete rows of the bucket)
              if (cf & SEED_LCX_TAIL) mode = 3;
              else { vp = ub_a = sp; vhits = ub_b = sp + cnt; vj = lcx_top_level(sp, cnt); mode = 4; }
            }
          }
        }
        if (mode == 0) {
          // a probed singleton (its BWT symbol already matched the next letter) goes straight to the text
          const bool skip_step = VERIFY && i > 0 && sp <= ep && verify_now(ep - sp + 1u, i, steps_done, verify_after);
          if (i > 0 && sp <= ep && !skip_step) {
            i--;
            const uint32_t c = (uint32_t)(w >> (2 * i)) & 3u;
            const uint32_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
            if (TALLY) { t_step++; t_blk += ((sp - 1) >> 8) == (ep >> 8) ? 1u : 2u; }
            quad_step(blocks, cl, sp, ep, c, l);
            steps_done++;
          }
          if (sp > ep || i == 0) {
            finished = true;
            out_count = sp > ep ? 0ull : (uint64_t)(ep - sp) + 1ull;
          } else if (VERIFY) {
            const uint32_t cnt = ep - sp + 1u;
            if (verify_now(cnt, i, steps_done, verify_after)) { mode = 1; vj = 0; vhits = 0; }
          }
        }
      } else if (mode == 1) {
        vp = ix.dense_sa[sp + (uint32_t)vj];
        if (TALLY) t_vsa++;
        if (vp >= (uint32_t)i) mode = 2;
        else vj++;
      } else if (mode == 2) {
        const uint32_t bad = quad_sum(verify_part(ix.text4, (uint64_t)vp - (uint64_t)i, i, 0, l, w));
        if (TALLY) t_vtxt++;
        if (!bad) vhits++;
        vj++;
        mode = 1;
      } else if (mode == 3) {  // the key slot of the bucket's last row holds the number of incomplete entries
        const uint32_t inc = (uint32_t)ix.lcx_key[ep];
        if (TALLY) t_lcx++;
        if (inc > (uint32_t)LCX_TAIL_MAX) { mode = 0; steps_done = 0; }  // (too many to check one by one: LF steps after all)
        else {
          const uint32_t nc = ep - sp + 1u - inc;
          steps_done = (int)inc;
          vp = ub_a = sp; vhits = ub_b = sp + nc; vj = nc ? lcx_top_level(sp, nc) : -1;
          mode = 4;
        }
      } else if (mode == 4) {
        uint64_t qlo, qhi;
        lcx_thresholds(w, i, &qlo, &qhi);
        const int lines = lcx_quad_step(ix, lq, qlo, qhi, l);
        if (TALLY) t_lcx += (uint32_t)lines;
        if (vj < 0) {
          vhits = ub_a - vp;  // the run of rows the keys select
          if (steps_done) { mode = 5; vj = 0; }
          else { finished = true; out_count = vhits; }
        }
      } else if (mode == 5) {  // incomplete entry vj: its text position
        vp = (uint32_t)ix.lcx_rowpos[ep + 1u - (uint32_t)steps_done + (uint32_t)vj];
        if (TALLY) t_rp++;
        if (vp >= (uint32_t)i) mode = 6;
        else vj++;
      } else {
        const uint32_t bad = quad_sum(verify_part(ix.text4, (uint64_t)vp - (uint64_t)i, i, 0, l, w));
        if (TALLY) t_vtxt++;
        if (!bad) vhits++;
        vj++;
        mode = 5;
      }
      if (VERIFY && mode == 1 && vj > (int)(ep - sp)) { finished = true; out_count = vhits; }
      if (mode == 5 && vj >= steps_done) { finished = true; out_count = vhits; }
      if (finished) {
        if (l == 0 && qidx != 0xFFFFFFFFu) counts[qidx] = out_count;
        r += r_stride;
        have = r < ns;
        fresh = true;
        mode = 0;
      }
    }
  }
  if (TALLY && l == 0) {
    atomicAdd(&tally[1], (unsigned long long)t_step);
    atomicAdd(&tally[2], (unsigned long long)t_blk);
    if (VERIFY) {
      atomicAdd(&tally[3], (unsigned long long)t_vsa);
      atomicAdd(&tally[4], (unsigned long long)t_vtxt);
      atomicAdd(&tally[6], (unsigned long long)t_lcx);
      atomicAdd(&tally[7], (unsigned long long)t_rp);
    }
  }
}

template <bool TALLY, bool VERIFY>
__global__ __launch_bounds__(256) void count_nt2_resume_kernel(DevIndex ix, Nt2Survivors sv, int L, uint64_t* __restrict__ counts,
                                                               unsigned long long* __restrict__ tally) {
  resume_block_list<TALLY, VERIFY>(ix, sv, (uint64_t)blockIdx.x * sv.cap, (uint64_t)sv.count[blockIdx.x], L, counts, tally, true, threadIdx.x >> 2, 64);
}
// v2 of the hot kernel: the query and result streams are staged through LDS in wave-private chunks so that
// both move as whole 128-B lines (v1 fetched one line per 8-B query word and wrote one partial line per
// 8-B result: 2 of its ~4 line requests per query).  A wave grabs a chunk of CHUNK consecutive queries with
// one atomic, loads it coalesced into LDS, hands the queries out to its 16 quads on demand (ballot + prefix
// popcount), stores each count over the query word it came from, and writes the chunk back coalesced.
constexpr int NT2_CHUNK = 256;  // queries per wave chunk: 2 KB of LDS per wave, 8 KB per 256-thread block

template <bool USE_SEED, bool TALLY>
__global__ __launch_bounds__(256) void count_nt2_chunk_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                              uint64_t* __restrict__ counts, unsigned long long* __restrict__ chunk_counter,
                                                              unsigned long long* __restrict__ tally) {
  __shared__ uint64_t lds_all[4 * NT2_CHUNK];
  const int lane = threadIdx.x & 63, l = lane & 3;
  volatile uint64_t* lds = lds_all + (threadIdx.x >> 6) * NT2_CHUNK;  // wave-private; volatile keeps the cross-lane order
  const uint64_t* __restrict__ blocks = ix.blocks;
  const SeedEntry* __restrict__ seed = ix.seed;
  const int k = USE_SEED ? ix.seed_k : 1;
  const uint32_t cA = (uint32_t)ix.prefix_sums[1], cC = (uint32_t)ix.prefix_sums[2], cG = (uint32_t)ix.prefix_sums[3],
                 cN = (uint32_t)ix.prefix_sums[4], cT = (uint32_t)ix.prefix_sums[5], cEnd = (uint32_t)ix.prefix_sums[6];
  const uint64_t nchunks = (n + NT2_CHUNK - 1) / NT2_CHUNK;
  const uint64_t quad_lt = (1ull << (lane & ~3)) - 1;  // leader lanes of the quads before this one
  uint32_t t_probe = 0, t_step = 0, t_blk = 0;

  for (;;) {
    unsigned long long c = 0;
    if (lane == 0) c = atomicAdd(chunk_counter, 1ull);
    c = __shfl(c, 0, 64);
    if (c >= nchunks) break;
    const uint64_t base = c * NT2_CHUNK;
    const int cn = (int)(n - base < (uint64_t)NT2_CHUNK ? n - base : (uint64_t)NT2_CHUNK);
    for (int j = 0; j < NT2_CHUNK / 64; j++) {
      const int s = j * 64 + lane;
      lds[s] = s < cn ? queries[base + s] : 0ull;
    }
    int cursor = 0;  // wave-uniform: next unassigned slot
    bool have = false, fresh = true;
    int slot = 0, i = 0;
    uint64_t w = 0;
    uint32_t sp = 1, ep = 0;
    for (;;) {
      // hand out queries to idle quads, in slot order
      const uint64_t needy = __ballot(!have && l == 0);
      if (!have) {
        const int idx = cursor + (int)__popcll(needy & quad_lt);
        if (idx < cn) { have = true; fresh = true; slot = idx; w = lds[idx]; }
      }
      cursor += (int)__popcll(needy);
      if (!__any(have)) break;
      if (have) {
        if (fresh) {
          if (USE_SEED) {
            const uint64_t sidx = ((w >> (2 * (L - k))) & ((1ull << (2 * k)) - 1));
            const SeedEntry e = seed[sidx];
            const uint32_t scnt = seed_cnt(e);
            sp = scnt ? e.sp : 1u;
            ep = scnt ? e.sp + scnt - 1u : 0u;
            i = L - k;
            if (scnt == 1u && i > 0) {
              const uint32_t nc = (uint32_t)(w >> (2 * (i - 1))) & 3u;
              if (seed_sym(e) != (int)(nc == 3u ? 5u : nc + 1u)) { sp = 1u; ep = 0u; }
            }
            if (scnt == SEED_CNT_SAT || (ix.seed_pos && scnt == 1u && sp <= ep && i > 0)) i = -1;
            if (TALLY) t_probe++;
          }
          if (!USE_SEED || i < 0) {
            const uint32_t ch = (uint32_t)(w >> (2 * (L - 1))) & 3u;  // SearchRange::new(last symbol)
            sp = ch == 0 ? cA : (ch == 1 ? cC : (ch == 2 ? cG : cT));
            ep = (ch == 0 ? cC : (ch == 1 ? cG : (ch == 2 ? cN : cEnd))) - 1;
            i = L - 1;
          }
          fresh = false;
        } else {
          i--;
          const uint32_t ch = (uint32_t)(w >> (2 * i)) & 3u;
          const uint32_t cl = ch == 0 ? cA : (ch == 1 ? cC : (ch == 2 ? cG : cT));
          if (TALLY) { t_step++; t_blk += ((sp - 1) >> 8) == (ep >> 8) ? 1u : 2u; }
          quad_step(blocks, cl, sp, ep, ch, l);
        }
        if (sp > ep || i == 0) {
          if (l == 0) lds[slot] = sp > ep ? 0ull : (uint64_t)(ep - sp) + 1ull;  // the count replaces the query word
          have = false;
        }
      }
    }
    for (int j = 0; j < NT2_CHUNK / 64; j++) {
      const int s = j * 64 + lane;
      if (s < cn) counts[base + s] = lds[s];
    }
  }
  if (TALLY && l == 0) {
    atomicAdd(&tally[0], (unsigned long long)t_probe);
    atomicAdd(&tally[1], (unsigned long long)t_step);
    atomicAdd(&tally[2], (unsigned long long)t_blk);
  }
}

// Seed table, level by level: entry o of level j+1 (window letters w_0..w_j, index = sum w_t 4^t with the
// LAST query symbol most significant) is one step of its parent o >> 2 with letter o & 3.
__global__ __launch_bounds__(256) void seed_level1_kernel(DevIndex ix, SeedEntry* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    const int idx = nt_index_of_letter((int)threadIdx.x);
    const uint64_t s = ix.prefix_sums[idx], e = ix.prefix_sums[idx + 1];
    out[threadIdx.x] = SeedEntry{(uint32_t)s, (uint32_t)(e - s)};
  }
}

__global__ __launch_bounds__(256) void seed_extend_kernel(DevIndex ix, const SeedEntry* __restrict__ parent,
                                                          SeedEntry* __restrict__ child, uint64_t nchild) {
  const int l = threadIdx.x & 3;
  const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  const uint32_t cA = (uint32_t)ix.prefix_sums[1], cC = (uint32_t)ix.prefix_sums[2], cG = (uint32_t)ix.prefix_sums[3],
                 cT = (uint32_t)ix.prefix_sums[5];
  for (uint64_t o = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; o < nchild; o += nquads) {
    const SeedEntry p = parent[o >> 2];
    SeedEntry r{p.sp, 0};
    if (p.cnt) {
      const uint32_t c = (uint32_t)(o & 3);
      const uint32_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
      uint32_t sp = p.sp, ep = p.sp + p.cnt - 1;
      quad_step(ix.blocks, cl, sp, ep, c, l);
      r.sp = sp;
      r.cnt = sp > ep ? 0u : ep - sp + 1u;
    }
    if (l == 0) child[o] = r;
  }
}

// last pass over the finished table: pack the BWT symbol of singleton ranges and saturate oversized counts
// (intermediate levels keep plain 32-bit counts because a child is derived from its parent's exact range)
__global__ __launch_bounds__(256) void seed_finalize_kernel(DevIndex ix, SeedEntry* __restrict__ table, uint64_t nentries) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nentries; o += stride) {
    SeedEntry e = table[o];
    if (e.cnt == 1u) e.cnt = 1u | ((uint32_t)symbol_at<NUCLEOTIDE>(ix, e.sp) << 29);
    else if (e.cnt >= SEED_CNT_SAT) e.cnt = SEED_CNT_SAT;
    else continue;
    table[o] = e;
  }
}

// Position seeds (DevIndex::seed_pos): every singleton entry's row is replaced by the text position of that row's
// suffix.  A query whose seed window occurs once in the text then needs no SA read: the entry itself says where the
// single candidate is, and the text decides (2 random lines per such query instead of 3).
// text4 != nullptr (nucleotide): where the SEED_CTX_LEN + extra letters in front of the occurrence exist and are all
// ACGT they go into the entry as well (SEED_CTX, layout.h).
// text8 != nullptr (amino): where the five residues in front of BWT[row]'s exist they go into the entry (AA_SEED_SPECIAL).
__global__ __launch_bounds__(256) void seed_rows_to_positions_kernel(SeedEntry* __restrict__ table, uint64_t nentries,
                                                                     const uint32_t* __restrict__ dense_sa, uint32_t cnt_mask,
                                                                     const uint32_t* __restrict__ text4, int extra,
                                                                     const uint8_t* __restrict__ text8 = nullptr) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const int clen = SEED_CTX_LEN + extra;  // <= 30
  auto letters16 = [](uint64_t x) {  // 16 nibbles -> 16 2-bit letters
    x &= 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    return (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  };
  for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nentries; o += stride) {
    SeedEntry e = table[o];
    if (text8) {  // amino entry: plain singletons only (bit 26 clear, count 1)
      if ((e.cnt & (AA_SEED_SPECIAL | AA_SEED_CNT_SAT)) != 1u) continue;
      const uint32_t p = dense_sa[e.sp];
      e.sp = p;
      if (p >= (uint32_t)AA_SEED_CTX_LEN) {  // text8[p - 1] is the BWT symbol already held in bits 27..31
        uint32_t ctx = 0;
        for (int j = 0; j < AA_SEED_CTX_LEN - 1; j++) ctx |= (uint32_t)(text8[p - 2 - j] & 0x1Fu) << (5 * j);
        e.cnt = (e.cnt & 0xF8000000u) | AA_SEED_SPECIAL | ctx;
      }
      table[o] = e;
      continue;
    }
    if ((e.cnt & cnt_mask) != 1u) continue;  // cnt_mask: SEED_CNT_SAT (nt)
    const uint32_t p = dense_sa[e.sp];
    e.sp = p;
    if (text4 && p >= (uint32_t)clen) {
      const uint64_t t0 = (uint64_t)p - clen;  // clen nibbles from nibble t0: at most five words
      const Text20 t = *reinterpret_cast<const Text20*>(text4 + (t0 >> 3));
      const int sh = 4 * (int)(t0 & 7);
      const uint64_t a0 = (uint64_t)t.w[0] | ((uint64_t)t.w[1] << 32), a1 = (uint64_t)t.w[2] | ((uint64_t)t.w[3] << 32), a2w = t.w[4];
      const uint64_t lo = sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;   // nibbles 0..15
      uint64_t hi = sh ? (a1 >> sh) | (a2w << (64 - sh)) : a1;         // nibbles 16..31
      hi &= clen > 16 ? (~0ull >> (4 * (32 - clen))) : 0ull;          // only the first clen nibbles count
      const uint64_t lo_used = clen >= 16 ? lo : (lo & ((1ull << (4 * clen)) - 1));
      if (((lo_used | hi) & 0x8888888888888888ull) == 0) {  // all of them are A, C, G or T
        const uint64_t full = letters16(lo_used) | (letters16(hi) << 32);  // text[p - clen + j] in bits [2j, 2j + 2)
        const uint32_t far = extra ? (uint32_t)(full & ((1ull << (2 * extra)) - 1)) : 0u;
        e.sp = p | (extra ? far << (32 - 2 * extra) : 0u);
        e.cnt = (e.cnt & 0xE0000000u) | SEED_CTX | (uint32_t)((full >> (2 * extra)) & SEED_CNT_SAT);
      }
    }
    table[o] = e;
  }
}

// Amino seed table (the 21 searchable symbols: 20 standard residues and X; '$' is never part of a window).  Same construction as the
// nucleotide table with sigma = 21 (layout.h, AA_SEED_SIGMA): entry o of level j+1 = one step of parent o / 21 with letter o % 21 (the
// leftmost window letter is the least significant digit).  Final entries pack the count in bits 0..26 (saturating
// at AA_SEED_CNT_SAT) and, for singletons, the 5-bit symbol index of BWT[sp] in bits 27..31.
__global__ __launch_bounds__(256) void aa_seed_level1_kernel(DevIndex ix, SeedEntry* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x < AA_SEED_SIGMA) {
    const int idx = aa_index_of_letter((int)threadIdx.x);
    const uint64_t s = ix.prefix_sums[idx], e = ix.prefix_sums[idx + 1];
    out[threadIdx.x] = SeedEntry{(uint32_t)s, (uint32_t)(e - s)};
  }
}

__global__ __launch_bounds__(256) void aa_seed_extend_kernel(DevIndex ix, const SeedEntry* __restrict__ parent,
                                                             SeedEntry* __restrict__ child, uint64_t nchild) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nchild; o += stride) {
    const SeedEntry p = parent[o / AA_SEED_SIGMA];
    SeedEntry r{p.sp, 0};
    if (p.cnt) {
      uint64_t sp = p.sp, ep = (uint64_t)p.sp + p.cnt - 1;
      step_scalar<AMINO>(ix, sp, ep, aa_index_of_letter((int)(o % AA_SEED_SIGMA)));
      r.sp = (uint32_t)sp;
      r.cnt = sp > ep ? 0u : (uint32_t)(ep - sp + 1);
    }
    child[o] = r;
  }
}

__global__ __launch_bounds__(256) void aa_seed_finalize_kernel(DevIndex ix, SeedEntry* __restrict__ table, uint64_t nentries) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nentries; o += stride) {
    SeedEntry e = table[o];
    if (e.cnt == 1u) e.cnt = 1u | ((uint32_t)symbol_at<AMINO>(ix, e.sp) << 27);
    else if (e.cnt >= 2u && e.cnt <= 4u) {  // the set of BWT symbols over the entry's rows (AA_SEED_MULTI, layout.h)
      uint32_t mask = 0;
      for (uint32_t j = 0; j < e.cnt; j++) mask |= 1u << symbol_at<AMINO>(ix, (uint64_t)e.sp + j);
      e.cnt = AA_SEED_SPECIAL | AA_SEED_MULTI | ((e.cnt - 2u) << 22) | mask;
    }
    else if (e.cnt >= AA_SEED_CNT_SAT) e.cnt = AA_SEED_CNT_SAT;
    else continue;
    table[o] = e;
  }
}

// ASCII queries -> packed words (letter j of a query in word j / 32, bits 2 (j % 32); W words per query, unused ones
// zero); *bad counts queries with a byte outside ACGTacgt and bad_list (if given, room for n entries) names them, in
// no particular order (U counts too: the caller redoes those queries with the generic kernel, which applies the full
// alphabet map).  RAGGED: query q is ascii[off[q] - base, off[q + 1] - base) and its length goes to
// lens[q]; otherwise every query has L bytes.
//
// A wave packs 64 consecutive queries at a time: their bytes are one contiguous range, fetched with coalesced 16-B
// loads into the wave's LDS tile, from which every lane packs its own query (one query per lane reading its bytes
// straight from global memory ran at 98 GB/s of ASCII).  Ranges that do not fit the tile are cut into fewer queries
// per pass; a single query longer than the tile is packed from global memory by its lane.
constexpr int PACK_TILE = 8192;  // bytes of LDS per wave
template <bool RAGGED>
__global__ __launch_bounds__(256) void pack_nt2_tile_kernel(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ off, uint64_t base,
                                                            uint64_t n, uint64_t total_bytes, int L, int W, uint64_t* __restrict__ words,
                                                            uint32_t* __restrict__ lens, unsigned long long* __restrict__ bad,
                                                            uint32_t* __restrict__ bad_list) {
  const int64_t mis = (int64_t)(reinterpret_cast<uintptr_t>(ascii) & 15);  // tile chunks are 16-B aligned in memory
  __shared__ __attribute__((aligned(16))) uint8_t s_tile[4][PACK_TILE + 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint8_t* tile = s_tile[wv];
  const uint64_t nwaves = (uint64_t)gridDim.x * 4, wave0 = (uint64_t)blockIdx.x * 4 + wv;
  for (uint64_t q0 = wave0 * 64; q0 < n; q0 += nwaves * 64) {  // wave-uniform trip count
    const uint64_t q = q0 + lane;
    const bool have = q < n;
    uint64_t s = 0, e = 0;  // this lane's query: bytes [s, e) of ascii
    if (have) {
      s = RAGGED ? off[q] - base : q * (uint64_t)L;
      e = RAGGED ? off[q + 1] - base : s + (uint64_t)L;
    }
    uint64_t done = 0;  // lanes [0, done) of this group of 64 are packed
    const uint64_t nq = n - q0 < 64 ? n - q0 : 64;
    while (done < nq) {
      // the longest run of queries starting at lane `done` whose bytes fit the tile (measured from a 16-B aligned start)
      const int64_t b0 = (int64_t)__shfl(s, (int)done, 64), a0 = ((b0 + mis) & ~15ll) - mis;  // may be < 0 by up to 15
      const bool fits = have && (uint64_t)lane >= done && (int64_t)e - a0 <= (int64_t)PACK_TILE;
      const uint64_t fm = __ballot(fits) >> done;
      const int m = fm == ~0ull ? 64 : __builtin_ctzll(~fm);  // leading run of fitting lanes
      const bool mine = (uint64_t)lane >= done && (uint64_t)lane < done + (m ? m : 1);
      auto pack_from = [&](auto src) {  // src: this lane's query bytes, in LDS or in global memory
        // eight letters per step, word-wise: upper-case, check that every byte is one of A C G T, take bits 1..2 of the
        // ASCII code (A 00, C 01, T 10, G 11), swap the last two, squeeze the eight 2-bit codes into 16 bits
        const int len = (int)(e - s);
        uint64_t w = 0;
        uint64_t ok = 0x8080808080808080ull;  // bit 7 of byte b stays set while letter b of every step was valid
        uint64_t* out = words + q * (uint64_t)W;
        constexpr uint64_t K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
        auto eq = [&](uint64_t u, uint64_t pat) { const uint64_t t = u ^ pat; return ((((t & K7F) + K7F) | t) & K80) ^ K80; };
        for (int j = 0; j < len; j += 8) {
          const int nb = len - j < 8 ? len - j : 8;
          uint64_t x = 0;
          if (nb == 8) {
            __builtin_memcpy(&x, &src[j], 8);
          } else {
            for (int t = 0; t < nb; t++) x |= (uint64_t)src[j + t] << (8 * t);
            x |= 0x4141414141414141ull << (8 * nb);  // pad with 'A': valid, and zero bits in the packed word
          }
          const uint64_t c = x & 0xDFDFDFDFDFDFDFDFull;  // upper-case
          const uint64_t valid = (eq(c, 0x4141414141414141ull) | eq(c, 0x4343434343434343ull) | eq(c, 0x4747474747474747ull) |
                                  eq(c, 0x5454545454545454ull)) & ~(x & K80);  // and no byte >= 0x80 before the case fold
          ok &= valid;
          uint64_t y = (c >> 1) & 0x0303030303030303ull;
          y ^= (y >> 1) & 0x0101010101010101ull;
          y = (y | (y >> 6)) & 0x000F000F000F000Full;
          y = (y | (y >> 12)) & 0x000000FF000000FFull;
          y = (y | (y >> 24)) & 0xFFFFull;
          w |= y << (2 * (j & 31));
          if ((j & 31) == 24 || j + 8 >= len) { out[j >> 5] = w; w = 0; }
        }
        for (int k2 = (len + 31) >> 5; k2 < W; k2++) out[k2] = 0;
        if (RAGGED) lens[q] = (uint32_t)len;
        if (ok != K80) {  // rare: the caller redoes this query with the generic kernel
          const unsigned long long at = atomicAdd(bad, 1ull);
          if (bad_list) bad_list[at] = (uint32_t)q;
        }
      };
      if (m > 0) {
        const int64_t b1 = (int64_t)__shfl(e, (int)(done + m - 1), 64);
        for (int64_t i = a0 + 16ll * lane; i < b1; i += 16ll * 64) {
          if (i >= 0 && i + 16 <= (int64_t)total_bytes) {
            *reinterpret_cast<uint4*>(tile + (i - a0)) = *reinterpret_cast<const uint4*>(ascii + i);
          } else {  // first / last chunk of the buffer: only the bytes that exist
            for (int t = 0; t < 16; t++)
              if (i + t >= 0 && i + t < (int64_t)total_bytes) tile[i - a0 + t] = ascii[i + t];
          }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (mine && have) pack_from(tile + ((int64_t)s - a0));
      } else if (mine && have) {
        pack_from(ascii + s);  // one query longer than the tile: its lane reads global memory directly
      }
      __builtin_amdgcn_wave_barrier();
      done += m ? m : 1;
    }
  }
}

// Packed reads of any length (W = ceil(L/32) words per query, letter j in word j/32, bits 2(j%32)): the quad design
// of the k-mer kernels, the current word re-read every 32 letters, the final range handed on for locate.
//
// VERIFY adds seed-and-verify, an MI355X-first shortcut the 288 GB of HBM pay for (dense SA + 4-bit text resident):
// once the range has shrunk to <= 8 rows, the remaining i symbols are not matched by i dependent LF steps (i random
// lines) but by comparing them with the text in front of each candidate suffix: 1 SA read + the i/2 contiguous bytes
// of text per candidate.  The rows that survive are exactly the rows whose suffixes extend to the whole query, in
// the same relative order as the final range (the suffixes share everything after the seed part), so counts and
// locations are unchanged; the locate pass receives the verified candidates instead of a row range (RS_* words).
// LIST: the quads of block b work through the reads block b of count_nt2_reads_probe_kernel left undecided
// (sv.q / sv.count, same grid) instead of all n reads.
// RAGGED: read q has lens[q] letters (1 <= lens[q] <= L); L only sets the stride of W words per read.
// (the kernel's body as a block-level function -- LIST: the block's quads work through list_q[0 .. n) -- so that
//  lcx_quad_reads_kernel can run it over what its lanes left undecided; allow_lcx = false there: those reads take LF steps)
template <bool USE_SEED, bool VERIFY, bool LIST, bool RAGGED>
__device__ __forceinline__ void reads_body(const DevIndex& ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                           uint64_t* __restrict__ counts, uint64_t* __restrict__ range_start,
                                           const uint32_t* __restrict__ list_q, const uint32_t* __restrict__ lens, bool allow_lcx,
                                           uint64_t r_start = ~0ull, uint64_t r_stride = 64) {
  const int l = threadIdx.x & 3;
  const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  uint64_t r = r_start == ~0ull ? threadIdx.x >> 2 : r_start;  // LIST: position in the list (a block's own: its 64 quads)
  uint64_t q = LIST ? (r < n ? list_q[r] : 0) : ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint64_t* __restrict__ blocks = ix.blocks;
  const SeedEntry* __restrict__ seed = ix.seed;
  const uint32_t* __restrict__ dense = ix.dense_sa;
  const uint32_t* __restrict__ text4 = ix.text4;
  const int k = USE_SEED ? ix.seed_k : 1, W = (L + 31) / 32;
  const int verify_after = (int)ix.verify_after;
  // left-context index (layout.h): a seed range of 2+ rows is narrowed by a search over its bucket's keys -- the 32 letters
  // left of the seed window in log16(rows) lines -- instead of one LF step per letter; what is left is compared with the text
  const bool lcx = USE_SEED && VERIFY && ix.lcx_key != nullptr && allow_lcx;
  const uint32_t cA = (uint32_t)ix.prefix_sums[1], cC = (uint32_t)ix.prefix_sums[2], cG = (uint32_t)ix.prefix_sums[3],
                 cN = (uint32_t)ix.prefix_sums[4], cT = (uint32_t)ix.prefix_sums[5], cEnd = (uint32_t)ix.prefix_sums[6];
  bool have = LIST ? r < n : q < n, fresh = true;
  uint64_t w = 0;
  uint32_t sp = 1, ep = 0;
  int i = 0, steps_done = 0;
  // quad-uniform state: mode 0 = LF steps, 1 = text position of candidate vj, 2 = compare text chunk vc;
  // left-context index: 3 = number of incomplete entries of the bucket, 4 = search its keys
  int mode = 0, vj = 0, vc = 0;
  uint32_t vmask = 0, vp = 0;
  bool pos_hit = false;  // position seed whose window is the whole read
  bool cand_lcx = false;  // modes 1 / 2: candidates sp..ep are entries of the left-context index, not rows
  bool tail_pass = false; // modes 1 / 2: the candidates are the bucket's incomplete entries (after the key search)
  LcxQ lq{0, 0, 0, 0, -1};
  uint64_t qlo = 0, qhi = 0;
  uint32_t b_sp = 0, b_cnt = 0, b_inc = 0, key_hits = 0, key_lb = 0;
  while (__any(have)) {
    if (have) {
      const uint64_t* qw = queries + q * W;
      bool finished = false;
      uint64_t out_count = 0, out_rs = 0;
      const bool hole = LIST && q == 0xFFFFFFFFull;  // an empty slot of lcx_quad_reads_kernel's list: nothing to do
      if (hole) {
        finished = true;
      } else if (mode == 0) {
        if (fresh) {
          const int Lq = RAGGED ? (int)lens[q] : L;
          const bool seeded = USE_SEED && (!RAGGED || Lq >= k);  // a read shorter than the seed starts without the table
          const int first = seeded ? Lq - k : 0;  // letters first .. Lq-1 form the seed window (leftmost letter least significant)
          const int a = first >> 5, sh = 2 * (first & 31);
          uint64_t win = qw[a] >> sh;
          if (sh && a + 1 < W) win |= qw[a + 1] << (64 - sh);
          SeedEntry e{1u, 0u};
          uint32_t scnt = SEED_CNT_SAT;
          if (seeded) {
            e = seed[(win & ((1ull << (2 * k)) - 1))];
            scnt = seed_cnt(e);
            sp = scnt ? e.sp : 1u;
            ep = scnt ? e.sp + scnt - 1u : 0u;
            i = first;
          }
          if (!seeded || scnt == SEED_CNT_SAT) {  // no table, or a count the entry cannot represent
            const uint32_t c = (uint32_t)(qw[(Lq - 1) >> 5] >> (2 * ((Lq - 1) & 31))) & 3u;  // SearchRange::new(last letter)
            sp = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
            ep = (c == 0 ? cC : (c == 1 ? cG : (c == 2 ? cN : cEnd))) - 1;
            i = Lq - 1;
          }
          steps_done = 0;
          w = i > 0 ? qw[(i - 1) >> 5] : 0;
          if (USE_SEED && scnt == 1u && i > 0) {  // singleton: it survives the next step only if BWT[sp] is the next letter
            const uint32_t nc = (uint32_t)(w >> (2 * ((i - 1) & 31))) & 3u;
            if (seed_sym(e) != (int)(nc == 3u ? 5u : nc + 1u)) { sp = 1u; ep = 0u; }
          }
          if (seeded && ix.seed_pos && scnt == 1u && sp <= ep) {
            // position seed: e.sp is SA[row], the text position of the single candidate -- there is no row to step from
            if (i == 0) {  // the read is the seed window itself
              pos_hit = true;
              vp = seed_position(e, (int)ix.ctx_extra);
            } else if (VERIFY && i < 65536) {  // straight to the text, no SA read
              vp = seed_position(e, (int)ix.ctx_extra);
              sp = ep = 0u;  // one candidate, index 0
              vj = 0;
              vmask = 0;
              cand_lcx = tail_pass = false;
              if (vp >= (uint32_t)i) { mode = 2; vc = 0; }
              else { mode = 1; vj = 1; }  // too close to the text's beginning: no match (finishes below)
            } else {  // start again without the table
              const uint32_t c = (uint32_t)(qw[(Lq - 1) >> 5] >> (2 * ((Lq - 1) & 31))) & 3u;
              sp = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
              ep = (c == 0 ? cC : (c == 1 ? cG : (c == 2 ? cN : cEnd))) - 1;
              i = Lq - 1;
              w = i > 0 ? qw[(i - 1) >> 5] : 0;
            }
          } else if (lcx && seeded && scnt >= 2u && scnt != SEED_CNT_SAT && !(e.cnt & SEED_LCX_NONE) && i > 0 && i < 65536) {
            b_sp = sp;
            b_cnt = scnt;
            b_inc = 0;
            lcx_thresholds(lcx_read_ctx(qw, W, i), i < LCX_CTX ? i : LCX_CTX, &qlo, &qhi);
            if (e.cnt & SEED_LCX_TAIL) mode = 3;
            else { lcx_begin(lq, b_sp, b_cnt); mode = 4; }
          }
          fresh = false;
        } else {
          i--;
          const uint32_t c = (uint32_t)(w >> (2 * (i & 31))) & 3u;
          const uint32_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
          quad_step(blocks, cl, sp, ep, c, l);
          steps_done++;
          if ((i & 31) == 0 && i > 0) w = qw[(i - 1) >> 5];
        }
        if (pos_hit) {
          finished = true;
          pos_hit = false;
          out_count = 1;
          out_rs = (RS_SINGLE << RS_MODE_SHIFT) | (uint64_t)vp;
        } else if (mode != 0) {
          // a position seed went straight to the text / the bucket's keys are searched
        } else if (sp > ep || i == 0) {
          finished = true;
          out_count = sp > ep ? 0ull : (uint64_t)(ep - sp) + 1ull;
          out_rs = (RS_PLAIN << RS_MODE_SHIFT) | sp;
        } else if (VERIFY) {
          const uint32_t cnt = ep - sp + 1u;
          if (verify_now(cnt, i, steps_done, verify_after) && i < 65536) { mode = 1; vj = 0; vmask = 0; cand_lcx = tail_pass = false; }
        }
      } else if (mode == 1) {  // text position of candidate vj: row sp + vj, or entry sp + vj of the left-context index
        vp = cand_lcx ? (uint32_t)ix.lcx_rowpos[sp + (uint32_t)vj] : dense[sp + (uint32_t)vj];
        if (vp >= (uint32_t)i) { mode = 2; vc = 0; }
        else vj++;  // the suffix starts too close to the text's beginning to have i symbols in front
      } else if (mode == 2) {      // compare window chunk vc of candidate vj
        const uint64_t g = (uint64_t)vp - (uint64_t)i;
        const int wi = 4 * vc + l;
        const uint32_t bad = quad_sum(verify_part(text4, g, i, vc, l, wi < W ? qw[wi] : 0ull));
        if (bad) { vj++; mode = 1; }
        else if (128 * (vc + 1) < i) vc++;
        else { vmask |= 1u << vj; vj++; mode = 1; }
      } else if (mode == 3) {  // the key slot of the bucket's last row holds the number of incomplete entries
        b_inc = (uint32_t)ix.lcx_key[b_sp + b_cnt - 1u];
        // (they can only match a read with fewer than 32 letters left of its seed window; more of them than are worth
        //  checking one by one: LF steps after all)
        if (i < LCX_CTX && b_inc > (uint32_t)LCX_TAIL_MAX) mode = 0;
        else { lcx_begin(lq, b_sp, b_cnt - b_inc); mode = 4; }
      } else {  // mode 4
        lcx_quad_step(ix, lq, qlo, qhi, l);
        if (lq.t < 0) {
          key_lb = lq.a0;
          key_hits = lq.a1 - lq.a0;
          if (i <= LCX_CTX) {
            // the keys hold every letter the read has left: the run IS the answer, but for the bucket's incomplete entries
            if (b_inc && i < LCX_CTX) {  // (at most LCX_TAIL_MAX <= 32 of them: vmask has a bit each)
              sp = b_sp + (b_cnt - b_inc); ep = b_sp + b_cnt - 1u;
              cand_lcx = tail_pass = true;
              mode = 1; vj = 0; vmask = 0;
            } else if (!range_start || key_hits <= 8u) {
              finished = true;
              out_count = key_hits;
              out_rs = key_hits ? ((RS_LCX << RS_MODE_SHIFT) | (uint64_t)key_lb | ((uint64_t)i << 32) | ((uint64_t)((1u << key_hits) - 1u) << 48))
                                : ((RS_PLAIN << RS_MODE_SHIFT) | 1ull);
            } else mode = 0;  // the locate pass wants the rows of a larger range: LF steps from the seed range
          } else if (key_hits == 0u) {
            finished = true;
            out_rs = (RS_PLAIN << RS_MODE_SHIFT) | 1ull;
          } else if (key_hits <= 8u) {  // the entries that agree on 32 letters: the rest of each is compared with the text
            sp = key_lb; ep = key_lb + key_hits - 1u;
            cand_lcx = true; tail_pass = false;
            mode = 1; vj = 0; vmask = 0;
          } else mode = 0;  // too many candidates still (a young or exact repeat): LF steps from the seed range
          if (mode == 0) { sp = b_sp; ep = b_sp + b_cnt - 1u; }
        }
      }
      if (VERIFY && mode == 1 && vj > (int)(ep - sp)) {  // all candidates checked
        if (tail_pass) {  // the bucket's incomplete entries: they add to the run the keys selected
          const uint32_t th = (uint32_t)__popc(vmask);
          tail_pass = cand_lcx = false;
          if (!range_start || (th == 0u && key_hits <= 8u)) {
            finished = true;
            out_count = (uint64_t)key_hits + th;
            out_rs = key_hits ? ((RS_LCX << RS_MODE_SHIFT) | (uint64_t)key_lb | ((uint64_t)i << 32) | ((uint64_t)((1u << key_hits) - 1u) << 48))
                              : ((RS_PLAIN << RS_MODE_SHIFT) | 1ull);
          } else { mode = 0; sp = b_sp; ep = b_sp + b_cnt - 1u; }
        } else {
          finished = true;
          out_count = (uint64_t)__popc(vmask);
          if (ep == sp && vmask) out_rs = (RS_SINGLE << RS_MODE_SHIFT) | ((uint64_t)vp - (uint64_t)i);
          else out_rs = ((cand_lcx ? RS_LCX : RS_MULTI) << RS_MODE_SHIFT) | (uint64_t)sp | ((uint64_t)i << 32) | ((uint64_t)vmask << 48);
          cand_lcx = false;
        }
      }
      if (finished) {
        if (l == 0 && !hole) {
          counts[q] = out_count;
          if (range_start) range_start[q] = out_rs;
        }
        if (LIST) {
          r += r_stride;
          have = r < n;
          q = have ? list_q[r] : 0;
        } else {
          q += nquads;
          have = q < n;
        }
        fresh = true;
        mode = 0;
      }
    }
  }
}

template <bool USE_SEED, bool VERIFY, bool LIST = false, bool RAGGED = false>
__global__ __launch_bounds__(256) void count_nt2_reads_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                              uint64_t* __restrict__ counts, uint64_t* __restrict__ range_start,
                                                              Nt2Survivors sv = Nt2Survivors{}, const uint32_t* __restrict__ lens = nullptr) {
  if (LIST) reads_body<USE_SEED, VERIFY, LIST, RAGGED>(ix, queries, (uint64_t)sv.count[blockIdx.x], L, counts, range_start, sv.q + (uint64_t)blockIdx.x * sv.cap, lens, true);
  else reads_body<USE_SEED, VERIFY, LIST, RAGGED>(ix, queries, n, L, counts, range_start, nullptr, lens, true);
}
// one device-wide list of reads (sv.count[0] of them at sv.q[0 ..)), walked by all quads of the grid with LF steps: what
// lcx_quad_reads_kernel could not settle
template <bool RAGGED>
__global__ __launch_bounds__(256) void count_nt2_reads_pool_kernel(DevIndex ix, const uint64_t* __restrict__ queries, int L, uint64_t* __restrict__ counts,
                                                                   uint64_t* __restrict__ range_start, Nt2Survivors sv, const uint32_t* __restrict__ lens) {
  reads_body<true, true, true, RAGGED>(ix, queries, (uint64_t)sv.count[0], L, counts, range_start, sv.q, lens, false,
                                       ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2, ((uint64_t)gridDim.x * blockDim.x) >> 2);
}

// Phase 1 of the two-phase schedule for reads (seed table + dense SA + 4-bit text resident, 3 <= L - k): one read per
// LANE.  The seed entry alone settles reads whose seed k-mer is absent or a singleton with the wrong BWT symbol; a
// singleton with the right symbol is one candidate, settled by SA[sp] and the L - k letters of text in front of it
// (queued in LDS so that full waves issue those loads, as in count_nt2_probe_kernel); the rest (2+ rows, saturated
// entries) goes to block-private lists that count_nt2_reads_kernel<.., LIST> works through with the quad machinery.
// Results are those of count_nt2_reads_kernel<true, true> (counts and RS_* range-start words).  RAGGED: read q has
// lens[q] letters; reads with fewer than 3 letters left of their seed window go to the lists unprobed.
template <bool RAGGED>
__global__ __launch_bounds__(256) void count_nt2_reads_probe_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                                    uint64_t* __restrict__ counts, uint64_t* __restrict__ range_start,
                                                                    Nt2Survivors sv, const uint32_t* __restrict__ lens) {
  constexpr int VQ = 192;
  __shared__ unsigned int s_count;
  __shared__ uint32_t s_vsp[4][VQ], s_vq[4][VQ];
  __shared__ uint8_t s_vn[4][VQ];
  __shared__ uint16_t s_vl[RAGGED ? 4 : 1][VQ];  // RAGGED: the read's length (<= 512 on this path)
  __shared__ uint64_t s_vw[4][3][VQ];  // the letters left of the seed window of a queued read (<= 96 of them: three words)
  if (threadIdx.x == 0) s_count = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && sv.lf_count) *sv.lf_count = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
  const SeedEntry* __restrict__ seed = ix.seed;
  const int k = ix.seed_k, W = (L + 31) / 32;  // RAGGED: L is the longest read, W the stride
  const bool pos = ix.seed_pos != 0;           // singleton entries hold SA[row]: no SA read, and no row to step from
  const int min_i0 = pos ? 1 : 3;              // fewest letters left of the seed window worth (or, with pos, needing) the text
  const int cx = (int)ix.ctx_extra, clen = SEED_CTX_LEN + cx;  // letters in front of the occurrence a context entry holds
  const uint64_t kmask = (1ull << (2 * k)) - 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t region = (uint64_t)blockIdx.x * sv.cap;
  // a queued read is compared with the text a few hundred seed probes after its words were read: by then the random seed
  // lines have pushed them out of L2, so the words wait in LDS beside the queue entry instead of being fetched again
  const bool stash = L - k <= 96;
  int vcount = 0;
  auto settle = [&](uint64_t q, uint64_t count, uint64_t rs) {
    counts[q] = count;
    if (range_start) range_start[q] = rs;
  };
  auto drain = [&](int base, int cnt) {  // queue entries [base, base + cnt), cnt <= 128: two per lane
    uint32_t q[2], sp[2], nc[2], vp[2];
    bool on[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int s = base + lane + 64 * h;
      on[h] = lane + 64 * h < cnt;
      q[h] = on[h] ? s_vq[wv_id][s] : 0;
      sp[h] = on[h] ? s_vsp[wv_id][s] : 0;
      nc[h] = on[h] ? s_vn[wv_id][s] : 0;
      vp[h] = on[h] ? ((pos && nc[h] == 1u) ? sp[h] : ix.dense_sa[sp[h]]) : 0;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (!on[h]) continue;
      const int i0 = (RAGGED ? (int)s_vl[RAGGED ? wv_id : 0][base + lane + 64 * h] : L) - k, nchunks = (i0 + 31) >> 5;
      const uint64_t* qw = queries + (uint64_t)q[h] * W;
      uint32_t mask = 0;
      uint64_t g1 = 0;
      for (uint32_t c2 = 0; c2 < nc[h]; c2++) {  // the rows of a range are neighbours in the dense SA: mostly one line
        const uint32_t p = c2 ? ix.dense_sa[sp[h] + c2] : vp[h];
        uint32_t bad = p >= (uint32_t)i0 ? 0u : 1u;  // else the suffix starts too close to the text's beginning
        const uint64_t g = (uint64_t)p - (uint64_t)i0;
        for (int c = 0; c < nchunks && !bad; c++)
          bad = verify_part(ix.text4, g, i0, c >> 2, c & 3, stash ? s_vw[wv_id][c][base + lane + 64 * h] : qw[c]);
        if (!bad) { mask |= 1u << c2; g1 = g; }
      }
      if (nc[h] == 1u && mask) settle(q[h], 1, (RS_SINGLE << RS_MODE_SHIFT) | g1);
      else settle(q[h], (uint64_t)__popc(mask), (RS_MULTI << RS_MODE_SHIFT) | (uint64_t)sp[h] | ((uint64_t)i0 << 32) | ((uint64_t)mask << 48));
    }
  };
  constexpr int NQ = 2;  // reads in flight per lane (3 and 4 measure the same)
  for (uint64_t wbase = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); wbase < n; wbase += NQ * stride) {
    uint64_t qv[NQ], win[NQ];
    uint64_t lw[NQ][3];
    uint32_t nc[NQ];
    bool probe[NQ];  // false: too short for the per-lane path (fewer than 3 letters left of the seed window)
    SeedEntry ev[NQ];
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      qv[h] = wbase + lane + (uint64_t)h * stride;
      win[h] = 0;
      nc[h] = 0;
      probe[h] = false;
      if (qv[h] < n) {
        const int i0 = (RAGGED ? (int)lens[qv[h]] : L) - k;
        probe[h] = i0 >= min_i0;
        if (probe[h]) {
          const uint64_t* qw = queries + qv[h] * W;
          const int wa = i0 >> 5, wsh = 2 * (i0 & 31);            // seed window: letters i0 .. L-1
          const int na = (i0 - 1) >> 5, nsh = 2 * ((i0 - 1) & 31);  // the letter in front of it
          win[h] = qw[wa] >> wsh;
          if (wsh && wa + 1 < W) win[h] |= qw[wa + 1] << (64 - wsh);
          nc[h] = (uint32_t)(qw[na] >> nsh) & 3u;
          if (stash) {
#pragma unroll
            for (int c = 0; c < 3; c++) lw[h][c] = c < W ? qw[c] : 0;
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      ev[h] = SeedEntry{1u, 0u};
      if (probe[h]) ev[h] = seed_probe(seed + (win[h] & kmask));
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      const bool valid = qv[h] < n;
      const SeedEntry e = ev[h];
      const uint32_t cnt = seed_cnt(e);
      bool survivor = false, queued = false;
      if (valid && !probe[h]) {
        survivor = true;
      } else if (valid) {
        if (cnt == 0u) settle(qv[h], 0, (RS_PLAIN << RS_MODE_SHIFT) | 1ull);
        else if (cnt == 1u && pos && seed_has_ctx(e)) {
          // the entry holds the letters in front of the one occurrence: a read with no more than that left of its
          // seed window is decided here; a longer one goes on to the text only if they agree
          const int i0 = (RAGGED ? (int)lens[qv[h]] : L) - k;
          const uint64_t* qw = queries + qv[h] * W;
          if (i0 <= clen) {
            const bool same = (qw[0] & ((1ull << (2 * i0)) - 1)) == (seed_full_ctx(e, cx) >> (2 * (clen - i0)));
            settle(qv[h], same ? 1 : 0,
                   same ? ((RS_SINGLE << RS_MODE_SHIFT) | (uint64_t)(seed_position(e, cx) - (uint32_t)i0)) : ((RS_PLAIN << RS_MODE_SHIFT) | 1ull));
          } else {
            const int f = i0 - clen, a = f >> 5, sh = 2 * (f & 31);
            uint64_t x = qw[a] >> sh;
            if (sh && a + 1 < W) x |= qw[a + 1] << (64 - sh);
            queued = (x & ((1ull << (2 * clen)) - 1)) == seed_full_ctx(e, cx);
            if (!queued) settle(qv[h], 0, (RS_PLAIN << RS_MODE_SHIFT) | 1ull);
          }
        } else if (cnt == 1u) {
          queued = seed_sym(e) == (int)(nc[h] == 3u ? 5u : nc[h] + 1u);
          if (!queued) settle(qv[h], 0, (RS_PLAIN << RS_MODE_SHIFT) | 1ull);  // BWT[sp] is not the next letter: absent
        } else if (cnt <= (uint32_t)VMULTI && (int)(3u * cnt) <= (RAGGED ? (int)lens[qv[h]] : L) - k) {
          queued = true;  // a handful of candidate rows: each is checked against the text here
        } else survivor = true;  // more rows, or a saturated entry
      }
      const uint64_t qm = __ballot(queued);
      if (qm) {
        if (queued) {
          const int s = vcount + (int)__popcll(qm & lane_lt);
          s_vsp[wv_id][s] = cnt == 1u && pos ? seed_position(e, cx) : e.sp;
          s_vq[wv_id][s] = (uint32_t)qv[h];
          s_vn[wv_id][s] = (uint8_t)cnt;
          if (RAGGED) s_vl[wv_id][s] = (uint16_t)lens[qv[h]];
          if (stash) {
#pragma unroll
            for (int c = 0; c < 3; c++) s_vw[wv_id][c][s] = lw[h][c];
          }
        }
        vcount += (int)__popcll(qm);
        __builtin_amdgcn_wave_barrier();
        if (vcount >= 128) {
          vcount -= 128;
          drain(vcount, 128);
          __builtin_amdgcn_wave_barrier();
        }
      }
      const uint64_t sm = __ballot(survivor);
      if (sm) {
        unsigned int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&s_count, (unsigned int)__popcll(sm));
        slot0 = __shfl(slot0, 0, 64);
        if (survivor) {
          const uint64_t s = region + slot0 + (uint64_t)__popcll(sm & lane_lt);
          sv.q[s] = (uint32_t)qv[h];
          if (sv.range) {  // for lcx_quad_reads_kernel: the probed entry (~0: not probed) and the <= 32 letters left of the seed window
            sv.range[s] = probe[h] ? ((uint64_t)e.sp | ((uint64_t)(cnt | (e.cnt & (SEED_LCX_NONE | SEED_LCX_TAIL))) << 32)) : ~0ull;
            sv.w[s] = probe[h] ? lcx_read_ctx(queries + qv[h] * W, W, (RAGGED ? (int)lens[qv[h]] : L) - k) : 0ull;
          }
        }
      }
    }
  }
  if (vcount > 0) drain(0, vcount);
  __syncthreads();
  if (threadIdx.x == 0) sv.count[blockIdx.x] = s_count;
}

// ------------------------------------------------------------------------------------------------
// locate v2: tiles of hits, per-lane walk state machines, optional dense device SA
// ------------------------------------------------------------------------------------------------
constexpr int LOC_TILE = 1024;  // hits per tile (one 256-thread block at a time)
constexpr int LOC_QCAP = 1024;  // query (offset, start) pairs cached in LDS per tile

// value of the suffix array at a sampled row: the file's bit-packed samples (rows r % sa_ratio == 0) or the
// dense device array (rows r % dense_ratio == 0, u32 entries) built by densify_sa_kernel
constexpr uint64_t LOC_WALK_FLAG = 1ull << 63;  // gpos[h] holds a BWT row that still has to walk to a sampled row

// global text position -> (record, offset): largest i with seq_starts[i] <= g (the intended semantics of
// src/sequence_index.rs:108-141, see SURVEY a-17)
__device__ __forceinline__ void localise(const DevIndex& ix, uint64_t g, uint64_t* __restrict__ out) {
  uint64_t a = 0, z = ix.nseq;
  if (ix.seq_bucket) {  // the records of g's bucket: from the one holding the bucket's first position to the one holding the next bucket's
    const uint64_t b = g >> ix.seq_bucket_shift;
    a = ix.seq_bucket[b];
    z = (uint64_t)ix.seq_bucket[b + 1] + 1;
  }
  while (z - a > 1) { uint64_t mid = (a + z) >> 1; if (ix.seq_starts[mid] <= g) a = mid; else z = mid; }
  out[0] = a;
  out[1] = g - (ix.nseq ? ix.seq_starts[a] : 0);
}
// the same over a copy of the record starts in LDS (a block loads it once): the search is a chain of dependent loads
constexpr int LOC_SEQ_LDS = 1024;
__device__ __forceinline__ void localise_lds(const uint64_t* s_starts, uint64_t nseq, uint64_t g, uint64_t* __restrict__ out) {
  uint64_t a = 0, z = nseq;
  while (z - a > 1) { uint64_t mid = (a + z) >> 1; if (s_starts[mid] <= g) a = mid; else z = mid; }
  out[0] = a;
  out[1] = g - (nseq ? s_starts[a] : 0);
}

// row -> "is it a sampled row" / sample index, without a 64-bit division per backstep (the test runs once per LF step;
// a generic u64 modulo is ~100 instructions and made the walk ALU-bound): power-of-two ratios (the default 8) use a
// mask and a shift, other ratios a 32-bit division while the row fits.
__device__ __forceinline__ bool ratio_divides(uint64_t ratio, uint64_t row) {
  if ((ratio & (ratio - 1)) == 0) return (row & (ratio - 1)) == 0;
  if ((row >> 32) == 0 && (ratio >> 32) == 0) return (uint32_t)row % (uint32_t)ratio == 0u;
  return row % ratio == 0;
}
__device__ __forceinline__ uint64_t ratio_quotient(uint64_t ratio, uint64_t row) {
  if ((ratio & (ratio - 1)) == 0) return row >> (63 - __clzll((long long)ratio));
  if ((row >> 32) == 0 && (ratio >> 32) == 0) return (uint32_t)row / (uint32_t)ratio;
  return row / ratio;
}
__device__ __forceinline__ bool row_is_sampled(const DevIndex& ix, const uint32_t* dense, uint32_t dense_ratio, uint64_t row) {
  return ratio_divides(dense ? (uint64_t)dense_ratio : ix.sa_ratio, row);
}
__device__ __forceinline__ uint64_t row_sample(const DevIndex& ix, const uint32_t* dense, uint32_t dense_ratio, uint64_t row) {
  return dense ? (uint64_t)dense[ratio_quotient(dense_ratio, row)] : sa_sample(ix, ratio_quotient(ix.sa_ratio, row));
}
// (sample + steps) % bwt_len of src/fm_index.rs:534; sample < bwt_len and a walk is shorter than the text
__device__ __forceinline__ uint64_t walked_position(uint64_t sample, uint64_t steps, uint64_t bwt_len) {
  const uint64_t g = sample + steps;
  return g >= bwt_len ? g - bwt_len : g;
}

// dense[j] = SA[j * dense_ratio] for every j, recovered from the file's samples by CHAINS: the thread of sampled row
// s (SA known) walks LF -- visiting the rows of text positions SA[s]-1, SA[s]-2, ... -- and fills them in until it
// meets the next sampled row, where another thread's chain starts.  Every row is visited exactly once (n LF steps in
// total).  Walking from every unsampled row to its next sample instead costs the SUM of those distances, which is
// quadratic in the gap length, and row sampling leaves gaps of millions of rows inside long N runs (LF moves by a
// constant stride there): 121 s instead of 1 s on a chr1-scale text.
template <int A>
__global__ __launch_bounds__(256) void densify_sa_kernel(DevIndex ix, uint32_t dense_ratio, uint64_t nsamples, uint32_t* __restrict__ dense) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t fr = ix.sa_ratio;  // rows and SA values fit 32 bits here (the dense SA needs bwt_len < 2^32)
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nsamples; j += stride) {
    uint32_t row = (uint32_t)(j * fr);
    uint32_t v = (uint32_t)sa_sample(ix, j);
    for (;;) {
      const uint32_t e = row / dense_ratio;
      if (e * dense_ratio == row) dense[e] = v;
      row = (uint32_t)backstep_scalar<A>(ix, row);
      if (row % fr == 0u) break;  // a sampled row: its own chain takes over
      v--;                        // the suffix one text position to the left (v > 0 here: only SA = 0 steps to row 0)
    }
  }
}

// the text as 4-bit codes, recovered from the index itself: T[SA[r] - 1] = BWT[r] (T[n-1] = '$' for the row with SA = 0).
// Needs the dense SA at ratio 1.  Codes: A0 C1 G2 T3, everything else has bit 3 set and never equals a query letter.
template <int A>
__global__ __launch_bounds__(256) void text4_scatter_kernel(DevIndex ix, uint32_t* __restrict__ text4) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ix.bwt_len; r += stride) {
    const uint64_t v = ix.dense_sa[r];
    const uint64_t p = v ? v - 1 : ix.bwt_len - 1;
    const int letter = nt_letter_of_index(symbol_at<A>(ix, r));
    const uint32_t code = letter >= 0 ? (uint32_t)letter : 8u;
    if (code) atomicOr(&text4[p >> 3], code << (4 * (p & 7)));
  }
}

// the text as symbol indices, one byte per position (DevIndex::text8), by the same identity
template <int A>
__global__ __launch_bounds__(256) void text8_scatter_kernel(DevIndex ix, uint8_t* __restrict__ text8) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < ix.bwt_len; r += stride) {
    const uint64_t v = ix.dense_sa[r];
    text8[v ? v - 1 : ix.bwt_len - 1] = (uint8_t)symbol_at<A>(ix, r);
  }
}

// largest query q >= lo with hit_off[q] <= h (the query that owns hit h; queries without hits are skipped because their
// offset equals their successor's).  Needs hit_off[lo] <= h; hit_off has n + 1 entries and hit_off[n] = total > h.
// Gallops from lo: the owner of a tile's first hit is usually a few queries past the previous tile's.
__device__ __forceinline__ uint64_t owner_from(const uint64_t* __restrict__ hit_off, uint64_t lo, uint64_t n, uint64_t h) {
  uint64_t a = lo, step = 1;
  while (a + step < n && hit_off[a + step] <= h) { a += step; step <<= 1; }
  uint64_t hi = a + step < n ? a + step : n;
  while (hi - a > 1) { const uint64_t mid = (a + hi) >> 1; if (hit_off[mid] <= h) a = mid; else hi = mid; }
  return a;
}

// Hits in tiles of LOC_TILE; a block draws RUNS of consecutive tiles from an atomic head (run_len tiles at a time: the
// launcher picks it so that every block still gets several runs).  Only the first tile of a run searches the whole offset
// array for the query that owns its first hit (27 dependent loads at GRCh38 batch sizes -- per tile that chain was most
// of this kernel's time); the next tile's owner is read off the offsets the block already holds in LDS.  Per tile the
// offsets and range words of up to LOC_QCAP queries are staged in LDS and every hit finds its query there.  A hit whose
// row is a sampled one (or whose count pass left a text position) is emitted here; the others are flagged for the walk
// kernels, so every hit costs the same and the hits of a tile are dealt out statically.
template <int A>
__global__ __launch_bounds__(256) void locate_tile_kernel(DevIndex ix, const uint64_t* __restrict__ range_start, int rs_stride,
                                                          const uint64_t* __restrict__ hit_off, uint64_t n, uint64_t total,
                                                          const uint32_t* __restrict__ dense, uint32_t dense_ratio,
                                                          uint64_t* __restrict__ gpos, uint64_t* __restrict__ pos,
                                                          unsigned long long* __restrict__ tile_counter, uint32_t run_len) {
  __shared__ uint64_t s_off[LOC_QCAP + 1];
  __shared__ uint64_t s_sp[LOC_QCAP];
  __shared__ uint64_t s_starts[LOC_SEQ_LDS];
  __shared__ unsigned long long s_tile;
  __shared__ uint64_t s_q0;
  const bool seq_lds = pos && ix.nseq <= (uint64_t)LOC_SEQ_LDS;
  if (seq_lds)
    for (uint64_t t = threadIdx.x; t < ix.nseq; t += blockDim.x) s_starts[t] = ix.seq_starts[t];
  const uint64_t ntiles = (total + LOC_TILE - 1) / LOC_TILE;
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = atomicAdd(tile_counter, (unsigned long long)run_len);
      s_tile = t0;
      if (t0 < ntiles) {  // owner of the run's first hit: plain binary search over all queries
        const uint64_t h = t0 * LOC_TILE;
        uint64_t lo = 0, hi = n;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (hit_off[mid] <= h) lo = mid; else hi = mid; }
        s_q0 = lo;
      }
    }
    __syncthreads();
    const uint64_t run0 = s_tile;
    if (run0 >= ntiles) break;
    const uint64_t run1 = run0 + run_len < ntiles ? run0 + run_len : ntiles;
    for (uint64_t tile = run0; tile < run1; tile++) {
      const uint64_t h0 = tile * LOC_TILE;
      const int tn = (int)(total - h0 < (uint64_t)LOC_TILE ? total - h0 : (uint64_t)LOC_TILE);
      const uint64_t h_last = h0 + (uint64_t)tn - 1;
      const uint64_t q0 = s_q0;
      const uint64_t nload = n - q0 < (uint64_t)LOC_QCAP ? n - q0 : (uint64_t)LOC_QCAP;  // queries q0 .. q0 + nload - 1
      for (uint64_t t = threadIdx.x; t <= nload; t += blockDim.x) s_off[t] = hit_off[q0 + t];
      __syncthreads();
      // do the staged queries own the whole tile?  (not when it spans more than LOC_QCAP queries, most of them without hits)
      const bool cached = s_off[nload] > h_last;
      uint64_t nq = nload;  // staged queries that own a hit of the tile: the range words of the others are not needed
      if (cached) {
        uint64_t lo = 0, hi = nload;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (s_off[mid] <= h_last) lo = mid; else hi = mid; }
        nq = lo + 1;
        for (uint64_t t = threadIdx.x; t < nq; t += blockDim.x) s_sp[t] = range_start[(q0 + t) * rs_stride];
      }
      __syncthreads();
      for (int t = threadIdx.x; t < tn; t += blockDim.x) {
        const uint64_t h = h0 + (uint64_t)t;
        uint64_t rs, j;  // the owning query's range-start word and the index of this hit inside the query
        if (cached) {
          uint64_t lo = 0, hi = nq;
          while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (s_off[mid] <= h) lo = mid; else hi = mid; }
          rs = s_sp[lo];
          j = h - s_off[lo];
        } else {
          const uint64_t lo = owner_from(hit_off, q0, n, h);
          rs = range_start[lo * rs_stride];
          j = h - hit_off[lo];
        }
        const uint64_t rmode = rs >> RS_MODE_SHIFT;
        bool direct = rmode != RS_PLAIN;
        uint64_t row = 0, gd = 0;
        if (rmode == RS_SINGLE) {
          gd = rs & ((1ull << 40) - 1);  // the count pass already verified this match against the text
        } else if (rmode == RS_MULTI) {
          uint32_t mask = (uint32_t)(rs >> 48) & 0xffu;
          for (uint64_t t2 = 0; t2 < j; t2++) mask &= mask - 1;  // drop the j lowest set bits
          const uint32_t cand = (uint32_t)__ffs((int)mask) - 1u;
          gd = (uint64_t)ix.dense_sa[(uint32_t)rs + cand] - ((rs >> 32) & 0xffffull);
        } else if (rmode == RS_LCX) {
          // matched entries of the left-context index (up to 8 neighbours of lcx_rowpos: one line): hit j is the one with
          // the (j + 1)-th smallest BWT row -- ascending row order inside a query, src/fm_index.rs:521
          const uint32_t mask = (uint32_t)(rs >> 48) & 0xffu, base = (uint32_t)rs;
          uint64_t rp[8];
#pragma unroll
          for (int c = 0; c < 8; c++) rp[c] = (mask >> c) & 1u ? ix.lcx_rowpos[base + c] : ~0ull;
          uint64_t pick = 0;
#pragma unroll
          for (int c = 0; c < 8; c++) {
            uint32_t below = 0;
#pragma unroll
            for (int d = 0; d < 8; d++) below += (rp[d] >> 32) < (rp[c] >> 32) ? 1u : 0u;
            if (((mask >> c) & 1u) && below == (uint32_t)j) pick = rp[c];
          }
          gd = (pick & 0xffffffffull) - ((rs >> 32) & 0xffffull);
        } else {
          row = rs + j;
        }
        if (direct || row_is_sampled(ix, dense, dense_ratio, row)) {
          const uint64_t g = direct ? gd : walked_position(row_sample(ix, dense, dense_ratio, row), 0, ix.bwt_len);  // src/fm_index.rs:534, 0 steps
          gpos[h] = g;
          if (pos) {
            if (seq_lds) localise_lds(s_starts, ix.nseq, g, pos + 2 * h);
            else localise(ix, g, pos + 2 * h);
          }
        } else {
          gpos[h] = row | LOC_WALK_FLAG;  // a walk kernel finishes this hit,
          if (pos) pos[2 * h] = ~0ull;    // localise_walked_kernel its record / offset
        }
      }
      if (threadIdx.x == 0 && tile + 1 < run1) {  // owner of the next tile's first hit
        const uint64_t hn = h0 + (uint64_t)LOC_TILE;
        if (cached) {
          uint64_t lo = nq - 1, hi = nload + 1;   // s_off[nq - 1] <= h_last < hn
          while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (s_off[mid] <= hn) lo = mid; else hi = mid; }
          s_q0 = lo < nload ? q0 + lo : owner_from(hit_off, q0 + nload, n, hn);  // (lo == nload: the staged offsets end at or before hn)
        } else {
          s_q0 = owner_from(hit_off, q0, n, hn);
        }
      }
      __syncthreads();
    }
  }
}

// Second pass of locate: the hits whose row is not a sampled one walk the LF-mapping to the next sample
// (src/fm_index.rs:521-540).  Walk lengths are geometric (mean ratio - 1, tail ~6x that), so hits are not tied to
// blocks or tiles: every wave draws batches of LOC_WALK_BATCH consecutive hit indices from one device-wide counter and
// its lanes take the next index of the batch whenever their walk ends -- no lane idles behind a long walk except at the
// very end of the launch.  (Inside the tile kernel the same walks ran at ~40 % lane utilisation.)
constexpr int LOC_WALK_BATCH = 512;
template <int A>
__global__ __launch_bounds__(256) void locate_walk_kernel(DevIndex ix, uint64_t total, const uint32_t* __restrict__ dense, uint32_t dense_ratio,
                                                          uint64_t* __restrict__ gpos, uint64_t* __restrict__ pos,
                                                          unsigned long long* __restrict__ batch_counter) {
  const int lane = threadIdx.x & 63;
  const uint64_t lane_lt = (1ull << lane) - 1;
  uint64_t cur = 0, end = 0;  // wave-uniform: the unassigned rest of this wave's batch
  bool exhausted = false;     // wave-uniform: the counter has run past the last hit
  bool need = true;
  uint64_t h = 0, row = 0, steps = 0;
  for (;;) {
    const uint64_t nm = __ballot(need);
    if (nm) {
      if (cur == end && !exhausted) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(batch_counter, (unsigned long long)LOC_WALK_BATCH);
        base = __shfl(base, 0, 64);
        cur = base < total ? base : total;
        end = base + LOC_WALK_BATCH < total ? base + LOC_WALK_BATCH : total;
        exhausted = cur == end;
      }
      if (exhausted && nm == ~0ull) break;  // nothing left to hand out and every lane is idle
      if (need) {
        const uint64_t idx = cur + (uint64_t)__popcll(nm & lane_lt);
        if (idx < end) {
          const uint64_t v = gpos[idx];
          if (v & LOC_WALK_FLAG) { h = idx; row = v & ~LOC_WALK_FLAG; steps = 0; need = false; }
        }
      }
      const uint64_t adv = cur + (uint64_t)__popcll(nm);
      cur = adv < end ? adv : end;
    }
    if (!need) {
      if (row_is_sampled(ix, dense, dense_ratio, row)) {
        gpos[h] = walked_position(row_sample(ix, dense, dense_ratio, row), steps, ix.bwt_len);  // src/fm_index.rs:534
        need = true;
      } else {
        row = backstep_scalar<A>(ix, row);
        steps++;
      }
    }
  }
}

// Third pass of locate, after a walk kernel: record / offset of the hits the tile pass deferred (pos[2h] == ~0).  Kept
// out of the walk kernels on purpose: the binary search over the record starts is a chain of dependent loads, and
// inside a state machine every wave iteration in which any quad emits would wait for all of it.
__global__ __launch_bounds__(256) void localise_walked_kernel(DevIndex ix, uint64_t total, const uint64_t* __restrict__ gpos,
                                                              uint64_t* __restrict__ pos) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < total; h += stride)
    if (pos[2 * h] == ~0ull) localise(ix, gpos[h], pos + 2 * h);
}

// SA values of the N block of the BWT (DevIndex::sa_nblock), by the chains of densify_sa_kernel: the thread of
// sampled row s walks LF until the next sampled row and records the rows of the window [lo, hi) it passes.
template <int A>
__global__ __launch_bounds__(256) void nblock_sa_kernel(DevIndex ix, uint64_t nsamples, uint32_t lo, uint32_t hi, uint32_t* __restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t fr = ix.sa_ratio;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nsamples; j += stride) {
    uint32_t row = (uint32_t)(j * fr);
    uint32_t v = (uint32_t)sa_sample(ix, j);
    for (;;) {
      if (row - lo < hi - lo) out[row - lo] = v;
      row = (uint32_t)backstep_scalar<A>(ix, row);
      if (row % fr == 0u) break;
      v--;
    }
  }
}

// Nucleotide walk kernel.  One hit per LANE, whole block per step: the lane fetches the 128-B block of its row with
// eight 16-B loads issued back to back (one line, one memory latency), then takes the BWT symbol and the rank from
// registers; the generic locate_walk_kernel above spends two dependent round trips per step (symbol, then rank) on
// ~14 separate 8-B loads.  Rows are u64, so it also serves indexes >= 2^32.
//
// Each lane is a three-state machine -- FETCH the next hit's row, WALK one backstep, EMIT at a sampled row -- and the
// lanes of a wave are in different states.  Written naively (load and use inside each branch) an iteration costs the
// SUM of the memory latencies of the branches present in the wave; here every iteration first issues the one load
// group each lane needs, under its state's predicate, and only then consumes the results.
//
// Measured on MI355X (GRCh38-scale, 5 M hits, ratio 8; locate = tile pass + walk + localise): walking inside the tile
// kernel 1.91 ms; this kernel 1.50 ms; a quad-cooperative variant (4 lanes per hit, 2 loads per lane, DPP reductions)
// 1.65 ms -- it was instruction-bound (~230 instructions per 16 steps), this one is bound by the texture path's
// per-lane line lookups (8 per step).
// LDS_SHARE: the blocks of a wave's 64 walks are fetched cooperatively -- eight lanes per block, so that one load
// instruction covers eight whole 128-B lines instead of 64 sixteen-byte pieces of 64 different lines (an eighth of the
// line lookups in the texture path) -- and handed to their lanes through a wave-private LDS tile.
// TALLY: tally[0] += LF steps taken, tally[1] += hits walked (the census the roofline figure of locate is computed from).
template <bool LDS_SHARE, bool TALLY = false>
__global__ __launch_bounds__(256) void locate_walk_nt_lane_kernel(DevIndex ix, uint64_t total, const uint32_t* __restrict__ dense,
                                                                  uint32_t dense_ratio, uint64_t* __restrict__ gpos,
                                                                  unsigned long long* __restrict__ batch_counter,
                                                                  unsigned long long* __restrict__ tally = nullptr) {
  constexpr int ROW = 9;  // 16-B pieces per tile row: 8 + 1 of padding against bank conflicts
  __shared__ ulonglong2 s_blk[LDS_SHARE ? 4 : 1][LDS_SHARE ? 64 * ROW : 1];
  const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t* __restrict__ blocks = ix.blocks;
  const uint64_t cA = ix.prefix_sums[1], cC = ix.prefix_sums[2], cG = ix.prefix_sums[3], cN = ix.prefix_sums[4], cT = ix.prefix_sums[5];
  const uint32_t* __restrict__ nblock = ix.sa_nblock;
  const uint64_t nspan = nblock ? cT - cN : 0ull;  // rows [cN, cN + nspan) resolve through nblock
  const uint64_t sa_bits = ix.sa_bits, bwt_len = ix.bwt_len, sentinel = ix.sentinel_row;
  const uint64_t ratio = dense ? (uint64_t)dense_ratio : (uint64_t)ix.sa_ratio;
  auto stops = [&](uint64_t row) { return ratio_divides(ratio, row) || row - cN < nspan; };
  uint64_t cur = 0, end = 0;  // wave-uniform: the unassigned rest of this wave's batch
  bool exhausted = false;     // wave-uniform
  enum { IDLE = 0, FETCH = 1, WALK = 2, EMIT = 3 };
  int state = IDLE;
  uint64_t h = 0, row = 0, steps = 0;
  unsigned long long t_steps = 0, t_hits = 0;
  for (;;) {
    const uint64_t nm = __ballot(state == IDLE);
    if (nm) {
      if (cur == end && !exhausted) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(batch_counter, (unsigned long long)LOC_WALK_BATCH);
        base = __shfl(base, 0, 64);
        cur = base < total ? base : total;
        end = base + LOC_WALK_BATCH < total ? base + LOC_WALK_BATCH : total;
        exhausted = cur == end;
      }
      if (exhausted && nm == ~0ull) break;
      if (state == IDLE) {
        const uint64_t idx = cur + (uint64_t)__popcll(nm & lane_lt);
        if (idx < end) { h = idx; state = FETCH; }
      }
      const uint64_t adv = cur + (uint64_t)__popcll(nm);
      cur = adv < end ? adv : end;
    }
    // ---- issue
    uint64_t v = 0, s0 = 0, s1 = 0, ss = 0;
    ulonglong2 B[8];
#pragma unroll
    for (int j = 0; j < 8; j++) B[j] = ulonglong2{0, 0};
    if (state == FETCH) v = gpos[h];
    const uint64_t wm = LDS_SHARE ? __ballot(state == WALK) : 0ull;
    if (LDS_SHARE) {
      if (wm) {  // round r: the eight lanes 8j..8j+7 fetch the block of lane 8r + j, one 16-B piece each
        const unsigned long long myblk = state == WALK ? (unsigned long long)(row >> 8) : 0ull;
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const int src = 8 * r + (lane >> 3);
          const unsigned long long b = __shfl(myblk, src, 64);
          if ((wm >> src) & 1ull) B[r] = reinterpret_cast<const ulonglong2*>(blocks + b * NT_BLOCK_WORDS)[lane & 7];
        }
      }
    } else if (state == WALK) {
      const ulonglong2* p = reinterpret_cast<const ulonglong2*>(blocks + (row >> 8) * NT_BLOCK_WORDS);
#pragma unroll
      for (int j = 0; j < 8; j++) B[j] = p[j];
    }
    if (state == EMIT) {
      if (row - cN < nspan) {
        s0 = nblock[row - cN];
      } else if (dense) {
        s0 = dense[ratio_quotient(ratio, row)];
      } else {  // src/compressed_suffix_array.rs:76-106
        const uint64_t off = ratio_quotient(ratio, row) * sa_bits;
        ss = off & 63;
        s0 = ix.sa_words[off >> 6];
        s1 = ix.sa_words[(off >> 6) + 1];  // the buffer has one word of slack
      }
    }
    asm volatile("" : "+v"(v), "+v"(s0), "+v"(s1), "+v"(B[0].x), "+v"(B[0].y), "+v"(B[1].x), "+v"(B[1].y), "+v"(B[2].x), "+v"(B[2].y),
                 "+v"(B[3].x), "+v"(B[3].y), "+v"(B[4].x), "+v"(B[4].y), "+v"(B[5].x), "+v"(B[5].y), "+v"(B[6].x), "+v"(B[6].y),
                 "+v"(B[7].x), "+v"(B[7].y));
    if (LDS_SHARE && wm) {  // pieces -> tile, then every walking lane picks up its own block
#pragma unroll
      for (int r = 0; r < 8; r++) s_blk[wv_id][(8 * r + (lane >> 3)) * ROW + (lane & 7)] = B[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (state == WALK) {
#pragma unroll
        for (int j = 0; j < 8; j++) B[j] = s_blk[wv_id][lane * ROW + j];
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- consume
    if (state == FETCH) {
      if (v & LOC_WALK_FLAG) { row = v & ~LOC_WALK_FLAG; steps = 0; state = stops(row) ? EMIT : WALK; }
      else state = IDLE;  // the tile pass already finished this hit
    } else if (state == EMIT) {
      uint64_t sample = s0;
      if (!(row - cN < nspan) && !dense) {
        sample = s0 >> ss;
        if (ss + sa_bits > 64) sample |= s1 << (64 - ss);
        if (sa_bits < 64) sample &= (1ull << sa_bits) - 1;
      }
      gpos[h] = walked_position(sample, steps, bwt_len);  // src/fm_index.rs:534
      if (TALLY) { t_steps += steps; t_hits++; }
      state = IDLE;
    } else if (state == WALK) {  // backstep, src/fm_index.rs:585-593; B[j] = {plane0[j], plane1[j]}, B[4 + j] = {plane2[j], milestone[j]}
      const uint64_t b = row >> 8;
      const int w = (int)((row >> 6) & 3), bit = (int)(row & 63);
      const uint64_t q0 = w == 0 ? B[0].x : (w == 1 ? B[1].x : (w == 2 ? B[2].x : B[3].x));
      const uint64_t q1 = w == 0 ? B[0].y : (w == 1 ? B[1].y : (w == 2 ? B[2].y : B[3].y));
      const uint64_t q2 = w == 0 ? B[4].x : (w == 1 ? B[5].x : (w == 2 ? B[6].x : B[7].x));
      const uint32_t code = (uint32_t)((q0 >> bit) & 1ull) | ((uint32_t)((q1 >> bit) & 1ull) << 1) | ((uint32_t)((q2 >> bit) & 1ull) << 2);
      steps++;
      if (code == 4u) {
        row = 0;  // '$': the walk continues from row 0
      } else {
        const int t = code == 6u ? 0 : (code == 5u ? 1 : (code == 3u ? 2 : (code == 1u ? 3 : -1)));  // A C G T, else N
        const uint32_t pc = t >= 0 ? code : 2u;  // as nt_index_of_code: anything else ranks as N (010)
        const uint64_t x0 = (pc & 1u) ? 0ull : ~0ull, x1 = (pc & 2u) ? 0ull : ~0ull, x2 = (pc & 4u) ? 0ull : ~0ull;
        const uint64_t last = ~0ull >> (63 - bit);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint64_t m = j < w ? ~0ull : (j == w ? last : 0ull);
          cnt += (uint32_t)__popcll((B[j].x ^ x0) & (B[j].y ^ x1) & (B[4 + j].x ^ x2) & m);
        }
        uint64_t ms, c0;
        if (t >= 0) {
          ms = t == 0 ? B[4].y : (t == 1 ? B[5].y : (t == 2 ? B[6].y : B[7].y));
          c0 = t == 0 ? cA : (t == 1 ? cC : (t == 2 ? cG : cT));
        } else {  // N: rows before the block that are neither A, C, G, T nor the single '$'
          ms = 256ull * b - (B[4].y + B[5].y + B[6].y + B[7].y) - (sentinel < 256ull * b ? 1ull : 0ull);
          c0 = cN;
        }
        row = c0 + ms + cnt - 1;
      }
      if (stops(row)) state = EMIT;
    }
  }
  if (TALLY && tally) {
    atomicAdd(&tally[0], t_steps);
    atomicAdd(&tally[1], t_hits);
  }
}

}  // namespace awry
#include "lcx_kernels.hip.h"
namespace awry {

// ------------------------------------------------------------------------------------------------
// Wide rows: nucleotide indexes of 2^32 rows or more (the reference is u64 throughout, src/search.rs:7).  The same quad
// design as above with 64-bit rows and 16-byte seed entries (SeedEntry64); no verify accelerators (the dense SA and the
// position seeds are 32-bit structures) -- every letter left of the seed window is an LF step.
// ------------------------------------------------------------------------------------------------

// sum over the 4 lanes of a quad of a 64-bit value
__device__ __forceinline__ uint64_t quad_sum64(uint64_t v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
__device__ __forceinline__ QuadBlock quad_load64(const uint64_t* __restrict__ blocks, uint64_t b, int l) {
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(blocks + b * NT_BLOCK_WORDS);
  QuadBlock q;
  q.lo = p[l];
  q.hi = p[4 + l];
  return q;
}
__device__ __forceinline__ uint64_t quad_rank_part64(const QuadBlock& d, const NtXor& x, uint64_t row, uint32_t c, int l) {
  const uint64_t pred = (d.lo.x ^ x.x0) & (d.lo.y ^ x.x1) & (d.hi.x ^ x.x2);
  const uint64_t cnt = (uint64_t)__popcll(pred & slice_mask((int)(row & 255u) - 64 * l));
  return cnt + ((uint32_t)l == c ? d.hi.y : 0ull);
}
// one backward-search step with letter c (src/fm_index.rs:559-582), 64-bit rows
__device__ __forceinline__ void quad_step64(const uint64_t* __restrict__ blocks, uint64_t cl, uint64_t& sp, uint64_t& ep, uint32_t c, int l) {
  const uint64_t r0 = sp - 1, r1 = ep;
  const uint64_t b0 = r0 >> 8, b1 = r1 >> 8;
  QuadBlock d0 = quad_load64(blocks, b0, l);
  QuadBlock d1 = d0;
  if (b1 != b0) d1 = quad_load64(blocks, b1, l);
  const NtXor x = nt_xor_of_letter(c);
  const uint64_t v0 = quad_sum64(quad_rank_part64(d0, x, r0, c, l));
  const uint64_t v1 = quad_sum64(quad_rank_part64(d1, x, r1, c, l));
  sp = cl + v0;
  ep = cl + v1 - 1;
}

__global__ __launch_bounds__(256) void seed64_level1_kernel(DevIndex ix, SeedEntry64* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    const int idx = nt_index_of_letter((int)threadIdx.x);
    const uint64_t s = ix.prefix_sums[idx], e = ix.prefix_sums[idx + 1];
    out[threadIdx.x] = SeedEntry64{s, e - s};
  }
}
__global__ __launch_bounds__(256) void seed64_extend_kernel(DevIndex ix, const SeedEntry64* __restrict__ parent,
                                                            SeedEntry64* __restrict__ child, uint64_t nchild) {
  const int l = threadIdx.x & 3;
  const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  const uint64_t cA = ix.prefix_sums[1], cC = ix.prefix_sums[2], cG = ix.prefix_sums[3], cT = ix.prefix_sums[5];
  for (uint64_t o = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; o < nchild; o += nquads) {
    const SeedEntry64 p = parent[o >> 2];
    SeedEntry64 r{p.sp, 0};
    if (p.cnt) {
      const uint32_t c = (uint32_t)(o & 3);
      const uint64_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
      uint64_t sp = p.sp, ep = p.sp + p.cnt - 1;
      quad_step64(ix.blocks, cl, sp, ep, c, l);
      r.sp = sp;
      r.cnt = sp > ep ? 0ull : ep - sp + 1ull;
    }
    if (l == 0) child[o] = r;
  }
}
__global__ __launch_bounds__(256) void seed64_finalize_kernel(DevIndex ix, SeedEntry64* __restrict__ table, uint64_t nentries) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nentries; o += stride) {
    SeedEntry64 e = table[o];
    if (e.cnt != 1ull) continue;
    e.cnt = 1ull | ((uint64_t)symbol_at<NUCLEOTIDE>(ix, e.sp) << 61);
    table[o] = e;
  }
}

// Packed reads / k-mers of any length on a wide-row index: W = ceil(L / 32) words per query (L <= 32: one word, the
// k-mer layout), one query per quad, strided.  RAGGED: read q has lens[q] letters.  Counts, and (optional) the first row
// of each range for the locate pass (RS_PLAIN words).  tally (nullable): [0] probes, [1] steps, [2] blocks ranked.
// LIST: the quads of block b work through the queries block b of count_nt2_wide_probe_kernel left undecided (sv.q, with
// the probed range in sv.range / sv.w, so the table is not read again) instead of all n.
template <bool USE_SEED, bool RAGGED, bool LIST = false>
__global__ __launch_bounds__(256) void count_nt2_wide_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                             uint64_t* __restrict__ counts, uint64_t* __restrict__ range_start,
                                                             const uint32_t* __restrict__ lens, unsigned long long* __restrict__ tally,
                                                             Nt2Survivors sv = Nt2Survivors{}) {
  const int l = threadIdx.x & 3;
  const uint64_t nquads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  const uint64_t region = LIST ? (uint64_t)blockIdx.x * sv.cap : 0;
  uint64_t r = threadIdx.x >> 2;  // LIST: position in the block's list
  if (LIST) n = sv.count[blockIdx.x];
  uint64_t q = LIST ? (r < n ? sv.q[region + r] : 0) : ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint64_t* __restrict__ blocks = ix.blocks;
  const SeedEntry64* __restrict__ seed = ix.seed64;
  const int k = USE_SEED ? ix.seed_k : 1, W = (L + 31) / 32;
  const uint64_t cA = ix.prefix_sums[1], cC = ix.prefix_sums[2], cG = ix.prefix_sums[3], cN = ix.prefix_sums[4], cT = ix.prefix_sums[5],
                 cEnd = ix.prefix_sums[6];
  bool have = LIST ? r < n : q < n, fresh = true;
  uint64_t w = 0, sp = 1, ep = 0;
  int i = 0;
  unsigned long long t_probe = 0, t_step = 0, t_blk = 0;
  while (__any(have)) {
    if (have) {
      const uint64_t* qw = queries + q * W;
      if (fresh) {
        const int Lq = RAGGED ? (int)lens[q] : L;
        const bool seeded = USE_SEED && Lq >= k;
        const int first = seeded ? Lq - k : 0;  // letters first .. Lq-1 form the seed window (leftmost letter least significant)
        const uint64_t probed = LIST ? sv.w[region + r] : 0ull;  // the entry's row count as phase 1 read it (0: it did not probe)
        if (LIST && seeded && probed) {
          sp = sv.range[region + r];
          ep = sp + probed - 1ull;
          i = first;
        } else if (seeded) {
          const int a = first >> 5, sh = 2 * (first & 31);
          uint64_t win = qw[a] >> sh;
          if (sh && a + 1 < W) win |= qw[a + 1] << (64 - sh);
          const SeedEntry64 e = seed[win & ((1ull << (2 * k)) - 1)];
          const uint64_t scnt = seed64_cnt(e);
          sp = scnt ? e.sp : 1ull;
          ep = scnt ? e.sp + scnt - 1ull : 0ull;
          i = first;
          if (scnt == 1ull && i > 0) {  // singleton: it survives the next step only if BWT[sp] is the next letter
            const uint32_t nc = (uint32_t)(qw[(i - 1) >> 5] >> (2 * ((i - 1) & 31))) & 3u;
            if (seed64_sym(e) != (int)(nc == 3u ? 5u : nc + 1u)) { sp = 1ull; ep = 0ull; }
          }
          t_probe++;
        } else {
          const uint32_t c = (uint32_t)(qw[(Lq - 1) >> 5] >> (2 * ((Lq - 1) & 31))) & 3u;  // SearchRange::new(last letter)
          sp = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
          ep = (c == 0 ? cC : (c == 1 ? cG : (c == 2 ? cN : cEnd))) - 1;
          i = Lq - 1;
        }
        w = i > 0 ? qw[(i - 1) >> 5] : 0;
        fresh = false;
      } else {
        i--;
        const uint32_t c = (uint32_t)(w >> (2 * (i & 31))) & 3u;
        const uint64_t cl = c == 0 ? cA : (c == 1 ? cC : (c == 2 ? cG : cT));
        t_step++;
        t_blk += ((sp - 1) >> 8) == (ep >> 8) ? 1u : 2u;
        quad_step64(blocks, cl, sp, ep, c, l);
        if ((i & 31) == 0 && i > 0) w = qw[(i - 1) >> 5];
      }
      if (sp > ep || i == 0) {
        if (l == 0) {
          counts[q] = sp > ep ? 0ull : ep - sp + 1ull;
          if (range_start) range_start[q] = (RS_PLAIN << RS_MODE_SHIFT) | sp;
        }
        if (LIST) {
          r += 64;
          have = r < n;
          q = have ? sv.q[region + r] : 0;
        } else {
          q += nquads;
          have = q < n;
        }
        fresh = true;
      }
    }
  }
  if (tally && l == 0) {
    atomicAdd(&tally[0], t_probe);
    atomicAdd(&tally[1], t_step);
    atomicAdd(&tally[2], t_blk);
  }
}

// Phase 1 of a two-phase schedule for wide-row indexes (the narrow path's count_nt2_probe_kernel / count_nt2_reads_probe_kernel
// without the 32-bit accelerators): one query per LANE, two in flight -- coalesced query reads and count writes, 64
// independent 16-byte seed probes per wave instruction (non-temporal).  The entry settles a query whose seed k-mer is absent,
// a singleton whose BWT symbol is not the next letter, and a query that is its own seed window; everything else -- ranges that
// have to be stepped, reads shorter than the seed -- is listed per block with its probed range for
// count_nt2_wide_kernel<.., LIST>, so the LF kernel's quads only see queries that need LF steps.
template <bool RAGGED, bool TALLY>
__global__ __launch_bounds__(256) void count_nt2_wide_probe_kernel(DevIndex ix, const uint64_t* __restrict__ queries, uint64_t n, int L,
                                                                   uint64_t* __restrict__ counts, uint64_t* __restrict__ range_start,
                                                                   Nt2Survivors sv, const uint32_t* __restrict__ lens,
                                                                   unsigned long long* __restrict__ tally) {
  __shared__ unsigned int s_count;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const SeedEntry64* __restrict__ seed = ix.seed64;
  const int k = ix.seed_k, W = (L + 31) / 32;
  const uint64_t kmask = (1ull << (2 * k)) - 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const uint64_t region = (uint64_t)blockIdx.x * sv.cap;
  unsigned long long t_probe = 0;
  constexpr int NQ = 2;
  for (uint64_t wbase = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); wbase < n; wbase += NQ * stride) {  // wave-uniform trip count
    uint64_t qv[NQ], win[NQ];
    uint32_t nc[NQ];
    int i0[NQ];
    bool probe[NQ];
    ulonglong2 ev[NQ];
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      qv[h] = wbase + lane + (uint64_t)h * stride;
      win[h] = 0;
      nc[h] = 0;
      i0[h] = 0;
      probe[h] = false;
      if (qv[h] < n) {
        i0[h] = (RAGGED ? (int)lens[qv[h]] : L) - k;
        probe[h] = i0[h] >= 0;
        if (probe[h]) {
          const uint64_t* qw = queries + qv[h] * W;
          const int wa = i0[h] >> 5, wsh = 2 * (i0[h] & 31);  // seed window: letters i0 .. Lq - 1
          win[h] = qw[wa] >> wsh;
          if (wsh && wa + 1 < W) win[h] |= qw[wa + 1] << (64 - wsh);
          if (i0[h] > 0) nc[h] = (uint32_t)(qw[(i0[h] - 1) >> 5] >> (2 * ((i0[h] - 1) & 31))) & 3u;  // the letter in front of it
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      ev[h] = ulonglong2{1ull, 0ull};
      if (probe[h]) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(seed + (win[h] & kmask));
        ev[h].x = __builtin_nontemporal_load(p);
        ev[h].y = __builtin_nontemporal_load(p + 1);
        if (TALLY) t_probe++;
      }
    }
#pragma unroll
    for (int h = 0; h < NQ; h++) {
      const bool valid = qv[h] < n;
      const SeedEntry64 e{ev[h].x, ev[h].y};
      const uint64_t cnt = seed64_cnt(e);
      bool survivor = false;
      if (valid && !probe[h]) survivor = true;  // shorter than the seed: LF steps from its last letter
      else if (valid) {
        uint64_t value = 0;
        bool settled = true;
        if (cnt == 0ull) value = 0;
        else if (i0[h] == 0) value = cnt;
        else if (cnt == 1ull && seed64_sym(e) != (int)(nc[h] == 3u ? 5u : nc[h] + 1u)) value = 0;  // BWT[sp] is not the next letter
        else { settled = false; survivor = true; }
        if (settled) {
          counts[qv[h]] = value;
          if (range_start) range_start[qv[h]] = (RS_PLAIN << RS_MODE_SHIFT) | (value ? e.sp : 1ull);
        }
      }
      const uint64_t sm = __ballot(survivor);
      if (sm) {
        unsigned int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(&s_count, (unsigned int)__popcll(sm));
        slot0 = __shfl(slot0, 0, 64);
        if (survivor) {
          const uint64_t s = region + slot0 + (uint64_t)__popcll(sm & lane_lt);
          sv.q[s] = (uint32_t)qv[h];
          sv.range[s] = e.sp;
          sv.w[s] = probe[h] ? cnt : 0ull;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) sv.count[blockIdx.x] = s_count;
  if (TALLY && tally && t_probe) atomicAdd(&tally[0], t_probe);
}

}  // namespace awry
