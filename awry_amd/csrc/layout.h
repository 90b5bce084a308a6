// layout.h -- HBM-resident index layout shared by host packers and gfx950 kernels.
//
// The reference stores a 256-symbol BWT block as P bit-planes (Vec256 = 4 x u64) followed by 8 / 24
// u64 milestones: 160 B (nucleotide) / 352 B (amino), 32-B aligned (/root/reference src/bwt.rs:12-30,
// 139-140).  Neither is a multiple of the 128-B L2 line, so one rank touches 2-4 lines.  The device
// layout is re-derived for CDNA4: a block is exactly one (nt) or two (aa) 128-B lines and is laid out
// so that the four lanes of a wavefront QUAD each own one 64-symbol slice (= one u64 word of every
// plane, the w-th word of the reference's Vec256) plus a share of the milestones.  A quad fetches a
// block with `global_load_dwordx4` instructions whose four lanes cover 64 contiguous bytes, ranks its
// slice with 64-bit popcounts, and sums the partials with two quad_perm DPP adds.
//
// Nucleotide block: 16 u64 words (128 B), quad lane l in 0..3:
//     word 2l   = plane0 word l      word 2l+1 = plane1 word l          (first 64-B half line)
//     word 8+2l = plane2 word l      word 9+2l = milestone of letter l  (second half), letters A,C,G,T
//   The '$' and N milestones of the reference (src/fm_index.rs:212-217) are derivable and not stored:
//     '$': exactly one sentinel row s  ->  [s < 256 b]
//     N  : 256 b - (A + C + G + T) - [s < 256 b]
// Amino block: 32 u64 words (256 B), quad lane l:
//     word 2l = plane0[l]  2l+1 = plane1[l] | 8+2l = plane2[l]  9+2l = plane3[l]
//     16+2l = plane4[l]    17+2l = ms32 pair 0  | 24+2l = ms32 pair 1  25+2l = ms32 pair 2
//   u32 milestone slot t = symbol_index - 1 (t in 0..20, i.e. A..W, X, Y) lives in lane t/6, pair
//   (t%6)/2, half (t%6)&1.  u32 milestones require bwt_len < 2^32 (Swiss-Prot: 9e7).
//
// Symbol codes inside the planes are the reference's (src/alphabet.rs:280-325), so converting to and
// from the .awry file layout (src/fm_index_file.rs:58-67) is a pure word permutation.
#pragma once
#include <stdint.h>

#if defined(__HIP__)  // compiled as HIP (kernels + runtime TU); plain C++ TUs get ordinary inline functions
#define AWRY_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define AWRY_HD inline
#endif

namespace awry {

enum : int { NUCLEOTIDE = 0, AMINO = 1 };

constexpr int NT_BLOCK_WORDS = 16;  // 128 B
constexpr int AA_BLOCK_WORDS = 32;  // 256 B
constexpr int SYMBOLS_PER_BLOCK = 256;  // src/bwt.rs:285

AWRY_HD int block_words(int alphabet) { return alphabet == NUCLEOTIDE ? NT_BLOCK_WORDS : AA_BLOCK_WORDS; }
AWRY_HD int num_planes(int alphabet) { return alphabet == NUCLEOTIDE ? 3 : 5; }
AWRY_HD int cardinality(int alphabet) { return alphabet == NUCLEOTIDE ? 6 : 22; }  // src/alphabet.rs:87-92

// word index of plane `b`, slice `l` inside a device block
AWRY_HD int plane_word(int alphabet, int b, int l) {
  if (alphabet == NUCLEOTIDE) return b < 2 ? 2 * l + b : 8 + 2 * l;
  return b < 2 ? 2 * l + b : (b < 4 ? 8 + 2 * l + (b - 2) : 16 + 2 * l);
}
// nucleotide: word of the u64 milestone of letter l (0..3 = A,C,G,T)
AWRY_HD int nt_ms_word(int l) { return 9 + 2 * l; }
// amino: (word, half) of u32 milestone slot t (= symbol index - 1)
AWRY_HD int aa_ms_word(int t) {
  int l = t / 6, j = t % 6;
  return j < 2 ? 17 + 2 * l : (j < 4 ? 24 + 2 * l : 25 + 2 * l);
}
AWRY_HD int aa_ms_half(int t) { return (t % 6) & 1; }

// nucleotide symbol index (src/alphabet.rs:230-235: $0 A1 C2 G3 N4 T5) -> milestone letter 0..3, or -1
AWRY_HD int nt_letter_of_index(int idx) { return idx == 5 ? 3 : (idx >= 1 && idx <= 3 ? idx - 1 : -1); }
AWRY_HD int nt_index_of_letter(int l) { return l == 3 ? 5 : l + 1; }

// seed-table entry: the search range of a k-mer as (start row, row count); count 0 = absent.
// Final-level nucleotide entries pack more into `cnt`: bits 0..27 the count, saturating at SEED_CNT_SAT ("at least
// this many: ignore the table for this query"), and -- for singleton ranges only -- bits 29..31 the symbol index
// stored in the BWT at row sp.  A singleton range survives a step with symbol c iff BWT[sp] == c, so most absent
// k-mers are rejected by the entry itself, without touching a BWT block.
// With position seeds (DevIndex::seed_pos) a singleton's sp is the text position p of its one occurrence, and bit 28
// (SEED_CTX) says that bits 0..27 hold the SEED_CTX_LEN = 14 letters in front of it instead of the count 1:
// text[p - 14 + j] in bits [2j, 2j + 2) -- the orientation of a packed query word -- so that a 31-mer probed with
// k = 17 is decided by its entry alone.  A text shorter than 2^32 leaves the top bits of p unused: they hold E =
// DevIndex::ctx_extra further letters, text[p - 14 - E + j] in bits [32 - 2E + 2j, ...) of sp.  With the default
// k = floor(log4 n) + 2 that makes 14 + E >= 31 - k for every text size: 31-mers are always decided by the entry.
// Set only where all 14 + E positions exist and are ACGT.
struct SeedEntry { uint32_t sp, cnt; };
constexpr uint32_t SEED_CNT_SAT = 0x0FFFFFFFu;
constexpr uint32_t SEED_CTX = 0x10000000u;
constexpr int SEED_CTX_LEN = 14;
AWRY_HD bool seed_has_ctx(SeedEntry e) { return (e.cnt & SEED_CTX) != 0; }
AWRY_HD uint32_t seed_cnt(SeedEntry e) { return seed_has_ctx(e) ? 1u : (e.cnt & SEED_CNT_SAT); }
AWRY_HD uint32_t seed_ctx(SeedEntry e) { return e.cnt & SEED_CNT_SAT; }
// text position of a position seed (extra = DevIndex::ctx_extra)
AWRY_HD uint32_t seed_position(SeedEntry e, int extra) { return extra && seed_has_ctx(e) ? e.sp & (0xFFFFFFFFu >> (2 * extra)) : e.sp; }
// all 14 + extra letters in front of the occurrence, text[p - 14 - extra + j] in bits [2j, 2j + 2)
AWRY_HD uint64_t seed_full_ctx(SeedEntry e, int extra) {
  return (extra ? (uint64_t)(e.sp >> (32 - 2 * extra)) : 0ull) | ((uint64_t)seed_ctx(e) << (2 * extra));
}
AWRY_HD int seed_sym(SeedEntry e) { return (int)(e.cnt >> 29); }
// Entries of 2+ rows leave bits 29..31 of cnt unused; while the left-context index (DevIndex::lcx_key, below) is resident:
constexpr uint32_t SEED_LCX_TAIL = 0x20000000u;  // the bucket's last rows have an incomplete left context (see LCX below)
constexpr uint32_t SEED_LCX_NONE = 0x40000000u;  // the bucket is not in the left-context index (too many rows): LF steps

// Seed entry of an index with 2^32 rows or more ("wide rows": the packed kernels then carry 64-bit rows): the exact range
// of a k-mer as (start row, count); final-level entries of singleton ranges also hold the symbol index stored in the BWT
// at that row in bits 61..63 of cnt.  16 bytes, so the table's k is one less than a 32-bit index of the same HBM would get.
struct SeedEntry64 { uint64_t sp, cnt; };
constexpr uint64_t SEED64_CNT_MASK = (1ull << 61) - 1;
AWRY_HD uint64_t seed64_cnt(SeedEntry64 e) { return e.cnt & SEED64_CNT_MASK; }
AWRY_HD int seed64_sym(SeedEntry64 e) { return (int)(e.cnt >> 61); }

// Amino seed entries (final level).  Plain: count in bits 0..25 (saturating at AA_SEED_CNT_SAT = "at least this many:
// ignore the table"), and for singletons the 5-bit symbol index of BWT[sp] -- the residue in front of the one occurrence
// -- in bits 27..31.  Bit 26 (AA_SEED_SPECIAL) marks the two encodings that let the ENTRY decide most k-mer queries:
//   * bit 25 clear: a singleton whose sp is the text position p of its occurrence (position seeds) and whose bits 0..24
//     hold five more residues in front of it, text[p - 2 - j] in bits [5j, 5j + 5) -- with the BWT symbol that is the
//     AA_SEED_CTX_LEN = 6 residues left of the seed window: a 12-mer probed with k = 7 needs no text access;
//   * bit 25 set (AA_SEED_MULTI): a range of 2..4 rows, sp its first row, count - 2 in bits 22..23, and in bits 0..21
//     the set of symbol indices that occur in the BWT over those rows: a query whose next residue is not among them
//     is absent (most random queries that hit such an entry).
constexpr uint32_t AA_SEED_CNT_SAT = 0x03FFFFFFu;
constexpr uint32_t AA_SEED_SPECIAL = 0x04000000u;
constexpr uint32_t AA_SEED_MULTI = 0x02000000u;
constexpr int AA_SEED_CTX_LEN = 6;
AWRY_HD bool aa_seed_is_ctx(SeedEntry e) { return (e.cnt & (AA_SEED_SPECIAL | AA_SEED_MULTI)) == AA_SEED_SPECIAL; }
AWRY_HD bool aa_seed_is_multi(SeedEntry e) { return (e.cnt & (AA_SEED_SPECIAL | AA_SEED_MULTI)) == (AA_SEED_SPECIAL | AA_SEED_MULTI); }
AWRY_HD uint32_t aa_seed_cnt(SeedEntry e) {
  return (e.cnt & AA_SEED_SPECIAL) ? ((e.cnt & AA_SEED_MULTI) ? ((e.cnt >> 22) & 3u) + 2u : 1u) : (e.cnt & AA_SEED_CNT_SAT);
}
AWRY_HD uint32_t aa_seed_sym(SeedEntry e) { return e.cnt >> 27; }        // singletons (plain or with context)
AWRY_HD uint32_t aa_seed_mask(SeedEntry e) { return e.cnt & 0x3FFFFFu; }  // AA_SEED_MULTI entries
AWRY_HD uint32_t aa_seed_ctx(SeedEntry e) { return e.cnt & 0x1FFFFFFu; }  // context entries: residues p-2 .. p-6
// digits of an amino seed-table index: the 21 searchable symbols -- 0..18 -> A..W (indices 1..19), 19 -> Y (21), 20 -> X (20).
// X is a digit like the others: record delimiters and every non-standard letter search as X, and a k-mer that spans a
// record boundary (or holds B / Z / U / O / J) is then decided by its entry like any other instead of falling back to
// LF steps from the last letter (2 % of the 12-mers drawn from a Swiss-Prot-scale text, a third of that batch's time).
constexpr int AA_SEED_SIGMA = 21;
AWRY_HD int aa_index_of_letter(int l) { return l < 19 ? l + 1 : (l == 19 ? 21 : 20); }
AWRY_HD int aa_letter_of_index(int idx) { return idx >= 1 && idx <= 19 ? idx - 1 : (idx == 21 ? 19 : (idx == 20 ? 20 : -1)); }

// Everything a kernel needs, passed by value (fits the kernarg segment).
struct DevIndex {
  const uint64_t* blocks;     // nblocks * block_words(alphabet), 128-B aligned
  const uint64_t* sa_words;   // bit-packed sampled SA, src/compressed_suffix_array.rs:51-64
  const SeedEntry* seed;      // sigma^seed_k entries or nullptr (indexes below 2^32 rows)
  const SeedEntry64* seed64;  // 4^seed_k entries of a wide-row nucleotide index, or nullptr
  const uint64_t* seq_starts; // nseq record start offsets
  // indexes of many records (protein databases, contig-level assemblies): seq_bucket[b] = the record that holds text
  // position b << seq_bucket_shift, so that a position's record is found among the few records of its bucket instead of
  // by ~20 dependent loads over all record starts; nullptr when the record starts fit the locate kernels' LDS copy
  const uint32_t* seq_bucket;
  uint32_t seq_bucket_shift, seq_bucket_pad;
  uint64_t nblocks, bwt_len, sentinel_row, nseq;
  uint64_t prefix_sums[24];   // C[i], src/fm_index.rs:233-240
  uint32_t sa_bits, sa_ratio;
  int32_t alphabet, seed_k;
  // optional device-only accelerators (nullptr / 0 when absent); results never depend on them
  const uint32_t* dense_sa;   // SA[j * dense_ratio] as u32
  const uint32_t* text4;      // the text as 4-bit codes (A0 C1 G2 T3, 8 = anything else), 8 symbols per u32, LSB first
  uint32_t dense_ratio;
  uint32_t verify_after;      // seed-and-verify: switch from LF steps to text comparison after this many steps
  const uint8_t* text8;       // the text as symbol indices (0 = '$'), one byte per position: lets the generic kernel finish a
                              //   query against the text like the packed kernels do, for any alphabet and any letters
  uint32_t ctx_extra;         // E: context letters kept in the unused top bits of a position seed's sp (layout.h, SeedEntry)
  uint32_t seed_pos;          // 1: singleton seed entries hold the TEXT POSITION SA[row] in .sp instead of the row (kept only
                              //   while the verify accelerators are resident and the two-phase schedules are the policy)
  const uint32_t* sa_nblock;  // SA of every row whose suffix starts with N (rows [C[N], C[T])), or nullptr: ends locate
                              //   walks that run into an N run, where LF moves by a constant stride and row sampling can
                              //   leave a walk without a sampled row for the length of the run
  // LCX, the left-context index (device-only accelerator; nullptr when absent).  A seed bucket -- the rows [sp, sp + cnt)
  // of the suffixes that start with one seed k-mer -- is sorted by what FOLLOWS the k-mer, while backward search has to
  // tell its rows apart by what PRECEDES it: one LF step (two random block lines) per letter, and a k-mer inside a repeat
  // family has 10^3..10^5 rows that stay together for dozens of letters.  lcx_key holds, for every row slot of a bucket
  // of 2+ rows, the 32 letters in front of one of the bucket's suffixes -- text[p - 32 .. p) packed like a query word, so
  // the letter next to the seed is the most significant -- with the bucket's entries SORTED by that key; lcx_rowpos holds
  // the same entry's text position p (low word) and its BWT row (high word).  The letters a query has left of its seed
  // window then select a contiguous run of a bucket's entries: its length is the count (up to 32 letters), its entries are
  // the candidates to compare with the text (more letters).  Entries whose 32 left letters do not all exist as ACGT sit at
  // the bucket's end, unsorted (SEED_LCX_TAIL; the key slot of the bucket's last row holds how many).
  // lcx_inner: every 16^t-th key, t = 1.. (level t at lcx_off[t]): a search runs top-down through one aligned 16-key node
  // (one 128-B line) per level, masked to the bucket's own rows -- log16(cnt) lines instead of 2 x (letters left).
  const uint64_t* lcx_key;
  const uint64_t* lcx_rowpos;
  const uint64_t* lcx_inner;
  uint32_t lcx_off[8];
};
constexpr int LCX_CTX = 32;  // letters of left context in a key

// encoding of a query's "range start" word handed from the count pass to the locate pass
constexpr uint64_t RS_MODE_SHIFT = 62;
constexpr uint64_t RS_PLAIN = 0;   // low bits = first BWT row of the final range
constexpr uint64_t RS_MULTI = 1;   // verified candidates: rows sp..sp+7, bit j of mask = candidate j matched;
                                   //   bits 0..31 sp, 32..47 symbols left of the seed part (i), 48..55 mask
constexpr uint64_t RS_SINGLE = 2;  // one verified match: low 40 bits = text position of the match
constexpr uint64_t RS_LCX = 3;     // matches found through the left-context index: entries base .. base + 7 of lcx_rowpos, bit j
                                   //   of mask = entry j matched; bits 0..31 base, 32..47 letters left of the seed (i), 48..55
                                   //   mask.  The locate pass emits them in ascending BWT row (lcx_rowpos holds each entry's row).

}  // namespace awry
