/*
 * awry_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 * See awry_oracle.h for scope and the parity pin.  Citations are file:line under
 * /root/reference/ (AWRY 0.3.1).  Written from the behaviour of that code, not copied:
 * the reference is Rust with AVX2/NEON intrinsics, this is scalar C11.
 */
#define _GNU_SOURCE
#include "awry_oracle.h"
#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ alphabet */

/* src/alphabet.rs:87-92 */
uint8_t orc_cardinality(int alphabet) { return alphabet == ORC_NUCLEOTIDE ? 6 : 22; }
/* src/alphabet.rs:95-97 */
static uint8_t num_encoding_symbols(int alphabet) { return (uint8_t)(orc_cardinality(alphabet) - 2); }

static const char AA_LETTERS[23] = "$ACDEFGHIKLMNPQRSTVWXY"; /* index -> ascii, src/alphabet.rs:339-362 */
static const uint8_t AA_CODES[22] = {                        /* index -> code,  src/alphabet.rs:280-303 */
    0x00, 0x0C, 0x17, 0x03, 0x06, 0x1E, 0x1A, 0x1B, 0x19, 0x15, 0x1C,
    0x1D, 0x08, 0x09, 0x04, 0x13, 0x0A, 0x05, 0x16, 0x01, 0x1F, 0x02};

/* Symbol::new_ascii upper-cases (src/alphabet.rs:109-114), then to_index (src/alphabet.rs:169-248) */
uint8_t orc_ascii_to_index(int alphabet, uint8_t a) {
  if (a >= 'a' && a <= 'z') a = (uint8_t)(a - 32);
  if (a == '#' || a == '$') return 0;
  if (alphabet == ORC_NUCLEOTIDE) {
    switch (a) {
      case 'A': return 1;
      case 'C': return 2;
      case 'G': return 3;
      case 'T': case 'U': return 5;
      default: return 4; /* ambiguity N */
    }
  }
  if (a == 'X') return 20;
  for (uint8_t i = 1; i < 22; i++)
    if ((uint8_t)AA_LETTERS[i] == a) return i;
  return 20; /* ambiguity X */
}

/* src/alphabet.rs:251-330 (Index arm) */
uint8_t orc_index_to_code(int alphabet, uint8_t idx) {
  if (alphabet == ORC_NUCLEOTIDE) {
    switch (idx) {
      case 0: return 4; /* 0b100 $ */
      case 1: return 6; /* 0b110 A */
      case 2: return 5; /* 0b101 C */
      case 3: return 3; /* 0b011 G */
      case 5: return 1; /* 0b001 T */
      default: return 2; /* 0b010 N */
    }
  }
  return idx < 22 ? AA_CODES[idx] : 0x1F;
}

uint8_t orc_ascii_to_code(int alphabet, uint8_t a) { return orc_index_to_code(alphabet, orc_ascii_to_index(alphabet, a)); }

/* src/alphabet.rs:169-248 (BitVector arm) */
uint8_t orc_code_to_index(int alphabet, uint8_t code) {
  if (alphabet == ORC_NUCLEOTIDE) {
    switch (code) {
      case 4: return 0;
      case 6: return 1;
      case 5: return 2;
      case 3: return 3;
      case 1: return 5;
      default: return 4;
    }
  }
  for (uint8_t i = 0; i < 22; i++)
    if (i != 20 && AA_CODES[i] == code) return i;
  return 20;
}

/* src/alphabet.rs:334-413 */
uint8_t orc_index_to_ascii(int alphabet, uint8_t idx) {
  if (alphabet == ORC_NUCLEOTIDE) {
    switch (idx) {
      case 0: return '$';
      case 1: return 'A';
      case 2: return 'C';
      case 3: return 'G';
      case 5: return 'T';
      default: return 'N';
    }
  }
  return idx < 22 ? (uint8_t)AA_LETTERS[idx] : 'X';
}
uint8_t orc_code_to_ascii(int alphabet, uint8_t code) { return orc_index_to_ascii(alphabet, orc_code_to_index(alphabet, code)); }

/* ------------------------------------------------------------------ Vec256 + blocks */

/* inclusive popcount of bits 0..=pos: src/simd_instructions.rs:96-121 */
uint32_t orc_masked_popcount(const uint64_t v[4], uint64_t pos) {
  uint64_t masks[4] = {0, 0, 0, 0};
  uint64_t w = pos / 64;
  for (uint64_t i = 0; i < w; i++) masks[i] = ~0ULL;
  masks[w] = ~0ULL >> (63 - (pos % 64));
  uint32_t c = 0;
  for (int i = 0; i < 4; i++) c += (uint32_t)__builtin_popcountll(v[i] & masks[i]);
  return c;
}

/* set_bit per plane whose code bit is 1: src/bwt.rs:65-79,177-195; src/simd_instructions.rs:65-71 */
void orc_block_set_symbol(uint64_t *planes, int nplanes, uint8_t code, uint64_t pos) {
  for (int b = 0; b < nplanes && code; b++, code >>= 1)
    if (code & 1) planes[4 * b + pos / 64] |= 1ULL << (pos % 64);
}

/* one bit per plane: src/bwt.rs:53-62,162-174; src/simd_instructions.rs:57-63 */
uint8_t orc_block_code_at(const uint64_t *planes, int nplanes, uint64_t pos) {
  uint8_t code = 0;
  for (int b = 0; b < nplanes; b++) code |= (uint8_t)(((planes[4 * b + pos / 64] >> (pos % 64)) & 1) << b);
  return code;
}

/* Vec256 and its three boolean ops: src/simd_instructions.rs:35-54,78-94.  With AVX2 (the build's -march=x86-64-v3, as
 * the reference's x86-64 back-end) they are the same intrinsics the reference uses -- _mm256_and_si256, _mm256_or_si256,
 * _mm256_andnot_si256 -- and the popcount stays four scalar popcnt over the extracted lanes (:96-121); without AVX2
 * (sanitizer builds with other flags) plain C over the four words. */
#if defined(__AVX2__)
#include <immintrin.h>
typedef union { __m256i v; uint64_t w[4]; } v256;
static inline v256 ld(const uint64_t *p) { v256 r; r.v = _mm256_loadu_si256((const __m256i *)p); return r; }
static inline v256 AND(v256 a, v256 b) { a.v = _mm256_and_si256(a.v, b.v); return a; }
static inline v256 OR(v256 a, v256 b) { a.v = _mm256_or_si256(a.v, b.v); return a; }
/* andnot(x, y) = !x & y : src/simd_instructions.rs:89-94 (the intrinsic's own operand order) */
static inline v256 ANDN(v256 a, v256 b) { a.v = _mm256_andnot_si256(a.v, b.v); return a; }
#else
typedef struct { uint64_t w[4]; } v256;
static inline v256 ld(const uint64_t *p) { v256 r; memcpy(r.w, p, 32); return r; }
static inline v256 AND(v256 a, v256 b) { for (int i = 0; i < 4; i++) a.w[i] &= b.w[i]; return a; }
static inline v256 OR(v256 a, v256 b) { for (int i = 0; i < 4; i++) a.w[i] |= b.w[i]; return a; }
/* andnot(x, y) = !x & y : src/simd_instructions.rs:89-94 */
static inline v256 ANDN(v256 a, v256 b) { for (int i = 0; i < 4; i++) a.w[i] = ~a.w[i] & b.w[i]; return a; }
#endif

/* src/bwt.rs:114-135 */
uint64_t orc_nt_block_occ(const uint64_t planes[12], const uint64_t ms[8], uint64_t pos, uint8_t sym) {
  v256 v0 = ld(planes), v1 = ld(planes + 4), v2 = ld(planes + 8), o;
  switch (sym) {
    case 1: o = AND(v1, v2); break;
    case 2: o = AND(v0, v2); break;
    case 3: o = AND(v0, v1); break;
    case 4: o = ANDN(v2, ANDN(v0, v1)); break;
    case 5: o = ANDN(v2, ANDN(v1, v0)); break;
    default: return ORC_PANIC;
  }
  return ms[sym] + orc_masked_popcount(o.w, pos);
}

/* src/bwt.rs:230-271 */
uint64_t orc_aa_block_occ(const uint64_t planes[20], const uint64_t ms[24], uint64_t pos, uint8_t sym) {
  v256 v0 = ld(planes), v1 = ld(planes + 4), v2 = ld(planes + 8), v3 = ld(planes + 12), v4 = ld(planes + 16), o;
  switch (sym) {
    case 1: o = AND(v2, ANDN(v4, v3)); break;
    case 2: o = ANDN(v3, AND(AND(v0, v1), v2)); break;
    case 3: o = ANDN(v4, AND(v0, v1)); break;
    case 4: o = ANDN(v4, AND(v1, v2)); break;
    case 5: o = ANDN(v0, AND(AND(v1, v2), v3)); break;
    case 6: o = ANDN(v2, ANDN(v0, v4)); break;
    case 7: o = ANDN(v2, AND(v0, AND(v1, v3))); break;
    case 8: o = ANDN(v2, ANDN(v1, v4)); break;
    case 9: o = ANDN(v1, ANDN(v3, v4)); break;
    case 10: o = ANDN(v1, ANDN(v0, v4)); break;
    case 11: o = ANDN(v1, AND(v3, AND(v2, v0))); break;
    case 12: o = ANDN(OR(v0, v1), ANDN(v2, v3)); break;
    case 13: o = AND(v3, ANDN(v4, v0)); break;
    case 14: o = ANDN(OR(v0, v1), ANDN(v3, v2)); break;
    case 15: o = ANDN(v2, ANDN(v3, v4)); break;
    case 16: o = AND(v1, ANDN(v4, v3)); break;
    case 17: o = AND(v0, ANDN(v4, v2)); break;
    case 18: o = ANDN(v3, ANDN(v0, v4)); break;
    case 19: o = ANDN(OR(v1, v2), ANDN(v3, v0)); break;
    case 20: o = AND(AND(v0, v1), AND(v2, v3)); break;
    case 21: o = ANDN(OR(v0, v2), ANDN(v3, v1)); break;
    default: return ORC_PANIC;
  }
  return ms[sym] + orc_masked_popcount(o.w, pos);
}

/* ------------------------------------------------------------------ CompressedSuffixArray */

/* src/compressed_suffix_array.rs:124-130 */
uint64_t orc_csa_bits_per_element(uint64_t bwt_len) {
  uint64_t largest = bwt_len - 1;
  return largest == 0 ? 0 : (uint64_t)(64 - __builtin_clzll(largest));
}
/* src/compressed_suffix_array.rs:113-123 */
uint64_t orc_csa_word_len(uint64_t bwt_len, uint64_t ratio) {
  uint64_t bits = orc_csa_bits_per_element(bwt_len);
  uint64_t n = (bwt_len + ratio - 1) / ratio;
  unsigned __int128 tot = (unsigned __int128)n * bits;
  return (uint64_t)((tot + 63) / 64);
}
/* src/compressed_suffix_array.rs:51-64 */
void orc_csa_set_value(uint64_t *data, uint64_t bits, uint64_t value, uint64_t position) {
  uint64_t word = (position * bits) / 64, bit = (position * bits) % 64;
  data[word] |= value << bit;
  if (bit + bits > 64) data[word + 1] |= (64 - bit) >= 64 ? 0 : value >> (64 - bit);
}
/* src/compressed_suffix_array.rs:76-106 */
int orc_csa_reconstruct(const uint64_t *data, uint64_t bits, uint64_t ratio, uint64_t position, uint64_t *out) {
  if (position % ratio != 0) return -1;
  uint64_t s = position / ratio;
  uint64_t word = (s * bits) / 64, start = (s * bits) % 64;
  uint64_t n1 = bits < 64 - start ? bits : 64 - start;
  uint64_t n2 = bits - n1;
  uint64_t m1 = n1 >= 64 ? ~0ULL : ((1ULL << n1) - 1);
  uint64_t v = (data[word] >> start) & m1;
  if (n2) v |= (data[word + 1] & ((1ULL << n2) - 1)) << n1;
  *out = v;
  return 0;
}

/* ------------------------------------------------------------------ suffix array + brute force */

/* Stand-in for libsufr 0.6.2 (crates.io, not in /root/reference): plain lexicographic SA over the
 * raw bytes of `text` (which ends in a unique '$').  Prefix doubling, O(n log^2 n); test sizes only. */
typedef struct { const uint64_t *rank; uint64_t h, n; } sa_ctx;
static int sa_cmp(const void *a, const void *b, void *c) {
  const sa_ctx *x = (const sa_ctx *)c;
  uint64_t i = *(const uint64_t *)a, j = *(const uint64_t *)b;
  if (x->rank[i] != x->rank[j]) return x->rank[i] < x->rank[j] ? -1 : 1;
  uint64_t ri = i + x->h < x->n ? x->rank[i + x->h] + 1 : 0;
  uint64_t rj = j + x->h < x->n ? x->rank[j + x->h] + 1 : 0;
  return ri < rj ? -1 : ri > rj;
}
int orc_suffix_array(const uint8_t *text, uint64_t n, uint64_t *sa) {
  if (n == 0) return 0;
  uint64_t *rank = malloc(n * 8), *tmp = malloc(n * 8);
  if (!rank || !tmp) { free(rank); free(tmp); return -1; }
  for (uint64_t i = 0; i < n; i++) { sa[i] = i; rank[i] = text[i]; }
  for (uint64_t h = 1;; h *= 2) {
    sa_ctx c = {rank, h, n};
    qsort_r(sa, n, 8, sa_cmp, &c);
    tmp[sa[0]] = 0;
    for (uint64_t i = 1; i < n; i++) tmp[sa[i]] = tmp[sa[i - 1]] + (sa_cmp(&sa[i - 1], &sa[i], &c) != 0);
    memcpy(rank, tmp, n * 8);
    if (rank[sa[n - 1]] == n - 1 || h >= n) break;
  }
  free(rank); free(tmp);
  return 0;
}

/* The definition the reference's integration test pins (src/fm_index.rs:612-664): occurrences of the
 * query in the text, after both go through the alphabet's ascii->index map (src/alphabet.rs:169-248). */
uint64_t orc_brute_locate(int alphabet, const uint8_t *text, uint64_t n, const uint8_t *pat, uint64_t m,
                          uint64_t *out, uint64_t cap) {
  uint64_t c = 0;
  if (m == 0 || m > n) return 0;
  for (uint64_t i = 0; i + m <= n; i++) {
    uint64_t j = 0;
    while (j < m && orc_ascii_to_index(alphabet, text[i + j]) == orc_ascii_to_index(alphabet, pat[j])) j++;
    if (j == m) { if (out && c < cap) out[c] = i; c++; }
  }
  return c;
}
uint64_t orc_brute_count(int alphabet, const uint8_t *text, uint64_t n, const uint8_t *pat, uint64_t m) {
  return orc_brute_locate(alphabet, text, n, pat, m, NULL, 0);
}

/* ------------------------------------------------------------------ FmIndex */

struct orc_index {
  int alphabet;              /* src/fm_index.rs:302-307 */
  uint64_t bwt_len, version, sa_ratio, sa_bits;
  uint8_t kmer_len;
  int nplanes, nms;          /* 3/8 or 5/24: src/bwt.rs:29-30,139-140 */
  uint64_t nblocks, block_words;
  uint64_t *blocks;          /* per block: planes (4 u64 each) then milestones -- the file order,
                                src/fm_index_file.rs:58-67 */
  uint64_t prefix_sums[24];  /* cardinality+1: src/fm_index.rs:233-240 */
  uint64_t *sa_words, n_sa_words;
  uint64_t *kmer_table, n_kmer; /* (start,end) pairs: src/kmer_lookup_table.rs:17-20 */
  uint64_t nseq, *seq_starts;
  char **headers;
  uint8_t *text;             /* kept for brute-force checks; NULL after load */
};

int orc_alphabet(const orc_index *x) { return x->alphabet; }
uint64_t orc_bwt_len(const orc_index *x) { return x->bwt_len; }
uint64_t orc_version(const orc_index *x) { return x->version; }
uint64_t orc_sa_ratio(const orc_index *x) { return x->sa_ratio; }
uint8_t orc_kmer_len(const orc_index *x) { return x->kmer_len; }
const uint64_t *orc_prefix_sums(const orc_index *x, uint64_t *len) { *len = orc_cardinality(x->alphabet) + 1u; return x->prefix_sums; }
const uint64_t *orc_block_words(const orc_index *x, uint64_t *n) { *n = x->nblocks * x->block_words; return x->blocks; }
const uint64_t *orc_sa_words(const orc_index *x, uint64_t *n) { *n = x->n_sa_words; return x->sa_words; }
const uint64_t *orc_kmer_table(const orc_index *x, uint64_t *n) { *n = x->n_kmer; return x->kmer_table; }
uint64_t orc_num_sequences(const orc_index *x) { return x->nseq; }
uint64_t orc_seq_start(const orc_index *x, uint64_t i) { return x->seq_starts[i]; }
const char *orc_seq_header(const orc_index *x, uint64_t i) { return x->headers[i]; }
const uint8_t *orc_text(const orc_index *x) { return x->text; }
void orc_free(void *p) { free(p); }

void orc_index_free(orc_index *x) {
  if (!x) return;
  free(x->blocks); free(x->sa_words); free(x->kmer_table); free(x->seq_starts); free(x->text);
  if (x->headers) for (uint64_t i = 0; i < x->nseq; i++) free(x->headers[i]);
  free(x->headers); free(x);
}

static void set_shape(orc_index *x) {
  x->nplanes = x->alphabet == ORC_NUCLEOTIDE ? 3 : 5;
  x->nms = x->alphabet == ORC_NUCLEOTIDE ? 8 : 24;
  x->block_words = (uint64_t)(4 * x->nplanes + x->nms);
  x->nblocks = (x->bwt_len + 255) / 256; /* src/bwt.rs:302-304 */
  x->sa_bits = orc_csa_bits_per_element(x->bwt_len);
  x->n_sa_words = orc_csa_word_len(x->bwt_len, x->sa_ratio);
}

/* src/bwt.rs:338-357 */
uint64_t orc_global_occurrence(const orc_index *x, uint64_t p, uint8_t sym) {
  const uint64_t *b = x->blocks + (p / 256) * x->block_words;
  return x->alphabet == ORC_NUCLEOTIDE ? orc_nt_block_occ(b, b + 12, p % 256, sym)
                                       : orc_aa_block_occ(b, b + 20, p % 256, sym);
}
/* src/bwt.rs:307-325 -> symbol index */
uint8_t orc_symbol_at(const orc_index *x, uint64_t p) {
  const uint64_t *b = x->blocks + (p / 256) * x->block_words;
  return orc_code_to_index(x->alphabet, orc_block_code_at(b, x->nplanes, p % 256));
}
/* src/search.rs:43-48, src/fm_index.rs:383-385 */
void orc_initial_range(const orc_index *x, uint8_t sym, uint64_t *sp, uint64_t *ep) {
  *sp = x->prefix_sums[sym];
  *ep = x->prefix_sums[sym + 1] - 1;
}
/* src/fm_index.rs:559-582 */
void orc_update_range(const orc_index *x, uint64_t sp, uint64_t ep, uint8_t sym, uint64_t *sp2, uint64_t *ep2) {
  uint64_t c = x->prefix_sums[sym];
  *sp2 = c + orc_global_occurrence(x, sp - 1, sym);
  *ep2 = c + orc_global_occurrence(x, ep, sym) - 1;
}
/* src/fm_index.rs:585-593 */
uint64_t orc_backstep(const orc_index *x, uint64_t p) {
  uint8_t s = orc_symbol_at(x, p);
  if (s == 0) return 0;
  return x->prefix_sums[s] + orc_global_occurrence(x, p, s) - 1;
}

/* src/kmer_lookup_table.rs:136-167 */
static void populate_rec(orc_index *x, uint64_t sp, uint64_t ep, unsigned cur_len, uint64_t cur_idx, uint64_t mult) {
  if (cur_len == x->kmer_len) { x->kmer_table[2 * cur_idx] = sp; x->kmer_table[2 * cur_idx + 1] = ep; return; }
  uint8_t nenc = num_encoding_symbols(x->alphabet);
  for (uint8_t i = 1; i < nenc; i++) {
    uint64_t s2, e2;
    orc_update_range(x, sp, ep, i, &s2, &e2);
    populate_rec(x, s2, e2, cur_len + 1, cur_idx + (uint64_t)i * mult, mult * nenc);
  }
}
/* src/kmer_lookup_table.rs:27-39,113-134 */
static int build_kmer_table(orc_index *x) {
  uint8_t nenc = num_encoding_symbols(x->alphabet);
  uint64_t n = 1;
  for (unsigned i = 0; i < x->kmer_len; i++) n *= nenc;
  x->n_kmer = n;
  x->kmer_table = malloc(n * 16);
  if (!x->kmer_table) return -1;
  for (uint64_t i = 0; i < n; i++) { x->kmer_table[2 * i] = 1; x->kmer_table[2 * i + 1] = 0; } /* SearchRange::zero() */
  if (x->kmer_len == 0) return 0;
  for (uint8_t s = 1; s < nenc; s++) {
    uint64_t sp, ep;
    orc_initial_range(x, s, &sp, &ep);
    populate_rec(x, sp, ep, 1, s, nenc);
  }
  return 0;
}

static int copy_seqs(orc_index *x, const uint64_t *starts, const char *const *headers, uint64_t nseq) {
  x->nseq = nseq;
  x->seq_starts = malloc((nseq ? nseq : 1) * 8);
  x->headers = calloc(nseq ? nseq : 1, sizeof(char *));
  if (!x->seq_starts || !x->headers) return -1;
  for (uint64_t i = 0; i < nseq; i++) {
    x->seq_starts[i] = starts[i];
    x->headers[i] = strdup(headers && headers[i] ? headers[i] : "");
  }
  return 0;
}

/* The single pass over the suffix array: src/fm_index.rs:182-240 (+ k-mer table :243-261). */
orc_index *orc_index_from_sa(const uint8_t *text, uint64_t bwt_len, const uint64_t *sa, int alphabet,
                             uint64_t sa_ratio, uint8_t kmer_len, const uint64_t *seq_starts,
                             const char *const *headers, uint64_t nseq) {
  orc_index *x = calloc(1, sizeof *x);
  if (!x) return NULL;
  x->alphabet = alphabet; x->bwt_len = bwt_len; x->version = 1;
  x->sa_ratio = sa_ratio ? sa_ratio : 8;                                   /* src/fm_index.rs:122 */
  x->kmer_len = kmer_len ? kmer_len : (alphabet == ORC_NUCLEOTIDE ? 10 : 4); /* src/kmer_lookup_table.rs:23-24 */
  set_shape(x);
  x->blocks = calloc(x->nblocks * x->block_words, 8);
  x->sa_words = calloc(x->n_sa_words ? x->n_sa_words : 1, 8);
  x->text = malloc(bwt_len);
  if (!x->blocks || !x->sa_words || !x->text) { orc_index_free(x); return NULL; }
  memcpy(x->text, text, bwt_len);
  uint8_t card = orc_cardinality(alphabet);
  uint64_t counts[24] = {0};
  for (uint64_t i = 0; i < bwt_len; i++) {
    uint64_t v = sa[i];
    if (i % x->sa_ratio == 0) orc_csa_set_value(x->sa_words, x->sa_bits, v, i / x->sa_ratio);
    uint64_t *blk = x->blocks + (i / 256) * x->block_words;
    if (i % 256 == 0)
      for (uint8_t c = 0; c < card; c++) blk[4 * x->nplanes + c] = counts[c]; /* src/bwt.rs:81-89,198-206 */
    uint8_t a = v == 0 ? '$' : text[v - 1];
    orc_block_set_symbol(blk, x->nplanes, orc_ascii_to_code(alphabet, a), i % 256);
    counts[orc_ascii_to_index(alphabet, a)]++;
  }
  uint64_t acc = 0;
  for (uint8_t i = 0; i <= card; i++) { x->prefix_sums[i] = acc; if (i != card) acc += counts[i]; }
  if (copy_seqs(x, seq_starts, headers, nseq) || build_kmer_table(x)) { orc_index_free(x); return NULL; }
  return x;
}

orc_index *orc_index_build(const uint8_t *text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio,
                           uint8_t kmer_len, const uint64_t *seq_starts, const char *const *headers,
                           uint64_t nseq) {
  uint64_t *sa = malloc(bwt_len * 8);
  if (!sa || orc_suffix_array(text, bwt_len, sa)) { free(sa); return NULL; }
  orc_index *x = orc_index_from_sa(text, bwt_len, sa, alphabet, sa_ratio, kmer_len, seq_starts, headers, nseq);
  free(sa);
  return x;
}

/* Text model inferred from src/fm_index.rs:148-153,182,220-223 (libsufr::util::read_sequence_file is
 * not in the tree): records joined by one delimiter byte ('N' nt / 'X' aa), terminated by one '$'.
 * Upper-cased (ignore_softmask: true, src/fm_index.rs:161).  Header = first whitespace-delimited token.
 * PARITY UNPINNED for anything but upper-case canonical letters. */
orc_index *orc_index_from_fasta(const char *path, int alphabet, uint64_t sa_ratio, uint8_t kmer_len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  size_t cap = 1 << 16, n = 0, nseq = 0, seqcap = 16;
  uint8_t *text = malloc(cap);
  uint64_t *starts = malloc(seqcap * 8);
  char **hdr = malloc(seqcap * sizeof(char *));
  char *line = NULL; size_t lcap = 0; ssize_t len;
  int fastq = 0, state = 0; /* fastq state: 0 header,1 seq,2 plus,3 qual */
  uint8_t delim = alphabet == ORC_NUCLEOTIDE ? 'N' : 'X';
  int first = 1;
  while ((len = getline(&line, &lcap, f)) >= 0) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
    if (first && len > 0) { fastq = line[0] == '@'; first = 0; }
    int is_header = fastq ? state == 0 : (len > 0 && line[0] == '>');
    if (is_header) {
      if (fastq && len == 0) continue;
      if (nseq == seqcap) { seqcap *= 2; starts = realloc(starts, seqcap * 8); hdr = realloc(hdr, seqcap * sizeof(char *)); }
      if (nseq > 0) { if (n + 1 >= cap) { cap *= 2; text = realloc(text, cap); } text[n++] = delim; }
      starts[nseq] = n;
      char *h = line + 1; size_t hl = 0;
      while (h[hl] && !isspace((unsigned char)h[hl])) hl++;
      hdr[nseq] = strndup(h, hl);
      nseq++;
      if (fastq) state = 1;
      continue;
    }
    if (fastq && state == 2) { state = 3; continue; }
    if (fastq && state == 3) { state = 0; continue; }
    if (n + (size_t)len + 2 >= cap) { while (n + (size_t)len + 2 >= cap) cap *= 2; text = realloc(text, cap); }
    for (ssize_t i = 0; i < len; i++) if (!isspace((unsigned char)line[i])) text[n++] = (uint8_t)toupper((unsigned char)line[i]);
    if (fastq) state = 2;
  }
  free(line); fclose(f);
  text[n++] = '$';
  orc_index *x = orc_index_build(text, n, alphabet, sa_ratio, kmer_len, starts, (const char *const *)hdr, nseq);
  for (size_t i = 0; i < nseq; i++) free(hdr[i]);
  free(hdr); free(starts); free(text);
  return x;
}

/* ------------------------------------------------------------------ search */

static inline void tally_step(orc_tally *t, uint64_t sp, uint64_t ep) {
  if (!t) return;
  t->steps++;
  t->block_reads += ((sp - 1) / 256 == ep / 256) ? 1 : 2;
}

/* src/fm_index.rs:402-438 with src/kmer_lookup_table.rs:90-110.  -1 where the reference panics
 * (empty query: unwrap on None) or is undefined ('$'/'#' symbols reach the `_ => panic!` arm of
 * global_occurrence or underflow start_ptr-1; non-ASCII bytes are outside the contract). */
int orc_search_range(const orc_index *x, const uint8_t *q, uint64_t len, uint64_t *sp_out, uint64_t *ep_out, orc_tally *t) {
  if (len == 0) return -1;
  for (uint64_t i = 0; i < len; i++)
    if (q[i] >= 0x80 || orc_ascii_to_index(x->alphabet, q[i]) == 0) return -1;
  uint64_t sp, ep, i = len - 1;
  orc_initial_range(x, orc_ascii_to_index(x->alphabet, q[i]), &sp, &ep);
  if (len < x->kmer_len) { /* path A: stop as soon as the range is empty */
    while (i-- > 0) {
      if (sp > ep) break;
      tally_step(t, sp, ep);
      orc_update_range(x, sp, ep, orc_ascii_to_index(x->alphabet, q[i]), &sp, &ep);
    }
  } else { /* path B: kmer_len-1 unconditional steps, then stop on empty */
    uint64_t uncond = x->kmer_len ? x->kmer_len - 1u : 0;
    while (uncond-- > 0 && i-- > 0) {
      tally_step(t, sp, ep);
      orc_update_range(x, sp, ep, orc_ascii_to_index(x->alphabet, q[i]), &sp, &ep);
    }
    while (i-- > 0) {
      if (sp > ep) break;
      tally_step(t, sp, ep);
      orc_update_range(x, sp, ep, orc_ascii_to_index(x->alphabet, q[i]), &sp, &ep);
    }
  }
  *sp_out = sp; *ep_out = ep;
  if (t) t->queries++;
  return 0;
}

/* src/fm_index.rs:499-501, src/search.rs:66-71 */
int orc_count_string(const orc_index *x, const uint8_t *q, uint64_t len, uint64_t *count) {
  uint64_t sp, ep;
  if (orc_search_range(x, q, len, &sp, &ep, NULL)) return -1;
  *count = sp > ep ? 0 : ep - sp + 1;
  return 0;
}

/* intended semantics of src/sequence_index.rs:108-141 */
void orc_seq_location(const orc_index *x, uint64_t gpos, orc_pos *out) {
  uint64_t lo = 0, hi = x->nseq;
  while (hi - lo > 1) { uint64_t mid = (lo + hi) / 2; if (x->seq_starts[mid] <= gpos) lo = mid; else hi = mid; }
  out->seq_idx = lo;
  out->local_pos = gpos - (x->nseq ? x->seq_starts[lo] : 0);
}
/* literal src/sequence_index.rs:115-141; the recursion (mid, hi) with mid == lo never shrinks */
int orc_seq_location_ref(const orc_index *x, uint64_t gpos, orc_pos *out) {
  uint64_t lo = 0, hi = x->nseq - 1;
  for (int guard = 0; guard < 200; guard++) {
    if (lo == hi) { out->seq_idx = lo; out->local_pos = gpos - x->seq_starts[lo]; return 0; }
    uint64_t mid = (lo + hi) / 2, ms = x->seq_starts[mid];
    if (ms > gpos) hi = mid;
    else if (ms < gpos) { if (mid == lo) return 1; lo = mid; }
    else { out->seq_idx = mid; out->local_pos = gpos - ms; return 0; }
  }
  return 1;
}

/* src/fm_index.rs:516-544 */
int orc_locate_string(const orc_index *x, const uint8_t *q, uint64_t len, uint64_t **gpos, orc_pos **pos,
                      uint64_t *nhits, orc_tally *t) {
  uint64_t sp, ep;
  if (orc_search_range(x, q, len, &sp, &ep, t)) return -1;
  uint64_t n = sp > ep ? 0 : ep - sp + 1;
  uint64_t *g = malloc((n ? n : 1) * 8);
  orc_pos *p = malloc((n ? n : 1) * sizeof *p);
  if (!g || !p) { free(g); free(p); return -2; }
  for (uint64_t k = 0; k < n; k++) {
    uint64_t row = sp + k, steps = 0, v = 0;
    while (row % x->sa_ratio != 0) { row = orc_backstep(x, row); steps++; } /* src/compressed_suffix_array.rs:109-111 */
    orc_csa_reconstruct(x->sa_words, x->sa_bits, x->sa_ratio, row, &v);
    g[k] = (v + steps) % x->bwt_len;                                          /* src/fm_index.rs:534 */
    orc_seq_location(x, g[k], &p[k]);
    if (t) { t->backsteps += steps; t->hits++; }
  }
  *gpos = g; *pos = p; *nhits = n;
  return 0;
}

/* ------------------------------------------------------------------ batch (rayon stand-in) */

typedef struct {
  const orc_index *x; const uint8_t *qb; const uint64_t *qo; uint64_t n;
  uint64_t *counts; uint64_t **g; orc_pos **p; uint64_t *nh;
  uint64_t next; int err; orc_tally tally; pthread_mutex_t mu; int locate;
} batch_ctx;

static void *batch_worker(void *arg) {
  batch_ctx *c = arg;
  orc_tally t = {0};
  const uint64_t CH = 1024;
  for (;;) {
    uint64_t b = __atomic_fetch_add(&c->next, CH, __ATOMIC_RELAXED);
    if (b >= c->n) break;
    uint64_t e = b + CH < c->n ? b + CH : c->n;
    for (uint64_t i = b; i < e; i++) {
      const uint8_t *q = c->qb + c->qo[i]; uint64_t len = c->qo[i + 1] - c->qo[i];
      if (c->locate) {
        if (orc_locate_string(c->x, q, len, &c->g[i], &c->p[i], &c->nh[i], &t)) { c->err = 1; c->nh[i] = 0; }
      } else {
        uint64_t sp, ep;
        if (orc_search_range(c->x, q, len, &sp, &ep, &t)) { c->err = 1; c->counts[i] = 0; }
        else c->counts[i] = sp > ep ? 0 : ep - sp + 1;
      }
    }
  }
  pthread_mutex_lock(&c->mu);
  c->tally.queries += t.queries; c->tally.steps += t.steps; c->tally.block_reads += t.block_reads;
  c->tally.backsteps += t.backsteps; c->tally.hits += t.hits;
  pthread_mutex_unlock(&c->mu);
  return NULL;
}

static void run_batch(batch_ctx *c, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  pthread_mutex_init(&c->mu, NULL);
  pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nthreads);
  for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, batch_worker, c);
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  free(th);
  pthread_mutex_destroy(&c->mu);
}

/* src/fm_index.rs:455-460: results in input order */
int orc_parallel_count(const orc_index *x, const uint8_t *qbytes, const uint64_t *qoff, uint64_t n,
                       uint64_t *counts, int nthreads, orc_tally *tally) {
  batch_ctx c = {0};
  c.x = x; c.qb = qbytes; c.qo = qoff; c.n = n; c.counts = counts;
  run_batch(&c, nthreads);
  if (tally) *tally = c.tally;
  return c.err ? -1 : 0;
}

/* src/fm_index.rs:479-487: outer order = input order, inner order = ascending BWT row */
int orc_parallel_locate(const orc_index *x, const uint8_t *qbytes, const uint64_t *qoff, uint64_t n,
                        uint64_t **hit_off, uint64_t **gpos, orc_pos **pos, int nthreads, orc_tally *tally) {
  batch_ctx c = {0};
  c.x = x; c.qb = qbytes; c.qo = qoff; c.n = n; c.locate = 1;
  c.g = calloc(n ? n : 1, sizeof(uint64_t *)); c.p = calloc(n ? n : 1, sizeof(orc_pos *)); c.nh = calloc(n ? n : 1, 8);
  run_batch(&c, nthreads);
  uint64_t *off = malloc((n + 1) * 8);
  off[0] = 0;
  for (uint64_t i = 0; i < n; i++) off[i + 1] = off[i] + c.nh[i];
  uint64_t tot = off[n];
  uint64_t *g = malloc((tot ? tot : 1) * 8);
  orc_pos *p = malloc((tot ? tot : 1) * sizeof *p);
  for (uint64_t i = 0; i < n; i++) {
    if (c.nh[i]) { memcpy(g + off[i], c.g[i], c.nh[i] * 8); memcpy(p + off[i], c.p[i], c.nh[i] * sizeof *p); }
    free(c.g[i]); free(c.p[i]);
  }
  free(c.g); free(c.p); free(c.nh);
  *hit_off = off; *gpos = g; *pos = p;
  if (tally) *tally = c.tally;
  return c.err ? -1 : 0;
}

/* ------------------------------------------------------------------ .awry v1 file format */

static const char MAGIC[11] = "AWRY-Index\n"; /* src/fm_index_file.rs:18 (11 bytes, no NUL) */

/* src/fm_index_file.rs:42-106; src/sequence_index.rs:144-152 */
int orc_index_save(const orc_index *x, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  uint64_t hdr[4] = {x->version, x->sa_ratio, x->bwt_len, (uint64_t)x->alphabet}; /* :165-181 */
  fwrite(MAGIC, 1, 11, f);
  fwrite(hdr, 8, 4, f);
  fwrite(x->blocks, 8, x->nblocks * x->block_words, f);
  fwrite(x->prefix_sums, 8, orc_cardinality(x->alphabet) + 1u, f);
  fwrite(x->sa_words, 8, x->n_sa_words, f);
  fwrite(&x->kmer_len, 1, 1, f);
  fwrite(x->kmer_table, 16, x->n_kmer, f);
  uint64_t ns = x->nseq;
  fwrite(&ns, 8, 1, f);
  for (uint64_t i = 0; i < ns; i++) {
    uint64_t hl = strlen(x->headers[i]);
    fwrite(&x->seq_starts[i], 8, 1, f);
    fwrite(&hl, 8, 1, f);
    fwrite(x->headers[i], 1, hl, f);
  }
  int bad = ferror(f);
  return fclose(f) || bad ? -1 : 0;
}

#define RD(ptr, sz, cnt) do { if (fread(ptr, sz, cnt, f) != (size_t)(cnt)) goto fail; } while (0)
/* src/fm_index_file.rs:132-287; src/kmer_lookup_table.rs:55-77; src/sequence_index.rs:154-183 */
orc_index *orc_index_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  orc_index *x = calloc(1, sizeof *x);
  char magic[11]; uint64_t hdr[4];
  RD(magic, 1, 11);
  if (memcmp(magic, MAGIC, 11)) goto fail;
  RD(hdr, 8, 4);
  x->version = hdr[0]; x->sa_ratio = hdr[1]; x->bwt_len = hdr[2];
  if (hdr[3] > 1 || x->sa_ratio == 0 || x->bwt_len == 0) goto fail;
  x->alphabet = (int)hdr[3];
  set_shape(x);
  x->blocks = malloc(x->nblocks * x->block_words * 8);
  x->sa_words = malloc((x->n_sa_words ? x->n_sa_words : 1) * 8);
  if (!x->blocks || !x->sa_words) goto fail;
  RD(x->blocks, 8, x->nblocks * x->block_words);
  RD(x->prefix_sums, 8, orc_cardinality(x->alphabet) + 1u);
  RD(x->sa_words, 8, x->n_sa_words);
  RD(&x->kmer_len, 1, 1);
  x->n_kmer = 1;
  for (unsigned i = 0; i < x->kmer_len; i++) x->n_kmer *= num_encoding_symbols(x->alphabet);
  x->kmer_table = malloc(x->n_kmer * 16);
  if (!x->kmer_table) goto fail;
  RD(x->kmer_table, 16, x->n_kmer);
  uint64_t ns;
  RD(&ns, 8, 1);
  x->nseq = ns;
  x->seq_starts = malloc((ns ? ns : 1) * 8);
  x->headers = calloc(ns ? ns : 1, sizeof(char *));
  for (uint64_t i = 0; i < ns; i++) {
    uint64_t hl;
    RD(&x->seq_starts[i], 8, 1);
    RD(&hl, 8, 1);
    x->headers[i] = calloc(hl + 1, 1);
    RD(x->headers[i], 1, hl);
  }
  fclose(f);
  return x;
fail:
  fclose(f);
  orc_index_free(x);
  return NULL;
}
