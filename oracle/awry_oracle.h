/*
 * awry_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference FM-index query path of AWRY 0.3.1
 * (the .rs files under /root/reference/src).  Every function cites the reference file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call this; the product (awry_amd/, libawry_hip.so) never does.
 *
 * Parity pin: the reference is Rust and cannot be built here (no cargo/rustc,
 * un-vendored crates libsufr 0.6.2 / rayon 1.10), so oracle/_ref does not
 * exist.  The oracle is pinned by (1) every deterministic known-answer test the
 * reference holds for this path (tests/test_oracle_kat.py restates them) and
 * (2) the definition the reference's integration tests pin
 * (src/fm_index.rs:612-664: count == #occurrences in text, sorted locate ==
 * occurrence positions), checked by brute force on seeded texts.
 */
#ifndef AWRY_ORACLE_H
#define AWRY_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_NUCLEOTIDE = 0, ORC_AMINO = 1 };
#define ORC_PANIC UINT64_MAX /* where the reference panics / is UB */

typedef struct orc_index orc_index;
typedef struct { uint64_t seq_idx, local_pos; } orc_pos;
typedef struct {
  uint64_t queries;        /* queries processed                                   */
  uint64_t steps;          /* executed update_range_with_symbol calls             */
  uint64_t block_reads;    /* distinct BWT blocks ranked, 1 or 2 per step         */
  uint64_t backsteps;      /* executed backstep calls (locate)                    */
  uint64_t hits;           /* located positions                                   */
} orc_tally;

/* ---- alphabet (src/alphabet.rs) ---- */
uint8_t orc_cardinality(int alphabet);
uint8_t orc_ascii_to_index(int alphabet, uint8_t ascii);
uint8_t orc_ascii_to_code(int alphabet, uint8_t ascii);
uint8_t orc_index_to_code(int alphabet, uint8_t idx);
uint8_t orc_code_to_index(int alphabet, uint8_t code);
uint8_t orc_index_to_ascii(int alphabet, uint8_t idx);
uint8_t orc_code_to_ascii(int alphabet, uint8_t code);

/* ---- Vec256 / blocks (src/simd_instructions.rs, src/bwt.rs) ---- */
uint32_t orc_masked_popcount(const uint64_t v[4], uint64_t pos);
void orc_block_set_symbol(uint64_t *planes, int nplanes, uint8_t code, uint64_t pos);
uint8_t orc_block_code_at(const uint64_t *planes, int nplanes, uint64_t pos);
uint64_t orc_nt_block_occ(const uint64_t planes[12], const uint64_t ms[8], uint64_t pos, uint8_t sym);
uint64_t orc_aa_block_occ(const uint64_t planes[20], const uint64_t ms[24], uint64_t pos, uint8_t sym);

/* ---- CompressedSuffixArray (src/compressed_suffix_array.rs) ---- */
uint64_t orc_csa_bits_per_element(uint64_t bwt_len);
uint64_t orc_csa_word_len(uint64_t bwt_len, uint64_t ratio);
void orc_csa_set_value(uint64_t *data, uint64_t bits, uint64_t value, uint64_t position);
int orc_csa_reconstruct(const uint64_t *data, uint64_t bits, uint64_t ratio, uint64_t position, uint64_t *out);

/* ---- definition-level checkers (what src/fm_index.rs:612-664 pins) ---- */
int orc_suffix_array(const uint8_t *text, uint64_t n, uint64_t *sa); /* text ends in '$' */
uint64_t orc_brute_count(int alphabet, const uint8_t *text, uint64_t n, const uint8_t *pat, uint64_t m);
uint64_t orc_brute_locate(int alphabet, const uint8_t *text, uint64_t n, const uint8_t *pat, uint64_t m,
                          uint64_t *out, uint64_t cap);

/* ---- FmIndex (src/fm_index.rs, src/kmer_lookup_table.rs, src/sequence_index.rs) ---- */
orc_index *orc_index_from_sa(const uint8_t *text, uint64_t bwt_len, const uint64_t *sa, int alphabet,
                             uint64_t sa_ratio, uint8_t kmer_len, const uint64_t *seq_starts,
                             const char *const *headers, uint64_t nseq);
orc_index *orc_index_build(const uint8_t *text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio,
                           uint8_t kmer_len, const uint64_t *seq_starts, const char *const *headers,
                           uint64_t nseq);
orc_index *orc_index_from_fasta(const char *path, int alphabet, uint64_t sa_ratio, uint8_t kmer_len);
void orc_index_free(orc_index *);
int orc_index_save(const orc_index *, const char *path);
orc_index *orc_index_load(const char *path);

int orc_alphabet(const orc_index *);
uint64_t orc_bwt_len(const orc_index *);
uint64_t orc_version(const orc_index *);
uint64_t orc_sa_ratio(const orc_index *);
uint8_t orc_kmer_len(const orc_index *);
const uint64_t *orc_prefix_sums(const orc_index *, uint64_t *len);
const uint64_t *orc_block_words(const orc_index *, uint64_t *nwords); /* reference layout */
const uint64_t *orc_sa_words(const orc_index *, uint64_t *nwords);
const uint64_t *orc_kmer_table(const orc_index *, uint64_t *nentries); /* (start,end) pairs */
uint64_t orc_num_sequences(const orc_index *);
uint64_t orc_seq_start(const orc_index *, uint64_t i);
const char *orc_seq_header(const orc_index *, uint64_t i);
const uint8_t *orc_text(const orc_index *); /* NULL for loaded indexes */

void orc_initial_range(const orc_index *, uint8_t sym_idx, uint64_t *sp, uint64_t *ep);
void orc_update_range(const orc_index *, uint64_t sp, uint64_t ep, uint8_t sym_idx, uint64_t *sp2, uint64_t *ep2);
uint64_t orc_backstep(const orc_index *, uint64_t p);
uint64_t orc_global_occurrence(const orc_index *, uint64_t p, uint8_t sym_idx);
uint8_t orc_symbol_at(const orc_index *, uint64_t p);

/* return 0 on success, -1 where the reference panics / is undefined (empty query, '$'/'#') */
int orc_search_range(const orc_index *, const uint8_t *q, uint64_t len, uint64_t *sp, uint64_t *ep, orc_tally *);
int orc_count_string(const orc_index *, const uint8_t *q, uint64_t len, uint64_t *count);
/* results malloc'ed; caller frees with orc_free.  gpos = (SA_sample + steps) % bwt_len */
int orc_locate_string(const orc_index *, const uint8_t *q, uint64_t len, uint64_t **gpos, orc_pos **pos,
                      uint64_t *nhits, orc_tally *);
void orc_free(void *);
/* intended semantics (largest i with start[i] <= pos) */
void orc_seq_location(const orc_index *, uint64_t gpos, orc_pos *out);
/* literal restatement of the reference recursion; returns 1 if it never terminates */
int orc_seq_location_ref(const orc_index *, uint64_t gpos, orc_pos *out);

/* order-preserving batch map (stand-in for rayon, src/fm_index.rs:455-487) */
int orc_parallel_count(const orc_index *, const uint8_t *qbytes, const uint64_t *qoff, uint64_t n,
                       uint64_t *counts, int nthreads, orc_tally *tally);
int orc_parallel_locate(const orc_index *, const uint8_t *qbytes, const uint64_t *qoff, uint64_t n,
                        uint64_t **hit_off, uint64_t **gpos, orc_pos **pos, int nthreads, orc_tally *tally);

#ifdef __cplusplus
}
#endif
#endif
