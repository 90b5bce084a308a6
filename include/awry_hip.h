/*
 * awry_hip.h -- C ABI of libawry_hip.so: the MI355X (gfx950) FM-index search engine that stands in for
 * the query path of the Rust crate AWRY 0.3.1.  Plain pointers and sizes only; no exception crosses.
 *
 * Each entry point names the reference interface it replaces (file:line under /root/reference).  The
 * reference has no FFI of its own -- its boundary is the `pub` surface of `FmIndex` (src/fm_index.rs) --
 * so a Rust shim crate re-creating that surface binds exactly these symbols.
 *
 * There is NO CPU search path in this library: every query entry point runs HIP kernels on the devices
 * selected with awry_set_devices() and fails with AWRY_ERR_NO_DEVICE / AWRY_ERR_HIP otherwise.
 */
#ifndef AWRY_HIP_H
#define AWRY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (0 = OK, < 0 = error class); awry_last_error() has the message */
enum {
  AWRY_OK = 0,
  AWRY_ERR_IO = -1,            /* file could not be opened / read / written                              */
  AWRY_ERR_FORMAT = -2,        /* not an .awry v1 file, corrupt header, unsupported size                 */
  AWRY_ERR_INVALID_QUERY = -3, /* empty query, '$' / '#', byte >= 0x80: the reference panics or is UB     */
  AWRY_ERR_HIP = -4,           /* a HIP runtime call or kernel failed                                    */
  AWRY_ERR_OOM = -5,
  AWRY_ERR_ARG = -6,           /* null pointer, bad alphabet id, bad device id ...                       */
  AWRY_ERR_NO_DEVICE = -7      /* query issued before awry_set_devices(), or no GPU present              */
};

enum { AWRY_NUCLEOTIDE = 0, AWRY_AMINO = 1 }; /* SymbolAlphabet, src/alphabet.rs:28-31,48-61 */

typedef struct awry_index awry_index_t; /* opaque: host copy of the index + one replica per selected GPU */

/* == LocalizedSequencePosition, src/sequence_index.rs:31-35 */
typedef struct { uint64_t seq_idx, local_pos; } awry_pos_t;

/* == SearchRange, src/search.rs:25-28 (closed interval of BWT rows; empty iff start_ptr > end_ptr) */
typedef struct { uint64_t start_ptr, end_ptr; } awry_range_t;

/* == FmBuildArgs, src/fm_index.rs:78-96 */
typedef struct {
  const char *input_path;  /* input_file_src: FASTA or FASTQ                                              */
  const char *sa_tmp_path; /* suffix_array_output_src: accepted for signature parity, unused (no .sufr)   */
  uint64_t sa_ratio;       /* suffix_array_compression_ratio, 0 => 8 (src/fm_index.rs:122)                */
  uint8_t kmer_len;        /* lookup_table_kmer_len, 0 => 10 nt / 4 aa (src/kmer_lookup_table.rs:23-24)   */
  uint8_t alphabet;        /* AWRY_NUCLEOTIDE / AWRY_AMINO                                                */
  uint64_t max_query_len;  /* accepted for signature parity; the full suffix array is always built        */
  uint8_t remove_tmp;      /* remove_intermediate_suffix_array_file: no intermediate file exists          */
} awry_build_args_t;

/* ---- construction / persistence ------------------------------------------------------------------ */
/* FmIndex::new, src/fm_index.rs:142-268 */
int awry_build(const awry_build_args_t *args, awry_index_t **out);
/* same, from an in-memory text that follows the reference's text model (records joined by 'N'/'X', one
 * trailing '$', src/fm_index.rs:148-153,220-223); seq_starts/headers describe the records */
int awry_build_from_text(const uint8_t *text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio, uint8_t kmer_len,
                         const uint64_t *seq_starts, const char *const *headers, uint64_t nseq, awry_index_t **out);
/* same with an explicit choice of where the suffix array and the BWT are constructed: a device id >= 0
 * (GPU prefix doubling + streaming pack kernels), AWRY_BUILD_HOST (host SA-IS) or AWRY_BUILD_AUTO (what
 * awry_build / awry_build_from_text use: the calling thread's current GPU if one is visible and bwt_len >= 2^20, else host; env AWRY_BUILD=host|gpu
 * overrides; when the GPU chosen this way cannot do it -- its HBM taken by replicas, say -- the host builder takes over, while
 * an explicit device id fails with AWRY_ERR_HIP).  All choices produce bit-identical indexes. */
enum { AWRY_BUILD_HOST = -1, AWRY_BUILD_AUTO = -2 };
int awry_build_from_text_on(const uint8_t *text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio, uint8_t kmer_len,
                            const uint64_t *seq_starts, const char *const *headers, uint64_t nseq, int build_device,
                            awry_index_t **out);
/* FmIndex::load / FmIndex::save, src/fm_index_file.rs:132,42 (.awry v1, byte-compatible) */
int awry_load(const char *path, awry_index_t **out);
int awry_save(awry_index_t *idx, const char *path); /* needs no device (the k-mer table is filled on a replica's GPU when there is one, else on the host) */
void awry_free(awry_index_t *idx);

/* ---- device placement (replaces rayon's global pool, src/fm_index.rs:455-487) ---------------------- */
/* replicate the index into the HBM of each listed device and build its seed table; batches are then
 * sharded contiguously over the replicas.  n_devices == 0 is an error: there is no CPU backend.  Replaces (and first
 * releases) any earlier replicas.  HBM use: the default policies size the seed table to 70 % of the free HBM and keep
 * the locate / verify accelerators (7 B per text symbol) when they fit in half of it -- ~160 GB for a GRCh38-scale
 * index; AWRY_HBM_BUDGET_GB caps the figure they plan with, AWRY_SEED_K / AWRY_VERIFY=0 pin the choices.  Later, the first
 * batch of >= 4096 nucleotide k-mers of a length L below the seed table's k adds a complete 4^L table for that length
 * (8 B x 4^L: 134 MB for L = 12, 34 GB for L = 16) when it fits half of the free HBM and all such tables of the replica stay
 * below AWRY_SEED_RUNG_GB (default 48); that first call builds the table (1.3 s for L = 16) while it holds the replica's
 * lane; AWRY_SEED_RUNGS=0 turns the feature off.
 * If building a replica fails (out of HBM on one of several GPUs, say) the index is left with NO replicas -- the old ones are
 * released first, because the policies size their tables from the HBM that is free -- and every query returns
 * AWRY_ERR_NO_DEVICE until a later awry_set_devices succeeds; do not call it while another thread is inside a batch call. */
int awry_set_devices(awry_index_t *idx, const int *device_ids, int n_devices);
/* device-side seed-table length (performance knob only; results do not depend on it).  0 disables,
 * -1 picks the default.  Takes effect immediately on all replicas. */
int awry_set_seed_kmer_len(awry_index_t *idx, int k);
int awry_seed_kmer_len(const awry_index_t *idx);
/* A/B switch for the packed-k-mer count kernel (process-wide; -1 policy, 0 strided quads, 1 LDS-staged chunks,
 * 2 groups of four queries per quad, 3 two-phase probe + resume).  All variants return identical counts. */
int awry_debug_set_count_kernel(int mode);
/* Indexes of 2^32 rows or more take "wide-row" packed kernels (64-bit rows, 16-byte seed entries, no 32-bit accelerators;
 * the reference is u64 throughout, src/search.rs:7).  on != 0 makes replicas placed AFTERWARDS (awry_set_devices) take
 * those kernels whatever the index size -- for tests: a real index of that size takes an hour of host SA-IS to build. */
int awry_debug_force_wide_rows(int on);
/* device pointer of replica `slot`'s dense SA (u32 SA[j * ratio]) or NULL -- for tests that dump it */
const void *awry_debug_dense_sa(const awry_index_t *idx, int slot);
/* name(s) of the kernel(s) awry_dev_count_nt2 launches for k-mers of length L on replica 0 (for profiling reports) */
const char *awry_count_schedule(const awry_index_t *idx, int L);
/* device-side SA sampling used by locate (performance knob only; locations do not depend on it): 0 = walk to the
 * file's row samples (suffix_array_compression_ratio, default), r >= 1 = additionally keep SA[j r] as u32 in HBM
 * (r = 1: one read per hit, no LF walk; GRCh38: 12.4 GB).  Needs bwt_len < 2^32. */
int awry_set_locate_sa_ratio(awry_index_t *idx, int ratio);
int awry_locate_sa_ratio(const awry_index_t *idx);
/* seed-and-verify for packed nucleotide reads (performance knob only; counts and locations do not depend on it):
 * keeps the ratio-1 dense SA and the text as 4-bit codes in HBM (GRCh38: 12.4 + 1.55 GB), both recovered from the
 * index on the device.  Once a range holds a single row -- or <= 8 rows after `after_steps` LF steps -- the rest of the
 * query is compared with the text in front of each candidate instead of being matched by one dependent LF step per
 * symbol.  after_steps = -1 switches it off.  Default policy (awry_set_devices): on with after_steps = 2 for nucleotide
 * indexes with bwt_len < 2^32 whose accelerators fit in half of the free HBM; env AWRY_VERIFY=0 disables, =N sets
 * after_steps.  Used by the read kernels (awry_dev_count_nt2_long, the host paths) and by the two-phase k-mer schedule;
 * the single-kernel k-mer schedule (dense seed tables) uses it only after awry_set_verify_kmers(idx, 1), because there
 * the extra state costs random batches ~15 %. */
int awry_set_verify(awry_index_t *idx, int after_steps);
int awry_set_verify_kmers(awry_index_t *idx, int on);
int awry_verify_enabled(const awry_index_t *idx);
/* left-context index (performance knob only; counts and locations do not depend on it): with the seed-and-verify accelerators
 * resident, every seed bucket of 2+ rows keeps the 32 letters in FRONT of each of its suffixes, sorted, plus each suffix's text
 * position and BWT row (16.6 B per row of the index: 51 GB for GRCh38).  The letters a query has left of its seed window are
 * then matched by a 16-ary search over its bucket -- log16(rows) random lines -- instead of one LF step (two lines) per letter:
 * what makes k-mers and reads from repeat families (10^3..10^5 rows per seed, dozens of letters before the rows part) cost a
 * few lines like any other query.  Default policy (awry_set_devices): built when it and its build scratch fit 3/4 of the HBM
 * still free once everything else is resident; env AWRY_LCX=0 / on = 0 switch it off (rebuilds the seed table), on = 1
 * restores the policy.  awry_lcx_enabled: is it resident on replica 0. */
int awry_set_lcx(awry_index_t *idx, int on);
int awry_lcx_enabled(const awry_index_t *idx);
/* device pointers of replica `slot`'s left-context index for tests that dump it: keys[bwt_len] (u64) and
 * rowpos[bwt_len] (u64: text position | BWT row << 32); NULL when it is not resident */
int awry_debug_lcx(const awry_index_t *idx, int slot, const void **d_keys, const void **d_rowpos);
int awry_num_devices(const awry_index_t *idx);

/* ---- batch queries --------------------------------------------------------------------------------- */
/* FmIndex::parallel_count, src/fm_index.rs:455-460.  Query i = qbytes[qoff[i] .. qoff[i+1]); results in
 * input order in caller-owned counts_out[n].  Any undefined query => AWRY_ERR_INVALID_QUERY. */
int awry_count_batch(awry_index_t *idx, const uint8_t *qbytes, const uint64_t *qoff, uint64_t n, uint64_t *counts_out);
/* no counterpart in the reference: the same for callers that already hold their k-mers packed (k-mer counters do) --
 * n k-mers of L <= 32 letters, letter j (0 = leftmost) of k-mer i in bits [2j, 2j + 2) of words[i], A0 C1 G2 T3;
 * nucleotide indexes of any size (2^32 rows or more: the wide-row kernels).  16 B per query cross PCIe instead of L + 8. */
int awry_count_packed_kmers(awry_index_t *idx, const uint64_t *words, uint64_t n, int L, uint64_t *counts_out);
/* FmIndex::parallel_locate, src/fm_index.rs:479-487.  CSR output, library-allocated (awry_free_buffer):
 * hits of query i are [hit_off[i], hit_off[i+1]) in ascending BWT-row order (src/fm_index.rs:521);
 * global_pos (nullable) receives (SA sample + steps) % bwt_len (src/fm_index.rs:534).  hits_out is nullable too: a
 * caller that wants text positions only (8 B per hit over PCIe instead of 24) passes NULL and (record, offset) pairs
 * are neither computed nor moved; with both NULL the call returns the offsets alone. */
int awry_locate_batch(awry_index_t *idx, const uint8_t *qbytes, const uint64_t *qoff, uint64_t n,
                      uint64_t **hit_off_out, awry_pos_t **hits_out, uint64_t **global_pos_out);
/* releases an array one of the calls above (or awry_locate / awry_read_query_file) returned.  Result arrays are pinned
 * host memory recycled through a process-wide pool (the device writes results straight into them); never pass them to
 * free().  AWRY_PINNED_CACHE_GB (default 4) bounds what the pool keeps between calls. */
void awry_free_buffer(void *p);

/* ---- scalar conveniences (each launches on replica 0) ---------------------------------------------- */
int awry_count(awry_index_t *idx, const uint8_t *q, uint64_t len, uint64_t *count);          /* count_string :499  */
/* get_search_range_for_string (pub(crate) in the reference, :402-438): the row interval of q.  It follows the reference's own
 * step schedule -- no device seed table; for len >= lookup_table_kmer_len the first kmer_len - 1 steps are taken whether or
 * not the range is already empty (src/kmer_lookup_table.rs:90-110) -- so an ABSENT query returns the same (start_ptr, end_ptr)
 * as the reference (start_ptr = end_ptr + 1, not a canonical {1, 0}); pinned against the oracle in tests/test_gpu_parity.py */
int awry_search_range(awry_index_t *idx, const uint8_t *q, uint64_t len, awry_range_t *out);
int awry_locate(awry_index_t *idx, const uint8_t *q, uint64_t len, awry_pos_t **hits_out,
                uint64_t **global_pos_out, uint64_t *n_hits);                                /* locate_string :516 */
int awry_initial_range(const awry_index_t *idx, uint8_t symbol_ascii, awry_range_t *out);    /* :383-385          */
int awry_update_range(awry_index_t *idx, awry_range_t in, uint8_t symbol_ascii, awry_range_t *out); /* :559-582   */
int awry_backstep(awry_index_t *idx, uint64_t row, uint64_t *out);                           /* :585-593          */
int awry_get_seq_location(const awry_index_t *idx, uint64_t global_pos, awry_pos_t *out);    /* sequence_index.rs:108 */

/* ---- accessors (src/fm_index.rs:302-399) ------------------------------------------------------------ */
int awry_alphabet(const awry_index_t *idx);
uint64_t awry_bwt_len(const awry_index_t *idx);
uint64_t awry_version(const awry_index_t *idx);
uint64_t awry_sa_ratio(const awry_index_t *idx);
uint8_t awry_kmer_len(const awry_index_t *idx);
const uint64_t *awry_prefix_sums(const awry_index_t *idx, uint64_t *len);
uint64_t awry_num_sequences(const awry_index_t *idx);
uint64_t awry_sequence_start(const awry_index_t *idx, uint64_t i);
const char *awry_sequence_header(const awry_index_t *idx, uint64_t i);
uint64_t awry_sentinel_row(const awry_index_t *idx);
/* device-layout BWT blocks and packed SA words of the host copy (layout.h); for tests and tooling */
const uint64_t *awry_block_words(const awry_index_t *idx, uint64_t *nwords);
const uint64_t *awry_sa_words(const awry_index_t *idx, uint64_t *nwords);
/* one block converted to the reference layout (planes, then 8 / 24 milestones), src/bwt.rs:12-25 */
int awry_block_reference_layout(const awry_index_t *idx, uint64_t block, uint64_t *out, uint64_t out_words);

const char *awry_last_error(void); /* thread-local message of the last non-zero status */

/* ---- host utilities ----------------------------------------------------------------------------------- */
/* query ingestion: every record of a FASTA / FASTQ file becomes one query; library-allocated CSR arrays
 * (awry_free_buffer) ready for awry_count_batch / awry_locate_batch */
int awry_read_query_file(const char *path, uint8_t **qbytes_out, uint64_t **qoff_out, uint64_t *n_out);
/* suffix array of a byte text ending in '$' (host SA-IS; stands in for libsufr, src/fm_index.rs:156-181) */
int awry_host_suffix_array(const uint8_t *text, uint64_t n, uint64_t *sa_out);
uint8_t awry_symbol_index(int alphabet, uint8_t ascii); /* Symbol::new_ascii(..).index(), src/alphabet.rs:109,152 */
/* the host half of awry_count_batch / awry_locate_batch for nucleotide batches (caller side of src/fm_index.rs:455-487):
 * n ASCII queries -> 2-bit words on the library's worker pool (AVX2), so that 8 B per 31-mer cross PCIe instead of 31.
 * qoff == NULL: n queries of L bytes back to back; else query i = qbytes[qoff[i] .. qoff[i+1]) with 1..L letters and
 * lens_out[i] receives its length.  W = ceil(L / 32) words per query (letter j in word j / 32, bits 2 (j % 32), A0 C1
 * G2 T3, unused bits zero).  Queries holding a byte outside ACGTacgt are listed (ascending) in bad_out[0 .. *nbad_out)
 * (room for n; nullable) -- the batch paths hand those to the generic kernel.  Needs no GPU; searches nothing. */
int awry_host_pack_nt2(const uint8_t *qbytes, const uint64_t *qoff, uint64_t n, uint64_t L, uint64_t *words_out,
                       uint32_t *lens_out, uint32_t *bad_out, uint64_t *nbad_out);
int awry_host_threads(void); /* size of that pool: the CPUs this process may use (cgroup quota / affinity; AWRY_HOST_THREADS) */
void awry_host_memcpy(void *dst, const void *src, uint64_t bytes); /* memcpy cut over that pool (what copies results out) */

/* ---- device-resident API: pointers are device memory on replica `slot`'s GPU, work is queued on
 *      `stream` (a hipStream_t, NULL = default stream) and NOT synchronised.  Scratch (survivor lists, work-queue
 *      heads) is kept per stream, so any number of launches may be in flight across streams.  Query byte buffers
 *      (d_qbytes, d_ascii) are read in aligned 8-byte words: they must be readable up to 8 bytes past the last
 *      query (allocations of awry_dev_malloc and hipMalloc are) ------------------------------------------------ */
int awry_replica_device(const awry_index_t *idx, int slot);
/* fixed-length ACGT reads, ASCII n*L bytes -> n * ceil(L/32) packed u64 words (letter j in word j/32, bits 2(j%32));
 * *d_bad (u64 on device, caller-zeroed) counts queries with other bytes */
int awry_dev_pack_nt2(awry_index_t *idx, int slot, const void *d_ascii, uint64_t n, int L, void *d_words,
                      void *d_bad, void *stream);
/* the hot kernel: count n packed k-mers -> u64 counts.  use_seed != 0 starts from the seed table */
int awry_dev_count_nt2(awry_index_t *idx, int slot, const void *d_words, uint64_t n, int L, void *d_counts,
                       int use_seed, void *stream);
/* same kernel with a work census for the roofline figure: d_tally[6] (u64, caller-zeroed) += {seed probes,
 * executed steps, distinct BWT blocks ranked, SA reads and text windows of seed-and-verify, blocks ranked by steps
 * after a query's first 10 (single-kernel schedule only: the ones whose lines no longer sit in the Infinity Cache), and --
 * caller passes d_tally[8] -- nodes of the left-context index consulted, its (position, row) entries read} --
 * the first three are the tallies SURVEY.md 8(d) prices at 16 B / 104 B each */
int awry_dev_count_nt2_tally(awry_index_t *idx, int slot, const void *d_words, uint64_t n, int L, void *d_counts,
                             int use_seed, void *d_tally, void *stream);
/* generic path: ASCII queries + u64 offsets[n+1] -> counts[n], optional ranges[2n] (start,end) and status[n] bytes */
int awry_dev_count_ascii(awry_index_t *idx, int slot, const void *d_qbytes, const void *d_qoff, uint64_t n,
                         void *d_counts, void *d_ranges, void *d_status, void *stream);
/* the count pass of a device-resident parallel_locate: as above, but d_locate_words[2n] receives what awry_dev_locate
 * (range_stride 2) needs per query -- a row interval, or the text position(s) the count pass already verified when the
 * seed-and-verify accelerators are resident -- instead of row intervals, which lets the fast schedules run (amino k-mers:
 * 10x the rate of awry_dev_count_ascii with ranges).  The words are opaque: feed them to awry_dev_locate, nothing else. */
int awry_dev_count_ascii_for_locate(awry_index_t *idx, int slot, const void *d_qbytes, const void *d_qoff, uint64_t n,
                                    void *d_counts, void *d_locate_words, void *d_status, void *stream);
/* n ASCII queries of `len` bytes each, back to back (no offsets) -> counts[n], optional status[n].  Nucleotide
 * indexes: packed on the device and served by the packed kernels (queries with letters outside ACGT are redone by the
 * generic kernel).  Amino queries of 8..1024 residues: a two-phase schedule of their own (one query per lane against the
 * seed table and the text -- the last 24 residues in registers, the rest compared with the text for surviving candidates --
 * and the generic kernel on the few it cannot decide).  Every other shape: the generic kernel
 * reading query q at q * len.  Scratch lives in the replica, per stream. */
int awry_dev_count_ascii_uniform(awry_index_t *idx, int slot, const void *d_qbytes, uint64_t n, uint64_t len,
                                 void *d_counts, void *d_status, void *stream);
/* the amino k-mer schedule of awry_dev_count_ascii_uniform with a work census: d_tally[5] (u64, caller-zeroed) +=
 * {seed probes, executed steps, distinct BWT blocks ranked, SA reads, text comparisons} over both of its passes
 * (168 B per block); slower than the plain call (per-event atomics) -- for untimed runs */
int awry_dev_count_ascii_uniform_tally(awry_index_t *idx, int slot, const void *d_qbytes, uint64_t n, uint64_t len,
                                       void *d_counts, void *d_tally, void *stream);
/* exclusive scan of counts[n] -> hit_off[n+1] (d_scratch: awry_dev_scan_scratch_bytes(n) bytes) */
uint64_t awry_dev_scan_scratch_bytes(uint64_t n);
int awry_dev_scan_counts(awry_index_t *idx, int slot, const void *d_counts, uint64_t n, void *d_hit_off,
                         void *d_scratch, void *stream);
/* backtrace `total` hits: d_ranges[q * range_stride] = first row of query q's range (stride 2 = the (start,end)
 * pairs of awry_dev_count_ascii, stride 1 = the starts of awry_dev_count_nt2_long), hit_off[n+1] ->
 * global_pos[total], pos[total] (nullable) */
int awry_dev_locate(awry_index_t *idx, int slot, const void *d_ranges, int range_stride, const void *d_hit_off, uint64_t n,
                    uint64_t total, void *d_global_pos, void *d_pos, void *stream);
/* awry_dev_locate with the walk kernel's census: d_tally[2] (u64, caller-zeroed) += {LF steps taken by the walks, hits
 * that had to walk} -- what SURVEY.md 8(d) prices at 104 B per backstep.  Nucleotide indexes. */
int awry_dev_locate_tally(awry_index_t *idx, int slot, const void *d_ranges, int range_stride, const void *d_hit_off, uint64_t n,
                          uint64_t total, void *d_global_pos, void *d_pos, void *d_tally, void *stream);
/* profiling aid: queues an empty kernel (phase_marker_kernel) of phase_id blocks of 64 threads on `stream`.  rocprofv3
 * counter passes report the grid size of every dispatch, so a run can be cut into named phases without marker tracing. */
int awry_dev_phase_marker(awry_index_t *idx, int slot, int phase_id, void *stream);
/* packed reads of any length: W = ceil(L/32) u64 words per query (letter j in word j/32, bits 2(j%32), as written by
 * awry_dev_pack_nt2); counts[n] and, if non-null, range_start[n] (first BWT row of each range) for awry_dev_locate */
int awry_dev_count_nt2_long(awry_index_t *idx, int slot, const void *d_words, uint64_t n, int L, void *d_counts,
                            void *d_range_start, int use_seed, void *stream);
/* plumbing for callers without a HIP binding of their own */
int awry_dev_malloc(awry_index_t *idx, int slot, uint64_t bytes, void **d_out);
int awry_dev_free(awry_index_t *idx, int slot, void *d);
int awry_dev_memcpy_h2d(awry_index_t *idx, int slot, void *d_dst, const void *h_src, uint64_t bytes);
int awry_dev_memcpy_d2h(awry_index_t *idx, int slot, void *h_dst, const void *d_src, uint64_t bytes);
int awry_dev_memset(awry_index_t *idx, int slot, void *d_dst, int value, uint64_t bytes);
int awry_dev_synchronize(awry_index_t *idx, int slot);
/* measurement aid: copies `bytes` (multiple of 16) from d_src to d_dst with 16-byte loads and stores per lane on `stream` --
 * the streaming rate printed next to the nominal HBM peak */
int awry_dev_stream_copy(awry_index_t *idx, int slot, void *d_dst, const void *d_src, uint64_t bytes, void *stream);
/* time a region on `stream` with HIP events: begin/end record, elapsed synchronises and returns ms */
int awry_dev_timer_begin(awry_index_t *idx, int slot, void *stream);
int awry_dev_timer_end(awry_index_t *idx, int slot, void *stream, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* AWRY_HIP_H */
