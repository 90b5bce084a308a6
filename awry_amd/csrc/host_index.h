// host_index.h -- host-side owner of the index content (everything FmIndex holds in the reference,
// /root/reference src/fm_index.rs:41-56), kept in the DEVICE block layout of layout.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "layout.h"

namespace awry {

struct HostIndex {
  int alphabet = NUCLEOTIDE;
  uint64_t bwt_len = 0;            // text length incl. '$', src/fm_index.rs:50,182
  uint64_t version = 1;            // src/fm_index.rs:19
  uint64_t sa_ratio = 8;           // src/fm_index.rs:122
  uint64_t sa_bits = 0;            // src/compressed_suffix_array.rs:124-130
  uint8_t kmer_len = 10;           // the reference's lookup_table_kmer_len (file-format field)
  uint64_t sentinel_row = 0;       // BWT row holding '$'
  uint64_t nblocks = 0;
  std::vector<uint64_t> prefix_sums;  // cardinality + 1
  std::vector<uint64_t> blocks;       // nblocks * block_words(alphabet), device layout
  std::vector<uint64_t> sa_words;     // packed sampled SA
  std::vector<uint64_t> seq_starts;
  std::vector<std::string> headers;
  // the reference's k-mer table content ((start,end) pairs, src/kmer_lookup_table.rs:17-20): kept when
  // loaded from a file, otherwise computed on the GPU when the index is saved
  std::vector<uint64_t> ref_kmer_table;
};

struct SequenceFile {
  std::vector<uint8_t> text;  // records joined by the delimiter, terminated by '$'
  std::vector<uint64_t> starts;
  std::vector<std::string> headers;
};

uint64_t csa_bits_per_element(uint64_t bwt_len);
uint64_t csa_word_len(uint64_t bwt_len, uint64_t ratio);
uint64_t ref_kmer_table_entries(int alphabet, unsigned kmer_len);

// FASTA / FASTQ -> text model of src/fm_index.rs:148-153 (throws std::runtime_error)
SequenceFile read_sequence_file(const std::string& path, int alphabet);

// FASTA / FASTQ -> one query per record, CSR (bytes, offsets[n+1]); malloc'ed arrays owned by the caller
void read_query_file(const std::string& path, uint8_t** bytes_out, uint64_t** offsets_out, uint64_t* n_out);

// The single pass over the suffix array of src/fm_index.rs:203-240, emitting the device layout.
// `sa` may be u32 or u64 values (sa32 != nullptr selects u32).
void pack_index(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, const uint64_t* sa64,
                const uint32_t* sa32);

// validates the arguments and fills the metadata fields (alphabet, ratios, records); no suffix array yet
void prepare_build(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio,
                   unsigned kmer_len, const uint64_t* seq_starts, const char* const* headers, uint64_t nseq);

// text (ending in '$') -> index, host suffix array (sais.hpp)
void build_from_text(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio,
                     unsigned kmer_len, const uint64_t* seq_starts, const char* const* headers, uint64_t nseq);

// same result as the suffix-array + pack_index part of build_from_text, computed on a GPU (sa_builder.hip);
// ix.alphabet / sa_ratio must be set; needs bwt_len < 2^32 - 1
void gpu_build_index(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, int device, bool verbose);

// reference block layout (planes then milestones, src/fm_index_file.rs:58-67) <-> device layout
void block_to_reference(const HostIndex& ix, uint64_t b, uint64_t* out /* 20 or 44 words */);
void block_from_reference(HostIndex& ix, uint64_t b, const uint64_t* in);

// the reference's k-mer table content (src/kmer_lookup_table.rs:121-167) computed on the host copy: what awry_save
// writes when the index has no GPU replica (with one, ref_kmer_table_kernel fills it)
void fill_ref_kmer_table_host(HostIndex& ix);

// .awry v1 (src/fm_index_file.rs:42-106,132-287); save needs ix.ref_kmer_table filled
void save_awry(const HostIndex& ix, const std::string& path);
void load_awry(HostIndex& ix, const std::string& path);

}  // namespace awry
