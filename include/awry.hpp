// awry.hpp -- header-only C++ host-side mirror of the reference's `FmIndex` surface
// (/root/reference src/fm_index.rs:41-119,142,302-399,455-544) over the C ABI of libawry_hip.so.
// Same names and argument meaning; where the reference panics the calls throw awry::Error.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "awry_hip.h"

namespace awry {

enum class SymbolAlphabet : uint8_t { Nucleotide = AWRY_NUCLEOTIDE, Amino = AWRY_AMINO };  // src/alphabet.rs:28-31

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* msg) : std::runtime_error(msg ? msg : "awry error"), code(c) {}
};
inline void check(int rc) { if (rc != AWRY_OK) throw Error(rc, awry_last_error()); }

struct FmBuildArgs {  // src/fm_index.rs:78-96
  std::string input_file_src;
  std::string suffix_array_output_src;            // Option<PathBuf>: empty = None
  uint64_t suffix_array_compression_ratio = 0;    // Option<u64>: 0 = None (=> 8)
  uint8_t lookup_table_kmer_len = 0;              // Option<u8>: 0 = None (=> 10 / 4)
  SymbolAlphabet alphabet = SymbolAlphabet::Nucleotide;
  uint64_t max_query_len = 0;                     // Option<usize>: 0 = None
  bool remove_intermediate_suffix_array_file = false;
};

struct LocalizedSequencePosition {  // src/sequence_index.rs:31-78
  uint64_t sequence_idx_, local_position_;
  uint64_t sequence_idx() const { return sequence_idx_; }
  uint64_t local_position() const { return local_position_; }
  bool operator<(const LocalizedSequencePosition& o) const {
    return sequence_idx_ != o.sequence_idx_ ? sequence_idx_ < o.sequence_idx_ : local_position_ < o.local_position_;
  }
  bool operator==(const LocalizedSequencePosition& o) const { return sequence_idx_ == o.sequence_idx_ && local_position_ == o.local_position_; }
};

struct SearchRange {  // src/search.rs:25-81
  uint64_t start_ptr = 1, end_ptr = 0;
  static SearchRange zero() { return {}; }
  bool is_empty() const { return start_ptr > end_ptr; }
  uint64_t len() const { return is_empty() ? 0 : end_ptr - start_ptr + 1; }
};

class FmIndex {
 public:
  // FmIndex::new, src/fm_index.rs:142.  `devices`: GPUs that receive a replica (queries need at least one).
  static FmIndex create(const FmBuildArgs& a, const std::vector<int>& devices = {0}) {
    awry_build_args_t c{};
    c.input_path = a.input_file_src.c_str();
    c.sa_tmp_path = a.suffix_array_output_src.empty() ? nullptr : a.suffix_array_output_src.c_str();
    c.sa_ratio = a.suffix_array_compression_ratio;
    c.kmer_len = a.lookup_table_kmer_len;
    c.alphabet = static_cast<uint8_t>(a.alphabet);
    c.max_query_len = a.max_query_len;
    c.remove_tmp = a.remove_intermediate_suffix_array_file;
    awry_index_t* h = nullptr;
    check(awry_build(&c, &h));
    FmIndex ix(h);
    ix.set_devices(devices);
    return ix;
  }
  // FmIndex::load, src/fm_index_file.rs:132
  static FmIndex load(const std::string& path, const std::vector<int>& devices = {0}) {
    awry_index_t* h = nullptr;
    check(awry_load(path.c_str(), &h));
    FmIndex ix(h);
    ix.set_devices(devices);
    return ix;
  }
  void save(const std::string& path) { check(awry_save(h_, path.c_str())); }  // src/fm_index_file.rs:42

  FmIndex(FmIndex&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  FmIndex& operator=(FmIndex&& o) noexcept { if (this != &o) { awry_free(h_); h_ = o.h_; o.h_ = nullptr; } return *this; }
  FmIndex(const FmIndex&) = delete;
  FmIndex& operator=(const FmIndex&) = delete;
  ~FmIndex() { awry_free(h_); }

  void set_devices(const std::vector<int>& ids) { if (!ids.empty()) check(awry_set_devices(h_, ids.data(), (int)ids.size())); }

  // accessors, src/fm_index.rs:302-399
  SymbolAlphabet alphabet() const { return static_cast<SymbolAlphabet>(awry_alphabet(h_)); }
  uint64_t suffix_array_compression_ratio() const { return awry_sa_ratio(h_); }
  uint64_t bwt_len() const { return awry_bwt_len(h_); }
  uint64_t version_number() const { return awry_version(h_); }
  std::vector<uint64_t> prefix_sums() const {
    uint64_t n = 0;
    const uint64_t* p = awry_prefix_sums(h_, &n);
    return std::vector<uint64_t>(p, p + n);
  }
  SearchRange initial_search_range(char symbol) const {
    awry_range_t r;
    check(awry_initial_range(h_, (uint8_t)symbol, &r));
    return {r.start_ptr, r.end_ptr};
  }

  // src/fm_index.rs:499, 516
  uint64_t count_string(std::string_view q) {
    uint64_t c = 0;
    check(awry_count(h_, (const uint8_t*)q.data(), q.size(), &c));
    return c;
  }
  std::vector<LocalizedSequencePosition> locate_string(std::string_view q) {
    awry_pos_t* hits = nullptr;
    uint64_t n = 0;
    check(awry_locate(h_, (const uint8_t*)q.data(), q.size(), &hits, nullptr, &n));
    std::vector<LocalizedSequencePosition> out(n);
    for (uint64_t i = 0; i < n; i++) out[i] = {hits[i].seq_idx, hits[i].local_pos};
    awry_free_buffer(hits);
    return out;
  }
  // src/fm_index.rs:455-460, 479-487 (results in input order; inner order = ascending BWT row)
  template <class StrRange>
  std::vector<uint64_t> parallel_count(const StrRange& queries) {
    std::vector<uint8_t> bytes; std::vector<uint64_t> off;
    pack(queries, bytes, off);
    std::vector<uint64_t> out(off.size() - 1);
    check(awry_count_batch(h_, bytes.data(), off.data(), out.size(), out.data()));
    return out;
  }
  // no counterpart in the reference: k-mers already packed 2 bits per letter (awry_count_packed_kmers)
  std::vector<uint64_t> parallel_count_packed(const std::vector<uint64_t>& words, int L) {
    std::vector<uint64_t> out(words.size());
    check(awry_count_packed_kmers(h_, words.data(), words.size(), L, out.data()));
    return out;
  }
  template <class StrRange>
  std::vector<std::vector<LocalizedSequencePosition>> parallel_locate(const StrRange& queries) {
    std::vector<uint8_t> bytes; std::vector<uint64_t> off;
    pack(queries, bytes, off);
    const uint64_t n = off.size() - 1;
    uint64_t* hoff = nullptr; awry_pos_t* hits = nullptr;
    check(awry_locate_batch(h_, bytes.data(), off.data(), n, &hoff, &hits, nullptr));
    std::vector<std::vector<LocalizedSequencePosition>> out(n);
    for (uint64_t i = 0; i < n; i++)
      for (uint64_t j = hoff[i]; j < hoff[i + 1]; j++) out[i].push_back({hits[j].seq_idx, hits[j].local_pos});
    awry_free_buffer(hoff); awry_free_buffer(hits);
    return out;
  }
  // src/fm_index.rs:559-582, 585-593
  SearchRange update_range_with_symbol(SearchRange r, char symbol) {
    awry_range_t o;
    check(awry_update_range(h_, awry_range_t{r.start_ptr, r.end_ptr}, (uint8_t)symbol, &o));
    return {o.start_ptr, o.end_ptr};
  }
  uint64_t backstep(uint64_t search_pointer) {
    uint64_t o = 0;
    check(awry_backstep(h_, search_pointer, &o));
    return o;
  }
  awry_index_t* handle() { return h_; }

 private:
  explicit FmIndex(awry_index_t* h) : h_(h) {}
  template <class StrRange>
  static void pack(const StrRange& qs, std::vector<uint8_t>& bytes, std::vector<uint64_t>& off) {
    off.assign(1, 0);
    for (const auto& q : qs) {
      std::string_view v(q);
      bytes.insert(bytes.end(), v.begin(), v.end());
      off.push_back(bytes.size());
    }
    if (bytes.empty()) bytes.push_back(0);
  }
  awry_index_t* h_ = nullptr;
};

}  // namespace awry
