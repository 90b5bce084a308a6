// host_index.cpp -- host-side index construction, sequence-file reader and .awry v1 (de)serialisation.
#include "host_index.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "alphabet.h"
#include "sais.hpp"

namespace awry {

// src/compressed_suffix_array.rs:124-130
uint64_t csa_bits_per_element(uint64_t bwt_len) {
  uint64_t largest = bwt_len - 1;
  return largest == 0 ? 0 : 64 - (uint64_t)__builtin_clzll(largest);
}
// src/compressed_suffix_array.rs:113-123
uint64_t csa_word_len(uint64_t bwt_len, uint64_t ratio) {
  unsigned __int128 bits = (unsigned __int128)((bwt_len + ratio - 1) / ratio) * csa_bits_per_element(bwt_len);
  return (uint64_t)((bits + 63) / 64);
}
// src/kmer_lookup_table.rs:113-118
uint64_t ref_kmer_table_entries(int alphabet, unsigned kmer_len) {
  uint64_t sigma = (uint64_t)cardinality(alphabet) - 2, n = 1;
  for (unsigned i = 0; i < kmer_len; i++) {
    if (n > (1ull << 40) / sigma) throw std::runtime_error("k-mer table too large");
    n *= sigma;
  }
  return n;
}

// ------------------------------------------------------------------ sequence files

static std::string first_token(const char* p, size_t n) {
  size_t k = 0;
  while (k < n && !isspace((unsigned char)p[k])) k++;
  return std::string(p, k);
}

namespace {

// A sequence file mapped into memory and cut into chunks that start at record boundaries, so that threads can
// parse them independently.  FASTA: multi-line records, a line starting with '>' is a header.  FASTQ: strict
// 4-line records.  Sequence bytes are kept as written, minus whitespace.
struct MappedFile {
  const char* p = nullptr;
  size_t n = 0;
  int fd = -1;
  bool mapped = false;
  std::vector<char> owned;
  explicit MappedFile(const std::string& path) {
    fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open sequence file: " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("cannot stat sequence file: " + path); }
    n = (size_t)st.st_size;
    if (n == 0) return;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) {
      p = static_cast<const char*>(m);
      mapped = true;
      (void)madvise(m, n, MADV_SEQUENTIAL);
    } else {  // not mappable (pipe, odd file system): read it
      owned.resize(n);
      size_t got = 0;
      while (got < n) {
        ssize_t r = read(fd, owned.data() + got, n - got);
        if (r <= 0) break;
        got += (size_t)r;
      }
      n = got;
      p = owned.data();
    }
  }
  ~MappedFile() {
    if (mapped) munmap(const_cast<char*>(p), n);
    if (fd >= 0) close(fd);
  }
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
};

inline const char* line_end(const char* q, const char* end) {
  const char* e = static_cast<const char*>(memchr(q, '\n', (size_t)(end - q)));
  return e ? e : end;
}

struct SeqScanner {
  const char* p;
  const char* end;
  bool fastq = false;
  std::vector<const char*> cuts;  // chunk t = [cuts[t], cuts[t + 1])

  SeqScanner(const MappedFile& f, unsigned threads) : p(f.p), end(f.p + f.n) {
    const char* q = p;
    while (q < end && (*q == '\n' || *q == '\r')) q++;  // format = first character of the first non-empty line
    fastq = q < end && *q == '@';
    cuts.push_back(p);
    const size_t n = f.n;
    for (unsigned t = 1; t < threads; t++) {
      const char* c = record_start_after(p + n / threads * t);
      if (c > cuts.back() && c < end) cuts.push_back(c);
    }
    cuts.push_back(end);
  }

  // first record start at or after q (end if none)
  const char* record_start_after(const char* q) const {
    if (q <= p) return p;
    q = line_end(q - 1, end);  // q - 1: a cut that already sits on a line start stays there
    while (q < end) {
      const char* ls = q + 1;  // start of the next line
      if (ls >= end) return end;
      if (!fastq) {
        if (*ls == '>') return ls;
      } else if (*ls == '@') {  // a header, unless it is a quality line: then line + 2 is a sequence, not a '+' line
        const char* l1 = line_end(ls, end);                         // end of the candidate header line
        const char* l2 = l1 < end ? line_end(l1 + 1, end) : end;    // end of the line after it
        if (l2 < end && l2 + 1 < end && l2[1] == '+') return ls;
      }
      q = line_end(ls, end);
    }
    return end;
  }

  // fn(header, header_len, emit) per record of chunk t, in file order; emit(ptr, len) delivers sequence bytes piecewise
  template <class Rec>
  void for_each_in_chunk(size_t t, Rec&& rec) const {
    const char* q = cuts[t];
    const char* e = cuts[t + 1];
    if (fastq) {
      int state = 0;  // 0 header, 1 sequence, 2 '+', 3 quality
      const char* hdr = nullptr;
      size_t hlen = 0;
      while (q < e) {
        const char* le = line_end(q, e);
        const char* te = le;
        while (te > q && (te[-1] == '\r' || te[-1] == '\n')) te--;
        if (state == 0) {
          if (te > q) { hdr = q + 1; hlen = (size_t)(te - q - 1); state = 1; }
        } else if (state == 1) {
          rec.record(hdr, hlen, q, (size_t)(te - q), true);
          state = 2;
        } else if (state == 2) {
          state = 3;
        } else {
          state = 0;
        }
        q = le < e ? le + 1 : e;
      }
      if (state == 1) rec.record(hdr, hlen, q, 0, true);  // header without a sequence line
    } else {
      bool open_rec = false;
      while (q < e) {
        const char* le = line_end(q, e);
        const char* te = le;
        while (te > q && (te[-1] == '\r' || te[-1] == '\n')) te--;
        if (te > q && *q == '>') {
          rec.record(q + 1, (size_t)(te - q - 1), nullptr, 0, false);
          open_rec = true;
        } else if (te > q) {
          if (!open_rec) { rec.record(q, 0, nullptr, 0, false); open_rec = true; }  // sequence data before any header
          rec.more(q, (size_t)(te - q));
        }
        q = le < e ? le + 1 : e;
      }
    }
  }
};

// C-locale isspace: ' ', \t \n \v \f \r
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
// lines are trimmed at the right already; inner whitespace is rare, so look for it with a branch-free scan first
inline bool has_space(const char* q, size_t n) {
  unsigned any = 0;
  for (size_t i = 0; i < n; i++) any |= (unsigned)is_space((unsigned char)q[i]);
  return any != 0;
}
inline size_t count_nonspace(const char* q, size_t n) {
  if (!has_space(q, n)) return n;
  size_t k = 0;
  for (size_t i = 0; i < n; i++) k += !is_space((unsigned char)q[i]);
  return k;
}
inline uint8_t* copy_nonspace(uint8_t* dst, const char* q, size_t n, bool upper) {
  if (n == 0) return dst;
  if (!has_space(q, n)) {
    if (!upper) { memcpy(dst, q, n); return dst + n; }
    for (size_t i = 0; i < n; i++) {  // ASCII upper-casing, vectorisable
      const unsigned char c = (unsigned char)q[i];
      dst[i] = (uint8_t)(c - ((c >= 'a' && c <= 'z') ? 32 : 0));
    }
    return dst + n;
  }
  for (size_t i = 0; i < n; i++) {
    const unsigned char c = (unsigned char)q[i];
    if (!is_space(c)) *dst++ = upper ? (uint8_t)toupper(c) : c;
  }
  return dst;
}

struct ChunkTally {  // pass 1
  uint64_t records = 0, bytes = 0;
  void record(const char*, size_t, const char* seq, size_t n, bool) { records++; bytes += count_nonspace(seq, n); }
  void more(const char* seq, size_t n) { bytes += count_nonspace(seq, n); }
};

unsigned reader_threads(size_t bytes) {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  return (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(hw, 16u), bytes >> 22));  // >= 4 MiB per thread
}

template <class F>
void run_chunks(size_t nchunks, F&& fn) {
  if (nchunks == 1) { fn(0); return; }
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> errs(nchunks);
  for (size_t t = 0; t < nchunks; t++)
    pool.emplace_back([&, t] { try { fn(t); } catch (...) { errs[t] = std::current_exception(); } });
  for (auto& th : pool) th.join();
  for (auto& e : errs) if (e) std::rethrow_exception(e);
}

}  // namespace

// Text model of src/fm_index.rs:148-153,182,220-223: records joined by one delimiter byte ('N' / 'X'),
// one trailing '$'.  The reader itself (libsufr::util::read_sequence_file) is not in the reference tree;
// letters are upper-cased (ignore_softmask: true, src/fm_index.rs:161) and the header is the first
// whitespace-delimited token.  Parity is pinned only for upper-case canonical letters.
// Two parallel passes over the mapped file: tally records / bytes per chunk, then fill.
SequenceFile read_sequence_file(const std::string& path, int alphabet) {
  MappedFile f(path);
  SeqScanner sc(f, reader_threads(f.n));
  const size_t nc = sc.cuts.size() - 1;
  std::vector<ChunkTally> tally(nc);
  run_chunks(nc, [&](size_t t) { sc.for_each_in_chunk(t, tally[t]); });
  std::vector<uint64_t> rec0(nc + 1, 0), byte0(nc + 1, 0);
  for (size_t t = 0; t < nc; t++) { rec0[t + 1] = rec0[t] + tally[t].records; byte0[t + 1] = byte0[t] + tally[t].bytes; }
  const uint64_t nrec = rec0[nc];
  if (nrec == 0) throw std::runtime_error("no sequence records in " + path);
  SequenceFile sf;
  sf.text.resize(byte0[nc] + (nrec - 1) + 1);  // + delimiters + '$'
  sf.starts.resize(nrec);
  sf.headers.resize(nrec);
  const uint8_t delim = alphabet == NUCLEOTIDE ? 'N' : 'X';
  struct Fill {
    SequenceFile& sf;
    uint64_t rec, at;  // next record index, next text position
    uint8_t delim;
    void record(const char* h, size_t hn, const char* seq, size_t n, bool) {
      if (rec) sf.text[at++] = delim;
      sf.starts[rec] = at;
      sf.headers[rec] = first_token(h, hn);
      rec++;
      at = (uint64_t)(copy_nonspace(sf.text.data() + at, seq, n, true) - sf.text.data());
    }
    void more(const char* seq, size_t n) { at = (uint64_t)(copy_nonspace(sf.text.data() + at, seq, n, true) - sf.text.data()); }
  };
  run_chunks(nc, [&](size_t t) {
    // record r of the file starts at text position bytes_before + r (one delimiter per earlier record)
    Fill fill{sf, rec0[t], byte0[t] + (rec0[t] ? rec0[t] - 1 : 0), delim};
    sc.for_each_in_chunk(t, fill);
  });
  sf.text.back() = '$';
  return sf;
}

// query ingestion: every record of a FASTA/FASTQ file becomes one query of a CSR batch.
// The arrays are malloc'ed (the C ABI hands them to the caller, awry_free_buffer = free).
void read_query_file(const std::string& path, uint8_t** bytes_out, uint64_t** offsets_out, uint64_t* n_out) {
  MappedFile f(path);
  SeqScanner sc(f, reader_threads(f.n));
  const size_t nc = sc.cuts.size() - 1;
  std::vector<ChunkTally> tally(nc);
  run_chunks(nc, [&](size_t t) { sc.for_each_in_chunk(t, tally[t]); });
  std::vector<uint64_t> rec0(nc + 1, 0), byte0(nc + 1, 0);
  for (size_t t = 0; t < nc; t++) { rec0[t + 1] = rec0[t] + tally[t].records; byte0[t + 1] = byte0[t] + tally[t].bytes; }
  const uint64_t nrec = rec0[nc];
  uint8_t* bytes = static_cast<uint8_t*>(malloc(std::max<uint64_t>(1, byte0[nc])));
  uint64_t* offsets = static_cast<uint64_t*>(malloc((nrec + 1) * 8));
  if (!bytes || !offsets) { free(bytes); free(offsets); throw std::bad_alloc(); }
  struct Fill {
    uint8_t* bytes;
    uint64_t* offsets;
    uint64_t rec, at;
    void record(const char*, size_t, const char* seq, size_t n, bool) {
      offsets[rec++] = at;
      at = (uint64_t)(copy_nonspace(bytes + at, seq, n, false) - bytes);
    }
    void more(const char* seq, size_t n) { at = (uint64_t)(copy_nonspace(bytes + at, seq, n, false) - bytes); }
  };
  try {
    run_chunks(nc, [&](size_t t) {
      Fill fill{bytes, offsets, rec0[t], byte0[t]};
      sc.for_each_in_chunk(t, fill);
    });
  } catch (...) { free(bytes); free(offsets); throw; }
  offsets[nrec] = byte0[nc];
  *bytes_out = bytes;
  *offsets_out = offsets;
  *n_out = nrec;
}

// ------------------------------------------------------------------ packing

namespace {

inline void set_symbol(uint64_t* blk, int alphabet, uint32_t code, unsigned pos) {
  const int l = (int)(pos >> 6);
  const uint64_t bit = 1ull << (pos & 63);
  for (int b = 0; code; b++, code >>= 1)
    if (code & 1) blk[plane_word(alphabet, b, l)] |= bit;
}

inline void set_milestones(uint64_t* blk, int alphabet, const uint64_t* counts) {
  if (alphabet == NUCLEOTIDE) {
    for (int l = 0; l < 4; l++) blk[nt_ms_word(l)] = counts[nt_index_of_letter(l)];
  } else {
    for (int t = 0; t < 21; t++) {
      uint64_t v = counts[t + 1] & 0xffffffffull;
      blk[aa_ms_word(t)] |= v << (32 * aa_ms_half(t));
    }
  }
}

unsigned worker_count(uint64_t items, uint64_t min_per_thread) {
  unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  uint64_t want = std::max<uint64_t>(1, items / std::max<uint64_t>(1, min_per_thread));
  return (unsigned)std::min<uint64_t>(std::min(hw, 32u), want);
}

template <class F>
void parallel_ranges(uint64_t n, uint64_t align, unsigned threads, F&& fn) {
  if (threads <= 1 || n == 0) { fn(0, 0, n); return; }
  uint64_t per = ((n + threads - 1) / threads + align - 1) / align * align;
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < threads; t++) {
    uint64_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
    pool.emplace_back([=, &fn] { fn(t, lo, hi); });
  }
  for (auto& th : pool) th.join();
}

}  // namespace

void pack_index(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, const uint64_t* sa64, const uint32_t* sa32) {
  const int A = ix.alphabet, card = cardinality(A), BW = block_words(A);
  if (A == AMINO && bwt_len >= (1ull << 32))
    throw std::runtime_error("amino indexes with bwt_len >= 2^32 are not supported (u32 milestones)");
  ix.bwt_len = bwt_len;
  ix.nblocks = (bwt_len + 255) / 256;  // src/bwt.rs:302-304
  ix.sa_bits = csa_bits_per_element(bwt_len);
  ix.blocks.assign(ix.nblocks * BW, 0);
  ix.sa_words.assign(csa_word_len(bwt_len, ix.sa_ratio), 0);
  auto SA = [&](uint64_t i) -> uint64_t { return sa32 ? (uint64_t)sa32[i] : sa64[i]; };
  auto prev_char = [&](uint64_t v) -> uint8_t { return v == 0 ? (uint8_t)'$' : text[v - 1]; };

  // pass 1: letter histogram per chunk -> exclusive counts at each chunk start
  const unsigned T = worker_count(bwt_len, 1u << 20);
  std::vector<std::vector<uint64_t>> hist(T, std::vector<uint64_t>(24, 0));
  std::vector<uint64_t> lo_of(T, 0), hi_of(T, 0);
  std::vector<uint64_t> sentinel(T, UINT64_MAX);
  parallel_ranges(bwt_len, 256, T, [&](unsigned t, uint64_t lo, uint64_t hi) {
    lo_of[t] = lo; hi_of[t] = hi;
    auto& h = hist[t];
    for (uint64_t i = lo; i < hi; i++) {
      uint64_t v = SA(i);
      if (v == 0) sentinel[t] = i;
      h[index_of_ascii(A, prev_char(v))]++;
    }
  });
  std::vector<std::vector<uint64_t>> base(T, std::vector<uint64_t>(24, 0));
  std::vector<uint64_t> total(24, 0);
  for (unsigned t = 0; t < T; t++) {
    base[t] = total;
    for (int c = 0; c < 24; c++) total[c] += hist[t][c];
    if (sentinel[t] != UINT64_MAX) ix.sentinel_row = sentinel[t];
  }
  // pass 2: planes + milestones (src/fm_index.rs:203-230)
  parallel_ranges(bwt_len, 256, T, [&](unsigned t, uint64_t lo, uint64_t hi) {
    std::vector<uint64_t> counts = base[t];
    for (uint64_t i = lo; i < hi; i++) {
      uint64_t* blk = ix.blocks.data() + (i >> 8) * BW;
      if ((i & 255) == 0) set_milestones(blk, A, counts.data());
      int idx = index_of_ascii(A, prev_char(SA(i)));
      set_symbol(blk, A, code_of_index(A, idx), (unsigned)(i & 255));
      counts[idx]++;
    }
  });
  // prefix sums (src/fm_index.rs:233-240)
  ix.prefix_sums.assign(card + 1, 0);
  uint64_t acc = 0;
  for (int i = 0; i <= card; i++) {
    ix.prefix_sums[i] = acc;
    if (i != card) acc += total[i];
  }
  // sampled SA, bit-packed LSB-first (src/compressed_suffix_array.rs:51-64); each worker owns a word range
  const uint64_t bits = ix.sa_bits, nsamp = (bwt_len + ix.sa_ratio - 1) / ix.sa_ratio, nw = ix.sa_words.size();
  if (bits)
    parallel_ranges(nw, 1, worker_count(nw, 1u << 18), [&](unsigned, uint64_t w0, uint64_t w1) {
      if (w0 >= w1) return;
      uint64_t j0 = (uint64_t)(((unsigned __int128)w0 * 64) / bits);
      uint64_t j1 = std::min<uint64_t>(nsamp, (uint64_t)(((unsigned __int128)w1 * 64 + bits - 1) / bits));
      for (uint64_t j = j0; j < j1; j++) {
        uint64_t v = SA(j * ix.sa_ratio);
        unsigned __int128 off = (unsigned __int128)j * bits;
        uint64_t w = (uint64_t)(off / 64), b = (uint64_t)(off % 64);
        if (w >= w0 && w < w1) ix.sa_words[w] |= v << b;
        if (b + bits > 64 && w + 1 >= w0 && w + 1 < w1) ix.sa_words[w + 1] |= v >> (64 - b);
      }
    });
}

void prepare_build(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio,
                   unsigned kmer_len, const uint64_t* seq_starts, const char* const* headers, uint64_t nseq) {
  if (bwt_len == 0 || text[bwt_len - 1] != '$') throw std::runtime_error("text must end with '$'");
  if (alphabet != NUCLEOTIDE && alphabet != AMINO) throw std::runtime_error("bad alphabet id");
  ix.alphabet = alphabet;
  ix.sa_ratio = sa_ratio ? sa_ratio : 8;                                     // src/fm_index.rs:122
  ix.kmer_len = (uint8_t)(kmer_len ? kmer_len : (alphabet == NUCLEOTIDE ? 10 : 4));  // src/kmer_lookup_table.rs:23-24
  ref_kmer_table_entries(alphabet, ix.kmer_len);                            // range check
  ix.seq_starts.assign(seq_starts, seq_starts + nseq);
  ix.headers.clear();
  for (uint64_t i = 0; i < nseq; i++) ix.headers.emplace_back(headers && headers[i] ? headers[i] : "");
  ix.ref_kmer_table.clear();
}

void build_from_text(HostIndex& ix, const uint8_t* text, uint64_t bwt_len, int alphabet, uint64_t sa_ratio,
                     unsigned kmer_len, const uint64_t* seq_starts, const char* const* headers, uint64_t nseq) {
  if (bwt_len == 0 || text[bwt_len - 1] != '$') throw std::runtime_error("text must end with '$'");
  if (alphabet != NUCLEOTIDE && alphabet != AMINO) throw std::runtime_error("bad alphabet id");
  ix.alphabet = alphabet;
  ix.sa_ratio = sa_ratio ? sa_ratio : 8;                                     // src/fm_index.rs:122
  ix.kmer_len = (uint8_t)(kmer_len ? kmer_len : (alphabet == NUCLEOTIDE ? 10 : 4));  // src/kmer_lookup_table.rs:23-24
  ref_kmer_table_entries(alphabet, ix.kmer_len);                            // range check
  ix.seq_starts.assign(seq_starts, seq_starts + nseq);
  ix.headers.clear();
  for (uint64_t i = 0; i < nseq; i++) ix.headers.emplace_back(headers && headers[i] ? headers[i] : "");
  ix.ref_kmer_table.clear();
  if (bwt_len < (1ull << 31)) {
    std::vector<int32_t> sa((size_t)bwt_len);
    sais_rec<uint8_t, int32_t>(text, sa.data(), (int32_t)bwt_len, 255);
    pack_index(ix, text, bwt_len, nullptr, reinterpret_cast<const uint32_t*>(sa.data()));
  } else {
    std::vector<uint64_t> sa((size_t)bwt_len);
    suffix_array_bytes(text, bwt_len, sa.data());
    pack_index(ix, text, bwt_len, sa.data(), nullptr);
  }
}

// ------------------------------------------------------------------ reference layout <-> device layout

void block_to_reference(const HostIndex& ix, uint64_t b, uint64_t* out) {
  const int A = ix.alphabet, P = num_planes(A), BW = block_words(A);
  const uint64_t* blk = ix.blocks.data() + b * BW;
  for (int p = 0; p < P; p++)
    for (int l = 0; l < 4; l++) out[4 * p + l] = blk[plane_word(A, p, l)];
  uint64_t* ms = out + 4 * P;
  const int nms = A == NUCLEOTIDE ? 8 : 24;  // src/bwt.rs:29,139
  std::fill(ms, ms + nms, 0);
  const uint64_t dollar = ix.sentinel_row < 256 * b ? 1 : 0;  // '$' occurs exactly once
  ms[0] = dollar;
  if (A == NUCLEOTIDE) {
    uint64_t sum = 0;
    for (int l = 0; l < 4; l++) { ms[nt_index_of_letter(l)] = blk[nt_ms_word(l)]; sum += blk[nt_ms_word(l)]; }
    ms[4] = 256 * b - sum - dollar;  // N
  } else {
    for (int t = 0; t < 21; t++) ms[t + 1] = (blk[aa_ms_word(t)] >> (32 * aa_ms_half(t))) & 0xffffffffull;
  }
}

void block_from_reference(HostIndex& ix, uint64_t b, const uint64_t* in) {
  const int A = ix.alphabet, P = num_planes(A), BW = block_words(A);
  uint64_t* blk = ix.blocks.data() + b * BW;
  std::fill(blk, blk + BW, 0);
  for (int p = 0; p < P; p++)
    for (int l = 0; l < 4; l++) blk[plane_word(A, p, l)] = in[4 * p + l];
  set_milestones(blk, A, in + 4 * P);
}

// ------------------------------------------------------------------ the reference's k-mer table, on the host

namespace {
// Occ(idx, row) inclusive of row on the host copy (device layout): src/bwt.rs:338-357.  Used only to write the k-mer
// table of an .awry file when no GPU replica exists -- never to answer a query.
uint64_t host_rank(const HostIndex& ix, uint64_t row, int idx) {
  const int A = ix.alphabet, BW = block_words(A), P = num_planes(A);
  const uint64_t b = row >> 8;
  const int p = (int)(row & 255);
  const uint64_t* blk = ix.blocks.data() + b * BW;
  const uint32_t code = code_of_index(A, idx);
  uint64_t cnt = 0;
  for (int l = 0; l < 4; l++) {
    const int t = p - 64 * l;
    if (t < 0) break;
    uint64_t pred = ~0ull;
    for (int pl = 0; pl < P; pl++) pred &= blk[plane_word(A, pl, l)] ^ (((code >> pl) & 1u) ? 0ull : ~0ull);
    cnt += (uint64_t)__builtin_popcountll(pred & (t >= 63 ? ~0ull : ((1ull << (t + 1)) - 1)));
  }
  uint64_t ms;
  if (A == NUCLEOTIDE) {
    const int letter = nt_letter_of_index(idx);
    if (letter >= 0) ms = blk[nt_ms_word(letter)];
    else {  // N: derived (layout.h)
      uint64_t sum = 0;
      for (int l = 0; l < 4; l++) sum += blk[nt_ms_word(l)];
      ms = 256 * b - sum - (ix.sentinel_row < 256 * b ? 1 : 0);
    }
  } else {
    const int t = idx - 1;
    ms = (blk[aa_ms_word(t)] >> (32 * aa_ms_half(t))) & 0xffffffffull;
  }
  return ms + cnt;
}
}  // namespace

// src/kmer_lookup_table.rs:121-167, as ref_kmer_table_kernel computes it on a GPU: slot = sum_j s_j sigma^j with s_0 the
// LAST symbol, digits 1..sigma-1 only, k-1 steps without an emptiness check, every other slot {1, 0}
void fill_ref_kmer_table_host(HostIndex& ix) {
  const uint64_t nslots = ref_kmer_table_entries(ix.alphabet, ix.kmer_len);
  if (ix.ref_kmer_table.size() == 2 * nslots) return;
  const uint64_t sigma = (uint64_t)cardinality(ix.alphabet) - 2;
  const int k = ix.kmer_len;
  std::vector<uint64_t> tab(2 * nslots);
  parallel_ranges(nslots, 1, worker_count(nslots, 1u << 14), [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t slot = lo; slot < hi; slot++) {
      uint64_t sp = 1, ep = 0, rem = slot;
      bool populated = k > 0;
      for (int j = 0; j < k; j++) { if (rem % sigma == 0) populated = false; rem /= sigma; }
      if (populated) {
        rem = slot;
        int idx = (int)(rem % sigma);
        rem /= sigma;
        sp = ix.prefix_sums[idx];
        ep = ix.prefix_sums[idx + 1] - 1;
        for (int j = 1; j < k; j++) {  // update_range_with_symbol, src/fm_index.rs:559-582
          idx = (int)(rem % sigma);
          rem /= sigma;
          const uint64_t c = ix.prefix_sums[idx];
          const uint64_t s2 = c + host_rank(ix, sp - 1, idx);
          ep = c + host_rank(ix, ep, idx) - 1;
          sp = s2;
        }
      }
      tab[2 * slot] = sp;
      tab[2 * slot + 1] = ep;
    }
  });
  ix.ref_kmer_table = std::move(tab);
}

// ------------------------------------------------------------------ .awry v1

static const char MAGIC[12] = "AWRY-Index\n";  // 11 bytes on disk, src/fm_index_file.rs:18

namespace {
struct Writer {
  FILE* f;
  void put(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("short write"); }
  void u64(uint64_t v) { put(&v, 8); }
};
struct Reader {
  FILE* f;
  void get(void* p, size_t n) { if (n && fread(p, 1, n, f) != n) throw std::runtime_error("truncated .awry file"); }
  uint64_t u64() { uint64_t v; get(&v, 8); return v; }
};
struct FileCloser { FILE* f; ~FileCloser() { if (f) fclose(f); } };
}  // namespace

// src/fm_index_file.rs:42-106, src/sequence_index.rs:144-152
void save_awry(const HostIndex& ix, const std::string& path) {
  const uint64_t nk = ref_kmer_table_entries(ix.alphabet, ix.kmer_len);
  if (ix.ref_kmer_table.size() != 2 * nk) throw std::runtime_error("k-mer table not materialised before save");
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot open for writing: " + path);
  FileCloser fc{f};
  Writer w{f};
  w.put(MAGIC, 11);
  w.u64(ix.version); w.u64(ix.sa_ratio); w.u64(ix.bwt_len); w.u64((uint64_t)ix.alphabet);  // :165-181
  const int RW = 4 * num_planes(ix.alphabet) + (ix.alphabet == NUCLEOTIDE ? 8 : 24);
  std::vector<uint64_t> buf((size_t)RW * 4096);
  for (uint64_t b0 = 0; b0 < ix.nblocks; b0 += 4096) {
    uint64_t nb = std::min<uint64_t>(4096, ix.nblocks - b0);
    for (uint64_t j = 0; j < nb; j++) block_to_reference(ix, b0 + j, buf.data() + j * RW);
    w.put(buf.data(), nb * RW * 8);
  }
  w.put(ix.prefix_sums.data(), ix.prefix_sums.size() * 8);
  w.put(ix.sa_words.data(), ix.sa_words.size() * 8);
  w.put(&ix.kmer_len, 1);
  w.put(ix.ref_kmer_table.data(), ix.ref_kmer_table.size() * 8);
  w.u64(ix.seq_starts.size());
  for (size_t i = 0; i < ix.seq_starts.size(); i++) {
    w.u64(ix.seq_starts[i]);
    w.u64(ix.headers[i].size());
    w.put(ix.headers[i].data(), ix.headers[i].size());
  }
  if (fflush(f) != 0) throw std::runtime_error("write failed: " + path);
}

// src/fm_index_file.rs:132-287, src/kmer_lookup_table.rs:55-77, src/sequence_index.rs:154-183
void load_awry(HostIndex& ix, const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open: " + path);
  FileCloser fc{f};
  Reader r{f};
  char magic[11];
  r.get(magic, 11);
  if (memcmp(magic, MAGIC, 11) != 0) throw std::invalid_argument("not an AWRY index file (bad label)");
  ix.version = r.u64();
  ix.sa_ratio = r.u64();
  ix.bwt_len = r.u64();
  uint64_t alpha = r.u64();
  if (alpha > 1) throw std::invalid_argument("invalid alphabet id in file");
  if (ix.sa_ratio == 0 || ix.bwt_len == 0) throw std::invalid_argument("corrupt header");
  ix.alphabet = (int)alpha;
  if (ix.alphabet == AMINO && ix.bwt_len >= (1ull << 32)) throw std::invalid_argument("amino bwt_len >= 2^32 unsupported");
  const int A = ix.alphabet, P = num_planes(A), BW = block_words(A), card = cardinality(A);
  const int RW = 4 * P + (A == NUCLEOTIDE ? 8 : 24);
  ix.nblocks = (ix.bwt_len + 255) / 256;
  ix.sa_bits = csa_bits_per_element(ix.bwt_len);
  ix.blocks.assign(ix.nblocks * BW, 0);
  ix.sentinel_row = UINT64_MAX;
  std::vector<uint64_t> buf((size_t)RW * 4096);
  for (uint64_t b0 = 0; b0 < ix.nblocks; b0 += 4096) {
    uint64_t nb = std::min<uint64_t>(4096, ix.nblocks - b0);
    r.get(buf.data(), nb * RW * 8);
    for (uint64_t j = 0; j < nb; j++) {
      const uint64_t* in = buf.data() + j * RW;
      block_from_reference(ix, b0 + j, in);
      // locate the sentinel row: nt code 0b100; amino code 0b00000 on a row < bwt_len
      for (int l = 0; l < 4; l++) {
        uint64_t m;
        if (A == NUCLEOTIDE) m = in[8 + l] & ~in[4 + l] & ~in[l];
        else {
          m = ~(in[l] | in[4 + l] | in[8 + l] | in[12 + l] | in[16 + l]);
          uint64_t row0 = (b0 + j) * 256 + 64 * l;
          if (row0 >= ix.bwt_len) m = 0;
          else if (ix.bwt_len - row0 < 64) m &= (1ull << (ix.bwt_len - row0)) - 1;
        }
        if (m) ix.sentinel_row = (b0 + j) * 256 + 64 * l + (uint64_t)__builtin_ctzll(m);
      }
    }
  }
  if (ix.sentinel_row == UINT64_MAX) throw std::invalid_argument("no sentinel row in BWT");
  ix.prefix_sums.resize(card + 1);
  r.get(ix.prefix_sums.data(), (card + 1) * 8);
  ix.sa_words.resize(csa_word_len(ix.bwt_len, ix.sa_ratio));
  r.get(ix.sa_words.data(), ix.sa_words.size() * 8);
  r.get(&ix.kmer_len, 1);
  ix.ref_kmer_table.resize(2 * ref_kmer_table_entries(A, ix.kmer_len));
  r.get(ix.ref_kmer_table.data(), ix.ref_kmer_table.size() * 8);
  uint64_t ns = r.u64();
  if (ns > (1ull << 40)) throw std::invalid_argument("corrupt sequence index");
  ix.seq_starts.resize(ns);
  ix.headers.resize(ns);
  for (uint64_t i = 0; i < ns; i++) {
    ix.seq_starts[i] = r.u64();
    uint64_t hl = r.u64();
    if (hl > (1ull << 32)) throw std::invalid_argument("corrupt header length");
    ix.headers[i].resize(hl);
    r.get(ix.headers[i].data(), hl);
  }
}

}  // namespace awry
