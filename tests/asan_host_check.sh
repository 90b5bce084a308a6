#!/bin/bash
# AddressSanitizer + UBSan over the host-side code paths that need no GPU (GPU ASan is not available on the pool):
# the oracle (index build, search, locate, save/load) and the product's host index code (FASTA reader, SA-IS,
# device-layout packing, .awry reader, layout round trips).  usage: tests/asan_host_check.sh
set -euo pipefail
cd "$(dirname "$0")/.."
T=$(mktemp -d)
cat > $T/main.cpp <<'CPP'
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "awry_amd/csrc/host_index.h"
#include "oracle/awry_oracle.h"
using namespace awry;
int main(int argc, char** argv) {
  std::mt19937_64 rng(7);
  for (int alphabet = 0; alphabet < 2; alphabet++) {
    const char* letters = alphabet == 0 ? "ACGTN" : "ACDEFGHIKLMNPQRSTVWYX";
    const size_t nl = strlen(letters);
    for (size_t n : {1ul, 2ul, 255ul, 256ul, 257ul, 5000ul, 70001ul}) {
      std::string fa = std::string(argv[1]) + "/t.fa";
      FILE* f = fopen(fa.c_str(), "w");
      size_t written = 0, rec = 0;
      while (written < n) {
        size_t len = std::min<size_t>(n - written, 1 + rng() % 3000);
        fprintf(f, ">rec%zu some description\n", rec++);
        for (size_t i = 0; i < len; i++) { fputc(letters[rng() % nl], f); if (i % 70 == 69) fputc('\n', f); }
        fputc('\n', f);
        written += len;
      }
      fclose(f);
      SequenceFile sf = read_sequence_file(fa, alphabet);
      HostIndex ix;
      std::vector<const char*> hdr;
      for (auto& h : sf.headers) hdr.push_back(h.c_str());
      build_from_text(ix, sf.text.data(), sf.text.size(), alphabet, 1 + rng() % 9, 3, sf.starts.data(), hdr.data(), sf.starts.size());
      orc_index* oi = orc_index_from_fasta(fa.c_str(), alphabet, ix.sa_ratio, 3);
      if (!oi || orc_bwt_len(oi) != ix.bwt_len) { printf("length mismatch\n"); return 1; }
      uint64_t nw = 0;
      const uint64_t* ob = orc_block_words(oi, &nw);
      const int RW = alphabet == 0 ? 20 : 44;
      std::vector<uint64_t> ref(RW);
      for (uint64_t b = 0; b < ix.nblocks; b++) {
        block_to_reference(ix, b, ref.data());
        if (memcmp(ref.data(), ob + b * RW, RW * 8)) { printf("block %llu differs\n", (unsigned long long)b); return 1; }
      }
      std::string path = std::string(argv[1]) + "/o.awry";
      if (orc_index_save(oi, path.c_str())) return 1;
      HostIndex ld;
      load_awry(ld, path);
      if (ld.blocks != ix.blocks || ld.sa_words != ix.sa_words || ld.prefix_sums != ix.prefix_sums || ld.sentinel_row != ix.sentinel_row) { printf("load differs\n"); return 1; }
      ld.ref_kmer_table = ld.ref_kmer_table;  // loaded table present: save works without a GPU
      save_awry(ld, std::string(argv[1]) + "/p.awry");
      orc_index* o2 = orc_index_load((std::string(argv[1]) + "/p.awry").c_str());
      if (!o2) return 1;
      for (int q = 0; q < 200; q++) {
        std::string s;
        for (size_t i = 0, L = 1 + rng() % 12; i < L; i++) s.push_back(letters[rng() % (nl - 1)]);
        uint64_t c1 = 0, c2 = 0, *g = nullptr, nh = 0; orc_pos* p = nullptr;
        orc_count_string(oi, (const uint8_t*)s.data(), s.size(), &c1);
        orc_locate_string(o2, (const uint8_t*)s.data(), s.size(), &g, &p, &nh, nullptr);
        c2 = nh; orc_free(g); orc_free(p);
        if (c1 != c2) { printf("count/locate mismatch\n"); return 1; }
      }
      uint8_t* qb = nullptr; uint64_t* qo = nullptr; uint64_t nq = 0;
      read_query_file(fa, &qb, &qo, &nq);
      if (nq != sf.starts.size() || qo[nq] + nq != sf.text.size()) { printf("query reader disagrees with the sequence reader\n"); return 1; }
      free(qb); free(qo);
      orc_index_free(oi); orc_index_free(o2);
    }
  }
  {  // files big enough for the chunked, multi-threaded readers (FASTQ and wrapped FASTA)
    std::string fq = std::string(argv[1]) + "/big.fq", fa = std::string(argv[1]) + "/big.fa";
    FILE* f = fopen(fq.c_str(), "w");
    FILE* g = fopen(fa.c_str(), "w");
    size_t total = 0, nrec = 0;
    while (total < (12u << 20)) {
      size_t len = rng() % 400;
      std::string s, ql;
      for (size_t i = 0; i < len; i++) { s.push_back("ACGTN"[rng() % 5]); ql.push_back((char)(33 + rng() % 60)); }
      if (len && nrec % 3 == 0) ql[0] = '@';
      fprintf(f, "@r%zu\n%s\n+\n%s\n", nrec, s.c_str(), ql.c_str());
      fprintf(g, ">r%zu\n", nrec);
      for (size_t i = 0; i < len; i += 61) fprintf(g, "%s\n", s.substr(i, 61).c_str());
      total += 2 * len + 16;
      nrec++;
    }
    fclose(f);
    fclose(g);
    uint8_t *b1 = nullptr, *b2 = nullptr; uint64_t *o1 = nullptr, *o2 = nullptr, n1 = 0, n2 = 0;
    read_query_file(fq, &b1, &o1, &n1);
    read_query_file(fa, &b2, &o2, &n2);
    if (n1 != nrec || n2 != nrec || o1[n1] != o2[n2] || memcmp(b1, b2, o1[n1]) || memcmp(o1, o2, (n1 + 1) * 8)) { printf("big reader mismatch\n"); return 1; }
    SequenceFile sf = read_sequence_file(fa, 0);
    if (sf.starts.size() != nrec || sf.text.size() != o1[n1] + nrec) { printf("big sequence reader mismatch\n"); return 1; }
    free(b1); free(b2); free(o1); free(o2);
  }
  puts("asan-host-check-ok");
  return 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I. -Iawry_amd/csrc $T/main.cpp awry_amd/csrc/host_index.cpp -x c oracle/awry_oracle.c -x none -lpthread -o $T/check 2>&1 | grep -v "warning" || true
# the GPU builder symbol is referenced by nothing in this build
ASAN_OPTIONS=detect_leaks=1 $T/check $T
rm -rf $T
