"""include/awry.hpp on a GPU: a C++ caller written like a caller of the reference crate (FmIndex::new -> count_string /
locate_string / parallel_count / parallel_locate / save) gets the oracle's answers through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

import awry_amd
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#include "awry.hpp"
int main(int argc, char** argv) {
  awry::FmBuildArgs a;
  a.input_file_src = argv[1];
  a.suffix_array_compression_ratio = 8;
  a.lookup_table_kmer_len = 5;
  try {
    awry::FmIndex ix = awry::FmIndex::create(a, {0});
    std::vector<std::string> qs;
    { std::ifstream f(argv[2]); std::string s; while (std::getline(f, s)) qs.push_back(s); }
    std::vector<uint64_t> counts = ix.parallel_count(qs);
    auto hits = ix.parallel_locate(qs);
    std::FILE* out = std::fopen(argv[3], "w");
    for (size_t i = 0; i < qs.size(); i++) {
      if (counts[i] != ix.count_string(qs[i]) || counts[i] != hits[i].size()) return 3;
      auto single = ix.locate_string(qs[i]);
      if (!(single.size() == hits[i].size())) return 4;
      std::fprintf(out, "%llu", (unsigned long long)counts[i]);
      for (size_t j = 0; j < hits[i].size(); j++) {
        if (!(single[j] == hits[i][j])) return 5;
        std::fprintf(out, " %llu:%llu", (unsigned long long)hits[i][j].sequence_idx(), (unsigned long long)hits[i][j].local_position());
      }
      std::fprintf(out, "\n");
    }
    std::fclose(out);
    ix.save(argv[4]);
    try { ix.count_string(""); return 6; } catch (const awry::Error& e) { if (e.code != AWRY_ERR_INVALID_QUERY) return 7; }
  } catch (const awry::Error& e) {
    std::printf("unexpected: %d %s\n", e.code, e.what());
    return 8;
  }
  std::puts("cpp-gpu-ok");
  return 0;
}
'''


@pytest.mark.gpu
def test_cpp_caller_gets_the_oracles_answers(oracle, tmp_path):
    text, st, hd = synth.make_text(60000, 0, 31, 3, 0.02)
    fa = str(tmp_path / "t.fa")
    synth.write_fasta(fa, text, st, hd, 60)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 5, st, hd)
    rng = np.random.default_rng(2)
    qs = [bytes(q) for q in synth.sampled_queries(text, 150, 18, 1)] + [bytes(q) for q in synth.random_queries(50, 9, 0, 2)]
    qs += [b"ACG", b"N", b"acgtn", bytes(text[st[1] - 3:st[1] + 3])]
    (tmp_path / "q.txt").write_bytes(b"\n".join(qs) + b"\n")
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    libdir = os.path.dirname(awry_amd.lib_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lawry_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe), fa, str(tmp_path / "q.txt"), str(tmp_path / "out.txt"), str(tmp_path / "x.awry")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cpp-gpu-ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])
    lines = (tmp_path / "out.txt").read_text().splitlines()
    assert len(lines) == len(qs)
    for q, line in zip(qs, lines):
        tok = line.split()
        _, pos = oi.locate_string(q)
        assert int(tok[0]) == oi.count_string(q) == len(pos)
        assert [tuple(int(x) for x in t.split(":")) for t in tok[1:]] == pos, q
    # the file the C++ caller saved is the oracle's, byte for byte
    oi.save(str(tmp_path / "o.awry"))
    assert (tmp_path / "x.awry").read_bytes() == (tmp_path / "o.awry").read_bytes()
