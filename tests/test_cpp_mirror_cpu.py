"""include/awry.hpp (the C++ host-side mirror of FmIndex) compiles against include/awry_hip.h, links
libawry_hip.so and reports the missing GPU through its exception type.  CPU only."""
import os
import subprocess
import sys

import awry_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <fstream>
#include "awry.hpp"
int main(int argc, char** argv) {
  { std::ofstream f(argv[1]); f << ">r0\nACGTACGTTTGACCA\nGGATTACA\n>r1 desc\nTTTTACGT\n"; }
  awry::FmBuildArgs a;
  a.input_file_src = argv[1];
  a.alphabet = awry::SymbolAlphabet::Nucleotide;
  try {
    awry::FmIndex ix = awry::FmIndex::create(a, {});   // build only, no device replica
    if (ix.bwt_len() != 15 + 8 + 1 + 8 + 1) return 2;  // two records joined by 'N' plus '$'
    if (ix.version_number() != 1 || ix.suffix_array_compression_ratio() != 8) return 3;
    auto ps = ix.prefix_sums();
    if (ps.size() != 7 || ps[0] != 0 || ps[1] != 1 || ps[6] != ix.bwt_len()) return 4;
    awry::SearchRange r = ix.initial_search_range('A');
    if (r.start_ptr != ps[1] || r.end_ptr != ps[2] - 1) return 5;
    try {
      ix.count_string("ACGT");
      return 6;  // must not succeed without a GPU replica
    } catch (const awry::Error& e) {
      if (e.code != AWRY_ERR_NO_DEVICE) return 7;
    }
  } catch (const awry::Error& e) {
    std::printf("unexpected: %d %s\n", e.code, e.what());
    return 8;
  }
  std::puts("cpp-mirror-ok");
  return 0;
}
'''


def test_cpp_mirror_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    so = awry_amd.lib_path()
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    libdir = os.path.dirname(so)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lawry_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, AWRY_BUILD="host")
    r = subprocess.run([str(exe), str(tmp_path / "x.fa")], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "cpp-mirror-ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
