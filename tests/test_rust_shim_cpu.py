"""The Rust shim ships as source (no Rust toolchain in the image): these checks keep it from rotting.
* rust/awry-hip-sys/src/lib.rs declares exactly the functions of include/awry_hip.h, with the same number and kind of arguments;
* regenerating it with tools/gen_rust_sys.py changes nothing;
* rust/awry exposes every item of the reference's public surface and only calls symbols the header declares."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "awry_hip.h")
SYS = os.path.join(ROOT, "rust", "awry-hip-sys", "src", "lib.rs")
SAFE = os.path.join(ROOT, "rust", "awry", "src")


def c_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(awry_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        args = " ".join(m.group(3).split())
        out[m.group(2)] = (m.group(1).strip(), [] if args in ("", "void") else [a.strip() for a in args.split(",")])
    return out


def rust_functions():
    src = open(SYS).read()
    block = src[src.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (awry_[a-z0-9_]+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (m.group(3).strip() if m.group(3) else "", args)
    return out


def kind(c_type):
    """pointer depth + base kind of a C parameter declaration ('const uint64_t *qoff' -> (1, 'u64'))"""
    t = re.sub(r"\b\w+$", "", c_type.strip()) if not c_type.strip().endswith("*") else c_type
    depth = t.count("*")
    base = [w for w in re.findall(r"\w+", t) if w != "const"][0]
    return depth, {"int": "c_int", "uint64_t": "u64", "uint8_t": "u8", "uint32_t": "u32", "float": "f32", "char": "c_char", "void": "c_void"}.get(base, base)


def rust_kind(r_type):
    depth = len(re.findall(r"\*(?:const|mut)", r_type))
    return depth, r_type.split()[-1]


def test_sys_crate_declares_exactly_the_header():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 60
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name, (ret, args) in c.items():
        rret, rargs = r[name]
        assert len(args) == len(rargs), name
        for a, ra in zip(args, rargs):
            assert kind(a) == rust_kind(ra.split(":", 1)[1].strip()), (name, a, ra)
        if ret == "void":
            assert rret == "", name
        else:
            assert kind(ret + " x") == rust_kind(rret), (name, ret, rret)


def test_sys_crate_is_what_the_generator_writes():
    before = open(SYS).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], stdout=subprocess.DEVNULL)
    assert open(SYS).read() == before, "rust/awry-hip-sys/src/lib.rs is stale: run tools/gen_rust_sys.py"


def test_safe_crate_covers_the_reference_surface():
    fm = open(os.path.join(SAFE, "fm_index.rs")).read()
    for item in ("pub struct FmIndex", "pub struct FmBuildArgs", "pub fn new(args: &FmBuildArgs) -> Result<Self, anyhow::Error>",
                 "pub fn load(fm_file_src: &Path) -> Result<FmIndex, std::io::Error>", "pub fn save(&self, file_output_src: &Path) -> Result<(), std::io::Error>",
                 "pub fn count_string(&self, query: &str) -> u64", "pub fn locate_string(&self, query: &str) -> Vec<LocalizedSequencePosition>",
                 "pub fn parallel_count<'a>(&self, queries: impl ParallelIterator<Item = &'a str>) -> Vec<u64>",
                 "pub fn parallel_locate<'a>(&self, queries: impl ParallelIterator<Item = &'a str>) -> Vec<Vec<LocalizedSequencePosition>>",
                 "pub fn update_range_with_symbol(&self, search_range: SearchRange, query_symbol: Symbol) -> SearchRange",
                 "pub fn backstep(&self, search_pointer: SearchPtr) -> SearchPtr", "pub fn initial_search_range(&self, s: Symbol) -> SearchRange",
                 "pub fn alphabet(&self) -> SymbolAlphabet", "pub fn bwt_len(&self) -> u64", "pub fn version_number(&self) -> u64",
                 "pub fn suffix_array_compression_ratio(&self) -> u64", "pub fn prefix_sums(&self) -> &Vec<u64>"):
        assert item in fm, item
    for field in ("input_file_src: PathBuf", "suffix_array_output_src: Option<PathBuf>", "suffix_array_compression_ratio: Option<u64>",
                  "lookup_table_kmer_len: Option<u8>", "alphabet: SymbolAlphabet", "max_query_len: Option<usize>",
                  "remove_intermediate_suffix_array_file: bool"):
        assert "pub " + field in fm, field
    seq = open(os.path.join(SAFE, "sequence_index.rs")).read()
    assert "#[derive(Clone, Debug, PartialEq, PartialOrd, Eq, Ord, Hash, Default)]" in seq
    for item in ("pub struct LocalizedSequencePosition", "pub fn new(sequence_idx: usize, local_position: usize) -> Self",
                 "pub fn sequence_idx(&self) -> usize", "pub fn local_position(&self) -> usize"):
        assert item in seq, item
    sr = open(os.path.join(SAFE, "search.rs")).read()
    for item in ("pub struct SearchRange", "pub start_ptr: SearchPtr", "pub end_ptr: SearchPtr", "pub fn new(fm_index: &FmIndex, symbol: Symbol) -> Self",
                 "pub fn zero() -> Self", "pub fn is_empty(&self) -> bool", "pub fn len(&self) -> SearchPtr", "pub fn range_iter(&self) -> core::ops::Range<SearchPtr>"):
        assert item in sr, item
    al = open(os.path.join(SAFE, "alphabet.rs")).read()
    for item in ("pub enum SymbolAlphabet", "Nucleotide,", "Amino,", "pub struct Symbol", "pub fn new_ascii(alphabet: SymbolAlphabet, ascii: char) -> Symbol",
                 "pub fn new_index(alphabet: SymbolAlphabet, index: u8) -> Symbol"):
        assert item in al, item
    # every sys:: symbol the safe crate calls exists in the header
    declared = set(c_functions())
    used = set()
    for f in os.listdir(SAFE):
        used |= set(re.findall(r"sys::(awry_[a-z0-9_]+)\b", open(os.path.join(SAFE, f)).read()))
    types = {"awry_index_t", "awry_pos_t", "awry_range_t", "awry_build_args_t"}
    assert used - types <= declared, sorted(used - types - declared)
    assert {"awry_build", "awry_load", "awry_save", "awry_count_batch", "awry_locate_batch", "awry_count", "awry_locate", "awry_update_range",
            "awry_backstep", "awry_set_devices", "awry_free"} <= used
