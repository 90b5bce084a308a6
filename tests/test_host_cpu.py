"""CPU-only checks of the product's host logic and of the C-ABI boundary (no kernel launches):
the library loads and exports every declared symbol, refuses to search without a GPU, and its host-side
index construction (suffix array, device-layout packing, .awry reader) agrees with the oracle bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import awry_amd
from awry_amd import _lib
from awry_amd.fm_index import ERR_ARG, ERR_FORMAT, ERR_IO, ERR_NO_DEVICE, AwryError, FmBuildArgs, FmIndex
from tests import synth


def test_library_exports_every_declared_symbol():
    L = awry_amd.load_library()
    names = _lib.header_symbols()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_search_path():
    """queries before set_devices() fail loudly; set_devices() without a GPU fails loudly"""
    text, st, hd = synth.make_text(300, 0, 1)
    ix = FmIndex.from_text(text, 0, 8, 0, st, hd)
    with pytest.raises(AwryError) as e:
        ix.count_string("ACGT")
    assert e.value.code == ERR_NO_DEVICE
    with pytest.raises(AwryError) as e:
        ix.parallel_locate(["ACGT"])
    assert e.value.code == ERR_NO_DEVICE
    with pytest.raises(AwryError) as e:
        ix.set_devices([])
    assert e.value.code == ERR_NO_DEVICE
    import shutil
    if not os.path.exists("/dev/kfd"):
        with pytest.raises(AwryError) as e:
            ix.set_devices([0])
        assert e.value.code in (ERR_NO_DEVICE, -4)


def test_symbol_index_map_matches_oracle(oracle):
    L, O = awry_amd.load_library(), oracle.lib()
    for alpha in (0, 1):
        for a in range(256):
            assert L.awry_symbol_index(alpha, a) == O.orc_ascii_to_index(alpha, a), (alpha, a)


@pytest.mark.parametrize("alphabet,n,recs,nfrac,seed", [
    (0, 1, 1, 0, 0), (0, 2, 1, 0, 1), (0, 7, 1, 0, 2), (0, 1000, 1, 0, 3), (0, 5000, 6, 0.2, 4),
    (1, 3000, 5, 0.05, 5), (0, 70000, 3, 0.07, 6), (1, 20000, 40, 0.0, 7)])
def test_host_suffix_array_matches_oracle(oracle, alphabet, n, recs, nfrac, seed):
    text, _, _ = synth.make_text(n, alphabet, seed, recs, nfrac)
    sa = np.zeros(len(text), dtype=np.uint64)
    rc = awry_amd.load_library().awry_host_suffix_array(text.ctypes.data, len(text), sa.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0
    assert np.array_equal(sa, oracle.suffix_array(text))


def test_host_suffix_array_degenerate_texts(oracle):
    L = awry_amd.load_library()
    for t in (b"$", b"A$", b"AAAAAAAAAAAAAAAA$", b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN$", b"ACACACACACACACAC$", b"TTTTGGGGCCCCAAAA$",
              b"ACGTNACGTNACGTN$", b"GATTACAGATTACAGATTACA$"):
        a = np.frombuffer(t, dtype=np.uint8)
        sa = np.zeros(len(a), dtype=np.uint64)
        assert L.awry_host_suffix_array(a.ctypes.data, len(a), sa.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        assert np.array_equal(sa, oracle.suffix_array(a)), t


@pytest.mark.parametrize("alphabet,n,recs,nfrac,ratio,seed", [
    (0, 1847, 1, 0, 8, 0), (1, 300, 1, 0, 8, 999), (0, 30000, 5, 0.07, 8, 3), (1, 9000, 20, 0.02, 3, 4),
    (0, 255, 1, 0, 1, 5), (0, 256, 2, 0, 7, 6), (0, 511, 1, 0.3, 16, 7), (0, 100000, 1, 0, 8, 8)])
def test_host_build_matches_oracle_bit_for_bit(oracle, alphabet, n, recs, nfrac, ratio, seed):
    """planes + milestones (reference layout), packed sampled SA and prefix sums: src/fm_index.rs:203-240"""
    text, st, hd = synth.make_text(n, alphabet, seed, recs, nfrac)
    ix = FmIndex.from_text(text, alphabet, ratio, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, alphabet, ratio, 0, st, hd)
    assert ix.bwt_len() == oi.bwt_len() == n + 1 and ix.alphabet() == alphabet and ix.version_number() == 1
    assert ix.suffix_array_compression_ratio() == ratio and ix.lookup_table_kmer_len() == oi.kmer_len()
    assert np.array_equal(ix.prefix_sums(), oi.prefix_sums())
    assert np.array_equal(ix.sa_words(), oi.sa_words())
    assert np.array_equal(ix.reference_block_words(), oi.block_words())
    assert ix.sequences() == oi.sequences()
    assert oi.symbol_at(ix.sentinel_row()) == 0


@pytest.mark.parametrize("alphabet", [0, 1])
def test_load_reference_format_file(oracle, tmp_path, alphabet):
    """an .awry v1 file written by the oracle's restatement of src/fm_index_file.rs:42-106 loads identically"""
    text, st, hd = synth.make_text(6000, alphabet, 31, 4, 0.05)
    oi = oracle.OracleIndex.from_text(text, alphabet, 5, 3, st, hd)
    p = str(tmp_path / "o.awry")
    oi.save(p)
    ix = FmIndex.load(p)
    built = FmIndex.from_text(text, alphabet, 5, 3, st, hd)
    assert ix.bwt_len() == oi.bwt_len() and ix.alphabet() == alphabet and ix.lookup_table_kmer_len() == 3
    assert ix.suffix_array_compression_ratio() == 5 and ix.sequences() == oi.sequences()
    assert np.array_equal(ix.prefix_sums(), oi.prefix_sums()) and np.array_equal(ix.sa_words(), oi.sa_words())
    assert np.array_equal(ix.reference_block_words(), oi.block_words())
    assert np.array_equal(ix.device_block_words(), built.device_block_words())
    assert ix.sentinel_row() == built.sentinel_row()


def test_load_rejects_bad_files(tmp_path):
    p = str(tmp_path / "bad.awry")
    open(p, "wb").write(b"not an index at all, definitely")
    with pytest.raises(AwryError) as e:
        FmIndex.load(p)
    assert e.value.code == ERR_FORMAT
    open(p, "wb").write(b"AWRY-Index\n" + (1).to_bytes(8, "little") + (8).to_bytes(8, "little"))
    with pytest.raises(AwryError) as e:
        FmIndex.load(p)
    assert e.value.code in (ERR_IO, ERR_FORMAT)
    with pytest.raises(AwryError) as e:
        FmIndex.load(str(tmp_path / "missing.awry"))
    assert e.value.code == ERR_IO


def test_build_from_fasta_and_fastq(oracle, tmp_path):
    """FmIndex::new text model: records joined by 'N', trailing '$' (src/fm_index.rs:148-153)"""
    text, st, hd = synth.make_text(2500, 0, 77, 6)
    fa = str(tmp_path / "x.fa")
    synth.write_fasta(fa, text, st, hd, 70)
    a = FmIndex.new(FmBuildArgs(fa, None, None, None, 0, None, True))
    b = FmIndex.from_text(text, 0, 8, 0, st, hd)
    assert np.array_equal(a.device_block_words(), b.device_block_words()) and a.sequences() == b.sequences()
    assert np.array_equal(a.sa_words(), b.sa_words()) and a.lookup_table_kmer_len() == 10
    fq = str(tmp_path / "x.fq")
    ends = [s - 1 for s in st[1:]] + [len(text) - 1]
    with open(fq, "wb") as f:
        for s, e, h in zip(st, ends, hd):
            f.write(b"@" + h.encode() + b" extra\n" + bytes(text[s:e]).lower() + b"\n+\n" + b"I" * (e - s) + b"\n")
    c = FmIndex.new(FmBuildArgs(fq, alphabet=0))
    assert np.array_equal(c.device_block_words(), b.device_block_words()) and c.sequences() == b.sequences()
    with pytest.raises(AwryError) as e:
        FmIndex.new(FmBuildArgs(str(tmp_path / "nope.fa")))
    assert e.value.code == ERR_IO


def test_argument_errors():
    with pytest.raises(AwryError) as e:
        FmIndex.from_text(b"ACGT", 0)  # no trailing '$'
    assert e.value.code in (ERR_IO, ERR_ARG)
    with pytest.raises(AwryError):
        FmIndex.from_text(b"ACGT$", 2)


def test_query_file_ingestion(tmp_path):
    """FASTA (wrapped lines) and FASTQ records become CSR query batches, bytes as written"""
    from awry_amd.fm_index import read_query_file
    fa = tmp_path / "q.fa"
    fa.write_bytes(b">a desc\nACGT\nacgtN\n>b\n\nGG\n>c\nT\n")
    qb, qo = read_query_file(str(fa))
    assert qo.tolist() == [0, 9, 11, 12] and bytes(qb) == b"ACGTacgtNGGT"
    fq = tmp_path / "q.fq"
    fq.write_bytes(b"@r1\nACGTT\n+\nIIIII\n@r2 x\nGATTACA\n+r2\n@@@@@@@\n")
    qb, qo = read_query_file(str(fq))
    assert qo.tolist() == [0, 5, 12] and bytes(qb) == b"ACGTTGATTACA"
    with pytest.raises(AwryError) as e:
        read_query_file(str(tmp_path / "missing.fq"))
    assert e.value.code == ERR_IO
    # odd but legal inputs: empty file, no newline at the end, CRLF, blank lines between records, header without sequence
    (tmp_path / "empty.fa").write_bytes(b"")
    qb, qo = read_query_file(str(tmp_path / "empty.fa"))
    assert qo.tolist() == [0] and len(qb) == 0
    (tmp_path / "tail.fq").write_bytes(b"@a\r\nAC\r\n+\r\nII\r\n\r\n@b\nGGT\n+\nIII")
    qb, qo = read_query_file(str(tmp_path / "tail.fq"))
    assert qo.tolist() == [0, 2, 5] and bytes(qb) == b"ACGGT"
    (tmp_path / "hdr.fa").write_bytes(b"ACG\n>x\n>y\nTT")
    qb, qo = read_query_file(str(tmp_path / "hdr.fa"))
    assert qo.tolist() == [0, 3, 3, 5] and bytes(qb) == b"ACGTT"
    with pytest.raises(AwryError):
        FmIndex.new(FmBuildArgs(str(tmp_path / "empty.fa")))


def test_query_file_ingestion_parallel_chunks(tmp_path):
    """files of more than a few MiB are cut at record boundaries and parsed by several threads: same CSR batch as a
    line-by-line parse, with quality lines that start with '@' or '+', CRLF line ends, wrapped FASTA and empty records"""
    from awry_amd.fm_index import read_query_file
    rng = np.random.default_rng(3)
    nt = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    seqs, fq = [], []
    for i in range(60000):
        L = int(rng.integers(0, 300)) if i % 97 else 0
        sq = bytes(nt[rng.integers(0, len(nt), size=L)])
        ql = bytes(rng.integers(33, 74, size=L).astype(np.uint8))
        if i % 5 == 0 and L:
            ql = b"@" + ql[1:]
        if i % 7 == 0 and L:
            ql = b"+" + ql[1:]
        eol = b"\r\n" if i % 11 == 0 else b"\n"
        fq.append(b"@read%d some description" % i + eol + sq + eol + (b"+read%d" % i if i % 3 else b"+") + eol + ql + eol)
        seqs.append(sq)
    path = tmp_path / "big.fq"
    path.write_bytes(b"".join(fq))
    assert path.stat().st_size > (8 << 20)
    qb, qo = read_query_file(str(path))
    assert len(qo) == len(seqs) + 1 and np.array_equal(np.diff(qo), [len(x) for x in seqs])
    assert bytes(qb) == b"".join(seqs)
    fa = []
    for i, sq in enumerate(seqs):
        w = 60 if i % 2 else 71
        fa.append(b">s%d x\n" % i + b"".join(sq[j:j + w] + (b"\r\n" if i % 13 == 0 else b"\n") for j in range(0, len(sq), w)) + (b"\n" if i % 17 == 0 else b""))
    path = tmp_path / "big.fa"
    path.write_bytes(b"".join(fa))
    assert path.stat().st_size > (8 << 20)
    qb, qo = read_query_file(str(path))
    assert np.array_equal(np.diff(qo), [len(x) for x in seqs]) and bytes(qb) == b"".join(seqs)


def test_sequence_file_reader_parallel_chunks(tmp_path, oracle):
    """FmIndex::new on a multi-record FASTA big enough for the chunked reader: same index as from the in-memory text"""
    rng = np.random.default_rng(4)
    recs = [synth.NT[rng.integers(0, 4, size=int(rng.integers(1000, 400000)))] for _ in range(60)]
    path = tmp_path / "multi.fa"
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">chr%d description %d\n" % (i, i))
            low = bytes(r).lower() if i % 4 == 0 else bytes(r)
            f.write(b"\n".join(low[j:j + 80] for j in range(0, len(low), 80)) + b"\n")
    assert path.stat().st_size > (8 << 20)
    ix = FmIndex.new(FmBuildArgs(str(path), suffix_array_compression_ratio=8, lookup_table_kmer_len=4))
    text = np.concatenate([np.concatenate([r, np.frombuffer(b"N", np.uint8)]) for r in recs])[:-1]
    text = np.concatenate([text, np.frombuffer(b"$", np.uint8)])
    starts = np.concatenate([[0], np.cumsum([len(r) + 1 for r in recs])[:-1]])
    want = FmIndex.from_text(text, 0, 8, 4, [int(x) for x in starts], ["chr%d" % i for i in range(len(recs))], build_device=awry_amd.fm_index.BUILD_HOST)
    assert ix.bwt_len() == want.bwt_len() and np.array_equal(ix.device_block_words(), want.device_block_words())
    assert ix.sequences() == want.sequences() and np.array_equal(ix.sa_words(), want.sa_words())


def test_c_abi_header_is_plain_c(tmp_path):
    """include/awry_hip.h compiles as C11 (no C++ in the boundary) and a C caller links the library"""
    import subprocess
    src = tmp_path / "c_caller.c"
    src.write_text('#include "awry_hip.h"\n#include <stdio.h>\nint main(void) {\n  awry_index_t *ix = 0;\n'
                   '  int rc = awry_load("/nonexistent.awry", &ix);\n  printf("%d %s\\n", rc, awry_last_error());\n'
                   '  return rc == AWRY_ERR_IO && ix == 0 ? 0 : 1;\n}\n')
    exe = tmp_path / "c_caller"
    libdir = os.path.dirname(awry_amd.lib_path())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lawry_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.stdout, r.stderr)


def _ref_pack_nt2(q2d):
    """numpy restatement of the packed-query layout: letter j in word j // 32, bits 2 (j % 32), A0 C1 G2 T3"""
    n, L = q2d.shape
    c = q2d & 0xDF
    code = np.full(q2d.shape, 255, np.uint8)
    for ch, v in ((65, 0), (67, 1), (71, 2), (84, 3)):
        code[c == ch] = v
    bad = np.nonzero((code == 255).any(1))[0]
    code[code == 255] = 0
    w = np.zeros((n, (L + 31) // 32), np.uint64)
    for j in range(L):
        w[:, j // 32] |= code[:, j].astype(np.uint64) << np.uint64(2 * (j % 32))
    return w, bad


@pytest.mark.parametrize("L", [1, 7, 31, 32, 33, 64, 101, 150])
def test_host_packer_matches_layout_and_lists_other_letters(L):
    """the host half of awry_count_batch (AVX2 packer on the worker pool): words equal the packed layout the kernels
    read, lower case folds, and exactly the queries with a byte outside ACGTacgt are listed -- uniform and ragged"""
    lib = awry_amd.load_library()
    assert lib.awry_host_threads() >= 1
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    rng = np.random.default_rng(L)
    n = 20011
    q = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), size=(n, L))
    bi = rng.integers(0, n, 50)
    q[bi, rng.integers(0, L, 50)] = rng.choice(np.frombuffer(b"NnU$\xc1RY@`#", np.uint8), 50)
    W = (L + 31) // 32
    words, bad, nb = np.zeros((n, W), np.uint64), np.zeros(n, np.uint32), C.c_uint64()
    flat = np.ascontiguousarray(q.reshape(-1))
    assert lib.awry_host_pack_nt2(flat.ctypes.data, None, n, L, words.ctypes.data_as(u64p), None, bad.ctypes.data_as(u32p), C.byref(nb)) == 0
    w, b = _ref_pack_nt2(q)
    good = np.ones(n, bool)
    good[b] = False
    assert np.array_equal(bad[:nb.value], b.astype(np.uint32))
    assert np.array_equal(words[good], w[good])
    # ragged: per-query lengths 1..L at a stride of W words
    lens = rng.integers(1, L + 1, n)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    rb = np.concatenate([q[i, :lens[i]] for i in range(n)])
    words2, lo = np.zeros((n, W), np.uint64), np.zeros(n, np.uint32)
    assert lib.awry_host_pack_nt2(rb.ctypes.data, off.ctypes.data_as(u64p), n, L, words2.ctypes.data_as(u64p), lo.ctypes.data_as(u32p),
                                  bad.ctypes.data_as(u32p), C.byref(nb)) == 0
    assert np.array_equal(lo, lens.astype(np.uint32))
    listed = set(bad[:nb.value].tolist())
    for i in range(0, n, 37):
        ww, bb = _ref_pack_nt2(q[i:i + 1, :lens[i]])
        assert (i in listed) == (len(bb) == 1)
        if not len(bb):
            exp = np.zeros(W, np.uint64)
            exp[:ww.shape[1]] = ww[0]
            assert np.array_equal(words2[i], exp), (L, i)


@pytest.mark.parametrize("alphabet", [0, 1])
def test_text_is_canonicalised_before_suffix_sorting(alphabet):
    """a text with lower case, IUPAC codes / non-standard residues and U is indexed as its canonical form (the map queries go
    through): same index as the canonical text gives, bit for bit; an inner '$' / '#' is an argument error"""
    rng = np.random.default_rng(alphabet)
    letters = b"ACGTacgtRYKMSWNnUu" if alphabet == 0 else b"ACDEFGHIKLMNPQRSTVWYacdxXBJOUZbz*"
    body = rng.choice(np.frombuffer(letters, np.uint8), size=5000)
    text = np.concatenate([body, np.frombuffer(b"$", np.uint8)])
    lib = awry_amd.load_library()
    canon = np.array([ord("$")] * 256, np.uint8)
    names = b"$ACGNT" if alphabet == 0 else b"$ACDEFGHIKLMNPQRSTVWXY"
    for b in range(256):
        canon[b] = names[lib.awry_symbol_index(alphabet, b)]
    ctext = canon[text]
    assert (ctext != text).any() and ctext[-1] == ord("$")
    a = FmIndex.from_text(text, alphabet, 4, 0, build_device=awry_amd.fm_index.BUILD_HOST)
    b = FmIndex.from_text(ctext, alphabet, 4, 0, build_device=awry_amd.fm_index.BUILD_HOST)
    assert np.array_equal(a.device_block_words(), b.device_block_words())
    assert np.array_equal(a.sa_words(), b.sa_words()) and np.array_equal(a.prefix_sums(), b.prefix_sums())
    bad = text.copy()
    bad[100] = ord("#")
    with pytest.raises(AwryError) as e:
        FmIndex.from_text(bad, alphabet, 4, 0, build_device=awry_amd.fm_index.BUILD_HOST)
    assert e.value.code == ERR_ARG


@pytest.mark.parametrize("alphabet,kmer_len", [(0, 0), (0, 4), (1, 0), (1, 3)])
def test_save_without_a_device_is_byte_identical_to_reference_format(oracle, tmp_path, alphabet, kmer_len):
    """FmIndex::save needs no GPU: with no replica the reference's (partially populated) k-mer table is computed on the
    host copy (src/kmer_lookup_table.rs:121-167), and the file equals the oracle's writer byte for byte"""
    text, st, hd = synth.make_text(9000, alphabet, 5 + alphabet, 3, 0.04)
    ix = FmIndex.from_text(text, alphabet, 6, kmer_len, st, hd, build_device=awry_amd.fm_index.BUILD_HOST)
    oi = oracle.OracleIndex.from_text(text, alphabet, 6, kmer_len, st, hd)
    a, b = str(tmp_path / "a.awry"), str(tmp_path / "b.awry")
    ix.save(a)
    oi.save(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    with pytest.raises(ValueError):
        FmIndex.from_text(text, alphabet, 6, kmer_len, st, hd[:-1])


def test_host_pool_serves_concurrent_callers():
    """the worker pool takes jobs from several threads at once (one per replica of a batch call over N replicas): eight
    threads pack different batches concurrently, each many times, and every result equals the serial one"""
    import threading
    lib = awry_amd.load_library()
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    rng = np.random.default_rng(8)
    L, n = 31, 300_000  # large enough for the packer to cut the batch over the pool
    batches = [np.ascontiguousarray(synth.NT[rng.integers(0, 4, size=(n, L))]) for _ in range(8)]
    for b in batches[::2]:
        b[rng.integers(0, n, 50), rng.integers(0, L, 50)] = ord("N")

    def pack(q):
        words, bad, nb = np.zeros((n, 1), np.uint64), np.zeros(n, np.uint32), C.c_uint64()
        assert lib.awry_host_pack_nt2(q.ctypes.data, None, n, L, words.ctypes.data_as(u64p), None, bad.ctypes.data_as(u32p), C.byref(nb)) == 0
        good = np.ones(n, bool)
        good[bad[:nb.value]] = False
        return words[good].copy(), np.sort(bad[:nb.value]).copy()

    want = [pack(q) for q in batches]
    errors = []

    def hammer(i):
        try:
            for _ in range(12):
                w, b = pack(batches[i])
                assert np.array_equal(w, want[i][0]) and np.array_equal(b, want[i][1])
                dst = np.empty_like(batches[i])
                lib.awry_host_memcpy(dst.ctypes.data, batches[i].ctypes.data, dst.nbytes)
                assert np.array_equal(dst, batches[i])
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=hammer, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in th), errors
