"""Pins the oracle's index + search path against (a) the committed golden vectors (pure-Python brute
force, tests/golden/make_golden.py) and (b) the definition src/fm_index.rs:612-664 pins, on seeded
texts.  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

from tests import synth

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


def load_golden(path):
    with open(path) as f:
        return json.load(f)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
@pytest.mark.parametrize("sa_ratio,kmer_len", [(8, 0), (1, 3), (5, 12)])
def test_oracle_matches_golden(oracle, path, sa_ratio, kmer_len):
    g = load_golden(path)
    if g["alphabet"] == 1:
        kmer_len = min(kmer_len, 5)  # 20^k table entries
    idx = oracle.OracleIndex.from_text(g["text"], g["alphabet"], sa_ratio, kmer_len, g["seq_starts"], g["headers"])
    starts = np.array(g["seq_starts"])
    for rec in g["queries"]:
        assert idx.count_string(rec["q"]) == rec["count"], rec["q"]
        gpos, pos = idx.locate_string(rec["q"])
        assert sorted(gpos.tolist()) == rec["pos"], rec["q"]
        for gp, (si, lp) in zip(gpos.tolist(), pos):  # intended get_seq_location semantics (SURVEY a-17)
            assert si == int(np.searchsorted(starts, gp, side="right") - 1) and lp == gp - starts[si]


def test_suffix_array_is_sorted(oracle):
    text, _, _ = synth.make_text(3000, 0, 3, n_records=4, n_frac=0.1)
    sa = oracle.suffix_array(text)
    tb = bytes(text)
    assert sorted(sa.tolist()) == list(range(len(tb)))
    suf = [tb[i:] for i in sa.tolist()]
    assert all(suf[i] < suf[i + 1] for i in range(len(suf) - 1))
    assert sa[0] == len(tb) - 1  # '$' sorts first


@pytest.mark.parametrize("alphabet,n,qlen", [(0, 20000, 9), (1, 8000, 3)])
def test_oracle_vs_c_brute_force(oracle, alphabet, n, qlen):
    text, starts, hdr = synth.make_text(n, alphabet, 21, n_records=3, n_frac=0.03)
    idx = oracle.OracleIndex.from_text(text, alphabet, 8, 0, starts, hdr)
    L = oracle.lib()
    qs = np.concatenate([synth.random_queries(100, qlen, alphabet, 4), synth.sampled_queries(text, 100, qlen, 5, False, alphabet)])
    for q in qs:
        cnt = L.orc_brute_count(alphabet, text.ctypes.data, len(text), q.ctypes.data, len(q))
        assert idx.count_string(q) == cnt
        buf = np.zeros(max(cnt, 1), dtype=np.uint64)
        L.orc_brute_locate(alphabet, text.ctypes.data, len(text), q.ctypes.data, len(q), buf.ctypes.data_as(oracle.u64p), len(buf))
        gpos, _ = idx.locate_string(q)
        assert sorted(gpos.tolist()) == buf[:cnt].tolist()


def test_path_a_and_b_agree(oracle):
    """src/fm_index.rs:402-438: the k-mer-table path and the plain path return the same range (SURVEY a-11)"""
    text, starts, hdr = synth.make_text(5000, 0, 8)
    a = oracle.OracleIndex.from_text(text, 0, 8, 3, starts, hdr)
    b = oracle.OracleIndex.from_text(text, 0, 8, 12, starts, hdr)
    for L in (1, 2, 3, 4, 11, 12, 13, 20):
        for q in np.concatenate([synth.random_queries(40, L, 0, L), synth.sampled_queries(text, 40, L, L + 1)]):
            assert a.count_string(q) == b.count_string(q)
            ra, rb = a.search_range(q), b.search_range(q)
            if ra[0] <= ra[1]:
                assert ra == rb


def test_undefined_queries_are_flagged(oracle):
    text, starts, hdr = synth.make_text(500, 0, 1)
    idx = oracle.OracleIndex.from_text(text, 0, 8, 0, starts, hdr)
    for bad in ("", "AC$GT", "#", "$", "AC\xe9"):
        with pytest.raises(ValueError):
            idx.count_string(bad)


def test_backstep_walks_the_text_backwards(oracle):
    """src/fm_index.rs:585-593: LF-mapping; from row of suffix i it reaches the row of suffix i-1"""
    text, starts, hdr = synth.make_text(1200, 0, 9, n_records=2, n_frac=0.05)
    sa = oracle.suffix_array(text)
    idx = oracle.OracleIndex.from_text(text, 0, 4, 0, starts, hdr, sa=sa)
    inv = np.empty(len(sa), dtype=np.int64)
    inv[sa.astype(np.int64)] = np.arange(len(sa))
    for i in range(1, len(text)):
        assert idx.backstep(int(inv[i])) == inv[i - 1]
    assert idx.backstep(int(inv[0])) == 0  # the sentinel row steps to row 0 (src/fm_index.rs:587-589)
    ps = idx.prefix_sums()
    assert ps[0] == 0 and ps[1] == 1 and ps[-1] == len(text)


def test_seq_location_matches_reference_where_it_terminates(oracle):
    """src/sequence_index.rs:108-141 does not terminate for most positions with >1 record (SURVEY a-17);
    wherever it does return, the oracle's intended-semantics answer is identical."""
    text, starts, hdr = synth.make_text(400, 0, 2, n_records=5)
    idx = oracle.OracleIndex.from_text(text, 0, 8, 0, starts, hdr)
    term = 0
    for p in range(len(text) - 1):
        ref = idx.seq_location_ref(p)
        if ref is not None:
            term += 1
            assert ref == idx.seq_location(p)
    assert 0 < term < len(text) - 1
    one = oracle.OracleIndex.from_text(*synth.make_text(300, 0, 3)[:1], 0, 8, 0)
    assert all(one.seq_location_ref(p) == (0, p) == one.seq_location(p) for p in range(300))


@pytest.mark.parametrize("alphabet", [0, 1])
def test_save_load_round_trip(oracle, tmp_path, alphabet):
    """src/fm_index.rs:1046-1088 save_load_equality_test, field by field"""
    text, starts, hdr = synth.make_text(4000, alphabet, 17, n_records=7)
    a = oracle.OracleIndex.from_text(text, alphabet, 8, 0 if alphabet == 0 else 3, starts, hdr)
    p = str(tmp_path / "x.awry")
    a.save(p)
    b = oracle.OracleIndex.load(p)
    assert a.alphabet() == b.alphabet() and a.bwt_len() == b.bwt_len() and a.version_number() == b.version_number() == 1
    assert a.suffix_array_compression_ratio() == b.suffix_array_compression_ratio() and a.kmer_len() == b.kmer_len()
    for f in ("prefix_sums", "block_words", "sa_words", "kmer_table"):
        assert np.array_equal(getattr(a, f)(), getattr(b, f)())
    assert a.sequences() == b.sequences()
    raw = open(p, "rb").read()
    assert raw[:11] == b"AWRY-Index\n" and int.from_bytes(raw[11:19], "little") == 1  # src/fm_index_file.rs:18,165-181
    for q in synth.sampled_queries(text, 20, 6, 3, False, alphabet):
        assert a.count_string(q) == b.count_string(q)


def test_kmer_table_population_pattern(oracle):
    """src/kmer_lookup_table.rs:121-167: only slots whose base-sigma digits are all in 1..sigma-1 are
    written (A,C,G for nt); every other slot stays SearchRange::zero() = {1,0} (SURVEY a-12)"""
    text, starts, hdr = synth.make_text(3000, 0, 4)
    idx = oracle.OracleIndex.from_text(text, 0, 8, 3, starts, hdr)
    tab = idx.kmer_table()
    assert tab.shape == (64, 2)
    for slot in range(64):
        digits = [(slot // 4**j) % 4 for j in range(3)]
        if all(d in (1, 2, 3) for d in digits):
            kmer = "".join("$ACG"[d] for d in reversed(digits))  # digit 0 = last char
            sp, ep = idx.search_range(kmer) if idx.count_string(kmer) else tuple(tab[slot])
            assert tuple(tab[slot]) == (sp, ep)
            assert (0 if tab[slot][0] > tab[slot][1] else tab[slot][1] - tab[slot][0] + 1) == idx.count_string(kmer)
        else:
            assert tuple(tab[slot]) == (1, 0)


def test_parallel_batch_keeps_input_order(oracle):
    text, starts, hdr = synth.make_text(30000, 0, 6, n_records=2)
    idx = oracle.OracleIndex.from_text(text, 0, 8, 0, starts, hdr)
    q2d = np.concatenate([synth.random_queries(1500, 12, 0, 1), synth.sampled_queries(text, 1500, 12, 2)])
    qb, qo = synth.fixed_to_csr(q2d)
    c1, t1 = idx.parallel_count(qb, qo, 1)
    c4, t4 = idx.parallel_count(qb, qo, 4)
    assert np.array_equal(c1, c4) and t1 == t4 and t1["queries"] == 3000
    assert [idx.count_string(q) for q in q2d[:50]] == c1[:50].tolist()
    assert t1["steps"] <= 3000 * 11 and t1["steps"] <= t1["block_reads"] <= 2 * t1["steps"]
    off, gpos, pos, t = idx.parallel_locate(qb, qo, 3)
    assert np.array_equal(np.diff(off), c1) and t["hits"] == int(c1.sum())
    for i in (0, 7, 1600, 2999):
        g, p = idx.locate_string(q2d[i])
        assert np.array_equal(g, gpos[off[i]:off[i + 1]]) and [tuple(r) for r in pos[off[i]:off[i + 1]].tolist()] == p


def test_fasta_text_model(oracle, tmp_path):
    """records joined by 'N'/'X', one trailing '$' (src/fm_index.rs:148-153,220-223)"""
    text, starts, hdr = synth.make_text(700, 0, 12, n_records=4)
    p = str(tmp_path / "t.fa")
    synth.write_fasta(p, text, starts, hdr, width=60)
    a = oracle.OracleIndex.from_fasta(p, 0, 8, 0)
    assert a.text() == bytes(text) and [s for s, _ in a.sequences()] == starts and [h for _, h in a.sequences()] == hdr
    b = oracle.OracleIndex.from_text(text, 0, 8, 0, starts, hdr)
    assert np.array_equal(a.block_words(), b.block_words()) and np.array_equal(a.sa_words(), b.sa_words())
