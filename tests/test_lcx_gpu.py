"""GPU tests of the left-context index (awry_amd/csrc/lcx.hip.h): the structure itself against a host recomputation,
and -- the bar -- bit-exact counts, locations and location ORDER against the oracle on repeat-rich texts, where the index
carries most queries, with the index on and off."""
import ctypes as C

import numpy as np
import pytest

from awry_amd.fm_index import FmIndex, pack_queries
from tests import synth

pytestmark = pytest.mark.gpu

NT_CODE = np.full(256, 8, dtype=np.uint8)
NT_CODE[[65, 67, 71, 84]] = [0, 1, 2, 3]


def family_text(n, seed, families, n_records=2, gap=0.01):
    """an i.i.d. text with planted repeat families (unit, copies, divergence): copy counts high enough that seed buckets hold
    hundreds to tens of thousands of rows at test sizes (the multi-level searches of the index)"""
    rng = np.random.default_rng(seed)
    body = synth.NT[rng.integers(0, 4, size=n, dtype=np.uint8)]
    for unit, copies, div in families:
        cons = synth.NT[rng.integers(0, 4, size=unit, dtype=np.uint8)]
        starts = rng.integers(0, n - unit, size=copies)
        vals = np.tile(cons, (copies, 1))
        m = rng.random(vals.shape) < div
        vals[m] = synth.NT[rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)]
        body[(starts[:, None] + np.arange(unit)[None, :]).reshape(-1)] = vals.reshape(-1)
    g = int(n * gap)
    if g:
        s = int(rng.integers(0, n - g))
        body[s:s + g] = ord("N")
    starts = [0]
    for c in sorted(rng.choice(np.arange(1, n - 1), size=n_records - 1, replace=False).tolist()):
        body[c] = ord("N")
        starts.append(c + 1)
    text = np.concatenate([body, np.frombuffer(b"$", np.uint8)])
    return text, starts, ["seq%d" % i for i in range(len(starts))]


def host_suffix_array(text):
    from awry_amd import _lib
    L = _lib.load_library()
    sa = np.zeros(len(text), dtype=np.uint64)
    assert L.awry_host_suffix_array(text.ctypes.data, len(text), sa.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    return sa.astype(np.int64)


def packed_windows(code, starts, m):
    """windows code[s .. s + m) (all starts valid) -> (uint64 packed 2 bits per letter, first letter least significant; all-ACGT mask)"""
    win = code[starts[:, None] + np.arange(m)[None, :]]
    ok = (win < 4).all(axis=1)
    val = np.zeros(len(starts), dtype=np.uint64)
    for j in range(m):
        val |= (win[:, j].astype(np.uint64) & np.uint64(3)) << np.uint64(2 * j)
    return val, ok


def test_structure_matches_a_host_recomputation():
    """every bucket of 2+ rows: its entries are a permutation of its rows, each with its suffix's text position; the keys are
    the 32 letters in front of that position; complete entries come first, ascending; the last key slot of a bucket with
    incomplete entries holds their number"""
    text, st, hd = family_text(600_000, 5, [(300, 1500, 0.10), (120, 4000, 0.03), (2000, 40, 0.02)], 3, 0.02)
    ix = FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
    assert ix.lcx_enabled()
    k = ix.seed_kmer_len()
    keys, rowpos = ix.debug_lcx()
    sa = host_suffix_array(text)
    n = len(text)
    code = NT_CODE[text]
    pad = np.concatenate([code, np.full(64, 8, np.uint8)])
    kmer, ok = packed_windows(pad, sa, k)
    # buckets: maximal runs of rows with the same all-ACGT k-mer
    lex = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        lex = (lex << np.uint64(2)) | ((kmer >> np.uint64(2 * j)) & np.uint64(3))
    ident = np.where(ok, lex.astype(np.int64), -1 - np.arange(n))
    cut = np.flatnonzero(np.diff(ident) != 0) + 1
    b_lo = np.concatenate([[0], cut])
    b_hi = np.concatenate([cut, [n]])
    multi = (b_hi - b_lo >= 2) & ok[b_lo]
    assert multi.sum() > 1000 and (b_hi - b_lo)[multi].max() > 300
    pos = (rowpos & np.uint64(0xFFFFFFFF)).astype(np.int64)
    row = (rowpos >> np.uint64(32)).astype(np.int64)
    # context keys of all positions with 32 letters in front
    ctx = np.zeros(n, dtype=np.uint64)
    cok = np.zeros(n, dtype=bool)
    have = np.arange(n) >= 32
    v, o = packed_windows(code, np.arange(n)[have] - 32, 32)
    ctx[have], cok[have] = v, o
    checked = with_tail = 0
    for lo, hi in zip(b_lo[multi].tolist(), b_hi[multi].tolist()):
        r = row[lo:hi]
        assert np.array_equal(np.sort(r), np.arange(lo, hi)), (lo, hi)
        assert np.array_equal(pos[lo:hi], sa[r])
        comp = cok[pos[lo:hi]]
        nc = int(comp.sum())
        assert comp[:nc].all() and not comp[nc:].any(), (lo, hi)  # complete entries first
        kk = keys[lo:lo + nc]
        assert np.array_equal(kk, ctx[pos[lo:lo + nc]])
        assert (kk[1:] >= kk[:-1]).all()
        same = kk[1:] == kk[:-1]
        assert (r[1:nc][same] > r[:nc - 1][same]).all()  # ties in row order (both sorts are stable)
        if nc < hi - lo:
            assert int(keys[hi - 1]) == hi - lo - nc
            with_tail += 1
        checked += 1
    assert checked == int(multi.sum()) and with_tail >= 1
    ix.close()


FAMILIES = [(300, 2500, 0.12), (150, 6000, 0.04), (6000, 60, 0.02), (171, 3000, 0.02), (40, 2000, 0.0)]  # copies at 3 Mbp


@pytest.mark.parametrize("n,seed", [(1_500_000, 3), (3_000_000, 4)])
def test_repeat_rich_counts_and_locations_match_the_oracle(oracle, n, seed):
    """k-mers and reads drawn from a text that is half repeat families -- seed buckets of up to tens of thousands of rows,
    counts of up to thousands -- through the packed kernels and the host entry points: counts, locations and their order are
    the oracle's, and identical with the index switched off"""
    text, st, hd = family_text(n, seed, [(u, c * n // 3_000_000, d) for u, c, d in FAMILIES], 3, 0.02)
    ix = FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    assert ix.lcx_enabled()
    rng = np.random.default_rng(seed)
    for L, nq in ((31, 6000), (21, 3000), (32, 2000), (18, 2000)):
        pres = synth.sampled_queries(text, nq, L, seed + L)
        near = pres[:nq // 2].copy()
        col = rng.integers(0, L, size=len(near))
        near[np.arange(len(near)), col] = synth.NT[(np.searchsorted(synth.NT[:4], near[np.arange(len(near)), col]) + 1) % 4]
        q2d = np.concatenate([pres, near, synth.random_queries(nq // 4, L, 0, L)])
        qb, qo = synth.fixed_to_csr(q2d)
        want, _ = oi.parallel_count(qb, qo, 4)
        assert (want > 100).sum() > 20, "the text should hold high-copy k-mers"
        got, census = ix.count_kmers_nt2(q2d, True, 0, True)
        assert np.array_equal(got, want), L
        if L > ix.seed_kmer_len():
            assert int(census[1]) <= len(q2d) // 50, ("LF steps should be the exception", census)
        assert np.array_equal(ix.parallel_count_csr(qb, qo), want), L
        assert np.array_equal(ix.count_kmers_nt2(q2d, False), want), L
    for L, nq in ((101, 5000), (64, 3000), (49, 2000), (50, 2000), (150, 2000), (250, 1000)):
        pres = synth.sampled_queries(text, nq, L, seed + L)
        near = pres[:nq // 3].copy()
        col = rng.integers(0, L, size=len(near))
        near[np.arange(len(near)), col] = synth.NT[(np.searchsorted(synth.NT[:4], near[np.arange(len(near)), col]) + 1) % 4]
        q2d = np.concatenate([pres, near, synth.random_queries(nq // 5, L, 0, L)])
        q2d = q2d[rng.permutation(len(q2d))]
        qb, qo = synth.fixed_to_csr(q2d)
        ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
        off, gpos, pos = ix.locate_reads_nt2(q2d)
        assert np.array_equal(off, ooff), L
        assert np.array_equal(gpos, ogpos), L   # same order: ascending BWT row (src/fm_index.rs:521)
        assert np.array_equal(pos, opos), L
        hoff, hg, hp = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(hoff, ooff) and np.array_equal(hg, ogpos) and np.array_equal(hp, opos), L
        assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(ooff)), L
    # reads of unequal lengths (the ragged kernels)
    lens = rng.integers(20, 140, size=4000)
    starts = rng.integers(0, n - 200, size=4000)
    qs = [bytes(text[s:s + ln]) for s, ln in zip(starts.tolist(), lens.tolist()) if b"N" not in bytes(text[s:s + ln])]
    qb, qo = pack_queries(qs)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    hoff, hg, hp = ix.parallel_locate_csr(qb, qo)
    assert np.array_equal(hoff, ooff) and np.array_equal(hg, ogpos) and np.array_equal(hp, opos)
    # the same batch with the index off: identical results by LF steps
    ix.set_lcx(False)
    assert not ix.lcx_enabled()
    hoff2, hg2, hp2 = ix.parallel_locate_csr(qb, qo)
    assert np.array_equal(hoff2, ooff) and np.array_equal(hg2, ogpos) and np.array_equal(hp2, opos)
    q2d = synth.sampled_queries(text, 4000, 31, 77)
    want, _ = oi.parallel_count(*synth.fixed_to_csr(q2d), 4)
    got, census_off = ix.count_kmers_nt2(q2d, True, 0, True)
    assert np.array_equal(got, want)
    ix.set_lcx(True)
    assert ix.lcx_enabled()
    got, census_on = ix.count_kmers_nt2(q2d, True, 0, True)
    assert np.array_equal(got, want)
    assert int(census_on[1]) * 20 < int(census_off[1]), ("the index should replace the LF steps", census_on, census_off)
    ix.close()
    oi.close()


def test_reads_next_to_gaps_and_the_text_start(oracle):
    """suffixes whose 32 left letters do not all exist (behind a run of N, a record delimiter, the text's first letters) sit in
    their buckets' unsorted tails: reads and k-mers that end right behind such places still count and locate like the oracle's"""
    rng = np.random.default_rng(9)
    unit = synth.NT[rng.integers(0, 4, size=200, dtype=np.uint8)]
    parts = []
    for j in range(400):  # 400 copies of one unit, every one right behind a delimiter or a short run of N
        cp = unit.copy()
        m = rng.random(200) < 0.03
        cp[m] = synth.NT[rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)]
        parts += [np.frombuffer(b"N" * int(rng.integers(1, 4)), np.uint8), synth.NT[rng.integers(0, 4, size=int(rng.integers(0, 40)), dtype=np.uint8)], cp]
    filler = synth.NT[rng.integers(0, 4, size=300_000, dtype=np.uint8)]
    body = np.concatenate([unit] + parts + [filler])
    text = np.concatenate([body, np.frombuffer(b"$", np.uint8)])
    ix = FmIndex.from_text(text, 0, 8, 0, [0], ["one"]).set_devices([0])
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, [0], ["one"])
    assert ix.lcx_enabled()
    k = ix.seed_kmer_len()
    qs = []
    for L in (k + 1, k + 5, 31, 40, 60, 101):
        for s in rng.integers(0, 200 - L if L < 200 else 1, size=60).tolist():
            qs.append(bytes(unit[s:s + L]))          # windows of the consensus: hundreds of hits, many behind an N
        for s in range(0, 12):
            qs.append(bytes(text[s:s + L]))          # the text's first letters
    qs = [q for q in qs if b"N" not in q and len(q) > 0]
    qb, qo = pack_queries(qs)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    off, gpos, pos = ix.parallel_locate_csr(qb, qo)
    assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(ooff))
    for L in (k + 2, 31):
        q2d = np.stack([unit[s:s + L] for s in range(0, 200 - L, 3)] + [text[s:s + L] for s in range(0, 8)])
        want, _ = oi.parallel_count(*synth.fixed_to_csr(q2d), 4)
        assert np.array_equal(ix.count_kmers_nt2(q2d, True), want)
        assert want.max() > 50
    ix.close()
    oi.close()
