"""GPU parity tests (run on a real MI355X via `pytest -m gpu`): the HIP path, called through the C ABI of
libawry_hip.so, against the committed golden vectors and the CPU oracle on the same seeded inputs, plus
size-independent properties at larger sizes.  Bar: bit-exact (integer work)."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

from awry_amd.fm_index import ERR_INVALID_QUERY, AwryError, FmIndex, SearchRange
from tests import synth

pytestmark = pytest.mark.gpu


def awry_build_host():
    from awry_amd.fm_index import BUILD_HOST
    return BUILD_HOST


def awry_pack(queries):
    from awry_amd.fm_index import pack_queries
    return pack_queries(queries)
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.json")))


def gpu_index(text, alphabet, ratio=8, kmer_len=0, st=(0,), hd=("seq0",), devices=(0,)):
    return FmIndex.from_text(text, alphabet, ratio, kmer_len, st, hd).set_devices(list(devices))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-5] for p in GOLDEN])
@pytest.mark.parametrize("sa_ratio", [8, 1, 5])
def test_golden_vectors(path, sa_ratio):
    g = json.load(open(path))
    ix = gpu_index(g["text"], g["alphabet"], sa_ratio, 0, g["seq_starts"], g["headers"])
    qs = [r["q"] for r in g["queries"]]
    counts = ix.parallel_count(qs)
    assert counts.tolist() == [r["count"] for r in g["queries"]]
    off, gpos, pos = ix.parallel_locate_csr(*__import__("awry_amd").fm_index.pack_queries(qs))
    starts = np.array(g["seq_starts"], dtype=np.uint64)
    for i, r in enumerate(g["queries"]):
        mine = gpos[off[i]:off[i + 1]]
        assert sorted(mine.tolist()) == r["pos"], r["q"]
        si = np.searchsorted(starts, mine, side="right") - 1
        assert np.array_equal(pos[off[i]:off[i + 1], 0], si.astype(np.uint64))
        assert np.array_equal(pos[off[i]:off[i + 1], 1], mine - starts[si])
    for r in g["queries"][:25]:  # scalar entry points
        assert ix.count_string(r["q"]) == r["count"]
        assert sorted(p.local_position + int(starts[p.sequence_idx]) for p in ix.locate_string(r["q"])) == r["pos"]


def mixed_queries(text, alphabet, seed):
    rng = np.random.default_rng(seed)
    qs = []
    for L in (1, 2, 3, 4, 7, 9, 10, 11, 15, 24, 31, 32, 33, 50, 101):
        qs += [bytes(q) for q in synth.random_queries(12, L, alphabet, seed + L)]
        qs += [bytes(q) for q in synth.sampled_queries(text, 12, L, seed + 100 + L, False, alphabet)]
    amb = b"N" if alphabet == 0 else b"X"
    qs += [amb, amb * 2, amb * 5, b"A" + amb, amb + b"A", b"acg" if alphabet == 0 else b"mkv", b"ACGU" if alphabet == 0 else b"BZJ"]
    qs += [q.lower() for q in qs[:20]]
    order = rng.permutation(len(qs))
    return [qs[i] for i in order]


@pytest.mark.parametrize("alphabet,n,recs,nfrac,ratio", [(0, 300000, 5, 0.07, 8), (1, 150000, 60, 0.01, 8), (0, 70000, 1, 0.0, 3)])
def test_count_and_locate_match_oracle(oracle, alphabet, n, recs, nfrac, ratio):
    text, st, hd = synth.make_text(n, alphabet, 11, recs, nfrac)
    ix = gpu_index(text, alphabet, ratio, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, alphabet, ratio, 0, st, hd)
    qs = [q for q in mixed_queries(text, alphabet, 5) if len(q) <= n]
    qb, qo = __import__("awry_amd").fm_index.pack_queries(qs)
    ocounts, _ = oi.parallel_count(qb, qo, 4)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), ocounts)
    # keep locate output bounded: drop queries with huge hit lists (single letters)
    keep = [i for i in range(len(qs)) if ocounts[i] <= 5000]
    qb2, qo2 = __import__("awry_amd").fm_index.pack_queries([qs[i] for i in keep])
    ooff, ogpos, opos, _ = oi.parallel_locate(qb2, qo2, 4)
    off, gpos, pos = ix.parallel_locate_csr(qb2, qo2)
    assert np.array_equal(off, ooff)
    assert np.array_equal(gpos, ogpos)  # same order too: ascending BWT row, src/fm_index.rs:521
    assert np.array_equal(pos, opos)
    # get_search_range_for_string: the reference's own rows, for ABSENT queries too -- awry_search_range runs the reference's
    # step schedule (lookup_table_kmer_len - 1 steps taken whether or not the range is empty, src/kmer_lookup_table.rs:90-110)
    absent = 0
    for q in qs[:120]:
        r = ix.search_range(q)
        assert (r.start_ptr, r.end_ptr) == oi.search_range(q), q
        absent += 0 if oi.count_string(q) else 1
    assert absent >= 10


@pytest.mark.parametrize("n,recs,nfrac", [(200000, 1, 0.0), (400000, 3, 0.07)])
def test_packed_kmer_kernel_matches_oracle(oracle, n, recs, nfrac):
    """the hot quad kernel (packed 2-bit k-mers, seed table on/off) against the oracle and the generic kernel"""
    text, st, hd = synth.make_text(n, 0, 23, recs, nfrac)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    for L in (1, 2, 5, 8, 9, 12, 16, 21, 31, 32):
        q2d = np.concatenate([synth.random_queries(700, L, 0, L), synth.sampled_queries(text, 700, L, 50 + L)])
        qb, qo = synth.fixed_to_csr(q2d)
        want, _ = oi.parallel_count(qb, qo, 4)
        for k in (-1, 0, 1, 4, 7):
            ix.set_seed_kmer_len(k)
            assert np.array_equal(ix.count_kmers_nt2(q2d, True), want), (L, k)
        assert np.array_equal(ix.count_kmers_nt2(q2d, False), want), L
        assert np.array_equal(ix.parallel_count_csr(qb, qo), want), L
    with pytest.raises(AwryError) as e:  # N is not packable: callers must take the generic path
        ix.count_kmers_nt2(np.frombuffer(b"ACGNACGT", dtype=np.uint8).reshape(2, 4))
    assert e.value.code == ERR_INVALID_QUERY


def test_scalar_entry_points_match_oracle(oracle):
    text, st, hd = synth.make_text(5000, 0, 9, 3, 0.05)
    ix = gpu_index(text, 0, 4, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 4, 0, st, hd)
    rng = np.random.default_rng(1)
    for row in rng.integers(0, len(text), 60).tolist() + [0, len(text) - 1, ix.sentinel_row()]:
        assert ix.backstep(int(row)) == oi.backstep(int(row))
    for ch, idx in (("A", 1), ("C", 2), ("G", 3), ("N", 4), ("T", 5), ("t", 5), ("U", 5), ("R", 4)):
        r0 = ix.initial_search_range(ch)
        assert (r0.start_ptr, r0.end_ptr) == oi.initial_search_range(idx)
        for ch2, idx2 in (("A", 1), ("C", 2), ("G", 3), ("N", 4), ("T", 5)):
            r1 = ix.update_range_with_symbol(r0, ch2)
            assert (r1.start_ptr, r1.end_ptr) == oi.update_range_with_symbol(r0.start_ptr, r0.end_ptr, idx2)
    aa_text, st, hd = synth.make_text(4000, 1, 3, 4)
    ax = gpu_index(aa_text, 1, 8, 0, st, hd)
    ao = oracle.OracleIndex.from_text(aa_text, 1, 8, 0, st, hd)
    for row in rng.integers(0, len(aa_text), 60).tolist():
        assert ax.backstep(int(row)) == ao.backstep(int(row))
    letters = "ACDEFGHIKLMNPQRSTVWXY"  # symbol indices 1..21 (src/alphabet.rs:280-303)
    for i, ch in enumerate(letters, start=1):
        r0 = ax.initial_search_range(ch)
        assert (r0.start_ptr, r0.end_ptr) == ao.initial_search_range(i)
        for j, ch2 in enumerate(letters, start=1):  # update_range_with_symbol over all 21 x 21 symbol pairs, and one step further
            r1 = ax.update_range_with_symbol(r0, ch2)
            assert (r1.start_ptr, r1.end_ptr) == ao.update_range_with_symbol(r0.start_ptr, r0.end_ptr, j), (ch, ch2)
            r2 = ax.update_range_with_symbol(r1, letters[(i + j) % 21])  # (also from empty ranges: emptiness is sticky, src/fm_index.rs:559-582)
            assert (r2.start_ptr, r2.end_ptr) == ao.update_range_with_symbol(r1.start_ptr, r1.end_ptr, (i + j) % 21 + 1), (ch, ch2)


def test_undefined_queries_are_rejected():
    """documented deviation: where the reference panics / is UB the call returns INVALID_QUERY"""
    text, st, hd = synth.make_text(2000, 0, 2)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    for bad in ([b""], [b"ACGT", b"AC$T"], [b"#"], [b"ACGT", b"AC\xc3\xa9"]):
        with pytest.raises(AwryError) as e:
            ix.parallel_count(bad)
        assert e.value.code == ERR_INVALID_QUERY
        with pytest.raises(AwryError) as e:
            ix.parallel_locate(bad)
        assert e.value.code == ERR_INVALID_QUERY
    assert ix.parallel_count([]).tolist() == []
    assert ix.parallel_locate([]) == []
    assert ix.count_string("A" * 3000) == 0  # longer than the text


@pytest.mark.parametrize("alphabet,kmer_len", [(0, 0), (0, 4), (1, 0), (1, 3)])
def test_save_is_byte_identical_to_reference_format(oracle, tmp_path, alphabet, kmer_len):
    """.awry v1 incl. the reference's partially populated k-mer table (src/kmer_lookup_table.rs:121-167)"""
    text, st, hd = synth.make_text(20000, alphabet, 13, 6, 0.04)
    ix = gpu_index(text, alphabet, 8, kmer_len, st, hd)
    oi = oracle.OracleIndex.from_text(text, alphabet, 8, kmer_len, st, hd)
    a, b = str(tmp_path / "a.awry"), str(tmp_path / "b.awry")
    ix.save(a)
    oi.save(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    again = FmIndex.load(a).set_devices([0])
    q = synth.sampled_queries(text, 200, 9, 3, False, alphabet)
    qb, qo = synth.fixed_to_csr(q)
    assert np.array_equal(again.parallel_count_csr(qb, qo), ix.parallel_count_csr(qb, qo))
    c = str(tmp_path / "c.awry")
    again.save(c)  # a loaded index re-saves byte-identically (save_load_equality_test, src/fm_index.rs:1046)
    assert open(c, "rb").read() == open(a, "rb").read()


def test_query_sharding_over_replicas_keeps_input_order(oracle):
    """two replicas (both on GPU 0 here): contiguous shards, concatenated in input order"""
    text, st, hd = synth.make_text(120000, 0, 4, 2, 0.03)
    one = gpu_index(text, 0, 8, 0, st, hd, devices=(0,))
    two = gpu_index(text, 0, 8, 0, st, hd, devices=(0, 0, 0))
    assert two.num_devices() == 3
    q2d = np.concatenate([synth.random_queries(2001, 14, 0, 1), synth.sampled_queries(text, 2000, 14, 2)])
    qb, qo = synth.fixed_to_csr(q2d)
    assert np.array_equal(one.parallel_count_csr(qb, qo), two.parallel_count_csr(qb, qo))
    a, b = one.parallel_locate_csr(qb, qo), two.parallel_locate_csr(qb, qo)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # unequal lengths, a few reads with N, amino-free generic batches and packed words shard the same way
    rng = np.random.default_rng(3)
    lens = rng.integers(5, 60, size=9001)
    ro = np.zeros(len(lens) + 1, dtype=np.uint64)
    ro[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - 70, size=len(lens))
    rb = text[np.repeat(starts, lens) + (np.arange(int(ro[-1])) - np.repeat(ro[:-1].astype(np.int64), lens))].copy()
    rb[rb == ord("$")] = ord("A")
    assert np.array_equal(one.parallel_count_csr(rb, ro), two.parallel_count_csr(rb, ro))
    a, b = one.parallel_locate_csr(rb, ro), two.parallel_locate_csr(rb, ro)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    code = np.searchsorted(synth.NT, q2d).astype(np.uint64)
    words = np.zeros(len(q2d), dtype=np.uint64)
    for j in range(q2d.shape[1]):
        words |= code[:, j] << np.uint64(2 * j)
    assert np.array_equal(one.parallel_count_packed(words, 14), two.parallel_count_packed(words, 14))
    assert np.array_equal(one.parallel_count_packed(words, 14), one.parallel_count_csr(qb, qo))


def test_medium_scale_properties():
    """size-independent properties at 2e7 symbols (the oracle is not needed): every located position really
    holds the query; count == number of locations; sampled queries are present; locations are distinct."""
    n, L = 20_000_000, 31
    text, st, hd = synth.make_text(n, 0, 77, 1, 0.07)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    q2d = np.concatenate([synth.sampled_queries(text, 20000, L, 5), synth.random_queries(20000, L, 0, 6)])
    qb, qo = synth.fixed_to_csr(q2d)
    counts = ix.parallel_count_csr(qb, qo)
    assert (counts[:20000] >= 1).all()
    assert np.array_equal(counts, ix.count_kmers_nt2(q2d, True)) and np.array_equal(counts, ix.count_kmers_nt2(q2d, False))
    off, gpos, pos = ix.parallel_locate_csr(qb, qo)
    assert np.array_equal(np.diff(off), counts)
    qi = np.repeat(np.arange(len(q2d)), counts.astype(np.int64))
    win = text[gpos.astype(np.int64)[:, None] + np.arange(L)[None, :]]
    assert np.array_equal(win, q2d[qi])
    assert len(np.unique(gpos + qi.astype(np.uint64) * np.uint64(n + 1))) == len(gpos)
    assert (pos[:, 0] == 0).all() and np.array_equal(pos[:, 1], gpos)


@pytest.mark.parametrize("alphabet,n,recs,nfrac,ratio,seed", [
    (0, 1, 1, 0.0, 8, 1), (0, 2, 1, 0.0, 8, 2), (0, 5, 1, 0.0, 1, 3), (0, 300, 2, 0.0, 8, 4), (0, 70000, 3, 0.2, 8, 5),
    (1, 50000, 30, 0.02, 5, 6), (0, 1_000_000, 4, 0.07, 8, 7), (1, 400_000, 900, 0.0, 8, 8), (0, 3_000_001, 1, 0.3, 16, 9)])
def test_gpu_construction_is_bit_identical_to_host(alphabet, n, recs, nfrac, ratio, seed):
    """FmIndex::new on the GPU (prefix-doubling SA + streaming pack kernels, sa_builder.hip) == host SA-IS + pack_index"""
    text, st, hd = synth.make_text(n, alphabet, seed, recs, nfrac)
    host = FmIndex.from_text(text, alphabet, ratio, 0, st, hd, build_device=-1)
    gpu = FmIndex.from_text(text, alphabet, ratio, 0, st, hd, build_device=0)
    assert gpu.bwt_len() == host.bwt_len() and gpu.sentinel_row() == host.sentinel_row()
    assert np.array_equal(gpu.prefix_sums(), host.prefix_sums())
    assert np.array_equal(gpu.sa_words(), host.sa_words())
    assert np.array_equal(gpu.device_block_words(), host.device_block_words())
    assert gpu.sequences() == host.sequences()


def test_gpu_construction_degenerate_texts():
    for t in (b"$", b"A$", b"A" * 5000 + b"$", b"N" * 70000 + b"$", b"AC" * 40000 + b"$", b"ACGTN" * 9000 + b"$",
              (b"GATTACA" * 3000 + b"N") * 7 + b"$", b"T" * 300 + b"G" * 300 + b"C" * 300 + b"A" * 300 + b"$"):
        host = FmIndex.from_text(t, 0, 4, 0, build_device=-1)
        gpu = FmIndex.from_text(t, 0, 4, 0, build_device=0)
        assert np.array_equal(gpu.device_block_words(), host.device_block_words()), t[:20]
        assert np.array_equal(gpu.sa_words(), host.sa_words()) and gpu.sentinel_row() == host.sentinel_row()
        assert np.array_equal(gpu.prefix_sums(), host.prefix_sums())


def test_locate_tiles_with_sparse_and_heavy_queries(oracle):
    """the locate tile kernel stages the offsets of up to 1024 queries per tile of 1024 hits and derives the next tile's first
    query from them: batches where most queries have no hit (a tile spans tens of thousands of queries), where single queries
    own tens of tiles, and mixtures, against the oracle -- through the host path (chunks) and the device-resident stages"""
    text, st, hd = synth.make_text(300000, 0, 61, 3, 0.02)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(8)
    sparse = [bytes(x) for x in synth.random_queries(120000, 12, 0, 3)]          # ~2 % present
    heavy = [b"A", b"CG", b"T", b"GAT"]                                            # 5 000 .. 75 000 hits each
    mixed = list(sparse[:50000])
    for j, pos in enumerate(rng.integers(0, len(mixed), size=12)):
        mixed.insert(int(pos), heavy[j % len(heavy)])
    dense = [bytes(x) for x in synth.sampled_queries(text, 30000, 20, 4)]          # one hit or a few per query
    for qs in (sparse, mixed, heavy * 3, dense + sparse[:30000] + dense):
        qb, qo = awry_pack(qs)
        ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
        for dens in (0, 1):
            ix.set_locate_sa_ratio(dens)
            off, gpos, pos = ix.parallel_locate_csr(qb, qo)
            assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos), (len(qs), dens)
    ix.set_locate_sa_ratio(0)


def test_automatic_gpu_construction_falls_back_to_the_host(monkeypatch):
    """AWRY_BUILD_AUTO picks the GPU for texts of 2^20 symbols and more; when that GPU cannot build (forced here; in the field:
    its HBM is taken) the host builder produces the same index, while an explicit device request fails loudly"""
    from awry_amd.fm_index import BUILD_AUTO
    text, st, hd = synth.make_text((1 << 20) + 5000, 0, 71, 2, 0.01)
    gpu = FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0)
    monkeypatch.setenv("AWRY_DEBUG_FAIL_GPU_BUILD", "1")
    with pytest.raises(AwryError) as e:
        FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0)
    assert "GPU index construction" in str(e.value)
    auto = FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=BUILD_AUTO)
    monkeypatch.delenv("AWRY_DEBUG_FAIL_GPU_BUILD")
    assert np.array_equal(auto.device_block_words(), gpu.device_block_words()) and np.array_equal(auto.sa_words(), gpu.sa_words())
    assert np.array_equal(auto.prefix_sums(), gpu.prefix_sums()) and auto.sentinel_row() == gpu.sentinel_row()
    auto.set_devices([0])
    q = synth.sampled_queries(text, 2000, 25, 3)
    assert (auto.parallel_count_csr(*synth.fixed_to_csr(q)) >= 1).all()


@pytest.mark.parametrize("L", [6, 7, 9, 10])
def test_kmers_shorter_than_the_seed_table(oracle, L):
    """k-mers shorter than the seed table's k get a complete table of their own length on first use (a "rung": the entry is
    the answer); counts and locations against the oracle, device-resident and through the host path, and equal to plain
    backward search (no table); a small batch (no rung) gives the same"""
    import torch
    text, st, hd = synth.make_text(1_000_000, 0, 33, 3, 0.02)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    assert ix.seed_kmer_len() > L and "table of its own" in ix.count_schedule(L)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    q2d = np.concatenate([synth.sampled_queries(text, 15000, L, L), synth.random_queries(15000, L, 0, L + 1)])
    qb, qo = synth.fixed_to_csr(q2d)
    want, _ = oi.parallel_count(qb, qo, 4)
    assert np.array_equal(ix.count_kmers_nt2(q2d, True), want)       # device-resident, rung table
    assert np.array_equal(ix.count_kmers_nt2(q2d, False), want)      # no table at all
    assert np.array_equal(ix.count_kmers_nt2(q2d[:100], True), want[:100])
    assert np.array_equal(ix.parallel_count_csr(qb, qo), want)       # host path
    sub = slice(0, 300)
    qb2, qo2 = synth.fixed_to_csr(q2d[sub])
    ooff, ogpos, opos, _ = oi.parallel_locate(qb2, qo2, 4)
    off, gpos, pos = ix.parallel_locate_csr(qb2, qo2)
    assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos)


@pytest.mark.parametrize("alphabet,records", [(0, 6), (1, 6), (0, 7000), (1, 3000)])
def test_device_resident_locate_pipeline_from_ascii(oracle, alphabet, records):
    """awry_dev_count_ascii_for_locate -> awry_dev_scan_counts -> awry_dev_locate, all on caller-owned device buffers:
    the oracle's offsets, text positions and (record, offset) pairs, with the accelerators on (verified positions in the
    locate words) and off (row intervals: hits walk to a sample and are localised afterwards); with a handful of records
    (their starts sit in LDS) and with thousands (contigs, proteins: a bucket table narrows the search, DevIndex::seq_bucket)"""
    import torch
    text, st, hd = synth.make_text(400000, alphabet, 17, records, 0.01 if alphabet == 0 else 0.0)
    ix = gpu_index(text, alphabet, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, alphabet, 8, 0, st, hd)
    rng = np.random.default_rng(3)
    nq = 20000
    lens = rng.integers(9, 41, size=nq)
    qo = np.zeros(nq + 1, dtype=np.uint64); qo[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - 45, size=nq)
    idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
    qb = text[idx].copy()
    qb[qb == ord("$")] = ord("A")
    rmask = np.repeat(rng.random(nq) < 0.3, lens)
    letters = synth.NT if alphabet == 0 else synth.AA
    qb[rmask] = letters[rng.integers(0, len(letters), size=int(rmask.sum()))]
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
    d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    for verify in (2, -1):
        ix.set_verify(verify)
        d_c = torch.zeros(nq, dtype=torch.int64, device=dev); d_w = torch.zeros(2 * nq, dtype=torch.int64, device=dev)
        d_s = torch.full((nq,), 9, dtype=torch.uint8, device=dev)
        d_ho = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
        d_sc = torch.zeros(ix.dev_scan_scratch_bytes(nq) // 8 + 8, dtype=torch.int64, device=dev)
        ix.dev_count_ascii_for_locate(d_q.data_ptr(), d_off.data_ptr(), nq, d_c.data_ptr(), d_w.data_ptr(), d_s.data_ptr(), stream, 0)
        ix.dev_scan_counts(d_c.data_ptr(), nq, d_ho.data_ptr(), d_sc.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        assert int(d_s.max()) == 0
        off = d_ho.cpu().numpy().astype(np.uint64)
        assert np.array_equal(off, ooff)
        total = int(off[-1])
        d_g = torch.zeros(total, dtype=torch.int64, device=dev); d_p = torch.zeros(2 * total, dtype=torch.int64, device=dev)
        ix.dev_locate(d_w.data_ptr(), d_ho.data_ptr(), nq, total, d_g.data_ptr(), d_p.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        assert np.array_equal(d_g.cpu().numpy().astype(np.uint64), ogpos), verify
        assert np.array_equal(d_p.cpu().numpy().astype(np.uint64).reshape(-1, 2), opos), verify
        got = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(got[0], ooff) and np.array_equal(got[1], ogpos) and np.array_equal(got[2], opos), verify


def test_genome_like_text_construction_and_repeats(oracle):
    """a chromosome-shaped text (megabase N gaps, satellite array, exact tandem array, segmental duplications,
    synth.genome_like_text): the GPU construction needs ~15 doubling rounds and stays bit-identical to host SA-IS, and
    queries from the repeats (wide ranges, thousands of hits) count and locate as in the oracle"""
    n = 3_000_000
    text, reg = synth.genome_like_text(n)
    host = FmIndex.from_text(text, 0, 8, 0, [0], ["chrS"], build_device=-1)
    gpu = FmIndex.from_text(text, 0, 8, 0, [0], ["chrS"], build_device=0)
    assert np.array_equal(gpu.device_block_words(), host.device_block_words()) and np.array_equal(gpu.sa_words(), host.sa_words())
    assert np.array_equal(gpu.prefix_sums(), host.prefix_sums()) and gpu.sentinel_row() == host.sentinel_row()
    gpu.set_devices([0])
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, [0], ["chrS"])
    for starts, L in (([reg["dup0"] + 1000 * i for i in range(40)], 101), ([reg["sat0"] + 171 * 7 * i + 3 for i in range(40)], 60),
                      ([reg["ex0"] + 5 * i for i in range(12)], 74), ([n // 2 + 5], 40)):
        q2d = np.stack([text[s: s + L] for s in starts])
        qb, qo = synth.fixed_to_csr(q2d)
        ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
        off, gpos, pos = gpu.parallel_locate_csr(qb, qo)
        assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos), L
        assert np.array_equal(gpu.parallel_count_csr(qb, qo), np.diff(ooff))
    assert int(np.diff(ooff)[0]) == (reg["gap"] - 39) + 2 * (reg["tel"] - 39)   # the run of N's, by hand


@pytest.mark.parametrize("L,nq", [(1000, 5000), (4096, 4200), (4097, 4200), (20000, 300)])
def test_very_long_queries(oracle, L, nq):
    """queries as long as contigs / long reads: 4096 letters is the longest packed query of the host path (128 words), 4097
    and beyond go through the generic kernel; present (drawn from the text), one letter changed, and random queries"""
    text, st, hd = synth.make_text(300000, 0, 53, 3, 0.0)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(L)
    own = np.concatenate([np.array(st[1:]) - 1, [len(text) - 1]])          # delimiters and '$'
    starts = rng.integers(0, len(text) - L - 1, size=nq)
    starts = starts[np.searchsorted(own, starts) == np.searchsorted(own, starts + L)]   # windows inside one record
    q2d = text[starts[:, None] + np.arange(L)[None, :]]
    k = len(q2d) // 3
    q2d[np.arange(k), rng.integers(0, L, size=k)] = ord("A")               # one letter set to A: absent three times in four
    q2d[k: k + 5] = synth.random_queries(5, L, 0, 9)
    qb, qo = synth.fixed_to_csr(q2d)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    assert (np.diff(ooff)[k + 5:] >= 1).all() and (np.diff(ooff) == 0).any()
    assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(ooff))
    off, gpos, pos = ix.parallel_locate_csr(qb, qo)
    assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos)
    # unequal lengths around the same size
    lens = rng.integers(max(1, L - 70), L + 1, size=len(q2d))
    qb2 = np.concatenate([q2d[i, L - lens[i]:] for i in range(len(q2d))])
    qo2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    oc2, _ = oi.parallel_count(qb2, qo2, 4)
    assert np.array_equal(ix.parallel_count_csr(qb2, qo2), oc2)


@pytest.mark.parametrize("L", [33, 50, 64, 65, 101, 150])
def test_long_packed_reads_count_and_locate(oracle, L):
    """multi-word packed reads through the quad kernel + tile locate, against the oracle (same order)"""
    text, st, hd = synth.make_text(300000, 0, 41, 4, 0.05)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    q2d = np.concatenate([synth.sampled_queries(text, 600, L, L), synth.random_queries(200, L, 0, L + 1)])
    q2d[5] = q2d[4]  # duplicates are independent queries
    qb, qo = synth.fixed_to_csr(q2d)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    for k in (-1, 0, 5):
        ix.set_seed_kmer_len(k)
        off, gpos, pos = ix.locate_reads_nt2(q2d, True)
        assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos), (L, k)


@pytest.mark.parametrize("alphabet", [0, 1])
def test_dense_device_sa_does_not_change_locations(oracle, tmp_path, alphabet):
    """awry_set_locate_sa_ratio is a performance knob: every density gives the oracle's locations, also on a
    loaded index (whose dense SA is recovered by LF walks to the file's samples)"""
    text, st, hd = synth.make_text(120000, alphabet, 19, 5, 0.05)
    ix = gpu_index(text, alphabet, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, alphabet, 8, 0, st, hd)
    qb, qo = synth.fixed_to_csr(synth.sampled_queries(text, 800, 7 if alphabet else 12, 3, False, alphabet))
    qb2, qo2 = synth.fixed_to_csr(synth.random_queries(300, 4 if alphabet else 7, alphabet, 4))
    want = [oi.parallel_locate(qb, qo, 4)[:3], oi.parallel_locate(qb2, qo2, 4)[:3]]
    p = str(tmp_path / "d.awry")
    ix.save(p)
    loaded = FmIndex.load(p).set_devices([0])
    for idx in (ix, loaded):
        idx.set_verify(-1)          # the default policy keeps the ratio-1 dense SA resident; start from the file's samples
        idx.set_locate_sa_ratio(0)
        assert idx.locate_sa_ratio() == 8
        for r in (1, 2, 3, 8, 16, 0):
            idx.set_locate_sa_ratio(r)
            for (b, o), w in zip(((qb, qo), (qb2, qo2)), want):
                got = idx.parallel_locate_csr(b, o)
                assert all(np.array_equal(x, y) for x, y in zip(got, w)), r


@pytest.mark.parametrize("L", [12, 31, 32, 40, 101])
def test_host_batch_fast_path_equals_generic(oracle, L, monkeypatch):
    """parallel_count on fixed-length batches takes the pipelined packed path; chunks with N / IUPAC / lower-case / U
    fall back per chunk to the generic kernel.  Both must give the oracle's counts."""
    text, st, hd = synth.make_text(400000, 0, 61, 3, 0.05)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    clean = np.concatenate([synth.sampled_queries(text, 3000, L, L), synth.random_queries(3000, L, 0, L + 1)])
    dirty = clean.copy()
    dirty[7, 3] = ord("N"); dirty[100, 0] = ord("u"); dirty[2999, L - 1] = ord("R"); dirty[11] = np.frombuffer(bytes(dirty[11]).lower(), np.uint8)
    for q2d in (clean, dirty):
        qb, qo = synth.fixed_to_csr(q2d)
        want, _ = oi.parallel_count(qb, qo, 4)
        assert np.array_equal(ix.parallel_count_csr(qb, qo), want)
    bad = clean.copy()
    bad[5, 2] = ord("$")
    with pytest.raises(AwryError) as e:
        ix.parallel_count_csr(*synth.fixed_to_csr(bad))
    assert e.value.code == ERR_INVALID_QUERY


@pytest.mark.parametrize("alphabet", [0, 1])
def test_generic_host_pipeline(oracle, alphabet):
    """batches the packed kernels do not take -- amino queries, nucleotide batches of very unequal lengths -- go through
    the generic kernel in the same pinned two-lane pipeline: the oracle's counts, undefined queries named by their index"""
    text, st, hd = synth.make_text(200000, alphabet, 17 + alphabet, 3, 0.02 if alphabet == 0 else 0.0)
    ix = gpu_index(text, alphabet, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, alphabet, 8, 0, st, hd)
    rng = np.random.default_rng(alphabet)
    nq = 30000
    lens = rng.integers(1, 24, size=nq)
    lens[::500] = rng.integers(600, 900, size=len(lens[::500]))  # a few long ones: no packed plan for this batch
    qo = np.zeros(nq + 1, dtype=np.uint64)
    qo[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - 1000, size=nq)
    idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
    qb = text[idx].copy()
    letters = synth.NT if alphabet == 0 else synth.AA
    rmask = np.repeat(rng.random(nq) < 0.4, lens)
    qb[rmask] = letters[rng.integers(0, len(letters), size=int(rmask.sum()))]
    qb[qb == ord("$")] = letters[0]
    low = np.repeat(rng.random(nq) < 0.1, lens)
    qb[low] = np.frombuffer(bytes(qb[low]).lower(), dtype=np.uint8)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(ooff))
    for verify in (2, -1):
        ix.set_verify(verify)
        off, g, p = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(off, ooff) and np.array_equal(g, ogpos) and np.array_equal(p, opos), verify
    bad = qb.copy()
    bad[int(qo[12345])] = ord("#")
    for fn in (ix.parallel_count_csr, ix.parallel_locate_csr):
        with pytest.raises(AwryError) as e:
            fn(bad, qo)
        assert e.value.code == ERR_INVALID_QUERY and "query 12345" in str(e.value)


def test_packed_kmer_host_entry_point(oracle):
    """awry_count_packed_kmers: k-mers the caller already holds 2 bits per letter -- the ASCII entry point's counts"""
    text, st, hd = synth.make_text(300000, 0, 4, 2, 0.02)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    for L in (31, 32, 12, 1):
        q2d = np.concatenate([synth.sampled_queries(text, 5000, L, L), synth.random_queries(5000, L, 0, L + 5)])
        code = np.searchsorted(synth.NT, q2d).astype(np.uint64)
        words = np.zeros(len(q2d), dtype=np.uint64)
        for j in range(L):
            words |= code[:, j] << np.uint64(2 * j)
        want = ix.parallel_count_csr(*synth.fixed_to_csr(q2d))
        assert np.array_equal(ix.parallel_count_packed(words, L), want), L
    with pytest.raises(AwryError):
        ix.parallel_count_packed(np.zeros(4, np.uint64), 33)


def test_host_pipelines_over_many_chunks(oracle):
    """batches larger than one pipeline chunk (count: 4 M queries, locate: 1 M reads) alternate between the two stream
    lanes; a sprinkling of reads with N goes through the per-read fallback in every chunk.  Checked against the oracle
    on a random sample of the queries and, for the rest, through count = number of locations"""
    text, st, hd = synth.make_text(600000, 0, 88, 2, 0.02)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(8)
    for ragged in (False, True):
        nq = 4_500_000
        lens = rng.integers(18, 31, size=nq) if ragged else np.full(nq, 24)
        qo = np.zeros(nq + 1, dtype=np.uint64)
        qo[1:] = np.cumsum(lens)
        starts = rng.integers(0, len(text) - 40, size=nq)
        idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
        qb = text[idx].copy()
        rnd = rng.random(nq) < 0.3  # 30 % random letters instead of a window of the text
        rmask = np.repeat(rnd, lens)
        qb[rmask] = synth.NT[rng.integers(0, 4, size=int(rmask.sum()))]
        qb[qb == ord("$")] = ord("A")
        counts = ix.parallel_count_csr(qb, qo)
        off, g, p = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(np.diff(off), counts) and len(g) == int(off[-1]) == len(p)
        sample = np.sort(rng.choice(nq, size=20000, replace=False))
        sb = np.concatenate([qb[int(qo[i]):int(qo[i + 1])] for i in sample])
        so = np.zeros(len(sample) + 1, dtype=np.uint64)
        so[1:] = np.cumsum(lens[sample])
        ooff, ogpos, opos, _ = oi.parallel_locate(sb, so, 4)
        assert np.array_equal(np.diff(ooff), counts[sample])
        got_g = np.concatenate([g[int(off[i]):int(off[i + 1])] for i in sample])
        got_p = np.concatenate([p[int(off[i]):int(off[i + 1])] for i in sample])
        assert np.array_equal(got_g, ogpos) and np.array_equal(got_p, opos)
        assert (counts[~rnd] >= 1).all() or (qb == ord("N")).any()  # windows of the text are found (N windows match too)


@pytest.mark.parametrize("n,recs", [(248_956_422, 1), (3_100_000_000, 25)], ids=["chr1_scale", "grch38_scale"])
def test_full_scale_properties(n, recs):
    """BASELINE.json's full sizes (index built on the GPU in seconds): size-independent properties.
    * every k-mer sampled from the text is found, at its own position, and every located window equals the query;
    * count == number of locations; the seeded, unseeded and generic kernels agree;
    * letter totals in the prefix sums equal the text's histogram (checksum of the BWT)."""
    text, st, hd = synth.make_text(n, 0, 0xA5A50000 + 2, recs, 0.05)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    hist = np.bincount(text, minlength=256)
    ps = ix.prefix_sums()
    assert ps[-1] == n + 1
    assert [int(ps[i + 1] - ps[i]) for i in range(6)] == [1, hist[65], hist[67], hist[71], hist[78], hist[84]]
    L, m = 31, 20000
    rng = np.random.default_rng(3)
    p = rng.integers(0, n - L, size=4 * m)
    win = text[p[:, None] + np.arange(L)[None, :]]
    ok = ~(win == ord("N")).any(axis=1)
    p, present = p[ok][:m], win[ok][:m]
    q2d = np.concatenate([present, synth.random_queries(m, L, 0, 6)])
    qb, qo = synth.fixed_to_csr(q2d)
    seeded = ix.count_kmers_nt2(q2d, True)
    assert (seeded[:len(present)] >= 1).all()
    assert np.array_equal(seeded, ix.count_kmers_nt2(q2d, False))
    d_q, d_o = ix.dev_upload(qb), ix.dev_upload(qo)
    d_c = ix.dev_malloc(8 * len(q2d))
    ix.dev_count_ascii(d_q, d_o, len(q2d), d_c)
    ix.dev_synchronize()
    assert np.array_equal(seeded, ix.dev_download(d_c, (len(q2d),), np.uint64))
    for ptr in (d_q, d_o, d_c):
        ix.dev_free(ptr)
    assert np.array_equal(seeded, ix.parallel_count_csr(qb, qo))  # host boundary (packed fast path)
    for ratio in (0, 4):
        ix.set_locate_sa_ratio(ratio)
        off, gpos, pos = ix.locate_reads_nt2(q2d)
        assert np.array_equal(np.diff(off), seeded)
        qi = np.repeat(np.arange(len(q2d)), seeded.astype(np.int64))
        assert np.array_equal(text[gpos.astype(np.int64)[:, None] + np.arange(L)[None, :]], q2d[qi])
        first = off[:len(present)].astype(np.int64)
        own = np.array([p[i] in gpos[off[i]:off[i + 1]] for i in range(0, len(present), 97)])
        assert own.all()
        starts = np.array(st, dtype=np.uint64)
        si = np.searchsorted(starts, gpos, side="right") - 1
        assert np.array_equal(pos[:, 0], si.astype(np.uint64)) and np.array_equal(pos[:, 1], gpos - starts[si])


def repeat_text(seed):
    """nucleotide text with repeated segments (so that several suffixes share long prefixes), N runs and 3 records"""
    rng = np.random.default_rng(seed)
    base = synth.NT[rng.integers(0, 4, size=60000, dtype=np.uint8)]
    parts = [base, base[1000:21000], synth.NT[rng.integers(0, 4, size=5000, dtype=np.uint8)], base[500:30500], base[1000:9000],
             np.full(300, ord("N"), np.uint8), base[40000:52000], base[1000:21000], base[40000:52000]]
    body = np.concatenate(parts)
    body[77777] = ord("N"); body[123456] = ord("N")
    text = np.concatenate([body, np.frombuffer(b"$", np.uint8)])
    return text, [0, 77778, 123457], ["r0", "r1", "r2"]


@pytest.mark.parametrize("L", [20, 40, 101, 150, 300])
@pytest.mark.parametrize("after", [0, 2])
def test_seed_and_verify_does_not_change_results(oracle, L, after):
    """awry_set_verify: text comparison instead of LF steps once a range is small -- same counts, same locations, same
    order as the oracle, on a text with repeats (1..8 candidates per read), N runs and reads that start at text position 0"""
    text, st, hd = repeat_text(5)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(L)
    starts = np.concatenate([rng.integers(0, len(text) - L - 1, size=1500), [0, 1, 2, 1000, 40000, 77778, 123457], np.arange(60000 - L, 60010)])
    reads = text[starts[:, None] + np.arange(L)[None, :]]
    reads = reads[~((reads == ord("N")) | (reads == ord("$"))).any(axis=1)]
    mut = reads[:300].copy()  # single mismatches at random places: mostly absent, sometimes another repeat copy
    mut[np.arange(len(mut)), rng.integers(0, L, size=len(mut))] = synth.NT[rng.integers(0, 4, size=len(mut))]
    q2d = np.concatenate([reads, mut, synth.random_queries(200, L, 0, 3)])
    qb, qo = synth.fixed_to_csr(q2d)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    assert int(np.diff(ooff).max()) >= 4  # the text really has multi-candidate reads
    for k in (-1, 6):
        ix.set_seed_kmer_len(k)
        ix.set_verify(-1)
        base = ix.locate_reads_nt2(q2d)
        ix.set_verify(after)
        assert ix.verify_enabled() and ix.locate_sa_ratio() == 1
        got = ix.locate_reads_nt2(q2d)
        for x, y, z in zip(got, base, (ooff, ogpos, opos)):
            assert np.array_equal(x, y) and np.array_equal(x, z), (L, after, k)
        assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(ooff))  # host fast path (long kernel, verify on)
    ix.set_locate_sa_ratio(4)  # leaving ratio 1 switches verify off again
    assert not ix.verify_enabled()
    assert all(np.array_equal(x, y) for x, y in zip(ix.locate_reads_nt2(q2d), (ooff, ogpos, opos)))


def test_seed_and_verify_in_the_kmer_kernel(oracle):
    """the packed k-mer kernel (L <= 32) with awry_set_verify on: same counts as the oracle for every length / seed k"""
    text, st, hd = repeat_text(9)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    for L in (9, 16, 24, 31, 32):
        rng = np.random.default_rng(L)
        starts = np.concatenate([rng.integers(0, len(text) - L - 1, size=2000), [0, 1, 2, 3]])
        reads = text[starts[:, None] + np.arange(L)[None, :]]
        reads = reads[~((reads == ord("N")) | (reads == ord("$"))).any(axis=1)]
        q2d = np.concatenate([reads, synth.random_queries(1001, L, 0, L)])
        want, _ = oi.parallel_count(*synth.fixed_to_csr(q2d), 4)
        ix.set_verify_kmers(True)
        for after in (0, 1, 3):
            ix.set_verify(after)
            for k in (-1, 3, 8):
                ix.set_seed_kmer_len(k)
                assert np.array_equal(ix.count_kmers_nt2(q2d, True), want), (L, after, k)
            assert np.array_equal(ix.count_kmers_nt2(q2d, False), want), (L, after)
        ix.set_verify(-1)


def test_per_lane_verify_of_the_two_phase_schedule(oracle):
    """phase 1 of the two-phase schedule settles probed singletons by itself (SA read + text window per lane, queued in
    LDS until two per lane are pending): exact counts for present / absent / near-miss k-mers, candidates too close to
    the text's beginning, windows crossing N runs and record delimiters, and batch sizes that leave partial queues"""
    from awry_amd import _lib
    L_ = _lib.load_library()
    text, st, hd = synth.make_text(300000, 0, 4711, 3, 0.03)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    ix.set_verify(0)
    assert ix.verify_enabled()
    rng = np.random.default_rng(12)
    try:
        L_.awry_debug_set_count_kernel(3)
        for L, k in ((31, 10), (32, 9), (20, 10), (13, 10), (12, 10), (31, 11)):
            i0 = L - k
            pres = synth.sampled_queries(text, 3000, L, L)
            near = pres.copy()  # one substitution left of the seed window: the singleton survives the probe, the text decides
            col = rng.integers(0, max(1, i0), size=len(near))
            near[np.arange(len(near)), col] = synth.NT[(np.searchsorted(synth.NT[:4], near[np.arange(len(near)), col]) + 1) % 4]
            # the seed part taken from the first few text positions, a random left part: candidate with vp < L - k
            head = np.stack([np.concatenate([synth.NT[rng.integers(0, 4, size=i0 - 1)], text[p - 1:p + k]]) for p in range(1, 12)])
            # windows that cross N runs / delimiters: reads starting right after every non-ACGT position
            bad = np.flatnonzero(~np.isin(text[:-L - 1], synth.NT[:4]))[:400]
            edge = text[(bad + 1)[:, None] + np.arange(L)[None, :]]
            edge = edge[np.isin(edge, synth.NT[:4]).all(axis=1)]
            q2d = np.concatenate([pres, near, head, edge, synth.random_queries(2001, L, 0, L)])
            q2d = q2d[np.isin(q2d, synth.NT).all(axis=1)]
            q2d = q2d[rng.permutation(len(q2d))]
            want, _ = oi.parallel_count(*synth.fixed_to_csr(q2d), 4)
            ix.set_seed_kmer_len(k)
            for nq in (len(q2d), 1, 63, 64, 129, 1000):
                assert np.array_equal(ix.count_kmers_nt2(q2d[:nq], True), want[:nq]), (L, k, nq)
            got, census = ix.count_kmers_nt2(q2d, True, tally=True)
            assert np.array_equal(got, want) and int(census[0]) == len(q2d)
            # (how many of them the text settled depends on how many letters the seed entries themselves hold:
            # 14 + the spare position bits of this small text -- so the census is reported, not asserted)
            assert int(census[4]) <= len(q2d) * 4, (L, k, census)
        assert "probe" in ix.count_schedule(31)
    finally:
        L_.awry_debug_set_count_kernel(-1)
        ix.set_verify(-1)


@pytest.mark.parametrize("L", [14, 40, 101, 150, 257, 600])
def test_two_phase_reads_schedule_matches_single_kernel_and_oracle(oracle, L):
    """reads of any length: the per-lane probe + verify pass followed by the quad kernel on the undecided reads gives the
    counts, locations and order of the single quad kernel and of the oracle (mostly-unique seeds, so the per-lane path
    carries the batch); L = 600 is beyond the two-phase limit and L - k < 3 below it: both fall back by themselves"""
    from awry_amd import _lib
    L_ = _lib.load_library()
    text, st, hd = synth.make_text(400000, 0, 99 + L, 2, 0.03)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(L)
    pres = synth.sampled_queries(text, 3000, L, L)
    near = pres[:1500].copy()
    col = rng.integers(0, L, size=len(near))
    near[np.arange(len(near)), col] = synth.NT[(np.searchsorted(synth.NT[:4], near[np.arange(len(near)), col]) + 1) % 4]
    k = 10 if L > 14 else 12
    head = np.stack([np.concatenate([synth.NT[rng.integers(0, 4, size=L - k - 1)], text[p - 1:p + k]]) for p in range(1, 9)])
    first = text[np.arange(0, 6)[:, None] + np.arange(L)[None, :]]  # reads that start at text positions 0..5
    q2d = np.concatenate([pres, near, head, first, synth.random_queries(1500, L, 0, L)])
    q2d = q2d[np.isin(q2d, synth.NT).all(axis=1)]
    q2d = q2d[rng.permutation(len(q2d))]
    ooff, ogpos, opos, _ = oi.parallel_locate(*synth.fixed_to_csr(q2d), 4)
    ix.set_verify(0)
    ix.set_seed_kmer_len(k)
    try:
        for nq in (len(q2d), 1, 64, 129, 777):
            cut = int(ooff[nq])
            L_.awry_debug_set_count_kernel(-1)
            two = ix.locate_reads_nt2(q2d[:nq])
            L_.awry_debug_set_count_kernel(0)
            one = ix.locate_reads_nt2(q2d[:nq])
            for x, y, z in zip(two, one, (ooff[:nq + 1], ogpos[:cut], opos[:cut])):
                assert np.array_equal(x, y) and np.array_equal(x, z), (L, nq)
    finally:
        L_.awry_debug_set_count_kernel(-1)
        ix.set_verify(-1)


@pytest.mark.parametrize("lo,hi", [(1, 40), (20, 40), (60, 160), (5, 330), (33, 35)])
def test_ragged_batches_take_the_packed_kernels(oracle, lo, hi, monkeypatch):
    """batches of unequal lengths (trimmed reads): packed on the device at a common stride, per-read lengths through
    the two-phase read kernels -- counts, locations and their order are the oracle's, for reads shorter than the seed,
    with and without the seed-and-verify accelerators, and equal to the generic path's"""
    text, st, hd = synth.make_text(300000, 0, 1000 + lo, 3, 0.03)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(lo * 1000 + hi)
    nq = 6000
    lens = rng.integers(lo, hi + 1, size=nq)
    starts = rng.integers(0, len(text) - hi - 2, size=nq)
    qs = []
    for i in range(nq):
        q = text[starts[i]:starts[i] + lens[i]].copy()
        kind = i % 4
        if kind == 1:  # random letters
            q = synth.NT[rng.integers(0, 4, size=lens[i])]
        elif kind == 2:  # one substitution
            j = int(rng.integers(0, lens[i]))
            q[j] = synth.NT[(int(np.searchsorted(synth.NT, q[j])) + 1) % 4] if q[j] in synth.NT else ord("A")
        if not np.isin(q, synth.NT).all():
            q = synth.NT[rng.integers(0, 4, size=lens[i])]
        qs.append(q)
    qb = np.concatenate(qs)
    qo = np.zeros(nq + 1, dtype=np.uint64)
    qo[1:] = np.cumsum(lens)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    for verify in (-1, 0, 2):
        ix.set_verify(verify)
        for k in (-1, 6, 12):
            ix.set_seed_kmer_len(k)
            assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(ooff)), (verify, k)
            off, g, p = ix.parallel_locate_csr(qb, qo)
            assert np.array_equal(off, ooff) and np.array_equal(g, ogpos) and np.array_equal(p, opos), (verify, k)
    # a few reads with N (some of them windows of the text that cross an N run or a record delimiter, so they have hits)
    # and lower-case letters: only those reads are redone by the generic kernels, the results are spliced back in order
    qb2 = qb.copy()
    qb2[int(qo[17])] = ord("N")
    qb2[int(qo[4000]):int(qo[4001])] = np.frombuffer(bytes(qb2[int(qo[4000]):int(qo[4001])]).lower(), dtype=np.uint8)
    npos = np.flatnonzero(text[:-hi - 2] == ord("N"))
    planted = 0
    for i in range(100, nq, 397):
        L = int(lens[i])
        p0 = int(npos[rng.integers(0, len(npos))]) - int(rng.integers(0, L))
        if p0 < 0:
            continue
        qb2[int(qo[i]):int(qo[i + 1])] = text[p0:p0 + L]
        planted += 1
    assert planted >= 5 and not (qb2 == ord("$")).any()
    ooff2, ogpos2, opos2, _ = oi.parallel_locate(qb2, qo, 4)
    for verify in (-1, 2):
        ix.set_verify(verify)
        off, g, p = ix.parallel_locate_csr(qb2, qo)
        assert np.array_equal(off, ooff2) and np.array_equal(g, ogpos2) and np.array_equal(p, opos2)
        assert np.array_equal(ix.parallel_count_csr(qb2, qo), np.diff(ooff2))
    # an undefined query among them is still reported with its index in the batch
    qb3 = qb2.copy()
    qb3[int(qo[1234])] = ord("$")
    with pytest.raises(AwryError) as e:
        ix.parallel_count_csr(qb3, qo)
    assert "query 1234" in str(e.value)
    with pytest.raises(AwryError) as e:
        ix.parallel_locate_csr(qb3, qo)
    assert "query 1234" in str(e.value)


def test_locate_walks_next_to_long_n_runs(oracle):
    """row sampling + N runs: inside a run LF moves by a constant stride (the number of runs at least that long), so a
    walk that enters a run at a row of the wrong residue meets no sampled row until the run ends -- tens of thousands of
    dependent steps for a read that starts right after a gap.  The replica keeps the SA of the BWT's N block for that
    case (DevIndex::sa_nblock); locations stay those of the oracle, which simply walks"""
    rng = np.random.default_rng(5)
    parts, cuts = [], []
    for run in (30000, 30000, 12000, 12000):  # two pairs of equal-length runs: strides 4 and 2 inside them
        parts.append(synth.NT[rng.integers(0, 4, size=20000)])
        parts.append(np.full(run, ord("N"), dtype=np.uint8))
        cuts.append(sum(len(x) for x in parts))
    parts.append(synth.NT[rng.integers(0, 4, size=20000)])
    text = np.concatenate(parts + [np.frombuffer(b"$", np.uint8)])
    ix = gpu_index(text, 0, 8, 0, [0], ["gappy"])
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, [0], ["gappy"])
    L = 28
    starts = np.concatenate([np.arange(c, c + 12) for c in cuts] + [rng.integers(0, 19000, size=300)])
    reads = text[starts[:, None] + np.arange(L)[None, :]]
    reads = reads[np.isin(reads, synth.NT).all(axis=1)]
    qb, qo = synth.fixed_to_csr(reads)
    ooff, ogpos, opos, tl = oi.parallel_locate(qb, qo, 4)
    assert tl["backsteps"] > 20000  # the reference-style walk really is long here
    ix.set_verify(-1)
    for ratio in (0, 3):  # the file's samples (ratio 8), a dense device SA that is not 1
        ix.set_locate_sa_ratio(ratio)
        off, g, p = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(off, ooff) and np.array_equal(g, ogpos) and np.array_equal(p, opos), ratio
        got = ix.locate_reads_nt2(reads)
        assert all(np.array_equal(x, y) for x, y in zip(got, (ooff, ogpos, opos))), ratio
    # queries made of N walk inside the runs themselves
    qn = [b"N" * 5, b"NNNNNNNNNNNNNNNNNNNN" + bytes(text[cuts[0]:cuts[0] + 6])]
    for q in qn:
        assert np.array_equal(ix.locate_string_raw(q)[0], oi.locate_string(q)[0])  # same order too: ascending BWT row


@pytest.mark.parametrize("L,verify", [(31, -1), (101, -1), (101, 2), (150, 0)])
def test_host_locate_fast_path_equals_oracle(oracle, L, verify):
    """parallel_locate on fixed-length read batches takes the packed kernels (and seed-and-verify when enabled);
    a batch with N / lower-case / IUPAC bytes is redone by the generic kernel.  Same CSR as the oracle either way."""
    text, st, hd = repeat_text(21)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    ix.set_verify(verify)
    rng = np.random.default_rng(L)
    starts = rng.integers(0, len(text) - L - 1, size=2500)
    reads = text[starts[:, None] + np.arange(L)[None, :]]
    reads = reads[~((reads == ord("N")) | (reads == ord("$"))).any(axis=1)]
    clean = np.concatenate([reads, synth.random_queries(300, L, 0, 2)])
    dirty = clean.copy()
    dirty[3, 5] = ord("N"); dirty[9] = np.frombuffer(bytes(dirty[9]).lower(), np.uint8); dirty[77, 0] = ord("Y")
    for q2d in (clean, dirty):
        qb, qo = synth.fixed_to_csr(q2d)
        want = oi.parallel_locate(qb, qo, 4)[:3]
        got = ix.parallel_locate_csr(qb, qo)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))


def test_seed_entry_count_saturation_falls_back():
    """a k-mer with >= 2^29 occurrences cannot be represented in a seed entry: the kernels must ignore the table
    for such queries.  Text = 6e8 x 'A' + a short tail; counts are known in closed form."""
    n_a = 600_000_000
    tail = b"CGTACGTTAGC"
    text = np.concatenate([np.full(n_a, ord("A"), np.uint8), np.frombuffer(tail + b"$", np.uint8)])
    ix = FmIndex.from_text(text, 0, 64, 0, build_device=0).set_devices([0])
    ix.set_seed_kmer_len(8)
    qs = {b"A" * 31: n_a - 30, b"A" * 9: n_a - 8, b"A" * 8: n_a - 7, b"A" * 20 + b"CGTACGTTAGC": 1, b"A" * 30 + b"C": 1,
          b"A" * 30 + b"G": 0, b"CGTACGTTAGC" + b"A" * 20: 0, b"A" * 12 + b"C" + b"A" * 18: 0}
    for L in (31, 9, 8):
        batch = [q for q in qs if len(q) == L]
        q2d = np.frombuffer(b"".join(batch), np.uint8).reshape(len(batch), L)
        want = np.array([qs[q] for q in batch], dtype=np.uint64)
        assert np.array_equal(ix.count_kmers_nt2(q2d, True), want)
        assert np.array_equal(ix.count_kmers_nt2(q2d, False), want)
        assert np.array_equal(ix.parallel_count_csr(*synth.fixed_to_csr(q2d)), want)
    ragged = [b"A" * 40, b"A" * 31 + b"C", b"AAC", b"A" * 33 + b"CGTACGTTAGC"]
    assert ix.parallel_count(ragged).tolist() == [n_a - 39, 1, 1, 1]


def test_fuzz_small_indexes_against_oracle(oracle):
    """many small seeded indexes (both alphabets, 1..5000 symbols, 1..6 records, N/X runs, SA ratios 1..64): counts,
    ranges and locations of mixed queries equal the oracle's; packed kernels agree where they apply"""
    rng = np.random.default_rng(2024)
    sizes = [1, 2, 3, 7, 31, 100, 255, 256, 257, 1000, 5000]
    for trial in range(44):
        alphabet = trial % 2
        n = sizes[trial % len(sizes)]
        recs = int(min(max(1, n // 3), rng.integers(1, 7)))
        nfrac = float(rng.choice([0.0, 0.1, 0.4]))
        ratio = int(rng.choice([1, 2, 8, 64]))
        text, st, hd = synth.make_text(n, alphabet, 7000 + trial, recs if n >= 7 else 1, nfrac)
        ix = gpu_index(text, alphabet, ratio, 0, st, hd)
        oi = oracle.OracleIndex.from_text(text, alphabet, ratio, 0, st, hd)
        letters = b"ACGTN" if alphabet == 0 else b"ACDEFGHIKLMNPQRSTVWYX"
        qs = []
        for _ in range(60):
            L = int(rng.integers(1, min(40, n + 3) + 1))
            if rng.random() < 0.5 and n >= L:
                p = int(rng.integers(0, n - L + 1))
                q = bytes(text[p:p + L])
            else:
                q = bytes(rng.choice(np.frombuffer(letters, np.uint8), size=L))
            if b"$" in q:
                continue
            qs.append(q.lower() if rng.random() < 0.1 else q)
        qb, qo = __import__("awry_amd").fm_index.pack_queries(qs)
        want_c, _ = oi.parallel_count(qb, qo, 2)
        assert np.array_equal(ix.parallel_count_csr(qb, qo), want_c), trial
        woff, wg, wp, _ = oi.parallel_locate(qb, qo, 2)
        off, g, p = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(off, woff) and np.array_equal(g, wg) and np.array_equal(p, wp), trial
        if alphabet == 0 and n >= 4:
            for L in (1, 3, min(12, n)):
                q2d = np.concatenate([synth.sampled_queries(text, 40, L, trial, False, 0) if n > L else synth.random_queries(40, L, 0, trial),
                                      synth.random_queries(40, L, 0, trial + 1)])
                q2d = q2d[~((q2d == ord("N")) | (q2d == ord("$"))).any(axis=1)]
                if len(q2d) == 0:
                    continue
                want, _ = oi.parallel_count(*synth.fixed_to_csr(q2d), 2)
                for verify in (-1, 0):
                    ix.set_verify(verify)
                    ix.set_verify_kmers(verify >= 0)
                    assert np.array_equal(ix.count_kmers_nt2(q2d, True), want), (trial, L, verify)
                    got = ix.locate_reads_nt2(q2d)
                    wl = oi.parallel_locate(*synth.fixed_to_csr(q2d), 2)[:3]
                    assert all(np.array_equal(x, y) for x, y in zip(got, wl)), (trial, L, verify)
                ix.set_verify(-1)


def test_concurrent_host_threads_share_one_index(oracle):
    """the handle is immutable after set_devices: query entry points may be called from several host threads at once
    (the reference's FmIndex is Send + Sync)"""
    import threading
    text, st, hd = synth.make_text(200000, 0, 33, 2, 0.03)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    jobs = []
    for t in range(6):
        L = (12, 20, 31, 40, 9, 33)[t]
        q2d = np.concatenate([synth.sampled_queries(text, 3000, L, t), synth.random_queries(2000, L, 0, 50 + t)])
        qb, qo = synth.fixed_to_csr(q2d)
        jobs.append((qb, qo, oi.parallel_count(qb, qo, 2)[0], oi.parallel_locate(qb, qo, 2)[:3]))
    errors = []

    def work(j):
        try:
            qb, qo, want_c, want_l = jobs[j]
            for _ in range(3):
                assert np.array_equal(ix.parallel_count_csr(qb, qo), want_c)
                got = ix.parallel_locate_csr(qb, qo)
                assert all(np.array_equal(x, y) for x, y in zip(got, want_l))
        except Exception as e:  # noqa: BLE001
            errors.append((j, repr(e)))

    threads = [threading.Thread(target=work, args=(j,)) for j in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_all_count_kernel_schedules_agree(oracle):
    """the four schedules of the packed k-mer kernel (strided quads, LDS chunks, groups of four, two-phase
    probe + resume) are interchangeable: identical counts = the oracle's, including ragged tails and tiny batches"""
    from awry_amd import _lib
    L_ = _lib.load_library()
    text, st, hd = synth.make_text(500000, 0, 71, 2, 0.04)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    try:
        for L, nq in ((31, 40003), (17, 1), (9, 2), (32, 257), (12, 1023)):
            q2d = np.concatenate([synth.sampled_queries(text, nq // 2 + 1, L, L), synth.random_queries(nq // 2, L, 0, L + 1)])[:nq]
            want, _ = oi.parallel_count(*synth.fixed_to_csr(q2d), 4)
            for k in (-1, 5):
                ix.set_seed_kmer_len(k)
                for mode in (0, 1, 2, 3):
                    L_.awry_debug_set_count_kernel(mode)
                    assert np.array_equal(ix.count_kmers_nt2(q2d, True), want), (L, nq, k, mode)
                    assert np.array_equal(ix.count_kmers_nt2(q2d, False), want), (L, nq, k, mode)
    finally:
        L_.awry_debug_set_count_kernel(-1)


def test_default_device_policies():
    """awry_set_devices picks the accelerators by itself: the smallest seed k with 4^k >= 12 n, seed-and-verify structures resident
    for reads; the k-mer kernel leaves verify off unless asked; everything can be switched off again"""
    text, st, hd = synth.make_text(1_000_000, 0, 3, 1, 0.02)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    assert ix.seed_kmer_len() == 12            # 4^12 = 1.7e7 >= 1.2e7 > 4^11
    assert ix.verify_enabled() and ix.locate_sa_ratio() == 1
    assert "probe" in ix.count_schedule(31)    # 4^12 >= 3 n: two-phase
    ix.set_seed_kmer_len(8)
    assert ix.count_schedule(31) == "count_nt2_quad4_kernel" and ix.count_schedule(5) == "count_nt2_quad4_kernel"
    ix.set_verify(-1)
    ix.set_locate_sa_ratio(0)
    assert not ix.verify_enabled() and ix.locate_sa_ratio() == 8
    aa, st, hd = synth.make_text(50_000, 1, 4, 3)
    ax = gpu_index(aa, 1, 8, 0, st, hd)
    assert ax.seed_kmer_len() == 4 and ax.verify_enabled()       # floor(log20(5e4)) + 1; dense SA + byte text for the generic kernel
    ax.set_verify(-1)
    assert not ax.verify_enabled()


def test_amino_seed_table_does_not_change_counts(oracle):
    """the 20^k amino seed table (standard residues only; X, lower case, unknown letters bypass it) against the oracle"""
    text, st, hd = synth.make_text(300000, 1, 77, 40, 0.02)
    ix = gpu_index(text, 1, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 1, 8, 0, st, hd)
    qs = []
    for L in (1, 2, 3, 4, 5, 6, 8, 12, 20):
        qs += [bytes(q) for q in synth.sampled_queries(text, 150, L, L, False, 1)]
        qs += [bytes(q) for q in synth.random_queries(150, L, 1, 100 + L)]
    qs += [b"X", b"AX", b"XA", b"MKVX", b"mkvl", b"ACDEFB", b"ZZZZ", b"AAAAAAAAAAAA"]
    qb, qo = __import__("awry_amd").fm_index.pack_queries(qs)
    want_c, _ = oi.parallel_count(qb, qo, 4)
    want_l = oi.parallel_locate(qb, qo, 4)[:3]
    assert ix.seed_kmer_len() == 5
    for k in (-1, 0, 1, 2, 4):
        ix.set_seed_kmer_len(k)
        assert np.array_equal(ix.parallel_count_csr(qb, qo), want_c), k
        got = ix.parallel_locate_csr(qb, qo)
        assert all(np.array_equal(x, y) for x, y in zip(got, want_l)), k


@pytest.mark.parametrize("L", [7, 8, 9, 12, 15, 16, 17, 23, 24, 25])
def test_amino_kmer_two_phase_schedule(oracle, L):
    """equal-length amino batches (BASELINE configs[3] shape): the per-lane probe pass plus the generic kernel on what it
    lists, device-resident (awry_dev_count_ascii_uniform) and through parallel_count, against the oracle and against the
    generic kernel with offsets -- residues from the text, random ones, X / unknown letters / lower case anywhere, with
    every seed length the table can have (also none) and with the accelerators on and off"""
    import torch
    text, st, hd = synth.make_text(300000, 1, 91, 40, 0.02)
    # a repeated region: seed entries with several rows and queries with several occurrences
    text = text.copy()
    text[200000:201000] = text[1000:2000]
    ix = gpu_index(text, 1, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 1, 8, 0, st, hd)
    rng = np.random.default_rng(L)
    nq = 20000
    pres = synth.sampled_queries(text, nq // 2, L, 5 + L, False, 1)
    pres[: nq // 8] = synth.sampled_queries(text[1000:2000], nq // 8, L, 9, False, 1)
    rnd = synth.random_queries(nq // 2, L, 1, 200 + L)
    q2d = np.concatenate([pres, rnd])
    sub = rng.random(nq) < 0.3  # one substitution somewhere
    cols = rng.integers(0, L, size=nq)
    q2d[sub, cols[sub]] = synth.AA[rng.integers(0, len(synth.AA), size=int(sub.sum()))]
    odd = rng.random(nq) < 0.05  # X, letters outside the alphabet, digits: all search as X
    q2d[odd, cols[odd]] = np.frombuffer(b"XBZJ7*", dtype=np.uint8)[rng.integers(0, 6, size=int(odd.sum()))]
    low = rng.random(nq) < 0.1
    q2d[low] |= 0x20
    q2d = q2d[rng.permutation(nq)]
    qb = np.ascontiguousarray(q2d.reshape(-1))
    qo = (np.arange(nq + 1, dtype=np.uint64) * np.uint64(L))
    want, _ = oi.parallel_count(qb, qo, 4)
    want_loc = oi.parallel_locate(qb, qo, 4)[:3]
    assert (want > 1).sum() > 100 and (want == 1).sum() > 1000 and (want == 0).sum() > 1000
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for verify in (2, -1):
        ix.set_verify(verify)
        for k in (-1, 0, 1, 3, 5):
            ix.set_seed_kmer_len(k)
            d_c = torch.full((nq,), -1, dtype=torch.int64, device=dev)
            d_s = torch.full((nq,), 77, dtype=torch.uint8, device=dev)
            ix.dev_count_ascii_uniform(d_q.data_ptr(), nq, L, d_c.data_ptr(), d_s.data_ptr(), stream, 0)
            d_c2 = torch.zeros(nq, dtype=torch.int64, device=dev)
            ix.dev_count_ascii(d_q.data_ptr(), d_off.data_ptr(), nq, d_c2.data_ptr(), None, None, stream, 0)
            torch.cuda.synchronize()
            assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), want), (verify, k)
            assert np.array_equal(d_c2.cpu().numpy().astype(np.uint64), want), (verify, k)
            assert int(d_s.max()) == 0
            assert np.array_equal(ix.parallel_count_csr(qb, qo), want), (verify, k)
            if k in (-1, 3):  # parallel_locate: the same schedule is the count pass and hands the locate pass its range words
                got = ix.parallel_locate_csr(qb, qo)
                assert all(np.array_equal(x, y) for x, y in zip(got, want_loc)), (verify, k)
    # undefined bytes are named by the same pass that counts the rest
    bad = q2d.copy()
    bad[777, L // 2] = ord("$")
    bad[4242, L - 1] = 0xC3
    d_qb = torch.from_numpy(np.concatenate([bad.reshape(-1), np.zeros(16, dtype=np.uint8)])).to(dev)
    d_s = torch.zeros(nq, dtype=torch.uint8, device=dev)
    d_c = torch.zeros(nq, dtype=torch.int64, device=dev)
    ix.set_verify(2)
    ix.set_seed_kmer_len(-1)
    ix.dev_count_ascii_uniform(d_qb.data_ptr(), nq, L, d_c.data_ptr(), d_s.data_ptr(), stream, 0)
    torch.cuda.synchronize()
    st_h = d_s.cpu().numpy()
    assert sorted(np.nonzero(st_h)[0].tolist()) == [777, 4242] and st_h[777] == 2 and st_h[4242] == 3
    ok = np.ones(nq, dtype=bool)
    ok[[777, 4242]] = False
    assert np.array_equal(d_c.cpu().numpy().astype(np.uint64)[ok], want[ok])
    for fn in (ix.parallel_count_csr, ix.parallel_locate_csr):
        with pytest.raises(AwryError) as e:
            fn(np.ascontiguousarray(bad.reshape(-1)), qo)
        assert e.value.code == ERR_INVALID_QUERY and "query 777" in str(e.value)


def test_uniform_entry_point_on_a_nucleotide_index(oracle):
    """awry_dev_count_ascii_uniform on a nucleotide index: packed on the device, packed kernels, the queries with letters
    outside ACGT redone by the generic kernel over the pack kernel's list (which also names undefined ones); with the
    accelerators off and without a seed table too; the generic kernel at q * len when the packed path does not apply"""
    import torch
    text, st, hd = synth.make_text(200000, 0, 5, 2, 0.03)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for L in (1, 5, 21, 32, 33, 40, 101):
        q2d = np.concatenate([synth.sampled_queries(text, 3000, L, L, True, 0), synth.random_queries(3000, L, 0, L + 1)])
        q2d[::97, L // 2] = ord("N")
        q2d[5] = np.frombuffer(bytes(q2d[5]).lower(), np.uint8)
        q2d[11, 0] = ord("R")
        qb = np.ascontiguousarray(q2d.reshape(-1))
        qo = np.arange(len(q2d) + 1, dtype=np.uint64) * np.uint64(L)
        want, _ = oi.parallel_count(qb, qo, 4)
        d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
        for verify, k in ((2, -1), (-1, -1), (2, 0)):
            ix.set_verify(verify)
            ix.set_seed_kmer_len(k)
            d_c = torch.full((len(q2d),), -1, dtype=torch.int64, device=dev)
            d_s = torch.full((len(q2d),), 9, dtype=torch.uint8, device=dev)
            ix.dev_count_ascii_uniform(d_q.data_ptr(), len(q2d), L, d_c.data_ptr(), d_s.data_ptr(), stream, 0)
            d_c2 = torch.full((len(q2d),), -1, dtype=torch.int64, device=dev)
            ix.dev_count_ascii_uniform(d_q.data_ptr(), len(q2d), L, d_c2.data_ptr(), None, stream, 0)
            torch.cuda.synchronize()
            assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), want), (L, verify, k)
            assert np.array_equal(d_c2.cpu().numpy().astype(np.uint64), want), (L, verify, k)
            assert int(d_s.max()) == 0
        ix.set_verify(2)
        ix.set_seed_kmer_len(-1)
        bad = q2d.copy()
        bad[1234, L - 1] = ord("$")
        d_b = torch.from_numpy(np.concatenate([bad.reshape(-1), np.zeros(16, dtype=np.uint8)])).to(dev)
        d_c = torch.zeros(len(q2d), dtype=torch.int64, device=dev)
        d_s = torch.zeros(len(q2d), dtype=torch.uint8, device=dev)
        ix.dev_count_ascii_uniform(d_b.data_ptr(), len(q2d), L, d_c.data_ptr(), d_s.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        st_h = d_s.cpu().numpy()
        assert np.nonzero(st_h)[0].tolist() == [1234] and st_h[1234] == 2
        ok = np.arange(len(q2d)) != 1234
        assert np.array_equal(d_c.cpu().numpy().astype(np.uint64)[ok], want[ok])


def test_amino_kmer_schedule_on_a_large_batch(oracle):
    """a batch of millions of amino k-mers (every block of the grid sees several rounds of queries and keeps a list of its
    own): same counts as the oracle, call after call on the same stream (the lists are reused) and on another stream"""
    import torch
    text, st, hd = synth.make_text(2_000_000, 1, 93, 300, 0.01)
    ix = gpu_index(text, 1, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 1, 8, 0, st, hd)
    L, nq = 12, 2_300_001
    q2d = np.concatenate([synth.sampled_queries(text, nq // 2, L, 3, False, 1), synth.random_queries(nq - nq // 2, L, 1, 4)])
    q2d = q2d[np.random.default_rng(1).permutation(nq)]
    qb = np.ascontiguousarray(q2d.reshape(-1))
    qo = np.arange(nq + 1, dtype=np.uint64) * np.uint64(L)
    want, _ = oi.parallel_count(qb, qo, 8)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(qb).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for rep in range(3):  # back to back: the lists of one call are reused by the next
        d_c = torch.full((nq,), -1, dtype=torch.int64, device=dev)
        d_s = torch.full((nq,), 9, dtype=torch.uint8, device=dev)
        ix.dev_count_ascii_uniform(d_q.data_ptr(), nq, L, d_c.data_ptr(), d_s.data_ptr(), stream, 0)
        outs.append((d_c, d_s))
    torch.cuda.synchronize()
    for d_c, d_s in outs:
        assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), want)
        assert int(d_s.max()) == 0
    side = torch.cuda.Stream()  # and on another stream of the caller's
    with torch.cuda.stream(side):
        d_c = torch.zeros(nq, dtype=torch.int64, device=dev)
        ix.dev_count_ascii_uniform(d_q.data_ptr(), nq, L, d_c.data_ptr(), None, side.cuda_stream, 0)
    side.synchronize()
    assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), want)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), want)


@pytest.mark.gpu
def test_host_packed_boundary_details(oracle):
    """the host-packed lanes of parallel_count / parallel_locate (AVX2 packer on the worker pool, pinned staging, counts as
    32-bit words over PCIe): a batch of several chunks in which a third of the queries hold other letters (N, IUPAC codes,
    U, lower case) and travel as compact batches for the generic kernel; the first undefined query is named by its index;
    hits_out = NULL returns the same offsets and text positions without (record, offset) pairs"""
    text, st, hd = synth.make_text(500000, 0, 91, 4, 0.004)  # (short runs of N: a query of N's inside one matches all of it)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(5)
    for L in (31, 75):
        nq = 1_300_000 if L == 31 else 300_000  # (L = 31: more than one full-size chunk of the host lanes)
        q2d = synth.sampled_queries(text, nq, L, 100 + L, skip_amb=False)
        q2d[q2d == ord("$")] = ord("A")
        rnd = rng.random(nq) < 0.3
        q2d[rnd] = synth.random_queries(int(rnd.sum()), L, 0, 7)
        other = np.nonzero(rng.random(nq) < 0.33)[0]
        q2d[other, rng.integers(0, L, len(other))] = rng.choice(np.frombuffer(b"NnRYKMSWUu", np.uint8), len(other))
        low = np.nonzero(rng.random(nq) < 0.05)[0]
        q2d[low] |= 0x20
        qb, qo = synth.fixed_to_csr(q2d)
        out = np.full(nq, 12345, dtype=np.uint64)
        counts = ix.parallel_count_csr(qb, qo, out)
        assert counts is out
        sample = np.sort(rng.choice(nq, size=40000, replace=False))
        sb, so = synth.fixed_to_csr(q2d[sample])
        ooff, ogpos, opos, _ = oi.parallel_locate(sb, so, 4)
        assert np.array_equal(np.diff(ooff), counts[sample])
        off, g, p = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(np.diff(off), counts) and len(g) == len(p) == int(off[-1])
        assert np.array_equal(np.concatenate([g[int(off[i]):int(off[i + 1])] for i in sample]), ogpos)
        assert np.array_equal(np.concatenate([p[int(off[i]):int(off[i + 1])] for i in sample]), opos)
        off2, g2, p2 = ix.parallel_locate_csr(qb, qo, want_pos=False)
        assert np.array_equal(off2, off) and np.array_equal(g2, g) and p2.shape == (0, 2)
        bad = q2d.copy()
        bad[nq - 7, L // 2] = ord("$")
        bad[nq - 3, 0] = ord("#")
        for fn in (ix.parallel_count_csr, ix.parallel_locate_csr):
            with pytest.raises(AwryError) as e:
                fn(*synth.fixed_to_csr(bad))
            assert e.value.code == ERR_INVALID_QUERY and "query %d:" % (nq - 7) in str(e.value)


@pytest.mark.parametrize("alphabet", [0, 1])
def test_non_canonical_text_letters(oracle, alphabet):
    """a text with IUPAC codes / non-standard residues, U and lower case is indexed as its canonical form, so counts and
    locations equal brute force over the symbol indices of text and query -- with the seed-and-verify accelerators and
    without them, on the host-built and on the GPU-built index (round 1 sorted suffixes by raw bytes: LF steps and text
    comparison then disagreed)"""
    rng = np.random.default_rng(40 + alphabet)
    letters = b"ACGT" * 6 + b"acgtRYKMSWNnUu" if alphabet == 0 else b"ACDEFGHIKLMNPQRSTVWY" * 3 + b"acdxXBJOUZbz"
    n = 4000
    body = rng.choice(np.frombuffer(letters, np.uint8), size=n)
    body[1000:1040] = ord("N") if alphabet == 0 else ord("X")
    text = np.concatenate([body, np.frombuffer(b"$", np.uint8)])
    O = oracle.lib()
    qs = []
    for L in (3, 6, 11, 17, 30):
        st = rng.integers(0, n - L, size=120)
        for s in st:
            qs.append(bytes(body[s:s + L]))
            qs.append(bytes(body[s:s + L]).upper())
        qs += [bytes(x) for x in synth.random_queries(40, L, alphabet, L)]
    u64p = C.POINTER(C.c_uint64)
    want_count, want_pos = [], []
    for q in qs:
        buf = np.zeros(n + 1, np.uint64)
        c = O.orc_brute_locate(alphabet, text.ctypes.data, len(text), q, len(q), buf.ctypes.data_as(u64p), len(buf))
        want_count.append(c)
        want_pos.append(np.sort(buf[:c]))
    for dev in (awry_build_host(), 0):
        ix = FmIndex.from_text(text, alphabet, 4, 0, build_device=dev).set_devices([0])
        for verify in (2, -1):
            ix.set_verify(verify)
            counts = ix.parallel_count(qs)
            assert np.array_equal(counts, np.array(want_count, np.uint64)), (dev, verify)
            off, g, p = ix.parallel_locate_csr(*awry_pack(qs))
            for i in range(len(qs)):
                assert np.array_equal(np.sort(g[int(off[i]):int(off[i + 1])]), want_pos[i]), (dev, verify, qs[i])
            for q, c in list(zip(qs, want_count))[::37]:
                assert ix.count_string(q) == c


def test_assumed_uniform_length_is_checked(oracle):
    """parallel_count assumes the first query's length for the whole batch and lets the host packer check the offsets as it
    goes; a batch whose lengths differ but whose bytes add up to n x that length must fall back to the planned path"""
    text, st, hd = synth.make_text(300000, 0, 17, 2, 0.02)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    n = 70001
    lens = np.full(n, 24)
    lens[1::2] = 23
    lens[2::2] = 25  # first query 24 letters, then 23 / 25 alternating: n * 24 bytes in all
    assert lens.sum() == n * 24
    rng = np.random.default_rng(3)
    qo = np.zeros(n + 1, dtype=np.uint64)
    qo[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - 40, size=n)
    idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
    qb = text[idx].copy()
    qb[qb == ord("$")] = ord("A")
    want, _ = oi.parallel_count(qb, qo, 4)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), want)
    # and the common case itself: one length, above the threshold of the assumption
    q2d = synth.sampled_queries(text, 80000, 24, 5)
    qb, qo = synth.fixed_to_csr(q2d)
    want, _ = oi.parallel_count(qb, qo, 4)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), want)


def _check_located(text, st, q2d, positions, off, g, p, counts):
    """size-independent properties of a locate result: count == number of locations, every located window holds its
    query, every query drawn from the text finds its own position, (record, offset) = the largest record start <= position"""
    L = q2d.shape[1]
    assert np.array_equal(np.diff(off), counts) and len(g) == int(off[-1])
    qi = np.repeat(np.arange(len(q2d)), counts.astype(np.int64))
    assert np.array_equal(text[g.astype(np.int64)[:, None] + np.arange(L)[None, :]], q2d[qi])
    for i in range(0, len(positions), 211):
        assert positions[i] in g[int(off[i]):int(off[i + 1])]
    if len(p):
        starts = np.array(st, dtype=np.uint64)
        si = np.searchsorted(starts, g, side="right") - 1
        assert np.array_equal(p[:, 0], si.astype(np.uint64)) and np.array_equal(p[:, 1], g - starts[si])


def test_grch38_scale_101bp_reads(oracle, tmp_path):
    """BASELINE.json configs[2] at its own index size: 101-bp reads sampled from a GRCh38-scale text (3.1 Gbp, 25 records)
    through awry_locate_batch and through the device-resident pipeline, with the accelerators (seed-and-verify, dense SA)
    and without them (LF steps, walks to the file's samples); the oracle (same index via an .awry round trip) on a sample"""
    n, recs, L = 3_100_000_000, 25, 101
    text, st, hd = synth.make_text(n, 0, 0xA5A50000 + 2, recs, 0.05)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(11)
    pos = rng.integers(0, n - L, size=300_000)
    win = text[pos[:, None] + np.arange(L)[None, :]]
    ok = ~(win == ord("N")).any(axis=1)
    pos, present = pos[ok][:200_000], win[ok][:200_000]
    mutated = present[:20_000].copy()
    mutated[np.arange(20_000), rng.integers(0, L, 20_000)] = synth.NT[rng.integers(0, 4, 20_000)]
    q2d = np.concatenate([present, mutated, synth.random_queries(20_000, L, 0, 12)])
    qb, qo = synth.fixed_to_csr(q2d)
    path = str(tmp_path / "g.awry")
    ix.save(path)
    oi = oracle.OracleIndex.load(path)
    ns = 20_000
    sample = np.concatenate([np.arange(0, 200_000, 20)[:ns // 2], np.arange(200_000, 200_000 + ns // 2)])
    sb, so = synth.fixed_to_csr(q2d[sample])
    ooff, ogpos, opos, _ = oi.parallel_locate(sb, so, 8)
    oi.close()
    os.remove(path)
    results = []
    for accel in (True, False):
        if not accel:
            ix.set_verify(-1)
            ix.set_locate_sa_ratio(0)
        assert bool(ix.verify_enabled()) == accel
        counts = ix.parallel_count_csr(qb, qo)
        assert (counts[:len(present)] >= 1).all()
        off, g, p = ix.parallel_locate_csr(qb, qo)                # the host boundary
        _check_located(text, st, q2d, pos, off, g, p, counts)
        off2, g2, p2 = ix.locate_reads_nt2(q2d)                   # packed words resident in HBM: count, scan, locate
        assert np.array_equal(off2, off) and np.array_equal(g2, g) and np.array_equal(p2, p)
        assert np.array_equal(np.diff(ooff), counts[sample])
        assert np.array_equal(np.concatenate([g[int(off[i]):int(off[i + 1])] for i in sample]), ogpos)
        assert np.array_equal(np.concatenate([p[int(off[i]):int(off[i + 1])] for i in sample]), opos)
        results.append((off, g))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])


def test_swissprot_scale_amino_12mers(oracle, tmp_path):
    """BASELINE.json configs[3] at its own size: 9e7 residues in 2.5e5 records, 12-mers through awry_count_batch (the amino
    k-mer schedule), the device entry point and awry_locate_batch; properties on everything, the oracle on 1e5 queries"""
    n, recs, L = 90_000_000, 250_000, 12
    text, st, hd = synth.make_text(n, 1, 0xA5A50004, recs, 0.0)
    ix = gpu_index(text, 1, 8, 0, st, hd)
    rng = np.random.default_rng(12)
    pos = rng.integers(0, n - L, size=200_000)
    present = text[pos[:, None] + np.arange(L)[None, :]]     # windows that span a record boundary hold X: they search as X
    q2d = np.concatenate([present, synth.random_queries(200_000, L, 1, 13)])
    qb, qo = synth.fixed_to_csr(q2d)
    counts = ix.parallel_count_csr(qb, qo)
    assert (counts[:len(present)] >= 1).all()
    d_q, d_c = ix.dev_upload(np.concatenate([qb, np.zeros(16, np.uint8)])), ix.dev_malloc(8 * len(q2d))
    ix.dev_count_ascii_uniform(d_q, len(q2d), L, d_c)
    ix.dev_synchronize()
    assert np.array_equal(ix.dev_download(d_c, (len(q2d),), np.uint64), counts)
    ix.dev_free(d_q)
    ix.dev_free(d_c)
    off, g, p = ix.parallel_locate_csr(qb, qo)
    _check_located(text, st, q2d, pos, off, g, p, counts)
    path = str(tmp_path / "a.awry")
    ix.save(path)
    oi = oracle.OracleIndex.load(path)
    sample = np.sort(rng.choice(len(q2d), size=100_000, replace=False))
    sb, so = synth.fixed_to_csr(q2d[sample])
    ooff, ogpos, opos, _ = oi.parallel_locate(sb, so, 8)
    oi.close()
    assert np.array_equal(np.diff(ooff), counts[sample])
    assert np.array_equal(np.concatenate([g[int(off[i]):int(off[i + 1])] for i in sample]), ogpos)
    assert np.array_equal(np.concatenate([p[int(off[i]):int(off[i + 1])] for i in sample]), opos)
    # unequal lengths (8..24 residues): no k-mer schedule for these, the generic pipeline must agree with the oracle too
    lens = rng.integers(8, 25, size=50_000)
    ro = np.zeros(len(lens) + 1, dtype=np.uint64)
    ro[1:] = np.cumsum(lens)
    starts = rng.integers(0, n - 30, size=len(lens))
    idx = np.repeat(starts, lens) + (np.arange(int(ro[-1])) - np.repeat(ro[:-1].astype(np.int64), lens))
    rb = text[idx].copy()
    want, _ = oracle.OracleIndex.load(path).parallel_count(rb, ro, 8)
    assert np.array_equal(ix.parallel_count_csr(rb, ro), want)


@pytest.mark.parametrize("L", [25, 31, 40, 64, 333, 1024, 1025])
def test_amino_long_queries_of_one_length(oracle, L):
    """amino queries of more than 24 residues (peptides, protein fragments; up to 1024 take the k-mer schedule's LONG pass:
    the last 24 residues as for k-mers, the rest screened for undefined bytes and compared with the text for surviving
    candidates; 1025 goes to the generic kernel): from the text (some from a duplicated region: several candidates), with
    one residue changed in the far or in the near part, random, X / unknown letters, lower case, windows at the very
    beginning of the text -- device-resident, parallel_count and parallel_locate against the oracle"""
    import torch
    text, st, hd = synth.make_text(300000, 1, 95, 12, 0.0)
    text = text.copy()
    text[150000:153000] = text[3000:6000]
    ix = gpu_index(text, 1, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 1, 8, 0, st, hd)
    rng = np.random.default_rng(L)
    nq = 9000
    starts = rng.integers(0, len(text) - L - 2, size=nq)
    starts[:1500] = rng.integers(3000, 6000 - min(L, 2900), size=1500)
    starts[1500:1520] = np.arange(20)
    q2d = text[starts[:, None] + np.arange(L)[None, :]].copy()
    q2d[q2d == ord("$")] = ord("A")
    far_mut = np.arange(3000, 4500)
    q2d[far_mut, rng.integers(0, L - 24, size=len(far_mut))] = synth.AA[rng.integers(0, 20, size=len(far_mut))]
    near_mut = np.arange(4500, 5500)
    q2d[near_mut, rng.integers(L - 24, L, size=len(near_mut))] = synth.AA[rng.integers(0, 20, size=len(near_mut))]
    q2d[5500:6000] = synth.random_queries(500, L, 1, L)
    odd = rng.random(q2d.shape) < 0.0005
    odd[1500:1520] = False
    q2d[odd] = np.frombuffer(b"XBZJ7*", dtype=np.uint8)[rng.integers(0, 6, size=int(odd.sum()))]
    q2d[6000:6500] |= 0x20
    qb, qo = synth.fixed_to_csr(q2d)
    want_loc = oi.parallel_locate(qb, qo, 4)[:3]
    want = np.diff(want_loc[0])
    assert (want > 1).sum() > 100 and (want == 0).sum() > 500 and (want[1500:1520] >= 1).all()
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for verify in (2, -1):
        ix.set_verify(verify)
        for k in (-1, 3):
            ix.set_seed_kmer_len(k)
            d_c = torch.full((nq,), -1, dtype=torch.int64, device=dev)
            ix.dev_count_ascii_uniform(d_q.data_ptr(), nq, L, d_c.data_ptr(), None, stream, 0)
            torch.cuda.synchronize()
            assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), want), (verify, k)
            assert np.array_equal(ix.parallel_count_csr(qb, qo), want), (verify, k)
            got = ix.parallel_locate_csr(qb, qo)
            assert all(np.array_equal(x, y) for x, y in zip(got, want_loc)), (verify, k)
    for where in (0, L - 25, L - 1):  # an undefined byte in the far part, at its end, in the tail
        bad = q2d.copy()
        bad[4321, where] = ord("$")
        with pytest.raises(AwryError) as e:
            ix.parallel_count_csr(*synth.fixed_to_csr(bad))
        assert e.value.code == ERR_INVALID_QUERY and "query 4321" in str(e.value)


@pytest.mark.parametrize("lo,hi", [(1, 40), (8, 24), (5, 13), (20, 130), (900, 1030)])
def test_amino_kmer_schedule_with_unequal_lengths(oracle, lo, hi):
    """amino batches of unequal lengths take the same two-phase schedule with per-query lengths (residues from the text,
    random ones, X / unknown letters, lower case; lengths below the seed length and above 24 residues are listed for the
    generic kernel): device-resident (awry_dev_count_ascii), parallel_count and parallel_locate against the oracle, with
    several seed lengths and with the accelerators on and off"""
    import torch
    text, st, hd = synth.make_text(300000, 1, 93, 30, 0.02)
    text = text.copy()
    dup = 1000 if hi <= 200 else 3000
    text[150000:150000 + dup] = text[3000:3000 + dup]
    ix = gpu_index(text, 1, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 1, 8, 0, st, hd)
    rng = np.random.default_rng(lo * 100 + hi)
    nq = 24000 if hi <= 200 else 6000
    lens = rng.integers(lo, hi + 1, size=nq)
    qo = np.zeros(nq + 1, dtype=np.uint64)
    qo[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - hi - 2, size=nq)
    starts[: nq // 8] = rng.integers(3000, 3900, size=nq // 8)
    starts[nq // 8: nq // 8 + 10] = np.arange(10)  # windows at the very beginning of the text
    idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
    qb = text[idx].copy()
    rmask = np.repeat(rng.random(nq) < 0.4, lens)
    qb[rmask] = synth.AA[rng.integers(0, 20, size=int(rmask.sum()))]
    odd = rng.random(len(qb)) < min(0.004, 0.3 / hi)
    qb[odd] = np.frombuffer(b"XBZJ7*", dtype=np.uint8)[rng.integers(0, 6, size=int(odd.sum()))]
    qb[qb == ord("$")] = ord("A")
    low = np.repeat(rng.random(nq) < 0.1, lens)
    qb[low] |= 0x20
    want_loc = oi.parallel_locate(qb, qo, 4)[:3]
    want = np.diff(want_loc[0])
    assert (want > 1).sum() > 50 and (want == 0).sum() > 1000
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for verify in (2, -1):
        ix.set_verify(verify)
        for k in (-1, 3, 0):
            ix.set_seed_kmer_len(k)
            d_c = torch.full((nq,), -1, dtype=torch.int64, device=dev)
            d_s = torch.full((nq,), 77, dtype=torch.uint8, device=dev)
            ix.dev_count_ascii(d_q.data_ptr(), d_off.data_ptr(), nq, d_c.data_ptr(), None, d_s.data_ptr(), stream, 0)
            torch.cuda.synchronize()
            assert np.array_equal(d_c.cpu().numpy().astype(np.uint64), want), (verify, k)
            assert int(d_s.max()) == 0
            assert np.array_equal(ix.parallel_count_csr(qb, qo), want), (verify, k)
            got = ix.parallel_locate_csr(qb, qo)
            assert all(np.array_equal(x, y) for x, y in zip(got, want_loc)), (verify, k)
    bad = qb.copy()
    bad[int(qo[1234])] = ord("#")
    with pytest.raises(AwryError) as e:
        ix.parallel_count_csr(bad, qo)
    assert e.value.code == ERR_INVALID_QUERY and "query 1234" in str(e.value)


def test_wide_row_kernels(oracle):
    """indexes of 2^32 rows or more take packed kernels with 64-bit rows and 16-byte seed entries (the reference is u64
    throughout, src/search.rs:7).  Such an index cannot be built in a test (an hour of host SA-IS), so the kernels are
    forced onto small ones (awry_debug_force_wide_rows) and must give the oracle's counts and locations: k-mers and reads of
    equal and unequal lengths, letters outside ACGT, with several seed lengths, device-resident and through the host paths"""
    import awry_amd
    L_ = awry_amd.load_library()
    text, st, hd = repeat_text(33)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    L_.awry_debug_force_wide_rows(1)
    try:
        ix = gpu_index(text, 0, 8, 0, st, hd)
    finally:
        L_.awry_debug_force_wide_rows(0)
    assert "count_nt2_wide_kernel" in ix.count_schedule(31) and not ix.verify_enabled() and ix.seed_kmer_len() >= 1
    rng = np.random.default_rng(44)
    for k in (-1, 0, 1, 5):
        ix.set_seed_kmer_len(k)
        for L in (1, 7, 31, 32, 33, 101, 150):
            q2d = np.concatenate([synth.sampled_queries(text, 1500, L, L), synth.random_queries(800, L, 0, L + 1)])
            qb, qo = synth.fixed_to_csr(q2d)
            want = oi.parallel_locate(qb, qo, 4)[:3]
            if L <= 32:
                assert np.array_equal(ix.count_kmers_nt2(q2d, True), np.diff(want[0])), (k, L)
                assert np.array_equal(ix.count_kmers_nt2(q2d, False), np.diff(want[0])), (k, L)
            off, g, p = ix.locate_reads_nt2(q2d)
            assert np.array_equal(off, want[0]) and np.array_equal(g, want[1]) and np.array_equal(p, want[2]), (k, L)
            if k != 0:  # the alternative two-phase schedule (per-lane probe pass + listed quads): same results
                L_.awry_debug_set_count_kernel(3)
                try:
                    if L <= 32:
                        assert np.array_equal(ix.count_kmers_nt2(q2d, True), np.diff(want[0])), (k, L)
                    off, g, p = ix.locate_reads_nt2(q2d)
                    assert np.array_equal(off, want[0]) and np.array_equal(g, want[1]) and np.array_equal(p, want[2]), (k, L)
                finally:
                    L_.awry_debug_set_count_kernel(-1)
            if k == -1:
                assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(want[0])), (k, L)
                got = ix.parallel_locate_csr(qb, qo)
                assert all(np.array_equal(x, y) for x, y in zip(got, want)), (k, L)
    # unequal lengths and letters outside ACGT through the host paths (host packer, compact redo, ragged wide kernel)
    ix.set_seed_kmer_len(-1)
    nq = 70000
    lens = rng.integers(1, 60, size=nq)
    qo = np.zeros(nq + 1, dtype=np.uint64)
    qo[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - 70, size=nq)
    idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
    qb = text[idx].copy()
    rmask = np.repeat(rng.random(nq) < 0.3, lens)
    qb[rmask] = synth.NT[rng.integers(0, 4, size=int(rmask.sum()))]
    other = rng.random(len(qb)) < 0.002
    qb[other] = np.frombuffer(b"NRYu", np.uint8)[rng.integers(0, 4, size=int(other.sum()))]
    qb[qb == ord("$")] = ord("A")
    want = oi.parallel_locate(qb, qo, 4)[:3]
    for mode in (-1, 3):  # (3: the two-phase schedule with per-read lengths)
        L_.awry_debug_set_count_kernel(mode)
        try:
            assert np.array_equal(ix.parallel_count_csr(qb, qo), np.diff(want[0]))
            got = ix.parallel_locate_csr(qb, qo)
            assert all(np.array_equal(x, y) for x, y in zip(got, want))
        finally:
            L_.awry_debug_set_count_kernel(-1)
    with pytest.raises(AwryError):
        ix.set_verify(2)  # the verify accelerators are 32-bit structures


def test_ecoli_scale_21mers_against_the_oracle(oracle):
    """BASELINE configs[0]'s shape on the HIP path: an E. coli K-12-sized text (4.6 Mbp, one record), 10 k random and 10 k
    present 21-mers, counts through the host boundary and the packed kernels, locations of the present ones -- all against
    the oracle (which is what configs[0] itself runs, on the CPU)."""
    text, st, hd = synth.make_text(4_641_652, 0, 0xA5A50000 + 1, 1, 0.0)
    ix = gpu_index(text, 0, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    q2d = np.concatenate([synth.random_queries(10_000, 21, 0, 0x5EED21), synth.sampled_queries(text, 10_000, 21, 5)])
    qb, qo = synth.fixed_to_csr(q2d)
    want, _ = oi.parallel_count(qb, qo, 4)
    assert np.array_equal(ix.parallel_count_csr(qb, qo), want)
    assert np.array_equal(ix.count_kmers_nt2(q2d, True), want)
    assert np.array_equal(ix.count_kmers_nt2(q2d, False), want)
    assert (want[10_000:] >= 1).all() and want[:10_000].sum() <= 10  # a random 21-mer in 4.6 Mbp: 1e-6 each
    off, gpos, pos = ix.parallel_locate_csr(qb, qo)
    ooff, ogpos, opos, _ = oi.parallel_locate(qb, qo, 4)
    assert np.array_equal(off, ooff) and np.array_equal(gpos, ogpos) and np.array_equal(pos, opos)
    for q in (q2d[0], q2d[10_000], q2d[19_999]):
        assert ix.count_string(bytes(q)) == oi.count_string(bytes(q))


@pytest.mark.parametrize("tail_len", [3, 7, 8, 9, 15, 17, 24])
def test_ragged_amino_batch_that_ends_with_its_allocation(oracle, tail_len):
    """the amino k-mer pass reads a query with 8-byte loads anchored at the query's END (and at its start only where 8 bytes
    exist), so a caller's buffer may end with its last query: the batch is uploaded into an allocation of exactly its own
    size (no slack), with the last -- and the first -- query shorter than a word, and counted against the oracle"""
    text, st, hd = synth.make_text(120000, 1, 41, 12, 0.01)
    ix = gpu_index(text, 1, 8, 0, st, hd)
    oi = oracle.OracleIndex.from_text(text, 1, 8, 0, st, hd)
    k = ix.seed_kmer_len()
    rng = np.random.default_rng(tail_len)
    nq = 6000  # enough queries for the two-phase schedule (>= 4096)
    lens = rng.integers(max(k, 1), 25, size=nq)
    lens[0] = max(k, 3)         # a first query within the buffer's first seven bytes
    lens[-1] = max(k, tail_len)  # the last query ends the allocation
    qo = np.zeros(nq + 1, dtype=np.uint64)
    qo[1:] = np.cumsum(lens)
    starts = rng.integers(0, len(text) - 30, size=nq)
    idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
    qb = text[idx].copy()
    qb[qb == ord("$")] = ord("A")
    want, _ = oi.parallel_count(qb, qo, 4)
    d_q = ix.dev_upload(qb)           # exactly len(qb) bytes
    d_off = ix.dev_upload(qo)
    d_c = ix.dev_malloc(8 * nq)
    d_s = ix.dev_malloc(nq)
    try:
        ix.dev_count_ascii(d_q, d_off, nq, d_c, None, d_s)
        ix.dev_synchronize()
        assert np.array_equal(ix.dev_download(d_c, (nq,), np.uint64), want)
        assert int(ix.dev_download(d_s, (nq,), np.uint8).max()) == 0
    finally:
        for p in (d_q, d_off, d_c, d_s):
            ix.dev_free(p)
