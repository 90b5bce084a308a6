"""extended differential fuzz against the oracle (not part of the test suite): many seeded indexes of 1..200 K symbols,
thousands of mixed queries each, host batch entry points and the single-query entry points, default device policies.
usage: fuzz_gpu.py [trials] [seed] [aa]   (aa: only equal-length amino batches of 6..26 residues, >= 5000 queries -- the
amino k-mer schedule of the host pipeline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from awry_amd.fm_index import pack_queries
from oracle import oracle_ffi
from tests import synth

oracle_ffi.build()
if os.environ.get("AWRY_FUZZ_WIDE"):  # every index on the wide-row (64-bit) kernels, as if it had 2^32 rows or more
    awry_amd.load_library().awry_debug_force_wide_rows(1)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
only_aa = len(sys.argv) > 3 and sys.argv[3] == "aa"
t0 = time.time()
for trial in range(trials):
    alphabet = 1 if only_aa else int(rng.integers(0, 2))
    n = int(rng.choice([300, 3000, 20000, 200000] if only_aa else [5, 40, 300, 3000, 20000, 200000]))
    recs = int(min(max(1, n // 50), rng.integers(1, 9)))
    if n >= 20000 and rng.random() < 0.2:  # thousands of records: the locate kernels' bucket table over the record starts
        recs = int(rng.integers(1100, 3000))
    nfrac = float(rng.choice([0.0, 0.02, 0.3])) if alphabet == 0 else 0.0
    ratio = int(rng.choice([1, 3, 8, 16]))
    text, st, hd = synth.make_text(n, alphabet, int(rng.integers(1, 1 << 30)), recs if n >= 40 else 1, nfrac)
    if alphabet == 0 and n >= 3000 and rng.random() < 0.5:  # plant repeats: large seed ranges
        unit = text[100:100 + 60].copy()
        for s in rng.integers(0, n - 100, size=40):
            if not (text[s:s + 60] == ord("N")).any() and not np.isin(np.arange(s, s + 60), np.array(st[1:]) - 1).any():
                text[s:s + 60] = unit
    if alphabet == 0 and n >= 20000 and rng.random() < 0.5:  # repeat families with many diverged copies: seed buckets of hundreds
        for _ in range(int(rng.integers(1, 4))):              # to thousands of rows -- the left-context index's multi-level searches
            ulen, div = int(rng.integers(40, 400)), float(rng.choice([0.0, 0.02, 0.06, 0.12]))
            cons = synth.NT[rng.integers(0, 4, size=ulen)]
            for s in rng.integers(0, n - ulen - 1, size=int(rng.integers(50, max(60, n // (2 * ulen))))):
                if (text[s:s + ulen] == ord("N")).any():
                    continue  # (keeps the record delimiters and the runs of N)
                cp = cons.copy()
                m = rng.random(ulen) < div
                cp[m] = synth.NT[rng.integers(0, 4, size=int(m.sum()))]
                text[s:s + ulen] = cp
    if only_aa and rng.random() < 0.4:  # ambiguity residues in the text, and so in the queries drawn from it
        text[rng.integers(0, n, size=max(1, n // 40))] = ord("X")
    ix = awry_amd.FmIndex.from_text(text, alphabet, ratio, 0, st, hd).set_devices([0])
    oi = oracle_ffi.OracleIndex.from_text(text, alphabet, ratio, 0, st, hd)
    letters = np.frombuffer(b"ACGT" if alphabet == 0 else b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    mode = int(rng.integers(0, 3))  # 0 fixed length, 1 ragged, 2 ragged with ambiguity letters / lower case
    nq = int(rng.choice([1, 70, 5000, 30000, 70000], p=[0.24, 0.24, 0.24, 0.2, 0.08]))  # 70000: above the host path's assumed-uniform threshold
    Lfix = int(rng.integers(1, min(120, n) + 1))
    if n >= 3000 and rng.random() < 0.15:  # long queries: amino LONG pass, multi-word reads
        Lfix = int(rng.integers(121, 700))
    if only_aa:
        mode, nq, Lfix = 0, int(rng.choice([5000, 12000])), int(rng.integers(6, 27))
    qs = []
    for i in range(nq):
        L = Lfix if mode == 0 else int(rng.integers(1, min(120, n) + 1))
        r = rng.random()
        if r < 0.55 and n > L:
            p = int(rng.integers(0, n - L))
            q = text[p:p + L].copy()
            if r < 0.15:
                q[int(rng.integers(0, L))] = letters[int(rng.integers(0, len(letters)))]
        else:
            q = letters[rng.integers(0, len(letters), size=L)]
        if (q == ord("$")).any() or (mode != 2 and not only_aa and not np.isin(q, letters).all()):
            q = letters[rng.integers(0, len(letters), size=L)]
        if L > 3 and not np.isin(q, letters).any():  # a run of N / X matches every window of a long N run: the GPU is fine with
            q = letters[rng.integers(0, len(letters), size=L)]  # that (tests/nheavy_gpu.py), the oracle's walks are not
        b = bytes(q)
        if mode == 2 and rng.random() < 0.1:
            b = b.lower()
        qs.append(b)
    qb, qo = pack_queries(qs)
    woff, wg, wp, _ = oi.parallel_locate(qb, qo, 4)
    for verify in ((-1,) if os.environ.get("AWRY_FUZZ_WIDE") else ((2,) if trial % 3 else (2, -1))):
        ix.set_verify(verify)
        c = ix.parallel_count_csr(qb, qo)
        assert np.array_equal(c, np.diff(woff)), ("count", trial, alphabet, n, mode, verify)
        off, g, p = ix.parallel_locate_csr(qb, qo)
        assert np.array_equal(off, woff) and np.array_equal(g, wg) and np.array_equal(p, wp), ("locate", trial, alphabet, n, mode, verify)
    for q in qs[:8]:  # single-query entry points
        assert ix.count_string(q) == oi.count_string(q)
        assert np.array_equal(ix.locate_string_raw(q)[0], oi.locate_string(q)[0])
        r = ix.search_range(q)
        assert (r.start_ptr, r.end_ptr) == oi.search_range(q)  # the reference's rows, absent queries included
    print("trial %d ok: alphabet %d, n %d, %d queries (mode %d), ratio %d, %.0f s" % (trial, alphabet, n, nq, mode, ratio, time.time() - t0), flush=True)
print("fuzz ok")
