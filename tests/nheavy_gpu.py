"""GPU-only check (no oracle: its LF walks through N runs take hours here) of a batch that an N-rich text makes huge:
30 % N in long runs, every eighth query a run of N -- 6e8 hits.  Counts against a sliding-window count, located
positions against the text."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import awry_amd
from tests import synth
text, st, hd = synth.make_text(200000, 0, 424242, 6, 0.3)
ix = awry_amd.FmIndex.from_text(text, 0, 3, 0, st, hd).set_devices([0])
rng = np.random.default_rng(1)
qs = []
for i in range(30000):
    L = int(rng.integers(1, 118))
    if i % 8 == 0:
        qs.append(b"N" * L)
    else:
        p = int(rng.integers(0, 200000 - L)); qs.append(bytes(text[p:p + L]).replace(b"$", b"A"))
qb, qo = awry_amd.fm_index.pack_queries(qs)
isn = (text == ord("N")) | ~np.isin(text, np.frombuffer(b"ACGT$", np.uint8))
for verify in (2, -1):
    ix.set_verify(verify)
    t = time.time(); c = ix.parallel_count_csr(qb, qo); t1 = time.time() - t
    t = time.time(); off, g, p = ix.parallel_locate_csr(qb, qo); t2 = time.time() - t
    assert np.array_equal(np.diff(off), c)
    print("verify", verify, "count %.2f s, locate %.2f s, hits %d" % (t1, t2, len(g)), flush=True)
    # every located position holds its query (N in the query matches any non-ACGT text symbol)
    chk = rng.integers(0, len(g), size=20000)
    qi = np.searchsorted(off, chk, side="right") - 1
    for h, q in zip(chk[:3000], qi[:3000]):
        L = int(qo[q + 1] - qo[q]); pos = int(g[h]); w = text[pos:pos + L]; qq = qb[int(qo[q]):int(qo[q + 1])]
        assert len(w) == L and all((a == b) or (b == ord("N") and a not in b"ACGT$") for a, b in zip(w, qq)), (q, pos)
    # counts of the all-N queries equal the number of windows made of N-like symbols only
    cs = np.concatenate([[0], np.cumsum(isn.astype(np.int64))])
    for q in range(0, 30000, 8 * 37):
        L = int(qo[q + 1] - qo[q]); want = int(((cs[L:] - cs[:-L]) == L).sum())
        assert int(c[q]) == want, (q, L, int(c[q]), want)
print("ok")
