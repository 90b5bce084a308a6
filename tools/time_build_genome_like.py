"""GPU index construction on a text shaped like an assembled chromosome rather than i.i.d. letters: megabase runs of N
(centromere / telomere gaps), a tandem satellite array with a little divergence, an exact tandem array, and exact
segmental duplications -- the inputs on which prefix doubling needs many rounds.  Prints the rounds and the time, and checks
counts / locations of queries drawn from the repeats against the text itself.
usage: time_build_genome_like.py [N=250e6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import synth
from awry_amd import FmIndex

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 250_000_000
os.environ["AWRY_VERBOSE"] = "1"
t0 = time.time()
text, reg = synth.genome_like_text(n)
gap, tel, sat0, ex0 = reg["gap"], reg["tel"], reg["sat0"], reg["ex0"]
print("text %.1fs" % (time.time() - t0), flush=True)

t0 = time.time()
ix = FmIndex.from_text(text.tobytes(), 0, 8, 0, [0], ["chrS"], build_device=0)
print("gpu build %.1fs (wall, with the host halves)" % (time.time() - t0), flush=True)
ix.set_devices([0])

def check(name, starts, L):
    q = np.stack([text[s: s + L] for s in starts])
    qb = q.reshape(-1).copy(); qo = (np.arange(len(q) + 1) * L).astype(np.uint64)
    counts = ix.parallel_count_csr(qb, qo)
    off, gpos, pos = ix.parallel_locate_csr(qb, qo)
    assert np.array_equal(np.diff(off), counts)
    qi = np.repeat(np.arange(len(q)), counts.astype(np.int64))
    win = text[gpos.astype(np.int64)[:, None] + np.arange(L)[None, :]]
    assert np.array_equal(win, q[qi]), name
    for j, s in enumerate(starts):
        assert (gpos[int(off[j]): int(off[j + 1])] == s).any(), (name, s)
    print("%-22s L=%d  counts min/median/max = %d / %d / %d" % (name, L, counts.min(), int(np.median(counts)), counts.max()), flush=True)

check("segmental duplication", [n // 4 + 1000 * i for i in range(50)], 101)
check("satellite array", [sat0 + 171 * 7 * i + 3 for i in range(50)], 60)
check("exact tandem array", [ex0 + 5 * i for i in range(20)], 74)
qn = np.full(40, ord("N"), np.uint8)                    # letters outside ACGT: the generic kernel; count only (18 M hits)
cn = ix.parallel_count_csr(qn, np.array([0, 40], np.uint64))
assert int(cn[0]) == (gap - 39) + 2 * (tel - 39), int(cn[0])
print("N gap                  L=40  count = %d" % int(cn[0]), flush=True)
print("ok", flush=True)
