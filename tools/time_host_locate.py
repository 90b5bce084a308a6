"""end-to-end throughput of the host locate boundary (ASCII reads in host memory -> positions in host memory).
usage: time_host_locate.py [text_len] [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_956_422
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
for L in (101, 31):
    q2d = synth.sampled_queries(text, nq, L, 5)
    qb, qo = synth.fixed_to_csr(q2d)
    for mode, name in ((2, "seed-and-verify (default policy)"), (-1, "LF steps + walks to the file's samples")):
        ix.set_verify(mode)
        if mode < 0:
            ix.set_locate_sa_ratio(0)
        for rep in range(3):
            t = time.perf_counter(); off, g, p = ix.parallel_locate_csr(qb, qo); dt = time.perf_counter() - t
        print("L=%d %s: %d reads, %d hits in %.1f ms -> %.1f M reads/s end-to-end (PCIe-inclusive)" % (L, name, nq, len(g), dt * 1e3, nq / dt / 1e6), flush=True)
# reads with ambiguity letters: one N in 0.5 % of the 101-bp reads (redone on the device by the generic kernel)
ix.set_verify(2)
q2d = synth.sampled_queries(text, nq, 101, 5)
sel = np.random.default_rng(3).random(nq) < 0.005
q2d[sel, 50] = ord("N")
qb, qo = synth.fixed_to_csr(q2d)
for rep in range(3):
    t = time.perf_counter(); off, g, p = ix.parallel_locate_csr(qb, qo); dt = time.perf_counter() - t
print("L=101 with one N in 0.5 %% of the reads (%d): %d hits in %.1f ms -> %.1f M reads/s" % (int(sel.sum()), len(g), dt * 1e3, nq / dt / 1e6), flush=True)
