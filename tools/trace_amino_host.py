import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth
text, st, hd = synth.make_text(90_000_000, 1, 0xA5A50004, 250_000, 0.0)
ix = awry_amd.FmIndex.from_text(text, 1, 8, 0, st, hd).set_devices([0])
q2d = synth.random_queries(10_000_000, 12, 1, 3)
qb, qo = synth.fixed_to_csr(q2d)
for rep in range(4):
    t = time.perf_counter(); c = ix.parallel_count_csr(qb, qo); dt = time.perf_counter() - t
    print("call %.1f ms" % (dt * 1e3), flush=True)
# parallel_locate of 12-mers sampled from the text (equal lengths: the amino k-mer schedule is the count pass)
q2d = synth.sampled_queries(text, 4_000_000, 12, 4, False, 1)
qb, qo = synth.fixed_to_csr(q2d)
for rep in range(4):
    t = time.perf_counter(); off, g, p = ix.parallel_locate_csr(qb, qo); dt = time.perf_counter() - t
    print("locate call %.1f ms, %d hits" % (dt * 1e3, len(g)), flush=True)
