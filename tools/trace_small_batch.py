"""AWRY_TRACE_HOST breakdown of awry_count_batch for small batches (chr1 scale).  usage: trace_small_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AWRY_TRACE_HOST"] = "1"
import numpy as np
import awry_amd
from tests import synth
text, st, hd = synth.make_text(100_000_000, 0, 7, 1, 0.02)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
for m in (1000, 4000, 10_000, 50_000, 200_000):
    qb, qo = synth.fixed_to_csr(synth.random_queries(m, 31, 0, 5))
    out = np.zeros(m, dtype=np.uint64)
    for _ in range(3): ix.parallel_count_csr(qb, qo, out)
    sys.stderr.write("---- %d queries\n" % m); sys.stderr.flush()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); ix.parallel_count_csr(qb, qo, out); ts.append((time.perf_counter() - t) * 1e6)
    sys.stderr.write("python-side: %s us\n" % ", ".join("%.0f" % x for x in ts)); sys.stderr.flush()
