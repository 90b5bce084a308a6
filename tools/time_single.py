"""latency of the scalar conveniences (count_string / locate_string / search_range), one query per call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth
text, st, hd = synth.make_text(20_000_000, 0, 5, 1, 0.02)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
qs = [bytes(q) for q in synth.sampled_queries(text, 2000, 31, 3)]
for name, fn in (("count_string", ix.count_string), ("locate_string", ix.locate_string), ("search_range", ix.search_range)):
    for q in qs[:50]:
        fn(q)
    t = time.perf_counter()
    for q in qs:
        fn(q)
    dt = time.perf_counter() - t
    print("%s: %.1f us per call" % (name, dt / len(qs) * 1e6), flush=True)
# small batches through parallel_count / parallel_locate (host CSR in, host arrays out)
from awry_amd.fm_index import pack_queries
for nq in (2, 16, 128, 1024, 4095, 4096, 20000):
    qb, qo = pack_queries([bytes(q) for q in synth.sampled_queries(text, nq, 31, 7)])
    for name, fn in (("parallel_count", ix.parallel_count_csr), ("parallel_locate", ix.parallel_locate_csr)):
        for _ in range(3):
            fn(qb, qo)
        t = time.perf_counter()
        for _ in range(20):
            fn(qb, qo)
        dt = (time.perf_counter() - t) / 20
        print("%s, %d 31-mers per call: %.0f us per call" % (name, nq, dt * 1e6), flush=True)
