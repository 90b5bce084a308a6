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
