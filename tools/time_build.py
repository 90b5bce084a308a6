"""times host vs GPU index construction at a given size; usage: time_build.py N [gpu|host|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import synth
from awry_amd import FmIndex
n = int(float(sys.argv[1])); mode = sys.argv[2] if len(sys.argv) > 2 else "both"
os.environ["AWRY_VERBOSE"] = "1"
t = time.time(); text, st, hd = synth.make_text(n, 0, 42, 1 if n < 1e9 else 25, 0.07 if n < 1e9 else 0.05); print("gen %.1fs" % (time.time() - t), flush=True)
g = h = None
if mode in ("gpu", "both"):
    t = time.time(); g = FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0); print("gpu build %.1fs" % (time.time() - t), flush=True)
if mode in ("host", "both"):
    t = time.time(); h = FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=-1); print("host build %.1fs" % (time.time() - t), flush=True)
if g is not None and h is not None:
    print("identical:", np.array_equal(g.device_block_words(), h.device_block_words()) and np.array_equal(g.sa_words(), h.sa_words())
          and np.array_equal(g.prefix_sums(), h.prefix_sums()) and g.sentinel_row() == h.sentinel_row())
