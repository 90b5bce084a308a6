"""A/B of the packed-k-mer count kernel variants on one index.  usage: ab_count.py [text_len] [seed_k,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
from awry_amd import _lib
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [14, 16]
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
L, nq = 31, 10_000_000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(7)
batches = [torch.randint(0, 1 << 62, (nq,), dtype=torch.int64, device=dev, generator=gen) for _ in range(4)]
counts = [torch.zeros(nq, dtype=torch.int64, device=dev) for _ in range(4)]
stream = torch.cuda.current_stream().cuda_stream
lib = _lib.load_library()
for k in ks:
    ix.set_seed_kmer_len(k)
    res = {}
    for mode, name in ((0, "strided"), (2, "quad4"), (3, "twophase")):
        lib.awry_debug_set_count_kernel(mode)
        for i in range(2): ix.dev_count_nt2(batches[i].data_ptr(), nq, L, counts[mode].data_ptr(), True, stream, 0)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(8): ix.dev_count_nt2(batches[i % 4].data_ptr(), nq, L, counts[mode].data_ptr(), True, stream, 0)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 8)
        res[name] = best
    same = bool(torch.equal(counts[0], counts[2]) and torch.equal(counts[0], counts[3]))
    print("k=%d  " % k + "  ".join("%s %.3f ms (%.2f Gq/s)" % (nm, ms, nq / ms / 1e6) for nm, ms in res.items()) + "  identical=%s" % same, flush=True)
