"""throughput of the ASCII -> 2-bit pack kernel alone (device-resident ASCII)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
from tests import synth
text, st, hd = synth.make_text(1_000_000, 0, 5, 1, 0.0)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for L, n in ((31, 20_000_000), (101, 5_000_000), (250, 2_000_000)):
    W = (L + 31) // 32
    asc = torch.from_numpy(synth.random_queries(n, L, 0, 3).reshape(-1)).to(dev)
    words = torch.zeros(n * W, dtype=torch.int64, device=dev)
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(2):
        ix.dev_pack_nt2(asc.data_ptr(), n, L, words.data_ptr(), bad.data_ptr(), stream, 0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ix.dev_pack_nt2(asc.data_ptr(), n, L, words.data_ptr(), bad.data_ptr(), stream, 0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("L=%d: %.3f ms per %d queries -> %.1f G queries/s, %.0f GB/s of ASCII" % (L, ms, n, n / ms / 1e6, n * L / ms / 1e6), flush=True)
