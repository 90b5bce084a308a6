"""device-resident throughput of the generic ASCII count kernel (ragged lengths) next to the packed kernels.
usage: time_generic_kernel.py [text_len]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_956_422
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
nq = 4_000_000
for name, lo, hi, present in (("random 20..39", 20, 40, False), ("present 80..150", 80, 151, True), ("random 31", 31, 32, False), ("present 101", 101, 102, True)):
    rng = np.random.default_rng(1)
    lens = rng.integers(lo, hi, size=nq)
    qo = np.zeros(nq + 1, dtype=np.uint64); qo[1:] = np.cumsum(lens)
    if present:
        starts = rng.integers(0, n - 200, size=nq)
        idx = np.repeat(starts, lens) + (np.arange(int(qo[-1])) - np.repeat(qo[:-1].astype(np.int64), lens))
        qb = text[idx]
        qb[~np.isin(qb, synth.NT)] = ord("A")
    else:
        qb = synth.NT[rng.integers(0, 4, size=int(qo[-1]), dtype=np.uint8)]
    d_q = torch.from_numpy(qb).to(dev); d_o = torch.from_numpy(qo.view(np.int64)).to(dev)
    d_c = torch.zeros(nq, dtype=torch.int64, device=dev)
    for _ in range(2):
        ix.dev_count_ascii(d_q.data_ptr(), d_o.data_ptr(), nq, d_c.data_ptr(), None, None, stream, 0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        ix.dev_count_ascii(d_q.data_ptr(), d_o.data_ptr(), nq, d_c.data_ptr(), None, None, stream, 0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print("%s: generic ASCII kernel %.2f ms -> %.2f G queries/s (%.1f GB/s of query bytes)" % (name, ms, nq / ms / 1e6, qb.nbytes / ms / 1e6), flush=True)
