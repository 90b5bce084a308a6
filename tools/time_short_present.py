"""k-mers of 18..31 letters drawn from the text (GRCh38 scale, device-resident): where small seed ranges are verified in the
probe pass or handed to the resume pass.  usage: time_short_present.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import awry_amd, bench
from tests import synth
n = 3_100_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0).set_devices([0])
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
d_text = torch.from_numpy(text).to(dev)
def timed(fn):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(6): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 6
m = 4_000_000
out = []
for L in (18, 19, 20, 21, 22, 23, 25, 28, 31):
    ascii_ = bench.device_sampled_reads(torch, d_text, m, L, 7, ord("N"))
    words = torch.zeros(m, dtype=torch.int64, device=dev); bad = torch.zeros(1, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(ascii_.contiguous().data_ptr(), m, L, words.data_ptr(), bad.data_ptr(), stream, 0)
    c = torch.zeros(m, dtype=torch.int64, device=dev)
    ms = timed(lambda: ix.dev_count_nt2(words.data_ptr(), m, L, c.data_ptr(), True, stream, 0))
    out.append("L=%d %.1f" % (L, m / ms / 1e6))
print("from the text, G/s: " + "  ".join(out), flush=True)
