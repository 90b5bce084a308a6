"""GRCh38-scale nucleotide index: device-resident count rate by query length (packed words resident), random and drawn from the
text, through the default dispatch -- where the schedules change hands.  usage: time_nt_lengths.py [text_len]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import awry_amd, bench
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0).set_devices([0])
print("seed k = %d" % ix.seed_kmer_len(), flush=True)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
d_text = torch.from_numpy(text).to(dev)
def timed(fn):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 4
gen = torch.Generator(device=dev); gen.manual_seed(3)
for L in (8, 12, 16, 17, 18, 20, 21, 25, 31, 32, 33, 50, 64, 101, 150, 250, 512, 513, 1000, 4096):
    m = 4_000_000 if L <= 150 else (1_000_000 if L <= 1000 else 200_000)
    W = (L + 31) // 32
    row = []
    for name in ("random", "from the text"):
        if name == "random":
            ascii_ = torch.from_numpy(np.frombuffer(b"ACGT", np.uint8).copy()).to(dev)[torch.randint(0, 4, (m, L), device=dev, generator=gen)]
        else:
            ascii_ = bench.device_sampled_reads(torch, d_text, m, L, 7, ord("N"))
        words = torch.zeros(m * W, dtype=torch.int64, device=dev); bad = torch.zeros(1, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(ascii_.contiguous().data_ptr(), m, L, words.data_ptr(), bad.data_ptr(), stream, 0)
        c = torch.zeros(m, dtype=torch.int64, device=dev)
        if L <= 32: f = lambda: ix.dev_count_nt2(words.data_ptr(), m, L, c.data_ptr(), True, stream, 0)
        else: f = lambda: ix.dev_count_nt2_long(words.data_ptr(), m, L, c.data_ptr(), None, True, stream, 0)
        ms = timed(f)
        row.append("%s %.2f G/s" % (name, m / ms / 1e6))
        del ascii_, words
    print("L = %4d: %s   (%s)" % (L, ", ".join(row), ix.count_schedule(L)), flush=True)
