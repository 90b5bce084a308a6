"""Swiss-Prot-scale amino index: device-resident count rate by query length (uniform and unequal lengths, random and drawn from
the text) through the default dispatch -- where the k-mer schedule ends (24 residues) and the generic kernel takes over.
usage: time_amino_lengths.py [text_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import awry_amd
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 90_000_000
text, st, hd = synth.make_text(n, 1, 0xA5A50004, 250_000 if n > 1e7 else 50, 0.0)
ix = awry_amd.FmIndex.from_text(text, 1, 8, 0, st, hd).set_devices([0])
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
def timed(fn):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 4
m = 4_000_000
for L in (8, 12, 16, 20, 24, 25, 28, 32, 40, 60):
    row = []
    for name, q2d in (("random", synth.random_queries(m, L, 1, 3)), ("from the text", synth.sampled_queries(text, m, L, 4, False, 1))):
        d_q = torch.from_numpy(np.concatenate([q2d.reshape(-1), np.zeros(16, dtype=np.uint8)])).to(dev)
        d_c = torch.zeros(m, dtype=torch.int64, device=dev)
        ms = timed(lambda: ix.dev_count_ascii_uniform(d_q.data_ptr(), m, L, d_c.data_ptr(), None, stream, 0))
        row.append("%s %.2f G/s" % (name, m / ms / 1e6))
    print("L = %2d uniform: %s" % (L, ", ".join(row)), flush=True)
rng = np.random.default_rng(1)
for lo, hi in ((8, 24), (8, 40), (20, 60)):
    lens = rng.integers(lo, hi + 1, size=m)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    starts = rng.integers(0, n - hi - 1, size=m)
    idx = np.repeat(starts - off[:-1], lens) + np.arange(off[-1])
    qb = text[idx]
    d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_c = torch.zeros(m, dtype=torch.int64, device=dev)
    ms = timed(lambda: ix.dev_count_ascii(d_q.data_ptr(), d_off.data_ptr(), m, d_c.data_ptr(), None, None, stream, 0))
    print("lengths %d..%d from the text (windows may cross records): %.2f G/s, %.1f %% found" % (lo, hi, m / ms / 1e6, 100.0 * float((d_c > 0).float().mean())), flush=True)
