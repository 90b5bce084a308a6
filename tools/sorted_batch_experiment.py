"""How much of the distance between the headline rate and the random-request ceiling of a small table is address translation /
DRAM page locality?  The same 10 M random 31-mers in random order and sorted by their seed-table slot (the sort is outside the
timed region: this measures the memory system, not a schedule anybody could run on unsorted input).
usage: sorted_batch_experiment.py [text_len]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import synth
import awry_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
L, nq = 31, 10_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0).set_devices([0])
k = ix.seed_kmer_len()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(5)
w = torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen)
counts = torch.zeros(nq, dtype=torch.int64, device=dev)
def run(words, name):
    for _ in range(3):
        ix.dev_count_nt2(words.data_ptr(), nq, L, counts.data_ptr(), True, stream, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ix.dev_count_nt2(words.data_ptr(), nq, L, counts.data_ptr(), True, stream, 0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print("%-60s %.4f ms  %.2f G q/s" % (name, ms, nq / ms / 1e6), flush=True)
    return counts.clone()
c0 = run(w, "random order (seed k = %d)" % k)
slot = (w >> (2 * (L - k))) & ((1 << (2 * k)) - 1)
order = torch.argsort(slot)
c1 = run(w[order].contiguous(), "sorted by seed slot")
assert torch.equal(c1, c0[order])
for bits in (4, 8, 12):
    o = torch.argsort(slot >> (2 * k - bits), stable=True)
    run(w[o].contiguous(), "bucketed by the top %d bits of the slot" % bits)
