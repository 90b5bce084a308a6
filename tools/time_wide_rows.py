"""the 64-bit-row kernels (indexes of 2^32 rows and more) timed on a GRCh38-scale index forced onto them
(awry_debug_force_wide_rows): random and present 31-mers, 101-bp reads, locate.  usage: time_wide_rows.py [text_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import synth
import awry_amd
import bench
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
L, nq = 31, 10_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0)
awry_amd.load_library().awry_debug_force_wide_rows(1)
t = time.time(); ix.set_devices([0]); print("set_devices (wide) %.1f s, seed k = %d" % (time.time() - t, ix.seed_kmer_len()), flush=True)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(5)
counts = torch.zeros(nq, dtype=torch.int64, device=dev)
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ws = [torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen) for _ in range(4)]
i = [0]
def step():
    ix.dev_count_nt2(ws[i[0] % 4].data_ptr(), nq, L, counts.data_ptr(), True, stream, 0); i[0] += 1
ms = timed(step); print("random 31-mers: %.3f ms, %.2f G q/s  (schedule %s)" % (ms, nq / ms / 1e6, ix.count_schedule(L)), flush=True)
def step0():
    ix.dev_count_nt2(ws[i[0] % 4].data_ptr(), nq, L, counts.data_ptr(), False, stream, 0); i[0] += 1
ms = timed(step0, 4); print("random 31-mers, no table: %.3f ms, %.2f G q/s" % (ms, nq / ms / 1e6), flush=True)
d_text = torch.from_numpy(text).to(dev)
for Lr, nr in ((31, 2_000_000), (101, 5_000_000)):
    reads = bench.device_sampled_reads(torch, d_text, nr, Lr, 99, ord("N"))
    W = (Lr + 31) // 32
    words = torch.zeros(nr * W, dtype=torch.int64, device=dev); bad = torch.zeros(1, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(reads.data_ptr(), nr, Lr, words.data_ptr(), bad.data_ptr(), stream, 0)
    c = torch.zeros(nr, dtype=torch.int64, device=dev)
    if Lr <= 32:
        f = lambda: ix.dev_count_nt2(words.data_ptr(), nr, Lr, c.data_ptr(), True, stream, 0)
    else:
        f = lambda: ix.dev_count_nt2_long(words.data_ptr(), nr, Lr, c.data_ptr(), None, True, stream, 0)
    ms = timed(f, 4); print("%d-bp from the text: %.3f ms, %.3f G q/s, all found: %s" % (Lr, ms, nr / ms / 1e6, bool((c >= 1).all())), flush=True)
