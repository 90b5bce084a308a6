"""A/B of the seed-and-verify threshold (LF steps before switching to text comparison) on present, random and mixed
batches, k-mers (L=31) and reads (L=101).  usage: ab_verify.py [text_len] [after,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
from tests import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
afters = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1, 0, 1, 2, 3]
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
print("index ready: seed k=%d" % ix.seed_kmer_len(), flush=True)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def pack(q2d):
    nq, L = q2d.shape
    W = (L + 31) // 32
    d_ascii = torch.from_numpy(q2d.reshape(-1)).to(dev)
    d_words = torch.zeros(nq * W, dtype=torch.int64, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(d_ascii.data_ptr(), nq, L, d_words.data_ptr(), d_bad.data_ptr(), stream, 0)
    torch.cuda.synchronize()
    assert int(d_bad.item()) == 0
    return d_words


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


cases = []
for L, nq in ((31, 4_000_000), (101, 2_000_000)):
    pres = synth.sampled_queries(text, nq, L, 77)
    rnd = synth.random_queries(nq, L, 0, 5)
    mix = pres.copy()
    mix[::2] = rnd[::2]
    # present reads with one substitution in the middle (absent, but long exact stretches)
    mut = pres.copy()
    col = mut[:, L // 2]
    mut[:, L // 2] = np.where(col == ord("A"), ord("C"), ord("A")).astype(np.uint8)
    for name, q in (("present", pres), ("random", rnd), ("mixed", mix), ("1-subst", mut)):
        cases.append((L, name, nq, pack(q)))
want = {}
for a in afters:
    ix.set_verify(a)
    row = []
    for L, name, nq, d_words in cases:
        counts = torch.zeros(nq, dtype=torch.int64, device=dev)
        if L <= 32:
            fn = lambda: ix.dev_count_nt2(d_words.data_ptr(), nq, L, counts.data_ptr(), True, stream, 0)
        else:
            fn = lambda: ix.dev_count_nt2_long(d_words.data_ptr(), nq, L, counts.data_ptr(), None, True, stream, 0)
        ms = timed(fn)
        key = (L, name)
        if key in want:
            assert torch.equal(want[key], counts), "counts changed with verify_after=%d on %s" % (a, key)
        else:
            want[key] = counts.clone()
        row.append("L%d %s %.2f" % (L, name, nq / ms / 1e6))
    print("after=%2d  G queries/s: " % a + " | ".join(row), flush=True)
