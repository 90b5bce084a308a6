"""long packed reads: the two-phase schedule (per-lane probe + listed pass) against the single quad kernel, by read length
(GRCh38 scale, device-resident) -- where the hand-over between them belongs.  usage: time_read_schedules.py [text_len]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import awry_amd, bench
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0).set_devices([0])
lib = awry_amd.load_library()
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
d_text = torch.from_numpy(text).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(3)
def timed(fn):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 4
for L in (101, 150, 200, 250, 300, 400, 512):
    m = 2_000_000
    W = (L + 31) // 32
    out = []
    for name in ("random", "from the text"):
        if name == "random":
            ascii_ = torch.from_numpy(np.frombuffer(b"ACGT", np.uint8).copy()).to(dev)[torch.randint(0, 4, (m, L), device=dev, generator=gen)]
        else:
            ascii_ = bench.device_sampled_reads(torch, d_text, m, L, 7, ord("N"))
        words = torch.zeros(m * W, dtype=torch.int64, device=dev); bad = torch.zeros(1, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(ascii_.contiguous().data_ptr(), m, L, words.data_ptr(), bad.data_ptr(), stream, 0)
        c = torch.zeros(m, dtype=torch.int64, device=dev)
        f = lambda: ix.dev_count_nt2_long(words.data_ptr(), m, L, c.data_ptr(), None, True, stream, 0)
        lib.awry_debug_set_count_kernel(-1); two = timed(f); c2 = c.clone()
        lib.awry_debug_set_count_kernel(2); one = timed(f)
        lib.awry_debug_set_count_kernel(-1)
        assert torch.equal(c, c2)
        out.append("%s: two-phase %.2f, single %.2f G/s" % (name, m / two / 1e6, m / one / 1e6))
        del ascii_, words
    print("L = %3d  %s" % (L, "   ".join(out)), flush=True)
