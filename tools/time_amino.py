"""config[3] shape: Swiss-Prot-scale amino index (9e7 aa, 2.5e5 records), 10M 12-mers, device-resident: the generic kernel
and the two-phase amino k-mer schedule (awry_dev_count_ascii_uniform); then the host boundary"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
from tests import synth
n, nq, L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 90_000_000, 10_000_000, 12
t = time.time(); text, st, hd = synth.make_text(n, 1, 0xA5A50004, 250_000 if n > 1e7 else 50, 0.0); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); ix = awry_amd.FmIndex.from_text(text, 1, 8, 0, st, hd).set_devices([0]); print("build+replicate %.1fs" % (time.time() - t), flush=True)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
for name, q2d in (("random", synth.random_queries(nq, L, 1, 3)), ("present", synth.sampled_queries(text, nq // 4, L, 4, False, 1))):
    m = len(q2d)
    d_q = torch.from_numpy(np.concatenate([q2d.reshape(-1), np.zeros(16, dtype=np.uint8)])).to(dev)
    d_off = (torch.arange(m + 1, dtype=torch.int64, device=dev) * L)
    res = {}
    for kind in ("generic", "two-phase"):
        d_c = torch.zeros(m, dtype=torch.int64, device=dev)
        if kind == "generic":
            fn = lambda: ix.dev_count_ascii(d_q.data_ptr(), d_off.data_ptr(), m, d_c.data_ptr(), None, None, stream, 0)
        else:
            fn = lambda: ix.dev_count_ascii_uniform(d_q.data_ptr(), m, L, d_c.data_ptr(), None, stream, 0)
        for _ in range(2): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        res[kind] = d_c
        print("amino %s %d-mers, %s: %.2f ms per %d queries -> %.2f G queries/s; mean count %.3f" % (name, L, kind, ms, m, m / ms / 1e6, float(d_c.float().mean())), flush=True)
    assert torch.equal(res["generic"], res["two-phase"])
# the host boundary: ASCII 12-mers in host memory -> counts in host memory (PCIe-inclusive)
q2d = synth.random_queries(nq, L, 1, 3)
qb, qo = synth.fixed_to_csr(q2d)
for rep in range(3):
    t = time.perf_counter(); c = ix.parallel_count_csr(qb, qo); dt = time.perf_counter() - t
print("amino host boundary: %d %d-mers in %.1f ms -> %.1f M queries/s end-to-end" % (nq, L, dt * 1e3, nq / dt / 1e6), flush=True)
