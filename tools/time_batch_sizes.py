"""rate and latency by batch size (GRCh38 scale): device-resident packed 31-mers, and the host boundary.
usage: time_batch_sizes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import awry_amd
from tests import synth
n = 3_100_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0).set_devices([0])
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(1)
L = 31
for m in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
    w = torch.randint(0, 1 << (2 * L), (m,), dtype=torch.int64, device=dev, generator=gen)
    c = torch.zeros(m, dtype=torch.int64, device=dev)
    f = lambda: ix.dev_count_nt2(w.data_ptr(), m, L, c.data_ptr(), True, stream, 0)
    for _ in range(5): f()
    torch.cuda.synchronize()
    reps = 200 if m <= 100_000 else 20
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    # one call + synchronise: what a caller that needs the answer before going on sees
    t = time.perf_counter()
    for _ in range(50): f(); torch.cuda.synchronize()
    lat = (time.perf_counter() - t) / 50 * 1e6
    q2d = synth.random_queries(m, L, 0, 5)
    qb, qo = synth.fixed_to_csr(q2d)
    out = np.zeros(m, dtype=np.uint64)
    for _ in range(3): ix.parallel_count_csr(qb, qo, out)
    t = time.perf_counter()
    hr = 30 if m <= 1_000_000 else 5
    for _ in range(hr): ix.parallel_count_csr(qb, qo, out)
    hus = (time.perf_counter() - t) / hr * 1e6
    print("%9d queries: device back to back %.1f us (%.2f G q/s), call + sync %.1f us; host boundary %.1f us per call (%.3f G q/s)"
          % (m, us, m / us / 1e3, lat, hus, m / hus / 1e3), flush=True)
