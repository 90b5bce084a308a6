"""What makes the FIRST awry_count_batch of a process slow when it is not the first thing the process does?  bench.py sees
20+ ms ("waiting for the GPU 16 ms" under AWRY_TRACE_HOST=1) where tools/trace_first_call.py, which calls right after
awry_set_devices, sees 2 ms.  One variant per process (a process has only one first call):
  0  call right after set_devices (the tool's case)
  1  torch initialised, 10 M-query device-resident counts on torch's stream first (what bench.py does for minutes)
  2  as 1, then accelerators dropped and rebuilt (set_verify(-1) -> set_verify(2)), as bench.py's LF-only variants do
  3  20 s of idleness between set_devices and the call
  4  as 1, with the host query bytes produced by torch (.cpu().numpy()) instead of numpy
  5  as 4, then the bytes copied once more by numpy into memory the runtime never saw (is it that region of memory?)
  6  as 4, then a device synchronise and 2 s of sleep before the call (is it work the runtime deferred?)
usage: first_call_triggers.py <variant> [text_len] [n_queries]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth

variant = int(sys.argv[1])
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 248_956_422
nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 5_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd)
ix.set_devices([0])
q31 = synth.random_queries(nq, 31, 0, 5)
qb, qo = synth.fixed_to_csr(q31)
if variant in (1, 2, 4, 5, 6):
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    words = torch.randint(0, 2**62, (10_000_000,), dtype=torch.int64, device=dev)
    counts = torch.zeros(10_000_000, dtype=torch.int64, device=dev)
    for _ in range(50):
        ix.dev_count_nt2(words.data_ptr(), 10_000_000, 31, counts.data_ptr(), True, stream, 0)
    torch.cuda.synchronize()
    if variant == 2:
        ix.set_verify(-1)
        for _ in range(5):
            ix.dev_count_nt2(words.data_ptr(), 10_000_000, 31, counts.data_ptr(), True, stream, 0)
        torch.cuda.synchronize()
        ix.set_verify(2)
        for _ in range(5):
            ix.dev_count_nt2(words.data_ptr(), 10_000_000, 31, counts.data_ptr(), True, stream, 0)
        torch.cuda.synchronize()
    if variant in (4, 5, 6):
        qb = torch.from_numpy(qb).to(dev).cpu().numpy()
    if variant == 5:
        qb = qb.copy()
    if variant == 6:
        torch.cuda.synchronize()
        time.sleep(2)
if variant == 3:
    time.sleep(20)
out = np.ones(nq, dtype=np.uint64)
for i in range(4):
    t = time.perf_counter()
    ix.parallel_count_csr(qb, qo, out)
    print("variant %d count call %d: %.2f ms" % (variant, i + 1, (time.perf_counter() - t) * 1e3), flush=True)
