"""first-call cost of the host batch entry points: a fresh process, one index, then awry_count_batch and awry_locate_batch
call by call (AWRY_TRACE_HOST=1 prints the library's own per-stage breakdown of each).
usage: trace_first_call.py [text_len] [n_queries] [iid|repeats]      (repeats: tests/synth.repeat_rich_text, several hits per read)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_956_422
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "iid"
if kind == "repeats":
    import torch
    text, st, hd, _ = synth.repeat_rich_text(n, seed=11, n_records=25, device="cuda" if torch.cuda.is_available() else "cpu")
    text = np.asarray(text.cpu().numpy() if hasattr(text, "cpu") else text)
else:
    text, st, hd = synth.make_text(n, 0, 0xA5A50002, 1, 0.05)
t = time.perf_counter()
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd)
t1 = time.perf_counter()
ix.set_devices([0])
print("build %.2f s, set_devices %.2f s (prewarm %s)" % (t1 - t, time.perf_counter() - t1, os.environ.get("AWRY_PREWARM", "1")), flush=True)
q31 = synth.random_queries(nq, 31, 0, 5)
qb, qo = synth.fixed_to_csr(q31)
out = np.zeros(nq, dtype=np.uint64)
out[:] = 1  # touch the caller's result array: its page faults are not the library's start-up cost
for i in range(5):
    t = time.perf_counter(); ix.parallel_count_csr(qb, qo, out); print("count call %d: %.2f ms" % (i + 1, (time.perf_counter() - t) * 1e3), flush=True)
reads = synth.sampled_queries(text, nq, 101, 9)
rb, ro = synth.fixed_to_csr(reads)
for i in range(5):
    t = time.perf_counter(); r = ix.parallel_locate_csr(rb, ro); dt = time.perf_counter() - t
    print("locate call %d: %.2f ms (%d hits, %.0f MB of results)" % (i + 1, dt * 1e3, len(r[1]), (len(r[0]) * 8 + len(r[1]) * 24) / 1e6), flush=True)
    del r
# and back: does a count call that follows locate calls pay anything again?  (callers alternate; the warm-up's locate pass
# was found to leave the next awry_count_batch with a 15-30 ms stall in its first copy -- tools/first_call_ab.sh)
for i in range(3):
    t = time.perf_counter(); ix.parallel_count_csr(qb, qo, out); print("count call after the locate calls %d: %.2f ms" % (i + 1, (time.perf_counter() - t) * 1e3), flush=True)
for i in range(2):
    t = time.perf_counter(); r = ix.parallel_locate_csr(rb, ro); dt = time.perf_counter() - t
    print("locate call after those %d: %.2f ms" % (i + 1, dt * 1e3), flush=True)
    del r
