"""Swiss-Prot-scale amino index: parallel_locate for 12-mers and 30-residue peptides drawn from the text -- device-resident
stages (count with range words -> scan -> locate) and the host boundary.  usage: time_amino_locate.py [text_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import awry_amd
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 90_000_000
text, st, hd = synth.make_text(n, 1, 0xA5A50004, 250_000 if n > 1e7 else 50, 0.0)
ix = awry_amd.FmIndex.from_text(text, 1, 8, 0, st, hd).set_devices([0])
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=4):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
m = 4_000_000
for L in (12, 30):
    q2d = synth.sampled_queries(text, m, L, 4, False, 1)
    qb, qo = synth.fixed_to_csr(q2d)
    d_q = torch.from_numpy(np.concatenate([qb, np.zeros(16, dtype=np.uint8)])).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    d_c = torch.zeros(m, dtype=torch.int64, device=dev); d_r = torch.zeros(2 * m, dtype=torch.int64, device=dev)
    d_ho = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    d_sc = torch.zeros(ix.dev_scan_scratch_bytes(m) // 8 + 8, dtype=torch.int64, device=dev)
    t_count = timed(lambda: ix.dev_count_ascii_for_locate(d_q.data_ptr(), d_off.data_ptr(), m, d_c.data_ptr(), d_r.data_ptr(), None, stream, 0))
    t_scan = timed(lambda: ix.dev_scan_counts(d_c.data_ptr(), m, d_ho.data_ptr(), d_sc.data_ptr(), stream, 0))
    total = int(d_ho[m].item())
    d_g = torch.zeros(total, dtype=torch.int64, device=dev); d_p = torch.zeros(2 * total, dtype=torch.int64, device=dev)
    t_loc = timed(lambda: ix.dev_locate(d_r.data_ptr(), d_ho.data_ptr(), m, total, d_g.data_ptr(), d_p.data_ptr(), stream, 0))
    print("L = %d: %d queries, %d hits: count + locate words %.3f ms (%.2f G/s), scan %.3f ms, locate %.3f ms (%.2f G hits/s); all stages %.2f G queries/s"
          % (L, m, total, t_count, m / t_count / 1e6, t_scan, t_loc, total / t_loc / 1e6, m / (t_count + t_scan + t_loc) / 1e6), flush=True)
    for rep in range(3):
        t = time.perf_counter(); off, gpos, pos = ix.parallel_locate_csr(qb, qo); dt = time.perf_counter() - t
    assert int(off[-1]) == total
    print("   host boundary (awry_locate_batch): %.1f ms -> %.3f G queries/s, %.3f G hits/s" % (dt * 1e3, m / dt / 1e9, total / dt / 1e9), flush=True)
    for rep in range(3):
        t = time.perf_counter(); c = ix.parallel_count_csr(qb, qo); dt = time.perf_counter() - t
    print("   host boundary (awry_count_batch): %.1f ms -> %.3f G queries/s" % (dt * 1e3, m / dt / 1e9), flush=True)
