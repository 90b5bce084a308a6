"""end-to-end throughput of the host boundary (ASCII queries in host memory -> counts in host memory).
usage: time_host_api.py [text_len] [n_queries]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_956_422
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
for L in (31, 101):
    q2d = synth.random_queries(nq if L == 31 else nq // 4, L, 0, 5)
    qb, qo = synth.fixed_to_csr(q2d)
    for rep in range(3):
        t = time.perf_counter(); c = ix.parallel_count_csr(qb, qo); dt = time.perf_counter() - t
    print("fixed L=%d: %d queries in %.1f ms -> %.1f M queries/s end-to-end (PCIe-inclusive, %.2f GB/s in)" % (L, len(q2d), dt * 1e3, len(q2d) / dt / 1e6, qb.nbytes / dt / 1e9), flush=True)
    os.environ["X"] = "1"
# packed k-mers from host memory: 16 B per query cross PCIe
q2d = synth.random_queries(nq, 31, 0, 5)
code = np.searchsorted(synth.NT, q2d).astype(np.uint64)
words = np.zeros(nq, dtype=np.uint64)
for j in range(31):
    words |= code[:, j] << np.uint64(2 * j)
for rep in range(3):
    t = time.perf_counter(); c = ix.parallel_count_packed(words, 31); dt = time.perf_counter() - t
print("packed 31-mers: %d queries in %.1f ms -> %.1f M queries/s end-to-end (PCIe-inclusive)" % (nq, dt * 1e3, nq / dt / 1e6), flush=True)
# generic path: ragged lengths
lens = np.random.default_rng(1).integers(20, 40, size=nq // 4)
qo = np.zeros(len(lens) + 1, dtype=np.uint64); qo[1:] = np.cumsum(lens)
qb = synth.NT[np.random.default_rng(2).integers(0, 4, size=int(qo[-1]), dtype=np.uint8)]
for rep in range(2):
    t = time.perf_counter(); c = ix.parallel_count_csr(qb, qo); dt = time.perf_counter() - t
print("ragged 20..39: %d queries in %.1f ms -> %.1f M queries/s (ragged packed path)" % (len(lens), dt * 1e3, len(lens) / dt / 1e6))
# reads with ambiguity letters: 0.5 % of 101-bp reads from the text get one N (the per-read fallback of the packed path)
reads = synth.sampled_queries(text, nq // 4, 101, 9)
qb, qo = synth.fixed_to_csr(reads)
for rep in range(3):
    t = time.perf_counter(); c0 = ix.parallel_count_csr(qb, qo); dt0 = time.perf_counter() - t
withn = reads.copy()
sel = np.random.default_rng(3).random(len(withn)) < 0.005
withn[sel, 50] = ord("N")
qb, qo = synth.fixed_to_csr(withn)
for rep in range(3):
    t = time.perf_counter(); c1 = ix.parallel_count_csr(qb, qo); dt1 = time.perf_counter() - t
print("101-bp reads from the text: %d reads in %.1f ms (%.1f M/s); with one N in 0.5 %% of them: %.1f ms (%.1f M/s), %d reads redone" %
      (len(reads), dt0 * 1e3, len(reads) / dt0 / 1e6, dt1 * 1e3, len(reads) / dt1 / 1e6, int(sel.sum())), flush=True)
