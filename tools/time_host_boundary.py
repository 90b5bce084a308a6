"""The drop-in boundary itself (parallel_count: ASCII queries in host memory -> counts in host memory, PCIe-inclusive).
usage: time_host_boundary.py [text_len] [n_queries]      (AWRY_TRACE_HOST=1 prints the per-stage breakdown of every call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import awry_amd
from tests import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_956_422
nq = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
print("pool threads:", awry_amd.load_library().awry_host_threads(), "seed k:", ix.seed_kmer_len(), flush=True)


def med(fn, reps=7):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return sorted(ts[1:])[len(ts[1:]) // 2]


for L, m in ((31, nq), (21, nq), (101, nq // 2)):
    q2d = synth.random_queries(m, L, 0, 5)
    qb, qo = synth.fixed_to_csr(q2d)
    out = np.zeros(m, dtype=np.uint64)
    dt_reuse = med(lambda: ix.parallel_count_csr(qb, qo, out))
    dt_fresh = med(lambda: ix.parallel_count_csr(qb, qo))
    print("L=%d, %d queries: %.2f ms = %.2f G queries/s into a reused result array; %.2f ms = %.2f G/s into a fresh one (%.1f GB/s of ASCII)"
          % (L, m, dt_reuse * 1e3, m / dt_reuse / 1e9, dt_fresh * 1e3, m / dt_fresh / 1e9, qb.nbytes / dt_reuse / 1e9), flush=True)
    if L == 31:
        code = np.searchsorted(synth.NT, q2d).astype(np.uint64)
        words = np.zeros(m, dtype=np.uint64)
        for j in range(31):
            words |= code[:, j] << np.uint64(2 * j)
        want = out.copy()
        dt = med(lambda: ix.parallel_count_packed(words, 31, out))
        assert np.array_equal(out, want)
        print("     the same k-mers handed over packed: %.2f ms = %.2f G queries/s" % (dt * 1e3, m / dt / 1e9), flush=True)
# reads from the text, 0.5 % of them with one N: the listed reads travel as a compact batch of their own
reads = synth.sampled_queries(text, nq // 2, 101, 9)
sel = np.random.default_rng(3).random(len(reads)) < 0.005
reads[sel, 50] = ord("N")
qb, qo = synth.fixed_to_csr(reads)
out = np.zeros(len(reads), dtype=np.uint64)
dt = med(lambda: ix.parallel_count_csr(qb, qo, out), 5)
print("101-bp reads from the text, one N in 0.5 %% of them: %.2f ms = %.2f G reads/s (%d listed)" % (dt * 1e3, len(reads) / dt / 1e9, int(sel.sum())), flush=True)
# unequal lengths 20..39
lens = np.random.default_rng(1).integers(20, 40, size=nq)
qo = np.zeros(len(lens) + 1, dtype=np.uint64); qo[1:] = np.cumsum(lens)
qb = synth.NT[np.random.default_rng(2).integers(0, 4, size=int(qo[-1]), dtype=np.uint8)]
out = np.zeros(nq, dtype=np.uint64)
dt = med(lambda: ix.parallel_count_csr(qb, qo, out), 5)
print("unequal lengths 20..39: %d queries in %.2f ms = %.2f G queries/s" % (nq, dt * 1e3, nq / dt / 1e9), flush=True)
