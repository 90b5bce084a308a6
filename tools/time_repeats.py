"""count / locate throughput on a repeat-rich text (families of diverged copies, as in a mammalian genome): the seed
ranges of reads from repeats hold many rows, so the quad kernels (LF steps until <= 8 rows, then the text) carry them.
usage: time_repeats.py [text_len] [n_reads]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
import bench
from tests import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 600_000_000
nr = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
rng = np.random.default_rng(12)
text = synth.NT[rng.integers(0, 4, size=n, dtype=np.uint8)]
covered = 0
for unit, copies, div in ((300, n // 3000, 0.12), (6000, n // 60000, 0.05), (150, n // 6000, 0.02)):  # ~10 % of the text each
    cons = synth.NT[rng.integers(0, 4, size=unit, dtype=np.uint8)]
    starts = rng.integers(0, n - unit, size=copies)
    for s in starts:
        cp = cons.copy()
        m = rng.random(unit) < div
        cp[m] = synth.NT[rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)]
        text[s:s + unit] = cp
    covered += unit * copies
text = np.concatenate([text, np.frombuffer(b"$", np.uint8)])
print("text %d bp, ~%.0f %% in repeat copies" % (n, 100.0 * covered / n), flush=True)
t = time.time()
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, [0], ["rep"]).set_devices([0])
print("index + replica %.1f s, seed k=%d" % (time.time() - t, ix.seed_kmer_len()), flush=True)
dev = torch.device("cuda", 0)
out = bench.locate_benchmark(ix, text, torch, dev, torch.cuda.current_stream().cuda_stream, nr, 101)
print(json.dumps({k: v for k, v in out.items() if k != "cpu_baseline"}, indent=1))
# the host boundary on the same reads (offsets, positions and (record, offset) pairs back in host memory)
reads = synth.sampled_queries(text, 1_000_000, 101, 4242)
qb, qo = synth.fixed_to_csr(reads)
for rep in range(2):
    t = time.perf_counter(); off, g, p = ix.parallel_locate_csr(qb, qo); dt = time.perf_counter() - t
print("host locate: %d reads, %d hits in %.1f ms -> %.1f M reads/s, %.1f M hits/s (PCIe-inclusive)" % (len(reads), len(g), dt * 1e3, len(reads) / dt / 1e6, len(g) / dt / 1e6))
chk = np.random.default_rng(1).integers(0, len(g), size=200000)
qi = np.repeat(np.arange(len(reads)), np.diff(off).astype(np.int64))
assert np.array_equal(text[g[chk].astype(np.int64)[:, None] + np.arange(101)[None, :]], reads[qi[chk]])
print("every sampled location holds its read")
