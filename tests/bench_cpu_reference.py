"""configs[0] of BASELINE.json: E. coli-scale nucleotide index, 10k random 21-mers, count_string on the CPU
restatement of the reference path (no GPU involved).  usage: bench_cpu_reference.py [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_ffi
from tests import synth

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
n, nq, L = 4_641_652, 10_000, 21
text, st, hd = synth.make_text(n, 0, 0xA5A50001, 1, 0.0)
t = time.time(); oi = oracle_ffi.OracleIndex.from_text(text, 0, 8, 0, st, hd); build = time.time() - t
q2d = synth.random_queries(nq, L, 0, 1)
t = time.perf_counter()
for q in q2d:
    oi.count_string(q)
scalar = time.perf_counter() - t
qb, qo = synth.fixed_to_csr(np.tile(q2d, (100, 1)))
t = time.perf_counter(); counts, tally = oi.parallel_count(qb, qo, threads); par = time.perf_counter() - t
print({"workload": "E. coli-scale synthetic (4,641,652 bp), 10k random 21-mers", "oracle_build_s": round(build, 1),
       "count_string_loop_queries_per_s (python call overhead included)": round(nq / scalar),
       "parallel_count_queries_per_s": round(len(qo) - 1) / par, "threads": threads,
       "steps_per_query": tally["steps"] / tally["queries"], "block_reads_per_query": tally["block_reads"] / tally["queries"]})
