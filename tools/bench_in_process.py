"""bench.py --in-process N: ONE process, N replicas -- the shape in which a Rust caller on an 8-GPU node uses the library
(FmIndex::parallel_count is one call over all of rayon's cores, /root/reference src/fm_index.rs:455-460; here one call
over all replicas: awry_set_devices(ids, N), then awry_count_batch / awry_count_packed_kmers / awry_locate_batch shard
the batch contiguously, one host thread and one set of pinned lanes per replica, the host packer's pool shared).

The index is built once on the host (GPU 0 builds the suffix array) and replicated by awry_set_devices: each replica
uploads the 3 GB of blocks + samples and derives its seed table and accelerators on its own GPU, all replicas
concurrently -- local HBM traffic, no peer copies, nothing shared afterwards.  With fewer GPUs than replicas the replicas
are stacked round-robin on the GPUs present (rehearsal on a one-GPU box; the seed table then shrinks to what fits).

Prints ONE JSON line in bench.py's format; `value` is the whole-job rate of the host boundary (PCIe-inclusive), the
device-resident rate of the replicas driven concurrently is reported beside it."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def log(*a):
    print("[bench-in-process]", *a, file=sys.stderr, flush=True)


def main(args):
    import torch
    import awry_amd
    from bench import WORKLOADS, unpack_nt2, workload_text
    from tests import synth

    N = args.in_process
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("no GPU visible")
    ids = [i % ndev for i in range(N)]
    n_text = args.text_len or WORKLOADS[args.workload][0]
    L, nq, K, W = args.qlen, args.queries, args.steps, args.warmup
    t0 = time.time()
    text, starts, headers, _ = workload_text(args, torch, torch.device("cuda", 0), None, n_text)
    ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, starts, headers, build_device=0)
    t1 = time.time()
    ix.set_devices(ids)
    if args.seed_k >= 0:
        ix.set_seed_kmer_len(args.seed_k)
    log("text + index %.1fs, %d replicas on devices %s: %.1fs, seed k=%d" % (t1 - t0, N, ids, time.time() - t1, ix.seed_kmer_len()))

    rng = np.random.default_rng(99)
    total = nq * N
    words = rng.integers(0, 1 << (2 * L), size=total, dtype=np.uint64)
    ascii2d = unpack_nt2(words, L)
    qb, qo = synth.fixed_to_csr(ascii2d)
    out = np.zeros(total, dtype=np.uint64)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            tp = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - tp)
        return sorted(ts)[len(ts) // 2], sum(ts)

    # the ceiling the host sets: the packer alone (ASCII -> 2-bit words on the worker pool, no GPU, no result array) and the
    # widening copy of the counts, both at this process's CPU quota -- what N replicas share however many GPUs serve them
    import ctypes as C
    lib = awry_amd.load_library()
    pw = np.zeros(total, dtype=np.uint64)
    pbad, pnb = np.zeros(total, dtype=np.uint32), C.c_uint64()
    med_pack, _ = timed(lambda: lib.awry_host_pack_nt2(qb.ctypes.data, None, total, L, pw.ctypes.data_as(C.POINTER(C.c_uint64)), None,
                                                       pbad.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pnb)), 1, 5)
    med, tot = timed(lambda: ix.parallel_count_csr(qb, qo, out), W, K)
    counts_ascii = out.copy()
    med_p, _ = timed(lambda: ix.parallel_count_packed(words, L, out), 1, max(3, K // 4))
    assert np.array_equal(out, counts_ascii), "packed and ASCII entry points disagree"
    # (the check that sharding keeps input order -- a single replica must give the same counts -- runs at the very end, on
    # this same index re-pointed at one device: a second index does not fit beside a GRCh38-scale replica's 209 GB)

    # device-resident: every replica counts its own resident batch, all replicas driven concurrently by host threads
    d_words, d_counts = [], []
    for s in range(N):
        lo, hi = total * s // N, total * (s + 1) // N
        d_words.append(ix.dev_upload(words[lo:hi], s))
        d_counts.append(ix.dev_malloc(8 * (hi - lo), s))

    def resident_round():
        def run(s):
            m = total * (s + 1) // N - total * s // N
            for _ in range(K):
                ix.dev_count_nt2(d_words[s], m, L, d_counts[s], True, None, s)
            ix.dev_synchronize(s)
        th = [threading.Thread(target=run, args=(s,)) for s in range(N)]
        tp = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - tp

    resident_round()
    dt_res = resident_round()
    for s in range(N):
        m = total * (s + 1) // N - total * s // N
        got = ix.dev_download(d_counts[s], (m,), np.uint64, s)
        assert np.array_equal(got, counts_ascii[total * s // N: total * s // N + m]), "device-resident counts of replica %d differ" % s
        ix.dev_free(d_words[s], s)
        ix.dev_free(d_counts[s], s)

    # locate over the replicas: 101-bp reads from the text
    nr = min(4_000_000, max(100_000, args.locate_reads // 25)) * N
    reads = synth.sampled_queries(text, nr, 101, 4242)
    rb, ro = synth.fixed_to_csr(reads)
    tl = []
    for rep in range(3):
        tp = time.perf_counter()
        hoff, hg, hp = ix.parallel_locate_csr(rb, ro)
        tl.append(time.perf_counter() - tp)
    assert int(hoff[-1]) >= nr and np.array_equal(text[hg[:1000, None].astype(np.int64) + np.arange(101)[None, :]], reads[np.searchsorted(hoff, np.arange(1000), side="right") - 1])
    dt_loc = sorted(tl[1:])[0]

    # replica sharding keeps input order: one replica alone must give the same counts as the batch call over N of them.  Last,
    # because it drops the N replicas (awry_set_devices frees the old ones first) and builds one: no second index, no HBM
    # beside the replicas -- at GRCh38 scale a second index's construction failed with out of memory next to the first.
    if N > 1:
        ix.set_devices([ids[0]])
        sample = rng.choice(total, size=min(total, 200_000), replace=False)
        sample.sort()
        sb, so = synth.fixed_to_csr(ascii2d[sample])
        assert np.array_equal(ix.parallel_count_csr(sb, so), counts_ascii[sample]), "sharded counts differ from a single replica's"
    single_check = True if N > 1 else "not applicable (one replica)"

    result = {
        "metric": "k-mer count queries/sec (parallel_count through awry_count_batch, random %d-mers, one process, %d replicas)" % (L, N),
        "value": total / med, "unit": "queries/s", "n_gpus": N, "steps": K, "warmup": W, "ms_per_step": med * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "mode": "in-process",
        "config": {"workload": "%s-scale synthetic nucleotide text (%d bp), %d uniform-random %d-mers per replica per call, ASCII in host memory -> "
                               "counts in host memory (PCIe-inclusive), seed table k=%d" % (args.workload, n_text, nq, L, ix.seed_kmer_len()),
                   "replica_devices": ids, "gpus_present": ndev, "host_threads": awry_amd.load_library().awry_host_threads(),
                   "sharding": "one process; index replicated per GPU by awry_set_devices; contiguous query shards, one host thread + pinned lanes "
                               "per replica, shared packer pool; no collective"},
        "caller_packed_kmers_queries_per_s": total / med_p,
        "host_packer_alone_queries_per_s": total / med_pack,
        "host_bound": {"packer_alone_queries_per_s": total / med_pack, "boundary_over_packer": (total / med) / (total / med_pack),
                       "note": "the packer (read L bytes, write 8 per query) on this process's CPU quota is the ceiling of the ASCII boundary for any number "
                               "of replicas; the replicas' shards are packed concurrently by one pool (jobs from several callers share its threads)"},
        "device_resident_queries_per_s": total * K / dt_res,
        "locate": {"reads": nr, "hits": int(hoff[-1]), "reads_per_s": nr / dt_loc, "note": "awry_locate_batch, PCIe-inclusive, best of 2 after 1 warm-up"},
        "checks": {"packed_equals_ascii": True, "sharded_equals_single_replica_on_sample": single_check, "device_resident_equals_host_path": True},
    }
    print(json.dumps(result), flush=True)
