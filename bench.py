#!/usr/bin/env python3
"""bench.py -- k-mer count throughput of the MI355X FM-index engine (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path (awry_dev_count_nt2: seeded backward search over packed 31-mers)
over one batch of synthetic queries per GPU, queries already resident in HBM.  The index is replicated
per GPU and the query batch is sharded by rank with no data-path collective -> weak scaling.
Rank 0 prints ONE JSON line with the roofline object (algorithmic bytes from an in-kernel work census,
HIP-event kernel time) and, at N == 1, the CPU baseline (the oracle = C restatement of the reference's
rayon/AVX2 path, timed on this host's cores on a bounded sample of the same batch).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (text length without '$', records, N fraction)   -- SURVEY.md 8(d) C1/C2/C3 shapes
    "ecoli": (4_641_652, 1, 0.0),
    "chr1": (248_956_422, 1, 0.07),
    "grch38": (3_100_000_000, 25, 0.05),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def effective_cpus():
    """CPUs this process may actually use: the cgroup CPU quota when there is one (the GPU box shows 256 logical CPUs
    but grants 16), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except (OSError, ValueError):
            pass
    return n


def unpack_nt2(words, L):
    """uint64[n] packed k-mers -> uint8[n, L] ASCII (letter j in bits 2j..2j+1)"""
    w = words.astype(np.uint64)
    out = np.empty((len(w), L), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for j in range(L):
        out[:, j] = lut[((w >> np.uint64(2 * j)) & np.uint64(3)).astype(np.int64)]
    return out


def amino_benchmark(torch, dev, stream, n_text=90_000_000, nq=10_000_000, L=12):
    """BASELINE.json configs[3]: Swiss-Prot-scale amino index (5-bit alphabet), 10 M 12-mers, count -- the two-phase
    amino k-mer schedule (per-lane probe of the 20^k seed table + byte-text verify, generic kernel on the rest), and the
    generic one-query-per-lane kernel alone beside it.  Device-resident ASCII, HIP events."""
    import awry_amd
    from tests import synth
    text, st, hd = synth.make_text(n_text, 1, 0xA5A50004, 250_000, 0.0)
    ix = awry_amd.FmIndex.from_text(text, 1, 8, 0, st, hd, build_device=dev.index).set_devices([dev.index])
    out = {"text_len": n_text, "records": len(st), "query_len": L, "seed_k": ix.seed_kmer_len()}
    for name, q2d in (("random", synth.random_queries(nq, L, 1, 3)), ("present", synth.sampled_queries(text, nq // 4, L, 4, False, 1))):
        m = len(q2d)
        d_q = torch.from_numpy(np.concatenate([q2d.reshape(-1), np.zeros(16, dtype=np.uint8)])).to(dev)
        d_off = torch.arange(m + 1, dtype=torch.int64, device=dev) * L
        d_c = torch.zeros(m, dtype=torch.int64, device=dev)
        d_g = torch.zeros(m, dtype=torch.int64, device=dev)

        def timed(fn, reps=3):
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps

        ms = timed(lambda: ix.dev_count_ascii_uniform(d_q.data_ptr(), m, L, d_c.data_ptr(), None, stream, 0))
        ms_g = timed(lambda: ix.dev_count_ascii(d_q.data_ptr(), d_off.data_ptr(), m, d_g.data_ptr(), None, None, stream, 0))
        assert torch.equal(d_c, d_g), "the amino k-mer schedule and the generic kernel disagree"
        if name == "present":
            assert bool((d_c >= 1).all()), "a 12-mer sampled from the text was not found"
        out[name] = {"queries": m, "queries_per_s": m / (ms * 1e-3), "kernel_ms": ms,
                     "generic_kernel_queries_per_s": m / (ms_g * 1e-3)}
    return out


def locate_benchmark(ix, text, torch, dev, stream, n_reads, read_len, oi=None, cores=1):
    """SA-locate hits/s (BASELINE.json's second metric; configs[2] shape: reads sampled from the text, exact match).
    Pipeline on the device: packed reads -> seeded quad count (+range starts) -> scan -> tile locate.  Timed per phase
    with HIP events, for the file's row samples (ratio 8, mean ~7 LF steps per hit) and for the dense device SA."""
    from tests import synth
    ix.set_verify(-1)  # the default policy keeps the accelerators resident; measure the plain pipelines first
    ix.set_locate_sa_ratio(0)
    reads = synth.sampled_queries(text, n_reads, read_len, 4242)
    W = (read_len + 31) // 32
    d_ascii = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_words = torch.zeros(n_reads * W, dtype=torch.int64, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    d_sp = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    d_scr = torch.zeros(ix.dev_scan_scratch_bytes(n_reads) // 8 + 1, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(d_ascii.data_ptr(), n_reads, read_len, d_words.data_ptr(), d_bad.data_ptr(), stream, 0)
    torch.cuda.synchronize()
    assert int(d_bad.item()) == 0

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    ms_count = timed(lambda: ix.dev_count_nt2_long(d_words.data_ptr(), n_reads, read_len, d_counts.data_ptr(), d_sp.data_ptr(), True, stream, 0))
    ms_scan = timed(lambda: ix.dev_scan_counts(d_counts.data_ptr(), n_reads, d_off.data_ptr(), d_scr.data_ptr(), stream, 0))
    total = int(d_off[-1].item())
    assert bool((d_counts >= 1).all()), "a read sampled from the text was not found"
    d_g = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    d_p = torch.zeros(2 * max(total, 1), dtype=torch.int64, device=dev)
    out = {"reads": n_reads, "read_len": read_len, "hits": total, "count_phase_ms": ms_count, "scan_ms": ms_scan,
           "count_phase_reads_per_s": n_reads / (ms_count * 1e-3), "seed_k": ix.seed_kmer_len()}
    ref = None
    for ratio in (0, 1):  # 0 = the file's samples (suffix_array_compression_ratio 8), 1 = dense device SA
        ix.set_locate_sa_ratio(ratio)
        ms = timed(lambda: ix.dev_locate(d_sp.data_ptr(), d_off.data_ptr(), n_reads, total, d_g.data_ptr(), d_p.data_ptr(), stream, 0, 1))
        g = d_g[:total].cpu().numpy().view(np.uint64)
        if ref is None:
            ref = g.copy()
            # size-independent property at full size: every located position holds its read
            qi = np.repeat(np.arange(n_reads), d_counts.cpu().numpy())
            chk = np.random.default_rng(1).integers(0, total, size=min(total, 200_000))
            win = text[g[chk].astype(np.int64)[:, None] + np.arange(read_len)[None, :]]
            assert np.array_equal(win, reads[qi[chk]]), "a located position does not hold its read"
        else:
            assert np.array_equal(ref, g), "dense-SA locate differs from the sampled-SA locate"
        steps = 7.0 if ratio == 0 else 0.0  # mean LF steps per hit with row sampling at ratio 8
        alg = total * (104.0 * steps + 8.0 + 16.0)
        out["sa_ratio_%d" % ix.locate_sa_ratio()] = {
            "locate_kernel_ms": ms, "hits_per_s": total / (ms * 1e-3), "algorithmic_GBs": alg / (ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "end_to_end_reads_per_s_device_resident": n_reads / ((ms_count + ms_scan + ms) * 1e-3)}
    # seed-and-verify: dense SA + 4-bit text resident; the rest of a read is compared with the text, not LF-stepped
    tv = time.time()
    ix.set_verify(2)
    build_s = time.time() - tv
    ms_count_v = timed(lambda: ix.dev_count_nt2_long(d_words.data_ptr(), n_reads, read_len, d_counts.data_ptr(), d_sp.data_ptr(), True, stream, 0))
    ms_scan_v = timed(lambda: ix.dev_scan_counts(d_counts.data_ptr(), n_reads, d_off.data_ptr(), d_scr.data_ptr(), stream, 0))
    assert int(d_off[-1].item()) == total
    ms_loc_v = timed(lambda: ix.dev_locate(d_sp.data_ptr(), d_off.data_ptr(), n_reads, total, d_g.data_ptr(), d_p.data_ptr(), stream, 0, 1))
    assert np.array_equal(ref, d_g[:total].cpu().numpy().view(np.uint64)), "seed-and-verify locate differs"
    out["seed_and_verify"] = {"count_phase_ms": ms_count_v, "count_phase_reads_per_s": n_reads / (ms_count_v * 1e-3),
                              "locate_kernel_ms": ms_loc_v, "hits_per_s": total / (ms_loc_v * 1e-3),
                              "end_to_end_reads_per_s_device_resident": n_reads / ((ms_count_v + ms_scan_v + ms_loc_v) * 1e-3),
                              "accelerator_build_s": build_s, "identical_locations": True}
    # the host boundary (parallel_locate: ASCII reads in host memory -> offsets + positions in host memory), PCIe-inclusive
    nh_reads = min(n_reads, 4_000_000)
    qb, qo = synth.fixed_to_csr(reads[:nh_reads])
    times, times_g = [], []
    for rep in range(4):
        tp = time.perf_counter()
        hoff, hg, hp = ix.parallel_locate_csr(qb, qo)
        times.append(time.perf_counter() - tp)
        del hp
        tp = time.perf_counter()
        hoff2, hg2, _ = ix.parallel_locate_csr(qb, qo, want_pos=False)
        times_g.append(time.perf_counter() - tp)
    nhh = int(hoff[-1])
    assert np.array_equal(hg, ref[:nhh]), "host-boundary locate differs from the device-resident pipeline"
    assert np.array_equal(hoff2, hoff) and np.array_equal(hg2, hg)
    dt, dtg = sorted(times[1:])[1], sorted(times_g[1:])[1]
    out["host_boundary_end_to_end"] = {"reads": nh_reads, "hits": nhh, "ms": dt * 1e3, "reads_per_s": nh_reads / dt,
                                       "positions_only_reads_per_s": nh_reads / dtg,
                                       "note": "awry_locate_batch, PCIe-inclusive, through the Python mirror, median of 3 after 1 warm-up; reads packed on "
                                               "the host; positions_only passes hits_out = NULL (8 B per hit back instead of 24)"}
    del hoff, hg, hoff2, hg2
    if oi is not None:
        ns = min(n_reads, 200_000)
        qb, qo = synth.fixed_to_csr(reads[:ns])
        tp = time.perf_counter()
        ooff, ogpos, opos, otally = oi.parallel_locate(qb, qo, cores)
        dt = time.perf_counter() - tp
        nh = int(ooff[-1])
        off_gpu = d_off[:ns + 1].cpu().numpy().view(np.uint64)
        parity = bool(np.array_equal(off_gpu, ooff) and np.array_equal(ref[:nh], ogpos))
        out["cpu_baseline"] = {"hits_per_s": nh / dt, "reads_per_s": ns / dt, "cores": cores, "kind": "port",
                               "sample": "first %d reads, %d hits, %.1f s" % (ns, nh, dt),
                               "backsteps_per_hit": otally["backsteps"] / max(nh, 1), "steps_per_read": otally["steps"] / ns,
                               "gpu_matches_oracle_on_sample": parity}
        assert parity, "GPU locate differs from the oracle on the sample"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("AWRY_BENCH_WORKLOAD", "grch38"), choices=sorted(WORKLOADS))
    ap.add_argument("--text-len", type=int, default=0, help="override the workload's text length")
    ap.add_argument("--queries", type=int, default=10_000_000, help="queries per GPU per step")
    ap.add_argument("--qlen", type=int, default=31)
    ap.add_argument("--seed-k", type=int, default=-1, help="device seed-table k (-1 = library default, 0 = off)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--locate-reads", type=int, default=20_000_000, help="101-bp reads in the locate measurement (N=1; BASELINE configs[2] has 100 M)")
    ap.add_argument("--sweep-seed-k", default="", help="comma list of seed k to time on rank 0 before the run (stderr)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: WORLD_SIZE %d != --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    # AWRY_BENCH_BACKEND=gloo + fewer GPUs than ranks is a rehearsal mode for the N > 1 code path on a 1-GPU box
    backend = os.environ.get("AWRY_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import awry_amd
    from tests import synth

    n_text, n_rec, n_frac = WORKLOADS[args.workload]
    if args.text_len:
        n_text = args.text_len
    L, nq, K, W = args.qlen, args.queries, args.steps, args.warmup

    # ---- index: same seeded text on every rank, one replica in this rank's HBM
    t0 = time.time()
    text, starts, headers = synth.make_text(n_text, 0, 0xA5A50000 + 2, n_rec, n_frac)
    t1 = time.time()
    ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, starts, headers, build_device=local_rank)  # each rank builds on its own GPU
    t2 = time.time()
    ix.set_devices([local_rank])
    if args.seed_k >= 0:
        ix.set_seed_kmer_len(args.seed_k)
    t3 = time.time()
    if rank == 0:
        log("text %.1fs, host index build %.1fs, replicate+seed(k=%d) %.1fs, bwt_len=%d" %
            (t1 - t0, t2 - t1, ix.seed_kmer_len(), t3 - t2, ix.bwt_len()))

    # ---- synthetic query batches, generated on the device: a uniform random L-mer is a uniform 2L-bit integer
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    n_batches = max(1, min(K + W, 8))
    batches = [torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen) for _ in range(n_batches)]
    counts = torch.zeros(nq, dtype=torch.int64, device=dev)
    tally = torch.zeros(5, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i, seeded=True):
        ix.dev_count_nt2(batches[i % n_batches].data_ptr(), nq, L, counts.data_ptr(), seeded, stream, 0)

    if args.sweep_seed_k and rank == 0:
        for k in [int(x) for x in args.sweep_seed_k.split(",")]:
            ix.set_seed_kmer_len(k)
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(5):
                step(i)
            b.record()
            torch.cuda.synchronize()
            log("sweep seed k=%d: %.3f ms/launch, %.2f G queries/s" % (k, a.elapsed_time(b) / 5, nq / (a.elapsed_time(b) / 5) / 1e6))
        ix.set_seed_kmer_len(args.seed_k)

    for i in range(W):
        step(i)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    for i in range(K):
        step(W + i)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = ev0.elapsed_time(ev1) / K  # HIP events on the stream the kernel runs on

    # ---- work census of the timed batches (same kernel, TALLY variant, untimed)
    for i in range(K):
        ix.dev_count_nt2_tally(batches[(W + i) % n_batches].data_ptr(), nq, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
    torch.cuda.synchronize()
    probes, steps_exec, blocks, vsa, vtxt = [int(x) / K for x in tally.cpu().tolist()[:5]]
    # probe 16 B, block 104 B, query 8 B, result 8 B; seed-and-verify (survivors only): 8 B per SA read
    # and the <= 8 B text window of the remaining letters
    alg_bytes = 16.0 * probes + 104.0 * blocks + nq * (8.0 + 8.0) + 8.0 * vsa + 8.0 * vtxt

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    total_q = float(nq) * K * world
    value = total_q / elapsed_max

    result = {
        "metric": "k-mer count queries/sec (parallel_count, random %d-mers, index resident in HBM)" % L,
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1000.0 * elapsed_max / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s-scale synthetic nucleotide text (%d bp, %d record(s), %.0f%% N), %d uniform-random %d-mers per GPU per step, "
                               "packed 2-bit queries resident in HBM, seed table k=%d, SA ratio 8"
                               % (args.workload, n_text, n_rec, 100 * n_frac, nq, L, ix.seed_kmer_len()),
                   "text_len": n_text, "queries_per_gpu_per_step": nq, "query_len": L, "seed_k": ix.seed_kmer_len(),
                   "sharding": "index replicated per GPU, queries sharded by rank, no collective"},
        "roofline": {"bound": "hbm", "kernel": ix.count_schedule(L), "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "census_per_launch": {"seed_probes": probes, "steps": steps_exec, "block_reads": blocks,
                                           "verify_sa_reads": vsa, "verify_text_windows": vtxt}},
    }

    # HBM traffic per launch from the committed rocprofv3 PMC passes of this exact configuration (null otherwise)
    try:
        tkey = "%s|nq=%d|L=%d|seed_k=%d|kernel=%s" % (args.workload if not args.text_len else "custom", nq, L, ix.seed_kmer_len(), ix.count_schedule(L))
        tentry = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(tkey)
        if tentry:
            result["roofline"]["traffic"] = tentry["traffic_bytes_per_launch"]
            result["roofline"]["traffic_source"] = tentry["source"]
            # the HBM bytes the kernel really moves per second (PMC traffic / live kernel time): every 8-B seed entry drags a
            # whole line, so this sits far above the algorithmic rate
            result["roofline"]["traffic_GBs"] = tentry["traffic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
            result["roofline"]["traffic_frac_of_peak"] = result["roofline"]["traffic_GBs"] / HBM_PEAK_GBS
    except (OSError, ValueError):
        pass
    # the ceiling that actually binds this access pattern: random 128-B line requests (profiles/r01_gather_calibration.txt)
    lines = probes + blocks + vsa + vtxt + nq * (8.0 + 8.0) / 128.0
    result["roofline"]["random_line_rate"] = {"achieved_Glines_s": lines / (kernel_ms * 1e-3) / 1e9, "measured_ceiling_Glines_s": 48.0,
                                              "note": "seed probe + ranked blocks + verify SA reads and text windows + coalesced share of query/result words; ceiling = "
                                                      "tools/calib_gather.hip: 8-B probes into a 34-137 GiB table (53 on 2 GiB; 44 when whole 128-B lines are consumed)"}

    if rank == 0:
        # nominal peak next to a measured streaming figure (device-to-device copy, read + write bytes)
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        result["roofline"]["peak_measured_stream_copy"] = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b

    if rank == 0 and world == 1:
        extra = {}
        if not args.no_variants:
            # the same batch without the seed table (the reference's step schedule minus nothing: every step executed)
            tally.zero_()
            for i in range(3):
                step(i, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5):
                step(i, False)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            for i in range(5):
                ix.dev_count_nt2_tally(batches[i % n_batches].data_ptr(), nq, L, counts.data_ptr(), tally.data_ptr(), False, stream, 0)
            torch.cuda.synchronize()
            p2, s2, b2 = [int(x) / 5 for x in tally.cpu().tolist()[:3]]
            ab = 104.0 * b2 + nq * 16.0
            extra["unseeded"] = {"queries_per_s": nq / (ms * 1e-3), "kernel_ms": ms, "achieved_GBs": ab / (ms * 1e-3) / 1e9,
                                 "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "steps_per_query": s2 / nq, "block_reads_per_query": b2 / nq,
                                 "note": "algorithmic bytes (104 B per ranked block) can exceed the HBM peak here: the blocks of the first ~10 steps "
                                         "of every query are shared by all queries and come from L2 / Infinity Cache"}
            # queries drawn from the text: present => all L - k steps execute
            ns = min(nq, 2_000_000)
            present = synth.sampled_queries(text, ns, L, 77)
            d_ascii = torch.from_numpy(present.reshape(-1)).to(dev)
            d_words = torch.zeros(ns, dtype=torch.int64, device=dev)
            d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
            ix.dev_pack_nt2(d_ascii.data_ptr(), ns, L, d_words.data_ptr(), d_bad.data_ptr(), stream, 0)
            torch.cuda.synchronize()
            assert int(d_bad.item()) == 0
            for _ in range(2):
                ix.dev_count_nt2(d_words.data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
            e0.record()
            for _ in range(5):
                ix.dev_count_nt2(d_words.data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            assert bool((counts[:ns] >= 1).all()), "a k-mer sampled from the text was not found"
            tally.zero_()
            ix.dev_count_nt2_tally(d_words.data_ptr(), ns, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
            torch.cuda.synchronize()
            p3, s3, b3, v3, t3 = [int(x) for x in tally.cpu().tolist()[:5]]
            ab = 16.0 * p3 + 104.0 * b3 + ns * 16.0 + 8.0 * v3 + 8.0 * t3
            extra["present_queries"] = {"queries": ns, "queries_per_s": ns / (ms * 1e-3), "kernel_ms": ms,
                                        "achieved_GBs": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "steps_per_query": s3 / ns, "verify_sa_reads_per_query": v3 / ns,
                                        "verify_text_windows_per_query": t3 / ns,
                                        "random_lines_per_s": (p3 + b3 + v3 + t3) / (ms * 1e-3)}
            # the same present k-mers by LF steps only (seed-and-verify accelerators dropped): the contrast to the default
            want_present = counts[:ns].clone()
            had_verify = ix.verify_enabled()
            ix.set_verify(-1)
            for _ in range(2):
                ix.dev_count_nt2(d_words.data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
            e0.record()
            for _ in range(5):
                ix.dev_count_nt2(d_words.data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
            e1.record()
            torch.cuda.synchronize()
            msv = e0.elapsed_time(e1) / 5
            assert bool(torch.equal(counts[:ns], want_present)), "seed-and-verify changed a count"
            extra["present_queries"]["seed_and_verify"] = bool(had_verify)
            extra["present_queries_lf_steps_only"] = {"queries_per_s": ns / (msv * 1e-3), "kernel_ms": msv, "identical_counts": True}
            if had_verify:
                ix.set_verify(2)
            # ASCII boundary with on-device packing in the timed region (31 B/query read instead of 8 B)
            na = min(nq, 5_000_000)
            asc = torch.from_numpy(unpack_nt2(batches[0][:na].cpu().numpy().view(np.uint64), L).reshape(-1)).to(dev)
            w2 = torch.zeros(na, dtype=torch.int64, device=dev)
            for rep in range(2):
                e0.record()
                for _ in range(5):
                    ix.dev_pack_nt2(asc.data_ptr(), na, L, w2.data_ptr(), d_bad.data_ptr(), stream, 0)
                    ix.dev_count_nt2(w2.data_ptr(), na, L, counts.data_ptr(), True, stream, 0)
                e1.record()
                torch.cuda.synchronize()
            extra["ascii_resident_pack_plus_count"] = {"queries": na, "queries_per_s": na / (e0.elapsed_time(e1) / 5 * 1e-3)}
            # the same through the one-call entry point (pack + packed kernels + the generic kernel over the list of
            # queries with other letters, scratch in the replica)
            c2 = torch.zeros(na, dtype=torch.int64, device=dev)
            for rep in range(2):
                e0.record()
                for _ in range(5):
                    ix.dev_count_ascii_uniform(asc.data_ptr(), na, L, c2.data_ptr(), None, stream, 0)
                e1.record()
                torch.cuda.synchronize()
            assert bool(torch.equal(c2, counts[:na])), "awry_dev_count_ascii_uniform disagrees with pack + count"
            extra["ascii_resident_one_call"] = {"queries": na, "queries_per_s": na / (e0.elapsed_time(e1) / 5 * 1e-3)}
            # the host boundary itself: ASCII + offsets in host memory -> awry_count_batch -> counts in host
            # memory; PCIe-inclusive, never the bench `value`
            h_q = asc.cpu().numpy()
            h_off = np.arange(na + 1, dtype=np.uint64) * np.uint64(L)

            def host_median(fn, reps=8):
                ts = []
                for _ in range(reps):
                    tp = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - tp)
                return sorted(ts[1:])[len(ts[1:]) // 2]

            h_counts = np.zeros(na, dtype=np.uint64)  # caller-owned counts_out, reused from call to call
            med = host_median(lambda: ix.parallel_count_csr(h_q, h_off, h_counts))
            assert np.array_equal(h_counts, counts[:na].cpu().numpy().view(np.uint64))
            med_fresh = host_median(lambda: ix.parallel_count_csr(h_q, h_off))
            h_words = batches[0][:na].cpu().numpy().view(np.uint64)
            med_packed = host_median(lambda: ix.parallel_count_packed(h_words, L, h_counts))
            assert np.array_equal(h_counts, counts[:na].cpu().numpy().view(np.uint64))
            extra["host_boundary_end_to_end"] = {
                "queries": na, "queries_per_s": na / med, "ms": med * 1e3, "host_in_GBs": h_q.nbytes / med / 1e9,
                "fresh_result_array_queries_per_s": na / med_fresh, "caller_packed_kmers_queries_per_s": na / med_packed,
                "host_threads": awry_amd.load_library().awry_host_threads(),
                "note": "awry_count_batch: ASCII + offsets in host memory -> counts in host memory, PCIe-inclusive, through the Python mirror, "
                        "median of 7 after 1 warm-up; the host packs 2 bits per letter on its worker pool (8 B per 31-mer over PCIe), "
                        "counts return as 32-bit words; queries_per_s reuses the caller's result array, fresh_result_array allocates "
                        "one per call (first-touch page faults + the allocator's mmap/munmap)"}
        result["variants"] = extra

        if args.cpu_seconds > 0:
            from oracle import oracle_ffi
            path = "/tmp/awry_bench_%d.awry" % os.getpid()
            ts = time.time()
            ix.save(path)  # .awry v1 in the reference's own layout: 160-B blocks, packed SA, k-mer table
            oi = oracle_ffi.OracleIndex.load(path)
            os.remove(path)
            log("oracle index via .awry round trip: %.1fs" % (time.time() - ts))
            cores = effective_cpus()
            sample = min(nq, 10_000_000)
            w0 = batches[W % n_batches][:sample].cpu().numpy().view(np.uint64)
            qb, qo = synth.fixed_to_csr(unpack_nt2(w0, L))
            probe_n = min(sample, 2_000_000)
            oi.parallel_count(qb[:probe_n * L], qo[:probe_n + 1], cores)  # warm the thread pool / page tables
            tp = time.perf_counter()
            oi.parallel_count(qb[:probe_n * L], qo[:probe_n + 1], cores)
            rate = probe_n / (time.perf_counter() - tp)
            passes = int(max(1, min(200, round(rate * args.cpu_seconds / sample))))
            tp = time.perf_counter()
            for _ in range(passes):  # the sample touches far more index bytes than the host caches hold
                ocounts, otally = oi.parallel_count(qb, qo, cores)
            dt = (time.perf_counter() - tp) / passes
            # parity of the timed GPU path against the oracle on the same sample
            ix.dev_count_nt2(batches[W % n_batches].data_ptr(), nq, L, counts.data_ptr(), True, stream, 0)
            torch.cuda.synchronize()
            gcounts = counts[:sample].cpu().numpy().view(np.uint64)
            parity = bool(np.array_equal(gcounts, ocounts))
            result["cpu_baseline"] = {"value": sample / dt, "unit": "queries/s", "cores": cores, "kind": "port",
                                      "sample": "first %d queries of timed batch 0 x %d passes (same index via .awry v1 round trip), "
                                                "reference step schedule, %d threads = this job's CPU quota (%d logical CPUs on the host), "
                                                "%.1f s total" % (sample, passes, cores, os.cpu_count() or 0, dt * passes),
                                      "steps_per_query": otally["steps"] / sample,
                                      "block_reads_per_query": otally["block_reads"] / sample,
                                      "gpu_matches_oracle_on_sample": parity}
            if not parity:
                log("PARITY FAILURE: GPU counts differ from the oracle on the sample")
                print(json.dumps(result))
                sys.exit(3)
        else:
            oi, cores = None, 1
        if not args.no_variants:
            del batches
            torch.cuda.empty_cache()
            result["locate"] = locate_benchmark(ix, text, torch, dev, stream, args.locate_reads, 101, oi, cores)
            if args.workload == "grch38":
                del ix
                torch.cuda.empty_cache()
                result["amino"] = amino_benchmark(torch, dev, stream)

    if world > 1:
        # parity re-checked at every G, outside the timed region.  All ranks count one common batch (half
        # k-mers of the text, half random; same seeds everywhere) twice: with the default schedule (seed table, context
        # and position seeds, text comparison) and by plain backward search from the last letter with no table and no
        # accelerator -- the reference's own algorithm on the GPU.  The two must agree on every rank, k-mers of the text
        # must be found, and the replicas must agree among themselves (checksums reduced with MIN / MAX).  The oracle
        # itself is compared against at N = 1 only (cpu_baseline).
        from tests import synth
        npar = 1_000_000
        common = np.concatenate([synth.sampled_queries(text, npar // 2, L, 4711), synth.random_queries(npar // 2, L, 0, 4712)])
        d_ascii = torch.from_numpy(common.reshape(-1)).to(dev)
        d_w = torch.zeros(npar, dtype=torch.int64, device=dev)
        d_b = torch.zeros(1, dtype=torch.int64, device=dev)
        d_c = torch.zeros(npar, dtype=torch.int64, device=dev)
        d_c2 = torch.zeros(npar, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(d_ascii.data_ptr(), npar, L, d_w.data_ptr(), d_b.data_ptr(), stream, 0)
        ix.dev_count_nt2(d_w.data_ptr(), npar, L, d_c.data_ptr(), True, stream, 0)
        ix.dev_count_nt2(d_w.data_ptr(), npar, L, d_c2.data_ptr(), False, stream, 0)
        torch.cuda.synchronize()
        local_ok = bool(torch.equal(d_c, d_c2)) and bool((d_c[:npar // 2] >= 1).all()) and int(d_b.item()) == 0
        weights = torch.arange(1, npar + 1, dtype=torch.int64, device=dev) % 1000003
        chk = torch.stack([d_c.sum(), (d_c * weights).sum(), torch.tensor(1 if local_ok else 0, dtype=torch.int64, device=dev)])
        if backend != "nccl":
            chk = chk.cpu()
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        agree = bool(torch.equal(lo_[:2], hi_[:2]))
        all_ok = int(lo_[2].item()) == 1
        if rank == 0:
            result["parity_check"] = {"queries": npar, "present_fraction": 0.5, "replicas_agree": agree,
                                      "default_schedule_equals_plain_backward_search_on_every_rank": all_ok}
            if not (agree and all_ok):
                log("PARITY FAILURE at %d GPUs" % world)
                print(json.dumps(result), flush=True)
                dist.destroy_process_group()
                sys.exit(3)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
