#!/usr/bin/env python3
"""bench.py -- k-mer count throughput of the MI355X FM-index engine (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)
    python bench.py --in-process N                         (one process, N replicas: the path a Rust caller on a node uses)

One "step" = one pass of the hot path (awry_dev_count_nt2: seeded backward search over packed 31-mers)
over one batch of synthetic queries per GPU, queries already resident in HBM.  The index is replicated
per GPU and the query batch is sharded by rank with no data-path collective -> weak scaling.
Rank 0 prints ONE JSON line with the roofline object (algorithmic bytes from an in-kernel work census,
HIP-event kernel time, HBM traffic from rocprofv3 counter passes of this very run) and, at N == 1, the CPU
baseline (the oracle = C restatement of the reference's rayon/AVX2 path, timed on this host's cores on a
bounded sample of the same batch).  At N == 1 the line also carries every other BASELINE config at its own size
(`variants`, `locate` = configs[2] with 100 M reads, `amino` = configs[3]), each checked against the oracle in the run.

HBM traffic: after its own measurements the run starts itself again under `rocprofv3 --pmc` (one child per counter
set: FETCH_SIZE / WRITE_SIZE / TCC hit+miss, as MI355X_MICROARCH.md prescribes) on the index it saved; the child
replays the device-resident phases, each announced by a marker kernel whose grid size is the phase id, and the
parent cuts the per-dispatch counter rows into phases.  traffic = 2 * FETCH_SIZE KB + WRITE_SIZE KB (gfx950).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (text length without '$', records, N fraction, composition)   -- SURVEY.md 8(d) C1/C2/C3 shapes
    "ecoli": (4_641_652, 1, 0.0, "iid"),
    "chr1": (248_956_422, 1, 0.07, "iid"),
    "grch38": (3_100_000_000, 25, 0.05, "iid"),
    # the default: GRCh38's size AND composition -- 43 % of the text in repeat families (300 bp .. 6 kb units, 10^3 .. 10^6
    # copies, 2 .. 15 % divergence), satellite and tandem arrays, segmental duplications, assembly gaps (tests/synth.py)
    "grch38-repeats": (3_100_000_000, 25, 0.05, "repeats"),
    "chr1-repeats": (248_956_422, 1, 0.05, "repeats"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
AMINO_TEXT, AMINO_RECORDS, AMINO_NQ, AMINO_L = 90_000_000, 250_000, 10_000_000, 12  # BASELINE.json configs[3]
PMC_SETS = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"]))


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


# Every host array this file takes from a device tensor is `.cpu().numpy().copy()`: numpy-owned memory, the torch CPU tensor
# freed at once.  While an array that a pageable HIP device-to-host copy has just written is still alive, the next
# submission of the process stalls once for 12-16 ms inside the runtime (tools/first_call_triggers.py, variants 4-6;
# profiles/r03x_first_call_triggers.txt) -- an effect of the bench's own copies that a caller reading a file never sees.
# It is NOT the whole story of `first_call_ms` below: with detached arrays the first awry_count_batch of THIS process, which
# comes after minutes of device-API work, still takes ~22 ms (5 ms in its first copies in, 16 ms waiting for the GPU;
# profiles/r03O_bench_first_call_trace.txt) where a fresh process that calls right after awry_set_devices sees 2 ms.
# `first_call_ms` is reported as measured.
HEADLINE_LOOP_MARKER = 59999  # phase id of the markers around the headline's timed loop (ids of the variants count up from 1)


class Ctx:
    """what every measurement needs: the device, the stream the kernels run on, and the phase markers"""

    def __init__(self, torch, dev, child=False):
        self.torch, self.dev, self.child = torch, dev, child
        self.stream = torch.cuda.current_stream().cuda_stream
        self.phases = {}  # name -> {"id": marker grid size, "launches": launches of the measured op inside the phase}
        self.ix = None    # the index whose replica queues the markers

    def phase(self, name, launches):
        pid = len(self.phases) + 1
        self.phases[name] = {"id": pid, "launches": launches}
        self.ix.dev_phase_marker(pid, self.stream, 0)

    def end_phase(self):
        self.ix.dev_phase_marker(60000, self.stream, 0)  # whatever follows belongs to no phase

    def timed(self, name, fn, warm=2, reps=3):
        """ms per call of fn (HIP events on the kernels' stream) over `reps` calls after `warm`; all of them inside phase `name`"""
        torch = self.torch
        self.phase(name, warm + reps)
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        self.end_phase()
        return a.elapsed_time(b) / reps


def workload_text(args, torch, dev, workload=None, n_text=None):
    """-> (text uint8 incl. '$', record starts, headers, info) of a workload: i.i.d. letters with runs of N (numpy), or the
    repeat-rich genome-shaped text (generated on the GPU)"""
    from tests import synth
    n, n_rec, n_frac, comp = WORKLOADS[workload or args.workload]
    n = n_text or n
    if comp == "repeats":
        text, starts, headers, info = synth.repeat_rich_text(n, 0xA5A50000 + 6, n_rec, device=str(dev))
        info = {k: v for k, v in info.items() if k != "families"} | {"families": {k: v["copies"] for k, v in info["families"].items()}}
        return text, starts, headers, dict(info, composition="repeat-rich (tests/synth.repeat_rich_text)")
    text, starts, headers = synth.make_text(n, 0, 0xA5A50000 + 2, n_rec, n_frac)
    return text, starts, headers, {"composition": "i.i.d. uniform ACGT", "n_fraction": n_frac, "repeat_fraction": 0.0}


def device_sampled_reads(torch, text_d, n_reads, L, seed, amb):
    """n_reads windows of L symbols drawn from the device copy of the text at uniform positions, windows holding the
    ambiguity letter (or the sentinel) skipped -- SURVEY.md 8(d) C3's reads, generated where they are used"""
    gen = torch.Generator(device=text_d.device)
    gen.manual_seed(seed)
    n = text_d.numel() - 1
    out = torch.empty((n_reads, L), dtype=torch.uint8, device=text_d.device)
    ar = torch.arange(L, device=text_d.device)
    filled, chunk = 0, 4_000_000
    while filled < n_reads:
        m = min(chunk, n_reads - filled)
        pos = torch.randint(0, n - L, (m + m // 8 + 64,), device=text_d.device, generator=gen)
        win = text_d[pos[:, None] + ar[None, :]]
        win = win[~(win == amb).any(dim=1)]
        k = min(win.shape[0], n_reads - filled)
        out[filled:filled + k] = win[:k]
        filled += k
        del pos, win
    return out


def attach_traffic(entry, ms, pmc, phase):
    """adds {traffic (HBM bytes per launch from the counter passes), traffic_GBs, traffic_frac_of_peak, l2 hit rate} to a variant"""
    t = (pmc or {}).get(phase)
    entry["traffic"] = t["traffic_bytes_per_launch"] if t else None
    if t:
        entry["traffic_GBs"] = t["traffic_bytes_per_launch"] / (ms * 1e-3) / 1e9
        entry["traffic_frac_of_peak"] = entry["traffic_GBs"] / HBM_PEAK_GBS
        entry["traffic_source"] = t["source"]
        if t.get("tcc_hit_per_launch") is not None:
            entry["l2_misses_per_launch"] = t["tcc_miss_per_launch"]
            entry["l2_hit_rate"] = t["tcc_hit_per_launch"] / max(1.0, t["tcc_hit_per_launch"] + t["tcc_miss_per_launch"])
    return entry


# ------------------------------------------------------------------------------------------------ device-resident legs
PRESENT_BATCHES = 4  # distinct batches of k-mers from the text, rotated: what one rotation touches (4 x 10 M queries x >= 1 random
                     # 128-B line each = 5 GB) is many times the 256 MiB Infinity Cache, so its hits cannot flatter rate or traffic


def lcx_price(t):
    """algorithmic bytes of the left-context index work in a census: 32 B per node consulted (the 4 keys a binary search
    inside a 16-key node compares) + 8 B per (position, row) entry read"""
    return 32.0 * t[6] + 8.0 * t[7]


def run_variants(ctx, ix, text_d, batches, nq, L, counts, tally, oi, cores, ablate_lcx=False):
    """the same index, other batches (N = 1): no seed table, k-mers drawn from the text (default and LF-only), ASCII resident"""
    torch, dev, stream = ctx.torch, ctx.dev, ctx.stream
    from tests import synth
    extra = {}
    nb = len(batches)

    def step(i, seeded):
        ix.dev_count_nt2(batches[i % nb].data_ptr(), nq, L, counts.data_ptr(), seeded, stream, 0)

    # the batch without the seed table (the reference's step schedule: every step executed)
    it = iter(range(1000))
    ms = ctx.timed("unseeded", lambda: step(next(it), False), 3, 5)
    tally.zero_()
    for i in range(5):
        ix.dev_count_nt2_tally(batches[i % nb].data_ptr(), nq, L, counts.data_ptr(), tally.data_ptr(), False, stream, 0)
    torch.cuda.synchronize()
    t = [int(x) / 5 for x in tally.cpu().tolist()]
    s2, b2, deep = t[1], t[2], t[5]
    ab_all, ab_deep = 104.0 * b2 + nq * 16.0, 104.0 * deep + nq * 16.0
    extra["unseeded"] = {"queries_per_s": nq / (ms * 1e-3), "kernel_ms": ms, "steps_per_query": s2 / nq, "block_reads_per_query": b2 / nq,
                         "block_reads_beyond_step_10_per_query": deep / nq,
                         "algorithmic_GBs_all_steps": ab_all / (ms * 1e-3) / 1e9,
                         "achieved_GBs": ab_deep / (ms * 1e-3) / 1e9, "frac": ab_deep / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "frac prices only the blocks ranked after a query's first 10 steps (+ query and result words): the <= 2 * 4^j lines "
                                 "of step j <= 10 are shared by all queries and stay in L2 / Infinity Cache; all steps priced at 104 B "
                                 "(algorithmic_GBs_all_steps) exceed what HBM can deliver"}
    # k-mers at uniform text positions (windows with an N skipped): present => every letter has to be matched.  PRESENT_BATCHES
    # distinct batches, rotated
    ns = nq
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    pres, first_ascii = [], None
    for j in range(PRESENT_BATCHES):
        a = device_sampled_reads(torch, text_d, ns, L, 7700 + j, ord("N"))
        w = torch.zeros(ns, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(a.data_ptr(), ns, L, w.data_ptr(), d_bad.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        if j == 0:
            first_ascii = a[:min(ns, 1_000_000)].cpu().numpy().copy()
        pres.append(w)
        del a
    assert int(d_bad.item()) == 0
    it = iter(range(1000))
    ms = ctx.timed("present", lambda: ix.dev_count_nt2(pres[next(it) % PRESENT_BATCHES].data_ptr(), ns, L, counts.data_ptr(), True, stream, 0), PRESENT_BATCHES, 2 * PRESENT_BATCHES)
    ix.dev_count_nt2(pres[0].data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
    assert bool((counts[:ns] >= 1).all()), "a k-mer sampled from the text was not found"
    want_present = counts[:ns].clone()
    tally.zero_()
    ix.dev_count_nt2_tally(pres[0].data_ptr(), ns, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
    torch.cuda.synchronize()
    tl = [int(x) for x in tally.cpu().tolist()]
    p3, s3, b3, v3, t3 = tl[:5]
    ab = 16.0 * p3 + 104.0 * b3 + ns * 16.0 + 8.0 * v3 + 8.0 * t3 + lcx_price(tl)
    cp = want_present.cpu().numpy().copy()
    extra["present_queries"] = {"queries": ns, "batches_rotated": PRESENT_BATCHES, "queries_per_s": ns / (ms * 1e-3), "kernel_ms": ms,
                                "achieved_GBs": ab / (ms * 1e-3) / 1e9, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "steps_per_query": s3 / ns, "verify_sa_reads_per_query": v3 / ns, "verify_text_windows_per_query": t3 / ns,
                                "lcx_nodes_per_query": tl[6] / ns, "lcx_entries_per_query": tl[7] / ns,
                                "random_lines_per_s": (p3 + b3 + v3 + t3 + tl[6] + tl[7]) / (ms * 1e-3), "seed_and_verify": bool(ix.verify_enabled()),
                                "left_context_index": bool(ix.lcx_enabled()),
                                "mean_count": float(cp.mean()), "fraction_count_gt_1": float((cp > 1).mean()), "fraction_count_gt_8": float((cp > 8).mean()),
                                "max_count": int(cp.max())}
    if oi is not None:  # the oracle on a sample of the same k-mers
        nso = len(first_ascii)
        ocounts, _ = oi.parallel_count(*synth.fixed_to_csr(first_ascii), cores)
        ok = bool(np.array_equal(ocounts, cp[:nso].view(np.uint64)))
        extra["present_queries"]["gpu_matches_oracle_on_sample"] = ok
        extra["present_queries"]["oracle_sample"] = nso
        assert ok, "GPU counts of k-mers from the text differ from the oracle"
    had_verify = ix.verify_enabled()
    if ablate_lcx and ix.lcx_enabled():  # round 2's schedule: seed-and-verify, no left-context index (LF steps until <= 8 rows)
        ix.set_lcx(False)
        it = iter(range(1000))
        msx = ctx.timed("present_no_lcx", lambda: ix.dev_count_nt2(pres[next(it) % PRESENT_BATCHES].data_ptr(), ns, L, counts.data_ptr(), True, stream, 0), PRESENT_BATCHES, PRESENT_BATCHES)
        ix.dev_count_nt2(pres[0].data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
        assert bool(torch.equal(counts[:ns], want_present)), "the left-context index changed a count"
        tally.zero_()
        ix.dev_count_nt2_tally(pres[0].data_ptr(), ns, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
        torch.cuda.synchronize()
        tx = [int(x) for x in tally.cpu().tolist()]
        extra["present_queries_without_left_context_index"] = {"queries_per_s": ns / (msx * 1e-3), "kernel_ms": msx, "steps_per_query": tx[1] / ns,
                                                               "block_reads_per_query": tx[2] / ns, "identical_counts": True}
        ix.set_lcx(True)
    # the same k-mers by LF steps only (seed-and-verify accelerators dropped): the contrast to the default
    ix.set_verify(-1)
    it = iter(range(1000))
    msv = ctx.timed("present_lf", lambda: ix.dev_count_nt2(pres[next(it) % PRESENT_BATCHES].data_ptr(), ns, L, counts.data_ptr(), True, stream, 0), 1, PRESENT_BATCHES)
    ix.dev_count_nt2(pres[0].data_ptr(), ns, L, counts.data_ptr(), True, stream, 0)
    assert bool(torch.equal(counts[:ns], want_present)), "seed-and-verify changed a count"
    tally.zero_()
    ix.dev_count_nt2_tally(pres[0].data_ptr(), ns, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
    torch.cuda.synchronize()
    p4, s4, b4 = [int(x) for x in tally.cpu().tolist()[:3]]
    ab = 16.0 * p4 + 104.0 * b4 + ns * 16.0
    extra["present_queries_lf_steps_only"] = {"queries": ns, "queries_per_s": ns / (msv * 1e-3), "kernel_ms": msv, "steps_per_query": s4 / ns,
                                              "achieved_GBs": ab / (msv * 1e-3) / 1e9, "frac": ab / (msv * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "identical_counts": True}
    if had_verify:
        ix.set_verify(2)
    del pres
    if ctx.child:
        return extra
    # ASCII boundary with on-device packing in the timed region (31 B/query read instead of 8 B)
    ix.dev_count_nt2(batches[0].data_ptr(), nq, L, counts.data_ptr(), True, stream, 0)
    na = min(nq, 5_000_000)
    asc = torch.from_numpy(unpack_nt2(batches[0][:na].cpu().numpy().copy().view(np.uint64), L).reshape(-1)).to(dev)
    w2 = torch.zeros(na, dtype=torch.int64, device=dev)

    def pack_count():
        ix.dev_pack_nt2(asc.data_ptr(), na, L, w2.data_ptr(), d_bad.data_ptr(), stream, 0)
        ix.dev_count_nt2(w2.data_ptr(), na, L, counts.data_ptr(), True, stream, 0)
    extra["ascii_resident_pack_plus_count"] = {"queries": na, "queries_per_s": na / (ctx.timed("ascii_pack_count", pack_count, 5, 5) * 1e-3)}
    c2 = torch.zeros(na, dtype=torch.int64, device=dev)
    ms1 = ctx.timed("ascii_one_call", lambda: ix.dev_count_ascii_uniform(asc.data_ptr(), na, L, c2.data_ptr(), None, stream, 0), 5, 5)
    assert bool(torch.equal(c2, counts[:na])), "awry_dev_count_ascii_uniform disagrees with pack + count"
    extra["ascii_resident_one_call"] = {"queries": na, "queries_per_s": na / (ms1 * 1e-3)}
    # the host boundary itself: ASCII + offsets in host memory -> awry_count_batch -> counts in host
    # memory; PCIe-inclusive, never the bench `value`
    import awry_amd
    h_q = asc.cpu().numpy().copy()
    h_off = np.arange(na + 1, dtype=np.uint64) * np.uint64(L)
    want_h = counts[:na].cpu().numpy().copy().view(np.uint64)

    def host_times(fn, reps=8):
        ts = []
        for _ in range(reps):
            tp = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - tp)
        return ts

    if os.environ.get("AWRY_BENCH_DIAG"):  # which threads of this process burn CPU while it is idle, right before the host leg
        def cpu_by_thread():
            r = {}
            for tid in os.listdir("/proc/self/task"):
                try:
                    f = open("/proc/self/task/%s/stat" % tid).read()
                    rest = f[f.rindex(")") + 2:].split()
                    r[tid] = (f[f.index("(") + 1:f.rindex(")")], (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
                except (OSError, ValueError):
                    pass
            return r
        c0 = cpu_by_thread(); time.sleep(1.0); c1 = cpu_by_thread()
        burn = sorted(((c1[t][1] - c0[t][1], c1[t][0]) for t in c1 if t in c0 and c1[t][1] - c0[t][1] > 0.01), reverse=True)
        log("diag: %d threads; idle for 1 s this process burnt %.2f CPU-s: %s" % (len(c1), sum(x for x, _ in burn), burn[:10]))
        try:
            log("diag: cpu.stat " + open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
        except OSError:
            pass
    h_counts = np.ones(na, dtype=np.uint64)  # caller-owned counts_out, reused from call to call (written once: its pages exist)
    # 24 calls, every one of them in the line (`call_ms`).  In this process the steady calls of a GRCh38-scale run take 2.1-2.4 ms
    # (flat over all 23: not a warm-up ramp), where `bench.py --in-process 1` makes the same calls at the same scale in 1.4 ms
    # (3.58 G/s, profiles/r03T_*); traces put the difference in the host packer (1.3-1.7 against 0.85-0.93 ms).  Why the packer
    # is slower inside this process is not known (host boundary row); the line reports what this process measures.
    ts = host_times(lambda: ix.parallel_count_csr(h_q, h_off, h_counts), 24)
    first_call, med = ts[0], sorted(ts[1:])[len(ts[1:]) // 2]
    med_2_8 = sorted(ts[1:8])[3]
    call_ms = [round(t * 1e3, 3) for t in ts]  # (ts is reused by the timing loops below)
    # the packer alone in THIS process, on the same bytes (ASCII -> 2-bit words on the worker pool; no GPU, no result array): inside
    # the pipeline it takes 1.1-1.4 ms per call here against 0.85 ms in a quiet process -- is it slow in isolation too?
    import ctypes as C
    _lib = awry_amd.load_library()
    _pw, _pbad, _pnb = np.zeros(na, dtype=np.uint64), np.zeros(na, dtype=np.uint32), C.c_uint64()
    tsp = host_times(lambda: _lib.awry_host_pack_nt2(h_q.ctypes.data, None, na, L, _pw.ctypes.data_as(C.POINTER(C.c_uint64)), None,
                                                     _pbad.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(_pnb)), 12)
    packer_alone_ms = sorted(tsp[1:])[len(tsp[1:]) // 2] * 1e3
    del _pw, _pbad
    assert np.array_equal(h_counts, want_h)
    kept = []  # results stay alive while the clock runs: releasing a 40 MB array (munmap) is the caller's cost, after the call
    ts = host_times(lambda: kept.append(ix.parallel_count_csr(h_q, h_off)))
    med_fresh = sorted(ts[1:])[len(ts[1:]) // 2]
    tp = time.perf_counter()
    n_kept = len(kept)
    del kept
    release_ms = (time.perf_counter() - tp) / n_kept * 1e3
    h_words = batches[0][:na].cpu().numpy().copy().view(np.uint64)
    ts = host_times(lambda: ix.parallel_count_packed(h_words, L, h_counts))
    med_packed = sorted(ts[1:])[len(ts[1:]) // 2]
    assert np.array_equal(h_counts, want_h)
    extra["host_boundary_end_to_end"] = {
        "queries": na, "queries_per_s": na / med, "ms": med * 1e3, "first_call_ms": first_call * 1e3, "first_call_queries_per_s": na / first_call,
        "call_ms": call_ms, "queries_per_s_median_of_calls_2_to_8": na / med_2_8, "host_packer_alone_ms": packer_alone_ms,
        "host_in_GBs": h_q.nbytes / med / 1e9,
        "fresh_result_array_queries_per_s": na / med_fresh, "fresh_result_array_release_ms": release_ms,
        "caller_packed_kmers_queries_per_s": na / med_packed,
        "host_threads": awry_amd.load_library().awry_host_threads(),
        "note": "awry_count_batch: ASCII + offsets in host memory -> counts in host memory, PCIe-inclusive, through the Python mirror; "
                "first_call = the process's very first call (pinned lane staging is set up by awry_set_devices), queries_per_s = median of the 23 "
                "calls after it; the host packs 2 bits per letter on its worker pool (8 B per 31-mer over PCIe), "
                "counts return as 32-bit words; queries_per_s reuses the caller's result array, fresh_result_array allocates "
                "one per call (mmap + first-touch page faults; the arrays are released after the clock stops: "
                "fresh_result_array_release_ms each, the allocator's munmap -- a cost the caller of any API that returns a new array pays)"}
    return extra


def locate_benchmark(ctx, ix, text_d, n_reads, read_len, oi=None, cores=1):
    """SA-locate hits/s (BASELINE.json's second metric; configs[2]: 100 M 101-bp reads sampled from the text, exact match).
    Pipeline on the device: packed reads -> count (+range starts) -> scan -> tile locate (-> walk -> localise).  Timed per
    phase with HIP events, for the file's row samples (ratio 8: LF walks, tallied by the walk kernel), for the dense device
    SA and for seed-and-verify (the default policy)."""
    torch, dev, stream = ctx.torch, ctx.dev, ctx.stream
    from tests import synth
    ix.set_verify(-1)  # the default policy keeps the accelerators resident; measure the plain pipelines first
    ix.set_locate_sa_ratio(0)
    d_reads = device_sampled_reads(torch, text_d, n_reads, read_len, 4242, ord("N"))
    W = (read_len + 31) // 32
    d_words = torch.zeros(n_reads * W, dtype=torch.int64, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    d_counts = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    d_sp = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    d_scr = torch.zeros(ix.dev_scan_scratch_bytes(n_reads) // 8 + 1, dtype=torch.int64, device=dev)
    d_tal = torch.zeros(8, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(d_reads.data_ptr(), n_reads, read_len, d_words.data_ptr(), d_bad.data_ptr(), stream, 0)
    torch.cuda.synchronize()
    assert int(d_bad.item()) == 0
    nh_reads = min(n_reads, 4_000_000)
    h_reads = d_reads[:nh_reads].cpu().numpy().copy()  # the host keeps only what the oracle and the host-boundary call need
    ms_count = ctx.timed("locate_count_lf", lambda: ix.dev_count_nt2_long(d_words.data_ptr(), n_reads, read_len, d_counts.data_ptr(), d_sp.data_ptr(), True, stream, 0), 1, 2)
    ms_scan = ctx.timed("locate_scan", lambda: ix.dev_scan_counts(d_counts.data_ptr(), n_reads, d_off.data_ptr(), d_scr.data_ptr(), stream, 0), 1, 2)
    total = int(d_off[-1].item())
    assert bool((d_counts >= 1).all()), "a read sampled from the text was not found"
    d_g = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    d_p = torch.zeros(2 * max(total, 1), dtype=torch.int64, device=dev)
    out = {"reads": n_reads, "read_len": read_len, "hits": total, "hits_per_read": total / n_reads,
           "fraction_of_reads_with_more_than_one_hit": float((d_counts > 1).float().mean().item()), "max_hits_of_a_read": int(d_counts.max().item()),
           "count_phase_ms": ms_count, "scan_ms": ms_scan,
           "count_phase_reads_per_s": n_reads / (ms_count * 1e-3), "seed_k": ix.seed_kmer_len()}
    ref = None
    for ratio in (0, 1):  # 0 = the file's samples (suffix_array_compression_ratio 8), 1 = dense device SA
        ix.set_locate_sa_ratio(ratio)
        ms = ctx.timed("locate_walk" if ratio == 0 else "locate_dense",
                       lambda: ix.dev_locate(d_sp.data_ptr(), d_off.data_ptr(), n_reads, total, d_g.data_ptr(), d_p.data_ptr(), stream, 0, 1), 1, 3)
        if ref is None:
            ref = d_g[:total].clone()
            # size-independent property at full size: every located position holds its read
            qi = torch.repeat_interleave(torch.arange(n_reads, device=dev), d_counts)
            chk = torch.randint(0, total, (min(total, 1_000_000),), device=dev)
            win = text_d[ref[chk][:, None] + torch.arange(read_len, device=dev)[None, :]]
            assert bool(torch.equal(win, d_reads[qi[chk]])), "a located position does not hold its read"
            del qi, chk, win
            # the walk kernel's own census of this launch (same kernel, TALLY instantiation, untimed)
            d_tal.zero_()
            ix.dev_locate_tally(d_sp.data_ptr(), d_off.data_ptr(), n_reads, total, d_g.data_ptr(), d_p.data_ptr(), d_tal.data_ptr(), stream, 0, 1)
            torch.cuda.synchronize()
            assert bool(torch.equal(d_g[:total], ref))
            lf_steps, walked = [int(x) for x in d_tal.cpu().tolist()[:2]]
            steps = lf_steps / max(total, 1)
        else:
            assert bool(torch.equal(ref, d_g[:total])), "dense-SA locate differs from the sampled-SA locate"
            steps, lf_steps, walked = 0.0, 0, 0
        alg = total * (8.0 + 16.0) + 104.0 * lf_steps  # 104 B per backstep + 8 B SA read + 16 B result
        out["sa_ratio_%d" % ix.locate_sa_ratio()] = {
            "locate_kernel_ms": ms, "hits_per_s": total / (ms * 1e-3), "backsteps_per_hit_tallied": steps, "hits_that_walked": walked,
            "algorithmic_GBs": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "end_to_end_reads_per_s_device_resident": n_reads / ((ms_count + ms_scan + ms) * 1e-3)}
    # seed-and-verify: dense SA + 4-bit text resident; the rest of a read is compared with the text, not LF-stepped
    torch.cuda.empty_cache()  # (the library sizes its accelerators from the HBM that is free: hand back what torch only caches)
    tv = time.time()
    ix.set_verify(2)
    build_s = time.time() - tv
    ms_count_v = ctx.timed("locate_count_sv", lambda: ix.dev_count_nt2_long(d_words.data_ptr(), n_reads, read_len, d_counts.data_ptr(), d_sp.data_ptr(), True, stream, 0), 1, 3)
    ix.dev_scan_counts(d_counts.data_ptr(), n_reads, d_off.data_ptr(), d_scr.data_ptr(), stream, 0)
    assert int(d_off[-1].item()) == total
    ms_loc_v = ctx.timed("locate_sv", lambda: ix.dev_locate(d_sp.data_ptr(), d_off.data_ptr(), n_reads, total, d_g.data_ptr(), d_p.data_ptr(), stream, 0, 1), 1, 3)
    assert bool(torch.equal(ref, d_g[:total])), "seed-and-verify locate differs"
    # per read: W query words + 16 B probe + ceil((L - k) / 2) B of 4-bit text (position seeds: no SA read) + count and range words
    alg_c = n_reads * (W * 8.0 + 16.0 + (read_len - ix.seed_kmer_len()) / 2.0 + 16.0)
    alg_l = total * (8.0 + 16.0)
    out["seed_and_verify"] = {"count_phase_ms": ms_count_v, "count_phase_reads_per_s": n_reads / (ms_count_v * 1e-3),
                              "count_phase_algorithmic_GBs": alg_c / (ms_count_v * 1e-3) / 1e9,
                              "count_phase_frac": alg_c / (ms_count_v * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "locate_kernel_ms": ms_loc_v, "hits_per_s": total / (ms_loc_v * 1e-3),
                              "frac": alg_l / (ms_loc_v * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "end_to_end_reads_per_s_device_resident": n_reads / ((ms_count_v + ms_scan + ms_loc_v) * 1e-3),
                              "accelerator_build_s": build_s, "identical_locations": True}
    off_h = d_off[:nh_reads + 1].cpu().numpy().copy().view(np.uint64)
    ref_h = ref[:int(off_h[-1])].cpu().numpy().copy().view(np.uint64)
    out["seed_and_verify"]["left_context_index"] = bool(ix.lcx_enabled())
    del d_reads, d_words, d_g, d_p, ref
    torch.cuda.empty_cache()
    if ctx.child:
        return out
    # the host boundary (parallel_locate: ASCII reads in host memory -> offsets + positions in host memory), PCIe-inclusive
    qb, qo = synth.fixed_to_csr(h_reads)
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
    assert np.array_equal(hoff, off_h) and np.array_equal(hg, ref_h[:nhh]), "host-boundary locate differs from the device-resident pipeline"
    assert np.array_equal(hoff2, hoff) and np.array_equal(hg2, hg)
    dt, dtg = sorted(times[1:])[1], sorted(times_g[1:])[1]
    out["host_boundary_end_to_end"] = {"reads": nh_reads, "hits": nhh, "ms": dt * 1e3, "reads_per_s": nh_reads / dt, "hits_per_s": nhh / dt,
                                       "first_call_ms": times[0] * 1e3, "first_call_reads_per_s": nh_reads / times[0],
                                       "positions_only_reads_per_s": nh_reads / dtg,
                                       "note": "awry_locate_batch, PCIe-inclusive, through the Python mirror: first_call = the process's first locate call "
                                               "(the pinned result pool and lane staging are set up by awry_set_devices), reads_per_s = median of the 3 calls "
                                               "after it; reads packed on the host; positions_only passes hits_out = NULL (8 B per hit back instead of 24)"}
    del hoff, hg, hoff2, hg2
    if oi is not None:
        ns = min(nh_reads, 200_000)
        qb, qo = synth.fixed_to_csr(h_reads[:ns])
        tp = time.perf_counter()
        ooff, ogpos, opos, otally = oi.parallel_locate(qb, qo, cores)
        dt = time.perf_counter() - tp
        nh = int(ooff[-1])
        parity = bool(np.array_equal(off_h[:ns + 1], ooff) and np.array_equal(ref_h[:nh], ogpos))
        out["cpu_baseline"] = {"hits_per_s": nh / dt, "reads_per_s": ns / dt, "cores": cores, "kind": "port",
                               "sample": "first %d reads, %d hits, %.1f s" % (ns, nh, dt),
                               "backsteps_per_hit": otally["backsteps"] / max(nh, 1), "steps_per_read": otally["steps"] / ns,
                               "gpu_matches_oracle_on_sample": parity}
        assert parity, "GPU locate differs from the oracle on the sample"
    return out


def amino_benchmark(ctx, ix, text, oi=None, cores=1, nq=AMINO_NQ, L=AMINO_L):
    """BASELINE.json configs[3]: Swiss-Prot-scale amino index (5-bit alphabet), 10 M 12-mers, count -- the two-phase
    amino k-mer schedule (per-lane probe of the 21^k seed table, entries that carry the residues in front of a single
    occurrence or the BWT symbols of a small range, byte-text verify, generic kernel on the listed rest), with equal
    lengths (no offsets) and with per-query offsets.  Device-resident ASCII, HIP events; a sample of each batch is counted
    by the oracle."""
    torch, dev, stream = ctx.torch, ctx.dev, ctx.stream
    from tests import synth
    text = np.asarray(text)
    out = {"text_len": len(text) - 1, "records": AMINO_RECORDS, "query_len": L, "seed_k": ix.seed_kmer_len()}
    d_tal = torch.zeros(8, dtype=torch.int64, device=dev)
    present_all = synth.sampled_queries(text, nq // 4 * PRESENT_BATCHES, L, 4, False, 1)
    for name, q2d_all in (("random", synth.random_queries(nq, L, 1, 3)), ("present", present_all)):
        # "present": PRESENT_BATCHES distinct batches rotated call by call, as in the nucleotide leg, so that no call finds
        # the lines of the one before it in L2 / Infinity Cache; checks and census run on batch 0
        parts = PRESENT_BATCHES if name == "present" else 1
        m = len(q2d_all) // parts
        q2d = q2d_all[:m]
        d_qs = [torch.from_numpy(np.concatenate([q2d_all[b * m:(b + 1) * m].reshape(-1), np.zeros(16, dtype=np.uint8)])).to(dev) for b in range(parts)]
        d_q = d_qs[0]
        d_off = torch.arange(m + 1, dtype=torch.int64, device=dev) * L
        d_c = torch.zeros(m, dtype=torch.int64, device=dev)
        d_g = torch.zeros(m, dtype=torch.int64, device=dev)
        turn = [0, 0]

        def uniform_call():
            ix.dev_count_ascii_uniform(d_qs[turn[0] % parts].data_ptr(), m, L, d_c.data_ptr(), None, stream, 0)
            turn[0] += 1

        def offsets_call():
            ix.dev_count_ascii(d_qs[turn[1] % parts].data_ptr(), d_off.data_ptr(), m, d_g.data_ptr(), None, None, stream, 0)
            turn[1] += 1

        ms = ctx.timed("amino_" + name, uniform_call, 2, max(5, 2 * parts))
        # the same batches handed over with offsets (awry_dev_count_ascii): the schedule with per-query lengths
        ms_g = ctx.timed("amino_offsets_" + name, offsets_call, 1, max(3, parts))
        ix.dev_count_ascii_uniform(d_q.data_ptr(), m, L, d_c.data_ptr(), None, stream, 0)  # batch 0 for the checks below
        ix.dev_count_ascii(d_q.data_ptr(), d_off.data_ptr(), m, d_g.data_ptr(), None, None, stream, 0)
        torch.cuda.synchronize()
        assert torch.equal(d_c, d_g), "the amino k-mer schedule with and without offsets disagree"
        if name == "present":
            assert bool((d_c >= 1).all()), "a 12-mer sampled from the text was not found"
        d_tal.zero_()
        ix.dev_count_ascii_uniform_tally(d_q.data_ptr(), m, L, d_g.data_ptr(), d_tal.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        assert torch.equal(d_c, d_g)
        probes, steps, blocks, vsa, vtxt = [int(x) for x in d_tal.cpu().tolist()[:5]]
        # 16 B per probe, 168 B per ranked amino block, L query bytes + 8 B result, 8 B per SA read, L - k text bytes
        alg = 16.0 * probes + 168.0 * blocks + m * (L + 8.0) + 8.0 * vsa + (L - ix.seed_kmer_len()) * vtxt
        out[name] = {"queries": m, "batches_rotated": parts, "queries_per_s": m / (ms * 1e-3), "kernel_ms": ms, "with_offsets_queries_per_s": m / (ms_g * 1e-3),
                     "census": {"seed_probes": probes, "steps": steps, "block_reads": blocks, "verify_sa_reads": vsa, "verify_text_windows": vtxt},
                     "achieved_GBs": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if oi is not None:
            nso = min(m, 1_000_000)
            tp = time.perf_counter()
            ocounts, _ = oi.parallel_count(*synth.fixed_to_csr(q2d[:nso]), cores)
            dt = time.perf_counter() - tp
            ok = bool(np.array_equal(ocounts, d_c[:nso].cpu().numpy().copy().view(np.uint64)))
            out[name]["gpu_matches_oracle_on_sample"] = ok
            out[name]["oracle_sample"] = nso
            out[name]["cpu_oracle_queries_per_s"] = nso / dt
            assert ok, "GPU amino counts differ from the oracle on the sample (%s)" % name
        del d_q, d_off, d_c, d_g
    return out


# ------------------------------------------------------------------------------------------------ PMC passes
def parse_pmc_dir(d, phases):
    """rocprofv3 counter_collection.csv of one pass -> {phase name: {counter: sum over the phase's dispatches, kernels: {...}}}"""
    by_id = {v["id"]: k for k, v in phases.items()}
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    cur, out = None, {}
    for r in rows:
        name = r["Kernel_Name"]
        if "phase_marker_kernel" in name:
            cur = by_id.get(int(r["Grid_Size"]) // 64)
            continue
        if cur is None:
            continue
        e = out.setdefault(cur, {"counters": {}, "kernels": {}})
        v = float(r["Counter_Value"])
        e["counters"][r["Counter_Name"]] = e["counters"].get(r["Counter_Name"], 0.0) + v
        short = name.split("(")[0].replace("awry::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        k = e["kernels"].setdefault(short, {})
        k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + v
    return out


def collect_pmc(args, state_path, tmpdir):
    """runs this script again under rocprofv3 --pmc, once per counter set; -> {phase: traffic entry} or {} when rocprofv3 is missing / fails"""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        log("rocprofv3 not found: no live counter passes")
        return {}
    phase_file = os.path.join(tmpdir, "phases.json")
    per_phase = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for tag, counters in PMC_SETS:
        d = os.path.join(tmpdir, "pmc_" + tag)
        cmd = [rocprof, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", state_path, "--phase-file", phase_file, "--steps", str(min(args.steps, 5)), "--warmup", "2",
               "--queries", str(args.queries), "--qlen", str(args.qlen), "--locate-reads", str(args.locate_reads), "--workload", args.workload]
        if args.no_variants:
            cmd.append("--no-variants")
        ts = time.time()
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            log("counter pass %s timed out after %d s: no further passes" % (tag, args.pmc_timeout))
            break
        if p.returncode != 0:
            log("counter pass %s failed (rc %d): %s" % (tag, p.returncode, p.stderr.decode(errors="replace")[-600:]))
            break
        phases = json.load(open(phase_file))
        parsed = parse_pmc_dir(d, phases)
        log("counter pass %s: %.0f s, %d phases" % (tag, time.time() - ts, len(parsed)))
        for name, e in parsed.items():
            tgt = per_phase.setdefault(name, {"launches": phases[name]["launches"], "counters": {}, "kernels": {}})
            tgt["counters"].update(e["counters"])
            for k, v in e["kernels"].items():
                tgt["kernels"].setdefault(k, {}).update(v)
        if args.keep_pmc:
            os.makedirs(args.keep_pmc, exist_ok=True)
            for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
                if "counter_collection" in f or "kernel_trace" in f:
                    shutil.copy(f, os.path.join(args.keep_pmc, "%s_%s" % (tag, os.path.basename(f).split("_", 1)[1])))
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for name, e in per_phase.items():
        c, n = e["counters"], max(1, e["launches"])
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        out[name] = {"traffic_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n,
                     "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"] / n, "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"] / n,
                     "tcc_hit_per_launch": c["TCC_HIT_sum"] / n if "TCC_HIT_sum" in c else None,
                     "tcc_miss_per_launch": c["TCC_MISS_sum"] / n if "TCC_MISS_sum" in c else None,
                     "launches": n, "kernels": {k: {cn: cv / n for cn, cv in v.items()} for k, v in e["kernels"].items()},
                     "source": "rocprofv3 --pmc child passes of this run (2*FETCH_SIZE + WRITE_SIZE KB, gfx950 correction)"}
    return out


def committed_traffic():
    """fallback when the live passes are unavailable: the committed passes of the same phases (profiles/traffic.json)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return {k: dict(v, source=v.get("source", "profiles/traffic.json")) for k, v in t.get("phases", {}).items()}
    except (OSError, ValueError):
        return {}


def pmc_child(args):
    """the run under rocprofv3 --pmc: loads what the parent saved and replays the device-resident phases"""
    import torch
    import awry_amd
    st = json.load(open(args.pmc_child))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = Ctx(torch, dev, child=True)
    L, nq, K, W = args.qlen, args.queries, args.steps, args.warmup
    ix = awry_amd.FmIndex.load(st["index"]).set_devices([0])
    if st["seed_k"] != ix.seed_kmer_len():
        ix.set_seed_kmer_len(st["seed_k"])
    ctx.ix = ix
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    n_batches = max(1, min(K + W, 8))
    batches = [torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen) for _ in range(n_batches)]
    counts = torch.zeros(nq, dtype=torch.int64, device=dev)
    tally = torch.zeros(8, dtype=torch.int64, device=dev)
    ctx.phase("headline", W + K)
    for i in range(W + K):
        ix.dev_count_nt2(batches[i % n_batches].data_ptr(), nq, L, counts.data_ptr(), True, ctx.stream, 0)
    torch.cuda.synchronize()
    ctx.end_phase()
    if not args.no_variants:
        text_d = torch.from_numpy(np.load(st["text"])).to(dev)
        run_variants(ctx, ix, text_d, batches, nq, L, counts, tally, None, 1)
        del batches, counts
        torch.cuda.empty_cache()
        locate_benchmark(ctx, ix, text_d, args.locate_reads, 101)
        del text_d
        if st.get("amino_index"):
            ix.close()
            torch.cuda.empty_cache()
            ax = awry_amd.FmIndex.load(st["amino_index"]).set_devices([0])
            ctx.ix = ax
            amino_benchmark(ctx, ax, np.load(st["amino_text"], mmap_mode="r"))
    torch.cuda.synchronize()
    json.dump(ctx.phases, open(args.phase_file, "w"))


def iid_comparison(ctx, args, torch, dev, local_rank, nq, L, n_text, n_reads):
    """the same three device-resident legs on an i.i.d. text of the same size (a second index, built and dropped here): what
    the rates of the repeat-rich default are to be read against"""
    import awry_amd
    from tests import synth
    stream = ctx.stream
    t0 = time.time()
    text, starts, headers, _ = workload_text(args, torch, dev, "grch38", n_text)
    ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, starts, headers, build_device=local_rank).set_devices([local_rank])
    text_d = torch.from_numpy(text).to(dev)
    del text
    out = {"text": "i.i.d. uniform ACGT, %d bp, 25 records, 5 %% N" % n_text, "seed_k": ix.seed_kmer_len(), "setup_s": time.time() - t0}
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)
    counts = torch.zeros(nq, dtype=torch.int64, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    rb = [torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen) for _ in range(4)]
    it = iter(range(1000))
    out["random_queries_per_s"] = nq / (timed(lambda: ix.dev_count_nt2(rb[next(it) % 4].data_ptr(), nq, L, counts.data_ptr(), True, stream, 0), 2, 8) * 1e-3)
    del rb
    pb = []
    for j in range(PRESENT_BATCHES):
        a = device_sampled_reads(torch, text_d, nq, L, 7700 + j, ord("N"))
        w = torch.zeros(nq, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(a.data_ptr(), nq, L, w.data_ptr(), d_bad.data_ptr(), stream, 0)
        torch.cuda.synchronize()
        pb.append(w)
        del a
    it = iter(range(1000))
    out["present_queries_per_s"] = nq / (timed(lambda: ix.dev_count_nt2(pb[next(it) % PRESENT_BATCHES].data_ptr(), nq, L, counts.data_ptr(), True, stream, 0), PRESENT_BATCHES, 2 * PRESENT_BATCHES) * 1e-3)
    assert bool((counts >= 1).all())
    del pb
    RL, W = 101, 4
    reads = device_sampled_reads(torch, text_d, n_reads, RL, 4242, ord("N"))
    words = torch.zeros(n_reads * W, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(reads.data_ptr(), n_reads, RL, words.data_ptr(), d_bad.data_ptr(), stream, 0)
    del reads
    rc = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    rs = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    out["reads_101_count_phase_reads_per_s"] = n_reads / (timed(lambda: ix.dev_count_nt2_long(words.data_ptr(), n_reads, RL, rc.data_ptr(), rs.data_ptr(), True, stream, 0), 1, 3) * 1e-3)
    out["reads"] = n_reads
    assert bool((rc >= 1).all())
    ix.close()
    del text_d, words, rc, rs, counts
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("AWRY_BENCH_WORKLOAD", "grch38-repeats"), choices=sorted(WORKLOADS))
    ap.add_argument("--text-len", type=int, default=0, help="override the workload's text length")
    ap.add_argument("--queries", type=int, default=0, help="queries per GPU per step (default: 10 M at N = 1; at N > 1 the 10^9 queries of "
                                                           "BASELINE configs[4] divided over the ranks and steps)")
    ap.add_argument("--qlen", type=int, default=31)
    ap.add_argument("--seed-k", type=int, default=-1, help="device seed-table k (-1 = library default, 0 = off)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--locate-reads", type=int, default=100_000_000, help="101-bp reads in the locate measurement (N=1): BASELINE configs[2] has 100 M")
    ap.add_argument("--sweep-seed-k", default="", help="comma list of seed k to time on rank 0 before the run (stderr)")
    ap.add_argument("--amino", action="store_true", help="run the amino leg (configs[3]) with any workload (default: with the grch38 workloads)")
    ap.add_argument("--no-iid", action="store_true", help="skip the i.i.d. comparison legs of the repeat-rich workload")
    ap.add_argument("--ablate-lcx", action="store_true", help="also time the k-mers from the text with the left-context index switched off")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (traffic falls back to profiles/traffic.json)")
    ap.add_argument("--pmc-timeout", type=int, default=300, help="seconds one counter pass may take")
    ap.add_argument("--keep-pmc", default="", help="directory that receives the counter CSVs of the live passes")
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--phase-file", default="", help=argparse.SUPPRESS)
    ap.add_argument("--in-process", type=int, default=0, help="one process, this many replicas (awry_set_devices): times the host batch entry points")
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.in_process:
        if not args.queries:
            args.queries = 10_000_000
        from tools import bench_in_process
        return bench_in_process.main(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # one process per GPU shares the host with the other ranks: each rank's packer pool gets its share of the CPU quota
    # (read once, when the library starts its pool)
    if world > 1 and "AWRY_HOST_THREADS" not in os.environ:
        os.environ["AWRY_HOST_THREADS"] = str(max(1, effective_cpus() // max(1, local_world)))

    import torch
    import torch.distributed as dist

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

    n_text, n_rec, n_frac, comp = WORKLOADS[args.workload]
    if args.text_len:
        n_text = args.text_len
    L, K, W = args.qlen, args.steps, args.warmup
    # queries per GPU per step: 10 M at N = 1 (configs[1] / the headline); at N > 1 BASELINE configs[4]'s 10^9 queries in
    # total, divided over the ranks and the K timed steps (N = 8, K = 20: 6.25 M per launch)
    TOTAL_N = 1_000_000_000
    nq = args.queries or (10_000_000 if world == 1 else max(1_000_000, -(-TOTAL_N // (world * K))))
    shm = "/tmp"  # scratch for the index / text handed to the other ranks and to the counter passes: RAM-backed when there is room
    try:
        st = os.statvfs("/dev/shm")
        if os.access("/dev/shm", os.W_OK) and st.f_bavail * st.f_frsize > 3 * n_text + (4 << 30):
            shm = "/dev/shm"
    except OSError:
        pass
    tmpdir = tempfile.mkdtemp(prefix="awry_bench_", dir=shm) if rank == 0 else None
    index_path = None
    want_pmc = not args.no_pmc and world == 1

    # ---- index: built ONCE (rank 0, on its GPU) and handed to the other ranks as an .awry v1 file; every rank then
    #      replicates it into its own GPU's HBM and builds its seed table / accelerators there, concurrently
    t0 = time.time()
    text, text_info = None, None
    if rank == 0:
        text, starts, headers, text_info = workload_text(args, torch, dev, None, n_text)
        torch.cuda.empty_cache()
        t1 = time.time()
        ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, starts, headers, build_device=local_rank)
        t2 = time.time()
        index_path = os.path.join(tmpdir, "index.awry")
        if world > 1 or args.cpu_seconds > 0 or want_pmc:
            ix.save(index_path)  # .awry v1 in the reference's own layout: 160-B blocks, packed SA, k-mer table
        log("text %.1fs (%s), index build %.1fs, save %.1fs" % (t1 - t0, json.dumps(text_info), t2 - t1, time.time() - t2))
    if world > 1:
        box = [index_path, text_info]
        dist.broadcast_object_list(box, src=0)
        index_path, text_info = box
        if rank != 0:
            ix = awry_amd.FmIndex.load(index_path)
    t2 = time.time()
    ix.set_devices([local_rank])
    if args.seed_k >= 0:
        ix.set_seed_kmer_len(args.seed_k)
    t3 = time.time()
    if rank == 0:
        log("replicate+seed(k=%d)+accelerators(left-context index: %s) %.1fs, bwt_len=%d" % (ix.seed_kmer_len(), bool(ix.lcx_enabled()), t3 - t2, ix.bwt_len()))
    ctx = Ctx(torch, dev)
    ctx.ix = ix
    stream = ctx.stream
    # the committed counter passes (profiles/traffic.json) describe exactly this configuration and no other
    default_config = args.workload == "grch38-repeats" and not args.text_len and nq == 10_000_000 and L == 31 and ix.seed_kmer_len() == 17

    # ---- synthetic query batches, generated on the device: a uniform random L-mer is a uniform 2L-bit integer
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    n_batches = max(1, min(K + W, 8))
    batches = [torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen) for _ in range(n_batches)]
    counts = torch.zeros(nq, dtype=torch.int64, device=dev)
    tally = torch.zeros(8, dtype=torch.int64, device=dev)

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
    # markers around the timed loop, queued outside the event pair and the wall-clock window: a kernel trace of this run
    # (tools/summarize_trace.py) can then average exactly the K timed launches and set them beside kernel_ms
    ix.dev_phase_marker(HEADLINE_LOOP_MARKER, stream, 0)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    for i in range(K):
        step(W + i)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    ix.dev_phase_marker(60000, stream, 0)
    kernel_ms = ev0.elapsed_time(ev1) / K  # HIP events on the stream the kernel runs on

    # ---- the same K steps once more, each bracketed by its own pair of events: the median beside the mean
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for i, (a, b) in enumerate(evs):
        a.record()
        step(W + i)
        b.record()
    torch.cuda.synchronize()
    per_step = sorted(a.elapsed_time(b) for a, b in evs)
    kernel_ms_median = per_step[len(per_step) // 2]

    # ---- work census of the timed batches (same kernel, TALLY variant, untimed)
    for i in range(K):
        ix.dev_count_nt2_tally(batches[(W + i) % n_batches].data_ptr(), nq, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
    torch.cuda.synchronize()
    tl = [int(x) / K for x in tally.cpu().tolist()]
    probes, steps_exec, blocks, vsa, vtxt = tl[:5]
    # probe 16 B, block 104 B, query 8 B, result 8 B; seed-and-verify (survivors only): 8 B per SA read
    # and the <= 8 B text window of the remaining letters; left-context index: 32 B per node consulted, 8 B per entry read
    alg_bytes = 16.0 * probes + 104.0 * blocks + nq * (8.0 + 8.0) + 8.0 * vsa + 8.0 * vtxt + lcx_price(tl)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    total_q = float(nq) * K * world
    value = total_q / elapsed_max

    tdesc = "%s-scale synthetic nucleotide text (%d bp, %d record(s), %.0f%% N, %s, %.0f%% of it in repeats)" % (
        args.workload.split("-")[0], n_text, n_rec, 100 * text_info.get("n_fraction", n_frac), text_info["composition"], 100 * text_info.get("repeat_fraction", 0.0))
    result = {
        "metric": "k-mer count queries/sec (parallel_count, random %d-mers, index resident in HBM)" % L,
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1000.0 * elapsed_max / K, "higher_is_better": True, "scaling": "weak" if world == 1 or args.queries else "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": tdesc + ", %d uniform-random %d-mers per GPU per step%s, packed 2-bit queries resident in HBM, seed table k=%d, SA ratio 8"
                               % (nq, L, "" if world == 1 or args.queries else " = BASELINE configs[4]'s 10^9 queries in all over %d GPUs x %d steps" % (world, K),
                                  ix.seed_kmer_len()),
                   "text_len": n_text, "text": text_info, "queries_per_gpu_per_step": nq, "total_queries_timed": int(total_q), "query_len": L,
                   "seed_k": ix.seed_kmer_len(), "left_context_index": bool(ix.lcx_enabled()),
                   "sharding": "index replicated per GPU (built once, handed over as .awry v1), queries sharded by rank, no collective",
                   "ranks": world, "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else None},
        "roofline": {"bound": "hbm", "kernel": ix.count_schedule(L), "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "kernel_ms": kernel_ms, "kernel_ms_median_of_%d" % K: kernel_ms_median,
                     "queries_per_s_at_median": nq / (kernel_ms_median * 1e-3), "algorithmic_bytes_per_launch": alg_bytes,
                     # the device's seed entry is 8 bytes, SURVEY 8(d) prices a probe at 16: the same fraction on the bytes actually needed
                     "frac_at_8_byte_probe": (alg_bytes - 8.0 * probes) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "census_per_launch": {"seed_probes": probes, "steps": steps_exec, "block_reads": blocks,
                                           "verify_sa_reads": vsa, "verify_text_windows": vtxt, "lcx_nodes": tl[6], "lcx_entries": tl[7]}},
    }
    # the ceiling that actually binds this access pattern: random 128-B line requests (profiles/r01_gather_calibration.txt)
    lines = probes + blocks + vsa + vtxt + tl[6] + tl[7] + nq * (8.0 + 8.0) / 128.0
    result["roofline"]["random_line_rate"] = {"achieved_Glines_s": lines / (kernel_ms * 1e-3) / 1e9, "measured_ceiling_Glines_s": 48.0,
                                              "note": "seed probe + ranked blocks + verify SA reads and text windows + coalesced share of query/result words; ceiling = "
                                                      "tools/calib_gather.hip: 8-B probes into a 34-137 GiB table (53 on 2 GiB; 44 when whole 128-B lines are consumed)"}

    if rank == 0:
        # nominal peak next to a measured streaming figure: a HIP copy kernel, 16 B per lane per step, read + write bytes
        a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        ix.dev_stream_copy(b.data_ptr(), a.data_ptr(), a.numel(), stream, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ix.dev_stream_copy(b.data_ptr(), a.data_ptr(), a.numel(), stream, 0)
        e1.record()
        torch.cuda.synchronize()
        result["roofline"]["peak_measured_stream_copy"] = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b

    if rank == 0 and world == 1:
        oi, cores = None, 1
        if args.cpu_seconds > 0:
            from oracle import oracle_ffi
            ts = time.time()
            oi = oracle_ffi.OracleIndex.load(index_path)
            log("oracle index via .awry round trip: %.1fs" % (time.time() - ts))
            cores = effective_cpus()
            sample = min(nq, 10_000_000)
            w0 = batches[W % n_batches][:sample].cpu().numpy().copy().view(np.uint64)
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
            gcounts = counts[:sample].cpu().numpy().copy().view(np.uint64)
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
        state = {"index": index_path, "seed_k": ix.seed_kmer_len()}
        if not args.no_variants:
            text_d = torch.from_numpy(text).to(dev)
            if want_pmc:
                state["text"] = os.path.join(tmpdir, "text.npy")
                np.save(state["text"], text)
            del text
            result["variants"] = run_variants(ctx, ix, text_d, batches, nq, L, counts, tally, oi, cores, args.ablate_lcx)
            del batches, counts
            torch.cuda.empty_cache()
            result["locate"] = locate_benchmark(ctx, ix, text_d, args.locate_reads, 101, oi, cores)
            del text_d
            if oi is not None:
                oi.close()
                oi = None
            ix.close()
            torch.cuda.empty_cache()
            if comp == "repeats" and not args.no_iid:
                result["iid_comparison"] = iid_comparison(ctx, args, torch, dev, local_rank, nq, L, n_text, min(args.locate_reads, 20_000_000))
                lo, vq, ii = result["locate"]["seed_and_verify"], result["variants"]["present_queries"], result["iid_comparison"]
                ii["ratio_iid_over_this_text"] = {"random_31mers": ii["random_queries_per_s"] / (nq / (kernel_ms * 1e-3)),
                                                  "present_31mers": ii["present_queries_per_s"] / vq["queries_per_s"],
                                                  "reads_101_count_phase": ii["reads_101_count_phase_reads_per_s"] / lo["count_phase_reads_per_s"]}
            if args.workload.startswith("grch38") or args.amino:
                atext, ast, ahd = synth.make_text(AMINO_TEXT, 1, 0xA5A50004, AMINO_RECORDS, 0.0)
                ax = awry_amd.FmIndex.from_text(atext, 1, 8, 0, ast, ahd, build_device=local_rank)
                aoi = None
                apath = os.path.join(tmpdir, "amino.awry")
                if args.cpu_seconds > 0 or want_pmc:
                    ax.save(apath)
                    state["amino_index"] = apath
                if args.cpu_seconds > 0:
                    from oracle import oracle_ffi
                    aoi = oracle_ffi.OracleIndex.load(apath)
                ax.set_devices([local_rank])
                ctx.ix = ax
                result["amino"] = amino_benchmark(ctx, ax, atext, aoi, cores)
                if aoi is not None:
                    aoi.close()
                ax.close()
                if want_pmc:
                    state["amino_text"] = os.path.join(tmpdir, "amino_text.npy")
                    np.save(state["amino_text"], atext)
                del atext
            # BASELINE's second metric and the drop-in boundary as top-level scalars of the line
            lo = result["locate"]
            result["locate_hits_per_s"] = lo["seed_and_verify"]["hits_per_s"]
            result["locate_reads_per_s_end_to_end"] = lo["seed_and_verify"]["end_to_end_reads_per_s_device_resident"]
            result["locate_hits_per_s_walks_to_file_samples"] = lo["sa_ratio_8"]["hits_per_s"]
            result["present_queries_per_s"] = result["variants"]["present_queries"]["queries_per_s"]
            result["host_boundary_count_queries_per_s"] = result["variants"]["host_boundary_end_to_end"]["queries_per_s"]
            result["host_boundary_locate_reads_per_s"] = lo["host_boundary_end_to_end"]["reads_per_s"]
            result["host_boundary_locate_hits_per_s"] = lo["host_boundary_end_to_end"]["hits_per_s"]
        else:
            del batches, counts
            ix.close()
        torch.cuda.empty_cache()
        # ---- HBM traffic of every phase from rocprofv3 counter passes of this run (children of this process, run after
        #      it has released its HBM); falls back to the committed passes of the same phases
        pmc = {}
        if want_pmc:
            if "text" not in state:
                state["text"] = os.path.join(tmpdir, "text.npy")
                np.save(state["text"], text)
            state_path = os.path.join(tmpdir, "state.json")
            json.dump(state, open(state_path, "w"))
            args.queries = nq
            tp = time.time()
            pmc = collect_pmc(args, state_path, tmpdir)
            log("counter passes: %.0f s, phases with traffic: %s" % (time.time() - tp, sorted(pmc)))
            if args.keep_pmc and pmc:
                json.dump(pmc, open(os.path.join(args.keep_pmc, "pmc_phases.json"), "w"), indent=1)
        if not pmc and default_config:
            pmc = committed_traffic()
        result["pmc_phases"] = {k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in pmc.items()}
        result["phase_ids"] = {k: v["id"] for k, v in ctx.phases.items()}  # the marker grid sizes of the device-resident legs, in run order
        attach_traffic(result["roofline"], kernel_ms, pmc, "headline")
        if "variants" in result:
            v = result["variants"]
            attach_traffic(v["unseeded"], v["unseeded"]["kernel_ms"], pmc, "unseeded")
            attach_traffic(v["present_queries"], v["present_queries"]["kernel_ms"], pmc, "present")
            attach_traffic(v["present_queries_lf_steps_only"], v["present_queries_lf_steps_only"]["kernel_ms"], pmc, "present_lf")
            lo = result["locate"]
            attach_traffic(lo["sa_ratio_8"], lo["sa_ratio_8"]["locate_kernel_ms"], pmc, "locate_walk")
            attach_traffic(lo["sa_ratio_1"], lo["sa_ratio_1"]["locate_kernel_ms"], pmc, "locate_dense")
            attach_traffic(lo["seed_and_verify"], lo["seed_and_verify"]["locate_kernel_ms"], pmc, "locate_sv")
            cnt = attach_traffic({}, lo["seed_and_verify"]["count_phase_ms"], pmc, "locate_count_sv")
            lo["seed_and_verify"].update({"count_phase_" + k: x for k, x in cnt.items()})
            cnt = attach_traffic({}, lo["count_phase_ms"], pmc, "locate_count_lf")
            lo.update({"count_phase_" + k: x for k, x in cnt.items()})
            if "amino" in result:
                for name in ("random", "present"):
                    attach_traffic(result["amino"][name], result["amino"][name]["kernel_ms"], pmc, "amino_" + name)

    if world > 1:
        # parity re-checked at every G, outside the timed region.  All ranks count one common batch (half
        # k-mers of the text, half random; made by rank 0, which holds the text, and broadcast) twice: with the default
        # schedule (seed table, context and position seeds, left-context index, text comparison) and by plain backward search
        # from the last letter with no table and no accelerator -- the reference's own algorithm on the GPU.  The two must
        # agree on every rank, k-mers of the text must be found, the replicas must agree among themselves (checksums reduced
        # with MIN / MAX), and rank 0 -- it holds the .awry file -- counts the same batch with the ORACLE.
        npar = 1_000_000
        if rank == 0:
            common = np.concatenate([synth.sampled_queries(text, npar // 2, L, 4711), synth.random_queries(npar // 2, L, 0, 4712)])
            d_ascii = torch.from_numpy(common.reshape(-1)).to(dev)
        else:
            d_ascii = torch.empty(npar * L, dtype=torch.uint8, device=dev)
        if backend == "nccl":
            dist.broadcast(d_ascii, src=0)
        else:
            h = d_ascii.cpu()
            dist.broadcast(h, src=0)
            d_ascii = h.to(dev)
        d_w = torch.zeros(npar, dtype=torch.int64, device=dev)
        d_b = torch.zeros(1, dtype=torch.int64, device=dev)
        d_c = torch.zeros(npar, dtype=torch.int64, device=dev)
        d_c2 = torch.zeros(npar, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(d_ascii.data_ptr(), npar, L, d_w.data_ptr(), d_b.data_ptr(), stream, 0)
        ix.dev_count_nt2(d_w.data_ptr(), npar, L, d_c.data_ptr(), True, stream, 0)
        ix.dev_count_nt2(d_w.data_ptr(), npar, L, d_c2.data_ptr(), False, stream, 0)
        torch.cuda.synchronize()
        local_ok = bool(torch.equal(d_c, d_c2)) and bool((d_c[:npar // 2] >= 1).all()) and int(d_b.item()) == 0
        oracle_ok, oracle_s = None, None
        if rank == 0 and args.cpu_seconds > 0:
            from oracle import oracle_ffi
            tp = time.time()
            oi = oracle_ffi.OracleIndex.load(index_path)
            ocounts, _ = oi.parallel_count(*synth.fixed_to_csr(common), max(1, effective_cpus() // max(1, local_world)))
            oi.close()
            oracle_ok = bool(np.array_equal(ocounts, d_c.cpu().numpy().copy().view(np.uint64)))
            oracle_s = time.time() - tp
            local_ok = local_ok and oracle_ok
        weights = torch.arange(1, npar + 1, dtype=torch.int64, device=dev) % 1000003
        chk = torch.stack([d_c.sum(), (d_c * weights).sum(), torch.tensor(1 if local_ok else 0, dtype=torch.int64, device=dev)])
        if backend != "nccl":
            chk = chk.cpu()
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        agree = bool(torch.equal(lo_[:2], hi_[:2]))
        all_ok = int(lo_[2].item()) == 1
        # ---- the drop-in boundary at N ranks: every rank pushes its own shard of ASCII 31-mers through awry_count_batch at
        #      the same time (barrier before, max over ranks after); each rank's packer pool has 1/N of the CPU quota
        nh = min(nq, 5_000_000)
        h_q = unpack_nt2(batches[0][:nh].cpu().numpy().copy().view(np.uint64), L).reshape(-1)
        h_off = np.arange(nh + 1, dtype=np.uint64) * np.uint64(L)
        h_counts = np.zeros(nh, dtype=np.uint64)
        ix.parallel_count_csr(h_q, h_off, h_counts)  # first call
        reps = 5
        barrier()
        tp = time.perf_counter()
        for _ in range(reps):
            ix.parallel_count_csr(h_q, h_off, h_counts)
        th = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        ix.dev_count_nt2(batches[0].data_ptr(), nq, L, counts.data_ptr(), True, stream, 0)
        torch.cuda.synchronize()
        host_ok = bool(np.array_equal(h_counts, counts[:nh].cpu().numpy().copy().view(np.uint64)))
        if rank == 0:
            result["parity_check"] = {"queries": npar, "present_fraction": 0.5, "replicas_agree": agree,
                                      "default_schedule_equals_plain_backward_search_on_every_rank": all_ok,
                                      "rank0_gpu_matches_oracle_on_the_common_batch": oracle_ok, "oracle_s": oracle_s}
            result["host_boundary_count_queries_per_s"] = float(nh) * reps * world / float(th.item())
            result["host_boundary"] = {"queries_per_rank_per_call": nh, "calls": reps, "aggregate_queries_per_s": result["host_boundary_count_queries_per_s"],
                                       "host_threads_per_rank": awry_amd.load_library().awry_host_threads(), "rank0_counts_equal_device_resident": host_ok,
                                       "note": "awry_count_batch on every rank at once (ASCII in host memory -> counts in host memory, PCIe-inclusive); "
                                               "the ranks share the host's CPU quota, AWRY_HOST_THREADS = quota / ranks each"}
            result["device_resident_queries_per_s"] = value
            if not (agree and all_ok and host_ok):
                log("PARITY FAILURE at %d GPUs" % world)
                print(json.dumps(result), flush=True)
                dist.destroy_process_group()
                sys.exit(3)

    if world > 1:
        dist.barrier()
        if rank == 0 and default_config:  # every GPU runs the single-GPU kernel on its own replica: the committed single-GPU passes
            attach_traffic(result["roofline"], kernel_ms, committed_traffic(), "headline")
    if rank == 0:
        shutil.rmtree(tmpdir, ignore_errors=True)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
