"""Why does awry_count_batch take 2.2-2.4 ms per 5 M 31-mers inside the full bench.py process and 1.4 ms in a process that
does nothing else (same GRCh38-scale index, same box)?  One process; the host calls are timed after each ingredient of the
bench process is added, in bench.py's order.  Not a test: the oracle is used here only to put the same load into the process
that bench.py's cpu_baseline leg does.
usage: python tests/host_gap_bisect.py [workload]        (default grch38-repeats)"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import awry_amd
from bench import WORKLOADS, workload_text, unpack_nt2
from tests import synth


class A:  # the few arguments workload_text reads
    workload = sys.argv[1] if len(sys.argv) > 1 else "grch38-repeats"
    text_len = 0


_marker = [0]


def measure(label, ix, qb, qo, out, calls=12):
    _marker[0] += 1  # a phase marker kernel in front of every block of calls: a kernel / copy trace can be cut by state
    ix.dev_phase_marker(_marker[0], torch.cuda.current_stream().cuda_stream, 0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(calls):
        t = time.perf_counter(); ix.parallel_count_csr(qb, qo, out); ts.append((time.perf_counter() - t) * 1e3)
    print("[marker %d] %-70s first %.2f ms, median of the rest %.2f ms   %s" % (_marker[0], label, ts[0], sorted(ts[1:])[len(ts[1:]) // 2], " ".join("%.2f" % t for t in ts)), flush=True)


def measure_locate(label, ix, rb, ro, calls=5):
    ts = []
    for _ in range(calls):
        t = time.perf_counter(); r = ix.parallel_locate_csr(rb, ro); ts.append((time.perf_counter() - t) * 1e3); hits = len(r[1]); del r
    print("            locate, %-52s %d reads, %d hits: %s ms" % (label, len(ro) - 1, hits, " ".join("%.1f" % t for t in ts)), flush=True)


def state(label, ix):
    """replica configuration, CPU the idle process burns in one second (per thread), the packer alone"""
    import ctypes as C
    def cpu_by_thread():
        r = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % tid).read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                r[tid] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
            except (OSError, ValueError):
                pass
        return r
    a = cpu_by_thread(); time.sleep(1.0); b = cpu_by_thread()
    burn = sorted(((b[t][1] - a[t][1], b[t][0]) for t in b if t in a and b[t][1] - a[t][1] > 0.02), reverse=True)
    lib = awry_amd.load_library()
    pw = np.zeros(na, dtype=np.uint64); pbad = np.zeros(na, dtype=np.uint32); pnb = C.c_uint64()
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        lib.awry_host_pack_nt2(qb.ctypes.data, None, na, L, pw.ctypes.data_as(C.POINTER(C.c_uint64)), None, pbad.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pnb))
        ts.append((time.perf_counter() - t) * 1e3)
    # the headline kernel itself, device-resident: 10 M random 31-mers per launch, 8 batches rotated
    global _dw, _dc
    if _dw is None:
        _dw = [torch.randint(0, 1 << (2 * L), (10_000_000,), dtype=torch.int64, device=dev) for _ in range(8)]
        _dc = torch.zeros(10_000_000, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(5):
        ix.dev_count_nt2(_dw[i % 8].data_ptr(), 10_000_000, L, _dc.data_ptr(), True, st, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        ix.dev_count_nt2(_dw[i % 8].data_ptr(), 10_000_000, L, _dc.data_ptr(), True, st, 0)
    e1.record(); torch.cuda.synchronize()
    # pinned copies, as the lanes make them: one chunk's words (4.6 MB in, 2.3 MB out) and a whole call's (40 / 20 MB)
    global _pin, _dbuf
    if _pin is None:
        _pin = torch.empty(40_000_000, dtype=torch.uint8).pin_memory()
        _dbuf = torch.empty(40_000_000, dtype=torch.uint8, device=dev)
    bw = []
    for nbytes, d2h in ((4_600_000, False), (40_000_000, False), (2_300_000, True), (20_000_000, True)):
        for rep in range(3):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(8):
                if d2h: _pin[:nbytes].copy_(_dbuf[:nbytes], non_blocking=True)
                else: _dbuf[:nbytes].copy_(_pin[:nbytes], non_blocking=True)
            c1.record(); torch.cuda.synchronize()
        bw.append("%s %.1f MB: %.1f GB/s" % ("out" if d2h else "in", nbytes / 1e6, 8 * nbytes / (c0.elapsed_time(c1) * 1e-3) / 1e9))
    print("   state %s pinned copies: %s" % (label, "; ".join(bw)), flush=True)
    free_b, total_b = torch.cuda.mem_get_info()
    print("   state %s: seed k %d, verify %s, left-context index %s; idle for 1 s the process burnt %.2f CPU-s; packer alone %.2f ms; "
          "device-resident 10 M 31-mers %.1f us per launch; HBM free %.1f GB" % (
        label, ix.seed_kmer_len(), ix.verify_enabled(), ix.lcx_enabled(), sum(x for x, _ in burn), sorted(ts[1:])[3],
        e0.elapsed_time(e1) / 20 * 1e3, free_b / 1e9), flush=True)


_dw = _dc = _pin = _dbuf = None


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
text, starts, headers, _ = workload_text(A, torch, dev)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, starts, headers, build_device=0)
ix.set_devices([0])
na, L = 5_000_000, 31
rng = np.random.default_rng(7)
words = rng.integers(0, 1 << (2 * L), size=na, dtype=np.uint64)
qb, qo = synth.fixed_to_csr(unpack_nt2(words, L))
out = np.ones(na, dtype=np.uint64)
measure("0 nothing else in the process", ix, qb, qo, out)
reads = synth.sampled_queries(np.asarray(text), 4_000_000, 101, 9)
rb, ro = synth.fixed_to_csr(reads)
measure_locate("0 nothing else in the process", ix, rb, ro)
state("0", ix)

stream = torch.cuda.current_stream().cuda_stream
d_words = [torch.randint(0, 1 << (2 * L), (10_000_000,), dtype=torch.int64, device=dev) for _ in range(8)]
d_counts = torch.zeros(10_000_000, dtype=torch.int64, device=dev)
for i in range(60):
    ix.dev_count_nt2(d_words[i % 8].data_ptr(), 10_000_000, L, d_counts.data_ptr(), True, stream, 0)
torch.cuda.synchronize()
measure("1 + device tensors, 60 device-resident counts on torch's stream", ix, qb, qo, out)

from oracle import oracle_ffi
oracle_ffi.build()
tmp = tempfile.mkdtemp(prefix="awry_gap_")
path = os.path.join(tmp, "ix.awry")
ix.save(path)
oi = oracle_ffi.OracleIndex.load(path)
measure("2 + index saved, the CPU oracle's copy of it loaded", ix, qb, qo, out)
t = time.perf_counter()
oi.parallel_count(qb[:1_000_000 * L], qo[:1_000_001], 16)
print("   (oracle: 1 M queries on 16 threads, %.1f s)" % (time.perf_counter() - t), flush=True)
measure("3 + the oracle's 16-thread count of 1 M queries has run", ix, qb, qo, out)
state("3", ix)

ix.set_verify(-1)
measure("4a accelerators dropped (seed table rebuilt with rows)", ix, qb, qo, out)
state("4a", ix)
for i in range(10):
    ix.dev_count_nt2(d_words[i % 8].data_ptr(), 10_000_000, L, d_counts.data_ptr(), True, stream, 0)
torch.cuda.synchronize()
ix.set_verify(2)
for i in range(10):
    ix.dev_count_nt2(d_words[i % 8].data_ptr(), 10_000_000, L, d_counts.data_ptr(), True, stream, 0)
torch.cuda.synchronize()
measure("4 + accelerators dropped and rebuilt", ix, qb, qo, out)
measure_locate("4 + accelerators dropped and rebuilt", ix, rb, ro)
state("4", ix)

ix.set_devices([0])
measure("8 awry_set_devices again (a fresh replica)", ix, qb, qo, out)
measure_locate("8 awry_set_devices again (a fresh replica)", ix, rb, ro)
state("8", ix)
