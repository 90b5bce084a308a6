"""Host packer alone (no GPU): rate in cache and from memory, to tell whether it is bound by instructions or by bandwidth.
usage: AWRY_HOST_THREADS=t time_host_pack.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from awry_amd import _lib

lib = _lib.load_library()
u64p = C.POINTER(C.c_uint64)
rng = np.random.default_rng(1)
T = lib.awry_host_threads()
for L in (31, 101):
    W = (L + 31) // 32
    for n, reps, what in ((16384 * T, 400, "in cache"), (20_000_000 * 31 // L, 6, "from memory")):
        q = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n * L)
        words = np.ones(n * W, np.uint64)
        nb = C.c_uint64()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            lib.awry_host_pack_nt2(q.ctypes.data, None, n, L, words.ctypes.data_as(u64p), None, None, C.byref(nb))
            ts.append(time.perf_counter() - t)
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        print("threads %d, L=%d, %s (%d queries): best %.2f / median %.2f G queries/s, %.1f GB/s of ASCII, %.3f G queries/s per thread"
              % (T, L, what, n, n / best / 1e9, n / med / 1e9, n * L / best / 1e9, n / best / 1e9 / T), flush=True)
# the pool's memcpy (what copies results out of pinned staging) and a plain one-thread copy for comparison
a = np.ones(1 << 28, np.uint8)
b = np.empty_like(a)
best = 1e9
for _ in range(6):
    t = time.perf_counter()
    lib.awry_host_memcpy(b.ctypes.data, a.ctypes.data, a.nbytes)
    best = min(best, time.perf_counter() - t)
print("pool memcpy of 256 MiB, %d threads: %.1f GB/s (read + write %.1f)" % (T, a.nbytes / best / 1e9, 2 * a.nbytes / best / 1e9))
del a, b
a = np.ones(1 << 27, np.uint8)
b = np.empty_like(a)
best = 1e9
for _ in range(6):
    t = time.perf_counter()
    np.copyto(b, a)
    best = min(best, time.perf_counter() - t)
print("numpy copy of 128 MiB, one thread: %.1f GB/s (read + write %.1f)" % (a.nbytes / best / 1e9, 2 * a.nbytes / best / 1e9))
