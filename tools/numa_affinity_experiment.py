"""Does it matter on which socket the host worker pool (and the caller's buffers) live?  The same host-boundary measurement in
child processes pinned to NUMA node 0, node 1, or left alone (the GPU hangs off one of them).  usage: numa_affinity_experiment.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def cpus_of(node):
    try:
        txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    except OSError:
        return None
    out = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out
if len(sys.argv) > 1 and sys.argv[1] == "child":
    which = sys.argv[2]
    if which != "all":
        os.sched_setaffinity(0, cpus_of(int(which)) & os.sched_getaffinity(0))
    sys.path.insert(0, ROOT)
    import time, numpy as np, awry_amd
    from tests import synth
    text, st, hd = synth.make_text(100_000_000, 0, 7, 1, 0.02)
    ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
    m = 5_000_000
    def med(fn, reps=9):
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return sorted(ts[2:])[len(ts[2:]) // 2]
    for L in (31, 101):
        mm = m if L == 31 else m // 2
        qb, qo = synth.fixed_to_csr(synth.random_queries(mm, L, 0, 5))
        out = np.zeros(mm, dtype=np.uint64)
        dt = med(lambda: ix.parallel_count_csr(qb, qo, out))
        print("affinity %-4s L=%3d: %.2f ms = %.2f G queries/s (%d CPUs allowed, %d pool threads)" % (which, L, dt * 1e3, mm / dt / 1e9, len(os.sched_getaffinity(0)), awry_amd.load_library().awry_host_threads()), flush=True)
    sys.exit(0)
for node in (0, 1):
    print("node %d cpus: %s" % (node, open("/sys/devices/system/node/node%d/cpulist" % node).read().strip() if os.path.exists("/sys/devices/system/node/node%d/cpulist" % node) else "?"))
os.system("for d in /sys/class/drm/card*/device; do [ -e $d/numa_node ] && echo \"$d numa_node $(cat $d/numa_node)\"; done 2>/dev/null | head -12")
for which in ("all", "0", "1", "all"):
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", which], check=False)
