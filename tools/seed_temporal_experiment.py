"""plain vs non-temporal loads for the seed probes, by table size (AWRY_SEED_TEMPORAL=1|0 in child processes): random and present
k-mers on indexes of 4.6 Mbp (table 512 MB), 50 Mbp (2 GB), 250 Mbp (8.6 GB).  usage: seed_temporal_experiment.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, awry_amd, bench
    from tests import synth
    n = int(sys.argv[2]); L = 31
    text, st, hd = synth.make_text(n, 0, 11, 1, 0.01)
    ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
    dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
    d_text = torch.from_numpy(text).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    m = 10_000_000
    def timed(fn):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 10
    ws = [torch.randint(0, 1 << (2 * L), (m,), dtype=torch.int64, device=dev, generator=gen) for _ in range(4)]
    c = torch.zeros(m, dtype=torch.int64, device=dev)
    i = [0]
    def step():
        ix.dev_count_nt2(ws[i[0] % 4].data_ptr(), m, L, c.data_ptr(), True, stream, 0); i[0] += 1
    r = m / timed(step) / 1e6
    asc = bench.device_sampled_reads(torch, d_text, 4_000_000, L, 7, ord("N"))
    w = torch.zeros(4_000_000, dtype=torch.int64, device=dev); bad = torch.zeros(1, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(asc.contiguous().data_ptr(), 4_000_000, L, w.data_ptr(), bad.data_ptr(), stream, 0)
    p = 4_000_000 / timed(lambda: ix.dev_count_nt2(w.data_ptr(), 4_000_000, L, c.data_ptr(), True, stream, 0)) / 1e6
    print("text %11d  seed k %2d (table %.1f GB)  AWRY_SEED_TEMPORAL=%s: random %.2f G/s, from the text %.2f G/s"
          % (n, ix.seed_kmer_len(), 8 * 4 ** ix.seed_kmer_len() / 1e9, os.environ.get("AWRY_SEED_TEMPORAL"), r, p), flush=True)
    sys.exit(0)
for n in (4_600_000, 50_000_000, 250_000_000):
    for t in ("1", "0"):
        env = dict(os.environ, AWRY_SEED_TEMPORAL=t)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)], env=env, check=False)
