"""count / locate rates of the default schedules on the repeat-rich GRCh38-shaped text (tests/synth.repeat_rich_text)
next to an i.i.d. text of the same size: random 31-mers, 31-mers at uniform text positions (rotating batches), 101-bp
reads (count phase and locate), each with the kernels' own work census.
usage: time_repeat_genome.py [text_len] [n_kmers] [n_reads] [iid|repeats|both]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import awry_amd
import bench
from tests import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
nk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
nr = int(float(sys.argv[3])) if len(sys.argv) > 3 else 20_000_000
which = sys.argv[4] if len(sys.argv) > 4 else "both"
if len(sys.argv) > 5:  # repeat families to leave out (an experiment: what do the young, high-copy families cost?)
    drop = set(sys.argv[5].split(","))
    synth.REPEAT_FAMILIES = tuple(f for f in synth.REPEAT_FAMILIES if f[0] not in drop)
    if "satellite" in drop:
        synth.SATELLITE_FRACTION = 0.0
    if "segdup" in drop:
        synth.SEGDUP_FRACTION = 0.0
    print("left out:", sorted(drop), flush=True)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, warm=1, reps=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(kind):
    t = time.time()
    if kind == "repeats":
        text, starts, headers, info = synth.repeat_rich_text(n, 11, 25, device="cuda")
        print(kind, "text %.1f s" % (time.time() - t), json.dumps({k: v for k, v in info.items() if k != "families"}), flush=True)
    else:
        text, starts, headers = synth.make_text(n, 0, 0xA5A50002, 25, 0.05)
        print(kind, "text %.1f s" % (time.time() - t), flush=True)
    t = time.time()
    ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, starts, headers, build_device=0)
    t1 = time.time()
    ix.set_devices([0])
    print("build %.1f s, replica %.1f s, seed k=%d, verify=%s" % (t1 - t, time.time() - t1, ix.seed_kmer_len(), ix.verify_enabled()), flush=True)
    out = {"kind": kind, "text_len": n, "seed_k": ix.seed_kmer_len()}
    L = 31
    text_d = torch.from_numpy(text).to(dev)
    counts = torch.zeros(nk, dtype=torch.int64, device=dev)
    tally = torch.zeros(8, dtype=torch.int64, device=dev)
    d_bad = torch.zeros(1, dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(99)
    rb = [torch.randint(0, 1 << (2 * L), (nk,), dtype=torch.int64, device=dev, generator=gen) for _ in range(4)]
    pb = []
    for j in range(4):
        a = bench.device_sampled_reads(torch, text_d, nk, L, 500 + j, ord("N"))
        w = torch.zeros(nk, dtype=torch.int64, device=dev)
        ix.dev_pack_nt2(a.data_ptr(), nk, L, w.data_ptr(), d_bad.data_ptr(), stream, 0)
        pb.append(w)
        del a
    torch.cuda.synchronize()
    for name, bs in (("random_31mers", rb), ("present_31mers", pb)):
        it = iter(range(10000))
        ms = timed(lambda: ix.dev_count_nt2(bs[next(it) % 4].data_ptr(), nk, L, counts.data_ptr(), True, stream, 0), 2, 8)
        tally.zero_()
        ix.dev_count_nt2_tally(bs[0].data_ptr(), nk, L, counts.data_ptr(), tally.data_ptr(), True, stream, 0)
        torch.cuda.synchronize()
        p, s, b, v, tx = [int(x) for x in tally.cpu().tolist()[:5]]
        c = counts.cpu().numpy()
        tl = tally.cpu().tolist()
        out[name] = {"G_per_s": nk / ms / 1e6, "ms": ms, "steps_per_query": s / nk, "blocks_per_query": b / nk, "sa_reads_per_query": v / nk,
                     "lcx_nodes_per_query": int(tl[6]) / nk, "lcx_entries_per_query": int(tl[7]) / nk,
                     "text_windows_per_query": tx / nk, "mean_count": float(c.mean()), "frac_count_gt1": float((c > 1).mean()),
                     "frac_count_gt8": float((c > 8).mean()), "max_count": int(c.max())}
        print(name, json.dumps(out[name]), flush=True)
    del rb, pb, counts
    # 101-bp reads at uniform text positions: count phase (default schedule) + scan + locate
    RL = 101
    W = (RL + 31) // 32
    reads = bench.device_sampled_reads(torch, text_d, nr, RL, 4242, ord("N"))
    words = torch.zeros(nr * W, dtype=torch.int64, device=dev)
    ix.dev_pack_nt2(reads.data_ptr(), nr, RL, words.data_ptr(), d_bad.data_ptr(), stream, 0)
    del reads
    d_counts = torch.zeros(nr, dtype=torch.int64, device=dev)
    d_sp = torch.zeros(nr, dtype=torch.int64, device=dev)
    d_off = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
    d_scr = torch.zeros(ix.dev_scan_scratch_bytes(nr) // 8 + 1, dtype=torch.int64, device=dev)
    ms_c = timed(lambda: ix.dev_count_nt2_long(words.data_ptr(), nr, RL, d_counts.data_ptr(), d_sp.data_ptr(), True, stream, 0), 1, 3)
    ms_c0 = timed(lambda: ix.dev_count_nt2_long(words.data_ptr(), nr, RL, d_counts.data_ptr(), None, True, stream, 0), 1, 3)
    ix.dev_scan_counts(d_counts.data_ptr(), nr, d_off.data_ptr(), d_scr.data_ptr(), stream, 0)
    total = int(d_off[-1].item())
    d_g = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
    ms_l = timed(lambda: ix.dev_locate(d_sp.data_ptr(), d_off.data_ptr(), nr, total, d_g.data_ptr(), None, stream, 0, 1), 1, 3)
    c = d_counts.cpu().numpy()
    out["reads_101"] = {"reads": nr, "hits": total, "count_G_reads_per_s": nr / ms_c / 1e6, "count_ms": ms_c, "count_only_G_reads_per_s": nr / ms_c0 / 1e6,
                        "locate_G_hits_per_s": total / ms_l / 1e6, "locate_ms": ms_l, "end_to_end_G_reads_per_s": nr / (ms_c + ms_l) / 1e6,
                        "frac_count_gt1": float((c > 1).mean()), "frac_count_gt8": float((c > 8).mean()), "max_count": int(c.max())}
    print("reads_101", json.dumps(out["reads_101"]), flush=True)
    # the same count phase by LF steps only
    ix.set_verify(-1)
    ms_lf = timed(lambda: ix.dev_count_nt2_long(words.data_ptr(), min(nr, 5_000_000), RL, d_counts.data_ptr(), d_sp.data_ptr(), True, stream, 0), 1, 2)
    out["reads_101"]["lf_only_G_reads_per_s"] = min(nr, 5_000_000) / ms_lf / 1e6
    print("reads lf-only %.3f G/s" % out["reads_101"]["lf_only_G_reads_per_s"], flush=True)
    ix.close()
    del text_d, words, d_g
    torch.cuda.empty_cache()
    return out


res = [run(k) for k in (("repeats", "iid") if which == "both" else (which,))]
print(json.dumps(res))
