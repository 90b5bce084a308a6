"""locate pipeline timing only (bench.py's locate_benchmark on a GRCh38-scale synthetic text, no CPU baseline).
usage: time_locate.py [text_len] [n_reads] [n_fraction]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import awry_amd
import bench
from tests import synth
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
nr = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
nfrac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, nfrac)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd).set_devices([0])
dev = torch.device("cuda", 0)
out = bench.locate_benchmark(ix, text, torch, dev, torch.cuda.current_stream().cuda_stream, nr, 101)
print(json.dumps({k: v for k, v in out.items() if k != "cpu_baseline"}, indent=1))
