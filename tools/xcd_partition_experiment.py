"""Does giving every XCD its own slice of the search tree pay for table-less backward search?  The first ~10 steps of all
queries share BWT blocks (<= 2 * 4^j lines at step j); each XCD's 4 MB L2 holds them up to j ~ 7.  If the queries an XCD works
on all start with the same two letters it sees 1/8 of every level, so two more levels fit.  The kernel maps group g of four
queries to block (g / 64) % grid and blocks go round-robin over the 8 XCDs, so the experiment needs no kernel change: the same
random batch is laid out so that block b only meets queries of letter class b % 8, and timed against the unpermuted batch.
usage: xcd_partition_experiment.py [text_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AWRY_SEED_K"] = "0"
import numpy as np, torch
from tests import synth
import awry_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
L, nq = 31, 10_000_000
text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05)
ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0).set_devices([0])
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
gen = torch.Generator(device=dev); gen.manual_seed(5)
w = torch.randint(0, 1 << (2 * L), (nq,), dtype=torch.int64, device=dev, generator=gen)
counts = torch.zeros(nq, dtype=torch.int64, device=dev)

def run(words, name):
    for _ in range(3):
        ix.dev_count_nt2(words.data_ptr(), nq, L, counts.data_ptr(), False, stream, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ix.dev_count_nt2(words.data_ptr(), nq, L, counts.data_ptr(), False, stream, 0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-44s %.3f ms  %.2f G q/s" % (name, ms, nq / ms / 1e6), flush=True)
    return counts.clone()

c0 = run(w, "random order")
grid = 2048
ngroups = (nq + 3) // 4
for nletters in (2, 3):
    cls = (w >> (2 * (L - nletters))) & ((1 << (2 * nletters)) - 1)      # the first letters the search consumes
    xcd_of_query = (cls * 8) >> (2 * nletters)                             # 8 classes of equal width
    order = torch.argsort(xcd_of_query, stable=True)
    sizes = torch.bincount(xcd_of_query, minlength=8).cpu().numpy()
    slot_x = ((torch.arange(ngroups, device=dev) // 64) % grid) % 8         # XCD of the block that owns group slot g
    slot_x4 = slot_x.repeat_interleave(4)[:nq]
    # rank of every slot among the slots of its XCD, and of every query among the queries of its class
    perm = torch.empty(nq, dtype=torch.int64, device=dev)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    leftovers_q, leftovers_s = [], []
    for x in range(8):
        slots = torch.nonzero(slot_x4 == x).flatten()
        qs = order[starts[x]: starts[x + 1]]
        m = min(len(slots), len(qs))
        perm[slots[:m]] = qs[:m]
        leftovers_q.append(qs[m:]); leftovers_s.append(slots[m:])
    lq, ls = torch.cat(leftovers_q), torch.cat(leftovers_s)
    perm[ls] = lq                                                           # class sizes differ a little: the rest goes anywhere
    wp = w[perm].contiguous()
    cp = run(wp, "block b sees letter class b %% 8 (%d letters)" % nletters)
    assert torch.equal(cp, c0[perm]), "counts differ under the permutation"
    print("   misplaced queries: %d of %d" % (len(lq), nq))
# control: the same sort without the XCD layout (queries sorted by class, blocks see all classes over time)
ws = w[torch.argsort((w >> (2 * (L - 3))) & 63, stable=True)].contiguous()
run(ws, "sorted by 3 letters, no XCD layout")
