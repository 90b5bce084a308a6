"""worker for tests/test_dist_cpu.py: run under torch.distributed.run with the gloo backend.  The engine is the
CPU oracle here (test infrastructure); on a GPU box the same functions take FmIndex.parallel_*_csr."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from awry_amd import dist as adist  # noqa: E402
from oracle import oracle_ffi  # noqa: E402
from tests import synth  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    text, st, hd = synth.make_text(40000, 0, 3, 3, 0.04)
    oi = oracle_ffi.OracleIndex.from_text(text, 0, 8, 0, st, hd)
    rng = np.random.default_rng(0)
    qs = [bytes(q) for L in (3, 8, 12, 20) for q in synth.sampled_queries(text, 150, L, L)] + \
         [bytes(q) for q in synth.random_queries(301, 11, 0, 9)]
    order = rng.permutation(len(qs))
    qs = [qs[i] for i in order]
    qb, qo = oracle_ffi.pack_queries(qs)
    want_c, _ = oi.parallel_count(qb, qo, 1)
    want_off, want_g, want_p, _ = oi.parallel_locate(qb, qo, 1)
    got_c = adist.sharded_count(lambda b, o: oi.parallel_count(b, o, 1)[0], qb, qo, dist)
    got_off, got_g, got_p = adist.sharded_locate(lambda b, o: oi.parallel_locate(b, o, 1)[:3], qb, qo, dist)
    ok = (np.array_equal(got_c, want_c) and np.array_equal(got_off, want_off) and np.array_equal(got_g, want_g)
          and np.array_equal(got_p, want_p))
    lo, hi = adist.shard_bounds(len(qs), world, rank)
    dist.barrier()
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%s %d %d %d\n" % ("OK" if ok else "MISMATCH", lo, hi, int(want_off[-1])))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
