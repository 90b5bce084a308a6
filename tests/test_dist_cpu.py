"""N > 1 path on CPU: two gloo ranks shard a query batch contiguously, run the engine on their shard and gather;
the result must equal the single-process answer (counts, CSR offsets, positions, order).  SURVEY.md 8e."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from awry_amd import dist as adist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_are_contiguous_and_cover():
    for n in (0, 1, 7, 1000, 10**9 + 7):
        for world in (1, 2, 3, 8):
            b = [adist.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_slice_csr_rebases_offsets():
    qb = np.frombuffer(b"ACGTTTGAC", dtype=np.uint8)
    qo = np.array([0, 4, 4, 7, 9], dtype=np.uint64)
    b, o = adist.slice_csr(qb, qo, 1, 4)
    assert bytes(b) == b"TTGAC" and o.tolist() == [0, 0, 3, 5]


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_sharded_count_and_locate(tmp_path, world, oracle):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = []
    for rank in range(world):
        status, lo, hi, total = open(tmp_path / ("rank%d.txt" % rank)).read().split()
        assert status == "OK"
        seen.append((int(lo), int(hi)))
    assert seen[0][0] == 0 and all(seen[i][1] == seen[i + 1][0] for i in range(world - 1))
