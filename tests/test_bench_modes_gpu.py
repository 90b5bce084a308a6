"""bench.py's multi-replica modes on the one GPU of the test box (small workload): the single-process path a Rust caller on
a node uses (--in-process N: awry_set_devices with N replicas, here stacked on GPU 0) and the one-process-per-GPU path of
the scaling runs, rehearsed with two gloo ranks that share the GPU (rank 0 builds the index once, rank 1 loads the .awry)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_in_process_replicas_mode():
    env = dict(os.environ, AWRY_SEED_K="11")  # three replicas share one GPU's HBM
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--in-process", "3", "--workload", "ecoli", "--queries", "300000",
                        "--steps", "3", "--warmup", "1", "--locate-reads", "2500000"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = _json_line(p.stdout)
    assert r["n_gpus"] == 3 and r["mode"] == "in-process" and r["config"]["replica_devices"] == [0, 0, 0]
    assert r["value"] > 0 and r["device_resident_queries_per_s"] > 0 and r["locate"]["hits"] >= r["locate"]["reads"]
    assert all(r["checks"].values())


def test_one_process_per_gpu_mode_rehearsed_with_gloo():
    env = dict(os.environ, AWRY_BENCH_BACKEND="gloo", AWRY_SEED_K="11", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ecoli", "--queries", "500000",
                        "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = _json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["parity_check"]["replicas_agree"] and r["parity_check"]["default_schedule_equals_plain_backward_search_on_every_rank"]
    assert "built once" in r["config"]["sharding"]
