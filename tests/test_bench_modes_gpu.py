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
    """the N > 1 line as the driver will launch it (no --queries): BASELINE configs[4]'s 10^9 queries in all, divided over the
    ranks and the timed steps; parity against the ORACLE on rank 0 besides the GPU-vs-GPU checks; device-resident and
    host-boundary aggregates; each rank's packer pool sized to its share of the CPU quota"""
    env = dict(os.environ, AWRY_BENCH_BACKEND="gloo", AWRY_SEED_K="11", MASTER_ADDR="127.0.0.1")
    env.pop("AWRY_HOST_THREADS", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ecoli",
                        "--steps", "5", "--warmup", "1"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = _json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["value"] > 0
    cfg = r["config"]
    assert cfg["queries_per_gpu_per_step"] == 100_000_000 and cfg["total_queries_timed"] == 1_000_000_000 and "10^9" in cfg["workload"]
    assert cfg["ranks"] == 2 and cfg["backend"] == "gloo"
    pc = r["parity_check"]
    assert pc["replicas_agree"] and pc["default_schedule_equals_plain_backward_search_on_every_rank"]
    assert pc["rank0_gpu_matches_oracle_on_the_common_batch"] is True and pc["queries"] == 1_000_000
    hb = r["host_boundary"]
    assert hb["aggregate_queries_per_s"] > 0 and hb["rank0_counts_equal_device_resident"] and r["host_boundary_count_queries_per_s"] == hb["aggregate_queries_per_s"]
    assert hb["host_threads_per_rank"] >= 1 and r["device_resident_queries_per_s"] == r["value"]
    assert "built once" in cfg["sharding"]


def test_explicit_batch_size_keeps_weak_scaling_label():
    env = dict(os.environ, AWRY_BENCH_BACKEND="gloo", AWRY_SEED_K="11", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ecoli", "--queries", "500000",
                        "--steps", "3", "--warmup", "1", "--cpu-seconds", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = _json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["queries_per_gpu_per_step"] == 500_000
    assert r["parity_check"]["rank0_gpu_matches_oracle_on_the_common_batch"] is None  # --cpu-seconds 0: no oracle in the run
