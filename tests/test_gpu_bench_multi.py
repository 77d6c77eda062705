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
    port = 27000 + os.getpid() % 2000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DGLA_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
             "--scale", "32"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=800) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    jl = lambda so: [l for l in so.splitlines() if l.startswith("{")]
    assert jl(outs[1][0]) == [] and len(jl(outs[0][0])) == 1  # rank 0 prints the one line
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
