"""RCCL itself, with the one rank a single-GPU box allows: the halo pull / push of
dgl_amd.parallel.HaloExchange run their REAL all_to_all_single calls over the "nccl" backend (= RCCL
on ROCm) with a rank that requests rows from itself (DGLA_FORCE_COLLECTIVES=1 keeps the collectives
a world of one would skip), plain and chunk-pipelined, including a kernel queued between
pull_async and the wait.  Mirrors the reference's one-rank NCCL test
(tests/python/pytorch/cuda/test_nccl.py:14-31: sparse push / pull with world_size 1)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["DGLA_ROOT"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from dgl_amd.parallel import HaloExchange
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
n_local, f = 5000, 100
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(n_local, f, device=dev, generator=gen)
rows = torch.randint(0, n_local, (3000,), device=dev, generator=gen)      # self-halo, duplicates allowed
for chunks in (1, 3):
    ex = HaloExchange(n_local, rows.numel(), f, dev, requests={0: rows}, chunks=chunks)
    halo = torch.full((rows.numel(), f), -1.0, device=dev)
    work = ex.pull_async(x, halo)
    assert work is not None                                                # the collective really ran
    y = x * 2                                                              # queued while the exchange is in flight
    for c in range(chunks):
        work.wait_chunk(c)
    work.wait()
    torch.cuda.synchronize()
    want = x[rows]
    if chunks > 1:                                                         # chunk-major halo layout
        o2n = ex.halo_old2new.to(dev)
        got = halo[o2n]
    else:
        got = halo
    assert torch.equal(got, want), ("pull", chunks)
    assert torch.equal(y, x * 2)
    # push = transpose of pull: every halo row's value is added into the row it came from
    hg = torch.rand(rows.numel(), f, device=dev, generator=gen)
    hg_layout = hg
    if chunks > 1:
        hg_layout = torch.empty_like(hg)
        hg_layout[o2n] = hg
    grad = torch.zeros(n_local, f, device=dev)
    ex.push(hg_layout, grad)
    torch.cuda.synchronize()
    ref = torch.zeros(n_local, f, device=dev, dtype=torch.float64).index_add_(0, rows, hg.double())
    assert torch.allclose(grad.double(), ref, rtol=1e-5, atol=1e-6), ("push", chunks)
# the reference's own two entry points (python/dgl/cuda/nccl.py:7-183) through RCCL
from dgl_amd.parallel import NDArrayPartition, sparse_all_to_all_pull, sparse_all_to_all_push
part = NDArrayPartition(n_local, 1, mode="remainder")
req = torch.randint(0, n_local, (4000,), device=dev, generator=gen)
assert torch.equal(sparse_all_to_all_pull(req, x, part), x[req])
ridx, rval = sparse_all_to_all_push(req, x[req].contiguous(), part)
assert torch.equal(ridx, req) and torch.equal(rval, x[req])
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
'''


def test_halo_pull_and_push_over_rccl_with_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), DGLA_FORCE_COLLECTIVES="1", DGLA_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
