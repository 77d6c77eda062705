#!/usr/bin/env python
"""Headline benchmark: g-SpMM copy_u+sum on an ogbn-products-shaped CSR.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (dgla_spmm_csr: merge kernel + fix-up kernel) over the
whole synthetic graph: N = 2,449,029 rows, E = 61,859,140 edges, F = 100, fp32, int32 ids,
column ids uniform ("variant U") with inputs resident in HBM.  Rank 0 prints
ONE JSON line with edges/s, the HBM roofline of the dominant kernel and the CPU baseline.

With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank owns one
C2-shaped partition (weak scaling) whose last `halo` columns are feature rows owned by the
other ranks; each step first pulls them with one RCCL all_to_all_single and then runs the
local SpMM.  value = total edges of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tests.graphgen import C2_EDGES, C2_FEAT, C2_NODES, synth_csr  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak reported too


def algorithmic_bytes(n_rows, n_edges, feat, s=4, i=4):
    # per edge F*s + i, per row F*s + i (+ one extra indptr entry)
    return n_edges * (feat * s + i) + (n_rows + 1) * i + n_rows * feat * s


def pmc_traffic(kernel_substr="spmm_csr_merge_kernel"):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    of this same command (profiles/<round>/pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv; collected by
    tools_profile.sh in separate --pmc runs).  Units/corrections per MI355X_MICROARCH.md §HBM:
    both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
    reads (calibrated here on the 4 GiB stream_copy_kernel: FETCH_SIZE*1024 = 2.147e9 for
    4.295e9 bytes read; WRITE_SIZE*1024 = 4.295e9 exact), so reads are doubled."""
    import csv
    import glob

    dirs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")))
    if not dirs:
        return None, None
    vals = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(dirs[-1], "pmc_%s.csv" % name)
        if not os.path.exists(path):
            return None, None
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
             if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == name]
        if not v:
            return None, None
        vals[name] = sum(v) / len(v)
    return 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024, os.path.relpath(dirs[-1], ROOT)


def measure_copy_peak(dev, nbytes=4 << 30, iters=5):
    from dgl_amd import _capi

    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.zero_()
    for _ in range(2):
        _capi.stream_copy(dst, src)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for k in range(iters):
        _capi.stream_copy(dst, src)
        ev[k + 1].record()
    torch.cuda.synchronize()
    best = min(ev[k].elapsed_time(ev[k + 1]) for k in range(iters))
    del src, dst
    return 2 * nbytes / (best * 1e-3) / 1e9


def cpu_baseline(g, x, budget_s=20.0):
    """CPU baseline on this host's cores, on the SAME graph and features.

    kind "reference": DGL's own CPU kernel — dgl::aten::SpMMCsr<kDGLCPU,int32,float> ->
    cpu::SpMMSumCsrNaive (src/array/cpu/spmm.h:45-74,121-160), compiled from the reference
    sources by oracle/Makefile into oracle/_ref/libdglref.so (libxsmm is an empty submodule in
    the checkout, so the reference's own non-libxsmm path is what runs).  Falls back to kind
    "port" (oracle.copy_u_sum_csr, our restatement of the same loop) when that library was
    not built.  Either way this is the CHECKER being timed, never the product path."""
    import oracle
    from oracle import ref

    indptr = g["indptr"].cpu().numpy()
    indices = g["indices"].cpu().numpy()
    xh = x.cpu().numpy()
    ncpu = os.cpu_count() or 1
    out = np.zeros((indptr.shape[0] - 1, xh.shape[1]), dtype=xh.dtype)
    use_ref = ref.available()

    def one_pass(threads):
        out[...] = 0  # both kernels accumulate into a pre-zeroed output (not timed)
        t0 = time.perf_counter()
        if use_ref:
            ref.set_num_threads(threads)
            ref.spmm_csr("copy_lhs", "sum", indptr, indices, None, xh, None,
                         num_cols=xh.shape[0], out=out)
        else:
            oracle.copy_u_sum_csr(indptr, indices, xh, threads, out)
        return time.perf_counter() - t0

    one_pass(ncpu)  # warm-up (page faults, OMP pool)
    # the gather is memory-bound: on a many-core host fewer threads than hardware threads can
    # win, so sweep a few counts and keep the fastest (its thread count is reported as `cores`)
    best, cores, reps = None, ncpu, 0
    t_start = time.perf_counter()
    for th in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4)}, reverse=True):
        for _ in range(2):
            dt = one_pass(th)
            reps += 1
            if best is None or dt < best:
                best, cores = dt, th
        if time.perf_counter() - t_start > budget_s:
            break
    what = ("oracle/_ref: DGL's own SpMMCsr<kDGLCPU,int32,float> (SpMMSumCsrNaive, "
            "src/array/cpu/spmm.h:45-74) built from the reference sources; libxsmm JIT absent "
            "from the checkout" if use_ref else
            "oracle.copy_u_sum_csr = C/OpenMP restatement of DGL SpMMSumCsrNaive "
            "(oracle/_ref not built)")
    return {
        "value": g["nnz"] / best, "unit": "edges/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": "full workload (%d edges, F=%d), best of %d passes (thread-count sweep over "
                  "%d hardware threads) after 1 warm-up; %s" % (g["nnz"], x.shape[1], reps, ncpu, what),
    }, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variant", default="U", choices=["U", "L"])
    ap.add_argument("--scale", type=int, default=1, help="divide N and E (debug only)")
    ap.add_argument("--halo-frac", type=float, default=0.1,
                    help="N>1: fraction of a partition's columns that are remote feature rows")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-peak", action="store_true")
    ap.add_argument("--extra", action="store_true", help="also time variant L / int64 / API path")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from dgl_amd import _capi

    n = C2_NODES // args.scale
    e = C2_EDGES // args.scale
    f = C2_FEAT
    n_halo = 0
    if world > 1:
        from dgl_amd.parallel import HaloExchange

        n_halo = int(n * args.halo_frac) // (world - 1) * (world - 1)
    g = synth_csr(n, n + n_halo, e, args.variant, seed=20250824 + rank, device=dev)
    torch.manual_seed(12345 + rank)
    x = torch.empty(n + n_halo, f, device=dev)
    x[:n] = torch.rand(n, f, device=dev) + 1
    out = torch.empty(n, f, device=dev)
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n + n_halo)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                     dtype=torch.uint8, device=dev)
    halo = HaloExchange(n, n_halo, f, dev, seed=7 + rank) if world > 1 else None

    plan_valid = [False]

    def step():
        if halo is not None:
            halo.pull(x)  # fills x[n:] with rows owned by the peers (RCCL all-to-all)
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws,
                       plan_valid=plan_valid[0])
        plan_valid[0] = True

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()

    # ---- timed region: EXACTLY K steps between barrier + synchronize -----------------
    k_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            for _ in range(args.steps)]
    for a, b in k_ev:  # create the HIP handles
        a.record()
        b.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in k_ev:
        _capi.set_profile_events(a, b)
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _capi.set_profile_events(None, None)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    kern_ms = [a.elapsed_time(b) for a, b in k_ev]
    kern_avg = float(np.mean(kern_ms))

    result = None
    if rank == 0:
        b_alg = algorithmic_bytes(n, e, f)
        achieved = b_alg / (kern_avg * 1e-3) / 1e9
        result = {
            "metric": "edges/sec for g-SpMM copy_u+sum (feat=100); % HBM roofline",
            "value": e * world / (ms_per_step * 1e-3),
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: g-SpMM copy_u+sum on ogbn-products-shaped CSR "
                            "(N=%d rows, E=%d edges, feat=%d) fp32, int32 ids, variant %s"
                            % (n, e, f, args.variant),
                "per_gpu_edges": e, "halo_rows_per_gpu": n_halo,
                "parallelism": "1 GPU" if world == 1 else
                               "row partition per GPU + halo pull (all_to_all_single over RCCL)",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                "kernel": "spmm_csr_merge_kernel<int,float,VEC=4,copy_lhs,sum>",
                "kernel_avg_ms": kern_avg, "kernel_min_ms": float(np.min(kern_ms)),
                "algorithmic_bytes_per_launch": b_alg,
            },
        }

    if rank == 0 and args.scale == 1 and args.variant == "U":
        t, src = pmc_traffic()
        result["roofline"]["traffic"] = t
        result["roofline"]["traffic_source"] = src

    # ---- extras on rank 0, outside the timed region ----------------------------------
    if rank == 0 and world == 1:
        if not args.no_peak:
            try:
                peak = measure_copy_peak(dev, nbytes=(4 << 30) // max(1, args.scale))
                result["roofline"]["measured_copy_peak"] = peak
                result["roofline"]["frac_of_measured_peak"] = result["roofline"]["achieved"] / peak
            except Exception as ex:  # pragma: no cover
                result["roofline"]["measured_copy_peak_error"] = repr(ex)
        if not args.no_cpu:
            cb, ref = cpu_baseline(g, x)
            result["cpu_baseline"] = cb
            # the full-size run doubles as a parity check against the oracle
            err = (out.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1e-30)
            result["parity_max_rel_err_vs_oracle"] = float(np.abs(err).max())
            del ref
        if args.extra:
            result["extra"] = extras(dev, n, e, f, args)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def time_spmm(dev, g, x, reps=10):
    from dgl_amd import _capi

    n = g["num_rows"]
    out = torch.empty(n, x.shape[1], device=dev, dtype=x.dtype)
    csr = _capi.make_csr(g["indptr"], g["indices"], g["eids"], g["num_cols"])
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                     dtype=torch.uint8, device=dev)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws)
    for _ in range(2):
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, plan_valid=True)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ts = [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]
    return float(np.median(ts)), float(np.min(ts))


def extras(dev, n, e, f, args):
    res = {}
    torch.manual_seed(12345)
    x = torch.rand(n, f, device=dev) + 1
    for variant in ("U", "L"):
        for idt in (torch.int32, torch.int64):
            g = synth_csr(n, n, e, variant, device=dev, idtype=idt)
            med, mn = time_spmm(dev, g, x)
            b = algorithmic_bytes(n, e, f, i=4 if idt == torch.int32 else 8)
            res["%s_%s" % (variant, "i32" if idt == torch.int32 else "i64")] = {
                "ms_median": med, "ms_min": mn, "edges_per_s": e / (med * 1e-3),
                "achieved_GBps": b / (med * 1e-3) / 1e9}
            del g
    return res


if __name__ == "__main__":
    main()
