#!/usr/bin/env python
"""Headline benchmark: g-SpMM copy_u+sum on an ogbn-products-shaped CSR.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (dgla_spmm_csr: split-row copy of X when the graph's
locality probe wants it, merge kernel, fix-up kernel) over the whole synthetic graph:
N = 2,449,029 rows, E = 61,859,140 edges, F = 100, fp32, int32 ids, column ids uniform
("variant U") with inputs resident in HBM; X is treated as a NEW tensor on
every step (nothing about it is cached between steps).  Rank 0 prints ONE JSON line with
edges/s, the HBM roofline of the dominant kernel and the CPU baseline; `variants` adds the
locality variant L and the static-feature mode, timed outside the K steps.

With N > 1 (launched by torch.distributed.run, one rank per GPU) the SAME graph is partitioned
over the ranks (strong scaling): node partition -> per-rank shard (own-column block +
halo-column block) -> each step = halo all-to-all over RCCL overlapped with the own-column
launch, then the halo-column launch (dgl_amd.parallel.ShardedSpMM).
value = E / max-over-ranks time; cut fraction and halo rows per rank are in `config`.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402

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

    # the most recent round whose PMC passes are committed
    dirs = [d for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")))
            if all(os.path.exists(os.path.join(d, "pmc_%s.csv" % c)) for c in ("FETCH_SIZE", "WRITE_SIZE"))]
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
    """Measured streaming peaks of THIS box with the library's tuned stream kernels
    (dgla_stream_copy_variant; benchmarks/bench_peak.py sweeps all of them): the best float4 COPY
    rate (read + write bytes) and the best READ-ONLY rate — a gather-dominated kernel (97 % reads)
    is up against the latter.  Returns (copy_GBps, read_GBps, variant names)."""
    from dgl_amd import _capi

    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.zero_()

    def best_of(variant):
        for _ in range(2):
            _capi.stream_copy_variant(dst, src, variant)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for k in range(iters):
            _capi.stream_copy_variant(dst, src, variant)
            ev[k + 1].record()
        torch.cuda.synchronize()
        return min(ev[k].elapsed_time(ev[k + 1]) for k in range(iters))

    # mode (0 nt copy, 1 plain copy, 2 read only) | 4 * blocks-per-CU selector | 16 * (U = 8) | 32 * one tile per workgroup
    copies = {"nt_persistent": 0, "nt_onetile_u4": 32, "nt_onetile_u8": 48, "plain_onetile_u4": 33}
    reads = {"read_persistent": 2, "read_onetile_u4": 34, "read_onetile_u8": 50, "read_bpc32_u8": 2 | (3 << 2) | 16}
    c = {k: 2 * nbytes / (best_of(v) * 1e-3) / 1e9 for k, v in copies.items()}
    r = {k: nbytes / (best_of(v) * 1e-3) / 1e9 for k, v in reads.items()}
    del src, dst
    kc, kr = max(c, key=c.get), max(r, key=r.get)
    return c[kc], r[kr], {"copy": kc, "read": kr}


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
    # (ii) the libxsmm-style organisation of the same sum (K-blocked, dynamic M blocks, re-tiled
    # on every call: oracle.copy_u_sum_csr_blocked ≙ spmm_blocking_libxsmm.h:432-557 with a plain
    # vectorised row add where the JIT kernel would be), at the thread count that won above
    blocked = None
    try:
        tb, info = None, None
        for _ in range(2):
            t0 = time.perf_counter()
            _, info = oracle.copy_u_sum_csr_blocked(indptr, indices, xh, cores, out=out)
            dt = time.perf_counter() - t0
            tb = dt if tb is None or dt < tb else tb
        blocked = {"value": g["nnz"] / tb, "unit": "edges/s", "cores": cores, "kind": "port",
                   "tiling": info,
                   "sample": "full workload, best of 2 passes incl. the per-call re-tiling; "
                             "DGL's libxsmm path restated (JIT row kernel -> vectorised add), "
                             "bit-identical to the naive kernel (tests/test_oracle_blocked.py)"}
        if use_ref:  # leave `out` = the reference's result for the parity check below
            one_pass(cores)
    except Exception as ex:  # pragma: no cover
        blocked = {"error": repr(ex)}
    # (iii) independent sanity lines: the same sum through scipy's CSR product (one
    # thread) and torch.sparse.mm (torch's CPU threads) on a BOUNDED sample — the first eighth of
    # the rows; both are checked against the reference's rows before their rate is reported
    sanity = {}
    try:
        import scipy.sparse as sp

        rows = (indptr.shape[0] - 1) // 8
        ip8 = indptr[: rows + 1]
        nz8 = int(ip8[-1])
        a = sp.csr_matrix((np.ones(nz8, dtype=xh.dtype), indices[:nz8], ip8), shape=(rows, xh.shape[0]))
        t0 = time.perf_counter()
        y = a @ xh
        dt = time.perf_counter() - t0
        ok = bool(np.allclose(y, out[:rows], rtol=1e-4, atol=1e-4)) if use_ref else None
        sanity["scipy_csr_matmul"] = {"value": nz8 / dt, "unit": "edges/s", "cores": 1, "agrees_with_reference": ok,
                                      "sample": "first %d rows (%d edges)" % (rows, nz8)}
        ta = torch.sparse_csr_tensor(torch.from_numpy(ip8.astype(np.int64)), torch.from_numpy(indices[:nz8].astype(np.int64)),
                                     torch.ones(nz8), size=(rows, xh.shape[0]))
        xt = torch.from_numpy(xh)
        torch.sparse.mm(ta, xt[:, :4].contiguous())  # warm-up
        t0 = time.perf_counter()
        yt = torch.sparse.mm(ta, xt)
        dt = time.perf_counter() - t0
        ok = bool(np.allclose(yt.numpy(), out[:rows], rtol=1e-4, atol=1e-4)) if use_ref else None
        sanity["torch_sparse_mm"] = {"value": nz8 / dt, "unit": "edges/s", "cores": torch.get_num_threads(),
                                     "agrees_with_reference": ok, "sample": "first %d rows (%d edges)" % (rows, nz8)}
    except Exception as ex:  # pragma: no cover
        sanity["error"] = repr(ex)
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    what = ("oracle/_ref: DGL's own SpMMCsr<kDGLCPU,int32,float> (SpMMSumCsrNaive, "
            "src/array/cpu/spmm.h:45-74) built from the reference sources; libxsmm JIT absent "
            "from the checkout" if use_ref else
            "oracle.copy_u_sum_csr = C/OpenMP restatement of DGL SpMMSumCsrNaive "
            "(oracle/_ref not built)")
    return {
        "value": g["nnz"] / best, "unit": "edges/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "cpu_model": cpu_model, "hardware_threads": ncpu,
        "sample": "full workload (%d edges, F=%d), best of %d passes (thread-count sweep over "
                  "%d hardware threads) after 1 warm-up; %s" % (g["nnz"], x.shape[1], reps, ncpu, what),
        "libxsmm_style_blocked": blocked,
        "sanity": sanity,
    }, out


def spawn_ranks(n, share_gpu=False, script=None):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks under
    torch.distributed.run on this node (one rank per GPU, rendezvous on 127.0.0.1) and return its
    exit code.  Refuses when fewer than N GPUs are visible (unless the gloo flow-test backend is
    selected, where the ranks share what is there)."""
    import socket
    import subprocess

    ndev = torch.cuda.device_count()
    if ndev < n and not share_gpu:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, ndev), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(script or __file__)] + sys.argv[1:]
    return subprocess.call(cmd, cwd=ROOT)


def time_events(fn, reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        fn()
        ev[k + 1].record()
    torch.cuda.synchronize()
    return [ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]


def merge_kernel_ms(fn, reps=5):
    """Average duration of the merge kernel inside `fn` (ONE dgla_spmm_csr call), from the HIP
    events the library records around it on the launch stream."""
    from dgl_amd import _capi

    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(reps)]
    for a, b in pairs:
        a.record()
        b.record()
    torch.cuda.synchronize()
    for a, b in pairs:
        _capi.set_profile_events(a, b)
        fn()
    torch.cuda.synchronize()
    _capi.set_profile_events(None, None)
    return float(np.mean([a.elapsed_time(b) for a, b in pairs]))


def single_gpu(args, dev, n, e, f):
    """N = 1: one dgla_spmm_csr call per step over the whole graph (split-row copy of X when the
    locality probe wants it + merge kernel + fix-up kernel), X treated as NEW on every step."""
    from dgl_amd import _capi

    g = synth_csr(n, n, e, args.variant, seed=20250824, device=dev)
    torch.manual_seed(12345)
    x = torch.rand(n, f, device=dev) + 1
    out = torch.empty(n, f, device=dev)
    csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out),
                     dtype=torch.uint8, device=dev)
    plan_valid = [False]

    def step():
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws,
                       plan_valid=plan_valid[0])
        plan_valid[0] = True

    ctx = {"g": g, "x": x, "out": out, "csr": csr, "ws": ws, "edges": e, "rows": n,
           "alg_bytes": algorithmic_bytes(n, e, f), "profile_in_step": True}
    return step, ctx


def exchange_profile(op, x_loc, out, dist, dev, reps=5):
    """Where a step's time goes on THIS rank, measured OUTSIDE the timed region (VERDICT r4 Next #8): the exchange
    alone (nothing to hide behind), then `reps` profiled steps (HIP events inside ShardedSpMM.step and, with the peer
    exchange, around the push on its own stream).  `overlap_frac` = the share of the exchange that the own-column launch
    hid: 1 - (time the caller's stream sat waiting for halo rows) / (exchange alone)."""
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    for _ in range(2):
        op.exchange_alone(x_loc)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        op.exchange_alone(x_loc)
    sync()
    dist.barrier()
    alone_ms = (time.perf_counter() - t0) / reps * 1e3
    op.profile(True)
    for _ in range(reps):
        op.step(x_loc, out)
    sync()
    prof = op.profile_summary()
    op.profile(False)
    dist.barrier()
    prof = {("%s_ms" % k if k != "profiled_steps" else k): (round(v, 4) if isinstance(v, float) else v)
            for k, v in prof.items()}
    prof["exchange_alone_ms"] = round(alone_ms, 4)
    prof["overlap_frac"] = round(max(0.0, min(1.0, 1.0 - prof.get("wait_ms", 0.0) / alone_ms)), 4) if alone_ms > 0 else None
    return prof


def close_exchange(ctx, dist):
    """Unmap / free the peer-mapped halo buffers of a finished run (all ranks together)."""
    ex = getattr(ctx.get("op"), "exchange", None) if ctx else None
    if ex is not None and hasattr(ex, "close"):
        torch.cuda.synchronize()
        dist.barrier()
        ex.check()
        ex.close()


def multi_gpu(args, dev, n, e, f, rank, world, dist, spmm=None):
    """N > 1: STRONG scaling of the one graph.  Every rank builds the same graph and features
    from the same seeds, rank 0 partitions the nodes (native k-way partitioner standing where
    METIS stands in the reference; `--partitioner range` = contiguous edge-balanced ranges),
    each rank cuts its own shard (own-column block + halo-column block) and the step is
    ShardedSpMM.step: halo all-to-all over RCCL overlapped with the own-column launch, then the
    halo-column launch accumulates."""
    from dgl_amd.parallel import (ShardedSpMM, partition_assignment, partition_rows, row_slice_csr,
                                  shard_from_partition)

    g = synth_csr(n, n, e, args.variant, seed=20250824, device=dev)
    torch.manual_seed(12345)
    x_full = torch.rand(n, f, device=dev) + 1
    t0 = time.perf_counter()
    part = torch.empty(n, dtype=torch.int64, device=dev)
    stats = {}
    # the assignment depends only on (graph generator inputs, k): kept on disk between runs of this
    # box (a SCALE series re-runs the same k only when repeated, but repeated runs are the common case
    # while tuning); a stale or unreadable file is ignored
    import tempfile

    cache = os.path.join(tempfile.gettempdir(), "dgl_amd_partition_v3_%s_n%d_e%d_k%d_%s.pt" % (
        args.variant, n, e, world, args.partitioner))
    if rank == 0:
        done = None
        if args.partitioner == "kway" and os.path.exists(cache):
            try:
                saved = torch.load(cache)
                if saved["part"].shape[0] == n:
                    part.copy_(saved["part"])
                    stats = dict(saved["stats"], cached=True)
                    done = (None, stats)
            except Exception:  # noqa: BLE001
                done = None
        if args.partitioner == "kway" and done is None:
            # the partitioner is host code (minutes on a graph of this size): run it in a daemon
            # thread with a time budget so that a slow host cannot stall the other ranks past the
            # collective time-out; past the budget the run falls back to contiguous ranges and says so
            import threading

            box = {}

            budget = getattr(args, "partition_budget", 240.0)

            def work():
                try:
                    # objtype="vol": what the exchange moves is DISTINCT remote rows, not cut edges
                    box["r"] = partition_assignment(g["indptr"], g["indices"], world, seed=1, objtype="vol")
                except BaseException as ex:  # re-raised on the main thread
                    box["e"] = ex

            th = threading.Thread(target=work, daemon=True)
            th.start()
            th.join(timeout=budget)
            if "e" in box:
                raise box["e"]
            done = box.get("r")
            if done is not None:
                part.copy_(done[0])
                stats = done[1]
                try:
                    torch.save({"part": done[0].cpu(), "stats": stats}, cache)
                except Exception:  # noqa: BLE001
                    pass
            else:
                stats = {"fallback": "k-way partitioner exceeded %.0f s: contiguous ranges used" % budget}
                print("bench.py: WARNING: %s (the line's config.partitioner_used says so too)" % stats["fallback"],
                      file=sys.stderr, flush=True)
        if done is None:
            bounds = partition_rows(g["indptr"].cpu(), world)
            part.copy_(torch.searchsorted(bounds[1:].contiguous(), torch.arange(n), right=True))
    dist.broadcast(part, src=0)
    t_part = time.perf_counter() - t0
    sh = shard_from_partition(g["indptr"], g["indices"], part, world, rank)
    # exchange: peer-mapped halo buffers written by the pack kernel (dgl_amd/peer_exchange.py; all ranks of one
    # node) unless --exchange alltoall asks for the RCCL all_to_all_single path (CPU flow tests pass spmm=...)
    use_peer = getattr(args, "exchange", "peer") == "peer" and spmm is None and dev.type == "cuda"
    peer_note = None
    try:
        op = ShardedSpMM(sh, (f,), x_full.dtype, dev, spmm=spmm,  # spmm=None: the library's kernels
                         chunks=max(1, int(getattr(args, "chunks", 1))), exchange="peer" if use_peer else None)
    except Exception as e:  # noqa: BLE001
        if not use_peer:
            raise
        # the peer set-up fails on EVERY rank together (dgl_amd/peer_exchange.py: agreement after each local step),
        # e.g. a driver without dmabuf IPC or GPUs that are not peers: run the RCCL all-to-all path and say so
        peer_note = "peer set-up failed, all_to_all_single used instead: %s" % (str(e)[:300],)
        if rank == 0:
            print("bench.py: WARNING: " + peer_note, file=sys.stderr, flush=True)
        use_peer = False
        op = ShardedSpMM(sh, (f,), x_full.dtype, dev, spmm=spmm, chunks=max(1, int(getattr(args, "chunks", 1))),
                         exchange=None)
    x_loc = x_full[sh["rows"]].contiguous()
    out = torch.empty(sh["n_local"], f, device=dev)

    def step():
        op.step(x_loc, out)

    # the same row shard against REPLICATED features (no exchange): timed after the K steps as a
    # variant ("... or features already resident (state which is measured)")
    rows_csr = row_slice_csr(g["indptr"], g["indices"], sh["rows"])
    out_rep = torch.empty(sh["n_local"], f, device=dev)

    def step_replicated():
        op.spmm("rows", rows_csr, n, x_full, out_rep, False)

    info = {"rank": rank, "rows": sh["n_local"], "edges": sh["nnz"], "cut_edges": sh["cut_edges"],
            "halo_rows": sh["n_halo"], "halo_bytes": sh["n_halo"] * f * 4}
    infos = [None] * world
    dist.all_gather_object(infos, info)
    ctx = {"g": g, "x": x_full, "out": out, "shard": sh, "op": op, "x_loc": x_loc, "edges": sh["nnz"],
           "rows": sh["n_local"], "alg_bytes": algorithmic_bytes(sh["n_local"], sh["nnz"], f),
           "profile_in_step": False, "infos": infos, "partition_s": t_part, "exchange": "peer" if use_peer else "alltoall", "exchange_note": peer_note,
           "partition_stats": stats, "step_replicated": step_replicated, "out_replicated": out_rep}
    return step, ctx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variant", default="U", choices=["U", "L", "C"],
                    help="column structure of the synthetic graph (tests/graphgen.py): U uniform, L local windows, C 64 planted "
                         "communities with shuffled ids")
    ap.add_argument("--scale", type=int, default=1, help="divide N and E (debug only)")
    ap.add_argument("--partitioner", default="kway", choices=["kway", "range"],
                    help="N>1: node partitioner (kway = native multilevel, range = contiguous rows)")
    ap.add_argument("--partition-budget", type=float, default=240.0,
                    help="N>1: seconds the k-way partitioner may take before contiguous ranges are used")
    ap.add_argument("--chunks", type=int, default=2,
                    help="N > 1: pipeline chunks of the halo exchange (chunk c's halo-column launch is queued "
                         "when chunk c has landed, while chunk c + 1 travels); 1 = one all-to-all per step")
    ap.add_argument("--exchange", default="peer", choices=["peer", "alltoall"],
                    help="N > 1: halo exchange — peer = the pack kernel writes into the peers' IPC-mapped halo buffers "
                         "and flags (no collective); alltoall = pack + all_to_all_single over RCCL")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-peak", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra variant-L / static-feature timings of the N=1 line")
    ap.add_argument("--extra", action="store_true", help="also time int64 ids")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback)")
    share_gpu = os.environ.get("DGLA_BENCH_BACKEND", "nccl") == "gloo"  # flow tests: ranks share a GPU
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one per GPU, like the
        # reference's multi-GPU bench spawns its own workers
        # (benchmarks/benchmarks/multigpu/bench_multigpu_sage.py:176-186)
        raise SystemExit(spawn_ranks(args.gpus, share_gpu))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks"
                         % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not share_gpu:
            raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible); one rank per GPU"
                             % (local_rank, ndev))
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # DGLA_BENCH_BACKEND=gloo: flow test of this file with several ranks on ONE GPU
        # (tests/test_gpu_bench_multi.py); the exchange is then staged through host memory
        if os.environ.get("DGLA_BENCH_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from dgl_amd import _capi

    n = C2_NODES // args.scale
    e = C2_EDGES // args.scale
    f = C2_FEAT
    if world == 1:
        step, ctx = single_gpu(args, dev, n, e, f)
    else:
        step, ctx = multi_gpu(args, dev, n, e, f, rank, world, dist)

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
        if ctx["profile_in_step"]:
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

    # ---- dominant-kernel duration (HIP events on the launch stream) -------------------
    if world == 1:
        kern_ms = [a.elapsed_time(b) for a, b in k_ev]
        kern_avg, kern_min = float(np.mean(kern_ms)), float(np.min(kern_ms))
        kernel_name = "spmm_csr_merge_kernel<int,float,VEC=4,copy_lhs,sum>"
    else:
        # a step has two launches of the merge kernel (own-column block, halo-column block):
        # time each alone, outside the timed region; the roofline is quoted on their sum
        op, sh = ctx["op"], ctx["shard"]
        scratch = torch.empty_like(ctx["out"])  # (the halo launch accumulates: keep the step's result intact)
        t_loc = merge_kernel_ms(lambda: op.spmm("local", sh["local"], sh["n_local"], ctx["x_loc"],
                                                scratch, False))
        halo_t = op.halo if op.halo is not None else op.exchange._halo[op.exchange.epoch & 1]
        t_halo = merge_kernel_ms(lambda: op.spmm("halo", sh["halo"], sh["n_halo"], halo_t,
                                                 scratch, True)) if sh["n_halo"] else 0.0
        del scratch
        kern_avg = kern_min = t_loc + t_halo
        kernel_name = "spmm_csr_merge_kernel<int,float,VEC=4,copy_lhs,sum> x2 (own-column + halo-column block)"

    profs = None
    if world > 1:
        mine = exchange_profile(ctx["op"], ctx["x_loc"], ctx["out"], dist, dev)
        mine["rank"] = rank
        mine["halo_MB"] = round(ctx["shard"]["n_halo"] * f * 4 / 1e6, 3)
        profs = [None] * world
        dist.all_gather_object(profs, mine)

    result = None
    if rank == 0:
        achieved = ctx["alg_bytes"] / (kern_avg * 1e-3) / 1e9
        kernel_only = achieved
        if world > 1:
            # N > 1: the roofline of the WHOLE STEP — exchange included: the job's algorithmic bytes over the
            # max-over-ranks step time, per GPU (VERDICT r3 Weak #7: "two launches timed alone" looked healthy
            # whatever the exchange cost)
            achieved = algorithmic_bytes(n, e, f) / (ms_per_step * 1e-3) / 1e9 / world
        result = {
            "metric": "edges/sec for g-SpMM copy_u+sum (feat=100); % HBM roofline",
            "value": e / (ms_per_step * 1e-3),
            "unit": "edges/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: g-SpMM copy_u+sum on ogbn-products-shaped CSR "
                            "(N=%d rows, E=%d edges, feat=%d) fp32, int32 ids, variant %s"
                            % (n, e, f, args.variant),
                "step": "one dgla_spmm_csr call over the whole graph, X treated as new on every "
                        "step (split-row copy of X when the locality probe wants it, merge kernel, "
                        "fix-up kernel)" if world == 1 else
                        ("ShardedSpMM.step on every rank: ONE pack launch writes the requested rows into the peers' "
                         "IPC-mapped halo buffers (%d chunk(s), flags), own-column launch, then per chunk a one-wavefront "
                         "flag wait + halo-column launch (accumulate)" if ctx.get("exchange") == "peer" else
                         "ShardedSpMM.step on every rank: pack + halo all-to-all (RCCL, %d pipeline chunk(s)) "
                         "overlapped with the own-column launch, then the halo-column launch(es) accumulate")
                        % max(1, int(getattr(args, "chunks", 1))),
                "parallelism": "1 GPU" if world == 1 else
                               "%d-way node partition (%s), destination rows + features sharded, halo rows %s"
                               % (world, args.partitioner, "written peer-to-peer into IPC-mapped buffers"
                                  if ctx.get("exchange") == "peer" else "pulled by all_to_all_single over RCCL"),
                "tuning_flags": _capi.get_tuning(),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                "kernel": kernel_name,
                "kernel_avg_ms": kern_avg, "kernel_min_ms": kern_min,
                "algorithmic_bytes_per_launch": ctx["alg_bytes"],
            },
        }
        if world > 1:
            infos = ctx["infos"]
            cut = sum(i["cut_edges"] for i in infos)
            result["config"].update({
                "cut_fraction": cut / e, "partition_seconds": ctx["partition_s"],
                "partition_stats": {k: (int(v) if hasattr(v, "__int__") and not isinstance(v, (str, float)) else v)
                                    for k, v in (ctx["partition_stats"] or {}).items()},
                "per_rank": infos,
                "halo_rows_max": max(i["halo_rows"] for i in infos),
                "exchange_bytes_per_step_max_rank": max(i["halo_bytes"] for i in infos),
            })
            result["config"]["partitioner_used"] = (ctx["partition_stats"] or {}).get(
                "fallback", (ctx["partition_stats"] or {}).get("method", args.partitioner))
            # what the exchange costs and how much of it hides (measured after the timed region, same shards):
            # exchange_ms = the exchange with nothing to overlap (max over ranks); wait_ms = what a step still waits
            # for halo rows (max over ranks); overlap_frac = 1 - wait / exchange (min over ranks)
            result["exchange_profile"] = {
                "exchange_ms": max(p["exchange_alone_ms"] for p in profs),
                "push_ms": max((p.get("push_ms") or 0.0) for p in profs) or None,
                "wait_ms": max(p.get("wait_ms", 0.0) for p in profs),
                "local_ms": max(p.get("local_ms", 0.0) for p in profs),
                "halo_launch_ms": max(p.get("halo_ms", 0.0) for p in profs),
                "overlap_frac": min((p["overlap_frac"] for p in profs if p["overlap_frac"] is not None), default=None),
                "halo_MB_per_rank": [p["halo_MB"] for p in profs],
                "per_rank": profs,
                "note": "HIP events inside ShardedSpMM.step over %d profiled steps outside the timed region; "
                        "exchange_ms = exchange alone between barriers" % profs[0].get("profiled_steps", 0),
            }
            result["config"]["exchange"] = ctx.get("exchange")
            if ctx.get("exchange_note"):
                result["config"]["exchange_note"] = ctx["exchange_note"]
            result["roofline"]["note"] = ("whole step, exchange included: the job's algorithmic bytes / max-over-ranks "
                                          "step time / n_gpus; `kernel` fields = rank 0's two merge launches timed alone")
            result["roofline"]["kernel_only_achieved_rank0"] = kernel_only
            result["roofline"]["kernel_only_frac_rank0"] = kernel_only / HBM_PEAK_GBPS
            result["roofline"]["algorithmic_bytes_per_launch"] = algorithmic_bytes(n, e, f)

    if rank == 0 and world == 1 and args.scale == 1 and args.variant == "U":
        t, src = pmc_traffic()
        result["roofline"]["traffic"] = t
        result["roofline"]["traffic_source"] = src

    # ---- N > 1: every rank checks its rows against the one-launch result on the whole graph
    if world > 1:
        sh = ctx["shard"]
        g, xf = ctx["g"], ctx["x"]
        full = torch.empty(n, f, device=dev)
        csr = _capi.make_csr(g["indptr"], g["indices"], None, n)
        ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, xf.dtype, xf, None, full),
                         dtype=torch.uint8, device=dev)
        _capi.spmm_csr("copy_lhs", "sum", csr, xf, None, full, None, None, ws)
        ref = full[sh["rows"]]
        err = ((ctx["out"] - ref).abs() / ref.abs().clamp_min(1e-30)).max()
        dist.all_reduce(err, op=dist.ReduceOp.MAX)
        # variant: features replicated on every GPU, rows sharded, no exchange — K steps between
        # barriers like the headline region
        rep = ctx["step_replicated"]
        for _ in range(max(args.warmup, 1)):
            rep()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rep()
        torch.cuda.synchronize()
        dist.barrier()
        t_rep = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t_rep, op=dist.ReduceOp.MAX)
        err_rep = ((ctx["out_replicated"] - ref).abs() / ref.abs().clamp_min(1e-30)).max()
        dist.all_reduce(err_rep, op=dist.ReduceOp.MAX)
        if rank == 0:
            result["parity_max_rel_err_vs_single_gpu_launch"] = float(err)
            ms_rep = float(t_rep.item()) / args.steps * 1e3
            # both readings of "edges/s at N GPUs" side by side ("state which is measured"): `value` is
            # the SHARDED-features job (halo exchange inside the step); resident = every GPU holds all source features
            result["value_sharded_features"] = result["value"]
            result["value_resident_features"] = e / (ms_rep * 1e-3)
            result["variants"] = {"features_replicated_no_exchange": {
                "ms_per_step": ms_rep, "edges_per_s": e / (ms_rep * 1e-3),
                "parity_max_rel_err_vs_single_gpu_launch": float(err_rep),
                "note": "same %d-way row shards, source features resident on every GPU (static input "
                        "features), no collective in the step" % world}}
        # variant: the SAME shards with the halo rows pulled by RCCL all_to_all_single instead of written peer-to-peer
        # (north_star: "halo all-gather on RCCL over xGMI"; reference analogue python/dgl/cuda/nccl.py:98-183) — K steps
        # between barriers like the headline, so that the two transports sit side by side in one line and RCCL moves
        # real halo rows across all N ranks (VERDICT r5 Next #6a)
        if not args.no_variants and ctx.get("exchange") == "peer":
            from dgl_amd.parallel import ShardedSpMM

            op_r = ShardedSpMM(sh, (f,), xf.dtype, dev, chunks=max(1, int(getattr(args, "chunks", 1))), exchange=None)
            out_r = torch.empty_like(ctx["out"])
            for _ in range(max(args.warmup, 1)):
                op_r.step(ctx["x_loc"], out_r)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                op_r.step(ctx["x_loc"], out_r)
            torch.cuda.synchronize()
            dist.barrier()
            t_r = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(t_r, op=dist.ReduceOp.MAX)
            err_r = ((out_r - ref).abs() / ref.abs().clamp_min(1e-30)).max()
            dist.all_reduce(err_r, op=dist.ReduceOp.MAX)
            same = torch.tensor([1.0 if torch.equal(out_r, ctx["out"]) else 0.0], device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if rank == 0:
                ms_r = float(t_r.item()) / args.steps * 1e3
                result["variants"]["exchange_alltoall_rccl"] = {
                    "ms_per_step": ms_r, "edges_per_s": e / (ms_r * 1e-3),
                    "parity_max_rel_err_vs_single_gpu_launch": float(err_r),
                    "bits_equal_to_peer_exchange_result": bool(same.item() == 1.0),
                    "backend": dist.get_backend(),
                    "note": "same %d-way shards and schedule, halo rows packed and exchanged with all_to_all_single "
                            "(%d pipeline chunk(s)) instead of peer-mapped writes" % (world, max(1, int(getattr(args, "chunks", 1))))}
            del op_r, out_r
        # variant: the FEATURE axis sharded instead of the rows — every rank holds the whole graph
        # and a slice of the feature columns (widths multiples of 4), no exchange at all; the
        # output comes out column-sharded.  Not the north_star's node-cut: what the same kernels
        # do when the communication is moved out of the SpMM (narrow rows gather less efficiently).
        if not args.no_variants:
            per = -(-f // world // 4) * 4 if f % 4 == 0 else -(-f // world)  # 16 of 100 columns at 8 ranks: 64-byte rows
            w4 = [max(0, min(per, f - r * per)) for r in range(world)]
            c0 = sum(w4[:rank])
            xc = xf[:, c0:c0 + w4[rank]].contiguous()
            oc = torch.empty(n, w4[rank], device=dev)
            if w4[rank] > 0:
                wsc = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, xc.dtype, xc, None, oc),
                                  dtype=torch.uint8, device=dev)
                pv = [False]

                def step_cols():
                    _capi.spmm_csr("copy_lhs", "sum", csr, xc, None, oc, None, None, wsc, plan_valid=pv[0])
                    pv[0] = True
            else:
                def step_cols():
                    pass
            for _ in range(max(args.warmup, 1)):
                step_cols()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_cols()
            torch.cuda.synchronize()
            dist.barrier()
            t_c = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(t_c, op=dist.ReduceOp.MAX)
            rc = full[:, c0:c0 + w4[rank]]
            err_c = ((oc - rc).abs() / rc.abs().clamp_min(1e-30)).max() if w4[rank] else torch.zeros((), device=dev)
            dist.all_reduce(err_c, op=dist.ReduceOp.MAX)
            if rank == 0:
                ms_c = float(t_c.item()) / args.steps * 1e3
                result["variants"]["feature_columns_sharded_no_exchange"] = {
                    "ms_per_step": ms_c, "edges_per_s": e / (ms_c * 1e-3), "column_widths": w4,
                    "parity_max_rel_err_vs_single_gpu_launch": float(err_c),
                    "note": "whole graph on every GPU, feature columns sharded (output column-sharded), "
                            "no collective in the step"}
            del xc, oc
        del full, ws
        # variant L (80 % of a row's neighbours within +-32 k rows) with contiguous row ranges — the
        # partition a locality-ordered graph gets for free: same schedule, smaller halo
        if args.variant == "U" and not args.no_variants:
            import copy

            close_exchange(ctx, dist)
            del ctx, step, rep
            torch.cuda.empty_cache()
            a2 = copy.copy(args)
            a2.variant, a2.partitioner = "L", "range"
            step_l, ctx_l = multi_gpu(a2, dev, n, e, f, rank, world, dist)
            for _ in range(max(args.warmup, 1)):
                step_l()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_l()
            torch.cuda.synchronize()
            dist.barrier()
            t_l = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(t_l, op=dist.ReduceOp.MAX)
            if rank == 0:
                ms_l = float(t_l.item()) / args.steps * 1e3
                infos = ctx_l["infos"]
                result["variants"]["L_range_partition_sharded_features"] = {
                    "ms_per_step": ms_l, "edges_per_s": e / (ms_l * 1e-3),
                    "cut_fraction": sum(i["cut_edges"] for i in infos) / e,
                    "halo_rows_max": max(i["halo_rows"] for i in infos),
                    "exchange_bytes_per_step_max_rank": max(i["halo_bytes"] for i in infos),
                    "note": "variant L, contiguous row ranges, the same ShardedSpMM.step (halo all-to-all "
                            "overlapped with the own-column launch)"}
            close_exchange(ctx_l, dist)
            del ctx_l, step_l
            # variant C (64 planted communities, ids shuffled) with the k-way partitioner: the graph a node-cut
            # partitioner actually helps — ranges cut it like U, the partitioner finds the communities
            torch.cuda.empty_cache()
            a3 = copy.copy(args)
            a3.variant, a3.partitioner = "C", "kway"
            step_c, ctx_c = multi_gpu(a3, dev, n, e, f, rank, world, dist)
            for _ in range(max(args.warmup, 1)):
                step_c()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_c()
            torch.cuda.synchronize()
            dist.barrier()
            t_c3 = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(t_c3, op=dist.ReduceOp.MAX)
            if rank == 0:
                ms_c3 = float(t_c3.item()) / args.steps * 1e3
                infos = ctx_c["infos"]
                pst = ctx_c["partition_stats"] or {}
                result["variants"]["C_kway_partition_sharded_features"] = {
                    "ms_per_step": ms_c3, "edges_per_s": e / (ms_c3 * 1e-3),
                    "cut_fraction": sum(i["cut_edges"] for i in infos) / e,
                    "volume_rows": pst.get("volume"), "halo_rows_max": max(i["halo_rows"] for i in infos),
                    "exchange_bytes_per_step_max_rank": max(i["halo_bytes"] for i in infos),
                    "partition_seconds": ctx_c["partition_s"],
                    "partitioner_used": pst.get("fallback", pst.get("method", "kway")),
                    "exchange": ctx_c.get("exchange"),
                    "note": "variant C: 64 planted communities, node ids shuffled; k-way partition (objtype vol), "
                            "the same ShardedSpMM.step"}
            close_exchange(ctx_c, dist)
            del ctx_c, step_c

    # ---- extras on rank 0, outside the timed region ----------------------------------
    if rank == 0 and world == 1:
        if not args.no_peak:
            try:
                peak, rpeak, which = measure_copy_peak(dev, nbytes=(4 << 30) // max(1, args.scale))
                result["roofline"]["measured_copy_peak"] = peak
                result["roofline"]["measured_read_peak"] = rpeak
                result["roofline"]["measured_peak_kernels"] = which
                result["roofline"]["frac_of_measured_copy_peak"] = result["roofline"]["achieved"] / peak
                result["roofline"]["frac_of_measured_read_peak"] = result["roofline"]["achieved"] / rpeak
                result["roofline"]["measured_peak_note"] = (
                    "streaming peaks of this box (float4, 4 GiB); `achieved` counts ALGORITHMIC bytes of a "
                    "gather whose 16-byte row tails are served by the Infinity Cache, so a fraction near or "
                    "above 1 of a streaming peak is possible and is not an HBM-efficiency claim")
            except Exception as ex:  # pragma: no cover
                result["roofline"]["measured_copy_peak_error"] = repr(ex)
        if not args.no_variants:
            result["variants"] = variants(dev, ctx, n, e, f, args)
            # the other operators north_star names, on this same clock (benchmarks/hot_path_variants.py): u_mul_e_sum,
            # SDDMM u_dot_v, edge softmax, the GAT attention block, the stacked bf16 R-GCN launch — each with its
            # algorithmic bytes and roofline fraction; timed outside the K steps
            if args.variant == "U":
                try:
                    from benchmarks.hot_path_variants import op_variants

                    t_ops = time.perf_counter()
                    result["variants"].update(op_variants(dev, ctx["g"], args.scale))
                    result["variants"]["op_variants_seconds"] = round(time.perf_counter() - t_ops, 1)
                except Exception as ex:  # pragma: no cover  (the headline line must still be printed)
                    result["variants"]["op_variants_error"] = repr(ex)
        if not args.no_cpu:
            cb, ref = cpu_baseline(ctx["g"], ctx["x"])
            result["cpu_baseline"] = cb
            # the full-size run doubles as a parity check against the oracle
            err = (ctx["out"].cpu().numpy() - ref) / np.maximum(np.abs(ref), 1e-30)
            result["parity_max_rel_err_vs_oracle"] = float(np.abs(err).max())
            del ref
        if args.extra:
            result["extra"] = extras(dev, n, e, f, args)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def variants(dev, ctx, n, e, f, args, reps=10):
    """Same kernel, other situations, timed outside the K steps (median of `reps` whole calls):
    the locality variant L of the graph, and static features (the split-row copy of X kept
    between calls: DGLA_SPLIT_KEEP / _VALID, what dgl_amd.static_features(x) selects)."""
    from dgl_amd import _capi

    res = {}
    b_alg = algorithmic_bytes(n, e, f)

    def line(ts, kern=None):
        med = float(np.median(ts))
        d = {"ms_per_call_median": med, "edges_per_s": e / (med * 1e-3)}
        if kern is not None:
            d["merge_kernel_ms"] = kern
            d["roofline_frac"] = b_alg / (kern * 1e-3) / 1e9 / HBM_PEAK_GBPS
        return d

    def run(csr, x, out, ws, **kw):
        _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws, **kw)  # plan, copy
        call = lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out, None, None, ws,
                                      plan_valid=True, **kw)
        call()
        torch.cuda.synchronize()
        return line(time_events(call, reps), merge_kernel_ms(call))

    x, out = ctx["x"], ctx["out"]
    if args.variant == "U":
        _capi.spmm_csr("copy_lhs", "sum", ctx["csr"], x, None, out, None, None, ctx["ws"],
                       plan_valid=True, split_keep=True)   # makes the copy; kept from here on
        call = lambda: _capi.spmm_csr("copy_lhs", "sum", ctx["csr"], x, None, out, None, None, ctx["ws"],
                                      plan_valid=True, split_keep=True, split_valid=True)
        call()
        torch.cuda.synchronize()
        res["U_static_features"] = line(time_events(call, reps), merge_kernel_ms(call))
        # the same consumer call when the PRODUCER of X prepared the operand (DGLA_PREPARE_ONLY on its own stream, as
        # part of finishing the tensor): what the producer pays is reported next to it
        prep = lambda: _capi.spmm_csr("copy_lhs", "sum", ctx["csr"], x, None, out, None, None, ctx["ws"],
                                      plan_valid=True, split_keep=True, prepare_only=True)
        prep()
        torch.cuda.synchronize()
        # MEASURED as the sequence it is: the producer's prepare on a stream of its own, then — ordered
        # after it — the consumer's split_valid call on the caller's stream; both durations from HIP events
        side = torch.cuda.Stream()
        cons_ms, prep_ms = [], []
        for _ in range(reps):
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                p0.record()
                prep()
                p1.record()
            torch.cuda.current_stream().wait_stream(side)
            c0.record()
            call()
            c1.record()
            torch.cuda.synchronize()
            prep_ms.append(p0.elapsed_time(p1))
            cons_ms.append(c0.elapsed_time(c1))
        res["U_operand_prepared_by_producer"] = dict(line(cons_ms, merge_kernel_ms(call)),
                                                     producer_prepare_ms=float(np.median(prep_ms)),
                                                     note="consumer call timed after a prepare_only call on a producer "
                                                          "stream, every repetition")
        # leave `out` as the per-call path produced it (bit-identical anyway)
    gl = synth_csr(n, n, e, "L" if args.variant == "U" else "U", seed=20250824, device=dev)
    csr = _capi.make_csr(gl["indptr"], gl["indices"], None, n)
    out2 = torch.empty_like(out)
    ws = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr, x.dtype, x, None, out2),
                     dtype=torch.uint8, device=dev)
    other = "L" if args.variant == "U" else "U"
    res[other + "_int32"] = run(csr, x, out2, ws)
    call = lambda: _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out2, None, None, ws,
                                  plan_valid=True, split_keep=True, split_valid=True)
    _capi.spmm_csr("copy_lhs", "sum", csr, x, None, out2, None, None, ws, plan_valid=True, split_keep=True)
    call()
    torch.cuda.synchronize()
    res[other + "_static_features"] = line(time_events(call, reps), merge_kernel_ms(call))
    del gl, csr, ws
    # DGL's default id type: the same graph with int64 ids (408 algorithmic bytes per edge)
    g64 = synth_csr(n, n, e, args.variant, seed=20250824, device=dev, idtype=torch.int64)
    csr64 = _capi.make_csr(g64["indptr"], g64["indices"], None, n)
    ws64 = torch.empty(_capi.spmm_csr_workspace_bytes("copy_lhs", "sum", csr64, x.dtype, x, None, out2),
                       dtype=torch.uint8, device=dev)
    d64 = run(csr64, x, out2, ws64)
    if "merge_kernel_ms" in d64:
        d64["roofline_frac"] = algorithmic_bytes(n, e, f, i=8) / (d64["merge_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
    res[args.variant + "_int64"] = d64
    return res


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
