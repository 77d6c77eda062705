"""Halo exchange through PEER-MAPPED buffers (one process per GPU, all GPUs of one node).

Replaces, for the per-step value exchange, ``sparse_all_to_all_pull`` of the reference
(python/dgl/cuda/nccl.py:98-183: three NCCL all-to-alls per call) and round 3's
pack -> send buffer -> ``all_to_all_single`` -> halo buffer (dgl_amd/parallel.py ``HaloExchange``):

  * set-up (once per graph): the request lists are exchanged as before; every rank allocates its two
    halo buffers (step parity) and its flag words with ``dgla_peer_alloc``, exports them with
    ``hipIpcGetMemHandle`` and opens every peer's (``hipIpcOpenMemHandle``: xGMI peer access);
  * per step: ONE pack launch (``dgla_peer_push``) writes every requested row straight into the halo
    buffer of the rank that wants it — chunk by chunk — and bumps that rank's flag for (owner, chunk)
    to the step number; the consumer enqueues ``dgla_peer_wait`` (one wavefront, bounded spin) in
    front of the halo-column launch of each chunk.  No send buffer, no collective, no host wait.

xGMI is point-to-point: the seven destination streams of one pack launch use the seven links at once.
Correctness of the hand-off (system-scope release / acquire, monotone flags, two buffers against
write-after-read) is argued next to ``peer_push_kernel`` in csrc/exchange.hip.  The pool this was built
on has single-GPU boxes only: the path is tested with several ranks SHARING one GPU (IPC works
same-device), i.e. everything except the xGMI hop itself.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from ._lib import LIB, DGLAMDError, check_call
from .parallel import HaloExchange

_SEG_WORDS = 5   # PeerSegment {row_begin, row_end, dst, flag, blk_begin}: five 8-byte words (csrc/exchange.hip)
_SELF_FLAG = 1 << 62


class _RawBuffer:
    """torch view of raw device memory through ``__cuda_array_interface__``."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


_TYPESTR = {torch.float32: "<f4", torch.float64: "<f8", torch.float16: "<f2", torch.int64: "<i8",
            torch.int32: "<i4", torch.uint8: "|u1"}


def _view(ptr, shape, dtype, device):
    if dtype == torch.bfloat16:   # no typestr for bf16: view the bytes as int16 first
        return torch.as_tensor(_RawBuffer(ptr, shape, "<i2"), device=device).view(torch.bfloat16)
    return torch.as_tensor(_RawBuffer(ptr, shape, _TYPESTR[dtype]), device=device)


def _gather_bytes(payload, device, group):
    """Every rank's ``payload`` (equal lengths), through the process group's own backend."""
    world = dist.get_world_size(group)
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    on_gpu = dist.get_backend(group) != "gloo"
    if on_gpu:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


def _err_field(err):
    """Fixed-size status field of the set-up agreements: one flag byte + up to 200 bytes of message."""
    return (b"\x01" if err else b"\x00") + err.encode("utf-8", "replace")[:200].ljust(200, b"\0")


class PeerHaloExchange:
    """Same role as :class:`dgl_amd.parallel.HaloExchange` for ``ShardedSpMM``; the halo buffers belong to
    the exchange (they are what the peers write into).  ``begin_step(x_local)`` starts the push of this
    rank's rows and returns the halo tensor of the step; ``wait_chunk(c)`` orders the caller's stream
    after chunk ``c`` of every peer."""

    def __init__(self, n_local, n_halo, feat_shape, dtype, device, requests, group=None, chunks=1,
                 max_spins=1 << 24):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        self.dtype = dtype
        self.feat_shape = tuple(int(d) for d in feat_shape)
        feat = 1
        for d in self.feat_shape:
            feat *= d
        self.row_bytes = feat * torch.empty(0, dtype=dtype).element_size()
        if self.row_bytes % 4:
            raise DGLAMDError("peer exchange: feature rows must be a whole number of 4-byte words")
        # request lists, serve lists and the chunk-major layout: exchanged exactly as for the all-to-all path
        idx = HaloExchange(n_local, n_halo, feat, device, requests=requests, group=group, chunks=chunks)
        self.n_local, self.n_halo, self.chunks = idx.n_local, idx.n_halo, idx.chunks
        self.chunk_bounds, self.halo_old2new = idx.chunk_bounds, idx.halo_old2new
        self.recv_splits, self.send_splits = idx.recv_splits, idx.send_splits
        self.serve_rows = idx.serve_rows.to(torch.int64).contiguous()
        C, W = self.chunks, self.world
        # where does (chunk c, owner q) start in MY halo buffer?  tell every owner its C offsets
        mine = torch.zeros(W, C, dtype=torch.int64)
        for c in range(C):
            off = self.chunk_bounds[c]
            for q in range(W):
                mine[q, c] = off
                off += idx.recv_pieces[c][q]
        theirs = torch.empty_like(mine)
        from .parallel import _all_to_all
        m_dev = mine.reshape(-1).to(self.device)
        t_dev = torch.empty_like(m_dev)
        _all_to_all(t_dev, m_dev, None, None, group)
        theirs = t_dev.cpu().reshape(W, C)       # theirs[p, c] = row offset of MY block of chunk c in p's halo buffer
        # ---- allocate + export + import ------------------------------------------------------------------
        # Every step that can fail LOCALLY (allocation, hipIpcGetMemHandle, hipIpcOpenMemHandle: a driver without
        # dmabuf IPC, GPUs that are not peers) is followed by an agreement over the group, so that all ranks raise
        # together — a rank that raised alone would leave the others inside the next collective — and the caller
        # (bench.py) can fall back to the all-to-all path on every rank.
        kind = {"plain": 0, "finegrained": 1, "uncached": 2}[os.environ.get("DGLA_PEER_ALLOC", "finegrained")]
        self._halo_bytes = max(self.n_halo * self.row_bytes, 256)
        self._halo_bytes = (self._halo_bytes + 255) // 256 * 256
        self._owned = []
        self._imported = []
        handles = bytearray(128)
        err = ""
        halo_ptr = flag_ptr = 0
        try:
            if os.environ.get("DGLA_PEER_FAIL_RANK", "") == str(self.rank):      # fault injection for the tests
                raise DGLAMDError("injected failure (DGLA_PEER_FAIL_RANK)")
            if kind == 0 and not self._ranks_share_one_device(group):
                raise DGLAMDError("DGLA_PEER_ALLOC=plain (coarse-grained hipMalloc) is only coherent between ranks "
                                  "that share ONE GPU; these ranks sit on different devices")
            halo_ptr = self._alloc(2 * self._halo_bytes, kind)
            flag_ptr = self._alloc(8 * C * W + 256, kind)
            self._flags = _view(flag_ptr, (C * W,), torch.int64, self.device)
            self._flags.zero_()
            self._flags.view(C, W)[:, self.rank] = _SELF_FLAG      # nobody writes my own column: never waited for
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
            torch.cuda.synchronize(self.device)
            buf = (ctypes.c_char * 128).from_buffer(handles)
            check_call(LIB.dgla_ipc_export(ctypes.c_void_p(halo_ptr), ctypes.addressof(buf)))
            check_call(LIB.dgla_ipc_export(ctypes.c_void_p(flag_ptr), ctypes.addressof(buf) + 64))
            del buf
        except Exception as e:  # noqa: BLE001 — reported to every rank below
            err = "rank %d: %s" % (self.rank, e)
        self._halo_ptr, self._flag_ptr = halo_ptr, flag_ptr
        everyone = _gather_bytes(bytes(handles) + os.getpid().to_bytes(8, "little") +
                                 self._halo_bytes.to_bytes(8, "little") + _err_field(err), self.device, group)
        self._agree([e[144:] for e in everyone], "allocating / exporting the halo buffers")
        peer_halo, peer_flag = [0] * W, [0] * W
        # (every rank sizes its halo buffers by ITS halo: the second buffer of rank p starts peer_half[p] bytes in)
        peer_half = [int.from_bytes(everyone[p][136:144], "little") for p in range(W)]
        try:
            for p in range(W):
                if p == self.rank:
                    peer_halo[p], peer_flag[p] = halo_ptr, flag_ptr
                    continue
                if int.from_bytes(everyone[p][128:136], "little") == os.getpid():
                    raise DGLAMDError("peer exchange: two ranks in one process")
                for off, dst in ((0, peer_halo), (64, peer_flag)):
                    out = ctypes.c_void_p()
                    h = (ctypes.c_char * 64).from_buffer_copy(everyone[p][off:off + 64])
                    check_call(LIB.dgla_ipc_import(ctypes.addressof(h), ctypes.byref(out)))
                    dst[p] = int(out.value)
                    self._imported.append(int(out.value))
        except Exception as e:  # noqa: BLE001
            err = "rank %d: %s" % (self.rank, e)
        self._agree(_gather_bytes(_err_field(err), self.device, group), "opening the peers' halo buffers")
        # ---- segment tables, one per step parity: chunk-major, destination ranks ascending ------------------
        rows, blk = [], 0
        for c in range(C):
            r0 = idx.serve_bounds[c]
            for p in range(W):
                n = idx.send_pieces[c][p]
                if p != self.rank:
                    rows.append((r0, r0 + n, p, c, blk))
                    blk += max(1, -(-n // 64))
                r0 += n
        self._nseg, self._nblk = len(rows), blk
        self._segs = []
        for parity in (0, 1):
            t = torch.zeros(max(self._nseg, 1), _SEG_WORDS, dtype=torch.int64)
            for i, (a, b, p, c, bb) in enumerate(rows):
                t[i, 0], t[i, 1] = a, b
                t[i, 2] = peer_halo[p] + parity * peer_half[p] + int(theirs[p, c]) * self.row_bytes
                t[i, 3] = peer_flag[p] + 8 * (c * W + self.rank)
                t[i, 4] = bb
            self._segs.append(t.to(self.device))
        self._arrive = torch.zeros(max(self._nseg, 1), dtype=torch.int32, device=self.device)
        self._halo = [_view(halo_ptr + q * self._halo_bytes, (self.n_halo,) + self.feat_shape, dtype, self.device)
                      for q in (0, 1)]
        self.epoch = 0
        self.max_spins = int(max_spins)
        self._push_stream = torch.cuda.Stream(self.device)
        self._push_done = torch.cuda.Event()
        self._pending = False
        dist.barrier(group=group)   # every rank has opened every buffer before anyone writes

    def _ranks_share_one_device(self, group):
        """True when every rank of the group drives the same physical GPU (the single-GPU test set-up)."""
        props = torch.cuda.get_device_properties(self.device)
        ident = ("%s|%s" % (getattr(props, "uuid", ""), getattr(props, "pci_bus_id", self.device.index))).encode()
        ident = (ident + b"\0" * 96)[:96]
        return len(set(_gather_bytes(ident, self.device, group))) == 1

    def _agree(self, fields, what):
        """All ranks raise together — with every failing rank's message — when any of them failed `what`."""
        bad = [(p, f[1:].rstrip(b"\0").decode("utf-8", "replace")) for p, f in enumerate(fields) if f[0]]
        if bad:
            self.close_quiet()
            raise DGLAMDError("peer exchange: %s failed on rank(s) %s: %s" % (what, [p for p, _ in bad],
                                                                                "; ".join(m for _, m in bad)))

    def close_quiet(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _alloc(self, nbytes, kind):
        out = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check_call(LIB.dgla_peer_alloc(int(nbytes), int(kind), ctypes.byref(out)))
        self._owned.append(int(out.value))
        return int(out.value)

    # -- per step ---------------------------------------------------------------------------------------------
    def begin_step(self, x_local):
        """Push this rank's requested rows into the peers' halo buffers (one launch on the current stream,
        which also orders it after the producer of ``x_local``) and return MY halo tensor of this step."""
        if not x_local.is_cuda or not x_local.is_contiguous() or x_local.dtype != self.dtype:
            raise DGLAMDError("peer exchange: x_local must be a contiguous tensor of the exchange's dtype on the GPU")
        self.epoch += 1
        parity = self.epoch & 1
        if self._nseg:
            # the push runs on a stream of its own, ordered after the producer of x_local: it is bound by the
            # xGMI links, not by this GPU, and overlaps the own-column launch the caller queues next
            cur = torch.cuda.current_stream(self.device)
            self._push_stream.wait_stream(cur)
            prof = getattr(self, "_prof", None)
            if prof is not None:       # ShardedSpMM.profile(): the push as the links see it, on its own stream
                a = torch.cuda.Event(enable_timing=True)
                a.record(self._push_stream)
            check_call(LIB.dgla_peer_push(x_local.data_ptr(), self.row_bytes, self.serve_rows.data_ptr(),
                                          self._segs[parity].data_ptr(), self._nseg, self._nblk, self.epoch,
                                          self._arrive.data_ptr(), self._push_stream.cuda_stream))
            if prof is not None:
                b = torch.cuda.Event(enable_timing=True)
                b.record(self._push_stream)
                prof.append((a, b))
            self._push_done.record(self._push_stream)
            self._pending = True
        return self._halo[parity]

    def finish_step(self):
        """Order the current stream after this step's push: x_local may be overwritten from here on."""
        if self._pending:
            torch.cuda.current_stream(self.device).wait_event(self._push_done)
            self._pending = False

    def wait_chunk(self, c):
        """Order the current stream after chunk ``c`` of every peer for the step begun last."""
        if self.world > 1:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check_call(LIB.dgla_peer_wait(self._flags.data_ptr() + 8 * c * self.world, self.world, self.epoch,
                                          self._status.data_ptr(), self.max_spins, stream))

    def wait(self):
        for c in range(self.chunks):
            self.wait_chunk(c)

    def check(self):
        """Raise if a wait ever gave up on a peer (synchronises).  A wait that gives up lets the GPU move on with
        STALE halo rows (and breaks the two-buffer write-after-read argument for the steps after it), so results
        produced since the last check must not be used when this raises: ``ShardedSpMM.step`` calls it every
        ``check_every`` steps and ``ShardedSpMM.close`` at the end; callers driving the exchange themselves must too."""
        if int(self._status.item()):
            raise DGLAMDError("peer exchange: a peer never delivered its rows (flag wait timed out); the results of "
                              "the steps since the last check were computed on stale halo rows")

    def bytes_per_step(self, elem_size=None):
        return self.n_halo * self.row_bytes

    def close(self):
        torch.cuda.synchronize(self.device)
        for p in self._imported:
            LIB.dgla_ipc_release(ctypes.c_void_p(p))
        self._imported = []
        for p in self._owned:
            LIB.dgla_peer_free(ctypes.c_void_p(p))
        self._owned = []
