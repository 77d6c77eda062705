"""Loader of libdgl_amd.so (the hand-written gfx950 kernels + C ABI).

There is NO fallback: if the shared library is missing the package raises at import of
this module, and if no ROCm device is present every compute entry point raises.  The
reference behaves the same way when libdgl.so is absent (python/dgl/_ffi/base.py:36-50).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdgl_amd.so")


class DGLAMDError(RuntimeError):
    """Raised for errors reported through the C ABI (the reference's DGLError,
    python/dgl/_ffi/base.py:58-70)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "dgl_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C dgl_amd/csrc`. There is no CPU/PyTorch fallback for the "
            "g-SpMM / g-SDDMM kernels." % LIB_PATH
        )
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


LIB = _load()

c_void_p, c_char_p, c_int, c_int64, c_size_t, c_uint32 = (
    ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t,
    ctypes.c_uint32,
)


class CSR(ctypes.Structure):  # dgla_csr
    _fields_ = [("num_rows", c_int64), ("num_cols", c_int64), ("nnz", c_int64),
                ("idtype_bits", ctypes.c_int32), ("indptr", c_void_p), ("indices", c_void_p),
                ("data", c_void_p)]


class COO(ctypes.Structure):  # dgla_coo
    _fields_ = [("num_rows", c_int64), ("num_cols", c_int64), ("nnz", c_int64),
                ("idtype_bits", ctypes.c_int32), ("row", c_void_p), ("col", c_void_p),
                ("data", c_void_p)]


class Tensor(ctypes.Structure):  # dgla_tensor
    _fields_ = [("data", c_void_p), ("ndim", ctypes.c_int32),
                ("shape", ctypes.POINTER(c_int64))]


P = ctypes.POINTER
LIB.dgla_last_error.restype = c_char_p
LIB.dgla_abi_version.restype = c_int
LIB.dgla_spmm_csr.restype = c_int
LIB.dgla_spmm_csr.argtypes = [c_char_p, c_char_p, P(CSR), c_int, P(Tensor), P(Tensor), P(Tensor),
                              c_void_p, c_void_p, c_void_p, c_size_t, c_uint32, c_void_p]
LIB.dgla_spmm_csr_workspace_bytes.restype = c_size_t
LIB.dgla_spmm_csr_workspace_bytes.argtypes = [c_char_p, c_char_p, P(CSR), c_int, P(Tensor),
                                              P(Tensor), P(Tensor)]
LIB.dgla_spmm_csr_stacked.restype = c_int
LIB.dgla_spmm_csr_stacked.argtypes = [c_char_p, P(CSR), c_void_p, c_int, c_int, P(Tensor), P(Tensor),
                                      c_void_p, c_void_p, P(Tensor), c_void_p, c_size_t, c_uint32,
                                      c_void_p]
LIB.dgla_spmm_csr_stacked_workspace_bytes.restype = c_size_t
LIB.dgla_spmm_csr_stacked_workspace_bytes.argtypes = [c_char_p, P(CSR), c_int, P(Tensor), P(Tensor),
                                                      P(Tensor)]
LIB.dgla_spmm_csr_stacked_cmp.restype = c_int
LIB.dgla_spmm_csr_stacked_cmp.argtypes = [c_char_p, c_char_p, P(CSR), c_void_p, c_int, c_void_p, c_void_p,
                                          c_int, P(Tensor), P(Tensor), c_void_p, c_void_p, P(Tensor),
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                          c_uint32, c_void_p]
LIB.dgla_spmm_csr_stacked_cmp_workspace_bytes.restype = c_size_t
LIB.dgla_spmm_csr_stacked_cmp_workspace_bytes.argtypes = [c_char_p, c_char_p, P(CSR), c_int, P(Tensor),
                                                          P(Tensor), P(Tensor)]
LIB.dgla_spmm_coo.restype = c_int
LIB.dgla_spmm_coo.argtypes = [c_char_p, c_char_p, P(COO), c_int, P(Tensor), P(Tensor), P(Tensor),
                              c_void_p, c_void_p, c_void_p]
LIB.dgla_sddmm_coo.restype = c_int
LIB.dgla_sddmm_coo.argtypes = [c_char_p, P(COO), c_int, P(Tensor), P(Tensor), P(Tensor), c_int,
                               c_int, c_void_p]
LIB.dgla_sddmm_csr.restype = c_int
LIB.dgla_sddmm_csr.argtypes = [c_char_p, P(CSR), c_int, P(Tensor), P(Tensor), P(Tensor), c_int,
                               c_int, c_void_p]
LIB.dgla_edge_softmax_forward.restype = c_int
LIB.dgla_edge_softmax_forward.argtypes = [P(CSR), c_int, P(Tensor), P(Tensor), c_void_p, c_size_t,
                                          c_uint32, c_void_p]
LIB.dgla_edge_softmax_workspace_bytes.restype = c_size_t
LIB.dgla_edge_softmax_workspace_bytes.argtypes = [P(CSR), c_int, c_int64]
LIB.dgla_edge_softmax_backward.restype = c_int
LIB.dgla_edge_softmax_backward.argtypes = [P(CSR), c_int, P(Tensor), P(Tensor), P(Tensor),
                                           c_void_p, c_size_t, c_uint32, c_void_p]
LIB.dgla_gat_attention_workspace_bytes.restype = c_size_t
LIB.dgla_gat_attention_workspace_bytes.argtypes = [P(CSR), c_int64, c_int64]
LIB.dgla_gat_attention_forward.restype = c_int
LIB.dgla_gat_attention_forward.argtypes = [P(CSR), c_int, P(Tensor), P(Tensor), P(Tensor), ctypes.c_float, P(Tensor),
                                           c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_gat_attention_backward.restype = c_int
LIB.dgla_gat_attention_backward.argtypes = [P(CSR), P(CSR), c_int, P(Tensor), P(Tensor), P(Tensor), P(Tensor), c_void_p,
                                            P(Tensor), ctypes.c_float, P(Tensor), P(Tensor), P(Tensor), c_void_p,
                                            c_size_t, c_void_p]
LIB.dgla_spmm_set_profile_events.restype = c_int
LIB.dgla_spmm_set_profile_events.argtypes = [c_void_p, c_void_p]
LIB.dgla_segment_reduce_workspace_bytes.restype = c_size_t
LIB.dgla_segment_reduce_workspace_bytes.argtypes = [c_char_p, c_int, c_int, P(Tensor), c_int64,
                                                    P(Tensor)]
LIB.dgla_segment_reduce.restype = c_int
LIB.dgla_segment_reduce.argtypes = [c_char_p, c_int, c_int, P(Tensor), c_void_p, c_int64, P(Tensor),
                                    c_void_p, c_void_p, c_size_t, c_uint32, c_void_p]
LIB.dgla_scatter_add.restype = c_int
LIB.dgla_scatter_add.argtypes = [c_int, c_int, P(Tensor), c_void_p, P(Tensor), c_void_p]
LIB.dgla_update_grad_minmax.restype = c_int
LIB.dgla_update_grad_minmax.argtypes = [c_int, c_int, P(Tensor), c_void_p, c_void_p, c_int64, P(Tensor), c_void_p]
for _n, _a in (("dgla_peer_alloc", [ctypes.c_size_t, c_int, P(c_void_p)]), ("dgla_peer_free", [c_void_p]),
               ("dgla_ipc_export", [c_void_p, c_void_p]), ("dgla_ipc_import", [c_void_p, P(c_void_p)]),
               ("dgla_ipc_release", [c_void_p]),
               ("dgla_peer_push", [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, ctypes.c_uint64, c_void_p,
                                   c_void_p]),
               ("dgla_peer_wait", [c_void_p, c_int, ctypes.c_uint64, c_void_p, c_int64, c_void_p])):
    getattr(LIB, _n).restype = c_int
    getattr(LIB, _n).argtypes = _a
LIB.dgla_spmm_cmp_backward.restype = c_int
LIB.dgla_spmm_cmp_backward.argtypes = [c_int, c_int, P(Tensor), c_void_p, P(Tensor), c_void_p, c_int64, P(Tensor),
                                       c_int, c_void_p]
LIB.dgla_spmm_cmp_mask_words.restype = c_int64
LIB.dgla_spmm_cmp_mask_words.argtypes = [c_int, c_int64]
LIB.dgla_spmm_cmp_mask_bytes.restype = c_size_t
LIB.dgla_spmm_cmp_mask_bytes.argtypes = [c_int, c_int64, c_int64, c_int64]
LIB.dgla_spmm_cmp_mask.restype = c_int
LIB.dgla_spmm_cmp_mask.argtypes = [P(CSR), c_int, c_void_p, c_int, P(Tensor), c_void_p, P(Tensor), c_void_p]
LIB.dgla_spmm_csr_masked_workspace_bytes.restype = c_size_t
LIB.dgla_spmm_csr_masked_workspace_bytes.argtypes = [P(CSR), c_int, P(Tensor), P(Tensor)]
LIB.dgla_spmm_csr_masked.restype = c_int
LIB.dgla_spmm_csr_masked.argtypes = [P(CSR), c_int, P(Tensor), c_void_p, P(Tensor), c_void_p, c_size_t, c_uint32,
                                     c_void_p]
LIB.dgla_backward_segment_cmp.restype = c_int
LIB.dgla_backward_segment_cmp.argtypes = [c_int, c_int, P(Tensor), c_void_p, P(Tensor), c_void_p]
LIB.dgla_segment_mm_workspace_bytes.restype = c_size_t
LIB.dgla_segment_mm_workspace_bytes.argtypes = [c_int, c_int64, c_int64, c_int64]
LIB.dgla_segment_mm.restype = c_int
LIB.dgla_segment_mm.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p]
LIB.dgla_segment_mm_backward_b.restype = c_int
LIB.dgla_segment_mm_backward_b.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int64, c_int64, c_int64, c_int64, c_void_p, c_size_t,
                                           c_void_p]
LIB.dgla_segment_mm_indexed.restype = c_int
LIB.dgla_segment_mm_indexed.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                        c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p]
LIB.dgla_segment_mm_backward_b_last_route.restype = c_int
LIB.dgla_segment_mm_backward_b_last_route.argtypes = [P(c_uint32), P(c_uint32)]
LIB.dgla_segment_mm_backward_b_indexed.restype = c_int
LIB.dgla_segment_mm_backward_b_indexed.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                   c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                                   c_size_t, c_void_p]
LIB.dgla_gather_mm.restype = c_int
LIB.dgla_gather_mm.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int64, c_int64, c_int64, c_void_p]
LIB.dgla_coo_to_csr_workspace_bytes.restype = c_size_t
LIB.dgla_coo_to_csr_workspace_bytes.argtypes = [c_int, c_int64, c_int64]
LIB.dgla_coo_to_csr.restype = c_int
LIB.dgla_coo_to_csr.argtypes = [c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_coo_to_csr_bounded.restype = c_int
LIB.dgla_coo_to_csr_bounded.argtypes = [c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_sample_neighbors_workspace_bytes.restype = c_size_t
LIB.dgla_sample_neighbors_workspace_bytes.argtypes = [c_int, c_int64]
LIB.dgla_sample_neighbors.restype = c_int
LIB.dgla_sample_neighbors.argtypes = [P(CSR), c_void_p, c_int64, c_int, c_int, ctypes.c_uint64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_sample_neighbors_weighted.restype = c_int
LIB.dgla_sample_neighbors_weighted.argtypes = [P(CSR), c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                                               ctypes.c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_size_t, c_void_p]
LIB.dgla_to_block_workspace_bytes.restype = c_size_t
LIB.dgla_to_block_workspace_bytes.argtypes = [c_int, c_int64]
LIB.dgla_to_block.restype = c_int
LIB.dgla_to_block.argtypes = [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_sample_neighbors_padded.restype = c_int
LIB.dgla_sample_neighbors_padded.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, c_int,
                                             ctypes.c_uint64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_size_t, c_void_p]
LIB.dgla_to_block_padded.restype = c_int
LIB.dgla_to_block_padded.argtypes = [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_partition_kway_ex.restype = c_int
LIB.dgla_partition_kway_ex.argtypes = [c_int, c_int64, c_void_p, c_void_p, c_int, ctypes.c_double, c_int, ctypes.c_uint64,
                                       c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
LIB.dgla_partition_kway.restype = c_int
LIB.dgla_partition_kway.argtypes = [c_int, c_int64, c_void_p, c_void_p, c_int, ctypes.c_double, c_int,
                                    ctypes.c_uint64, c_void_p, c_void_p]
LIB.dgla_gather_rows.restype = c_int
LIB.dgla_gather_rows.argtypes = [c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
LIB.dgla_scatter_rows.restype = c_int
LIB.dgla_scatter_rows.argtypes = [c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
LIB.dgla_partition_map.restype = c_int
LIB.dgla_partition_map.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                   c_void_p]
LIB.dgla_partition_to_global.restype = c_int
LIB.dgla_partition_to_global.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                         c_void_p]
LIB.dgla_set_tuning.restype = c_int
LIB.dgla_set_tuning.argtypes = [c_uint32]
LIB.dgla_get_tuning.restype = c_uint32
LIB.dgla_narrow_reduce_calls.restype = c_int64
LIB.dgla_narrow_reduce_calls.argtypes = []
LIB.dgla_stream_copy.restype = c_int
LIB.dgla_stream_copy.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
LIB.dgla_stream_copy_variant.restype = c_int
LIB.dgla_stream_copy_variant.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p]

DGLA_ACCUMULATE = 1
DGLA_PLAN_VALID = 2
DGLA_MEAN = 4
DGLA_SPLIT_VALID = 8
DGLA_SPLIT_KEEP = 16
DGLA_PREPARE_ONLY = 32
DGLA_ESM_OUT_POSITION = 128
DGLA_ESM_B_IS_GRAD = 256
DGLA_TUNE_XCD, DGLA_TUNE_SPLIT, DGLA_TUNE_GLDS = 1, 8, 16
DGLA_TUNE_NO_GATE, DGLA_TUNE_NO_STAGE_W = 4096, 8192
DGLA_TUNE_SPLIT_FORCE, DGLA_TUNE_MM_F32, DGLA_TUNE_MM_X3 = 64, 128, 2048   # (2, 4, 32, 256, 512, 1024 were retired in round 4)
DEFAULT_TUNING = DGLA_TUNE_XCD | DGLA_TUNE_SPLIT | DGLA_TUNE_GLDS  # csrc/common.h kDefaultTuning


def check_call(ret):
    if ret != 0:
        raise DGLAMDError(LIB.dgla_last_error().decode("utf-8", "replace"))
