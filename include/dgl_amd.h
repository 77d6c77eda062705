/*
 * dgl_amd.h — C ABI of the MI355X-native g-SpMM / g-SDDMM hot path (libdgl_amd.so).
 *
 * Two layers are exported:
 *
 *  (1) the GRAPH-FREE SEAM `dgla_*` — plain pointers and sizes.  It replaces the
 *      reference's aten::CSRSpMM / COOSpMM / CSRSDDMM / COOSDDMM
 *      (src/array/array.cc:1148-1233, declared include/dgl/aten/csr.h:1046-1066 and
 *      include/dgl/aten/coo.h:842-867), i.e. the template seam
 *      SpMMCsr<XPU,IdType,DType> / SDDMMCsr / SDDMMCoo of src/array/kernel_decl.h:23-87
 *      with XPU = the ROCm device.  This is what a libdgl build would link against.
 *
 *  (2) the REGISTRY LAYER `DGL*` — the PackedFunc-style FFI that python/dgl/_ffi binds
 *      (include/dgl/runtime/c_runtime_api.h:205-212,336-338,437-445): functions are found
 *      by name ("sparse._CAPI_DGLKernelSpMM", ...) and called with (DGLValue*, type codes).
 *      See the second half of this header.
 *
 * All device pointers are HIP device pointers on the current device; every call is
 * asynchronous on the given stream (reference: kernels run on the current PyTorch stream,
 * src/runtime/cuda/cuda_device_api.cc:362-367).  Functions return 0 on success and -1 on
 * error with the message available from dgla_last_error() / DGLGetLastError()
 * (reference convention: src/runtime/runtime_base.h:14-45).
 */
#ifndef DGL_AMD_H_
#define DGL_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGLA_ABI_VERSION 2

/* Feature element types (DGLDataType {code,bits}: float32/64, float16, bfloat16 —
 * ATEN_FLOAT_TYPE_SWITCH_16BITS, include/dgl/aten/macro.h:137-166). */
typedef enum { DGLA_F32 = 0, DGLA_F64 = 1, DGLA_F16 = 2, DGLA_BF16 = 3 } dgla_dtype;

/* Flags of dgla_spmm_csr. */
#define DGLA_ACCUMULATE 1u /* out += result: the reference's contract, `out` pre-zeroed by the
                              caller (src/array/cuda/spmm.cuh:528-534, _sparse_ops.py:227).
                              Without it rows are written, not accumulated (no memset, no
                              read of `out`).  Only meaningful for reduce == "sum". */
#define DGLA_PLAN_VALID 2u /* `workspace` still holds the merge plan built by an earlier
                              call on the SAME csr (indptr contents, num_rows, nnz). */
#define DGLA_MEAN 4u       /* reduce == "sum" only: every output row is divided by
                              max(in-degree, 1) before it is stored — the `mean` reducer of
                              dgl.ops.gspmm (python/dgl/ops/spmm.py:109-114: sum, then
                              / clamp(in_degrees, 1)) without the second pass over `out`;
                              same two roundings as the reference's sum-then-divide. */
#define DGLA_SPLIT_KEEP 16u /* the caller promises to hand the SAME, unchanged ufeat to later calls
                              on this csr + workspace (static input features, e.g. layer 0 of
                              full-graph training or inference): the split-row copy
                              (DGLA_TUNE_SPLIT) is made whatever the locality probe says — its
                              cost is paid once — and later calls pass DGLA_SPLIT_VALID. */
#define DGLA_SPLIT_VALID 8u /* `workspace` still holds the split-row copy an earlier call (with
                              DGLA_SPLIT_KEEP) made of THIS ufeat, whose contents have not
                              changed since: the re-layout copy is skipped.  Needs
                              DGLA_PLAN_VALID.  Never set it for a tensor you do not own. */
#define DGLA_ESM_OUT_POSITION 128u /* dgla_edge_softmax_forward: see there */
#define DGLA_ESM_B_IS_GRAD 256u    /* dgla_edge_softmax_backward: see there */
#define DGLA_PREPARE_ONLY 32u /* (with DGLA_SPLIT_KEEP) build the merge plan if needed and make the side copy of
                              ufeat's ragged row ends — nothing else: no output is written.  For the PRODUCER of
                              ufeat (the previous layer's epilogue, a feature loader): it prepares the operand on
                              its own stream when it finishes the tensor, and the g-SpMM that consumes it passes
                              DGLA_SPLIT_VALID and launches the merge kernel alone (0.10 ms of the headline
                              step moved out of the consumer's critical path). */

/* CSRMatrix (include/dgl/aten/csr.h:40-49).  For SpMM the rows are DESTINATION nodes
 * (the in-edge CSR / "CSC", src/array/kernel.cc:20-44); for SDDMM rows are SOURCE nodes. */
typedef struct {
  int64_t num_rows, num_cols, nnz;
  int32_t idtype_bits;  /* 32 or 64: element type of indptr / indices / data */
  const void* indptr;   /* [num_rows + 1] */
  const void* indices;  /* [nnz] column ids */
  const void* data;     /* [nnz] edge-id map, or NULL: edge id == position */
} dgla_csr;

/* COOMatrix (include/dgl/aten/coo.h): row = source ids, col = destination ids. */
typedef struct {
  int64_t num_rows, num_cols, nnz;
  int32_t idtype_bits;
  const void* row;
  const void* col;
  const void* data;
} dgla_coo;

/* Dense, contiguous, row-major feature tensor.  shape[0] is the number of nodes / edges,
 * ndim >= 2 (src/array/check.h:46-50).  data == NULL means "operand absent" (the
 * reference passes an empty int64 NDArray, python/dgl/ndarray.py:309-312). */
typedef struct {
  void* data;
  int32_t ndim;
  const int64_t* shape;
} dgla_tensor;

const char* dgla_last_error(void);
int dgla_abi_version(void);

/*
 * g-SpMM on CSR:  out[r, k] = reduce_{j in row r} op(ufeat[indices[j], lhs_off(k)],
 *                                                  efeat[eid(j), rhs_off(k)])
 * Replaces aten::CSRSpMM (src/array/array.cc:1148-1168) -> SpMMCsr<kDGLCUDA,...>
 * (src/array/cuda/spmm.cu:26-106).
 *   op      "add" | "sub" | "mul" | "div" | "copy_lhs" | "copy_rhs"
 *   reduce  "sum" | "max" | "min"
 *   arg_u / arg_e   [same shape as out], element type = idtype; written for max/min only
 *                   (arg_u iff op uses lhs, arg_e iff op uses rhs); may be NULL for sum.
 *   workspace       device scratch of at least dgla_spmm_csr_workspace_bytes() bytes.
 */
int dgla_spmm_csr(const char* op, const char* reduce, const dgla_csr* csr, dgla_dtype dtype,
                  const dgla_tensor* ufeat, const dgla_tensor* efeat, const dgla_tensor* out,
                  void* arg_u, void* arg_e, void* workspace, size_t workspace_bytes,
                  uint32_t flags, void* hip_stream);

size_t dgla_spmm_csr_workspace_bytes(const char* op, const char* reduce, const dgla_csr* csr,
                                     dgla_dtype dtype, const dgla_tensor* ufeat,
                                     const dgla_tensor* efeat, const dgla_tensor* out);

/*
 * Multi-relation g-SpMM with reduce = sum in ONE launch.  Replaces the per-relation loop of
 * SpMMCsrHetero<kDGLCUDA,...> (src/array/cuda/spmm_hetero.cu:26-200), which launches one
 * accumulating kernel per edge type and so re-reads and re-writes the destination buffer
 * once per relation.
 *   csr         row-wise concatenation ("stack") of the in-edge CSRs of all relations that
 *               share the destination node type: row r lists relation 0's edges into r, then
 *               relation 1's, ...; `data` maps a stacked position to the relation-local edge id.
 *   rel         uint8 [nnz]: relation index of every stacked edge, in [0, num_rel)
 *   ufeat0 / efeat0   relation 0's operands (feature shapes are taken from them; all
 *               relations must agree, src/array/kernel.cc:194-199)
 *   ufeat_ptrs / efeat_ptrs   DEVICE arrays of num_rel device pointers: each relation's
 *               source-node / edge feature tensor (NULL when the operator does not use it)
 *   op          "copy_lhs" | "copy_rhs" | "mul"
 * Workspace as for dgla_spmm_csr (same byte count for the stacked csr).
 */
size_t dgla_spmm_csr_stacked_workspace_bytes(const char* op, const dgla_csr* csr,
                                             dgla_dtype dtype, const dgla_tensor* ufeat0,
                                             const dgla_tensor* efeat0, const dgla_tensor* out);
int dgla_spmm_csr_stacked(const char* op, const dgla_csr* csr, const void* rel, int num_rel,
                          dgla_dtype dtype, const dgla_tensor* ufeat0, const dgla_tensor* efeat0,
                          const void* const* ufeat_ptrs, const void* const* efeat_ptrs,
                          const dgla_tensor* out, void* workspace, size_t workspace_bytes,
                          uint32_t flags, void* hip_stream);

/*
 * Multi-relation g-SpMM with reduce = max / min in ONE launch, with the node / edge type trackers
 * of SpMMCsrHetero's compare path (src/array/cuda/spmm_hetero.cu:87-117,160-188 ->
 * SpMMCmpCsrHeteroKernel, spmm.cuh:552-606: a running compare relation by relation, an earlier
 * relation keeps a tie).  `csr`, `rel`, the operand tables and `op` as for dgla_spmm_csr_stacked.
 *   src_ntype / etype   HOST arrays [num_rel]: source node type / edge type of every stacked relation
 *   arg_u, arg_u_ntype  [num_rows, out_len] ids: winning source node and its node type
 *                       (required when the operator reads ufeat, else NULL)
 *   arg_e, arg_e_etype  [num_rows, out_len] ids: winning relation-local edge id and its edge type
 *                       (required when the operator reads efeat, else NULL)
 * An output element no edge reaches holds the reducer's identity, args 0 and trackers -1.
 */
size_t dgla_spmm_csr_stacked_cmp_workspace_bytes(const char* op, const char* reduce, const dgla_csr* csr,
                                                 dgla_dtype dtype, const dgla_tensor* ufeat0,
                                                 const dgla_tensor* efeat0, const dgla_tensor* out);
int dgla_spmm_csr_stacked_cmp(const char* op, const char* reduce, const dgla_csr* csr, const void* rel,
                              int num_rel, const int32_t* src_ntype, const int32_t* etype,
                              dgla_dtype dtype, const dgla_tensor* ufeat0, const dgla_tensor* efeat0,
                              const void* const* ufeat_ptrs, const void* const* efeat_ptrs,
                              const dgla_tensor* out, void* arg_u, void* arg_e, void* arg_u_ntype,
                              void* arg_e_etype, void* workspace, size_t workspace_bytes,
                              uint32_t flags, void* hip_stream);

/*
 * g-SpMM on COO (edge-parallel with device atomics).  Replaces aten::COOSpMM
 * (array.cc:1170-1190) -> SpMMCoo<kDGLCUDA,...> (spmm.cu:80-106, spmm.cuh:624-682).
 * `out` (and arg_*) are fully written by the call.  fp16 / bf16 are refused like the
 * reference does (spmm.cuh:633-641).
 */
int dgla_spmm_coo(const char* op, const char* reduce, const dgla_coo* coo, dgla_dtype dtype,
                  const dgla_tensor* ufeat, const dgla_tensor* efeat, const dgla_tensor* out,
                  void* arg_u, void* arg_e, void* hip_stream);

/*
 * g-SDDMM:  out[eid, k] = op(lhs[sel(lhs_target), lhs_off(k)], rhs[sel(rhs_target), rhs_off(k)])
 * Replaces aten::COOSDDMM / CSRSDDMM (array.cc:1192-1233) -> SDDMMCoo / SDDMMCsr
 * (src/array/cuda/sddmm.cu:17-42, sddmm.hip.h:97-362).
 *   op       "add" | "sub" | "mul" | "div" | "copy_lhs" | "copy_rhs" | "dot"
 *   targets  0 = u (source node), 1 = e (edge), 2 = v (destination node)
 */
int dgla_sddmm_coo(const char* op, const dgla_coo* coo, dgla_dtype dtype,
                   const dgla_tensor* lhs, const dgla_tensor* rhs, const dgla_tensor* out,
                   int lhs_target, int rhs_target, void* hip_stream);

int dgla_sddmm_csr(const char* op, const dgla_csr* csr, dgla_dtype dtype,
                   const dgla_tensor* lhs, const dgla_tensor* rhs, const dgla_tensor* out,
                   int lhs_target, int rhs_target, void* hip_stream);

/*
 * Fused edge softmax over the in-edge CSR (rows = destination nodes), forward and
 * backward.  The reference only has these on CPU (src/array/cpu/spmm.h:484-570, FFI
 * src/array/kernel.cc:542-561; GPU is a TODO at kernel.cc:313,331 and runs 5 kernels,
 * python/dgl/backend/pytorch/sparse.py:709-713).
 *   score / out / grad tensors: [nnz, ...] indexed by edge id.
 *   workspace  optional device scratch of dgla_edge_softmax_workspace_bytes() bytes: with it
 *              (and a feature length <= 16) the degree-balanced merge-path kernels run and
 *              hub rows cost no more than any other 256 edges; without it (NULL / 0) a
 *              row-per-lane-group kernel needing no scratch is used.  DGLA_PLAN_VALID in
 *              `flags`: the workspace still holds the plan of an earlier call on this csr.
 *   DGLA_ESM_OUT_POSITION (merge-path kernels only).  Forward: `score` is read by edge id through the CSR's map as
 *              ever, `out` is written in the CSR's POSITION order — a softmax kept in position order costs one
 *              scattered 32-byte READ per edge instead of a scattered read and a scattered WRITE.  Backward: `out`
 *              (that position-ordered softmax) is read and `back` written by position, `sds` (the caller's gradient)
 *              is read by edge id through the map.  The Python side keeps the position-ordered tensors to itself and
 *              hands out edge-id order through one gather by the inverse map (dgl_amd/autograd.py EdgeSoftmax).
 *   DGLA_ESM_B_IS_GRAD (backward, merge-path kernels only): `sds` holds the upstream gradient g itself; the product
 *              out * g of python/dgl/backend/pytorch/sparse.py:709-713 is formed inside the kernel (rounded to the
 *              storage type as the separate elementwise kernel leaves it: same bits, 6 bytes per element less traffic).
 */
size_t dgla_edge_softmax_workspace_bytes(const dgla_csr* csr, dgla_dtype dtype, int64_t dim);
int dgla_edge_softmax_forward(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* score,
                              const dgla_tensor* out, void* workspace, size_t workspace_bytes,
                              uint32_t flags, void* hip_stream);
int dgla_edge_softmax_backward(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* out,
                               const dgla_tensor* sds, const dgla_tensor* back, void* workspace,
                               size_t workspace_bytes, uint32_t flags, void* hip_stream);

/*
 * GAT attention block as ONE operator (csrc/gat_attention.hip):
 *     out[v, h, :] = sum_{u -> v} softmax_v( leaky_relu(el[u, h] + er[v, h]) ) * ft[u, h, :]
 * i.e. the sequence u_add_v -> leaky_relu -> edge_softmax -> u_mul_e_sum of the reference's GATConv
 * (python/dgl/nn/pytorch/conv/gatconv.py:330-347; the softmax alone is five launches on the reference's GPU path,
 * python/dgl/backend/pytorch/sparse.py:709-713, with a fused version left as a TODO at src/array/kernel.cc:313,331)
 * in one pass over the in-edges: no (E, H) tensor is written or read, so an edge-id map costs nothing.
 *   csc        in-edge CSR (rows = destination nodes); its `data` (edge-id map) is not read
 *   ft         (N_src, H, D) fp32;  el (N_src, H, 1);  er (N_dst, H, 1);  out (N_dst, H, D)
 *              D a power of two >= 4 and H * D <= 256, otherwise -1 (callers compose the four operators instead)
 *   mz         float [N_dst, H, 2]: the row's softmax maximum and normaliser, written by the forward and read by
 *              the backward, which recomputes the attention weights from them
 *   workspace  dgla_gat_attention_workspace_bytes(csc, H, D) bytes of device scratch (required when nnz > 0)
 * Backward: `csr` is the out-edge CSR (rows = source nodes) of the same graph; d_ft / d_el / d_er get the gradients
 * of ft / el / er for the upstream gradient `dout` (every row is written, rows without edges with zeros).
 * Rows without in-edges: out = 0.  Deterministic: no atomics, partial rows are merged in a fixed order.
 */
size_t dgla_gat_attention_workspace_bytes(const dgla_csr* csc, int64_t heads, int64_t dim);
int dgla_gat_attention_forward(const dgla_csr* csc, dgla_dtype dtype, const dgla_tensor* ft, const dgla_tensor* el,
                               const dgla_tensor* er, float negative_slope, const dgla_tensor* out, void* mz,
                               void* workspace, size_t workspace_bytes, void* hip_stream);
int dgla_gat_attention_backward(const dgla_csr* csc, const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ft,
                                const dgla_tensor* el, const dgla_tensor* er, const dgla_tensor* out, const void* mz,
                                const dgla_tensor* dout, float negative_slope, const dgla_tensor* d_ft,
                                const dgla_tensor* d_el, const dgla_tensor* d_er, void* workspace,
                                size_t workspace_bytes, void* hip_stream);

/* ---- segment reduce / scatter add ----------------------------------------
 * Replace SegmentReduce / ScatterAdd / BackwardSegmentCmp<kDGLCUDA,…>
 * (src/array/kernel_decl.h, kernels src/array/cuda/segment_reduce.cuh:30-113; registered as
 * sparse._CAPI_DGLKernelSegmentReduce / ScatterAdd / BwdSegmentCmp, src/array/kernel.cc:658-708).
 *
 * dgla_segment_reduce: out[i, :] = reduce_{j in [offsets[i], offsets[i+1])} feat[j, :],
 *   reduce in {"sum","max","min"}; offsets has num_segments + 1 entries of idtype_bits;
 *   `arg` (idtype, shape of out; required for max/min) receives the winning row j, or -1 for
 *   an element nothing won (empty segment).  Empty segments give 0 / -inf / +inf like the
 *   reference.  `workspace` may be NULL: scratch is then taken from the stream-ordered HIP
 *   allocator for the duration of the call.
 * dgla_scatter_add: out[idx[i], :] += feat[i, :]   (atomic; out is NOT zeroed)
 * dgla_backward_segment_cmp: out[arg[i, k], k] = feat[i, k] wherever arg[i, k] >= 0
 * dgla_update_grad_minmax: out[idx[i, k], k] += feat[i, k] wherever idx_type[i, k] == type —
 *   one relation's share of the max / min gradient on a heterograph
 *   (UpdateGradMinMax_hetero<kDGLCUDA,…>, src/array/cuda/segment_reduce.cuh:73-92,184-225):
 *   idx = the winning node / edge ids the forward pass recorded, idx_type = the node / edge type
 *   of every winner (-1: nothing won), both [n, dim] of idtype; out is NOT zeroed (atomic).
 * dgla_spmm_cmp_backward: backward of g-SpMM max / min on one relation, ONE pass
 *   (python/dgl/backend/pytorch/sparse.py:217-244 is two .long() casts, a gather and an atomic
 *   scatter_add_): out[arg[i, k], k] (+)= dz[i, k] * (other ? other[arg_other[i, k], (k / other_group) %
 *   row_len(other)] : 1) wherever arg[i, k] >= 0; arg / arg_other = the winners the forward pass recorded
 *   ([n, dim] of idtype_bits).  atomic = 0 for the EDGE operand (an edge has one destination: every
 *   target element is written at most once, plain stores, deterministic), 1 for the NODE operand (a node
 *   can win at many destinations: hardware float atomics, like the reference).  out is NOT zeroed. */
size_t dgla_segment_reduce_workspace_bytes(const char* reduce, int idtype_bits, dgla_dtype dtype,
                                           const dgla_tensor* feat, int64_t num_segments,
                                           const dgla_tensor* out);
int dgla_segment_reduce(const char* reduce, int idtype_bits, dgla_dtype dtype,
                        const dgla_tensor* feat, const void* offsets, int64_t num_segments,
                        const dgla_tensor* out, void* arg, void* workspace,
                        size_t workspace_bytes, uint32_t flags, void* hip_stream);
int dgla_scatter_add(int idtype_bits, dgla_dtype dtype, const dgla_tensor* feat, const void* idx,
                     const dgla_tensor* out, void* hip_stream);
int dgla_update_grad_minmax(int idtype_bits, dgla_dtype dtype, const dgla_tensor* feat,
                            const void* idx, const void* idx_type, int64_t type,
                            const dgla_tensor* out, void* hip_stream);
int dgla_backward_segment_cmp(int idtype_bits, dgla_dtype dtype, const dgla_tensor* feat,
                              const void* arg, const dgla_tensor* out, void* hip_stream);
int dgla_spmm_cmp_backward(int idtype_bits, dgla_dtype dtype, const dgla_tensor* dz, const void* arg,
                           const dgla_tensor* other, const void* arg_other, int64_t other_group,
                           const dgla_tensor* out, int atomic, void* hip_stream);
/* The NODE operand's max / min gradient for copy_lhs / add WITHOUT atomics — the same scatter_add_ of
 * python/dgl/backend/pytorch/sparse.py:216-224 turned into a gather over the reverse graph, so that the result has the
 * same bits on every run:  dX[u, k] = sum over the out-edges e = (u -> v) of [e delivered the winner of (v, k)] dZ[v, k].
 *   dgla_spmm_cmp_mask    `csr` = the FORWARD matrix (rows = destinations); `arg` [num_rows, F] of idtype = arg_u
 *       (by_edge = 0: the first edge of row v whose source is arg_u[v, k] gets bit k) or arg_e (by_edge != 0: the
 *       edge whose id it names).  Writes, for every edge in the CSR's POSITION order, dgla_spmm_cmp_mask_words(dtype, F)
 *       words of the feature type's width (16 / 32 / 64 bits), bit (k mod width) of word k / width = column k.  An
 *       element no edge of its row claims (empty row, or nothing beat the identity: arg = 0) goes to dX[0, k] here,
 *       exactly where the reference's scatter sends it, summed in a fixed order: `dx` must be ZEROED before this call.
 *       Two-step form (round 6, saves the zero fill and the g-SpMM's read of dx): `by_edge | DGLA_CMP_MASK_DEFER` writes
 *       the bits and keeps the sums aside WITHOUT touching dx; dgla_spmm_csr_masked then STORES its rows (no
 *       DGLA_ACCUMULATE, dx may be uninitialised memory); the same call again with `by_edge | DGLA_CMP_MASK_FINISH` adds
 *       what was kept aside (and, only if an unclaimed element named a non-zero target, scans again for those).
 *   dgla_spmm_csr_masked  `csr` = the REVERSE matrix (rows = sources) whose `data` maps each of ITS positions to the
 *       forward position of the same edge (always present); `ufeat` = dZ, `mask` from above, `out` = dX with
 *       DGLA_ACCUMULATE (the atomics above are already in it).  One merge-path launch of the g-SpMM kernel.
 */
#define DGLA_CMP_MASK_DEFER 0x100
#define DGLA_CMP_MASK_FINISH 0x200
int64_t dgla_spmm_cmp_mask_words(dgla_dtype dtype, int64_t feat_len);
/* bytes of `mask`: the words of every edge + the mask pass's own partial sums behind them (no allocation inside the
 * call: it can be captured in a hipGraph) */
size_t dgla_spmm_cmp_mask_bytes(dgla_dtype dtype, int64_t num_rows, int64_t nnz, int64_t feat_len);
int dgla_spmm_cmp_mask(const dgla_csr* csr, dgla_dtype dtype, const void* arg, int by_edge, const dgla_tensor* dz,
                       void* mask, const dgla_tensor* dx, void* hip_stream);
size_t dgla_spmm_csr_masked_workspace_bytes(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ufeat,
                                            const dgla_tensor* out);
int dgla_spmm_csr_masked(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ufeat, const void* mask,
                         const dgla_tensor* out, void* workspace, size_t workspace_bytes, uint32_t flags,
                         void* hip_stream);

/* ---- segment / gather matrix multiply --------------------------------------
 * Replace SegmentMM / SegmentMMBackwardB / GatherMM / GatherMMScatter<kDGLCUDA,…>
 * (src/array/cuda/gather_mm.cu:201-360; registered as sparse._CAPI_DGLKernelSEGMENTMM,
 * …SEGMENTMMBackwardB, …GATHERMM, …GATHERMMSCATTER, src/array/kernel.cc:501-540).
 * All matrices row-major and contiguous; `seglen` has num_rel entries of idtype_bits and may
 * live in HOST memory (seglen_on_host = 1, as the reference passes it) or on the device.
 *
 * dgla_segment_mm:  for every relation r with rows [o_r, o_r + seglen[r]) of A:
 *     b_trans == 0:  C[rows] = A[rows] (m x k) . B[r] (k x n),  B is [num_rel, k, n]
 *     b_trans != 0:  C[rows] = A[rows] (m x k) . B[r]^T,        B is [num_rel, n, k]
 *   ONE grouped launch on the MFMA units (fp32 accumulate; fp64 on the vector ALUs).  Rows
 *   beyond sum(seglen) come back ZERO, as the reference's do (its output starts as th.zeros,
 *   python/dgl/backend/pytorch/sparse.py:975).
 * dgla_segment_mm_backward_b:  dB[r] (d1 x d2) = A[rows_r]^T (d1 x m) . dC[rows_r] (m x d2);
 *   relations without rows get zeros.  fp32 default route (round 6): operands as two fp16 terms under per-column
 *   power-of-two scales estimated from a row sample and verified in the launch (>= 2^-21 per element; elements more than
 *   26 binades under their column's sample maximum are added exactly from a short list); inputs that leave the estimated
 *   range, Inf / NaN included, are redone by the three-bf16-term kernel inside the same call.  DGLA_TUNE_MM_X3 /
 *   DGLA_TUNE_MM_F32 select the three-term / v_mfma_f32_32x32x2_f32 routes.
 * dgla_segment_mm_backward_b_last_route: statistics of the most recent fp32 two-term weight-gradient launch on the
 *   current device (synchronising read): whether it was redone by the three-term kernel, and how many elements went
 *   through the exact list.  No reference counterpart (diagnostics for tests / benchmarks).
 * dgla_gather_mm:  C[idx_c ? idx_c[i] : i] = A[idx_a ? idx_a[i] : i] (1 x k) . B[idx_b ? idx_b[i] : i]
 *   (k x n), one wavefront per row (the small-shape path of dgl.ops.gather_mm).
 * `workspace` may be NULL (stream-ordered scratch for the duration of the call). */
size_t dgla_segment_mm_workspace_bytes(dgla_dtype dtype, int64_t num_rel, int64_t d1, int64_t d2);
int dgla_segment_mm(int idtype_bits, dgla_dtype dtype, const void* a, const void* b, void* c,
                    const void* seglen, int seglen_on_host, int64_t num_rows, int64_t num_rel,
                    int64_t k, int64_t n, int b_trans, void* workspace, size_t workspace_bytes,
                    void* hip_stream);
int dgla_segment_mm_backward_b(int idtype_bits, dgla_dtype dtype, const void* a, const void* dc,
                               void* db, const void* seglen, int seglen_on_host,
                               int64_t num_rows, int64_t num_rel, int64_t d1, int64_t d2,
                               void* workspace, size_t workspace_bytes, void* hip_stream);
int dgla_segment_mm_backward_b_last_route(uint32_t* fell_back, uint32_t* listed_elements);
/* The same two operators over rows that are NOT stored grouped by relation: logical row r (the
 * order `seglen` describes) lives at physical row row_index[r] of A and of C (forward) / of A and
 * dC (weight gradient).  This is dgl.ops.gather_mm for large inputs without its two
 * index_select copies (python/dgl/ops/gather_mm.py:44-60): sort idx_b once, pass the permutation. */
int dgla_segment_mm_indexed(int idtype_bits, dgla_dtype dtype, const void* a, const void* b, void* c,
                            const void* seglen, int seglen_on_host, const int64_t* row_index,
                            int64_t num_rows, int64_t num_rel, int64_t k, int64_t n, int b_trans,
                            void* workspace, size_t workspace_bytes, void* hip_stream);
int dgla_segment_mm_backward_b_indexed(int idtype_bits, dgla_dtype dtype, const void* a, const void* dc,
                                       void* db, const void* seglen, int seglen_on_host,
                                       const int64_t* row_index, int64_t num_rows, int64_t num_rel,
                                       int64_t d1, int64_t d2, void* workspace, size_t workspace_bytes,
                                       void* hip_stream);
int dgla_gather_mm(int idtype_bits, dgla_dtype dtype, const void* a, const void* b, void* c,
                   const void* idx_a, const void* idx_b, const void* idx_c, int64_t num_rows,
                   int64_t k, int64_t n, void* hip_stream);

/* ---- COO -> CSR / CSC materialisation -----------------------------------
 * Replaces aten::COOToCSR<kDGLCUDA> (src/array/cuda/coo2csr.cu:28-110 = COOSort by row +
 * cusparseXcoo2csr), which UnitGraph::GetInCSR / GetOutCSR run on first use
 * (src/graph/unit_graph.cc:1418-1450).  Stable: edges keep their COO order inside a row.
 *   row / col   [nnz] of idtype_bits: the major index (compressed) and the minor one; pass
 *               (dst, src) for the in-edge CSR ("CSC"), (src, dst) for the out-edge CSR
 *   eids        optional [nnz] edge ids of the COO entries (NULL: position)
 *   indptr      out [num_rows + 1];  indices, eids_out: out [nnz] — eids_out[i] is the edge id
 *               of CSR position i (the reference's csr.data)
 * `workspace` may be NULL (stream-ordered scratch for the duration of the call). */
size_t dgla_coo_to_csr_workspace_bytes(int idtype_bits, int64_t num_rows, int64_t nnz);
int dgla_coo_to_csr(int idtype_bits, int64_t num_rows, int64_t nnz, const void* row, const void* col,
                    const void* eids, void* indptr, void* indices, void* eids_out, void* workspace,
                    size_t workspace_bytes, void* hip_stream);
/* The same with the caller's promise that every minor id (`col`) is below num_minor: with int64 ids
 * and num_rows, num_minor, nnz < 2^31 the sort then runs on 32-bit keys with the minor id packed into
 * its 64-bit value, like the int32 form (3.35 -> ~2.1 ms at 62 M edges). */
int dgla_coo_to_csr_bounded(int idtype_bits, int64_t num_rows, int64_t num_minor, int64_t nnz, const void* row,
                            const void* col, const void* eids, void* indptr, void* indices, void* eids_out,
                            void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- uniform neighbour sampling and block construction -------------------
 * Replace CSRRowWiseSamplingUniform<kDGLCUDA> (src/array/cuda/rowwise_sampling.cu:43-330, behind
 * dgl.sampling.sample_neighbors) and ToBlock<kDGLCUDA> (src/graph/transform/cuda/cuda_to_block.cu,
 * behind dgl.to_block) for the uniform, single-relation case of mini-batch GraphSAGE.
 *
 * dgla_sample_neighbors: for every seed (a row of the in-edge CSR `csc`) picks min(deg, fanout)
 *   in-neighbours uniformly without replacement (fanout with replacement when `replace`;
 *   fanout = -1: all).  Output is itself a CSR over the seeds: out_indptr [num_seeds + 1],
 *   out_src (GLOBAL source ids) and out_eids (edge ids) of out_indptr[num_seeds] entries —
 *   allocate num_seeds * fanout (or call once with out_src == NULL to get out_indptr only, for
 *   fanout = -1).  The picks depend only on (rng_seed, seed node): reproducible.
 * dgla_to_block: renumbers `src` (global ids, nnz entries) into block-local ids with the seeds
 *   first in the given order and the remaining nodes by ascending id.  `node_map` is caller-owned
 *   int32 scratch with one entry per node of the graph, all -1 on entry and again on exit;
 *   src_nodes [>= num_seeds + nnz] receives the global id of every local source node,
 *   *num_src_out (device memory) their number. */
 /* dgla_sample_neighbors_weighted: the same with per-edge probabilities `prob` (float32 / float64,
 *   indexed by EDGE ID like every edge feature) — CSRRowWiseSampling<kDGLCUDA> of
 *   src/array/cuda/rowwise_sampling_prob.cu: without replacement the A-Res rule (here in its
 *   exponential-clock form, keys recomputed from a counter-based generator instead of stored and
 *   sorted), with replacement inverse-CDF picks; edges with prob <= 0 are never returned, so a row
 *   yields min(fanout, #positive edges) entries (the reference removes them after sampling).
 *   Call with out_src == NULL first to get out_indptr when the caller does not want to
 *   over-allocate num_seeds * fanout. */
size_t dgla_sample_neighbors_workspace_bytes(int idtype_bits, int64_t num_seeds);
int dgla_sample_neighbors_weighted(const dgla_csr* csc, const void* prob, dgla_dtype prob_dtype,
                                   const void* seeds, int64_t num_seeds, int fanout, int replace,
                                   uint64_t rng_seed, void* out_indptr, void* out_src, void* out_eids,
                                   void* workspace, size_t workspace_bytes, void* hip_stream);
int dgla_sample_neighbors(const dgla_csr* csc, const void* seeds, int64_t num_seeds, int fanout,
                          int replace, uint64_t rng_seed, void* out_indptr, void* out_src,
                          void* out_eids, void* workspace, size_t workspace_bytes, void* hip_stream);
size_t dgla_to_block_workspace_bytes(int idtype_bits, int64_t nnz);
int dgla_to_block(int idtype_bits, const void* seeds, int64_t num_seeds, const void* src, int64_t nnz,
                  void* node_map, void* local_src, void* src_nodes, int64_t* num_src_out,
                  void* workspace, size_t workspace_bytes, void* hip_stream);

/* Padded (static-shape) forms of the two calls above: no size ever travels to the host, so a whole
 * mini-batch step — sample, build blocks, gather, g-SpMM, backward, update — can be captured in ONE
 * hipGraph and replayed (the reference synchronises after every sampling layer to size the next
 * allocation: python/dgl/dataloading/neighbor_sampler.py sample_blocks -> cuda_to_block.cu).
 * dgla_sample_neighbors_padded: `seeds` has num_seeds SLOTS of which the first *num_valid (device
 *   memory; NULL = all) are real — the rest are padding (any valid node id) and pick nothing.
 *   fanout > 0.  out_src / out_eids hold cap = num_seeds * fanout entries: the picks in
 *   [0, out_indptr[num_seeds]), then the edges of `sink_rows` extra SINK rows (rows num_seeds ..
 *   num_seeds + sink_rows - 1 in equal shares; out_indptr has num_seeds + 1 + sink_rows entries,
 *   the last = cap) that point at the real seeds in turn.  A consumer sees a well-formed CSR with
 *   num_seeds + sink_rows rows and cap edges whose real rows are exactly the unpadded call's; the
 *   sink rows' results are garbage to be ignored.  The draw counter is
 *   rng_seed + *rng_counter * 0x9E3779B97F4A7C15 (rng_counter: device int64 or NULL), so a
 *   captured launch samples afresh whenever the caller bumps the counter on the device.
 *   `prob` != NULL: weighted picks as dgla_sample_neighbors_weighted (float32 / float64 per EDGE ID).
 *   `workspace` (dgla_sample_neighbors_workspace_bytes) is REQUIRED: the call allocates nothing.
 * dgla_to_block_padded: like dgla_to_block over all cap entries of the padded `src`; only the first
 *   *num_valid seeds claim a local id (padding slots keep their row — the destination nodes of the
 *   block are still its first num_seeds source nodes — but edges never resolve to them); new nodes
 *   get ids from num_seeds upwards; entries of src_nodes past *num_src_out are left untouched
 *   (pre-fill the buffer with a valid node id).  num_nodes > 0 (the graph's node count) lets the
 *   sort look at the used key bits only.  `workspace` (dgla_to_block_workspace_bytes(nnz))
 *   is required. */
int dgla_sample_neighbors_padded(const dgla_csr* csc, const void* prob, dgla_dtype prob_dtype, const void* seeds,
                                 int64_t num_seeds, const int64_t* num_valid, int fanout, int replace,
                                 uint64_t rng_seed, const int64_t* rng_counter, int sink_rows, void* out_indptr,
                                 void* out_src, void* out_eids, void* workspace, size_t workspace_bytes,
                                 void* hip_stream);
int dgla_to_block_padded(int idtype_bits, const void* seeds, int64_t num_seeds, const int64_t* num_valid,
                         const void* src, int64_t nnz, int64_t num_nodes, void* node_map, void* local_src,
                         void* src_nodes, int64_t* num_src_out, void* workspace, size_t workspace_bytes,
                         void* hip_stream);

/* ---- k-way node-cut partitioner (host code) ---------------------------------
 * Stands where METIS stands in the reference: metis_partition_assignment
 * (python/dgl/partition.py:278-397 -> _CAPI_DGLMetisPartition_Hetero).  Multilevel
 * size-constrained label propagation (csrc/partition.cc).  The CSR (HOST pointers, rows =
 * destination nodes, square) is symmetrised internally like the reference does.
 *   imbalance      allowed excess of a part's weight over the average, e.g. 0.03
 *   balance_edges  vertex weight = 1 + in-degree instead of 1 (the reference's flag)
 *   out_part       [num_nodes] int64, the part of every node
 *   stats          optional [4]: cut edges, heaviest part weight, average part weight, levels */
int dgla_partition_kway(int idtype_bits, int64_t num_nodes, const void* indptr,
                        const void* indices, int num_parts, double imbalance, int balance_edges,
                        uint64_t seed, int64_t* out_part, int64_t* stats);
/* The reference's other METIS options (python/dgl/partition.py:278-397 -> src/graph/metis_partition.cc:60-90):
 *   objtype     0 = "cut" (edge cut), 1 = "vol" (total communication volume: the number of distinct remote
 *               columns the parts read = the rows a row-sharded SpMM exchanges per step);
 *   ntype       (may be NULL) node type of every node, values in [0, num_ntypes): balance_ntypes — every node
 *               type is balanced across the parts by its own constraint;
 *   init_part   (may be NULL) an assignment to REFINE instead of running the multilevel scheme (e.g. contiguous
 *               ranges of a graph whose vertex order already reflects its structure).
 * Refinement under either objective is a greedy k-way pass over the directed CSR with exact gains from a
 * per-(column, part) reference-count table.  stats8 = {cut edges, heaviest part, average part, levels, total
 * volume, largest halo (distinct remote columns of one part), largest (part, type) excess over its limit,
 * refinement moves}. */
int dgla_partition_kway_ex(int idtype_bits, int64_t num_nodes, const void* indptr, const void* indices,
                           int num_parts, double imbalance, int balance_edges, uint64_t seed, int objtype,
                           int num_ntypes, const int32_t* ntype, const int64_t* init_part, int64_t* out_part,
                           int64_t* stats8);

/* ---- multi-GPU exchange step -----------------------------------------
 * Device side of the halo all-to-all: the collective itself is RCCL (torch.distributed), these
 * are the pack / unpack kernels around it and the NDArrayPartition index maps.
 *
 * dgla_gather_rows: dst[i, :] = src[idx[i], :], rows of `row_bytes` bytes — `value[perm]`,
 *   `value[resp_idx]` in python/dgl/cuda/nccl.py:69-70,176 (torch index kernels there); the
 *   scatter-add direction of the gradient push is dgla_scatter_add above.
 * dgla_partition_map / dgla_partition_to_global: MapToLocalFromRemainder / ...FromRange and
 *   MapToGlobal... of src/partition/cuda/partition_op.cu (declared src/partition/partition_op.h:
 *   36-130; objects src/partition/ndarray_partition.cc:30-230).
 *   mode 0 = remainder: part = id % num_parts, local = id / num_parts
 *   mode 1 = range:     part = the p with range[p] <= id < range[p + 1], local = id - range[p];
 *                       `range` = DEVICE array of num_parts + 1 ids (idtype_bits), as the
 *                       reference requires (ndarray_partition.cc:104-112)
 *   part_out / local_out may be NULL.  GeneratePermutation = dgla_partition_map (parts) +
 *   dgla_coo_to_csr over the part ids (a stable sort: eids_out is the permutation, indptr the
 *   prefix of the per-part counts). */
int dgla_gather_rows(int idtype_bits, const void* src, const void* idx, int64_t n,
                     int64_t row_bytes, void* dst, void* hip_stream);
/* dst[idx[i], :] = src[i, :] (idx a permutation, or at least injective). */
int dgla_scatter_rows(int idtype_bits, const void* src, const void* idx, int64_t n,
                      int64_t row_bytes, void* dst, void* hip_stream);

int dgla_partition_map(int idtype_bits, int mode, int num_parts, const void* range, const void* idx,
                       int64_t n, void* part_out, void* local_out, void* hip_stream);
int dgla_partition_to_global(int idtype_bits, int mode, int num_parts, const void* range,
                             const void* local_idx, int64_t n, int part_id, void* out,
                             void* hip_stream);

/* ---- peer-mapped halo exchange (replaces the value exchange of sparse_all_to_all_pull,
 * python/dgl/cuda/nccl.py:98-183, when every GPU of the node can map every other GPU's memory) ----
 * dgla_peer_alloc: device memory that can be exported (kind 0 = hipMalloc, 1 = fine-grained, 2 = uncached;
 *   falls back to hipMalloc when the flavour is refused), zero filled.  dgla_ipc_export / _import / _release =
 *   hipIpcGetMemHandle / OpenMemHandle / CloseMemHandle on 64-byte handles.
 * dgla_peer_push: ONE launch writes rows serve_rows[seg.row_begin .. seg.row_end) of x_local into every
 *   segment's destination (a peer's halo buffer) and then stores `epoch` into the segment's flag in the peer's
 *   memory.  `segments` = DEVICE array of num_segments records {int64 row_begin, row_end; void* dst;
 *   uint64* flag; int64 blk_begin} with blk_begin = exclusive prefix of max(1, ceil(rows / 64)), num_blocks
 *   its total; `arrive` = num_segments zeroed uint32 counters in device memory (kept zero between calls).
 * dgla_peer_wait: one wavefront on the stream polls flags[0 .. num_flags) until each is >= epoch (bounded:
 *   after max_spins polls *status (device int) becomes 1 and the stream continues). */
int dgla_peer_alloc(size_t bytes, int kind, void** out);
int dgla_peer_free(void* ptr);
int dgla_ipc_export(void* ptr, void* handle64);
int dgla_ipc_import(const void* handle64, void** out);
int dgla_ipc_release(void* ptr);
int dgla_peer_push(const void* x_local, int64_t row_bytes, const int64_t* serve_rows, const void* segments,
                   int num_segments, int64_t num_blocks, uint64_t epoch, void* arrive, void* hip_stream);
int dgla_peer_wait(const void* flags, int num_flags, uint64_t epoch, void* status, int64_t max_spins,
                   void* hip_stream);

/* Process-wide tuning bits.  The SpMM bits change no result bit; they select memory-system behaviour
 * and exist so that a benchmark can A/B them:
 *   DGLA_TUNE_XCD     units visit the merge path in XCD-contiguous order (block b runs on XCD
 *                     b % 8, so each XCD's L2 sees one contiguous eighth of the rows)
 *   DGLA_TUNE_SPLIT   when a gathered ufeat row touches one 128-byte line more than its length needs
 *                     (F = 100 fp32: 400 B = 4 lines per gather) the call first copies the rows' ragged
 *                     ends (16-byte lanes, rows of >= 256 bytes: one side line per row + a dense array of
 *                     the last bytes; the line-aligned interior is gathered in place) or the straddling
 *                     rows (8-byte lanes, e.g. bf16 F = 100) into the workspace — unless the locality
 *                     probe made with the merge plan found at least 15/16 of the sampled edges within
 *                     64 Ki rows of their own row (the copy costs 0.1 ms on the headline graph and pays
 *                     even at 81 % local edges)
 *   DGLA_TUNE_SPLIT_FORCE  ignore the locality probe
 *   DGLA_TUNE_GLDS    dgla_segment_mm / dgla_gather_mm, 16-bit and fp32 storage, operands in
 *                     whole aligned 16-byte pieces: operands go global -> LDS directly (global_load_lds,
 *                     slab rings) instead of through registers; 16-bit results bit-identical,
 *                     fp32 contracts k in a permuted order; the 16-bit weight gradient reads
 *                     its fragments with transposing LDS loads (default on)
 *   DGLA_TUNE_MM_F32  dgla_segment_mm / dgla_gather_mm and their gradients, fp32: multiply on
 *                     v_mfma_f32_32x32x2_f32.  Default (bit off): every fp32 operand is split into
 *                     three round-to-nearest bf16 terms and the six products of order <= 2 run on
 *                     v_mfma_f32_32x32x16_bf16 with fp32 accumulation — fp32-level accuracy (dropped
 *                     terms < 2^-26 |a b|) at 2.7x less matrix-pipe time; gfx950 has no xf32 MFMA
 *   DGLA_TUNE_MM_X3   dgla_segment_mm / dgla_gather_mm forward, fp32, weights-stationary kernel (long inputs, K <= 256):
 *                     the three-bf16-term split above.  Default (bit off, round 5): TWO fp16 terms under exact per-row /
 *                     per-column power-of-two scales — three products (hh, hl, lh) instead of six, dropped term
 *                     < 2^-22 |a b|; a row whose non-zero magnitudes span more than 2^18, or whose maximum is not finite
 *                     or outside 2^+-60, is recomputed as the plain fp32 dot product inside the same launch
 *   DGLA_TUNE_NO_GATE dgla_spmm_csr_masked (max / min backward as a gather): gather every piece of every row.  Default (bit
 *                     off, round 6): the bit words are fetched one batch ahead and a lane whose columns are all off for an
 *                     edge reads a fixed cached address instead of its piece of the row — a 128-byte line of the gathered
 *                     operand is requested only when one of its columns is wanted.  Changes no result bit (A/B switch).
 *   DGLA_TUNE_NO_STAGE_W  dgla_spmm_csr with ONE 4-byte edge weight per edge and the sum reducer (u_mul_e / u_add_e + sum): read the
 *                     weight from global memory in every gather batch.  Default (bit off, round 6): the unit's weights are
 *                     staged in LDS together with its column ids (one load per edge).  Changes no result bit (A/B switch).
 * Removed in round 4 (values retired, dgla_set_tuning rejects them): NT_OUT 2 and NT_IDX 4 (non-temporal
 * output-row stores / index-stream loads: measured neutral), SPLIT_NT 32, SPLIT_CLASSIC 256
 * (whole-row copy: 0.33 ms against 0.10), TAIL_PASS 512 (column-sliced pass over the 16-byte row tails:
 * -5 % .. +3.5 %), NT_STREAM 1024 (now a fixed rule: copy_rhs without an edge-id map over rows of >= 64
 * edges on average loads the edge operand non-temporally; measured 1.10 -> 1.01 ms on a 64-segment sum,
 * +7 .. 10 % on short rows, hence the rule).  The reference has no counterpart (its kernels take no hints). */
#define DGLA_TUNE_XCD 1u
#define DGLA_TUNE_SPLIT 8u
#define DGLA_TUNE_GLDS 16u
#define DGLA_TUNE_SPLIT_FORCE 64u
#define DGLA_TUNE_MM_F32 128u
#define DGLA_TUNE_MM_X3 2048u
#define DGLA_TUNE_NO_GATE 4096u
#define DGLA_TUNE_NO_STAGE_W 8192u
int dgla_set_tuning(uint32_t flags);
uint32_t dgla_get_tuning(void);

/* dgla_spmm_csr / dgla_segment_reduce calls of this process served by the narrow-feature kernels (csrc/narrow_reduce.hip:
 * 1 ... 8 fp32 output columns — copy_rhs, copy_lhs, u (+ - * /) e —, one lane per four EDGES instead of one lane per column; same results as the
 * merge kernel up to the order of fp32 additions, winners and their edge ids identical).  A counter for tests and
 * benches to see which kernel family took a call; DGLA_NARROW_REDUCE=0 in the environment keeps the merge kernel.
 * (DGLA_NARROW_STAGE=0: edge rows of 4 / 8 columns in position order are gathered lane by lane instead of being fetched as
 * whole wavefront loads and transposed through LDS — an A/B switch, same bits either way.)
 * The reference has one kernel for every width (src/array/cuda/spmm.cuh:440-520). */
int64_t dgla_narrow_reduce_calls(void);

/* Benchmark hook: when both are non-NULL (hipEvent_t), the calling thread's next
 * dgla_spmm_csr calls record `before` / `after` on the launch stream around the dominant
 * (merge) kernel only, so its duration can be read with hipEventElapsedTime.  NULL disables. */
int dgla_spmm_set_profile_events(void* before, void* after);

/* Measured-peak helper used by bench.py: streams `bytes` from src to dst with 16-byte
 * lane accesses (the "float4 copy" the HBM roofline is quoted against). */
int dgla_stream_copy(void* dst, const void* src, size_t bytes, void* hip_stream);
/* Same with an explicit kernel variant (benchmarks/bench_peak.py): bits 0-1 = 0 copy with
 * non-temporal accesses, 1 copy with default cache policy, 2 read-only (dst receives one
 * 16-byte value per lane of the grid: needs >= 32 MiB); bits 2-3 = log2(blocks per CU) - 2;
 * bit 4 = 8 instead of 4 loads in flight per lane. */
int dgla_stream_copy_variant(void* dst, const void* src, size_t bytes, int variant,
                             void* hip_stream);

/* ======================================================================================
 * (2) Registry layer — same names and layouts as include/dgl/runtime/c_runtime_api.h.
 * ====================================================================================== */

/* type codes (c_runtime_api.h:66-91) */
typedef enum {
  kObjectInt = 0,
  kObjectUInt = 1,
  kObjectFloat = 2,
  kHandle = 3,
  kNull = 4,
  kDGLDataType = 5,
  kDGLContext = 6,
  kArrayHandle = 7,
  kObjectHandle = 8,
  kModuleHandle = 9,
  kFuncHandle = 10,
  kStr = 11,
  kBytes = 12,
  kNDArrayContainer = 13
} DGLTypeCode;

/* Device types.  kDGLROCM = 10 is DLPack's kDLROCM, which PyTorch-ROCm exports; the
 * reference only knows kDGLCPU = 1 / kDGLCUDA = 2 (c_runtime_api.h:51-60). */
typedef enum { kDGLCPU = 1, kDGLCUDA = 2, kDGLROCM = 10 } DGLDeviceType;

typedef struct {
  int32_t device_type;
  int32_t device_id;
} DGLContext;

typedef struct {
  uint8_t code; /* 0 int, 1 uint, 2 float, 4 bfloat (DLPack codes) */
  uint8_t bits;
  uint16_t lanes;
} DGLDataType;

/* DGLValue (c_runtime_api.h:205-212) */
typedef union {
  int64_t v_int64;
  double v_float64;
  void* v_handle;
  const char* v_str;
  DGLDataType v_type;
  DGLContext v_ctx;
} DGLValue;

/* DGLArray (c_runtime_api.h:150-196): DLTensor-compatible. */
typedef struct {
  void* data;
  DGLContext ctx;
  int32_t ndim;
  DGLDataType dtype;
  int64_t* shape;
  int64_t* strides; /* NULL = compact row-major */
  uint64_t byte_offset;
} DGLArray;

typedef void* DGLFunctionHandle;
typedef DGLArray* DGLArrayHandle;

const char* DGLGetLastError(void);
void DGLAPISetLastError(const char* msg);
int DGLFuncListGlobalNames(int* out_size, const char*** out_array);
int DGLFuncGetGlobal(const char* name, DGLFunctionHandle* out);
int DGLFuncCall(DGLFunctionHandle func, DGLValue* args, int* type_codes, int num_args,
                DGLValue* ret_val, int* ret_type_code);
int DGLFuncFree(DGLFunctionHandle func);

/* DLPack hand-over (include/dgl/runtime/dlpack_convert.h:60-80, src/runtime/dlpack_convert.cc:
 * 57-140; the reference's Python wraps every tensor this way, backend/pytorch/tensor.py:432-435).
 * DGLArrayFromDLPack takes ownership of `from` (its deleter runs when the array is freed) and
 * returns a handle that points at a DGLArray — the first member of the array's container, as
 * with NDArray::Container.  device_type travels unchanged: PyTorch-ROCm exports kDLROCM = 10 =
 * kDGLROCM.  DGLArrayToDLPack shares the memory and keeps the array alive until the consumer
 * calls the returned tensor's deleter. */
#ifndef DLPACK_DLPACK_H_
typedef struct {
  int32_t device_type;
  int32_t device_id;
} DLDevice;
typedef struct {
  uint8_t code;
  uint8_t bits;
  uint16_t lanes;
} DLDataType;
typedef struct {
  void* data;
  DLDevice device;
  int32_t ndim;
  DLDataType dtype;
  int64_t* shape;
  int64_t* strides;
  uint64_t byte_offset;
} DLTensor;
typedef struct DLManagedTensor {
  DLTensor dl_tensor;
  void* manager_ctx;
  void (*deleter)(struct DLManagedTensor* self);
} DLManagedTensor;
#endif
int DGLArrayFromDLPack(DLManagedTensor* from, void** out /* DGLArrayHandle* */);
int DGLArrayToDLPack(void* from /* DGLArrayHandle */, DLManagedTensor** out, int alignment);
int DGLArrayFree(void* handle /* DGLArrayHandle made by DGLArrayFromDLPack */);
void DGLDLManagedTensorCallDeleter(DLManagedTensor* dltensor);
/* include/dgl/runtime/c_object_api.h: DGLObjectFree — releases an object handle returned by a
 * registry function: the List<Value> / Value boxes of the global functions `_List` / `_Value`
 * (how the reference passes Python lists, python/dgl/_ffi/object_generic.py:27-59), the
 * heterograph handle of dgl_amd._CAPI_HeteroGraphCreate, a unit-graph handle. */
int DGLObjectFree(void* handle);
/* c_runtime_api.h: DGLSetStream — the stream kernels of this thread are queued on. */
int DGLSetStream(int device_type, int device_id, void* stream);
int DGLGetStream(int device_type, int device_id, void** stream);

#ifdef __cplusplus
}
#endif
#endif /* DGL_AMD_H_ */
