// C-ABI entry points of libdgl_amd.so, layer (1): the graph-free seam (include/dgl_amd.h).
// Host-only code: argument checking (reference: src/array/check.h:19-56 and the lambdas in
// src/array/kernel.cc:473-499,603-626), broadcast analysis (reference: src/bcast.cc:36-90),
// dtype / idtype dispatch (reference: src/array/kernel.cc:20-44,224-248).
#include "../../include/dgl_amd.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace dgla {

std::string& last_error() {
  static thread_local std::string err;
  return err;
}

uint32_t& tuning_flags() {
  static uint32_t flags = kDefaultTuning;
  return flags;
}

ProfileEvents& profile_events() {
  static thread_local ProfileEvents pe;
  return pe;
}

// per-dtype launchers (spmm_<dtype>.hip, sddmm_<dtype>.hip, aux_kernels.hip)
bool narrow_reduce_eligible(const SpmmLaunch&);          // narrow_reduce.hip: copy_e with 1 ... 8 fp32 columns
size_t narrow_reduce_workspace_bytes(const SpmmLaunch&);
int launch_narrow_reduce(const SpmmLaunch&, void* ws);
int64_t narrow_reduce_calls();
int launch_spmm_csr_f32(const SpmmLaunch&);
int launch_spmm_csr_f64(const SpmmLaunch&);
int launch_spmm_csr_f16(const SpmmLaunch&);
int launch_spmm_csr_bf16(const SpmmLaunch&);
size_t spmm_csr_workspace_f32(const SpmmLaunch&);
size_t spmm_csr_workspace_f64(const SpmmLaunch&);
size_t spmm_csr_workspace_f16(const SpmmLaunch&);
size_t spmm_csr_workspace_bf16(const SpmmLaunch&);
int launch_sddmm_f32(const SddmmLaunch&);
int launch_sddmm_f64(const SddmmLaunch&);
int launch_sddmm_f16(const SddmmLaunch&);
int launch_sddmm_bf16(const SddmmLaunch&);
int launch_stream_copy(void*, const void*, size_t, int, hipStream_t);
int launch_spmm_coo(const CooView&, int, int, int, const void*, const void*, void*, void*, void*,
                    int64_t, int64_t, int64_t, bool, const BcastDims&, hipStream_t);
int launch_edge_softmax(const CsrView&, int, const void*, const void*, void*, int64_t, bool,
                        void*, size_t, bool, hipStream_t, bool out_pos = false, bool b_is_grad = false);
size_t edge_softmax_workspace_bytes(int64_t, int64_t, int, int64_t);

static int fail(const std::string& msg) {
  last_error() = msg;
  return -1;
}

static int parse_op(const char* s, bool allow_dot) {
  if (!s) return -1;
  if (!strcmp(s, "add")) return kAdd;
  if (!strcmp(s, "sub")) return kSub;
  if (!strcmp(s, "mul")) return kMul;
  if (!strcmp(s, "div")) return kDiv;
  if (!strcmp(s, "copy_lhs")) return kCopyLhs;
  if (!strcmp(s, "copy_rhs")) return kCopyRhs;
  if (allow_dot && !strcmp(s, "dot")) return kDot;
  return -1;
}

static int parse_reduce(const char* s) {
  if (!s) return -1;
  if (!strcmp(s, "sum")) return kSum;
  if (!strcmp(s, "max")) return kMax;
  if (!strcmp(s, "min")) return kMin;
  return -1;
}

// An operand is absent when the descriptor is NULL or has no dimensions; an EMPTY tensor
// (0 rows, data may be NULL) is present.
static bool present(const dgla_tensor* t) {
  if (!t || t->ndim <= 0 || !t->shape) return false;
  if (t->data) return true;
  for (int i = 0; i < t->ndim; ++i)
    if (t->shape[i] == 0) return true;
  return false;
}

static int64_t feat_len(const dgla_tensor* t) {
  int64_t n = 1;
  for (int i = 1; i < t->ndim; ++i) n *= t->shape[i];
  return n;
}

// Result of the broadcast analysis: the BcastOff fields of include/dgl/bcast.h:20-53 plus
// the closed-form description our kernels use instead of offset tables.
struct BcastInfo {
  int64_t lhs_len, rhs_len, out_len, reduce_size;
  bool use_bcast;
  int mode;       // Bcast
  int rhs_group;  // kBcRhsGroup
  BcastDims dims;
};

// Mirrors CalcBcastOff (src/bcast.cc:36-90).  `lhs` / `rhs` are feature shapes WITHOUT the
// leading node/edge dimension.
static int analyse_bcast(int op, const std::vector<int64_t>& lhs, const std::vector<int64_t>& rhs,
                         BcastInfo* b) {
  b->lhs_len = b->rhs_len = 1;
  for (int64_t d : lhs) b->lhs_len *= d;
  for (int64_t d : rhs) b->rhs_len *= d;
  b->reduce_size = 1;
  b->mode = kBcNone;
  b->rhs_group = 1;
  b->dims.ndim = 0;
  bool use = false;
  if (op != kCopyLhs && op != kCopyRhs) use = lhs != rhs;  // bcast.cc:18-26
  b->use_bcast = use;
  if (!use) {
    b->out_len = (op == kCopyRhs) ? b->rhs_len : b->lhs_len;
    if (op == kDot) {
      if (lhs.empty()) return fail("dot needs at least one feature dimension");
      b->reduce_size = lhs.back();
      b->out_len /= b->reduce_size;
    }
    return 0;
  }
  // right-align both shapes; for dot the last axis is the reduce axis and takes no part in
  // the offsets, which are then counted in units of reduce_size (bcast.cc:50-54, sddmm.hip.h).
  std::vector<int64_t> l = lhs, r = rhs;
  if (op == kDot) {
    if (l.empty() || r.empty() || l.back() != r.back())
      return fail("dot: the last dimensions of lhs and rhs must match");
    b->reduce_size = l.back();
    l.pop_back();
    r.pop_back();
  }
  const size_t nd = l.size() > r.size() ? l.size() : r.size();
  l.insert(l.begin(), nd - l.size(), 1);
  r.insert(r.begin(), nd - r.size(), 1);
  std::vector<int64_t> od(nd);
  for (size_t i = 0; i < nd; ++i) {
    if (l[i] != r[i] && l[i] != 1 && r[i] != 1)
      return fail("feature shapes are not valid for broadcasting");
    od[i] = l[i] > r[i] ? l[i] : r[i];
  }
  b->out_len = 1;
  for (int64_t d : od) b->out_len *= d;
  // strides (in elements; in reduce_size units for dot), 0 on broadcast axes
  std::vector<int64_t> ls(nd), rs(nd);
  int64_t sl = 1, sr = 1;
  for (size_t i = nd; i-- > 0;) {
    ls[i] = l[i] == 1 ? 0 : sl;
    rs[i] = r[i] == 1 ? 0 : sr;
    sl *= l[i];
    sr *= r[i];
  }
  // drop size-1 output axes, then merge neighbours that are contiguous for both operands
  std::vector<int64_t> md, mls, mrs;
  for (size_t i = 0; i < nd; ++i) {
    if (od[i] == 1) continue;
    if (!md.empty()) {
      const size_t j = md.size() - 1;
      const bool lm = (mls[j] == ls[i] * od[i]) || (mls[j] == 0 && ls[i] == 0);
      const bool rm = (mrs[j] == rs[i] * od[i]) || (mrs[j] == 0 && rs[i] == 0);
      if (lm && rm) {
        md[j] *= od[i];
        mls[j] = ls[i];
        mrs[j] = rs[i];
        continue;
      }
    }
    md.push_back(od[i]);
    mls.push_back(ls[i]);
    mrs.push_back(rs[i]);
  }
  if (md.size() > static_cast<size_t>(kMaxBcastDims))
    return fail("broadcast pattern needs more than 6 independent axes");
  b->dims.ndim = static_cast<int>(md.size());
  for (size_t i = 0; i < md.size(); ++i) {
    b->dims.dims[i] = static_cast<int32_t>(md[i]);
    b->dims.lstride[i] = static_cast<int32_t>(mls[i]);
    b->dims.rstride[i] = static_cast<int32_t>(mrs[i]);
  }
  b->mode = kBcGeneral;
  // rhs broadcast over a trailing block of lhs (e.g. (H, D) x (H, 1), or rhs scalar):
  // after merging this is exactly {dims = [A, G], l = [G, 1], r = [1, 0]} or {[G], [1], [0]}.
  if (op != kDot) {
    if (md.size() == 1 && mls[0] == 1 && mrs[0] == 0) {
      b->mode = kBcRhsGroup;
      b->rhs_group = static_cast<int>(md[0]);
    } else if (md.size() == 2 && mls[0] == md[1] && mls[1] == 1 && mrs[0] == 1 && mrs[1] == 0) {
      b->mode = kBcRhsGroup;
      b->rhs_group = static_cast<int>(md[1]);
    }
  }
  return 0;
}

static std::vector<int64_t> feat_shape(const dgla_tensor* t) {
  return std::vector<int64_t>(t->shape + 1, t->shape + t->ndim);
}

static int check_tensor(const dgla_tensor* t, const char* name, int64_t dim0) {
  // src/array/check.h:39-56: ndim >= 2 and the first dimension matches the graph
  if (t->ndim < 2) return fail(std::string(name) + ": expect at least 2 dimensions");
  if (t->shape[0] != dim0)
    return fail(std::string(name) + ": first dimension " + std::to_string(t->shape[0]) +
                " does not match the graph (" + std::to_string(dim0) + ")");
  return 0;
}

static int fill_csr(const dgla_csr* c, CsrView* v) {
  if (!c) return fail("csr is null");
  if (c->idtype_bits != 32 && c->idtype_bits != 64) return fail("idtype must be int32 or int64");
  if (c->num_cols > 0x7fffffffLL) return fail("more than 2^31-1 columns are not supported");
  if (!c->indptr || (c->nnz > 0 && !c->indices)) return fail("csr arrays are null");
  v->num_rows = c->num_rows;
  v->num_cols = c->num_cols;
  v->nnz = c->nnz;
  v->idbits = c->idtype_bits;
  v->indptr = c->indptr;
  v->indices = c->indices;
  v->eids = c->data;
  return 0;
}

static int fill_coo(const dgla_coo* c, CooView* v) {
  if (!c) return fail("coo is null");
  if (c->idtype_bits != 32 && c->idtype_bits != 64) return fail("idtype must be int32 or int64");
  if (c->nnz > 0 && (!c->row || !c->col)) return fail("coo arrays are null");
  v->num_rows = c->num_rows;
  v->num_cols = c->num_cols;
  v->nnz = c->nnz;
  v->idbits = c->idtype_bits;
  v->row = c->row;
  v->col = c->col;
  v->eids = c->data;
  return 0;
}

// Common SpMM argument analysis.  num_src / num_dst / nnz come from the sparse matrix.
static int prepare_spmm(int op, int64_t num_src, int64_t num_dst, int64_t nnz,
                        const dgla_tensor* ufeat, const dgla_tensor* efeat,
                        const dgla_tensor* out, BcastInfo* bc) {
  const bool ul = op_uses_lhs(op), ur = op_uses_rhs(op);
  if (ul && !present(ufeat)) return fail("operator needs the source-node feature (ufeat)");
  if (ur && !present(efeat)) return fail("operator needs the edge feature (efeat)");
  if (!present(out)) return fail("out is null");
  if (ul && check_tensor(ufeat, "ufeat", num_src)) return -1;
  if (ur && check_tensor(efeat, "efeat", nnz)) return -1;
  if (check_tensor(out, "out", num_dst)) return -1;
  const std::vector<int64_t> l = ul ? feat_shape(ufeat) : feat_shape(efeat);
  const std::vector<int64_t> r = ur ? feat_shape(efeat) : feat_shape(ufeat);
  if (analyse_bcast(op, l, r, bc)) return -1;
  if (feat_len(out) != bc->out_len)
    return fail("out has " + std::to_string(feat_len(out)) + " features per row, expected " +
                std::to_string(bc->out_len));
  if (bc->out_len > 0x7fffffffLL / 4) return fail("feature length too large");
  return 0;
}

static int build_spmm_launch(const char* op_s, const char* red_s, const dgla_csr* csr,
                             dgla_dtype dtype, const dgla_tensor* ufeat,
                             const dgla_tensor* efeat, const dgla_tensor* out, SpmmLaunch* L) {
  const int op = parse_op(op_s, false);
  if (op < 0) return fail(std::string("Unsupported SpMM binary operator: ") + (op_s ? op_s : "(null)"));
  const int red = parse_reduce(red_s);
  if (red < 0) return fail(std::string("Unsupported SpMM reducer: ") + (red_s ? red_s : "(null)"));
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return fail("unsupported feature dtype");
  if (fill_csr(csr, &L->csr)) return -1;
  BcastInfo bc;
  if (prepare_spmm(op, csr->num_cols, csr->num_rows, csr->nnz, ufeat, efeat, out, &bc)) return -1;
  L->op = op;
  L->red = red;
  L->dtype = dtype;
  L->ufeat = op_uses_lhs(op) ? ufeat->data : nullptr;
  L->efeat = op_uses_rhs(op) ? efeat->data : nullptr;
  L->out = out->data;
  L->out_len = bc.out_len;
  L->lhs_len = bc.lhs_len;
  L->rhs_len = bc.rhs_len;
  L->bcast = bc.mode;
  L->rhs_group = bc.rhs_group;
  L->bdims = bc.dims;
  L->tune = tuning_flags();
  return 0;
}

}  // namespace dgla

using namespace dgla;

// ---- stacked max / min: winning stacked position -> (column, edge id, node type, edge type) ----
struct StackedTypes {
  int32_t src_ntype[256];
  int32_t etype[256];
};

template <typename Idx>
__global__ void stacked_args_kernel(Idx* __restrict__ pos_buf, const Idx* __restrict__ indices,
                                    const Idx* __restrict__ eids, const uint8_t* __restrict__ rel,
                                    Idx* __restrict__ arg_u, Idx* __restrict__ arg_e,
                                    Idx* __restrict__ arg_u_ntype, Idx* __restrict__ arg_e_etype,
                                    const StackedTypes types, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
    const int64_t pos = static_cast<int64_t>(pos_buf[i]);  // read before any tracker aliasing it is written
    Idx u = 0, e = 0, nt = -1, et = -1;  // no edge: args stay 0, trackers -1 (spmm_hetero.cu:100-117)
    if (pos >= 0) {
      const int r = rel[pos];
      if (arg_u) u = indices[pos];
      if (arg_e) e = eids ? eids[pos] : static_cast<Idx>(pos);
      nt = static_cast<Idx>(types.src_ntype[r]);
      et = static_cast<Idx>(types.etype[r]);
    }
    if (arg_u) arg_u[i] = u;
    if (arg_e) arg_e[i] = e;
    if (arg_u_ntype) arg_u_ntype[i] = nt;
    if (arg_e_etype) arg_e_etype[i] = et;
  }
}

extern "C" {

const char* dgla_last_error(void) { return last_error().c_str(); }
int64_t dgla_narrow_reduce_calls(void) { return narrow_reduce_calls(); }
int dgla_abi_version(void) { return DGLA_ABI_VERSION; }

size_t dgla_spmm_csr_workspace_bytes(const char* op, const char* reduce, const dgla_csr* csr,
                                     dgla_dtype dtype, const dgla_tensor* ufeat,
                                     const dgla_tensor* efeat, const dgla_tensor* out) {
  SpmmLaunch L{};
  if (build_spmm_launch(op, reduce, csr, dtype, ufeat, efeat, out, &L)) return 0;
  // narrow copy_e (narrow_reduce.hip): its records go BEHIND what the merge kernel would take for this call, so that the
  // merge plan at the front of a workspace shared by all the operators of a graph stays valid
  if (narrow_reduce_eligible(L))
    return ((spmm_csr_workspace_f32(L) + 255) & ~static_cast<size_t>(255)) + narrow_reduce_workspace_bytes(L);
  switch (dtype) {
    case DGLA_F32: return spmm_csr_workspace_f32(L);
    case DGLA_F64: return spmm_csr_workspace_f64(L);
    case DGLA_F16: return spmm_csr_workspace_f16(L);
    case DGLA_BF16: return spmm_csr_workspace_bf16(L);
  }
  return 0;
}

int dgla_spmm_csr(const char* op, const char* reduce, const dgla_csr* csr, dgla_dtype dtype,
                  const dgla_tensor* ufeat, const dgla_tensor* efeat, const dgla_tensor* out,
                  void* arg_u, void* arg_e, void* workspace, size_t workspace_bytes,
                  uint32_t flags, void* hip_stream) {
  SpmmLaunch L{};
  if (build_spmm_launch(op, reduce, csr, dtype, ufeat, efeat, out, &L)) return -1;
  if (L.red != kSum) {
    if (op_uses_lhs(L.op) && !arg_u) return fail("arg_u is required for max/min");
    if (op_uses_rhs(L.op) && !arg_e) return fail("arg_e is required for max/min");
    if (flags & DGLA_ACCUMULATE) return fail("DGLA_ACCUMULATE is only defined for reduce == sum");
  }
  L.arg_u = arg_u;
  L.arg_e = arg_e;
  L.accumulate = (flags & DGLA_ACCUMULATE) != 0;
  L.mean = (flags & DGLA_MEAN) != 0;
  if (L.mean && (L.red != kSum || L.accumulate))
    return fail("DGLA_MEAN is defined for reduce == sum without DGLA_ACCUMULATE");
  L.plan_valid = (flags & DGLA_PLAN_VALID) != 0;
  L.split_valid = (flags & DGLA_SPLIT_VALID) != 0 && L.plan_valid;
  L.split_keep = L.split_valid || (flags & DGLA_SPLIT_KEEP) != 0;
  L.prepare_only = (flags & DGLA_PREPARE_ONLY) != 0;
  if (L.prepare_only && !(flags & DGLA_SPLIT_KEEP)) return fail("DGLA_PREPARE_ONLY needs DGLA_SPLIT_KEEP");
  L.workspace = workspace;
  L.workspace_bytes = workspace_bytes;
  L.stream = static_cast<hipStream_t>(hip_stream);
  if (csr->num_rows == 0 || L.out_len == 0) return 0;
  const DeviceGuard dev(L.stream, L.out);
  if (narrow_reduce_eligible(L)) {
    const size_t front = (spmm_csr_workspace_f32(L) + 255) & ~static_cast<size_t>(255);
    if (workspace && workspace_bytes >= front + narrow_reduce_workspace_bytes(L)) {
      // the contract of this entry point: after a successful call the workspace holds the graph's merge plan (callers
      // pass DGLA_PLAN_VALID from then on, whatever operator comes next) — so a first call builds it here as well
      if (!L.plan_valid) {
        SpmmLaunch P = L;
        P.prepare_only = true;
        if (const int rc = launch_spmm_csr_f32(P)) return rc;
      }
      L.arg_empty = 0;
      return launch_narrow_reduce(L, static_cast<char*>(workspace) + front);
    }   // (a caller that sized its workspace before this route existed: the merge kernel takes the call)
  }
  switch (dtype) {
    case DGLA_F32: return launch_spmm_csr_f32(L);
    case DGLA_F64: return launch_spmm_csr_f64(L);
    case DGLA_F16: return launch_spmm_csr_f16(L);
    case DGLA_BF16: return launch_spmm_csr_bf16(L);
  }
  return fail("unsupported feature dtype");
}

// g-SpMM (copy_lhs, sum) gated per edge and output column by bit masks (dgla_spmm_cmp_mask writes them): the merge-path
// kernel's `mul` with a grouped rhs, the rhs word read as bits (spmm_csr.hip.h: rhs_mask).
static int build_masked_launch(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ufeat, const void* mask,
                               const dgla_tensor* out, int64_t* shapes, dgla_tensor* views, SpmmLaunch* L) {
  if (!csr) return fail("csr is null");
  if (!present(ufeat) || !out || !out->data || out->ndim < 1 || !out->shape) return fail("ufeat / out is null");
  int64_t f = 1, fo = 1;
  for (int d = 1; d < ufeat->ndim; ++d) f *= ufeat->shape[d];
  for (int d = 1; d < out->ndim; ++d) fo *= out->shape[d];
  if (f != fo) return fail("ufeat and out have different feature shapes");
  shapes[0] = ufeat->shape[0]; shapes[1] = f;
  shapes[2] = out->shape[0];   shapes[3] = f;
  shapes[4] = csr->nnz;        shapes[5] = 1;
  views[0] = dgla_tensor{ufeat->data, 2, shapes};
  views[1] = dgla_tensor{out->data, 2, shapes + 2};
  views[2] = dgla_tensor{const_cast<void*>(mask), 2, shapes + 4};
  if (csr->nnz && !mask) return fail("mask is null");
  if (build_spmm_launch("mul", "sum", csr, dtype, &views[0], &views[2], &views[1], L)) return -1;
  const int bits = dtype == DGLA_F64 ? 64 : (dtype == DGLA_F32 ? 32 : 16);
  L->bcast = kBcRhsGroup;
  L->rhs_group = bits;
  L->rhs_len = (f + bits - 1) / bits;
  L->rhs_mask = true;
  return 0;
}

size_t dgla_spmm_csr_masked_workspace_bytes(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ufeat,
                                            const dgla_tensor* out) {
  SpmmLaunch L{};
  int64_t shapes[6];
  dgla_tensor views[3];
  static const char dummy = 0;
  if (build_masked_launch(csr, dtype, ufeat, &dummy, out, shapes, views, &L)) return 0;
  switch (dtype) {
    case DGLA_F32: return spmm_csr_workspace_f32(L);
    case DGLA_F64: return spmm_csr_workspace_f64(L);
    case DGLA_F16: return spmm_csr_workspace_f16(L);
    case DGLA_BF16: return spmm_csr_workspace_bf16(L);
  }
  return 0;
}

int dgla_spmm_csr_masked(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ufeat, const void* mask,
                         const dgla_tensor* out, void* workspace, size_t workspace_bytes, uint32_t flags,
                         void* hip_stream) {
  SpmmLaunch L{};
  int64_t shapes[6];
  dgla_tensor views[3];
  if (build_masked_launch(csr, dtype, ufeat, mask, out, shapes, views, &L)) return -1;
  if (flags & ~(DGLA_ACCUMULATE | DGLA_PLAN_VALID)) return fail("dgla_spmm_csr_masked: DGLA_ACCUMULATE | DGLA_PLAN_VALID only");
  L.accumulate = (flags & DGLA_ACCUMULATE) != 0;
  L.plan_valid = (flags & DGLA_PLAN_VALID) != 0;
  L.workspace = workspace;
  L.workspace_bytes = workspace_bytes;
  L.stream = static_cast<hipStream_t>(hip_stream);
  if (csr->num_rows == 0 || L.out_len == 0) return 0;
  const DeviceGuard dev(L.stream, L.out);
  switch (dtype) {
    case DGLA_F32: return launch_spmm_csr_f32(L);
    case DGLA_F64: return launch_spmm_csr_f64(L);
    case DGLA_F16: return launch_spmm_csr_f16(L);
    case DGLA_BF16: return launch_spmm_csr_bf16(L);
  }
  return fail("unsupported feature dtype");
}

// Shapes are taken from relation 0's operands; every relation must use the same feature
// shape (src/array/kernel.cc:194-199).  The first dimensions of ufeat0 / efeat0 are not
// comparable with the stacked matrix (each relation has its own node and edge counts), so
// they are aligned with it before the common checks.
static int build_stacked_launch(const char* op, const dgla_csr* csr, dgla_dtype dtype,
                                const dgla_tensor* ufeat0, const dgla_tensor* efeat0,
                                const dgla_tensor* out, SpmmLaunch* L, const char* reduce = "sum") {
  if (!csr) return fail("csr is null");
  dgla_csr c = *csr;
  const int opc = parse_op(op, false);
  if (opc < 0) return fail(std::string("Unsupported SpMM binary operator: ") + (op ? op : "(null)"));
  if (op_uses_lhs(opc) && present(ufeat0)) c.num_cols = ufeat0->shape[0];
  dgla_tensor e0{nullptr, 0, nullptr};
  std::vector<int64_t> eshape;
  if (op_uses_rhs(opc)) {
    if (!present(efeat0)) return fail("operator needs the edge feature (efeat)");
    eshape.assign(efeat0->shape, efeat0->shape + efeat0->ndim);
    eshape[0] = c.nnz;
    e0 = dgla_tensor{efeat0->data, efeat0->ndim, eshape.data()};
  }
  return build_spmm_launch(op, reduce, &c, dtype, ufeat0, &e0, out, L);
}

size_t dgla_spmm_csr_stacked_workspace_bytes(const char* op, const dgla_csr* csr,
                                             dgla_dtype dtype, const dgla_tensor* ufeat0,
                                             const dgla_tensor* efeat0, const dgla_tensor* out) {
  SpmmLaunch L{};
  if (build_stacked_launch(op, csr, dtype, ufeat0, efeat0, out, &L)) return 0;
  switch (dtype) {
    case DGLA_F32: return spmm_csr_workspace_f32(L);
    case DGLA_F64: return spmm_csr_workspace_f64(L);
    case DGLA_F16: return spmm_csr_workspace_f16(L);
    case DGLA_BF16: return spmm_csr_workspace_bf16(L);
  }
  return 0;
}

int dgla_spmm_csr_stacked(const char* op, const dgla_csr* csr, const void* rel, int num_rel,
                          dgla_dtype dtype, const dgla_tensor* ufeat0, const dgla_tensor* efeat0,
                          const void* const* ufeat_ptrs, const void* const* efeat_ptrs,
                          const dgla_tensor* out, void* workspace, size_t workspace_bytes,
                          uint32_t flags, void* hip_stream) {
  if (!rel) return fail("rel is null");
  if (num_rel < 1 || num_rel > 256) return fail("num_rel must be in [1, 256]");
  SpmmLaunch L{};
  if (build_stacked_launch(op, csr, dtype, ufeat0, efeat0, out, &L)) return -1;
  if (op_uses_lhs(L.op) && !ufeat_ptrs) return fail("ufeat_ptrs is null");
  if (op_uses_rhs(L.op) && !efeat_ptrs) return fail("efeat_ptrs is null");
  L.rel = rel;
  L.ufeat_tab = ufeat_ptrs;
  L.efeat_tab = efeat_ptrs;
  L.num_rel = num_rel;
  L.accumulate = (flags & DGLA_ACCUMULATE) != 0;
  L.plan_valid = (flags & DGLA_PLAN_VALID) != 0;
  L.workspace = workspace;
  L.workspace_bytes = workspace_bytes;
  L.stream = static_cast<hipStream_t>(hip_stream);
  if (csr->num_rows == 0 || L.out_len == 0) return 0;
  const DeviceGuard dev(L.stream, L.out);
  switch (dtype) {
    case DGLA_F32: return launch_spmm_csr_f32(L);
    case DGLA_F64: return launch_spmm_csr_f64(L);
    case DGLA_F16: return launch_spmm_csr_f16(L);
    case DGLA_BF16: return launch_spmm_csr_bf16(L);
  }
  return fail("unsupported feature dtype");
}

size_t dgla_spmm_csr_stacked_cmp_workspace_bytes(const char* op, const char* reduce, const dgla_csr* csr,
                                                 dgla_dtype dtype, const dgla_tensor* ufeat0,
                                                 const dgla_tensor* efeat0, const dgla_tensor* out) {
  SpmmLaunch L{};
  if (build_stacked_launch(op, csr, dtype, ufeat0, efeat0, out, &L, reduce)) return 0;
  switch (dtype) {
    case DGLA_F32: return spmm_csr_workspace_f32(L);
    case DGLA_F64: return spmm_csr_workspace_f64(L);
    case DGLA_F16: return spmm_csr_workspace_f16(L);
    case DGLA_BF16: return spmm_csr_workspace_bf16(L);
  }
  return 0;
}

int dgla_spmm_csr_stacked_cmp(const char* op, const char* reduce, const dgla_csr* csr, const void* rel,
                              int num_rel, const int32_t* src_ntype, const int32_t* etype,
                              dgla_dtype dtype, const dgla_tensor* ufeat0, const dgla_tensor* efeat0,
                              const void* const* ufeat_ptrs, const void* const* efeat_ptrs,
                              const dgla_tensor* out, void* arg_u, void* arg_e, void* arg_u_ntype,
                              void* arg_e_etype, void* workspace, size_t workspace_bytes,
                              uint32_t flags, void* hip_stream) {
  if (!rel) return fail("rel is null");
  if (num_rel < 1 || num_rel > 256) return fail("num_rel must be in [1, 256]");
  if (!src_ntype || !etype) return fail("src_ntype / etype tables are null");
  SpmmLaunch L{};
  if (build_stacked_launch(op, csr, dtype, ufeat0, efeat0, out, &L, reduce)) return -1;
  if (L.red == kSum) return fail("dgla_spmm_csr_stacked_cmp is for max / min (sum: dgla_spmm_csr_stacked)");
  const bool use_u = op_uses_lhs(L.op), use_e = op_uses_rhs(L.op);
  if (use_u && !ufeat_ptrs) return fail("ufeat_ptrs is null");
  if (use_e && !efeat_ptrs) return fail("efeat_ptrs is null");
  if (use_u && (!arg_u || !arg_u_ntype)) return fail("arg_u and arg_u_ntype are required for max/min");
  if (use_e && (!arg_e || !arg_e_etype)) return fail("arg_e and arg_e_etype are required for max/min");
  if (flags & DGLA_ACCUMULATE) return fail("max / min do not accumulate");
  // the kernels leave the winner's stacked position in one of the tracker arrays; the pass below
  // turns it into the four outputs (that array last)
  void* pos_buf = use_u ? arg_u_ntype : arg_e_etype;
  L.rel = rel;
  L.ufeat_tab = ufeat_ptrs;
  L.efeat_tab = efeat_ptrs;
  L.num_rel = num_rel;
  L.arg_u = nullptr;
  L.arg_e = pos_buf;
  L.arg_empty = -1;
  L.plan_valid = (flags & DGLA_PLAN_VALID) != 0;
  L.workspace = workspace;
  L.workspace_bytes = workspace_bytes;
  L.stream = static_cast<hipStream_t>(hip_stream);
  if (csr->num_rows == 0 || L.out_len == 0) return 0;
  const DeviceGuard dev(L.stream, L.out);
  int rc = -1;
  switch (dtype) {
    case DGLA_F32: rc = launch_spmm_csr_f32(L); break;
    case DGLA_F64: rc = launch_spmm_csr_f64(L); break;
    case DGLA_F16: rc = launch_spmm_csr_f16(L); break;
    case DGLA_BF16: rc = launch_spmm_csr_bf16(L); break;
  }
  if (rc) return rc;
  StackedTypes types;
  for (int k = 0; k < 256; ++k) {
    types.src_ntype[k] = k < num_rel ? src_ntype[k] : -1;
    types.etype[k] = k < num_rel ? etype[k] : -1;
  }
  const int64_t n = csr->num_rows * static_cast<int64_t>(L.out_len);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (csr->idtype_bits == 32)
    hipLaunchKernelGGL(stacked_args_kernel<int32_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, L.stream,
                       static_cast<int32_t*>(pos_buf), static_cast<const int32_t*>(csr->indices),
                       static_cast<const int32_t*>(csr->data), static_cast<const uint8_t*>(rel),
                       static_cast<int32_t*>(use_u ? arg_u : nullptr), static_cast<int32_t*>(use_e ? arg_e : nullptr),
                       static_cast<int32_t*>(use_u ? arg_u_ntype : nullptr),
                       static_cast<int32_t*>(use_e ? arg_e_etype : nullptr), types, n);
  else
    hipLaunchKernelGGL(stacked_args_kernel<int64_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, L.stream,
                       static_cast<int64_t*>(pos_buf), static_cast<const int64_t*>(csr->indices),
                       static_cast<const int64_t*>(csr->data), static_cast<const uint8_t*>(rel),
                       static_cast<int64_t*>(use_u ? arg_u : nullptr), static_cast<int64_t*>(use_e ? arg_e : nullptr),
                       static_cast<int64_t*>(use_u ? arg_u_ntype : nullptr),
                       static_cast<int64_t*>(use_e ? arg_e_etype : nullptr), types, n);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int dgla_spmm_coo(const char* op_s, const char* red_s, const dgla_coo* coo, dgla_dtype dtype,
                  const dgla_tensor* ufeat, const dgla_tensor* efeat, const dgla_tensor* out,
                  void* arg_u, void* arg_e, void* hip_stream) {
  const int op = parse_op(op_s, false);
  if (op < 0) return fail(std::string("Unsupported SpMM binary operator: ") + (op_s ? op_s : "(null)"));
  const int red = parse_reduce(red_s);
  if (red < 0) return fail(std::string("Unsupported SpMM reducer: ") + (red_s ? red_s : "(null)"));
  CooView v;
  if (fill_coo(coo, &v)) return -1;
  BcastInfo bc;
  if (prepare_spmm(op, coo->num_rows, coo->num_cols, coo->nnz, ufeat, efeat, out, &bc)) return -1;
  if (red != kSum) {
    if (op_uses_lhs(op) && !arg_u) return fail("arg_u is required for max/min");
    if (op_uses_rhs(op) && !arg_e) return fail("arg_e is required for max/min");
  }
  if (coo->num_cols == 0 || bc.out_len == 0) return 0;
  const DeviceGuard dev(static_cast<hipStream_t>(hip_stream), out->data);
  if (red == kSum && std::getenv("USE_DETERMINISTIC_ALG") != nullptr && coo->nnz > 0) {
    // The reference's USE_DETERMINISTIC_ALG (src/array/cuda/spmm.cu:33-35) asks for run-to-run
    // identical sums.  The CSR kernels always are; the COO kernel adds with float atomics, whose
    // order is not fixed.  Under the flag a COO sum therefore goes through the deterministic
    // route: compress the COO by destination (stable sort, csrc/coo2csr.hip) in stream-ordered
    // scratch and run the merge-path CSR kernel over it — edges of a row are added in COO order.
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t ib = coo->idtype_bits / 8;
    const size_t a_ptr = (ib * (coo->num_cols + 1) + 255) / 256 * 256, a_e = (ib * coo->nnz + 255) / 256 * 256;
    char* scratch = nullptr;
    DGLA_CHECK_HIP(hipMallocAsync(reinterpret_cast<void**>(&scratch), a_ptr + 2 * a_e, s));
    int rc = dgla_coo_to_csr(coo->idtype_bits, coo->num_cols, coo->nnz, coo->col, coo->row, coo->data,
                             scratch, scratch + a_ptr, scratch + a_ptr + a_e, nullptr, 0, hip_stream);
    void* ws = nullptr;
    if (rc == 0) {
      dgla_csr csc{coo->num_cols, coo->num_rows, coo->nnz, coo->idtype_bits, scratch, scratch + a_ptr,
                   scratch + a_ptr + a_e};
      const size_t need = dgla_spmm_csr_workspace_bytes(op_s, red_s, &csc, dtype, ufeat, efeat, out);
      if (need) {
        const hipError_t e = hipMallocAsync(&ws, need, s);
        if (e != hipSuccess) rc = fail(std::string("hipMallocAsync: ") + hipGetErrorString(e));
      }
      if (rc == 0) rc = dgla_spmm_csr(op_s, red_s, &csc, dtype, ufeat, efeat, out, nullptr, nullptr, ws, need, 0, hip_stream);
    }
    if (ws) (void)hipFreeAsync(ws, s);
    (void)hipFreeAsync(scratch, s);
    return rc;
  }
  return launch_spmm_coo(v, op, red, dtype, op_uses_lhs(op) ? ufeat->data : nullptr,
                         op_uses_rhs(op) ? efeat->data : nullptr, out->data, arg_u, arg_e,
                         bc.out_len, bc.lhs_len, bc.rhs_len, bc.use_bcast, bc.dims,
                         static_cast<hipStream_t>(hip_stream));
}

static int sddmm_common(const char* op_s, bool use_coo, const dgla_csr* csr, const dgla_coo* coo,
                        dgla_dtype dtype, const dgla_tensor* lhs, const dgla_tensor* rhs,
                        const dgla_tensor* out, int lhs_target, int rhs_target, void* stream) {
  const int op = parse_op(op_s, true);
  if (op < 0) return fail(std::string("Unsupported SDDMM binary operator: ") + (op_s ? op_s : "(null)"));
  if (lhs_target < 0 || lhs_target > 2 || rhs_target < 0 || rhs_target > 2)
    return fail("targets must be 0 (u), 1 (e) or 2 (v)");
  SddmmLaunch L{};
  L.use_coo = use_coo;
  int64_t nsrc, ndst, nnz;
  if (use_coo) {
    if (fill_coo(coo, &L.coo)) return -1;
    nsrc = coo->num_rows, ndst = coo->num_cols, nnz = coo->nnz;
  } else {
    if (fill_csr(csr, &L.csr)) return -1;
    nsrc = csr->num_rows, ndst = csr->num_cols, nnz = csr->nnz;
  }
  const bool ul = op_uses_lhs(op), ur = op_uses_rhs(op);
  if (ul && !present(lhs)) return fail("operator needs lhs");
  if (ur && !present(rhs)) return fail("operator needs rhs");
  if (!present(out) && nnz > 0) return fail("out is null");
  const int64_t dim0[3] = {nsrc, nnz, ndst};
  if (ul && check_tensor(lhs, "lhs", dim0[lhs_target])) return -1;
  if (ur && check_tensor(rhs, "rhs", dim0[rhs_target])) return -1;
  if (nnz == 0) return 0;  // python/dgl/_sparse_ops.py:553
  if (check_tensor(out, "out", nnz)) return -1;
  BcastInfo bc;
  const std::vector<int64_t> l = ul ? feat_shape(lhs) : feat_shape(rhs);
  const std::vector<int64_t> r = ur ? feat_shape(rhs) : feat_shape(lhs);
  if (analyse_bcast(op, l, r, &bc)) return -1;
  if (feat_len(out) != bc.out_len)
    return fail("out has " + std::to_string(feat_len(out)) + " features per edge, expected " +
                std::to_string(bc.out_len));
  L.op = op;
  L.dtype = dtype;
  L.lhs = ul ? lhs->data : nullptr;
  L.rhs = ur ? rhs->data : nullptr;
  L.out = out->data;
  L.lhs_target = lhs_target;
  L.rhs_target = rhs_target;
  L.out_len = bc.out_len;
  L.lhs_len = bc.lhs_len;
  L.rhs_len = bc.rhs_len;
  L.reduce_size = bc.reduce_size;
  L.bcast = bc.use_bcast ? kBcGeneral : kBcNone;
  L.bdims = bc.dims;
  L.stream = static_cast<hipStream_t>(stream);
  if (bc.out_len == 0) return 0;
  const DeviceGuard dev(L.stream, L.out);
  switch (dtype) {
    case DGLA_F32: return launch_sddmm_f32(L);
    case DGLA_F64: return launch_sddmm_f64(L);
    case DGLA_F16: return launch_sddmm_f16(L);
    case DGLA_BF16: return launch_sddmm_bf16(L);
  }
  return fail("unsupported feature dtype");
}

int dgla_sddmm_coo(const char* op, const dgla_coo* coo, dgla_dtype dtype, const dgla_tensor* lhs,
                   const dgla_tensor* rhs, const dgla_tensor* out, int lhs_target, int rhs_target,
                   void* hip_stream) {
  return sddmm_common(op, true, nullptr, coo, dtype, lhs, rhs, out, lhs_target, rhs_target,
                      hip_stream);
}

int dgla_sddmm_csr(const char* op, const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* lhs,
                   const dgla_tensor* rhs, const dgla_tensor* out, int lhs_target, int rhs_target,
                   void* hip_stream) {
  return sddmm_common(op, false, csr, nullptr, dtype, lhs, rhs, out, lhs_target, rhs_target,
                      hip_stream);
}

size_t dgla_edge_softmax_workspace_bytes(const dgla_csr* csr, dgla_dtype dtype, int64_t dim) {
  if (!csr) return 0;
  return edge_softmax_workspace_bytes(csr->num_rows, csr->nnz, dtype, dim);
}

int dgla_edge_softmax_forward(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* score,
                              const dgla_tensor* out, void* workspace, size_t workspace_bytes,
                              uint32_t flags, void* hip_stream) {
  CsrView v;
  if (fill_csr(csr, &v)) return -1;
  if (!present(score) || !present(out)) return csr->nnz == 0 ? 0 : fail("score / out is null");
  if (check_tensor(score, "score", csr->nnz) || check_tensor(out, "out", csr->nnz)) return -1;
  if (feat_len(score) != feat_len(out)) return fail("score and out shapes differ");
  if (csr->nnz == 0 || feat_len(score) == 0) return 0;
  const DeviceGuard dev(static_cast<hipStream_t>(hip_stream), out->data);
  return launch_edge_softmax(v, dtype, score->data, nullptr, out->data, feat_len(score), false,
                             workspace, workspace_bytes, (flags & DGLA_PLAN_VALID) != 0,
                             static_cast<hipStream_t>(hip_stream), (flags & DGLA_ESM_OUT_POSITION) != 0);
}

int dgla_edge_softmax_backward(const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* out,
                               const dgla_tensor* sds, const dgla_tensor* back, void* workspace,
                               size_t workspace_bytes, uint32_t flags, void* hip_stream) {
  CsrView v;
  if (fill_csr(csr, &v)) return -1;
  if (!present(out) || !present(sds) || !present(back))
    return csr->nnz == 0 ? 0 : fail("out / sds / back is null");
  if (check_tensor(out, "out", csr->nnz) || check_tensor(sds, "sds", csr->nnz) ||
      check_tensor(back, "back", csr->nnz))
    return -1;
  if (feat_len(out) != feat_len(sds) || feat_len(out) != feat_len(back))
    return fail("out, sds and back shapes differ");
  if (csr->nnz == 0 || feat_len(out) == 0) return 0;
  const DeviceGuard dev(static_cast<hipStream_t>(hip_stream), back->data);
  return launch_edge_softmax(v, dtype, out->data, sds->data, back->data, feat_len(out), true,
                             workspace, workspace_bytes, (flags & DGLA_PLAN_VALID) != 0,
                             static_cast<hipStream_t>(hip_stream), (flags & DGLA_ESM_OUT_POSITION) != 0,
                             (flags & DGLA_ESM_B_IS_GRAD) != 0);
}

int dgla_set_tuning(uint32_t flags) {
  if (flags & ~kTuneKnown) {
    last_error() = "dgla_set_tuning: unknown or retired tuning bit in " + std::to_string(flags);
    return -1;
  }
  tuning_flags() = flags;
  return 0;
}

uint32_t dgla_get_tuning(void) { return tuning_flags(); }

int dgla_spmm_set_profile_events(void* before, void* after) {
  profile_events().before = static_cast<hipEvent_t>(before);
  profile_events().after = static_cast<hipEvent_t>(after);
  return 0;
}

int dgla_stream_copy(void* dst, const void* src, size_t bytes, void* hip_stream) {
  const DeviceGuard dev(static_cast<hipStream_t>(hip_stream), dst);
  return launch_stream_copy(dst, src, bytes, 1 << 2, static_cast<hipStream_t>(hip_stream));
}

int dgla_stream_copy_variant(void* dst, const void* src, size_t bytes, int variant,
                             void* hip_stream) {
  const DeviceGuard dev(static_cast<hipStream_t>(hip_stream), dst);
  return launch_stream_copy(dst, src, bytes, variant, static_cast<hipStream_t>(hip_stream));
}

}  // extern "C"
