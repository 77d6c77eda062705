// oracle/_ref driver — TEST INFRASTRUCTURE ONLY (never linked into or called by dgl_amd).
//
// Plain-pointer C entry points around the REFERENCE's own CPU g-SpMM / g-SDDMM code, which
// oracle/Makefile compiles from the files where they lie under /root/reference:
//     src/array/cpu/spmm.cc  (+ spmm.h, spmm_binary_ops.h)   SpMMCsr / SpMMCoo / Edge_softmax_*
//     src/array/cpu/sddmm.cc (+ sddmm.h, ../selector.h)      SDDMMCsr / SDDMMCoo
//     src/bcast.cc                                           CalcBcastOff
//     src/array/cpu/segment_reduce.cc (+ segment_reduce.h)   SegmentReduce / ScatterAdd / BackwardSegmentCmp
//     src/array/cpu/spmat_op_impl_coo.cc                     COOToCSR (tsl::robin_map shimmed, ref_shim/tsl/)
// Nothing of the reference is copied into this repository.  The empty third_party/dmlc-core
// submodule is replaced by the from-scratch headers in oracle/ref_shim/dmlc/.  libxsmm is
// absent (empty submodule), so USE_LIBXSMM is undefined and the reference takes its naive
// path (spmm.h:144-160) — the same arithmetic in plain CSR-position order.
//
// What this file adds of its own: (1) NDArray views over caller memory (an
// NDArray::Container whose dl_tensor points at the caller's buffer; the public struct is
// declared in include/dgl/runtime/ndarray.h:410-480), (2) the two out-of-line NDArray
// members the kernels call, NumElements() / GetSize(), and NDArray::Empty (used by
// aten::NullArray and the hetero accumulators) as a plain host allocation — their home,
// src/runtime/ndarray.cc, drags in the device API and tensor adapter and is not compiled —
// and (3) the
// idtype x dtype switch that src/array/kernel.cc:20-44,224-248 performs with ATEN_* macros.
#include <dgl/array.h>
#include <dgl/bcast.h>

#include <algorithm>
#include <cstdint>
#include <ostream>
#include <cstring>
#include <string>
#include <vector>

// Declarations of the reference templates (src/array/kernel_decl.h:23-87 declares the same;
// the definitions and explicit instantiations live in the .cc files listed above).
namespace dgl {
namespace aten {
template <int XPU, typename IdType, typename DType>
void SpMMCsr(const std::string& op, const std::string& reduce, const BcastOff& bcast,
             const CSRMatrix& csr, NDArray ufeat, NDArray efeat, NDArray out,
             std::vector<NDArray> out_aux);
template <int XPU, typename IdType, typename DType>
void SpMMCoo(const std::string& op, const std::string& reduce, const BcastOff& bcast,
             const COOMatrix& coo, NDArray ufeat, NDArray efeat, NDArray out,
             std::vector<NDArray> out_aux);
template <int XPU, typename IdType, typename DType>
void SDDMMCsr(const std::string& op, const BcastOff& bcast, const CSRMatrix& csr, NDArray lhs,
              NDArray rhs, NDArray out, int lhs_target, int rhs_target);
template <int XPU, typename IdType, typename DType>
void SDDMMCoo(const std::string& op, const BcastOff& bcast, const COOMatrix& coo, NDArray lhs,
              NDArray rhs, NDArray out, int lhs_target, int rhs_target);
template <int XPU, typename IdType, typename DType>
void Edge_softmax_csr_forward(const std::string& op, const BcastOff& bcast,
                              const CSRMatrix& csr, NDArray ufeat, NDArray efeat, NDArray out);
template <int XPU, typename IdType, typename DType>
void Edge_softmax_csr_backward(const std::string& op, const BcastOff& bcast,
                               const CSRMatrix& csr, NDArray out, NDArray sds,
                               NDArray back_out);
// src/array/cpu/spmm.cc:45-150 (declared in src/array/kernel_decl.h:37-47)
template <int XPU, typename IdType, typename DType>
void SpMMCsrHetero(const std::string& op, const std::string& reduce, const BcastOff& bcast,
                   const std::vector<CSRMatrix>& csr, const std::vector<NDArray>& ufeat,
                   const std::vector<NDArray>& efeat, std::vector<NDArray>* out,
                   std::vector<std::vector<NDArray>>* out_aux,
                   const std::vector<dgl_type_t>& ufeat_node_tids,
                   const std::vector<dgl_type_t>& out_node_tids);
// src/array/cpu/segment_reduce.cc:18-57 (declared in src/array/kernel_decl.h)
template <int XPU, typename IdType, typename DType>
void SegmentReduce(const std::string& op, NDArray feat, NDArray offsets, NDArray out, NDArray arg);
template <int XPU, typename IdType, typename DType>
void ScatterAdd(NDArray feat, NDArray idx, NDArray out);
template <int XPU, typename IdType, typename DType>
void BackwardSegmentCmp(NDArray feat, NDArray arg, NDArray out);
namespace impl {
// src/array/cpu/spmat_op_impl_coo.cc:747-764 (declared in src/array/array_op.h)
template <DGLDeviceType XPU, typename IdType>
CSRMatrix COOToCSR(COOMatrix coo);
}  // namespace impl
}  // namespace aten

namespace runtime {
// src/runtime/ndarray.cc:119-128 (not compiled, see header comment).
int64_t NDArray::NumElements() const {
  if (data_->dl_tensor.ndim == 0) return 0;
  int64_t n = 1;
  for (int i = 0; i < data_->dl_tensor.ndim; ++i) n *= data_->dl_tensor.shape[i];
  return n;
}
size_t NDArray::GetSize() const {
  const DGLArray& t = data_->dl_tensor;
  return static_cast<size_t>(NumElements()) * ((t.dtype.bits * t.dtype.lanes + 7) / 8);
}
}  // namespace runtime
}  // namespace dgl

namespace {
using dgl::runtime::NDArray;

struct View : NDArray::Container {
  int64_t shape_store[8];
  void* owned = nullptr;
};
void view_deleter(NDArray::Container* c) {
  View* v = static_cast<View*>(c);
  free(v->owned);
  delete v;
}
}  // namespace

namespace dgl {
namespace runtime {
// Host-only stand-in for src/runtime/ndarray.cc:206-225 (device API not compiled).
NDArray NDArray::Empty(std::vector<int64_t> shape, DGLDataType dtype, DGLContext ctx) {
  View* v = new View();
  int64_t n = 1;
  for (size_t i = 0; i < shape.size(); ++i) {
    v->shape_store[i] = shape[i];
    n *= shape[i];
  }
  const size_t bytes = static_cast<size_t>(n) * ((dtype.bits * dtype.lanes + 7) / 8);
  v->owned = bytes ? aligned_alloc(64, (bytes + 63) / 64 * 64) : nullptr;
  v->dl_tensor.data = v->owned;
  v->dl_tensor.ctx = ctx;
  v->dl_tensor.ndim = static_cast<int>(shape.size());
  v->dl_tensor.dtype = dtype;
  v->dl_tensor.shape = v->shape_store;
  v->dl_tensor.strides = nullptr;
  v->dl_tensor.byte_offset = 0;
  v->deleter = view_deleter;
  return NDArray(v);
}
// Host-only stand-in for src/runtime/ndarray.cc:286-294 (device API not compiled).
template <typename T>
NDArray NDArray::FromVector(const std::vector<T>& vec, DGLContext ctx) {
  DGLDataType dt;
  dt.code = 0;
  dt.bits = sizeof(T) * 8;
  dt.lanes = 1;
  NDArray ret = NDArray::Empty({static_cast<int64_t>(vec.size())}, dt, ctx);
  if (!vec.empty()) std::memcpy(ret->data, vec.data(), vec.size() * sizeof(T));
  return ret;
}
template NDArray NDArray::FromVector<int32_t>(const std::vector<int32_t>&, DGLContext);
template NDArray NDArray::FromVector<int64_t>(const std::vector<int64_t>&, DGLContext);
}  // namespace runtime

namespace aten {
// Host-only stand-in for aten::Full (src/array/array.cc:54-66 -> cpu impl: allocate + fill).
IdArray Full(int64_t val, int64_t length, uint8_t nbits, DGLContext ctx) {
  DGLDataType dt;
  dt.code = 0;
  dt.bits = nbits;
  dt.lanes = 1;
  IdArray ret = runtime::NDArray::Empty({length}, dt, ctx);
  if (nbits == 32)
    std::fill_n(static_cast<int32_t*>(ret->data), length, static_cast<int32_t>(val));
  else
    std::fill_n(static_cast<int64_t*>(ret->data), length, val);
  return ret;
}
}  // namespace aten
}  // namespace dgl

// only reached on a failing CHECK in the reference (include/dgl/aten/array_ops.h declares it)
std::ostream& operator<<(std::ostream& os, dgl::runtime::NDArray) { return os << "<NDArray>"; }

namespace {

// code: 0 = int, 2 = float, 4 = bfloat (c_runtime_api.h DGLDataTypeCode)
NDArray make_view(void* data, int ndim, const int64_t* shape, uint8_t code, uint8_t bits) {
  View* v = new View();
  v->dl_tensor.data = data;
  v->dl_tensor.ctx.device_type = kDGLCPU;
  v->dl_tensor.ctx.device_id = 0;
  v->dl_tensor.ndim = ndim;
  v->dl_tensor.dtype.code = code;
  v->dl_tensor.dtype.bits = bits;
  v->dl_tensor.dtype.lanes = 1;
  for (int i = 0; i < ndim; ++i) v->shape_store[i] = shape[i];
  v->dl_tensor.shape = v->shape_store;
  v->dl_tensor.strides = nullptr;
  v->dl_tensor.byte_offset = 0;
  v->deleter = view_deleter;
  return NDArray(v);
}

// "absent operand" = empty int64 array, IsNullArray <=> shape[0] == 0
// (include/dgl/aten/array_ops.h:28-37, python/dgl/ndarray.py:309-312)
NDArray null_array() {
  const int64_t z = 0;
  return make_view(nullptr, 1, &z, 0, 64);
}

NDArray id_view(const void* p, int64_t n, int idbits) {
  if (p == nullptr) return null_array();
  return make_view(const_cast<void*>(p), 1, &n, 0, static_cast<uint8_t>(idbits));
}

struct Feat {
  void* data;
  int32_t ndim;
  const int64_t* shape;
};

// dtype: 0 f32, 1 f64, 3 bf16 (codes of include/dgl_amd.h; fp16 has no CPU kernels in the
// reference: ATEN_FLOAT_TYPE_SWITCH_16BITS on CPU offers bf16 only, aten/macro.h:137-166)
NDArray feat_view(const Feat* f, int dtype) {
  if (f == nullptr || f->data == nullptr) return null_array();
  if (dtype == 0) return make_view(f->data, f->ndim, f->shape, 2, 32);
  if (dtype == 1) return make_view(f->data, f->ndim, f->shape, 2, 64);
  return make_view(f->data, f->ndim, f->shape, 4, 16);
}

thread_local std::string g_err;

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

#define REF_TYPE_SWITCH(idbits, dtype, IdType, DType, ...)          \
  do {                                                              \
    if ((idbits) == 32) {                                           \
      typedef int32_t IdType;                                       \
      REF_DTYPE_SWITCH(dtype, DType, __VA_ARGS__);                  \
    } else {                                                        \
      typedef int64_t IdType;                                       \
      REF_DTYPE_SWITCH(dtype, DType, __VA_ARGS__);                  \
    }                                                               \
  } while (0)
#define REF_DTYPE_SWITCH(dtype, DType, ...)   \
  do {                                        \
    if ((dtype) == 0) {                       \
      typedef float DType;                    \
      { __VA_ARGS__ }                         \
    } else if ((dtype) == 1) {                \
      typedef double DType;                   \
      { __VA_ARGS__ }                         \
    } else {                                  \
      typedef BFloat16 DType;                 \
      { __VA_ARGS__ }                         \
    }                                         \
  } while (0)

dgl::aten::CSRMatrix make_csr(int64_t num_rows, int64_t num_cols, int64_t nnz, int idbits,
                              const void* indptr, const void* indices, const void* eids) {
  dgl::aten::CSRMatrix csr;
  csr.num_rows = num_rows;
  csr.num_cols = num_cols;
  csr.indptr = id_view(indptr, num_rows + 1, idbits);
  csr.indices = id_view(indices, nnz, idbits);
  // an empty relation: libdgl's zero-length arrays still carry a non-null data pointer, which the
  // kernels CHECK (spmm.h:135) before looping zero times
  static int64_t empty_storage[2] = {0, 0};
  if (nnz == 0) csr.indices = make_view(empty_storage, 1, &nnz, 0, static_cast<uint8_t>(idbits));
  csr.data = id_view(eids, nnz, idbits);
  return csr;
}

dgl::aten::COOMatrix make_coo(int64_t num_rows, int64_t num_cols, int64_t nnz, int idbits,
                              const void* row, const void* col, const void* eids) {
  dgl::aten::COOMatrix coo;
  coo.num_rows = num_rows;
  coo.num_cols = num_cols;
  coo.row = make_view(const_cast<void*>(row), 1, &nnz, 0, static_cast<uint8_t>(idbits));
  coo.col = make_view(const_cast<void*>(col), 1, &nnz, 0, static_cast<uint8_t>(idbits));
  coo.data = id_view(eids, nnz, idbits);
  return coo;
}
}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// BcastOff of (op, lhs, rhs) as the reference computes it; offsets may be NULL.
// Returns use_bcast; fills lens[4] = {lhs_len, rhs_len, out_len, reduce_size}.
int ref_calc_bcast_off(const char* op, const Feat* lhs, const Feat* rhs, int64_t* lens,
                       int64_t* lhs_off, int64_t* rhs_off, int64_t off_cap) {
  int use = -1;
  guarded([&] {
    dgl::BcastOff b = dgl::CalcBcastOff(op, feat_view(lhs, 0), feat_view(rhs, 0));
    lens[0] = b.lhs_len;
    lens[1] = b.rhs_len;
    lens[2] = b.out_len;
    lens[3] = b.reduce_size;
    if (b.use_bcast && lhs_off && rhs_off) {
      for (size_t i = 0; i < b.lhs_offset.size() && static_cast<int64_t>(i) < off_cap; ++i) {
        lhs_off[i] = b.lhs_offset[i];
        rhs_off[i] = b.rhs_offset[i];
      }
    }
    use = b.use_bcast ? 1 : 0;
  });
  return use;
}

// aten::CSRSpMM semantics (src/array/kernel.cc:20-44 -> SpMMCsr<kDGLCPU>).  `out` must be
// pre-zeroed by the caller exactly as python/dgl/_sparse_ops.py:227 does.
int ref_spmm_csr(const char* op, const char* reduce, int idbits, int dtype, int64_t num_rows,
                 int64_t num_cols, int64_t nnz, const void* indptr, const void* indices,
                 const void* eids, const Feat* ufeat, const Feat* efeat, const Feat* out,
                 void* arg_u, void* arg_e) {
  return guarded([&] {
    NDArray U = feat_view(ufeat, dtype), E = feat_view(efeat, dtype), O = feat_view(out, dtype);
    const dgl::BcastOff bcast = dgl::CalcBcastOff(op, U, E);
    const auto csr = make_csr(num_rows, num_cols, nnz, idbits, indptr, indices, eids);
    Feat au{arg_u, out->ndim, out->shape}, ae{arg_e, out->ndim, out->shape};
    std::vector<NDArray> aux = {
        arg_u ? make_view(arg_u, out->ndim, out->shape, 0, static_cast<uint8_t>(idbits)) : null_array(),
        arg_e ? make_view(arg_e, out->ndim, out->shape, 0, static_cast<uint8_t>(idbits)) : null_array()};
    (void)au;
    (void)ae;
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::SpMMCsr<kDGLCPU, IdType, DType>(op, reduce, bcast, csr, U, E, O, aux);
    });
  });
}

int ref_spmm_coo(const char* op, const char* reduce, int idbits, int dtype, int64_t num_src,
                 int64_t num_dst, int64_t nnz, const void* row, const void* col,
                 const void* eids, const Feat* ufeat, const Feat* efeat, const Feat* out,
                 void* arg_u, void* arg_e) {
  return guarded([&] {
    NDArray U = feat_view(ufeat, dtype), E = feat_view(efeat, dtype), O = feat_view(out, dtype);
    const dgl::BcastOff bcast = dgl::CalcBcastOff(op, U, E);
    const auto coo = make_coo(num_src, num_dst, nnz, idbits, row, col, eids);
    std::vector<NDArray> aux = {
        arg_u ? make_view(arg_u, out->ndim, out->shape, 0, static_cast<uint8_t>(idbits)) : null_array(),
        arg_e ? make_view(arg_e, out->ndim, out->shape, 0, static_cast<uint8_t>(idbits)) : null_array()};
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::SpMMCoo<kDGLCPU, IdType, DType>(op, reduce, bcast, coo, U, E, O, aux);
    });
  });
}

// aten::CSRSDDMM / COOSDDMM semantics (src/array/kernel.cc:224-248).
int ref_sddmm_csr(const char* op, int idbits, int dtype, int64_t num_rows, int64_t num_cols,
                  int64_t nnz, const void* indptr, const void* indices, const void* eids,
                  const Feat* lhs, const Feat* rhs, const Feat* out, int lhs_target,
                  int rhs_target) {
  return guarded([&] {
    NDArray L = feat_view(lhs, dtype), R = feat_view(rhs, dtype), O = feat_view(out, dtype);
    const dgl::BcastOff bcast = dgl::CalcBcastOff(op, L, R);
    const auto csr = make_csr(num_rows, num_cols, nnz, idbits, indptr, indices, eids);
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::SDDMMCsr<kDGLCPU, IdType, DType>(op, bcast, csr, L, R, O, lhs_target, rhs_target);
    });
  });
}

int ref_sddmm_coo(const char* op, int idbits, int dtype, int64_t num_src, int64_t num_dst,
                  int64_t nnz, const void* row, const void* col, const void* eids,
                  const Feat* lhs, const Feat* rhs, const Feat* out, int lhs_target,
                  int rhs_target) {
  return guarded([&] {
    NDArray L = feat_view(lhs, dtype), R = feat_view(rhs, dtype), O = feat_view(out, dtype);
    const dgl::BcastOff bcast = dgl::CalcBcastOff(op, L, R);
    const auto coo = make_coo(num_src, num_dst, nnz, idbits, row, col, eids);
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::SDDMMCoo<kDGLCPU, IdType, DType>(op, bcast, coo, L, R, O, lhs_target, rhs_target);
    });
  });
}

// _CAPI_DGLKernelEdge_softmax_forward / _backward (src/array/kernel.cc:312-346,542-561):
// op is always "copy_rhs"; ufeat is the null array.
int ref_edge_softmax_forward(int idbits, int dtype, int64_t num_rows, int64_t num_cols,
                             int64_t nnz, const void* indptr, const void* indices,
                             const void* eids, const Feat* score, const Feat* out) {
  return guarded([&] {
    NDArray U = null_array(), E = feat_view(score, dtype), O = feat_view(out, dtype);
    const dgl::BcastOff bcast = dgl::CalcBcastOff("copy_rhs", U, E);
    const auto csr = make_csr(num_rows, num_cols, nnz, idbits, indptr, indices, eids);
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::Edge_softmax_csr_forward<kDGLCPU, IdType, DType>("copy_rhs", bcast, csr, U, E, O);
    });
  });
}

int ref_edge_softmax_backward(int idbits, int dtype, int64_t num_rows, int64_t num_cols,
                              int64_t nnz, const void* indptr, const void* indices,
                              const void* eids, const Feat* out, const Feat* sds,
                              const Feat* back) {
  return guarded([&] {
    NDArray O = feat_view(out, dtype), S = feat_view(sds, dtype), B = feat_view(back, dtype);
    const dgl::BcastOff bcast = dgl::CalcBcastOff("copy_rhs", O, S);
    const auto csr = make_csr(num_rows, num_cols, nnz, idbits, indptr, indices, eids);
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::Edge_softmax_csr_backward<kDGLCPU, IdType, DType>("copy_rhs", bcast, csr, O, S, B);
    });
  });
}

// aten::SpMMHetero semantics for the CSC-capable case (src/array/kernel.cc:173-221 ->
// SpMMCsrHetero<kDGLCPU>, spmm.cc:45-150).  Per relation: its in-edge CSR and (src, dst) node
// type; per node type: ufeat, out and the four aux arrays (NULL entries allowed where the
// operator / reducer does not use them); per relation: efeat.  `out` arrives zero-filled.
int ref_spmm_csr_hetero(const char* op, const char* reduce, int idbits, int dtype, int num_etypes,
                        int num_ntypes, const int64_t* num_rows, const int64_t* num_cols,
                        const int64_t* nnz, const void* const* indptr, const void* const* indices,
                        const void* const* eids, const int32_t* src_ntype, const int32_t* dst_ntype,
                        const Feat* ufeat, const Feat* efeat, const Feat* out, void* const* arg_u,
                        void* const* arg_e, void* const* arg_u_ntype, void* const* arg_e_etype) {
  return guarded([&] {
    std::vector<dgl::aten::CSRMatrix> csrs;
    std::vector<dgl::dgl_type_t> u_tids, o_tids;
    for (int et = 0; et < num_etypes; ++et) {
      csrs.push_back(make_csr(num_rows[et], num_cols[et], nnz[et], idbits, indptr[et], indices[et], eids[et]));
      u_tids.push_back(src_ntype[et]);
      o_tids.push_back(dst_ntype[et]);
    }
    const bool use_u = std::string(op) != "copy_rhs", use_e = std::string(op) != "copy_lhs";
    std::vector<NDArray> U, E, O;
    std::vector<std::vector<NDArray>> aux(4);
    for (int nt = 0; nt < num_ntypes; ++nt) {
      if (use_u) U.push_back(feat_view(&ufeat[nt], dtype));
      O.push_back(feat_view(&out[nt], dtype));
      void* const* src[4] = {arg_u, arg_e, arg_u_ntype, arg_e_etype};
      for (int a = 0; a < 4; ++a)
        aux[a].push_back(src[a] && src[a][nt]
                             ? make_view(src[a][nt], out[nt].ndim, out[nt].shape, 0, static_cast<uint8_t>(idbits))
                             : null_array());
    }
    if (use_e)
      for (int et = 0; et < num_etypes; ++et) E.push_back(feat_view(&efeat[et], dtype));
    // the broadcast is taken from relation 0's operands, as kernel.cc:181-192 does
    NDArray u0 = use_u ? U[u_tids[0]] : null_array(), e0 = use_e ? E[0] : null_array();
    const dgl::BcastOff bcast = dgl::CalcBcastOff(op, u0, e0);
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::SpMMCsrHetero<kDGLCPU, IdType, DType>(op, reduce, bcast, csrs, U, E, &O, &aux, u_tids, o_tids);
    });
  });
}

// SegmentReduce<kDGLCPU> (src/array/cpu/segment_reduce.cc:18-35).  For "sum" `out` must be
// pre-zeroed by the caller (python/dgl/_sparse_ops.py:665 allocates F.zeros); max/min fill it.
int ref_segment_reduce(const char* op, int idbits, int dtype, int64_t num_segments,
                       const void* offsets, const Feat* feat, const Feat* out, void* arg) {
  return guarded([&] {
    NDArray F = feat_view(feat, dtype), O = feat_view(out, dtype);
    NDArray Off = id_view(offsets, num_segments + 1, idbits);
    NDArray A = arg ? make_view(arg, out->ndim, out->shape, 0, static_cast<uint8_t>(idbits))
                    : null_array();
    if (dtype != 0 && dtype != 1) throw std::runtime_error("segment reduce: f32 / f64 only");
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::SegmentReduce<kDGLCPU, IdType, DType>(op, F, Off, O, A);
    });
  });
}

int ref_scatter_add(int idbits, int dtype, int64_t n, const void* idx, const Feat* feat,
                    const Feat* out) {
  return guarded([&] {
    NDArray F = feat_view(feat, dtype), O = feat_view(out, dtype);
    NDArray I = id_view(idx, n, idbits);
    if (dtype != 0 && dtype != 1) throw std::runtime_error("scatter add: f32 / f64 only");
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::ScatterAdd<kDGLCPU, IdType, DType>(F, I, O);
    });
  });
}

int ref_backward_segment_cmp(int idbits, int dtype, const Feat* feat, const void* arg,
                             const Feat* out) {
  return guarded([&] {
    NDArray F = feat_view(feat, dtype), O = feat_view(out, dtype);
    NDArray A = make_view(const_cast<void*>(arg), feat->ndim, feat->shape, 0,
                          static_cast<uint8_t>(idbits));
    if (dtype != 0 && dtype != 1) throw std::runtime_error("backward segment cmp: f32 / f64 only");
    REF_TYPE_SWITCH(idbits, dtype, IdType, DType, {
      dgl::aten::BackwardSegmentCmp<kDGLCPU, IdType, DType>(F, A, O);
    });
  });
}

// aten::impl::COOToCSR<kDGLCPU> (src/array/cpu/spmat_op_impl_coo.cc:747-764): chooses between
// its sorted / small / sparse / dense algorithms itself.  `data` may be NULL (edge id = position).
int ref_coo_to_csr(int idbits, int64_t num_rows, int64_t num_cols, int64_t nnz, const void* row,
                   const void* col, const void* data, void* indptr, void* indices, void* data_out) {
  return guarded([&] {
    dgl::aten::COOMatrix coo = make_coo(num_rows, num_cols, nnz, idbits, row, col, data);
    coo.row_sorted = coo.col_sorted = false;
    dgl::aten::CSRMatrix csr = idbits == 32 ? dgl::aten::impl::COOToCSR<kDGLCPU, int32_t>(coo)
                                            : dgl::aten::impl::COOToCSR<kDGLCPU, int64_t>(coo);
    const size_t ib = idbits / 8;
    std::memcpy(indptr, csr.indptr->data, ib * (num_rows + 1));
    if (nnz) std::memcpy(indices, csr.indices->data, ib * nnz);
    if (dgl::aten::IsNullArray(csr.data)) {  // "no data" = position
      for (int64_t i = 0; i < nnz; ++i) {
        if (idbits == 32) static_cast<int32_t*>(data_out)[i] = static_cast<int32_t>(i);
        else static_cast<int64_t*>(data_out)[i] = i;
      }
    } else if (nnz) {
      std::memcpy(data_out, csr.data->data, ib * nnz);
    }
  });
}

}  // extern "C"
