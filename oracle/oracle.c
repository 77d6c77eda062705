/*
 * oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C + OpenMP) of the reference's g-SpMM / g-SDDMM CPU kernels
 * (src/array/cpu/spmm.h, spmm.cc, sddmm.h, spmm_binary_ops.h, src/array/selector.h of
 * dmlc/dgl @ 2025-08-24).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (dgl_amd/) never does.
 *
 * PARITY PINNING: pinned to the reference itself.  oracle/Makefile compiles the reference's
 * own src/array/cpu/{spmm,sddmm}.cc and src/bcast.cc where they lie under /root/reference
 * (oracle/_ref/libdglref.so; the empty third_party/dmlc-core submodule is replaced by the
 * from-scratch headers in oracle/ref_shim/), and tests/test_oracle_vs_reference.py demands
 * bit equality between this file and that library on a 2242-case sweep (every op x reducer
 * x idtype x dtype x broadcast shape x format x target pair, degenerate graphs, edge
 * softmax fwd/bwd) plus the broadcast offset tables.  Outputs of the reference build are
 * committed as tests/golden/reference_cpu_outputs.npz (tests/golden/make_golden.py) so the
 * pin travels to machines without /root/reference.  Additionally checked against the
 * closed-form cases of the reference's tests/docstrings and scipy / torch / dense
 * (tests/test_oracle_known_answers.py).  NOT pinned: the reference's libxsmm JIT path
 * (third_party/libxsmm is an empty submodule; its K-blocked summation order differs from
 * the naive kernel's) and cuSPARSE's order on NVIDIA - both are only tolerance-tested by
 * the reference's own suite (rtol 1e-4, tests/python/common/ops/test_ops.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_DIV = 3, OP_COPY_LHS = 4, OP_COPY_RHS = 5, OP_DOT = 6 };

static inline int op_uses_lhs(int op) { return op != OP_COPY_RHS; }
static inline int op_uses_rhs(int op) { return op != OP_COPY_LHS; }

#define DTYPE float
#define EXPFN expf
#define IDTYPE int32_t
#define SFX f32_i32
#include "kernels_impl.inc"
#undef IDTYPE
#undef SFX
#define IDTYPE int64_t
#define SFX f32_i64
#include "kernels_impl.inc"
#undef IDTYPE
#undef SFX
#undef DTYPE
#undef EXPFN

#define DTYPE double
#define EXPFN exp
#define IDTYPE int32_t
#define SFX f64_i32
#include "kernels_impl.inc"
#undef IDTYPE
#undef SFX
#define IDTYPE int64_t
#define SFX f64_i64
#include "kernels_impl.inc"
#undef IDTYPE
#undef SFX
#undef DTYPE
#undef EXPFN

int oracle_abi_version(void) { return 1; }
