"""The boundary really is a C ABI: include/dgl_amd.h compiles as plain C11 (gcc, -pedantic
-Werror), a C program links against libdgl_amd.so using only that header, and the calls that
need no GPU behave (version, error strings, registry lookup, host-side partitioner)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r'''
#include <stdio.h>
#include <string.h>
#include "dgl_amd.h"

int main(void) {
  if (dgla_abi_version() != DGLA_ABI_VERSION) return 1;
  /* a failing call returns -1 and leaves a message; nothing throws across the boundary */
  if (dgla_spmm_csr("copy_lhs", "sum", NULL, DGLA_F32, NULL, NULL, NULL, NULL, NULL, NULL, 0, 0, NULL) != -1) return 2;
  if (strstr(dgla_last_error(), "null") == NULL) return 3;
  /* registry: the reference's names resolve to handles, unknown names to NULL without error */
  DGLFunctionHandle h = NULL;
  if (DGLFuncGetGlobal("sparse._CAPI_DGLKernelSpMM", &h) != 0 || h == NULL) return 4;
  if (DGLFuncGetGlobal("sparse._CAPI_DGLKernelSEGMENTMM", &h) != 0 || h == NULL) return 5;
  if (DGLFuncGetGlobal("no.such.name", &h) != 0 || h != NULL) return 6;
  /* host code: 2 cliques joined by one edge split into their cliques */
  int64_t indptr[9] = {0, 3, 6, 9, 13, 17, 20, 23, 26};
  int64_t indices[26] = {1, 2, 3, 0, 2, 3, 0, 1, 3, 0, 1, 2, 4, 3, 5, 6, 7, 4, 6, 7, 4, 5, 7, 4, 5, 6};
  int64_t part[8], stats[4];
  if (dgla_partition_kway(64, 8, indptr, indices, 2, 0.1, 0, 1, part, stats) != 0) return 7;
  for (int i = 1; i < 4; ++i) if (part[i] != part[0]) return 8;
  for (int i = 5; i < 8; ++i) if (part[i] != part[4]) return 9;
  if (part[0] == part[4]) return 10;
  if (stats[0] != 2) return 11; /* the bridge is stored in both directions: 2 stored edges cut */
  printf("C ABI ok, cut=%lld\n", (long long)stats[0]);
  return 0;
}
'''


def test_header_is_plain_c_and_links(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text(C_SRC)
    exe = tmp_path / "abi"
    lib_dir = os.path.join(ROOT, "dgl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror",
                           "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-ldgl_amd", "-Wl,-rpath," + lib_dir])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = lib_dir + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "C ABI ok, cut=2" in out.stdout
