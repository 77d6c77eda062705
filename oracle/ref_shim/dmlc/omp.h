/* Shim for dmlc-core's <dmlc/omp.h> (empty submodule): it only forwards to OpenMP. */
#pragma once
#include <omp.h>
