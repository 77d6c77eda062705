#!/bin/bash
# round 2, GPU call E: full suite after the clean-up + weighted sampling + deterministic COO
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2e
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
tail -30 $OUT/tests.log
