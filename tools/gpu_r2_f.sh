#!/bin/bash
# round 2, GPU call F: register-resident edge softmax — tests, then scale bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2f
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_softmax_kernels.py tests/test_golden.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_seam.py tests/test_gpu_large.py -m gpu -q -x -k "softmax or gat or GAT" > $OUT/tests_sm.log 2>&1
tail -25 $OUT/tests_sm.log
timeout 300 python benchmarks/exp_softmax_scale.py > $OUT/softmax_scale.jsonl 2> $OUT/softmax_scale.err
cat $OUT/softmax_scale.jsonl; tail -3 $OUT/softmax_scale.err
