#!/bin/bash
# round 2, GPU call C: new kernels' tests, full suite, softmax + edge-order measurements
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2c
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_softmax_kernels.py tests/test_gpu_edge_order.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/tests_new.log 2>&1
tail -25 $OUT/tests_new.log
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log
timeout 300 python benchmarks/exp_softmax_scale.py > $OUT/softmax_scale.jsonl 2> $OUT/softmax_scale.err
cat $OUT/softmax_scale.jsonl; tail -3 $OUT/softmax_scale.err
timeout 300 python benchmarks/exp_edge_order.py > $OUT/edge_order.jsonl 2> $OUT/edge_order.err
cat $OUT/edge_order.jsonl; tail -3 $OUT/edge_order.err
