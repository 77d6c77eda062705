#!/bin/bash
# round 2, GPU call D: hetero gradients + fixed re-order kernels, full suite, kernel split of the softmax
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2d
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_hetero_grad.py tests/test_gpu_edge_order.py tests/test_gpu_softmax_kernels.py tests/test_gpu_sharded.py -m gpu -q > $OUT/tests_new.log 2>&1
tail -40 $OUT/tests_new.log
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log
timeout 300 python benchmarks/exp_edge_order.py > $OUT/edge_order.jsonl 2> $OUT/edge_order.err
cat $OUT/edge_order.jsonl; tail -3 $OUT/edge_order.err
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sm -o t -- python $ROOT/benchmarks/exp_softmax_scale.py > $OUT/trace_sm.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace_sm/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
