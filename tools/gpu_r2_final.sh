#!/bin/bash
# round 2: the measurements DESIGN.md / profiles/r2 quote — full GPU suite, bench line, rocprofv3
# kernel stats + PMC passes of bench.py, operator-level lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2final
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-1500 $OUT/bench.json; tail -2 $OUT/bench.err
bash tools_profile.sh r2 > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log
timeout 600 python benchmarks/bench_ops.py > $OUT/bench_ops.jsonl 2> $OUT/bench_ops.err
tail -3 $OUT/bench_ops.err; wc -l $OUT/bench_ops.jsonl
timeout 200 python benchmarks/bench_tune.py --flags 17,25,89 --feats 100 --split-valid > $OUT/tune.jsonl 2> $OUT/tune.err
