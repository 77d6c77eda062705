#!/bin/bash
# round 2, GPU call B: full GPU suite (new tests first) + bench line + simulated-rank scaling
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > $OUT/tests_new.log 2>&1
tail -15 $OUT/tests_new.log
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -8 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 400 python benchmarks/bench_sharded_sim.py > $OUT/sharded_sim.jsonl 2> $OUT/sharded_sim.err
cut -c1-700 $OUT/sharded_sim.jsonl; tail -3 $OUT/sharded_sim.err
