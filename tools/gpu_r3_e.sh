#!/bin/bash
# round 3, call E: wave-level fix-up slots — full suite + headline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r3/e_tests.log
cat gpurun_out/r3/e_tests.log
timeout 600 python bench.py --no-cpu > gpurun_out/r3/bench_e.json 2> gpurun_out/r3/bench_e.err
cut -c1-1400 gpurun_out/r3/bench_e.json
timeout 600 python benchmarks/bench_tune.py --variants U --feats 100 --flags 0,25,281 > gpurun_out/r3/tune_e.jsonl 2>&1
cat gpurun_out/r3/tune_e.jsonl
