#!/bin/bash
# round 3, call D: edge layout of the split rows — parity + A/B against the classic layout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_gpu_seam.py tests/test_gpu_sharded.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_alignment_streams.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3/d_tests.log
cat gpurun_out/r3/d_tests.log
timeout 900 python benchmarks/bench_tune.py --variants U --feats 100 --flags 0,25,281 --split-valid > gpurun_out/r3/tune_d.jsonl 2>&1
cat gpurun_out/r3/tune_d.jsonl
timeout 600 python bench.py --no-cpu > gpurun_out/r3/bench_d.json 2> gpurun_out/r3/bench_d.err
cut -c1-1500 gpurun_out/r3/bench_d.json
