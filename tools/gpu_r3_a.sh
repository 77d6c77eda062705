#!/bin/bash
# round 3, call A: new tests + full-size verify + bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_mm.py tests/test_gpu_hetero_stacked_cmp.py tests/test_gpu_bench_multi.py tests/test_sampling.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3/a_tests.log
timeout 1500 python benchmarks/verify_full.py > gpurun_out/r3/verify_full.jsonl 2> gpurun_out/r3/verify_full.err
echo "verify rc=$?" >> gpurun_out/r3/a_tests.log
timeout 600 python bench.py > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err
timeout 600 python benchmarks/bench_ops.py --only MM > gpurun_out/r3/mm_a.jsonl 2>&1
cat gpurun_out/r3/a_tests.log; tail -3 gpurun_out/r3/verify_full.err; cat gpurun_out/r3/verify_full.jsonl | cut -c1-400; cut -c1-1500 gpurun_out/r3/bench_a.json
