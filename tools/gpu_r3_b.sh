#!/bin/bash
# round 3, call B: new multi-rank / hetero / non-finite tests + fp32 segment_mm timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_mm.py tests/test_sharded_hetero.py tests/test_gpu_bench_multi.py tests/test_bench_sage_multi.py tests/test_gpu_sharded.py tests/test_gpu_fuzz_mm.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r3/b_tests.log
timeout 600 python benchmarks/bench_ops.py --only MM 2>&1 | grep float32 | cut -c1-260 > gpurun_out/r3/mm_b.jsonl
cat gpurun_out/r3/b_tests.log; cat gpurun_out/r3/mm_b.jsonl
