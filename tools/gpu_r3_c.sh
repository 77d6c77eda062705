#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3/c_tests.log
cat gpurun_out/r3/c_tests.log
timeout 900 python benchmarks/bench_ops.py --only EO 2>&1 | cut -c1-330 > gpurun_out/r3/eo_c.jsonl
cat gpurun_out/r3/eo_c.jsonl
