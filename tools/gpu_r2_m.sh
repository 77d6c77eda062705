#!/bin/bash
# GAT-size (ogbn-arxiv shape) operators: per-kernel durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2m; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $R/benchmarks/bench_ops.py --only C3,GAT > $O/run.log 2>&1
grep -E '"config"' $O/run.log | cut -c1-200
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
head -25 "$f" | cut -c1-170
