#!/bin/bash
# round 2, GPU call A: regression tests + split-row A/B + traces/PMC of the split variant
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2a
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 300 python benchmarks/bench_tune.py --flags 17,25,89,121 --feats 100 --split-valid > $OUT/tune.jsonl 2> $OUT/tune.err
cat $OUT/tune.jsonl
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/benchmarks/bench_tune.py --flags 89 --feats 100 --variants U --reps 5 > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- python $ROOT/benchmarks/bench_tune.py --flags 17,89 --feats 100 --variants U --reps 3 > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections, json
out = "$OUT"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r)
for d in sorted(glob.glob(out + "/pmc_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(d.split("/")[-2], k, c, len(v), [round(x) for x in v[-6:]])
PY
