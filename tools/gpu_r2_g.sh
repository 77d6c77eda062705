#!/bin/bash
# round 2, GPU call G: kernel split + cache counters of the edge softmax at 62 M edges, no edge-id map
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/benchmarks/exp_softmax_scale.py 2 nomap > $OUT/trace.log 2>&1
for C in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- python $ROOT/benchmarks/exp_softmax_scale.py 2 nomap > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
out = "$OUT"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
for d in sorted(glob.glob(out + "/pmc_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "edge_softmax" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][12:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(k, c, len(v), round(sum(v) / len(v)))
PY
