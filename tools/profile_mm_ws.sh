#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics and two PMC passes (issue / wait states,
# instruction mix) of the weights-stationary segment_mm kernels.  Usage: bash tools/profile_mm_ws.sh <tag>
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mm_ws_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/benchmarks/bench_mm_ws.py --profile"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -o p -- $CMD > $OUT/tcc.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections, json
out = "$OUT"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "segment_mm_ws" in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            name = name[name.index("segment_mm_ws"):][:60]
            rows[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.jsonl", "w") as fo:
    for (name, grid), cs in sorted(rows.items()):
        fo.write(json.dumps({"kernel": name, "grid": grid, "counters_mean_per_launch": {c: sum(v) / len(v) for c, v in sorted(cs.items())},
                             "launches": {c: len(v) for c, v in cs.items()}}) + "\n")
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    import shutil
    shutil.copy(f, out + "/kernel_stats.csv")
print(open(out + "/pmc_summary.jsonl").read()[:3000])
PY
