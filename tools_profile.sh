#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + PMC passes of bench.py.
# Usage: bash tools_profile.sh <tag>
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu --no-variants --no-peak"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- $B --steps 3 --warmup 1 > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, os, collections, json
out = "$OUT"
summary = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    summary["kernel_stats"] = [r for r in csv.DictReader(open(f))][:12]
for d in glob.glob(out + "/pmc_*/"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary[os.path.basename(d.rstrip("/"))] = {k: {c: {"n": len(v), "mean": sum(v)/len(v)} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
PY
