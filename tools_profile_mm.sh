ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_glds
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/benchmarks/bench_ops.py --only C3,SEG,MM,SAMPLE,GAT > $OUT/trace.log 2>&1
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- python $ROOT/benchmarks/prof_mm.py > $OUT/pmc.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections, json, shutil
out = "$OUT"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, out + "/kernel_stats_ops.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items() if "segment_mm" in k}
for k, cs in res.items():
    if cs.get("SQ_WAVE_CYCLES"):
        cs["derived: wave time parked (WAIT_ANY / WAVE_CYCLES)"] = cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"]
json.dump(res, open(out + "/pmc_segment_mm.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $OUT/trace $OUT/pmc
