#!/bin/bash
# segment_mm fp32 (3 x bf16) at the R-GCN shape: SQ counters of the kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2l
mkdir -p $O
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $R/benchmarks/prof_mm_f32.py > $O/pmc_$N.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "segment_mm" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("$O/pmc_SQ_VALU*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in agg.items():
    v = sorted(dur.get(k, [0]))
    print(k, "median us", v[len(v)//2])
    for c, x in sorted(cs.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(x), sum(x)/len(x)))
PY
