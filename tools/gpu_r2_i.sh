#!/bin/bash
# per-kernel split of the fused edge softmax at 62 M edges, no edge-id map; SQ counters of the main kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o sm -- python $R/benchmarks/exp_softmax_scale.py 2 nomap > $O/run.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-200
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $R/benchmarks/exp_softmax_scale.py 2 nomap > $O/pmc_$N.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "softmax" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%d mean=%.4g" % (c, len(v), sum(v)/len(v)))
PY
