#!/bin/bash
# softmax: correctness subset, timing at 62 M edges, dynamic instruction counts of the kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
python tools/diag_softmax.py 2>&1 | grep bad
python -m pytest tests -m gpu -q -x -k "softmax" 2>&1 | tail -3
python benchmarks/exp_softmax_scale.py 2 nomap
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
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
