#!/bin/bash
# Runs on the GPU box: SQ_* + GRBM_GUI_ACTIVE of the fp32 weights-stationary segment_mm, siblings paced (variant 0) and not (4):
# which clock does the chip run the kernel at?  (profiles/r4/segment_mm_ws_pmc.jsonl, second run)
ROOT=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for V in 0 4; do
  DGLA_MM_WS_VARIANT=$V timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/ws_pmc3/v$V -o p -- python $ROOT/benchmarks/bench_mm_ws.py --profile > $ROOT/gpurun_out/ws_pmc3_v$V.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
for V in (0, 4):
    rows = collections.defaultdict(dict)
    for f in glob.glob("gpurun_out/ws_pmc3/v%d/**/*counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if "segment_mm_ws_kernel<float" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    kt = {}
    for f in glob.glob("gpurun_out/ws_pmc3/v%d/**/*kernel_trace.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if "segment_mm_ws_kernel<float" in r["Kernel_Name"]:
                kt[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for d in sorted(rows)[:6]:
        print(V, d, "ms", kt.get(d), {k: round(v / 1e6, 1) for k, v in sorted(rows[d].items())})
PY
