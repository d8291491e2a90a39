#!/bin/bash
# kernel-only durations of the prepared L0 kernel against the batch size (tools/l0_load.py under rocprofv3 --kernel-trace); run on the GPU box from the repo root
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_l0load
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o k --output-format csv -- python3 $ROOT/tools/l0_load.py $ROOT/qpcontrol.jl_amd/libqpcontrol_hip.so ${1:-0} > $OUT/log.txt 2>&1
grep "^problem" $OUT/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "prep" in r["Kernel_Name"]:
            print("grid", r.get("Grid_Size") or r.get("Grid_Size_X"), "wg", r.get("Workgroup_Size") or r.get("Workgroup_Size_X"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
