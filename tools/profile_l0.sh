#!/bin/bash
# rocprofv3 kernel statistics of the L0 kernels over tools/bench_latency.py --l0 1 8192 (run on the GPU box from the repo root):
#   tools/profile_l0.sh <tag>   ->  gpurun_out/prof_l0_<tag>/  (kernel_stats.csv is what gets committed under profiles/)
TAG=${1:-l0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_l0_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o k --output-format csv -- python3 $ROOT/tools/bench_latency.py --l0 1 8192 > $OUT/bench.json 2> $OUT/log.txt
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("qpc_qp_sp")]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
    if d:
        print("qpc_qp_sp_kernel launches", len(d), "min %.1f us  median %.1f us  max %.1f us" % (d[0], d[len(d) // 2], d[-1]))
PY
