#!/bin/bash
# usage: pmc_l0.sh <tag> <counters...>   (run from the repo root on the GPU box): PMC counters of the L0 workgroup kernels over
# tools/bench_latency.py --l0 8192 (the prepared entry's cold batch of 8 192 mixed problems is the longest launch of each kernel)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_l0_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" -d $OUT -o k --output-format csv -- python3 $ROOT/tools/bench_latency.py --l0 8192 > /dev/null 2> $OUT/log.txt
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("qpc_qp_"):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    # the launch with the most wave cycles = the cold batch of 8 192
    key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in d else list(d)[0]
    i = max(range(len(d[key])), key=lambda j: d[key][j])
    print("$TAG", k, {c: v[i] for c, v in d.items()}, "launches", len(d[key]))
PY
