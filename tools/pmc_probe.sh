#!/bin/bash
# usage: [PMC_BENCH_ARGS="--f32 --config 5"] pmc_probe.sh <tag> <counters...>   (run from repo root on the GPU box)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" -d $OUT -o k --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary $PMC_BENCH_ARGS > /dev/null 2> $OUT/log.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("qpc_tick_kernel"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    v=v[2:] if len(v)>2 else v
    print("$TAG", k, sum(v)/len(v))
PY
