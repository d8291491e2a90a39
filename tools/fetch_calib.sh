#!/bin/bash
# FETCH_SIZE calibrated on known byte counts in the tick's access pattern (run on the GPU box through gpurun, from the repo root):
#   tools/fetch_calib.sh <tag>   ->  gpurun_out/<tag>_fetch_calibration.json
# One --pmc group per pass; the program itself follows `--`.
set -e -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib_$TAG
mkdir -p $OUT
hipcc -O2 --offload-arch=gfx950 $ROOT/tools/fetch_calib.hip -o $OUT/fetch_calib
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o k --output-format csv -- $OUT/fetch_calib > $OUT/known.jsonl 2> $OUT/fetch.log
# the raw request counters FETCH_SIZE is derived from, for the calibration kernels and for the tick (each pass may be refused by this ROCm: not fatal)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/raw -o k --output-format csv -- $OUT/fetch_calib > /dev/null 2> $OUT/raw.log || echo "raw pass failed (see raw.log)"
rocprofv3 --pmc TCC_BUBBLE_sum TCC_REQ_sum -d $OUT/raw2 -o k --output-format csv -- $OUT/fetch_calib > /dev/null 2> $OUT/raw2.log || echo "raw2 pass failed (see raw2.log)"
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --rotate-inputs 9"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $OUT/tick_raw -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/tick_raw.log || echo "tick raw pass failed"
rocprofv3 --pmc TCC_BUBBLE_sum TCC_REQ_sum -d $OUT/tick_raw2 -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/tick_raw2.log || echo "tick raw2 pass failed"
rocprofv3 --pmc FETCH_SIZE -d $OUT/tick_fetch -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/tick_fetch.log
python3 $ROOT/tools/fetch_calib_summary.py $OUT $ROOT/gpurun_out/${TAG}_fetch_calibration.json
