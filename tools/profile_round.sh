#!/bin/bash
# Profiles of one bench.py run for profiles/ (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh <tag>      ->  gpurun_out/prof_<tag>/{stats,fetch,write,sq_a,sq_b}/...
# rocprofv3 passes are kept separate (kernel trace + stats; one --pmc group per pass), the program itself
# follows `--` (no env / shell hops), outputs land under gpurun_out/.
set -e -o pipefail
TAG=${1:-vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o k --output-format csv -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/write.log
# the same two counters with the inputs ROTATED through 9 distinct copies (295 MB > the 256 MB MALL): consecutive launches read different
# memory -- if FETCH_SIZE then rises to the compulsory bytes, the shortfall of the plain pass was cache reuse between launches
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_rot -o k --output-format csv -- $BENCH --rotate-inputs 9 > /dev/null 2> $OUT/fetch_rot.log || echo "fetch_rot pass failed"
rocprofv3 --pmc WRITE_SIZE -d $OUT/write_rot -o k --output-format csv -- $BENCH --rotate-inputs 9 > /dev/null 2> $OUT/write_rot.log || echo "write_rot pass failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq_a -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/sq_a.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d $OUT/sq_b -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/sq_b.log
# lane utilisation of the vector instructions: thread-cycles / (instruction-cycles x 64) = rocprofiler's VALUUtilization
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/sq_c -o k --output-format csv -- $BENCH > /dev/null 2> $OUT/sq_c.log || echo "sq_c pass failed (see sq_c.log)"
# the solver-level plug (L0): kernel durations of the dense and the prepared entry at B = 1 and B = 8192 (device time without PCIe)
rocprofv3 --kernel-trace --stats -d $OUT/l0 -o k --output-format csv -- python3 $ROOT/tools/bench_latency.py --l0 1 8192 > $OUT/l0_latency_under_rocprof.json 2> $OUT/l0.log || echo "l0 pass failed (see l0.log)"
# (SKIP_FULL=1: the full bench line -- cpu_baseline, secondary points, L0 -- is taken in a gpurun call of its own)
[ -n "$SKIP_FULL" ] || python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/bench_full.json 2> $OUT/bench_full.err
find $OUT -name "*.csv" | head -20
