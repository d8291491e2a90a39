#!/bin/bash
# dynamic instruction counts of the tick up to each QPC_STOP point (qpc_tick.h): one rocprofv3 --pmc pass per stop point
# usage (GPU box, repo root): bash tools/pmc_stages.sh > gpurun_out/stages.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for n in 1 2 3 4 5 6 7 0; do
  if [ $n != 0 ]; then export QPC_JIT_EXTRA_FLAGS="-DQPC_STOP_AFTER=$n"; else unset QPC_JIT_EXTRA_FLAGS; fi
  bash $ROOT/tools/pmc_probe.sh stop$n SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY || exit 1
done
