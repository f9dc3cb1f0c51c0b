#!/bin/bash
# SQ counters of the cost kernels at the 650 MB sizes (run from the repository root through gpurun).
# usage: pmc_cost.sh <set: valu|wait> [workloads...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
SET=${1:-valu}; shift || true
case $SET in
  valu) CNT="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT";;
  wait) CNT="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS";;
esac
cd /tmp && export TMPDIR=/tmp
for w in ${@:-cfg4_ce cfg4 cfg5}; do
    timeout -s KILL 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc_cost_${SET}_$w -- \
        python3 $GRAFT_REPO_ROOT/tools/kbench.py --workload $w --which cost --iters 20 > $OUT/pmc_cost_${SET}_$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $(for w in ${@:-cfg4_ce cfg4 cfg5}; do echo $OUT/pmc_cost_${SET}_$w; done)
