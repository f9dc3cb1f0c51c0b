#!/bin/bash
# One round's profile set on an MI355X (run from the repository root through gpurun):
#   kernel-trace statistics of the default bench command, the two memory-traffic counter passes
#   (separate --pmc runs, kernel trace only), SQ counters of the step's kernel, and the per-workload
#   kernel timings.  Everything lands under gpurun_out/<tag>_*; what is kept is copied to profiles/ by hand.
set -e
TAG=${1:-r3}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o bench -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/${TAG}_pmc_write.log 2>&1
for w in cfg2 cfg4; do
    timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU \
        --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq_$w -- \
        python3 $GRAFT_REPO_ROOT/tools/kbench.py --workload $w --which fused --iters 20 > $OUT/${TAG}_pmc_sq_$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write cost_small_kernel=cfg2 cost_tile_kernel=cfg4 lap_block_kernel=cfg2 \
    > $OUT/${TAG}_pmc_traffic.json
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_sq_cfg2 $OUT/${TAG}_pmc_sq_cfg4 > $OUT/${TAG}_pmc_sq.txt
for w in cfg1 cfg2 cfg3 cfg3_hw cfg3_hw2 cfg4 cfg4_ce cfg5; do
    echo "== $w"
    timeout -k 10 120 python3 tools/kbench.py --workload $w --which cost,lap,both,fused --iters 200
done > $OUT/${TAG}_kbench_all.txt 2>&1
echo done
