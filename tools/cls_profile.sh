#!/bin/bash
# GPU box: the classification-loss tests, then a kernel trace of tools/clsbench.py at the cfg4 shape and its summary.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_cls_loss.py -x -q --timeout 120 --timeout-method thread > gpurun_out/cls_tests.txt 2>&1 || { tail -n 30 gpurun_out/cls_tests.txt; exit 1; }
rm -rf gpurun_out/prof_cls
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_cls" -- python3 "$GRAFT_REPO_ROOT/tools/clsbench.py" 256 900 92 > "$GRAFT_REPO_ROOT/gpurun_out/prof_cls.log" 2>&1) || exit 1
trace=$(find gpurun_out/prof_cls -name "*kernel_trace.csv" | head -n 1)
python tools/cls_trace_summary.py "$trace" 256 900 92 | tee gpurun_out/cls_trace_summary.txt
