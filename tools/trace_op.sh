#!/bin/bash
# (GPU box) rocprofv3 kernel trace of tools/opbench.py for the operators whose names contain the arguments; per-call-site
# kernel durations go to gpurun_out/r03/trace_op.csv.   usage: bash tools/trace_op.sh "knn_feat B32" ["three_interp_bwd" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/gpurun_out/r03" && cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python3 "$R/tools/opbench.py" "$@" > "$R/gpurun_out/r03/trace_op.log" 2>&1 &&
python3 "$R/tools/rocpd_export.py" trace $(find /tmp/tr -name "*_results.db") > /tmp/tr/trace.csv &&
python3 "$R/tools/callsite_stats.py" /tmp/tr/trace.csv > "$R/gpurun_out/r03/trace_op.csv"
echo rc=$?
