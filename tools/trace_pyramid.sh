#!/bin/bash
# (GPU box) rocprofv3 kernel trace of the C4 pyramid step (bench.py --workload c4_pyramid): per-call-site kernel durations
# -> gpurun_out/r03/trace_pyramid.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/gpurun_out/r03" && cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/trp -o t -- python3 "$R/bench.py" --workload c4_pyramid --steps 10 --warmup 3 --no-cpu-baseline --no-per-config > "$R/gpurun_out/r03/trace_pyramid.json" 2> "$R/gpurun_out/r03/trace_pyramid.err" &&
python3 "$R/tools/rocpd_export.py" trace $(find /tmp/trp -name "*_results.db") > /tmp/trp/trace.csv &&
python3 "$R/tools/callsite_stats.py" /tmp/trp/trace.csv > "$R/gpurun_out/r03/trace_pyramid.csv"
echo rc=$?
