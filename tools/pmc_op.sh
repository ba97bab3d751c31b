#!/bin/bash
# (GPU box) one rocprofv3 counter pass over tools/opbench.py; per-(kernel, grid) counter means go to gpurun_out/r03/pmc_op_<tag>.csv.
# usage: bash tools/pmc_op.sh <tag> "<counter> <counter> ..." "<opbench filter>" [...]   (at most 8 SQ_ / 4 TCC_ / 2 GRBM_ per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CTRS=$2; shift 2
mkdir -p "$R/gpurun_out/r03" && cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$TAG &&
timeout -s KILL 200 rocprofv3 --pmc $CTRS -d /tmp/pmc_$TAG -o p -- python3 "$R/tools/opbench.py" "$@" > "$R/gpurun_out/r03/pmc_op_$TAG.log" 2>&1 &&
python3 "$R/tools/rocpd_export.py" counters $(find /tmp/pmc_$TAG -name "*_results.db") > "$R/gpurun_out/r03/pmc_op_$TAG.csv"
echo rc=$?
