#!/bin/bash
# Round-3 profiling pass (run on the GPU box through gpurun): kernel trace + stats of the bench command, HBM traffic
# counters (FETCH_SIZE / WRITE_SIZE in separate passes), MFMA counters on the pairwise / feature-kNN shapes.
# Every step under its own time limit; the first failure ends the script.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d "$O/prof_bench" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err" &&
echo "trace done" &&
# MFMA counters: one pair of passes per shape (every sweep launch has the same grid, so launches of different shapes
# cannot be told apart inside one pass)
for S in "pairwise B32 N1024" "pairwise B1 N8192" "pairwise B1 N16384" "knn_feat B32" "knn_feat B1 N12500" "knn_feat B1 N100000"; do
  T=$(echo "$S" | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$O/pmc_mfma1_$T" -o mfma -- python3 "$R/tools/opbench.py" "$S" > "$O/pmc_mfma1_$T.log" 2>&1 &&
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU SQ_BUSY_CYCLES -d "$O/pmc_mfma2_$T" -o mfma -- python3 "$R/tools/opbench.py" "$S" > "$O/pmc_mfma2_$T.log" 2>&1 &&
  python3 "$R/tools/rocpd_export.py" counters $(find "$O/pmc_mfma1_$T" "$O/pmc_mfma2_$T" -name "*_results.db") | awk -v s="$S" 'NR==1{print "shape," $0; next}{print s "," $0}' > "$O/pmc_mfma_$T.csv" &&
  rm -rf "$O/pmc_mfma1_$T" "$O/pmc_mfma2_$T" || exit 1
done
awk 'FNR==1 && NR!=1{next}{print}' "$O"/pmc_mfma_*.csv > "$O/pmc_mfma_summary.csv" &&
echo "mfma passes done" &&
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d "$O/pmc_fetch" -o fetch -- python3 "$R/bench.py" --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-per-config > "$O/pmc_fetch.log" 2>&1 &&
echo "fetch pass done" &&
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d "$O/pmc_write" -o write -- python3 "$R/bench.py" --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-per-config > "$O/pmc_write.log" 2>&1 &&
echo "write pass done" &&
timeout -k 10 420 python3 "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err" &&
echo "bench done" &&
timeout -k 10 120 python3 "$R/bench.py" --streams 1 --no-cpu-baseline --no-per-config > "$O/bench_streams1.json" 2> "$O/bench_streams1.err" &&
echo "runs done" &&
python3 "$R/tools/rocpd_export.py" stats $(find "$O/prof_bench" -name "*_results.db") > "$O/bench_kernel_stats.csv" &&
python3 "$R/tools/rocpd_export.py" trace $(find "$O/prof_bench" -name "*_results.db") > "$O/bench_kernel_trace.csv" &&
python3 "$R/tools/rocpd_export.py" counters $(find "$O/pmc_fetch" "$O/pmc_write" -name "*_results.db") > "$O/pmc_traffic_summary.csv" &&
rm -rf "$O/prof_bench" "$O/pmc_fetch" "$O/pmc_write" &&
echo "all done"
