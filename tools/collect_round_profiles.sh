#!/bin/bash
# After tools/run_profiles.sh (on the GPU box): the files a round commits, under gpurun_out/prof/round/ (copied to profiles/rNN_* by hand).
set -u
cd "$(dirname "$0")/.."
P=gpurun_out/prof
R=$P/round
mkdir -p $R/rocprofv3
one() { ls $1 2>/dev/null | head -1; }
cp "$(one "$P/trace/*/*kernel_stats.csv")" $R/rocprofv3/kernel_stats.csv
cp "$(one "$P/trace_lone/*/*kernel_stats.csv")" $R/rocprofv3/kernel_stats_one_box_at_a_time.csv
cp "$(one "$P/trace_lone/*/*kernel_trace.csv")" $R/rocprofv3/kernel_trace_one_box_at_a_time.csv
cp "$(one "$P/trace_ec/*/*kernel_stats.csv")" $R/rocprofv3/kernel_stats_curve_groups.csv
cp "$(one "$P/pmc_fetch/*/*counter_collection.csv")" $R/rocprofv3/pmc_fetch_counter_collection.csv
cp "$(one "$P/pmc_write/*/*counter_collection.csv")" $R/rocprofv3/pmc_write_counter_collection.csv
cp $P/bench_under_rocprof.json $R/rocprofv3/bench_under_rocprof.json
python3 tools/summarize_pmc.py $R/rocprofv3/pmc_fetch_counter_collection.csv $R/rocprofv3/pmc_write_counter_collection.csv $R/pmc_traffic.json
python3 tools/summarize_sq.py "$(one "$P/pmc_sq/*/*counter_collection.csv")" $R/sq_summary.json $R/rocprofv3/kernel_trace_one_box_at_a_time.csv
ls -la $R $R/rocprofv3
