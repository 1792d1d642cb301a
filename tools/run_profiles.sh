#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's kernels (run on the GPU box through gpurun): profiles/rNN_* are copies of what
# this writes under gpurun_out/prof (tools/summarize_sq.py, tools/summarize_pmc.py make the JSON summaries).
#   1. --kernel-trace --stats of the default bench configuration (boxes overlapping, as timed)
#   2. --kernel-trace of boxes verified ONE AT A TIME (MPVSS_BENCH_DEPTH=1): isolated kernel durations
#   3. PMC passes (separate runs, no tracing): FETCH_SIZE / WRITE_SIZE, SQ busy / wait counters
# The program goes directly after `--` (no env/bash wrappers: the profiler initialises the GPU before the program).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
ARGS="bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 2 --steady-steps 0 --drop-in-threads 0"
LONE="bench.py --steps 3 --warmup 1 --lone-boxes 2 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --steady-steps 0 --drop-in-threads 0"
PMCARGS="bench.py --steps 2 --warmup 1 --lone-boxes 0 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 2 --host-boxes 0 --config-boxes 0 --steady-steps 0 --drop-in-threads 0"
ECARGS="bench.py --steps 2 --warmup 1 --lone-boxes 0 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --host-boxes 0 --config-boxes 0 --steady-steps 0 --drop-in-threads 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
grep "^{\"metric\"" $OUT/trace.log > $OUT/bench_under_rocprof.json
export MPVSS_BENCH_DEPTH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lone -- python3 $LONE > $OUT/trace_lone.log 2>&1
export MPVSS_BENCH_DEPTH=2
export MPVSS_BENCH_EC_DEPTH=4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $PMCARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $PMCARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $PMCARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 $PMCARGS > $OUT/pmc_mfma.log 2>&1
unset MPVSS_BENCH_DEPTH MPVSS_BENCH_EC_DEPTH
# 4. the curve groups' boxes (batched X paths), kernel trace + stats of bench.py's `ec` part
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ec -- python3 $ECARGS > $OUT/trace_ec.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
find $OUT -name "*.csv" | xargs ls -la
