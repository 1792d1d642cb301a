#!/bin/bash
# SQ counters of the C5 slice's stepping kernels, persistent stages (MPVSS_FD_TILE=0) against wide launches (MPVSS_FD_TILE=2), kernels one at a
# time under counter collection:  tools/pmc_c5_step.sh   ->  gpurun_out/pmc_c5_{persistent,tile}.json (tools/summarize_sq.py)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MPVSS_BENCH_CONFIGS=c5_slice MPVSS_BENCH_C5_DEPTH=2
for v in persistent tile; do
  if [ $v = tile ]; then export MPVSS_FD_TILE=2; else export MPVSS_FD_TILE=0; fi
  rm -rf gpurun_out/pmc_c5_$v
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv \
    -d gpurun_out/pmc_c5_$v -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 \
    --config-boxes 16 --lone-boxes 0 --steady-steps 0 > gpurun_out/pmc_c5_$v.log 2>&1
  CSV=$(find gpurun_out/pmc_c5_$v -name '*counter_collection.csv' | head -1)
  python3 tools/summarize_sq.py "$CSV" gpurun_out/pmc_c5_$v.json > /dev/null 2>&1
  python3 -c "
import json
d = json.load(open('gpurun_out/pmc_c5_$v.json'))['kernels']
for k, v in d.items():
    if 'fd_step' in k or 'commit_eval' in k or 'fd_table' in k or 'dual_exp_w6' in k:
        print('$v', k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x in ('launches', 'grid', 'vgprs', 'valu_insts_per_wave', 'active_valu_frac', 'wait_inst_frac', 'wait_any_frac', 'simd_valu_util')})
"
  rm -rf gpurun_out/pmc_c5_$v
done
