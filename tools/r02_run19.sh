#!/bin/bash
mkdir -p gpurun_out/r02s; O=$PWD/gpurun_out/r02s
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0 > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/ec_timeline.py $f k_secp
python3 tools/ec_timeline.py $f k_rist
tail -c 600 $O/trace.log
