#!/bin/bash
# VALU instructions and HBM traffic of ONE curve-group verification (n=65536, t=256): PMC passes over tools/ec_box_for_pmc.py
# with K = 0 and K = 32 verifications (16 boxes in flight, X paths batched: the call bench.py times); tools/summarize_ec.py takes the differences.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_ec
rm -rf $OUT; mkdir -p $OUT
for g in secp256k1 ristretto255; do
  for k in 0 32; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_${g}_$k -- python3 tools/ec_box_for_pmc.py $g $k > $OUT/sq_${g}_$k.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_${g}_$k -- python3 tools/ec_box_for_pmc.py $g $k > $OUT/fetch_${g}_$k.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_${g}_$k -- python3 tools/ec_box_for_pmc.py $g $k > $OUT/write_${g}_$k.log 2>&1
  done
done
find $OUT -name "*counter_collection.csv" | xargs ls -la
