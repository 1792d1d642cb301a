#!/bin/bash
# SQ counters of the curve kernels (synchronous boxes, one kernel at a time), for issue utilisation
mkdir -p gpurun_out/r02o; O=$PWD/gpurun_out/r02o
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/bench_ec.py --steps 2 > $O/pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2 -- python3 tools/bench_ec.py --steps 2 > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/bench_ec.py --steps 2 > $O/trace.log 2>&1
tail -3 $O/pmc2.log
python3 - <<'PY'
import csv,glob,collections
for d in ('pmc_sq','pmc_sq2'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(f'gpurun_out/r02o/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
            if r['Counter_Name']=='SQ_INSTS_VALU': cnt[k]+=1
    for k,v in agg.items():
        if cnt[k]: print(d, k, cnt[k], {c: round(x/cnt[k]) for c,x in v.items()})
for f in glob.glob('gpurun_out/r02o/trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]: print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e6,3), 'ms')
PY
