# usage: tools/lone_box.sh  -- kernel durations of boxes verified one at a time (no overlap), with and without stream priorities
export TMPDIR=/tmp
for cfg in "1" "0"; do
  rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
  MPVSS_STREAM_PRIO=$cfg MPVSS_BENCH_DEPTH=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --wb-shares 0 --registered-keys 0 > gpurun_out/tl/log.txt 2>&1
  echo "stream_prio=$cfg"
  python3 - <<'PY'
import csv,glob,os
f=sorted(glob.glob('gpurun_out/tl/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-40:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if d>5: print(f"{d:8.2f} q{r['Queue_Id']} {r['Kernel_Name'][:26]} grid={r['Grid_Size_X']}")
PY
done
