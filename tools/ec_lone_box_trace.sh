# usage: tools/ec_lone_box_trace.sh [K]  -- kernel timeline of curve-group boxes verified ONE AT A TIME (K per call, default 1; the call is made three times, the last is shown) for both groups:
# the last box's launches in start order with their durations, and the span from its first launch to its last end.
export TMPDIR=/tmp
K=${1:-1}
for g in secp256k1 ristretto255; do
  rm -rf gpurun_out/tl_ec_$g; mkdir -p gpurun_out/tl_ec_$g
  MPVSS_BOX_REPEAT=3 MPVSS_BENCH_EC_DEPTH=1 MPVSS_BENCH_EC_HASH_THREADS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_ec_$g -- python3 tools/ec_box_for_pmc.py $g $K > gpurun_out/tl_ec_$g/log.txt 2>&1
  tail -1 gpurun_out/tl_ec_$g/log.txt
  python3 - $g <<'PY'
import csv, glob, os, sys
g = sys.argv[1]
f = sorted(glob.glob(f'gpurun_out/tl_ec_{g}/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last box: everything from the last k_*_decode of commitments (first launch of a box's X path) on
pre = 'k_secp' if g == 'secp256k1' else 'k_rist'
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(pre + '_decode')]
i0 = starts[-1]
while i0 > 0 and int(rows[i0]['Start_Timestamp']) - int(rows[i0 - 1]['End_Timestamp']) < 300_000: i0 -= 1
box = rows[i0:]
t0 = int(box[0]['Start_Timestamp'])
for r in box:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"  +{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e6:8.3f} ms  q{r['Queue_Id']} {r['Kernel_Name'][:34]:34} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']}")
print(f"  {g}: span of the last box {(max(int(r['End_Timestamp']) for r in box) - t0) / 1e6:.2f} ms, sum of launches {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in box) / 1e6:.2f} ms")
PY
done
