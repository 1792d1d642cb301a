"""Summary of a rocprofv3 kernel trace of bench.py's C2 leg (tools/trace_c2.sh): per kernel name the launches, total and
mean duration inside the window of the last LAST launches, the union of busy time and the wall time of that window."""
import csv, sys, collections
path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), r.get("Queue_Id", "")))
rows.sort()
if last: rows = rows[-last:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, nme, g, q in rows:
    agg[nme][0] += 1; agg[nme][1] += e - s
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches, some kernel running {busy / 1e6:.2f} ms, queues {len(set(r[4] for r in rows))}")
for nme, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / 1e6:10.2f} ms {c:6d} x {d / c / 1e3:9.1f} us  {nme[:90]}")
