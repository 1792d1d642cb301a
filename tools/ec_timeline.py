"""Concurrency picture of the curve-group boxes from a rocprofv3 --kernel-trace CSV: for the kernels whose name contains
`pat` (default k_secp), per kernel name the calls, mean duration and share of busy time; the distribution of how many of
them run at once; per hardware queue the busy fraction."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if (sys.argv[2] if len(sys.argv) > 2 else "k_secp") in r["Kernel_Name"]]
ev = []
per = collections.defaultdict(lambda: [0, 0.0])
qbusy = collections.defaultdict(float)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    k = r["Kernel_Name"].split("(")[0]
    per[k][0] += 1; per[k][1] += (e - s) / 1e6
    qbusy[r["Queue_Id"]] += (e - s) / 1e6
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
hist = collections.defaultdict(float)
cur, last = 0, t0
for t, d in ev:
    hist[cur] += (t - last) / 1e6
    cur += d; last = t
span = (t1 - t0) / 1e6
print(f"span {span:.1f} ms, {len(rows)} launches")
for k, (c, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32} calls {c:5d} mean {ms / c:8.3f} ms  sum {ms:9.1f} ms")
print("  kernels running at once -> share of the span:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
print("  per queue busy fraction:", {q: round(v / span, 2) for q, v in sorted(qbusy.items())})
