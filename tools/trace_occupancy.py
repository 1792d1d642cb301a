#!/usr/bin/env python3
"""What runs beside what: from a rocprofv3 --kernel-trace CSV, per kernel name -- launches, mean / total duration -- and a time-weighted
histogram of how many kernels (and how many WAVES' worth of grid) were running at once, over the window [skip_ms, end - tail_ms].
  python3 tools/trace_occupancy.py kernel_trace.csv [kernel-name-filter for the window = first .. last launch of it]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else None
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["waves"] = (int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)) // 64
sel = [r for r in rows if filt in r["Kernel_Name"]] if filt else rows
t0, t1 = min(r["s"] for r in sel), max(r["e"] for r in sel)
win = [r for r in rows if r["e"] > t0 and r["s"] < t1]
print(f"window {(t1 - t0) / 1e6:.1f} ms, {len(win)} launches")
by = collections.defaultdict(lambda: [0, 0.0, 0])
for r in win:
    k = r["Kernel_Name"].split("(")[0][:44]
    by[k][0] += 1
    by[k][1] += (min(r["e"], t1) - max(r["s"], t0)) / 1e6
    by[k][2] = max(by[k][2], r["waves"])
tot = (t1 - t0) / 1e6
print(f"{'kernel':44} {'launches':>8} {'sum ms':>10} {'mean ms':>9} {'queues':>7} {'max waves':>9}")
for k, (c, ms, wv) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44} {c:8d} {ms:10.1f} {ms / c:9.2f} {ms / tot:7.2f} {wv:9d}")
ev = []
for r in win:
    ev.append((max(r["s"], t0), 1, r["waves"]))
    ev.append((min(r["e"], t1), -1, -r["waves"]))
ev.sort()
hist = collections.Counter()
wide = collections.Counter()
n = w = 0
last = t0
for t, d, dw in ev:
    hist[n] += t - last
    wide[min(w // 512, 16)] += t - last
    last = t
    n += d
    w += dw
print("kernels running at once (share of the window):", {k: round(v / (t1 - t0), 3) for k, v in sorted(hist.items())})
print("waves in the grids of the running kernels, in units of 512 (share of the window):", {k: round(v / (t1 - t0), 3) for k, v in sorted(wide.items())})
