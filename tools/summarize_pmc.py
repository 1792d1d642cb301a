"""profiles/pmc_traffic.json from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/run_profiles.sh.
Per kernel: average over the launches of the bench's full-size box (largest grid of that kernel) of
(2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md)."""
import csv
import json
import sys
from collections import defaultdict

fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    by_dispatch = defaultdict(float)
    meta = {}
    for r in rows:
        by_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["Grid_Size"]))
    per = defaultdict(list)
    for d, v in by_dispatch.items():
        per[meta[d][0]].append((meta[d][1], v))
    res = {}
    for k, lst in per.items():
        gmax = max(g for g, _ in lst)
        vals = [v for g, v in lst if g == gmax]
        res[k] = (sum(vals) / len(vals), gmax, len(vals))
    return res


f = per_kernel(fetch_csv, "FETCH_SIZE")
w = per_kernel(write_csv, "WRITE_SIZE")
doc = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; counter collection serialises the "
              "kernels), bench.py n=65536 t=256, averaged over the launches with the largest grid of each kernel",
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B requests of wide "
                  "coalesced reads as 32 B, MI355X_MICROARCH.md HBM section; the scattered 16-byte table reads of "
                  "these kernels are outside the calibrated pattern, so treat as indicative)",
}
for k in sorted(f):
    if not k.startswith("k_"):
        continue
    fk, grid, calls = f[k]
    wk = w.get(k, (0.0, 0, 0))[0]
    doc[k + "_bytes_per_launch"] = int((2 * fk + wk) * 1024)
    doc[k + "_raw"] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "grid": grid, "launches": calls}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
