#!/usr/bin/env python3
"""profiles/rNN_ec_counters.json from tools/run_profiles_ec.sh: per curve group, per verified box (n=65536, t=256):
VALU wave-instructions (SQ_INSTS_VALU) and HBM bytes ((2 FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of
MI355X_MICROARCH.md), each as (sum over all dispatches with K = 32 verifications - sum with none) / K.
usage: summarize_ec.py <gpurun_out/prof_ec> <out.json>"""
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]


def total(pattern, counter):
    files = sorted(glob.glob(f"{src}/{pattern}/*/*counter_collection.csv"), key=os.path.getmtime)      # (the newest pass: gpurun merges runs)
    assert files, pattern
    files = files[-1:]
    return sum(float(r["Counter_Value"]) for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == counter)


K = 32
HOW = (f"rocprofv3 --pmc passes over tools/ec_box_for_pmc.py (tools/run_profiles_ec.sh): K = {K} minus K = 0 verifications through "
       "mpvss_ec_verify_many with bench.py's depth (16 boxes in flight, X paths of 16 boxes per launch) and hash threads, per box")
doc = {"source": HOW}
for g in ("secp256k1", "ristretto255"):
    valu = (total(f"sq_{g}_{K}", "SQ_INSTS_VALU") - total(f"sq_{g}_0", "SQ_INSTS_VALU")) / K
    fetch = (total(f"fetch_{g}_{K}", "FETCH_SIZE") - total(f"fetch_{g}_0", "FETCH_SIZE")) / K
    write = (total(f"write_{g}_{K}", "WRITE_SIZE") - total(f"write_{g}_0", "WRITE_SIZE")) / K
    doc[g] = {"valu_wave_insts_per_box": valu, "valu_insts_per_lane": valu / 1024,      # one share per lane, 1024 waves per box
              "hbm_bytes_per_box": (2 * fetch + write) * 1024, "fetch_size_kb": fetch, "write_size_kb": write, "how": HOW}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1))
