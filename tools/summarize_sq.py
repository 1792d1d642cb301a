"""profiles/rNN_sq_summary.json from the rocprofv3 SQ counter pass of tools/run_profiles.sh.
Per kernel (launches with the largest grid only, i.e. the full-size boxes), averaged over those launches:
  wave_cycles, active_valu, wait_inst_any, wait_any   SQ_* in quad-cycles (MI355X_MICROARCH.md, PMC section), summed over the chip
  valu_insts_per_wave                                  SQ_INSTS_VALU / SQ_WAVES
  active_valu_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES       share of a resident wave's time in which it issues VALU work
  wait_inst_frac   = SQ_WAIT_INST_ANY   / SQ_WAVE_CYCLES       issue stalls (dependency / pipe)
  wait_any_frac    = SQ_WAIT_ANY        / SQ_WAVE_CYCLES       parked on s_waitcnt / barrier / sleep
  simd_valu_util   = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE * 1024)   VALU-issue utilisation of the 1024 SIMDs while the
                     kernel ran ALONE (counter collection serialises the kernels): the figure to hold against bench.py's
                     compute.frac, which is measured with eight boxes overlapping
  With the kernel_trace.csv of the one-box-at-a-time run as third argument the isolated duration (the shortest launch of
  the same grid: the one that had the GPU to itself) gives the issue rate: valu_insts_per_wave * waves / 1024 SIMDs / duration, as cycles per
  wave-instruction and SIMD at the nominal 2.4 GHz and at the ~2.1 GHz the chip holds at its power cap (4 cycles = one
  wave64 instruction on a 16-lane SIMD = every issue slot used).
usage: summarize_sq.py <pmc_sq counter_collection.csv> <out.json> [<kernel_trace_one_box_at_a_time.csv>]"""
import csv
import json
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
lone_ns = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        key = (r["Kernel_Name"], int(r["Grid_Size_X"]))
        ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        lone_ns[key] = min(lone_ns.get(key, 1e30), ns)
per_dispatch = defaultdict(dict)
meta = {}
for r in csv.DictReader(open(src)):
    d = r["Dispatch_Id"]
    per_dispatch[d][r["Counter_Name"]] = per_dispatch[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[d] = (r["Kernel_Name"], int(r["Grid_Size"]), int(r["VGPR_Count"]), int(r["Scratch_Size"]), int(r["LDS_Block_Size"]))
by_kernel = defaultdict(list)
for d, c in per_dispatch.items():
    by_kernel[meta[d][0]].append((meta[d], c))
doc = {"source": "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY "
                 "GRBM_GUI_ACTIVE -- bench.py n=65536 t=256 (tools/run_profiles.sh); kernels run one at a time under counter collection",
       "kernels": {}}
for k, lst in sorted(by_kernel.items()):
    if not k.startswith("k_"):
        continue
    gmax = max(m[1] for m, _ in lst)
    sel = [(m, c) for m, c in lst if m[1] == gmax]
    avg = lambda name: sum(c.get(name, 0.0) for _, c in sel) / len(sel)
    wc, av, wi, wa, waves, insts, gui = (avg(x) for x in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                                                         "SQ_WAVES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"))
    m = sel[0][0]
    doc["kernels"][k] = {
        "launches": len(sel), "grid": gmax, "vgprs": m[2], "scratch_bytes": m[3], "lds_bytes": m[4],
        "waves": waves, "valu_insts_per_wave": insts / waves if waves else None,
        "wave_cycles": wc, "active_valu": av, "wait_inst_any": wi, "wait_any": wa, "gui_active_cycles": gui,
        "active_valu_frac": av / wc if wc else None, "wait_inst_frac": wi / wc if wc else None, "wait_any_frac": wa / wc if wc else None,
        "simd_valu_util": av * 4 / (gui * 1024) if gui else None,
    }
    if (k, gmax) in lone_ns and waves and insts > 0:
        per_simd = insts / 1024.0                      # VALU wave-instructions per SIMD and launch
        ns = lone_ns[(k, gmax)]
        doc["kernels"][k].update({
            "isolated_launch_ms": ns / 1e6, "valu_insts_per_simd": per_simd,
            "ns_per_valu_inst_per_simd": ns / per_simd,
            "cycles_per_valu_inst_at_2p4GHz": ns / per_simd * 2.4, "cycles_per_valu_inst_at_2p1GHz": ns / per_simd * 2.1,
            "issue_slot_use_at_2p1GHz": 4.0 / (ns / per_simd * 2.1)})
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: {x: (round(v, 3) if isinstance(v, float) else v) for x, v in d.items() if x in ("launches", "grid", "vgprs", "valu_insts_per_wave", "active_valu_frac", "wait_inst_frac", "wait_any_frac", "isolated_launch_ms", "cycles_per_valu_inst_at_2p1GHz", "issue_slot_use_at_2p1GHz")} for k, d in doc["kernels"].items()}, indent=1))
