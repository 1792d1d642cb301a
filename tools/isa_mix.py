#!/usr/bin/env python3
"""Instruction mix per Montgomery operation of the SHIPPED a2 kernel (k_modp_dual_exp_w6_pair), by issue class, from the ISA hipcc
emits for mpvss_rs_amd/csrc/modp_pair_kernels.hip:
    cd /tmp/x && hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 --save-temps -c .../modp_pair_kernels.hip
    python3 tools/isa_mix.py /tmp/x/modp_pair_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
The kernel's main loop is one Montgomery operation per iteration: a 2-iteration phase-A loop for a product, another for a squaring
(one of the two runs), and the reduction (straight line, shared).  dynamic(squaring) = loop body - both phase-A blocks + 2 x the
squaring block; dynamic(product) likewise.  Classes are what tools/ubench_valu.hip / mpvss_issue_probe price separately:
  mad64   v_mad_u64_u32, v_mad_i64_i32          shift64  v_lshl_add_u64, v_lshrrev_b64, v_lshlrev_b64, v_ashrrev_i64 (two passes)
  swap    v_permlane32_swap                     mov64    v_mov_b64
  alu32   every other VALU instruction          mfma     v_mfma_* (holds the issue for two slots)
bench.py's PAIR_MIX is this script's output."""
import collections
import json
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2] if len(sys.argv) > 2 else "k_modp_dual_exp_w6_pair"
i = txt.index(name + ":")
L = txt[i:txt.index("s_endpgm", i)].splitlines()
labels = {m.group(1): k for k, ln in enumerate(L) for m in [re.match(r"^(\.LBB\d+_\d+):", ln.strip())] if m}
loops = []
for k, ln in enumerate(L):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        loops.append((labels[m.group(1)], k))


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad64"
    if op.startswith(("v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_add_co", "v_addc_co")):
        return "shift64"
    if op.startswith("v_permlane32_swap"):
        return "swap"
    if op.startswith("v_mov_b64"):
        return "mov64"
    if op.startswith("v_"):
        return "alu32"
    if op.startswith("ds_"):
        return "lds"
    return None


def hist(a, b):
    c = collections.Counter()
    for ln in L[a:b + 1]:
        t = ln.strip().split()
        if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
            k = cls(t[0])
            if k:
                c[k] += 1
    return c


big = sorted((ab for ab in loops if ab[1] - ab[0] > 500), key=lambda ab: ab[1] - ab[0])
outer = big[-1]
inner = [ab for ab in big[:-1] if outer[0] <= ab[0] and ab[1] <= outer[1]]
assert len(inner) == 2, inner
h_out = hist(*outer)
h_in = [hist(*ab) for ab in inner]
sq = min(h_in, key=lambda h: h["mad64"])
mul = max(h_in, key=lambda h: h["mad64"])
rest = h_out - sq - mul
res = {"kernel": name, "numbers_per_wave": 32,
       "squaring": dict(rest + sq + sq), "product": dict(rest + mul + mul),
       "phase_a_block_squaring": dict(sq), "phase_a_block_product": dict(mul), "reduction_and_loop": dict(rest)}
print(json.dumps(res, indent=1))
