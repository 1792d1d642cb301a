#!/usr/bin/env python3
"""VALU / MFMA / LDS instructions per Montgomery operation of the pair layout, counted in the ISA the compiler emits for
the microbenchmark chains (ubench.hip):  python3 count_isa.py <ubench-...-gfx950.s>
dynamic per operation = the chain's loop body, with the 36-row phase-A block counted twice (it is a 2-iteration loop)."""
import re
import sys

txt = open(sys.argv[1]).read()


def kernel(name):
    i = txt.index(name + ":")
    j = txt.index("s_endpgm", i)
    return txt[i:j].splitlines()


def count(lines):
    c = {"valu": 0, "mfma": 0, "lds": 0, "salu": 0, "vmem": 0}
    for ln in lines:
        t = ln.strip().split()
        if not t or t[0].startswith(";") or t[0].endswith(":") or t[0].startswith("."):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("global_") or op.startswith("scratch_"):
            c["vmem"] += 1
    return c


for nm, label in (("_Z12k_chain_pairILb0EEvPKjPjiiPKN2mm6TablesE", "squaring"), ("_Z12k_chain_pairILb1EEvPKjPjiiPKN2mm6TablesE", "product")):
    L = kernel(nm)
    labels = {}
    for k, ln in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", ln.strip())
        if m:
            labels[m.group(1)] = k
    loops = []
    for k, ln in enumerate(L):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loops.append((labels[m.group(1)], k))
    big = [ab for ab in loops if ab[1] - ab[0] > 500]
    inner = min(big, key=lambda ab: ab[1] - ab[0])
    outer = max(big, key=lambda ab: ab[1] - ab[0])
    ci, co = count(L[inner[0]:inner[1] + 1]), count(L[outer[0]:outer[1] + 1])
    dyn = {k: co[k] + ci[k] for k in co}
    print(f"{label}: phase-A block (runs twice) {ci}; loop body static {co}; per operation (32 numbers) {dyn}")
