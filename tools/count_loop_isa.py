#!/usr/bin/env python3
"""Instructions of the longest loop of a kernel, by class, from the ISA hipcc emits (no GPU needed):
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEC_PART=1 -S --cuda-device-only -x hip mpvss_rs_amd/csrc/ec_kernels.hip -o /tmp/ec.s
  python3 tools/count_loop_isa.py /tmp/ec.s k_secp_fd_step_quad k_secp_fd_step_oct k_secp_fd_table_quad
A lone wave issues a v_mad_u64_u32 every 8.75 cycles and other VALU instructions every ~4.5 (profiles/r01_ubench_valu_issue_rates.txt);
LDS reads, ds_bpermute and v_cndmask cost it two to four VALU slots each -- the step time of the stage pipelines follows from these counts."""
import re
import sys

txt = open(sys.argv[1]).read()


def classify(ln):
    t = ln.strip().split()
    if not t or t[0].startswith(";") or t[0].endswith(":") or t[0].startswith("."):
        return None
    op = t[0]
    if op.startswith("v_mad_u64_u32"):
        return "mad64"
    if "dpp" in ln:
        return "dpp"
    if op.startswith("ds_bpermute"):
        return "bpermute"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    return "other"


for name in sys.argv[2:]:
    i = txt.index(name + ":")
    L = txt[i:txt.index("s_endpgm", i)].splitlines()
    labels = {m.group(1): k for k, ln in enumerate(L) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
    best = None
    for k, ln in enumerate(L):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and labels.get(m.group(1), 1 << 30) < k:
            body = L[labels[m.group(1)]:k]
            if best is None or len(body) > len(best):
                best = body
    c = {}
    for ln in best or []:
        k = classify(ln)
        if k:
            c[k] = c.get(k, 0) + 1
    print(f"{name}: longest loop {sum(c.values())} instructions", dict(sorted(c.items(), key=lambda kv: -kv[1])))
