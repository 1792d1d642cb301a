#!/usr/bin/env python3
"""Runs tools/mfma_mont/ubench on the GPU box and checks both kernels' outputs against Python integers:
after S Montgomery squarings x_S = x_0^(2^S) * R^-(2^S - 1) mod N (outputs are < 2N, compared mod N)."""
import os
import random
import struct
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from model import L, N, R, W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out_dir = sys.argv[3] if len(sys.argv) > 3 else "/tmp"
rng = random.Random(12345)
xs = [rng.randrange(2 * N) for _ in range(n)]
xs[0], xs[1], xs[2] = 2 * N - 1, 0, 1
M29 = (1 << W) - 1
with open(os.path.join(out_dir, "in.bin"), "wb") as f:
    for x in xs:
        f.write(struct.pack("<72I", *[(x >> (W * k)) & M29 for k in range(L)]))
exe = os.environ.get("UBENCH", os.path.join(HERE, "ubench"))
res = subprocess.run([exe, os.path.join(out_dir, "in.bin"), os.path.join(out_dir, "out"), str(n), str(S), "3"],
                     capture_output=True, text=True)
print(res.stdout, res.stderr)
if res.returncode != 0:
    sys.exit(res.returncode)
Rinv = pow(R, -1, N)
e = pow(2, S)
check = sorted(set([0, 1, 2, 3, 31, 32, 33, n - 1] + [rng.randrange(n) for _ in range(40)]))
for kind in ("pair", "quad", "pairmul"):
    raw = open(os.path.join(out_dir, f"out.{kind}.bin"), "rb").read()
    bad = 0
    for i in check:
        limbs = struct.unpack_from("<72I", raw, 288 * i)
        v = sum(l << (W * k) for k, l in enumerate(limbs))
        want = pow(xs[i], e, N) * pow(Rinv, e - 1, N) % N if kind != "pairmul" else pow(xs[i], S + 1, N) * pow(Rinv, S, N) % N
        if v % N != want or v >= 2 * N or max(limbs) > M29 + 512:
            bad += 1
            if bad <= 3:
                print(f"  {kind}: number {i} WRONG (v < 2N: {v < 2 * N}, max limb {max(limbs):#x})")
    print(f"{kind}: {len(check) - bad} / {len(check)} sampled numbers exact")
    if bad:
        sys.exit(4)
