"""Rate of the Montgomery-product site in isolation: batch_exp (k_modp_dual_exp with one table) on a grid that
fills the chip exactly (3 waves per SIMD = 49152 numbers) and on multiples of it.  Prints products/s against the
v_mad_u64_u32 issue peak used by bench.py."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (initialises HIP before the library, see tests/conftest.py)
from mpvss_rs_amd import Engine  # noqa: E402

PEAK = 2740642605.751602
eng = Engine(0)
rng = random.Random(1)
for n in (16384, 32768, 49152, 98304):
    bases = rng.randbytes(256 * n)
    bases = b"".join(b"\x7f" + bases[i * 256 + 1:(i + 1) * 256] for i in range(n))
    exps = rng.randbytes(256 * n)
    eng.batch_exp(bases, exps)
    best = 1e9
    for _ in range(3):
        eng.batch_exp(bases, exps)
        best = min(best, eng.kernel_ms(3))
    prods = n * (2044 + 511 + 1)
    print(f"n={n:6d} waves/SIMD={n / 16 / 1024:.2f} dual_exp {best:8.2f} ms  {prods / best / 1e6:8.1f} M products/s  "
          f"{prods / (best * 1e-3) / PEAK:.3f} of mad peak")
