"""Shared helpers for the parity tests: seeded synthetic PVSS instances built with the oracle."""
import math
import random

import mpvss_oracle as O

EB = 256


def modp_keygen(g, rng):
    """modp.rs:162-174: uniform below q with gcd(k, q-1) == 1."""
    while True:
        k = rng.randrange(g.q)
        if math.gcd(k, g.q - 1) == 1:
            return k


def make_modp_instance(n, t, seed, secret=0x48656C6C6F):
    g = O.ModpGroup()
    rng = random.Random(seed)
    privs = [modp_keygen(g, rng) for _ in range(n)]
    pks = [g.generate_public_key(k) for k in privs]
    coeffs = [rng.randrange(g.q - 1) for _ in range(t)]
    ws = [modp_keygen(g, rng) for _ in range(n)]
    box = O.distribute_secret(g, secret, pks, t, coeffs, ws)
    return g, privs, pks, coeffs, ws, box


def cat(g, elems):
    return b"".join(g.element_to_fixed(e) for e in elems)


def split(b):
    return [int.from_bytes(b[i:i + EB], "big") for i in range(0, len(b), EB)]
