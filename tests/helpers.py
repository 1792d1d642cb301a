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


# ---- full-size boxes and parallel oracle work (tests/test_gpu_configs.py) ---------------------------------------
import concurrent.futures  # noqa: E402
import multiprocessing  # noqa: E402
import os  # noqa: E402

MODP_Q = O.ModpGroup().q
MODP_ORDER = MODP_Q - 1


def _poly_chunk(args):
    coeffs_rev, positions, order = args
    out = []
    for i in positions:
        acc = 0
        for a in coeffs_rev:
            acc = (acc * i + a) % order
        out.append(acc)
    return out


def worker_count(limit=40):
    try:
        return max(1, min(limit, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(limit, os.cpu_count() or 1))


def parallel_map(fn, items, procs=None):
    """fn(item) for every item on freshly spawned oracle-only worker processes (never forked from a process that
    holds a GPU context)."""
    items = list(items)
    procs = min(procs or worker_count(), max(1, len(items)))
    if procs == 1:
        return [fn(x) for x in items]
    ctx = multiprocessing.get_context("spawn")
    with concurrent.futures.ProcessPoolExecutor(max_workers=procs, mp_context=ctx) as ex:
        return list(ex.map(fn, items))


def poly_values(coeffs, positions, order):
    """P(i) mod order for every position (polynomial.rs:50-58 followed by the caller's `% order`,
    participant.rs:202); big shapes are cut over worker processes."""
    rc = list(reversed(coeffs))
    if len(coeffs) * len(positions) < (1 << 22):
        return _poly_chunk((rc, positions, order))
    w = worker_count()
    step = -(-len(positions) // (4 * w))
    chunks = [(rc, positions[k:k + step], order) for k in range(0, len(positions), step)]
    return [v for part in parallel_map(_poly_chunk, chunks) for v in part]


def ec_reference_share(args):
    """One share of verify_distribution_shares in the REFERENCE operation order (participant.rs:1404-1430 /
    1847-1873 via oracle/mpvss_oracle.py): returns the encodings of X_i, a1_i, a2_i."""
    name, cm_enc, position, y, Y, r, c = args
    G = O.GROUPS[name]()
    L = G.elem_len
    cm = [G.element_from_fixed(cm_enc[k:k + L]) for k in range(0, len(cm_enc), L)]
    X = O.commitment_eval(G, cm, position)
    a1, a2 = O.dleq_verifier_commitments(G, G.subgroup_generator(), X, G.element_from_fixed(y), G.element_from_fixed(Y),
                                         G.scalar_from_fixed(r), G.scalar_from_fixed(c))
    return G.element_to_bytes(X), G.element_to_bytes(a1), G.element_to_bytes(a2)


def ec_reference_x(args):
    name, cm_enc, position = args
    G = O.GROUPS[name]()
    L = G.elem_len
    cm = [G.element_from_fixed(cm_enc[k:k + L]) for k in range(0, len(cm_enc), L)]
    return G.element_to_bytes(O.commitment_eval(G, cm, position))


def modp_reference_x(args):
    cm, position = args
    G = O.ModpGroup()
    return O.commitment_eval(G, cm, position)


def modp_fast_share(args):
    """(X_i, a1_i, a2_i) of one MODP share as 256-byte strings by the FAST form of the reference's arithmetic: X_i by Horner's
    rule in the exponent, X = (..(C_{t-1}^i * C_{t-2})^i ..)^i * C_0 -- the same group element as the loop of
    src/participant.rs:423-434 (tests/test_oracle_reference_kats.py pins that equivalence against the reference-order
    oracle) -- and a1 = g^r X^c, a2 = y^r Y^c (src/dleq.rs:66-84) with CPython's pow.  A share costs ~0.1 s at t = 1024
    instead of ~30 core-seconds in the reference order, so a full 1 % sample of the largest boxes is affordable.
    args: (commitments bytes, position, y, Y, r, c) with 256-byte big-endian fields; picklable for parallel_map."""
    cm, i, y, Y, r, c = args
    q = MODP_Q
    t = len(cm) // 256
    x = int.from_bytes(cm[(t - 1) * 256:t * 256], "big") % q
    for j in range(t - 2, -1, -1):
        x = pow(x, i, q) * int.from_bytes(cm[j * 256:(j + 1) * 256], "big") % q
    yi, Yi, ri, ci = (int.from_bytes(v, "big") for v in (y, Y, r, c))
    a1 = pow(4, ri, q) * pow(x, ci, q) % q
    a2 = pow(yi, ri, q) * pow(Yi, ci, q) % q
    return tuple(v.to_bytes(256, "big") for v in (x, a1, a2))


def modp_dual_pow_chunk(items):
    """[(b1, e1, b2, e2), ..] -> [b1^e1 * b2^e2 mod q, ..] with CPython's pow (src/dleq.rs:66-84: two exp, one mul);
    picklable for parallel_map."""
    q = MODP_Q
    return [pow(b1, e1, q) * pow(b2, e2, q) % q for b1, e1, b2, e2 in items]
