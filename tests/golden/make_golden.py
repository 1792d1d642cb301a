#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle with fixed seeds.

The reference (Rust, un-vendored arithmetic crates) cannot be built or imported in this image, so
these vectors are produced by oracle/mpvss_oracle.py -- itself pinned to the reference's own
known-answer tests and to RFC/SEC vectors in tests/test_oracle_reference_kats.py.  They freeze the
oracle's outputs so that a later change to either the oracle or the kernels is caught, and they
travel to the GPU box (the oracle does too, but fixtures also guard the oracle itself).

    python tests/golden/make_golden.py      # rewrites tests/golden/*.json
"""
import json
import math
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mpvss_oracle as O  # noqa: E402


def keygen(G, rng):
    if G.name == "modp2048":
        while True:
            k = rng.randrange(G.q)
            if math.gcd(k, G.q - 1) == 1:
                return k
    return rng.randrange(1 << 256) % G.group_order_int()


def hexs(G, elems):
    return [G.element_to_fixed(e).hex() for e in elems]


def make(name, n, t, seed, secret):
    G = O.GROUPS[name]()
    rng = random.Random(seed)
    privs = [keygen(G, rng) for _ in range(n)]
    pks = [G.generate_public_key(k) for k in privs]
    coeffs = [rng.randrange(G.group_order_int()) for _ in range(t)]
    ws = [keygen(G, rng) for _ in range(n)]
    box = O.distribute_secret(G, secret, pks, t, coeffs, ws)
    assert O.verify_distribution_shares(G, box)
    keys = [G.element_to_bytes(p) for p in pks]
    w_extract = keygen(G, rng)
    sbs = [O.extract_secret_share(G, box, k, w_extract) for k in privs]
    assert all(O.verify_share(G, sb, box, pk) for sb, pk in zip(sbs, pks))
    rec = O.reconstruct(G, sbs[:t], box)
    assert rec == secret
    fx = {
        "group": name, "n": n, "t": t, "seed": seed, "secret": hex(secret),
        "inputs": {
            "private_keys": [hex(k) for k in privs],
            "coefficients": [hex(c) for c in coeffs],
            "witnesses": [hex(w) for w in ws],
            "extract_witness": hex(w_extract),
        },
        "box": {
            "commitments": hexs(G, box["commitments"]),
            "positions": [box["positions"][k] for k in keys],
            "publickeys": hexs(G, pks),
            "shares": hexs(G, [box["shares"][k] for k in keys]),
            "responses": [G.scalar_to_fixed(box["responses"][k]).hex() for k in keys],
            "challenge": G.scalar_to_fixed(box["challenge"]).hex(),
            "U": hex(box["U"]),
        },
        "expected": {
            "X": hexs(G, box["_X"]), "a1": hexs(G, box["_a1"]), "a2": hexs(G, box["_a2"]),
            "transcript_digest": box["_digest"].hex(),
            "verify_distribution": True,
            "share_boxes": [{"share": G.element_to_fixed(sb["share"]).hex(),
                             "challenge": G.scalar_to_fixed(sb["challenge"]).hex(),
                             "response": G.scalar_to_fixed(sb["response"]).hex()} for sb in sbs],
            "verify_share": [True] * n,
            "reconstructed": hex(rec),
        },
        "tampered": [],
    }
    # tampered variants: one flipped bit -> expected verdict False and the digest the verifier computes
    for field, idx, bit in (("responses", n // 2, 3), ("shares", 0, 9), ("commitments", t - 1, 1), ("challenge", 0, 0)):
        bad = {k: (dict(v) if isinstance(v, dict) else v) for k, v in box.items()}
        if field == "responses":
            bad["responses"][keys[idx]] ^= 1 << bit
        elif field == "shares":
            e = bad["shares"][keys[idx]]
            if name == "modp2048":
                bad["shares"][keys[idx]] = e ^ (1 << bit)
            else:
                bad["shares"][keys[idx]] = G.mul(e, G.generator())
        elif field == "commitments":
            bad["commitments"] = list(box["commitments"])
            c = bad["commitments"][idx]
            bad["commitments"][idx] = (c ^ (1 << bit)) if name == "modp2048" else G.mul(c, G.generator())
        else:
            bad["challenge"] = box["challenge"] ^ (1 << bit)
        tr = {}
        verdict = O.verify_distribution_shares(G, bad, tr)
        assert verdict is False
        flat = O.box_to_flat(G, bad)
        fx["tampered"].append({
            "field": field, "index": idx,
            "commitments": [flat["commitments"][i * G.elem_len:(i + 1) * G.elem_len].hex() for i in range(t)],
            "shares": [flat["shares"][i * G.elem_len:(i + 1) * G.elem_len].hex() for i in range(n)],
            "responses": [flat["responses"][i * G.scalar_len:(i + 1) * G.scalar_len].hex() for i in range(n)],
            "challenge": flat["challenge"].hex(),
            "verify_distribution": verdict, "transcript_digest": tr["digest"].hex(),
        })
    return fx


def main():
    secret = O.string_to_secret("Hello MPVSS Example.")
    jobs = [("modp2048", 3, 3, 101), ("modp2048", 8, 4, 102), ("secp256k1", 3, 3, 201), ("secp256k1", 8, 4, 202),
            ("ristretto255", 3, 3, 301), ("ristretto255", 8, 4, 302)]
    for name, n, t, seed in jobs:
        fx = make(name, n, t, seed, secret)
        path = os.path.join(HERE, f"{name}_n{n}_t{t}.json")
        with open(path, "w") as f:
            json.dump(fx, f, indent=1)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
