"""The helpers the full-size GPU tests lean on (tests/helpers.py): polynomial values over worker processes and the
reference-order share evaluation on spawned oracle processes must agree with the plain oracle calls."""
import random

import mpvss_oracle as O
from helpers import MODP_ORDER, _poly_chunk, ec_reference_share, ec_reference_x, parallel_map, poly_values


def test_poly_values_in_worker_processes():
    rng = random.Random(1)
    coeffs = [rng.randrange(MODP_ORDER) for _ in range(70)]
    pos = list(range(1, 60001))                       # 4.2 M steps: takes the multi-process path
    vals = poly_values(coeffs, pos, MODP_ORDER)
    for i in (1, 777, 60000):
        assert vals[i - 1] == O.poly_get_value(coeffs, i) % MODP_ORDER      # polynomial.rs:50-58 + participant.rs:202
    assert vals[:50] == _poly_chunk((list(reversed(coeffs)), pos[:50], MODP_ORDER))


def test_reference_order_share_on_spawned_workers():
    rng = random.Random(2)
    for name in ("secp256k1", "ristretto255"):
        G = O.GROUPS[name]()
        order = G.group_order_int()
        L = G.elem_len
        cm = [G.generate_public_key(rng.randrange(order)) for _ in range(4)]
        enc = b"".join(G.element_to_bytes(c) for c in cm)
        y, Y = G.element_to_bytes(cm[0]), G.element_to_bytes(cm[1])
        r, c = rng.randrange(order), rng.randrange(order)
        out = parallel_map(ec_reference_share, [(name, enc, p, y, Y, G.scalar_to_fixed(r), G.scalar_to_fixed(c)) for p in (3, 4)],
                           procs=2)
        for p, (x, a1, a2) in zip((3, 4), out):
            X = O.commitment_eval(G, cm, p)
            e1, e2 = O.dleq_verifier_commitments(G, G.subgroup_generator(), X, cm[0], cm[1], r, c)
            assert (x, a1, a2) == tuple(G.element_to_bytes(e) for e in (X, e1, e2))
        ident = G.element_to_bytes(G.identity())
        doctored = enc[:L] + ident + enc[2 * L:]
        want = O.commitment_eval(G, [cm[0], G.identity(), cm[2], cm[3]], 9)
        assert parallel_map(ec_reference_x, [(name, doctored, 9)])[0] == G.element_to_bytes(want)
