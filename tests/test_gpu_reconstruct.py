"""Batched reconstruct for the three groups against the oracle: G^s, the mask and the recovered
secret from several subsets of shares (src/participant.rs:462-561, 1452-1557, 1895-2002)."""
import random

import pytest

import mpvss_oracle as O
from helpers import modp_keygen
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu
GID = {"modp2048": 0, "secp256k1": capi.GROUP_SECP256K1, "ristretto255": capi.GROUP_RISTRETTO255}


def instance(name, n, t, seed, secret):
    G = O.GROUPS[name]()
    rng = random.Random(seed)
    order = G.group_order_int()
    if name == "modp2048":
        privs = [modp_keygen(G, rng) for _ in range(n)]
        wits = [modp_keygen(G, rng) for _ in range(2 * n)]
    else:
        privs = [rng.randrange(1, order) for _ in range(n)]
        wits = [rng.randrange(1, order) for _ in range(2 * n)]
    pks = [G.generate_public_key(k) for k in privs]
    coeffs = [rng.randrange(order) for _ in range(t)]
    box = O.distribute_secret(G, secret, pks, t, coeffs, wits[:n])
    sbs = [O.extract_secret_share(G, box, k, w) for k, w in zip(privs, wits[n:])]
    return G, pks, coeffs, box, sbs


@pytest.mark.parametrize("name", ["modp2048", "secp256k1", "ristretto255"])
def test_reconstruct_matches_oracle(engine, name):
    n, t, secret = 9, 4, 0x48656C6C6F204D50565353
    G, pks, coeffs, box, sbs = instance(name, n, t, 31, secret)
    gid = GID[name]
    keys = [G.element_to_bytes(p) for p in pks]
    pos_of = [box["positions"][k] for k in keys]
    efix = G.element_to_fixed
    # G^s: s = P(0) (MODP reduces it, the curves convert the coefficient, participant.rs:267-268 / 1234-1242)
    s0 = coeffs[0] % G.group_order_int() if name == "modp2048" else G.scalar_from_bigint(coeffs[0])
    gs_want = efix(G.exp(G.generator(), s0))
    for subset in ([0, 1, 2, 3], [8, 6, 3, 1], list(range(9)), [2, 4, 5, 7, 8], [0, 8, 4, 2, 6]):
        chosen = [sbs[i] for i in subset]
        want = O.reconstruct(G, chosen, box)
        assert want == secret
        positions = [pos_of[i] for i in subset]
        shares = b"".join(efix(sb["share"]) for sb in chosen)
        if gid == 0:
            gs, mask = engine.reconstruct(positions, shares)
        else:
            gs, mask = engine.ec_reconstruct(gid, positions, shares)
        assert gs == gs_want
        assert int.from_bytes(mask, "big") ^ box["U"] == want
        assert int.from_bytes(mask, "big") == G.secret_mask(G.exp(G.generator(), s0))
    # a wrong share gives the wrong secret -- the same wrong secret as the oracle's
    chosen = [dict(sb) for sb in sbs[:4]]
    chosen[2]["share"] = sbs[5]["share"]
    want = O.reconstruct(G, chosen, box)
    assert want != secret
    positions = [pos_of[i] for i in range(4)]
    shares = b"".join(efix(sb["share"]) for sb in chosen)
    gs, mask = engine.reconstruct(positions, shares) if gid == 0 else engine.ec_reconstruct(gid, positions, shares)
    assert int.from_bytes(mask, "big") ^ box["U"] == want
    # duplicate positions are rejected
    with pytest.raises(capi.EngineError):
        if gid == 0:
            engine.reconstruct([1, 2, 2, 3], shares)
        else:
            engine.ec_reconstruct(gid, [1, 2, 2, 3], shares)


def test_reconstruct_large_threshold_modp(engine):
    """t = 300 shares out of positions up to 600: the scalar-field Lagrange path (no O(t^2)-bit integers) and the
    product tree; checked through G^s = 2^P(0) for a polynomial the test knows."""
    G = O.ModpGroup()
    rng = random.Random(77)
    order = G.group_order_int()
    t = 300
    coeffs = [rng.randrange(order) for _ in range(t)]
    positions = sorted(rng.sample(range(1, 601), t))
    pv = capi.poly_eval(0, b"".join(c.to_bytes(256, "big") for c in coeffs), positions)
    shares = engine.batch_exp_fixed_base((2).to_bytes(256, "big"), pv)          # S_i = G^P(i)
    gs, mask = engine.reconstruct(positions, shares)
    assert int.from_bytes(gs, "big") == pow(2, coeffs[0], G.q)
