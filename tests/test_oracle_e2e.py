"""End-to-end protocol runs on the oracle, mirroring the reference's own tests
(tests/mpvss_tests.rs:10-87, src/participant.rs:593-743, examples/mpvss_all*.rs) for all three groups."""
import math
import random

import pytest

import mpvss_oracle as O


def _keygen(G, rng):
    if G.name == "modp2048":
        while True:
            k = rng.randrange(G.q)
            if math.gcd(k, G.q - 1) == 1:
                return k
    return rng.randrange(1 << 256) % G.group_order_int()


@pytest.mark.parametrize("name", ["modp2048", "secp256k1", "ristretto255"])
def test_full_protocol_n4_t3(name):
    G = O.GROUPS[name]()
    rng = random.Random(hash(name) & 0xFFFF)
    n, t = 4, 3
    privs = [_keygen(G, rng) for _ in range(n)]
    pks = [G.generate_public_key(k) for k in privs]
    coeffs = [rng.randrange(G.group_order_int()) for _ in range(t)]
    ws = [_keygen(G, rng) for _ in range(n)]
    secret = O.string_to_secret("Hello MPVSS Example.")
    box = O.distribute_secret(G, secret, pks, t, coeffs, ws)
    assert O.verify_distribution_shares(G, box)
    w = _keygen(G, rng)
    sbs = [O.extract_secret_share(G, box, k, w) for k in privs]
    for sb, pk in zip(sbs, pks):
        assert O.verify_share(G, sb, box, pk)
    assert O.string_from_secret(O.reconstruct(G, sbs[:3], box)) == "Hello MPVSS Example."
    assert O.string_from_secret(O.reconstruct(G, [sbs[0], sbs[2], sbs[3]], box)) == "Hello MPVSS Example."
    assert O.reconstruct(G, sbs[:2], box) is None                       # fewer than t shares
    # negative cases the reference never tests
    key = G.element_to_bytes(pks[1])
    bad = dict(box, responses=dict(box["responses"]))
    bad["responses"][key] ^= 1
    assert not O.verify_distribution_shares(G, bad)
    bad_sb = dict(sbs[0], response=sbs[0]["response"] ^ 1)
    assert not O.verify_share(G, bad_sb, box, pks[0])
    missing = dict(box, shares={k: v for k, v in box["shares"].items() if k != key})
    assert not O.verify_distribution_shares(G, missing)                 # participant.rs:415-420


def test_threshold_2_positions_1_and_3_modp():
    # participant.rs:703-743 regression
    G = O.ModpGroup()
    rng = random.Random(99)
    privs = [_keygen(G, rng) for _ in range(3)]
    pks = [G.generate_public_key(k) for k in privs]
    box = O.distribute_secret(G, 123456, pks, 2, [rng.randrange(G.q - 1) for _ in range(2)],
                              [_keygen(G, rng) for _ in range(3)])
    w = rng.randrange(G.q)
    s1 = O.extract_secret_share(G, box, privs[0], w)
    s3 = O.extract_secret_share(G, box, privs[2], w)
    assert O.reconstruct(G, [s1, s3], box) == 123456


def test_threshold_greater_than_n_panics():
    G = O.ModpGroup()
    with pytest.raises(AssertionError):
        O.distribute_secret(G, 1, [4], 2, [1, 2], [3])                  # participant.rs:166
