"""The plain-C restatement (oracle/modp_ref.c, also the bench's CPU baseline) against the Python oracle."""
import random

import mpvss_oracle as O
from helpers import make_modp_instance
from modp_ref import ModpRef

G = O.ModpGroup()
Q = G.q


def test_c_modpow_and_mulmod():
    R = ModpRef()
    rng = random.Random(1)
    for _ in range(12):
        a, e, b = (rng.randrange(1 << 2048) for _ in range(3))
        assert R.modpow(a, e) == pow(a, e, Q)
        assert R.mulmod(a, b) == a * b % Q
    for a, e in [(0, 0), (0, 5), (Q, 3), (Q + 1, 7), (2, 0), (5, 1), ((1 << 2048) - 1, (1 << 2048) - 1), (4, Q - 1)]:
        assert R.modpow(a, e) == pow(a, e, Q)
    assert R.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"


def test_c_verify_distribution_matches_python_oracle():
    R = ModpRef()
    g, privs, pks, coeffs, ws, box = make_modp_instance(5, 3, 7)
    flat = O.box_to_flat(g, box)
    res = R.verify_distribution(flat, dump=True)
    trace = {}
    assert O.verify_distribution_shares(g, box, trace)
    assert res["verdict"] is True and res["digest"] == trace["digest"]
    assert [int.from_bytes(res["X"][i * 256:(i + 1) * 256], "big") for i in range(5)] == trace["X"]
    assert [int.from_bytes(res["a1"][i * 256:(i + 1) * 256], "big") for i in range(5)] == trace["a1"]
    assert [int.from_bytes(res["a2"][i * 256:(i + 1) * 256], "big") for i in range(5)] == trace["a2"]
    bad = dict(flat)
    buf = bytearray(flat["responses"]); buf[100] ^= 4
    bad["responses"] = bytes(buf)
    assert R.verify_distribution(bad)["verdict"] is False
