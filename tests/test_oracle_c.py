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


# ---- oracle/ec_ref.c: the curve groups' per-share work in plain C against the Python oracle ----------------------
import pytest  # noqa: E402

from ec_ref import GROUP_ID, EcRef  # noqa: E402
from helpers import ec_reference_share  # noqa: E402


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_c_curve_exp_matches_python_oracle_and_published_multiples(name):
    R = EcRef()
    Gc = O.GROUPS[name]()
    gid = GROUP_ID[name]
    order = Gc.group_order_int()
    gen = Gc.element_to_bytes(Gc.generator())
    rng = random.Random(3)
    for k in [0, 1, 2, 3, 5, order - 1, order - 2, 1 << 255 if name == "secp256k1" else (1 << 252), rng.randrange(order), rng.randrange(order)]:
        k %= order
        want = Gc.element_to_bytes(Gc.exp(Gc.generator(), k))
        assert R.exp(gid, gen, Gc.scalar_to_fixed(k)) == want, k
    # SURVEY appendix B known answers (SEC2 / RFC 9496 A.1 multiples of the generator)
    if name == "secp256k1":
        assert R.exp(gid, gen, (2).to_bytes(32, "big")).hex() == "02c6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5"
        assert R.exp(gid, gen, (3).to_bytes(32, "big")).hex() == "02f9308a019258c31049344f85f89d5229b531c845836f99b08601f113bce036f9"
    else:
        assert R.exp(gid, gen, (2).to_bytes(32, "little")).hex() == "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919"
        assert R.exp(gid, gen, (5).to_bytes(32, "little")).hex() == "e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e"
    # a point that is not the generator
    P = Gc.exp(Gc.generator(), rng.randrange(order))
    k = rng.randrange(order)
    assert R.exp(gid, Gc.element_to_bytes(P), Gc.scalar_to_fixed(k)) == Gc.element_to_bytes(Gc.exp(P, k))


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_c_curve_share_work_matches_python_oracle(name):
    """X_i, a1_i, a2_i of a share (reference operation order) -- honest values, a duplicated commitment, the identity
    as a commitment and as a share, position 0 and a large position."""
    R = EcRef()
    Gc = O.GROUPS[name]()
    gid = GROUP_ID[name]
    order = Gc.group_order_int()
    rng = random.Random(4)
    L = Gc.elem_len
    t = 5
    cm = [Gc.exp(Gc.generator(), rng.randrange(order)) for _ in range(t)]
    cm[3] = cm[2]
    enc = b"".join(Gc.element_to_bytes(c) for c in cm)
    ident = Gc.element_to_bytes(Gc.identity())
    enc_id = enc[:L] + ident + enc[2 * L:]
    y = Gc.element_to_bytes(Gc.exp(Gc.generator(), rng.randrange(order)))
    Y = Gc.element_to_bytes(Gc.exp(Gc.generator(), rng.randrange(order)))
    for cmb, pos, yy, YY in [(enc, 1, y, Y), (enc, 65536, y, Y), (enc_id, 7, y, ident), (enc, 0, y, y), (enc, (1 << 62) + 3, Y, y)]:
        r, c = Gc.scalar_to_fixed(rng.randrange(order)), Gc.scalar_to_fixed(rng.randrange(order))
        assert R.share_work(gid, cmb, pos, yy, YY, r, c) == ec_reference_share((name, cmb, pos, yy, YY, r, c)), (pos,)
    with pytest.raises(ValueError):
        R.share_work(gid, enc, 1, bytes([5]) + bytes(L - 1), Y, Gc.scalar_to_fixed(1), Gc.scalar_to_fixed(1))


def test_openssl_secp256k1_baseline_follows_the_reference_sequence():
    """oracle/openssl_ref.py::OpenSslSecpRef (bench.py's strong-CPU line for secp256k1: libcrypto's EC_POINT_mul in the place of
    k256) against the Python oracle in the reference order, incl. the identity as a commitment and position 0."""
    import random
    import mpvss_oracle as O
    import openssl_ref
    if not openssl_ref.ec_available():
        pytest.skip("libcrypto without secp256k1")
    G = O.GROUPS["secp256k1"]()
    n = G.group_order_int()
    rng = random.Random(31)
    B = G.generator()
    e, s = G.element_to_bytes, G.scalar_to_bytes
    ref = openssl_ref.OpenSslSecpRef()
    for t in (1, 4):
        cm = [G.exp(B, rng.randrange(n)) for _ in range(t)]
        if t == 4:
            cm[2] = G.identity()
        y, Y = G.exp(B, rng.randrange(1, n)), G.exp(B, rng.randrange(1, n))
        r, c = rng.randrange(n), rng.randrange(n)
        for pos in (0, 1, 9, 65536):
            X = O.commitment_eval(G, cm, pos)
            a1, a2 = O.dleq_verifier_commitments(G, G.subgroup_generator(), X, y, Y, r, c)
            got = ref.share_work(b"".join(map(e, cm)), pos, e(y), e(Y), s(r), s(c))
            assert got == (e(X), e(a1), e(a2)), (t, pos)
