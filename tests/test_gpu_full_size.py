"""Full-size GPU checks through size-independent properties: an honest box produced by the engine's own dealer
path must verify with the dealer's transcript digest; one flipped bit anywhere must be rejected; a sample of
shares must equal the C restatement (reference operation order).  The BASELINE configs at their full sizes are in
tests/test_gpu_configs.py; this file keeps a quick mid-size case and the homomorphic property."""
import hashlib
import math
import random

import pytest

pytestmark = pytest.mark.gpu
EB = 256
Q = int("ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404ddef9519b3cd3a431b"
        "302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7edee386bfb5a899fa5ae9f24117c4b1fe6"
        "49286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb9ed529077096966d"
        "670c354e4abc9804f1746c08ca18217c32905e462e36ce3be39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf695581718"
        "3995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff", 16)
ORDER = Q - 1


def fx(v):
    return v.to_bytes(EB, "big")


def keygen(rng):
    while True:
        k = rng.randrange(Q)
        if math.gcd(k, ORDER) == 1:
            return k


def make_box(engine, n, t, seed):
    rng = random.Random(seed)
    coeffs = [rng.randrange(ORDER) for _ in range(t)]
    privs = [keygen(rng) for _ in range(n)]
    wits = [keygen(rng) for _ in range(n)]
    positions = list(range(1, n + 1))
    rc = list(reversed(coeffs))
    pvals = []
    for i in positions:
        acc = 0
        for a in rc:
            acc = acc * i + a
        pvals.append(acc % ORDER)
    cm = engine.batch_exp_fixed_base(fx(4), b"".join(map(fx, coeffs)))
    pks = engine.batch_exp_fixed_base(fx(2), b"".join(map(fx, privs)))
    d = engine.distribute(cm, positions, pks, b"".join(map(fx, pvals)), b"".join(map(fx, wits)))
    c = int.from_bytes(hashlib.sha256(d["digest"]).digest(), "big") % ((Q - 1) // 2)
    responses = b"".join(fx((w - p * c) % ORDER) for w, p in zip(wits, pvals))
    return {"cm": cm, "pos": positions, "pk": pks, "Y": d["Y"], "r": responses, "c": fx(c), "d": d}


@pytest.mark.parametrize("n,t,sample", [(2048, 32, 6)])
def test_round_trip_and_tamper(engine, n, t, sample):
    box = make_box(engine, n, t, seed=n + t)
    res = engine.verify_distribution(box["cm"], box["pos"], box["pk"], box["Y"], box["r"], box["c"], dump=True)
    assert res["verdict"] is True
    assert res["digest"] == box["d"]["digest"]            # verifier transcript == dealer transcript
    assert res["X"] == box["d"]["X"] and res["a1"] == box["d"]["a1"] and res["a2"] == box["d"]["a2"]
    rng = random.Random(5)
    for field in ("r", "Y", "cm"):
        buf = bytearray(box[field])
        buf[rng.randrange(len(buf))] ^= 1 << rng.randrange(8)
        args = dict(box)
        args[field] = bytes(buf)
        bad = engine.verify_distribution(args["cm"], args["pos"], args["pk"], args["Y"], args["r"], args["c"])
        assert bad["verdict"] is False and bad["digest"] != box["d"]["digest"]
    if sample:
        from modp_ref import ModpRef
        ref = ModpRef()
        for i in [int((j + 0.5) * n / sample) for j in range(sample)]:
            s = slice(i * EB, (i + 1) * EB)
            x, a1, a2 = ref.share_work(box["cm"], box["pos"][i], box["pk"][s], box["Y"][s], box["r"][s], box["c"])
            assert (x, a1, a2) == (res["X"][s], res["a1"][s], res["a2"][s])


def test_homomorphic_property_of_commit_eval(engine):
    """X_i(C * C') == X_i(C) * X_i(C') mod q (commitment evaluation is multiplicative in the commitments)."""
    rng = random.Random(77)
    t, n = 9, 300
    c1 = [pow(4, rng.randrange(ORDER), Q) for _ in range(t)]
    c2 = [pow(4, rng.randrange(ORDER), Q) for _ in range(t)]
    pos = [rng.randrange(1, 1 << 20) for _ in range(n)]
    x1 = engine.commit_eval(b"".join(map(fx, c1)), pos)
    x2 = engine.commit_eval(b"".join(map(fx, c2)), pos)
    x12 = engine.commit_eval(b"".join(fx(a * b % Q) for a, b in zip(c1, c2)), pos)
    assert engine.batch_mul(x1, x2) == x12
