"""GPU engine (through the C ABI) against the committed golden fixtures: no oracle in the loop."""
import glob
import json
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
MODP = sorted(glob.glob(os.path.join(HERE, "golden", "modp2048_*.json")))


@pytest.mark.parametrize("path", MODP, ids=os.path.basename)
def test_modp_fixture_on_gpu(engine, path):
    fx = json.load(open(path))
    b = fx["box"]
    cat = lambda hs: bytes.fromhex("".join(hs))
    res = engine.verify_distribution(cat(b["commitments"]), b["positions"], cat(b["publickeys"]), cat(b["shares"]),
                                     cat(b["responses"]), bytes.fromhex(b["challenge"]), dump=True)
    assert res["verdict"] is True
    assert res["digest"].hex() == fx["expected"]["transcript_digest"]
    assert res["X"].hex() == "".join(fx["expected"]["X"])
    assert res["a1"].hex() == "".join(fx["expected"]["a1"])
    assert res["a2"].hex() == "".join(fx["expected"]["a2"])
    for tam in fx["tampered"]:
        res = engine.verify_distribution(cat(tam["commitments"]), b["positions"], cat(b["publickeys"]),
                                         cat(tam["shares"]), cat(tam["responses"]), bytes.fromhex(tam["challenge"]))
        assert res["verdict"] is False
        assert res["digest"].hex() == tam["transcript_digest"]
    # share boxes: verify_share batch
    sb = fx["expected"]["share_boxes"]
    verdicts = engine.verify_shares(cat(b["publickeys"]), cat([s["share"] for s in sb]), cat(b["shares"]),
                                    cat([s["challenge"] for s in sb]), cat([s["response"] for s in sb]))
    assert list(verdicts) == [1] * fx["n"]
    # dealer side: commitments from coefficients, Y/a1/a2 from recorded randomness
    q1 = int("ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404ddef9519b3cd"
             "3a431b302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7edee386bfb5a899fa5ae"
             "9f24117c4b1fe649286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f35620"
             "8552bb9ed529077096966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3be39e772c180e86039b2783a2ec07a28fb5"
             "c55df06f4c52c9de2bcbf6955817183995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff", 16) - 1
    coeffs = [int(c, 16) for c in fx["inputs"]["coefficients"]]
    fx256 = lambda v: v.to_bytes(256, "big")
    cm = engine.batch_exp_fixed_base(fx256(4), b"".join(map(fx256, coeffs)))
    assert cm.hex() == "".join(b["commitments"])
    pvals = [sum(c * i ** j for j, c in enumerate(coeffs)) % q1 for i in b["positions"]]
    ws = [int(w, 16) for w in fx["inputs"]["witnesses"]]
    d = engine.distribute(cm, b["positions"], cat(b["publickeys"]), b"".join(map(fx256, pvals)), b"".join(map(fx256, ws)))
    assert d["X"].hex() == "".join(fx["expected"]["X"]) and d["Y"].hex() == "".join(b["shares"])
    assert d["a1"].hex() == "".join(fx["expected"]["a1"]) and d["a2"].hex() == "".join(fx["expected"]["a2"])
    assert d["digest"].hex() == fx["expected"]["transcript_digest"]


def test_serialized_boxes_verify_from_the_wire_format(engine):
    """mpvss_box_verify_wire: a box serialized to "MPVSSBX1" (any of the three groups) verifies from its bytes with the
    fixture's transcript digest; the tampered variants are rejected with theirs."""
    import glob

    from mpvss_rs_amd import capi
    gid_of = {"modp2048": 0, "secp256k1": capi.GROUP_SECP256K1, "ristretto255": capi.GROUP_RISTRETTO255}
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.json")))
    assert len(paths) >= 6
    for path in paths:
        fx = json.load(open(path))
        b = fx["box"]
        cat = lambda hs: bytes.fromhex("".join(hs))
        gid = gid_of[fx["group"]]
        wire = capi.box_serialize(gid, cat(b["commitments"]), b["positions"], cat(b["publickeys"]), cat(b["shares"]),
                                  cat(b["responses"]), bytes.fromhex(b["challenge"]), b"\x01\x02")
        verdict, digest = engine.verify_wire(wire)
        assert verdict is True and digest.hex() == fx["expected"]["transcript_digest"], path
        for tam in fx["tampered"]:
            wire = capi.box_serialize(gid, cat(tam["commitments"]), b["positions"], cat(b["publickeys"]), cat(tam["shares"]),
                                      cat(tam["responses"]), bytes.fromhex(tam["challenge"]))
            verdict, digest = engine.verify_wire(wire)
            assert verdict is False and digest.hex() == tam["transcript_digest"], path
    with pytest.raises(capi.EngineError):
        engine.verify_wire(wire[:-3])
