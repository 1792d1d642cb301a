"""The oracle (and, for MODP, the C restatement) against the committed golden fixtures."""
import glob
import json
import os

import pytest

import mpvss_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def load_box(G, fx, override=None):
    b = dict(fx["box"])
    if override:
        b.update({k: override[k] for k in ("commitments", "shares", "responses", "challenge")})
    pks = [G.element_from_fixed(bytes.fromhex(h)) for h in fx["box"]["publickeys"]]
    keys = [G.element_to_bytes(p) for p in pks]
    return {
        "commitments": [G.element_from_fixed(bytes.fromhex(h)) for h in b["commitments"]],
        "publickeys": pks,
        "positions": dict(zip(keys, fx["box"]["positions"])),
        "shares": dict(zip(keys, [G.element_from_fixed(bytes.fromhex(h)) for h in b["shares"]])),
        "responses": dict(zip(keys, [G.scalar_from_fixed(bytes.fromhex(h)) for h in b["responses"]])),
        "challenge": G.scalar_from_fixed(bytes.fromhex(b["challenge"])),
        "U": int(fx["box"]["U"], 16),
    }


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_fixture(path):
    fx = json.load(open(path))
    G = O.GROUPS[fx["group"]]()
    box = load_box(G, fx)
    tr = {}
    assert O.verify_distribution_shares(G, box, tr) is fx["expected"]["verify_distribution"]
    assert tr["digest"].hex() == fx["expected"]["transcript_digest"]
    for name in ("X", "a1", "a2"):
        assert [G.element_to_fixed(e).hex() for e in tr[name]] == fx["expected"][name]
    # dealer side from the recorded randomness
    pks = box["publickeys"]
    coeffs = [int(c, 16) for c in fx["inputs"]["coefficients"]]
    ws = [int(w, 16) for w in fx["inputs"]["witnesses"]]
    again = O.distribute_secret(G, int(fx["secret"], 16), pks, fx["t"], coeffs, ws)
    assert [G.element_to_fixed(c).hex() for c in again["commitments"]] == fx["box"]["commitments"]
    assert G.scalar_to_fixed(again["challenge"]).hex() == fx["box"]["challenge"]
    assert hex(again["U"]) == fx["box"]["U"]
    # share boxes
    privs = [int(k, 16) for k in fx["inputs"]["private_keys"]]
    w = int(fx["inputs"]["extract_witness"], 16)
    sbs = []
    for k, exp in zip(privs, fx["expected"]["share_boxes"]):
        sb = O.extract_secret_share(G, box, k, w)
        assert G.element_to_fixed(sb["share"]).hex() == exp["share"]
        assert G.scalar_to_fixed(sb["challenge"]).hex() == exp["challenge"]
        assert G.scalar_to_fixed(sb["response"]).hex() == exp["response"]
        sbs.append(sb)
    assert [O.verify_share(G, sb, box, pk) for sb, pk in zip(sbs, pks)] == fx["expected"]["verify_share"]
    assert hex(O.reconstruct(G, sbs[: fx["t"]], box)) == fx["expected"]["reconstructed"]
    for tam in fx["tampered"]:
        tr = {}
        assert O.verify_distribution_shares(G, load_box(G, fx, tam), tr) is tam["verify_distribution"]
        assert tr["digest"].hex() == tam["transcript_digest"]


@pytest.mark.parametrize("path", [p for p in FIXTURES if "modp2048" in p], ids=os.path.basename)
def test_c_restatement_reproduces_modp_fixture(path):
    from modp_ref import ModpRef
    fx = json.load(open(path))
    R = ModpRef()
    for variant in [fx["box"]] + fx["tampered"]:
        flat = {"n": fx["n"], "t": fx["t"], "positions": fx["box"]["positions"],
                "commitments": bytes.fromhex("".join(variant["commitments"])),
                "publickeys": bytes.fromhex("".join(fx["box"]["publickeys"])),
                "shares": bytes.fromhex("".join(variant["shares"])),
                "responses": bytes.fromhex("".join(variant["responses"])),
                "challenge": bytes.fromhex(variant["challenge"])}
        res = R.verify_distribution(flat, dump=True)
        if variant is fx["box"]:
            assert res["verdict"] is True and res["digest"].hex() == fx["expected"]["transcript_digest"]
            assert res["X"].hex() == "".join(fx["expected"]["X"])
            assert res["a1"].hex() == "".join(fx["expected"]["a1"])
            assert res["a2"].hex() == "".join(fx["expected"]["a2"])
        else:
            assert res["verdict"] is False and res["digest"].hex() == variant["transcript_digest"]
