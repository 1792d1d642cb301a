"""The curve groups' dealer END TO END on the device: P(i) mod n / l and r_i = w_i - P(i) c as kernels (one share per lane,
Montgomery arithmetic mod the group order), mpvss_ec_deal_compute and mpvss_ec_deal -- against the golden fixtures, the oracle's
distribute_secret (src/participant.rs:1094-1274 secp256k1, 1573-1717 ristretto255; P(i) `% n` :1155-1157 / :1619-1621,
responses :1200-1230 / :1662-1690) and the host functions of the same ABI."""
import json
import os
import random

import pytest
import torch

import mpvss_oracle as O
from mpvss_rs_amd import capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GID = {"secp256k1": capi.GROUP_SECP256K1, "ristretto255": capi.GROUP_RISTRETTO255}


def dev_u8(b):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).to("cuda:0")


def dbytes(t):
    return bytes(t.cpu().numpy().tobytes())


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
@pytest.mark.parametrize("shape", ["n3_t3", "n8_t4"])
def test_ec_deal_reproduces_the_golden_boxes(engine, name, shape):
    fx = json.load(open(os.path.join(HERE, "golden", f"{name}_{shape}.json")))
    G, gid = O.GROUPS[name](), GID[name]
    sb = G.scalar_to_fixed
    cat = lambda hs: bytes.fromhex("".join(hs))
    b = fx["box"]
    coeffs = b"".join(sb(int(c, 16)) for c in fx["inputs"]["coefficients"])
    wits = b"".join(sb(int(x, 16)) for x in fx["inputs"]["witnesses"])
    d = engine.ec_deal(gid, coeffs, b["positions"], cat(b["publickeys"]), wits)
    assert d["X"] == cat(fx["expected"]["X"]) and d["a1"] == cat(fx["expected"]["a1"]) and d["a2"] == cat(fx["expected"]["a2"])
    assert d["Y"] == cat(b["shares"]) and d["digest"] == bytes.fromhex(fx["expected"]["transcript_digest"])
    assert d["challenge"] == bytes.fromhex(b["challenge"]) and d["responses"] == cat(b["responses"])     # (encoded scalars in the fixture)


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_scalar_ring_on_the_device(engine, name):
    """P(i) and the responses against Python integers and the host functions: positions 0, 1, 2^63 - 1, a negative one (it
    enters as `position as u64`), coefficients and operands at 0, 1, order - 1 and NOT reduced (order, 2^256 - 1)."""
    G, gid = O.GROUPS[name](), GID[name]
    order = G.group_order_int()
    be = name == "secp256k1"
    sb = lambda v: v.to_bytes(32, "big" if be else "little")
    rng = random.Random(0xEC5CA1)
    edge = [0, 1, order - 1, order, order + 1, (1 << 256) - 1, 1 << 255, (1 << 252) - 1]
    for t in (1, 2, 7, 64):
        coeffs = [rng.randrange(1 << 256) for _ in range(t)]
        for k, v in enumerate(edge[:t]):
            coeffs[-1 - k] = v
        pos = list(range(1, 130)) + [0, 2**63 - 1, -5, 65536, 65535, (1 << 40) + 7]
        n = len(pos)
        d_pos = torch.tensor(pos, dtype=torch.int64, device="cuda:0")
        out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        cb = b"".join(map(sb, coeffs))
        engine.ec_poly_eval_device(gid, cb, d_pos.data_ptr(), n, out.data_ptr())
        want = [sum(c * pow(p % (1 << 64), j, order) for j, c in enumerate(coeffs)) % order for p in pos]
        assert dbytes(out) == b"".join(map(sb, want)), (name, t)
        assert dbytes(out) == capi.poly_eval(gid, cb, pos), (name, t)
    n = 200
    w = [rng.randrange(1 << 256) for _ in range(n - len(edge))] + edge
    a = edge + [rng.randrange(1 << 256) for _ in range(n - len(edge))]
    for c in (0, 1, order - 1, rng.randrange(order), (1 << 256) - 1):
        out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        d_w, d_a = dev_u8(b"".join(map(sb, w))), dev_u8(b"".join(map(sb, a)))
        engine.ec_dleq_responses_device(gid, d_w.data_ptr(), d_a.data_ptr(), sb(c), n, out.data_ptr())
        assert dbytes(out) == b"".join(sb((x - y * c) % order) for x, y in zip(w, a)), (name, c)
        assert dbytes(out) == capi.dleq_responses(gid, b"".join(map(sb, w)), b"".join(map(sb, a)), sb(c)), (name, c)


@pytest.mark.parametrize("name", ["secp256k1", "ristretto255"])
def test_ec_deal_compute_blocks_and_one_call_deal_of_a_larger_box(engine, name):
    """A 2100-share box with t = 33: mpvss_ec_deal (one call, host buffers) equals the block form mpvss_ec_deal_compute +
    mpvss_ec_distribute_absorb + mpvss_ec_dleq_responses_device (two blocks in flight) and the synchronous dealer fed with
    Python's P(i); the box verifies with the dealer's digest; a sampled share equals the oracle in the reference order."""
    G, gid = O.GROUPS[name](), GID[name]
    order = G.group_order_int()
    be = name == "secp256k1"
    sb = lambda v: v.to_bytes(32, "big" if be else "little")
    L = G.elem_len
    rng = random.Random(0xDEA1 + len(name))
    n, t = 2100, 33
    coeffs = [rng.randrange(order) for _ in range(t)]
    privs = [rng.randrange(1, order) for _ in range(n)]
    wits = [rng.randrange(1, order) for _ in range(n)]
    pos = list(range(1, n + 1))
    cb, wb = b"".join(map(sb, coeffs)), b"".join(map(sb, wits))
    cm = engine.ec_batch_exp_generator(gid, cb)
    pks = engine.ec_batch_exp_generator(gid, b"".join(map(sb, privs)))
    pv = [sum(c * pow(p, j, order) for j, c in enumerate(coeffs)) % order for p in pos]
    want = engine.ec_distribute(gid, cm, pos, pks, b"".join(map(sb, pv)), wb)
    d = engine.ec_deal(gid, cb, pos, pks, wb)
    assert (d["X"], d["Y"], d["a1"], d["a2"], d["digest"]) == (want["X"], want["Y"], want["a1"], want["a2"], want["digest"])
    c = capi.ec_hash_to_scalar(gid, d["digest"])
    ci = int.from_bytes(c, "big" if be else "little")
    assert d["challenge"] == c and d["responses"] == b"".join(sb((w - p * ci) % order) for w, p in zip(wits, pv))
    res = engine.ec_verify_distribution(gid, cm, pos, pks, d["Y"], d["responses"], d["challenge"])
    assert res["verdict"] is True and res["digest"] == d["digest"]
    # block form, two in flight
    d_pos = torch.tensor(pos, dtype=torch.int64, device="cuda:0")
    d_pk, d_w = dev_u8(pks), dev_u8(wb)
    d_p = [torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(2):
        engine.ec_deal_compute(gid, cb, d_pos.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[k].data_ptr())
    for k in range(2):
        st, X, Y, a1, a2 = engine.ec_distribute_absorb(gid, capi.transcript_init(), n)
        assert (X, Y, a1, a2) == (want["X"], want["Y"], want["a1"], want["a2"])
        assert capi.ec_transcript_verdict(gid, st, bytes(32))[1] == want["digest"]
        assert dbytes(d_p[k]) == b"".join(map(sb, pv))
        d_r = torch.zeros(n * 32, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        engine.ec_dleq_responses_device(gid, d_w.data_ptr(), d_p[k].data_ptr(), c, n, d_r.data_ptr())
        assert dbytes(d_r) == d["responses"]
    # claimed blocks (mpvss_block_claim + mpvss_ec_block_absorb_claimed): two dealers' blocks, absorbed in the REVERSE order
    import ctypes as C
    cb2 = b"".join(map(sb, [rng.randrange(order) for _ in range(t)]))
    engine.ec_deal_compute(gid, cb, d_pos.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[0].data_ptr())
    t0 = engine.block_claim()
    engine.ec_deal_compute(gid, cb2, d_pos.data_ptr(), d_pk.data_ptr(), d_w.data_ptr(), n, d_p[1].data_ptr())
    t1 = engine.block_claim()
    digests = {}
    for tk in (t1, t0):
        st = (C.c_uint8 * capi.TRANSCRIPT_STATE_BYTES).from_buffer_copy(capi.transcript_init())
        engine._check(engine.lib.mpvss_ec_block_absorb_claimed(engine.ctx, tk, st, None, None, None, None), "ec_block_absorb_claimed")
        digests[tk] = capi.ec_transcript_verdict(gid, bytes(st), bytes(32))[1]
    assert digests[t0] == want["digest"] and digests[t1] != want["digest"] and engine.blocks_in_flight() == (0, 0)
    # one share through the oracle in the reference order
    i = 1234
    cmx = [G.element_from_fixed(cm[k:k + L]) for k in range(0, len(cm), L)]
    X = O.commitment_eval(G, cmx, pos[i])
    assert G.element_to_bytes(X) == d["X"][i * L:(i + 1) * L]
    y = G.element_from_fixed(pks[i * L:(i + 1) * L])
    assert G.element_to_bytes(G.exp(y, pv[i])) == d["Y"][i * L:(i + 1) * L]
    with pytest.raises(capi.EngineError, match="threshold"):
        engine.ec_deal(gid, cb, pos[:5], pks[:5 * L], wb[:5 * 32])
