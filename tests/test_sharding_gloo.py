"""world_size-2 CPU test (gloo) of the multi-GPU host logic: contiguous blocks, the running-hash state
carried rank to rank, verdict broadcast, and the all-gather of per-share verdicts.  The GPU engine is
replaced by an oracle-backed stand-in with the same block interface (tests may use the oracle)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBlockEngine:
    """Same block interface as mpvss_rs_amd.capi.Engine, computed by the CPU oracle."""

    def __init__(self):
        import mpvss_oracle as O
        self.O, self.G = O, O.ModpGroup()
        self._pending = b""

    def verify_block_compute(self, commitments, positions, pubkeys, shares, responses, challenge):
        O, G = self.O, self.G
        if any(p < 0 for p in positions):       # the engine's MPVSS_E_INVALID (the reference panics: negative exponent)
            from mpvss_rs_amd import capi
            raise capi.EngineError("verify_block_compute failed: rc=-1 negative position")
        sp = lambda b: [int.from_bytes(b[i:i + 256], "big") for i in range(0, len(b), 256)]
        cm, pk, sh, rs = sp(commitments), sp(pubkeys), sp(shares), sp(responses)
        c = int.from_bytes(challenge, "big")
        out = bytearray()
        for i, y, Y, r in zip(positions, pk, sh, rs):
            X = O.commitment_eval(G, cm, i)
            a1, a2 = O.dleq_verifier_commitments(G, G.subgroup_generator(), X, y, Y, r, c)
            for e in (X, Y, a1, a2):
                out += e.to_bytes(256, "big")
        self._pending = bytes(out)

    def verify_block_absorb(self, state):
        from mpvss_rs_amd import capi
        st = capi.transcript_absorb(state, self._pending)
        self._pending = b""
        return st

    def wellformed(self, pubkeys, shares, responses):
        """include/mpvss_hip.h, mpvss_modp_verify_block_compute_flags: canonical encodings of y_i, Y_i, r_i"""
        q = self.G.q
        sp = lambda b: [int.from_bytes(b[i:i + 256], "big") for i in range(0, len(b), 256)]
        return bytes(int(0 < y < q and 0 < Y < q and r < q - 1) for y, Y, r in zip(sp(pubkeys), sp(shares), sp(responses)))

    def ec_verify_block_compute(self, group, commitments, positions, pubkeys, shares, responses, challenge):
        O = self.O
        E = O.GROUPS["secp256k1" if group == 1 else "ristretto255"]()
        L = E.elem_len
        el = lambda b: [E.element_from_fixed(b[i:i + L]) for i in range(0, len(b), L)]
        sc = lambda b: [E.scalar_from_fixed(b[i:i + 32]) for i in range(0, len(b), 32)]
        cm, c = el(commitments), E.scalar_from_fixed(challenge)
        out = bytearray()
        for i, y, Y, r in zip(positions, el(pubkeys), el(shares), sc(responses)):
            X = O.commitment_eval(E, cm, i)
            a1, a2 = O.dleq_verifier_commitments(E, E.subgroup_generator(), X, y, Y, r, c)
            for e in (X, Y, a1, a2):
                out += E.element_to_bytes(e)
        self._pending_ec = (group, bytes(out))

    def ec_verify_block_absorb(self, state):
        from mpvss_rs_amd import capi
        group, data = self._pending_ec
        return capi.ec_transcript_absorb(group, state, data)

    def verify_shares(self, pk, s, y, c, r):
        O, G = self.O, self.G
        sp = lambda b: [int.from_bytes(b[i:i + 256], "big") for i in range(0, len(b), 256)]
        return bytes(int(O.dleq_verify(G, G.generator(), *a)) for a in zip(sp(pk), sp(s), sp(y), sp(c), sp(r)))

    def ec_verify_shares(self, group, pk, s, y, c, r):
        O = self.O
        E = O.GROUPS["secp256k1" if group == 1 else "ristretto255"]()
        L = E.elem_len
        el = lambda b: [E.element_from_fixed(b[i:i + L]) for i in range(0, len(b), L)]
        sc = lambda b: [E.scalar_from_fixed(b[i:i + 32]) for i in range(0, len(b), 32)]
        return bytes(int(O.dleq_verify(E, E.generator(), *a)) for a in zip(el(pk), el(s), el(y), sc(c), sc(r)))


def _worker(rank, world, port, tamper, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random

    import torch.distributed as dist

    import mpvss_oracle as O
    from helpers import cat, make_modp_instance, modp_keygen
    from mpvss_rs_amd.sharding import ShardedVerifier, block_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, t = 7, 3     # ragged split: 3 + 4
        g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, seed=77)
        flat = O.box_to_flat(g, box)
        if tamper:
            buf = bytearray(flat["responses"]); buf[5 * 256 + 17] ^= 2; flat["responses"] = bytes(buf)
        lo, hi = block_range(n, world, rank)
        sl = slice(lo * 256, hi * 256)
        sv = ShardedVerifier(OracleBlockEngine())
        verdict, digest, counts = sv.verify_distribution(flat["commitments"], flat["positions"][lo:hi],
                                                         flat["publickeys"][sl], flat["shares"][sl],
                                                         flat["responses"][sl], flat["challenge"])
        # the same box once more with the per-share well-formedness bytes all-gathered: share 2's public key is replaced
        # by q (not a canonical encoding) and share 4's response by q - 1; the box verdict is then False for a reason of
        # its own (the transcript changes), the flags say where
        block7 = max(block_range(n, world, k)[1] - block_range(n, world, k)[0] for k in range(world))
        pkb, rsb = bytearray(flat["publickeys"]), bytearray(flat["responses"])
        pkb[2 * 256:3 * 256] = g.q.to_bytes(256, "big")
        rsb[4 * 256:5 * 256] = (g.q - 1).to_bytes(256, "big")
        _, _, _, wf = sv.verify_distribution(flat["commitments"], flat["positions"][lo:hi], bytes(pkb)[sl], flat["shares"][sl],
                                             bytes(rsb)[sl], flat["challenge"], block=block7)
        assert list(wf) == [1, 1, 0, 1, 0, 1, 1], list(wf)
        v4 = sv.verify_distribution(flat["commitments"], flat["positions"][lo:hi], flat["publickeys"][sl], flat["shares"][sl],
                                    flat["responses"][sl], flat["challenge"], block=block7)
        assert (v4[0], v4[1], list(v4[3])) == (verdict, digest, [1] * n)
        # share-box verdicts, all-gathered
        rng = random.Random(5)
        sbs = [O.extract_secret_share(g, box, k, modp_keygen(g, rng)) for k in privs]
        keys = [g.element_to_bytes(p) for p in pks]
        S = [s["share"] for s in sbs]; c = [s["challenge"] for s in sbs]; r = [s["response"] for s in sbs]
        r[1] ^= 1; r[6] ^= 4
        Y = [box["shares"][k] for k in keys]
        block = max(block_range(n, world, k)[1] - block_range(n, world, k)[0] for k in range(world))
        allv = sv.verify_shares(cat(g, pks[lo:hi]), cat(g, S[lo:hi]), cat(g, Y[lo:hi]), cat(g, c[lo:hi]),
                                cat(g, r[lo:hi]), block)
        # the same collective for a curve group (ec_group): 5 share boxes of secp256k1 split 2 + 3, one proof spoiled
        E = O.GROUPS["secp256k1"]()
        erng = random.Random(9)
        eorder = E.group_order_int()
        eprivs = [erng.randrange(1, eorder) for _ in range(5)]
        epks = [E.generate_public_key(k) for k in eprivs]
        ebox = O.distribute_secret(E, 0x77, epks, 2, [erng.randrange(eorder) for _ in range(2)],
                                   [erng.randrange(1, eorder) for _ in range(5)])
        esb = [O.extract_secret_share(E, ebox, k, erng.randrange(1, eorder)) for k in eprivs]
        er = [x["response"] for x in esb]
        er[3] = (er[3] + 1) % eorder
        enc = lambda pts: b"".join(E.element_to_bytes(p) for p in pts)
        scb = lambda ks: b"".join(E.scalar_to_fixed(k) for k in ks)
        elo, ehi = block_range(5, world, rank)
        eY = [ebox["shares"][E.element_to_bytes(p)] for p in epks]
        eblock = max(block_range(5, world, k)[1] - block_range(5, world, k)[0] for k in range(world))
        ev = sv.verify_shares(enc(epks[elo:ehi]), enc([x["share"] for x in esb[elo:ehi]]), enc(eY[elo:ehi]),
                              scb([x["challenge"] for x in esb[elo:ehi]]), scb(er[elo:ehi]), eblock, ec_group=1)
        # verify_distribution_shares of a curve group, sharded the same way (participant.rs:1384-1442)
        eflat = O.box_to_flat(E, ebox)
        esl = slice(elo * E.elem_len, ehi * E.elem_len)
        ever = sv.verify_distribution(eflat["commitments"], eflat["positions"][elo:ehi], eflat["publickeys"][esl],
                                      eflat["shares"][esl], eflat["responses"][elo * 32:ehi * 32], eflat["challenge"], ec_group=1)
        etr = {}
        assert O.verify_distribution_shares(E, ebox, etr) is True
        assert ever[0] is True and ever[1] == etr["digest"] and ever[2] == [ehi - elo if k == rank else 5 - (ehi - elo) for k in range(world)]
        q.put((rank, verdict, digest, counts, list(allv), list(ev)))
    finally:
        dist.destroy_process_group()


def _worker_bad_block(rank, world, port, bad_rank, q):
    """one rank's block holds a negative position: its engine fails, nobody may hang, every rank raises"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import mpvss_oracle as O
    from helpers import make_modp_instance
    from mpvss_rs_amd.sharding import ShardedVerifier, ShardError, block_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, t = 7, 3
        g, privs, pks, coeffs, ws, box = make_modp_instance(n, t, seed=77)
        flat = O.box_to_flat(g, box)
        lo, hi = block_range(n, world, rank)
        sl = slice(lo * 256, hi * 256)
        pos = list(flat["positions"][lo:hi])
        if rank == bad_rank:
            pos[1] = -pos[1]
        sv = ShardedVerifier(OracleBlockEngine())
        try:
            sv.verify_distribution(flat["commitments"], pos, flat["publickeys"][sl], flat["shares"][sl],
                                   flat["responses"][sl], flat["challenge"])
            q.put((rank, "no error"))
        except ShardError as err:
            q.put((rank, str(err)))
        # the group is still usable: the honest box verifies afterwards
        verdict, digest, counts = sv.verify_distribution(flat["commitments"], flat["positions"][lo:hi], flat["publickeys"][sl],
                                                         flat["shares"][sl], flat["responses"][sl], flat["challenge"])
        q.put((rank, verdict))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_a_failing_rank_does_not_hang_the_group(bad_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad_block, args=(r, 2, port, bad_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errors = [m for _, m in results if isinstance(m, str)]
    assert len(errors) == 2 and all(f"rank(s) [{bad_rank}]" in m for m in errors), results
    assert sorted(m for _, m in results if isinstance(m, bool)) == [True, True]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("tamper", [False, True])
def test_two_rank_sharded_verification(tamper):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mpvss_oracle as O
    from helpers import make_modp_instance
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tamper, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process oracle answer for the same (possibly tampered) box
    g, privs, pks, coeffs, ws, box = make_modp_instance(7, 3, seed=77)
    if tamper:
        key = g.element_to_bytes(pks[5])
        rbytes = bytearray(box["responses"][key].to_bytes(256, "big")); rbytes[17] ^= 2
        box["responses"][key] = int.from_bytes(rbytes, "big")
    tr = {}
    expect = O.verify_distribution_shares(g, box, tr)
    assert expect is (not tamper)
    for rank, verdict, digest, counts, allv, ev in results:
        assert verdict is expect and digest == tr["digest"]
        assert counts == [3, 4]
        assert allv == [1, 0, 1, 1, 1, 1, 0]
        assert ev == [1, 1, 1, 0, 1]


def test_block_range_partitions_everything():
    from mpvss_rs_amd.sharding import block_range
    for n in (0, 1, 7, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            blocks = [block_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[k][1] == blocks[k + 1][0] for k in range(world - 1))
