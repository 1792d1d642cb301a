"""Multi-GPU form of the hot path: participants shard over ranks in contiguous position blocks
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).

What is exchanged, and why:
  * `verify_distribution_shares` (src/participant.rs:399-455) has ONE verdict per box, produced by
    a single SHA-256 that runs over every share in position order (src/dleq.rs:87-99).  The group
    arithmetic of a share touches only that share, so it shards with no collective; the ordered
    hash is carried from the rank holding block k to the rank holding block k+1 as a 128-byte
    running state (point-to-point send/recv).  Each rank enqueues its GPU work first and only
    then waits for the state, so the wait overlaps compute.  The verdict of the last rank is
    broadcast, and a per-rank status record is all-gathered (one small collective per box).
    On request the ranks also all-gather one well-formedness byte per share (canonical encodings
    of y_i, Y_i, r_i) -- the per-share verdict bytes of W_A that SURVEY 8(e) describes; they never
    enter the box verdict.
  * `verify_share` (src/participant.rs:361-386) has one verdict per share box: each rank verifies
    its block and the per-share verdict bytes are all-gathered -- the collective the north star
    names.
No data-path collective moves group elements between GPUs.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import capi

EB = capi.EB


def block_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank` when n shares are split over `world` ranks; the
    concatenation of the blocks in rank order is the original share order."""
    return (rank * n) // world, ((rank + 1) * n) // world


def _comm_device(group=None) -> torch.device:
    backend = dist.get_backend(group)
    if backend == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class ShardError(capi.EngineError):
    """Raised on EVERY rank, after the collectives of the call have completed, when some rank's engine failed."""


STATE = capi.TRANSCRIPT_STATE_BYTES


class ShardedVerifier:
    """Runs one box over all ranks.  `engine` is anything with the block interface of
    mpvss_rs_amd.capi.Engine (verify_block_compute / verify_block_absorb / verify_shares).

    A rank whose engine fails (a negative position in its block, an allocation failure, ...) still takes part in
    the chain and in every collective: it forwards the hash state marked as poisoned, so no other rank is left
    waiting; afterwards all ranks raise ShardError together (the reference panics on such a box)."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.dev = _comm_device(group)

    def _gather_status(self, count: int, failed: bool):
        mine = torch.tensor([count, int(failed)], dtype=torch.int64, device=self.dev)
        gathered = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(gathered, mine, group=self.group)
        rows = [[int(v) for v in g.cpu().tolist()] for g in gathered]
        return [r[0] for r in rows], [k for k, r in enumerate(rows) if r[1]]

    # -- verify_distribution_shares ---------------------------------------------------------
    def verify_distribution(self, commitments: bytes, positions: Sequence[int], pubkeys: bytes, shares: bytes,
                            responses: bytes, challenge: bytes, block: Optional[int] = None,
                            ec_group: Optional[int] = None):
        """Arguments are THIS rank's block (commitments and challenge are replicated).
        Returns (verdict, digest, counts) on every rank; raises ShardError on every rank if any rank failed.
        ec_group: capi.GROUP_SECP256K1 / GROUP_RISTRETTO255 for the curve groups (src/participant.rs:1384-1442,
        1827-1885; None: MODP-2048, :399-455).
        block (MODP): the padded block length (equal on all ranks).  Then every rank also produces one well-formedness byte
        per share of its block -- 1 iff y_i, Y_i, r_i are canonical encodings (0 < y, Y < q, r < q - 1) -- and the bytes of
        ALL ranks are all-gathered (RCCL when the group is "nccl": the engine writes them into the device tensor that is
        gathered, mpvss_modp_verify_block_compute_flags); the call returns (verdict, digest, counts, wellformed) with the
        padding removed.  The reference validates nothing at this point (src/groups/modp.rs:154-156), so these bytes are
        a per-share record for the operator and never part of the box verdict, which is the transcript comparison."""
        eng, rank, world = self.engine, self.rank, self.world
        error: Optional[Exception] = None
        enqueued = False
        n = len(positions)
        flags = torch.zeros(block, dtype=torch.uint8, device=self.dev) if block is not None else None
        if flags is not None and self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)      # torch's zero fill runs on torch's stream, the engine writes on its own
        try:
            if ec_group:
                eng.ec_verify_block_compute(ec_group, commitments, positions, pubkeys, shares, responses, challenge)
            elif flags is not None and self.dev.type == "cuda" and hasattr(eng, "verify_block_compute_flags"):
                eng.verify_block_compute_flags(commitments, positions, pubkeys, shares, responses, challenge, flags.data_ptr())
            else:
                eng.verify_block_compute(commitments, positions, pubkeys, shares, responses, challenge)
                if flags is not None and n:
                    flags[:n] = torch.frombuffer(bytearray(eng.wellformed(pubkeys, shares, responses)), dtype=torch.uint8).to(self.dev)
            enqueued = True
        except Exception as exc:      # still join the chain below
            error = exc
        poisoned = False
        if rank == 0:
            state = capi.transcript_init()
        else:
            buf = torch.empty(STATE + 1, dtype=torch.uint8, device=self.dev)
            dist.recv(buf, src=rank - 1, group=self.group)
            raw = bytes(buf.cpu().numpy().tobytes())
            state, poisoned = raw[:STATE], bool(raw[STATE])
        if enqueued:
            try:      # also frees the block slot when the chain is poisoned
                state = eng.ec_verify_block_absorb(state) if ec_group else eng.verify_block_absorb(state)
            except Exception as exc:
                error = error or exc
        poisoned = poisoned or error is not None
        out = torch.zeros(34, dtype=torch.uint8, device=self.dev)
        if rank + 1 < world:
            msg = torch.frombuffer(bytearray(state + bytes([int(poisoned)])), dtype=torch.uint8).to(self.dev)
            dist.send(msg, dst=rank + 1, group=self.group)
        else:
            if poisoned:
                verdict, digest = False, bytes(32)
            elif ec_group:
                verdict, digest = capi.ec_transcript_verdict(ec_group, state, challenge)
            else:
                verdict, digest = capi.transcript_verdict(state, challenge)
            out = torch.frombuffer(bytearray(bytes([int(verdict), int(poisoned)]) + digest), dtype=torch.uint8).to(self.dev)
        dist.broadcast(out, src=world - 1, group=self.group)
        raw = bytes(out.cpu().numpy().tobytes())
        allf = None
        if flags is not None:         # the block has been absorbed: the engine's flag bytes are final
            allf = torch.zeros(block * world, dtype=torch.uint8, device=self.dev)
            dist.all_gather_into_tensor(allf, flags, group=self.group)        # RCCL all-gather of per-share verdict bytes (W_A)
        counts, failed = self._gather_status(n, error is not None)   # one small all-gather per box
        if failed:
            raise ShardError(f"verify_distribution failed on rank(s) {failed}" + (f": {error}" if error else "")) from error
        if allf is None:
            return bool(raw[0]), raw[2:34], counts
        rawf = bytes(allf.cpu().numpy().tobytes())
        return bool(raw[0]), raw[2:34], counts, b"".join(rawf[k * block: k * block + cnt] for k, cnt in enumerate(counts))

    # -- verify_share, batched ------------------------------------------------------------------
    def verify_shares(self, pk: bytes, s: bytes, y: bytes, c: bytes, r: bytes, block: int,
                      ec_group: Optional[int] = None) -> bytes:
        """Arguments are this rank's block of share boxes; `block` is the (equal) padded block
        length.  Returns the verdict bytes of ALL ranks, in rank order, padding removed.
        ec_group: capi.GROUP_SECP256K1 / GROUP_RISTRETTO255 for the curve groups (None: MODP-2048).

        With RCCL ("nccl" group) the engine writes its verdict bytes straight into the device tensor that is
        all-gathered: the per-share verdicts never visit the host before the collective."""
        n = len(r) // 32 if ec_group else len(pk) // EB
        t = torch.zeros(block, dtype=torch.uint8, device=self.dev)
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)      # (as above: the fill must be done before the engine's stream writes verdicts)
        error: Optional[Exception] = None
        pre = (ec_group,) if ec_group else ()
        names = ("ec_verify_shares_compute", "ec_verify_shares_absorb", "ec_verify_shares") if ec_group else \
                ("verify_shares_compute", "verify_shares_absorb", "verify_shares")
        try:
            if n and self.dev.type == "cuda" and hasattr(self.engine, names[0]):
                getattr(self.engine, names[0])(*pre, pk, s, y, c, r, verdicts_dev_ptr=t.data_ptr())
                getattr(self.engine, names[1])(n)               # waits for the batch: t[:n] is final
            elif n:
                mine = getattr(self.engine, names[2])(*pre, pk, s, y, c, r)
                t[:n] = torch.frombuffer(bytearray(mine), dtype=torch.uint8).to(self.dev)
        except Exception as exc:
            error = exc
        allv = torch.zeros(block * self.world, dtype=torch.uint8, device=self.dev)
        dist.all_gather_into_tensor(allv, t, group=self.group)   # RCCL all-gather of per-share verdicts
        counts, failed = self._gather_status(n, error is not None)
        if failed:
            raise ShardError(f"verify_shares failed on rank(s) {failed}" + (f": {error}" if error else "")) from error
        raw = bytes(allv.cpu().numpy().tobytes())
        return b"".join(raw[k * block: k * block + cnt] for k, cnt in enumerate(counts))
