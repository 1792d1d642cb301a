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


class ShardedVerifier:
    """Runs one box over all ranks.  `engine` is anything with the block interface of
    mpvss_rs_amd.capi.Engine (verify_block_compute / verify_block_absorb / verify_shares)."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.dev = _comm_device(group)

    # -- verify_distribution_shares ---------------------------------------------------------
    def verify_distribution(self, commitments: bytes, positions: Sequence[int], pubkeys: bytes, shares: bytes,
                            responses: bytes, challenge: bytes):
        """Arguments are THIS rank's block (commitments and challenge are replicated).
        Returns (verdict, digest, statuses) on every rank."""
        eng, rank, world = self.engine, self.rank, self.world
        eng.verify_block_compute(commitments, positions, pubkeys, shares, responses, challenge)
        if rank == 0:
            state = capi.transcript_init()
        else:
            buf = torch.empty(capi.TRANSCRIPT_STATE_BYTES, dtype=torch.uint8, device=self.dev)
            dist.recv(buf, src=rank - 1, group=self.group)
            state = bytes(buf.cpu().numpy().tobytes())
        state = eng.verify_block_absorb(state)
        out = torch.zeros(33, dtype=torch.uint8, device=self.dev)
        if rank + 1 < world:
            msg = torch.frombuffer(bytearray(state), dtype=torch.uint8).to(self.dev)
            dist.send(msg, dst=rank + 1, group=self.group)
        else:
            verdict, digest = capi.transcript_verdict(state, challenge)
            out = torch.frombuffer(bytearray(bytes([int(verdict)]) + digest), dtype=torch.uint8).to(self.dev)
        dist.broadcast(out, src=world - 1, group=self.group)
        raw = bytes(out.cpu().numpy().tobytes())
        # per-rank status record: (shares in block); one small all-gather per box
        mine = torch.tensor([len(positions)], dtype=torch.int64, device=self.dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=self.group)
        return bool(raw[0]), raw[1:33], [int(g.item()) for g in gathered]

    # -- verify_share, batched ------------------------------------------------------------------
    def verify_shares(self, pk: bytes, s: bytes, y: bytes, c: bytes, r: bytes, block: int) -> bytes:
        """Arguments are this rank's block of share boxes; `block` is the (equal) padded block
        length.  Returns the verdict bytes of ALL ranks, in rank order, padding removed by the
        caller via the returned per-rank counts."""
        n = len(pk) // EB
        mine = self.engine.verify_shares(pk, s, y, c, r) if n else b""
        t = torch.zeros(block, dtype=torch.uint8, device=self.dev)
        if n:
            t[:n] = torch.frombuffer(bytearray(mine), dtype=torch.uint8).to(self.dev)
        allv = torch.zeros(block * self.world, dtype=torch.uint8, device=self.dev)
        dist.all_gather_into_tensor(allv, t, group=self.group)   # RCCL all-gather of per-share verdicts
        counts = torch.tensor([n], dtype=torch.int64, device=self.dev)
        gathered = [torch.zeros_like(counts) for _ in range(self.world)]
        dist.all_gather(gathered, counts, group=self.group)
        raw = bytes(allv.cpu().numpy().tobytes())
        out = b"".join(raw[k * block: k * block + int(g.item())] for k, g in enumerate(gathered))
        return out
