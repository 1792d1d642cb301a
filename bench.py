#!/usr/bin/env python3
"""Headline benchmark: DLEQ share verifications/sec, 2048-bit MODP, n=65536 t=256 per GPU.

A "step" is one complete `verify_distribution_shares` (src/participant.rs:399-455 of the
reference) over a synthetic honest-dealer box whose arrays are already resident in HBM:
commitment multi-exp X_i, DLEQ commitments a1_i/a2_i (HIP kernels), device->host copy of
X/Y/a1/a2, the ordered SHA-256 transcript on the host and the challenge comparison.  The box
must verify (verdict True) and reproduce the dealer's transcript digest or the run aborts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (weak scaling:
      the box has N x 65536 participants, rank g holds the contiguous block g)

Prints ONE JSON line on rank 0.
"""
import os as _os
import sys as _sys


def _launch_own_ranks():
    """`python3 bench.py --gpus N` (N > 1) without a launcher around it: start the N ranks ourselves, as a CHILD
    process (`python -m torch.distributed.run --nproc-per-node N bench.py ...`), relay rank 0's JSON line and exit with
    the child's code.  This runs before torch or the engine library is imported, i.e. before anything in this process
    can have touched the GPU (a process that has initialised HIP must not exec or fork GPU work on this pool)."""
    if "WORLD_SIZE" in _os.environ or "RANK" in _os.environ:
        return
    n = 1
    argv = _sys.argv[1:]
    for k, a in enumerate(argv):
        if a == "--gpus" and k + 1 < len(argv):
            n = int(argv[k + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(_os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), _os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:          # rank 0 prints ONE JSON line; anything else on stdout goes to stderr
        if line.startswith("{\"metric\""):
            _sys.stdout.write(line)
            _sys.stdout.flush()
        else:
            _sys.stderr.write(line)
    raise SystemExit(proc.wait())


if __name__ == "__main__":
    _launch_own_ranks()

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue
# run one after the other.  Every box in flight has its own stream pair, so 8 queues let the latency-bound launches
# of more boxes run side by side (no effect on the headline shape, +60 % for 4096-share boxes; more than 16 queues
# oversubscribe the hardware and hurt).  Must be set before the HIP runtime initialises; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# RCCL between processes needs dmabuf IPC on this pool's host driver (already exported on the GPU boxes; kept here so that a rank
# started by a launcher with a scrubbed environment still has it)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse
import collections
import concurrent.futures
import ctypes as C
import json
import math
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench_line  # noqa: E402
from mpvss_rs_amd import capi  # noqa: E402

EB = 256
Q = int(
    "ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404ddef9519b3cd3a43"
    "1b302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7edee386bfb5a899fa5ae9f24117c4b"
    "1fe649286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb9ed5290770"
    "96966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3be39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf6"
    "955817183995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff", 16)
ORDER = Q - 1
SEED = 0x6D70767373          # "mpvss"; Python random.Random (MT19937) streams, documented in DESIGN.md
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SHARE = 1536  # read y,Y,r + write X,a1,a2 (6 x 256 B)
# measured sustained issue rate (tools/ubench_clock.hip, profiles/r01_ubench_sustained_mad_clock.txt): one
# v_mad_u64_u32 wave-instruction per 2.07 ns per SIMD from 2 waves/SIMD up (~4.35 cycles at the ~2.1 GHz the
# chip holds under this load); a 2048-bit Montgomery product = 2*72*72 lane-mads = 2592 wave-mads per 16 numbers
MAD_NS_PER_SIMD = 2.07
PEAK_MODMUL_PER_S = 1024 / (2 * 72 * 18 / 16 * MAD_NS_PER_SIMD * 1e-9)   # 256 CUs x 4 SIMDs -> 3.05e9
# the same 4.35 cycles per wave-mad at the nominal 2.4 GHz instead of the ~2.1 GHz the chip holds at its power cap
PEAK_MODMUL_NOMINAL = 1024 / (2 * 72 * 18 / 16 * (4.35 / 2.4) * 1e-9)   # 3.49e9


def fx(v: int) -> bytes:
    return v.to_bytes(EB, "big")


def keygen(rng: random.Random) -> int:
    while True:                                   # modp.rs:162-174
        k = rng.randrange(Q)
        if math.gcd(k, ORDER) == 1:
            return k


PIPE_DEPTH = int(os.environ.get("MPVSS_BENCH_DEPTH", "10"))  # boxes with GPU work pending (the engine has capi.BLOCK_SLOTS block slots)
HASH_THREADS = int(os.environ.get("MPVSS_BENCH_HASH_THREADS", "8"))   # host threads absorbing (hashing) boxes at N=1
USE_VERIFY_MANY = os.environ.get("MPVSS_BENCH_VERIFY_MANY", "1") != "0"   # N=1: the library's own pipeline (0: Python threads)


SQ_COST = (72 * (9.5 + 18)) / (72 * 36)      # mads of a dedicated squaring relative to a general product (0.764)


def horner_modmuls(positions, t):
    """Work of the commit_eval kernel in full-product equivalents (a squaring counts SQ_COST), per 16-share wave
    (see k_modp_commit_eval): per Horner step (nb-1) squarings, one product per lower bit position at which any
    share of the wave has the bit set, plus the product with C_j."""
    total = 0.0
    for w in range(0, len(positions), 16):
        grp = positions[w:w + 16]
        nb = max(grp).bit_length()
        anyset = 0
        for p in grp:
            anyset |= p
        mults = bin(anyset & ((1 << max(nb - 1, 0)) - 1)).count("1")
        per_step = max(nb - 1, 0) * SQ_COST + mults + 1
        total += len(grp) * per_step * (t - 1)
    return total


EC_COUNTERS_FILE = os.path.join("profiles", "r04_ec_counters.json")      # PMC evidence taken on the headline shapes (committed files;
TRAFFIC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")          # the curve kernels have not changed since round 4's passes)
for _name, _old in (("EC_COUNTERS_FILE", ("r04_", "r03_")), ("TRAFFIC_FILE", ("r06_", "r05_"))):      # (the round before's file until this round's exists)
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), globals()[_name])):
        globals()[_name] = globals()[_name].replace(*_old)

EC = {
    "secp256k1": {"gid": 1, "enc": 33, "be": True, "algo_bytes": 197,     # 33+33+32 in, 3 x 33 out
                  "order": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
                  "gen": bytes.fromhex("0279BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798"),
                  "ref": "src/participant.rs:1384-1442"},
    "ristretto255": {"gid": 2, "enc": 32, "be": False, "algo_bytes": 192,  # 3 x 32 in, 3 x 32 out
                     "order": 2**252 + 27742317777372353535851937790883648493,
                     "gen": bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"),
                     "ref": "src/participant.rs:1827-1885"},
}


def bench_ec(eng, name, args):
    """BASELINE configs C3 (secp256k1) / C4 (ristretto255): verify_distribution_shares at n=65536, t=256 on one GPU,
    inputs resident in HBM; the box is dealt by the engine, must verify and reproduce the dealer's digest.  Returns the
    `ec` object of the JSON line: value, ms_per_box, roofline (SURVEY 8(d) bytes per share over the isolated duration of
    the dominant kernel) and a CPU baseline (oracle/ec_ref.c, reference operation sequence, sampled)."""
    cfg = EC[name]
    gid, L, order = cfg["gid"], cfg["enc"], cfg["order"]
    n, t = args.ec_n, args.ec_t
    sb = (lambda k: k.to_bytes(32, "big")) if cfg["be"] else (lambda k: k.to_bytes(32, "little"))
    rng = random.Random(SEED + gid)
    coeffs = [rng.randrange(order) for _ in range(t)]
    privs = [rng.randrange(1, order) for _ in range(n)]
    wits = [rng.randrange(1, order) for _ in range(n)]
    positions = list(range(1, n + 1))
    rc = list(reversed(coeffs))
    pvals = []
    for i in positions:
        acc = 0
        for a in rc:
            acc = (acc * i + a) % order
        pvals.append(acc)
    cm = eng.ec_batch_exp_generator(gid, b"".join(map(sb, coeffs)))       # C_j = a_j G (fixed-base comb)
    pks = eng.ec_batch_exp_generator(gid, b"".join(map(sb, privs)))      # y_i = x_i G
    t_deal = time.perf_counter()
    d = eng.ec_distribute(gid, cm, positions, pks, b"".join(map(sb, pvals)), b"".join(map(sb, wits)))
    deal_s = time.perf_counter() - t_deal
    cbytes = capi.ec_hash_to_scalar(gid, d["digest"])
    c = int.from_bytes(cbytes, "big" if cfg["be"] else "little")
    responses = b"".join(sb((w - p * c) % order) for w, p in zip(wits, pvals))       # dleq.rs:42-50
    dev = torch.device("cuda", torch.cuda.current_device())
    dbuf = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_cm, d_pk, d_Y, d_r = dbuf(cm), dbuf(pks), dbuf(d["Y"]), dbuf(responses)
    d_pos = torch.tensor(positions, dtype=torch.int64, device=dev)
    chal = (C.c_uint8 * 32).from_buffer_copy(cbytes)
    vp = lambda x: C.c_void_p(x.data_ptr())
    torch.cuda.synchronize()
    X = (C.c_uint8 * (n * L))(); A1 = (C.c_uint8 * (n * L))(); A2 = (C.c_uint8 * (n * L))()

    def verify(dump=False):
        verdict = C.c_int(0)
        dg = (C.c_uint8 * 32)()
        eng._check(eng.lib.mpvss_ec_verify_distribution(eng.ctx, gid, capi.MPVSS_DEVICE, vp(d_cm), t, vp(d_pos), vp(d_pk), vp(d_Y),
                                                        vp(d_r), n, C.cast(chal, C.c_void_p), C.byref(verdict), C.cast(dg, C.c_void_p),
                                                        X if dump else None, A1 if dump else None, A2 if dump else None),
                   "ec_verify_distribution")
        assert bool(verdict.value) and bytes(dg) == d["digest"], f"parity gate failed ({name})"

    verify(dump=True)                                   # warm-up + the outputs for the CPU comparison
    # isolated launches: one synchronous box, nothing else on the GPU
    eng.pipeline_stats(reset=True)
    torch.cuda.synchronize()
    t_l0 = time.perf_counter()
    verify()
    lone_wall = (time.perf_counter() - t_l0) * 1e3
    lst = eng.pipeline_stats(reset=True)
    lone = {"x_path": lst["kernel_ms"][0], "dual_win": lst["kernel_ms"][1], "tables": lst["kernel_ms"][2],
            "encode": lst["kernel_ms"][3], "dual_win_launches": lst["kernel_launches"][1],
            # the whole synchronous call (a2 runs beside the X path, so the GPU part is shorter than the sum of the launches above):
            # enqueue + wait = the box on the GPU, then its transcript on one host thread
            "call_wall_ms": lone_wall, "enqueue_ms": lst["enqueue_ms"], "wait_for_gpu_ms": lst["wait_ms"],
            "box_on_the_gpu_ms": lst["enqueue_ms"] + lst["wait_ms"],
            "sha256_transcript_ms": lst["hash_ms"],
            "x_path_is": "seed kernel with windowed x^lo, difference tables and stepping as pipelines of quad-lane stages (ec_quad.h): "
                         "what a box that has the chip to itself takes; batched boxes keep one workgroup per chain"}
    dual_ms = lone["dual_win"] / max(lone["dual_win_launches"], 1)
    # timed: K boxes through the library's pipeline (mpvss_ec_verify_many): EC_DEPTH boxes in flight in ONE context
    k = args.ec_boxes
    depth, threads = int(os.environ.get("MPVSS_BENCH_EC_DEPTH", "16")), int(os.environ.get("MPVSS_BENCH_EC_HASH_THREADS", "6"))
    box = capi.EcBox(d_cm.data_ptr(), t, d_pos.data_ptr(), d_pk.data_ptr(), d_Y.data_ptr(), d_r.data_ptr(), n,
                     C.cast(chal, C.c_void_p))

    def run_many(count, depth=depth):
        arr = (capi.EcBox * count)(*([box] * count))
        verdicts = (C.c_int * count)()
        digests = (C.c_uint8 * (32 * count))()
        eng._check(eng.lib.mpvss_ec_verify_many(eng.ctx, gid, capi.MPVSS_DEVICE, arr, count, depth, threads, verdicts,
                                                C.cast(digests, C.c_void_p)), "ec_verify_many")
        raw = bytes(digests)
        assert all(verdicts[i] == 1 and raw[32 * i:32 * i + 32] == d["digest"] for i in range(count)), f"parity gate failed ({name})"

    init = min(depth + threads + 4, capi.BLOCK_SLOTS - 1)
    run_many(init, init)                                # slot workspaces: as many boxes at once as can ever be in flight
    eng.pipeline_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_many(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    pst = eng.pipeline_stats(reset=True)
    nb = max(pst["blocks"], 1)
    out = {"value": n / dt, "unit": "share verifications/s", "ms_per_box": dt * 1e3, "boxes": k,
           "boxes_verified_in_this_process": 2 + init + k,      # (what a counter pass over this run divides by)
           "config": {"workload": f"{name} verify_distribution_shares n={n} t={t}, honest-dealer box, inputs resident in HBM, "
                                  f"{depth} boxes in flight in one context, {threads} hash threads ({cfg['ref']})"},
           "dtype": "u32 limbs (radix 2^26), u64 accumulators",
           "roofline": {"bound": "hbm", "kernel": ("k_secp_dual_win" if gid == 1 else "k_rist_dual_win"),
                        "achieved": cfg["algo_bytes"] * n / (dual_ms * 1e-3) / 1e9 if dual_ms > 0 else None, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": cfg["algo_bytes"] * n / (dual_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if dual_ms > 0 else None,
                        "traffic": None, "kernel_ms": dual_ms,
                        "kernel_ms_is": "average over the windowed double-scalar-multiplication launches of a box alone on the GPU "
                                        "(synchronous call after the timed region: a2 = r y + c Y, and a1 = r G + c X in its two "
                                        "halves -- the generator half runs beside the X path of a lone box)"},
           "kernel_ms_isolated": lone,
           "host_per_box_ms": {"enqueue": pst["enqueue_ms"] / nb, "wait_for_gpu": pst["wait_ms"] / nb,
                               "sha256_transcript": pst["hash_ms"] / nb},
           "distribute_shares_per_s": n / deal_s}
    # counter evidence of the same shape (profiles/r03_ec_counters.json: SQ_INSTS_VALU and FETCH/WRITE per verified box, PMC
    # passes of tools/run_profiles_ec.sh; profiles/r03_pmc_traffic.json: bytes per launch of the dominant kernel)
    try:
        if (n, t) == (65536, 256):
            cfile = EC_COUNTERS_FILE
            ecc = json.load(open(os.path.join(ROOT, cfile)))[name]
            slots = ecc["valu_wave_insts_per_box"] / dt
            out["compute"] = {"bound": "valu issue", "achieved": slots, "peak": PEAK_VALU_SLOTS_PER_S, "peak_is": PEAK_SOURCE,
                              "frac": slots / PEAK_VALU_SLOTS_PER_S,
                              "unit": "VALU wave-instruction issue slots/s (SQ_INSTS_VALU per verified box x boxes/s)",
                              "valu_wave_insts_per_box": ecc["valu_wave_insts_per_box"],
                              "hbm_bytes_per_box": ecc["hbm_bytes_per_box"], "hbm_gb_per_s": ecc["hbm_bytes_per_box"] / dt / 1e9,
                              "source": f"{cfile}: {ecc.get('how', 'rocprofv3 --pmc passes over the same call (mpvss_ec_verify_many, X paths batched), counters per verified box')}; "
                                        "the box rate is this run's"}
            tr = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
            out["roofline"]["traffic"] = tr.get(out["roofline"]["kernel"] + "_bytes_per_launch")
            out["roofline"]["traffic_source"] = f"{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed; not measured in this run)"
    except (OSError, KeyError, ValueError):
        pass
    if os.environ.get("MPVSS_BENCH_EC_VERIFY_ONLY") == "1":       # counter passes: nothing but verifications after the set-up
        return out
    # dealer side in block form (mpvss_ec_distribute_compute / _absorb): inputs resident in HBM, X_i = P(i) G through the
    # comb, 8 blocks in flight, absorbed (validated + hashed) by a few host threads; the digest must be the dealer's
    import concurrent.futures
    d_pv, d_wt = dbuf(b"".join(map(sb, pvals))), dbuf(b"".join(map(sb, wits)))
    zero_c = bytes(32)

    def deal_absorb():
        st = (C.c_uint8 * capi.TRANSCRIPT_STATE_BYTES).from_buffer_copy(capi.transcript_init())
        eng._check(eng.lib.mpvss_ec_distribute_absorb(eng.ctx, st, None, None, None, None), "ec_distribute_absorb")
        return capi.ec_transcript_verdict(gid, bytes(st), zero_c)[1]

    def deal_many(count, in_flight=8):
        with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
            issued, pend, got = 0, [], []
            while issued < count or pend:
                if issued < count and len(pend) < in_flight:
                    eng._check(eng.lib.mpvss_ec_distribute_compute(eng.ctx, gid, capi.MPVSS_DEVICE, None, 0, None, vp(d_pk), vp(d_pv),
                                                                   vp(d_wt), n, None, None, None, None), "ec_distribute_compute")
                    issued += 1
                    pend.append(pool.submit(deal_absorb))
                else:
                    got.append(pend.pop(0).result())
        return got

    deal_many(12)
    torch.cuda.synchronize()
    t_d = time.perf_counter()
    dgs = deal_many(16)
    torch.cuda.synchronize()
    deal_blk_s = (time.perf_counter() - t_d) / 16
    assert all(x == d["digest"] for x in dgs), f"dealer block API: transcript digest differs ({name})"
    out["distribute"] = {"value": n / deal_blk_s, "unit": "shares dealt/s", "ms_per_box": deal_blk_s * 1e3, "boxes_in_flight": 8,
                         "value_synchronous_host_buffers": n / deal_s}
    # ... and the dealer END TO END with the scalar side on the device as well: every box its own polynomial (t coefficients from
    # the host), P(i) mod order by mpvss_ec_deal_compute on the block's stream, the group work, the transcript, the challenge
    # c = hash_to_scalar(digest), the responses r_i = w_i - P(i) c by mpvss_ec_dleq_responses_device; only the coefficients, the
    # digest and the challenge cross the bus (participant.rs:1134-1168, 1200-1230 / 1607-1631, 1662-1690)
    e2e_boxes, e2e_depth = 16, 8
    coeff_sets = []
    for b in range(4):
        rb = random.Random(SEED + 31 * gid + b)
        coeff_sets.append(coeffs if b == 0 else [rb.randrange(order) for _ in range(t)])
    coeff_bytes = [b"".join(map(sb, cs_)) for cs_ in coeff_sets]
    ring_p = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(e2e_depth + 6)]     # P(i) of a box lives until its responses are out
    d_rr = [torch.empty(n * 32, dtype=torch.uint8, device=dev) for _ in range(e2e_depth + 6)]
    torch.cuda.synchronize()

    def e2e_resp(b, d_pb, digest):
        cc = capi.ec_hash_to_scalar(gid, digest)
        d_r = d_rr[b % len(d_rr)]
        eng.ec_dleq_responses_device(gid, d_wt.data_ptr(), d_pb.data_ptr(), cc, n, d_r.data_ptr())
        return bytes(d_r.cpu().numpy().tobytes()) if b == 0 else None

    def deal_e2e(count):
        # the block of every box is claimed as soon as it is enqueued (mpvss_block_claim: the ticket says which block a thread
        # holds), three threads wait for and hash blocks side by side, the challenge and the responses of a box are another pool's
        with concurrent.futures.ThreadPoolExecutor(max_workers=3) as absorb_pool, concurrent.futures.ThreadPoolExecutor(max_workers=3) as resp_pool:
            def e2e_post(b, d_pb, ticket):
                st = (C.c_uint8 * capi.TRANSCRIPT_STATE_BYTES).from_buffer_copy(capi.transcript_init())
                eng._check(eng.lib.mpvss_ec_block_absorb_claimed(eng.ctx, ticket, st, None, None, None, None), "ec_block_absorb_claimed")
                digest = capi.ec_transcript_verdict(gid, bytes(st), zero_c)[1]
                return digest, resp_pool.submit(e2e_resp, b, d_pb, digest)
            post = []
            for b in range(count):
                d_pb = ring_p[b % len(ring_p)]
                while len(post) - sum(f.done() for f in post) >= e2e_depth:
                    time.sleep(0.0002)
                eng.ec_deal_compute(gid, coeff_bytes[b % len(coeff_bytes)], d_pos.data_ptr(), d_pk.data_ptr(), d_wt.data_ptr(), n, d_pb.data_ptr())
                post.append(absorb_pool.submit(e2e_post, b, d_pb, eng.block_claim()))
            outs = [f.result() for f in post]
            return [(dg_, fut.result()) for dg_, fut in outs]

    deal_e2e(4)
    torch.cuda.synchronize()
    t_e = time.perf_counter()
    e2e = deal_e2e(e2e_boxes)
    torch.cuda.synchronize()
    e2e_s = (time.perf_counter() - t_e) / e2e_boxes
    assert e2e[0][0] == d["digest"] and e2e[0][1] == responses, f"end-to-end dealer ({name}): box 0 differs from the dealer's"
    ec_call, ec_outputs = eng.ec_deal_call(gid, coeff_bytes[0], positions, pks, b"".join(map(sb, wits)))
    ec_call()                      # (the library call alone over ctypes buffers made once; mean of 3 after a warm call)
    t_one = time.perf_counter()
    for _ in range(3):
        ec_call()
    one_s = (time.perf_counter() - t_one) / 3
    one = ec_outputs()
    assert one["digest"] == d["digest"] and one["responses"] == responses and one["Y"] == d["Y"], f"mpvss_ec_deal differs ({name})"
    out["distribute"].update({"value_end_to_end": n / e2e_s, "end_to_end_ms_per_box": e2e_s * 1e3, "end_to_end_boxes_in_flight": e2e_depth,
                              "value_one_call_host_buffers_end_to_end": n / one_s,
                              "note": "`value`: group work of the dealer's blocks with P(i) given (mpvss_ec_distribute_compute / _absorb); "
                                      "`value_end_to_end`: every box from its own t coefficients -- P(i) mod order and the responses on the device "
                                      "(mpvss_ec_deal_compute, mpvss_ec_dleq_responses_device), group work, transcript, challenge, boxes pipelined; "
                                      "`value_one_call_host_buffers_end_to_end`: one mpvss_ec_deal call from host buffers"})
    # W_B for the curve group (participant.rs:1346-1371 / 1789-1814): a batch of decrypted-share proofs -- a1 = r G + c pk, a2 = r S + c Y
    # and the per-share hash verdict (K7) on the device --, inputs resident in HBM, several batches in flight in one context
    m = min(n, 16384)
    xinv = b"".join(sb(pow(x_, -1, order)) for x_ in privs[:m])
    rng_w = random.Random(SEED + 97 * gid)
    wit_b = [rng_w.randrange(1, order) for _ in range(m)]
    S_, cb_ = eng.ec_extract_shares(gid, pks[:m * L], d["Y"][:m * L], xinv, b"".join(map(sb, wit_b)))
    from_b = (lambda b_: int.from_bytes(b_, "big")) if cfg["be"] else (lambda b_: int.from_bytes(b_, "little"))
    rb_ = b"".join(sb((w_ - x_ * from_b(cb_[i * 32:(i + 1) * 32])) % order) for i, (w_, x_) in enumerate(zip(wit_b, privs[:m])))   # dleq.rs:42-50
    d_S, d_cb, d_rb = dbuf(S_), dbuf(cb_), dbuf(rb_)
    verd = (C.c_uint8 * m)()
    inflight, batches = 4, 12
    torch.cuda.synchronize()

    def wb_pipelined(count):
        issued = done = 0
        while done < count:
            while issued < count and issued - done < inflight:
                eng._check(eng.lib.mpvss_ec_verify_shares_compute(eng.ctx, gid, capi.MPVSS_DEVICE, vp(d_pk), vp(d_S), vp(d_Y), vp(d_cb), vp(d_rb),
                                                                  m, None), "ec_verify_shares_compute")
                issued += 1
            eng._check(eng.lib.mpvss_ec_verify_shares_absorb(eng.ctx, verd), "ec_verify_shares_absorb")
            assert bytes(verd) == b"\x01" * m, f"verify_share verdicts ({name})"
            done += 1

    wb_pipelined(inflight)
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    wb_pipelined(batches)
    torch.cuda.synchronize()
    wb_s = (time.perf_counter() - t_w) / batches
    out["verify_share"] = {"value": m / wb_s, "unit": "share-box verifications/s", "batch": m, "batches_in_flight": inflight,
                           "note": "W_B for the curve group: a1 = r G + c pk, a2 = r S + c Y and the per-share hash verdict on the device "
                                   "(mpvss_ec_verify_shares_compute / _absorb), share boxes made by mpvss_ec_extract_shares, inputs resident in HBM"}
    if args.cpu_sample != 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from concurrent.futures import ThreadPoolExecutor

        from ec_ref import EcRef
        ref = EcRef()
        k = 256 if args.cpu_sample < 0 else min(args.cpu_sample, n)
        idx = sorted({int((j + 0.5) * n / k) for j in range(k)})
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
        Xb, A1b, A2b = bytes(X), bytes(A1), bytes(A2)

        def work(i):
            return ref.share_work(gid, cm, positions[i], pks[i * L:(i + 1) * L], d["Y"][i * L:(i + 1) * L],
                                  responses[i * 32:(i + 1) * 32], cbytes)
        tc = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            outs = list(ex.map(work, idx))
        cpu_s = time.perf_counter() - tc
        for i, (x, a1, a2) in zip(idx, outs):
            s_ = slice(i * L, (i + 1) * L)
            assert (x, a1, a2) == (Xb[s_], A1b[s_], A2b[s_]), f"GPU/CPU mismatch at {name} share {i}"
        out["cpu_baseline"] = {
            "value": len(idx) / cpu_s, "unit": "share verifications/s", "cores": cores, "kind": "port",
            "sample": f"{len(idx)} of {n} shares spread over [1,{n}], all t={t} commitments: reference operation sequence (t+4 "
                      f"scalar multiplications with full-width exponents, t+2 additions, an affine conversion each for secp256k1) in "
                      f"oracle/ec_ref.c (textbook double-and-add; the reference's k256 / curve25519-dalek are about 3-4x faster per "
                      f"multiplication) on {cores} threads, {cpu_s:.1f}s; GPU X/a1/a2 of those shares checked equal"}
        if gid == 1:      # the strong-CPU line: the same sequence with libcrypto's EC_POINT_mul (oracle/openssl_ref.py; no ristretto255 there)
            try:
                import openssl_ref
                if openssl_ref.ec_available():
                    tls = threading.local()

                    def work_ssl(i):
                        if not hasattr(tls, "ref"):
                            tls.ref = openssl_ref.OpenSslSecpRef()
                        return tls.ref.share_work(cm, positions[i], pks[i * L:(i + 1) * L], d["Y"][i * L:(i + 1) * L],
                                                  responses[i * 32:(i + 1) * 32], cbytes)
                    t1 = time.perf_counter(); work_ssl(idx[0]); one = time.perf_counter() - t1
                    ks_ = max(cores, min(16 * cores, int(4.0 * cores / max(one, 1e-3))))
                    idx_s = sorted({int((j + 0.5) * n / ks_) for j in range(ks_)})
                    tc = time.perf_counter()
                    with ThreadPoolExecutor(max_workers=cores) as ex:
                        outs = list(ex.map(work_ssl, idx_s))
                    ssl_s = time.perf_counter() - tc
                    for i, (x, a1, a2) in zip(idx_s, outs):
                        s_ = slice(i * L, (i + 1) * L)
                        assert (x, a1, a2) == (Xb[s_], A1b[s_], A2b[s_]), f"GPU/OpenSSL mismatch at {name} share {i}"
                    out["cpu_baseline"]["openssl"] = {
                        "value": len(idx_s) / ssl_s, "unit": "share verifications/s", "cores": cores, "kind": "port",
                        "library": openssl_ref.version(),
                        "sample": f"{len(idx_s)} shares, the same operation sequence with libcrypto's EC_POINT_mul / EC_POINT_add on {cores} "
                                  f"threads, {ssl_s:.1f}s; one share on one thread {one * 1e3:.0f} ms; GPU X/a1/a2 of those shares checked equal"}
            except Exception as exc:      # noqa: BLE001 - optional line
                out["cpu_baseline"]["openssl"] = {"value": None, "note": f"skipped: {exc}"}
    return out


def inv_tree_products(m):                                         # simultaneous inversion: 3 products per node
    total = 0
    while m > 1:
        total += 3 * m
        m = -(-m // 16)
    return total


def sliding_windows(c):
    """(windows, bit position of the lowest bit of the top window) of the width-4 sliding-window schedule the library
    makes from a challenge (sliding_schedule in mpvss_capi.cpp)"""
    c_win, c_top, i = 0, 0, 255
    while i >= 0:
        if not (c >> i) & 1:
            i -= 1
            continue
        low = max(i - 3, 0)
        while not (c >> low) & 1:
            low += 1
        c_top = low if c_win == 0 else c_top
        c_win += 1
        i = low - 1
    return c_win, c_top


# VALU wave-instruction issue slots per number and Montgomery operation (what bounds these kernels: every VALU instruction
# costs one issue slot of ~4 cycles per SIMD whatever it is; profiles/r01_ubench_*):
#   quad layout (bn_quad.h):  72 rows x 41 (product) / 32.5 (squaring) instructions + ~110 for the final passes, 16 numbers per wave
#   pair layout (bn_pair.h):  3 690 (product) / 2 501 (squaring) VALU instructions per wave of 32 numbers, counted in the ISA
#                             (tools/mfma_mont/count_isa.py), + 114 MFMAs that each hold the SIMD's issue for 2 slots
QUAD_MUL_SLOTS, QUAD_SQ_SLOTS = (72 * 41 + 110) / 16.0, (72 * 32.5 + 110) / 16.0
PAIR_MUL_SLOTS, PAIR_SQ_SLOTS = (3690 + 2 * 114) / 32.0, (2501 + 2 * 114) / 32.0
# The peak of the issue-slot accounting is a physical one: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at the
# nominal 2.4 GHz shader clock -- nothing can exceed it, so no `frac` of it can exceed 1.  What the chip SUSTAINS is lower (it
# lowers its clock under load: about 2.0-2.1 GHz in these kernels, and a v_mad_u64_u32 issues in ~4.35 cycles): that figure is
# measured live in every run by mpvss_issue_probe (`compute.sustained_probe`) and the achieved rate is also given relative to
# it (`vs_sustained_mad64`, a ratio of two measurements of different instruction mixes: it is not called a fraction).
# Instruction mix per Montgomery operation and wave, by issue class (tools/isa_mix.py over the ISA of the shipped a2 kernel,
# profiles/r05_a2_isa_mix.json; the quad layout from bn_quad.h's row: 36 / 27.5 v_mad_u64_u32, 2 DPP ands, v_mov_b64, v_lshrrev_b64,
# v_lshl_add_u64 per row, ~110 instructions of final passes).  `compute.peak_mix_weighted` prices these with what mpvss_issue_probe
# measures for every class IN THIS RUN (four waves per SIMD back to back): the time the chip needs for the run's instructions if
# every class issued at its own sustained rate -- a v_add_u32 is not a v_mad_u64_u32 (round 4's flat 4-cycle slot said it was).
MIX_CLASSES = ("mad64", "alu32", "shift64", "swap", "mov64", "vop2")
PAIR_MIX = {"numbers": 32,
            "sq": {"mad64": 1441, "alu32": 745, "shift64": 223, "swap": 84, "mov64": 72, "mfma": 114},
            "mul": {"mad64": 2701, "alu32": 675, "shift64": 223, "swap": 84, "mov64": 72, "mfma": 114}}
QUAD_MIX = {"numbers": 16,
            "sq": {"mad64": 72 * 27.5, "alu32": 72 * 2 + 110, "shift64": 144, "mov64": 72},
            "mul": {"mad64": 72 * 36, "alu32": 72 * 2 + 110, "shift64": 144, "mov64": 72}}


def mix_seconds_per_simd(by_layout, tau):
    """SIMD-seconds the operations in `by_layout` (numbers x operations, keys pair_sq / pair_mul / quad_sq / quad_mul) need when
    every instruction class issues at the rate measured for it: tau[c] = seconds per wave-instruction per SIMD; an MFMA holds the
    issue for two alu32 slots."""
    total = 0.0
    for key, cnt in by_layout.items():
        mix = PAIR_MIX if key.startswith("pair") else QUAD_MIX
        per_wave = sum(v * (2 * tau["alu32"] if c == "mfma" else tau[c]) for c, v in mix["sq" if key.endswith("sq") else "mul"].items())
        total += cnt / mix["numbers"] * per_wave
    return total


NOMINAL_CLOCK_GHZ = 2.4
PEAK_VALU_SLOTS_PER_S = 1024 * NOMINAL_CLOCK_GHZ * 1e9 / 4.0
PEAK_SOURCE = "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction (MI355X_MICROARCH.md: chip parameters, cycle constants)"
PAIR_MASK = int(os.environ.get("MPVSS_PAIR", "49")) & 63
FD_PAIR_MIN_T = int(os.environ.get("MPVSS_FD_PAIR_MIN_T", "512"))     # from this many commitments the X path steps in the pair layout


def modp_work(n, t, positions, cs):
    """Montgomery operations the kernels execute for ONE verification of a block of n shares at `positions` with t
    commitments, averaged over the challenges `cs` of the timed boxes; mirrors the choices of eval_x() /
    verify_block_compute_locked() in mpvss_capi.cpp for a block that is part of a pipelined run.  Two accountings:
    mm_total -- product equivalents (a squaring counts SQ_COST, its share of the mads of the quad layout), as in rounds 1-2;
    slots    -- VALU issue slots, per layout of the kernel that does the work (the a2 kernel runs in the pair layout)."""
    sq_n = {"x": 0.0, "a1": 0.0, "a2": 0.0, "tab": 0.0}
    mul_n = {"x": 0.0, "a1": 0.0, "a2": 0.0, "tab": 0.0}
    fd = os.environ.get("MPVSS_FD", "1") != "0" and 16 <= t <= 1024 and n >= 16 * t and n >= int(os.environ.get("MPVSS_FD_MIN_SHARES", "4096"))
    if fd:
        chains = max(1, min(max(min(2048 // t, n // 8192), n // 16384, 4),
                            n // (4 * t)))                                   # as eval_x() in mpvss_capi.cpp
        chain_len = -(-n // chains)
        w0 = (chain_len - t) // 2                                        # seeds sit in the middle of every chain
        m0 = chains * t                                                  # the seeds: m0 consecutive positions
        steps = (chain_len - 1 - w0) + (w0 + t - 1)                      # forward + backward pipeline of every chain
        two_level = os.environ.get("MPVSS_FD_L1", "1") != "0" and chains > 1
        if two_level:     # Horner for t seeds only; a stride-1 chain steps through the other (chains-1)*t seed positions
            seed_work = (horner_modmuls(positions[chains * w0:chains * w0 + t], t) + inv_tree_products(t) + t * (t - 1)
                         + t * (m0 - 1))
            seed_txt = f"{t} Horner seeds, a stride-1 chain through the other {m0 - t} seed positions"
        else:
            seed_work = horner_modmuls(positions[chains * w0:chains * w0 + m0], t)
            seed_txt = f"{m0} Horner seeds"
        mm_x = seed_work + inv_tree_products(m0) + chains * t * (t - 1) + chains * t * steps + n
        if (PAIR_MASK & 32) and t >= FD_PAIR_MIN_T:          # the stepping products (both levels) run in the pair layout
            mul_n["x_pair"] = chains * t * steps + (t * (m0 - 1) if two_level else 0)
        x_path = (f"forward differences: {chains} strided chains stepping both ways from {m0} seeds in their middle "
                  f"(also outputs; {seed_txt}), inverses by simultaneous inversion, {steps} lock-step products per chain and level")
    else:
        mm_x = horner_modmuls(positions, t) + n
        x_path = "Horner in the exponent"
    mul_n["x"] = mm_x - mul_n.get("x_pair", 0.0)               # (Horner's squarings are folded in at SQ_COST: a few % of the X path)
    comb_min = int(os.environ.get("MPVSS_COMB16_MIN", "8192"))
    gr = 127 if (comb_min > 0 and n >= comb_min) else 511     # g^r: wide comb (16-bit windows) or 4-bit comb
    w6 = n >= 1024   # 6-bit windows for y^r (64-entry table) or 4-bit
    # X^c and Y^c: 64 fixed 4-bit windows, or -- one c for the whole box, forward-difference path -- the sliding-window
    # schedule the library makes from it (width 4, odd digits; the tables of X and Y then hold the odd powers only)
    tot_a2 = 0.0
    for c in cs:
        sliding = fd and w6 and c > 0
        c_win, c_top = sliding_windows(c) if sliding else (0, 0)
        yc = c_win if sliding else 64                          # products with the table of Y (a2) / X (a1; its first is a load)
        xc_sq, xc = (c_top, c_win - 1) if sliding else (252, 63)
        a2_sq, a2_mul = (2046, 341 + yc + 1) if w6 else (2044, 511 + 64 + 1)
        tot_a2 += a2_sq * SQ_COST + a2_mul
        sq_n["a2"] += n * a2_sq
        mul_n["a2"] += n * a2_mul
        sq_n["a1"] += n * xc_sq
        mul_n["a1"] += n * (xc + 1 + gr + 2)                   # a1: comb for g^r + X^c windows + the two closing products
        sq_n["tab"] += n * (2 if sliding else 0)               # b^2 of the two odd-power tables
        mul_n["tab"] += n * ((2 * 8 if sliding else 2 * 15) + (63 if w6 else 15))   # tables of X, Y and y (+1 conversion each)
    k = max(len(cs), 1)
    for d in (sq_n, mul_n):
        for key in ("a1", "a2", "tab"):
            d[key] /= k
    mm_total = sum(mul_n.values()) + SQ_COST * sum(sq_n.values())
    pair_bit = {"a2": 1, "tab": 2, "a1": 8}                     # MPVSS_PAIR bits (g^r, bit 2, is counted with a1)
    slots = mul_n.get("x_pair", 0.0) * PAIR_MUL_SLOTS
    by_layout = {"pair_sq": 0.0, "pair_mul": mul_n.get("x_pair", 0.0), "quad_sq": 0.0, "quad_mul": 0.0}     # operations of the block
    for key in ("x", "a1", "a2", "tab"):
        pair = bool(PAIR_MASK & pair_bit.get(key, 0)) and n >= 16
        slots += sq_n[key] * (PAIR_SQ_SLOTS if pair else QUAD_SQ_SLOTS) + mul_n[key] * (PAIR_MUL_SLOTS if pair else QUAD_MUL_SLOTS)
        by_layout["pair_sq" if pair else "quad_sq"] += sq_n[key]
        by_layout["pair_mul" if pair else "quad_mul"] += mul_n[key]
    a2_pair = bool(PAIR_MASK & 1)
    return {"mm_total": mm_total, "mm_x": mm_x, "x_path": x_path, "a2_products": tot_a2 / k, "w6": w6, "fd": fd, "slots": slots,
            "by_layout": by_layout,
            "a2_by_layout": {("pair_sq" if a2_pair else "quad_sq"): sq_n["a2"], ("pair_mul" if a2_pair else "quad_mul"): mul_n["a2"]},
            "a2_slots": sq_n["a2"] * (PAIR_SQ_SLOTS if PAIR_MASK & 1 else QUAD_SQ_SLOTS) + mul_n["a2"] * (PAIR_MUL_SLOTS if PAIR_MASK & 1 else QUAD_MUL_SLOTS),
            "ops": {"squarings": sum(sq_n.values()) / n, "products": sum(mul_n.values()) / n}}


class Box:
    """One dealer's box as this rank holds it: its block of the shares resident in HBM plus what the checks need."""
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--participants", dest="n", type=int, default=65536, help="participants per GPU")
    ap.add_argument("--threshold", dest="t", type=int, default=256, help="threshold")
    ap.add_argument("--distinct-boxes", type=int, default=-1, help="different dealers' boxes cycled through by the timed steps "
                    "(-1: min(steps, 24); every box has its own polynomial, witnesses, challenge)")
    ap.add_argument("--registered-keys", type=int, default=1,
                    help="also time the opt-in registered-key variant at N=1 (0: skip)")
    ap.add_argument("--wb-shares", type=int, default=-1, help="share boxes in the verify_share figure (-1: 16384, 0: skip)")
    ap.add_argument("--ec-boxes", type=int, default=64, help="boxes timed per curve group for the `ec` objects (0: skip)")
    ap.add_argument("--host-boxes", type=int, default=20, help="boxes verified from HOST buffers (PCIe included) for the "
                    "`host_buffers` figure (0: skip)")
    ap.add_argument("--scaling", choices=("weak", "strong", "both"), default="both",
                    help="N > 1: `weak` -- every GPU its own --participants shares of every box (the box grows with N; `value`, as in every "
                         "round); `strong` -- the metric's box of --participants shares split over the N GPUs (`value` is then that, "
                         "`scaling` says so); `both` (default) -- `value` stays the weak figure and the line also carries `strong.value`")
    ap.add_argument("--drop-in-threads", default="4,8,12,16", help="host threads of the `drop_in` figure: each calls the ONE-box entry point "
                    "mpvss_modp_verify_distribution on the shared context (comma-separated counts; empty or 0: skip)")
    ap.add_argument("--ec-n", type=int, default=65536)
    ap.add_argument("--ec-t", type=int, default=256)
    ap.add_argument("--lone-boxes", type=int, default=2, help="boxes verified one at a time after the timed region "
                                                              "(isolated kernel durations for the roofline; 0: skip)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="shares timed on the all-core CPU port (-1: 1 per core, 0: skip all CPU legs)")
    ap.add_argument("--steady-steps", type=int, default=100, help="boxes of the `value_steady_state` figure at N=1, after the timed "
                    "region (0: skip)")
    ap.add_argument("--config-boxes", type=int, default=-1, help="boxes timed for the other BASELINE shapes in `configs` "
                    "(C2 n=4096 t=64, one GPU's slice of C5 n=131072 t=1024; -1: 48 / 6, 0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N needs N ranks: run `python bench.py --gpus N` directly (it starts them) or under torch.distributed.run")
    import torch.distributed as dist
    # MPVSS_BENCH_SMOKE_ONE_GPU=1: run every rank on cuda:0 with the gloo backend -- only to exercise the
    # multi-rank control flow on a single-GPU box; real runs use one GPU per rank and RCCL ("nccl").
    smoke_one_gpu = os.environ.get("MPVSS_BENCH_SMOKE_ONE_GPU") == "1"
    if smoke_one_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if smoke_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # The running hash state (128 bytes per box and hop) is produced and consumed by host code: it travels over a
    # gloo (CPU) group, so that a hop never waits for a free wave slot on a GPU that is saturated with long-running
    # workgroups.  The barrier, the max-reduction of the timing and the per-box all-gather of the shares' well-formedness
    # bytes (device data) go over the default group: RCCL in a real run.
    commdev = torch.device("cpu")
    chain = dist.new_group(backend="gloo") if (world > 1 and not smoke_one_gpu) else None

    if world > 1:
        os.environ.setdefault("MPVSS_PIPELINED", "1")    # blocks driven from here are never alone on the GPU for long
    eng = capi.Engine(local_rank)     # raises if the HIP library or the GPU is missing
    lib, ctx = eng.lib, eng.ctx
    n, t = args.n, args.t
    strong_main = world > 1 and args.scaling == "strong"
    if strong_main:           # the metric's fixed box over N GPUs: contiguous blocks of n / N positions (participant.rs:408-448)
        if args.n % world:
            raise SystemExit("--scaling strong needs --participants divisible by the number of GPUs")
        n = args.n // world
    n_total = n * world
    lo = rank * n
    import hashlib

    def dev_u8(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    def vp(tensor):
        return C.c_void_p(tensor.data_ptr())

    def chain_digest(tag, inter):
        """SHA-256 of the dealer's transcript over ALL ranks' blocks in rank order (the 128-byte state goes rank to rank)"""
        if world == 1:
            return capi.transcript_verdict(capi.transcript_absorb(capi.transcript_init(), inter), bytes(EB))[1]
        if rank == 0:
            state = capi.transcript_init()
        else:
            buf = torch.empty(capi.TRANSCRIPT_STATE_BYTES, dtype=torch.uint8, device=commdev)
            dist.recv(buf, src=rank - 1, group=chain, tag=tag)
            state = bytes(buf.cpu().numpy().tobytes())
        state = capi.transcript_absorb(state, inter)
        if rank + 1 < world:
            dist.send(torch.frombuffer(bytearray(state), dtype=torch.uint8).to(commdev), dst=rank + 1, group=chain, tag=tag)
            dg = torch.zeros(32, dtype=torch.uint8, device=commdev)
        else:
            dg = torch.frombuffer(bytearray(capi.transcript_verdict(state, bytes(EB))[1]), dtype=torch.uint8).to(commdev)
        dist.broadcast(dg, src=world - 1, group=chain)
        return bytes(dg.cpu().numpy().tobytes())

    # ---------------- synthetic workload (deterministic) ----------------
    # The participants -- private keys, public keys, positions -- are the same for every box (long-lived keys); every
    # dealer's box has its own polynomial, witnesses, shares, challenge and responses.
    def make_participants(n_, rank_seed):
        rng = random.Random(rank_seed)
        privs = [keygen(rng) for _ in range(n_)]
        wits = [keygen(rng) for _ in range(n_)]
        return privs, wits

    def make_box(b, n_, t_, positions, pubkeys, wit_bytes0, python_scalars, tag_base=0):
        """box number b of the run: polynomial from SEED + b, witnesses w_i * k_b (k_b a unit, so gcd(w, q-1) = 1 stays),
        dealt by the engine (oracle-checked batch calls), challenge and responses as participant.rs:251-264."""
        bx = Box()
        rng_c = random.Random(SEED + 7919 * b)            # polynomial coefficients: same on every rank
        bx.coeffs = [rng_c.randrange(ORDER) for _ in range(t_)]
        coeff_bytes = b"".join(fx(a) for a in bx.coeffs)
        if python_scalars:       # P(i) mod (q-1) with Python integers (polynomial.rs:50-58 evaluates over Z, the caller reduces)
            pvals = []
            rc = list(reversed(bx.coeffs))
            for i in positions:
                acc = 0
                for a in rc:
                    acc = acc * i + a
                pvals.append(acc % ORDER)
            bx.pv_bytes = b"".join(map(fx, pvals))
        else:                    # the same through the C ABI's scalar side (checked against Python integers on box 0)
            bx.pv_bytes = capi.poly_eval(0, coeff_bytes, positions)
        if b == 0:
            bx.wit_bytes = wit_bytes0
        else:                    # w' = w * k_b = 0 - w * (order - k_b)
            k_b = keygen(rng_c)
            bx.wit_bytes = capi.dleq_responses(0, bytes(len(wit_bytes0)), wit_bytes0, fx(ORDER - k_b))
        bx.commitments = eng.batch_exp_fixed_base(fx(4), coeff_bytes)                     # C_j = g^a_j
        dres = eng.distribute(bx.commitments, positions, pubkeys, bx.pv_bytes, bx.wit_bytes)
        bx.shares = dres["Y"]
        if world == 1:
            bx.dealer_digest = dres["digest"]
        else:
            inter = bytearray()
            for i in range(n_):
                s = slice(i * EB, (i + 1) * EB)
                inter += dres["X"][s] + dres["Y"][s] + dres["a1"][s] + dres["a2"][s]
            bx.dealer_digest = chain_digest((1 << 20) + tag_base + b, bytes(inter))
        bx.c = int.from_bytes(hashlib.sha256(bx.dealer_digest).digest(), "big") % ((Q - 1) // 2)    # modp.rs:142-148
        bx.challenge = fx(bx.c)
        if python_scalars:
            bx.responses = b"".join(fx((int.from_bytes(bx.wit_bytes[i * EB:(i + 1) * EB], "big") - (p * bx.c) % ORDER) % ORDER)
                                    for i, p in enumerate(pvals))                         # dleq.rs:42-50
        else:
            bx.responses = capi.dleq_responses(0, bx.wit_bytes, bx.pv_bytes, bx.challenge)
        bx.dres = dres if b == 0 else None
        bx.d_cm, bx.d_sh, bx.d_rs = dev_u8(bx.commitments), dev_u8(bx.shares), dev_u8(bx.responses)
        bx.ch_buf = (C.c_uint8 * EB).from_buffer_copy(bx.challenge)
        bx.n, bx.t = n_, t_
        return bx

    t_setup = time.time()
    privs, wits = make_participants(n, SEED * 1000003 + rank)
    positions = list(range(lo + 1, lo + n + 1))
    pubkeys = eng.batch_exp_fixed_base(fx(2), b"".join(fx(k) for k in privs))             # y_i = G^x_i
    wit_bytes = b"".join(map(fx, wits))
    d_pk = dev_u8(pubkeys)
    d_pos = torch.tensor(positions, dtype=torch.int64, device=dev)
    n_distinct = args.distinct_boxes if args.distinct_boxes > 0 else max(1, min(args.steps, 24))
    boxes = [make_box(0, n, t, positions, pubkeys, wit_bytes, True)]
    boxes += [make_box(b, n, t, positions, pubkeys, wit_bytes, False) for b in range(1, n_distinct)]
    box0 = boxes[0]
    commitments, shares, responses, challenge, dres = box0.commitments, box0.shares, box0.responses, box0.challenge, box0.dres
    pv_bytes, coeffs, dealer_digest, c = box0.pv_bytes, box0.coeffs, box0.dealer_digest, box0.c
    d_cm, d_sh, d_rs, ch_buf = box0.d_cm, box0.d_sh, box0.d_rs, box0.ch_buf
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    keyset = [None]      # set for the secondary "registered keys" figure only

    class Cur:           # the participants and boxes the step functions below work on (the headline's; bench_shape swaps them)
        pass
    cur = Cur()
    cur.boxes, cur.d_pk, cur.d_pos, cur.n = boxes, d_pk, d_pos, n

    # N > 1: every box also leaves one well-formedness byte per share in HBM (mpvss_modp_verify_block_compute_flags), and
    # the ranks all-gather those bytes once per box over the default process group -- RCCL over xGMI in a real run
    # (the per-share verdict bytes of W_A; they never enter the box verdict, which is the transcript digest).
    rccl = {"issued": 0, "works": collections.deque(), "bad": 0, "ring": 64, "done": set(), "next": 0}

    def rccl_buffers(n_):
        cdev = commdev if smoke_one_gpu else dev
        rccl["mine"] = [torch.zeros(n_, dtype=torch.uint8, device=dev) for _ in range(rccl["ring"])]
        rccl["all"] = [torch.zeros(n_ * world, dtype=torch.uint8, device=cdev) for _ in range(rccl["ring"])]
        torch.cuda.synchronize()      # the zero fills run on torch's stream: they must not land on top of flags a block's own stream wrote
    if world > 1:
        rccl_buffers(n)
    enq_boxes = []                  # box behind every block enqueued on this rank, in enqueue (= claim) order

    def rccl_reap(limit):
        while len(rccl["works"]) > limit:
            work, slot = rccl["works"].popleft()
            work.wait()
            rccl["bad"] += int((rccl["all"][slot] != 1).sum().item())

    def rccl_gather_flags(seq):
        """box `seq` has been absorbed on this rank: its flag bytes are final -> one all-gather (asynchronous; at most
        ring/2 outstanding, so a buffer is never reused while its collective is in flight)"""
        slot = seq % rccl["ring"]
        rccl_reap(rccl["ring"] // 2 - 1)
        src = rccl["mine"][slot].cpu() if smoke_one_gpu else rccl["mine"][slot]
        rccl["works"].append((dist.all_gather_into_tensor(rccl["all"][slot], src, async_op=True), slot))
        rccl["issued"] += 1

    def compute_block(bx):
        enq_boxes.append(bx)
        if world > 1 and keyset[0] is None:
            slot = (len(enq_boxes) - 1) % rccl["ring"]
            eng._check(lib.mpvss_modp_verify_block_compute_flags(ctx, capi.MPVSS_DEVICE, vp(bx.d_cm), bx.t, vp(cur.d_pos), vp(cur.d_pk), vp(bx.d_sh),
                                                                 vp(bx.d_rs), bx.n, C.cast(bx.ch_buf, C.c_void_p), vp(rccl["mine"][slot])),
                       "verify_block_compute_flags")
            return
        if keyset[0] is not None:
            rcode = lib.mpvss_modp_verify_block_compute_keyset(ctx, capi.MPVSS_DEVICE, vp(bx.d_cm), bx.t, vp(cur.d_pos), keyset[0], 0,
                                                               vp(bx.d_sh), vp(bx.d_rs), bx.n, C.cast(bx.ch_buf, C.c_void_p))
        else:
            rcode = lib.mpvss_modp_verify_block_compute(ctx, capi.MPVSS_DEVICE, vp(bx.d_cm), bx.t, vp(cur.d_pos), vp(cur.d_pk),
                                                        vp(bx.d_sh), vp(bx.d_rs), bx.n, C.cast(bx.ch_buf, C.c_void_p))
        eng._check(rcode, "verify_block_compute")

    hash_pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(HASH_THREADS, 1))
    claim_lock = threading.Lock()
    box_seq = [0]                   # boxes claimed so far on this rank (every rank counts the same boxes in the same order)

    def finish_block():
        """Absorb the oldest in-flight block into its transcript; returns (box number, (verdict, digest)) -- None instead
        of the pair on the ranks that only pass the state on.  Runs on the hash threads, several boxes at a time: the boxes are independent
        transcripts, only the ranks of ONE box form a chain.  claim -> (state of this box from the previous rank) ->
        wait for the GPU and hash -> (state to the next rank); the message tag is the box's sequence number."""
        with claim_lock:
            ticket = eng.block_claim()
            seq = box_seq[0]
            box_seq[0] += 1
            bx = enq_boxes[seq]
        if world == 1 or rank == 0:
            state = capi.transcript_init()
        else:
            buf = torch.empty(capi.TRANSCRIPT_STATE_BYTES, dtype=torch.uint8, device=commdev)
            dist.recv(buf, src=rank - 1, group=chain, tag=seq)
            state = bytes(buf.cpu().numpy().tobytes())
        state = eng.verify_block_absorb_claimed(ticket, state)   # waits for this block's GPU work, then hashes it
        if world > 1 and rank + 1 < world:
            # hand the state on; only the last rank knows the verdicts, they are broadcast once at the end of run_steps()
            dist.send(torch.frombuffer(bytearray(state), dtype=torch.uint8).to(commdev), dst=rank + 1, group=chain, tag=seq)
            return seq, None
        return seq, capi.transcript_verdict(state, bx.challenge)

    # Boxes in flight per rank when the blocks are driven from here (N > 1): a box stays in flight until its transcript is
    # absorbed, and on rank g that waits for the g ranks before it (40 ms of SHA-256 each) on top of the GPU work -- with the
    # N = 1 figure of 8 the later ranks of an eight-rank box would run dry (8 boxes per ~0.8 s of latency).  One more box per
    # 58 ms of chain latency; an explicit MPVSS_BENCH_DEPTH is taken as it is.
    if "MPVSS_BENCH_DEPTH" in os.environ:
        RANK_DEPTH = min(PIPE_DEPTH, 8)
    else:
        RANK_DEPTH = min(8 + (7 * (world - 1) + 9) // 10 + (2 if world > 1 else 0), capi.BLOCK_SLOTS - max(HASH_THREADS, 1) - 4)

    hash_threads_now = [HASH_THREADS]          # (the registered-keys leg raises it: its boxes are shorter on the GPU than their hash)

    def run_steps_many(seq_boxes, depth):
        """N = 1: complete verifications of the given boxes in ONE library call (mpvss_modp_verify_many): the calling thread
        enqueues the GPU work of up to `depth` boxes ahead, HASH_THREADS library threads absorb (wait for and hash)
        the boxes in order.  No Python in the loop."""
        ks = keyset[0]
        k = len(seq_boxes)
        arr = (capi.ModpBox * k)(*[capi.ModpBox(bx.d_cm.data_ptr(), bx.t, cur.d_pos.data_ptr(), None if ks is not None else cur.d_pk.data_ptr(),
                                                bx.d_sh.data_ptr(), bx.d_rs.data_ptr(), bx.n, C.cast(bx.ch_buf, C.c_void_p), ks, 0)
                                   for bx in seq_boxes])
        verdicts = (C.c_int * k)()
        digests = (C.c_uint8 * (32 * k))()
        eng._check(lib.mpvss_modp_verify_many(ctx, capi.MPVSS_DEVICE, arr, k, depth, max(hash_threads_now[0], 1), verdicts,
                                              C.cast(digests, C.c_void_p)), "verify_many")
        raw = bytes(digests)
        return [(bool(verdicts[i]), raw[32 * i:32 * i + 32]) for i in range(k)]

    CHAINED = os.environ.get("MPVSS_BENCH_CHAINED", "1")      # 0: blocks driven from Python threads (round 3); 2: the chained call at N = 1 too

    def run_steps_chained(seq_boxes, depth):
        """N > 1: the SAME library pipeline as N = 1 (mpvss_modp_verify_many_chained: the calling thread enqueues, library threads
        absorb), with the 128-byte running state of every box travelling rank to rank through two callbacks the library makes
        around each box's absorb: state_in receives this rank's starting state of the box from the rank before (gloo, tag = box
        number), state_out sends it on.  One more thread issues the per-box all-gather of the well-formedness bytes in box order."""
        import queue
        k = len(seq_boxes)
        seq0 = len(enq_boxes)
        enq_boxes.extend(seq_boxes)
        arr = (capi.ModpBox * k)(*[capi.ModpBox(bx.d_cm.data_ptr(), bx.t, cur.d_pos.data_ptr(), cur.d_pk.data_ptr(), bx.d_sh.data_ptr(),
                                                bx.d_rs.data_ptr(), bx.n, C.cast(bx.ch_buf, C.c_void_p), None, 0) for bx in seq_boxes])
        wf = None
        if world > 1:
            wf = (C.c_void_p * k)(*[rccl["mine"][(seq0 + i) % rccl["ring"]].data_ptr() for i in range(k)])
        verdicts = (C.c_int * k)()
        digests = (C.c_uint8 * (32 * k))()
        done_q, errors = queue.Queue(), []
        SB = capi.TRANSCRIPT_STATE_BYTES

        def cb_in(user, box, state, ok):
            try:
                buf = torch.empty(SB + 1, dtype=torch.uint8, device=commdev)
                dist.recv(buf, src=rank - 1, group=chain, tag=seq0 + box)
                raw = bytes(buf.cpu().numpy().tobytes())
                C.memmove(state, raw[1:], SB)
                return 0 if raw[0] else 1
            except Exception as exc:      # noqa: BLE001 - reported after the call
                errors.append(exc)
                return 1

        def cb_out(user, box, state, ok):
            try:
                if world > 1 and rank + 1 < world:
                    msg = bytes([1 if ok else 0]) + C.string_at(state, SB)
                    dist.send(torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(commdev), dst=rank + 1, group=chain, tag=seq0 + box)
            except Exception as exc:      # noqa: BLE001
                errors.append(exc)
            done_q.put(seq0 + box)        # this rank's flags of the box are final
            return 0

        def gatherer():
            torch.cuda.set_device(dev)
            pend, nxt = set(), seq0
            while nxt < seq0 + k:
                pend.add(done_q.get())
                while nxt in pend:
                    pend.discard(nxt)
                    if world > 1:
                        rccl_gather_flags(nxt)
                    nxt += 1

        gt = threading.Thread(target=gatherer)
        gt.start()
        c_in = capi.CHAIN_CB(cb_in) if (world > 1 and rank > 0) else capi.CHAIN_CB()
        c_out = capi.CHAIN_CB(cb_out)
        threads = min(32, max(HASH_THREADS, 1) + (min(world - 1, 8) if world > 1 else 0))
        rc = lib.mpvss_modp_verify_many_chained(ctx, capi.MPVSS_DEVICE, arr, k, depth, threads, wf, c_in, c_out, None, verdicts,
                                                C.cast(digests, C.c_void_p))
        gt.join()
        eng._check(rc, "verify_many_chained")
        if errors:
            raise errors[0]
        raw = bytes(digests)
        return [(bool(verdicts[i]), raw[32 * i:32 * i + 32]) for i in range(k)]

    def run_steps(k, depth=None, first=0):
        """k complete verifications -- step s verifies box (first + s) mod (number of distinct boxes) -- software-pipelined:
        up to `depth` boxes have their GPU work enqueued while host threads hash the oldest ones.  On one GPU the whole
        pipeline runs inside the library (run_steps_many); with several ranks the running hash state of every box travels
        rank to rank: the same library pipeline with two callbacks around every box's absorb (run_steps_chained; with
        MPVSS_BENCH_CHAINED=0 the blocks are driven from Python threads instead: compute / claim / absorb_claimed).
        Returns [(verdict, digest, box)]."""
        seq_boxes = [cur.boxes[(first + s) % len(cur.boxes)] for s in range(k)]
        if world == 1 and USE_VERIFY_MANY and k > 0 and (CHAINED != "2" or keyset[0] is not None):
            res = run_steps_many(seq_boxes, min(depth or PIPE_DEPTH, capi.BLOCK_SLOTS))
            return [(v, d, bx) for (v, d), bx in zip(res, seq_boxes)]
        if k > 0 and keyset[0] is None and ((world > 1 and CHAINED != "0") or CHAINED == "2"):
            results = run_steps_chained(seq_boxes, min(depth or (PIPE_DEPTH if world == 1 else RANK_DEPTH), capi.BLOCK_SLOTS - 8))
            if world > 1:
                rccl_reap(0)
                assert rccl["bad"] == 0, "a share of an honest box was reported as not well-formed"
                if rank == world - 1:
                    flat = b"".join(bytes([int(v)]) + d for v, d in results)
                    out = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(commdev)
                else:
                    out = torch.zeros(33 * k, dtype=torch.uint8, device=commdev)
                dist.broadcast(out, src=world - 1, group=chain)
                raw = bytes(out.cpu().numpy().tobytes())
                results = [(bool(raw[33 * i]), raw[33 * i + 1:33 * i + 33]) for i in range(k)]
            return [(v, d, bx) for (v, d), bx in zip(results, seq_boxes)]
        # absorbing threads hold the oldest blocks, so leave them slack in the ring of block slots
        depth = min(depth or RANK_DEPTH, capi.BLOCK_SLOTS - max(HASH_THREADS, 1))
        results = []
        issued = 0
        seq0 = len(enq_boxes)
        while issued < min(depth, k):
            compute_block(seq_boxes[issued])
            issued += 1
        pending = collections.deque(hash_pool.submit(finish_block) for _ in range(issued))
        while pending:
            results.append(pending.popleft().result())
            if world > 1:                    # gather in box order: the same sequence of collectives on every rank
                rccl["done"].add(results[-1][0])
                while rccl["next"] in rccl["done"]:
                    rccl["done"].discard(rccl["next"])
                    rccl_gather_flags(rccl["next"])
                    rccl["next"] += 1
            if issued < k:
                compute_block(seq_boxes[issued])
                issued += 1
                pending.append(hash_pool.submit(finish_block))
        results = [r for _, r in sorted(results, key=lambda sr: sr[0])]      # threads claim boxes in order, finish in any
        assert len(enq_boxes) == seq0 + k
        if world > 1:
            rccl_reap(0)
            assert rccl["bad"] == 0, "a share of an honest box was reported as not well-formed"
        if world > 1:                      # one broadcast of all k verdicts and digests from the last rank
            if rank == world - 1:
                flat = b"".join(bytes([int(v)]) + d for v, d in results)
                out = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(commdev)
            else:
                out = torch.zeros(33 * k, dtype=torch.uint8, device=commdev)
            dist.broadcast(out, src=world - 1, group=chain)
            raw = bytes(out.cpu().numpy().tobytes())
            results = [(bool(raw[33 * i]), raw[33 * i + 1:33 * i + 33]) for i in range(k)]
        return [(v, d, bx) for (v, d), bx in zip(results, seq_boxes)]

    def gate(results, what):
        for verdict, digest, bx in results:
            assert verdict is True and digest == bx.dealer_digest, f"parity gate failed ({what})"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Allocation pass: every block slot the pipeline will use gets its workspace (about 1.6 GB of HBM), stream pair and
    # pinned staging now, one box per slot -- first-use allocation (hipMalloc, page pinning) is set-up, not verification,
    # and must not fall into the timed region when W is smaller than the number of boxes in flight.
    # (slots are reused most-recently-released first, so the pass enqueues as many boxes AT ONCE as the pipeline can
    # ever have in flight: boxes with GPU work pending + boxes being hashed + slack)
    slot_init = min((PIPE_DEPTH if (world == 1 and USE_VERIFY_MANY) else RANK_DEPTH) + max(HASH_THREADS, 1) + 4,
                    capi.BLOCK_SLOTS - 1)
    gate(run_steps(slot_init, depth=slot_init), "slot initialisation")
    if args.warmup > 0:
        gate(run_steps(args.warmup, first=len(boxes) - args.warmup % len(boxes)), "warm-up")
    eng.pipeline_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    results = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    pst = eng.pipeline_stats(reset=True)          # host and kernel accounting of exactly the timed steps
    gate(results, "timed steps: a GPU box did not verify")
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=commdev if smoke_one_gpu else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    value = n_total * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    nb = max(pst["blocks"], 1)
    x_ms, a1_ms, tb_ms, a2_ms = (pst["kernel_ms"][k] / nb for k in (0, 1, 2, 3))    # overlapped: boxes share the chip
    a2_n = max(pst["kernel_launches"][3] / nb, 1.0)   # a2 launches per step (1 unless the box is split)
    a2_kernel = "k_modp_dual_exp_w6_pair" if PAIR_MASK & 1 else "k_modp_dual_exp_w6"
    shares_per_a2_launch = n / a2_n
    # Isolated launches: the same box verified alone (one box in flight, nothing else on the GPU) after the timed
    # region -- the duration of a launch that has the chip to itself is what a roofline can be read from; in the timed
    # region up to PIPE_DEPTH boxes share the chip and a launch's wall duration exceeds the step time.
    lone = None
    if world == 1 and args.lone_boxes > 0:
        torch.cuda.synchronize()
        gate(run_steps(args.lone_boxes, depth=1), "lone box")
        lst = eng.pipeline_stats(reset=True)
        lb = max(lst["blocks"], 1)
        lone = {"x_path": lst["kernel_ms"][0] / lb, "a1_comb_dual_exp": lst["kernel_ms"][1] / lb,
                "tables": lst["kernel_ms"][2] / lb, "a2_dual_exp": lst["kernel_ms"][3] / lb,
                "a2_launches": max(lst["kernel_launches"][3] / lb, 1.0), "boxes": lb}
    # ... and the dominant kernel entirely alone on the chip: the verifier's commitments of the same shares through the
    # synchronous entry point (mpvss_modp_dleq_commitments: a1, then the tables of y and Y, then a2 = y^r Y^c -- the same
    # k_modp_dual_exp_w6 launch on the same operands, with nothing beside it; even a lone box runs its X path beside a2)
    alone_ms = None
    if world == 1 and args.lone_boxes > 0 and keyset[0] is None:
        d_X = dev_u8(dres["X"])
        o1 = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        o2 = torch.empty(n * EB, dtype=torch.uint8, device=dev)
        g_host = (C.c_uint8 * EB).from_buffer_copy(fx(4))
        for _ in range(2):
            eng._check(lib.mpvss_modp_dleq_commitments(ctx, capi.MPVSS_DEVICE, C.cast(g_host, C.c_void_p), vp(d_X), vp(d_pk), vp(d_sh),
                                                       vp(d_rs), C.cast(ch_buf, C.c_void_p), 0, n, vp(o1), vp(o2)), "dleq_commitments")
        alone_ms = eng.kernel_ms(3) / max(eng.kernel_launches(3), 1)
        assert bytes(o2.cpu().numpy().tobytes()) == dres["a2"] and bytes(o1.cpu().numpy().tobytes()) == dres["a1"], \
            "verifier commitments differ from the dealer's"
        del d_X, o1, o2
    # steady state: the same pipeline over many more boxes (the K-step region ends with the boxes in flight finishing together
    # and their hashes running with nothing beside them; over 100 boxes that tail no longer shows)
    steady = None
    if world == 1 and args.steady_steps > 0:
        barrier()
        ts0 = time.perf_counter()
        res_ss = run_steps(args.steady_steps)
        barrier()
        ss_s = time.perf_counter() - ts0
        gate(res_ss, "steady-state steps")
        steady = {"value": n * args.steady_steps / ss_s, "steps": args.steady_steps, "ms_per_step": ss_s / args.steady_steps * 1e3}
        eng.pipeline_stats(reset=True)
    # what this device sustains of the kernels' basic instructions right now (4 waves per SIMD issuing back to back for ~40 ms)
    probe = None
    if world == 1:
        p0, p1 = eng.issue_probe(0, 40.0), eng.issue_probe(1, 40.0)
        probe = {"mad64_insts_per_s": p0["insts_per_s"], "mad64_shader_clock_ghz": p0["shader_clock_ghz"],
                 "mad64_cycles_per_inst": 1024 * p0["shader_clock_ghz"] * 1e9 / p0["insts_per_s"] if p0["insts_per_s"] else None,
                 "alu32_insts_per_s": p1["insts_per_s"], "alu32_shader_clock_ghz": p1["shader_clock_ghz"],
                 "how": "mpvss_issue_probe, live in this run after the timed region: 4 waves per SIMD on every CU issue v_mad_u64_u32 "
                        "(mad64) / v_add3_u32, v_and_b32, v_lshl_add_u32 (alu32) back to back for ~40 ms; the clock is s_memtime "
                        "against s_memrealtime inside the kernel"}
    # the other instruction classes of the kernels' mix, same probe (kinds 2-5): seconds per wave-instruction per SIMD
    tau = None
    if probe and probe["mad64_insts_per_s"]:
        rates = {"mad64": probe["mad64_insts_per_s"], "alu32": probe["alu32_insts_per_s"]}
        for kind, cname in ((2, "shift64"), (3, "swap"), (4, "mov64"), (5, "vop2")):
            pk = eng.issue_probe(kind, 30.0)
            rates[cname] = pk["insts_per_s"]
            probe[cname + "_insts_per_s"] = pk["insts_per_s"]
            probe[cname + "_shader_clock_ghz"] = pk["shader_clock_ghz"]
        tau = {c: 1024.0 / r for c, r in rates.items() if r}
    a2_launch_ms_overlapped = a2_ms / a2_n
    a2_one_box_ms = lone["a2_dual_exp"] / lone["a2_launches"] if lone else None
    a2_launch_ms = alone_ms if alone_ms else (a2_one_box_ms if a2_one_box_ms else a2_launch_ms_overlapped)
    if alone_ms or a2_one_box_ms:          # those launches cover all n shares; in the pipeline a box's a2 goes out in `a2_n` share ranges
        shares_per_a2_launch = n

    # work accounting: Montgomery products the kernels execute per step on this rank (averaged over the timed boxes)
    wk = modp_work(n, t, positions, [bx.c for _, _, bx in results])
    mm_total, w6, a2_products = wk["mm_total"], wk["w6"], wk["a2_products"]
    achieved_modmul = mm_total / (ms_per_step * 1e-3)         # against the step's wall time (kernels overlap)
    peak_modmul = PEAK_MODMUL_PER_S

    result = {
        "metric": "DLEQ share verifications/sec, 2048-bit MODP, n=65536 t=256",
        "value": value,
        "value_steady_state": steady["value"] if steady else None,
        "steady_state": steady,
        "unit": "share verifications/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong_main else "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "dtype_detail": "u32 limbs (radix 2^29), u64 accumulators; the Montgomery reduction as int8 digit products on the matrix cores",
        "data": "synthetic",
        "config": {"workload": f"ModpGroup 2048-bit verify_distribution_shares n={n} t={t} per GPU",
                   "workload_detail": f"{n_total} participants in the box, honest-dealer boxes, inputs resident in HBM",
                   "n_per_gpu": n, "t": t, "parallelism": f"participants sharded x{world}",
                   "distinct_boxes": len(boxes),
                   "boxes": "every timed step verifies another dealer's box (own polynomial, witnesses, shares, challenge, "
                            "responses) against the same participants' public keys"
                            + ("" if len(boxes) >= args.steps else f"; {len(boxes)} boxes cycled through {args.steps} steps")},
        "roofline": {
            "bound": "hbm",
            "kernel": a2_kernel,
            "achieved": ALGO_BYTES_PER_SHARE * shares_per_a2_launch / (a2_launch_ms * 1e-3) / 1e9 if a2_launch_ms > 0 else None,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": (ALGO_BYTES_PER_SHARE * shares_per_a2_launch / (a2_launch_ms * 1e-3) / 1e9) / HBM_PEAK_GBPS if a2_launch_ms > 0 else None,
            "traffic": None,
            "kernel_ms": a2_launch_ms,
            "kernel_ms_is": ("the launch alone on the chip (same operands through mpvss_modp_dleq_commitments, live after the "
                             "timed region)" if alone_ms else
                             "launch of a box that is alone in flight (its own X path runs beside it)" if lone
                             else "overlapped launch (timed region)"),
            "kernel_ms_one_box_in_flight": a2_one_box_ms,
            "kernel_ms_overlapped": a2_launch_ms_overlapped,
            "launches_per_step": a2_n,
            "shares_per_launch": shares_per_a2_launch,
        },
        "compute": {
            "bound": "valu issue",
            "achieved": wk["slots"] / (ms_per_step * 1e-3), "peak": PEAK_VALU_SLOTS_PER_S, "peak_is": PEAK_SOURCE,
            "unit": "VALU wave-instruction issue slots/s (all kernels / step wall time; one slot = one wave64 VALU instruction on one "
                    "SIMD; an MFMA holds the issue for two slots)",
            "frac": wk["slots"] / (ms_per_step * 1e-3) / PEAK_VALU_SLOTS_PER_S,
            "sustained_probe": probe,
            **({"frac_mix_weighted": mix_seconds_per_simd(wk["by_layout"], tau) / 1024.0 / (ms_per_step * 1e-3),
                "peak_mix_weighted": wk["slots"] / (mix_seconds_per_simd(wk["by_layout"], tau) / 1024.0),
                "mix_weighted_is": "the time the step's instructions need when every class (v_mad_u64_u32 / 32-bit VOP3 / 64-bit shifts / "
                                   "v_permlane32_swap / v_mov_b64; an MFMA = two 32-bit slots) issues at the rate mpvss_issue_probe measured "
                                   "for it in this run, over the step's wall time; peak_mix_weighted = the same as issue slots/s"}
               if tau else {}),
            "vs_sustained_mad64": (wk["slots"] / (ms_per_step * 1e-3) / probe["mad64_insts_per_s"]) if probe and probe["mad64_insts_per_s"] else None,
            "vs_sustained_mad64_is": "achieved slots/s over the v_mad_u64_u32 rate the probe sustained in this run (a ratio of two "
                                     "measurements: a kernel whose mix is lighter than pure 64-bit mads can exceed 1)",
            "slots_source": "instruction counts per Montgomery operation from the ISA (tools/mfma_mont/count_isa.py for the pair layout, "
                            "bn_quad.h row counts for the quad layout) x the operations the kernels execute for this run's challenges",
            "valu_slots_per_share": wk["slots"] / n,
            "slots_per_operation": {"quad_product": QUAD_MUL_SLOTS, "quad_squaring": QUAD_SQ_SLOTS, "pair_product": PAIR_MUL_SLOTS,
                                    "pair_squaring": PAIR_SQ_SLOTS,
                                    "note": "per number; quad = bn_quad.h (VALU only), pair = bn_pair.h (Montgomery reduction on the "
                                            "matrix cores); MPVSS_PAIR bit mask in use: %d" % PAIR_MASK},
            "operations_per_share": wk["ops"],
            "modmul_per_share": mm_total / n,
            "modmul_equivalents": {"achieved": achieved_modmul, "peak_valu_only": peak_modmul, "ratio_to_valu_only_product_rate": achieved_modmul / peak_modmul,
                                   "note": "rounds 1-2 accounting: product equivalents per second (a squaring 0.764) relative to what the VALU-only "
                                           "product reached in round 1's microbenchmark (3.05 G/s) -- a ratio to another formulation's rate, not a "
                                           "fraction of a peak: with the reduction on the matrix cores it may exceed 1"},
            "x_path": wk["x_path"],
            "a2_kernel_alone": ({"ms": alone_ms, "valu_slots_per_s": wk["a2_slots"] / (alone_ms * 1e-3),
                                 "frac": wk["a2_slots"] / (alone_ms * 1e-3) / PEAK_VALU_SLOTS_PER_S,
                                 "vs_sustained_mad64": (wk["a2_slots"] / (alone_ms * 1e-3) / probe["mad64_insts_per_s"]) if probe else None,
                                 "frac_mix_weighted": (mix_seconds_per_simd(wk["a2_by_layout"], tau) / 1024.0 / (alone_ms * 1e-3)) if tau else None,
                                 "products_per_share": a2_products} if alone_ms else None),
            "kernel_ms_sums": {"x_path": x_ms, "a1_comb_dual_exp": a1_ms, "a2_dual_exp": a2_ms, "tables": tb_ms,
                               "note": "per-kind sums of launch durations per step in the timed region; boxes and kinds "
                                       "overlap, so the sums exceed the step time"},
            "kernel_ms_isolated": lone,
        },
        "host": {"sha_ni": bool(lib.mpvss_sha256_uses_shani()),
                 "per_box_ms": {"enqueue": pst["enqueue_ms"] / nb, "wait_for_gpu": pst["wait_ms"] / nb,
                                "sha256_transcript": pst["hash_ms"] / nb},
                 "hash_threads": max(HASH_THREADS, 1),
                 "boxes_in_flight": (min(PIPE_DEPTH, capi.BLOCK_SLOTS) if (world == 1 and USE_VERIFY_MANY)
                                     else min(RANK_DEPTH, capi.BLOCK_SLOTS - max(HASH_THREADS, 1))),
                 "pipeline": ("mpvss_modp_verify_many (library threads; boxes_in_flight = boxes with GPU work pending)"
                              if (world == 1 and USE_VERIFY_MANY and CHAINED != "2")
                              else "mpvss_modp_verify_many_chained (the same library pipeline; the 128-byte hash state of every box travels "
                                   "rank to rank through the state_in / state_out callbacks: gloo, tag = box)"
                              if ((world > 1 and CHAINED != "0") or CHAINED == "2")
                              else "verify_block_compute / block_claim / absorb_claimed from a Python thread pool"
                                   + ("; the hash state of every box travels rank to rank (gloo, tag = box)" if world > 1 else "")),
                 "slot_init_boxes": slot_init,
                 "hbm": (lambda fr_tot: {"bytes_in_use_on_this_rank": fr_tot[1] - fr_tot[0], "bytes_total": fr_tot[1],
                                         "note": "device memory in use on rank 0's GPU after the timed region (torch.cuda.mem_get_info): "
                                                 "block-slot workspaces of the boxes in flight, the wide combs (2.4 GB per generator), "
                                                 "inputs of the distinct boxes, torch's own pool"})(torch.cuda.mem_get_info(dev)),
                 "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                 "setup_s": setup_s},
    }
    if world > 1:
        result["rccl"] = {
            "backend": dist.get_backend(), "rccl_world_size": dist.get_world_size(),
            "data_collectives": rccl["issued"], "per_box": 1, "bytes_per_rank_per_box": n,
            "collective": "all_gather_into_tensor of the shares' well-formedness bytes (device tensors written by "
                          "mpvss_modp_verify_block_compute_flags), one per box and rank, asynchronous, inside the timed region; "
                          "every byte of every rank checked == 1",
            "also": "barrier and max-reduction of the timing over the same group; the 128-byte hash state per box and hop over gloo"}
    traffic_file = os.path.join(ROOT, TRAFFIC_FILE)
    if os.path.exists(traffic_file) and (n, t) == (65536, 256):     # the PMC run was taken on the headline shape
        try:
            result["roofline"]["traffic"] = json.load(open(traffic_file)).get(a2_kernel + "_bytes_per_launch")
            result["roofline"]["traffic_source"] = (f"{TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command "
                                                    "(tools/run_profiles.sh), a committed file -- not measured in this run")
        except Exception:
            pass

    # Everything below is secondary to the headline (`value`, `roofline`, `compute` are complete at this point): a failure
    # there -- an allocation that does not fit, a parity gate of a secondary figure -- must not cost the line; it is
    # reported in `secondary_error` and the process exits non-zero after printing.
    secondary_error = None
    try:
        # (Order: the Python-driven secondary legs of the MODP group first -- W_B, extract, the dealer --, then the CPU baselines, the
        # other shapes, host buffers, registered keys and the curve groups.  With the CPU legs and the large shapes ahead of them the
        # Python-driven legs read 15-20 % lower in the same process (0.92 against 1.10 M for verify_share; profiles/r04_wb_order.txt):
        # an artefact of the process's state, not of the library.)
        # ---------------- W_B: decrypted-share verifications (participant.rs:361-386), SURVEY 8(d) ----------------
        # Secondary figure, rank 0 at N=1 only, outside the timed region above: a bounded batch of share boxes
        # (built with the engine's own extract_shares), inputs resident in HBM, verdicts checked.
        if rank == 0 and world == 1 and args.wb_shares != 0:
            m = min(n, args.wb_shares if args.wb_shares > 0 else 16384)
            sl = slice(0, m * EB)
            rng_w = random.Random(SEED + 7)
            wit_b = [keygen(rng_w) for _ in range(m)]
            xinv = b"".join(fx(pow(x, -1, ORDER)) for x in privs[:m])
            wit_bytes_b = b"".join(map(fx, wit_b))
            S, cb = eng.extract_shares(pubkeys[sl], shares[sl], xinv, wit_bytes_b)            # also warms the path up
            t_x = time.perf_counter()
            S2, cb2 = eng.extract_shares(pubkeys[sl], shares[sl], xinv, wit_bytes_b)
            extract_s = time.perf_counter() - t_x
            assert (S2, cb2) == (S, cb)
            # block form: several batches in flight (compute forms e2 = w / x on host threads and enqueues; the challenge hash runs
            # on the device); 4 in flight, 10 timed
            # (buffers made once and the library called directly: what a compiled caller pays -- round 5's loop re-marshalled 16 MB of
            # Python bytes per batch inside the timed region, and formed e2 = w / x on host threads; both depended on the host)
            xb_in = [(C.c_uint8 * (m * EB)).from_buffer_copy(b) for b in (pubkeys[sl], shares[sl], xinv, wit_bytes_b)]
            xb_S, xb_c = (C.c_uint8 * (m * EB))(), (C.c_uint8 * (m * EB))()

            def extract_pipelined(count, inflight=4):
                issued = done = 0
                while done < count:
                    while issued < count and issued - done < inflight:
                        eng._check(lib.mpvss_modp_extract_shares_compute(ctx, xb_in[0], xb_in[1], xb_in[2], xb_in[3], m), "extract_shares_compute")
                        issued += 1
                    eng._check(lib.mpvss_modp_extract_shares_absorb(ctx, xb_S, xb_c), "extract_shares_absorb")
                    done += 1
                return bytes(xb_S), bytes(xb_c)
            assert extract_pipelined(4) == (S, cb), "extract_shares block form differs from the synchronous call"
            torch.cuda.synchronize()
            t_x = time.perf_counter()
            extract_pipelined(12)
            extract_blk_s = (time.perf_counter() - t_x) / 12
            result["extract_shares"] = {"value": m / extract_blk_s, "unit": "shares decrypted and proven/s", "batch": m,
                                        "batches_in_flight": 4, "value_synchronous_call": m / extract_s,
                                        "note": "extract_secret_share for `batch` participants per call, host buffers (participant.rs:294-353): "
                                                "S_i = Y_i^(1/x_i), a1 = G^w_i, a2 = S_i^w_i (S and a2 from one chain of squarings) and the "
                                                "per-share challenge hash; `value`: mpvss_modp_extract_shares_compute/_absorb, several batches "
                                                "in flight, challenge hash on the device (K7); `value_synchronous_call`: one "
                                                "mpvss_modp_extract_shares call, hash on the host"}
            rb = b"".join(fx((w - x * int.from_bytes(cb[i * EB:(i + 1) * EB], "big")) % ORDER)
                          for i, (w, x) in enumerate(zip(wit_b, privs[:m])))                      # dleq.rs:42-50
            d_S, d_cb, d_rb = dev_u8(S), dev_u8(cb), dev_u8(rb)
            verd = (C.c_uint8 * m)()
            torch.cuda.synchronize()
            reps = 3
            for it in range(reps + 1):
                if it == 1:
                    tw = time.perf_counter()
                eng._check(lib.mpvss_modp_verify_shares(ctx, capi.MPVSS_DEVICE, vp(d_pk), vp(d_S), vp(d_sh), vp(d_cb), vp(d_rb),
                                                        m, verd), "verify_shares")
            wb_s = (time.perf_counter() - tw) / reps
            assert bytes(verd) == b"\x01" * m, "verify_share verdicts"
            # the block form: several batches in flight in ONE context (compute = enqueue only, absorb = wait + n verdict bytes)
            inflight, batches = 4, 12
            d_verd = [torch.zeros(m, dtype=torch.uint8, device=dev) for _ in range(inflight)]
            torch.cuda.synchronize()      # torch's zero fills must not land on top of verdicts the engine's streams write

            def wb_pipelined(count):
                issued = done = 0
                while done < count:
                    while issued < count and issued - done < inflight:
                        eng._check(lib.mpvss_modp_verify_shares_compute(ctx, capi.MPVSS_DEVICE, vp(d_pk), vp(d_S), vp(d_sh), vp(d_cb),
                                                                        vp(d_rb), m, vp(d_verd[issued % inflight])), "verify_shares_compute")
                        issued += 1
                    eng._check(lib.mpvss_modp_verify_shares_absorb(ctx, verd), "verify_shares_absorb")
                    assert bytes(verd) == b"\x01" * m, "verify_share verdicts (block API)"
                    done += 1

            wb_pipelined(inflight)                                # slot workspaces
            torch.cuda.synchronize()
            tw = time.perf_counter()
            wb_pipelined(batches)
            torch.cuda.synchronize()
            wbp_s = (time.perf_counter() - tw) / batches
            assert all(bool((dv == 1).all()) for dv in d_verd), "device verdict tensors"
            # `value`: n share boxes of one distribution box in ONE library call (the reference verifies them one verify_share call each,
            # participant.rs:361-386): the m distinct proofs tiled to n = 65536 device-resident rows, three calls, nothing of the caller's
            # in the loop -- what round 5's driver run showed to depend on the host when the batches were driven from Python
            tile = max(1, n // m)
            big = [torch.cat([x] * tile) for x in (d_pk[:m * EB], d_S, d_sh[:m * EB], d_cb, d_rb)]
            verd_big = (C.c_uint8 * (m * tile))()
            torch.cuda.synchronize()
            for it in range(4):
                if it == 1:
                    tw = time.perf_counter()
                eng._check(lib.mpvss_modp_verify_shares(ctx, capi.MPVSS_DEVICE, vp(big[0]), vp(big[1]), vp(big[2]), vp(big[3]), vp(big[4]),
                                                        m * tile, verd_big), "verify_shares (one call)")
            wb_one_s = (time.perf_counter() - tw) / 3
            assert bytes(verd_big) == b"\x01" * (m * tile), "verify_share verdicts (one call)"
            del big
            result["verify_share"] = {"value": m * tile / wb_one_s, "unit": "share-box verifications/s", "shares_per_call": m * tile,
                                      "distinct_proofs": m, "value_batches_in_flight": m / wbp_s, "batch": m,
                                      "batches_in_flight": inflight, "value_synchronous_calls": m / wb_s,
                                      "note": "W_B: a1 = G^r pk^c, a2 = S^r Y^c and the per-share SHA-256 verdict (K7) on the device; "
                                              "inputs resident in HBM; `value`: ONE mpvss_modp_verify_shares call over `shares_per_call` proofs; "
                                              f"`value_batches_in_flight`: mpvss_modp_verify_shares_compute/_absorb, {inflight} batches of `batch` in "
                                              "flight driven from Python; `value_synchronous_calls`: one call of `batch` proofs at a time"}
        if world == 1:
            # dealer side in block form: inputs resident in HBM, DEAL_DEPTH boxes in flight, X_i = g^P(i) through the comb,
            # host hashing of the oldest box beside the GPU work of the next ones; plus the scalar side of one box
            # (P(i), responses) behind the C ABI, timed separately
            d_pv, d_wt = dev_u8(pv_bytes), dev_u8(wit_bytes)
            deal_depth, deal_boxes = 8, 12

            def deal_absorb():
                st = (C.c_uint8 * capi.TRANSCRIPT_STATE_BYTES).from_buffer_copy(capi.transcript_init())
                eng._check(lib.mpvss_modp_distribute_absorb(ctx, st, None, None, None, None), "distribute_absorb")
                return capi.transcript_verdict(bytes(st), bytes(EB))[1]

            def deal_pipelined(count):
                issued = 0
                pend, digests = collections.deque(), []
                while issued < count or pend:
                    if issued < count and len(pend) < deal_depth:
                        eng._check(lib.mpvss_modp_distribute_compute(ctx, capi.MPVSS_DEVICE, None, 0, None, vp(d_pk), vp(d_pv), vp(d_wt), n,
                                                                     None, None, None, None), "distribute_compute")
                        issued += 1
                        pend.append(hash_pool.submit(deal_absorb))        # absorbs take the blocks in FIFO order
                    else:
                        digests.append(pend.popleft().result())
                return digests

            deal_pipelined(16)                 # every slot's workspace grows to the dealer's size here
            torch.cuda.synchronize()
            t_d = time.perf_counter()
            dg = deal_pipelined(deal_boxes)
            torch.cuda.synchronize()
            deal_blk_s = (time.perf_counter() - t_d) / deal_boxes
            if world == 1 and rank == 0:
                assert all(x == dealer_digest for x in dg), "dealer block API: transcript digest differs"
            t_sync = time.perf_counter()
            eng.distribute(commitments, positions, pubkeys, pv_bytes, wit_bytes)
            deal_s = time.perf_counter() - t_sync
            # the whole dealer in one call from host buffers (P(i), group work, digest, challenge, responses)
            coeff_bytes0 = b"".join(fx(a) for a in coeffs)
            # (the library call alone, over ctypes buffers made once: what a compiled caller pays per box; mean of 3 after a warm call)
            deal_call, deal_outputs = eng.deal_call(coeff_bytes0, positions, pubkeys, wit_bytes)
            deal_call()
            t_one = time.perf_counter()
            for _ in range(3):
                deal_call()
            deal_one_s = (time.perf_counter() - t_one) / 3
            one = deal_outputs()
            assert one["digest"] == dealer_digest and one["responses"] == responses and one["Y"] == shares, "mpvss_modp_deal differs"
            t_s = time.perf_counter()
            pv2 = capi.poly_eval(0, b"".join(fx(a) for a in coeffs), positions)
            rs2 = capi.dleq_responses(0, wit_bytes, pv2, challenge)
            scalar_s = time.perf_counter() - t_s
            assert pv2 == pv_bytes and rs2 == responses, "scalar side (C ABI) differs from the Python integers"
            # ... and the dealer END TO END: per box the scalar side before (P(i) for the box's own polynomial: forward
            # differences in Z/(q-1) on host threads, H2D of the values) and after the group work (challenge from the
            # transcript digest, responses r_i = w_i - P(i) c), pipelined over the boxes on host threads beside the GPU
            e2e_boxes = 16
            coeff_sets = [b"".join(fx(a) for a in boxes[b % len(boxes)].coeffs) for b in range(e2e_boxes)]
            scalar_pool = concurrent.futures.ThreadPoolExecutor(max_workers=3)
            absorb_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)      # blocks are absorbed in FIFO order

            def e2e_post(pv):
                digest = deal_absorb()
                cc = fx(int.from_bytes(hashlib.sha256(digest).digest(), "big") % ((Q - 1) // 2))
                return digest, scalar_pool.submit(capi.dleq_responses, 0, wit_bytes, pv, cc)

            def deal_e2e(count):
                pre = [scalar_pool.submit(capi.poly_eval, 0, coeff_sets[b], positions) for b in range(count)]
                post, keep = [], collections.deque()
                for b in range(count):
                    pv = pre[b].result()
                    d_pvb = dev_u8(pv)
                    keep.append(d_pvb)
                    while len(post) - sum(f.done() for f in post) >= deal_depth:      # at most deal_depth boxes in flight
                        time.sleep(0.0005)
                    eng._check(lib.mpvss_modp_distribute_compute(ctx, capi.MPVSS_DEVICE, None, 0, None, vp(d_pk), vp(d_pvb), vp(d_wt), n,
                                                                 None, None, None, None), "distribute_compute")
                    post.append(absorb_pool.submit(e2e_post, pv))
                outs = [f.result() for f in post]
                return [(dgst, fut.result()) for dgst, fut in outs]

            deal_e2e(3)
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            e2e = deal_e2e(e2e_boxes)
            torch.cuda.synchronize()
            e2e_host_s = (time.perf_counter() - t_e) / e2e_boxes
            assert e2e[0][0] == dealer_digest and e2e[0][1] == responses, "end-to-end dealer: box 0 differs"
            scalar_pool.shutdown()

            # ... and with the scalar side on the device as well: P(i) mod (q-1) by mpvss_modp_poly_eval_device into HBM, the
            # group work on those values, the challenge from the transcript digest, r_i by mpvss_modp_dleq_responses_device --
            # nothing of a box but its t coefficients, its digest and its challenge crosses the bus
            ring = [torch.empty(n * EB, dtype=torch.uint8, device=dev) for _ in range(e2e_boxes)]     # P(i) of a box lives until its responses are out

            resp_pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
            d_rs_e2e = [torch.empty(n * EB, dtype=torch.uint8, device=dev) for _ in range(e2e_boxes)]

            def e2e_resp(b, d_pvb, cc):
                d_r = d_rs_e2e[b % len(d_rs_e2e)]
                eng.dleq_responses_device(d_wt.data_ptr(), d_pvb.data_ptr(), cc, n, d_r.data_ptr())
                return bytes(d_r.cpu().numpy().tobytes()) if b == 0 else None

            def e2e_post_dev(b, d_pvb, ticket):
                st = eng.verify_block_absorb_claimed(ticket, capi.transcript_init())       # dealer blocks are absorbed like verifier blocks
                digest = capi.transcript_verdict(st, bytes(EB))[1]
                cc = fx(int.from_bytes(hashlib.sha256(digest).digest(), "big") % ((Q - 1) // 2))
                return digest, resp_pool.submit(e2e_resp, b, d_pvb, cc)      # the responses wait for wave slots, not the next absorb

            def deal_e2e_dev(count):
                post = []
                for b in range(count):
                    d_pvb = ring[b % len(ring)]
                    while len(post) - sum(f.done() for f in post) >= deal_depth:      # at most deal_depth boxes in flight
                        time.sleep(0.0005)
                    eng.deal_compute(coeff_sets[b], cur.d_pos.data_ptr(), d_pk.data_ptr(), d_wt.data_ptr(), n, d_pvb.data_ptr())
                    post.append(hash_pool.submit(e2e_post_dev, b, d_pvb, eng.block_claim()))     # several boxes are hashed at a time
                outs = [f.result() for f in post]
                return [(dgst, fut.result()) for dgst, fut in outs]

            deal_e2e_dev(3)
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            e2e = deal_e2e_dev(e2e_boxes)
            torch.cuda.synchronize()
            e2e_s = (time.perf_counter() - t_e) / e2e_boxes
            assert e2e[0][0] == dealer_digest and e2e[0][1] == responses, "end-to-end dealer (device scalars): box 0 differs"
            absorb_pool.shutdown()
            resp_pool.shutdown()
            t_s = time.perf_counter()
            eng.poly_eval_device(coeff_sets[0], cur.d_pos.data_ptr(), n, ring[0].data_ptr())
            eng.dleq_responses_device(d_wt.data_ptr(), ring[0].data_ptr(), challenge, n, d_rs_e2e[0].data_ptr())
            scalar_dev_s = time.perf_counter() - t_s
            assert bytes(d_rs_e2e[0].cpu().numpy().tobytes()) == responses, "device scalar side differs from the Python integers"
            result["distribute"] = {"value": n / deal_blk_s, "unit": "shares dealt/s", "ms_per_box": deal_blk_s * 1e3,
                                    "boxes_in_flight": deal_depth, "value_synchronous_host_buffers": n / deal_s,
                                    "value_one_call_host_buffers_end_to_end": n / deal_one_s,
                                    "scalar_side_ms_per_box": scalar_s * 1e3, "scalar_side_on_device_ms_per_box": scalar_dev_s * 1e3,
                                    "value_end_to_end": n / e2e_s, "end_to_end_ms_per_box": e2e_s * 1e3,
                                    "value_end_to_end_host_scalars": n / e2e_host_s,
                                    "note": "dealer side of distribute_secret (participant.rs:160-286): X_i = g^P(i), Y_i = y_i^P(i), "
                                            "a1 = g^w_i, a2 = y_i^w_i and the ordered transcript hash; `value`: "
                                            "mpvss_modp_distribute_compute/_absorb, inputs resident in HBM, several boxes in flight; "
                                            "`value_synchronous_host_buffers`: one mpvss_modp_distribute call (X from the commitments, "
                                            "PCIe included); `value_one_call_host_buffers_end_to_end`: one mpvss_modp_deal call (P(i), group "
                                            "work, digest, challenge, responses; host buffers; the library call alone over ctypes buffers made once, mean of 3); scalar_side: P(i) and the responses for one box through "
                                            "mpvss_modp_poly_eval / mpvss_modp_dleq_responses (host threads), not in `value`; `value_end_to_end`: "
                                            "every box with its own polynomial -- P(i) mod (q-1) and the responses on the device "
                                            "(mpvss_modp_poly_eval_device / _dleq_responses_device: residues mod (q-1)/2 in the Montgomery "
                                            "kernels, parity beside), the group work, the transcript hash and the challenge, boxes pipelined; "
                                            "`value_end_to_end_host_scalars`: the same with P(i) (forward differences in Z/(q-1)) and the "
                                            "responses on host threads and the values uploaded"}

        # ---------------- CPU baselines (rank 0, N == 1 only), SURVEY 8(d) ----------------
        # (i) the C port of the reference's operation sequence on ONE thread -- how src/participant.rs:408 runs (no rayon
        # on this path); (ii) the same on every free core (`cpu_baseline.value`); (iii) the same sequence with OpenSSL's
        # BN_mod_exp_mont where libcrypto is present ("strong CPU" line).  About 15 s of wall time altogether; the GPU's
        # X / a1 / a2 of every sampled share must equal the CPU's.
        if rank == 0 and world == 1 and args.cpu_sample != 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from concurrent.futures import ThreadPoolExecutor

            from modp_ref import ModpRef
            ref = ModpRef()
            # Threads that really run in parallel here (cpu_count() ignores cgroup quotas): calibrate with
            # short full-width modpows, 1 thread vs many.
            nthreads = min(len(os.sched_getaffinity(0)) or 1, 64)
            bb, ee = int.from_bytes(commitments[:EB], "big"), ORDER - 12345
            t1 = time.perf_counter(); ref.modpow(bb, ee); single = time.perf_counter() - t1
            with ThreadPoolExecutor(max_workers=nthreads) as ex:
                t1 = time.perf_counter()
                list(ex.map(lambda _: ref.modpow(bb, ee), range(2 * nthreads)))
                par = time.perf_counter() - t1
            cores = max(1, min(nthreads, int(round(2 * nthreads * single / par))))
            cpu_model = "unknown"
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.lower().startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            # GPU outputs for the sampled shares (one more, untimed, dumped verification)
            X = (C.c_uint8 * (n * EB))(); A1 = (C.c_uint8 * (n * EB))(); A2 = (C.c_uint8 * (n * EB))()
            v = C.c_int(0); dg = (C.c_uint8 * 32)()
            eng._check(lib.mpvss_modp_verify_distribution(ctx, capi.MPVSS_DEVICE, vp(d_cm), t, vp(d_pos), vp(d_pk),
                                                          vp(d_sh), vp(d_rs), n, C.cast(ch_buf, C.c_void_p), C.byref(v),
                                                          dg, X, A1, A2), "verify_distribution(dump)")
            Xb, A1b, A2b = bytes(X), bytes(A1), bytes(A2)

            def spread(k):
                k = max(1, min(k, n))
                return sorted({int((j + 0.5) * n / k) for j in range(k)})

            def timed(work, idx, threads):
                tc = time.perf_counter()
                if threads == 1:
                    outs = [work(i) for i in idx]
                else:
                    with ThreadPoolExecutor(max_workers=threads) as ex:
                        outs = list(ex.map(work, idx))
                secs = time.perf_counter() - tc
                for i, (x, a1, a2) in zip(idx, outs):
                    s = slice(i * EB, (i + 1) * EB)
                    assert (x, a1, a2) == (Xb[s], A1b[s], A2b[s]), f"GPU/CPU mismatch at share {i}"
                return secs

            def work(i):
                s = slice(i * EB, (i + 1) * EB)
                return ref.share_work(commitments, positions[i], pubkeys[s], shares[s], responses[s], challenge)
            # one share costs ~(sum of exponent bits) * 1.5 products of the port: ~6 s of wall time per leg
            est_share_s = single * (sum(min(17 * j, 2048) for j in range(t)) + 2 * 2048 + 2 * 256) / 2048.0
            k1 = max(1, min(4, int(6.0 / max(est_share_s, 1e-3))))
            idx1 = spread(k1)
            s1 = timed(work, idx1, 1)
            k = args.cpu_sample if args.cpu_sample > 0 else max(cores, min(8 * cores, int(6.0 * cores / max(est_share_s, 1e-3))))
            idx = spread(k)
            cpu_s = timed(work, idx, cores)
            result["cpu_baseline"] = {
                "value": len(idx) / cpu_s, "unit": "share verifications/s", "cores": cores, "kind": "port",
                "cpu_model": cpu_model, "hardware_threads": os.cpu_count(),
                "sample": f"{len(idx)} of {n} shares, oracle/modp_ref.c on {cores} threads, {cpu_s:.1f} s",
                "sample_detail": f"{len(idx)} of {n} shares (positions spread over [1,{n}], all t={t} commitments), "
                                 f"reference operation sequence (t+4 modpow, t+2 mul per share) in oracle/modp_ref.c on {cores} "
                                 f"threads, {cpu_s:.1f}s; GPU X/a1/a2 of those shares checked equal",
                "single_thread": {"value": len(idx1) / s1, "unit": "share verifications/s", "cores": 1, "kind": "port",
                                  "sample": f"{len(idx1)} share(s), same C port on one thread ({s1:.1f}s) -- the reference runs "
                                            "this path on one thread (src/participant.rs:408-448 has no rayon)"},
            }
            try:
                import openssl_ref
                if openssl_ref.available():
                    tls = threading.local()

                    def work_ssl(i):
                        if not hasattr(tls, "ref"):
                            tls.ref = openssl_ref.OpenSslRef()
                        s = slice(i * EB, (i + 1) * EB)
                        return tls.ref.share_work(commitments, positions[i], pubkeys[s], shares[s], responses[s], challenge)
                    t1 = time.perf_counter(); work_ssl(spread(1)[0]); one = time.perf_counter() - t1
                    idx_s = spread(max(cores, min(16 * cores, int(3.0 * cores / max(one, 1e-3)))))
                    ssl_s = timed(work_ssl, idx_s, cores)
                    result["cpu_baseline"]["openssl"] = {
                        "value": len(idx_s) / ssl_s, "unit": "share verifications/s", "cores": cores, "kind": "port",
                        "library": openssl_ref.version(),
                        "sample": f"{len(idx_s)} shares, the same operation sequence with libcrypto's BN_mod_exp_mont / BN_mod_mul "
                                  f"(oracle/openssl_ref.py) on {cores} threads, {ssl_s:.1f}s; one share on one thread {one * 1e3:.0f} ms; "
                                  "GPU X/a1/a2 of those shares checked equal"}
            except Exception as exc:      # noqa: BLE001 - the strong-CPU line is optional
                result["cpu_baseline"]["openssl"] = {"value": None, "note": f"skipped: {exc}"}
        # ---------------- the other MODP shapes of BASELINE.json: C2 and one GPU's slice of C5 ----------------
        def bench_shape(n_, t_, k, depth, lo_, what, one_box_of=0):
            """k boxes of another shape through the same step functions (own participants, up to 4 distinct boxes, the slots
            grow to the shape in an untimed pass); returns value / ms_per_box / compute of that shape.
            one_box_of = N > 0: the participants are positions lo_+1 .. lo_+n_ of ONE seeded list of N (the same on every rank and for
            every world size: the box -- keys, shares, transcript digest -- does not depend on how many GPUs it is split over)."""
            saved = (cur.boxes, cur.d_pk, cur.d_pos, cur.n)
            if one_box_of:
                privs_all, wits_all = make_participants(one_box_of, SEED * 1000003 + 977 * t_ + 31337)
                privs_, wits_ = privs_all[lo_:lo_ + n_], wits_all[lo_:lo_ + n_]
            else:
                privs_, wits_ = make_participants(n_, SEED * 1000003 + 977 * t_ + rank)
            pos_ = list(range(lo_ + 1, lo_ + n_ + 1))
            pub_ = eng.batch_exp_fixed_base(fx(2), b"".join(fx(x) for x in privs_))
            witb_ = b"".join(map(fx, wits_))
            try:
                cur.d_pk, cur.d_pos, cur.n = dev_u8(pub_), torch.tensor(pos_, dtype=torch.int64, device=dev), n_
                cur.boxes = [make_box(b, n_, t_, pos_, pub_, witb_, False, tag_base=100000 + 1000 * t_) for b in range(min(k, 4))]
                if world > 1:
                    rccl_reap(0)
                    rccl_buffers(n_)
                init = min(depth + max(HASH_THREADS, 1) + 2, capi.BLOCK_SLOTS - 1)
                if world == 1 and n_ <= 16384:
                    # small boxes travel in groups (one block for a run of boxes, mpvss_modp_verify_many): the same run, untimed,
                    # grows exactly the slots the timed one uses
                    gate(run_steps(k, depth=depth), f"{what}: slot initialisation")
                else:
                    gate(run_steps(init, depth=init), f"{what}: slot initialisation")
                eng.pipeline_stats(reset=True)
                barrier()
                t_s = time.perf_counter()
                res_s = run_steps(k, depth=depth)
                barrier()
                dt = time.perf_counter() - t_s
                gate(res_s, what)
                if world > 1:
                    tt_ = torch.tensor([dt], dtype=torch.float64, device=commdev if smoke_one_gpu else dev)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    dt = float(tt_.item())
                wk_ = modp_work(n_, t_, pos_, [bx.c for _, _, bx in res_s])
                rate = wk_["slots"] * k / dt
                st_ = eng.pipeline_stats()
                return {"value": n_ * world * k / dt, "unit": "share verifications/s", "ms_per_box": dt / k * 1e3, "boxes": k,
                        "digest_of_digests": hashlib.sha256(b"".join(d for _, d, _ in res_s)).hexdigest(),
                        "host_ms_per_box": {"enqueue": st_["enqueue_ms"] / max(st_["blocks"], 1), "wait": st_["wait_ms"] / max(st_["blocks"], 1),
                                            "hash": st_["hash_ms"] / max(st_["blocks"], 1)},
                        "boxes_in_flight": depth, "distinct_boxes": len(cur.boxes),
                        "config": {"workload": f"ModpGroup 2048-bit verify_distribution_shares n={n_} t={t_} per GPU ({what}), "
                                               f"honest-dealer boxes, inputs resident in HBM"},
                        "compute": {"bound": "valu issue", "achieved": rate, "peak": PEAK_VALU_SLOTS_PER_S, "peak_is": PEAK_SOURCE,
                                    "frac": rate / PEAK_VALU_SLOTS_PER_S,
                                    "unit": "VALU wave-instruction issue slots/s", "valu_slots_per_share": wk_["slots"] / n_,
                                    "modmul_per_share": wk_["mm_total"] / n_, "x_path": wk_["x_path"]}}
            finally:
                cur.boxes, cur.d_pk, cur.d_pos, cur.n = saved
                if world > 1:
                    rccl_reap(0)
                    rccl_buffers(n)

        if args.config_boxes != 0 and keyset[0] is None:
            if world == 1:
                k2, k5 = (192, 20) if args.config_boxes < 0 else (args.config_boxes, max(2, args.config_boxes // 8))
                only = os.environ.get("MPVSS_BENCH_CONFIGS", "c2,c5_slice").split(",")
                result["configs"] = {}
                if "c2" in only:
                    result["configs"]["c2"] = bench_shape(4096, 64, k2, int(os.environ.get("MPVSS_BENCH_C2_DEPTH", "14")), 0,
                                                          "BASELINE config C2")
                if "c5_slice" in only:
                    result["configs"]["c5_slice"] = bench_shape(131072, 1024, k5, int(os.environ.get("MPVSS_BENCH_C5_DEPTH", "10")), 0,
                                                                "one GPU's slice of BASELINE config C5: positions 1..131072 of 1048576")
            elif world == 8 or os.environ.get("MPVSS_BENCH_C5") == "1":
                # BASELINE config C5 itself: ONE box of world x 131072 participants, t = 1024, every rank its block
                k5 = 10 if args.config_boxes < 0 else max(2, args.config_boxes)
                n5 = int(os.environ.get("MPVSS_BENCH_C5_N", "131072"))
                result["c5"] = bench_shape(n5, int(os.environ.get("MPVSS_BENCH_C5_T", "1024")), k5, 12, rank * n5,
                                           f"BASELINE config C5: {n5 * world} participants over {world} GPUs")
        # ---------------- N > 1: the metric's FIXED box (n = --participants) split over the N GPUs, beside the weak figure ----------------
        # Rank g verifies positions [g n/N, (g+1) n/N) of every box (8192 shares per GPU at N = 8) through the same chained pipeline; the
        # transcript's running state still travels rank to rank, but a block's hash is 35/N ms now, so the chain fill is (N-1) x 35/N ms
        # instead of (N-1) x 35 ms.  Blocks are small: more of them in flight per rank.
        if ((world > 1 and args.scaling == "both") or os.environ.get("MPVSS_BENCH_STRONG_AT_N1") == "1") and args.n % world == 0 \
                and keyset[0] is None:
            depth_s = min(RANK_DEPTH * min(world, 4), capi.BLOCK_SLOTS - 16)
            result["strong"] = bench_shape(args.n // world, t, args.steps, depth_s, rank * (args.n // world),
                                           f"strong scaling: ONE box of {args.n} participants split over {world} GPUs", one_box_of=args.n)
            result["strong"]["scaling"] = "strong"
            result["strong"]["n_per_gpu"] = args.n // world
        # ---------------- the same boxes handed over in HOST memory (PCIe included); never `value` ----------------
        if world == 1 and args.host_boxes > 0:
            pos_arr = (C.c_int64 * n)(*positions)
            hb = [(C.c_uint8 * len(b)).from_buffer_copy(b) for b in (commitments, pubkeys, shares, responses)]
            hbox = capi.ModpBox(C.addressof(hb[0]), t, C.addressof(pos_arr), C.addressof(hb[1]), C.addressof(hb[2]),
                                C.addressof(hb[3]), n, C.cast(ch_buf, C.c_void_p), None, 0)

            def host_many(count):
                arr = (capi.ModpBox * count)(*([hbox] * count))
                vd = (C.c_int * count)()
                dg = (C.c_uint8 * (32 * count))()
                eng._check(lib.mpvss_modp_verify_many(ctx, capi.MPVSS_HOST, arr, count, min(PIPE_DEPTH, capi.BLOCK_SLOTS), max(HASH_THREADS, 1), vd,
                                                      C.cast(dg, C.c_void_p)), "verify_many(host)")
                return all(vd[i] == 1 for i in range(count)) and all(bytes(dg)[32 * i:32 * i + 32] == dealer_digest for i in range(count))

            assert host_many(16), "host-buffer boxes: verdict or digest wrong"      # every slot grows its pinned staging here
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            ok_h = host_many(args.host_boxes)
            host_s = (time.perf_counter() - t_h) / args.host_boxes
            assert ok_h, "host-buffer boxes: verdict or digest wrong"
            result["host_buffers"] = {"value": n / host_s, "unit": "share verifications/s", "ms_per_box": host_s * 1e3,
                                      "boxes": args.host_boxes,
                                      "note": "same boxes, every input in pageable host memory: the library copies them into pinned "
                                              "staging and over PCIe (3 x n x 256 B per box) inside the timed calls; not `value`"}
        # ---------------- ONE call at the reference's own sizes: latency, not throughput; never `value` ----------------
        # BASELINE config C1 (n = 3, t = 3: examples/mpvss_all.rs) and config C2 (n = 4096, t = 64) as ONE call each from host buffers, on
        # an idle chip, best of 5: mpvss_modp_deal (participant.rs:160-286), mpvss_modp_verify_distribution (:399-455),
        # mpvss_modp_extract_shares (:294-353), mpvss_modp_verify_shares (:361-386); the dealer's box must verify.
        if world == 1 and args.config_boxes != 0 and keyset[0] is None:
            def best_ms(f, reps=5):
                f()
                out = []
                for _ in range(reps):
                    t_b = time.perf_counter()
                    f()
                    out.append((time.perf_counter() - t_b) * 1e3)
                return min(out)

            one_call = {}
            rng_o = random.Random(20261004)
            sc_o = lambda k: b"".join(fx(rng_o.randrange(1, 1 << 2040)) for _ in range(k))
            for name_o, n_o, t_o in (("c1", 3, 3), ("c2", 4096, 64)):
                pos_o = list(range(1, n_o + 1))
                co_o, wi_o = sc_o(t_o), sc_o(n_o)
                xs_o = [rng_o.randrange(3, Q - 1) | 1 for _ in range(n_o)]                      # odd and below q - 1: invertible mod q - 1
                pk_o = eng.batch_exp_fixed_base(fx(2), b"".join(map(fx, xs_o)))
                cm_o = eng.batch_exp_fixed_base(fx(4), co_o)
                bx_o = eng.deal(co_o, pos_o, pk_o, wi_o)
                res_o = eng.verify_distribution(cm_o, pos_o, pk_o, bx_o["Y"], bx_o["responses"], bx_o["challenge"])
                assert res_o["verdict"] and res_o["digest"] == bx_o["digest"], f"one_call {name_o}: the dealer's box does not verify"
                xi_o = b"".join(fx(pow(x, -1, Q - 1)) for x in xs_o)
                s_o, c_o = eng.extract_shares(pk_o, bx_o["Y"], xi_o, wi_o)
                r_o = capi.dleq_responses(0, wi_o, b"".join(map(fx, xs_o)), c_o)
                assert all(eng.verify_shares(pk_o, s_o, bx_o["Y"], c_o, r_o)), f"one_call {name_o}: a share box does not verify"
                one_call[name_o] = {
                    "n": n_o, "t": t_o,
                    "deal_ms": best_ms(lambda: eng.deal(co_o, pos_o, pk_o, wi_o)),
                    "verify_distribution_ms": best_ms(lambda: eng.verify_distribution(cm_o, pos_o, pk_o, bx_o["Y"], bx_o["responses"], bx_o["challenge"])),
                    "extract_shares_ms": best_ms(lambda: eng.extract_shares(pk_o, bx_o["Y"], xi_o, wi_o)),
                    "verify_shares_ms": best_ms(lambda: eng.verify_shares(pk_o, s_o, bx_o["Y"], c_o, r_o))}
            one_call["note"] = ("ONE call from host buffers on an idle chip, best of 5 (Python marshalling included): the sizes the reference's own "
                                "example and BASELINE config C2 use; small batches take the row-layout kernels (16 lanes per number) -- a call is the "
                                "latency of one 2048-bit exponentiation chain; not `value`")
            result["one_call"] = one_call
        # ---------------- the reference's call shape: ONE box per call, T concurrent callers on one context; never `value` ----------------
        # What `rust/src/batch.rs::verify_distribution_shares` / `Participant::verify_distribution_shares` bind (participant.rs:399-455):
        # mpvss_modp_verify_distribution, one box, blocking.  T host threads each take their share of the SAME K distinct boxes of the
        # headline run (the crate goes parallel the same way, participant.rs:490-500); ctypes releases the GIL for the duration of a
        # foreign call, so these are compiled callers as far as the library is concerned.  Every call's verdict and digest is gated.
        dthreads = [int(x) for x in args.drop_in_threads.split(",") if x.strip() and int(x) > 0]
        if world == 1 and dthreads and keyset[0] is None:
            def one_call(bx, space, hbufs):
                verdict, dg = C.c_int(0), (C.c_uint8 * 32)()
                if space == capi.MPVSS_DEVICE:
                    rcode = lib.mpvss_modp_verify_distribution(ctx, space, vp(bx.d_cm), bx.t, vp(d_pos), vp(d_pk), vp(bx.d_sh), vp(bx.d_rs), bx.n,
                                                               C.cast(bx.ch_buf, C.c_void_p), C.byref(verdict), dg, None, None, None)
                else:
                    rcode = lib.mpvss_modp_verify_distribution(ctx, space, hbufs[0], bx.t, C.cast(pos_arr_d, C.c_void_p), pk_host, hbufs[1], hbufs[2],
                                                               bx.n, C.cast(bx.ch_buf, C.c_void_p), C.byref(verdict), dg, None, None, None)
                eng._check(rcode, "verify_distribution (drop_in)")
                if verdict.value != 1 or bytes(dg) != bx.dealer_digest:
                    raise AssertionError("drop_in: verdict or digest wrong")

            def drop_in_run(T, space, seq, hb):
                errs = []

                def worker(k):
                    try:
                        for i in range(k, len(seq), T):
                            one_call(seq[i], space, hb[i % len(hb)] if hb else None)
                    except BaseException as exc:      # noqa: BLE001
                        errs.append(exc)
                ths = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
                torch.cuda.synchronize()
                t_d = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                el = time.perf_counter() - t_d
                if errs:
                    raise errs[0]
                return el

            seq_d = [boxes[s % len(boxes)] for s in range(max(2 * args.steps, 40))]      # (the K distinct boxes twice: a run of T callers ends ragged)
            pos_arr_d = (C.c_int64 * n)(*positions)
            pk_host = (C.c_uint8 * len(pubkeys)).from_buffer_copy(pubkeys)
            drop_in_run(max(dthreads) + 2, capi.MPVSS_DEVICE, seq_d[:max(dthreads) + 2], None)      # slots of the most callers, untimed
            by_t = {}
            for T in dthreads:
                by_t[T] = min(drop_in_run(T, capi.MPVSS_DEVICE, seq_d, None) for _ in range(2)) / len(seq_d)
            t_best = min(by_t, key=by_t.get)
            # the other legs (host buffers, key cache, dealers) at a thread count that does not hang on which T won the device-buffer
            # comparison above by a per cent: shorter boxes need more callers (a keyed box is 37 ms on the GPU and 35 ms of hashing per caller)
            t_many = max(t_best, min(16, max(dthreads)))
            lone_s = drop_in_run(1, capi.MPVSS_DEVICE, seq_d[:4], None) / 4
            # host buffers (what a Rust caller's Vec<u8>s are), at the best thread count: PCIe and the pinned copies inside the calls
            hb_d = [[(C.c_uint8 * len(b)).from_buffer_copy(b) for b in (bx.commitments, bx.shares, bx.responses)] for bx in boxes]
            drop_in_run(t_many + 2, capi.MPVSS_HOST, seq_d[:t_many + 2], hb_d)                     # pinned staging of those slots, untimed
            host_s = min(drop_in_run(t_many, capi.MPVSS_HOST, seq_d, hb_d) for _ in range(2)) / len(seq_d)
            lone_host_s = drop_in_run(1, capi.MPVSS_HOST, seq_d[:4], hb_d) / 4
            # the dealers' side of the same shape: T threads each calling the one-call mpvss_modp_deal (participant.rs:160-286: P(i), group
            # work, transcript, challenge, responses) from host buffers; every dealer its own polynomial; the library hands each call a
            # set of input / secret buffers from its pool
            deal_calls = []
            for k in range(t_many):
                bx_k = boxes[k % len(boxes)]
                call_k, outs_k = eng.deal_call(b"".join(fx(a) for a in bx_k.coeffs), positions, pubkeys, bx_k.wit_bytes)
                deal_calls.append((call_k, outs_k, bx_k))
            deal_errs = []

            def deal_worker(k, reps):
                try:
                    for _ in range(reps):
                        deal_calls[k][0]()
                except BaseException as exc:      # noqa: BLE001
                    deal_errs.append(exc)

            def deal_run(reps):
                ths = [threading.Thread(target=deal_worker, args=(k, reps)) for k in range(t_many)]
                t_dd = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                if deal_errs:
                    raise deal_errs[0]
                return time.perf_counter() - t_dd
            def deal_check():
                for call_k, outs_k, bx_k in deal_calls:          # every caller gets ITS box: the digest of the block-form dealer's
                    got_k = outs_k()
                    assert got_k["digest"] == bx_k.dealer_digest and got_k["Y"] == bx_k.shares and got_k["challenge"] == bx_k.challenge, \
                        "drop_in deal: a concurrent mpvss_modp_deal call returned another box"
            deal_run(1)
            deal_dd_s = deal_run(2) / (2 * t_many)
            deal_check()
            # ... and with the context's cross-call key cache on (mpvss_ctx_set_key_cache_lru; ONE session for the verifiers and the
            # dealers: the 19 GB of tables are allocated once): the same one-box calls, the library recognises the participants' key
            # array by its SHA-256 (hashed inside every call) and verifies against per-key tables it built at the second box -- the
            # tables are there when the timed calls start; opt-in, never `value`.  The dealers to the same keys then take
            # Y_i = y_i^P(i) and a2_i = y_i^w_i from those tables (k_modp_keyset_twin_exp_pair) instead of the bucket kernels -- the
            # same boxes, byte for byte
            kc_s = kc_lone_s = None
            deal_kc_s = deal_kc_lone_s = None
            try:
                eng.set_key_cache_lru(1, 2)
                drop_in_run(2, capi.MPVSS_HOST, seq_d[:4], hb_d)                                # second sighting: tables built here
                drop_in_run(t_many + 2, capi.MPVSS_HOST, seq_d[:t_many + 2], hb_d)
                kc_s = min(drop_in_run(t_many, capi.MPVSS_HOST, seq_d, hb_d) for _ in range(2)) / len(seq_d)
                kc_lone_s = drop_in_run(1, capi.MPVSS_HOST, seq_d[:4], hb_d) / 4
                deal_run(1)                                      # every dealing caller's slot warm
                deal_kc_s = deal_run(2) / (2 * t_many)
                deal_check()
                t_dd = time.perf_counter()
                for _ in range(3):
                    deal_calls[0][0]()
                deal_kc_lone_s = (time.perf_counter() - t_dd) / 3
            finally:
                eng.set_key_cache_lru(0)
            del deal_calls, hb_d
            result["drop_in"] = {"value": n / by_t[t_best], "threads": t_best, "threads_other_legs": t_many, "value_lone": n / lone_s,
                                 "value_host_buffers": n / host_s, "value_lone_host_buffers": n / lone_host_s,
                                 "value_key_cache": n / kc_s if kc_s else None, "value_lone_key_cache": n / kc_lone_s if kc_lone_s else None,
                                 "deal_value": n / deal_dd_s, "deal_value_key_cache": n / deal_kc_s if deal_kc_s else None,
                                 "deal_value_lone_key_cache": n / deal_kc_lone_s if deal_kc_lone_s else None,
                                 "by_threads": {str(T): n / v for T, v in by_t.items()}, "boxes": len(seq_d), "unit": "share verifications/s",
                                 "vs_verify_many": (n / by_t[t_best]) / value,
                                 "note": "T host threads, each calling the ONE-box mpvss_modp_verify_distribution (participant.rs:399-455; what "
                                         "rust/src/participant.rs binds) on ONE context over the K distinct boxes of the headline run, best of two "
                                         "passes per T; `value_lone`: one caller, one box at a time (its box's latency); `value_key_cache` / `value_lone_key_cache`: host "
                                         "buffers with the cross-call key cache on (tables of the participants' keys built once, 19.3 GB; the key array "
                                         "is hashed inside every call); `deal_value`: T threads x the one-call mpvss_modp_deal (shares dealt/s), "
                                         "`deal_value_key_cache` / `deal_value_lone_key_cache`: the same with the dealer's y^P(i), y^w from the key tables; not `value`"}
        # ---------------- opt-in variant: registered public keys (include/mpvss_hip.h) ----------------
        # NOT the headline: `value` above recomputes y_i^r_i from the bare keys in every step.  Here the per-key tables
        # are built once (timed separately) and the same K steps are repeated against them -- the situation of a verifier
        # that checks many dealers' boxes against one set of long-lived participant keys.
        keyset_ok = False
        if world == 1 and args.registered_keys:
            torch.cuda.synchronize()
            tk = time.perf_counter()
            h = C.c_void_p()
            try:
                eng._check(lib.mpvss_modp_keyset_create(ctx, capi.MPVSS_DEVICE, vp(d_pk), n, C.byref(h)), "keyset_create")
                keyset_ok = True
            except capi.EngineError as err:      # 295 KB per key: very large key sets do not fit beside the workspaces
                result["registered_keys"] = {"value": None, "note": f"skipped: {err}"}
        if keyset_ok:
            try:
                torch.cuda.synchronize()
                build_s = time.perf_counter() - tk
                keyset[0] = h
                # a box is ~37 ms on the GPU now and ~35 ms of hashing: more boxes in flight and more hash threads than the headline's
                # (measured, profiles/r05_keyset_ab.txt: 10 / 8 -> 1.64 M, 16 / 12 -> 1.78 M, 24 -> 1.3-1.4 M)
                ks_depth = int(os.environ.get("MPVSS_BENCH_KEYSET_DEPTH", "16"))
                hash_threads_now[0] = int(os.environ.get("MPVSS_BENCH_KEYSET_HASH_THREADS", "12"))
                gate(run_steps(ks_depth + hash_threads_now[0] + 2, depth=ks_depth), "registered keys, slot initialisation and warm-up")
                barrier()
                t1 = time.perf_counter()
                res_k = run_steps(args.steps, depth=ks_depth)
                barrier()
                el_k = time.perf_counter() - t1
                gate(res_k, "registered keys")
                ks_steady = None
                if args.steady_steps > 0:
                    barrier()
                    t1 = time.perf_counter()
                    res_s = run_steps(args.steady_steps, depth=ks_depth)
                    barrier()
                    ks_steady = n * args.steady_steps / (time.perf_counter() - t1)
                    gate(res_s, "registered keys, steady state")
                keyset[0] = None
                table_bytes = int(lib.mpvss_modp_keyset_bytes(h))
                lib.mpvss_modp_keyset_destroy(ctx, h)
                # ... and behind the UNCHANGED call: with the context's key cache on, mpvss_modp_verify_many sees that the K boxes of the call
                # present the same key array, builds the tables itself -- INSIDE the timed call -- and frees them when it returns
                eng.set_key_cache(3)
                try:
                    gate(run_steps(4, depth=ks_depth), "key cache, warm-up")
                    barrier()
                    t1 = time.perf_counter()
                    res_t = run_steps(args.steps, depth=ks_depth)
                    barrier()
                    el_t = time.perf_counter() - t1
                    gate(res_t, "key cache behind verify_many")
                finally:
                    eng.set_key_cache(0)
                hash_threads_now[0] = HASH_THREADS
                # instead of the y tables and the 2046-squaring chain: 252 squarings, 37 x 8 table products (7-bit windows of the eight
                # 256-bit rows of r), 64 nibbles of c against Y's full table (15 products instead of the odd-power table's 9), the closing one
                mm_k = mm_total - n * a2_products - n * (63 if w6 else 15) + n * (252 * SQ_COST + 296 + 64 + 1 + (15 - 9))
                result["registered_keys"] = {
                    "value": n * args.steps / el_k, "value_steady_state": ks_steady, "boxes_in_flight": ks_depth,
                    "value_transparent": n * args.steps / el_t,
                    "value_transparent_is": "the same K steps through the plain mpvss_modp_verify_many call with mpvss_ctx_set_key_cache(ctx, 3): "
                                            "the library registers the key array the boxes share by itself and builds its tables inside the timed call "
                                            "(in the buffer the context kept from the warm-up call's tables)",
                    "unit": "share verifications/s", "ms_per_step": el_k / args.steps * 1e3,
                    "table_bytes": table_bytes, "table_build_s": build_s, "modmul_per_share": mm_k / n,
                    "modmul_equivalents_per_s": mm_k / (el_k / args.steps),
                    "note": "opt-in mpvss_modp_keyset_*: per-key tables y^(d 2^(256 j)), d < 128, in HBM (295 KB per key), built once per key "
                            "set; a2 = y^r Y^c in 613 products instead of 2620, on the pair layout with g^r and a1; same verdict and "
                            "transcript digest; not the headline"}
            except capi.EngineError as err:
                keyset[0] = None
                hash_threads_now[0] = HASH_THREADS
                result["registered_keys"] = {"value": None, "note": f"skipped: {err}"}

        # ---------------- C3 / C4: the curve groups at n=65536, t=256 (rank 0, N == 1) ----------------
        if rank == 0 and world == 1 and args.ec_boxes > 0:
            result["ec"] = {name: bench_ec(eng, name, args) for name in ("secp256k1", "ristretto255")}

    except Exception as exc:      # noqa: BLE001 - reported in the line and through the exit code
        import traceback
        secondary_error = "".join(traceback.format_exception_only(type(exc), exc)).strip()
        result["secondary_error"] = secondary_error
        traceback.print_exc()
    fd_blocks, fd_fallbacks = eng.fd_stats()
    result["compute"]["fd_blocks"] = fd_blocks
    result["compute"]["fd_fallbacks"] = fd_fallbacks          # boxes whose pipeline gave up and were recomputed by Horner
    if rank == 0:
        # The driver parses the LAST stdout line: a compact object (bench_line.py: the contract's keys, `roofline`, `compute`,
        # `cpu_baseline`, one number per secondary leg; < 4 KB).  Everything measured, prose included, goes to the detail file
        # and to stderr.
        detail = json.dumps(result)
        detail_path = os.environ.get("MPVSS_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(detail_path, "w") as fh:
                fh.write(detail + "\n")
            result["detail"] = os.path.relpath(detail_path, ROOT)
        except OSError:
            result["detail"] = "stderr"
        sys.stderr.write(detail + "\n")
        sys.stderr.flush()
        print(bench_line.compact_line(result), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    if secondary_error:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
