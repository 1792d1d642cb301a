"""The multi-rank flow of bench.py with the REAL engine: two fresh processes under torch.distributed.run, both on the
one GPU of the test box (MPVSS_BENCH_SMOKE_ONE_GPU=1: gloo instead of RCCL for the rendezvous, everything else as in
the N > 1 runs the driver launches).  Each rank holds a contiguous block of the box, the SHA-256 running state travels
rank 0 -> rank 1 per box, the verdicts are broadcast; bench.py aborts unless every box verifies with the dealer's
transcript digest, so a clean exit with a JSON line IS the parity check."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _line(out):
    """(compact, detail): the ONE stdout line the driver parses -- held to bench_line.py's contract here, on a real run --
    and the full object bench.py writes to stderr (and bench_detail.json) beside it."""
    sys.path.insert(0, ROOT)
    import bench_line
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{\"metric\"")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0].encode()) < bench_line.MAX_LINE_BYTES
    compact = json.loads(lines[0])
    for k in bench_line.REQUIRED:
        if k != "cpu_baseline":              # rank 0 at N = 1 only
            assert k in compact, k
    detail = json.loads([ln for ln in out.stderr.splitlines() if ln.startswith("{\"metric\"")][-1])
    assert compact["value"] == pytest.approx(detail["value"], rel=1e-5) and compact["n_gpus"] == detail["n_gpus"]
    return compact, detail


def _run(ranks, steps, n, depth, timeout=900, extra_env=None, config_boxes="0", scaling="weak", threshold="64"):
    env = dict(os.environ, MPVSS_BENCH_SMOKE_ONE_GPU="1", MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    if depth is not None:                  # None: the boxes in flight per rank follow bench.py's own formula for that world size
        env["MPVSS_BENCH_DEPTH"] = str(depth)
    else:
        env.pop("MPVSS_BENCH_DEPTH", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps",
           str(steps), "--warmup", "1", "--participants", str(n), "--threshold", threshold, "--cpu-sample", "0", "--wb-shares", "0",
           "--registered-keys", "0", "--ec-boxes", "0", "--lone-boxes", "0", "--host-boxes", "0", "--config-boxes", config_boxes,
           "--scaling", scaling, "--drop-in-threads", "0", "--steady-steps", "0"]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_the_drivers_own_command_starts_its_ranks():
    """Exactly what the driver runs for N > 1 -- `python3 bench.py --gpus 2 --steps 3 --warmup 1`, no launcher around it:
    bench.py starts its two ranks itself (a child `torch.distributed.run`, before the parent touches the GPU), relays
    rank 0's ONE JSON line and its exit code.  Headline shape per rank (65536, 256), three different dealers' boxes; each
    box's per-share well-formedness bytes are all-gathered over the default process group (gloo here, RCCL on a real
    node) and the line says how many ranks that group had."""
    env = dict(os.environ, MPVSS_BENCH_SMOKE_ONE_GPU="1", MPVSS_BENCH_DEPTH="3")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{\"metric\""), out.stdout[-2000:]           # ONE line on stdout
    compact, res = _line(out)
    assert compact["rccl"]["rccl_world_size"] == 2 and compact["roofline"]["bound"] == "hbm" and compact["config"]["t"] == 256
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["n_per_gpu"] == 65536 and "131072 participants in the box" in res["config"]["workload_detail"]
    assert res["config"]["distinct_boxes"] == 3
    assert res["value"] > 0 and res["compute"]["fd_fallbacks"] == 0
    assert res["host"]["pipeline"].startswith("mpvss_modp_verify_many_chained")      # N > 1 runs the library's own pipeline
    # one data collective per box over a group of two ranks: slot initialisation + warm-up + timed boxes at least
    assert res["rccl"]["rccl_world_size"] == 2 and res["rccl"]["per_box"] == 1 and res["rccl"]["data_collectives"] >= 3 + 1
    assert res["rccl"]["bytes_per_rank_per_box"] == 65536


def test_four_ranks_many_boxes_every_rank_absorbing_several_at_once():
    """The driver's N > 1 shape in small: 4 ranks (one GPU), 14 timed boxes of 4 x 4096 shares, 6 boxes in flight per rank
    and as many hash threads claiming blocks, receiving by box tag and sending on -- a lost or crossed state would fail
    bench.py's digest check, a missing recv would hang into the timeout."""
    out = _run(4, 14, 4096, 6, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    _, res = _line(out)
    assert res["n_gpus"] == 4 and res["steps"] == 14 and "16384 participants in the box" in res["config"]["workload_detail"]
    assert res["host"]["hash_threads"] >= 2 and res["compute"]["fd_fallbacks"] == 0
    assert res["rccl"]["rccl_world_size"] == 4 and res["rccl"]["data_collectives"] >= 14


def test_the_c5_object_of_an_eight_gpu_run_in_small():
    """At N = 8 the line also carries `c5`: ONE box of 8 x 131072 participants, t = 1024, every rank its block.  That leg
    cannot run here at size; the same code path with two ranks, 2 x 16384 participants and t = 64 (MPVSS_BENCH_C5*): own
    participants per rank, boxes dealt across the ranks (the dealer's transcript chained rank to rank), slots grown in an
    untimed pass, per-box flag gather with re-sized buffers, and the headline figure of the same run untouched."""
    out = _run(2, 3, 8192, 3, extra_env={"MPVSS_BENCH_C5": "1", "MPVSS_BENCH_C5_N": "16384", "MPVSS_BENCH_C5_T": "64"},
               config_boxes="3")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    _, res = _line(out)
    assert "secondary_error" not in res
    c5 = res["c5"]
    assert c5["boxes"] == 3 and c5["value"] > 0 and "32768 participants over 2 GPUs" in c5["config"]["workload"]
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["rccl"]["data_collectives"] >= 3 + 3


@pytest.fixture(scope="module")
def eight_rank_run():
    """ONE eight-rank run for the two tests below: `--scaling both` (the default the driver gets at N > 1), the forced-small `c5` object."""
    out = _run(8, 6, 2048, None, timeout=600, extra_env={"MPVSS_BENCH_C5_N": "4096", "MPVSS_BENCH_C5_T": "64", "OMP_NUM_THREADS": "2"},
               config_boxes="3", scaling="both")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return _line(out)


def test_eight_ranks_at_the_real_world_size(eight_rank_run):
    """The first real `--gpus 8` is the driver's: everything that depends on the WORLD SIZE rather than on the GPU count runs
    here with eight ranks on the one GPU of the test box (gloo) -- the boxes in flight per rank from bench.py's own formula
    (no MPVSS_BENCH_DEPTH), eight hash threads per rank receiving the running state by box tag from the rank before and sending
    it on, the per-box all-gather of eight ranks' flag bytes reaped in order, the verdict broadcast from the last rank, and the
    `c5` object that only an eight-rank line carries (forced small: 8 x 4096 participants, t = 64).  A clean exit with the line
    IS the parity check (bench.py aborts unless every box verifies with the dealer's digest on every rank)."""
    _, res = eight_rank_run
    assert "secondary_error" not in res
    assert res["n_gpus"] == 8 and res["steps"] == 6 and "16384 participants in the box" in res["config"]["workload_detail"]
    assert res["rccl"]["rccl_world_size"] == 8 and res["rccl"]["data_collectives"] >= 6
    assert res["host"]["boxes_in_flight"] >= 8 + 5            # 8 + one more box per 58 ms of chain latency at eight ranks + 2
    assert res["host"]["hbm"]["bytes_in_use_on_this_rank"] > 0 and res["host"]["hbm"]["bytes_total"] > 2 ** 37
    c5 = res["c5"]
    assert c5["boxes"] == 3 and c5["value"] > 0 and "32768 participants over 8 GPUs" in c5["config"]["workload"]
    assert res["compute"]["fd_fallbacks"] == 0


def test_strong_scaling_the_same_box_over_one_and_eight_ranks(eight_rank_run):
    """`--scaling both` (the default at N > 1): beside the weak figure the line carries `strong` -- the metric's FIXED box (here 2048
    participants) split into N contiguous blocks, rank g verifying positions [g n/N, (g+1) n/N).  The box is made from one seeded list
    of participants whatever the world size, so the digests of its K boxes must be the SAME at 8 ranks (256 shares per rank) and in the
    one-rank run (MPVSS_BENCH_STRONG_AT_N1: the whole box on one engine): a lost or crossed running state, or a block at the wrong
    offset, changes them.  (`--scaling strong`, which makes that figure the line's `value`, is exercised two tests below.)"""
    compact, res = eight_rank_run
    st = res["strong"]
    assert st["scaling"] == "strong" and st["n_per_gpu"] == 256 and st["boxes"] == 6 and st["value"] > 0
    assert compact["strong"]["value"] == pytest.approx(st["value"], rel=1e-5)
    assert res["scaling"] == "weak" and res["config"]["n_per_gpu"] == 2048            # `value` is still the weak figure
    out = _run(1, 6, 2048, 4, timeout=420, scaling="both", extra_env={"MPVSS_BENCH_STRONG_AT_N1": "1"})
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    _, one = _line(out)
    assert one["strong"]["n_per_gpu"] == 2048 and one["strong"]["boxes"] == 6
    assert one["strong"]["digest_of_digests"] == st["digest_of_digests"]


def test_the_python_driven_blocks_still_agree_with_the_chained_pipeline():
    """MPVSS_BENCH_CHAINED=0 keeps round 3's N > 1 flow (blocks driven from a Python thread pool: compute / claim /
    absorb_claimed) beside the library's chained pipeline; both must verify the same boxes with the dealers' digests."""
    for mode, name in (("0", "verify_block_compute / block_claim / absorb_claimed"), ("1", "mpvss_modp_verify_many_chained")):
        # (the chained run with `--scaling strong`: the timed region itself is the fixed 8192-share box split over the two ranks)
        out = _run(2, 5, 4096 if mode == "0" else 8192, 4, timeout=420, extra_env={"MPVSS_BENCH_CHAINED": mode},
                   scaling="weak" if mode == "0" else "strong")
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        _, res = _line(out)
        assert res["host"]["pipeline"].startswith(name) and res["rccl"]["data_collectives"] >= 5 and res["value"] > 0
        if mode == "1":
            assert res["scaling"] == "strong" and res["config"]["n_per_gpu"] == 4096 and "strong" not in res
            assert "8192 participants in the box" in res["config"]["workload_detail"]
