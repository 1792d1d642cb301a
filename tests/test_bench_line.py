"""bench.py's stdout line must stay parseable by the driver: round 4's 20.9 KB line was recorded as `"parsed": null`.
`bench_line.compact_line` is held here to < 4096 bytes, valid JSON, and every key the contract names, on round 4's own
full result (profiles/r04_bench_full_line.json, the line that broke the parser) and on inflated / degenerate variants."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_line  # noqa: E402


@pytest.fixture(scope="module")
def full():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench_full_line.json")))


def test_round_4s_line_was_too_long_and_its_compact_form_is_not(full):
    assert len(json.dumps(full)) > 16000                       # the object that lost the round's number
    line = bench_line.compact_line(full)
    assert "\n" not in line and len(line.encode()) < 4096
    res = json.loads(line)
    for k in bench_line.REQUIRED:
        assert k in res, k
    assert res["metric"].startswith("DLEQ share verifications/sec, 2048-bit MODP, n=65536 t=256")
    assert res["value"] == pytest.approx(full["value"], rel=1e-5) and res["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert res["n_gpus"] == 1 and res["steps"] == 20 and res["warmup"] == 5 and res["higher_is_better"] is True
    assert res["scaling"] == "weak" and res["vs_baseline"] is None and res["data"] == "synthetic"
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_modp_dual_exp_w6_pair"
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4) and rf["traffic"] == full["roofline"]["traffic"]
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == full["cpu_baseline"]["cores"] and cb["value"] > 0 and cb["sample"]
    assert cb["single_thread"]["value"] > 0 and cb["openssl"]["value"] > 0
    assert res["compute"]["frac"] == pytest.approx(full["compute"]["frac"], rel=1e-5)
    assert res["configs"]["c5_slice"]["value"] > 0 and res["ec"]["secp256k1"]["value"] > 0 and res["ec"]["ristretto255"]["value"] > 0


def test_no_prose_survives(full):
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    res = json.loads(bench_line.compact_line(full))
    assert all(len(s) <= bench_line.MAX_STR for s in strings(res))
    assert "note" not in json.dumps(res)


def test_an_inflated_result_sheds_secondary_legs_but_never_the_contract(full):
    big = copy.deepcopy(full)
    for k in range(400):                                        # secondary objects the whitelist does not know stay out
        big[f"extra_{k}"] = {"value": 1.0 * k, "note": "x" * 200}
    assert len(bench_line.compact_line(big).encode()) < 4096
    worst = copy.deepcopy(full)                                 # every whitelisted string at its cap
    worst["cpu_baseline"]["sample"] = "s" * 500
    worst["config"]["workload"] = "w" * 500
    worst["secondary_error"] = "e" * 5000
    line = bench_line.compact_line(worst)
    assert len(line.encode()) < 4096
    res = json.loads(line)
    for k in bench_line.REQUIRED:
        assert k in res
    assert len(res["secondary_error"]) <= bench_line.MAX_STR


def test_a_multi_rank_line_without_secondary_legs(full):
    r = {k: full[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                              "dtype", "data", "config", "roofline", "compute")}
    r["n_gpus"] = 8
    r["rccl"] = {"backend": "nccl", "rccl_world_size": 8, "data_collectives": 33, "per_box": 1, "bytes_per_rank_per_box": 65536,
                 "collective": "c" * 300}
    r["value"] = float("nan")                                    # a broken figure becomes null, not invalid JSON
    res = json.loads(bench_line.compact_line(r))
    assert res["value"] is None and res["rccl"] == {"backend": "nccl", "rccl_world_size": 8, "data_collectives": 33,
                                                    "bytes_per_rank_per_box": 65536}
    assert "cpu_baseline" not in res                            # rank 0 at N = 1 only
