"""The ONE stdout line of bench.py, kept small enough for the driver's parser.

Round 4's line had grown to 20.9 KB of numbers and prose and the driver recorded `"parsed": null` for it.  bench.py now
collects everything it measures in one `result` object as before, writes that object to `bench_detail.json` (and to
stderr), and prints `compact_line(result)`: the contract's keys, `roofline`, `compute`, `cpu_baseline` and one number per
secondary leg -- numbers and short identifiers only, no prose.  `tests/test_bench_line.py` holds it to < 4096 bytes on a
canned result of round 4's size.

No torch, no engine: importable on any box.
"""
import json

MAX_LINE_BYTES = 4096
MAX_STR = 96          # a string longer than this is prose and belongs in the detail file

# (path in the full result) -> kept as is when present.  Paths are tuples of keys; a trailing dict is never copied
# wholesale, only the leaves named here.
_TOP = ("metric", "value", "value_steady_state", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data")
_PATHS = (
    ("config", "workload"), ("config", "n_per_gpu"), ("config", "t"), ("config", "parallelism"),
    ("config", "distinct_boxes"),
    ("roofline", "bound"), ("roofline", "kernel"), ("roofline", "achieved"), ("roofline", "peak"), ("roofline", "unit"),
    ("roofline", "frac"), ("roofline", "traffic"), ("roofline", "kernel_ms"), ("roofline", "kernel_ms_overlapped"),
    ("roofline", "shares_per_launch"), ("roofline", "algorithmic_bytes_per_share"),
    ("compute", "bound"), ("compute", "achieved"), ("compute", "peak"), ("compute", "frac"),
    ("compute", "peak_mix_weighted"), ("compute", "frac_mix_weighted"),
    ("compute", "vs_sustained_mad64"), ("compute", "valu_slots_per_share"), ("compute", "modmul_per_share"),
    ("compute", "a2_kernel_alone", "ms"), ("compute", "a2_kernel_alone", "frac"),
    ("compute", "a2_kernel_alone", "frac_mix_weighted"),
    ("compute", "a2_kernel_alone", "vs_sustained_mad64"),
    ("compute", "sustained_probe", "mad64_insts_per_s"), ("compute", "sustained_probe", "mad64_shader_clock_ghz"),
    ("compute", "sustained_probe", "alu32_insts_per_s"),
    ("compute", "fd_blocks"), ("compute", "fd_fallbacks"),
    ("cpu_baseline", "value"), ("cpu_baseline", "unit"), ("cpu_baseline", "cores"), ("cpu_baseline", "kind"),
    ("cpu_baseline", "cpu_model"), ("cpu_baseline", "sample"), ("cpu_baseline", "single_thread", "value"),
    ("cpu_baseline", "openssl", "value"),
    ("host", "hash_threads"), ("host", "boxes_in_flight"), ("host", "per_box_ms", "sha256_transcript"),
    ("host", "hbm", "bytes_in_use_on_this_rank"),
    ("configs", "c2", "value"), ("configs", "c2", "compute", "frac"),
    ("configs", "c5_slice", "value"), ("configs", "c5_slice", "ms_per_box"), ("configs", "c5_slice", "compute", "frac"),
    ("c5", "value"), ("c5", "ms_per_box"),
    ("strong", "value"), ("strong", "ms_per_box"), ("strong", "n_per_gpu"), ("strong", "scaling"),
    ("c5_whole_box", "value"), ("c5_whole_box", "ms_per_box"), ("c5_whole_box", "blocks"),
    ("rccl", "backend"), ("rccl", "rccl_world_size"), ("rccl", "data_collectives"), ("rccl", "bytes_per_rank_per_box"),
    ("verify_share", "value"), ("extract_shares", "value"),
    ("distribute", "value"), ("distribute", "value_end_to_end"), ("distribute", "value_one_call_host_buffers_end_to_end"),
    ("host_buffers", "value"),
    ("one_call", "c1", "deal_ms"), ("one_call", "c1", "verify_distribution_ms"), ("one_call", "c2", "verify_distribution_ms"),
    ("drop_in", "value"), ("drop_in", "threads"), ("drop_in", "value_lone"), ("drop_in", "value_host_buffers"),
    ("drop_in", "value_key_cache"), ("drop_in", "value_lone_key_cache"), ("drop_in", "deal_value"), ("drop_in", "deal_value_key_cache"),
    ("registered_keys", "value"), ("registered_keys", "value_steady_state"), ("registered_keys", "value_transparent"), ("registered_keys", "table_bytes"), ("registered_keys", "table_build_s"),
    ("ec", "secp256k1", "value"), ("ec", "secp256k1", "compute", "frac"),
    ("ec", "secp256k1", "kernel_ms_isolated", "box_on_the_gpu_ms"),
    ("ec", "secp256k1", "distribute", "value_end_to_end"), ("ec", "secp256k1", "distribute", "value_one_call_host_buffers_end_to_end"),
    ("ec", "secp256k1", "cpu_baseline", "value"),
    ("ec", "ristretto255", "value"), ("ec", "ristretto255", "compute", "frac"),
    ("ec", "ristretto255", "kernel_ms_isolated", "box_on_the_gpu_ms"),
    ("ec", "ristretto255", "distribute", "value_end_to_end"), ("ec", "ristretto255", "distribute", "value_one_call_host_buffers_end_to_end"),
    ("ec", "ristretto255", "cpu_baseline", "value"),
    ("secondary_error",), ("detail",),
)
# dropped in this order, a whole top-level object at a time, should a line still come out too long
_EXPENDABLE = ("host", "host_buffers", "extract_shares", "verify_share", "registered_keys", "distribute", "ec", "configs",
               "rccl", "c5", "c5_whole_box", "strong", "drop_in")
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _get(obj, path):
    for k in path:
        if not isinstance(obj, dict) or k not in obj:
            return None, False
        obj = obj[k]
    return obj, True


def _put(obj, path, v):
    for k in path[:-1]:
        obj = obj.setdefault(k, {})
    obj[path[-1]] = v


def _leaf(v):
    """Numbers rounded to 6 significant digits (the line is read by people and parsers, not diffed), prose cut."""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.6g}")
    if isinstance(v, str):
        return v if len(v) <= MAX_STR else v[:MAX_STR - 1] + "~"
    return None          # lists and dicts are never leaves of the compact line


def compact(result):
    out = {}
    for k in _TOP:
        if k in result:
            out[k] = _leaf(result[k])
    for path in _PATHS:
        v, ok = _get(result, path)
        if ok and not isinstance(v, (dict, list)):
            _put(out, path, _leaf(v))
    return out


def compact_line(result):
    """One line of JSON, below MAX_LINE_BYTES, holding every key the driver's contract names."""
    out = compact(result)
    line = json.dumps(out, separators=(",", ":"))
    for k in _EXPENDABLE:
        if len(line.encode()) < MAX_LINE_BYTES:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line.encode()) >= MAX_LINE_BYTES:
        raise ValueError(f"compact bench line is {len(line)} bytes")
    return line
