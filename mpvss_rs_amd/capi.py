"""ctypes binding of the engine's C ABI (include/mpvss_hip.h).

This is the same set of symbols a Rust `extern "C"` block would bind.  There
is NO CPU fallback here or in the library: if libmpvss_hip.so is missing, or no HIP device is
present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPVSS_HIP_LIB", os.path.join(_HERE, "libmpvss_hip.so"))   # override only for A/B tuning builds

MPVSS_HOST = 0
MPVSS_DEVICE = 1
EB = 256

# every symbol include/mpvss_hip.h declares
EXPORTED_SYMBOLS = (
    "mpvss_device_count", "mpvss_ctx_create", "mpvss_ctx_destroy", "mpvss_last_error",
    "mpvss_ctx_set_stream", "mpvss_ctx_synchronize",
    "mpvss_modp_batch_exp", "mpvss_modp_batch_mul", "mpvss_modp_batch_exp_fixed_base",
    "mpvss_modp_commit_eval", "mpvss_modp_dleq_commitments", "mpvss_modp_verify_distribution",
    "mpvss_modp_verify_shares", "mpvss_modp_distribute", "mpvss_sha256", "mpvss_modp_hash_to_scalar",
    "mpvss_last_kernel_ms",
    "mpvss_transcript_init", "mpvss_modp_verify_block_compute", "mpvss_modp_verify_block_absorb",
    "mpvss_block_claim", "mpvss_modp_verify_block_absorb_claimed",
    "mpvss_modp_transcript_verdict", "mpvss_modp_transcript_absorb",
    "mpvss_ec_batch_exp", "mpvss_ec_batch_mul", "mpvss_ec_commit_eval", "mpvss_ec_dleq_commitments",
    "mpvss_ec_verify_distribution", "mpvss_ec_verify_shares", "mpvss_ec_verify_shares_compute", "mpvss_ec_verify_shares_absorb", "mpvss_ec_distribute", "mpvss_ec_distribute_compute", "mpvss_ec_distribute_absorb", "mpvss_ec_hash_to_scalar",
    "mpvss_ec_block_absorb_claimed", "mpvss_ec_poly_eval_device", "mpvss_ec_dleq_responses_device", "mpvss_ec_deal_compute", "mpvss_ec_deal",
    "mpvss_modp_extract_shares", "mpvss_ec_extract_shares", "mpvss_last_kernel_launches",
    "mpvss_modp_keyset_create", "mpvss_modp_keyset_destroy", "mpvss_modp_keyset_bytes", "mpvss_ctx_set_key_cache", "mpvss_ctx_set_key_cache_lru",
    "mpvss_modp_verify_block_compute_keyset", "mpvss_modp_fd_stats",
    "mpvss_modp_verify_many", "mpvss_modp_verify_many_chained", "mpvss_pipeline_stats_get", "mpvss_blocks_in_flight", "mpvss_sha256_uses_shani", "mpvss_issue_probe",
    "mpvss_modp_verify_shares_compute", "mpvss_modp_verify_shares_absorb",
    "mpvss_ec_batch_exp_generator", "mpvss_ec_verify_block_compute", "mpvss_ec_verify_block_absorb",
    "mpvss_ec_transcript_absorb", "mpvss_ec_transcript_verdict", "mpvss_ec_verify_many",
    "mpvss_modp_scalar_mul", "mpvss_modp_scalar_sub", "mpvss_ec_scalar_mul", "mpvss_ec_scalar_sub",
    "mpvss_modp_dleq_responses", "mpvss_ec_dleq_responses", "mpvss_modp_poly_eval", "mpvss_ec_poly_eval",
    "mpvss_modp_poly_eval_device", "mpvss_modp_dleq_responses_device", "mpvss_modp_deal_compute", "mpvss_modp_deal_compute_keyset", "mpvss_modp_deal",
    "mpvss_modp_reconstruct", "mpvss_ec_reconstruct",
    "mpvss_box_wire_size", "mpvss_box_serialize", "mpvss_box_parse", "mpvss_box_verify_wire",
    "mpvss_modp_distribute_compute", "mpvss_modp_distribute_absorb",
    "mpvss_process_init", "mpvss_modp_verify_block_compute_flags",
    "mpvss_modp_extract_shares_compute", "mpvss_modp_extract_shares_absorb",
)

GROUP_SECP256K1 = 1
GROUP_RISTRETTO255 = 2
EC_ENC = {GROUP_SECP256K1: 33, GROUP_RISTRETTO255: 32}

TRANSCRIPT_STATE_BYTES = 128
BLOCK_SLOTS = 64                 # MPVSS_BLOCK_SLOTS: blocks of the block API in flight per context


class EngineError(RuntimeError):
    pass


class ModpBox(C.Structure):
    """struct mpvss_modp_box (include/mpvss_hip.h)"""
    _fields_ = [("commitments", C.c_void_p), ("t", C.c_size_t), ("positions", C.c_void_p), ("pubkeys", C.c_void_p),
                ("shares", C.c_void_p), ("responses", C.c_void_p), ("n", C.c_size_t), ("challenge_host", C.c_void_p),
                ("keyset", C.c_void_p), ("key_offset", C.c_size_t)]


class EcBox(C.Structure):
    """struct mpvss_ec_box"""
    _fields_ = [("commitments", C.c_void_p), ("t", C.c_size_t), ("positions", C.c_void_p), ("pubkeys", C.c_void_p),
                ("shares", C.c_void_p), ("responses", C.c_void_p), ("n", C.c_size_t), ("challenge_host", C.c_void_p)]


class BoxView(C.Structure):
    """struct mpvss_box_view"""
    _fields_ = [("group", C.c_int), ("element_bytes", C.c_size_t), ("scalar_bytes", C.c_size_t), ("n", C.c_size_t),
                ("t", C.c_size_t), ("u_len", C.c_size_t), ("commitments", C.c_void_p), ("positions", C.c_void_p),
                ("pubkeys", C.c_void_p), ("shares", C.c_void_p), ("responses", C.c_void_p), ("challenge", C.c_void_p),
                ("u_be", C.c_void_p)]


class PipelineStats(C.Structure):
    """struct mpvss_pipeline_stats"""
    _fields_ = [("enqueue_ms", C.c_double), ("wait_ms", C.c_double), ("hash_ms", C.c_double),
                ("kernel_ms", C.c_double * 4), ("kernel_launches", C.c_ulonglong * 4), ("blocks", C.c_ulonglong)]


# state_in / state_out of mpvss_modp_verify_many_chained: int cb(void* user, size_t box, uint8_t* state, int ok)
CHAIN_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8), C.c_int)


def load_library() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mpvss_rs_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, u8p, i64p, sz, ci = C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int
    lib.mpvss_device_count.restype = ci
    lib.mpvss_ctx_create.argtypes = [ci, C.POINTER(vp)]
    lib.mpvss_ctx_destroy.argtypes = [vp]
    lib.mpvss_ctx_destroy.restype = None
    lib.mpvss_last_error.argtypes = [vp]
    lib.mpvss_last_error.restype = C.c_char_p
    lib.mpvss_ctx_set_stream.argtypes = [vp, vp]
    lib.mpvss_ctx_synchronize.argtypes = [vp]
    lib.mpvss_modp_fd_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.mpvss_modp_keyset_create.argtypes = [vp, ci, u8p, sz, C.POINTER(vp)]
    lib.mpvss_ctx_set_key_cache.argtypes = [vp, ci]
    lib.mpvss_ctx_set_key_cache_lru.argtypes = [vp, ci, ci]
    lib.mpvss_ctx_set_key_cache_lru.restype = ci
    lib.mpvss_modp_keyset_destroy.argtypes = [vp, vp]
    lib.mpvss_modp_keyset_destroy.restype = None
    lib.mpvss_modp_keyset_bytes.argtypes = [vp]
    lib.mpvss_modp_keyset_bytes.restype = sz
    lib.mpvss_modp_verify_block_compute_keyset.argtypes = [vp, ci, u8p, sz, i64p, vp, sz, u8p, u8p, sz, u8p]
    lib.mpvss_modp_batch_exp.argtypes = [vp, ci, u8p, u8p, sz, u8p]
    lib.mpvss_modp_batch_mul.argtypes = [vp, ci, u8p, u8p, sz, u8p]
    lib.mpvss_modp_batch_exp_fixed_base.argtypes = [vp, ci, u8p, u8p, sz, u8p]
    lib.mpvss_modp_commit_eval.argtypes = [vp, ci, u8p, sz, i64p, sz, u8p]
    lib.mpvss_modp_dleq_commitments.argtypes = [vp, ci, u8p, u8p, u8p, u8p, u8p, u8p, ci, sz, u8p, u8p]
    lib.mpvss_modp_verify_distribution.argtypes = [vp, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p,
                                                   C.POINTER(ci), u8p, u8p, u8p, u8p]
    lib.mpvss_modp_verify_shares.argtypes = [vp, ci, u8p, u8p, u8p, u8p, u8p, sz, u8p]
    lib.mpvss_modp_verify_shares_compute.argtypes = [vp, ci, u8p, u8p, u8p, u8p, u8p, sz, u8p]
    lib.mpvss_modp_verify_shares_absorb.argtypes = [vp, u8p]
    lib.mpvss_modp_distribute.argtypes = [vp, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_modp_distribute_compute.argtypes = [vp, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p, u8p, u8p, u8p]
    lib.mpvss_modp_distribute_absorb.argtypes = [vp, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_sha256.argtypes = [u8p, sz, u8p]
    lib.mpvss_sha256.restype = None
    lib.mpvss_modp_hash_to_scalar.argtypes = [u8p, sz, u8p]
    lib.mpvss_modp_hash_to_scalar.restype = None
    lib.mpvss_last_kernel_ms.argtypes = [vp, ci]
    lib.mpvss_last_kernel_ms.restype = C.c_double
    lib.mpvss_last_kernel_launches.argtypes = [vp, ci]
    lib.mpvss_transcript_init.argtypes = [u8p]
    lib.mpvss_transcript_init.restype = None
    lib.mpvss_modp_verify_block_compute.argtypes = [vp, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p]
    lib.mpvss_modp_verify_block_absorb.argtypes = [vp, u8p, u8p, u8p, u8p]
    lib.mpvss_modp_verify_block_compute_flags.argtypes = [vp, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p, u8p]
    lib.mpvss_process_init.restype = ci
    lib.mpvss_modp_extract_shares_compute.argtypes = [vp, u8p, u8p, u8p, u8p, sz]
    lib.mpvss_modp_extract_shares_absorb.argtypes = [vp, u8p, u8p]
    lib.mpvss_block_claim.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    lib.mpvss_modp_verify_block_absorb_claimed.argtypes = [vp, C.c_ulonglong, u8p, u8p, u8p, u8p]
    lib.mpvss_modp_transcript_verdict.argtypes = [u8p, u8p, C.POINTER(ci), u8p]
    lib.mpvss_modp_transcript_absorb.argtypes = [u8p, u8p, sz]
    lib.mpvss_ec_batch_exp.argtypes = [vp, ci, ci, u8p, u8p, sz, u8p]
    lib.mpvss_ec_batch_mul.argtypes = [vp, ci, ci, u8p, u8p, sz, u8p]
    lib.mpvss_ec_commit_eval.argtypes = [vp, ci, ci, u8p, sz, i64p, sz, u8p]
    lib.mpvss_ec_dleq_commitments.argtypes = [vp, ci, ci, u8p, u8p, u8p, u8p, u8p, u8p, ci, sz, u8p, u8p]
    lib.mpvss_ec_verify_distribution.argtypes = [vp, ci, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p,
                                                 C.POINTER(ci), u8p, u8p, u8p, u8p]
    lib.mpvss_ec_verify_shares.argtypes = [vp, ci, ci, u8p, u8p, u8p, u8p, u8p, sz, u8p]
    lib.mpvss_ec_verify_shares_compute.argtypes = [vp, ci, ci, u8p, u8p, u8p, u8p, u8p, sz, u8p]
    lib.mpvss_ec_verify_shares_absorb.argtypes = [vp, u8p]
    lib.mpvss_ec_distribute.argtypes = [vp, ci, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_distribute_compute.argtypes = [vp, ci, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_distribute_absorb.argtypes = [vp, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_block_absorb_claimed.argtypes = [vp, C.c_ulonglong, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_poly_eval_device.argtypes = [vp, ci, u8p, sz, vp, sz, vp]
    lib.mpvss_ec_dleq_responses_device.argtypes = [vp, ci, vp, vp, u8p, sz, vp]
    lib.mpvss_ec_deal_compute.argtypes = [vp, ci, u8p, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp]
    lib.mpvss_ec_deal.argtypes = [vp, ci, u8p, sz, i64p, u8p, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_hash_to_scalar.argtypes = [ci, u8p, sz, u8p]
    lib.mpvss_ec_batch_exp_generator.argtypes = [vp, ci, ci, u8p, sz, u8p]
    lib.mpvss_box_wire_size.argtypes = [ci, sz, sz, sz]
    lib.mpvss_box_wire_size.restype = sz
    lib.mpvss_box_serialize.argtypes = [ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p, u8p, sz, u8p, sz, C.POINTER(sz)]
    lib.mpvss_box_parse.argtypes = [u8p, sz, C.POINTER(BoxView)]
    lib.mpvss_box_verify_wire.argtypes = [vp, u8p, sz, C.POINTER(ci), u8p]
    lib.mpvss_modp_scalar_mul.argtypes = [u8p, u8p, u8p]
    lib.mpvss_modp_scalar_sub.argtypes = [u8p, u8p, u8p]
    lib.mpvss_ec_scalar_mul.argtypes = [ci, u8p, u8p, u8p]
    lib.mpvss_ec_scalar_sub.argtypes = [ci, u8p, u8p, u8p]
    lib.mpvss_modp_dleq_responses.argtypes = [u8p, u8p, u8p, ci, sz, u8p, ci]
    lib.mpvss_ec_dleq_responses.argtypes = [ci, u8p, u8p, u8p, ci, sz, u8p, ci]
    lib.mpvss_modp_poly_eval.argtypes = [u8p, sz, i64p, sz, u8p, ci]
    lib.mpvss_modp_poly_eval_device.argtypes = [vp, u8p, sz, vp, sz, vp]
    lib.mpvss_modp_dleq_responses_device.argtypes = [vp, vp, vp, u8p, sz, vp]
    lib.mpvss_modp_deal_compute.argtypes = [vp, u8p, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp]
    lib.mpvss_modp_deal_compute_keyset.argtypes = [vp, u8p, sz, vp, vp, sz, vp, sz, vp, vp, vp, vp, vp]
    lib.mpvss_modp_deal.argtypes = [vp, u8p, sz, i64p, u8p, u8p, sz, u8p, u8p, u8p, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_poly_eval.argtypes = [ci, u8p, sz, i64p, sz, u8p, ci]
    lib.mpvss_modp_reconstruct.argtypes = [vp, ci, i64p, u8p, sz, u8p, u8p]
    lib.mpvss_ec_reconstruct.argtypes = [vp, ci, ci, i64p, u8p, sz, u8p, u8p]
    lib.mpvss_ec_verify_block_compute.argtypes = [vp, ci, ci, u8p, sz, i64p, u8p, u8p, u8p, sz, u8p]
    lib.mpvss_ec_verify_block_absorb.argtypes = [vp, u8p, u8p, u8p, u8p]
    lib.mpvss_ec_transcript_absorb.argtypes = [ci, u8p, u8p, sz]
    lib.mpvss_ec_transcript_verdict.argtypes = [ci, u8p, u8p, C.POINTER(ci), u8p]
    lib.mpvss_ec_verify_many.argtypes = [vp, ci, ci, C.POINTER(EcBox), sz, ci, ci, C.POINTER(ci), u8p]
    lib.mpvss_modp_verify_many.argtypes = [vp, ci, C.POINTER(ModpBox), sz, ci, ci, C.POINTER(ci), u8p]
    lib.mpvss_modp_verify_many_chained.argtypes = [vp, ci, C.POINTER(ModpBox), sz, ci, ci, C.POINTER(vp), CHAIN_CB, CHAIN_CB, vp,
                                                   C.POINTER(ci), u8p]
    lib.mpvss_pipeline_stats_get.argtypes = [vp, C.POINTER(PipelineStats), ci]
    lib.mpvss_blocks_in_flight.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.mpvss_sha256_uses_shani.restype = ci
    lib.mpvss_issue_probe.argtypes = [vp, ci, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mpvss_modp_extract_shares.argtypes = [vp, ci, u8p, u8p, u8p, u8p, sz, u8p, u8p]
    lib.mpvss_ec_extract_shares.argtypes = [vp, ci, ci, u8p, u8p, u8p, u8p, sz, u8p, u8p]
    return lib


def _buf(b: Optional[bytes]):
    """bytes -> (keepalive, void*)"""
    if b is None:
        return None, None
    arr = (C.c_uint8 * len(b)).from_buffer_copy(b)
    return arr, C.cast(arr, C.c_void_p)


def _out(nbytes: int):
    arr = (C.c_uint8 * max(nbytes, 1))()
    return arr, C.cast(arr, C.c_void_p)


class Engine:
    """One engine context bound to one GPU (host-buffer convenience API; device-pointer calls go
    through `.lib` / `.ctx` directly, see bench.py)."""

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        if self.lib.mpvss_device_count() <= 0:
            raise EngineError("no HIP device visible: the MI355X engine has no CPU fallback")
        ctx = C.c_void_p()
        rc = self.lib.mpvss_ctx_create(device_id, C.byref(ctx))
        if rc != 0:
            raise EngineError(f"mpvss_ctx_create failed: {rc}")
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mpvss_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.mpvss_last_error(self.ctx)
            raise EngineError(f"{what} failed: rc={rc} {msg.decode() if msg else ''}")

    def last_error(self) -> str:
        msg = self.lib.mpvss_last_error(self.ctx)
        return msg.decode() if msg else ""

    def verify_block_compute_flags(self, commitments: bytes, positions: Sequence[int], pubkeys: bytes, shares: bytes,
                                   responses: bytes, challenge: bytes, wellformed_dev_ptr: int) -> None:
        """mpvss_modp_verify_block_compute that also leaves one well-formedness byte per share at the DEVICE address
        wellformed_dev_ptr (n bytes, e.g. a torch uint8 tensor's data_ptr(); valid once the block is absorbed)."""
        n = len(positions)
        pos = (C.c_int64 * max(n, 1))(*positions)
        bufs = [_buf(x) for x in (commitments, pubkeys, shares, responses, challenge)]
        self._check(self.lib.mpvss_modp_verify_block_compute_flags(self.ctx, MPVSS_HOST, bufs[0][1], len(commitments) // EB,
                                                                   C.cast(pos, C.c_void_p), bufs[1][1], bufs[2][1], bufs[3][1], n,
                                                                   bufs[4][1], C.c_void_p(wellformed_dev_ptr)),
                    "verify_block_compute_flags")

    def issue_probe(self, kind: int, target_ms: float = 40.0) -> dict:
        """what the device sustains of the kernels' basic instruction (0: v_mad_u64_u32, 1: 32-bit integer work), 4 waves per SIMD"""
        rate, clk, ms = C.c_double(0), C.c_double(0), C.c_double(0)
        self._check(self.lib.mpvss_issue_probe(self.ctx, kind, float(target_ms), C.byref(rate), C.byref(clk), C.byref(ms)), "issue_probe")
        return {"insts_per_s": rate.value, "shader_clock_ghz": clk.value, "ms": ms.value}

    def kernel_ms(self, kernel_id: int) -> float:
        return self.lib.mpvss_last_kernel_ms(self.ctx, kernel_id)

    def kernel_launches(self, kernel_id: int) -> int:
        return self.lib.mpvss_last_kernel_launches(self.ctx, kernel_id)

    # ---- Group ops
    def batch_mul(self, a: bytes, b: bytes) -> bytes:
        n = len(a) // EB
        ka, pa = _buf(a); kb, pb = _buf(b); ko, po = _out(n * EB)
        self._check(self.lib.mpvss_modp_batch_mul(self.ctx, MPVSS_HOST, pa, pb, n, po), "batch_mul")
        return bytes(ko)[: n * EB]

    def batch_exp(self, bases: bytes, exps: bytes) -> bytes:
        n = len(bases) // EB
        ka, pa = _buf(bases); kb, pb = _buf(exps); ko, po = _out(n * EB)
        self._check(self.lib.mpvss_modp_batch_exp(self.ctx, MPVSS_HOST, pa, pb, n, po), "batch_exp")
        return bytes(ko)[: n * EB]

    def batch_exp_fixed_base(self, base: bytes, exps: bytes) -> bytes:
        n = len(exps) // EB
        ka, pa = _buf(base); kb, pb = _buf(exps); ko, po = _out(n * EB)
        self._check(self.lib.mpvss_modp_batch_exp_fixed_base(self.ctx, MPVSS_HOST, pa, pb, n, po),
                    "batch_exp_fixed_base")
        return bytes(ko)[: n * EB]

    def commit_eval(self, commitments: bytes, positions: Sequence[int]) -> bytes:
        t = len(commitments) // EB
        n = len(positions)
        kc, pc = _buf(commitments)
        pos = (C.c_int64 * max(n, 1))(*positions)
        ko, po = _out(n * EB)
        self._check(self.lib.mpvss_modp_commit_eval(self.ctx, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), n, po),
                    "commit_eval")
        return bytes(ko)[: n * EB]

    def dleq_commitments(self, g1: bytes, h1: bytes, g2: bytes, h2: bytes, r: bytes, c: bytes,
                         c_per_share: bool) -> Tuple[bytes, bytes]:
        n = len(h1) // EB
        k = [_buf(x) for x in (g1, h1, g2, h2, r, c)]
        k1, p1 = _out(n * EB); k2, p2 = _out(n * EB)
        self._check(self.lib.mpvss_modp_dleq_commitments(self.ctx, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1],
                                                         k[4][1], k[5][1], int(c_per_share), n, p1, p2),
                    "dleq_commitments")
        return bytes(k1)[: n * EB], bytes(k2)[: n * EB]

    def verify_distribution(self, commitments: bytes, positions: Sequence[int], pubkeys: bytes, shares: bytes,
                            responses: bytes, challenge: bytes, dump: bool = False):
        t = len(commitments) // EB
        n = len(positions)
        kc, pc = _buf(commitments); ky, py = _buf(pubkeys); kY, pY = _buf(shares); kr, pr = _buf(responses)
        kch, pch = _buf(challenge)
        pos = (C.c_int64 * max(n, 1))(*positions)
        verdict = C.c_int(0)
        kd, pd = _out(32)
        if dump:
            kx, px = _out(n * EB); k1, p1 = _out(n * EB); k2, p2 = _out(n * EB)
        else:
            kx = k1 = k2 = None
            px = p1 = p2 = None
        self._check(self.lib.mpvss_modp_verify_distribution(
            self.ctx, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), py, pY, pr, n, pch, C.byref(verdict), pd,
            px, p1, p2), "verify_distribution")
        out = {"verdict": bool(verdict.value), "digest": bytes(kd)[:32]}
        if dump:
            out.update(X=bytes(kx)[: n * EB], a1=bytes(k1)[: n * EB], a2=bytes(k2)[: n * EB])
        return out

    # ---- sharded verification (one engine per GPU; see mpvss_rs_amd/sharding.py)
    def verify_block_compute(self, commitments: bytes, positions: Sequence[int], pubkeys: bytes, shares: bytes,
                             responses: bytes, challenge: bytes) -> None:
        t = len(commitments) // EB
        n = len(positions)
        kc, pc = _buf(commitments); ky, py = _buf(pubkeys); kY, pY = _buf(shares); kr, pr = _buf(responses)
        kch, pch = _buf(challenge)
        pos = (C.c_int64 * max(n, 1))(*positions)
        self._check(self.lib.mpvss_modp_verify_block_compute(self.ctx, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p),
                                                             py, pY, pr, n, pch), "verify_block_compute")

    # ---- registered public keys (per-key tables for y^r, reused by every box verified against the same keys)
    def keyset_create(self, pubkeys: bytes):
        kk, pk = _buf(pubkeys)
        h = C.c_void_p()
        self._check(self.lib.mpvss_modp_keyset_create(self.ctx, MPVSS_HOST, pk, len(pubkeys) // EB, C.byref(h)),
                    "keyset_create")
        return h

    def set_key_cache(self, min_boxes: int) -> int:
        """verify_many builds per-key tables by itself for key arrays that >= min_boxes large boxes of one call share (0: off)"""
        rc = int(self.lib.mpvss_ctx_set_key_cache(self.ctx, int(min_boxes)))
        self._check(rc if rc < 0 else 0, "set_key_cache")
        return rc

    def set_key_cache_lru(self, max_sets: int, min_sightings: int = 2) -> int:
        """mpvss_ctx_set_key_cache_lru: key tables ACROSS one-box calls, keyed by the SHA-256 of the host key array (0: off).
        Returns the previous max_sets."""
        rc = int(self.lib.mpvss_ctx_set_key_cache_lru(self.ctx, int(max_sets), int(min_sightings)))
        if rc < 0:
            self._check(rc, "set_key_cache_lru")
        return rc

    def keyset_destroy(self, keyset) -> None:
        self.lib.mpvss_modp_keyset_destroy(self.ctx, keyset)

    def keyset_bytes(self, keyset) -> int:
        return int(self.lib.mpvss_modp_keyset_bytes(keyset))

    def verify_block_compute_keyset(self, commitments: bytes, positions: Sequence[int], keyset, key_offset: int,
                                    shares: bytes, responses: bytes, challenge: bytes) -> None:
        t = len(commitments) // EB
        n = len(positions)
        kc, pc = _buf(commitments); kY, pY = _buf(shares); kr, pr = _buf(responses)
        kch, pch = _buf(challenge)
        pos = (C.c_int64 * max(n, 1))(*positions)
        self._check(self.lib.mpvss_modp_verify_block_compute_keyset(self.ctx, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p),
                                                                    keyset, key_offset, pY, pr, n, pch),
                    "verify_block_compute_keyset")

    def verify_block_absorb_dump(self, state: bytes, n: int):
        """absorb the oldest block and also return its X, a1, a2 arrays"""
        ks, ps = _buf(state)
        kx, px = _out(n * EB); k1, p1 = _out(n * EB); k2, p2 = _out(n * EB)
        self._check(self.lib.mpvss_modp_verify_block_absorb(self.ctx, ps, px, p1, p2), "verify_block_absorb")
        return bytes(ks), bytes(kx)[: n * EB], bytes(k1)[: n * EB], bytes(k2)[: n * EB]

    def verify_many(self, boxes: Sequence[dict], depth: int = 8, hash_threads: int = 4):
        """boxes: dicts with commitments, positions, pubkeys, shares, responses, challenge (host bytes).
        Returns [(verdict, digest)] in box order."""
        keep, arr = [], (ModpBox * max(len(boxes), 1))()
        for i, b in enumerate(boxes):
            n = len(b["positions"])
            pos = (C.c_int64 * max(n, 1))(*b["positions"])
            bufs = [_buf(b[k]) for k in ("commitments", "pubkeys", "shares", "responses", "challenge")]
            keep.append((pos, bufs))
            arr[i] = ModpBox(bufs[0][1], len(b["commitments"]) // EB, C.cast(pos, C.c_void_p), bufs[1][1], bufs[2][1],
                             bufs[3][1], n, bufs[4][1], None, 0)
        verdicts = (C.c_int * max(len(boxes), 1))()
        kd, pd = _out(32 * len(boxes))
        self._check(self.lib.mpvss_modp_verify_many(self.ctx, MPVSS_HOST, arr, len(boxes), depth, hash_threads, verdicts, pd),
                    "verify_many")
        raw = bytes(kd)
        return [(bool(verdicts[i]), raw[32 * i:32 * i + 32]) for i in range(len(boxes))]

    def poly_eval_device(self, coeffs: bytes, positions_dev_ptr: int, n: int, out_dev_ptr: int) -> None:
        """P(i) mod (q-1) for n positions in HBM (int64) into n x 256 bytes in HBM; the coefficients are host bytes"""
        kc, pc = _buf(coeffs)
        self._check(self.lib.mpvss_modp_poly_eval_device(self.ctx, pc, len(coeffs) // EB, positions_dev_ptr, n, out_dev_ptr),
                    "poly_eval_device")

    def deal_compute(self, coeffs: bytes, positions_dev_ptr: int, pubkeys_dev_ptr: int, witnesses_dev_ptr: int, n: int,
                     p_dev_out_ptr: int) -> None:
        """one dealer's block, P(i) included, enqueued (absorb with distribute_absorb / mpvss_modp_distribute_absorb)"""
        kc, pc = _buf(coeffs)
        self._check(self.lib.mpvss_modp_deal_compute(self.ctx, pc, len(coeffs) // EB, positions_dev_ptr, pubkeys_dev_ptr,
                                                     witnesses_dev_ptr, n, p_dev_out_ptr, None, None, None, None), "deal_compute")

    def deal_compute_keyset(self, coeffs: bytes, positions_dev_ptr: int, keyset, key_offset: int, witnesses_dev_ptr: int, n: int,
                            p_dev_out_ptr: int) -> None:
        """deal_compute to registered keys (keyset_create): Y and a2 from the key tables, same outputs"""
        kc, pc = _buf(coeffs)
        self._check(self.lib.mpvss_modp_deal_compute_keyset(self.ctx, pc, len(coeffs) // EB, positions_dev_ptr, keyset, key_offset,
                                                            witnesses_dev_ptr, n, p_dev_out_ptr, None, None, None, None),
                    "deal_compute_keyset")

    def deal_call(self, coeffs: bytes, positions: Sequence[int], pubkeys: bytes, witnesses: bytes):
        """(call, outputs) for mpvss_modp_deal over ctypes buffers made ONCE: `call()` is the library call alone -- what a compiled
        caller (the Rust crate's distribute_secret over its own Vec<u8>s) pays per box --, `outputs()` the dict of deal()."""
        n = len(positions)
        pos = (C.c_int64 * max(n, 1))(*positions)
        k = [_buf(b) for b in (coeffs, pubkeys, witnesses)]
        outs = [_out(n * EB) for _ in range(5)]
        kd, pd = _out(32)
        kc, pc = _out(EB)
        t = len(coeffs) // EB

        def call():
            self._check(self.lib.mpvss_modp_deal(self.ctx, k[0][1], t, C.cast(pos, C.c_void_p), k[1][1], k[2][1], n,
                                                 outs[0][1], outs[1][1], outs[2][1], outs[3][1], pd, pc, outs[4][1]), "deal")

        def outputs():
            X, Y, a1, a2, r = (bytes(o[0])[: n * EB] for o in outs)
            return {"X": X, "Y": Y, "a1": a1, "a2": a2, "digest": bytes(kd)[:32], "challenge": bytes(kc)[:EB], "responses": r}
        return call, outputs

    def deal(self, coeffs: bytes, positions: Sequence[int], pubkeys: bytes, witnesses: bytes) -> dict:
        """the dealer's whole box from host buffers in one call: X, Y, a1, a2, digest, challenge, responses"""
        call, outputs = self.deal_call(coeffs, positions, pubkeys, witnesses)
        call()
        return outputs()

    def dleq_responses_device(self, w_dev_ptr: int, alpha_dev_ptr: int, c: bytes, n: int, out_dev_ptr: int) -> None:
        """r[i] = w[i] - alpha[i] c mod (q-1), everything but the shared c in HBM"""
        kc, pc = _buf(c)
        self._check(self.lib.mpvss_modp_dleq_responses_device(self.ctx, w_dev_ptr, alpha_dev_ptr, pc, n, out_dev_ptr),
                    "dleq_responses_device")

    def pipeline_stats(self, reset: bool = False) -> dict:
        st = PipelineStats()
        self._check(self.lib.mpvss_pipeline_stats_get(self.ctx, C.byref(st), int(reset)), "pipeline_stats_get")
        return {"enqueue_ms": st.enqueue_ms, "wait_ms": st.wait_ms, "hash_ms": st.hash_ms,
                "kernel_ms": list(st.kernel_ms), "kernel_launches": list(st.kernel_launches), "blocks": int(st.blocks)}

    def blocks_in_flight(self) -> Tuple[int, int]:
        """(blocks enqueued and not yet fully absorbed, of those: GPU work still pending)"""
        a, b = C.c_int(0), C.c_int(0)
        self._check(self.lib.mpvss_blocks_in_flight(self.ctx, C.byref(a), C.byref(b)), "blocks_in_flight")
        return int(a.value), int(b.value)

    def fd_stats(self) -> Tuple[int, int]:
        """(blocks absorbed through the forward-difference path, of those: fell back to Horner on the device)"""
        b, f = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self.lib.mpvss_modp_fd_stats(self.ctx, C.byref(b), C.byref(f)), "fd_stats")
        return int(b.value), int(f.value)

    def verify_block_absorb(self, state: bytes) -> bytes:
        ks, ps = _buf(state)
        self._check(self.lib.mpvss_modp_verify_block_absorb(self.ctx, ps, None, None, None), "verify_block_absorb")
        return bytes(ks)

    def block_claim(self) -> int:
        """take the oldest MODP distribution block in flight; returns its ticket (blocks count in enqueue order)"""
        tk = C.c_ulonglong(0)
        self._check(self.lib.mpvss_block_claim(self.ctx, C.byref(tk)), "block_claim")
        return int(tk.value)

    def verify_block_absorb_claimed(self, ticket: int, state: bytes) -> bytes:
        ks, ps = _buf(state)
        self._check(self.lib.mpvss_modp_verify_block_absorb_claimed(self.ctx, ticket, ps, None, None, None),
                    "verify_block_absorb_claimed")
        return bytes(ks)

    def extract_shares_compute(self, pk: bytes, y: bytes, xinv: bytes, w: bytes) -> int:
        """enqueue one batch of extract_secret_share (returns its size); absorb with extract_shares_absorb(n)"""
        n = len(pk) // EB
        k = [_buf(x) for x in (pk, y, xinv, w)]
        self._check(self.lib.mpvss_modp_extract_shares_compute(self.ctx, k[0][1], k[1][1], k[2][1], k[3][1], n), "extract_shares_compute")
        return n

    def extract_shares_absorb(self, n: int) -> Tuple[bytes, bytes]:
        ks, ps = _out(n * EB); kc, pc = _out(n * EB)
        self._check(self.lib.mpvss_modp_extract_shares_absorb(self.ctx, ps, pc), "extract_shares_absorb")
        return bytes(ks)[: n * EB], bytes(kc)[: n * EB]

    def extract_shares(self, pk: bytes, y: bytes, xinv: bytes, w: bytes) -> Tuple[bytes, bytes]:
        n = len(pk) // EB
        k = [_buf(x) for x in (pk, y, xinv, w)]
        ks, ps = _out(n * EB); kc, pc = _out(n * EB)
        self._check(self.lib.mpvss_modp_extract_shares(self.ctx, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1], n, ps, pc),
                    "extract_shares")
        return bytes(ks)[: n * EB], bytes(kc)[: n * EB]

    def ec_extract_shares(self, group: int, pk: bytes, y: bytes, xinv: bytes, w: bytes) -> Tuple[bytes, bytes]:
        e = EC_ENC[group]
        n = len(xinv) // 32
        k = [_buf(x) for x in (pk, y, xinv, w)]
        ks, ps = _out(n * e); kc, pc = _out(n * 32)
        self._check(self.lib.mpvss_ec_extract_shares(self.ctx, group, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1], n,
                                                     ps, pc), "ec_extract_shares")
        return bytes(ks)[: n * e], bytes(kc)[: n * 32]

    # ---- elliptic-curve groups (group = GROUP_SECP256K1 | GROUP_RISTRETTO255)
    def ec_batch_exp(self, group: int, bases: bytes, scalars: bytes) -> bytes:
        n = len(scalars) // 32
        ka, pa = _buf(bases); kb, pb = _buf(scalars); ko, po = _out(n * EC_ENC[group])
        self._check(self.lib.mpvss_ec_batch_exp(self.ctx, group, MPVSS_HOST, pa, pb, n, po), "ec_batch_exp")
        return bytes(ko)[: n * EC_ENC[group]]

    def verify_wire(self, wire: bytes):
        """verify_distribution_shares of a serialized box (any group): (verdict, digest)"""
        buf = (C.c_uint64 * ((len(wire) + 7) // 8))()          # 8-byte aligned copy
        C.memmove(buf, wire, len(wire))
        v = C.c_int(0)
        kd, pd = _out(32)
        self._check(self.lib.mpvss_box_verify_wire(self.ctx, C.cast(buf, C.c_void_p), len(wire), C.byref(v), pd), "box_verify_wire")
        return bool(v.value), bytes(kd)[:32]

    def reconstruct(self, positions: Sequence[int], shares: bytes):
        """(G^s, mask) from m decrypted shares: secret = int(mask) ^ U (participant.rs:462-519)"""
        m = len(positions)
        pos = (C.c_int64 * max(m, 1))(*positions)
        ks, ps = _buf(shares); kg, pg = _out(EB); km, pm = _out(32)
        self._check(self.lib.mpvss_modp_reconstruct(self.ctx, MPVSS_HOST, C.cast(pos, C.c_void_p), ps, m, pg, pm), "reconstruct")
        return bytes(kg)[:EB], bytes(km)[:32]

    def ec_reconstruct(self, group: int, positions: Sequence[int], shares: bytes):
        m = len(positions)
        pos = (C.c_int64 * max(m, 1))(*positions)
        ks, ps = _buf(shares); kg, pg = _out(EC_ENC[group]); km, pm = _out(32)
        self._check(self.lib.mpvss_ec_reconstruct(self.ctx, group, MPVSS_HOST, C.cast(pos, C.c_void_p), ps, m, pg, pm),
                    "ec_reconstruct")
        return bytes(kg)[:EC_ENC[group]], bytes(km)[:32]

    def ec_batch_exp_generator(self, group: int, scalars: bytes) -> bytes:
        n = len(scalars) // 32
        kb, pb = _buf(scalars); ko, po = _out(n * EC_ENC[group])
        self._check(self.lib.mpvss_ec_batch_exp_generator(self.ctx, group, MPVSS_HOST, pb, n, po), "ec_batch_exp_generator")
        return bytes(ko)[: n * EC_ENC[group]]

    def ec_verify_block_compute(self, group: int, commitments: bytes, positions: Sequence[int], pubkeys: bytes,
                                shares: bytes, responses: bytes, challenge: bytes) -> None:
        e = EC_ENC[group]
        t, n = len(commitments) // e, len(positions)
        kc, pc = _buf(commitments); ky, py = _buf(pubkeys); kY, pY = _buf(shares); kr, pr = _buf(responses)
        kch, pch = _buf(challenge)
        pos = (C.c_int64 * max(n, 1))(*positions)
        self._check(self.lib.mpvss_ec_verify_block_compute(self.ctx, group, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p),
                                                           py, pY, pr, n, pch), "ec_verify_block_compute")

    def ec_verify_block_absorb(self, state: bytes) -> bytes:
        ks, ps = _buf(state)
        self._check(self.lib.mpvss_ec_verify_block_absorb(self.ctx, ps, None, None, None), "ec_verify_block_absorb")
        return bytes(ks)

    def ec_verify_many(self, group: int, boxes: Sequence[dict], depth: int = 6, hash_threads: int = 3):
        """boxes: dicts with commitments, positions, pubkeys, shares, responses, challenge (host bytes).
        Returns [(verdict, digest)] in box order."""
        keep, arr = [], (EcBox * max(len(boxes), 1))()
        e = EC_ENC[group]
        for i, b in enumerate(boxes):
            n = len(b["positions"])
            pos = (C.c_int64 * max(n, 1))(*b["positions"])
            bufs = [_buf(b[k]) for k in ("commitments", "pubkeys", "shares", "responses", "challenge")]
            keep.append((pos, bufs))
            arr[i] = EcBox(bufs[0][1], len(b["commitments"]) // e, C.cast(pos, C.c_void_p), bufs[1][1], bufs[2][1],
                           bufs[3][1], n, bufs[4][1])
        verdicts = (C.c_int * max(len(boxes), 1))()
        kd, pd = _out(32 * len(boxes))
        self._check(self.lib.mpvss_ec_verify_many(self.ctx, group, MPVSS_HOST, arr, len(boxes), depth, hash_threads, verdicts,
                                                  pd), "ec_verify_many")
        raw = bytes(kd)
        return [(bool(verdicts[i]), raw[32 * i:32 * i + 32]) for i in range(len(boxes))]

    def ec_batch_mul(self, group: int, a: bytes, b: bytes) -> bytes:
        n = len(a) // EC_ENC[group]
        ka, pa = _buf(a); kb, pb = _buf(b); ko, po = _out(n * EC_ENC[group])
        self._check(self.lib.mpvss_ec_batch_mul(self.ctx, group, MPVSS_HOST, pa, pb, n, po), "ec_batch_mul")
        return bytes(ko)[: n * EC_ENC[group]]

    def ec_commit_eval(self, group: int, commitments: bytes, positions: Sequence[int]) -> bytes:
        e = EC_ENC[group]
        t, n = len(commitments) // e, len(positions)
        kc, pc = _buf(commitments)
        pos = (C.c_int64 * max(n, 1))(*positions)
        ko, po = _out(n * e)
        self._check(self.lib.mpvss_ec_commit_eval(self.ctx, group, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), n, po),
                    "ec_commit_eval")
        return bytes(ko)[: n * e]

    def ec_dleq_commitments(self, group: int, g1: bytes, h1: bytes, g2: bytes, h2: bytes, r: bytes, c: bytes,
                            c_per_share: bool) -> Tuple[bytes, bytes]:
        e = EC_ENC[group]
        n = len(r) // 32
        k = [_buf(x) for x in (g1, h1, g2, h2, r, c)]
        k1, p1 = _out(n * e); k2, p2 = _out(n * e)
        self._check(self.lib.mpvss_ec_dleq_commitments(self.ctx, group, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1],
                                                       k[4][1], k[5][1], int(c_per_share), n, p1, p2),
                    "ec_dleq_commitments")
        return bytes(k1)[: n * e], bytes(k2)[: n * e]

    def ec_verify_distribution(self, group: int, commitments: bytes, positions: Sequence[int], pubkeys: bytes,
                               shares: bytes, responses: bytes, challenge: bytes, dump: bool = False):
        e = EC_ENC[group]
        t, n = len(commitments) // e, len(positions)
        kc, pc = _buf(commitments); ky, py = _buf(pubkeys); kY, pY = _buf(shares); kr, pr = _buf(responses)
        kch, pch = _buf(challenge)
        pos = (C.c_int64 * max(n, 1))(*positions)
        verdict = C.c_int(0)
        kd, pd = _out(32)
        kx, px = _out(n * e) if dump else (None, None)
        k1, p1 = _out(n * e) if dump else (None, None)
        k2, p2 = _out(n * e) if dump else (None, None)
        self._check(self.lib.mpvss_ec_verify_distribution(
            self.ctx, group, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), py, pY, pr, n, pch, C.byref(verdict), pd,
            px, p1, p2), "ec_verify_distribution")
        out = {"verdict": bool(verdict.value), "digest": bytes(kd)[:32]}
        if dump:
            out.update(X=bytes(kx)[: n * e], a1=bytes(k1)[: n * e], a2=bytes(k2)[: n * e])
        return out

    def ec_verify_shares(self, group: int, pk: bytes, s: bytes, y: bytes, c: bytes, r: bytes) -> bytes:
        n = len(r) // 32
        k = [_buf(x) for x in (pk, s, y, c, r)]
        kv, pv = _out(n)
        self._check(self.lib.mpvss_ec_verify_shares(self.ctx, group, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1],
                                                    k[4][1], n, pv), "ec_verify_shares")
        return bytes(kv)[:n]

    def ec_distribute_compute(self, group: int, commitments: Optional[bytes], positions: Optional[Sequence[int]],
                              pubkeys: bytes, p_values: bytes, witnesses: bytes) -> int:
        """Enqueue one dealer block of a curve group (returns its size).  commitments None: X_i = p_i * G."""
        e = EC_ENC[group]
        n = len(pubkeys) // e
        t = len(commitments) // e if commitments else 0
        kc, pc = _buf(commitments) if commitments else (None, None)
        ky, py = _buf(pubkeys); kp, pp = _buf(p_values); kw, pw = _buf(witnesses)
        pos = (C.c_int64 * max(n, 1))(*(positions or [0] * n))
        self._check(self.lib.mpvss_ec_distribute_compute(self.ctx, group, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), py, pp, pw,
                                                         n, None, None, None, None), "ec_distribute_compute")
        return n

    def ec_poly_eval_device(self, group: int, coeffs: bytes, positions_dev_ptr: int, n: int, out_dev_ptr: int) -> None:
        kc, pc = _buf(coeffs)
        self._check(self.lib.mpvss_ec_poly_eval_device(self.ctx, group, pc, len(coeffs) // 32, positions_dev_ptr, n, out_dev_ptr),
                    "ec_poly_eval_device")

    def ec_dleq_responses_device(self, group: int, w_dev_ptr: int, alpha_dev_ptr: int, c: bytes, n: int, out_dev_ptr: int) -> None:
        kc, pc = _buf(c)
        self._check(self.lib.mpvss_ec_dleq_responses_device(self.ctx, group, w_dev_ptr, alpha_dev_ptr, pc, n, out_dev_ptr),
                    "ec_dleq_responses_device")

    def ec_deal_compute(self, group: int, coeffs: bytes, positions_dev_ptr: int, pubkeys_dev_ptr: int, witnesses_dev_ptr: int, n: int,
                        p_dev_out_ptr: int) -> None:
        """one curve-group dealer's block, P(i) included, enqueued (absorb with ec_distribute_absorb)"""
        kc, pc = _buf(coeffs)
        self._check(self.lib.mpvss_ec_deal_compute(self.ctx, group, pc, len(coeffs) // 32, positions_dev_ptr, pubkeys_dev_ptr,
                                                   witnesses_dev_ptr, n, p_dev_out_ptr, None, None, None, None), "ec_deal_compute")

    def ec_deal_call(self, group: int, coeffs: bytes, positions: Sequence[int], pubkeys: bytes, witnesses: bytes):
        """(call, outputs) for mpvss_ec_deal over ctypes buffers made once (see deal_call)"""
        n = len(positions)
        L = 33 if group == GROUP_SECP256K1 else 32
        pos = (C.c_int64 * max(n, 1))(*positions)
        k = [_buf(b) for b in (coeffs, pubkeys, witnesses)]
        outs = [_out(n * L) for _ in range(4)]
        kr, pr = _out(n * 32)
        kd, pd = _out(32)
        kc, pc = _out(32)
        t = len(coeffs) // 32

        def call():
            self._check(self.lib.mpvss_ec_deal(self.ctx, group, k[0][1], t, pos, k[1][1], k[2][1], n,
                                               outs[0][1], outs[1][1], outs[2][1], outs[3][1], pd, pc, pr), "ec_deal")

        def outputs():
            X, Y, a1, a2 = (bytes(o[0])[: n * L] for o in outs)
            return {"X": X, "Y": Y, "a1": a1, "a2": a2, "digest": bytes(kd)[:32], "challenge": bytes(kc)[:32],
                    "responses": bytes(kr)[: n * 32]}
        return call, outputs

    def ec_deal(self, group: int, coeffs: bytes, positions: Sequence[int], pubkeys: bytes, witnesses: bytes) -> dict:
        """a curve group's whole box from host buffers in one call: X, Y, a1, a2, digest, challenge, responses"""
        call, outputs = self.ec_deal_call(group, coeffs, positions, pubkeys, witnesses)
        call()
        return outputs()

    def ec_distribute_absorb(self, group: int, state: bytes, n: int):
        """(state, X, Y, a1, a2) of the oldest dealer block"""
        e = EC_ENC[group]
        ks, ps = _buf(state)
        outs = [_out(n * e) for _ in range(4)]
        self._check(self.lib.mpvss_ec_distribute_absorb(self.ctx, ps, outs[0][1], outs[1][1], outs[2][1], outs[3][1]),
                    "ec_distribute_absorb")
        return (bytes(ks),) + tuple(bytes(o[0])[: n * e] for o in outs)

    def ec_verify_shares_compute(self, group: int, pk: bytes, s: bytes, y: bytes, c: bytes, r: bytes,
                                 verdicts_dev_ptr: int = 0) -> int:
        """Enqueue one batch of share-box proofs of a curve group (returns its size)"""
        n = len(r) // 32
        k = [_buf(x) for x in (pk, s, y, c, r)]
        self._check(self.lib.mpvss_ec_verify_shares_compute(self.ctx, group, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1],
                                                            k[4][1], n, C.c_void_p(verdicts_dev_ptr or None)),
                    "ec_verify_shares_compute")
        return n

    def ec_verify_shares_absorb(self, n: int) -> bytes:
        kv, pv = _out(n)
        self._check(self.lib.mpvss_ec_verify_shares_absorb(self.ctx, pv), "ec_verify_shares_absorb")
        return bytes(kv)[:n]

    def ec_distribute(self, group: int, commitments: bytes, positions: Sequence[int], pubkeys: bytes, p_values: bytes,
                      witnesses: bytes):
        e = EC_ENC[group]
        t, n = len(commitments) // e, len(positions)
        kc, pc = _buf(commitments); ky, py = _buf(pubkeys); kp, pp = _buf(p_values); kw, pw = _buf(witnesses)
        pos = (C.c_int64 * max(n, 1))(*positions)
        outs = [_out(n * e) for _ in range(4)]
        kd, pd = _out(32)
        self._check(self.lib.mpvss_ec_distribute(self.ctx, group, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), py, pp, pw,
                                                 n, outs[0][1], outs[1][1], outs[2][1], outs[3][1], pd), "ec_distribute")
        X, Y, a1, a2 = (bytes(o[0])[: n * e] for o in outs)
        return {"X": X, "Y": Y, "a1": a1, "a2": a2, "digest": bytes(kd)[:32]}

    def verify_shares(self, pk: bytes, s: bytes, y: bytes, c: bytes, r: bytes) -> bytes:
        n = len(pk) // EB
        k = [_buf(x) for x in (pk, s, y, c, r)]
        kv, pv = _out(n)
        self._check(self.lib.mpvss_modp_verify_shares(self.ctx, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1],
                                                      k[4][1], n, pv), "verify_shares")
        return bytes(kv)[:n]

    def distribute_compute(self, commitments: Optional[bytes], positions: Optional[Sequence[int]], pubkeys: bytes,
                           p_values: bytes, witnesses: bytes) -> int:
        """Enqueue one dealer block (returns its size).  commitments None: X_i = g^p_i (the dealer knows the polynomial)."""
        n = len(pubkeys) // EB
        t = len(commitments) // EB if commitments else 0
        kc, pc = _buf(commitments) if commitments else (None, None)
        ky, py = _buf(pubkeys); kp, pp = _buf(p_values); kw, pw = _buf(witnesses)
        pos = (C.c_int64 * max(n, 1))(*(positions or [0] * n))
        self._check(self.lib.mpvss_modp_distribute_compute(self.ctx, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), py, pp, pw, n,
                                                           None, None, None, None), "distribute_compute")
        return n

    def distribute_absorb(self, state: bytes, n: int):
        """(state, X, Y, a1, a2) of the oldest dealer block"""
        ks, ps = _buf(state)
        outs = [_out(n * EB) for _ in range(4)]
        self._check(self.lib.mpvss_modp_distribute_absorb(self.ctx, ps, outs[0][1], outs[1][1], outs[2][1], outs[3][1]),
                    "distribute_absorb")
        return (bytes(ks),) + tuple(bytes(o[0])[: n * EB] for o in outs)

    def verify_shares_compute(self, pk: bytes, s: bytes, y: bytes, c: bytes, r: bytes, verdicts_dev_ptr: int = 0) -> int:
        """Enqueue one batch of share-box proofs (returns its size); verdicts_dev_ptr: optional device address that
        receives the n verdict bytes in stream order (e.g. a torch uint8 tensor's data_ptr())."""
        n = len(pk) // EB
        k = [_buf(x) for x in (pk, s, y, c, r)]
        self._check(self.lib.mpvss_modp_verify_shares_compute(self.ctx, MPVSS_HOST, k[0][1], k[1][1], k[2][1], k[3][1],
                                                              k[4][1], n, C.c_void_p(verdicts_dev_ptr or None)),
                    "verify_shares_compute")
        return n

    def verify_shares_absorb(self, n: int) -> bytes:
        kv, pv = _out(n)
        self._check(self.lib.mpvss_modp_verify_shares_absorb(self.ctx, pv), "verify_shares_absorb")
        return bytes(kv)[:n]

    def distribute(self, commitments: bytes, positions: Sequence[int], pubkeys: bytes, p_values: bytes,
                   witnesses: bytes):
        t = len(commitments) // EB
        n = len(positions)
        kc, pc = _buf(commitments); ky, py = _buf(pubkeys); kp, pp = _buf(p_values); kw, pw = _buf(witnesses)
        pos = (C.c_int64 * max(n, 1))(*positions)
        outs = [_out(n * EB) for _ in range(4)]
        kd, pd = _out(32)
        self._check(self.lib.mpvss_modp_distribute(self.ctx, MPVSS_HOST, pc, t, C.cast(pos, C.c_void_p), py, pp, pw,
                                                   n, outs[0][1], outs[1][1], outs[2][1], outs[3][1], pd),
                    "distribute")
        X, Y, a1, a2 = (bytes(o[0])[: n * EB] for o in outs)
        return {"X": X, "Y": Y, "a1": a1, "a2": a2, "digest": bytes(kd)[:32]}


def sha256(data: bytes) -> bytes:
    lib = load_library()
    kd, pd = _buf(data if data else b"\0")
    ko, po = _out(32)
    lib.mpvss_sha256(pd, len(data), po)
    return bytes(ko)[:32]


def transcript_init() -> bytes:
    lib = load_library()
    ks, ps = _out(TRANSCRIPT_STATE_BYTES)
    lib.mpvss_transcript_init(ps)
    return bytes(ks)[:TRANSCRIPT_STATE_BYTES]


def transcript_verdict(state: bytes, challenge: bytes):
    lib = load_library()
    ks, ps = _buf(state); kc, pc = _buf(challenge); kd, pd = _out(32)
    v = C.c_int(0)
    rc = lib.mpvss_modp_transcript_verdict(ps, pc, C.byref(v), pd)
    if rc != 0:
        raise EngineError(f"transcript_verdict failed: {rc}")
    return bool(v.value), bytes(kd)[:32]


def transcript_absorb(state: bytes, elements: bytes) -> bytes:
    lib = load_library()
    ks, ps = _buf(state); ke, pe = _buf(elements if elements else b"\0")
    rc = lib.mpvss_modp_transcript_absorb(ps, pe, len(elements) // EB)
    if rc != 0:
        raise EngineError(f"transcript_absorb failed: {rc}")
    return bytes(ks)


def ec_hash_to_scalar(group: int, data: bytes) -> bytes:
    lib = load_library()
    kd, pd = _buf(data if data else b"\0")
    ko, po = _out(32)
    rc = lib.mpvss_ec_hash_to_scalar(group, pd, len(data), po)
    if rc != 0:
        raise EngineError(f"ec_hash_to_scalar failed: {rc}")
    return bytes(ko)[:32]


def ec_transcript_verdict(group: int, state: bytes, challenge: bytes):
    lib = load_library()
    ks, ps = _buf(state); kc, pc = _buf(challenge); kd, pd = _out(32)
    v = C.c_int(0)
    rc = lib.mpvss_ec_transcript_verdict(group, ps, pc, C.byref(v), pd)
    if rc != 0:
        raise EngineError(f"ec_transcript_verdict failed: {rc}")
    return bool(v.value), bytes(kd)[:32]


def ec_transcript_absorb(group: int, state: bytes, elements: bytes) -> bytes:
    lib = load_library()
    ks, ps = _buf(state); ke, pe = _buf(elements if elements else b"\0")
    rc = lib.mpvss_ec_transcript_absorb(group, ps, pe, len(elements) // EC_ENC[group])
    if rc != 0:
        raise EngineError(f"ec_transcript_absorb failed: {rc}")
    return bytes(ks)


# ---- scalar-field side (host only) -------------------------------------------------------------------------------
def _scalar_width(group: int) -> int:
    return EB if group == 0 else 32


def scalar_mul(group: int, a: bytes, b: bytes) -> bytes:
    """Group::scalar_mul; group 0 = MODP-2048 (256-byte big-endian), else GROUP_*"""
    lib = load_library()
    ka, pa = _buf(a); kb, pb = _buf(b); ko, po = _out(_scalar_width(group))
    rc = lib.mpvss_modp_scalar_mul(pa, pb, po) if group == 0 else lib.mpvss_ec_scalar_mul(group, pa, pb, po)
    if rc != 0:
        raise EngineError(f"scalar_mul failed: {rc}")
    return bytes(ko)[:_scalar_width(group)]


def scalar_sub(group: int, a: bytes, b: bytes) -> bytes:
    lib = load_library()
    ka, pa = _buf(a); kb, pb = _buf(b); ko, po = _out(_scalar_width(group))
    rc = lib.mpvss_modp_scalar_sub(pa, pb, po) if group == 0 else lib.mpvss_ec_scalar_sub(group, pa, pb, po)
    if rc != 0:
        raise EngineError(f"scalar_sub failed: {rc}")
    return bytes(ko)[:_scalar_width(group)]


def dleq_responses(group: int, w: bytes, alpha: bytes, c: bytes, threads: int = 0) -> bytes:
    """r_i = w_i - alpha_i * c_i mod order; c is one scalar or one per proof"""
    lib = load_library()
    sw = _scalar_width(group)
    n = len(w) // sw
    kw, pw = _buf(w); ka, pa = _buf(alpha); kc, pc = _buf(c); ko, po = _out(n * sw)
    per = int(len(c) != sw)
    rc = (lib.mpvss_modp_dleq_responses(pw, pa, pc, per, n, po, threads) if group == 0
          else lib.mpvss_ec_dleq_responses(group, pw, pa, pc, per, n, po, threads))
    if rc != 0:
        raise EngineError(f"dleq_responses failed: {rc}")
    return bytes(ko)[: n * sw]


def poly_eval(group: int, coeffs: bytes, positions: Sequence[int], threads: int = 0) -> bytes:
    """P(i) mod order for every position (polynomial.rs:50-58 + the caller's % order)"""
    lib = load_library()
    sw = _scalar_width(group)
    t, n = len(coeffs) // sw, len(positions)
    kc, pc = _buf(coeffs); ko, po = _out(n * sw)
    pos = (C.c_int64 * max(n, 1))(*positions)
    rc = (lib.mpvss_modp_poly_eval(pc, t, C.cast(pos, C.c_void_p), n, po, threads) if group == 0
          else lib.mpvss_ec_poly_eval(group, pc, t, C.cast(pos, C.c_void_p), n, po, threads))
    if rc != 0:
        raise EngineError(f"poly_eval failed: {rc}")
    return bytes(ko)[: n * sw]


# ---- flat wire format ("MPVSSBX1") ---------------------------------------------------------------------------------
def box_serialize(group: int, commitments: bytes, positions: Sequence[int], pubkeys: bytes, shares: bytes, responses: bytes,
                  challenge: bytes, u_be: bytes = b"") -> bytes:
    lib = load_library()
    e = EB if group == 0 else EC_ENC[group]
    t, n = len(commitments) // e, len(positions)
    size = lib.mpvss_box_wire_size(group, n, t, len(u_be))
    if size == 0:
        raise EngineError("box_serialize: bad dimensions")
    k = [_buf(x if x else b"\0") for x in (commitments, pubkeys, shares, responses, challenge, u_be)]
    pos = (C.c_int64 * max(n, 1))(*positions)
    ko, po = _out(size)
    out_len = C.c_size_t(0)
    rc = lib.mpvss_box_serialize(group, k[0][1], t, C.cast(pos, C.c_void_p), k[1][1], k[2][1], k[3][1], n, k[4][1], k[5][1],
                                 len(u_be), po, size, C.byref(out_len))
    if rc != 0:
        raise EngineError(f"box_serialize failed: {rc}")
    return bytes(ko)[: out_len.value]


def box_parse(wire: bytes) -> dict:
    lib = load_library()
    buf = (C.c_uint64 * ((len(wire) + 7) // 8))()
    C.memmove(buf, wire, len(wire))
    view = BoxView()
    rc = lib.mpvss_box_parse(C.cast(buf, C.c_void_p), len(wire), C.byref(view))
    if rc != 0:
        raise EngineError(f"box_parse failed: {rc}")
    e, s_, n, t = view.element_bytes, view.scalar_bytes, view.n, view.t
    at = lambda p, ln: C.string_at(p, ln) if ln else b""
    return {"group": view.group, "n": n, "t": t, "commitments": at(view.commitments, t * e),
            "positions": list((C.c_int64 * n).from_address(view.positions)) if n else [],
            "pubkeys": at(view.pubkeys, n * e), "shares": at(view.shares, n * e), "responses": at(view.responses, n * s_),
            "challenge": at(view.challenge, s_), "U": at(view.u_be, view.u_len)}
