//! `extern "C"` declarations of include/mpvss_hip.h -- one item per exported symbol, same order as the header.
//! Uncompiled in this repository's environment (no Rust toolchain); the symbol list is checked against the header by
//! tests/test_capi_host.py.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_ulonglong, c_void};

pub const MPVSS_OK: c_int = 0;
pub const MPVSS_E_INVALID: c_int = -1;
pub const MPVSS_E_DEVICE: c_int = -2;
pub const MPVSS_E_NOMEM: c_int = -3;
pub const MPVSS_E_UNSUPPORTED: c_int = -4;
pub const MPVSS_HOST: c_int = 0;
pub const MPVSS_DEVICE: c_int = 1;
pub const MPVSS_MODP_BYTES: usize = 256;
pub const MPVSS_TRANSCRIPT_STATE_BYTES: usize = 128;
pub const MPVSS_BLOCK_SLOTS: usize = 64;
pub const MPVSS_GROUP_SECP256K1: c_int = 1;
pub const MPVSS_GROUP_RISTRETTO255: c_int = 2;

#[repr(C)]
pub struct mpvss_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mpvss_keyset {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mpvss_modp_box {
    pub commitments: *const u8,
    pub t: usize,
    pub positions: *const i64,
    pub pubkeys: *const u8,
    pub shares: *const u8,
    pub responses: *const u8,
    pub n: usize,
    pub challenge_host: *const u8,
    pub keyset: *const mpvss_keyset,
    pub key_offset: usize,
}
#[repr(C)]
pub struct mpvss_ec_box {
    pub commitments: *const u8,
    pub t: usize,
    pub positions: *const i64,
    pub pubkeys: *const u8,
    pub shares: *const u8,
    pub responses: *const u8,
    pub n: usize,
    pub challenge_host: *const u8,
}
#[repr(C)]
pub struct mpvss_pipeline_stats {
    pub enqueue_ms: c_double,
    pub wait_ms: c_double,
    pub hash_ms: c_double,
    pub kernel_ms: [c_double; 4],
    pub kernel_launches: [c_ulonglong; 4],
    pub blocks: c_ulonglong,
}
#[repr(C)]
pub struct mpvss_box_view {
    pub group: c_int,
    pub element_bytes: usize,
    pub scalar_bytes: usize,
    pub n: usize,
    pub t: usize,
    pub u_len: usize,
    pub commitments: *const u8,
    pub positions: *const i64,
    pub pubkeys: *const u8,
    pub shares: *const u8,
    pub responses: *const u8,
    pub challenge: *const u8,
    pub u_be: *const u8,
}

/// `state_in` / `state_out` of mpvss_modp_verify_many_chained (None = NULL)
pub type mpvss_chain_cb = Option<unsafe extern "C" fn(user: *mut c_void, box_index: usize, state: *mut u8, ok: c_int) -> c_int>;

#[link(name = "mpvss_hip")]
unsafe extern "C" {
    // ---- context
    pub fn mpvss_process_init() -> c_int;
    pub fn mpvss_device_count() -> c_int;
    pub fn mpvss_ctx_create(device_id: c_int, out: *mut *mut mpvss_ctx) -> c_int;
    pub fn mpvss_ctx_destroy(ctx: *mut mpvss_ctx);
    pub fn mpvss_last_error(ctx: *const mpvss_ctx) -> *const c_char;
    pub fn mpvss_ctx_set_stream(ctx: *mut mpvss_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn mpvss_ctx_synchronize(ctx: *mut mpvss_ctx) -> c_int;
    // ---- MODP group operations, batched
    pub fn mpvss_modp_batch_exp(ctx: *mut mpvss_ctx, space: c_int, bases: *const u8, exps: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn mpvss_modp_batch_mul(ctx: *mut mpvss_ctx, space: c_int, a: *const u8, b: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn mpvss_modp_batch_exp_fixed_base(ctx: *mut mpvss_ctx, space: c_int, base_host: *const u8, exps: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn mpvss_modp_commit_eval(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64, n: usize, x_out: *mut u8) -> c_int;
    pub fn mpvss_modp_dleq_commitments(ctx: *mut mpvss_ctx, space: c_int, g1_host: *const u8, h1: *const u8, g2: *const u8, h2: *const u8,
                                       r: *const u8, c: *const u8, c_per_share: c_int, n: usize, a1_out: *mut u8, a2_out: *mut u8) -> c_int;
    // ---- verify_distribution_shares
    pub fn mpvss_modp_verify_distribution(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                          pubkeys: *const u8, shares: *const u8, responses: *const u8, n: usize, challenge_host: *const u8,
                                          verdict: *mut c_int, digest32_out: *mut u8, x_out_host: *mut u8, a1_out_host: *mut u8,
                                          a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_transcript_init(state: *mut u8);
    pub fn mpvss_modp_verify_block_compute(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                           pubkeys: *const u8, shares: *const u8, responses: *const u8, n: usize, challenge_host: *const u8) -> c_int;
    pub fn mpvss_modp_verify_block_absorb(ctx: *mut mpvss_ctx, state: *mut u8, x_out_host: *mut u8, a1_out_host: *mut u8, a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_modp_verify_block_compute_flags(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                                 pubkeys: *const u8, shares: *const u8, responses: *const u8, n: usize, challenge_host: *const u8,
                                                 wellformed_dev_out: *mut u8) -> c_int;
    pub fn mpvss_block_claim(ctx: *mut mpvss_ctx, ticket_out: *mut c_ulonglong) -> c_int;
    pub fn mpvss_modp_verify_block_absorb_claimed(ctx: *mut mpvss_ctx, ticket: c_ulonglong, state: *mut u8,
        x_out_host: *mut u8, a1_out_host: *mut u8, a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_modp_transcript_absorb(state: *mut u8, elements: *const u8, count: usize) -> c_int;
    pub fn mpvss_modp_transcript_verdict(state: *const u8, challenge_host: *const u8, verdict: *mut c_int, digest32_out: *mut u8) -> c_int;
    pub fn mpvss_modp_verify_many(ctx: *mut mpvss_ctx, space: c_int, boxes: *const mpvss_modp_box, count: usize, depth: c_int, hash_threads: c_int,
                                  verdicts: *mut c_int, digests32: *mut u8) -> c_int;
    pub fn mpvss_modp_verify_many_chained(ctx: *mut mpvss_ctx, space: c_int, boxes: *const mpvss_modp_box, count: usize, depth: c_int,
                                          hash_threads: c_int, wellformed_dev_out: *const *mut u8, state_in: mpvss_chain_cb,
                                          state_out: mpvss_chain_cb, user: *mut c_void, verdicts: *mut c_int, digests32: *mut u8) -> c_int;
    // ---- registered public keys
    pub fn mpvss_modp_keyset_create(ctx: *mut mpvss_ctx, space: c_int, pubkeys: *const u8, n: usize, out: *mut *mut mpvss_keyset) -> c_int;
    pub fn mpvss_ctx_set_key_cache(ctx: *mut mpvss_ctx, min_boxes: c_int) -> c_int;
    pub fn mpvss_ctx_set_key_cache_lru(ctx: *mut mpvss_ctx, max_sets: c_int, min_sightings: c_int) -> c_int;
    pub fn mpvss_modp_keyset_destroy(ctx: *mut mpvss_ctx, keyset: *mut mpvss_keyset);
    pub fn mpvss_modp_keyset_bytes(keyset: *const mpvss_keyset) -> usize;
    pub fn mpvss_modp_verify_block_compute_keyset(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                                  keyset: *const mpvss_keyset, key_offset: usize, shares: *const u8, responses: *const u8,
                                                  n: usize, challenge_host: *const u8) -> c_int;
    // ---- verify_share, batched
    pub fn mpvss_modp_verify_shares(ctx: *mut mpvss_ctx, space: c_int, pk: *const u8, s: *const u8, y: *const u8, c: *const u8, r: *const u8,
                                    n: usize, verdicts_host: *mut u8) -> c_int;
    pub fn mpvss_modp_verify_shares_compute(ctx: *mut mpvss_ctx, space: c_int, pk: *const u8, s: *const u8, y: *const u8, c: *const u8,
                                            r: *const u8, n: usize, verdicts_dev_out: *mut u8) -> c_int;
    pub fn mpvss_modp_verify_shares_absorb(ctx: *mut mpvss_ctx, verdicts_host: *mut u8) -> c_int;
    // ---- distribute_secret, group part
    pub fn mpvss_modp_distribute(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64, pubkeys: *const u8,
                                 p_values: *const u8, witnesses: *const u8, n: usize, x_out: *mut u8, y_out: *mut u8, a1_out: *mut u8,
                                 a2_out: *mut u8, digest32_out: *mut u8) -> c_int;
    pub fn mpvss_modp_distribute_compute(ctx: *mut mpvss_ctx, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                         pubkeys: *const u8, p_values: *const u8, witnesses: *const u8, n: usize, x_dev_out: *mut u8,
                                         y_dev_out: *mut u8, a1_dev_out: *mut u8, a2_dev_out: *mut u8) -> c_int;
    pub fn mpvss_modp_distribute_absorb(ctx: *mut mpvss_ctx, state: *mut u8, x_out_host: *mut u8, y_out_host: *mut u8, a1_out_host: *mut u8,
                                        a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_modp_deal(ctx: *mut mpvss_ctx, coeffs_host: *const u8, t: usize, positions_host: *const i64, pubkeys_host: *const u8,
                           witnesses_host: *const u8, n: usize, x_out: *mut u8, y_out: *mut u8, a1_out: *mut u8, a2_out: *mut u8,
                           digest32_out: *mut u8, challenge_out256: *mut u8, r_out: *mut u8) -> c_int;
    pub fn mpvss_modp_deal_compute(ctx: *mut mpvss_ctx, coeffs_host: *const u8, t: usize, positions_dev: *const i64, pubkeys_dev: *const u8,
                                   witnesses_dev: *const u8, n: usize, p_dev_out: *mut u8, x_dev_out: *mut u8, y_dev_out: *mut u8,
                                   a1_dev_out: *mut u8, a2_dev_out: *mut u8) -> c_int;
    pub fn mpvss_modp_deal_compute_keyset(ctx: *mut mpvss_ctx, coeffs_host: *const u8, t: usize, positions_dev: *const i64,
                                          keyset: *const mpvss_keyset, key_offset: usize, witnesses_dev: *const u8, n: usize,
                                          p_dev_out: *mut u8, x_dev_out: *mut u8, y_dev_out: *mut u8, a1_dev_out: *mut u8,
                                          a2_dev_out: *mut u8) -> c_int;
    // ---- curve groups
    pub fn mpvss_ec_batch_exp(ctx: *mut mpvss_ctx, group: c_int, space: c_int, bases: *const u8, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn mpvss_ec_batch_mul(ctx: *mut mpvss_ctx, group: c_int, space: c_int, a: *const u8, b: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn mpvss_ec_commit_eval(ctx: *mut mpvss_ctx, group: c_int, space: c_int, commitments: *const u8, t: usize, positions: *const i64, n: usize,
                                x_out: *mut u8) -> c_int;
    pub fn mpvss_ec_dleq_commitments(ctx: *mut mpvss_ctx, group: c_int, space: c_int, g1_host: *const u8, h1: *const u8, g2: *const u8,
                                     h2: *const u8, r: *const u8, c: *const u8, c_per_share: c_int, n: usize, a1_out: *mut u8, a2_out: *mut u8) -> c_int;
    pub fn mpvss_ec_batch_exp_generator(ctx: *mut mpvss_ctx, group: c_int, space: c_int, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn mpvss_ec_verify_distribution(ctx: *mut mpvss_ctx, group: c_int, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                        pubkeys: *const u8, shares: *const u8, responses: *const u8, n: usize, challenge_host: *const u8,
                                        verdict: *mut c_int, digest32_out: *mut u8, x_out_host: *mut u8, a1_out_host: *mut u8,
                                        a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_ec_verify_block_compute(ctx: *mut mpvss_ctx, group: c_int, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                                         pubkeys: *const u8, shares: *const u8, responses: *const u8, n: usize, challenge_host: *const u8) -> c_int;
    pub fn mpvss_ec_verify_block_absorb(ctx: *mut mpvss_ctx, state: *mut u8, x_out_host: *mut u8, a1_out_host: *mut u8, a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_ec_transcript_absorb(group: c_int, state: *mut u8, elements: *const u8, count: usize) -> c_int;
    pub fn mpvss_ec_transcript_verdict(group: c_int, state: *const u8, challenge_host: *const u8, verdict: *mut c_int, digest32_out: *mut u8) -> c_int;
    pub fn mpvss_ec_verify_many(ctx: *mut mpvss_ctx, group: c_int, space: c_int, boxes: *const mpvss_ec_box, count: usize, depth: c_int,
                                hash_threads: c_int, verdicts: *mut c_int, digests32: *mut u8) -> c_int;
    pub fn mpvss_ec_verify_shares(ctx: *mut mpvss_ctx, group: c_int, space: c_int, pk: *const u8, s: *const u8, y: *const u8, c: *const u8,
                                  r: *const u8, n: usize, verdicts_host: *mut u8) -> c_int;
    pub fn mpvss_ec_verify_shares_compute(ctx: *mut mpvss_ctx, group: c_int, space: c_int, pk: *const u8, s: *const u8,
        y: *const u8, c: *const u8, r: *const u8, n: usize, verdicts_dev_out: *mut u8) -> c_int;
    pub fn mpvss_ec_verify_shares_absorb(ctx: *mut mpvss_ctx, verdicts_host: *mut u8) -> c_int;
    pub fn mpvss_ec_distribute(ctx: *mut mpvss_ctx, group: c_int, space: c_int, commitments: *const u8, t: usize, positions: *const i64,
                               pubkeys: *const u8, p_values: *const u8, witnesses: *const u8, n: usize, x_out: *mut u8, y_out: *mut u8,
                               a1_out: *mut u8, a2_out: *mut u8, digest32_out: *mut u8) -> c_int;
    pub fn mpvss_ec_distribute_compute(ctx: *mut mpvss_ctx, group: c_int, space: c_int, commitments: *const u8, t: usize,
        positions: *const i64, pubkeys: *const u8, p_values: *const u8, witnesses: *const u8, n: usize,
        x_dev_out: *mut u8, y_dev_out: *mut u8, a1_dev_out: *mut u8, a2_dev_out: *mut u8) -> c_int;
    pub fn mpvss_ec_distribute_absorb(ctx: *mut mpvss_ctx, state: *mut u8, x_out_host: *mut u8, y_out_host: *mut u8,
        a1_out_host: *mut u8, a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_ec_block_absorb_claimed(ctx: *mut mpvss_ctx, ticket: c_ulonglong, state: *mut u8, x_out_host: *mut u8,
                                         y_out_host: *mut u8, a1_out_host: *mut u8, a2_out_host: *mut u8) -> c_int;
    pub fn mpvss_ec_poly_eval_device(ctx: *mut mpvss_ctx, group: c_int, coeffs_host: *const u8, t: usize, positions_dev: *const i64,
                                     n: usize, out_dev: *mut u8) -> c_int;
    pub fn mpvss_ec_dleq_responses_device(ctx: *mut mpvss_ctx, group: c_int, w_dev: *const u8, alpha_dev: *const u8,
                                          c_host32: *const u8, n: usize, r_dev_out: *mut u8) -> c_int;
    pub fn mpvss_ec_deal_compute(ctx: *mut mpvss_ctx, group: c_int, coeffs_host: *const u8, t: usize, positions_dev: *const i64,
                                 pubkeys_dev: *const u8, witnesses_dev: *const u8, n: usize, p_dev_out: *mut u8,
                                 x_dev_out: *mut u8, y_dev_out: *mut u8, a1_dev_out: *mut u8, a2_dev_out: *mut u8) -> c_int;
    pub fn mpvss_ec_deal(ctx: *mut mpvss_ctx, group: c_int, coeffs_host: *const u8, t: usize, positions_host: *const i64,
                         pubkeys_host: *const u8, witnesses_host: *const u8, n: usize, x_out: *mut u8, y_out: *mut u8,
                         a1_out: *mut u8, a2_out: *mut u8, digest32_out: *mut u8, challenge_out32: *mut u8, r_out: *mut u8) -> c_int;
    pub fn mpvss_ec_hash_to_scalar(group: c_int, data: *const u8, len: usize, out32: *mut u8) -> c_int;
    // ---- extract_secret_share, batched
    pub fn mpvss_modp_extract_shares(ctx: *mut mpvss_ctx, space: c_int, pk: *const u8, y: *const u8, xinv: *const u8, w: *const u8, n: usize,
                                     s_out: *mut u8, c_out_host: *mut u8) -> c_int;
    pub fn mpvss_ec_extract_shares(ctx: *mut mpvss_ctx, group: c_int, space: c_int, pk: *const u8, y: *const u8, xinv: *const u8, w: *const u8,
                                   n: usize, s_out: *mut u8, c_out_host: *mut u8) -> c_int;
    pub fn mpvss_modp_extract_shares_compute(ctx: *mut mpvss_ctx, pk: *const u8, y: *const u8, xinv: *const u8, w: *const u8, n: usize) -> c_int;
    pub fn mpvss_modp_extract_shares_absorb(ctx: *mut mpvss_ctx, s_out_host: *mut u8, c_out_host: *mut u8) -> c_int;
    // ---- scalar-field side (host only)
    pub fn mpvss_modp_scalar_mul(a256: *const u8, b256: *const u8, out256: *mut u8) -> c_int;
    pub fn mpvss_ec_scalar_mul(group: c_int, a32: *const u8, b32: *const u8, out32: *mut u8) -> c_int;
    pub fn mpvss_modp_scalar_sub(a256: *const u8, b256: *const u8, out256: *mut u8) -> c_int;
    pub fn mpvss_ec_scalar_sub(group: c_int, a32: *const u8, b32: *const u8, out32: *mut u8) -> c_int;
    pub fn mpvss_modp_dleq_responses(w: *const u8, alpha: *const u8, c: *const u8, c_per_share: c_int, n: usize, r_out: *mut u8, threads: c_int) -> c_int;
    pub fn mpvss_ec_dleq_responses(group: c_int, w: *const u8, alpha: *const u8, c: *const u8, c_per_share: c_int, n: usize, r_out: *mut u8,
                                   threads: c_int) -> c_int;
    pub fn mpvss_modp_poly_eval(coeffs: *const u8, t: usize, positions: *const i64, n: usize, out: *mut u8, threads: c_int) -> c_int;
    pub fn mpvss_ec_poly_eval(group: c_int, coeffs: *const u8, t: usize, positions: *const i64, n: usize, out: *mut u8, threads: c_int) -> c_int;
    pub fn mpvss_modp_poly_eval_device(ctx: *mut mpvss_ctx, coeffs_host: *const u8, t: usize, positions_dev: *const i64, n: usize, out_dev: *mut u8) -> c_int;
    pub fn mpvss_modp_dleq_responses_device(ctx: *mut mpvss_ctx, w_dev: *const u8, alpha_dev: *const u8, c_host: *const u8, n: usize, r_dev_out: *mut u8) -> c_int;
    // ---- reconstruct
    pub fn mpvss_modp_reconstruct(ctx: *mut mpvss_ctx, space: c_int, positions_host: *const i64, shares: *const u8, m: usize, gs_out256: *mut u8,
                                  mask_out32: *mut u8) -> c_int;
    pub fn mpvss_ec_reconstruct(ctx: *mut mpvss_ctx, group: c_int, space: c_int, positions_host: *const i64, shares: *const u8, m: usize,
                                gs_out: *mut u8, mask_out32: *mut u8) -> c_int;
    // ---- wire format
    pub fn mpvss_box_wire_size(group: c_int, n: usize, t: usize, u_len: usize) -> usize;
    pub fn mpvss_box_serialize(group: c_int, commitments: *const u8, t: usize, positions: *const i64, pubkeys: *const u8, shares: *const u8,
                               responses: *const u8, n: usize, challenge: *const u8, u_be: *const u8, u_len: usize, out: *mut u8,
                               out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn mpvss_box_parse(buf: *const u8, len: usize, view: *mut mpvss_box_view) -> c_int;
    pub fn mpvss_box_verify_wire(ctx: *mut mpvss_ctx, buf: *const u8, len: usize, verdict: *mut c_int, digest32_out: *mut u8) -> c_int;
    // ---- hashing helpers
    pub fn mpvss_sha256(data: *const u8, len: usize, out32: *mut u8);
    pub fn mpvss_modp_hash_to_scalar(data: *const u8, len: usize, out256: *mut u8);
    // ---- timing hooks / accounting
    pub fn mpvss_last_kernel_ms(ctx: *const mpvss_ctx, kernel_id: c_int) -> c_double;
    pub fn mpvss_last_kernel_launches(ctx: *const mpvss_ctx, kernel_id: c_int) -> c_int;
    pub fn mpvss_modp_fd_stats(ctx: *mut mpvss_ctx, blocks: *mut c_ulonglong, fallbacks: *mut c_ulonglong) -> c_int;
    pub fn mpvss_pipeline_stats_get(ctx: *mut mpvss_ctx, out: *mut mpvss_pipeline_stats, reset: c_int) -> c_int;
    pub fn mpvss_blocks_in_flight(ctx: *mut mpvss_ctx, in_flight_out: *mut c_int, gpu_pending_out: *mut c_int) -> c_int;
    pub fn mpvss_sha256_uses_shani() -> c_int;
    pub fn mpvss_issue_probe(ctx: *mut mpvss_ctx, kind: c_int, target_ms: c_double, insts_per_s_out: *mut c_double,
                             shader_clock_ghz_out: *mut c_double, ms_out: *mut c_double) -> c_int;
}
