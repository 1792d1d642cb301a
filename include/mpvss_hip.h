/* mpvss_hip.h -- C ABI of the MI355X-native batch group-exponentiation engine for mpvss-rs.
 *
 * This is the drop-in boundary for ONE hot path of the reference crate: the modexp inner
 * loops behind `distribute_secret`, `DLEQ::prove/verify` and `verify_distribution_shares`.
 * A Rust `impl Group for HipModpGroup` binds these symbols with
 * `extern "C"`; nothing here exposes C++, HIP or torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MPVSS_E_* code; nothing throws or
 *     aborts across the boundary; `mpvss_last_error(ctx)` gives a human-readable reason.
 *     (The reference returns `false`/`None` for structural problems and only panics on
 *     programmer errors such as threshold > n, src/participant.rs:166 -> MPVSS_E_INVALID.)
 *   - the caller owns every buffer; the library keeps no pointer after a call returns.
 *   - `space` says where the array arguments live: MPVSS_HOST (ordinary host memory) or
 *     MPVSS_DEVICE (HIP device memory of the context's GPU, e.g. a torch tensor's
 *     data_ptr()).  Scalar-like arguments documented as "host" are always host pointers.
 *   - MODP-2048 encoding: every element and scalar is a fixed 256-byte big-endian unsigned
 *     integer (zero padded).  The reference's minimal-length `element_to_bytes`
 *     (src/groups/modp.rs:150-152) only matters inside the Fiat-Shamir transcript, which
 *     the library frames itself (src/dleq.rs:58-61).  Elements need not be reduced: like
 *     `BigInt::modpow` the engine reduces them mod q (src/groups/modp.rs:154-156 does no
 *     validation).  Positions are the 1-based share indices (src/participant.rs:186).
 *   - a context may be used from several host threads (calls are serialised internally);
 *     `Group: Send + Sync` in the reference (src/group.rs:24).
 */
#ifndef MPVSS_HIP_H
#define MPVSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPVSS_OK 0
#define MPVSS_E_INVALID (-1)   /* bad argument (null pointer, n == 0 where forbidden, t > n, position < 0) */
#define MPVSS_E_DEVICE (-2)    /* HIP runtime error (see mpvss_last_error) */
#define MPVSS_E_NOMEM (-3)     /* workspace allocation failed */
#define MPVSS_E_UNSUPPORTED (-4)

#define MPVSS_HOST 0
#define MPVSS_DEVICE 1

#define MPVSS_MODP_BYTES 256   /* element / scalar width of the RFC 3526 group-14 encoding */

typedef struct mpvss_ctx mpvss_ctx;

/* ---- context ------------------------------------------------------------------------ */

/* Optional, process-wide, and only meaningful BEFORE the process's first HIP call (any library's): puts
 * GPU_MAX_HW_QUEUES=8 into the environment unless the variable is already set -- the block pipeline keeps one stream
 * pair per box in flight and wants 8 hardware queues (the runtime's default is 4; 16 or more oversubscribe the
 * hardware).  Returns 1 when it set the variable, 0 when a value was already there.  Not thread-safe against
 * concurrent getenv: call it at the top of main, or set the variable in the launcher instead.  The library never
 * changes the environment on its own. */
int mpvss_process_init(void);

/* Number of visible HIP devices (0 when there is no GPU; the library never falls back to
 * a CPU path -- every compute entry point then fails with MPVSS_E_DEVICE). */
int mpvss_device_count(void);

/* Create an engine bound to one GPU.  Replaces `ModpGroup::new()` (src/groups/modp.rs:44-69)
 * as the object the host-side Group impl holds in its `Arc`. */
int mpvss_ctx_create(int device_id, mpvss_ctx** out);
void mpvss_ctx_destroy(mpvss_ctx* ctx);
const char* mpvss_last_error(const mpvss_ctx* ctx);

/* Optional: run the engine's kernels on a caller-provided hipStream_t (passed as void*). */
int mpvss_ctx_set_stream(mpvss_ctx* ctx, void* hip_stream);
/* Block until everything queued by this context has finished. */
int mpvss_ctx_synchronize(mpvss_ctx* ctx);

/* ---- Group operations, batched ------------------------------------------------------- */

/* out[i] = bases[i]^exps[i] mod q.            Replaces n calls of ModpGroup::exp
 * (src/groups/modp.rs:122-128).  bases/exps/out: n x 256 bytes. */
int mpvss_modp_batch_exp(mpvss_ctx* ctx, int space, const uint8_t* bases, const uint8_t* exps, size_t n,
                         uint8_t* out);

/* out[i] = a[i]*b[i] mod q.                   Replaces ModpGroup::mul (src/groups/modp.rs:130-132). */
int mpvss_modp_batch_mul(mpvss_ctx* ctx, int space, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);

/* out[i] = base^exps[i] mod q for ONE base (host pointer, 256 bytes): the shape of
 * generate_public_key (modp.rs:176-178), commitments C_j = g^a_j (participant.rs:189-193) and
 * a1 = g^w (dleq.rs:207-216). */
int mpvss_modp_batch_exp_fixed_base(mpvss_ctx* ctx, int space, const uint8_t* base_host, const uint8_t* exps,
                                    size_t n, uint8_t* out);

/* ---- the commitment multi-exp -------------------------------------------------------- */

/* X[i] = prod_{j<t} C_j^(positions[i]^j mod (q-1)) mod q.
 * Replaces the loop at src/participant.rs:423-434 (= 207-215, = src/mpvss.rs:110-123).
 * commitments: t x 256 bytes; positions: n int64 (each >= 0); x_out: n x 256 bytes. */
int mpvss_modp_commit_eval(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                           const int64_t* positions, size_t n, uint8_t* x_out);

/* ---- DLEQ verifier commitments -------------------------------------------------------- */

/* a1[i] = g1^r[i] * h1[i]^c_i,  a2[i] = g2[i]^r[i] * h2[i]^c_i   (src/dleq.rs:66-84).
 * g1_host: one shared 256-byte base (host pointer).  c: one shared 256-byte challenge when
 * c_per_share == 0 (always a host pointer), else n x 256 bytes in `space`.
 * h1, g2, h2, r, a1_out, a2_out: n x 256 bytes in `space`. */
int mpvss_modp_dleq_commitments(mpvss_ctx* ctx, int space, const uint8_t* g1_host, const uint8_t* h1,
                                const uint8_t* g2, const uint8_t* h2, const uint8_t* r, const uint8_t* c,
                                int c_per_share, size_t n, uint8_t* a1_out, uint8_t* a2_out);

/* ---- verify_distribution_shares ------------------------------------------------------- */

/* Whole-box verification, src/participant.rs:399-455 (= src/mpvss.rs:90-144):
 *   for each share i (in array order): X_i (commit_eval), (a1_i, a2_i) = DLEQ commitments with
 *   (g, X_i, y_i, Y_i, r_i, c); transcript += framed(X_i) framed(Y_i) framed(a1_i) framed(a2_i)
 *   (minimal-length big-endian bytes, u64-BE length prefix, src/dleq.rs:58-61,87-99);
 *   *verdict = ( int(SHA256(SHA256(transcript))) mod (q-1)/2 == challenge ).
 * commitments t x 256; positions n; pubkeys (y_i), shares (Y_i), responses (r_i): n x 256, all in
 * `space`.  challenge: 256 bytes, host.  Outputs are host pointers: verdict (0/1),
 * digest32_out (optional, SHA-256 of the transcript), and optional n x 256 dumps of X, a1, a2
 * (pass NULL to skip).  An empty box (n == 0) hashes the empty transcript, as the reference does. */
int mpvss_modp_verify_distribution(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                   const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                   const uint8_t* responses, size_t n, const uint8_t* challenge_host,
                                   int* verdict, uint8_t* digest32_out, uint8_t* x_out_host,
                                   uint8_t* a1_out_host, uint8_t* a2_out_host);

/* Sharded form of the same verification (one engine per GPU, contiguous blocks of shares per
 * engine, blocks absorbed in share order).  `state` is an opaque MPVSS_TRANSCRIPT_STATE_BYTES
 * running-hash state that travels from the engine holding block k to the one holding block k+1:
 *   mpvss_transcript_init(state);
 *   for each block in order:  ..._verify_block_compute(block);  ..._verify_block_absorb(state);
 *   mpvss_modp_transcript_verdict(state, challenge, &verdict, digest);
 * `compute` only enqueues GPU work (kernels + device-to-host copies) and returns; `absorb` waits
 * for it, so the wait for the previous block's state overlaps this block's GPU work.
 * Up to MPVSS_BLOCK_SLOTS blocks may be in flight per engine (compute, compute, ..., absorb, absorb in FIFO order): the
 * host hash of box k then overlaps the GPU work of boxes k+1.. (bench.py).
 * mpvss_modp_verify_distribution == init + compute + absorb + verdict on one engine. */
#define MPVSS_TRANSCRIPT_STATE_BYTES 128
/* blocks of the block API that may be in flight in one context (any mix of the compute calls below) */
#define MPVSS_BLOCK_SLOTS 64
void mpvss_transcript_init(uint8_t* state);
int mpvss_modp_verify_block_compute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                    const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                    const uint8_t* responses, size_t n, const uint8_t* challenge_host);
int mpvss_modp_verify_block_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host, uint8_t* a1_out_host,
                                   uint8_t* a2_out_host);
/* mpvss_modp_verify_block_compute that also leaves one well-formedness byte per share in DEVICE memory
 * (wellformed_dev_out, n bytes, valid once the block has been absorbed): 1 iff 0 < y_i < q, 0 < Y_i < q and
 * r_i < q - 1, i.e. the three inputs of the share are canonical encodings.  The reference validates nothing here
 * (src/groups/modp.rs:154-156; src/participant.rs:408-448 hashes whatever comes out), so these bytes never enter the box
 * verdict: they are the per-share record that the ranks of a sharded verification all-gather over RCCL,
 * telling every rank which block holds a non-canonical share. */
int mpvss_modp_verify_block_compute_flags(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                          const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                          const uint8_t* responses, size_t n, const uint8_t* challenge_host,
                                          uint8_t* wellformed_dev_out);
/* The same in two steps, for callers whose transcript state arrives from elsewhere (one rank of a sharded verification:
 * the state of box b comes from the previous rank): mpvss_block_claim takes the oldest block in flight (a verifier's or
 * dealer's distribution block of any of the three groups) and returns its ticket -- tickets count the blocks of this context in enqueue order --,
 * mpvss_modp_verify_block_absorb_claimed waits for and hashes exactly that block.  Several threads may hold claimed
 * blocks at once; every claimed block must be absorbed. */
int mpvss_block_claim(mpvss_ctx* ctx, unsigned long long* ticket_out);
int mpvss_modp_verify_block_absorb_claimed(mpvss_ctx* ctx, unsigned long long ticket, uint8_t* state,
                                           uint8_t* x_out_host, uint8_t* a1_out_host, uint8_t* a2_out_host);
/* Host-only: extend the transcript with `count` 256-byte elements, each framed as
 * u64-BE(minimal length) || minimal-length big-endian bytes (src/dleq.rs:58-61, modp.rs:150-152). */
int mpvss_modp_transcript_absorb(uint8_t* state, const uint8_t* elements, size_t count);
int mpvss_modp_transcript_verdict(const uint8_t* state, const uint8_t* challenge_host, int* verdict,
                                  uint8_t* digest32_out);

/* Many boxes at once -- the situation of a participant that checks every dealer's box (each call of
 * src/participant.rs:399-455 is independent of the others).  The library pipelines them itself: the calling thread
 * enqueues the GPU work of up to `depth` boxes (1..16) ahead while `hash_threads` (1..8) library threads wait for the
 * boxes in order and hash their transcripts.  verdicts[i] / digests32[32 i .. 32 i + 31] (optional) belong to
 * boxes[i]; all array pointers of a box live in `space`, the challenge is a host pointer; `keyset` may be NULL
 * (registered keys: then `pubkeys` may be NULL, see below).  Equivalent to count calls of
 * mpvss_modp_verify_distribution. */
typedef struct mpvss_keyset mpvss_keyset;
typedef struct mpvss_modp_box {
  const uint8_t* commitments; size_t t;
  const int64_t* positions;
  const uint8_t* pubkeys; const uint8_t* shares; const uint8_t* responses; size_t n;
  const uint8_t* challenge_host;
  const mpvss_keyset* keyset; size_t key_offset;
} mpvss_modp_box;
int mpvss_modp_verify_many(mpvss_ctx* ctx, int space, const mpvss_modp_box* boxes, size_t count, int depth,
                           int hash_threads, int* verdicts, uint8_t* digests32);

/* The same pipeline for ONE box held by SEVERAL engines (one per GPU, src/participant.rs:399-455 with the participants
 * sharded): this engine verifies its contiguous block of every box; a box's transcript is one ordered hash, so the 128-byte
 * running state travels engine to engine through two callbacks the library calls on its own threads, several boxes at a time,
 * each box exactly once:
 *   state_in(user, box, state, 1)   fill in the state this engine's block of `box` starts from (may block until the engine before
 *                                   has sent it); return non-zero when an earlier engine failed on the box.  NULL: first engine.
 *   state_out(user, box, state, ok) hand the state on after the block has been absorbed (ok = 0: this or an earlier engine failed
 *                                   on the box, the state is zero).  NULL: last engine.
 * verdicts / digests32 come from this engine's final state: meaningful on the last engine.  wellformed_dev_out: NULL or `count`
 * device buffers of n bytes each that receive the shares' well-formedness flags (mpvss_modp_verify_block_compute_flags). */
typedef int (*mpvss_chain_cb)(void* user, size_t box, uint8_t* state, int ok);
int mpvss_modp_verify_many_chained(mpvss_ctx* ctx, int space, const mpvss_modp_box* boxes, size_t count, int depth,
                                   int hash_threads, uint8_t* const* wellformed_dev_out, mpvss_chain_cb state_in,
                                   mpvss_chain_cb state_out, void* user, int* verdicts, uint8_t* digests32);

/* ---- registered public keys (opt-in) ---------------------------------------------------- */

/* In a PVSS deployment the participants' public keys are long-lived: every dealer distributes to the same keys, so
 * a verifier checks many boxes against one key set (src/participant.rs:399-455 takes the same `publickeys` each
 * time).  A key set holds, in HBM, per-key tables for y_i^r (295 KB per key: 19.3 GB for 65536 keys) built once
 * (about as much work as verifying 1.5 boxes); verify_block_compute_keyset then computes a2_i = y_i^r_i * Y_i^c with
 * 613 products instead of 2 620.  Results are identical to mpvss_modp_verify_block_compute on the same keys.
 * Shares i of the call use keys key_offset + i of the set.  Destroy a key set only after the blocks using it have
 * been absorbed.  The DEALER to registered keys (round 6: mpvss_modp_deal_compute_keyset, and mpvss_modp_deal / mpvss_modp_distribute
 * through the cross-call cache below) takes Y_i = y_i^P(i) and a2_i = y_i^w_i (participant.rs:219, dleq.rs:213-216) from the same
 * tables: 2 x 548 instead of 2 865 Montgomery operations per share, byte-identical results. */
int mpvss_modp_keyset_create(mpvss_ctx* ctx, int space, const uint8_t* pubkeys, size_t n, mpvss_keyset** out);
/* The same behind the UNCHANGED call (round 5): with min_boxes >= 2, mpvss_modp_verify_many builds the tables by itself for every
 * public-key array that at least min_boxes large boxes (more shares than the grouped small boxes have) of ONE call present -- the
 * same pointer and n inside one call is the same array: exact, nothing is hashed or compared -- uses them for those boxes and frees
 * them when the call returns.  Building costs about one box of GPU time (60 ms per 65536 keys, 295 KB of HBM per key); a box then
 * takes 37 instead of 59 ms: it pays from the third box.  What a verifier of many dealers' boxes against the same participants
 * (src/participant.rs:399-455 with the same `publickeys` each time) gets without touching its code.  0 (the default): off.
 * While either key cache is on the context keeps the table BUFFER of the last set it dropped (one buffer) for the next set it builds
 * -- allocating 19 GB can cost ten times what building the tables does --; it is released when both caches are off, with the
 * context, or as soon as a workspace allocation of the context fails for lack of memory.
 * Returns the previous setting, or a negative error. */
int mpvss_ctx_set_key_cache(mpvss_ctx* ctx, int min_boxes);
/* ... and ACROSS calls, for callers that verify ONE box per call (the crate's call shape, src/participant.rs:399-455) against the
 * same participants again and again (round 6): with max_sets >= 1, mpvss_modp_verify_distribution identifies a HOST public-key array
 * by a SHA-256 tree hash of its bytes (16.8 MB per 65536 keys: eight slices hashed side by side, 1-2 ms, outside the context lock -- a
 * pointer may be re-used for other keys between calls, content may not), and the min_sightings-th box against the same array (default 2) builds its per-key
 * tables once; every later box against it takes the registered-key path.  At most max_sets (<= 8) sets have tables at a time (19.3 GB
 * per 65536 keys): the least recently used set that no block in flight reads makes room; tables that do not fit beside the block
 * slots' workspaces, or whose build fails, cost speed, not the call.  Same verdicts and digests as without the cache
 * (tests/test_gpu_keyset.py).  Boxes in device memory, small boxes (n <= 16384), boxes larger than one chunk and boxes whose challenge
 * does not fit 256 bits are left alone.  Dealers count as well: mpvss_modp_deal and mpvss_modp_distribute (host buffers, n > 16384)
 * look their key array up the same way, and a dealer's call is a sighting like a verifier's.  0: off (the default; frees the sets no
 * block reads).  Returns the previous max_sets, or a negative error. */
int mpvss_ctx_set_key_cache_lru(mpvss_ctx* ctx, int max_sets, int min_sightings);
void mpvss_modp_keyset_destroy(mpvss_ctx* ctx, mpvss_keyset* keyset);
size_t mpvss_modp_keyset_bytes(const mpvss_keyset* keyset);
int mpvss_modp_verify_block_compute_keyset(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                                           const int64_t* positions, const mpvss_keyset* keyset, size_t key_offset,
                                           const uint8_t* shares, const uint8_t* responses, size_t n,
                                           const uint8_t* challenge_host);

/* ---- verify_share, batched -------------------------------------------------------------- */

/* n independent share-box proofs, src/participant.rs:361-386 -> src/dleq.rs:275-302:
 *   a1 = G^r_i * pk_i^c_i, a2 = S_i^r_i * Y_i^c_i,
 *   verdicts[i] = ( hash_to_scalar(SHA256(framed(pk_i) framed(Y_i) framed(a1) framed(a2))) == c_i ).
 * Everything runs on the device, the hashing too (one lane per share, verdict_kernels.hip); only the n verdict bytes
 * come back.  pk, s (decrypted shares S_i), y (encrypted shares Y_i), c, r: n x 256 in `space`; verdicts: n bytes, host. */
int mpvss_modp_verify_shares(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* s, const uint8_t* y,
                             const uint8_t* c, const uint8_t* r, size_t n, uint8_t* verdicts_host);
/* The same in two steps, so that several batches are in flight (they share the MPVSS_BLOCK_SLOTS block slots and the FIFO order
 * of mpvss_modp_verify_block_compute / _absorb): `compute` only enqueues GPU work and returns; `absorb` waits for the
 * oldest batch and hands out its verdict bytes.  verdicts_dev_out (optional, device memory, n bytes) receives the
 * verdicts in stream order as well -- the tensor a multi-GPU caller all-gathers over RCCL without a host round trip
 * (valid once the batch has been absorbed or mpvss_ctx_synchronize has returned). */
int mpvss_modp_verify_shares_compute(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* s, const uint8_t* y,
                                     const uint8_t* c, const uint8_t* r, size_t n, uint8_t* verdicts_dev_out);
int mpvss_modp_verify_shares_absorb(mpvss_ctx* ctx, uint8_t* verdicts_host);

/* ---- distribute_secret, group part ------------------------------------------------------ */

/* Dealer side of src/participant.rs:160-286 with the randomness as INPUT (the reference draws
 * it from thread_rng, polynomial.rs:34-47 / modp.rs:162-174): given the polynomial values
 * p_i = P(i) mod (q-1) and witnesses w_i, computes
 *   X_i = prod C_j^(i^j)  (same loop as the verifier, participant.rs:207-215),
 *   Y_i = y_i^p_i (participant.rs:219), a1_i = g^w_i, a2_i = y_i^w_i (dleq.rs:207-216),
 * and the transcript digest SHA256(framed(X_i) framed(Y_i) framed(a1_i) framed(a2_i) ...)
 * (participant.rs:238-252).  Scalar-field work (responses r_i = w_i - p_i*c, polynomial
 * evaluation) stays on the host.  All arrays in `space`; digest32_out is a host pointer. */
int mpvss_modp_distribute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t,
                          const int64_t* positions, const uint8_t* pubkeys, const uint8_t* p_values,
                          const uint8_t* witnesses, size_t n, uint8_t* x_out, uint8_t* y_out,
                          uint8_t* a1_out, uint8_t* a2_out, uint8_t* digest32_out);

/* The same in the compute / absorb form of the verifier's blocks (shared slots and FIFO order): `compute` enqueues the
 * GPU work of one block of shares and returns, `absorb` waits for it and extends the dealer's transcript `state` with
 * framed(X_i) framed(Y_i) framed(a1_i) framed(a2_i) in share order (participant.rs:238-245); the digest of the finished
 * transcript is mpvss_modp_transcript_verdict's digest32_out.  With space == MPVSS_DEVICE the four *_dev_out arrays (optional)
 * receive the results in HBM; absorb's host pointers are optional copies.
 * commitments == NULL: X_i is computed as g^p_i through the fixed-base comb -- the same element as the loop of
 * participant.rs:207-215 whenever the box's commitments are C_j = g^a_j of the polynomial behind p_i = P(i), which is what
 * the dealer of participant.rs:160-286 builds (t and positions are then unused). */
int mpvss_modp_distribute_compute(mpvss_ctx* ctx, int space, const uint8_t* commitments, size_t t, const int64_t* positions,
                                  const uint8_t* pubkeys, const uint8_t* p_values, const uint8_t* witnesses, size_t n,
                                  uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out, uint8_t* a2_dev_out);
int mpvss_modp_distribute_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host, uint8_t* y_out_host,
                                 uint8_t* a1_out_host, uint8_t* a2_out_host);
/* The dealer's block with the polynomial evaluation on the device as well (participant.rs:196-249 for one box or one block of
 * it): `compute` takes the t coefficients (host memory; 256-byte big-endian scalars) and n positions, public keys and witnesses
 * in HBM, evaluates p_i = P(position_i) mod (q-1) into p_dev_out (n x 256 bytes in HBM, kept for
 * mpvss_modp_dleq_responses_device) on the block's own stream and goes on as mpvss_modp_distribute_compute with
 * commitments == NULL: X_i = g^p_i, Y_i = y_i^p_i, a1_i = g^w_i, a2_i = y_i^w_i.  Absorbed by mpvss_modp_distribute_absorb (a
 * negative position fails the block there).  n <= 262144 shares per call. */
/* ... and the whole of it in one call for HOST buffers: P(i), X_i, Y_i, a1_i, a2_i, the transcript digest, the challenge
 * c = hash_to_scalar(digest) (participant.rs:251-252) and the responses r_i = w_i - P(i) c (:255-264), the 2048-bit scalar
 * arithmetic on the device too.  What is left to the caller of participant.rs:160-286 is drawing the polynomial and the
 * witnesses, the commitments C_j = g^a_j (mpvss_modp_batch_exp_fixed_base) and U.  x_out, a1_out, a2_out, digest32_out and
 * challenge_out256 are optional; t <= n < 2^31: like the reference, the call has no size limit of its own -- a box of more than
 * 262144 shares is cut into blocks internally (one transcript; the GPU works on one block while the host hashes the one before).
 * The call keeps its inputs and intermediate secrets in buffers of its own (a set per deal in flight from the context's pool:
 * several host threads may deal on one context at once), and zeroes P(i), the witnesses and the staged coefficients (device and
 * pinned host copies) before it returns. */
int mpvss_modp_deal(mpvss_ctx* ctx, const uint8_t* coeffs_host, size_t t, const int64_t* positions_host,
                    const uint8_t* pubkeys_host, const uint8_t* witnesses_host, size_t n, uint8_t* x_out, uint8_t* y_out,
                    uint8_t* a1_out, uint8_t* a2_out, uint8_t* digest32_out, uint8_t* challenge_out256, uint8_t* r_out);
int mpvss_modp_deal_compute(mpvss_ctx* ctx, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev,
                            const uint8_t* pubkeys_dev, const uint8_t* witnesses_dev, size_t n, uint8_t* p_dev_out,
                            uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out, uint8_t* a2_dev_out);
/* mpvss_modp_deal_compute to REGISTERED keys: shares i of the block go to keys key_offset + i of `keyset` (mpvss_modp_keyset_create);
 * Y_i and a2_i come from the key tables.  Same outputs as mpvss_modp_deal_compute with those keys. */
int mpvss_modp_deal_compute_keyset(mpvss_ctx* ctx, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev,
                                   const mpvss_keyset* keyset, size_t key_offset, const uint8_t* witnesses_dev, size_t n,
                                   uint8_t* p_dev_out, uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out,
                                   uint8_t* a2_dev_out);

/* ---- elliptic-curve groups --------------------------------------------------------------------
 * The same entry points for the reference's two curve groups, selected by `group`:
 *   MPVSS_GROUP_SECP256K1    src/groups/secp256k1.rs:38-189     elements 33-byte SEC1 compressed
 *                            (the identity is 33 zero bytes, k256's GroupEncoding), scalars 32-byte big-endian
 *   MPVSS_GROUP_RISTRETTO255 src/groups/ristretto255.rs:45-253  elements 32-byte canonical ristretto255,
 *                            scalars 32-byte little-endian
 * Group law is written additively in the reference (exp = scalar multiplication, mul = point addition).
 * The reference's typed elements/scalars cannot hold an invalid encoding or a scalar >= the group
 * order; here such input makes the call fail with MPVSS_E_INVALID (mpvss_last_error names the index).
 * hash_to_scalar: SHA-256 then mod n (secp256k1.rs:121-131); SHA-512, little-endian, mod l
 * (ristretto255.rs:196-205).  Array shapes mirror the MODP calls with 256 replaced by the group's
 * element / scalar width; the challenge is 32 bytes. */
#define MPVSS_GROUP_SECP256K1 1
#define MPVSS_GROUP_RISTRETTO255 2

/* out[i] = scalars[i] * bases[i]      Secp256k1Group::exp secp256k1.rs:91-100, Ristretto255Group::exp ristretto255.rs:161-170 */
int mpvss_ec_batch_exp(mpvss_ctx* ctx, int group, int space, const uint8_t* bases, const uint8_t* scalars, size_t n,
                       uint8_t* out);
/* out[i] = a[i] + b[i]                ::mul secp256k1.rs:102-107, ristretto255.rs:172-177 */
int mpvss_ec_batch_mul(mpvss_ctx* ctx, int group, int space, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);
/* X[i] = sum_j (positions[i]^j mod order) * C_j     src/participant.rs:1404-1417 (secp256k1), 1847-1860 (ristretto255) */
int mpvss_ec_commit_eval(mpvss_ctx* ctx, int group, int space, const uint8_t* commitments, size_t t,
                         const int64_t* positions, size_t n, uint8_t* x_out);
/* a1[i] = r[i]*g1 + c_i*h1[i], a2[i] = r[i]*g2[i] + c_i*h2[i]      src/dleq.rs:66-84 */
int mpvss_ec_dleq_commitments(mpvss_ctx* ctx, int group, int space, const uint8_t* g1_host, const uint8_t* h1,
                              const uint8_t* g2, const uint8_t* h2, const uint8_t* r, const uint8_t* c, int c_per_share,
                              size_t n, uint8_t* a1_out, uint8_t* a2_out);
/* out[i] = scalars[i] * G for the group's generator: keygen (secp256k1.rs:168-171, ristretto255.rs:239-242), commitments
 * C_j = a_j G (participant.rs:1145-1152, 1626-1633), a1 = w G (dleq.rs:207-211) -- fixed-base comb, no doublings */
int mpvss_ec_batch_exp_generator(mpvss_ctx* ctx, int group, int space, const uint8_t* scalars, size_t n, uint8_t* out);
/* src/participant.rs:1384-1442 (secp256k1), 1827-1885 (ristretto255).  Positions become scalars by
 * `Scalar::from(position as u64)` (participant.rs:1419, 1862), so a negative position wraps instead of failing. */
int mpvss_ec_verify_distribution(mpvss_ctx* ctx, int group, int space, const uint8_t* commitments, size_t t,
                                 const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                 const uint8_t* responses, size_t n, const uint8_t* challenge_host, int* verdict,
                                 uint8_t* digest32_out, uint8_t* x_out_host, uint8_t* a1_out_host, uint8_t* a2_out_host);
/* The same split into compute (enqueue only) / absorb (wait, validate, hash) / verdict, sharing the MPVSS_BLOCK_SLOTS block slots
 * and the FIFO order of the MODP block calls, so that several boxes are in flight inside one context; and the
 * library-pipelined form for many boxes (see mpvss_modp_verify_many).  An invalid encoding or a response that is not
 * below the group order is reported by `absorb` (MPVSS_E_INVALID). */
int mpvss_ec_verify_block_compute(mpvss_ctx* ctx, int group, int space, const uint8_t* commitments, size_t t,
                                  const int64_t* positions, const uint8_t* pubkeys, const uint8_t* shares,
                                  const uint8_t* responses, size_t n, const uint8_t* challenge_host);
int mpvss_ec_verify_block_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host, uint8_t* a1_out_host,
                                 uint8_t* a2_out_host);
int mpvss_ec_transcript_absorb(int group, uint8_t* state, const uint8_t* elements, size_t count);
int mpvss_ec_transcript_verdict(int group, const uint8_t* state, const uint8_t* challenge_host, int* verdict,
                                uint8_t* digest32_out);
typedef struct mpvss_ec_box {
  const uint8_t* commitments; size_t t;
  const int64_t* positions;
  const uint8_t* pubkeys; const uint8_t* shares; const uint8_t* responses; size_t n;
  const uint8_t* challenge_host;
} mpvss_ec_box;
int mpvss_ec_verify_many(mpvss_ctx* ctx, int group, int space, const mpvss_ec_box* boxes, size_t count, int depth,
                         int hash_threads, int* verdicts, uint8_t* digests32);
/* src/participant.rs:1346-1371 (secp256k1), 1789-1814 (ristretto255) */
int mpvss_ec_verify_shares(mpvss_ctx* ctx, int group, int space, const uint8_t* pk, const uint8_t* s, const uint8_t* y,
                           const uint8_t* c, const uint8_t* r, size_t n, uint8_t* verdicts_host);
/* The same in two steps (several batches in flight; they share the block slots and the FIFO order with the other block
 * calls): compute enqueues, absorb waits, validates (an invalid encoding or scalar is reported here) and hands out the
 * verdict bytes.  verdicts_dev_out: optional device buffer that receives the n verdict bytes in stream order. */
int mpvss_ec_verify_shares_compute(mpvss_ctx* ctx, int group, int space, const uint8_t* pk, const uint8_t* s,
                                   const uint8_t* y, const uint8_t* c, const uint8_t* r, size_t n,
                                   uint8_t* verdicts_dev_out);
int mpvss_ec_verify_shares_absorb(mpvss_ctx* ctx, uint8_t* verdicts_host);
/* group part of src/participant.rs:1094-1274 (secp256k1), 1573-1717 (ristretto255); randomness is input */
int mpvss_ec_distribute(mpvss_ctx* ctx, int group, int space, const uint8_t* commitments, size_t t,
                        const int64_t* positions, const uint8_t* pubkeys, const uint8_t* p_values,
                        const uint8_t* witnesses, size_t n, uint8_t* x_out, uint8_t* y_out, uint8_t* a1_out,
                        uint8_t* a2_out, uint8_t* digest32_out);
/* The same as a block (the verifier's slots, staging layout and absorbing hash; mpvss_ec_transcript_verdict(state, zero
 * challenge) yields the dealer's digest): compute enqueues, absorb waits, validates, hashes and hands out X, Y, a1, a2.
 * commitments == NULL: X_i = P(i) * G through the fixed-base comb (the dealer knows the polynomial; positions unused).
 * *_dev_out (MPVSS_DEVICE only, optional): the results also stay in HBM. */
int mpvss_ec_distribute_compute(mpvss_ctx* ctx, int group, int space, const uint8_t* commitments, size_t t,
                                const int64_t* positions, const uint8_t* pubkeys, const uint8_t* p_values,
                                const uint8_t* witnesses, size_t n, uint8_t* x_dev_out, uint8_t* y_dev_out,
                                uint8_t* a1_dev_out, uint8_t* a2_dev_out);
int mpvss_ec_distribute_absorb(mpvss_ctx* ctx, uint8_t* state, uint8_t* x_out_host, uint8_t* y_out_host,
                               uint8_t* a1_out_host, uint8_t* a2_out_host);
/* The curve groups' dealer with the SCALAR side on the device as well (src/participant.rs:1134-1168, 1200-1230 secp256k1;
 * 1607-1631, 1662-1690 ristretto255) -- the counterparts of mpvss_modp_poly_eval_device / _dleq_responses_device / _deal_compute /
 * _deal.  Scalars are 32 bytes in the group's byte order; positions enter as `position as u64`, as in the reference.
 *   poly_eval_device     out_dev[i] = P(positions_dev[i]) mod order (t coefficients from the host, Horner's rule, one share per lane)
 *   dleq_responses_device r_dev_out[i] = w_dev[i] - alpha_dev[i] c mod order (one shared challenge from the host)
 *   deal_compute         one dealer's block with P(i) evaluated on the block's own stream into p_dev_out (kept for the responses),
 *                        then as mpvss_ec_distribute_compute with commitments == NULL; absorbed by mpvss_ec_distribute_absorb
 *   deal                 the whole box in one call from HOST buffers: P(i), X_i, Y_i, a1_i, a2_i, digest, challenge, responses;
 *                        x_out, a1_out, a2_out, digest32_out, challenge_out32 are optional; t <= n.  One deal at a time per context;
 *                        P(i), the witnesses and the coefficients are zeroed on the device before it returns. */
/* a curve-group block taken by mpvss_block_claim: waits for and hashes exactly that block (cf. mpvss_modp_verify_block_absorb_claimed) */
int mpvss_ec_block_absorb_claimed(mpvss_ctx* ctx, unsigned long long ticket, uint8_t* state, uint8_t* x_out_host,
                                  uint8_t* y_out_host, uint8_t* a1_out_host, uint8_t* a2_out_host);
int mpvss_ec_poly_eval_device(mpvss_ctx* ctx, int group, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev,
                              size_t n, uint8_t* out_dev);
int mpvss_ec_dleq_responses_device(mpvss_ctx* ctx, int group, const uint8_t* w_dev, const uint8_t* alpha_dev,
                                   const uint8_t* c_host32, size_t n, uint8_t* r_dev_out);
int mpvss_ec_deal_compute(mpvss_ctx* ctx, int group, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev,
                          const uint8_t* pubkeys_dev, const uint8_t* witnesses_dev, size_t n, uint8_t* p_dev_out,
                          uint8_t* x_dev_out, uint8_t* y_dev_out, uint8_t* a1_dev_out, uint8_t* a2_dev_out);
int mpvss_ec_deal(mpvss_ctx* ctx, int group, const uint8_t* coeffs_host, size_t t, const int64_t* positions_host,
                  const uint8_t* pubkeys_host, const uint8_t* witnesses_host, size_t n, uint8_t* x_out, uint8_t* y_out,
                  uint8_t* a1_out, uint8_t* a2_out, uint8_t* digest32_out, uint8_t* challenge_out32, uint8_t* r_out);
/* Group::hash_to_scalar(data), host only; out32 in the group's scalar byte order */
int mpvss_ec_hash_to_scalar(int group, const uint8_t* data, size_t len, uint8_t out32[32]);

/* ---- extract_secret_share, batched (the callers on the other side of the path) --------------------------------
 * n participants decrypt their encrypted share and build the DLEQ proof at once, src/participant.rs:294-353
 * (secp256k1 :1282-1338, ristretto255 :1725-1781):  S_i = Y_i^(1/x_i),  a1_i = G^w_i,  a2_i = S_i^w_i,
 * c_i = hash_to_scalar(SHA256(framed(pk_i) framed(Y_i) framed(a1_i) framed(a2_i))).
 * xinv = x_i^-1 mod the group order (host work: util.rs:33-41 / Scalar::invert) and the witnesses w are inputs;
 * the response r_i = w_i - x_i c_i (dleq.rs:42-50) stays with the host.  s_out in `space`; c_out_host: n scalars. */
int mpvss_modp_extract_shares(mpvss_ctx* ctx, int space, const uint8_t* pk, const uint8_t* y, const uint8_t* xinv,
                              const uint8_t* w, size_t n, uint8_t* s_out, uint8_t* c_out_host);
int mpvss_ec_extract_shares(mpvss_ctx* ctx, int group, int space, const uint8_t* pk, const uint8_t* y,
                            const uint8_t* xinv, const uint8_t* w, size_t n, uint8_t* s_out, uint8_t* c_out_host);
/* MODP, in two steps, so that several batches are in flight (they share the MPVSS_BLOCK_SLOTS block slots and the FIFO order of
 * the other block calls): `compute` forms e2 = w / x mod (q-1) on host threads and only ENQUEUES the GPU work -- S and a2 from
 * one chain of squarings, a1 through the comb, and the challenge hash of every proof on the device (K7) --, `absorb` waits for
 * the oldest batch and hands out S and c (n x 256 bytes each).  Host buffers.  A batch with an encrypted share that is 0 mod q
 * (no group element) is refused with MPVSS_E_UNSUPPORTED: it belongs to mpvss_modp_extract_shares. */
int mpvss_modp_extract_shares_compute(mpvss_ctx* ctx, const uint8_t* pk, const uint8_t* y, const uint8_t* xinv, const uint8_t* w,
                                      size_t n);
int mpvss_modp_extract_shares_absorb(mpvss_ctx* ctx, uint8_t* s_out_host, uint8_t* c_out_host);

/* ---- scalar-field side (host only, no context needed) ------------------------------------------------------------
 * The reference's scalar rings: Z/(q-1) for MODP-2048 (256-byte big-endian scalars), Z/n for secp256k1 (32-byte
 * big-endian), Z/l for ristretto255 (32-byte little-endian).  O(n) / O(n t) word operations per box against O(3000 n)
 * 2048-bit products on the group side: threaded over the shares on the host (threads <= 0: up to 16). */
/* Group::scalar_mul: (a * b) % order        src/group.rs:108, modp.rs:180-182, secp256k1.rs:173-176, ristretto255.rs:244-247 */
int mpvss_modp_scalar_mul(const uint8_t* a256, const uint8_t* b256, uint8_t* out256);
int mpvss_ec_scalar_mul(int group, const uint8_t* a32, const uint8_t* b32, uint8_t* out32);
/* Group::scalar_sub: a - b normalised into [0, order)   src/group.rs:113, modp.rs:184-192, secp256k1.rs:178-181 */
int mpvss_modp_scalar_sub(const uint8_t* a256, const uint8_t* b256, uint8_t* out256);
int mpvss_ec_scalar_sub(int group, const uint8_t* a32, const uint8_t* b32, uint8_t* out32);
/* r[i] = w[i] - alpha[i] * c_i (mod order): Prover::response / DLEQ::get_r (src/dleq.rs:42-50,221-228) for n proofs --
 * the dealer's responses (participant.rs:255-264: alpha = P(i), one shared c) and the participants'
 * (participant.rs:345-350: alpha = private key, c per share).  c: one scalar (c_per_share == 0) or n. */
int mpvss_modp_dleq_responses(const uint8_t* w, const uint8_t* alpha, const uint8_t* c, int c_per_share, size_t n,
                              uint8_t* r_out, int threads);
int mpvss_ec_dleq_responses(int group, const uint8_t* w, const uint8_t* alpha, const uint8_t* c, int c_per_share, size_t n,
                            uint8_t* r_out, int threads);
/* out[i] = P(positions[i]) mod order for P = sum_j coeffs[j] x^j: Polynomial::get_value (src/polynomial.rs:50-58) followed
 * by the caller's `% order` (participant.rs:202, 1155-1157, 1619-1621).  MODP positions must be >= 0. */
int mpvss_modp_poly_eval(const uint8_t* coeffs, size_t t, const int64_t* positions, size_t n, uint8_t* out, int threads);
int mpvss_ec_poly_eval(int group, const uint8_t* coeffs, size_t t, const int64_t* positions, size_t n, uint8_t* out, int threads);
/* The MODP pair on the DEVICE (2048-bit scalars: the host loops above cost more than the group work of a box): P(i) for
 * positions in HBM (>= 0) from the dealer's coefficients in host memory, results in HBM as 256-byte big-endian scalars in
 * [0, q-1) (what mpvss_modp_distribute_compute takes as p_values); and r[i] = w[i] - alpha[i] c mod (q-1) for ONE shared c
 * (host), w, alpha and r in HBM.  Same values as the host functions (src/polynomial.rs:50-58, src/dleq.rs:42-50);
 * synchronous: the results are complete when the call returns. */
int mpvss_modp_poly_eval_device(mpvss_ctx* ctx, const uint8_t* coeffs_host, size_t t, const int64_t* positions_dev, size_t n,
                                uint8_t* out_dev);
int mpvss_modp_dleq_responses_device(mpvss_ctx* ctx, const uint8_t* w_dev, const uint8_t* alpha_dev, const uint8_t* c_host, size_t n,
                                     uint8_t* r_dev_out);

/* ---- reconstruct ---------------------------------------------------------------------------------------------------
 * G^s = prod_i S_i^lambda_i from m >= t decrypted shares S_i at pairwise different positions (host int64):
 * src/participant.rs:462-561 (MODP; positions >= 1 as util.rs:47-64 assumes), 1452-1557 (secp256k1), 1895-2002
 * (ristretto255).  Lagrange coefficients in the scalar field on the host, the m exponentiations and their product on
 * the GPU.  gs_out: G^s (256 / 33 / 32 bytes); mask_out32 (optional): the 32-byte big-endian mask the reference XORs
 * onto U -- int_BE(SHA256(bytes(G^s))) mod q / mod n / mod l (participant.rs:512-517, 1495-1511, 1939-1949):
 * secret = mask XOR U.  shares in `space`. */
int mpvss_modp_reconstruct(mpvss_ctx* ctx, int space, const int64_t* positions_host, const uint8_t* shares, size_t m,
                           uint8_t* gs_out256, uint8_t* mask_out32);
int mpvss_ec_reconstruct(mpvss_ctx* ctx, int group, int space, const int64_t* positions_host, const uint8_t* shares, size_t m,
                         uint8_t* gs_out, uint8_t* mask_out32);

/* ---- flat wire format of a DistributionSharesBox ("MPVSSBX1") ---------------------------------------------------------
 * The reference keeps a box as three HashMap<Vec<u8>, _> (src/sharebox.rs:74-86) and defines no serialisation.  This
 * format is the boundary's own layout -- rows in `publickeys` order (the order src/participant.rs:408-448 iterates in),
 * fixed-width encodings -- so that a box can be handed to the engine, or cut into byte ranges for the ranks of a sharded
 * verification, without rebuilding maps.  Integers little-endian; every section starts at a multiple of 8 bytes
 * (zero padding, checked by the parser: an accepted buffer is its own serialisation); E = element bytes (256 / 33 / 32),
 * S = scalar bytes (256 / 32 / 32):
 *   0  "MPVSSBX1"   8  u32 group (0 MODP-2048, 1 secp256k1, 2 ristretto255)   12  u32 E   16  u64 n   24  u64 t
 *   32 u64 u_len    40 commitments [t][E] | positions i64 [n] | publickeys [n][E] | shares [n][E] | responses [n][S] |
 *   challenge [S] | U (big-endian magnitude, u_len bytes).                          (full specification: INTEGRATION.md) */
typedef struct mpvss_box_view {
  int group;
  size_t element_bytes, scalar_bytes, n, t, u_len;
  const uint8_t* commitments;
  const int64_t* positions;
  const uint8_t *pubkeys, *shares, *responses, *challenge, *u_be;
} mpvss_box_view;
size_t mpvss_box_wire_size(int group, size_t n, size_t t, size_t u_len);     /* 0: unknown group or absurd sizes */
int mpvss_box_serialize(int group, const uint8_t* commitments, size_t t, const int64_t* positions, const uint8_t* pubkeys,
                        const uint8_t* shares, const uint8_t* responses, size_t n, const uint8_t* challenge,
                        const uint8_t* u_be, size_t u_len, uint8_t* out, size_t out_cap, size_t* out_len);
/* Zero-copy view into `buf` (which must be 8-byte aligned and stay alive); rejects anything that is not exactly one box. */
int mpvss_box_parse(const uint8_t* buf, size_t len, mpvss_box_view* view);
/* verify_distribution_shares of a serialized box in host memory (any of the three groups) */
int mpvss_box_verify_wire(mpvss_ctx* ctx, const uint8_t* buf, size_t len, int* verdict, uint8_t* digest32_out);

/* ---- hashing helpers (host only; Group::hash_to_scalar, src/groups/modp.rs:142-148) ------ */

/* out32 = SHA-256(data) */
void mpvss_sha256(const uint8_t* data, size_t len, uint8_t out32[32]);
/* out256 = int_BE(SHA256(data)) mod (q-1)/2 as 256-byte big-endian */
void mpvss_modp_hash_to_scalar(const uint8_t* data, size_t len, uint8_t out256[256]);

/* ---- timing hooks for bench.py ------------------------------------------------------------ */

/* Milliseconds the GPU spent in the kernels of the most recent compute call on this context,
 * measured with hipEvents on the stream each kernel was launched on, summed per kind:
 *   0 = the X path (k_modp_commit_eval and, for consecutive positions, the forward-difference kernels),
 *   1 = k_modp_comb_dual_exp (a1), 2 = table builds, 3 = k_modp_dual_exp (a2, batch_exp).
 * Kernels of different kinds may overlap in time (two streams), so the sums can exceed the wall time.
 * mpvss_last_kernel_launches gives the number of launches behind each sum.  Negative when unavailable. */
double mpvss_last_kernel_ms(const mpvss_ctx* ctx, int kernel_id);
int mpvss_last_kernel_launches(const mpvss_ctx* ctx, int kernel_id);
/* Absorbed verify blocks whose X_i went through the forward-difference path, and how many of those fell back to
 * Horner's rule on the device (positions not consecutive, an X that is 0 mod q, a pipeline stage that gave up):
 * same results either way, the fall-back is just slower -- a counter for operators and tests.  The curve groups' blocks whose
 * X path ran behind a device gate (stage pipelines, device-resident positions) count the same way. */
int mpvss_modp_fd_stats(mpvss_ctx* ctx, unsigned long long* blocks, unsigned long long* fallbacks);

/* Host-side accounting of the block pipeline (verify_block_compute / _absorb / verify_many), sums over the blocks
 * absorbed since the last reset: time the calling threads spent enqueueing GPU work, waiting for the GPU and
 * hashing transcripts, and the per-kind kernel durations of mpvss_last_kernel_ms summed over those blocks. */
typedef struct mpvss_pipeline_stats {
  double enqueue_ms, wait_ms, hash_ms;
  double kernel_ms[4];
  unsigned long long kernel_launches[4];
  unsigned long long blocks;
} mpvss_pipeline_stats;
int mpvss_pipeline_stats_get(mpvss_ctx* ctx, mpvss_pipeline_stats* out, int reset);
/* Blocks of the block API currently in flight in this context (enqueued, not yet fully absorbed) and how many of them
 * still have GPU work pending: for monitoring and for callers that schedule compute / absorb themselves. */
int mpvss_blocks_in_flight(mpvss_ctx* ctx, int* in_flight_out, int* gpu_pending_out);
/* 1 when the transcript hash uses the CPU's SHA extensions (about 1.7 GB/s per thread), 0 for the portable code */
int mpvss_sha256_uses_shani(void);
/* Measurement aid (no reference counterpart): what this device sustains, now, of the instruction the kernels are made of -- four
 * waves per SIMD on every CU issue it back to back for about target_ms.  kind 0: v_mad_u64_u32 (the limb product of all three
 * groups), kind 1: 32-bit integer work (v_add3_u32, v_and_b32, v_lshl_add_u32), 2: the 64-bit shifts of a retire step (v_lshl_add_u64,
 * v_lshrrev_b64), 3: v_permlane32_swap, 4: v_mov_b64, 5: VOP2 v_add_u32 -- the classes bench.py prices the kernels' instruction mix
 * with (`compute.peak_mix_weighted`).  Wave-instructions per second over the whole chip, the shader clock the waves saw
 * (s_memtime against the 100 MHz s_memrealtime) and the duration; the last two are optional. */
int mpvss_issue_probe(mpvss_ctx* ctx, int kind, double target_ms, double* insts_per_s_out, double* shader_clock_ghz_out,
                      double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MPVSS_HIP_H */
