// Internal launch interface of the per-share verdict kernels (verdict_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* verdict[i] = hash_to_scalar(SHA256(framed(h1_i) framed(h2_i) framed(a1_i) framed(a2_i))) == c_i; all arrays [count][256] */
int verdict_launch_modp(const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2, const uint8_t* c,
                        int count, uint8_t* verdict, hipStream_t s);
/* c_out32[i] = SHA256(SHA256(framed(h1_i) framed(h2_i) framed(a1_i) framed(a2_i))) as 32 big-endian bytes: the challenge of a
 * share-box proof (hash_to_scalar of the transcript digest, participant.rs:329-343; the mod (q-1)/2 is the identity) */
int verdict_launch_modp_challenge(const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2, int count,
                                  uint8_t* c_out32, hipStream_t s);
/* out[i] = 1 iff 0 < y_i < q, 0 < Y_i < q, r_i < q-1 (canonical encodings); bounds = [q | q-1], 2 x 256 big-endian bytes */
int verdict_launch_modp_wellformed(const uint8_t* y, const uint8_t* Y, const uint8_t* r, const uint8_t* bounds, int count,
                                   uint8_t* out, hipStream_t s);
/* group 1 = secp256k1 (33-byte elements), 2 = ristretto255 (32-byte); c, r: [count][32] in the group's byte order;
 * ok[i] = 0 when c_i or r_i is not below the group order */
int verdict_launch_ec(int group, const uint8_t* h1, const uint8_t* h2, const uint8_t* a1, const uint8_t* a2,
                      const uint8_t* c, const uint8_t* r, int count, uint8_t* verdict, uint8_t* ok, hipStream_t s);
/* *first_bad = min(*first_bad, index of the first scalar >= the group order); scalars [count][32] */
int verdict_launch_check_scalars(int group, const uint8_t* scalars, int count, int* first_bad, hipStream_t s);
/* the curve groups' scalar ring on the device: out[i] = P(positions[i]) mod order for the t coefficients at coeffs_dev (32 bytes
 * each, the group's byte order); coef_m_scratch: t x 8 words of device scratch.  r[i] = w[i] - alpha[i] c mod order (c_stride 0:
 * one shared c) */
int ec_scalar_launch_poly_eval(int group, const uint8_t* coeffs_dev, int t, const int64_t* positions, int count,
                               uint32_t* coef_m_scratch, uint8_t* out, hipStream_t s);
int ec_scalar_launch_responses(int group, const uint8_t* w, const uint8_t* alpha, const uint8_t* c, size_t c_stride, int count,
                               uint8_t* out, hipStream_t s);
/* measurement aid: `blocks` x 4 waves, each issuing iters x 64 VALU instructions of `kind` (0: v_mad_u64_u32, 1: 32-bit integer
 * work); stamps[2 w] / [2 w + 1] = shader-clock / 100 MHz wall ticks of wave w; out: blocks x 256 words */
int issue_probe_launch(uint32_t* out, unsigned long long* stamps, int blocks, int iters, int kind, hipStream_t s);
#ifdef __cplusplus
}
#endif
