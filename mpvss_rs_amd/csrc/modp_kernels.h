// Internal launch interface between the C-ABI layer (mpvss_capi.cpp) and the MODP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "modp2048_consts.h"

#define MODP_N0INV_C MODP_N0INV
#define MODP_TABLE_WORDS (16 * MODP_L)   /* one 4-bit window table, words */

#ifdef __cplusplus
extern "C" {
#endif
int modp_consts_upload(void** dev_consts);
int modp_launch_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int count, const void* cs, hipStream_t s);
int modp_launch_to_mont(const uint8_t* in, uint32_t* out_m, int count, const void* cs, hipStream_t s);
int modp_launch_commit_eval(const uint32_t* cm, int t, const int64_t* positions, int count, uint32_t* x_m,
                            uint8_t* x_be, const void* cs, hipStream_t s);
int modp_launch_build_table(const uint8_t* base_be, int count, uint32_t* tab, const void* cs, hipStream_t s);
int modp_launch_dual_exp(const uint32_t* tab1, size_t tab1_stride, const uint32_t* tab2, size_t tab2_stride,
                         const uint8_t* e1, const uint8_t* e2, size_t e2_stride, int e2_windows, int count,
                         uint8_t* out, const void* cs, hipStream_t s);
#define MODP_COMB_WORDS (512 * 16 * MODP_L)   /* fixed-base comb table, words */
int modp_launch_comb_build(const uint8_t* base_be_dev, uint32_t* comb, const void* cs, hipStream_t s);
#define MODP_COMB16_WORDS ((size_t)128 * 65536 * MODP_L)   /* wide comb (16-bit windows): 2.5 GB */
int modp_launch_comb16_build(const uint32_t* comb4, uint32_t* comb16, const void* cs, hipStream_t s);
int modp_launch_comb_dual_exp(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride, const uint8_t* e1,
                              const uint8_t* e2, size_t e2_stride, int e2_windows, int count, uint8_t* out,
                              int comb_bits, const void* cs, hipStream_t s);
int modp_launch_comb_dual_exp_split(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride, const uint8_t* e1,
                                    const uint8_t* e2, size_t e2_stride, int e2_windows, int count, uint8_t* out, int mode,
                                    uint32_t* p_m, int comb_bits, const void* cs, hipStream_t s);
/* a2 = y^r Y^c with a 64-entry (6-bit window) table for y: tab1 [count][64][72], tab2 [count][16][72]; every c < 2^256, c_stride 0 = one c for all */
int modp_launch_build_table64(const uint8_t* base_be, int count, uint32_t* tab, const void* cs, hipStream_t s);
int modp_launch_dual_exp_w6(const uint32_t* tab1, const uint32_t* tab2, const uint8_t* e1, const uint8_t* c,
                            size_t c_stride, int count, uint8_t* out, const void* cs, hipStream_t s);
/* One challenge c for all shares of a box, as a sliding-window schedule (device memory, uint16: [0] = windows, then
   (lowest bit position of the window, odd digit) by descending position); tables of the second base need their odd
   entries only (modp_launch_build_table_odd).  About 51 products instead of 64 per base, 8 instead of 14 per table. */
int modp_launch_build_table_odd(const uint8_t* base_be, int count, uint32_t* tab, const void* cs, hipStream_t s);
int modp_launch_dual_exp_w6_sched(const uint32_t* tab1, const uint32_t* tab2, const uint8_t* e1, const uint16_t* c_sched,
                                  int count, uint8_t* out, const void* cs, hipStream_t s);
int modp_launch_comb_dual_exp_sched(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride, const uint16_t* c_sched,
                                    int count, uint8_t* out, uint32_t* p_m, int comb_bits, const void* cs, hipStream_t s);
/* one base, two exponents, two results (dealer: Y = y^p, a2 = y^w): right-to-left 5-bit buckets in HBM, shared squarings.
   buckets: count * modp_twin_exp_bucket_words() u32 of scratch, occupancy: count * 2 u32 of scratch */
size_t modp_twin_exp_bucket_words(void);
int modp_launch_twin_exp(const uint8_t* base_be, const uint8_t* e1, const uint8_t* e2, int count, uint32_t* buckets,
                         uint32_t* occupancy, uint8_t* out1, uint8_t* out2, const void* cs, hipStream_t s);
/* registered public keys: per-key tables for y^r (see modp_kernels.hip) */
size_t modp_keyset_words_per_key(void);
int modp_launch_keyset_build(const uint8_t* pk_be, int count, uint32_t* ks, const void* cs, hipStream_t s);
int modp_launch_keyset_dual_exp(const uint32_t* ks, const uint32_t* tab2, const uint8_t* r, const uint8_t* c, int count,
                                uint8_t* out, const void* cs, hipStream_t s);
/* pair layout (modp_pair_kernels.hip): a2 = y^r Y^c with the Montgomery reduction on the matrix cores; same tables, same
   exponents, same schedule conventions as modp_launch_dual_exp_w6 / _sched (c_sched null: fixed windows of c, c null too:
   y^r alone).  pair_tables: device copy of the constant digit matrices (modp_pair_tables_upload, once per context). */
int modp_pair_tables_upload(void** dev_tables);
int modp_launch_dual_exp_w6_pair(const uint32_t* tab1, const uint32_t* tab2, const uint8_t* e1, const uint8_t* c,
                                 size_t c_stride, const uint16_t* c_sched, int count, uint8_t* out, const void* cs,
                                 const void* pair_tables, hipStream_t s);
/* pair-layout forms of the table builders (entries 64, or 16 with odd_only), of the wide-comb exponentiation g^e1 (mode 1
   of modp_launch_comb_dual_exp_split with the 16-bit comb) and of p_m * B2^c for a scheduled shared c (modp_launch_comb_dual_exp_sched) */
int modp_launch_build_table_pair(const uint8_t* base_be, int count, uint32_t* tab, int entries, int odd_only, const void* cs,
                                 const void* pair_tables, hipStream_t s);
int modp_launch_comb16_exp_pair(const uint32_t* comb16, const uint8_t* e1, int count, uint32_t* p_m, const void* cs,
                                const void* pair_tables, hipStream_t s);
int modp_launch_sched_exp_mul_pair(const uint32_t* tab2, size_t tab2_stride, const uint16_t* c_sched, const uint32_t* p_m,
                                   int count, uint8_t* out, const void* cs, const void* pair_tables, hipStream_t s);
int modp_occupancy_report(int* out5);
/* forward-difference evaluation of X_i for consecutive positions (see modp_kernels.hip) */
int modp_launch_commit_eval_gated(const uint32_t* cm_a, const uint32_t* cm_b, int split, int t, const int64_t* positions,
                                  int count, uint32_t* x_m, uint8_t* x_be, const int* gate, int want, const void* cs,
                                  hipStream_t s);
int modp_fd_tpad(int t);
int modp_launch_fd_check_positions(const int64_t* positions, int count, int* flag, hipStream_t s);
/* simultaneous inversion of m Montgomery-form numbers: one level up / down of the product tree (groups of G) */
int modp_launch_binv_up(const uint32_t* a, int m, int G, uint32_t* prefix, uint32_t* totals, const int* gate,
                        const void* cs, hipStream_t s);
int modp_launch_binv_down(const uint32_t* a, const uint32_t* prefix, const uint32_t* tot_inv, int m, int G,
                          uint32_t* a_inv, const int* gate, const void* cs, hipStream_t s);
int modp_launch_fd_apply_ok(const int* ok, int* flag, hipStream_t s);
/* pipelined single-wave stages; `hand` = zeroed handoff buffer of modp_fd_*_hand_words() 32-bit words */
size_t modp_fd_table_hand_words(int chains, int t);
size_t modp_fd_step_hand_words(int chains, int t, int chain_len);
int modp_launch_fd_table(const uint32_t* x, const uint32_t* x_inv, int chains, int t, uint32_t* state,
                         uint32_t* state_back, uint32_t* hand, int* gate, int inject_fault, const void* cs, hipStream_t s);
int modp_launch_fd_step(const uint32_t* state, const uint32_t* state_back, int chains, int t, int w0, int chain_len,
                        int count, uint32_t* x_m, uint32_t* hand, int* gate, int inject_fault, const void* cs,
                        hipStream_t s);
int modp_launch_from_mont(const uint32_t* x_m, int count, uint8_t* out_be, const int* gate, const void* cs,
                          hipStream_t s);
/* the same phases for a GROUP of same-shaped boxes in one launch each (box b = blockIdx.y; box_*: strides from one box to the next,
   in 32-bit words for limb buffers, in elements for positions and output rows) */
int modp_launch_commit_eval_boxes(const uint32_t* cm, int t, const int64_t* positions, size_t box_positions, int count, int boxes,
                                  uint32_t* x_m, uint8_t* x_be, size_t box_out, const int* gate, int want, const void* cs,
                                  hipStream_t s);
int modp_launch_fd_check_positions_boxes(const int64_t* positions, size_t box_positions, int count, int boxes, int* flag,
                                         hipStream_t s);
int modp_launch_fd_table_boxes(const uint32_t* x, size_t box_x, const uint32_t* x_inv, size_t box_xinv, int chains, int t,
                               uint32_t* state, uint32_t* state_back, size_t box_state, uint32_t* hand, size_t box_hand, int boxes,
                               int* gate, int inject_fault, const void* cs, hipStream_t s);
int modp_launch_fd_step_boxes(const uint32_t* state, const uint32_t* state_back, size_t box_state, int chains, int t, int w0,
                              int chain_len, int count, uint32_t* x_m, size_t box_xm, uint32_t* hand, size_t box_hand, int boxes,
                              int* gate, int inject_fault, const void* cs, hipStream_t s);
/* twin exponentiation (one base, two exponents): bucket phase on the pair layout + the shared combine phase */
#define MODP_BUCKET_W 5
#define MODP_TWIN_SLACK_BYTES 65536   /* once per buffer: running powers of the padding lanes of the last workgroup */
#define MODP_TWIN_EXTRA_WORDS 80      /* per share behind the buckets: 2 occupancy masks, padding, the running power (pair kernel) */
int modp_launch_twin_exp_pair(const uint8_t* base_be, const uint8_t* e1, const uint8_t* e2, int count, uint32_t* buckets,
                              uint32_t* occupancy, uint8_t* out1, uint8_t* out2, const void* cs, const void* pair_tables, hipStream_t s);
int modp_launch_bucket_combine(const uint32_t* buckets, const uint32_t* occupancy, int count, uint8_t* out1, uint8_t* out2,
                               const void* cs, hipStream_t s);
/* out = g^r * B2^c, c a 256-bit exponent per share (stride c_stride; 0: shared), B2's 16-entry table in tab2 (stride 16 entries), pair layout */
int modp_launch_comb16_dual_exp_pair(const uint32_t* comb16, const uint32_t* tab2, const uint8_t* r, const uint8_t* c, size_t c_stride, int count,
                                     uint8_t* out, const void* cs, const void* pair_tables, hipStream_t s);
/* out1 = g^e1, out2 = g^e2 through the wide comb (16-bit teeth), canonical bytes, pair layout (the dealer's X and a1; e2 == NULL: out1 alone) */
int modp_launch_comb16_twin_exp_pair(const uint32_t* comb16, const uint8_t* e1, const uint8_t* e2, int count, uint8_t* out1, uint8_t* out2,
                                     const void* cs, const void* pair_tables, hipStream_t s);
/* the dealer against registered keys: out1 = y^e1, out2 = y^e2 from the key tables (full-width exponents), pair layout */
int modp_launch_keyset_twin_exp_pair(const uint32_t* ks, const uint8_t* e1, const uint8_t* e2, int count, uint8_t* out1, uint8_t* out2,
                                     const void* cs, const void* pair_tables, hipStream_t s);
/* a2 = y^r Y^c against a registered key's table, pair layout (same table and program as modp_launch_keyset_dual_exp) */
int modp_launch_keyset_dual_exp_pair(const uint32_t* ks, const uint32_t* tab2, const uint8_t* r, const uint8_t* c, int count, uint8_t* out,
                                     const void* cs, const void* pair_tables, hipStream_t s);
/* forward-difference stepping on the pair layout (stages of 32 levels; same state, hand-over buffers and outputs) */
int modp_launch_fd_step_pair_boxes(const uint32_t* state, const uint32_t* state_back, size_t box_state, int chains, int t, int w0,
                                   int chain_len, int count, uint32_t* x_m, size_t box_xm, uint32_t* hand, size_t box_hand, int boxes,
                                   int* gate, int inject_fault, const void* cs, const void* pair_tables, hipStream_t s);
/* the same stepping as one wide launch per anti-diagonal of the (stage, block of tile_steps steps) grid: no wave waits for another,
 * no time-out; `state` / `state_back` are updated in place (modp_pair_kernels.hip: k_modp_fd_step_pair_tile) */
int modp_launch_fd_step_pair_tiled_boxes(uint32_t* state, uint32_t* state_back, size_t box_state, int chains, int t, int w0, int chain_len,
                                         int count, uint32_t* x_m, size_t box_xm, uint32_t* hand, size_t box_hand, int boxes,
                                         const int* gate, int tile_steps, const void* cs, const void* pair_tables, hipStream_t s);
/* scalar ring Z/(q-1) on the device (constants of q' = (q-1)/2: modq_consts_upload) */
int modq_consts_upload(void** dev_consts);
int modq_launch_poly_eval(const uint32_t* coef, int t, const int64_t* positions, int count, int par_even, int par_odd, uint8_t* out_be,
                          const void* cs_q, hipStream_t s);
int modq_launch_responses(const uint8_t* w_be, const uint8_t* alpha_be, const uint8_t* cneg_be, int c_parity, int count,
                          uint8_t* out_be, const void* cs_q, hipStream_t s);
/* row layout (modp_row_kernels.hip, bn_row.h: 16 lanes per number): the Horner seeds of an X path that has the chip to itself -- X at
 * `count` positions per box in Montgomery limb form; gate != null: only when *gate == gate_want */
int modp_launch_commit_eval_row_boxes(const uint32_t* cm, int t, const int64_t* positions, size_t box_positions, int count, int boxes,
                                      uint32_t* x_m, size_t box_out, const int* gate, int gate_want, const void* cs, hipStream_t s,
                                      int prio);
/* test hook: out = a * b R^-1 (sq == 0) or a^2 R^-1 through the row-layout product, limb form in and out */
/* out_m = B1^e1 * B2^e2 (and, with e1b, a second exponent set over the same tables behind it) in Montgomery limb form: small batches on the
 * row layout (the latency of one number's chain); the tables and exponents of modp_launch_dual_exp */
int modp_launch_dual_exp_row(const uint32_t* tab1, size_t tab1_stride, const uint32_t* tab2, size_t tab2_stride, const uint8_t* e1,
                             const uint8_t* e2, size_t e2_stride, int e2_windows, const uint8_t* e1b, const uint8_t* e2b, int count,
                             uint32_t* out_m, const void* cs, hipStream_t s);
int modp_launch_row_unit(const uint32_t* a_m, const uint32_t* b_m, int count, int sq, uint32_t* out_m, const void* cs, hipStream_t s);
/* out[i] = a[i] * b[i] mod (q-1), 256-byte big-endian each (device pointers) */
int modq_launch_mul(const uint8_t* a_be, const uint8_t* b_be, int count, uint8_t* out_be, const void* cs_q, hipStream_t s);
int modp_launch_gather_rows(const uint32_t* src, size_t src_stride_words, size_t row_words, int boxes, uint32_t* dst, hipStream_t s);
int modp_launch_spread_rows(const uint8_t* rows, int group, int count, uint8_t* out, hipStream_t s);
#ifdef __cplusplus
}
#endif
