// Internal launch interface between the C-ABI layer and the elliptic-curve kernels.
// group: 1 = secp256k1 (33-byte SEC1 compressed points, 32-byte big-endian scalars),
//        2 = ristretto255 (32-byte points, 32-byte little-endian scalars).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int ec_point_words(int group);
int ec_launch_decode(int group, const uint8_t* enc, int count, uint32_t* pts, uint8_t* ok, hipStream_t s);
/* gate / want: run only if *gate == want (null: always) -- the forward-difference kernels and Horner's rule exclude
 * each other through a device flag when the positions are device-resident */
int ec_launch_commit_eval(int group, const uint32_t* cm, int t, const int64_t* positions, int count, uint8_t* x_enc,
                          const int* gate, int want, hipStream_t s);
int ec_launch_dual_mul(int group, const uint8_t* p1, size_t p1_stride, const uint8_t* k1, const uint8_t* p2,
                       const uint8_t* k2, size_t k2_stride, int count, uint8_t* out, uint8_t* ok, hipStream_t s);
/* forward differences for consecutive positions: seeds at chain indices w0..w0+t-1, tables, stepping both ways,
 * encoding; pts [count][point words], state_fwd / state_bwd [chains*t][point words] */
/* the same launches for `boxes` boxes of one shape laid out one after the other (strides per box: words of commitments,
   positions, words of points, words of difference tables, bytes of encodings; gate[b] per box) */
int ec_launch_fd_boxes(int group, const uint32_t* cm, int t, const int64_t* positions, int count, int chains, int w0,
                       int chain_len, uint32_t* pts, uint32_t* state_fwd, uint32_t* state_bwd, uint32_t* state_l1,
                       uint8_t* x_enc, int split_seeds, const int* gate, int boxes, size_t cm_stride, size_t pos_stride,
                       size_t pts_stride, size_t state_stride, size_t enc_stride, hipStream_t s);
/* the same with the stepping launches as pipelines of quad-lane stages (quad != null): for boxes that have the chip to
   themselves.  hand: ec_fd_quad_hand_words() words per box (zeroed by the launcher), gate: the boxes' writable gate
   (a stage that gives up clears it: the gated Horner launch must follow), fault: test hook (1 = one stage gives up) */
typedef struct EcQuadStepping {
  uint32_t* hand;
  size_t hand_box_words;
  int* gate;
  int fault;
  uint32_t* wtab;          /* scratch window tables of the seed kernel: ec_fd_seed_tab_words() words per box, or null (8 lanes per seed, bit by bit) */
  size_t wtab_box_words;
  int table;               /* the difference tables by the same pipeline (1) or one workgroup per chain (0) */
  int oct;                 /* secp256k1: eight lanes per point, six products side by side (1) or four lanes (0) */
} EcQuadStepping;
size_t ec_fd_quad_hand_words(int group, int oct, int t, int chains, int w0, int chain_len);
size_t ec_fd_seed_tab_words(int group, int seeds);
int ec_launch_fd_boxes_q(int group, const uint32_t* cm, int t, const int64_t* positions, int count, int chains, int w0,
                         int chain_len, uint32_t* pts, uint32_t* state_fwd, uint32_t* state_bwd, uint32_t* state_l1,
                         uint8_t* x_enc, int split_seeds, const int* gate, int boxes, size_t cm_stride, size_t pos_stride,
                         size_t pts_stride, size_t state_stride, size_t enc_stride, const EcQuadStepping* quad, hipStream_t s);
int ec_launch_commit_eval_boxes(int group, const uint32_t* cm, int t, const int64_t* positions, int count, uint8_t* x_enc,
                                const int* gate, int want, int boxes, size_t cm_stride, size_t pos_stride, size_t enc_stride,
                                hipStream_t s);
int ec_launch_encode_boxes(int group, const uint32_t* pts, int count, uint8_t* enc, const int* gate, int boxes,
                           size_t pts_stride, size_t enc_stride, hipStream_t s);
int ec_launch_fd(int group, const uint32_t* cm, int t, const int64_t* positions, int count, int chains, int w0,
                 int chain_len, uint32_t* pts, uint32_t* state_fwd, uint32_t* state_bwd, uint32_t* state_l1, uint8_t* x_enc,
                 int split_seeds, const int* gate, hipStream_t s);   /* state_l1: 2 * t points of scratch (two-level seeding) or null */
int ec_launch_add(int group, const uint8_t* a, const uint8_t* b, int count, uint8_t* out, uint8_t* ok, hipStream_t s);
/* windowed double-scalar multiplication (signed 4-bit windows; see ec_kernels.hip):
 *   comb: 65 x 8 packed affine multiples of the generator (ec_comb_words() words, built once by ec_launch_comb_build)
 *   tables: [count][8][ec_cached_words()] multiples P .. 8P of per-share bases, from encodings or internal points
 *   dual_win: out_pts[x] = k1[x] * (G | P1[x]) + k2[x] * P2[x] in internal coordinates; encode: -> canonical bytes */
int ec_cached_words(int group);
int ec_comb_words(int group);
int ec_launch_comb_build(int group, uint32_t* comb, hipStream_t s);
int ec_launch_build_tables(int group, const uint8_t* enc, size_t enc_stride, const uint32_t* pts, int count, uint32_t* tab,
                           uint8_t* ok, const int* gate, hipStream_t s);
int ec_launch_dual_win(int group, const uint32_t* comb, const uint32_t* tab1, const uint8_t* k1, size_t k1_stride,
                       const uint32_t* tab2, const uint8_t* k2, size_t k2_stride, int count, uint32_t* out_pts, hipStream_t s);
int ec_launch_encode(int group, const uint32_t* pts, int count, uint8_t* enc, hipStream_t s);
/* out[x] = a[x] + b[x], internal coordinates */
int ec_launch_add_pts(int group, const uint32_t* a, const uint32_t* b, int count, uint32_t* out, hipStream_t s);
/* out (one internal point, may alias pts) = sum of the m internal points at pts */
int ec_launch_sum(int group, const uint32_t* pts, int m, uint32_t* out, hipStream_t s);
int ec_launch_encode_gated(int group, const uint32_t* pts, int count, uint8_t* enc, const int* gate, hipStream_t s);
#ifdef __cplusplus
}
#endif
