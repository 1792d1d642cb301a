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
int modp_launch_comb_dual_exp(const uint32_t* comb, const uint32_t* tab2, size_t tab2_stride, const uint8_t* e1,
                              const uint8_t* e2, size_t e2_stride, int e2_windows, int count, uint8_t* out,
                              const void* cs, hipStream_t s);
int modp_occupancy_report(int* out5);
#ifdef __cplusplus
}
#endif
