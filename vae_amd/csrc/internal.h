// Internal interfaces used by step.hip (not part of the public C ABI): variants of public entry points that also
// record max|output| for the two-f16-term GEMMs, and those GEMMs themselves.
#pragma once
#include <stdint.h>

// slot = max(slot, max|X|) with an integer atomic; the caller zeroes the slot once per step
int vae_absmax_accumulate(const float* X, int64_t rows, int64_t cols, int64_t ld, float* slot, void* stream);

// max|w0| / max|w1| of two dense 16-byte aligned arrays raised into slot0 / slot1 (which must start at 0) and nzero floats
// at zero_slots cleared, one launch
int vae_absmax_weights(const float* w0, int64_t n0, const float* w1, int64_t n1, float* slot0, float* slot1, float* zero_slots,
                       int nzero, void* stream);

// vae_gemm (slabs_out == null) / vae_gemm_slabs (slabs_out != null: partial slabs left in the workspace) on two scaled
// f16 terms per element and three matrix-core products; the power-of-two scales are derived inside the kernel from the
// max|A| / max|B| slots the producers of the operands filled
int vae_gemm_f16x2(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* absmax_a,
                   const float* B, int64_t ldb, const float* absmax_b, float* C, int64_t ldc, const float* bias,
                   int accumulate, float* workspace, int64_t workspace_bytes, void* stream, int* slabs_out);

// vae_normalize_rows / _u8 (x_dtype) that also raises *absmax to max|xn| (absmax may be null)
int vae_normalize_rows_ex(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float* mean, float* xn,
                          int64_t ld_xn, float* absmax, void* stream);
// vae_gene_dropout that also raises *absmax to max|xn| of the masked, normalised counts
int vae_gene_dropout_ex(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float p, float alpha, uint64_t seed,
                        uint32_t step, float* running, int32_t* counts, uint32_t* threshold, float* stats, float* mean,
                        float* xn, int64_t ld_xn, float* absmax, void* stream);
// vae_nbzinb_bwd_overwrite / vae_nbzinb_bwd_u8(overwrite = 1) that also raises *absmax to max|dlogits|
int vae_nbzinb_bwd_ex(int x_dtype, int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                      const void* x, int64_t ld_x, const float* library, const float* gene_table, const float* rowmax,
                      const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B, int64_t G,
                      float* dlogit_scale, int64_t ld_ds, float* dlogit_dropout, int64_t ld_dp, float* dlibrary, float* dpx_r,
                      float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes, float* absmax,
                      void* stream);
