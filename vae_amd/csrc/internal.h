// Internal interfaces used by step.hip (not part of the public C ABI): variants of public entry points that also
// record max|output| for the two-f16-term GEMMs, and those GEMMs themselves.
#pragma once
#include <stdint.h>

// Operand maxima travel as arrays of partial maxima (common.h): the producer's waves / workgroups store, the GEMM reduces.
#define VAE_ABSMAX_PARTIALS 1024   /* entries vae_absmax_partials writes; scanning workgroups of vae_step_prologue */
#define VAE_ABSMAX_W_SPLIT 440     /* of which the first weight array (encoder, 7.7 M of 18 M elements) gets these */

// partial[0 .. VAE_ABSMAX_PARTIALS) = per-workgroup maxima of |X|
int vae_absmax_partials(const float* X, int64_t rows, int64_t cols, int64_t ld, float* partial, void* stream);
// part0[0 .. VAE_ABSMAX_W_SPLIT), part1[0 .. VAE_ABSMAX_PARTIALS - VAE_ABSMAX_W_SPLIT): partial maxima of two dense arrays,
// and table = vae_gene_table(px_r, G), in one launch (epilogue.hip)
int vae_step_prologue(const float* w0, int64_t n0, const float* w1, int64_t n1, float* part0, float* part1, const float* px_r,
                      int64_t G, float* table, void* stream);

// vae_gemm (slabs_out == null) / vae_gemm_slabs (slabs_out != null: partial slabs left in the workspace) on two scaled
// f16 terms per element and three matrix-core products; the power-of-two scales are derived inside the kernel from the
// n_max_a / n_max_b partial maxima of |A| / |B| that the producers of the operands stored
// Request for the softmax statistics of the output rows over columns [0, cols) of an NT product (decoder-head logits):
// when the shape runs the split kernel's 16-byte-store epilogue, the GEMM writes, per row and per wave-wide column span
// ("slot"), max and sum of exp(v - max) to m / s [slot][M] and sets slots > 0; otherwise slots = 0 and nothing is written
// (the caller then computes the statistics with its own pass).  capacity: floats available in each of m and s.
struct VaeGemmRowStats {
  float* m;
  float* s;
  int cols;
  int64_t capacity;
  int slots;     // out
};
// The COUNT operand of the two encoder products (A of the NT forward, B of the TN weight gradient) handed over as RAW
// counts instead of x / mean - 1 (the caller applies (x W^T) / mean - sum_g W and (dpre / mean)^T x - colsum(dpre) 1^T,
// modules.py:264): integer counts below 2048 are exact in ONE scaled f16 term, so that operand's lo plane and one of the
// three products fall away.  flags: n_flags floats (16-byte aligned) left by the kernel that read the counts
// (vae_row_mean_ex), any > 0 = some count needs its lo term -> the kernel runs the plain two-term loop instead (decided
// on the device, once per workgroup); flags == null: always the two-term loop.  row_sub (nullable, TN only): C[m, n] -= row_sub[m].
// mode 1, the OPTIMISTIC forward product (NT, slabs; vae_gemm_counts_optimistic_ok): no kernel has read the counts. They
// are taken as exact in one UNSCALED f16 term (true of integer counts below 2048) and checked while they pass: the
// workgroups leave per-slab row sums in xs_sum[slab][M] (and of B in ws_sum[slab][N]) and per-workgroup entries of xs_max (max |x|) and
// xs_flag (1 = some count was not exact that way, the product is void), the rest of the xs_n entries zeroed -- the same
// arrays vae_row_stats fills.  mode 2, the launch that must follow with the same operands and flags = xs_flag: it
// finishes mean_out[m] = sum_slabs xs_sum / K and wsum_out[n] = sum_slabs ws_sum = sum_k B[n, k], and repeats the product in
// the scaled form above only if a flag is set (its maxima: xs_max) -- every workgroup decides that for itself.
struct VaeGemmCounts {
  const float* flags;
  int n_flags;
  const float* row_sub;
  int mode = 0;
  float* xs_sum = nullptr;
  float* xs_max = nullptr;
  float* xs_flag = nullptr;
  int xs_n = 0;
  float* ws_sum = nullptr;   // per-slab row sums of B [slab][N]
  // != 0: the count operand (A of NT in mode 1, B of TN with flags) is the uint8 store itself (elements = bytes, rows 4-byte
  // aligned): it is expanded in registers on its way into the f16 plane.  The mode 2 launch never reads it (uint8 counts
  // are exact, its gate stays shut); shapes: vae_gemm_counts_optimistic_ok (NT), vae_gemm_counts_u8_tn_ok (TN).
  int x_u8 = 0;
  float* mean_out = nullptr;
  float* wsum_out = nullptr;
};
bool vae_gemm_counts_optimistic_ok(int64_t M, int64_t N, int64_t K, int n_stat);
bool vae_gemm_counts_u8_tn_ok(int64_t M, int64_t N, int64_t K);
int vae_gemm_f16x2(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* max_a, int n_max_a,
                   const float* B, int64_t ldb, const float* max_b, int n_max_b, float* C, int64_t ldc, const float* bias,
                   int accumulate, float* workspace, int64_t workspace_bytes, void* stream, int* slabs_out,
                   VaeGemmRowStats* stats = nullptr, const VaeGemmCounts* counts = nullptr);
// Per-row scales of the two-f16-term wgrad: row m's scale comes from max over c < chunks of
// (m < split ? max0[c * stride + m] : max1[c * stride + m - split])
struct VaeGemmRowScale {
  const float* max0;
  const float* max1;
  int64_t split;
  int chunks;
  int64_t stride;
};
bool vae_gemm_tile_pipeline_ok(int layout, int64_t M, int64_t N, int64_t K);
int vae_gemm_f16x2_rowscaled(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const VaeGemmRowScale* rsc, const float* B,
                             int64_t ldb, const float* max_b, int n_max_b, float* C, int64_t ldc, float* workspace,
                             int64_t workspace_bytes, void* stream);
int vae_gemm_rowstats(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, float* workspace, int64_t workspace_bytes,
                      VaeGemmRowStats* stats, void* stream);
// bf16-mode product with one operand or the output held as bf16 in HBM (io = VAE_IO_A_BF16 | VAE_IO_B_BF16 | VAE_IO_C_BF16,
// one bit; shapes: vae_gemm_bf16_storage_ok); n_slabs != null leaves fp32 split-K slabs in the workspace
int vae_gemm_bf16io(int layout, int io, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                    void* C, int64_t ldc, const float* bias, float* workspace, int64_t workspace_bytes, int* n_slabs,
                    VaeGemmRowStats* stats, void* stream);
// bf16 images of up to two dense fp32 arrays (s1 nullable), one launch (gemm.hip)
int vae_cast_bf16(const float* s0, void* d0, int64_t n0, const float* s1, void* d1, int64_t n1, void* stream);
// rowmax[b] / rowsum[b] from the `slots` per-slot statistics a GEMM left (epilogue.hip)
// Atomic-free column sums of the unfused backward: the row kernels leave per-workgroup partial sums, one launch adds them
#define VAE_COLPART_MAX 40
struct VaeColPartList {
  const float* part[VAE_COLPART_MAX]; float* out[VAE_COLPART_MAX];
  int blocks[VAE_COLPART_MAX], width[VAE_COLPART_MAX], block0[VAE_COLPART_MAX]; int64_t stride[VAE_COLPART_MAX];
  int count;
};
int vae_colpart_blocks(int64_t B);
int vae_colpart_reduce(VaeColPartList& z, void* stream);
int vae_gelu_ln_bwd_part(const float* dout, int64_t ld_dout, const float* pre, int64_t ld_pre, const float* mean,
                         const float* rstd, const float* gamma, int64_t B, int64_t W, float* dpre, int64_t ld_dpre,
                         float* dgamma, float* dbeta, float* dbias, float* part, int* blocks_out, uint16_t* dpre16, int64_t ld16,
                         void* stream);
int vae_gelu_ln_fwd_ex(const float* pre, int64_t ld_pre, const float* gamma, const float* beta, int64_t B, int64_t W, float* out,
                       int64_t ld_out, float* mean, float* rstd, uint16_t* out16, int64_t ld16, void* stream);
int vae_dropout_apply_ex(float* x, int64_t ld, int64_t B, int64_t W, float p, uint64_t seed, uint32_t step, uint32_t site,
                         uint16_t* x16, int64_t ld16, void* stream);
int vae_reparam_kl_bwd_part(const float* dz, int64_t ld_dz, const float* g_kl, float g_kl_scalar, const float* heads,
                            int64_t ld_heads, const float* eps, int64_t ld_eps, const float* qv, int64_t B, int64_t L,
                            float prior_mean, float prior_var, float* dheads, int64_t ld_dheads, float* dbias, float* part,
                            int* blocks_out, void* stream);

// px_r != null: extra workgroups of the same launch build the gene table (vae_gene_table) as well
int vae_softmax_stats_merge(const float* stat_m, const float* stat_s, int slots, int64_t B, float* rowmax, float* rowsum,
                            const float* px_r, int64_t G, float* table, void* stream);

// Entries of the partial-maxima arrays the producers below fill (all of them, every call)
#define VAE_NORM_WAVES 4           /* vae_normalize_rows_ex / vae_gene_dropout_ex: B * 4 entries      */
#define VAE_MID_WAVES 8            /* vae_mid_forward / vae_mid_backward: B * 8 entries               */
int64_t vae_nbzinb_bwd_workgroups(int64_t B, int64_t G);
int vae_nbzinb_bwd_chunks(int64_t B, int64_t G);           /* cell chunks = slabs [chunk][VAE_SLAB_ROWS][G] it leaves */   /* vae_nbzinb_bwd_ex: one entry per workgroup */

// vae_normalize_rows / _u8 (x_dtype) that also stores the per-wave maxima of |xn| (absmax may be null)
// xn_bf16 != 0: xn is written as bf16 (ld_xn in elements, % 8 == 0)
int vae_normalize_rows_ex(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float* mean, void* xn, int xn_bf16,
                          int64_t ld_xn, float* absmax, void* stream);
// (vae_row_stats: include/vae_amd.h)
// fp32 copy of a uint8 batch (G % 4 == 0, 4-byte aligned rows in, 16-byte aligned rows out)
int vae_cast_u8_f32(const void* x, int64_t ldx, int64_t B, int64_t G, float* out, int64_t ldo, void* stream);
// vae_gene_dropout that also stores the per-wave maxima of |xn| of the masked, normalised counts
int vae_gene_dropout_ex(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float p, float alpha, uint64_t seed,
                        uint32_t step, float* running, int32_t* counts, uint32_t* threshold, float* stats, float* mean,
                        void* xn, int xn_bf16, int64_t ld_xn, float* absmax, void* stream);
// The column reduction that vae_nbzinb_bwd_ex leaves to a later launch (defer != null: it fills this in instead of
// launching colreduce_kernel; the slabs in `workspace` must then stay untouched until that launch has run).
struct VaeDeferredColreduce {
  const float* slab;
  int chunks, G, nk;       // nk = 2 (NB) or 3 (ZINB) column vectors
  float* out[3];           // d px_r, d bias_scale, d bias_dropout (overwritten)
};

// The ELBO reduction left to the backward (vae_step_config.defer_elbo_reduce): `part` = the per-strip partials of the streaming
// likelihood forward ([strips][2][B], still in the workspace); every workgroup of vae_nbzinb_bwd_ex forms the S_b of its cells
// from them (its S argument may then be null) and one extra workgroup of the same launch writes ll / S / recon (nullable) and
// the five scalars of vae_elbo_reduce.
struct VaeElboFromParts {
  const float* part;
  int strips;
  float *ll, *S, *recon, *out5;
  const float *kl_z, *kl_l;
  float kl_weight, coeff;
};
int vae_nbzinb_fwd_strips(int64_t G);   /* strips of the streaming forward's partials */
// vae_nbzinb_bwd_overwrite / vae_nbzinb_bwd_u8(overwrite = 1) that also stores the per-workgroup maxima of |dlogits|
// l_dtype: storage of the logits AND of their gradients (VAE_L_F32 | VAE_L_BF16; leading dimensions in elements)
enum { VAE_L_F32 = 0, VAE_L_BF16 = 1 };
int vae_nbzinb_bwd_ex(int x_dtype, int l_dtype, int zinb, const void* logit_scale, int64_t ld_s, const void* logit_dropout,
                      int64_t ld_p, const void* x, int64_t ld_x, const float* library, const float* gene_table, const float* rowmax,
                      const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B, int64_t G,
                      void* dlogit_scale, int64_t ld_ds, void* dlogit_dropout, int64_t ld_dp, float* dlibrary, float* dpx_r,
                      float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes, float* absmax,
                      VaeDeferredColreduce* defer, void* stream, const VaeElboFromParts* elbo = nullptr);

// vae_nbzinb_fwd_ws (streaming forward) for the step: have_stats != 0 skips the statistics launch (rowmax / rowsum came
// from the decoder-head GEMM); ll == S == null leaves the per-strip partials ([*strips_out][2][B]) in the workspace for
// vae_elbo_reduce_parts, which adds them up in strip order while it forms the ELBO scalars (one launch less).
int vae_nbzinb_fwd_tiled_ex(int x_dtype, int l_dtype, int zinb, const void* logit_scale, int64_t ld_s, const void* logit_dropout,
                            int64_t ld_p, const void* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                            int64_t G, float* ll, float* rowmax, float* rowsum, float* S, int have_stats, float* workspace,
                            int64_t workspace_bytes, int* strips_out, void* stream);
// vae_px_rate over logits of either storage type
int vae_px_rate_ex(int l_dtype, const void* logit_scale, int64_t ld_s, const float* library, const float* rowmax,
                   const float* rowsum, int64_t B, int64_t G, float* px_rate, int64_t ld_r, void* stream);
int vae_nbzinb_fwd_finish(const float* part, int strips, int64_t B, float* ll, float* S, void* stream);
// vae_elbo_reduce whose ll[b] (and S[b]) are first formed from `strips` per-strip partials part[(s*2 + k)*B + b]
int vae_elbo_reduce_parts(const float* part, int strips, float* ll, float* S, const float* kl_z, const float* kl_l, int64_t B,
                          float kl_weight, float gene_loss_coeff, float* out5, float* recon, void* stream);
