/* vae_amd.h -- C ABI of libvae_amd.so: hand-written gfx950 (MI355X) kernels for the training
 * hot path of the scVI-style VAE of DiseaseNeuroGenomics/vae (fwd + bwd ELBO).
 *
 * The reference has no FFI layer: the path sits behind Python nn.Module / function
 * signatures (SURVEY.md par. 8b).  These entry points are what a binding for that path
 * attaches to; each comment names the reference code the call replaces
 * (paths relative to the reference's src/).  The reference-side stub is in INTEGRATION.md.
 *
 * Conventions
 *   - all matrices are fp32, row-major, with an explicit leading dimension (elements);
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates,
 *     frees or retains memory (scratch is passed in; sizes come from the *_workspace_bytes
 *     queries);
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work;
 *   - return value: 0 = ok, VAE_ERR_* otherwise, message via vae_last_error() (thread-local);
 *     no C++ exception crosses the boundary;
 *   - outputs documented as "accumulated" are added to (zero them for a plain result).
 */
#ifndef VAE_AMD_H
#define VAE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAE_OK 0
#define VAE_ERR_ARG 1
#define VAE_ERR_LAUNCH 2
#define VAE_ERR_UNSUPPORTED 3

/* layouts / precisions of vae_gemm */
#define VAE_NT 0 /* C[m,n] = sum_k A[m,k] B[n,k]   nn.Linear forward            */
#define VAE_NN 1 /* C[m,n] = sum_k A[m,k] B[k,n]   autograd dgrad               */
#define VAE_TN 2 /* C[m,n] = sum_k A[k,m] B[k,n]   autograd wgrad               */
#define VAE_FP32 0 /* fp32-grade: three bf16 terms per element, six products (aligned shapes), else the fp32 MFMA */
#define VAE_BF16 1 /* operands rounded to bf16, fp32 accumulate (..32x32x16_bf16) */
#define VAE_BF16X3 2 /* vae_gemm only: the three-term split on the generic (ragged-shape) kernel         */
#define VAE_FP32_F16X2 3 /* fp32-grade normwise: two scaled f16 terms, three products (vae_gemm_scaled); as a step
                          * precision: the five gene-sized products use it, the operand maxima come from the
                          * kernels that produce the operands; every other product as VAE_FP32            */

enum { VAE_X_F32 = 0, VAE_X_U8 = 1 };   /* dtype of a count matrix */

/* ---- library ---------------------------------------------------------------------------- */
int vae_version(void);             /* 100 * major + minor */
const char* vae_arch(void);        /* "gfx950" */
const char* vae_last_error(void);  /* message of the last failing call on this thread */
int vae_device_count(void);        /* GPUs visible to HIP; 0 on a CPU-only host, never fails */

/* ---- GEMM: nn.Linear forward / backward (modules.py:111,248-249,361-369 + autograd) ------ */
/* x_mean (nullable): per-cell mean; normalises the count operand on the fly, x/mean - 1
 * (Encoder.forward, modules.py:264): rows of A for VAE_NT, rows (k) of B for VAE_TN.
 * bias (nullable): [N], added to every row.  accumulate != 0: C += result.
 * workspace: >= vae_gemm_workspace_bytes(M,N,K) bytes, 16-byte aligned (may be NULL if 0). */
int64_t vae_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int vae_gemm(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
             int64_t ldb, float* C, int64_t ldc, const float* bias, const float* x_mean, int accumulate,
             float* workspace, int64_t workspace_bytes, void* stream);

/* vae_gemm for callers that cannot allocate device memory or bring a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.  Same arguments and results (bit
 * for bit) as vae_gemm without the workspace pair: the split-K slabs go to a static device array of
 * vae_gemm_noalloc_max_bytes() = 40 MiB that every GPU the library runs on holds until the process ends; shapes whose
 * vae_gemm_workspace_bytes exceeds it are rejected with VAE_ERR_ARG.  Calls no hipMalloc / hipFree.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU (different streams, or threads) race -- serialise them.
 * The per-device address of the array is looked up on the first call on a device (hipGetSymbolAddress, a runtime query
 * that must not happen inside a graph capture): call vae_gemm_noalloc_init() once per device, with that device
 * current, before capturing.  `stream` must belong to the current device.                                          */
int64_t vae_gemm_noalloc_max_bytes(void);
int vae_gemm_noalloc_init(void);
int vae_gemm_noalloc(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                     const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, const float* x_mean,
                     int accumulate, void* stream);

/* The same product left as *n_slabs partial (M,N) slabs (row stride N, slab stride M*N) in the
 * workspace: no reduce pass, no bias.  For a consumer that sums the split-K slabs while it reads
 * them (the fused encoder-tail / decoder-head kernels of vae_elbo_step_*).                       */
int64_t vae_gemm_slabs_workspace_bytes(int64_t M, int64_t N, int64_t K);
int vae_gemm_slabs(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                   const float* B, int64_t ldb, float* workspace, int64_t workspace_bytes, int* n_slabs, void* stream);

/* The same products on TWO f16 terms per operand element and three matrix-core products (half the matrix
 * work of vae_gemm's fp32 mode, two LDS planes instead of three): element a is first multiplied by a power of
 * two *scale (device pointer; vae_absmax_scale computes it from max|A| so that the scaled operand fills the
 * upper part of f16's range), then split as hi = rn_f16(a s), lo = rn_f16(a s - hi) -- 22 significant bits -- and
 * hi*hi + hi*lo + lo*hi accumulate in fp32; the result is scaled back exactly.  The error of a product is below
 * 2^-21 of max|A| max|B| terms, i.e. NORMWISE fp32-grade for operands whose large entries matter (every operand of
 * the training step); an element 2^-12 below the matrix maximum keeps fewer bits (vae_gemm keeps all).
 * Shapes outside the split pipeline (small or unaligned) run the exact fp32 kernel.                     */
int64_t vae_absmax_workspace_bytes(void);
int vae_absmax_scale(const float* X, int64_t rows, int64_t cols, int64_t ld, float* scale_out, float* workspace,
                     int64_t workspace_bytes, void* stream);
int vae_gemm_scaled(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* scale_a,
                    const float* B, int64_t ldb, const float* scale_b, float* C, int64_t ldc, const float* bias,
                    int accumulate, float* workspace, int64_t workspace_bytes, void* stream);
int vae_gemm_scaled_slabs(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* scale_a,
                          const float* B, int64_t ldb, const float* scale_b, float* workspace, int64_t workspace_bytes,
                          int* n_slabs, void* stream);

/* The count-operand form of the encoder's first layer (Encoder.forward, modules.py:264,267: Linear over x / mean - 1).
 * x / mean - 1 is never materialised:
 *   forward          (x / m - 1) W^T = (x W^T) / m - sum_g W[:, g]
 *   weight gradient  dpre^T (x / m - 1) = (dpre / m)^T x - colsum(dpre) 1^T
 * so both gene-sized products stream the RAW fp32 counts, which are exact in ONE scaled f16 term when they are integers
 * below 2048 (single-cell counts; the reference stores them as uint8, create_dataset.py:131-142): that operand's lo
 * term, its LDS plane and one of the three matrix-core products fall away.
 * vae_row_stats: mean[b] of row b of x (B, G), per-wave maxima of |x| and per-wave flags "some count needs its lo term"
 *   (4 entries per row each: absmax / flags hold B * 4 floats, 16-byte aligned), and wsum[h] = sum_g W[h, g] of the (HE, G)
 *   weight, one launch, fixed summation order.
 * vae_gemm_counts: vae_gemm on two f16 terms (as vae_gemm_scaled, the scales derived from n_max partial maxima per
 *   operand) where the count operand is A for VAE_NT and B for VAE_TN.  VAE_NT: when every flag is 0 the counts run on their
 *   hi term alone, otherwise the kernel takes the plain two-term loop (decided on the device).  VAE_TN: the same on
 *   128 x 128 tiles when flags are given (NULL: always two terms).  row_sub (nullable, VAE_TN, M floats): C[m, n] -= row_sub[m].  Shapes outside the 128-wide tile pipeline are rejected (vae_gemm_rowscaled_ok's conditions,
 *   with the layout's own alignment: K % 4 == 0, TN: M % 4 == N % 4 == 0).                                              */
int vae_row_stats(const float* x, int64_t ldx, int64_t B, int64_t G, float* mean, float* absmax, float* flags, const float* W,
                  int64_t ldw, int64_t HE, float* wsum, void* stream);
int vae_gemm_counts(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* max_a, int64_t n_max_a,
                    const float* B, int64_t ldb, const float* max_b, int64_t n_max_b, float* C, int64_t ldc, const float* flags,
                    int64_t n_flags, const float* row_sub, float* workspace, int64_t workspace_bytes, void* stream);

/* The weight-gradient product C (M, N) = A^T B (A stored (K, M), B stored (K, N): layout VAE_TN; dW = dy^T x of an
 * nn.Linear, modules.py:361-369) on two f16 terms with ONE POWER-OF-TWO SCALE PER OUTPUT ROW m instead of one for all of A:
 * the decoder-head weight gradient of the training step, where column m of A = dlogits is gene m's gradient and its
 * magnitude follows the gene's softmax mass (> 2^20 across genes), so that a single scale would leave quiet genes' rows
 * with few correct bits.  col_max: (n_chunks, ld_max >= M) maxima of |A[k, m]| over chunks of the K rows (any partition;
 * the step's backward epilogue leaves one row per cell chunk); row m's scale comes from their maximum and is removed
 * from the output row again -- both exact.  b_max: n_b_max > 0 partial maxima of |B| (16-byte aligned).  Needs a shape
 * on the 128-wide tile pipeline (vae_gemm_rowscaled_ok: M, N >= 4, M % 4 == N % 4 == K % 4 == 0, enough work for 128-wide
 * tiles) and 16-byte aligned rows; VAE_ERR_ARG otherwise (use vae_gemm).                                                  */
int vae_gemm_rowscaled_ok(int64_t M, int64_t N, int64_t K);
int vae_gemm_rowscaled(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* col_max, int64_t n_chunks,
                       int64_t ld_max, const float* B, int64_t ldb, const float* b_max, int64_t n_b_max, float* C, int64_t ldc,
                       float* workspace, int64_t workspace_bytes, void* stream);

/* VAE_BF16 products with operands and / or the output stored as bf16 in HBM (io = VAE_IO_* bits; the other matrices
 * fp32).  This is the reference's autocast(bfloat16) storage (config.py:87: nn.Linear outputs are bf16 tensors,
 * and what autograd saves / hands to the wgrad and dgrad products is bf16): the bytes of the three gene-sized
 * activations (normalised counts, logits, logit gradients) halve.  Rounding an operand to bf16 when it is WRITTEN is
 * bit-identical to rounding it when the product reads it (vae_gemm(VAE_BF16)); a bf16 OUTPUT is the fp32 accumulator + bias
 * rounded to nearest even.  Leading dimensions count elements of the matrix's own type; bf16 matrices need 8-byte
 * aligned rows (the output: 16-byte, ldc % 8 == 0).  Shapes: vae_gemm_bf16_storage_ok() != 0 (the 128-wide tile pipeline:
 * no dimension <= 64, K % 4 == 0, 16-byte rows); combinations: VAE_NT {A, A|B, C, A|B|C}, VAE_NN {A, A|B}, VAE_TN {A, B, A|B}
 * -- what a bf16-mode training step uses (it keeps bf16 images of the gene-sized weights and of the small activations). */
enum { VAE_IO_A_BF16 = 1, VAE_IO_B_BF16 = 2, VAE_IO_C_BF16 = 4 };
int vae_gemm_bf16_storage_ok(int layout, int io, int64_t M, int64_t N, int64_t K);
int vae_gemm_bf16_storage(int layout, int io, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                          int64_t ldb, void* C, int64_t ldc, const float* bias, float* workspace, int64_t workspace_bytes,
                          void* stream);

/* ---- row kernels ----------------------------------------------------------------------------- */
/* mean[b] = mean_g x[b,g]                                   Encoder.forward, modules.py:264      */
int vae_row_mean(const float* x, int64_t ldx, int64_t B, int64_t G, float* mean, void* stream);
/* mean[b] as above and xn[b,g] = x[b,g] / mean[b] - 1 (IEEE division, then the subtraction, as
 * torch evaluates it)                                         Encoder.forward, modules.py:264      */
int vae_normalize_rows(const float* x, int64_t ldx, int64_t B, int64_t G, float* mean, float* xn, int64_t ld_xn,
                       void* stream);
/* the same for counts kept as uint8, the reference's on-disk format (create_dataset.py:131-142,
 * data.py:406-415): the uint8 -> fp32 expansion of the loader happens in the kernel */
int vae_normalize_rows_u8(const uint8_t* x, int64_t ldx, int64_t B, int64_t G, float* mean, float* xn, int64_t ld_xn,
                          void* stream);

/* out = LayerNorm(GELU(pre)) * gamma + beta, eps 1e-5; saves per-row mean / rstd of GELU(pre)
 *                                                           FCLayers, modules.py:112-114          */
int vae_gelu_ln_fwd(const float* pre, int64_t ld_pre, const float* gamma, const float* beta, int64_t B, int64_t W,
                    float* out, int64_t ld_out, float* mean, float* rstd, void* stream);
/* backward of the above; dpre may alias dout.  dgamma, dbeta, dbias (= column sums of dpre, the
 * bias gradient of the Linear that produced `pre`) are accumulated (each nullable).              */
int vae_gelu_ln_bwd(const float* dout, int64_t ld_dout, const float* pre, int64_t ld_pre, const float* mean,
                    const float* rstd, const float* gamma, int64_t B, int64_t W, float* dpre, int64_t ld_dpre,
                    float* dgamma, float* dbeta, float* dbias, void* stream);

/* fused reparameterisation sample + Gaussian KL.  heads[b,0:L] = q_m, heads[b,L] = pre-activation
 * of the ONE shared variance:  q_v = softplus(.)+1e-4, z = q_m + eps*sqrt(q_v),
 * kl[b] = sum_j KL(N(q_m_j, sqrt q_v) || N(prior_mean, sqrt prior_var))
 *                        Encoder.forward modules.py:268-270, :11-12; LitVAE.gene_loss task.py:188-198 */
int vae_reparam_kl_fwd(const float* heads, int64_t ld_heads, const float* eps, int64_t ld_eps, int64_t B, int64_t L,
                       float prior_mean, float prior_var, float* qv, float* z, int64_t ld_z, float* kl, void* stream);
/* dheads (B, L+1) from dz (B, L) and the upstream gradient on kl[b] = g_kl_scalar * g_kl[b]
 * (g_kl nullable = 1).  dbias (nullable, [L+1]): column sums of dheads are accumulated = the bias
 * gradient of the [mean_encoder; var_encoder] heads.  L <= 64.                                     */
int vae_reparam_kl_bwd(const float* dz, int64_t ld_dz, const float* g_kl, float g_kl_scalar, const float* heads,
                       int64_t ld_heads, const float* eps, int64_t ld_eps, const float* qv, int64_t B, int64_t L,
                       float prior_mean, float prior_var, float* dheads, int64_t ld_dheads, float* dbias,
                       void* stream);

/* dst[i] += src[i]  (adds the cell-property heads' latent gradient, task.py:469-471, to dz)         */
int vae_add_inplace(const float* src, float* dst, int64_t n, void* stream);

/* out[n] += sum_m in[m,n]   (bias gradients of the small Linears)                                  */
int vae_colsum(const float* in, int64_t ld, int64_t M, int64_t N, float* out, void* stream);

/* out5 = { mean(-ll), mean(kl_z), mean(kl_l), gene_loss, gene_loss_coeff*gene_loss },
 * gene_loss = mean(-ll) + kl_weight*mean(kl_l + kl_z)      LitVAE.gene_loss task.py:202, :489
 * recon (nullable, [B]) = -ll, the per-cell reconstruction loss of task.py:155-165              */
int vae_elbo_reduce(const float* ll, const float* kl_z, const float* kl_l, int64_t B, float kl_weight,
                    float gene_loss_coeff, float* out5, float* recon, void* stream);

/* ---- fused decoder epilogue ---------------------------------------------------------------------
 * softmax over genes (modules.py:362,401), px_rate = exp(library)*px_scale (:404),
 * theta = exp(px_r) (task.py:481-483), NB / ZINB log-likelihood (distributions.py:22-62, :65-90),
 * sum over genes (task.py:156-165) -- and its backward.                                           */
/* px_rate[b,g] = exp(library[b]) * exp(l[b,g] - rowmax[b]) / rowsum[b]  (modules.py:401,404), with the
 * rowmax / rowsum vae_nbzinb_fwd saved: exactly the mean the likelihood was evaluated at.              */
int vae_px_rate(const float* logit_scale, int64_t ld_s, const float* library, const float* rowmax, const float* rowsum,
                int64_t B, int64_t G, float* px_rate, int64_t ld_r, void* stream);
int64_t vae_gene_table_rows(void);
/* per-gene constants of the epilogue, (vae_gene_table_rows(), G); rebuild whenever px_r changes   */
int vae_gene_table(const float* px_r, int64_t G, float* table, void* stream);
int64_t vae_nbzinb_max_genes(void);   /* limit of vae_nbzinb_fwd / _u8 (a row parked in LDS): 32768; _ws and the step: none */
/* ll[b] = sum_g log p(x[b,g]); rowmax / rowsum / S are what the backward needs (each [B])         */
int vae_nbzinb_fwd(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                   const float* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B, int64_t G,
                   float* ll, float* rowmax, float* rowsum, float* S, void* stream);
/* The same results by a STREAMING pass (what vae_elbo_step_forward runs): one launch computes the softmax statistics of
 * every row, then (strip of 1024 genes) x (chunk of cells) workgroups evaluate the likelihood with loads and arithmetic
 * overlapped; per-strip partial sums of ll / S go through `workspace` (>= vae_nbzinb_fwd_workspace_bytes) and are added
 * in strip order (deterministic).  x: float (VAE_X_F32) or uint8 (VAE_X_U8) counts.  have_stats != 0: rowmax / rowsum
 * are INPUTS (max_g and sum_g exp(l - max) of every row, e.g. from the producer of the logits) and the statistics
 * launch is skipped -- the step gets them from the decoder-head GEMM's epilogue.  ll == S == NULL: the final sum over
 * strips is left to the caller: workspace[(s * 2 + k) * B + b], s < ceil(G / 1024), holds strip s's share of ll[b] (k = 0)
 * and S[b] (k = 1) -- the step adds them up inside its ELBO reduction.                                                */
int64_t vae_nbzinb_fwd_workspace_bytes(int64_t B, int64_t G);
int vae_nbzinb_fwd_ws(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                      const void* x, int x_dtype, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                      int64_t G, float* ll, float* rowmax, float* rowsum, float* S, int have_stats, float* workspace,
                      int64_t workspace_bytes, void* stream);
/* uint8 counts (needs G % 4 == 0, ld % 4 == 0, 4-byte aligned rows)                               */
int vae_nbzinb_fwd_u8(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                   const uint8_t* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B, int64_t G,
                   float* ll, float* rowmax, float* rowsum, float* S, void* stream);
/* gradient of sum_b g_scalar*g_ll[b]*ll[b] (g_ll nullable = 1).  dlogit_* may alias logit_*.
 * dlibrary [B] is written; dpx_r, dbias_scale, dbias_dropout [G] are accumulated -- in a fixed
 * order through `workspace` (>= vae_nbzinb_bwd_workspace_bytes) or, if workspace is NULL, with
 * float atomics.                                                                                   */
int64_t vae_nbzinb_bwd_workspace_bytes(int64_t B, int64_t G);
int vae_nbzinb_bwd(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                   const float* x, int64_t ld_x, const float* library, const float* gene_table, const float* rowmax,
                   const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B, int64_t G,
                   float* dlogit_scale, int64_t ld_ds, float* dlogit_dropout, int64_t ld_dp, float* dlibrary,
                   float* dpx_r, float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes,
                   void* stream);
/* the same with dpx_r, dbias_scale, dbias_dropout OVERWRITTEN (workspace required): no zero-fill
 * of the gradient arrays is needed before the call                                              */
int vae_nbzinb_bwd_overwrite(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                   const float* x, int64_t ld_x, const float* library, const float* gene_table, const float* rowmax,
                   const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B, int64_t G,
                   float* dlogit_scale, int64_t ld_ds, float* dlogit_dropout, int64_t ld_dp, float* dlibrary,
                   float* dpx_r, float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes,
                   void* stream);
/* uint8 counts; overwrite != 0: as vae_nbzinb_bwd_overwrite                                        */
int vae_nbzinb_bwd_u8(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                   const uint8_t* x, int64_t ld_x, const float* library, const float* gene_table, const float* rowmax,
                   const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B, int64_t G,
                   float* dlogit_scale, int64_t ld_ds, float* dlogit_dropout, int64_t ld_dp, float* dlibrary,
                   float* dpx_r, float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes,
                   int overwrite, void* stream);

/* The streaming forward and the backward over logits stored as bf16 (uint16_t = the upper half of the fp32 pattern;
 * leading dimensions in elements; needs G % 4 == 0, ld % 4 == 0, 8-byte aligned rows) -- what a VAE_BF16 step runs on the
 * logits its decoder-head product wrote (reference: nn.Linear outputs under autocast(bfloat16), config.py:87).  The
 * gradients are written as bf16 (nearest even; may alias the logits); dpx_r / dbias_* are OVERWRITTEN (workspace required);
 * the arithmetic is the fp32 arithmetic of the entry points above.                                                    */
int vae_nbzinb_fwd_ws_bf16(int zinb, const uint16_t* logit_scale, int64_t ld_s, const uint16_t* logit_dropout, int64_t ld_p,
                           const void* x, int x_dtype, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                           int64_t G, float* ll, float* rowmax, float* rowsum, float* S, int have_stats, float* workspace,
                           int64_t workspace_bytes, void* stream);
int vae_nbzinb_bwd_bf16(int zinb, const uint16_t* logit_scale, int64_t ld_s, const uint16_t* logit_dropout, int64_t ld_p,
                        const void* x, int x_dtype, int64_t ld_x, const float* library, const float* gene_table,
                        const float* rowmax, const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B,
                        int64_t G, uint16_t* dlogit_scale, int64_t ld_ds, uint16_t* dlogit_dropout, int64_t ld_dp, float* dlibrary,
                        float* dpx_r, float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes,
                        void* stream);

/* ---- stand-alone pieces behind the module API ----------------------------------------------------
 * (px_scale, px_rate) of DecoderSCVI.forward, modules.py:401-404, both (B,G) contiguous           */
int vae_softmax_rate_fwd(const float* logit_scale, int64_t ld_s, const float* library, int64_t B, int64_t G,
                         float* px_scale, float* px_rate, void* stream);
int vae_softmax_rate_bwd(const float* px_scale, const float* library, const float* d_scale, const float* d_rate,
                         int64_t B, int64_t G, float* dlogits, int64_t ld_d, float* dlibrary, void* stream);
/* Poisson log-probability, element-wise (reconstruction_loss "poisson", task.py:167-168;
 * torch.distributions.Poisson.log_prob = xlogy(x, rate) - rate - lgamma(x + 1)) and its gradient
 * drate = gout * (x / rate - 1).                                                                    */
int vae_poisson_fwd(const float* x, int64_t ld_x, const float* rate, int64_t ld_rate, int64_t B, int64_t G, float* out,
                    int64_t ld_out, void* stream);
int vae_poisson_bwd(const float* gout, int64_t ld_g, const float* x, int64_t ld_x, const float* rate, int64_t ld_rate,
                    int64_t B, int64_t G, float* drate, int64_t ld_d, void* stream);

/* ---- input side: batches cut out of a uint8 cells x genes store held in HBM ----
 * vae_gather_counts replaces SingleCellDataset._get_gene_vals_batch (data.py:406-415): out[b, g] =
 * store[cell_rows[b] * ld_store + (gene_idx ? gene_idx[g] : g)] for b < B, g < G, written as uint8 (VAE_X_U8: what the
 * fused step reads directly) or fp32 (VAE_X_F32).  store: device, n_cells rows of ld_store bytes
 * (create_dataset.py:131-142); cell_rows: device int64 (B); gene_idx: device int32 (G) or null (the first G genes).
 * Rows / genes outside the store give zeros (validate on the host); B <= 65535.                                      */
int vae_gather_counts(const uint8_t* store, int64_t n_cells, int64_t ld_store, const int64_t* cell_rows, int64_t B,
                      const int32_t* gene_idx, int64_t G, void* out, int out_dtype, int64_t ld_out, void* stream);
/* vae_mixup_counts replaces the array arithmetic of SingleCellDataset._mixup (data.py:457-477):
 * out[i, :] = alpha[i] * x[i, :] + (1 - alpha[i]) * x[partner[i], :], evaluated as numpy does (float64 products and
 * sum without fused multiply-add, rounded once to fp32: bit-identical).  x: (B, G) uint8 or fp32 (x_dtype), also
 * used for the (B, P) cell-property values and masks; partner: device int32 (B) in [0, B); alpha: device double (B);
 * out must not alias x.                                                                                              */
int vae_mixup_counts(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, const int32_t* partner,
                     const double* alpha, float* out, int64_t ld_out, void* stream);
/* vae_cutmix_counts replaces the array part of SingleCellDataset._cutmix (data.py:431-454): out[i, g] =
 * take[i, g] ? x[partner[i], g] : x[i, g].  take: device (B, G) bytes, nonzero = gene g of cell i comes from the partner
 * (the host replays the reference's np.random.choice(n_genes, int(alpha_i n_genes), replace=False) per cell and marks
 * the subset); x: uint8 or fp32, out: uint8 (uint8 input only) or fp32 -- counts are copied, never blended, so uint8 stays
 * exact.  The (B, P) label / mask tables are blended with alpha as in mixup: vae_mixup_counts.  out must not alias x.  */
int vae_cutmix_counts(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, const int32_t* partner,
                      const uint8_t* take, int64_t ld_take, void* out, int out_dtype, int64_t ld_out, void* stream);

/* element-wise log_nb_positive / log_zinb_positive (distributions.py:65-90 / :22-62), eps = 1e-8;
 * theta addressed as theta[b*ld_theta + g] (ld_theta = 0 broadcasts a (G,) vector)                */
int vae_loglik_elem_fwd(int zinb, const float* x, int64_t ld_x, const float* mu, int64_t ld_mu, const float* theta,
                        int64_t ld_theta, const float* pi, int64_t ld_pi, int64_t B, int64_t G, float* out,
                        int64_t ld_out, void* stream);
int vae_loglik_elem_bwd(int zinb, const float* gout, int64_t ld_g, const float* x, int64_t ld_x, const float* mu,
                        int64_t ld_mu, const float* theta, int64_t ld_theta, const float* pi, int64_t ld_pi, int64_t B,
                        int64_t G, float* dmu, float* dtheta, float* dpi, void* stream);


/* ---- dropout ------------------------------------------------------------------------------------
 * nn.Dropout of FCLayers (modules.py:115): x[b, c] *= keep / (1 - p).  torch's generator cannot be
 * reproduced, so the stream is this library's own and documented: Philox4x32-10 with
 *   key = (seed low 32, seed high 32), counter = (elem low 32, elem high 32, site, step),
 *   elem = b * W + c,   keep  <=>  output word 0 >= floor(p * 2^32).
 * site: 4 * layer + 0 = z-encoder hidden layer, 4 * layer + 2 = decoder hidden layer (the library encoder
 * is built without dropout, networks.py:215-222).  The step (vae_elbo_step_*) draws the same masks from cfg->rng_seed / rng_step in the
 * forward and the backward pass; tests regenerate them on the host (tests/helpers.py).               */
int vae_dropout_apply(float* x, int64_t ld, int64_t B, int64_t W, float p, uint64_t seed, uint32_t step, uint32_t site,
                      void* stream);

/* GeneDropout on the z-encoder's input (modules.py:190-209,263) followed by the row normalisation
 * (modules.py:264), training mode:  running = alpha*running + (1-alpha)*mean_b(x > 0);
 * p_g = clip(p * running_g / mean(running), 0, 0.9);  gene g of cell b is KEPT with probability p_g;
 * xn = xm / mean_g(xm) - 1 with xm = x * keep  (the reference's extra division by mean_g(p_g) is a per-batch
 * scale that this normalisation cancels).  Mask stream: Philox4x32-10 as above with site 1 and counter element
 * b * (G/4) + g/4; gene g is kept iff output word g%4 < threshold[g] = floor(p_g * 2^32).
 * running (G, in/out) is the module's state; counts (G int32) and threshold (G uint32) are scratch / output;
 * stats[0] = mean_g(p_g).  x_dtype: VAE_X_F32 / VAE_X_U8.  Needs G % 4 == 0.                              */
int vae_gene_dropout(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float p, float alpha, uint64_t seed,
                     uint32_t step, float* running, int32_t* counts, uint32_t* threshold, float* stats, float* mean,
                     float* xn, int64_t ld_xn, void* stream);

/* ---- the whole training step -------------------------------------------------------------------
 * LitVAE.training_step (task.py:453-502) for dispersion "gene":
 * one call enqueues the forward, one the backward; all intermediates live in `workspace`.
 * Parameter / gradient pointers follow the reference's state_dict (SURVEY.md par. 3.4); tensors the
 * kernels address as one stacked matrix must be adjacent in memory (vae_amd/arena.py):
 *   enc_w0 = [z_encoder.encoder.fc_layers.0.weight ; l_encoder.encoder.fc_layers.0.weight] (H+Hl, G)
 *   hz_w   = [z_encoder.mean_encoder.weight ; z_encoder.var_encoder.weight]                 (L+1, H)
 *   hl_w   = [l_encoder.mean_encoder.weight ; l_encoder.var_encoder.weight]                 (2, Hl)
 *   heads_w= [decoder.px_scale_decoder.0.weight ; decoder.px_dropout_decoder.weight]        (2G, Hd)
 *            (NB: the first G rows only), and the matching bias vectors.                          */
#define VAE_MAX_LAYERS 4

typedef struct {
  int64_t B, G, H, Hl, L, Hd;
  int32_t n_layers;  /* 1..VAE_MAX_LAYERS (z-encoder and decoder; the library encoder has 1)        */
  int32_t zinb;      /* 1 = ZINB, 0 = NB                                                          */
  int32_t precision; /* VAE_FP32 / VAE_BF16 / VAE_FP32_F16X2                                        */
  float l_mean, l_var;        /* prior of the library-size KL (task.py:230-231)                     */
  float kl_weight;            /* task.py:213-220                                                    */
  float gene_loss_coeff;      /* task.py:489                                                        */
  float loss_scale;           /* multiplies every gradient (1 for plain training)                   */
  int32_t x_dtype;            /* VAE_X_F32: counts are float; VAE_X_U8: uint8 (the reference's store)  */
  float dropout_rate;         /* p of nn.Dropout after the z-encoder / decoder LayerNorms (modules.py:115); 0 = off */
  uint32_t rng_step;          /* step counter of the dropout stream: change it every step               */
  uint64_t rng_seed;          /* key of the dropout stream (give every rank its own)                     */
  float input_dropout_rate;   /* p of GeneDropout on the z-encoder input (modules.py:190-209); 0 = off   */
  float input_dropout_alpha;  /* its running-average factor (0.95 in the reference)                      */
  float* gene_running;        /* (G) device: GeneDropout.running_non_zero_means, updated by the forward  */
  /* VAE_FP32_F16X2 only, optional: partial maxima of |enc_w0| and |heads_w| (weight_max_n floats each, 16-byte aligned,
   * every entry >= 0 and their maximum = the matrix's largest magnitude) kept up to date by whoever last changed the
   * weights -- vae_adamw_step_wmax writes them while it stores the updated weights.  With both set the forward skips its
   * own scan of the two weight matrices (72 MB at 20k genes).  NULL: the step scans.                              */
  const float* weight_max_enc;
  const float* weight_max_heads;
  int32_t weight_max_n;
  /* Batch covariates (FCLayers.forward, modules.py:121-130; networks.py:187,218,229): the reference appends the masked
   * one-hot block cov = [one_hot(label_p) * mask_p for every batch property p] (B, n_cov) to the input of the FIRST
   * layer of both encoders (after the count normalisation, modules.py:264,267) and of the decoder.  With n_cov > 0
   *   enc_w0 rows hold [G count columns | n_cov covariate columns | padding] with row stride ld_enc_w (0 = G + n_cov),
   *   dec_w[0] is (Hd, L + n_cov) contiguous, in params and grads alike;
   * the gene-wide products keep running on the count columns, the covariate columns add two small products
   * (cov W_c^T forward, dpre^T cov for the gradient).  Padding columns are neither read nor written.  n_cov = 0: none. */
  int32_t n_cov;
  const float* batch_cov;     /* (B, n_cov) device, row stride ld_cov                                    */
  int64_t ld_cov;
  int64_t ld_enc_w;
  /* VAE_BF16 steps that keep bf16 storage (vae_elbo_step_bf16_storage), optional: bf16 images (nearest even) of enc_w0 --
   * all (H + Hl) * ld_enc_w elements, padding included -- and of heads_w, 16-byte aligned, kept current by whoever
   * updates the weights: vae_adamw_step_bf16 writes them while it stores the updated weights.  With both set the
   * forward skips its own cast of the two matrices (one launch, 6 bytes per weight) and every product streams these. */
  const void* weight_bf16_enc;
  const void* weight_bf16_heads;
  /* != 0: the caller ALWAYS follows vae_elbo_step_forward with vae_elbo_step_backward on the same workspace and
   * configuration (a training step).  The forward then leaves the likelihood's per-strip partial sums in the workspace and
   * launches no ELBO reduction; the backward's first launch forms S_b from them and, in one extra workgroup, writes
   * VAE_WS_LL / VAE_WS_RECON / VAE_WS_SCALARS (task.py:202,489) -- the same values, one launch less on the step's
   * critical path (the sums over cells run in a different, equally fixed order).  Those three slices are valid only once
   * the backward has been enqueued.  Batches above 1024 cells ignore it.  0: the forward is complete on its own
   * (validation, vae_elbo_step_px_rate).                                                                            */
  int32_t defer_elbo_reduce;
  /* != 0: the caller expects float counts that are NOT exact in one unscaled f16 term (mixup / cutmix batches,
   * data.py:431-477, hold fractions): the first encoder product then runs behind a vae_row_stats launch as
   * vae_elbo_step_counts_path's form 1 does, instead of form 2, whose optimistic product would be repeated by its gated
   * launch every step (same results either way; about 8 us per step at B = 512, G = 20000).                       */
  int32_t fractional_counts;
} vae_step_config;

typedef struct {
  float* px_r;                                   /* (G)                                              */
  float *enc_w0, *enc_b0;                        /* (H+Hl, G) [batch covariates: see vae_step_config], (H+Hl) */
  float *encz_ln_g[VAE_MAX_LAYERS], *encz_ln_b[VAE_MAX_LAYERS]; /* z-encoder LayerNorm per layer (H)  */
  float *encz_w[VAE_MAX_LAYERS], *encz_b[VAE_MAX_LAYERS];       /* layers >= 1: (H,H), (H); [0] unused */
  float *encl_ln_g, *encl_ln_b;                  /* library-encoder LayerNorm (Hl)                   */
  float *hz_w, *hz_b, *hl_w, *hl_b;              /* (L+1,H), (L+1), (2,Hl), (2)                      */
  float *dec_w[VAE_MAX_LAYERS], *dec_b[VAE_MAX_LAYERS];         /* (Hd, L + n_cov) / (Hd, Hd), (Hd)  */
  float *dec_ln_g[VAE_MAX_LAYERS], *dec_ln_b[VAE_MAX_LAYERS];   /* (Hd)                              */
  float *heads_w, *heads_b;                      /* (2G or G, Hd), (2G or G)                         */
} vae_step_tensors;

/* != 0 when a step of this configuration keeps its gene-sized activations (the decoder-head logits and their
 * gradients) as bf16 in the workspace: precision == VAE_BF16 and shapes on the tile pipeline (G % 8 == 0, B > 64, ...).
 * This is the reference's autocast(bfloat16) storage (config.py:87); the parity tests pick the oracle definition with
 * it (oracle/vae_oracle.py, logits_bf16).  Environment VAE_BF16_F32_STORAGE=1 switches it off (A/B measurements). */
int vae_elbo_step_bf16_activations(const vae_step_config* cfg);
/* How a step of this configuration feeds the counts to the two first-layer encoder products (modules.py:264,267), given
 * 16-byte aligned rows of float counts: 0 = a normalised x / mean - 1 matrix in the workspace (vae_normalize_rows);
 * 1 = the raw counts as the GEMM operand behind one vae_row_stats launch; 2 = the raw counts with NO launch ahead -- the
 * forward product takes them as exact in one unscaled f16 term (integer counts below 2048 are), gathers the cells' sums,
 * maxima and exactness flags while they pass, and a gated launch behind it finishes the means and sum_g W and repeats the
 * product in the scaled two-term form only when a flag came up.  uint8 counts (x_dtype = VAE_X_U8, 4-byte aligned rows)
 * take the same forms: in form 2 the two products read the uint8 rows as their operand (four counts per 4-byte load,
 * expanded in registers), in form 1 -- and for shapes without those kernels -- over an fp32 copy that one cast launch
 * leaves in the workspace; the likelihood kernels read the uint8 rows in every case.
 * (Environment VAE_NO_XOPT / VAE_NO_XDIRECT / VAE_NO_XU8: A/B knobs.)                                                */
int vae_elbo_step_counts_path(const vae_step_config* cfg);

/* named slices of the workspace that a caller may read after the forward (offsets in floats)      */
enum { VAE_WS_QZ_HEADS = 0, /* (B, L+1): [:, :L] = qz_m                                            */
       VAE_WS_QZ_V, VAE_WS_Z, VAE_WS_KL_Z, VAE_WS_QL_HEADS, VAE_WS_QL_V, VAE_WS_LIBRARY, VAE_WS_KL_L,
       VAE_WS_LL,      /* (B) log-likelihood row sums; recon = -ll                                 */
       VAE_WS_SCALARS, /* 5 floats, see vae_elbo_reduce                                            */
       VAE_WS_DZ_EXTRA,/* (B, L): written by the CALLER between forward and backward (cell heads)  */
       VAE_WS_RECON,   /* (B) = -ll                                                                */
       VAE_WS_GENE_THRESHOLD, /* (G) uint32 keep thresholds of GeneDropout, then 1 float mean_g(p_g); -1 when off */
       VAE_WS_OPERAND_MAXIMA, /* start of the partial maxima of the VAE_FP32_F16X2 operands: rewritten in full by the
                               * kernels of every step, nothing for the caller to initialise or clear            */
       VAE_WS_NSLICES };
int64_t vae_elbo_step_workspace_bytes(const vae_step_config* cfg);
int64_t vae_elbo_step_ws_offset(const vae_step_config* cfg, int slice);
int vae_elbo_step_forward(const vae_step_config* cfg, const vae_step_tensors* params, const void* x, int64_t ldx,
                          const float* eps_z, const float* eps_l, float* workspace, int64_t workspace_bytes,
                          void* stream);
/* Inference / validation (LitVAE.validation_step, task.py:235-282) = vae_elbo_step_forward alone, then
 * optionally the decoder mean px_rate = exp(library) * softmax(logits) (modules.py:404; exported at
 * task.py:263) from the logits and row statistics the forward left in the workspace.                 */
int vae_elbo_step_px_rate(const vae_step_config* cfg, const float* workspace, int64_t workspace_bytes, float* px_rate,
                          int64_t ld_r, void* stream);
/* grads: same layout as params; small tensors (biases, LayerNorm, px_r) are ACCUMULATED into, the
 * weight matrices are overwritten.  use_dz_extra != 0 adds workspace[VAE_WS_DZ_EXTRA] to dz.
 * event_heads_ready (nullable hipEvent_t) is recorded once the decoder-head weight gradient -- the
 * largest all-reduce bucket -- has been enqueued.                                                   */
int vae_elbo_step_backward(const vae_step_config* cfg, const vae_step_tensors* params, const vae_step_tensors* grads,
                           const void* x, int64_t ldx, const float* eps_z, const float* eps_l, int use_dz_extra,
                           float* workspace, int64_t workspace_bytes, void* event_heads_ready, void* stream);
/* The same with a second release point for a data-parallel caller (the reference's DDP reduces gradient buckets while
 * the backward still runs, train.py:57): event_small_ready (nullable hipEvent_t) is recorded once EVERY gradient except
 * the first-layer encoder weights (enc_w0, written by the last GEMM of the backward) is final.                       */
int vae_elbo_step_backward_events(const vae_step_config* cfg, const vae_step_tensors* params, const vae_step_tensors* grads,
                                  const void* x, int64_t ldx, const float* eps_z, const float* eps_l, int use_dz_extra,
                                  float* workspace, int64_t workspace_bytes, void* event_heads_ready,
                                  void* event_small_ready, void* stream);

/* The same with the LAST bucket of a data-parallel caller cut into row slabs: the (H + Hl, G) first-layer encoder weight
 * gradient is written by the last GEMM of the backward, so its all-reduce cannot hide behind later compute; here that
 * product runs as n_enc_slabs launches over the rows [enc_row_bounds[i], enc_row_bounds[i + 1]) (i < n_enc_slabs; bounds
 * from 0 to H + Hl in multiples of 4 -- 128 keeps every slab on the tile pipeline; with input dropout one bound at H), each
 * a contiguous range of the gradient arena, and enc_slab_events[i] (hipEvent_t, nullable) is recorded behind slab i: slab
 * i travels while slab i + 1 is computed.  n_enc_slabs = 0: one launch, as vae_elbo_step_backward_events.              */
int vae_elbo_step_backward_slabs(const vae_step_config* cfg, const vae_step_tensors* params, const vae_step_tensors* grads,
                                 const void* x, int64_t ldx, const float* eps_z, const float* eps_l, int use_dz_extra,
                                 float* workspace, int64_t workspace_bytes, void* event_heads_ready, void* event_small_ready,
                                 const int64_t* enc_row_bounds, void* const* enc_slab_events, int n_enc_slabs, void* stream);

/* ---- cell-property heads ------------------------------------------------------
 * CellDecoder with linear heads (modules.py:297-324; cell_decoder_hidden_layer = False is the shipped setting,
 * config.py:67) and LitVAE._cell_loss (task.py:419-451), forward and backward on the (B, L) latent:
 *   head i: logits = z_i W_i^T + b_i, z_i = z (dz_coeff 1), z detached (stop_grad: dz_coeff 0) or z behind
 *   GradReverse(lambda) (dz_coeff = -lambda, functional.py:5-22);
 *   discrete: cross_entropy(logits, int(targets[:, i]), ignore_index = -100, reduction none); else (logits - targets[:, i])^2;
 *   loss = sum_i mean_b(per_i[b] * mask[b, i])    (the mean runs over ALL cells of the batch).
 * Writes *loss (unscaled), and -- dz != NULL -- dz (B, L) = d(loss_scale * loss)/dz summed over the heads with their
 * dz_coeff, dw[i] / db[i] (overwritten).  Two or three launches, fixed summation order.  L <= 64, sum of widths <= 64.
 * A discrete target outside [0, width) is treated like ignore_index (no loss, no gradient; torch raises there).
 * class_weight[i] (nullable; discrete heads only): `width` per-class weights of the head's cross-entropy, as
 * nn.CrossEntropyLoss(weight=clip(1/freq, 0.1, 10), reduction="none") under balance_classes (task.py:129-132): cell b's
 * loss and logit gradient are multiplied by class_weight[i][target_b].                                                  */
#define VAE_MAX_HEADS 7
typedef struct {
  int32_t n_heads;
  int32_t width[VAE_MAX_HEADS];      /* classes of a discrete head; 1 for a continuous head               */
  int32_t discrete[VAE_MAX_HEADS];
  float dz_coeff[VAE_MAX_HEADS];
  const float* w[VAE_MAX_HEADS];     /* (width, L)                                                        */
  const float* b[VAE_MAX_HEADS];     /* (width)                                                           */
  float* dw[VAE_MAX_HEADS];
  float* db[VAE_MAX_HEADS];
  const float* class_weight[VAE_MAX_HEADS];   /* (width) or NULL = unweighted                              */
} vae_cell_heads;
int64_t vae_cell_heads_workspace_bytes(int64_t B, int64_t L, const vae_cell_heads* heads);
int vae_cell_heads_step(const vae_cell_heads* heads, const float* z, int64_t B, int64_t L, const float* targets, int64_t ld_t,
                        const float* mask, int64_t ld_m, float loss_scale, float* dz, float* loss, float* workspace,
                        int64_t workspace_bytes, void* stream);

/* ---- fused optimiser step ------------------------------------------------------
 * One call = torch.nn.utils.clip_grad_norm_(max_grad_norm) (Trainer(gradient_clip_val=0.5),
 * train.py:54) followed by torch.optim.AdamW.step() (betas, eps, decoupled per-group weight decay,
 * bias correction; task.py:534-553) over ranges of the flat arenas.  Elements outside every
 * segment (parameters without a gradient) are not touched, as torch skips grad=None parameters.
 * grads are read as grad_scale * grads (1 / loss scale) and are NOT modified (torch scales them
 * in place when clipping).  lr is the scheduled learning rate of this step
 * (WarmupConstantAndDecaySchedule, task.py:662-676, evaluated by the host); step counts from 1.
 * max_grad_norm <= 0 disables clipping.  grad_norm_out (nullable, device): the unclipped norm.   */
typedef struct {
  int64_t begin, end;   /* element range [begin, end) of the arenas                                */
  float weight_decay;   /* of the optimiser group the range belongs to                              */
} vae_adamw_segment;
int64_t vae_adamw_workspace_bytes(void);
int vae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2, double eps,
                   int64_t step, double max_grad_norm, double grad_scale, float* workspace, int64_t workspace_bytes,
                   float* grad_norm_out, void* stream);
/* The same step that also leaves, for two element ranges [max_ranges[0], max_ranges[1]) and [max_ranges[2], max_ranges[3])
 * of the arena, vae_adamw_wmax_partials() partial maxima each of the UPDATED |params| in wmax_out (range 0 first): what
 * vae_step_config.weight_max_enc / weight_max_heads take.  HBM-bound as before: the maxima ride in the update pass.   */
int64_t vae_adamw_wmax_partials(void);
int vae_adamw_step_wmax(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2, double eps,
                        int64_t step, double max_grad_norm, double grad_scale, float* workspace, int64_t workspace_bytes,
                        float* grad_norm_out, const int64_t* max_ranges, float* wmax_out, void* stream);

/* The same step that also leaves bf16 images (nearest even) of the UPDATED params of the two element ranges
 * [img_ranges[0], img_ranges[1]) and [img_ranges[2], img_ranges[3]) (ends multiples of 4) in img0 / img1 (8-byte aligned,
 * range-length elements each): what vae_step_config.weight_bf16_enc / weight_bf16_heads take.  2 more bytes written per
 * tracked weight in the update pass instead of 6 moved by a cast launch in every step.                                    */
int vae_adamw_step_bf16(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2, double eps,
                        int64_t step, double max_grad_norm, double grad_scale, float* workspace, int64_t workspace_bytes,
                        float* grad_norm_out, const int64_t* img_ranges, void* img0, void* img1, void* stream);

/* L1 penalty of LitVAE.training_step (task.py:491-496): sum_out[0] = sum of |p| over the segments (their
 * weight_decay field is ignored) and, when grads is not null, grads += grad_coeff * sign(p) -- the gradient of
 * grad_coeff * sum|p| (torch.norm(p, 1)).  Deterministic two-stage sum.  workspace as for vae_adamw_step.   */
int vae_l1_penalty(const float* params, float* grads, int64_t n, const vae_adamw_segment* segments, int n_segments,
                   double grad_coeff, float* workspace, int64_t workspace_bytes, float* sum_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAE_AMD_H */
