// The fused training step as ONE native call per direction: LitVAE.training_step
// (reference task.py:453-502) = encoders -> decoder -> NB/ZINB + KL -> loss, and its backward,
// enqueued on a HIP stream as a fixed launch sequence over a caller-provided workspace.
// No allocation, no host synchronisation: safe to replay in a tight loop (or to capture).
#include "../../include/vae_amd.h"
#include "common.h"
#include "middle.h"
#include "internal.h"
#include "philox.h"

#include <stdlib.h>
#include <string.h>

namespace {

// Operands of the two-f16-term GEMMs whose largest magnitude the step tracks: the producing kernel's waves / workgroups
// store partial maxima (plain stores, every entry, every step), the consuming GEMM reduces them in its prologue.
enum { AM_XN = 0, AM_XNZ, AM_PX, AM_DLOGITS, AM_DPRE, AM_WENC, AM_WHEADS, AM_COUNT };

struct Layout {
  // offsets in floats into the workspace
  int64_t mean, xn, ld_xn, pre_enc0, h_enc0, ln_mz[VAE_MAX_LAYERS], ln_rz[VAE_MAX_LAYERS], ln_ml, ln_rl;
  int64_t encz_pre[VAE_MAX_LAYERS], encz_h[VAE_MAX_LAYERS];  // layers >= 1
  int64_t heads_z, heads_l, qz_v, z, kl_z, ql_v, library, kl_l;
  int64_t dec_pre[VAE_MAX_LAYERS], dec_h[VAE_MAX_LAYERS], ln_md[VAE_MAX_LAYERS], ln_rd[VAE_MAX_LAYERS];
  int64_t logits, table, ll, recon, rowmax, rowsum, S, scalars;
  int64_t dz_extra, dlibrary, dheads_z, dheads_l, dpre_enc0, dh, dpx, dz;
  int64_t dbet_d, dgam_d, dbet_e, dgam_e;  // per-cell LayerNorm-gradient rows of the fused middle
  int64_t xn_z, mean_z, gene_thr, gene_counts;  // GeneDropout: the z-encoder's own normalised counts (-1: off)
  int64_t amax[8]; int amax_n[8];  // partial maxima of the operands of the two-f16-term GEMMs (AM_*): offset, entries
  int64_t stat_m, stat_s, stat_cap;   // per-slot softmax statistics the decoder-head GEMM leaves (internal.h)
  int64_t w_enc16, w_heads16, px16, dpre16;   // bf16 images (bf16_storage): the two gene-sized weight matrices, px, dpre_enc0
  int64_t wsum, xflags, dpre_s;   // count-operand form of the encoder products (Ctx::xd): sum_g W, per-wave flags, dpre / mean
  int64_t xs_sum, ws_sum;         // ... its optimistic start (xoptimistic()): per-slab row sums of the counts [64][B] and of W [64][HE]
  int64_t fwd_part;       // per-strip partials of ll[b] / S_b of the streaming likelihood pass
  int64_t bwd_slabs;      // column-sum slabs of the epilogue backward (kept until the last launch of the backward)
  int64_t scratch;        // split-K slabs (shared, used by one launch at a time)
  int64_t scratch_floats;
  int64_t colpart;        // unfused backward only: per-workgroup column sums of its row kernels, added by one launch
  int64_t colpart_floats;
  int64_t total;
};

inline int64_t up4(int64_t v) { return (v + 3) & ~(int64_t)3; }

// A VAE_BF16 step keeps its gene-sized ACTIVATIONS -- the normalised counts and the (B, 2G) logits, the latter
// overwritten in place by their gradient -- as bf16 in the workspace, as the reference's autocast(bfloat16) does
// (config.py:87: nn.Linear outputs are bf16 tensors, and what autograd saves for the wgrad products is bf16): the
// decoder-head product rounds its output once, the likelihood passes read 2 B per logit, and the encoder / wgrad / dgrad
// products stage these operands without a conversion.  Only the logits' rounding changes a result (oracle: logits_bf16);
// an operand rounded when written has the bits it would get when read.  Needs the tile pipeline's shapes;
// VAE_BF16_F32_STORAGE=1 keeps fp32 (A/B knob).
bool bf16_storage(const vae_step_config& c) {
  static const bool off = getenv("VAE_BF16_F32_STORAGE") != nullptr;
  if (off || c.precision != VAE_BF16 || c.G % 8 != 0 || c.Hd % 4 != 0 || c.H % 4 != 0 || c.Hl % 4 != 0) return false;
  const int64_t NH = c.zinb ? 2 * (int64_t)c.G : c.G, HE = c.H + c.Hl;
  auto ok = vae_gemm_bf16_storage_ok;
  if (c.B % 4 != 0 || (c.B * (int64_t)c.Hd) % 4 != 0) return false;
  if (!(ok(VAE_NT, 7, c.B, NH, c.Hd) && ok(VAE_TN, 3, NH, c.Hd, c.B) && ok(VAE_NN, 3, c.B, c.Hd, NH))) return false;
  if (c.input_dropout_rate > 0.f)   // GeneDropout: one product per encoder
    return ok(VAE_NT, 3, c.B, c.H, c.G) && ok(VAE_NT, 3, c.B, c.Hl, c.G) && ok(VAE_TN, 3, c.H, c.G, c.B) && ok(VAE_TN, 3, c.Hl, c.G, c.B);
  return ok(VAE_NT, 3, c.B, HE, c.G) && ok(VAE_TN, 3, HE, c.G, c.B);
}
inline int64_t maxi(int64_t a, int64_t b) { return a > b ? a : b; }
// batch covariates (vae_step_config.n_cov): row strides of the two first-layer weights that hold covariate columns
inline int64_t ld_enc_w(const vae_step_config& c) { return c.n_cov > 0 ? (c.ld_enc_w > 0 ? c.ld_enc_w : c.G + c.n_cov) : c.G; }
inline int64_t ld_dec_w0(const vae_step_config& c) { return c.L + (c.n_cov > 0 ? c.n_cov : 0); }

bool make_layout(const vae_step_config& c, Layout* lo) {
  if (c.B <= 0 || c.G <= 0 || c.H <= 0 || c.Hl <= 0 || c.L <= 0 || c.Hd <= 0) return false;
  if (c.n_layers < 1 || c.n_layers > VAE_MAX_LAYERS) return false;
  int64_t off = 0;
  auto take = [&](int64_t n) { const int64_t o = off; off = up4(off + n); return o; };
  const int64_t B = c.B, G = c.G, H = c.H, Hl = c.Hl, L = c.L, Hd = c.Hd;
  const int64_t NH = c.zinb ? 2 * G : G;
  lo->mean = take(B);
  const bool hb = bf16_storage(c);
  lo->ld_xn = hb ? (G + 7) & ~(int64_t)7 : up4(G);   // normalised counts x/mean - 1, rows padded to 16 bytes (bf16: 8 elements)
  lo->xn = take(hb ? B * lo->ld_xn / 2 : B * lo->ld_xn);
  lo->pre_enc0 = take(B * (H + Hl));
  lo->h_enc0 = take(B * (H + Hl));
  for (int n = 0; n < c.n_layers; ++n) { lo->ln_mz[n] = take(B); lo->ln_rz[n] = take(B); }
  lo->ln_ml = take(B); lo->ln_rl = take(B);
  for (int n = 1; n < c.n_layers; ++n) { lo->encz_pre[n] = take(B * H); lo->encz_h[n] = take(B * H); }
  lo->heads_z = take(B * (L + 1)); lo->heads_l = take(B * 2);
  lo->qz_v = take(B); lo->z = take(B * L); lo->kl_z = take(B);
  lo->ql_v = take(B); lo->library = take(B); lo->kl_l = take(B);
  for (int n = 0; n < c.n_layers; ++n) {
    lo->dec_pre[n] = take(B * Hd); lo->dec_h[n] = take(B * Hd); lo->ln_md[n] = take(B); lo->ln_rd[n] = take(B);
  }
  lo->logits = take(bf16_storage(c) ? (B * NH + 1) / 2 : B * NH);   // bf16 storage: two logits per float slot
  lo->table = take(vae_gene_table_rows() * G);
  lo->ll = take(B); lo->recon = take(B); lo->rowmax = take(B); lo->rowsum = take(B); lo->S = take(B); lo->scalars = take(8);
  lo->dz_extra = take(B * L); lo->dlibrary = take(B);
  lo->dheads_z = take(B * (L + 1)); lo->dheads_l = take(B * 2);
  lo->dpre_enc0 = take(B * (H + Hl)); lo->dh = take(B * H); lo->dpx = take(B * Hd); lo->dz = take(B * L);
  lo->dbet_d = take(B * Hd); lo->dgam_d = take(B * Hd);
  lo->dbet_e = take(B * (H + Hl)); lo->dgam_e = take(B * (H + Hl));
  {  // partial maxima of the two-f16-term operands: every entry is rewritten by its producer in every step
    const bool mid = vae_mid_supported(c);
    const int64_t n[AM_COUNT] = {B * VAE_NORM_WAVES, B * VAE_NORM_WAVES, mid ? B * VAE_MID_WAVES : VAE_ABSMAX_PARTIALS,
                                 vae_nbzinb_bwd_workgroups(B, G), mid ? B * VAE_MID_WAVES : VAE_ABSMAX_PARTIALS,
                                 VAE_ABSMAX_W_SPLIT, VAE_ABSMAX_PARTIALS - VAE_ABSMAX_W_SPLIT};
    for (int i = 0; i < AM_COUNT; ++i) { lo->amax[i] = take(n[i]); lo->amax_n[i] = (int)n[i]; }
  }
  lo->w_enc16 = lo->w_heads16 = lo->px16 = lo->dpre16 = -1;
  if (hb) {
    lo->w_enc16 = take((H + Hl) * ld_enc_w(c) / 2 + 4);
    lo->w_heads16 = take(NH * Hd / 2 + 4);
    lo->px16 = take(B * Hd / 2 + 4);
    lo->dpre16 = take(B * (H + Hl) / 2 + 4);
  }
  lo->xn_z = lo->mean_z = lo->gene_thr = lo->gene_counts = -1;
  if (c.input_dropout_rate > 0.f) {
    lo->xn_z = take(hb ? B * lo->ld_xn / 2 : B * lo->ld_xn);
    lo->mean_z = take(B);
    lo->gene_thr = take(G + 4);      // G thresholds, then mean_g(p_g)
    lo->gene_counts = take(G);
  }
  // scratch: the largest split-K / column-slab need of any launch in the step
  int64_t s = 0;
  auto gw = [&](int64_t M, int64_t N, int64_t K) { s = maxi(s, vae_gemm_workspace_bytes(M, N, K)); };
  gw(B, H + Hl, G); gw(H + Hl, G, B);                       // encoder first layer fwd / wgrad
  gw(B, H, G); gw(H, G, B); gw(B, Hl, G); gw(Hl, G, B);     // the same per encoder (GeneDropout)
  gw(B, H, H); gw(H, H, B);                                 // deeper encoder layers
  gw(B, L + 1, H); gw(L + 1, H, B); gw(B, H, L + 1);        // z heads
  gw(B, 2, Hl); gw(2, Hl, B); gw(B, Hl, 2);                 // library heads
  gw(B, Hd, L); gw(Hd, L, B); gw(B, L, Hd);                 // decoder first layer
  gw(B, Hd, Hd); gw(Hd, Hd, B);                             // deeper decoder layers
  gw(B, NH, Hd); gw(NH, Hd, B); gw(B, Hd, NH);              // decoder heads
  if (c.n_cov > 0) {                                          // the covariate columns' small products
    const int64_t Cc = c.n_cov;
    gw(B, H + Hl, Cc); gw(H + Hl, Cc, B); gw(B, Hd, Cc); gw(Hd, Cc, B);
  }
  s = maxi(s, vae_gemm_slabs_workspace_bytes(B, H + Hl, G));  // slabs consumed by the fused middle
  s = maxi(s, vae_gemm_slabs_workspace_bytes(B, Hd, NH));
  s = maxi(s, vae_nbzinb_bwd_workspace_bytes(B, G));
  lo->wsum = take(H + Hl); lo->xflags = take(B * VAE_NORM_WAVES); lo->dpre_s = take(B * (H + Hl));
  lo->xs_sum = take(64 * B); lo->ws_sum = take(64 * (H + Hl));
  lo->fwd_part = take(vae_nbzinb_fwd_workspace_bytes(B, G) / 4 + 4);
  lo->stat_cap = B * ((G + 31) / 32);            // narrowest slot of the GEMM's row statistics: 32 columns
  lo->stat_m = take(lo->stat_cap); lo->stat_s = take(lo->stat_cap);
  lo->bwd_slabs = take(vae_nbzinb_bwd_workspace_bytes(B, G) / 4 + 4);
  lo->scratch_floats = up4(s / 4 + 4);
  lo->scratch = take(lo->scratch_floats);
  lo->colpart = -1; lo->colpart_floats = 0;
  if (!vae_mid_supported(c)) {   // per-workgroup column sums of the unfused backward's row kernels (vae_colpart_reduce)
    const int64_t nb = vae_colpart_blocks(B);
    lo->colpart_floats = nb * 3 * (c.n_layers * Hd + (c.n_layers - 1) * H + H + Hl) + nb * (L + 1) + nb * 2;
    lo->colpart = take(lo->colpart_floats);
  }
  lo->total = off;
  return true;
}

#define TRY(call)               \
  do {                          \
    const int rc__ = (call);    \
    if (rc__ != VAE_OK) return rc__; \
  } while (0)

struct Ctx {
  const vae_step_config& c;
  float* ws;
  Layout lo;
  void* st;
  const vae_step_tensors* par;   // the parameters (for offsets into their bf16 images)
  bool xd;                       // count-operand form of the two encoder products (xdirect() below)
  const float* xf = nullptr;     // ... the fp32 counts they read: the caller's, or the copy of a uint8 batch in lo.xn
  int64_t ldxf = 0;
  bool xu8 = false;              // ... or no copy at all: the products expand the uint8 rows themselves (xcounts_u8() below)
  uint16_t* h16(int64_t off) const { return reinterpret_cast<uint16_t*>(ws + off); }
  // the two-f16-term pipeline needs 16-byte rows everywhere; ragged shapes run as VAE_FP32
  bool hb() const { return bf16_storage(c); }   // logits / logit gradients held as bf16
  int ldt() const { return hb() ? VAE_L_BF16 : VAE_L_F32; }
  bool fx() const {
    return c.precision == VAE_FP32_F16X2 && c.G % 4 == 0 && c.Hd % 4 == 0 && (c.H + c.Hl) % 4 == 0 && ld_enc_w(c) % 4 == 0;
  }
  int64_t ldw() const { return ld_enc_w(c); }
  bool cov() const { return c.n_cov > 0; }
  // the two weight matrices' maxima may come from outside (vae_step_config.weight_max_*: the optimiser keeps them)
  bool wext() const { return c.weight_max_enc && c.weight_max_heads && c.weight_max_n > 0; }
  // ... and so may their bf16 images (vae_step_config.weight_bf16_*); else the forward casts them into the workspace
  bool wimg() const { return c.weight_bf16_enc && c.weight_bf16_heads; }
  const uint16_t* w_enc16() const { return wimg() ? (const uint16_t*)c.weight_bf16_enc : h16(lo.w_enc16); }
  const uint16_t* w_heads16() const { return wimg() ? (const uint16_t*)c.weight_bf16_heads : h16(lo.w_heads16); }
  float* am(int slot) const {
    if (!fx()) return nullptr;
    if (wext() && slot == AM_WENC) return const_cast<float*>(c.weight_max_enc);
    if (wext() && slot == AM_WHEADS) return const_cast<float*>(c.weight_max_heads);
    return ws + lo.amax[slot];
  }
  int an(int slot) const { return (wext() && (slot == AM_WENC || slot == AM_WHEADS)) ? c.weight_max_n : lo.amax_n[slot]; }
  // one of the five gene-sized products: two-f16-term split when the step runs in VAE_FP32_F16X2, else as `gemm`;
  // slabs_out != null leaves split-K slabs in the scratch (no bias) for the fused middle
  int big(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int slot_a, const float* Bm, int64_t ldb,
          int slot_b, float* C, int64_t ldc, const float* bias, int* slabs_out, VaeGemmRowStats* stats = nullptr) const {
    if (fx())
      return vae_gemm_f16x2(layout, M, N, K, A, lda, am(slot_a), an(slot_a), Bm, ldb, am(slot_b), an(slot_b), C, ldc,
                            bias, 0, ws + lo.scratch,
                            lo.scratch_floats * 4, st, slabs_out, stats);
    // a VAE_FP32_F16X2 step whose shapes are outside the two-f16-term pipeline (fx() false) has no operand maxima:
    // its gene-sized products run as VAE_FP32 like every other product of the step
    const int prec = c.precision == VAE_FP32_F16X2 ? VAE_FP32 : c.precision;
    if (slabs_out)
      return vae_gemm_slabs(layout, prec, M, N, K, A, lda, Bm, ldb, ws + lo.scratch, lo.scratch_floats * 4, slabs_out, st);
    return vae_gemm_rowstats(layout, prec, M, N, K, A, lda, Bm, ldb, C, ldc, bias, ws + lo.scratch, lo.scratch_floats * 4, stats, st);
  }
  // the two products over the normalised counts (xn: fp32, or bf16 when hb()): forward NT (xn = A), weight gradient TN (xn = B)
  int enc_fwd(int64_t N, const float* xn, int slot_x, const float* W, float* C, int64_t ldc, const float* bias, int* slabs_out) const {
    if (hb())   // bf16 xn x the bf16 image of the weight rows
      return vae_gemm_bf16io(VAE_NT, VAE_IO_A_BF16 | VAE_IO_B_BF16, c.B, N, c.G, xn, lo.ld_xn, w_enc16() + (W - par->enc_w0), ldw(), C,
                             ldc, bias, ws + lo.scratch, lo.scratch_floats * 4, slabs_out, nullptr, st);
    return big(VAE_NT, c.B, N, c.G, xn, lo.ld_xn, slot_x, W, ldw(), AM_WENC, C, ldc, bias, slabs_out);
  }
  int enc_wgrad(int64_t M, const float* dpre, int64_t ld_dpre, const float* xn, int slot_x, float* dW) const {
    if (hb())   // the bf16 image of dpre (cast by the caller) x bf16 xn
      return vae_gemm_bf16io(VAE_TN, VAE_IO_A_BF16 | VAE_IO_B_BF16, M, c.G, c.B, h16(lo.dpre16) + (dpre - (ws + lo.dpre_enc0)), ld_dpre, xn,
                             lo.ld_xn, dW, ldw(), nullptr, ws + lo.scratch, lo.scratch_floats * 4, nullptr, nullptr, st);
    return big(VAE_TN, M, c.G, c.B, dpre, ld_dpre, AM_DPRE, xn, lo.ld_xn, slot_x, dW, ldw(), nullptr, nullptr);
  }
  // nn.Dropout after a hidden LayerNorm (forward: the activation; backward: the gradient arriving there)
  int drop(float* x, int64_t ld, int64_t W, uint32_t site) const {
    return vae_dropout_apply(x, ld, c.B, W, c.dropout_rate, c.rng_seed, c.rng_step, site, st);
  }
  int gemm(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* Bm, int64_t ldb,
           float* C, int64_t ldc, const float* bias, const float* mean, int acc) const {
    return vae_gemm(layout, c.precision == VAE_FP32_F16X2 ? VAE_FP32 : c.precision, M, N, K, A, lda, Bm, ldb, C, ldc, bias, mean, acc,
                    ws + lo.scratch,
                    lo.scratch_floats * 4, st);
  }
};

// The two encoder products of a two-f16-term step on the fused-middle path stream the RAW fp32 counts (VaeGemmCounts,
// internal.h): no x / mean - 1 matrix is written or read, and integer counts run on one f16 term.  A/B knob: VAE_NO_XDIRECT.
bool xdirect(const vae_step_config& c, const Ctx& k, const void* x, int64_t ldx) {
  static const bool off = getenv("VAE_NO_XDIRECT") != nullptr;
  if (off || !k.fx() || !vae_mid_supported(c) || c.input_dropout_rate > 0.f) return false;
  if (c.x_dtype == VAE_X_U8) {   // the products read an fp32 copy made in the workspace (one cast launch; B bounds its grid)
    if (ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 3u) != 0 || c.B >= 65536) return false;
  } else if (ldx % 4 != 0 || !vae_aligned16(x) || ldx >= (1 << 22)) return false;
  const int64_t HE = c.H + c.Hl;
  return vae_gemm_tile_pipeline_ok(VAE_NT, c.B, HE, c.G) && vae_gemm_tile_pipeline_ok(VAE_TN, HE, c.G, c.B);
}

// the fp32 counts of a count-operand step: the caller's rows, or (uint8 batch) their copy in the slice that otherwise holds x / mean - 1
void set_count_rows(Ctx& k, const void* xv, int64_t ldx) {
  if (k.c.x_dtype == VAE_X_U8) { k.xf = k.ws + k.lo.xn; k.ldxf = k.lo.ld_xn; }
  else { k.xf = (const float*)xv; k.ldxf = ldx; }
}

// ... and nobody reads the counts ahead of the forward product: it takes them as exact in one unscaled f16 term, gathers
// the cells' sums / maxima / exactness flags on the way, and a gated launch behind it finishes the means and sum_g W and
// repeats the product in the scaled form only when a flag came up (VaeGemmCounts, internal.h).  One launch that reads
// the batch (and the first-layer weight) less per step.  A/B knob: VAE_NO_XOPT.
bool xoptimistic(const vae_step_config& c, const Ctx& k) {
  static const bool off = getenv("VAE_NO_XOPT") != nullptr;
  return !off && k.xd && !c.fractional_counts && vae_gemm_counts_optimistic_ok(c.B, c.H + c.Hl, c.G, (int)(c.B * VAE_NORM_WAVES));
}

// A uint8 batch whose forward product runs optimistically needs no fp32 copy: that kernel and the 128 x 128 weight-gradient
// kernel read the uint8 rows as their operand (four counts per 4-byte load).  A/B knob: VAE_NO_XU8 (then: one cast launch).
bool xcounts_u8(const vae_step_config& c, const Ctx& k) {
  static const bool off = getenv("VAE_NO_XU8") != nullptr;
  return !off && c.x_dtype == VAE_X_U8 && xoptimistic(c, k);
}

int check_common(const vae_step_config* cfg, const void* a, const void* b, const void* x, int64_t ldx,
                 const float* eps_z, const float* eps_l, float* ws, int64_t ws_bytes, Layout* lo, const char* who) {
  if (!cfg || !a || !b || !x || !eps_z || !eps_l || !ws) return vae_set_error(VAE_ERR_ARG, "%s: null pointer", who);
  if (!make_layout(*cfg, lo)) return vae_set_error(VAE_ERR_ARG, "%s: bad configuration", who);
  if (ldx < cfg->G) return vae_set_error(VAE_ERR_ARG, "%s: ldx < G", who);
  if (cfg->x_dtype != VAE_X_F32 && cfg->x_dtype != VAE_X_U8) return vae_set_error(VAE_ERR_ARG, "%s: bad x_dtype", who);
  if (cfg->B >= (1 << 30) || cfg->G >= (1 << 24))   // (the likelihood kernels address the gene table with 24-bit multiplies)
    return vae_set_error(VAE_ERR_ARG, "%s: B=%lld / G=%lld too large", who, (long long)cfg->B, (long long)cfg->G);
  if (!(cfg->l_var > 0.f)) return vae_set_error(VAE_ERR_ARG, "%s: l_var must be positive", who);
  if (cfg->precision != VAE_FP32 && cfg->precision != VAE_BF16 && cfg->precision != VAE_FP32_F16X2)
    return vae_set_error(VAE_ERR_ARG, "%s: precision must be VAE_FP32, VAE_BF16 or VAE_FP32_F16X2", who);
  if (!(cfg->dropout_rate >= 0.f && cfg->dropout_rate < 1.f))
    return vae_set_error(VAE_ERR_ARG, "%s: dropout_rate must be in [0, 1)", who);
  if (!(cfg->input_dropout_rate >= 0.f && cfg->input_dropout_rate <= 1.f))
    return vae_set_error(VAE_ERR_ARG, "%s: input_dropout_rate must be in [0, 1]", who);
  if (cfg->input_dropout_rate > 0.f && (!cfg->gene_running || cfg->G % 4 != 0))
    return vae_set_error(VAE_ERR_ARG, "%s: input dropout needs gene_running and G %% 4 == 0", who);
  if (cfg->n_cov < 0 || (cfg->n_cov > 0 && (!cfg->batch_cov || cfg->ld_cov < cfg->n_cov ||
                                            (cfg->ld_enc_w != 0 && cfg->ld_enc_w < cfg->G + cfg->n_cov))))
    return vae_set_error(VAE_ERR_ARG, "%s: batch covariates need batch_cov (B, n_cov), ld_cov >= n_cov and ld_enc_w >= G + n_cov (or 0)", who);
  if (ws_bytes < lo->total * 4)
    return vae_set_error(VAE_ERR_ARG, "%s: workspace too small: %lld < %lld bytes", who, (long long)ws_bytes,
                         (long long)(lo->total * 4));
  if (!vae_aligned16(ws)) return vae_set_error(VAE_ERR_ARG, "%s: workspace must be 16-byte aligned", who);
  return VAE_OK;
}

}  // namespace

extern "C" int64_t vae_elbo_step_workspace_bytes(const vae_step_config* cfg) {
  Layout lo;
  if (!cfg || !make_layout(*cfg, &lo)) return -1;
  return lo.total * 4;
}

extern "C" int vae_elbo_step_bf16_activations(const vae_step_config* cfg) { return cfg && bf16_storage(*cfg) ? 1 : 0; }

extern "C" int vae_elbo_step_counts_path(const vae_step_config* cfg) {
  Layout lo;
  if (!cfg || !make_layout(*cfg, &lo)) return -1;
  Ctx k{*cfg, nullptr, lo, nullptr, nullptr, false};
  alignas(16) static const float aligned_rows[4] = {0.f, 0.f, 0.f, 0.f};
  k.xd = xdirect(*cfg, k, aligned_rows, (cfg->G + 3) / 4 * 4);
  return !k.xd ? 0 : (xoptimistic(*cfg, k) ? 2 : 1);
}

extern "C" int64_t vae_elbo_step_ws_offset(const vae_step_config* cfg, int slice) {
  Layout lo;
  if (!cfg || !make_layout(*cfg, &lo)) return -1;
  switch (slice) {
    case VAE_WS_QZ_HEADS: return lo.heads_z;
    case VAE_WS_QZ_V: return lo.qz_v;
    case VAE_WS_Z: return lo.z;
    case VAE_WS_KL_Z: return lo.kl_z;
    case VAE_WS_QL_HEADS: return lo.heads_l;
    case VAE_WS_QL_V: return lo.ql_v;
    case VAE_WS_LIBRARY: return lo.library;
    case VAE_WS_KL_L: return lo.kl_l;
    case VAE_WS_LL: return lo.ll;
    case VAE_WS_SCALARS: return lo.scalars;
    case VAE_WS_DZ_EXTRA: return lo.dz_extra;
    case VAE_WS_RECON: return lo.recon;
    case VAE_WS_GENE_THRESHOLD: return lo.gene_thr;
    case VAE_WS_OPERAND_MAXIMA: return lo.amax[0];
    default: return -1;
  }
}

extern "C" int vae_elbo_step_forward(const vae_step_config* cfg, const vae_step_tensors* p, const void* xv, int64_t ldx,
                                     const float* eps_z, const float* eps_l, float* ws, int64_t ws_bytes, void* stream) {
  Layout lo;
  TRY(check_common(cfg, p, p, xv, ldx, eps_z, eps_l, ws, ws_bytes, &lo, "vae_elbo_step_forward"));
  const vae_step_config& c = *cfg;
  Ctx k{c, ws, lo, stream, p, false};
  k.xd = xdirect(c, k, xv, ldx);
  const int64_t B = c.B, G = c.G, H = c.H, Hl = c.Hl, L = c.L, Hd = c.Hd, HE = c.H + c.Hl;
  const int64_t NH = c.zinb ? 2 * G : G;
  if (k.xd) {
    set_count_rows(k, xv, ldx);
    k.xu8 = xcounts_u8(c, k);
    if (c.x_dtype == VAE_X_U8 && !k.xu8) TRY(vae_cast_u8_f32(xv, ldx, B, G, ws + lo.xn, lo.ld_xn, stream));
  }

  bool table_built = false;   // the gene table rides in another launch where one is at hand
  if (k.fx() && k.wext()) {  // the caller keeps the weight maxima current: only the gene table is left to build (below)
    if (!vae_aligned16(c.weight_max_enc) || !vae_aligned16(c.weight_max_heads))
      return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_forward: weight_max_* must be 16-byte aligned");
  } else if (k.fx()) {  // weight maxima of the two-f16-term products and the gene table, one launch
    TRY(vae_step_prologue(p->enc_w0, HE * k.ldw(), p->heads_w, NH * Hd, k.am(AM_WENC), k.am(AM_WHEADS), p->px_r, G, ws + lo.table,
                          stream));
    table_built = true;
  }
  if (k.hb() && k.wimg()) {   // the caller keeps the bf16 images of the two gene-sized weight matrices current
    if (!vae_aligned16(c.weight_bf16_enc) || !vae_aligned16(c.weight_bf16_heads))
      return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_forward: weight_bf16_* must be 16-byte aligned");
  } else if (k.hb()) {        // bf16 images of the two gene-sized weight matrices: every product of this step streams those
    TRY(vae_cast_bf16(p->enc_w0, k.h16(lo.w_enc16), HE * k.ldw(), p->heads_w, k.h16(lo.w_heads16), NH * Hd, stream));
  }
  // encoders: x/mean-1 -> Linear(G, H+Hl) -> GELU -> LayerNorm (z and library halves)
  const bool xopt = xoptimistic(c, k);
  if (xopt) {
    // (nothing here: the forward product gathers the statistics itself)
  } else if (k.xd)   // raw counts go to the products: the cells' means, maxima and exactness flags, and sum_g W (one launch)
    TRY(vae_row_stats(k.xf, k.ldxf, B, G, ws + lo.mean, k.am(AM_XN), ws + lo.xflags, p->enc_w0, k.ldw(), HE, ws + lo.wsum, stream));
  else
    TRY(vae_normalize_rows_ex(xv, c.x_dtype, ldx, B, G, ws + lo.mean, ws + lo.xn, k.hb(), lo.ld_xn, k.am(AM_XN), stream));
  // GeneDropout (training): the z-encoder reads its own masked + normalised counts, the library encoder xn
  const bool gd = c.input_dropout_rate > 0.f;
  if (gd)
    TRY(vae_gene_dropout_ex(xv, c.x_dtype, ldx, B, G, c.input_dropout_rate, c.input_dropout_alpha, c.rng_seed, c.rng_step,
                            c.gene_running, (int32_t*)(ws + lo.gene_counts), (uint32_t*)(ws + lo.gene_thr),
                            ws + lo.gene_thr + G, ws + lo.mean_z, ws + lo.xn_z, k.hb(), lo.ld_xn, k.am(AM_XNZ), stream));
  const float* px;
  if (vae_mid_supported(c)) {
    // one launch from the encoder slabs to the decoder hidden layer
    int nslab = 0;
    const float* slabs = ws + lo.scratch;
    if (gd) {  // two products (different inputs), left without bias in pre_enc0 = a single "slab"
      TRY(k.enc_fwd(H, ws + lo.xn_z, AM_XNZ, p->enc_w0, ws + lo.pre_enc0, HE, nullptr, nullptr));
      TRY(k.enc_fwd(Hl, ws + lo.xn, AM_XN, p->enc_w0 + H * k.ldw(), ws + lo.pre_enc0 + H, HE, nullptr, nullptr));
      slabs = ws + lo.pre_enc0;
      nslab = 1;
    } else if (xopt) {   // the same, optimistically: product first, statistics on the way, repeated only if they object
      VaeGemmCounts xc{nullptr, 0, nullptr};
      xc.mode = 1; xc.xs_sum = ws + lo.xs_sum; xc.ws_sum = ws + lo.ws_sum; xc.xs_max = k.am(AM_XN); xc.xs_flag = ws + lo.xflags; xc.xs_n = (int)(B * VAE_NORM_WAVES);
      xc.x_u8 = k.xu8 ? 1 : 0;
      TRY(vae_gemm_f16x2(VAE_NT, B, HE, G, k.xu8 ? (const float*)xv : k.xf, k.xu8 ? ldx : k.ldxf, k.am(AM_XN), k.an(AM_XN), p->enc_w0,
                         k.ldw(), k.am(AM_WENC), k.an(AM_WENC), nullptr, HE, nullptr, 0, ws + lo.scratch, lo.scratch_floats * 4, stream,
                         &nslab, nullptr, &xc));
      xc.x_u8 = 0;   // (the gated launch never reads the counts of a uint8 batch -- they are exact: k.xf is its placeholder)
      if (nslab > 64) return vae_set_error(VAE_ERR_LAUNCH, "vae_elbo_step_forward: %d slabs of row sums", nslab);
      int nslab2 = 0;
      xc.mode = 2; xc.flags = ws + lo.xflags; xc.n_flags = xc.xs_n; xc.mean_out = ws + lo.mean; xc.wsum_out = ws + lo.wsum;
      TRY(vae_gemm_f16x2(VAE_NT, B, HE, G, k.xf, k.ldxf, k.am(AM_XN), k.an(AM_XN), p->enc_w0, k.ldw(), k.am(AM_WENC),
                         k.an(AM_WENC), nullptr, HE, nullptr, 0, ws + lo.scratch, lo.scratch_floats * 4, stream, &nslab2, nullptr, &xc));
      if (nslab2 != nslab) return vae_set_error(VAE_ERR_LAUNCH, "vae_elbo_step_forward: slab counts differ (%d, %d)", nslab, nslab2);
    } else if (k.xd) {   // x W^T of the raw counts; the middle applies 1 / mean and - sum_g W
      const VaeGemmCounts xc{ws + lo.xflags, (int)(B * VAE_NORM_WAVES), nullptr};
      TRY(vae_gemm_f16x2(VAE_NT, B, HE, G, k.xf, k.ldxf, k.am(AM_XN), k.an(AM_XN), p->enc_w0, k.ldw(), k.am(AM_WENC),
                         k.an(AM_WENC), nullptr, HE, nullptr, 0, ws + lo.scratch, lo.scratch_floats * 4, stream, &nslab, nullptr, &xc));
    } else {
      TRY(k.enc_fwd(HE, ws + lo.xn, AM_XN, p->enc_w0, nullptr, HE, nullptr, &nslab));
    }
    TRY(vae_mid_forward(c, p, slabs, nslab, eps_z, eps_l, ws + lo.pre_enc0, ws + lo.h_enc0, ws + lo.ln_mz[0],
                        ws + lo.ln_rz[0], ws + lo.ln_ml, ws + lo.ln_rl, ws + lo.heads_z, ws + lo.heads_l, ws + lo.qz_v,
                        ws + lo.z, ws + lo.kl_z, ws + lo.ql_v, ws + lo.library, ws + lo.kl_l, ws + lo.dec_pre[0],
                        ws + lo.dec_h[0], ws + lo.ln_md[0], ws + lo.ln_rd[0], k.am(AM_PX), k.hb() ? k.h16(lo.px16) : nullptr,
                        k.xd ? ws + lo.mean : nullptr, k.xd ? ws + lo.wsum : nullptr, stream));
    px = ws + lo.dec_h[0];
  } else {
    if (gd) {
      TRY(k.enc_fwd(H, ws + lo.xn_z, AM_XNZ, p->enc_w0, ws + lo.pre_enc0, HE, p->enc_b0, nullptr));
      TRY(k.enc_fwd(Hl, ws + lo.xn, AM_XN, p->enc_w0 + H * k.ldw(), ws + lo.pre_enc0 + H, HE, p->enc_b0 + H, nullptr));
    } else {
      TRY(k.enc_fwd(HE, ws + lo.xn, AM_XN, p->enc_w0, ws + lo.pre_enc0, HE, p->enc_b0, nullptr));
    }
    if (k.cov())   // + cov W_c^T: the covariate columns of both encoders' first layer (modules.py:130)
      TRY(k.gemm(VAE_NT, B, HE, c.n_cov, c.batch_cov, c.ld_cov, p->enc_w0 + G, k.ldw(), ws + lo.pre_enc0, HE, nullptr, nullptr, 1));
    TRY(vae_gelu_ln_fwd(ws + lo.pre_enc0, HE, p->encz_ln_g[0], p->encz_ln_b[0], B, H, ws + lo.h_enc0, HE,
                        ws + lo.ln_mz[0], ws + lo.ln_rz[0], stream));
    TRY(vae_gelu_ln_fwd(ws + lo.pre_enc0 + H, HE, p->encl_ln_g, p->encl_ln_b, B, Hl, ws + lo.h_enc0 + H, HE,
                        ws + lo.ln_ml, ws + lo.ln_rl, stream));
    TRY(k.drop(ws + lo.h_enc0, HE, H, VAE_SITE_ENC_Z(0)));
    const float* hz = ws + lo.h_enc0;
    int64_t ld_hz = HE;
    for (int n = 1; n < c.n_layers; ++n) {
      TRY(k.gemm(VAE_NT, B, H, H, hz, ld_hz, p->encz_w[n], H, ws + lo.encz_pre[n], H, p->encz_b[n], nullptr, 0));
      TRY(vae_gelu_ln_fwd(ws + lo.encz_pre[n], H, p->encz_ln_g[n], p->encz_ln_b[n], B, H, ws + lo.encz_h[n], H,
                          ws + lo.ln_mz[n], ws + lo.ln_rz[n], stream));
      TRY(k.drop(ws + lo.encz_h[n], H, H, VAE_SITE_ENC_Z(n)));
      hz = ws + lo.encz_h[n];
      ld_hz = H;
    }
    // latent heads + reparameterisation + KL
    TRY(k.gemm(VAE_NT, B, L + 1, H, hz, ld_hz, p->hz_w, H, ws + lo.heads_z, L + 1, p->hz_b, nullptr, 0));
    TRY(k.gemm(VAE_NT, B, 2, Hl, ws + lo.h_enc0 + H, HE, p->hl_w, Hl, ws + lo.heads_l, 2, p->hl_b, nullptr, 0));
    TRY(vae_reparam_kl_fwd(ws + lo.heads_z, L + 1, eps_z, L, B, L, 0.f, 1.f, ws + lo.qz_v, ws + lo.z, L, ws + lo.kl_z, stream));
    TRY(vae_reparam_kl_fwd(ws + lo.heads_l, 2, eps_l, 1, B, 1, c.l_mean, c.l_var, ws + lo.ql_v, ws + lo.library, 1,
                           ws + lo.kl_l, stream));
    // decoder hidden layers
    px = ws + lo.z;
    int64_t kin = L;
    for (int n = 0; n < c.n_layers; ++n) {
      const int64_t ldwd = n == 0 ? ld_dec_w0(c) : Hd;
      TRY(k.gemm(VAE_NT, B, Hd, kin, px, kin, p->dec_w[n], ldwd, ws + lo.dec_pre[n], Hd, p->dec_b[n], nullptr, 0));
      if (n == 0 && k.cov())   // the decoder's covariate columns (modules.py:400,130)
        TRY(k.gemm(VAE_NT, B, Hd, c.n_cov, c.batch_cov, c.ld_cov, p->dec_w[0] + L, ldwd, ws + lo.dec_pre[0], Hd, nullptr, nullptr, 1));
      // the last layer's output again as bf16 for the decoder-head product of a bf16-storage step: written by the dropout
      // launch when there is one, else by the LayerNorm kernel
      uint16_t* px16 = (k.hb() && n == c.n_layers - 1) ? k.h16(lo.px16) : nullptr;
      const bool dropped = c.dropout_rate > 0.f;
      TRY(vae_gelu_ln_fwd_ex(ws + lo.dec_pre[n], Hd, p->dec_ln_g[n], p->dec_ln_b[n], B, Hd, ws + lo.dec_h[n], Hd,
                             ws + lo.ln_md[n], ws + lo.ln_rd[n], dropped ? nullptr : px16, Hd, stream));
      TRY(vae_dropout_apply_ex(ws + lo.dec_h[n], Hd, B, Hd, c.dropout_rate, c.rng_seed, c.rng_step, VAE_SITE_DEC(n), px16, Hd, stream));
      px = ws + lo.dec_h[n];
      kin = Hd;
    }
  }
  // decoder heads -> fused softmax / library scale / NB-ZINB / row sum -> ELBO scalars
  if (k.fx() && !vae_mid_supported(c)) TRY(vae_absmax_partials(px, B, Hd, Hd, k.am(AM_PX), stream));
  // ... whose epilogue also leaves the softmax statistics of the scale logits (per row and column span) when the shape
  // runs the split kernel; a tiny launch merges them.  Otherwise the likelihood pass computes them itself.
  VaeGemmRowStats rs{ws + lo.stat_m, ws + lo.stat_s, (int)G, lo.stat_cap, 0};
  static const bool no_gemm_stats = getenv("VAE_NO_GEMM_STATS") != nullptr;   // A/B knob
  if (k.hb()) {
    // (px16, the bf16 image of px: left by the fused middle or by the last hidden layer's LayerNorm / dropout launch)
    TRY(vae_gemm_bf16io(VAE_NT, VAE_IO_A_BF16 | VAE_IO_B_BF16 | VAE_IO_C_BF16, B, NH, Hd, k.h16(lo.px16), Hd, k.w_heads16(), Hd,
                        ws + lo.logits, NH, p->heads_b, ws + lo.scratch, lo.scratch_floats * 4, nullptr, no_gemm_stats ? nullptr : &rs,
                        stream));
  }
  else
    TRY(k.big(VAE_NT, B, NH, Hd, px, Hd, AM_PX, p->heads_w, Hd, AM_WHEADS, ws + lo.logits, NH, p->heads_b, nullptr,
              no_gemm_stats ? nullptr : &rs));
  if (rs.slots > 0) {   // ... and, in extra workgroups of the same launch, the gene table when it is still to build
    TRY(vae_softmax_stats_merge(ws + lo.stat_m, ws + lo.stat_s, rs.slots, B, ws + lo.rowmax, ws + lo.rowsum,
                                table_built ? nullptr : p->px_r, G, ws + lo.table, stream));
    table_built = true;
  }
  if (!table_built) TRY(vae_gene_table(p->px_r, G, ws + lo.table, stream));
  // streaming likelihood pass: row statistics, then (gene strip x cell chunk) tiles; its per-strip partial sums are added
  // up by the ELBO reduction
  int strips = 0;
  const void* lgp = k.hb() ? (const void*)((const uint16_t*)(ws + lo.logits) + G) : (const void*)(ws + lo.logits + G);
  TRY(vae_nbzinb_fwd_tiled_ex(c.x_dtype, k.ldt(), c.zinb, ws + lo.logits, NH, c.zinb ? lgp : nullptr, NH, xv, ldx,
                              ws + lo.library, ws + lo.table, B, G, nullptr, ws + lo.rowmax, ws + lo.rowsum, nullptr, rs.slots > 0,
                              ws + lo.fwd_part, vae_nbzinb_fwd_workspace_bytes(B, G), &strips, stream));
  if (B > 1024) {   // one workgroup walking strips x B partials is a latency chain (27 us at B = 4096): finish ll / S_b on
                    // B / 256 workgroups first, then the scalars from three short arrays
    TRY(vae_nbzinb_fwd_finish(ws + lo.fwd_part, strips, B, ws + lo.ll, ws + lo.S, stream));
    TRY(vae_elbo_reduce(ws + lo.ll, ws + lo.kl_z, ws + lo.kl_l, B, c.kl_weight, c.gene_loss_coeff, ws + lo.scalars, ws + lo.recon,
                        stream));
  } else if (!c.defer_elbo_reduce)   // (deferred: the backward's first launch reduces the partials, vae_step_config)
    TRY(vae_elbo_reduce_parts(ws + lo.fwd_part, strips, ws + lo.ll, ws + lo.S, ws + lo.kl_z, ws + lo.kl_l, B, c.kl_weight,
                              c.gene_loss_coeff, ws + lo.scalars, ws + lo.recon, stream));
  return VAE_OK;
}

// After vae_elbo_step_forward on the same workspace: the decoder's mean px_rate (B, G), which the
// validation step exports (task.py:240,263).  The logits are still in the workspace until a backward runs.
extern "C" int vae_elbo_step_px_rate(const vae_step_config* cfg, const float* ws, int64_t ws_bytes, float* px_rate,
                                     int64_t ld_r, void* stream) {
  Layout lo;
  if (!cfg || !ws || !px_rate) return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_px_rate: null pointer");
  if (!make_layout(*cfg, &lo)) return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_px_rate: bad configuration");
  if (ws_bytes < lo.total * 4) return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_px_rate: workspace too small");
  const int64_t NH = cfg->zinb ? 2 * (int64_t)cfg->G : cfg->G;
  return vae_px_rate_ex(bf16_storage(*cfg) ? VAE_L_BF16 : VAE_L_F32, ws + lo.logits, NH, ws + lo.library, ws + lo.rowmax,
                        ws + lo.rowsum, cfg->B, cfg->G, px_rate, ld_r, stream);
}

extern "C" int vae_elbo_step_backward(const vae_step_config* cfg, const vae_step_tensors* p, const vae_step_tensors* g,
                                      const void* xv, int64_t ldx, const float* eps_z, const float* eps_l,
                                      int use_dz_extra, float* ws, int64_t ws_bytes, void* event_heads_ready,
                                      void* stream) {
  return vae_elbo_step_backward_events(cfg, p, g, xv, ldx, eps_z, eps_l, use_dz_extra, ws, ws_bytes, event_heads_ready, nullptr,
                                       stream);
}

extern "C" int vae_elbo_step_backward_events(const vae_step_config* cfg, const vae_step_tensors* p, const vae_step_tensors* g,
                                             const void* xv, int64_t ldx, const float* eps_z, const float* eps_l,
                                             int use_dz_extra, float* ws, int64_t ws_bytes, void* event_heads_ready,
                                             void* event_small_ready, void* stream) {
  return vae_elbo_step_backward_slabs(cfg, p, g, xv, ldx, eps_z, eps_l, use_dz_extra, ws, ws_bytes, event_heads_ready,
                                      event_small_ready, nullptr, nullptr, 0, stream);
}

extern "C" int vae_elbo_step_backward_slabs(const vae_step_config* cfg, const vae_step_tensors* p, const vae_step_tensors* g,
                                            const void* xv, int64_t ldx, const float* eps_z, const float* eps_l,
                                            int use_dz_extra, float* ws, int64_t ws_bytes, void* event_heads_ready,
                                            void* event_small_ready, const int64_t* enc_row_bounds, void* const* enc_slab_events,
                                            int n_enc_slabs, void* stream) {
  Layout lo;
  TRY(check_common(cfg, p, g, xv, ldx, eps_z, eps_l, ws, ws_bytes, &lo, "vae_elbo_step_backward"));
  const vae_step_config& c = *cfg;
  Ctx k{c, ws, lo, stream, p, false};
  k.xd = xdirect(c, k, xv, ldx);
  if (k.xd) {   // (a uint8 batch: the forward's fp32 copy is still in the workspace -- unless the products read the uint8 rows)
    set_count_rows(k, xv, ldx);
    k.xu8 = xcounts_u8(c, k);
  }
  const int64_t B = c.B, G = c.G, H = c.H, Hl = c.Hl, L = c.L, Hd = c.Hd, HE = c.H + c.Hl;
  const int64_t NH = c.zinb ? 2 * G : G;
  // d loss / d ll[b] = -coeff/B ; d loss / d kl[b] = coeff*kl_weight/B   (task.py:202,489)
  const float g_ll = -c.gene_loss_coeff / (float)B * c.loss_scale;
  const float g_kl = c.gene_loss_coeff * c.kl_weight / (float)B * c.loss_scale;
  float* logits = ws + lo.logits;
  void* lp = !c.zinb ? nullptr : (k.hb() ? (void*)((uint16_t*)logits + G) : (void*)(logits + G));

  // epilogue backward: logits := dlogits (in place); d px_r and the head-bias gradients are overwritten
  // its column sums (d px_r, head-bias gradients) stay as per-chunk slabs; on the fused-middle path the last launch of
  // the backward reduces them (one launch less), else a reduce launch follows at once
  VaeDeferredColreduce cr{};
  // the forward of a defer_elbo_reduce step left its per-strip partials un-reduced (B <= 1024): this launch reduces them
  const VaeElboFromParts ef{ws + lo.fwd_part, vae_nbzinb_fwd_strips(G), ws + lo.ll, ws + lo.S, ws + lo.recon, ws + lo.scalars,
                            ws + lo.kl_z, ws + lo.kl_l, c.kl_weight, c.gene_loss_coeff};
  TRY(vae_nbzinb_bwd_ex(c.x_dtype, k.ldt(), c.zinb, logits, NH, lp, NH, xv, ldx, ws + lo.library, ws + lo.table, ws + lo.rowmax,
                        ws + lo.rowsum, ws + lo.S, nullptr, g_ll, B, G, logits, NH, lp, NH, ws + lo.dlibrary, g->px_r, g->heads_b,
                        c.zinb ? g->heads_b + G : nullptr, ws + lo.bwd_slabs, vae_nbzinb_bwd_workspace_bytes(B, G),
                        k.am(AM_DLOGITS), vae_mid_supported(c) ? &cr : nullptr, stream,
                        c.defer_elbo_reduce && B <= 1024 ? &ef : nullptr));
  const float* px = ws + lo.dec_h[c.n_layers - 1];
  // dW = dlogits^T px.  In VAE_FP32_F16X2 this ONE product keeps the exact three-bf16-term split: the magnitude of a gene's
  // column of dlogits follows the gene's softmax mass, which spans > 2^20 across genes, and ONE power-of-two scale for the
  // whole operand (the two-f16-term scheme) would leave the weight-gradient ROWS of quiet genes with few correct bits --
  // invisible in a tensor-wide norm, but AdamW normalises per element (tests/test_configs_gpu.py, row accuracy).  The
  // dgrad below sums over genes, where the quiet ones do not matter, and every other operand of the step is uniform.
  if (k.hb())   // px16: the bf16 image the forward made
    TRY(vae_gemm_bf16io(VAE_TN, VAE_IO_A_BF16 | VAE_IO_B_BF16, NH, Hd, B, logits, NH, k.h16(lo.px16), Hd, g->heads_w, Hd, nullptr,
                        ws + lo.scratch, lo.scratch_floats * 4, nullptr, nullptr, stream));
  else if (k.fx() && vae_gemm_tile_pipeline_ok(VAE_TN, NH, Hd, B) && !getenv("VAE_WGRAD_EXACT")) {
    // two f16 terms with one scale PER GENE: the backward epilogue left per-chunk maxima of every gene's column of both
    // gradient arrays in its slabs (rows 3, 4), the product derives a power of two per output row from them
    const VaeGemmRowScale rsc{ws + lo.bwd_slabs + 3 * G, c.zinb ? ws + lo.bwd_slabs + 4 * G : nullptr, G, vae_nbzinb_bwd_chunks(B, G),
                              (int64_t)VAE_SLAB_ROWS * G};
    TRY(vae_gemm_f16x2_rowscaled(NH, Hd, B, logits, NH, &rsc, px, Hd, k.am(AM_PX), k.an(AM_PX), g->heads_w, Hd, ws + lo.scratch,
                                 lo.scratch_floats * 4, stream));
  } else
    TRY(k.gemm(VAE_TN, NH, Hd, B, logits, NH, px, Hd, g->heads_w, Hd, nullptr, nullptr, 0));
  if (event_heads_ready) {
    if (hipEventRecord((hipEvent_t)event_heads_ready, (hipStream_t)stream) != hipSuccess)
      return vae_set_error(VAE_ERR_LAUNCH, "vae_elbo_step_backward: hipEventRecord failed");
  }
  if (vae_mid_supported(c)) {
    int nslab = 0;
    if (k.hb())
      TRY(vae_gemm_bf16io(VAE_NN, VAE_IO_A_BF16 | VAE_IO_B_BF16, B, Hd, NH, logits, NH, k.w_heads16(), Hd, nullptr, Hd, nullptr,
                          ws + lo.scratch, lo.scratch_floats * 4, &nslab, nullptr, stream));
    else
      TRY(k.big(VAE_NN, B, Hd, NH, logits, NH, AM_DLOGITS, p->heads_w, Hd, AM_WHEADS, nullptr, Hd, nullptr, &nslab));
    TRY(vae_mid_backward(c, p, g, ws + lo.scratch, nslab, eps_z, eps_l, use_dz_extra ? ws + lo.dz_extra : nullptr, g_kl,
                         ws + lo.dec_pre[0], ws + lo.ln_md[0], ws + lo.ln_rd[0], ws + lo.heads_z, ws + lo.heads_l,
                         ws + lo.qz_v, ws + lo.ql_v, ws + lo.dlibrary, ws + lo.pre_enc0, ws + lo.ln_mz[0], ws + lo.ln_rz[0],
                         ws + lo.ln_ml, ws + lo.ln_rl, ws + lo.z, ws + lo.h_enc0, ws + lo.dbet_d, ws + lo.dgam_d,
                         ws + lo.dpx, ws + lo.dheads_z, ws + lo.dheads_l, ws + lo.dbet_e, ws + lo.dgam_e,
                         ws + lo.dpre_enc0, k.am(AM_DPRE), k.hb() ? k.h16(lo.dpre16) : nullptr, &cr,
                         k.xd ? ws + lo.mean : nullptr, k.xd ? ws + lo.dpre_s : nullptr, stream));
  } else {
    // The row kernels of the unfused backward leave their LayerNorm / bias column sums per workgroup in lo.colpart; ONE
    // launch at the end adds them in workgroup order and overwrites the gradients: no float atomics, no clearing, and the
    // same bits on every run.
    static_assert(3 * (2 * VAE_MAX_LAYERS + 1) + 2 <= VAE_COLPART_MAX, "one entry per LayerNorm / bias vector of the deepest model");
    VaeColPartList cp;
    cp.count = 0;
    int64_t cp_used = 0;
    float* cp_cur = nullptr;
    int cp_blocks = 0;
    auto cp_take = [&](int64_t floats) {   // the next partial buffer (capacity is what make_layout summed)
      cp_cur = cp_used + floats <= lo.colpart_floats ? ws + lo.colpart + cp_used : nullptr;
      cp_used += floats;
      return cp_cur ? VAE_OK : vae_set_error(VAE_ERR_ARG, "vae_elbo_step_backward: column-sum workspace too small");
    };
    auto cp_add = [&](int64_t first, int64_t width, int64_t stride, float* out) {
      if (!out) return VAE_OK;
      if (cp.count >= VAE_COLPART_MAX) return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_backward: too many gradient segments");
      cp.part[cp.count] = cp_cur + first; cp.out[cp.count] = out; cp.blocks[cp.count] = cp_blocks;
      cp.width[cp.count] = (int)width; cp.stride[cp.count] = stride; ++cp.count;
      return VAE_OK;
    };
    const int64_t cp_nb = vae_colpart_blocks(B);
    // GELU/LayerNorm backward of one layer: dgamma / dbeta / dbias through the partial buffer
    auto ln_bwd = [&](const float* dout, int64_t ld_dout, const float* pre, int64_t ld_pre, const float* mean, const float* rstd,
                      const float* gamma, int64_t W, float* dpre, int64_t ld_dpre, float* dgamma, float* dbeta, float* dbias,
                      uint16_t* dpre16 = nullptr, int64_t ld16 = 0) {
      TRY(cp_take(cp_nb * 3 * W));
      TRY(vae_gelu_ln_bwd_part(dout, ld_dout, pre, ld_pre, mean, rstd, gamma, B, W, dpre, ld_dpre, dgamma, dbeta, dbias, cp_cur,
                               &cp_blocks, dpre16, ld16, stream));
      TRY(cp_add(0, W, 3 * W, dgamma)); TRY(cp_add(W, W, 3 * W, dbeta)); TRY(cp_add(2 * W, W, 3 * W, dbias));
      return VAE_OK;
    };
    if (k.hb())
      TRY(vae_gemm_bf16io(VAE_NN, VAE_IO_A_BF16 | VAE_IO_B_BF16, B, Hd, NH, logits, NH, k.w_heads16(), Hd, ws + lo.dpx, Hd, nullptr,
                          ws + lo.scratch, lo.scratch_floats * 4, nullptr, nullptr, stream));
    else
      TRY(k.big(VAE_NN, B, Hd, NH, logits, NH, AM_DLOGITS, p->heads_w, Hd, AM_WHEADS, ws + lo.dpx, Hd, nullptr, nullptr));

    // decoder hidden layers
    float* dcur = ws + lo.dpx;
    for (int n = c.n_layers - 1; n >= 0; --n) {
      const float* inp = n == 0 ? ws + lo.z : ws + lo.dec_h[n - 1];
      const int64_t kin = n == 0 ? L : Hd;
      TRY(k.drop(dcur, Hd, Hd, VAE_SITE_DEC(n)));
      TRY(ln_bwd(dcur, Hd, ws + lo.dec_pre[n], Hd, ws + lo.ln_md[n], ws + lo.ln_rd[n], p->dec_ln_g[n], Hd, dcur, Hd,
                 g->dec_ln_g[n], g->dec_ln_b[n], g->dec_b[n]));
      const int64_t ldwd = n == 0 ? ld_dec_w0(c) : Hd;
      TRY(k.gemm(VAE_TN, Hd, kin, B, dcur, Hd, inp, kin, g->dec_w[n], ldwd, nullptr, nullptr, 0));
      if (n == 0 && k.cov())   // gradient of the covariate columns: dpre^T cov
        TRY(k.gemm(VAE_TN, Hd, c.n_cov, B, dcur, Hd, c.batch_cov, c.ld_cov, g->dec_w[0] + L, ldwd, nullptr, nullptr, 0));
      if (n > 0) {
        // dgrad into the scratch-free buffer of the layer below: reuse dec_pre[n] (no longer needed)
        float* nxt = ws + lo.dec_pre[n];
        TRY(k.gemm(VAE_NN, B, Hd, Hd, dcur, Hd, p->dec_w[n], Hd, nxt, Hd, nullptr, nullptr, 0));
        dcur = nxt;
      } else {
        TRY(k.gemm(VAE_NN, B, L, Hd, dcur, Hd, p->dec_w[0], ldwd, ws + lo.dz, L, nullptr, nullptr, 0));
      }
    }
    // reparameterisation / KL backward (cell-head gradient enters through dz_extra)
    const float* dz = ws + lo.dz;
    if (use_dz_extra) TRY(vae_add_inplace(ws + lo.dz_extra, ws + lo.dz, B * L, stream));
    TRY(cp_take(cp_nb * (L + 1)));
    TRY(vae_reparam_kl_bwd_part(dz, L, nullptr, g_kl, ws + lo.heads_z, L + 1, eps_z, L, ws + lo.qz_v, B, L, 0.f, 1.f,
                                ws + lo.dheads_z, L + 1, g->hz_b, cp_cur, &cp_blocks, stream));
    TRY(cp_add(0, L + 1, L + 1, g->hz_b));
    TRY(cp_take(cp_nb * 2));
    TRY(vae_reparam_kl_bwd_part(ws + lo.dlibrary, 1, nullptr, g_kl, ws + lo.heads_l, 2, eps_l, 1, ws + lo.ql_v, B, 1, c.l_mean,
                                c.l_var, ws + lo.dheads_l, 2, g->hl_b, cp_cur, &cp_blocks, stream));
    TRY(cp_add(0, 2, 2, g->hl_b));
    const float* hz = c.n_layers > 1 ? ws + lo.encz_h[c.n_layers - 1] : ws + lo.h_enc0;
    const int64_t ld_hz = c.n_layers > 1 ? H : HE;
    TRY(k.gemm(VAE_TN, L + 1, H, B, ws + lo.dheads_z, L + 1, hz, ld_hz, g->hz_w, H, nullptr, nullptr, 0));
    TRY(k.gemm(VAE_TN, 2, Hl, B, ws + lo.dheads_l, 2, ws + lo.h_enc0 + H, HE, g->hl_w, Hl, nullptr, nullptr, 0));
    TRY(k.gemm(VAE_NN, B, H, L + 1, ws + lo.dheads_z, L + 1, p->hz_w, H, ws + lo.dh, H, nullptr, nullptr, 0));
    TRY(k.gemm(VAE_NN, B, Hl, 2, ws + lo.dheads_l, 2, p->hl_w, Hl, ws + lo.dpre_enc0 + H, HE, nullptr, nullptr, 0));

    // deeper z-encoder layers
    float* dh = ws + lo.dh;
    for (int n = c.n_layers - 1; n >= 1; --n) {
      const float* inp = n == 1 ? ws + lo.h_enc0 : ws + lo.encz_h[n - 1];
      const int64_t ld_in = n == 1 ? HE : H;
      TRY(k.drop(dh, H, H, VAE_SITE_ENC_Z(n)));
      TRY(ln_bwd(dh, H, ws + lo.encz_pre[n], H, ws + lo.ln_mz[n], ws + lo.ln_rz[n], p->encz_ln_g[n], H, dh, H, g->encz_ln_g[n],
                 g->encz_ln_b[n], g->encz_b[n]));
      TRY(k.gemm(VAE_TN, H, H, B, dh, H, inp, ld_in, g->encz_w[n], H, nullptr, nullptr, 0));
      float* nxt = ws + lo.encz_pre[n];  // free from here on
      TRY(k.gemm(VAE_NN, B, H, H, dh, H, p->encz_w[n], H, nxt, H, nullptr, nullptr, 0));
      dh = nxt;
    }
    // first encoder layer: GELU/LayerNorm backward of both halves, then ONE wgrad over the counts
    TRY(k.drop(dh, H, H, VAE_SITE_ENC_Z(0)));
    // (a bf16-storage step: dpre again as bf16 for the weight-gradient product over the counts)
    uint16_t* d16 = k.hb() ? k.h16(lo.dpre16) : nullptr;
    TRY(ln_bwd(dh, H, ws + lo.pre_enc0, HE, ws + lo.ln_mz[0], ws + lo.ln_rz[0], p->encz_ln_g[0], H, ws + lo.dpre_enc0, HE,
               g->encz_ln_g[0], g->encz_ln_b[0], g->enc_b0, d16, HE));
    TRY(ln_bwd(ws + lo.dpre_enc0 + H, HE, ws + lo.pre_enc0 + H, HE, ws + lo.ln_ml, ws + lo.ln_rl, p->encl_ln_g, Hl,
               ws + lo.dpre_enc0 + H, HE, g->encl_ln_g, g->encl_ln_b, g->enc_b0 ? g->enc_b0 + H : nullptr, d16 ? d16 + H : nullptr, HE));
    TRY(vae_colpart_reduce(cp, stream));
    if (k.fx()) {
      TRY(vae_absmax_partials(ws + lo.dpre_enc0, B, HE, HE, k.am(AM_DPRE), stream));
    }
  }
  // every gradient except the first-layer encoder weights is final from here on (second all-reduce bucket)
  if (event_small_ready) {
    if (hipEventRecord((hipEvent_t)event_small_ready, (hipStream_t)stream) != hipSuccess)
      return vae_set_error(VAE_ERR_LAUNCH, "vae_elbo_step_backward: hipEventRecord failed");
  }
  if (k.cov())   // covariate columns of both encoders' first-layer weight gradient: dpre^T cov (small; before the release point
                 // it would delay -- the whole enc_w0 slice travels last anyway)
    TRY(k.gemm(VAE_TN, HE, c.n_cov, B, ws + lo.dpre_enc0, HE, c.batch_cov, c.ld_cov, g->enc_w0 + G, k.ldw(), nullptr, nullptr, 0));
  // (dpre16, the bf16 image of dpre: left by the fused middle or by the first layer's LayerNorm backward launches)
  // The (H + Hl, G) weight gradient is the last and, for a data-parallel caller, the only exposed all-reduce bucket: with
  // n_enc_slabs > 1 it is produced in row slabs [enc_row_bounds[i], enc_row_bounds[i + 1]) -- contiguous ranges of the
  // gradient arena -- by one launch each, and enc_slab_events[i] is recorded behind slab i, so that slab i travels while
  // slab i + 1 is computed.  (One launch per slab fills the chip less well than one launch over all rows: a caller without
  // a reducer passes n_enc_slabs = 0.)
  int nsl = c.input_dropout_rate > 0.f ? 2 : 1;   // GeneDropout: one product per encoder, each over its own input
  const int64_t one[3] = {0, c.input_dropout_rate > 0.f ? H : HE, HE};
  const int64_t* rb = one;
  if (enc_row_bounds && n_enc_slabs > 0) {
    if (n_enc_slabs > 64 || enc_row_bounds[0] != 0 || enc_row_bounds[n_enc_slabs] != HE)
      return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_backward: enc_row_bounds must run from 0 to H + Hl in at most 64 slabs");
    for (int i = 0; i < n_enc_slabs; ++i)
      if (enc_row_bounds[i + 1] <= enc_row_bounds[i] || enc_row_bounds[i] % 4 != 0)
        return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_backward: enc_row_bounds must increase in multiples of 4");
    if (c.input_dropout_rate > 0.f) {   // GeneDropout: one product per encoder -- a slab must not straddle the two
      bool at_h = false;
      for (int i = 0; i <= n_enc_slabs; ++i) at_h = at_h || enc_row_bounds[i] == H;
      if (!at_h) return vae_set_error(VAE_ERR_ARG, "vae_elbo_step_backward: with input dropout a slab boundary must sit at row H");
    }
    nsl = n_enc_slabs;
    rb = enc_row_bounds;
  }
  bool u8_copied = false;
  for (int i = 0; i < nsl; ++i) {
    const int64_t r0 = rb[i], rows = rb[i + 1] - rb[i];
    const float* dpre_i = ws + lo.dpre_enc0 + r0;
    float* dw_i = g->enc_w0 + r0 * k.ldw();
    if (c.input_dropout_rate > 0.f) {  // GeneDropout: each encoder's weight gradient over its own input
      TRY(k.enc_wgrad(rows, dpre_i, HE, r0 < H ? ws + lo.xn_z : ws + lo.xn, r0 < H ? AM_XNZ : AM_XN, dw_i));
    } else if (k.xd) {   // dW = (dpre / mean)^T x - colsum(dpre) 1^T; colsum(dpre) = the first-layer bias gradient, final by now
      VaeGemmCounts xc{ws + lo.xflags, (int)(B * VAE_NORM_WAVES), g->enc_b0 + r0};   // (the flags the forward left: one term of x when all clear)
      bool u8_rows = k.xu8 && vae_gemm_counts_u8_tn_ok(rows, G, B);
      if (k.xu8 && !u8_rows && !u8_copied) {   // a row slab whose plan has no uint8 kernel: the fp32 copy after all, made once
        TRY(vae_cast_u8_f32(xv, ldx, B, G, ws + lo.xn, lo.ld_xn, stream));
        u8_copied = true;
      }
      xc.x_u8 = u8_rows ? 1 : 0;
      const float* xrows = u8_rows ? (const float*)xv : k.xf;
      const int64_t ldrows = u8_rows ? ldx : k.ldxf;
      TRY(vae_gemm_f16x2(VAE_TN, rows, G, B, ws + lo.dpre_s + r0, HE, k.am(AM_DPRE), k.an(AM_DPRE), xrows, ldrows, k.am(AM_XN),
                         k.an(AM_XN), dw_i, k.ldw(), nullptr, 0, ws + lo.scratch, lo.scratch_floats * 4, stream, nullptr, nullptr, &xc));
    } else {
      TRY(k.enc_wgrad(rows, dpre_i, HE, ws + lo.xn, AM_XN, dw_i));
    }
    if (enc_slab_events && rb != one && enc_slab_events[i]) {
      if (hipEventRecord((hipEvent_t)enc_slab_events[i], (hipStream_t)stream) != hipSuccess)
        return vae_set_error(VAE_ERR_LAUNCH, "vae_elbo_step_backward: hipEventRecord failed");
    }
  }
  return VAE_OK;
}
