// Fused decoder epilogue of the scVI-style VAE (gfx950): softmax over genes, library-size
// scaling and the NB / ZINB log-likelihood with its row reduction -- forward and backward.
//
// Reference semantics (all fp32):
//   px_scale = softmax_g(logit_scale)            modules.py:362,401
//   px_rate  = exp(library) * px_scale           modules.py:404
//   theta    = exp(px_r)                         task.py:481-483
//   ll       = log_zinb_positive / log_nb_positive(x, px_rate, theta, logit_dropout)
//                                                distributions.py:22-62, :65-90
//   recon    = -ll.sum(-1)                       task.py:156-165
//
// HBM-bound by design: forward reads (x, logit_scale, logit_dropout) exactly once = 12 B per
// cell*gene (NB: 8 B); backward reads them once more and writes both logit gradients in place
// = 20 B (NB: 12 B).  To keep the VALU under the HBM time:
//   * everything that depends on the gene only (theta, log(theta+eps), 1/(theta+eps),
//     lgamma(theta+k)-lgamma(theta)-lgamma(k+1) and digamma(theta+k)-digamma(theta) for counts
//     k < 16) is tabulated once per step ("gene table", L2-resident) -- no lgamma / digamma in the
//     per-element path for the counts that make up >99% of single-cell data;
//   * logs / exps / reciprocals use the hardware v_log_f32 / v_exp_f32 / v_rcp_f32 (1 ulp);
//     the softmax exponential carries a compensation term for the argument scaling.
//
// Forward: one 512-thread workgroup per cell keeps the scale-logit row in REGISTERS between the
// softmax statistics (wavefront shuffles + LDS) and the likelihood pass, and also reduces
// S_b = sum_g mu * dll/dmu, which makes the backward a single element-wise pass:
//   dlogit_scale = delta_b * (a_g - p_g * S_b).
// Backward: (strip of 512 genes: two per thread; 1024 with VAE_LL_BWD_EPT=4) x (chunk of cells) per workgroup; per-gene column sums
// (d px_r, both bias gradients) stay in registers across the chunk and are flushed with one
// float atomic per gene per workgroup; the next cell's loads are issued before the current
// cell's stores so that HBM latency is overlapped although the gradients alias the logits.
#include "../../include/vae_amd.h"
#include "common.h"
#include "internal.h"

#include <stdlib.h>

namespace {

constexpr float EPS = 1e-8f;
// Counts 0..KTAB-1 are served by the gene table.  32, not 16: a count beyond the table sends its WAVE through the closed-form
// branch (two Stirling lgammas / a digamma), and with 16 that was no rare event -- a lane's slot j covers 64 genes of a
// wave, and one gene among them whose counts reach 16 (0.9 % of the genes of the SURVEY 8(d) synthetic matrix, a handful
// of highly expressed genes in real data) took the slot's whole wave there in every cell: 42 % of the (wave, slot, cell)
// evaluations, ~30 of the forward's 92 instructions per element.  With 32 rows it is 8 %.
constexpr int KTAB = 32;
constexpr int KSEG = KTAB / 4;            // the table is built four counts per thread
constexpr int ROW_TH = 0, ROW_LTE = 1, ROW_ITE = 2, ROW_LG = 3, ROW_PSI = 4, ROW_LGD = 5, ROW_DGD = 5 + KTAB;
constexpr int TABLE_ROWS = 5 + 2 * KTAB;

// ---- hardware-rate math ------------------------------------------------------------------
__device__ __forceinline__ float flog(float v) { return __builtin_amdgcn_logf(v) * 0.69314718055994531f; }
__device__ __forceinline__ float fexp(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896341f); }
__device__ __forceinline__ float frcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float fexp_acc(float v) { return vae_exp_acc(v); }   // compensated exp (common.h)
// softmax weight exp(l - lse) of the streaming passes: ONE fused multiply-add (l * log2(e) - lse * log2(e), the second
// product precomputed per cell) and the hardware exp2.  |l - lse| <= ~25, so the single rounding of the argument is
// <= 2e-6 absolute = 1.3e-6 relative in the weight; the forward, the backward and the px_rate export use the SAME
// expression, so S_b, the gradients and the exported mean stay mutually consistent (the statistics keep the compensated exp).
constexpr float LOG2E = 1.44269504088896341f;
__device__ __forceinline__ float softmax_w(float l, float neg_lse_log2e) {
  return __builtin_amdgcn_exp2f(fmaf(l, LOG2E, neg_lse_log2e));
}
__device__ __forceinline__ float fsoftplus(float v) { return v > 20.f ? v : flog(1.f + fexp(v)); }
__device__ __forceinline__ float fsigmoid20(float v) {  // softplus'(v) as torch evaluates it
  const float z = fexp(fminf(v, 20.f));
  return v > 20.f ? 1.f : z * frcp(z + 1.f);
}

// Stirling series, z >= 8 (truncation < 3e-8)
__device__ __forceinline__ float lgamma_stirling(float z) {
  const float iz = frcp(z), iz2 = iz * iz;
  return (z - 0.5f) * flog(z) - z + 0.91893853320467274178f +
         iz * (0.083333333333f - iz2 * (0.0027777777778f - iz2 * 0.00079365079365f));
}
// the count-dependent term beyond the table, x >= KTAB (integer or not): both arguments are far above 8, plain Stirling
__device__ __forceinline__ float lgd_large(float x, float th, float lg_th) {
  return lgamma_stirling(x + th) - lg_th - lgamma_stirling(x + 1.f);
}
// lgamma for any z > 0: below 8 shift up by 8 with the recurrence (one extra log)
__device__ __forceinline__ float lgamma_any(float z) {
  if (z >= 8.f) return lgamma_stirling(z);
  float p = z;
#pragma unroll
  for (int j = 1; j < 8; ++j) p *= z + (float)j;
  return lgamma_stirling(z + 8.f) - flog(p);
}
__device__ __forceinline__ float digamma_asym(float z) {  // z >= 6
  const float iz = frcp(z), iz2 = iz * iz;
  return flog(z) - 0.5f * iz - iz2 * (0.083333333333f - iz2 * (0.0083333333333f - iz2 * 0.003968253968f));
}
__device__ __forceinline__ float digamma_any(float z) {
  if (z >= 6.f) return digamma_asym(z);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 6; ++j) s += frcp(z + (float)j);
  return digamma_asym(z + 6.f) - s;
}

// count-dependent terms of a count the table does not cover: beyond it (the usual case: plain asymptotics, no recurrence)
// or fractional inside its range (mixup)
__device__ __forceinline__ float lgd_beyond(float x, float th, float lg_th) {
  if (x >= (float)KTAB) return lgd_large(x, th, lg_th);
  return lgamma_any(x + th) - lg_th - lgamma_any(x + 1.f);
}
__device__ __forceinline__ float dgd_beyond(float x, float th, float psi_th) {
  if (x >= (float)KTAB) return digamma_asym(x + th) - psi_th;
  return digamma_any(x + th) - psi_th;
}

// ---- gene table --------------------------------------------------------------------------
// built with the accurate libm-grade functions: it runs over G genes once per step
__device__ __forceinline__ float digamma_exact(float z) {
  float r = 0.f;
  while (z < 6.f) {
    r -= 1.f / z;
    z += 1.f;
  }
  const float iz = 1.f / z, iz2 = iz * iz;
  return r + logf(z) - 0.5f * iz - iz2 * (0.083333333333f - iz2 * (0.0083333333333f - iz2 * 0.003968253968f));
}
// unit u < KSEG * G: counts [4 q, 4 q + 4) of gene g, q = u / G (unit 0 of a gene also writes its five base rows).  The two
// running sums at the unit's first count are formed without walking the counts before it one logarithm at a time:
// log of the rising-factorial ratio in chunks of four factors per logarithm (theta <= ~1e4: a chunk stays below 1e16;
// all terms of a gene's sum share a sign, so nothing cancels), the digamma difference as plain reciprocals.  At most
// 7 + 4 logarithms per thread where one thread per gene would walk 32.
__device__ __forceinline__ void gene_table_entry(const float* __restrict__ px_r, int G, float* __restrict__ T, int u) {
  const int q = u / G, g = u - q * G;
  if (q >= KSEG) return;
  const float th = expf(px_r[g]);
  if (q == 0) {
    T[ROW_TH * G + g] = th;
    // 1 / (theta + eps) by the SAME hardware instruction the kernels use (frcp), so that expressions built from it on
    // both sides of a difference agree bit for bit; log(theta + eps) is no longer subtracted from anything (elem_core)
    T[ROW_LTE * G + g] = logf(th + EPS);
    T[ROW_ITE * G + g] = frcp(th + EPS);
    T[ROW_LG * G + g] = lgammaf(th);
    T[ROW_PSI * G + g] = digamma_exact(th);
  }
  float lg = 0.f, dg = 0.f;  // k = 0
  for (int c = 0; c < q; ++c) {
    float p = 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float k = (float)(4 * c + i);
      p *= (th + k) / (k + 1.f);
      dg += 1.f / (th + k);
    }
    lg += logf(p);
  }
  for (int k = 4 * q; k < 4 * q + 4; ++k) {
    T[(ROW_LGD + k) * G + g] = lg;
    T[(ROW_DGD + k) * G + g] = dg;
    lg += logf((th + (float)k) / (float)(k + 1));
    dg += 1.f / (th + (float)k);
  }
}
__host__ __device__ inline unsigned gene_table_blocks(int64_t G) { return (unsigned)((KSEG * G + 255) / 256); }
__global__ void gene_table_kernel(const float* __restrict__ px_r, int G, float* __restrict__ T) {
  gene_table_entry(px_r, G, T, blockIdx.x * blockDim.x + threadIdx.x);
}
// First launch of a VAE_FP32_F16X2 forward: workgroups [0, VAE_ABSMAX_PARTIALS) scan the two gene-sized weight matrices
// for the partial maxima their GEMMs need (blocks [0, VAE_ABSMAX_W_SPLIT) -> part0 of w0, the rest -> part1 of w1), the
// workgroups after them build the gene table -- two independent jobs that otherwise cost a launch each.
__global__ __launch_bounds__(256) void step_prologue_kernel(const float* __restrict__ w0, int64_t n0, const float* __restrict__ w1,
                                                            int64_t n1, float* __restrict__ part0, float* __restrict__ part1,
                                                            const float* __restrict__ px_r, int G, float* __restrict__ T) {
  __shared__ float red[4];
  const int blk = blockIdx.x;
  if (blk >= VAE_ABSMAX_PARTIALS) {
    gene_table_entry(px_r, G, T, (blk - VAE_ABSMAX_PARTIALS) * 256 + threadIdx.x);
    return;
  }
  const bool first = blk < VAE_ABSMAX_W_SPLIT;
  const int bi = first ? blk : blk - VAE_ABSMAX_W_SPLIT;
  const float m = first ? vae_absmax_scan(w0, n0 >> 2, VAE_ABSMAX_W_SPLIT, bi, red)
                        : vae_absmax_scan(w1, n1 >> 2, VAE_ABSMAX_PARTIALS - VAE_ABSMAX_W_SPLIT, bi, red);
  if (threadIdx.x == 0) (first ? part0 : part1)[bi] = m;
}

// ---- one (cell, gene) element ------------------------------------------------------------
// The count-dependent table entry (row LGD+k / DGD+k of the gene, k = tab_index(x)) is loaded by
// the caller ahead of time, unconditionally, so that the L2 round trip overlaps other work;
// counts >= KTAB or non-integer take the (rare) closed-form branch here.
__device__ __forceinline__ int tab_index(float x) {  // 0 for zeros and for counts the table does not cover
  const int xi = (int)x;
  return (x > EPS && xi < KTAB && x == (float)xi) ? xi : 0;
}

// Count-indexed table entry: base = T + first_row * G + gene (hoisted by the caller), row k < 16: a 24-bit multiply
// (v_mul_u32_u24, full rate -- the 32-bit v_mul_lo_u32 is a quarter-rate instruction) and an unsigned offset.
__device__ __forceinline__ float tab_load(const float* __restrict__ base, int k, int G) {
#ifdef VAE_PROBE_NO_GATHER   // timing probe only (wrong results): what the count-indexed gathers cost
  return 0.f;
#endif
  return base[__umul24((unsigned)k, (unsigned)G)];
}

struct Fwd { float ll, a; };

// log(theta + eps) - log(theta + mu + eps) = -log1p(r), r = mu / (theta + eps), evaluated WITHOUT the cancellation of the
// reference's two logarithms: u = fl(1 + r), log1p(r) = log(u) + (r - (u - 1)) / u (first-order repair of the rounding of
// u; exact to O(ulp^2)).  The reference's fp32 form loses theta * ulp(log theta) ~ 1e-6 absolute in ptl, which is all of
// d ll / d pi = sigmoid(-pi) - sigmoid(ptl) for a quiet gene; this form keeps full relative accuracy of the difference,
// and a gene whose mu is below fp32's resolution of theta gives EXACTLY ptl = -pi, d pi = 0, as in torch.  Also returns
// log(theta + mu + eps) and 1 / (theta + mu + eps) = ite / u.  One log, one reciprocal, as before.
struct Core { float dlt, ltm, itm, rru; };   // rru = r / u = 1 - (theta + eps) / (theta + mu + eps)
__device__ __forceinline__ Core elem_core(float mu, float lte, float ite) {
  Core c;
  const float r = mu * ite;
  const float u = 1.f + r;
  const float ru = frcp(u);
  const float l1p = fmaf(__builtin_amdgcn_logf(u), 0.69314718055994531f, (r - (u - 1.f)) * ru);
  c.dlt = -l1p;
  c.ltm = lte + l1p;
  c.itm = ite * ru;
  c.rru = r * ru;
  return c;
}

// Transcendentals per element: exp(-pi), log(1 + e^-pi), log(1 + r), 1 / (1 + r), exp(ptl) and ONE shared log / reciprocal
// pair of s = (x > 0 ? mu + eps : 1 + e^ptl): a positive count needs log(mu + eps) and 1 / (mu + eps), a zero needs
// softplus(ptl) = log(1 + e^ptl) and its derivative e^ptl / (1 + e^ptl) -- never both.
template <bool ZINB>
__device__ __forceinline__ Fwd elem_fwd(float x, float mu, float pi, float th, float lte, float ite, float lgd_tab,
                                        const float* __restrict__ Tg, int G) {
  Fwd o;
  const Core k = elem_core(mu, lte, ite);
  const float ltm = k.ltm, itm = k.itm, dlt = k.dlt;
  const bool pos = x > EPS;
  float lgd = lgd_tab;
  if (pos && !(x < (float)KTAB && x == truncf(x)))  // rare: large or fractional count
    lgd = lgd_beyond(x, th, Tg[ROW_LG * G]);
  if (ZINB) {
    const float sp = fsoftplus(-pi);
    const float ptl = fmaf(th, dlt, -pi);
    const float z = fexp(fminf(ptl, 20.f));
    const bool big = ptl > 20.f;
    const float s = pos ? mu + EPS : 1.f + z;
    const float ls = flog(s), is = frcp(s);
    const float nzterm = x * (ls - ltm) + lgd;                  // pos: ls = log(mu + eps)
    const float a_nz = (x * is - (x + th) * itm) * mu;          // pos: is = 1 / (mu + eps)
    const float spz = big ? ptl : ls;                           // zero: softplus(ptl), threshold 20
    const float sgz = big ? 1.f : z * is;                       // zero: its derivative
    const bool zero = x < EPS;
    o.ll = pos ? (-sp + ptl + nzterm) : (zero ? spz - sp : 0.f);
    o.a = pos ? a_nz : (zero ? -sgz * th * itm * mu : 0.f);
  } else {
    const float s = mu + EPS;
    const float nzterm = x * (flog(s) - ltm) + lgd;
    const float a_nz = (x * frcp(s) - (x + th) * itm) * mu;
    o.ll = th * dlt + (pos ? nzterm : 0.f);
    o.a = pos ? a_nz : -th * itm * mu;
  }
  return o;
}

struct Bwd { float a, dpi, dth; };

template <bool ZINB>
__device__ __forceinline__ Bwd elem_bwd(float x, float mu, float pi, float th, float lte, float ite, float dgd_tab,
                                        const float* __restrict__ Tg, int G) {
  Bwd o;
  const Core k = elem_core(mu, lte, ite);
  const float itm = k.itm, dlt = k.dlt;
  // d ptl / d theta = (lte - ltm) + theta (1/(theta+eps) - 1/(theta+mu+eps)); the bracket is ite * r / u: no cancellation
  const float dptl_dth = fmaf(th * ite, k.rru, dlt);
  const bool pos = x > EPS;
  float dgd = dgd_tab;
  if (pos && !(x < (float)KTAB && x == truncf(x))) dgd = dgd_beyond(x, th, Tg[ROW_PSI * G]);
  const float a_nz = (x * frcp(mu + EPS) - (x + th) * itm) * mu;
  const float dth_nz = dptl_dth - x * itm + dgd;
  if (ZINB) {
    const float sg_pi = fsigmoid20(-pi);
    const float sg = fsigmoid20(fmaf(th, dlt, -pi));
    const bool zero = x < EPS;
    o.a = pos ? a_nz : (zero ? -sg * th * itm * mu : 0.f);
    o.dpi = pos ? sg_pi - 1.f : (zero ? sg_pi - sg : 0.f);
    o.dth = pos ? dth_nz : (zero ? sg * dptl_dth : 0.f);
  } else {
    o.dpi = 0.f;
    o.a = pos ? a_nz : -th * itm * mu;
    o.dth = pos ? dth_nz : dptl_dth;
  }
  return o;
}

// ---- two genes per instruction ------------------------------------------------------------------------------------
// gfx950 issues v_pk_{add,mul,fma}_f32 on register PAIRS at the rate of the scalar forms: the streaming passes evaluate a
// lane's four genes as two pairs.  Same expressions as elem_core / elem_fwd / elem_bwd, component-wise; the
// transcendentals (v_exp / v_log / v_rcp) and the selects stay per component.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 v; v.x = a; v.y = b; return v; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 rcp2(f2 v) { return mk2(frcp(v.x), frcp(v.y)); }
__device__ __forceinline__ f2 log2_2(f2 v) { return mk2(__builtin_amdgcn_logf(v.x), __builtin_amdgcn_logf(v.y)); }
__device__ __forceinline__ f2 exp2_2(f2 v) { return mk2(__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)); }
__device__ __forceinline__ f2 sel2(bool cx, bool cy, f2 a, f2 b) { return mk2(cx ? a.x : b.x, cy ? a.y : b.y); }
constexpr float LN2 = 0.69314718055994531f;

struct Core2 { f2 dlt, ltm, itm, rru; };
__device__ __forceinline__ Core2 elem_core2(f2 mu, f2 lte, f2 ite) {
  Core2 c;
  const f2 r = mu * ite;
  const f2 u = 1.f + r;
  const f2 ru = rcp2(u);
  const f2 l1p = fma2(log2_2(u), mk2(LN2, LN2), (r - (u - 1.f)) * ru);
  c.dlt = -l1p;
  c.ltm = lte + l1p;
  c.itm = ite * ru;
  c.rru = r * ru;
  return c;
}

struct Fwd2 { f2 ll, a; };
template <bool ZINB>
__device__ __forceinline__ Fwd2 elem_fwd2(f2 x, f2 mu, f2 pi, f2 th, f2 lte, f2 ite, f2 lgd, const float* __restrict__ Tg0,
                                          const float* __restrict__ Tg1, int G) {
  Fwd2 o;
  const Core2 k = elem_core2(mu, lte, ite);
  const bool px = x.x > EPS, py = x.y > EPS;
  if (px && !(x.x < (float)KTAB && x.x == truncf(x.x)))  // rare: large or fractional count
    lgd.x = lgd_beyond(x.x, th.x, Tg0[ROW_LG * G]);
  if (py && !(x.y < (float)KTAB && x.y == truncf(x.y)))
    lgd.y = lgd_beyond(x.y, th.y, Tg1[ROW_LG * G]);
  const f2 L2E2 = mk2(LOG2E, LOG2E), LN22 = mk2(LN2, LN2);
  if (ZINB) {
    const f2 npi = -pi;
    const f2 spl = log2_2(1.f + exp2_2(npi * L2E2)) * LN22;
    const f2 sp = sel2(npi.x > 20.f, npi.y > 20.f, npi, spl);
    const f2 ptl = fma2(th, k.dlt, npi);
    const f2 z = exp2_2(mk2(fminf(ptl.x, 20.f), fminf(ptl.y, 20.f)) * L2E2);
    const bool bx = ptl.x > 20.f, by = ptl.y > 20.f;
    const f2 s = sel2(px, py, mu + EPS, 1.f + z);
    const f2 ls = log2_2(s) * LN22, is = rcp2(s);
    const f2 nzterm = fma2(x, ls - k.ltm, lgd);                 // pos: ls = log(mu + eps)
    const f2 a_nz = (x * is - (x + th) * k.itm) * mu;           // pos: is = 1 / (mu + eps)
    const f2 spz = sel2(bx, by, ptl, ls);                       // zero: softplus(ptl), threshold 20
    const f2 sgz = sel2(bx, by, mk2(1.f, 1.f), z * is);         // zero: its derivative
    const bool zx = x.x < EPS, zy = x.y < EPS;
    const f2 llp = (ptl - sp) + nzterm, llz = spz - sp;
    const f2 az = -(sgz * th) * (k.itm * mu);
    o.ll = mk2(px ? llp.x : (zx ? llz.x : 0.f), py ? llp.y : (zy ? llz.y : 0.f));
    o.a = mk2(px ? a_nz.x : (zx ? az.x : 0.f), py ? a_nz.y : (zy ? az.y : 0.f));
  } else {
    const f2 s = mu + EPS;
    const f2 nzterm = fma2(x, log2_2(s) * LN22 - k.ltm, lgd);
    const f2 a_nz = (x * rcp2(s) - (x + th) * k.itm) * mu;
    const f2 base = th * k.dlt, az = -(th * k.itm) * mu;
    o.ll = mk2(base.x + (px ? nzterm.x : 0.f), base.y + (py ? nzterm.y : 0.f));
    o.a = sel2(px, py, a_nz, az);
  }
  return o;
}

struct Bwd2 { f2 a, dpi, dth; };
__device__ __forceinline__ f2 sigmoid20_2(f2 v) {   // fsigmoid20, component-wise
  const f2 z = exp2_2(mk2(fminf(v.x, 20.f), fminf(v.y, 20.f)) * mk2(LOG2E, LOG2E));
  const f2 q = z * rcp2(z + 1.f);
  return mk2(v.x > 20.f ? 1.f : q.x, v.y > 20.f ? 1.f : q.y);
}
template <bool ZINB>
__device__ __forceinline__ Bwd2 elem_bwd2(f2 x, f2 mu, f2 pi, f2 th, f2 lte, f2 ite, f2 dgd, const float* __restrict__ Tg0,
                                          const float* __restrict__ Tg1, int G) {
  Bwd2 o;
  const Core2 k = elem_core2(mu, lte, ite);
  const f2 dptl_dth = fma2(th * ite, k.rru, k.dlt);
  const bool px = x.x > EPS, py = x.y > EPS;
  if (px && !(x.x < (float)KTAB && x.x == truncf(x.x))) dgd.x = dgd_beyond(x.x, th.x, Tg0[ROW_PSI * G]);
  if (py && !(x.y < (float)KTAB && x.y == truncf(x.y))) dgd.y = dgd_beyond(x.y, th.y, Tg1[ROW_PSI * G]);
  const f2 a_nz = (x * rcp2(mu + EPS) - (x + th) * k.itm) * mu;
  const f2 dth_nz = (dptl_dth - x * k.itm) + dgd;
  if (ZINB) {
    const f2 sg_pi = sigmoid20_2(-pi);
    const f2 sg = sigmoid20_2(fma2(th, k.dlt, -pi));
    const bool zx = x.x < EPS, zy = x.y < EPS;
    const f2 az = -(sg * th) * (k.itm * mu), dpz = sg_pi - sg, dtz = sg * dptl_dth, dpp = sg_pi - 1.f;
    o.a = mk2(px ? a_nz.x : (zx ? az.x : 0.f), py ? a_nz.y : (zy ? az.y : 0.f));
    o.dpi = mk2(px ? dpp.x : (zx ? dpz.x : 0.f), py ? dpp.y : (zy ? dpz.y : 0.f));
    o.dth = mk2(px ? dth_nz.x : (zx ? dtz.x : 0.f), py ? dth_nz.y : (zy ? dtz.y : 0.f));
  } else {
    const f2 az = -(th * k.itm) * mu;
    o.dpi = mk2(0.f, 0.f);
    o.a = sel2(px, py, a_nz, az);
    o.dth = sel2(px, py, dth_nz, dptl_dth);
  }
  return o;
}

// ---- forward: one workgroup per cell ------------------------------------------------------
// All HBM loads of the cell (scale-logit row, x row, then the dropout-logit row) are issued up
// front; exp(l - max) is parked in LDS between the softmax statistics and the likelihood pass
// (each lane only ever touches its own slots: LDS is used as register overflow, conflict-free),
// which keeps the kernel at <= 128 VGPRs so that two workgroups share a CU and one loads while
// the other computes.
template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* __restrict__ row, int i, int G, float fill) {
  // elements [4i, 4i+4) of a row of G floats
  if (VEC) return reinterpret_cast<const float4*>(row)[i];
  float4 v;
  const int g = 4 * i;
  v.x = g < G ? row[g] : fill;
  v.y = g + 1 < G ? row[g + 1] : fill;
  v.z = g + 2 < G ? row[g + 2] : fill;
  v.w = g + 3 < G ? row[g + 3] : fill;
  return v;
}

// counts stored as uint8 (the reference's on-disk format, create_dataset.py:131-142): 4 B per lane
template <bool VEC>
__device__ __forceinline__ float4 ld4(const uint8_t* __restrict__ row, int i, int G, float fill) {
  if (VEC) {
    const uchar4 u = reinterpret_cast<const uchar4*>(row)[i];
    return make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
  }
  float4 v;
  const int g = 4 * i;
  v.x = g < G ? (float)row[g] : fill;
  v.y = g + 1 < G ? (float)row[g + 1] : fill;
  v.z = g + 2 < G ? (float)row[g + 2] : fill;
  v.w = g + 3 < G ? (float)row[g + 3] : fill;
  return v;
}

// logits kept as bf16 (a bf16-mode step: the reference's autocast leaves nn.Linear outputs in bf16, config.py:87):
// 8 B per lane; the element is the upper half of the fp32 pattern.  Vector path only (the step checks the alignment).
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf16_up(uint32_t h) { return __uint_as_float(h << 16); }
template <bool VEC>
__device__ __forceinline__ float4 ld4(const bf16_t* __restrict__ row, int i, int G, float fill) {
  if (VEC) {
    const uint2 u = reinterpret_cast<const uint2*>(row)[i];
    return make_float4(bf16_up(u.x & 0xffffu), __uint_as_float(u.x & 0xffff0000u), bf16_up(u.y & 0xffffu),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  float4 v;
  const int g = 4 * i;
  v.x = g < G ? bf16_up(row[g]) : fill;
  v.y = g + 1 < G ? bf16_up(row[g + 1]) : fill;
  v.z = g + 2 < G ? bf16_up(row[g + 2]) : fill;
  v.w = g + 3 < G ? bf16_up(row[g + 3]) : fill;
  return v;
}
__device__ __forceinline__ bf16_t bf16_rn(float v) { return __builtin_bit_cast(bf16_t, (__bf16)v); }   // nearest even
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
  uint2 u;
  u.x = (uint32_t)bf16_rn(a) | ((uint32_t)bf16_rn(b) << 16);
  u.y = (uint32_t)bf16_rn(c) | ((uint32_t)bf16_rn(d) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = bf16_rn(v); }

// EPT (4 or 2) consecutive elements of a row from element g0 on, for the streaming kernels whose threads own EPT genes:
// one 16 / 8-byte load of fp32, 4 / 2 bytes of uint8 counts, 8 / 4 bytes of bf16 logits (VEC), or guarded scalar loads.
template <bool VEC, int EPT>
__device__ __forceinline__ void ldv(const float* __restrict__ row, int g0, int G, float fill, float (&o)[EPT]) {
  if (VEC && EPT == 4) {
    const float4 v = *reinterpret_cast<const float4*>(row + g0);
    o[0] = v.x; o[1] = v.y; o[EPT - 2] = v.z; o[EPT - 1] = v.w;
  } else if (VEC) {
    const float2 v = *reinterpret_cast<const float2*>(row + g0);
    o[0] = v.x; o[1] = v.y;
  } else {
#pragma unroll
    for (int j = 0; j < EPT; ++j) o[j] = g0 + j < G ? row[g0 + j] : fill;
  }
}
template <bool VEC, int EPT>
__device__ __forceinline__ void ldv(const uint8_t* __restrict__ row, int g0, int G, float fill, float (&o)[EPT]) {
  if (VEC && EPT == 4) {
    const uchar4 u = *reinterpret_cast<const uchar4*>(row + g0);
    o[0] = (float)u.x; o[1] = (float)u.y; o[EPT - 2] = (float)u.z; o[EPT - 1] = (float)u.w;
  } else if (VEC) {
    const uchar2 u = *reinterpret_cast<const uchar2*>(row + g0);
    o[0] = (float)u.x; o[1] = (float)u.y;
  } else {
#pragma unroll
    for (int j = 0; j < EPT; ++j) o[j] = g0 + j < G ? (float)row[g0 + j] : fill;
  }
}
template <bool VEC, int EPT>
__device__ __forceinline__ void ldv(const bf16_t* __restrict__ row, int g0, int G, float fill, float (&o)[EPT]) {
  if (VEC && EPT == 4) {
    const uint2 u = *reinterpret_cast<const uint2*>(row + g0);
    o[0] = bf16_up(u.x & 0xffffu); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[EPT - 2] = bf16_up(u.y & 0xffffu); o[EPT - 1] = __uint_as_float(u.y & 0xffff0000u);
  } else if (VEC) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(row + g0);
    o[0] = bf16_up(u & 0xffffu); o[1] = __uint_as_float(u & 0xffff0000u);
  } else {
#pragma unroll
    for (int j = 0; j < EPT; ++j) o[j] = g0 + j < G ? bf16_up(row[g0 + j]) : fill;
  }
}
template <int EPT>
__device__ __forceinline__ void stv(float* p, const float (&v)[EPT]) {
  if (EPT == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[EPT - 2], v[EPT - 1]);
  else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}
template <int EPT>
__device__ __forceinline__ void stv(bf16_t* p, const float (&v)[EPT]) {
  if (EPT == 4) {
    uint2 u;
    u.x = (uint32_t)bf16_rn(v[0]) | ((uint32_t)bf16_rn(v[1]) << 16);
    u.y = (uint32_t)bf16_rn(v[EPT - 2]) | ((uint32_t)bf16_rn(v[EPT - 1]) << 16);
    *reinterpret_cast<uint2*>(p) = u;
  } else {
    *reinterpret_cast<uint32_t*>(p) = (uint32_t)bf16_rn(v[0]) | ((uint32_t)bf16_rn(v[1]) << 16);
  }
}

template <bool ZINB, bool VEC, int EPT, int THREADS, typename XT>
__global__ __launch_bounds__(THREADS, 4) void nbzinb_fwd_kernel(
    const float* __restrict__ ls, int64_t ld_s, const float* __restrict__ lp, int64_t ld_p, const XT* __restrict__ x,
    int64_t ld_x, const float* __restrict__ library, const float* __restrict__ T, int G,
    float* __restrict__ ll_out, float* __restrict__ rowmax, float* __restrict__ rowsum, float* __restrict__ S_out) {
  extern __shared__ __attribute__((aligned(16))) float4 srow[];  // EPT*THREADS float4, then NW floats
  constexpr int NW = THREADS / 64;
  float* red = reinterpret_cast<float*>(srow + EPT * THREADS);
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* r_s = ls + (int64_t)b * ld_s;
  const float* r_p = ZINB ? lp + (int64_t)b * ld_p : nullptr;
  const XT* r_x = x + (int64_t)b * ld_x;
  const int n4 = (G + 3) >> 2;
  const float NEG = -INFINITY;
  const float4 Z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // phase 1: the whole scale-logit row is requested at once (EPT x 16 B per lane in flight)
  float4 e[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int i = tid + j * THREADS;
    e[j] = i < n4 ? ld4<VEC>(r_s, i, G, NEG) : make_float4(NEG, NEG, NEG, NEG);
  }
  // the first two likelihood-pass operand sets are requested before the reductions so they travel meanwhile
  auto hbm = [&](int i, float4& xv, float4& pv) {
    xv = i < n4 ? ld4<VEC>(r_x, i, G, 0.f) : Z4;
    pv = (ZINB && i < n4) ? ld4<VEC>(r_p, i, G, 0.f) : Z4;
  };
  float4 xa, pa, xb, pb;
  hbm(tid, xa, pa);
  hbm(tid + THREADS, xb, pb);
  float m = NEG;
#pragma unroll
  for (int j = 0; j < EPT; ++j) m = fmaxf(fmaxf(m, fmaxf(e[j].x, e[j].y)), fmaxf(e[j].z, e[j].w));
  m = block_max<NW>(m, red);
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    float4 v;
    v.x = fexp_acc(e[j].x - m); v.y = fexp_acc(e[j].y - m);
    v.z = fexp_acc(e[j].z - m); v.w = fexp_acc(e[j].w - m);
    se += (v.x + v.y) + (v.z + v.w);
    srow[tid + j * THREADS] = v;  // own slot: LDS as register overflow, conflict-free
  }
  se = block_sum<NW>(se, red);
  const float scale = expf(library[b]) / se;  // mu = exp(library) * e / sum

  // phase 2: software-pipelined likelihood pass.  Iteration j evaluates float4 j while the HBM
  // operands of j+1 and the L2 table entries of j+1 (theta, count-indexed lgamma term) travel.
  auto tab = [&](const float4& xv, int i, float4& th, float4& lt, float (&lg)[4]) {
    const int ii = i < n4 ? i : 0;
    th = ld4<VEC>(T + ROW_TH * G, ii, G, 1.f);
    lt = ld4<VEC>(T + ROW_LTE * G, ii, G, 0.f);
    const float xc[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) lg[c] = T[(ROW_LGD + tab_index(xc[c])) * G + min(4 * ii + c, G - 1)];
  };
  // vmcnt retires in issue order, so per iteration the order is: (1) table loads of j+1 (their
  // x landed an iteration ago), (2) HBM loads of j+2, (3) evaluate j -- waiting for (1) next
  // iteration then never waits for the younger HBM loads of (2).
  float4 tha, thb, lta, ltb;
  float lga[4], lgb[4];
  tab(xa, tid, tha, lta, lga);
  float acc_ll = 0.f, acc_a = 0.f;
#pragma unroll 1
  for (int j = 0; j < EPT; ++j) {
    const int i = tid + j * THREADS;
    tab(xb, i + THREADS, thb, ltb, lgb);
    float4 xc, pc;
    hbm(j + 2 < EPT ? i + 2 * THREADS : n4, xc, pc);
    const float4 ev = srow[i];
    const int g = 4 * i;
    // 1 / (theta + eps): the instruction the gene table uses (ROW_ITE), recomputed here instead of a third table load
    Fwd f0 = elem_fwd<ZINB>(xa.x, scale * ev.x, pa.x, tha.x, lta.x, frcp(tha.x + EPS), lga[0], T + min(g, G - 1), G);
    Fwd f1 = elem_fwd<ZINB>(xa.y, scale * ev.y, pa.y, tha.y, lta.y, frcp(tha.y + EPS), lga[1], T + min(g + 1, G - 1), G);
    Fwd f2 = elem_fwd<ZINB>(xa.z, scale * ev.z, pa.z, tha.z, lta.z, frcp(tha.z + EPS), lga[2], T + min(g + 2, G - 1), G);
    Fwd f3 = elem_fwd<ZINB>(xa.w, scale * ev.w, pa.w, tha.w, lta.w, frcp(tha.w + EPS), lga[3], T + min(g + 3, G - 1), G);
    const bool v0 = g < G, v1 = g + 1 < G, v2 = g + 2 < G, v3 = g + 3 < G;
    acc_ll += (v0 ? f0.ll : 0.f) + (v1 ? f1.ll : 0.f) + (v2 ? f2.ll : 0.f) + (v3 ? f3.ll : 0.f);
    acc_a += (v0 ? f0.a : 0.f) + (v1 ? f1.a : 0.f) + (v2 ? f2.a : 0.f) + (v3 ? f3.a : 0.f);
    xa = xb; pa = pb; tha = thb; lta = ltb;
    xb = xc; pb = pc;
#pragma unroll
    for (int c = 0; c < 4; ++c) lga[c] = lgb[c];
  }
  acc_ll = block_sum<NW>(acc_ll, red);
  acc_a = block_sum<NW>(acc_a, red);
  if (tid == 0) {
    ll_out[b] = acc_ll;
    rowmax[b] = m;
    rowsum[b] = se;
    S_out[b] = acc_a;
  }
}

// ---- forward, streaming form: (strip of 1024 genes) x (chunk of cells) per workgroup ------------------------------
// The one-workgroup-per-cell kernel above loads a whole row, then reduces, then evaluates: at B = 512 every workgroup of
// the single generation is in the same phase at the same time, so HBM time and VALU time ADD (57 us for 123 MB).  Here
// the row statistics (max, sum of exponentials) are known before the pass starts -- from softmax_stats_kernel below or
// from the decoder-head GEMM's epilogue -- and the likelihood becomes one streaming pass tiled like the backward:
// ~1000 workgroups whose loads and arithmetic overlap, per-gene constants in registers, the same 3-stage software
// pipeline over the cells of a chunk.  Per-cell sums (ll and S_b) leave as per-strip partials, added up in strip order
// by the ELBO reduction (deterministic).
constexpr int BWD_STRIP = 1024;   // 256 threads x float4
constexpr int BWD_MAX_ROWS = 128; // cells per workgroup (per-cell constants staged in LDS)
constexpr int FWD_STRIP = BWD_STRIP;
constexpr int FWD_MAX_ROWS = 128; // cells per workgroup of the streaming forward
// (98 registers = 4 waves per SIMD; compiled for 5 (8 spills) or 6 (24 spills) the pass takes the same 40.6 us at
// B = 512: occupancy is not what bounds it -- profiles/r02_pmc_epilogue.txt)

// One workgroup per cell: running (max, sum of exp(v - max)) per thread over the row, merged across the workgroup.
template <bool VEC, typename LT>
__global__ __launch_bounds__(256) void softmax_stats_kernel(const LT* __restrict__ ls, int64_t ld_s, int G,
                                                            float* __restrict__ rowmax, float* __restrict__ rowsum) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const LT* r_s = ls + (int64_t)b * ld_s;
  const int n4 = (G + 3) >> 2;
  const float NEG = -INFINITY;
  float m = NEG, s = 0.f;
  for (int i0 = tid; i0 < n4; i0 += 4 * 256) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      v[u] = i < n4 ? ld4<VEC>(r_s, i, G, NEG) : make_float4(NEG, NEG, NEG, NEG);
    }
    float mx = m;
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = fmaxf(fmaxf(mx, fmaxf(v[u].x, v[u].y)), fmaxf(v[u].z, v[u].w));
    if (mx > m) { s *= fexp_acc(m - mx); m = mx; }   // first touch: s = 0
#pragma unroll
    for (int u = 0; u < 4; ++u)
      s += (fexp_acc(v[u].x - m) + fexp_acc(v[u].y - m)) + (fexp_acc(v[u].z - m) + fexp_acc(v[u].w - m));
  }
  const float M = block_max<4>(m, red);
  s = block_sum<4>(s * fexp_acc(m - M), red);
  if (tid == 0) { rowmax[b] = M; rowsum[b] = s; }
}

// rowmax / rowsum from the per-slot statistics the decoder-head GEMM's epilogue wrote: one WAVE per row, its lanes
// across the slots (every load in flight at once), shuffles for the two reductions; fixed order -> deterministic
// Workgroups past the (B + 3) / 4 merging ones build the gene table (px_r != null): two short, independent jobs that
// would otherwise cost a launch each (both are needed by the likelihood pass that follows).
__global__ __launch_bounds__(256) void softmax_stats_merge_kernel(const float* __restrict__ sm, const float* __restrict__ ss,
                                                                  int slots, int B, float* __restrict__ rowmax,
                                                                  float* __restrict__ rowsum, const float* __restrict__ px_r,
                                                                  int G, float* __restrict__ T) {
  const int merge_blocks = (B + 3) / 4;
  if ((int)blockIdx.x >= merge_blocks) {
    gene_table_entry(px_r, G, T, ((int)blockIdx.x - merge_blocks) * 256 + threadIdx.x);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float m = -INFINITY, s = 0.f, M;
  constexpr int PER_LANE = 8;
  if (slots <= 64 * PER_LANE) {   // every statistic of the row requested at once: one round trip instead of one per 64 slots and pass
    float vm[PER_LANE], vs[PER_LANE];
#pragma unroll
    for (int u = 0; u < PER_LANE; ++u) {
      const int t = lane + 64 * u;
      vm[u] = t < slots ? sm[(int64_t)t * B + b] : -INFINITY;
      vs[u] = t < slots ? ss[(int64_t)t * B + b] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PER_LANE; ++u) m = fmaxf(m, vm[u]);
    M = wave_max(m);
#pragma unroll
    for (int u = 0; u < PER_LANE; ++u)   // (the same terms in the same order as the loop below)
      if (lane + 64 * u < slots) s += vs[u] * fexp_acc(vm[u] - M);
  } else {
    for (int t = lane; t < slots; t += 64) m = fmaxf(m, sm[(int64_t)t * B + b]);
    M = wave_max(m);
    for (int t = lane; t < slots; t += 64) s += ss[(int64_t)t * B + b] * fexp_acc(sm[(int64_t)t * B + b] - M);
  }
  s = wave_sum(s);
  if (lane == 0) { rowmax[b] = M; rowsum[b] = s; }
}

// The cells [b0, b1) of grid row blockIdx.y: rows_per_block > 0 -- uniform chunks, the last one short; 0 -- the B cells dealt
// out evenly over the gridDim.y chunks, the first B % chunks of them one cell longer (bwd_grid).
__device__ __forceinline__ void chunk_cells(int B, int rows_per_block, int& b0, int& b1) {
  const int by = (int)blockIdx.y;
  if (rows_per_block > 0) {
    b0 = by * rows_per_block;
    b1 = min(B, b0 + rows_per_block);
  } else {
    const int c = (int)gridDim.y, q = B / c, r = B - q * c;
    b0 = by * q + min(by, r);
    b1 = b0 + q + (by < r ? 1 : 0);
  }
}

template <int EPT>
struct RowIn { float l[EPT], p[EPT], x[EPT]; };

template <bool ZINB, bool VEC, int EPT, typename XT, typename LT>
__device__ __forceinline__ RowIn<EPT> load_row(const LT* ls, int64_t ld_s, const LT* lp, int64_t ld_p,
                                               const XT* __restrict__ x, int64_t ld_x, int b, int g0, int G) {
  RowIn<EPT> r;
  ldv<VEC, EPT>(ls + (int64_t)b * ld_s, g0, G, 0.f, r.l);
  ldv<VEC, EPT>(x + (int64_t)b * ld_x, g0, G, 0.f, r.x);
  if (ZINB) ldv<VEC, EPT>(lp + (int64_t)b * ld_p, g0, G, 0.f, r.p);
  else {
#pragma unroll
    for (int j = 0; j < EPT; ++j) r.p[j] = 0.f;
  }
  return r;
}


template <bool ZINB, bool VEC, int EPT, typename XT, typename LT>
__global__ __launch_bounds__(256) void nbzinb_fwd_tiled_kernel(
    const LT* __restrict__ ls, int64_t ld_s, const LT* __restrict__ lp, int64_t ld_p, const XT* __restrict__ x,
    int64_t ld_x, const float* __restrict__ library, const float* __restrict__ T, const float* __restrict__ rowmax,
    const float* __restrict__ rowsum, int B, int G, int rows_per_block, float* __restrict__ part) {
  __shared__ float2 rc[FWD_MAX_ROWS];         // per cell: {-(max + log(sum exp)) log2(e), exp(library)}
  // per cell and wave: {sum ll, sum a}.  (Per-THREAD partials in LDS, reduced once at the end, were measured: the same
  // 41 us, and 32 KB of LDS per workgroup caps the occupancy at four workgroups.)
  __shared__ float2 red[FWD_MAX_ROWS][16];   // [wave * 4 + 16-lane row of the wave]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b0, b1;
  chunk_cells(B, rows_per_block, b0, b1);
  // lanes past the last gene keep running (the wave reductions need every lane): clamped addresses, zero weight
  const int g_raw = blockIdx.x * (256 * EPT) + tid * EPT;
  const int nv = g_raw < G ? min(EPT, G - g_raw) : 0;
  const int g0 = nv > 0 ? g_raw : 0;
  float th[EPT], lt[EPT], it[EPT];
  const float* Tg[EPT];
  const float* Tk[EPT];   // + first row of the count-indexed rows (tab_load)
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int g = min(g0 + j, G - 1);
    Tg[j] = T + g;
    Tk[j] = T + (int64_t)ROW_LGD * G + g;
    th[j] = T[ROW_TH * G + g];
    lt[j] = T[ROW_LTE * G + g];
    it[j] = T[ROW_ITE * G + g];
  }
  auto load = [&](int b) { return load_row<ZINB, VEC, EPT, XT, LT>(ls, ld_s, lp, ld_p, x, ld_x, b, g0, G); };
  // stage A (cell b+2): HBM loads; stage B (cell b+1): count-indexed table loads (L2); stage C (cell b): evaluate.
  // vmcnt retires in issue order: the table loads of b+1 are issued BEFORE the HBM loads of b+2.
  // The first two cells' rows are requested BEFORE the per-cell constants are formed: the workgroups of this grid are one
  // generation, and constants -> barrier -> rows -> table was a chain of three round trips ahead of the first evaluation.
  RowIn<EPT> c0 = load(b0);
  RowIn<EPT> c1 = b0 + 1 < b1 ? load(b0 + 1) : c0;
  for (int r = tid; r < b1 - b0; r += 256)
    rc[r] = make_float2(-(rowmax[b0 + r] + logf(rowsum[b0 + r])) * LOG2E, expf(library[b0 + r]));
  __syncthreads();
  float d0[EPT], d1[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) d1[j] = 0.f;
  {
#pragma unroll
    for (int j = 0; j < EPT; ++j) d0[j] = tab_load(Tk[j], tab_index(c0.x[j]), G);
  }
  for (int b = b0; b < b1; ++b) {
    if (b + 1 < b1) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) d1[j] = tab_load(Tk[j], tab_index(c1.x[j]), G);
    }
    RowIn<EPT> c2 = c1;
    if (b + 2 < b1) c2 = load(b + 2);
    const float2 k = rc[b - b0];
    const float (&l4)[EPT] = c0.l;
    const float (&p4)[EPT] = c0.p;
    const float (&x4)[EPT] = c0.x;
    float sll = 0.f, sa = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; j += 2) {              // two genes per packed instruction (elem_fwd2)
      const f2 p = mk2(softmax_w(l4[j], k.x), softmax_w(l4[j + 1], k.x));   // softmax weights, as the backward evaluates them
      const Fwd2 f = elem_fwd2<ZINB>(mk2(x4[j], x4[j + 1]), k.y * p, mk2(p4[j], p4[j + 1]), mk2(th[j], th[j + 1]),
                                     mk2(lt[j], lt[j + 1]), mk2(it[j], it[j + 1]), mk2(d0[j], d0[j + 1]), Tg[j], Tg[j + 1], G);
      sll += ((VEC || j < nv) ? f.ll.x : 0.f) + ((VEC || j + 1 < nv) ? f.ll.y : 0.f);   // VEC: all four genes valid, or none
      sa += ((VEC || j < nv) ? f.a.x : 0.f) + ((VEC || j + 1 < nv) ? f.a.y : 0.f);
    }
    // DPP butterfly inside the 16-lane rows; the four row totals of a wave go to LDS as they are (one 8-byte store by four
    // lanes) and are added after the cell loop in wave_sum_dpp's order -- the same bits, without the eight v_readlane and six
    // additions per cell that brought them through scalar registers
    sll = row_sum_dpp(nv > 0 ? sll : 0.f);
    sa = row_sum_dpp(nv > 0 ? sa : 0.f);
    if ((lane & 15) == 0) red[b - b0][wave * 4 + (lane >> 4)] = make_float2(sll, sa);
    c0 = c1;
    c1 = c2;
#pragma unroll
    for (int j = 0; j < EPT; ++j) d0[j] = d1[j];
  }
  __syncthreads();
  for (int r = tid; r < b1 - b0; r += 256) {   // this strip's share of ll[b] and S_b, waves in a fixed order
    float2 w4[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float2 q0 = red[r][4 * w], q1 = red[r][4 * w + 1], q2 = red[r][4 * w + 2], q3 = red[r][4 * w + 3];
      w4[w] = make_float2((q0.x + q1.x) + (q2.x + q3.x), (q0.y + q1.y) + (q2.y + q3.y));
    }
    part[((int64_t)blockIdx.x * 2 + 0) * B + b0 + r] = ((w4[0].x + w4[1].x) + w4[2].x) + w4[3].x;
    part[((int64_t)blockIdx.x * 2 + 1) * B + b0 + r] = ((w4[0].y + w4[1].y) + w4[2].y) + w4[3].y;
  }
}

// ll[b] = sum over strips of the partials, in strip order; likewise S[b]
__global__ __launch_bounds__(256) void fwd_partials_finish_kernel(const float* __restrict__ part, int strips, int B,
                                                                  float* __restrict__ ll, float* __restrict__ S) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float a = 0.f, c = 0.f;
  int s = 0;
  for (; s + 8 <= strips; s += 8) {   // sixteen loads in flight, added in strip order
    float qa[8], qc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { qa[u] = part[((int64_t)(s + u) * 2 + 0) * B + b]; qc[u] = part[((int64_t)(s + u) * 2 + 1) * B + b]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += qa[u]; c += qc[u]; }
  }
  for (; s < strips; ++s) {
    a += part[((int64_t)s * 2 + 0) * B + b];
    c += part[((int64_t)s * 2 + 1) * B + b];
  }
  ll[b] = a;
  S[b] = c;
}

// ---- px_rate export (validation / inference: LitVAE.validation_step, task.py:240,263) ------
// px_rate[b,g] = exp(library[b]) * exp(l[b,g] - max_b) / sum_b: the mu the likelihood pass above used
// (same expression, same rounding), from the row statistics it saved.  Grid (strips of 1024 genes, cells).
template <bool VEC, typename LT>
__global__ __launch_bounds__(256) void px_rate_kernel(const LT* __restrict__ ls, int64_t ld_s,
                                                      const float* __restrict__ library,
                                                      const float* __restrict__ rowmax,
                                                      const float* __restrict__ rowsum, int G, float* __restrict__ out,
                                                      int64_t ld_o) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;  // float4 index in the row
  if (4 * i >= G) return;
  const float m = -(rowmax[b] + logf(rowsum[b])) * LOG2E;
  const float scale = expf(library[b]);
  const float4 l = ld4<VEC>(ls + (int64_t)b * ld_s, i, G, 0.f);
  float4 r;
  r.x = scale * softmax_w(l.x, m); r.y = scale * softmax_w(l.y, m);
  r.z = scale * softmax_w(l.z, m); r.w = scale * softmax_w(l.w, m);
  float* o = out + (int64_t)b * ld_o;
  if (VEC) {
    reinterpret_cast<float4*>(o)[i] = r;
  } else {
    const int g = 4 * i;
    o[g] = r.x;
    if (g + 1 < G) o[g + 1] = r.y;
    if (g + 2 < G) o[g + 2] = r.z;
    if (g + 3 < G) o[g + 3] = r.w;
  }
}

// ---- backward: (gene strip of 256 x EPT) x (chunk of cells) per workgroup -------------------

// The ELBO reduction of a training step whose forward left it to the backward (vae_step_config.defer_elbo_reduce): the
// likelihood forward's per-strip partials part[(s * 2 + k) * B + b] (k = 0: ll, 1: S) are still in the workspace.  Every
// workgroup of the backward forms the S_b of its own cells from them (strip order, as vae_elbo_reduce_parts does), and the
// first workgroup writes ll / S / recon and the five scalars before it turns to its own tile (rowops.hip,
// elbo_reduce_kernel: same expressions; the sums over cells run over 256 threads here instead of 1024).
struct ElboParts {
  const float* part;   // null = off
  int strips;
  float *ll, *S, *recon, *out5;
  const float *kl_z, *kl_l;
  float kl_weight, coeff;
};
__device__ __forceinline__ float strip_sum(const float* __restrict__ p, int strips, int64_t st) {
  float v = 0.f;
  int s = 0;
  for (; s + 16 <= strips; s += 16) {   // sixteen loads in flight, summed in strip order
    float q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = p[(s + u) * st];
#pragma unroll
    for (int u = 0; u < 16; ++u) v += q[u];
  }
  for (; s + 8 <= strips; s += 8) {   // eight
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = p[(s + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += q[u];
  }
  for (; s + 4 <= strips; s += 4) {
    const float a = p[s * st], b2 = p[(s + 1) * st], c2 = p[(s + 2) * st], d2 = p[(s + 3) * st];
    v = (((v + a) + b2) + c2) + d2;
  }
  for (; s < strips; ++s) v += p[s * st];
  return v;
}
__device__ __forceinline__ void elbo_from_parts(const ElboParts& e, int B) {
  __shared__ float red4[4][4];
  const int64_t st = (int64_t)2 * B;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float l = strip_sum(e.part + i, e.strips, st), sb = strip_sum(e.part + B + i, e.strips, st);
    e.ll[i] = l;
    e.S[i] = sb;
    if (e.recon) e.recon[i] = -l;
    a -= l;
    b += e.kl_z[i];
    c += e.kl_l[i];
    d += e.kl_l[i] + e.kl_z[i];
  }
  a = wave_sum_dpp(a); b = wave_sum_dpp(b); c = wave_sum_dpp(c); d = wave_sum_dpp(d);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red4[0][w] = a; red4[1][w] = b; red4[2][w] = c; red4[3][w] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = b = c = d = 0.f;
    for (int w = 0; w < 4; ++w) { a += red4[0][w]; b += red4[1][w]; c += red4[2][w]; d += red4[3][w]; }
    const float n = (float)B;
    const float gl = a / n + e.kl_weight * (d / n);
    e.out5[0] = a / n;
    e.out5[1] = b / n;
    e.out5[2] = c / n;
    e.out5[3] = gl;
    e.out5[4] = e.coeff * gl;
  }
}

// EPT: genes per thread (4: a strip of 1024 genes per workgroup; 2: 512 -- fewer registers, more waves per SIMD)
template <bool ZINB, bool VEC, bool SLAB, int EPT, typename XT, typename LT>
__global__ __launch_bounds__(256) void nbzinb_bwd_kernel(
    const LT* ls, int64_t ld_s, const LT* lp, int64_t ld_p, const XT* __restrict__ x, int64_t ld_x,
    const float* __restrict__ library, const float* __restrict__ T, const float* __restrict__ rowmax,
    const float* __restrict__ rowsum, const float* __restrict__ S, const float* __restrict__ g_ll, float g_scalar, int B,
    int G, int rows_per_block, LT* dls, int64_t ld_ds, LT* dlp, int64_t ld_dp, float* __restrict__ dlibrary,
    float* __restrict__ dpx_r, float* __restrict__ dbias_s, float* __restrict__ dbias_p, float* __restrict__ slab,
    float* __restrict__ absmax, ElboParts ep) {
  // a deferred ELBO reduction is done by workgroup (0, 0) before its own tile: ~5 us of latency chain in a workgroup of the
  // first generation, which still ends before the kernel does.  (As an extra row of workgroups it was measured slower: last
  // in the grid it starts when the others finish and becomes the tail, +2 us; first in the grid, +7 us.)
  const int by = (int)blockIdx.y;
  if (ep.part && blockIdx.x == 0 && by == 0) elbo_from_parts(ep, B);
  __shared__ float4 rc[BWD_MAX_ROWS];  // per cell: {-(max + log(sumexp)) log2(e), exp(library), S_b, delta_b}
  __shared__ unsigned int wg_absmax;    // max |gradient| written by this workgroup (absmax != null), as float bits
  if (threadIdx.x == 0) wg_absmax = 0u;
  float am = 0.f;
  const int g0 = blockIdx.x * (256 * EPT) + threadIdx.x * EPT;
  int b0, b1;
  chunk_cells(B, rows_per_block, b0, b1);
  // Requested BEFORE the per-cell constants are formed (the workgroups of this grid are one generation; S_b from the
  // forward's partials -> barrier -> rows -> table was a chain of five round trips ahead of the first evaluation): the
  // per-gene constants and the first two cells' rows.  Threads past the last gene read the head of the row and leave
  // after the barrier.
  const bool live = g0 < G;
  const int gl = live ? g0 : 0;
  const int nv = min(EPT, G - gl);  // valid columns of this thread
  float th[EPT], lt[EPT], it[EPT], acc_t[EPT], acc_s[EPT], acc_p[EPT], max_s[EPT], max_p[EPT];
  const float* Tg[EPT];
  const float* Tk[EPT];   // + first row of the count-indexed rows (tab_load)
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int g = min(gl + j, G - 1);
    Tg[j] = T + g;
    Tk[j] = T + (int64_t)ROW_DGD * G + g;
    th[j] = T[ROW_TH * G + g]; lt[j] = T[ROW_LTE * G + g]; it[j] = T[ROW_ITE * G + g];
    acc_t[j] = acc_s[j] = acc_p[j] = max_s[j] = max_p[j] = 0.f;
  }
  // 3-stage software pipeline over the cells of the chunk:
  //   stage A (cell b+2): HBM loads of (logit_scale, logit_dropout, x)
  //   stage B (cell b+1): x has landed -> issue the count-indexed table loads (L2)
  //   stage C (cell b)  : evaluate, store both gradients (they alias the logits of cell b only)
  RowIn<EPT> c0 = load_row<ZINB, VEC, EPT, XT, LT>(ls, ld_s, lp, ld_p, x, ld_x, b0, gl, G);
  RowIn<EPT> c1 = c0;
  if (b0 + 1 < b1) c1 = load_row<ZINB, VEC, EPT, XT, LT>(ls, ld_s, lp, ld_p, x, ld_x, b0 + 1, gl, G);
  for (int r = threadIdx.x; r < b1 - b0; r += 256) {
    const int b = b0 + r;
    const float delta = g_scalar * (g_ll ? g_ll[b] : 1.f);
    const float Sb = ep.part ? strip_sum(ep.part + B + b, ep.strips, (int64_t)2 * B) : S[b];
    rc[r] = make_float4(-(rowmax[b] + logf(rowsum[b])) * LOG2E, expf(library[b]), Sb, delta);
    if (blockIdx.x == 0 && dlibrary) dlibrary[b] = delta * Sb;
  }
  __syncthreads();
  if (!live) return;
  float d0[EPT], d1[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) d1[j] = 0.f;
  {
#pragma unroll
    for (int j = 0; j < EPT; ++j) d0[j] = tab_load(Tk[j], tab_index(c0.x[j]), G);
  }
  for (int b = b0; b < b1; ++b) {
    // vmcnt retires in issue order: table loads of cell b+1 FIRST, HBM loads of cell b+2 after them
    if (b + 1 < b1) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) d1[j] = tab_load(Tk[j], tab_index(c1.x[j]), G);
    }
    RowIn<EPT> c2 = c1;
    if (b + 2 < b1) c2 = load_row<ZINB, VEC, EPT, XT, LT>(ls, ld_s, lp, ld_p, x, ld_x, b + 2, g0, G);
    const float4 k = rc[b - b0];
    const float (&l4)[EPT] = c0.l;
    const float (&p4)[EPT] = c0.p;
    const float (&x4)[EPT] = c0.x;
    float o_s[EPT], o_p[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j += 2) {              // two genes per packed instruction (elem_bwd2)
      const f2 p = mk2(softmax_w(l4[j], k.x), softmax_w(l4[j + 1], k.x));
      const f2 th2 = mk2(th[j], th[j + 1]);
      const Bwd2 e = elem_bwd2<ZINB>(mk2(x4[j], x4[j + 1]), k.y * p, mk2(p4[j], p4[j + 1]), th2, mk2(lt[j], lt[j + 1]),
                                     mk2(it[j], it[j + 1]), mk2(d0[j], d0[j + 1]), Tg[j], Tg[j + 1], G);
      const f2 os = k.w * (e.a - p * k.z), op = k.w * e.dpi, ot = (k.w * e.dth) * th2;
      o_s[j] = os.x; o_s[j + 1] = os.y;
      o_p[j] = op.x; o_p[j + 1] = op.y;
      if (!SLAB) {   // (with slabs the per-gene maxima below hold the same information: folded into am after the loop)
        if (j < nv) am = fmaxf(am, fmaxf(fabsf(os.x), ZINB ? fabsf(op.x) : 0.f));
        if (j + 1 < nv) am = fmaxf(am, fmaxf(fabsf(os.y), ZINB ? fabsf(op.y) : 0.f));
      }
      acc_t[j] += ot.x; acc_t[j + 1] += ot.y;
      acc_s[j] += os.x; acc_s[j + 1] += os.y;
      acc_p[j] += op.x; acc_p[j + 1] += op.y;
      if (SLAB) {   // per-gene maxima of the gradients this chunk writes (AS STORED: a bf16 store rounds first)
        max_s[j] = fmaxf(max_s[j], fabsf(os.x)); max_s[j + 1] = fmaxf(max_s[j + 1], fabsf(os.y));
        max_p[j] = fmaxf(max_p[j], fabsf(op.x)); max_p[j + 1] = fmaxf(max_p[j + 1], fabsf(op.y));
      }
    }
    LT* w_s = dls + (int64_t)b * ld_ds + g0;
    LT* w_p = ZINB ? dlp + (int64_t)b * ld_dp + g0 : nullptr;
    if (VEC) {
      stv<EPT>(w_s, o_s);
      if (ZINB) stv<EPT>(w_p, o_p);
    } else {
#pragma unroll
      for (int j = 0; j < EPT; ++j)
        if (j < nv) {
          st1(w_s + j, o_s[j]);
          if (ZINB) st1(w_p + j, o_p[j]);
        }
    }
    c0 = c1;
    c1 = c2;
#pragma unroll
    for (int j = 0; j < EPT; ++j) d0[j] = d1[j];
  }
  if (SLAB) {
#pragma unroll
    for (int j = 0; j < EPT; ++j)
      if (j < nv) am = fmaxf(am, fmaxf(max_s[j], ZINB ? max_p[j] : 0.f));
  }
  if (absmax) {  // threads past G have left: LDS integer max of the rest, then this workgroup's entry of the partial maxima
    atomicMax(&wg_absmax, __float_as_uint(am));
    __syncthreads();
    if (threadIdx.x == 0) absmax[by * gridDim.x + blockIdx.x] = __uint_as_float(wg_absmax);
  }
  if (SLAB) {
    // partial column sums of this cell chunk: plain coalesced stores into slab[chunk][3][G];
    // colreduce_kernel adds them up in chunk order (deterministic, no float atomics)
    float* sl = slab + (int64_t)by * VAE_SLAB_ROWS * G + g0;
#pragma unroll
    for (int j = 0; j < EPT; ++j)
      if (j < nv) {
        sl[j] = acc_t[j];
        sl[G + j] = acc_s[j];
        sl[3 * G + j] = max_s[j];
        if (ZINB) { sl[2 * G + j] = acc_p[j]; sl[4 * G + j] = max_p[j]; }
      }
  } else {
#pragma unroll
    for (int j = 0; j < EPT; ++j)
      if (j < nv) {
        atomicAdd(dpx_r + g0 + j, acc_t[j]);
        if (dbias_s) atomicAdd(dbias_s + g0 + j, acc_s[j]);
        if (ZINB && dbias_p) atomicAdd(dbias_p + g0 + j, acc_p[j]);
      }
  }
}

// out_k[g] (+)= sum_c slab[c][k][g], k = 0 (d px_r), 1 (d bias_scale), 2 (d bias_dropout)
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ slab, int chunks, int G, int nk,
                                                        float* __restrict__ o0, float* __restrict__ o1,
                                                        float* __restrict__ o2, int overwrite) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (g >= G || k >= nk) return;
  float* o = k == 0 ? o0 : (k == 1 ? o1 : o2);
  if (!o) return;
  vae_colreduce_column(slab, chunks, G, k, g, o, overwrite);
}

template <bool ZINB, bool VEC, int EPT, int THREADS, typename XT>
void launch_fwd(const float* ls, int64_t ld_s, const float* lp, int64_t ld_p, const XT* x, int64_t ld_x,
                const float* library, const float* T, int B, int G, float* ll, float* rowmax, float* rowsum, float* S,
                hipStream_t st, hipError_t* err) {
  const size_t shmem = (size_t)EPT * THREADS * sizeof(float4) + 64;
  auto kern = nbzinb_fwd_kernel<ZINB, VEC, EPT, THREADS, XT>;
  *err = hipSuccess;
  if (shmem > 64 * 1024)
    *err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (*err != hipSuccess) return;
  hipLaunchKernelGGL(kern, dim3(B), dim3(THREADS), shmem, st, ls, ld_s, lp, ld_p, x, ld_x, library, T, G, ll, rowmax,
                     rowsum, S);
}

template <bool ZINB, bool VEC, typename XT>
void dispatch_fwd(const float* ls, int64_t ld_s, const float* lp, int64_t ld_p, const XT* x, int64_t ld_x,
                  const float* library, const float* T, int B, int G, float* ll, float* rowmax, float* rowsum, float* S,
                  hipStream_t st, hipError_t* err) {
#define FWD_CASE(E, TH) launch_fwd<ZINB, VEC, E, TH, XT>(ls, ld_s, lp, ld_p, x, ld_x, library, T, B, G, ll, rowmax, rowsum, S, st, err)
  // (EPT, THREADS): 4*EPT*THREADS >= G; 512 threads while two workgroups fit a CU's LDS, else 1024
  if (G <= 1024) FWD_CASE(1, 256);
  else if (G <= 2048) FWD_CASE(1, 512);
  else if (G <= 4096) FWD_CASE(2, 512);
  else if (G <= 8192) FWD_CASE(4, 512);
  else if (G <= 12288) FWD_CASE(6, 512);
  else if (G <= 16384) FWD_CASE(8, 512);
  else if (G <= 20480) FWD_CASE(10, 512);
  else if (G <= 24576) FWD_CASE(6, 1024);
  else FWD_CASE(8, 1024);   // 128 KB of LDS + the reduction scratch: the most one workgroup can hold
#undef FWD_CASE
}

}  // namespace

extern "C" int64_t vae_gene_table_rows(void) { return TABLE_ROWS; }

extern "C" int vae_gene_table(const float* px_r, int64_t G, float* table, void* stream) {
  VAE_REQUIRE(px_r && table && G > 0, "vae_gene_table: bad arguments");
  hipLaunchKernelGGL(gene_table_kernel, dim3(gene_table_blocks(G)), dim3(256), 0, (hipStream_t)stream, px_r,
                     (int)G, table);
  VAE_CHECK_LAUNCH("vae_gene_table");
  return VAE_OK;
}

// part0[0 .. VAE_ABSMAX_W_SPLIT) / part1[0 .. VAE_ABSMAX_PARTIALS - VAE_ABSMAX_W_SPLIT) = partial maxima of |w0| / |w1|
// (n0, n1 multiples of 4, 16-byte aligned) and table = vae_gene_table(px_r), in one launch.
int vae_step_prologue(const float* w0, int64_t n0, const float* w1, int64_t n1, float* part0, float* part1, const float* px_r,
                      int64_t G, float* table, void* stream) {
  VAE_REQUIRE(w0 && w1 && part0 && part1 && n0 > 0 && n1 > 0 && n0 % 4 == 0 && n1 % 4 == 0 && vae_aligned16(w0) &&
                  vae_aligned16(w1) && px_r && table && G > 0, "vae_step_prologue: bad arguments");
  hipLaunchKernelGGL(step_prologue_kernel, dim3(VAE_ABSMAX_PARTIALS + gene_table_blocks(G)), dim3(256), 0,
                     (hipStream_t)stream, w0, n0, w1, n1, part0, part1, px_r, (int)G, table);
  VAE_CHECK_LAUNCH("vae_step_prologue");
  return VAE_OK;
}

// the one-workgroup-per-cell kernel parks a whole row in LDS (vae_nbzinb_fwd / _u8); the streaming form
// (vae_nbzinb_fwd_ws, vae_elbo_step_*) has no such limit
extern "C" int64_t vae_nbzinb_max_genes(void) { return 8 * 1024 * 4; }

template <typename XT>
static int nbzinb_fwd_impl(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                           const XT* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B, int64_t G,
                           float* ll, float* rowmax, float* rowsum, float* S, void* stream) {
  constexpr bool U8 = sizeof(XT) == 1;
  VAE_REQUIRE(logit_scale && x && library && gene_table && ll && rowmax && rowsum && S, "vae_nbzinb_fwd: null pointer");
  VAE_REQUIRE(!zinb || logit_dropout, "vae_nbzinb_fwd: ZINB needs logit_dropout");
  VAE_REQUIRE(B > 0 && G > 0 && G <= vae_nbzinb_max_genes(), "vae_nbzinb_fwd: G=%lld outside [1, %lld]", (long long)G,
              (long long)vae_nbzinb_max_genes());
  VAE_REQUIRE(ld_s >= G && ld_x >= G && (!zinb || ld_p >= G), "vae_nbzinb_fwd: leading dimension too small");
  const bool x_ok = U8 ? ((reinterpret_cast<uintptr_t>(x) & 3u) == 0) : vae_aligned16(x);
  const int vec = (G % 4 == 0) && (ld_s % 4 == 0) && (ld_x % 4 == 0) && (!zinb || ld_p % 4 == 0) &&
                  vae_aligned16(logit_scale) && x_ok && (!zinb || vae_aligned16(logit_dropout)) &&
                  vae_aligned16(gene_table);
  VAE_REQUIRE(!U8 || vec, "vae_nbzinb_fwd_u8: needs G %% 4 == 0, leading dimensions %% 4 == 0 and aligned rows");
  hipStream_t st = (hipStream_t)stream;
  hipError_t err = hipSuccess;
#define DISPATCH(Z, V, LP, LDP) dispatch_fwd<Z, V, XT>(logit_scale, ld_s, LP, LDP, x, ld_x, library, gene_table, (int)B, (int)G, ll, rowmax, rowsum, S, st, &err)
  if (U8) {  // uint8 counts: vector path only
    if (zinb) DISPATCH(true, true, logit_dropout, ld_p); else DISPATCH(false, true, nullptr, 0);
  } else if (zinb) { if (vec) DISPATCH(true, true, logit_dropout, ld_p); else DISPATCH(true, false, logit_dropout, ld_p); }
  else             { if (vec) DISPATCH(false, true, nullptr, 0); else DISPATCH(false, false, nullptr, 0); }
#undef DISPATCH
  if (err != hipSuccess) return vae_set_error(VAE_ERR_LAUNCH, "vae_nbzinb_fwd: cannot raise LDS limit: %s", hipGetErrorString(err));
  VAE_CHECK_LAUNCH("vae_nbzinb_fwd");
  return VAE_OK;
}

extern "C" int vae_nbzinb_fwd(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                              const float* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                              int64_t G, float* ll, float* rowmax, float* rowsum, float* S, void* stream) {
  return nbzinb_fwd_impl<float>(zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, B, G, ll, rowmax,
                                rowsum, S, stream);
}

extern "C" int vae_nbzinb_fwd_u8(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                                 const uint8_t* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                                 int64_t G, float* ll, float* rowmax, float* rowsum, float* S, void* stream) {
  return nbzinb_fwd_impl<uint8_t>(zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, B, G, ll,
                                  rowmax, rowsum, S, stream);
}

static bool logits_aligned(const void* p, int l_dtype) {
  return l_dtype == VAE_L_BF16 ? (reinterpret_cast<uintptr_t>(p) & 7u) == 0 : vae_aligned16(p);
}

int vae_px_rate_ex(int l_dtype, const void* logit_scale, int64_t ld_s, const float* library, const float* rowmax,
                   const float* rowsum, int64_t B, int64_t G, float* px_rate, int64_t ld_r, void* stream) {
  VAE_REQUIRE(logit_scale && library && rowmax && rowsum && px_rate, "vae_px_rate: null pointer");
  VAE_REQUIRE(B > 0 && B <= 65535 && G > 0 && ld_s >= G && ld_r >= G, "vae_px_rate: bad shape B=%lld G=%lld", (long long)B,
              (long long)G);
  const bool vec = (G % 4 == 0) && (ld_s % 4 == 0) && (ld_r % 4 == 0) && logits_aligned(logit_scale, l_dtype) && vae_aligned16(px_rate);
  const dim3 grid((unsigned)((G + 1023) / 1024), (unsigned)B);
  if (l_dtype == VAE_L_BF16) {
    VAE_REQUIRE(vec, "vae_px_rate: bf16 logits need G %% 4 == 0 and aligned rows");
    hipLaunchKernelGGL((px_rate_kernel<true, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logit_scale, ld_s,
                       library, rowmax, rowsum, (int)G, px_rate, ld_r);
  } else if (vec)
    hipLaunchKernelGGL((px_rate_kernel<true, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)logit_scale, ld_s,
                       library, rowmax, rowsum, (int)G, px_rate, ld_r);
  else
    hipLaunchKernelGGL((px_rate_kernel<false, float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)logit_scale, ld_s,
                       library, rowmax, rowsum, (int)G, px_rate, ld_r);
  VAE_CHECK_LAUNCH("vae_px_rate");
  return VAE_OK;
}

extern "C" int vae_px_rate(const float* logit_scale, int64_t ld_s, const float* library, const float* rowmax,
                           const float* rowsum, int64_t B, int64_t G, float* px_rate, int64_t ld_r, void* stream) {
  return vae_px_rate_ex(VAE_L_F32, logit_scale, ld_s, library, rowmax, rowsum, B, G, px_rate, ld_r, stream);
}

namespace {
// Genes per thread of the backward (nbzinb_bwd_kernel's EPT).  Four: 128 registers, four waves per SIMD.  Two: 80
// registers, six waves -- the kernel is bound by its arithmetic, yet loses 20 % when a fifth register over 128 takes a wave
// away, and gains 3-6 % from the two more (same box: cfg2 45.4 -> 42.9 us, cfg3 296 -> 288).  The forward was built the same
// way (66 registers, seven waves) and is no faster (35.3-37.4 against 35.2 us; 246-260 against 244): it stays at four.
int bwd_ept() {
  static const int e = getenv("VAE_LL_BWD_EPT") ? atoi(getenv("VAE_LL_BWD_EPT")) : 2;   // A/B knob
  return e == 4 ? 4 : 2;
}
void ll_grid(int64_t B, int64_t G, int strip, int target, int* strips, int* chunks, int* rows_per_block) {
  *strips = (int)((G + strip - 1) / strip);
  static const bool uniform = getenv("VAE_LL_UNIFORM_CHUNKS") != nullptr;   // A/B knob: the chunking before the balanced one
  // uniform chunks: ~4 workgroups per CU, every chunk rows_per_block cells, the last one what is left
  int c = (target + *strips - 1) / *strips;
  if (c > 64) c = 64;                        // bounds the slab reduction for small gene counts
  if (c > B) c = (int)B;
  int rpb = (int)((B + c - 1) / c);
  if (rpb > (uniform ? 64 : BWD_MAX_ROWS)) rpb = uniform ? 64 : BWD_MAX_ROWS;
  const int cu = (int)((B + rpb - 1) / rpb);
  const int64_t rem = B - (int64_t)(cu - 1) * rpb;
  *rows_per_block = rpb;
  *chunks = cu;
  if (uniform || (int64_t)*strips * cu <= target || rem * 4 <= rpb) return;
  // The kernels run four workgroups per CU (128 registers): 1024 slots.  A grid beyond them leaves its last workgroups to
  // start when the others end -- a second generation at a quarter of the occupancy (B = 4096, 20 strips: 1280 workgroups of
  // 64 cells, backward 349 us; 1020 of 80 or 81 cells: 295 us).  Harmless only where that tail is a short remainder chunk
  // (B = 512: 1040 workgroups, the last twenty of 2 cells, measured 0.5 us FASTER than 1020 balanced ones).  Otherwise: as
  // many chunks as fit in ONE generation (rounded down), the cells dealt out evenly -- the first B % chunks chunks take one
  // cell more (rows_per_block = 0 tells the kernels, which derive the deal from gridDim.y: chunk_cells).
  c = target / *strips;
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  if (c > B) c = (int)B;
  if ((B + c - 1) / c > BWD_MAX_ROWS) c = (int)((B + BWD_MAX_ROWS - 1) / BWD_MAX_ROWS);
  *chunks = c;
  *rows_per_block = 0;
}
int ll_target() {
  static const int target = getenv("VAE_BWD_WORKGROUPS") ? atoi(getenv("VAE_BWD_WORKGROUPS")) : 1024;  // tuning knob
  return target;
}
void bwd_grid(int64_t B, int64_t G, int* strips, int* chunks, int* rows_per_block) {
  // two genes per thread: 1536 slots.  Long chunks fill them (B = 4096, 40 strips: 38 chunks of 108 cells 288 us, 32 of
  // 128 cells 319); short ones (B = 512: 14-16 cells) pay more for a workgroup's start than the sixth wave brings, and 1280
  // workgroups are five per CU exactly (42.9 us; 1160 / 1400 / 1480 workgroups: 45.6 / 45.7 / 44.7).
  static const int target2 = getenv("VAE_BWD_WORKGROUPS2") ? atoi(getenv("VAE_BWD_WORKGROUPS2")) : 0;  // tuning knob (EPT 2)
  if (bwd_ept() == 2) ll_grid(B, G, 512, target2 > 0 ? target2 : (B >= 2048 ? 1536 : 1280), strips, chunks, rows_per_block);
  else ll_grid(B, G, BWD_STRIP, ll_target(), strips, chunks, rows_per_block);
}
}  // namespace

extern "C" int64_t vae_nbzinb_bwd_workspace_bytes(int64_t B, int64_t G) {
  int strips, chunks, rpb;
  bwd_grid(B, G, &strips, &chunks, &rpb);
  return (int64_t)chunks * VAE_SLAB_ROWS * G * (int64_t)sizeof(float);
}

int vae_nbzinb_bwd_chunks(int64_t B, int64_t G) {
  int strips, chunks, rpb;
  bwd_grid(B, G, &strips, &chunks, &rpb);
  return chunks;
}

int64_t vae_nbzinb_bwd_workgroups(int64_t B, int64_t G) {
  int strips, chunks, rpb;
  bwd_grid(B, G, &strips, &chunks, &rpb);
  return (int64_t)strips * chunks;
}

template <typename XT, typename LT = float>
static int nbzinb_bwd_impl(int zinb, const LT* logit_scale, int64_t ld_s, const LT* logit_dropout, int64_t ld_p,
                           const XT* x, int64_t ld_x, const float* library, const float* gene_table,
                           const float* rowmax, const float* rowsum, const float* S, const float* g_ll,
                           float g_scalar, int64_t B, int64_t G, LT* dlogit_scale, int64_t ld_ds,
                           LT* dlogit_dropout, int64_t ld_dp, float* dlibrary, float* dpx_r, float* dbias_scale,
                           float* dbias_dropout, float* workspace, int64_t workspace_bytes, void* stream, int overwrite,
                           float* absmax = nullptr, VaeDeferredColreduce* defer = nullptr, const VaeElboFromParts* elbo = nullptr) {
  VAE_REQUIRE(!overwrite || workspace, "vae_nbzinb_bwd_overwrite: needs the workspace (deterministic mode)");
  VAE_REQUIRE(logit_scale && x && library && gene_table && rowmax && rowsum && (S || elbo) && dlogit_scale && dpx_r,
              "vae_nbzinb_bwd: null pointer");
  VAE_REQUIRE(!zinb || (logit_dropout && dlogit_dropout), "vae_nbzinb_bwd: ZINB needs the dropout logits and their gradient");
  VAE_REQUIRE(B > 0 && G > 0 && G < (1 << 24), "vae_nbzinb_bwd: bad shape (the gene table is addressed with 24-bit multiplies: G < 2^24)");
  VAE_REQUIRE(ld_s >= G && ld_x >= G && ld_ds >= G && (!zinb || (ld_p >= G && ld_dp >= G)),
              "vae_nbzinb_bwd: leading dimension too small");
  constexpr bool U8 = sizeof(XT) == 1;
  constexpr bool L16 = sizeof(LT) == 2;
  constexpr int LD = L16 ? VAE_L_BF16 : VAE_L_F32;
  const bool x_ok = U8 ? ((reinterpret_cast<uintptr_t>(x) & 3u) == 0) : vae_aligned16(x);
  const bool vec = (G % 4 == 0) && (ld_s % 4 == 0) && (ld_x % 4 == 0) && (ld_ds % 4 == 0) &&
                   (!zinb || (ld_p % 4 == 0 && ld_dp % 4 == 0)) && logits_aligned(logit_scale, LD) && x_ok &&
                   logits_aligned(dlogit_scale, LD) && (!zinb || (logits_aligned(logit_dropout, LD) && logits_aligned(dlogit_dropout, LD)));
  VAE_REQUIRE(!L16 || vec, "vae_nbzinb_bwd: bf16 logits need G %% 4 == 0, leading dimensions %% 4 == 0 and aligned rows");
  int strips, chunks, rows_per_block;
  bwd_grid(B, G, &strips, &chunks, &rows_per_block);
  const bool slab = workspace != nullptr;
  VAE_REQUIRE(!slab || workspace_bytes >= vae_nbzinb_bwd_workspace_bytes(B, G),
              "vae_nbzinb_bwd: workspace too small (see vae_nbzinb_bwd_workspace_bytes)");
  ElboParts ep{};
  if (elbo) {
    VAE_REQUIRE(elbo->part && elbo->strips > 0 && elbo->ll && elbo->S && elbo->kl_z && elbo->kl_l && elbo->out5,
                "vae_nbzinb_bwd: incomplete deferred ELBO reduction");
    ep.part = elbo->part; ep.strips = elbo->strips; ep.ll = elbo->ll; ep.S = elbo->S; ep.recon = elbo->recon; ep.out5 = elbo->out5;
    ep.kl_z = elbo->kl_z; ep.kl_l = elbo->kl_l; ep.kl_weight = elbo->kl_weight; ep.coeff = elbo->coeff;
  }
  dim3 grid(strips, chunks);
  hipStream_t st = (hipStream_t)stream;
  const int ept = bwd_ept();
#define LAUNCH_BWD(Z, V, S_)                                                                                          \
  do {                                                                                                                \
    if (ept == 2)                                                                                                     \
      hipLaunchKernelGGL((nbzinb_bwd_kernel<Z, V, S_, 2, XT, LT>), grid, dim3(256), 0, st, logit_scale, ld_s, logit_dropout, ld_p, x, \
                         ld_x, library, gene_table, rowmax, rowsum, S, g_ll, g_scalar, (int)B, (int)G, rows_per_block,  \
                         dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r, dbias_scale, dbias_dropout, workspace, absmax, ep); \
    else                                                                                                              \
      hipLaunchKernelGGL((nbzinb_bwd_kernel<Z, V, S_, 4, XT, LT>), grid, dim3(256), 0, st, logit_scale, ld_s, logit_dropout, ld_p, x, \
                         ld_x, library, gene_table, rowmax, rowsum, S, g_ll, g_scalar, (int)B, (int)G, rows_per_block,  \
                         dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r, dbias_scale, dbias_dropout, workspace, absmax, ep); \
  } while (0)
#define LAUNCH_BWD2(Z, V) do { if (slab) LAUNCH_BWD(Z, V, true); else LAUNCH_BWD(Z, V, false); } while (0)
  VAE_REQUIRE(!U8 || vec, "vae_nbzinb_bwd_u8: needs G %% 4 == 0, leading dimensions %% 4 == 0 and aligned rows");
  if (U8 || L16) { if (zinb) LAUNCH_BWD2(true, true); else LAUNCH_BWD2(false, true); }
  else if (zinb) { if (vec) LAUNCH_BWD2(true, true); else LAUNCH_BWD2(true, false); }
  else           { if (vec) LAUNCH_BWD2(false, true); else LAUNCH_BWD2(false, false); }
#undef LAUNCH_BWD2
#undef LAUNCH_BWD
  VAE_CHECK_LAUNCH("vae_nbzinb_bwd");
  if (slab && defer && overwrite && dbias_scale && (!zinb || dbias_dropout)) {   // the caller folds the reduction into a later launch
    defer->slab = workspace; defer->chunks = chunks; defer->G = (int)G; defer->nk = zinb ? 3 : 2;
    defer->out[0] = dpx_r; defer->out[1] = dbias_scale; defer->out[2] = zinb ? dbias_dropout : nullptr;
  } else if (slab) {
    if (defer) defer->slab = nullptr;
    hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)((G + 255) / 256), zinb ? 3 : 2), dim3(256), 0, st, workspace,
                       chunks, (int)G, zinb ? 3 : 2, dpx_r, dbias_scale, dbias_dropout, overwrite);
    VAE_CHECK_LAUNCH("vae_nbzinb_bwd(colreduce)");
  }
  return VAE_OK;
}

int vae_nbzinb_bwd_ex(int x_dtype, int l_dtype, int zinb, const void* logit_scale_v, int64_t ld_s, const void* logit_dropout_v,
                      int64_t ld_p, const void* x, int64_t ld_x, const float* library, const float* gene_table, const float* rowmax,
                      const float* rowsum, const float* S, const float* g_ll, float g_scalar, int64_t B, int64_t G,
                      void* dlogit_scale_v, int64_t ld_ds, void* dlogit_dropout_v, int64_t ld_dp, float* dlibrary, float* dpx_r,
                      float* dbias_scale, float* dbias_dropout, float* workspace, int64_t workspace_bytes, float* absmax,
                      VaeDeferredColreduce* defer, void* stream, const VaeElboFromParts* elbo) {
  if (l_dtype == VAE_L_BF16) {
    const bf16_t *ls = (const bf16_t*)logit_scale_v, *lp = (const bf16_t*)logit_dropout_v;
    bf16_t *ds = (bf16_t*)dlogit_scale_v, *dp = (bf16_t*)dlogit_dropout_v;
    if (x_dtype == VAE_X_U8)
      return nbzinb_bwd_impl<uint8_t, bf16_t>(zinb, ls, ld_s, lp, ld_p, (const uint8_t*)x, ld_x, library, gene_table, rowmax, rowsum, S,
                                              g_ll, g_scalar, B, G, ds, ld_ds, dp, ld_dp, dlibrary, dpx_r, dbias_scale,
                                              dbias_dropout, workspace, workspace_bytes, stream, 1, absmax, defer, elbo);
    return nbzinb_bwd_impl<float, bf16_t>(zinb, ls, ld_s, lp, ld_p, (const float*)x, ld_x, library, gene_table, rowmax, rowsum, S, g_ll,
                                          g_scalar, B, G, ds, ld_ds, dp, ld_dp, dlibrary, dpx_r, dbias_scale, dbias_dropout,
                                          workspace, workspace_bytes, stream, 1, absmax, defer, elbo);
  }
  const float *logit_scale = (const float*)logit_scale_v, *logit_dropout = (const float*)logit_dropout_v;
  float *dlogit_scale = (float*)dlogit_scale_v, *dlogit_dropout = (float*)dlogit_dropout_v;
  if (x_dtype == VAE_X_U8)
    return nbzinb_bwd_impl<uint8_t>(zinb, logit_scale, ld_s, logit_dropout, ld_p, (const uint8_t*)x, ld_x, library, gene_table,
                                    rowmax, rowsum, S, g_ll, g_scalar, B, G, dlogit_scale, ld_ds, dlogit_dropout, ld_dp,
                                    dlibrary, dpx_r, dbias_scale, dbias_dropout, workspace, workspace_bytes, stream, 1, absmax, defer, elbo);
  return nbzinb_bwd_impl<float>(zinb, logit_scale, ld_s, logit_dropout, ld_p, (const float*)x, ld_x, library, gene_table, rowmax,
                                rowsum, S, g_ll, g_scalar, B, G, dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r,
                                dbias_scale, dbias_dropout, workspace, workspace_bytes, stream, 1, absmax, defer, elbo);
}

extern "C" int vae_nbzinb_bwd(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                              const float* x, int64_t ld_x, const float* library, const float* gene_table,
                              const float* rowmax, const float* rowsum, const float* S, const float* g_ll,
                              float g_scalar, int64_t B, int64_t G, float* dlogit_scale, int64_t ld_ds,
                              float* dlogit_dropout, int64_t ld_dp, float* dlibrary, float* dpx_r, float* dbias_scale,
                              float* dbias_dropout, float* workspace, int64_t workspace_bytes, void* stream) {
  return nbzinb_bwd_impl<float>(zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, rowmax, rowsum, S, g_ll,
                         g_scalar, B, G, dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r, dbias_scale,
                         dbias_dropout, workspace, workspace_bytes, stream, 0);
}

extern "C" int vae_nbzinb_bwd_overwrite(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout,
                                        int64_t ld_p, const float* x, int64_t ld_x, const float* library,
                                        const float* gene_table, const float* rowmax, const float* rowsum, const float* S,
                                        const float* g_ll, float g_scalar, int64_t B, int64_t G, float* dlogit_scale,
                                        int64_t ld_ds, float* dlogit_dropout, int64_t ld_dp, float* dlibrary, float* dpx_r,
                                        float* dbias_scale, float* dbias_dropout, float* workspace,
                                        int64_t workspace_bytes, void* stream) {
  return nbzinb_bwd_impl<float>(zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, rowmax, rowsum, S, g_ll,
                         g_scalar, B, G, dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r, dbias_scale,
                         dbias_dropout, workspace, workspace_bytes, stream, 1);
}

/* uint8 counts; overwrite != 0 as vae_nbzinb_bwd_overwrite */
extern "C" int vae_nbzinb_bwd_u8(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                                 const uint8_t* x, int64_t ld_x, const float* library, const float* gene_table,
                                 const float* rowmax, const float* rowsum, const float* S, const float* g_ll, float g_scalar,
                                 int64_t B, int64_t G, float* dlogit_scale, int64_t ld_ds, float* dlogit_dropout,
                                 int64_t ld_dp, float* dlibrary, float* dpx_r, float* dbias_scale, float* dbias_dropout,
                                 float* workspace, int64_t workspace_bytes, int overwrite, void* stream) {
  return nbzinb_bwd_impl<uint8_t>(zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, rowmax, rowsum, S,
                                  g_ll, g_scalar, B, G, dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r,
                                  dbias_scale, dbias_dropout, workspace, workspace_bytes, stream, overwrite);
}

// ---- forward, streaming form (host side) ----------------------------------------------------------------------------
namespace {
void fwd_grid(int64_t B, int64_t G, int* strips, int* chunks, int* rows_per_block) {
  ll_grid(B, G, FWD_STRIP, ll_target(), strips, chunks, rows_per_block);
  if (*rows_per_block == 0) return;   // balanced deal: at most BWD_MAX_ROWS = FWD_MAX_ROWS cells per chunk
  if (*rows_per_block > FWD_MAX_ROWS) *rows_per_block = FWD_MAX_ROWS;
  *chunks = (int)((B + *rows_per_block - 1) / *rows_per_block);
}
}  // namespace
int vae_nbzinb_fwd_strips(int64_t G) { return (int)((G + FWD_STRIP - 1) / FWD_STRIP); }
extern "C" int64_t vae_nbzinb_fwd_workspace_bytes(int64_t B, int64_t G) {
  int strips, chunks, rpb;
  fwd_grid(B, G, &strips, &chunks, &rpb);
  return (int64_t)strips * 2 * B * (int64_t)sizeof(float);
}

// have_stats != 0: rowmax / rowsum already hold the softmax statistics of every row (decoder-head GEMM epilogue).
// ll / S may be null: the per-strip partials then stay in `workspace` ([strips][2][B]) for vae_elbo_reduce_parts.
int vae_nbzinb_fwd_tiled_ex(int x_dtype, int l_dtype, int zinb, const void* logit_scale, int64_t ld_s, const void* logit_dropout,
                            int64_t ld_p, const void* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                            int64_t G, float* ll, float* rowmax, float* rowsum, float* S, int have_stats, float* workspace,
                            int64_t workspace_bytes, int* strips_out, void* stream) {
  VAE_REQUIRE(logit_scale && x && library && gene_table && rowmax && rowsum && workspace, "vae_nbzinb_fwd_ws: null pointer");
  VAE_REQUIRE((ll == nullptr) == (S == nullptr), "vae_nbzinb_fwd_ws: ll and S go together");
  VAE_REQUIRE(!zinb || logit_dropout, "vae_nbzinb_fwd_ws: ZINB needs logit_dropout");
  VAE_REQUIRE(B > 0 && G > 0 && B < (1 << 30) && G < (1 << 24), "vae_nbzinb_fwd_ws: bad shape (G < 2^24: 24-bit table addressing)");
  VAE_REQUIRE(ld_s >= G && ld_x >= G && (!zinb || ld_p >= G), "vae_nbzinb_fwd_ws: leading dimension too small");
  VAE_REQUIRE(x_dtype == VAE_X_F32 || x_dtype == VAE_X_U8, "vae_nbzinb_fwd_ws: bad x_dtype");
  VAE_REQUIRE(workspace_bytes >= vae_nbzinb_fwd_workspace_bytes(B, G), "vae_nbzinb_fwd_ws: workspace too small");
  const bool u8 = x_dtype == VAE_X_U8;
  const bool x_ok = u8 ? ((reinterpret_cast<uintptr_t>(x) & 3u) == 0) : vae_aligned16(x);
  const bool l16 = l_dtype == VAE_L_BF16;
  VAE_REQUIRE(l16 || l_dtype == VAE_L_F32, "vae_nbzinb_fwd_ws: bad logit dtype");
  const bool vec = (G % 4 == 0) && (ld_s % 4 == 0) && (ld_x % 4 == 0) && (!zinb || ld_p % 4 == 0) &&
                   logits_aligned(logit_scale, l_dtype) && x_ok && (!zinb || logits_aligned(logit_dropout, l_dtype)) &&
                   vae_aligned16(gene_table);
  VAE_REQUIRE(!(u8 || l16) || vec, "vae_nbzinb_fwd_u8 / bf16 logits: need G %% 4 == 0, leading dimensions %% 4 == 0 and aligned rows");
  hipStream_t st = (hipStream_t)stream;
  if (!have_stats) {
    if (l16)      hipLaunchKernelGGL((softmax_stats_kernel<true, bf16_t>), dim3((unsigned)B), dim3(256), 0, st, (const bf16_t*)logit_scale, ld_s, (int)G, rowmax, rowsum);
    else if (vec) hipLaunchKernelGGL((softmax_stats_kernel<true, float>), dim3((unsigned)B), dim3(256), 0, st, (const float*)logit_scale, ld_s, (int)G, rowmax, rowsum);
    else          hipLaunchKernelGGL((softmax_stats_kernel<false, float>), dim3((unsigned)B), dim3(256), 0, st, (const float*)logit_scale, ld_s, (int)G, rowmax, rowsum);
    VAE_CHECK_LAUNCH("vae_nbzinb_fwd_ws(stats)");
  }
  int strips, chunks, rows_per_block;
  fwd_grid(B, G, &strips, &chunks, &rows_per_block);
  VAE_REQUIRE(chunks <= 65535, "vae_nbzinb_fwd_ws: batch too large for one launch (B <= 16 * 65535)");
  const dim3 grid(strips, chunks);
#define LAUNCH_FT(Z, V, XT_, LT_)                                                                                         \
  hipLaunchKernelGGL((nbzinb_fwd_tiled_kernel<Z, V, 4, XT_, LT_>), grid, dim3(256), 0, st, (const LT_*)logit_scale, ld_s,    \
                     (const LT_*)logit_dropout, ld_p, (const XT_*)x, ld_x, library, gene_table, rowmax, rowsum, (int)B, (int)G, \
                     rows_per_block, workspace)
  if (l16) {
    if (u8) { if (zinb) LAUNCH_FT(true, true, uint8_t, bf16_t); else LAUNCH_FT(false, true, uint8_t, bf16_t); }
    else    { if (zinb) LAUNCH_FT(true, true, float, bf16_t); else LAUNCH_FT(false, true, float, bf16_t); }
  }
  else if (u8) { if (zinb) LAUNCH_FT(true, true, uint8_t, float); else LAUNCH_FT(false, true, uint8_t, float); }
  else if (zinb) { if (vec) LAUNCH_FT(true, true, float, float); else LAUNCH_FT(true, false, float, float); }
  else           { if (vec) LAUNCH_FT(false, true, float, float); else LAUNCH_FT(false, false, float, float); }
#undef LAUNCH_FT
  VAE_CHECK_LAUNCH("vae_nbzinb_fwd_ws");
  if (strips_out) *strips_out = strips;
  if (ll) {
    hipLaunchKernelGGL(fwd_partials_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, workspace, strips, (int)B, ll, S);
    VAE_CHECK_LAUNCH("vae_nbzinb_fwd_ws(finish)");
  }
  return VAE_OK;
}

// ll[b] / S[b] from the per-strip partials a vae_nbzinb_fwd_tiled_ex(ll = S = null) call left in its workspace
int vae_nbzinb_fwd_finish(const float* part, int strips, int64_t B, float* ll, float* S, void* stream) {
  VAE_REQUIRE(part && strips > 0 && B > 0 && ll && S, "vae_nbzinb_fwd_finish: bad arguments");
  hipLaunchKernelGGL(fwd_partials_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, strips,
                     (int)B, ll, S);
  VAE_CHECK_LAUNCH("vae_nbzinb_fwd_finish");
  return VAE_OK;
}

extern "C" int vae_nbzinb_fwd_ws(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                                 const void* x, int x_dtype, int64_t ld_x, const float* library, const float* gene_table,
                                 int64_t B, int64_t G, float* ll, float* rowmax, float* rowsum, float* S, int have_stats,
                                 float* workspace, int64_t workspace_bytes, void* stream) {
  return vae_nbzinb_fwd_tiled_ex(x_dtype, VAE_L_F32, zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, B, G,
                                 ll, rowmax, rowsum, S, have_stats, workspace, workspace_bytes, nullptr, stream);
}

// bf16 logits (and, in the backward, bf16 logit gradients, in place or not): what a bf16-mode step runs
extern "C" int vae_nbzinb_fwd_ws_bf16(int zinb, const uint16_t* logit_scale, int64_t ld_s, const uint16_t* logit_dropout, int64_t ld_p,
                                      const void* x, int x_dtype, int64_t ld_x, const float* library, const float* gene_table,
                                      int64_t B, int64_t G, float* ll, float* rowmax, float* rowsum, float* S, int have_stats,
                                      float* workspace, int64_t workspace_bytes, void* stream) {
  return vae_nbzinb_fwd_tiled_ex(x_dtype, VAE_L_BF16, zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, B, G,
                                 ll, rowmax, rowsum, S, have_stats, workspace, workspace_bytes, nullptr, stream);
}
extern "C" int vae_nbzinb_bwd_bf16(int zinb, const uint16_t* logit_scale, int64_t ld_s, const uint16_t* logit_dropout, int64_t ld_p,
                                   const void* x, int x_dtype, int64_t ld_x, const float* library, const float* gene_table,
                                   const float* rowmax, const float* rowsum, const float* S, const float* g_ll, float g_scalar,
                                   int64_t B, int64_t G, uint16_t* dlogit_scale, int64_t ld_ds, uint16_t* dlogit_dropout,
                                   int64_t ld_dp, float* dlibrary, float* dpx_r, float* dbias_scale, float* dbias_dropout,
                                   float* workspace, int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(x_dtype == VAE_X_F32 || x_dtype == VAE_X_U8, "vae_nbzinb_bwd_bf16: bad x_dtype");
  return vae_nbzinb_bwd_ex(x_dtype, VAE_L_BF16, zinb, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, library, gene_table, rowmax,
                           rowsum, S, g_ll, g_scalar, B, G, dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r, dbias_scale,
                           dbias_dropout, workspace, workspace_bytes, nullptr, nullptr, stream);
}

int vae_softmax_stats_merge(const float* stat_m, const float* stat_s, int slots, int64_t B, float* rowmax, float* rowsum,
                            const float* px_r, int64_t G, float* table, void* stream) {
  VAE_REQUIRE(stat_m && stat_s && slots > 0 && B > 0 && rowmax && rowsum, "vae_softmax_stats_merge: bad arguments");
  VAE_REQUIRE(!px_r || (table && G > 0), "vae_softmax_stats_merge: the gene table needs px_r, G and the table");
  const unsigned extra = px_r ? gene_table_blocks(G) : 0u;
  hipLaunchKernelGGL(softmax_stats_merge_kernel, dim3((unsigned)((B + 3) / 4) + extra), dim3(256), 0, (hipStream_t)stream, stat_m,
                     stat_s, slots, (int)B, rowmax, rowsum, px_r, (int)G, table);
  VAE_CHECK_LAUNCH("vae_softmax_stats_merge");
  return VAE_OK;
}
