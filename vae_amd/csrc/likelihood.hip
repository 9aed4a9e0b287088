// Fused decoder epilogue of the scVI-style VAE (gfx950): softmax over genes, library-size
// scaling and the NB / ZINB log-likelihood with its row reduction, forward and backward.
//
// Reference semantics (all fp32):
//   px_scale = softmax_g(logit_scale)            modules.py:362,401
//   px_rate  = exp(library) * px_scale           modules.py:404
//   theta    = exp(px_r)                         task.py:481-483
//   ll       = log_zinb_positive / log_nb_positive(x, px_rate, theta, logit_dropout)
//                                                distributions.py:22-62, :65-90
//   recon    = -ll.sum(-1)                       task.py:156-165
//
// Forward is row-organised: one workgroup owns one cell, stages its scale-logit row in LDS,
// reduces max / sum-exp with wavefront shuffles + LDS, then makes ONE pass over
// (x, logit_dropout) -- algorithmic HBM traffic 12 B per cell*gene (ZINB), 8 B (NB).  It
// also reduces S_b = sum_g mu * dll/dmu, which is all the backward needs to be a single
// element-wise pass: dlogit_scale = delta_b * (a_g - p_g * S_b).  Backward is therefore
// tiled over (gene strip x cell chunk), keeps per-gene column sums (d px_r, both bias
// gradients) in registers across the cells of its chunk and flushes them with one float
// atomic per gene per workgroup: 12 B read + 8 B written per cell*gene.
#include "common.h"

namespace {

constexpr float EPS = 1e-8f;

// ---- special functions ---------------------------------------------------------------
// digamma for z > 0: recurrence up to z >= 6, then the asymptotic series.
__device__ __forceinline__ float digamma_pos(float z) {
  float r = 0.f;
  while (z < 6.f) {
    r -= 1.f / z;
    z += 1.f;
  }
  const float iz = 1.f / z, iz2 = iz * iz;
  return r + logf(z) - 0.5f * iz - iz2 * (0.083333333333f - iz2 * (0.0083333333333f - iz2 * 0.003968253968f));
}
// Stirling series, z >= 8 (truncation error < 3e-8)
__device__ __forceinline__ float lgamma_stirling(float z) {
  const float iz = 1.f / z, iz2 = iz * iz;
  return (z - 0.5f) * logf(z) - z + 0.91893853320467274178f +
         iz * (0.083333333333f - iz2 * (0.0027777777778f - iz2 * 0.00079365079365f));
}
__device__ __forceinline__ bool small_int(float x, float lim) { return x <= lim && x == floorf(x); }

// lgamma(x + theta) - lgamma(theta), x >= 0.  For small integer counts this is the log of a
// rising factorial: exact identity, no cancellation, one log.
__device__ __forceinline__ float lgamma_diff(float x, float th, float lg_th) {
  if (small_int(x, 8.f) && th <= 1e4f) {
    float p = 1.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) p *= ((float)k < x) ? th + (float)k : 1.f;
    return logf(p);
  }
  const float z = x + th;
  return (z >= 8.f ? lgamma_stirling(z) : lgammaf(z)) - lg_th;
}
// lgamma(x + 1)
__device__ __forceinline__ float lfact(float x) {
  if (small_int(x, 8.f)) {
    float p = 1.f;
#pragma unroll
    for (int k = 2; k <= 8; ++k) p *= ((float)k <= x) ? (float)k : 1.f;
    return p == 1.f ? 0.f : logf(p);
  }
  const float z = x + 1.f;
  return z >= 8.f ? lgamma_stirling(z) : lgammaf(z);
}
// digamma(x + theta) - digamma(theta): sum_{k<x} 1/(theta+k) for small integer counts
__device__ __forceinline__ float digamma_diff(float x, float th, float psi_th) {
  if (small_int(x, 16.f)) {
    float s = 0.f;
    const int n = (int)x;
    for (int k = 0; k < n; ++k) s += 1.f / (th + (float)k);
    return s;
  }
  return digamma_pos(x + th) - psi_th;
}

// ---- one (cell, gene) element ------------------------------------------------------------
struct Elem {
  float ll;    // log-likelihood
  float a;     // mu * d ll / d mu   (gradient w.r.t. log mu)
  float dpi;   // d ll / d logit_dropout
  float dth;   // d ll / d theta
};

template <bool ZINB, bool GRAD>
__device__ __forceinline__ Elem elem(float x, float mu, float pi, float th, float lte, float lg_th, float psi_th) {
  Elem e;
  const float tm = th + mu + EPS;
  const float ltm = logf(tm);
  const float dlt = lte - ltm;
  const float itm = 1.f / tm;
  const float dptl_dth = dlt + th * (1.f / (th + EPS) - itm);
  e.dpi = 0.f;
  if (ZINB) {
    const float sp = softplus20(-pi);
    const float ptl = -pi + th * dlt;
    const float sg_pi = softplus20_grad(-pi);
    if (x < EPS) {
      e.ll = softplus20(ptl) - sp;
      const float sg = softplus20_grad(ptl);
      e.a = -sg * th * itm * mu;
      e.dpi = sg_pi - sg;
      e.dth = sg * dptl_dth;
    } else if (x > EPS) {
      e.ll = -sp + ptl + x * (logf(mu + EPS) - ltm) + lgamma_diff(x, th, lg_th) - lfact(x);
      e.a = (x / (mu + EPS) - (x + th) * itm) * mu;
      e.dpi = sg_pi - 1.f;
      e.dth = GRAD ? dptl_dth - x * itm + digamma_diff(x, th, psi_th) : 0.f;
    } else {
      e.ll = e.a = e.dth = 0.f;
    }
  } else {
    e.ll = th * dlt + x * (logf(mu + EPS) - ltm) + lgamma_diff(x, th, lg_th) - lfact(x);
    e.a = (x / (mu + EPS) - (x + th) * itm) * mu;
    e.dth = GRAD ? dptl_dth - x * itm + digamma_diff(x, th, psi_th) : 0.f;
  }
  return e;
}

// ---- per-gene table: theta, log(theta+eps), lgamma(theta), digamma(theta) -----------------
__global__ void gene_table_kernel(const float* __restrict__ px_r, int G, float* __restrict__ T) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const float th = expf(px_r[g]);
  T[g] = th;
  T[G + g] = logf(th + EPS);
  T[2 * G + g] = lgammaf(th);
  T[3 * G + g] = digamma_pos(th);
}

// ---- forward: one workgroup per cell -------------------------------------------------------
constexpr int FWD_THREADS = 512;

template <bool ZINB>
__global__ __launch_bounds__(FWD_THREADS) void nbzinb_fwd_kernel(
    const float* __restrict__ ls, int64_t ld_s, const float* __restrict__ lp, int64_t ld_p, const float* __restrict__ x,
    int64_t ld_x, const float* __restrict__ library, const float* __restrict__ T, int G, int vec,
    float* __restrict__ ll_out, float* __restrict__ rowmax, float* __restrict__ rowsum, float* __restrict__ S_out) {
  extern __shared__ __attribute__((aligned(16))) float srow[];  // G floats (+ pad) then 8 floats scratch
  float* red = srow + ((G + 3) & ~3);
  constexpr int NW = FWD_THREADS / 64;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* r_s = ls + (int64_t)b * ld_s;
  const float* r_p = ZINB ? lp + (int64_t)b * ld_p : nullptr;
  const float* r_x = x + (int64_t)b * ld_x;
  const int n4 = vec ? (G >> 2) : 0;

  // sweep 1: HBM -> LDS, running max
  float m = -INFINITY;
  for (int i = tid; i < n4; i += FWD_THREADS) {
    const float4 v = reinterpret_cast<const float4*>(r_s)[i];
    reinterpret_cast<float4*>(srow)[i] = v;
    m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  for (int i = (n4 << 2) + tid; i < G; i += FWD_THREADS) {
    const float v = r_s[i];
    srow[i] = v;
    m = fmaxf(m, v);
  }
  m = block_max<NW>(m, red);
  // sweep 2 (LDS only): e = exp(l - max), kept in LDS for sweep 3; sum
  float se = 0.f;
  for (int i = tid; i < n4; i += FWD_THREADS) {
    float4 v = reinterpret_cast<float4*>(srow)[i];
    v.x = expf(v.x - m); v.y = expf(v.y - m); v.z = expf(v.z - m); v.w = expf(v.w - m);
    reinterpret_cast<float4*>(srow)[i] = v;
    se += (v.x + v.y) + (v.z + v.w);
  }
  for (int i = (n4 << 2) + tid; i < G; i += FWD_THREADS) {
    const float v = expf(srow[i] - m);
    srow[i] = v;
    se += v;
  }
  se = block_sum<NW>(se, red);
  const float scale = expf(library[b]) / se;  // mu = exp(library) * e / sum

  // sweep 3: the likelihood pass over (x, logit_dropout)
  float acc_ll = 0.f, acc_a = 0.f;
  for (int i = tid; i < n4; i += FWD_THREADS) {
    const float4 e4 = reinterpret_cast<float4*>(srow)[i];
    const float4 x4 = reinterpret_cast<const float4*>(r_x)[i];
    float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ZINB) p4 = reinterpret_cast<const float4*>(r_p)[i];
    const float4 th4 = reinterpret_cast<const float4*>(T)[i];
    const float4 lt4 = reinterpret_cast<const float4*>(T + G)[i];
    const float4 lg4 = reinterpret_cast<const float4*>(T + 2 * G)[i];
    Elem e;
    e = elem<ZINB, false>(x4.x, scale * e4.x, p4.x, th4.x, lt4.x, lg4.x, 0.f); acc_ll += e.ll; acc_a += e.a;
    e = elem<ZINB, false>(x4.y, scale * e4.y, p4.y, th4.y, lt4.y, lg4.y, 0.f); acc_ll += e.ll; acc_a += e.a;
    e = elem<ZINB, false>(x4.z, scale * e4.z, p4.z, th4.z, lt4.z, lg4.z, 0.f); acc_ll += e.ll; acc_a += e.a;
    e = elem<ZINB, false>(x4.w, scale * e4.w, p4.w, th4.w, lt4.w, lg4.w, 0.f); acc_ll += e.ll; acc_a += e.a;
  }
  for (int i = (n4 << 2) + tid; i < G; i += FWD_THREADS) {
    const Elem e = elem<ZINB, false>(r_x[i], scale * srow[i], ZINB ? r_p[i] : 0.f, T[i], T[G + i], T[2 * G + i], 0.f);
    acc_ll += e.ll;
    acc_a += e.a;
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

// ---- backward: (gene strip of 1024) x (chunk of cells) per workgroup ----------------------
constexpr int BWD_STRIP = 1024;  // 256 threads x float4

template <bool ZINB, bool VEC>
__global__ __launch_bounds__(256) void nbzinb_bwd_kernel(
    const float* ls, int64_t ld_s, const float* lp, int64_t ld_p, const float* __restrict__ x, int64_t ld_x,
    const float* __restrict__ library, const float* __restrict__ T, const float* __restrict__ rowmax,
    const float* __restrict__ rowsum, const float* __restrict__ S, const float* __restrict__ g_ll, float g_scalar, int B,
    int G, int rows_per_block, float* dls, int64_t ld_ds, float* dlp, int64_t ld_dp, float* __restrict__ dlibrary,
    float* __restrict__ dpx_r, float* __restrict__ dbias_s, float* __restrict__ dbias_p) {
  const int g0 = blockIdx.x * BWD_STRIP + threadIdx.x * 4;
  const int b0 = blockIdx.y * rows_per_block;
  const int b1 = min(B, b0 + rows_per_block);
  if (blockIdx.x == 0 && dlibrary)
    for (int b = b0 + threadIdx.x; b < b1; b += 256) dlibrary[b] = g_scalar * (g_ll ? g_ll[b] : 1.f) * S[b];
  if (g0 >= G) return;
  const int nv = min(4, G - g0);  // valid columns of this thread
  float th[4], lt[4], lg[4], ps[4], acc_t[4], acc_s[4], acc_p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = min(g0 + j, G - 1);
    th[j] = T[g]; lt[j] = T[G + g]; lg[j] = T[2 * G + g]; ps[j] = T[3 * G + g];
    acc_t[j] = acc_s[j] = acc_p[j] = 0.f;
  }
  for (int b = b0; b < b1; ++b) {
    const float delta = g_scalar * (g_ll ? g_ll[b] : 1.f);
    const float m = rowmax[b], inv = 1.f / rowsum[b], el = expf(library[b]), Sb = S[b];
    float l4[4], p4[4] = {0.f, 0.f, 0.f, 0.f}, x4[4];
    const float* r_s = ls + (int64_t)b * ld_s + g0;
    const float* r_p = ZINB ? lp + (int64_t)b * ld_p + g0 : nullptr;
    const float* r_x = x + (int64_t)b * ld_x + g0;
    if (VEC) {
      *reinterpret_cast<float4*>(l4) = *reinterpret_cast<const float4*>(r_s);
      *reinterpret_cast<float4*>(x4) = *reinterpret_cast<const float4*>(r_x);
      if (ZINB) *reinterpret_cast<float4*>(p4) = *reinterpret_cast<const float4*>(r_p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = j < nv;
        l4[j] = ok ? r_s[j] : 0.f;
        x4[j] = ok ? r_x[j] : 0.f;
        if (ZINB) p4[j] = ok ? r_p[j] : 0.f;
      }
    }
    float o_s[4], o_p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = expf(l4[j] - m) * inv;
      const Elem e = elem<ZINB, true>(x4[j], el * p, p4[j], th[j], lt[j], lg[j], ps[j]);
      o_s[j] = delta * (e.a - p * Sb);
      o_p[j] = delta * e.dpi;
      acc_t[j] += delta * e.dth * th[j];
      acc_s[j] += o_s[j];
      acc_p[j] += o_p[j];
    }
    float* w_s = dls + (int64_t)b * ld_ds + g0;
    float* w_p = ZINB ? dlp + (int64_t)b * ld_dp + g0 : nullptr;
    if (VEC) {
      *reinterpret_cast<float4*>(w_s) = *reinterpret_cast<float4*>(o_s);
      if (ZINB) *reinterpret_cast<float4*>(w_p) = *reinterpret_cast<float4*>(o_p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nv) {
          w_s[j] = o_s[j];
          if (ZINB) w_p[j] = o_p[j];
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < nv) {
      atomicAdd(dpx_r + g0 + j, acc_t[j]);
      if (dbias_s) atomicAdd(dbias_s + g0 + j, acc_s[j]);
      if (ZINB && dbias_p) atomicAdd(dbias_p + g0 + j, acc_p[j]);
    }
}

// ---- element-wise API kernels (log_zinb_positive / log_nb_positive as standalone ops) -----
// theta is addressed as theta[b * ld_th + g]; ld_th = 0 broadcasts a (G,) vector.
template <bool ZINB>
__global__ void elem_fwd_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ mu, int64_t ld_mu,
                                const float* __restrict__ theta, int64_t ld_th, const float* __restrict__ pi,
                                int64_t ld_pi, int B, int G, float* __restrict__ out, int64_t ld_o) {
  const int64_t total = (int64_t)B * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / G, g = i % G;
    const float th = theta[b * ld_th + g];
    const Elem e = elem<ZINB, false>(x[b * ld_x + g], mu[b * ld_mu + g], ZINB ? pi[b * ld_pi + g] : 0.f, th,
                                     logf(th + EPS), lgammaf(th), 0.f);
    out[b * ld_o + g] = e.ll;
  }
}
template <bool ZINB>
__global__ void elem_bwd_kernel(const float* __restrict__ gout, int64_t ld_g, const float* __restrict__ x, int64_t ld_x,
                                const float* __restrict__ mu, int64_t ld_mu, const float* __restrict__ theta,
                                int64_t ld_th, const float* __restrict__ pi, int64_t ld_pi, int B, int G,
                                float* __restrict__ dmu, float* __restrict__ dtheta, float* __restrict__ dpi) {
  const int64_t total = (int64_t)B * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / G, g = i % G;
    const float th = theta[b * ld_th + g], m = mu[b * ld_mu + g], go = gout[b * ld_g + g];
    const Elem e = elem<ZINB, true>(x[b * ld_x + g], m, ZINB ? pi[b * ld_pi + g] : 0.f, th, logf(th + EPS), lgammaf(th),
                                    digamma_pos(th));
    // e.a = mu * dll/dmu; undo the factor (mu = 0 has dll/dmu finite only through eps: recompute)
    const float xx = x[b * ld_x + g];
    float dm;
    const float tm = th + m + EPS;
    if (ZINB && xx < EPS) {
      const float ptl = -pi[b * ld_pi + g] + th * (logf(th + EPS) - logf(tm));
      dm = -softplus20_grad(ptl) * th / tm;
    } else if (ZINB && !(xx > EPS)) {
      dm = 0.f;
    } else {
      dm = xx / (m + EPS) - (xx + th) / tm;
    }
    dmu[i] = go * dm;
    dtheta[i] = go * e.dth;
    if (ZINB) dpi[i] = go * e.dpi;
  }
}

// ---- standalone softmax + library scaling (DecoderSCVI.forward outputs, modules.py:401-404) ----
// px_scale = softmax(logits), px_rate = exp(library) * px_scale; one workgroup per cell, row in LDS.
__global__ __launch_bounds__(FWD_THREADS) void softmax_rate_fwd_kernel(const float* __restrict__ ls, int64_t ld_s,
                                                                       const float* __restrict__ library, int G,
                                                                       float* __restrict__ px_scale,
                                                                       float* __restrict__ px_rate) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  float* red = srow + ((G + 3) & ~3);
  constexpr int NW = FWD_THREADS / 64;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* r = ls + (int64_t)b * ld_s;
  float m = -INFINITY;
  for (int i = tid; i < G; i += FWD_THREADS) {
    const float v = r[i];
    srow[i] = v;
    m = fmaxf(m, v);
  }
  m = block_max<NW>(m, red);
  float se = 0.f;
  for (int i = tid; i < G; i += FWD_THREADS) {
    const float v = expf(srow[i] - m);
    srow[i] = v;
    se += v;
  }
  se = block_sum<NW>(se, red);
  const float el = expf(library[b]);
  for (int i = tid; i < G; i += FWD_THREADS) {
    const float p = srow[i] / se;
    px_scale[(int64_t)b * G + i] = p;
    px_rate[(int64_t)b * G + i] = el * p;
  }
}
// dlogits = p * (t - sum_g p*t), t = d_scale + exp(library) * d_rate ; dlibrary = sum_g d_rate * rate
__global__ __launch_bounds__(FWD_THREADS) void softmax_rate_bwd_kernel(const float* __restrict__ px_scale,
                                                                       const float* __restrict__ library,
                                                                       const float* __restrict__ d_scale,
                                                                       const float* __restrict__ d_rate, int G,
                                                                       float* __restrict__ dlogits, int64_t ld_d,
                                                                       float* __restrict__ dlibrary) {
  __shared__ float red[16];
  constexpr int NW = FWD_THREADS / 64;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float el = expf(library[b]);
  const float* p = px_scale + (int64_t)b * G;
  float dot = 0.f, dl = 0.f;
  for (int i = tid; i < G; i += FWD_THREADS) {
    const float ds = d_scale ? d_scale[(int64_t)b * G + i] : 0.f;
    const float dr = d_rate ? d_rate[(int64_t)b * G + i] : 0.f;
    dot += p[i] * (ds + el * dr);
    dl += dr * el * p[i];
  }
  dot = block_sum<NW>(dot, red);
  dl = block_sum<NW>(dl, red);
  for (int i = tid; i < G; i += FWD_THREADS) {
    const float ds = d_scale ? d_scale[(int64_t)b * G + i] : 0.f;
    const float dr = d_rate ? d_rate[(int64_t)b * G + i] : 0.f;
    dlogits[(int64_t)b * ld_d + i] = p[i] * (ds + el * dr - dot);
  }
  if (tid == 0) dlibrary[b] = dl;
}

}  // namespace

extern "C" int vae_gene_table(const float* px_r, int64_t G, float* table4G, void* stream) {
  VAE_REQUIRE(px_r && table4G && G > 0, "vae_gene_table: bad arguments");
  hipLaunchKernelGGL(gene_table_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, (hipStream_t)stream, px_r,
                     (int)G, table4G);
  VAE_CHECK_LAUNCH("vae_gene_table");
  return VAE_OK;
}

extern "C" int64_t vae_nbzinb_max_genes(void) { return (160 * 1024 - 64) / 4 - 4; }

extern "C" int vae_nbzinb_fwd(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                              const float* x, int64_t ld_x, const float* library, const float* gene_table, int64_t B,
                              int64_t G, float* ll, float* rowmax, float* rowsum, float* S, void* stream) {
  VAE_REQUIRE(logit_scale && x && library && gene_table && ll && rowmax && rowsum && S, "vae_nbzinb_fwd: null pointer");
  VAE_REQUIRE(!zinb || logit_dropout, "vae_nbzinb_fwd: ZINB needs logit_dropout");
  VAE_REQUIRE(B > 0 && G > 0 && G <= vae_nbzinb_max_genes(), "vae_nbzinb_fwd: G=%lld outside [1, %lld]", (long long)G,
              (long long)vae_nbzinb_max_genes());
  VAE_REQUIRE(ld_s >= G && ld_x >= G && (!zinb || ld_p >= G), "vae_nbzinb_fwd: leading dimension too small");
  const int vec = (G % 4 == 0) && (ld_s % 4 == 0) && (ld_x % 4 == 0) && (!zinb || ld_p % 4 == 0) &&
                  vae_aligned16(logit_scale) && vae_aligned16(x) && (!zinb || vae_aligned16(logit_dropout)) &&
                  vae_aligned16(gene_table);
  const size_t shmem = (size_t)(((G + 3) & ~3) + 16) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (shmem > 64 * 1024) {  // opt in to the large-LDS carve-out (per device, cheap, idempotent)
    const hipError_t e = zinb ? hipFuncSetAttribute((const void*)nbzinb_fwd_kernel<true>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                              : hipFuncSetAttribute((const void*)nbzinb_fwd_kernel<false>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return vae_set_error(VAE_ERR_LAUNCH, "vae_nbzinb_fwd: cannot raise LDS limit: %s", hipGetErrorString(e));
  }
  if (zinb)
    hipLaunchKernelGGL(nbzinb_fwd_kernel<true>, dim3((unsigned)B), dim3(FWD_THREADS), shmem, st, logit_scale, ld_s,
                       logit_dropout, ld_p, x, ld_x, library, gene_table, (int)G, vec, ll, rowmax, rowsum, S);
  else
    hipLaunchKernelGGL(nbzinb_fwd_kernel<false>, dim3((unsigned)B), dim3(FWD_THREADS), shmem, st, logit_scale, ld_s,
                       nullptr, 0, x, ld_x, library, gene_table, (int)G, vec, ll, rowmax, rowsum, S);
  VAE_CHECK_LAUNCH("vae_nbzinb_fwd");
  return VAE_OK;
}

extern "C" int vae_nbzinb_bwd(int zinb, const float* logit_scale, int64_t ld_s, const float* logit_dropout, int64_t ld_p,
                              const float* x, int64_t ld_x, const float* library, const float* gene_table,
                              const float* rowmax, const float* rowsum, const float* S, const float* g_ll,
                              float g_scalar, int64_t B, int64_t G, float* dlogit_scale, int64_t ld_ds,
                              float* dlogit_dropout, int64_t ld_dp, float* dlibrary, float* dpx_r, float* dbias_scale,
                              float* dbias_dropout, void* stream) {
  VAE_REQUIRE(logit_scale && x && library && gene_table && rowmax && rowsum && S && dlogit_scale && dpx_r,
              "vae_nbzinb_bwd: null pointer");
  VAE_REQUIRE(!zinb || (logit_dropout && dlogit_dropout), "vae_nbzinb_bwd: ZINB needs the dropout logits and their gradient");
  VAE_REQUIRE(B > 0 && G > 0, "vae_nbzinb_bwd: bad shape");
  VAE_REQUIRE(ld_s >= G && ld_x >= G && ld_ds >= G && (!zinb || (ld_p >= G && ld_dp >= G)),
              "vae_nbzinb_bwd: leading dimension too small");
  const bool vec = (G % 4 == 0) && (ld_s % 4 == 0) && (ld_x % 4 == 0) && (ld_ds % 4 == 0) &&
                   (!zinb || (ld_p % 4 == 0 && ld_dp % 4 == 0)) && vae_aligned16(logit_scale) && vae_aligned16(x) &&
                   vae_aligned16(dlogit_scale) && (!zinb || (vae_aligned16(logit_dropout) && vae_aligned16(dlogit_dropout)));
  const int strips = (int)((G + BWD_STRIP - 1) / BWD_STRIP);
  // enough workgroups to fill the chip (>= ~4 per CU), as few atomics as that allows
  int chunks = (1024 + strips - 1) / strips;
  if (chunks > B) chunks = (int)B;
  const int rows_per_block = (int)((B + chunks - 1) / chunks);
  chunks = (int)((B + rows_per_block - 1) / rows_per_block);
  dim3 grid(strips, chunks);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BWD(Z, V)                                                                                              \
  hipLaunchKernelGGL((nbzinb_bwd_kernel<Z, V>), grid, dim3(256), 0, st, logit_scale, ld_s, logit_dropout, ld_p, x, ld_x, \
                     library, gene_table, rowmax, rowsum, S, g_ll, g_scalar, (int)B, (int)G, rows_per_block,          \
                     dlogit_scale, ld_ds, dlogit_dropout, ld_dp, dlibrary, dpx_r, dbias_scale, dbias_dropout)
  if (zinb) { if (vec) LAUNCH_BWD(true, true); else LAUNCH_BWD(true, false); }
  else      { if (vec) LAUNCH_BWD(false, true); else LAUNCH_BWD(false, false); }
#undef LAUNCH_BWD
  VAE_CHECK_LAUNCH("vae_nbzinb_bwd");
  return VAE_OK;
}

extern "C" int vae_loglik_elem_fwd(int zinb, const float* x, int64_t ld_x, const float* mu, int64_t ld_mu,
                                   const float* theta, int64_t ld_theta, const float* pi, int64_t ld_pi, int64_t B,
                                   int64_t G, float* out, int64_t ld_out, void* stream) {
  VAE_REQUIRE(x && mu && theta && out && (!zinb || pi) && B > 0 && G > 0, "vae_loglik_elem_fwd: bad arguments");
  int blocks = (int)((B * G + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (zinb)
    hipLaunchKernelGGL(elem_fwd_kernel<true>, dim3(blocks), dim3(256), 0, st, x, ld_x, mu, ld_mu, theta, ld_theta, pi,
                       ld_pi, (int)B, (int)G, out, ld_out);
  else
    hipLaunchKernelGGL(elem_fwd_kernel<false>, dim3(blocks), dim3(256), 0, st, x, ld_x, mu, ld_mu, theta, ld_theta,
                       nullptr, 0, (int)B, (int)G, out, ld_out);
  VAE_CHECK_LAUNCH("vae_loglik_elem_fwd");
  return VAE_OK;
}

extern "C" int vae_loglik_elem_bwd(int zinb, const float* gout, int64_t ld_g, const float* x, int64_t ld_x,
                                   const float* mu, int64_t ld_mu, const float* theta, int64_t ld_theta, const float* pi,
                                   int64_t ld_pi, int64_t B, int64_t G, float* dmu, float* dtheta, float* dpi,
                                   void* stream) {
  VAE_REQUIRE(gout && x && mu && theta && dmu && dtheta && (!zinb || (pi && dpi)) && B > 0 && G > 0,
              "vae_loglik_elem_bwd: bad arguments");
  int blocks = (int)((B * G + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (zinb)
    hipLaunchKernelGGL(elem_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, gout, ld_g, x, ld_x, mu, ld_mu, theta,
                       ld_theta, pi, ld_pi, (int)B, (int)G, dmu, dtheta, dpi);
  else
    hipLaunchKernelGGL(elem_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, gout, ld_g, x, ld_x, mu, ld_mu, theta,
                       ld_theta, nullptr, 0, (int)B, (int)G, dmu, dtheta, nullptr);
  VAE_CHECK_LAUNCH("vae_loglik_elem_bwd");
  return VAE_OK;
}

extern "C" int vae_softmax_rate_fwd(const float* logit_scale, int64_t ld_s, const float* library, int64_t B, int64_t G,
                                    float* px_scale, float* px_rate, void* stream) {
  VAE_REQUIRE(logit_scale && library && px_scale && px_rate && B > 0 && G > 0 && ld_s >= G, "vae_softmax_rate_fwd: bad arguments");
  VAE_REQUIRE(G <= vae_nbzinb_max_genes(), "vae_softmax_rate_fwd: G=%lld exceeds %lld", (long long)G, (long long)vae_nbzinb_max_genes());
  const size_t shmem = (size_t)(((G + 3) & ~3) + 16) * sizeof(float);
  if (shmem > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute((const void*)softmax_rate_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return vae_set_error(VAE_ERR_LAUNCH, "vae_softmax_rate_fwd: cannot raise LDS limit: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(softmax_rate_fwd_kernel, dim3((unsigned)B), dim3(FWD_THREADS), shmem, (hipStream_t)stream, logit_scale,
                     ld_s, library, (int)G, px_scale, px_rate);
  VAE_CHECK_LAUNCH("vae_softmax_rate_fwd");
  return VAE_OK;
}

extern "C" int vae_softmax_rate_bwd(const float* px_scale, const float* library, const float* d_scale, const float* d_rate,
                                    int64_t B, int64_t G, float* dlogits, int64_t ld_d, float* dlibrary, void* stream) {
  VAE_REQUIRE(px_scale && library && dlogits && dlibrary && B > 0 && G > 0 && ld_d >= G, "vae_softmax_rate_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_rate_bwd_kernel, dim3((unsigned)B), dim3(FWD_THREADS), 0, (hipStream_t)stream, px_scale, library,
                     d_scale, d_rate, (int)G, dlogits, ld_d, dlibrary);
  VAE_CHECK_LAUNCH("vae_softmax_rate_bwd");
  return VAE_OK;
}
