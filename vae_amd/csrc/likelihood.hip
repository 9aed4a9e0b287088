// Element-wise NB / ZINB log-likelihood kernels behind the reference's function API
// (log_zinb_positive / log_nb_positive, distributions.py:22-90) with libm-grade special
// functions, plus the standalone softmax + library-scaling kernels behind DecoderSCVI.forward
// (modules.py:401-404).  The training hot path uses the fused kernels of epilogue.hip instead.
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

constexpr int FWD_THREADS = 512;

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

// ---- Poisson (reconstruction_loss == "poisson", task.py:167-168) ---------------------------------
// torch.distributions.Poisson.log_prob: xlogy(x, rate) - rate - lgamma(x + 1); d/d rate = x / rate - 1.
__global__ void poisson_fwd_kernel(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ rate, int64_t ld_r,
                                   int B, int G, float* __restrict__ out, int64_t ld_o) {
  const int64_t total = (int64_t)B * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / G, g = i % G;
    const float xx = x[b * ld_x + g], r = rate[b * ld_r + g];
    const float xl = xx == 0.f ? 0.f : xx * logf(r);   // xlogy
    out[b * ld_o + g] = xl - r - lgammaf(xx + 1.f);
  }
}
__global__ void poisson_bwd_kernel(const float* __restrict__ gout, int64_t ld_g, const float* __restrict__ x, int64_t ld_x,
                                   const float* __restrict__ rate, int64_t ld_r, int B, int G, float* __restrict__ drate,
                                   int64_t ld_d) {
  const int64_t total = (int64_t)B * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / G, g = i % G;
    const float xx = x[b * ld_x + g], r = rate[b * ld_r + g];
    drate[b * ld_d + g] = gout[b * ld_g + g] * ((xx == 0.f ? 0.f : xx / r) - 1.f);
  }
}

extern "C" int vae_poisson_fwd(const float* x, int64_t ld_x, const float* rate, int64_t ld_rate, int64_t B, int64_t G,
                               float* out, int64_t ld_out, void* stream) {
  VAE_REQUIRE(x && rate && out && B > 0 && G > 0 && ld_x >= G && ld_rate >= G && ld_out >= G, "vae_poisson_fwd: bad arguments");
  int blocks = (int)((B * G + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(poisson_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld_x, rate, ld_rate, (int)B, (int)G,
                     out, ld_out);
  VAE_CHECK_LAUNCH("vae_poisson_fwd");
  return VAE_OK;
}

extern "C" int vae_poisson_bwd(const float* gout, int64_t ld_g, const float* x, int64_t ld_x, const float* rate,
                               int64_t ld_rate, int64_t B, int64_t G, float* drate, int64_t ld_d, void* stream) {
  VAE_REQUIRE(gout && x && rate && drate && B > 0 && G > 0 && ld_g >= G && ld_x >= G && ld_rate >= G && ld_d >= G,
              "vae_poisson_bwd: bad arguments");
  int blocks = (int)((B * G + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(poisson_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gout, ld_g, x, ld_x, rate, ld_rate,
                     (int)B, (int)G, drate, ld_d);
  VAE_CHECK_LAUNCH("vae_poisson_bwd");
  return VAE_OK;
}

extern "C" int64_t vae_nbzinb_max_genes(void);

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
