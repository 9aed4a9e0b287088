// Row-wise kernels of the VAE hot path (gfx950): per-cell mean of the count row, the
// GELU -> LayerNorm half of an FCLayers layer (forward + backward), the fused
// reparameterisation-sample + Gaussian-KL kernel (forward + backward), column sums for bias
// gradients and the ELBO reduction.
//
// Layout rule: one 64-lane wavefront owns one cell (row); the hidden width (<= 1024) lives in
// registers, row reductions are wavefront shuffles, per-column reductions over cells are
// accumulated in registers across the rows a wave visits, combined through LDS and flushed
// with one float atomic per column per workgroup.
#include "../../include/vae_amd.h"
#include "common.h"
#include <stdlib.h>
#include "philox.h"
#include "internal.h"

namespace {

constexpr int MAXW = 1024;           // widest hidden layer a wave keeps in registers
constexpr int NPL_MAX = MAXW / WAVE;  // elements per lane at the widest layer; the kernels are instantiated for 2 / 4 / 8 / 16
                                      // (a 256-wide layer evaluated with 16 slots per lane spends 3/4 of its erf on padding)

// ---------------------------------------------------------------------------------------
// per-cell mean of x (Encoder.forward: x.mean(dim=-1), modules.py:264).  Counts are integers
// whose row sums stay below 2^24, so any summation order gives torch's result exactly.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ x, int64_t ld, int B, int G, int vec,
                                                       float* __restrict__ mean) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* row = x + (int64_t)b * ld;
  float s = 0.f;
  if (vec) {
    const int n4 = G >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = reinterpret_cast<const float4*>(row)[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
    for (int i = (n4 << 2) + threadIdx.x; i < G; i += 256) s += row[i];
  } else {
    for (int i = threadIdx.x; i < G; i += 256) s += row[i];
  }
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) mean[b] = s / (float)G;
}

// Row mean AND the normalised counts xn = x / mean - 1 (Encoder.forward, modules.py:264) in one
// launch: the row is summed from HBM, then re-read (L2-resident: one row is G*4 bytes) and written
// normalised.  The encoder GEMMs (forward and wgrad) then stream xn with plain LDS-DMA loads.
// REG: the whole row (G <= 256*4*EPT, vector-aligned) is held in registers: one HBM pass with
// every load of the row in flight.  XT = float, or uint8_t for counts kept in the reference's
// on-disk format (create_dataset.py:131-142): the expansion to fp32 happens here.
__device__ __forceinline__ float4 load4(const float* row, int i) { return reinterpret_cast<const float4*>(row)[i]; }
__device__ __forceinline__ float4 load4(const uint8_t* row, int i) {
  const uchar4 u = reinterpret_cast<const uchar4*>(row)[i];
  return make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
}

// OT = float, or uint16_t: xn written as bf16 (round to nearest even) for a bf16-mode step, whose encoder products would
// round it to bf16 when they read it anyway (same bits, half the bytes of the largest activation after the logits)
__device__ __forceinline__ uint16_t to_bf16(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ void store4(float* row, int i, const float4& v) { reinterpret_cast<float4*>(row)[i] = v; }
__device__ __forceinline__ void store4(uint16_t* row, int i, const float4& v) {
  uint2 u;
  u.x = (uint32_t)to_bf16(v.x) | ((uint32_t)to_bf16(v.y) << 16);
  u.y = (uint32_t)to_bf16(v.z) | ((uint32_t)to_bf16(v.w) << 16);
  reinterpret_cast<uint2*>(row)[i] = u;
}
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(uint16_t* p, float v) { *p = to_bf16(v); }

template <bool REG, typename XT, typename OT>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const XT* __restrict__ x, int64_t ld, int B, int G,
                                                             int vec, float* __restrict__ mean,
                                                             OT* __restrict__ xn, int64_t ld_xn,
                                                             float* __restrict__ absmax) {
  constexpr int EPT = 20;
  __shared__ float red[4];
  float am = 0.f;   // max|xn| of this thread's elements (absmax != null)
  auto track = [&](float v) { am = fmaxf(am, fabsf(v)); return v; };
  auto finish = [&]() {
    if (absmax) vae_wave_absmax_store(absmax + blockIdx.x * VAE_NORM_WAVES + (threadIdx.x >> 6), am);
  };
  const int b = blockIdx.x;
  const XT* row = x + (int64_t)b * ld;
  OT* orow = xn + (int64_t)b * ld_xn;
  if (REG) {
    const int n4 = G >> 2;
    float4 v[EPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = threadIdx.x + 256 * j;
      v[j] = i < n4 ? load4(row, i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    for (int i = (n4 << 2) + threadIdx.x; i < G; i += 256) s += (float)row[i];
    s = block_sum<4>(s, red);
    const float m = s / (float)G;
    if (threadIdx.x == 0) mean[b] = m;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < n4) {
        float4 o;
        o.x = track(v[j].x / m - 1.f); o.y = track(v[j].y / m - 1.f);
        o.z = track(v[j].z / m - 1.f); o.w = track(v[j].w / m - 1.f);
        store4(orow, i, o);
      }
    }
    for (int i = (n4 << 2) + threadIdx.x; i < G; i += 256) store1(orow + i, track((float)row[i] / m - 1.f));
    finish();
    return;
  }
  float s = 0.f;
  const int n4 = vec ? (G >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4 v = load4(row, i);
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < G; i += 256) s += (float)row[i];
  s = block_sum<4>(s, red);
  const float m = s / (float)G;
  if (threadIdx.x == 0) mean[b] = m;
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 v = load4(row, i);
    v.x = track(v.x / m - 1.f); v.y = track(v.y / m - 1.f); v.z = track(v.z / m - 1.f); v.w = track(v.w / m - 1.f);
    store4(orow, i, v);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < G; i += 256) store1(orow + i, track((float)row[i] / m - 1.f));
  finish();
}

// ---------------------------------------------------------------------------------------
// The count-operand form of the encoder's first layer (step.hip, VaeGemmCounts): x / mean - 1 is never materialised.
//   (x / m - 1) W^T = (x W^T) / m - sum_g W[:, g]          dW = dpre^T (x / m - 1) = (dpre / m)^T x - colsum(dpre) 1^T
// so the two gene-sized products stream the RAW counts -- exact in one scaled f16 term when they are integers below
// 2048 -- and this launch provides what they need: workgroups [HE, HE + B) read a cell's counts once (whole row in registers)
// and leave its mean (modules.py:264), the per-wave maxima of |x| (the operand scale) and per-wave flags "some count needs
// its lo term" (significant bits beyond f16's 11, or a magnitude from 2^15 on, where a scale that keeps the largest count
// in range could push a count of 1 out of it); workgroups [0, HE) sum one row of the (HE, G) first-layer weight each,
// in a fixed order (they are the first HE workgroups of the grid).  72 MB read, nothing gene-sized written (normalize_rows: 41 MB read + 41 MB written).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int64_t ld, int B, int G, int vec,
                                                        float* __restrict__ mean, float* __restrict__ absmax,
                                                        float* __restrict__ flags, const float* __restrict__ W, int64_t ldw,
                                                        int vecw, float* __restrict__ wsum) {
  constexpr int EPT = 20;
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int HE = (int)gridDim.x - B;
  if ((int)blockIdx.x < HE) {   // one weight row (these workgroups come FIRST: a tail of them would be exposed)
    const int h = blockIdx.x;
    const float* row = W + (int64_t)h * ldw;
    float s = 0.f;
    if (vecw && (G & 3) == 0 && G <= 256 * 4 * EPT) {   // the whole row in flight at once, as a count row below
      const int n4 = G >> 2;
      float4 v[EPT];
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int i = tid + 256 * j;
        v[j] = i < n4 ? load4(row, i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < EPT; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    } else {
      const int n4 = vecw ? (G >> 2) : 0;
      for (int i0 = tid; i0 < n4; i0 += 4 * 256) {   // four 16-byte loads in flight per lane
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + 256 * u < n4 ? load4(row, i0 + 256 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      }
      for (int i = (n4 << 2) + tid; i < G; i += 256) s += row[i];
    }
    s = block_sum<4>(s, red);
    if (tid == 0) wsum[h] = s;
    return;
  }
  const int b = blockIdx.x - HE;
  const float* row = x + (int64_t)b * ld;
  float s = 0.f, am = 0.f;
  uint32_t bits = 0u;
  auto take = [&](float v) { s += v; am = fmaxf(am, fabsf(v)); bits |= __float_as_uint(v); };
  if (vec && G <= 256 * 4 * EPT) {
    const int n4 = G >> 2;
    float4 v[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = tid + 256 * j;
      v[j] = i < n4 ? load4(row, i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {   // (the same association as normalize_rows_kernel: integer counts sum exactly anyway)
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      am = fmaxf(fmaxf(am, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
      bits |= (__float_as_uint(v[j].x) | __float_as_uint(v[j].y)) | (__float_as_uint(v[j].z) | __float_as_uint(v[j].w));
    }
  } else {
    const int n4 = vec ? (G >> 2) : 0;
    for (int i = tid; i < n4; i += 256) {
      const float4 v = load4(row, i);
      take(v.x); take(v.y); take(v.z); take(v.w);
    }
    for (int i = (n4 << 2) + tid; i < G; i += 256) take(row[i]);
  }
  // per wave: maximum and "needs a lo term"
  am = wave_max_dpp(am);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, o, 64);
  if ((tid & 63) == 0) {
    absmax[b * VAE_NORM_WAVES + (tid >> 6)] = am;
    flags[b * VAE_NORM_WAVES + (tid >> 6)] = ((bits & 0x1fffu) != 0u || !(am < 32768.f)) ? 1.f : 0.f;
  }
  s = block_sum<4>(s, red);
  if (tid == 0) mean[b] = s / (float)G;
}

// ---------------------------------------------------------------------------------------
// GELU(erf) -> LayerNorm(eps 1e-5, affine) forward (FCLayers, modules.py:112-114).
// `pre` = Linear output incl. bias.  Saves mean / rstd of the GELU output per row.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t bf16_bits(float v) {   // nearest even, as vae_cast_bf16
  const __bf16 h = (__bf16)v;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

template <int NPL>
__global__ __launch_bounds__(256) void gelu_ln_fwd_kernel(const float* __restrict__ pre, int64_t ld_pre,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int B, int W, float* __restrict__ out, int64_t ld_out,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          uint16_t* __restrict__ out16, int64_t ld16) {
  // out16 != null: the output again as bf16 (nearest even), for a bf16-storage product that reads it next
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* p = pre + (int64_t)row * ld_pre;
  float g[NPL];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int c = lane + j * WAVE;
    g[j] = c < W ? gelu_f(p[c]) : 0.f;
    s += g[j];
  }
  const float mu = wave_sum_dpp(s) / (float)W;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int c = lane + j * WAVE;
    const float d = c < W ? g[j] - mu : 0.f;
    q += d * d;
  }
  const float rs = 1.f / sqrtf(wave_sum(q) / (float)W + 1e-5f);
  float* o = out + (int64_t)row * ld_out;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int c = lane + j * WAVE;
    if (c < W) {
      const float v = (g[j] - mu) * rs * gamma[c] + beta[c];
      o[c] = v;
      if (out16) out16[(int64_t)row * ld16 + c] = bf16_bits(v);
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// Backward of the above plus the three per-column reductions over cells that sit on it:
// dgamma += sum_b dout*xhat, dbeta += sum_b dout, dbias += sum_b dpre (bias of the Linear
// that produced `pre`).  dpre may alias dout.
template <int NPL>
__global__ __launch_bounds__(256) void gelu_ln_bwd_kernel(const float* dout, int64_t ld_dout,
                                                          const float* __restrict__ pre, int64_t ld_pre,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, int B, int W, float* dpre,
                                                          int64_t ld_dpre, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dbias,
                                                          float* __restrict__ part, uint16_t* __restrict__ dpre16,
                                                          int64_t ld16) {
  // dpre16 != null: dpre again as bf16 (nearest even), for the bf16-storage weight-gradient product
  // part != null: the workgroup's three column sums go to part[(block * 3 + k) * W + c] (k = 0 dgamma, 1 dbeta, 2 dbias)
  // for a reduction in block order (vae_colpart_reduce: deterministic, no atomics); else float atomics on the outputs.
  __shared__ float comb[3][4][WAVE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NPL], ab[NPL], ap[NPL], gam[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    ag[j] = ab[j] = ap[j] = 0.f;
    const int c = lane + j * WAVE;
    gam[j] = c < W ? gamma[c] : 0.f;
  }
  // A wave walks its cells with the NEXT cell's loads issued before this cell's stores (dpre may alias dout, so the
  // compiler cannot hoist them itself): one memory latency per cell instead of two in a row.
  struct RowIn { float pv[NPL], dy[NPL], mu, rs; };
  auto load_row = [&](int row) {
    RowIn r;
    const float* p = pre + (int64_t)row * ld_pre;
    const float* d = dout + (int64_t)row * ld_dout;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int c = lane + j * WAVE;
      r.pv[j] = c < W ? p[c] : 0.f;
      r.dy[j] = c < W ? d[c] : 0.f;
    }
    r.mu = mean[row];
    r.rs = rstd[row];
    return r;
  };
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  RowIn cur, nxt;
  if (row < B) cur = load_row(row);
  for (; row < B; row += stride) {
    if (row + stride < B) nxt = load_row(row + stride);
    const float mu = cur.mu, rs = cur.rs;
    float xh[NPL], dx[NPL], pv[NPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int c = lane + j * WAVE;
      const bool ok = c < W;
      pv[j] = cur.pv[j];
      const float dy = cur.dy[j];
      xh[j] = ok ? (gelu_f(pv[j]) - mu) * rs : 0.f;
      dx[j] = dy * gam[j];
      s1 += dx[j];
      s2 += dx[j] * xh[j];
      ag[j] += dy * xh[j];
      ab[j] += dy;
    }
    s1 = wave_sum_dpp(s1) / (float)W;
    s2 = wave_sum_dpp(s2) / (float)W;
    float* o = dpre + (int64_t)row * ld_dpre;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int c = lane + j * WAVE;
      if (c < W) {
        const float dg = rs * (dx[j] - s1 - xh[j] * s2);
        const float v = dg * gelu_grad(pv[j]);
        o[c] = v;
        if (dpre16) dpre16[(int64_t)row * ld16 + c] = bf16_bits(v);
        ap[j] += v;
      }
    }
    cur = nxt;
  }
  // combine the 4 waves of the workgroup, then one atomic per column per workgroup
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    if (j * WAVE >= W) break;
    __syncthreads();
    comb[0][wave][lane] = ag[j];
    comb[1][wave][lane] = ab[j];
    comb[2][wave][lane] = ap[j];
    __syncthreads();
    if (wave < 3) {
      const int c = lane + j * WAVE;
      if (c < W) {
        const float v = comb[wave][0][lane] + comb[wave][1][lane] + comb[wave][2][lane] + comb[wave][3][lane];
        float* dst = wave == 0 ? dgamma : (wave == 1 ? dbeta : dbias);
        if (part) part[((int64_t)blockIdx.x * 3 + wave) * W + c] = v;
        else if (dst) atomicAdd(dst + c, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// reparameterisation sample + Gaussian KL (Encoder.forward modules.py:268-270 with
// reparameterize_gaussian :11-12, and LitVAE.gene_loss task.py:188-198).
// heads[b, 0:L] = q_m, heads[b, L] = pre-activation of the ONE shared variance (modules.py:249).
//   q_v = softplus(.) + 1e-4 ; z = q_m + eps * sqrt(q_v)
//   kl  = sum_j 0.5 * (r + ((q_m_j - pm)/sp)^2 - 1 - log r),  r = (sqrt(q_v)/sp)^2
// One wavefront per cell: lane j owns latent dimension j (L <= 64), reductions are shuffles.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ heads, int64_t ld_h,
                                                             const float* __restrict__ eps, int64_t ld_e, int B, int L,
                                                             float prior_mean, float prior_var, float* __restrict__ qv,
                                                             float* __restrict__ z, int64_t ld_z, float* __restrict__ kl) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* hrow = heads + (int64_t)b * ld_h;
  const float v = softplus20(hrow[L]) + 1e-4f;
  const float sq = sqrtf(v), sp = sqrtf(prior_var);
  const float r0 = sq / sp;
  const float r = r0 * r0;
  const float lr = logf(r);
  float acc = 0.f;
  if (lane < L) {
    const float m = hrow[lane];
    const float t = (m - prior_mean) / sp;
    acc = 0.5f * (r + t * t - 1.f - lr);
    z[(int64_t)b * ld_z + lane] = m + eps[(int64_t)b * ld_e + lane] * sq;
  }
  acc = wave_sum_dpp(acc);
  if (lane == 0) {
    qv[b] = v;
    kl[b] = acc;
  }
}

// Backward: dheads from dz (gradient reaching the sample) and the upstream gradient on kl[b]
// (g_kl_scalar * g_kl[b], g_kl may be null).  dbias (nullable, [L+1]) accumulates the column sums
// of dheads = the bias gradient of the [mean; variance] heads.
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ dz, int64_t ld_dz,
                                                             const float* __restrict__ g_kl, float g_kl_scalar,
                                                             const float* __restrict__ heads, int64_t ld_h,
                                                             const float* __restrict__ eps, int64_t ld_e,
                                                             const float* __restrict__ qv, int B, int L, float prior_mean,
                                                             float prior_var, float* __restrict__ dheads, int64_t ld_dh,
                                                             float* __restrict__ dbias, float* __restrict__ part) {
  __shared__ float comb[4][WAVE + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float col = 0.f, colv = 0.f;  // column sums: lane j < L -> dq_m[j]; colv (lane 0) -> d var pre-activation
  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    const float gk = g_kl_scalar * (g_kl ? g_kl[b] : 1.f);
    const float* hrow = heads + (int64_t)b * ld_h;
    const float v = qv[b];
    const float sq = sqrtf(v);
    float dv = 0.f;
    if (lane < L) {
      const float d = dz[(int64_t)b * ld_dz + lane];
      const float o = d + gk * (hrow[lane] - prior_mean) / prior_var;
      dheads[(int64_t)b * ld_dh + lane] = o;
      col += o;
      dv = d * eps[(int64_t)b * ld_e + lane];
    }
    dv = wave_sum_dpp(dv);
    // d z_j / d v = eps_j / (2 sqrt v);  d kl / d v = L * 0.5 * (1/pv - 1/v)
    dv = dv * 0.5f / sq + gk * (float)L * 0.5f * (1.f / prior_var - 1.f / v);
    const float o = dv * softplus20_grad(hrow[L]);
    if (lane == 0) {
      dheads[(int64_t)b * ld_dh + L] = o;
      colv += o;
    }
  }
  if (dbias || part) {   // part != null: this workgroup's L + 1 column sums, for vae_colpart_reduce
    comb[wave][lane] = col;
    if (lane == 0) comb[wave][WAVE] = colv;
    __syncthreads();
    if (wave == 0) {
      const float a = comb[0][lane] + comb[1][lane] + comb[2][lane] + comb[3][lane];
      const float v = comb[0][WAVE] + comb[1][WAVE] + comb[2][WAVE] + comb[3][WAVE];
      if (part) {
        if (lane < L) part[(int64_t)blockIdx.x * (L + 1) + lane] = a;
        if (lane == 0) part[(int64_t)blockIdx.x * (L + 1) + L] = v;
      } else {
        if (lane < L) atomicAdd(dbias + lane, a);
        if (lane == 0) atomicAdd(dbias + L, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// column sums over cells: out[n] += sum_m in[m, n]   (bias gradients of small Linears)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int64_t ld, int M, int N,
                                                     int rows_per_block, float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += in[(int64_t)r * ld + n];
  atomicAdd(out + n, s);
}

__global__ void add_inplace_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] += src[i];
}

// ---------------------------------------------------------------------------------------
// ELBO reduction (LitVAE.gene_loss task.py:202 and training_step :489):
//   out[0] = mean(recon) = -mean(ll), out[1] = mean(kl_z), out[2] = mean(kl_l),
//   out[3] = gene_loss = out[0] + kl_weight*(mean(kl_l + kl_z)), out[4] = coeff*gene_loss
// single workgroup, fixed summation order (deterministic).
// ---------------------------------------------------------------------------------------
// part != null: ll[i] / S[i] are first formed (and stored) from the per-strip partials of the streaming likelihood pass:
// thread pairs (row, k) with four strips in flight, strip order fixed (deterministic)
__global__ __launch_bounds__(1024) void elbo_reduce_kernel(float* __restrict__ ll, const float* __restrict__ kl_z,
                                                           const float* __restrict__ kl_l, int B, float kl_weight,
                                                           float coeff, float* __restrict__ out,
                                                           float* __restrict__ recon, const float* __restrict__ part,
                                                           int strips, float* __restrict__ S) {
  if (part) {
    for (int j = threadIdx.x; j < 2 * B; j += 1024) {
      const int i = j >> 1, k = j & 1;
      const float* p = part + (int64_t)k * B + i;
      const int64_t st = (int64_t)2 * B;
      float v = 0.f;
      int s = 0;
      for (; s + 8 <= strips; s += 8) {   // eight loads in flight, summed in strip order
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
      (k ? S : ll)[i] = v;
    }
    __syncthreads();   // ll written by this (the only) workgroup is read below
  }
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int i = threadIdx.x; i < B; i += 1024) {
    if (recon) recon[i] = -ll[i];
    a -= ll[i];
    b += kl_z[i];
    c += kl_l[i];
    d += kl_l[i] + kl_z[i];
  }
  // the four block sums share one pair of barriers (the per-wave order of additions is that of block_sum)
  __shared__ float red4[4][16];
  a = wave_sum_dpp(a); b = wave_sum_dpp(b); c = wave_sum_dpp(c); d = wave_sum_dpp(d);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red4[0][w] = a; red4[1][w] = b; red4[2][w] = c; red4[3][w] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = b = c = d = 0.f;
    for (int w = 0; w < 16; ++w) { a += red4[0][w]; b += red4[1][w]; c += red4[2][w]; d += red4[3][w]; }
  }
  if (threadIdx.x == 0) {
    const float n = (float)B;
    const float gl = a / n + kl_weight * (d / n);
    out[0] = a / n;
    out[1] = b / n;
    out[2] = c / n;
    out[3] = gl;
    out[4] = coeff * gl;
  }
}

}  // namespace

extern "C" int vae_row_mean(const float* x, int64_t ldx, int64_t B, int64_t G, float* mean, void* stream) {
  VAE_REQUIRE(x && mean && B > 0 && G > 0 && ldx >= G, "vae_row_mean: bad arguments");
  const int vec = (ldx % 4 == 0) && vae_aligned16(x);
  hipLaunchKernelGGL(row_mean_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)B, (int)G, vec,
                     mean);
  VAE_CHECK_LAUNCH("vae_row_mean");
  return VAE_OK;
}

// ---- GeneDropout (modules.py:190-209), z-encoder input only ---------------------------------------
// counts[g] += #cells of this chunk with x[b,g] > 0   (integer atomics: order-independent, deterministic)
template <typename XT>
__global__ __launch_bounds__(256) void gene_count_kernel(const XT* __restrict__ x, int64_t ld, int B, int G, int rows,
                                                         int* __restrict__ counts) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const int b0 = blockIdx.y * rows, b1 = min(B, b0 + rows);
  int c = 0;
  for (int b = b0; b < b1; ++b) c += (float)x[(int64_t)b * ld + g] > 0.f;
  if (c) atomicAdd(counts + g, c);
}
// One workgroup: running = alpha * running + (1 - alpha) * counts / B;  p_g = clip(p * running_g / mean(running), 0, 0.9);
// threshold_g = floor(p_g * 2^32); stats = {mean_g p_g}.
__global__ __launch_bounds__(1024) void gene_prob_kernel(const int* __restrict__ counts, int B, int G, float p, float alpha,
                                                         float* __restrict__ running, uint32_t* __restrict__ thr,
                                                         float* __restrict__ stats) {
  __shared__ float red[16];
  float s = 0.f;
  for (int g = threadIdx.x; g < G; g += 1024) {
    const float nz = (float)counts[g] / (float)B;
    const float r = alpha * running[g] + (1.f - alpha) * nz;
    running[g] = r;
    s += r;
  }
  s = block_sum<16>(s, red);
  const float rmean = s / (float)G;
  __syncthreads();
  float ps = 0.f;
  for (int g = threadIdx.x; g < G; g += 1024) {
    const float pg = fminf(fmaxf(p * running[g] / rmean, 0.f), 0.9f);
    thr[g] = (uint32_t)((double)pg * 4294967296.0);   // pg <= 0.9: no overflow
    ps += pg;
  }
  ps = block_sum<16>(ps, red);
  if (threadIdx.x == 0) stats[0] = ps / (float)G;
}
// x/mean - 1 of the MASKED row: xz = x * keep, mean over all G genes, out = xz / mean - 1.  The reference also
// divides the masked counts by mean_g(p_g) (modules.py:209); the row normalisation that follows (modules.py:264)
// cancels any per-batch scale, so it is not applied.  Requires G % 4 == 0 (vector path), any G.
template <typename XT, typename OT>
__global__ __launch_bounds__(256) void normalize_rows_masked_kernel(const XT* __restrict__ x, int64_t ld, int B, int G,
                                                                    const uint32_t* __restrict__ thr, uint64_t seed,
                                                                    uint32_t step, float* __restrict__ mean,
                                                                    OT* __restrict__ xn, int64_t ld_xn,
                                                                    float* __restrict__ absmax) {
  __shared__ float red[4];
  float am = 0.f;
  const int b = blockIdx.x;
  const XT* row = x + (int64_t)b * ld;
  OT* orow = xn + (int64_t)b * ld_xn;
  const int n4 = G >> 2;
  auto masked = [&](int i) {
    float4 v = load4(row, i);
    const uint4 t = reinterpret_cast<const uint4*>(thr)[i];
    const uint4 w = vae_philox4(seed, step, VAE_SITE_GENES, (uint64_t)b * (uint64_t)n4 + (uint64_t)i);
    v.x = w.x < t.x ? v.x : 0.f; v.y = w.y < t.y ? v.y : 0.f;
    v.z = w.z < t.z ? v.z : 0.f; v.w = w.w < t.w ? v.w : 0.f;
    return v;
  };
  float s = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4 v = masked(i);
    s += (v.x + v.y) + (v.z + v.w);
  }
  s = block_sum<4>(s, red);
  const float m = s / (float)G;
  if (threadIdx.x == 0) mean[b] = m;
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 v = masked(i);
    v.x = v.x / m - 1.f; v.y = v.y / m - 1.f; v.z = v.z / m - 1.f; v.w = v.w / m - 1.f;
    am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    store4(orow, i, v);
  }
  if (absmax) vae_wave_absmax_store(absmax + blockIdx.x * VAE_NORM_WAVES + (threadIdx.x >> 6), am);
}

template <typename XT, typename OT = float>
static int gene_dropout_impl(const XT* x, int64_t ldx, int64_t B, int64_t G, float p, float alpha, uint64_t seed,
                             uint32_t step, float* running, int* counts, uint32_t* thr, float* stats, float* mean, OT* xn,
                             int64_t ld_xn, float* absmax, void* stream) {
  VAE_REQUIRE(x && running && counts && thr && stats && mean && xn && B > 0 && G > 0 && ldx >= G && ld_xn >= G,
              "vae_gene_dropout: bad arguments");
  VAE_REQUIRE(p > 0.f && p <= 1.f && alpha >= 0.f && alpha <= 1.f, "vae_gene_dropout: p in (0, 1], alpha in [0, 1]");
  const bool x_ok = sizeof(XT) == 1 ? ((reinterpret_cast<uintptr_t>(x) & 3u) == 0) : vae_aligned16(x);
  VAE_REQUIRE(G % 4 == 0 && ldx % 4 == 0 && ld_xn % (sizeof(OT) == 2 ? 8 : 4) == 0 && x_ok && vae_aligned16(xn) && vae_aligned16(thr),
              "vae_gene_dropout: needs G %% 4 == 0, leading dimensions %% 4 == 0 and aligned rows");
  hipStream_t st = (hipStream_t)stream;
  const int rows = 32;
  VAE_REQUIRE(hipMemsetAsync(counts, 0, (size_t)G * sizeof(int), st) == hipSuccess, "vae_gene_dropout: hipMemsetAsync failed");
  hipLaunchKernelGGL((gene_count_kernel<XT>), dim3((unsigned)((G + 255) / 256), (unsigned)((B + rows - 1) / rows)), dim3(256), 0,
                     st, x, ldx, (int)B, (int)G, rows, counts);
  VAE_CHECK_LAUNCH("vae_gene_dropout(count)");
  hipLaunchKernelGGL(gene_prob_kernel, dim3(1), dim3(1024), 0, st, counts, (int)B, (int)G, p, alpha, running, thr, stats);
  VAE_CHECK_LAUNCH("vae_gene_dropout(prob)");
  hipLaunchKernelGGL((normalize_rows_masked_kernel<XT, OT>), dim3((unsigned)B), dim3(256), 0, st, x, ldx, (int)B, (int)G, thr,
                     seed, step, mean, xn, ld_xn, absmax);
  VAE_CHECK_LAUNCH("vae_gene_dropout(normalize)");
  return VAE_OK;
}

int vae_gene_dropout_ex(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float p, float alpha, uint64_t seed,
                        uint32_t step, float* running, int32_t* counts, uint32_t* threshold, float* stats, float* mean,
                        void* xn_v, int xn_bf16, int64_t ld_xn, float* absmax, void* stream) {
  VAE_REQUIRE(x_dtype == VAE_X_F32 || x_dtype == VAE_X_U8, "vae_gene_dropout: bad x_dtype");
  if (xn_bf16) {
    uint16_t* xn16 = (uint16_t*)xn_v;
    if (x_dtype == VAE_X_U8)
      return gene_dropout_impl<uint8_t, uint16_t>((const uint8_t*)x, ldx, B, G, p, alpha, seed, step, running, counts, threshold,
                                                  stats, mean, xn16, ld_xn, absmax, stream);
    return gene_dropout_impl<float, uint16_t>((const float*)x, ldx, B, G, p, alpha, seed, step, running, counts, threshold, stats,
                                              mean, xn16, ld_xn, absmax, stream);
  }
  float* xn = (float*)xn_v;
  if (x_dtype == VAE_X_U8)
    return gene_dropout_impl<uint8_t>((const uint8_t*)x, ldx, B, G, p, alpha, seed, step, running, counts, threshold, stats,
                                      mean, xn, ld_xn, absmax, stream);
  return gene_dropout_impl<float>((const float*)x, ldx, B, G, p, alpha, seed, step, running, counts, threshold, stats, mean,
                                  xn, ld_xn, absmax, stream);
}

extern "C" int vae_gene_dropout(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float p, float alpha,
                                uint64_t seed, uint32_t step, float* running, int32_t* counts, uint32_t* threshold,
                                float* stats, float* mean, float* xn, int64_t ld_xn, void* stream) {
  return vae_gene_dropout_ex(x, x_dtype, ldx, B, G, p, alpha, seed, step, running, counts, threshold, stats, mean, xn, 0, ld_xn,
                             nullptr, stream);
}

template <typename XT, typename OT = float>
static int normalize_rows_impl(const XT* x, int64_t ldx, int64_t B, int64_t G, float* mean, OT* xn, int64_t ld_xn, float* absmax,
                               void* stream) {
  VAE_REQUIRE(x && mean && xn && B > 0 && G > 0 && ldx >= G && ld_xn >= G, "vae_normalize_rows: bad arguments");
  const bool x_ok = sizeof(XT) == 1 ? ((reinterpret_cast<uintptr_t>(x) & 3u) == 0) : vae_aligned16(x);
  const int vec = (ldx % 4 == 0) && (ld_xn % (sizeof(OT) == 2 ? 8 : 4) == 0) && x_ok && vae_aligned16(xn);
  if (vec && G <= 256 * 4 * 20)
    hipLaunchKernelGGL((normalize_rows_kernel<true, XT, OT>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (int)B, (int)G, vec, mean, xn, ld_xn, absmax);
  else
    hipLaunchKernelGGL((normalize_rows_kernel<false, XT, OT>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (int)B, (int)G, vec, mean, xn, ld_xn, absmax);
  VAE_CHECK_LAUNCH("vae_normalize_rows");
  return VAE_OK;
}

int vae_normalize_rows_ex(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, float* mean, void* xn_v, int xn_bf16,
                          int64_t ld_xn, float* absmax, void* stream) {
  if (xn_bf16) {
    if (x_dtype == VAE_X_U8)
      return normalize_rows_impl<uint8_t, uint16_t>((const uint8_t*)x, ldx, B, G, mean, (uint16_t*)xn_v, ld_xn, absmax, stream);
    return normalize_rows_impl<float, uint16_t>((const float*)x, ldx, B, G, mean, (uint16_t*)xn_v, ld_xn, absmax, stream);
  }
  float* xn = (float*)xn_v;
  if (x_dtype == VAE_X_U8)
    return normalize_rows_impl<uint8_t>((const uint8_t*)x, ldx, B, G, mean, xn, ld_xn, absmax, stream);
  return normalize_rows_impl<float>((const float*)x, ldx, B, G, mean, xn, ld_xn, absmax, stream);
}

extern "C" int vae_normalize_rows(const float* x, int64_t ldx, int64_t B, int64_t G, float* mean, float* xn,
                                  int64_t ld_xn, void* stream) {
  return normalize_rows_impl<float>(x, ldx, B, G, mean, xn, ld_xn, nullptr, stream);
}

extern "C" int vae_normalize_rows_u8(const uint8_t* x, int64_t ldx, int64_t B, int64_t G, float* mean, float* xn,
                                     int64_t ld_xn, void* stream) {
  return normalize_rows_impl<uint8_t>(x, ldx, B, G, mean, xn, ld_xn, nullptr, stream);
}

// mean[b] of the raw fp32 counts, absmax / flags [B * VAE_NORM_WAVES] and wsum[h] = sum_g W[h, g], h < HE (row_stats_kernel)
namespace {
// uint8 counts -> fp32 counts, V bytes per thread (16: one 16-byte load, four 16-byte stores; 4: rows that are only
// 4-byte aligned).  Grid: (column blocks, rows).
template <int V>
__global__ __launch_bounds__(256) void cast_u8_f32_kernel(const uint8_t* __restrict__ x, int64_t ldx, int G, float* __restrict__ out,
                                                          int64_t ldo) {
  const int b = blockIdx.y;
  const int g0 = (blockIdx.x * 256 + threadIdx.x) * V;
  if (g0 >= G) return;
  const uint8_t* p = x + (int64_t)b * ldx + g0;
  float* o = out + (int64_t)b * ldo + g0;
  uint32_t w[V / 4];
  if (V == 16) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    w[0] = q.x; w[V / 4 > 1 ? 1 : 0] = q.y; w[V / 4 > 2 ? 2 : 0] = q.z; w[V / 4 > 3 ? 3 : 0] = q.w;
  } else {
    w[0] = *reinterpret_cast<const uint32_t*>(p);
  }
#pragma unroll
  for (int i = 0; i < V / 4; ++i)
    if (g0 + 4 * i < G)   // (G % 4 == 0: whole quads)
      reinterpret_cast<float4*>(o)[i] = make_float4((float)(w[i] & 0xffu), (float)((w[i] >> 8) & 0xffu), (float)((w[i] >> 16) & 0xffu),
                                                    (float)(w[i] >> 24));
}
}  // namespace
// out[b, g] = (float) x[b, g], g < G (G % 4 == 0, rows of x 4-byte aligned, rows of out 16-byte aligned): the fp32 copy of a
// uint8 batch that the count-operand encoder products read (step.hip)
int vae_cast_u8_f32(const void* x, int64_t ldx, int64_t B, int64_t G, float* out, int64_t ldo, void* stream) {
  VAE_REQUIRE(x && out && B > 0 && G > 0 && G % 4 == 0 && ldx >= G && ldo >= G && ldx % 4 == 0 && ldo % 4 == 0 && vae_aligned16(out) &&
                  (reinterpret_cast<uintptr_t>(x) & 3u) == 0 && B < 65536,
              "vae_cast_u8_f32: bad arguments");
  const bool v16 = ldx % 16 == 0 && vae_aligned16(x);
  if (v16)
    hipLaunchKernelGGL(cast_u8_f32_kernel<16>, dim3((unsigned)((G + 4095) / 4096), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)x, ldx, (int)G, out, ldo);
  else
    hipLaunchKernelGGL(cast_u8_f32_kernel<4>, dim3((unsigned)((G + 1023) / 1024), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)x, ldx, (int)G, out, ldo);
  VAE_CHECK_LAUNCH("vae_cast_u8_f32");
  return VAE_OK;
}

extern "C" int vae_row_stats(const float* x, int64_t ldx, int64_t B, int64_t G, float* mean, float* absmax, float* flags, const float* W,
                  int64_t ldw, int64_t HE, float* wsum, void* stream) {
  VAE_REQUIRE(x && mean && absmax && flags && W && wsum && B > 0 && G > 0 && HE > 0 && ldx >= G && ldw >= G && B + HE < (1 << 30),
              "vae_row_stats: bad arguments");
  const int vec = (G % 4 == 0) && (ldx % 4 == 0) && vae_aligned16(x);
  const int vecw = (ldw % 4 == 0) && vae_aligned16(W);
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)(B + HE)), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)B, (int)G, vec, mean,
                     absmax, flags, W, ldw, vecw, wsum);
  VAE_CHECK_LAUNCH("vae_row_stats");
  return VAE_OK;
}

extern "C" int vae_gelu_ln_fwd(const float* pre, int64_t ld_pre, const float* gamma, const float* beta, int64_t B,
                               int64_t W, float* out, int64_t ld_out, float* mean, float* rstd, void* stream) {
  return vae_gelu_ln_fwd_ex(pre, ld_pre, gamma, beta, B, W, out, ld_out, mean, rstd, nullptr, 0, stream);
}
// out16 != null: the output again as bf16 with row stride ld16
int vae_gelu_ln_fwd_ex(const float* pre, int64_t ld_pre, const float* gamma, const float* beta, int64_t B, int64_t W, float* out,
                       int64_t ld_out, float* mean, float* rstd, uint16_t* out16, int64_t ld16, void* stream) {
  VAE_REQUIRE(!out16 || ld16 >= W, "vae_gelu_ln_fwd: bf16 leading dimension too small");
  VAE_REQUIRE(pre && gamma && beta && out && mean && rstd, "vae_gelu_ln_fwd: null pointer");
  VAE_REQUIRE(B > 0 && W > 0 && W <= MAXW, "vae_gelu_ln_fwd: width %lld not in [1, %d]", (long long)W, MAXW);
  VAE_REQUIRE(ld_pre >= W && ld_out >= W, "vae_gelu_ln_fwd: leading dimension too small");
#define VAE_LN_FWD(N) hipLaunchKernelGGL(gelu_ln_fwd_kernel<N>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pre, \
                                         ld_pre, gamma, beta, (int)B, (int)W, out, ld_out, mean, rstd, out16, ld16)
  if (W <= 2 * WAVE) VAE_LN_FWD(2);
  else if (W <= 4 * WAVE) VAE_LN_FWD(4);
  else if (W <= 8 * WAVE) VAE_LN_FWD(8);
  else VAE_LN_FWD(NPL_MAX);
#undef VAE_LN_FWD
  VAE_CHECK_LAUNCH("vae_gelu_ln_fwd");
  return VAE_OK;
}

extern "C" int vae_gelu_ln_bwd(const float* dout, int64_t ld_dout, const float* pre, int64_t ld_pre, const float* mean,
                               const float* rstd, const float* gamma, int64_t B, int64_t W, float* dpre, int64_t ld_dpre,
                               float* dgamma, float* dbeta, float* dbias, void* stream) {
  return vae_gelu_ln_bwd_part(dout, ld_dout, pre, ld_pre, mean, rstd, gamma, B, W, dpre, ld_dpre, dgamma, dbeta, dbias, nullptr,
                              nullptr, nullptr, 0, stream);
}
int vae_colpart_blocks(int64_t B) { return (int)(((B + 3) / 4) < 256 ? ((B + 3) / 4) : 256); }
// part != null: no atomics; the (*blocks_out, 3, W) partial column sums are left in `part` (>= 256 * 3 * W floats)
// dpre16 != null: dpre again as bf16 with row stride ld16
int vae_gelu_ln_bwd_part(const float* dout, int64_t ld_dout, const float* pre, int64_t ld_pre, const float* mean,
                         const float* rstd, const float* gamma, int64_t B, int64_t W, float* dpre, int64_t ld_dpre,
                         float* dgamma, float* dbeta, float* dbias, float* part, int* blocks_out, uint16_t* dpre16, int64_t ld16,
                         void* stream) {
  VAE_REQUIRE(!dpre16 || ld16 >= W, "vae_gelu_ln_bwd: bf16 leading dimension too small");
  VAE_REQUIRE(dout && pre && mean && rstd && gamma && dpre, "vae_gelu_ln_bwd: null pointer");
  VAE_REQUIRE(B > 0 && W > 0 && W <= MAXW, "vae_gelu_ln_bwd: width %lld not in [1, %d]", (long long)W, MAXW);
  VAE_REQUIRE(ld_dout >= W && ld_pre >= W && ld_dpre >= W, "vae_gelu_ln_bwd: leading dimension too small");
  int blocks = (int)((B + 3) / 4);
  static const int cap = getenv("VAE_LN_BWD_BLOCKS") ? atoi(getenv("VAE_LN_BWD_BLOCKS")) : 256;   // tuning knob
  if (blocks > cap) blocks = cap;
  if (part) blocks = vae_colpart_blocks(B);
  if (blocks_out) *blocks_out = blocks;
#define VAE_LN_BWD(N) hipLaunchKernelGGL(gelu_ln_bwd_kernel<N>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, ld_dout, pre, \
                                         ld_pre, mean, rstd, gamma, (int)B, (int)W, dpre, ld_dpre, dgamma, dbeta, dbias, part, dpre16, ld16)
  if (W <= 2 * WAVE) VAE_LN_BWD(2);
  else if (W <= 4 * WAVE) VAE_LN_BWD(4);
  else if (W <= 8 * WAVE) VAE_LN_BWD(8);
  else VAE_LN_BWD(NPL_MAX);
#undef VAE_LN_BWD
  VAE_CHECK_LAUNCH("vae_gelu_ln_bwd");
  return VAE_OK;
}

extern "C" int vae_reparam_kl_fwd(const float* heads, int64_t ld_heads, const float* eps, int64_t ld_eps, int64_t B,
                                  int64_t L, float prior_mean, float prior_var, float* qv, float* z, int64_t ld_z,
                                  float* kl, void* stream) {
  VAE_REQUIRE(heads && eps && qv && z && kl, "vae_reparam_kl_fwd: null pointer");
  VAE_REQUIRE(B > 0 && L > 0 && ld_heads >= L + 1 && ld_eps >= L && ld_z >= L, "vae_reparam_kl_fwd: bad shape");
  VAE_REQUIRE(prior_var > 0.f, "vae_reparam_kl_fwd: prior variance must be positive");
  VAE_REQUIRE(L <= WAVE, "vae_reparam_kl_fwd: latent size %lld exceeds %d", (long long)L, WAVE);
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, heads,
                     ld_heads, eps, ld_eps, (int)B, (int)L, prior_mean, prior_var, qv, z, ld_z, kl);
  VAE_CHECK_LAUNCH("vae_reparam_kl_fwd");
  return VAE_OK;
}

extern "C" int vae_reparam_kl_bwd(const float* dz, int64_t ld_dz, const float* g_kl, float g_kl_scalar,
                                  const float* heads, int64_t ld_heads, const float* eps, int64_t ld_eps,
                                  const float* qv, int64_t B, int64_t L, float prior_mean, float prior_var,
                                  float* dheads, int64_t ld_dheads, float* dbias, void* stream) {
  return vae_reparam_kl_bwd_part(dz, ld_dz, g_kl, g_kl_scalar, heads, ld_heads, eps, ld_eps, qv, B, L, prior_mean, prior_var, dheads,
                                 ld_dheads, dbias, nullptr, nullptr, stream);
}
// part != null: no atomics; the (*blocks_out, L + 1) partial column sums are left in `part` (>= 256 * (L + 1) floats)
int vae_reparam_kl_bwd_part(const float* dz, int64_t ld_dz, const float* g_kl, float g_kl_scalar, const float* heads,
                            int64_t ld_heads, const float* eps, int64_t ld_eps, const float* qv, int64_t B, int64_t L,
                            float prior_mean, float prior_var, float* dheads, int64_t ld_dheads, float* dbias, float* part,
                            int* blocks_out, void* stream) {
  VAE_REQUIRE(dz && heads && eps && qv && dheads, "vae_reparam_kl_bwd: null pointer");
  VAE_REQUIRE(B > 0 && L > 0 && L <= WAVE && ld_heads >= L + 1 && ld_eps >= L && ld_dz >= L && ld_dheads >= L + 1,
              "vae_reparam_kl_bwd: bad shape");
  int blocks = (int)((B + 3) / 4);
  static const int cap = getenv("VAE_REPARAM_BWD_BLOCKS") ? atoi(getenv("VAE_REPARAM_BWD_BLOCKS")) : 256;   // tuning knob
  if (blocks > cap) blocks = cap;  // few workgroups: one float atomic per column per workgroup
  if (part) blocks = vae_colpart_blocks(B);
  if (blocks_out) *blocks_out = blocks;
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, ld_dz, g_kl,
                     g_kl_scalar, heads, ld_heads, eps, ld_eps, qv, (int)B, (int)L, prior_mean, prior_var, dheads,
                     ld_dheads, dbias, part);
  VAE_CHECK_LAUNCH("vae_reparam_kl_bwd");
  return VAE_OK;
}

extern "C" int vae_colsum(const float* in, int64_t ld, int64_t M, int64_t N, float* out, void* stream) {
  VAE_REQUIRE(in && out && M > 0 && N > 0 && ld >= N, "vae_colsum: bad arguments");
  int row_blocks = (int)((M + 63) / 64);
  if (row_blocks > 64) row_blocks = 64;
  const int rows_per_block = (int)((M + row_blocks - 1) / row_blocks);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 255) / 256), row_blocks), dim3(256), 0, (hipStream_t)stream, in,
                     ld, (int)M, (int)N, rows_per_block, out);
  VAE_CHECK_LAUNCH("vae_colsum");
  return VAE_OK;
}

extern "C" int vae_elbo_reduce(const float* ll, const float* kl_z, const float* kl_l, int64_t B, float kl_weight,
                               float gene_loss_coeff, float* out5, float* recon, void* stream) {
  VAE_REQUIRE(ll && kl_z && kl_l && out5 && B > 0, "vae_elbo_reduce: bad arguments");
  hipLaunchKernelGGL(elbo_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, const_cast<float*>(ll), kl_z, kl_l, (int)B,
                     kl_weight, gene_loss_coeff, out5, recon, (const float*)nullptr, 0, (float*)nullptr);
  VAE_CHECK_LAUNCH("vae_elbo_reduce");
  return VAE_OK;
}

int vae_elbo_reduce_parts(const float* part, int strips, float* ll, float* S, const float* kl_z, const float* kl_l, int64_t B,
                          float kl_weight, float gene_loss_coeff, float* out5, float* recon, void* stream) {
  VAE_REQUIRE(part && strips > 0 && ll && S && kl_z && kl_l && out5 && B > 0, "vae_elbo_reduce_parts: bad arguments");
  hipLaunchKernelGGL(elbo_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ll, kl_z, kl_l, (int)B, kl_weight,
                     gene_loss_coeff, out5, recon, part, strips, S);
  VAE_CHECK_LAUNCH("vae_elbo_reduce_parts");
  return VAE_OK;
}

// x[b, c] *= keep(b, c) / (1 - p): nn.Dropout (modules.py:115) with the mask of philox.h.  Applied to an
// activation in the forward pass and to the gradient arriving at the same place in the backward pass.
__global__ __launch_bounds__(256) void dropout_apply_kernel(float* __restrict__ x, int64_t ld, int64_t B, int W, VaeDropout d,
                                                            uint32_t site, uint16_t* __restrict__ x16, int64_t ld16) {
  const int64_t total = B * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / W;
    const int c = (int)(i - b * W);
    const float v = x[b * ld + c] * d.mult(site, b, W, c);
    x[b * ld + c] = v;
    if (x16) x16[b * ld16 + c] = bf16_bits(v);   // the result again as bf16
  }
}

extern "C" int vae_dropout_apply(float* x, int64_t ld, int64_t B, int64_t W, float p, uint64_t seed, uint32_t step,
                                 uint32_t site, void* stream) {
  return vae_dropout_apply_ex(x, ld, B, W, p, seed, step, site, nullptr, 0, stream);
}
// x16 != null (and p > 0): the result again as bf16 with row stride ld16; p == 0 launches nothing and writes nothing
int vae_dropout_apply_ex(float* x, int64_t ld, int64_t B, int64_t W, float p, uint64_t seed, uint32_t step, uint32_t site,
                         uint16_t* x16, int64_t ld16, void* stream) {
  VAE_REQUIRE(!x16 || ld16 >= W, "vae_dropout_apply: bf16 leading dimension too small");
  VAE_REQUIRE(x && B > 0 && W > 0 && W < (1 << 30) && ld >= W, "vae_dropout_apply: bad arguments");
  VAE_REQUIRE(p >= 0.f && p < 1.f, "vae_dropout_apply: p must be in [0, 1), got %g", (double)p);
  if (p == 0.f) return VAE_OK;
  int blocks = (int)((B * W + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ld, B, (int)W,
                     vae_make_dropout(p, seed, step), site, x16, ld16);
  VAE_CHECK_LAUNCH("vae_dropout_apply");
  return VAE_OK;
}

// out[c] = sum over blocks of part[b * stride + c], in block order: the column sums the unfused backward's row kernels
// leave per workgroup (gelu_ln_bwd / reparam_kl_bwd with `part`), every entry of the list in ONE launch
namespace {
// A workgroup owns 64 columns of one entry; its four waves each add a quarter of the partial rows (16 loads in flight:
// the walk is one load latency per batch) and wave 0 adds the four quarters in order.
__global__ __launch_bounds__(256) void colpart_reduce_kernel(VaeColPartList z) {
  __shared__ float quarter[4][WAVE];
  int e = 0;
#pragma unroll
  for (int i = 1; i < VAE_COLPART_MAX; ++i) e += (i < z.count && (int)blockIdx.x >= z.block0[i]) ? 1 : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - z.block0[e]) * WAVE + lane;
  const bool ok = c < z.width[e];
  const float* p = z.part[e] + (ok ? c : 0);
  const int64_t st = z.stride[e];
  const int per = (z.blocks[e] + 3) / 4;
  const int b0 = wave * per, b1 = min(z.blocks[e], b0 + per);
  float s = 0.f;
  int b = b0;
  for (; b + 16 <= b1; b += 16) {
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = p[(b + j) * st];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
  }
  for (; b < b1; ++b) s += p[b * st];
  quarter[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && ok) z.out[e][c] = ((quarter[0][lane] + quarter[1][lane]) + quarter[2][lane]) + quarter[3][lane];
}
}  // namespace
int vae_colpart_reduce(VaeColPartList& z, void* stream) {
  if (z.count <= 0) return VAE_OK;
  VAE_REQUIRE(z.count <= VAE_COLPART_MAX, "vae_colpart_reduce: too many entries");
  int blocks = 0;
  for (int i = 0; i < z.count; ++i) { z.block0[i] = blocks; blocks += (z.width[i] + WAVE - 1) / WAVE; }
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z);
  VAE_CHECK_LAUNCH("vae_colpart_reduce");
  return VAE_OK;
}

extern "C" int vae_add_inplace(const float* src, float* dst, int64_t n, void* stream) {
  VAE_REQUIRE(src && dst && n > 0, "vae_add_inplace: bad arguments");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  VAE_CHECK_LAUNCH("vae_add_inplace");
  return VAE_OK;
}
