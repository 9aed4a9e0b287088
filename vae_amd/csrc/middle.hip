// The "middle" of the single-hidden-layer training step (gfx950): everything between the two
// gene-sized GEMM groups, fused into one launch per direction plus one launch for the small
// parameter gradients.
//
//   forward  (Encoder.forward modules.py:262-270, FCLayers :111-114, DecoderSCVI.forward :394-399):
//     split-K slabs of the first encoder Linear -> +bias -> GELU -> LayerNorm (z and library halves)
//     -> [mean; var] heads -> softplus, reparameterised sample, Gaussian KL (task.py:188-198)
//     -> first decoder Linear (latent -> hidden) -> GELU -> LayerNorm.
//   backward: the exact reverse, from the split-K slabs of the decoder-head dgrad down to the
//     gradient of the encoder pre-activations (the operand of the gene-sized wgrad GEMM).
//   small_wgrad_batch: every parameter gradient that is a sum over cells of per-cell quantities
//     (weight gradients of the three small Linears as outer products, all bias / LayerNorm
//     gradients as column sums), one launch, fixed summation order, no float atomics.
//
// One 512-thread workgroup owns one cell: the hidden vectors (<= 1024 wide) live in registers,
// element e of a vector belongs to thread e % 512.  Per-cell quantities needed for cross-cell
// sums are written as rows and reduced by small_wgrad_batch.
#include "middle.h"
#include "philox.h"
#include "common.h"
#include "internal.h"

#include <stdlib.h>

#ifdef VAE_PROBE
__device__ unsigned long long vae_probe[4096 * 16];
#define STAMP(i) do { if (threadIdx.x == 0) vae_probe[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

constexpr int T = 512;   // threads per workgroup (the kernels are latency-bound: more lanes, fewer rounds per phase)
constexpr int NWV = T / 64;
static_assert(NWV == VAE_MID_WAVES, "step.hip sizes the partial maxima of the fused middle as B * VAE_MID_WAVES");
constexpr int ME = 2;    // elements per thread: hidden widths up to 1024

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

// sum_{i<n} a[i*sa] * w[i*sw] with the global loads of w issued U at a time (a is LDS / cached).
template <int U>
__device__ __forceinline__ float dot_batched(const float* a, int sa, const float* __restrict__ w, int64_t sw, int n) {
  float acc = 0.f;
  for (int i = 0; i < n; i += U) {  // the last batch is masked, not serialised
    float wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = (i + u < n) ? w[(int64_t)(i + u) * sw] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fmaf((i + u < n) ? a[(i + u) * sa] : 0.f, wv[u], acc);
  }
  return acc;
}

// Sums the nslab split-K partials of NE elements per thread (element e = t + T*i < W) in slab
// order, with up to 48 loads in flight per thread: a dependent round trip costs ~2 us at the low
// occupancy of these kernels, so the number of rounds is what matters.
template <int NE>
__device__ __forceinline__ void slab_sum_ne(const float* __restrict__ base, int nslab, int64_t stride, int W, float (&v)[ME]) {
  constexpr int U = 48 / NE;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < ME; ++i) v[i] = 0.f;
  for (int s = 0; s < nslab; s += U) {
    float w[NE][U];
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
      for (int u = 0; u < U; ++u)
        w[i][u] = (t + T * i < W && s + u < nslab) ? base[(int64_t)(s + u) * stride + t + T * i] : 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
      for (int u = 0; u < U; ++u) v[i] += w[i][u];
  }
}
__device__ __forceinline__ void slab_sum(const float* __restrict__ base, int nslab, int64_t stride, int W, float (&v)[ME]) {
  static_assert(ME == 2, "slab_sum dispatches on ME == 2");
  if (W <= T) slab_sum_ne<1>(base, nslab, stride, W, v);
  else slab_sum_ne<2>(base, nslab, stride, W, v);
}

// wave_sum_dpp / wave_max_dpp: common.h
// Sums N values over the workgroup's NWV waves (wave order); red holds >= NWV*N floats.  Every thread gets all sums.
template <int N>
__device__ __forceinline__ void block_sums(float (&v)[N], float* red) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum_dpp(v[i]);
  __syncthreads();
  if (l == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[i * NWV + w] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t = red[i * NWV];
#pragma unroll
    for (int j = 1; j < NWV; ++j) t += red[i * NWV + j];
    v[i] = t;
  }
}

// LayerNorm statistics of g[] over two adjacent segments [0,W0) and [W0,W0+W1) (W1 may be 0).
__device__ __forceinline__ void ln_stats2(const float (&g)[ME], int W0, int W1, float* red, float (&mu)[2],
                                          float (&rs)[2]) {
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = threadIdx.x + T * i;
    if (e < W0) s[0] += g[i];
    else if (e < W0 + W1) s[1] += g[i];
  }
  block_sums<2>(s, red);
  mu[0] = s[0] / (float)W0;
  mu[1] = W1 > 0 ? s[1] / (float)W1 : 0.f;
  float q[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = threadIdx.x + T * i;
    if (e < W0) { const float d = g[i] - mu[0]; q[0] += d * d; }
    else if (e < W0 + W1) { const float d = g[i] - mu[1]; q[1] += d * d; }
  }
  block_sums<2>(q, red);
  rs[0] = 1.f / sqrtf(q[0] / (float)W0 + 1e-5f);
  rs[1] = W1 > 0 ? 1.f / sqrtf(q[1] / (float)W1 + 1e-5f) : 0.f;
}

struct MidFwd {
  const float* slabs; int nslab; int64_t slab_stride;   // [nslab][B][H+Hl] partial pre-activations
  const float* enc_b0;
  // count-operand form (nullable pair): the slabs hold x W^T of the RAW counts; pre = slabs / x_mean[b] - wsum[e] + bias[e]
  const float* x_mean; const float* wsum;
  const float *gz, *bz, *gl, *bl;                        // LayerNorm affine, z / library encoder
  const float *hz_w, *hz_b, *hl_w, *hl_b;               // heads: (L+1,H),(L+1) and (2,Hl),(2)
  const float *eps_z, *eps_l;
  const float *d_w, *d_b, *gd, *bd;                      // decoder layer 0: (Hd,L), (Hd), LayerNorm affine
  float l_mean, l_var;
  int B, H, Hl, L, Hd;
  float *pre_enc0, *h_enc0;                              // (B, H+Hl)
  float *ln_mz, *ln_rz, *ln_ml, *ln_rl;
  float *heads_z, *heads_l, *qz_v, *z, *kl_z, *ql_v, *library, *kl_l;
  float *dec_pre, *dec_h, *ln_md, *ln_rd;
  VaeDropout drop;                                       // nn.Dropout after each LayerNorm (off: scale == 0)
  float* absmax_px;                                      // nullable: B * NWV per-wave maxima of |dec_h| (f16x2 GEMMs)
  uint16_t* px16;                                        // nullable: dec_h again as bf16 (a bf16-storage step's operand image)
};

// softplus variance, sample and KL for one encoder, executed by ONE wave (lane j = latent j).
// hd[0:L] = q_m, hd[L] = variance pre-activation (LDS).  Mirrors reparam_kl_fwd_kernel (rowops.hip).
__device__ __forceinline__ void reparam_wave(const float* hd, int L, float eps_lane, float pm, float pv, int lane,
                                             float* z_row, float* z_lds, float* qv_out, float* kl_out) {
  const float v = softplus20(hd[L]) + 1e-4f;
  const float sq = sqrtf(v), sp = sqrtf(pv);
  const float r0 = sq / sp;
  const float r = r0 * r0;
  const float lr = logf(r);
  float acc = 0.f;
  if (lane < L) {
    const float m = hd[lane];
    const float t = (m - pm) / sp;
    acc = 0.5f * (r + t * t - 1.f - lr);
    const float zz = m + eps_lane * sq;
    z_row[lane] = zz;
    if (z_lds) z_lds[lane] = zz;
  }
  acc = wave_sum_dpp(acc);
  if (lane == 0) {
    *qv_out = v;
    *kl_out = acc;
  }
}

// Copies the first decoder weight (Hd, L) into LDS with an odd row stride LP (coalesced loads; the
// per-row reads that follow are conflict-free).  Completed by the caller's next __syncthreads().
__device__ __forceinline__ void stage_dec_w(const float* __restrict__ w, int Hd, int L, int LP, float* ws) {
  const int total = Hd * L;
  for (int i0 = 0; i0 < total; i0 += T * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + threadIdx.x + T * u;
      v[u] = i < total ? w[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + threadIdx.x + T * u;
      if (i < total) ws[(i / L) * LP + (i % L)] = v[u];
    }
  }
}

// The same image filled by LDS-DMA (global_load_lds_dword: lane i of a wave-instruction copies 4 bytes from its OWN source
// address to LDS dword base + i): no staging registers and, above all, no wait -- stage_dec_w's load -> wait -> LDS store
// rounds were two of the three or four dependent global round trips that made up half of a cell's 16 us
// (tools/micro/mid_probe.hip: 9.2 us before the first LayerNorm, 8.2 us with ONE slab).  The destination is contiguous, so
// the odd stride is realised on the SOURCE side: LDS dword d = n * LP + j takes w[n * L + j] (pad columns j >= L and the
// dwords past Hd * LP, up to the next multiple of 64, take w[0]; nobody reads them).  The copies complete with the
// wave's vmcnt -- the slab sums that follow wait for everything issued before them -- and the workgroup's barriers before
// the first use publish them.  `ws` needs (Hd * LP + 63) / 64 * 64 floats.
typedef __attribute__((address_space(3))) void* mid_lds_ptr;
__device__ __forceinline__ void stage_dec_w_dma(const float* __restrict__ w, int Hd, int L, int LP, float* ws) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dwords = Hd * LP;
  const float invLP = 1.f / (float)LP;
  const uint32_t base = (uint32_t)(uintptr_t)(mid_lds_ptr)ws;
  for (int p = wave; p * 64 < dwords; p += NWV) {
    const int d = p * 64 + lane;
    int n = (int)((float)d * invLP);       // d / LP for d < 2^20: the float quotient is off by at most one
    n += (n + 1) * LP <= d ? 1 : 0;
    n -= n * LP > d ? 1 : 0;
    const int j = d - n * LP;
    const float* src = (j < L && n < Hd) ? w + n * L + j : w;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (uint32_t)p * 256u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
  }
}
__host__ __device__ inline size_t dec_w_lds_bytes(int Hd, int L) { return (size_t)((Hd * (L | 1) + 63) / 64 * 64) * 4; }

__global__ __launch_bounds__(T) void mid_fwd_kernel(MidFwd a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // decoder weight, Hd x LP
  __shared__ float hs[ME * T];   // LayerNorm outputs of both encoders (H+Hl)
  __shared__ float hd[72];       // head outputs: [0:L] q_m z, [L] var z, [L+1] q_m l, [L+2] var l
  __shared__ float zs[64];
  __shared__ float red[2 * NWV];
  const int t = threadIdx.x, b = blockIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int H = a.H, Hl = a.Hl, HE = a.H + a.Hl, L = a.L, Hd = a.Hd;
  const int LP = L | 1;

  STAMP(0);
  // parameters this thread needs later: requested now, so their latency hides behind the slab sums
  float eb[ME], gam[ME], bet[ME], db[ME], dgam[ME], dbet[ME];
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    const bool zh = e < H;
    eb[i] = e < HE ? a.enc_b0[e] - (a.wsum ? a.wsum[e] : 0.f) : 0.f;
    gam[i] = e < HE ? (zh ? a.gz[e] : a.gl[e - H]) : 0.f;
    bet[i] = e < HE ? (zh ? a.bz[e] : a.bl[e - H]) : 0.f;
    db[i] = e < Hd ? a.d_b[e] : 0.f;
    dgam[i] = e < Hd ? a.gd[e] : 0.f;
    dbet[i] = e < Hd ? a.bd[e] : 0.f;
  }
  stage_dec_w_dma(a.d_w, Hd, L, LP, dyn);
  // head weights of this wave's outputs o = wave + NWV r (lanes along k: coalesced row reads), k < 256
  constexpr int RH = (36 + NWV - 1) / NWV;  // outputs per wave per pass: one pass covers L + 3 <= 36
  float hw[RH][4];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int o = wave + NWV * r;
    const bool zo = o <= L;
    const float* w = zo ? a.hz_w + (int64_t)o * H : a.hl_w + (int64_t)(o - L - 1) * Hl;
    const int n = o < L + 3 ? (zo ? H : Hl) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) hw[r][i] = lane + 64 * i < n ? w[lane + 64 * i] : 0.f;
  }
  float hbias[RH];   // bias of this wave's head outputs (first pass), requested with everything else
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int o = wave + NWV * r;
    hbias[r] = o < L + 3 ? (o <= L ? a.hz_b[o] : a.hl_b[o - L - 1]) : 0.f;
  }
  float epsz = 0.f, epsl = 0.f;
  if (wave == 0 && lane < L) epsz = a.eps_z[(int64_t)b * L + lane];
  if (wave == 1 && lane == 0) epsl = a.eps_l[b];

  // 1. sum the split-K slabs, add the bias: pre-activation of both first layers
  float pre[ME], g[ME];
  const float inv_m = a.x_mean ? 1.f / a.x_mean[b] : 1.f;
  slab_sum(a.slabs + (int64_t)b * HE, a.nslab, a.slab_stride, HE, pre);
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    if (e < HE) {
      pre[i] = a.x_mean ? fmaf(pre[i], inv_m, eb[i]) : pre[i] + eb[i];
      a.pre_enc0[(int64_t)b * HE + e] = pre[i];
    }
    g[i] = e < HE ? gelu_f(pre[i]) : 0.f;
  }
  STAMP(1);
  // 2. LayerNorm of the two halves
  float mu[2], rs[2];
  ln_stats2(g, H, Hl, red, mu, rs);
  STAMP(2);
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    if (e < HE) {
      const bool zh = e < H;
      float h = (g[i] - (zh ? mu[0] : mu[1])) * (zh ? rs[0] : rs[1]) * gam[i] + bet[i];
      if (zh && a.drop.on()) h *= a.drop.mult(VAE_SITE_ENC_Z(0), b, H, e);   // the library encoder has no dropout
      hs[e] = h;
      a.h_enc0[(int64_t)b * HE + e] = h;
    }
  }
  if (t == 0) {
    a.ln_mz[b] = mu[0]; a.ln_rz[b] = rs[0];
    a.ln_ml[b] = mu[1]; a.ln_rl[b] = rs[1];
  }
  __syncthreads();
  STAMP(3);
  // 3. heads: output o <= L -> z heads over H, o = L+1, L+2 -> library heads over Hl.  A wave owns
  //    RH outputs at a time; its lanes run along k, so every weight load is a coalesced row read.
  {
    const int nmax = max(H, Hl);
    for (int base = wave; base < L + 3; base += NWV * RH) {
      float acc[RH];
#pragma unroll
      for (int r = 0; r < RH; ++r) acc[r] = 0.f;
      for (int c = 0; c < nmax; c += 256) {
        if (base != wave || c != 0) {  // beyond the prefetched block: load now
#pragma unroll
          for (int r = 0; r < RH; ++r) {
            const int o = base + NWV * r;
            const bool zo = o <= L;
            const float* w = zo ? a.hz_w + (int64_t)o * H : a.hl_w + (int64_t)(o - L - 1) * Hl;
            const int n = o < L + 3 ? (zo ? H : Hl) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) hw[r][i] = c + lane + 64 * i < n ? w[c + lane + 64 * i] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < RH; ++r) {
          const int o = base + NWV * r;
          const bool zo = o <= L;
          const int n = o < L + 3 ? (zo ? H : Hl) : 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = c + lane + 64 * i;
            if (k < n) acc[r] = fmaf(hs[(zo ? 0 : H) + k], hw[r][i], acc[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        const int o = base + NWV * r;
        const float v = wave_sum_dpp(acc[r]);
        if (lane == 0 && o < L + 3) hd[o] = v + (base == wave ? hbias[r] : (o <= L ? a.hz_b[o] : a.hl_b[o - L - 1]));
      }
    }
  }
  __syncthreads();
  STAMP(4);
  // 4. reparameterised samples + KL: wave 0 -> z, wave 1 -> library
  if (wave == 0) {
    if (lane <= L) a.heads_z[(int64_t)b * (L + 1) + lane] = hd[lane];
    reparam_wave(hd, L, epsz, 0.f, 1.f, lane, a.z + (int64_t)b * L, zs, a.qz_v + b, a.kl_z + b);
  } else if (wave == 1) {
    if (lane < 2) a.heads_l[(int64_t)b * 2 + lane] = hd[L + 1 + lane];
    reparam_wave(hd + L + 1, 1, epsl, a.l_mean, a.l_var, lane, a.library + b, nullptr, a.ql_v + b, a.kl_l + b);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the decoder weight (long landed)
  __syncthreads();
  STAMP(5);
  // 5. first decoder layer: Linear(L -> Hd) + GELU + LayerNorm, weight rows from LDS
  float gd[ME];
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int n = t + T * i;
    float v = 0.f;
    if (n < Hd) {
      const float* w = dyn + n * LP;
      for (int j = 0; j < L; ++j) v = fmaf(zs[j], w[j], v);
      v += db[i];
      a.dec_pre[(int64_t)b * Hd + n] = v;
    }
    gd[i] = n < Hd ? gelu_f(v) : 0.f;
  }
  STAMP(6);
  ln_stats2(gd, Hd, 0, red, mu, rs);
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int n = t + T * i;
    if (n < Hd) {
      float h = (gd[i] - mu[0]) * rs[0] * dgam[i] + dbet[i];
      if (a.drop.on()) h *= a.drop.mult(VAE_SITE_DEC(0), b, Hd, n);
      a.dec_h[(int64_t)b * Hd + n] = h;
      if (a.px16) a.px16[(int64_t)b * Hd + n] = __builtin_bit_cast(uint16_t, (__bf16)h);
      am = fmaxf(am, fabsf(h));
    }
  }
  if (a.absmax_px) vae_wave_absmax_store(a.absmax_px + b * NWV + wave, am);   // this wave's maximum of |dec_h|
  if (t == 0) {
    a.ln_md[b] = mu[0];
    a.ln_rd[b] = rs[0];
  }
  STAMP(7);
}

struct MidBwd {
  const float* slabs; int nslab; int64_t slab_stride;   // [nslab][B][Hd] partial d(loss)/d(dec_h)
  const float *dec_pre, *ln_md, *ln_rd, *gd;
  const float* d_w;                                      // (Hd, L)
  const float* dz_extra;                                 // (B, L) or null: gradient from the cell heads
  const float *heads_z, *heads_l, *eps_z, *eps_l, *qz_v, *ql_v, *dlibrary;
  float g_kl, l_mean, l_var;
  const float *hz_w, *hl_w;
  const float *pre_enc0, *ln_mz, *ln_rz, *ln_ml, *ln_rl, *gz, *gl;
  int B, H, Hl, L, Hd;
  float *dbet_d, *dgam_d, *dpre_d;                       // (B, Hd) rows
  float *dq_z, *dq_l;                                    // (B, L+1), (B, 2)
  float *dbet_e, *dgam_e, *dpre_e;                       // (B, H+Hl) rows
  VaeDropout drop;
  // count-operand form (nullable pair): dpre_s = dpre_e / x_mean[b] is the A operand of the weight gradient over the raw
  // counts; the maxima are then taken of dpre_s
  const float* x_mean; float* dpre_s;
  float* absmax_dpre;                                    // nullable: B * NWV per-wave maxima of |dpre_e|
  uint16_t* dpre16;                                      // nullable: dpre_e again as bf16 (a bf16-storage step's operand image)
};

// Backward of reparam_wave, by one wave.  Mirrors reparam_kl_bwd_kernel (rowops.hip).
__device__ __forceinline__ void reparam_bwd_wave(const float* dz_lds, const float* hrow, const float* eps_row, float v,
                                                 float gk, int L, float pm, float pv, int lane, float* dq_row,
                                                 float* dq_lds) {
  const float sq = sqrtf(v);
  float dv = 0.f;
  if (lane < L) {
    const float d = dz_lds[lane];
    const float o = d + gk * (hrow[lane] - pm) / pv;
    dq_row[lane] = o;
    dq_lds[lane] = o;
    dv = d * eps_row[lane];
  }
  dv = wave_sum_dpp(dv);
  dv = dv * 0.5f / sq + gk * (float)L * 0.5f * (1.f / pv - 1.f / v);
  const float o = dv * softplus20_grad(hrow[L]);
  if (lane == 0) {
    dq_row[L] = o;
    dq_lds[L] = o;
  }
}

// The same with the wave's inputs already in registers (requested at the top of the kernel, so that no global round trip
// sits between the barriers of the chain): hr = lane's head output (lane L: the variance pre-activation), ep = lane's eps.
__device__ __forceinline__ void reparam_bwd_regs(const float* dz_lds, float hr, float ep, float v, float gk, int L, float pm,
                                                 float pv, int lane, float* dq_row, float* dq_lds) {
  const float sq = sqrtf(v);
  float dv = 0.f;
  if (lane < L) {
    const float d = dz_lds[lane];
    const float o = d + gk * (hr - pm) / pv;
    dq_row[lane] = o;
    dq_lds[lane] = o;
    dv = d * ep;
  }
  dv = wave_sum_dpp(dv);
  dv = dv * 0.5f / sq + gk * (float)L * 0.5f * (1.f / pv - 1.f / v);
  const float o = dv * softplus20_grad(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(hr), L)));
  if (lane == 0) {
    dq_row[L] = o;
    dq_lds[L] = o;
  }
}

__global__ __launch_bounds__(T) void mid_bwd_kernel(MidBwd a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // decoder weight, Hd x LP
  __shared__ float ds[ME * T];   // d pre-activation of the decoder layer
  __shared__ float dzs[64];
  __shared__ float dq[72];       // [0:L+1] z heads, [L+1:L+3] library heads
  __shared__ float red[4 * NWV];
  const int t = threadIdx.x, b = blockIdx.x;
  const int H = a.H, Hl = a.Hl, HE = a.H + a.Hl, L = a.L, Hd = a.Hd;

  const int LP = L | 1;
  stage_dec_w_dma(a.d_w, Hd, L, LP, dyn);
  // 1. d(dec_h): sum of the dgrad slabs; LayerNorm + GELU backward of the decoder layer
  const float mud = a.ln_md[b], rsd = a.ln_rd[b];
  const float mu0 = a.ln_mz[b], rs0 = a.ln_rz[b], mu1 = a.ln_ml[b], rs1 = a.ln_rl[b];
  float pd[ME], gdv[ME], pe[ME], gev[ME];  // requested now, used after the slab sums
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    pd[i] = e < Hd ? a.dec_pre[(int64_t)b * Hd + e] : 0.f;
    gdv[i] = e < Hd ? a.gd[e] : 0.f;
    pe[i] = e < HE ? a.pre_enc0[(int64_t)b * HE + e] : 0.f;
    gev[i] = e < HE ? (e < H ? a.gz[e] : a.gl[e - H]) : 0.f;
  }
  // head-weight column of this thread's first element (e = t): all of its L+1 (or 2) rows, requested now
  constexpr int JP = 32;
  float hwc[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    float v = 0.f;
    if (t < H && j <= L) v = a.hz_w[(int64_t)j * H + t];
    hwc[j] = v;
  }
  float hlc[ME][2];  // library-head weight columns of every library element of this thread
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    const bool lh = e >= H && e < HE;
    hlc[i][0] = lh ? a.hl_w[e - H] : 0.f;
    hlc[i][1] = lh ? a.hl_w[Hl + e - H] : 0.f;
  }
  // what the later stages of the chain read per cell, requested now: the reparameterisation inputs of waves 0 / 1, the
  // cell-head gradient of this thread's latent, 1 / mean (each was a global round trip between two barriers)
  float r_h = 0.f, r_e = 0.f, r_v = 1.f, r_dlib = 0.f;
  if ((t >> 6) == 0) {
    if (t <= L) r_h = a.heads_z[(int64_t)b * (L + 1) + t];
    if (t < L) r_e = a.eps_z[(int64_t)b * L + t];
    r_v = a.qz_v[b];
  } else if ((t >> 6) == 1) {
    const int lane = t & 63;
    if (lane < 2) r_h = a.heads_l[(int64_t)b * 2 + lane];
    if (lane < 1) r_e = a.eps_l[b];
    r_v = a.ql_v[b];
    r_dlib = a.dlibrary[b];
  }
  const float dze0 = (a.dz_extra && (t & 7) == 0 && (t >> 3) < L) ? a.dz_extra[(int64_t)b * L + (t >> 3)] : 0.f;
  const float x_mean_b = a.x_mean ? a.x_mean[b] : 1.f;
  float dx[ME], xh[ME], pv[ME], dyv[ME];
  float s[2] = {0.f, 0.f};
  slab_sum(a.slabs + (int64_t)b * Hd, a.nslab, a.slab_stride, Hd, dyv);
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int n = t + T * i;
    dx[i] = xh[i] = pv[i] = 0.f;
    if (n < Hd) {
      float dy = dyv[i];
      if (a.drop.on()) dy *= a.drop.mult(VAE_SITE_DEC(0), b, Hd, n);
      pv[i] = pd[i];
      xh[i] = (gelu_f(pv[i]) - mud) * rsd;
      dx[i] = dy * gdv[i];
      a.dbet_d[(int64_t)b * Hd + n] = dy;
      a.dgam_d[(int64_t)b * Hd + n] = dy * xh[i];
      s[0] += dx[i];
      s[1] += dx[i] * xh[i];
    }
  }
  block_sums<2>(s, red);
  s[0] /= (float)Hd;
  s[1] /= (float)Hd;
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int n = t + T * i;
    if (n < Hd) {
      const float dg = rsd * (dx[i] - s[0] - xh[i] * s[1]);
      const float v = dg * gelu_grad(pv[i]);
      ds[n] = v;
      a.dpre_d[(int64_t)b * Hd + n] = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the decoder weight (long landed; also the stores above)
  __syncthreads();
  // 2. dz[j] = sum_n dpre_d[n] W[n][j] (+ cell-head gradient); 8 threads per output
  {
    const int part = t & 7;
    for (int j = t >> 3; j < L; j += T / 8) {
      float acc = 0.f;
      for (int n = part; n < Hd; n += 8) acc = fmaf(ds[n], dyn[n * LP + j], acc);
      acc += __shfl_xor(acc, 1, WAVE);
      acc += __shfl_xor(acc, 2, WAVE);
      acc += __shfl_xor(acc, 4, WAVE);
      if (part == 0) dzs[j] = acc + (a.dz_extra ? (j == (t >> 3) ? dze0 : a.dz_extra[(int64_t)b * L + j]) : 0.f);
    }
  }
  __syncthreads();
  // 3. reparameterisation / KL backward: wave 0 -> z heads, wave 1 -> library heads
  {
    const int lane = t & 63, wave = t >> 6;
    if (wave == 0) {
      reparam_bwd_regs(dzs, r_h, r_e, r_v, a.g_kl, L, 0.f, 1.f, lane, a.dq_z + (int64_t)b * (L + 1), dq);
    } else if (wave == 1) {
      __shared__ float dlib[1];
      if (lane == 0) dlib[0] = r_dlib;
      reparam_bwd_regs(dlib, r_h, r_e, r_v, a.g_kl, 1, a.l_mean, a.l_var, lane, a.dq_l + (int64_t)b * 2, dq + L + 1);
    }
  }
  __syncthreads();
  // 4. gradient at the LayerNorm outputs of both encoders, then LayerNorm + GELU backward
  float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    dx[i] = xh[i] = pv[i] = 0.f;
    if (e < HE) {
      const bool zh = e < H;
      float dy = 0.f;
      if (!zh) {
        dy = fmaf(dq[L + 2], hlc[i][1], dq[L + 1] * hlc[i][0]);
      } else if (i == 0 && L + 1 <= JP) {  // prefetched column
#pragma unroll
        for (int j = 0; j < JP; ++j) dy = fmaf(j <= L ? dq[j] : 0.f, hwc[j], dy);
      } else {
        dy = dot_batched<16>(dq, 1, a.hz_w + e, H, L + 1);
      }
      if (zh && a.drop.on()) dy *= a.drop.mult(VAE_SITE_ENC_Z(0), b, H, e);
      pv[i] = pe[i];
      xh[i] = (gelu_f(pv[i]) - (zh ? mu0 : mu1)) * (zh ? rs0 : rs1);
      dx[i] = dy * gev[i];
      a.dbet_e[(int64_t)b * HE + e] = dy;
      a.dgam_e[(int64_t)b * HE + e] = dy * xh[i];
      q4[zh ? 0 : 2] += dx[i];
      q4[zh ? 1 : 3] += dx[i] * xh[i];
    }
  }
  block_sums<4>(q4, red);
  float am = 0.f;
  const float inv_m = a.x_mean ? 1.f / x_mean_b : 1.f;
  q4[0] /= (float)H; q4[1] /= (float)H;
  q4[2] /= (float)Hl; q4[3] /= (float)Hl;
#pragma unroll
  for (int i = 0; i < ME; ++i) {
    const int e = t + T * i;
    if (e < HE) {
      const bool zh = e < H;
      const float dg = (zh ? rs0 : rs1) * (dx[i] - (zh ? q4[0] : q4[2]) - xh[i] * (zh ? q4[1] : q4[3]));
      const float o = dg * gelu_grad(pv[i]);
      a.dpre_e[(int64_t)b * HE + e] = o;
      if (a.dpre16) a.dpre16[(int64_t)b * HE + e] = __builtin_bit_cast(uint16_t, (__bf16)o);
      if (a.dpre_s) {
        const float os = o * inv_m;
        a.dpre_s[(int64_t)b * HE + e] = os;
        am = fmaxf(am, fabsf(os));
      } else
        am = fmaxf(am, fabsf(o));
    }
  }
  if (a.absmax_dpre) vae_wave_absmax_store(a.absmax_dpre + b * NWV + (t >> 6), am);
}

// ---------------------------------------------------------------------------------------
// Wave-per-cell form of the two kernels above.  The workgroup-per-cell kernels are one latency chain per cell -- a dozen
// workgroup barriers and three dependent global round trips, ~20 us whatever the batch -- and a CU holds three of them: at
// B = 4096 the pair costs 260 us for a few MB and a few MFLOP per cell.  Here ONE WAVE owns a cell: lane l holds elements
// l, l + 64, ... of every hidden vector, all reductions are wave shuffles (no barrier after the set-up), and the
// weights every cell needs -- the first decoder Linear (Hd, L) and the latent heads (L + 1, H), (2, Hl) -- are staged
// in LDS ONCE per workgroup, whose WPB waves then walk over cells b = first, first + stride, ...  Results are the same
// expressions in a different summation order (lane-strided partial sums, then a butterfly).
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wave_sums(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum_dpp(v[i]);
}
__device__ __forceinline__ float lane_bcast(float v, int src) {   // src wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// LDS image shared by both directions: Wd[Hd][LP] (LP = L | 1: a lane's row reads are conflict-free), Wz[L + 1][H],
// Wl[2][Hl]
struct MidLds {
  float *wd, *wz, *wl;
  int LP;
};
__device__ __forceinline__ MidLds mid_stage_weights(float* dyn, const float* __restrict__ d_w, const float* __restrict__ hz_w,
                                                    const float* __restrict__ hl_w, int H, int Hl, int L, int Hd, int nthreads) {
  MidLds m;
  m.LP = L | 1;
  m.wd = dyn;
  m.wz = dyn + ((Hd * m.LP + 3) & ~3);
  m.wl = m.wz + (L + 1) * H;
  const int t = threadIdx.x;
  // every load of a batch is in flight before the first LDS store (a dependent round trip costs ~2 us here): the head
  // weights as 16-byte copies (H % 4 == 0, 16-byte aligned: the host checks), 8 per thread and batch; the decoder weight
  // element-wise into its odd-stride image (d_w is (Hd, L) with L odd or even: no vector alignment), 16 per batch
  const int nz4 = ((L + 1) * H) >> 2;
  for (int i0 = 0; i0 < nz4; i0 += nthreads * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + t + nthreads * u;
      v[u] = i < nz4 ? reinterpret_cast<const float4*>(hz_w)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + t + nthreads * u;
      if (i < nz4) reinterpret_cast<float4*>(m.wz)[i] = v[u];
    }
  }
  const int total = Hd * L;
  const float invL = 1.f / (float)L;
  for (int i0 = 0; i0 < total; i0 += nthreads * 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + t + nthreads * u;
      v[u] = i < total ? d_w[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + t + nthreads * u;
      int r = (int)((float)i * invL);        // i / L for i < 2^20: the float quotient is off by at most one
      r += (r + 1) * L <= i ? 1 : 0;
      r -= r * L > i ? 1 : 0;
      if (i < total) m.wd[r * m.LP + (i - r * L)] = v[u];
    }
  }
  for (int i = t; i < 2 * Hl; i += nthreads) m.wl[i] = hl_w[i];
  return m;
}

// Sum of the nslab split-K partials of this lane's NE elements of row `base`, slab order, 16 slabs in flight.
template <int NE, int WPB>
__device__ __forceinline__ void slab_sum_wave(const float* __restrict__ base, int nslab, int64_t stride, int W, int lane,
                                              float (&v)[NE]) {
  constexpr int U = (WPB <= 4 ? 128 : 64) / NE;   // loads in flight per lane: workgroups of <= 4 waves have 256 registers
#pragma unroll
  for (int i = 0; i < NE; ++i) v[i] = 0.f;
  for (int s = 0; s < nslab; s += U) {
    float w[NE][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NE; ++i)
        w[i][u] = (lane + 64 * i < W && s + u < nslab) ? base[(int64_t)(s + u) * stride + lane + 64 * i] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] += w[i][u];
  }
}

template <int NEE, int NED, int WPB>
__global__ __launch_bounds__(64 * WPB) void mid_fwd_wave_kernel(MidFwd a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, Hl = a.Hl, HE = a.H + a.Hl, L = a.L, Hd = a.Hd;
  STAMP(0);
  const MidLds w = mid_stage_weights(dyn, a.d_w, a.hz_w, a.hl_w, H, Hl, L, Hd, 64 * WPB);
  const int LP = w.LP;
  // per-lane parameters, kept for every cell of this wave
  float eb[NEE], gam[NEE], bet[NEE], db[NED], dgam[NED], dbet[NED];
#pragma unroll
  for (int i = 0; i < NEE; ++i) {
    const int e = lane + 64 * i;
    const bool zh = e < H;
    eb[i] = e < HE ? a.enc_b0[e] - (a.wsum ? a.wsum[e] : 0.f) : 0.f;
    gam[i] = e < HE ? (zh ? a.gz[e] : a.gl[e - H]) : 0.f;
    bet[i] = e < HE ? (zh ? a.bz[e] : a.bl[e - H]) : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NED; ++i) {
    const int n = lane + 64 * i;
    db[i] = n < Hd ? a.d_b[n] : 0.f;
    dgam[i] = n < Hd ? a.gd[n] : 0.f;
    dbet[i] = n < Hd ? a.bd[n] : 0.f;
  }
  // head biases: lane j <= L holds the bias of z head j; lanes 0, 1 those of the library heads
  const float hzb = lane <= L ? a.hz_b[lane] : 0.f;
  const float hlb = lane < 2 ? a.hl_b[lane] : 0.f;
  __syncthreads();   // the LDS images are complete; no barrier below
  STAMP(1);

  for (int b = blockIdx.x * WPB + wave; b < a.B; b += gridDim.x * WPB) {
    const float epsz = lane < L ? a.eps_z[(int64_t)b * L + lane] : 0.f;
    const float epsl = a.eps_l[b];
    // 1. pre-activation of both first layers, GELU
    float pre[NEE], g[NEE];
    const float inv_m = a.x_mean ? 1.f / a.x_mean[b] : 1.f;
    slab_sum_wave<NEE, WPB>(a.slabs + (int64_t)b * HE, a.nslab, a.slab_stride, HE, lane, pre);
    float st[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      if (e < HE) {
        pre[i] = a.x_mean ? fmaf(pre[i], inv_m, eb[i]) : pre[i] + eb[i];
        a.pre_enc0[(int64_t)b * HE + e] = pre[i];
      }
      g[i] = e < HE ? gelu_f(pre[i]) : 0.f;
      st[e < H ? 0 : 1] += g[i];
    }
    STAMP(2);
    // 2. LayerNorm of the two halves
    wave_sums<2>(st);
    const float mu0 = st[0] / (float)H, mu1 = st[1] / (float)Hl;
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      if (e < H) { const float d = g[i] - mu0; q[0] += d * d; }
      else if (e < HE) { const float d = g[i] - mu1; q[1] += d * d; }
    }
    wave_sums<2>(q);
    const float rs0 = 1.f / sqrtf(q[0] / (float)H + 1e-5f), rs1 = 1.f / sqrtf(q[1] / (float)Hl + 1e-5f);
    float h[NEE];
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      h[i] = 0.f;
      if (e < HE) {
        const bool zh = e < H;
        float v = (g[i] - (zh ? mu0 : mu1)) * (zh ? rs0 : rs1) * gam[i] + bet[i];
        if (zh && a.drop.on()) v *= a.drop.mult(VAE_SITE_ENC_Z(0), b, H, e);
        h[i] = v;
        a.h_enc0[(int64_t)b * HE + e] = v;
      }
    }
    if (lane == 0) {
      a.ln_mz[b] = mu0; a.ln_rz[b] = rs0;
      a.ln_ml[b] = mu1; a.ln_rl[b] = rs1;
    }
    STAMP(3);
    // 3. heads: lane j <= L ends up with z head j, lanes 0 / 1 of `hl` with the library heads.  Branch-free inner loops
    //    (clamped LDS offsets, zeroed multipliers) so that the LDS reads of several outputs are in flight together.
    float hz = 0.f, hl = 0.f;
    float hzv[NEE], hlv[NEE];
    int oz[NEE], ol[NEE];
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      hzv[i] = e < H ? h[i] : 0.f;
      hlv[i] = (e >= H && e < HE) ? h[i] : 0.f;
      oz[i] = min(e, H - 1);
      ol[i] = min(max(e - H, 0), Hl - 1);
    }
    for (int o0 = 0; o0 <= L; o0 += 4) {   // four outputs at a time: independent reduction chains
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* wr = w.wz + min(o0 + u, L) * H;
#pragma unroll
        for (int i = 0; i < NEE; ++i) acc[u] = fmaf(hzv[i], wr[oz[i]], acc[u]);
      }
      wave_sums<4>(acc);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane == o0 + u) hz = acc[u];
    }
    {
      float acc[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NEE; ++i) {
        acc[0] = fmaf(hlv[i], w.wl[ol[i]], acc[0]);
        acc[1] = fmaf(hlv[i], w.wl[Hl + ol[i]], acc[1]);
      }
      wave_sums<2>(acc);
      hl = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : 0.f);
    }
    hz += hzb;
    hl += hlb;
    if (lane <= L) a.heads_z[(int64_t)b * (L + 1) + lane] = hz;
    if (lane < 2) a.heads_l[(int64_t)b * 2 + lane] = hl;
    STAMP(4);
    // 4. reparameterised samples + KL (reparam_wave above, with the heads in lanes instead of LDS)
    float zj;
    {
      const float v = softplus20(lane_bcast(hz, L)) + 1e-4f;
      const float sq = sqrtf(v);
      const float r = sq * sq;     // prior N(0, 1): (sq / 1)^2 as reparam_wave forms it
      const float lr = logf(r);
      float acc = 0.f;
      zj = 0.f;
      if (lane < L) {
        acc = 0.5f * (r + hz * hz - 1.f - lr);
        zj = hz + epsz * sq;
        a.z[(int64_t)b * L + lane] = zj;
      }
      acc = wave_sum_dpp(acc);
      if (lane == 0) { a.qz_v[b] = v; a.kl_z[b] = acc; }
    }
    {
      const float m = lane_bcast(hl, 0);
      const float v = softplus20(lane_bcast(hl, 1)) + 1e-4f;
      const float sq = sqrtf(v), sp = sqrtf(a.l_var);
      const float r0 = sq / sp, r = r0 * r0;
      const float tt = (m - a.l_mean) / sp;
      if (lane == 0) {
        a.ql_v[b] = v;
        a.kl_l[b] = 0.5f * (r + tt * tt - 1.f - logf(r));
        a.library[b] = m + epsl * sq;
      }
    }
    STAMP(5);
    // 5. first decoder layer: Linear(L -> Hd) + GELU + LayerNorm; z_j broadcast from lane j, weight rows from LDS
    float pd[NED];
    int od[NED];
#pragma unroll
    for (int i = 0; i < NED; ++i) { pd[i] = 0.f; od[i] = min(lane + 64 * i, Hd - 1) * LP; }
#pragma unroll 6
    for (int j = 0; j < L; ++j) {
      const float z1 = lane_bcast(zj, j);
#pragma unroll
      for (int i = 0; i < NED; ++i) pd[i] = fmaf(z1, w.wd[od[i] + j], pd[i]);
    }
    STAMP(6);
    float gd[NED];
    float sd[1] = {0.f};
#pragma unroll
    for (int i = 0; i < NED; ++i) {
      const int n = lane + 64 * i;
      gd[i] = 0.f;
      if (n < Hd) {
        pd[i] += db[i];
        a.dec_pre[(int64_t)b * Hd + n] = pd[i];
        gd[i] = gelu_f(pd[i]);
      }
      sd[0] += gd[i];
    }
    wave_sums<1>(sd);
    const float mud = sd[0] / (float)Hd;
    float qd[1] = {0.f};
#pragma unroll
    for (int i = 0; i < NED; ++i)
      if (lane + 64 * i < Hd) { const float d = gd[i] - mud; qd[0] += d * d; }
    wave_sums<1>(qd);
    const float rsd = 1.f / sqrtf(qd[0] / (float)Hd + 1e-5f);
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < NED; ++i) {
      const int n = lane + 64 * i;
      if (n < Hd) {
        float v = (gd[i] - mud) * rsd * dgam[i] + dbet[i];
        if (a.drop.on()) v *= a.drop.mult(VAE_SITE_DEC(0), b, Hd, n);
        a.dec_h[(int64_t)b * Hd + n] = v;
        if (a.px16) a.px16[(int64_t)b * Hd + n] = __builtin_bit_cast(uint16_t, (__bf16)v);
        am = fmaxf(am, fabsf(v));
      }
    }
    if (lane == 0) { a.ln_md[b] = mud; a.ln_rd[b] = rsd; }
    if (a.absmax_px) {   // the array has NWV entries per cell (the workgroup-per-cell kernel's waves): entry 0 = the maximum
      am = wave_max_dpp(am);
      if (lane < NWV) a.absmax_px[b * NWV + lane] = lane == 0 ? am : 0.f;
    }
    STAMP(7);
  }
}

template <int NEE, int NED, int WPB>
__global__ __launch_bounds__(64 * WPB) void mid_bwd_wave_kernel(MidBwd a) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, Hl = a.Hl, HE = a.H + a.Hl, L = a.L, Hd = a.Hd;
  const MidLds w = mid_stage_weights(dyn, a.d_w, a.hz_w, a.hl_w, H, Hl, L, Hd, 64 * WPB);
  const int LP = w.LP;
  float gdv[NED], gev[NEE];
#pragma unroll
  for (int i = 0; i < NED; ++i) gdv[i] = lane + 64 * i < Hd ? a.gd[lane + 64 * i] : 0.f;
#pragma unroll
  for (int i = 0; i < NEE; ++i) {
    const int e = lane + 64 * i;
    gev[i] = e < HE ? (e < H ? a.gz[e] : a.gl[e - H]) : 0.f;
  }
  __syncthreads();

  for (int b = blockIdx.x * WPB + wave; b < a.B; b += gridDim.x * WPB) {
    // everything this cell needs from global memory, requested before the slab sums
    const float mud = a.ln_md[b], rsd = a.ln_rd[b];
    const float mu0 = a.ln_mz[b], rs0 = a.ln_rz[b], mu1 = a.ln_ml[b], rs1 = a.ln_rl[b];
    float pdv[NED], pe[NEE];
#pragma unroll
    for (int i = 0; i < NED; ++i) pdv[i] = lane + 64 * i < Hd ? a.dec_pre[(int64_t)b * Hd + lane + 64 * i] : 0.f;
#pragma unroll
    for (int i = 0; i < NEE; ++i) pe[i] = lane + 64 * i < HE ? a.pre_enc0[(int64_t)b * HE + lane + 64 * i] : 0.f;
    const float hzr = lane <= L ? a.heads_z[(int64_t)b * (L + 1) + lane] : 0.f;
    const float epz = lane < L ? a.eps_z[(int64_t)b * L + lane] : 0.f;
    const float dzx = (a.dz_extra && lane < L) ? a.dz_extra[(int64_t)b * L + lane] : 0.f;
    const float qzv = a.qz_v[b], qlv = a.ql_v[b], dlib = a.dlibrary[b], epl = a.eps_l[b];
    const float hl0 = a.heads_l[(int64_t)b * 2], hl1 = a.heads_l[(int64_t)b * 2 + 1];
    const float x_mean_b = a.x_mean ? a.x_mean[b] : 1.f;
    // 1. d(dec_h): sum of the dgrad slabs; LayerNorm + GELU backward of the decoder layer
    float dyv[NED], dx[NED], xh[NED], ds[NED];
    slab_sum_wave<NED, WPB>(a.slabs + (int64_t)b * Hd, a.nslab, a.slab_stride, Hd, lane, dyv);
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NED; ++i) {
      const int n = lane + 64 * i;
      dx[i] = xh[i] = 0.f;
      if (n < Hd) {
        float dy = dyv[i];
        if (a.drop.on()) dy *= a.drop.mult(VAE_SITE_DEC(0), b, Hd, n);
        xh[i] = (gelu_f(pdv[i]) - mud) * rsd;
        dx[i] = dy * gdv[i];
        a.dbet_d[(int64_t)b * Hd + n] = dy;
        a.dgam_d[(int64_t)b * Hd + n] = dy * xh[i];
        s[0] += dx[i];
        s[1] += dx[i] * xh[i];
      }
    }
    wave_sums<2>(s);
    s[0] /= (float)Hd;
    s[1] /= (float)Hd;
#pragma unroll
    for (int i = 0; i < NED; ++i) {
      const int n = lane + 64 * i;
      ds[i] = 0.f;
      if (n < Hd) {
        const float dg = rsd * (dx[i] - s[0] - xh[i] * s[1]);
        ds[i] = dg * gelu_grad(pdv[i]);
        a.dpre_d[(int64_t)b * Hd + n] = ds[i];
      }
    }
    // 2. dz[j] = sum_n dpre_d[n] W[n][j] (+ cell-head gradient): lane j keeps dz[j]
    float dz = 0.f;
    int od[NED];
#pragma unroll
    for (int i = 0; i < NED; ++i) od[i] = min(lane + 64 * i, Hd - 1) * LP;   // ds[i] = 0 past Hd
    for (int j0 = 0; j0 < L; j0 += 4) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + u, L - 1);
#pragma unroll
        for (int i = 0; i < NED; ++i) acc[u] = fmaf(ds[i], w.wd[od[i] + j], acc[u]);
      }
      wave_sums<4>(acc);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane == j0 + u && j0 + u < L) dz = acc[u];
    }
    dz += dzx;
    // 3. reparameterisation / KL backward (reparam_bwd_wave above): lane j <= L keeps d head j; dl0 / dl1 the library pair
    float dqz;
    {
      const float sq = sqrtf(qzv);
      float dv = wave_sum_dpp(lane < L ? dz * epz : 0.f);
      dv = dv * 0.5f / sq + a.g_kl * (float)L * 0.5f * (1.f - 1.f / qzv);
      const float ov = dv * softplus20_grad(lane_bcast(hzr, L));
      dqz = lane < L ? dz + a.g_kl * hzr : (lane == L ? ov : 0.f);
      if (lane <= L) a.dq_z[(int64_t)b * (L + 1) + lane] = dqz;
    }
    float dl0, dl1;
    {
      const float sq = sqrtf(qlv), pv = a.l_var;
      dl0 = dlib + a.g_kl * (hl0 - a.l_mean) / pv;
      float dv = dlib * epl;
      dv = dv * 0.5f / sq + a.g_kl * 0.5f * (1.f / pv - 1.f / qlv);
      dl1 = dv * softplus20_grad(hl1);
      if (lane == 0) { a.dq_l[(int64_t)b * 2] = dl0; a.dq_l[(int64_t)b * 2 + 1] = dl1; }
    }
    // 4. gradient at the LayerNorm outputs of both encoders, then LayerNorm + GELU backward
    float dye[NEE], dyz[NEE];
    int oz[NEE];
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      const int el = min(max(e - H, 0), Hl - 1);
      dye[i] = fmaf(dl1, w.wl[Hl + el], dl0 * w.wl[el]);   // library half (used where H <= e < HE)
      dyz[i] = 0.f;
      oz[i] = min(e, H - 1);
    }
#pragma unroll 4
    for (int j = 0; j <= L; ++j) {   // z half, branch-free: every lane accumulates at a clamped offset
      const float dq1 = lane_bcast(dqz, j);
      const float* wr = w.wz + j * H;
#pragma unroll
      for (int i = 0; i < NEE; ++i) dyz[i] = fmaf(dq1, wr[oz[i]], dyz[i]);
    }
#pragma unroll
    for (int i = 0; i < NEE; ++i)
      if (lane + 64 * i < H) dye[i] = dyz[i];
    float q4[4] = {0.f, 0.f, 0.f, 0.f};
    float dxe[NEE], xe[NEE];
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      dxe[i] = xe[i] = 0.f;
      if (e < HE) {
        const bool zh = e < H;
        float dy = dye[i];
        if (zh && a.drop.on()) dy *= a.drop.mult(VAE_SITE_ENC_Z(0), b, H, e);
        xe[i] = (gelu_f(pe[i]) - (zh ? mu0 : mu1)) * (zh ? rs0 : rs1);
        dxe[i] = dy * gev[i];
        a.dbet_e[(int64_t)b * HE + e] = dy;
        a.dgam_e[(int64_t)b * HE + e] = dy * xe[i];
        q4[zh ? 0 : 2] += dxe[i];
        q4[zh ? 1 : 3] += dxe[i] * xe[i];
      }
    }
    wave_sums<4>(q4);
    q4[0] /= (float)H; q4[1] /= (float)H;
    q4[2] /= (float)Hl; q4[3] /= (float)Hl;
    float am = 0.f;
    const float inv_m = a.x_mean ? 1.f / x_mean_b : 1.f;
#pragma unroll
    for (int i = 0; i < NEE; ++i) {
      const int e = lane + 64 * i;
      if (e < HE) {
        const bool zh = e < H;
        const float dg = (zh ? rs0 : rs1) * (dxe[i] - (zh ? q4[0] : q4[2]) - xe[i] * (zh ? q4[1] : q4[3]));
        const float o = dg * gelu_grad(pe[i]);
        a.dpre_e[(int64_t)b * HE + e] = o;
        if (a.dpre16) a.dpre16[(int64_t)b * HE + e] = __builtin_bit_cast(uint16_t, (__bf16)o);
        if (a.dpre_s) {
          const float os = o * inv_m;
          a.dpre_s[(int64_t)b * HE + e] = os;
          am = fmaxf(am, fabsf(os));
        } else
          am = fmaxf(am, fabsf(o));
      }
    }
    if (a.absmax_dpre) {
      am = wave_max_dpp(am);
      if (lane < NWV) a.absmax_dpre[b * NWV + lane] = lane == 0 ? am : 0.f;
    }
  }
}

// (elements per lane of the encoder / decoder hidden vectors, waves per workgroup) of the wave-per-cell kernels for a shape;
// false: the workgroup-per-cell kernels run
struct MidWavePlan { int nee, ned, wpb, grid; size_t lds; };
bool mid_wave_plan(const vae_step_config& c, const float* hz_w, MidWavePlan* p) {
  static const bool off = getenv("VAE_MID_WORKGROUP_PER_CELL") != nullptr;   // A/B knob
  const int HE = (int)(c.H + c.Hl), Hd = (int)c.Hd, L = (int)c.L;
  // one wave per cell is one long dependent chain (~25 us at cfg2's widths): it wins once a CU has several cells to
  // overlap (B = 4096: 44 / 60 us against 131 / 128); below, the workgroup-per-cell kernels' 22 us stand
  static const int min_b = getenv("VAE_MID_WAVE_MIN_B") ? atoi(getenv("VAE_MID_WAVE_MIN_B")) : 768;
  if (off || c.B < min_b || HE > 1024 || Hd > 1024 || c.H % 4 != 0 || L + 1 > 64 || !vae_aligned16(hz_w)) return false;
  p->nee = HE <= 384 ? 6 : (HE <= 512 ? 8 : 16);
  p->ned = Hd <= 256 ? 4 : (Hd <= 512 ? 8 : 16);
  if (p->nee == 6 && p->ned != 4) p->nee = 8;           // instantiated pairs: (6, 4), (8, 4), (16, 8), (16, 16)
  if (p->nee == 8 && p->ned == 8) p->nee = 16;
  if (p->nee == 16 && p->ned == 4) p->ned = 8;
  p->lds = (size_t)(((Hd * (L | 1) + 3) & ~3) + (L + 1) * (int)c.H + 2 * (int)c.Hl) * sizeof(float);
  if (p->lds > 150 * 1024) return false;
  p->wpb = c.B <= 1024 ? 4 : 8;
  const int groups = (int)((c.B + p->wpb - 1) / p->wpb);
  const int per_cu = p->lds <= 76 * 1024 ? 2 : 1;
  p->grid = groups < 256 * per_cu ? groups : 256 * per_cu;
  return true;
}
template <typename A, typename K>
int mid_wave_launch(K kern, const MidWavePlan& p, const A& a, hipStream_t st) {
  if (p.lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
    return vae_set_error(VAE_ERR_LAUNCH, "vae_elbo_step (fused middle): cannot raise the LDS limit");
  hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(64 * p.wpb), p.lds, st, a);
  return VAE_OK;
}
#define MID_WAVE_DISPATCH(KERN, A)                                                                             \
  do {                                                                                                         \
    int rc__;                                                                                                  \
    if (pl.nee == 6 && pl.wpb == 4)       rc__ = mid_wave_launch(KERN<6, 4, 4>, pl, A, (hipStream_t)stream);   \
    else if (pl.nee == 6)                 rc__ = mid_wave_launch(KERN<6, 4, 8>, pl, A, (hipStream_t)stream);   \
    else if (pl.nee == 8 && pl.wpb == 4)  rc__ = mid_wave_launch(KERN<8, 4, 4>, pl, A, (hipStream_t)stream);   \
    else if (pl.nee == 8)                 rc__ = mid_wave_launch(KERN<8, 4, 8>, pl, A, (hipStream_t)stream);   \
    else if (pl.ned == 8 && pl.wpb == 4)  rc__ = mid_wave_launch(KERN<16, 8, 4>, pl, A, (hipStream_t)stream);  \
    else if (pl.ned == 8)                 rc__ = mid_wave_launch(KERN<16, 8, 8>, pl, A, (hipStream_t)stream);  \
    else if (pl.wpb == 4)                 rc__ = mid_wave_launch(KERN<16, 16, 4>, pl, A, (hipStream_t)stream); \
    else                                  rc__ = mid_wave_launch(KERN<16, 16, 8>, pl, A, (hipStream_t)stream); \
    if (rc__ != VAE_OK) return rc__;                                                                           \
  } while (0)

// ---------------------------------------------------------------------------------------
// out[p*sp + q*sq] = sum_b U[b][p] * V[b][q]   (U null: column sums of V), for a table of tasks.
// A workgroup (1024 threads) owns 8 p x 64 q outputs of one task: 256 cells at a time are staged in LDS (all
// the global loads of a round in flight together); thread (q = t%64, r = t/64) sums the cells
// b = r (mod 16) in ascending order, the sixteen partial sums are combined r = 0..15: deterministic.
// Large batches (nsplit > 1): the ~70 workgroups of the task table would each walk B / 256 rounds, one load latency
// after the other (60 us at B = 4096).  The cells are then cut into nsplit chunks (grid.y), every chunk leaves its
// partial outputs in `part` ([task][chunk][P][Q]), and small_wgrad_combine_kernel adds them in chunk order.
// ---------------------------------------------------------------------------------------
constexpr int MAX_TASKS = 24;
constexpr int PT = 8;
struct WgradTask {
  const float* U; int ldU, P;
  const float* V; int ldV, Q;
  float* out; int sp, sq;
  int tile0;  // first workgroup of this task
  int part0;  // offset of this task's partial outputs in WgradBatch::part (nsplit > 1)
};
struct WgradBatch {
  WgradTask task[MAX_TASKS];
  int ntask, B;
  int nsplit, chunk;         // cells [y * chunk, (y + 1) * chunk) for grid row y
  float* part;               // nsplit > 1: partial outputs; part_total floats = sum of P * Q over the tasks, per chunk
  int part_total;
  int cr_tile0;              // workgroups from here on reduce the epilogue backward's column slabs (cr.slab != null)
  VaeDeferredColreduce cr;
};

constexpr int TW = 1024;                  // threads of a small_wgrad workgroup
constexpr int NR = TW / 64;               // cell residues: thread (l, r) sums the cells b = r (mod NR)
__global__ __launch_bounds__(TW) void small_wgrad_kernel(WgradBatch wb) {
  constexpr int RC = 256;                 // cells staged per round
  __shared__ float Vs[RC][64];            // V[b0+row][q0+col]
  __shared__ float Us[RC][PT];            // U[b0+row][p0+i] (ones for a column sum)
  float (*comb)[PT][64] = reinterpret_cast<float (*)[PT][64]>(&Vs[0][0]);  // [NR][PT][64], reuses Vs after the last round
  if ((int)blockIdx.x >= wb.cr_tile0) {   // d px_r / d head biases: one column per thread, slabs summed in chunk order
    if (blockIdx.y != 0) return;
    const int strips = (wb.cr.G + TW - 1) / TW;
    const int t = blockIdx.x - wb.cr_tile0, kk = t / strips, g = (t - kk * strips) * TW + threadIdx.x;
    if (g < wb.cr.G && wb.cr.out[kk]) vae_colreduce_column(wb.cr.slab, wb.cr.chunks, wb.cr.G, kk, g, wb.cr.out[kk], 1);
    return;
  }
  int k = 0;
#pragma unroll
  for (int i = 1; i < MAX_TASKS; ++i) k += (i < wb.ntask && (int)blockIdx.x >= wb.task[i].tile0) ? 1 : 0;
  const WgradTask tk = wb.task[k];
  const int tile = blockIdx.x - tk.tile0;
  const int qtiles = (tk.Q + 63) / 64;
  const int p0 = (tile / qtiles) * PT, q0 = (tile % qtiles) * 64;
  const int l = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int np = min(PT, tk.P - p0);
  const bool qok = q0 + l < tk.Q;
  float acc[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) acc[i] = 0.f;

  const int bbeg = min(wb.B, (int)blockIdx.y * wb.chunk), bend = min(wb.B, bbeg + wb.chunk);   // this grid row's cells
  constexpr int UJ = (RC * PT + TW - 1) / TW;
  float vreg[RC / NR], ureg[UJ];
  auto fetch = [&](int b0) {  // thread (l, r) loads column l of rows r, r+NR, ...: coalesced, all in flight
    const int rows = min(RC, bend - b0);
#pragma unroll
    for (int j = 0; j < RC / NR; ++j) {
      const int row = r + NR * j;
      vreg[j] = (row < rows && qok) ? tk.V[(int64_t)(b0 + row) * tk.ldV + q0 + l] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < UJ; ++j) {
      const int idx = threadIdx.x + TW * j, row = idx / PT, i = idx % PT;
      float u = 0.f;
      if (row < rows && i < np) u = tk.U ? tk.U[(int64_t)(b0 + row) * tk.ldU + p0 + i] : 1.f;
      ureg[j] = u;
    }
  };
  fetch(bbeg);
  for (int b0 = bbeg; b0 < bend; b0 += RC) {
    const int rows = min(RC, bend - b0);
    __syncthreads();  // previous round's reads are done
#pragma unroll
    for (int j = 0; j < RC / NR; ++j) Vs[r + NR * j][l] = vreg[j];
#pragma unroll
    for (int j = 0; j < UJ; ++j) {
      const int idx = threadIdx.x + TW * j;
      if (idx < RC * PT) Us[idx / PT][idx % PT] = ureg[j];
    }
    __syncthreads();
    if (b0 + RC < bend) fetch(b0 + RC);  // next round's loads fly during this round's sums
    // thread (l, r) accumulates rows r, r+NR, ... in ascending order
    for (int row = r; row < rows; row += NR) {
      const float v = Vs[row][l];
      const float4 u0 = *reinterpret_cast<const float4*>(&Us[row][0]);
      const float4 u1 = *reinterpret_cast<const float4*>(&Us[row][4]);
      acc[0] = fmaf(u0.x, v, acc[0]); acc[1] = fmaf(u0.y, v, acc[1]);
      acc[2] = fmaf(u0.z, v, acc[2]); acc[3] = fmaf(u0.w, v, acc[3]);
      acc[4] = fmaf(u1.x, v, acc[4]); acc[5] = fmaf(u1.y, v, acc[5]);
      acc[6] = fmaf(u1.z, v, acc[6]); acc[7] = fmaf(u1.w, v, acc[7]);
    }
  }
  __syncthreads();  // the last round's reads of Vs are done
#pragma unroll
  for (int i = 0; i < PT; ++i) comb[r][i][l] = acc[i];
  __syncthreads();
  if (r < np && qok) {   // wave r finishes output row p0 + r: the NR partial sums in residue order
    float v = comb[0][r][l];
#pragma unroll
    for (int j = 1; j < NR; ++j) v += comb[j][r][l];
    if (wb.nsplit > 1) wb.part[(int64_t)blockIdx.y * wb.part_total + tk.part0 + (p0 + r) * tk.Q + q0 + l] = v;
    else tk.out[(int64_t)(p0 + r) * tk.sp + (int64_t)(q0 + l) * tk.sq] = v;
  }
}
// out = sum over the chunks of the partial outputs, in chunk order: one thread per output element
__global__ __launch_bounds__(256) void small_wgrad_combine_kernel(WgradBatch wb) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= wb.part_total) return;
  int k = 0;
#pragma unroll
  for (int i = 1; i < MAX_TASKS; ++i) k += (i < wb.ntask && e >= wb.task[i].part0) ? 1 : 0;
  const WgradTask tk = wb.task[k];
  const int o = e - tk.part0, p = o / tk.Q, q = o - p * tk.Q;
  float v = 0.f;
  for (int s = 0; s < wb.nsplit; ++s) v += wb.part[(int64_t)s * wb.part_total + e];
  tk.out[(int64_t)p * tk.sp + (int64_t)q * tk.sq] = v;
}

// Launches a filled task table (task[], ntask, B, cr_tile0, cr): one pass, or -- from 2048 cells on, when `space`
// (capacity floats) can hold the chunks' partial outputs -- chunks of 512 cells plus the combine launch.
int launch_wgrad_batch(WgradBatch& wb, int tiles, int ptotal, float* space, int64_t capacity, void* stream) {
  const int split_min_b = getenv("VAE_SMALL_WGRAD_SPLIT_MIN_B") ? atoi(getenv("VAE_SMALL_WGRAD_SPLIT_MIN_B")) : 2048;   // A/B knob
  wb.nsplit = 1; wb.chunk = wb.B; wb.part = nullptr; wb.part_total = ptotal;
  if (wb.B >= split_min_b && space) {
    // chunks of 512 cells (two staging rounds) -- unless that many chunks need more than TWO generations of workgroups
    // (1024 threads at 68 registers: one workgroup per CU, 256 slots; only the task tiles multiply with the chunks, the
    // column-reduction tiles work in grid row 0): then longer chunks of whole 256-cell rounds.  cfg3 (B = 4096), same box:
    // 512 cells 28.1 us, 768 (this rule) 26.0, 1024 30.2, 2048 (one generation) 29.0.  (Capped at 64 registers the kernel
    // runs two workgroups per CU with 12 bytes of scratch: 24.3 us there, but 11.2 -> 11.6 at B = 512 -- not taken.)
    static const int chunk_env = getenv("VAE_SMALL_WGRAD_CHUNK") ? atoi(getenv("VAE_SMALL_WGRAD_CHUNK")) : 0;   // A/B knob
    int chunk = 512;
    const int task_tiles = wb.cr_tile0 > 0 ? wb.cr_tile0 : tiles, other = tiles - task_tiles;
    if (chunk_env > 0) chunk = (chunk_env + 255) / 256 * 256;
    else
      while (chunk < wb.B && (int64_t)task_tiles * ((wb.B + chunk - 1) / chunk) + other > 512) chunk += 256;
    const int ns = (wb.B + chunk - 1) / chunk;
    if (ns > 1 && (int64_t)ns * ptotal <= capacity) { wb.nsplit = ns; wb.chunk = chunk; wb.part = space; }
  }
  hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)tiles, (unsigned)wb.nsplit), dim3(TW), 0, (hipStream_t)stream, wb);
  VAE_CHECK_LAUNCH("small_wgrad");
  if (wb.nsplit > 1) {
    hipLaunchKernelGGL(small_wgrad_combine_kernel, dim3((unsigned)((ptotal + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wb);
    VAE_CHECK_LAUNCH("small_wgrad_combine");
  }
  return VAE_OK;
}

// ---------------------------------------------------------------------------------------
// Cell-property heads on the latent: CellDecoder with linear heads (modules.py:297-324, the shipped
// cell_decoder_hidden_layer = False, config.py:67) + LitVAE._cell_loss (task.py:419-451) and their backward.
// One wave per cell: lane j < C (all heads' outputs side by side) forms its logit from the LDS-staged weights, the lanes
// of a discrete head share a softmax (cross-entropy, ignore_index -100), a continuous head is a squared error; every
// head's per-cell loss is weighted by its mask and averaged over ALL cells of the batch.  The same wave turns the logit
// gradients into d z (each head scaled by dz_coeff: 1, 0 for a stop_grad head that sees the latent detached, -lambda
// behind a gradient-reversal layer) and leaves them as a (B, C) matrix for the weight / bias sums (small_wgrad tasks).
// ---------------------------------------------------------------------------------------
struct HeadsArgs {
  vae_cell_heads h;
  const float* z; const float* targets; const float* mask;
  int64_t ld_t, ld_m;
  int B, L, C;
  float scale;          // loss_scale / B on the gradients
  float inv_b;
  float* dz;            // nullable (forward only)
  float* dlg;           // (B, C) logit gradients (nullable with dz)
  float* loss_cell;     // (B)
};
// lanes of ONE wave hand values to each other through LDS: order the stores before the loads (no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(256) void cell_heads_kernel(HeadsArgs a) {
  constexpr int MAXC = 64, MAXL = 64;
  __shared__ float Ws[MAXC][MAXL + 1];
  __shared__ float bs[MAXC], coef[MAXC];
  __shared__ int seg0[MAXC], segn[MAXC], head[MAXC], disc[MAXC];
  __shared__ float zs[4][MAXL], lg[4][MAXC], dls[4][MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = a.L, C = a.C;
  if (tid < MAXC) {   // output column -> head
    int j0 = 0, hh = 0;
    for (int i = 0; i < a.h.n_heads; ++i) {
      if (tid >= j0 && tid < j0 + a.h.width[i]) hh = i;
      j0 += a.h.width[i];
    }
    int s0 = 0;
    for (int i = 0; i < hh; ++i) s0 += a.h.width[i];
    head[tid] = hh; seg0[tid] = s0; segn[tid] = a.h.width[hh]; disc[tid] = a.h.discrete[hh]; coef[tid] = a.h.dz_coeff[hh];
    if (tid < C) bs[tid] = a.h.b[hh][tid - s0];
  }
  __syncthreads();
  for (int i = tid; i < C * L; i += 256) {
    const int j = i / L, l = i - j * L;
    Ws[j][l] = a.h.w[head[j]][(j - seg0[j]) * L + l];
  }
  __syncthreads();
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;   // no barrier below
  zs[wave][lane] = lane < L ? a.z[(int64_t)b * L + lane] : 0.f;
  wave_lds_sync();
  float logit = 0.f;
  if (lane < C) {
    logit = bs[lane];
    for (int l = 0; l < L; ++l) logit = fmaf(Ws[lane][l], zs[wave][l], logit);
  }
  lg[wave][lane] = logit;
  wave_lds_sync();
  float dl = 0.f, lossc = 0.f;
  if (lane < C) {
    const int hh = head[lane], s0 = seg0[lane], n = segn[lane];
    const float t = a.targets[(int64_t)b * a.ld_t + hh];
    const float mk = a.mask[(int64_t)b * a.ld_m + hh];
    if (disc[lane]) {
      float m = lg[wave][s0];
      for (int k = 1; k < n; ++k) m = fmaxf(m, lg[wave][s0 + k]);
      float ssum = 0.f;
      for (int k = 0; k < n; ++k) ssum += expf(lg[wave][s0 + k] - m);
      const int ti = (int)t;
      const bool valid = ti != -100 && ti >= 0 && ti < n;   // ignore_index (task.py:437)
      if (valid) {
        const float p = expf(logit - m) / ssum;
        const float* cw = a.h.class_weight[hh];              // balance_classes (task.py:129-132): weight of the target's class
        const float wk = cw ? cw[ti] * mk : mk;
        dl = (p - (lane - s0 == ti ? 1.f : 0.f)) * wk * a.scale;
        if (lane == s0) lossc = (logf(ssum) + m - lg[wave][s0 + ti]) * wk * a.inv_b;
      }
    } else {
      const float d = logit - t;
      dl = 2.f * d * mk * a.scale;
      lossc = d * d * mk * a.inv_b;
    }
  }
  lossc = wave_sum_dpp(lossc);
  if (lane == 0) a.loss_cell[b] = lossc;
  if (!a.dz) return;
  if (lane < C) a.dlg[(int64_t)b * C + lane] = dl;
  dls[wave][lane] = lane < C ? dl * coef[lane] : 0.f;
  wave_lds_sync();
  if (lane < L) {
    float acc = 0.f;
    for (int j = 0; j < C; ++j) acc = fmaf(dls[wave][j], Ws[j][lane], acc);
    a.dz[(int64_t)b * L + lane] = acc;
  }
}

}  // namespace

extern "C" int64_t vae_cell_heads_workspace_bytes(int64_t B, int64_t L, const vae_cell_heads* h) {
  if (!h || B <= 0 || L <= 0 || h->n_heads <= 0 || h->n_heads > VAE_MAX_HEADS) return -1;
  int64_t C = 0;
  for (int i = 0; i < h->n_heads; ++i) C += h->width[i];
  const int64_t ptotal = C * L + C + 1, ns = (B + 511) / 512;
  return (B * C + B + 8 + ns * ptotal + 8) * (int64_t)sizeof(float);
}

extern "C" int vae_cell_heads_step(const vae_cell_heads* h, const float* z, int64_t B, int64_t L, const float* targets,
                                   int64_t ld_t, const float* mask, int64_t ld_m, float loss_scale, float* dz, float* loss,
                                   float* workspace, int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(h && z && targets && mask && loss && workspace, "vae_cell_heads_step: null pointer");
  VAE_REQUIRE(h->n_heads > 0 && h->n_heads <= VAE_MAX_HEADS, "vae_cell_heads_step: 1..%d heads", VAE_MAX_HEADS);
  VAE_REQUIRE(B > 0 && B < (1 << 30) && L > 0 && L <= 64, "vae_cell_heads_step: need B > 0 and a latent of at most 64");
  int C = 0;
  for (int i = 0; i < h->n_heads; ++i) {
    VAE_REQUIRE(h->width[i] >= 1 && (h->discrete[i] || h->width[i] == 1), "vae_cell_heads_step: head %d: bad width", i);
    VAE_REQUIRE(h->w[i] && h->b[i] && (!dz || (h->dw[i] && h->db[i])), "vae_cell_heads_step: head %d: null parameter / gradient", i);
    C += h->width[i];
  }
  VAE_REQUIRE(C <= 64, "vae_cell_heads_step: at most 64 outputs over all heads, got %d", C);
  VAE_REQUIRE(ld_t >= h->n_heads && ld_m >= h->n_heads, "vae_cell_heads_step: targets / mask need one column per head");
  VAE_REQUIRE(workspace_bytes >= vae_cell_heads_workspace_bytes(B, L, h) && vae_aligned16(workspace),
              "vae_cell_heads_step: workspace too small or misaligned (vae_cell_heads_workspace_bytes)");
  HeadsArgs a;
  a.h = *h; a.z = z; a.targets = targets; a.mask = mask; a.ld_t = ld_t; a.ld_m = ld_m;
  a.B = (int)B; a.L = (int)L; a.C = C;
  a.inv_b = 1.f / (float)B; a.scale = loss_scale / (float)B;
  a.dz = dz;
  a.dlg = workspace;
  a.loss_cell = workspace + ((B * C + 3) & ~(int64_t)3);
  float* part = a.loss_cell + ((B + 3) & ~(int64_t)3);
  hipLaunchKernelGGL(cell_heads_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  VAE_CHECK_LAUNCH("vae_cell_heads_step");
  // sums over cells: d W_h = dlogits_h^T z, d b_h = column sums, loss = sum of the per-cell losses (fixed order)
  WgradBatch wb;
  int n = 0, tiles = 0, ptotal = 0;
  auto add = [&](const float* U, int ldU, int P, const float* V, int ldV, int Q, float* out, int sp, int sq) {
    WgradTask& k = wb.task[n++];
    k.U = U; k.ldU = ldU; k.P = P; k.V = V; k.ldV = ldV; k.Q = Q; k.out = out; k.sp = sp; k.sq = sq;
    k.tile0 = tiles;
    k.part0 = ptotal;
    tiles += ((P + PT - 1) / PT) * ((Q + 63) / 64);
    ptotal += P * Q;
  };
  add(nullptr, 0, 1, a.loss_cell, 1, 1, loss, 0, 1);
  if (dz) {
    int s0 = 0;
    for (int i = 0; i < h->n_heads; ++i) {
      add(a.dlg + s0, C, h->width[i], z, (int)L, (int)L, h->dw[i], (int)L, 1);
      add(nullptr, 0, 1, a.dlg + s0, C, h->width[i], h->db[i], 0, 1);
      s0 += h->width[i];
    }
  }
  wb.ntask = n; wb.B = (int)B; wb.cr_tile0 = tiles; wb.cr = VaeDeferredColreduce{};
  const int64_t cap = workspace_bytes / 4 - (part - workspace);
  return launch_wgrad_batch(wb, tiles, ptotal, part, cap, stream);
}

// ---- internal entry points used by step.hip (not part of the public C ABI) -----------------
bool vae_mid_supported(const vae_step_config& c) {
  static const bool off = getenv("VAE_NO_FUSED_MIDDLE") != nullptr;  // A/B knob: run the unfused launch sequence
  if (off) return false;
  if (c.n_cov > 0) return false;   // batch covariates widen the first decoder weight and add a product to both first layers
  return c.n_layers == 1 && c.H + c.Hl <= ME * T && c.Hd <= ME * T && c.L + 3 <= 64 && c.L >= 1 &&
         c.Hd * (c.L | 1) * 4 <= 48 * 1024;  // the first decoder weight is staged in LDS
}

int vae_mid_forward(const vae_step_config& c, const vae_step_tensors* p, const float* slabs, int nslab, const float* eps_z,
                    const float* eps_l, float* pre_enc0, float* h_enc0, float* ln_mz, float* ln_rz, float* ln_ml,
                    float* ln_rl, float* heads_z, float* heads_l, float* qz_v, float* z, float* kl_z, float* ql_v,
                    float* library, float* kl_l, float* dec_pre, float* dec_h, float* ln_md, float* ln_rd, float* absmax_px,
                    uint16_t* px16, const float* x_mean, const float* wsum, void* stream) {
  MidFwd a;
  a.x_mean = x_mean; a.wsum = x_mean ? wsum : nullptr;
  a.absmax_px = absmax_px;
  a.px16 = px16;
  a.slabs = slabs; a.nslab = nslab; a.slab_stride = c.B * (c.H + c.Hl);
  a.enc_b0 = p->enc_b0;
  a.gz = p->encz_ln_g[0]; a.bz = p->encz_ln_b[0]; a.gl = p->encl_ln_g; a.bl = p->encl_ln_b;
  a.hz_w = p->hz_w; a.hz_b = p->hz_b; a.hl_w = p->hl_w; a.hl_b = p->hl_b;
  a.eps_z = eps_z; a.eps_l = eps_l;
  a.d_w = p->dec_w[0]; a.d_b = p->dec_b[0]; a.gd = p->dec_ln_g[0]; a.bd = p->dec_ln_b[0];
  a.l_mean = c.l_mean; a.l_var = c.l_var;
  a.B = (int)c.B; a.H = (int)c.H; a.Hl = (int)c.Hl; a.L = (int)c.L; a.Hd = (int)c.Hd;
  a.pre_enc0 = pre_enc0; a.h_enc0 = h_enc0;
  a.ln_mz = ln_mz; a.ln_rz = ln_rz; a.ln_ml = ln_ml; a.ln_rl = ln_rl;
  a.heads_z = heads_z; a.heads_l = heads_l; a.qz_v = qz_v; a.z = z; a.kl_z = kl_z;
  a.ql_v = ql_v; a.library = library; a.kl_l = kl_l;
  a.dec_pre = dec_pre; a.dec_h = dec_h; a.ln_md = ln_md; a.ln_rd = ln_rd;
  a.drop = vae_make_dropout(c.dropout_rate, c.rng_seed, c.rng_step);
  MidWavePlan pl;
  if (mid_wave_plan(c, p->hz_w, &pl)) MID_WAVE_DISPATCH(mid_fwd_wave_kernel, a);
  else
    hipLaunchKernelGGL(mid_fwd_kernel, dim3((unsigned)c.B), dim3(T), dec_w_lds_bytes((int)c.Hd, (int)c.L), (hipStream_t)stream, a);
  VAE_CHECK_LAUNCH("vae_elbo_step_forward(mid)");
  return VAE_OK;
}

int vae_mid_backward(const vae_step_config& c, const vae_step_tensors* p, const vae_step_tensors* g, const float* slabs,
                     int nslab, const float* eps_z, const float* eps_l, const float* dz_extra, float g_kl,
                     const float* dec_pre, const float* ln_md, const float* ln_rd, const float* heads_z,
                     const float* heads_l, const float* qz_v, const float* ql_v, const float* dlibrary,
                     const float* pre_enc0, const float* ln_mz, const float* ln_rz, const float* ln_ml,
                     const float* ln_rl, const float* z, const float* h_enc0, float* dbet_d, float* dgam_d,
                     float* dpre_d, float* dq_z, float* dq_l, float* dbet_e, float* dgam_e, float* dpre_e,
                     float* absmax_dpre, uint16_t* dpre16, const VaeDeferredColreduce* cr, const float* x_mean, float* dpre_s,
                     void* stream) {
  MidBwd a;
  a.x_mean = dpre_s ? x_mean : nullptr; a.dpre_s = x_mean ? dpre_s : nullptr;
  a.dpre16 = dpre16;
  a.absmax_dpre = absmax_dpre;
  a.slabs = slabs; a.nslab = nslab; a.slab_stride = c.B * c.Hd;
  a.dec_pre = dec_pre; a.ln_md = ln_md; a.ln_rd = ln_rd; a.gd = p->dec_ln_g[0];
  a.d_w = p->dec_w[0];
  a.dz_extra = dz_extra;
  a.heads_z = heads_z; a.heads_l = heads_l; a.eps_z = eps_z; a.eps_l = eps_l; a.qz_v = qz_v; a.ql_v = ql_v;
  a.dlibrary = dlibrary;
  a.g_kl = g_kl; a.l_mean = c.l_mean; a.l_var = c.l_var;
  a.hz_w = p->hz_w; a.hl_w = p->hl_w;
  a.pre_enc0 = pre_enc0; a.ln_mz = ln_mz; a.ln_rz = ln_rz; a.ln_ml = ln_ml; a.ln_rl = ln_rl;
  a.gz = p->encz_ln_g[0]; a.gl = p->encl_ln_g;
  a.B = (int)c.B; a.H = (int)c.H; a.Hl = (int)c.Hl; a.L = (int)c.L; a.Hd = (int)c.Hd;
  a.dbet_d = dbet_d; a.dgam_d = dgam_d; a.dpre_d = dpre_d;
  a.dq_z = dq_z; a.dq_l = dq_l;
  a.dbet_e = dbet_e; a.dgam_e = dgam_e; a.dpre_e = dpre_e;
  a.drop = vae_make_dropout(c.dropout_rate, c.rng_seed, c.rng_step);
  MidWavePlan pl;
  if (mid_wave_plan(c, p->hz_w, &pl)) MID_WAVE_DISPATCH(mid_bwd_wave_kernel, a);
  else
    hipLaunchKernelGGL(mid_bwd_kernel, dim3((unsigned)c.B), dim3(T), dec_w_lds_bytes((int)c.Hd, (int)c.L), (hipStream_t)stream, a);
  VAE_CHECK_LAUNCH("vae_elbo_step_backward(mid)");

  // every small parameter gradient of the step, one launch
  const int B = (int)c.B, H = (int)c.H, Hl = (int)c.Hl, HE = H + Hl, L = (int)c.L, Hd = (int)c.Hd;
  WgradBatch wb;
  int n = 0, tiles = 0, ptotal = 0;
  auto add = [&](const float* U, int ldU, int P, const float* V, int ldV, int Q, float* out, int sp, int sq) {
    WgradTask& k = wb.task[n++];
    k.U = U; k.ldU = ldU; k.P = P; k.V = V; k.ldV = ldV; k.Q = Q; k.out = out; k.sp = sp; k.sq = sq;
    k.tile0 = tiles;
    k.part0 = ptotal;
    tiles += ((P + PT - 1) / PT) * ((Q + 63) / 64);
    ptotal += P * Q;
  };
  add(z, L, L, dpre_d, Hd, Hd, g->dec_w[0], 1, L);              // dW_dec0[n][j] = sum_b dpre_d[b][n] z[b][j]
  add(dq_z, L + 1, L + 1, h_enc0, HE, H, g->hz_w, H, 1);         // dW_heads_z[j][k] = sum_b dq_z[b][j] h_z[b][k]
  add(dq_l, 2, 2, h_enc0 + H, HE, Hl, g->hl_w, Hl, 1);           // dW_heads_l
  add(nullptr, 0, 1, dpre_d, Hd, Hd, g->dec_b[0], 0, 1);
  add(nullptr, 0, 1, dgam_d, Hd, Hd, g->dec_ln_g[0], 0, 1);
  add(nullptr, 0, 1, dbet_d, Hd, Hd, g->dec_ln_b[0], 0, 1);
  add(nullptr, 0, 1, dq_z, L + 1, L + 1, g->hz_b, 0, 1);
  add(nullptr, 0, 1, dq_l, 2, 2, g->hl_b, 0, 1);
  add(nullptr, 0, 1, dgam_e, HE, H, g->encz_ln_g[0], 0, 1);
  add(nullptr, 0, 1, dgam_e + H, HE, Hl, g->encl_ln_g, 0, 1);
  add(nullptr, 0, 1, dbet_e, HE, H, g->encz_ln_b[0], 0, 1);
  add(nullptr, 0, 1, dbet_e + H, HE, Hl, g->encl_ln_b, 0, 1);
  add(nullptr, 0, 1, dpre_e, HE, HE, g->enc_b0, 0, 1);
  wb.ntask = n;
  wb.B = B;
  wb.cr_tile0 = tiles;
  wb.cr = VaeDeferredColreduce{};
  if (cr && cr->slab) {
    wb.cr = *cr;
    tiles += cr->nk * ((cr->G + TW - 1) / TW);
  }
  // chunks of 512 cells from B = 2048 on; their partial outputs go where the dgrad slabs were (mid_bwd, enqueued above,
  // was their last reader)
  return launch_wgrad_batch(wb, tiles, ptotal, nslab > 0 ? const_cast<float*>(slabs) : nullptr, (int64_t)nslab * B * Hd, stream);
}
