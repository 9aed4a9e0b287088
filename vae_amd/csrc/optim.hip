// Fused optimiser step over the flat parameter / gradient arenas (gfx950):
//   global-norm gradient clipping  (Trainer(gradient_clip_val=0.5) -> torch clip_grad_norm_, train.py:54)
//   AdamW, decoupled weight decay per range of the arena, bias correction
//                                  (torch.optim.AdamW(betas=(0.9, 0.998), eps=1e-7), task.py:534-553)
// Two launches for all 18 M parameters: (1) per-workgroup partial sums of g^2, (2) every
// workgroup re-sums the partials in the same fixed order (-> identical clip coefficient in every
// workgroup, deterministic) and updates its slice.  HBM-bound: 28 bytes per parameter
// (read p, g, m, v; write p, m, v) plus 4 for the norm pass.
#include "../../include/vae_amd.h"
#include "common.h"

#include <math.h>

namespace {

constexpr int MAX_SEG = 16;
constexpr int NORM_BLOCKS = 1024;

struct Segs {
  int64_t begin[MAX_SEG], end[MAX_SEG];
  float decay[MAX_SEG];  // 1 - lr * weight_decay of the range
  int n;
};

// index of the active range holding element i, or -1
__device__ __forceinline__ int seg_of(const Segs& s, int64_t i) {
  int k = -1;
#pragma unroll
  for (int j = 0; j < MAX_SEG; ++j)
    if (j < s.n && i >= s.begin[j] && i < s.end[j]) k = j;
  return k;
}

// partial[block] = this workgroup's share of sum over the ACTIVE ranges of (g * gscale)^2.  Range by range: the 16-byte
// aligned interior of a range is a plain grid-stride stream with eight loads in flight per thread and no range question per
// element (one dependent load per iteration plus a 16-range lookup per element ran at 1.7 TB/s; per-lane range checks on
// every element still cost as much as the loads); the up to three elements at either unaligned end go to workgroup 0.
// Fixed order per thread, fixed partial per workgroup: deterministic.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, Segs s, float gscale,
                                                    float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  constexpr int UF = 8;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  auto sq4 = [&](const float4& v) {
    const float f0 = v.x * gscale, f1 = v.y * gscale, f2 = v.z * gscale, f3 = v.w * gscale;
    return (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
  };
  for (int k = 0; k < s.n; ++k) {
    const int64_t b = s.begin[k], e = s.end[k] < n ? s.end[k] : n;
    if (e <= b) continue;
    const int64_t b4 = (b + 3) >> 2, e4 = e >> 2;   // 16-byte chunks [b4, e4) lie wholly inside the range
    int64_t i = b4 + t0;
    for (; i + (UF - 1) * stride < e4; i += UF * stride) {
      float4 v[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) v[u] = reinterpret_cast<const float4*>(g)[i + u * stride];
#pragma unroll
      for (int u = 0; u < UF; ++u) acc += sq4(v[u]);
    }
    for (; i < e4; i += stride) acc += sq4(reinterpret_cast<const float4*>(g)[i]);
    if (blockIdx.x == 0 && threadIdx.x < 8) {   // the unaligned ends (none in the arena's own ranges)
      const int64_t head_end = (b4 << 2) < e ? (b4 << 2) : e, tail_begin = (e4 << 2) > head_end ? (e4 << 2) : head_end;
      const int64_t el = threadIdx.x < 4 ? b + threadIdx.x : tail_begin + (threadIdx.x - 4);
      const bool ok = threadIdx.x < 4 ? el < head_end : el < e;
      if (ok) { const float f = g[el] * gscale; acc += f * f; }
    }
  }
  acc = block_sum<4>(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* base, int64_t i4) {
  const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(base) + i4);
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(float* base, int64_t i4, float4 v) {
  nt_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(base) + i4);
}
template <bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, Segs s,
                                                    float gscale, float beta1, float beta2, float eps, float step_size,
                                                    float bc2_sqrt, float max_norm, const float* __restrict__ partial,
                                                    int npartial, float* __restrict__ norm_out, int64_t r0b, int64_t r0e,
                                                    int64_t r1b, int64_t r1e, float* __restrict__ wmax,
                                                    uint16_t* __restrict__ img0, uint16_t* __restrict__ img1) {
  // img0 / img1 != null: bf16 images (nearest even) of the updated parameters of the two tracked ranges
  auto bf = [](float v) { const __bf16 h = (__bf16)v; return __builtin_bit_cast(uint16_t, h); };
  __shared__ float red[4];
  float wm0 = 0.f, wm1 = 0.f;   // max |updated parameter| of this workgroup inside the two tracked ranges (wmax != null)
  // clip coefficient: every workgroup sums the same partials in the same order
  float coef = 1.f;
  if (max_norm > 0.f) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < npartial; i += 256) acc += partial[i];
    acc = block_sum<4>(acc, red);
    const float total = sqrtf(acc);
    coef = fminf(max_norm / (total + 1e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_
    if (total != total) coef = total;               // ... whose clamp(max=1) lets a NaN norm through (fminf would drop it)
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = total;
  }
  const float gs = gscale * coef;
  auto upd = [&](float& pw, float gw, float& mw, float& vw, float decay) {
    const float gg = gw * gs;
    pw *= decay;                                   // p.mul_(1 - lr * wd)
    mw = mw + (gg - mw) * (1.f - beta1);           // exp_avg.lerp_(grad, 1 - beta1)
    vw = vw * beta2 + (1.f - beta2) * gg * gg;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vw) / bc2_sqrt + eps;
    pw = pw + (-step_size * mw) / denom;           // p.addcdiv_(exp_avg, denom, value=-step_size)
  };
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  // two 16-byte vectors of each of the four arrays in flight per thread and iteration
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    int64_t idx[2] = {i0, i0 + stride};
    int kk[2];
    float4 pv[2], gv[2], mv[2], vv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      kk[u] = -2;  // -2: past the end, -1: not (entirely) inside one range
      if (idx[u] < n4) {
        const int k0 = seg_of(s, idx[u] << 2), k3 = seg_of(s, (idx[u] << 2) + 3);
        kk[u] = (k0 >= 0 && k0 == k3) ? k0 : -1;
      }
      if (kk[u] >= 0) {
        pv[u] = reinterpret_cast<float4*>(p)[idx[u]];
        gv[u] = reinterpret_cast<const float4*>(g)[idx[u]];
        if (NT) {   // the two moment arrays are touched once per step: keep them out of the caches the weights live in
          mv[u] = nt_load4(m, idx[u]);
          vv[u] = nt_load4(v, idx[u]);
        } else {
          mv[u] = reinterpret_cast<float4*>(m)[idx[u]];
          vv[u] = reinterpret_cast<float4*>(v)[idx[u]];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (kk[u] >= 0) {
        const float d = s.decay[kk[u]];
        upd(pv[u].x, gv[u].x, mv[u].x, vv[u].x, d); upd(pv[u].y, gv[u].y, mv[u].y, vv[u].y, d);
        upd(pv[u].z, gv[u].z, mv[u].z, vv[u].z, d); upd(pv[u].w, gv[u].w, mv[u].w, vv[u].w, d);
        reinterpret_cast<float4*>(p)[idx[u]] = pv[u];
        if (NT) {
          nt_store4(m, idx[u], mv[u]);
          nt_store4(v, idx[u], vv[u]);
        } else {
          reinterpret_cast<float4*>(m)[idx[u]] = mv[u];
          reinterpret_cast<float4*>(v)[idx[u]] = vv[u];
        }
        if (wmax) {   // a 16-byte chunk inside one segment lies inside at most one tracked range (ranges are segments' parts)
          const int64_t e = idx[u] << 2;
          const float mx = fmaxf(fmaxf(fabsf(pv[u].x), fabsf(pv[u].y)), fmaxf(fabsf(pv[u].z), fabsf(pv[u].w)));
          if (e >= r0b && e < r0e) wm0 = fmaxf(wm0, mx);          // range ends are multiples of 4: a chunk is in or out
          else if (e >= r1b && e < r1e) wm1 = fmaxf(wm1, mx);
        }
        if (img0) {   // (img1 is set with it) 8-byte store of the chunk's four images
          const int64_t e = idx[u] << 2;
          uint16_t* d = (e >= r0b && e < r0e) ? img0 + (e - r0b) : ((e >= r1b && e < r1e) ? img1 + (e - r1b) : nullptr);
          if (d) {
            const uint2 o = make_uint2((uint32_t)bf(pv[u].x) | ((uint32_t)bf(pv[u].y) << 16),
                                       (uint32_t)bf(pv[u].z) | ((uint32_t)bf(pv[u].w) << 16));
            *reinterpret_cast<uint2*>(d) = o;
          }
        }
      } else if (kk[u] == -1) {
        const int64_t e = idx[u] << 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = seg_of(s, e + j);
          if (k >= 0) {
            upd(p[e + j], g[e + j], m[e + j], v[e + j], s.decay[k]);
            if (wmax && e + j >= r0b && e + j < r0e) wm0 = fmaxf(wm0, fabsf(p[e + j]));
            if (wmax && e + j >= r1b && e + j < r1e) wm1 = fmaxf(wm1, fabsf(p[e + j]));
            if (img0 && e + j >= r0b && e + j < r0e) img0[e + j - r0b] = bf(p[e + j]);
            if (img0 && e + j >= r1b && e + j < r1e) img1[e + j - r1b] = bf(p[e + j]);
          }
        }
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    const int k = seg_of(s, e);
    if (k >= 0) upd(p[e], g[e], m[e], v[e], s.decay[k]);   // (past every tracked range: their ends are multiples of 4)
  }
  if (wmax) {   // plain stores, every entry every step: nothing to clear (the tracked ranges are 16-byte aligned multiples of 4)
    wm0 = block_max<4>(wm0, red);
    wm1 = block_max<4>(wm1, red);
    if (threadIdx.x == 0) {
      wmax[blockIdx.x] = wm0;
      wmax[gridDim.x + blockIdx.x] = wm1;
    }
  }
}

// L1 penalty over ranges of the arena (LitVAE.training_step, task.py:491-496): partial sums of |p| per
// workgroup, and grads += coeff * sign(p).  l1_finish_kernel adds the partials in index order.
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ p, float* __restrict__ g, Segs s, float coeff,
                                                 float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int k = 0; k < s.n; ++k)
    for (int64_t i = s.begin[k] + (int64_t)blockIdx.x * 256 + threadIdx.x; i < s.end[k]; i += (int64_t)gridDim.x * 256) {
      const float w = p[i];
      acc += fabsf(w);
      if (g) g[i] += w > 0.f ? coeff : (w < 0.f ? -coeff : 0.f);
    }
  acc = block_sum<4>(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void l1_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = block_sum<4>(acc, red);
  if (threadIdx.x == 0) out[0] = acc;
}

}  // namespace

extern "C" int vae_l1_penalty(const float* params, float* grads, int64_t n, const vae_adamw_segment* segments,
                              int n_segments, double grad_coeff, float* workspace, int64_t workspace_bytes,
                              float* sum_out, void* stream) {
  VAE_REQUIRE(params && sum_out && n > 0, "vae_l1_penalty: null pointer or empty arena");
  VAE_REQUIRE(segments && n_segments > 0 && n_segments <= MAX_SEG, "vae_l1_penalty: 1..%d segments, got %d", MAX_SEG, n_segments);
  VAE_REQUIRE(workspace && workspace_bytes >= NORM_BLOCKS * (int64_t)sizeof(float),
              "vae_l1_penalty: needs vae_adamw_workspace_bytes() of workspace");
  Segs s;
  s.n = n_segments;
  int64_t total = 0;
  for (int i = 0; i < MAX_SEG; ++i) { s.begin[i] = s.end[i] = 0; s.decay[i] = 1.f; }
  for (int i = 0; i < n_segments; ++i) {
    VAE_REQUIRE(segments[i].begin >= 0 && segments[i].begin <= segments[i].end && segments[i].end <= n,
                "vae_l1_penalty: segment %d out of range", i);
    VAE_REQUIRE(i == 0 || segments[i].begin >= segments[i - 1].end, "vae_l1_penalty: segments must be sorted and disjoint");
    s.begin[i] = segments[i].begin;
    s.end[i] = segments[i].end;
    total += segments[i].end - segments[i].begin;
  }
  int blocks = (int)((total + 1023) / 1024);
  if (blocks > NORM_BLOCKS) blocks = NORM_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_kernel, dim3(blocks), dim3(256), 0, st, params, grads, s, (float)grad_coeff, workspace);
  VAE_CHECK_LAUNCH("vae_l1_penalty");
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, st, workspace, blocks, sum_out);
  VAE_CHECK_LAUNCH("vae_l1_penalty(finish)");
  return VAE_OK;
}

// NORM_BLOCKS partial sums of g^2, then 2 x 4096 raw partial maxima of vae_adamw_step_wmax
extern "C" int64_t vae_adamw_workspace_bytes(void) { return (NORM_BLOCKS + 2 * 4096) * (int64_t)sizeof(float); }

constexpr int WMAX_OUT = 64;         // vae_adamw_wmax_partials(): entries per tracked range a consumer has to reduce
extern "C" int64_t vae_adamw_wmax_partials(void) { return WMAX_OUT; }
namespace {
// raw[r * blocks + b] (one per update workgroup and range) -> out[r * WMAX_OUT + j] = max over b = j (mod WMAX_OUT)
__global__ __launch_bounds__(2 * WMAX_OUT) void wmax_finish_kernel(const float* __restrict__ raw, int blocks, float* __restrict__ out) {
  const int r = threadIdx.x / WMAX_OUT, j = threadIdx.x % WMAX_OUT;
  float m = 0.f;
  int b = j;
  for (; b + 15 * WMAX_OUT < blocks; b += 16 * WMAX_OUT) {   // sixteen loads in flight (thousands of workgroups: a chain otherwise)
    float q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = raw[r * blocks + b + u * WMAX_OUT];
#pragma unroll
    for (int u = 0; u < 16; ++u) m = fmaxf(m, q[u]);
  }
  for (; b < blocks; b += WMAX_OUT) m = fmaxf(m, raw[r * blocks + b]);
  out[r * WMAX_OUT + j] = m;
}
}  // namespace

static int adamw_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2,
                      double eps, int64_t step, double max_grad_norm, double grad_scale, float* workspace,
                      int64_t workspace_bytes, float* grad_norm_out, const int64_t* max_ranges, float* wmax, void* stream,
                      uint16_t* img0 = nullptr, uint16_t* img1 = nullptr);

extern "C" int vae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2,
                              double eps, int64_t step, double max_grad_norm, double grad_scale, float* workspace,
                              int64_t workspace_bytes, float* grad_norm_out, void* stream) {
  return adamw_impl(params, grads, exp_avg, exp_avg_sq, n, segments, n_segments, lr, beta1, beta2, eps, step, max_grad_norm,
                    grad_scale, workspace, workspace_bytes, grad_norm_out, nullptr, nullptr, stream);
}

extern "C" int vae_adamw_step_wmax(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2,
                                   double eps, int64_t step, double max_grad_norm, double grad_scale, float* workspace,
                                   int64_t workspace_bytes, float* grad_norm_out, const int64_t* max_ranges, float* wmax_out,
                                   void* stream) {
  VAE_REQUIRE(max_ranges && wmax_out, "vae_adamw_step_wmax: null max_ranges / wmax_out");
  VAE_REQUIRE(workspace && workspace_bytes >= vae_adamw_workspace_bytes(), "vae_adamw_step_wmax: needs vae_adamw_workspace_bytes()");
  for (int r = 0; r < 2; ++r)
    VAE_REQUIRE(max_ranges[2 * r] >= 0 && max_ranges[2 * r] <= max_ranges[2 * r + 1] && max_ranges[2 * r + 1] <= n &&
                    max_ranges[2 * r] % 4 == 0 && max_ranges[2 * r + 1] % 4 == 0,
                "vae_adamw_step_wmax: tracked range %d must lie in the arena with both ends multiples of 4", r);
  return adamw_impl(params, grads, exp_avg, exp_avg_sq, n, segments, n_segments, lr, beta1, beta2, eps, step, max_grad_norm,
                    grad_scale, workspace, workspace_bytes, grad_norm_out, max_ranges, wmax_out, stream);
}

extern "C" int vae_adamw_step_bf16(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2,
                                   double eps, int64_t step, double max_grad_norm, double grad_scale, float* workspace,
                                   int64_t workspace_bytes, float* grad_norm_out, const int64_t* img_ranges, void* img0, void* img1,
                                   void* stream) {
  VAE_REQUIRE(img_ranges && img0 && img1, "vae_adamw_step_bf16: null img_ranges / img0 / img1");
  VAE_REQUIRE((reinterpret_cast<uintptr_t>(img0) & 7u) == 0 && (reinterpret_cast<uintptr_t>(img1) & 7u) == 0,
              "vae_adamw_step_bf16: the images must be 8-byte aligned");
  for (int r = 0; r < 2; ++r)
    VAE_REQUIRE(img_ranges[2 * r] >= 0 && img_ranges[2 * r] <= img_ranges[2 * r + 1] && img_ranges[2 * r + 1] <= n &&
                    img_ranges[2 * r] % 4 == 0 && img_ranges[2 * r + 1] % 4 == 0,
                "vae_adamw_step_bf16: tracked range %d must lie in the arena with both ends multiples of 4", r);
  VAE_REQUIRE(img_ranges[1] <= img_ranges[2] || img_ranges[3] <= img_ranges[0], "vae_adamw_step_bf16: the two ranges overlap");
  return adamw_impl(params, grads, exp_avg, exp_avg_sq, n, segments, n_segments, lr, beta1, beta2, eps, step, max_grad_norm,
                    grad_scale, workspace, workspace_bytes, grad_norm_out, img_ranges, nullptr, stream, (uint16_t*)img0,
                    (uint16_t*)img1);
}

static int adamw_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const vae_adamw_segment* segments, int n_segments, double lr, double beta1, double beta2,
                      double eps, int64_t step, double max_grad_norm, double grad_scale, float* workspace,
                      int64_t workspace_bytes, float* grad_norm_out, const int64_t* max_ranges, float* wmax, void* stream,
                      uint16_t* img0, uint16_t* img1) {
  VAE_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0, "vae_adamw_step: null pointer or empty arena");
  VAE_REQUIRE(segments && n_segments > 0 && n_segments <= MAX_SEG, "vae_adamw_step: 1..%d segments, got %d", MAX_SEG, n_segments);
  VAE_REQUIRE(step >= 1, "vae_adamw_step: step counts from 1");
  VAE_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "vae_adamw_step: bad hyper-parameter");
  VAE_REQUIRE(vae_aligned16(params) && vae_aligned16(grads) && vae_aligned16(exp_avg) && vae_aligned16(exp_avg_sq),
              "vae_adamw_step: the four arenas must be 16-byte aligned");
  VAE_REQUIRE(max_grad_norm <= 0 || (workspace && workspace_bytes >= vae_adamw_workspace_bytes()),
              "vae_adamw_step: clipping needs vae_adamw_workspace_bytes() of workspace");
  Segs s;
  s.n = n_segments;
  for (int i = 0; i < n_segments; ++i) {
    VAE_REQUIRE(segments[i].begin >= 0 && segments[i].begin <= segments[i].end && segments[i].end <= n,
                "vae_adamw_step: segment %d out of range", i);
    VAE_REQUIRE(i == 0 || segments[i].begin >= segments[i - 1].end, "vae_adamw_step: segments must be sorted and disjoint");
    s.begin[i] = segments[i].begin;
    s.end[i] = segments[i].end;
    s.decay[i] = (float)(1.0 - lr * (double)segments[i].weight_decay);
  }
  for (int i = n_segments; i < MAX_SEG; ++i) { s.begin[i] = s.end[i] = 0; s.decay[i] = 1.f; }
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  int blocks = (int)(((n >> 2) + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const int nblocks = blocks < NORM_BLOCKS ? blocks : NORM_BLOCKS;
  if (max_grad_norm > 0) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks), dim3(256), 0, st, grads, n, s, (float)grad_scale, workspace);
    VAE_CHECK_LAUNCH("vae_adamw_step(norm)");
  }
  // nontemporal loads / stores of the two moment arrays (stream-once data: 16 of the 28 bytes per parameter): 140 -> 136 us
  // alone, 0.572 -> 0.5635 ms per end-to-end training step at cfg2 (tools/ab_optim.sh); VAE_ADAMW_NT=0: plain accesses
  static const bool nt = getenv("VAE_ADAMW_NT") ? atoi(getenv("VAE_ADAMW_NT")) != 0 : true;
#define VAE_ADAMW_LAUNCH(NT)                                                                                              \
  hipLaunchKernelGGL(adamw_kernel<NT>, dim3(blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, s,         \
                     (float)grad_scale, (float)beta1, (float)beta2, (float)eps, step_size, bc2_sqrt, (float)max_grad_norm, \
                     workspace, nblocks, grad_norm_out, max_ranges ? max_ranges[0] : 0, max_ranges ? max_ranges[1] : 0,  \
                     max_ranges ? max_ranges[2] : 0, max_ranges ? max_ranges[3] : 0, wmax ? workspace + NORM_BLOCKS : nullptr, img0, \
                     img1)
  if (nt) VAE_ADAMW_LAUNCH(true);
  else VAE_ADAMW_LAUNCH(false);
#undef VAE_ADAMW_LAUNCH
  VAE_CHECK_LAUNCH("vae_adamw_step");
  if (wmax) {
    hipLaunchKernelGGL(wmax_finish_kernel, dim3(1), dim3(2 * WMAX_OUT), 0, st, workspace + NORM_BLOCKS, blocks, wmax);
    VAE_CHECK_LAUNCH("vae_adamw_step_wmax(finish)");
  }
  return VAE_OK;
}
