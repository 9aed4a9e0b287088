// Shared device/host helpers for the vae_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VAE_OK 0
#define VAE_ERR_ARG 1
#define VAE_ERR_LAUNCH 2
#define VAE_ERR_UNSUPPORTED 3

// Records a message retrievable through vae_last_error(); returns `code`.
int vae_set_error(int code, const char* fmt, ...);

#define VAE_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return vae_set_error(VAE_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define VAE_CHECK_LAUNCH(name)                                                            \
  do {                                                                                    \
    hipError_t e__ = hipGetLastError();                                                   \
    if (e__ != hipSuccess)                                                                \
      return vae_set_error(VAE_ERR_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline bool vae_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#ifdef __HIPCC__
constexpr int WAVE = 64;

// Operand maxima of the two-f16-term products travel as ARRAYS of partial maxima: every wave (or workgroup) of the
// kernel that produces an operand stores the largest magnitude it wrote into its own entry (plain stores: nothing to
// clear, no atomics, trivially deterministic) and the GEMM that consumes the operand reduces the array in its prologue.
// Wave-level form: every lane passes its candidate, lane 0 stores the wave's maximum.
__device__ __forceinline__ void vae_wave_absmax_store(float* entry, float v) {
  v = fabsf(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) *entry = v;
}
// Slabs of the epilogue backward: [chunk][VAE_SLAB_ROWS][G] = per-chunk column SUMS (rows 0..2: d px_r, d bias_scale,
// d bias_dropout) and per-chunk column MAXIMA of |dlogit_scale|, |dlogit_dropout| (rows 3, 4: the per-gene scales of the
// two-f16-term decoder weight gradient).
#define VAE_SLAB_ROWS 5
// Column g of out_k = sum over `chunks` slabs of slab[c][k][g], in chunk order -- the
// deterministic reduction of the epilogue backward's per-chunk column sums (four loads in flight).
__device__ __forceinline__ void vae_colreduce_column(const float* __restrict__ slab, int chunks, int G, int k, int g,
                                                     float* __restrict__ o, int overwrite) {
  const float* p = slab + (int64_t)k * G + g;
  const int64_t st = (int64_t)VAE_SLAB_ROWS * G;
  float s = 0.f;
  int c = 0;
  for (; c + 16 <= chunks; c += 16) {   // sixteen loads in flight (the chunks are a chain of round trips otherwise); same order
    float q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = p[(c + u) * st];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += q[u];
  }
  for (; c + 4 <= chunks; c += 4) {
    const float a = p[c * st], b = p[(c + 1) * st], d = p[(c + 2) * st], e = p[(c + 3) * st];
    s = (((s + a) + b) + d) + e;
  }
  for (; c < chunks; ++c) s += p[c * st];
  o[g] = overwrite ? s : o[g] + s;
}

// The power of two that puts an operand with maximum magnitude m into [2^13, 2^14): the scale of the two-f16-term
// products (gemm.hip).  1 for m = 0, inf or NaN.
__device__ __forceinline__ float vae_scale_from_absmax(float m) {
  if (!(m > 0.f) || !(m < INFINITY)) return 1.f;
  int e;
  (void)frexpf(m, &e);
  return ldexpf(1.f, 14 - e);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// Wave-wide sum / max on the DPP path: quad butterfly, two rotations inside the 16-lane row, then the four row totals
// through scalar registers -- ~12 short instructions, where __shfl_xor (ds_bpermute) costs an LDS round trip per step
// (a 31-output head loop of shuffled reductions was 9 us of a cell's 28).  Every lane gets the result; fixed order.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_total(float v) {   // lanes 0, 16, 32, 48 hold their row's value
  const int i = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}
// the first half of wave_sum_dpp: every lane ends with the total of its 16-lane row
__device__ __forceinline__ float row_sum_dpp(float v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_move<0x124>(v);   // row_ror:4
  v += dpp_move<0x128>(v);   // row_ror:8
  return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_move<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_move<0x124>(v);   // row_ror:4
  v += dpp_move<0x128>(v);   // row_ror:8
  return row_total(v);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x124>(v));
  v = fmaxf(v, dpp_move<0x128>(v));
  const int i = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}

// Block-wide sum over NW waves; `scratch` holds >= NW floats of LDS. Every thread gets the result.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += scratch[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  float t = scratch[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, scratch[i]);
  return t;
}

// Workgroup `bi` of `nb` (256 threads each) scans its share of n4 16-byte chunks of w and returns max|w| to every
// thread; four 16-byte loads in flight per lane.  `red`: 4 floats of LDS.
__device__ __forceinline__ float vae_absmax_scan(const float* __restrict__ w, int64_t n4, int64_t nb, int64_t bi, float* red) {
  auto m4 = [](const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float m = 0.f;
  const int64_t stride = nb * 256;
  int64_t i = bi * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = w4[i], b = w4[i + stride], c = w4[i + 2 * stride], d = w4[i + 3 * stride];
    m = fmaxf(m, fmaxf(fmaxf(m4(a), m4(b)), fmaxf(m4(c), m4(d))));
  }
  for (; i < n4; i += stride) m = fmaxf(m, m4(w4[i]));
  return block_max<4>(m, red);
}


// exp with the rounding error of v*log2(e) fed back (keeps softmax weights at ~1 ulp for |v| <= 88); -inf -> 0
__device__ __forceinline__ float vae_exp_acc(float v) {
  v = fmaxf(v, -100.f);  // exp underflows to 0 well before; keeps -inf (masked lanes) out of the fma
  const float t = v * 1.44269504088896341f;
  float r = fmaf(v, 1.44269504088896341f, -t);
  r = fmaf(v, 1.925963033500011e-8f, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.69314718055994531f, e);
}

// torch.nn.functional.softplus with beta=1, threshold=20 (distributions.py:42,47; modules.py:269)
__device__ __forceinline__ float softplus20(float v) { return v > 20.f ? v : log1pf(expf(v)); }
// its derivative as torch's softplus_backward evaluates it: z/(z+1), z = exp(v)
__device__ __forceinline__ float softplus20_grad(float v) {
  if (v > 20.f) return 1.f;
  const float z = expf(v);
  return z / (z + 1.f);
}
#endif
