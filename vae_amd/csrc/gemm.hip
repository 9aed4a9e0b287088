// MFMA GEMMs for the VAE hot path (gfx950).
//
// One tiled kernel family covers every contraction of the training step
// (reference: nn.Linear forward / autograd dgrad / wgrad, modules.py:111,248-249,361-369):
//   NT  C[m,n] = sum_k A[m,k] * B[n,k]      (Linear forward:  y = x W^T)
//   NN  C[m,n] = sum_k A[m,k] * B[k,n]      (dgrad:           dx = dy W)
//   TN  C[m,n] = sum_k A[k,m] * B[k,n]      (wgrad:           dW = dy^T x)
// All matrices are fp32 row-major in HBM.  precision 0 = exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32), precision 1 = operands rounded to bf16 while staging into
// LDS, fp32 accumulate (v_mfma_f32_32x32x16_bf16).
//
// The "x operand" (A in NT, B in TN) can be normalised on the fly with the per-cell mean:
// x/mean - 1 (Encoder.forward, modules.py:264), so the normalised count matrix is never
// materialised in HBM.
//
// Tiles: 256-thread workgroups, 4 waves as 2x2.  Large shapes use 128x128x32 (each wave a 64x64
// output = 2x2 MFMA 32x32 accumulators); the many small contractions of the step (latent heads,
// decoder hidden layer: one dimension <= 64) use 64x64x32 (one accumulator per wave) so that
// they spread over enough workgroups.  Global->register prefetch of tile t+1 overlaps the
// MFMAs of tile t; skinny shapes (K = genes or cells) are split over K into fp32 slabs that a
// second kernel sums in a fixed order (deterministic, no float atomics).
#include "common.h"
#include "internal.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>

extern "C" int vae_gemm_bf16_storage_ok(int layout, int io, int64_t M, int64_t N, int64_t K);
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 32;
constexpr int NT = 0, NN = 1, TN = 2;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;            // output, or slab base when split > 1
  const float* bias;   // [N] or null (ignored when split > 1)
  const float* mean;   // per-cell mean for the x operand, or null
  int64_t lda, ldb, ldc;
  int M, N, K;
  int k_per_split;     // multiple of BK
  int split;
  int vecA, vecB;      // 16-byte vector loads allowed
  int vecC;            // 16-byte vector stores allowed (C / the slabs 16-byte aligned, ldc % 4 == 0, N % 4 == 0)
  int accumulate;      // C += (only when split == 1)
  int tiles_m, tiles_n;
  int group_m;         // gemm_x3_kernel: > 0 = row tiles rasterised in groups of this many (see the kernel)
  const float* sa;     // gemm_x3_kernel<NP = 2> (two f16 terms): device pointers to the power-of-two factors the
  const float* sb;     // operands are multiplied by before the split (null = 1); the product is scaled back
  int na, nb;          // > 0: sa / sb are arrays of na / nb partial maxima of |A| / |B| written by the producers of the
                       // operands; the kernel reduces them and derives the factors (0: sa / sb are the factors)
  // gemm_x3_kernel, 16-byte-store epilogue: softmax statistics of the output rows over columns [0, stat_cols), one
  // (max, sum of exp(v - max)) pair per row and per wave-wide column span ("slot" = tile_n * WAVES_N + wave_n), written
  // to stat_m / stat_s [slot][M] -- the decoder-head GEMM hands the likelihood pass its softmax denominators this way
  float* stat_m;
  float* stat_s;
  int stat_cols;
  // gemm_x3_kernel<TN, 128, 128, 2, 2, 0, ROWSC = true>: one power-of-two scale PER ROW m of the product (per column of
  // the k-major A operand) instead of one for all of A, derived from per-chunk column maxima of |A|: max over c < rmax_chunks
  // of (m < rmax_split ? rmax0[c * rmax_stride + m] : rmax1[c * rmax_stride + m - rmax_split])
  const float* rmax0;
  const float* rmax1;
  int rmax_split, rmax_chunks;
  int64_t rmax_stride;
  // gemm_x3_kernel<..., XONE>: the COUNT operand (XONE = 1: A of an NT product, XONE = 2: B of a TN product) is raw counts;
  // integer counts below 2048 are exact in ONE scaled f16 term, so its lo plane, that plane's LDS traffic and one of the
  // three products fall away.  xone_flags: n_xone floats written by the kernel that read the counts (vae_row_mean_ex), any
  // entry > 0 = some count needs its lo term: the kernel then runs the plain two-term loop (a uniform branch, decided
  // once).  row_sub (nullable): C[m, n] -= row_sub[m] in the output stage (the -sum_b dpre term of the weight gradient
  // over x / mean - 1 written as (dpre / mean)^T x - colsum(dpre) 1^T).
  const float* xone_flags;
  int n_xone;
  const float* row_sub;
  // XONE = 3, the OPTIMISTIC count operand (NT, pipelined 128 x 128 tiles): nobody has read the counts yet.  The kernel
  // takes every count as exact in one UNSCALED f16 term and gathers, from the registers it stages anyway, what a statistics
  // pass would have: per-slab row sums xs_sum[slab][M] (the column tiles of a row tile share its staging chunks), and per
  // gathering workgroup an entry of xs_max / xs_flag (maximum of |x|; 1 when some count is NOT exact that way -- then the
  // product is void and the gated launch below repeats it).  Entries past the writers' up to xs_n are zeroed, so the
  // arrays read like the ones vae_row_stats fills.
  float* xs_sum;
  float* xs_max;
  float* xs_flag;
  int xs_n;
  float* ws_sum;   // likewise the per-slab row sums of B, ws_sum[slab][N] (the row tiles share them)
  // XONE = 1 with gate != 0, the launch behind the optimistic one: its first workgroups finish the row means
  // (mean_out[m] = sum over slabs of xs_sum / K) and sum the rows of B (wsum_out[n], nullable); then every workgroup
  // leaves unless some flag is set -- in which case this is the plain count-operand product, scaled by the maxima.
  int gate;
  float* mean_out;
  float* wsum_out;
  int stagger;               // experiment (VAE_GEMM_STAGGER): start-phase offset of co-resident workgroups, units of 1024 cycles
};

// ---- global -> register tile loaders (R*8/256 float4 per thread per operand, R = 64 or 128) ----
// KC: operand stored (rows, K) row-major, k contiguous.  chunk c: row = c>>3, k4 = c&7.
template <bool NORM, int R>
__device__ __forceinline__ void load_kc(const float* __restrict__ X, int64_t ld, int r0, int rmax, int k0, int kend,
                                        int vec, const float* __restrict__ mean, float4 (&reg)[R / 32]) {
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = r0 + (c >> 3), k = k0 + (c & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rmax && k < kend) {
      const float* p = X + (int64_t)r * ld + k;
      const bool m1 = k + 1 < kend, m2 = k + 2 < kend, m3 = k + 3 < kend;
      if (vec && m3) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        v.x = p[0];
        if (m1) v.y = p[1];
        if (m2) v.z = p[2];
        if (m3) v.w = p[3];
      }
      if (NORM) {  // x/mean - 1 as fma(x, 1/mean, -1): one reciprocal per 4 counts
        const float inv = 1.f / mean[r];
        v.x = fmaf(v.x, inv, -1.f);
        v.y = m1 ? fmaf(v.y, inv, -1.f) : 0.f;
        v.z = m2 ? fmaf(v.z, inv, -1.f) : 0.f;
        v.w = m3 ? fmaf(v.w, inv, -1.f) : 0.f;
      }
    }
    reg[i] = v;
  }
}
// RC: operand stored (K, rows) row-major, row index contiguous.  chunk pair q = c / (R/4): k = 2q, 2q+1; r4 = c % (R/4).
template <bool NORM, int R>
__device__ __forceinline__ void load_rc(const float* __restrict__ X, int64_t ld, int r0, int rmax, int k0, int kend,
                                        int vec, const float* __restrict__ mean, float4 (&reg)[R / 32]) {
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int c = threadIdx.x + 256 * (i >> 1);  // registers (2j, 2j+1) hold k = (2q, 2q+1) of the same 4 rows
    const int k = k0 + 2 * (c / (R / 4)) + (i & 1), r = r0 + (c % (R / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend && r < rmax) {
      const float* p = X + (int64_t)k * ld + r;
      const bool m1 = r + 1 < rmax, m2 = r + 2 < rmax, m3 = r + 3 < rmax;
      if (vec && m3) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        v.x = p[0];
        if (m1) v.y = p[1];
        if (m2) v.z = p[2];
        if (m3) v.w = p[3];
      }
      if (NORM) {
        const float inv = 1.f / mean[k];
        v.x = fmaf(v.x, inv, -1.f);
        v.y = m1 ? fmaf(v.y, inv, -1.f) : 0.f;
        v.z = m2 ? fmaf(v.z, inv, -1.f) : 0.f;
        v.w = m3 ? fmaf(v.w, inv, -1.f) : 0.f;
      }
    }
    reg[i] = v;
  }
}

// ---- fp32 LDS image: k-major [BK][LD] floats; operand reads are conflict-free ds_read_b32 ----
template <int R> struct Ld32 {
  static constexpr int KC = R + 2;  // filled by 4-byte scatter writes (2-way on writes: free)
  static constexpr int RC = R + 4;  // filled by 16-byte row writes (needs LD % 4 == 0)
};

template <int R>
__device__ __forceinline__ void store32_kc(float* s, const float4 (&reg)[R / 32]) {
  constexpr int LD = Ld32<R>::KC;
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = c >> 3, k = (c & 7) * 4;
    s[(k + 0) * LD + r] = reg[i].x;
    s[(k + 1) * LD + r] = reg[i].y;
    s[(k + 2) * LD + r] = reg[i].z;
    s[(k + 3) * LD + r] = reg[i].w;
  }
}
template <int R>
__device__ __forceinline__ void store32_rc(float* s, const float4 (&reg)[R / 32]) {
  constexpr int LD = Ld32<R>::RC;
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int c = threadIdx.x + 256 * (i >> 1);
    const int k = 2 * (c / (R / 4)) + (i & 1), r = (c % (R / 4)) * 4;
    *reinterpret_cast<float4*>(&s[k * LD + r]) = reg[i];
  }
}

// ---- bf16 LDS image: row-major [R][LD16] bf16, k contiguous; 80-byte rows make the
//      ds_read_b128 fragment reads conflict-free (5*row mod 16 is a bijection on 16 rows) ----
constexpr int LD16 = 40;

template <int R>
__device__ __forceinline__ void store16_kc(__bf16* s, const float4 (&reg)[R / 32]) {
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = c >> 3, k = (c & 7) * 4;
    bf16x4 v;
    v[0] = (__bf16)reg[i].x; v[1] = (__bf16)reg[i].y; v[2] = (__bf16)reg[i].z; v[3] = (__bf16)reg[i].w;
    *reinterpret_cast<bf16x4*>(&s[r * LD16 + k]) = v;
  }
}
template <int R>
__device__ __forceinline__ void store16_rc(__bf16* s, const float4 (&reg)[R / 32]) {
  // transposing store: each register pair carries (k, k+1) for 4 rows -> one packed 4-byte write per row
#pragma unroll
  for (int i = 0; i < R / 32; i += 2) {
    const int c = threadIdx.x + 256 * (i >> 1);
    const int k = 2 * (c / (R / 4)), r = (c % (R / 4)) * 4;
    const float lo[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    const float hi[4] = {reg[i + 1].x, reg[i + 1].y, reg[i + 1].z, reg[i + 1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x2 v;
      v[0] = (__bf16)lo[j];
      v[1] = (__bf16)hi[j];
      *reinterpret_cast<bf16x2*>(&s[(r + j) * LD16 + k]) = v;
    }
  }
}

// ---- bf16x3 LDS image: three bf16 planes (hi, mid, lo) of the same [R][LD16] layout.  An fp32
//      value is split EXACTLY into hi + mid + lo (8 + 8 + 8 mantissa bits, each remainder is
//      representable), so hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid reproduces the fp32
//      product to 2^-25: six bf16 MFMAs (16x the fp32 MFMA rate each) replace eight fp32 MFMAs. ----
__device__ __forceinline__ void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
  // hi is rounded to nearest (|a - hi| <= 2^-9 |a|, sign random); mid and lo are the next two
  // 8-bit slices by truncation (bit masks run at the full VALU rate, conversions do not):
  // every remainder is exact, and the dropped cross terms stay below 2^-24 of the product
  h = (__bf16)a;
  float r = a - (float)h;
  const unsigned rb = __float_as_uint(r) & 0xffff0000u;
  m = __builtin_bit_cast(__bf16, (unsigned short)(rb >> 16));
  r = r - __uint_as_float(rb);
  l = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(r) >> 16));
}
template <int R>
__device__ __forceinline__ void store16x3_kc(__bf16* s, int plane, const float4 (&reg)[R / 32]) {
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = c >> 3, k = (c & 7) * 4;
    const float v[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __bf16 a, b, d;
      split3(v[j], a, b, d);
      h[j] = a; m[j] = b; l[j] = d;
    }
    *reinterpret_cast<bf16x4*>(&s[r * LD16 + k]) = h;
    *reinterpret_cast<bf16x4*>(&s[plane + r * LD16 + k]) = m;
    *reinterpret_cast<bf16x4*>(&s[2 * plane + r * LD16 + k]) = l;
  }
}
template <int R>
__device__ __forceinline__ void store16x3_rc(__bf16* s, int plane, const float4 (&reg)[R / 32]) {
#pragma unroll
  for (int i = 0; i < R / 32; i += 2) {
    const int c = threadIdx.x + 256 * (i >> 1);
    const int k = 2 * (c / (R / 4)), r = (c % (R / 4)) * 4;
    const float lo[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
    const float hi[4] = {reg[i + 1].x, reg[i + 1].y, reg[i + 1].z, reg[i + 1].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x2 h, m, l;
      __bf16 a, b, d;
      split3(lo[j], a, b, d);
      h[0] = a; m[0] = b; l[0] = d;
      split3(hi[j], a, b, d);
      h[1] = a; m[1] = b; l[1] = d;
      *reinterpret_cast<bf16x2*>(&s[(r + j) * LD16 + k]) = h;
      *reinterpret_cast<bf16x2*>(&s[plane + (r + j) * LD16 + k]) = m;
      *reinterpret_cast<bf16x2*>(&s[2 * plane + (r + j) * LD16 + k]) = l;
    }
  }
}

// TW = MFMA 32x32 accumulators per wave per dimension: 2 -> 128x128 tile, 1 -> 64x64 tile
template <int LAYOUT, int PREC, bool NORM, int TW>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  constexpr int R = 64 * TW;  // tile rows (= BM = BN)
  constexpr int NR = R / 32;  // float4 registers per thread per operand
  constexpr bool A_KC = (LAYOUT != TN);
  constexpr bool B_KC = (LAYOUT == NT);
  constexpr bool NORM_A = NORM && LAYOUT == NT;
  constexpr bool NORM_B = NORM && LAYOUT == TN;
  constexpr int SMEM32 = BK * Ld32<R>::RC;  // floats per operand (RC >= KC)
  constexpr int SMEM16 = R * LD16;          // bf16 per operand

  __shared__ __attribute__((aligned(16))) float smem[PREC == 0 ? 2 * SMEM32 : (PREC == 1 ? SMEM16 : 3 * SMEM16)];
  float* sA32 = smem;
  float* sB32 = smem + SMEM32;
  __bf16* sA16 = reinterpret_cast<__bf16*>(smem);
  __bf16* sB16 = sA16 + (PREC == 2 ? 3 * SMEM16 : SMEM16);

  const int tile = blockIdx.x;
  const int m0 = (tile / g.tiles_n) * R, n0 = (tile % g.tiles_n) * R;
  const int s = blockIdx.y;
  const int kbeg = s * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one register stage: tile t+1 is requested while tile t is multiplied (a second stage costs
  // ~70 VGPRs = one resident workgroup per CU, which measured slower than the latency it hides)
  float4 ra0[NR], rb0[NR];
  auto fetch = [&](int k0, float4 (&ra)[NR], float4 (&rb)[NR]) {
    if (A_KC) load_kc<NORM_A, R>(g.A, g.lda, m0, g.M, k0, kend, g.vecA, g.mean, ra);
    else      load_rc<false, R>(g.A, g.lda, m0, g.M, k0, kend, g.vecA, nullptr, ra);
    if (B_KC) load_kc<false, R>(g.B, g.ldb, n0, g.N, k0, kend, g.vecB, nullptr, rb);
    else      load_rc<NORM_B, R>(g.B, g.ldb, n0, g.N, k0, kend, g.vecB, g.mean, rb);
  };
  auto stage = [&](const float4 (&ra)[NR], const float4 (&rb)[NR]) {
    __syncthreads();  // previous tile's fragment reads are done
    if (PREC == 0) {
      if (A_KC) store32_kc<R>(sA32, ra); else store32_rc<R>(sA32, ra);
      if (B_KC) store32_kc<R>(sB32, rb); else store32_rc<R>(sB32, rb);
    } else if (PREC == 1) {
      if (A_KC) store16_kc<R>(sA16, ra); else store16_rc<R>(sA16, ra);
      if (B_KC) store16_kc<R>(sB16, rb); else store16_rc<R>(sB16, rb);
    } else {
      if (A_KC) store16x3_kc<R>(sA16, SMEM16, ra); else store16x3_rc<R>(sA16, SMEM16, ra);
      if (B_KC) store16x3_kc<R>(sB16, SMEM16, rb); else store16x3_rc<R>(sB16, SMEM16, rb);
    }
    __syncthreads();
  };
  auto multiply = [&]() {
    if (PREC == 0) {
      constexpr int LDA = A_KC ? Ld32<R>::KC : Ld32<R>::RC;
      constexpr int LDB = B_KC ? Ld32<R>::KC : Ld32<R>::RC;
      const float* pa = sA32 + h * LDA + wm * (32 * TW) + l31;
      const float* pb = sB32 + h * LDB + wn * (32 * TW) + l31;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[TW], b[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          a[t] = pa[kk * 2 * LDA + 32 * t];
          b[t] = pb[kk * 2 * LDB + 32 * t];
        }
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int j = 0; j < TW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else if (PREC == 2) {
      const __bf16* pa = sA16 + (wm * (32 * TW) + l31) * LD16 + h * 8;
      const __bf16* pb = sB16 + (wn * (32 * TW) + l31) * LD16 + h * 8;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[TW][3], b[TW][3];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            a[t][q] = *reinterpret_cast<const bf16x8*>(pa + q * SMEM16 + 32 * t * LD16 + ks * 16);
            b[t][q] = *reinterpret_cast<const bf16x8*>(pb + q * SMEM16 + 32 * t * LD16 + ks * 16);
          }
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int j = 0; j < TW; ++j) {
            f32x16 c = acc[i][j];  // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
            acc[i][j] = c;
          }
      }
    } else {
      const __bf16* pa = sA16 + (wm * (32 * TW) + l31) * LD16 + h * 8;
      const __bf16* pb = sB16 + (wn * (32 * TW) + l31) * LD16 + h * 8;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[TW], b[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          a[t] = *reinterpret_cast<const bf16x8*>(pa + 32 * t * LD16 + ks * 16);
          b[t] = *reinterpret_cast<const bf16x8*>(pb + 32 * t * LD16 + ks * 16);
        }
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int j = 0; j < TW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (kbeg < kend) fetch(kbeg, ra0, rb0);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    stage(ra0, rb0);
    if (k0 + BK < kend) fetch(k0 + BK, ra0, rb0);  // in flight during the MFMAs below
    multiply();
  }

  // C/D map of the 32x32 MFMAs: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = g.C;
  int64_t ldc = g.ldc;
  if (g.split > 1) {
    out = g.C + (int64_t)s * g.M * g.N;
    ldc = g.N;
  }
#pragma unroll
  for (int mi = 0; mi < TW; ++mi)
#pragma unroll
    for (int ni = 0; ni < TW; ++ni) {
      const int n = n0 + wn * (32 * TW) + ni * 32 + l31;
      if (n >= g.N) continue;
      const float bv = (g.split == 1 && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (32 * TW) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M) {
          float* p = out + (int64_t)m * ldc + n;
          float v = acc[mi][ni][r] + bv;
          if (g.split == 1 && g.accumulate) v += *p;
          *p = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// fp32 pipeline with LDS-DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write
// pass, two LDS stages, one barrier per K-tile.  Tile 128x128x32, 4 waves as 2x2.
//   KC operand image: [128 rows][8 x 16 B], 16-byte slot = chunk ^ ((row >> 1) & 7) -- the swizzle
//     is applied to the SOURCE address (the DMA destination is lane-linear) and again on the read;
//     fragments are ds_read_b128 (4 k per lane), conflict-free over the b128 lane groups.
//   RC operand image: [32 k][128 rows], fragments are ds_read_b32.
// The contraction order inside a K-tile is permuted (step j, lane half h -> k = 8(j/4)+4h+(j%4))
// identically for A and B, so a 16-byte fragment feeds four consecutive MFMAs.
// Requirements (checked on the host; otherwise the register-staged kernel above runs):
// K % 32 == 0, 16-byte aligned rows, RC extents % 4 == 0.  Rows beyond M / N are clamped to a
// valid row on the load side (their products are never stored).
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

// One wave-instruction: lane i copies 16 bytes from its own `src` to LDS byte `lds_wave_base + 16 i`.
// Issued through inline asm so that the compiler does not count it: hipcc would otherwise drain
// vmcnt before the next ds_read (it cannot tell which LDS stage the DMA writes).  The kernel waits
// for the DMA itself (dma_wait) before the barrier that publishes a stage.
__device__ __forceinline__ void dma16(const float* src, uint32_t lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds_wave_base)
               : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// BM x BN = 128x128 (waves 2x2, 2x2 accumulators each), 128x160 (waves 4x1, 1x5) or 160x128
// (waves 1x4, 5x1): the 160-wide shapes exist so that the tile count of the 40000-row decoder
// head GEMMs lands just under a multiple of the 512 resident workgroups.
template <int LAYOUT, int BM, int BN, int WAVES_M>
__global__ __launch_bounds__(256, 2) void gemm32_dma_kernel(GemmArgs g) {
  constexpr bool A_KC = (LAYOUT != TN);
  constexpr bool B_KC = (LAYOUT == NT);
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = BM / (32 * WAVES_M), TNN = BN / (32 * WAVES_N);  // accumulators per wave
  constexpr int PA = BM / 32, PB = BN / 32;                           // DMA pieces per wave
  constexpr int OPA = 32 * BM, OPB = 32 * BN;                         // floats per operand per stage
  constexpr int STAGE = OPA + OPB;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];  // 64-72 KB: two workgroups per CU

  // XCD-aware placement: consecutive logical ids share an XCD (hardware deals blockIdx round-robin
  // over the 8 XCDs), and the tiles that share the large operand are consecutive.
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tiles_m = g.tiles_m, tiles_n = g.tiles_n;
  const int per_split = tiles_m * tiles_n;
  const int s = id / per_split, t = id - s * per_split;
  int tm, tn;
  if (tiles_m <= tiles_n) { tm = t % tiles_m; tn = t / tiles_m; }
  else                    { tn = t % tiles_n; tm = t / tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = s * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- per-lane DMA source pointers: one 1 KB piece per wave-instruction ----
  const float* pa[PA];
  const float* pb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int piece = wave * PA + i;
    if (A_KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c8 = (lane & 7) ^ ((row >> 1) & 7);
      pa[i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + kbeg + 4 * c8;
    } else {
      const int c = piece * 64 + lane, k = c / (BM / 4), r4 = c % (BM / 4);
      pa[i] = g.A + (int64_t)(kbeg + k) * g.lda + min(m0 + 4 * r4, g.M - 4);
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int piece = wave * PB + i;
    if (B_KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c8 = (lane & 7) ^ ((row >> 1) & 7);
      pb[i] = g.B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + kbeg + 4 * c8;
    } else {
      const int c = piece * 64 + lane, k = c / (BN / 4), r4 = c % (BN / 4);
      pb[i] = g.B + (int64_t)(kbeg + k) * g.ldb + min(n0 + 4 * r4, g.N - 4);
    }
  }
  const int64_t step_a = A_KC ? 32 : 32 * g.lda;
  const int64_t step_b = B_KC ? 32 : 32 * g.ldb;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)smem;
  auto issue = [&](int stage) {  // stage index is wave-uniform
    const uint32_t base_a = lds0 + (uint32_t)stage * (STAGE * 4u) + (uint32_t)wave * (PA * 1024u);
    const uint32_t base_b = lds0 + (uint32_t)stage * (STAGE * 4u) + OPA * 4u + (uint32_t)wave * (PB * 1024u);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      dma16(pa[i], base_a + i * 1024u);
      pa[i] += step_a;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      dma16(pb[i], base_b + i * 1024u);
      pb[i] += step_b;
    }
  };

  // ---- fragment addressing ----
  const int ra = wm * (32 * TM) + l31, rb = wn * (32 * TNN) + l31;  // first 32-row block of the wave
  const int ea = (h ^ ((ra >> 1) & 7)), eb = (h ^ ((rb >> 1) & 7));
  const int kc_a = ra * 32, kc_b = rb * 32;                    // KC: + t*1024 + (((2p)^e) << 2)
  const int rc_a = 4 * h * BM + ra, rc_b = 4 * h * BN + rb;    // RC: + (8p+i)*BM + 32t

  f32x16 acc[TM][TNN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto frag_a = [&](const float* sx, int p, float (&f)[TM][4]) {
#pragma unroll
    for (int tt = 0; tt < TM; ++tt) {
      if (A_KC) {
        const float4 v = *reinterpret_cast<const float4*>(sx + kc_a + tt * 1024 + (((2 * p) ^ ea) << 2));
        f[tt][0] = v.x; f[tt][1] = v.y; f[tt][2] = v.z; f[tt][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[tt][i] = sx[rc_a + (8 * p + i) * BM + 32 * tt];
      }
    }
  };
  auto frag_b = [&](const float* sx, int p, float (&f)[TNN][4]) {
#pragma unroll
    for (int tt = 0; tt < TNN; ++tt) {
      if (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(sx + kc_b + tt * 1024 + (((2 * p) ^ eb) << 2));
        f[tt][0] = v.x; f[tt][1] = v.y; f[tt][2] = v.z; f[tt][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[tt][i] = sx[rc_b + (8 * p + i) * BN + 32 * tt];
      }
    }
  };

  auto multiply = [&](const float* stage) {
    const float* sa = stage;
    const float* sb = stage + OPA;
    float a[2][TM][4], b[2][TNN][4];  // [buffer][32-row block][k]
    frag_a(sa, 0, a[0]);
    frag_b(sb, 0, b[0]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int cur = p & 1;
      if (p + 1 < 4) {  // next group's fragments are in flight during this group's MFMAs
        frag_a(sa, p + 1, a[cur ^ 1]);
        frag_b(sb, p + 1, b[cur ^ 1]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
          for (int tj = 0; tj < TNN; ++tj)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ti][i], b[cur][tj][i], acc[ti][tj], 0, 0, 0);
    }
  };

  if (kbeg < kend) issue(0);
  dma_wait();
  __syncthreads();  // stage 0 published
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if (k0 + BK < kend) issue(cur ^ 1);  // lands during the MFMAs below
    multiply(smem + cur * STAGE);
    dma_wait();
    __syncthreads();  // next stage published; every wave is done reading this one
    cur ^= 1;
  }

  float* out = g.C;
  int64_t ldc = g.ldc;
  if (g.split > 1) {
    out = g.C + (int64_t)s * g.M * g.N;
    ldc = g.N;
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNN; ++ni) {
      const int n = n0 + wn * (32 * TNN) + ni * 32 + l31;
      if (n >= g.N) continue;
      const float bv = (g.split == 1 && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (32 * TM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M) {
          float* p = out + (int64_t)m * ldc + n;
          float v = acc[mi][ni][r] + bv;
          if (g.split == 1 && g.accumulate) v += *p;
          *p = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores: every fp32 operand element is split EXACTLY into three
// bf16 values (split3 above) while it is staged into LDS, and each fragment pair runs the six
// significant products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid (dropped terms < 2^-25 of
// the product) with fp32 accumulation.  Measured against fp64 the result is as accurate as the
// fp32 MFMA path (tools/acc_gemm.py), at 6/16 of its matrix-core time.
//   KC operand planes: [R][40] bf16 (80-byte rows), fragments by ds_read_b128.
//   RC operand planes: [32 k][160] bf16 (320-byte rows), fragments by the transposing
//     ds_read_b64_tr_b16 (a 16-lane group reads 4 k x 16 rows and receives them row-major).
// Same shape requirements, tiles and placement as gemm32_dma_kernel.
// ---------------------------------------------------------------------------------------
#ifdef VAE_PROBE   // tools/micro/gemm_probe.hip: cycle stamps of workgroup VAE_PROBE's K loop
__device__ long long g_probe[4][256];
#define GSTAMP() do { if (blockIdx.x == VAE_PROBE && lane == 0 && pi < 256) g_probe[wave][pi++] = clock64(); } while (0)
#else
#define GSTAMP() do { } while (0)
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// Output stage shared by the split-precision kernel and the bf16 LDS-DMA kernel below: accumulators (+ bias, scaled back)
// -> C or the split-K slab, as fp32 or bf16, with the softmax statistics of the rows when asked for.  `smem` is the kernel's
// LDS (operand storage, dead by now; at least 4 * 32 * 40 floats when PATCHED); every thread of the workgroup calls it.
template <int LAYOUT, int BM, int BN, int WAVES_M, int NP, bool CBF, bool PATCHED>
__device__ __forceinline__ void x3_epilogue(const GemmArgs& g, f32x16 (&acc)[BM / (32 * WAVES_M)][BN / (32 * (4 / WAVES_M))],
                                            void* smem, int m0, int n0, int tn, int s, float scA, float scB,
                                            const float* row_inv = nullptr,     // LDS: 1 / (row scale) of the tile's BM rows
                                            float rs_pref = 0.f) {              // thread tid < BM: row_sub[m0 + tid], loaded at kernel start
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = BM / (32 * WAVES_M), TNN = BN / (32 * WAVES_N);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, h = lane >> 5;
  const float inv_sc = 1.f / (scA * scB);   // powers of two: exact
  float* out = g.C;
  int64_t ldc = g.ldc;
  if (g.split > 1) {
    out = g.C + (int64_t)s * g.M * g.N;
    ldc = g.N;
  }
  // 16-byte stores: every 32 x 32 accumulator block goes through a private 32 x 40-float LDS patch of its wave (the
  // operand planes are dead by now) and leaves as 4 instructions of 8 rows x 128 B instead of 16 of 2 rows x 128 B --
  // a quarter of the vector-memory instructions against the 63-deep in-order queue of the wave.
  const bool vec_out = PATCHED && g.vecC && !(g.split == 1 && g.accumulate);
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 40);
  // the bias of every column block of this wave, requested before the barrier: a load inside the block loop is a
  // dependent L2 round trip per block
  float bvs[TNN];
#pragma unroll
  for (int ni = 0; ni < TNN; ++ni) {
    const int n = n0 + wn * (32 * TNN) + ni * 32 + l31;
    bvs[ni] = (g.split == 1 && g.bias && n < g.N) ? g.bias[n] : 0.f;
  }
  if (vec_out) __syncthreads();   // every wave is done with the operand planes
  // row_sub of this tile's rows, parked behind the waves' patches (vector path; the element path reads it from memory)
  float* rsub = reinterpret_cast<float*>(smem) + 4 * (32 * 40);
  const bool sub = NP == 2 && g.row_sub != nullptr && g.split == 1;   // (NP == 2: the LDS behind the patches exists)
  if (vec_out && sub) {
    if (tid < BM) rsub[tid] = rs_pref;
    __syncthreads();
  }
  // running softmax statistics of this wave's rows over its column span: lane (8 r' + c) holds rows r' + 8 i, i < 4
  constexpr bool STAT_OK = LAYOUT == NT && TM <= 2;         // tile shapes whose epilogue carries the statistics registers
  const bool stats = STAT_OK && g.stat_m != nullptr;        // the host enables it only when every block takes the path below
  float rmax[STAT_OK ? TM : 1][4], rsum[STAT_OK ? TM : 1][4];
#pragma unroll
  for (int mi = 0; mi < (STAT_OK ? TM : 1); ++mi)
#pragma unroll
    for (int i = 0; i < 4; ++i) { rmax[mi][i] = -INFINITY; rsum[mi][i] = 0.f; }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TNN; ++ni) {
      const int nb = n0 + wn * (32 * TNN) + ni * 32;
      const int n = nb + l31;
      const float bv = bvs[ni];
      if (CBF && vec_out && nb + 32 <= g.N) {
        // bf16 output: the 32 x 32 block leaves as 2 instructions of 16 rows x 64 B (8 bf16 per lane)
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * 40 + l31] = acc[mi][ni][r] + bv;
        uint16_t* out16 = reinterpret_cast<uint16_t*>(out);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (lane >> 2) + 16 * i;
          const int m = m0 + wm * (32 * TM) + mi * 32 + row;
          const float4 v0 = *reinterpret_cast<const float4*>(patch + row * 40 + 8 * (lane & 3));
          const float4 v1 = *reinterpret_cast<const float4*>(patch + row * 40 + 8 * (lane & 3) + 4);
          bf16x8 o;
          o[0] = (__bf16)v0.x; o[1] = (__bf16)v0.y; o[2] = (__bf16)v0.z; o[3] = (__bf16)v0.w;
          o[4] = (__bf16)v1.x; o[5] = (__bf16)v1.y; o[6] = (__bf16)v1.z; o[7] = (__bf16)v1.w;
          if (m < g.M) *reinterpret_cast<bf16x8*>(out16 + (int64_t)m * ldc + nb + 8 * (lane & 3)) = o;
          if (STAT_OK && stats && nb < g.stat_cols) {   // as below, over the 8 ROUNDED values this lane stores
            constexpr int MI = STAT_OK ? 1 : 0;
            constexpr float L2E = 1.44269504088896341f;
            const int c0 = nb + 8 * (lane & 3);
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = c0 + j < g.stat_cols ? (float)o[j] : -INFINITY;
            const float om = rmax[mi * MI][i];
            const float nm = fmaxf(fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]))), om);
            const float nb2 = -nm * L2E;
            if (nm > -INFINITY) {
              float e = 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) e += __builtin_amdgcn_exp2f(fmaf(a[j], L2E, nb2));
              const float rescale = om == nm ? 1.f : __builtin_amdgcn_exp2f(fmaxf(fmaf(om, L2E, nb2), -126.f));
              rsum[mi * MI][i] = fmaf(rsum[mi * MI][i], rescale, e);
              rmax[mi * MI][i] = nm;
            }
          }
        }
        continue;
      }
      if (CBF) {   // ragged edge block of a bf16 output: element stores
        if (n >= g.N) continue;
        uint16_t* out16 = reinterpret_cast<uint16_t*>(out);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (32 * TM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < g.M) out16[(int64_t)m * ldc + n] = __builtin_bit_cast(uint16_t, (__bf16)(acc[mi][ni][r] + bv));
        }
        continue;
      }
      if (vec_out && nb + 32 <= g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float is = row_inv ? inv_sc * row_inv[wm * (32 * TM) + mi * 32 + rl] : inv_sc;
          patch[rl * 40 + l31] = (NP == 2 ? acc[mi][ni][r] * is : acc[mi][ni][r]) + (sub ? bv - rsub[wm * (32 * TM) + mi * 32 + rl] : bv);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (lane >> 3) + 8 * i;
          const int m = m0 + wm * (32 * TM) + mi * 32 + row;
          const float4 v = *reinterpret_cast<const float4*>(patch + row * 40 + 4 * (lane & 7));
          if (m < g.M) *reinterpret_cast<float4*>(out + (int64_t)m * ldc + nb + 4 * (lane & 7)) = v;
          if (STAT_OK && stats && nb < g.stat_cols) {
            // lane-local running (max, sum of exp) over the columns THIS lane stores: no cross-lane traffic in the loop.
            // Plain exp2 (no compensation): the terms that matter sit within a few units of the maximum, where the
            // argument rounding is < 1e-7 relative; the softmax WEIGHTS downstream keep the compensated form.
            constexpr int MI = STAT_OK ? 1 : 0;   // (index 0 when the registers are not carried)
            constexpr float L2E = 1.44269504088896341f;
            const int c0 = nb + 4 * (lane & 7);
            float a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
            if (nb + 32 > g.stat_cols) {   // (uniform) only the block that straddles the last statistics column masks
              a0 = c0 < g.stat_cols ? v.x : -INFINITY; a1 = c0 + 1 < g.stat_cols ? v.y : -INFINITY;
              a2 = c0 + 2 < g.stat_cols ? v.z : -INFINITY; a3 = c0 + 3 < g.stat_cols ? v.w : -INFINITY;
            }
            const float om = rmax[mi * MI][i];
            const float nm = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), om);
            const float nb2 = -nm * L2E;   // nm is finite unless every column so far was masked (then the terms are exp2(nan)?)
            if (nm > -INFINITY) {
              const float e = (__builtin_amdgcn_exp2f(fmaf(a0, L2E, nb2)) + __builtin_amdgcn_exp2f(fmaf(a1, L2E, nb2))) +
                              (__builtin_amdgcn_exp2f(fmaf(a2, L2E, nb2)) + __builtin_amdgcn_exp2f(fmaf(a3, L2E, nb2)));
              const float rescale = om == nm ? 1.f : __builtin_amdgcn_exp2f(fmaxf(fmaf(om, L2E, nb2), -126.f));
              rsum[mi * MI][i] = fmaf(rsum[mi * MI][i], rescale, e);
              rmax[mi * MI][i] = nm;
            }
          }
        }
        continue;
      }
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (32 * TM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M) {
          float* p = out + (int64_t)m * ldc + n;
          const float is = row_inv ? inv_sc * row_inv[m - m0] : inv_sc;
          float v = (NP == 2 ? acc[mi][ni][r] * is : acc[mi][ni][r]) + (sub ? bv - g.row_sub[m] : bv);
          if (g.split == 1 && g.accumulate) v += *p;
          *p = v;
        }
      }
    }
  if (STAT_OK && stats && n0 + wn * (32 * TNN) < g.stat_cols) {   // (uniform per wave: every lane takes part in the shuffles)
    const int slot = tn * WAVES_N + wn;
#pragma unroll
    for (int mi = 0; mi < (STAT_OK ? TM : 1); ++mi)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // merge the 8 (bf16 output: 4) lanes that share this row: max, rescale, sum
        if (CBF && i >= 2) continue;
        float M = rmax[mi][i];
        M = fmaxf(M, __shfl_xor(M, 1, 64));
        M = fmaxf(M, __shfl_xor(M, 2, 64));
        if (!CBF) M = fmaxf(M, __shfl_xor(M, 4, 64));
        float sv = rmax[mi][i] > -INFINITY ? rsum[mi][i] * vae_exp_acc(rmax[mi][i] - M) : 0.f;
        sv += __shfl_xor(sv, 1, 64);
        sv += __shfl_xor(sv, 2, 64);
        if (!CBF) sv += __shfl_xor(sv, 4, 64);
        const int m = m0 + wm * (32 * TM) + mi * 32 + (CBF ? (lane >> 2) + 16 * i : (lane >> 3) + 8 * i);
        if ((lane & (CBF ? 3 : 7)) == 0 && m < g.M) {
          g.stat_m[(int64_t)slot * g.M + m] = M;
          g.stat_s[(int64_t)slot * g.M + m] = sv;
        }
      }
  }
}

// Two elements -> their packed f16 hi terms and (WITH_LO) packed f16 lo terms: hi = rn16(a s), lo = rn16(a s - hi).  s is a
// power of two and a s - hi is exact in fp32, so the mixed-precision FMA (fp32 arithmetic, f16 result in one half of the
// destination, the f16 hi term read straight from its half) gives the bits of the plain C expression -- in 4 instructions
// per pair where the compiler spends 7-8 (it builds the packed hi through separate multiplies and a pack conversion).
#ifndef VAE_X3_MIXCVT
#define VAE_X3_MIXCVT 1
#endif
// p[0] + p[stride] + ... (n terms, in order): sixteen loads in flight instead of a chain of round trips
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, int n, int64_t stride) {
  float v = 0.f;
  int i = 0;
  for (; i + 16 <= n; i += 16) {
    float q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = p[(i + u) * stride];
#pragma unroll
    for (int u = 0; u < 16; ++u) v += q[u];
  }
  for (; i + 4 <= n; i += 4) {
    const float a = p[i * stride], b = p[(i + 1) * stride], c = p[(i + 2) * stride], d = p[(i + 3) * stride];
    v = (((v + a) + b) + c) + d;
  }
  for (; i < n; ++i) v += p[i * stride];
  return v;
}
template <typename WITH_LO>
__device__ __forceinline__ void split_pair_f16(float a0, float a1, float s0, float s1, uint32_t& hi, uint32_t& lo, WITH_LO) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a0), "v"(s0));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(a1), "v"(s1));
  if constexpr (WITH_LO::value) {
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a0), "v"(s0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(a1), "v"(s1), "v"(hi));
  } else {
    lo = 0u;
  }
}

// IO (NP == 1 only): storage type of the operands / the output in HBM.  bit 0: A holds bf16, bit 1: B holds bf16 (staged
// by 8-byte loads straight into the operand plane -- no conversion, half the bytes), bit 2: C is written as bf16 (rounded
// to nearest even; the softmax statistics are then taken from the ROUNDED values, the ones the consumer will read).
// Leading dimensions count elements of the respective type.
// PIPE (NP <= 2): software-pipelined K loop.  The workgroups resident on a CU run the same code from the same start and
// stay in phase: all of them convert (VALU) at the same time, then all of them multiply (matrix cores) at the same
// time, so the two never overlap (tools/micro/gemm_probe.hip: the phases ADD).  Here the overlap is built into each
// wave: two LDS stages and two sets of staging registers; while the MFMAs of K step i run from stage i & 1, the same
// wave converts step i + 1 (loaded one iteration earlier) into the other stage between them, and the loads of step
// i + 2 are in flight -- one barrier per K step instead of two, a whole iteration of slack for the global loads.
#ifndef VAE_X3_PIPE
#define VAE_X3_PIPE 1
#endif
#ifndef VAE_X3_SCHEDB
#define VAE_X3_SCHEDB 0
#endif
#ifndef VAE_X3_RELOAD
#define VAE_X3_RELOAD 1
#endif
#ifndef VAE_X3_SETS
#define VAE_X3_SETS 1
#endif
#ifndef VAE_STAT_INTERLEAVE   // decoder forward: statistics and plain column tiles alternate in launch order
#define VAE_STAT_INTERLEAVE 1
#endif
#ifndef VAE_ROWSC_BATCH32
#define VAE_ROWSC_BATCH32 1
#endif
#ifndef VAE_TN_XONE   // one-term count operand in the TN weight gradient too (128 x 128 tiles)
#define VAE_TN_XONE 1
#endif
#ifndef VAE_X3_RELOAD_FLAT   // the same in the unpipelined loop (the 160-wide tiles), NT products
#define VAE_X3_RELOAD_FLAT 1
#endif
template <int LAYOUT, int BM, int BN, int WAVES_M, int NP, int IO = 0, bool ROWSC = false, int XONE = 0>
__global__ __launch_bounds__(256, ((VAE_X3_PIPE && NP <= 2 && BM == 128 && BN == 128) || ROWSC || XONE != 0 || BN > 160 ? 2 : (NP <= 2 ? 3 : 2))) void gemm_x3_kernel(GemmArgs g) {
  static_assert((IO & 7) == 0 || NP == 1, "bf16 storage goes with the one-plane bf16 product");
  static_assert((IO & 24) == 0 || (NP == 2 && (IO & 7) == 0 && ((IO == 8 && LAYOUT == NT) || (IO == 16 && LAYOUT == TN))),
                "uint8 counts: the A operand of the NT or the B operand of the TN two-f16-term product");
  static_assert(!ROWSC || (LAYOUT == TN && NP == 2), "per-row scales: the two-f16-term wgrad");
  static_assert(XONE == 0 || (NP == 2 && !ROWSC && (((XONE == 1 || XONE == 3) && LAYOUT == NT) || (XONE == 2 && LAYOUT == TN))),
                "one-term count operand: A of the two-f16-term NT product or B of the TN product");
  static_assert(XONE != 3 || (VAE_X3_PIPE && VAE_X3_RELOAD && BM == 128 && BN == 128), "optimistic counts: the pipelined 128 x 128 loop");
  constexpr bool XOPT = XONE == 3;
  __shared__ __attribute__((aligned(16))) float row_sc[ROWSC ? BM : 4];
  __shared__ float row_inv[ROWSC ? BM : 4];
  constexpr bool PIPE = VAE_X3_PIPE && NP <= 2 && BM == 128 && BN == 128;   // the 160-wide tiles do not fit two waves per SIMD
  constexpr int STAGES = PIPE ? 2 : 1;
  constexpr bool ABF = (IO & 1) != 0, BBF = (IO & 2) != 0, CBF = (IO & 4) != 0;
  // IO bits 8 / 16: the count operand (A of NT / B of TN) is the uint8 store itself (create_dataset.py:131-142): four counts
  // per 4-byte load, expanded in registers on their way into the f16 plane -- no fp32 copy of the batch exists
  constexpr bool AU8 = (IO & 8) != 0, BU8 = (IO & 16) != 0;
  using TA = std::conditional_t<AU8, uint8_t, std::conditional_t<ABF, uint16_t, float>>;
  using TB = std::conditional_t<BU8, uint8_t, std::conditional_t<BBF, uint16_t, float>>;
  using RA = std::conditional_t<AU8, uint32_t, std::conditional_t<ABF, uint2, float4>>;   // four consecutive elements
  using RB = std::conditional_t<BU8, uint32_t, std::conditional_t<BBF, uint2, float4>>;
  constexpr bool A_KC = (LAYOUT != TN);
  constexpr bool B_KC = (LAYOUT == NT);
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = BM / (32 * WAVES_M), TNN = BN / (32 * WAVES_N);
  constexpr int NA = BM / 32, NB = BN / 32;                 // float4 per thread per operand
  constexpr int LDK = 40, LDR = 160;                        // bf16 row strides of the two images
  constexpr int PLA = A_KC ? BM * LDK : 32 * LDR;           // bf16 per plane
  constexpr int PLB = B_KC ? BN * LDK : 32 * LDR;
  constexpr int STG = NP * (PLA + PLB);                     // bf16 per LDS stage
  __shared__ __attribute__((aligned(16))) __bf16 smem[STAGES * STG];
  __bf16* sA = smem;
  __bf16* sB = smem + NP * PLA;

  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  if (g.stagger > 0) {   // experiment: start the co-resident workgroups of a CU at different phases
    constexpr int PER_CU = ((VAE_X3_PIPE && NP <= 2 && BM == 128 && BN == 128) || ROWSC || XONE != 0 || BN > 160) ? 2 : (NP <= 2 ? 3 : 2);
    const int sl = (blockIdx.x >> 8) % PER_CU;
    for (int i = 0; i < sl * g.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  const int tiles_m = g.tiles_m, tiles_n = g.tiles_n;
  const int per_split = tiles_m * tiles_n;
  const int s = id / per_split, t = id - s * per_split;
  int tm, tn;
  if (g.group_m > 0) {
    // many row tiles AND many column tiles (decoder heads at B >= 2048): walking all row tiles per column tile makes the
    // workgroups resident on an XCD sweep the whole A operand (4 MB at B = 4096) between two uses of a line, and the
    // output stream evicts it from the 4 MB L2 (measured: 787 MB fetched for 45 MB of operands).  Row tiles go in groups
    // of group_m: ids walk the group's rows fastest, then every column tile, then the next group -- the resident set is
    // group_m A panels + a dozen B tiles.
    const int gsz = g.group_m * tiles_n;
    const int grp = t / gsz, r = t - grp * gsz;
    const int rows = min(g.group_m, tiles_m - grp * g.group_m);
    tm = grp * g.group_m + r % rows;
    tn = r / rows;
  } else if (tiles_m <= tiles_n) { tm = t % tiles_m; tn = t / tiles_m; }
  else                           { tn = t % tiles_n; tm = t / tiles_n; }
  if (VAE_STAT_INTERLEAVE && LAYOUT == NT && g.stat_m != nullptr) {
    // the column tiles that carry the softmax statistics in their output stage (the first stat_cols columns) take longer than
    // the others: alternate the two kinds along the launch order, so that the workgroups sharing a CU are a mix of both
    const int ns = min(tiles_n, (g.stat_cols + BN - 1) / BN), both = 2 * min(ns, tiles_n - ns);
    if (tn < both) tn = (tn & 1) ? ns + (tn >> 1) : (tn >> 1);
    else if (ns < tiles_n - ns) tn = tn;                 // the tail: plain tiles only, already in place (tn >= 2 ns)
    else tn = tn - (tiles_n - ns);                       // the tail: statistics tiles only
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = s * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, h = lane >> 5;
  // GemmArgs::row_sub of this tile's rows: requested now, parked in LDS by the output stage (a load there would be a
  // dependent round trip that every workgroup of the single generation pays at the same time)
  const float rs_pref = (NP == 2 && g.row_sub && g.split == 1 && tid < BM && m0 + tid < g.M) ? g.row_sub[m0 + tid] : 0.f;

  // ---- global -> register staging: unconditional 16-byte (bf16 storage: 8-byte) loads from clamped addresses ----
  // addresses = a workgroup-uniform base (the tile's origin at this K step: scalar registers, advanced per step) + a
  // per-lane 32-bit byte offset that stays the same for every step (row inside the tile x leading dimension: < 2^32 for
  // any leading dimension below 6 M elements) -- one register per load instead of a 64-bit pointer
  const char* bA = reinterpret_cast<const char*>(reinterpret_cast<const TA*>(g.A) + (A_KC ? (int64_t)m0 * g.lda + kbeg : (int64_t)kbeg * g.lda));
  const char* bB = reinterpret_cast<const char*>(reinterpret_cast<const TB*>(g.B) + (B_KC ? (int64_t)n0 * g.ldb + kbeg : (int64_t)kbeg * g.ldb));
  uint32_t oa[NA], ob[NB];
  int wa[NA], wb[NB];  // LDS element offsets inside a plane
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = tid + 256 * i;
    if (A_KC) {
      const int row = c >> 3, k4 = c & 7;
      oa[i] = (uint32_t)(((int64_t)min(row, g.M - 1 - m0) * g.lda + 4 * k4) * (int64_t)sizeof(TA));
      wa[i] = row * LDK + 4 * k4;
    } else {
      const int k = c / (BM / 4), r4 = c % (BM / 4);
      oa[i] = (uint32_t)(((int64_t)k * g.lda + min(m0 + 4 * r4, g.M - 4)) * (int64_t)sizeof(TA));
      wa[i] = k * LDR + 4 * r4;
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int c = tid + 256 * i;
    if (B_KC) {
      const int row = c >> 3, k4 = c & 7;
      ob[i] = (uint32_t)(((int64_t)min(row, g.N - 1 - n0) * g.ldb + 4 * k4) * (int64_t)sizeof(TB));
      wb[i] = row * LDK + 4 * k4;
    } else {
      const int k = c / (BN / 4), r4 = c % (BN / 4);
      ob[i] = (uint32_t)(((int64_t)k * g.ldb + min(n0 + 4 * r4, g.N - 4)) * (int64_t)sizeof(TB));
      wb[i] = k * LDR + 4 * r4;
    }
  }
  const int64_t step_a = (A_KC ? 32 : 32 * g.lda) * (int64_t)sizeof(TA);   // bytes per K step
  const int64_t step_b = (B_KC ? 32 : 32 * g.ldb) * (int64_t)sizeof(TB);
  constexpr int NSET = (VAE_X3_PIPE && NP <= 2 && BM == 128 && BN == 128 && VAE_X3_RELOAD) ? VAE_X3_SETS : 1;
  RA ra_[NSET][NA];   // staging registers (pipelined loop: NSET sets, K steps of equal index mod NSET share one)
  RB rb_[NSET][NB];
  RA (&ra)[NA] = ra_[0];
  RB (&rb)[NB] = rb_[0];
  auto load_to = [&](RA* ra, RB* rb) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const RA*>(bA + oa[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const RB*>(bB + ob[i]);
  };
  auto fetch_to = [&](RA* ra, RB* rb) {
    load_to(ra, rb);
    bA += step_a;
    bB += step_b;
  };
  auto fetch = [&]() { fetch_to(ra, rb); };
  // last, partial K step (K % 32 != 0, K % 4 == 0): chunks past kend read a safe address and count as zeros
  auto fetch_tail_to = [&](int tail, RA* ra, RB* rb) {
    RA ZA;
    RB ZB;
    memset(&ZA, 0, sizeof(ZA));
    memset(&ZB, 0, sizeof(ZB));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + 256 * i;
      const int kk = A_KC ? 4 * (c & 7) : c / (BM / 4);
      ra[i] = kk < tail ? *reinterpret_cast<const RA*>(bA + oa[i]) : ZA;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + 256 * i;
      const int kk = B_KC ? 4 * (c & 7) : c / (BN / 4);
      rb[i] = kk < tail ? *reinterpret_cast<const RB*>(bB + ob[i]) : ZB;
    }
  };
  auto fetch_tail = [&](int tail) { fetch_tail_to(tail, ra, rb); };
  float scA = 1.f, scB = 1.f;   // NP == 2: operand scales
  int x_one = 0;                // XONE: every count is exact in one scaled f16 term (workgroup-uniform)
  // Maximum of n partial maxima (16-byte aligned array): 16-byte loads, four in flight per lane
  auto partial_max = [&](const float* __restrict__ p, int n) {
    float m = 0.f;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int n4 = n >> 2;
    auto m4 = [](const float4& v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); };
    int i = tid;
    for (; i + 768 < n4; i += 1024) {
      const float4 a = p4[i], b = p4[i + 256], c = p4[i + 512], d = p4[i + 768];
      m = fmaxf(m, fmaxf(fmaxf(m4(a), m4(b)), fmaxf(m4(c), m4(d))));
    }
    for (; i < n4; i += 256) m = fmaxf(m, m4(p4[i]));
    for (int j = (n4 << 2) + tid; j < n; j += 256) m = fmaxf(m, p[j]);
    return m;
  };
  // Called once, after the first tile's loads are issued (its latency hides behind them) and before the first split
  // ROWSC: the scales of the four product rows (columns of the k-major A) that chunk i of this thread stages -- its 16 bytes
  // of the (BM / 4)-quad row, a 16-byte LDS read per chunk (for BM = 128 the same quad every time)
  auto scale_a4 = [&](int i) {
    if (ROWSC) return *reinterpret_cast<const float4*>(&row_sc[4 * ((tid + 256 * i) % (BM / 4))]);
    return make_float4(scA, scA, scA, scA);
  };
  auto scales = [&]() {
    if (NP != 2) return;
    if (ROWSC) {   // per-row scales from the producer's per-chunk column maxima; B keeps one scale
      __shared__ float red_b[4];
      if (tid < BM) {
        const int m = m0 + tid;
        float mx = 0.f;
        if (m < g.M) {
          const float* q = m < g.rmax_split ? g.rmax0 + m : g.rmax1 + (m - g.rmax_split);
          int c = 0;
#if VAE_ROWSC_BATCH32
          for (; c + 32 <= g.rmax_chunks; c += 32) {   // (B = 512: the backward leaves exactly 32 chunks -- one round trip)
            float t[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) t[u] = q[(int64_t)(c + u) * g.rmax_stride];
#pragma unroll
            for (int u = 0; u < 32; ++u) mx = fmaxf(mx, t[u]);
          }
#endif
          for (; c + 16 <= g.rmax_chunks; c += 16) {   // sixteen loads in flight, not a chain of round trips
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = q[(int64_t)(c + u) * g.rmax_stride];
#pragma unroll
            for (int u = 0; u < 16; ++u) mx = fmaxf(mx, t[u]);
          }
          for (; c + 4 <= g.rmax_chunks; c += 4) {
            const float t0 = q[(int64_t)c * g.rmax_stride], t1 = q[(int64_t)(c + 1) * g.rmax_stride];
            const float t2 = q[(int64_t)(c + 2) * g.rmax_stride], t3 = q[(int64_t)(c + 3) * g.rmax_stride];
            mx = fmaxf(fmaxf(mx, fmaxf(t0, t1)), fmaxf(t2, t3));
          }
          for (; c < g.rmax_chunks; ++c) mx = fmaxf(mx, q[(int64_t)c * g.rmax_stride]);
        }
        const float sc = vae_scale_from_absmax(mx);
        row_sc[tid] = sc;
        row_inv[tid] = 1.f / sc;
      }
      float mb = partial_max(g.sb, g.nb);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mb = fmaxf(mb, __shfl_xor(mb, o, 64));
      if (lane == 0) red_b[wave] = mb;
      __syncthreads();
      scB = vae_scale_from_absmax(fmaxf(fmaxf(red_b[0], red_b[1]), fmaxf(red_b[2], red_b[3])));
      return;
    }
    if (g.na > 0 || g.nb > 0) {  // reduce the producers' partial maxima (a few KB, L2-resident)
      // 8 floats: own storage, or (PIPE: two stages of exactly 40 KB = two workgroups per CU, not a byte to spare) the start
      // of LDS stage 1, which nothing writes before the barrier that follows the first conversion
      __shared__ float red_own[PIPE ? 1 : 12];
      float* red_s = PIPE ? reinterpret_cast<float*>(smem + STG) : red_own;
      float ma = XOPT ? 0.f : partial_max(g.sa, g.na), mb = partial_max(g.sb, g.nb);
      float mf = (XONE != 0 && !XOPT) ? partial_max(g.xone_flags, g.n_xone) : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o, 64));
        mb = fmaxf(mb, __shfl_xor(mb, o, 64));
        if (XONE != 0) mf = fmaxf(mf, __shfl_xor(mf, o, 64));
      }
      if (lane == 0) { red_s[wave] = ma; red_s[4 + wave] = mb; if (XONE != 0) red_s[8 + wave] = mf; }
      __syncthreads();
      scA = vae_scale_from_absmax(fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3])));
      scB = vae_scale_from_absmax(fmaxf(fmaxf(red_s[4], red_s[5]), fmaxf(red_s[6], red_s[7])));
      if (XONE != 0)
        x_one = __builtin_amdgcn_readfirstlane(fmaxf(fmaxf(red_s[8], red_s[9]), fmaxf(red_s[10], red_s[11])) > 0.f ? 0 : 1);
      if (XOPT) { scA = 1.f; x_one = 1; }   // counts as they are; whether that was right is found out while they pass
    } else {
      if (g.sa) scA = *g.sa;
      if (g.sb) scB = *g.sb;
    }
  };
  // with_lo (a std::bool_constant): false = the operand is exact in its hi term (XONE): no lo plane is computed or written
  auto put = [&](__bf16* base, int plane, int off, const auto& vin, const float4 sc4, auto with_lo) {
    if constexpr (std::is_same_v<std::decay_t<decltype(vin)>, uint2>) {   // bf16 in HBM: already the operand image
      *reinterpret_cast<uint2*>(base + off) = vin;
      return;
    } else {
    float f[4];
    if constexpr (std::is_same_v<std::decay_t<decltype(vin)>, uint32_t>) {   // four uint8 counts
      f[0] = (float)(vin & 0xffu); f[1] = (float)((vin >> 8) & 0xffu); f[2] = (float)((vin >> 16) & 0xffu); f[3] = (float)(vin >> 24);
    } else {
      const float4 v = vin;
      f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    if (NP == 2 && VAE_X3_MIXCVT) {   // the same two terms, by hand: 2 instructions per element instead of the compiler's ~ 4
      const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
      uint2 h4, l4;
      split_pair_f16(f[0], f[1], sc[0], sc[1], h4.x, l4.x, with_lo);
      split_pair_f16(f[2], f[3], sc[2], sc[3], h4.y, l4.y, with_lo);
      *reinterpret_cast<uint2*>(base + off) = h4;
      if constexpr (decltype(with_lo)::value) *reinterpret_cast<uint2*>(base + plane + off) = l4;
      return;
    }
    if (NP == 2) {  // two f16 terms of the scaled value: hi = rn(a s), lo = rn(a s - hi): 22 significant bits
      const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
      f16x4 h4, l4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float as = f[j] * sc[j];
        const _Float16 h = (_Float16)as;
        h4[j] = h;
        if constexpr (decltype(with_lo)::value) l4[j] = (_Float16)(as - (float)h);
      }
      *reinterpret_cast<bf16x4*>(base + off) = __builtin_bit_cast(bf16x4, h4);
      if constexpr (decltype(with_lo)::value) *reinterpret_cast<bf16x4*>(base + plane + off) = __builtin_bit_cast(bf16x4, l4);
      return;
    }
    bf16x4 hi, mi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __bf16 a, b, d;
      split3(f[j], a, b, d);
      hi[j] = a; mi[j] = b; lo[j] = d;
    }
    *reinterpret_cast<bf16x4*>(base + off) = hi;
    if (NP == 3) {
      *reinterpret_cast<bf16x4*>(base + plane + off) = mi;
      *reinterpret_cast<bf16x4*>(base + 2 * plane + off) = lo;
    }
    }
  };
#ifdef VAE_PROBE
  int pi = 0;
#endif
  // `one` (a std::integral_constant<int, 0 | 1 | 2>): which operand runs on its hi term alone in this instantiation of the
  // K loop (0: none, 1: A, 2: B) -- see XONE
  // again: the base pointers stand on the NEXT full K step and every staging register is requested anew as soon as its
  // contents are converted (VAE_X3_RELOAD) -- the loads then travel during the rest of the conversion and the barrier too
  auto stage = [&](auto one, bool again = false) {
    constexpr int ONE = decltype(one)::value;
    GSTAMP();
    __syncthreads();  // the previous tile's fragment reads are done
    GSTAMP();
    if (again) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        put(sA, PLA, wa[i], ra[i], scale_a4(i), std::bool_constant<ONE != 1>{});
        ra[i] = *reinterpret_cast<const RA*>(bA + oa[i]);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        put(sB, PLB, wb[i], rb[i], make_float4(scB, scB, scB, scB), std::bool_constant<ONE != 2>{});
        rb[i] = *reinterpret_cast<const RB*>(bB + ob[i]);
      }
      bA += step_a;
      bB += step_b;
      GSTAMP();
      __syncthreads();
      GSTAMP();
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) put(sA, PLA, wa[i], ra[i], scale_a4(i), std::bool_constant<ONE != 1>{});
#pragma unroll
    for (int i = 0; i < NB; ++i) put(sB, PLB, wb[i], rb[i], make_float4(scB, scB, scB, scB), std::bool_constant<ONE != 2>{});
    GSTAMP();
    __syncthreads();
    GSTAMP();
  };

  // ---- fragment addressing ----
  // KC: lane (row l31, k-group h) reads 8 consecutive k: [row][8h + 16 ks]
  // RC: 16-lane group j = lane>>4: rows 16 (j&1) .. +15, k-group j>>1; lane 4q+p of the group points
  //     at [k = 8 (j>>1) + q (+4)][row0 + 4p]
  const int la = lane & 15, lq = la >> 2, lp = la & 3, lj = lane >> 4;
  const int fa = A_KC ? (wm * (32 * TM) + l31) * LDK + 8 * h
                      : (8 * (lj >> 1) + lq) * LDR + wm * (32 * TM) + 16 * (lj & 1) + 4 * lp;
  const int fb = B_KC ? (wn * (32 * TNN) + l31) * LDK + 8 * h
                      : (8 * (lj >> 1) + lq) * LDR + wn * (32 * TNN) + 16 * (lj & 1) + 4 * lp;

  auto frag = [&](const __bf16* plane, bool kc, int f0, int tt, int ks) -> bf16x8 {
    if (kc) return *reinterpret_cast<const bf16x8*>(plane + f0 + tt * 32 * LDK + ks * 16);
    const __bf16* p = plane + f0 + ks * 16 * LDR + tt * 32;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * LDR));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    return u.v;
  };

  f32x16 acc[TM][TNN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // `between(group)` runs after the MFMAs of each of the 2 * TNN (ks, tj) groups: the pipelined loop converts the next
  // K step there, in the shadow of the matrix cores
  auto multiply_from = [&](auto one, const __bf16* sAst, const __bf16* sBst, auto&& between) {
    constexpr int ONE = decltype(one)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[TM][NP];
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int q = 0; q < (ONE == 1 ? 1 : NP); ++q) a[ti][q] = frag(sAst + q * PLA, A_KC, fa, ti, ks);
#pragma unroll
      for (int tj = 0; tj < TNN; ++tj) {
        bf16x8 b[NP];
#pragma unroll
        for (int q = 0; q < (ONE == 2 ? 1 : NP); ++q) b[q] = frag(sBst + q * PLB, B_KC, fb, tj, ks);
#pragma unroll
        for (int ti = 0; ti < TM; ++ti) {
          f32x16 c = acc[ti][tj];  // smallest terms first
          if (NP == 1) {  // plain bf16 operands (VAE_BF16): one product
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[0], c, 0, 0, 0);
            continue;
          }
#if defined(VAE_ABL) && (VAE_ABL & 4)
          if (NP == 2) continue;
#endif
          if (NP == 2) {  // f16 hi / lo: lo*hi, hi*lo, hi*hi (the dropped lo*lo is < 2^-22 of the product)
            const f16x8 ah = __builtin_bit_cast(f16x8, a[ti][0]), al = __builtin_bit_cast(f16x8, a[ti][NP > 1 ? 1 : 0]);
            const f16x8 bh = __builtin_bit_cast(f16x8, b[0]), bl = __builtin_bit_cast(f16x8, b[NP > 1 ? 1 : 0]);
            if constexpr (ONE != 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);   // (an exact operand has no lo term)
            if constexpr (ONE != 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
            acc[ti][tj] = c;
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][NP > 1 ? 1 : 0], b[NP > 1 ? 1 : 0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][NP > 2 ? 2 : 0], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[NP > 2 ? 2 : 0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][NP > 1 ? 1 : 0], b[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[NP > 1 ? 1 : 0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[0], c, 0, 0, 0);
          acc[ti][tj] = c;
        }
        between(ks * TNN + tj);
      }
    }
  };
  auto multiply = [&](auto one) { multiply_from(one, sA, sB, [](int) {}); };

  (void)stage; (void)multiply; (void)fetch; (void)fetch_tail;   // the unpipelined loop's helpers
  GSTAMP();
  const int nfull = kend > kbeg ? (kend - kbeg) / BK : 0;
  const int tail = kend > kbeg ? (kend - kbeg) % BK : 0;   // only the last slab of a K that is not a multiple of 32
  // The first K step's loads are issued, then the operand scales (and the XONE decision) are derived while they travel;
  // run(one) is the rest of the K loop, instantiated per `one` and chosen ONCE per workgroup (uniform branch).
  int left = nfull;                                      // PIPE: full steps not fetched yet
  auto fetch_full = [&]() {                              // branch-free: past the end the last step is read again (unused)
    load_to(ra, rb);
    bA += left > 1 ? step_a : 0;
    bB += left > 1 ? step_b : 0;
    --left;
  };
  (void)fetch_full;
  // PIPE: a staging register is loaded again as soon as its contents are converted (step i converts the registers of
  // step i + 1 and requests step i + 2 into them): a whole K step between request and use at no cost in registers
  auto advance = [&]() {                                 // to the next full step; stays on the last one past the end
    --left;
    bA += left > 0 ? step_a : 0;
    bB += left > 0 ? step_b : 0;
  };
  auto reload_one = [&](int pidx, auto setc) {
    constexpr int SET = decltype(setc)::value;
    if (pidx < NA) ra_[SET][pidx] = *reinterpret_cast<const RA*>(bA + oa[pidx]);
    else rb_[SET][pidx - NA] = *reinterpret_cast<const RB*>(bB + ob[pidx - NA]);
  };
  (void)advance; (void)reload_one;
  // XOPT, column tile 0: what this thread sees of the counts -- sums of its NA rows, maximum, OR of the bit patterns
  float xs[XOPT ? NA : 1];
  float xam = 0.f;
  uint32_t xbits = 0u;
#pragma unroll
  for (int i = 0; i < (XOPT ? NA : 1); ++i) xs[i] = 0.f;
  // the column tiles share the extra work: staging chunk i (rows (tid >> 3) + 32 i of the row tile) belongs to column tile
  // i mod min(tiles_n, NA)
  const int xown = min(tiles_n, NA);
  auto xmine = [&](int i) { return XOPT && tn == i % xown; };
  const bool xstat = XOPT && tn < xown;
  // ... and the row sums of B (sum_g W of the encoder's first layer) are shared the same way among the row tiles
  const int wown = min(tiles_m, NB);
  auto wmine = [&](int j) { return XOPT && tm == j % wown; };
  float wsr[XOPT ? NB : 1];
#pragma unroll
  for (int j = 0; j < (XOPT ? NB : 1); ++j) wsr[j] = 0.f;
  auto ws_take = [&](int j, const RB& r) {
    if constexpr (XOPT && std::is_same_v<RB, float4>) wsr[j] += (r.x + r.y) + (r.z + r.w);
  };
  (void)ws_take; (void)wmine;
  auto xs_take = [&](int pidx, const RA& rr) {
    if constexpr (XOPT && (std::is_same_v<RA, float4> || std::is_same_v<RA, uint32_t>)) {
      float4 r;
      if constexpr (std::is_same_v<RA, uint32_t>)
        r = make_float4((float)(rr & 0xffu), (float)((rr >> 8) & 0xffu), (float)((rr >> 16) & 0xffu), (float)(rr >> 24));
      else
        r = rr;
      xs[pidx] += (r.x + r.y) + (r.z + r.w);
      xam = fmaxf(fmaxf(xam, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
      xbits |= (__float_as_uint(r.x) | __float_as_uint(r.y)) | (__float_as_uint(r.z) | __float_as_uint(r.w));
    }
  };
  (void)xs_take; (void)xstat;
  if constexpr (XONE == 1) {
    if (g.gate) {   // the launch behind an optimistic one (GemmArgs::gate)
      float* red_g = reinterpret_cast<float*>(smem);   // nothing is staged yet
      // the first few workgroups finish the means, the next few the row sums of B
      const int n_mean = (g.M + 255) / 256;
      const int n_jobs = n_mean + (g.N + 255) / 256;
      for (int job = id; job < n_jobs; job += nwg) {   // (one job per workgroup unless the grid is tiny)
        if (job < n_mean) {
          const int m = job * 256 + tid;
          if (m < g.M) g.mean_out[m] = slab_sum(g.xs_sum + m, g.split, g.M) / (float)g.K;
        } else {
          const int n = (job - n_mean) * 256 + tid;
          if (n < g.N) g.wsum_out[n] = slab_sum(g.ws_sum + n, g.split, g.N);
        }
      }
      float mf = partial_max(g.xone_flags, g.n_xone);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mf = fmaxf(mf, __shfl_xor(mf, o, 64));
      __syncthreads();
      if (lane == 0) red_g[4 + wave] = mf;
      __syncthreads();
      const float any = fmaxf(fmaxf(red_g[4], red_g[5]), fmaxf(red_g[6], red_g[7]));
      __syncthreads();   // (the scale reduction borrows this LDS next)
      if (!(any > 0.f)) return;   // the optimistic product stands
    }
  }
  if constexpr (PIPE) {
    if (nfull > 0) {
      load_to(ra, rb);
      if (VAE_X3_RELOAD) advance(); else { bA += left > 1 ? step_a : 0; bB += left > 1 ? step_b : 0; --left; }
      if constexpr (NSET == 2) {   // the second set: step 1
        load_to(ra_[NSET - 1], rb_[NSET - 1]);
        advance();
      }
    }
  } else {
    if (nfull > 0) fetch();
    else if (tail) fetch_tail(tail);
  }
#if defined(VAE_ABL) && (VAE_ABL & 16)
  (void)scales;
#else
  scales();
#endif
  auto run = [&](auto one) {
    constexpr int ONE = decltype(one)::value;
    if constexpr (PIPE) {
      // the pipelined loop covers the FULL K steps; a partial last step (K % 32 != 0) runs unpipelined after it
      // Step i: the loads of step i + 1 are issued first and land during the first half of the MFMA groups; the second
      // half alternates with their conversion into the other LDS stage (after the last step the registers hold a re-read
      // of it, written to the stage nobody reads again).
#ifndef VAE_X3_G0
#define VAE_X3_G0 (NGRP / 2)
#endif
      constexpr int NGRP = 2 * TNN, G0 = VAE_X3_G0, NPUT = NA + NB;
      auto put_one = [&](int pidx, __bf16* base, auto setc) {
        constexpr int SET = decltype(setc)::value;
        if (pidx < NA) put(base, PLA, wa[pidx], ra_[SET][pidx], scale_a4(pidx), std::bool_constant<ONE != 1>{});
        else put(base + NP * PLA, PLB, wb[pidx - NA], rb_[SET][pidx - NA], make_float4(scB, scB, scB, scB), std::bool_constant<ONE != 2>{});
      };
      auto step = [&](auto curc, bool more) {   // more: the registers converted in this step hold a step of the K range
        constexpr int cur = decltype(curc)::value;
        using SetC = std::integral_constant<int, NSET == 2 ? (cur ^ 1) : 0>;   // the registers of step i + 1
        GSTAMP();
        if (!VAE_X3_RELOAD) fetch_full();
        __bf16* nxt = smem + (cur ^ 1) * STG;
        if constexpr (XOPT) {   // ahead of the interleaved section, which stays one basic block (branches only here)
          if (more) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
              if (xmine(i)) xs_take(i, ra_[SetC::value][i]);
#pragma unroll
            for (int j = 0; j < NB; ++j)
              if (wmine(j)) ws_take(j, rb_[SetC::value][j]);
          }
        }
        multiply_from(one, smem + cur * STG, smem + cur * STG + NP * PLA, [&](int grp) {
#pragma unroll
          for (int pidx = 0; pidx < NPUT; ++pidx)
            if (G0 + pidx * (NGRP - G0) / NPUT == grp) {
#ifdef VAE_ABL   // ablations for tools/micro/gemm_probe.hip: 1 = no conversion / LDS writes, 2 = no global loads, 4 = no multiply
              if (!(VAE_ABL & 1)) put_one(pidx, nxt, SetC{});
              if (VAE_X3_RELOAD && !(VAE_ABL & 2)) reload_one(pidx, SetC{});
#else
              put_one(pidx, nxt, SetC{});
              if (VAE_X3_RELOAD) reload_one(pidx, SetC{});
#endif
            }
          if (VAE_X3_SCHEDB) __builtin_amdgcn_sched_barrier(0);   // keep a group's conversions in its own slot
        });
        if (VAE_X3_RELOAD) advance();
        GSTAMP();
        __syncthreads();   // stage cur ^ 1 published; every wave is done reading stage cur
        GSTAMP();
      };
      if (nfull > 0) {
#pragma unroll
        for (int pidx = 0; pidx < NPUT; ++pidx) {
          put_one(pidx, smem, std::integral_constant<int, 0>{});
          if (XOPT && pidx < NA) { if (xmine(pidx)) xs_take(pidx, ra_[0][pidx < NA ? pidx : 0]); }
          if (XOPT && pidx >= NA) { if (wmine(pidx - NA)) ws_take(pidx >= NA ? pidx - NA : 0, rb_[0][pidx >= NA ? pidx - NA : 0]); }
          if (VAE_X3_RELOAD) reload_one(pidx, std::integral_constant<int, 0>{});
        }
        if (VAE_X3_RELOAD) advance();
        __syncthreads();
        for (int i = 0; i < nfull; i += 2) {
          step(std::integral_constant<int, 0>{}, i + 1 < nfull);
          if (i + 1 < nfull) step(std::integral_constant<int, 1>{}, i + 2 < nfull);
        }
        if (tail) {   // the base pointers stopped at the last full step: move on to the partial one
          bA += step_a;
          bB += step_b;
        }
      }
      if (tail) {
        fetch_tail(tail);
        if constexpr (XOPT) {   // the partial last K step of the last slab (zero-filled past K) counts too
#pragma unroll
          for (int i = 0; i < NA; ++i)
            if (xmine(i)) xs_take(i, ra[i]);
#pragma unroll
          for (int j = 0; j < NB; ++j)
            if (wmine(j)) ws_take(j, rb[j]);
        }
        stage(one);
        multiply(one);
        __syncthreads();
      }
    } else {
      for (int i = 0; i < nfull; ++i) {
#ifdef VAE_ABL   // ablations for tools/micro/gemm_probe.hip: 1 = stage once, 2 = fetch once, 4 = no multiply
        if (!(VAE_ABL & 1) || i == 0) stage(one);
        if ((!(VAE_ABL & 2) || i == 0) && i + 1 < nfull) fetch();
        if (!(VAE_ABL & 4)) multiply(one);
#else
        if (VAE_X3_RELOAD && VAE_X3_RELOAD_FLAT && LAYOUT == NT && !ROWSC) {   // (measured: decoder forward -1.8 us; the row-scaled TN weight gradient +1.5, so not there)
          stage(one, i + 1 < nfull);
          if (i + 1 >= nfull && tail) fetch_tail(tail);
        } else {
          stage(one);
          if (i + 1 < nfull) fetch();  // in flight during the MFMAs below
          else if (tail) fetch_tail(tail);
        }
        multiply(one);
#endif
      }
      if (tail) {
        stage(one);
        multiply(one);
      }
    }
  };
  if constexpr (XOPT) {
    run(std::integral_constant<int, 1>{});
  } else if constexpr (XONE != 0) {
    if (x_one) run(std::integral_constant<int, XONE>{});
    else run(std::integral_constant<int, 0>{});
  } else {
    run(std::integral_constant<int, 0>{});
  }
  GSTAMP();
  if constexpr (XOPT) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (!wmine(j)) continue;
      float v = wsr[j];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      const int row = (tid >> 3) + 32 * j;
      if ((tid & 7) == 0 && n0 + row < g.N) g.ws_sum[(int64_t)s * g.N + n0 + row] = v;
    }
    if (xstat) {   // (uniform per workgroup)
      const int writers = g.tiles_m * xown * g.split;
#pragma unroll
      for (int i = 0; i < NA; ++i) {   // the eight lanes that staged a row's 32 counts per step
        if (!xmine(i)) continue;
        float v = xs[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        const int row = (tid >> 3) + 32 * i;
        if ((tid & 7) == 0 && m0 + row < g.M) g.xs_sum[(int64_t)s * g.M + m0 + row] = v;
      }
      float am = xam;
      uint32_t bits = xbits;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        am = fmaxf(am, __shfl_xor(am, o, 64));
        bits |= (uint32_t)__shfl_xor((int)bits, o, 64);
      }
      float* red_x = reinterpret_cast<float*>(smem);
      __syncthreads();   // every wave is done with the operand stages
      if (lane == 0) { red_x[wave] = am; red_x[4 + wave] = __uint_as_float(bits); }
      __syncthreads();
      if (tid == 0) {
        am = fmaxf(fmaxf(red_x[0], red_x[1]), fmaxf(red_x[2], red_x[3]));
        bits = (__float_as_uint(red_x[4]) | __float_as_uint(red_x[5])) | (__float_as_uint(red_x[6]) | __float_as_uint(red_x[7]));
        // exact in one unscaled f16 term: at most 11 significant bits, no overflow, and nothing a scale would have had to
        // lift out of f16's subnormal range (a batch of tiny values keeps the scaled two-term product)
        const bool bad = (bits & 0x1fffu) != 0u || !(am < 32768.f) || (am > 0.f && am < 1.f);
        const int w = (tm * xown + tn) * g.split + s;
        g.xs_max[w] = am;
        g.xs_flag[w] = bad ? 1.f : 0.f;
        for (int e = writers + w; e < g.xs_n; e += writers) { g.xs_max[e] = 0.f; g.xs_flag[e] = 0.f; }
      }
      __syncthreads();   // (the output stage parks its patches in the same LDS)
    }
  }

#if defined(VAE_ABL) && (VAE_ABL & 8)   // no output stage (one element keeps the accumulators alive)
  if (acc[0][0][0] == 12345.f) g.C[0] = acc[TM - 1][TNN - 1][3];
#else
  x3_epilogue<LAYOUT, BM, BN, WAVES_M, NP, CBF, (NP * (PLA + PLB) * 2 >= 4 * 32 * 40 * 4)>(g, acc, smem, m0, n0, tn, s, scA, scB,
                                                                                            ROWSC ? row_inv : nullptr, rs_pref);
#endif
  GSTAMP();
}

// ---------------------------------------------------------------------------------------
// bf16 product with BOTH operands stored as bf16 in HBM (a bf16-mode step: bf16 activations and bf16 images of the
// weights): nothing to convert, so the operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), two stages, one
// barrier per 64-deep K-tile, no staging registers -- the structure of gemm32_dma_kernel with half-size elements:
//   KC operand image: [rows][64 k] bf16 = 128-byte rows of 8 x 16 B; slot = chunk ^ ((row >> 1) & 7), applied to the
//     SOURCE address (the DMA destination is lane-linear) and again on the read; fragments by ds_read_b128.
//   RC operand image: [64 k][BM or BN] bf16, fragments by the transposing ds_read_b64_tr_b16; 128-wide images swizzle
//     the 16-byte chunk with 2 (k & 3) so that the four k rows a 16-lane group reads fall into different banks
//     (160-wide rows are 320 B apart and need none).
// The register-staged split kernel spends a K-tile as: wait for the global loads, write LDS, two barriers, then the
// MFMAs -- at B = 4096 a workgroup's 32-deep K step takes ~4300 cycles for 320 cycles of matrix work per wave
// (tools/micro/gemm_probe.hip).  Here a K-tile's loads are issued a whole tile of MFMAs ahead and land by themselves.
// K tail (K % 64 != 0, K % 8 == 0): chunks past the end read a zero page.  Output stage: x3_epilogue.
// ---------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};
__device__ __forceinline__ void dma16h(const uint16_t* src, uint32_t lds_wave_base) {
  dma16(reinterpret_cast<const float*>(src), lds_wave_base);
}

template <int LAYOUT, int BM, int BN, int WAVES_M, bool CBF>
__global__ __launch_bounds__(256, 2) void gemm16_dma_kernel(GemmArgs g) {
  constexpr bool A_KC = (LAYOUT != TN);
  constexpr bool B_KC = (LAYOUT == NT);
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = BM / (32 * WAVES_M), TNN = BN / (32 * WAVES_N);
  constexpr int KT = 64;
  constexpr int PA = BM / 32, PB = BN / 32;          // 1 KB DMA pieces per wave and operand
  constexpr int OPA = KT * BM, OPB = KT * BN;        // bf16 elements per operand per stage
  constexpr int STAGE = OPA + OPB;
  constexpr bool SWZ_A = !A_KC && (BM % 64 == 0), SWZ_B = !B_KC && (BN % 64 == 0);
  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * STAGE];   // 64-72 KB: two workgroups per CU

  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tiles_m = g.tiles_m, tiles_n = g.tiles_n;
  const int per_split = tiles_m * tiles_n;
  const int s = id / per_split, t = id - s * per_split;
  int tm, tn;
  if (g.group_m > 0) {   // as gemm_x3_kernel
    const int gsz = g.group_m * tiles_n;
    const int grp = t / gsz, r = t - grp * gsz;
    const int rows = min(g.group_m, tiles_m - grp * g.group_m);
    tm = grp * g.group_m + r % rows;
    tn = r / rows;
  } else if (tiles_m <= tiles_n) { tm = t % tiles_m; tn = t / tiles_m; }
  else                           { tn = t % tiles_n; tm = t / tiles_n; }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = s * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- per-lane DMA sources: one 1 KB piece per wave-instruction; ka / kb = the k offset of the lane's chunk in a tile ----
  const uint16_t* gA = reinterpret_cast<const uint16_t*>(g.A);
  const uint16_t* gB = reinterpret_cast<const uint16_t*>(g.B);
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero16);
  const uint16_t* pa[PA];
  const uint16_t* pb[PB];
  int ka[PA], kb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int piece = wave * PA + i;
    if (A_KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c8 = (lane & 7) ^ ((row >> 1) & 7);
      ka[i] = 8 * c8;
      pa[i] = gA + (int64_t)min(m0 + row, g.M - 1) * g.lda + kbeg + 8 * c8;
    } else {
      const int c = piece * 64 + lane, k = c / (BM / 8), ch = c % (BM / 8);
      const int sc = SWZ_A ? (ch ^ (2 * (k & 3))) : ch;
      ka[i] = k;
      pa[i] = gA + (int64_t)(kbeg + k) * g.lda + min(m0 + 8 * sc, g.M - 8);
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int piece = wave * PB + i;
    if (B_KC) {
      const int row = piece * 8 + (lane >> 3);
      const int c8 = (lane & 7) ^ ((row >> 1) & 7);
      kb[i] = 8 * c8;
      pb[i] = gB + (int64_t)min(n0 + row, g.N - 1) * g.ldb + kbeg + 8 * c8;
    } else {
      const int c = piece * 64 + lane, k = c / (BN / 8), ch = c % (BN / 8);
      const int sc = SWZ_B ? (ch ^ (2 * (k & 3))) : ch;
      kb[i] = k;
      pb[i] = gB + (int64_t)(kbeg + k) * g.ldb + min(n0 + 8 * sc, g.N - 8);
    }
  }
  const int64_t step_a = A_KC ? KT : (int64_t)KT * g.lda;
  const int64_t step_b = B_KC ? KT : (int64_t)KT * g.ldb;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)smem;
  // left = k extent still ahead of this tile's first row (>= KT except for the last tile of a ragged K)
  auto issue = [&](int stage, int left) {
    const uint32_t base_a = lds0 + (uint32_t)stage * (STAGE * 2u) + (uint32_t)wave * (PA * 1024u);
    const uint32_t base_b = lds0 + (uint32_t)stage * (STAGE * 2u) + OPA * 2u + (uint32_t)wave * (PB * 1024u);
    const bool full = left >= KT;   // wave-uniform
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      dma16h((full || ka[i] + (A_KC ? 8 : 1) <= left) ? pa[i] : zero, base_a + i * 1024u);
      pa[i] += step_a;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      dma16h((full || kb[i] + (B_KC ? 8 : 1) <= left) ? pb[i] : zero, base_b + i * 1024u);
      pb[i] += step_b;
    }
  };

  // ---- fragment addressing (bf16 element offsets inside an operand image) ----
  // KC: lane (row l31, k half h) reads chunk (2 ks + h) ^ ((row >> 1) & 7) of its row
  // RC: 16-lane group j: rows 16 (j & 1) .. +15 of the 32-row block, k group j >> 1; lane 4q + p of the group points at
  //     [k = 8 (j >> 1) + q (+ 4)][row0 + 4p]
  const int la = lane & 15, lq = la >> 2, lp = la & 3, lj = lane >> 4;
  const int ea = (l31 >> 1) & 7;
  const int kc_a = (wm * (32 * TM) + l31) * KT, kc_b = (wn * (32 * TNN) + l31) * KT;
  const int rc_ka = 8 * (lj >> 1) + lq;                         // k row inside a 16-deep step (+4 for the second half)
  const int rc_ea = wm * (32 * TM) + 16 * (lj & 1) + 4 * lp;    // element column before the 32 tt offset
  const int rc_eb = wn * (32 * TNN) + 16 * (lj & 1) + 4 * lp;

  f32x16 acc[TM][TNN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto frag_kc = [&](const uint16_t* img, int base, int tt, int ks) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(img + base + tt * 32 * KT + (((2 * ks + h) ^ ea) << 3));
  };
  auto frag_rc = [&](const uint16_t* img, int W, bool swz, int e_base, int tt, int ks) -> bf16x8 {
    const int e0 = e_base + tt * 32;
    const int col = swz ? ((((e0 >> 3) ^ (2 * lq)) << 3) | (e0 & 7)) : e0;
    const uint16_t* p = img + (ks * 16 + rc_ka) * W + col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * W));
    union { s16x4 s2[2]; bf16x8 v; } u;
    u.s2[0] = lo;
    u.s2[1] = hi;
    return u.v;
  };
  auto multiply = [&](const uint16_t* stage) {
    const uint16_t* sa = stage;
    const uint16_t* sb = stage + OPA;
#pragma unroll
    for (int ks = 0; ks < KT / 16; ++ks) {
      bf16x8 a[TM];
#pragma unroll
      for (int ti = 0; ti < TM; ++ti) a[ti] = A_KC ? frag_kc(sa, kc_a, ti, ks) : frag_rc(sa, BM, SWZ_A, rc_ea, ti, ks);
#pragma unroll
      for (int tj = 0; tj < TNN; ++tj) {
        const bf16x8 b = B_KC ? frag_kc(sb, kc_b, tj, ks) : frag_rc(sb, BN, SWZ_B, rc_eb, tj, ks);
#pragma unroll
        for (int ti = 0; ti < TM; ++ti) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti], b, acc[ti][tj], 0, 0, 0);
      }
    }
  };

  const int ntile = kend > kbeg ? (kend - kbeg + KT - 1) / KT : 0;
  if (ntile > 0) issue(0, kend - kbeg);
  dma_wait();
  __syncthreads();  // stage 0 published
  for (int i = 0; i < ntile; ++i) {
    if (i + 1 < ntile) issue((i + 1) & 1, kend - kbeg - (i + 1) * KT);  // lands during the MFMAs below
    multiply(smem + (i & 1) * STAGE);
    dma_wait();
    __syncthreads();  // next stage published; every wave is done reading this one
  }
  x3_epilogue<LAYOUT, BM, BN, WAVES_M, 1, CBF, true>(g, acc, smem, m0, n0, tn, s, 1.f, 1.f);
}

// Sums the split-K slabs in slab order, adds the bias, writes (or accumulates into) C.
// VEC: N % 4 == 0, ldc % 4 == 0, 16-byte aligned pointers -> 16 B per lane, 4 slabs in flight.
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int split, int M, int N,
                                                            const float* __restrict__ bias, float* __restrict__ C,
                                                            int64_t ldc, int accumulate, const float* __restrict__ row_sub) {
  const int64_t total = (int64_t)M * N;
  if (VEC) {
    const int64_t total4 = total >> 2;
    const int n4 = N >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
      const float4* p = reinterpret_cast<const float4*>(ws) + i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      int s = 0;
      for (; s + 8 <= split; s += 8) {  // eight slabs in flight, added in slab order
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = p[(int64_t)(s + u) * total4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { v.x += q[u].x; v.y += q[u].y; v.z += q[u].z; v.w += q[u].w; }
      }
      for (; s + 4 <= split; s += 4) {  // fixed order: slab s, s+1, s+2, s+3
        const float4 a = p[(int64_t)s * total4], b = p[(int64_t)(s + 1) * total4];
        const float4 c = p[(int64_t)(s + 2) * total4], d = p[(int64_t)(s + 3) * total4];
        v.x = (((v.x + a.x) + b.x) + c.x) + d.x; v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
        v.z = (((v.z + a.z) + b.z) + c.z) + d.z; v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
      }
      for (; s < split; ++s) {
        const float4 a = p[(int64_t)s * total4];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      const int m = (int)(i / n4), n = (int)(i % n4) * 4;
      if (bias) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + n);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      if (row_sub) {   // GemmArgs::row_sub
        const float r = row_sub[m];
        v.x -= r; v.y -= r; v.z -= r; v.w -= r;
      }
      float4* o = reinterpret_cast<float4*>(C + (int64_t)m * ldc + n);
      if (accumulate) {
        const float4 c0 = *o;
        v.x += c0.x; v.y += c0.y; v.z += c0.z; v.w += c0.w;
      }
      *o = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int m = (int)(i / N), n = (int)(i % N);
      float v = 0.f;
      for (int s = 0; s < split; ++s) v += ws[(int64_t)s * total + i];
      if (bias) v += bias[n];
      if (row_sub) v -= row_sub[m];
      float* p = C + (int64_t)m * ldc + n;
      if (accumulate) v += *p;
      *p = v;
    }
  }
}

// Tile size and K split for a shape.  small = 64x64 tiles (one dimension <= 64, or little work).
struct Plan { int tile; int split; };
Plan make_plan(int M, int N, int K, bool short_k_rule = true) {
  Plan p;
  static const int force_tile = getenv("VAE_GEMM_TILE") ? atoi(getenv("VAE_GEMM_TILE")) : 0;  // tuning knob
  bool small = (M <= 64 || N <= 64) || (2.0 * M * N * K < 2e8);
  // a short contraction with few 128-wide tiles (hidden-layer products of a large batch: 4096 x 256 x 256 = 64 tiles)
  // cannot buy parallelism with a K split worth its reduce launch: 64-wide tiles fill the chip instead
  static const int short_k = getenv("VAE_GEMM_SHORT_K") ? atoi(getenv("VAE_GEMM_SHORT_K")) : 512;   // A/B knob, 0 = off
  // (not for products over bf16-stored operands: those exist on the 128-wide pipeline only)
  if (short_k_rule && !small && K <= short_k && ((M + 127) / 128) * ((N + 127) / 128) < 128) small = true;
  if (force_tile == 64) small = true;
  if (force_tile == 128) small = false;
  p.tile = small ? 64 : 128;
  const int tiles = ((M + p.tile - 1) / p.tile) * ((N + p.tile - 1) / p.tile);
  const int ktiles = (K + BK - 1) / BK;
  const int target = small ? 256 : 512;               // workgroups wanted on the chip
  const int min_ktiles = small ? 2 : 4;                // K-tiles per slab at least
  int split = 1;
  if (tiles < (small ? 128 : 192)) {
    split = (target + tiles - 1) / tiles;
    const int max_split = ktiles / min_ktiles > 0 ? ktiles / min_ktiles : 1;
    if (split > max_split) split = max_split;
    if (split > 64) split = 64;
    if (split < 1) split = 1;
  }
  p.split = split;
  return p;
}

// Plan of the LDS-DMA pipeline: tile shape and K split that minimise a makespan model -- two
// workgroups per CU (512 slots) run in generations; a split pays for writing and summing its slabs.
struct DmaPlan { int bm, bn, split; };
DmaPlan make_dma_plan(int M, int N, int K, bool wide = false) {
  static const int force = getenv("VAE_GEMM_DMA_TILE") ? atoi(getenv("VAE_GEMM_DMA_TILE")) : 0;  // 0 | 128 | 160 | 320
  const int shapes[3][2] = {{128, 128}, {128, 160}, {160, 128}};
  if (wide && force == 320 && N >= 320 && M >= 128) return {128, 320, 1};
  const int ktiles = (K + BK - 1) / BK;
  DmaPlan best = {128, 128, 1};
  double best_us = 1e30;
  for (int c = 0; c < 3; ++c) {
    const int bm = shapes[c][0], bn = shapes[c][1];
    if (force == 128 && c != 0) continue;
    if (force == 160 && c == 0 && (M > 2048 || N > 2048)) continue;
    const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const int max_split = ktiles / 4 > 0 ? (ktiles / 4 < 64 ? ktiles / 4 : 64) : 1;
    for (int split = 1; split <= max_split; ++split) {
      const int kt = (ktiles + split - 1) / split;
      const int nsplit = (ktiles + kt - 1) / kt;
      if (nsplit != split) continue;  // same slab count as a smaller request
      const double gens = (double)((tiles * split + 511) / 512);
      // a generation of paired 128x128x32 K-tiles takes ~3.9 us; ~3 us of prologue + epilogue per generation
      double us = gens * (kt * 3.9 * (bm * bn / 16384.0) + 3.0);
      if (tiles * split <= 256) us = kt * 2.2 * (bm * bn / 16384.0) + 3.0;  // one workgroup per CU
      if (split > 1) us += 3.0 + split * (double)M * N * 4.0 / 5.0e6;  // the reduce pass reads the slabs at ~5 TB/s
      if (us < best_us) { best_us = us; best = {bm, bn, split}; }
    }
  }
  static const int force_split = getenv("VAE_GEMM_DMA_SPLIT") ? atoi(getenv("VAE_GEMM_DMA_SPLIT")) : 0;  // tuning knob
  if (force_split > 0 && ktiles / 4 > 1) {
    const int cap = ktiles / 4 < 64 ? ktiles / 4 : 64;
    best.split = force_split < cap ? force_split : cap;
  }
  return best;
}

template <int LAYOUT, int PREC, int TW>
void launch(const GemmArgs& g, bool norm, dim3 grid, hipStream_t st) {
  static const int pad = getenv("VAE_GEMM_LDS_PAD") ? atoi(getenv("VAE_GEMM_LDS_PAD")) : 0;  // tuning knob
  if (norm) hipLaunchKernelGGL((gemm_kernel<LAYOUT, PREC, true, TW>), grid, dim3(256), pad, st, g);
  else      hipLaunchKernelGGL((gemm_kernel<LAYOUT, PREC, false, TW>), grid, dim3(256), pad, st, g);
}
template <int LAYOUT, int PREC>
void launch_t(const GemmArgs& g, bool norm, dim3 grid, hipStream_t st, int tile) {
  if (tile == 128) launch<LAYOUT, PREC, 2>(g, norm, grid, st);
  else             launch<LAYOUT, PREC, 1>(g, norm, grid, st);
}

}  // namespace

extern "C" int64_t vae_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int split = 1;
  for (int rule = 0; rule < 2; ++rule) {   // with and without the short-contraction rule (bf16-stored operands: without)
    const Plan p = make_plan((int)M, (int)N, (int)K, rule != 0);
    if (p.split > split) split = p.split;
    if (K % 4 == 0 && p.tile == 128) {  // whichever pipeline the operands' alignment selects at run time
      const DmaPlan d = make_dma_plan((int)M, (int)N, (int)K);
      if (d.split > split) split = d.split;
    }
  }
  return split > 1 ? (int64_t)split * M * N * (int64_t)sizeof(float) : 0;
}

namespace {
// slabs_out != null: leave the product as *slabs_out partial (M,N) slabs (ld N) in the workspace
// -- no reduce pass, no bias -- for a consumer kernel that sums them while it reads.
int gemm_impl(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
              int64_t ldb, float* C, int64_t ldc, const float* bias, const float* x_mean, int accumulate,
              float* workspace, int64_t workspace_bytes, void* stream, int* slabs_out, const float* scale_a = nullptr,
              const float* scale_b = nullptr, int n_max_a = 0, int n_max_b = 0, VaeGemmRowStats* stats = nullptr,
              int io = 0, const VaeGemmRowScale* rsc = nullptr, const VaeGemmCounts* xo = nullptr) {
  VAE_REQUIRE(layout >= 0 && layout <= 2, "vae_gemm: layout must be 0 (NT), 1 (NN) or 2 (TN), got %d", layout);
  VAE_REQUIRE(precision >= 0 && precision <= 3, "vae_gemm: precision must be 0 (fp32), 1 (bf16), 2 (bf16x3) or 3 (f16x2), got %d",
              precision);
  // 3 = two scaled f16 terms, three products: only the split pipeline implements it; shapes outside it take the
  // exact fp32 MFMA kernel (the scales are then not needed)
  // Without scales (the public vae_gemm(precision = 3), or a step whose shapes fell out of the pipeline) the operands
  // could under- or overflow f16 unscaled: such calls run the three-bf16-term split, which needs no scale.
  const bool f16x2 = precision == 3 && ((scale_a && scale_b) || (rsc && scale_b));
  if (precision == 3) precision = 0;
  VAE_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30),
              "vae_gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  VAE_REQUIRE(A && B && (C || slabs_out), "vae_gemm: null operand");
  const int64_t a_min = layout == TN ? M : K, b_min = layout == NT ? K : N;
  VAE_REQUIRE(lda >= a_min && ldb >= b_min && ldc >= N, "vae_gemm: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)",
              (long long)lda, (long long)ldb, (long long)ldc);
  VAE_REQUIRE(!(x_mean && layout == NN), "vae_gemm: x_mean normalisation applies to NT (A) or TN (B) only");

  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.mean = x_mean;
  VAE_REQUIRE((n_max_a <= 0 || vae_aligned16(scale_a)) && (n_max_b <= 0 || vae_aligned16(scale_b)),
              "vae_gemm: partial maxima must be 16-byte aligned");
  g.sa = scale_a; g.sb = scale_b; g.na = n_max_a; g.nb = n_max_b;
  g.stat_m = g.stat_s = nullptr; g.stat_cols = 0;
  g.rmax0 = g.rmax1 = nullptr; g.rmax_split = g.rmax_chunks = 0; g.rmax_stride = 0;
  g.xone_flags = nullptr; g.n_xone = 0; g.row_sub = nullptr;
  g.stagger = 0;   // set below, once the tile shape is known
  if (stats) stats->slots = 0;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.accumulate = accumulate;
  const bool a_u8 = xo && xo->x_u8 && layout == NT, b_u8 = xo && xo->x_u8 && layout == TN;   // the count operand is the uint8 store
  g.vecA = (lda % 4 == 0) && (a_u8 ? (reinterpret_cast<uintptr_t>(A) & 3u) == 0 : (io & 1) ? (reinterpret_cast<uintptr_t>(A) & 7u) == 0 : vae_aligned16(A));
  g.vecB = (ldb % 4 == 0) && (b_u8 ? (reinterpret_cast<uintptr_t>(B) & 3u) == 0 : (io & 2) ? (reinterpret_cast<uintptr_t>(B) & 7u) == 0 : vae_aligned16(B));
  g.group_m = 0;
  const Plan plan = make_plan(g.M, g.N, g.K, io == 0 && !rsc && !xo);   // (bf16 storage and per-row scales exist on the 128-wide tiles only)
  static const int no_dma = getenv("VAE_GEMM_NO_DMA") ? atoi(getenv("VAE_GEMM_NO_DMA")) : 0;  // tuning knob
  const bool norm = x_mean != nullptr;
  const bool a_rc = layout == TN, b_rc = layout != NT;
  static const int x3_on = getenv("VAE_GEMM_X3") ? atoi(getenv("VAE_GEMM_X3")) : 1;
  // the split-precision pipeline takes any K % 4 == 0 (zero-filled last K step); the fp32 LDS-DMA one needs K % 32 == 0
  const bool k_ok = (x3_on || precision == 1 || f16x2) ? (K % 4 == 0) : (K % BK == 0);
  // (the tile kernels address a tile's rows with 32-bit byte offsets from its origin: 160 rows x ld x 4 B < 2^32)
  const bool ld_ok = lda < (1 << 22) && ldb < (1 << 22);
  const bool dma_ok = precision <= 1 && !norm && plan.tile == 128 && !no_dma && k_ok && g.vecA && g.vecB && ld_ok &&
                      (!a_rc || M % 4 == 0) && (!b_rc || N % 4 == 0) && M >= 4 && N >= 4;
  int bm = plan.tile, bn = plan.tile, split = plan.split;
  if (dma_ok) {
    const DmaPlan d = make_dma_plan(g.M, g.N, g.K, f16x2 && layout == NT && io == 0 && !rsc && !xo && !slabs_out);
    bm = d.bm; bn = d.bn; split = d.split;
    if (io & 4) split = 1;   // a bf16 output is written by the product kernel itself
  }
  VAE_REQUIRE(!rsc || (f16x2 && layout == TN && dma_ok && n_max_b > 0 && rsc->max0 && rsc->chunks > 0 &&
                       (rsc->split >= M || rsc->max1)),
              "vae_gemm: per-row scales need the two-f16-term wgrad (TN) on the tile pipeline and the maxima of B");
  const bool xopt = xo && xo->mode == 1;   // optimistic count operand (GemmArgs::xs_*): no maxima of A exist yet
  VAE_REQUIRE(!xo || (f16x2 && !rsc && layout != NN && dma_ok && (n_max_a > 0 || xopt) && n_max_b > 0 &&
                      (!xo->flags || (xo->n_flags > 0 && vae_aligned16(xo->flags))) &&
                      (!xo->row_sub || (layout == TN && !slabs_out && !accumulate))),
              "vae_gemm: the count-operand form needs a two-f16-term NT / TN product on the tile pipeline with operand maxima");
  VAE_REQUIRE(!(a_u8 || b_u8) || (dma_ok && bm == 128 && bn == 128 && io == 0 && (a_u8 ? xopt : (VAE_TN_XONE && xo->flags != nullptr))),
              "vae_gemm: uint8 counts as the operand need the 128 x 128 count kernels (vae_gemm_counts_u8_ok)");
  if (xo) { g.xone_flags = xo->flags; g.n_xone = xo->n_flags; g.row_sub = xo->row_sub; }
  g.xs_sum = g.xs_max = g.xs_flag = g.ws_sum = g.mean_out = g.wsum_out = nullptr;
  g.xs_n = g.gate = 0;
  if (rsc) {
    g.rmax0 = rsc->max0; g.rmax1 = rsc->max1; g.rmax_split = (int)(rsc->split < M ? rsc->split : M); g.rmax_chunks = rsc->chunks;
    g.rmax_stride = rsc->stride;
  }
  // bf16 storage exists in the one-plane split kernel only: the caller (step.hip) checks the shapes with vae_gemm_bf16io_ok
  VAE_REQUIRE(io == 0 || (precision == 1 && dma_ok && io > 0 && io <= 7 && vae_gemm_bf16_storage_ok(layout, io, M, N, K) &&
                          !((io & 4) && (slabs_out || accumulate || ldc % 8 != 0 || !vae_aligned16(C)))),
              "vae_gemm: bf16 operand / output storage is not available for this shape, layout or alignment");
  const int tiles_m = (g.M + bm - 1) / bm;
  g.tiles_m = tiles_m;
  g.tiles_n = (g.N + bn - 1) / bn;
  const int ktiles = (g.K + BK - 1) / BK;
  // both operands bf16 in HBM: the LDS-DMA kernel (64-deep K tiles, 16-byte chunks of 8 elements)
  static const int no_dma16 = getenv("VAE_GEMM_NO_DMA16") ? atoi(getenv("VAE_GEMM_NO_DMA16")) : 0;  // A/B knob
  // (short contractions -- the decoder heads, K = 256: four K tiles, then a long output stage -- stay on the split kernel,
  // whose three workgroups per CU overlap each other's output stage better than two: 220 vs 255 us at B = 4096)
  const bool dma16k = (io & 3) == 3 && dma_ok && !no_dma16 && K >= 1024 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && vae_aligned16(A) &&
                      vae_aligned16(B) && (!a_rc || M % 8 == 0) && (!b_rc || N % 8 == 0) && M >= 8 && N >= 8;
  int tiles_per_split = (ktiles + split - 1) / split;
  if (dma16k) tiles_per_split = (tiles_per_split + 1) & ~1;   // slabs of whole 64-deep tiles
  g.k_per_split = tiles_per_split * BK;
  split = (ktiles + tiles_per_split - 1) / tiles_per_split;  // drop empty slabs
  g.split = split;
  hipStream_t st = (hipStream_t)stream;
  if (slabs_out) {
    VAE_REQUIRE(workspace && workspace_bytes >= (int64_t)split * M * N * (int64_t)sizeof(float) && vae_aligned16(workspace),
                "vae_gemm_slabs: workspace too small or misaligned: need %lld bytes", (long long)((int64_t)split * M * N * 4));
    g.C = workspace;
    g.ldc = N;
    g.bias = nullptr;
    g.accumulate = 0;
    *slabs_out = split;
  } else if (split > 1) {
    VAE_REQUIRE(workspace && workspace_bytes >= (int64_t)split * M * N * (int64_t)sizeof(float),
                "vae_gemm: workspace too small: need %lld bytes (see vae_gemm_workspace_bytes)",
                (long long)((int64_t)split * M * N * 4));
    VAE_REQUIRE(vae_aligned16(workspace), "vae_gemm: workspace must be 16-byte aligned");
    g.C = workspace;
  }
  if (xo && xo->mode != 0) {
    VAE_REQUIRE(layout == NT && slabs_out && bm == 128 && bn == 128 && K % 4 == 0 && xo->xs_sum && xo->xs_max && xo->xs_flag && xo->ws_sum &&
                    tiles_m * (g.tiles_n < 4 ? g.tiles_n : 4) * split <= xo->xs_n && vae_aligned16(xo->xs_max) && vae_aligned16(xo->xs_flag) &&
                    (xo->mode == 1 || (xo->mean_out && xo->wsum_out && xo->flags == xo->xs_flag && xo->n_flags == xo->xs_n)),
                "vae_gemm: the optimistic count operand needs the pipelined 128 x 128 slab product (vae_gemm_counts_optimistic_ok)");
    g.xs_sum = xo->xs_sum; g.xs_max = xo->xs_max; g.xs_flag = xo->xs_flag; g.xs_n = xo->xs_n; g.ws_sum = xo->ws_sum;
    if (xo->mode == 2) { g.gate = 1; g.mean_out = xo->mean_out; g.wsum_out = xo->wsum_out; }
  }
  g.vecC = (N % 4 == 0) && vae_aligned16(g.C) && (split > 1 || slabs_out || ldc % 4 == 0);
  {  // grouped rasterisation of the row tiles (kernel comment): when both tile counts are large
    static const int gm_env = getenv("VAE_GEMM_GROUP_M") ? atoi(getenv("VAE_GEMM_GROUP_M")) : 8;
    if (dma_ok && split == 1 && gm_env > 0 && tiles_m >= 2 * gm_env && g.tiles_n >= 2 * gm_env) g.group_m = gm_env;
  }
  dim3 grid(tiles_m * g.tiles_n, split);
  // softmax statistics ride in the 16-byte-store epilogue of the split kernel: every 32-column block must take it
  if (stats && stats->m && stats->s && stats->cols > 0 && dma_ok && layout == NT && split == 1 && !slabs_out && !accumulate &&
      g.vecC && N % 32 == 0 && stats->cols <= N && bm == 128) {   // bm == 128: the tile shapes that carry the registers
    const int waves_n = bn >= 160 ? 1 : (bm == 160 ? 4 : 2);      // WAVES_N of the tile shapes launched below
    const int span = bn / waves_n;
    const int slots = (stats->cols + span - 1) / span;
    if ((int64_t)slots * M <= stats->capacity) {
      g.stat_m = stats->m; g.stat_s = stats->s; g.stat_cols = stats->cols;
      stats->slots = slots;
    }
  }
  if (dma_ok) {
    dim3 grid1(tiles_m * g.tiles_n * split);
    {   // Start-phase offset of the workgroups that share a CU (gemm_x3_kernel).  Workgroups launched together run in
        // lock-step -- all wait for loads, all convert, all multiply, all store at the same time, and the phases add up
        // (tools/micro/gemm_probe.hip ablations: 18 us of MFMA + 12 us of conversion + 19 us of memory = the 49.5 us of
        // the decoder-head forward).  The short-K NT product with its long output stage (three workgroups per CU, 1.3
        // generations) gains 10 % when the second and third workgroup of a CU start a third of a workgroup's lifetime
        // later each (cfg2, same box: 66.5 -> 60.3 us for 12-20 units, 61.1 at 8, 62.6 at 32).  The single-generation
        // products (two per CU: dgrad, both weight gradients, encoder forward) lose what the delay costs (measured at 8 and
        // 24 units each: step +0 ... +8 us), so only that class is staggered.  Unit: 1024 cycles.
      auto env = [](const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; };
      static const int st_decf = env("VAE_STAG_DECF", -1), st_encf = env("VAE_STAG_ENCF", 0), st_dgrad = env("VAE_STAG_DGRAD", 0),
                       st_decw = env("VAE_STAG_DECW", 0), st_encw = env("VAE_STAG_ENCW", 0);
      const bool decf = layout == NT && K <= 512 && f16x2 && !xo && !(bm == 128 && bn == 128) && bn <= 160;
      int rule = 0;
      if (decf && grid1.x > 256 && grid1.x <= 4 * 768) {
        rule = (int)((tiles_per_split * 4.8 + 11.0) / 3.0 + 0.5);   // lifetime ~ K steps x 4.8 k cycles + 11 k of output stage
        if (rule > 32) rule = 32;
      }
      g.stagger = layout == NT ? (K <= 512 ? (st_decf >= 0 ? st_decf : rule) : st_encf) : layout == NN ? st_dgrad : (rsc ? st_decw : st_encw);
    }
#define VAE_DMA_LAUNCH(LAY)                                                                              \
    do {                                                                                                   \
      if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm32_dma_kernel<LAY, 128, 128, 2>), grid1, dim3(256), 0, st, g); \
      else if (bn == 160)         hipLaunchKernelGGL((gemm32_dma_kernel<LAY, 128, 160, 4>), grid1, dim3(256), 0, st, g); \
      else                        hipLaunchKernelGGL((gemm32_dma_kernel<LAY, 160, 128, 1>), grid1, dim3(256), 0, st, g); \
    } while (0)
#define VAE_X3_LAUNCH(LAY, NPL)                                                                               \
    do {                                                                                                   \
      if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm_x3_kernel<LAY, 128, 128, 2, NPL>), grid1, dim3(256), 0, st, g); \
      else if (bn == 160)         hipLaunchKernelGGL((gemm_x3_kernel<LAY, 128, 160, 4, NPL>), grid1, dim3(256), 0, st, g); \
      else                        hipLaunchKernelGGL((gemm_x3_kernel<LAY, 160, 128, 1, NPL>), grid1, dim3(256), 0, st, g); \
    } while (0)
    static const int use_x3 = getenv("VAE_GEMM_X3") ? atoi(getenv("VAE_GEMM_X3")) : 1;  // 0: fp32 MFMA pipeline
#define VAE_X3_LAUNCH_IO(LAY, IOF)                                                                          \
    do {                                                                                                   \
      if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm_x3_kernel<LAY, 128, 128, 2, 1, IOF>), grid1, dim3(256), 0, st, g); \
      else if (bn == 160)         hipLaunchKernelGGL((gemm_x3_kernel<LAY, 128, 160, 4, 1, IOF>), grid1, dim3(256), 0, st, g); \
      else                        hipLaunchKernelGGL((gemm_x3_kernel<LAY, 160, 128, 1, 1, IOF>), grid1, dim3(256), 0, st, g); \
    } while (0)
#define VAE_DMA16_LAUNCH(LAY, CB)                                                                           \
    do {                                                                                                   \
      if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm16_dma_kernel<LAY, 128, 128, 2, CB>), grid1, dim3(256), 0, st, g); \
      else if (bn == 160)         hipLaunchKernelGGL((gemm16_dma_kernel<LAY, 128, 160, 4, CB>), grid1, dim3(256), 0, st, g); \
      else                        hipLaunchKernelGGL((gemm16_dma_kernel<LAY, 160, 128, 1, CB>), grid1, dim3(256), 0, st, g); \
    } while (0)
    if (dma16k) {
      if (layout == NT && (io & 4)) VAE_DMA16_LAUNCH(NT, true);
      else if (layout == NT) VAE_DMA16_LAUNCH(NT, false);
      else if (layout == NN) VAE_DMA16_LAUNCH(NN, false);
      else VAE_DMA16_LAUNCH(TN, false);
    } else if (io) {   // bf16 storage of operands / of the output (vae_gemm_bf16_storage_ok lists the combinations)
      if (layout == NT && io == 1) VAE_X3_LAUNCH_IO(NT, 1);
      else if (layout == NT && io == 3) VAE_X3_LAUNCH_IO(NT, 3);
      else if (layout == NT && io == 4) VAE_X3_LAUNCH_IO(NT, 4);
      else if (layout == NT) VAE_X3_LAUNCH_IO(NT, 7);
      else if (layout == NN && io == 1) VAE_X3_LAUNCH_IO(NN, 1);
      else if (layout == NN) VAE_X3_LAUNCH_IO(NN, 3);
      else if (io == 1) VAE_X3_LAUNCH_IO(TN, 1);
      else if (io == 2) VAE_X3_LAUNCH_IO(TN, 2);
      else VAE_X3_LAUNCH_IO(TN, 3);
    } else if (precision == 1) {
      if (layout == NT) VAE_X3_LAUNCH(NT, 1);
      else if (layout == NN) VAE_X3_LAUNCH(NN, 1);
      else VAE_X3_LAUNCH(TN, 1);
    } else if (xopt && a_u8) {   // ... straight from the uint8 store
      hipLaunchKernelGGL((gemm_x3_kernel<NT, 128, 128, 2, 2, 8, false, 3>), grid1, dim3(256), 0, st, g);
    } else if (xopt) {   // counts taken as exact in one unscaled f16 term, checked on the way (GemmArgs::xs_*)
      hipLaunchKernelGGL((gemm_x3_kernel<NT, 128, 128, 2, 2, 0, false, 3>), grid1, dim3(256), 0, st, g);
    } else if (xo && xo->flags && layout == NT) {   // two f16 terms, the count operand A on ONE term when every count is exact in it
      // (the TN form -- counts = B -- follows below for the 128 x 128 tiles.  Before the pipelined loop asked for its
      // staging registers again right after converting them it was SLOWER than the plain two-term kernel, 41.8 vs 37.5 us
      // on the cfg2 weight gradient -- bound by the latency of its K steps; with the reload it is 32.9 vs 35.9.)
      if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm_x3_kernel<NT, 128, 128, 2, 2, 0, false, 1>), grid1, dim3(256), 0, st, g);
      else if (bn == 160)         hipLaunchKernelGGL((gemm_x3_kernel<NT, 128, 160, 4, 2, 0, false, 1>), grid1, dim3(256), 0, st, g);
      else                        hipLaunchKernelGGL((gemm_x3_kernel<NT, 160, 128, 1, 2, 0, false, 1>), grid1, dim3(256), 0, st, g);
    } else if (VAE_TN_XONE && xo && xo->flags && layout == TN && bm == 128 && bn == 128) {   // ... and the count operand B of the weight gradient
      if (b_u8) hipLaunchKernelGGL((gemm_x3_kernel<TN, 128, 128, 2, 2, 16, false, 2>), grid1, dim3(256), 0, st, g);
      else      hipLaunchKernelGGL((gemm_x3_kernel<TN, 128, 128, 2, 2, 0, false, 2>), grid1, dim3(256), 0, st, g);
    } else if (rsc) {   // two f16 terms, one scale per product row (the decoder weight gradient)
      if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm_x3_kernel<TN, 128, 128, 2, 2, 0, true>), grid1, dim3(256), 0, st, g);
      else if (bn == 160)         hipLaunchKernelGGL((gemm_x3_kernel<TN, 128, 160, 4, 2, 0, true>), grid1, dim3(256), 0, st, g);
      else                        hipLaunchKernelGGL((gemm_x3_kernel<TN, 160, 128, 1, 2, 0, true>), grid1, dim3(256), 0, st, g);
    } else if (f16x2 && bn == 320) {
      hipLaunchKernelGGL((gemm_x3_kernel<NT, 128, 320, 4, 2>), grid1, dim3(256), 0, st, g);
    } else if (f16x2) {
      if (layout == NT) VAE_X3_LAUNCH(NT, 2);
      else if (layout == NN) VAE_X3_LAUNCH(NN, 2);
      else VAE_X3_LAUNCH(TN, 2);
    } else if (use_x3) {
      if (layout == NT) VAE_X3_LAUNCH(NT, 3);
      else if (layout == NN) VAE_X3_LAUNCH(NN, 3);
      else VAE_X3_LAUNCH(TN, 3);
    } else if (layout == NT) VAE_DMA_LAUNCH(NT);
    else if (layout == NN) VAE_DMA_LAUNCH(NN);
    else VAE_DMA_LAUNCH(TN);
#undef VAE_DMA_LAUNCH
#undef VAE_X3_LAUNCH
#undef VAE_X3_LAUNCH_IO
#undef VAE_DMA16_LAUNCH
  } else if (precision == 0) {
    if (layout == NT) launch_t<NT, 0>(g, norm, grid, st, plan.tile);
    else if (layout == NN) launch_t<NN, 0>(g, false, grid, st, plan.tile);
    else launch_t<TN, 0>(g, norm, grid, st, plan.tile);
  } else if (precision == 1) {
    if (layout == NT) launch_t<NT, 1>(g, norm, grid, st, plan.tile);
    else if (layout == NN) launch_t<NN, 1>(g, false, grid, st, plan.tile);
    else launch_t<TN, 1>(g, norm, grid, st, plan.tile);
  } else {
    if (layout == NT) launch_t<NT, 2>(g, norm, grid, st, plan.tile);
    else if (layout == NN) launch_t<NN, 2>(g, false, grid, st, plan.tile);
    else launch_t<TN, 2>(g, norm, grid, st, plan.tile);
  }
  VAE_CHECK_LAUNCH("vae_gemm");
  if (split > 1 && !slabs_out) {
    const int64_t total = M * N;
    const bool vec = (N % 4 == 0) && (ldc % 4 == 0) && vae_aligned16(C) && (!bias || vae_aligned16(bias)) &&
                     (total % 4 == 0);
    int blocks = (int)(((vec ? total / 4 : total) + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (vec)
      hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, workspace, split, g.M, g.N, bias, C,
                         ldc, accumulate, g.row_sub);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, workspace, split, g.M, g.N, bias,
                         C, ldc, accumulate, g.row_sub);
    VAE_CHECK_LAUNCH("vae_gemm(splitk_reduce)");
  }
  return VAE_OK;
}
}  // namespace

namespace {
// scale[0] = 2^e with max|X| * 2^e in [2^13, 2^14) (1 for an all-zero matrix): the factor that puts an operand of
// the two-f16-term product in the upper part of f16's range.  Two stages, fixed order.
__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int cols,
                                                             float* __restrict__ partial) {
  __shared__ float red[4];
  float m = 0.f;
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    m = fmaxf(m, fabsf(X[r * ld + (i - r * cols)]));
  }
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = m;
}
__global__ __launch_bounds__(256) void absmax_scale_kernel(const float* __restrict__ partial, int n, float* __restrict__ scale) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) {
    int e = 0;
    float s = 1.f;
    if (m > 0.f && m < INFINITY) {
      frexpf(m, &e);               // m = f * 2^e, f in [0.5, 1)
      s = ldexpf(1.f, 14 - e);     // m * s in [2^13, 2^14)
    }
    scale[0] = s;
  }
}
__global__ __launch_bounds__(256) void absmax_raw_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int cols,
                                                         float* __restrict__ slot) {
  __shared__ float red[4];
  float m = 0.f;
  const int64_t total = rows * cols;
  if ((cols & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0) {
    const int c4 = cols >> 2;
    const int64_t t4 = rows * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t4; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / c4;
      const float4 v = *reinterpret_cast<const float4*>(X + r * ld + 4 * (i - r * c4));
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / cols;
      m = fmaxf(m, fabsf(X[r * ld + (i - r * cols)]));
    }
  }
  m = block_max<4>(m, red);
  if (threadIdx.x == 0) slot[blockIdx.x] = m;
}
}  // namespace

// ---- internal (internal.h): used by step.hip --------------------------------------------------
// partial[0 .. VAE_ABSMAX_PARTIALS) = per-workgroup maxima of |X| (every entry written)
int vae_absmax_partials(const float* X, int64_t rows, int64_t cols, int64_t ld, float* partial, void* stream) {
  VAE_REQUIRE(X && partial && rows > 0 && cols > 0 && cols < (1 << 30) && ld >= cols, "vae_absmax_partials: bad arguments");
  hipLaunchKernelGGL(absmax_raw_kernel, dim3(VAE_ABSMAX_PARTIALS), dim3(256), 0, (hipStream_t)stream, X, ld, rows, (int)cols,
                     partial);
  VAE_CHECK_LAUNCH("vae_absmax_partials");
  return VAE_OK;
}
// vae_gemm / vae_gemm_slabs on two scaled f16 terms, the scales derived in the kernel from the partial maxima of |A|, |B|
int vae_gemm_f16x2(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* max_a, int n_max_a,
                   const float* B, int64_t ldb, const float* max_b, int n_max_b, float* C, int64_t ldc, const float* bias,
                   int accumulate, float* workspace, int64_t workspace_bytes, void* stream, int* slabs_out,
                   VaeGemmRowStats* stats, const VaeGemmCounts* counts) {
  VAE_REQUIRE(max_a && max_b && n_max_a > 0 && n_max_b > 0, "vae_gemm_f16x2: operand maxima missing");
  return gemm_impl(layout, 3, M, N, K, A, lda, B, ldb, slabs_out ? nullptr : C, slabs_out ? N : ldc, slabs_out ? nullptr : bias,
                   nullptr, slabs_out ? 0 : accumulate, workspace, workspace_bytes, stream, slabs_out, max_a, max_b, n_max_a,
                   n_max_b, stats, 0, nullptr, counts);
}
// whether the NT count-operand product of this shape (left as slabs) can run optimistically: the pipelined 128 x 128 loop,
// and room for its per-workgroup entries in arrays of n_stat floats
bool vae_gemm_counts_optimistic_ok(int64_t M, int64_t N, int64_t K, int n_stat) {
  if (!VAE_X3_PIPE || !VAE_X3_RELOAD || K % 4 != 0 || !vae_gemm_tile_pipeline_ok(NT, M, N, K)) return false;
  if (make_plan((int)M, (int)N, (int)K, false).tile != 128) return false;
  const DmaPlan d = make_dma_plan((int)M, (int)N, (int)K, false);
  if (d.bm != 128 || d.bn != 128) return false;
  const int ktiles = (int)((K + BK - 1) / BK), tps = (ktiles + d.split - 1) / d.split, split = (ktiles + tps - 1) / tps;
  const int64_t tn = (N + 127) / 128;
  return (int64_t)((M + 127) / 128) * (tn < 4 ? tn : 4) * split <= n_stat;
}
// whether the TN count-operand product (M, N, K) = (rows of the first-layer weight, genes, cells) runs on the 128 x 128 tiles
// without a K split -- where the kernel exists that reads its count operand from the uint8 store
bool vae_gemm_counts_u8_tn_ok(int64_t M, int64_t N, int64_t K) {
  if (!VAE_TN_XONE || !vae_gemm_tile_pipeline_ok(TN, M, N, K) || make_plan((int)M, (int)N, (int)K, false).tile != 128) return false;
  const DmaPlan d = make_dma_plan((int)M, (int)N, (int)K, false);
  return d.bm == 128 && d.bn == 128;
}
// whether fp32 operands with 16-byte aligned rows of this shape run the 128-wide tile pipeline (split / LDS-DMA kernels)
bool vae_gemm_tile_pipeline_ok(int layout, int64_t M, int64_t N, int64_t K) {
  return vae_gemm_bf16_storage_ok(layout, 1, M, N, K) != 0;   // the same shape conditions
}
// C (M, N) = A^T B on two f16 terms with ONE SCALE PER ROW m of the product (per column of the k-major A), from per-chunk
// column maxima of |A| (VaeGemmRowScale), and one scale for B from its partial maxima: the decoder weight gradient
int vae_gemm_f16x2_rowscaled(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const VaeGemmRowScale* rsc, const float* B,
                             int64_t ldb, const float* max_b, int n_max_b, float* C, int64_t ldc, float* workspace,
                             int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(rsc && max_b && n_max_b > 0, "vae_gemm_f16x2_rowscaled: maxima missing");
  return gemm_impl(TN, 3, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, workspace, workspace_bytes, stream, nullptr,
                   nullptr, max_b, 0, n_max_b, nullptr, 0, rsc);
}
// vae_gemm that may also leave the softmax statistics of its output rows (see VaeGemmRowStats)
int vae_gemm_rowstats(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, float* workspace, int64_t workspace_bytes,
                      VaeGemmRowStats* stats, void* stream) {
  return gemm_impl(layout, precision, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, 0, workspace, workspace_bytes, stream,
                   nullptr, nullptr, nullptr, 0, 0, stats);
}

// bf16-mode product with ONE operand or the output stored as bf16 in HBM (io: VAE_IO_*).  n_slabs != null: split-K slabs
// (fp32) stay in the workspace as for vae_gemm_slabs (not with VAE_IO_C_BF16).  stats: as vae_gemm_rowstats.
int vae_gemm_bf16io(int layout, int io, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                    void* C, int64_t ldc, const float* bias, float* workspace, int64_t workspace_bytes, int* n_slabs,
                    VaeGemmRowStats* stats, void* stream) {
  return gemm_impl(layout, 1, M, N, K, (const float*)A, lda, (const float*)B, ldb, n_slabs ? nullptr : (float*)C, n_slabs ? N : ldc,
                   n_slabs ? nullptr : bias, nullptr, 0, workspace, workspace_bytes, stream, n_slabs, nullptr, nullptr, 0, 0, stats,
                   io);
}
namespace {
// fp32 -> bf16 (nearest even) of up to two dense arrays in one launch: the bf16 images of the gene-sized weight matrices
// (and of small activations) that a bf16-mode step's products stream instead of the fp32 originals -- half the L2 -> LDS
// bytes of a product whose tiles are re-read by many workgroups.  n0, n1 multiples of 4; blocks [0, split) take array 0.
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s0, uint16_t* __restrict__ d0, int64_t n0,
                                                        const float* __restrict__ s1, uint16_t* __restrict__ d1, int64_t n1,
                                                        int split) {
  const bool first = (int)blockIdx.x < split;
  const float* s = first ? s0 : s1;
  uint16_t* d = first ? d0 : d1;
  const int64_t n4 = (first ? n0 : n1) >> 2;
  const int nb = first ? split : (int)gridDim.x - split, bi = first ? (int)blockIdx.x : (int)blockIdx.x - split;
  for (int64_t i = (int64_t)bi * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256) {
    const float4 v = reinterpret_cast<const float4*>(s)[i];
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    reinterpret_cast<bf16x4*>(d)[i] = o;
  }
}
}  // namespace
// d0[i] = bf16(s0[i]), i < n0, and (s1 != null) d1[i] = bf16(s1[i]), i < n1: dense, 16-byte aligned, n % 4 == 0
int vae_cast_bf16(const float* s0, void* d0, int64_t n0, const float* s1, void* d1, int64_t n1, void* stream) {
  VAE_REQUIRE(s0 && d0 && n0 > 0 && n0 % 4 == 0 && vae_aligned16(s0) && (reinterpret_cast<uintptr_t>(d0) & 7u) == 0,
              "vae_cast_bf16: bad first array");
  const bool two = s1 != nullptr;
  VAE_REQUIRE(!two || (d1 && n1 > 0 && n1 % 4 == 0 && vae_aligned16(s1) && (reinterpret_cast<uintptr_t>(d1) & 7u) == 0),
              "vae_cast_bf16: bad second array");
  auto blocks = [](int64_t n) { int64_t b = (n / 4 + 255) / 256 / 4; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); };
  const int b0 = blocks(n0), b1 = two ? blocks(n1) : 0;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(b0 + b1), dim3(256), 0, (hipStream_t)stream, s0, (uint16_t*)d0, n0, s1, (uint16_t*)d1,
                     two ? n1 : 0, b0);
  VAE_CHECK_LAUNCH("vae_cast_bf16");
  return VAE_OK;
}
extern "C" int vae_gemm_bf16_storage_ok(int layout, int io, int64_t M, int64_t N, int64_t K) {
  if (layout < 0 || layout > 2 || M < 4 || N < 4 || K < 4 || M >= (1 << 30) || N >= (1 << 30) || K >= (1 << 30)) return 0;
  // combinations with a kernel behind them: NT {A, A+B, C, A+B+C}, NN {A, A+B}, TN {A, B, A+B}
  const bool combo = layout == NT ? (io == 1 || io == 3 || io == 4 || io == 7) : layout == NN ? (io == 1 || io == 3) : (io >= 1 && io <= 3);
  if (!combo) return 0;
  if (make_plan((int)M, (int)N, (int)K, false).tile != 128 || K % 4 != 0) return 0;
  if ((layout == TN && M % 4 != 0) || (layout != NT && N % 4 != 0)) return 0;
  if ((io & 4) && N % 8 != 0) return 0;
  return 1;
}
extern "C" int vae_gemm_bf16_storage(int layout, int io, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                                     int64_t ldb, void* C, int64_t ldc, const float* bias, float* workspace,
                                     int64_t workspace_bytes, void* stream) {
  return vae_gemm_bf16io(layout, io, M, N, K, A, lda, B, ldb, C, ldc, bias, workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int64_t vae_absmax_workspace_bytes(void) { return 1024 * (int64_t)sizeof(float); }

extern "C" int vae_absmax_scale(const float* X, int64_t rows, int64_t cols, int64_t ld, float* scale_out, float* workspace,
                                int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(X && scale_out && rows > 0 && cols > 0 && cols < (1 << 30) && ld >= cols, "vae_absmax_scale: bad arguments");
  VAE_REQUIRE(workspace && workspace_bytes >= vae_absmax_workspace_bytes(), "vae_absmax_scale: needs vae_absmax_workspace_bytes()");
  int blocks = (int)((rows * cols + 4095) / 4096);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(absmax_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ld, rows, (int)cols, workspace);
  VAE_CHECK_LAUNCH("vae_absmax_scale");
  hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, blocks, scale_out);
  VAE_CHECK_LAUNCH("vae_absmax_scale(finish)");
  return VAE_OK;
}

extern "C" int vae_gemm_scaled(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* scale_a,
                               const float* B, int64_t ldb, const float* scale_b, float* C, int64_t ldc, const float* bias,
                               int accumulate, float* workspace, int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(scale_a && scale_b, "vae_gemm_scaled: null scale");
  return gemm_impl(layout, 3, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, accumulate, workspace, workspace_bytes, stream,
                   nullptr, scale_a, scale_b);
}

extern "C" int vae_gemm_counts(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* max_a,
                               int64_t n_max_a, const float* B, int64_t ldb, const float* max_b, int64_t n_max_b, float* C, int64_t ldc,
                               const float* flags, int64_t n_flags, const float* row_sub, float* workspace, int64_t workspace_bytes,
                               void* stream) {
  VAE_REQUIRE(layout == NT || layout == TN, "vae_gemm_counts: layout must be VAE_NT (counts = A) or VAE_TN (counts = B)");
  VAE_REQUIRE((layout == TN || flags) && n_flags >= 0 && n_flags < (1 << 30) && (!flags || n_flags > 0) && n_max_a > 0 && n_max_a < (1 << 30) && n_max_b > 0 && n_max_b < (1 << 30) &&
                  max_a && max_b, "vae_gemm_counts: maxima / flags missing");
  VAE_REQUIRE(M > 0 && N > 0 && K > 0 && vae_gemm_tile_pipeline_ok(layout, M, N, K), "vae_gemm_counts: M=%lld N=%lld K=%lld is outside the tile pipeline",
              (long long)M, (long long)N, (long long)K);
  VAE_REQUIRE(A && B && C && lda % 4 == 0 && ldb % 4 == 0 && vae_aligned16(A) && vae_aligned16(B), "vae_gemm_counts: operands need 16-byte aligned rows");
  const VaeGemmCounts xc{flags, (int)n_flags, row_sub};
  return vae_gemm_f16x2(layout, M, N, K, A, lda, max_a, (int)n_max_a, B, ldb, max_b, (int)n_max_b, C, ldc, nullptr, 0, workspace,
                        workspace_bytes, stream, nullptr, nullptr, &xc);
}

extern "C" int vae_gemm_rowscaled_ok(int64_t M, int64_t N, int64_t K) { return vae_gemm_tile_pipeline_ok(TN, M, N, K) ? 1 : 0; }

extern "C" int vae_gemm_rowscaled(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* col_max, int64_t n_chunks,
                                  int64_t ld_max, const float* B, int64_t ldb, const float* b_max, int64_t n_b_max, float* C,
                                  int64_t ldc, float* workspace, int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(col_max && b_max && n_chunks > 0 && n_chunks < (1 << 20) && ld_max >= M && n_b_max > 0 && n_b_max < (1 << 24),
              "vae_gemm_rowscaled: maxima missing or misshapen");
  VAE_REQUIRE(M > 0 && N > 0 && K > 0 && vae_gemm_rowscaled_ok(M, N, K), "vae_gemm_rowscaled: M=%lld N=%lld K=%lld is outside the tile pipeline",
              (long long)M, (long long)N, (long long)K);
  VAE_REQUIRE(A && B && C && lda % 4 == 0 && ldb % 4 == 0 && vae_aligned16(A) && vae_aligned16(B), "vae_gemm_rowscaled: operands need 16-byte aligned rows");
  const VaeGemmRowScale rsc{col_max, nullptr, M, (int)n_chunks, ld_max};
  return vae_gemm_f16x2_rowscaled(M, N, K, A, lda, &rsc, B, ldb, b_max, (int)n_b_max, C, ldc, workspace, workspace_bytes, stream);
}

extern "C" int vae_gemm_scaled_slabs(int layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                     const float* scale_a, const float* B, int64_t ldb, const float* scale_b,
                                     float* workspace, int64_t workspace_bytes, int* n_slabs, void* stream) {
  VAE_REQUIRE(n_slabs && scale_a && scale_b, "vae_gemm_scaled_slabs: null argument");
  return gemm_impl(layout, 3, M, N, K, A, lda, B, ldb, nullptr, N, nullptr, nullptr, 0, workspace, workspace_bytes, stream,
                   n_slabs, scale_a, scale_b);
}

extern "C" int vae_gemm(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                        const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, const float* x_mean,
                        int accumulate, float* workspace, int64_t workspace_bytes, void* stream) {
  return gemm_impl(layout, precision, M, N, K, A, lda, B, ldb, C, ldc, bias, x_mean, accumulate, workspace,
                   workspace_bytes, stream, nullptr);
}

// ---- vae_gemm for callers that cannot allocate (include/vae_amd.h) --------------------------------------------------
// The split-K slabs live in a file-scope device array instead of a caller workspace.  `scratch` is a host-side stand-in
// in host code, never a device address: every GPU holds its own copy at its own address, looked up once per device
// with hipGetSymbolAddress and kept; only that pointer goes to kernels.
namespace {
constexpr int64_t NOALLOC_BYTES = 40ll << 20;   // covers the cfg2 encoder forward (42 slabs x 512 x 384 floats = 33.0 MB)
constexpr int NOALLOC_MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) float scratch[NOALLOC_BYTES / sizeof(float)];
namespace {
float* g_scratch_ptr[NOALLOC_MAX_DEVICES];       // per device, null until the first call on that device
std::mutex g_scratch_mutex;

// stream != null: the lookup is keyed on the STREAM's device, which must be the current one (as for any launch on it)
int noalloc_scratch(float** out, void* stream) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return vae_set_error(VAE_ERR_LAUNCH, "vae_gemm_noalloc: hipGetDevice failed (no GPU?)");
  if (stream) {
    int sdev = -1;
    if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && sdev != dev)
      return vae_set_error(VAE_ERR_ARG, "vae_gemm_noalloc: the stream belongs to device %d, the current device is %d", sdev, dev);
  }
  VAE_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "vae_gemm_noalloc: device ordinal %d outside [0, %d)", dev, NOALLOC_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // the address of THIS device's copy
    if (e != hipSuccess || !p)
      return vae_set_error(VAE_ERR_LAUNCH, "vae_gemm_noalloc: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
    g_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_scratch_ptr[dev];
  return VAE_OK;
}
}  // namespace

// Resolves the current device's scratch address now (idempotent): callers that will run vae_gemm_noalloc under HIP graph
// capture call this once per device BEFORE capturing, so that no runtime query happens inside the capture.
extern "C" int vae_gemm_noalloc_init(void) {
  float* p = nullptr;
  return noalloc_scratch(&p, nullptr);
}

extern "C" int64_t vae_gemm_noalloc_max_bytes(void) { return NOALLOC_BYTES; }

extern "C" int vae_gemm_noalloc(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, const float* x_mean,
                                int accumulate, void* stream) {
  VAE_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30),
              "vae_gemm_noalloc: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  const int64_t need = vae_gemm_workspace_bytes(M, N, K);
  VAE_REQUIRE(need <= NOALLOC_BYTES,
              "vae_gemm_noalloc: M=%lld N=%lld K=%lld needs %lld bytes of split-K scratch, the static scratch holds %lld "
              "(use vae_gemm with a caller workspace)", (long long)M, (long long)N, (long long)K, (long long)need,
              (long long)NOALLOC_BYTES);
  float* ws = nullptr;
  if (need > 0) {   // slabs are fully overwritten before they are read: nothing to clear between calls
    const int rc = noalloc_scratch(&ws, stream);
    if (rc != VAE_OK) return rc;
  }
  return gemm_impl(layout, precision, M, N, K, A, lda, B, ldb, C, ldc, bias, x_mean, accumulate, ws, need, stream, nullptr);
}

extern "C" int64_t vae_gemm_slabs_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const int64_t b = vae_gemm_workspace_bytes(M, N, K);
  return b > 0 ? b : M * N * (int64_t)sizeof(float);
}

extern "C" int vae_gemm_slabs(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                              const float* B, int64_t ldb, float* workspace, int64_t workspace_bytes, int* n_slabs,
                              void* stream) {
  VAE_REQUIRE(n_slabs, "vae_gemm_slabs: n_slabs is null");
  return gemm_impl(layout, precision, M, N, K, A, lda, B, ldb, nullptr, N, nullptr, nullptr, 0, workspace,
                   workspace_bytes, stream, n_slabs);
}
