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
// Tile: 128x128x32 per 256-thread workgroup, 4 waves as 2x2, each wave a 64x64 output
// (2x2 MFMA 32x32 accumulators).  Global->register prefetch of tile t+1 overlaps the MFMAs
// of tile t; skinny shapes (K = genes) are split over K into fp32 slabs that a second
// kernel sums in a fixed order (deterministic, no float atomics).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
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
  int accumulate;      // C += (only when split == 1)
  int tiles_n;
};

// ---- global -> register tile loaders (4 float4 per thread per operand) ------------------
// KC: operand stored (rows, K) row-major, k contiguous.  chunk c: row = c>>3, k4 = c&7.
template <bool NORM>
__device__ __forceinline__ void load_kc(const float* __restrict__ X, int64_t ld, int r0, int rmax, int k0, int kend,
                                        int vec, const float* __restrict__ mean, float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = r0 + (c >> 3), k = k0 + (c & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rmax && k < kend) {
      const float* p = X + (int64_t)r * ld + k;
      bool m0 = true, m1 = k + 1 < kend, m2 = k + 2 < kend, m3 = k + 3 < kend;
      if (vec && m3) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        v.x = p[0];
        if (m1) v.y = p[1];
        if (m2) v.z = p[2];
        if (m3) v.w = p[3];
      }
      if (NORM) {
        const float mu = mean[r];
        v.x = m0 ? v.x / mu - 1.f : 0.f;
        v.y = m1 ? v.y / mu - 1.f : 0.f;
        v.z = m2 ? v.z / mu - 1.f : 0.f;
        v.w = m3 ? v.w / mu - 1.f : 0.f;
      }
    }
    reg[i] = v;
  }
}
// RC: operand stored (K, rows) row-major, row index contiguous.  chunk c: k = c>>5, r4 = c&31.
template <bool NORM>
__device__ __forceinline__ void load_rc(const float* __restrict__ X, int64_t ld, int r0, int rmax, int k0, int kend,
                                        int vec, const float* __restrict__ mean, float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int k = k0 + (c >> 5), r = r0 + (c & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend && r < rmax) {
      const float* p = X + (int64_t)k * ld + r;
      bool m1 = r + 1 < rmax, m2 = r + 2 < rmax, m3 = r + 3 < rmax;
      if (vec && m3) {
        v = *reinterpret_cast<const float4*>(p);
      } else {
        v.x = p[0];
        if (m1) v.y = p[1];
        if (m2) v.z = p[2];
        if (m3) v.w = p[3];
      }
      if (NORM) {
        const float mu = mean[k];
        v.x = v.x / mu - 1.f;
        v.y = m1 ? v.y / mu - 1.f : 0.f;
        v.z = m2 ? v.z / mu - 1.f : 0.f;
        v.w = m3 ? v.w / mu - 1.f : 0.f;
      }
    }
    reg[i] = v;
  }
}

// ---- fp32 LDS image: k-major [BK][LD] floats; operand reads are conflict-free ds_read_b32 ----
constexpr int LD32_KC = 130;  // filled by 4-byte scatter writes (2-way on writes: free)
constexpr int LD32_RC = 132;  // filled by 16-byte row writes (needs LD % 4 == 0)

__device__ __forceinline__ void store32_kc(float* s, const float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = c >> 3, k = (c & 7) * 4;
    s[(k + 0) * LD32_KC + r] = reg[i].x;
    s[(k + 1) * LD32_KC + r] = reg[i].y;
    s[(k + 2) * LD32_KC + r] = reg[i].z;
    s[(k + 3) * LD32_KC + r] = reg[i].w;
  }
}
__device__ __forceinline__ void store32_rc(float* s, const float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int k = c >> 5, r = (c & 31) * 4;
    *reinterpret_cast<float4*>(&s[k * LD32_RC + r]) = reg[i];
  }
}

// ---- bf16 LDS image: row-major [128][LD16] bf16, k contiguous; 80-byte rows make the
//      ds_read_b128 fragment reads conflict-free (5*row mod 16 is a bijection on 16 rows) ----
constexpr int LD16 = 40;

__device__ __forceinline__ void store16_kc(__bf16* s, const float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int r = c >> 3, k = (c & 7) * 4;
    bf16x4 v;
    v[0] = (__bf16)reg[i].x; v[1] = (__bf16)reg[i].y; v[2] = (__bf16)reg[i].z; v[3] = (__bf16)reg[i].w;
    *reinterpret_cast<bf16x4*>(&s[r * LD16 + k]) = v;
  }
}
__device__ __forceinline__ void store16_rc(__bf16* s, const float4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int k = c >> 5, r = (c & 31) * 4;
    s[(r + 0) * LD16 + k] = (__bf16)reg[i].x;
    s[(r + 1) * LD16 + k] = (__bf16)reg[i].y;
    s[(r + 2) * LD16 + k] = (__bf16)reg[i].z;
    s[(r + 3) * LD16 + k] = (__bf16)reg[i].w;
  }
}

constexpr int SMEM32 = BK * (LD32_RC > LD32_KC ? LD32_RC : LD32_KC);  // floats per operand
constexpr int SMEM16 = 128 * LD16;                                       // bf16 per operand

template <int LAYOUT, int PREC, bool NORM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  constexpr bool A_KC = (LAYOUT != TN);
  constexpr bool B_KC = (LAYOUT == NT);
  constexpr bool NORM_A = NORM && LAYOUT == NT;
  constexpr bool NORM_B = NORM && LAYOUT == TN;

  __shared__ __attribute__((aligned(16))) float smem[PREC == 0 ? 2 * SMEM32 : SMEM16];
  float* sA32 = smem;
  float* sB32 = smem + SMEM32;
  __bf16* sA16 = reinterpret_cast<__bf16*>(smem);
  __bf16* sB16 = sA16 + SMEM16;

  const int tile = blockIdx.x;
  const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
  const int s = blockIdx.y;
  const int kbeg = s * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], rb[4];
  auto fetch = [&](int k0) {
    if (A_KC) load_kc<NORM_A>(g.A, g.lda, m0, g.M, k0, kend, g.vecA, g.mean, ra);
    else      load_rc<false>(g.A, g.lda, m0, g.M, k0, kend, g.vecA, nullptr, ra);
    if (B_KC) load_kc<false>(g.B, g.ldb, n0, g.N, k0, kend, g.vecB, nullptr, rb);
    else      load_rc<NORM_B>(g.B, g.ldb, n0, g.N, k0, kend, g.vecB, g.mean, rb);
  };

  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile's fragment reads are done
    if (PREC == 0) {
      if (A_KC) store32_kc(sA32, ra); else store32_rc(sA32, ra);
      if (B_KC) store32_kc(sB32, rb); else store32_rc(sB32, rb);
    } else {
      if (A_KC) store16_kc(sA16, ra); else store16_rc(sA16, ra);
      if (B_KC) store16_kc(sB16, rb); else store16_rc(sB16, rb);
    }
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);  // in flight during the MFMAs below

    if (PREC == 0) {
      constexpr int LDA = A_KC ? LD32_KC : LD32_RC;
      constexpr int LDB = B_KC ? LD32_KC : LD32_RC;
      const float* pa = sA32 + h * LDA + wm * 64 + l31;
      const float* pb = sB32 + h * LDB + wn * 64 + l31;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const float a0 = pa[kk * 2 * LDA], a1 = pa[kk * 2 * LDA + 32];
        const float b0 = pb[kk * 2 * LDB], b1 = pb[kk * 2 * LDB + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    } else {
      const __bf16* pa = sA16 + (wm * 64 + l31) * LD16 + h * 8;
      const __bf16* pb = sB16 + (wn * 64 + l31) * LD16 + h * 8;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(pa + ks * 16);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(pa + 32 * LD16 + ks * 16);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(pb + ks * 16);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(pb + 32 * LD16 + ks * 16);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }

  // C/D map of the 32x32 MFMAs: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = g.C;
  int64_t ldc = g.ldc;
  if (g.split > 1) {
    out = g.C + (int64_t)s * g.M * g.N;
    ldc = g.N;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + l31;
      if (n >= g.N) continue;
      const float bv = (g.split == 1 && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M) {
          float* p = out + (int64_t)m * ldc + n;
          float v = acc[mi][ni][r] + bv;
          if (g.split == 1 && g.accumulate) v += *p;
          *p = v;
        }
      }
    }
}

// Sums the split-K slabs in slab order, adds the bias, writes (or accumulates into) C.
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int split, int M, int N, const float* __restrict__ bias,
                                     float* __restrict__ C, int64_t ldc, int accumulate) {
  const int64_t total = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    float v = 0.f;
    for (int s = 0; s < split; ++s) v += ws[(int64_t)s * total + i];
    if (bias) v += bias[n];
    float* p = C + (int64_t)m * ldc + n;
    if (accumulate) v += *p;
    *p = v;
  }
}

int choose_split(int M, int N, int K) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles >= 192) return 1;
  const int ktiles = (K + BK - 1) / BK;
  int split = (512 + tiles - 1) / tiles;       // ~2 workgroups per CU
  const int max_split = ktiles / 4 > 0 ? ktiles / 4 : 1;  // >= 4 K-tiles (128 k) per slab
  if (split > max_split) split = max_split;
  if (split > 64) split = 64;
  return split < 1 ? 1 : split;
}

template <int LAYOUT, int PREC>
void launch(const GemmArgs& g, bool norm, dim3 grid, hipStream_t st) {
  if (norm) hipLaunchKernelGGL((gemm_kernel<LAYOUT, PREC, true>), grid, dim3(256), 0, st, g);
  else      hipLaunchKernelGGL((gemm_kernel<LAYOUT, PREC, false>), grid, dim3(256), 0, st, g);
}

}  // namespace

extern "C" int64_t vae_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const int split = choose_split((int)M, (int)N, (int)K);
  return split > 1 ? (int64_t)split * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int vae_gemm(int layout, int precision, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                        const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, const float* x_mean,
                        int accumulate, float* workspace, int64_t workspace_bytes, void* stream) {
  VAE_REQUIRE(layout >= 0 && layout <= 2, "vae_gemm: layout must be 0 (NT), 1 (NN) or 2 (TN), got %d", layout);
  VAE_REQUIRE(precision == 0 || precision == 1, "vae_gemm: precision must be 0 (fp32) or 1 (bf16), got %d", precision);
  VAE_REQUIRE(M > 0 && N > 0 && K > 0 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30),
              "vae_gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  VAE_REQUIRE(A && B && C, "vae_gemm: null operand");
  const int64_t a_min = layout == TN ? M : K, b_min = layout == NT ? K : N;
  VAE_REQUIRE(lda >= a_min && ldb >= b_min && ldc >= N, "vae_gemm: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)",
              (long long)lda, (long long)ldb, (long long)ldc);
  VAE_REQUIRE(!(x_mean && layout == NN), "vae_gemm: x_mean normalisation applies to NT (A) or TN (B) only");

  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.mean = x_mean;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.accumulate = accumulate;
  g.vecA = (lda % 4 == 0) && vae_aligned16(A);
  g.vecB = (ldb % 4 == 0) && vae_aligned16(B);
  const int tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  int split = choose_split(g.M, g.N, g.K);
  const int ktiles = (g.K + BK - 1) / BK;
  const int tiles_per_split = (ktiles + split - 1) / split;
  g.k_per_split = tiles_per_split * BK;
  split = (ktiles + tiles_per_split - 1) / tiles_per_split;  // drop empty slabs
  g.split = split;
  hipStream_t st = (hipStream_t)stream;
  if (split > 1) {
    VAE_REQUIRE(workspace && workspace_bytes >= (int64_t)split * M * N * (int64_t)sizeof(float),
                "vae_gemm: workspace too small: need %lld bytes (see vae_gemm_workspace_bytes)",
                (long long)((int64_t)split * M * N * 4));
    VAE_REQUIRE(vae_aligned16(workspace), "vae_gemm: workspace must be 16-byte aligned");
    g.C = workspace;
  }
  dim3 grid(tiles_m * g.tiles_n, split);
  const bool norm = x_mean != nullptr;
  if (precision == 0) {
    if (layout == NT) launch<NT, 0>(g, norm, grid, st);
    else if (layout == NN) launch<NN, 0>(g, false, grid, st);
    else launch<TN, 0>(g, norm, grid, st);
  } else {
    if (layout == NT) launch<NT, 1>(g, norm, grid, st);
    else if (layout == NN) launch<NN, 1>(g, false, grid, st);
    else launch<TN, 1>(g, norm, grid, st);
  }
  VAE_CHECK_LAUNCH("vae_gemm");
  if (split > 1) {
    const int64_t total = M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, split, g.M, g.N, bias, C, ldc,
                       accumulate);
    VAE_CHECK_LAUNCH("vae_gemm(splitk_reduce)");
  }
  return VAE_OK;
}
