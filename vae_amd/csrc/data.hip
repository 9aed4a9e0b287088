// Input side of the training step: batches are cut out of a uint8 cells x genes store
// that lives in HBM (a 5 M-cell x 30 k-gene store is 150 GB: it fits one MI355X), so a step moves no count over PCIe.
//
//   vae_gather_counts  <- SingleCellDataset._get_gene_vals_batch (data.py:406-415): rows `cell_rows` of the store,
//                         optionally the gene subset `gene_idx`, as uint8 (the fused step reads uint8 directly) or fp32
//   vae_mixup_counts   <- SingleCellDataset._mixup (data.py:457-477): out[i] = a_i x[i] + (1 - a_i) x[partner_i],
//                         evaluated as numpy does (float64 products and sum, no fused multiply-add, one rounding to fp32)
//
// Both are pure HBM streams: 16 B (fp32) / 4 B (uint8) per lane, one workgroup per 1024 genes of one cell.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vae_amd.h"
#include "common.h"

namespace {

constexpr int STRIP = 1024;  // genes per workgroup: 256 lanes x 4

template <typename OT>
__device__ __forceinline__ OT as_out(uint8_t v);
template <>
__device__ __forceinline__ uint8_t as_out<uint8_t>(uint8_t v) { return v; }
template <>
__device__ __forceinline__ float as_out<float>(uint8_t v) { return (float)v; }

// VIN: rows of the store start 4-byte aligned and no gene subset -> one 4-byte load per lane
// VOUT: rows of the output are aligned for one 4-element store per lane
template <typename OT, bool VIN, bool VOUT>
__global__ __launch_bounds__(256) void gather_counts_kernel(const uint8_t* __restrict__ store, int64_t n_cells, int64_t ld_store,
                                                            const int64_t* __restrict__ cell_rows,
                                                            const int32_t* __restrict__ gene_idx, int G,
                                                            OT* __restrict__ out, int64_t ld_out) {
  const int b = blockIdx.y;
  const int g0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (g0 >= G) return;
  const int64_t r = cell_rows[b];
  const bool row_ok = r >= 0 && r < n_cells;   // rows outside the store give zeros (the host wrapper rejects them)
  const uint8_t* src = store + (row_ok ? r : 0) * ld_store;
  uint8_t v[4] = {0, 0, 0, 0};
  const int nv = min(4, G - g0);
  if (row_ok) {
    if (VIN && nv == 4) {
      const uchar4 u = *reinterpret_cast<const uchar4*>(src + g0);
      v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nv) {
          const int64_t c = gene_idx ? (int64_t)gene_idx[g0 + j] : (int64_t)(g0 + j);
          v[j] = (c >= 0 && c < ld_store) ? src[c] : (uint8_t)0;
        }
    }
  }
  OT* dst = out + (int64_t)b * ld_out + g0;
  if (VOUT && nv == 4) {
    if (sizeof(OT) == 1) {
      *reinterpret_cast<uchar4*>(dst) = make_uchar4(v[0], v[1], v[2], v[3]);
    } else {
      *reinterpret_cast<float4*>(dst) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nv) dst[j] = as_out<OT>(v[j]);
  }
}

template <typename OT>
int gather_impl(const uint8_t* store, int64_t n_cells, int64_t ld_store, const int64_t* cell_rows, int64_t B,
                const int32_t* gene_idx, int64_t G, OT* out, int64_t ld_out, hipStream_t st) {
  const bool vin = !gene_idx && (ld_store % 4 == 0) && ((reinterpret_cast<uintptr_t>(store) & 3u) == 0);
  const bool vout = (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & (4 * sizeof(OT) - 1)) == 0);
  const dim3 grid((unsigned)((G + STRIP - 1) / STRIP), (unsigned)B);
#define LAUNCH(VI, VO)                                                                                                     \
  hipLaunchKernelGGL((gather_counts_kernel<OT, VI, VO>), grid, dim3(256), 0, st, store, n_cells, ld_store, cell_rows, gene_idx, \
                     (int)G, out, ld_out)
  if (vin) { if (vout) LAUNCH(true, true); else LAUNCH(true, false); }
  else     { if (vout) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  VAE_CHECK_LAUNCH("vae_gather_counts");
  return VAE_OK;
}

__device__ __forceinline__ float mix(double a, double b1, float xi, float xj) {
  // numpy: alpha[i] * x[i, :] and (1 - alpha[i]) * x[j, :] are float64 arrays, their sum is rounded once into the
  // float32 result (data.py:473) -- explicit _rn operations so that no fused multiply-add is formed
  return (float)__dadd_rn(__dmul_rn(a, (double)xi), __dmul_rn(b1, (double)xj));
}

template <typename XT, bool VEC>
__global__ __launch_bounds__(256) void mixup_kernel(const XT* __restrict__ x, int64_t ldx, int B, int G,
                                                    const int32_t* __restrict__ partner, const double* __restrict__ alpha,
                                                    float* __restrict__ out, int64_t ld_out) {
  const int b = blockIdx.y;
  const int g0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (g0 >= G) return;
  int j = partner[b];
  j = (j >= 0 && j < B) ? j : b;   // a partner outside the batch mixes the cell with itself (the host wrapper rejects it)
  const double a = alpha[b], b1 = 1.0 - a;
  const XT* ri = x + (int64_t)b * ldx + g0;
  const XT* rj = x + (int64_t)j * ldx + g0;
  float* o = out + (int64_t)b * ld_out + g0;
  const int nv = min(4, G - g0);
  if (VEC && nv == 4) {
    float4 vi, vj;
    if (sizeof(XT) == 1) {
      const uchar4 ui = *reinterpret_cast<const uchar4*>(ri), uj = *reinterpret_cast<const uchar4*>(rj);
      vi = make_float4((float)ui.x, (float)ui.y, (float)ui.z, (float)ui.w);
      vj = make_float4((float)uj.x, (float)uj.y, (float)uj.z, (float)uj.w);
    } else {
      vi = *reinterpret_cast<const float4*>(ri);
      vj = *reinterpret_cast<const float4*>(rj);
    }
    *reinterpret_cast<float4*>(o) = make_float4(mix(a, b1, vi.x, vj.x), mix(a, b1, vi.y, vj.y), mix(a, b1, vi.z, vj.z),
                                                mix(a, b1, vi.w, vj.w));
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < nv) o[c] = mix(a, b1, (float)ri[c], (float)rj[c]);
  }
}

// cutmix (data.py:431-454): out[i, g] = take[i, g] ? x[partner[i], g] : x[i, g]; the reference draws the gene subset of
// every cell with np.random.choice(n_genes, int(alpha n_genes), replace=False) -- the host replays those draws and hands
// the subsets over as a (B, G) byte mask.  Counts stay what they are (no arithmetic): out has the dtype asked for.
template <typename XT, typename OT>
__global__ __launch_bounds__(256) void cutmix_kernel(const XT* __restrict__ x, int64_t ldx, int B, int G,
                                                     const int32_t* __restrict__ partner, const uint8_t* __restrict__ take,
                                                     int64_t ld_take, OT* __restrict__ out, int64_t ld_out) {
  const int b = blockIdx.y;
  const int g0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (g0 >= G) return;
  int j = partner[b];
  j = (j >= 0 && j < B) ? j : b;
  const XT* ri = x + (int64_t)b * ldx + g0;
  const XT* rj = x + (int64_t)j * ldx + g0;
  const uint8_t* tk = take + (int64_t)b * ld_take + g0;
  OT* o = out + (int64_t)b * ld_out + g0;
  const int nv = min(4, G - g0);
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < nv) o[c] = (OT)(tk[c] ? rj[c] : ri[c]);
}

template <typename XT>
int mixup_impl(const XT* x, int64_t ldx, int64_t B, int64_t G, const int32_t* partner, const double* alpha, float* out,
               int64_t ld_out, hipStream_t st) {
  const bool vec = (ldx % 4 == 0) && (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & (4 * sizeof(XT) - 1)) == 0) &&
                   vae_aligned16(out);
  const dim3 grid((unsigned)((G + STRIP - 1) / STRIP), (unsigned)B);
  if (vec)
    hipLaunchKernelGGL((mixup_kernel<XT, true>), grid, dim3(256), 0, st, x, ldx, (int)B, (int)G, partner, alpha, out, ld_out);
  else
    hipLaunchKernelGGL((mixup_kernel<XT, false>), grid, dim3(256), 0, st, x, ldx, (int)B, (int)G, partner, alpha, out, ld_out);
  VAE_CHECK_LAUNCH("vae_mixup_counts");
  return VAE_OK;
}

}  // namespace

extern "C" int vae_gather_counts(const uint8_t* store, int64_t n_cells, int64_t ld_store, const int64_t* cell_rows, int64_t B,
                                 const int32_t* gene_idx, int64_t G, void* out, int out_dtype, int64_t ld_out,
                                 void* stream) {
  VAE_REQUIRE(store && cell_rows && out, "vae_gather_counts: null pointer");
  VAE_REQUIRE(n_cells > 0 && ld_store > 0 && B > 0 && B <= 65535 && G > 0 && G < (1 << 30) && ld_out >= G,
              "vae_gather_counts: bad shape (B <= 65535)");
  VAE_REQUIRE(gene_idx || G <= ld_store, "vae_gather_counts: G exceeds the store's row without a gene index");
  VAE_REQUIRE(out_dtype == VAE_X_U8 || out_dtype == VAE_X_F32, "vae_gather_counts: out_dtype must be VAE_X_U8 or VAE_X_F32");
  if (out_dtype == VAE_X_U8)
    return gather_impl<uint8_t>(store, n_cells, ld_store, cell_rows, B, gene_idx, G, (uint8_t*)out, ld_out, (hipStream_t)stream);
  return gather_impl<float>(store, n_cells, ld_store, cell_rows, B, gene_idx, G, (float*)out, ld_out, (hipStream_t)stream);
}

extern "C" int vae_cutmix_counts(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, const int32_t* partner,
                                 const uint8_t* take, int64_t ld_take, void* out, int out_dtype, int64_t ld_out, void* stream) {
  VAE_REQUIRE(x && partner && take && out, "vae_cutmix_counts: null pointer");
  VAE_REQUIRE(B > 0 && B <= 65535 && G > 0 && G < (1 << 30) && ldx >= G && ld_out >= G && ld_take >= G,
              "vae_cutmix_counts: bad shape (B <= 65535)");
  VAE_REQUIRE((x_dtype == VAE_X_U8 || x_dtype == VAE_X_F32) && (out_dtype == VAE_X_U8 || out_dtype == VAE_X_F32) &&
                  !(x_dtype == VAE_X_F32 && out_dtype == VAE_X_U8),
              "vae_cutmix_counts: dtypes must be VAE_X_U8 / VAE_X_F32 (fp32 counts cannot go out as uint8)");
  VAE_REQUIRE(out != x, "vae_cutmix_counts: in-place mixing is not possible (rows read their partners)");
  const dim3 grid((unsigned)((G + STRIP - 1) / STRIP), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == VAE_X_U8 && out_dtype == VAE_X_U8)
    hipLaunchKernelGGL((cutmix_kernel<uint8_t, uint8_t>), grid, dim3(256), 0, st, (const uint8_t*)x, ldx, (int)B, (int)G, partner,
                       take, ld_take, (uint8_t*)out, ld_out);
  else if (x_dtype == VAE_X_U8)
    hipLaunchKernelGGL((cutmix_kernel<uint8_t, float>), grid, dim3(256), 0, st, (const uint8_t*)x, ldx, (int)B, (int)G, partner,
                       take, ld_take, (float*)out, ld_out);
  else
    hipLaunchKernelGGL((cutmix_kernel<float, float>), grid, dim3(256), 0, st, (const float*)x, ldx, (int)B, (int)G, partner, take,
                       ld_take, (float*)out, ld_out);
  VAE_CHECK_LAUNCH("vae_cutmix_counts");
  return VAE_OK;
}

extern "C" int vae_mixup_counts(const void* x, int x_dtype, int64_t ldx, int64_t B, int64_t G, const int32_t* partner,
                                const double* alpha, float* out, int64_t ld_out, void* stream) {
  VAE_REQUIRE(x && partner && alpha && out, "vae_mixup_counts: null pointer");
  VAE_REQUIRE(B > 0 && B <= 65535 && G > 0 && G < (1 << 30) && ldx >= G && ld_out >= G, "vae_mixup_counts: bad shape (B <= 65535)");
  VAE_REQUIRE(x_dtype == VAE_X_U8 || x_dtype == VAE_X_F32, "vae_mixup_counts: x_dtype must be VAE_X_U8 or VAE_X_F32");
  VAE_REQUIRE((const void*)out != x, "vae_mixup_counts: in-place mixing is not possible (rows read their partners)");
  if (x_dtype == VAE_X_U8)
    return mixup_impl<uint8_t>((const uint8_t*)x, ldx, B, G, partner, alpha, out, ld_out, (hipStream_t)stream);
  return mixup_impl<float>((const float*)x, ldx, B, G, partner, alpha, out, ld_out, (hipStream_t)stream);
}
