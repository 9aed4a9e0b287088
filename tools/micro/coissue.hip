// Do VALU instructions of one wave issue in the shadow of another wave's MFMAs on the same SIMD?
// hipcc --offload-arch=gfx950 -O3 tools/micro/coissue.hip -o tools/micro/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// mode bit0: waves 0-3 run MFMAs; bit1: waves 4-7 run VALU work of kind `kind`
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float sink = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      bf16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)threadIdx.x; b[j] = (__bf16)1.f; }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 15; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sink += acc[i][r];
    }
  } else if (mode & 2) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 60; ++u) {
        float& v = x[u & 7];
        if (KIND == 0) v = fmaf(v, 1.0001f, 0.5f);
        else if (KIND == 1) { bf16x2 p; p[0] = (__bf16)v; p[1] = (__bf16)(v + 1.f); v = (float)p[0] + (float)p[1]; }
        else { unsigned b = __float_as_uint(v); b = (b + 0x8000u) & 0xffff0000u; v = v - __uint_as_float(b) + 1.f; }
      }
    }
    for (int j = 0; j < 8; ++j) sink += x[j];
  }
  if (sink == 12345.678f) out[0] = sink;
}
template <int KIND>
void run(const char* name, float* d) {
  const int iters = 4000;
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(d, 10, mode);
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(d, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s mode=%d (%s): %.3f ms  -> %.0f ns per iteration (15 MFMA | 60 VALU-steps per wave)\n", name, mode,
           mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms, ms * 1e6 / iters);
  }
}
int main() {
  float* d; hipMalloc(&d, 4);
  run<0>("fma      ", d);
  run<1>("cvt_bf16 ", d);
  run<2>("int-round", d);
  return 0;
}
