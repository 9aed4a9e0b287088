// Peak-rate probe: back-to-back MFMAs with no memory traffic.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f;
  bf16x8 a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)a; b8[j] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0);
      else    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}
template <int NACC, bool BF>
void run(const char* name, int wg_per_cu, float* d) {
  int iters = BF ? 20000 : 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, BF><<<256 * wg_per_cu, 256>>>(d, 10);
  hipEventRecord(e0);
  k<NACC, BF><<<256 * wg_per_cu, 256>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)256 * wg_per_cu * 4 * iters * NACC * (BF ? 32768.0 : 4096.0);
  printf("%s nacc=%d wg/cu=%d: %.3f ms  %.1f TFLOP/s\n", name, NACC, wg_per_cu, ms, fl / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 4);
  run<4, false>("f32 32x32x2", 1, d);
  run<4, false>("f32 32x32x2", 2, d);
  run<1, false>("f32 32x32x2", 1, d);
  run<1, false>("f32 32x32x2", 4, d);
  run<4, true>("bf16 32x32x16", 1, d);
  run<4, true>("bf16 32x32x16", 2, d);
  run<1, true>("bf16 32x32x16", 4, d);
  return 0;
}
