// Do MFMA and VALU instructions overlap on a gfx950 SIMD?  One workgroup per CU, W waves per SIMD, each wave runs
//   mode 0: MFMA only (4 independent 32x32x16 f16 accumulators)     mode 1: VALU only (independent v_fma_f32 chains)
//   mode 2: both, interleaved 1 MFMA : R VALU in program order      mode 3: both, in two separate blocks (MFMAs, then VALU)
// and reports cycles per iteration (s_memtime of wave 0).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NM = 16, R = 6;   // MFMAs per iteration, VALU per MFMA
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.1f + i;
  const float c0 = 1.0001f, c1 = 0.5f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NM * R; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], c0, c1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (MODE != 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
        if (MODE != 0) {
#pragma unroll
          for (int q = 0; q < R; ++q) v[(m * R + q) & 7] = __builtin_fmaf(v[(m * R + q) & 7], c0, c1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(int waves_per_simd, float* out, long long* cyc) {
  const int iters = 200;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("mode %d, %d wave(s)/SIMD: %.1f cycles (s_memtime ticks x?) per iteration of %d MFMA + %d VALU\n", MODE, waves_per_simd, (double)h / iters, MODE == 1 ? 0 : NM,
         MODE == 0 ? 0 : NM * R);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  for (int w = 1; w <= 2; ++w) { run<0>(w, out, cyc); run<1>(w, out, cyc); run<2>(w, out, cyc); run<3>(w, out, cyc); }
  return 0;
}
