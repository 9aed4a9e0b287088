// How fast can workgroups write a (M x N) fp32 matrix tile by tile (128 x 160 tiles, 4 waves of 32 rows)?
//   V0: the GEMM epilogue's pattern -- dword stores, a wave instruction = 2 rows x 128 B
//   V1: 16-byte stores, a wave instruction = 1.6 rows x 640 B (whole tile rows back to back)
//   V2: linear fill of the same bytes (calibration)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_probe.hip -o tools/micro/store_probe ; ./store_probe [M]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int V>
__global__ __launch_bounds__(256, 2) void store_kernel(float* __restrict__ out, int M, int N, int tiles_m, float val) {
  const int t = blockIdx.x, tm = t % tiles_m, tn = t / tiles_m;
  const int m0 = tm * 128, n0 = tn * 160;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  if (V == 0) {
#pragma unroll
    for (int ni = 0; ni < 5; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(long)m * N + n0 + ni * 32 + l31] = val + r;
      }
  } else if (V == 1) {
    // the wave's 32 x 160 block = 1280 float4, 20 instructions of 64
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int c = j * 64 + lane, row = c / 40, c4 = c % 40;
      *reinterpret_cast<float4*>(out + (long)(m0 + wave * 32 + row) * N + n0 + 4 * c4) = make_float4(val, val + 1, val + 2, val + j);
    }
  } else {
    float4* o = reinterpret_cast<float4*>(out) + (long)t * (128 * 160 / 4);
#pragma unroll
    for (int j = 0; j < 20; ++j) o[j * 256 + tid] = make_float4(val, val + 1, val + 2, val + j);
  }
}

template <int V>
void run(float* out, int M, int N) {
  const int tiles_m = M / 128, grid = tiles_m * (N / 160);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<V>, dim3(grid), dim3(256), 0, 0, out, M, N, tiles_m, 1.f);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<V>, dim3(grid), dim3(256), 0, 0, out, M, N, tiles_m, 1.f);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("V%d M=%d: %.1f us, %.2f TB/s\n", V, M, ms * 100, (double)M * N * 4 / (ms * 1e-4) / 1e12 * 1e-0 / 1e0 / 1.0 / 1e0);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = 40000;
  float* out;
  hipMalloc(&out, (size_t)M * N * 4);
  run<0>(out, M, N); run<1>(out, M, N); run<2>(out, M, N);
  return 0;
}
