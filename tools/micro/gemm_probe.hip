// Cycle stamps of one workgroup of gemm_x3_kernel (decoder-head forward shape, cfg2).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVAE_PROBE=600 tools/micro/gemm_probe.hip vae_amd/csrc/runtime.hip -o tools/micro/gemm_probe
//   add -DVAE_ABL=<bits> to ablate: 1 = stage only once, 2 = fetch only once, 4 = no multiply
//   ./gemm_probe [layout 0 (NT, decoder heads forward) | 1 (NN, decoder dgrad)]
//   -DPROBE_CFG="128, 160, 4, 1" -DPROBE_BN=160 picks the tile / wave layout / planes; ./gemm_probe 0 4096 probes B = 4096
#include "../../vae_amd/csrc/gemm.hip"
#ifndef PROBE_CFG
#define PROBE_CFG 128, 128, 2, 3
#define PROBE_BN 128
#endif
#include <stdio.h>
#include <vector>

template <int LAY>
void run(GemmArgs g, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_x3_kernel<LAY, PROBE_CFG>), dim3(grid), dim3(256), 0, 0, g);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((gemm_x3_kernel<LAY, PROBE_CFG>), dim3(grid), dim3(256), 0, 0, g);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("kernel %.1f us (grid %d)\n", ms * 100, grid);
}

int main(int argc, char** argv) {
  const int lay = argc > 1 ? atoi(argv[1]) : 0;
  const int M = argc > 2 ? atoi(argv[2]) : 512, N = 40000, K = 256;   // NT: A (M,K), B (N,K)
  const size_t na = (size_t)M * K, nb = (size_t)N * K;
  float *A, *B, *C;
  hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemset(A, 0, na * 4); hipMemset(B, 0, nb * 4);
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.lda = K; g.ldb = K; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.k_per_split = K; g.split = 1; g.vecA = g.vecB = 1; g.vecC = 1; g.group_m = getenv("GROUP") ? atoi(getenv("GROUP")) : 0; g.tiles_m = (M + 127) / 128; g.tiles_n = (N + PROBE_BN - 1) / PROBE_BN;
  if (lay == 1) { /* NN dgrad: M=512, N=256, K=40000 (A (M,K), B (K,N)) split 8 */
    g.lda = 40000; g.ldb = 256; g.ldc = 256; g.M = 512; g.N = 256; g.K = 40000;
    g.tiles_m = 4; g.tiles_n = 2; g.split = getenv("SPLIT") ? atoi(getenv("SPLIT")) : 63; g.k_per_split = ((1250 + g.split - 1) / g.split) * 32;
    g.split = (40000 + g.k_per_split - 1) / g.k_per_split;
    hipFree(A); hipFree(C); hipMalloc(&A, (size_t)512 * 40000 * 4); hipMemset(A, 0, (size_t)512 * 40000 * 4);
    hipMalloc(&C, (size_t)g.split * 512 * 256 * 4); g.A = A; g.C = C;
  }
  if (lay == 2) { /* TN row-scaled decoder wgrad: M=40000, N=256, K=512 (A (K,M) = dlogits, B (K,N) = px); build with -DPROBE_CFG="160, 128, 1, 2, 0, true" */
    const int G = 20000, chunks = 52;
    g.lda = 40000; g.ldb = 256; g.ldc = 256; g.M = 40000; g.N = 256; g.K = 512;
    g.tiles_m = 250; g.tiles_n = 2; g.split = 1; g.k_per_split = 512;
    hipFree(A); hipFree(B); hipFree(C);
    hipMalloc(&A, (size_t)512 * 40000 * 4); hipMemset(A, 0, (size_t)512 * 40000 * 4);
    hipMalloc(&B, (size_t)512 * 256 * 4); hipMemset(B, 0, (size_t)512 * 256 * 4);
    hipMalloc(&C, (size_t)40000 * 256 * 4);
    float *rm, *sb;
    hipMalloc(&rm, (size_t)chunks * 5 * G * 4); hipMemset(rm, 0, (size_t)chunks * 5 * G * 4);
    hipMalloc(&sb, 4096 * 4); hipMemset(sb, 0, 4096 * 4);
    g.A = A; g.B = B; g.C = C; g.rmax0 = rm + 3 * G; g.rmax1 = rm + 4 * G; g.rmax_split = G; g.rmax_chunks = chunks; g.rmax_stride = 5 * G;
    g.sb = sb; g.nb = 4096;
  }
  const int grid = g.tiles_m * g.tiles_n * g.split;
#ifdef PROBE_TN   // (the row-scaled instantiation exists for TN only)
  run<TN>(g, grid);
#else
  if (lay == 0) run<NT>(g, grid);
  else if (lay == 1) run<NN>(g, grid);
  else run<TN>(g, grid);
#endif
  static long long h[4][256];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe), sizeof(h));
  // stamps: [0] start, then per K-step 4 (loop top, after sync1, after LDS writes, after sync2), then end of loop, end of kernel
  const int ksteps = g.k_per_split / 32;
  if (getenv("PIPE_STAMPS")) {   // pipelined loop: [0] start, [1] before the loop, then per K step 3 (top, MFMAs + conversion issued, after the barrier)
    for (int w = 0; w < 4; ++w) {
      printf("wave %d: prologue %lld |", w, h[w][2] - h[w][0]);
      for (int k = 0; k < (ksteps < 12 ? ksteps : 12); ++k) {
        const long long* s = &h[w][2 + 3 * k];
        printf(" [work %lld barrier %lld]", s[1] - s[0], s[2] - s[1]);
      }
      printf(" | loop %lld\n", h[w][2 + 3 * ksteps] - h[w][2]);
    }
    return 0;
  }
  for (int w = 0; w < 4; ++w) {
    printf("wave %d: prologue %lld |", w, h[w][1] - h[w][0]);
    const int show = ksteps < 10 ? ksteps : 10;
    for (int k = 0; k < show; ++k) {
      const long long* s = &h[w][1 + 4 * k];
      const long long next = (k + 1 < ksteps) ? s[4] : h[w][1 + 4 * ksteps];
      printf(" [sync1 %lld wr %lld sync2 %lld mm %lld]", s[1] - s[0], s[2] - s[1], s[3] - s[2], next - s[3]);
    }
    printf(" | epilogue %lld total %lld\n", h[w][2 + 4 * ksteps] - h[w][1 + 4 * ksteps], h[w][2 + 4 * ksteps] - h[w][0]);
  }
  return 0;
}
