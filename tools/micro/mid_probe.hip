// Stage timing of mid_fwd_kernel with in-kernel wall_clock64 stamps.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVAE_PROBE -I vae_amd/csrc tools/micro/mid_probe.hip vae_amd/csrc/runtime.hip -o tools/micro/mid_probe
#include "../../vae_amd/csrc/middle.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 512, H = 256, Hl = 128, HE = 384, L = 30, Hd = 256, NS = argc > 2 ? atoi(argv[2]) : 42;
  vae_step_config c{};
  c.B = B; c.G = 20000; c.H = H; c.Hl = Hl; c.L = L; c.Hd = Hd; c.n_layers = 1; c.zinb = 1; c.l_mean = 9.f; c.l_var = 1.f;
  auto dmalloc = [](size_t n) { float* p; hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); return p; };
  float* slabs = dmalloc((size_t)NS * B * HE);
  vae_step_tensors p{};
  p.enc_b0 = dmalloc(HE); p.encz_ln_g[0] = dmalloc(H); p.encz_ln_b[0] = dmalloc(H); p.encl_ln_g = dmalloc(Hl); p.encl_ln_b = dmalloc(Hl);
  p.hz_w = dmalloc((L + 1) * H); p.hz_b = dmalloc(L + 1); p.hl_w = dmalloc(2 * Hl); p.hl_b = dmalloc(2);
  p.dec_w[0] = dmalloc(Hd * L); p.dec_b[0] = dmalloc(Hd); p.dec_ln_g[0] = dmalloc(Hd); p.dec_ln_b[0] = dmalloc(Hd);
  float* buf[20];
  for (int i = 0; i < 20; ++i) buf[i] = dmalloc((size_t)B * HE);
  float* eps_z = dmalloc(B * L); float* eps_l = dmalloc(B);
  for (int it = 0; it < 3; ++it)
    vae_mid_forward(c, &p, slabs, NS, eps_z, eps_l, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7], buf[8], buf[9],
                    buf[10], buf[11], buf[12], buf[13], buf[14], buf[15], buf[16], buf[17], nullptr, nullptr, nullptr, nullptr, nullptr);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4096 * 16);
  hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(vae_probe), h.size() * 8);
  unsigned long long t0 = ~0ull, t1 = 0;
  double d[8] = {0};
  const int NB = getenv("NBLK") ? atoi(getenv("NBLK")) : B;
  for (int b = 0; b < NB; ++b) {
    if (h[b * 16] < t0) t0 = h[b * 16];
    if (h[b * 16 + 7] > t1) t1 = h[b * 16 + 7];
    for (int i = 1; i < 8; ++i) d[i] += (double)(h[b * 16 + i] - h[b * 16 + i - 1]);
  }
  printf("kernel span (first start -> last end): %.2f us\n", (t1 - t0) / 100.0);
  // wave-per-cell kernel: stamps by thread 0 (wave 0 of each workgroup, its LAST cell wins): 1 = weights staged, 2 = slab sum + gelu,
  // 3 = LN, 4 = heads, 5 = reparam, 6 = decoder layer, 7 = decoder LN + stores.  Only workgroups that ran count.
  const char* nm[8] = {"", "stage weights", "slab sum+gelu", "LN + h store", "heads", "reparam", "decoder layer", "LN stats + store"};
  for (int i = 1; i < 8; ++i) printf("  stage %d %-18s avg %.2f us\n", i, nm[i], d[i] / NB / 100.0);
  double first = 0, last = 0;
  for (int b = 0; b < NB; ++b) { first += (h[b * 16] - t0); last += (h[b * 16 + 7] - t0); }
  printf("avg block start offset %.2f us, avg block end offset %.2f us\n", first / NB / 100.0, last / NB / 100.0);
  return 0;
}
