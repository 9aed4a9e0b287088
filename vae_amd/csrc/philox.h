// Counter-based random numbers for the dropout masks of the training step (gfx950 device code).
//
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), the generator
// behind torch's CUDA/HIP dropout; the STREAM is this library's own and is part of its interface
// (include/vae_amd.h, "dropout"), so that a mask can be regenerated in the backward pass, on another
// rank, or on the host by a test:
//   key     = (seed low 32 bits, seed high 32 bits)
//   counter = (elem low, elem high, site, step)      elem = cell * width + column
//   keep    = word 0 of the output  >=  floor(p * 2^32)         -> P(keep) = 1 - p
// One call per element (only word 0 is used): the masks cover a few hundred values per cell.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint4 vae_philox4(uint64_t seed, uint32_t step, uint32_t site, uint64_t elem) {
  uint32_t c0 = (uint32_t)elem, c1 = (uint32_t)(elem >> 32), c2 = site, c3 = step;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ uint32_t vae_philox_word0(uint64_t seed, uint32_t step, uint32_t site, uint64_t elem) {
  return vae_philox4(seed, step, site, elem).x;
}

// Dropout as nn.Dropout applies it (modules.py:115): kept values are scaled by 1 / (1 - p).
struct VaeDropout {
  uint64_t seed;
  uint32_t step, threshold;  // keep iff word >= threshold
  float scale;               // 1 / (1 - p); 0 marks "off"
  __device__ __forceinline__ bool on() const { return scale != 0.f; }
  // multiplier of element (cell, col) of a (B, width) activation at `site`
  __device__ __forceinline__ float mult(uint32_t site, int64_t cell, int width, int col) const {
    return vae_philox_word0(seed, step, site, (uint64_t)cell * (uint64_t)width + (uint64_t)col) >= threshold ? scale : 0.f;
  }
};

// sites: which activation of the network a mask belongs to (layer = hidden layer index); the library
// encoder is built with dropout_rate 0 (networks.py:215-222) and has no site
#define VAE_SITE_ENC_Z(layer) (4u * (uint32_t)(layer) + 0u)
#define VAE_SITE_DEC(layer) (4u * (uint32_t)(layer) + 2u)
// GeneDropout on the z-encoder's input (modules.py:190-209): counter element = cell * ceil(G / 4) + g / 4,
// output word g % 4; gene g of the cell is KEPT iff that word < threshold[g] = floor(p_g * 2^32)
#define VAE_SITE_GENES 1u

static inline VaeDropout vae_make_dropout(float p, uint64_t seed, uint32_t step) {
  VaeDropout d;
  d.seed = seed;
  d.step = step;
  d.scale = 0.f;
  d.threshold = 0;
  if (p > 0.f) {
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    d.threshold = (uint32_t)t;
    d.scale = (float)(1.0 / (1.0 - (double)p));
  }
  return d;
}
