// Internal interface of middle.hip (fused encoder-tail / decoder-head kernels), used by step.hip.
#pragma once
#include "../../include/vae_amd.h"
#include "internal.h"

bool vae_mid_supported(const vae_step_config& c);

int vae_mid_forward(const vae_step_config& c, const vae_step_tensors* p, const float* slabs, int nslab, const float* eps_z,
                    const float* eps_l, float* pre_enc0, float* h_enc0, float* ln_mz, float* ln_rz, float* ln_ml,
                    float* ln_rl, float* heads_z, float* heads_l, float* qz_v, float* z, float* kl_z, float* ql_v,
                    float* library, float* kl_l, float* dec_pre, float* dec_h, float* ln_md, float* ln_rd,
                    float* absmax_px /* nullable: B * VAE_MID_WAVES per-wave maxima of |dec_h| */,
                    uint16_t* px16 /* nullable: dec_h again as bf16 */,
                    const float* x_mean, const float* wsum /* nullable pair: the slabs hold x W^T of the raw counts (VaeGemmCounts):
                                                              pre = slabs / x_mean[b] - wsum[e] + bias[e] */,
                    void* stream);

int vae_mid_backward(const vae_step_config& c, const vae_step_tensors* p, const vae_step_tensors* g, const float* slabs,
                     int nslab, const float* eps_z, const float* eps_l, const float* dz_extra, float g_kl,
                     const float* dec_pre, const float* ln_md, const float* ln_rd, const float* heads_z,
                     const float* heads_l, const float* qz_v, const float* ql_v, const float* dlibrary,
                     const float* pre_enc0, const float* ln_mz, const float* ln_rz, const float* ln_ml,
                     const float* ln_rl, const float* z, const float* h_enc0, float* dbet_d, float* dgam_d,
                     float* dpre_d, float* dq_z, float* dq_l, float* dbet_e, float* dgam_e, float* dpre_e,
                     float* absmax_dpre /* nullable: B * VAE_MID_WAVES per-wave maxima of |dpre_e| */,
                     uint16_t* dpre16 /* nullable: dpre_e again as bf16 */,
                     const VaeDeferredColreduce* cr /* nullable: column slabs to reduce in the last launch */,
                     const float* x_mean, float* dpre_s /* nullable pair: dpre_s = dpre_e / x_mean[b] is written as well and the
                                                           maxima are taken of it (the wgrad over the raw counts) */,
                     void* stream);
