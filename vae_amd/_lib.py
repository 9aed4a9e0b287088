"""ctypes binding of the C-ABI HIP library ``vae_amd/lib/libvae_amd.so`` (include/vae_amd.h).

The product path has NO fallback: if the library is missing or fails to load, every op
raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C vae_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAE_AMD_LIB") or os.path.join(_HERE, "lib", "libvae_amd.so")   # env: an alternative build

_i64, _i32, _f32, _ptr = C.c_int64, C.c_int, C.c_float, C.c_void_p
X_F32, X_U8 = 0, 1   # enum { VAE_X_F32, VAE_X_U8 } of include/vae_amd.h

# name -> (restype, argtypes); mirrors include/vae_amd.h one to one
SIGNATURES = {
    "vae_version": (_i32, []),
    "vae_arch": (C.c_char_p, []),
    "vae_last_error": (C.c_char_p, []),
    "vae_device_count": (_i32, []),
    "vae_gemm_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "vae_gemm": (_i32, [_i32, _i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _ptr, _i64, _ptr]),
    "vae_gemm_noalloc_max_bytes": (_i64, []),
    "vae_gemm_noalloc_init": (_i32, []),
    "vae_gemm_noalloc": (_i32, [_i32, _i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _ptr]),
    "vae_elbo_step_bf16_activations": (_i32, [_ptr]),
    "vae_elbo_step_counts_path": (_i32, [_ptr]),
    "vae_gemm_bf16_storage_ok": (_i32, [_i32, _i32, _i64, _i64, _i64]),
    "vae_gemm_bf16_storage": (_i32, [_i32, _i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr]),
    "vae_gemm_slabs_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "vae_gemm_slabs": (_i32, [_i32, _i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, C.POINTER(C.c_int), _ptr]),
    "vae_px_rate": (_i32, [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr]),
    "vae_elbo_step_px_rate": (_i32, [_ptr, _ptr, _i64, _ptr, _i64, _ptr]),
    "vae_l1_penalty": (_i32, [_ptr, _ptr, _i64, _ptr, _i32, C.c_double, _ptr, _i64, _ptr, _ptr]),
    "vae_dropout_apply": (_i32, [_ptr, _i64, _i64, _i64, _f32, C.c_uint64, C.c_uint32, C.c_uint32, _ptr]),
    "vae_gene_dropout": (_i32, [_ptr, _i32, _i64, _i64, _i64, _f32, _f32, C.c_uint64, C.c_uint32, _ptr, _ptr, _ptr, _ptr, _ptr,
                                _ptr, _i64, _ptr]),
    "vae_gather_counts": (_i32, [_ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i32, _i64, _ptr]),
    "vae_cutmix_counts": (_i32, [_ptr, _i32, _i64, _i64, _i64, _ptr, _ptr, _i64, _ptr, _i32, _i64, _ptr]),
    "vae_mixup_counts": (_i32, [_ptr, _i32, _i64, _i64, _i64, _ptr, _ptr, _ptr, _i64, _ptr]),
    "vae_poisson_fwd": (_i32, [_ptr, _i64, _ptr, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    "vae_poisson_bwd": (_i32, [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    "vae_absmax_workspace_bytes": (_i64, []),
    "vae_absmax_scale": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _ptr]),
    "vae_gemm_scaled": (_i32, [_i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i32, _ptr, _i64,
                               _ptr]),
    "vae_gemm_scaled_slabs": (_i32, [_i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _ptr, _i64,
                                     C.POINTER(C.c_int), _ptr]),
    "vae_gemm_rowscaled_ok": (_i32, [_i64, _i64, _i64]),
    "vae_gemm_rowscaled": (_i32, [_i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr]),
    "vae_row_stats": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _ptr]),
    "vae_gemm_counts": (_i32, [_i32, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr,
                               _i64, _ptr]),
    "vae_row_mean": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr]),
    "vae_normalize_rows": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _ptr]),
    "vae_normalize_rows_u8": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _ptr]),
    "vae_gelu_ln_fwd": (_i32, [_ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr]),
    "vae_gelu_ln_bwd": (_i32, [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    "vae_reparam_kl_fwd": (_i32, [_ptr, _i64, _ptr, _i64, _i64, _i64, _f32, _f32, _ptr, _ptr, _i64, _ptr, _ptr]),
    "vae_reparam_kl_bwd": (_i32, [_ptr, _i64, _ptr, _f32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _f32, _f32, _ptr, _i64, _ptr, _ptr]),
    "vae_colsum": (_i32, [_ptr, _i64, _i64, _i64, _ptr, _ptr]),
    "vae_elbo_reduce": (_i32, [_ptr, _ptr, _ptr, _i64, _f32, _f32, _ptr, _ptr, _ptr]),
    "vae_gene_table_rows": (_i64, []),
    "vae_gene_table": (_i32, [_ptr, _i64, _ptr, _ptr]),
    "vae_nbzinb_max_genes": (_i64, []),
    "vae_nbzinb_fwd": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "vae_nbzinb_fwd_workspace_bytes": (_i64, [_i64, _i64]),
    "vae_nbzinb_fwd_ws": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i32, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i32,
                                 _ptr, _i64, _ptr]),
    "vae_nbzinb_fwd_u8": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "vae_nbzinb_fwd_ws_bf16": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i32, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i32,
                               _ptr, _i64, _ptr]),
    "vae_nbzinb_bwd_bf16": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i32, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _i64, _i64,
                            _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    "vae_nbzinb_bwd_u8": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _i64, _i64,
                                 _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i32, _ptr]),
    "vae_nbzinb_bwd_workspace_bytes": (_i64, [_i64, _i64]),
    "vae_nbzinb_bwd": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _i64, _i64,
                              _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    "vae_nbzinb_bwd_overwrite": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _i64, _i64,
                              _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    "vae_softmax_rate_fwd": (_i32, [_ptr, _i64, _ptr, _i64, _i64, _ptr, _ptr, _ptr]),
    "vae_softmax_rate_bwd": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr, _ptr]),
    "vae_add_inplace": (_i32, [_ptr, _ptr, _i64, _ptr]),
    "vae_elbo_step_workspace_bytes": (_i64, [_ptr]),
    "vae_elbo_step_ws_offset": (_i64, [_ptr, _i32]),
    "vae_elbo_step_forward": (_i32, [_ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr]),
    "vae_elbo_step_backward": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _i32, _ptr, _i64, _ptr, _ptr]),
    "vae_elbo_step_backward_events": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _i32, _ptr, _i64, _ptr, _ptr, _ptr]),
    "vae_elbo_step_backward_slabs": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _i32, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _ptr]),
    "vae_adamw_workspace_bytes": (_i64, []),
    "vae_adamw_step": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i32, C.c_double, C.c_double, C.c_double, C.c_double,
                              _i64, C.c_double, C.c_double, _ptr, _i64, _ptr, _ptr]),
    "vae_adamw_wmax_partials": (_i64, []),
    "vae_adamw_step_wmax": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i32, C.c_double, C.c_double, C.c_double, C.c_double,
                                   _i64, C.c_double, C.c_double, _ptr, _i64, _ptr, _ptr, _ptr, _ptr]),
    "vae_adamw_step_bf16": (_i32, [_ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i32, C.c_double, C.c_double, C.c_double, C.c_double,
                                   _i64, C.c_double, C.c_double, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "vae_cell_heads_workspace_bytes": (_i64, [_i64, _i64, _ptr]),
    "vae_cell_heads_step": (_i32, [_ptr, _ptr, _i64, _i64, _ptr, _i64, _ptr, _i64, _f32, _ptr, _ptr, _ptr, _i64, _ptr]),
    "vae_loglik_elem_fwd": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr, _i64, _ptr]),
    "vae_loglik_elem_bwd": (_i32, [_i32, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr]),
}

VAE_MAX_LAYERS = 4
_fp = C.POINTER(C.c_float)


class AdamWSegment(C.Structure):
    """vae_adamw_segment of include/vae_amd.h"""
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("weight_decay", C.c_float)]


class StepConfig(C.Structure):
    """vae_step_config of include/vae_amd.h"""
    _fields_ = [("B", _i64), ("G", _i64), ("H", _i64), ("Hl", _i64), ("L", _i64), ("Hd", _i64),
                ("n_layers", C.c_int32), ("zinb", C.c_int32), ("precision", C.c_int32),
                ("l_mean", _f32), ("l_var", _f32), ("kl_weight", _f32), ("gene_loss_coeff", _f32), ("loss_scale", _f32),
                ("x_dtype", C.c_int32), ("dropout_rate", _f32), ("rng_step", C.c_uint32), ("rng_seed", C.c_uint64),
                ("input_dropout_rate", _f32), ("input_dropout_alpha", _f32), ("gene_running", _ptr),
                ("weight_max_enc", _ptr), ("weight_max_heads", _ptr), ("weight_max_n", C.c_int32),
                ("n_cov", C.c_int32), ("batch_cov", _ptr), ("ld_cov", _i64), ("ld_enc_w", _i64),
                ("weight_bf16_enc", _ptr), ("weight_bf16_heads", _ptr), ("defer_elbo_reduce", C.c_int32), ("fractional_counts", C.c_int32)]


VAE_MAX_HEADS = 7


class CellHeads(C.Structure):
    """vae_cell_heads of include/vae_amd.h"""
    _I, _F, _P = C.c_int32 * VAE_MAX_HEADS, C.c_float * VAE_MAX_HEADS, _ptr * VAE_MAX_HEADS
    _fields_ = [("n_heads", C.c_int32), ("width", _I), ("discrete", _I), ("dz_coeff", _F), ("w", _P), ("b", _P), ("dw", _P),
                ("db", _P), ("class_weight", _P)]


class StepTensors(C.Structure):
    """vae_step_tensors of include/vae_amd.h (device pointers as void*)"""
    _A = _ptr * VAE_MAX_LAYERS
    _fields_ = [("px_r", _ptr), ("enc_w0", _ptr), ("enc_b0", _ptr),
                ("encz_ln_g", _A), ("encz_ln_b", _A), ("encz_w", _A), ("encz_b", _A),
                ("encl_ln_g", _ptr), ("encl_ln_b", _ptr),
                ("hz_w", _ptr), ("hz_b", _ptr), ("hl_w", _ptr), ("hl_b", _ptr),
                ("dec_w", _A), ("dec_b", _A), ("dec_ln_g", _A), ("dec_ln_b", _A),
                ("heads_w", _ptr), ("heads_b", _ptr)]


WS_SLICES = ("QZ_HEADS", "QZ_V", "Z", "KL_Z", "QL_HEADS", "QL_V", "LIBRARY", "KL_L", "LL", "SCALARS", "DZ_EXTRA", "RECON",
             "GENE_THRESHOLD", "OPERAND_MAXIMA")

_lib = None


def load() -> C.CDLL:
    """Load the library once; raise loudly (no CPU fallback) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"vae_amd: HIP library not found at {LIB_PATH}. Build it (make -C vae_amd/csrc); "
            "there is no CPU fallback for the product path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().vae_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"vae_amd {what} failed (status {rc}): {msg}")
