"""Accuracy of the GEMM precisions against fp64 (operands = the fp32 values the kernels see):
   python tools/acc_gemm.py
P: 0 = fp32 mode (three bf16 terms, six products), F = fp32 MFMA pipeline (run with VAE_GEMM_X3=0 to get it under P=0),
   S = two scaled f16 terms, three products (gemm_scaled), 1 = bf16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import ops
g = torch.Generator().manual_seed(3)
for (M, N, K) in [(512, 384, 20000), (512, 1024, 256), (1024, 256, 512), (512, 256, 40000)]:
    # rows of A differ in magnitude by e^(+-1) (cells of different depth); one in eight entries is 2^-10 of the rest
    a = torch.randn(M, K, generator=g, dtype=torch.float64) * torch.exp(torch.randn(M, 1, generator=g, dtype=torch.float64))
    a = a * torch.where(torch.rand(M, K, generator=g) < 0.125, 2.0 ** -10, 1.0)
    b = torch.randn(K, N, generator=g, dtype=torch.float64) * 0.05
    scale = (a.float().double().abs() @ b.float().double().abs())
    for lay, name in ((ops.NT, "NT"), (ops.NN, "NN"), (ops.TN, "TN")):
        A = (a if lay != ops.TN else a.t().contiguous()).float().cuda()
        Bm = (b.t().contiguous() if lay == ops.NT else b).float().cuda()
        ref32 = (A.double().cpu() if lay != ops.TN else A.double().cpu().t()) @ (Bm.double().cpu().t() if lay == ops.NT else Bm.double().cpu())
        for P in ("0", "S", "1"):
            out = (ops.gemm_scaled(lay, A, Bm) if P == "S" else ops.gemm(lay, A, Bm, precision=int(P))).double().cpu()
            err = ((out - ref32).abs() / scale).max().item()
            rel = ((out - ref32).norm() / ref32.norm()).item()
            print(f"{M}x{N}x{K} {name} P={P}: max|err|/sum|ab| = {err:.3e}   rel fro = {rel:.3e}")
