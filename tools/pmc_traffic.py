"""profiles/pmc_traffic.json (what bench.py's roofline.traffic reads) from the committed counter summaries of tools/run_pmc_step.sh:
   python3 tools/pmc_traffic.py profiles/r03_s_cfg2_pmc_summary.json cfg2/fp32/B512 profiles/r03_s_cfg3_pmc_summary.json cfg3/bf16/B4096"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# bench.py's kernel names -> the kernel of the step (substring of the summary's row name), per configuration
MAP = {
    "cfg2/fp32/B512": {"gemm_dec_wgrad": "gemm_x3_kernel<2, 160, 128", "gemm_dec_fwd": "gemm_x3_kernel<0, 128, 160", "gemm_dec_dgrad": "gemm_x3_kernel<1, 128, 128",
                       "gemm_enc_wgrad": "gemm_x3_kernel<2, 128, 128", "gemm_enc_fwd": "gemm_x3_kernel<0, 128, 128", "nbzinb_fwd": "nbzinb_fwd_tiled_kernel",
                       "nbzinb_bwd": "nbzinb_bwd_kernel", "row_stats": "row_stats_kernel"},
    "cfg3/bf16/B4096": {"gemm_dec_wgrad": "gemm16_dma_kernel<2, 160, 128", "gemm_dec_fwd": "gemm_x3_kernel<0, 128, 160", "gemm_dec_dgrad": "gemm16_dma_kernel<1, 128, 128",
                        "gemm_enc_wgrad": "gemm16_dma_kernel<2, 128, 128", "gemm_enc_fwd": "gemm16_dma_kernel<0, 128, 128", "nbzinb_fwd": "nbzinb_fwd_tiled_kernel",
                        "nbzinb_bwd": "nbzinb_bwd_kernel"},
}
out = {}
for path, key in zip(sys.argv[1::2], sys.argv[2::2]):
    rows = json.load(open(path))["kernels"]
    out[key] = {}
    for name, sub in MAP[key].items():
        hit = [(k, r) for k, r in rows.items() if sub in k]
        if not hit:
            continue
        k, r = hit[0]
        out[key][name] = {"fetch_bytes": int(r["fetch_mb"] * 1e6), "write_bytes": int(r["write_mb"] * 1e6), "mfma_util": r["mfma_util"] or 0.0,
                          "valu_busy": r["valu_busy"], "in_step_us": r["avg_us"],
                          "source": f"{os.path.relpath(path, ROOT)[:-5]}.txt ({k[:44]}): rocprofv3 --pmc FETCH_SIZE (KB, x2 per the gfx950 correction) and --pmc WRITE_SIZE, "
                                    f"separate passes over bench.py --config {key.split('/')[0]}, in-step launches"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print({k: list(v) for k, v in out.items()})
