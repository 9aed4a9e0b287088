# Split-K sweep of the two slab products (VAE_GEMM_DMA_SPLIT forces both) with the kernels that read their slabs: SPLITS="32 42 63" bash tools/sweep_split.sh
for sp in ${SPLITS:-32 42 50 63}; do
  echo "== split $sp"
  export VAE_GEMM_DMA_SPLIT=$sp
  CFG=${CFG:-cfg2} TAG=sp_$sp STEPS=${STEPS:-200} bash tools/run_prof.sh 2>&1 | grep -E "ms/step|128, 128, 2, 2, 0, false, [03]>|<1, 128, 128|mid_fwd|mid_bwd" || exit 7
done
