# Kernel-profile A/B of library variants (tools/build_variant.sh) on ONE box: VARIANTS="default noreload rl_sb" [CFG=cfg2] [REPS=1] bash tools/ab_variants.sh
CFG=${CFG:-cfg2}
for rep in $(seq 1 ${REPS:-1}); do
  for v in $VARIANTS; do
    echo "== $v (rep $rep)"
    if [ "$v" = default ]; then unset VAE_AMD_LIB; else export VAE_AMD_LIB=$PWD/vae_amd/lib/variants/libvae_amd_$v.so; fi
    CFG=$CFG TAG=abv_${v}_$rep STEPS=${STEPS:-300} bash tools/run_prof.sh 2>&1 | grep -E "gemm|ms/step" || exit 7
  done
done
