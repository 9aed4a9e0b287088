# Kernel-profile A/B of library variants on ONE box: VARIANTS="occ4" [CFG=cfg2] bash tools/ab_lib.sh   (main = vae_amd/lib/libvae_amd.so)
CFG=${CFG:-cfg2}
for v in main ${VARIANTS}; do
  if [ "$v" = main ]; then unset VAE_AMD_LIB; else export VAE_AMD_LIB=$GRAFT_REPO_ROOT/vae_amd/lib/variants/libvae_amd_$v.so; fi
  echo "== $v"
  CFG=$CFG TAG=abl_$v STEPS=${STEPS:-300} bash tools/run_prof.sh 2>&1 | grep -E "${PAT:-gemm|ms/step|nbzinb|mid_|row_stats|normalize|small_wgrad|elbo|merge}" || exit 7
done
