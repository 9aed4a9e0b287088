# A/B of library variants on ONE box: VARIANTS="nopipe schedb" CFG=cfg2 bash tools/run_ab.sh  (default lib = "main")
CFG=${CFG:-cfg2}
for v in main ${VARIANTS}; do
  if [ "$v" = main ]; then unset VAE_AMD_LIB; else export VAE_AMD_LIB=$GRAFT_REPO_ROOT/vae_amd/lib/variants/libvae_amd_$v.so; fi
  echo "== $v"
  CFG=$CFG TAG=ab_$v bash tools/run_prof.sh 2>&1 | grep -E "gemm|ms/step" || exit 7
done
