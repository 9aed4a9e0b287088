for k in NONE VAE_STAG_DGRAD=8 VAE_STAG_ENCW=8 VAE_STAG_DECW=8 VAE_STAG_ENCF=8 VAE_STAG_DECF=6 VAE_STAG_DECF=20; do
  echo "== $k"
  if [ "$k" != NONE ]; then export $k; fi
  CFG=cfg2 TAG=stag STEPS=200 bash tools/run_prof.sh 2>&1 | grep -E "ms/step|gemm_x3" | cut -c1-120
  if [ "$k" != NONE ]; then unset ${k%%=*}; fi
done
