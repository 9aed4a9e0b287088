# A/B of the likelihood backward's genes per thread on ONE box: [CFG=cfg2] bash tools/ab_ll_ept.sh "1280 1536"   (workgroup targets of the two-gene form)
CFG=${CFG:-cfg2}
run() { CFG=$CFG TAG=ept_$1 STEPS=${STEPS:-300} bash tools/run_prof.sh 2>&1 | grep -E "nbzinb|ms/step|gemm_x3_kernel<2, 160|small_wgrad" || exit 7; }
echo "== four genes per thread"; export VAE_LL_BWD_EPT=4; run e4
for t in $1; do
  echo "== two genes per thread, target $t"; export VAE_LL_BWD_EPT=2 VAE_BWD_WORKGROUPS2=$t; run e2_$t
done
echo "== four genes per thread, again"; export VAE_LL_BWD_EPT=4; run e4b
