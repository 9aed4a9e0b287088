# Kernel-profile A/B of an environment knob on ONE box: KNOB="VAE_NO_XDIRECT=1" [CFG=cfg2] [REPS=2] bash tools/ab_prof.sh
CFG=${CFG:-cfg2}
for rep in $(seq 1 ${REPS:-2}); do
  for v in default knob; do
    echo "== $v (rep $rep)"
    if [ "$v" = knob ]; then export $KNOB; else unset ${KNOB%%=*}; fi
    CFG=$CFG TAG=ab_${v}_$rep STEPS=${STEPS:-300} bash tools/run_prof.sh 2>&1 | grep -E "gemm|ms/step|nbzinb|mid_|row_stats|normalize|small_wgrad|elbo|merge" || exit 7
  done
done
