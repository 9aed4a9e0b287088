# Kernel-profile sweep of one environment variable on ONE box: VAR=VAE_BWD_WORKGROUPS VALUES="512 760 1024" [PAT=...] bash tools/sweep_env.sh
CFG=${CFG:-cfg2}
for v in $VALUES; do
  export $VAR=$v
  echo "== $VAR=$v"
  CFG=$CFG TAG=sw_$v STEPS=${STEPS:-200} bash tools/run_prof.sh 2>&1 | grep -E "${PAT:-ms/step|nbzinb}" || exit 7
done
