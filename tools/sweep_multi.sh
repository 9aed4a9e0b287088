# Several environment settings on ONE box, kernel profile each: SETTINGS="A=1 B=2|A=3|" PAT=... bash tools/sweep_multi.sh
CFG=${CFG:-cfg2}
IFS='|' read -ra SETS <<< "$SETTINGS"
i=0
for st in "${SETS[@]}" ""; do
  echo "== [$st]"
  env $st CFG=$CFG TAG=sm_$i STEPS=${STEPS:-200} bash tools/run_prof.sh 2>&1 | grep -E "${PAT:-ms/step|gemm_x3}" || exit 7
  i=$((i+1))
done
