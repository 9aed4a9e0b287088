# Plain bench A/B of a bench.py flag on ONE box: FLAG="--scan-weights" CFGS="cfg3 cfg5" bash tools/ab_flag.sh
for cfg in ${CFGS:-cfg3}; do
  for rep in 1 2; do
    for v in default flag; do
      if [ "$v" = flag ]; then extra="$FLAG"; else extra=""; fi
      timeout -k 10 200 python bench.py --config $cfg --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --no-breakdown $extra > gpurun_out/ab_flag_${v}_$cfg.log 2>&1 || exit 6
      echo "$cfg $v $(grep '^{' gpurun_out/ab_flag_${v}_$cfg.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
    done
  done
done
