# Plain bench A/B of an environment knob on ONE box: KNOB="VAE_STEP_OVERLAP=0" CFGS="cfg2" bash tools/ab_env.sh
for cfg in ${CFGS:-cfg2}; do
  for rep in 1 2 3; do
    for v in default knob; do
      if [ "$v" = knob ]; then pre="env $KNOB"; else pre=""; fi
      $pre timeout -k 10 200 python bench.py --config $cfg --steps ${STEPS:-300} --warmup 50 --no-cpu-baseline --no-breakdown > gpurun_out/ab_env_${v}_$cfg.log 2>&1 || exit 6
      echo "$cfg $v $(tail -1 gpurun_out/ab_env_${v}_$cfg.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
    done
  done
done
