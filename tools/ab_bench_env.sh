# Plain bench A/B of an environment knob on ONE box (no profiler): KNOB="VAE_NO_BWD_FORK=1" [REPS=3] bash tools/ab_bench_env.sh
for rep in $(seq 1 ${REPS:-3}); do
  for v in default knob; do
    if [ "$v" = knob ]; then export $KNOB; else unset ${KNOB%%=*}; fi
    timeout -k 10 200 python bench.py --config ${CFG:-cfg2} --steps ${STEPS:-300} --warmup 50 --no-cpu-baseline --no-breakdown > gpurun_out/ab_env_$v.log 2>&1 || exit 6
    echo "$v $(grep '^{' gpurun_out/ab_env_$v.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
  done
done
