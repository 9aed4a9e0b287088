timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/x.log 2>&1; tail -2 gpurun_out/x.log
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
  for e in 4 2; do
    VAE_LL_BWD_EPT=$e timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --no-cpu-baseline --no-breakdown > gpurun_out/b_${c}_$e.log 2>&1 || exit 5
    grep "^{" gpurun_out/b_${c}_$e.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c ept$e', d['ms_per_step'])"
  done
done
