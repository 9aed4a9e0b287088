# End-of-round capture on ONE box: GPU suite, the driver's bench call, steady-state bench + kernel summary at cfg2, then bench
# lines and kernel summaries of cfg3 / cfg4 / cfg5.  TAG=z bash tools/run_final_r3.sh  ->  gpurun_out/r3_<TAG>_*, gpurun_out/zb/*
set -o pipefail
TAG=${TAG:-z} bash tools/r3_check.sh || exit $?
mkdir -p gpurun_out/zb
for cfg in ${CFGS:-cfg3 cfg4 cfg5}; do
  timeout -k 10 300 python bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/zb/${cfg}_bench.log 2>&1 || exit 6
  tail -1 gpurun_out/zb/${cfg}_bench.log > gpurun_out/zb/${cfg}_bench.json
  python -c "import json; r=json.load(open('gpurun_out/zb/${cfg}_bench.json')); print('$cfg', r['ms_per_step'], r['value'])"
  CFG=$cfg STEPS=30 TAG=zb bash tools/run_prof.sh > gpurun_out/zb/${cfg}_prof.txt 2>&1 || exit 7
done
