# Bench lines, kernel summaries and the counter table of the multi-layer configurations on ONE box (profiles/r02_zb_*)
set -o pipefail
mkdir -p gpurun_out/zb
for cfg in ${CFGS:-cfg3 cfg4 cfg5}; do
  timeout -k 10 300 python bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/zb/${cfg}_bench.log 2>&1 || exit 6
  tail -1 gpurun_out/zb/${cfg}_bench.log > gpurun_out/zb/${cfg}_bench.json
  python -c "import json; r=json.load(open('gpurun_out/zb/${cfg}_bench.json')); print('$cfg', r['ms_per_step'], r['value'])"
  CFG=$cfg STEPS=30 TAG=zb bash tools/run_prof.sh > gpurun_out/zb/${cfg}_prof.txt 2>&1 || exit 7
done
CFG=cfg4 TAG=zb STEPS=3 bash tools/run_pmc_step.sh > gpurun_out/zb/cfg4_pmc.txt 2>&1 || exit 8
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown 2>&1 | tail -1 > gpurun_out/zb/cfg2_bench.json
python -c "import json; r=json.load(open('gpurun_out/zb/cfg2_bench.json')); print('cfg2', r['ms_per_step'], r['value'])"
