# rocprofv3 kernel summary of one bench configuration: CFG=cfg2 TAG=x [BENCH_ARGS=...] bash tools/run_prof.sh
CFG=${CFG:-cfg2}; TAG=${TAG:-p}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2${TAG}_$CFG -o k -- python bench.py --config $CFG --steps ${STEPS:-50} --warmup 10 --no-cpu-baseline --no-breakdown ${BENCH_ARGS:-} > gpurun_out/r2_prof_${TAG}_$CFG.log 2>&1 || exit 6
grep "^{" gpurun_out/r2_prof_${TAG}_$CFG.log | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms/step under the profiler', r['ms_per_step'])"
f=$(find gpurun_out/prof_r2${TAG}_$CFG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    if r['Name'].startswith('void at::') or 'at::native' in r['Name']: continue
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.2f} us")
PY
