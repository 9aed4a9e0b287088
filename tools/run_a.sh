# GPU round trip: parity suite, bench line, rocprofv3 kernel summary (one gpurun call).  TAG names the outputs.
TAG=${TAG:-a}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider -s > gpurun_out/r2_tests_$TAG.log 2>&1
rc=$?
tail -8 gpurun_out/r2_tests_$TAG.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 100 --warmup 20 ${BENCH_ARGS:-} > gpurun_out/r2_bench_$TAG.json 2> gpurun_out/r2_bench_$TAG.err || exit 5
python - <<PY
import json
r = json.load(open("gpurun_out/r2_bench_$TAG.json"))
print("ms/step", r["ms_per_step"], "cells/s", r["value"], "roofline", r["roofline"]["kernel"], r["roofline"]["frac"])
print({k: round(v["ms"] * 1e3, 1) for k, v in r["kernels"].items()})
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2$TAG -o k -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-breakdown ${BENCH_ARGS:-} > gpurun_out/r2_prof_$TAG.log 2>&1 || exit 6
f=$(find gpurun_out/prof_r2$TAG -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.2f} us")
PY
