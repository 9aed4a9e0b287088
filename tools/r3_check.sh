# Round-3 GPU check: full GPU suite, the driver's bench call, a steady-state bench and a kernel profile.
# TAG=x [SKIP_TESTS=1] bash tools/r3_check.sh   (run through gpurun)
TAG=${TAG:-a}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/r3_${TAG}_tests.log; exit 5; }
  tail -2 gpurun_out/r3_${TAG}_tests.log
fi
timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_${TAG}_bench20.json 2>gpurun_out/r3_${TAG}_bench20.err || { tail -20 gpurun_out/r3_${TAG}_bench20.err; exit 6; }
timeout -k 10 120 python bench.py --no-cpu-baseline > gpurun_out/r3_${TAG}_bench300.json 2>/dev/null || exit 7
python - <<PY
import json
for n in ("20", "300"):
    r = json.loads([l for l in open("gpurun_out/r3_${TAG}_bench%s.json" % n) if l.startswith("{")][-1])
    print("bench", n, "ms/step %.4f  cells/s %.0f" % (r["ms_per_step"], r["value"]), r.get("scan_weights", {}).get("ms_per_step_scan_weights"), r.get("scan_weights", {}).get("ms_per_step_headline_repeat"))
    if n == "300": print({k: round(v["ms"] * 1e3, 1) for k, v in r["kernels"].items()})
PY
CFG=${CFG:-cfg2} TAG=r3${TAG} STEPS=300 bash tools/run_prof.sh
