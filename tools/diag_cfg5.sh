# cfg5 step time against run length, optimiser-kept bf16 weight images vs the per-step cast (clock / power phases of a box)
run() { echo "$1: $(timeout -k 10 200 python bench.py --config ${CFG:-cfg5} --warmup 20 --no-cpu-baseline --no-breakdown $2 2>&1 | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"; }
for n in 30 100 300 600; do
  run "images steps$n" "--steps $n"
  run "cast   steps$n" "--steps $n --scan-weights"
done
