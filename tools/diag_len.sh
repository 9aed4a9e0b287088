# cfg2 step time against warm-up / run length on one box
for sw in "50 10" "100 20" "300 50" "1000 100" "50 10" "300 50"; do
  set -- $sw
  echo "steps $1 warmup $2: $(timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-breakdown 2>&1 | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
done
