# optimiser step alone and the end-to-end training step, with / without nontemporal moment traffic (one box)
for v in 0 1 0 1; do
  echo "VAE_ADAMW_NT=$v: $(VAE_ADAMW_NT=$v timeout -k 10 120 python tools/prof_optim.py 2>&1 | grep 'fused clip')"
done
for v in 0 1 0 1; do
  echo "e2e VAE_ADAMW_NT=$v: $(VAE_ADAMW_NT=$v timeout -k 10 200 python tools/bench_e2e.py --u8 --steps 300 2>&1 | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
done
