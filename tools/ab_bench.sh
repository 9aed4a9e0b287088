# Plain bench A/B of library variants on ONE box (no profiler): VARIANTS="head" CFGS="cfg4 cfg5" bash tools/ab_bench.sh
for cfg in ${CFGS:-cfg2}; do
  for rep in 1 2; do
    for v in main ${VARIANTS}; do
      if [ "$v" = main ]; then unset VAE_AMD_LIB; else export VAE_AMD_LIB=$GRAFT_REPO_ROOT/vae_amd/lib/variants/libvae_amd_$v.so; fi
      timeout -k 10 200 python bench.py --config $cfg --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-breakdown > gpurun_out/ab_bench_${v}_$cfg.log 2>&1 || exit 6
      echo "$cfg $v $(tail -1 gpurun_out/ab_bench_${v}_$cfg.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"
    done
  done
done
