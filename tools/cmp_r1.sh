# same-box comparison with the end-of-round-1 tree (copied to gpurun_in_r1/, not committed): short and long runs
#   git worktree add /tmp/r1tree dea417deba67 && make -C /tmp/r1tree/vae_amd/csrc -j8
#   mkdir gpurun_in_r1 && (cd /tmp/r1tree && cp -r bench.py vae_amd oracle include tools __graft_entry__.py $OLDPWD/gpurun_in_r1/)
#   gpurun -- 'bash tools/cmp_r1.sh'        (gpurun_in_r1/ is git-ignored; remove it and the worktree afterwards)
for rep in 1 2; do
  for sw in "20 5" "300 50"; do
    set -- $sw
    a=$(cd gpurun_in_r1 && timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-breakdown 2>&1 | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-breakdown 2>&1 | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "steps $1 warmup $2: round-1 tree $a   this tree $b"
  done
done
