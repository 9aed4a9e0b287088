# kernel summaries of the end-of-round-1 tree (gpurun_in_r1/) and this tree on one box, steady state
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cmp
(cd gpurun_in_r1 && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cmp/r1 -o k -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-breakdown > $GRAFT_REPO_ROOT/gpurun_out/cmp/r1.log 2>&1) || exit 6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cmp/r2 -o k -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-breakdown > gpurun_out/cmp/r2.log 2>&1 || exit 7
grep '^{' gpurun_out/cmp/r1.log | tail -1 | cut -c1-200
grep '^{' gpurun_out/cmp/r2.log | tail -1 | cut -c1-200
