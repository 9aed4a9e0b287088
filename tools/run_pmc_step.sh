# Counter passes over the whole step (bench.py, no breakdown / CPU baseline): one rocprofv3 --pmc run per counter group,
# no trace flags beside them, the program itself after `--`.  CFG=cfg2|cfg3..., TAG names the output directory.
#   pass mfma : MFMA busy cycles, MFMA instruction counts and op counts (f16 / bf16), VALU, wave cycles, GRBM_GUI_ACTIVE
#   pass fetch: FETCH_SIZE (3 TCC slots)      pass write: WRITE_SIZE (2 TCC slots)
# tools/pmc_summary.py turns the three CSVs (+ an un-profiled kernel trace for the durations) into one table and JSON.
CFG=${CFG:-cfg2}; TAG=${TAG:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_step_${CFG}_$TAG
mkdir -p $out
args="bench.py --config $CFG --steps ${STEPS:-5} --warmup 2 --no-breakdown --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o k -- python3 $args > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 6; }
run_pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o p -- python3 $args > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; exit 7; }
}
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
python3 tools/pmc_summary.py $out $CFG | tee $out/summary.txt
