# LDS / wait counters over the whole step, one pass (8 SQ counters): CFG=cfg2 TAG=a bash tools/run_pmc_lds.sh
# per kernel: share of busy cycles the LDS pipe is active, share of those lost to bank conflicts, wave cycles spent waiting
CFG=${CFG:-cfg2}; TAG=${TAG:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_lds_${CFG}_$TAG
mkdir -p $out
args="bench.py --config $CFG --steps ${STEPS:-5} --warmup 2 --no-breakdown --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL --output-format csv -d $out/lds -o p -- python3 $args > $out/lds.log 2>&1 || { echo "pass lds failed"; tail -5 $out/lds.log; exit 7; }
python3 - $out <<'PY' | tee $out/summary.txt
import collections, csv, glob, re, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/lds/**/p_counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), d in per.items():
        for c, v in d.items():
            acc[re.sub(r"\(anonymous namespace\)::|^void ", "", k)[:70]][c].append(v)
print(f"{'kernel':70s} {'lds_act/busy':>12s} {'conflict/act':>12s} {'wait_lds/wave':>13s} {'wait_any/wave':>13s} {'fifo_full/busy':>14s}")
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get('SQ_BUSY_CYCLES', 0) <= 0 or 'at::' in k: continue
    act = m.get('SQ_LDS_IDX_ACTIVE', 0.0)
    print(f"{k:70s} {act / m['SQ_BUSY_CYCLES']:12.3f} {(m.get('SQ_LDS_BANK_CONFLICT', 0) / act if act else 0):12.3f} "
          f"{m.get('SQ_WAIT_INST_LDS', 0) / m['SQ_WAVE_CYCLES']:13.3f} {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:13.3f} "
          f"{m.get('SQ_LDS_DATA_FIFO_FULL', 0) / m['SQ_BUSY_CYCLES']:14.3f}")
PY
