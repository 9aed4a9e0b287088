# PMC passes over the epilogue kernels alone (tools/prof_epilogue.py): one rocprofv3 --pmc run per counter group.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_epi
B=${B:-512}; G=${G:-20000}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_epi/$tag -o p -- python3 tools/prof_epilogue.py $B $G 3 > gpurun_out/pmc_epi/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 gpurun_out/pmc_epi/$tag.log; exit 7; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_epi/*/p_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "nbzinb" in k:
            acc[k.split("<")[0].split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
