"""Joins the rocprofv3 passes of tools/run_pmc_step.sh into one per-kernel table:
     python3 tools/pmc_summary.py <dir> <cfg>        -> prints the table, writes <dir>/summary.json
   Per kernel (averages over its dispatches):
     avg_us      un-profiled kernel-trace duration
     mfma_util   sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   [rocprofv3 sums GUI_ACTIVE over 8 XCDs]
     mfma_tflops (MOPS_F16 + MOPS_BF16) * 512 / avg_us      -- operations the matrix cores executed, not algorithmic flops
     fetch_mb    FETCH_SIZE (KB) * 2 / 1000  (gfx950 counts a 128-byte request as 64: MI355X_MICROARCH.md, HBM)
     write_mb    WRITE_SIZE (KB) / 1000"""
import collections
import csv
import glob
import json
import re
import sys

root, cfg = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "cfg2"
SIMDS = 1024


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/{sub}/**/p_counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))     # one dispatch can have several rows per counter
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), d in per.items():
            for c, v in d.items():
                acc[short(k)][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


dur = {}
for f in glob.glob(f"{root}/trace/**/k_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
mf, fe, wr = counters("mfma"), counters("fetch"), counters("write")
rows = {}
for k, (us, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if k.startswith("at::") or "at::native" in k or us * calls < 1.0:
        continue
    m = mf.get(k, {})
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8
    mops = m.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    rows[k] = dict(avg_us=round(us, 2), calls=calls,
                   mfma_busy_cycles=m.get("SQ_VALU_MFMA_BUSY_CYCLES"), mfma_insts=m.get("SQ_INSTS_MFMA"),
                   gui_active_per_xcd=gui or None,
                   mfma_util=round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS), 4) if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in m else None,
                   mfma_tflops=round(mops * 512 / us / 1e6, 1) if mops else 0.0,
                   valu_insts=m.get("SQ_INSTS_VALU"),
                   valu_busy=round(4 * m["SQ_ACTIVE_INST_VALU"] / (gui * SIMDS), 4) if gui and "SQ_ACTIVE_INST_VALU" in m else None,
                   fetch_mb=round(fe[k]["FETCH_SIZE"] * 2 / 1e3, 2) if k in fe else None,
                   write_mb=round(wr[k]["WRITE_SIZE"] / 1e3, 2) if k in wr else None)
json.dump({"config": cfg, "kernels": rows}, open(f"{root}/summary.json", "w"), indent=1)
print(f"{'kernel':110s} {'calls':>5s} {'avg us':>8s} {'MFMA util':>9s} {'MFMA TF':>8s} {'VALU busy':>9s} {'fetch MB':>9s} {'write MB':>9s}")
for k, r in rows.items():
    f = lambda v, n=2: "-" if v is None else f"{v:.{n}f}"
    print(f"{k:110s} {r['calls']:5d} {r['avg_us']:8.2f} {f(r['mfma_util'], 3):>9s} {f(r['mfma_tflops'], 1):>8s} {f(r['valu_busy'], 3):>9s} "
          f"{f(r['fetch_mb']):>9s} {f(r['write_mb']):>9s}")
