#!/bin/bash
# One PMC pass (SQ instruction mix) over tools/prof_run.py; prints per-wave figures of the render kernels.
# Usage (through gpurun, from the repo root): bash tools/prof_quick.sh TAG [prof_run.py arguments]
set -u
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    -d $OUT/pmc -o p --output-format csv -- python3 $R/tools/prof_run.py "$@" > $OUT/pmc.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/pmc.log; exit 1; }
tail -1 $OUT/pmc.log
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ptr::" not in k or int(r["Grid_Size"]) < 10_000_000: continue
        agg[(k.split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
ks = sorted({k for k, _ in agg})
for k in ks:
    m = {c: sum(v) / len(v) for (kk, c), v in agg.items() if kk == k}
    w = m["SQ_WAVES"]
    print(k, "waves %.0f  per wave: VALU %.1f SALU %.1f SMEM %.1f  active-lane frac %.3f  wave cycles %.0f  wait-any frac %.3f" % (
        w, m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_SMEM"] / w,
        m["SQ_ACTIVE_INST_VALU"] / max(1, m["SQ_INSTS_VALU"]) / 4 if False else 0.0, m["SQ_WAVE_CYCLES"] / w,
        m["SQ_WAIT_INST_ANY"] / max(1, m["SQ_WAVE_CYCLES"])))
PY
