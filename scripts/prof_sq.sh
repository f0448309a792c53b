#!/bin/bash
# SQ-level counters for the bench kernels (own PMC passes, kernel-trace only).  usage: bash scripts/prof_sq.sh <tag> [bench args]
set -o pipefail
TAG=${1:-sq}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 $ROOT/bench.py "$@" > $OUT/pmc_sq1.log 2>&1 || { echo "sq1 failed"; tail -5 $OUT/pmc_sq1.log; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py "$@" > $OUT/pmc_sq2.log 2>&1 || { echo "sq2 failed"; tail -5 $OUT/pmc_sq2.log; }
cd $ROOT
python3 - <<PY
import csv, glob, os
from collections import defaultdict
for sub in ("pmc_sq1", "pmc_sq2"):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in glob.glob(os.path.join("$OUT", sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "golap" not in k: continue
            acc[k.split("(")[0]][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
    for k, d in acc.items():
        if any(x in k for x in ("scatter", "join_lds", "k_hist")):
            print(sub, k, {c: f"{v:.3e}" for c, v in sorted(d.items())})
PY
