#!/bin/bash
# SQ/LDS counters of one workload: bash tools/pmc_case.sh <case> <tag>
set -e -o pipefail
CASE=${1:-seed_small}; TAG=${2:-c}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
CMD="python3 tools/run_case.py $CASE 3"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -o pmc -- $CMD > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq2" -o pmc -- $CMD > "$OUT/b.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/sq3" -o pmc -- $CMD > "$OUT/c.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
for sub in ("sq", "sq2", "sq3"):
    fs = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:50]
        if "rt_" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in sorted(d.items()): print(f"    {c:30s} {sum(v)/len(v):.4g}")
PY
