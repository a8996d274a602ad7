#!/bin/bash
# quick SQ-counter passes over the kernels: bash tools/pmc_quick.sh <tag>
set -e -o pipefail
TAG=${1:-q}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -o pmc -- $CMD > "$OUT/bench.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq2" -o pmc -- $CMD > "$OUT/bench2.log" 2>&1 || true
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d "$OUT/sq3" -o pmc -- $CMD > "$OUT/bench3.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
for sub in ("sq", "sq2", "sq3"):
    fs = glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = defaultdict(lambda: defaultdict(list)); meta = {}
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("Scratch_Size"))
    for k, d in acc.items():
        if "rt_" not in k: continue
        print(k, "vgpr/sgpr/lds/grid/wg/scratch", meta[k])
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {sum(v)/len(v):.4g}")
PY
