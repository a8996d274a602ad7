#!/bin/bash
# Profiles `python bench.py` on the GPU box with rocprofv3 (run through gpurun):
#   pass 1: --kernel-trace --stats          -> per-kernel time
#   pass 2: --pmc FETCH_SIZE                -> HBM read side  (TCC, 3 slots)
#   pass 3: --pmc WRITE_SIZE                -> HBM write side (TCC, 2 slots)
#   pass 4: --pmc SQ counters               -> VALU utilisation / occupancy
# PMC passes are separate runs with no tracing domains beside --kernel-trace
# (MI355X_MICROARCH.md "HBM", "rocprofv3 PMC slots").  Outputs land under
# gpurun_out/prof_<tag>/ ; summarise with profiles/summarize.py.
set -e -o pipefail
TAG=${1:-r01}
STEPS=${2:-20}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/bench_trace.log" 2>&1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $CMD > "$OUT/bench_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $CMD > "$OUT/bench_write.log" 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o pmc -- $CMD > "$OUT/bench_sq.log" 2>&1 || echo "sq pass failed"
echo "sq done"
python3 profiles/summarize.py "$OUT" "$TAG" > "$OUT/summary.json"; tail -5 "$OUT/bench_trace.log"
