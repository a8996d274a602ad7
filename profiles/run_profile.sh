#!/bin/bash
# Profiles `python bench.py` on the GPU box with rocprofv3 (run through gpurun):
#   pass 1: --kernel-trace --stats          -> per-kernel time
#   pass 2: --pmc FETCH_SIZE                -> HBM read side  (TCC, 3 slots)
#   pass 3: --pmc WRITE_SIZE                -> HBM write side (TCC, 2 slots)
#   pass 4/5: --pmc SQ counters             -> instruction counts / issue utilisation, shader clock
#   pass 6-7: trace + SQ pass of the two-kernel form of the step (RT_HIP_FUSED=2)
#   then trace / FETCH / WRITE / SQ passes for BASELINE config 5 (--workload config5)
#   and for the seeded half of config 3 (--workload seed_medium)
#   usage: bash profiles/run_profile.sh <tag> <steps> <git head the snapshot was taken at>
# PMC passes are separate runs with no tracing domains beside --kernel-trace
# (MI355X_MICROARCH.md "HBM", "rocprofv3 PMC slots").  Outputs land under
# gpurun_out/prof_<tag>/ ; summarise with profiles/summarize.py.
set -e -o pipefail
TAG=${1:-r05}
STEPS=${2:-20}
HEAD=${3:-unknown}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps $STEPS --warmup 3 --no-extras"
# (the trace pass is the default bench command's timed loop: its last $STEPS launches are the timed ones of the bench
# line; summarize.py averages those beside the all-calls average)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/bench_trace.log" 2>&1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $CMD > "$OUT/bench_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $CMD > "$OUT/bench_write.log" 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o pmc -- $CMD > "$OUT/bench_sq.log" 2>&1 || echo "sq pass failed"
echo "sq done"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq2" -o pmc -- $CMD > "$OUT/bench_sq2.log" 2>&1 || echo "sq2 pass failed"
echo "sq2 done"
# the same step as two launches, for the per-kernel durations and lane utilisation of the march alone
export RT_HIP_FUSED=2
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2" -o trace -- $CMD > "$OUT/bench_trace2.log" 2>&1 || echo "trace2 failed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq_2k" -o pmc -- $CMD > "$OUT/bench_sq_2k.log" 2>&1 || echo "sq 2k pass failed"
unset RT_HIP_FUSED
echo "two-kernel passes done"
C5="python3 bench.py --workload config5 --steps 3 --warmup 1 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_trace" -o trace -- $C5 > "$OUT/c5_trace.log" 2>&1 || echo "c5 trace failed"
echo "c5 trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/c5_fetch" -o pmc -- $C5 > "$OUT/c5_fetch.log" 2>&1 || echo "c5 fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/c5_write" -o pmc -- $C5 > "$OUT/c5_write.log" 2>&1 || echo "c5 write failed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/c5_sq" -o pmc -- $C5 > "$OUT/c5_sq.log" 2>&1 || echo "c5 sq failed"
echo "c5 pmc done"
SM="python3 bench.py --workload seed_medium --steps 3 --warmup 1 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sm_trace" -o trace -- $SM > "$OUT/sm_trace.log" 2>&1 || echo "sm trace failed"
echo "sm trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/sm_fetch" -o pmc -- $SM > "$OUT/sm_fetch.log" 2>&1 || echo "sm fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/sm_write" -o pmc -- $SM > "$OUT/sm_write.log" 2>&1 || echo "sm write failed"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sm_sq" -o pmc -- $SM > "$OUT/sm_sq.log" 2>&1 || echo "sm sq failed"
echo "sm pmc done"
python3 profiles/summarize.py "$OUT" "$TAG" "$HEAD" "$STEPS" > "$OUT/summary.json"; tail -3 "$OUT/bench_trace.log"
