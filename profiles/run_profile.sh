#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel (run on the MI355X box via gpurun):
#   1. --kernel-trace --stats  : per-kernel durations of the SAME bench command
#   2. --pmc passes (own runs, no trace domains besides kernel-trace): HBM bytes and SQ activity
# Outputs land under gpurun_out/prof_<tag>/ ; the summaries to be judged are copied into profiles/.
# usage: profiles/run_profile.sh <tag> [extra bench.py args]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# 4 warm-up + 8 timed steps: the first launches of a process run slower (first touch of 25 GB of workspaces, clocks ramping)
BENCH="bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-alt-precision --no-extras $*"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"

pmc() {   # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 $BENCH > "$OUT/pmc_$name.log" 2>&1
    echo "pmc $name rc=$?"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
ls -R "$OUT" | head -60
