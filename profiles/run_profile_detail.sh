#!/bin/bash
# Extra PMC passes for bottleneck analysis of the dominant kernel (k_path / k_bounce) (instruction mix, instruction cache, occupancy, latency).
# usage: profiles/run_profile_detail.sh <tag> [extra bench.py args]
set -u
TAG=${1:-detail}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision --no-extras $*"
pmc() { local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 $BENCH > "$OUT/pmc_$name.log" 2>&1
    echo "pmc $name rc=$?"; }
want() { [ -z "${PASSES:-}" ] || [[ " $PASSES " == *" $1 "* ]]; }     # PASSES="mix1 f64mix" runs only those
pmc_if() { want "$1" && pmc "$@"; }
pmc_if mix1 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU
pmc_if mix2 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VSKIPPED SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
pmc_if icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pmc_if ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
pmc_if dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc_if f64mix SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64
