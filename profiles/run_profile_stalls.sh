#!/bin/bash
# PMC passes for the "where do wave-cycles go" question on the dominant kernel — k_path (default organisation), k_path_metal, k_bounce —
# (issue vs wait vs dependency stalls).
# usage: profiles/run_profile_stalls.sh <tag> [extra bench.py args]
set -u
TAG=${1:-stalls}; shift || true
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precision --no-extras $*"
pmc() { local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o pmc -- python3 $BENCH > "$OUT/pmc_$name.log" 2>&1
    echo "pmc $name rc=$?"; }
pmc w1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pmc w2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_SALU
pmc w3 SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_w*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        dom = next((d for d in ("k_path_metal", "k_path", "k_bounce") if d in k), None)      # whichever organisation the bench command ran
        if dom is None: continue
        k = k[k.index(dom):k.index("(")] if "(" in k else k
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
json.dump(agg, open(out + "/stalls.json", "w"), indent=1)
for k, v in agg.items():
    print(k)
    for a, b in sorted(v.items()): print("   %-28s %.4g" % (a, b))
PY
