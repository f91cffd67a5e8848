#!/bin/bash
# Where does k_bounce's time go?  Builds timing-only variants of the library (results are WRONG by design) that
# drop one part of the segment each, and times them on S3 (closed box: 8 segments/sample whatever the rays do).
#   step 1 (anywhere, no GPU needed):   profiles/ablate.sh build
#   step 2 (GPU box, one call):         profiles/ablate.sh run [f32|f64]
cd "$(dirname "$0")/.." || exit 1
SRC=julia-spira_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function"
declare -A V=( [full]="" [no_hit]="-DSPIRA_ABL_NO_HIT" [no_hit_sampler]="-DSPIRA_ABL_NO_HIT -DSPIRA_ABL_NO_SAMPLER"
               [no_hit_sampler_key]="-DSPIRA_ABL_NO_HIT -DSPIRA_ABL_NO_SAMPLER -DSPIRA_ABL_NO_KEY"
               [no_hit_sampler_key_L]="-DSPIRA_ABL_NO_HIT -DSPIRA_ABL_NO_SAMPLER -DSPIRA_ABL_NO_KEY -DSPIRA_ABL_NO_L"
               [no_L]="-DSPIRA_ABL_NO_L" [no_sampler_key]="-DSPIRA_ABL_NO_SAMPLER -DSPIRA_ABL_NO_KEY" )
ORDER="full no_hit no_hit_sampler no_hit_sampler_key no_hit_sampler_key_L no_L no_sampler_key full"
case "${1:-}" in
build)
    mkdir -p build/ablate
    for v in "${!V[@]}"; do /opt/rocm/bin/hipcc $F ${V[$v]} -shared -o build/ablate/$v.so $SRC/spira_hip.hip 2>/dev/null & done; wait
    ls -la build/ablate ;;
run)
    for v in $ORDER; do
        line=$(SPIRA_HIP_LIB=build/ablate/$v.so timeout -k 10 120 python3 bench.py --steps 3 --scene s3 --prec ${2:-f64} --no-cpu-baseline --no-alt-precision 2>&1 | tail -1)
        echo "$v $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.2f ms/step  algorithmic %.0f GB/s (frac %.3f)" % (d["ms_per_step"], r["achieved"], r["frac"]))')"
    done ;;
*) echo "usage: $0 build | run [f32|f64]"; exit 2 ;;
esac
