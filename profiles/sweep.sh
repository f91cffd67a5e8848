#!/bin/bash
# Parameter sweep helper (GPU box): prints "<env settings> <scene> -> Msamples/s, roofline frac".
# usage: profiles/sweep.sh "<VAR=val ...>" ["<VAR=val ...>" ...]   (scene via SCENES="s1 s3"; extra bench args via BENCH_ARGS)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for cfg in "$@"; do
  for sc in ${SCENES:-s1 s3}; do
    line=$(env $cfg timeout -k 10 120 python3 bench.py --steps ${STEPS:-3} --no-cpu-baseline --no-alt-precision --no-extras --scene $sc ${BENCH_ARGS:-} 2>&1 | tail -1)
    echo "$cfg ${BENCH_ARGS:-} $sc -> $(echo "$line" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}
    print("%.0f Msamples/s  %.2f ms/step  frac=%s  GB/s=%s  B/sample=%s seg/sample=%s kernel_ms=%s" % (d["value"], d["ms_per_step"], r.get("frac"), r.get("achieved"), r.get("bytes_per_sample"), r.get("segments_per_sample"), r.get("kernel_ms_per_step")))
except Exception as e:
    print("FAILED", e)')" | tee -a gpurun_out/sweep.log
  done
done
