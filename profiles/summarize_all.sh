#!/bin/bash
# profiles/summarize_all.sh [round tag]: gpurun_out/prof_<tag>_<scene>_<prec>/ -> profiles/<tag>_<scene>_<prec>.md + .json and profiles/traffic_<scene>_<prec>.json
# (needs profiles/isa_cost.json for the priced compute-side figures: `make -C julia-spira_amd/csrc asm && python3 profiles/isa_cost.py` first)
cd "$(dirname "$0")/.." || exit 1
R=${1:-r04}
for sc in s1 s3 s4 s5 s4_ext s2g_ext; do for pr in f64 f32; do
  tag=${R}_${sc}_${pr}
  [ -d gpurun_out/prof_$tag ] || continue
  python3 profiles/summarize.py gpurun_out/prof_$tag profiles/$tag.md > /dev/null && cp profiles/$tag.json profiles/traffic_${sc}_${pr}.json && echo "summarised $tag"
done; done
