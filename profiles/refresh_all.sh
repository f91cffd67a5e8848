#!/bin/bash
# Re-collects every committed rocprofv3 summary of the round (GPU box): S1 and S3 at the headline shape, S4 = config 5 (mesh, depth 12), both precisions.
# usage: profiles/refresh_all.sh [round tag, default r03]     -> gpurun_out/prof_<tag>_<scene>_<prec>/ ; then profiles/summarize_all.sh <tag>
cd "$(dirname "$0")/.." || exit 1
R=${1:-r03}
for sc in s1 s3 s4; do for pr in f64 f32; do
  tag=${R}_${sc}_${pr}
  [ "$sc" = s4 ] && extra="--config c5" || extra="--scene $sc"
  rm -rf gpurun_out/prof_$tag
  profiles/run_profile.sh $tag $extra --prec $pr > gpurun_out/prof_$tag.log 2>&1
  PASSES="mix1 f64mix" profiles/run_profile_detail.sh $tag $extra --prec $pr >> gpurun_out/prof_$tag.log 2>&1
  # the raw per-dispatch CSVs are large: keep what summarize.py reads, drop the rest before gpurun merges the directory back
  find gpurun_out/prof_$tag -name "*.db" -delete
  echo "done $tag $(date +%T)"
done; done
