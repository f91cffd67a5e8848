#!/bin/bash
# Re-collects every committed rocprofv3 summary of the round (GPU box): S1 and S3 at the headline shape, S4 = config 5 (mesh, depth 12), S5 = the mesh stress scene,
# both precisions; and the extension instantiations (EXT = true): config 5 + SPIRA_EXT_SPECTRAL, the glass scene + both extensions.
# usage: profiles/refresh_all.sh [round tag, default r04] [scenes, default "s1 s3 s4 s5 s4_ext s2g_ext"]   -> gpurun_out/prof_<tag>_<scene>_<prec>/ ; then profiles/summarize_all.sh <tag>
cd "$(dirname "$0")/.." || exit 1
R=${1:-r04}
SCENES=${2:-"s1 s3 s4 s5 s4_ext s2g_ext"}
for sc in $SCENES; do for pr in f64 f32; do
  tag=${R}_${sc}_${pr}
  case $sc in
    s4) extra="--config c5" ;;
    s5) extra="--config c5 --scene s5" ;;
    s4_ext) extra="--config c5 --ext spectral" ;;
    s2g_ext) extra="--scene s2g --ext both" ;;
    *) extra="--scene $sc" ;;
  esac
  rm -rf gpurun_out/prof_$tag
  profiles/run_profile.sh $tag $extra --prec $pr > gpurun_out/prof_$tag.log 2>&1
  PASSES="mix1 f64mix" profiles/run_profile_detail.sh $tag $extra --prec $pr >> gpurun_out/prof_$tag.log 2>&1
  # the raw per-dispatch CSVs are large: keep what summarize.py reads, drop the rest before gpurun merges the directory back
  find gpurun_out/prof_$tag -name "*.db" -delete
  echo "done $tag $(date +%T)"
done; done
