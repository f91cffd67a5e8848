#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for sc in s1 s3; do for pr in f64 f32; do
  tag=r02_${sc}_${pr}
  rm -rf gpurun_out/prof_$tag
  profiles/run_profile.sh $tag --scene $sc --prec $pr > gpurun_out/prof_$tag.log 2>&1
  PASSES="mix1 f64mix" profiles/run_profile_detail.sh $tag --scene $sc --prec $pr >> gpurun_out/prof_$tag.log 2>&1
  echo "done $tag $(date +%T)"
done; done
