run() { # label, env..., prec, scene
  python bench.py --config c5 --scene $SC --prec $PR --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; t=r['traversal']; print('$LABEL', j['config']['scene'], j['dtype'], 'ms', j['ms_per_step'], 'walk_ms', t['walk_launch_ms'], 'trips/ray', t['trips_per_ray'], 'lane_util', t['walk_lane_utilisation'])"
}
for SC in s5 s4; do for PR in f64; do
  LABEL=base run
  LABEL=dual64 SPIRA_HIP_LIB=$GRAFT_REPO_ROOT/julia-spira_amd/csrc/libspira_hip_dual64.so run
done; done
for SC in s5 s4; do for PR in f32 f64; do
  for fw in 8 32; do LABEL=fat$fw SPIRA_MESH_FAT_WAVES_PER_CU=$fw run; done
  for rf in 8 32; do LABEL=refill$rf SPIRA_MESH_REFILL=$rf run; done
  LABEL=base run
done; done
