run() {
  python bench.py --config c3 --scene $SC --prec f64 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-alt-precision $EXTRA 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$LABEL', j['config']['scene'], j['dtype'], 'ms', j['ms_per_step'])"
}
for rep in 1 2; do
  for SC in s3 s2; do
  LABEL=tri_waves4 run
  LABEL=tri_waves3 SPIRA_HIP_LIB=$GRAFT_REPO_ROOT/julia-spira_amd/csrc/libspira_hip_ab.so run
  done
  SC=s2g EXTRA="--ext both" LABEL=glass_ext_now run
done
