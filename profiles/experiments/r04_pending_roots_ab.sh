run() {
  python bench.py --config $CF --scene $SC --prec $PR --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$LABEL', j['config']['scene'], j['dtype'], 'ms', j['ms_per_step'], 'Msamples/s', j['value'])"
}
for rep in 1 2; do
for CS in "c3 s1" "c3 s3" "c5 s4"; do set -- $CS; CF=$1; SC=$2; for PR in f64 f32; do
  LABEL=pending run
  LABEL=in_place SPIRA_HIP_LIB=$GRAFT_REPO_ROOT/julia-spira_amd/csrc/libspira_hip_ab.so run
done; done; done
