# dense-continuation threshold (SPIRA_DENSE_PCT) and workgroups per CU on the round's final kernels; S1 and configs[4]
run() {
  python bench.py --config $CF --prec $PR --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-alt-precision 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$LABEL', j['config']['scene'], j['dtype'], 'ms', j['ms_per_step'])"
}
for rep in 1 2; do for PR in f64 f32; do
  for d in 80 75 70 65; do CF=c3 LABEL=dense$d SPIRA_DENSE_PCT=$d run; done
  for d in 80 70; do CF=c5 LABEL=dense$d SPIRA_DENSE_PCT=$d run; done
done; done
