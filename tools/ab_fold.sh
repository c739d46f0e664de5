export RPO_VERBOSE=0
for W in ${1:-cart_ddpg}; do
  for F in 0 1 0 1; do
    RPO_FOLD_ADAM=$F python bench.py --no-cpu-baseline --no-clinic --no-extras --workload $W 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W fold=$F', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us', d['steps'])"
  done
done
