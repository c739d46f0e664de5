export RPO_VERBOSE=0
for X in "-DRPO_FOLD_STRIDE=16" "-DRPO_FOLD_STRIDE=16 -DRPO_FOLD_NOWAIT"; do
  touch rpo_amd/csrc/nsplit.hip; HIPCC_EXTRA="$X" python rpo_amd/csrc/build.py > /dev/null 2>&1
  for F in 1 0; do
    RPO_FOLD_ADAM=$F python bench.py --no-cpu-baseline --no-clinic --no-extras --workload cart_ddpg 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$X fold=$F', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us', d['steps'])"
  done
done
