for S in 15 30 50; do
  for W in pen_ddpg pen_sac; do
  echo "== split $S $W"; RPO_RIDE_SPLIT=$S timeout 300 python bench.py --no-cpu-baseline --no-clinic --no-extras --workload $W --steps 4000 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
