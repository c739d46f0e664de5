timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_full_size_gpu.py -q -x -m gpu -k "fused_front or ridden or full_size or ride" 2>&1 | tail -4
for round in 1 2; do
  for W in pen_ddpg pen_sac; do
  echo "== $W"; timeout 300 python bench.py --no-cpu-baseline --no-clinic --no-extras --workload $W --steps 4000 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
