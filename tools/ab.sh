# A/B of two builds of the library on one box (through gpurun): build the old tree, cp librpo_hip.so rpo_amd/csrc/librpo_old.so.bak,
# build the new one, cp it to librpo_new.so.bak, then:  bash tools/ab.sh "<workloads>" "<pytest args>"
WL=${1:-cart_ddpg cart_sac}
cd rpo_amd/csrc
for round in 1 2; do
for V in old new; do
  cp librpo_$V.so.bak librpo_hip.so
  cd ../..
  for W in $WL; do
  echo "== $V $W"; timeout 300 python bench.py --no-cpu-baseline --no-clinic --no-extras --workload $W --steps 4000 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('utd_matched_us_per_update'))"
  done
  cd rpo_amd/csrc
done
done
cp librpo_new.so.bak librpo_hip.so
cd ../..
[ -n "${2:-}" ] && timeout 900 python -m pytest $2 -q -x 2>&1 | tail -3
