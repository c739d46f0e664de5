set -u
OUT=$PWD/gpurun_out/s2
mkdir -p $OUT
export TMPDIR=/tmp RPO_VERBOSE=0
ROOT=$PWD
cd /tmp
for W in cart_ddpg cart_sac; do
  rm -rf /tmp/p_$W
  rocprofv3 --kernel-trace --stats -d /tmp/p_$W -o t -- python3 $ROOT/bench.py --no-cpu-baseline --no-clinic --no-extras \
      --workload $W --steps 2000 --warmup 200 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  DB=$(ls /tmp/p_$W/*results.db 2>/dev/null | head -1)
  python3 $ROOT/tools/rocpd_timeline.py $DB rollout_kernel 200 > $OUT/${W}_timeline.txt
done
head -24 $OUT/cart_ddpg_timeline.txt; head -24 $OUT/cart_sac_timeline.txt
