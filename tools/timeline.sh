# Kernel timeline of the steady-state iterations of bench workloads (through gpurun):  bash tools/timeline.sh "<workloads>" [tag]
set -u
WL=${1:-cart_ddpg cart_sac}
TAG=${2:-s2}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp RPO_VERBOSE=0
ROOT=$PWD
cd /tmp
for W in $WL; do
  rm -rf /tmp/p_$W
  rocprofv3 --kernel-trace --stats -d /tmp/p_$W -o t -- python3 $ROOT/bench.py --no-cpu-baseline --no-clinic --no-extras \
      --workload $W --steps 2000 --warmup 200 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  DB=$(ls /tmp/p_$W/*results.db 2>/dev/null | head -1)
  A=rollout_kernel; case $W in evopf*) A=evopf_step_kernel;; esac
  python3 $ROOT/tools/rocpd_timeline.py $DB $A 200 > $OUT/${W}_timeline.txt
  python3 $ROOT/tools/rocpd_summary.py $DB > $OUT/${W}_kernel_stats.txt 2>/dev/null
  head -40 $OUT/${W}_timeline.txt
done
