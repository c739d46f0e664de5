#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repository root):
#   kernel-trace + stats of the bench command (default workload and two others), the launch timeline of the graph
#   windows, and HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate passes) of eager iterations.
# Everything lands in gpurun_out/prof_rNN/ as text; copy what is to be judged into profiles/.
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp RPO_VERBOSE=0
ROOT=$PWD
cd /tmp
for W in cart_ddpg cart_sac pen_sac pen_ddpg evopf_ddpg evopf_sac; do
  rm -rf /tmp/p_$W
  rocprofv3 --kernel-trace --stats -d /tmp/p_$W -o t -- python3 $ROOT/bench.py --no-cpu-baseline --no-clinic --no-extras \
      --workload $W --steps 2000 --warmup 200 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  DB=$(ls /tmp/p_$W/*results.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    python3 $ROOT/tools/rocpd_summary.py $DB 40 > $OUT/${W}_kernel_stats.txt
    A=rollout_kernel; case $W in evopf_*) A=evopf_step_kernel;; esac
    python3 $ROOT/tools/rocpd_timeline.py $DB $A 200 > $OUT/${W}_timeline.txt
  fi
done
for C in FETCH_SIZE WRITE_SIZE; do
  for P in step step65k iter ride; do
    rm -rf /tmp/q_${C}_$P
    rocprofv3 --pmc $C --kernel-trace -d /tmp/q_${C}_$P -o t -- python3 $ROOT/tools/kernel_probe.py $P > /dev/null 2> $OUT/pmc_${C}_$P.err
    DB=$(ls /tmp/q_${C}_$P/*results.db 2>/dev/null | head -1)
    [ -n "$DB" ] && python3 $ROOT/tools/rocpd_pmc.py $DB > $OUT/pmc_${C}_$P.txt
  done
done
# HBM traffic of every workload's own launches (roofline.traffic of bench.py --workload W): policy_fre periods of the bench
# trainer, eagerly, one counter per pass
for W in cart_ddpg cart_sac pen_ddpg pen_sac evopf_ddpg evopf_sac; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/w_${C}_$W
    rocprofv3 --pmc $C --kernel-trace -d /tmp/w_${C}_$W -o t -- python3 $ROOT/tools/kernel_probe.py window:$W > /dev/null 2> $OUT/pmc_${C}_$W.err
    DB=$(ls /tmp/w_${C}_$W/*results.db 2>/dev/null | head -1)
    [ -n "$DB" ] && python3 $ROOT/tools/rocpd_pmc.py $DB > $OUT/pmc_${C}_window_$W.txt
  done
  python3 $ROOT/tools/pmc_to_json.py $OUT/pmc_traffic_$W.json --workload $W $OUT/pmc_FETCH_SIZE_window_$W.txt $OUT/pmc_WRITE_SIZE_window_$W.txt
done
# issue / wait counters of the EVOPF solver kernel (next to profiles/r01_pmc_evopf_sq.txt), one small group per pass
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
  T=$(echo $G | tr ' ' '_')
  rm -rf /tmp/s_$T
  rocprofv3 --pmc $G --kernel-trace -d /tmp/s_$T -o t -- python3 $ROOT/tools/kernel_probe.py window:evopf_ddpg > /dev/null 2> $OUT/pmc_sq_$T.err
  DB=$(ls /tmp/s_$T/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/rocpd_pmc.py $DB | grep evopf_act_project >> $OUT/pmc_evopf_sq.txt
done
cd $ROOT
python3 tools/pmc_to_json.py $OUT/pmc_traffic.json $OUT/pmc_FETCH_SIZE_step.txt $OUT/pmc_WRITE_SIZE_step.txt \
    $OUT/pmc_FETCH_SIZE_step65k.txt@65536 $OUT/pmc_WRITE_SIZE_step65k.txt@65536 \
    $OUT/pmc_FETCH_SIZE_iter.txt $OUT/pmc_WRITE_SIZE_iter.txt $OUT/pmc_FETCH_SIZE_ride.txt $OUT/pmc_WRITE_SIZE_ride.txt
ls -la $OUT
