#!/bin/bash
# SQ / GRBM counters and kernel-trace durations of the streaming rollout launch (rollout_stream.hip) and the other large-lane
# launches of tools/kernel_probe.py step (4096 and 2^20 lanes) / step65k, one small counter group per pass (--pmc with
# --kernel-trace only).  Output: gpurun_out/pmc_rollout_stream.txt, gpurun_out/rollout_stream_kernel_stats.txt
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_rollout_stream.txt
export TMPDIR=/tmp RPO_VERBOSE=0
: > $OUT
cd /tmp
for P in step step65k; do
  echo "# tools/kernel_probe.py $P" >> $OUT
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    T=${P}_$(echo $G | tr ' ' '_')
    rm -rf /tmp/rs_$T
    rocprofv3 --pmc $G --kernel-trace -d /tmp/rs_$T -o t -- python3 $ROOT/tools/kernel_probe.py $P > /dev/null 2> /tmp/rs_$T.err
    DB=$(ls /tmp/rs_$T/*results.db 2>/dev/null | head -1)
    if [ -n "$DB" ]; then python3 $ROOT/tools/rocpd_pmc.py $DB | grep -i "rollout_stream\|cartsafe_step\|replay_sample\|#" >> $OUT; else echo "# no db for $G" >> $OUT; tail -3 /tmp/rs_$T.err >> $OUT; fi
  done
done
rm -rf /tmp/rs_trace
rocprofv3 --kernel-trace --stats -d /tmp/rs_trace -o t -- python3 $ROOT/tools/kernel_probe.py step > /dev/null 2> /tmp/rs_trace.err
DB=$(ls /tmp/rs_trace/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_summary.py $DB 30 > $ROOT/gpurun_out/rollout_stream_kernel_stats.txt
cat $OUT
