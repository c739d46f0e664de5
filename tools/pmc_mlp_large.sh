#!/bin/bash
# SQ / GRBM counters of the large-batch MLP launches (tools/probe_mlp_large.py <n> once), one small counter group per pass
# (--pmc with --kernel-trace only).  Output: gpurun_out/pmc_mlp_large.txt
set -u
N=${1:-1048576}
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_mlp_large.txt
export TMPDIR=/tmp RPO_VERBOSE=0
: > $OUT
cd /tmp
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  T=$(echo $G | tr ' ' '_')
  rm -rf /tmp/m_$T
  rocprofv3 --pmc $G --kernel-trace -d /tmp/m_$T -o t -- python3 $ROOT/tools/probe_mlp_large.py $N once > /dev/null 2> /tmp/m_$T.err
  DB=$(ls /tmp/m_$T/*results.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then python3 $ROOT/tools/rocpd_pmc.py $DB | grep -i "mlp_\|#" >> $OUT; else echo "# no db for $G" >> $OUT; tail -3 /tmp/m_$T.err >> $OUT; fi
done
# durations of the same launches from a kernel trace
rm -rf /tmp/m_trace
rocprofv3 --kernel-trace --stats -d /tmp/m_trace -o t -- python3 $ROOT/tools/probe_mlp_large.py $N once > /dev/null 2> /tmp/m_trace.err
DB=$(ls /tmp/m_trace/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_summary.py $DB 30 > $ROOT/gpurun_out/mlp_large_kernel_stats.txt
cat $OUT
