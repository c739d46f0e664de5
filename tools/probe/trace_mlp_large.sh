#!/bin/bash
# per-kernel durations of the large-batch MLP launches (rocprofv3 kernel trace of tools/probe_mlp_large.py <n> once)
export TMPDIR=/tmp RPO_VERBOSE=0
ROOT=$PWD
cd /tmp; rm -rf /tmp/m_trace
rocprofv3 --kernel-trace --stats -d /tmp/m_trace -o t -- python3 $ROOT/tools/probe_mlp_large.py ${1:-1048576} once > /dev/null 2> /tmp/m_trace.err
DB=$(ls /tmp/m_trace/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_summary.py $DB 30 | cut -c1-90,100-170 > $ROOT/gpurun_out/mlp_large_kernel_stats.txt
cat $ROOT/gpurun_out/mlp_large_kernel_stats.txt
