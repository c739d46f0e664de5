#!/bin/bash
# per-kernel durations of the large-batch MLP launches (tools/probe_mlp_large.py, default arguments: several repetitions)
set -u
ROOT=$PWD
export TMPDIR=/tmp RPO_VERBOSE=0
cd /tmp
rm -rf /tmp/m_trace
rocprofv3 --kernel-trace --stats -d /tmp/m_trace -o t -- python3 $ROOT/tools/probe_mlp_large.py "$@" > /dev/null 2> /tmp/m_trace.err
DB=$(ls /tmp/m_trace/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python3 $ROOT/tools/rocpd_summary.py $DB 30 | cut -c1-100,104-170 | grep -i "bwd_\|splitk\|#"
