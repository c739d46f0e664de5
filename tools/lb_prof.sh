export TMPDIR=/tmp RPO_VERBOSE=0
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/lb
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_lb -o t -- python3 $ROOT/tools/probe_large_batch.py cart_sac 1048576 4096 > $ROOT/gpurun_out/lb/probe.txt 2>&1
DB=$(ls /tmp/p_lb/*results.db 2>/dev/null | head -1)
python3 $ROOT/tools/rocpd_summary.py $DB > $ROOT/gpurun_out/lb/sac_kernel_stats.txt 2>/dev/null
head -30 $ROOT/gpurun_out/lb/sac_kernel_stats.txt
python3 $ROOT/tools/rocpd_timeline.py $DB rollout_kernel 300 > $ROOT/gpurun_out/lb/sac_timeline.txt
head -60 $ROOT/gpurun_out/lb/sac_timeline.txt
