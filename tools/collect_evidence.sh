#!/bin/bash
# Round evidence in one GPU-box call (run through gpurun from the repository root):  bash tools/collect_evidence.sh r02
#   1. the GPU test log, 2. the default bench line (with the CPU baseline legs) and one line per workload,
#   3. tools/collect_profiles.sh (rocprofv3 kernel stats / timelines of the bench commands, PMC traffic passes).
# Everything lands in gpurun_out/evidence_rNN/; copy what is to be judged into profiles/.
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/evidence_$R
mkdir -p $OUT
export RPO_VERBOSE=0
python3 tools/run_group.py 2700 python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -1 $OUT/pytest_gpu.log
python3 tools/run_group.py 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 tools/run_group.py 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-clinic --no-cpu-baseline > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
# the data-parallel code path over a ONE-rank RCCL group: all-reduce + rpo_absmax_slots inside the graph windows (intercept of 1 -> N)
for W in cart_sac cart_ddpg; do
  python3 tools/run_group.py 600 python3 bench.py --force-dist --workload $W --no-extras --no-clinic --no-cpu-baseline 2> $OUT/bench_force_dist_$W.err | grep '^{' > $OUT/bench_force_dist_$W.json
done
python3 tools/run_group.py 300 python3 tools/probe_project.py > $OUT/probe_project.txt 2>&1
for W in cart_ddpg cart_sac; do
  python3 tools/run_group.py 600 python3 tools/probe_large_batch.py $W 1048576 4096 2>&1 | grep -v "@1M\|@64K\|amdgpu.ids" > $OUT/probe_large_batch_$W.txt
done
for W in cart_sac pen_ddpg pen_sac evopf_ddpg evopf_sac; do
  python3 tools/run_group.py 900 python3 bench.py --workload $W > $OUT/bench_line_$W.json 2> $OUT/bench_line_$W.err
done
# bench.py's N > 1 control flow on this one GPU (gloo, two ranks sharing cuda:0): a control-flow check, not a scaling figure
RPO_BENCH_BACKEND=gloo python3 tools/run_group.py 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_gloo_2ranks.json 2> $OUT/bench_gloo_2ranks.err
# round 6: the one-launch rollout form by form at 65 536 / 2^20 lanes; the multi-GPU pre-flight in its one-GPU (gloo) form
for W in cart_ddpg cart_sac pen_sac; do
  python3 tools/run_group.py 300 python3 tools/probe/lanes_probe.py $W 65536 1048576 2>&1 | grep -v amdgpu.ids >> $OUT/probe_lanes.txt
done
python3 tools/run_group.py 600 python3 tools/scale_preflight.py --gpus 2 --backend gloo --lanes 512 --quick > $OUT/scale_preflight_gloo.json 2> $OUT/scale_preflight_gloo.err
bash tools/pmc_rollout_stream.sh > $OUT/pmc_rollout_stream.log 2>&1
cp gpurun_out/pmc_rollout_stream.txt gpurun_out/rollout_stream_kernel_stats.txt $OUT/ 2>/dev/null
# the large-batch update's GEMM-shaped launches: event-timed rooflines, rocprofv3 kernel stats, SQ / GRBM counters
python3 tools/run_group.py 600 python3 tools/probe_mlp_large.py > $OUT/probe_mlp_large.txt 2>&1
( export TMPDIR=/tmp; ROOT=$PWD; cd /tmp; rm -rf /tmp/p_lb
  rocprofv3 --kernel-trace --stats -d /tmp/p_lb -o t -- python3 $ROOT/tools/probe_large_batch.py cart_ddpg 1048576 4096 > $OUT/large_batch_cart_ddpg_profiled.txt 2>&1
  DB=$(ls /tmp/p_lb/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python3 $ROOT/tools/rocpd_summary.py $DB 30 > $OUT/large_batch_cart_ddpg_kernel_stats.txt )
bash tools/pmc_mlp_large.sh > $OUT/pmc_mlp_large.log 2>&1
cp gpurun_out/pmc_mlp_large.txt gpurun_out/mlp_large_kernel_stats.txt $OUT/ 2>/dev/null
# what a captured branch / a dependent layer launch costs, and the two kinds of EVOPF iteration
bash tools/collect_probes.sh $OUT
bash tools/collect_profiles.sh $R > $OUT/collect_profiles.log 2>&1
cp -r gpurun_out/prof_$R/* $OUT/ 2>/dev/null
ls $OUT | head -80
