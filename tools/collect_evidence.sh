#!/bin/bash
# Round evidence in one GPU-box call (run through gpurun from the repository root):  bash tools/collect_evidence.sh r02
#   1. the GPU test log, 2. the default bench line (with the CPU baseline legs) and one line per workload,
#   3. tools/collect_profiles.sh (rocprofv3 kernel stats / timelines of the bench commands, PMC traffic passes).
# Everything lands in gpurun_out/evidence_rNN/; copy what is to be judged into profiles/.
set -u
R=${1:-r02}
OUT=$PWD/gpurun_out/evidence_$R
mkdir -p $OUT
export RPO_VERBOSE=0
timeout 1500 python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -1 $OUT/pytest_gpu.log
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
for W in cart_sac pen_ddpg pen_sac evopf_ddpg evopf_sac; do
  timeout 900 python3 bench.py --workload $W > $OUT/bench_line_$W.json 2> $OUT/bench_line_$W.err
done
bash tools/collect_profiles.sh $R > $OUT/collect_profiles.log 2>&1
cp -r gpurun_out/prof_$R/* $OUT/ 2>/dev/null
ls $OUT | head -80
