#!/bin/bash
# Copy what is to be judged from gpurun_out/evidence_rNN/ (tools/collect_evidence.sh) into profiles/ under round-prefixed names.
set -u
R=${1:-r06}
E=gpurun_out/evidence_$R
P=profiles
for w in cart_ddpg cart_sac pen_ddpg pen_sac evopf_ddpg evopf_sac; do
  cp $E/${w}_kernel_stats.txt $P/${R}_bench_${w}_kernel_stats.txt
  cp $E/bench_$w.json $P/${R}_bench_${w}_profiled.json
  cp $E/${w}_timeline.txt $P/${R}_timeline_$w.txt
  cp $E/pmc_traffic_$w.json $P/${R}_pmc_traffic_$w.json
done
cp $E/pmc_traffic.json $P/${R}_pmc_traffic.json
cp $E/pmc_evopf_sq.txt $P/${R}_pmc_evopf_sq.txt
for c in FETCH_SIZE WRITE_SIZE; do for p in step step65k iter ride; do cp $E/pmc_${c}_$p.txt $P/${R}_pmc_${c}_$p.txt; done; done
cp $E/bench.json $P/${R}_bench.json
cp $E/bench_steps20.json $P/${R}_bench_steps20.json
for w in cart_sac pen_ddpg pen_sac evopf_ddpg evopf_sac; do cp $E/bench_line_$w.json $P/${R}_bench_$w.json; done
for w in cart_ddpg cart_sac; do cp $E/bench_force_dist_$w.json $P/${R}_bench_force_dist_$w.json; cp $E/probe_large_batch_$w.txt $P/${R}_probe_large_batch_$w.txt; done
cp $E/pytest_gpu.log $P/${R}_pytest_gpu.log
cp $E/probe_project.txt $P/${R}_probe_project.txt
# (the in-suite statistical samples are sanity samples since round 5: the resolved evidence is profiles/r05_stat_rows_* and
#  profiles/r05_cadence_*, collected by tools/statistical_parity.py / tools/cadence_learning.py)
for f in gpurun_out/statistical_parity_*.json; do cp $f $P/${R}_suite_$(basename $f) 2>/dev/null; done
cp $E/bench_gloo_2ranks.json $P/${R}_bench_gloo_2ranks.json
cp $E/probe_mlp_large.txt $P/${R}_probe_mlp_large.txt
cp $E/large_batch_cart_ddpg_kernel_stats.txt $P/${R}_large_batch_cart_ddpg_kernel_stats.txt
cp $E/pmc_mlp_large.txt $P/${R}_pmc_mlp_large.txt
cp $E/mlp_large_kernel_stats.txt $P/${R}_mlp_large_kernel_stats.txt
for f in probe_branch probe_branch3 probe_gemm_launch probe_evopf_period; do cp $E/$f.txt $P/${R}_$f.txt; done
ls $P | grep -c "^$R"
# round 6
cp $E/probe_lanes.txt $P/${R}_probe_lanes.txt
cp $E/scale_preflight_gloo.json $P/${R}_scale_preflight_gloo.json
cp $E/pmc_rollout_stream.txt $P/${R}_pmc_rollout_stream.txt
cp $E/rollout_stream_kernel_stats.txt $P/${R}_rollout_stream_kernel_stats.txt
ls $P | grep -c "^$R"
