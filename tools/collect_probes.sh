#!/bin/bash
# The schedule probes of DESIGN.md 4b (round 5) into one directory:  bash tools/collect_probes.sh gpurun_out/evidence_r05
OUT=${1:-gpurun_out/probes}
mkdir -p $OUT
python3 tools/run_group.py 300 python3 tools/probe/branch_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_branch.txt
python3 tools/run_group.py 300 python3 tools/probe/branch3_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_branch3.txt
python3 tools/run_group.py 300 python3 tools/probe/gemm_launch_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_gemm_launch.txt
python3 tools/run_group.py 300 python3 tools/probe/gemm_launch_probe.py mlp_gemm=0 2>&1 | grep -v amdgpu.ids >> $OUT/probe_gemm_launch.txt
for W in evopf_ddpg evopf_sac; do python3 tools/run_group.py 600 python3 tools/probe/evopf_period.py $W 2>&1 | grep -v amdgpu.ids; done > $OUT/probe_evopf_period.txt
cat $OUT/probe_evopf_period.txt
