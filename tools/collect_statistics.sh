#!/bin/bash
# The statistical evidence of a round in one GPU-box call (run through gpurun from the repository root, AFTER the kernel sources
# are final: every file carries the ABI version and a hash of rpo_amd/csrc, tools/provenance.py):
#   bash tools/collect_statistics.sh r06   ->  gpurun_out/statistics_r06/{stat_rows_*.npz,json, cadence_learning.json, cadence_lanes.json}
# then  for f in gpurun_out/statistics_r06/*; do cp $f profiles/r06_$(basename $f); done   (~25 GPU-minutes)
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out/statistics_$R
mkdir -p $OUT
export RPO_VERBOSE=0
# num_envs = 1, 3000 iterations per seed, as many GPU seeds as the reference has (cart-RPODDPG 1536, pendulum-RPODDPG 576 x 2,
# cart-RPOSAC 1152, pendulum-RPOSAC 192 x 2): tests/test_statistical_evidence.py
python3 tools/run_group.py 900 python3 tools/statistical_parity.py ddpg cart 1536 > $OUT/log_ddpg_cart.txt 2>&1
python3 tools/run_group.py 900 python3 tools/statistical_parity.py ddpg pendulum 1152 > $OUT/log_ddpg_pendulum.txt 2>&1
python3 tools/run_group.py 1500 python3 tools/statistical_parity.py sac cart 1152 > $OUT/log_sac_cart.txt 2>&1
python3 tools/run_group.py 900 python3 tools/statistical_parity.py sac pendulum 384 > $OUT/log_sac_pendulum.txt 2>&1
cp gpurun_out/stat_rows_*.npz gpurun_out/stat_rows_*.json $OUT/
# the vectorised cadences at matched updates: 128 seeds (batch 256 per vector step) + 32 (one 2^20-row batch); the lane sweep
python3 tools/run_group.py 2400 python3 tools/cadence_learning.py 128 3000 32 > $OUT/log_cadence.txt 2>&1
python3 tools/run_group.py 1500 python3 tools/cadence_learning.py lanes 1,16,256 128 > $OUT/log_cadence_lanes.txt 2>&1
cp gpurun_out/cadence_learning.json gpurun_out/cadence_lanes.json $OUT/
tail -3 $OUT/log_*.txt
ls -la $OUT
