#!/bin/bash
# Round-6 A/B in one GPU-box call (build the variants first: bash tools/probe/build_flag_variant.sh noslp -fno-slp-vectorize;
# bash tools/probe/build_flag_variant.sh noslpf -fno-slp-vectorize -- fused.hip): (1) -fno-slp-vectorize (whole library / fused.hip only) against the default build on the
# streaming kernels and the headline, (2) the EVOPF projection placement switch.  Output: gpurun_out/ab_r06/*.txt
set -u
OUT=$PWD/gpurun_out/ab_r06; mkdir -p $OUT
L=$PWD/rpo_amd/csrc
export RPO_VERBOSE=0
for round in 1 2; do
  for V in "" noslp noslpf; do
    LIB=$L/librpo_hip${V:+_$V}.so
    [ -f $LIB ] || continue
    RPO_HIP_LIBRARY=$LIB python tools/run_group.py 300 python tools/probe/lanes_probe.py cart_ddpg 65536 1048576 2>&1 | grep -v amdgpu.ids >> $OUT/lanes.txt
    RPO_HIP_LIBRARY=$LIB python tools/run_group.py 300 python tools/probe/lanes_probe.py cart_sac 1048576 2>&1 | grep -v amdgpu.ids >> $OUT/lanes.txt
    [ "$V" = noslpf ] && continue
    echo "== ${V:-default}" >> $OUT/mlp_large.txt
    RPO_HIP_LIBRARY=$LIB python tools/run_group.py 300 python tools/probe_mlp_large.py 2>&1 | grep -v amdgpu.ids >> $OUT/mlp_large.txt
    echo "== ${V:-default}" >> $OUT/headline.txt
    RPO_HIP_LIBRARY=$LIB python tools/run_group.py 300 python bench.py --no-cpu-baseline --no-clinic --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $OUT/headline.txt
  done
done
python tools/run_group.py 900 python tools/ab_tuning.py evopf_ddpg evopf_place 0 1 2 > $OUT/evopf_place.txt 2>&1
cat $OUT/lanes.txt $OUT/mlp_large.txt $OUT/headline.txt $OUT/evopf_place.txt
