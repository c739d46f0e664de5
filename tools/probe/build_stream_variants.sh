#!/bin/bash
# Timing-only variants of the streaming forward (RPO_STREAM_SKIP bits, csrc/mlp_stream.h), the streaming backward (KIND=bwd,
# RPO_BWDS_SKIP), the layer-by-layer launches (KIND=gemm, RPO_GEMM_SKIP, csrc/mlp_gemm.h) or the one-launch rollout (KIND=rollout,
# RPO_ROLLOUT_SKIP, csrc/fused.hip): rpo_amd/csrc/librpo_hip_skipN.so
set -e
cd $(dirname $0)/../../rpo_amd/csrc
for N in "$@"; do
  GEMM_SKIP=0
  if [ "${KIND:-fwd}" = rollout ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DRPO_ROLLOUT_SKIP=${ROLL:-$N} -DRPO_TILE_SKIP=${TILE:-0} -c fused.hip -o /tmp/fused_skip$N.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librpo_hip_skip$N.so cartsafe.o pendulum.o evopf.o replay.o train_ops.o mlp.o /tmp/fused_skip$N.o nsplit.o rollout_stream.o mlp_bwd_stream.o
    continue
  fi
  if [ "${KIND:-fwd}" = bwd ]; then BWDS_SKIP=$N; STREAM_SKIP=0; elif [ "${KIND:-fwd}" = gemm ]; then GEMM_SKIP=$N; BWDS_SKIP=0; STREAM_SKIP=0; else STREAM_SKIP=$N; BWDS_SKIP=0; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DRPO_STREAM_SKIP=${STREAM_SKIP:-0} -DRPO_BWDS_SKIP=${BWDS_SKIP:-0} -DRPO_GEMM_SKIP=$GEMM_SKIP -c mlp.hip -o /tmp/mlp_skip$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librpo_hip_skip$N.so cartsafe.o pendulum.o evopf.o replay.o train_ops.o /tmp/mlp_skip$N.o fused.o nsplit.o rollout_stream.o mlp_bwd_stream.o
done
ls -la librpo_hip_skip*.so
