#!/bin/bash
# A/B build of the whole library with extra compiler flags:  bash tools/probe/build_flag_variant.sh NAME [-flag ...] [-- file.hip ...]
#   -> rpo_amd/csrc/librpo_hip_NAME.so (objects under /tmp/rpo_variant_NAME/); select it with RPO_HIP_LIBRARY.
# Files listed after `--` get the extra flags, the others are compiled as usual (default: every file gets them).
set -e
NAME=$1; shift
FLAGS=(); FILES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
FILES=("$@")
cd $(dirname $0)/../../rpo_amd/csrc
OBJ=/tmp/rpo_variant_$NAME; mkdir -p $OBJ
ALL="cartsafe pendulum evopf replay train_ops mlp fused nsplit rollout_stream mlp_bwd_stream"
for f in $ALL; do
  EXTRA=("${FLAGS[@]}")
  if [ ${#FILES[@]} -gt 0 ]; then case " ${FILES[*]} " in *" $f.hip "*) ;; *) EXTRA=();; esac; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "${EXTRA[@]}" -c $f.hip -o $OBJ/$f.o &
  if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librpo_hip_$NAME.so $(for f in $ALL; do echo $OBJ/$f.o; done)
ls -la librpo_hip_$NAME.so
