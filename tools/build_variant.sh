#!/bin/bash
# A second build of the library with extra compiler flags, for A/B runs in one call on one device (tools/abn.sh):
#   bash tools/build_variant.sh NAME "-DFLAG=1 ..."   ->  experiments/lib_NAME.so   (objects under /tmp/fthmc_NAME)
ROOT=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; EXTRA=$2; D=/tmp/fthmc_$NAME; mkdir -p $D
LDSFLAGS="-mllvm -amdgpu-load-store-vectorizer=0"
PRELOAD=${PRELOAD--mllvm -amdgpu-kernarg-preload-count=16}       # PRELOAD= (empty) builds without kernel-argument preload
NOLICM=${NOLICM--mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-memory-clause}                         # NOLICM= (empty) builds flow_small with MachineLICM
SHA=$(python3 "$ROOT/tools/csrc_sha.py")
cd "$ROOT/fthmc_amd/csrc" || exit 1
for f in wilson flow flow_fwd flow_bwd_gather flow_bwd_train flow_wgrad flow_small flow_generic rng api; do
  fl=""; case $f in flow_fwd|flow_bwd_gather) fl="$LDSFLAGS $PRELOAD";; flow_wgrad|flow_bwd_train) fl="$LDSFLAGS";; flow_small) fl="$LDSFLAGS $NOLICM";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function $fl $EXTRA -DFTHMC_SRC_SHA=\"$SHA\" -c $f.hip -o $D/$f.o || exit 1 &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/experiments/lib_$NAME.so" $D/*.o && ls -la "$ROOT/experiments/lib_$NAME.so"
