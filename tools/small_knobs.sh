#!/bin/bash
# Timing-only builds of the small-lattice kernel (csrc/flow_small.hip FT_KNOB bits) -> experiments/small_k<N>.so.
#   here:            bash tools/small_knobs.sh build 1 2 4 8 16 31
#   on the GPU box:  bash tools/small_knobs.sh run 1 2 4 8 16 31     (profile of one config-2 trajectory per build)
cd "$(dirname "$0")/.." || exit 1
mode=$1; shift
if [ "$mode" = build ]; then
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-load-store-vectorizer=0"
  for k in "$@"; do
    /opt/rocm/bin/hipcc $FL -DFT_KNOB=$k -c fthmc_amd/csrc/flow_small.hip -o /tmp/flow_small_k$k.o 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o experiments/small_k$k.so $(ls fthmc_amd/csrc/*.o | grep -v flow_small.o) /tmp/flow_small_k$k.o || exit 1
    echo built experiments/small_k$k.so
  done
else
  python3 tools/small_profile.py
  for k in "$@"; do echo "== FT_KNOB=$k"; FTHMC_LIB=$PWD/experiments/small_k$k.so python3 tools/small_profile.py; done
fi
