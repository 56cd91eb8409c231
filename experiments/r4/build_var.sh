#!/bin/bash
# builds experiments/r4/libpb_v_<wpe>_<budget>.so: the bilinear kernels compiled for <wpe> waves per SIMD with the mode's window budget <budget>
cd "$(dirname "$0")/../.."
for v in "$@"; do
  wpe=${v%%_*}; bud=${v##*_}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared -fvisibility=hidden -DPB_BIL_WPE=$wpe -DPB_BIL_WIN_BUDGET=$bud photonbend_amd/csrc/photonbend_hip.hip -o experiments/r4/libpb_v_${wpe}_${bud}.so &
done
wait; ls experiments/r4/*.so
