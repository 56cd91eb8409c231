#!/bin/bash
# builds experiments/r4/libpb_abl<N>.so for the given PB_BIL_ABL values (timing experiments: wrong pixels)
cd "$(dirname "$0")/../.."
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared -fvisibility=hidden -DPB_BIL_ABL=$n photonbend_amd/csrc/photonbend_hip.hip -o experiments/r4/libpb_abl$n.so &
done
wait; ls -la experiments/r4/*.so
