#!/bin/bash
# builds experiments/r5/libpb_<name>.so for "name:-DFLAG1,-DFLAG2" specs (timing / A-B variants of the library; never the product)
cd "$(dirname "$0")/../.."
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared -fvisibility=hidden $flags photonbend_amd/csrc/photonbend_hip.hip -o experiments/r5/libpb_$name.so &
done
wait; ls -la experiments/r5/*.so
