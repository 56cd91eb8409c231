#!/bin/bash
# builds experiments/r4/libpb_f_<tag>.so with extra -D flags: build_f64.sh tag "-DA=1 -DB=2" ...
cd "$(dirname "$0")/../.."
while [ $# -gt 1 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared -fvisibility=hidden $defs photonbend_amd/csrc/photonbend_hip.hip -o experiments/r4/libpb_f_$tag.so -Rpass-analysis=kernel-resource-usage 2> /tmp/res_$tag.txt &
done
wait
for f in /tmp/res_*.txt; do echo "== $f"; python3 - "$f" <<'PY'
import re,sys
t=open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ',t)[1:]:
    name=b.split()[0]
    if any(k in name for k in ('pb_remap_kernel','pb_certify')):
        g=lambda k: re.search(k+r': (\d+)',b).group(1)
        print('  ',name[:40],'VGPR',g('VGPRs'),'occ',g(r'Occupancy \[waves/SIMD\]'),'scratch',g(r'ScratchSize \[bytes/lane\]'))
PY
done
