#!/bin/bash
# same-box A/B of the bilinear mode's speed-only features through the diagnostic build's PB_BIL_OFF knob: off_ab.sh "<off values>" <configs...>
R=$GRAFT_REPO_ROOT; OFFS=$1; shift; cd $R
for rep in 1 2; do for off in $OFFS; do for c in "$@"; do
  l=$(PB_LIB_PATH=$R/build/libphotonbend_hip_diag.so PB_BIL_OFF=$off timeout -k 10 120 python3 bench.py --config $c --sampling bilinear --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1)
  echo "off=$off $c $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')"
done; done; done
