#!/bin/bash
# the bilinear launch under each launch order (diagnostic build: PB_ORDER=1 plain walk / 2 rows outwards / 3 heaviest super-tiles first; 0 = the plan's own rule)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for o in ${ORDERS:-0 1 2 3}; do for c in "$@"; do
  l=$(PB_LIB_PATH=$R/build/libphotonbend_hip_diag.so PB_ORDER=$o timeout -k 10 120 python3 bench.py --config $c --sampling bilinear --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1)
  echo "order=$o $c $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')"
done; done; done
