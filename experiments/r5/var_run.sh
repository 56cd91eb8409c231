#!/bin/bash
# bilinear kernel time per config under each variant build: var_run.sh <tag> <libs: "product name1 name2"> <configs...>
R=$GRAFT_REPO_ROOT; T=$1; LIBS=$2; shift 2; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for rep in $(seq 1 ${REPS:-2}); do
for lib in $LIBS; do
  for c in "$@"; do
    if [ $lib = product ]; then p=""; else p=$R/experiments/r5/libpb_$lib.so; fi
    l=$(PB_LIB_PATH=$p timeout -k 10 120 python3 bench.py --config $c --sampling ${SAMPLING:-bilinear} --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>>$O/err.log | tail -1)
    echo "$lib $c $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')" | tee -a $O/var.log
  done
done
done
