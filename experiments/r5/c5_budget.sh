cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for b in 7168 12288; do
  l=$(timeout -k 10 120 python3 bench.py --config c5 --sampling bilinear --budget $b --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | tail -1)
  echo "c5 bilinear, nearest budget $b: $(echo "$l" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_frame"])')"
done
done
