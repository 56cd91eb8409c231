"""bench.py's host-path block alone (NumPy in -> NumPy out, c2), three times.   python experiments/r6/host_path_only.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
torch.cuda.set_device(0)
cfg = bench.CONFIGS["c2"]
d, rots, s = bench.build_projs(cfg)
for _ in range(3):
    r = bench.host_path(cfg, d, rots, s)
    print(json.dumps({k: v for k, v in r.items() if k.startswith("ms_") and not k.endswith("note") or k in ("h2d_ms", "d2h_ms")}), flush=True)
