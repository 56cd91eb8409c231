"""bench.py's ONE line must carry every measured figure in under 6 KB (the driver keeps the parsed core of the line and a
few KB of stdout tail - round 5's 14 KB line lost the nearest-mode configs and the un-amortised figures).  The verbose record of a real
run (profiles/r06_bench_detail.json, written by bench.py --detail on an MI355X) goes through compact_line here."""

import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_holds_every_figure_under_the_limit():
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_LIMIT == 6144, len(text)
    # the contract's keys
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["config"]["workload"].startswith("c2:") and "model" not in line["config"]
    # every config of the block, nearest AND bilinear, and the un-amortised figures of the headline
    assert set(line["configs"]) == {"c1", "c3", "c5", "c4shard", "c5shard", "c1_bilinear", "c2_bilinear", "c3_bilinear", "c5_bilinear"}
    assert all(isinstance(v[0], float) and v[0] > 0 for v in line["configs"].values())
    for k in ("single_image_ms", "faithful_kernel_ms", "plan_create_warm_ms"):
        assert isinstance(line[k], float), k
    assert set(line["roofline"]["frac_unamortised"]) >= {"single_image", "faithful_kernel"}
    assert line["scattered_batch"]["u8v_us_per_frame"] > 0
    assert isinstance(line["flavours"]["svml"], list) and isinstance(line["flavours"]["libm"], list)
    # numbers only: no prose beyond the workload name and the CPU sample
    longest = max((len(v) for v in _strings(line)), default=0)
    assert longest <= 120, longest


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)
