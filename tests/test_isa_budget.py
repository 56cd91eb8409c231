"""The hot kernels' register budgets, read from the compiler's own listing of the product build (hipcc cross-compiles gfx950 without a GPU).
Round 6 found that the bilinear tile kernels sit at the edge of the scalar register file (a 64-dword tile entry + the kernel arguments):
whether the compiler keeps the entry in SGPRs or drags it through VGPR lanes (v_writelane / v_readlane at every use) changed with the
spelling of an unrelated `if` - +40 % vector instructions per wave, c5 bilinear 80 -> 91 us, same pixels, every test green.  Nothing but
the listing shows it, so the listing is pinned here: VGPRs (occupancy), scratch, float64 in the float32 kernels, and the lane traffic."""

import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stats(tmp_path_factory):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("needs hipcc")
    spec = importlib.util.spec_from_file_location("isa_stats", os.path.join(ROOT, "experiments", "r6", "isa_stats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.kernel_stats(out=str(tmp_path_factory.mktemp("isa") / "pb.s"))
    return {r["name"]: r for r in rows}


def _pick(stats, prefix):
    got = {k: v for k, v in stats.items() if k.startswith(prefix)}
    assert got, prefix
    return got


def test_no_frame_loop_kernel_spills_to_scratch(stats):
    """every kernel a frame can run through, and certification (the dominant cost of plan creation)"""
    per_frame = ("pb_hot_", "pb_bilinear_hot_kernel", "pb_bilinear_double_hot_kernel", "pb_remap_kernel", "pb_sep_double_kernel", "pb_fix_kernel", "pb_certify_kernel",
                 "pb_sample_map", "pb_coordmap", "pb_rotate", "pb_index_kernel", "pb_gather")
    seen = {k: v["scratch"] for k, v in stats.items() if k.startswith(per_frame)}
    assert len(seen) >= 30, sorted(seen)
    bad = {k: v for k, v in seen.items() if v != 0}
    assert not bad, bad


def test_nearest_hot_kernel_budget(stats):
    for name, r in _pick(stats, "pb_hot_win_kernel").items():
        assert r["vgpr"] <= 64 and r["f64"] == 0 and r["lane_traffic"] == 0 and r["occupancy"] >= 7, (name, r)


def test_bilinear_tile_kernels_budget(stats):
    """<= 128 VGPRs (four waves per SIMD), the tile entry stays in scalar registers (a handful of lane moves per tile path is the compiler's
    normal bookkeeping; the spilled entry was 1 300-1 500), no float64 in the single-source kernel."""
    for prefix in ("pb_bilinear_hot_kernel", "pb_bilinear_double_hot_kernel"):
        for name, r in _pick(stats, prefix).items():
            assert r["vgpr"] <= 128 and r["occupancy"] >= 4, (name, r)
            assert r["lane_traffic"] <= 200, f"{name}: {r['lane_traffic']} v_readlane / v_writelane - the tile entry is being spilled through VGPR lanes"
    for name, r in _pick(stats, "pb_bilinear_hot_kernel").items():
        assert r["f64"] == 0, (name, r)


def test_single_frame_double_kernel_budget(stats):
    for name, r in _pick(stats, "pb_hot_double_kernel").items():
        if ", true, " in name:  # the single-frame instantiations (what pb_remap_u8 launches frame by frame)
            assert r["vgpr"] <= 136 and r["lane_traffic"] <= 200, (name, r)
