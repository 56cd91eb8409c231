#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy figures of the product build's gfx950 code (the flags of photonbend_amd/build.py), read from
the compiler's own assembly listing - and the SGPR-spill traffic (v_readlane / v_writelane), which decides whether a tile kernel keeps its
64-dword tile entry in scalar registers or drags it through VGPR lanes at every use (round 6: +40 % vector instructions per wave from the
spelling of one `if`; tests/test_isa_budget.py pins the figures).
    python experiments/r6/isa_stats.py [-DNAME ...] [--out file.s] [--all] [--reuse]     (default: the hot kernels only)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from photonbend_amd.build import HIPCC_FLAGS, sources  # noqa: E402

HOT = ("pb_hot_win_kernel", "pb_hot_double_kernel", "pb_bilinear_hot_kernel", "pb_bilinear_double_hot_kernel", "pb_certify_kernel")


def kernel_stats(defs=(), out="/tmp/pb_isa.s", reuse=False):
    """-> [{name, vgpr, agpr, sgpr, scratch, occupancy, f64, valu, instr, lane_traffic}] for every kernel of the device code"""
    if not (reuse and os.path.exists(out)):
        flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC", "-fvisibility=hidden")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, *defs, "-S", "--cuda-device-only", "-o", out, *sources()], stderr=subprocess.DEVNULL)
    t = open(out).read()
    rows = []
    for m in re.finditer(r"^(_Z\w+):\s*; @", t, re.M):
        end = t.find(".Lfunc_end", m.end())
        body, tail = t[m.end():end], t[end:end + 6000]

        def grab(pat):
            mm = re.search(pat, tail)
            return int(mm.group(1)) if mm else -1

        ops = [l.split()[0] for l in body.splitlines() if l.strip() and l.strip()[0] not in ".;" and not l.strip().endswith(":")]
        rows.append({"mangled": m.group(1), "vgpr": grab(r"; NumVgprs: (\d+)"), "agpr": grab(r"; NumAgprs: (\d+)"), "sgpr": grab(r"; TotalNumSgprs: (\d+)"),
                     "scratch": grab(r"; ScratchSize: (\d+)"), "occupancy": grab(r"; Occupancy: (\d+)"),
                     "f64": sum(1 for o in ops if o.startswith("v_") and "f64" in o), "valu": sum(1 for o in ops if o.startswith("v_")), "instr": len(ops),
                     "lane_traffic": sum(1 for o in ops if o in ("v_readlane_b32", "v_writelane_b32"))})
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", d).replace("void ", "")
    return rows


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "/tmp/pb_isa.s"
    print(f"{'kernel':70s} vgpr agpr sgpr scratch occ  f64  valu  instr  v_readlane+v_writelane (SGPR spill traffic)")
    for r in kernel_stats(defs, out, "--reuse" in sys.argv):
        if "--all" in sys.argv or any(h in r["name"] for h in HOT):
            print(f"{r['name'][:70]:70s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['scratch']:7d} {r['occupancy']:3d} {r['f64']:4d} {r['valu']:5d} {r['instr']:6d} {r['lane_traffic']:6d}")
