#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy figures of the product build's gfx950 code (the flags of photonbend_amd/build.py), read from
the compiler's own assembly listing.
    python experiments/r6/isa_stats.py [-DNAME ...] [--out file.s] [--all]     (default: the hot kernels only)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from photonbend_amd.build import HIPCC_FLAGS, sources  # noqa: E402

defs = [a for a in sys.argv[1:] if a.startswith("-D")]
out = "/tmp/pb_isa.s"
if "--out" in sys.argv:
    out = sys.argv[sys.argv.index("--out") + 1]
flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC", "-fvisibility=hidden")]
if "--reuse" not in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, *defs, "-S", "--cuda-device-only", "-o", out, *sources()], stderr=subprocess.DEVNULL)
t = open(out).read()
hot = ("pb_hot_win_kernel", "pb_hot_double_kernel", "pb_bilinear_hot_kernel", "pb_bilinear_double_hot_kernel", "pb_certify_kernel")
rows = []
for m in re.finditer(r"^(_Z\w+):\s*; @", t, re.M):
    name = m.group(1)
    end = t.find(".Lfunc_end", m.end())
    body = t[m.end():end]
    tail = t[end:end + 6000]
    def grab(pat):
        mm = re.search(pat, tail)
        return int(mm.group(1)) if mm else -1
    ops = [l.split()[0] for l in body.splitlines() if l.strip() and l.strip()[0] not in ".;" and not l.strip().endswith(":")]
    rows.append((name, grab(r"; NumVgprs: (\d+)"), grab(r"; NumAgprs: (\d+)"), grab(r"; NumSgprs: (\d+)"), grab(r"; ScratchSize: (\d+)"), grab(r"; Occupancy: (\d+)"),
                 sum(1 for o in ops if o.startswith("v_") and "f64" in o), sum(1 for o in ops if o.startswith("v_")), len(ops)))
demangle = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} vgpr agpr sgpr scratch occ  f64  valu  instr")
for r, d in zip(rows, demangle):
    if "--all" not in sys.argv and not any(h in d for h in hot):
        continue
    d = re.sub(r"\(.*", "", d)[:70]
    print(f"{d:70s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:7d} {r[5]:3d} {r[6]:4d} {r[7]:5d} {r[8]:6d}")
