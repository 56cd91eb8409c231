#!/usr/bin/env python3
"""ISA reading aid: compile the library's device code with -DPB_MARKS and count the instructions between the "; PBMARK <name>" comments
of one kernel, per marked region (static counts of the straight-line path code: a per-tile figure, 16 pixels per lane).
    python experiments/r4/isa_count.py [kernel-name-substring]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/pb_marks.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DPB_MARKS", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, "photonbend_amd", "csrc", "photonbend_hip.hip")], stderr=subprocess.DEVNULL)
t = open(out).read()
names = [a for a in sys.argv[1:] if not a.startswith("-")]
want = names[0] if names else "pb_bilinear_hot_kernelILi0E"
m = re.search(r"^(_Z\w*" + re.escape(want) + r"\w*):", t, re.M)
k = t[m.start():t.index(".Lfunc_end", m.start())]
region, counts = "prologue", collections.OrderedDict()
for line in k.splitlines():
    line = line.strip()
    mm = re.match(r"; PBMARK (\w+)", line)
    if mm:
        region = mm.group(1) if mm.group(1) != "end" else "other"
        continue
    if not line or line[0] in ".;" or line.endswith(":"):
        continue
    op = line.split()[0]
    cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
    c = counts.setdefault(region, collections.Counter())
    c[cls] += 1
    c["op:" + op] += 1
print(m.group(1)[:70])
for r, c in counts.items():
    print(f"{r:18s} valu {c['valu']:5d}  salu {c['salu']:4d}  lds {c['lds']:4d}  vmem {c['vmem']:4d}")
    if "-v" in sys.argv:
        for op, n in sorted(((o, n) for o, n in c.items() if o.startswith("op:")), key=lambda x: -x[1])[:18]:
            print(f"      {op[3:]:28s} {n}")
