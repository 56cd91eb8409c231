"""A scratch build with ONE `#define NAME value` of csrc/ replaced, from a COPY of the sources: build/libpb_<tag>.so.
    python experiments/r6/build_define.py tag NAME=value [NAME=value ...]"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from photonbend_amd.build import HIPCC_FLAGS
tag, defs = sys.argv[1], dict(a.split("=", 1) for a in sys.argv[2:])
tmp = f"/tmp/csrc_{tag}"
shutil.rmtree(tmp, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "photonbend_amd", "csrc"), os.path.join(tmp, "photonbend_amd", "csrc"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
for name, val in defs.items():
    hit = 0
    for f in os.listdir(os.path.join(tmp, "photonbend_amd", "csrc")):
        p = os.path.join(tmp, "photonbend_amd", "csrc", f)
        s = open(p).read()
        s2, n = re.subn(r"^#define " + re.escape(name) + r"\b[^\n]*", f"#define {name} {val}", s, flags=re.M)
        if n:
            open(p, "w").write(s2)
            hit += n
    assert hit == 1, (name, hit)
out = os.path.join(ROOT, "build", f"libpb_{tag}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", *HIPCC_FLAGS, os.path.join(tmp, "photonbend_amd", "csrc", "photonbend_hip.hip"), "-o", out], stderr=subprocess.DEVNULL)
print(out)
