"""Scratch builds with the slot-class skip hook (experiments/r6/skip_hook.txt) patched into a COPY of csrc/: build/libpb_r6_skip<bits>.so.
    python experiments/r6/build_skip.py 1 2 ..."""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from photonbend_amd.build import HIPCC_FLAGS
tmp = "/tmp/csrc_skip"
shutil.rmtree(tmp, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "photonbend_amd", "csrc"), os.path.join(tmp, "photonbend_amd", "csrc"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
p = os.path.join(tmp, "photonbend_amd", "csrc", "pb_kernels_bilinear.hpp")
s = open(p).read()
old = "    if ((flags & PB_TILE_SKIP) || (entry.bil_off >= 0 && !bil_xy)) {\n        if (pair) {"
hook = open(os.path.join(ROOT, "experiments", "r6", "skip_hook.txt")).read()
hook = hook[hook.index("#ifdef PB_R6_SKIP"):]
assert s.count(old) == 1
s = s.replace(old, "    bool leave = (flags & PB_TILE_SKIP) || (entry.bil_off >= 0 && !bil_xy);\n" + hook + "    if (leave) {\n        if (pair) {")
open(p, "w").write(s)
s = s.replace("#define PB_BIL_WPE_DBL 4", "#ifndef PB_BIL_WPE_DBL\n#define PB_BIL_WPE_DBL 4\n#endif")
open(p, "w").write(s)
def build(v):  # "<skip bits>" or "<skip bits>w<waves per SIMD of the double kernel>"
    bits, _, wpe = v.partition("w")
    out = os.path.join(ROOT, "build", f"libpb_r6_skip{v}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *HIPCC_FLAGS, f"-DPB_R6_SKIP={bits}", *([f"-DPB_BIL_WPE_DBL={wpe}"] if wpe else []),
                           os.path.join(tmp, "photonbend_amd", "csrc", "photonbend_hip.hip"), "-o", out], stderr=subprocess.DEVNULL)
    return out
with ThreadPoolExecutor(6) as ex:
    print(list(ex.map(build, sys.argv[1:])))
