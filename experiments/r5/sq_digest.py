#!/usr/bin/env python3
"""Digest of the SQ-counter passes (gpurun_out/r5_sq/<tag>_p{1,2,3}.txt -> one block per kernel): per-wave instruction counts, VALU busy
(SQ_ACTIVE_INST_VALU / (8 x SQ_BUSY_CYCLES): 256 CUs over 32 shader engines' cycle counters), wait shares, LDS bank-conflict share.
usage: sq_digest.py <dir> <tag> [kernel-substring]"""
import re, sys, glob, os
d, tag = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else ""
vals = {}
for f in sorted(glob.glob(os.path.join(d, tag + "_p*.txt"))):
    k = None
    for line in open(f):
        if not line.startswith(" "):
            k = line.strip()
            if k == "--": k = None
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean=(\S+)", line)
        if m and k: vals.setdefault(k, {})[m.group(1)] = float(m.group(3))
for k, v in vals.items():
    if want not in k or "SQ_WAVES" not in v: continue
    w = v["SQ_WAVES"]; g = lambda n: v.get(n, float("nan"))
    cyc = g("SQ_BUSY_CYCLES") / 32.0
    print(f"{tag}: {k}")
    print(f"  waves {w:.0f}   kernel cycles (SQ_BUSY_CYCLES / 32 SEs) {cyc:.0f}   wave life {4 * g('SQ_WAVE_CYCLES') / w:.0f} cycles (SQ_WAVE_CYCLES counts quad-cycles; VALU issue = 4 cycles per instruction)")
    print(f"  per wave: VALU {g('SQ_INSTS_VALU') / w:.0f} (of which CVT {g('SQ_INSTS_VALU_CVT') / w:.0f})  SALU {g('SQ_INSTS_SALU') / w:.0f}  SMEM {g('SQ_INSTS_SMEM') / w:.1f}  "
          f"LDS {g('SQ_INSTS_LDS') / w:.0f}  VMEM rd {g('SQ_INSTS_VMEM_RD') / w:.1f} wr {g('SQ_INSTS_VMEM_WR') / w:.1f}")
    print(f"  VALU busy {g('SQ_ACTIVE_INST_VALU') / (8 * g('SQ_BUSY_CYCLES')):.3f}   any-instruction busy {g('SQ_ACTIVE_INST_ANY') / (8 * g('SQ_BUSY_CYCLES')):.3f}   "
          f"scalar busy {g('SQ_ACTIVE_INST_SCA') / (8 * g('SQ_BUSY_CYCLES')):.3f}   LDS busy {g('SQ_ACTIVE_INST_LDS') / (8 * g('SQ_BUSY_CYCLES')):.3f}")
    print(f"  of a wave's cycles: waiting for anything {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}   waiting to issue (SQ_WAIT_INST_ANY) {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}   "
          f"on LDS {g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES'):.3f}")
    print(f"  LDS: bank-conflict cycles / index-active cycles {g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.3f}   address conflicts {g('SQ_LDS_ADDR_CONFLICT'):.0f}   unaligned stalls {g('SQ_LDS_UNALIGNED_STALL'):.0f}")
