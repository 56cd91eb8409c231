#!/usr/bin/env python3
"""Generates pb_math_libm.hpp: glibc 2.35's double-precision asin, acos, atan and tan - the functions NumPy's np.arcsin / np.arccos /
np.arctan / np.tan reach on an x86-64 host WITHOUT AVX512_SKX (there NumPy has no SIMD kernel for them and calls libm; the ifunc picks the
`_fma` builds on every CPU with FMA + AVX2) - as straight C, instruction by instruction.

Why generated: the reference's bits on such a host are these functions' bits (tests/golden/npmath_libm.npz), they are not correctly
rounded (0.04-0.25 % of their results differ from the correctly rounded value), and the sources are not in this image.  Round 4 restated
sin / cos / sincos / atan2 by hand from the machine code; this script does the same mechanically for the four remaining functions:
it disassembles `__atan_fma`, `__asin_fma`, `__acos_fma`, `__tan_fma` out of the installed libm.so.6 (objdump), and writes every
instruction as one C statement on the same values - the scalar double operations with their fused multiply-adds exactly where the build
has them, the integer index arithmetic, the branches as gotos - with the constants and lookup tables (asincos.tbl, uatan.tbl `cij`,
utan.tbl `xfg`) read out of the same file.  What is left out: the stack protector, the save / restore of the rounding mode (the device
rounds to nearest), and tan's huge-argument reduction (|x| >= 2^27 or so calls __branred: the generated function returns NaN there; no
lens argument comes near).

Licence: the output restates GNU C Library code (LGPL-2.1-or-later; IBM Accurate Mathematical Library) - see NOTICE.
Run on a glibc 2.35 x86-64 machine; the output is committed:
    python photonbend_amd/csrc/gen_libm_flavour.py > photonbend_amd/csrc/pb_math_libm.hpp"""
import ctypes
import re
import struct
import subprocess
import sys

FUNCS = [("atan", 0x76EE0, 0x772B0), ("asin", 0x772B0, 0x77960), ("acos", 0x77960, 0x78060), ("tan", 0x799D0, 0x7A250)]
G64 = ["rax", "rbx", "rcx", "rdx", "rsi", "rdi", "rbp", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "r15"]
SUB = {}
for r in ("a", "b", "c", "d"):
    SUB["e%sx" % r] = ("r%sx" % r, 32, 0)
    SUB["%sx" % r] = ("r%sx" % r, 16, 0)
    SUB["%sl" % r] = ("r%sx" % r, 8, 0)
    SUB["%sh" % r] = ("r%sx" % r, 8, 8)
for r in ("si", "di", "bp"):
    SUB["e" + r] = ("r" + r, 32, 0)
    SUB[r + "l"] = ("r" + r, 8, 0)
    SUB[r] = ("r" + r, 16, 0)
for n in range(8, 16):
    SUB["r%dd" % n] = ("r%d" % n, 32, 0)
    SUB["r%db" % n] = ("r%d" % n, 8, 0)
for r in G64:
    SUB[r] = (r, 64, 0)


def libm_path():
    ctypes.CDLL("libm.so.6")
    return next(l.split()[-1] for l in open("/proc/self/maps") if "libm.so" in l)


class Elf:
    def __init__(self, path):
        self.blob = open(path, "rb").read()
        out = subprocess.run(["readelf", "-lW", path], capture_output=True, text=True, check=True).stdout
        self.loads = []
        for line in out.splitlines():
            f = line.split()
            if f and f[0] == "LOAD":
                self.loads.append((int(f[2], 16), int(f[1], 16), int(f[4], 16)))

    def read(self, vaddr, n):
        for va, off, size in self.loads:
            if va <= vaddr and vaddr + n <= va + size:
                return self.blob[off + vaddr - va : off + vaddr - va + n]
        raise KeyError(hex(vaddr))

    def u64(self, vaddr):
        return struct.unpack("<Q", self.read(vaddr, 8))[0]


def split_ops(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Gen:
    def __init__(self, elf, name, lo, hi, path):
        self.elf, self.name, self.lo, self.hi = elf, name, lo, hi
        txt = subprocess.run(["objdump", "-d", path, "--start-address=%#x" % lo, "--stop-address=%#x" % hi, "--no-show-raw-insn"],
                             capture_output=True, text=True, check=True).stdout
        self.ins = []
        for line in txt.splitlines():
            m = re.match(r"^\s+([0-9a-f]+):\t(\S+)\s*(.*)$", line)
            if not m:
                continue
            addr, mn, rest = int(m.group(1), 16), m.group(2), m.group(3)
            cmt = None
            if "#" in rest:
                rest, c = rest.split("#", 1)
                cmt = int(c.split()[0], 16)
            rest = re.sub(r"<[^>]*>", "", rest).strip()
            self.ins.append((addr, mn, split_ops(rest), cmt))
        self.targets = set()
        for addr, mn, ops, cmt in self.ins:
            if mn.startswith("j") and ops:
                self.targets.add(int(ops[0].split()[0], 16))
        self.tables = set()  # absolute addresses loaded by lea: table bases

    # ---- operands ------------------------------------------------------------------------------------------------
    def greg_read(self, name):
        base, w, sh = SUB[name]
        if w == 64:
            return base
        return "((%s >> %d) & %#xull)" % (base, sh, (1 << w) - 1) if sh else "(%s & %#xull)" % (base, (1 << w) - 1)

    def greg_write(self, name, expr):
        base, w, sh = SUB[name]
        if w == 64:
            return "%s = (unsigned long long)(%s);" % (base, expr)
        if w == 32:
            return "%s = (unsigned long long)(unsigned)(%s);" % (base, expr)  # a 32-bit write zero-extends
        mask = ((1 << w) - 1) << sh
        return "%s = (%s & ~%#xull) | ((((unsigned long long)(%s)) << %d) & %#xull);" % (base, base, mask, expr, sh, mask)

    def width(self, op):
        return SUB[op[1:]][1] if op.startswith("%") and op[1:] in SUB else None

    def mem_addr(self, op, cmt):
        """C expression of the effective address of a memory operand (None: stack / %fs)."""
        if "%fs:" in op:
            return None
        m = re.match(r"^(-?0x[0-9a-f]+|-?\d+)?\((%\w+)?(?:,(%\w+),(\d))?\)$", op)
        if not m:
            raise ValueError("operand? " + op)
        disp = int(m.group(1), 0) if m.group(1) else 0
        base, idx, sc = m.group(2), m.group(3), m.group(4)
        if base == "%rip":
            return "%#xull" % cmt
        if base == "%rsp":
            return ("stack", disp)
        parts = []
        if base:
            parts.append(self.greg_read(base[1:]))
        if idx:
            parts.append("%s * %sull" % (self.greg_read(idx[1:]), sc))
        if disp:
            parts.append("(unsigned long long)(long long)%d" % disp)
        return " + ".join(parts) if parts else "0ull"

    def const64(self, cmt):
        return "%#xull" % self.elf.u64(cmt)

    def xsrc(self, op, cmt):
        """bits of a scalar-double source operand"""
        if op.startswith("%xmm"):
            return "X[%d]" % int(op[4:])
        a = self.mem_addr(op, cmt)
        if isinstance(a, tuple):
            return "STK64(%d)" % a[1]
        if re.match(r"^0x[0-9a-f]+ull$", a):  # rip-relative constant
            return self.const64(cmt)
        return "pb_libm_ld_%s(%s)" % (self.name, a)

    def isrc(self, op, cmt, w=None):
        if op.startswith("$"):
            v = int(op[1:], 0)
            return "(long long)%#xull" % (v & 0xFFFFFFFFFFFFFFFF) if (v >= 1 << 63 or v < 0) else "%dll" % v
        if op.startswith("%"):
            return self.greg_read(op[1:])
        a = self.mem_addr(op, cmt)
        if a is None:
            return "0ull"  # the stack protector's canary
        if isinstance(a, tuple):
            return "STK%d(%d)" % (w or 64, a[1])
        if re.match(r"^0x[0-9a-f]+ull$", a):  # a constant (or a pointer nothing restated follows) loaded into an integer register
            try:
                v = self.elf.u64(cmt)
            except KeyError:
                v = 0
            return "%#xull" % (v & ((1 << (w or 64)) - 1))
        raise ValueError("integer load from " + op)

    # ---- one instruction -> C ----------------------------------------------------------------------------------------
    def emit(self, addr, mn, ops, cmt):
        D = lambda b: "pb_from_bits(%s)" % b
        B = lambda e: "pb_bits(%s)" % e
        xs = lambda i: self.xsrc(ops[i], cmt)
        xd = lambda: "X[%d]" % int(ops[-1][4:])
        if mn in ("endbr64", "nop", "nopl", "nopw", "xchg", "cs", "data16", "push", "pop", "vldmxcsr"):
            return ""
        if any("%fs:" in o for o in ops):
            if mn == "sub":  # canary check: equal
                return "FL_RES(0ll, 64);"
            return ""
        if any("%rsp" == o for o in ops) and mn in ("sub", "add"):
            return ""
        if mn == "vstmxcsr":
            a = self.mem_addr(ops[0], cmt)
            return "STK32W(%d, 0x1f80u);" % a[1]
        if mn == "call":
            if "e240" in ops[0]:
                return ""
            return "return PB_LIBM_DEFER;  /* the huge-argument reduction (__branred) is not restated */"
        if mn == "ret":
            return "return pb_from_bits(X[0]);"
        if mn == "jmp":
            return "goto L%x;" % int(ops[0].split()[0], 16)
        if mn.startswith("j"):
            return "if (CC_%s) goto L%x;" % (mn[1:].upper(), int(ops[0].split()[0], 16))
        if mn == "cmove":
            return "if (CC_E) { %s }" % self.greg_write(ops[1][1:], self.isrc(ops[0], cmt))
        # ---- scalar double ----
        if mn in ("vmovsd", "vmovq", "vmovapd", "vmovaps"):
            if len(ops) == 3:  # vmovsd %xmm1,%xmm2,%xmm3: low half from the first
                return "%s = %s;" % (xd(), xs(0))
            s, d = ops
            if d.startswith("%xmm"):
                if s.startswith("%") and not s.startswith("%xmm"):
                    return "%s = %s;" % (xd(), self.greg_read(s[1:]))
                return "%s = %s;" % (xd(), xs(0))
            if d.startswith("%"):
                return self.greg_write(d[1:], xs(0))
            a = self.mem_addr(d, cmt)
            return "STK64W(%d, %s);" % (a[1], xs(0))
        two = {"vaddsd": "+", "vsubsd": "-", "vmulsd": "*", "vdivsd": "/"}
        if mn in two:  # AT&T: op a, b, dst  ->  dst = b op a
            return "%s = %s;" % (xd(), B("%s %s %s" % (D(xs(1)), two[mn], D(xs(0)))))
        bit = {"vandpd": "&", "vxorpd": "^", "vorpd": "|"}
        if mn in bit:
            return "%s = %s %s %s;" % (xd(), xs(1), bit[mn], xs(0))
        if mn == "vsqrtsd":
            return "%s = %s;" % (xd(), B("sqrt(%s)" % D(xs(0))))
        m = re.match(r"^vf(n?)m(add|sub)(132|213|231)sd$", mn)
        if m:
            o3, o2, o1 = xs(0), xs(1), xd()
            a, b, c = {"132": (o1, o3, o2), "213": (o2, o1, o3), "231": (o2, o3, o1)}[m.group(3)]
            neg_a = "-" if m.group(1) else ""
            neg_c = "-" if m.group(2) == "sub" else ""
            return "%s = %s;" % (xd(), B("fma(%s%s, %s, %s%s)" % (neg_a, D(a), D(b), neg_c, D(c))))
        if mn in ("vcomisd", "vucomisd"):
            return "FL_FP(%s, %s);" % (D(xs(1)), D(xs(0)))
        if mn == "vcmpltsd":
            return "%s = (%s < %s) ? ~0ull : 0ull;" % (xd(), D(xs(1)), D(xs(0)))
        if mn == "vcmpnltsd":
            return "%s = (%s < %s) ? 0ull : ~0ull;" % (xd(), D(xs(1)), D(xs(0)))
        if mn == "vblendvpd":
            return "%s = (X[%d] >> 63) ? %s : %s;" % (xd(), int(ops[0][4:]), xs(1), xs(2))
        if mn in ("vcvttsd2si", "cvttsd2si"):
            w = self.width(ops[1])
            conv = "(long long)pb_libm_cvtt%d(%s)" % (w, D(xs(0)))
            return self.greg_write(ops[1][1:], conv)
        # ---- integer ----
        if mn in ("mov", "movl", "movq", "movabs"):
            s, d = ops
            if d.startswith("%"):
                return self.greg_write(d[1:], self.isrc(s, cmt, self.width(d)))
            a = self.mem_addr(d, cmt)
            if a is None:
                return ""
            w = self.width(s) or 32
            return "STK%dW(%d, %s);" % (w, a[1], self.isrc(s, cmt))
        if mn == "movslq":
            return self.greg_write(ops[1][1:], "(long long)(int)%s" % self.isrc(ops[0], cmt, 32))
        if mn == "cltq":
            return "rax = (unsigned long long)(long long)(int)rax;"
        if mn == "lea":
            a = self.mem_addr(ops[0], cmt)
            if isinstance(a, tuple):
                return self.greg_write(ops[1][1:], "0ull")  # (a stack address: only the argument of a call that is not restated)
            if re.match(r"^0x[0-9a-f]+ull$", a):
                self.tables.add(cmt)
            return self.greg_write(ops[1][1:], a)
        ari = {"add": "+", "sub": "-", "and": "&", "or": "|", "xor": "^", "imul": "*"}
        if mn in ari:
            if mn == "imul" and len(ops) == 3:
                w = self.width(ops[2])
                e = "(%s) * (%s)" % (self.isrc(ops[1], cmt, w), self.isrc(ops[0], cmt, w))
                d = ops[2]
            else:
                s, d = ops
                w = self.width(d)
                e = "(%s) %s (%s)" % (self.isrc(d, cmt, w), ari[mn], self.isrc(s, cmt, w))
            flag = "FL_SUB(%s, %s, %d); " % (self.isrc(d, cmt, w), self.isrc(ops[0], cmt, w), w) if mn == "sub" else ""
            st = self.greg_write(d[1:], e)
            if mn != "sub":
                st += " FL_RES(%s, %d);" % (self.isrc(d, cmt, w), w)
            return flag + st
        if mn in ("shl", "sar", "shr"):
            s, d = ops if len(ops) == 2 else ("$1", ops[0])
            w = self.width(d)
            n = int(s[1:], 0)
            v = self.isrc(d, cmt, w)
            if mn == "shl":
                e = "(%s) << %d" % (v, n)
            elif mn == "shr":
                e = "(%s) >> %d" % (v, n)
            else:
                e = "(unsigned long long)(SX(%s, %d) >> %d)" % (v, w, n)
            return self.greg_write(d[1:], e) + " FL_RES(%s, %d);" % (self.isrc(d, cmt, w), w)
        if mn == "cmp":
            s, d = ops
            w = self.width(d) or self.width(s) or 32
            return "FL_SUB(%s, %s, %d);" % (self.isrc(d, cmt, w), self.isrc(s, cmt, w), w)
        if mn == "test":
            s, d = ops
            w = self.width(d) or self.width(s) or 32
            return "FL_RES((%s) & (%s), %d);" % (self.isrc(d, cmt, w), self.isrc(s, cmt, w), w)
        raise ValueError("%x: %s %s" % (addr, mn, ops))

    def body(self):
        out = []
        for addr, mn, ops, cmt in self.ins:
            c = self.emit(addr, mn, ops, cmt)
            lab = "L%x: " % addr if addr in self.targets else ""
            if lab or c:
                out.append("    %s%s  /* %x: %s %s */" % (lab, c or ";", addr, mn, ", ".join(ops)))
        return out


PRELUDE = r'''// pb_math_libm.hpp - GENERATED by gen_libm_flavour.py from the machine code of glibc 2.35's libm.so.6 (__atan_fma, __asin_fma, __acos_fma,
// __tan_fma); do not edit.
// SPDX-License-Identifier: LGPL-2.1-or-later  (restates GNU C Library code: IBM Accurate Mathematical Library; see NOTICE)
//
// np.arcsin / np.arccos / np.arctan / np.tan as an x86-64 host WITHOUT AVX512_SKX computes them: there NumPy calls libm, and libm's ifunc
// picks these builds on every CPU with FMA and AVX2.  One C statement per instruction (the comment names it), same operation order, same
// fused multiply-adds, constants and tables read out of the same file; the second "math flavour" of the float64 chain (PB_MATH_LIBM,
// pb_plan_create_ex), pinned by tests/golden/npmath_libm.npz.  Not restated: tan's huge-argument reduction (NaN there; PB_LIBM_DEFER).
#pragma once
#define PB_LIBM_DEFER pb_from_bits(0x7ff8000000000000ull)
// the processor flags the compare / test / arithmetic instructions leave, as the conditional jumps read them
struct PbLibmFlags {
    int kind;  // 0: dst - src (integers), 1: a result (logic / shift / add), 2: an ordered / unordered double compare
    long long sa, sb;
    unsigned long long ua, ub;
    double fa, fb;
};
#define SX(v, w) ((long long)((unsigned long long)(v) << (64 - (w))) >> (64 - (w)))
#define FL_SUB(a, b, w) do { F.kind = 0; F.sa = SX(a, w); F.sb = SX(b, w); F.ua = (unsigned long long)(a) & (~0ull >> (64 - (w))); F.ub = (unsigned long long)(b) & (~0ull >> (64 - (w))); } while (0)
#define FL_RES(r, w) do { F.kind = 1; F.sa = SX(r, w); F.sb = 0; F.ua = (unsigned long long)(r) & (~0ull >> (64 - (w))); F.ub = 0; } while (0)
#define FL_FP(a, b) do { F.kind = 2; F.fa = (a); F.fb = (b); } while (0)
#define FP_UNORD (F.fa != F.fa || F.fb != F.fb)
#define CC_E (F.kind == 2 ? (FP_UNORD || F.fa == F.fb) : F.sa == F.sb)
#define CC_NE (!CC_E)
#define CC_G (F.sa > F.sb)
#define CC_GE (F.sa >= F.sb)
#define CC_L (F.sa < F.sb)
#define CC_LE (F.sa <= F.sb)
#define CC_S ((F.kind == 0 ? F.sa - F.sb : F.sa) < 0)
#define CC_A (F.kind == 2 ? (!FP_UNORD && F.fa > F.fb) : F.ua > F.ub)
#define CC_AE (F.kind == 2 ? (!FP_UNORD && F.fa >= F.fb) : F.ua >= F.ub)
#define CC_B (F.kind == 2 ? (FP_UNORD || F.fa < F.fb) : F.ua < F.ub)
#define CC_BE (F.kind == 2 ? (FP_UNORD || F.fa <= F.fb) : F.ua <= F.ub)
#define CC_P (F.kind == 2 && FP_UNORD)
#define STK64(o) stk[(o) >> 3]
#define STK64W(o, v) stk[(o) >> 3] = (v)
#define STK32(o) ((stk[(o) >> 3] >> (((o) & 4) * 8)) & 0xffffffffull)
#define STK32W(o, v) stk[(o) >> 3] = (stk[(o) >> 3] & ~(0xffffffffull << (((o) & 4) * 8))) | (((unsigned long long)(v) & 0xffffffffull) << (((o) & 4) * 8))
PB_MATH_FN int pb_libm_cvtt32(double v) { return (v != v || v >= 2147483648.0 || v < -2147483648.0) ? (int)0x80000000 : (int)v; }
PB_MATH_FN long long pb_libm_cvtt64(double v) { return (v != v || v >= 9223372036854775808.0 || v < -9223372036854775808.0) ? (long long)0x8000000000000000ull : (long long)v; }
'''


def main():
    path = libm_path()
    libc = ctypes.CDLL(None)
    libc.gnu_get_libc_version.restype = ctypes.c_char_p
    if libc.gnu_get_libc_version().decode() != "2.35":
        sys.exit("this machine's glibc is not 2.35")
    elf = Elf(path)
    assert elf.read(FUNCS[0][1], 4) == b"\xf3\x0f\x1e\xfa", "libm.so.6 is not the build this script was written against"
    # Which bytes of libm's read-only data each function's table loads can reach: found by running the --probe build of this very
    # output over EVERY high word of the argument with both signs (experiments/r5/libm_sweep.cpp: 1.5e9 arguments, 0 mismatches against
    # the machine's libm) - asincos.tbl's `asncs` (2 719 doubles around the base the code addresses), uatan.tbl's `cij` (241 x 7),
    # utan.tbl's `xfg` (186 x 4).  A load outside them returns NaN.
    extents = {"atan": [(0xB56E0, 0xB8B98)], "asin": [(0xB8BD8, 0xBE0D0)], "acos": [(0xB8BD8, 0xBE0D0)], "tan": [(0xC15C0, 0xC2CF8)]}
    probe = "--probe" in sys.argv
    print(PRELUDE)
    emitted = {}
    for name, lo, hi in FUNCS:
        g = Gen(elf, name, lo, hi, path)
        body = g.body()
        bases = sorted(g.tables)
        # the lookup tables this function addresses: [lo, hi) byte ranges around each base that the index arithmetic can reach
        print("// ---- %s (%#x-%#x) " % (name, lo, hi) + "-" * 60)
        merged = [(min(bases) - 8 * 4096, max(bases) + 8 * 8192)] if probe else extents[name]
        for k, (a, b) in enumerate(merged):
            if (a, b) in emitted:  # (asin and acos share asincos.tbl)
                print("#define PB_LIBM_%s_T%d %s" % (name.upper(), k, emitted[(a, b)]))
                continue
            emitted[(a, b)] = "PB_LIBM_%s_T%d" % (name.upper(), k)
            vals = struct.unpack("<%dQ" % ((b - a) // 8), elf.read(a, b - a))
            print("PB_MATH_CONST unsigned long long PB_LIBM_%s_T%d[%d] = {" % (name.upper(), k, len(vals)))
            for i in range(0, len(vals), 6):
                print("    " + ", ".join("%#xull" % v for v in vals[i : i + 6]) + ",")
            print("};")
        print("PB_MATH_FN unsigned long long pb_libm_ld_%s(unsigned long long a) {" % name)
        if probe:
            print("    pb_libm_probe(a);")
        for k, (a, b) in enumerate(merged):
            print("    if (a >= %#xull && a < %#xull) return PB_LIBM_%s_T%d[(a - %#xull) >> 3];" % (a, b, name.upper(), k, a))
        print("    return 0x7ff8000000000000ull;  // (outside every table: NaN, never a wild read)")
        print("}")
        print("PB_MATH_FN double pb_%s_libm(double x0) {" % name)
        print("    unsigned long long X[16] = {pb_bits(x0), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};")
        print("    unsigned long long " + ", ".join("%s = 0" % r for r in G64) + ", stk[8] = {0, 0, 0, 0, 0, 0, 0, 0};")
        print("    PbLibmFlags F = {1, 0, 0, 0, 0, 0.0, 0.0};")
        print("    (void)rbx; (void)rbp; (void)rsi; (void)rdi; (void)r8; (void)r9; (void)r10; (void)r11; (void)r12; (void)r13; (void)r14; (void)r15; (void)stk;")
        print("\n".join(body))
        print("    return pb_from_bits(X[0]);")
        print("}")
    print("#undef SX\n#undef FL_SUB\n#undef FL_RES\n#undef FL_FP\n#undef FP_UNORD\n#undef CC_E\n#undef CC_NE\n#undef CC_G\n#undef CC_GE\n#undef CC_L\n#undef CC_LE\n#undef CC_S\n#undef CC_A\n#undef CC_AE\n#undef CC_B\n#undef CC_BE\n#undef CC_P\n#undef STK64\n#undef STK64W\n#undef STK32\n#undef STK32W")


if __name__ == "__main__":
    main()
