#!/usr/bin/env python3
"""Generates pb_np_tables.hpp: the data behind pb_math_np.hpp, the restatement of the float64 arcsin / arccos / arctan / tan that
NumPy 2.2.6 runs on an AVX512_SKX machine (Intel SVML's `_ha` kernels, which NumPy vendors as assembly).

The data is what the AVX-512 instructions VRSQRT14PD and VRCP14PD return, which those kernels start from.  Both are deterministic
functions of the operand's exponent and its top mantissa bits (15 bits + the exponent's parity for the reciprocal square root, 16
bits for the reciprocal; an exact power of four / two returns the exact result).  They are SAMPLED here from the CPU this script
runs on - which must have AVX-512F - one probe per bucket at its two ends and a point inside, and stored as 16-bit mantissas,
delta-coded: one 16-bit base per 16 buckets and 2-bit steps (consecutive buckets differ by 0..3 units), 24 KiB + 24 KiB instead of
128 + 128 KiB.  (The kernels' polynomial coefficients and small lookup tables are written out in pb_math_np.hpp itself.)
Run once on an AVX-512 machine; the output is committed:
    python photonbend_amd/csrc/gen_np_tables.py > photonbend_amd/csrc/pb_np_tables.hpp"""
import os
import subprocess
import sys
import tempfile

PROBE = r"""
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static double fromb(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t rsq(uint64_t xb) { double o[8]; _mm512_storeu_pd(o, _mm512_rsqrt14_pd(_mm512_set1_pd(fromb(xb)))); return bits(o[0]); }
static uint64_t rcp(uint64_t xb) { double o[8]; _mm512_storeu_pd(o, _mm512_rcp14_pd(_mm512_set1_pd(fromb(xb)))); return bits(o[0]); }
int main(void) {
    /* reciprocal square root: operand 2^p * 1.m, p = 0, 1; bucket = m >> 37 */
    for (int p = 0; p < 2; ++p)
        for (uint64_t i = 0; i < 32768; ++i) {
            const uint64_t b = ((uint64_t)(0x3ff + p) << 52) | (i << 37);
            const uint64_t a = rsq(b | 1), z = rsq(b | ((1ull << 37) - 1)), c = rsq(b | 0x155555555ull);
            if (a != z || a != c || (a >> 52) != 0x3fe || (a & ((1ull << 36) - 1))) return 2;
            printf("%llu\n", (unsigned long long)((a >> 36) & 0xffff));
        }
    /* reciprocal: operand 1.m; bucket = m >> 36 */
    for (uint64_t i = 0; i < 65536; ++i) {
        const uint64_t b = (0x3ffull << 52) | (i << 36);
        const uint64_t a = rcp(b | 1), z = rcp(b | ((1ull << 36) - 1)), c = rcp(b | 0x555555555ull);
        if (a != z || a != c || (a >> 52) != 0x3fe || (a & ((1ull << 36) - 1))) return 3;
        printf("%llu\n", (unsigned long long)((a >> 36) & 0xffff));
    }
    /* exact powers, and the scaling by the exponent */
    if (rsq(bits(1.0)) != bits(1.0) || rsq(bits(4.0)) != bits(0.5) || rsq(bits(0x1p-40)) != bits(0x1p+20)) return 4;
    if (rcp(bits(1.0)) != bits(1.0) || rcp(bits(2.0)) != bits(0.5) || rcp(bits(0x1p-40)) != bits(0x1p+40)) return 5;
    for (int e = -300; e <= 300; e += 7) {
        const double x = 1.2345678901234567, s = ldexp(1.0, 2 * e);
        if (rsq(bits(x * s)) != bits(fromb(rsq(bits(x))) * ldexp(1.0, -e))) return 6;
        if (rcp(bits(x * s)) != bits(fromb(rcp(bits(x))) / s)) return 7;
        if (rcp(bits(-x * s)) != bits(-fromb(rcp(bits(x))) / s)) return 8;
    }
    return 0;
}
"""


def sample():
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "probe.c"), os.path.join(tmp, "probe")
        open(src, "w").write(PROBE)
        subprocess.run(["gcc", "-O2", "-mavx512f", src, "-o", exe, "-lm"], check=True)
        res = subprocess.run([exe], capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(f"probe failed ({res.returncode}): this CPU's VRSQRT14PD / VRCP14PD do not have the expected structure (or there is no AVX-512F)")
        v = [int(t) for t in res.stdout.split()]
        assert len(v) == 2 * 32768 + 65536
        return v[:65536], v[65536:]


def delta_code(tab):
    base, steps = [], []
    for b in range(0, len(tab), 16):
        blk = tab[b : b + 16]
        base.append(blk[0])
        w = 0
        for j in range(15):
            d = blk[j] - blk[j + 1]
            assert 0 <= d <= 3, (b, j, d)
            w |= d << (2 * j)
        steps.append(w)
    return base, steps


def decode(base, steps, i):  # what pb_math_np.hpp does; checked below against the sampled table
    w = steps[i >> 4] & ((1 << (2 * (i & 15))) - 1)
    return base[i >> 4] - bin(w & 0x55555555).count("1") - 2 * bin(w & 0xAAAAAAAA).count("1")


def emit_u(name, ctype, vals, per_line, fmt):
    print(f"PB_MATH_CONST {ctype} {name}[{len(vals)}] = {{")
    for i in range(0, len(vals), per_line):
        print("    " + ", ".join(fmt % v for v in vals[i : i + per_line]) + ",")
    print("};")


def main():
    rsq, rcp = sample()
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    print("// pb_np_tables.hpp - GENERATED by gen_np_tables.py; do not edit.")
    print("// SPDX-License-Identifier: BSD-3-Clause  (Intel SVML tables as vendored by NumPy; see NOTICE)")
    print(f"// VRSQRT14PD / VRCP14PD sampled on: {cpu}")
    print("#pragma once")
    for name, tab in (("PB_RSQRT14", rsq), ("PB_RCP14", rcp)):
        base, steps = delta_code(tab)
        assert all(decode(base, steps, i) == tab[i] for i in range(len(tab)))
        emit_u(name + "_BASE", "unsigned short", base, 16, "0x%04x")
        emit_u(name + "_STEP", "unsigned", steps, 16, "0x%08x")


if __name__ == "__main__":
    main()
