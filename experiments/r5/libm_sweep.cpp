#include <cmath>
#include <initializer_list>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define PB_MATH_FN static inline
#define PB_MATH_CONST static const
static inline unsigned long long pb_bits(double d) { unsigned long long u; memcpy(&u, &d, 8); return u; }
static inline double pb_from_bits(unsigned long long u) { double d; memcpy(&d, &u, 8); return d; }
static unsigned long long g_min = ~0ull, g_max = 0;
static inline void pb_libm_probe(unsigned long long a) { if (a < g_min) g_min = a; if (a > g_max) g_max = a; }
#include GEN  // -DGEN=\"...\": the generated header (--probe output for the extents, the committed file for the check)
typedef double (*fn_t)(double);
// every high word of the argument (both signs) with three low words: every table index the functions can form
static void sweep(const char* name, fn_t mine, fn_t ref, unsigned hi0, unsigned hi1) {
    g_min = ~0ull; g_max = 0;
    long bad = 0, n = 0;
    for (unsigned hi = hi0; hi <= hi1; ++hi)
        for (int s = 0; s < 2; ++s)
            for (unsigned lo : {0u, 0x9e3779b9u, 0xffffffffu}) {
                const double x = pb_from_bits(((unsigned long long)(hi | (s ? 0x80000000u : 0u)) << 32) | lo);
                const double a = mine(x), b = ref(x);
                ++n;
                if (pb_bits(a) != pb_bits(b) && !(a != a && b != b)) { if (bad < 3) printf("  %s(%a) = %a, libm %a\n", name, x, a, b); ++bad; }
            }
    printf("%s: %ld arguments, %ld mismatches, table addresses [%#llx, %#llx]\n", name, n, bad, g_min, g_max);
}
int main() {
    sweep("asin", pb_asin_libm, asin, 0x3e000000u, 0x3ff00000u);
    sweep("acos", pb_acos_libm, acos, 0x3c000000u, 0x3ff00000u);
    sweep("atan", pb_atan_libm, atan, 0x3e000000u, 0x43500000u);
    sweep("tan", pb_tan_libm, tan, 0x3e000000u, 0x41900000u);
    return 0;
}
