/* Dense accuracy sweep of the oracle's restatements of Zig's std maths (oracle/zigmath.c, oracle/colorspaces.c) against
 * correctly rounded values: the reference is glibc's long double function (64-bit significand, < 1 ulp of THAT format, i.e.
 * 2^-40 of an f32 ulp) and the error is measured in units of the f32 ulp of the exact result.
 * usage: ulp_sweep <fn> — fn: exp | log | sin | cos | cbrt | pow24 | gamma | atan2. Prints one line:
 *   fn=<name> n=<inputs> max_ulp=<worst error> misrounded=<results that are not the correctly rounded f32> worst_x=<hex>
 * TEST INFRASTRUCTURE (tests/test_math_pin.py builds and runs it). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/zo.h"

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* error of `got` in ulps of the f32 binade of the exact value `ref` */
static double ulp_error(float got, long double ref) {
    if (isnan(got) || isinf(got)) return isnan((double)ref) || isinf((double)ref) ? 0 : 1e30;
    int e;
    frexpl(ref, &e); /* |ref| = m 2^e, m in [0.5, 1) -> ulp = 2^(e - 24), floored at the subnormal spacing */
    if (e < -125) e = -125;
    const long double ulp = ldexpl(1.0L, e - 24);
    return (double)(fabsl((long double)got - ref) / ulp);
}

struct range { uint32_t lo, hi; }; /* bit patterns of positive floats, inclusive; negatives by the caller */

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const char *fn = argv[1];
    double worst = 0;
    uint64_t n = 0, misrounded = 0;
    uint32_t worst_bits = 0;
#define VISIT(x, got, ref) do { \
        const long double r_ = (ref); const float g_ = (got); \
        const double e_ = ulp_error(g_, r_); ++n; \
        if (g_ != (float)r_) ++misrounded; \
        if (e_ > worst) { worst = e_; worst_bits = f2u(x); } } while (0)

    if (!strcmp(fn, "exp")) { /* @exp of -(x^2)/(2 sigma^2): (-inf, 0]; swept on [-104, 0) and (0, 16] densely where taps live */
        for (uint32_t b = f2u(0x1p-30f); b <= f2u(104.0f); b += 7) { const float x = -u2f(b); VISIT(x, zo_expf(x), expl((long double)x)); }
        for (uint32_t b = f2u(0x1p-30f); b <= f2u(16.0f); b += 29) { const float x = u2f(b); VISIT(x, zo_expf(x), expl((long double)x)); }
    } else if (!strcmp(fn, "log")) {
        for (uint32_t b = f2u(0x1p-40f); b <= f2u(0x1p40f); b += 37) { const float x = u2f(b); VISIT(x, zo_logf(x), logl((long double)x)); }
    } else if (!strcmp(fn, "sin") || !strcmp(fn, "cos")) { /* rotation / hue angles: |x| <= 64 pi, every sign */
        const int is_sin = fn[0] == 's';
        for (uint32_t b = f2u(0x1p-20f); b <= f2u(201.1f); b += 11)
            for (int sgn = 0; sgn < 2; ++sgn) {
                const float x = sgn ? -u2f(b) : u2f(b);
                if (is_sin) VISIT(x, zo_sinf(x), sinl((long double)x));
                else VISIT(x, zo_cosf(x), cosl((long double)x));
            }
    } else if (!strcmp(fn, "cbrt")) { /* LMS values of xyzToOklab: (0, ~1.3]; swept on [2^-40, 8] and negatives */
        for (uint32_t b = f2u(0x1p-40f); b <= f2u(8.0f); b += 23)
            for (int sgn = 0; sgn < 2; ++sgn) { const float x = sgn ? -u2f(b) : u2f(b); VISIT(x, zo_cbrtf(x), cbrtl((long double)x)); }
    } else if (!strcmp(fn, "pow24")) { /* gammaToLinear's pow((c + 0.055) / 1.055, 2.4): base in (0.09, 1]; swept on [2^-8, 4] */
        for (uint32_t b = f2u(0x1p-8f); b <= f2u(4.0f); b += 5) { const float x = u2f(b); VISIT(x, zo_powf(x, 2.4f), powl((long double)x, (long double)2.4f)); }
    } else if (!strcmp(fn, "gamma")) { /* the whole transfer function over c in [0, 1]: every float of [2^-12, 1] */
        for (uint32_t b = f2u(0x1p-12f); b <= f2u(1.0f); b += 3) {
            const float c = u2f(b), base = (c + 0.055f) / 1.055f; /* the f32 steps of color.zig:1255, then the exact power of that base */
            float out;
            zo_math_apply(8, &c, NULL, &out, 1);
            VISIT(c, out, c > 0.04045f ? powl((long double)base, (long double)2.4f) : (long double)(c / 12.92f));
        }
    } else if (!strcmp(fn, "atan2")) { /* hue = atan2(b, a): a lattice of directions and magnitudes */
        uint32_t s = 12345;
        for (int i = 0; i < (1 << 24); ++i) {
            s = s * 1664525u + 1013904223u; const float y = (float)((int32_t)s) * 0x1p-31f * 1.5f;
            s = s * 1664525u + 1013904223u; const float x = (float)((int32_t)s) * 0x1p-31f * 1.5f;
            VISIT(y, zo_atan2f(y, x), atan2l((long double)y, (long double)x));
        }
    } else {
        return 2;
    }
    printf("fn=%s n=%llu max_ulp=%.4f misrounded=%llu worst_x=0x%08x\n", fn, (unsigned long long)n, worst, (unsigned long long)misrounded, worst_bits);
    return 0;
}
