/* Exhaustive check of the two-operation quotient used by bg_mean_u8_fast_kernel for frame/(n + 1):
 *     y  = RN(1/d);  yh = y with the low 9 significand bits cleared (44 bits);
 *     e  = fma(-d, y, 1)           (exact residual of the reciprocal)
 *     yl = fma(y, e, y - yh)       (= 1/d - yh to 2^-53 relative, i.e. 2^-96 relative to 1/d)
 *     q  = fma(f, yl, f * yh)      (f * yh is exact for an integer f < 2^9)
 * against the IEEE quotient f / d, for every f in 0..255 and every d in [1, 2^LOG2D] (default 24), plus random
 * d up to 2^40.  q differs from RN(f/d) only if f/d (1 + 2^-96) crosses a rounding midpoint, and a
 * non-dyadic f/d stays 2^-54 / d away from every midpoint (relative): impossible for d < 2^40.
 *     gcc -O2 -fopenmp -ffp-contract=off tools/verify_small_div.c -o /tmp/vsd -lm && /tmp/vsd [LOG2D]   */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int check_d(double d)
{
    double y = 1.0 / d;
    uint64_t b;
    memcpy(&b, &y, 8);
    b &= ~(uint64_t)0x1FF;
    double yh;
    memcpy(&yh, &b, 8);
    double e = fma(-d, y, 1.0), yl = fma(y, e, y - yh);
    int bad = 0;
    for (int f = 0; f < 256; f++) {
        double fr = (double)f, q = fma(fr, yl, fr * yh);
        if (q != fr / d)
            bad++;
    }
    return bad;
}

int main(int argc, char **argv)
{
    int lg = argc > 1 ? atoi(argv[1]) : 24;
    long long bad = 0, n = 1ll << lg;
#pragma omp parallel for reduction(+ : bad) schedule(static, 65536)
    for (long long d = 1; d <= n; d++)
        bad += check_d((double)d);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    long long badr = 0;
    for (int i = 0; i < 20000000; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        badr += check_d((double)(1 + (s >> 24)));           /* d up to 2^40 */
    }
    printf("d in [1, 2^%d] x f in [0, 255]: %lld mismatches; 2e7 random d up to 2^40: %lld mismatches\n", lg, bad, badr);
    return bad || badr ? 1 : 0;
}
