/* log2_check.c - the host build of qvz_amd/csrc/qvz_log2_glibc.h against this machine's libm log2, bit for bit.
 * TEST INFRASTRUCTURE (tests/test_host_cpu.py builds it with gcc -O2 -ffp-contract=off).   log2_check <count> <seed> */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qvz_log2_glibc.h"

static uint64_t s;
static uint64_t rnd(void) { uint64_t x = (s += 0x9E3779B97F4A7C15ull); x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    s = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    long bad = 0, near1 = 0;
    for (long it = 0; it < n; ++it) {
        double x;
        const uint64_t r = rnd();
        switch (it & 7) {
        case 0: case 1: {                                   /* a probability as the pmfs hold them: count / total */
            const uint64_t t = 1 + (rnd() >> (rnd() % 40 + 24)), c = 1 + r % t;
            x = (double)c / (double)t; break; }
        case 2: x = qvz_log2_from_bits((r >> 12) | ((uint64_t)(1023 - (rnd() % 64)) << 52)); break;   /* any mantissa, exponents 0 .. -63 */
        case 3: x = 1.0 - ldexp((double)(r >> 11), -53 - (int)(rnd() % 30)); break;                 /* just below 1 */
        case 4: x = qvz_log2_from_bits(0x3FEEA4AF00000000ull + r % 0x210AA00000000ull + (rnd() % 5) - 2); break;   /* the close-to-1 branch and its edges */
        case 5: x = ldexp(1.0, -(int)(r % 1000)) * (1.0 + (double)(rnd() % 3) * 0x1p-52); break;    /* powers of two and neighbours */
        case 6: x = (double)(1 + r % 1000000) * (double)(1 + rnd() % 1000000) * 0x1p-60; break;     /* products of small factors */
        default: x = qvz_log2_from_bits(r & 0x7FEFFFFFFFFFFFFFull); if (!(x > 0x1p-1022)) x = 0.5; break;   /* any normal double */
        }
        const double a = log2(x), b = qvz_log2_glibc(x);
        if (qvz_log2_bits(x) - 0x3FEEA4AF00000000ull < 0x210AA00000000ull) ++near1;
        if (qvz_log2_bits(a) != qvz_log2_bits(b) && ++bad <= 10) printf("MISMATCH x=%a libm=%a mine=%a\n", x, a, b);
    }
    printf("log2_check: %ld arguments, %ld in the close-to-1 branch, %ld mismatches\n", n, near1, bad);
    return bad != 0;
}
