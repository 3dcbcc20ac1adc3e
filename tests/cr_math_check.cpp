// Checks atom_amd/csrc/cr_math.h on the host: cr_exp and cr_pow against the x87 long-double functions (64 significand
// bits: decides the correct rounding of a double result unless the value lies within 2^-9 ulp of a rounding boundary) and
// counts how often glibc's double functions return the other neighbour.  Prints one line per function; exit code 1 if
// cr_* is ever on the wrong side of a decidable case.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "cr_math.h"

static int decidable_mismatch(double got, long double exact) {
    const double nearest = (double)exact;  // rounds to nearest
    if (got == nearest) return 0;
    // ambiguous when the long double sits within 2^-9 ulp of the midpoint between two doubles
    const double other = got;
    const long double mid = ((long double)nearest + (long double)other) / 2;
    const long double ulp = fabsl((long double)nearest - (long double)other);
    if (fabsl(exact - mid) < ulp / 512) return 0;
    return 1;
}

int main() {
    std::mt19937_64 rng(12345);
    int rc = 0;
    {
        std::uniform_real_distribution<double> ux(-40., 40.);
        long n = 4000000, bad = 0, libm_differs = 0;
        for (long i = 0; i < n; i++) {
            const double x = i % 4 == 0 ? ux(rng) * 17. : ux(rng);
            const double a = cr_exp(x);
            bad += decidable_mismatch(a, expl((long double)x));
            libm_differs += a != exp(x);
        }
        printf("exp: %ld arguments, cr_exp wrong in %ld decidable cases, glibc differs from cr_exp in %ld (%.3f %%)\n", n, bad,
               libm_differs, 100. * libm_differs / n);
        rc |= bad != 0;
    }
    {
        const double ys[] = { 1. / 3., 2. / 3., 7. / 9., 13. / 9., 1. / 6., 4. / 9., 5. / 26., 8. / 13., 1.5, 5.2558, 3.4161, -2.5, 0.286, 17.3 };
        std::uniform_real_distribution<double> ub(-12., 6.);
        long n = 4000000, bad = 0, libm_differs = 0;
        for (long i = 0; i < n; i++) {
            const double x = i % 3 == 0 ? 1. + (ub(rng) * 1e-3) : exp2(ub(rng));
            const double y = i % 5 == 0 ? ub(rng) : ys[i % (sizeof ys / sizeof ys[0])];
            const double a = cr_pow(x, y);
            bad += decidable_mismatch(a, powl((long double)x, (long double)y));
            libm_differs += a != pow(x, y);
        }
        printf("pow: %ld arguments, cr_pow wrong in %ld decidable cases, glibc differs from cr_pow in %ld (%.3f %%)\n", n, bad,
               libm_differs, 100. * libm_differs / n);
        rc |= bad != 0;
    }
    {
        std::uniform_real_distribution<double> ux(150., 330.), uy(1e6, 2e10);
        long n = 4000000, bad = 0, libm_differs = 0;
        for (long i = 0; i < n; i++) {
            const double x = ux(rng), y = uy(rng);
            const double a = cr_pow4(x), b = cr_root4(y);
            bad += decidable_mismatch(a, powl((long double)x, 4.L)) + decidable_mismatch(b, sqrtl(sqrtl((long double)y)));
            libm_differs += (a != pow(x, 4.)) + (b != pow(y, 1. / 4.));
        }
        printf("x^4, y^(1/4): %ld arguments each, wrong in %ld decidable cases, glibc's pow differs in %ld (%.3f %%)\n", n, bad,
               libm_differs, 50. * libm_differs / n);
        rc |= bad != 0;
    }
    return rc;
}
