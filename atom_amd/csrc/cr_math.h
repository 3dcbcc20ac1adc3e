// exp and pow in double-double arithmetic: results that are the correctly rounded value except where the exact value
// lies within about 2^-15 ulp of a rounding boundary.  The host's libm (glibc) rounds its exp and pow correctly in all but
// about one call in a hundred, the device's library in far fewer; the column physics feeds these functions into an
// iteration that multiplies last-place differences, so the kernels there call these instead.  Plain C++ with explicit
// fused multiply-adds: compiles for the host (tests) and the device with the same bits.
#ifndef CR_MATH_H_
#define CR_MATH_H_

#include <cmath>
#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#define CR_FN __host__ __device__ inline
#define CR_TABLE static __device__ const
#else
#define CR_FN inline
#define CR_TABLE static const
#endif

#include "cr_tables.h"

#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
// host pass of a HIP translation unit: the tables live in device memory; host code of the library does not call these
#define CR_HOST_PASS_OF_HIP 1
#endif

namespace cr {

struct dd {
    double hi, lo;
};

CR_FN dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return { s, (a - (s - bb)) + (b - bb) };
}
CR_FN dd fast_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return { s, b - (s - a) };
}
CR_FN dd two_prod(double a, double b) {
    const double p = a * b;
    return { p, __builtin_fma(a, b, -p) };
}
CR_FN uint64_t bits_of(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
CR_FN double double_of(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}

// exp(xh + xl) for |xh| < 700, |xl| <= ulp(xh)
CR_FN double exp_dd(double xh, double xl) {
    const double k = __builtin_rint(xh * CR_EXP_INV);
    const long ki = (long)k;
    const int j = (int)(ki & 127);
    const long m = ki >> 7;
    // r = x - k ln2/128: the first product is exact, the second is carried as a pair
    const double r0 = __builtin_fma(-k, CR_EXP_C1, xh);
    const dd p = two_prod(k, CR_EXP_C2);
    dd r = two_sum(r0, -p.hi);
    r.lo += (xl - p.lo) - k * CR_EXP_C3;
    r = fast_two_sum(r.hi, r.lo);
    // exp(r) - 1 - r for |r| < 0.0028: the Taylor terms from the second on, 2^-71 relative to 1
    const double x = r.hi;
    const double q = x * x * (0.5 + x * (1. / 6. + x * (1. / 24. + x * (1. / 120. + x * (1. / 720. + x * (1. / 5040. + x * (1. / 40320.)))))));
    const dd e = two_sum(r.hi, r.lo + q);
#ifdef CR_HOST_PASS_OF_HIP
    const double th = 0., tl = 0.; (void)j;
#else
    const double th = cr_exp_tab[j][0], tl = cr_exp_tab[j][1];
#endif
    // t (1 + e)
    const dd pq = two_prod(th, e.hi);
    dd s = fast_two_sum(th, pq.hi);
    s.lo += pq.lo + (th * e.lo + (tl + tl * e.hi));
    const double y = s.hi + s.lo;
    // 2^m: exact while the result is a normal number
    if (m >= -1000 && m <= 1000) return y * double_of((uint64_t)(m + 1023) << 52);
    return ldexp(y, (int)m);
}

// ln(x) for a positive normal x, as a pair
CR_FN dd log_dd(double x) {
    const uint64_t ix = bits_of(x);
    const uint64_t tmp = ix - CR_LOG_OFF;
    const int j = (int)((tmp >> 45) & 127);
    const long e = (long)((int64_t)tmp >> 52);
    const double z = double_of(ix - (tmp & 0xfff0000000000000ULL));
#ifdef CR_HOST_PASS_OF_HIP
    const double invc = 1., lch = 0., lcl = 0.; (void)j;
#else
    const double invc = cr_log_tab[j][0], lch = cr_log_tab[j][1], lcl = cr_log_tab[j][2];
#endif
    // r = z / c - 1, exact as a pair (|r| <= 2^-8)
    const dd zc = two_prod(z, invc);
    const double rh = zc.hi - 1.;  // exact: zc.hi lies in [0.99, 1.01]
    const dd r = fast_two_sum(rh, zc.lo);
    // ln(1 + r) = r - r^2/2 + r^3 (1/3 - r/4 + ...): the square as a pair, the rest in double (2^-70 of the result)
    const double x1 = r.hi;
    dd sq = two_prod(x1, x1);
    sq.lo += 2. * x1 * r.lo;
    const double q = x1 * x1 * x1 * (1. / 3. + x1 * (-1. / 4. + x1 * (1. / 5. + x1 * (-1. / 6. + x1 * (1. / 7. + x1 * (-1. / 8. + x1 * (1. / 9.
                                   + x1 * (-1. / 10. + x1 * (1. / 11.)))))))));
    dd t = fast_two_sum(x1, -0.5 * sq.hi);
    t.lo += (r.lo - 0.5 * sq.lo) + q;
    // e ln 2 + ln c + ln(1 + r)
    const double ed = (double)e;
    const dd u = two_sum(ed * CR_LN2_1, lch);  // the product is exact
    const dd v = two_sum(u.hi, t.hi);
    const double lo = ((u.lo + v.lo) + (ed * CR_LN2_2 + lcl)) + (t.lo + ed * CR_LN2_3);
    return fast_two_sum(v.hi, lo);
}

}  // namespace cr

CR_FN double cr_exp(double x) {
    if (!(fabs(x) < 700.)) return exp(x);  // overflow, underflow, NaN: the library's treatment
    return cr::exp_dd(x, 0.);
}

CR_FN double cr_pow(double x, double y) {
    // the general cases of the library function are not needed here: finite positive normal base, finite exponent
    if (!(x >= 0x1p-1022 && x < INFINITY) || !(fabs(y) < 0x1p60)) return pow(x, y);
    if (y == 0. || x == 1.) return 1.;
    const cr::dd l = cr::log_dd(x);
    cr::dd p = cr::two_prod(y, l.hi);
    p.lo += y * l.lo;
    p = cr::fast_two_sum(p.hi, p.lo);
    if (!(fabs(p.hi) < 700.)) return pow(x, y);
    return cr::exp_dd(p.hi, p.lo);
}

// x^4 and y^(1/4), which the reference writes as pow(x, 4.) and pow(y, 1. / 4.): the square of the exact square as a
// pair; the root from two square roots and one correction step against the exact fourth power.
CR_FN double cr_pow4(double x) {
    const cr::dd p = cr::two_prod(x, x);
    const cr::dd q = cr::two_prod(p.hi, p.hi);
    return q.hi + (q.lo + 2. * p.hi * p.lo);
}
CR_FN double cr_root4(double y) {
    if (!(y > 0x1p-900 && y < 0x1p900)) return sqrt(sqrt(y));
    const double s = sqrt(sqrt(y));
    const cr::dd p = cr::two_prod(s, s);
    cr::dd q = cr::two_prod(p.hi, p.hi);
    q.lo += 2. * p.hi * p.lo;
    const double diff = (y - q.hi) - q.lo;  // the first difference is exact: s^4 is within a few ulp of y
    return s + diff / (4. * p.hi * s);
}

#endif  // CR_MATH_H_
