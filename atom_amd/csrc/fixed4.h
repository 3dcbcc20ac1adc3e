// "%.4f" without printf: the text an ofstream with ios::fixed and precision(4) (or printf("%.4f")) gives for a double,
// byte for byte.  glibc converts exactly and rounds the exact binary value to the nearest decimal with four places,
// ties to the even last digit; here the nearest integer r to x * 10^4 is found with the product's exact residual (one
// fused multiply-add: its sign is the sign of the exact difference), so the decision at a half is the exact one, and r
// is written as digits.  Values of 4e11 and beyond, infinities and NaNs go through snprintf.
#ifndef ATOMDEV_FIXED4_H_
#define ATOMDEV_FIXED4_H_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

// writes the text at p (at least 32 bytes, 400 if the value may be huge) and returns the end; no terminating zero
inline char* put_fixed4(char* p, double x) {
    if (!(std::fabs(x) < 4.0e11)) return p + std::snprintf(p, 400, "%.4f", x);
    double r = std::rint(x * 1.0e4);
    const double d = std::fma(x, 1.0e4, -r);
    if (d > 0.) {
        const double t = std::fma(x, 1.0e4, -(r + .5));
        if (t > 0. || (t == 0. && std::fmod(r, 2.) != 0.)) r += 1.;
    } else if (d < 0.) {
        const double t = std::fma(x, 1.0e4, -(r - .5));
        if (t < 0. || (t == 0. && std::fmod(r, 2.) != 0.)) r -= 1.;
    }
    if (std::signbit(x)) *p++ = '-';
    const uint64_t a = (uint64_t)std::fabs(r);
    uint64_t whole = a / 10000u;
    const unsigned frac = (unsigned)(a % 10000u);
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + whole % 10u); whole /= 10u; } while (whole);
    while (n) *p++ = tmp[--n];
    *p++ = '.';
    p[0] = (char)('0' + frac / 1000u); p[1] = (char)('0' + frac / 100u % 10u);
    p[2] = (char)('0' + frac / 10u % 10u); p[3] = (char)('0' + frac % 10u);
    return p + 4;
}

inline void append_fixed4(std::string& s, double x) {
    char b[400];
    s.append(b, (size_t)(put_fixed4(b, x) - b));
}

#endif  // ATOMDEV_FIXED4_H_
