// Host check of atom_amd/csrc/fixed4.h (built and run by tests/test_abi.py): put_fixed4 against snprintf("%.4f"),
// byte for byte, on random values over forty binades, on halves of the last decimal place and their neighbours, on
// zeros of both signs, huge values, infinities and NaNs.
#include "fixed4.h"
#include <random>
#include <vector>
int main() {
    std::mt19937_64 rng(11);
    std::vector<double> xs;
    for (int n = 0; n < 4000000; n++)
        xs.push_back(std::ldexp((double)(rng() >> 11) / 9007199254740992.0 - .5, (int)(rng() % 60) - 30));
    for (int n = 0; n < 400000; n++) {  // dyadic values: the only doubles that can sit exactly on a half
        const double x = std::ldexp((double)((long long)(rng() % 2000001) - 1000000), -(int)(rng() % 20));
        xs.push_back(x); xs.push_back(std::nextafter(x, 1e300)); xs.push_back(std::nextafter(x, -1e300));
    }
    for (long long k = -30000; k <= 30000; k++) {  // (k + 1/2) / 10^4 as nearly as a double gets, and its neighbours
        const double x = ((double)k + .5) / 1.0e4;
        xs.push_back(x); xs.push_back(std::nextafter(x, 1e300)); xs.push_back(std::nextafter(x, -1e300));
    }
    for (double x : { 0., -0., 1e-300, -1e-300, 4.9e-324, 0.00005, -0.00005, 0.00015, 0.03125, -0.03125, 0.09375, 1e11, 3.9999e11,
                      4e11, -4e11, 1e15, 1e22, -1e300, 1.7976931348623157e308, (double)INFINITY, (double)-INFINITY, (double)NAN,
                      -(double)NAN, 99999.99995, 0.99995, 9.99995, 0.49995, 123456.78905 })
        xs.push_back(x);
    size_t bad = 0;
    char a[400], b[400];
    for (double x : xs) {
        const size_t la = (size_t)(put_fixed4(a, x) - a);
        const size_t lb = (size_t)std::snprintf(b, sizeof b, "%.4f", x);
        if (la != lb || std::memcmp(a, b, la) != 0) {
            if (bad++ < 10) { a[la] = 0; std::printf("MISMATCH %a: %s vs %s\n", x, a, b); }
        }
    }
    std::printf("%zu values, %zu mismatches\n", xs.size(), bad);
    return bad != 0;
}
