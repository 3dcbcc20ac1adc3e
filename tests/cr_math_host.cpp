// atom_amd/csrc/cr_math.h compiled for the host behind a C interface, so that tests can hold the device's results
// against the bits the same source gives here.
#include <cstddef>

#include "cr_math.h"

extern "C" void cr_host_eval(int kind, size_t n, const double* x, const double* y, double* out) {
    for (size_t q = 0; q < n; q++) {
        switch (kind) {
        case 0: out[q] = cr_exp(x[q]); break;
        case 1: out[q] = cr_pow(x[q], y[q]); break;
        case 2: out[q] = cr_pow4(x[q]); break;
        case 3: out[q] = cr_root4(x[q]); break;
        }
    }
}
