/*
 * ref_shim.h - what the two shim translation units around the compiled reference share: the context that
 * holds the reference's jagged Array objects next to the caller's flat arrays, and the copies between them.
 * TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).
 */
#ifndef REF_SHIM_H_
#define REF_SHIM_H_
#include <cstring>
#include <new>
#include <iostream>
#include <sstream>
#include <tuple>
#include <vector>

#include "Array.h"
#include "Array_1D.h"
#include "Array_2D.h"
#include "Utils.h"
#include "RHS_Atm.h"
#include "RungeKutta_Atm.h"
#include "Pressure_Atm.h"
#include "BC_Atm.h"
#include "BC_Bath_Atm.h"
#include "Accuracy_Atm.h"
#include "Results_Atm.h"
#include "PostProcess_Atm.h"
#include "MinMax_Atm.h"
// the reference reuses these include guards for the hydrosphere headers
#undef _PRESSURE_
#undef _ACCURACY_
#undef _MINMAX_
#include "RHS_Hyd.h"
#include "RungeKutta_Hyd.h"
#include "Pressure_Hyd.h"
#include "BC_Hyd.h"
#include "BC_Bath_Hyd.h"
#include "BC_Thermohalin.h"
#include "Accuracy_Hyd.h"
#include "PostProcess_Hyd.h"
#include "MinMax_Hyd.h"
#include "Results_Hyd.h"

#include "../include/atomdev.h"


struct RefCtx {
    int model, im, jm, km;
    atomdev_params p;
    double* f[ATOMDEV_N3D];
    double* g[ATOMDEV_N2D];
    Array* a[ATOMDEV_N3D];
    Array_2D* b[ATOMDEV_N2D];
    Array rhs[8];
    Array_1D rad, the, phi;
};

// The reference prints progress tables on std::cout; keep test output clean.
struct CoutSilencer {
    std::streambuf* old;
    std::ostringstream sink;
    CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~CoutSilencer() { std::cout.rdbuf(old); }
};

inline void copy_in(RefCtx* c) {
    for (int n = 0; n < ATOMDEV_N3D; n++)
        if (c->f[n])
            for (int i = 0; i < c->im; i++)
                for (int j = 0; j < c->jm; j++)
                    std::memcpy(c->a[n]->x[i][j], c->f[n] + ((size_t)i * c->jm + j) * c->km, sizeof(double) * c->km);
    for (int n = 0; n < ATOMDEV_N2D; n++)
        if (c->g[n])
            for (int j = 0; j < c->jm; j++)
                std::memcpy(c->b[n]->y[j], c->g[n] + (size_t)j * c->km, sizeof(double) * c->km);
}

inline void copy_out(RefCtx* c) {
    for (int n = 0; n < ATOMDEV_N3D; n++)
        if (c->f[n])
            for (int i = 0; i < c->im; i++)
                for (int j = 0; j < c->jm; j++)
                    std::memcpy(c->f[n] + ((size_t)i * c->jm + j) * c->km, c->a[n]->x[i][j], sizeof(double) * c->km);
    for (int n = 0; n < ATOMDEV_N2D; n++)
        if (c->g[n])
            for (int j = 0; j < c->jm; j++)
                std::memcpy(c->g[n] + (size_t)j * c->km, c->b[n]->y[j], sizeof(double) * c->km);
}

#define A(F) (*c->a[ATOMDEV_##F])
#define B(G) (*c->b[ATOMDEV_##G])



#endif
