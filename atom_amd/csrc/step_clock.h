// Wall time of the steps of the one-call loops and slices.
#ifndef ATOMDEV_STEP_CLOCK_H_
#define ATOMDEV_STEP_CLOCK_H_

#include "atomdev_internal.h"

// With ATOMDEV_STEP_TIMES set in the environment every step of the loops and of the slice is bracketed by a stream
// synchronisation and its wall time goes to stderr ("step <ms> <call>"): where a slice spends its time, host work
// (readers, writers) included.  Off by default: the steps are then enqueued back to back.
#include <chrono>
#include <cstdio>
#include <cstdlib>
namespace {
struct StepTimes {
    bool on;
    StepTimes() : on(std::getenv("ATOMDEV_STEP_TIMES") != nullptr) {}
};
const StepTimes g_step_times;
struct StepClock {
    atomdev_t* h; const char* what; std::chrono::steady_clock::time_point t0;
    StepClock(atomdev_t* h_, const char* w) : h(h_), what(w) {
        if (g_step_times.on) { atomdev_sync(h); t0 = std::chrono::steady_clock::now(); }
    }
    ~StepClock() {
        if (!g_step_times.on) return;
        atomdev_sync(h);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "step %.3f %s\n", ms, what);
    }
};
}  // namespace

#endif  // ATOMDEV_STEP_CLOCK_H_
