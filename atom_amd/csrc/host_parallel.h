// A few host threads for the text writers: run(n, fn) calls fn(0) ... fn(n - 1), each once, on up to
// writer_threads() threads (ATOMDEV_WRITER_THREADS in the environment, default 8; 1 runs everything in the caller).
#ifndef ATOMDEV_HOST_PARALLEL_H_
#define ATOMDEV_HOST_PARALLEL_H_

#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

inline int writer_threads() {
    static const int n = [] {
        const char* e = std::getenv("ATOMDEV_WRITER_THREADS");
        const int v = e ? std::atoi(e) : 8;
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}

template <class Fn>
void host_parallel_run(size_t n, Fn fn) {
    const size_t threads = n < (size_t)writer_threads() ? n : (size_t)writer_threads();
    if (threads <= 1) { for (size_t c = 0; c < n; c++) fn(c); return; }
    std::atomic<size_t> next(0);
    auto work = [&]() { for (size_t c; (c = next.fetch_add(1)) < n;) fn(c); };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
}

#endif  // ATOMDEV_HOST_PARALLEL_H_
