// slice_threads.cpp — the slice API (host `&mut [F]` in, host out: the literal drop-in for best_fft's callers) under N
// concurrent caller threads on ONE context, as the reference calls best_fft from one scoped thread per register
// (/root/reference/src/arp/per_register/mod.rs:43-49).  Compiled code, no interpreter in the loop.
//   g++ -O2 -std=c++17 -pthread bench/slice_threads.cpp -Lhodor_amd -lhodor_gpu -Wl,-rpath,$PWD/hodor_amd -o /tmp/slice_threads
//   /tmp/slice_threads [reps=8] [pinned=0] [only_log_n=0] [only_threads=0]      (HODOR_SLICE_TRACE=1: the library prints each call's phases)
// Prints, per size and thread count: wall time per transform (all threads together), the aggregate rate, and the ratio to
// N x the single-caller rate.  A transform moves n*32 bytes up and n*32 bytes down; the link gives ~57 GB/s per direction.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../include/hodor_gpu.h"

static const uint64_t MODULUS[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 8;
    const bool pinned = argc > 2 && atoi(argv[2]) != 0;
    const unsigned only_log = argc > 3 ? atoi(argv[3]) : 0;
    const int only_threads = argc > 4 ? atoi(argv[4]) : 0;
    hodor_ctx *ctx = nullptr;
    if (hodor_ctx_create(MODULUS, 7, 0, &ctx)) { fprintf(stderr, "no context\n"); return 1; }
    printf("slice API, %s host memory, %d transforms per thread; knobs: [%s]\n", pinned ? "registered (pinned)" : "pageable", reps, hodor_knobs_set());
    for (unsigned log_n : {20u, 22u, 24u}) {
        if (only_log && log_n != only_log) continue;
        const size_t n = (size_t)1 << log_n;
        double single = 0;
        for (int threads : {1, 2, 3, 4, 6}) {
            if (only_threads && threads != 1 && threads != only_threads) continue;
            std::vector<std::vector<hodor_fr>> bufs(threads);
            std::mt19937_64 rng(log_n * 100 + threads);
            for (auto &b : bufs) {
                b.resize(n);
                for (auto &e : b) { for (int k = 0; k < 4; k++) e.l[k] = rng(); e.l[3] &= (1ull << 62) - 1; }
                if (pinned && hodor_host_register(ctx, b.data(), n * 32)) { fprintf(stderr, "register failed\n"); return 1; }
                if (hodor_poly_fft(ctx, b.data(), n)) { fprintf(stderr, "warm-up failed: %s\n", hodor_last_error(ctx)); return 1; }
            }
            int bad = 0;
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> ts;
            for (int t = 0; t < threads; t++)
                ts.emplace_back([&, t] { for (int r = 0; r < reps; r++) if (hodor_poly_fft(ctx, bufs[t].data(), n)) bad++; });
            for (auto &t : ts) t.join();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const double per = ms / (threads * reps);
            if (threads == 1) single = per;
            printf("2^%u, %d thread%s: %8.3f ms per transform  %6.2f GB/s each way  %5.2f x the single caller%s\n", log_n, threads,
                   threads > 1 ? "s" : " ", per, n * 32 / (per * 1e-3) / 1e9, single / per, bad ? "  (ERRORS)" : "");
            if (pinned) for (auto &b : bufs) hodor_host_unregister(ctx, b.data());
        }
    }
    hodor_ctx_destroy(ctx);
    return 0;
}
