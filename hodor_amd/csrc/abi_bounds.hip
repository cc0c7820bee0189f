// abi_bounds.hip — host side of the bounds-checked build (bounds.cuh; `make bounds`): the report block the kernels write
// their violations to, the registry of the library's own device allocations that launchers' extent declarations are
// checked against, and the poll that turns a hit into an error at the library's synchronisation points.  In the shipped
// build this translation unit is empty.
#include "ctx.hpp"

#ifdef HODOR_BOUNDS
#include <cstdio>
#include <unistd.h>

namespace hodor {
namespace {
std::mutex g_mu;
std::map<uintptr_t, size_t> g_allocs;      // device allocations of the library: base -> bytes REQUESTED
BoundsReport *g_rep = nullptr;
unsigned long long g_seen_hits = 0, g_host_hits = 0;
std::atomic<unsigned long long> g_launches{0};
std::string g_first;

const char *kid_name(uint32_t k)
{
    static const char *N[] = {"?", "k_ntt_pass", "k_pow_table", "k_pow_table_w3", "k_pow_table_w9", "k_distribute_powers_small",
                              "k_distribute_powers", "k_degree_one_small", "k_degree_one", "k_scale", "k_binary", "k_add_scaled", "k_unary",
                              "k_quotient_term", "k_batchinv_forward", "k_batchinv_backward", "k_evaluate_at", "k_evaluate_at_table",
                              "k_twiddle_mul", "k_gen_elements", "k_store_elems", "k_count_diff", "k_dense_divisor", "k_merkle_subtree",
                              "k_merkle_levels", "k_iop_query", "k_iop_query_coset2", "k_challenge", "k_fri_round_table", "k_fri_fold",
                              "k_fri_fold_coeffs", "k_fri_tail", "k_pack_slabs", "k_transpose"};
    return k < sizeof(N) / sizeof(N[0]) ? N[k] : "?";
}

struct ExitSummary {   // HODOR_BOUNDS_REPORT=<file>: one line per process, appended when the library is unloaded
    ~ExitSummary()
    {
        const char *path = getenv("HODOR_BOUNDS_REPORT");
        if (!path || !*path) return;
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "pid %d: %llu checked launches, %llu device-side hits, %llu host-side hits%s%s\n", (int)getpid(),
                    g_launches.load(), g_seen_hits, g_host_hits, g_first.empty() ? "" : " | first: ", g_first.c_str());
            fclose(f);
        }
    }
} g_exit_summary;
}  // namespace

BoundsReport *bounds_report_dev()
{
    g_launches.fetch_add(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rep) {
        if (hipMalloc((void **)&g_rep, sizeof(BoundsReport)) != hipSuccess || hipMemset(g_rep, 0, sizeof(BoundsReport)) != hipSuccess) {
            (void)hipGetLastError();
            fprintf(stderr, "hodor bounds build: cannot allocate the report block\n");
            abort();   // a bounds build that cannot report is worse than none
        }
    }
    return g_rep;
}

bool bounds_shrink()
{
    static const bool on = [] { const char *e = getenv("HODOR_BOUNDS_SHRINK"); return e && *e && *e != '0'; }();
    return on;
}

void bounds_alloc_note(const void *p, size_t bytes)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs[(uintptr_t)p] = bytes;
}

void bounds_alloc_forget(const void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs.erase((uintptr_t)p);
}

// a launcher declares [p, p + bytes) for a kernel: when p lies inside an allocation of the library's, the declared range
// must too (memory the library did not allocate — a caller's buffer — is the caller's promise)
void bounds_check_declared(uint32_t kid, const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return;
    --it;
    const uintptr_t base = it->first, end = base + it->second;
    if ((uintptr_t)p >= end) return;                       // past that allocation: not the library's memory
    if ((uintptr_t)p + bytes <= end) return;
    g_host_hits++;
    char msg[256];
    snprintf(msg, sizeof(msg), "launcher of %s declares %zu bytes at offset %zu of an allocation of %zu bytes", kid_name(kid), bytes,
             (size_t)((uintptr_t)p - base), it->second);
    if (g_first.empty()) g_first = msg;
    fprintf(stderr, "hodor bounds: %s\n", msg);
}

// At a point where the host has just waited for the device: any new violation?  1 = yes (ctx's error message says what).
int bounds_poll(hodor_ctx *ctx)
{
    BoundsReport r;
    unsigned long long host_hits;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_rep) return 0;
        if (hipMemcpy(&r, g_rep, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0; }
        host_hits = g_host_hits;
        if (r.hits == g_seen_hits && host_hits == 0) return 0;
        if (r.hits != g_seen_hits && g_first.empty()) {
            char msg[320];
            static const char *KIND[] = {"range", "undeclared base", "LDS slot"};
            snprintf(msg, sizeof(msg), "%s site %u (%s): elements [%llu, +%llu) of the buffer at %#llx, extent %llu", kid_name(r.kid), r.site,
                     KIND[r.kind < 3 ? r.kind : 0], r.offset, r.count, r.base, r.extent);
            g_first = msg;
        }
        if (r.hits == g_seen_hits && host_hits) {
            // host-side hits only: reported once per poll that sees them
        }
        g_seen_hits = r.hits;
    }
    if (ctx) set_err(ctx, "bounds build: " + std::to_string(r.hits) + " device-side and " + std::to_string(host_hits) +
                              " host-side violations so far; first: " + g_first);
    return 1;
}

}  // namespace hodor

// for the harness: violations seen so far (device side as of the last poll + host side); forces a poll
extern "C" unsigned long long hodor_bounds_hits(void)
{
    (void)hipDeviceSynchronize();
    (void)hodor::bounds_poll(nullptr);
    return hodor::g_seen_hits + hodor::g_host_hits;
}
#endif
