// bounds.cuh — the bounds-checked device build (`make bounds`, -DHODOR_BOUNDS -> libhodor_gpu_bounds.so).
//
// ROCm has no compute-sanitizer and AddressSanitizer does not survive the GPU boxes, while the handle API hands out
// POOLED device memory (abi_poly.hip): an out-of-range store lands in a neighbouring polynomial and only a lucky
// differential test sees it.  In this build every kernel carries, as its last argument, the EXTENTS of the buffers it was
// given (`BX`: up to 32 (base pointer, bytes) pairs filled in by its host launcher from the sizes of the call), and every
// global load / store goes through BAT(site, base, offset, count): the element range [offset, offset + count) of the
// buffer that starts at `base` must lie inside the extent declared for THAT base — exact per buffer, views into a shared
// slab included.  LDS slots go through LAT(site, slot, slots).  A violation is recorded (kernel id, site, offset, extent;
// the first one in full, all of them counted) in a report block in device memory, the load is served from offset 0 and
// the store is dropped into a sink; the host looks at the report at every point where it waits for the device anyway
// (HostXfer::finish, hodor_ctx_synchronize, hodor_ctx_destroy) and turns a hit into HODOR_ERR_DEVICE with the details in
// hodor_last_error.  The launchers additionally check every extent they declare against the library's own allocation
// registry (pool blocks at their REQUESTED size, scratch, tables, work buffers): a launcher that promises more than the
// allocation holds is reported on the host before the kernel runs.
//
// The shipped library is compiled without HODOR_BOUNDS: BXPARAM / BXARG vanish and BAT is plain pointer arithmetic.
// Run: `make -C hodor_amd/csrc bounds`, then HODOR_LIB=hodor_amd/libhodor_gpu_bounds.so python -m pytest tests -m gpu
// (bench/bounds_suite.sh; log under profiles/r06/).
#pragma once
#include <stdint.h>

namespace hodor {

// kernel ids (BX::kid) — what a report names
enum : uint32_t {
    KID_NTT_PASS = 1, KID_POW_TABLE, KID_POW_TABLE_W3, KID_POW_TABLE_W9,
    KID_DISTRIBUTE_SMALL, KID_DISTRIBUTE, KID_DEGREE_ONE_SMALL, KID_DEGREE_ONE, KID_SCALE, KID_BINARY, KID_ADD_SCALED,
    KID_UNARY, KID_QUOTIENT_TERM, KID_BATCHINV_FWD, KID_BATCHINV_BWD, KID_EVALUATE_AT, KID_EVALUATE_AT_TABLE,
    KID_TWIDDLE_MUL, KID_GEN_ELEMENTS, KID_STORE_ELEMS, KID_COUNT_DIFF, KID_DENSE_DIVISOR,
    KID_MERKLE_SUBTREE, KID_MERKLE_LEVELS, KID_IOP_QUERY, KID_IOP_QUERY_COSET2, KID_CHALLENGE,
    KID_FRI_ROUND_TABLE, KID_FRI_FOLD, KID_FRI_FOLD_COEFFS, KID_FRI_TAIL, KID_SIXSTEP_PACK, KID_TRANSPOSE,
    KID_COUNT
};

#ifdef HODOR_BOUNDS

struct BoundsReport {           // device memory, one per process
    unsigned long long hits;    // violations so far
    unsigned int first_taken;   // 0 -> 1 by the first violation, which fills in the fields below
    unsigned int kid, site, kind;   // kind: 0 global range, 1 undeclared base, 2 LDS slot
    unsigned long long offset, count, extent, base;
};

constexpr int BX_SLOTS = 32;
struct BX {
    const void *lo[BX_SLOTS];
    unsigned long long bytes[BX_SLOTS];
    BoundsReport *rep;
    uint32_t kid, n;
};

__device__ __attribute__((noinline)) inline void bx_report(const BX &X, uint32_t site, uint32_t kind, unsigned long long off,
                                                          unsigned long long cnt, unsigned long long extent, const void *base)
{
    atomicAdd(&X.rep->hits, 1ull);
    if (atomicCAS(&X.rep->first_taken, 0u, 1u) == 0u) {
        X.rep->kid = X.kid;
        X.rep->site = site;
        X.rep->kind = kind;
        X.rep->offset = off;
        X.rep->count = cnt;
        X.rep->extent = extent;
        X.rep->base = (unsigned long long)base;
    }
}

static __device__ uint4 hodor_bounds_sink[64];   // where refused stores go (per translation unit; never read)

// elements [off, off + cnt) of the array of T at `base`: inside the extent declared for `base`?
template <class T>
__device__ __forceinline__ T *bx_at(const BX &X, uint32_t site, T *base, unsigned long long off, unsigned long long cnt, bool store)
{
    for (uint32_t i = 0; i < X.n; i++)
        if (X.lo[i] == (const void *)base) {
            const unsigned long long ext = X.bytes[i] / sizeof(T);
            if (off <= ext && cnt <= ext - off) return base + off;
            bx_report(X, site, 0, off, cnt, ext, (const void *)base);
            return store ? (T *)(void *)hodor_bounds_sink : base;
        }
    bx_report(X, site, 1, off, cnt, 0, (const void *)base);
    return base + off;     // undeclared: the access goes where the shipped build would send it
}
__device__ __forceinline__ uint32_t bx_lds(const BX &X, uint32_t site, uint32_t slot, uint32_t slots)
{
    if (slot < slots) return slot;
    bx_report(X, site, 2, slot, 1, slots, nullptr);
    return 0;
}

#define BXPARAM , BX X
#define BXPARAM_DEF , BX X = BX()   // after parameters that have default arguments
#define BXDECL BX X
#define BXARG(x) , x
#define BXPASS , X
#define BAT(site, base, off, cnt) hodor::bx_at(X, (site), (base), (unsigned long long)(off), (unsigned long long)(cnt), false)
#define BATS(site, base, off, cnt) hodor::bx_at(X, (site), (base), (unsigned long long)(off), (unsigned long long)(cnt), true)
#define LAT(site, slot, slots) hodor::bx_lds(X, (site), (slot), (slots))
// BATP / BATSP: the shipped build keeps the pointer expression it always had (`plain`), so that its code does not move
#define BATP(site, base, off, cnt, plain) ((void)(plain), BAT(site, base, off, cnt))
#define BATSP(site, base, off, cnt, plain) ((void)(plain), BATS(site, base, off, cnt))

#else

#define BXPARAM
#define BXPARAM_DEF
#define BXARG(x)
#define BXPASS
#define BAT(site, base, off, cnt) ((base) + (off))
#define BATS(site, base, off, cnt) ((base) + (off))
#define LAT(site, slot, slots) (slot)
#define BATP(site, base, off, cnt, plain) (plain)
#define BATSP(site, base, off, cnt, plain) (plain)

#endif

}  // namespace hodor

// ---- host side (launchers): BXB(kid) starts the extents of one launch, .add(ptr, bytes) declares a buffer
#ifdef HODOR_BOUNDS
namespace hodor {
BoundsReport *bounds_report_dev();                                   // abi_bounds.hip: the process's report block
void bounds_check_declared(uint32_t kid, const void *p, size_t bytes);   // against the allocation registry (host)
void bounds_alloc_note(const void *p, size_t bytes);                  // a device allocation of the library's (REQUESTED size)
void bounds_alloc_forget(const void *p);
bool bounds_shrink();
struct BXB {
    BX x;
    explicit BXB(uint32_t kid)
    {
        x.n = 0;
        x.kid = kid;
        x.rep = bounds_report_dev();
        for (int i = 0; i < BX_SLOTS; i++) { x.lo[i] = nullptr; x.bytes[i] = 0; }
    }
    BXB &add(const void *p, size_t bytes)
    {
        if (!p) return *this;
        // HODOR_BOUNDS_SHRINK=1 (the test of the test, tests/test_gpu_bounds.py): every extent is declared one element
        // short, so that the kernels' own last accesses are violations
        if (bounds_shrink() && bytes >= 64) bytes -= 32;
        for (uint32_t i = 0; i < x.n; i++)
            if (x.lo[i] == p) { if (bytes > x.bytes[i]) x.bytes[i] = bytes; return *this; }   // in place: one buffer, two roles
        if (x.n < (uint32_t)BX_SLOTS) {
            x.lo[x.n] = p;
            x.bytes[x.n] = bytes;
            x.n++;
        }
        bounds_check_declared(x.kid, p, bytes);
        return *this;
    }
    operator BX() const { return x; }
};
}  // namespace hodor
#define BX_BEGIN(name, kid) hodor::BXB name(kid)
#define BX_ADD(name, p, bytes) name.add((const void *)(p), (size_t)(bytes))
#define BOUNDS_NOTE(p, bytes) hodor::bounds_alloc_note((p), (bytes))
#define BOUNDS_FORGET(p) hodor::bounds_alloc_forget((p))
#else
#define BX_BEGIN(name, kid) do {} while (0)
#define BX_ADD(name, p, bytes) do {} while (0)
#define BOUNDS_NOTE(p, bytes) ((void)0)
#define BOUNDS_FORGET(p) ((void)0)
#endif
