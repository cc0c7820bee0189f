// knobs.hpp — the library's tuning knobs (host side), shared by the launch helpers of every .hip file.
#pragma once

namespace hodor {

// Tuning knobs (environment), read ONCE per process by a C++11 thread-safe static in abi_host.hip.
// Results never depend on them; defaults are the measured optimum on MI355X.
struct Knobs {
    int max_log_r;        // HODOR_MAX_LOG_R       largest per-pass NTT radix (log2)                     9
    int tile_log;         // HODOR_TILE_LOG        elements per NTT workgroup tile (log2)               10
    int min_log_c;        // HODOR_MIN_LOG_C       fewest tile columns of an NTT pass (log2)                2
    int tw_hi_max_log;    // HODOR_TW_HI_MAX_LOG   largest hi-only twiddle split (log2 entries)         17
    int ntt_threads;      // HODOR_NTT_THREADS     workgroup size override of k_ntt_pass (0 = auto)      0
    int ntt_tw_sub;       // HODOR_NTT_TW_SUB      sub-sampled LDS twiddle table (last step from L2)      1
    int ntt_w9;           // HODOR_NTT_W9          wave-uniform W9 twiddles in the first steps: 0 off, 1 on, 2 on + skip products by one   2
    int ntt_p1;           // HODOR_NTT_P1          subtraction instead of v_mul_lo for the Montgomery digit when p = 1 mod 2^29   1
    int merkle_tail_log;  // HODOR_MERKLE_TAIL_LOG level width at which a throughput chunk stops         6
    int merkle_lat_log;   // HODOR_MERKLE_LAT_LOG  largest level on the latency schedule                19
    int fri_tail;         // HODOR_FRI_TAIL        fused tail of the FRI commit                          1
    int fri_fuse_fold;    // HODOR_FRI_FUSE_FOLD   fold inside the tree's leaf launch: 0 never, 1 always, 2 small rounds only   1
    int batchinv_seq;     // HODOR_BATCHINV_SEQ    elements per lane and level in batch inversion        8
    int table_cache;      // HODOR_TABLE_CACHE     power tables kept per context before the cache is emptied    40
    int slice_serial;     // HODOR_SLICE_SERIAL    slice API: one upload and one download at a time (0: every lane copies at will)   1
    int pool_cache_gib;   // HODOR_POOL_CACHE_GIB  idle device-pool bytes a context keeps before it frees the largest  64
    char set[256];        // "NAME=value ..." of the variables that were present in the environment
};
const Knobs &knobs();

}  // namespace hodor
