// host_blake2s.hpp — scalar BLAKE2s (RFC 7693) on the host, for the pieces of the IOP that are
// O(log n) per call and stay on the CPU side of the boundary: the key-block midstate handed to the
// kernels, get_path's sibling leaf hash and verify (/root/reference/src/iop/blake2s_trivial_iop.rs
// :8-16, :236-279).  Bulk hashing is in merkle.hip.
#pragma once
#include <stdint.h>
#include <string.h>

namespace hodor {

struct HostBlake2s {
    static constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                       0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

    static void compress(uint32_t h[8], const uint8_t block[64], uint64_t t, bool last)
    {
        static const uint8_t SIGMA[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
            {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
            {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
            {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
            {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
            {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++) {
            uint32_t w;
            memcpy(&w, block + 4 * i, 4);   // little-endian host
            m[i] = w;
        }
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
        v[12] ^= (uint32_t)t;
        v[13] ^= (uint32_t)(t >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] += v[b] + x; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] += v[d];     v[b] = rotr(v[b] ^ v[c], 12);
            v[a] += v[b] + y; v[d] = rotr(v[d] ^ v[a], 8);
            v[c] += v[d];     v[b] = rotr(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const uint8_t *s = SIGMA[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }

    // chaining value after the (zero-padded) key block: Params::new().hash_length(32).key(k)
    // .personal(p).to_state()
    static void keyed_midstate(uint32_t h[8], const uint8_t *key, size_t keylen,
                               const uint8_t *personal, size_t plen)
    {
        uint8_t param[32];
        memset(param, 0, 32);
        param[0] = 32;
        param[1] = (uint8_t)keylen;
        param[2] = 1;
        param[3] = 1;
        memcpy(param + 24, personal, plen > 8 ? 8 : plen);
        for (int i = 0; i < 8; i++) {
            uint32_t w;
            memcpy(&w, param + 4 * i, 4);
            h[i] = IV[i] ^ w;
        }
        uint8_t block[64];
        memset(block, 0, 64);
        memcpy(block, key, keylen);
        compress(h, block, 64, false);
    }

    // finish a hash of `len` (<= 64, > 0) message bytes on top of the midstate
    static void finish(const uint32_t mid[8], const uint8_t *msg, size_t len, uint8_t out[32])
    {
        uint32_t h[8];
        memcpy(h, mid, 32);
        uint8_t block[64];
        memset(block, 0, 64);
        memcpy(block, msg, len);
        compress(h, block, 64 + len, true);
        memcpy(out, h, 32);
    }
};

// Streaming keyed BLAKE2s with a NON-destructive finalize, as blake2s_simd::State behaves
// ("finalize is idempotent and the state can keep absorbing"): the transcript of
// /root/reference/src/transcript/mod.rs:39-80 relies on exactly that.
class HostBlake2sStream {
  public:
    HostBlake2sStream(const uint8_t *key, size_t keylen, const uint8_t *personal, size_t plen)
    {
        uint8_t param[32];
        memset(param, 0, 32);
        param[0] = 32;
        param[1] = (uint8_t)keylen;
        param[2] = 1;
        param[3] = 1;
        memcpy(param + 24, personal, plen > 8 ? 8 : plen);
        for (int i = 0; i < 8; i++) {
            uint32_t w;
            memcpy(&w, param + 4 * i, 4);
            h_[i] = HostBlake2s::IV[i] ^ w;
        }
        memset(buf_, 0, 64);
        if (keylen) {   // the zero-padded key is the first block; it stays buffered until more data arrives
            memcpy(buf_, key, keylen);
            buflen_ = 64;
        }
    }
    void update(const uint8_t *data, size_t len)
    {
        while (len) {
            if (buflen_ == 64) {   // more input follows, so the buffered block is not the last one
                t_ += 64;
                HostBlake2s::compress(h_, buf_, t_, false);
                buflen_ = 0;
            }
            size_t take = 64 - buflen_;
            if (take > len) take = len;
            memcpy(buf_ + buflen_, data, take);
            buflen_ += take;
            data += take;
            len -= take;
        }
    }
    void finalize(uint8_t out[32]) const
    {
        uint32_t h[8];
        memcpy(h, h_, 32);
        uint8_t block[64];
        memset(block, 0, 64);
        memcpy(block, buf_, buflen_);
        HostBlake2s::compress(h, block, t_ + buflen_, true);
        memcpy(out, h, 32);
    }

  private:
    uint32_t h_[8];
    uint8_t buf_[64];
    size_t buflen_ = 0;
    uint64_t t_ = 0;
};

}  // namespace hodor
