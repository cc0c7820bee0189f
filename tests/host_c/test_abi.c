/* Plain C11 against include/hodor_gpu.h: the boundary is a C ABI — no C++, no torch types.  Needs a GPU for
 * the compute part; without one it checks that compute entry points refuse with HODOR_ERR_DEVICE.
 * Build: gcc -std=c11 test_abi.c -L<repo>/hodor_amd -lhodor_gpu */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/hodor_gpu.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static const uint64_t MODULUS[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};

int main(int argc, char **argv)
{
    int device = argc > 1 ? atoi(argv[1]) : -1;
    hodor_ctx *ctx = NULL;
    CHECK(hodor_ctx_create(MODULUS, 7, device, &ctx) == HODOR_OK);
    hodor_field_info fi;
    CHECK(hodor_ctx_field_info(ctx, &fi) == HODOR_OK && fi.s == 32 && fi.num_bits == 255);
    enum { LOG_N = 10, N = 1 << LOG_N };
    static hodor_fr a[N], b[N];
    for (int i = 0; i < N; i++) {            /* canonical residues i + 1, converted to Montgomery form */
        uint64_t c[4] = {(uint64_t)i + 1, 0, 0, 0};
        CHECK(hodor_fr_from_repr(ctx, c, &a[i]) == HODOR_OK);
    }
    memcpy(b, a, sizeof(a));
    if (device < 0) {
        CHECK(hodor_poly_fft(ctx, b, N) == HODOR_ERR_DEVICE);      /* no CPU fallback */
        printf("host_c: no device, compute refused as designed\n");
        hodor_ctx_destroy(ctx);
        return 0;
    }
    CHECK(hodor_poly_fft(ctx, b, N) == HODOR_OK);                  /* Polynomial::fft */
    CHECK(memcmp(a, b, sizeof(a)) != 0);
    /* X[0] = sum of the inputs = N (N + 1) / 2 */
    uint64_t c0[4];
    CHECK(hodor_fr_into_repr(ctx, &b[0], c0) == HODOR_OK);
    CHECK(c0[0] == (uint64_t)N * (N + 1) / 2 && c0[1] == 0 && c0[2] == 0 && c0[3] == 0);
    CHECK(hodor_poly_ifft(ctx, b, N) == HODOR_OK);                 /* Polynomial::ifft */
    CHECK(memcmp(a, b, sizeof(a)) == 0);
    static uint8_t nodes[N * 32];
    CHECK(hodor_iop_create(ctx, a, N, nodes) == HODOR_OK);         /* Blake2sIopTree::create */
    uint8_t l[32], r[32], h[32];
    CHECK(hodor_hash_leaf(ctx, &a[0], l) == HODOR_OK && hodor_hash_leaf(ctx, &a[1], r) == HODOR_OK);
    CHECK(hodor_hash_node(ctx, l, r, h) == HODOR_OK);
    CHECK(memcmp(h, nodes + 32 * (N / 2), 32) == 0);               /* first node of the lowest stored level */
    CHECK(hodor_fft(ctx, b, N + 1, &fi.root_of_unity, LOG_N) == HODOR_ERR_SIZE);   /* assert_eq!(n, 1 << log_n) */
    /* the 4-step transform with the library's own exchange, one rank (what a multi-GPU caller does per rank):
     * columns -> hodor_sixstep_exchange_dev -> wait -> rows, then the inverse, on a 2^5 x 2^5 split */
    if (hodor_exchange_available()) {
        enum { L1 = 5, L2 = 5 };
        uint8_t id[HODOR_EXCHANGE_ID_BYTES];
        hodor_exchange *x = NULL;
        void *d_a = NULL, *d_s = NULL, *d_r = NULL, *d_b = NULL;
        uint64_t ticket = 0, dom_size = 0;
        uint32_t dom_log = 0;
        hodor_fr omega;
        CHECK(hodor_domain_new_for_size(ctx, N, &dom_size, &dom_log, &omega) == HODOR_OK && dom_log == L1 + L2);
        CHECK(hodor_exchange_unique_id(id) == HODOR_OK);
        CHECK(hodor_exchange_create(ctx, id, 1, 0, &x) == HODOR_OK);
        CHECK(hodor_buf_alloc(ctx, sizeof(a), &d_a) == HODOR_OK && hodor_buf_alloc(ctx, sizeof(a), &d_s) == HODOR_OK);
        CHECK(hodor_buf_alloc(ctx, sizeof(a), &d_r) == HODOR_OK && hodor_buf_alloc(ctx, sizeof(a), &d_b) == HODOR_OK);
        CHECK(hodor_buf_upload(ctx, d_a, a, sizeof(a)) == HODOR_OK);
        CHECK(hodor_sixstep_columns_dev(ctx, NULL, d_a, d_s, L1, L2, 0, 0, &omega, 0, 0, 0) == HODOR_OK);
        CHECK(hodor_sixstep_exchange_dev(x, NULL, d_s, d_r, N, 0, 0, &ticket) == HODOR_OK && ticket == 1);
        CHECK(hodor_sixstep_exchange_wait_dev(x, NULL, ticket) == HODOR_OK);
        CHECK(hodor_sixstep_rows_dev(ctx, NULL, d_r, d_b, L1, L2, 0, 0, &omega, 0, 0, 0) == HODOR_OK);
        /* B = X[k1 + N1 k2] as an N1 x N2 matrix: transposed it is the natural-order transform of above */
        CHECK(hodor_transpose_dev(ctx, NULL, d_b, d_s, 1u << L1, 1u << L2) == HODOR_OK);
        CHECK(hodor_buf_download(ctx, b, d_s, sizeof(a)) == HODOR_OK);
        {
            static hodor_fr want[N];
            memcpy(want, a, sizeof(a));
            CHECK(hodor_poly_fft(ctx, want, N) == HODOR_OK);
            CHECK(memcmp(want, b, sizeof(a)) == 0);
        }
        CHECK(hodor_sixstep_rows_dev(ctx, NULL, d_b, d_s, L1, L2, 0, 0, &omega, 1, 0, 0) == HODOR_OK);
        CHECK(hodor_sixstep_exchange_dev(x, NULL, d_s, d_r, N, 0, 0, &ticket) == HODOR_OK && ticket == 2);
        CHECK(hodor_sixstep_exchange_wait_dev(x, NULL, 0) == HODOR_OK);
        CHECK(hodor_sixstep_columns_dev(ctx, NULL, d_r, d_b, L1, L2, 0, 0, &omega, 1, 0, 0) == HODOR_OK);
        CHECK(hodor_buf_download(ctx, b, d_b, sizeof(a)) == HODOR_OK);
        CHECK(memcmp(a, b, sizeof(a)) == 0);
        hodor_exchange_destroy(x);
        hodor_buf_free(ctx, d_a); hodor_buf_free(ctx, d_s); hodor_buf_free(ctx, d_r); hodor_buf_free(ctx, d_b);
        printf("host_c: 4-step transform through the library's exchange (one-rank RCCL) ok\n");
    }
    printf("host_c: all tests passed\n");
    hodor_ctx_destroy(ctx);
    return 0;
}
