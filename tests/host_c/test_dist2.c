/* test_dist2.c — TWO PROCESSES, plain C11, nothing but include/hodor_gpu.h: what two Rust prover processes of a node
 * do with the library's own multi-GPU schedules (csrc/abi_dist.hip) over the transports that map peer memory.
 *
 * The parent forks before anything touches HIP; parent = rank 0, child = rank 1.  Each rank creates its context and a
 * direct-transport exchange handle, lets the LIBRARY allocate its receive buffers (fine-grained device memory),
 * exports them and its flag block as hipIpc handles over a socketpair, imports the peer's, and then runs
 *   1. hodor_dist_ntt_natural_dev forward and inverse   (three exchanges each) against the single-device transform,
 *   2. hodor_dist_lde_commit_dev                        (LDE by cosets + commit by subtrees) against the single-device
 *                                                       LDE and tree,
 *   3. a SOAK of the transport's memory model: G generations of hodor_dist_ntt_forward_dev on ALTERNATING payloads, small
 *      enough to sit in L2 (2^12 points) and larger (2^18): every generation's output block is compared with the known
 *      answer for its payload — a line of the previous generation surviving anywhere (the consumer's L2s, a write still
 *      in flight when the flag arrived) is a mismatch.  Both the direct stores and the copy engine.
 * On a single-GPU box both ranks use device 0 (what the test suite runs); on a node, `test_dist2 G 0 1` runs the same
 * program between two devices — the first thing to do on the first multi-GPU box.
 *
 *   gcc -std=c11 -O2 test_dist2.c -L<repo>/hodor_amd -lhodor_gpu -o test_dist2 ;  ./test_dist2 [generations] [dev0] [dev1] */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/hodor_gpu.h"

static const uint64_t MODULUS[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static int RANK = -1;

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "rank %d: CHECK failed %s:%d: %s  [%s]\n", RANK, __FILE__, __LINE__, #c, \
                                          g_ctx ? hodor_last_error(g_ctx) : ""); return 1; } } while (0)
static hodor_ctx *g_ctx = NULL;

static int xfer(int fd, void *buf, size_t n, int sending)
{
    uint8_t *p = (uint8_t *)buf;
    while (n) {
        ssize_t k = sending ? write(fd, p, n) : read(fd, p, n);
        if (k <= 0) return -1;
        p += k;
        n -= (size_t)k;
    }
    return 0;
}
static int barrier(int fd)   /* both ranks have reached this point */
{
    uint8_t t = 1, r = 0;
    if (xfer(fd, &t, 1, 1) || xfer(fd, &r, 1, 0)) return -1;
    return 0;
}

enum { SLOTS = 4 };

static int run(int rank, int fd, int device, long generations)
{
    RANK = rank;
    hodor_ctx *ctx = NULL;
    CHECK(hodor_ctx_create(MODULUS, 7, device, &ctx) == HODOR_OK);
    g_ctx = ctx;
    void *stream = NULL;   /* the HIP default stream */
    enum { P = 2, LOG_BIG = 18, LOG_SMALL = 12, LDE_LOG_N = 13, LDE_FACTOR = 8 };
    const size_t n_big = (size_t)1 << LOG_BIG, m_big = n_big / P, lde_big = ((size_t)1 << LDE_LOG_N) * LDE_FACTOR;
    const size_t recv_elems = m_big > lde_big / P ? m_big : lde_big / P;

    /* ---- the handle, its buffers, the peer's */
    hodor_exchange *x = NULL;
    CHECK(hodor_exchange_create_direct(ctx, P, (uint32_t)rank, SLOTS, &x) == HODOR_OK);
    void *recv[SLOTS], *flags = NULL;
    size_t flag_bytes = 0;
    CHECK(hodor_exchange_direct_alloc_recv(x, recv_elems, 0 /* fine-grained */, recv) == HODOR_OK);
    CHECK(hodor_exchange_direct_flags(x, &flags, &flag_bytes) == HODOR_OK);
    uint8_t mine[1 + SLOTS][HODOR_IPC_HANDLE_BYTES], theirs[1 + SLOTS][HODOR_IPC_HANDLE_BYTES];
    CHECK(hodor_ipc_export(ctx, flags, mine[0]) == HODOR_OK);
    for (int s = 0; s < SLOTS; s++) CHECK(hodor_ipc_export(ctx, recv[s], mine[1 + s]) == HODOR_OK);
    CHECK(xfer(fd, mine, sizeof mine, 1) == 0 && xfer(fd, theirs, sizeof theirs, 0) == 0);
    void *peer_flags = NULL, *peer_recv[SLOTS];
    CHECK(hodor_ipc_import(ctx, theirs[0], &peer_flags) == HODOR_OK);
    for (int s = 0; s < SLOTS; s++) CHECK(hodor_ipc_import(ctx, theirs[1 + s], &peer_recv[s]) == HODOR_OK);
    for (int s = 0; s < SLOTS; s++) {
        void *r[P], *f[P];
        r[rank] = recv[s]; r[1 - rank] = peer_recv[s];
        f[rank] = flags;   f[1 - rank] = peer_flags;
        CHECK(hodor_exchange_direct_set_peers(x, (uint32_t)s, r, f) == HODOR_OK);
    }
    CHECK(barrier(fd) == 0);

    for (int transport = HODOR_TRANSPORT_DIRECT; transport <= HODOR_TRANSPORT_COPY; transport++) {
        CHECK(hodor_dist_set_transport(x, transport, 0) == HODOR_OK);
        const char *tname = transport == HODOR_TRANSPORT_DIRECT ? "direct stores" : "copy engine";

        /* ---- 1. natural-order transform of 2^18 points over the two ranks */
        {
            void *d_full = NULL, *d_spec = NULL, *d_out = NULL, *d_back = NULL;
            uint64_t dom = 0;
            uint32_t lg = 0;
            hodor_fr omega;
            CHECK(hodor_domain_new_for_size(ctx, n_big, &dom, &lg, &omega) == HODOR_OK && lg == LOG_BIG);
            CHECK(hodor_buf_alloc(ctx, n_big * 32, &d_full) == HODOR_OK && hodor_buf_alloc(ctx, n_big * 32, &d_spec) == HODOR_OK);
            CHECK(hodor_buf_alloc(ctx, m_big * 32, &d_out) == HODOR_OK && hodor_buf_alloc(ctx, m_big * 32, &d_back) == HODOR_OK);
            CHECK(hodor_gen_elements_dev(ctx, stream, (hodor_fr *)d_full, 0, n_big, 901) == HODOR_OK);
            CHECK(hodor_poly_fft_dev(ctx, stream, (const hodor_fr *)d_full, (hodor_fr *)d_spec, LOG_BIG) == HODOR_OK);
            const hodor_fr *my_block = (const hodor_fr *)d_full + (size_t)rank * m_big;
            CHECK(hodor_dist_ntt_natural_dev(x, stream, my_block, (hodor_fr *)d_out, m_big, LOG_BIG, &omega, 0) == HODOR_OK);
            CHECK(hodor_dist_ntt_natural_dev(x, stream, (const hodor_fr *)d_out, (hodor_fr *)d_back, m_big, LOG_BIG, &omega, 1) == HODOR_OK);
            CHECK(hodor_ctx_synchronize(ctx) == HODOR_OK);
            hodor_fr *h_a = malloc(m_big * 32), *h_b = malloc(m_big * 32);
            CHECK(h_a && h_b);
            CHECK(hodor_buf_download(ctx, h_a, d_out, m_big * 32) == HODOR_OK);
            CHECK(hodor_buf_download(ctx, h_b, (const hodor_fr *)d_spec + (size_t)rank * m_big, m_big * 32) == HODOR_OK);
            CHECK(memcmp(h_a, h_b, m_big * 32) == 0);                       /* my block of the single-device transform */
            CHECK(hodor_buf_download(ctx, h_a, d_back, m_big * 32) == HODOR_OK);
            CHECK(hodor_buf_download(ctx, h_b, my_block, m_big * 32) == HODOR_OK);
            CHECK(memcmp(h_a, h_b, m_big * 32) == 0);                       /* and back */
            free(h_a); free(h_b);
            hodor_buf_free(ctx, d_full); hodor_buf_free(ctx, d_spec); hodor_buf_free(ctx, d_out); hodor_buf_free(ctx, d_back);
        }

        /* ---- 2. LDE x8 of 2^13 coefficients by cosets + commit by subtrees, both tree formats */
        for (int combiner = HODOR_COMBINER_TRIVIAL; combiner <= HODOR_COMBINER_COSET2; combiner++) {
            const size_t n = (size_t)1 << LDE_LOG_N, B = lde_big / P, entries = combiner ? lde_big / 2 : lde_big;
            void *d_c = NULL, *d_lde = NULL, *d_nodes = NULL, *d_blk = NULL, *d_local = NULL;
            CHECK(hodor_buf_alloc(ctx, n * 32, &d_c) == HODOR_OK && hodor_buf_alloc(ctx, lde_big * 32, &d_lde) == HODOR_OK);
            CHECK(hodor_buf_alloc(ctx, entries * 32, &d_nodes) == HODOR_OK && hodor_buf_alloc(ctx, B * 32, &d_blk) == HODOR_OK);
            CHECK(hodor_buf_alloc(ctx, B * 32, &d_local) == HODOR_OK);
            CHECK(hodor_gen_elements_dev(ctx, stream, (hodor_fr *)d_c, 0, n, 902) == HODOR_OK);
            CHECK(hodor_poly_lde_dev(ctx, stream, (const hodor_fr *)d_c, (hodor_fr *)d_lde, LDE_LOG_N, LDE_FACTOR, 1) == HODOR_OK);
            CHECK(hodor_iop_create_combined_dev(ctx, stream, (const hodor_fr *)d_lde, lde_big, combiner, (uint8_t *)d_nodes) == HODOR_OK);
            uint8_t top[2 * P * 32], root[32], ref_top[2 * P * 32];
            CHECK(hodor_dist_lde_commit_dev(x, stream, (const hodor_fr *)d_c, LDE_LOG_N, LDE_FACTOR, 1, combiner, (hodor_fr *)d_blk,
                                            (uint8_t *)d_local, top, root) == HODOR_OK);
            CHECK(hodor_buf_download(ctx, ref_top, d_nodes, sizeof ref_top) == HODOR_OK);
            CHECK(memcmp(root, ref_top + 32, 32) == 0);                                      /* the root of the single-device tree */
            CHECK(memcmp(top + 32, ref_top + 32, (2 * P - 1) * 32) == 0);                    /* and the replicated levels */
            hodor_fr *h_blk = malloc(B * 32), *h_ref = malloc(B * 32);
            CHECK(h_blk && h_ref);
            CHECK(hodor_buf_download(ctx, h_blk, d_blk, B * 32) == HODOR_OK);
            if (!combiner) {                              /* my natural block */
                CHECK(hodor_buf_download(ctx, h_ref, (const hodor_fr *)d_lde + (size_t)rank * B, B * 32) == HODOR_OK);
            } else {                                      /* my PAIRED block: [d B/2, (d+1) B/2) and N/2 + the same */
                CHECK(hodor_buf_download(ctx, h_ref, (const hodor_fr *)d_lde + (size_t)rank * (B / 2), (B / 2) * 32) == HODOR_OK);
                CHECK(hodor_buf_download(ctx, h_ref + B / 2, (const hodor_fr *)d_lde + lde_big / 2 + (size_t)rank * (B / 2), (B / 2) * 32) == HODOR_OK);
            }
            CHECK(memcmp(h_blk, h_ref, B * 32) == 0);
            free(h_blk); free(h_ref);
            hodor_buf_free(ctx, d_c); hodor_buf_free(ctx, d_lde); hodor_buf_free(ctx, d_nodes); hodor_buf_free(ctx, d_blk); hodor_buf_free(ctx, d_local);
        }

        /* ---- 3. the soak: G generations, alternating payloads, every generation's block compared */
        const int logs[2] = {LOG_SMALL, LOG_BIG};
        for (int li = 0; li < 2; li++) {
            const uint32_t log_n = (uint32_t)logs[li];
            const size_t n = (size_t)1 << log_n, m = n / P;
            const long G = li == 0 ? generations : generations / 20 + 3;
            uint32_t l1, l2;
            hodor_dist_split(log_n, &l1, &l2);
            uint64_t dom = 0;
            uint32_t lg = 0;
            hodor_fr omega;
            CHECK(hodor_domain_new_for_size(ctx, n, &dom, &lg, &omega) == HODOR_OK);
            void *d_a[2], *d_b = NULL;
            hodor_fr *expect[2], *got = malloc(m * 32);
            CHECK(got != NULL);
            CHECK(hodor_buf_alloc(ctx, m * 32, &d_b) == HODOR_OK);
            /* payload p: layout A (my column block) of stream 910 + p; the known answer: my block of layout B, obtained
             * ONCE through the same schedule and cross-checked against the single-device transform */
            for (int p = 0; p < 2; p++) {
                void *d_full = NULL, *d_spec = NULL;
                hodor_fr *h_full = malloc(n * 32), *h_spec = malloc(n * 32), *h_a = malloc(m * 32);
                CHECK(h_full && h_spec && h_a);
                CHECK(hodor_buf_alloc(ctx, n * 32, &d_full) == HODOR_OK && hodor_buf_alloc(ctx, n * 32, &d_spec) == HODOR_OK);
                CHECK(hodor_gen_elements_dev(ctx, stream, (hodor_fr *)d_full, 0, n, 910 + (uint64_t)p) == HODOR_OK);
                CHECK(hodor_poly_fft_dev(ctx, stream, (const hodor_fr *)d_full, (hodor_fr *)d_spec, log_n) == HODOR_OK);
                CHECK(hodor_ctx_synchronize(ctx) == HODOR_OK);
                CHECK(hodor_buf_download(ctx, h_full, d_full, n * 32) == HODOR_OK && hodor_buf_download(ctx, h_spec, d_spec, n * 32) == HODOR_OK);
                const size_t N1 = (size_t)1 << l1, N2 = (size_t)1 << l2, c2 = N2 / P, r1 = N1 / P;
                for (size_t i = 0; i < N1; i++)               /* a[n1][j] = x[n1 N2 + rank c2 + j] */
                    memcpy(h_a + i * c2, h_full + i * N2 + (size_t)rank * c2, c2 * 32);
                expect[p] = malloc(m * 32);
                CHECK(expect[p] != NULL);
                for (size_t i = 0; i < r1; i++)               /* b[i][k2] = X[(rank r1 + i) + N1 k2] */
                    for (size_t k2 = 0; k2 < N2; k2++) expect[p][i * N2 + k2] = h_spec[((size_t)rank * r1 + i) + N1 * k2];
                CHECK(hodor_buf_alloc(ctx, m * 32, &d_a[p]) == HODOR_OK);
                CHECK(hodor_buf_upload(ctx, d_a[p], h_a, m * 32) == HODOR_OK);
                free(h_full); free(h_spec); free(h_a);
                hodor_buf_free(ctx, d_full); hodor_buf_free(ctx, d_spec);
            }
            CHECK(barrier(fd) == 0);
            long bad = 0;
            for (long g = 0; g < G; g++) {
                const int p = (int)(g & 1);
                CHECK(hodor_dist_ntt_forward_dev(x, stream, (const hodor_fr *)d_a[p], (hodor_fr *)d_b, m, log_n, &omega, 0) == HODOR_OK);
                CHECK(hodor_buf_download(ctx, got, d_b, m * 32) == HODOR_OK);   /* (blocking: the generation is complete) */
                if (memcmp(got, expect[p], m * 32) != 0) {
                    if (!bad) fprintf(stderr, "rank %d: %s, 2^%u points: generation %ld differs from its payload's answer%s\n", rank, tname,
                                      log_n, g, memcmp(got, expect[1 - p], m * 32) == 0 ? " — it is the PREVIOUS generation's (stale data)" : "");
                    bad++;
                }
            }
            CHECK(hodor_exchange_direct_status(x) == HODOR_OK);
            if (rank == 0) printf("dist2: soak, %s, 2^%u points over 2 ranks: %ld generations, %ld mismatches\n", tname, log_n, G, bad);
            CHECK(bad == 0);
            free(got); free(expect[0]); free(expect[1]);
            hodor_buf_free(ctx, d_a[0]); hodor_buf_free(ctx, d_a[1]); hodor_buf_free(ctx, d_b);
        }
    }
    CHECK(barrier(fd) == 0);
    CHECK(hodor_ctx_synchronize(ctx) == HODOR_OK);
    hodor_exchange_destroy(x);
    CHECK(hodor_ipc_close(ctx, peer_flags) == HODOR_OK);
    for (int s = 0; s < SLOTS; s++) CHECK(hodor_ipc_close(ctx, peer_recv[s]) == HODOR_OK);
    hodor_ctx_destroy(ctx);
    g_ctx = NULL;
    return 0;
}

int main(int argc, char **argv)
{
    const long generations = argc > 1 ? atol(argv[1]) : 10000;
    const int dev0 = argc > 2 ? atoi(argv[2]) : 0, dev1 = argc > 3 ? atoi(argv[3]) : 0;
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 2; }
    fflush(NULL);
    pid_t child = fork();                 /* before any HIP call: each process initialises its own runtime */
    if (child < 0) { perror("fork"); return 2; }
    if (child == 0) {
        close(sv[0]);
        int rc = run(1, sv[1], dev1, generations);
        fflush(NULL);
        _exit(rc);
    }
    close(sv[1]);
    int rc = run(0, sv[0], dev0, generations);
    close(sv[0]);
    int status = 0;
    waitpid(child, &status, 0);
    const int child_rc = WIFEXITED(status) ? WEXITSTATUS(status) : 100 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
    if (rc == 0 && child_rc == 0) { printf("dist2: all tests passed (2 processes, devices %d / %d)\n", dev0, dev1); return 0; }
    fprintf(stderr, "dist2: rank 0 -> %d, rank 1 -> %d\n", rc, child_rc);
    return 1;
}
