/* A plain-C consumer of the reference's plug-in boundary: includes the cgo-style header, links -lmultiexp, and replays the
 * KZG call sequence of Porla's Client and Server (porla/Client/Client.hpp:159-167,348-354,411-419,1637-1662;
 * porla/Server/Server.hpp:183-188,363-398,550-558,900-901) through the 14 symbols, checking the identities the protocol
 * relies on.  It shows that the engine is a link-time drop-in for libmultiexp.so (same names, same GoSlice ABI) without
 * any Python or C++ in between.
 *
 *   gcc -O2 -I../../include harness.c -L../../porla_amd -lmultiexp -Wl,-rpath,$PWD/../../porla_amd -o harness
 *   ./harness cpu    host-side calls only (client side: needs no GPU)
 *   ./harness gpu    everything, including compute_digest_from_srs / create_proof / compute_multi_exp on the MI355X
 *   ./harness bench [threads] [calls]   the server's call pattern timed: `threads` pool threads (default 8, Server.hpp:1054-1078)
 *                    each calling compute_digest_from_srs one row at a time; prints one JSON line (commits/s, latency per call)
 */
#include "libmultiexp.h"
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#define NUM_CHUNKS 128                 /* porla/config.hpp:22 */
#define BLOCK_SIZE (NUM_CHUNKS * 32)   /* porla/config.hpp:20 */

static const uint8_t TAU_KEY[16] = {0xff, 0xee, 0xdd, 0xcc, 0xbb, 0xaa, 0x99, 0x88, 0x77, 0x66, 0x55, 0x44, 0x33, 0x22, 0x11, 0x00};
static const uint8_t SECRET_KEY[16] = {0x00, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xaa, 0xbb, 0xcc, 0xdd, 0xee, 0xff};

static GoSlice slice(void* p, long long n) { GoSlice s; s.data = p; s.len = n; s.cap = n; return s; }
static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL: %s\n", what); failures++; } else printf("ok:   %s\n", what); } while (0)

/* bn254_scalar_set_int, porla/Utils/utils.h:271-275 */
static void scalar_set_int(uint8_t out[32], uint32_t v) { memset(out, 0, 32); out[28] = v >> 24; out[29] = v >> 16; out[30] = v >> 8; out[31] = v; }
static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s; }

typedef struct { int calls; uint32_t seed; int bad; } bench_job;
static void* bench_worker(void* arg) {
    bench_job* j = (bench_job*)arg;
    uint8_t row[BLOCK_SIZE], out[64], first[64];
    uint32_t s = j->seed;
    for (int i = 0; i < BLOCK_SIZE; i++) row[i] = (uint8_t)(lcg(&s) >> 24);
    GoSlice s_row = slice(row, BLOCK_SIZE), s_out = slice(out, 64);
    for (int k = 0; k < j->calls; k++) {
        compute_digest_from_srs(&s_row, &s_out);
        if (k == 0) memcpy(first, out, 64);
        else if (memcmp(first, out, 64) != 0) j->bad++;
    }
    return NULL;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv) {
    const int bench = argc > 1 && strcmp(argv[1], "bench") == 0;
    const int gpu = bench || (argc > 1 && strcmp(argv[1], "gpu") == 0);
    uint8_t tau[16], alpha[16];
    memcpy(tau, TAU_KEY, 16); memcpy(alpha, SECRET_KEY, 16);
    GoSlice s_tau = slice(tau, 16), s_alpha = slice(alpha, 16);
    init_key(&s_tau, &s_alpha);                                         /* Client.hpp:159-167 */

    static uint8_t srs_blob[32 * NUM_CHUNKS + 132];
    GoSlice s_blob = slice(srs_blob, sizeof srs_blob);
    GoInt64 blob_len = 0;
    init_SRS(NUM_CHUNKS, &s_blob, &blob_len);                           /* Client.hpp:348-354 */
    CHECK(blob_len == 32 * NUM_CHUNKS + 132, "init_SRS wire blob is 32n+132 bytes (Client.hpp:350-357)");
    CHECK(srs_blob[0] == 0 && srs_blob[1] == 0 && srs_blob[2] == 0 && srs_blob[3] == NUM_CHUNKS, "blob starts with the big-endian count");

    if (bench) {
        int T = argc > 2 ? atoi(argv[2]) : 8, calls = argc > 3 ? atoi(argv[3]) : 2000;
        if (T < 1) T = 1;
        if (T > 64) T = 64;
        GoSlice s_in = slice(srs_blob, (long long)blob_len);
        init_SRS_from_data(NUM_CHUNKS, &s_in);
        pthread_t th[64];
        bench_job jobs[64];
        bench_job warm = {20, 99u, 0};
        bench_worker(&warm);                                            /* builds the SRS table */
        double t0 = now_s();
        for (int t = 0; t < T; t++) { jobs[t].calls = calls; jobs[t].seed = 1000u + t; jobs[t].bad = 0; pthread_create(&th[t], NULL, bench_worker, &jobs[t]); }
        int bad = 0;
        for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); bad += jobs[t].bad; }
        double el = now_s() - t0;
        printf("{\"call\": \"compute_digest_from_srs\", \"caller\": \"C, pthreads\", \"threads\": %d, \"calls_per_thread\": %d, "
               "\"commits_per_s\": %.1f, \"latency_ms_per_call\": %.4f, \"consistent\": %s}\n",
               T, calls, T * (double)calls / el, el / calls * 1e3, bad ? "false" : "true");
        return bad ? 1 : 0;
    }

    /* a data block: chunk 0 = block id, the others pseudo-random 256-bit values (Client.hpp:367-372), as big-endian scalars */
    static uint8_t block[BLOCK_SIZE];
    uint32_t seed = 12345;
    for (int i = 0; i < BLOCK_SIZE; i++) block[i] = (uint8_t)(lcg(&seed) >> 24);
    memset(block, 0, 32); block[31] = 7;

    uint8_t digest[64], g1[64] = {0}, two_g[64], tmp[64], k32[32];
    GoSlice s_block = slice(block, BLOCK_SIZE), s_digest = slice(digest, 64);
    compute_digest(&s_block, &s_digest);                                /* client: alpha * f(tau) * G, main.go:70-89 */
    g1[31] = 1; g1[63] = 2;                                             /* G = (1, 2) */
    memcpy(two_g, g1, 64);
    GoSlice s_a = slice(two_g, 64), s_b = slice(g1, 64);
    add_point(&s_a, &s_b);                                              /* 2G by addition */
    memcpy(tmp, g1, 64); scalar_set_int(k32, 2);
    GoSlice s_t = slice(tmp, 64), s_k = slice(k32, 32);
    mult_point(&s_t, &s_k);                                             /* 2G by scalar multiplication */
    GoSlice s_two = slice(two_g, 64);
    CHECK(compare_commitment(&s_two, &s_t) == 1, "G + G == 2 * G (add_point / mult_point / compare_commitment)");
    static const uint8_t two_g_x[32] = {0x03, 0x06, 0x44, 0xe7, 0x2e, 0x13, 0x1a, 0x02, 0x9b, 0x85, 0x04, 0x5b, 0x68, 0x18, 0x15, 0x85,
                                        0xd9, 0x78, 0x16, 0xa9, 0x16, 0x87, 0x1c, 0xa8, 0xd3, 0xc2, 0x08, 0xc1, 0x6d, 0x87, 0xcf, 0xd3};
    CHECK(memcmp(two_g, two_g_x, 32) == 0, "2G has the EIP-196 x coordinate");
    memcpy(tmp, two_g, 64);
    neg_point(&s_t);
    add_point(&s_t, &s_two);                                            /* -2G + 2G */
    uint8_t inf[64]; memset(inf, 0xff, 64);
    GoSlice s_inf = slice(inf, 64);
    set_inf_point(&s_inf);
    CHECK(compare_commitment(&s_t, &s_inf) == 1, "P + (-P) is the 64-zero-byte infinity (neg_point / set_inf_point)");
    uint8_t c1[64], c2[64], c3[64], r1[32], r2[32], r3[32];
    scalar_set_int(r1, 1000); scalar_set_int(r2, 234); scalar_set_int(r3, 1234);
    GoSlice s_r1 = slice(r1, 32), s_r2 = slice(r2, 32), s_r3 = slice(r3, 32), s_c1 = slice(c1, 64), s_c2 = slice(c2, 64), s_c3 = slice(c3, 64);
    compute_digest_complement(&s_r1, &s_c1); compute_digest_complement(&s_r2, &s_c2); compute_digest_complement(&s_r3, &s_c3);
    add_point(&s_c1, &s_c2);
    CHECK(compare_commitment(&s_c1, &s_c3) == 1, "compute_digest_complement is linear in its scalar (main.go:91-101)");

    if (gpu) {
        /* server side: SRS from the wire blob (Server.hpp:183-188), commitment from the SRS (Server.hpp:550-558) */
        GoSlice s_blob_in = slice(srs_blob, blob_len);
        init_SRS_from_data(NUM_CHUNKS, &s_blob_in);
        uint8_t commit[64];
        GoSlice s_commit = slice(commit, 64);
        compute_digest_from_srs(&s_block, &s_commit);
        uint8_t alpha32[32] = {0};
        memcpy(alpha32 + 16, SECRET_KEY, 16);
        memcpy(tmp, commit, 64);
        GoSlice s_alpha32 = slice(alpha32, 32);
        mult_point(&s_t, &s_alpha32);
        CHECK(compare_commitment(&s_t, &s_digest) == 1, "compute_digest(f) == alpha * compute_digest_from_srs(f)  (main.go:81-88 vs :114)");

        uint8_t pc[64], ph[64], pp[32], py[32];
        GoSlice s_pc = slice(pc, 64), s_ph = slice(ph, 64), s_pp = slice(pp, 32), s_py = slice(py, 32);
        create_proof(0x1122334455667788ull, &s_block, &s_pc, &s_ph, &s_pp, &s_py);          /* Server.hpp:363-398 */
        CHECK(memcmp(pc, commit, 64) == 0, "create_proof's commitment == compute_digest_from_srs");
        CHECK(verify_proof(&s_pc, &s_ph, &s_pp, &s_py) == 1, "verify_proof accepts the opening (Client.hpp:1637-1662)");
        py[31] ^= 1;
        CHECK(verify_proof(&s_pc, &s_ph, &s_pp, &s_py) == 0, "verify_proof rejects a wrong claim");

        /* audit MSM (Server.hpp:900): 1 408 = NUM_CHECK_AUDIT * height points at N = 2^10, abs(int32) coefficients over
         * repeated MACs; checked against the same sum built from mult_point / add_point */
        enum { NP = 1408, DISTINCT = 11 };
        static uint8_t pts[NP * 64], scs[NP * 32], base[DISTINCT][64];
        for (int d = 0; d < DISTINCT; d++) { memcpy(base[d], g1, 64); GoSlice sb = slice(base[d], 64); scalar_set_int(k32, 1000003u * (d + 1)); mult_point(&sb, &s_k); }
        uint64_t coeff_sum[DISTINCT] = {0};
        for (int i = 0; i < NP; i++) {
            uint32_t cf = lcg(&seed) >> 1;
            int d = i % DISTINCT;
            memcpy(pts + 64 * i, base[d], 64);
            scalar_set_int(scs + 32 * i, cf);
            coeff_sum[d] += cf;
        }
        uint8_t msm[64], want[64];
        GoSlice s_scs = slice(scs, NP * 32), s_pts = slice(pts, NP * 64), s_msm = slice(msm, 64), s_want = slice(want, 64);
        compute_multi_exp(&s_scs, &s_pts, NP, &s_msm);
        set_inf_point(&s_want);
        for (int d = 0; d < DISTINCT; d++) {
            uint8_t k[32] = {0};
            for (int b = 0; b < 8; b++) k[31 - b] = (uint8_t)(coeff_sum[d] >> (8 * b));
            memcpy(tmp, base[d], 64);
            GoSlice sk = slice(k, 32);
            mult_point(&s_t, &sk);
            add_point(&s_want, &s_t);
        }
        CHECK(compare_commitment(&s_msm, &s_want) == 1, "compute_multi_exp over 1 408 audit pairs == sum of mult_point / add_point");
    }
    printf(failures ? "HARNESS FAILED (%d)\n" : "HARNESS OK\n", failures);
    return failures ? 1 : 0;
}
