/* harness.c -- replays the IPA scheme's secp256k1_ecmult_multi_var call sequence of BASELINE config 1 (1024-block file:
 * SURVEY.md s8(d) cfg 1) THROUGH the include shim, the way the reference's translation unit reaches it:
 *     #include "secp256k1.c"        (porla/Utils/utils.h:7, resolved through the include path: the shim comes first)
 * Compiles as C and as C++11.  Built twice by tests/test_ipa_shim.py:
 *   -DPORLA_HARNESS_STUB   porla_secp256k1_msm_host is a counting stub over the CPU oracle: dispatch logic, marshalling,
 *                          scratch handling and the 8-thread call pattern are checked without a GPU
 *   (default)              linked with -lmultiexp like porla/Makefile:13: the same sequence on the MI355X engine
 * The libsecp256k1 internals are the model of secp256k1_internals_model.h (the vendored tree needs the installed public
 * header, absent from this image); results are checked against oracle/_build/liboracle.so (argv[1]). */
#include "secp256k1.c"

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>

typedef void (*oracle_multi_fn)(const unsigned char*, const unsigned char*, size_t, unsigned char*, int, int);
typedef void (*oracle_chain_fn)(size_t, unsigned char*);
static oracle_multi_fn oracle_multi;
static void cpu_msm(const unsigned char *sc, const unsigned char *pt, size_t n, unsigned char out[64]) { oracle_multi(sc, pt, n, out, 1, 0); }

#ifdef PORLA_HARNESS_STUB
static volatile long stub_engine_calls = 0;
#ifdef __cplusplus
extern "C" {
#endif
int porla_secp256k1_msm_host(const unsigned char *scalars, const unsigned char *points, size_t n, unsigned char out_affine[64]) {
    __sync_fetch_and_add(&stub_engine_calls, 1);
    oracle_multi(scalars, points, n, out_affine, 1, 0);
    return 0;
}
const char *porla_gpu_last_error(void) { return "stub"; }
#ifdef __cplusplus
}
#endif
#define ENGINE_CALLS() stub_engine_calls
#else
#define ENGINE_CALLS() (-1L)
#endif

static int failures = 0;
#define CHECK(cond, what) do { if (cond) printf("ok: %s\n", what); else { printf("FAIL: %s\n", what); failures++; } } while (0)

/* the reference's callback over parallel arrays (utils.h:166-171) */
typedef struct { secp256k1_scalar *sc; secp256k1_ge *pt; } ecmult_multi_data;
static int ecmult_multi_callback(secp256k1_scalar *sc, secp256k1_ge *pt, size_t idx, void *cbdata) {
    ecmult_multi_data *data = (ecmult_multi_data*)cbdata;
    *sc = data->sc[idx];
    *pt = data->pt[idx];
    return 1;
}
static int failing_callback(secp256k1_scalar *sc, secp256k1_ge *pt, size_t idx, void *cbdata) {
    if (idx == 70) return 0;
    return ecmult_multi_callback(sc, pt, idx, cbdata);
}
static void on_error(const char *text, void *data) { (void)data; printf("error callback: %s\n", text); failures++; }
static const secp256k1_callback error_callback = { on_error, NULL };

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static void gej_to_bytes(unsigned char out[64], const secp256k1_gej *r) {
    if (r->infinity) { memset(out, 0, 64); return; }
    memcpy(out, r->x.b, 32); memcpy(out + 32, r->y.b, 32);           /* z = 1 from the shim and from the model */
}
static void expect(const secp256k1_scalar *sc, const secp256k1_ge *pt, size_t n, unsigned char out[64]) {
    unsigned char *s = (unsigned char*)malloc(32 * n + 1), *p = (unsigned char*)malloc(64 * n + 1);
    size_t i;
    for (i = 0; i < n; i++) {
        secp256k1_scalar_get_b32(s + 32 * i, &sc[i]);
        if (pt[i].infinity) memset(p + 64 * i, 0, 64);
        else { memcpy(p + 64 * i, pt[i].x.b, 32); memcpy(p + 64 * i + 32, pt[i].y.b, 32); }
    }
    oracle_multi(s, p, n, out, 1, 0);
    free(s); free(p);
}
/* sum of affine parts (the callers fold their threads' parts with secp256k1_gej_add_var, Client.hpp:402-404,783-785) */
static void sum_parts(const secp256k1_gej *parts, size_t count, unsigned char out[64]) {
    unsigned char sc[32 * 8], pt[64 * 8];
    size_t i;
    memset(sc, 0, sizeof sc);
    for (i = 0; i < count; i++) { sc[32 * i + 31] = 1; gej_to_bytes(pt + 64 * i, &parts[i]); }
    oracle_multi(sc, pt, count, out, 1, 0);
}

typedef struct { secp256k1_scratch *scratch; secp256k1_gej *out; ecmult_multi_data data; size_t n; int rc; } job;
static secp256k1_scalar szero;
static void *run_job(void *arg) {
    job *j = (job*)arg;
    j->rc = secp256k1_ecmult_multi_var(&error_callback, j->scratch, j->out, &szero, ecmult_multi_callback, &j->data, j->n);
    return NULL;
}
/* 8 pool threads, each its range and its own scratch (Client.hpp:376-400, 761-782) */
static int threaded(secp256k1_scalar *sc, secp256k1_ge *pt, size_t n_points, secp256k1_gej parts[8]) {
    pthread_t th[8];
    job jobs[8];
    size_t each = n_points / 8, start = 0;
    int t, ok = 1;
    for (t = 0; t < 8; t++) {
        size_t n = t == 7 ? n_points - each * 7 : each;
        int bucket_window = secp256k1_pippenger_bucket_window(n);
        size_t scratch_size = secp256k1_pippenger_scratch_size(n, bucket_window);
        jobs[t].scratch = secp256k1_scratch_create(&error_callback, scratch_size + PIPPENGER_SCRATCH_OBJECTS * ALIGNMENT);
        jobs[t].out = &parts[t];
        jobs[t].data.sc = sc + start; jobs[t].data.pt = pt + start;
        jobs[t].n = n;
        start += n;
    }
    for (t = 0; t < 8; t++) pthread_create(&th[t], NULL, run_job, &jobs[t]);
    for (t = 0; t < 8; t++) { pthread_join(th[t], NULL); ok &= jobs[t].rc; secp256k1_scratch_destroy(&error_callback, jobs[t].scratch); }
    return ok;
}

int main(int argc, char **argv) {
    enum { NPTS = 1408 };
    void *lib;
    oracle_chain_fn chain;
    unsigned char *raw, got[64], want[64];
    secp256k1_ge *pt;
    secp256k1_scalar *sc_full, *sc_audit;
    secp256k1_gej parts[8], r;
    secp256k1_scratch *scratch;
    ecmult_multi_data data;
    size_t i;
    long cpu0, eng0;
    int rc, bucket_window;
    size_t scratch_size;
    if (argc < 2) { fprintf(stderr, "usage: %s path/to/liboracle.so\n", argv[0]); return 2; }
    lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    oracle_multi = (oracle_multi_fn)dlsym(lib, "oracle_secp256k1_multi");
    chain = (oracle_chain_fn)dlsym(lib, "oracle_secp256k1_doubling_chain");
    if (!oracle_multi || !chain) { fprintf(stderr, "oracle symbols missing\n"); return 2; }
    model_cpu_msm = cpu_msm;
    secp256k1_scalar_set_int(&szero, 0);

    raw = (unsigned char*)malloc(64 * NPTS);
    chain(NPTS, raw);                                              /* P_i = 2^i G, as bench_ecmult.c:328-337 */
    pt = (secp256k1_ge*)malloc(sizeof(secp256k1_ge) * NPTS);
    sc_full = (secp256k1_scalar*)malloc(sizeof(secp256k1_scalar) * NPTS);
    sc_audit = (secp256k1_scalar*)malloc(sizeof(secp256k1_scalar) * NPTS);
    for (i = 0; i < NPTS; i++) {
        secp256k1_fe x, y;
        secp256k1_fe_set_b32(&x, raw + 64 * i); secp256k1_fe_set_b32(&y, raw + 64 * i + 32);
        secp256k1_ge_set_xy(&pt[i], &x, &y);
        sc_full[i].d[0] = rng(); sc_full[i].d[1] = rng(); sc_full[i].d[2] = rng(); sc_full[i].d[3] = rng() >> 1;   /* data chunks < 2^255 */
        secp256k1_scalar_set_int(&sc_audit[i], (unsigned int)(rng() & 0x7fffffffu));                                 /* abs(int32), prg.h:84-97 */
    }

    /* 1. compute_commitment: 128 generators as 8 x 16 points from 8 threads (Client.hpp:374-406) -- below the threshold */
    cpu0 = model_cpu_calls; eng0 = ENGINE_CALLS();
    rc = threaded(sc_full, pt, 128, parts);
    sum_parts(parts, 8, got); expect(sc_full, pt, 128, want);
    CHECK(rc == 1 && memcmp(got, want, 64) == 0, "compute_commitment: 8 x 16 points, sum of parts = 128-point commitment");
    CHECK(model_cpu_calls - cpu0 == 8, "16-point calls stay on the vendored CPU body");
#ifdef PORLA_HARNESS_STUB
    CHECK(ENGINE_CALLS() - eng0 == 0, "... and never reach the engine");
#endif

    /* 2. client-side audit: n_points = 1408 as 8 x 176 from 8 threads, abs(int32) coefficients (Client.hpp:756-787) */
    cpu0 = model_cpu_calls; eng0 = ENGINE_CALLS();
    rc = threaded(sc_audit, pt, NPTS, parts);
    sum_parts(parts, 8, got); expect(sc_audit, pt, NPTS, want);
    CHECK(rc == 1 && memcmp(got, want, 64) == 0, "client audit: 8 concurrent 176-point calls through the engine, sum = 1408-point MSM");
    CHECK(model_cpu_calls - cpu0 == 0, "176-point calls leave the CPU body alone");
#ifdef PORLA_HARNESS_STUB
    CHECK(ENGINE_CALLS() - eng0 == 8, "... 8 engine calls");
#endif

    /* 3. server-side audit: two 1408-point MSMs on one scratch from the main thread (Server.hpp:838-848) */
    bucket_window = secp256k1_pippenger_bucket_window(NPTS);
    scratch_size = secp256k1_pippenger_scratch_size(NPTS, bucket_window);
    scratch = secp256k1_scratch_create(&error_callback, scratch_size + PIPPENGER_SCRATCH_OBJECTS * ALIGNMENT);
    data.sc = sc_audit; data.pt = pt;
    cpu0 = model_cpu_calls;
    rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &szero, ecmult_multi_callback, &data, NPTS);
    gej_to_bytes(got, &r);
    CHECK(rc == 1 && memcmp(got, want, 64) == 0, "server audit: 1408-point MSM (MACs)");
    data.sc = sc_full;
    rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &szero, ecmult_multi_callback, &data, NPTS);
    gej_to_bytes(got, &r); expect(sc_full, pt, NPTS, want);
    CHECK(rc == 1 && memcmp(got, want, 64) == 0, "server audit: second 1408-point MSM on the same scratch, full-width scalars");
    CHECK(model_cpu_calls - cpu0 == 0 && secp256k1_scratch_checkpoint(&error_callback, scratch) == 0, "engine route, scratch handed back");

    /* 4. the small MSMs around the proof (2 points each): CPU body */
    cpu0 = model_cpu_calls;
    for (i = 0; i < 4; i++) {
        data.sc = sc_full + 2 * i; data.pt = pt + 2 * i;
        rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &szero, ecmult_multi_callback, &data, 2);
        gej_to_bytes(got, &r); expect(sc_full + 2 * i, pt + 2 * i, 2, want);
        if (!(rc == 1 && memcmp(got, want, 64) == 0)) failures++;
    }
    CHECK(model_cpu_calls - cpu0 == 4, "4 x 2-point calls: CPU body");

    /* 5. edges of the dispatch */
    {
        secp256k1_scalar one;
        secp256k1_scratch *tiny = secp256k1_scratch_create(&error_callback, 1024);
        secp256k1_scalar *zeros = (secp256k1_scalar*)calloc(200, sizeof(secp256k1_scalar));
        secp256k1_scalar_set_int(&one, 1);
        data.sc = sc_audit; data.pt = pt;
        cpu0 = model_cpu_calls;
        rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &one, ecmult_multi_callback, &data, 200);
        CHECK(rc == 1 && model_cpu_calls - cpu0 == 1, "g_sc != 0 -> CPU body");
        rc = secp256k1_ecmult_multi_var(&error_callback, NULL, &r, &szero, ecmult_multi_callback, &data, 200);
        CHECK(rc == 1 && model_cpu_calls - cpu0 == 2, "no scratch -> CPU body");
        rc = secp256k1_ecmult_multi_var(&error_callback, tiny, &r, &szero, ecmult_multi_callback, &data, 200);
        gej_to_bytes(got, &r); expect(sc_audit, pt, 200, want);
        CHECK(rc == 1 && model_cpu_calls - cpu0 == 3 && memcmp(got, want, 64) == 0 && secp256k1_scratch_checkpoint(&error_callback, tiny) == 0,
              "scratch too small for the staging buffers -> CPU body, scratch untouched");
        rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &szero, failing_callback, &data, 200);
        CHECK(rc == 0 && secp256k1_scratch_checkpoint(&error_callback, scratch) == 0, "callback failure -> 0, scratch handed back");
        data.sc = zeros;
        rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &szero, ecmult_multi_callback, &data, 200);
        CHECK(rc == 1 && r.infinity == 1, "all-zero scalars through the engine -> infinity");
        pt[5].infinity = 1;
        data.sc = sc_audit;
        rc = secp256k1_ecmult_multi_var(&error_callback, scratch, &r, &szero, ecmult_multi_callback, &data, 200);
        gej_to_bytes(got, &r); expect(sc_audit, pt, 200, want);
        CHECK(rc == 1 && memcmp(got, want, 64) == 0, "an infinity point among the inputs");
        pt[5].infinity = 0;
        free(zeros);
        secp256k1_scratch_destroy(&error_callback, tiny);
    }
    secp256k1_scratch_destroy(&error_callback, scratch);
    free(raw); free(pt); free(sc_full); free(sc_audit);
    printf(failures ? "HARNESS FAILED (%d)\n" : "HARNESS OK\n", failures);
    return failures ? 1 : 0;
}
