/* secp256k1_internals_model.h -- a MODEL of the few libsecp256k1 internals that the include shim's wrapper touches
 * (integration/secp256k1_shim/porla_ecmult_multi_gpu.h), written for this repository's tests.
 *
 * Why a model: the vendored unity file (porla/Utils/secp256k1_lib/secp256k1.c) and its table file include the INSTALLED public
 * header ../include/secp256k1.h, which this build image lacks, so the vendored tree cannot be compiled or linked here.  The
 * model keeps the NAMES, SIGNATURES and CALL CONTRACTS the wrapper relies on (scalar = 4 x uint64 little-endian limbs,
 * scalar_4x64.h:13-15; ge = {x, y, infinity}, group.h:13-17; scratch checkpoint / alloc / apply_checkpoint, scratch.h:24-40;
 * callback, util.h:21-52; the multi-callback type, ecmult.h:29-35) with trivially simple representations (field elements are
 * 32 big-endian bytes).  Its "vendored" secp256k1_ecmult_multi_var body drains the callback and hands the pairs to a
 * function pointer the harness sets (the CPU oracle), counting how often the CPU route was taken.  Nothing of this ships. */
#ifndef PORLA_SECP256K1_INTERNALS_MODEL_H
#define PORLA_SECP256K1_INTERNALS_MODEL_H
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { void (*fn)(const char *text, void *data); const void *data; } secp256k1_callback;
static void secp256k1_callback_call(const secp256k1_callback *cb, const char *text) { cb->fn(text, (void*)cb->data); }

typedef struct { uint64_t d[4]; } secp256k1_scalar;
typedef struct { unsigned char b[32]; int normalized; } secp256k1_fe;
typedef struct { secp256k1_fe x, y; int infinity; } secp256k1_ge;
typedef struct { secp256k1_fe x, y, z; int infinity; } secp256k1_gej;
typedef struct { unsigned char magic[8]; void *data; size_t alloc_size; size_t max_size; } secp256k1_scratch;
typedef int (secp256k1_ecmult_multi_callback)(secp256k1_scalar *sc, secp256k1_ge *pt, size_t idx, void *data);

#define PIPPENGER_SCRATCH_OBJECTS 6
#define ALIGNMENT 16

static void secp256k1_scalar_get_b32(unsigned char *bin, const secp256k1_scalar *a) {
    int i, j;
    for (i = 0; i < 4; i++) for (j = 0; j < 8; j++) bin[8 * i + j] = (unsigned char)(a->d[3 - i] >> (56 - 8 * j));
}
static void secp256k1_scalar_set_b32(secp256k1_scalar *r, const unsigned char *bin) {
    int i, j;
    for (i = 0; i < 4; i++) { r->d[3 - i] = 0; for (j = 0; j < 8; j++) r->d[3 - i] = (r->d[3 - i] << 8) | bin[8 * i + j]; }
}
static void secp256k1_scalar_set_int(secp256k1_scalar *r, unsigned int v) { r->d[0] = v; r->d[1] = r->d[2] = r->d[3] = 0; }
static int secp256k1_scalar_is_zero(const secp256k1_scalar *a) { return (a->d[0] | a->d[1] | a->d[2] | a->d[3]) == 0; }
static void secp256k1_fe_normalize_var(secp256k1_fe *r) { r->normalized = 1; }
static void secp256k1_fe_get_b32(unsigned char *r, const secp256k1_fe *a) { memcpy(r, a->b, 32); }
static int secp256k1_fe_set_b32(secp256k1_fe *r, const unsigned char *a) { memcpy(r->b, a, 32); r->normalized = 1; return 1; }
static int secp256k1_ge_is_infinity(const secp256k1_ge *a) { return a->infinity; }
static void secp256k1_ge_set_xy(secp256k1_ge *r, const secp256k1_fe *x, const secp256k1_fe *y) { r->infinity = 0; r->x = *x; r->y = *y; }
static void secp256k1_gej_set_infinity(secp256k1_gej *r) { memset(r, 0, sizeof *r); r->infinity = 1; }
static void secp256k1_gej_set_ge(secp256k1_gej *r, const secp256k1_ge *a) {
    memset(r, 0, sizeof *r);
    r->infinity = a->infinity; r->x = a->x; r->y = a->y; r->z.b[31] = 1;
}

static secp256k1_scratch *secp256k1_scratch_create(const secp256k1_callback *error_callback, size_t size) {
    secp256k1_scratch *s = (secp256k1_scratch*)malloc(sizeof *s + size);
    (void)error_callback;
    if (s) { memcpy(s->magic, "scratch", 8); s->data = (void*)(s + 1); s->alloc_size = 0; s->max_size = size; }
    return s;
}
static void secp256k1_scratch_destroy(const secp256k1_callback *error_callback, secp256k1_scratch *s) { (void)error_callback; free(s); }
static size_t secp256k1_scratch_checkpoint(const secp256k1_callback *error_callback, const secp256k1_scratch *s) { (void)error_callback; return s->alloc_size; }
static void secp256k1_scratch_apply_checkpoint(const secp256k1_callback *error_callback, secp256k1_scratch *s, size_t checkpoint) {
    if (checkpoint > s->alloc_size) { secp256k1_callback_call(error_callback, "invalid checkpoint"); return; }
    s->alloc_size = checkpoint;
}
static void *secp256k1_scratch_alloc(const secp256k1_callback *error_callback, secp256k1_scratch *s, size_t size) {
    void *ret;
    size_t rounded = (size + ALIGNMENT - 1) / ALIGNMENT * ALIGNMENT;
    (void)error_callback;
    if (rounded > s->max_size - s->alloc_size) return NULL;
    ret = (void*)((char*)s->data + s->alloc_size);
    s->alloc_size += rounded;
    return ret;
}

/* the sizing helpers the callers invoke directly (Client.hpp:119-123,756-758; Server.hpp:121-129,838-840); the model keeps
 * the reference's shape: scratch_size >= (2n + 2) entries of >= 160 bytes */
static int secp256k1_pippenger_bucket_window(size_t n) { return n <= 1 ? 1 : n <= 4 ? 2 : n <= 20 ? 3 : n <= 57 ? 4 : n <= 136 ? 5 : n <= 235 ? 6 : n <= 1260 ? 7 : n <= 4420 ? 9 : 12; }
static size_t secp256k1_pippenger_scratch_size(size_t n_points, int bucket_window) {
    return ((size_t)128 << bucket_window) + 64 + (2 * n_points + 2) * 160;
}

/* the CPU route: set by the harness (the oracle's MSM on canonical encodings); counts its uses */
typedef void (*model_cpu_msm_fn)(const unsigned char *scalars, const unsigned char *points, size_t n, unsigned char out[64]);
static model_cpu_msm_fn model_cpu_msm = NULL;
static volatile long model_cpu_calls = 0;

static int secp256k1_ecmult_multi_var(const secp256k1_callback *error_callback, secp256k1_scratch *scratch, secp256k1_gej *r,
                                      const secp256k1_scalar *inp_g_sc, secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n) {
    unsigned char *sc = (unsigned char*)malloc(32 * n + 1), *pt = (unsigned char*)malloc(64 * n + 1), out[64];
    size_t i;
    secp256k1_ge res;
    (void)error_callback; (void)scratch; (void)inp_g_sc;
    __sync_fetch_and_add(&model_cpu_calls, 1);
    for (i = 0; i < n; i++) {
        secp256k1_scalar s;
        secp256k1_ge p;
        if (!cb(&s, &p, i, cbdata)) { free(sc); free(pt); return 0; }
        secp256k1_scalar_get_b32(sc + 32 * i, &s);
        if (p.infinity) memset(pt + 64 * i, 0, 64);
        else { memcpy(pt + 64 * i, p.x.b, 32); memcpy(pt + 64 * i + 32, p.y.b, 32); }
    }
    model_cpu_msm(sc, pt, n, out);
    free(sc); free(pt);
    for (i = 0; i < 64 && out[i] == 0; i++) {}
    if (i == 64) { secp256k1_gej_set_infinity(r); return 1; }
    res.infinity = 0; memcpy(res.x.b, out, 32); memcpy(res.y.b, out + 32, 32);
    secp256k1_gej_set_ge(r, &res);
    return 1;
}
#endif
