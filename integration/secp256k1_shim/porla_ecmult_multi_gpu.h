/* porla_ecmult_multi_gpu.h -- the wrapper body of the secp256k1.c include shim (see secp256k1.c beside this file).
 *
 * Defines  static int secp256k1_ecmult_multi_var(...)  with the reference's signature and return convention
 * (porla/Utils/secp256k1_lib/ecmult_impl.h:814-860; ecmult.h:36-47: 1 on success, 0 if the scratch is too small or the
 * callback returns 0; r is overwritten).  It must be included AFTER the vendored internals have been included with
 *     #define secp256k1_ecmult_multi_var secp256k1_ecmult_multi_var_cpu
 * in force, so that the vendored body is available under that name.
 *
 * Dispatch: n >= the threshold and g_sc == 0 (Porla always passes &szero: Client.hpp:395,778; Server.hpp:842,848) -> drain
 * the callback (utils.h:166-178) into canonical encodings and call the engine; everything else -> the vendored CPU body.
 * The staging buffers come from the CALLER'S scratch space (every call site sizes it with
 * secp256k1_pippenger_scratch_size(n, window) >= (2n + 2) * 160 bytes and passes one scratch per pool thread:
 * Client.hpp:119-123,756-758,778; Server.hpp:121-129,838-840), so there is no malloc per call and no shared state between
 * the 8 pool threads; a scratch that cannot hold 96 n bytes sends the call to the CPU body, as the reference falls back to
 * its simple algorithm.  Threshold: PORLA_GPU_MSM_THRESHOLD at compile time (default 64: the vendored path costs ~23 us per
 * point at these sizes, BASELINE.md s2, against one launch-latency-bound engine call), overridable at run time with the
 * environment variable of the same name. */
#ifndef PORLA_ECMULT_MULTI_GPU_H
#define PORLA_ECMULT_MULTI_GPU_H

#ifndef PORLA_GPU_MSM_THRESHOLD
#define PORLA_GPU_MSM_THRESHOLD 64
#endif

#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif
/* include/porla_gpu.h of the engine (libmultiexp.so, already on the reference's link line: porla/Makefile:13) */
int porla_secp256k1_msm_host(const unsigned char *scalars, const unsigned char *points, size_t n, unsigned char out_affine[64]);
const char *porla_gpu_last_error(void);
#ifdef __cplusplus
}
#endif

static size_t porla_gpu_msm_threshold(void) {
    static size_t cached = 0;            /* benign race: every thread computes the same value */
    if (cached == 0) {
        const char *e = getenv("PORLA_GPU_MSM_THRESHOLD");
        long v = e ? atol(e) : (long)PORLA_GPU_MSM_THRESHOLD;
        cached = v < 1 ? 1 : (size_t)v;
    }
    return cached;
}

static int secp256k1_ecmult_multi_var(const secp256k1_callback* error_callback, secp256k1_scratch *scratch,
                                      secp256k1_gej *r, const secp256k1_scalar *inp_g_sc,
                                      secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n) {
    size_t i, checkpoint;
    unsigned char *sc, *pt, out[64];
    secp256k1_fe x, y;
    secp256k1_ge res;
    int ok;
    if (n < porla_gpu_msm_threshold() || scratch == NULL || (inp_g_sc != NULL && !secp256k1_scalar_is_zero(inp_g_sc))) {
        return secp256k1_ecmult_multi_var_cpu(error_callback, scratch, r, inp_g_sc, cb, cbdata, n);
    }
    checkpoint = secp256k1_scratch_checkpoint(error_callback, scratch);
    sc = (unsigned char*)secp256k1_scratch_alloc(error_callback, scratch, 32 * n);
    pt = (unsigned char*)secp256k1_scratch_alloc(error_callback, scratch, 64 * n);
    if (sc == NULL || pt == NULL) {
        secp256k1_scratch_apply_checkpoint(error_callback, scratch, checkpoint);
        return secp256k1_ecmult_multi_var_cpu(error_callback, scratch, r, inp_g_sc, cb, cbdata, n);
    }
    ok = 1;
    for (i = 0; i < n && ok; i++) {              /* drain the callback into canonical encodings */
        secp256k1_scalar s;
        secp256k1_ge p;
        if (!cb(&s, &p, i, cbdata)) { ok = 0; break; }
        secp256k1_scalar_get_b32(sc + 32 * i, &s);
        if (secp256k1_ge_is_infinity(&p)) {
            memset(pt + 64 * i, 0, 64);
        } else {
            secp256k1_fe_normalize_var(&p.x);
            secp256k1_fe_normalize_var(&p.y);
            secp256k1_fe_get_b32(pt + 64 * i, &p.x);
            secp256k1_fe_get_b32(pt + 64 * i + 32, &p.y);
        }
    }
    if (ok && porla_secp256k1_msm_host(sc, pt, n, out) != 0) {
        secp256k1_scratch_apply_checkpoint(error_callback, scratch, checkpoint);
        secp256k1_callback_call(error_callback, porla_gpu_last_error());   /* default: print + abort (util.h:29-52) */
        return 0;
    }
    secp256k1_scratch_apply_checkpoint(error_callback, scratch, checkpoint);
    if (!ok) return 0;
    for (i = 0; i < 64 && out[i] == 0; i++) {}
    if (i == 64) { secp256k1_gej_set_infinity(r); return 1; }
    if (!secp256k1_fe_set_b32(&x, out) || !secp256k1_fe_set_b32(&y, out + 32)) return 0;
    secp256k1_ge_set_xy(&res, &x, &y);
    secp256k1_gej_set_ge(r, &res);               /* z = 1: any Jacobian representative is fine for the callers */
    return 1;
}

#endif /* PORLA_ECMULT_MULTI_GPU_H */
