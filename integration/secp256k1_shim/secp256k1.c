/* secp256k1.c -- include-path shim that routes Porla's IPA multi-scalar multiplications to the MI355X engine.
 *
 * The reference unity-builds libsecp256k1's internals into its C++ translation unit with
 *     #include "secp256k1.c"                       (porla/Utils/utils.h:7)
 * resolved through INCLUDE_PATH = -I. -I../ -I../Utils/secp256k1_lib -I/usr/local/include (porla/Makefile:3).
 * Put the directory holding THIS file ahead of -I../Utils/secp256k1_lib and add -lmultiexp (already there) --
 * Server.hpp / Client.hpp / utils.h stay untouched.  Every static helper the callers use directly
 * (secp256k1_pippenger_bucket_window, _scratch_size, secp256k1_scratch_create, secp256k1_ecmult,
 * secp256k1_ecmult_const, ...) remains the vendored one; only secp256k1_ecmult_multi_var
 * (porla/Utils/secp256k1_lib/ecmult_impl.h:814-860) is wrapped: the vendored body is kept under the name
 * secp256k1_ecmult_multi_var_cpu and used below PORLA_GPU_MSM_THRESHOLD points (the reference's own call sites
 * pass 8, 16, n/8 and n <= 1408 points: Client.hpp:395,778,1614; Server.hpp:349,509,842-848,2360,2410).
 *
 * This file is integration glue for the reference's build; it is not compiled in this repository (the vendored
 * tree needs the installed public header secp256k1.h, absent here).
 */
#ifndef PORLA_GPU_MSM_THRESHOLD
#define PORLA_GPU_MSM_THRESHOLD 1024
#endif

#define secp256k1_ecmult_multi_var secp256k1_ecmult_multi_var_cpu
#include "../../Utils/secp256k1_lib/secp256k1.c"   /* the real vendored unity file, by explicit path */
#undef secp256k1_ecmult_multi_var

#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif
/* include/porla_gpu.h of the engine */
int porla_secp256k1_msm_host(const unsigned char *scalars, const unsigned char *points, size_t n, unsigned char out_affine[64]);
const char *porla_gpu_last_error(void);
#ifdef __cplusplus
}
#endif

static int secp256k1_ecmult_multi_var(const secp256k1_callback* error_callback, secp256k1_scratch *scratch,
                                      secp256k1_gej *r, const secp256k1_scalar *inp_g_sc,
                                      secp256k1_ecmult_multi_callback cb, void *cbdata, size_t n) {
    size_t i;
    unsigned char *sc, *pt, out[64];
    secp256k1_fe x, y;
    secp256k1_ge res;
    if (n < PORLA_GPU_MSM_THRESHOLD || (inp_g_sc != NULL && !secp256k1_scalar_is_zero(inp_g_sc))) {
        return secp256k1_ecmult_multi_var_cpu(error_callback, scratch, r, inp_g_sc, cb, cbdata, n);
    }
    sc = (unsigned char*)malloc(32 * n);
    pt = (unsigned char*)malloc(64 * n);
    if (sc == NULL || pt == NULL) { free(sc); free(pt); return 0; }
    for (i = 0; i < n; i++) {                    /* drain the callback (utils.h:166-178) into canonical encodings */
        secp256k1_scalar s;
        secp256k1_ge p;
        if (!cb(&s, &p, i, cbdata)) { free(sc); free(pt); return 0; }
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
    if (porla_secp256k1_msm_host(sc, pt, n, out) != 0) {
        secp256k1_callback_call(error_callback, porla_gpu_last_error());   /* default: print + abort (util.h:29-52) */
        free(sc); free(pt);
        return 0;
    }
    free(sc); free(pt);
    for (i = 0; i < 64 && out[i] == 0; i++) {}
    if (i == 64) { secp256k1_gej_set_infinity(r); return 1; }
    if (!secp256k1_fe_set_b32(&x, out) || !secp256k1_fe_set_b32(&y, out + 32)) return 0;
    secp256k1_ge_set_xy(&res, &x, &y);
    secp256k1_gej_set_ge(r, &res);               /* z = 1: any Jacobian representative is fine for the callers */
    return 1;
}
