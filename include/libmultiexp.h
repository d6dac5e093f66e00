/* libmultiexp.h -- the KZG plug-in boundary of Porla, served by the MI355X engine.
 *
 * These are exactly the 14 unmangled symbols the reference's cgo build exports
 * (reference: porla/Utils/libmultiexp.h:71-84, generated from porla/main.go:31-230) with the
 * same GoSlice ABI (porla/Utils/libmultiexp.h:61: {void* data; GoInt len; GoInt cap},
 * GoInt = long long), so the unmodified C++ Server/Client (wrappers porla/Utils/utils.h:235-305,
 * direct calls Client.hpp:159-167,348-354,411-419,445-453,1637-1662, Server.hpp:183-188,365-397,
 * 550-558) link against this library with `-lmultiexp` exactly as before.
 *
 * Semantics kept from main.go: the caller owns every buffer; the callee reads `len` bytes of the
 * inputs and writes at most `len` bytes of the outputs (Go copy()); add/mult/neg/set_inf work in
 * place; there is no error channel (compare_commitment / verify_proof return 0/1 and print one
 * line on failure); add_point, mult_point, neg_point, set_inf_point and compute_digest_from_srs
 * may be called concurrently (Server.hpp:1054-1078, 1530-1535, 1600-1608).
 *
 * Difference: the multi-scalar multiplications (compute_multi_exp, compute_digest_from_srs,
 * create_proof) run as HIP kernels on the current MI355X device; if no gfx950 device or code
 * object is available they print an error and abort() -- there is no CPU fallback.
 */
#ifndef PORLA_LIBMULTIEXP_H
#define PORLA_LIBMULTIEXP_H
#include <stddef.h>
#include <stdint.h>

#ifndef GO_CGO_PROLOGUE_H   /* do not clash with the cgo-generated header if both are included */
#define GO_CGO_PROLOGUE_H
typedef long long GoInt64;
typedef unsigned long long GoUint64;
typedef unsigned char GoUint8;
typedef GoInt64 GoInt;
typedef struct { void *data; GoInt len; GoInt cap; } GoSlice;
#endif

#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the engine itself is built with -fvisibility=hidden: only this ABI is exported */
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* main.go:31-40   (libmultiexp.h:71) */
extern void init_key(GoSlice* tau_key_in, GoSlice* alpha_key_in);
/* main.go:42-60   (libmultiexp.h:72): SRS.G1[i] = tau^i G; writes the 32*n+132-byte wire blob */
extern void init_SRS(GoInt SRS_size, GoSlice* out, GoInt64* out_len);
/* main.go:62-68   (libmultiexp.h:73) */
extern void init_SRS_from_data(GoInt SRS_size, GoSlice* in);
/* main.go:70-89   (libmultiexp.h:74): alpha * f(tau) * G */
extern void compute_digest(GoSlice* data_in, GoSlice* data_out);
/* main.go:91-101  (libmultiexp.h:75) */
extern void compute_digest_complement(GoSlice* data_in, GoSlice* data_out);
/* main.go:103-116 (libmultiexp.h:76): kzg.Commit -- n_samples-point MSM against the resident SRS */
extern void compute_digest_from_srs(GoSlice* data_in, GoSlice* data_out);
/* main.go:118-138 (libmultiexp.h:77): sum (s_i mod r) * P_i */
extern void compute_multi_exp(GoSlice* scalars, GoSlice* points, GoInt length, GoSlice* result_out);
/* main.go:140-151 (libmultiexp.h:78) */
extern GoUint8 compare_commitment(GoSlice* commitment_a, GoSlice* commitment_b);
/* main.go:153-175 (libmultiexp.h:79) */
extern void create_proof(GoUint64 random_point, GoSlice* data_in, GoSlice* commitment_out, GoSlice* proof_H,
                         GoSlice* proof_point, GoSlice* proof_claim);
/* main.go:177-193 (libmultiexp.h:80) */
extern GoUint8 verify_proof(GoSlice* commitment_in, GoSlice* proof_H, GoSlice* proof_point, GoSlice* proof_claim);
/* main.go:195-202 (libmultiexp.h:81) */
extern void add_point(GoSlice* point_a, GoSlice* point_b);
/* main.go:204-214 (libmultiexp.h:82) */
extern void mult_point(GoSlice* point_a, GoSlice* scalar);
/* main.go:216-222 (libmultiexp.h:83) */
extern void neg_point(GoSlice* point);
/* main.go:224-230 (libmultiexp.h:84) */
extern void set_inf_point(GoSlice* point);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif
