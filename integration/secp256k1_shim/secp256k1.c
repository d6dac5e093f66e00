/* secp256k1.c -- include-path shim that routes Porla's IPA multi-scalar multiplications to the MI355X engine.
 *
 * The reference unity-builds libsecp256k1's internals into its C++ translation unit with
 *     #include "secp256k1.c"                       (porla/Utils/utils.h:7)
 * resolved through INCLUDE_PATH = -I. -I../ -I../Utils/secp256k1_lib -I/usr/local/include (porla/Makefile:3).
 * Put the directory holding THIS file ahead of -I../Utils/secp256k1_lib and keep -lmultiexp (already
 * there) -- Server.hpp / Client.hpp / utils.h stay untouched.  Every static helper the callers use directly
 * (secp256k1_pippenger_bucket_window, _scratch_size, secp256k1_scratch_create, secp256k1_ecmult,
 * secp256k1_ecmult_const, ...) remains the vendored one; only secp256k1_ecmult_multi_var
 * (porla/Utils/secp256k1_lib/ecmult_impl.h:814-860) is wrapped: the vendored body is kept under the name
 * secp256k1_ecmult_multi_var_cpu and used below the threshold (porla_ecmult_multi_gpu.h).
 *
 * The vendored unity file needs the INSTALLED public header (secp256k1.c:9 -> /usr/local/include/secp256k1.h), which this
 * repository's build image does not have: tests/test_ipa_shim.py therefore compiles the same rename + wrapper around the
 * vendored INTERNAL headers (as C and as C++11, include order of porla/Makefile:3) and replays the reference's call sequence
 * through it; this three-line file is exercised verbatim only where libsecp256k1 is installed.
 */
#define secp256k1_ecmult_multi_var secp256k1_ecmult_multi_var_cpu
#ifdef PORLA_VENDORED_SECP256K1_C
#include PORLA_VENDORED_SECP256K1_C
#else
#include "../../Utils/secp256k1_lib/secp256k1.c"   /* the real vendored unity file, by explicit path */
#endif
#undef secp256k1_ecmult_multi_var

#include "porla_ecmult_multi_gpu.h"
