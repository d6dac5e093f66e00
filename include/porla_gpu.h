/* porla_gpu.h -- C ABI of the MI355X engine beyond the 14 cgo symbols of libmultiexp.h.
 *
 * Plain pointers and sizes only.  "d_" pointers are device (HBM) addresses on the current HIP
 * device; `hip_stream` is a hipStream_t passed as void* (NULL = default stream).  All functions
 * return 0 on success and a negative code on failure (porla_gpu_last_error() gives the text);
 * nothing here ever falls back to a CPU implementation of the hot path.
 *
 * What each entry point replaces in the reference:
 *   porla_bn254_msm_*        compute_multi_exp, porla/main.go:118-138 (libmultiexp.h:77), for callers that
 *                            keep the (scalar, point) arrays resident in HBM (bench, multi-GPU sharding)
 *   porla_bn254_jac_sum      the partial-sum fold of range-sharded MSMs; the reference does the same fold
 *                            across its 8 pool threads with gej_add_var, porla/Client/Client.hpp:761-787
 *   porla_kzg_commit_batch_* compute_digest_from_srs (main.go:103-116) hoisted over many rows
 *                            (callers Server.hpp:550-560, 1077-1078, 2061-2062); uses the SRS loaded by
 *                            init_SRS / init_SRS_from_data of libmultiexp.h
 *   porla_fixed_base_*       the same batched commitment against any fixed base: the IPA twin is the Pedersen
 *                            commitment over the fixed generators[] (compute_commitment, Client.hpp:374-406,
 *                            Server.hpp:329-361 -> secp256k1_ecmult_multi_var over 128 fixed points)
 *   porla_secp256k1_msm_*    secp256k1_ecmult_multi_var with g_sc = 0,
 *                            porla/Utils/secp256k1_lib/ecmult_impl.h:814-860 (call sites Server.hpp:842-848,
 *                            Client.hpp:395,778), reached through the include shim in INTEGRATION.md
 *   porla_icc_encode_*       the CRebuild_Cached butterfly network, porla/Server/Server.hpp:1548-1687, and the
 *                            align_MAC scalar derivation, Server.hpp:531-541 (no function boundary exists
 *                            in the reference: INTEGRATION.md documents the patch site)
 *   porla_icc_mac_encode_*   the MAC halves of the same network ("FFT in the exponent"), Server.hpp:1590-1609, 1658-1676
 *   porla_icc_mix_*, porla_icc_mac_mix_*, porla_server_mix_device
 *                            Server::mix, Server.hpp:1209-1328 (data rows, MAC commitments, MAC alignments; the last: all three)
 *   porla_audit_combine_device, porla_*_msm_pair_*, porla_*_audit_msm_pair_*, porla_kzg_audit_device, porla_ipa_audit_device
 *                            Server::audit after the challenge, Server.hpp:790-907: the row combine (:790-828), the two MSMs over
 *                            one scalar array (:842-848 / :900-901), and -- the last two -- the whole audit in one call
 *   porla_kzg_{digest,complement,mac}_batch_{device,host}
 *                            compute_digest / compute_digest_complement hoisted over the blocks of Client::initialize,
 *                            Client.hpp:408-455; the last: both and the add_point that joins them (Client.hpp:229-236, 468-478)
 *
 * Byte formats (identical to the reference's wire formats):
 *   scalar   32 bytes big-endian (bn254_scalar, utils.h:64,307-318); reduced mod the group order
 *   point    64 bytes X||Y big-endian, regular (non-Montgomery) form; 64 zero bytes = infinity
 *            (KZG MAC_Block, utils.h:65; config.hpp:26)
 *   jacobian 96 bytes X||Y||Z big-endian regular form, x = X/Z^2, y = Y/Z^3, Z = 0 = infinity
 */
#ifndef PORLA_GPU_H
#define PORLA_GPU_H
#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the engine itself is built with -fvisibility=hidden: only this ABI is exported */
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define PORLA_OK            0
#define PORLA_ERR_NO_DEVICE (-1)   /* no gfx950 device / HIP runtime failure */
#define PORLA_ERR_HIP       (-2)   /* a HIP call failed (see porla_gpu_last_error) */
#define PORLA_ERR_ARG       (-3)   /* bad argument */
#define PORLA_ERR_STATE     (-4)   /* e.g. SRS not initialised */

#define PORLA_JACOBIAN_BYTES 96    /* X||Y||Z big-endian, the partial sum of one pair range */
#define PORLA_DIST_ID_BYTES  128   /* an ncclUniqueId */

/* ---- runtime ---- */
int         porla_gpu_device_count(void);
int         porla_gpu_set_device(int device);
const char *porla_gpu_last_error(void);
/* the range rule of every split in the engine (MSM pair ranges, commitment rows, ICC columns): shard `rank` of `world` owns
 * units [rank n / world, (rank + 1) n / world).  One process per GPU: each rank calls the ordinary entry points on its own
 * range (rows, columns) -- those paths need no collective. */
int         porla_shard_range(size_t n, int rank, int world, size_t *begin, size_t *end);
/* Per-kernel timing with HIP events recorded on the launch stream.  enable=1 starts (and clears) the
 * accumulation for every kernel, enable=2 for each workload's dominant kernel only (two event packets per recorded
 * kernel leave the GPU idle for ~10 us around it -- bench.py times with 2 and takes the full breakdown separately); porla_gpu_profile_get(i, ...) returns kernel name, summed milliseconds and launch count
 * for slot i, or a negative value past the last slot. */
int         porla_gpu_profile_enable(int enable);
int         porla_gpu_profile_get(int slot, char *name, size_t name_cap, double *total_ms, long long *launches);
/* frees the MSM scratch of every workspace slot of every device (reallocated by the next MSM); PORLA_ERR_STATE while a
 * two-phase MSM is pending */
int         porla_gpu_release_msm_workspaces(void);
/* MSM tuning override (0 = automatic): window bits c */
int         porla_gpu_set_msm_window(int c);
/* inputs of up to 32768 pairs (every MSM the reference issues: n_points <= 3200, Server.hpp:585-587) take a single-launch path;
 * on = 0 sends them through the general path instead, window_bits in 2..8 fixes its window width (0 = automatic) */
int         porla_gpu_set_msm_small(int on, int window_bits);
/* 1: split every scalar with the curve endomorphism (half the windows); 0: plain windows over the full scalar;
 * -1 (default): per curve -- secp256k1 on (as the reference does, ecmult_impl.h:621-634), BN254 off (measured slower) */
int         porla_gpu_set_msm_glv(int on);
/* diagnostic: window bits, window count and GLV flag of the most recently launched MSM (what the automatic choice was) */
int         porla_gpu_last_msm_shape(int *c, int *windows, int *glv);
/* diagnostic: HOST execution of the 8 x 32-bit field helpers the kernels are built from (borrow / carry chains, negation, the
 * bounded reduction of the ICC finish step), for the CPU test suite.  op 0: reduce a value below 5 x modulus (modulus 0 = BN254
 * group order) or below the modulus + 2^256 - n (modulus 1 = secp256k1 group order: one conditional subtraction) to its residue;
 * 1: negate; 2: conditional negate (taken); 3: a - b; 4: a + b (operands canonical).  32-byte little-endian values. */
int         porla_diag_fe_op(int op, int modulus, const uint8_t a_le[32], const uint8_t b_le[32], uint8_t out_le[32]);
/* diagnostic: the scalar split the digit kernel applies (host execution of the same code): scalar mod n = k1 + lambda*k2,
 * magnitudes as 16-byte big-endian, signs as 0/1.  curve: 0 = BN254, 1 = secp256k1. */
int         porla_glv_split(int curve, const uint8_t scalar_be[32], uint8_t k1_mag_be[16], int *k1_neg,
                            uint8_t k2_mag_be[16], int *k2_neg);

/* ---- BN254 G1 MSM ---- */
int porla_bn254_msm_device(const void *d_scalars, const void *d_points, size_t n, uint8_t out_affine[64],
                           void *hip_stream);
int porla_bn254_msm_device_partial(const void *d_scalars, const void *d_points, size_t n, uint8_t out_jacobian[96],
                                   void *hip_stream);
/* The audit's pair: ONE scalar array over TWO point arrays -- replaces the two back-to-back calls
 *   bn254_multi_exp(combined_MAC, ptc, sc, n); bn254_multi_exp(combined_align, pta, sc, n);    (porla/Server/Server.hpp:900-901,
 *   and the secp256k1 twins secp256k1_ecmult_multi_var x2 at :842-848)
 * For n <= 32 768 both run in ONE kernel launch (half the chip each); above that they run one after the other.
 * out_a = sum scalars[i] * points_a[i], out_b = sum scalars[i] * points_b[i]; encodings as porla_bn254_msm_device. */
int porla_bn254_msm_pair_device(const void *d_scalars, const void *d_points_a, const void *d_points_b, size_t n,
                                uint8_t out_a[64], uint8_t out_b[64], void *hip_stream);
int porla_bn254_msm_pair_host(const uint8_t *scalars, const uint8_t *points_a, const uint8_t *points_b, size_t n,
                              uint8_t out_a[64], uint8_t out_b[64]);
/* ... and straight from the server's resident MAC arrays: per challenged row i the points store_a[idx[i]], store_b[idx[i]] (64-byte
 * affine each) and the scalar coef[i] (abs(int32), bn254_scalar_set_int, utils.h:271-275) -- the gather Server::audit does on the
 * host into ptc / pta / sc (Server.hpp:838-848, 893-899) runs on the device; idx / coef are the arrays porla_audit_combine_device
 * takes.  All pointers device pointers; outputs host. */
int porla_bn254_audit_msm_pair_device(const void *d_store_a, const void *d_store_b, const uint64_t *d_idx, const uint32_t *d_coef,
                                      size_t n, uint8_t out_a[64], uint8_t out_b[64], void *hip_stream);
/* two-phase form (slots 1..3 as porla_bn254_msm_device_begin): begin gathers and launches on hip_stream and returns, end waits and
 * folds -- the audit's other chain (row combine -> alignment commitment -> proof) runs in between; 1 .. 32 768 challenged rows */
int porla_bn254_audit_msm_pair_begin(int slot, const void *d_store_a, const void *d_store_b, const uint64_t *d_idx,
                                     const uint32_t *d_coef, size_t n, void *hip_stream);
int porla_bn254_audit_msm_pair_end(int slot, uint8_t out_a[64], uint8_t out_b[64]);
/* Two-phase form for independent MSMs in flight at once (e.g. the audit's two MSMs, Server.hpp:900-901): begin enqueues
 * every kernel of one MSM on hip_stream and returns; end waits for that slot, folds the reduction tree's sums on the host and writes
 * 64 bytes affine (jacobian = 0) or 96 bytes Jacobian (jacobian = 1).  slot in 1..3 (0 is used by the blocking calls);
 * one begin per slot until its end.  Overlap comes from using a different stream per slot.  A slot belongs to the device that
 * was current at begin: end must run with the same current device (PORLA_ERR_STATE otherwise, or without a begin). */
int porla_bn254_msm_device_begin(int slot, const void *d_scalars, const void *d_points, size_t n, void *hip_stream);
int porla_bn254_msm_device_end(int slot, uint8_t *out, int jacobian);
int porla_bn254_msm_host(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out_affine[64]);
/* The same over several pair ranges and devices of this process -- the multi-GPU form that stays behind the C ABI (the
 * reference folds the partial sums of its 8 pool threads the same way, porla/Client/Client.hpp:761-787): the pairs are cut
 * into `shards` contiguous ranges, device g of `devices` (counted from the current device, modulo the visible ones) owns a
 * contiguous block of ranges and runs them from its own host thread, stream and workspace slots, uploading range k+1 under
 * the kernels of range k; the range totals are added on the host.  shards <= 0 / devices <= 0: automatic (devices: as many
 * visible ones as get >= 2^17 pairs each, or PORLA_MSM_DEVICES; shards: up to 4 ranges of >= 2^17 pairs per device).
 * porla_bn254_msm_host -- and therefore compute_multi_exp -- takes this path from 2^18 pairs on (PORLA_MSM_SPLIT_MIN).
 * Any (shards, devices) gives the same 64 bytes. */
int porla_bn254_msm_host_multi(const uint8_t *scalars, const uint8_t *points, size_t n, int shards, int devices,
                               uint8_t out_affine[64]);
/* diagnostic: ranges and devices the most recent *_msm_host_multi used */
int porla_gpu_last_msm_multi(int *shards, int *devices);
int porla_bn254_jac_sum(const uint8_t *jacobians, size_t count, uint8_t out_affine[64]);
/* the MSM's host tail on its own (no device needed): the bucket-reduction tree leaves, per window w < windows, S_w and
 * M_{w,k} (k < window_bits - 1) -- here as 64-byte affine points, sums[w * window_bits + 0] = S_w, [.. + 1 + k] = M_{w,k} --
 * and the result is sum_w 2^(window_bits * w) * (S_w + sum_k 2^k M_{w,k}).  Exists for the CPU test of that fold. */
int porla_bn254_tree_fold(const uint8_t *sums_affine, int windows, int window_bits, uint8_t out_affine[64]);

/* ---- secp256k1 MSM (canonical encodings: 32-byte BE scalar, 64-byte x||y BE affine, zeros = infinity) ---- */
int porla_secp256k1_msm_device(const void *d_scalars, const void *d_points, size_t n, uint8_t out_affine[64],
                               void *hip_stream);
int porla_secp256k1_msm_pair_device(const void *d_scalars, const void *d_points_a, const void *d_points_b, size_t n,
                                    uint8_t out_a[64], uint8_t out_b[64], void *hip_stream);
int porla_secp256k1_msm_pair_host(const uint8_t *scalars, const uint8_t *points_a, const uint8_t *points_b, size_t n,
                                  uint8_t out_a[64], uint8_t out_b[64]);
int porla_secp256k1_audit_msm_pair_device(const void *d_store_a, const void *d_store_b, const uint64_t *d_idx,
                                          const uint32_t *d_coef, size_t n, uint8_t out_a[64], uint8_t out_b[64], void *hip_stream);
int porla_secp256k1_audit_msm_pair_begin(int slot, const void *d_store_a, const void *d_store_b, const uint64_t *d_idx,
                                         const uint32_t *d_coef, size_t n, void *hip_stream);
int porla_secp256k1_audit_msm_pair_end(int slot, uint8_t out_a[64], uint8_t out_b[64]);
int porla_secp256k1_msm_device_partial(const void *d_scalars, const void *d_points, size_t n,
                                       uint8_t out_jacobian[96], void *hip_stream);
int porla_secp256k1_msm_device_begin(int slot, const void *d_scalars, const void *d_points, size_t n, void *hip_stream);
int porla_secp256k1_msm_device_end(int slot, uint8_t *out, int jacobian);   /* the two MSMs of the IPA audit, Server.hpp:842-848 */
int porla_secp256k1_msm_host(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out_affine[64]);
int porla_secp256k1_msm_host_multi(const uint8_t *scalars, const uint8_t *points, size_t n, int shards, int devices,
                                   uint8_t out_affine[64]);
int porla_secp256k1_jac_sum(const uint8_t *jacobians, size_t count, uint8_t out_affine[64]);
int porla_secp256k1_tree_fold(const uint8_t *sums_affine, int windows, int window_bits, uint8_t out_affine[64]);

/* ---- one process per GPU: the range-sharded MSM across processes (BASELINE config 3) ----
 * Every rank owns a pair range resident in its GPU's HBM, runs a full MSM over it and contributes ONE 96-byte partial
 * Jacobian sum; the only exchange step is one ncclAllGather of world x 96 bytes (RCCL over xGMI, issued from C++ on the
 * engine's own stream), followed by world - 1 group additions and one inversion on every rank's host.  RCCL is bound with
 * dlopen at the first porla_dist_* call (PORLA_RCCL_LIB overrides the library name).
 *   rank 0: porla_dist_unique_id(id), hand the 128 bytes to the other ranks by any means (the launcher's store, MPI, a file)
 *   all   : porla_gpu_set_device(local_rank); porla_dist_init(id, rank, world)      [collective]
 *   all   : porla_bn254_msm_device_dist(my range ...) -> the whole job's 64-byte result on every rank   [collective]
 *           or a partial from porla_*_msm_device_partial / _device_end(slot, out, 1) handed to porla_*_dist_fold
 *   all   : porla_dist_finalize()
 * porla_dist_init is bounded: a peer that never arrives makes it fail with PORLA_ERR_STATE after PORLA_DIST_INIT_TIMEOUT_S
 * (default 180 s) instead of waiting forever.  The pending RCCL call cannot be cancelled: its helper thread is marked abandoned
 * (a communicator it still obtains is aborted at once), every later porla_dist_* call of the process fails with
 * PORLA_ERR_STATE, and the caller must leave with a non-zero _exit() -- not exit(): static destructors would run under a
 * thread that is still inside RCCL -- and continue, if at all, in a fresh child process. */
int porla_dist_unique_id(uint8_t id_out[PORLA_DIST_ID_BYTES]);
int porla_dist_init(const uint8_t id[PORLA_DIST_ID_BYTES], int rank, int world);
int porla_dist_info(int *rank, int *world);     /* world = 0 before porla_dist_init */
int porla_dist_finalize(void);
int porla_dist_allgather_partials(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t *all_out /* world * 96 bytes */);
int porla_bn254_dist_fold(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t out_affine[64]);
int porla_secp256k1_dist_fold(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t out_affine[64]);
int porla_bn254_msm_device_dist(const void *d_scalars, const void *d_points, size_t n_local, uint8_t out_affine[64],
                                void *hip_stream);
int porla_secp256k1_msm_device_dist(const void *d_scalars, const void *d_points, size_t n_local, uint8_t out_affine[64],
                                    void *hip_stream);

/* ---- batched fixed-base commitments ----
 * out[r] = sum_{i < n_coeffs} (row_r[i] mod order) * base[i] for every row r, as 64-byte X||Y big-endian affine points.
 * rows: coefficient i of row r at rows + r*row_stride + 32*i, 32 bytes big-endian (bn254_scalar, utils.h:307-318).
 * The base is expanded once into a table of window multiples resident in HBM (window_bits c, 0 = automatic: the
 * widest c <= 20 whose table fits min(a quarter of the free HBM, PORLA_COMMIT_TABLE_GB = 64 GB): 20 bits = 56 GB for
 * 128 BN254 points on a 288 GB MI355X); a commitment is then n_coeffs * ceil((bits+1)/c) mixed additions.  curve: 0 = BN254 G1, 1 = secp256k1. */
typedef struct porla_fixed_base porla_fixed_base;
int  porla_fixed_base_create(int curve, const uint8_t *points, size_t n_points, int window_bits, porla_fixed_base **out);
int  porla_fixed_base_info(const porla_fixed_base *fb, int *window_bits, int *windows, unsigned long long *table_bytes);
int  porla_fixed_base_commit_device(porla_fixed_base *fb, const void *d_rows, size_t n_rows, size_t n_coeffs,
                                    size_t row_stride, void *d_out, void *hip_stream);
int  porla_fixed_base_commit_host(porla_fixed_base *fb, const uint8_t *rows, size_t n_rows, size_t n_coeffs,
                                  size_t row_stride, uint8_t *out);
/* Server::audit for the IPA build up to the inner-product proof, in ONE call (porla/Server/Server.hpp:790-857): row combine and
 * alignment scalars, the two secp256k1 MSMs over the challenged MACs (arguments as porla_secp256k1_audit_msm_pair_device), and the
 * two Pedersen commitments over the fixed generators -- compute_commitment(c) of align_MAC (:495-529) and compute_commitment(B)
 * (:856) -- as one two-row launch.  generators_fb = porla_fixed_base_create(1, generators, n_cols, ...).  b_out (may be NULL): B mod
 * p_icc as n_cols 32-byte big-endian values, the input of inner_product_prove.  Outputs on the host; blocking. */
int porla_ipa_audit_device(porla_fixed_base *generators_fb, const void *d_rows64, const uint64_t *d_idx64,
                           const uint32_t *d_coef64, size_t n64, const void *d_rows32, const uint64_t *d_idx32,
                           const uint32_t *d_coef32, size_t n32, size_t n_cols, const void *d_mac_store,
                           const void *d_align_store, const uint64_t *d_mac_idx, const uint32_t *d_mac_coef, size_t n_macs,
                           uint8_t combined_mac[64], uint8_t combined_align[64], uint8_t align_value[64],
                           uint8_t commitment[64], uint8_t *b_out, void *hip_stream);
void porla_fixed_base_destroy(porla_fixed_base *fb);
/* KZG: rows of n_samples coefficients (4096 bytes per row for NUM_CHUNKS = 128) against the resident SRS */
int  porla_kzg_commit_batch_device(const void *d_rows, size_t n_rows, void *d_out, void *hip_stream);
int  porla_kzg_commit_batch_host(const uint8_t *rows, size_t n_rows, uint8_t *out);
/* Server::audit (KZG build) after the challenge has been drawn, in ONE call (porla/Server/Server.hpp:564-931): the row combine and
 * alignment scalars (arguments as porla_audit_combine_device, n_cols = the SRS size), the two MSMs over the challenged MACs
 * (arguments as porla_bn254_audit_msm_pair_device; they run on a stream of their own beside the rest), align_MAC's commitment and
 * create_proof(random_point, B) -- the three commitments as one launch.  Outputs on the host: combined_MAC, combined_align,
 * align_value = Commit(c), and the proof (commitment = Commit(B), H, point, claim; main.go:153-175); b_out (may be NULL): B mod
 * p_icc as n_cols 32-byte big-endian values.  Everything else device pointers; blocking; one audit at a time per process.
 * Stream contract (also porla_ipa_audit_device): hip_stream orders the INPUTS -- every kernel of the audit, on whichever
 * internal stream it runs, waits for what the caller had enqueued on hip_stream (NULL: the null stream) when the call was made,
 * so index / coefficient arrays uploaded asynchronously on that stream just before the call are safe. */
int  porla_kzg_audit_device(const void *d_rows64, const uint64_t *d_idx64, const uint32_t *d_coef64, size_t n64,
                            const void *d_rows32, const uint64_t *d_idx32, const uint32_t *d_coef32, size_t n32,
                            const void *d_mac_store, const void *d_align_store, const uint64_t *d_mac_idx,
                            const uint32_t *d_mac_coef, size_t n_macs, unsigned long long random_point,
                            uint8_t combined_mac[64], uint8_t combined_align[64], uint8_t align_value[64],
                            uint8_t commitment[64], uint8_t proof_h[64], uint8_t proof_point[32], uint8_t proof_claim[32],
                            uint8_t *b_out, void *hip_stream);
/* The last encode stage of a large CRebuild (KZG build) in ONE call, everything resident in HBM (porla/Server/Server.hpp:1487-1833,
 * :2059-2065): rows_in = n_rows x n_samples raw 32-byte chunks; outputs per part (X, Y): the rows mod p_icc (aligned_x / aligned_y:
 * n_rows * n_samples * 32 bytes each, may be NULL), the alignment scalars of BOTH parts back to back (scalars_xy: 2 * n_rows *
 * n_samples * 32 bytes, X first), their commitments (commits_xy: 2 * n_rows * 64 bytes: compute_digest_from_srs per row), and the
 * two MAC encodes (macs_in / macs_x / macs_y: n_rows points of 64 bytes).  = porla_icc_encode_xy_device + ONE
 * porla_kzg_commit_batch_device over the 2 n rows + porla_icc_mac_encode_xy_device, the MAC network on a second stream inside,
 * started first and with register room kept for it on every SIMD (the two sides overlap instead of queueing).  Asynchronous on
 * hip_stream; same bytes as the separate calls. */
int  porla_kzg_crebuild_stage_device(const void *d_rows_in, size_t n_rows, unsigned long long write_step, void *d_aligned_x,
                                     void *d_aligned_y, void *d_scalars_xy, void *d_commits_xy, const void *d_macs_in,
                                     void *d_macs_x, void *d_macs_y, void *hip_stream);
/* rows resident on the device, results wanted on the host at once (the audit's align_MAC commitment on the scalars
 * porla_audit_combine_device left in HBM, Server.hpp:903 -> :550-560): up to 64 rows run as ONE launch on hip_stream, behind whatever
 * produced the rows there, and the call returns when the pinned result has arrived; blocking */
int  porla_kzg_commit_batch_device_to_host(const void *d_rows, size_t n_rows, uint8_t *out /* n_rows * 64 */, void *hip_stream);
/* the same with the row range split over `devices` GPUs of this process (0 = every visible one), one host thread and one
 * resident copy of the SRS table per device; rows are independent, nothing is exchanged (Server.hpp:1077-1078, 2061-2062) */
int  porla_kzg_commit_batch_host_multi(const uint8_t *rows, size_t n_rows, uint8_t *out, int devices);
/* frees the HBM copies of the KZG state (SRS + window table, one-point tables, scratch); rebuilt on the next use */
int  porla_kzg_release_device_memory(void);
/* Client side, batched (Client::initialize computes both per block, porla/Client/Client.hpp:408-455):
 *   digest     = compute_digest (main.go:70-89) per row: alpha * f(tau) * G1[0]; rows as for commit_batch
 *   complement = compute_digest_complement (main.go:91-101) per scalar: s * h_MAC; scalars 32 bytes big-endian each
 *                (the reference passes its 16-byte AES output: left-pad it with 16 zero bytes)
 * Needs init_key + init_SRS in this process (tau, alpha and the hiding base are the client's secrets). */
int  porla_kzg_digest_batch_device(const void *d_rows, size_t n_rows, void *d_out, void *hip_stream);
int  porla_kzg_complement_batch_device(const void *d_scalars, size_t n, void *d_out, void *hip_stream);
/* the block's MAC as the client sends it -- digest(row r) + complement(scalar r), the add_point of Client.hpp:229-236 / 468-478
 * included -- as one two-coefficient commitment per row against the table of (G1[0], h_MAC): d_out[r] = 64 bytes */
int  porla_kzg_mac_batch_device(const void *d_rows, const void *d_scalars, size_t n_rows, void *d_out, void *hip_stream);
/* the three batches on caller-owned host buffers (pageable is fine): staged, computed on the engine's stream, copied back; blocking.
 * The copies dominate -- 4 KiB per block over PCIe */
int  porla_kzg_digest_batch_host(const uint8_t *rows, size_t n_rows, uint8_t *out);
int  porla_kzg_complement_batch_host(const uint8_t *scalars, size_t n, uint8_t *out);
int  porla_kzg_mac_batch_host(const uint8_t *rows, const uint8_t *scalars, size_t n_rows, uint8_t *out);
/* diagnostics of the host pairing behind verify_proof (main.go:177-193): scalar * G2 generator as 128 bytes
 * X.A1 || X.A0 || Y.A1 || Y.A0 big-endian; e(p1, q1) * e(p2, q2) == 1 ? (returns 1 / 0; slow = 1: the literal form with
 * affine Miller steps and the exponent (p^12 - 1)/r, kept as the reference for the fast form) */
int  porla_bn254_g2_mul_generator(const uint8_t scalar_be[32], uint8_t out[128]);
int  porla_bn254_pairing_product_is_one(const uint8_t p1[64], const uint8_t q1[128], const uint8_t p2[64], const uint8_t q2[128],
                                        int slow);
/* the same predicate on EIP-197's own input layout, any number of pairs: n_pairs x 192 bytes (G1 X || Y, then G2
 * x_im || x_re || y_im || y_re, 32-byte big-endian each; zeros = infinity).  1 / 0; PORLA_ERR_ARG for inputs the precompile
 * rejects (coordinate >= p, point off its curve, G2 point outside the order-r subgroup).  Host code, no device needed. */
int  porla_bn254_pairing_check(const uint8_t *input, size_t n_pairs, int slow);
/* window bits used when the SRS table is (re)built; 0 = automatic */
int  porla_kzg_set_commit_window(int window_bits);
/* diagnostic: window bits / windows per coefficient of the SRS table currently resident (0, 0 before the first batch) */
int  porla_kzg_commit_shape(int *window_bits, int *windows);
/* coefficients per commitment row = the SRS size given to init_SRS / init_SRS_from_data (main.go:45-68; the reference's
 * NUM_CHUNKS = 128, config.hpp), 0 before either ran: the row stride of every *_batch_* entry point is 32 bytes times this */
int  porla_kzg_row_coefficients(size_t *n_out);

/* ---- ICC encode (CRebuild_Cached data part + align_MAC scalar part) ----
 * rows_in : n_rows * n_cols elements, 32 bytes little-endian each (8 x uint32 LE words, utils.h:353-364; the layout
 *           of the U/<i> block files, utils.h:592-608), row-major; n_rows a power of two >= 2; n_cols = NUM_CHUNKS = 128
 * curve   : 0 = BN254 order (ENABLE_KZG), 1 = secp256k1 order (IPA)  -- selects q in LCM = p_icc * q (utils.h:33-34,42-43)
 * part    : 0 = X part; 1 = Y part (elements pre-scaled by wt = w^reverse_bits(write_step % n_rows, height-1),
 *           Server.hpp:1494,1522)
 * x_out   : n_rows * n_cols * 64 bytes, values in [0, LCM) little-endian (512-bit row format, utils.h:473-517)  (or NULL)
 * aligned : n_rows * n_cols * 32 bytes, values mod p_icc little-endian (256-bit row format)                     (or NULL)
 * scalars : n_rows * n_cols * 32 bytes, alignment scalars c = (A mod p_icc - A) mod q (Server.hpp:535-538),
 *           big-endian (bn254_scalar, utils.h:307-318) or, with scalar_le = 1, little-endian limbs
 *           (secp256k1_scalar d[4], scalar_4x64.h:13-15)                                                        (or NULL) */
int porla_icc_encode_device(const void *d_rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                            int part, void *d_x_out, void *d_aligned_out, void *d_scalars_out, int scalar_le,
                            void *hip_stream);
int porla_icc_encode_host(const uint8_t *rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                          int part, uint8_t *x_out, uint8_t *aligned_out, uint8_t *scalars_out, int scalar_le);
/* Column sharding (the reference splits the columns of every stage over its 8 pool threads, Server.hpp:1564-1686; the 128
 * per-column transforms are independent): only columns [col_begin, col_end) of the row-major input are uploaded (strided),
 * encoded and written back into the same columns of the full-width outputs.  One process per GPU: rank g passes
 * porla_shard_range(n_cols, g, G) -- 16 columns each on 8 GPUs; or let _host_multi run `devices` GPUs of this process
 * (0 = every visible one) from one host thread each.  No collective. */
/* BOTH parts from one run of the network (device pointers; any output may be NULL): the network is linear over Z/LCM and the Y
 * part's chunks are the X part's times wt (Server.hpp:1494, :1512-1522), so Y_k = wt X_k mod LCM -- one product per residue and
 * symbol in the last pass instead of a second encode; the bytes are those of the part = 0 and part = 1 calls */
int porla_icc_encode_xy_device(const void *d_rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                               void *d_x_out, void *d_aligned_out, void *d_scalars_out, void *d_y_x_out, void *d_y_aligned_out,
                               void *d_y_scalars_out, int scalar_le, void *hip_stream);
int porla_icc_encode_cols_host(const uint8_t *rows_in, size_t n_rows, size_t n_cols, size_t col_begin, size_t col_end, int curve,
                               unsigned long long write_step, int part, uint8_t *x_out, uint8_t *aligned_out,
                               uint8_t *scalars_out, int scalar_le);
int porla_icc_encode_host_multi(const uint8_t *rows_in, size_t n_rows, size_t n_cols, int curve, unsigned long long write_step,
                                int part, uint8_t *x_out, uint8_t *aligned_out, uint8_t *scalars_out, int scalar_le, int devices);

/* ---- MAC-side ICC encode ("FFT in the exponent": the MAC halves of CRebuild_Cached, Server.hpp:1523-1536 init
 * scaling, :1590-1609 / :1658-1676 butterflies tm = v^j * MAC[k+m2]; MAC[k] = um + tm; MAC[k+m2] = um - tm; client twin
 * Client.hpp:1040-1453) ----
 * macs_in / macs_out : n_rows points, 64 bytes X||Y big-endian affine each (KZG MAC_Block, utils.h:65; for IPA the
 *           include shim converts secp256k1_gej <-> this canonical form), n_rows a power of two >= 2
 * curve, write_step, part : as for porla_icc_encode_* (part 1 = Y: inputs pre-multiplied by wt) */
int porla_icc_mac_encode_device(const void *d_macs_in, size_t n_rows, int curve, unsigned long long write_step, int part,
                                void *d_macs_out, void *hip_stream);
/* BOTH parts from one butterfly network (what CRebuild_Cached needs: Server.hpp:1548-1687 X part, :1691-1830 Y part): the network
 * is linear over Z_q and the Y part's inputs are wt * MAC_U (:1528-1536), so Y_k = wt * X_k -- one scalar multiplication per
 * row instead of a second run of log2(n_rows) dependent stages; the 64 output bytes per point are those of the two single-part
 * calls */
int porla_icc_mac_encode_xy_device(const void *d_macs_in, size_t n_rows, int curve, unsigned long long write_step,
                                   void *d_macs_x_out, void *d_macs_y_out, void *hip_stream);
int porla_icc_mac_encode_xy_host(const uint8_t *macs_in, size_t n_rows, int curve, unsigned long long write_step,
                                 uint8_t *macs_x_out, uint8_t *macs_y_out);
int porla_icc_mac_encode_host(const uint8_t *macs_in, size_t n_rows, int curve, unsigned long long write_step, int part,
                              uint8_t *macs_out);
/* row counts <= n_rows use the matrix form (N commitments against the per-call base), larger ones the stage-by-stage ladder
 * form.  Default 0 = always the ladder, which since round 5 is the faster one at every size; the matrix form stays as an
 * independent formulation of the same map (the tests run both).  Both are bit-exact. */
int porla_icc_mac_set_matrix_max(size_t n_rows);

/* ---- Server::mix, the incremental form of one butterfly stage between two sub-levels (Server.hpp:1209-1328) ----
 * data part (Server.hpp:1269-1278): out[i] = (A0[i] + v^i A1[i]) % LCM, out[i+len] = (A0[i] - v^i A1[i]) % LCM, v = w^(n_total/len)
 *   a0, a1 : len x n_cols symbols, 64 bytes little-endian each, values < LCM (512-bit rows, utils.h:473-517); out : 2*len rows
 * MAC part (Server.hpp:1281-1318, used for the MACs and for the alignments): the same on 64-byte big-endian affine points.
 * len and n_total (= num_blocks, which fixes w) are powers of two, len <= n_total. */
int porla_icc_mix_device(const void *d_a0, const void *d_a1, size_t len, size_t n_cols, size_t n_total, int curve, void *d_out,
                         void *hip_stream);
int porla_icc_mix_host(const uint8_t *a0, const uint8_t *a1, size_t len, size_t n_cols, size_t n_total, int curve, uint8_t *out);
int porla_icc_mac_mix_device(const void *d_a0, const void *d_a1, size_t len, size_t n_total, int curve, void *d_out, void *hip_stream);
/* both point butterflies of a mix -- the MAC commitments (a) and the MAC alignments (b), same v^i -- in one launch */
int porla_icc_mac_mix_pair_device(const void *d_a0, const void *d_a1, const void *d_b0, const void *d_b1, size_t len, size_t n_total,
                                  int curve, void *d_out_a, void *d_out_b, void *hip_stream);
int porla_icc_mac_mix_host(const uint8_t *a0, const uint8_t *a1, size_t len, size_t n_total, int curve, uint8_t *out);
/* Server::mix(is_x, level) in one call: the data rows on hip_stream, both point arrays on a second stream beside them; asynchronous
 * (hip_stream continues when all three outputs are written) */
int porla_server_mix_device(const void *d_data_a0, const void *d_data_a1, const void *d_mac_a0, const void *d_mac_a1,
                            const void *d_align_a0, const void *d_align_a1, size_t len, size_t n_cols, size_t n_total, int curve,
                            void *d_data_out, void *d_mac_out, void *d_align_out, void *hip_stream);

/* ---- Server::HAdd / Client::HAdd and the HRebuild chains (Server.hpp:1388-1477, 1329-1386; Client.hpp:978-1038) ----
 * HAdd's arithmetic on ONE incoming block (the level bookkeeping around it stays the caller's):
 *   porla_icc_hadd_host      data_b2[i] = (data[i] * wt) mod p_icc and the alignment scalars c_i = (data_b2[i] - data[i] * wt) mod q
 *                            (align_MAC, Server.hpp:531-541 / 495-504), wt = w^reverse_bits(write_step % n_total, height-1);
 *                            wt_scalar_out: wt as the 32-byte big-endian scalar of the MAC side (convert_ZZ_to_scalar)
 *   porla_icc_mac_scale_host MAC_B2 = wt * MAC (host: one 64-byte operand; also Client::HAdd, Client.hpp:996-1014)
 *   porla_kzg_hadd_host      all three outputs of Server::HAdd for the KZG build: data_B2, MAC_B2 and MAC_align_B2 = Commit(c)
 *                            (IPA: take the scalars of porla_icc_hadd_host with scalar_le = 1 to the generators' fixed base,
 *                            porla_fixed_base_commit_host, or to secp256k1_ecmult_multi_var through the shim)
 * HRebuildX / HRebuildY: the chain of mixes that carries the block up to `level`, on the device in one call.  levels[i] points at
 * level i's rows (i <= level): 2 * 2^i rows, the first 2^i resident, the second 2^i incoming (levels[0] + one row = the new block);
 * step i mixes the halves of level i (Server::mix / Client::mix) into the incoming half of level i + 1; at the end level
 * `level`'s incoming half is copied over its resident half, as the reference does.  A row is n_cols 64-byte symbols
 * (porla_icc_hrebuild_host: the data levels) or one 64-byte affine point (porla_icc_mac_hrebuild_host: MAC commitments,
 * MAC alignments, the client's complements). */
int porla_icc_hadd_host(const uint8_t *data_in, size_t n_cols, size_t n_total, unsigned long long write_step, int curve,
                        uint8_t *data_b2_out, uint8_t *scalars_out, int scalar_le, uint8_t wt_scalar_out[32]);
int porla_icc_mac_scale_host(const uint8_t mac_in[64], size_t n_total, unsigned long long write_step, int curve, uint8_t mac_out[64]);
int porla_kzg_hadd_host(const uint8_t *data_in, const uint8_t mac_in[64], size_t n_total, unsigned long long write_step,
                        uint8_t *data_b2_out, uint8_t mac_b2_out[64], uint8_t mac_align_b2_out[64]);
int porla_icc_hrebuild_host(uint8_t *const *levels, int level, size_t n_cols, size_t n_total, int curve);
int porla_icc_mac_hrebuild_host(uint8_t *const *levels, int level, size_t n_total, int curve);

/* ---- audit row combine (Server::audit, Server.hpp:790-828) + the scalar part of align_MAC on the result (:531-541) ----
 * B_j = sum_i coeff_i * row_i[j] (exact integer), then aligned_j = B_j mod p_icc, c_j = (aligned_j - B_j) mod q.
 * The challenged rows are addressed inside row stores resident in HBM:
 *   d_rows64 : store of 512-bit rows, n_cols * 64 bytes per row, little-endian values < LCM (utils.h:473-517, cached levels)
 *   d_rows32 : store of 256-bit rows, n_cols * 32 bytes per row, little-endian (levels kept aligned)
 *   d_idx*   : row numbers inside the store (uint64), d_coef* : abs(int32) coefficients (Server.hpp:617-621), n* : counts
 * outputs (each may be NULL): exact 80 bytes/column LE (the unreduced B_j), aligned 32 B LE, aligned_be 32 B BE (the
 * coefficient format of create_proof / compute_digest_from_srs), scalars c_j 32 B BE.  curve selects q as for the ICC encode. */
int porla_audit_combine_device(const void *d_rows64, const uint64_t *d_idx64, const uint32_t *d_coef64, size_t n64,
                               const void *d_rows32, const uint64_t *d_idx32, const uint32_t *d_coef32, size_t n32,
                               size_t n_cols, int curve, void *d_exact_out, void *d_aligned_out, void *d_aligned_be_out,
                               void *d_scalars_out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif
