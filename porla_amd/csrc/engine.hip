// MI355X engine: HIP launch logic for the bucket MSM + the device-pointer C ABI (include/porla_gpu.h).
// Host language is C++ (the reference's plug-in is Go over cgo; no Go toolchain exists in this image,
// and the reference's callers are C++: porla/Utils/utils.h:277-292).
#include "engine.hpp"
#include "host_fold64.hpp"
#include "../../include/porla_gpu.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace porla {

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    set_last_error(buf);
    return PORLA_ERR_HIP;
}

int ensure_device() {
    static std::once_flag once;
    static int status = PORLA_ERR_NO_DEVICE;
    std::call_once(once, [] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            set_last_error("porla: no HIP device available (the MSM path has no CPU fallback)");
            status = PORLA_ERR_NO_DEVICE;
            return;
        }
        status = PORLA_OK;
    });
    if (status != PORLA_OK) set_last_error("porla: no HIP device available (the MSM path has no CPU fallback)");
    return status;
}

// ------------------------------------------------------------------------------------------------ profiling
struct ProfSlot { std::string name; double ms = 0; long long launches = 0; };
struct PendingEv { int slot; int dev; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static int g_prof_level = 0;   // 0 off, 1 every scope, 2 dominant kernels only
static std::vector<ProfSlot> g_prof;
static std::vector<PendingEv> g_pending;

static int prof_slot(const char* name) {
    for (size_t i = 0; i < g_prof.size(); i++) if (g_prof[i].name == name) return (int)i;
    g_prof.push_back(ProfSlot{name, 0, 0});
    return (int)g_prof.size() - 1;
}
// events are recycled (creating two per kernel launch costs more than the launch); an event belongs to the device that was
// current when it was created, so the pool is per device
constexpr int MAX_DEVICES = 16;
static std::vector<hipEvent_t> g_event_pool[MAX_DEVICES];
static int current_device_index() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
    return d;
}
static hipEvent_t take_event(int dev) {
    auto& pool = g_event_pool[dev];
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(const char* name, hipStream_t s, bool dominant) : slot(-1), stream(s), e0(nullptr), e1(nullptr), on(false) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_level == 0 || (g_prof_level == 2 && !dominant)) return;
    on = true;
    slot = prof_slot(name);
    const int dev = current_device_index();
    e0 = take_event(dev);
    e1 = take_event(dev);
    (void)hipEventRecord(e0, stream);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_pending.push_back(PendingEv{slot, current_device_index(), e0, e1});
}
void prof_flush() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_pending) {
        float ms = 0;
        (void)hipEventSynchronize(p.e1);
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            g_prof[p.slot].ms += ms;
            g_prof[p.slot].launches += 1;
        }
        g_event_pool[p.dev].push_back(p.e0);
        g_event_pool[p.dev].push_back(p.e1);
    }
    g_pending.clear();
}

// ------------------------------------------------------------------------------------------------ workspace
std::mutex g_ws_mu;  // the workspace registry; the use of a slot is serialised by its own Workspace::mu
static std::vector<Workspace*> g_ws;
int g_window_override = 0;
int g_small_mode = getenv("PORLA_MSM_SMALL") ? atoi(getenv("PORLA_MSM_SMALL")) : 1;
int g_small_c = 0;        // porla_gpu_set_msm_small sets it
int g_last_shape[3] = {0, 0, 0};
int g_last_multi[2] = {0, 0};
std::mutex g_multi_mu;   // one range-sharded host MSM at a time (it occupies the pipeline slots of every device it uses)
int g_use_glv = -1;       // -1: per-curve default; porla_gpu_set_msm_glv sets it

int get_workspace_slot(int slot, Workspace** out) {
    if (slot < 0 || slot >= MSM_SLOTS) { set_last_error("porla: MSM slot out of range"); return PORLA_ERR_ARG; }
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto* w : g_ws) if (w->device == dev && w->slot == slot) { *out = w; return PORLA_OK; }
    Workspace* w = new Workspace();
    w->device = dev;
    w->slot = slot;
    PORLA_HIP(hipStreamCreateWithFlags(&w->own_stream, hipStreamNonBlocking));
    g_ws.push_back(w);
    *out = w;
    return PORLA_OK;
}
int get_workspace(Workspace** out) { return get_workspace_slot(0, out); }
int lease_blocking_slot(Workspace** out) {
    Workspace* w0 = nullptr;
    int rc = get_workspace_slot(0, &w0);
    if (rc) return rc;
    if (w0->mu.try_lock()) { *out = w0; return PORLA_OK; }
    for (int k = MSM_POOL_SLOT0; k < MSM_POOL_SLOT0 + MSM_POOL_SLOTS; k++) {
        Workspace* w = nullptr;
        if ((rc = get_workspace_slot(k, &w))) return rc;
        if (w->mu.try_lock()) { *out = w; return PORLA_OK; }
    }
    w0->mu.lock();
    *out = w0;
    return PORLA_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI helpers
template <class C>
static int abi_msm_device(const void* d_scalars, const void* d_points, size_t n, uint8_t* out, void* stream, bool jac) {
    using M = typename C::Fp;
    if (n && (!d_scalars || !d_points || !out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<M> tot;
    int rc = msm_device<C>((const uint8_t*)d_scalars, (const uint8_t*)d_points, n, (hipStream_t)stream, &tot);
    if (rc) return rc;
    if (jac) h_xyzz_to_jac_bytes<M>(out, tot);
    else h_affine_to_bytes<M>(out, h_xyzz_to_affine64<M>(tot));
    return PORLA_OK;
}
template <class C>
static int abi_msm_host(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out) {
    using M = typename C::Fp;
    if (n && (!scalars || !points || !out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<M> tot;
    int rc = msm_host<C>(scalars, points, n, &tot);
    if (rc) return rc;
    h_affine_to_bytes<M>(out, h_xyzz_to_affine64<M>(tot));
    return PORLA_OK;
}
template <class C>
static int abi_msm_pair(const void* scalars, const void* points_a, const void* points_b, size_t n, uint8_t* out_a, uint8_t* out_b, void* stream,
                        bool device) {
    using M = typename C::Fp;
    if (!out_a || !out_b || (n && (!scalars || !points_a || !points_b))) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<M> ta, tb;
    int rc = device ? msm_pair_device<C>((const uint8_t*)scalars, (const uint8_t*)points_a, (const uint8_t*)points_b, n, (hipStream_t)stream, &ta, &tb)
                    : msm_pair_host<C>((const uint8_t*)scalars, (const uint8_t*)points_a, (const uint8_t*)points_b, n, &ta, &tb);
    if (rc) return rc;
    h_affine_to_bytes<M>(out_a, h_xyzz_to_affine64<M>(ta));
    h_affine_to_bytes<M>(out_b, h_xyzz_to_affine64<M>(tb));
    return PORLA_OK;
}
template <class C>
static int abi_audit_msm_pair(const void* d_store_a, const void* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef, size_t n,
                              uint8_t* out_a, uint8_t* out_b, void* stream) {
    using M = typename C::Fp;
    if (!out_a || !out_b || (n && (!d_store_a || !d_store_b || !d_idx || !d_coef))) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<M> ta, tb;
    int rc = msm_pair_gather_device<C>((const uint8_t*)d_store_a, (const uint8_t*)d_store_b, d_idx, d_coef, n, (hipStream_t)stream, &ta, &tb);
    if (rc) return rc;
    h_affine_to_bytes<M>(out_a, h_xyzz_to_affine64<M>(ta));
    h_affine_to_bytes<M>(out_b, h_xyzz_to_affine64<M>(tb));
    return PORLA_OK;
}
template <class C>
static int abi_msm_host_multi(const uint8_t* scalars, const uint8_t* points, size_t n, int shards, int devices, uint8_t* out) {
    using M = typename C::Fp;
    if (n && (!scalars || !points || !out)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<M> tot;
    int rc = msm_host_multi<C>(scalars, points, n, shards, devices, &tot);
    if (rc) return rc;
    h_affine_to_bytes<M>(out, h_xyzz_to_affine64<M>(tot));
    return PORLA_OK;
}
template <class C>
static int abi_tree_fold(const uint8_t* sums, int W, int c, uint8_t* out) {
    using M = typename C::Fp;
    if (W < 1 || c < 2 || c > 20 || !sums || !out) { set_last_error("porla: bad argument"); return PORLA_ERR_ARG; }
    std::vector<XYZZ<M>> fin((size_t)W * c);
    for (size_t i = 0; i < fin.size(); i++) fin[i] = xyzz_from_affine<M>(h_affine_from_bytes<M>(sums + 64 * i));
    h_affine_to_bytes<M>(out, h_xyzz_to_affine64<M>(h_fold_tree64<M>(fin.data(), W, c)));
    return PORLA_OK;
}
template <class C>
static int abi_jac_sum(const uint8_t* jacs, size_t count, uint8_t* out) {
    using M = typename C::Fp;
    if (count && !jacs) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<M> acc = xyzz_inf<M>();
    for (size_t i = 0; i < count; i++) {
        XYZZ<M> p = h_xyzz_from_jac_bytes<M>(jacs + 96 * i);
        xyzz_add<M>(acc, p);
    }
    h_affine_to_bytes<M>(out, h_xyzz_to_affine64<M>(acc));
    return PORLA_OK;
}

int order_after_caller(Workspace* ws, hipStream_t caller, hipStream_t a, hipStream_t b) {
    // nothing pending on the caller's stream (the usual case of a blocking caller): nothing to order, and a query costs a
    // fraction of an event record plus waits (the audit is a 150 us call)
    if (hipStreamQuery(caller) == hipSuccess) return PORLA_OK;
    (void)hipGetLastError();                    // (hipErrorNotReady is the expected answer otherwise)
    if (!ws->caller_ev) PORLA_HIP(hipEventCreateWithFlags(&ws->caller_ev, hipEventDisableTiming));
    PORLA_HIP(hipEventRecord(ws->caller_ev, caller));
    if (a && a != caller) PORLA_HIP(hipStreamWaitEvent(a, ws->caller_ev, 0));
    if (b && b != caller && b != a) PORLA_HIP(hipStreamWaitEvent(b, ws->caller_ev, 0));
    return PORLA_OK;
}

hipStream_t engine_stream() {
    Workspace* ws;
    if (get_workspace(&ws)) return nullptr;
    return ws->own_stream;
}

}  // namespace porla

using namespace porla;

namespace {
struct IccSecp256k1FnHost {  // secp256k1 group order, only P is needed (fe_reduce_plain)
    static constexpr uint32_t P[8] = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    static constexpr int SPARE_BITS = 0;          // (fe_add of porla_diag_fe_op)
};
}  // namespace

struct porla_fixed_base {
    int curve;
    FixedBase<Bn254G1> bn;
    FixedBase<Secp256k1G> secp;
};

extern "C" {

int porla_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int porla_gpu_set_device(int device) {
    PORLA_HIP(hipSetDevice(device));
    return PORLA_OK;
}
const char* porla_gpu_last_error(void) { return g_last_error.c_str(); }

int porla_gpu_profile_enable(int enable) {
    prof_flush();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_level = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    if (enable) g_prof.clear();
    return PORLA_OK;
}
int porla_gpu_profile_get(int slot, char* name, size_t name_cap, double* total_ms, long long* launches) {
    prof_flush();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < 0 || (size_t)slot >= g_prof.size()) return PORLA_ERR_ARG;
    if (name && name_cap) snprintf(name, name_cap, "%s", g_prof[slot].name.c_str());
    if (total_ms) *total_ms = g_prof[slot].ms;
    if (launches) *launches = g_prof[slot].launches;
    return PORLA_OK;
}
// frees the MSM scratch of every workspace slot (it grows with the largest call seen: ~0.5 GiB per slot at 2^20 pairs);
// refused while a two-phase MSM is still pending
int porla_gpu_release_msm_workspaces(void) {
    // lock order everywhere: a slot's mutex is never held while g_ws_mu is taken (msm_host_multi looks its slots up first), and
    // here the slots are only try-locked under the registry lock: a slot a host MSM is using makes the call fail, not wait
    std::lock_guard<std::mutex> lk(g_ws_mu);
    std::vector<Workspace*> held;
    auto unlock_all = [&]() { for (auto* w : held) w->mu.unlock(); };
    for (auto* w : g_ws) {
        if (!w->mu.try_lock()) { unlock_all(); set_last_error("porla: an MSM is running on a workspace slot"); return PORLA_ERR_STATE; }
        held.push_back(w);
        if (w->pend_W || w->begun) { unlock_all(); set_last_error("porla: an MSM is still pending (call the matching _end first)"); return PORLA_ERR_STATE; }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto* w : g_ws) {
        (void)hipSetDevice(w->device);
        Buf* bufs[] = {&w->pts, &w->keys, &w->entries, &w->counts, &w->starts, &w->fill, &w->cursor, &w->buckets, &w->in_scalars,
                       &w->in_points, &w->order, &w->blk_hist, &w->blk_off, &w->tile_off, &w->heavy, &w->chunk_out, &w->tree_s,
                       &w->tree_m, &w->tree_mt, &w->small_part, &w->multi_acc};
        for (Buf* b : bufs) b->release();
    }
    (void)hipSetDevice(cur);
    unlock_all();
    return PORLA_OK;
}
// unit u of n belongs to shard u * world / n: shard `rank` owns [rank n / world, (rank+1) n / world) -- the one rule every
// range split of the engine uses (MSM pair ranges, commitment rows, ICC columns)
int porla_shard_range(size_t n, int rank, int world, size_t* begin, size_t* end) {
    if (world < 1 || rank < 0 || rank >= world || !begin || !end) { set_last_error("porla: bad shard"); return PORLA_ERR_ARG; }
    *begin = (size_t)((unsigned __int128)n * (unsigned)rank / (unsigned)world);
    *end = (size_t)((unsigned __int128)n * (unsigned)(rank + 1) / (unsigned)world);
    return PORLA_OK;
}
int porla_gpu_set_msm_window(int c) { g_window_override = c; return PORLA_OK; }
int porla_gpu_last_msm_shape(int* c, int* windows, int* glv) {
    if (c) *c = g_last_shape[0];
    if (windows) *windows = g_last_shape[1];
    if (glv) *glv = g_last_shape[2];
    return PORLA_OK;
}
int porla_gpu_set_msm_small(int on, int window_bits) {
    g_small_mode = on != 0;
    g_small_c = window_bits >= 1 && window_bits <= SMALL_MAX_C ? window_bits : 0;
    return PORLA_OK;
}
int porla_gpu_set_msm_glv(int on) { g_use_glv = on < 0 ? -1 : (on != 0); return PORLA_OK; }

// host execution of the very code the digit kernel runs (glv.hip.h is __host__ __device__): lets the CPU tests pin the
// split against the Python model of tools/gen_glv.py
int porla_glv_split(int curve, const uint8_t scalar_be[32], uint8_t k1_mag_be[16], int* k1_neg, uint8_t k2_mag_be[16], int* k2_neg) {
    if (!scalar_be || !k1_mag_be || !k2_mag_be || !k1_neg || !k2_neg || (curve != 0 && curve != 1)) return PORLA_ERR_ARG;
    uint32_t k[8], m1[4], m2[4];
    bool n1, n2;
    h_load_be(k, scalar_be);
    if (curve == 0) { fe_reduce_plain<Bn254Fr>(k, 8); glv_split<GlvBn254>(k, m1, n1, m2, n2); }
    else { fe_reduce_plain<IccSecp256k1FnHost>(k, 2); glv_split<GlvSecp256k1>(k, m1, n1, m2, n2); }
    for (int i = 0; i < 4; i++) {
        for (int b = 0; b < 4; b++) {
            k1_mag_be[15 - (4 * i + b)] = (uint8_t)(m1[i] >> (8 * b));
            k2_mag_be[15 - (4 * i + b)] = (uint8_t)(m2[i] >> (8 * b));
        }
    }
    *k1_neg = n1; *k2_neg = n2;
    return PORLA_OK;
}

// diagnostics: HOST execution of the 8 x 32-bit helpers the kernels use (fe.hip.h), for the CPU suite -- the same source, the
// portable branch of sbb32 / adc32.  op 0: fe_reduce_small (value below (K + 1) modulus -> value mod modulus; modulus 0 = the BN254
// group order with K = 4, as the ICC finish step reduces A mod p_icc; 1 = the secp256k1 group order with K = 0);
// op 1: fe_neg; op 2: fe_neg_if(., true); op 3: fe_sub(a, b); op 4: fe_add(a, b).  Values are 32-byte little-endian.
int porla_diag_fe_op(int op, int modulus, const uint8_t a_le[32], const uint8_t b_le[32], uint8_t out_le[32]) {
    if (!a_le || !out_le || (modulus != 0 && modulus != 1) || op < 0 || op > 4 || (op >= 3 && !b_le)) return PORLA_ERR_ARG;
    uint32_t a[8], b[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8];
    for (int i = 0; i < 8; i++) {
        a[i] = (uint32_t)a_le[4 * i] | ((uint32_t)a_le[4 * i + 1] << 8) | ((uint32_t)a_le[4 * i + 2] << 16) | ((uint32_t)a_le[4 * i + 3] << 24);
        if (b_le) b[i] = (uint32_t)b_le[4 * i] | ((uint32_t)b_le[4 * i + 1] << 8) | ((uint32_t)b_le[4 * i + 2] << 16) | ((uint32_t)b_le[4 * i + 3] << 24);
    }
    auto run = [&](auto tag, auto ktag) {
        using M = decltype(tag);
        constexpr int K = decltype(ktag)::value;
        Fe<M> x, y, z;
        for (int i = 0; i < 8; i++) { x.v[i] = a[i]; y.v[i] = b[i]; }
        if (op == 0) { for (int i = 0; i < 8; i++) r[i] = a[i]; fe_reduce_small<M, K>(r); return; }
        if (op == 1) z = fe_neg<M>(x);
        else if (op == 2) z = fe_neg_if<M>(x, true);
        else if (op == 3) z = fe_sub<M>(x, y);
        else z = fe_add<M>(x, y);
        for (int i = 0; i < 8; i++) r[i] = z.v[i];
    };
    if (modulus == 0) run(Bn254Fr{}, std::integral_constant<int, 4>{});
    else run(IccSecp256k1FnHost{}, std::integral_constant<int, 0>{});
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++) out_le[4 * i + k] = (uint8_t)(r[i] >> (8 * k));
    return PORLA_OK;
}

int porla_bn254_msm_device(const void* d_scalars, const void* d_points, size_t n, uint8_t out_affine[64], void* s) {
    return abi_msm_device<Bn254G1>(d_scalars, d_points, n, out_affine, s, false);
}
int porla_bn254_msm_device_partial(const void* d_scalars, const void* d_points, size_t n, uint8_t out_jac[96], void* s) {
    return abi_msm_device<Bn254G1>(d_scalars, d_points, n, out_jac, s, true);
}
int porla_bn254_msm_host(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t out_affine[64]) {
    return abi_msm_host<Bn254G1>(scalars, points, n, out_affine);
}
int porla_bn254_msm_pair_device(const void* d_scalars, const void* d_points_a, const void* d_points_b, size_t n, uint8_t out_a[64],
                                uint8_t out_b[64], void* s) {
    return abi_msm_pair<Bn254G1>(d_scalars, d_points_a, d_points_b, n, out_a, out_b, s, true);
}
int porla_bn254_msm_pair_host(const uint8_t* scalars, const uint8_t* points_a, const uint8_t* points_b, size_t n, uint8_t out_a[64],
                              uint8_t out_b[64]) {
    return abi_msm_pair<Bn254G1>(scalars, points_a, points_b, n, out_a, out_b, nullptr, false);
}
int porla_bn254_audit_msm_pair_device(const void* d_store_a, const void* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef, size_t n,
                                      uint8_t out_a[64], uint8_t out_b[64], void* s) {
    return abi_audit_msm_pair<Bn254G1>(d_store_a, d_store_b, d_idx, d_coef, n, out_a, out_b, s);
}
int porla_bn254_audit_msm_pair_begin(int slot, const void* d_store_a, const void* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef,
                                     size_t n, void* s) {
    if (!d_store_a || !d_store_b || !d_idx || !d_coef) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    if (slot < 1 || slot >= MSM_USER_SLOTS) { set_last_error("porla: MSM slot out of range (1 .. 3; 0 belongs to the blocking calls)"); return PORLA_ERR_ARG; }
    return msm_pair_gather_begin<Bn254G1>(slot, (const uint8_t*)d_store_a, (const uint8_t*)d_store_b, d_idx, d_coef, n, (hipStream_t)s);
}
int porla_bn254_audit_msm_pair_end(int slot, uint8_t out_a[64], uint8_t out_b[64]) {
    if (!out_a || !out_b) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    if (slot < 1 || slot >= MSM_USER_SLOTS) { set_last_error("porla: MSM slot out of range (1 .. 3; 0 belongs to the blocking calls)"); return PORLA_ERR_ARG; }
    XYZZ<Bn254Fp> ta, tb;
    int rc = msm_pair_end<Bn254G1>(slot, &ta, &tb);
    if (rc) return rc;
    h_affine_to_bytes<Bn254Fp>(out_a, h_xyzz_to_affine64<Bn254Fp>(ta));
    h_affine_to_bytes<Bn254Fp>(out_b, h_xyzz_to_affine64<Bn254Fp>(tb));
    return PORLA_OK;
}
int porla_bn254_msm_host_multi(const uint8_t* scalars, const uint8_t* points, size_t n, int shards, int devices, uint8_t out_affine[64]) {
    return abi_msm_host_multi<Bn254G1>(scalars, points, n, shards, devices, out_affine);
}
int porla_secp256k1_msm_pair_device(const void* d_scalars, const void* d_points_a, const void* d_points_b, size_t n, uint8_t out_a[64],
                                    uint8_t out_b[64], void* s) {
    return abi_msm_pair<Secp256k1G>(d_scalars, d_points_a, d_points_b, n, out_a, out_b, s, true);
}
int porla_secp256k1_msm_pair_host(const uint8_t* scalars, const uint8_t* points_a, const uint8_t* points_b, size_t n, uint8_t out_a[64],
                                  uint8_t out_b[64]) {
    return abi_msm_pair<Secp256k1G>(scalars, points_a, points_b, n, out_a, out_b, nullptr, false);
}
int porla_secp256k1_audit_msm_pair_device(const void* d_store_a, const void* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef,
                                          size_t n, uint8_t out_a[64], uint8_t out_b[64], void* s) {
    return abi_audit_msm_pair<Secp256k1G>(d_store_a, d_store_b, d_idx, d_coef, n, out_a, out_b, s);
}
int porla_secp256k1_audit_msm_pair_begin(int slot, const void* d_store_a, const void* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef,
                                         size_t n, void* s) {
    if (!d_store_a || !d_store_b || !d_idx || !d_coef) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    if (slot < 1 || slot >= MSM_USER_SLOTS) { set_last_error("porla: MSM slot out of range (1 .. 3; 0 belongs to the blocking calls)"); return PORLA_ERR_ARG; }
    return msm_pair_gather_begin<Secp256k1G>(slot, (const uint8_t*)d_store_a, (const uint8_t*)d_store_b, d_idx, d_coef, n, (hipStream_t)s);
}
int porla_secp256k1_audit_msm_pair_end(int slot, uint8_t out_a[64], uint8_t out_b[64]) {
    if (!out_a || !out_b) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    if (slot < 1 || slot >= MSM_USER_SLOTS) { set_last_error("porla: MSM slot out of range (1 .. 3; 0 belongs to the blocking calls)"); return PORLA_ERR_ARG; }
    XYZZ<Secp256k1Fp> ta, tb;
    int rc = msm_pair_end<Secp256k1G>(slot, &ta, &tb);
    if (rc) return rc;
    h_affine_to_bytes<Secp256k1Fp>(out_a, h_xyzz_to_affine64<Secp256k1Fp>(ta));
    h_affine_to_bytes<Secp256k1Fp>(out_b, h_xyzz_to_affine64<Secp256k1Fp>(tb));
    return PORLA_OK;
}
int porla_secp256k1_msm_host_multi(const uint8_t* scalars, const uint8_t* points, size_t n, int shards, int devices, uint8_t out_affine[64]) {
    return abi_msm_host_multi<Secp256k1G>(scalars, points, n, shards, devices, out_affine);
}
int porla_gpu_last_msm_multi(int* shards, int* devices) {
    if (shards) *shards = g_last_multi[0];
    if (devices) *devices = g_last_multi[1];
    return PORLA_OK;
}
int porla_bn254_jac_sum(const uint8_t* jacs, size_t count, uint8_t out_affine[64]) {
    return abi_jac_sum<Bn254G1>(jacs, count, out_affine);
}
int porla_bn254_tree_fold(const uint8_t* sums, int windows, int window_bits, uint8_t out_affine[64]) {
    return abi_tree_fold<Bn254G1>(sums, windows, window_bits, out_affine);
}
int porla_secp256k1_tree_fold(const uint8_t* sums, int windows, int window_bits, uint8_t out_affine[64]) {
    return abi_tree_fold<Secp256k1G>(sums, windows, window_bits, out_affine);
}

static int user_slot_ok(int slot) {
    if (slot >= 0 && slot < MSM_USER_SLOTS) return PORLA_OK;
    set_last_error("porla: MSM slot out of range (0..3)");
    return PORLA_ERR_ARG;
}
int porla_bn254_msm_device_begin(int slot, const void* d_scalars, const void* d_points, size_t n, void* s) {
    if (user_slot_ok(slot)) return PORLA_ERR_ARG;
    if (n && (!d_scalars || !d_points)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    return msm_device_begin<Bn254G1>(slot, (const uint8_t*)d_scalars, (const uint8_t*)d_points, n, (hipStream_t)s);
}
int porla_bn254_msm_device_end(int slot, uint8_t* out, int jacobian) {
    if (user_slot_ok(slot)) return PORLA_ERR_ARG;
    if (!out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<Bn254Fp> tot;
    int rc = msm_device_end<Bn254G1>(slot, &tot);
    if (rc) return rc;
    if (jacobian) h_xyzz_to_jac_bytes<Bn254Fp>(out, tot);
    else h_affine_to_bytes<Bn254Fp>(out, h_xyzz_to_affine64<Bn254Fp>(tot));
    return PORLA_OK;
}

int porla_secp256k1_msm_device_begin(int slot, const void* d_scalars, const void* d_points, size_t n, void* s) {
    if (user_slot_ok(slot)) return PORLA_ERR_ARG;
    if (n && (!d_scalars || !d_points)) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    return msm_device_begin<Secp256k1G>(slot, (const uint8_t*)d_scalars, (const uint8_t*)d_points, n, (hipStream_t)s);
}
int porla_secp256k1_msm_device_end(int slot, uint8_t* out, int jacobian) {
    if (user_slot_ok(slot)) return PORLA_ERR_ARG;
    if (!out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    XYZZ<Secp256k1Fp> tot;
    int rc = msm_device_end<Secp256k1G>(slot, &tot);
    if (rc) return rc;
    if (jacobian) h_xyzz_to_jac_bytes<Secp256k1Fp>(out, tot);
    else h_affine_to_bytes<Secp256k1Fp>(out, h_xyzz_to_affine64<Secp256k1Fp>(tot));
    return PORLA_OK;
}

int porla_secp256k1_msm_device(const void* d_scalars, const void* d_points, size_t n, uint8_t out_affine[64], void* s) {
    return abi_msm_device<Secp256k1G>(d_scalars, d_points, n, out_affine, s, false);
}
int porla_secp256k1_msm_device_partial(const void* d_scalars, const void* d_points, size_t n, uint8_t out_jac[96], void* s) {
    return abi_msm_device<Secp256k1G>(d_scalars, d_points, n, out_jac, s, true);
}
int porla_secp256k1_msm_host(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t out_affine[64]) {
    return abi_msm_host<Secp256k1G>(scalars, points, n, out_affine);
}
int porla_secp256k1_jac_sum(const uint8_t* jacs, size_t count, uint8_t out_affine[64]) {
    return abi_jac_sum<Secp256k1G>(jacs, count, out_affine);
}

// MAC_B2 = wt * MAC: Server::HAdd (Server.hpp:1400-1417), Client::HAdd (Client.hpp:996-1014), the Y halves of Client::CRebuild
// (Client.hpp:1066-1075) -- one scalar multiplication on a 64-byte operand: host ("replicas only")
int porla_icc_mac_scale_host(const uint8_t mac_in[64], size_t n_total, unsigned long long write_step, int curve, uint8_t mac_out[64]) {
    if (!mac_in || !mac_out || (curve != 0 && curve != 1)) { set_last_error("porla: bad argument"); return PORLA_ERR_ARG; }
    uint8_t wt[32];
    int rc = icc_wt_scalar_be(n_total, write_step, wt);
    if (rc) return rc;
    uint32_t k[8];
    h_load_be(k, wt);
    if (curve == 0) {
        fe_reduce_plain<Bn254Fr>(k, 8);                 // bn254_mult -> fr.SetBytes reduces mod r (main.go:209)
        h_affine_to_bytes<Bn254Fp>(mac_out, h_xyzz_to_affine64<Bn254Fp>(h_scalar_mul64_glv<Bn254Fp, GlvBn254>(h_affine_from_bytes<Bn254Fp>(mac_in), k)));
    } else {
        fe_reduce_plain<IccSecp256k1FnHost>(k, 2);
        h_affine_to_bytes<Secp256k1Fp>(mac_out, h_xyzz_to_affine64<Secp256k1Fp>(h_scalar_mul64_glv<Secp256k1Fp, GlvSecp256k1>(h_affine_from_bytes<Secp256k1Fp>(mac_in), k)));
    }
    return PORLA_OK;
}

int porla_fixed_base_create(int curve, const uint8_t* points, size_t n_points, int window_bits, porla_fixed_base** out) {
    if (!out || (n_points && !points) || (curve != 0 && curve != 1)) { set_last_error("porla: bad argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    hipStream_t s = engine_stream();
    porla_fixed_base* fb = new porla_fixed_base();
    fb->curve = curve;
    rc = curve == 0 ? fb->bn.build_from_host_bytes(points, n_points, window_bits, s)
                    : fb->secp.build_from_host_bytes(points, n_points, window_bits, s);
    if (rc) { porla_fixed_base_destroy(fb); return rc; }
    *out = fb;
    return PORLA_OK;
}
int porla_fixed_base_info(const porla_fixed_base* fb, int* window_bits, int* windows, unsigned long long* table_bytes) {
    if (!fb) return PORLA_ERR_ARG;
    int c = fb->curve == 0 ? fb->bn.c : fb->secp.c, W = fb->curve == 0 ? fb->bn.W : fb->secp.W;
    size_t n = fb->curve == 0 ? fb->bn.n_points : fb->secp.n_points;
    if (window_bits) *window_bits = c;
    if (windows) *windows = W;
    if (table_bytes) *table_bytes = c ? (unsigned long long)n * W * ((size_t)1 << (c - 1)) * 64 : 0;
    return PORLA_OK;
}
int porla_fixed_base_commit_device(porla_fixed_base* fb, const void* d_rows, size_t n_rows, size_t n_coeffs,
                                   size_t row_stride, void* d_out, void* stream) {
    if (!fb || (n_rows && (!d_rows || !d_out))) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    if (fb->curve == 0) {
        std::lock_guard<std::mutex> lk(fb->bn.mu);
        return fb->bn.commit_device((const uint8_t*)d_rows, n_rows, n_coeffs, row_stride, (uint8_t*)d_out, (hipStream_t)stream);
    }
    std::lock_guard<std::mutex> lk(fb->secp.mu);
    return fb->secp.commit_device((const uint8_t*)d_rows, n_rows, n_coeffs, row_stride, (uint8_t*)d_out, (hipStream_t)stream);
}
int porla_fixed_base_commit_host(porla_fixed_base* fb, const uint8_t* rows, size_t n_rows, size_t n_coeffs,
                                 size_t row_stride, uint8_t* out) {
    if (!fb || (n_rows && (!rows || !out))) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    hipStream_t s = engine_stream();
    if (fb->curve == 0) {
        std::lock_guard<std::mutex> lk(fb->bn.mu);
        return fb->bn.commit_host(rows, n_rows, n_coeffs, row_stride, out, s);
    }
    std::lock_guard<std::mutex> lk(fb->secp.mu);
    return fb->secp.commit_host(rows, n_rows, n_coeffs, row_stride, out, s);
}
// Server::audit for the IPA build up to the inner-product proof (porla/Server/Server.hpp:790-857): the row combine and alignment
// scalars, the two secp256k1 MSMs over the challenged MACs (on the audit slot's own stream), and the two Pedersen commitments over
// the fixed generators -- compute_commitment(c) inside align_MAC (:495-529) and compute_commitment(B) (:856) -- as ONE two-row
// launch.  generators_fb: porla_fixed_base_create(1, generators, n_cols, ...).  B (n_cols 32-byte big-endian values mod p_icc) goes
// back to the caller for inner_product_prove (its L / R points: INTEGRATION.md s3c).
static std::mutex g_ipa_audit_mu;
static uint8_t* g_ipa_pin = nullptr;
static size_t g_ipa_pin_cap = 0;
static int g_ipa_pin_dev = -1;
int porla_ipa_audit_device(porla_fixed_base* generators_fb, const void* d_rows64, const uint64_t* d_idx64, const uint32_t* d_coef64,
                           size_t n64, const void* d_rows32, const uint64_t* d_idx32, const uint32_t* d_coef32, size_t n32, size_t n_cols,
                           const void* d_mac_store, const void* d_align_store, const uint64_t* d_mac_idx, const uint32_t* d_mac_coef,
                           size_t n_macs, uint8_t combined_mac[64], uint8_t combined_align[64], uint8_t align_value[64],
                           uint8_t commitment[64], uint8_t* b_out, void* hip_stream) {
    if (!generators_fb || generators_fb->curve != 1 || !combined_mac || !combined_align || !align_value || !commitment || n_cols == 0 ||
        (n_macs && (!d_mac_store || !d_align_store || !d_mac_idx || !d_mac_coef))) {
        set_last_error("porla: bad argument to porla_ipa_audit_device (generators_fb must be a secp256k1 fixed base)");
        return PORLA_ERR_ARG;
    }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_ipa_audit_mu);
    int dev = 0;
    PORLA_HIP(hipGetDevice(&dev));
    if (g_ipa_pin_cap < 64 * n_cols || g_ipa_pin_dev != dev) {
        if (g_ipa_pin) PORLA_HIP(hipHostFree(g_ipa_pin));
        g_ipa_pin = nullptr; g_ipa_pin_cap = 0;
        PORLA_HIP(hipHostMalloc((void**)&g_ipa_pin, 64 * n_cols, hipHostMallocMapped | hipHostMallocCoherent));
        g_ipa_pin_cap = 64 * n_cols;
        g_ipa_pin_dev = dev;
    }
    void* pin_dev = nullptr;
    PORLA_HIP(hipHostGetDevicePointer(&pin_dev, g_ipa_pin, 0));
    // hip_stream orders the INPUTS: the combine runs on it as given (NULL = the null stream), the pair on the audit slot's own
    // stream behind an event recorded on hip_stream now -- index / coefficient arrays the caller has just uploaded asynchronously
    // on it are complete before any kernel of the audit reads them (one record + one wait: ~4 us of a 150 us call)
    hipStream_t stream = (hipStream_t)hip_stream;
    Workspace* aw = nullptr;
    if ((rc = get_workspace_slot(MSM_AUDIT_SLOT, &aw))) return rc;
    if ((rc = order_after_caller(aw, (hipStream_t)hip_stream, stream, aw->own_stream))) return rc;
    // the combine first (its short kernels take their compute units before the pair's long-lived blocks), then the pair
    rc = porla_audit_combine_device(d_rows64, d_idx64, d_coef64, n64, d_rows32, d_idx32, d_coef32, n32, n_cols, 1, nullptr, nullptr,
                                    pin_dev, (uint8_t*)pin_dev + 32 * n_cols, stream);
    const bool pair = n_macs >= 1 && n_macs <= 32768;
    bool pair_begun = false;
    if (rc == PORLA_OK && pair) {
        rc = msm_pair_gather_begin<Secp256k1G>(MSM_AUDIT_SLOT, (const uint8_t*)d_mac_store, (const uint8_t*)d_align_store, d_mac_idx,
                                                   d_mac_coef, n_macs, aw->own_stream);
        pair_begun = rc == PORLA_OK;
    }
    auto collect_pair = [&]() -> int {
        if (!pair_begun) return PORLA_OK;
        XYZZ<Secp256k1Fp> ta, tb;
        int r2 = msm_pair_end<Secp256k1G>(MSM_AUDIT_SLOT, &ta, &tb);
        if (r2) return r2;
        h_affine_to_bytes<Secp256k1Fp>(combined_mac, h_xyzz_to_affine64<Secp256k1Fp>(ta));
        h_affine_to_bytes<Secp256k1Fp>(combined_align, h_xyzz_to_affine64<Secp256k1Fp>(tb));
        return PORLA_OK;
    };
    if (rc == PORLA_OK && hipStreamSynchronize(stream) != hipSuccess) {
        set_last_error("porla: hipStreamSynchronize failed in the audit");
        rc = PORLA_ERR_HIP;
    }
    if (rc) { (void)collect_pair(); return rc; }
    // rows: the alignment scalars, then B (both already 32-byte big-endian per column)
    std::vector<uint8_t> two(64 * n_cols);
    memcpy(two.data(), g_ipa_pin + 32 * n_cols, 32 * n_cols);
    memcpy(two.data() + 32 * n_cols, g_ipa_pin, 32 * n_cols);
    if (b_out) memcpy(b_out, g_ipa_pin, 32 * n_cols);
    uint8_t outs[128];
    {
        std::lock_guard<std::mutex> lkfb(generators_fb->secp.mu);
        rc = generators_fb->secp.commit_host(two.data(), 2, n_cols, 32 * n_cols, outs, engine_stream());
    }
    int rc2 = collect_pair();
    if (rc) return rc;
    if (rc2) return rc2;
    if (!pair && (rc = porla_secp256k1_audit_msm_pair_device(d_mac_store, d_align_store, d_mac_idx, d_mac_coef, n_macs, combined_mac,
                                                             combined_align, stream)))
        return rc;
    memcpy(align_value, outs, 64);
    memcpy(commitment, outs + 64, 64);
    return PORLA_OK;
}

void porla_fixed_base_destroy(porla_fixed_base* fb) {
    if (!fb) return;
    fb->bn.release();
    fb->secp.release();
    delete fb;
}

}  // extern "C"
