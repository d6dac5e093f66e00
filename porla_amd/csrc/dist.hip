// Multi-process form of the range-sharded MSM: one process per GPU, the 96-byte partial Jacobian sums exchanged with ONE
// ncclAllGather (RCCL over xGMI) issued from C++, then folded on the host -- the fold the reference performs across its 8 pool
// threads' partial sums (porla/Client/Client.hpp:761-787; SURVEY.md s8e: EC addition is not an RCCL reduce op, hence
// gather + fold).  RCCL is bound at run time (dlopen): libmultiexp.so keeps loading on a host without it, and only the
// porla_dist_* entry points fail -- loudly -- there.
#include "engine.hpp"
#include "../../include/porla_gpu.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; nothing links against librccl

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace porla;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional: only the timed-out initialisation's helper uses it
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct DistState {
    std::mutex mu;
    Rccl lib;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0, device = -1;
    bool poisoned = false;       // an initialisation timed out: a helper thread may still sit inside RCCL (see porla_dist_init)
    hipStream_t stream = nullptr;
    uint8_t* d_send = nullptr;   // 96 B
    uint8_t* d_recv = nullptr;   // world * 96 B
    uint8_t* h_recv = nullptr;   // pinned, world * 96 B
};
DistState D;

int load_rccl() {
    if (D.lib.handle) return PORLA_OK;
    // PORLA_RCCL_LIB names THE library to bind (nothing else is tried: an explicit choice that fails is an error, not a
    // reason to pick another build silently); without it the usual names
    const char* forced = getenv("PORLA_RCCL_LIB");
    const bool only_forced = forced && *forced;
    const char* names[] = {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        if (only_forced && nm != forced) break;
        void* h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) { D.lib.handle = h; break; }
        const char* e = dlerror();          // one call: it returns the message AND clears it
        tried += std::string(nm) + ": " + (e ? e : "?") + "; ";
    }
    if (!D.lib.handle) { set_last_error("porla: RCCL not found (" + tried + ")"); return PORLA_ERR_STATE; }
    auto sym = [&](const char* n) { return dlsym(D.lib.handle, n); };
    D.lib.GetUniqueId = (decltype(D.lib.GetUniqueId))sym("ncclGetUniqueId");
    D.lib.CommInitRank = (decltype(D.lib.CommInitRank))sym("ncclCommInitRank");
    D.lib.CommDestroy = (decltype(D.lib.CommDestroy))sym("ncclCommDestroy");
    D.lib.CommAbort = (decltype(D.lib.CommAbort))sym("ncclCommAbort");
    D.lib.AllGather = (decltype(D.lib.AllGather))sym("ncclAllGather");
    D.lib.GetErrorString = (decltype(D.lib.GetErrorString))sym("ncclGetErrorString");
    if (!D.lib.GetUniqueId || !D.lib.CommInitRank || !D.lib.CommDestroy || !D.lib.AllGather || !D.lib.GetErrorString) {
        set_last_error("porla: the RCCL library found lacks a required symbol");
        dlclose(D.lib.handle);
        D.lib = Rccl();
        return PORLA_ERR_STATE;
    }
    return PORLA_OK;
}

int nccl_fail(ncclResult_t r, const char* what) {
    char buf[256];
    snprintf(buf, sizeof buf, "RCCL error %d (%s) in %s", (int)r, D.lib.GetErrorString ? D.lib.GetErrorString(r) : "?", what);
    set_last_error(buf);
    return PORLA_ERR_HIP;
}

void free_buffers() {
    if (D.d_send) (void)hipFree(D.d_send);
    if (D.d_recv) (void)hipFree(D.d_recv);
    if (D.h_recv) (void)hipHostFree(D.h_recv);
    if (D.stream) (void)hipStreamDestroy(D.stream);
    D.d_send = D.d_recv = D.h_recv = nullptr;
    D.stream = nullptr;
}

// all ranks' partials into out (world * 96 bytes); D.mu held
int allgather_locked(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t* out) {
    if (D.poisoned) { set_last_error("porla: porla_dist_init timed out earlier in this process: leave with a non-zero _exit()"); return PORLA_ERR_STATE; }
    if (!D.comm) { set_last_error("porla: porla_dist_init first"); return PORLA_ERR_STATE; }
    int cur = -1;
    PORLA_HIP(hipGetDevice(&cur));
    if (cur != D.device) { set_last_error("porla: the communicator belongs to another device than the current one"); return PORLA_ERR_STATE; }
    PORLA_HIP(hipMemcpyAsync(D.d_send, partial, PORLA_JACOBIAN_BYTES, hipMemcpyHostToDevice, D.stream));
    ncclResult_t r = D.lib.AllGather(D.d_send, D.d_recv, PORLA_JACOBIAN_BYTES, ncclUint8, D.comm, D.stream);
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllGather");
    PORLA_HIP(hipMemcpyAsync(D.h_recv, D.d_recv, (size_t)D.world * PORLA_JACOBIAN_BYTES, hipMemcpyDeviceToHost, D.stream));
    PORLA_HIP(hipStreamSynchronize(D.stream));
    memcpy(out, D.h_recv, (size_t)D.world * PORLA_JACOBIAN_BYTES);
    return PORLA_OK;
}

template <int CURVE>
int gather_fold(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t out_affine[64]) {
    if (!partial || !out_affine) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    std::vector<uint8_t> all;
    {
        std::lock_guard<std::mutex> lk(D.mu);
        all.resize((size_t)(D.world > 0 ? D.world : 1) * PORLA_JACOBIAN_BYTES);
        int rc = allgather_locked(partial, all.data());
        if (rc) return rc;
    }
    return CURVE == 0 ? porla_bn254_jac_sum(all.data(), all.size() / PORLA_JACOBIAN_BYTES, out_affine)
                      : porla_secp256k1_jac_sum(all.data(), all.size() / PORLA_JACOBIAN_BYTES, out_affine);
}

}  // namespace

namespace porla {
// ranks of the in-library communicator (0 without one): msm_impl's implicit range split stays on the current device when > 1
int dist_world_size() {
    std::lock_guard<std::mutex> lk(D.mu);
    return D.comm ? D.world : 0;
}
}  // namespace porla

extern "C" {

int porla_dist_unique_id(uint8_t id_out[PORLA_DIST_ID_BYTES]) {
    static_assert(PORLA_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return PORLA_ERR_ARG;
    std::lock_guard<std::mutex> lk(D.mu);
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = D.lib.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return PORLA_OK;
}

int porla_dist_init(const uint8_t id_in[PORLA_DIST_ID_BYTES], int rank, int world) {
    if (!id_in || world < 1 || rank < 0 || rank >= world) { set_last_error("porla: bad argument"); return PORLA_ERR_ARG; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(D.mu);
    if (D.poisoned) {
        set_last_error("porla: an earlier porla_dist_init timed out in this process; leave with a non-zero _exit and start a fresh process");
        return PORLA_ERR_STATE;
    }
    if (D.comm) { set_last_error("porla: porla_dist_init called twice (porla_dist_finalize first)"); return PORLA_ERR_STATE; }
    if ((rc = load_rccl())) return rc;
    PORLA_HIP(hipGetDevice(&D.device));
    ncclUniqueId id;
    memcpy(id.internal, id_in, NCCL_UNIQUE_ID_BYTES);
    // ncclCommInitRank is collective and has no timeout of its own: a peer that never arrives would leave this rank waiting
    // forever.  It runs on a helper thread; if it has not returned after PORLA_DIST_INIT_TIMEOUT_S (default 180 s) the call
    // fails with PORLA_ERR_STATE.  The helper cannot be cancelled: it is marked ABANDONED (should RCCL hand it a communicator
    // later, it aborts that communicator at once -- no peer is left believing in a group this rank has given up), the state is
    // POISONED (every later porla_dist_* call fails: no second helper on the same id), and the caller must leave the process
    // with a non-zero _exit() -- a plain exit() would run static destructors under a thread that is still inside RCCL -- and
    // continue, if at all, in a fresh child process (never a re-exec of this one: it has initialised the GPU).
    double limit_s = 180.0;
    if (const char* t = getenv("PORLA_DIST_INIT_TIMEOUT_S")) { double v = atof(t); if (v > 0) limit_s = v; }
    struct Pending {
        std::mutex mu; std::condition_variable cv;
        bool done = false, abandoned = false;
        ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr;
    };
    auto pend = std::make_shared<Pending>();
    const int device = D.device;
    auto init_fn = D.lib.CommInitRank;
    auto abort_fn = D.lib.CommAbort ? D.lib.CommAbort : D.lib.CommDestroy;
    std::thread([pend, device, init_fn, abort_fn, world, id, rank]() {
        ncclComm_t c = nullptr;
        ncclResult_t r = hipSetDevice(device) == hipSuccess ? init_fn(&c, world, id, rank) : ncclUnhandledCudaError;
        std::unique_lock<std::mutex> lk(pend->mu);
        if (pend->abandoned) {                  // nobody will ever take this communicator: tear it down now
            lk.unlock();
            if (r == ncclSuccess && c) (void)abort_fn(c);
            return;
        }
        pend->r = r; pend->comm = c; pend->done = true;
        pend->cv.notify_all();
    }).detach();
    ncclComm_t comm = nullptr;
    {
        std::unique_lock<std::mutex> lk(pend->mu);
        if (!pend->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return pend->done; })) {
            pend->abandoned = true;
            D.poisoned = true;
            char buf[224];
            snprintf(buf, sizeof buf, "porla: ncclCommInitRank (rank %d of %d) did not return within %.0f s; the process must now "
                     "leave with a non-zero _exit()", rank, world, limit_s);
            set_last_error(buf);
            return PORLA_ERR_STATE;
        }
        if (pend->r != ncclSuccess) return nccl_fail(pend->r, "ncclCommInitRank");
        comm = pend->comm;
    }
    D.comm = comm; D.rank = rank; D.world = world;
    hipError_t e = hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&D.d_send, PORLA_JACOBIAN_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&D.d_recv, (size_t)world * PORLA_JACOBIAN_BYTES);
    if (e == hipSuccess) e = hipHostMalloc((void**)&D.h_recv, (size_t)world * PORLA_JACOBIAN_BYTES, hipHostMallocDefault);
    if (e != hipSuccess) {
        free_buffers();
        (void)D.lib.CommDestroy(D.comm);
        D.comm = nullptr; D.world = 0;
        return hip_fail(e, "porla_dist_init buffers", __FILE__, __LINE__);
    }
    return PORLA_OK;
}

int porla_dist_info(int* rank, int* world) {
    std::lock_guard<std::mutex> lk(D.mu);
    if (rank) *rank = D.rank;
    if (world) *world = D.comm ? D.world : 0;
    return PORLA_OK;
}

int porla_dist_finalize(void) {
    std::lock_guard<std::mutex> lk(D.mu);
    if (!D.comm) return PORLA_OK;
    free_buffers();
    ncclResult_t r = D.lib.CommDestroy(D.comm);
    D.comm = nullptr; D.world = 0; D.device = -1;
    if (r != ncclSuccess) return nccl_fail(r, "ncclCommDestroy");
    return PORLA_OK;
}

int porla_dist_allgather_partials(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t* all_out) {
    if (!partial || !all_out) { set_last_error("porla: null argument"); return PORLA_ERR_ARG; }
    std::lock_guard<std::mutex> lk(D.mu);
    return allgather_locked(partial, all_out);
}

int porla_bn254_dist_fold(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t out_affine[64]) { return gather_fold<0>(partial, out_affine); }
int porla_secp256k1_dist_fold(const uint8_t partial[PORLA_JACOBIAN_BYTES], uint8_t out_affine[64]) { return gather_fold<1>(partial, out_affine); }

int porla_bn254_msm_device_dist(const void* d_scalars, const void* d_points, size_t n_local, uint8_t out_affine[64], void* hip_stream) {
    uint8_t part[PORLA_JACOBIAN_BYTES];
    int rc = porla_bn254_msm_device_partial(d_scalars, d_points, n_local, part, hip_stream);
    if (rc) return rc;
    return gather_fold<0>(part, out_affine);
}
int porla_secp256k1_msm_device_dist(const void* d_scalars, const void* d_points, size_t n_local, uint8_t out_affine[64], void* hip_stream) {
    uint8_t part[PORLA_JACOBIAN_BYTES];
    int rc = porla_secp256k1_msm_device_partial(d_scalars, d_points, n_local, part, hip_stream);
    if (rc) return rc;
    return gather_fold<1>(part, out_affine);
}

}  // extern "C"
