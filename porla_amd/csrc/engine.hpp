// Internal C++ interface between the translation units of libmultiexp.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include "../../include/porla_gpu.h"
#include "host_curve.hpp"
#include "msm.cuh"

namespace porla {

void set_last_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define PORLA_HIP(call)                                                      \
    do {                                                                     \
        hipError_t _e = (call);                                              \
        if (_e != hipSuccess) return ::porla::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

// profiling slots (HIP-event timing per kernel on the launch stream)
struct ProfScope {
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
    int slot;
    hipStream_t stream;
    hipEvent_t e0, e1;
    bool on;
};
void prof_flush();  // resolve pending events into the accumulators (synchronises them)

// Device-resident MSM over one curve.  d_scalars: n*32 B big-endian, d_points: n*64 B big-endian affine.
// The folded total (XYZZ, Montgomery limbs) is returned on the host.
template <class C>
int msm_device(const uint8_t* d_scalars, const uint8_t* d_points, size_t n, hipStream_t stream,
               XYZZ<typename C::Fp>* total);

// Same, but the points are already Montgomery limbs in HBM (resident fixed base, e.g. the SRS).
template <class C>
int msm_device_mont(const uint8_t* d_scalars, const Affine<typename C::Fp>* d_points_mont, size_t n,
                    hipStream_t stream, XYZZ<typename C::Fp>* total);

// host scalars (n*32 B big-endian) against a resident Montgomery-form base in HBM
template <class C>
int msm_host_scalars(const uint8_t* scalars, const Affine<typename C::Fp>* d_points_mont, size_t n,
                     XYZZ<typename C::Fp>* total);

// host buffers -> device -> msm
template <class C>
int msm_host(const uint8_t* scalars, const uint8_t* points, size_t n, XYZZ<typename C::Fp>* total);

int ensure_device();  // selects/validates the current device, fails loudly without one

// ---- per-device scratch, shared by the per-curve translation units
struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PORLA_OK;
        if (p) PORLA_HIP(hipFree(p));
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        PORLA_HIP(hipMalloc(&p, want));
        cap = want;
        return PORLA_OK;
    }
};
struct Workspace {
    int device = -1;
    Buf pts, keys, entries, counts, starts, fill, cursor, buckets, partial, windows, in_scalars, in_points;
    Buf order, blk_hist, blk_off, tile_off;
    void* h_windows = nullptr;  // pinned
    size_t h_windows_cap = 0;
    hipStream_t own_stream = nullptr;
};
extern std::mutex g_ws_mu;
extern int g_window_override;
extern int g_legacy_sort;  // PORLA_LEGACY_SORT=1: global-atomic counting sort (kept for A/B timing)
int get_workspace(Workspace** out);

}  // namespace porla
