// Internal C++ interface between the translation units of libmultiexp.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include "../../include/porla_gpu.h"
#include "host_curve.hpp"
#include "msm.hip.h"
#include "msm_small.hip.h"

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
    // dominant: the workload's dominant kernel -- the only scopes recorded at profile level 2 (bench.py's timed region:
    // every recorded scope costs two event packets on the stream, ~10 us of idle GPU between the kernels around it)
    ProfScope(const char* name, hipStream_t s, bool dominant = false);
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

// two-phase form (see msm_impl.hip.h)
template <class C>
int msm_pair_device(const uint8_t* d_scalars, const uint8_t* d_points_a, const uint8_t* d_points_b, size_t n, hipStream_t stream,
                    XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b);
template <class C>
int msm_pair_gather_device(const uint8_t* d_store_a, const uint8_t* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef, size_t n,
                           hipStream_t stream, XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b);
template <class C>
int msm_pair_gather_begin(int slot, const uint8_t* d_store_a, const uint8_t* d_store_b, const uint64_t* d_idx, const uint32_t* d_coef, size_t n,
                          hipStream_t stream);
template <class C>
int msm_pair_end(int slot, XYZZ<typename C::Fp>* total_a, XYZZ<typename C::Fp>* total_b);
template <class C>
int msm_pair_host(const uint8_t* scalars, const uint8_t* points_a, const uint8_t* points_b, size_t n, XYZZ<typename C::Fp>* total_a,
                  XYZZ<typename C::Fp>* total_b);
template <class C>
int msm_device_begin(int slot, const uint8_t* d_scalars, const uint8_t* d_points, size_t n, hipStream_t stream);
template <class C>
int msm_device_end(int slot, XYZZ<typename C::Fp>* total);

// host buffers -> device -> msm
template <class C>
int msm_host(const uint8_t* scalars, const uint8_t* points, size_t n, XYZZ<typename C::Fp>* total);
// host buffers, range-sharded: `shards` contiguous pair ranges spread over `devices` visible devices (one host thread,
// stream and workspace slot per device; the shards of one device are pipelined: the upload of a range overlaps the kernels
// of the previous one); the range totals are folded on the host as the reference folds its 8 pool threads' partial sums
// (porla/Client/Client.hpp:761-787).  shards / devices <= 0: automatic.
template <class C>
int msm_host_multi(const uint8_t* scalars, const uint8_t* points, size_t n, int shards, int devices,
                   XYZZ<typename C::Fp>* total);

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
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};
// Stream ordering of scratch that is shared between calls: the *_device entry points leave their kernels in flight on the
// caller's stream, and the next call may come on ANOTHER stream (or on the engine's own) and reuse the same scratch.  Every such
// scratch set carries a fence: enter() makes the new stream wait for the event the previous use recorded (no-op on the same
// stream, which orders itself); leave() records the event after the last kernel of this use.  Both are called under the
// mutex that serialises the enqueue.
struct UseFence {
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool used = false;
    int enter(hipStream_t s) {
        if (used && last != s) PORLA_HIP(hipStreamWaitEvent(s, ev, 0));
        return PORLA_OK;
    }
    int leave(hipStream_t s) {
        if (!ev) PORLA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        PORLA_HIP(hipEventRecord(ev, s));
        last = s;
        used = true;
        return PORLA_OK;
    }
    void reset() { if (ev) (void)hipEventDestroy(ev); ev = nullptr; used = false; last = nullptr; }
};

struct Workspace {
    std::mutex mu;   // held while kernels are enqueued on / results folded from this slot (the registry has its own lock)
    int device = -1;
    Buf pts, keys, entries, counts, starts, fill, cursor, buckets, in_scalars, in_points;
    Buf order, blk_hist, blk_off, tile_off, heavy, chunk_out;
    Buf small_part;               // single-launch path: the blocks' sums + the per-window arrival counters
    uint32_t small_seq = 0;
    Buf tree_s, tree_m, tree_mt;  // bucket reduction tree: S levels, M ping-pong halves (all windows / tail-private)
    Buf multi_acc;                // msm_host_multi: the bucket sums of a device's ranges, merged before one reduction
    hipEvent_t merged = nullptr;  // ... recorded after this slot's range has been merged into multi_acc
    void* h_windows = nullptr;  // pinned, device-mapped: the tree's last level writes [W][c] (S, M_0 .. M_(c-2)) per window
    size_t h_windows_cap = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream = nullptr;            // second stream of the reduction tree (msm_tree_launch)
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    hipEvent_t front_fork_ev = nullptr, front_join_ev = nullptr;   // points_to_mont beside the digit / sort chain (msm_launch)
    hipEvent_t caller_ev = nullptr;  // order_after_caller: the caller's stream at the time of the call
    int slot = 0;
    hipEvent_t done = nullptr;  // recorded after the last kernel + D2H copy of a launched MSM
    int pend_W = 0, pend_c = 0; // window count / width of the launched, not yet folded MSM (0 = none)
    bool lone = false;          // the caller waits for this MSM (blocking entry points): the reduction tree may use two streams
    bool pair_pending = false;  // the begun launch is a pair (msm_pair_gather_begin): two result regions, msm_pair_end collects them
    bool begun = false;         // two-phase API: a begin without its end (also set for n == 0, where pend_W stays 0)
};
// workspace slots per device -- 0: blocking calls, 1..3: the two-phase C ABI, 4..7: msm_host_multi's pipeline,
// 8..15: taken by blocking calls that find slot 0 busy (the reference issues its IPA MSMs from 8 pool threads at once,
// Client.hpp:395,778: they run side by side, each on its own slot and stream, instead of queueing on one mutex),
// 16: the pair of MSMs inside porla_kzg_audit_device
constexpr int MSM_SLOTS = 17;
constexpr int MSM_POOL_SLOTS = 8;
constexpr int MSM_AUDIT_SLOT = 16;
constexpr int MSM_USER_SLOTS = 4;
constexpr int MSM_MULTI_SLOT0 = 4;
constexpr int MSM_MULTI_SLOTS = 4;
constexpr int MSM_POOL_SLOT0 = 8;
// Makes `a` (and `b`, if given) wait for everything enqueued so far on `caller` -- the stream the entry point was handed, which
// may be the null stream -- so that work the library runs on its own non-blocking streams sees the caller's earlier copies and
// kernels (the audit entry points read index / coefficient arrays the caller may just have uploaded asynchronously).
int order_after_caller(Workspace* ws, hipStream_t caller, hipStream_t a, hipStream_t b = nullptr);
constexpr size_t MSM_SCAN_MAX = 1u << 16;  // inputs up to this size are scanned for their longest scalar first
constexpr size_t MSM_RANGE_MAX = 1u << 22; // larger inputs run as ranges of this size into one bucket array (msm_launch)
// Batched fixed-base commitments (fixed_base.hip.h): resident table of window multiples of one base.
// Calls on one object are serialised by `mu`; the *_device form leaves its kernels in flight on the caller's stream, and a
// later call on another stream waits (on the device, through `fence`) for them before it reuses the slice-partial scratch.
template <class C>
struct FixedBase {
    int device = -1;
    int c = 0, W = 0;
    size_t n_points = 0;
    Affine<typename C::Fp>* table = nullptr;
    size_t table_cap = 0;                     // bytes
    XYZZ<typename C::Fp>* pow_buf = nullptr;  // construction buffers, kept between builds when keep_build_buffers
    size_t pow_cap = 0;
    XYZZ<typename C::Fp>* scratch_buf = nullptr;
    size_t scratch_cap = 0;
    bool keep_build_buffers = false;          // true: a base that changes per call (mac_fft.hip); false: one-off (SRS)
    XYZZ<typename C::Fp>* partial = nullptr;
    size_t partial_cap = 0;
    uint8_t* io_rows = nullptr;
    size_t io_rows_cap = 0;
    uint8_t* io_out = nullptr;
    size_t io_out_cap = 0;
    hipStream_t io_stream2 = nullptr;         // commit_host: the second of the two streams the chunks alternate on
    uint32_t last_S = 1;                      // slices per row of the last commit_device (layout of `partial`)
    UseFence fence;                           // orders `partial` (and the table after a rebuild) between calls on different streams
    static constexpr size_t HOST_FINISH_MAX_ROWS = 256;   // batches up to this size are normalised on the host (commit_host)
    // single-launch path for a handful of host rows (fixed_base.hip.h:k_fb_commit_small): pinned staging (header, row sums,
    // rows) + the blocks' partial sums and arrival counters in HBM
    void* h_small = nullptr;
    void* d_small = nullptr;
    uint32_t small_seq = 0;
    hipEvent_t small_done = nullptr;
    std::mutex mu;
    void release();
    int build(const Affine<typename C::Fp>* d_base, size_t n, int window_bits, hipStream_t stream);
    int build_from_host_bytes(const uint8_t* points_be, size_t n, int window_bits, hipStream_t stream);
    // d_out == nullptr: stop after the slice fold; the row sums stay in `partial` (row r at partial[r * last_S])
    // guest_room: launch the commitment kernel in its two-waves-per-SIMD form, which leaves register room for one wave of a
    // kernel running beside it (fixed_base.hip.h:k_fb_commit<C, true>)
    int commit_device(const uint8_t* d_rows, size_t n_rows, size_t n_coeffs, size_t row_stride, uint8_t* d_out,
                      hipStream_t stream, bool guest_room = false);
    int commit_host(const uint8_t* rows, size_t n_rows, size_t n_coeffs, size_t row_stride, uint8_t* out,
                    hipStream_t stream);
    // <= FB_SMALL_MAX_ROWS rows given by pointer (they need not be contiguous: the coalescing front of compute_digest_from_srs
    // hands over the callers' own buffers), one launch, results polled from pinned memory and normalised on the host
    // d_rows != nullptr: the rows are resident on the device (contiguous, 32 * n_coeffs bytes each) and row_ptrs is ignored
    // raw_sums != nullptr: the rows' projective sums are handed back as they are (outs is ignored): the caller normalises them
    // together with other points (one inversion for all)
    int commit_small(const uint8_t* const* row_ptrs, size_t n_rows, size_t n_coeffs, uint8_t* const* outs, hipStream_t stream,
                     const uint8_t* d_rows = nullptr, XYZZ<typename C::Fp>* raw_sums = nullptr);
    static bool small_ok(size_t n_rows, size_t n_coeffs);
};

extern std::mutex g_ws_mu;   // the workspace registry (lookup / creation / release); a slot's use is under Workspace::mu
extern int g_window_override;
extern int g_small_mode;      // single-launch path for n <= SMALL_MAX_N (32 768): 1 on (default), 0 off
extern int g_small_c;         // its window bits, 0 = automatic
extern int g_last_multi[2];   // ranges, devices of the most recent msm_host_multi (diagnostic)
extern int g_last_shape[3];   // window bits, window count, GLV flag of the most recently launched MSM (diagnostic)
extern int g_use_glv;  // 1: GLV split of every scalar; 0: plain signed windows over the full scalar; -1: the curve's default
int get_workspace(Workspace** out);                 // slot 0 of the current device
int get_workspace_slot(int slot, Workspace** out);  // takes g_ws_mu itself
int lease_blocking_slot(Workspace** out);           // slot 0, or a free pool slot when it is busy; returned LOCKED (ws->mu)
hipStream_t engine_stream();  // this device's engine-owned non-blocking stream

// icc.hip: wt = w^reverse_bits(write_step % n_total, height - 1) mod p_icc as a 32-byte big-endian integer (the MAC-side scalar of
// Server::HAdd / Client::HAdd / CRebuild's Y halves)
int icc_wt_scalar_be(size_t n_total, unsigned long long write_step, uint8_t out[32]);
// icc.hip: the ICC butterfly network as an n x n matrix of 32-byte big-endian coefficients mod the group order
int icc_network_matrix_device(int curve, size_t n, unsigned long long write_step, int part, uint8_t* d_rows_out,
                              hipStream_t stream);

}  // namespace porla
