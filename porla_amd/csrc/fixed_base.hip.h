// Batched fixed-base commitments on gfx950 (MI355X): many rows of coefficients against ONE resident base.
//
// This is what a large Porla encode actually executes: `compute_digest_from_srs`
// (porla/main.go:103-116, kzg.Commit = 128-point MSM against the fixed SRS) is called twice per last-stage butterfly
// and per `mix` output row (porla/Server/Server.hpp:550-560, 1077-1078, 2061-2062); the IPA twin is the Pedersen
// commitment over the fixed generators (`compute_commitment`, porla/Client/Client.hpp:374-406 ->
// secp256k1_ecmult_multi_var over 128 fixed points).  The reference runs one 128-point Pippenger/Strauss per row.
//
// MI355X-first design: the base never changes, HBM is 288 GB, so trade memory for arithmetic.  For every base point
// G_i and every c-bit window w the table holds the affine multiples (k+1) * 2^(c*w) * G_i, k < 2^(c-1) (signed
// digits halve the table).  A commitment is then n_coeffs * W mixed additions and NO doublings and NO bucket
// pass: c = 16 -> 16 windows -> 2048 additions per 128-coefficient row (a per-row Pippenger needs ~8000 group
// operations), table = 128 * 16 * 2^15 * 64 B = 4.3 GB.  One lane owns one row (optionally one slice of a row), so
// the 64 lanes of a wave walk (i, w) in lock-step and their gathers fall into the same 2 MiB sub-table.
//
//   k_fb_base_powers   thread per base point: 2^(c*w) * G_i for all w (c*W dependent doublings)
//   k_fb_multiples     wave per run of entries, lanes interleaved: (m+1)*B by double-and-add, then additions of 64*B -> XYZZ scratch
//   k_fb_normalize     XYZZ -> affine with one inversion per 32 entries (Montgomery's trick) -> table
//   k_fb_commit        lane = row (x slice): scalar -> reduce mod order -> signed digits -> gather + 8M+2S mixed add
//   k_fb_fold_quad     fold a row's slice partials (reduced-radix memory form): in-place pairwise tree, four lanes per addition
//                      (k_fb_fold: the same with G lanes per row on 8 x 32-bit limbs, for a curve without the reduced-radix form)
//   k_fb_finish        lane = 1, 2 or 4 rows (by batch size): one division-step inversion (fe_inv_safegcd), Montgomery -> big-endian X||Y (64 zero bytes = infinity)
//   k_fb_commit_small  <= 64 host rows in ONE launch (block = row x slice), sums polled from pinned memory
#pragma once
#include "msm.hip.h"
#include "msm_small.hip.h"
#include "inv30.hip.h"
#include <type_traits>

namespace porla {

// ---------------------------------------------------------------- device-side inversion (Fermat, a^(p-2))
template <class M>
__device__ __noinline__ Fe<M> fe_inv_dev(Fe<M> a) {
    // exponent p - 2, scanned from the top bit
    Fe<M> acc = fe_one<M>();
#pragma unroll
    for (int l = 7; l >= 0; l--) {
        const uint32_t limb = (l == 0) ? M::P[0] - 2u : M::P[l];  // P is odd and P[0] >= 2: no borrow past limb 0
#pragma unroll 1
        for (int b = 31; b >= 0; b--) {
            acc = fe_sqr_call<M>(acc);
            if ((limb >> b) & 1) acc = fe_mul_call<M>(acc, a);
        }
    }
    return acc;
}

// XYZZ -> affine on the device (cold path): x = X * (ZZ*I)^2, y = Y * I with I = 1/ZZZ  (ZZ^3 = ZZZ^2)
template <class M>
__device__ __forceinline__ Affine<M> xyzz_to_affine_with_inv(const XYZZ<M>& p, const Fe<M>& inv_zzz) {
    Affine<M> r;
    Fe<M> t = fe_mul_call<M>(p.zz, inv_zzz);     // ZZ/ZZZ = 1/Z
    Fe<M> t2 = fe_sqr_call<M>(t);                // 1/ZZ
    r.x = fe_mul_call<M>(p.x, t2);
    r.y = fe_mul_call<M>(p.y, inv_zzz);
    return r;
}

template <class M>
__device__ __forceinline__ void store_affine(Affine<M>* dst, const Affine<M>& a) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    store_fe<M>(d, a.x);
    store_fe<M>(d + 8, a.y);
}

// ---------------------------------------------------------------- table construction
// pow[i*W + w] = 2^(c*w) * base[i]
template <class C>
__global__ void __launch_bounds__(64)
k_fb_base_powers(const Affine<typename C::Fp>* __restrict__ base, uint32_t n_points, int c, int W,
                 XYZZ<typename C::Fp>* __restrict__ pow) {
    using M = typename C::Fp;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    XYZZ<M> p = xyzz_from_affine<M>(load_affine<M>(base, i));
    for (int w = 0; w < W; w++) {
        store_xyzz<M>(pow + (size_t)i * W + w, p);
        if (w + 1 < W)
            for (int d = 0; d < c; d++) xyzz_double_cold<M>(&p);
    }
}

// scratch[(pair - pair0) * H + k] = (k+1) * pow[pair], k < H.  A group of `lanes` lanes owns a run of `lanes * K`
// consecutive entries of one pair, INTERLEAVED (lane l: entries l, l + lanes, l + 2 lanes, ...): a store of the group covers
// consecutive entries (8 KiB when lanes = 64), and a lane still pays one addition per entry (of lanes * B) after its
// double-and-add start.  Small tables (H < 1024) put 64 / lanes pairs in a wave so that K stays 16.
template <class C>
__global__ void __launch_bounds__(64)
k_fb_multiples(const XYZZ<typename C::Fp>* __restrict__ pow, uint32_t pair0, uint32_t n_pairs, uint32_t H, uint32_t lanes,
               uint32_t K, XYZZ<typename C::Fp>* __restrict__ scratch) {
    using M = typename C::Fp;
    const uint32_t run = lanes * K;                 // entries per lane group
    const uint32_t runs = H / run;                  // groups per pair
    const uint32_t groups_per_wave = 64 / lanes;
    const uint32_t group = blockIdx.x * groups_per_wave + threadIdx.x / lanes;
    if (group >= n_pairs * runs) return;
    const uint32_t pair = group / runs, j = group % runs, lane = threadIdx.x % lanes;
    XYZZ<M> Bp = load_xyzz<M>(pow + pair0 + pair);
    // first = (j*run + lane + 1) * B by left-to-right double-and-add
    const uint32_t m = j * run + lane + 1;
    XYZZ<M> acc = xyzz_inf<M>();
    for (int bit = 31 - __clz(m); bit >= 0; bit--) {
        xyzz_double_cold<M>(&acc);
        if ((m >> bit) & 1) xyzz_add_cold<M>(&acc, &Bp);
    }
    XYZZ<M> step = Bp;                               // lanes * B (lanes is a power of two)
    for (uint32_t l = lanes; l > 1; l >>= 1) xyzz_double_cold<M>(&step);
    XYZZ<M>* dst = scratch + (size_t)pair * H + (size_t)j * run + lane;
    for (uint32_t k = 0; k < K; k++) {
        store_xyzz<M>(dst + (size_t)k * lanes, acc);
        if (k + 1 < K) xyzz_add_cold<M>(&acc, &step);
    }
}

// table[k] = affine(scratch[k]); one inversion per NB entries of a lane (Montgomery's trick; the running prefix products are
// staged in the x half of the lane's own output slots, which are overwritten by the result in the second pass).  A wave owns
// 64 * NB consecutive entries, lanes interleaved, so loads and stores are coalesced.
template <class C>
__global__ void __launch_bounds__(64)
k_fb_normalize(const XYZZ<typename C::Fp>* __restrict__ scratch, size_t n, Affine<typename C::Fp>* __restrict__ table) {
    using M = typename C::Fp;
    constexpr int NB = 32;
    const size_t base = (size_t)blockIdx.x * 64 * NB + threadIdx.x;
    if (base >= n) return;
    int cnt = 0;
    while (cnt < NB && base + (size_t)cnt * 64 < n) cnt++;
    Fe<M> run = fe_one<M>();
#pragma unroll 1
    for (int k = 0; k < cnt; k++) {
        const size_t e = base + (size_t)k * 64;
        Fe<M> z = load_fe<M>(reinterpret_cast<const uint32_t*>(scratch + e) + 24);  // zzz
        if (fe_is_zero<M>(z)) z = fe_one<M>();
        store_fe<M>(reinterpret_cast<uint32_t*>(table + e), run);
        run = fe_mul_call<M>(run, z);
    }
    Fe<M> inv;
    if constexpr (C::F30_BUCKETS) inv = fe_inv_safegcd<M>(run);
    else inv = fe_inv_dev<M>(run);
#pragma unroll 1
    for (int k = cnt - 1; k >= 0; k--) {
        const size_t e = base + (size_t)k * 64;
        XYZZ<M> p = load_xyzz<M>(scratch + e);
        Affine<M> a;
        if (xyzz_is_inf<M>(p)) {
            a.x = fe_zero<M>(); a.y = fe_zero<M>();
        } else {
            Fe<M> prefix = load_fe<M>(reinterpret_cast<const uint32_t*>(table + e));
            Fe<M> iz = fe_mul_call<M>(inv, prefix);
            inv = fe_mul_call<M>(inv, p.zzz);
            a = xyzz_to_affine_with_inv<M>(p, iz);
            if constexpr (C::F30_BUCKETS) {
                // the table feeds k_fb_commit's reduced-radix accumulation: residues in the 2^270 Montgomery form
                Fe<M> r1;
#pragma unroll
                for (int q = 0; q < 8; q++) r1.v[q] = M::R1_30[q];
                a.x = fe_mul_call<M>(a.x, r1);
                a.y = fe_mul_call<M>(a.y, r1);
            }
        }
        store_affine<M>(table + e, a);
    }
}

// ---------------------------------------------------------------- the commitment kernel
// rows: row r, coefficient i at rows + r*row_stride + 32*i, 32 bytes big-endian (bn254_scalar, utils.h:307-318;
//       fr.SetBytes semantics: reduced mod the group order, porla/main.go:110).
// grid.x covers rows (lane = row), grid.y = slice s of the coefficient range; partial[r*S + s] = slice sum.
// (three waves per SIMD at ~161 registers; held to 128 registers for four waves it measured SLOWER: 19.7-20.0 against 18.95-19.1 ms
// for 2^17 rows on one box, profiles/r03_q_fb_commit_4waves_ab.txt)
// GUEST_ROOM: the kernel declares 192 vector registers (it uses ~161): a SIMD then holds TWO of its waves instead of three and
// keeps 128 registers free -- room for one wave of another kernel.  The last stage of a CRebuild runs the MAC butterflies (one
// latency-bound wave per SIMD, 123 registers, 15 dependent launches) BESIDE the commitments: with three 161-register waves per
// SIMD the chip had no slot for them, and every one of the 15 stage launches waited for a block of this kernel to retire
// (3.6-4.8 ms instead of 0.6): the two streams ran one after the other.
template <class C, bool GUEST_ROOM = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GUEST_ROOM ? 2 : C::FB_COMMIT_WAVES, 4)))
k_fb_commit(const uint8_t* __restrict__ rows, uint32_t n_rows, uint32_t n_coeffs, size_t row_stride,
            const Affine<typename C::Fp>* __restrict__ table, int c, int W, uint32_t S,
            XYZZ<typename C::Fp>* __restrict__ partial) {
    using M = typename C::Fp;
    if constexpr (GUEST_ROOM) asm volatile("" ::: "v191");
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y;
    if (r >= n_rows) return;
    const uint32_t per = (n_coeffs + S - 1) / S;
    const uint32_t i0 = s * per;
    const uint32_t i1 = (i0 + per < n_coeffs) ? i0 + per : n_coeffs;
    const uint32_t Bh = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    const uint8_t* row = rows + (size_t)r * row_stride;
    // accumulator: the reduced-radix form of ec30.hip.h where the curve has it (the table is then in the 2^270 form too)
    using Acc = typename std::conditional<C::F30_BUCKETS, XYZZ30<M>, XYZZ<M>>::type;
    Acc acc;
    if constexpr (C::F30_BUCKETS) acc.inf = true;
    else acc = xyzz_inf<M>();
    // One software pipeline over all (coefficient, window) pairs of the lane: the gather of a window's table entry is issued
    // before the addition of the previous one -- also across the boundary between two coefficients -- and the next
    // coefficient's 32 bytes are requested a whole coefficient ahead (a pipeline that restarted per coefficient left one
    // gather and one row read exposed per 15 additions).
    // (reduced-radix form: the accumulator changes sign with every addition, ec30.hip.h:xyzz30_madd_flip -- the sign it carries is
    // folded into the digit's sign of the incoming entry and undone once at the end)
    bool flip = false;
    auto madd_entry = [&](const Affine<M>& e, bool neg) {
        if constexpr (C::F30_BUCKETS) {
            if (aff_is_inf<M>(e)) return;
            Affine<M> a = aff_neg_if<M>(e, xyzz30_flip_neg<M>(neg, flip));
            xyzz30_madd_flip<M>(acc, flip, f30_from_fe<M>(a.x), f30_from_fe<M>(a.y));
        } else {
            xyzz_madd<M>(acc, aff_neg_if<M>(e, neg));
        }
    };
    Affine<M> cur;
    bool cur_valid = false, cur_neg = false;
    Affine<M> mid;
    bool mid_valid = false, mid_neg = false;
    uint32_t tn[8];
    if (i0 < i1) load_be256(tn, row + (size_t)i0 * 32);
    for (uint32_t i = i0; i < i1; i++) {
        uint32_t t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = tn[k];
        if (i + 1 < i1) load_be256(tn, row + (size_t)(i + 1) * 32);
        for (int q = 0; q < C::MAX_Q; q++) {
            uint32_t d[8];
            uint32_t br = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint64_t x = (uint64_t)t[k] - C::ORDER[k] - br;
                d[k] = (uint32_t)x;
                br = (uint32_t)(x >> 63);
            }
            if (br) break;
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = d[k];
        }
        const Affine<M>* tab_i = table + (size_t)i * W * Bh;
        uint32_t carry = 0;
        for (int w = 0; w < W; w++) {
            Affine<M> nxt;
            bool nxt_valid = false, nxt_neg = false;
            const int lo = w * c;
            uint32_t raw = 0;
            if (lo < 256) {
                const int limb = lo >> 5, sh = lo & 31;
                // the two words the window straddles: `limb` is the same for every lane (it depends on w and c only), so this is a
                // scalar branch to one of eight copies -- a select chain over the words cost 16 instructions per window
                uint32_t a = 0, b = 0;
                switch (limb) {
                    case 0: a = t[0]; b = t[1]; break;
                    case 1: a = t[1]; b = t[2]; break;
                    case 2: a = t[2]; b = t[3]; break;
                    case 3: a = t[3]; b = t[4]; break;
                    case 4: a = t[4]; b = t[5]; break;
                    case 5: a = t[5]; b = t[6]; break;
                    case 6: a = t[6]; b = t[7]; break;
                    default: a = t[7]; b = 0; break;
                }
                uint64_t v = ((uint64_t)b << 32) | a;
                raw = (uint32_t)(v >> sh) & mask;
            }
            raw += carry;
            uint32_t mag;
            if (raw > Bh) { carry = 1; mag = (1u << c) - raw; nxt_neg = true; }
            else { carry = 0; mag = raw; }
            if (mag) {
                nxt_valid = true;
                nxt = load_affine<M>(tab_i + (size_t)w * Bh, mag - 1);
            }
            // two gathers in flight: the entry added now was requested two windows ago (18.6 -> 18.4 ms for 2^17 rows against one
            // in flight, profiles/r03_q_fb_commit_4waves_ab.txt; 160 registers, still three waves per SIMD)
            if (cur_valid) madd_entry(cur, cur_neg);
            if (mid_valid) cur = mid;
            cur_valid = mid_valid; cur_neg = mid_neg;
            if (nxt_valid) mid = nxt;
            mid_valid = nxt_valid; mid_neg = nxt_neg;
        }
    }
    if (cur_valid) madd_entry(cur, cur_neg);
    if (mid_valid) madd_entry(mid, mid_neg);
    if constexpr (C::F30_BUCKETS) {
        xyzz30_flip_finish<M>(acc, flip);
        // a slice partial goes to k_fb_fold in the reduced-radix memory form (no conversion products here, reduced-radix additions
        // there); a whole row's sum (S = 1) in the 2^256 form k_fb_finish and the host read
        if (C::F30_LAZY && S > 1) xyzz30_store_lazy<M>(partial + (size_t)r * S + s, acc);
        else store_xyzz<M>(partial + (size_t)r * S + s, xyzz30_to_xyzz<M>(acc));
    } else {
        store_xyzz<M>(partial + (size_t)r * S + s, acc);
    }
}

// ---------------------------------------------------------------- a handful of rows, latency-bound (the reference's real pattern)
// compute_digest_from_srs is called ONE row at a time, from up to 8 pool threads (porla/Server/Server.hpp:550-560,
// 1077-1078, 2061-2062), and create_proof commits two rows (main.go:164,170).  Through the batch kernels above a single row
// costs three launches, two copies and 15 dependent additions per lane (0.28 ms).  Here ONE launch does it:
//   block (row, slice) owns a slice of the row's (coefficient, window) pairs: a lane recodes the digit of each of its <= ~2
//   pairs (the carry into a window from one masked compare, as in msm_small.hip.h), gathers the table entry and accumulates;
//   the 256 lane sums are folded in LDS with four lanes per addition (8 levels); the LAST block of a row to arrive folds the
//   slices' sums and writes the row's projective sum into pinned host memory; the last row to finish publishes the sequence
//   number the host polls for.  The rows are read where they lie: `rows` may be device memory or mapped pinned host memory
//   (4 KB per row over PCIe costs less than a copy packet).  The host normalises (one inversion per row, ~25 us).
constexpr int FB_SMALL_DONE_SLOT = FB_SMALL_MAX_ROWS;      // counters[0 .. rows): slices arrived; [FB_SMALL_MAX_ROWS]: rows finished

template <class C>
__global__ void __launch_bounds__(SMALL_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_fb_commit_small(const uint8_t* __restrict__ rows, uint32_t n_rows, uint32_t n_coeffs, size_t row_stride,
                  const Affine<typename C::Fp>* __restrict__ table, int c, int W, uint32_t SL,
                  XYZZ<typename C::Fp>* __restrict__ part, uint32_t* __restrict__ counters, uint32_t* __restrict__ hdr,
                  XYZZ<typename C::Fp>* __restrict__ out_sums, uint32_t seq) {
    using M = typename C::Fp;
    __shared__ XYZZ<M> pts[SMALL_THREADS];
    __shared__ uint32_t pat[8];
    __shared__ uint32_t last_flag;
    const uint32_t tid = threadIdx.x;
    const uint32_t r = blockIdx.x / SL, sl = blockIdx.x % SL;
    if (tid < 8) {                                             // bits p of limb tid with p mod c == c - 1
        uint32_t v = 0;
        for (uint32_t b = (uint32_t)c - 1 - (32 * tid) % (uint32_t)c; b < 32; b += (uint32_t)c) v |= 1u << b;
        pat[tid] = v;
    }
    __syncthreads();
    const uint32_t P = n_coeffs * (uint32_t)W;
    const uint32_t p0 = (uint32_t)((uint64_t)sl * P / SL), p1 = (uint32_t)((uint64_t)(sl + 1) * P / SL);
    const uint32_t Bh = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    const uint8_t* row = rows + (size_t)r * row_stride;
    XYZZ30<M> acc;
    acc.inf = true;
    bool flip = false;
    for (uint32_t p = p0 + tid; p < p1; p += SMALL_THREADS) {
        const uint32_t i = p / (uint32_t)W, w = p % (uint32_t)W;
        uint32_t t[8];
        load_be256(t, row + (size_t)i * 32);
        for (int q = 0; q < C::MAX_Q; q++) {                       // fr.SetBytes: reduced mod the group order (main.go:110)
            uint32_t d[8];
            uint32_t br = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint64_t x = (uint64_t)t[k] - C::ORDER[k] - br;
                d[k] = (uint32_t)x;
                br = (uint32_t)(x >> 63);
            }
            if (br) break;
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = d[k];
        }
        const uint32_t wc = w * (uint32_t)c;
        bool gt = false, eq = true;                                // carry into window w (see msm_small.hip.h)
#pragma unroll
        for (int q = 7; q >= 0; q--) {
            const uint32_t below = wc > 32u * q ? wc - 32u * q : 0u;
            const uint32_t m = below >= 32u ? 0xffffffffu : ((1u << below) - 1u);
            const uint32_t a = t[q] & m, b = pat[q] & m;
            gt = eq ? (a > b) : gt;
            eq = eq && (a == b);
        }
        uint32_t rawd = gt ? 1u : 0u;
        if (wc < 256u) {
            const uint32_t limb = wc >> 5, sh = wc & 31u;
            uint32_t a = 0, b = 0;
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) {
                a = (q == limb) ? t[q] : a;
                b = (q == limb + 1) ? t[q] : b;
            }
            rawd += (uint32_t)((((uint64_t)b << 32) | a) >> sh) & mask;
        }
        uint32_t mag = rawd;
        bool neg = false;
        if (rawd > Bh) { mag = (1u << c) - rawd; neg = true; }
        if (!mag) continue;
        Affine<M> a = load_affine<M>(table + ((size_t)i * W + w) * Bh, mag - 1);
        if (aff_is_inf<M>(a)) continue;
        static_assert(C::F30_BUCKETS, "k_fb_commit_small needs the reduced-radix form");
        // (the sign-alternating addition with one reduction for Y3, ec30.hip.h:xyzz30_madd_flip: a lone wave pays for every instruction)
        a = aff_neg_if<M>(a, xyzz30_flip_neg<M>(neg, flip));
        xyzz30_madd_flip<M>(acc, flip, f30_from_fe<M>(a.x), f30_from_fe<M>(a.y));
    }
    xyzz30_flip_finish<M>(acc, flip);
    xyzz30_store_lazy<M>(&pts[tid], acc);
    __syncthreads();
    for (uint32_t h = 1; h < SMALL_THREADS; h <<= 1) {            // 256 lane sums -> pts[0]
        small_quad_adds<M>(SMALL_THREADS / (2 * h), [&](uint32_t t, const XYZZ<M>*& pa, const XYZZ<M>*& pb, XYZZ<M>*& out) {
            pa = &pts[t * 2 * h]; pb = pa + h; out = &pts[t * 2 * h];
        });
        __syncthreads();
    }
    if (tid < 8) reinterpret_cast<uint4*>(part + blockIdx.x)[tid] = reinterpret_cast<const uint4*>(pts)[tid];
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const uint32_t old = atomicAdd(&counters[r], 1u);
        last_flag = (old == SL - 1) ? 1u : 0u;
        if (last_flag) counters[r] = 0;
    }
    __syncthreads();
    if (!last_flag) return;
    __threadfence();
    if (tid < SL * 8) reinterpret_cast<uint4*>(pts)[tid] = reinterpret_cast<const uint4*>(part + (size_t)r * SL)[tid];
    __syncthreads();
    for (uint32_t cnt = SL; cnt > 1;) {
        const uint32_t half = (cnt + 1) / 2, pairs = cnt - half;
        small_quad_adds<M>(pairs, [&](uint32_t t, const XYZZ<M>*& pa, const XYZZ<M>*& pb, XYZZ<M>*& out) {
            pa = &pts[t]; pb = &pts[t + half]; out = &pts[t];
        });
        __syncthreads();
        cnt = half;
    }
    if (tid == 0) Node<C>::store_final(out_sums + r, Node<C>::load(&pts[0]));
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const uint32_t old = atomicAdd(&counters[FB_SMALL_DONE_SLOT], 1u);
        if (old == n_rows - 1) {
            counters[FB_SMALL_DONE_SLOT] = 0;
            __threadfence_system();
            __atomic_store_n(&hdr[0], seq, __ATOMIC_RELEASE);
        }
    }
}

// G lanes per row (G a power of two, 2 <= G <= 64, G <= S): fold the S slice partials of a row into partial[row * S]
template <class C>
__global__ void __launch_bounds__(64)
k_fb_fold(XYZZ<typename C::Fp>* __restrict__ partial, uint32_t n_rows, uint32_t S, uint32_t G) {
    using M = typename C::Fp;
    const uint32_t rows_per_wave = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t row = blockIdx.x * rows_per_wave + lane / G;
    const uint32_t sub = lane % G;
    XYZZ<M> acc = xyzz_inf<M>();
    if (row < n_rows) {
        for (uint32_t s = sub; s < S; s += G) {
            XYZZ<M> p = load_xyzz<M>(partial + (size_t)row * S + s);
            xyzz_add_cold<M>(&acc, &p);
        }
    }
#pragma unroll 1
    for (uint32_t m = G >> 1; m >= 1; m >>= 1) {
        XYZZ<M> o = xyzz_shfl_xor<M>(acc, (int)m);
        xyzz_add_cold<M>(&acc, &o);
    }
    if (sub == 0 && row < n_rows) store_xyzz<M>(partial + (size_t)row * S, acc);
}

// The same fold where the partials are in the reduced-radix memory form (k_fb_commit with S > 1): a pairwise tree in place --
// level n adds partial[q + n] to partial[q] for q < n on the four lanes of quad q (ec30.hip.h:xyzz30_add_quad: ~2 us per level
// against ~6 us for a one-lane addition and ~12 us for the out-of-line 8 x 32-bit one above), a barrier between levels; the
// last level leaves the row's sum in the 2^256 form k_fb_finish and the host read.  S a power of two (2 .. 128); a row takes
// S / 2 quads, a block of 256 lanes 128 / S rows.
template <class C>
__global__ void __launch_bounds__(256)
k_fb_fold_quad(XYZZ<typename C::Fp>* __restrict__ partial, uint32_t n_rows, uint32_t S) {
    using M = typename C::Fp;
    const uint32_t half = S >> 1;                        // quads per row
    const uint32_t quad = threadIdx.x >> 2;
    const uint32_t q = quad % half;
    uint32_t row = blockIdx.x * (64u / half) + quad / half;
    const bool row_ok = row < n_rows;
    if (!row_ok) row = n_rows - 1;                       // padding quads compute on a valid pair and store nothing
    XYZZ<M>* base = partial + (size_t)row * S;
    const uint32_t wave_q0 = ((threadIdx.x & ~63u) >> 2) % half;     // the wave's first quad inside its row (half >= 16), else 0
#pragma unroll 1
    for (uint32_t n = half; n >= 1; n >>= 1) {
        if (half < 16u || wave_q0 < n) {                 // wave-uniform: this wave holds at least one quad with work
            const bool live = row_ok && q < n;
            const uint32_t qq = q < n ? q : 0u;
            xyzz30_add_quad<M>(base + qq, base + qq + n, base + qq, n == 1u, live, threadIdx.x & 63u);
        }
        __syncthreads();
    }
}

// a lane takes FB_FINISH_ROWS rows: partial[row * S] -> affine with ONE inversion for the lane's rows (Montgomery's trick: the
// inversion is a chain of ~380 dependent products, a row's share of the trick three), Montgomery -> big-endian X||Y (64 zero
// bytes = infinity).  Rows of a lane are block-strided so that neighbouring lanes still read neighbouring rows.
// Up to FB_FINISH_SPREAD rows (two waves per compute unit) a lane takes ONE row, up to twice that two: the chip is idle otherwise,
// and the three products per row of the trick plus the conversions of four rows are a third of the lane's dependent chain
// (0.078 -> 0.054 ms up to 16 384 rows, 0.057 at 32 768; at 65 536 rows one row per lane loses: 0.085 against 0.080).
constexpr int FB_FINISH_ROWS_MAX = 4;
constexpr size_t FB_FINISH_SPREAD = 32768;
template <class C, int FB_FINISH_ROWS>
__global__ void __launch_bounds__(64)
k_fb_finish(const XYZZ<typename C::Fp>* __restrict__ partial, uint32_t n_rows, uint32_t S, uint8_t* __restrict__ out) {
    using M = typename C::Fp;
    const uint32_t base = blockIdx.x * (64u * FB_FINISH_ROWS) + threadIdx.x;
    Fe<M> zzz[FB_FINISH_ROWS], pre[FB_FINISH_ROWS];
    bool live[FB_FINISH_ROWS];
    Fe<M> acc = fe_one<M>();
#pragma unroll
    for (int k = 0; k < FB_FINISH_ROWS; k++) {
        const uint32_t row = base + 64u * k;
        live[k] = false;
        zzz[k] = fe_one<M>();
        if (row < n_rows) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(partial + (size_t)row * S) + 24;     // ZZZ
            Fe<M> z = load_fe<M>(src);
            if (!fe_is_zero<M>(z)) { live[k] = true; zzz[k] = z; }
        }
        pre[k] = acc;
        acc = fe_mul_call<M>(acc, zzz[k]);
    }
    Fe<M> inv;
    if constexpr (C::F30_BUCKETS) inv = fe_inv_safegcd<M>(acc);
    else inv = fe_inv_dev<M>(acc);
    Fe<M> one = fe_zero<M>();
    one.v[0] = 1;
#pragma unroll
    for (int k = FB_FINISH_ROWS - 1; k >= 0; k--) {
        const uint32_t row = base + 64u * k;
        const Fe<M> inv_k = fe_mul_call<M>(inv, pre[k]);             // 1 / ZZZ of row k
        inv = fe_mul_call<M>(inv, zzz[k]);
        if (row >= n_rows) continue;
        uint8_t* dst = out + (size_t)row * 64;
        if (!live[k]) {
            uint4 z = make_uint4(0, 0, 0, 0);
            uint4* q = reinterpret_cast<uint4*>(dst);
            q[0] = z; q[1] = z; q[2] = z; q[3] = z;
            continue;
        }
        const XYZZ<M> p = load_xyzz<M>(partial + (size_t)row * S);
        Affine<M> a = xyzz_to_affine_with_inv<M>(p, inv_k);
        Fe<M> x = fe_mul_call<M>(a.x, one), y = fe_mul_call<M>(a.y, one);  // out of Montgomery form
        store_be256(dst, x.v);
        store_be256(dst + 32, y.v);
    }
}

}  // namespace porla
