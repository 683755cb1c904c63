// sphx_wave.h - wave64 building blocks shared by the search kernels: lane exchanges without the
// LDS crossbar where the ISA allows it, bitonic networks on (u64 key, u32 index) pairs and on
// unique 32-bit keys, SGPR broadcasts, SciPy-compatible squared distance.
#pragma once
#include "sphx_internal.h"

#define KNN_INF 0x7FF0000000000000ull

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bool kv_less(u64 ka, u32 va, u64 kb, u32 vb) {
    return ka < kb || (ka == kb && va < vb);
}

// ---- lane exchange lane ^ J without the LDS crossbar where the ISA allows it -----------------
// DPP quad_perm / row_ror / row_half_mirror move data inside a row of 16 lanes in the VALU;
// xor 16 uses ds_swizzle (no address VGPR), xor 32 ds_bpermute (v_permlane32_swap + select
// measured no faster).
template <int J> __device__ __forceinline__ u32 xchg32(u32 v) {
    if constexpr (J == 1) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);        // row_half_mirror: ^7
        return (u32)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, true);          // quad_perm [3,2,1,0]: ^3
    } else if constexpr (J == 8) {
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);    // row_ror:8
    } else if constexpr (J == 16) {
        return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                      // bitmode xor 0x10
    } else {
        return (u32)__shfl_xor((int)v, 32, 64);
    }
}

// Lane patterns of the networks are compile-time constants.  Expressed as lane predicates the
// compiler hoists 30+ of them into SGPR pairs and then spills them (2 v_readlane per use);
// expressed as 64-bit literals they are rematerialised by s_mov where needed.
template <int SIZE, int J, bool DESC> constexpr u64 sort_keepmin_mask() {
    u64 m = 0;
    for (int l = 0; l < 64; ++l) {
        const bool up = (((l & SIZE) == 0) != DESC);
        const bool lower = (l & J) == 0;
        if (lower == up) m |= (1ull << l);
    }
    return m;
}
template <int J> constexpr u64 merge_keepmin_mask() {
    u64 m = 0;
    for (int l = 0; l < 64; ++l)
        if ((l & J) == 0) m |= (1ull << l);
    return m;
}
// dst = MASK[lane] ? a : b with a compile-time mask: the literal is moved into VCC right at the
// select (2 SALU), so no SGPR pair stays live for it
template <u64 MASK> __device__ __forceinline__ u32 select_const(u32 a, u32 b) {
    u32 d;
    asm("s_mov_b32 vcc_lo, %3\n\ts_mov_b32 vcc_hi, %4\n\tv_cndmask_b32_e32 %0, %1, %2, vcc"
        : "=v"(d)
        : "v"(b), "v"(a), "i"((int)(u32)(MASK & 0xFFFFFFFFull)), "i"((int)(u32)(MASK >> 32))
        : "vcc");
    return d;
}
// dst = mask[lane] ? a : b   with the mask in an SGPR pair
__device__ __forceinline__ u32 select_mask(u64 mask, u32 a, u32 b) {
    u32 d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(mask));
    return d;
}

// compare-exchange with lane ^ J; keep_min: this lane keeps the smaller of the pair.
// All (key,id) pairs are distinct except the (INF, ~0) padding, for which either choice is
// the same value, so one lexicographic compare decides both directions.
template <int J> __device__ __forceinline__ void cmpx(u64& k, u32& v, bool keep_min) {
    const u32 plo = xchg32<J>((u32)k), phi = xchg32<J>((u32)(k >> 32));
    const u64 pk = ((u64)phi << 32) | plo;
    const u32 pv = xchg32<J>(v);
    const bool p_lt = kv_less(pk, pv, k, v);
    const bool take = (p_lt == keep_min);
    k = take ? pk : k;
    v = take ? pv : v;
}
// the same with the keep-min lanes given as a constant mask
template <int J> __device__ __forceinline__ void cmpx_m(u64& k, u32& v, u64 keepmin_mask) {
    const u32 klo = (u32)k, khi = (u32)(k >> 32);
    const u32 plo = xchg32<J>(klo), phi = xchg32<J>(khi);
    const u64 pk = ((u64)phi << 32) | plo;
    const u32 pv = xchg32<J>(v);
    const u64 lt = __builtin_amdgcn_ballot_w64(kv_less(pk, pv, k, v));
    const u64 take = ~(lt ^ keepmin_mask);                  // take the partner's where p_lt == keep_min
    k = ((u64)select_mask(take, phi, khi) << 32) | select_mask(take, plo, klo);
    v = select_mask(take, pv, v);
}

template <int J> __device__ __forceinline__ void merge_stage(u64& k, u32& v, int lane) {
    cmpx_m<J>(k, v, merge_keepmin_mask<J>());
    if constexpr (J > 1) merge_stage<J / 2>(k, v, lane);
}
// ---- 32-bit network: the common case -------------------------------------------------------
// Staged survivors sit in LDS; their order is found on a UNIQUE 32-bit key
//     (monotone 26-bit quantisation of d2 / R^2) << 6 | staging slot
// so one exchange + v_min/v_max/v_cndmask does a compare-exchange (vs 3 exchanges + 3 compares
// + 3 selects on (u64,u32)).  The full (d2 bits, index) pairs are then fetched from LDS by
// slot.  Two survivors in the same quantisation bin (about 1 particle in 40 000) are put into
// their exact order by a few odd-even steps with the full compare.
template <int J, u64 KEEPMIN> __device__ __forceinline__ void cmpx32(u32& k) {
    const u32 p = xchg32<J>(k);
    const u32 mn = k < p ? k : p, mx = k < p ? p : k;
    k = select_const<KEEPMIN>(mn, mx);
}
template <int SIZE, int J> __device__ __forceinline__ void sort32_stage(u32& k, int lane) {
    cmpx32<J, sort_keepmin_mask<SIZE, J, false>()>(k);
    if constexpr (J > 1) sort32_stage<SIZE, J / 2>(k, lane);
}
template <int SIZE> __device__ __forceinline__ void sort32_sizes(u32& k, int lane) {
    if constexpr (SIZE > 2) sort32_sizes<SIZE / 2>(k, lane);
    sort32_stage<SIZE, SIZE / 2>(k, lane);
}


// number of set bits of a wave mask below this lane (v_mbcnt pair: no lane-mask registers)
__device__ __forceinline__ int lanes_below(u64 m) {
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// inclusive wave64 prefix sum in six DPP adds: row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (out-of-row sources read 0)
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// broadcast lane `src` (wave-uniform) of a double / int to the whole wave through SGPRs
__device__ __forceinline__ double bcast_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)b, src);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)b >> 32), src);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

__device__ __forceinline__ double dist2_nofma(double dx, double dy, double dz) {
#pragma clang fp contract(off)
    double s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

