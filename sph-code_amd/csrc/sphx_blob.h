// sphx_blob.h - shared pieces of the LDS-staged sum passes (sphx_blob.hip: hydro_update's sums;
// sphx_loopforms.hip: the loop forms of the step's loop-form mode): blob geometry, the LDS image
// layout, staging, slot-list access and the lanes-per-particle reductions.
#pragma once
#include "sphx_internal.h"

#ifndef BLOB_P
#define BLOB_P 128                  // particles per workgroup
#endif
#define BLOB_T (2 * BLOB_P)          // threads of the dedup kernel (two per particle)
#define LPP SPHX_SUM_PARTS           // lanes per particle in the passes = partial sums per total
#define PASS_T (BLOB_P * LPP)       // threads per workgroup of the passes
#ifndef BLOB_S
#define BLOB_S 960                  // hash-table entries = image slots
#endif
#ifndef DEDUP_PRIO
#define DEDUP_PRIO 2                 // wave priority of the list dedup
#endif
#ifndef BLOB_STAGE_PRIO
#define BLOB_STAGE_PRIO 1            // wave priority while a workgroup stages its image (0: as every other wave)
#endif
#ifndef PASS_MINW
#define PASS_MINW 4                  // waves per SIMD the pass kernels are compiled for
#endif
#define BLOB_PROBES 96
#define SLOT_NONE 0xFFFFu           // no neighbour (list shorter than K)
#define SLOT_OVER 0xFFFEu           // neighbour not staged: read it from global memory
#define DD_BATCH 8                  // list entries fetched together per lane by the dedup kernel

typedef unsigned short u16;

__device__ __forceinline__ unsigned slot_hash(int j) {
    return (unsigned)(((u64)((unsigned)j * 2654435761u) * (u64)BLOB_S) >> 32);
}

// ---- helpers -----------------------------------------------------------------------------------
struct Q4 { double a, b, c, d; };
__device__ __forceinline__ Q4 gload4(const double* p) {
    const double2 lo = *reinterpret_cast<const double2*>(p);
    const double2 hi = *reinterpret_cast<const double2*>(p + 2);
    return Q4{lo.x, lo.y, hi.x, hi.y};
}
// chunks 2c, 2c+1 of slot s
__device__ __forceinline__ Q4 lload4(const double2* img, int s, int c2) {
    const double2 lo = img[(2 * c2) * BLOB_S + s];
    const double2 hi = img[(2 * c2 + 1) * BLOB_S + s];
    return Q4{lo.x, lo.y, hi.x, hi.y};
}
// sqrt for the distances of the neighbour loops: the library's correctly rounded sequence (v_rsq_f64
// seed, two coupled Newton steps on g ~ sqrt(x), h ~ 1/(2 sqrt(x)), residual corrections) without its
// exponent rescaling and class checks - squared distances here are 0 or sit mid-range (1e20..1e45 m^2).
// Measured: library sqrt = 18 fp64-multiply issue slots, this = 11 (a pass spends ~80 per neighbour).
// Bit-identical to sqrt() on that range (test_step_loop_variants_are_bit_identical compares against
// the gather kernels, which call sqrt()).
__device__ __forceinline__ double sqrt_mid(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d1, h, g);
    return x > 0.0 ? g : 0.0;
}

// the value held by the other lane of the pair (lane ^ 1), moved inside the VALU
__device__ __forceinline__ double pair_swap(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double pair_swap2(double v) {                                   // lane ^ 2
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x4E, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// total of the particle's LPP partial sums, in the fixed order (p0 + p1) [+ (p2 + p3)] that the gather
// kernels use as well (valid in every lane of the group)
__device__ __forceinline__ double group_total(double acc) {
    acc = acc + pair_swap(acc);
    if (LPP == 4) acc = acc + pair_swap2(acc);
    return acc;
}
__device__ __forceinline__ double group_max(double v) {
    v = fmax(v, pair_swap(v));
    if (LPP == 4) v = fmax(v, pair_swap2(v));
    return v;
}

#define NSTAGE ((BLOB_S + PASS_T - 1) / PASS_T)
#define NB (8 / LPP) // neighbours in flight per lane (one batch = 8 list positions)
#define KPAD(k) ((((k) + 7) / 8) * 8)      // slot tile rows: whole batches
#define IMG_BYTES(per_slot, k) ((size_t)BLOB_S * (per_slot) + (size_t)KPAD(k) * BLOB_P * sizeof(u16))

// Fill the workgroup's LDS: slot lists (16-B pieces; rows k..KPAD(k) read as "no neighbour") and the
// records of the occupied table entries.  NSIDE 1: one 8-B side value per slot.  NSIDE 2 (pass 3):
// g0 replaces the record's last double (cs, unused there) and g1 is the side value.  All global
// loads are issued before the first use.  (One lane per record: four lanes per record - 16 whole records per load
// instruction instead of 64 quarter records, a quarter of the distinct lines per instruction - measured SLOWER, passes
// +10 % (162 / 181 / 149 -> 180 / 200 / 163 us): eight rounds of loads per thread instead of two, 4-way conflicts on the
// image stores; round 3.)
template <int NSIDE, class Rec>
__device__ __forceinline__ void stage(double2* img, double* side, u16* tile, const Rec* __restrict__ rec,
                                      const double* __restrict__ g0, int g0_stride,
                                      const double* __restrict__ g1, int g1_stride,
                                      const int* __restrict__ uq, const u16* __restrict__ slot16, int npad,
                                      int k, int b) {
    // the few instructions of a staging go ahead of the other resident workgroup's arithmetic: its loads are out a little
    // earlier (passes -4 % at 1e6: 0.160 / 0.199 / 0.150 -> 0.153 / 0.195 / 0.145 ms; priority 3 the same; raised again while
    // a batch's LDS reads are issued: nothing more)
    __builtin_amdgcn_s_setprio(BLOB_STAGE_PRIO);
    int ju[NSTAGE];
#pragma unroll
    for (int r = 0; r < NSTAGE; ++r) {
        const int s = threadIdx.x + r * PASS_T;
        ju[r] = (s < BLOB_S) ? uq[s] : -1;
    }
    const int pieces = KPAD(k) * (BLOB_P / 8);
    for (int q = threadIdx.x; q < pieces; q += PASS_T) {
        const int kk = q / (BLOB_P / 8), c = q % (BLOB_P / 8);
        uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (kk < k) v = *reinterpret_cast<const uint4*>(slot16 + (size_t)kk * npad + (size_t)b * BLOB_P + c * 8);
        *reinterpret_cast<uint4*>(tile + kk * BLOB_P + c * 8) = v;
    }
    double2 c[NSTAGE][4];
    double e0[NSTAGE], e1[NSTAGE];
#pragma unroll
    for (int r = 0; r < NSTAGE; ++r) {
        const int j = ju[r] < 0 ? 0 : ju[r];
        const double2* g = reinterpret_cast<const double2*>(&rec[j]);       // empty entry: particle 0, not stored
        c[r][0] = g[0]; c[r][1] = g[1]; c[r][2] = g[2]; c[r][3] = g[3];
        e0[r] = (NSIDE > 0) ? g0[(size_t)j * g0_stride] : 0.0;
        e1[r] = (NSIDE > 1) ? g1[(size_t)j * g1_stride] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < NSTAGE; ++r) {
        const int s = threadIdx.x + r * PASS_T;
        if (ju[r] >= 0) {
            if (NSIDE == 2) c[r][3].y = e0[r];
            img[0 * BLOB_S + s] = c[r][0]; img[1 * BLOB_S + s] = c[r][1];
            img[2 * BLOB_S + s] = c[r][2]; img[3 * BLOB_S + s] = c[r][3];
            if (NSIDE == 1) side[s] = e0[r];
            if (NSIDE == 2) side[s] = e1[r];
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// the lane's slot numbers for batch m0 (tile rows beyond k hold SLOT_NONE)
__device__ __forceinline__ void load_slots(unsigned (&sl)[NB], const u16* tile, int m0, int half, int t) {
#pragma unroll
    for (int u = 0; u < NB; ++u) sl[u] = tile[(LPP * (m0 + u) + half) * BLOB_P + t];
}
__device__ __forceinline__ bool all_staged(const unsigned (&sl)[NB]) {
    unsigned worst = sl[0];
#pragma unroll
    for (int u = 1; u < NB; ++u) worst = worst > sl[u] ? worst : sl[u];
    return __ballot(worst >= SLOT_OVER) == 0ull;
}


// The blobs a launch works through: all nblk of them, or one of blob_split_kernel's lists (decomposed runs: the
// interior blobs while a halo phase is in flight, the boundary blobs after it) - its length read from device memory.
// list: interior blobs, then boundary blobs (each in blob order); cnt = {interior, boundary, idle}.
// mode 1: the interior ones, 2: the boundary ones, 3: both (every blob with something to compute).
struct BlobSel { const int* list; const int* cnt; int mode; };
__device__ __forceinline__ int blob_sel_count(const BlobSel& s, int nblk) {
    if (!s.list) return nblk;
    return (s.mode == 1) ? s.cnt[0] : (s.mode == 2) ? s.cnt[1] : s.cnt[0] + s.cnt[1];
}
__device__ __forceinline__ int blob_sel_at(const BlobSel& s, int bi, int count) {
    const int q = xcd_block(bi, count);
    return s.list ? s.list[q + (s.mode == 2 ? s.cnt[0] : 0)] : q;
}
// part: 0 all blobs (with a valid split: all but the idle ones), 1 the interior ones, 2 the boundary ones
BlobSel sphx_blob_sel(sphx_ctx* ctx, int part);

int sphx_blob_grid(sphx_ctx* ctx, int nblk);       // persistent grid of the LDS passes (2 workgroups per CU)
