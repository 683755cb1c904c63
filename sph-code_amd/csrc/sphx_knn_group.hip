// sphx_knn_group.hip - the hinted search of the step loop: ONE LANE PER QUERY, one wave per group of 64
// consecutive queries of the blob (Hilbert) order, candidates staged once per group in LDS
// (replaces nsc:541-552 inside the step; the wave-per-query kernel of sphx_knn.hip stays the general path:
// un-hinted searches, the array API, and every query this kernel cannot certify).
//
// Why: sphx_knn.hip spends ~520 VALU wave-instructions per query, most of them on keeping ONE query's
// candidates sorted across the 64 lanes (a 64-key network per query, row set-up with half-empty lanes).
// Here a wave-instruction serves 64 queries:
//   1. the group's 64 queries (positions, hinted radii R_i = rscale * previous h_i) give a box; every cell
//      row (cy,cz) that can hold a particle within max R_i of the box contributes the chord of cells that
//      can (box + sphere, the Minkowski sum) - contiguous particle ranges of the cell-sorted arrays;
//   2. those ~1000 candidates are staged ONCE into an LDS tile as 16-B records {x,y,z relative to the
//      group's centre in cell units as fp32, particle index}: coalesced loads, full lanes;
//   3. phase A: every query against the whole tile - on the MATRIX CORES where the error bound allows (d^2 - R^2 as a
//      16-deep inner product of fp16 hi/lo splits, v_mfma_f32_32x32x16_f16: 32 candidates x 32 queries per instruction,
//      the signs shifted into per-lane mask words), else in packed fp32 with broadcast LDS reads;
//   4. phase B: each lane turns its ~50 set bits into 32-bit keys  (21-bit quantised d^2/R^2) << 11 | tile
//      slot  held in 64 REGISTERS, and orders them with Batcher's odd-even merge network on registers
//      (543 compare-exchanges = 1086 v_min/v_max for 64 queries at once: 17 instructions per query);
//   5. the first K keys name the neighbours; h_i = sqrt of the EXACT fp64 distance to the K-th.
//
// Exactness.  fp32 distances from coordinates relative to the group's centre carry an error of at most
//   err(d^2) <= eps R^2 (6.93 E/R + 7.5),  eps = 2^-24, E = largest |relative coordinate| in the tile
// (derivation in DESIGN.md 5.2b).  A lane accepts its order only if every two consecutive keys among the
// first K+1 are further apart than twice that bound (doubled again for safety) - then the fp32 order IS the
// order of the exact fp64 distances, the K-th is certain, and the result equals the wave-per-query kernel's
// bit for bit (same indices, same order, h from the same fp64 expression).  Anything else - a near tie
// (~0.7 % of the queries), more than 64 candidates inside the radius, fewer than K, a tile that does not fit,
// no usable hint - puts the query on a list that knn_kernel then searches exactly (list mode).
#include "sphx_internal.h"
#include "sphx_wave.h"
#include "sphx_knn_group.h"


typedef float f32x2 __attribute__((ext_vector_type(2)));

// The tile holds the candidates as four arrays by slot (x, y, z, index): the matrix-core form of phase A, the keys and
// the output read single slots; the packed-fp32 form of phase A reads the pairs (2 w, 2 w + 1) as three 8-byte loads.

// wave-wide min / max of doubles: lane exchanges by DPP inside the rows of 16 (sphx_wave.h xchg32), one ds_swizzle
// and one bpermute for the last two steps
template <int J> __device__ __forceinline__ double xchg64(double v) {
    const u64 b = (u64)__double_as_longlong(v);
    const u32 lo = xchg32<J>((u32)b), hi = xchg32<J>((u32)(b >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ double wmin(double v) {
    v = fmin(v, xchg64<1>(v)); v = fmin(v, xchg64<2>(v)); v = fmin(v, xchg64<4>(v));
    v = fmin(v, xchg64<8>(v)); v = fmin(v, xchg64<16>(v)); v = fmin(v, xchg64<32>(v));
    return v;
}
__device__ __forceinline__ double wmax(double v) {
    v = fmax(v, xchg64<1>(v)); v = fmax(v, xchg64<2>(v)); v = fmax(v, xchg64<4>(v));
    v = fmax(v, xchg64<8>(v)); v = fmax(v, xchg64<16>(v)); v = fmax(v, xchg64<32>(v));
    return v;
}
__device__ __forceinline__ int cell_of_coord(double v, double vmin, double inv_cell, int nmax1) {
    double t = (v - vmin) * inv_cell;
    t = fmin(fmax(t, 0.0), (double)nmax1);
    return (int)t;
}

// m = (m << 1) | sign(s): one v_alignbit (the funnel shift (m:s) >> 31)
__device__ __forceinline__ u32 shift_in_sign(u32 m, float s) {
    return __builtin_amdgcn_alignbit(m, __float_as_uint(s), 31);
}

// Batcher's odd-even merge sort on 64 registers (ascending), every index a compile-time constant
#define KG_CE(a, b) { const u32 lo_ = min(key[a], key[b]); const u32 hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
template <int P, int Kk, int J, int I> struct OemInner {
    static __device__ __forceinline__ void run(u32 (&key)[64]) {
        if constexpr (I < Kk && J + I + Kk < 64) {
            if constexpr ((I + J) / (2 * P) == (I + J + Kk) / (2 * P)) KG_CE(I + J, I + J + Kk);
            OemInner<P, Kk, J, I + 1>::run(key);
        }
    }
};
template <int P, int Kk, int J> struct OemJ {
    static __device__ __forceinline__ void run(u32 (&key)[64]) {
        if constexpr (J + Kk < 64) {
            OemInner<P, Kk, J, 0>::run(key);
            OemJ<P, Kk, J + 2 * Kk>::run(key);
        }
    }
};
template <int P, int Kk> struct OemK {
    static __device__ __forceinline__ void run(u32 (&key)[64]) {
        if constexpr (Kk >= 1) {
            OemJ<P, Kk, Kk % P>::run(key);
            OemK<P, Kk / 2>::run(key);
        }
    }
};
template <int P> struct OemP {
    static __device__ __forceinline__ void run(u32 (&key)[64]) {
        if constexpr (P < 64) {
            OemK<P, P>::run(key);
            OemP<P * 2>::run(key);
        }
    }
};

// Workgroup = 4 waves = ONE group of 64 queries: in every wave lane l is query l.  The waves share the
// set-up of the tile (row chords: one row per thread; staging: one slot per thread and pass) and split phase A
// by mask word (wave w takes words w, w + 4, ...), appending to the queries' shared slot lists through an LDS
// counter per query.  Wave 0 alone then builds the keys, orders them in registers, certifies and writes out;
// waves 1-3 have left by then - their SIMD slots go to the phase A of other groups (LDS per group ~38 KB:
// four groups = 16 waves per CU instead of the four a wave-per-group kernel could hold).
struct KgShared {
    double Ox, Oy, Oz, E;           // tile origin, largest |relative coordinate| (cell units)
    double cx0, cx1, cy0, cy1, cz0, cz1, R2, Rcov;   // box clamped into the grid, covered radius
    float hx, hy, hz, rc2;          // the box's half extents and (covered radius)^2, cell units, rounded up
    int ry0, rz0, ysp, nrows, T, ok;
    int wtot[4], wne[4];
};

#ifndef KG_MINWAVES
#define KG_MINWAVES 5
#endif
#ifndef KG_FRONT_PRIO
#define KG_FRONT_PRIO 1             // wave priority until phase A begins (0: as every other wave)
#endif
#ifndef KG_STAGE_U
#define KG_STAGE_U 2                 // candidates per thread whose loads are in flight together while the tile is staged
#endif
#ifndef KG_MFMA
#define KG_MFMA 1                    // phase A on the matrix cores where its error bound allows
#endif
#define KG_MFMA_PAD 2.0e-3           // acceptance pad of the matrix-core form: D = d^2 - R^2 (1 + pad) < 0
#define KG_MFMA_KAPPA 6.0e-5         // its error bound: kappa E^2 (cell units)
#define KG_MFMA_ERRMAX 1.75e-3       // ... which must not exceed this share of the smallest R^2 of the group
#define KG_MFMA_EMAX 40.0            // and the tile's half extent this many cells (fp16 range: |c|^2 <= 4800, pads at 75)

// ---- operands of the matrix-core form of phase A (see the kernel; also driven by sphx_selftest_mfma_cull) ----
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ u32 kg_pk(float lo, float hi) { return __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(lo, hi)); }
__device__ __forceinline__ float kg_flo(u32 d) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(d & 0xFFFFu)); }
__device__ __forceinline__ float kg_fhi(u32 d) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(d >> 16)); }
// row of candidate (cx, cy, cz): k 0..7 in the lanes of half 0, k 8..15 in those of half 1
__device__ __forceinline__ f16x8 kg_cand_frag(float cx, float cy, float cz, int hh) {
    const u32 h01 = kg_pk(cx, cy), h2_ = kg_pk(cz, 0.0f);
    u32 w2, w3;
    if (hh == 0) {
        w2 = kg_pk(cx - kg_flo(h01), cy - kg_fhi(h01)); w3 = kg_pk(cz - kg_flo(h2_), 0.0f);
    } else {
        const float C2 = fmaf(cz, cz, fmaf(cy, cy, cx * cx));
        const u32 ch = kg_pk(C2, 0.0f);
        w2 = kg_pk(C2, C2 - kg_flo(ch)); w3 = 0x3C003C00u;                     // (1, 1)
    }
    return __builtin_bit_cast(f16x8, make_uint4(h01, h2_, w2, w3));
}
// column of query (sx, sy, sz) with padded squared radius sr (sr <= 0: no usable query - the column never accepts)
__device__ __forceinline__ f16x8 kg_query_frag(float sx, float sy, float sz, float sr, int hh) {
    const u32 h01 = kg_pk(sx, sy), h2_ = kg_pk(sz, 0.0f);
    u32 w0, w1, w2, w3;
    if (hh == 0) {
        w0 = kg_pk(-2.0f * kg_flo(h01), -2.0f * kg_fhi(h01)); w1 = kg_pk(-2.0f * kg_flo(h2_), 0.0f);
        w2 = w0; w3 = w1;
    } else {
        const float lx = sx - kg_flo(h01), ly = sy - kg_fhi(h01), lz = sz - kg_flo(h2_);
        const u32 l01 = kg_pk(lx, ly), l2_ = kg_pk(lz, 0.0f);
        w0 = kg_pk(-2.0f * kg_flo(l01), -2.0f * kg_fhi(l01)); w1 = kg_pk(-2.0f * kg_flo(l2_), 0.0f);
        w2 = 0x3C003C00u;
        float Q = fmaf(sz, sz, fmaf(sy, sy, sx * sx)) - sr;
        if (!(sr > 0.0f)) Q = 60000.0f;
        const u32 qh = kg_pk(Q, 0.0f);
        w3 = kg_pk(Q, Q - kg_flo(qh));
    }
    return __builtin_bit_cast(f16x8, make_uint4(w0, w1, w2, w3));
}
__global__ __launch_bounds__(256, KG_MINWAVES) void knn_group_kernel(KnnGroupArgs a) {
    // the tile: four arrays by slot (x, y, z relative to the group's centre in cell units, storage index).  (Until the matrix
    // cores took phase A the candidates lay in pairs {x0,x1,y0,y1}{z0,z1,i0,i1} for packed math: every other access paid
    // three instructions for the offset.)
    __shared__ __attribute__((aligned(16))) float tile_soa[4 * KG_TCAP];
    float* tx = tile_soa;
    float* ty = tile_soa + KG_TCAP;
    float* tz = tile_soa + 2 * KG_TCAP;
    int* ti = reinterpret_cast<int*>(tile_soa + 3 * KG_TCAP);
    __shared__ unsigned short slist[64 * 64];       // [slot][query]: tile slots inside the query's radius;
                                                    // while staging: row of each slot (u8) + the rows' bases
    static_assert(KG_TPRE * 2 + KG_MAXROWS * 4 <= 64 * 64 * 2, "slot->row ids + row bases share the list's memory");
        unsigned short* rowof = slist;                                               // KG_TPRE row ids
    int* rowbase = reinterpret_cast<int*>(slist) + KG_TPRE / 2;                  // KG_MAXROWS ints
    __shared__ int cntq[64];
    __shared__ KgShared sh;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // in-kernel section clock (diagnostic launches only: a.prof == nullptr in the product)
    u64 tprev = a.prof ? __builtin_readcyclecounter() : 0;
#define KG_STAMP(sec) if (a.prof) { const u64 tn_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&a.prof[sec], tn_ - tprev); tprev = tn_; }
    // the latency-bound front of a group (queries, rows, staging: a handful of dependent round trips and few instructions)
    // goes ahead of the other resident groups' arithmetic (search 0.459 -> 0.443 ms; priority 3 the same)
    __builtin_amdgcn_s_setprio(KG_FRONT_PRIO);
    const int p = xcd_block(blockIdx.x, gridDim.x) * 64 + lane;       // processing slot = list column
    const GridParams g = a.g;
    const int K = a.k;

    // ---- the group's queries (every wave holds all 64) -------------------------------------------
    double qx = 0.0, qy = 0.0, qz = 0.0, R = 0.0;
    int qs = 0, qid = 0x7FFFFFFF;
    const bool inrange = p < a.n;
    if (inrange) {
        qs = a.qorder ? a.qorder[p] : p;
        qx = a.x[qs]; qy = a.y[qs]; qz = a.z[qs];
        qid = a.id[qs];
        R = a.rsearch[a.hint_by_id ? qid : qs] * a.rscale;
    }
    const bool is_query = inrange && qid < a.n_active;                // ghosts are candidates, never queries
    // a usable hint: finite, positive, not beyond the caller's bound (bounded searches take the general kernel)
    bool ok = is_query && R > 0.0 && R < 1e300 && R < a.rbound && isfinite(qx) && isfinite(qy) && isfinite(qz);
    bool fail = is_query && !ok;
    // one wide radius among 64 (a rim particle among surface particles) would size the tile for all of them and push the
    // whole group over its caps: a query KG_RSPREAD times above the group's smallest radius goes to the general kernel alone
    const double Rmin = wmin(ok ? R : (double)INFINITY);
    if (ok && R > KG_RSPREAD * Rmin) { ok = false; fail = true; }
    int why = fail ? 1 : 0;                     // diagnostics: 1 no hint, 2 tile, 3 tolerance, 4 > 64 inside, 5 < K inside, 6 near tie
    const u64 okmask = __builtin_amdgcn_ballot_w64(ok);
    if (wave == 0) {
        cntq[lane] = 0;
        const double INF = INFINITY;
        const double bx0 = wmin(ok ? qx : INF), bx1 = wmax(ok ? qx : -INF);
        const double by0 = wmin(ok ? qy : INF), by1 = wmax(ok ? qy : -INF);
        const double bz0 = wmin(ok ? qz : INF), bz1 = wmax(ok ? qz : -INF);
        const double Rmax = wmax(ok ? R : 0.0);
        const double Rcov = Rmax * 1.0002;           // an accepted candidate lies within R_i (1 + 1.6e-4), see below
        // rows are tested against the box clamped into the grid: the boundary cells are half-infinite (out-of-box
        // particles are clamped into them, sphx_grid.hip), their open side can only lie behind the clamped box
        const double gx1 = g.xmin + g.nx * g.cell, gy1 = g.ymin + g.ny * g.cell, gz1 = g.zmin + g.nz * g.cell;
        const double cy0 = fmin(fmax(by0, g.ymin), gy1), cy1 = fmin(fmax(by1, g.ymin), gy1);
        const double cz0 = fmin(fmax(bz0, g.zmin), gz1), cz1 = fmin(fmax(bz1, g.zmin), gz1);
        const int ry0 = cell_of_coord(cy0 - Rcov, g.ymin, g.inv_cell, g.ny - 1);
        const int ry1 = cell_of_coord(cy1 + Rcov, g.ymin, g.inv_cell, g.ny - 1);
        const int rz0 = cell_of_coord(cz0 - Rcov, g.zmin, g.inv_cell, g.nz - 1);
        const int rz1 = cell_of_coord(cz1 + Rcov, g.zmin, g.inv_cell, g.nz - 1);
        if (lane == 0) {
            sh.Ox = 0.5 * (bx0 + bx1); sh.Oy = 0.5 * (by0 + by1); sh.Oz = 0.5 * (bz0 + bz1);
            sh.E = (fmax(fmax(bx1 - bx0, by1 - by0), bz1 - bz0) * 0.5 + Rcov) * g.inv_cell * 1.0001;
            sh.hx = (float)((bx1 - bx0) * 0.5 * g.inv_cell) * 1.00001f + 1e-6f;
            sh.hy = (float)((by1 - by0) * 0.5 * g.inv_cell) * 1.00001f + 1e-6f;
            sh.hz = (float)((bz1 - bz0) * 0.5 * g.inv_cell) * 1.00001f + 1e-6f;
            sh.rc2 = (float)((Rcov * g.inv_cell) * (Rcov * g.inv_cell)) * 1.0002f;
            sh.cx0 = fmin(fmax(bx0, g.xmin), gx1); sh.cx1 = fmin(fmax(bx1, g.xmin), gx1);
            sh.cy0 = cy0; sh.cy1 = cy1; sh.cz0 = cz0; sh.cz1 = cz1;
            sh.R2 = Rcov * Rcov; sh.Rcov = Rcov;
            sh.ry0 = ry0; sh.rz0 = rz0; sh.ysp = ry1 - ry0 + 1;
            sh.nrows = (ry1 - ry0 + 1) * (rz1 - rz0 + 1);
            sh.T = 0;
            sh.ok = (okmask != 0ull && sh.nrows <= KG_ROWS_ALL) ? 1 : 0;
            if (a.counters && okmask != 0ull && sh.nrows > KG_ROWS_ALL) atomicAdd(&a.counters[SC_KGDBG + 7], 1ull);   // diagnostics: groups over the row cap
        }
    }
    __syncthreads();
    KG_STAMP(0)
    bool group_ok = sh.ok != 0;
    const double Ox = sh.Ox, Oy = sh.Oy, Oz = sh.Oz;
    // Phase A on the matrix cores (below) when its error bound KG_MFMA_KAPPA E^2 fits the acceptance pad of every query
    // of the group (R_i >= Rmin) and the tile's numbers fit fp16's range; else the packed-fp32 form
    const double Rcmin = Rmin * g.inv_cell;
    const bool use_mfma = KG_MFMA && sh.E <= KG_MFMA_EMAX && KG_MFMA_KAPPA * sh.E * sh.E <= KG_MFMA_ERRMAX * Rcmin * Rcmin;
    int T = 0;
    if (group_ok) {
        // ---- the rows of cells, 512 at a time (two per thread): the chord of cells that can hold a particle within Rcov
        // of the box.  Only NON-EMPTY rows are kept (compact ids: the diffuse rim's groups span thousands of rows of
        // which a few dozen hold anything); a slot remembers its row's id, a row its first particle. ----
        const double INF = INFINITY;
        int nne = 0;                                                     // non-empty rows so far (uniform)
        for (int r0 = 0; r0 < sh.nrows && group_ok; r0 += 512) {
            int s_row[2] = {0, 0}, cnt[2] = {0, 0};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + 2 * tid + u;
                if (r < sh.nrows) {
                    const int ysp = sh.ysp;
                    const int rz = r / ysp, ry = r - rz * ysp;
                    const int cy = sh.ry0 + ry, cz = sh.rz0 + rz;
                    // distance from the (clamped) box to the row's (y,z) column of cells; boundary rows are open
                    const double ylo = (cy == 0) ? -INF : g.ymin + cy * g.cell, yhi = (cy == g.ny - 1) ? INF : g.ymin + (cy + 1) * g.cell;
                    const double zlo = (cz == 0) ? -INF : g.zmin + cz * g.cell, zhi = (cz == g.nz - 1) ? INF : g.zmin + (cz + 1) * g.cell;
                    const double dy = fmax(fmax(ylo - sh.cy1, sh.cy0 - yhi), 0.0) * 0.999999;      // (never over-estimated)
                    const double dz = fmax(fmax(zlo - sh.cz1, sh.cz0 - zhi), 0.0) * 0.999999;
                    const double rem = sh.R2 - (dy * dy + dz * dz);
                    if (rem >= 0.0) {
                        const double xr = sqrt(rem) * 1.000001;
                        const int x0 = cell_of_coord(sh.cx0 - xr, g.xmin, g.inv_cell, g.nx - 1);
                        const int x1 = cell_of_coord(sh.cx1 + xr, g.xmin, g.inv_cell, g.nx - 1);
                        const int row = (cz * g.ny + cy) * g.nx;
                        s_row[u] = a.cell_start[row + x0];
                        cnt[u] = a.cell_start[row + x1 + 1] - s_row[u];
                    }
                }
            }
            const int mycnt = cnt[0] + cnt[1], myne = (cnt[0] > 0 ? 1 : 0) + (cnt[1] > 0 ? 1 : 0);
            const int incl = wave_scan_incl(mycnt), incn = wave_scan_incl(myne);
            if (lane == 63) { sh.wtot[wave] = incl; sh.wne[wave] = incn; }
            __syncthreads();
            int woff = 0, wn = 0, totT = 0, totN = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int tw = sh.wtot[w], nw_ = sh.wne[w];
                if (w < wave) { woff += tw; wn += nw_; }
                totT += tw; totN += nw_;
            }
            if (T + totT > KG_TPRE || nne + totN > KG_MAXROWS) {          // (uniform over the workgroup)
                group_ok = false;
                if (a.counters && tid == 0) atomicAdd(&a.counters[SC_KGDBG + (T + totT > KG_TPRE ? 0 : 7)], 1ull);   // diagnostics
            } else {
                int off = T + woff + incl - mycnt;                         // the first slot of the thread's rows in the tile
                int id = nne + wn + incn - myne;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (cnt[u] > 0) {
                        rowbase[id] = s_row[u] - off;                      // slot t of row `id` is particle rowbase[id] + t
                        for (int q = 0; q < cnt[u]; ++q) rowof[off + q] = (unsigned short)id;
                        off += cnt[u];
                        ++id;
                    }
                }
            }
            T += totT; nne += totN;
            __syncthreads();
        }
        __syncthreads();
        KG_STAMP(1)
        // ---- stage: one candidate per thread and pass (two passes' loads in flight).  The rows hold whole cells; a
        // candidate is kept only if it lies within Rcov of the group's box (the exact box + sphere region, not its cover
        // by cells: a quarter fewer slots for phase A), the kept ones compacted per wave (ballot + one LDS add). ----
        if (group_ok) {
            const float hx = sh.hx, hy = sh.hy, hz = sh.hz, rc2 = sh.rc2;
            for (int t0 = 0; t0 < T; t0 += 256 * KG_STAGE_U) {
                int pp[KG_STAGE_U];
#pragma unroll
                for (int u = 0; u < KG_STAGE_U; ++u) {
                    const int t = t0 + 256 * u + tid;
                    pp[u] = t < T ? rowbase[rowof[t]] + t : -1;
                }
                double X[KG_STAGE_U], Y[KG_STAGE_U], Z[KG_STAGE_U];
#pragma unroll
                for (int u = 0; u < KG_STAGE_U; ++u) {
                    const int q = pp[u] >= 0 ? pp[u] : 0;
                    X[u] = a.x[q]; Y[u] = a.y[q]; Z[u] = a.z[q];
                }
#pragma unroll
                for (int u = 0; u < KG_STAGE_U; ++u) {
                    if (256 * u >= T - t0) break;                 // (uniform: nothing left for this and the later slots)
                    const float rx = (float)((X[u] - Ox) * g.inv_cell), ry = (float)((Y[u] - Oy) * g.inv_cell),
                                rz = (float)((Z[u] - Oz) * g.inv_cell);
                    const float ex = fmaxf(fabsf(rx) - hx, 0.0f), ey = fmaxf(fabsf(ry) - hy, 0.0f), ez = fmaxf(fabsf(rz) - hz, 0.0f);
                    const bool keep = pp[u] >= 0 && fmaf(ez, ez, fmaf(ey, ey, ex * ex)) <= rc2;
                    const u64 km = __builtin_amdgcn_ballot_w64(keep);
                    int base = 0;
                    if (km) {
                        if (lane == 0) base = atomicAdd(&sh.T, __popcll(km));
                        base = __builtin_amdgcn_readfirstlane(base);
                    }
                    const int slot = base + lanes_below(km);
                    if (keep && slot < KG_TCAP) {
                        tx[slot] = rx; ty[slot] = ry; tz[slot] = rz;
                        ti[slot] = pp[u];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (group_ok) {
        T = sh.T;                                                         // candidates kept
        if (T > KG_TCAP) group_ok = false;
        else {
            // pad the tile to whole mask words with records nobody accepts
            const int nwp = (T + 31) >> 5;
            if (tid < nwp * 32 - T) {
                const int o = T + tid;
                // (matrix-core form: finite in fp16, |c|^2 = 16875 < 65504, and >= 35 cells from every query on each axis)
                const float padv = use_mfma ? 75.0f : 1e30f;
                tx[o] = padv; ty[o] = padv; tz[o] = padv;
                ti[o] = -1;
            }
        }
    }
    __syncthreads();                     // the tile is complete; rowof / rowbase are dead (the lists take their memory)
    KG_STAMP(2)
    if (!group_ok) {
        if (is_query && !fail) why = 2;
        fail = is_query;
        ok = false;
    }
    const int nw = group_ok ? (T + 31) >> 5 : 0;
    __builtin_amdgcn_s_setprio(0);

    // ---- phase A: every query against the tile, the waves splitting it by mask word -----------------
    // accept d2 <= Rc^2 (1 + 2e-4): with err(d2) <= 1e-4 Rc^2 (checked below) every candidate truly inside R_i is
    // accepted, and every accepted one lies within R_i (1 + 1.6e-4) < the radius the tile covers
    const double Rc = R * g.inv_cell;
    const float fqx = (float)((qx - Ox) * g.inv_cell), fqy = (float)((qy - Oy) * g.inv_cell), fqz = (float)((qz - Oz) * g.inv_cell);
    // err(d2)/Rc^2 <= eps (6.93 E/Rc + 5); 7.5 taken
    const double tol_rel = ok ? 5.9604644775390625e-08 * (6.93 * sh.E / Rc + 7.5) : 0.0;
    if (ok && !(tol_rel <= 0.5e-4)) { ok = false; fail = true; why = 3; }
    const float r2f = ok ? (float)(Rc * Rc * 1.0002) : -1.0f;
    if (use_mfma) {
        // ---- phase A on the matrix cores.  d^2 - R^2 = |c|^2 - 2 c.q + |q|^2 - R^2 is a 16-deep inner product of a row
        // made from the candidate and a column made from the query, every fp32 number carried as an fp16 pair hi + lo
        // (hi = RTZ16(x), lo = RTZ16(x - hi): 20 bits):
        //   k:          0..2       3    4..6      7   |  8..10      11   12    13   14   15
        //   candidate:  c_hi       0    c_lo      0   |  c_hi       0    C2hi  C2lo 1    1        (C2 = |c|^2)
        //   query:     -2 q_hi     0   -2 q_hi    0   | -2 q_lo     0    1     1    Qhi  Qlo      (Q = |q|^2 - R^2 (1 + pad))
        // One v_mfma_f32_32x32x16_f16 = 32 candidates x 32 queries: lane l holds the column of query l & 31 and the rows
        // (candidates)  4 (l >> 5) + 8 (r >> 2) + (r & 3)  in its 16 accumulator registers r, so 16 funnel shifts per
        // lane turn a block's signs into mask bits - 6.4 cycles per candidate and 64 queries where the packed-fp32 form
        // (3 v_pk_add + 3 v_pk_fma per pair + 1 v_alignbit per candidate, all 4-cycle issues) takes 16.8.
        // Error: |D - (d^2 - R^2 (1 + pad))| <= KG_MFMA_KAPPA E^2 for coordinates |.| <= E - dropped c_lo q_lo
        // and the two splits' remainders 18 x 2^-20 E^2, the splits of C2 and Q 7 x 2^-20 E^2, thirteen fp32 additions of
        // terms summing to <= 13 E^2 in absolute value at 2^-23 each 2.0e-5 E^2: 4.4e-5, 6e-5 taken.  With pad = 2e-3 and
        // kappa E^2 <= 1.75e-3 Rc^2: every candidate within R_i (as fp32 sees it: + tol_rel <= 0.5e-4) is listed, and an
        // unlisted one lies beyond R_i^2 x 1.0002 - what the certification below assumes of the packed-fp32 form as well.
        const int n32 = lane & 31, hh = lane >> 5;
        f16x8 Bq[2];
#pragma unroll
        for (int H = 0; H < 2; ++H) {
            const int srcl = 32 * H + n32;
            const float sx = __shfl(fqx, srcl, 64), sy = __shfl(fqy, srcl, 64), sz = __shfl(fqz, srcl, 64);
            const float sr = __shfl(ok ? (float)(Rc * Rc * (1.0 + KG_MFMA_PAD)) : -1.0f, srcl, 64);
            Bq[H] = kg_query_frag(sx, sy, sz, sr, hh);
        }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // MFMA row `n32` carries the candidate of tile slot 32 w + pi(n32), pi chosen so that - after the two halves of the
        // wave have swapped what they hold of each other's query (v_permlane32_swap: lane l then owns query l, like
        // everywhere else in this kernel) - bit j of a lane's word (from the top) is tile slot 32 w + j:
        //   rows 8 a + b (b < 4, accumulator registers of lanes 0..31)  -> slots 4 a + b,  rows 8 a + 4 + b -> 16 + 4 a + b
        const int pi32 = ((n32 & 4) << 2) | ((n32 >> 3) << 2) | (n32 & 3);
        for (int w = wave; w < nw; w += 4) {
            const int o = w * 32 + pi32;
            const float cx = tx[o], cy = ty[o], cz = tz[o];
            const f16x8 Ac = kg_cand_frag(cx, cy, cz, hh);
            const f32x16 D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac, Bq[0], zero16, 0, 0, 0);
            const f32x16 D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ac, Bq[1], zero16, 0, 0, 0);
            u32 m0 = 0, m1 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) m0 = shift_in_sign(m0, D0[r]);       // query n32:      register r -> bit 15 - r
#pragma unroll
            for (int r = 0; r < 16; ++r) m1 = shift_in_sign(m1, D1[r]);       // query 32 + n32
            // lanes 0..31 keep m0 and get the upper lanes' m0 (the other 16 rows of query n32); lanes 32..63 get the lower
            // lanes' m1 and keep their own
            const auto sw = __builtin_amdgcn_permlane32_swap(m0, m1, false, false);
            u32 m = (sw[0] << 16) | sw[1];
            if (__builtin_amdgcn_ballot_w64(m != 0u)) {
                int slot = 0;
                if (m != 0u) slot = atomicAdd(&cntq[lane], __popc(m));
                while (__builtin_amdgcn_ballot_w64(m != 0u)) {
                    if (m != 0u) {
                        const int j = __clz((int)m);
                        m &= ~(0x80000000u >> j);
                        if (slot < 64) slist[slot * 64 + lane] = (unsigned short)(w * 32 + j);
                        ++slot;
                    }
                }
            }
        }
    } else {
        const f32x2 qx2 = {fqx, fqx}, qy2 = {fqy, fqy}, qz2 = {fqz, fqz};
        const f32x2 nr2 = {-r2f, -r2f};            // (+1 for lanes without a usable query: never negative)
        for (int w = wave; w < nw; w += 4) {
            // one mask word = 2 x 8 pairs: a half's LDS reads are all issued before its arithmetic
            u32 m = 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x2 X[8], Y[8], Z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int o = 2 * (w * 16 + hh * 8 + j);               // the pair of slots o, o + 1
                    X[j] = *reinterpret_cast<const f32x2*>(&tx[o]);
                    Y[j] = *reinterpret_cast<const f32x2*>(&ty[o]);
                    Z[j] = *reinterpret_cast<const f32x2*>(&tz[o]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // d2 - r2 accumulated from -r2 (three packed fmas for two candidates); negative = inside
                    const f32x2 dx = X[j] - qx2, dy = Y[j] - qy2, dz = Z[j] - qz2;
                    f32x2 e2 = __builtin_elementwise_fma(dx, dx, nr2);
                    e2 = __builtin_elementwise_fma(dy, dy, e2);
                    e2 = __builtin_elementwise_fma(dz, dz, e2);
                    m = shift_in_sign(m, e2.x);
                    m = shift_in_sign(m, e2.y);
                }
            }
            // the word's set bits -> the query's slot list (candidates entered at bit 0 and moved up: first = highest);
            // the four waves append through the query's counter (the order of a list does not matter: it is sorted).
            // ONE reservation per lane and word (a returning LDS atomic is a round trip; in the loop - one per set bit -
            // it was a chain of up to seven per word for the wave's busiest lane), then plain stores.
            if (__builtin_amdgcn_ballot_w64(m != 0u)) {
                int slot = 0;
                if (m != 0u) slot = atomicAdd(&cntq[lane], __popc(m));
                while (__builtin_amdgcn_ballot_w64(m != 0u)) {
                    if (m != 0u) {
                        const int j = __clz((int)m);
                        m &= ~(0x80000000u >> j);
                        if (slot < 64) slist[slot * 64 + lane] = (unsigned short)(w * 32 + j);
                        ++slot;
                    }
                }
            }
        }
    }
    __syncthreads();
    KG_STAMP(3)
    // ---- the tail: wave w finishes queries 16 w .. 16 w + 15 with FOUR LANES PER QUERY ------------------------------
    // (lane = 4 * local query + part).  The query's slot list is dealt cyclically over its four lanes (slot = part +
    // 4 u: balanced by construction), each lane builds 16 keys, and the 64 keys of a query are ordered by a bitonic
    // network whose element index is  e = 16 * part + u : the 18 stages with partner distance < 16 are compare-
    // exchanges between a lane's own registers, the 3 stages at distance 16 / 32 exchange with lane ^ 1 / lane ^ 2 of
    // the quad through DPP.  All four waves work to the end (the first version left the tail to wave 0 alone: 46 % of
    // a group's lifetime with one wave in four busy).
    const int ql = lane >> 2, part = lane & 3;
    const int src = wave * 16 + ql;                        // the lane that holds this query's data in every wave
    const bool q_ok0 = __shfl((int)ok, src, 64) != 0;
    const bool q_isq = __shfl((int)is_query, src, 64) != 0;
    const bool q_fail0 = __shfl((int)fail, src, 64) != 0;
    int q_why = __shfl(why, src, 64);
    const float tqx = __shfl(fqx, src, 64), tqy = __shfl(fqy, src, 64), tqz = __shfl(fqz, src, 64);
    const double Rc_q = __shfl(Rc, src, 64);
    const double tol_q = __shfl(tol_rel, src, 64);
    const int qs_q = __shfl(qs, src, 64), qid_q = __shfl(qid, src, 64);
    const int pq = xcd_block(blockIdx.x, gridDim.x) * 64 + src;             // the query's processing slot = list column
    const int cnt = cntq[src];
    bool okq = q_ok0, failq = q_fail0;
    // (a query with fewer than K inside its radius tells the general kernel how much wider to start: the count sizes the
    //  step as in its own ladder, x1.6 .. x3 in eight steps, carried in the list entry's top three bits)
    unsigned grow_code = 0;
    if (okq && (cnt > 64 || cnt < K)) {
        okq = false; failq = true; q_why = cnt > 64 ? 4 : 5;
        if (cnt < K) {
            const float gr = fminf(fmaxf(cbrtf(1.5f * (float)K / (float)(cnt > 0 ? cnt : 1)), 1.6f), 3.0f);
            grow_code = 1u + (unsigned)((gr - 1.6f) * (6.0f / 1.4f) + 0.5f);
        }
    }
    const int maxcnt = (int)wmax((double)(okq ? cnt : 0));

    // ---- phase B: the lane's 16 list entries -> keys ----
    u32 key[16];
    const float kscale = okq ? (float)(2097152.0 / (Rc_q * Rc_q * 1.0002)) : 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {                       // 8 entries at a time: list reads, then tile reads, then arithmetic
        if (c * 32 < maxcnt) {
            int tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tt[u] = slist[(part + 4 * (c * 8 + u)) * 64 + src];
            float cx[8], cy[8], cz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int o = tt[u] & 2047;
                cx[u] = tx[o]; cy[u] = ty[o]; cz[u] = tz[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dx = cx[u] - tqx, dy = cy[u] - tqy, dz = cz[u] - tqz;
                float d2 = dx * dx;
                d2 = fmaf(dy, dy, d2);
                d2 = fmaf(dz, dz, d2);
                u32 qd = (u32)(d2 * kscale);
                qd = qd > 2097151u ? 2097151u : qd;
                key[c * 8 + u] = (okq && part + 4 * (c * 8 + u) < cnt) ? ((qd << 11) | (u32)tt[u]) : 0xFFFFFFFFu;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) key[c * 8 + u] = 0xFFFFFFFFu;
        }
    }
    KG_STAMP(4)
    // ---- order: bitonic network on e = 16 * part + u (ascending) ----
    // for k = 2 .. 64, j = k/2 .. 1: compare-exchange (e, e | j), ascending where (e & k) == 0.  j < 16: both elements in
    // one lane; j = 16 / 32: partner in lane ^ 1 / lane ^ 2 (DPP quad_perm).  The direction is a compile-time property
    // of the register pair for k <= 8; for k = 16 it depends on part & 1, for k = 32 on part & 2: lanes that must sort
    // descending carry COMPLEMENTED keys through that phase (descending on x = ascending on ~x), so every
    // compare-exchange is the same min/max pair in all lanes.
    {
#define KG_CEA(a_, b_) { const u32 lo_ = min(key[a_], key[b_]); const u32 hi_ = max(key[a_], key[b_]); key[a_] = lo_; key[b_] = hi_; }
#define KG_CED(a_, b_) { const u32 lo_ = min(key[a_], key[b_]); const u32 hi_ = max(key[a_], key[b_]); key[a_] = hi_; key[b_] = lo_; }
#pragma unroll
        for (int k2 = 2; k2 <= 8; k2 <<= 1) {
#pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if ((u & j) == 0) {
                        if ((u & k2) == 0) KG_CEA(u, u | j) else KG_CED(u, u | j)
                    }
            }
        }
        const u32 c16 = (part & 1) ? 0xFFFFFFFFu : 0u;         // k = 16: odd parts sort descending
        const u32 c32 = (part & 2) ? 0xFFFFFFFFu : 0u;         // k = 32: parts 2, 3 merge descending
#pragma unroll
        for (int u = 0; u < 16; ++u) key[u] ^= c16;
#pragma unroll
        for (int j = 8; j > 0; j >>= 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if ((u & j) == 0) KG_CEA(u, u | j)
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) key[u] ^= (c16 ^ c32);    // out of the k = 16 phase, into the k = 32 phase
        // k = 32, j = 16: lane ^ 1; the lower lane of the pair (part & 1 == 0) keeps the minimum
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u32 pk = xchg32<1>(key[u]);
            const u32 lo_ = min(key[u], pk), hi_ = max(key[u], pk);
            key[u] = select_const<0x5555555555555555ull>(lo_, hi_);
        }
#pragma unroll
        for (int j = 8; j > 0; j >>= 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if ((u & j) == 0) KG_CEA(u, u | j)
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) key[u] ^= c32;            // true keys again; k = 64 is ascending everywhere
        // k = 64, j = 32: lane ^ 2 (lower: part & 2 == 0), j = 16: lane ^ 1
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u32 pk = xchg32<2>(key[u]);
            const u32 lo_ = min(key[u], pk), hi_ = max(key[u], pk);
            key[u] = select_const<0x3333333333333333ull>(lo_, hi_);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u32 pk = xchg32<1>(key[u]);
            const u32 lo_ = min(key[u], pk), hi_ = max(key[u], pk);
            key[u] = select_const<0x5555555555555555ull>(lo_, hi_);
        }
#pragma unroll
        for (int j = 8; j > 0; j >>= 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if ((u & j) == 0) KG_CEA(u, u | j)
        }
    }
    // rank r of a query now sits in lane part = r / 16, register r % 16

    // ---- certify: consecutive keys among the first K+1 further apart than twice the error bound ----
    u32 res16 = 0;                              // bit u: the near tie of ranks 16 part + u, + 1 goes to the tie list
    int tie_base = 0;
    {
        // bins of 2^-21 R^2(1.0002): each d2 within tol of the truth, each key a floor of (d2 x a rounded scale): two keys
        // further apart than 2 tol + 2 bins are in their true order
        const u32 win = okq ? (u32)(2.0 * tol_q * 2097152.0) + 3u : 0u;
        const u32 nextfirst = (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)key[0], 0x39, 0xF, 0xF, false);  // quad_perm [1,2,3,0]: lane + 1
        // A particle that was NOT listed has d2 > R^2 (1.0002 - tol) in truth: the K-th listed one must lie below that with
        // the same margin again, or an unlisted particle could be the true K-th (the list is then too short in truth)
        const u32 safe = okq ? (u32)(2097152.0 * (1.0 - 2.0 * tol_q / 1.0002)) - 2u : 0u;
        bool amb = false, edge = false;
        u32 g16 = 0;                            // bit u: ranks r = 16 part + u and r + 1 (r <= K) are closer than the window
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = part * 16 + u;
            const u32 nxt = (u < 15) ? key[(u + 1) & 15] : (part < 3 ? nextfirst : 0xFFFFFFFFu);
            if (r <= K && nxt != 0xFFFFFFFFu && (nxt >> 11) - (key[u] >> 11) <= win) g16 |= 1u << u;
            if (r == K - 1 && (key[u] >> 11) > safe) edge = true;
        }
        {
            // A near tie of TWO consecutive ranks with certain neighbours on both sides (ranks r - 1 | r, r + 1 | r + 2 in
            // certain order) is left to the tie blocks of the list-mode launch (sphx_knn.hip), which order the two by their exact fp64 distances (index breaks
            // exact ties, as in the general kernel) - half of what this kernel used to hand on.  Chains of
            // three and pairs that straddle two lanes still fail over.
            const int nk = K - part * 16;                                    // ranks r < K held by this lane
            const u32 kmask = nk >= 16 ? 0xFFFFu : (nk <= 0 ? 0u : ((1u << nk) - 1u));
            const u32 prevG = part > 0 ? (((u32)__builtin_amdgcn_update_dpp(0, (int)g16, 0x90, 0xF, 0xF, false) >> 15) & 1u) : 0u;   // quad_perm [0,0,1,2]: lane - 1
            const u32 a16 = g16 & kmask;
            res16 = (a.tie_list && okq) ? (a16 & ~((g16 << 1) | prevG) & ~(g16 >> 1) & 0x7FFFu) : 0u;
            amb = (a16 & ~res16) != 0u;
        }
        // any of the query's four lanes
        int am = (amb ? 1 : 0) | (edge ? 2 : 0);
        am |= __builtin_amdgcn_update_dpp(0, am, 0xB1, 0xF, 0xF, true);
        am |= __builtin_amdgcn_update_dpp(0, am, 0x4E, 0xF, 0xF, true);
        if (okq && (am & 2)) { okq = false; failq = true; q_why = 5; }
        #ifdef SPHX_EXPERIMENTS
        if (a.exp_noamb) am &= ~1;
#endif
        if (okq && (am & 1)) { okq = false; failq = true; q_why = 6; }
        // room in the tie list is reserved here: a query whose entries do not fit fails over like any other near tie
        if (__builtin_amdgcn_ballot_w64(okq && res16 != 0u)) {          // (rare: under 1 % of the queries)
            const int ne = (okq && res16 != 0u) ? __popc(res16) : 0;
            if (ne) tie_base = atomicAdd(a.tie_count, ne);
            int ov = (ne && tie_base + ne > a.tie_cap) ? 1 : 0;
            ov |= __builtin_amdgcn_update_dpp(0, ov, 0xB1, 0xF, 0xF, true);
            ov |= __builtin_amdgcn_update_dpp(0, ov, 0x4E, 0xF, 0xF, true);
            if (okq && ov) { okq = false; failq = true; q_why = 6; }
        }
    }
    KG_STAMP(5)
    // ---- outputs: rank r = 16 part + u ----
    __builtin_amdgcn_s_setprio(KG_FRONT_PRIO);          // (index and position loads of the K-th neighbour, row stores: -3 us)
    {
        const int* tidx = ti;
        const bool wr = pq < a.npad && (okq || !q_isq);    // failed queries are written by the list-mode launch that
                                                            // follows; non-queries (ghosts, padding) get -1
        int idx[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) idx[u] = (part * 16 + u < K) ? tidx[okq ? (key[u] & 2047u) : 0u] : -1;
        // the K-th neighbour's exact distance: its loads go out before the row stores
        const int rl = K - 1;
        int lastidx = -1;
#pragma unroll
        for (int u = 0; u < 16; ++u) if ((rl & 15) == u) lastidx = idx[u];
        const bool mine = okq && part == (rl >> 4);
        double lx = 0.0, ly = 0.0, lz = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
        if (mine) {
            lx = a.x[lastidx]; ly = a.y[lastidx]; lz = a.z[lastidx];
            sx = a.x[qs_q]; sy = a.y[qs_q]; sz = a.z[qs_q];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = part * 16 + u;
            if (r < K && wr) a.nbr[(size_t)r * a.npad + pq] = okq ? idx[u] : -1;
        }
        if (mine) {
            const double hval = sqrt(dist2_nofma(lx - sx, ly - sy, lz - sz));
            if (a.h_by_id) a.h_by_id[qid_q] = hval; else a.h_sorted[qs_q] = hval;
        }
        if (__builtin_amdgcn_ballot_w64(okq && res16 != 0u)) {
            int base = tie_base;
            const u32 mine16 = okq ? res16 : 0u;
#pragma unroll
            for (int u = 0; u < 15; ++u) {
                if ((mine16 >> u) & 1u) {
                    a.tie_list[base] = make_int4(pq, part * 16 + u, tidx[key[u] & 2047u], tidx[key[u + 1] & 2047u]);
                    ++base;
                }
            }
        }
    }
    KG_STAMP(6)
    const bool rep = part == 0;                            // one lane speaks for the query
    const u64 failmask = __builtin_amdgcn_ballot_w64(failq && rep);
    if (failmask) {
        int base = 0;
        if (lane == 0) base = atomicAdd(a.fail_count, __popcll(failmask));
        base = __builtin_amdgcn_readfirstlane(base);
        if (failq && rep) a.fail_list[base + lanes_below(failmask)] = (int)((unsigned)pq | (grow_code << 29));
        if (a.counters) {
            for (int wq = 1; wq <= 6; ++wq) {
                const u64 mk = __builtin_amdgcn_ballot_w64(failq && rep && q_why == wq);
                if (lane == 0 && mk) atomicAdd(&a.counters[SC_KGDBG + wq], (u64)__popcll(mk));
            }
        }
    }
    if (tid == 0 && a.counters) atomicAdd(&a.counters[SC_CAND], (u64)T * (u64)__popcll(okmask));
    KG_STAMP(7)
}

int sphx_knn_group(sphx_ctx* ctx, const KnnGroupArgs& a0) {
    KnnGroupArgs a = a0;
    const int blocks = a.npad / 64;
    a.prof = nullptr;
#ifdef SPHX_EXPERIMENTS
    static const bool noamb = getenv("SPHX_KG_EXP_NOAMB") != nullptr;     // timing experiment only: near ties NOT handed on (wrong results)
    a.exp_noamb = noamb ? 1 : 0;
    static const bool prof = getenv("SPHX_KG_PROF") != nullptr;
#else
    a.exp_noamb = 0;
    const bool prof = false;
#endif
    if (prof) {           // diagnostic: per-section shader cycles of one launch (summed over the waves' lane 0)
        SPHX_TRY(sphx_ensure(ctx, ctx->scal_tmp, 4096));
        u64* pd = ctx->scal_tmp.as<u64>() + 256;
        HIPCHK(hipMemsetAsync(pd, 0, 64, ctx->stream));
        a.prof = pd;
    }
    hipLaunchKernelGGL(knn_group_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    if (prof) {
        u64 h[8];
        HIPCHK(hipMemcpyAsync(h, a.prof, 64, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const double w = (double)blocks;
        fprintf(stderr, "[sphx] grouped search cycles/group (thread 0): setup %.0f rows %.0f stage %.0f phaseA %.0f phaseB %.0f sort+certify %.0f output %.0f tail %.0f\n",
                h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, h[7] / w);
    }
    return SPHX_OK;
}

// ---- self-test of the matrix-core form's error bound on the hardware it runs on (tests/test_gpu_parity.py) ----
// One wave per block; every block draws 32 candidates and 32 queries with coordinates uniform in [-E, E] and radii in
// [0.2 E, E], forms D as phase A does and compares all 1024 entries with d^2 - R^2 (1 + pad) in fp64 from the same fp32
// inputs.  out[0] = max |D - exact| / E^2 (the bound is KG_MFMA_KAPPA), out[1] = entries whose sign differs although
// |exact| > KG_MFMA_KAPPA E^2 (must be 0).
__global__ __launch_bounds__(64) void kg_selftest_kernel(float E, unsigned seed, double* out) {
    __shared__ float cx[32], cy[32], cz[32], qx[32], qy[32], qz[32], qr[32];
    const int lane = threadIdx.x, n32 = lane & 31, hh = lane >> 5;
    unsigned st = seed * 2654435761u + (blockIdx.x * 64u + lane) * 40503u + 12345u;
    auto rnd = [&]() -> float { st = st * 1664525u + 1013904223u; return (float)(st >> 8) * (1.0f / 16777216.0f); };
    if (hh == 0) { cx[n32] = (2.f * rnd() - 1.f) * E; cy[n32] = (2.f * rnd() - 1.f) * E; cz[n32] = (2.f * rnd() - 1.f) * E; }
    else {
        qx[n32] = (2.f * rnd() - 1.f) * E; qy[n32] = (2.f * rnd() - 1.f) * E; qz[n32] = (2.f * rnd() - 1.f) * E;
        const float R = (0.2f + 0.8f * rnd()) * E;
        qr[n32] = (float)((double)R * (double)R * (1.0 + KG_MFMA_PAD));
    }
    __syncthreads();
    const f16x8 A = kg_cand_frag(cx[n32], cy[n32], cz[n32], hh);
    const f16x8 B = kg_query_frag(qx[n32], qy[n32], qz[n32], qr[n32], hh);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero16, 0, 0, 0);
    double worst = 0.0, wrong = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 4 * hh + 8 * (r >> 2) + (r & 3);
        const double dx = (double)cx[row] - (double)qx[n32], dy = (double)cy[row] - (double)qy[n32], dz = (double)cz[row] - (double)qz[n32];
        const double exact = dx * dx + dy * dy + dz * dz - (double)qr[n32];
        const double err = fabs((double)D[r] - exact) / ((double)E * (double)E);
        worst = fmax(worst, err);
        if (fabs(exact) > KG_MFMA_KAPPA * (double)E * (double)E && ((D[r] < 0.0f) != (exact < 0.0))) wrong += 1.0;
    }
    worst = wmax(worst);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wrong += __shfl_xor(wrong, o, 64);
    if (lane == 0) {
        atomicMax(reinterpret_cast<unsigned long long*>(&out[0]), (unsigned long long)__double_as_longlong(worst));
        atomicAdd(&out[1], wrong);
    }
}
extern "C" int sphx_selftest_mfma_cull(sphx_ctx* ctx, double E, int blocks, unsigned seed, double* max_err_over_E2,
                                       double* wrong_signs, double* kappa) {
    if (!ctx || !max_err_over_E2 || !wrong_signs || blocks < 1 || !(E > 0.0) || E > KG_MFMA_EMAX)
        return sphx_set_err(ctx, SPHX_E_ARG, "sphx_selftest_mfma_cull: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_ensure(ctx, ctx->scal_tmp, 4096));
    double* out = ctx->scal_tmp.as<double>();
    HIPCHK(hipMemsetAsync(out, 0, 2 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(kg_selftest_kernel, dim3(blocks), dim3(64), 0, ctx->stream, (float)E, seed, out);
    HIPCHK(hipGetLastError());
    double h[2];
    HIPCHK(hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *max_err_over_E2 = h[0]; *wrong_signs = h[1];
    if (kappa) *kappa = KG_MFMA_KAPPA;
    return SPHX_OK;
}
