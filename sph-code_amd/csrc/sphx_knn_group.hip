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
//   3. phase A: every lane runs over the whole tile with broadcast LDS reads (one ds_read_b128 per
//      candidate per wave) and shifts "inside my radius" into a per-lane bit mask (v_addc: one instruction);
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

// The tile holds the candidates in PAIRS, laid out for packed fp32 arithmetic (v_pk_add/mul/fma_f32 work on
// two candidates at once) and for whole-width LDS reads:  tile_xy[pair] = {x0, x1, y0, y1},
// tile_zi[pair] = {z0, z1, index0, index1}.  Phase A reads one b128 + one b64 per pair (3 LDS cycles per
// candidate; a 12-byte read of an {x,y,z,idx} record would cost 8).
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(16))) TileZI { float z0, z1; int i0, i1; };

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
    int ry0, rz0, ysp, nrows, T, ok;
    int wtot[4];
};

#ifndef KG_MINWAVES
#define KG_MINWAVES 4
#endif
__global__ __launch_bounds__(256, KG_MINWAVES) void knn_group_kernel(KnnGroupArgs a) {
    __shared__ f32x4 tile_xy[KG_TCAP / 2];
    __shared__ TileZI tile_zi[KG_TCAP / 2];
    float* txy = reinterpret_cast<float*>(tile_xy);
    float* tzi = reinterpret_cast<float*>(tile_zi);
    // slot t: x at txy[(t >> 1) * 4 + (t & 1)], y two floats on; z, index likewise in tzi
#define KG_OFF(t) ((((t) >> 1) << 2) | ((t) & 1))
    __shared__ unsigned short slist[64 * 64];       // [slot][query]: tile slots inside the query's radius;
                                                    // while staging: row of each slot (u8) + the rows' bases
    static_assert(KG_TCAP * 2 + KG_MAXROWS * 4 <= 64 * 64 * 2, "slot->row ids + row bases share the list's memory");
    static_assert(KG_MAXROWS == 512, "two rows per thread");
    unsigned short* rowof = slist;                                               // KG_TCAP row ids
    int* rowbase = reinterpret_cast<int*>(slist) + KG_TCAP / 2;                  // KG_MAXROWS ints
    __shared__ int cntq[64];
    __shared__ KgShared sh;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // in-kernel section clock (diagnostic launches only: a.prof == nullptr in the product)
    u64 tprev = a.prof ? __builtin_readcyclecounter() : 0;
#define KG_STAMP(sec) if (a.prof) { const u64 tn_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&a.prof[sec], tn_ - tprev); tprev = tn_; }
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
            sh.cx0 = fmin(fmax(bx0, g.xmin), gx1); sh.cx1 = fmin(fmax(bx1, g.xmin), gx1);
            sh.cy0 = cy0; sh.cy1 = cy1; sh.cz0 = cz0; sh.cz1 = cz1;
            sh.R2 = Rcov * Rcov; sh.Rcov = Rcov;
            sh.ry0 = ry0; sh.rz0 = rz0; sh.ysp = ry1 - ry0 + 1;
            sh.nrows = (ry1 - ry0 + 1) * (rz1 - rz0 + 1);
            sh.T = 0;
            sh.ok = (okmask != 0ull && sh.nrows <= KG_MAXROWS) ? 1 : 0;
        }
    }
    __syncthreads();
    KG_STAMP(0)
    bool group_ok = sh.ok != 0;
    const double Ox = sh.Ox, Oy = sh.Oy, Oz = sh.Oz;
    int T = 0;
    if (group_ok) {
        // ---- two rows of cells per thread (rows 2 tid, 2 tid + 1): the chord of cells that can hold a particle
        // within Rcov of the box ----
        const double INF = INFINITY;
        int s_row[2] = {0, 0}, cnt[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = 2 * tid + u;
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
        const int incl = wave_scan_incl(cnt[0] + cnt[1]);
        if (lane == 63) sh.wtot[wave] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int tw = sh.wtot[w]; if (w < wave) woff += tw; T += tw; }
        if (T > KG_TCAP) group_ok = false;                           // (uniform over the workgroup)
        if (group_ok) {
            int off = woff + incl - (cnt[0] + cnt[1]);                // the first slot of the thread's rows in the tile
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                rowbase[2 * tid + u] = s_row[u] - off;                // slot t of row r is particle rowbase[r] + t
                for (int q = 0; q < cnt[u]; ++q) rowof[off + q] = (unsigned short)(2 * tid + u);
                off += cnt[u];
            }
        }
        __syncthreads();
        KG_STAMP(1)
        // ---- stage: one slot per thread and pass; two passes' loads in flight ----
        if (group_ok) {
            for (int t0 = 0; t0 < T; t0 += 512) {
                int pp[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + 256 * u + tid;
                    pp[u] = t < T ? rowbase[rowof[t]] + t : -1;
                }
                double X[2], Y[2], Z[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = pp[u] >= 0 ? pp[u] : 0;
                    X[u] = a.x[q]; Y[u] = a.y[q]; Z[u] = a.z[q];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (pp[u] >= 0) {
                        const int o = KG_OFF(t0 + 256 * u + tid);
                        txy[o] = (float)((X[u] - Ox) * g.inv_cell);
                        txy[o + 2] = (float)((Y[u] - Oy) * g.inv_cell);
                        tzi[o] = (float)((Z[u] - Oz) * g.inv_cell);
                        reinterpret_cast<int*>(tzi)[o + 2] = pp[u];
                    }
                }
            }
            // pad the tile to whole mask words with records nobody accepts
            const int nwp = (T + 31) >> 5;
            if (tid < nwp * 32 - T) {
                const int o = KG_OFF(T + tid);
                txy[o] = 1e30f; txy[o + 2] = 1e30f; tzi[o] = 1e30f;
                reinterpret_cast<int*>(tzi)[o + 2] = -1;
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

    // ---- phase A: every query against the tile, the waves splitting it by mask word -----------------
    // accept d2 <= Rc^2 (1 + 2e-4): with err(d2) <= 1e-4 Rc^2 (checked below) every candidate truly inside R_i is
    // accepted, and every accepted one lies within R_i (1 + 1.6e-4) < the radius the tile covers
    const double Rc = R * g.inv_cell;
    const float fqx = (float)((qx - Ox) * g.inv_cell), fqy = (float)((qy - Oy) * g.inv_cell), fqz = (float)((qz - Oz) * g.inv_cell);
    // err(d2)/Rc^2 <= eps (6.93 E/Rc + 5); 7.5 taken
    const double tol_rel = ok ? 5.9604644775390625e-08 * (6.93 * sh.E / Rc + 7.5) : 0.0;
    if (ok && !(tol_rel <= 0.5e-4)) { ok = false; fail = true; why = 3; }
    const float r2f = ok ? (float)(Rc * Rc * 1.0002) : -1.0f;
    {
        const f32x2 qx2 = {fqx, fqx}, qy2 = {fqy, fqy}, qz2 = {fqz, fqz};
        const f32x2 nr2 = {-r2f, -r2f};            // (+1 for lanes without a usable query: never negative)
        for (int w = wave; w < nw; w += 4) {
            // one mask word = 2 x 8 pairs: a half's LDS reads are all issued before its arithmetic
            u32 m = 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 A[8];
                f32x2 Z[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    A[j] = tile_xy[w * 16 + hh * 8 + j];
                    Z[j] = *reinterpret_cast<const f32x2*>(&tile_zi[w * 16 + hh * 8 + j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // d2 - r2 accumulated from -r2 (three packed fmas for two candidates); negative = inside
                    const f32x2 dx = A[j].xy - qx2, dy = A[j].zw - qy2, dz = Z[j] - qz2;
                    f32x2 e2 = __builtin_elementwise_fma(dx, dx, nr2);
                    e2 = __builtin_elementwise_fma(dy, dy, e2);
                    e2 = __builtin_elementwise_fma(dz, dz, e2);
                    m = shift_in_sign(m, e2.x);
                    m = shift_in_sign(m, e2.y);
                }
            }
            // the word's set bits -> the query's slot list (candidates entered at bit 0 and moved up: first = highest);
            // the four waves append through the query's counter (the order of a list does not matter: it is sorted)
            while (__builtin_amdgcn_ballot_w64(m != 0u)) {
                if (m != 0u) {
                    const int j = __clz((int)m);
                    m &= ~(0x80000000u >> j);
                    const int slot = atomicAdd(&cntq[lane], 1);
                    if (slot < 64) slist[slot * 64 + lane] = (unsigned short)(w * 32 + j);
                }
            }
        }
    }
    __syncthreads();
    KG_STAMP(3)
    if (wave != 0) return;               // wave 0 finishes the group
    const int cnt = cntq[lane];
    if (ok && (cnt > 64 || cnt < K)) { ok = false; fail = true; why = cnt > 64 ? 4 : 5; }
    const int maxcnt = (int)wmax((double)(ok ? cnt : 0));

    // ---- phase B: slot lists -> keys in registers ----
    u32 key[64];
    const float kscale = ok ? (float)(2097152.0 / (Rc * Rc * 1.0002)) : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {                       // 8 slots at a time: list reads, then tile reads, then arithmetic
        if (c * 8 < maxcnt) {
            int tt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tt[u] = slist[(c * 8 + u) * 64 + lane];
            float cx[8], cy[8], cz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int o = KG_OFF(tt[u] & 2047);
                cx[u] = txy[o]; cy[u] = txy[o + 2]; cz[u] = tzi[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dx = cx[u] - fqx, dy = cy[u] - fqy, dz = cz[u] - fqz;
                float d2 = dx * dx;
                d2 = fmaf(dy, dy, d2);
                d2 = fmaf(dz, dz, d2);
                u32 qd = (u32)(d2 * kscale);
                qd = qd > 2097151u ? 2097151u : qd;
                key[c * 8 + u] = (ok && c * 8 + u < cnt) ? ((qd << 11) | (u32)tt[u]) : 0xFFFFFFFFu;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) key[c * 8 + u] = 0xFFFFFFFFu;
        }
    }
    KG_STAMP(4)
    // ---- order the 64 registers ----
    OemP<1>::run(key);

    // ---- certify: consecutive keys among the first K+1 further apart than twice the error bound ----
    {
        // bins of 2^-21 R^2(1.0002): each d2 within tol of the truth, each key a floor of (d2 x a rounded scale): two keys
        // further apart than 2 tol + 2 bins are in their true order
        const u32 win = ok ? (u32)(2.0 * tol_rel * 2097152.0) + 3u : 0u;
        bool amb = false;
#pragma unroll
        for (int r = 0; r < 63; ++r)
            if (r < K && key[r + 1] != 0xFFFFFFFFu && (key[r + 1] >> 11) - (key[r] >> 11) <= win) amb = true;
        if (ok && amb) { ok = false; fail = true; why = 6; }
    }
    KG_STAMP(5)
    // ---- outputs: 8 list positions at a time (index reads from the tile batched, then the row stores) ----
    {
        int lastidx = -1;
        const int* tidx = reinterpret_cast<const int*>(tzi);
        const bool wr = p < a.npad && (ok || !is_query);   // failed queries are written by the list-mode launch that
                                                            // follows; non-queries (ghosts, padding) get -1
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c * 8 < K) {
                int idx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) idx[u] = tidx[KG_OFF(ok ? (key[c * 8 + u] & 2047u) : 0u) + 2];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = c * 8 + u;
                    const int v = ok ? idx[u] : -1;
                    if (r == K - 1) lastidx = v;
                    if (r < K && wr) a.nbr[(size_t)r * a.npad + p] = v;
                }
            }
        }
        if (ok) {
            const double d2 = dist2_nofma(a.x[lastidx] - qx, a.y[lastidx] - qy, a.z[lastidx] - qz);
            const double hval = sqrt(d2);
            if (a.h_by_id) a.h_by_id[qid] = hval; else a.h_sorted[qs] = hval;
        }
    }
    KG_STAMP(6)
    const u64 failmask = __builtin_amdgcn_ballot_w64(fail);
    if (failmask) {
        int base = 0;
        if (lane == 0) base = atomicAdd(a.fail_count, __popcll(failmask));
        base = __builtin_amdgcn_readfirstlane(base);
        if (fail) a.fail_list[base + lanes_below(failmask)] = p;
        if (a.counters) {
            for (int wq = 1; wq <= 6; ++wq) {
                const u64 mk = __builtin_amdgcn_ballot_w64(fail && why == wq);
                if (lane == 0 && mk) atomicAdd(&a.counters[SC_KGDBG + wq], (u64)__popcll(mk));
            }
        }
    }
    if (lane == 0 && a.counters) atomicAdd(&a.counters[SC_CAND], (u64)T * (u64)__popcll(okmask));
    KG_STAMP(7)
}

int sphx_knn_group(sphx_ctx* ctx, const KnnGroupArgs& a0) {
    KnnGroupArgs a = a0;
    const int blocks = a.npad / 64;
    a.prof = nullptr;
    static const bool prof = getenv("SPHX_KG_PROF") != nullptr;
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
