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

#define KG_WORDS (KG_TCAP / 32)
#define KG_FLAGCAP KG_TCAP               // candidates per 64-row chunk served by the flag lookup

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The tile holds the candidates in PAIRS, laid out for packed fp32 arithmetic (v_pk_add/mul/fma_f32 work on
// two candidates at once) and for whole-width LDS reads:  tile_xy[pair] = {x0, x1, y0, y1},
// tile_zi[pair] = {z0, z1, index0, index1}.  Phase A reads one b128 + one b64 per pair (3 LDS cycles per
// candidate; a 12-byte read of an {x,y,z,idx} record would cost 8).
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(16))) TileZI { float z0, z1; int i0, i1; };

__device__ __forceinline__ double wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int cell_of_coord(double v, double vmin, double inv_cell, int nmax1) {
    double t = (v - vmin) * inv_cell;
    t = fmin(fmax(t, 0.0), (double)nmax1);
    return (int)t;
}

// m = (m << 1) | (d2 <= r2): compare into VCC, add-with-carry shifts it in
__device__ __forceinline__ u32 shift_in_le(u32 m, float d2, float r2) {
    asm("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
        : "+v"(m) : "v"(d2), "v"(r2) : "vcc");
    return m;
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

__global__ __launch_bounds__(64) void knn_group_kernel(KnnGroupArgs a) {
    __shared__ f32x4 tile_xy[KG_TCAP / 2];
    __shared__ TileZI tile_zi[KG_TCAP / 2];
    float* txy = reinterpret_cast<float*>(tile_xy);
    float* tzi = reinterpret_cast<float*>(tile_zi);
    // slot t: x at txy[(t >> 1) * 4 + (t & 1)], y two floats on; z, index likewise in tzi
#define KG_OFF(t) ((((t) >> 1) << 2) | ((t) & 1))
    __shared__ unsigned short slist[64 * 64];       // [slot][lane]: tile slots inside the lane's radius, in tile order;
                                                    // the row flags / bases live here while staging
    static_assert(KG_FLAGCAP + 256 <= 64 * 64 * 2, "row flags + bases share the list's memory");
    unsigned char* rflag = reinterpret_cast<unsigned char*>(slist);              // KG_FLAGCAP bytes
    int* rbase = reinterpret_cast<int*>(slist) + KG_FLAGCAP / 4;                 // 64 ints

    const int lane = threadIdx.x;
    // in-kernel section clock (diagnostic build of the launch only: a.prof == nullptr in the product)
    u64 tprev = a.prof ? __builtin_readcyclecounter() : 0;
#define KG_STAMP(sec) if (a.prof) { const u64 tn_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&a.prof[sec], tn_ - tprev); tprev = tn_; }
    const int p = xcd_block(blockIdx.x, gridDim.x) * 64 + lane;       // processing slot = list column
    const GridParams g = a.g;
    const int K = a.k;

    // ---- the group's queries ---------------------------------------------------------------------
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
    int T = 0;
    double Ox = 0.0, Oy = 0.0, Oz = 0.0, E = 0.0;
    bool group_ok = okmask != 0ull;
    if (group_ok) {
        const double INF = INFINITY;
        double bx0 = wmin(ok ? qx : INF), bx1 = wmax(ok ? qx : -INF);
        double by0 = wmin(ok ? qy : INF), by1 = wmax(ok ? qy : -INF);
        double bz0 = wmin(ok ? qz : INF), bz1 = wmax(ok ? qz : -INF);
        const double Rmax = wmax(ok ? R : 0.0);
        const double Rcov = Rmax * 1.0002;           // an accepted candidate lies within R_i (1 + 1.6e-4), see below
        Ox = 0.5 * (bx0 + bx1); Oy = 0.5 * (by0 + by1); Oz = 0.5 * (bz0 + bz1);
        E = (fmax(fmax(bx1 - bx0, by1 - by0), bz1 - bz0) * 0.5 + Rcov) * g.inv_cell * 1.0001;   // cell units
        // rows are tested against the box clamped into the grid: the boundary cells are half-infinite (out-of-box
        // particles are clamped into them, sphx_grid.hip), their open side can only lie behind the clamped box
        const double gx1 = g.xmin + g.nx * g.cell, gy1 = g.ymin + g.ny * g.cell, gz1 = g.zmin + g.nz * g.cell;
        const double cx0 = fmin(fmax(bx0, g.xmin), gx1), cx1 = fmin(fmax(bx1, g.xmin), gx1);
        const double cy0 = fmin(fmax(by0, g.ymin), gy1), cy1 = fmin(fmax(by1, g.ymin), gy1);
        const double cz0 = fmin(fmax(bz0, g.zmin), gz1), cz1 = fmin(fmax(bz1, g.zmin), gz1);
        const int ry0 = cell_of_coord(cy0 - Rcov, g.ymin, g.inv_cell, g.ny - 1);
        const int ry1 = cell_of_coord(cy1 + Rcov, g.ymin, g.inv_cell, g.ny - 1);
        const int rz0 = cell_of_coord(cz0 - Rcov, g.zmin, g.inv_cell, g.nz - 1);
        const int rz1 = cell_of_coord(cz1 + Rcov, g.zmin, g.inv_cell, g.nz - 1);
        const int ysp = ry1 - ry0 + 1;
        const int nrows = ysp * (rz1 - rz0 + 1);
        if (nrows > KG_MAXROWS) group_ok = false;
        const double R2 = Rcov * Rcov;
        KG_STAMP(0)
        // ---- stage the candidates: 64 rows of cells at a time, one lane per row ------------------
        for (int q = lane; q < KG_FLAGCAP / 4; q += 64) reinterpret_cast<u32*>(rflag)[q] = 0u;
        wave_sync();
        for (int rb = 0; group_ok && rb < nrows; rb += 64) {
            const int r = rb + lane;
            int s_row = 0, cnt = 0;
            if (r < nrows) {
                const int rz = r / ysp, ry = r - rz * ysp;
                const int cy = ry0 + ry, cz = rz0 + rz;
                // distance from the (clamped) box to the row's (y,z) column of cells; boundary rows are open
                const double ylo = (cy == 0) ? -INF : g.ymin + cy * g.cell, yhi = (cy == g.ny - 1) ? INF : g.ymin + (cy + 1) * g.cell;
                const double zlo = (cz == 0) ? -INF : g.zmin + cz * g.cell, zhi = (cz == g.nz - 1) ? INF : g.zmin + (cz + 1) * g.cell;
                const double dy = fmax(fmax(ylo - cy1, cy0 - yhi), 0.0) * 0.999999;      // (never over-estimated)
                const double dz = fmax(fmax(zlo - cz1, cz0 - zhi), 0.0) * 0.999999;
                const double rem = R2 - (dy * dy + dz * dz);
                if (rem >= 0.0) {
                    const double xr = sqrt(rem) * 1.000001;
                    const int x0 = cell_of_coord(cx0 - xr, g.xmin, g.inv_cell, g.nx - 1);
                    const int x1 = cell_of_coord(cx1 + xr, g.xmin, g.inv_cell, g.nx - 1);
                    const int row = (cz * g.ny + cy) * g.nx;
                    s_row = a.cell_start[row + x0];
                    cnt = a.cell_start[row + x1 + 1] - s_row;
                }
            }
            const int incl = wave_scan_incl(cnt);
            const int off = incl - cnt;
            const int Tc = __builtin_amdgcn_readlane(incl, 63);
            if (Tc > KG_FLAGCAP || T + Tc > KG_TCAP) { group_ok = false; break; }
            const bool ne = cnt > 0;
            const u64 nem = __builtin_amdgcn_ballot_w64(ne);
            if (ne) {
                rbase[lanes_below(nem)] = s_row - off;          // slot t of this chunk is particle rbase[ordinal] + t
                rflag[off] = 1;
            }
            wave_sync();
            int carry = 0;
            for (int t0 = 0; t0 < Tc; t0 += 256) {            // four batches of 64 slots: 12 loads in flight per lane
                int pp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + 64 * u + lane;
                    const bool valid = t < Tc;
                    const int tt = valid ? t : 0;
                    const bool fl = rflag[tt] != 0;
                    const u64 M = __builtin_amdgcn_ballot_w64(valid && fl);
                    const int ord = carry + lanes_below(M) - ((valid && fl) ? 0 : 1);
                    carry += __popcll(M);
                    pp[u] = valid ? rbase[ord > 0 ? ord : 0] + tt : -1;
                }
                double X[4], Y[4], Z[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = pp[u] >= 0 ? pp[u] : 0;
                    X[u] = a.x[q]; Y[u] = a.y[q]; Z[u] = a.z[q];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (pp[u] >= 0) {
                        const int o = KG_OFF(T + t0 + 64 * u + lane);
                        txy[o] = (float)((X[u] - Ox) * g.inv_cell);
                        txy[o + 2] = (float)((Y[u] - Oy) * g.inv_cell);
                        tzi[o] = (float)((Z[u] - Oz) * g.inv_cell);
                        reinterpret_cast<int*>(tzi)[o + 2] = pp[u];
                    }
                }
            }
            wave_sync();
            if (ne) rflag[off] = 0;
            wave_sync();
            T += Tc;
        }
    }
    if (!group_ok) {
        if (is_query && !fail) why = 2;
        fail = is_query;
        ok = false;
    }
    const int nw = (T + 31) >> 5;
    if (group_ok) {
        // pad the tile to whole mask words with records nobody accepts
        if (lane < nw * 32 - T) {
            const int o = KG_OFF(T + lane);
            txy[o] = 1e30f; txy[o + 2] = 1e30f; tzi[o] = 1e30f;
            reinterpret_cast<int*>(tzi)[o + 2] = -1;
        }
        wave_sync();
    }

    KG_STAMP(1)
    // ---- phase A: every lane against the whole tile -----------------------------------------------
    // accept d2 <= Rc^2 (1 + 2e-4): with err(d2) <= 1e-4 Rc^2 (checked below) every candidate truly inside R_i is
    // accepted, and every accepted one lies within R_i (1 + 1.6e-4) < the radius the tile covers
    const double Rc = R * g.inv_cell;
    const float fqx = (float)((qx - Ox) * g.inv_cell), fqy = (float)((qy - Oy) * g.inv_cell), fqz = (float)((qz - Oz) * g.inv_cell);
    // err(d2)/Rc^2 <= eps (6.93 E/Rc + 7.5); doubled for safety
    const double tol_rel = ok ? 2.0 * 5.9604644775390625e-08 * (6.93 * E / Rc + 7.5) : 0.0;
    if (ok && !(tol_rel <= 1e-4)) { ok = false; fail = true; why = 3; }
    const float r2f = ok ? (float)(Rc * Rc * 1.0002) : -1.0f;
    int cnt = 0;
    if (group_ok) {
        const f32x2 qx2 = {fqx, fqx}, qy2 = {fqy, fqy}, qz2 = {fqz, fqz};
        for (int w = 0; w < nw; ++w) {
            // one mask word = 16 pairs: all 32 LDS reads are issued before the arithmetic (one latency per word)
            f32x4 A[16];
            f32x2 Z[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                A[j] = tile_xy[w * 16 + j];
                Z[j] = *reinterpret_cast<const f32x2*>(&tile_zi[w * 16 + j]);
            }
            u32 m = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const f32x2 dx = A[j].xy - qx2, dy = A[j].zw - qy2, dz = Z[j] - qz2;
                f32x2 d2 = dx * dx;
                d2 = __builtin_elementwise_fma(dy, dy, d2);
                d2 = __builtin_elementwise_fma(dz, dz, d2);
                m = shift_in_le(m, d2.x, r2f);
                m = shift_in_le(m, d2.y, r2f);
            }
            // the word's set bits -> the lane's slot list (candidates entered at bit 0 and moved up: first = highest)
            while (__builtin_amdgcn_ballot_w64(m != 0u)) {
                if (m != 0u) {
                    const int j = __clz((int)m);
                    m &= ~(0x80000000u >> j);
                    if (cnt < 64) slist[cnt * 64 + lane] = (unsigned short)(w * 32 + j);
                    ++cnt;
                }
            }
        }
    }
    if (ok && (cnt > 64 || cnt < K)) { ok = false; fail = true; why = cnt > 64 ? 4 : 5; }
    const int maxcnt = (int)wmax((double)(ok ? cnt : 0));

    KG_STAMP(2)
    // ---- phase B: set bits -> keys in registers (slot-major; each lane walks its own mask words) ----
    u32 key[64];
    const float kscale = ok ? (float)(2097152.0 / (Rc * Rc * 1.0002)) : 0.0f;
    wave_sync();
#pragma unroll
    for (int c = 0; c < 4; ++c) {                       // 16 slots at a time: list reads, then tile reads, then arithmetic
        if (c * 16 < maxcnt) {
            int tt[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) tt[u] = slist[(c * 16 + u) * 64 + lane];
            float cx[16], cy[16], cz[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int o = KG_OFF(tt[u] & 2047);
                cx[u] = txy[o]; cy[u] = txy[o + 2]; cz[u] = tzi[o];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float dx = cx[u] - fqx, dy = cy[u] - fqy, dz = cz[u] - fqz;
                float d2 = dx * dx;
                d2 = fmaf(dy, dy, d2);
                d2 = fmaf(dz, dz, d2);
                u32 qd = (u32)(d2 * kscale);
                qd = qd > 2097151u ? 2097151u : qd;
                key[c * 16 + u] = (ok && c * 16 + u < cnt) ? ((qd << 11) | (u32)tt[u]) : 0xFFFFFFFFu;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) key[c * 16 + u] = 0xFFFFFFFFu;
        }
    }
    KG_STAMP(3)
    // ---- order the 64 registers ----
    OemP<1>::run(key);

    KG_STAMP(4)
    // ---- certify: consecutive keys among the first K+1 further apart than twice the error bound ----
    {
        // bins of 2^-21 R^2(1.0002); each key within tol of the truth, +1 bin for the floor of the quantisation
        const u32 win = ok ? (u32)(2.0 * tol_rel * 2097152.0) + 3u : 0u;
        bool amb = false;
#pragma unroll
        for (int r = 0; r < 63; ++r)
            if (r < K && key[r + 1] != 0xFFFFFFFFu && (key[r + 1] >> 11) - (key[r] >> 11) <= win) amb = true;
        if (ok && amb) { ok = false; fail = true; why = 6; }
    }

    KG_STAMP(5)
    // ---- outputs ----
    {
        int lastidx = -1;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (r < K) {
                int idx = -1;
                if (ok) idx = reinterpret_cast<const int*>(tzi)[KG_OFF(key[r] & 2047u) + 2];
                if (r == K - 1) lastidx = idx;
                // failed queries are written by the list-mode launch that follows; non-queries (ghosts, padding) get -1
                if (p < a.npad && (ok || !is_query)) a.nbr[(size_t)r * a.npad + p] = idx;
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
    hipLaunchKernelGGL(knn_group_kernel, dim3(blocks), dim3(64), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    if (prof) {
        u64 h[8];
        HIPCHK(hipMemcpyAsync(h, a.prof, 64, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const double w = (double)blocks;
        fprintf(stderr, "[sphx] grouped search cycles/wave: setup %.0f stage %.0f phaseA %.0f phaseB %.0f sort %.0f certify %.0f output %.0f tail %.0f\n",
                h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, h[7] / w);
    }
    return SPHX_OK;
}
