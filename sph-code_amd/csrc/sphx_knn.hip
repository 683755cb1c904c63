// sphx_knn.hip - exact k-nearest-neighbour search on a cell list (replaces nsc:541-552).
//
// One 64-wide wavefront per query particle; K <= 64 so the running top-64 candidate set lives
// in registers, ONE ENTRY PER LANE, kept sorted by a bitonic network on (d^2 bits, index).
//   * particles are cell-sorted (sphx_grid.hip); a row of cells cx0..cx1 at fixed (cy,cz) is
//     one contiguous range of the sorted arrays, so candidate loads are coalesced SoA reads;
//   * the rows of the search cube are flattened with a wave prefix sum so every batch of 64
//     candidates uses all 64 lanes;
//   * candidates below the current threshold (search radius, then the running K-th best) are
//     ballot-compacted into a per-wave LDS staging ring; only when 64 survivors have
//     accumulated does the wave pay for a 64-key bitonic sort + merge;
//   * the search radius is per particle (previous h, or a cell-count density estimate); if
//     fewer than K candidates lie inside it the same wave enlarges the radius and repeats, so
//     the result is always the exact kNN (ties broken by position in the deterministic cell order).
// Distances are accumulated as ((dx*dx + dy*dy) + dz*dz) without FMA contraction, the same
// arithmetic SciPy's cKDTree uses, so orderings agree with the oracle bit for bit.
#include "sphx_internal.h"

#define KNN_BLOCK 256
#define KNN_PPB 64          // particles per workgroup (16 per wave)
#ifndef KNN_LIST_BLOCKS
#define KNN_LIST_BLOCKS 512     // list mode: workgroups walking the list, at least (measured at 1e6 particles, 1.5 % of them
#endif                          // listed: 256: 0.579 ms search, 384: 0.561, 512: 0.555, 1024: 0.575, 2048: 0.617, 4096: 0.70;
#ifndef KNN_LIST_BLOCKS_MAX     // 7 % listed - the uniform cube's faces: 512: 1.12, 768: 0.99, 1024: 0.93, 2048: 0.92) - the
#define KNN_LIST_BLOCKS_MAX 2048 // launcher sizes the grid from the previous search's list: one workgroup per 32 entries
#endif
#ifndef KNN_LIST_PPB
#define KNN_LIST_PPB 16     // list mode: queries per workgroup and pass
#endif
#define KNN_MAX_TRIES 48
#define KNN_FLAG_CAP 1024      // candidate slots per row chunk served by the flag lookup (else binary search)

struct KnnArgs {
    int n, k, npad;
    int n_active;              // queries with id >= n_active (ghosts) are skipped
    const double *x, *y, *z;
    const int* id;
    const int* inv;
    const int* qorder;         // processing order: query slot p works on stored particle qorder[p] (nullable)
    const int* cell_start;
    const double* tbox;        // true bounding box {min xyz, max xyz}
    GridParams g;
    const double* rsearch;
    int hint_by_id;            // rsearch is indexed by id (caller order) instead of sorted order
    double rscale;
    double rbound;
    int* nbr;
    int* list64;               // [n][64] the full sorted candidate set kept per particle (nullable)
    double* dref;              // [n] every particle NOT in list64[i] was farther than dref[i] (nullable)
    double* h_sorted;
    long long* idx64;
    double* dist;
    long long* nontriv;
    double* h_by_id;
    u64* counters;
    // list mode (LIST = 1): the queries are the processing slots qlist[0 .. *qcount) - the ones the grouped
    // search (sphx_knn_group.hip) could not certify; the grid is fixed and walks the list
    const int* qlist;
    const int* qcount;
    OutLevels ol;              // OUTL = 1: the outlier levels of the current grid (ol.L >= 1)
    int distrust;              // hints are upper bounds at best: every radius is seeded from the local cell counts
    // list mode: the launch's blocks beyond list_blocks order the near ties the grouped search left (sphx_knn_group.hip:
    // the entries' queries are not on the list) - beside the list's queries instead of in a launch of their own
    const int4* tie_list;
    const int* tie_count;
    int tie_cap, list_blocks;
};

#include "sphx_wave.h"
#include "sphx_knn_group.h"

// Sort the `cnt` staged entries at ring position `head` ascending by (key, index) into (ck, cv);
// lanes >= cnt get the (INF, ~0) padding.  r2 = trial radius^2 bounds every staged key.
template <int ABL>
__device__ __forceinline__ void sort_staged(const u64* skey, const u32* sid, int head, int cnt, double r2,
                                            int lane, u64& ck, u32& cv) {
    ck = KNN_INF;
    cv = 0xFFFFFFFFu;
    if (lane < cnt) {
        ck = skey[(head + lane) & 127];
        cv = sid[(head + lane) & 127];
    }
    if (ABL == 1) return;              // timing experiment: no ordering network
    // degenerate radius (0 or overflowed): every key lands in bin 0 and the exact odd-even
    // fix-up below does all the ordering (slow, correct, practically never taken)
    // (any positive scale orders correctly - ties are repaired below - so the hardware reciprocal will do)
    const double qscale = (r2 > 0.0 && r2 < 1e300) ? 67108862.0 * __builtin_amdgcn_rcp(r2) : 0.0;
    const double qd = fmin(__longlong_as_double((long long)ck) * qscale, 67108862.0);
    u32 k32 = (lane < cnt) ? (((u32)qd << 6) | (u32)lane) : 0xFFFFFFFFu;
    sort32_sizes<64>(k32, lane);
    const bool v = (k32 != 0xFFFFFFFFu);
    const int slot = (head + (int)(k32 & 63u)) & 127;
    ck = v ? skey[slot] : KNN_INF;
    cv = v ? sid[slot] : 0xFFFFFFFFu;
    const u32 nextq = (u32)__shfl_down((int)(k32 >> 6), 1, 64);
    if (__builtin_amdgcn_ballot_w64(v && lane < 63 && nextq == (k32 >> 6))) {
        // same-bin neighbours: odd-even transposition on the exact keys until ordered
        for (int it = 0; it < 64; ++it) {
            u64 k0 = ck; u32 v0 = cv;
            cmpx<1>(ck, cv, (lane & 1) == 0);                          // pairs (0,1)(2,3)...
            const int partner = (lane & 1) ? lane + 1 : lane - 1;     // pairs (1,2)(3,4)...
            const int pc = partner < 0 ? 0 : (partner > 63 ? 63 : partner);
            const u64 pk = __shfl(ck, pc, 64);
            const u32 pv = __shfl(cv, pc, 64);
            if (partner >= 0 && partner <= 63) {
                const bool p_lt = kv_less(pk, pv, ck, cv);
                const bool keep_min = (lane & 1) != 0;                  // the odd lane is the lower of its pair
                if (p_lt == keep_min) { ck = pk; cv = pv; }
            }
            if (!__builtin_amdgcn_ballot_w64(k0 != ck || v0 != cv)) break;
        }
    }
}

// merge an ascending `cand` into the ascending `best` (both 64 wide): 64 smallest, ascending
__device__ __forceinline__ void merge_sorted(u64& bk, u32& bv, u64 ck, u32 cv, int lane) {
    const u64 rk = __shfl(ck, 63 - lane, 64);
    const u32 rv = __shfl(cv, 63 - lane, 64);
    if (kv_less(rk, rv, bk, bv)) { bk = rk; bv = rv; }   // bitonic: min of asc and desc
    merge_stage<32>(bk, bv, lane);
}

__device__ __forceinline__ int cell_coord(double v, double vmin, double inv_cell, int nmax1) {
    double t = (v - vmin) * inv_cell;
    t = fmin(fmax(t, 0.0), (double)nmax1);
    return (int)t;
}

// particles and cells in the (clipped) 3x3x3 block of cells around (xi, yi, zi): lanes 0..8 read one x-run each
__device__ __forceinline__ void block27_count(const GridParams& g, const int* cell_start, double xi, double yi, double zi,
                                              int lane, int& cnt_out, int& nc_out) {
    const int cxi = cell_coord(xi, g.xmin, g.inv_cell, g.nx - 1);
    const int cyi = cell_coord(yi, g.ymin, g.inv_cell, g.ny - 1);
    const int czi = cell_coord(zi, g.zmin, g.inv_cell, g.nz - 1);
    int cnt = 0, nc = 0;
    if (lane < 9) {
        int cy = cyi - 1 + lane % 3, cz = czi - 1 + lane / 3;
        if (cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz) {
            int xlo = max(cxi - 1, 0), xhi = min(cxi + 1, g.nx - 1);
            int row = (cz * g.ny + cy) * g.nx;
            cnt = cell_start[row + xhi + 1] - cell_start[row + xlo];
            nc = xhi - xlo + 1;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64);
        nc += __shfl_xor(nc, o, 64);
    }
    cnt_out = __shfl(cnt, 0, 64);
    nc_out = __shfl(nc, 0, 64);
}

#ifndef KNN_MIN_WAVES
#define KNN_MIN_WAVES 6      // waves per SIMD the register budget is held to (6 -> <= 80 VGPRs; measured fastest)
#endif
// ABL != 0: timing experiments with a section removed (outputs are then meaningless and are
// never written: the wrapper passes null output pointers); ABL == 0 is the product kernel.
// LEAN = 1: the step loop's variant - only the K-major list and h (sorted order) are produced,
// so the API / Verlet-list pointers are never loaded (17 pointers in SGPRs otherwise: measured
// 20 % slower).  LEAN = 2: the same for the device API (h written by id).
// OUTL = 1 (list mode only): a query OUTSIDE the grid box whose search sphere is wider than OLEV_MIN_RC cells does not walk
// the grid's boundary faces (every escaper hashed there: whole faces of one-particle rows) but the outlier levels its
// sphere can reach (OutLevels) and then the grid with the outliers filtered out - the same candidates, each exactly once.
#ifndef SPHX_KNN_PROF_LONG
#define SPHX_KNN_PROF_LONG 10000000ull     // -DSPHX_KNN_PROF builds: a query taking more cycles than this is described
#endif
#ifndef KNN_LIST_WAVES
#define KNN_LIST_WAVES 4     // list / outlier-level variants: fewer, fatter waves (the prefetched batch needs registers)
#endif
#ifndef KNN_OUTL_WAVES
#define KNN_OUTL_WAVES 3     // the variants that also walk the outlier levels: 168 VGPRs instead of 128 + 130..170 B of scratch (search -1..2 % on the cube and the blast)
#endif
template <int ABL, int LEAN = 0, int LIST = 0, int OUTL = 0>
__global__ __launch_bounds__(KNN_BLOCK, OUTL ? KNN_OUTL_WAVES : LIST ? KNN_LIST_WAVES : KNN_MIN_WAVES) void knn_kernel(KnnArgs a) {
    extern __shared__ int tile_dyn[];                 // [K][KNN_PPB + 1] result tile (sized at launch)
#define tile(kk, li) tile_dyn[(kk) * (KNN_PPB + 1) + (li)]
    __shared__ u64 stg_key[KNN_BLOCK / 64][128];
    __shared__ u32 stg_id[KNN_BLOCK / 64][128];
    // candidate-slot -> row lookup: one start flag per candidate slot + the compacted row bases
    __shared__ unsigned char row_flag[KNN_BLOCK / 64][KNN_FLAG_CAP];
    __shared__ int row_base[KNN_BLOCK / 64][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = (LIST && a.tie_list) ? a.list_blocks : (int)gridDim.x;      // blocks that walk queries
    if (LIST && a.tie_list && (int)blockIdx.x >= nblk) {
        // order = (exact fp64 d^2, storage index), kv_less below; entry {query slot p, rank r, candidates a (rank r), b (r + 1)}
        const int nt = min(*a.tie_count, a.tie_cap);
        for (int e = ((int)blockIdx.x - nblk) * KNN_BLOCK + (int)threadIdx.x; e < nt; e += ((int)gridDim.x - nblk) * KNN_BLOCK) {
            const int4 t = a.tie_list[e];
            const int p = t.x, r = t.y, ia = t.z, ib = t.w;
            const int qs_ = a.qorder ? a.qorder[p] : p;
            const double qx_ = a.x[qs_], qy_ = a.y[qs_], qz_ = a.z[qs_];
            const double da = dist2_nofma(a.x[ia] - qx_, a.y[ia] - qy_, a.z[ia] - qz_);
            const double db = dist2_nofma(a.x[ib] - qx_, a.y[ib] - qy_, a.z[ib] - qz_);
            if (db < da || (db == da && ib < ia)) {
                a.nbr[(size_t)r * a.npad + p] = ib;
                if (r + 1 < a.k) a.nbr[(size_t)(r + 1) * a.npad + p] = ia;
                if (r >= a.k - 2) {          // rank K - 1 changed hands: r = K - 1, b was the unlisted (K+1)-th; r = K - 2, the K-th is now a
                    const double hv = sqrt(r == a.k - 1 ? db : da);
                    if (LEAN == 2) a.h_by_id[a.id[qs_]] = hv; else a.h_sorted[qs_] = hv;
                }
            }
        }
        return;
    }
    const int total = LIST ? *a.qcount : a.n;          // queries: list entries, or all particles
    int vblock = LIST ? (int)blockIdx.x : xcd_block(blockIdx.x, gridDim.x);
    const GridParams g = a.g;
    const int K = a.k;
    const int KT = (!LEAN && a.list64) ? 64 : K;
    u64* skey = stg_key[wave];
    u32* sid = stg_id[wave];
    unsigned char* rflag = row_flag[wave];
    int* rbase = row_base[wave];
    for (int q = lane; q < KNN_FLAG_CAP / 4; q += 64) reinterpret_cast<u32*>(rflag)[q] = 0u;
    u64 ncand = 0, nretry = 0, nshort = 0, nfar = 0, nbad = 0;
#ifdef SPHX_KNN_PROF
    u64 p_sum[2] = {0, 0}, p_n[2] = {0, 0}, p_max[2] = {0, 0}, p_tries[2] = {0, 0};
    u64 p_sec[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // row set-up, candidates, ranking (cycles); batches of rows
#endif

    do {
    // list mode: 4 queries per wave and pass instead of 16 - the list is short (a few per cent of the queries), so
    // the launch is bound by how long one wave takes, not by how many waves there are
    constexpr int PPB = LIST ? KNN_LIST_PPB : KNN_PPB;
    const int base = vblock * PPB;
    // Which query is the workgroup's li-th (li = 4 wave + its turn)?  Normally the li-th of its block of consecutive ones.
    // In list mode the list is dealt out wave by wave instead - entry r NW + w goes to wave w of all NW as its r-th -
    // because the list's heavy entries come in runs (the 64 queries of an overflowed group, a pile of escapers in one
    // boundary cell): consecutive entries must not queue up behind each other in one wave.
    const int list_pass = LIST ? (vblock - (int)blockIdx.x) / nblk : 0;
    const int NWV = nblk * (KNN_BLOCK / 64);
    auto entry_of = [&](int li) -> int {
        if (!LIST) return base + li;
        const int wv = li / (PPB / 4), turn = li % (PPB / 4);
        return (list_pass * (PPB / 4) + turn) * NWV + (int)blockIdx.x * (KNN_BLOCK / 64) + wv;
    };
    if (LIST && entry_of(0) >= total) break;       // (the workgroup's smallest entry of this pass and of all later ones)
    // the wave's 16 query particles are fetched in ONE coalesced round trip (lane l holds
    // particle l) and broadcast through SGPRs as each comes up
    double qx = 0.0, qy = 0.0, qz = 0.0, qr = 0.0;
    int qid = 0x7FFFFFFF;
    int qs = 0;                      // where the lane's query particle is stored
    {
        const int ip = entry_of(wave * (PPB / 4) + (lane & 15));
        if (lane < PPB / 4 && ip < total) {
            // (a list entry: processing slot in the low 29 bits; in the top three, how much wider than the hint to start -
            //  the grouped kernel counted fewer than K inside it: 1 .. 7 = x1.6 .. x3, sphx_knn_group.hip)
            const unsigned raw = LIST ? (unsigned)a.qlist[ip] : (unsigned)ip;
            const int slot = (int)(raw & 0x1FFFFFFFu);
            const unsigned gc = LIST ? raw >> 29 : 0u;
            qs = a.qorder ? a.qorder[slot] : slot;
            qx = a.x[qs]; qy = a.y[qs]; qz = a.z[qs];
            qid = a.id[qs];
            qr = a.rsearch ? a.rsearch[a.hint_by_id ? qid : qs] * a.rscale : 0.0;
            if (gc) qr *= 1.6 + (double)(gc - 1u) * (1.4 / 6.0);
        }
    }

    for (int t16 = 0; t16 < PPB / 4; ++t16) {
        const int li = wave * (PPB / 4) + t16;
        const int oid = __builtin_amdgcn_readlane(qid, t16);
        // wave-uniform: past the end, or a ghost (a candidate, never a query)
        if (entry_of(li) >= total || oid >= a.n_active) {
            if (lane < K) tile(lane, li) = -1;
            continue;
        }
        const double xi = bcast_f64(qx, t16), yi = bcast_f64(qy, t16), zi = bcast_f64(qz, t16);
        double R = bcast_f64(qr, t16);
        if (!(R > 0.0)) {
            // density estimate from the 3x3x3 block of cells around the particle
            int cnt, nc;
            block27_count(g, a.cell_start, xi, yi, zi, lane, cnt, nc);
            double vol = (double)nc * g.cell * g.cell * g.cell;
            double dens = (double)(cnt > 0 ? cnt : 1) / vol;
            R = 1.3 * cbrt((double)K / (4.1887902047863905 * dens));
        }
        if (R > a.rbound) R = a.rbound;
        // list mode: is the query outside the grid box (wave-uniform)?  Counted when its sphere is wide (SC_FARQ: the host
        // builds the outlier levels for the next search when there are many), served by the levels when OUTL
        const bool q_out = (LIST || OUTL || a.distrust) && sphx_outside_box(g, xi, yi, zi);
        const double R_given = R;
        bool far_counted = false;
        // which outlier levels hold anything at all (one round trip per far query: lane l asks for level l + 1): an escaper's
        // sphere reaches most of the 18-20 levels, nearly all of them empty - 1024 rows each to test, per rung of its ladder.
        // (Measured and dropped: with few outliers in all, ONE pass with an unbounded radius instead of the ladder - fewer
        // retries, the same time: the ladder is not what a far query's 0.5 ms are made of.)
        u64 occm = ~0ull;
        if (OUTL && q_out) {
            int c_lv = 0;
            if (lane < a.ol.L) {
                const int* stl = a.ol.start + (size_t)lane * (OLEV_N * OLEV_N * OLEV_N);
                c_lv = stl[OLEV_N * OLEV_N * OLEV_N] - stl[0];
            }
            occm = __builtin_amdgcn_ballot_w64(c_lv > 0);
        }
#ifdef SPHX_KNN_PROF
        const long long prof_t0 = clock64();
        const u64 prof_c0 = ncand;
        int prof_cat = q_out ? 3 : 0;
        u64 q_sec[4] = {0, 0, 0, 0};
#endif
        // A stale hint: the particle has moved (a diverging run moves it by several h per step) into a neighbourhood far
        // denser than its previous radius implies - the 3x3x3 block of cells around it alone holds many times what a sphere
        // of that radius should.  The hinted sphere would then cover thousands of times the candidates needed (the whole
        // cloud for an escaper flying back through the core): start from the block's density estimate instead and climb
        // towards the hint only if that comes up short.
        double R_hint = 0.0;              // > 0: the ladder climbs back to it before it goes beyond
        if ((LIST || a.distrust) && !q_out) {
            const double rc = R * g.inv_cell;
            if (rc > 3.0 || a.distrust) {
                int cnt, nc;
                block27_count(g, a.cell_start, xi, yi, zi, lane, cnt, nc);
                if (cnt < 1) cnt = 1;
                const double expect = (double)K * (double)nc / (4.1887902047863905 * rc * rc * rc);
                if (a.distrust || (double)cnt > 4.0 * expect + 8.0) {
                    const double dens = (double)cnt / ((double)nc * g.cell * g.cell * g.cell);
                    const double Re = 1.3 * cbrt((double)K / (4.1887902047863905 * dens));
                    if (Re < R) { R_hint = R; R = Re; }
#ifdef SPHX_KNN_PROF
                    if (R_hint > 0.0) prof_cat = 1;
#endif
                }
            }
        }

        u64 bk;
        u32 bv;
        int tries = 0;
        bool done, saw_all = false;
        do {
            bk = KNN_INF;
            bv = 0xFFFFFFFFu;
            // threshold: accept key < tk.  Inside the trial radius R the test is d2 <= R^2; at the
            // caller's bound it is the strict d < dist of cKDTree; later the running K-th best, ties
            // with it included (a superset: the merge orders by (key, index) and drops the excess).
            const bool at_bound = (R >= a.rbound);
            u64 tk = (u64)__double_as_longlong(R * R) + (at_bound ? 0ull : 1ull);
            const double R2 = R * R;
            int nst = 0, head = 0;            // staging ring occupancy / head (wave-uniform)
            bool have_best = false;           // best[] still empty: first flush is a plain sort
            // ---- which index structures this try walks: the grid alone, or (OUTL, far query) levels lv_hi .. lv_lo of
            // the outliers and then the grid with the outliers filtered out
            bool multi = false;
            int lv_lo = 0, lv_hi = 0, lv_own = 0;
            double Rcur = R;                  // radius still to be searched: shrinks to the K-th best once K are in hand
            if (q_out && (float)(R * g.inv_cell) > OLEV_MIN_RC) {
                if (!far_counted) { ++nfar; far_counted = true; }
                if (OUTL) {
                    // Chebyshev distance from the centre, in units of hmax, of the nearest / farthest point of the sphere
                    // (|(|p|_inf - |q|_inf)| <= |p - q|_2), padded by 1e-9 against the rounding of the level build;
                    // level 1 holds m < 2, level l holds [2^(l-1), 2^l), the top level everything from 2^(L-1) on
                    const double mq = fmax(fmax(fabs(xi - a.ol.cx), fabs(yi - a.ol.cy)), fabs(zi - a.ol.cz));
                    const double mlo = fmax(mq - R, 0.0) * (1.0 - 1e-9) / a.ol.hmax, mhi = (mq + R) * (1.0 + 1e-9) / a.ol.hmax;
                    lv_lo = 1;
                    while (lv_lo < a.ol.L && ldexp(1.0, lv_lo) <= mlo) ++lv_lo;          // levels below end before mlo
                    lv_hi = lv_lo;
                    while (lv_hi < a.ol.L && ldexp(1.0, lv_hi) <= mhi) ++lv_hi;          // levels above start beyond mhi
                    // the query's own level goes first: its neighbours are most likely there, and with K of them in hand
                    // the other structures are searched only out to the K-th (a stale, far too wide hint - an escaper
                    // flying back past the core - then costs its own level, not the whole cloud)
                    int e = 0;
                    (void)frexp(mq / a.ol.hmax, &e);
                    lv_own = e < lv_lo ? lv_lo : (e > lv_hi ? lv_hi : e);
                    multi = true;
#ifdef SPHX_KNN_PROF
                    prof_cat = 2;
#endif
                }
            }
            bool own_done = false;
            for (int lv = multi ? lv_own : 0;;) {
            // Query position and radius in CELL units of this structure: fp64 once, then all range geometry in fp32.
            // Every fp32 quantity is padded (radius x(1+1e-5) + 2e-3 cells, distances - 1e-3 cells;
            // fp32 resolves 1.2e-4 cells at index 2047), so the clipped ranges can only grow: a
            // superset of the exact fp64 ranges, never a lost neighbour.
            double ox = g.xmin, oy = g.ymin, oz = g.zmin, icell = g.inv_cell;
            float nx1 = g.fnx1, ny1 = g.fny1, nz1 = g.fnz1;
            int gnx = g.nx, gny = g.ny;
            const int* cstart = a.cell_start;
            const bool indirect = OUTL && lv > 0;         // candidate slots index the outlier list
            const bool filter = OUTL && multi && lv == 0; // grid pass of a far query: outliers were seen in their levels
            if (indirect) {
                const double W = ldexp(a.ol.hmax, lv);
                ox = a.ol.cx - W; oy = a.ol.cy - W; oz = a.ol.cz - W;
                icell = (double)OLEV_N / (2.0 * W);
                nx1 = ny1 = nz1 = (float)(OLEV_N - 1);
                gnx = gny = OLEV_N;
                cstart = a.ol.start + (size_t)(lv - 1) * (OLEV_N * OLEV_N * OLEV_N);
            }
            const float fx = (float)((xi - ox) * icell);
            const float fy = (float)((yi - oy) * icell);
            const float fz = (float)((zi - oz) * icell);
            const float Rc = (float)(Rcur * icell) * 1.00001f + 2e-3f;
            // rows are tested against the query clamped into the grid: with it the half-infinite
            // boundary cells need no special case (their open side can only lie behind the query).  A structure of
            // finite extent (the grid with the outliers filtered out, every level but the top one) holds nothing
            // beyond its cube: there the true distances apply, and a sphere that misses the cube walks no rows
            const bool finite = OUTL && multi && (lv == 0 || lv < a.ol.L);
            const float fyc = finite ? fy : fminf(fmaxf(fy, 0.0f), ny1 + 1.0f), fzc = finite ? fz : fminf(fmaxf(fz, 0.0f), nz1 + 1.0f);
            const int cy0 = (int)fminf(fmaxf(fy - Rc, 0.0f), ny1);
            const int cy1 = (int)fminf(fmaxf(fy + Rc, 0.0f), ny1);
            const int cz0 = (int)fminf(fmaxf(fz - Rc, 0.0f), nz1);
            const int cz1 = (int)fminf(fmaxf(fz + Rc, 0.0f), nz1);
            const int ysp = cy1 - cy0 + 1;
            int nrows = __mul24(ysp, cz1 - cz0 + 1);
            if (indirect && !((occm >> (lv - 1)) & 1ull)) nrows = 0;      // an empty level
            if (finite) {
                const float ex = fmaxf(fmaxf(-fx, fx - (nx1 + 1.0f)), 0.0f), ey = fmaxf(fmaxf(-fy, fy - (ny1 + 1.0f)), 0.0f),
                            ez = fmaxf(fmaxf(-fz, fz - (nz1 + 1.0f)), 0.0f);
                // (relative padding: the coordinates of a far query are large numbers of cells)
                if ((ex * ex + ey * ey + ez * ez) * 0.9999f > Rc * Rc) nrows = 0;
            }
            // The axis along which the query lies farther OUTSIDE the structure runs outermost (list / level variants): the
            // sphere of a far query cuts a wide, shallow cap out of the cloud's near side - every layer of the other axis, only
            // the first few rows of this one - and with this axis outermost those rows come in the first batches; with it
            // innermost they are the first few rows of every layer, one batch - one round trip - per layer (the Sedov blast's
            // escaper, level with the cloud in z, outside it in y: 284 batches of rows, 5.5e5 of its 8.8e5 cycles in row set-up).
            const float outy = fmaxf(fmaxf(-fy, fy - (ny1 + 1.0f)), 0.0f), outz = fmaxf(fmaxf(-fz, fz - (nz1 + 1.0f)), 0.0f);
            const bool swap_yz = (LIST || OUTL) && outy > outz;
            const int isp = swap_yz ? cz1 - cz0 + 1 : ysp;           // extent of the INNER axis
            const float inv_ysp = __builtin_amdgcn_rcpf((float)isp);
            // rows are visited CENTRE-OUT from the row of the (clamped) query, in z (outer) and in y (inner): j = 0, 1, 2 ...
            // -> cq, cq + 1, cq - 1, cq + 2 ..., then on along the longer side
            const int cyq = min(max((int)fyc, cy0), cy1), czq = min(max((int)fzc, cz0), cz1);
            const int my = min(cyq - cy0, cy1 - cyq), mz = min(czq - cz0, cz1 - czq);
            const bool up_y = cy1 - cyq > cyq - cy0, up_z = cz1 - czq > czq - cz0;
            float Rc2 = Rc * Rc;             // shrinks with the running K-th best (rows and chords still to come are clipped to it)
            // A query FAR outside the structure (an escaper at the drv:233 clamp, 1e5 cloud radii = 1e7 cells away) and its
            // sphere are numbers of cells fp32 cannot resolve: the relative padding above (1e-5 of 1e7 cells = 100 cells: the
            // whole cloud) made every chord the whole row - also after K candidates were in hand and the chords should have
            // shrunk to the thin cap nearer than the K-th best - so ONE such query read all 1e6 particles, 64 per round trip:
            // 10 ms, the whole launch (Sedov blast under hydro_update's sums, from the step its first particle escaped).  Its
            // range geometry is done in fp64 instead (list / level variants only; wave-uniform; padded by 1e-9 relative).
            const bool farq = (LIST || OUTL) && (fabsf(fx) > 3e4f || fabsf(fy) > 3e4f || fabsf(fz) > 3e4f || Rc > 3e4f);
            // (its fp64 quantities are recomputed per batch of rows from what is live anyway - held across the candidate
            //  loop they cost every query of the list-mode launch 40 B of scratch per lane: list mode +10..20 %)

            for (int rb = 0; rb < (ABL == 3 ? 0 : nrows); rb += 64) {
#ifdef SPHX_KNN_PROF
                const long long ts0 = clock64();
                u64 d_iter = 0;
#endif
                // ---- one lane per (cy,cz) row of cells: clip the row to the search SPHERE ----
                const int r = rb + lane;
                int s_row = 0, cnt = 0;
                // (read with every lane active: the shuffle below sits outside the divergent row set-up)
                u64 kth_far = KNN_INF;
                if ((LIST || OUTL) && farq && have_best) kth_far = __shfl(bk, KT - 1, 64);
                if (r < nrows) {
                    // r / ysp: fp32 estimate (r < 2^24 rows, quotient <= 4096: off by one at most) + fix-up
                    int rz = (int)(((float)r + 0.5f) * inv_ysp);       // (outer index, inner index) = (rz, ry) ...
                    int ry = r - __mul24(rz, isp);
                    if (ry < 0) { --rz; ry += isp; }
                    if (ry >= isp) { ++rz; ry -= isp; }
                    if ((LIST || OUTL) && swap_yz) { const int t_ = ry; ry = rz; rz = t_; }      // ... or (ry, rz)
                    // The rows nearest to the query first: with K candidates in hand the rest is clipped to the K-th best, and
                    // the nearer the first K, the less is left.  (Until round 3 the sweep ran from the nearer END of the
                    // range: a query outside the box in y but level with the cloud in z - the Sedov blast's first escaper,
                    // four box widths out - then met the cloud's far cap first and its candidates in DECREASING distance:
                    // every one passed the running threshold, 1.03e6 candidates and 2.4e7 cycles for one query.)
                    const int offy = (ry + 1) >> 1, offz = (rz + 1) >> 1;
                    const int cy = ry <= 2 * my ? ((ry & 1) ? cyq + offy : cyq - offy) : (up_y ? cyq + (ry - my) : cyq - (ry - my));
                    const int cz = rz <= 2 * mz ? ((rz & 1) ? czq + offz : czq - offz) : (up_z ? czq + (rz - mz) : czq - (rz - mz));
                    // distance (in cells) from the query to the row's (y,z) cell column.  Boundary cells
                    // are half-infinite: out-of-box coordinates are clamped into them (sphx_grid.hip).
                    const float cyf = (float)cy, czf = (float)cz;
                    bool meets;
                    int rx0, rx1;
                    if ((LIST || OUTL) && farq) {
                        // Only the BOUNDARY rows are half-infinite (out-of-box particles are clamped into them): they are tested
                        // against the query clamped into the grid.  An interior row holds particles whose y and z truly lie in
                        // it, so the true - far - query applies: with the clamped one a query off in two or three axes (a corner
                        // of the clamp cube) kept every row's whole chord however close its K-th best already was.
                        double icd = icell;
                        asm volatile("" : "+v"(icd));          // keep this block's values out of the loop's preheader
                        const double qxd = (xi - ox) * icd, qyd = (yi - oy) * icd, qzd = (zi - oz) * icd;
                        const double qycd = finite ? qyd : fmin(fmax(qyd, 0.0), (double)ny1 + 1.0),
                                     qzcd = finite ? qzd : fmin(fmax(qzd, 0.0), (double)nz1 + 1.0);
                        double Rc2d = (Rcur * icd) * (1.0 + 1e-9) + 1e-6;
                        Rc2d *= Rc2d;
                        if (kth_far != KNN_INF) {               // the running K-th best (it only ever shrinks)
                            const double rp = sqrt(__longlong_as_double((long long)kth_far)) * icd * (1.0 + 1e-9) + 1e-6;
                            Rc2d = fmin(Rc2d, rp * rp);
                        }
                        const double qye = (cy == 0 || cy == (int)ny1) ? qycd : qyd, qze = (cz == 0 || cz == (int)nz1) ? qzcd : qzd;
                        const double dyd = fmax(fmax((double)cy - qye, qye - ((double)cy + 1.0)) - 1e-6, 0.0);
                        const double dzd = fmax(fmax((double)cz - qze, qze - ((double)cz + 1.0)) - 1e-6, 0.0);
                        const double remd = Rc2d - (dyd * dyd + dzd * dzd);
                        meets = remd >= 0.0;
                        const double hcd = sqrt(meets ? remd : 0.0) * (1.0 + 1e-9) + 1e-6;
                        rx0 = (int)fmin(fmax(qxd - hcd, 0.0), (double)nx1);
                        rx1 = (int)fmin(fmax(qxd + hcd, 0.0), (double)nx1);
                    } else {
                        // (boundary rows against the clamped query, interior rows against the true one - see the fp64 form above:
                        //  against the clamped query a sphere as wide as the distance to the cloud contains the whole grid, and
                        //  an escaper a few box widths out read all 1e6 particles however near its K-th best already was)
                        const float fye = (cy == 0 || cy == (int)ny1) ? fyc : fy, fze = (cz == 0 || cz == (int)nz1) ? fzc : fz;
                        const float dy = fmaxf(fmaxf(cyf - fye, fye - (cyf + 1.0f)) - 1e-3f, 0.0f);
                        const float dz = fmaxf(fmaxf(czf - fze, fze - (czf + 1.0f)) - 1e-3f, 0.0f);
                        const float rem = Rc2 - (dy * dy + dz * dz);
                        meets = rem >= 0.0f;
                        const float hc = __builtin_amdgcn_sqrtf(meets ? rem : 0.0f) * 1.00001f + 1e-3f;   // 1 ulp: inside the padding
                        rx0 = (int)fminf(fmaxf(fx - hc, 0.0f), nx1);
                        rx1 = (int)fminf(fmaxf(fx + hc, 0.0f), nx1);
                    }
                    if (meets) {        // the row meets the sphere: chord along x
                        // < 2^23 cells in all: 24-bit multiplies, 32-bit byte offsets
                        const int row = __mul24(__mul24(cz, gny) + cy, gnx);
                        const char* cs = (const char*)cstart;
                        s_row = *(const int*)(cs + ((u32)(row + rx0) << 2));
                        cnt = *(const int*)(cs + ((u32)(row + rx1 + 1) << 2)) - s_row;
                    }
                }
                if ((LIST || OUTL) && !__builtin_amdgcn_ballot_w64(cnt > 0)) continue;     // nothing in these 64 rows
                const int incl = wave_scan_incl(cnt);   // incl[r] = first slot of row r+1
                const int sb = s_row - (incl - cnt);      // candidate slot t of row r is particle sb[r] + t
                const int T = __builtin_amdgcn_readlane(incl, 63);
                ncand += (u64)T;
#ifdef SPHX_KNN_PROF
                const long long ts1 = clock64();
                q_sec[0] += (u64)(ts1 - ts0); q_sec[3] += 1;
#endif

                const int off = incl - cnt;
                // Slot -> particle map.  Non-empty rows are compacted (rbase[ordinal] = sb) and each
                // marks the slot where it starts; a batch then finds its rows with one flag read, one
                // ballot and a popcount (the flags are cleared again after the chunk).  Chunks with
                // more than KNN_FLAG_CAP slots use the 6-step shuffle search instead.
                const bool use_flags = (T <= KNN_FLAG_CAP);
                const bool ne = cnt > 0;
                if (use_flags) {
                    const u64 nem = __builtin_amdgcn_ballot_w64(ne);
                    if (ne) {
                        rbase[lanes_below(nem)] = sb;
                        rflag[off] = 1;
                    }
                    wave_sync();
                }
                int carry = 0;                // non-empty rows that started before this batch
                // candidate slots t0 .. t0+63 -> particle indices
                auto slots_to_particles = [&](int t0) -> int {
                    const int t = t0 + lane;
                    const bool valid = t < T;
                    const int tt = valid ? t : 0;
                    int p;
                    if (use_flags) {
                        const bool fl = rflag[tt] != 0;
                        const u64 M = __builtin_amdgcn_ballot_w64(valid) & __builtin_amdgcn_ballot_w64(fl);
                        const int ord = carry + lanes_below(M) - ((valid && fl) ? 0 : 1);
                        carry += __popcll(M);
                        p = rbase[valid ? ord : 0] + tt;
                    } else {
                        int rr = 0;           // largest row with off[row] <= tt
#pragma unroll
                        for (int st = 32; st >= 1; st >>= 1) {
                            int pr = rr + st;
                            int v = __shfl(off, pr, 64);
                            if (v <= tt) rr = pr;
                        }
                        p = __shfl(sb, rr, 64) + tt;
                    }
                    if (indirect) p = valid ? a.ol.list[p] : 0;
                    return p;
                };
                // list mode is bound by the latency of its longest queries (one wave walking thousands of batches): the
                // next batch's positions are requested before this one's are looked at
                constexpr bool PIPE = (LIST || OUTL) && ABL == 0;
                int p_next = 0;
                double nxp = 0.0, nyp = 0.0, nzp = 0.0;
                if (PIPE && T > 0) {
                    p_next = slots_to_particles(0);
                    const u32 bo = (u32)p_next << 3;
                    nxp = *(const double*)((const char*)a.x + bo); nyp = *(const double*)((const char*)a.y + bo);
                    nzp = *(const double*)((const char*)a.z + bo);
                }
                for (int t0 = 0; t0 < (ABL == 2 ? 0 : T); t0 += 64) {
                    const int t = t0 + lane;
                    const bool valid = t < T;
                    int p;
                    double cxp, cyp, czp;
                    if (PIPE) {
                        p = p_next; cxp = nxp; cyp = nyp; czp = nzp;
                        if (t0 + 64 < T) {
                            p_next = slots_to_particles(t0 + 64);
                            const u32 bo = (u32)p_next << 3;
                            nxp = *(const double*)((const char*)a.x + bo); nyp = *(const double*)((const char*)a.y + bo);
                            nzp = *(const double*)((const char*)a.z + bo);
                        }
                    } else {
                        p = slots_to_particles(t0);
                        // 32-bit byte offsets (n <= 2^29, checked by the launcher): scalar base + one VGPR
                        const u32 boff = (u32)p << 3;
                        cxp = *(const double*)((const char*)a.x + boff); cyp = *(const double*)((const char*)a.y + boff);
                        czp = *(const double*)((const char*)a.z + boff);
                    }
                    const double d2 = dist2_nofma(cxp - xi, cyp - yi, czp - zi);
                    u64 key = (u64)__double_as_longlong(d2);
                    if (filter && sphx_outside_box(g, cxp, cyp, czp)) key = KNN_INF;   // (never below the threshold)
                    // d2 >= 0 so the bit pattern orders like the value; NaN keys (> INF) never pass.
                    // ties in d2 are broken by the candidate's position in the cell-sorted order,
                    // which sphx_grid.hip makes deterministic (cells sorted by previous index)
                    const u32 pid = (u32)p;
                    const bool keep = valid && key < tk;
                    const u64 mask = __builtin_amdgcn_ballot_w64(valid) & __builtin_amdgcn_ballot_w64(key < tk);
                    const int c = __popcll(mask);
                    if (c && ABL != 4) {
                        if (keep) {
                            int pos = (head + nst + lanes_below(mask)) & 127;
                            skey[pos] = key;
                            sid[pos] = pid;
                        }
                        nst += c;
                        if (nst >= 64) {
#ifdef SPHX_KNN_PROF
                            const long long td0 = clock64();
#endif
                            wave_sync();
                            u64 ck; u32 cv;
                            sort_staged<ABL>(skey, sid, head, 64, R2, lane, ck, cv);
                            wave_sync();
                            head = (head + 64) & 127;
                            nst -= 64;
                            if (have_best) {
                                merge_sorted(bk, bv, ck, cv, lane);
                            } else {
                                bk = ck; bv = cv;
                                have_best = true;
                            }
                            // running threshold: the K-th best - or the 64th when the whole register
                            // set is kept as a Verlet list, which must then be exact to its last entry
                            const u64 kth = __shfl(bk, KT - 1, 64);
                            if (kth != KNN_INF) {
                                tk = kth + 1ull;
                                // nothing beyond the K-th best can still enter: clip what remains of this structure to it
                                // (same padding as the trial radius; an oversized sphere - a stale hint, a ladder step too
                                // far - then costs about what the right one would have)
                                const double rkd = sqrt(__longlong_as_double((long long)kth)) * icell;
                                const float rk = (float)rkd * 1.00001f + 2e-3f;
                                Rc2 = fminf(Rc2, rk * rk);
                            }
#ifdef SPHX_KNN_PROF
                            d_iter += (u64)(clock64() - td0);
#endif
                        }
                    }
                }
                if (use_flags) {              // leave the flag array clean for the next chunk
                    wave_sync();
                    if (ne) rflag[off] = 0;
                    wave_sync();
                }
#ifdef SPHX_KNN_PROF
                q_sec[1] += (u64)(clock64() - ts1) - d_iter; q_sec[2] += d_iter;
#endif
            }
            if (!OUTL || !multi || lv == 0) break;
            // between structures: rank what is staged; with K candidates in hand nothing beyond the K-th can matter
            if (nst > 0) {
                wave_sync();
                u64 ck; u32 cv;
                sort_staged<ABL>(skey, sid, head, nst, R2, lane, ck, cv);
                wave_sync();
                head = (head + nst) & 127;
                nst = 0;
                if (have_best) {
                    merge_sorted(bk, bv, ck, cv, lane);
                } else {
                    bk = ck; bv = cv;
                    have_best = true;
                }
            }
            {
                const u64 kth_now = __shfl(bk, K - 1, 64);
                if (kth_now != KNN_INF) {
                    tk = kth_now + 1ull;
                    const double rk = sqrt(__longlong_as_double((long long)kth_now)) * (1.0 + 1e-15);
                    if (rk < Rcur) {
                        Rcur = rk;
                        const double mq = fmax(fmax(fabs(xi - a.ol.cx), fabs(yi - a.ol.cy)), fabs(zi - a.ol.cz));
                        const double mlo = fmax(mq - Rcur, 0.0) * (1.0 - 1e-9) / a.ol.hmax, mhi = (mq + Rcur) * (1.0 + 1e-9) / a.ol.hmax;
                        while (lv_lo < a.ol.L && ldexp(1.0, lv_lo) <= mlo) ++lv_lo;      // (the range can only narrow)
                        while (lv_hi > lv_lo && ldexp(1.0, lv_hi - 1) > mhi) --lv_hi;
                    }
                }
            }
            // next structure: the levels from the top down (the own one is done), then the grid
            if (!own_done) { own_done = true; lv = lv_hi + 1; }
            do { --lv; } while (lv == lv_own && lv >= lv_lo);
            if (lv < lv_lo) lv = 0;
            }
            if (nst > 0) {
                wave_sync();
                u64 ck; u32 cv;
                sort_staged<ABL>(skey, sid, head, nst, R2, lane, ck, cv);
                wave_sync();
                if (have_best) {
                    merge_sorted(bk, bv, ck, cv, lane);
                } else {
                    bk = ck; bv = cv;
                }
            }
            const u64 kth = __shfl(bk, K - 1, 64);
            const bool full = (kth != KNN_INF);
            // the sphere contains the true bounding box of all particles (which may be much larger
            // than the grid box when particles have escaped): every particle has been a candidate
            // (evaluated only when the try came up short; the box is read from memory then)
            bool covers = false;
            if (!full) {
                const double* tb = a.tbox;
                const double fx2 = fmax(fabs(xi - tb[0]), fabs(xi - tb[3]));
                const double fy2 = fmax(fabs(yi - tb[1]), fabs(yi - tb[4]));
                const double fz2 = fmax(fabs(zi - tb[2]), fabs(zi - tb[5]));
                covers = !(R2 < fx2 * fx2 + fy2 * fy2 + fz2 * fz2);
            }
            saw_all = covers;
            done = full || covers || at_bound || (++tries >= KNN_MAX_TRIES);
            if (ABL == 0 && !full && !covers && !at_bound && tries >= KNN_MAX_TRIES) ++nshort;   // gave up short: reported
            if (ABL != 0) done = true;    // timing experiments never retry
            if (!done) {
                double grow = 1.6;
                if (LIST || a.distrust) {
                    // list mode: the short try's own count sizes the next step (c of K found inside R: uniform density
                    // puts K inside R (K/c)^(1/3)), between the x1.6 of the general ladder and x3
                    const int c = __popcll(__builtin_amdgcn_ballot_w64(bk != KNN_INF));
                    grow = fmin(fmax(cbrt(1.5 * (double)K / (double)(c > 0 ? c : 1)), 1.6), 3.0);
                }
                if (R_hint > 0.0 && R < R_hint) {
                    R = fmin(R * grow, R_hint);
                    if (R >= R_hint) R_hint = 0.0;
                } else {
                    R *= grow;
                }
                if (R > a.rbound) R = a.rbound;
                ++nretry;
            }
        } while (!done);

        // ---- outputs: lanes 0..K-1 hold the neighbours in ascending (d2, id) order ----
        const bool valid = (lane < K) && (bk != KNN_INF);
        const int found = __popcll(__builtin_amdgcn_ballot_w64(valid));
        const double d = valid ? sqrt(__longlong_as_double((long long)bk)) : 0.0;
        const double dlast = __shfl(d, found > 0 ? found - 1 : 0, 64);
        const double hval = found > 0 ? dlast : 0.0;
#ifdef SPHX_KNN_PROF
        {
            const u64 dtc = (u64)(clock64() - prof_t0);
            const int pc = prof_cat >= 2 ? 1 : 0;
            p_sum[pc] += dtc; p_n[pc] += 1; p_tries[pc] += (u64)(tries + 1);
            for (int q = 0; q < 4; ++q) p_sec[pc][q] += q_sec[q];
            if (dtc > p_max[pc]) p_max[pc] = dtc;
            if (dtc > SPHX_KNN_PROF_LONG && lane == 0 && a.counters) {        // a monster: who is it?
                double* dbg = (double*)(a.counters + SC_KNNPROF + 16);
                dbg[0] = xi; dbg[1] = yi; dbg[2] = zi; dbg[3] = R_given; dbg[4] = hval; dbg[5] = (double)tries;
                dbg[6] = (double)(ncand - prof_c0); dbg[7] = (double)dtc;
            }
        }
#endif
        if (a.distrust && a.rsearch && (hval * a.rscale > 1.5 * R_given || hval * a.rscale < 0.5 * R_given)) ++nbad;
        if (lane < K) {
            if (LEAN || a.nbr) tile(lane, li) = valid ? (int)bv : -1;
            if (!LEAN && a.idx64) a.idx64[(long long)oid * K + lane] = valid ? (long long)a.id[bv] : (long long)a.n;
            if (!LEAN && a.dist) a.dist[(long long)oid * K + lane] = d;
        }
        const int i = __builtin_amdgcn_readlane(qs, t16);        // storage index of this query
        if (!LEAN && a.list64) {
            // Verlet list for sphx_refresh.hip: the 64 nearest inside the final radius R.  Anything
            // not listed was farther than the 64th entry (list full) or than R (list not full).
            a.list64[(size_t)i * 64 + lane] = (bk != KNN_INF) ? (int)bv : -1;
            const u64 k63 = __shfl(bk, 63, 64);
            if (lane == 0)
                a.dref[i] = saw_all ? 1e300 : (k63 != KNN_INF ? sqrt(__longlong_as_double((long long)k63)) : R);
        }
        if (lane == 0) {
            if (LEAN == 1 || (!LEAN && a.h_sorted)) a.h_sorted[i] = hval;
            if (LEAN == 2 || (!LEAN && a.h_by_id)) a.h_by_id[oid] = hval;
            if (!LEAN && a.nontriv) a.nontriv[oid] = found;
        }
    }
    if (LEAN || a.nbr) {
        __syncthreads();
        if (LIST) {
            const int i = entry_of(lane < PPB ? lane : 0);
            const int slot = (lane < PPB && i < total) ? (int)((unsigned)a.qlist[i] & 0x1FFFFFFFu) : -1;
            for (int kk = wave; kk < K; kk += KNN_BLOCK / 64)
                if (slot >= 0) a.nbr[(long long)kk * a.npad + slot] = tile(kk, lane);
            __syncthreads();                           // the tile is reused by the next list chunk
        } else {
            for (int kk = wave; kk < K; kk += KNN_BLOCK / 64) {
                int i = base + lane;
                if (i < a.npad) a.nbr[(long long)kk * a.npad + i] = tile(kk, lane);
            }
        }
    }
    vblock += nblk;
    } while (LIST);
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[SC_CAND], ncand);
        if (nretry) atomicAdd(&a.counters[SC_RETRY], nretry);
        if (nshort) atomicAdd(&a.counters[SC_SHORT], nshort);
        if (nfar) atomicAdd(&a.counters[SC_FARQ], nfar);
        if (nbad) atomicAdd(&a.counters[SC_BADHINT], nbad);
        if (LIST && blockIdx.x == 0 && threadIdx.x == 0) a.counters[SC_NFAILQ] = (u64)(u32)total;      // this search's list length
#ifdef SPHX_KNN_PROF
        for (int pc = 0; pc < 2; ++pc)
            if (p_n[pc]) {
                atomicAdd(&a.counters[SC_KNNPROF + 2 * pc], p_sum[pc]);
                atomicAdd(&a.counters[SC_KNNPROF + 4 + 2 * pc], p_n[pc]);
                atomicMax(&a.counters[SC_KNNPROF + 8 + 2 * pc], p_max[pc]);
                atomicAdd(&a.counters[SC_KNNPROF + 12 + 2 * pc], p_tries[pc]);
                for (int q = 0; q < 4; ++q) atomicAdd(&a.counters[SC_KNNPROF + 4 * q + 2 * pc + 1], p_sec[pc][q]);
            }
#endif
    }
}

int sphx_knn(sphx_ctx* ctx, int64_t n, int k, const double* xs, const double* ys,
             const double* zs, const int32_t* id, const int32_t* inv, const double* rsearch,
             double rscale, double rbound, const KnnOut& out) {
    if (k < 1 || k > SPHX_MAX_K) return sphx_set_err(ctx, SPHX_E_ARG, "k=%d not in 1..%d", k, SPHX_MAX_K);
    if (n < 1 || n > (1ll << 29)) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range (1..2^29)", (long long)n);
    ctx->nbr_api_valid = false;          // the K-major list is about to be overwritten
    KnnArgs a;
    a.n = (int)n;
    a.k = k;
    a.npad = (int)sphx_pad64(n);
    a.n_active = ctx->map_perm ? ctx->map_nactive : 0x7FFFFFFF;
    a.x = xs; a.y = ys; a.z = zs;
    a.id = id; a.inv = inv;
    a.qorder = ctx->qorder;
    a.cell_start = ctx->cell_start.as<int>();
    a.g = ctx->grid;
    a.tbox = ctx->tbox;
    a.rsearch = rsearch;
    a.hint_by_id = ctx->knn_hint_by_id ? 1 : 0;
    a.rscale = rscale;
    a.tie_list = nullptr; a.tie_count = nullptr; a.tie_cap = 0; a.list_blocks = 0;
    a.rbound = (rbound > 0.0) ? rbound : INFINITY;
    a.nbr = out.nbr;
    a.list64 = out.list64;
    a.dref = out.dref;
    a.h_sorted = out.h_sorted;
    a.idx64 = (long long*)out.idx64;
    a.dist = out.dist;
    a.nontriv = (long long*)out.nontriv;
    a.h_by_id = out.h_by_id;
    a.counters = ctx->scal.as<u64>();
    int blocks = (int)(sphx_pad64(n) / KNN_PPB);
#ifdef SPHX_EXPERIMENTS
    if (ctx->exp_knn >= 0) {       // timing experiment (SPHX_KNN_ABL), results discarded
        KnnArgs b = a;
        b.nbr = nullptr; b.list64 = nullptr; b.dref = nullptr; b.h_sorted = nullptr; b.idx64 = nullptr; b.dist = nullptr; b.nontriv = nullptr;
        b.h_by_id = nullptr; b.counters = nullptr;
        const int mode = ctx->exp_knn;
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, ctx->stream));
        if (mode == 1) hipLaunchKernelGGL((knn_kernel<1, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, b);
        else if (mode == 2) hipLaunchKernelGGL((knn_kernel<2, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, b);
        else if (mode == 3) hipLaunchKernelGGL((knn_kernel<3, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, b);
        else if (mode == 4) hipLaunchKernelGGL((knn_kernel<4, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, b);
        else if (mode == 5 && a.nbr && a.h_sorted) hipLaunchKernelGGL((knn_kernel<5, 1>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);   // lean kernel, one try only (same outputs but for the few short queries; the real launch follows)
        else hipLaunchKernelGGL((knn_kernel<0, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, b);
        HIPCHK(hipEventRecord(e1, ctx->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        fprintf(stderr, "[sphx] knn ablation %d: %.4f ms\n", mode, ms);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
#endif
    a.qlist = nullptr; a.qcount = nullptr;
    a.distrust = 0;
    a.ol.L = 0; a.ol.start = nullptr; a.ol.list = nullptr; a.ol.cx = a.ol.cy = a.ol.cz = 0.0; a.ol.hmax = 1.0;
    const bool lean = a.nbr && a.h_sorted && !a.list64 && !a.idx64 && !a.dist && !a.nontriv && !a.h_by_id &&
                      a.counters;
    const bool lean2 = a.nbr && a.h_by_id && !a.h_sorted && !a.list64 && !a.idx64 && !a.dist && !a.nontriv &&
                       a.counters;
    // Hinted searches of the step loop and of the device API: the lane-per-query grouped kernel first, then this
    // kernel in list mode for whatever it could not certify (sphx_knn_group.hip).
    if ((lean || lean2) && ctx->use_group && ctx->knn_hinted && rsearch && ctx->exp_knn < 0) {
        const size_t tile_bytes = (size_t)k * (KNN_PPB + 1) * sizeof(int);
        // What the previous hinted search reported (copied out behind it, on the host long since: no wait): queries it
        // left to the general kernel, far queries (outside the grid box, wide spheres), radii that contradicted their hints
        // (the fused loop copies these slots out together with its mean-h read-back and hands them over: knn_lag_*)
        const bool lagged = ctx->olev_mode == 2 || ctx->distrust_mode == 2;
        const bool ext = ctx->knn_lag_external;
        if (!ctx->capturing && lagged && (ext ? ctx->knn_lag_valid : ctx->olev_ev_valid)) {
            if (!ext) HIPCHK(hipEventSynchronize(ctx->olev_ev));
            const u64* v = ext ? ctx->knn_lag : (const u64*)((const char*)ctx->pinned + 3072);   // slots SC_NFAILQ .. SC_BADHINT
            const int64_t fb = (int64_t)(u32)v[0];
            ctx->list_len_last = fb;
            ctx->farq_last = (int64_t)(v[2] >= ctx->farq_seen ? v[2] - ctx->farq_seen : v[2]);
            ctx->farq_seen = v[2];
            const int64_t bad = (int64_t)(v[3] >= ctx->badhint_seen ? v[3] - ctx->badhint_seen : v[3]);
            ctx->badhint_seen = v[3];
            ctx->crowded_last = (int64_t)(v[4] >= ctx->crowded_seen ? v[4] - ctx->crowded_seen : v[4]);
            ctx->crowded_seen = v[4];
            ctx->densep_last = (int64_t)(v[5] >= ctx->densep_seen ? v[5] - ctx->densep_seen : v[5]);
            ctx->densep_seen = v[5];
            if (ctx->distrust_mode == 2) {
                if (!ctx->distrust) ctx->distrust = fb * 4 > n;
                else ctx->distrust = bad * 20 > n;
            }
            ctx->olev_ev_valid = false;
            ctx->knn_lag_valid = false;
        }
        if (ctx->distrust_mode != 2) ctx->distrust = ctx->distrust_mode == 1;
        // Outlier levels: built when that search met many far queries - a diverging run's escapers, an expanding cloud's rim
        // ... or when the true bounding box the grid build knew reaches far beyond the grid box
        double excess = 0.0;
        {
            const GridParams& g = ctx->grid;
            const double lo[3] = {g.xmin, g.ymin, g.zmin}, hi[3] = {g.xmin + g.nx * g.cell, g.ymin + g.ny * g.cell, g.zmin + g.nz * g.cell};
            for (int c = 0; c < 3; ++c) {
                const double e0 = (lo[c] - ctx->tbox_h[c]) * g.inv_cell, e1 = (ctx->tbox_h[3 + c] - hi[c]) * g.inv_cell;
                if (e0 > excess) excess = e0;
                if (e1 > excess) excess = e1;
            }
        }
        const bool levels = ctx->olev_mode == 1 || (ctx->olev_mode == 2 && (ctx->farq_last >= 256 || excess > 64.0));
        a.ol.L = 0;
        if (levels) {
            SPHX_TRY(sphx_build_outlier_levels(ctx, n, xs, ys, zs));
            a.ol = ctx->olev;
        }
        ctx->stats.far_queries = ctx->farq_last;
        ctx->stats.outlier_levels = a.ol.L;
        if (ctx->distrust) {
            // every query by the general kernel, radii seeded from the cell counts (the hint is one rung of the ladder)
            a.distrust = 1;
            if (a.ol.L > 0) {
                if (lean) hipLaunchKernelGGL((knn_kernel<0, 1, 0, 1>), dim3(blocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
                else hipLaunchKernelGGL((knn_kernel<0, 2, 0, 1>), dim3(blocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
            } else {
                if (lean) hipLaunchKernelGGL((knn_kernel<0, 1>), dim3(blocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
                else hipLaunchKernelGGL((knn_kernel<0, 2>), dim3(blocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(ctx->scal.as<u64>() + SC_NFAILQ, 0, sizeof(u64), ctx->stream));
        } else {
            SPHX_TRY(sphx_ensure(ctx, ctx->fail_list, ((size_t)a.npad + 64) * sizeof(int)));
            int* flist = ctx->fail_list.as<int>();
            int* fcount = flist + a.npad;
            if (!(ctx->fcount_zeroed == ctx->fail_list.p && ctx->fcount_zeroed_n == n))     // (else: the grid build's first kernel did it)
                HIPCHK(hipMemsetAsync(fcount, 0, 2 * sizeof(int), ctx->stream));           // (fail count, tie count)
            ctx->fcount_zeroed = nullptr;
            KnnGroupArgs ga;
            ga.n = a.n; ga.k = a.k; ga.npad = a.npad; ga.n_active = a.n_active;
            ga.x = a.x; ga.y = a.y; ga.z = a.z; ga.id = a.id; ga.qorder = a.qorder; ga.cell_start = a.cell_start;
            ga.g = a.g; ga.rsearch = a.rsearch; ga.hint_by_id = a.hint_by_id; ga.rscale = a.rscale; ga.rbound = a.rbound;
            ga.nbr = a.nbr; ga.h_sorted = lean ? a.h_sorted : nullptr; ga.h_by_id = lean2 ? a.h_by_id : nullptr;
            ga.fail_list = flist; ga.fail_count = fcount; ga.counters = a.counters;
            ga.tie_list = nullptr; ga.tie_count = fcount + 1; ga.tie_cap = 0;
            if (ctx->tie_fix) {
                ga.tie_cap = a.npad / 16 + 1024;
                SPHX_TRY(sphx_ensure(ctx, ctx->tie_list, (size_t)ga.tie_cap * sizeof(int4)));
                ga.tie_list = ctx->tie_list.as<int4>();
            }
            SPHX_TRY(sphx_knn_group(ctx, ga));
            a.qlist = flist; a.qcount = fcount;
            int lblocks = (int)((ctx->list_len_last / 32 + 255) / 256) * 256;
            lblocks = lblocks < KNN_LIST_BLOCKS ? KNN_LIST_BLOCKS : (lblocks > KNN_LIST_BLOCKS_MAX ? KNN_LIST_BLOCKS_MAX : lblocks);
            if (lblocks > blocks) lblocks = blocks;
            int tblocks = 0;                      // near ties: ordered by 16 further blocks of the same launch
            if (ga.tie_list) {
                a.tie_list = ga.tie_list; a.tie_count = ga.tie_count; a.tie_cap = ga.tie_cap; a.list_blocks = lblocks;
                tblocks = 16;
            }
            if (a.ol.L > 0) {
                if (lean) hipLaunchKernelGGL((knn_kernel<0, 1, 1, 1>), dim3(lblocks + tblocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
                else hipLaunchKernelGGL((knn_kernel<0, 2, 1, 1>), dim3(lblocks + tblocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
            } else {
                if (lean) hipLaunchKernelGGL((knn_kernel<0, 1, 1>), dim3(lblocks + tblocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
                else hipLaunchKernelGGL((knn_kernel<0, 2, 1>), dim3(lblocks + tblocks), dim3(KNN_BLOCK), tile_bytes, ctx->stream, a);
            }
            HIPCHK(hipGetLastError());
            // (SC_NFAILQ is written by the list-mode launch itself)
        }
        if (lagged && !ext && !ctx->capturing) {
            if (!ctx->olev_ev) HIPCHK(hipEventCreateWithFlags(&ctx->olev_ev, hipEventDisableTiming));
            HIPCHK(hipMemcpyAsync((char*)ctx->pinned + 3072, ctx->scal.as<u64>() + SC_NFAILQ, 10 * sizeof(u64), hipMemcpyDeviceToHost,
                                  ctx->stream));
            HIPCHK(hipEventRecord(ctx->olev_ev, ctx->stream));
            ctx->olev_ev_valid = true;
        }
        return SPHX_OK;
    }
    if (lean) hipLaunchKernelGGL((knn_kernel<0, 1>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
    else if (lean2) hipLaunchKernelGGL((knn_kernel<0, 2>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
    else hipLaunchKernelGGL((knn_kernel<0, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
