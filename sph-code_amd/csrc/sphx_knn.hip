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
#define KNN_LIST_BLOCKS 2048
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

#ifndef KNN_MIN_WAVES
#define KNN_MIN_WAVES 6      // waves per SIMD the register budget is held to (6 -> <= 80 VGPRs; measured fastest)
#endif
// ABL != 0: timing experiments with a section removed (outputs are then meaningless and are
// never written: the wrapper passes null output pointers); ABL == 0 is the product kernel.
// LEAN = 1: the step loop's variant - only the K-major list and h (sorted order) are produced,
// so the API / Verlet-list pointers are never loaded (17 pointers in SGPRs otherwise: measured
// 20 % slower).  LEAN = 2: the same for the device API (h written by id).
template <int ABL, int LEAN = 0, int LIST = 0>
__global__ __launch_bounds__(KNN_BLOCK, KNN_MIN_WAVES) void knn_kernel(KnnArgs a) {
    extern __shared__ int tile_dyn[];                 // [K][KNN_PPB + 1] result tile (sized at launch)
#define tile(kk, li) tile_dyn[(kk) * (KNN_PPB + 1) + (li)]
    __shared__ u64 stg_key[KNN_BLOCK / 64][128];
    __shared__ u32 stg_id[KNN_BLOCK / 64][128];
    // candidate-slot -> row lookup: one start flag per candidate slot + the compacted row bases
    __shared__ unsigned char row_flag[KNN_BLOCK / 64][KNN_FLAG_CAP];
    __shared__ int row_base[KNN_BLOCK / 64][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    u64 ncand = 0, nretry = 0, nshort = 0;

    do {
    // list mode: 4 queries per wave and pass instead of 16 - the list is short (a few per cent of the queries), so
    // the launch is bound by how long one wave takes, not by how many waves there are
    constexpr int PPB = LIST ? KNN_LIST_PPB : KNN_PPB;
    if (LIST && vblock * PPB >= total) break;
    const int base = vblock * PPB;
    // the wave's 16 query particles are fetched in ONE coalesced round trip (lane l holds
    // particle l) and broadcast through SGPRs as each comes up
    double qx = 0.0, qy = 0.0, qz = 0.0, qr = 0.0;
    int qid = 0x7FFFFFFF;
    int qs = 0;                      // where the lane's query particle is stored
    {
        const int ip = base + wave * (PPB / 4) + (lane & 15);
        if (lane < PPB / 4 && ip < total) {
            const int slot = LIST ? a.qlist[ip] : ip;
            qs = a.qorder ? a.qorder[slot] : slot;
            qx = a.x[qs]; qy = a.y[qs]; qz = a.z[qs];
            qid = a.id[qs];
            qr = a.rsearch ? a.rsearch[a.hint_by_id ? qid : qs] * a.rscale : 0.0;
        }
    }

    for (int t16 = 0; t16 < PPB / 4; ++t16) {
        const int li = wave * (PPB / 4) + t16;
        const int oid = __builtin_amdgcn_readlane(qid, t16);
        // wave-uniform: past the end, or a ghost (a candidate, never a query)
        if (base + li >= total || oid >= a.n_active) {
            if (lane < K) tile(lane, li) = -1;
            continue;
        }
        const double xi = bcast_f64(qx, t16), yi = bcast_f64(qy, t16), zi = bcast_f64(qz, t16);
        double R = bcast_f64(qr, t16);
        if (!(R > 0.0)) {
            // density estimate from the 3x3x3 block of cells around the particle
            const int cxi = cell_coord(xi, g.xmin, g.inv_cell, g.nx - 1);
            const int cyi = cell_coord(yi, g.ymin, g.inv_cell, g.ny - 1);
            const int czi = cell_coord(zi, g.zmin, g.inv_cell, g.nz - 1);
            int cnt = 0, nc = 0;
            if (lane < 9) {
                int cy = cyi - 1 + lane % 3, cz = czi - 1 + lane / 3;
                if (cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz) {
                    int xlo = max(cxi - 1, 0), xhi = min(cxi + 1, g.nx - 1);
                    int row = (cz * g.ny + cy) * g.nx;
                    cnt = a.cell_start[row + xhi + 1] - a.cell_start[row + xlo];
                    nc = xhi - xlo + 1;
                }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                cnt += __shfl_xor(cnt, o, 64);
                nc += __shfl_xor(nc, o, 64);
            }
            cnt = __shfl(cnt, 0, 64);
            nc = __shfl(nc, 0, 64);
            double vol = (double)nc * g.cell * g.cell * g.cell;
            double dens = (double)(cnt > 0 ? cnt : 1) / vol;
            R = 1.3 * cbrt((double)K / (4.1887902047863905 * dens));
        }
        if (R > a.rbound) R = a.rbound;

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
            // Query position and radius in CELL units: fp64 once, then all range geometry in fp32.
            // Every fp32 quantity is padded (radius x(1+1e-5) + 2e-3 cells, distances - 1e-3 cells;
            // fp32 resolves 1.2e-4 cells at index 2047), so the clipped ranges can only grow: a
            // superset of the exact fp64 ranges, never a lost neighbour.
            const float fx = (float)((xi - g.xmin) * g.inv_cell);
            const float fy = (float)((yi - g.ymin) * g.inv_cell);
            const float fz = (float)((zi - g.zmin) * g.inv_cell);
            const float Rc = (float)(R * g.inv_cell) * 1.00001f + 2e-3f;
            const float nx1 = g.fnx1, ny1 = g.fny1, nz1 = g.fnz1;
            // rows are tested against the query clamped into the grid: with it the half-infinite
            // boundary cells need no special case (their open side can only lie behind the query)
            const float fyc = fminf(fmaxf(fy, 0.0f), ny1 + 1.0f), fzc = fminf(fmaxf(fz, 0.0f), nz1 + 1.0f);
            const int cy0 = (int)fminf(fmaxf(fy - Rc, 0.0f), ny1);
            const int cy1 = (int)fminf(fmaxf(fy + Rc, 0.0f), ny1);
            const int cz0 = (int)fminf(fmaxf(fz - Rc, 0.0f), nz1);
            const int cz1 = (int)fminf(fmaxf(fz + Rc, 0.0f), nz1);
            const int ysp = cy1 - cy0 + 1;
            const int nrows = __mul24(ysp, cz1 - cz0 + 1);
            const float inv_ysp = __builtin_amdgcn_rcpf((float)ysp);
            const float Rc2 = Rc * Rc;
            int nst = 0, head = 0;            // staging ring occupancy / head (wave-uniform)
            bool have_best = false;           // best[] still empty: first flush is a plain sort

            for (int rb = 0; rb < (ABL == 3 ? 0 : nrows); rb += 64) {
                // ---- one lane per (cy,cz) row of cells: clip the row to the search SPHERE ----
                const int r = rb + lane;
                int s_row = 0, cnt = 0;
                if (r < nrows) {
                    // r / ysp: fp32 estimate (r < 2^24 rows, quotient <= 4096: off by one at most) + fix-up
                    int rz = (int)(((float)r + 0.5f) * inv_ysp);
                    int ry = r - __mul24(rz, ysp);
                    if (ry < 0) { --rz; ry += ysp; }
                    if (ry >= ysp) { ++rz; ry -= ysp; }
                    const int cy = cy0 + ry, cz = cz0 + rz;
                    // distance (in cells) from the query to the row's (y,z) cell column.  Boundary cells
                    // are half-infinite: out-of-box coordinates are clamped into them (sphx_grid.hip).
                    const float cyf = (float)cy, czf = (float)cz;
                    const float dy = fmaxf(fmaxf(cyf - fyc, fyc - (cyf + 1.0f)) - 1e-3f, 0.0f);
                    const float dz = fmaxf(fmaxf(czf - fzc, fzc - (czf + 1.0f)) - 1e-3f, 0.0f);
                    const float rem = Rc2 - (dy * dy + dz * dz);
                    if (rem >= 0.0f) {        // the row meets the sphere: chord along x
                        const float hc = __builtin_amdgcn_sqrtf(rem) * 1.00001f + 1e-3f;   // 1 ulp: inside the padding
                        const int rx0 = (int)fminf(fmaxf(fx - hc, 0.0f), nx1);
                        const int rx1 = (int)fminf(fmaxf(fx + hc, 0.0f), nx1);
                        // < 2^23 cells in all: 24-bit multiplies, 32-bit byte offsets
                        const int row = __mul24(__mul24(cz, g.ny) + cy, g.nx);
                        const char* cs = (const char*)a.cell_start;
                        s_row = *(const int*)(cs + ((u32)(row + rx0) << 2));
                        cnt = *(const int*)(cs + ((u32)(row + rx1 + 1) << 2)) - s_row;
                    }
                }
                const int incl = wave_scan_incl(cnt);   // incl[r] = first slot of row r+1
                const int sb = s_row - (incl - cnt);      // candidate slot t of row r is particle sb[r] + t
                const int T = __builtin_amdgcn_readlane(incl, 63);
                ncand += (u64)T;

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
                for (int t0 = 0; t0 < (ABL == 2 ? 0 : T); t0 += 64) {
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
                    // 32-bit byte offsets (n <= 2^29, checked by the launcher): scalar base + one VGPR
                    const u32 boff = (u32)p << 3;
                    const double d2 = dist2_nofma(*(const double*)((const char*)a.x + boff) - xi,
                                                  *(const double*)((const char*)a.y + boff) - yi,
                                                  *(const double*)((const char*)a.z + boff) - zi);
                    const u64 key = (u64)__double_as_longlong(d2);
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
                            if (kth != KNN_INF) tk = kth + 1ull;
                        }
                    }
                }
                if (use_flags) {              // leave the flag array clean for the next chunk
                    wave_sync();
                    if (ne) rflag[off] = 0;
                    wave_sync();
                }
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
                R *= 1.6;
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
            const int i = base + lane;
            const int slot = (lane < PPB && i < total) ? a.qlist[i] : -1;
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
    vblock += gridDim.x;
    } while (LIST);
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[SC_CAND], ncand);
        if (nretry) atomicAdd(&a.counters[SC_RETRY], nretry);
        if (nshort) atomicAdd(&a.counters[SC_SHORT], nshort);
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
    a.qlist = nullptr; a.qcount = nullptr;
    const bool lean = a.nbr && a.h_sorted && !a.list64 && !a.idx64 && !a.dist && !a.nontriv && !a.h_by_id &&
                      a.counters;
    const bool lean2 = a.nbr && a.h_by_id && !a.h_sorted && !a.list64 && !a.idx64 && !a.dist && !a.nontriv &&
                       a.counters;
    // Hinted searches of the step loop and of the device API: the lane-per-query grouped kernel first, then this
    // kernel in list mode for whatever it could not certify (sphx_knn_group.hip).
    if ((lean || lean2) && ctx->use_group && ctx->knn_hinted && rsearch && ctx->exp_knn < 0) {
        SPHX_TRY(sphx_ensure(ctx, ctx->fail_list, ((size_t)a.npad + 64) * sizeof(int)));
        int* flist = ctx->fail_list.as<int>();
        int* fcount = flist + a.npad;
        HIPCHK(hipMemsetAsync(fcount, 0, sizeof(int), ctx->stream));
        KnnGroupArgs ga;
        ga.n = a.n; ga.k = a.k; ga.npad = a.npad; ga.n_active = a.n_active;
        ga.x = a.x; ga.y = a.y; ga.z = a.z; ga.id = a.id; ga.qorder = a.qorder; ga.cell_start = a.cell_start;
        ga.g = a.g; ga.rsearch = a.rsearch; ga.hint_by_id = a.hint_by_id; ga.rscale = a.rscale; ga.rbound = a.rbound;
        ga.nbr = a.nbr; ga.h_sorted = lean ? a.h_sorted : nullptr; ga.h_by_id = lean2 ? a.h_by_id : nullptr;
        ga.fail_list = flist; ga.fail_count = fcount; ga.counters = a.counters;
        SPHX_TRY(sphx_knn_group(ctx, ga));
        a.qlist = flist; a.qcount = fcount;
        int lblocks = blocks < KNN_LIST_BLOCKS ? blocks : KNN_LIST_BLOCKS;
        if (lean) hipLaunchKernelGGL((knn_kernel<0, 1, 1>), dim3(lblocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
        else hipLaunchKernelGGL((knn_kernel<0, 2, 1>), dim3(lblocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->scal.as<u64>() + SC_NFAILQ, fcount, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        return SPHX_OK;
    }
    if (lean) hipLaunchKernelGGL((knn_kernel<0, 1>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
    else if (lean2) hipLaunchKernelGGL((knn_kernel<0, 2>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
    else hipLaunchKernelGGL((knn_kernel<0, 0>), dim3(blocks), dim3(KNN_BLOCK), (size_t)k * (KNN_PPB + 1) * sizeof(int), ctx->stream, a);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
