// sphx_blob.hip - the sum passes of the step loop with the neighbour records staged in LDS.
//
// In blob order (sphx_grid.hip: sphx_build_blob_order) the BLOB_P particles of a workgroup form a
// compact blob and share most of their neighbours: 128 particles x 40 references name only ~600
// distinct particles.  blob_dedup_kernel finds that distinct set once per step with an open-
// addressing hash table in LDS whose entry number IS the slot number: it rewrites the workgroup's
// part of the neighbour list as 16-bit slot numbers and stores the table (slot -> particle, -1 =
// empty).  Each pass then fetches every distinct record ONCE per workgroup into an LDS image
// (~5 gathers per particle instead of 40) and runs its neighbour loop out of LDS.
//
// LPP = 4 lanes serve one particle: lane 4t+q sums the list positions k = q mod 4, and the four
// partial sums are added at the end as (p0 + p1) + (p2 + p3) by two DPP exchanges - the same fixed
// order the gather kernels of sphx_sums.hip use, so the results are bit-identical to theirs.  512
// threads per workgroup, two workgroups per CU (LDS-bound): four waves per SIMD hide LDS latency
// (two lanes per particle, i.e. two waves per SIMD, measured 10 % slower), and one workgroup stages
// while the other computes.
//
// LDS image: BLOB_S slots; a record is kept as 16-byte chunks, chunk c of slot s at
// (c * BLOB_S + s) * 16, so one ds_read_b128 of 16 lanes meets 16 bank-quads selected by s mod 16.
// The workgroup's slot lists (K x BLOB_P x 2 B) sit next to it: the neighbour loop touches only LDS.
// A reference that found no free entry within BLOB_PROBES probes (blobs with more than ~900
// distinct neighbours; the most seen at BLOB_P = 128 is ~800) keeps the 0xFFFE marker and is
// fetched from global memory through the int32 list.
#include "sphx_blob.h"
#include "sphx_wave.h"
#pragma clang fp contract(off)
#include <float.h>
#include <stdlib.h>
#include <stdio.h>

// ---- once per step: distinct neighbours of each workgroup ------------------------------------
// Phase 1: every reference is inserted into a sparse open-addressing table of particle indices
// (DD_TAB entries for <= ~900 distinct keys: 1.2 probes on average, where a table as small as the
// image needed 3-4 and its slowest lane 15) and remembers the entry it landed in (LDS tile, 16 bit).
// Phase 2: the occupied entries are numbered in table order by a workgroup prefix sum - the image
// slots, dense from 0 - and written out as the slot -> particle table.  Phase 3: the tile is
// translated entry -> slot in place and leaves as whole 256-B rows.  (The slot numbering depends on
// which lane won an entry; no sum depends on it.)
#define DD_TAB 2048
#define DD_PROBES 64
__device__ __forceinline__ unsigned dd_hash(int j) { return ((unsigned)j * 2654435761u) >> 21; }

// bclass (device API, decomposed runs; nullptr otherwise): what the workgroup's particles need from other ranks -
// 0 "interior": some of its particles are owned and every neighbour of theirs is owned too, so its sums can run before
// the ghosts' values of the step have arrived; 1 "boundary"; 2: all of its particles are ghosts (nothing to compute).
__global__ __launch_bounds__(BLOB_T) void blob_dedup_kernel(int n, int npad, int k, int slots,
                                                            const int* __restrict__ nbr, u16* slot16,
                                                            int* uniq, const int* __restrict__ qorder,
                                                            const int* __restrict__ omap, int n_active,
                                                            unsigned char* bclass) {
    __builtin_amdgcn_s_setprio(DEDUP_PRIO);      // ahead of the record build that streams beside it on the other stream: this kernel is the one the passes wait for (step -13 us)
    extern __shared__ u16 dd_tile[];              // [k][BLOB_P] entry / slot numbers
    __shared__ int key[DD_TAB];
    __shared__ u16 slot_of[DD_TAB];
    __shared__ int wave_tot[BLOB_T / 64];
    const int b = xcd_block(blockIdx.x, gridDim.x);
    const int t = threadIdx.x >> 1, half = threadIdx.x & 1;
    const int p = b * BLOB_P + t;
    for (int q = threadIdx.x; q < DD_TAB; q += BLOB_T) key[q] = -1;
    __syncthreads();
    {
        const int nm = (k + 1) >> 1;
        const bool live = p < n;
        int jn[DD_BATCH];                         // the next batch is in flight while this one is hashed
#pragma unroll
        for (int u = 0; u < DD_BATCH; ++u) {
            const int kk = 2 * u + half;
            jn[u] = (kk < k && live) ? nbr[(size_t)kk * npad + p] : -1;
        }
        for (int m0 = 0; m0 < nm; m0 += DD_BATCH) {
            int jb[DD_BATCH];
#pragma unroll
            for (int u = 0; u < DD_BATCH; ++u) {
                jb[u] = jn[u];
                const int kk = 2 * (m0 + DD_BATCH + u) + half;
                jn[u] = (kk < k && live) ? nbr[(size_t)kk * npad + p] : -1;
            }
            // first probe of the whole batch side by side (the table is sparse: 1.2 probes on average, so nearly every
            // reference is settled here with its LDS round trips overlapped instead of chained), then the stragglers
            unsigned hb[DD_BATCH];
            int eb[DD_BATCH];
#pragma unroll
            for (int u = 0; u < DD_BATCH; ++u) {
                hb[u] = dd_hash(jb[u] < 0 ? 0 : jb[u]);
                eb[u] = key[hb[u]];
            }
#pragma unroll
            for (int u = 0; u < DD_BATCH; ++u)
                if (jb[u] >= 0 && eb[u] == -1) {
                    const int old = atomicCAS(&key[hb[u]], -1, jb[u]);
                    eb[u] = (old == -1) ? jb[u] : old;
                }
#pragma unroll
            for (int u = 0; u < DD_BATCH; ++u) {
                const int kk = 2 * (m0 + u) + half;
                if (kk >= k) continue;
                const int j = jb[u];
                unsigned s = SLOT_NONE;
                if (j >= 0) {
                    unsigned h = hb[u];
                    s = SLOT_OVER;
                    if (eb[u] == j) {
                        s = h;
                    } else {
                        for (int probe = 1; probe < DD_PROBES; ++probe) {
                            h = (h + 1) & (DD_TAB - 1);
                            int e = key[h];
                            if (e == -1) {
                                e = atomicCAS(&key[h], -1, j);
                                if (e == -1) e = j;
                            }
                            if (e == j) { s = h; break; }
                        }
                    }
                }
                dd_tile[kk * BLOB_P + t] = (u16)s;
            }
        }
    }
    __syncthreads();
    // number the occupied entries: DD_TAB / BLOB_T consecutive entries per thread
    constexpr int EPT = DD_TAB / BLOB_T;
    const int e0 = threadIdx.x * EPT;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) cnt += key[e0 + q] != -1;
    const int incl = wave_scan_incl(cnt);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int base = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < BLOB_T / 64; ++w) {
        const int tw = wave_tot[w];
        if (w < wv) base += tw;
        total += tw;
    }
    int* uq = uniq + (size_t)b * BLOB_S;
    int needs_ghost = 0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int j = key[e0 + q];
        if (j != -1) {
            slot_of[e0 + q] = (u16)(base < slots ? base : SLOT_OVER);
            if (base < slots) uq[base] = j;
            ++base;
            if (bclass && omap[j] >= n_active) needs_ghost = 1;
        }
    }
    for (int q = (total < slots ? total : slots) + threadIdx.x; q < BLOB_S; q += BLOB_T) uq[q] = -1;
    __syncthreads();
    // entry -> slot, then out in 16-B pieces of whole rows
    const int pieces = k * (BLOB_P / 8);
    for (int q = threadIdx.x; q < pieces; q += BLOB_T) {
        u16* src = dd_tile + q * 8;
        const int kk = q / (BLOB_P / 8), c = q % (BLOB_P / 8);
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned e = src[u];
            v[u] = e < DD_TAB ? (unsigned)slot_of[e] : e;
            if (v[u] == SLOT_OVER) needs_ghost = 1;      // (an unstaged neighbour is not looked up here: taken as foreign)
        }
        if ((size_t)b * BLOB_P + c * 8 < (size_t)npad)
            *reinterpret_cast<uint4*>(slot16 + (size_t)kk * npad + (size_t)b * BLOB_P + c * 8) =
                make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
    if (bclass) {                                          // (uniform: a kernel argument)
        const int owned_q = (half == 0 && p < n && omap[qorder[p]] < n_active) ? 1 : 0;
        const int any_owned = __syncthreads_or(owned_q);
        const int any_ghost = __syncthreads_or(needs_ghost);
        if (threadIdx.x == 0) bclass[b] = (unsigned char)(!any_owned ? 2 : (any_ghost ? 1 : 0));
    }
}

// interior blobs, then boundary blobs, each in blob order: list[0 .. cnt[0]) and list[cnt[0] .. cnt[0] + cnt[1]).
// One workgroup (a few thousand blobs per 10^6 particles); each thread takes a run of consecutive blobs.
#define SPLIT_T 1024
__global__ __launch_bounds__(SPLIT_T) void blob_split_kernel(int nblk, const unsigned char* __restrict__ bclass, int* list,
                                                             int* cnt) {
    __shared__ int wsum[2][SPLIT_T / 64];
    const int per = (nblk + SPLIT_T - 1) / SPLIT_T;
    const int b0 = threadIdx.x * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
    int c0 = 0, c1 = 0;
    for (int b = b0; b < b1; ++b) { const int c = bclass[b]; c0 += c == 0; c1 += c == 1; }
    const int i0 = wave_scan_incl(c0), i1 = wave_scan_incl(c1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) { wsum[0][wv] = i0; wsum[1][wv] = i1; }
    __syncthreads();
    int o0 = i0 - c0, o1 = i1 - c1, t0 = 0, t1 = 0;
    for (int w = 0; w < SPLIT_T / 64; ++w) {
        if (w < wv) { o0 += wsum[0][w]; o1 += wsum[1][w]; }
        t0 += wsum[0][w]; t1 += wsum[1][w];
    }
    for (int b = b0; b < b1; ++b) {
        const int c = bclass[b];
        if (c == 0) list[o0++] = b;
        else if (c == 1) list[t0 + o1++] = b;
    }
    if (threadIdx.x == 0) { cnt[0] = t0; cnt[1] = t1; cnt[2] = nblk - t0 - t1; }
}

int sphx_blob_translate(sphx_ctx* ctx, int64_t n, int k) {
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
    SPHX_TRY(sphx_ensure(ctx, ctx->slot16, (size_t)k * npad * sizeof(u16) + 512));   // tiles are read in whole rows
    SPHX_TRY(sphx_ensure(ctx, ctx->uniq, (size_t)nblk * BLOB_S * sizeof(int)));
    int slots = ctx->blob_slots;
    if (slots < 1) slots = 1;
    if (slots > BLOB_S) slots = BLOB_S;
    // decomposed runs (device API): blobs sorted into interior / boundary, so that the interior ones can run under a halo phase
    const bool split = ctx->map_perm != nullptr && ctx->blob_split_on;
    unsigned char* bclass = nullptr;
    ctx->blob_split_valid = false;
    if (split) {
        SPHX_TRY(sphx_ensure(ctx, ctx->blob_class, (size_t)nblk));
        SPHX_TRY(sphx_ensure(ctx, ctx->blob_split, ((size_t)nblk + 4) * sizeof(int)));
        bclass = ctx->blob_class.as<unsigned char>();
    }
    hipLaunchKernelGGL(blob_dedup_kernel, dim3(nblk), dim3(BLOB_T), (size_t)k * BLOB_P * sizeof(u16), ctx->stream, (int)n, (int)npad, k, slots,
                       ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(), ctx->qorder, ctx->map_perm,
                       ctx->map_nactive, bclass);
    if (split) {
        int* list = ctx->blob_split.as<int>();
        hipLaunchKernelGGL(blob_split_kernel, dim3(1), dim3(SPLIT_T), 0, ctx->stream, nblk, bclass, list, list + (size_t)nblk);
        ctx->blob_split_valid = true;
        ctx->blob_split_nblk = nblk;
    }
    HIPCHK(hipGetLastError());
    ctx->blob_lists = true;
    return SPHX_OK;
}

// A batch of NB list positions of this lane.  FAST: every lane of the wave has a staged neighbour at
// each of them (the usual case): straight-line LDS reads and arithmetic, nothing to branch on.
// Otherwise a position may be empty (skipped) or unstaged (fetched through the int32 list).
struct DensAcc { double rho, rd, n, gx, gy, gz; };
// WOUT: the batch's species weights Nw_j W_ij (nsc:626) are handed back as well (0 where the list has no neighbour)
template <bool FAST, bool CLIP, bool WOUT = false>
__device__ __forceinline__ void density_batch(DensAcc& a, const unsigned (&sl)[NB], const double2* img,
                                              const RecA* __restrict__ rec, const int* __restrict__ nbr,
                                              size_t col0, size_t colstep, double xr, double yr, double zr,
                                              double hi2, double ci, double Ai, double* wout = nullptr) {
    Q4 q0b[NB], q1b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (FAST || sl[u] < SLOT_OVER) { q0b[u] = lload4(img, (int)sl[u], 0); q1b[u] = lload4(img, (int)sl[u], 1); }
        else if (sl[u] == SLOT_OVER) {
            const double* q = reinterpret_cast<const double*>(&rec[nbr[col0 + u * colstep]]);
            q0b[u] = gload4(q); q1b[u] = gload4(q + 4);
        } else { q0b[u] = Q4{xr, yr, zr, 0.0}; q1b[u] = Q4{0.0, 0.0, 0.0, 0.0}; }
    }
    if (FAST) __builtin_amdgcn_sched_barrier(0);       // all of the batch's LDS reads are issued before its arithmetic
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (WOUT) wout[u] = 0.0;
        if (!FAST && sl[u] == SLOT_NONE) continue;
        const Q4 q0 = q0b[u], q1 = q1b[u];
        const double dx = q0.a - xr, dy = q0.b - yr, dz = q0.c - zr;
        const double r = sqrt_mid(dx * dx + dy * dy + dz * dz);   // nsc:586
        const double r2 = r * r;                              // nsc:588 squares the rounded distance
        const double qj = q0.d - r2;
        const double c1 = q1.a, ms = q1.b, Aj = q1.c, Nw = q1.d;
        double W = c1 * (qj * qj * qj);                       // nsc:588
        W = (W < 0.0) ? 0.0 : W;                              // nsc:589
        const double cb = (CLIP && !(qj > 0.0)) ? 0.0 : -6.0 * c1 * (qj * qj);   // nsc:591 (not clipped; CLIP: nsc:689)
        const double qi = hi2 - r2;
        const double ca = ci * (qi * qi);                     // nsc:592
        a.rho += fmax(ms, 0.0) * W;                           // nsc:605
        a.rd += fmax(-ms, 0.0) * W;                           // nsc:606
        const double nww = Nw * W;
        a.n += nww;                                           // nsc:607
        if (WOUT) wout[u] = nww;                              // nsc:626
        // nsc:615, the pair's common factor taken out of the three components: (A_j g_b + A_i g_a) / 2 = t (dx, dy, dz) with
        // t = (A_j c_b + A_i c_a) / 2 - 10 fp64 operations instead of 21, each a 4-cycle issue; a regrouping of
        // the reference's products (a few ulp per term against a bound of 1e-12 x sum|term|), the same in every variant
        const double tg = (Aj * cb + Ai * ca) * 0.5;
        a.gx += tg * dx;
        a.gy += tg * dy;
        a.gz += tg * dz;
    }
}

// ---- pass 1: rho, rho_dust, n, grad P        nsc:588-619 --------------------------------------
// EXP != 0: timing experiments on an extra, discarded launch (SPHX_BLOB_EXP): 1 = staging only,
// 2 = neighbour loop only (image not filled), 3 = both but no global stores at the end.
template <int EXP>
__global__ __launch_bounds__(PASS_T, PASS_MINW) void blob_density_kernel(int n, int npad, int k, int nblk, int clip,
                                                              const int* __restrict__ nbr,
                                                              const u16* __restrict__ slot16,
                                                              const int* __restrict__ uniq,
                                                              const int* __restrict__ qorder,
                                                              const int* __restrict__ omap, int n_active,
                                                              const RecA* __restrict__ rec, double* rho_s,
                                                              double* rho, double* rhod, double* nden, double* G,
                                                              double* ha, BlobSel sel) {
    extern __shared__ double2 img[];                       // 4 * BLOB_S chunks, then the slot tile
    u16* tile = reinterpret_cast<u16*>(img + 4 * BLOB_S);
    const int t = threadIdx.x / LPP, half = threadIdx.x & (LPP - 1);     // half: which partial sum
    // persistent workgroups (two per CU): blob after blob, no dispatch gap between them
    const int nsel = blob_sel_count(sel, nblk);
    for (int bi = blockIdx.x; bi < nsel; bi += gridDim.x) {
        const int b = blob_sel_at(sel, bi, nsel);
        const int p = b * BLOB_P + t;
        const int i = (p < n) ? qorder[p] : 0;
        if (EXP != 2)
            stage<0>(img, nullptr, tile, rec, nullptr, 0, nullptr, 0, uniq + (size_t)b * BLOB_S, slot16, npad, k, b);
        else
            for (int q = threadIdx.x; q < KPAD(k) * BLOB_P; q += PASS_T) tile[q] = (u16)((q * 37 + b) % BLOB_S);
        const double* self = reinterpret_cast<const double*>(&rec[i]);
        const Q4 s0 = gload4(self), s1 = gload4(self + 4);     // x y z h2 | c1 ms A Nw
        // outputs go to the caller's index o (device API: ghosts, o >= n_active, are candidates only)
        const int o = (p < n) ? (omap ? omap[i] : i) : 0x7FFFFFFF;
        __syncthreads();
        if (o < n_active && EXP == 1 && s0.a == 1.2345e-300) rho[o] = img[threadIdx.x].x;
        if (o < n_active && EXP != 1) {
            double xr = s0.a, yr = s0.b, zr = s0.c;
            {
                const unsigned sl0 = tile[t];
                if (sl0 < SLOT_OVER) { const Q4 r = lload4(img, (int)sl0, 0); xr = r.a; yr = r.b; zr = r.c; }
                else if (sl0 == SLOT_OVER) { const int j0 = nbr[p]; xr = rec[j0].x; yr = rec[j0].y; zr = rec[j0].z; }
            }
            const double hi2 = s0.d, ci = -6.0 * s1.a, Ai = s1.c;
            DensAcc a{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            const int nm = KPAD(k) / LPP;
            unsigned sl[NB];
            load_slots(sl, tile, 0, half, t);
            for (int m0 = 0; m0 < nm; m0 += NB) {
                unsigned cur[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) cur[u] = sl[u];
                if (m0 + NB < nm) load_slots(sl, tile, m0 + NB, half, t);    // next batch's slots, behind this one's reads
                const size_t col0 = (size_t)(LPP * m0 + half) * npad + p;
                const bool fast = all_staged(cur);
                if (fast && !clip) density_batch<true, false>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai);
                else if (fast) density_batch<true, true>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai);
                else if (!clip) density_batch<false, false>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai);
                else density_batch<false, true>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai);
            }
            const double s_rho = group_total(a.rho), s_rd = group_total(a.rd), s_n = group_total(a.n);
            const double gx = group_total(a.gx), gy = group_total(a.gy), gz = group_total(a.gz);
            if (EXP == 3) {
                if (s_rho + gx + s_n + s_rd + gy + gz == 1.2345e-300) rho[o] = 0.0;
            } else if (!half) {
                rho[o] = s_rho; rhod[o] = s_rd; nden[o] = s_n;
                rho_s[i] = s_rho;                                     // storage order: staged by pass 2
                if (G) { G[3 * (size_t)o + 0] = -gx; G[3 * (size_t)o + 1] = -gy; G[3 * (size_t)o + 2] = -gz; }
                ha[3 * (size_t)o + 0] = -gx / s_rho;                  // nsc:619
                ha[3 * (size_t)o + 1] = -gy / s_rho;
                ha[3 * (size_t)o + 2] = -gz / s_rho;
            }
        }
        __syncthreads();                                   // the image is rewritten by the next blob
    }
}

// ---- pass 2: Pi_i, crossing time             nsc:639-649, nsc:776-786 --------------------------
template <bool FAST>
__device__ __forceinline__ void pi_batch(double& s_pi, double& maxrel, const unsigned (&sl)[NB], const double2* img,
                                         const double* lrho, const RecB* __restrict__ recb,
                                         const double* __restrict__ rho_s, const int* __restrict__ nbr, size_t col0,
                                         size_t colstep, const Q4& r0, const Q4& rv, double rho_i, double cs_i) {
    Q4 q0b[NB], qvb[NB];
    double rhob[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (FAST || sl[u] < SLOT_OVER) {
            q0b[u] = lload4(img, (int)sl[u], 0); qvb[u] = lload4(img, (int)sl[u], 1);
            rhob[u] = lrho[sl[u]];
        } else if (sl[u] == SLOT_OVER) {
            const int jj = nbr[col0 + u * colstep];
            const double* qb = reinterpret_cast<const double*>(&recb[jj]);
            q0b[u] = gload4(qb); qvb[u] = gload4(qb + 4);
            rhob[u] = rho_s[jj];
        } else { q0b[u] = r0; qvb[u] = rv; rhob[u] = rho_i; }
    }
    if (FAST) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (!FAST && sl[u] == SLOT_NONE) continue;
        const Q4 q0 = q0b[u], qv = qvb[u];
        const double rho_j = rhob[u];
        const double dx = q0.a - r0.a, dy = q0.b - r0.b, dz = q0.c - r0.c;
        const double dvx = qv.a - rv.a, dvy = qv.b - rv.b, dvz = qv.c - rv.c;
        const double r2 = dx * dx + dy * dy + dz * dz;
        const double dot = dvx * dx + dvy * dy + dvz * dz;
        double w = dot / sqrt_mid(r2 + 0.01 * q0.d);                    // nsc:643
        w = (w > 0.0) ? 0.0 : w;                                        // nsc:644
        const double rho_ab = (rho_j + rho_i) / 2.0;                    // nsc:646
        const double c_ab = 0.5 * (qv.d + cs_i);                        // nsc:647
        s_pi += -0.5 * (c_ab * 2.0 - 3.0 * w) * w / rho_ab;             // nsc:649
        maxrel = fmax(maxrel, dvx * dvx + dvy * dvy + dvz * dvz);       // nsc:780
    }
}

__global__ __launch_bounds__(PASS_T, PASS_MINW) void blob_pi_kernel(int n, int npad, int k, int nblk,
                                                         const int* __restrict__ nbr,
                                                         const u16* __restrict__ slot16,
                                                         const int* __restrict__ uniq,
                                                         const int* __restrict__ qorder,
                                                         const int* __restrict__ omap, int n_active,
                                                         const RecB* __restrict__ recb,
                                                         const double* __restrict__ rho_s,
                                                         const RecSelf* __restrict__ selfr, RecBC* bc, double* Pi,
                                                         double* BwOut, u64* ct_bits, BlobSel sel) {
    extern __shared__ double2 img[];                       // 4 * BLOB_S chunks, BLOB_S doubles, slot tile
    __shared__ u64 sm[PASS_T / 64];
    double* lrho = reinterpret_cast<double*>(img + 4 * BLOB_S);
    u16* tile = reinterpret_cast<u16*>(lrho + BLOB_S);
    const int t = threadIdx.x / LPP, half = threadIdx.x & (LPP - 1);     // half: which partial sum
    u64 my_ct = 0x7FF0000000000000ull;       // +inf: "no crossing time"
    const int nsel = blob_sel_count(sel, nblk);
    for (int bi = blockIdx.x; bi < nsel; bi += gridDim.x) {
        const int b = blob_sel_at(sel, bi, nsel);
        const int p = b * BLOB_P + t;
        const int i = (p < n) ? qorder[p] : 0;
        stage<1>(img, lrho, tile, recb, rho_s, 1, nullptr, 0, uniq + (size_t)b * BLOB_S, slot16, npad, k, b);
        const RecSelf sf = selfr[i];
        const double rho_i = rho_s[i];
        const double* selfq = reinterpret_cast<const double*>(&recb[i]);
        const Q4 self0 = gload4(selfq), selfv = gload4(selfq + 4);
        const int o = (p < n) ? (omap ? omap[i] : i) : 0x7FFFFFFF;
        __syncthreads();
        if (o < n_active) {
            Q4 r0 = self0, rv = selfv;
            {
                const unsigned sl0 = tile[t];
                if (sl0 < SLOT_OVER) { r0 = lload4(img, (int)sl0, 0); rv = lload4(img, (int)sl0, 1); }
                else if (sl0 == SLOT_OVER) {
                    const double* rq = reinterpret_cast<const double*>(&recb[nbr[p]]);
                    r0 = gload4(rq); rv = gload4(rq + 4);
                }
            }
            const double cs_i = sf.csi, ms_i = sf.mg, h_i = sf.h;
            double s_pi = 0.0, maxrel = 0.0;
            const int nm = KPAD(k) / LPP;
            unsigned sl[NB];
            load_slots(sl, tile, 0, half, t);
            for (int m0 = 0; m0 < nm; m0 += NB) {
                unsigned cur[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) cur[u] = sl[u];
                if (m0 + NB < nm) load_slots(sl, tile, m0 + NB, half, t);
                const size_t col0 = (size_t)(LPP * m0 + half) * npad + p;
                if (all_staged(cur))
                    pi_batch<true>(s_pi, maxrel, cur, img, lrho, recb, rho_s, nbr, col0, LPP * (size_t)npad, r0, rv, rho_i, cs_i);
                else
                    pi_batch<false>(s_pi, maxrel, cur, img, lrho, recb, rho_s, nbr, col0, LPP * (size_t)npad, r0, rv, rho_i, cs_i);
            }
            s_pi = group_total(s_pi);
            maxrel = group_max(maxrel);
            if (!half) {
                Pi[o] = s_pi;
                const double bw = fmax(ms_i, 0.0) * s_pi;                       // m Pi [t==0]  nsc:651
                bc[i].Bw = bw;
                if (BwOut) BwOut[o] = bw;
                if (ms_i > 0.0) {                                               // gas only     nsc:782
                    double ct = h_i / sqrt(maxrel);
                    if (ct != ct) ct = 0.0;                                     // nan_to_num
                    if (ct > DBL_MAX) ct = DBL_MAX;
                    if (ct > 0.0) { const u64 cb = (u64)__double_as_longlong(ct); my_ct = cb < my_ct ? cb : my_ct; }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 q = __shfl_xor(my_ct, o, 64);
        my_ct = q < my_ct ? q : my_ct;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = my_ct;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 r = sm[0];
        for (int w = 1; w < PASS_T / 64; ++w) r = sm[w] < r ? sm[w] : r;
        if (r != 0x7FF0000000000000ull) atomicMin(ct_bits, r);
    }
}

// ---- pass 3: viscous acceleration + heat      nsc:651-654 --------------------------------------
struct ViscAcc { double x, y, z, h; };
template <bool FAST, bool CLIP>
__device__ __forceinline__ void visc_batch(ViscAcc& a, const unsigned (&sl)[NB], const double2* img,
                                           const double* lc1, const RecB* __restrict__ recb,
                                           const RecBC* __restrict__ bc, const int* __restrict__ nbr, size_t col0,
                                           size_t colstep, const Q4& r0, const Q4& rv, double hi2, double ci,
                                           double Bi) {
    Q4 q0b[NB], qvb[NB];
    double c1b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (FAST || sl[u] < SLOT_OVER) {
            q0b[u] = lload4(img, (int)sl[u], 0); qvb[u] = lload4(img, (int)sl[u], 1);     // qv.d = Bw_j
            c1b[u] = lc1[sl[u]];
        } else if (sl[u] == SLOT_OVER) {
            const int jj = nbr[col0 + u * colstep];
            const double* qb = reinterpret_cast<const double*>(&recb[jj]);
            q0b[u] = gload4(qb); qvb[u] = gload4(qb + 4);
            const double2 tt = *reinterpret_cast<const double2*>(&bc[jj]);
            qvb[u].d = tt.x; c1b[u] = tt.y;
        } else { q0b[u] = r0; qvb[u] = rv; c1b[u] = 0.0; }
    }
    if (FAST) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (!FAST && sl[u] == SLOT_NONE) continue;
        const Q4 q0 = q0b[u], qv = qvb[u];
        const double c1 = c1b[u], Bj = qv.d;
        const double dx = q0.a - r0.a, dy = q0.b - r0.b, dz = q0.c - r0.c;
        const double r = sqrt_mid(dx * dx + dy * dy + dz * dz);
        const double r2 = r * r;
        const double qj = q0.d - r2, qi = hi2 - r2;
        const double cb = (CLIP && !(qj > 0.0)) ? 0.0 : -6.0 * c1 * (qj * qj);
        const double ca = ci * (qi * qi);
        const double tb = (Bj * cb + Bi * ca) / 2.0;                          // nsc:651, the common factor taken out (see pass 1)
        const double bx = tb * dx, by = tb * dy, bz = tb * dz;
        a.x += bx; a.y += by; a.z += bz;
        a.h += bx * (qv.a - rv.a) + by * (qv.b - rv.b) + bz * (qv.c - rv.c);   // nsc:653
    }
}

__global__ __launch_bounds__(PASS_T, PASS_MINW) void blob_visc_kernel(int n, int npad, int k, int nblk, int clip,
                                                           const int* __restrict__ nbr,
                                                           const u16* __restrict__ slot16,
                                                           const int* __restrict__ uniq,
                                                           const int* __restrict__ qorder,
                                                           const int* __restrict__ omap, int n_active,
                                                           const RecB* __restrict__ recb,
                                                           const RecBC* __restrict__ bc,
                                                           const double* __restrict__ m, double* va, double* vh,
                                                           BlobSel sel) {
    extern __shared__ double2 img[];                       // 4 * BLOB_S chunks {x y | z h2 | vx vy | vz Bw}, c1, tile
    double* lc1 = reinterpret_cast<double*>(img + 4 * BLOB_S);
    u16* tile = reinterpret_cast<u16*>(lc1 + BLOB_S);
    const int t = threadIdx.x / LPP, half = threadIdx.x & (LPP - 1);     // half: which partial sum
    const double* bcd = reinterpret_cast<const double*>(bc);
    const int nsel = blob_sel_count(sel, nblk);
    for (int bi = blockIdx.x; bi < nsel; bi += gridDim.x) {
        const int b = blob_sel_at(sel, bi, nsel);
        const int p = b * BLOB_P + t;
        const int i = (p < n) ? qorder[p] : 0;
        stage<2>(img, lc1, tile, recb, bcd, 2, bcd + 1, 2, uniq + (size_t)b * BLOB_S, slot16, npad, k, b);
        const double2 bci = *reinterpret_cast<const double2*>(&bc[i]);       // Bw, c1
        const double* selfq = reinterpret_cast<const double*>(&recb[i]);
        const Q4 self0 = gload4(selfq), selfv = gload4(selfq + 4);
        const int o = (p < n) ? (omap ? omap[i] : i) : 0x7FFFFFFF;
        const double mi = (o < n_active) ? m[o] : 0.0;                      // m in output order
        __syncthreads();
        if (o < n_active) {
            Q4 r0 = self0, rv = selfv;
            {
                const unsigned sl0 = tile[t];
                if (sl0 < SLOT_OVER) { r0 = lload4(img, (int)sl0, 0); rv = lload4(img, (int)sl0, 1); }
                else if (sl0 == SLOT_OVER) {
                    const double* rq = reinterpret_cast<const double*>(&recb[nbr[p]]);
                    r0 = gload4(rq); rv = gload4(rq + 4);
                }
            }
            const double hi2 = self0.d, ci = -6.0 * bci.y, Bi = bci.x;
            ViscAcc a{0.0, 0.0, 0.0, 0.0};
            const int nm = KPAD(k) / LPP;
            unsigned sl[NB];
            load_slots(sl, tile, 0, half, t);
            for (int m0 = 0; m0 < nm; m0 += NB) {
                unsigned cur[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) cur[u] = sl[u];
                if (m0 + NB < nm) load_slots(sl, tile, m0 + NB, half, t);
                const size_t col0 = (size_t)(LPP * m0 + half) * npad + p;
                const bool fast = all_staged(cur);
                if (fast && !clip) visc_batch<true, false>(a, cur, img, lc1, recb, bc, nbr, col0, LPP * (size_t)npad, r0, rv, hi2, ci, Bi);
                else if (fast) visc_batch<true, true>(a, cur, img, lc1, recb, bc, nbr, col0, LPP * (size_t)npad, r0, rv, hi2, ci, Bi);
                else if (!clip) visc_batch<false, false>(a, cur, img, lc1, recb, bc, nbr, col0, LPP * (size_t)npad, r0, rv, hi2, ci, Bi);
                else visc_batch<false, true>(a, cur, img, lc1, recb, bc, nbr, col0, LPP * (size_t)npad, r0, rv, hi2, ci, Bi);
            }
            const double ax = group_total(a.x), ay = group_total(a.y), az = group_total(a.z), heat = group_total(a.h);
            if (!half) {
                va[3 * (size_t)o + 0] = -ax; va[3 * (size_t)o + 1] = -ay; va[3 * (size_t)o + 2] = -az;
                vh[o] = heat * mi / 2.0;                                        // nsc:654
            }
        }
        __syncthreads();
    }
}

// ---- pass 1 + species pass in one kernel (the step loop with a composition, hydro_update's sums) -----------------------
// The species pass's first sweep repeats pass 1's staging and its W_ij: here pass 1 keeps every lane's weights Nw_j W_ij
// in registers and the two composition sweeps of blob_species_kernel follow on the same slot lists - one staging and one
// set of kernel evaluations fewer per blob (with-species step 1.65 -> see DESIGN 6.1).  Same expressions in the same
// order as the two kernels run one after the other: bit-identical outputs.
template <int SPEC_MAXM>
__global__ __launch_bounds__(PASS_T, PASS_MINW) void blob_density_species_kernel(int n, int npad, int k, int nblk, int clip,
        const int* __restrict__ nbr, const u16* __restrict__ slot16, const int* __restrict__ uniq,
        const int* __restrict__ qorder, const RecA* __restrict__ rec, double* rho_s, double* rho, double* rhod,
        double* nden, double* G, double* ha,
        int S, const double* __restrict__ fun, const int* __restrict__ row_of, const double* __restrict__ m, AgbTable agb,
        int agb_on, double* F, double* Zout, double* agb_out) {
    extern __shared__ double2 img[];                       // 4 * BLOB_S chunks, then the slot tile
    u16* tile = reinterpret_cast<u16*>(img + 4 * BLOB_S);
    const int t = threadIdx.x / LPP, half = threadIdx.x & (LPP - 1);
    const int nm = KPAD(k) / LPP;
    for (int bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const int b = xcd_block(bi, nblk);
        const int p = b * BLOB_P + t;
        const bool live = p < n;
        const int i = live ? qorder[p] : 0;
        const int* uq = uniq + (size_t)b * BLOB_S;
        stage<0>(img, nullptr, tile, rec, nullptr, 0, nullptr, 0, uq, slot16, npad, k, b);
        const double* self = reinterpret_cast<const double*>(&rec[i]);
        const Q4 s0 = gload4(self), s1 = gload4(self + 4);     // x y z h2 | c1 ms A Nw
        __syncthreads();
        double w[SPEC_MAXM];
#pragma unroll
        for (int mm = 0; mm < SPEC_MAXM; ++mm) w[mm] = 0.0;
        if (live) {
            double xr = s0.a, yr = s0.b, zr = s0.c;
            {
                const unsigned sl0 = tile[t];
                if (sl0 < SLOT_OVER) { const Q4 r = lload4(img, (int)sl0, 0); xr = r.a; yr = r.b; zr = r.c; }
                else if (sl0 == SLOT_OVER) { const int j0 = nbr[p]; xr = rec[j0].x; yr = rec[j0].y; zr = rec[j0].z; }
            }
            const double hi2 = s0.d, ci = -6.0 * s1.a, Ai = s1.c;
            DensAcc a{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m0 = 0; m0 < SPEC_MAXM; m0 += NB) {
                if (m0 < nm) {
                    unsigned cur[NB];
                    load_slots(cur, tile, m0, half, t);
                    const size_t col0 = (size_t)(LPP * m0 + half) * npad + p;
                    const bool fast = all_staged(cur);
                    double wo[NB];
                    if (fast && !clip) density_batch<true, false, true>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai, wo);
                    else if (fast) density_batch<true, true, true>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai, wo);
                    else if (!clip) density_batch<false, false, true>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai, wo);
                    else density_batch<false, true, true>(a, cur, img, rec, nbr, col0, LPP * (size_t)npad, xr, yr, zr, hi2, ci, Ai, wo);
#pragma unroll
                    for (int u = 0; u < NB; ++u) if (m0 + u < SPEC_MAXM) w[m0 + u] = wo[u];
                }
            }
            const double s_rho = group_total(a.rho), s_rd = group_total(a.rd), s_n = group_total(a.n);
            const double gx = group_total(a.gx), gy = group_total(a.gy), gz = group_total(a.gz);
            if (!half) {
                rho[i] = s_rho; rhod[i] = s_rd; nden[i] = s_n;
                rho_s[i] = s_rho;
                if (G) { G[3 * (size_t)i + 0] = -gx; G[3 * (size_t)i + 1] = -gy; G[3 * (size_t)i + 2] = -gz; }
                ha[3 * (size_t)i + 0] = -gx / s_rho;                  // nsc:619
                ha[3 * (size_t)i + 1] = -gy / s_rho;
                ha[3 * (size_t)i + 2] = -gz / s_rho;
            }
        }
        // ---- the composition sweeps of blob_species_kernel, on the weights in hand
        double tot[16];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            __syncthreads();                                   // everybody is done with the previous image
            {
                int ju[NSTAGE];
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int s = threadIdx.x + r * PASS_T;
                    ju[r] = (s < BLOB_S) ? uq[s] : -1;
                }
                double2 c[NSTAGE][4];
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int jr = ju[r] < 0 ? 0 : ju[r];
                    const double2* g = reinterpret_cast<const double2*>(fun + (size_t)(row_of ? row_of[jr] : jr) * 16) + 4 * hh;
                    c[r][0] = g[0]; c[r][1] = g[1]; c[r][2] = g[2]; c[r][3] = g[3];
                }
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int s = threadIdx.x + r * PASS_T;
                    if (ju[r] >= 0) {
                        img[0 * BLOB_S + s] = c[r][0]; img[1 * BLOB_S + s] = c[r][1];
                        img[2 * BLOB_S + s] = c[r][2]; img[3 * BLOB_S + s] = c[r][3];
                    }
                }
            }
            __syncthreads();
            double acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.0;
#pragma unroll
            for (int mm = 0; mm < SPEC_MAXM; ++mm) {
                if (mm < nm) {
                    const unsigned sl = tile[(LPP * mm + half) * BLOB_P + t];
                    if (sl != SLOT_NONE && live) {
                        Q4 f0, f1;
                        if (sl < SLOT_OVER) { f0 = lload4(img, (int)sl, 0); f1 = lload4(img, (int)sl, 1); }
                        else {
                            const int jo = nbr[(size_t)(LPP * mm + half) * npad + p];
                            const double* q = fun + (size_t)(row_of ? row_of[jo] : jo) * 16 + 8 * hh;
                            f0 = gload4(q); f1 = gload4(q + 4);
                        }
                        const double wm = w[mm];
                        acc[0] += wm * f0.a; acc[1] += wm * f0.b; acc[2] += wm * f0.c; acc[3] += wm * f0.d;
                        acc[4] += wm * f1.a; acc[5] += wm * f1.b; acc[6] += wm * f1.c; acc[7] += wm * f1.d;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) tot[8 * hh + q] = group_total(acc[q]);
        }
        if (live && !half) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < S) F[(size_t)q * n + i] = tot[q];
        }
        if (live && agb_on) {
            double heavy = 0.0, all = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q < S) {
                    const double ww = tot[q] * agb.mu[q];
                    all += ww;
                    if (q >= 6) heavy += ww;
                }
            }
            const double Z = heavy / all;                      // drv:663
            if (!half) Zout[i] = Z;
            const double Mi = m[i];
            double* row = agb_out + (size_t)i * S;
            for (int q = half; q < S; q += LPP)
                if (!((agb.covered >> q) & 1u)) row[q] = 0.0;
            for (int o = half; o < agb.nspl; o += LPP) {
                double val;
                const int target = agb_one_spline(agb, o, Mi, Z, val);
                if (target >= 0) row[target] = val;
            }
        }
        __syncthreads();                                       // the image is rewritten by the next blob
    }
}

int sphx_blob_density_species(sphx_ctx* ctx, int64_t n, int k, int S, const double* fun, const int* row_of, const double* m_sorted,
                              double* F, double* Z, double* agb, int agb_on) {
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_density_species_kernel<10>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(64, SPHX_MAX_K)));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_density_species_kernel<16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(64, SPHX_MAX_K)));
        attr = true;
    }
#define DS_ARGS (int)n, (int)npad, k, nblk, ctx->clip_grad, ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(), ctx->qorder, \
                ctx->rec1.as<RecA>(), ctx->rho_s.as<double>(), ctx->rho.as<double>(), ctx->rhod.as<double>(), ctx->nden.as<double>(), \
                ctx->lean_outputs ? nullptr : ctx->G.as<double>(), ctx->ha.as<double>(), S, fun, row_of, m_sorted, ctx->agb, agb_on, F, Z, agb
    if (KPAD(k) / LPP <= 10)
        hipLaunchKernelGGL(blob_density_species_kernel<10>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(64, k), ctx->stream, DS_ARGS);
    else
        hipLaunchKernelGGL(blob_density_species_kernel<16>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(64, k), ctx->stream, DS_ARGS);
#undef DS_ARGS
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- species pass (nsc:624-627) out of LDS, + metallicity and AGB yields ---------------------------
// F[s,i] = sum_k Nw_j W_ij f_un[j,s] needs, per neighbour, the 64-B record (for W) AND the 128-B composition row: 184 KB
// for a blob's 960 slots, more than a CU has.  So the blob is walked three times over the same slot lists: (1) the
// records are staged and every lane turns its list positions into weights Nw_j W_ij kept in REGISTERS (K/4 doubles);
// (2), (3) the image is refilled with the lower / upper 64 bytes of the distinct neighbours' composition rows and the
// lane accumulates weight x row.  Each distinct row is fetched once per blob (~5 per particle) instead of once per
// reference (40 per particle: the gather form, sphx_sums.hip, bound by exactly that: 1.77 ms at 1e6 particles).
// Sums: the lane's positions k = q mod 4 in ascending k, then (p0 + p1) + (p2 + p3) as in the other LDS passes.
// SPEC_MAXM: list positions per lane the registers are sized for (K <= 40: 10; else 16)
template <int SPEC_MAXM>
__global__ __launch_bounds__(PASS_T, PASS_MINW) void blob_species_kernel(int n, int npad, int k, int nblk, int S,
                                                              const int* __restrict__ nbr,
                                                              const u16* __restrict__ slot16,
                                                              const int* __restrict__ uniq,
                                                              const int* __restrict__ qorder,
                                                              const RecA* __restrict__ rec,
                                                              const double* __restrict__ fun,       // rows of 16 doubles
                                                              const int* __restrict__ row_of,       // (nullable) row of particle j
                                                              const double* __restrict__ m, AgbTable agb, int agb_on,
                                                              double* F, double* Zout, double* agb_out) {
    extern __shared__ double2 img[];                       // 4 * BLOB_S chunks, then the slot tile
    u16* tile = reinterpret_cast<u16*>(img + 4 * BLOB_S);
    const int t = threadIdx.x / LPP, half = threadIdx.x & (LPP - 1);
    const int nm = KPAD(k) / LPP;
    for (int bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const int b = xcd_block(bi, nblk);
        const int p = b * BLOB_P + t;
        const bool live = p < n;
        const int i = live ? qorder[p] : 0;
        const int* uq = uniq + (size_t)b * BLOB_S;
        stage<0>(img, nullptr, tile, rec, nullptr, 0, nullptr, 0, uq, slot16, npad, k, b);
        __syncthreads();
        // ---- (1) weights of this lane's list positions
        double w[SPEC_MAXM];
        {
            double xr, yr, zr;
            {
                const int j0 = live ? nbr[p] : 0;                   // deltas are relative to the first neighbour (nsc:580-581)
                const unsigned sl0 = tile[t];
                if (sl0 < SLOT_OVER) { const Q4 r = lload4(img, (int)sl0, 0); xr = r.a; yr = r.b; zr = r.c; }
                else { const int jj = j0 < 0 ? i : j0; xr = rec[jj].x; yr = rec[jj].y; zr = rec[jj].z; }
            }
#pragma unroll
            for (int mm = 0; mm < SPEC_MAXM; ++mm) {
                w[mm] = 0.0;
                if (mm < nm) {
                    const unsigned sl = tile[(LPP * mm + half) * BLOB_P + t];
                    if (sl != SLOT_NONE && live) {
                        Q4 q0, q1;
                        if (sl < SLOT_OVER) { q0 = lload4(img, (int)sl, 0); q1 = lload4(img, (int)sl, 1); }
                        else {
                            const double* q = reinterpret_cast<const double*>(&rec[nbr[(size_t)(LPP * mm + half) * npad + p]]);
                            q0 = gload4(q); q1 = gload4(q + 4);
                        }
                        const double dx = q0.a - xr, dy = q0.b - yr, dz = q0.c - zr;
                        const double r = sqrt_mid(dx * dx + dy * dy + dz * dz);
                        const double qj = q0.d - r * r;
                        double W = q1.a * (qj * qj * qj);
                        W = (W < 0.0) ? 0.0 : W;
                        w[mm] = q1.d * W;
                    }
                }
            }
        }
        double tot[16];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            __syncthreads();                                   // everybody is done with the previous image
            // ---- (2), (3): this half of the distinct neighbours' composition rows into the image
            {
                int ju[NSTAGE];
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int s = threadIdx.x + r * PASS_T;
                    ju[r] = (s < BLOB_S) ? uq[s] : -1;
                }
                double2 c[NSTAGE][4];
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int jr = ju[r] < 0 ? 0 : ju[r];
                    const double2* g = reinterpret_cast<const double2*>(fun + (size_t)(row_of ? row_of[jr] : jr) * 16) + 4 * hh;
                    c[r][0] = g[0]; c[r][1] = g[1]; c[r][2] = g[2]; c[r][3] = g[3];
                }
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int s = threadIdx.x + r * PASS_T;
                    if (ju[r] >= 0) {
                        img[0 * BLOB_S + s] = c[r][0]; img[1 * BLOB_S + s] = c[r][1];
                        img[2 * BLOB_S + s] = c[r][2]; img[3 * BLOB_S + s] = c[r][3];
                    }
                }
            }
            __syncthreads();
            double acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.0;
#pragma unroll
            for (int mm = 0; mm < SPEC_MAXM; ++mm) {
                if (mm < nm) {
                    const unsigned sl = tile[(LPP * mm + half) * BLOB_P + t];
                    if (sl != SLOT_NONE && live) {
                        Q4 f0, f1;
                        if (sl < SLOT_OVER) { f0 = lload4(img, (int)sl, 0); f1 = lload4(img, (int)sl, 1); }
                        else {
                            const int jo = nbr[(size_t)(LPP * mm + half) * npad + p];
                            const double* q = fun + (size_t)(row_of ? row_of[jo] : jo) * 16 + 8 * hh;
                            f0 = gload4(q); f1 = gload4(q + 4);
                        }
                        const double wm = w[mm];
                        acc[0] += wm * f0.a; acc[1] += wm * f0.b; acc[2] += wm * f0.c; acc[3] += wm * f0.d;
                        acc[4] += wm * f1.a; acc[5] += wm * f1.b; acc[6] += wm * f1.c; acc[7] += wm * f1.d;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) tot[8 * hh + q] = group_total(acc[q]);
        }
        if (live && !half) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < S) F[(size_t)q * n + i] = tot[q];
        }
        if (live && agb_on) {
            // every lane of the group holds the totals: the metallicity in all four, the splines dealt over them
            double heavy = 0.0, all = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q < S) {
                    const double ww = tot[q] * agb.mu[q];
                    all += ww;
                    if (q >= 6) heavy += ww;
                }
            }
            const double Z = heavy / all;                      // drv:663
            if (!half) Zout[i] = Z;
            const double Mi = m[i];
            double* row = agb_out + (size_t)i * S;
            for (int q = half; q < S; q += LPP)
                if (!((agb.covered >> q) & 1u)) row[q] = 0.0;  // species no spline writes
            for (int o = half; o < agb.nspl; o += LPP) {
                double val;
                const int target = agb_one_spline(agb, o, Mi, Z, val);
                if (target >= 0) row[target] = val;
            }
        }
        __syncthreads();                                       // the image is rewritten by the next blob
    }
}

int sphx_blob_species(sphx_ctx* ctx, int64_t n, int k, int S, const double* fun, const int* row_of, const double* m_sorted, double* F,
                      double* Z, double* agb, int agb_on) {
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
    static bool attr = false;
    if (!attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_species_kernel<10>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(64, SPHX_MAX_K)));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_species_kernel<16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(64, SPHX_MAX_K)));
        attr = true;
    }
    if (KPAD(k) / LPP <= 10)
        hipLaunchKernelGGL(blob_species_kernel<10>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(64, k), ctx->stream, (int)n,
                           (int)npad, k, nblk, S, ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(), ctx->qorder,
                           ctx->rec1.as<RecA>(), fun, row_of, m_sorted, ctx->agb, agb_on, F, Z, agb);
    else
        hipLaunchKernelGGL(blob_species_kernel<16>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(64, k), ctx->stream, (int)n,
                           (int)npad, k, nblk, S, ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(), ctx->qorder,
                           ctx->rec1.as<RecA>(), fun, row_of, m_sorted, ctx->agb, agb_on, F, Z, agb);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- launchers (buffers are sized by the callers in sphx_sums.hip) -----------------------------
// persistent grid: two workgroups per CU (what the LDS image allows), a multiple of the 8 XCDs
// ---- gas-dust drag out of LDS                 nsc:719-742 (net_impulse; gather form: sphx_sums.hip) ---------------------
// Only dust neighbours count (a tenth of the references in the two-phase cloud), and every one of them costs the
// gather form a chain of gathers (type, record, m, grain mass, cross-section) and an atomic with return for its place
// in the receiver's slice of the ordered scatter.  Here the blob's distinct neighbours carry a dust flag in LDS, the
// references to each are counted in LDS, and the blob reserves its share of a receiver's slice with ONE global atomic
// per distinct dust neighbour (ranks inside the share come from an LDS counter; the order within a slice is free: the
// reduction sorts by key).  FILL = false: the counting pass of the scatter plan (sphx_drag_scatter_plan).
#define DRAG_XLDS (BLOB_S * (2 * sizeof(int) + 1))          // count, base, dust flag per image slot
template <bool FILL>
__global__ __launch_bounds__(PASS_T, PASS_MINW) void blob_drag_kernel(int n, int npad, int k, int nblk,
                                                           const int* __restrict__ nbr,
                                                           const u16* __restrict__ slot16,
                                                           const int* __restrict__ uniq,
                                                           const int* __restrict__ qorder,
                                                           const RecB* __restrict__ recb,
                                                           const double* __restrict__ m,
                                                           const double* __restrict__ ptype,
                                                           const double* __restrict__ mgm,
                                                           const double* __restrict__ mcs,
                                                           const int* __restrict__ id, double* onto, DragScatter sc) {
    extern __shared__ double2 img[];                       // FILL: 4 * BLOB_S chunks; then the slot tile, counts, bases, flags
    u16* tile = reinterpret_cast<u16*>(img + (FILL ? 4 * BLOB_S : 0));
    int* cntL = reinterpret_cast<int*>(tile + KPAD(k) * BLOB_P);
    int* baseL = cntL + BLOB_S;
    unsigned char* dust = reinterpret_cast<unsigned char*>(baseL + BLOB_S);
    const int t = threadIdx.x / LPP, part = threadIdx.x & (LPP - 1);
    const int nm = KPAD(k) / LPP;
    for (int bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
        const int b = xcd_block(bi, nblk);
        const int p = b * BLOB_P + t;
        const bool live = p < n;
        const int i = live ? qorder[p] : 0;
        const int* uq = uniq + (size_t)b * BLOB_S;
        if (FILL) {
            stage<0>(img, nullptr, tile, recb, nullptr, 0, nullptr, 0, uq, slot16, npad, k, b);
        } else {
            const int pieces = KPAD(k) * (BLOB_P / 8);
            for (int q = threadIdx.x; q < pieces; q += PASS_T) {
                const int kk = q / (BLOB_P / 8), c = q % (BLOB_P / 8);
                uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                if (kk < k) v = *reinterpret_cast<const uint4*>(slot16 + (size_t)kk * npad + (size_t)b * BLOB_P + c * 8);
                *reinterpret_cast<uint4*>(tile + kk * BLOB_P + c * 8) = v;
            }
        }
        for (int s = threadIdx.x; s < BLOB_S; s += PASS_T) {
            const int j = uq[s];
            dust[s] = (j >= 0 && ptype[j] == 2.0) ? 1 : 0;                       // nsc:736
            cntL[s] = 0;
        }
        __syncthreads();
        // references per distinct dust neighbour (a particle's reference to itself casts no reaction: nsc:741)
        if (live) {
            for (int m0 = 0; m0 < nm; ++m0) {
                const int kk = LPP * m0 + part;
                const unsigned sl = tile[kk * BLOB_P + t];
                if (sl < SLOT_OVER) {
                    if (dust[sl] && uq[sl] != i) atomicAdd(&cntL[sl], 1);
                } else if (sl == SLOT_OVER && !FILL) {
                    const int j = nbr[(size_t)kk * npad + p];
                    if (j != i && ptype[j] == 2.0) atomicAdd(&sc.cnt[j], 1);
                }
            }
        }
        __syncthreads();
        if (!FILL) {
            for (int s = threadIdx.x; s < BLOB_S; s += PASS_T)
                if (cntL[s]) atomicAdd(&sc.cnt[uq[s]], cntL[s]);
            __syncthreads();
            continue;
        }
        // the blob's share of every receiver's slice (handed out from the slice's end, as the gather form does)
        for (int s = threadIdx.x; s < BLOB_S; s += PASS_T) {
            const int c = cntL[s];
            if (c) {
                const int j = uq[s];
                baseL[s] = sc.start[j] + atomicSub(&sc.cnt[j], c) - c;
                cntL[s] = 0;
            }
        }
        const double* rq = reinterpret_cast<const double*>(&recb[i]);
        const Q4 r0 = gload4(rq), rv = gload4(rq + 4);
        __syncthreads();
        double ox = 0.0, oy = 0.0, oz = 0.0;
        if (live) {
            // which of the lane's list positions name a dust neighbour: one in ten.  Walking all of them, nearly every step
            // finds SOME lane of the wave at a dust neighbour and the other sixty waiting for its gathers; walking only the
            // marked ones, the wave is through after as many steps as its busiest lane has dust neighbours (3-4, not 10)
            unsigned dm = 0;
            for (int m0 = 0; m0 < nm; ++m0) {
                const int kk = LPP * m0 + part;
                const unsigned sl = tile[kk * BLOB_P + t];
                bool d = false;
                if (sl < SLOT_OVER) d = dust[sl] != 0;
                else if (sl == SLOT_OVER) d = ptype[nbr[(size_t)kk * npad + p]] == 2.0;
                dm |= (d ? 1u : 0u) << m0;
            }
            while (dm) {                                   // (ascending list position: the order of the partial sum)
                const int m0 = __builtin_ctz(dm);
                dm &= dm - 1;
                const int kk = LPP * m0 + part;
                const unsigned sl = tile[kk * BLOB_P + t];
                int j;
                Q4 q0, qv;
                if (sl < SLOT_OVER) {
                    j = uq[sl];
                    q0 = lload4(img, (int)sl, 0); qv = lload4(img, (int)sl, 1);
                } else {
                    j = nbr[(size_t)kk * npad + p];
                    const double* qb = reinterpret_cast<const double*>(&recb[j]);
                    q0 = gload4(qb); qv = gload4(qb + 4);
                }
                const double dx = q0.a - r0.a, dy = q0.b - r0.b, dz = q0.c - r0.c;
                const double ds2 = q0.d, ds = sqrt(ds2);
                const double q = ds2 - (dx * dx + dy * dy + dz * dz);
                const double ds4 = ds2 * ds2;
                const double wf = m[j] * 315.0 * (q * q * q) / (201.06192982974676 * (ds4 * ds4 * ds));   // nsc:678-681
                double fx = 0.0, fy = 0.0, fz = 0.0;
                if (wf > 0.0) {
                    const double dvx = qv.a - rv.a, dvy = qv.b - rv.b, dvz = qv.c - rv.c;
                    const double coef = wf / mgm[j] * mcs[j] * sqrt(dvx * dvx + dvy * dvy + dvz * dvz);
                    fx = coef * dvx; fy = coef * dvy; fz = coef * dvz;
                    ox += fx; oy += fy; oz += fz;
                }
                if (j != i) {                                              // nsc:741
                    const int slot = (sl < SLOT_OVER) ? baseL[sl] + atomicAdd(&cntL[sl], 1)
                                                      : sc.start[j] + atomicSub(&sc.cnt[j], 1) - 1;
                    sphx_drag_put(sc, slot, ((u64)(unsigned)id[i] << 8) | (u64)kk, -fx, -fy, -fz);
                }
            }
        }
        const double tx = group_total(ox), ty = group_total(oy), tz = group_total(oz);
        if (live && part == 0) { onto[3 * (size_t)i] = tx; onto[3 * (size_t)i + 1] = ty; onto[3 * (size_t)i + 2] = tz; }
        __syncthreads();
    }
}

// counting pass of the scatter plan (true) / the pass itself (false -> fill) on the blob lists
int sphx_blob_drag(sphx_ctx* ctx, int64_t n, int k, bool count_only, const double* m, const double* ptype, const double* mgm,
                   const double* mcs, const int* id, double* onto, const DragScatter& sc) {
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
    if (!ctx->drag_attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_drag_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(IMG_BYTES(64, SPHX_MAX_K) + DRAG_XLDS)));
        ctx->drag_attr_set = true;
    }
    if (count_only)
        hipLaunchKernelGGL(blob_drag_kernel<false>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T),
                           (size_t)KPAD(k) * BLOB_P * sizeof(u16) + DRAG_XLDS, ctx->stream, (int)n, (int)npad, k, nblk,
                           ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(), ctx->qorder, nullptr, m, ptype, mgm, mcs,
                           id, onto, sc);
    else
        hipLaunchKernelGGL(blob_drag_kernel<true>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(64, k) + DRAG_XLDS,
                           ctx->stream, (int)n, (int)npad, k, nblk, ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(),
                           ctx->qorder, ctx->recv.as<RecB>(), m, ptype, mgm, mcs, id, onto, sc);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

int sphx_blob_join(sphx_ctx* ctx) {
    if (ctx->dedup_pending) {
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        ctx->dedup_pending = false;
    }
    return SPHX_OK;
}

BlobSel sphx_blob_sel(sphx_ctx* ctx, int part) {
    if (!ctx->blob_split_valid) return BlobSel{nullptr, nullptr, 0};
    const int* list = ctx->blob_split.as<int>();
    return BlobSel{list, list + (size_t)ctx->blob_split_nblk, part == 0 ? 3 : part};
}

int sphx_blob_grid(sphx_ctx* ctx, int nblk) {
    if (ctx->blob_grid <= 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        ctx->blob_grid = ((cus * 2 + 7) / 8) * 8;
    }
    return nblk < ctx->blob_grid ? nblk : ctx->blob_grid;
}

static int blob_attr_once(sphx_ctx* ctx) {
    if (ctx->blob_attr_set) return SPHX_OK;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_density_kernel<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(64, SPHX_MAX_K)));
#ifdef SPHX_EXPERIMENTS
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_density_kernel<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_density_kernel<2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_density_kernel<3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_pi_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(72, SPHX_MAX_K)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(blob_visc_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)IMG_BYTES(72, SPHX_MAX_K)));
    ctx->blob_attr_set = true;
    return SPHX_OK;
}

int sphx_blob_density(sphx_ctx* ctx, int64_t n, int k) {
    SPHX_TRY(blob_attr_once(ctx));
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
#ifdef SPHX_EXPERIMENTS
    if (ctx->exp_blob) {        // timing experiments (SPHX_BLOB_EXP), outputs discarded
        SPHX_TRY(sphx_ensure(ctx, ctx->in_j, (size_t)n * 12 * sizeof(double)));
        double* d = ctx->in_j.as<double>();
        size_t lds = IMG_BYTES(64, k);
        if (ctx->exp_blob_lds) lds = ctx->exp_blob_lds;      // e.g. 100000: one workgroup per CU
        for (int mode = 0; mode < 4; ++mode) {
            if (!(ctx->exp_blob & (1 << mode))) continue;
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            HIPCHK(hipEventRecord(e0, ctx->stream));
#define BLOB_EXP_LAUNCH(M)                                                                                           \
            hipLaunchKernelGGL(blob_density_kernel<M>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), lds, ctx->stream, (int)n, (int)npad, \
                               k, nblk, ctx->clip_grad, ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(), ctx->qorder, \
                               nullptr, (int)n, ctx->rec1.as<RecA>(), d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 8 * n, \
                               BlobSel{nullptr, nullptr, 0})
            if (mode == 0) BLOB_EXP_LAUNCH(0);
            else if (mode == 1) BLOB_EXP_LAUNCH(1);
            else if (mode == 2) BLOB_EXP_LAUNCH(2);
            else BLOB_EXP_LAUNCH(3);
            HIPCHK(hipEventRecord(e1, ctx->stream));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            fprintf(stderr, "[sphx] blob_density experiment %d (lds %zu): %.4f ms\n", mode, lds, ms);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
#endif
    hipLaunchKernelGGL(blob_density_kernel<0>, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(64, k), ctx->stream, (int)n, (int)npad, k, nblk, ctx->clip_grad,
                       ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(),
                       ctx->qorder, ctx->map_perm, ctx->map_perm ? ctx->map_nactive : (int)n, ctx->rec1.as<RecA>(),
                       ctx->rho_s.as<double>(), ctx->rho.as<double>(),
                       ctx->rhod.as<double>(), ctx->nden.as<double>(), ctx->lean_outputs ? nullptr : ctx->G.as<double>(), ctx->ha.as<double>(),
                       sphx_blob_sel(ctx, ctx->pass_part));
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

int sphx_blob_pi(sphx_ctx* ctx, int64_t n, int k, u64* ct_bits) {
    SPHX_TRY(blob_attr_once(ctx));
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
    hipLaunchKernelGGL(blob_pi_kernel, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(72, k), ctx->stream, (int)n, (int)npad, k, nblk,
                       ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(),
                       ctx->qorder, ctx->map_perm, ctx->map_perm ? ctx->map_nactive : (int)n, ctx->recv.as<RecB>(),
                       ctx->rho_s.as<double>(), ctx->self_s.as<RecSelf>(), ctx->bc_s.as<RecBC>(), ctx->Pi.as<double>(),
                       ctx->map_perm ? ctx->Bw.as<double>() : nullptr, ct_bits, sphx_blob_sel(ctx, ctx->pass_part));
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

int sphx_blob_visc(sphx_ctx* ctx, int64_t n, int k, const double* m) {
    SPHX_TRY(blob_attr_once(ctx));
    const int64_t npad = sphx_pad64(n);
    const int nblk = (int)((npad + BLOB_P - 1) / BLOB_P);
    hipLaunchKernelGGL(blob_visc_kernel, dim3(sphx_blob_grid(ctx, nblk)), dim3(PASS_T), IMG_BYTES(72, k), ctx->stream, (int)n, (int)npad, k, nblk, ctx->clip_grad,
                       ctx->nbr.as<int>(), ctx->slot16.as<u16>(), ctx->uniq.as<int>(),
                       ctx->qorder, ctx->map_perm, ctx->map_perm ? ctx->map_nactive : (int)n, ctx->recv.as<RecB>(),
                       ctx->bc_s.as<RecBC>(), m, ctx->va.as<double>(),
                       ctx->vh.as<double>(), sphx_blob_sel(ctx, ctx->pass_part));
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
