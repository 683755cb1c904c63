// sphx_integrate.hip - state permutation, time-step control and the leapfrog update
// (restated from the driver text: drv:222-238 dt + clamps, drv:460-491 integrator).
#include "sphx_internal.h"
// NumPy never fuses a multiply into an add: keep every operation separately rounded so that
// cancellations such as h_j^2 - r^2 at the kernel edge reproduce the reference bit for bit.
#pragma clang fp contract(off)
#include <float.h>

#include "sphx_leapfrog.h"
#define nan_to_num sphx_nan_to_num

// ---- drv:233-238: clamp |x| <= 1e11 AU, nan_to_num(x), nan_to_num(v) -----------------------
__global__ __launch_bounds__(256) void clamp_kernel(int n, double lim, double* x, double* y, double* z,
                                                    double* vx, double* vy, double* vz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double* p[3] = {x, y, z};
    double* v[3] = {vx, vy, vz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        p[c][i] = sphx_clamp_pos(p[c][i], lim);
        v[c][i] = nan_to_num(v[c][i]);
    }
}

int sphx_clamp(sphx_ctx* ctx, int64_t n, StateArrays& s) {
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                       ctx->cst.pos_clamp, s.x.as<double>(), s.y.as<double>(), s.z.as<double>(),
                       s.vx.as<double>(), s.vy.as<double>(), s.vz.as<double>());
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- gather up to 16 f64 arrays (+ id, + optional (n,s) composition) by perm ----------------
struct GatherArgs {
    int n, narr, s;
    const int* perm;
    const double* src[18];
    double* dst[18];
    const int* id_src; int* id_dst; int* inv;
    const double* fun_src; double* fun_dst;
    // the blob order's scatter (sphx_grid.hip blob_scatter) carried along: porder[...] = t  (bs_porder == nullptr: not)
    GridParams bs_g; BlobBits bs_b;
    const int *bs_cell_of, *bs_cell_start, *bs_mstart;
    int* bs_porder;
    int* bs_mcount;                    // the curve's per-cell counts: put back to zero here (the scan has read them)
};
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const int p = a.perm[t];
    if (a.bs_porder) {
        const int c = a.bs_cell_of[p];
        const int cx = c % a.bs_g.nx, cy = (c / a.bs_g.nx) % a.bs_g.ny, cz = c / (a.bs_g.nx * a.bs_g.ny);
        const unsigned rk = blob_rank(cx, cy, cz, a.bs_b);
        a.bs_porder[a.bs_mstart[rk] + (t - a.bs_cell_start[c])] = t;
        if (t == a.bs_cell_start[c]) a.bs_mcount[rk] = 0;          // (one member per cell cleans up)
    }
    for (int q = 0; q < a.narr; ++q) a.dst[q][t] = a.src[q][p];
    if (a.id_src) {
        const int id = a.id_src[p];
        a.id_dst[t] = id;
        if (a.inv) a.inv[id] = t;
    }
    if (a.fun_src) {                       // rows of a.s doubles, a multiple of 16: aligned 16-B copies
        const double2* src = reinterpret_cast<const double2*>(a.fun_src + (size_t)p * a.s);
        double2* dst = reinterpret_cast<double2*>(a.fun_dst + (size_t)t * a.s);
        for (int q = 0; q < a.s / 2; ++q) dst[q] = src[q];
    }
}

// split = true (the fused loop, a side stream at hand): what the search reads - positions, previous radii, ids - is
// permuted on the main stream, everything else on the side stream beside the search (which is bound by instruction
// issue and leaves the memory system idle); the caller makes the main stream wait for ctx->ev_perm before it reads those.
// after_first: an event the caller wants recorded behind the first launch (its search-timing start): doubles as the fork.
int sphx_permute_state(sphx_ctx* ctx, int64_t n, bool split, hipEvent_t after_first) {
    StateArrays& a = ctx->st;
    StateArrays& b = ctx->alt;
    // (the search's arrays first)
    DevBuf* src[] = {&a.x, &a.y, &a.z, &a.hprev, &a.vx, &a.vy, &a.vz, &a.ax, &a.ay, &a.az,
                     &a.m, &a.T, &a.mu, &a.gam, &a.E, &a.ptype};
    DevBuf* dst[] = {&b.x, &b.y, &b.z, &b.hprev, &b.vx, &b.vy, &b.vz, &b.ax, &b.ay, &b.az,
                     &b.m, &b.T, &b.mu, &b.gam, &b.E, &b.ptype};
    GatherArgs g;
    g.bs_porder = nullptr;
    if (ctx->blob_scatter_pending) {
        g.bs_g = ctx->grid; g.bs_b = ctx->blob_scatter_bits;
        g.bs_cell_of = ctx->cell_of.as<int>(); g.bs_cell_start = ctx->cell_start.as<int>();
        g.bs_mstart = ctx->blob_scatter_mstart; g.bs_porder = ctx->porder.as<int>();
        g.bs_mcount = ctx->mcount.as<int>();
        ctx->blob_scatter_pending = false;
    }
    g.n = (int)n; g.narr = 16; g.s = ctx->sp;
    g.perm = ctx->perm.as<int>();
    for (int q = 0; q < 16; ++q) {
        SPHX_TRY(sphx_ensure(ctx, *dst[q], (size_t)n * sizeof(double)));
        g.src[q] = src[q]->as<double>();
        g.dst[q] = dst[q]->as<double>();
    }
    if (ctx->drag) {                          // the drag coefficients travel with the particles
        SPHX_TRY(sphx_ensure(ctx, b.mgm, (size_t)n * sizeof(double)));
        SPHX_TRY(sphx_ensure(ctx, b.mcs, (size_t)n * sizeof(double)));
        g.src[16] = a.mgm.as<double>(); g.dst[16] = b.mgm.as<double>();
        g.src[17] = a.mcs.as<double>(); g.dst[17] = b.mcs.as<double>();
        g.narr = 18;
    }
    SPHX_TRY(sphx_ensure(ctx, b.id, (size_t)n * sizeof(int)));
    // (no inverse permutation: nothing in the step reads one - a scattered 4-byte store per particle saved)
    g.id_src = a.id.as<int>(); g.id_dst = b.id.as<int>(); g.inv = nullptr;
    g.fun_src = nullptr; g.fun_dst = nullptr;      // (the composition rows are not moved: ctx->fun_id, reached through the id)
    split = split && ctx->side_stream && ctx->ev_perm_fork && ctx->ev_perm;
    GatherArgs rest = g;                       // velocities ... ptype (+ drag coefficients, + composition): side stream
    if (split) {
        rest.narr = g.narr - 4;
        for (int q = 0; q < rest.narr; ++q) { rest.src[q] = g.src[q + 4]; rest.dst[q] = g.dst[q + 4]; }
        rest.id_src = nullptr;
        rest.bs_porder = nullptr;
        g.narr = 4;
        g.fun_src = nullptr; g.fun_dst = nullptr;
    }
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g);
    if (after_first) HIPCHK(hipEventRecord(after_first, ctx->stream));
    if (split) {
        // (behind the search's part, not beside it: both are bound by the same memory system)
        hipEvent_t fork_ev = after_first ? after_first : ctx->ev_perm_fork;
        if (!after_first) HIPCHK(hipEventRecord(fork_ev, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->side_stream, fork_ev, 0));
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->side_stream, rest);
        HIPCHK(hipEventRecord(ctx->ev_perm, ctx->side_stream));
    }
    HIPCHK(hipGetLastError());
    StateArrays tmp = ctx->st; ctx->st = ctx->alt; ctx->alt = tmp;
    return SPHX_OK;
}

// ---- clipped sum of h (next grid's cell size) -------------------------------------------------
// Escaped particles have kNN radii many orders of magnitude above the cloud's; values above
// `clip` (8 x the previous mean) are left out.  out[0] += sum, out[3] += count (SC_HSUM, SC_HCNT).
// No pre-zeroed accumulators: every block leaves its partial sums, the last one to finish (a ticket
// that wraps back to 0 by itself) adds them up in block order - one launch, and the same bits every run.
__global__ __launch_bounds__(256) void hsum_kernel(int n, const double* h, double clip, double* partial,
                                                   unsigned* ticket, double* out_sum, double* out_cnt, u64* counters) {
    __shared__ double sm[4], sc[4];
    __shared__ bool last;
    double s = 0.0, c = 0.0;
    unsigned nbad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = h[i];
        if (v > 0.0 && (clip <= 0.0 || v <= clip)) { s += v; c += 1.0; }
        nbad += (v > 0.0 && v <= DBL_MAX) ? 0u : 1u;           // 0 (coincident points), NaN, inf: BAD_H
    }
    if (__ballot(nbad != 0u)) {                               // (never in a sane run)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nbad += __shfl_xor(nbad, o, 64);
        if ((threadIdx.x & 63) == 0)
            atomicAdd(&counters[(((blockIdx.x << 2) | (threadIdx.x >> 6)) & (BADC_BUCKETS - 1)) * BADC_STRIDE + BAD_H], (u64)nbad);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
        partial[2 * blockIdx.x + 1] = sc[0] + sc[1] + sc[2] + sc[3];
        __threadfence();
        last = atomicInc(ticket, gridDim.x - 1) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {                                    // (wave-uniform: the whole block or none of it)
        __threadfence();
        double ts = 0.0, tc = 0.0;
        for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) {      // fixed assignment and tree
            ts += __builtin_nontemporal_load(&partial[2 * b]);
            tc += __builtin_nontemporal_load(&partial[2 * b + 1]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ts += __shfl_xor(ts, o, 64); tc += __shfl_xor(tc, o, 64); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = ts; sc[threadIdx.x >> 6] = tc; }
        __syncthreads();
        if (threadIdx.x == 0) {
            *out_sum = (sm[0] + sm[1]) + (sm[2] + sm[3]);
            *out_cnt = (sc[0] + sc[1]) + (sc[2] + sc[3]);
        }
    }
}
#define HSUM_BLOCKS 512
int sphx_hsum(sphx_ctx* ctx, int64_t n, const double* h) {
    double* out = ctx->scal.as<double>() + SC_HSUM;
    double* cnt = ctx->scal.as<double>() + SC_HCNT;
    const bool fresh = ctx->hsum_tmp.p == nullptr;
    SPHX_TRY(sphx_ensure(ctx, ctx->hsum_tmp, (size_t)(2 * HSUM_BLOCKS + 2) * sizeof(double)));
    double* partial = ctx->hsum_tmp.as<double>();
    unsigned* ticket = reinterpret_cast<unsigned*>(partial + 2 * HSUM_BLOCKS);
    if (fresh) HIPCHK(hipMemsetAsync(ticket, 0, sizeof(double), ctx->stream));
    int blocks = (int)((n + 255) / 256);
    if (blocks > HSUM_BLOCKS) blocks = HSUM_BLOCKS;
    hipLaunchKernelGGL(hsum_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (int)n, h, ctx->h_clip, partial, ticket,
                       out, cnt, ctx->badc.as<u64>());
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

__global__ void prime_ct_kernel(u64* ct_bits) { *ct_bits = SPHX_CT_NONE; }
int sphx_prime_ct(sphx_ctx* ctx, u64* ct_bits) {
    hipLaunchKernelGGL(prime_ct_kernel, dim3(1), dim3(1), 0, ctx->stream, ct_bits);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- drv:222-229: dt from the crossing time ------------------------------------------------
__global__ void dt_kernel(u64* ct_bits, double* dt_out, int first, double fixed_dt, double dt_0,
                          double max_age) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double dt;
    const u64 b = *ct_bits;
    *ct_bits = SPHX_CT_NONE;                   // "none yet" again: the next step's pass 2 needs no priming
    if (fixed_dt > 0.0) {
        dt = fixed_dt;
    } else {
        const double ct = sphx_ct_value(b == SPHX_CT_NONE, __longlong_as_double((long long)b), dt_0);   // nsc:783-786
        dt = sphx_dt_rule(ct, first, dt_0, max_age);                                 // drv:223-229
    }
    *dt_out = dt;
}

int sphx_compute_dt(sphx_ctx* ctx, int first, double fixed_dt) {
    hipLaunchKernelGGL(dt_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->scal.as<u64>() + SC_CT_BITS,
                       ctx->scal.as<double>() + SC_DT, first, fixed_dt, ctx->cst.dt_0, ctx->cst.max_age);
    HIPCHK(hipGetLastError());
    ctx->ct_primed = true;
    return SPHX_OK;
}

// ---- drv:460-491: acceleration assembly, leapfrog, energy ------------------------------------
struct IntegArgs {
    int n;
    double *x, *y, *z, *vx, *vy, *vz, *ax, *ay, *az;   // ax.. = previous total accel (in/out)
    double *E, *T;
    const double *m, *mu, *gam, *ptype;
    const double *ha, *va, *vh;                        // hydro_update outputs (reference sign)
    const double *rho, *rhod, *drag_on, *drag_re;      // drag terms (drag_on == nullptr: none)
    const double* grav;                                // self-gravity (nullptr: none)
    const double* G;                                   // loop-form mode: del_pressure (physical sign); nullptr otherwise
    const double* dt;
    double m_h, kB;
    int no_old;                                        // drv:484-485: no previous acceleration of this shape
    // the fused loop: dt worked out here from the step's crossing-time vote (drv:222-229; was a launch of its own) and
    // left in *dt_out by the first thread; the vote is reset by the next step's first kernel (grid_count_fused)
    const u64* ct_bits;                                // nullptr: dt is read from *dt
    double* dt_out;
    int first;
    double fixed_dt, dt_0, max_age;
    u64* counters;                                     // failure counters (BAD_*); nullptr: not counted
};
__global__ __launch_bounds__(256) void integrate_kernel(IntegArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    double dt;
    if (a.ct_bits) {
        if (a.fixed_dt > 0.0) {
            dt = a.fixed_dt;
        } else {
            const u64 b = *a.ct_bits;
            const double ct = sphx_ct_value(b == SPHX_CT_NONE, __longlong_as_double((long long)b), a.dt_0);   // nsc:783-786
            dt = sphx_dt_rule(ct, a.first, a.dt_0, a.max_age);                                                  // drv:223-229
        }
        if (i == 0) *a.dt_out = dt;
    } else {
        dt = *a.dt;
    }
    const double g = (a.ptype[i] == 0.0) ? 1.0 : 0.0;
    const double v[3] = {a.vx[i], a.vy[i], a.vz[i]};
    double pa[3], vis[3];
    bool bad_acc = false;                                     // (what nan_to_num is about to hide: counted, BAD_ACCEL)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (a.G) {                                            // loop forms carry the physical sign themselves
            const double praw = a.G[3 * (size_t)i + c] / a.rho[i] * g;
            pa[c] = nan_to_num(praw);                                       // drv:460
            vis[c] = a.va[3 * (size_t)i + c];                               // av[0], drv:473
            bad_acc = bad_acc || !sphx_finite(praw) || !sphx_finite(vis[c]);
        } else {
            const double praw = -a.ha[3 * (size_t)i + c] * g, vraw = -a.va[3 * (size_t)i + c] * g;
            pa[c] = nan_to_num(praw);                             // physical sign, drv:460
            vis[c] = nan_to_num(vraw);
            bad_acc = bad_acc || !sphx_finite(praw) || !sphx_finite(vraw);
        }
        if (a.drag_on) {                                      // drv:462-463,473
            const double draw = a.drag_on[3 * (size_t)i + c] * a.rhod[i] / a.rho[i] * g, rraw = a.drag_re[3 * (size_t)i + c];
            const double dg = nan_to_num(draw);
            vis[c] = dg + nan_to_num(rraw) + vis[c];
            bad_acc = bad_acc || !sphx_finite(draw) || !sphx_finite(rraw);
        }
    }
    double x[3] = {a.x[i], a.y[i], a.z[i]};
    double vv[3] = {v[0], v[1], v[2]};
    double gr[3], old[3], tot[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gr[c] = a.grav ? a.grav[3 * (size_t)i + c] : 0.0;
    old[0] = a.ax[i]; old[1] = a.ay[i]; old[2] = a.az[i];
    sphx_leapfrog_update(dt, x, vv, vis, pa, a.grav ? gr : nullptr, a.no_old ? nullptr : old, tot);   // drv:475-486
    a.x[i] = x[0]; a.y[i] = x[1]; a.z[i] = x[2];
    a.vx[i] = vv[0]; a.vy[i] = vv[1]; a.vz[i] = vv[2];
    a.ax[i] = tot[0]; a.ay[i] = tot[1]; a.az[i] = tot[2];
    double E = a.E[i], T;
    const double heat = a.vh[i];
    const bool bad_en = !sphx_finite(E) || !sphx_finite(heat * dt);
    sphx_energy_update(dt, heat, a.mu[i], a.gam[i], a.m[i], a.m_h, a.kB, E, T);   // drv:490-491
    a.E[i] = E;
    a.T[i] = T;
    if (a.counters) {
        bool bad_st = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) bad_st = bad_st || !sphx_finite(x[c]) || !sphx_finite(vv[c]);
        sphx_count_bad(a.counters, BAD_ACCEL, bad_acc);
        sphx_count_bad(a.counters, BAD_ENERGY, bad_en);
        sphx_count_bad(a.counters, BAD_STATE, bad_st);
    }
}

int sphx_integrate(sphx_ctx* ctx, int64_t n, int fold_dt, int first, double fixed_dt) {
    StateArrays& s = ctx->st;
    IntegArgs a;
    a.ct_bits = fold_dt ? ctx->scal.as<u64>() + SC_CT_BITS : nullptr;
    a.dt_out = ctx->scal.as<double>() + SC_DT;
    a.first = first; a.fixed_dt = fixed_dt; a.dt_0 = ctx->cst.dt_0; a.max_age = ctx->cst.max_age;
    a.n = (int)n;
    a.x = s.x.as<double>(); a.y = s.y.as<double>(); a.z = s.z.as<double>();
    a.vx = s.vx.as<double>(); a.vy = s.vy.as<double>(); a.vz = s.vz.as<double>();
    a.ax = s.ax.as<double>(); a.ay = s.ay.as<double>(); a.az = s.az.as<double>();
    SPHX_TRY(sphx_ensure(ctx, ctx->alt.T, (size_t)n * sizeof(double)));
    a.E = s.E.as<double>(); a.T = ctx->alt.T.as<double>();      // (the kernel never reads T: written elsewhere, swapped below)
    a.m = s.m.as<double>(); a.mu = s.mu.as<double>(); a.gam = s.gam.as<double>();
    a.ptype = s.ptype.as<double>();
    a.ha = ctx->ha.as<double>(); a.va = ctx->va.as<double>(); a.vh = ctx->vh.as<double>();
    a.rho = ctx->rho.as<double>(); a.rhod = ctx->rhod.as<double>();
    a.drag_on = ctx->drag ? ctx->drag_on.as<double>() : nullptr;
    a.drag_re = ctx->drag ? ctx->drag_re.as<double>() : nullptr;
    a.grav = ctx->gravity ? ctx->grav.as<double>() : nullptr;
    a.G = ctx->loop_forms ? ctx->G.as<double>() : nullptr;
    a.dt = ctx->scal.as<double>() + SC_DT;
    a.m_h = ctx->cst.m_h; a.kB = ctx->cst.k_B;
    a.no_old = 0;
    a.counters = ctx->badc.as<u64>();
    hipLaunchKernelGGL(integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    { DevBuf t = s.T; s.T = ctx->alt.T; ctx->alt.T = t; }       // st.T: the new temperatures; alt.T: those the sums read
    ctx->tprev_valid = true;
    return SPHX_OK;
}

// P_i = n_i k_B T_i with n and T of the same instant - the one the step's sums were formed at (nsc:607; the reference's
// own `pressure` line is commented out at nsc:608) - scattered into the caller's order
__global__ __launch_bounds__(256) void pressure_by_id(int n, const int* id, const double* nden, const double* T, double kB,
                                                      double* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[id[t]] = nden[t] * kB * T[t];
}
extern "C" int sphx_state_download_pressure(sphx_ctx* ctx, double* pressure) {
    if (!ctx || !pressure) return SPHX_E_ARG;
    if (!ctx->has_state) return sphx_set_err(ctx, SPHX_E_STATE, "no state uploaded");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = ctx->n;
    if (ctx->step_count < 1 || !ctx->tprev_valid || !ctx->alt.T.p) { memset(pressure, 0, (size_t)n * sizeof(double)); return SPHX_OK; }
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * sizeof(double)));
    hipLaunchKernelGGL(pressure_by_id, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                       ctx->st.id.as<int>(), ctx->nden.as<double>(), ctx->alt.T.as<double>(), ctx->cst.k_B, ctx->out_a.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(pressure, ctx->out_a.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

// ---- the driver's inline statements as array functions (host pointers) ------------------------
__global__ __launch_bounds__(256) void dt_rule_kernel(int n, const double* ct, const int* first, double dt_0,
                                                      double max_age, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sphx_dt_rule(ct[i], first[i], dt_0, max_age);
}
static int up_(sphx_ctx* ctx, DevBuf& b, const void* host, size_t bytes) {
    SPHX_TRY(sphx_ensure(ctx, b, bytes));
    HIPCHK(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SPHX_OK;
}
#define NEEDP(p) do { if (!(p)) return sphx_set_err(ctx, SPHX_E_ARG, "%s: argument %s is NULL", __func__, #p); } while (0)

extern "C" int sphx_dt_rule(sphx_ctx* ctx, int64_t n, const double* ct, const int32_t* first, double* dt) {
    if (!ctx) return SPHX_E_ARG;
    NEEDP(ct); NEEDP(first); NEEDP(dt);
    if (n < 1) return SPHX_OK;
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(up_(ctx, ctx->in_a, ct, (size_t)n * 8));
    SPHX_TRY(up_(ctx, ctx->in_b, first, (size_t)n * 4));
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * 8));
    hipLaunchKernelGGL(dt_rule_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                       ctx->in_a.as<double>(), ctx->in_b.as<int>(), ctx->cst.dt_0, ctx->cst.max_age, ctx->out_a.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dt, ctx->out_a.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

__global__ __launch_bounds__(256) void clamp_aos_kernel(int n3, double lim, double* pos, double* vel) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n3) return;
    pos[e] = sphx_clamp_pos(pos[e], lim);
    vel[e] = nan_to_num(vel[e]);
}
extern "C" int sphx_clamp_arrays(sphx_ctx* ctx, int64_t n, double* points, double* velocities) {
    if (!ctx) return SPHX_E_ARG;
    NEEDP(points); NEEDP(velocities);
    if (n < 1) return SPHX_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb3 = (size_t)n * 24;
    SPHX_TRY(up_(ctx, ctx->in_a, points, nb3));
    SPHX_TRY(up_(ctx, ctx->in_b, velocities, nb3));
    hipLaunchKernelGGL(clamp_aos_kernel, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, ctx->stream, (int)(3 * n),
                       ctx->cst.pos_clamp, ctx->in_a.as<double>(), ctx->in_b.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(points, ctx->in_a.p, nb3, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(velocities, ctx->in_b.p, nb3, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

// drv:460-491 on host arrays: the SAME kernel the fused loop runs (integrate_kernel in loop-form
// mode), fed from scratch buffers instead of the resident state.
extern "C" int sphx_leapfrog(sphx_ctx* ctx, int64_t n, double* points, double* velocities, double* total_accel,
                             const double* old_accel, double* E_internal, double* T, const double* mass,
                             const double* mu, const double* gamma, const double* ptype, const double* grav_accel,
                             const double* delp, const double* densities, const double* dust_densities,
                             const double* drag_on_gas, const double* drag_reaction, const double* av_accel,
                             const double* av_heat, double dt) {
    if (!ctx) return SPHX_E_ARG;
    NEEDP(points); NEEDP(velocities); NEEDP(total_accel); NEEDP(E_internal); NEEDP(T); NEEDP(mass); NEEDP(mu);
    NEEDP(gamma); NEEDP(ptype); NEEDP(delp); NEEDP(densities); NEEDP(av_accel); NEEDP(av_heat);
    if ((drag_on_gas != nullptr) != (drag_reaction != nullptr) || (drag_on_gas && !dust_densities))
        return sphx_set_err(ctx, SPHX_E_ARG, "sphx_leapfrog: drag needs drag_on_gas, drag_reaction and dust_densities");
    if (n < 1) return SPHX_OK;
    if (n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb = (size_t)n * 8;
    // one scratch allocation: 9 SoA columns + E,T + 5 scalars + up to 7 (n,3)/(n,) inputs
    DevBuf& w = ctx->out_b;
    SPHX_TRY(sphx_ensure(ctx, w, 40 * nb));
    double* base = w.as<double>();
    size_t off = 0;
    auto take = [&](size_t cols) { double* q = base + off; off += cols * (size_t)n; return q; };
    double *x = take(1), *y = take(1), *z = take(1), *vx = take(1), *vy = take(1), *vz = take(1);
    double *ax = take(1), *ay = take(1), *az = take(1), *E = take(1), *Tt = take(1);
    double *m = take(1), *muu = take(1), *gm = take(1), *pt = take(1), *rho = take(1), *rhod = take(1), *heat = take(1);
    double *aos = take(3), *G = take(3), *av = take(3), *gr = take(3), *don = take(3), *dre = take(3);
    auto h2d = [&](double* d, const double* h, size_t cols) {
        return hipMemcpyAsync(d, h, cols * nb, hipMemcpyHostToDevice, ctx->stream);
    };
    HIPCHK(h2d(aos, points, 3));
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, aos, x, y, z));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(h2d(aos, velocities, 3));
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, aos, vx, vy, vz));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (old_accel) {
        HIPCHK(h2d(aos, old_accel, 3));
        SPHX_TRY(sphx_aos_to_soa3(ctx, n, aos, ax, ay, az));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    HIPCHK(h2d(E, E_internal, 1)); HIPCHK(h2d(m, mass, 1)); HIPCHK(h2d(muu, mu, 1)); HIPCHK(h2d(gm, gamma, 1));
    HIPCHK(h2d(pt, ptype, 1)); HIPCHK(h2d(rho, densities, 1)); HIPCHK(h2d(heat, av_heat, 1));
    HIPCHK(h2d(G, delp, 3)); HIPCHK(h2d(av, av_accel, 3));
    if (dust_densities) HIPCHK(h2d(rhod, dust_densities, 1));
    if (grav_accel) HIPCHK(h2d(gr, grav_accel, 3));
    if (drag_on_gas) { HIPCHK(h2d(don, drag_on_gas, 3)); HIPCHK(h2d(dre, drag_reaction, 3)); }
    double* dtd = ctx->scal.as<double>() + 15;          // a slot nothing else uses
    HIPCHK(hipMemcpyAsync(dtd, &dt, 8, hipMemcpyHostToDevice, ctx->stream));
    IntegArgs a;
    a.ct_bits = nullptr; a.dt_out = nullptr; a.first = 0; a.fixed_dt = 0.0; a.dt_0 = 0.0; a.max_age = 0.0;
    a.n = (int)n;
    a.x = x; a.y = y; a.z = z; a.vx = vx; a.vy = vy; a.vz = vz; a.ax = ax; a.ay = ay; a.az = az;
    a.E = E; a.T = Tt; a.m = m; a.mu = muu; a.gam = gm; a.ptype = pt;
    a.ha = nullptr; a.va = av; a.vh = heat; a.rho = rho; a.rhod = rhod;
    a.drag_on = drag_on_gas ? don : nullptr; a.drag_re = drag_on_gas ? dre : nullptr;
    a.grav = grav_accel ? gr : nullptr;
    a.G = G;
    a.dt = dtd;
    a.m_h = ctx->cst.m_h; a.kB = ctx->cst.k_B;
    a.no_old = old_accel ? 0 : 1;
    a.counters = nullptr;              // (an array call on the caller's arrays: its NaNs are the caller's to see)
    hipLaunchKernelGGL(integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    struct Out { double* host; double *a, *b, *c; } outs[] = {{points, x, y, z}, {velocities, vx, vy, vz}, {total_accel, ax, ay, az}};
    for (Out& o : outs) {
        SPHX_TRY(sphx_soa3_to_aos_by_id(ctx, n, nullptr, o.a, o.b, o.c, aos));
        HIPCHK(hipMemcpyAsync(o.host, aos, 3 * nb, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(E_internal, E, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(T, Tt, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

// ---- small layout helpers -------------------------------------------------------------------
__global__ __launch_bounds__(256) void aos_to_soa3(int n, const double* aos, double* x, double* y, double* z) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[i] = aos[3 * (size_t)i]; y[i] = aos[3 * (size_t)i + 1]; z[i] = aos[3 * (size_t)i + 2];
}
int sphx_aos_to_soa3(sphx_ctx* ctx, int64_t n, const double* aos, double* x, double* y, double* z) {
    hipLaunchKernelGGL(aos_to_soa3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, aos, x, y, z);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
// out_aos[id[t]] = (x[t], y[t], z[t])   (id == nullptr: identity)
__global__ __launch_bounds__(256) void soa3_to_aos_by_id(int n, const int* id, const double* x,
                                                         const double* y, const double* z, double* aos) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    size_t o = id ? (size_t)id[t] : (size_t)t;
    aos[3 * o] = x[t]; aos[3 * o + 1] = y[t]; aos[3 * o + 2] = z[t];
}
int sphx_soa3_to_aos_by_id(sphx_ctx* ctx, int64_t n, const int* id, const double* x, const double* y,
                           const double* z, double* aos) {
    hipLaunchKernelGGL(soa3_to_aos_by_id, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, id, x, y, z, aos);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
// out[id[t]*w + c] = in[t*w + c]
__global__ __launch_bounds__(256) void scatter_rows_by_id(int n, int w, const int* id, const double* in, double* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    size_t o = (size_t)id[t];
    for (int c = 0; c < w; ++c) out[o * w + c] = in[(size_t)t * w + c];
}
int sphx_scatter_rows_by_id(sphx_ctx* ctx, int64_t n, int w, const int* id, const double* in, double* out) {
    hipLaunchKernelGGL(scatter_rows_by_id, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, w, id, in, out);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
// gather three sorted SoA arrays: xs[t] = x[perm[t]] ...
__global__ __launch_bounds__(256) void gather3(int n, const int* perm, const double* x, const double* y,
                                               const double* z, double* xs, double* ys, double* zs) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int p = perm[t];
    xs[t] = x[p]; ys[t] = y[p]; zs[t] = z[p];
}
int sphx_gather3(sphx_ctx* ctx, int64_t n, const int* perm, const double* x, const double* y,
                 const double* z, double* xs, double* ys, double* zs) {
    hipLaunchKernelGGL(gather3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, perm, x, y, z, xs, ys, zs);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
__global__ __launch_bounds__(256) void iota_kernel(int n, int* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = t;
}
int sphx_iota(sphx_ctx* ctx, int64_t n, int* out) {
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, out);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
