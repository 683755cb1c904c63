// sphx_integrate.hip - state permutation, time-step control and the leapfrog update
// (restated from the driver text: drv:222-238 dt + clamps, drv:460-491 integrator).
#include "sphx_internal.h"
// NumPy never fuses a multiply into an add: keep every operation separately rounded so that
// cancellations such as h_j^2 - r^2 at the kernel edge reproduce the reference bit for bit.
#pragma clang fp contract(off)
#include <float.h>

__device__ __forceinline__ double nan_to_num(double v) {
    if (v != v) return 0.0;
    if (v > DBL_MAX) return DBL_MAX;
    if (v < -DBL_MAX) return -DBL_MAX;
    return v;
}

// ---- drv:233-238: clamp |x| <= 1e11 AU, nan_to_num(x), nan_to_num(v) -----------------------
__global__ __launch_bounds__(256) void clamp_kernel(int n, double lim, double* x, double* y, double* z,
                                                    double* vx, double* vy, double* vz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double* p[3] = {x, y, z};
    double* v[3] = {vx, vy, vz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double q = p[c][i];
        q = (q > lim) ? lim : q;
        q = (q < -lim) ? -lim : q;
        p[c][i] = nan_to_num(q);
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
};
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const int p = a.perm[t];
    for (int q = 0; q < a.narr; ++q) a.dst[q][t] = a.src[q][p];
    const int id = a.id_src[p];
    a.id_dst[t] = id;
    a.inv[id] = t;
    if (a.fun_src) {
        for (int q = 0; q < a.s; ++q) a.fun_dst[(size_t)t * a.s + q] = a.fun_src[(size_t)p * a.s + q];
    }
}

int sphx_permute_state(sphx_ctx* ctx, int64_t n) {
    StateArrays& a = ctx->st;
    StateArrays& b = ctx->alt;
    DevBuf* src[] = {&a.x, &a.y, &a.z, &a.vx, &a.vy, &a.vz, &a.ax, &a.ay, &a.az,
                     &a.m, &a.T, &a.mu, &a.gam, &a.E, &a.hprev, &a.ptype};
    DevBuf* dst[] = {&b.x, &b.y, &b.z, &b.vx, &b.vy, &b.vz, &b.ax, &b.ay, &b.az,
                     &b.m, &b.T, &b.mu, &b.gam, &b.E, &b.hprev, &b.ptype};
    GatherArgs g;
    g.n = (int)n; g.narr = 16; g.s = ctx->s;
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
    SPHX_TRY(sphx_ensure(ctx, ctx->inv, (size_t)n * sizeof(int)));
    g.id_src = a.id.as<int>(); g.id_dst = b.id.as<int>(); g.inv = ctx->inv.as<int>();
    g.fun_src = nullptr; g.fun_dst = nullptr;
    if (ctx->s > 0 && a.fun.p) {
        SPHX_TRY(sphx_ensure(ctx, b.fun, (size_t)n * ctx->s * sizeof(double)));
        g.fun_src = a.fun.as<double>(); g.fun_dst = b.fun.as<double>();
    }
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g);
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
                                                   unsigned* ticket, double* out_sum, double* out_cnt) {
    __shared__ double sm[4], sc[4];
    __shared__ bool last;
    double s = 0.0, c = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = h[i];
        if (v > 0.0 && (clip <= 0.0 || v <= clip)) { s += v; c += 1.0; }
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
                       out, cnt);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- drv:222-229: dt from the crossing time ------------------------------------------------
__global__ void dt_kernel(u64* ct_bits, double* dt_out, int first, double fixed_dt, double dt_0,
                          double max_age) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double dt;
    const u64 b = *ct_bits;
    *ct_bits = 0x7F7F7F7F7F7F7F7Full;          // "none yet" again: the next step's pass 2 needs no memset
    if (fixed_dt > 0.0) {
        dt = fixed_dt;
    } else {
        double ct = (b == 0x7F7F7F7F7F7F7F7Full) ? dt_0 / 10.0                       // nsc:783-784
                                                 : __longlong_as_double((long long)b) + 0.0001;  // nsc:786
        dt = first ? dt_0 / 10.0 : fmax(dt_0 / 5.0, fmin(dt_0 * 2.0, ct));           // drv:223-226
        if (ct > max_age) dt = max_age / 100.0;                                      // drv:228-229
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
};
__global__ __launch_bounds__(256) void integrate_kernel(IntegArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double dt = *a.dt;
    const double g = (a.ptype[i] == 0.0) ? 1.0 : 0.0;
    const double v[3] = {a.vx[i], a.vy[i], a.vz[i]};
    double pa[3], vis[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (a.G) {                                            // loop forms carry the physical sign themselves
            pa[c] = nan_to_num(a.G[3 * (size_t)i + c] / a.rho[i] * g);      // drv:460
            vis[c] = a.va[3 * (size_t)i + c];                               // av[0], drv:473
        } else {
            pa[c] = nan_to_num(-a.ha[3 * (size_t)i + c] * g);     // physical sign, drv:460
            vis[c] = nan_to_num(-a.va[3 * (size_t)i + c] * g);
        }
        if (a.drag_on) {                                      // drv:462-463,473
            const double dg = nan_to_num(a.drag_on[3 * (size_t)i + c] * a.rhod[i] / a.rho[i] * g);
            vis[c] = dg + nan_to_num(a.drag_re[3 * (size_t)i + c]) + vis[c];
        }
    }
    const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double an = sqrt(vis[0] * vis[0] + vis[1] * vis[1] + vis[2] * vis[2]);
    if (vn - an * dt < 0.0) {                                 // drv:475
#pragma unroll
        for (int c = 0; c < 3; ++c) vis[c] = -v[c] / dt;
    }
    double* P[3] = {a.x, a.y, a.z};
    double* V[3] = {a.vx, a.vy, a.vz};
    double* A[3] = {a.ax, a.ay, a.az};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // drv:477: grav_accel + pressure_accel + visc_accel, added left to right
        const double tot = a.grav ? (a.grav[3 * (size_t)i + c] + pa[c]) + vis[c] : pa[c] + vis[c];
        const double old = A[c][i];
        P[c][i] = P[c][i] + (tot * (dt * dt)) / 2.0 + v[c] * dt;   // drv:481
        V[c][i] = v[c] + (tot + old) / 2.0 * dt;              // drv:482-486
        A[c][i] = tot;
    }
    const double E = nan_to_num(a.E[i]) + nan_to_num(a.vh[i] * dt);          // drv:490
    a.E[i] = E;
    a.T[i] = nan_to_num(E * (a.mu[i] * a.m_h) / (a.gam[i] * a.m[i] * a.kB));  // drv:491
}

int sphx_integrate(sphx_ctx* ctx, int64_t n) {
    StateArrays& s = ctx->st;
    IntegArgs a;
    a.n = (int)n;
    a.x = s.x.as<double>(); a.y = s.y.as<double>(); a.z = s.z.as<double>();
    a.vx = s.vx.as<double>(); a.vy = s.vy.as<double>(); a.vz = s.vz.as<double>();
    a.ax = s.ax.as<double>(); a.ay = s.ay.as<double>(); a.az = s.az.as<double>();
    a.E = s.E.as<double>(); a.T = s.T.as<double>();
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
    hipLaunchKernelGGL(integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
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
