// sphx_loopforms.hip - the per-particle loop forms the reference's time loop calls
// (nsc:673-816; drv:451-458): mass-derived smoothing length h(m) = (m/m_0)^(1/3) d with the
// driver-injected global d, clipped gradients, physical sign.  One thread per particle over
// the K-major neighbour list; deltas are relative to the particle itself.
#include "sphx_internal.h"
// NumPy never fuses a multiply into an add: keep every operation separately rounded so that
// cancellations such as h_j^2 - r^2 at the kernel edge reproduce the reference bit for bit.
#pragma clang fp contract(off)
#include <float.h>

#define PI64 201.06192982974676      /* 64 pi */

__device__ __forceinline__ double nan_to_num_d(double v) {
    if (v != v) return 0.0;
    if (v > DBL_MAX) return DBL_MAX;
    if (v < -DBL_MAX) return -DBL_MAX;
    return v;
}
__device__ __forceinline__ double pow9(double d) {
    double d2 = d * d, d4 = d2 * d2;
    return d4 * d4 * d;
}
// nsc:673-676  m*315*(m_0/m)^3*((m/m_0)^(2/3) d^2 - r^2)^3/(64 pi d^9)
__device__ __forceinline__ double weigh2(double r2, double m, double d, double m0, double d9) {
    const double a = m0 / m;
    const double q = pow(m / m0, 2.0 / 3.0) * (d * d) - r2;
    return m * 315.0 * (a * a * a) * (q * q * q) / (PI64 * d9);
}
// nsc:678-681
__device__ __forceinline__ double weigh2_dust(double r2, double m, double ds) {
    const double q = ds * ds - r2;
    return m * 315.0 * (q * q * q) / (PI64 * pow9(ds));
}
// nsc:683-690: scalar coefficient of (x - x_0); clipped, gas neighbours only
__device__ __forceinline__ double grad_coef(double r2, double m, double d, double m0, double d9, double tk) {
    const double a = m0 / m;
    const double q = pow(m / m0, 2.0 / 3.0) * (d * d) - r2;
    double c = -315.0 * 6.0 * (a * a * a) / (PI64 * d9) * (q * q) * ((tk == 0.0) ? 1.0 : 0.0);
    return (q > 0.0) ? c : 0.0;
}

struct LoopArgs {
    int n, npad, k;
    const int* nbr;
    const double *pos, *vel;                 // (n,3) AoS, as the reference passes them
    const double *m, *pt, *h, *mu, *gam, *E, *T, *rho, *mgm, *mcs;
    double d, m0, m_h, kB, amu;
    double *out1, *out3, *out3b;             // (n,), (n,3), (n,3)
    u64* ct_bits;
};

// mode 0 density (nsc:693), 1 dust_density (nsc:704), 2 num_dens (nsc:744)
template <int MODE>
__global__ __launch_bounds__(256) void loop_density_kernel(LoopArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double xi = a.pos[3 * (size_t)i], yi = a.pos[3 * (size_t)i + 1], zi = a.pos[3 * (size_t)i + 2];
    const double d9 = pow9(a.d);
    double s = 0.0;
    for (int kk = 0; kk < a.k; ++kk) {
        int j = a.nbr[(size_t)kk * a.npad + i];
        if (j < 0) continue;
        const double dx = a.pos[3 * (size_t)j] - xi, dy = a.pos[3 * (size_t)j + 1] - yi,
                     dz = a.pos[3 * (size_t)j + 2] - zi;
        const double r2 = dx * dx + dy * dy + dz * dz;
        double v;
        if (MODE == 0) v = weigh2(r2, a.m[j], a.d, a.m0, d9) * ((a.pt[j] == 0.0) ? 1.0 : 0.0);
        else if (MODE == 1) v = weigh2_dust(r2, a.m[j], a.h[j]) * ((a.pt[j] == 2.0) ? 1.0 : 0.0);
        else v = weigh2(r2, a.m[j], a.d, a.m0, d9) / (a.mu[j] * a.m_h);
        if (v > 0.0) s += v;
    }
    a.out1[i] = s;
}

// nsc:755-774
__global__ __launch_bounds__(256) void loop_del_pressure_kernel(LoopArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    double gx = 0.0, gy = 0.0, gz = 0.0;
    if (a.pt[i] == 0.0) {
        const double xi = a.pos[3 * (size_t)i], yi = a.pos[3 * (size_t)i + 1], zi = a.pos[3 * (size_t)i + 2];
        const double d9 = pow9(a.d), Ei = a.E[i];
        for (int kk = 0; kk < a.k; ++kk) {
            int j = a.nbr[(size_t)kk * a.npad + i];
            if (j < 0) continue;
            const double dx = a.pos[3 * (size_t)j] - xi, dy = a.pos[3 * (size_t)j + 1] - yi,
                         dz = a.pos[3 * (size_t)j + 2] - zi;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double c = grad_coef(r2, a.m[j], a.d, a.m0, d9, a.pt[j]);
            const double f = (a.E[j] + Ei) / a.gam[j];
            gx += 0.5 * nan_to_num_d(c * dx) * f;
            gy += 0.5 * nan_to_num_d(c * dy) * f;
            gz += 0.5 * nan_to_num_d(c * dz) * f;
        }
    }
    a.out3[3 * (size_t)i] = gx; a.out3[3 * (size_t)i + 1] = gy; a.out3[3 * (size_t)i + 2] = gz;
}

// nsc:788-816
__global__ __launch_bounds__(256) void loop_av_kernel(LoopArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    double ax = 0.0, ay = 0.0, az = 0.0, heat = 0.0;
    if (a.pt[i] == 0.0) {
        const double xi = a.pos[3 * (size_t)i], yi = a.pos[3 * (size_t)i + 1], zi = a.pos[3 * (size_t)i + 2];
        const double vxi = a.vel[3 * (size_t)i], vyi = a.vel[3 * (size_t)i + 1], vzi = a.vel[3 * (size_t)i + 2];
        const double d9 = pow9(a.d), mi = a.m[i], rhoi = a.rho[i];
        const double csi = nan_to_num_d(sqrt(a.gam[i] * a.kB * a.T[i] / (a.mu[i] * a.amu)));
        for (int kk = 0; kk < a.k; ++kk) {
            int j = a.nbr[(size_t)kk * a.npad + i];
            if (j < 0 || a.pt[j] != 0.0) continue;            // sums run over gas neighbours only
            const double dx = a.pos[3 * (size_t)j] - xi, dy = a.pos[3 * (size_t)j + 1] - yi,
                         dz = a.pos[3 * (size_t)j + 2] - zi;
            const double dvx = a.vel[3 * (size_t)j] - vxi, dvy = a.vel[3 * (size_t)j + 1] - vyi,
                         dvz = a.vel[3 * (size_t)j + 2] - vzi;
            const double r2 = dx * dx + dy * dy + dz * dz;
            double w = (dvx * dx + dvy * dy + dvz * dz) / sqrt(r2);
            w = (w > 0.0) ? 0.0 : w;
            w = nan_to_num_d(w);                              // self pair: 0/0
            const double csj = nan_to_num_d(sqrt(a.gam[j] * a.kB * a.T[j] / (a.mu[j] * a.amu)));
            const double vsig = csj + csi - 3.0 * w;
            const double rho_ij = (a.rho[j] + rhoi) / 2.0;
            const double PI = -0.5 * vsig * w / rho_ij;
            const double c = grad_coef(r2, mi, a.d, a.m0, d9, a.pt[j]);     // h(m_i)  nsc:805
            const double gwx = nan_to_num_d(c * dx), gwy = nan_to_num_d(c * dy), gwz = nan_to_num_d(c * dz);
            const double mb = (a.m[j] + mi) / 2.0;
            ax += mb * PI * gwx; ay += mb * PI * gwy; az += mb * PI * gwz;
            heat += 0.5 * mb * PI * (dvx * gwx + dvy * gwy + dvz * gwz);
        }
    }
    a.out3[3 * (size_t)i] = ax; a.out3[3 * (size_t)i + 1] = ay; a.out3[3 * (size_t)i + 2] = az;
    a.out1[i] = heat;
}

// nsc:776-786
__global__ __launch_bounds__(256) void loop_ct_kernel(LoopArgs a) {
    __shared__ u64 sm[4];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    u64 mine = 0x7FF0000000000000ull;
    if (i < a.n && a.pt[i] == 0.0) {
        const double vxi = a.vel[3 * (size_t)i], vyi = a.vel[3 * (size_t)i + 1], vzi = a.vel[3 * (size_t)i + 2];
        double mx = 0.0;
        for (int kk = 0; kk < a.k; ++kk) {
            int j = a.nbr[(size_t)kk * a.npad + i];
            if (j < 0) continue;
            const double dvx = a.vel[3 * (size_t)j] - vxi, dvy = a.vel[3 * (size_t)j + 1] - vyi,
                         dvz = a.vel[3 * (size_t)j + 2] - vzi;
            mx = fmax(mx, dvx * dvx + dvy * dvy + dvz * dvz);
        }
        double ct = nan_to_num_d(a.h[i] / sqrt(mx));
        if (ct > 0.0) mine = (u64)__double_as_longlong(ct);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u64 p = __shfl_xor(mine, o, 64);
        mine = p < mine ? p : mine;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 r = sm[0];
        for (int w = 1; w < 4; ++w) r = sm[w] < r ? sm[w] : r;
        if (r != 0x7FF0000000000000ull) atomicMin(a.ct_bits, r);
    }
}

// nsc:719-742 (reaction is a scatter-add: float atomics, order not reproducible bit for bit)
__global__ __launch_bounds__(256) void loop_impulse_kernel(LoopArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double xi = a.pos[3 * (size_t)i], yi = a.pos[3 * (size_t)i + 1], zi = a.pos[3 * (size_t)i + 2];
    const double vxi = a.vel[3 * (size_t)i], vyi = a.vel[3 * (size_t)i + 1], vzi = a.vel[3 * (size_t)i + 2];
    double ox = 0.0, oy = 0.0, oz = 0.0;
    for (int kk = 0; kk < a.k; ++kk) {
        int j = a.nbr[(size_t)kk * a.npad + i];
        if (j < 0 || a.pt[j] != 2.0) continue;
        const double dx = a.pos[3 * (size_t)j] - xi, dy = a.pos[3 * (size_t)j + 1] - yi,
                     dz = a.pos[3 * (size_t)j + 2] - zi;
        const double wf = weigh2_dust(dx * dx + dy * dy + dz * dz, a.m[j], a.h[j]);
        if (!(wf > 0.0)) continue;
        const double dvx = a.vel[3 * (size_t)j] - vxi, dvy = a.vel[3 * (size_t)j + 1] - vyi,
                     dvz = a.vel[3 * (size_t)j + 2] - vzi;
        const double coef = wf / a.mgm[j] * a.mcs[j] * sqrt(dvx * dvx + dvy * dvy + dvz * dvz);
        const double fx = coef * dvx, fy = coef * dvy, fz = coef * dvz;
        ox += fx; oy += fy; oz += fz;
        if (j != i) {
            atomicAdd(&a.out3b[3 * (size_t)j], -fx);
            atomicAdd(&a.out3b[3 * (size_t)j + 1], -fy);
            atomicAdd(&a.out3b[3 * (size_t)j + 2], -fz);
        }
    }
    a.out3[3 * (size_t)i] = ox; a.out3[3 * (size_t)i + 1] = oy; a.out3[3 * (size_t)i + 2] = oz;
}

// ---- host side ---------------------------------------------------------------------------------
static int up(sphx_ctx* ctx, DevBuf& b, const void* host, size_t bytes) {
    SPHX_TRY(sphx_ensure(ctx, b, bytes));
    HIPCHK(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SPHX_OK;
}
#define NEED(p)                                                                              \
    do {                                                                                     \
        if (!(p)) return sphx_set_err(ctx, SPHX_E_ARG, "%s: argument %s is NULL", __func__, #p); \
    } while (0)

static int loop_begin(sphx_ctx* ctx, int64_t n, int k, const int64_t* neighbor, LoopArgs* a) {
    if (n < 1 || n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    if (k < 1 || k > 4096) return sphx_set_err(ctx, SPHX_E_ARG, "k=%d out of range", k);
    HIPCHK(hipSetDevice(ctx->device));
    ctx->map_perm = nullptr;
    ctx->qorder = nullptr;
    SPHX_TRY(up(ctx, ctx->idx64, neighbor, (size_t)n * k * sizeof(int64_t)));
    SPHX_TRY(sphx_transpose_nbr(ctx, n, k, ctx->idx64.as<int64_t>()));
    memset(a, 0, sizeof(*a));
    a->n = (int)n; a->npad = (int)sphx_pad64(n); a->k = k;
    a->nbr = ctx->nbr.as<int>();
    a->m0 = ctx->cst.m_0; a->m_h = ctx->cst.m_h; a->kB = ctx->cst.k_B; a->amu = ctx->cst.amu;
    return SPHX_OK;
}
#define UPF(buf, host, cnt, field)                                                           \
    do {                                                                                     \
        SPHX_TRY(up(ctx, ctx->buf, host, (size_t)(cnt) * sizeof(double)));                   \
        a.field = ctx->buf.as<double>();                                                     \
    } while (0)
#define LAUNCH1(kern)                                                                        \
    do {                                                                                     \
        hipLaunchKernelGGL(kern, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a); \
        HIPCHK(hipGetLastError());                                                           \
    } while (0)
static int down(sphx_ctx* ctx, void* host, const void* dev, size_t bytes) {
    HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

extern "C" int sphx_density(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                            const double* particle_type, const int64_t* neighbor, double d, double* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(points); NEED(mass); NEED(particle_type); NEED(neighbor); NEED(out);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_a, points, 3 * n, pos); UPF(in_c, mass, n, m); UPF(in_h, particle_type, n, pt);
    a.d = d;
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * sizeof(double)));
    a.out1 = ctx->out_a.as<double>();
    LAUNCH1(loop_density_kernel<0>);
    return down(ctx, out, a.out1, (size_t)n * sizeof(double));
}

extern "C" int sphx_dust_density(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                                 const int64_t* neighbor, const double* particle_type,
                                 const double* sizes, double* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(points); NEED(mass); NEED(particle_type); NEED(neighbor); NEED(sizes); NEED(out);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_a, points, 3 * n, pos); UPF(in_c, mass, n, m); UPF(in_h, particle_type, n, pt);
    UPF(in_d, sizes, n, h);
    a.d = 1.0;
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * sizeof(double)));
    a.out1 = ctx->out_a.as<double>();
    LAUNCH1(loop_density_kernel<1>);
    return down(ctx, out, a.out1, (size_t)n * sizeof(double));
}

extern "C" int sphx_num_dens(sphx_ctx* ctx, int64_t n, int k, const double* mass, const double* points,
                             const double* mu_array, const int64_t* neighbor, double d, double* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(points); NEED(mass); NEED(mu_array); NEED(neighbor); NEED(out);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_a, points, 3 * n, pos); UPF(in_c, mass, n, m); UPF(in_f, mu_array, n, mu);
    a.d = d;
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * sizeof(double)));
    a.out1 = ctx->out_a.as<double>();
    LAUNCH1(loop_density_kernel<2>);
    return down(ctx, out, a.out1, (size_t)n * sizeof(double));
}

extern "C" int sphx_del_pressure(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                                 const double* particle_type, const int64_t* neighbor,
                                 const double* E_internal, const double* gamma_array, double d, double* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(points); NEED(mass); NEED(particle_type); NEED(neighbor); NEED(E_internal); NEED(gamma_array); NEED(out);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_a, points, 3 * n, pos); UPF(in_c, mass, n, m); UPF(in_h, particle_type, n, pt);
    UPF(in_e, E_internal, n, E); UPF(in_g, gamma_array, n, gam);
    a.d = d;
    SPHX_TRY(sphx_ensure(ctx, ctx->out_b, (size_t)n * 3 * sizeof(double)));
    a.out3 = ctx->out_b.as<double>();
    LAUNCH1(loop_del_pressure_kernel);
    return down(ctx, out, a.out3, (size_t)n * 3 * sizeof(double));
}

extern "C" int sphx_artificial_viscosity(sphx_ctx* ctx, int64_t n, int k, const int64_t* neighbor,
                                         const double* points, const double* particle_type,
                                         const double* sizes, const double* mass, const double* densities,
                                         const double* velocities, const double* T,
                                         const double* gamma_array, const double* mu_array, double d,
                                         double* visc_accel, double* visc_heat) {
    if (!ctx) return SPHX_E_ARG;
    (void)sizes;
    NEED(points); NEED(mass); NEED(particle_type); NEED(neighbor); NEED(densities); NEED(velocities);
    NEED(T); NEED(gamma_array); NEED(mu_array); NEED(visc_accel); NEED(visc_heat);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_a, points, 3 * n, pos); UPF(in_b, velocities, 3 * n, vel); UPF(in_c, mass, n, m);
    UPF(in_h, particle_type, n, pt); UPF(in_d, densities, n, rho); UPF(in_e, T, n, T);
    UPF(in_g, gamma_array, n, gam); UPF(in_f, mu_array, n, mu);
    a.d = d;
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->out_b, (size_t)n * 3 * sizeof(double)));
    a.out1 = ctx->out_a.as<double>(); a.out3 = ctx->out_b.as<double>();
    LAUNCH1(loop_av_kernel);
    SPHX_TRY(down(ctx, visc_accel, a.out3, (size_t)n * 3 * sizeof(double)));
    return down(ctx, visc_heat, a.out1, (size_t)n * sizeof(double));
}

extern "C" int sphx_crossing_time(sphx_ctx* ctx, int64_t n, int k, const int64_t* neighbor,
                                  const double* velocities, const double* sizes,
                                  const double* particle_type, double* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(neighbor); NEED(velocities); NEED(sizes); NEED(particle_type); NEED(out);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_b, velocities, 3 * n, vel); UPF(in_d, sizes, n, h); UPF(in_h, particle_type, n, pt);
    a.ct_bits = ctx->scal.as<u64>() + SC_CT_BITS;
    HIPCHK(hipMemsetAsync(a.ct_bits, 0x7F, sizeof(u64), ctx->stream));
    LAUNCH1(loop_ct_kernel);
    u64 bits = 0;
    SPHX_TRY(down(ctx, &bits, a.ct_bits, sizeof(u64)));
    if (bits == 0x7F7F7F7F7F7F7F7Full) {
        *out = ctx->cst.dt_0 / 10.0;                      // nsc:783-784
    } else {
        double v;
        memcpy(&v, &bits, 8);
        *out = v + 0.0001;                                // nsc:786
    }
    return SPHX_OK;
}

extern "C" int sphx_net_impulse(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                                const double* sizes, const double* velocities, const double* particle_type,
                                const int64_t* neighbor, const double* mean_grain_mass,
                                const double* mean_cross, double* accel_onto, double* accel_reaction) {
    if (!ctx) return SPHX_E_ARG;
    NEED(points); NEED(mass); NEED(sizes); NEED(velocities); NEED(particle_type); NEED(neighbor);
    NEED(mean_grain_mass); NEED(mean_cross); NEED(accel_onto); NEED(accel_reaction);
    LoopArgs a;
    SPHX_TRY(loop_begin(ctx, n, k, neighbor, &a));
    UPF(in_a, points, 3 * n, pos); UPF(in_b, velocities, 3 * n, vel); UPF(in_c, mass, n, m);
    UPF(in_h, particle_type, n, pt); UPF(in_d, sizes, n, h); UPF(in_e, mean_grain_mass, n, mgm);
    UPF(in_f, mean_cross, n, mcs);
    SPHX_TRY(sphx_ensure(ctx, ctx->out_b, (size_t)n * 3 * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->out_c, (size_t)n * 3 * sizeof(double)));
    a.out3 = ctx->out_b.as<double>(); a.out3b = ctx->out_c.as<double>();
    HIPCHK(hipMemsetAsync(a.out3b, 0, (size_t)n * 3 * sizeof(double), ctx->stream));
    LAUNCH1(loop_impulse_kernel);
    SPHX_TRY(down(ctx, accel_onto, a.out3, (size_t)n * 3 * sizeof(double)));
    return down(ctx, accel_reaction, a.out3b, (size_t)n * 3 * sizeof(double));
}

// ---- the loop forms on the device-resident, cell-sorted state of the step loop ------------------
// drv:451-458 as the reference's time loop evaluates them, on the step's own neighbour list:
//   rho = density, rho_dust = dust_density, n = num_dens, delp = del_pressure,
//   (av accel, av heat) = artificial_viscosity(..., densities = rho, ...), ct = crossing_time.
// Same expressions, in the same order of operations, as the array-API kernels above (which the golden
// vectors pin) - but gathered the way sphx_sums.hip gathers: per-particle factors ((m/m_0)^(2/3) d^2,
// (m_0/m)^3, the gradient prefactor, c_s, mu m_h) are worked out once per step into dense records, and
// the six sums run as two passes of one thread per particle with four neighbours in flight
// (6.5 ms -> ~1 ms per step at 1e6 particles: the array-API kernels issue one scattered 8-B load per
// field and neighbour and call pow() per neighbour).
struct RecLA { double x, y, z, hq, m, a3, g1, pt; };     // pass 1: position, h(m)^2, mass, (m_0/m)^3, grad prefactor, type
struct RecLB { double E, gam, mun, ds; };                // pass 1: E, gamma, mu m_h, sizes
struct RecLV { double x, y, z, pt, vx, vy, vz, cs; };    // pass 2: position, type, velocity, sound speed
struct RecLR { double rho, m; };                         // pass 2: density (from pass 1), mass

struct LoopPrepArgs {
    int n;
    const double *x, *y, *z, *vx, *vy, *vz, *m, *pt, *h, *mu, *gam, *E, *T;
    double d, m0, m_h, kB, amu;
    RecLA* la; RecLB* lb; RecLV* lv;
};
__global__ __launch_bounds__(256) void loop_prep_kernel(LoopPrepArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double m = a.m[i], d9 = pow9(a.d);
    const double aa = a.m0 / m;
    RecLA r;
    r.x = a.x[i]; r.y = a.y[i]; r.z = a.z[i];
    r.hq = pow(m / a.m0, 2.0 / 3.0) * (a.d * a.d);              // nsc:675
    r.m = m;
    r.a3 = aa * aa * aa;
    r.g1 = -315.0 * 6.0 * (aa * aa * aa) / (PI64 * d9);          // nsc:688 up to (q*q) and the type mask
    r.pt = a.pt[i];
    a.la[i] = r;
    RecLB b;
    b.E = a.E[i]; b.gam = a.gam[i]; b.mun = a.mu[i] * a.m_h; b.ds = a.h[i];
    a.lb[i] = b;
    RecLV v;
    v.x = r.x; v.y = r.y; v.z = r.z; v.pt = r.pt;
    v.vx = a.vx[i]; v.vy = a.vy[i]; v.vz = a.vz[i];
    v.cs = nan_to_num_d(sqrt(a.gam[i] * a.kB * a.T[i] / (a.mu[i] * a.amu)));     // nsc:792
    a.lv[i] = v;
}

struct Q4L { double a, b, c, d; };
__device__ __forceinline__ Q4L ld4(const double* p) {
    const double2 lo = *reinterpret_cast<const double2*>(p);
    const double2 hi = *reinterpret_cast<const double2*>(p + 2);
    return Q4L{lo.x, lo.y, hi.x, hi.y};
}
#define LNB 4

// pass 1: density (nsc:693), dust_density (nsc:704), num_dens (nsc:744), del_pressure (nsc:755)
__global__ __launch_bounds__(256) void loop_pass1_kernel(int n, int npad, int k, double d9,
                                                         const int* __restrict__ nbr,
                                                         const RecLA* __restrict__ la,
                                                         const RecLB* __restrict__ lb,
                                                         const int* __restrict__ qorder, double* rho, double* rhod,
                                                         double* nden, double* G, RecLR* lr) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;      // list column (blob order)
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;                                           // stored particle
    const double* sp = reinterpret_cast<const double*>(&la[i]);
    const Q4L s0 = ld4(sp), s1 = ld4(sp + 4);
    const double xi = s0.a, yi = s0.b, zi = s0.c, Ei = lb[i].E;
    const bool gas_i = (s1.d == 0.0);
    double s_rho = 0.0, s_d = 0.0, s_n = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    for (int kk0 = 0; kk0 < k; kk0 += LNB) {
        int jb[LNB];
        Q4L q0b[LNB], q1b[LNB], q2b[LNB];
#pragma unroll
        for (int u = 0; u < LNB; ++u) jb[u] = (kk0 + u < k) ? nbr[(size_t)(kk0 + u) * npad + p] : -1;
#pragma unroll
        for (int u = 0; u < LNB; ++u) {
            const int jj = jb[u] < 0 ? i : jb[u];
            const double* q = reinterpret_cast<const double*>(&la[jj]);
            q0b[u] = ld4(q); q1b[u] = ld4(q + 4);
            q2b[u] = ld4(reinterpret_cast<const double*>(&lb[jj]));
        }
#pragma unroll
        for (int u = 0; u < LNB; ++u) {
            if (jb[u] < 0) continue;
            const Q4L q0 = q0b[u], q1 = q1b[u], q2 = q2b[u];         // x y z hq | m a3 g1 pt | E gam mun ds
            const double dx = q0.a - xi, dy = q0.b - yi, dz = q0.c - zi;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double q = q0.d - r2;
            const double w = q1.a * 315.0 * q1.b * (q * q * q) / (PI64 * d9);        // Weigh2, nsc:673-676
            const double vr = w * ((q1.d == 0.0) ? 1.0 : 0.0);
            if (vr > 0.0) s_rho += vr;                                               // nsc:700
            const double vn = w / q2.c;
            if (vn > 0.0) s_n += vn;                                                 // nsc:751
            if (q1.d == 2.0) {                                                       // Weigh2_dust, nsc:678-681
                const double vd = weigh2_dust(r2, q1.a, q2.d);
                if (vd > 0.0) s_d += vd;                                             // nsc:715
            }
            if (gas_i) {                                                             // nsc:755-774
                double c = q1.c * (q * q) * ((q1.d == 0.0) ? 1.0 : 0.0);
                c = (q > 0.0) ? c : 0.0;
                const double f = (q2.a + Ei) / q2.b;
                gx += 0.5 * nan_to_num_d(c * dx) * f;
                gy += 0.5 * nan_to_num_d(c * dy) * f;
                gz += 0.5 * nan_to_num_d(c * dz) * f;
            }
        }
    }
    rho[i] = s_rho; rhod[i] = s_d; nden[i] = s_n;
    G[3 * (size_t)i] = gx; G[3 * (size_t)i + 1] = gy; G[3 * (size_t)i + 2] = gz;
    lr[i] = RecLR{s_rho, s1.a};
}

// pass 2: artificial_viscosity (nsc:788-816) and crossing_time (nsc:776-786)
__global__ __launch_bounds__(256) void loop_pass2_kernel(int n, int npad, int k,
                                                         const int* __restrict__ nbr,
                                                         const RecLA* __restrict__ la,
                                                         const RecLV* __restrict__ lv,
                                                         const RecLR* __restrict__ lr,
                                                         const double* __restrict__ h,
                                                         const int* __restrict__ qorder, double* va, double* vh,
                                                         u64* ct_bits) {
    __shared__ u64 sm[4];
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    u64 mine = 0x7FF0000000000000ull;
    if (p < n) {
        const int i = qorder ? qorder[p] : p;
        const double* sp = reinterpret_cast<const double*>(&lv[i]);
        const Q4L s0 = ld4(sp), sv = ld4(sp + 4);                    // x y z pt | vx vy vz cs
        double ax = 0.0, ay = 0.0, az = 0.0, heat = 0.0;
        if (s0.d == 0.0) {
            const double hq_i = la[i].hq, g1_i = la[i].g1;
            const RecLR ri = lr[i];
            double mx = 0.0;
            for (int kk0 = 0; kk0 < k; kk0 += LNB) {
                int jb[LNB];
                Q4L q0b[LNB], qvb[LNB];
                double2 rb[LNB];
#pragma unroll
                for (int u = 0; u < LNB; ++u) jb[u] = (kk0 + u < k) ? nbr[(size_t)(kk0 + u) * npad + p] : -1;
#pragma unroll
                for (int u = 0; u < LNB; ++u) {
                    const int jj = jb[u] < 0 ? i : jb[u];
                    const double* q = reinterpret_cast<const double*>(&lv[jj]);
                    q0b[u] = ld4(q); qvb[u] = ld4(q + 4);
                    rb[u] = *reinterpret_cast<const double2*>(&lr[jj]);
                }
#pragma unroll
                for (int u = 0; u < LNB; ++u) {
                    if (jb[u] < 0) continue;
                    const Q4L q0 = q0b[u], qv = qvb[u];
                    const double dvx = qv.a - sv.a, dvy = qv.b - sv.b, dvz = qv.c - sv.c;
                    mx = fmax(mx, dvx * dvx + dvy * dvy + dvz * dvz);                 // nsc:780 (every neighbour)
                    if (q0.d != 0.0) continue;                                        // sums over gas neighbours only
                    const double dx = q0.a - s0.a, dy = q0.b - s0.b, dz = q0.c - s0.c;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    double w = (dvx * dx + dvy * dy + dvz * dz) / sqrt(r2);
                    w = (w > 0.0) ? 0.0 : w;
                    w = nan_to_num_d(w);                                              // self pair: 0/0
                    const double vsig = qv.d + sv.d - 3.0 * w;
                    const double rho_ij = (rb[u].x + ri.rho) / 2.0;
                    const double PI = -0.5 * vsig * w / rho_ij;
                    const double q = hq_i - r2;                                       // h(m_i), nsc:805
                    double c = g1_i * (q * q) * 1.0;
                    c = (q > 0.0) ? c : 0.0;
                    const double gwx = nan_to_num_d(c * dx), gwy = nan_to_num_d(c * dy), gwz = nan_to_num_d(c * dz);
                    const double mb = (rb[u].y + ri.m) / 2.0;
                    ax += mb * PI * gwx; ay += mb * PI * gwy; az += mb * PI * gwz;
                    heat += 0.5 * mb * PI * (dvx * gwx + dvy * gwy + dvz * gwz);
                }
            }
            const double ct = nan_to_num_d(h[i] / sqrt(mx));
            if (ct > 0.0) mine = (u64)__double_as_longlong(ct);
        }
        va[3 * (size_t)i] = ax; va[3 * (size_t)i + 1] = ay; va[3 * (size_t)i + 2] = az;
        vh[i] = heat;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 p = __shfl_xor(mine, o, 64);
        mine = p < mine ? p : mine;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 r = sm[0];
        for (int w = 1; w < 4; ++w) r = sm[w] < r ? sm[w] : r;
        if (r != 0x7FF0000000000000ull) atomicMin(ct_bits, r);
    }
}

int sphx_loop_step_sums(sphx_ctx* ctx, int64_t n, int k, double d) {
    StateArrays& st = ctx->st;
    const size_t nb = (size_t)n * sizeof(double);
    DevBuf* outs1[] = {&ctx->rho, &ctx->rhod, &ctx->nden, &ctx->vh};
    for (DevBuf* b : outs1) SPHX_TRY(sphx_ensure(ctx, *b, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->G, 3 * nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->va, 3 * nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->lrec_a, (size_t)n * sizeof(RecLA)));
    SPHX_TRY(sphx_ensure(ctx, ctx->lrec_v, (size_t)n * sizeof(RecLV)));
    SPHX_TRY(sphx_ensure(ctx, ctx->lrec_b, (size_t)n * sizeof(RecLB)));
    SPHX_TRY(sphx_ensure(ctx, ctx->lrec_r, (size_t)n * sizeof(RecLR)));
    const unsigned grid = (unsigned)((n + 255) / 256);
    LoopPrepArgs p;
    p.n = (int)n;
    p.x = st.x.as<double>(); p.y = st.y.as<double>(); p.z = st.z.as<double>();
    p.vx = st.vx.as<double>(); p.vy = st.vy.as<double>(); p.vz = st.vz.as<double>();
    p.m = st.m.as<double>(); p.pt = st.ptype.as<double>(); p.h = st.hprev.as<double>();
    p.mu = st.mu.as<double>(); p.gam = st.gam.as<double>(); p.E = st.E.as<double>(); p.T = st.T.as<double>();
    p.d = d; p.m0 = ctx->cst.m_0; p.m_h = ctx->cst.m_h; p.kB = ctx->cst.k_B; p.amu = ctx->cst.amu;
    p.la = ctx->lrec_a.as<RecLA>(); p.lb = ctx->lrec_b.as<RecLB>(); p.lv = ctx->lrec_v.as<RecLV>();
    hipLaunchKernelGGL(loop_prep_kernel, dim3(grid), dim3(256), 0, ctx->stream, p);
    const double d2 = d * d, d4 = d2 * d2, d9 = d4 * d4 * d;               // pow9(d), as the kernels form it
    hipLaunchKernelGGL(loop_pass1_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, (int)sphx_pad64(n), k, d9,
                       ctx->nbr.as<int>(), p.la, p.lb, ctx->qorder, ctx->rho.as<double>(), ctx->rhod.as<double>(),
                       ctx->nden.as<double>(), ctx->G.as<double>(), ctx->lrec_r.as<RecLR>());
    u64* ct = ctx->scal.as<u64>() + SC_CT_BITS;
    HIPCHK(hipMemsetAsync(ct, 0x7F, sizeof(u64), ctx->stream));       // 0x7F7F.. = huge finite "none yet"
    hipLaunchKernelGGL(loop_pass2_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, (int)sphx_pad64(n), k,
                       ctx->nbr.as<int>(), p.la, p.lv, ctx->lrec_r.as<RecLR>(), st.hprev.as<double>(), ctx->qorder,
                       ctx->va.as<double>(), ctx->vh.as<double>(), ct);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
