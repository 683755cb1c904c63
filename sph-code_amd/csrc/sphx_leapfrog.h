// sphx_leapfrog.h - the reference driver's time-step rule and leapfrog update as device functions
// (drv:223-229, drv:233-238, drv:472-491), shared by the fused step loop (sphx_integrate.hip), the
// device-pointer API (sphx_dev.hip) and the array entry points sphx_dt_rule / sphx_leapfrog.
// Pinned bit for bit by tests/golden/driver_integrator.npz: the two statement blocks of the driver,
// executed by tests/golden/make_golden_driver.py on seeded arrays.
// Include only from translation units compiled with `#pragma clang fp contract(off)`: NumPy never
// fuses a multiply into an add.
#pragma once
#include <float.h>

__device__ __forceinline__ double sphx_nan_to_num(double v) {
    if (v != v) return 0.0;
    if (v > DBL_MAX) return DBL_MAX;
    if (v < -DBL_MAX) return -DBL_MAX;
    return v;
}

// drv:223-229.  ct = nsc.crossing_time's return value (nsc:783-786).
__device__ __forceinline__ double sphx_dt_rule(double ct, int first, double dt_0, double max_age) {
    double dt = first ? dt_0 / 10.0 : fmax(dt_0 / 5.0, fmin(dt_0 * 2.0, ct));   // drv:223-226
    if (ct > max_age) dt = max_age / 100.0;                                     // drv:228-229
    return dt;
}

// nsc:783-786 from the device-side minimum: `none` = no gas particle voted -> dt_0/10, else min + 1e-4
__device__ __forceinline__ double sphx_ct_value(bool none, double ct_min, double dt_0) {
    return none ? dt_0 / 10.0 : ct_min + 0.0001;
}

// drv:233-234,237 (the assignments at drv:235-236 test the already clamped positions: no effect)
__device__ __forceinline__ double sphx_clamp_pos(double q, double lim) {
    q = (q > lim) ? lim : q;
    q = (q < -lim) ? -lim : q;
    return sphx_nan_to_num(q);
}

// drv:475-486 for one particle.  vis = viscous acceleration before the limiter (drv:473), pa =
// pressure acceleration (drv:460), grav / old nullable (old == nullptr: shapes differ, drv:484-485).
// x, v are updated in place; tot receives the new total acceleration.
__device__ __forceinline__ void sphx_leapfrog_update(double dt, double x[3], double v[3], double vis[3],
                                                     const double pa[3], const double* grav, const double* old,
                                                     double tot[3]) {
    const double vn = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const double an = sqrt((vis[0] * vis[0] + vis[1] * vis[1]) + vis[2] * vis[2]);
    if (vn - an * dt < 0.0) {                                                   // drv:475
#pragma unroll
        for (int c = 0; c < 3; ++c) vis[c] = -v[c] / dt;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // drv:477: grav_accel + pressure_accel + visc_accel, added left to right
        tot[c] = grav ? (grav[c] + pa[c]) + vis[c] : pa[c] + vis[c];
        x[c] = x[c] + ((tot[c] * (dt * dt)) / 2.0 + v[c] * dt);                 // drv:481: points += (a dt^2/2 + v dt)
        v[c] = old ? v[c] + (tot[c] + old[c]) / 2.0 * dt                        // drv:482-483
                   : v[c] + tot[c] * dt;                                        // drv:484-485
    }
}

__device__ __forceinline__ bool sphx_finite(double v) { return v - v == 0.0; }
// failure counters of the update kernels: one ballot per class, one atomic per wave that met any (never in a sane run)
// (counters: [64 buckets][16] u64, the bucket chosen by wave - sphx_internal.h BADC_*; 1-D launches of 256 threads)
__device__ __forceinline__ void sphx_count_bad(unsigned long long* counters, int which, bool bad) {
    const unsigned long long m = __ballot(bad);
    if (m && (unsigned long long)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) == 0ull && bad) {
        const unsigned bucket = ((blockIdx.x << 2) | (threadIdx.x >> 6)) & 63u;
        atomicAdd(&counters[bucket * 16u + (unsigned)which], (unsigned long long)__popcll(m));
    }
}

// drv:490-491
__device__ __forceinline__ void sphx_energy_update(double dt, double heat, double mu, double gam, double m,
                                                   double m_h, double kB, double& E, double& T) {
    E = sphx_nan_to_num(E) + sphx_nan_to_num(heat * dt);
    T = sphx_nan_to_num(E * (mu * m_h) / (gam * m * kB));
}
