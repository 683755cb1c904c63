// sphx_gravity.hip - self-gravity by direct summation with Plummer softening.
//
// The reference's gravity (nsc:252-415, grav_force_calculation_new) sums, for every particle, the
// monopoles  G m_c (c - x) / (|c - x|^2 + eps^2)^(3/2)  of a few kd-tree nodes, with eps = median of
// the smoothing lengths (nsc:358).  It cannot be run here (Python-2 idioms on SciPy's old pure-Python
// KDTree), and its node selection has no published definition to restate, so there is nothing to pin an
// approximation against.  What IS well defined is the sum it approximates: every particle taken as its
// own monopole with the same softening.  That exact sum is computed here - the known-answer test any
// tree code is validated against, and a usable solver up to a few 10^5 particles (O(N^2): 5 ms at
// 10^5, 0.5 s at 10^6 on MI355X).
//
// One thread per target, sources streamed through LDS in tiles of 256 {x,y,z,m} (32 B); the sum runs
// over sources in storage order (deterministic).  1 / r^3 is rsqrt-based: v_rsq_f64 plus two Newton
// steps (relative error < 1e-15).  No MFMA: the pair kernel is not a contraction (r^-3 of a difference).
#include "sphx_internal.h"
#include <rocprim/rocprim.hpp>

#define GRAV_TILE 256

__global__ __launch_bounds__(GRAV_TILE) void gravity_direct_kernel(int n, const double* __restrict__ x,
                                                                   const double* __restrict__ y,
                                                                   const double* __restrict__ z, int ps,
                                                                   const double* __restrict__ m,
                                                                   const double* eps_ptr, double eps_val, double G,
                                                                   const int* __restrict__ omap, double* acc) {
    __shared__ double4 tile[GRAV_TILE];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double eps = eps_ptr ? *eps_ptr : eps_val;
    const double e2 = eps * eps;
    double xi = 0.0, yi = 0.0, zi = 0.0;
    if (i < n) { xi = x[(size_t)i * ps]; yi = y[(size_t)i * ps]; zi = z[(size_t)i * ps]; }
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int j0 = 0; j0 < n; j0 += GRAV_TILE) {
        const int j = j0 + threadIdx.x;
        double4 s = make_double4(0.0, 0.0, 0.0, 0.0);          // padding sources have no mass
        if (j < n) s = make_double4(x[(size_t)j * ps], y[(size_t)j * ps], z[(size_t)j * ps], m[j]);
        __syncthreads();
        tile[threadIdx.x] = s;
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < GRAV_TILE; ++t) {
            const double4 q = tile[t];
            const double dx = q.x - xi, dy = q.y - yi, dz = q.z - zi;
            const double r2 = dx * dx + dy * dy + dz * dz + e2;
            // r2^(-3/2); a coincident pair with eps = 0 (r2 == 0) contributes nothing
            double inv = r2 > 0.0 ? rsqrt(r2) : 0.0;
            const double w = q.w * (inv * inv * inv);
            ax += w * dx; ay += w * dy; az += w * dz;
        }
    }
    if (i < n) {
        const int o = omap ? omap[i] : i;
        acc[3 * (size_t)o] = G * ax; acc[3 * (size_t)o + 1] = G * ay; acc[3 * (size_t)o + 2] = G * az;
    }
}

// median of h (n values) -> *out, NumPy's definition (mean of the two middle values for even n)
__global__ void median_pick_kernel(int n, const double* sorted, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *out = (n & 1) ? sorted[n / 2] : 0.5 * (sorted[n / 2 - 1] + sorted[n / 2]);
}

int sphx_median(sphx_ctx* ctx, int64_t n, const double* v, double* out_dev) {
    SPHX_TRY(sphx_ensure(ctx, ctx->grav_sort, (size_t)n * sizeof(double)));
    size_t tmp = 0;
    HIPCHK(rocprim::radix_sort_keys(nullptr, tmp, v, ctx->grav_sort.as<double>(), (size_t)n, 0, 64, ctx->stream));
    SPHX_TRY(sphx_ensure(ctx, ctx->grav_tmp, tmp));
    HIPCHK(rocprim::radix_sort_keys(ctx->grav_tmp.p, tmp, v, ctx->grav_sort.as<double>(), (size_t)n, 0, 64,
                                    ctx->stream));
    hipLaunchKernelGGL(median_pick_kernel, dim3(1), dim3(64), 0, ctx->stream, (int)n, ctx->grav_sort.as<double>(),
                       out_dev);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// device arrays; eps from device memory (eps_dev) or by value
int sphx_gravity_launch(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z, int ps,
                        const double* m, const double* eps_dev, double eps, double G, const int* omap,
                        double* acc) {
    hipLaunchKernelGGL(gravity_direct_kernel, dim3((unsigned)((n + GRAV_TILE - 1) / GRAV_TILE)), dim3(GRAV_TILE), 0,
                       ctx->stream, (int)n, x, y, z, ps, m, eps_dev, eps, G, omap, acc);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

extern "C" int sphx_gravity_direct(sphx_ctx* ctx, int64_t n, const double* mass, const double* points,
                                   const double* sizes, double softening, double G, double* accel) {
    if (!ctx) return SPHX_E_ARG;
    if (!mass || !points || !accel) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_gravity_direct: NULL argument");
    if (n < 1 || n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    if (!sizes && !(softening >= 0.0)) return sphx_set_err(ctx, SPHX_E_ARG, "softening must be >= 0");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb = (size_t)n * sizeof(double);
    SPHX_TRY(sphx_ensure(ctx, ctx->in_a, 3 * nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_b, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->out_b, 3 * nb));
    HIPCHK(hipMemcpyAsync(ctx->in_a.p, points, 3 * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->in_b.p, mass, nb, hipMemcpyHostToDevice, ctx->stream));
    const double* eps_dev = nullptr;
    if (sizes) {                                   // eps = median(sizes), nsc:358
        SPHX_TRY(sphx_ensure(ctx, ctx->in_c, nb));
        HIPCHK(hipMemcpyAsync(ctx->in_c.p, sizes, nb, hipMemcpyHostToDevice, ctx->stream));
        double* slot = ctx->scal.as<double>() + SC_GRAV_EPS;
        SPHX_TRY(sphx_median(ctx, n, ctx->in_c.as<double>(), slot));
        eps_dev = slot;
    }
    const double* p = ctx->in_a.as<double>();
    SPHX_TRY(sphx_gravity_launch(ctx, n, p, p + 1, p + 2, 3, ctx->in_b.as<double>(), eps_dev, softening, G, nullptr,
                                 ctx->out_b.as<double>()));
    HIPCHK(hipMemcpyAsync(accel, ctx->out_b.p, 3 * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}
