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
#include <string.h>

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
            double inv = 0.0;
            if (r2 > 0.0) {            // hardware seed (2^-26) + two Newton steps: < 1e-15, without the library
                const double hr = 0.5 * r2;                    // routine's scaling and class checks (r2 is mid-range)
                double yv = __builtin_amdgcn_rsq(r2);
                yv = yv * __builtin_fma(-hr * yv, yv, 1.5);
                inv = yv * __builtin_fma(-hr * yv, yv, 1.5);
            }
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

// =================================================================================================
// Tree form: monopoles of a cell pyramid over the search grid
// =================================================================================================
// The fine grid of the neighbour search (cells of 0.6 mean h, particles stored cell by cell) is the
// leaf level; level l has cells of 2^l fine cells a side with (mass, centre of mass).  A particle in
// level-l cell C takes, at each level l >= 1, the monopoles of the children of its parent's
// neighbours (+-ws parents) that are not themselves neighbours of C (+-ws cells) - the interaction
// list of a uniform-grid FMM, at most (2(2ws+1))^3 - (2ws+1)^3 cells (189 at ws = 1, 875 at ws = 2) -
// and sums directly over the particles of the level-1 cells within +-ws of its own.  Every source is
// counted exactly once; softening (eps) applies to monopoles and particles alike, as in the
// reference.  ws = 1: ~1.5e3 terms per particle, ~1 % rms force error; ws = 2: ~7e3 terms, ~0.2 %.
// One thread per particle in processing (blob) order, so the lanes of a wave walk the same cells
// and their loads of a cell record coalesce into one request.
#define PYR_MAX 12
struct Pyr {
    int nlev;                                   // levels 1..nlev exist; level nlev is a single cell
    int nx[PYR_MAX + 1], ny[PYR_MAX + 1], nz[PYR_MAX + 1];
    long long off[PYR_MAX + 1];                 // first record of level l in the pyramid buffer
};

__global__ __launch_bounds__(256) void pyr_level1_kernel(GridParams g, Pyr py, const int* __restrict__ cell_start,
                                                         const double* __restrict__ x, const double* __restrict__ y,
                                                         const double* __restrict__ z, const double* __restrict__ m,
                                                         double4* pyr, double4* quad) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int nx1 = py.nx[1], ny1 = py.ny[1], nz1 = py.nz[1];
    if (c >= nx1 * ny1 * nz1) return;
    const int X = c % nx1, Y = (c / nx1) % ny1, Z = c / (nx1 * ny1);
    const int fx0 = 2 * X, fx1 = min(2 * X + 2, g.nx);
    double sm = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    for (int fz = 2 * Z; fz < min(2 * Z + 2, g.nz); ++fz)
        for (int fy = 2 * Y; fy < min(2 * Y + 2, g.ny); ++fy) {
            const int row = (fz * g.ny + fy) * g.nx;
            const int s = cell_start[row + fx0], e = cell_start[row + fx1];
            for (int j = s; j < e; ++j) {
                const double mj = m[j];
                sm += mj; sx += mj * x[j]; sy += mj * y[j]; sz += mj * z[j];
            }
        }
    const double inv = sm > 0.0 ? 1.0 / sm : 0.0;
    const double cxm = sx * inv, cym = sy * inv, czm = sz * inv;
    pyr[py.off[1] + c] = make_double4(sm, cxm, cym, czm);
    if (quad) {
        // second moments about the centre of mass, S_ab = sum m d_a d_b (not made traceless: the
        // softened kernel is not harmonic, its expansion keeps the trace)
        double qxx = 0.0, qyy = 0.0, qzz = 0.0, qxy = 0.0, qxz = 0.0, qyz = 0.0;
        for (int fz = 2 * Z; fz < min(2 * Z + 2, g.nz); ++fz)
            for (int fy = 2 * Y; fy < min(2 * Y + 2, g.ny); ++fy) {
                const int row = (fz * g.ny + fy) * g.nx;
                const int s = cell_start[row + fx0], e = cell_start[row + fx1];
                for (int j = s; j < e; ++j) {
                    const double mj = m[j], dx = x[j] - cxm, dy = y[j] - cym, dz = z[j] - czm;
                    qxx += mj * dx * dx; qyy += mj * dy * dy; qzz += mj * dz * dz;
                    qxy += mj * dx * dy; qxz += mj * dx * dz; qyz += mj * dy * dz;
                }
            }
        quad[2 * (py.off[1] + c)] = make_double4(qxx, qyy, qzz, qxy);
        quad[2 * (py.off[1] + c) + 1] = make_double4(qxz, qyz, 0.0, 0.0);
    }
}

__global__ __launch_bounds__(256) void pyr_up_kernel(Pyr py, int l, double4* pyr, double4* quad) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int nxl = py.nx[l], nyl = py.ny[l], nzl = py.nz[l];
    if (c >= nxl * nyl * nzl) return;
    const int X = c % nxl, Y = (c / nxl) % nyl, Z = c / (nxl * nyl);
    const int cx = py.nx[l - 1], cy = py.ny[l - 1], cz = py.nz[l - 1];
    const double4* ch = pyr + py.off[l - 1];
    double sm = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
    for (int k = 2 * Z; k < min(2 * Z + 2, cz); ++k)
        for (int j = 2 * Y; j < min(2 * Y + 2, cy); ++j)
            for (int i = 2 * X; i < min(2 * X + 2, cx); ++i) {
                const double4 q = ch[((size_t)k * cy + j) * cx + i];
                sm += q.x; sx += q.x * q.y; sy += q.x * q.z; sz += q.x * q.w;
            }
    const double inv = sm > 0.0 ? 1.0 / sm : 0.0;
    const double cxm = sx * inv, cym = sy * inv, czm = sz * inv;
    pyr[py.off[l] + c] = make_double4(sm, cxm, cym, czm);
    if (quad) {
        // S = sum over children of S_child + M_child s s^T, s = child's centre - this centre
        const double4* qh = quad + 2 * py.off[l - 1];
        double qxx = 0.0, qyy = 0.0, qzz = 0.0, qxy = 0.0, qxz = 0.0, qyz = 0.0;
        for (int k = 2 * Z; k < min(2 * Z + 2, cz); ++k)
            for (int j = 2 * Y; j < min(2 * Y + 2, cy); ++j)
                for (int i = 2 * X; i < min(2 * X + 2, cx); ++i) {
                    const size_t cc = ((size_t)k * cy + j) * cx + i;
                    const double4 q = ch[cc];
                    if (!(q.x > 0.0)) continue;
                    const double4 b = qh[2 * cc], e = qh[2 * cc + 1];
                    const double dx = q.y - cxm, dy = q.z - cym, dz = q.w - czm;
                    qxx += b.x + q.x * dx * dx; qyy += b.y + q.x * dy * dy; qzz += b.z + q.x * dz * dz;
                    qxy += b.w + q.x * dx * dy; qxz += e.x + q.x * dx * dz; qyz += e.y + q.x * dy * dz;
                }
        quad[2 * (py.off[l] + c)] = make_double4(qxx, qyy, qzz, qxy);
        quad[2 * (py.off[l] + c) + 1] = make_double4(qxz, qyz, 0.0, 0.0);
    }
}

// Monopole + second moments of a cell (M at c, S_ab = sum m d_a d_b about c) on a particle at x, r = c - x:
// the Taylor expansion of the SOFTENED kernel 1 / sqrt(r^2 + eps^2) to second order, s^2 = r^2 + eps^2,
//   a = (M / s^3 + 15/2 (r.S.r) / s^7 - 3/2 tr(S) / s^5) r - 3 (S.r) / s^5
// (for eps = 0 this is the usual traceless-quadrupole term).  f = 1 for a cell of the lane's list, else 0.
// Needs s2 > 0: softening, or a cell that is not the particle's own - a well-separated cell never is.
__device__ __forceinline__ void grav_term_quad(double4 a4, double4 b4, double4 c4, double f, double xi, double yi,
                                               double zi, double e2, double& ax, double& ay, double& az) {
    const double dx = a4.y - xi, dy = a4.z - yi, dz = a4.w - zi;
    const double r2 = dx * dx + dy * dy + dz * dz + e2;
    const double y0 = __builtin_amdgcn_rsq(r2);
    const double inv = y0 * __builtin_fma(-0.5 * r2 * y0, y0, 1.5);
    const double inv2 = inv * inv, inv3 = inv * inv2, inv5 = inv3 * inv2, inv7 = inv5 * inv2;
    const double srx = b4.x * dx + b4.w * dy + c4.x * dz;
    const double sry = b4.w * dx + b4.y * dy + c4.y * dz;
    const double srz = c4.x * dx + c4.y * dy + b4.z * dz;
    const double rsr = dx * srx + dy * sry + dz * srz;
    const double tr = b4.x + b4.y + b4.z;
    const double cr = f * (a4.x * inv3 + 7.5 * rsr * inv7 - 1.5 * tr * inv5), cq = f * 3.0 * inv5;
    ax += cr * dx - cq * srx; ay += cr * dy - cq * sry; az += cr * dz - cq * srz;
}

// the same term where the softening guarantees r2 > 0 (no guard, no select)
__device__ __forceinline__ void grav_term_soft(double qx, double qy, double qz, double qm, double xi, double yi,
                                               double zi, double e2, double& ax, double& ay, double& az) {
    const double dx = qx - xi, dy = qy - yi, dz = qz - zi;
    const double r2 = dx * dx + dy * dy + dz * dz + e2;
    const double y0 = __builtin_amdgcn_rsq(r2);
    const double inv = y0 * __builtin_fma(-0.5 * r2 * y0, y0, 1.5);
    const double w = qm * (inv * inv * inv);
    ax += w * dx; ay += w * dy; az += w * dz;
}
__device__ __forceinline__ void grav_term(double qx, double qy, double qz, double qm, double xi, double yi, double zi,
                                          double e2, double& ax, double& ay, double& az) {
    const double dx = qx - xi, dy = qy - yi, dz = qz - zi;
    const double r2 = dx * dx + dy * dy + dz * dz + e2;
    // 1/sqrt: the hardware seed (v_rsq_f64, ~2^-26) and one Newton step (~2^-51) instead of the library
    // routine's two steps, scaling and class checks - r2 sits mid-range (m^2), and a monopole sum that is
    // good to 1e-3 has no use for the last bit.  (The exact direct sum above keeps rsqrt().)
    double inv = 0.0;
    if (r2 > 0.0) {
        const double y0 = __builtin_amdgcn_rsq(r2);
        inv = y0 * __builtin_fma(-0.5 * r2 * y0, y0, 1.5);
    }
    const double w = qm * (inv * inv * inv);
    ax += w * dx; ay += w * dy; az += w * dz;
}

// per-lane walks (each thread its own loops and loads): the per-thread kernel, and the wave kernel's
// fall-back where a wave's particles are not neighbours (blob order has a few long jumps)
__device__ __forceinline__ void near_walk_lane(const GridParams& g, int ws, int fx, int fy, int fz,
                                               const int* __restrict__ cell_start, const double* __restrict__ x,
                                               const double* __restrict__ y, const double* __restrict__ z,
                                               const double* __restrict__ m, double xi, double yi, double zi, double e2,
                                               double& ax, double& ay, double& az) {
    const int X = fx >> 1, Y = fy >> 1, Z = fz >> 1;
    const int x0 = max(2 * (X - ws), 0), x1 = min(2 * (X + ws) + 2, g.nx);
    const int y0 = max(2 * (Y - ws), 0), y1 = min(2 * (Y + ws) + 2, g.ny);
    const int z0 = max(2 * (Z - ws), 0), z1 = min(2 * (Z + ws) + 2, g.nz);
    for (int kz = z0; kz < z1; ++kz)
        for (int ky = y0; ky < y1; ++ky) {
            const int row = (kz * g.ny + ky) * g.nx;
            const int s = cell_start[row + x0], e = cell_start[row + x1];
            for (int j = s; j < e; ++j) grav_term(x[j], y[j], z[j], m[j], xi, yi, zi, e2, ax, ay, az);
        }
}
__device__ __forceinline__ void far_walk_lane(int ws, int X, int Y, int Z, int nxl, int nyl, int nzl,
                                              const double4* __restrict__ lev, const double4* __restrict__ qlev,
                                              double xi, double yi, double zi,
                                              double e2, double& ax, double& ay, double& az) {
    const int PX = X >> 1, PY = Y >> 1, PZ = Z >> 1;
    const int x0 = max(2 * (PX - ws), 0), x1 = min(2 * (PX + ws) + 1, nxl - 1);
    const int y0 = max(2 * (PY - ws), 0), y1 = min(2 * (PY + ws) + 1, nyl - 1);
    const int z0 = max(2 * (PZ - ws), 0), z1 = min(2 * (PZ + ws) + 1, nzl - 1);
    for (int kz = z0; kz <= z1; ++kz) {
        const bool nz_ = abs(kz - Z) <= ws;
        for (int ky = y0; ky <= y1; ++ky) {
            const bool nyz = nz_ && abs(ky - Y) <= ws;
            const double4* rowp = lev + ((size_t)kz * nyl + ky) * nxl;
            for (int kx = x0; kx <= x1; ++kx) {
                if (nyz && abs(kx - X) <= ws) continue;          // a neighbour: resolved at a finer level
                const double4 q = rowp[kx];
                if (!(q.x > 0.0)) continue;
                if (qlev) {
                    const size_t cc = ((size_t)kz * nyl + ky) * nxl + kx;
                    grav_term_quad(q, qlev[2 * cc], qlev[2 * cc + 1], 1.0, xi, yi, zi, e2, ax, ay, az);
                } else {
                    grav_term(q.y, q.z, q.w, q.x, xi, yi, zi, e2, ax, ay, az);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void gravity_tree_kernel(int n, GridParams g, Pyr py, int ws,
                                                           const int* __restrict__ cell_of_sorted,
                                                           const int* __restrict__ cell_start,
                                                           const double* __restrict__ x, const double* __restrict__ y,
                                                           const double* __restrict__ z, const double* __restrict__ m,
                                                           const double4* __restrict__ pyr,
                                                           const double4* __restrict__ quad, const double* eps_ptr,
                                                           double eps_val, double G, const int* __restrict__ qorder,
                                                           const int* __restrict__ omap, double* acc) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;
    const double eps = eps_ptr ? *eps_ptr : eps_val;
    const double e2 = eps * eps;
    const double xi = x[i], yi = y[i], zi = z[i];
    const int c0 = cell_of_sorted[i];
    const int fx = c0 % g.nx, fy = (c0 / g.nx) % g.ny, fz = c0 / (g.nx * g.ny);
    double ax = 0.0, ay = 0.0, az = 0.0;
    near_walk_lane(g, ws, fx, fy, fz, cell_start, x, y, z, m, xi, yi, zi, e2, ax, ay, az);
    for (int l = 1; l < py.nlev; ++l)
        far_walk_lane(ws, fx >> l, fy >> l, fz >> l, py.nx[l], py.ny[l], py.nz[l], pyr + py.off[l],
                      quad ? quad + 2 * py.off[l] : nullptr, xi, yi, zi, e2, ax, ay, az);
    const int o = omap ? omap[i] : i;
    acc[3 * (size_t)o] = G * ax; acc[3 * (size_t)o + 1] = G * ay; acc[3 * (size_t)o + 2] = G * az;
}

// ---- the same sum, one WAVE per 64 particles working through LDS -----------------------------------
// The 64 particles of a wave are neighbours (blob order), so their near regions and interaction lists
// overlap almost entirely.  At every level the wave takes the UNION box of its lanes' lists, stages it
// plane by plane into LDS with coalesced loads (each cell record / particle once per wave, not once per
// lane), and every lane runs over the staged records with broadcast LDS reads, a per-lane bit mask
// saying which of them belong to ITS list.  A lane adds its own terms in the same order as the
// per-thread kernel above (masked ones add +0): the result is bit-identical to it.
#define GT_CB 192                       // cell records / particles staged per wave
#ifndef GT_SLACK
#define GT_SLACK 1                      // cells a wave's union box may exceed one lane's box by, per axis
#endif
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
// bits a..b (inclusive) of a 32-bit word, the range clipped to 0..31
__device__ __forceinline__ unsigned range_bits(int a, int b) {
    a = max(a, 0); b = min(b, 31);
    if (b < a) return 0u;
    const unsigned upto_b = (b == 31) ? 0xFFFFFFFFu : ((2u << b) - 1u);
    return upto_b & ~((1u << a) - 1u);
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// QUAD: cells carry quadrupoles too (three 32-B pieces per staged cell, half as many cells per stage)
template <bool QUAD>
__global__ __launch_bounds__(256) void gravity_tree_wave_kernel(int n, GridParams g, Pyr py, int ws,
                                                                const int* __restrict__ cell_of_sorted,
                                                                const int* __restrict__ cell_start,
                                                                const double* __restrict__ x, const double* __restrict__ y,
                                                                const double* __restrict__ z, const double* __restrict__ m,
                                                                const double4* __restrict__ pyr,
                                                                const double4* __restrict__ quad, const double* eps_ptr,
                                                                double eps_val, double G, const int* __restrict__ qorder,
                                                                const int* __restrict__ omap, double* acc) {
    constexpr int CB = QUAD ? GT_CB / 2 : GT_CB;      // cells staged at a time
    __shared__ double4 cbuf_all[4][QUAD ? 3 * (GT_CB / 2) : GT_CB];
    __shared__ int ibuf_all[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double4* cbuf = cbuf_all[wave];
    int* ibuf = ibuf_all[wave];
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const bool act = p < n;
    if (__builtin_amdgcn_ballot_w64(act) == 0ull) return;          // (a wave past the end)
    const int i = act ? (qorder ? qorder[p] : p) : 0;
    const double eps = eps_ptr ? *eps_ptr : eps_val;
    const double e2 = eps * eps;
    const bool soft = e2 > 1e-290;                                  // (wave-uniform)
    const double xi = x[i], yi = y[i], zi = z[i];
    int c0 = cell_of_sorted[i];
    c0 = act ? c0 : __builtin_amdgcn_readfirstlane(c0);             // idle lanes (last wave) shadow lane 0
    const int fx = c0 % g.nx, fy = (c0 / g.nx) % g.ny, fz = c0 / (g.nx * g.ny);
    double ax = 0.0, ay = 0.0, az = 0.0;
    // ---- near field: particles of the level-1 cells within +-ws of the lane's level-1 cell ---------
    {
        const int X = fx >> 1, Y = fy >> 1, Z = fz >> 1;
        const int lx0 = max(2 * (X - ws), 0), lx1 = min(2 * (X + ws) + 2, g.nx) - 1;      // fine cells, inclusive
        const int ly0 = max(2 * (Y - ws), 0), ly1 = min(2 * (Y + ws) + 2, g.ny) - 1;
        const int lz0 = max(2 * (Z - ws), 0), lz1 = min(2 * (Z + ws) + 2, g.nz) - 1;
        const int ux0 = wave_min_i(lx0), ux1 = wave_max_i(lx1);
        const int uy0 = wave_min_i(ly0), uy1 = wave_max_i(ly1);
        const int uz0 = wave_min_i(lz0), uz1 = wave_max_i(lz1);
        // the wave's particles are neighbours: its union box is little more than one lane's box
        const int own = 4 * ws + 2 + GT_SLACK;
        const bool compact = ux1 - ux0 < own && uy1 - uy0 < own && uz1 - uz0 < own;
        if (!compact) near_walk_lane(g, ws, fx, fy, fz, cell_start, x, y, z, m, xi, yi, zi, e2, ax, ay, az);
        for (int kz = uz0; compact && kz <= uz1; ++kz) {
            const bool zin = kz >= lz0 && kz <= lz1;
            for (int ky = uy0; ky <= uy1; ++ky) {
                const bool rowin = zin && ky >= ly0 && ky <= ly1;
                const int row = (kz * g.ny + ky) * g.nx;
                const int s = __builtin_amdgcn_readfirstlane(cell_start[row + ux0]);
                const int e = __builtin_amdgcn_readfirstlane(cell_start[row + ux1 + 1]);
                for (int j0 = s; j0 < e; j0 += 64) {
                    const int cnt = min(64, e - j0);
                    if (lane < cnt) {
                        const int j = j0 + lane;
                        cbuf[lane] = make_double4(x[j], y[j], z[j], m[j]);
                        ibuf[lane] = cell_of_sorted[j] - row;                   // the particle's fine x cell
                    }
                    lds_sync();
                    for (int t = 0; t < cnt; ++t) {
                        const double4 q = cbuf[t];
                        const int cx = ibuf[t];
                        const bool mine = rowin && cx >= lx0 && cx <= lx1;
                        grav_term(q.x, q.y, q.z, mine ? q.w : 0.0, xi, yi, zi, e2, ax, ay, az);
                    }
                    lds_sync();
                }
            }
        }
    }
    // ---- far field: interaction lists, level by level ------------------------------------------------
    for (int l = 1; l < py.nlev; ++l) {
        const int X = fx >> l, Y = fy >> l, Z = fz >> l;
        const int PX = X >> 1, PY = Y >> 1, PZ = Z >> 1;
        const int nxl = py.nx[l], nyl = py.ny[l], nzl = py.nz[l];
        const int lx0 = max(2 * (PX - ws), 0), lx1 = min(2 * (PX + ws) + 1, nxl - 1);
        const int ly0 = max(2 * (PY - ws), 0), ly1 = min(2 * (PY + ws) + 1, nyl - 1);
        const int lz0 = max(2 * (PZ - ws), 0), lz1 = min(2 * (PZ + ws) + 1, nzl - 1);
        const double4* lev = pyr + py.off[l];
        const int ux0 = wave_min_i(lx0), ux1 = wave_max_i(lx1);
        const int uy0 = wave_min_i(ly0), uy1 = wave_max_i(ly1);
        const int uz0 = wave_min_i(lz0), uz1 = wave_max_i(lz1);
        const int W = ux1 - ux0 + 1;
        const int own = 4 * ws + 2 + GT_SLACK;
        if (W > 32 || W > own || uy1 - uy0 >= own || uz1 - uz0 >= own) {
            // a wave spread over more cells than neighbours would be: the per-lane walk
            far_walk_lane(ws, X, Y, Z, nxl, nyl, nzl, lev, QUAD ? quad + 2 * py.off[l] : nullptr, xi, yi, zi, e2,
                          ax, ay, az);
            continue;
        }
        const double4* qlev = QUAD ? quad + 2 * py.off[l] : nullptr;
        const unsigned xlist = range_bits(lx0 - ux0, lx1 - ux0);          // the lane's x range within the union row
        const unsigned xnear = range_bits(X - ws - ux0, X + ws - ux0);    // ... and its own neighbourhood in it
        const int rpc = CB / W;                                           // rows staged at a time
        for (int kz = uz0; kz <= uz1; ++kz) {
            const bool zin = kz >= lz0 && kz <= lz1;
            const bool znear = abs(kz - Z) <= ws;
            for (int kyc = uy0; kyc <= uy1; kyc += rpc) {
                const int nr = min(rpc, uy1 - kyc + 1);
                for (int r = 0; r < nr; ++r)
                    if (lane < W) {
                        const size_t cc = ((size_t)kz * nyl + (kyc + r)) * nxl + ux0 + lane;
                        cbuf[r * W + lane] = lev[cc];
                        if (QUAD) { cbuf[CB + r * W + lane] = qlev[2 * cc]; cbuf[2 * CB + r * W + lane] = qlev[2 * cc + 1]; }
                    }
                lds_sync();
                for (int r = 0; r < nr; ++r) {
                    const int ky = kyc + r;
                    const bool yin = zin && ky >= ly0 && ky <= ly1;
                    const bool near_row = znear && abs(ky - Y) <= ws;
                    const unsigned bits = yin ? (near_row ? (xlist & ~xnear) : xlist) : 0u;
                    const double4* rowq = cbuf + r * W;
                    if (QUAD) {
                        for (int t = 0; t < W; ++t) {
                            const double4 q = rowq[t];
                            if (!(q.x > 0.0)) continue;                           // (uniform: an empty cell)
                            grav_term_quad(q, rowq[CB + t], rowq[2 * CB + t], ((bits >> t) & 1u) ? 1.0 : 0.0, xi, yi, zi,
                                           e2, ax, ay, az);
                        }
                    } else if (soft) {
                        // r2 >= eps^2 > 0: no guard; two records in flight per trip
                        int t = 0;
                        for (; t + 1 < W; t += 2) {
                            const double4 q0 = rowq[t], q1 = rowq[t + 1];
                            grav_term_soft(q0.y, q0.z, q0.w, ((bits >> t) & 1u) ? q0.x : 0.0, xi, yi, zi, e2, ax, ay, az);
                            grav_term_soft(q1.y, q1.z, q1.w, ((bits >> (t + 1)) & 1u) ? q1.x : 0.0, xi, yi, zi, e2, ax, ay, az);
                        }
                        if (t < W) {
                            const double4 q0 = rowq[t];
                            grav_term_soft(q0.y, q0.z, q0.w, ((bits >> t) & 1u) ? q0.x : 0.0, xi, yi, zi, e2, ax, ay, az);
                        }
                    } else {
                        for (int t = 0; t < W; ++t) {
                            const double4 q = rowq[t];
                            if (!(q.x > 0.0)) continue;                           // (uniform: an empty cell)
                            grav_term(q.y, q.z, q.w, ((bits >> t) & 1u) ? q.x : 0.0, xi, yi, zi, e2, ax, ay, az);
                        }
                    }
                }
                lds_sync();
            }
        }
    }
    if (act) {
        const int o = omap ? omap[i] : i;
        acc[3 * (size_t)o] = G * ax; acc[3 * (size_t)o + 1] = G * ay; acc[3 * (size_t)o + 2] = G * az;
    }
}

__global__ __launch_bounds__(256) void gather1_kernel(int n, const int* perm, const double* in, double* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = in[perm[t]];
}

__global__ __launch_bounds__(256) void sorted_cells_kernel(int n, const int* cell_of, const int* perm, int* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = cell_of[perm[t]];
}

// Gravity of the particles x,y,z,m held in the cell-sorted order of the CURRENT grid (ctx->grid,
// cell_start, cell_of, perm as sphx_build_grid left them).  acc (n,3) is written at omap[i] (or i).
int sphx_gravity_tree_launch(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z,
                             const double* m, int ws, const double* eps_dev, double eps, double G, const int* omap,
                             double* acc) {
    const GridParams g = ctx->grid;
    Pyr py;
    memset(&py, 0, sizeof(py));
    py.nx[0] = g.nx; py.ny[0] = g.ny; py.nz[0] = g.nz;
    long long tot = 0;
    int l = 0;
    do {
        ++l;
        py.nx[l] = (py.nx[l - 1] + 1) / 2; py.ny[l] = (py.ny[l - 1] + 1) / 2; py.nz[l] = (py.nz[l - 1] + 1) / 2;
        py.off[l] = tot;
        tot += (long long)py.nx[l] * py.ny[l] * py.nz[l];
    } while ((py.nx[l] > 1 || py.ny[l] > 1 || py.nz[l] > 1) && l < PYR_MAX);
    py.nlev = l;
    SPHX_TRY(sphx_ensure(ctx, ctx->grav_pyr, (size_t)tot * sizeof(double4)));
    SPHX_TRY(sphx_ensure(ctx, ctx->grav_cell, (size_t)n * sizeof(int)));
    double4* pyr = ctx->grav_pyr.as<double4>();
    double4* quad = nullptr;                            // order 2: two more 32-B pieces per cell
    if (ctx->grav_order >= 2) {
        SPHX_TRY(sphx_ensure(ctx, ctx->grav_quad, (size_t)tot * 2 * sizeof(double4)));
        quad = ctx->grav_quad.as<double4>();
    }
    const int n1 = py.nx[1] * py.ny[1] * py.nz[1];
    hipLaunchKernelGGL(pyr_level1_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, g, py,
                       ctx->cell_start.as<int>(), x, y, z, m, pyr, quad);
    for (int q = 2; q <= py.nlev; ++q) {
        const int nq = py.nx[q] * py.ny[q] * py.nz[q];
        hipLaunchKernelGGL(pyr_up_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream, py, q, pyr, quad);
    }
    hipLaunchKernelGGL(sorted_cells_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                       ctx->cell_of.as<int>(), ctx->perm.as<int>(), ctx->grav_cell.as<int>());
    if (ctx->grav_per_thread)         // SPHX_GRAV_KERNEL=0: the per-thread walk (the wave kernel's reference)
        hipLaunchKernelGGL(gravity_tree_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, g, py,
                           ws, ctx->grav_cell.as<int>(), ctx->cell_start.as<int>(), x, y, z, m, pyr, quad, eps_dev, eps, G,
                           ctx->qorder, omap, acc);
    else if (quad)
        hipLaunchKernelGGL(gravity_tree_wave_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (int)n, g, py, ws, ctx->grav_cell.as<int>(), ctx->cell_start.as<int>(), x, y, z, m, pyr, quad,
                           eps_dev, eps, G, ctx->qorder, omap, acc);
    else
        hipLaunchKernelGGL(gravity_tree_wave_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (int)n, g, py, ws, ctx->grav_cell.as<int>(), ctx->cell_start.as<int>(), x, y, z, m, pyr, quad,
                           eps_dev, eps, G, ctx->qorder, omap, acc);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

extern "C" int sphx_gravity_tree(sphx_ctx* ctx, int64_t n, const double* mass, const double* points,
                                 const double* sizes, double softening, double G, int ws, int k_cells,
                                 double* accel) {
    if (!ctx) return SPHX_E_ARG;
    if (!mass || !points || !accel) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_gravity_tree: NULL argument");
    if (n < 1 || n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    if (ws < 1 || ws > 4) return sphx_set_err(ctx, SPHX_E_ARG, "ws=%d not in 1..4", ws);
    if (!sizes && !(softening >= 0.0)) return sphx_set_err(ctx, SPHX_E_ARG, "softening must be >= 0");
    if (k_cells < 1) k_cells = 40;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->map_perm = nullptr;
    ctx->qorder = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    DevBuf* bufs[] = {&ctx->in_b, &ctx->in_c, &ctx->in_d, &ctx->in_e, &ctx->in_f, &ctx->in_g, &ctx->in_h, &ctx->in_i};
    for (DevBuf* b : bufs) SPHX_TRY(sphx_ensure(ctx, *b, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_a, 3 * nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->out_b, 3 * nb));
    HIPCHK(hipMemcpyAsync(ctx->in_a.p, points, 3 * nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->in_h.p, mass, nb, hipMemcpyHostToDevice, ctx->stream));
    double *x = ctx->in_b.as<double>(), *y = ctx->in_c.as<double>(), *z = ctx->in_d.as<double>();
    double *xs = ctx->in_e.as<double>(), *ys = ctx->in_f.as<double>(), *zs = ctx->in_g.as<double>();
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, ctx->in_a.as<double>(), x, y, z));
    ctx->clip_valid = false;
    SPHX_TRY(sphx_build_grid(ctx, n, k_cells, x, y, z, 0.0));
    SPHX_TRY(sphx_gather3(ctx, n, ctx->perm.as<int>(), x, y, z, xs, ys, zs));
    hipLaunchKernelGGL(gather1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                       ctx->perm.as<int>(), ctx->in_h.as<double>(), ctx->in_i.as<double>());
    const double* eps_dev = nullptr;
    if (sizes) {
        HIPCHK(hipMemcpyAsync(ctx->in_a.p, sizes, nb, hipMemcpyHostToDevice, ctx->stream));
        double* slot = ctx->scal.as<double>() + SC_GRAV_EPS;
        SPHX_TRY(sphx_median(ctx, n, ctx->in_a.as<double>(), slot));
        eps_dev = slot;
    }
    SPHX_TRY(sphx_gravity_tree_launch(ctx, n, xs, ys, zs, ctx->in_i.as<double>(), ws, eps_dev, softening, G,
                                      ctx->perm.as<int>(), ctx->out_b.as<double>()));
    HIPCHK(hipMemcpyAsync(accel, ctx->out_b.p, 3 * nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}
