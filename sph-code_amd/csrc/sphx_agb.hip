// sphx_agb.hip - AGB dust-yield lookup, sph/config_helper.py:180-211 (calculate_interpolation).
//
// The reference fits 11 yields on the (metallicity, mass) table with SciPy's RectBivariateSpline
// (kx = ky = 1) and evaluates every spline once per star in a Python loop.  Here one thread takes one
// star: tensor-product degree-1 splines are evaluated the way FITPACK's fpbisp/fpbspl does (arguments
// clamped to the knot range, two de Boor weights per direction, products accumulated in FITPACK's
// order), yields are written through `mapto` (a repeated target keeps the LAST spline, as NumPy's fancy
// assignment does), divided, clipped at zero, and the gas return (config_helper.py:193-209) follows.
// Row sums follow NumPy's pairwise summation so results agree with the reference to the last bits.
#include "sphx_internal.h"
#pragma clang fp contract(off)

#include "sphx_agb.h"

struct AgbArgs {
    int n;
    AgbTable t;
    const double *mass, *met, *comp;
    double *dust, *gas;
};

// numpy's pairwise_sum for n < 128 (8 strided partial sums, then the remainder in order)
__device__ __forceinline__ double np_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

__global__ __launch_bounds__(128) void agb_kernel(AgbArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double M = a.mass[i], Z = a.met[i];
    const int nspec = a.t.nspec;
    double dust[AGB_MAX_SPEC];
    agb_dust_yields(a.t, M, Z, dust);
    for (int s = 0; s < nspec; ++s) a.dust[(size_t)i * nspec + s] = dust[s];
    if (!a.gas) return;
    const double mass_wd = (0.55 + (M / a.t.solar - 1.) * 0.45 / (7. - 1.)) * a.t.solar;     // :182
    const double gas_mass = M - np_sum(dust, nspec) - mass_wd;                          // :191
    double num[AGB_MAX_SPEC], w[AGB_MAX_SPEC];
    for (int s = 0; s < nspec; ++s) num[s] = a.comp[(size_t)i * nspec + s] / a.t.mu[s]; // :194
    const double ion = num[3] * 0.1, h2 = num[0] * 0.1, h = num[2] * 0.1;                 // :196-198
    num[0] -= h2;
    num[1] += h / 4. + h2 / 2. + ion / 4.;
    num[2] -= h;
    num[3] -= ion;
    num[5] -= ion;
    for (int s = 0; s < nspec; ++s) w[s] = num[s] * a.t.mu[s];
    const double tot = np_sum(w, nspec);                                               // :206
    for (int s = 0; s < nspec; ++s) a.gas[(size_t)i * nspec + s] = w[s] / tot * gas_mass;   // :206,211
}

extern "C" int sphx_agb_yields(sphx_ctx* ctx, int64_t n, const double* masses, const double* metallicities,
                               int nspl, const int32_t* ntx, const int32_t* nty, const double* tx,
                               const double* ty, const double* coeffs, const int32_t* mapto, double divisor,
                               int nspecies, const double* mu_specie, const double* composition,
                               double solar_mass, double* dust_out, double* gas_out) {
    if (!ctx) return SPHX_E_ARG;
    if (!masses || !metallicities || !ntx || !nty || !tx || !ty || !coeffs || !mapto || !mu_specie || !dust_out)
        return sphx_set_err(ctx, SPHX_E_ARG, "sphx_agb_yields: NULL argument");
    if (gas_out && !composition) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_agb_yields: gas_out needs composition");
    if (n < 1 || n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    if (nspl < 1 || nspl > AGB_MAX_SPL || nspecies < 6 || nspecies > AGB_MAX_SPEC)
        return sphx_set_err(ctx, SPHX_E_ARG, "sphx_agb_yields: %d splines / %d species not supported", nspl, nspecies);
    if (!(divisor != 0.0)) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_agb_yields: divisor is 0");
    HIPCHK(hipSetDevice(ctx->device));
    AgbArgs a;
    a.n = (int)n; a.t.nspl = nspl; a.t.nspec = nspecies;
    size_t ntx_tot = 0, nty_tot = 0, nc_tot = 0;
    for (int o = 0; o < nspl; ++o) {
        if (ntx[o] < 4 || nty[o] < 4)
            return sphx_set_err(ctx, SPHX_E_ARG, "sphx_agb_yields: spline %d has fewer than 4 knots", o);
        if (mapto[o] < 0 || mapto[o] >= nspecies)
            return sphx_set_err(ctx, SPHX_E_ARG, "sphx_agb_yields: mapto[%d]=%d out of range", o, mapto[o]);
        ntx_tot += ntx[o]; nty_tot += nty[o]; nc_tot += (size_t)(ntx[o] - 2) * (nty[o] - 2);
    }
    size_t ox = 0, oy = ntx_tot, oc = ntx_tot + nty_tot;
    for (int o = 0; o < nspl; ++o) {
        a.t.tx_off[o] = (int)ox; a.t.ty_off[o] = (int)oy; a.t.c_off[o] = (int)oc;
        a.t.ntx[o] = ntx[o]; a.t.nty[o] = nty[o]; a.t.mapto[o] = mapto[o];
        ox += ntx[o]; oy += nty[o]; oc += (size_t)(ntx[o] - 2) * (nty[o] - 2);
    }
    for (int s = 0; s < nspecies; ++s) a.t.mu[s] = mu_specie[s];
    a.t.divisor = divisor; a.t.solar = solar_mass;
    const size_t nk = ntx_tot + nty_tot + nc_tot;
    const size_t ns = (size_t)n * nspecies * sizeof(double);
    SPHX_TRY(sphx_ensure(ctx, ctx->in_a, nk * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_b, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_c, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->out_b, ns));
    double* kd = ctx->in_a.as<double>();
    HIPCHK(hipMemcpyAsync(kd, tx, ntx_tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(kd + ntx_tot, ty, nty_tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(kd + ntx_tot + nty_tot, coeffs, nc_tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->in_b.p, masses, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->in_c.p, metallicities, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    a.t.knots = kd; a.mass = ctx->in_b.as<double>(); a.met = ctx->in_c.as<double>();
    a.comp = nullptr; a.gas = nullptr;
    a.dust = ctx->out_b.as<double>();
    if (gas_out) {
        SPHX_TRY(sphx_ensure(ctx, ctx->in_d, ns));
        SPHX_TRY(sphx_ensure(ctx, ctx->out_c, ns));
        HIPCHK(hipMemcpyAsync(ctx->in_d.p, composition, ns, hipMemcpyHostToDevice, ctx->stream));
        a.comp = ctx->in_d.as<double>();
        a.gas = ctx->out_c.as<double>();
    }
    hipLaunchKernelGGL(agb_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dust_out, a.dust, ns, hipMemcpyDeviceToHost, ctx->stream));
    if (gas_out) HIPCHK(hipMemcpyAsync(gas_out, a.gas, ns, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}
