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

#define AGB_MAX_SPL 32
#define AGB_MAX_SPEC 32

struct AgbArgs {
    int n, nspl, nspec;
    const double *mass, *met, *comp, *knots;     // knots: tx | ty | coeffs, flattened
    int tx_off[AGB_MAX_SPL], ty_off[AGB_MAX_SPL], c_off[AGB_MAX_SPL], ntx[AGB_MAX_SPL], nty[AGB_MAX_SPL];
    int mapto[AGB_MAX_SPL];
    double mu[AGB_MAX_SPEC];
    double divisor, solar;
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

// interval l (0-based index of the left knot) and the two weights of a degree-1 spline
__device__ __forceinline__ void agb_weights(const double* t, int nt, double x, int& l, double& w0, double& w1) {
    const double tb = t[1], te = t[nt - 2];
    x = x < tb ? tb : x;
    x = x > te ? te : x;
    l = 1;
    while (l < nt - 3 && x >= t[l + 1]) ++l;
    const double f = 1.0 / (t[l + 1] - t[l]);
    w0 = f * (t[l + 1] - x);
    w1 = f * (x - t[l]);
}

__global__ __launch_bounds__(128) void agb_kernel(AgbArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double M = a.mass[i], Z = a.met[i];
    double dust[AGB_MAX_SPEC];
    for (int s = 0; s < a.nspec; ++s) dust[s] = 0.0;
    for (int o = 0; o < a.nspl; ++o) {
        const double* tx = a.knots + a.tx_off[o];
        const double* ty = a.knots + a.ty_off[o];
        const double* c = a.knots + a.c_off[o];
        const int ny = a.nty[o] - 2;
        int lx, ly;
        double wx0, wx1, wy0, wy1;
        agb_weights(tx, a.ntx[o], Z, lx, wx0, wx1);
        agb_weights(ty, a.nty[o], M, ly, wy0, wy1);
        const double* c0 = c + (lx - 1) * ny + (ly - 1);
        double sp = c0[0] * wx0 * wy0;
        sp = sp + c0[1] * wx0 * wy1;
        sp = sp + c0[ny] * wx1 * wy0;
        sp = sp + c0[ny + 1] * wx1 * wy1;
        dust[a.mapto[o]] = sp;                         // config_helper.py:185
    }
    for (int s = 0; s < a.nspec; ++s) {
        double d = dust[s] / a.divisor;                // :188
        d = (d < 0.0) ? 0.0 : d;                       // :189
        dust[s] = d;
        a.dust[(size_t)i * a.nspec + s] = d;
    }
    if (!a.gas) return;
    const double mass_wd = (0.55 + (M / a.solar - 1.) * 0.45 / (7. - 1.)) * a.solar;     // :182
    const double gas_mass = M - np_sum(dust, a.nspec) - mass_wd;                          // :191
    double num[AGB_MAX_SPEC], w[AGB_MAX_SPEC];
    for (int s = 0; s < a.nspec; ++s) num[s] = a.comp[(size_t)i * a.nspec + s] / a.mu[s]; // :194
    const double ion = num[3] * 0.1, h2 = num[0] * 0.1, h = num[2] * 0.1;                 // :196-198
    num[0] -= h2;
    num[1] += h / 4. + h2 / 2. + ion / 4.;
    num[2] -= h;
    num[3] -= ion;
    num[5] -= ion;
    for (int s = 0; s < a.nspec; ++s) w[s] = num[s] * a.mu[s];
    const double tot = np_sum(w, a.nspec);                                               // :206
    for (int s = 0; s < a.nspec; ++s) a.gas[(size_t)i * a.nspec + s] = w[s] / tot * gas_mass;   // :206,211
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
    a.n = (int)n; a.nspl = nspl; a.nspec = nspecies;
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
        a.tx_off[o] = (int)ox; a.ty_off[o] = (int)oy; a.c_off[o] = (int)oc;
        a.ntx[o] = ntx[o]; a.nty[o] = nty[o]; a.mapto[o] = mapto[o];
        ox += ntx[o]; oy += nty[o]; oc += (size_t)(ntx[o] - 2) * (nty[o] - 2);
    }
    for (int s = 0; s < nspecies; ++s) a.mu[s] = mu_specie[s];
    a.divisor = divisor; a.solar = solar_mass;
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
    a.knots = kd; a.mass = ctx->in_b.as<double>(); a.met = ctx->in_c.as<double>();
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
