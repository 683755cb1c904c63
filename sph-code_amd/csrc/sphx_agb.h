// sphx_agb.h - the AGB dust-yield table (sph/config_helper.py:138-189) as the device sees it, and its evaluation:
// shared by the stand-alone lookup (sphx_agb.hip, sphx_agb_yields) and by the species pass of the step loop
// (sphx_sums.hip), where the lookup is fused behind the per-particle metallicity (BASELINE configs[4]).
// Include only from translation units compiled with `#pragma clang fp contract(off)`.
#pragma once

#define AGB_MAX_SPL 32
#define AGB_MAX_SPEC 32

struct AgbTable {
    int nspl, nspec;
    const double* knots;                         // tx | ty | coeffs, flattened (device)
    int tx_off[AGB_MAX_SPL], ty_off[AGB_MAX_SPL], c_off[AGB_MAX_SPL], ntx[AGB_MAX_SPL], nty[AGB_MAX_SPL];
    int mapto[AGB_MAX_SPL];
    double mu[AGB_MAX_SPEC];
    double divisor, solar;
    // the same per-spline integers in device memory, for lanes that evaluate DIFFERENT splines at once (sphx_blob.hip):
    // knots[meta_off + 6 o ..] = {tx_off, ty_off, c_off, ntx, nty, target}, target = mapto[o], or -1 when a later spline
    // writes the same species (config_helper.py:185: the last one stays); covered: bit s set when some spline writes species s
    int meta_off;
    unsigned covered;
};
// interval l (0-based index of the left knot) and the two weights of a degree-1 spline, FITPACK's way
// (arguments clamped to the knot range)
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

// one spline at (Z, M) from the device-resident description: (target species or -1, value already divided and clipped)
__device__ __forceinline__ int agb_one_spline(const AgbTable& a, int o, double M, double Z, double& out) {
    const double* md = a.knots + a.meta_off + 6 * o;
    const int target = (int)md[5];
    if (target < 0) return -1;
    const double* tx = a.knots + (int)md[0];
    const double* ty = a.knots + (int)md[1];
    const double* c = a.knots + (int)md[2];
    const int ntx = (int)md[3], nty = (int)md[4];
    const int ny = nty - 2;
    int lx, ly;
    double wx0, wx1, wy0, wy1;
    agb_weights(tx, ntx, Z, lx, wx0, wx1);
    agb_weights(ty, nty, M, ly, wy0, wy1);
    const double* c0 = c + (lx - 1) * ny + (ly - 1);
    double sp = c0[0] * wx0 * wy0;
    sp = sp + c0[1] * wx0 * wy1;
    sp = sp + c0[ny] * wx1 * wy0;
    sp = sp + c0[ny + 1] * wx1 * wy1;
    const double d = sp / a.divisor;                   // config_helper.py:188
    out = (d < 0.0) ? 0.0 : d;                         // :189
    return target;
}


// config_helper.py:183-189: every spline at (Z, M), written through mapto (a repeated target keeps the LAST
// spline, as NumPy's fancy assignment does), divided, clipped at zero.  dust[0 .. nspec)
__device__ __forceinline__ void agb_dust_yields(const AgbTable& a, double M, double Z, double* dust) {
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
        dust[s] = (d < 0.0) ? 0.0 : d;                 // :189
    }
}
