// sphx_dev.hip - device-pointer building blocks for spatially decomposed (multi-GPU) runs.
//
// One rank holds n_total = n_owned + n_ghost particles in the CALLER's order (owned first).
// The library sorts them by cell internally, searches/sums for the owned ones only, and returns
// every output in the caller's order.  Between the phases the caller exchanges the ghosts'
// h_j, rho_j and m Pi_j (8 B each) with their owners over RCCL (sph_code_amd/multigpu.py):
//   sphx_dev_search -> [halo h] -> sphx_dev_prep -> sphx_dev_density -> [halo rho]
//   -> sphx_dev_pi -> [halo m Pi] -> sphx_dev_visc -> sphx_dev_integrate
// All pointers are device pointers; calls are asynchronous on the context's stream except
// sphx_dev_search, which synchronises once (bounding-box read-back for the cell grid).
#include "sphx_internal.h"
#pragma clang fp contract(off)
#include <float.h>
#include "sphx_leapfrog.h"

extern "C" int sphx_set_stream(sphx_ctx* ctx, void* stream) {
    if (!ctx) return SPHX_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)stream;          // NULL is HIP's default (null) stream
    return SPHX_OK;
}

extern "C" int sphx_reset_stream(sphx_ctx* ctx) {
    if (!ctx) return SPHX_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return SPHX_OK;
}

__global__ __launch_bounds__(256) void gather3_aos_kernel(int n, const int* perm, const double* aos,
                                                          double* xs, double* ys, double* zs, int* inv) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int p = perm[t];
    xs[t] = aos[3 * (size_t)p]; ys[t] = aos[3 * (size_t)p + 1]; zs[t] = aos[3 * (size_t)p + 2];
    if (inv) inv[p] = t;
}

// dst[s * stride] = src[perm[s]]: caller-order values into a sorted-order compact array
__global__ __launch_bounds__(256) void inject_field_kernel(int n, const int* perm, const double* src,
                                                           double* dst, int stride) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    dst[(size_t)t * stride] = src[perm[t]];
}

#define NEED(p)                                                                              \
    do {                                                                                     \
        if (!(p)) return sphx_set_err(ctx, SPHX_E_ARG, "%s: argument %s is NULL", __func__, #p); \
    } while (0)

// fold the kNN launch time of the last sphx_dev_search into the statistics (events on the library's stream)
int sphx_dev_collect(sphx_ctx* ctx) {
    if (!ctx->dev_ev_pending) return SPHX_OK;
    ctx->dev_ev_pending = false;
    HIPCHK(hipEventSynchronize(ctx->ev[2]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
    ctx->stats.ms_search += ms;
    ctx->stats.steps += 1;
    ctx->stats.search_steps += 1;
    ctx->stats.n = ctx->map_nactive;
    return SPHX_OK;
}

extern "C" int sphx_dev_search(sphx_ctx* ctx, int64_t n_total, int64_t n_owned, int k, const double* pos,
                               const double* hint, double rscale, double dist, double* h_out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(h_out);
    if (n_total < 1 || n_total > 0x7FFFFFF0ll || n_owned < 0 || n_owned > n_total)
        return sphx_set_err(ctx, SPHX_E_ARG, "n_total=%lld n_owned=%lld out of range", (long long)n_total,
                            (long long)n_owned);
    if (k < 1 || k > SPHX_MAX_K) return sphx_set_err(ctx, SPHX_E_ARG, "N_NEIGH=%d not in 1..%d", k, SPHX_MAX_K);
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));                 // (a search whose lists nobody used: they are rebuilt below)
    SPHX_TRY(sphx_dev_collect(ctx));
    const int64_t n = n_total;
    const size_t nb = (size_t)n * sizeof(double);
    ctx->map_perm = nullptr;
    ctx->qorder = nullptr;
    DevBuf* bufs[] = {&ctx->in_b, &ctx->in_c, &ctx->in_d, &ctx->in_e, &ctx->in_f, &ctx->in_g};
    for (DevBuf* b : bufs) SPHX_TRY(sphx_ensure(ctx, *b, nb));
    double *x = ctx->in_b.as<double>(), *y = ctx->in_c.as<double>(), *z = ctx->in_d.as<double>();
    double *xs = ctx->in_e.as<double>(), *ys = ctx->in_f.as<double>(), *zs = ctx->in_g.as<double>();
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, pos, x, y, z));
    double cell_hint = 0.0;
    if (ctx->dev_hmean > 0.0) cell_hint = ctx->cell_factor * sphx_cell_feedback(ctx, n) * ctx->dev_hmean;
    // grid sized from the previous search's box statistics (sphx_grid.hip): the decomposed driver's step
    // then has ONE host wait - its end-of-step scalars - and the host queues the whole step ahead of the GPU
    ctx->lag_on = true;
    const int rc_grid = sphx_build_grid(ctx, n, k, x, y, z, cell_hint);
    ctx->lag_on = false;
    SPHX_TRY(rc_grid);
    hipLaunchKernelGGL(gather3_aos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (int)n, ctx->perm.as<int>(), pos, xs, ys, zs, (int*)nullptr);
    HIPCHK(hipGetLastError());
    SPHX_TRY(sphx_ensure(ctx, ctx->nbr, (size_t)k * sphx_pad64(n) * sizeof(int)));
    ctx->map_perm = ctx->perm.as<int>();
    ctx->qorder = nullptr;
    ctx->blob_lists = false;
    ctx->blob_split_valid = false;
    ctx->pass_part = 0;
    if (ctx->use_blob) SPHX_TRY(sphx_build_blob_order(ctx, n));     // queries and passes in blob order
    ctx->map_nactive = (int)n_owned;
    ctx->n = n;
    ctx->npad = sphx_pad64(n);
    ctx->k = k;
    KnnOut o;
    o.nbr = ctx->nbr.as<int>();
    o.h_sorted = nullptr; o.idx64 = nullptr; o.dist = nullptr; o.nontriv = nullptr;
    o.h_by_id = h_out;
    ctx->knn_hint_by_id = (hint != nullptr);
    ctx->knn_hinted = (hint != nullptr);
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    int rc = sphx_knn(ctx, n, k, xs, ys, zs, ctx->perm.as<int>(), ctx->inv.as<int>(), hint,
                      rscale > 0.0 ? rscale : ctx->rscale, dist, o);
    ctx->knn_hint_by_id = false;
    ctx->knn_hinted = false;
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    ctx->dev_ev_pending = (rc == SPHX_OK);
    if (rc == SPHX_OK && ctx->qorder && ctx->use_lds) {
        if (ctx->side_stream && ctx->ev_join && ctx->dev_fork_dedup) {
            // the slot lists (0.11 ms of LDS bookkeeping) are built on the side stream, beside whatever the caller does
            // between the search and the first pass: the record build, the h_j halo phase
            hipStream_t main_stream = ctx->stream;
            HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->ev[2], 0));
            ctx->stream = ctx->side_stream;
            rc = sphx_blob_translate(ctx, n, k);
            ctx->stream = main_stream;
            if (rc == SPHX_OK) {
                HIPCHK(hipEventRecord(ctx->ev_join, ctx->side_stream));
                ctx->dedup_pending = true;
            }
        } else {
            rc = sphx_blob_translate(ctx, n, k);
        }
    }
    return rc;
}

// perm of the last sphx_dev_search: out[s] = caller index of the s-th particle in cell order
extern "C" int sphx_dev_get_order(sphx_ctx* ctx, int64_t n_total, int32_t* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(out);
    if (!ctx->map_perm || ctx->n != n_total)
        return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_get_order: no search over %lld particles", (long long)n_total);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(out, ctx->map_perm, (size_t)n_total * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    return SPHX_OK;
}

extern "C" int sphx_dev_select_blobs(sphx_ctx* ctx, int part) {
    if (!ctx) return SPHX_E_ARG;
    if (part < 0 || part > 2) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_dev_select_blobs: part %d not in 0..2", part);
    ctx->pass_part = 0;
    if (part == 0) return 0;
    if (!(ctx->map_perm && ctx->qorder && ctx->blob_lists && ctx->blob_split_valid)) return 0;
    ctx->pass_part = part;
    return 1;
}

extern "C" int sphx_dev_blob_split_counts(sphx_ctx* ctx, int32_t counts[3]) {
    if (!ctx) return SPHX_E_ARG;
    NEED(counts);
    if (!ctx->blob_split_valid) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_blob_split_counts: no split (LDS passes off, or no search yet)");
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));
    HIPCHK(hipMemcpyAsync(counts, ctx->blob_split.as<int>() + (size_t)ctx->blob_split_nblk, 3 * sizeof(int),
                          hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

extern "C" int sphx_dev_set_mean_h(sphx_ctx* ctx, double mean_h) {
    if (!ctx) return SPHX_E_ARG;
    ctx->dev_hmean = mean_h;
    return SPHX_OK;
}

extern "C" int sphx_dev_prep(sphx_ctx* ctx, const double* pos, const double* vel, const double* mass,
                             const double* h, const double* T, const double* mu, const double* gamma,
                             const double* ptype) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(vel); NEED(mass); NEED(h); NEED(T); NEED(mu); NEED(gamma); NEED(ptype);
    if (!ctx->map_perm) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_prep before sphx_dev_search");
    HIPCHK(hipSetDevice(ctx->device));
    return sphx_prep(ctx, ctx->n, nullptr, nullptr, nullptr, pos, nullptr, nullptr, nullptr, vel, mass, h, T,
                     mu, gamma, ptype);
}

// A pass writes its results (already in the caller's order, through the OutMap) into context buffers; for
// the duration of a sphx_dev_* call the caller's own array stands in for that buffer, so nothing is copied.
struct Borrow {
    DevBuf& b;
    DevBuf saved;
    Borrow(DevBuf& buf, void* p, size_t bytes) : b(buf), saved(buf) {
        if (p) { b.p = p; b.cap = bytes; }
    }
    ~Borrow() { b = saved; }
    Borrow(const Borrow&) = delete;
    Borrow& operator=(const Borrow&) = delete;
};

extern "C" int sphx_dev_density(sphx_ctx* ctx, double* rho, double* rho_dust, double* nden,
                                double* hydro_accel) {
    if (!ctx) return SPHX_E_ARG;
    if (!ctx->map_perm) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_density before sphx_dev_search");
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));
    const size_t nb = (size_t)ctx->n * sizeof(double);
    Borrow b1(ctx->rho, rho, nb), b2(ctx->rhod, rho_dust, nb), b3(ctx->nden, nden, nb), b4(ctx->ha, hydro_accel, 3 * nb);
    SPHX_TRY(sphx_pass_density(ctx, ctx->n, ctx->k));
    return SPHX_OK;
}

extern "C" int sphx_dev_pi(sphx_ctx* ctx, const double* rho_complete, double* Pi, double* Bw, double* ct_out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(rho_complete);
    if (!ctx->map_perm) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_pi before sphx_dev_search");
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));
    const int64_t n = ctx->n;
    // (interior blobs: every density they read was left in sorted order by pass 1 itself; boundary blobs after the
    //  interior ones: the crossing-time votes already cast are kept)
    if (ctx->pass_part != 1)
        hipLaunchKernelGGL(inject_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                           ctx->map_perm, rho_complete, ctx->rho_s.as<double>(), 1);
    HIPCHK(hipGetLastError());
    if (ctx->pass_part == 2) ctx->ct_primed = true;
    {
        Borrow b1(ctx->Pi, Pi, (size_t)n * sizeof(double)), b2(ctx->Bw, Bw, (size_t)n * sizeof(double));
        SPHX_TRY(sphx_pass_pi(ctx, n, ctx->k, nullptr, nullptr));
    }
    if (ct_out)   // the positive double whose bits are the minimum (+inf = none found)
        HIPCHK(hipMemcpyAsync(ct_out, ctx->scal.as<u64>() + SC_CT_BITS, 8, hipMemcpyDeviceToDevice, ctx->stream));
    return SPHX_OK;
}

extern "C" int sphx_dev_visc(sphx_ctx* ctx, const double* Bw_complete, const double* mass,
                             double* visc_accel, double* visc_heat) {
    if (!ctx) return SPHX_E_ARG;
    NEED(Bw_complete); NEED(mass);
    if (!ctx->map_perm) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_visc before sphx_dev_search");
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));
    const int64_t n = ctx->n;
    if (ctx->pass_part != 1)          // (interior blobs: pass 2 left every m Pi they read in the sorted records)
        hipLaunchKernelGGL(inject_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                           ctx->map_perm, Bw_complete, ctx->bc_s.as<double>(), 2);
    HIPCHK(hipGetLastError());
    Borrow b1(ctx->va, visc_accel, 3 * (size_t)n * sizeof(double)), b2(ctx->vh, visc_heat, (size_t)n * sizeof(double));
    SPHX_TRY(sphx_pass_visc(ctx, n, ctx->k, mass));
    return SPHX_OK;
}

// ---- drv:460-491 on caller-order (n,3) arrays, owned particles only ---------------------------
#define nan_to_num_v sphx_nan_to_num
struct DevIntegArgs {
    int n;
    double *pos, *vel, *acc, *E, *T;
    const double *m, *mu, *gam, *ptype, *ha, *va, *vh;
    const double *G, *rho;        // loop-form mode (G != nullptr): ha unused; pressure = G / rho [gas], visc = va (drv:460,473)
    const double *drag_on, *drag_re, *drho, *drhod;   // gas-dust drag (drv:462-463,473); drag_on == nullptr: none
    double dt, m_h, kB, lim;
    // device-side verdict and dt (sphx_dev_integrate_auto): red2 = {halo too thin?, -min crossing time}
    const double* red2;
    double* dt_out;
    int first;
    double fixed_dt, dt_0, max_age;
    u64* counters;                // failure counters (BAD_*)
};
__global__ __launch_bounds__(256) void dev_integrate_kernel(DevIntegArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    double dt = a.dt;
    if (a.red2) {
        if (a.red2[0] > 0.5) {                    // the halo was too thin somewhere: the step will be redone
            if (i == 0) *a.dt_out = 0.0;
            return;
        }
        const double ct_min = -a.red2[1];
        // +inf = "no gas particle voted" (votes are at most DBL_MAX)                      nsc:783-786
        const double ctv = sphx_ct_value(ct_min > DBL_MAX, ct_min, a.dt_0);
        dt = (a.fixed_dt > 0.0) ? a.fixed_dt : sphx_dt_rule(ctv, a.first, a.dt_0, a.max_age);   // drv:223-229
        if (i == 0) *a.dt_out = dt;
    }
    const double g = (a.ptype[i] == 0.0) ? 1.0 : 0.0;
    double v[3], pa[3], vis[3];
    bool bad_acc = false;                                     // (what nan_to_num is about to hide: counted, BAD_ACCEL)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        v[c] = a.vel[3 * (size_t)i + c];
        if (a.G) {                                            // the loop forms carry the physical sign themselves
            const double praw = a.G[3 * (size_t)i + c] / a.rho[i] * g;
            pa[c] = nan_to_num_v(praw);                                      // drv:460
            vis[c] = a.va[3 * (size_t)i + c];                                // av[0], drv:473
            bad_acc = bad_acc || !sphx_finite(praw) || !sphx_finite(vis[c]);
        } else {
            const double praw = -a.ha[3 * (size_t)i + c] * g, vraw = -a.va[3 * (size_t)i + c] * g;
            pa[c] = nan_to_num_v(praw);
            vis[c] = nan_to_num_v(vraw);
            bad_acc = bad_acc || !sphx_finite(praw) || !sphx_finite(vraw);
        }
        if (a.drag_on) {                                      // drv:462-463,473
            const double draw = a.drag_on[3 * (size_t)i + c] * a.drhod[i] / a.drho[i] * g, rraw = a.drag_re[3 * (size_t)i + c];
            const double dg = nan_to_num_v(draw);
            vis[c] = dg + nan_to_num_v(rraw) + vis[c];
            bad_acc = bad_acc || !sphx_finite(draw) || !sphx_finite(rraw);
        }
    }
    double x[3], old[3], tot[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = a.pos[3 * (size_t)i + c]; old[c] = a.acc[3 * (size_t)i + c]; }
    sphx_leapfrog_update(dt, x, v, vis, pa, nullptr, old, tot);                  // drv:475-486
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.pos[3 * (size_t)i + c] = x[c];
        a.vel[3 * (size_t)i + c] = v[c];
        a.acc[3 * (size_t)i + c] = tot[c];
    }
    double E = a.E[i], T;
    const double heat = a.vh[i];
    const bool bad_en = !sphx_finite(E) || !sphx_finite(heat * dt);
    sphx_energy_update(dt, heat, a.mu[i], a.gam[i], a.m[i], a.m_h, a.kB, E, T);   // drv:490-491
    a.E[i] = E;
    a.T[i] = T;
    if (a.counters) {
        bool bad_st = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) bad_st = bad_st || !sphx_finite(x[c]) || !sphx_finite(v[c]);
        sphx_count_bad(a.counters, BAD_ACCEL, bad_acc);
        sphx_count_bad(a.counters, BAD_ENERGY, bad_en);
        sphx_count_bad(a.counters, BAD_STATE, bad_st);
    }
}
// the drag terms handed over by sphx_dev_set_drag_terms are consumed by the next update
static void take_drag_terms(sphx_ctx* ctx, DevIntegArgs& a) {
    a.drag_on = ctx->dev_drag_on; a.drag_re = ctx->dev_drag_re; a.drho = ctx->dev_drag_rho; a.drhod = ctx->dev_drag_rhod;
    ctx->dev_drag_on = ctx->dev_drag_re = ctx->dev_drag_rho = ctx->dev_drag_rhod = nullptr;
}
// drv:233-238 on (n,3) arrays
__global__ __launch_bounds__(256) void dev_clamp_kernel(int n3, double lim, double* pos, double* vel) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n3) return;
    pos[e] = sphx_clamp_pos(pos[e], lim);
    vel[e] = nan_to_num_v(vel[e]);
}

extern "C" int sphx_dev_clamp(sphx_ctx* ctx, int64_t n, double* pos, double* vel) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(vel);
    HIPCHK(hipSetDevice(ctx->device));
    if (n < 1) return SPHX_OK;
    hipLaunchKernelGGL(dev_clamp_kernel, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (int)(3 * n), ctx->cst.pos_clamp, pos, vel);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

extern "C" int sphx_dev_integrate(sphx_ctx* ctx, int64_t n_owned, double* pos, double* vel, double* accel_old,
                                  double* E_internal, double* T, const double* mass, const double* mu,
                                  const double* gamma, const double* ptype, const double* hydro_accel,
                                  const double* visc_accel, const double* visc_heat, double dt) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(vel); NEED(accel_old); NEED(E_internal); NEED(T); NEED(mass); NEED(mu); NEED(gamma);
    NEED(ptype); NEED(hydro_accel); NEED(visc_accel); NEED(visc_heat);
    HIPCHK(hipSetDevice(ctx->device));
    if (n_owned < 1) return SPHX_OK;
    DevIntegArgs a;
    a.n = (int)n_owned;
    a.pos = pos; a.vel = vel; a.acc = accel_old; a.E = E_internal; a.T = T;
    a.m = mass; a.mu = mu; a.gam = gamma; a.ptype = ptype;
    a.ha = hydro_accel; a.va = visc_accel; a.vh = visc_heat;
    a.G = nullptr; a.rho = nullptr;
    take_drag_terms(ctx, a);
    a.dt = dt; a.m_h = ctx->cst.m_h; a.kB = ctx->cst.k_B; a.lim = ctx->cst.pos_clamp;
    a.red2 = nullptr; a.dt_out = nullptr; a.first = 0; a.fixed_dt = 0.0; a.dt_0 = ctx->cst.dt_0; a.max_age = ctx->cst.max_age;
    a.counters = ctx->badc.as<u64>();
    hipLaunchKernelGGL(dev_integrate_kernel, dim3((unsigned)((n_owned + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// The same update with the step's verdict and dt taken on the device: red2 (device) = {1 if some rank's halo
// was too thin, -(global minimum crossing time)} as the driver's one reduction left it.  Nothing is
// changed when red2[0] > 0.5 (the driver redoes the step); otherwise dt follows drv:222-229 and is written
// to dt_out (device).  The host can read verdict and dt AFTER launching this - no round trip in between.
extern "C" int sphx_dev_integrate_auto(sphx_ctx* ctx, int64_t n_owned, double* pos, double* vel, double* accel_old,
                                       double* E_internal, double* T, const double* mass, const double* mu,
                                       const double* gamma, const double* ptype, const double* hydro_accel,
                                       const double* visc_accel, const double* visc_heat, const double* red2,
                                       int first, double fixed_dt, double* dt_out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(vel); NEED(accel_old); NEED(E_internal); NEED(T); NEED(mass); NEED(mu); NEED(gamma);
    NEED(ptype); NEED(hydro_accel); NEED(visc_accel); NEED(visc_heat); NEED(red2); NEED(dt_out);
    HIPCHK(hipSetDevice(ctx->device));
    if (n_owned < 1) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_dev_integrate_auto: n_owned=%lld", (long long)n_owned);
    DevIntegArgs a;
    a.n = (int)n_owned;
    a.pos = pos; a.vel = vel; a.acc = accel_old; a.E = E_internal; a.T = T;
    a.m = mass; a.mu = mu; a.gam = gamma; a.ptype = ptype;
    a.ha = hydro_accel; a.va = visc_accel; a.vh = visc_heat;
    a.G = nullptr; a.rho = nullptr;
    take_drag_terms(ctx, a);
    a.dt = 0.0; a.m_h = ctx->cst.m_h; a.kB = ctx->cst.k_B; a.lim = ctx->cst.pos_clamp;
    a.red2 = red2; a.dt_out = dt_out; a.first = first; a.fixed_dt = fixed_dt;
    a.dt_0 = ctx->cst.dt_0; a.max_age = ctx->cst.max_age;
    a.counters = ctx->badc.as<u64>();
    hipLaunchKernelGGL(dev_integrate_kernel, dim3((unsigned)((n_owned + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// drv:460-491 on the loop forms' outputs (multigpu.py, forms = "loop"): pressure_accel = delp / rho [gas],
// visc_accel = av[0]; red2 == NULL: dt as given; else verdict and dt on the device as sphx_dev_integrate_auto.
extern "C" int sphx_dev_integrate_loop(sphx_ctx* ctx, int64_t n_owned, double* pos, double* vel, double* accel_old,
                                       double* E_internal, double* T, const double* mass, const double* mu,
                                       const double* gamma, const double* ptype, const double* delp, const double* rho,
                                       const double* av_accel, const double* av_heat, const double* red2, int first,
                                       double fixed_dt, double dt, double* dt_out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(vel); NEED(accel_old); NEED(E_internal); NEED(T); NEED(mass); NEED(mu); NEED(gamma);
    NEED(ptype); NEED(delp); NEED(rho); NEED(av_accel); NEED(av_heat);
    if (red2 && !dt_out) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_dev_integrate_loop: red2 without dt_out");
    HIPCHK(hipSetDevice(ctx->device));
    if (n_owned < 1) return SPHX_OK;
    DevIntegArgs a;
    a.n = (int)n_owned;
    a.pos = pos; a.vel = vel; a.acc = accel_old; a.E = E_internal; a.T = T;
    a.m = mass; a.mu = mu; a.gam = gamma; a.ptype = ptype;
    a.ha = nullptr; a.va = av_accel; a.vh = av_heat;
    a.G = delp; a.rho = rho;
    take_drag_terms(ctx, a);
    a.dt = dt; a.m_h = ctx->cst.m_h; a.kB = ctx->cst.k_B; a.lim = ctx->cst.pos_clamp;
    a.red2 = red2; a.dt_out = dt_out; a.first = first; a.fixed_dt = fixed_dt;
    a.dt_0 = ctx->cst.dt_0; a.max_age = ctx->cst.max_age;
    a.counters = ctx->badc.as<u64>();
    hipLaunchKernelGGL(dev_integrate_kernel, dim3((unsigned)((n_owned + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- gas-dust drag (nsc:719-742) on owned + ghost arrays --------------------------------------------------------------
// sorted[s] = caller[perm[s]] for four arrays at once
__global__ __launch_bounds__(256) void dev_gather4_kernel(int n, const int* perm, const double* a0, const double* a1,
                                                          const double* a2, const double* a3, double* d0, double* d1, double* d2,
                                                          double* d3) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int c = perm[s];
    d0[s] = a0[c]; d1[s] = a1[c]; d2[s] = a2[c]; d3[s] = a3[c];
}
// caller[perm[s], :] = sorted[s, :] for the first n_keep caller indices (w doubles per particle)
__global__ __launch_bounds__(256) void dev_scatter_rows_kernel(int n, const int* perm, int n_keep, int w, const double* src,
                                                               double* dst) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int c = perm[s];
    if (c >= n_keep) return;
    for (int q = 0; q < w; ++q) dst[(size_t)c * w + q] = src[(size_t)s * w + q];
}
// after sphx_dev_prep (records with the complete h): drag of the dust neighbours onto each OWNED particle -> drag_on
// (n_total,3, owned rows written), and the scatter-added reaction onto the neighbours -> drag_reaction (n_total,3, every
// row: the ghosts' rows are what their owners must still add - the driver's reverse halo).  mass, ptype,
// mean_grain_mass, mean_cross: (n_total,) caller order, ghosts included.
extern "C" int sphx_dev_drag(sphx_ctx* ctx, const double* mass, const double* ptype, const double* mean_grain_mass,
                             const double* mean_cross, double* drag_on, double* drag_reaction) {
    if (!ctx) return SPHX_E_ARG;
    NEED(mass); NEED(ptype); NEED(mean_grain_mass); NEED(mean_cross); NEED(drag_on); NEED(drag_reaction);
    if (!ctx->map_perm) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_drag before sphx_dev_search");
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));
    const int64_t n = ctx->n;
    const unsigned grid = (unsigned)((n + 255) / 256);
    StateArrays& st = ctx->alt;
    DevBuf* bufs[] = {&st.m, &st.ptype, &st.mgm, &st.mcs};
    for (DevBuf* b : bufs) SPHX_TRY(sphx_ensure(ctx, *b, (size_t)n * sizeof(double)));
    hipLaunchKernelGGL(dev_gather4_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, ctx->map_perm, mass, ptype,
                       mean_grain_mass, mean_cross, st.m.as<double>(), st.ptype.as<double>(), st.mgm.as<double>(),
                       st.mcs.as<double>());
    SPHX_TRY(sphx_pass_drag(ctx, n, ctx->k, st.m.as<double>(), st.ptype.as<double>(), st.mgm.as<double>(), st.mcs.as<double>()));
    hipLaunchKernelGGL(dev_scatter_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, ctx->map_perm, ctx->map_nactive, 3,
                       ctx->drag_on.as<double>(), drag_on);
    hipLaunchKernelGGL(dev_scatter_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, ctx->map_perm, (int)n, 3,
                       ctx->drag_re.as<double>(), drag_reaction);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
// drv:462-463,473 for the NEXT sphx_dev_integrate / _auto / _loop call: visc += drag_on rho_dust / rho [gas] + drag_reaction
// (all (n_owned, ...) device arrays, the reaction complete: own scatter + what the peers sent back)
extern "C" int sphx_dev_set_drag_terms(sphx_ctx* ctx, const double* drag_on, const double* drag_reaction, const double* rho,
                                       const double* rho_dust) {
    if (!ctx) return SPHX_E_ARG;
    NEED(drag_on); NEED(drag_reaction); NEED(rho); NEED(rho_dust);
    ctx->dev_drag_on = drag_on; ctx->dev_drag_re = drag_reaction; ctx->dev_drag_rho = rho; ctx->dev_drag_rhod = rho_dust;
    return SPHX_OK;
}

// ---- species pass (nsc:624-627) + metallicity + AGB yields on owned + ghost arrays ---------------------------------------
__global__ __launch_bounds__(256) void dev_gather_fun_kernel(int n, int S, int SP, const int* perm, const double* fun,
                                                             const double* m, double* fun_s, double* m_s) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t c = (size_t)perm[s];
    for (int q = 0; q < SP; ++q) fun_s[(size_t)s * SP + q] = q < S ? fun[c * S + q] : 0.0;
    m_s[s] = m[c];
}
__global__ __launch_bounds__(256) void dev_scatter_species_kernel(int n, int S, const int* perm, int n_keep, const double* in,
                                                                  double* out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int c = perm[s];
    if (c >= n_keep) return;
    for (int q = 0; q < S; ++q) out[(size_t)q * n + c] = in[(size_t)q * n + s];
}
extern "C" int sphx_dev_set_agb(sphx_ctx* ctx, int nspecies, int nspl, const int32_t* ntx, const int32_t* nty, const double* tx,
                                const double* ty, const double* coeffs, const int32_t* mapto, double divisor,
                                const double* mu_specie, double solar_mass) {
    if (!ctx) return SPHX_E_ARG;
    return sphx_agb_table_set(ctx, nspecies, nspl, ntx, nty, tx, ty, coeffs, mapto, divisor, mu_specie, solar_mass);
}
// after sphx_dev_prep: f_un (n_total,S) and mass (n_total,), ghosts included -> F (S,n_total) species-major, and with a
// table set Z (n_total,) and agb_dust (n_total,S); owned entries written; Z / agb_dust may be NULL
extern "C" int sphx_dev_species(sphx_ctx* ctx, int nspecies, const double* f_un, const double* mass, double* F, double* Z,
                                double* agb_dust) {
    if (!ctx) return SPHX_E_ARG;
    NEED(f_un); NEED(mass); NEED(F);
    if (!ctx->map_perm) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_species before sphx_dev_search");
    if (nspecies < 1 || nspecies > SPHX_MAX_SPECIES) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_dev_species: %d species", nspecies);
    if ((Z || agb_dust) && !(ctx->agb_on && Z && agb_dust && ctx->agb.nspec == nspecies))
        return sphx_set_err(ctx, SPHX_E_STATE, "sphx_dev_species: Z / agb_dust need both pointers and a table for %d species", nspecies);
    HIPCHK(hipSetDevice(ctx->device));
    SPHX_TRY(sphx_blob_join(ctx));
    const int64_t n = ctx->n;
    const int S = nspecies, SP = (S + 15) & ~15;
    const unsigned grid = (unsigned)((n + 255) / 256);
    StateArrays& st = ctx->alt;
    SPHX_TRY(sphx_ensure(ctx, st.fun, (size_t)n * SP * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, st.m, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->F, (size_t)n * S * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->Zmet, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->agb_dust, (size_t)n * S * sizeof(double)));
    hipLaunchKernelGGL(dev_gather_fun_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, S, SP, ctx->map_perm, f_un, mass,
                       st.fun.as<double>(), st.m.as<double>());
    SPHX_TRY(sphx_species_on(ctx, n, ctx->k, S, SP, st.fun.as<double>(), nullptr, st.m.as<double>(), ctx->F.as<double>(),
                             Z ? ctx->Zmet.as<double>() : nullptr, Z ? ctx->agb_dust.as<double>() : nullptr));
    hipLaunchKernelGGL(dev_scatter_species_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, S, ctx->map_perm,
                       ctx->map_nactive, ctx->F.as<double>(), F);
    if (Z) {
        hipLaunchKernelGGL(dev_scatter_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, ctx->map_perm, ctx->map_nactive,
                           1, ctx->Zmet.as<double>(), Z);
        hipLaunchKernelGGL(dev_scatter_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, (int)n, ctx->map_perm, ctx->map_nactive,
                           S, ctx->agb_dust.as<double>(), agb_dust);
    }
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

extern "C" int sphx_sync(sphx_ctx* ctx) {
    if (!ctx) return SPHX_E_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

// ---- fused row packing / regrouping for the halo and migration protocol -----------------------
// The driver keeps its particles as separate arrays (n,) / (n,3); what travels between ranks are
// rows (one particle = W doubles).  Two kernels replace the dozens of small gather / concatenate
// launches a tensor library needs for that:
//   pack:     rows[t, :] = concat_f fields_f[idx[t], :]                  (idx == NULL: identity)
//   regroup:  out_f[t]   = fields_f[sel[t]]           for t <  n_sel     (sel == NULL: identity)
//             out_f[t]   = rows[t - n_sel, col_f ...] for t >= n_sel
// Elements are copied as 8-byte words (double, or int64 ids).
#define ROWS_MAX_FIELDS 24
struct RowFields {
    int nf, W;
    const double* src[ROWS_MAX_FIELDS];
    double* dst[ROWS_MAX_FIELDS];
    int width[ROWS_MAX_FIELDS], col[ROWS_MAX_FIELDS];
};

__global__ __launch_bounds__(256) void pack_rows_kernel(long long n, const long long* idx, RowFields f, double* rows) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * f.W) return;
    const long long t = e / f.W;
    const int c = (int)(e - t * f.W);
    const long long i = idx ? idx[t] : t;
    int q = 0;
    while (q + 1 < f.nf && c >= f.col[q + 1]) ++q;
    rows[e] = f.src[q][i * f.width[q] + (c - f.col[q])];
}

__global__ __launch_bounds__(256) void regroup_kernel(long long n_sel, long long n_rows, const long long* sel,
                                                      const double* rows, RowFields f) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = n_sel + n_rows;
    if (e >= n * f.W) return;
    const long long t = e / f.W;
    const int c = (int)(e - t * f.W);
    int q = 0;
    while (q + 1 < f.nf && c >= f.col[q + 1]) ++q;
    double v;
    if (t < n_sel) {
        const long long i = sel ? sel[t] : t;
        v = f.src[q][i * f.width[q] + (c - f.col[q])];
    } else {
        v = rows[(t - n_sel) * f.W + c];
    }
    f.dst[q][t * f.width[q] + (c - f.col[q])] = v;
}

static int row_fields(sphx_ctx* ctx, int nf, const double* const* src, double* const* dst, const int32_t* widths,
                      RowFields* f) {
    if (nf < 1 || nf > ROWS_MAX_FIELDS) return sphx_set_err(ctx, SPHX_E_ARG, "%d fields not in 1..%d", nf, ROWS_MAX_FIELDS);
    f->nf = nf;
    int col = 0;
    for (int q = 0; q < nf; ++q) {
        if (widths[q] < 1) return sphx_set_err(ctx, SPHX_E_ARG, "field %d has width %d", q, widths[q]);
        f->src[q] = src ? src[q] : nullptr;
        f->dst[q] = dst ? dst[q] : nullptr;
        f->width[q] = widths[q];
        f->col[q] = col;
        col += widths[q];
    }
    f->W = col;
    return SPHX_OK;
}

extern "C" int sphx_dev_pack_rows(sphx_ctx* ctx, int64_t n, const int64_t* idx, int nf, const double* const* fields,
                                  const int32_t* widths, double* rows) {
    if (!ctx) return SPHX_E_ARG;
    NEED(fields); NEED(widths);
    if (n < 0) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld < 0", (long long)n);
    if (n == 0) return SPHX_OK;
    NEED(rows);
    HIPCHK(hipSetDevice(ctx->device));
    RowFields f;
    SPHX_TRY(row_fields(ctx, nf, fields, nullptr, widths, &f));
    for (int q = 0; q < nf; ++q) NEED(fields[q]);
    const long long tot = (long long)n * f.W;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n,
                       (const long long*)idx, f, rows);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

extern "C" int sphx_dev_regroup(sphx_ctx* ctx, int64_t n_sel, const int64_t* sel, int64_t n_rows, const double* rows,
                                int nf, const double* const* fields_in, const int32_t* widths,
                                double* const* fields_out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(widths); NEED(fields_out);
    if (n_sel < 0 || n_rows < 0) return sphx_set_err(ctx, SPHX_E_ARG, "negative count");
    if (n_sel + n_rows == 0) return SPHX_OK;
    if (n_sel > 0) NEED(fields_in);
    if (n_rows > 0) NEED(rows);
    HIPCHK(hipSetDevice(ctx->device));
    RowFields f;
    SPHX_TRY(row_fields(ctx, nf, n_sel > 0 ? fields_in : nullptr, fields_out, widths, &f));
    for (int q = 0; q < nf; ++q) NEED(fields_out[q]);
    const long long tot = (long long)(n_sel + n_rows) * f.W;
    hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                       (long long)n_sel, (long long)n_rows, (const long long*)sel, rows, f);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- need map of the decomposed driver (multigpu.py: DistributedSim._need_map) ----------------
// out[c] = 1 for every cell c of a G^3 grid of cubic cells (origin g_lo, edge cs) that lies within
// floor(w_i / cs) + 1 cells (Chebyshev) of the cell holding an owned particle i claiming reach w_i:
// a superset of the cells from which a point can lie within w_i of particle i.  One thread per
// particle writes its cube (27 bytes for nearly all; racing stores of the same value).
// Claims wider than rwide cells are not walked here: they go into wide0 (the widest per cell) for the target-side pass below.
__global__ __launch_bounds__(256) void need_map_kernel(long long n, const double* pos, const double* w, double lx,
                                                       double ly, double lz, double cs, int G,
                                                       unsigned char* out, int rwide, u64* wide0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double wi = w[i];
    if (wi != wi) wi = INFINITY;                   // a NaN reach (a broken radius) claims everything: never a missing ghost
    if (!(wi > 0.0)) return;                       // claims nothing
    const double g1 = (double)(G - 1);
    // the particle's own cell: floor + clamp, as torch.floor(...).clamp_(0, G - 1) does
    const double tx = fmin(fmax(floor((pos[3 * i] - lx) / cs), 0.0), g1);
    const double ty = fmin(fmax(floor((pos[3 * i + 1] - ly) / cs), 0.0), g1);
    const double tz = fmin(fmax(floor((pos[3 * i + 2] - lz) / cs), 0.0), g1);
    const int cx = (tx == tx) ? (int)tx : 0, cy = (ty == ty) ? (int)ty : 0, cz = (tz == tz) ? (int)tz : 0;
    const double q = wi / cs;
    const double rr = floor(q) + 1.0;
    const int r = rr < (double)G ? (int)rr : G;
    if (r > rwide) {
        atomicMax(&wide0[((size_t)cz * G + cy) * G + cx], (u64)__double_as_longlong(wi));     // (positive doubles order like their bits)
        return;
    }
    const double q2 = q * q;                       // (inf for an overflowing reach: everything is claimed)
    const int x0 = max(cx - r, 0), x1 = min(cx + r, G - 1);
    const int y0 = max(cy - r, 0), y1 = min(cy + r, G - 1);
    const int z0 = max(cz - r, 0), z1 = min(cz + r, G - 1);
    // a SPHERE of cells, not the cube around it (whose corners reach 1.7 x as far - into the dense cloud, for the
    // wide claims of rim particles): a cell d = (dx, dy, dz) cells away can hold a point within w only if
    // sum_a max(|d_a| - 1, 0)^2 <= (w / cell)^2  (integers against one double: the tensor-library form does the same)
    for (int z = z0; z <= z1; ++z) {
        const int az = max(abs(z - cz) - 1, 0);
        for (int y = y0; y <= y1; ++y) {
            const int ay = max(abs(y - cy) - 1, 0);
            const int s2 = az * az + ay * ay;
            if (!((double)s2 <= q2)) continue;
            unsigned char* row = out + ((size_t)z * G + y) * G;
            for (int x = x0; x <= x1; ++x) {
                const int ax = max(abs(x - cx) - 1, 0);
                if ((double)(s2 + ax * ax) <= q2) row[x] = 1;
            }
        }
    }
}

// The same map for any mix of claims at a bounded cost.  The kernel above walks, for every owned particle, the whole cube
// of cells around it: fine for claims of a few cells, but an expanding cloud's rim claims spheres 50+ cells wide -
// (2r+1)^3 cells each, one thread per particle, thousands of them overlapping: 43 ms for 5000 such claims at G = 96
// (measured) where the whole map has 885 k cells.  So claims wider than NM_RWIDE cells are marked from the TARGET side:
// (1) the widest such claim per cell (the rule is monotone in w: a cell's widest claim stands for all its particles),
// (2) a pyramid of maxima over 2x2x2 blocks, (3) one thread per target cell descends the pyramid - nearest child
// first, a node pruned when even its nearest cell is out of its widest claim's reach - and stops at the first source
// cell that reaches it.  Same bytes (test_need_map_kernel_matches_tensor_form); 43 ms -> 0.7 ms on that case, and next
// to nothing when no claim is wide (every top node is empty).
#define NM_RWIDE 6
#define NM_MAXLEV 12
struct NeedPyr { int nlev; int G[NM_MAXLEV]; long long off[NM_MAXLEV]; };      // level l: G[l]^3 u64 at off[l]; level 0 = cells
__global__ __launch_bounds__(256) void nm_up_kernel(int Gc, int Gp, const u64* child, u64* parent) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Gp * Gp * Gp) return;
    const int px = t % Gp, py = (t / Gp) % Gp, pz = t / (Gp * Gp);
    u64 m = 0;
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int x = 2 * px + dx, y = 2 * py + dy, z = 2 * pz + dz;
                if (x < Gc && y < Gc && z < Gc) {
                    const u64 v = child[((size_t)z * Gc + y) * Gc + x];
                    m = v > m ? v : m;
                }
            }
    parent[t] = m;
}
__global__ __launch_bounds__(256) void nm_target_kernel(NeedPyr p, const u64* pyr, double cs, unsigned char* out) {
    const int G = p.G[0];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * G * G) return;
    const int cx = t % G, cy = (t / G) % G, cz = t / (G * G);
    // depth-first, explicit stack of (level, node) codes; children visited nearest first
    unsigned stack[64 + 8 * NM_MAXLEV];            // the top level's nodes (at most 4^3) + seven more per level of descent
    int sp = 0;
    const int top = p.nlev - 1, Gt = p.G[top];
    for (int q = Gt * Gt * Gt - 1; q >= 0; --q) stack[sp++] = ((unsigned)top << 27) | (unsigned)q;      // (Gt <= 4: at most 64)
    bool hit = false;
    while (sp > 0 && !hit) {
        const unsigned code = stack[--sp];
        const int lev = (int)(code >> 27);
        const int Gl = p.G[lev];
        const int q = (int)(code & 0x7FFFFFFu);
        const int bx = q % Gl, by = (q / Gl) % Gl, bz = q / (Gl * Gl);
        const u64 wb = pyr[p.off[lev] + q];
        if (wb == 0) continue;                                       // nobody claims from here
        // the node's cells: [b << lev, min(((b + 1) << lev) - 1, G - 1)] per axis; per-axis distance to the nearest of them
        const int x0 = bx << lev, y0 = by << lev, z0 = bz << lev;
        const int x1 = min(((bx + 1) << lev) - 1, G - 1), y1 = min(((by + 1) << lev) - 1, G - 1), z1 = min(((bz + 1) << lev) - 1, G - 1);
        const int dx = max(max(x0 - cx, cx - x1), 0), dy = max(max(y0 - cy, cy - y1), 0), dz = max(max(z0 - cz, cz - z1), 0);
        const int ax = max(dx - 1, 0), ay = max(dy - 1, 0), az = max(dz - 1, 0);
        const double qv = __longlong_as_double((long long)wb) / cs;
        if (!((double)(ax * ax + ay * ay + az * az) <= qv * qv)) continue;          // out of the widest claim's reach
        if (lev == 0) { hit = true; break; }
        // children: the one nearest to the target last on the stack (= first off it)
        const int lc = lev - 1, Gc = p.G[lc];
        const int px = (cx >> lc) > 2 * bx ? 1 : 0, py = (cy >> lc) > 2 * by ? 1 : 0, pz = (cz >> lc) > 2 * bz ? 1 : 0;
        const int pref = px | (py << 1) | (pz << 2);
        for (int k = 7; k >= 0; --k) {
            const int c = k ^ pref;
            const int x = 2 * bx + (c & 1), y = 2 * by + ((c >> 1) & 1), z = 2 * bz + (c >> 2);
            if (x < Gc && y < Gc && z < Gc) stack[sp++] = ((unsigned)lc << 27) | (unsigned)((z * Gc + y) * Gc + x);
        }
    }
    if (hit) out[t] = 1;                               // (beside the narrow claims' marks)
}

extern "C" int sphx_dev_need_map(sphx_ctx* ctx, int64_t n, const double* pos, const double* w, const double* g_lo,
                                 double g_cs, int G, unsigned char* out) {
    if (!ctx) return SPHX_E_ARG;
    NEED(g_lo); NEED(out);
    if (n < 0 || G < 1 || G > 512 || !(g_cs > 0.0)) return sphx_set_err(ctx, SPHX_E_ARG, "need map: n=%lld G=%d (1..512) cs=%g", (long long)n, G, g_cs);
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) {
        HIPCHK(hipMemsetAsync(out, 0, (size_t)G * G * G, ctx->stream));
        return SPHX_OK;
    }
    NEED(pos); NEED(w);
    NeedPyr p;
    p.nlev = 0;
    long long tot = 0;
    for (int g = G;; g = (g + 1) / 2) {
        p.G[p.nlev] = g; p.off[p.nlev] = tot;
        tot += (long long)g * g * g;
        ++p.nlev;
        if (g <= 4 || p.nlev == NM_MAXLEV) break;
    }
    if (p.G[p.nlev - 1] > 4) return sphx_set_err(ctx, SPHX_E_ARG, "need map: G=%d too large for the pyramid", G);
    SPHX_TRY(sphx_ensure(ctx, ctx->need_pyr, (size_t)tot * sizeof(u64)));
    u64* pyr = ctx->need_pyr.as<u64>();
    HIPCHK(hipMemsetAsync(pyr, 0, (size_t)G * G * G * sizeof(u64), ctx->stream));        // (level 0; the others are overwritten)
    HIPCHK(hipMemsetAsync(out, 0, (size_t)G * G * G, ctx->stream));
    hipLaunchKernelGGL(need_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, pos, w,
                       g_lo[0], g_lo[1], g_lo[2], g_cs, G, out, NM_RWIDE, pyr);
    for (int l = 1; l < p.nlev; ++l) {
        const int gp = p.G[l];
        hipLaunchKernelGGL(nm_up_kernel, dim3((unsigned)((gp * gp * gp + 255) / 256)), dim3(256), 0, ctx->stream, p.G[l - 1], gp,
                           pyr + p.off[l - 1], pyr + p.off[l]);
    }
    hipLaunchKernelGGL(nm_target_kernel, dim3((unsigned)((G * G * G + 255) / 256)), dim3(256), 0, ctx->stream, p, pyr, g_cs, out);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- small fused helpers of the decomposed driver (each replaces a dozen tensor-library launches) ------
// reach claimed by every owned particle (DistributedSim._replan):
//   w_i = max((halo + skin) h_i, halo h_i + |v_i| dt)
// or, with a cap c > 0 on the head-room (sphx_dev_set_reach_cap: an absolute length, a few mean radii - a particle whose
// radius is many times the mean, the rim of an expanding cloud, then claims h_i + c instead of 1.3 h_i):
//   w_i = max(h_i + min((halo + skin - 1) h_i, c), h_i + min((halo - 1) h_i, c) + |v_i| dt)
__device__ __forceinline__ double reach_of(double hi, double speed, double dt, double halo, double skin, double cap) {
#pragma clang fp contract(off)          // each operation rounded on its own, as the tensor-library form does
    if (!(cap > 0.0)) return fmax((halo + skin) * hi, halo * hi + speed * dt);
    const double a = fmin((halo + skin - 1.0) * hi, cap), b = fmin((halo - 1.0) * hi, cap);
    return fmax(hi + a, (hi + b) + speed * dt);
}
__global__ __launch_bounds__(256) void reach_kernel(long long n, const double* h, const double* vel, double halo,
                                                    double skin, double dt, double cap, double* w) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
    const double speed = sqrt((vx * vx + vy * vy) + vz * vz);
    const double hi = h[i];
    w[i] = reach_of(hi, speed, dt, halo, skin, cap);
}
extern "C" int sphx_dev_set_reach_cap(sphx_ctx* ctx, double cap) {
    if (!ctx) return SPHX_E_ARG;
    ctx->reach_cap = cap > 0.0 ? cap : 0.0;
    return SPHX_OK;
}
extern "C" int sphx_dev_reach(sphx_ctx* ctx, int64_t n, const double* h, const double* vel, double halo_scale,
                              double skin_frac, double dt_last, double* w) {
    if (!ctx) return SPHX_E_ARG;
    if (n < 0) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld", (long long)n);
    if (n == 0) return SPHX_OK;
    NEED(h); NEED(vel); NEED(w);
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(reach_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, h, vel,
                       halo_scale, skin_frac, dt_last, ctx->reach_cap, w);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// the same with dt read from device memory (the step's dt as sphx_dev_integrate_auto left it): the next step's plan is
// made before the host has seen this step's scalars
__global__ __launch_bounds__(256) void reach_dt_kernel(long long n, const double* h, const double* vel, double halo,
                                                       double skin, const double* dt_dev, double cap, double* w) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dt = *dt_dev;
    const double vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
    const double speed = sqrt((vx * vx + vy * vy) + vz * vz);
    const double hi = h[i];
    w[i] = reach_of(hi, speed, dt, halo, skin, cap);
}
extern "C" int sphx_dev_reach_dt(sphx_ctx* ctx, int64_t n, const double* h, const double* vel, double halo_scale,
                                 double skin_frac, const double* dt_dev, double* w) {
    if (!ctx) return SPHX_E_ARG;
    if (n < 0) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld", (long long)n);
    if (n == 0) return SPHX_OK;
    NEED(h); NEED(vel); NEED(w); NEED(dt_dev);
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(reach_dt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, h, vel,
                       halo_scale, skin_frac, dt_dev, ctx->reach_cap, w);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// Who gets which owned particle as a ghost (DistributedSim._plan): mask[p][i] = 1 when rank p's need map covers the
// coarse cell of particle i (p != rank), counts[p] = how many - one launch instead of the tensor-library chain (cell
// ids, a W x n gather, compare, sum).  The cell is the one need_map_kernel gives the particle.
__global__ __launch_bounds__(256) void plan_mask_kernel(long long n, const double* pos, double lx, double ly, double lz,
                                                        double cs, int G, int W, int rank, const unsigned char* maps,
                                                        unsigned char* mask, long long* counts) {
    __shared__ int wsum[4];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    size_t cell = 0;
    if (i < n) {
        const double g1 = (double)(G - 1);
        const double tx = fmin(fmax(floor((pos[3 * i] - lx) / cs), 0.0), g1);
        const double ty = fmin(fmax(floor((pos[3 * i + 1] - ly) / cs), 0.0), g1);
        const double tz = fmin(fmax(floor((pos[3 * i + 2] - lz) / cs), 0.0), g1);
        const int cx = (tx == tx) ? (int)tx : 0, cy = (ty == ty) ? (int)ty : 0, cz = (tz == tz) ? (int)tz : 0;
        cell = ((size_t)cz * G + cy) * G + cx;
    }
    const size_t G3 = (size_t)G * G * G;
    for (int p = 0; p < W; ++p) {
        unsigned char m = 0;
        if (i < n && p != rank) m = maps[(size_t)p * G3 + cell] != 0 ? 1 : 0;
        if (i < n) mask[(size_t)p * n + i] = m;
        const int c = __popcll(__builtin_amdgcn_ballot_w64(m != 0));
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (t) atomicAdd((unsigned long long*)&counts[p], (unsigned long long)t);
        }
        __syncthreads();
    }
}
extern "C" int sphx_dev_plan_mask(sphx_ctx* ctx, int64_t n, const double* pos, const double* g_lo, double g_cs, int G,
                                  int world, int rank, const unsigned char* maps, unsigned char* mask, int64_t* counts) {
    if (!ctx) return SPHX_E_ARG;
    NEED(g_lo); NEED(counts);
    if (n < 0 || G < 1 || G > 1024 || !(g_cs > 0.0) || world < 1 || world > 4096 || rank < 0 || rank >= world)
        return sphx_set_err(ctx, SPHX_E_ARG, "plan mask: n=%lld G=%d cs=%g world=%d rank=%d", (long long)n, G, g_cs, world, rank);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemsetAsync(counts, 0, (size_t)world * sizeof(int64_t), ctx->stream));
    if (n == 0) return SPHX_OK;
    NEED(pos); NEED(maps); NEED(mask);
    hipLaunchKernelGGL(plan_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, pos,
                       g_lo[0], g_lo[1], g_lo[2], g_cs, G, world, rank, maps, mask, (long long*)counts);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// end-of-step scalars of one rank (DistributedSim.step): out[0] = 1 if any h_i + 2 D > w_i (the halo was too
// thin), out[1] = -crossing time (*ct, NULL: left alone), out[2] = max h, out[3] = mean of the h <= hclip
// (hclip <= 0: of all).  One launch; the last block to finish (a ticket that wraps to 0) folds the partials
// in block order.
#define SCAL_BLOCKS 256
__global__ __launch_bounds__(256) void step_scalars_kernel(long long n, const double* h, const double* w, double D2,
                                                           double hclip, const double* ct, double* partial,
                                                           unsigned* ticket, double* out) {
    __shared__ double sm[4][4];
    __shared__ bool last;
    double bad = 0.0, hmax = 0.0, hs = 0.0, hc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double hi = h[i];
        if (hi + D2 > w[i]) bad = 1.0;
        hmax = fmax(hmax, hi);
        if (!(hclip > 0.0) || hi <= hclip) { hs += hi; hc += 1.0; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bad = fmax(bad, __shfl_xor(bad, o, 64)); hmax = fmax(hmax, __shfl_xor(hmax, o, 64));
        hs += __shfl_xor(hs, o, 64); hc += __shfl_xor(hc, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wv][0] = bad; sm[wv][1] = hmax; sm[wv][2] = hs; sm[wv][3] = hc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* pp = partial + 4 * blockIdx.x;
        pp[0] = fmax(fmax(sm[0][0], sm[1][0]), fmax(sm[2][0], sm[3][0]));
        pp[1] = fmax(fmax(sm[0][1], sm[1][1]), fmax(sm[2][1], sm[3][1]));
        pp[2] = (sm[0][2] + sm[1][2]) + (sm[2][2] + sm[3][2]);
        pp[3] = (sm[0][3] + sm[1][3]) + (sm[2][3] + sm[3][3]);
        __threadfence();
        last = atomicInc(ticket, gridDim.x - 1) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {                                    // (the whole block or none of it) one partial per thread,
        __threadfence();                           // then the same fixed tree as above
        double b = 0.0, m = 0.0, s_ = 0.0, c = 0.0;
        for (unsigned q = threadIdx.x; q < gridDim.x; q += blockDim.x) {
            const double* pp = partial + 4 * q;
            b = fmax(b, __builtin_nontemporal_load(pp)); m = fmax(m, __builtin_nontemporal_load(pp + 1));
            s_ += __builtin_nontemporal_load(pp + 2); c += __builtin_nontemporal_load(pp + 3);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            b = fmax(b, __shfl_xor(b, o, 64)); m = fmax(m, __shfl_xor(m, o, 64));
            s_ += __shfl_xor(s_, o, 64); c += __shfl_xor(c, o, 64);
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { sm[wv][0] = b; sm[wv][1] = m; sm[wv][2] = s_; sm[wv][3] = c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            out[0] = fmax(fmax(sm[0][0], sm[1][0]), fmax(sm[2][0], sm[3][0]));
            if (ct) out[1] = -*ct;
            out[2] = fmax(fmax(sm[0][1], sm[1][1]), fmax(sm[2][1], sm[3][1]));
            out[3] = ((sm[0][2] + sm[1][2]) + (sm[2][2] + sm[3][2])) / fmax((sm[0][3] + sm[1][3]) + (sm[2][3] + sm[3][3]), 1.0);
        }
    }
}
extern "C" int sphx_dev_step_scalars(sphx_ctx* ctx, int64_t n_owned, const double* h, const double* w_plan, double D,
                                     double hclip, const double* ct, double* out4) {
    if (!ctx) return SPHX_E_ARG;
    NEED(out4);
    if (n_owned < 1) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_dev_step_scalars: n_owned=%lld", (long long)n_owned);
    NEED(h); NEED(w_plan);
    HIPCHK(hipSetDevice(ctx->device));
    const bool fresh = ctx->scal_tmp.p == nullptr;
    SPHX_TRY(sphx_ensure(ctx, ctx->scal_tmp, (size_t)(4 * SCAL_BLOCKS + 2) * sizeof(double)));
    double* partial = ctx->scal_tmp.as<double>();
    unsigned* ticket = reinterpret_cast<unsigned*>(partial + 4 * SCAL_BLOCKS);
    if (fresh) HIPCHK(hipMemsetAsync(ticket, 0, sizeof(double), ctx->stream));
    int blocks = (int)((n_owned + 255) / 256);
    if (blocks > SCAL_BLOCKS) blocks = SCAL_BLOCKS;
    hipLaunchKernelGGL(step_scalars_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (long long)n_owned, h, w_plan,
                       2.0 * D, hclip, ct, partial, ticket, out4);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}
