// sphx_api.hip - the C ABI of include/sphx.h: context, host-pointer entry points that mirror
// the reference's Python callables, and the device-resident step loop.
#include "sphx_internal.h"
#include <stdarg.h>
#include <stdlib.h>
#include <new>

int sphx_set_err(sphx_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

int sphx_ensure(sphx_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (b.cap >= bytes) return SPHX_OK;
    if (b.p) {
        if (ctx->capturing) return sphx_set_err(ctx, SPHX_E_STATE, "a buffer would have to grow while a step is being captured");
        sphx_graph_drop(ctx);                  // (a recorded step holds the old address)
        // buffers may still be referenced by queued work
        HIPCHK(hipStreamSynchronize(ctx->stream));
        sphx_graph_reap(ctx);
        HIPCHK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return sphx_set_err(ctx, SPHX_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return SPHX_OK;
}

static void default_constants(sphx_constants* c) {
    c->k_B = 1.380649e-23;
    c->amu = 1.66053906892e-27;
    c->m_h = 1.0008 * c->amu;
    c->m_0 = 31.622776601683793 * 1.989e30;          // 10**1.5 * solar_mass, nsc:36
    c->dt_0 = 60. * 60. * 24. * 365. * 250000.;
    c->max_age = 3e7 * 60. * 60. * 24. * 365.;
    c->pos_clamp = 1e11 * 149597870700.0;
}

extern "C" int sphx_version(void) { return 100; }

// How this library was built, and what a context read from the environment when it was created (bench.py puts both into
// its line).  "experiments=0" is the product: no result-changing or extra-launch switches are compiled in.
extern "C" const char* sphx_build_info(void) {
#ifdef SPHX_EXPERIMENTS
    return "gfx950 experiments=1"
#else
    return "gfx950 experiments=0"
#endif
#ifdef SPHX_KNN_PROF
           " knn_prof=1"
#endif
        ;
}
extern "C" const char* sphx_tunables(const sphx_ctx* ctx) { return ctx ? ctx->tunables : ""; }

extern "C" int sphx_host_alloc(void** out, size_t bytes) {
    if (!out) return SPHX_E_ARG;
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return SPHX_E_NOMEM;
    }
    return SPHX_OK;
}
extern "C" int sphx_host_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) return SPHX_E_HIP;
    return SPHX_OK;
}

// The step's main stream and its side stream (state permutation's second half, record build, h sums, read-backs).
// The main stream at the device's highest priority, the side stream at its lowest: what the step waits for is dispatched
// first (step 1.257 -> 1.242 ms, four runs each; SPHX_STREAM_PRIO=0: both at the default priority).
static hipError_t sphx_stream_create(hipStream_t* st, int side, bool prio) {
    if (!prio) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, side ? least : greatest);
}

extern "C" int sphx_create(sphx_ctx** out, int device) {
    if (!out) return SPHX_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        fprintf(stderr, "sphx_create: no usable HIP device %d (found %d). libsphx has no CPU path.\n",
                device, count);
        return SPHX_E_HIP;
    }
    sphx_ctx* ctx = new (std::nothrow) sphx_ctx();
    if (!ctx) return SPHX_E_NOMEM;
    ctx->device = device;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    memset(&ctx->grid, 0, sizeof(ctx->grid));
    default_constants(&ctx->cst);
    // Tunables: read from the environment ONCE, here, and recorded (sphx_tunables): a bench line can show the
    // configuration it ran.  None of them changes a result (every variant is pinned bit for bit to the default by the
    // tests); switches that do - timing experiments, diagnostics - exist only in -DSPHX_EXPERIMENTS builds.
    auto env = [&](const char* name) -> const char* {
        const char* e = getenv(name);
        if (e) {
            const size_t used = strlen(ctx->tunables);
            snprintf(ctx->tunables + used, sizeof(ctx->tunables) - used, "%s%s=%s", used ? " " : "", name, e);
        }
        return e;
    };
    if (const char* e = env("SPHX_RSCALE")) { double v = atof(e); if (v >= 1.0) ctx->rscale = v; }
    if (const char* e = env("SPHX_CELL")) { double v = atof(e); if (v > 0.0) ctx->cell_factor = v; }
    if (const char* e = env("SPHX_RSCALE_BUILD")) { double v = atof(e); if (v >= 1.0) ctx->rscale_build = v; }
    if (const char* e = env("SPHX_VERLET")) ctx->use_verlet = atoi(e) != 0;
    if (const char* e = env("SPHX_BLOB")) ctx->use_blob = atoi(e) != 0;
    if (const char* e = env("SPHX_BLOB_CURVE")) ctx->blob_curve = atoi(e);
    if (const char* e = env("SPHX_KNN_GROUP")) ctx->use_group = atoi(e) != 0;
    if (const char* e = env("SPHX_HCLIP")) { double v = atof(e); if (v > 1.0) ctx->h_clip_factor = v; }
    if (const char* e = env("SPHX_TIMING_DETAIL")) ctx->timing_detail = atoi(e) != 0;
    if (const char* e = env("SPHX_MAX_CELLS")) ctx->max_cells = atoll(e);
#ifndef SPHX_EXPERIMENTS
    if (ctx->max_cells < 0) ctx->max_cells = 0;      // (lifting the limit is an experiment)
#endif
    if (const char* e = env("SPHX_CELL_FEEDBACK")) ctx->cell_feedback = atoi(e) != 0;
    if (const char* e = env("SPHX_CELL_FB_HI")) ctx->cell_fb_hi = atof(e);
    if (const char* e = env("SPHX_CELL_FB_LO")) ctx->cell_fb_lo = atof(e);
    if (const char* e = env("SPHX_DRAG_LDS")) ctx->drag_lds = atoi(e) != 0;
    if (const char* e = env("SPHX_SPLIT_PERM")) ctx->split_perm = atoi(e) != 0;
    if (const char* e = env("SPHX_BLOB_SPLIT")) ctx->blob_split_on = atoi(e) != 0;
    if (const char* e = env("SPHX_DEV_FORK_DEDUP")) ctx->dev_fork_dedup = atoi(e) != 0;
    if (const char* e = env("SPHX_TIE_FIX")) ctx->tie_fix = atoi(e) != 0;
    if (const char* e = env("SPHX_STREAM_PRIO")) ctx->stream_prio = atoi(e) != 0;
    if (const char* e = env("SPHX_SCATTER_RANK")) ctx->scatter_by_rank = atoi(e) != 0;
    if (const char* e = env("SPHX_SPECIES_FUSED")) ctx->species_fused = atoi(e) != 0;
    if (const char* e = env("SPHX_BB_DIRECT")) ctx->bb_direct = atoi(e) != 0;
    if (const char* e = env("SPHX_SCAN_ROCPRIM")) ctx->scan_rocprim = atoi(e) != 0;
    if (const char* e = env("SPHX_SPECIES_LDS")) ctx->species_lds = atoi(e) != 0;
    if (const char* e = env("SPHX_HINT_DISTRUST")) ctx->distrust_mode = atoi(e);     // 0 never, 1 always, 2 auto
    if (const char* e = env("SPHX_OUTLIER_LEVELS")) ctx->olev_mode = atoi(e);     // 0 off, 1 always, 2 when far queries were met
    if (const char* e = env("SPHX_FUSE_COUNT")) ctx->fuse_count = atoi(e) != 0;
    if (const char* e = env("SPHX_BOX_SIGMAS")) { double v = atof(e); if (v >= 1.0) ctx->box_sigmas = v; }
    if (const char* e = env("SPHX_GRAV_KERNEL")) ctx->grav_per_thread = atoi(e) == 0;
    if (const char* e = env("SPHX_GRAV_ORDER")) { int v = atoi(e); if (v == 1 || v == 2) ctx->grav_order = v; }
    if (const char* e = env("SPHX_GRAV_WS")) { int v = atoi(e); if (v >= 1 && v <= 4) ctx->grav_ws = v; }
    if (const char* e = env("SPHX_LDS")) ctx->use_lds = atoi(e) != 0;
    if (const char* e = env("SPHX_BLOB_SLOTS")) ctx->blob_slots = atoi(e);
    if (const char* e = env("SPHX_GRAPH")) ctx->graph_mode = atoi(e);           // 0 never, 1 always, 2 for n <= SPHX_GRAPH_MAX_N
    if (const char* e = env("SPHX_GRAPH_MAX_N")) ctx->graph_max_n = atoll(e);
    if (const char* e = env("SPHX_GRAPH_EPOCH")) { int v = atoi(e); if (v >= 2) ctx->graph_epoch = v; }
    if (const char* e = env("SPHX_BLOB_WGS")) {      // workgroups per CU of the LDS passes' persistent grid
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        if (atoi(e) > 0) ctx->blob_grid = ((cus * atoi(e) + 7) / 8) * 8;
    }
#ifdef SPHX_EXPERIMENTS
    // timing experiments and diagnostics: extra discarded launches, cut-down kernels, counters printed to stderr
    if (const char* e = env("SPHX_KNN_ABL")) ctx->exp_knn = atoi(e);
    if (const char* e = env("SPHX_BLOB_EXP")) ctx->exp_blob = atoi(e);
    if (const char* e = env("SPHX_BLOB_EXP_LDS")) ctx->exp_blob_lds = (size_t)atoi(e);
    if (const char* e = env("SPHX_PASS_EXP")) ctx->exp_pass = atoi(e);
    ctx->exp_no_agb = env("SPHX_EXP_NO_AGB") != nullptr;
    ctx->knn_prof_print = env("SPHX_KNN_PROF") != nullptr;
    ctx->kg_debug_print = env("SPHX_KG_DEBUG") != nullptr;
    (void)env("SPHX_KG_EXP_NOAMB"); (void)env("SPHX_KG_PROF");       // (read where they act, sphx_knn_group.hip; recorded here)
#endif
    bool ok = hipSetDevice(device) == hipSuccess &&
              sphx_stream_create(&ctx->stream, 0, ctx->stream_prio) == hipSuccess &&
              hipHostMalloc(&ctx->pinned, 16384, hipHostMallocDefault) == hipSuccess;
    ctx->own_stream = ctx->stream;
    for (int i = 0; ok && i < 10; ++i) ok = hipEventCreate(&ctx->ev[i]) == hipSuccess;
    for (int r = 0; ok && r < 3; ++r)
        for (int i = 0; ok && i < 10; ++i) ok = hipEventCreate(&ctx->evring[r][i]) == hipSuccess;
    for (int r = 0; ok && r < 2; ++r)
        ok = hipEventCreateWithFlags(&ctx->lag_bev[r], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->lag_hev[r], hipEventDisableTiming) == hipSuccess;
    if (ok) ok = sphx_stream_create(&ctx->side_stream, 1, ctx->stream_prio) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->ev_perm_fork, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->ev_perm, hipEventDisableTiming) == hipSuccess;
    if (ok) ok = sphx_ensure(ctx, ctx->scal, SC_NSLOTS * 8) == SPHX_OK &&
                 hipMemsetAsync(ctx->scal.p, 0, SC_NSLOTS * 8, ctx->stream) == hipSuccess &&
                 sphx_ensure(ctx, ctx->badc, (size_t)BADC_BUCKETS * BADC_STRIDE * sizeof(u64)) == SPHX_OK &&
                 hipMemsetAsync(ctx->badc.p, 0, (size_t)BADC_BUCKETS * BADC_STRIDE * sizeof(u64), ctx->stream) == hipSuccess;
    if (!ok) {
        fprintf(stderr, "sphx_create: HIP initialisation failed on device %d\n", device);
        delete ctx;
        return SPHX_E_HIP;
    }
    *out = ctx;
    return SPHX_OK;
}

static void free_buf(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}
static void free_state(StateArrays& s) {
    DevBuf* all[] = {&s.x, &s.y, &s.z, &s.vx, &s.vy, &s.vz, &s.ax, &s.ay, &s.az, &s.m, &s.T,
                     &s.mu, &s.gam, &s.E, &s.hprev, &s.ptype, &s.id, &s.fun, &s.mgm, &s.mcs};
    for (DevBuf* b : all) free_buf(*b);
}

extern "C" void sphx_destroy(sphx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    sphx_graph_drop(ctx);
    sphx_graph_reap(ctx);
    DevBuf* all[] = {&ctx->rec1, &ctx->recv, &ctx->nbr, &ctx->rho, &ctx->rhod, &ctx->nden, &ctx->G,
                     &ctx->Pi, &ctx->Bw, &ctx->rho_s, &ctx->bc_s, &ctx->self_s, &ctx->drag_on, &ctx->drag_re, &ctx->grav, &ctx->grav_sort, &ctx->grav_tmp, &ctx->grav_pyr, &ctx->grav_cell, &ctx->lrec_a, &ctx->lrec_v, &ctx->porder, &ctx->mcount, &ctx->mstart, &ctx->slot16, &ctx->uniq, &ctx->list64, &ctx->dref, &ctx->pos0, &ctx->pos4, &ctx->va, &ctx->vh, &ctx->ha, &ctx->F,
                     &ctx->scal, &ctx->cell_of, &ctx->cell_start, &ctx->cell_fill, &ctx->perm,
                     &ctx->inv, &ctx->scan_tmp, &ctx->bbox_tmp, &ctx->in_a, &ctx->in_b, &ctx->in_c,
                     &ctx->in_d, &ctx->in_e, &ctx->in_f, &ctx->in_g, &ctx->in_h, &ctx->in_i,
                     &ctx->in_j, &ctx->out_a, &ctx->out_b, &ctx->out_c, &ctx->idx64, &ctx->dist_out,
                     &ctx->nontriv, &ctx->h_api, &ctx->hsum_tmp, &ctx->grav_quad, &ctx->scal_tmp, &ctx->fail_list,
                     &ctx->agb_knots, &ctx->Zmet, &ctx->agb_dust, &ctx->need_pyr, &ctx->ds_cnt, &ctx->ds_start, &ctx->ds_ent,
                     &ctx->loop_side, &ctx->crowded, &ctx->fun_id, &ctx->olev_start, &ctx->olev_fill, &ctx->olev_list, &ctx->olev_key, &ctx->blob_class, &ctx->blob_split, &ctx->badc, &ctx->tie_list, &ctx->lbs_state[0], &ctx->lbs_state[1], &ctx->cell_rank};
    for (DevBuf* b : all) free_buf(*b);
    free_state(ctx->st);
    free_state(ctx->alt);
    for (int i = 0; i < 10; ++i)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int r = 0; r < 3; ++r)
        for (int i = 0; i < 10; ++i)
            if (ctx->evring[r][i]) (void)hipEventDestroy(ctx->evring[r][i]);
    for (int r = 0; r < 2; ++r) {
        if (ctx->lag_bev[r]) (void)hipEventDestroy(ctx->lag_bev[r]);
        if (ctx->lag_hev[r]) (void)hipEventDestroy(ctx->lag_hev[r]);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->olev_ev) (void)hipEventDestroy(ctx->olev_ev);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_perm_fork) (void)hipEventDestroy(ctx->ev_perm_fork);
    if (ctx->ev_perm) (void)hipEventDestroy(ctx->ev_perm);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char* sphx_last_error(const sphx_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int sphx_set_constants(sphx_ctx* ctx, const sphx_constants* c) {
    if (!ctx || !c) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    ctx->cst = *c;
    return SPHX_OK;
}
extern "C" int sphx_set_tuning(sphx_ctx* ctx, double rscale, double cell_factor) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    if (rscale > 0.0) {
        if (rscale < 1.0) return sphx_set_err(ctx, SPHX_E_ARG, "rscale %g < 1", rscale);
        ctx->rscale = rscale;
    }
    if (cell_factor > 0.0) ctx->cell_factor = cell_factor;
    return SPHX_OK;
}
extern "C" int sphx_set_incremental(sphx_ctx* ctx, int verlet, double rscale_build) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    ctx->use_verlet = verlet != 0;
    if (!ctx->use_verlet) ctx->list_valid = false;
    if (rscale_build > 0.0) {
        if (rscale_build < 1.0) return sphx_set_err(ctx, SPHX_E_ARG, "rscale_build %g < 1", rscale_build);
        ctx->rscale_build = rscale_build;
    }
    return SPHX_OK;
}
extern "C" int sphx_get_constants(const sphx_ctx* ctx, sphx_constants* c) {
    if (!ctx || !c) return SPHX_E_ARG;
    *c = ctx->cst;
    return SPHX_OK;
}

// ---- staging helpers ------------------------------------------------------------------------
static int upload(sphx_ctx* ctx, DevBuf& b, const void* host, size_t bytes) {
    SPHX_TRY(sphx_ensure(ctx, b, bytes));
    HIPCHK(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SPHX_OK;
}
static int download(sphx_ctx* ctx, void* host, const void* dev, size_t bytes) {
    if (!host) return SPHX_OK;
    HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return SPHX_OK;
}
#define NEED(p)                                                                              \
    do {                                                                                     \
        if (!(p)) return sphx_set_err(ctx, SPHX_E_ARG, "%s: argument %s is NULL", __func__, #p); \
    } while (0)

// =============================================================================================
// nsc.neighbors                                                                nsc:541-552
// =============================================================================================
extern "C" int sphx_neighbors(sphx_ctx* ctx, int64_t n, int k, const double* points, double dist,
                              double eps, int64_t* idx, double* dist_out, int64_t* nontriv,
                              double* h) {
    (void)eps;
    if (!ctx) return SPHX_E_ARG;
    NEED(points);
    if (n < 1) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld < 1", (long long)n);
    if (k < 1 || k > SPHX_MAX_K) return sphx_set_err(ctx, SPHX_E_ARG, "N_NEIGH=%d not in 1..%d", k, SPHX_MAX_K);
    HIPCHK(hipSetDevice(ctx->device));
    ctx->map_perm = nullptr;
    ctx->qorder = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    SPHX_TRY(upload(ctx, ctx->in_a, points, 3 * nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_b, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_c, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_d, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_e, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_f, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->in_g, nb));
    double *x = ctx->in_b.as<double>(), *y = ctx->in_c.as<double>(), *z = ctx->in_d.as<double>();
    double *xs = ctx->in_e.as<double>(), *ys = ctx->in_f.as<double>(), *zs = ctx->in_g.as<double>();
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, ctx->in_a.as<double>(), x, y, z));
    ctx->clip_valid = false;                 // a fresh point set: no history for the robust box
    SPHX_TRY(sphx_build_grid(ctx, n, k, x, y, z, 0.0));
    SPHX_TRY(sphx_gather3(ctx, n, ctx->perm.as<int>(), x, y, z, xs, ys, zs));
    SPHX_TRY(sphx_ensure(ctx, ctx->idx64, (size_t)n * k * sizeof(int64_t)));
    SPHX_TRY(sphx_ensure(ctx, ctx->dist_out, (size_t)n * k * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->nontriv, (size_t)n * sizeof(int64_t)));
    SPHX_TRY(sphx_ensure(ctx, ctx->h_api, nb));
    KnnOut o;
    o.nbr = nullptr; o.h_sorted = nullptr;
    o.idx64 = ctx->idx64.as<int64_t>(); o.dist = ctx->dist_out.as<double>();
    o.nontriv = ctx->nontriv.as<int64_t>(); o.h_by_id = ctx->h_api.as<double>();
    // sorted -> original index is the permutation itself
    SPHX_TRY(sphx_knn(ctx, n, k, xs, ys, zs, ctx->perm.as<int>(), nullptr, nullptr, 1.0, dist, o));
    SPHX_TRY(download(ctx, idx, o.idx64, (size_t)n * k * sizeof(int64_t)));
    SPHX_TRY(download(ctx, dist_out, o.dist, (size_t)n * k * sizeof(double)));
    SPHX_TRY(download(ctx, nontriv, o.nontriv, (size_t)n * sizeof(int64_t)));
    SPHX_TRY(download(ctx, h, o.h_by_id, nb));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

// =============================================================================================
// nsc.hydro_update                                                             nsc:556-671
// =============================================================================================
extern "C" int sphx_hydro_update(sphx_ctx* ctx, int64_t n, int k, int s, const int64_t* neighbor,
                                 const double* points, const double* mass, const double* sizes,
                                 const double* f_un, const double* particle_type, const double* T,
                                 const double* mu_array, const double* gamma_array,
                                 const double* velocities, int visc_mode, double* hydro_accel,
                                 double* visc_accel, double* visc_heat, double* rho, double* nden,
                                 double* f_un_nb, double* rho_dust) {
    if (!ctx) return SPHX_E_ARG;
    NEED(points); NEED(mass); NEED(sizes); NEED(particle_type); NEED(T);
    NEED(mu_array); NEED(gamma_array); NEED(velocities);
    if (n < 1 || n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    if (k < 1 || k > 4096) return sphx_set_err(ctx, SPHX_E_ARG, "k=%d out of range", k);
    if (visc_mode != 0) return sphx_set_err(ctx, SPHX_E_ARG, "visc_mode %d unknown (0 = ref_axis0)", visc_mode);
    if (f_un_nb && (!f_un || s < 1 || s > SPHX_MAX_SPECIES))
        return sphx_set_err(ctx, SPHX_E_ARG, "species output needs f_un and 1 <= s <= %d", SPHX_MAX_SPECIES);
    HIPCHK(hipSetDevice(ctx->device));
    ctx->map_perm = nullptr;
    ctx->qorder = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    if (!neighbor && !(ctx->nbr_api_valid && ctx->nbr_api_n == n && ctx->nbr_api_k == k))
        return sphx_set_err(ctx, SPHX_E_STATE, "neighbor == NULL but no (%lld, %d) list is held from a previous call",
                            (long long)n, k);
    if (neighbor) SPHX_TRY(upload(ctx, ctx->idx64, neighbor, (size_t)n * k * sizeof(int64_t)));
    SPHX_TRY(upload(ctx, ctx->in_a, points, 3 * nb));
    SPHX_TRY(upload(ctx, ctx->in_b, velocities, 3 * nb));
    SPHX_TRY(upload(ctx, ctx->in_c, mass, nb));
    SPHX_TRY(upload(ctx, ctx->in_d, sizes, nb));
    SPHX_TRY(upload(ctx, ctx->in_e, T, nb));
    SPHX_TRY(upload(ctx, ctx->in_f, mu_array, nb));
    SPHX_TRY(upload(ctx, ctx->in_g, gamma_array, nb));
    SPHX_TRY(upload(ctx, ctx->in_h, particle_type, nb));
    if (neighbor) {
        SPHX_TRY(sphx_transpose_nbr(ctx, n, k, ctx->idx64.as<int64_t>()));
        ctx->nbr_api_valid = true; ctx->nbr_api_n = n; ctx->nbr_api_k = k;
    }
    SPHX_TRY(sphx_prep(ctx, n, nullptr, nullptr, nullptr, ctx->in_a.as<double>(), nullptr, nullptr,
                       nullptr, ctx->in_b.as<double>(), ctx->in_c.as<double>(), ctx->in_d.as<double>(),
                       ctx->in_e.as<double>(), ctx->in_f.as<double>(), ctx->in_g.as<double>(),
                       ctx->in_h.as<double>()));
    SPHX_TRY(sphx_pass_density(ctx, n, k));
    SPHX_TRY(sphx_pass_pi(ctx, n, k, ctx->in_d.as<double>(), ctx->in_h.as<double>()));
    SPHX_TRY(sphx_pass_visc(ctx, n, k, ctx->in_c.as<double>()));
    if (f_un_nb) {
        SPHX_TRY(upload(ctx, ctx->in_i, f_un, (size_t)n * s * sizeof(double)));
        SPHX_TRY(sphx_ensure(ctx, ctx->F, (size_t)n * s * sizeof(double)));
        SPHX_TRY(sphx_pass_species(ctx, n, k, s, ctx->in_i.as<double>(), ctx->F.as<double>()));
        SPHX_TRY(download(ctx, f_un_nb, ctx->F.p, (size_t)n * s * sizeof(double)));
    }
    SPHX_TRY(download(ctx, hydro_accel, ctx->ha.p, 3 * nb));
    SPHX_TRY(download(ctx, visc_accel, ctx->va.p, 3 * nb));
    SPHX_TRY(download(ctx, visc_heat, ctx->vh.p, nb));
    SPHX_TRY(download(ctx, rho, ctx->rho.p, nb));
    SPHX_TRY(download(ctx, nden, ctx->nden.p, nb));
    SPHX_TRY(download(ctx, rho_dust, ctx->rhod.p, nb));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}

// =============================================================================================
// device-resident state + step                                       drv:217-238, 437-491
// =============================================================================================
extern "C" int sphx_state_upload(sphx_ctx* ctx, int64_t n, int s, const double* pos, const double* vel,
                                 const double* mass, const double* ptype, const double* f_un,
                                 const double* T, const double* mu, const double* gamma,
                                 const double* E_internal, const double* accel_old) {
    if (!ctx) return SPHX_E_ARG;
    NEED(pos); NEED(vel); NEED(mass); NEED(ptype); NEED(T); NEED(mu); NEED(gamma); NEED(E_internal);
    if (n < 1 || n > 0x7FFFFFF0ll) return sphx_set_err(ctx, SPHX_E_ARG, "n=%lld out of range", (long long)n);
    if (f_un && (s < 1 || s > SPHX_MAX_SPECIES)) return sphx_set_err(ctx, SPHX_E_ARG, "s=%d out of range", s);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb = (size_t)n * sizeof(double);
    StateArrays& st = ctx->st;
    DevBuf* scal[] = {&st.x, &st.y, &st.z, &st.vx, &st.vy, &st.vz, &st.ax, &st.ay, &st.az,
                      &st.m, &st.T, &st.mu, &st.gam, &st.E, &st.hprev, &st.ptype};
    for (DevBuf* b : scal) SPHX_TRY(sphx_ensure(ctx, *b, nb));
    SPHX_TRY(sphx_ensure(ctx, st.id, (size_t)n * sizeof(int)));
    SPHX_TRY(upload(ctx, ctx->in_a, pos, 3 * nb));
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, ctx->in_a.as<double>(), st.x.as<double>(), st.y.as<double>(), st.z.as<double>()));
    SPHX_TRY(upload(ctx, ctx->in_b, vel, 3 * nb));
    SPHX_TRY(sphx_aos_to_soa3(ctx, n, ctx->in_b.as<double>(), st.vx.as<double>(), st.vy.as<double>(), st.vz.as<double>()));
    if (accel_old) {
        SPHX_TRY(upload(ctx, ctx->in_c, accel_old, 3 * nb));
        SPHX_TRY(sphx_aos_to_soa3(ctx, n, ctx->in_c.as<double>(), st.ax.as<double>(), st.ay.as<double>(), st.az.as<double>()));
    } else {
        HIPCHK(hipMemsetAsync(st.ax.p, 0, nb, ctx->stream));
        HIPCHK(hipMemsetAsync(st.ay.p, 0, nb, ctx->stream));
        HIPCHK(hipMemsetAsync(st.az.p, 0, nb, ctx->stream));
    }
    HIPCHK(hipMemcpyAsync(st.m.p, mass, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(st.T.p, T, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(st.mu.p, mu, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(st.gam.p, gamma, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(st.E.p, E_internal, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(st.ptype.p, ptype, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(st.hprev.p, 0, nb, ctx->stream));
    SPHX_TRY(sphx_iota(ctx, n, st.id.as<int>()));
    ctx->s = 0;
    if (f_un) {
        // rows padded to whole 128-B lines (16 doubles for the reference's 15 species): one aligned line per neighbour
        // in the species pass; kept in upload order (ctx->fun_id): the pass reaches a row through the particle's id
        const int sp = (s + 15) & ~15;
        SPHX_TRY(upload(ctx, ctx->in_d, f_un, (size_t)n * s * sizeof(double)));
        SPHX_TRY(sphx_ensure(ctx, ctx->fun_id, (size_t)n * sp * sizeof(double)));
        HIPCHK(hipMemsetAsync(ctx->fun_id.p, 0, (size_t)n * sp * sizeof(double), ctx->stream));
        HIPCHK(hipMemcpy2DAsync(ctx->fun_id.p, (size_t)sp * sizeof(double), ctx->in_d.p, (size_t)s * sizeof(double),
                                (size_t)s * sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->s = s;
        ctx->sp = sp;
    }
    HIPCHK(hipMemsetAsync(ctx->scal.p, 0, SC_NSLOTS * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->badc.p, 0, (size_t)BADC_BUCKETS * BADC_STRIDE * sizeof(u64), ctx->stream));
    ctx->ct_primed = false;       // SC_CT_BITS was just zeroed: the next pass 2 must prime it again
    ctx->nbr_api_valid = false;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->n = n;
    ctx->npad = sphx_pad64(n);
    ctx->has_state = true;
    ctx->drag = false;
    ctx->agb_on = false;
    ctx->gravity = 0;
    ctx->loop_forms = 0;
    ctx->list_valid = false;
    ctx->clip_valid = false;
    ctx->h_clip = 0.0;
    ctx->lag_bvalid[0] = ctx->lag_bvalid[1] = ctx->lag_hvalid[0] = ctx->lag_hvalid[1] = false;
    ctx->step_count = 0;
    ctx->tprev_valid = false;
    sphx_graph_drop(ctx);
    ctx->dt_last = 0.0;
    return SPHX_OK;
}

extern "C" int sphx_state_set_drag(sphx_ctx* ctx, const double* mean_grain_mass, const double* mean_cross) {
    if (!ctx) return SPHX_E_ARG;
    if (!ctx->has_state || ctx->step_count != 0)
        return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_set_drag must follow sphx_state_upload directly");
    HIPCHK(hipSetDevice(ctx->device));
    if (!mean_grain_mass || !mean_cross) { ctx->drag = false; return SPHX_OK; }
    const size_t nb = (size_t)ctx->n * sizeof(double);
    SPHX_TRY(sphx_ensure(ctx, ctx->st.mgm, nb));
    SPHX_TRY(sphx_ensure(ctx, ctx->st.mcs, nb));
    HIPCHK(hipMemcpyAsync(ctx->st.mgm.p, mean_grain_mass, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->st.mcs.p, mean_cross, nb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->drag = true;
    return SPHX_OK;
}

extern "C" int sphx_state_set_loop_forms(sphx_ctx* ctx, int on, double d) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    if (!ctx->has_state) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_set_loop_forms before sphx_state_upload");
    if (on && !(d > 0.0)) return sphx_set_err(ctx, SPHX_E_ARG, "the loop forms need the driver's global d > 0 (drv:68)");
    if (on && ctx->use_verlet) return sphx_set_err(ctx, SPHX_E_STATE, "loop-form steps are not combined with incremental search");
    ctx->loop_forms = on ? 1 : 0;
    ctx->loop_d = d;
    return SPHX_OK;
}

extern "C" int sphx_set_gravity_order(sphx_ctx* ctx, int order) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    if (order != 1 && order != 2) return sphx_set_err(ctx, SPHX_E_ARG, "gravity order %d not 1 (monopole) or 2 (quadrupole)", order);
    ctx->grav_order = order;
    return SPHX_OK;
}

extern "C" int sphx_set_clip_grad(sphx_ctx* ctx, int on) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    ctx->clip_grad = on ? 1 : 0;
    return SPHX_OK;
}

extern "C" int sphx_state_set_gravity(sphx_ctx* ctx, int mode, double G) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    if (!ctx->has_state) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_set_gravity before sphx_state_upload");
    if (mode < 0 || mode > 2) return sphx_set_err(ctx, SPHX_E_ARG, "gravity mode %d not in {0, 1, 2}", mode);
    if (mode == 2 && ctx->use_verlet)
        return sphx_set_err(ctx, SPHX_E_STATE, "tree gravity needs the cell grid of every step: not with incremental search");
    ctx->gravity = mode;
    ctx->grav_G = G;
    return SPHX_OK;
}

// call_first / call_last: the first / last step of this sphx_step call (they carry the call's start and end events)
static int one_step(sphx_ctx* ctx, int k, double dist, int first, double fixed_dt, bool call_first, bool call_last) {
    const int64_t n = ctx->n;
    const int ring = (int)(ctx->step_count % 3);
    hipEvent_t* ev = ctx->evring[ring];
    // Timing events: around the step and around the search always (the bench's roofline needs the search's launch time);
    // one per pass only on request (sphx_set_timing_detail / SPHX_TIMING_DETAIL=1) - an event record between two
    // dependent kernels costs the stream ~10 us, six of them 4 % of a 1.5 ms step.
    const bool cap = ctx->capturing;         // recorded into a step graph, not run: nothing here may wait, read back or time
    const bool detail = ctx->timing_detail && !cap;
    ctx->ev_detail[ring] = detail;
    // (the step's own start and end events likewise: without them the call's first and last step bracket the call,
    //  whose time is then shared out over its steps)
    const bool rec0 = (detail || call_first) && !cap, rec7 = (detail || call_last) && !cap;
    ctx->ev_has07[ring] = (rec0 ? 1 : 0) | (rec7 ? 2 : 0);
    if (!cap) ctx->replay_ok = false;        // (set at the end of a real step that a replay may freeze)
    ctx->map_perm = nullptr;
    ctx->qorder = nullptr;
    ctx->blob_lists = false;
    ctx->blob_split_valid = false;
    ctx->pass_part = 0;
    SPHX_TRY(sphx_blob_join(ctx));   // (a context that also serves the device API: its side-stream work ends first)
    ctx->nbr_api_valid = false;   // the step overwrites the K-major list (search or Verlet refresh)
    if (rec0) HIPCHK(hipEventRecord(ev[0], ctx->stream));
    // drv:233-238: applied by the grid build's first pass over the particles (sphx_grid.hip); the Verlet path looks at
    // the positions before any grid is built, so it clamps here
    ctx->clamp_vx = nullptr;
    if (ctx->use_verlet) {
        SPHX_TRY(sphx_clamp(ctx, n, ctx->st));
    } else {
        ctx->clamp_vx = ctx->st.vx.as<double>(); ctx->clamp_vy = ctx->st.vy.as<double>(); ctx->clamp_vz = ctx->st.vz.as<double>();
    }
    SPHX_TRY(sphx_ensure(ctx, ctx->nbr, (size_t)k * ctx->npad * sizeof(int)));
    // ---- incremental exact kNN from the Verlet lists (sphx_refresh.hip) ---------------------
    bool searched = false;
    if (ctx->use_verlet && ctx->list_valid && ctx->list_n == n && ctx->list_k == k) {
        StateArrays& r = ctx->st;
        if (cap) return sphx_set_err(ctx, SPHX_E_STATE, "a Verlet step cannot be captured");
        HIPCHK(hipEventRecord(ev[1], ctx->stream));
        int64_t nfail = 0;
        SPHX_TRY(sphx_knn_refresh(ctx, n, k, r.x.as<double>(), r.y.as<double>(), r.z.as<double>(),
                                  ctx->nbr.as<int>(), r.hprev.as<double>(), &nfail));
        if (nfail == 0) {
            searched = true;
            ctx->stats.refresh_steps++;
        } else {
            ctx->list_valid = false;         // some result could not be proven exact: rebuild
        }
    }
    if (!searched) {
        // cell size from the previous step's mean h (read back together with the bounding box)
        double cell_hint = 0.0;
        if (cap) {                            // the real step's figures, frozen
            cell_hint = ctx->cap_cell_hint;
            ctx->h_clip = ctx->cap_h_clip;
        } else if (ctx->step_count > 0) {
            const double* hs;                                                  // [0] sum ... [3] count
            if (ctx->lag_hvalid[ctx->lag_hslot]) {
                // copied out right after the previous step's search: no wait on that step's tail
                HIPCHK(hipEventSynchronize(ctx->lag_halias[ctx->lag_hslot] ? ctx->lag_halias[ctx->lag_hslot]
                                                                             : ctx->lag_hev[ctx->lag_hslot]));
                hs = (const double*)((char*)ctx->pinned + LAG_OFF + 512 * ctx->lag_hslot + 256);
            } else {
                HIPCHK(hipMemcpyAsync((char*)ctx->pinned + 256, ctx->scal.as<double>() + SC_HSUM, 4 * sizeof(double),
                                      hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                hs = (const double*)((char*)ctx->pinned + 256);
            }
            if (ctx->lag_hvalid[ctx->lag_hslot]) {       // the same copy carries the previous search's counters
                const u64* sv = (const u64*)hs;           // slots SC_HSUM ..: [5] SC_NFAILQ [6] SC_SHORT [7] SC_FARQ [8] SC_BADHINT
                for (int q = 0; q < 10; ++q) ctx->knn_lag[q] = sv[(SC_NFAILQ - SC_HSUM) + q];      // .. SC_KGDBG + 2
                ctx->knn_lag_valid = true;
            }
            const double hmean = hs[3] > 0.0 ? hs[0] / hs[3] : 0.0;
            if (hmean > 0.0 && isfinite(hmean)) {
                // SPHX_CELL_FEEDBACK: when a sizeable share of the particles lives in cells of >= DENSE_CELL members (a core
                // much denser than the mean radius suggests: the 27 cells around a group overflow its tile and the search hands
                // the group's queries on), the cells shrink, 3 % a step; they relax again, 1 % a step, once that share is small.
                // (A property of the positions alone: every variant of the search sees the same grid.)
                cell_hint = ctx->cell_factor * sphx_cell_feedback(ctx, n) * hmean;
                ctx->h_clip = ctx->h_clip_factor * hmean;
            }
            ctx->cap_cell_hint = cell_hint;
            ctx->cap_h_clip = ctx->h_clip;
        }
        {
            StateArrays& r = ctx->st;       // the box statistics are the previous step's when there are any
            const bool blob = ctx->use_blob && !ctx->use_verlet;
            ctx->lag_on = true;
            ctx->in_fused_step = true;
            ctx->step_ev1 = cap ? nullptr : ev[1];            // (recorded below, behind the state's permutation)
            ctx->defer_cell_sort = blob;      // the blob-order pass over the cells sorts their members too
            const int rc_ = sphx_build_grid(ctx, n, k, r.x.as<double>(), r.y.as<double>(), r.z.as<double>(), cell_hint);
            ctx->lag_on = false;
            ctx->in_fused_step = false;
            ctx->step_ev1 = nullptr;
            ctx->defer_cell_sort = false;
            ctx->clamp_vx = nullptr;
            SPHX_TRY(rc_);
            // (the blob order needs cell_of / perm / cell_start only: before the state is permuted, so that the
            //  deferred member sort has run when perm is used)
            if (blob) {
                ctx->defer_blob_scatter = true;          // (its last scatter: in sphx_permute_state's kernel, right below)
                const int rc_b = sphx_build_blob_order(ctx, n);
                ctx->defer_blob_scatter = false;
                SPHX_TRY(rc_b);
            }
        }
        const bool split_perm = ctx->split_perm && ctx->side_stream && !ctx->use_verlet;
        SPHX_TRY(sphx_permute_state(ctx, n, split_perm, cap ? nullptr : ev[1]));      // (records ev[1] behind the search's part)
        StateArrays& r = ctx->st;
        KnnOut o;
        o.nbr = ctx->nbr.as<int>();
        o.h_sorted = r.hprev.as<double>();   // read as the search-radius hint, then overwritten
        o.idx64 = nullptr; o.dist = nullptr; o.nontriv = nullptr; o.h_by_id = nullptr;
        double rs = ctx->rscale;
        if (ctx->use_verlet) {
            SPHX_TRY(sphx_ensure(ctx, ctx->list64, (size_t)n * 64 * sizeof(int)));
            SPHX_TRY(sphx_ensure(ctx, ctx->dref, (size_t)n * sizeof(double)));
            o.list64 = ctx->list64.as<int>();
            o.dref = ctx->dref.as<double>();
            rs = ctx->rscale_build;          // wider: the list must hold 64 entries to earn its margin
        }
        ctx->knn_hinted = ctx->step_count > 0 && !ctx->use_verlet;     // hprev holds the previous step's radii
        ctx->knn_lag_external = true;
        const int rc_knn = sphx_knn(ctx, n, k, r.x.as<double>(), r.y.as<double>(), r.z.as<double>(), r.id.as<int>(),
                                    ctx->inv.as<int>(), r.hprev.as<double>(), rs, dist, o);
        ctx->knn_hinted = false;
        ctx->knn_lag_external = false;
        SPHX_TRY(rc_knn);
        if (split_perm) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_perm, 0));    // the rest of the state is in place
        if (ctx->use_verlet) {
            SPHX_TRY(sphx_save_list_positions(ctx, n, r.x.as<double>(), r.y.as<double>(), r.z.as<double>()));
            ctx->list_valid = true;
            ctx->list_n = n;
            ctx->list_k = k;
        }
        ctx->stats.rebuild_steps++;
    }
    StateArrays& s = ctx->st;
    hipEvent_t ev_search_end = cap ? ctx->ev_fork : ev[2];       // (a timing event is not recorded into a graph)
    HIPCHK(hipEventRecord(ev_search_end, ctx->stream));
    // the record build (bandwidth-bound) does not depend on the list dedup (latency-bound): side by side - and with
    // it the sum of h and its copy to the host (the next step's cell size, read there when the next grid is sized;
    // the search's counters travel in the same copy: SC_HSUM .. SC_BADHINT are consecutive slots)
    const bool fork = ctx->qorder && ctx->use_lds && !ctx->loop_forms && ctx->side_stream;
    hipStream_t hs_stream = ctx->stream;
    if (fork) {
        HIPCHK(hipStreamWaitEvent(ctx->side_stream, ev_search_end, 0));       // (the search's end event doubles as the fork)
        hs_stream = ctx->side_stream;
    }
    // (forked: BEHIND the record build on the side stream - the passes wait for that one, not for this; nobody but the
    //  next step's host code reads the sums)
    auto h_sums_out = [&]() -> int {
        hipStream_t main_stream = ctx->stream;
        ctx->stream = hs_stream;
        const int rc_h = sphx_hsum(ctx, n, s.hprev.as<double>());
        ctx->stream = main_stream;
        SPHX_TRY(rc_h);
        if (cap) return SPHX_OK;              // (the sums stay on the device: the next REAL step reads them, with a wait)
        const int hsl = ctx->lag_hslot ^ 1;
        HIPCHK(hipMemcpyAsync((char*)ctx->pinned + LAG_OFF + 512 * hsl + 256, ctx->scal.as<double>() + SC_HSUM,
                              (SC_KGDBG + 2 - SC_HSUM + 1) * sizeof(double), hipMemcpyDeviceToHost, hs_stream));
        HIPCHK(hipEventRecord(ctx->lag_hev[hsl], hs_stream));
        ctx->lag_halias[hsl] = nullptr;
        ctx->lag_hvalid[hsl] = true;
        ctx->lag_hslot = hsl;
        return SPHX_OK;
    };
    if (!fork) SPHX_TRY(h_sums_out());
    if (ctx->qorder && ctx->use_lds) SPHX_TRY(sphx_blob_translate(ctx, n, k));
    if (ctx->loop_forms) {
        // the reference's time loop (drv:451-458): loop forms on this step's neighbour list
        const bool species = ctx->s > 0 && ctx->fun_id.p;      // the species pass reads hydro_update's records (RecA)
        if (ctx->drag || species)
            SPHX_TRY(sphx_prep(ctx, n, s.x.as<double>(), s.y.as<double>(), s.z.as<double>(), nullptr,
                               s.vx.as<double>(), s.vy.as<double>(), s.vz.as<double>(), nullptr, s.m.as<double>(),
                               s.hprev.as<double>(), s.T.as<double>(), s.mu.as<double>(), s.gam.as<double>(),
                               s.ptype.as<double>()));
        if (detail) HIPCHK(hipEventRecord(ev[3], ctx->stream));
        SPHX_TRY(sphx_loop_step_sums(ctx, n, k, ctx->loop_d));
        if (detail) HIPCHK(hipEventRecord(ev[9], ctx->stream));
        if (species) SPHX_TRY(sphx_step_species(ctx, n, k));       // nsc:624-627 (+ metallicity, AGB yields)
        if (detail) HIPCHK(hipEventRecord(ev[4], ctx->stream));
        if (detail) HIPCHK(hipEventRecord(ev[5], ctx->stream));
        if (ctx->drag)
            SPHX_TRY(sphx_pass_drag(ctx, n, k, s.m.as<double>(), s.ptype.as<double>(), s.mgm.as<double>(),
                                    s.mcs.as<double>()));
    } else {
    {
        hipStream_t main_stream = ctx->stream;
        if (fork) ctx->stream = ctx->side_stream;          // (it is already behind the search: the h sums went there first)
        const int rc_prep = sphx_prep(ctx, n, s.x.as<double>(), s.y.as<double>(), s.z.as<double>(), nullptr,
                                      s.vx.as<double>(), s.vy.as<double>(), s.vz.as<double>(), nullptr,
                                      s.m.as<double>(), s.hprev.as<double>(), s.T.as<double>(), s.mu.as<double>(),
                                      s.gam.as<double>(), s.ptype.as<double>());
        ctx->stream = main_stream;
        if (rc_prep != SPHX_OK) return rc_prep;
        if (fork) {
            if (cap) SPHX_TRY(h_sums_out());          // (inside the fork: a captured side stream must rejoin with nothing behind)
            HIPCHK(hipEventRecord(ctx->ev_join, ctx->side_stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            if (!cap) SPHX_TRY(h_sums_out());
        }
    }
    if (detail) HIPCHK(hipEventRecord(ev[3], ctx->stream));
    ctx->lean_outputs = true;          // (nothing in the step reads G: hydro_accel = G / rho is what the update takes)
    // nsc:624-627 on the step's list (+ metallicity and AGB yields when a table is set): when the state carries f_un -
    // inside pass 1's kernel where both run out of LDS (their first sweeps are the same), else behind it
    const bool species = ctx->s > 0 && ctx->fun_id.p;
    const bool sp_fused = species && ctx->species_fused && ctx->species_lds && ctx->use_lds && ctx->qorder && ctx->blob_lists &&
                          !ctx->map_perm && ctx->sp == 16 && ctx->s <= 16 && k <= SPHX_MAX_K;
    int rc_dens;
    if (sp_fused) {
        const int S = ctx->s;
        rc_dens = sphx_ensure(ctx, ctx->rho, (size_t)n * sizeof(double));
        if (rc_dens == SPHX_OK) rc_dens = sphx_ensure(ctx, ctx->rhod, (size_t)n * sizeof(double));
        if (rc_dens == SPHX_OK) rc_dens = sphx_ensure(ctx, ctx->nden, (size_t)n * sizeof(double));
        if (rc_dens == SPHX_OK) rc_dens = sphx_ensure(ctx, ctx->G, (size_t)n * 3 * sizeof(double));
        if (rc_dens == SPHX_OK) rc_dens = sphx_ensure(ctx, ctx->ha, (size_t)n * 3 * sizeof(double));
        if (rc_dens == SPHX_OK) rc_dens = sphx_ensure(ctx, ctx->F, (size_t)n * S * sizeof(double));
        if (rc_dens == SPHX_OK && ctx->agb_on) {
            rc_dens = sphx_ensure(ctx, ctx->Zmet, (size_t)n * sizeof(double));
            if (rc_dens == SPHX_OK) rc_dens = sphx_ensure(ctx, ctx->agb_dust, (size_t)n * S * sizeof(double));
        }
        if (rc_dens == SPHX_OK)
            rc_dens = sphx_blob_density_species(ctx, n, k, S, ctx->fun_id.as<double>(), ctx->st.id.as<int>(), ctx->st.m.as<double>(),
                                                ctx->F.as<double>(), ctx->Zmet.as<double>(), ctx->agb_dust.as<double>(),
                                                (ctx->agb_on && ctx->Zmet.p && ctx->agb_dust.p) ? 1 : 0);
    } else {
        rc_dens = sphx_pass_density(ctx, n, k);
    }
    ctx->lean_outputs = false;
    SPHX_TRY(rc_dens);
    if (detail) HIPCHK(hipEventRecord(ev[9], ctx->stream));
    if (species && !sp_fused) SPHX_TRY(sphx_step_species(ctx, n, k));
    if (detail) HIPCHK(hipEventRecord(ev[4], ctx->stream));
    SPHX_TRY(sphx_pass_pi(ctx, n, k, s.hprev.as<double>(), s.ptype.as<double>()));
    if (detail) HIPCHK(hipEventRecord(ev[5], ctx->stream));
    SPHX_TRY(sphx_pass_visc(ctx, n, k, s.m.as<double>()));
    if (ctx->drag)
        SPHX_TRY(sphx_pass_drag(ctx, n, k, s.m.as<double>(), s.ptype.as<double>(), s.mgm.as<double>(),
                                s.mcs.as<double>()));
    }
    if (detail) HIPCHK(hipEventRecord(ev[6], ctx->stream));
    if (ctx->gravity) {                          // drv:448-449; softening = median(h), nsc:358
        SPHX_TRY(sphx_ensure(ctx, ctx->grav, (size_t)n * 3 * sizeof(double)));
        double* eps = ctx->scal.as<double>() + SC_GRAV_EPS;
        SPHX_TRY(sphx_median(ctx, n, s.hprev.as<double>(), eps));
        if (ctx->gravity == 1)
            SPHX_TRY(sphx_gravity_launch(ctx, n, s.x.as<double>(), s.y.as<double>(), s.z.as<double>(), 1,
                                         s.m.as<double>(), eps, 0.0, ctx->grav_G, nullptr, ctx->grav.as<double>()));
        else
            SPHX_TRY(sphx_gravity_tree_launch(ctx, n, s.x.as<double>(), s.y.as<double>(), s.z.as<double>(),
                                              s.m.as<double>(), ctx->grav_ws, eps, 0.0, ctx->grav_G, nullptr,
                                              ctx->grav.as<double>()));
    }
    if (detail) HIPCHK(hipEventRecord(ev[8], ctx->stream));
    // (dt by drv:222-229 inside the update kernel; the crossing-time vote is reset by the next step's first kernel, or
    //  primed by its pass 2 when that grid build is not the fused one)
    SPHX_TRY(sphx_integrate(ctx, n, 1, first, fixed_dt));
    if (rec7) HIPCHK(hipEventRecord(ev[7], ctx->stream));
    ctx->step_count++;
    // May the next steps be replays of this one's decisions?  A hinted step on the fused grid build, nothing that decides
    // on the host from this step's own results (Verlet refresh, outlier levels, hint distrust, the drag's scatter plan,
    // gravity's sort) and no per-pass timing.
    if (!cap)
        ctx->replay_ok = searched == false && !first && ctx->step_count >= 2 && ctx->grid_fused && !ctx->use_verlet && ctx->olev.L == 0 &&
                         !ctx->distrust && !ctx->drag && !ctx->gravity && !(ctx->s > 0 && ctx->fun_id.p) && !ctx->timing_detail && ctx->use_blob && ctx->use_group &&
                         ctx->side_stream != nullptr;
    return SPHX_OK;
}

static int collect_stats(sphx_ctx* ctx, int ring) {
    if (!(ctx->ev_pending & (1u << ring))) return SPHX_OK;
    ctx->ev_pending &= ~(1u << ring);
    hipEvent_t* ev = ctx->evring[ring];
    const int has = ctx->ev_has07[ring];
    HIPCHK(hipEventSynchronize((has & 2) ? ev[7] : ev[2]));
    sphx_stats& st = ctx->stats;
    float ms[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    HIPCHK(hipEventElapsedTime(&ms[1], ev[1], ev[2]));
    st.ms_search += ms[1];
    st.steps += 1;
    st.search_steps += 1;
    st.n = ctx->n;
    if (!ctx->ev_detail[ring]) return SPHX_OK;    // (only the search's events were recorded for this step; ms_total: per call)
    float tot;
    HIPCHK(hipEventElapsedTime(&tot, ev[0], ev[7]));
    HIPCHK(hipEventElapsedTime(&ms[0], ev[0], ev[1]));
    st.ms_grid += ms[0]; st.ms_total += tot;
    for (int i = 2; i < 6; ++i) HIPCHK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    st.ms_prep += ms[2];
    float mg = 0.f, mi = 0.f;
    HIPCHK(hipEventElapsedTime(&mg, ev[6], ev[8]));
    HIPCHK(hipEventElapsedTime(&mi, ev[8], ev[7]));
    ms[6] = mi;
    st.ms_gravity += mg;
    {                                             // (ev[9] is recorded between the sums before it and the species pass)
        float md = 0.f, msp = 0.f;
        HIPCHK(hipEventElapsedTime(&md, ev[3], ev[9]));
        HIPCHK(hipEventElapsedTime(&msp, ev[9], ev[4]));
        ms[3] = md;
        st.ms_species += msp;
    }
    st.ms_density += ms[3]; st.ms_pi += ms[4]; st.ms_visc += ms[5]; st.ms_integrate += ms[6];
    st.detail_steps += 1;
    return SPHX_OK;
}

// ---- step graphs (see sphx_internal.h) --------------------------------------------------------------------------
// (an executable graph may still be RUNNING when the host decides to forget it - the loop runs ahead of the GPU - so it
//  is only retired here and destroyed behind the next wait for the stream: sphx_graph_reap)
void sphx_graph_drop(sphx_ctx* ctx) {
    for (int p = 0; p < 2; ++p) {
        if (!ctx->gexec[p]) continue;
        if (ctx->n_retired == (int)(sizeof(ctx->retired) / sizeof(ctx->retired[0]))) {
            (void)hipStreamSynchronize(ctx->stream);
            sphx_graph_reap(ctx);
        }
        ctx->retired[ctx->n_retired++] = ctx->gexec[p];
        ctx->gexec[p] = nullptr;
    }
    ctx->graph_left = 0;
}
void sphx_graph_reap(sphx_ctx* ctx) {          // the stream has been waited for: nothing retired can still be running
    for (int q = 0; q < ctx->n_retired; ++q) (void)hipGraphExecDestroy(ctx->retired[q]);
    ctx->n_retired = 0;
}

// what a replayed step does to the HOST's view of the state: the permutation swaps the two sets of arrays, the update
// swaps the temperature buffers back (sphx_permute_state, sphx_integrate)
static void graph_host_side(sphx_ctx* ctx) {
    StateArrays t = ctx->st; ctx->st = ctx->alt; ctx->alt = t;
    DevBuf tt = ctx->st.T; ctx->st.T = ctx->alt.T; ctx->alt.T = tt;
    ctx->tprev_valid = true;
    ctx->step_count++;
    ctx->stats.rebuild_steps++;
    ctx->stats.graph_steps++;
    ctx->stats.steps++;
}

// Record one step into a graph (nothing runs), instantiate it for the current parity and launch it.  On any failure the
// host's view is put back, graphs are switched off for this context and the caller runs the step the ordinary way.
static int capture_and_launch(sphx_ctx* ctx, int k, double dist, double fixed_dt, bool* done) {
    *done = false;
    const int par = (int)(ctx->step_count & 1);
    const StateArrays st0 = ctx->st, alt0 = ctx->alt;
    const int64_t sc0 = ctx->step_count;
    const bool tp0 = ctx->tprev_valid;
    const sphx_stats stats0 = ctx->stats;
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        ctx->graph_mode = 0;
        return SPHX_OK;
    }
    ctx->capturing = true;
    const int rc = one_step(ctx, k, dist, 0, fixed_dt, false, false);
    ctx->capturing = false;
    const hipError_t e_end = hipStreamEndCapture(ctx->stream, &g);
    hipGraphExec_t ge = nullptr;
    hipError_t e_inst = hipErrorUnknown;
    if (rc == SPHX_OK && e_end == hipSuccess && g) e_inst = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (rc != SPHX_OK || e_end != hipSuccess || e_inst != hipSuccess || !ge) {
        (void)hipGetLastError();
        ctx->st = st0; ctx->alt = alt0; ctx->step_count = sc0; ctx->tprev_valid = tp0; ctx->stats = stats0;
        ctx->graph_mode = 0;                       // (this runtime / this state cannot be captured: the ordinary way from here on)
        sphx_graph_drop(ctx);
        return SPHX_OK;
    }
    ctx->gexec[par] = ge;
    ctx->stats.graph_steps++;
    ctx->stats.steps++;
    HIPCHK(hipGraphLaunch(ge, ctx->stream));
    *done = true;
    return SPHX_OK;
}

extern "C" int sphx_step(sphx_ctx* ctx, int nsteps, int k, double dist, int first, double fixed_dt) {
    if (!ctx) return SPHX_E_ARG;
    if (!ctx->has_state) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_step before sphx_state_upload");
    if (k < 1 || k > SPHX_MAX_K) return sphx_set_err(ctx, SPHX_E_ARG, "N_NEIGH=%d not in 1..%d", k, SPHX_MAX_K);
    if (nsteps < 1) return sphx_set_err(ctx, SPHX_E_ARG, "nsteps=%d < 1", nsteps);
    HIPCHK(hipSetDevice(ctx->device));
    if (!(dist > 0.0) || !isfinite(dist)) dist = 0.0;
    ctx->k = k;
    const bool per_call = !ctx->timing_detail;             // (else every step times itself)
    // the call's wall time on the stream, over all its steps (events of its own: a replayed step records none)
    if (per_call) HIPCHK(hipEventRecord(ctx->ev[8], ctx->stream));
    const bool graphs = (ctx->graph_mode == 1 || (ctx->graph_mode == 2 && ctx->n <= ctx->graph_max_n)) && ctx->stream == ctx->own_stream;
    if (ctx->gexec[0] || ctx->gexec[1])
        if (!graphs || k != ctx->graph_k || dist != ctx->graph_dist || fixed_dt != ctx->graph_fixed_dt) sphx_graph_drop(ctx);
    for (int it = 0; it < nsteps; ++it) {
        const int par = (int)(ctx->step_count & 1);
        if (graphs && !(first && it == 0) && ctx->graph_left > 0) {
            if (ctx->gexec[par]) {                                  // replay
                HIPCHK(hipGraphLaunch(ctx->gexec[par], ctx->stream));
                graph_host_side(ctx);
                if (--ctx->graph_left == 0) sphx_graph_drop(ctx);
                continue;
            }
            bool done = false;                                      // the epoch's other parity: record it now
            SPHX_TRY(capture_and_launch(ctx, k, dist, fixed_dt, &done));
            if (done) {
                if (--ctx->graph_left == 0) sphx_graph_drop(ctx);
                continue;
            }
        }
        // an ordinary step: sizes the grid and takes the search's decisions from the lagged read-backs
        sphx_graph_drop(ctx);
        const int ring = (int)(ctx->step_count % 3);
        SPHX_TRY(collect_stats(ctx, ring));            // (the step launched three steps ago, if still uncollected)
        SPHX_TRY(one_step(ctx, k, dist, first && it == 0, fixed_dt, false, false));
        ctx->ev_pending |= 1u << ring;
        SPHX_TRY(collect_stats(ctx, (ring + 1) % 3));  // two steps ago: finished long since, no wait
        if (graphs && ctx->replay_ok) {                 // the next graph_epoch steps replay this one's decisions
            ctx->graph_left = ctx->graph_epoch;
            ctx->graph_k = k; ctx->graph_dist = dist; ctx->graph_fixed_dt = fixed_dt;
        }
    }
    if (per_call) HIPCHK(hipEventRecord(ctx->ev[9], ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->pinned, ctx->scal.p, SC_NSLOTS * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    sphx_graph_reap(ctx);
    for (int r = 0; r < 3; ++r) SPHX_TRY(collect_stats(ctx, (int)((ctx->step_count + r) % 3)));   // oldest first
    if (per_call) {                                        // the call's wall time on the stream, over all its steps
        float tot = 0.f;
        HIPCHK(hipEventElapsedTime(&tot, ctx->ev[8], ctx->ev[9]));
        ctx->stats.ms_total += tot;
    }
    const u64* sc = (const u64*)ctx->pinned;
    ctx->dt_last = ((const double*)ctx->pinned)[SC_DT];
    ctx->stats.candidates = (int64_t)sc[SC_CAND];
    ctx->stats.retries = (int64_t)sc[SC_RETRY];
    ctx->stats.fallback_queries = (int64_t)(u32)sc[SC_NFAILQ];
    ctx->stats.short_rows = (int64_t)sc[SC_SHORT];
    SPHX_TRY(sphx_badc_read(ctx));
#ifdef SPHX_EXPERIMENTS
    if (ctx->knn_prof_print) {
        const char* nm[4] = {"in-box", "-", "outside the box", "-"};
        for (int c = 0; c < 4; c += 2)          // (the odd slots: cycles by section of the same two classes, below)
            if (sc[SC_KNNPROF + 4 + c])
                fprintf(stderr, "[sphx] general search, %-17s: %9llu queries, %8.0f cycles each, longest %10llu, %.2f tries each\n", nm[c],
                        sc[SC_KNNPROF + 4 + c], (double)sc[SC_KNNPROF + c] / (double)sc[SC_KNNPROF + 4 + c], sc[SC_KNNPROF + 8 + c],
                        (double)sc[SC_KNNPROF + 12 + c] / (double)sc[SC_KNNPROF + 4 + c]);
        for (int c = 0; c < 4; c += 2)
            if (sc[SC_KNNPROF + 4 + c])
                fprintf(stderr, "[sphx]   %-17s per query: %.1f batches of rows; cycles in row set-up %.0f, candidates %.0f, ranking %.0f\n", nm[c],
                        (double)sc[SC_KNNPROF + 13 + c] / (double)sc[SC_KNNPROF + 4 + c], (double)sc[SC_KNNPROF + 1 + c] / (double)sc[SC_KNNPROF + 4 + c],
                        (double)sc[SC_KNNPROF + 5 + c] / (double)sc[SC_KNNPROF + 4 + c], (double)sc[SC_KNNPROF + 9 + c] / (double)sc[SC_KNNPROF + 4 + c]);
        const double* dbg = (const double*)(sc + SC_KNNPROF + 16);
        if (dbg[7] > 0.0)
            fprintf(stderr, "[sphx]   a long one: at (%.4g %.4g %.4g), radius given %.4g, found h %.4g, %g retries, %g candidates, %g cycles | box origin (%.4g %.4g %.4g) cell %.4g dims %d %d %d\n",
                    dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], ctx->grid.xmin, ctx->grid.ymin, ctx->grid.zmin,
                    ctx->grid.cell, ctx->grid.nx, ctx->grid.ny, ctx->grid.nz);
        HIPCHK(hipMemsetAsync(ctx->scal.as<u64>() + SC_KNNPROF, 0, 24 * sizeof(u64), ctx->stream));
    }
    if (ctx->kg_debug_print)
        fprintf(stderr, "[sphx] grouped search, handed on (cumulative): no-hint %llu tile %llu tol %llu >64 %llu <K %llu near-tie %llu | groups over the row cap %llu, over the pre-cull cap %llu\n",
                sc[SC_KGDBG + 1], sc[SC_KGDBG + 2], sc[SC_KGDBG + 3], sc[SC_KGDBG + 4], sc[SC_KGDBG + 5], sc[SC_KGDBG + 6],
                sc[SC_KGDBG + 7], sc[SC_KGDBG + 0]);
#endif
    return SPHX_OK;
}

extern "C" int sphx_state_download(sphx_ctx* ctx, double* pos, double* vel, double* accel,
                                   double* E_internal, double* T, double* sizes, double* rho,
                                   double* nden, double* visc_heat, double* dt_last) {
    if (!ctx) return SPHX_E_ARG;
    if (!ctx->has_state) return sphx_set_err(ctx, SPHX_E_STATE, "no state uploaded");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = ctx->n;
    const size_t nb = (size_t)n * sizeof(double);
    StateArrays& s = ctx->st;
    const int* id = s.id.as<int>();
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, 3 * nb));
    double* stage = ctx->out_a.as<double>();
    struct V3 { double* host; DevBuf *a, *b, *c; };
    V3 v3[] = {{pos, &s.x, &s.y, &s.z}, {vel, &s.vx, &s.vy, &s.vz}, {accel, &s.ax, &s.ay, &s.az}};
    for (V3& v : v3) {
        if (!v.host) continue;
        SPHX_TRY(sphx_soa3_to_aos_by_id(ctx, n, id, v.a->as<double>(), v.b->as<double>(), v.c->as<double>(), stage));
        SPHX_TRY(download(ctx, v.host, stage, 3 * nb));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    struct V1 { double* host; const double* dev; };
    const bool stepped = ctx->step_count > 0;
    V1 v1[] = {{E_internal, s.E.as<double>()}, {T, s.T.as<double>()}, {sizes, s.hprev.as<double>()},
               {rho, stepped ? ctx->rho.as<double>() : nullptr},
               {nden, stepped ? ctx->nden.as<double>() : nullptr},
               {visc_heat, stepped ? ctx->vh.as<double>() : nullptr}};
    for (V1& v : v1) {
        if (!v.host) continue;
        if (!v.dev) { memset(v.host, 0, nb); continue; }
        SPHX_TRY(sphx_scatter_rows_by_id(ctx, n, 1, id, v.dev, stage));
        SPHX_TRY(download(ctx, v.host, stage, nb));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    if (dt_last) *dt_last = ctx->dt_last;
    return SPHX_OK;
}

// The AGB dust-yield table (config_helper.py:138-178: 11 degree-1 splines over (metallicity, mass)) into the context
int sphx_agb_table_set(sphx_ctx* ctx, int S, int nspl, const int32_t* ntx, const int32_t* nty, const double* tx, const double* ty,
                       const double* coeffs, const int32_t* mapto, double divisor, const double* mu_specie, double solar_mass) {
    ctx->agb_on = false;
    sphx_graph_drop(ctx);
    if (nspl == 0) return SPHX_OK;
    if (S < 7 || S > AGB_MAX_SPEC) return sphx_set_err(ctx, SPHX_E_ARG, "AGB table: %d species (7..%d)", S, AGB_MAX_SPEC);
    if (!ntx || !nty || !tx || !ty || !coeffs || !mapto || !mu_specie)
        return sphx_set_err(ctx, SPHX_E_ARG, "AGB table: NULL argument");
    if (nspl < 1 || nspl > AGB_MAX_SPL) return sphx_set_err(ctx, SPHX_E_ARG, "AGB table: %d splines", nspl);
    if (!(divisor != 0.0)) return sphx_set_err(ctx, SPHX_E_ARG, "AGB table: divisor is 0");
    HIPCHK(hipSetDevice(ctx->device));
    AgbTable& t = ctx->agb;
    t.nspl = nspl; t.nspec = S;
    size_t ntx_tot = 0, nty_tot = 0, nc_tot = 0;
    for (int o = 0; o < nspl; ++o) {
        if (ntx[o] < 4 || nty[o] < 4) return sphx_set_err(ctx, SPHX_E_ARG, "AGB table: spline %d has fewer than 4 knots", o);
        if (mapto[o] < 0 || mapto[o] >= S) return sphx_set_err(ctx, SPHX_E_ARG, "AGB table: mapto[%d]=%d out of range", o, mapto[o]);
        ntx_tot += ntx[o]; nty_tot += nty[o]; nc_tot += (size_t)(ntx[o] - 2) * (nty[o] - 2);
    }
    size_t ox = 0, oy = ntx_tot, oc = ntx_tot + nty_tot;
    for (int o = 0; o < nspl; ++o) {
        t.tx_off[o] = (int)ox; t.ty_off[o] = (int)oy; t.c_off[o] = (int)oc;
        t.ntx[o] = ntx[o]; t.nty[o] = nty[o]; t.mapto[o] = mapto[o];
        ox += ntx[o]; oy += nty[o]; oc += (size_t)(ntx[o] - 2) * (nty[o] - 2);
    }
    for (int s = 0; s < S; ++s) t.mu[s] = mu_specie[s];
    t.divisor = divisor; t.solar = solar_mass;
    SPHX_TRY(sphx_ensure(ctx, ctx->agb_knots, (ntx_tot + nty_tot + nc_tot + 6 * (size_t)nspl) * sizeof(double)));
    double* kd = ctx->agb_knots.as<double>();
    {   // the per-spline integers once more, in device memory (AgbTable::meta_off)
        double meta[6 * AGB_MAX_SPL];
        t.covered = 0u;
        for (int o = 0; o < nspl; ++o) {
            bool later = false;
            for (int o2 = o + 1; o2 < nspl; ++o2) later = later || mapto[o2] == mapto[o];
            meta[6 * o + 0] = t.tx_off[o]; meta[6 * o + 1] = t.ty_off[o]; meta[6 * o + 2] = t.c_off[o];
            meta[6 * o + 3] = t.ntx[o]; meta[6 * o + 4] = t.nty[o]; meta[6 * o + 5] = later ? -1.0 : (double)mapto[o];
            t.covered |= 1u << mapto[o];
        }
        t.meta_off = (int)(ntx_tot + nty_tot + nc_tot);
        HIPCHK(hipMemcpyAsync(kd + t.meta_off, meta, 6 * (size_t)nspl * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));       // (meta is a stack array)
    }
    HIPCHK(hipMemcpyAsync(kd, tx, ntx_tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(kd + ntx_tot, ty, nty_tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(kd + ntx_tot + nty_tot, coeffs, nc_tot * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    t.knots = kd;
    ctx->agb_on = true;
    return SPHX_OK;
}

// ... for the step's species pass: with it every step also leaves Z_i (drv:663 on the smoothed composition) and the
// yields of config_helper.py:183-189 at (Z_i, m_i).  nspl == 0 switches it off.  Call after sphx_state_upload with f_un.
extern "C" int sphx_state_set_agb(sphx_ctx* ctx, int nspl, const int32_t* ntx, const int32_t* nty, const double* tx,
                                  const double* ty, const double* coeffs, const int32_t* mapto, double divisor,
                                  const double* mu_specie, double solar_mass) {
    if (!ctx) return SPHX_E_ARG;
    if (!ctx->has_state) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_set_agb before sphx_state_upload");
    ctx->agb_on = false;
    if (nspl == 0) return SPHX_OK;
    if (ctx->s < 7 || !ctx->fun_id.p)
        return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_set_agb: the state carries no composition (f_un) of >= 7 species");
    return sphx_agb_table_set(ctx, ctx->s, nspl, ntx, nty, tx, ty, coeffs, mapto, divisor, mu_specie, solar_mass);
}

// F (S,n) species number densities of the last step (nsc:624-627), Z (n,), agb_dust (n,S) - caller order; any may be NULL
__global__ __launch_bounds__(256) void scatter_species_major(int n, int S, const int* id, const double* in, double* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const size_t o = (size_t)id[t];
    for (int s = 0; s < S; ++s) out[(size_t)s * n + o] = in[(size_t)s * n + t];
}
extern "C" int sphx_state_download_species(sphx_ctx* ctx, double* F, double* Z, double* agb_dust) {
    if (!ctx) return SPHX_E_ARG;
    if (!ctx->has_state || ctx->s < 1 || !ctx->fun_id.p || ctx->step_count < 1)
        return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_download_species: no species pass has run (state without f_un, "
                                                 "or no step yet)");
    if ((Z || agb_dust) && !ctx->agb_on) return sphx_set_err(ctx, SPHX_E_STATE, "sphx_state_download_species: no AGB table set");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = ctx->n;
    const int S = ctx->s;
    const int* id = ctx->st.id.as<int>();
    SPHX_TRY(sphx_ensure(ctx, ctx->out_a, (size_t)n * S * sizeof(double)));
    double* stage = ctx->out_a.as<double>();
    if (F) {
        hipLaunchKernelGGL(scatter_species_major, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, S, id,
                           ctx->F.as<double>(), stage);
        SPHX_TRY(download(ctx, F, stage, (size_t)n * S * sizeof(double)));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    if (Z) {
        SPHX_TRY(sphx_scatter_rows_by_id(ctx, n, 1, id, ctx->Zmet.as<double>(), stage));
        SPHX_TRY(download(ctx, Z, stage, (size_t)n * sizeof(double)));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    if (agb_dust) {
        SPHX_TRY(sphx_scatter_rows_by_id(ctx, n, S, id, ctx->agb_dust.as<double>(), stage));
        SPHX_TRY(download(ctx, agb_dust, stage, (size_t)n * S * sizeof(double)));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return SPHX_OK;
}

int sphx_badc_read(sphx_ctx* ctx) {
    u64* hb = (u64*)((char*)ctx->pinned + 8192);
    HIPCHK(hipMemcpyAsync(hb, ctx->badc.p, (size_t)BADC_BUCKETS * BADC_STRIDE * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u64 tot[4] = {0, 0, 0, 0};
    for (int b = 0; b < BADC_BUCKETS; ++b)
        for (int q = 0; q < 4; ++q) tot[q] += hb[b * BADC_STRIDE + q];
    ctx->stats.bad_accel = (int64_t)tot[BAD_ACCEL]; ctx->stats.bad_energy = (int64_t)tot[BAD_ENERGY];
    ctx->stats.bad_state = (int64_t)tot[BAD_STATE]; ctx->stats.bad_h = (int64_t)tot[BAD_H];
    return SPHX_OK;
}

extern "C" int sphx_get_stats(sphx_ctx* ctx, sphx_stats* out) {
    if (!ctx || !out) return SPHX_E_ARG;
    SPHX_TRY(sphx_dev_collect(ctx));
    if (ctx->map_perm) {               // the device-pointer API has no sphx_step to read the failure counters back: here
        HIPCHK(hipSetDevice(ctx->device));
        SPHX_TRY(sphx_badc_read(ctx));
    }
    *out = ctx->stats;
    // the grid build's single-launch scan bounds its waits; a wait that ran out raised a flag (and left wrong numbers)
    for (int w = 0; w < 2; ++w) {
        if (!ctx->lbs_state[w].p) continue;
        int flag = 0;
        HIPCHK(hipSetDevice(ctx->device));
        HIPCHK(hipMemcpy(&flag, ctx->lbs_state[w].as<int>() + 2, sizeof(int), hipMemcpyDeviceToHost));
        if (flag) return sphx_set_err(ctx, SPHX_E_STATE, "lookback_scan_kernel: a tile's sum never arrived (results since then are wrong)");
    }
    return SPHX_OK;
}
extern "C" int sphx_set_timing_detail(sphx_ctx* ctx, int on) {
    if (!ctx) return SPHX_E_ARG;
    sphx_graph_drop(ctx);
    ctx->timing_detail = on != 0;
    return SPHX_OK;
}

extern "C" int sphx_reset_stats(sphx_ctx* ctx) {
    if (!ctx) return SPHX_E_ARG;
    SPHX_TRY(sphx_dev_collect(ctx));
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    HIPCHK(hipMemsetAsync(ctx->scal.as<u64>() + SC_CAND, 0, 2 * sizeof(u64), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->scal.as<u64>() + SC_SHORT, 0, sizeof(u64), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->badc.p, 0, (size_t)BADC_BUCKETS * BADC_STRIDE * sizeof(u64), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SPHX_OK;
}
