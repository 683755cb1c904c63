/* sphx.h - C ABI of libsphx.so: the MI355X (gfx950) implementation of the SPH inner
 * loop of dmuley/sph-code.
 *
 * The reference has no FFI; its boundary for this path is the set of module-level
 * Python callables of sph/navier_stokes_cleaned.py ("nsc") that the driver
 * sph/code_running.py ("drv") invokes as nsc.<name>(...).  Each entry point below
 * names the reference callable it replaces (file:line).  The Python side
 * (sph_code_amd/compat.py) binds them with ctypes and keeps the reference's
 * positional signatures and return tuples.
 *
 * Conventions
 *   - plain pointers and sizes only; every array is C-contiguous; "f64" = double.
 *   - host-pointer entry points copy in, run on the context's HIP stream, copy out
 *     and synchronise before returning; no pointer is retained after return.
 *   - *_dev entry points take DEVICE pointers (memory owned by the caller, e.g. a
 *     torch tensor's data_ptr()) and are asynchronous on the context's stream.
 *   - return value: 0 = ok, < 0 = error (SPHX_E_*); sphx_last_error(ctx) gives text.
 *   - one context per GPU per process; a context is not thread-safe; different
 *     contexts may be used concurrently.
 *   - missing neighbours are encoded as index == n (reference convention, nsc:545-548)
 *     and contribute zero to every sum (the reference raises IndexError).
 */
#ifndef SPHX_H
#define SPHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHX_OK            0
#define SPHX_E_ARG        -1   /* bad argument (null pointer, k > 64, n < 1 ...) */
#define SPHX_E_HIP        -2   /* a HIP runtime call failed                      */
#define SPHX_E_NOMEM      -3   /* device allocation failed                       */
#define SPHX_E_STATE      -4   /* call sequence error (e.g. step before upload)  */

#define SPHX_MAX_K        64   /* one neighbour per lane of a 64-wide wavefront  */
#define SPHX_MAX_SPECIES  32

typedef struct sphx_ctx sphx_ctx;

/* Physical constants read by the path (nsc:21-52).  Defaults are the values the
 * golden fixtures were generated with; sphx_set_constants overrides them. */
typedef struct sphx_constants {
    double k_B;      /* nsc:22  scipy.constants.Boltzmann                  */
    double amu;      /* nsc:25  atomic mass constant                       */
    double m_h;      /* nsc:29  1.0008 amu                                 */
    double m_0;      /* nsc:36  10^1.5 solar masses                        */
    double dt_0;     /* nsc:38  250000 yr                                  */
    double max_age;  /* drv:79  3e7 yr                                     */
    double pos_clamp;/* drv:233 1e11 AU                                    */
} sphx_constants;

/* Per-pass timing / traffic model of the last sphx_step call. */
typedef struct sphx_stats {
    double ms_grid;        /* bbox + cell hash + counting sort + state permute */
    double ms_search;      /* kNN kernel                                        */
    double ms_prep;        /* gather-record build                               */
    double ms_density;     /* pass 1: rho, rho_dust, n, grad P                  */
    double ms_pi;          /* pass 2: Pi_i + crossing time                      */
    double ms_visc;        /* pass 3: viscous accel + heat                      */
    double ms_integrate;   /* dt + leapfrog                                     */
    double ms_total;       /* events around the whole step                      */
    int64_t n;             /* particles                                         */
    int64_t steps;         /* steps accumulated in the ms_* sums                */
    int64_t candidates;    /* candidate distance evaluations in the search      */
    int64_t retries;       /* searches repeated with a larger radius            */
    int64_t cells;         /* cells of the last grid                            */
    int64_t refresh_steps; /* steps whose kNN came from the Verlet lists         */
    int64_t rebuild_steps; /* steps with cell sort + full search                 */
    double  cell_size;     /* edge of the last grid's cells                     */
    double  ms_gravity;    /* self-gravity (0 unless sphx_state_set_gravity)    */
    int64_t fallback_queries; /* last step: queries the grouped search left to the general kernel */
    double  ms_species;    /* species pass of the step (+ metallicity, AGB yields); ms_density excludes it */
    int64_t short_rows;    /* searches that gave up after the last radius of the retry ladder with fewer than K
                              neighbours although more particles exist (pathological states only; 0 otherwise) */
    int64_t detail_steps;  /* steps accumulated in ms_prep .. ms_integrate, ms_gravity, ms_species (sphx_set_timing_detail) */
    int64_t far_queries;   /* last hinted search but one: queries outside the grid box with a search sphere wider than 8 cells */
    int64_t outlier_levels;/* last hinted search: nested outlier levels it was given (0: none built) */
    /* Failure counters (particles, summed over the steps since sphx_reset_stats; 0 in a sane run).  The reference guards
     * its loop with nan_to_num alone (sph/code_running.py:233-238, 460-463, 490-491) and carries on; so does the step -
     * these say how often that guard was needed.  Counted on the device by ballot where the values are in registers.  */
    int64_t bad_accel;     /* pressure / viscous / drag acceleration NaN or inf before nan_to_num (rho_i = 0 or NaN, ...)   */
    int64_t bad_energy;    /* E or heat x dt NaN or inf before the nan_to_num of drv:490                                    */
    int64_t bad_state;     /* updated position or velocity NaN or inf (the next step's clamp, drv:233-238, catches them)    */
    int64_t bad_h;         /* kNN radius 0 (coincident points), NaN or inf                                                  */
    int64_t graph_steps;   /* of `steps`: replays of a captured step graph (launch-bound sizes; SPHX_GRAPH)      */
    int64_t search_steps;  /* steps accumulated in ms_search (replayed steps carry no timing events)                         */
} sphx_stats;

/* ---- context ----------------------------------------------------------------------- */
int         sphx_create(sphx_ctx** out, int device);
void        sphx_destroy(sphx_ctx* ctx);
const char* sphx_last_error(const sphx_ctx* ctx);
int         sphx_version(void);
/* How the library was built ("gfx950 experiments=0" is the product: timing experiments and diagnostics that add
 * launches or change results exist only in -DSPHX_EXPERIMENTS builds), and the SPHX_* environment variables a context
 * read when it was created ("NAME=value ...": performance tunables only, read once, in sphx_create).  A benchmark line
 * carries both, so that it can be shown to be the configuration the parity tests ran.                           */
const char* sphx_build_info(void);
const char* sphx_tunables(const sphx_ctx* ctx);
/* Hardware self-test of the search's matrix-core cull (sphx_knn_group.hip, phase A): `blocks` random 32 x 32 blocks of
 * candidates and queries with coordinates in [-E, E] (cell units, E <= 40) through v_mfma_f32_32x32x16_f16 exactly as the
 * search forms them, compared with fp64.  max_err_over_E2: the largest |D - (d^2 - R^2 (1 + pad))| / E^2 met; kappa: the
 * bound the search's certification assumes; wrong_signs: entries beyond that bound whose sign differed (must be 0).
 * Test infrastructure (tests/test_gpu_parity.py); replaces nothing in the reference. */
/* Self-test of the grid build's single-launch prefix sum (sphx_grid.hip: lookback_scan_kernel) against rocPRIM's scan on
 * the device: n pseudo-random counts through both, entries that differ in *mismatches (must be 0); *single_launch: whether
 * n ran through the single-launch form (scans of more than 512 tiles of 8192 items take rocPRIM's either way).
 * Test infrastructure; replaces nothing in the reference. */
int sphx_selftest_scan(sphx_ctx* ctx, int n, unsigned seed, long long* mismatches, int* single_launch);
int sphx_selftest_mfma_cull(sphx_ctx* ctx, double E, int blocks, unsigned seed, double* max_err_over_E2,
                            double* wrong_signs, double* kappa);
/* Page-locked host memory for the big arrays the array entry points hand back ((N,K) int64 + float64 from
 * sphx_neighbors: 640 MB at N = 1e6, K = 40): device-to-host copies into it run at the link's rate instead of
 * being staged through the runtime's bounce buffers (~4x).  Any host pointer is accepted everywhere; this is
 * an optimisation a binding may use for its output arrays.  SPHX_E_NOMEM if the pages cannot be locked.   */
int         sphx_host_alloc(void** out, size_t bytes);
int         sphx_host_free(void* p);
int         sphx_set_constants(sphx_ctx* ctx, const sphx_constants* c);
int         sphx_get_constants(const sphx_ctx* ctx, sphx_constants* c);
/* Search tuning (performance only, never results): the step loop searches inside
 * rscale * h_previous (default 1.2) and bins particles into cells of edge
 * cell_factor * mean(h) (default 0.6).  Values <= 0 keep the current setting. */
int         sphx_set_tuning(sphx_ctx* ctx, double rscale, double cell_factor);
/* sphx_step times the whole call (ms_total) and every step's search launches (ms_search) with HIP events always; one event
 * per pass (ms_grid, ms_prep ... ms_integrate, ms_gravity, ms_species; stats.detail_steps counts the steps they cover) only
 * when asked: an event record between two dependent kernels costs the stream ~10 us.  Off by default (SPHX_TIMING_DETAIL=1). */
int         sphx_set_timing_detail(sphx_ctx* ctx, int on);
/* Incremental search (off by default).  With verlet != 0 a full search also keeps each
 * particle's 64 nearest candidates; following steps take the exact kNN from those lists as long
 * as it can be PROVEN exact from the displacements since (sphx_refresh.hip), else the step
 * rebuilds.  It pays when the fastest particle moves much less than 0.08 h per step; in the
 * BASELINE workloads (dt >= dt_0/5, sph/code_running.py:226) it does not, hence the default.
 * rscale_build (default 1.3, <= 0 keeps) is the search radius factor on rebuilding steps.      */
int         sphx_set_incremental(sphx_ctx* ctx, int verlet, double rscale_build);

/* ---- nsc.neighbors(points, dist, N_NEIGH)                          nsc:541-552 ----- *
 * Exact k nearest neighbours (Euclidean) within `dist`; `eps` is accepted for API
 * compatibility (the reference passes 0.1 to cKDTree) and ignored: the result is the
 * eps = 0 answer, which satisfies every guarantee of the eps > 0 one.
 *   points  (n,3) f64      idx      (n,k) int64, sorted by distance, column 0 = self,
 *                                   missing = n
 *   dist_out (n,k) f64 (missing = 0)   nontriv (n,) int64   h (n,) f64 = max distance  */
int sphx_neighbors(sphx_ctx* ctx, int64_t n, int k, const double* points, double dist,
                   double eps, int64_t* idx, double* dist_out, int64_t* nontriv, double* h);

/* ---- nsc.hydro_update(neighbor, points, mass, sizes, f_un, particle_type, T, mu_array,
 *                       gamma_array, velocities)                     nsc:556-671 ----- *
 * visc_mode 0 = "ref_axis0": Pi_i = sum_k pi_ik, the axis repair of nsc:649.
 * Outputs keep the reference's sign (+grad P / rho) and layouts:
 *   hydro_accel, visc_accel (n,3); visc_heat, rho, nden, rho_dust (n,); f_un_nb (s,n).
 * Any output pointer may be NULL (that output is skipped).
 * neighbor (here and in the loop forms below) may be NULL = "the (n,k) list of the previous array call
 * on this context": the reference's loop passes one list to eight functions (drv:451-458), which is
 * 320 MB of PCIe traffic each at N = 1e6, K = 40.  SPHX_E_STATE if no list of that shape is held
 * (a search or a step on the context discards it).                                      */
int sphx_hydro_update(sphx_ctx* ctx, int64_t n, int k, int s, const int64_t* neighbor,
                      const double* points, const double* mass, const double* sizes,
                      const double* f_un, const double* particle_type, const double* T,
                      const double* mu_array, const double* gamma_array,
                      const double* velocities, int visc_mode,
                      double* hydro_accel, double* visc_accel, double* visc_heat,
                      double* rho, double* nden, double* f_un_nb, double* rho_dust);

/* ---- loop forms (h = (m/m_0)^(1/3) d, d = the driver-injected global nsc.d) ---------- */
/* nsc.density(points,mass,particle_type,neighbor)                    nsc:693-702        */
int sphx_density(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                 const double* particle_type, const int64_t* neighbor, double d, double* out);
/* nsc.dust_density(points,mass,neighbor,particle_type,sizes)         nsc:704-717        */
int sphx_dust_density(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                      const int64_t* neighbor, const double* particle_type, const double* sizes,
                      double* out);
/* nsc.num_dens(mass,points,mu_array,neighbor)                        nsc:744-753        */
int sphx_num_dens(sphx_ctx* ctx, int64_t n, int k, const double* mass, const double* points,
                  const double* mu_array, const int64_t* neighbor, double d, double* out);
/* nsc.del_pressure(points,mass,particle_type,neighbor,E_internal,gamma_array) nsc:755-774 */
int sphx_del_pressure(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                      const double* particle_type, const int64_t* neighbor,
                      const double* E_internal, const double* gamma_array, double d,
                      double* out /* (n,3) */);
/* nsc.artificial_viscosity(neighbor,points,particle_type,sizes,mass,densities,velocities,
 *                          T,gamma_array,mu_array)                   nsc:788-816        */
int sphx_artificial_viscosity(sphx_ctx* ctx, int64_t n, int k, const int64_t* neighbor,
                              const double* points, const double* particle_type,
                              const double* sizes, const double* mass, const double* densities,
                              const double* velocities, const double* T,
                              const double* gamma_array, const double* mu_array, double d,
                              double* visc_accel /* (n,3) */, double* visc_heat /* (n,) */);
/* nsc.crossing_time(neighbor,velocities,sizes,particle_type)         nsc:776-786        */
int sphx_crossing_time(sphx_ctx* ctx, int64_t n, int k, const int64_t* neighbor,
                       const double* velocities, const double* sizes,
                       const double* particle_type, double* out /* scalar */);
/* nsc.net_impulse(points,mass,sizes,velocities,particle_type,neighbor,f_un) nsc:719-742
 * mean_grain_mass/mean_cross are the per-particle sums meff.f_un, seff.f_un (nsc:720-726),
 * formed by the caller from nsc.grain_mass / nsc.sigma_effective.                        */
int sphx_net_impulse(sphx_ctx* ctx, int64_t n, int k, const double* points, const double* mass,
                     const double* sizes, const double* velocities, const double* particle_type,
                     const int64_t* neighbor, const double* mean_grain_mass,
                     const double* mean_cross, double* accel_onto /* (n,3) */,
                     double* accel_reaction /* (n,3) */);

/* Softened direct-sum gravity on host arrays: accel (n,3).  sizes != NULL: eps = median(sizes)
 * (nsc:358, as grav_force_calculation_new(mass, points, sizes) takes it), else eps = softening.   */
int sphx_gravity_direct(sphx_ctx* ctx, int64_t n, const double* mass, const double* points,
                        const double* sizes, double softening, double G, double* accel);

/* The same sum by multipoles of a cell pyramid over a search-style grid (cells sized for k_cells
 * neighbours, 40 if < 1): level-l cells of 2^l fine cells a side carry (mass, centre of mass) and, at
 * order 2 (sphx_set_gravity_order; the default), their second moments; a particle sums the cells that
 * are children of its parent's +-ws neighbours but not its own +-ws neighbours, level by level, and
 * the particles of the +-ws level-1 cells directly.  ws = 1..4.  rms force error against the exact sum:
 * order 2: ~0.1 % at ws = 1, ~0.01 % at ws = 2; order 1: ~1 %, ~0.2 %.  Every cell is softened like a
 * particle (nsc:385): the expansion is that of the softened kernel.                                 */
int sphx_gravity_tree(sphx_ctx* ctx, int64_t n, const double* mass, const double* points,
                      const double* sizes, double softening, double G, int ws, int k_cells,
                      double* accel);

/* AGB dust and gas return per star: calculate_interpolation, sph/config_helper.py:180-211, on the
 * splines of interpolate_amounts (config_helper.py:138-178; RectBivariateSpline kx = ky = 1).
 * Spline o has ntx[o] / nty[o] knots and (ntx[o]-2)*(nty[o]-2) coefficients; tx, ty, coeffs are the
 * splines' arrays one after the other (SciPy's get_knots() / get_coeffs()).  x = metallicity,
 * y = mass; arguments outside the knot range are clamped as FITPACK does.  dust[i, mapto[o]] =
 * spline_o (a repeated target keeps the last spline), then / divisor and clipped at 0.
 * gas_out (and composition, both (n, nspecies)) may be NULL.  nspecies 6..32, nspl <= 32.        */
int sphx_agb_yields(sphx_ctx* ctx, int64_t n, const double* masses, const double* metallicities,
                    int nspl, const int32_t* ntx, const int32_t* nty, const double* tx,
                    const double* ty, const double* coeffs, const int32_t* mapto, double divisor,
                    int nspecies, const double* mu_specie, const double* composition,
                    double solar_mass, double* dust_out /* (n,nspecies) */,
                    double* gas_out /* (n,nspecies) */);

/* ---- the driver's inline integrator statements as array functions ----------------------- *
 * The reference has no function for these: they are straight-line statements inside its time
 * loop.  Each entry replaces the cited block on host arrays; the device code is the one the fused
 * step loop runs (sphx_leapfrog.h).  Pinned by tests/golden/driver_integrator.npz.          */
/* code_running.py:223-229: dt[i] from crossing time ct[i] (nsc.crossing_time's return value) and
 * first[i] (age == 0): dt_0/10 | max(dt_0/5, min(2 dt_0, ct)); ct > MAX_AGE -> MAX_AGE/100.   */
int sphx_dt_rule(sphx_ctx* ctx, int64_t n, const double* ct, const int32_t* first, double* dt);
/* code_running.py:233-238: |x| <= 1e11 AU, nan_to_num on points and velocities ((n,3), in place) */
int sphx_clamp_arrays(sphx_ctx* ctx, int64_t n, double* points, double* velocities);
/* code_running.py:460-491: pressure_accel = delp/rho [gas]; visc = drag_on_gas rho_dust/rho [gas] +
 * drag_reaction + av_accel (drag pointers NULL: av_accel alone); limiter (:475); total = grav +
 * pressure + visc (grav NULL: none); points += total dt^2/2 + v dt; v += (total + old)/2 dt
 * (old_accel NULL: shapes differ, dv = total dt, :484-485); E += av_heat dt; T = E mu m_h/(gamma m k).
 * In place: points, velocities (n,3), E_internal (n,); out: total_accel (n,3), T (n,).        */
int sphx_leapfrog(sphx_ctx* ctx, int64_t n, double* points, double* velocities, double* total_accel,
                  const double* old_accel, double* E_internal, double* T, const double* mass,
                  const double* mu, const double* gamma, const double* ptype, const double* grav_accel,
                  const double* delp, const double* densities, const double* dust_densities,
                  const double* drag_on_gas, const double* drag_reaction, const double* av_accel,
                  const double* av_heat, double dt);

/* ---- device-resident simulation state: the fused hot path of drv:217-491 ------------- *
 * upload once, step many times (search -> dt -> sums -> leapfrog, nothing leaves HBM),
 * download when needed.  Particle order on the device changes every step (cell sort);
 * download returns arrays in the original order (by particle id).
 *   pos, vel, accel_old (n,3); mass, ptype, T, mu, gamma, E (n,); f_un (n,s) or NULL.   */
int sphx_state_upload(sphx_ctx* ctx, int64_t n, int s, const double* pos, const double* vel,
                      const double* mass, const double* ptype, const double* f_un,
                      const double* T, const double* mu, const double* gamma,
                      const double* E_internal, const double* accel_old);
/* Enable the dust->gas drag of nsc.net_impulse (nsc:719-742) inside the step loop:
 * visc_accel = drag*rho_dust/rho*[gas] + reaction + viscous accel (drv:455,462-463,473).
 * mean_grain_mass / mean_cross (n,) as in sphx_net_impulse; NULL disables.  Call directly after
 * sphx_state_upload.  The reaction (nsc:741) is an ordered scatter: contributions added by source particle (caller
 * index), then list position - np.add.at's order, the same bits on every run (n*K*32 B of device memory). */
int sphx_state_set_drag(sphx_ctx* ctx, const double* mean_grain_mass, const double* mean_cross);
/* Species pass inside the step: when sphx_state_upload was given f_un, every sphx_step (hydro_update mode) also forms
 * F[s,i] = sum_k m_j/(mu_j amu) [gas_j] f_un[j,s] W (nsc:624-627) on its own neighbour list.  With an AGB table set
 * (config_helper.py:138-178: nspl degree-1 splines over (metallicity, mass) as sphx_agb_yields takes them; nspl = 0
 * switches it off) the same pass leaves, without another trip over memory, the per-particle metallicity
 * Z_i = sum_{s>=6} F[s,i] mu_s / sum_s F[s,i] mu_s (the expression of code_running.py:663 on the smoothed composition)
 * and the AGB dust yields of config_helper.py:183-189 at (Z_i, m_i) - BASELINE configs[4].
 * sphx_state_download_species: F (s,n) species-major as nsc:671 returns it, Z (n,), agb_dust (n,s); caller order;
 * any pointer may be NULL.                                                                                          */
int sphx_state_set_agb(sphx_ctx* ctx, int nspl, const int32_t* ntx, const int32_t* nty, const double* tx,
                       const double* ty, const double* coeffs, const int32_t* mapto, double divisor,
                       const double* mu_specie, double solar_mass);
int sphx_state_download_species(sphx_ctx* ctx, double* F, double* Z, double* agb_dust);
/* Step mode "loop forms": on != 0 makes sphx_step evaluate, on its own neighbour list, exactly what the
 * reference's time loop evaluates (drv:451-458): density, dust_density, num_dens, del_pressure,
 * artificial_viscosity and crossing_time in their loop forms (nsc:673-816; smoothing length
 * h(m) = (m/m_0)^(1/3) d with the driver's global d, clipped gradients, physical sign), then
 * pressure_accel = delp / rho [gas], visc_accel = av[0] (+ drag), limiter, leapfrog, E += av[1] dt
 * (drv:460-491).  The default (on == 0) uses the vectorised sums of nsc.hydro_update instead.
 * Call after sphx_state_upload; `dist` of sphx_step is the search bound the driver passes (drv:437). */
int sphx_state_set_loop_forms(sphx_ctx* ctx, int on, double d);
/* Physics option, off by default.  hydro_update's neighbour-side kernel gradient
 * -6 C h_j^-9 (h_j^2 - r^2)^2 is not clipped for r > h_j (nsc:591) - reproduced for parity, but it grows as
 * r^4 and makes long runs of the vectorised form diverge; the reference's own time loop uses the loop forms,
 * which clip it (nsc:689).  on != 0 clips it (0 for h_j^2 - r^2 <= 0) in sphx_hydro_update, sphx_step and
 * the sphx_dev_* passes of this context.                                                          */
int sphx_set_clip_grad(sphx_ctx* ctx, int on);
/* Self-gravity inside the step loop (drv:448-449,477): mode 1 = direct summation with Plummer
 * softening eps = median(h) of the step (nsc:358), G m_j (x_j - x_i) / (|x_j - x_i|^2 + eps^2)^(3/2)
 * summed over ALL particles - the sum the reference's tree approximates (sphx_gravity_direct).
 * O(N^2): meant for N up to a few 10^5.  mode 2 = the same sum by cell-pyramid multipoles
 * (sphx_gravity_tree with ws = 1, order 2; SPHX_GRAV_WS / sphx_set_gravity_order override), ~1.5e3 terms
 * per particle.  mode 0 switches it off.  Call after sphx_state_upload. */
int sphx_state_set_gravity(sphx_ctx* ctx, int mode, double G);
/* Multipole order of the tree's cells (sphx_gravity_tree and mode 2 above): 1 = monopoles, 2 (default) =
 * monopoles + second moments about each cell's centre of mass, the second-order Taylor term of the
 * softened kernel - ws = 1 then beats the accuracy monopoles need ws = 2 for, at 40 % of the cost
 * (DESIGN 5.7).                                                                                         */
int sphx_set_gravity_order(sphx_ctx* ctx, int order);
/* One or more passes of the hot path.  k = N_NEIGH, dist = distance_upper_bound of
 * nsc:544 (<= 0 or inf: unbounded), first != 0: the first step uses dt_0/10 (drv:223-224);
 * fixed_dt > 0 overrides the crossing-time rule (drv:225-229).                           */
int sphx_step(sphx_ctx* ctx, int nsteps, int k, double dist, int first, double fixed_dt);
/* Any output pointer may be NULL.  The particle state (pos ... sizes) belongs to the loop alone; rho, nden
 * and visc_heat are the last step's pass outputs, which live in work buffers the array entry points above
 * also use - ask for them before calling those on the same context (or give the loop a context of its own). */
int sphx_state_download(sphx_ctx* ctx, double* pos, double* vel, double* accel,
                        double* E_internal, double* T, double* sizes, double* rho,
                        double* nden, double* visc_heat, double* dt_last);
/* P_i = n_i k_B T_i (n,), caller order: number density (nsc:607) and temperature of ONE instant - the one the last step's
 * sums were formed at.  (The reference's own `pressure` line is commented out, nsc:608; north_star names P as an output.
 * T of sphx_state_download is the UPDATED temperature, drv:491; the one the sums read is kept by the update at no cost.)
 * Zeros before the first step.  Same caveat as rho / nden above about array calls on the same context.            */
int sphx_state_download_pressure(sphx_ctx* ctx, double* pressure);
int sphx_get_stats(sphx_ctx* ctx, sphx_stats* out);
int sphx_reset_stats(sphx_ctx* ctx);

/* ---- device-pointer building blocks for spatially decomposed (multi-GPU) runs ---------- *
 * A rank holds n_total = n_owned + n_ghost particles in the caller's order, owned first
 * (ghost = copy of a particle owned by another rank, sph_code_amd/multigpu.py).  Searches
 * and sums are computed for the owned particles only; every output array has n_total rows in
 * the caller's order with the owned rows written.  Between the calls the caller fills the
 * ghost rows of h, rho and m*Pi from their owners (RCCL halo exchange), because the kernel
 * uses the NEIGHBOUR's h (nsc:587-588), rho (nsc:646) and Pi (nsc:651).  All pointers are
 * DEVICE pointers; work is queued on the context's stream (sphx_set_stream: e.g. torch's
 * current stream, so no extra synchronisation is needed).                                  */
int sphx_set_stream(sphx_ctx* ctx, void* hip_stream /* a hipStream_t; NULL = HIP's default stream */);
int sphx_reset_stream(sphx_ctx* ctx);            /* back to the context's own stream */
int sphx_sync(sphx_ctx* ctx);
/* mean smoothing length of the previous search (sets the cell size of the next grid)     */
int sphx_dev_set_mean_h(sphx_ctx* ctx, double mean_h);
/* nsc.neighbors over owned+ghost candidates, owned queries.  pos (n_total,3); hint (n_total)
 * previous h per particle or NULL; rscale <= 0: context default; h_out (n_total).          */
int sphx_dev_search(sphx_ctx* ctx, int64_t n_total, int64_t n_owned, int k, const double* pos,
                    const double* hint, double rscale, double dist, double* h_out);
/* cell order of the last sphx_dev_search: out[s] (int32, n_total) = caller index of the s-th
 * particle.  A caller that re-orders its owned arrays accordingly keeps the library's gathers
 * and scatters nearly sequential on the next step.                                            */
int sphx_dev_get_order(sphx_ctx* ctx, int64_t n_total, int32_t* out);
/* gather records; h must be complete (owned from sphx_dev_search, ghosts from their owners) */
int sphx_dev_prep(sphx_ctx* ctx, const double* pos, const double* vel, const double* mass,
                  const double* h, const double* T, const double* mu, const double* gamma,
                  const double* ptype);
/* nsc:588-619; outputs (n_total,) / (n_total,3), any may be NULL                            */
int sphx_dev_density(sphx_ctx* ctx, double* rho, double* rho_dust, double* nden, double* hydro_accel);
/* nsc:639-649 + nsc:776-786; rho_complete (n_total); ct_out: device double = min crossing
 * time over the owned gas particles, or +inf when there is none (votes are at most DBL_MAX)  */
int sphx_dev_pi(sphx_ctx* ctx, const double* rho_complete, double* Pi, double* Bw, double* ct_out);
/* nsc:651-654; Bw_complete (n_total) = m Pi [t==0]; mass (n_total)                           */
int sphx_dev_visc(sphx_ctx* ctx, const double* Bw_complete, const double* mass, double* visc_accel,
                  double* visc_heat);
/* drv:233-238 and drv:460-491 on caller-order (n,3) arrays                                   */
int sphx_dev_clamp(sphx_ctx* ctx, int64_t n, double* pos, double* vel);
/* Halo / migration plumbing of the decomposed driver (sph_code_amd/multigpu.py): particles live in
 * separate device arrays of 8-byte elements, field q being (n, widths[q]); what travels are rows of
 * W = sum(widths) elements.  All pointers are device pointers (the pointer TABLES are host arrays).
 *   pack_rows: rows[t,:] = fields[idx[t],:] for t < n            (idx NULL: identity)
 *   regroup:   fields_out[t] = fields_in[sel[t]] for t < n_sel   (sel NULL: identity),
 *              fields_out[n_sel + r] = rows[r] for r < n_rows.                                   */
int sphx_dev_pack_rows(sphx_ctx* ctx, int64_t n, const int64_t* idx, int nf, const double* const* fields,
                       const int32_t* widths, double* rows);
int sphx_dev_regroup(sphx_ctx* ctx, int64_t n_sel, const int64_t* sel, int64_t n_rows, const double* rows,
                     int nf, const double* const* fields_in, const int32_t* widths,
                     double* const* fields_out);
/* multigpu.py DistributedSim._need_map: out (G^3 bytes, device) = 1 for every cell of the coarse grid
 * (host array g_lo[3] = origin, g_cs = cubic cell edge) within floor(w_i/g_cs) + 1 cells of the cell of an
 * owned particle i (pos (n,3), w (n): device) - the cells a neighbour of i can lie in.                 */
int sphx_dev_need_map(sphx_ctx* ctx, int64_t n, const double* pos, const double* w, const double* g_lo,
                      double g_cs, int G, unsigned char* out);
/* sphx_dev_integrate with the step's verdict and dt taken on the device: red2 (device) = {1 if some rank's halo
 * was too thin, -(global minimum crossing time)}, the driver's one reduction.  red2[0] > 0.5: nothing is changed
 * (the step is redone) and *dt_out = 0; otherwise dt follows drv:222-229 (first, fixed_dt as in sphx_step) and
 * is written to dt_out (device).  The host reads verdict and dt after launching this: no round trip between
 * the sums and the update.                                                                                */
int sphx_dev_integrate_auto(sphx_ctx* ctx, int64_t n_owned, double* pos, double* vel, double* accel_old,
                            double* E_internal, double* T, const double* mass, const double* mu,
                            const double* gamma, const double* ptype, const double* hydro_accel,
                            const double* visc_accel, const double* visc_heat, const double* red2,
                            int first, double fixed_dt, double* dt_out);
/* ---- the same protocol on the LOOP FORMS of the reference's time loop (drv:451-458; multigpu.py, forms = "loop") ---- *
 * after sphx_dev_search: gather the caller's owned + ghost arrays (n_total; E_internal too: del_pressure reads the
 * neighbour's, nsc:755) and build the step's records; d = the driver's global d (drv:68).                               */
int sphx_dev_loop_prep(sphx_ctx* ctx, const double* pos, const double* vel, const double* mass, const double* T,
                       const double* mu, const double* gamma, const double* ptype, const double* E_internal, double d);
/* nsc.density, dust_density, num_dens, del_pressure (nsc:693-717,744-774) for the owned particles; h_complete (n_total):
 * owned from sphx_dev_search, ghosts from their owners (dust_density uses the neighbour's radius, nsc:711).
 * Outputs caller order, (n_total,) / (n_total,3), owned entries written, any may be NULL.                              */
int sphx_dev_loop_pass1(sphx_ctx* ctx, const double* h_complete, double* rho, double* rho_dust, double* nden, double* delp);
/* nsc.artificial_viscosity + crossing_time (nsc:776-816); rho_complete (n_total): ghosts' from their owners (nsc:803).
 * ct_out as sphx_dev_pi.                                                                                                */
int sphx_dev_loop_pass2(sphx_ctx* ctx, const double* rho_complete, double* visc_accel, double* visc_heat, double* ct_out);
/* Overlap of pass 2 with the rho_j halo phase.  The search sorts its workgroups of 128 particles ("blobs") by what they
 * need from other ranks: interior (every neighbour of its owned particles is owned), boundary, idle (ghosts only).
 * sphx_dev_loop_pass2_interior, called after sphx_dev_loop_pass1 and BEFORE the ghosts' densities have arrived, launches
 * pass 2 for the interior blobs (they read only densities pass 1 left on the device); the sphx_dev_loop_pass2 that
 * follows then runs the boundary blobs and delivers the outputs.  Returns 1 if launched, 0 if there is nothing to
 * split (LDS passes off: sphx_dev_loop_pass2 does everything), < 0 on error.  Results do not depend on the split.
 * sphx_dev_blob_split_counts: {interior, boundary, idle} blobs of the last search (a device-to-host read: diagnostics). */
int sphx_dev_loop_pass2_interior(sphx_ctx* ctx);
/* Head-room of the reach sphx_dev_reach / _reach_dt claim, limited to an absolute length `cap` (0: no limit):
 * w_i = max(h_i + min((halo + skin - 1) h_i, cap), h_i + min((halo - 1) h_i, cap) + |v_i| dt).  The driver verifies
 * every plan after the search and redoes the step if a radius outgrew its claim, so this only trades ghosts for redos. */
int sphx_dev_set_reach_cap(sphx_ctx* ctx, double cap);
/* The same overlap for hydro_update's sums.  sphx_dev_select_blobs(part): the calls of sphx_dev_prep / _density / _pi /
 * _visc that follow work on  1: the interior blobs (prep: the owned particles' records),  2: the boundary blobs (prep:
 * the ghosts' records),  0: everything (the default after every sphx_dev_search).  A pass is then called twice with the
 * same output arrays: part 1 while the halo phase it does not depend on is in flight (ghost entries of its *_complete
 * input are not read), part 2 after it.  Returns 1 if the selection is in force, 0 if there is no split (the selection
 * stays "everything": do not call a pass twice), < 0 on error. */
int sphx_dev_select_blobs(sphx_ctx* ctx, int part);
int sphx_dev_blob_split_counts(sphx_ctx* ctx, int32_t counts[3]);
/* drv:460-491 on those outputs: pressure_accel = delp / rho [gas], visc_accel = av[0].  red2 == NULL: the step `dt`;
 * else verdict and dt on the device as sphx_dev_integrate_auto (dt_out required).                                       */
int sphx_dev_integrate_loop(sphx_ctx* ctx, int64_t n_owned, double* pos, double* vel, double* accel_old,
                            double* E_internal, double* T, const double* mass, const double* mu, const double* gamma,
                            const double* ptype, const double* delp, const double* rho, const double* av_accel,
                            const double* av_heat, const double* red2, int first, double fixed_dt, double dt,
                            double* dt_out);
/* Gas-dust drag (nsc:719-742) in the decomposed step: after sphx_dev_prep; mass, ptype, mean_grain_mass, mean_cross
 * (n_total,) ghosts included -> drag_on (n_total,3; owned rows: the sum over each owned particle's dust neighbours) and
 * drag_reaction (n_total,3; EVERY row: the scatter-added reaction nsc:741 - the ghosts' rows are what their owners still
 * have to add: the driver sends them back, the reverse halo).  sphx_dev_set_drag_terms hands the completed terms
 * ((n_owned,...) arrays) to the NEXT sphx_dev_integrate / _auto / _loop call: drv:462-463,473.                        */
int sphx_dev_drag(sphx_ctx* ctx, const double* mass, const double* ptype, const double* mean_grain_mass,
                  const double* mean_cross, double* drag_on, double* drag_reaction);
int sphx_dev_set_drag_terms(sphx_ctx* ctx, const double* drag_on, const double* drag_reaction, const double* rho,
                            const double* rho_dust);
/* Species pass (nsc:624-627) of the decomposed step: after sphx_dev_prep; f_un (n_total,nspecies), mass (n_total,) ghosts
 * included -> F (nspecies,n_total); with a table set (sphx_dev_set_agb, arguments as sphx_state_set_agb) also Z (n_total,)
 * and agb_dust (n_total,nspecies) as sphx_state_download_species defines them.  Owned entries written.                  */
int sphx_dev_set_agb(sphx_ctx* ctx, int nspecies, int nspl, const int32_t* ntx, const int32_t* nty, const double* tx,
                     const double* ty, const double* coeffs, const int32_t* mapto, double divisor,
                     const double* mu_specie, double solar_mass);
int sphx_dev_species(sphx_ctx* ctx, int nspecies, const double* f_un, const double* mass, double* F, double* Z,
                     double* agb_dust);
/* multigpu.py DistributedSim._replan: w_i = max((halo_scale + skin_frac) h_i, halo_scale h_i + |v_i| dt_last), the
 * reach an owned particle claims (h, w (n), vel (n,3): device).                                          */
int sphx_dev_reach(sphx_ctx* ctx, int64_t n, const double* h, const double* vel, double halo_scale,
                   double skin_frac, double dt_last, double* w);
/* the same with dt read from device memory (as sphx_dev_integrate_auto / _loop left it), and the send mask of a plan:
   mask[p][i] = 1 when rank p's need map (maps: (world, G^3) bytes) covers the coarse cell of owned particle i (p != rank),
   counts[p] = how many.  Both let DistributedSim plan the NEXT step's halo before the host has read this step's scalars. */
int sphx_dev_reach_dt(sphx_ctx* ctx, int64_t n, const double* h, const double* vel, double halo_scale,
                      double skin_frac, const double* dt_dev, double* w);
int sphx_dev_plan_mask(sphx_ctx* ctx, int64_t n, const double* pos, const double* g_lo, double g_cs, int G,
                       int world, int rank, const unsigned char* maps, unsigned char* mask, int64_t* counts);
/* multigpu.py DistributedSim.step, this rank's end-of-step scalars in one launch: out4 (device) =
 * { any(h_i + 2 D > w_i) ? 1 : 0,  -(*ct) (ct NULL: untouched),  max h,  mean of the h <= hclip (hclip <= 0: all) }
 * over the n_owned first entries of h and w_plan.                                                        */
int sphx_dev_step_scalars(sphx_ctx* ctx, int64_t n_owned, const double* h, const double* w_plan, double D,
                          double hclip, const double* ct, double* out4);
int sphx_dev_integrate(sphx_ctx* ctx, int64_t n_owned, double* pos, double* vel, double* accel_old,
                       double* E_internal, double* T, const double* mass, const double* mu,
                       const double* gamma, const double* ptype, const double* hydro_accel,
                       const double* visc_accel, const double* visc_heat, double dt);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_H */
