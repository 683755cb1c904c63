// sphx_internal.h - shared declarations of libsphx (gfx950 only; no host fallback).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/sphx.h"
#include "sphx_agb.h"

#define SPHX_WAVE 64
#define SPHX_W6_C 1.5666814710608448    /* 315 / (64 pi), nsc:588 */

typedef unsigned long long u64;
typedef unsigned int u32;

// ------------------------------------------------------------------------------------------
// device buffer with grow-only capacity
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return (T*)p; }
};

// Gather records.  What limits the sum passes is the number of distinct cache lines a
// (particle, neighbour) pair touches, so each pass gets ONE dense 64-B record per neighbour
// (two particles per 128-B line: cell-adjacent neighbours share lines) plus, where a ninth
// value is needed, an 8/16-B gather from a compact array (16 / 8 particles per line):
//   pass 1: RecA                      pass 2: RecB + rho_s[j]        pass 3: RecB + bc_s[j]
struct __attribute__((aligned(64))) RecA {
    double x, y, z;   // position
    double h2;        // h_j^2
    double c1;        // 315 / (64 pi h_j^9)                               nsc:588
    double ms;        // +m (gas), -m (dust), 0 (star): m*[t==0], m*[t==2]  nsc:605-606
    double A;         // m/mu/amu*k*T*[t==0]   pressure weight             nsc:615
    double Nw;        // m/mu/amu*[t==0]       number weight               nsc:607,626
};
struct __attribute__((aligned(64))) RecB {
    double x, y, z, h2;
    double vx, vy, vz;
    double cs;        // sqrt(gamma k T/mu/amu [t==0])   neighbour form    nsc:647
};
struct __attribute__((aligned(16))) RecBC {   // pass-3 companion
    double Bw;        // m Pi [t==0], written by pass 2                     nsc:651
    double c1;        // 315 / (64 pi h^9)
};
struct __attribute__((aligned(32))) RecSelf { // read only by the owner's thread (coalesced)
    double csi;       // sqrt(gamma k T/(mu amu) [t==0]) own form           nsc:647
    double h;         // smoothing length (crossing time)                   nsc:782
    double mg;        // m [t==0]
    double pad;
};

struct GridParams {
    double xmin, ymin, zmin;      // origin of the (robust) grid box
    double cell, inv_cell;
    int nx, ny, nz;
    int ncells;
    float fnx1, fny1, fnz1;       // (float)(nx - 1) ...: kernel arguments cost no VALU conversions
};

// Outlier levels (sphx_grid.hip, sphx_knn.hip): nested cubes around the grid box, each OLEV_N^3 cells, level l reaching
// hmax * 2^l from the box centre along every axis; the particles OUTSIDE the grid box are listed per (level, cell), each in
// the lowest level whose cube holds it (everything beyond the top cube is clamped into its boundary cells).  The grid
// itself still holds every particle (outliers in its boundary cells), so the levels are a second, complete index of
// the outliers only: a far query with a wide search sphere walks a few coarse cells instead of whole faces of the grid.
#define OLEV_N 32
#define OLEV_MAX 20
#define OLEV_MIN_RC 8.0f     // a query outside the box uses the levels when its search radius exceeds this many grid cells
struct OutLevels {
    int L;                   // levels 1..L (0: none built)
    double cx, cy, cz;       // centre of the grid box
    double hmax;             // half of its longest edge
    const int* start;        // [L * OLEV_N^3 + 1]: slice of `list` per (level - 1, cell)
    const int* list;         // storage indices of the outliers
};
// position of a particle relative to the grid box in cell units, exactly as the cell hash computes it (sub, then mul: no
// contraction possible), so "outside" means the same in the grid build, the level build and the search
__device__ __forceinline__ bool sphx_outside_box(const GridParams& g, double x, double y, double z) {
    const double tx = (x - g.xmin) * g.inv_cell, ty = (y - g.ymin) * g.inv_cell, tz = (z - g.zmin) * g.inv_cell;
    return tx < 0.0 || tx >= (double)g.nx || ty < 0.0 || ty >= (double)g.ny || tz < 0.0 || tz >= (double)g.nz;
}

// ---- blob order: the rank of a cell along the space-filling curve (sphx_grid.hip builds the order; the kernel that
// permutes the state, sphx_integrate.hip, can carry its last scatter) ----
struct BlobBits { int bx, by, bz; int hilbert; };
__device__ __forceinline__ unsigned hilbert_rank(unsigned x, unsigned y, unsigned z, int b) {
    // Skilling's transform (axes -> transposed Hilbert index), then the transposed bits interleaved
    unsigned X[3] = {x, y, z};
    const unsigned M = 1u << (b - 1);
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    unsigned out = 0;
    for (int q = 0; q < b; ++q) {
        out |= ((X[2] >> q) & 1u) << (3 * q);
        out |= ((X[1] >> q) & 1u) << (3 * q + 1);
        out |= ((X[0] >> q) & 1u) << (3 * q + 2);
    }
    return out;
}
__device__ __forceinline__ unsigned blob_rank(int cx, int cy, int cz, BlobBits b) {
    if (b.hilbert) return hilbert_rank((unsigned)cx, (unsigned)cy, (unsigned)cz, b.hilbert);
    unsigned out = 0;
    int pos = 0;
    for (int q = 0; q < 11; ++q) {
        if (q < b.bx) out |= (unsigned)((cx >> q) & 1) << pos++;
        if (q < b.by) out |= (unsigned)((cy >> q) & 1) << pos++;
        if (q < b.bz) out |= (unsigned)((cz >> q) & 1) << pos++;
    }
    return out;
}

// device-resident particle state, structure of arrays (one set; `alt` is the permute target)
struct StateArrays {
    DevBuf x, y, z, vx, vy, vz, ax, ay, az;      // position, velocity, previous total accel
    DevBuf m, T, mu, gam, E, hprev;              // per-particle scalars
    DevBuf ptype;                                // f64 0/1/2 as in the reference (drv:127)
    DevBuf id;                                   // int32 persistent particle id
    DevBuf fun;                                  // (n,s) f64 composition (optional)
    DevBuf mgm, mcs;                             // mean grain mass / cross-section (drag, optional)
};

struct sphx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream every launch goes to (own_stream or the caller's)
    hipStream_t own_stream = nullptr;
    hipStream_t side_stream = nullptr;  // independent work beside the main chain (record build while the lists are deduplicated)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_perm_fork = nullptr, ev_perm = nullptr;   // the state's permutation split over the two streams (sphx_permute_state)
    double dev_hmean = 0.0;             // device API: mean h of the previous search (cell size)
    bool knn_hint_by_id = false;        // device API: search-radius hints are in caller order
    char err[512] = {0};
    sphx_constants cst;
    double rscale = 1.08, cell_factor = 0.55;    // search tuning (sphx_set_tuning)
    sphx_stats stats;

    // ---- working set (any particle order) ----
    int64_t n = 0, npad = 0;
    int k = 0, s = 0;
    int sp = 0;                   // doubles per particle of the resident composition rows: s padded to whole 128-B lines
    DevBuf rec1, recv;            // RecA[n], RecB[n]
    DevBuf rho_s, bc_s, self_s;   // sorted-order compact arrays: rho[n], RecBC[n], RecSelf[n]
    DevBuf drag_on, drag_re;      // (n,3) dust->gas drag and its scatter-added reaction (nsc:719-742)
    DevBuf need_pyr;              // sphx_dev_need_map: widest claim per coarse cell + the pyramid of maxima over it
    // the composition rows f_un (sp doubles each) in the ORDER OF UPLOAD, never permuted: the species pass reaches a row
    // through the particle's id (128 B per particle and step that the state's permutation does not move - beside the
    // search, where that copy cost the grouped kernel 84 us on the two-phase cloud)
    DevBuf fun_id;
    DevBuf crowded;                              // cells of 17 .. 512 members, listed by blob_count for cell_sort_crowded
    DevBuf loop_side;                            // loop-form pass 1, LDS form: gamma | dust mass | -1 per particle
    DevBuf ds_cnt, ds_start, ds_ent;             // ordered scatter of the reaction (DragScatter)
    const void* ds_cnt_zeroed = nullptr;
    bool drag = false;            // gas-dust drag enabled in the step loop (sphx_state_set_drag)
    bool dev_ev_pending = false;  // sphx_dev_search recorded ev[1]/ev[2] around its kNN launch: not yet read
    int loop_forms = 0;           // step mode: the loop forms of the reference's time loop (sphx_state_set_loop_forms)
    double loop_d = 0.0;          // their global d (drv:68)
    DevBuf lrec_a, lrec_v;        // loop-form records (sphx_loopforms.hip)
    bool loop_attr_set = false;
    int clip_grad = 0;            // physics option: neighbour-side gradient clipped beyond h_j (sphx_set_clip_grad)
    int gravity = 0;              // 1: direct-sum self-gravity each step (sphx_state_set_gravity)
    double grav_G = 0.0;
    DevBuf grav, grav_sort, grav_tmp;   // (n,3) accelerations, sorted h, radix-sort scratch
    DevBuf grav_pyr, grav_cell;         // cell pyramid (mass, centre of mass), fine cell of each sorted particle
    int grav_order = 2;                 // multipole order of the tree's cells: 1 monopoles, 2 + second moments (sphx_set_gravity_order)
    DevBuf grav_quad;
    bool grav_per_thread = false;       // SPHX_GRAV_KERNEL=0: tree walk per thread instead of per wave through LDS
    int grav_ws = 1;                    // well-separatedness of the tree form (cells)
    // ---- Verlet refresh (sphx_refresh.hip) ----
    DevBuf list64, dref, pos0, pos4;   // int32[n][64], f64[n], f64[3n] positions at list build, f64[4n] packed current
    bool list_valid = false, use_verlet = false;   // opt-in (sphx_set_incremental): pays only for slow drift
    int64_t list_n = 0;
    int list_k = 0;
    double rscale_build = 1.3;          // search radius factor on list-building steps
    DevBuf nbr;                   // int32 [k][npad], K-major, -1 = missing
    const int* map_perm = nullptr;  // device API: sorted -> caller index (nullptr: identity)
    // Processing order of the step loop: column p of the neighbour list / thread p of a pass works
    // on the particle stored at qorder[p].  Storage stays x-fastest by cell (the search walks rows);
    // the processing order follows the cells along a Morton curve, so a workgroup's particles form
    // a compact blob and share most of their neighbours (sphx_grid.hip: sphx_build_blob_order).
    const int* qorder = nullptr;    // nullptr: identity
    DevBuf porder, mcount, mstart;
    bool use_blob = true;
    // device-pointer API: drag terms handed to the next sphx_dev_integrate* call (sphx_dev_set_drag_terms)
    const double *dev_drag_on = nullptr, *dev_drag_re = nullptr, *dev_drag_rho = nullptr, *dev_drag_rhod = nullptr;
    // species pass inside the step (nsc:624-627) + per-particle metallicity + fused AGB yields (sphx_state_set_agb)
    bool agb_on = false;
    AgbTable agb;
    DevBuf agb_knots, Zmet, agb_dust;
    // the fused loop's first pass (sphx_grid.hip grid_count_fused): clamp + box statistics + cell histogram in one kernel
    bool fuse_count = true;               // SPHX_FUSE_COUNT=0: the separate kernels
    double *clamp_vx = nullptr, *clamp_vy = nullptr, *clamp_vz = nullptr;   // set by the step: the grid build applies drv:233-238
    bool bbox_ticket_zeroed = false;
    bool defer_cell_sort = false, cells_unsorted = false;   // the per-cell member sort rides in the blob-order pass
    bool defer_blob_scatter = false, blob_scatter_pending = false;   // the blob order's last scatter rides in the state's permutation
    const void* mcount_zeroed = nullptr;   // the blob-count allocation known to be all zero (cleaned by the deferred scatter)
    int mcount_zeroed_M = 0;
    BlobBits blob_scatter_bits;
    const int* blob_scatter_mstart = nullptr;
    bool species_lds = true;        // SPHX_SPECIES_LDS=0: the species pass by gathers (sphx_sums.hip) also when blob lists exist
    bool split_perm = true;         // SPHX_SPLIT_PERM=0: the whole state permuted in one launch before the search
    bool use_group = true;          // hinted searches by the lane-per-query grouped kernel (SPHX_KNN_GROUP=0: off)
    bool knn_hinted = false;        // set by the callers of sphx_knn whose rsearch holds real previous radii
    DevBuf fail_list;               // queries the grouped kernel hands to the general one (+ their count)
    // outlier levels (see OutLevels): built for a hinted search when the previous one met enough far queries
    int olev_mode = 2;              // SPHX_OUTLIER_LEVELS: 0 never, 1 whenever a particle lies outside the box, 2 auto
    OutLevels olev;                 // olev.L == 0: not built for the current grid
    DevBuf olev_start, olev_fill, olev_list, olev_key;
    const void* olev_fill_zeroed = nullptr;
    double tbox_h[6] = {0, 0, 0, 0, 0, 0};   // host copy of the true bounding box the last grid build knew (may lag a step)
    hipEvent_t olev_ev = nullptr;   // recorded behind the copy of SC_FARQ to the host
    bool olev_ev_valid = false;
    u64 farq_seen = 0;              // SC_FARQ as last read (the counter only grows)
    int64_t farq_last = 0;          // far queries met by the previous hinted search
    DevBuf lbs_state[2];                   // tile words of the look-back scan (sphx_grid.hip), per launching stream
    unsigned lbs_epoch[2] = {0u, 0u};
    bool bb_direct = true;                 // the grid build's box statistics written to pinned memory by the kernel that folds them
    bool species_fused = true;             // the step's species pass inside pass 1's kernel (SPHX_SPECIES_FUSED=0: a kernel of its own)
    DevBuf cell_rank;                       // the particles' arrival numbers in their cells (grid build)
    bool scatter_by_rank = true;           // SPHX_SCATTER_RANK=0: the scatter hands out slots with an atomic of its own
    bool stream_prio = true;               // main stream at the highest, side stream at the lowest device priority
    bool scan_rocprim = false;             // SPHX_SCAN_ROCPRIM=1: rocPRIM's scan instead
    DevBuf tie_list;                       // int4 {query slot, rank, index a, index b}: near ties the grouped search leaves to the list-mode launch's tie blocks
    bool tie_fix = true;
    const void* fcount_zeroed = nullptr;   // the fail-list allocation whose counter the grid build has zeroed for this step
    int64_t fcount_zeroed_n = 0;
    int64_t list_len_last = 0;      // queries the previous hinted search left to the general kernel (sizes the list-mode grid)
    bool knn_lag_external = false;  // the caller copies SC_NFAILQ .. SC_BADHINT out behind the search and hands them back
    bool knn_lag_valid = false;
    u64 knn_lag[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // slots SC_NFAILQ .. SC_KGDBG + 2
    u64 densep_seen = 0;            // SC_DENSEP as last read
    int64_t densep_last = 0;        // particles in cells of >= DENSE_CELL members at the last grid build the host knows of
    u64 crowded_seen = 0;           // SC_CROWDED as last read (the counter only grows)
    bool lean_outputs = false;      // the step loop: the pressure term itself (G; hydro_update returns G / rho) is not stored
    bool cell_feedback = true;      // SPHX_CELL_FEEDBACK=0 switches it off: cells shrink while groups' tiles overflow (sphx_api.hip)
    double cell_scale = 1.0, cell_fb_hi = 0.30, cell_fb_lo = 0.10;     // (SPHX_CELL_FB_HI / _LO)
    int64_t crowded_last = 0;       // cells of 17 .. 512 members at the last grid build the host knows of
    // hint distrust (an experiment kept as an option, off by default): skip the grouped kernel and seed every radius
    // from the local cell counts.  Auto mode enters when the previous hinted search left more than a quarter of its
    // queries to the general kernel (a diverging run: particles move by several h per step) and leaves once fewer than
    // 5 % of the radii found lie beyond [0.5, 1.5] x hint.  Measured on the diverged cube: no faster than the grouped
    // kernel certifying what it can and the list-mode kernel re-seeding the stale hints it meets.
    bool distrust = false;
    u64 badhint_seen = 0;
    int distrust_mode = 0;          // SPHX_HINT_DISTRUST: 0 never, 1 always, 2 auto
    int blob_curve = 0;             // 0: Hilbert where its code space fits, 1: Morton always (SPHX_BLOB_CURVE)
    // sphx_blob.hip: per-workgroup distinct-neighbour lists + 16-bit slot lists for the LDS passes
    DevBuf slot16, uniq;
    bool use_lds = true;            // run the step loop's passes out of LDS (needs blob order)
    bool blob_lists = false;        // slot lists valid for the current neighbour list
    // decomposed runs: blobs by what they need from other ranks (sphx_blob.hip: blob_dedup_kernel's bclass).
    // blob_split: int list[nblk] (interior blobs, then boundary blobs), then cnt[3] {interior, boundary, idle}
    DevBuf blob_class, blob_split;
    bool blob_split_on = true, blob_split_valid = false;
    int blob_split_nblk = 0;
    bool loop2_interior_done = false;
    bool drag_attr_set = false, drag_lds = true;
    bool dedup_pending = false;     // device API: the dedup of the last search runs on the side stream (joined by sphx_blob_join)
    // (off: measured at one rank, 10^6 particles - dedup 109 -> 169 us and the record build 73 -> 95 us when they run side
    //  by side, both HBM-bound: the pair takes the 182 us it takes back to back.  SPHX_DEV_FORK_DEDUP=1 to try it where
    //  the main stream would otherwise wait for the network between the search and the first pass)
    bool dev_fork_dedup = false;
    int64_t max_cells = 0;          // SPHX_MAX_CELLS: > 0 lowers the limit on grid cells, < 0 (experiment) raises it to |value|
    double reach_cap = 0.0;         // sphx_dev_set_reach_cap: head-room of a claimed reach limited to this length (0: not)
    int pass_part = 0;              // which blobs hydro_update's passes and the record build take: 0 all, 1 interior, 2 boundary
    bool blob_attr_set = false;
    int blob_grid = 0;              // persistent workgroups of the LDS passes (0: not yet derived)
    int blob_slots = 1 << 20;       // distinct neighbours staged per workgroup (clamped to the image size)
    int map_nactive = 0;            // device API: callers' particles below this are computed
    DevBuf rho, rhod, nden, G, Pi, Bw, va, vh, ha, F;
    DevBuf scal;                  // small device scalars: ct bits, dt, counters
    // ---- grid ----
    GridParams grid;
    const double* tbox = nullptr;   // device: TRUE bounding box {min xyz, max xyz} of the last grid build
    double clip_lo[3] = {0, 0, 0}, clip_hi[3] = {0, 0, 0};   // statistics window of the robust grid box
    bool clip_valid = false;
    double box_sigmas = 3.0;        // the grid covers mean +- this many standard deviations of the positions (SPHX_BOX_SIGMAS)
    double h_clip = 0.0;            // h above this is left out of the mean that sizes the cells (0: none)
    double h_clip_factor = 8.0;     // ... = this many times the previous mean (SPHX_HCLIP)
    DevBuf cell_of, cell_start, cell_fill, perm, inv, scan_tmp, bbox_tmp;
    // ---- host-API staging ----
    DevBuf in_a, in_b, in_c, in_d, in_e, in_f, in_g, in_h, in_i, in_j, out_a, out_b, out_c;
    DevBuf idx64, dist_out, nontriv, h_api;
    // ---- simulation state ----
    StateArrays st, alt;
    // The update writes the new temperatures into alt.T (dead once the state has been permuted) and swaps it with st.T:
    // what is left in alt.T is T at the instant of the step's sums, in the step's sorted order like rho / nden - P_i = n_i
    // k_B T_i of ONE instant at no cost to the step (sphx_state_download_pressure).  (st.T and alt.T are then the same two
    // buffers at the start of every step; the other arrays of st / alt swap roles every step: period 2, which the step
    // graphs below rely on.)
    bool tprev_valid = false;
    // ---- step graphs (sphx_api.hip sphx_step): the launch-bound sizes.  A quiet step's ~25 launches replayed as a hipGraph
    // cost half the launch overhead of the stream (tools/graphrate.hip: 25 dependent small kernels 100 -> 50 us).  One REAL
    // step sizes the grid and takes the search's decisions from the lagged read-backs as always; the next two steps are
    // CAPTURED with those decisions frozen (one per parity of the double-buffered state) and then replayed alternately for
    // up to graph_epoch steps - the grid box and cell size only steer performance (out-of-box particles are clamped into the
    // boundary cells, which the search handles exactly), and every replayed step still computes its own true bounding box on
    // the device.  Frozen too: the list-mode launch's grid, no outlier levels (a state that needs them is not replayed).
    int graph_mode = 0;                 // SPHX_GRAPH: 0 never (default: measured slower), 1 whenever a step can be replayed, 2 for n <= graph_max_n
    int64_t graph_max_n = 200000;       // SPHX_GRAPH_MAX_N
    int graph_epoch = 64;               // SPHX_GRAPH_EPOCH: replays between two real steps
    bool in_fused_step = false;         // sphx_build_grid is being called by the fused loop's step (any other grid build drops the graphs)
    bool grid_fused = false;            // the last grid build took its statistics from the fused count kernel (no host wait)
    bool capturing = false;             // one_step is being recorded, not run: no host waits, no timing events, no read-backs
    bool replay_ok = false;             // the last real step left decisions a replay may freeze (hinted, fused, no levels ...)
    hipGraphExec_t gexec[2] = {nullptr, nullptr};      // by parity of step_count
    hipGraphExec_t retired[16] = {nullptr};            // forgotten, possibly still running: destroyed behind the next stream wait
    int n_retired = 0;
    int graph_k = 0; double graph_dist = 0.0, graph_fixed_dt = 0.0;     // the sphx_step arguments the graphs were captured with
    int graph_left = 0;                 // replays left in this epoch
    double cap_bb[13] = {0}, cap_cell_hint = 0.0, cap_h_clip = 0.0;   // what the real step read from the lagged slots
    int64_t graph_steps = 0;            // steps that were replays (statistics)
    DevBuf badc;                  // failure counters, BADC_BUCKETS x BADC_STRIDE u64 (zeroed at sphx_create / sphx_reset_stats)
    bool has_state = false;
    int64_t step_count = 0;
    double dt_last = 0.0;
    hipEvent_t ev[10] = {nullptr};
    // The fused step loop never waits for the step it is launching: its timing events live in a
    // ring (collected two steps late), and the two host read-backs that size the grid - bounding
    // box statistics and mean h - are taken from the PREVIOUS step's copies (slots in `pinned` at
    // LAG_OFF; the grid box only steers performance: out-of-box particles are clamped into the
    // boundary cells, which the search handles exactly).
    // timing experiments (SPHX_KNN_ABL / SPHX_BLOB_EXP / SPHX_BLOB_EXP_LDS / SPHX_PASS_EXP: an extra,
    // discarded launch of a cut-down kernel), read from the environment once, at sphx_create
    const void* cell_fill_zeroed = nullptr;   // the cell_fill allocation known to be all zero between grid builds
    // the K-major list in `nbr` is the caller-order list of the last array-API call with this shape
    // (a later call may pass neighbor == NULL instead of uploading the same (N, K) int64 array again)
    bool nbr_api_valid = false;
    int64_t nbr_api_n = 0;
    int nbr_api_k = 0;
    bool ct_primed = false;             // SC_CT_BITS holds "none yet" (left so by dt_kernel)
    DevBuf scal_tmp;                    // step_scalars_kernel's per-block partials + its ticket
    DevBuf hsum_tmp;                    // hsum_kernel's per-block partial sums + its ticket
    int exp_knn = -1, exp_blob = 0, exp_pass = -1;      // (-DSPHX_EXPERIMENTS builds only: never set otherwise)
    size_t exp_blob_lds = 0;
    bool exp_no_agb = false, knn_prof_print = false, kg_debug_print = false;
    char tunables[1024] = {0};          // "NAME=value ..." of every SPHX_* variable sphx_create read (sphx_tunables)
    hipEvent_t evring[3][10] = {{nullptr}};
    bool timing_detail = false;         // per-pass timing events in sphx_step (sphx_set_timing_detail)
    bool ev_detail[3] = {false, false, false};
    int ev_has07[3] = {0, 0, 0};        // bit 0 / 1: the slot's step recorded its start / end event
    unsigned ev_pending = 0;            // bit s: ring slot s holds an uncollected step
    hipEvent_t lag_bev[2] = {nullptr, nullptr}, lag_hev[2] = {nullptr, nullptr};
    // An event record costs the stream ~10 us: where the fused loop records one anyway right behind a read-back's copy
    // (the search's start event behind the box statistics, the side stream's join event behind the h sums), the host
    // waits on that one instead of a record of its own (the alias; nullptr: lag_bev / lag_hev were recorded).
    hipEvent_t lag_balias[2] = {nullptr, nullptr}, lag_halias[2] = {nullptr, nullptr};
    hipEvent_t step_ev1 = nullptr;      // set by the fused loop around its grid build: the event it records before the search
    bool lag_on = false;                // set by the fused loop around its grid build
    bool lag_bvalid[2] = {false, false}, lag_hvalid[2] = {false, false};
    int64_t lag_bn[2] = {0, 0};
    int lag_bslot = 0, lag_hslot = 0;
    void* pinned = nullptr;       // small pinned host scratch for scalar read-back
};

int sphx_set_err(sphx_ctx* ctx, int code, const char* fmt, ...);
void sphx_graph_drop(sphx_ctx* ctx);      // forget the step graphs (new state, other step mode ...)
void sphx_graph_reap(sphx_ctx* ctx);      // ... and destroy them, once the stream has been waited for
int sphx_ensure(sphx_ctx* ctx, DevBuf& b, size_t bytes);

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return sphx_set_err(ctx, SPHX_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__,       \
                                #expr, hipGetErrorString(e_));                               \
    } while (0)
#define SPHX_TRY(expr)                                                                       \
    do {                                                                                     \
        int r_ = (expr);                                                                     \
        if (r_ != SPHX_OK) return r_;                                                        \
    } while (0)

static inline int64_t sphx_pad64(int64_t n) { return (n + 63) & ~int64_t(63); }

// XCD-aware block remap: workgroups are dealt round-robin over the 8 XCDs (block b runs on
// XCD b % 8, each with its own 4 MiB L2).  Map the blocks of one XCD onto one CONTIGUOUS
// chunk of the (cell-sorted) particle range so an XCD's L2 only ever sees its own spatial
// slab plus a halo.  Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, x = b & 7, s = b >> 3;
    return x * q + (x < r ? x : r) + s;
}

// Every neighbour sum is kept as this many partial sums over the list positions k mod SPHX_SUM_PARTS,
// combined as (p0 + p1) [+ (p2 + p3)]: one fixed order for the gather kernels (sphx_sums.hip) and the
// lanes-per-particle split of the LDS kernels (sphx_blob.hip), so all variants agree bit for bit.
#ifndef SPHX_SUM_PARTS
#define SPHX_SUM_PARTS 4
#define LAG_OFF 1024                 // byte offset of the lag slots in ctx->pinned: slot s at LAG_OFF + 512 s (box), + 256 (h sums)
#endif

// "no gas particle has voted a crossing time yet": the bits of +inf.  Votes are nan_to_num'ed
// (at most DBL_MAX = 0x7FEF...), so no vote can produce it and atomicMin on the bits keeps any vote.
#define SPHX_CT_NONE 0x7FF0000000000000ull
int sphx_prime_ct(sphx_ctx* ctx, u64* ct_bits);      // writes SPHX_CT_NONE on the context's stream

// scalar slots in ctx->scal (8-byte units)
enum {
    SC_CT_BITS = 0,   // u64: min crossing time (bits of a positive double)
    SC_DT = 1,        // f64: time step
    SC_CAND = 2,      // u64: candidate evaluations
    SC_RETRY = 3,     // u64: retried searches
    SC_HSUM = 4,      // f64: sum of h (for the next grid's cell size)
    SC_DISP2 = 5,     // u64: bits of the max squared displacement since the Verlet list was built
    SC_NFAIL = 6,     // u64: particles whose refreshed kNN could not be proven exact
    SC_HCNT = 7,      // f64: number of h values in SC_HSUM
    SC_GRAV_EPS = 8,  // f64: gravitational softening = median(h) of this step (nsc:358)
    SC_NFAILQ = 9,    // u32: queries of the last hinted search left to the general kernel
    SC_SHORT = 10,    // u64: searches that gave up (KNN_MAX_TRIES radii) with fewer than K neighbours although more exist
    SC_FARQ = 11,     // u64, only grows: queries outside the grid box with a search sphere wider than OLEV_MIN_RC cells
    SC_BADHINT = 12,  // u64, only grows: hinted queries (distrust mode) whose radius came out beyond [0.5, 1.5] x the hint
    SC_CROWDED = 13,  // u64, only grows: cells of 17 .. 512 members met by blob_count (sorted by a launch of their own when many)
    SC_DENSEP = 14,   // u64, only grows: particles in cells of >= DENSE_CELL members (a tile's 27 such cells overflow it)
    SC_KGDBG = 16,    // u64[8]: grouped search, queries handed on by reason (diagnostics)
    SC_KNNPROF = 24,  // u64[16]: general search, cycles / queries / longest / tries by query class (-DSPHX_KNN_PROF builds)
    SC_NSLOTS = 48
};

// Failure counters (the reference's only guard is the nan_to_num of drv:233-238, 460-463, 490-491), u64 and
// only growing, counted by ballot where the values are in registers anyway: one atomic per wave that saw any.  Under
// hydro_update's sums EVERY particle is counted from the second step on (T < 0), so the counts are spread over
// BADC_BUCKETS cache lines by wave - 31 000 atomics per step on ONE address cost the update kernel 0.33 ms (measured), on 64
// lines nothing - and added up on the host (ctx->badc: [bucket][BADC_STRIDE] u64).
enum {
    BAD_ACCEL = 0,   // pressure / viscous / drag acceleration NaN or inf before drv:460-463's nan_to_num (rho_i = 0 or NaN, T_j < 0 ...)
    BAD_ENERGY = 1,  // E or heat x dt NaN or inf before drv:490's nan_to_num
    BAD_STATE = 2,   // updated position or velocity NaN or inf (the next step's clamp, drv:233-238, catches them)
    BAD_H = 3,       // kNN radius 0 (coincident points), NaN or inf
    BADC_BUCKETS = 64, BADC_STRIDE = 16
};
int sphx_badc_read(sphx_ctx* ctx);          // sums the buckets into ctx->stats.bad_* (waits for the stream)

// ---- kernel launch wrappers (defined in the .hip files) ---------------------------------
// grid
int sphx_bbox(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z,
              double out_minmax[13], bool use_clip);
int sphx_median(sphx_ctx* ctx, int64_t n, const double* v, double* out_dev);
int sphx_gravity_launch(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z, int ps,
                        const double* m, const double* eps_dev, double eps, double G, const int* omap,
                        double* acc);
int sphx_gravity_tree_launch(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z,
                             const double* m, int ws, const double* eps_dev, double eps, double G, const int* omap,
                             double* acc);
int sphx_dev_collect(sphx_ctx* ctx);
int sphx_loop_step_sums(sphx_ctx* ctx, int64_t n, int k, double d);
int sphx_build_blob_order(sphx_ctx* ctx, int64_t n);
// the factor on the cell size for the next grid over n particles (cell_scale, moved one step by the last known dense share)
static inline double sphx_cell_feedback(sphx_ctx* ctx, int64_t n) {
    if (ctx->cell_feedback && n > 0) {
        const double dense = (double)ctx->densep_last / (double)n;
        if (dense > ctx->cell_fb_hi) ctx->cell_scale = ctx->cell_scale * 0.97 > 0.35 ? ctx->cell_scale * 0.97 : 0.35;
        else if (dense < ctx->cell_fb_lo) ctx->cell_scale = ctx->cell_scale * 1.01 < 1.0 ? ctx->cell_scale * 1.01 : 1.0;
    }
    return ctx->cell_scale;
}
int sphx_blob_translate(sphx_ctx* ctx, int64_t n, int k);
int sphx_blob_density_species(sphx_ctx* ctx, int64_t n, int k, int S, const double* fun, const int* row_of, const double* m_sorted,
                              double* F, double* Z, double* agb, int agb_on);
// device API: the main stream waits for the slot lists of the last search (built beside the record build / the h_j phase)
int sphx_blob_join(sphx_ctx* ctx);
int sphx_blob_density(sphx_ctx* ctx, int64_t n, int k);
int sphx_blob_pi(sphx_ctx* ctx, int64_t n, int k, u64* ct_bits);
int sphx_blob_visc(sphx_ctx* ctx, int64_t n, int k, const double* m);
int sphx_blob_species(sphx_ctx* ctx, int64_t n, int k, int S, const double* fun, const int* row_of, const double* m_sorted, double* F,
                      double* Z, double* agb, int agb_on);
int sphx_build_grid(sphx_ctx* ctx, int64_t n, int k, const double* x, const double* y,
                    const double* z, double cell_hint);   // fills grid, cell_start, perm
int sphx_excl_scan_int(sphx_ctx* ctx, const int* in, int* out, int n);   // out[0..n] = exclusive prefix sums, out[n] = total (in[n] must be 0)
// The drag reaction (nsc:741: every particle adds -f to each of its dust neighbours) as an ORDERED scatter: the
// contributions to a particle are first laid side by side (slices from a count + scan), then added in the order the
// reference's np.add.at adds them - by source particle (caller index), then list position - so the sum is the same bits on
// every run, and the reference's.  sphx_drag_scatter_plan: count + scan; the drag kernels fill; sphx_drag_scatter_reduce adds.
// (an entry is one aligned 32-byte record - key and the three components leave in one piece: written as four scattered
//  8-byte stores, the fill cost four memory sectors per reference)
struct alignas(32) DragEntry { u64 key; double x, y, z; };
struct DragScatter { int* cnt; const int* start; DragEntry* ent; };
__device__ __forceinline__ void sphx_drag_put(const DragScatter& sc, int slot, u64 key, double x, double y, double z) {
    double2* q = reinterpret_cast<double2*>(sc.ent + slot);
    q[0] = make_double2(__longlong_as_double((long long)key), x);
    q[1] = make_double2(y, z);
}
int sphx_drag_scatter_plan(sphx_ctx* ctx, int64_t n, int k, const int* nbr, const double* ptype, const int* qorder, DragScatter* out,
                           bool blob = false);
int sphx_drag_scatter_reduce(sphx_ctx* ctx, int64_t n, const DragScatter& d, double* react);
// the step's drag pass on the blob lists (sphx_blob.hip): count_only = the counting pass of the scatter plan
int sphx_blob_drag(sphx_ctx* ctx, int64_t n, int k, bool count_only, const double* m, const double* ptype, const double* mgm,
                   const double* mcs, const int* id, double* onto, const DragScatter& sc);
int sphx_build_outlier_levels(sphx_ctx* ctx, int64_t n, const double* xs, const double* ys, const double* zs);   // (sorted order)
// knn
struct KnnOut {
    int32_t* nbr;       // [k][npad] sorted indices (nullable)
    int32_t* list64 = nullptr;   // [n][64] Verlet candidate list (nullable)
    double* dref = nullptr;      // [n] exclusion radius of list64 (nullable)
    double* h_sorted;   // [n] in sorted order (nullable)
    int64_t* idx64;     // (n,k) by id (nullable)
    double* dist;       // (n,k) by id (nullable)
    int64_t* nontriv;   // (n,) by id (nullable)
    double* h_by_id;    // (n,) by id (nullable)
};
int sphx_knn(sphx_ctx* ctx, int64_t n, int k, const double* xs, const double* ys,
             const double* zs, const int32_t* id, const int32_t* inv, const double* rsearch,
             double rscale, double rbound, const KnnOut& out);
// sums
int sphx_prep(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z,
              const double* pos_aos, const double* vx, const double* vy, const double* vz,
              const double* vel_aos, const double* m, const double* h, const double* T,
              const double* mu, const double* gam, const double* ptype);
int sphx_pass_density(sphx_ctx* ctx, int64_t n, int k);
int sphx_pass_pi(sphx_ctx* ctx, int64_t n, int k, const double* h, const double* ptype);
int sphx_pass_visc(sphx_ctx* ctx, int64_t n, int k, const double* m);
int sphx_pass_species(sphx_ctx* ctx, int64_t n, int k, int s, const double* fun, double* F);
int sphx_step_species(sphx_ctx* ctx, int64_t n, int k);
// fun: composition rows; row_of (nullable: identity): the row of sorted particle j is fun + row_of[j] * SP
int sphx_species_on(sphx_ctx* ctx, int64_t n, int k, int S, int SP, const double* fun, const int* row_of, const double* m_sorted,
                    double* F, double* Z, double* agb);
int sphx_agb_table_set(sphx_ctx* ctx, int S, int nspl, const int32_t* ntx, const int32_t* nty, const double* tx, const double* ty,
                       const double* coeffs, const int32_t* mapto, double divisor, const double* mu_specie, double solar_mass);
int sphx_transpose_nbr(sphx_ctx* ctx, int64_t n, int k, const int64_t* nb_rowmajor);

// integrate / layout helpers (sphx_integrate.hip)
int sphx_clamp(sphx_ctx* ctx, int64_t n, StateArrays& s);
int sphx_permute_state(sphx_ctx* ctx, int64_t n, bool split = false, hipEvent_t after_first = nullptr);
int sphx_hsum(sphx_ctx* ctx, int64_t n, const double* h);
int sphx_compute_dt(sphx_ctx* ctx, int first, double fixed_dt);
int sphx_integrate(sphx_ctx* ctx, int64_t n, int fold_dt = 0, int first = 0, double fixed_dt = 0.0);
int sphx_pass_drag(sphx_ctx* ctx, int64_t n, int k, const double* m, const double* ptype, const double* mgm,
                   const double* mcs);
int sphx_aos_to_soa3(sphx_ctx* ctx, int64_t n, const double* aos, double* x, double* y, double* z);
int sphx_soa3_to_aos_by_id(sphx_ctx* ctx, int64_t n, const int* id, const double* x, const double* y,
                           const double* z, double* aos);
int sphx_scatter_rows_by_id(sphx_ctx* ctx, int64_t n, int w, const int* id, const double* in, double* out);
int sphx_gather3(sphx_ctx* ctx, int64_t n, const int* perm, const double* x, const double* y,
                 const double* z, double* xs, double* ys, double* zs);
int sphx_iota(sphx_ctx* ctx, int64_t n, int* out);
// Verlet refresh (sphx_refresh.hip)
int sphx_knn_refresh(sphx_ctx* ctx, int64_t n, int k, const double* x, const double* y, const double* z,
                     int32_t* nbr, double* h_sorted, int64_t* nfail_out);
int sphx_save_list_positions(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z);
