// sphx_sums.hip - kernel-weighted SPH summations of nsc.hydro_update (nsc:556-671).
//
// Three dependent passes over a K-major int32 neighbour list nbr[k][npad] (coalesced for one
// thread per particle), gathering dense 64-B records per neighbour (RecA / RecB):
//   pass 1  rho, rho_dust, n, grad P            needs h_j        nsc:588-619
//   pass 2  Pi_i = sum_k pi_ik, crossing time   needs rho_j      nsc:639-649, nsc:776-786
//   pass 3  viscous accel + heat                needs Pi_j       nsc:651-654
// plus the species pass F[s,i] (nsc:624-627).  Every sum is kept as SPHX_SUM_PARTS partial sums
// over the list positions k mod SPHX_SUM_PARTS (ascending distance), added at the end: a fixed
// order shared with the lanes-per-particle split of the LDS kernels of sphx_blob.hip, so all
// variants agree bit for bit.  In registers, no atomics: results are bitwise reproducible.
// Deltas are taken relative to nbr[0][i] (the reference subtracts neighbour 0, nsc:580-581).
#include "sphx_internal.h"
#include "sphx_wave.h"
// NumPy never fuses a multiply into an add: keep every operation separately rounded so that
// cancellations such as h_j^2 - r^2 at the kernel edge reproduce the reference bit for bit.
#pragma clang fp contract(off)
#include <float.h>
#include <stdlib.h>

struct PrepArgs {
    int n;
    const double *x, *y, *z; int ps;       // position pointers + element stride (1 SoA, 3 AoS)
    const double *vx, *vy, *vz; int vs;
    const double *m, *h, *T, *mu, *gam, *ptype;
    double kB, amu;
    RecA* ra; RecB* rb; RecBC* bc; RecSelf* self;
    const int* perm;                       // sorted -> caller index of the inputs (nullptr: identity)
    int which, n_active;                   // decomposed runs: 0 every particle, 1 the owned ones (perm < n_active), 2 the ghosts
};

__global__ __launch_bounds__(256) void prep_kernel(PrepArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n) return;
    const int i = a.perm ? a.perm[t] : t;
    if (a.which && ((i < a.n_active) != (a.which == 1))) return;
    const double m = a.m[i], h = a.h[i], T = a.T[i], mu = a.mu[i], gam = a.gam[i], pt = a.ptype[i];
    const double g = (pt == 0.0) ? 1.0 : 0.0;
    RecA r; RecB v;
    r.x = a.x[(size_t)i * a.ps]; r.y = a.y[(size_t)i * a.ps]; r.z = a.z[(size_t)i * a.ps];
    const double h2 = h * h, h4 = h2 * h2, h8 = h4 * h4;
    r.h2 = h2;
    r.c1 = 315.0 / (201.06192982974676 * (h8 * h));          // 315/(64 pi h^9)   nsc:588
    r.ms = (pt == 0.0) ? m : ((pt == 2.0) ? -m : 0.0);
    const double nw = m / mu / a.amu;
    r.A = nw * a.kB * T * g;                                  // nsc:615
    r.Nw = nw * g;                                            // nsc:607, nsc:626
    v.x = r.x; v.y = r.y; v.z = r.z; v.h2 = h2;
    v.vx = a.vx[(size_t)i * a.vs]; v.vy = a.vy[(size_t)i * a.vs]; v.vz = a.vz[(size_t)i * a.vs];
    v.cs = sqrt(gam * a.kB * T / mu / a.amu * g);             // nsc:647 neighbour form
    RecBC bc; bc.Bw = 0.0; bc.c1 = r.c1;
    RecSelf sf;
    sf.csi = sqrt(gam * a.kB * T / (mu * a.amu) * g);         // nsc:647 own form
    sf.h = h; sf.mg = m * g; sf.pad = 0.0;
    a.ra[t] = r;
    a.rb[t] = v;
    a.bc[t] = bc;
    a.self[t] = sf;
}

int sphx_prep(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z,
              const double* pos_aos, const double* vx, const double* vy, const double* vz,
              const double* vel_aos, const double* m, const double* h, const double* T,
              const double* mu, const double* gam, const double* ptype) {
    SPHX_TRY(sphx_ensure(ctx, ctx->rec1, (size_t)n * sizeof(RecA)));
    SPHX_TRY(sphx_ensure(ctx, ctx->recv, (size_t)n * sizeof(RecB)));
    SPHX_TRY(sphx_ensure(ctx, ctx->rho_s, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->bc_s, (size_t)n * sizeof(RecBC)));
    SPHX_TRY(sphx_ensure(ctx, ctx->self_s, (size_t)n * sizeof(RecSelf)));
    PrepArgs a;
    a.n = (int)n;
    if (pos_aos) { a.x = pos_aos; a.y = pos_aos + 1; a.z = pos_aos + 2; a.ps = 3; }
    else { a.x = x; a.y = y; a.z = z; a.ps = 1; }
    if (vel_aos) { a.vx = vel_aos; a.vy = vel_aos + 1; a.vz = vel_aos + 2; a.vs = 3; }
    else { a.vx = vx; a.vy = vy; a.vz = vz; a.vs = 1; }
    a.m = m; a.h = h; a.T = T; a.mu = mu; a.gam = gam; a.ptype = ptype;
    a.kB = ctx->cst.k_B; a.amu = ctx->cst.amu;
    a.ra = ctx->rec1.as<RecA>();
    a.rb = ctx->recv.as<RecB>();
    a.bc = ctx->bc_s.as<RecBC>();
    a.self = ctx->self_s.as<RecSelf>();
    a.perm = ctx->map_perm;
    a.which = ctx->map_perm ? ctx->pass_part : 0;
    a.n_active = ctx->map_nactive;
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- neighbour list (n,k) int64 row-major -> [k][npad] int32 ------------------------------
__global__ __launch_bounds__(256) void transpose_nbr_kernel(long long n, int k, int npad,
                                                            const long long* nb, int* out) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * k) return;
    long long i = e / k;
    int kk = (int)(e - i * k);
    long long j = nb[e];
    out[(long long)kk * npad + i] = (j >= 0 && j < n) ? (int)j : -1;
}

int sphx_transpose_nbr(sphx_ctx* ctx, int64_t n, int k, const int64_t* nb_dev) {
    int64_t npad = sphx_pad64(n);
    SPHX_TRY(sphx_ensure(ctx, ctx->nbr, (size_t)k * npad * sizeof(int)));
    HIPCHK(hipMemsetAsync(ctx->nbr.p, 0xFF, (size_t)k * npad * sizeof(int), ctx->stream));
    long long tot = (long long)n * k;
    hipLaunchKernelGGL(transpose_nbr_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       ctx->stream, (long long)n, k, (int)npad, (const long long*)nb_dev,
                       ctx->nbr.as<int>());
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// Output mapping.  The passes run in cell-sorted order; with a map, sorted particle i is the
// caller's particle perm[i], only callers' particles below n_active (the rank's OWNED ones, ghosts
// follow) are computed, and outputs are written in the caller's order.  perm == nullptr: identity.
struct OutMap { const int* perm; int n_active; };
__device__ __forceinline__ int out_index(const OutMap& m, int i) { return m.perm ? m.perm[i] : i; }

// 32-B pieces of a record, loaded as two 16-B vectors each
struct Q4 { double a, b, c, d; };
__device__ __forceinline__ Q4 load4(const double* p) {
    const double2 lo = *reinterpret_cast<const double2*>(p);
    const double2 hi = *reinterpret_cast<const double2*>(p + 2);
    return Q4{lo.x, lo.y, hi.x, hi.y};
}

// The neighbour loops run in chunks of NBATCH: all NBATCH indices are fetched first, then all
// NBATCH records, so several independent gathers are in flight per lane before the first use
// (the passes are bound by gather latency, not arithmetic); the sums still accumulate in list order.
#define NBATCH 4

// (p0 + p1) [+ (p2 + p3)]: the order the lane groups of sphx_blob.hip combine their partial sums in
__device__ __forceinline__ double parts_total(const double (&a)[SPHX_SUM_PARTS]) {
    if (SPHX_SUM_PARTS == 4) return (a[0] + a[1]) + (a[2] + a[3]);
    return a[0] + a[SPHX_SUM_PARTS - 1];
}

// ---- pass 1 ---------------------------------------------------------------------------------
// EXP != 0: timing experiments (extra discarded launch, SPHX_PASS_EXP); 1 = half the record loads
template <int EXP>
__global__ __launch_bounds__(256) void pass_density_kernel(int n, int npad, int k, int clip,
                                                           const int* __restrict__ nbr,
                                                           const RecA* __restrict__ rec, double* rho_s,
                                                           const int* __restrict__ qorder, OutMap om, double* rho,
                                                           double* rhod, double* nden, double* G,
                                                           double* ha) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;    // list column
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;                                         // stored particle
    const int o = out_index(om, i);
    if (o >= om.n_active) return;
    const double* self = reinterpret_cast<const double*>(&rec[i]);
    const Q4 s0 = load4(self), s1 = load4(self + 4);          // x y z h2 | c1 ms A Nw
    int j0 = nbr[p];
    double xr = s0.a, yr = s0.b, zr = s0.c;
    if (j0 >= 0 && j0 != i) { xr = rec[j0].x; yr = rec[j0].y; zr = rec[j0].z; }
    const double hi2 = s0.d, ci = -6.0 * s1.a, Ai = s1.c;
    double a_rho[SPHX_SUM_PARTS] = {}, a_rd[SPHX_SUM_PARTS] = {}, a_n[SPHX_SUM_PARTS] = {};
    double a_gx[SPHX_SUM_PARTS] = {}, a_gy[SPHX_SUM_PARTS] = {}, a_gz[SPHX_SUM_PARTS] = {};
    for (int kk0 = 0; kk0 < k; kk0 += NBATCH) {
      int jb[NBATCH];
      Q4 q0b[NBATCH], q1b[NBATCH];
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) jb[u] = (kk0 + u < k) ? nbr[(size_t)(kk0 + u) * npad + p] : -1;
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
          const double* q = reinterpret_cast<const double*>(&rec[jb[u] < 0 ? i : jb[u]]);
          q0b[u] = load4(q); q1b[u] = (EXP == 1) ? q0b[u] : load4(q + 4);
      }
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        if (jb[u] < 0) continue;
        const Q4 q0 = q0b[u], q1 = q1b[u];
        const double dx = q0.a - xr, dy = q0.b - yr, dz = q0.c - zr;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);   // nsc:586
        const double r2 = r * r;                              // nsc:588 squares the rounded distance
        const double qj = q0.d - r2;
        const double c1 = q1.a, ms = q1.b, Aj = q1.c, Nw = q1.d;
        double W = c1 * (qj * qj * qj);                       // nsc:588
        W = (W < 0.0) ? 0.0 : W;                              // nsc:589
        const double cb = (clip && !(qj > 0.0)) ? 0.0 : -6.0 * c1 * (qj * qj);   // nsc:591 (not clipped; clip: nsc:689)
        const double qi = hi2 - r2;
        const double ca = ci * (qi * qi);                     // nsc:592
        a_rho[u & (SPHX_SUM_PARTS - 1)] += fmax(ms, 0.0) * W;                    // nsc:605
        a_rd[u & (SPHX_SUM_PARTS - 1)] += fmax(-ms, 0.0) * W;                    // nsc:606
        a_n[u & (SPHX_SUM_PARTS - 1)] += Nw * W;                                 // nsc:607
        const double tg = (Aj * cb + Ai * ca) * 0.5;                                 // nsc:615 (as sphx_blob.hip density_batch)
        a_gx[u & (SPHX_SUM_PARTS - 1)] += tg * dx;
        a_gy[u & (SPHX_SUM_PARTS - 1)] += tg * dy;
        a_gz[u & (SPHX_SUM_PARTS - 1)] += tg * dz;
      }
    }
    const double s_rho = parts_total(a_rho), s_rd = parts_total(a_rd), s_n = parts_total(a_n);
    const double gx = parts_total(a_gx), gy = parts_total(a_gy), gz = parts_total(a_gz);
    rho[o] = s_rho; rhod[o] = s_rd; nden[o] = s_n;
    rho_s[i] = s_rho;                                         // sorted order: gathered by pass 2
    if (G) { G[3 * (size_t)o + 0] = -gx; G[3 * (size_t)o + 1] = -gy; G[3 * (size_t)o + 2] = -gz; }
    ha[3 * (size_t)o + 0] = -gx / s_rho;                      // nsc:619
    ha[3 * (size_t)o + 1] = -gy / s_rho;
    ha[3 * (size_t)o + 2] = -gz / s_rho;
}

int sphx_pass_density(sphx_ctx* ctx, int64_t n, int k) {
    SPHX_TRY(sphx_ensure(ctx, ctx->rho, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->rhod, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->nden, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->G, (size_t)n * 3 * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->ha, (size_t)n * 3 * sizeof(double)));
    if (ctx->qorder && ctx->blob_lists) return sphx_blob_density(ctx, n, k);
#ifdef SPHX_EXPERIMENTS
    if (ctx->exp_pass >= 0) {        // timing experiment (SPHX_PASS_EXP), outputs discarded
        const int mode = ctx->exp_pass;
        SPHX_TRY(sphx_ensure(ctx, ctx->in_j, (size_t)n * 12 * sizeof(double)));
        double* d = ctx->in_j.as<double>();
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, ctx->stream));
        if (mode == 1)
            hipLaunchKernelGGL(pass_density_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (int)n, (int)sphx_pad64(n), k, ctx->clip_grad, ctx->nbr.as<int>(), ctx->rec1.as<RecA>(), d,
                               ctx->qorder, OutMap{nullptr, (int)n}, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 8 * n);
        else
            hipLaunchKernelGGL(pass_density_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (int)n, (int)sphx_pad64(n), k, ctx->clip_grad, ctx->nbr.as<int>(), ctx->rec1.as<RecA>(), d,
                               ctx->qorder, OutMap{nullptr, (int)n}, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 8 * n);
        HIPCHK(hipEventRecord(e1, ctx->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        fprintf(stderr, "[sphx] pass_density experiment %d: %.4f ms\n", mode, ms);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
#endif
    hipLaunchKernelGGL(pass_density_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (int)n, (int)sphx_pad64(n), k, ctx->clip_grad, ctx->nbr.as<int>(), ctx->rec1.as<RecA>(),
                       ctx->rho_s.as<double>(), ctx->qorder, OutMap{ctx->map_perm, ctx->map_perm ? ctx->map_nactive : (int)n},
                       ctx->rho.as<double>(), ctx->rhod.as<double>(), ctx->nden.as<double>(),
                       ctx->lean_outputs ? nullptr : ctx->G.as<double>(), ctx->ha.as<double>());
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- pass 2 ---------------------------------------------------------------------------------
__device__ __forceinline__ u64 block_min_u64(u64 v) {
    __shared__ u64 sm[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u64 p = __shfl_xor(v, o, 64);
        v = p < v ? p : v;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 r = sm[0];
    for (int w = 1; w < 4; ++w) r = sm[w] < r ? sm[w] : r;
    return r;
}

__global__ __launch_bounds__(256) void pass_pi_kernel(int n, int npad, int k, const int* __restrict__ nbr,
                                                      const RecB* __restrict__ recb,
                                                      const double* __restrict__ rho_s,
                                                      const RecSelf* __restrict__ selfr, RecBC* bc,
                                                      const int* __restrict__ qorder, OutMap om, double* Pi, double* BwOut,
                                                      u64* ct_bits) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    u64 my_ct = 0x7FF0000000000000ull;       // +inf: "no crossing time"
    const int i = (p < n) ? (qorder ? qorder[p] : p) : n;
    const int o = (i < n) ? out_index(om, i) : 0x7FFFFFFF;
    if (i < n && o < om.n_active) {
        int j0 = nbr[p];
        if (j0 < 0) j0 = i;
        const double* rq = reinterpret_cast<const double*>(&recb[j0]);
        const Q4 r0 = load4(rq), rv = load4(rq + 4);                         // x y z h2 | vx vy vz cs
        const RecSelf sf = selfr[i];
        const double rho_i = rho_s[i], cs_i = sf.csi, ms_i = sf.mg, h_i = sf.h;
        double a_pi[SPHX_SUM_PARTS] = {}, maxrel = 0.0;
        for (int kk0 = 0; kk0 < k; kk0 += NBATCH) {
          int jb[NBATCH];
          Q4 q0b[NBATCH], qvb[NBATCH];
          double rhob[NBATCH];
#pragma unroll
          for (int u = 0; u < NBATCH; ++u) jb[u] = (kk0 + u < k) ? nbr[(size_t)(kk0 + u) * npad + p] : -1;
#pragma unroll
          for (int u = 0; u < NBATCH; ++u) {
              const int jj = jb[u] < 0 ? i : jb[u];
              const double* qb = reinterpret_cast<const double*>(&recb[jj]);
              q0b[u] = load4(qb); qvb[u] = load4(qb + 4);
              rhob[u] = rho_s[jj];
          }
#pragma unroll
          for (int u = 0; u < NBATCH; ++u) {
            if (jb[u] < 0) continue;
            const Q4 q0 = q0b[u], qv = qvb[u];
            const double rho_j = rhob[u];
            const double dx = q0.a - r0.a, dy = q0.b - r0.b, dz = q0.c - r0.c;
            const double dvx = qv.a - rv.a, dvy = qv.b - rv.b, dvz = qv.c - rv.c;
            const double r2 = dx * dx + dy * dy + dz * dz;
            const double dot = dvx * dx + dvy * dy + dvz * dz;
            double w = dot / sqrt(r2 + 0.01 * q0.d);                        // nsc:643
            w = (w > 0.0) ? 0.0 : w;                                        // nsc:644
            const double rho_ab = (rho_j + rho_i) / 2.0;                    // nsc:646
            const double c_ab = 0.5 * (qv.d + cs_i);                        // nsc:647
            a_pi[u & (SPHX_SUM_PARTS - 1)] += -0.5 * (c_ab * 2.0 - 3.0 * w) * w / rho_ab;      // nsc:649
            maxrel = fmax(maxrel, dvx * dvx + dvy * dvy + dvz * dvz);       // nsc:780
          }
        }
        const double s_pi = parts_total(a_pi);
        Pi[o] = s_pi;
        const double bw = fmax(ms_i, 0.0) * s_pi;                           // m Pi [t==0]  nsc:651
        bc[i].Bw = bw;
        if (BwOut) BwOut[o] = bw;
        if (ms_i > 0.0) {                                                   // gas only     nsc:782
            double ct = h_i / sqrt(maxrel);
            if (ct != ct) ct = 0.0;                                         // nan_to_num
            if (ct > DBL_MAX) ct = DBL_MAX;
            if (ct > 0.0) my_ct = (u64)__double_as_longlong(ct);
        }
    }
    u64 bm = block_min_u64(my_ct);
    if (threadIdx.x == 0 && bm != 0x7FF0000000000000ull) atomicMin(ct_bits, bm);
}

int sphx_pass_pi(sphx_ctx* ctx, int64_t n, int k, const double* h, const double* ptype) {
    (void)ptype; (void)h;
    SPHX_TRY(sphx_ensure(ctx, ctx->Pi, (size_t)n * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->Bw, (size_t)n * sizeof(double)));
    u64* ct = ctx->scal.as<u64>() + SC_CT_BITS;
    // SPHX_CT_NONE = "none yet"; in the fused loop the previous step's dt_kernel left it so
    if (!ctx->ct_primed) SPHX_TRY(sphx_prime_ct(ctx, ct));
    ctx->ct_primed = false;
    if (ctx->qorder && ctx->blob_lists) return sphx_blob_pi(ctx, n, k, ct);
    hipLaunchKernelGGL(pass_pi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (int)n, (int)sphx_pad64(n), k, ctx->nbr.as<int>(), ctx->recv.as<RecB>(),
                       ctx->rho_s.as<double>(), ctx->self_s.as<RecSelf>(), ctx->bc_s.as<RecBC>(), ctx->qorder,
                       OutMap{ctx->map_perm, ctx->map_perm ? ctx->map_nactive : (int)n},
                       ctx->Pi.as<double>(), ctx->map_perm ? ctx->Bw.as<double>() : nullptr, ct);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- pass 3 ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pass_visc_kernel(int n, int npad, int k, int clip,
                                                        const int* __restrict__ nbr,
                                                        const RecB* __restrict__ recb,
                                                        const RecBC* __restrict__ bc,
                                                        const int* __restrict__ qorder, OutMap om,
                                                        const double* __restrict__ m, double* va,
                                                        double* vh) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;
    const int o = out_index(om, i);
    if (o >= om.n_active) return;
    int j0 = nbr[p];
    if (j0 < 0) j0 = i;
    const double* rq = reinterpret_cast<const double*>(&recb[j0]);
    const Q4 r0 = load4(rq), rv = load4(rq + 4);
    const double2 bci = *reinterpret_cast<const double2*>(&bc[i]);       // Bw, c1
    const double hi2 = recb[i].h2, ci = -6.0 * bci.y, Bi = bci.x;
    double a_x[SPHX_SUM_PARTS] = {}, a_y[SPHX_SUM_PARTS] = {}, a_z[SPHX_SUM_PARTS] = {}, a_h[SPHX_SUM_PARTS] = {};
    for (int kk0 = 0; kk0 < k; kk0 += NBATCH) {
      int jb[NBATCH];
      Q4 q0b[NBATCH], qvb[NBATCH];
      double2 bcb[NBATCH];
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) jb[u] = (kk0 + u < k) ? nbr[(size_t)(kk0 + u) * npad + p] : -1;
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
          const int jj = jb[u] < 0 ? i : jb[u];
          const double* qb = reinterpret_cast<const double*>(&recb[jj]);
          q0b[u] = load4(qb); qvb[u] = load4(qb + 4);
          bcb[u] = *reinterpret_cast<const double2*>(&bc[jj]);
      }
#pragma unroll
      for (int u = 0; u < NBATCH; ++u) {
        if (jb[u] < 0) continue;
        const Q4 q0 = q0b[u], qv = qvb[u];
        const double2 bcj = bcb[u];
        const double c1 = bcj.y, Bj = bcj.x;
        const double dx = q0.a - r0.a, dy = q0.b - r0.b, dz = q0.c - r0.c;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const double r2 = r * r;
        const double qj = q0.d - r2, qi = hi2 - r2;
        const double cb = (clip && !(qj > 0.0)) ? 0.0 : -6.0 * c1 * (qj * qj);
        const double ca = ci * (qi * qi);
        const double tb = (Bj * cb + Bi * ca) / 2.0;                          // nsc:651, the common factor taken out (see pass 1)
        const double bx = tb * dx, by = tb * dy, bz = tb * dz;
        a_x[u & (SPHX_SUM_PARTS - 1)] += bx; a_y[u & (SPHX_SUM_PARTS - 1)] += by; a_z[u & (SPHX_SUM_PARTS - 1)] += bz;
        a_h[u & (SPHX_SUM_PARTS - 1)] += bx * (qv.a - rv.a) + by * (qv.b - rv.b) + bz * (qv.c - rv.c);   // nsc:653
      }
    }
    const double ax = parts_total(a_x), ay = parts_total(a_y), az = parts_total(a_z), heat = parts_total(a_h);
    va[3 * (size_t)o + 0] = -ax; va[3 * (size_t)o + 1] = -ay; va[3 * (size_t)o + 2] = -az;
    vh[o] = heat * m[o] / 2.0;                                              // nsc:654  (m in output order)
}

int sphx_pass_visc(sphx_ctx* ctx, int64_t n, int k, const double* m) {
    SPHX_TRY(sphx_ensure(ctx, ctx->va, (size_t)n * 3 * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->vh, (size_t)n * sizeof(double)));
    if (ctx->qorder && ctx->blob_lists) return sphx_blob_visc(ctx, n, k, m);
    hipLaunchKernelGGL(pass_visc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (int)n, (int)sphx_pad64(n), k, ctx->clip_grad, ctx->nbr.as<int>(), ctx->recv.as<RecB>(),
                       ctx->bc_s.as<RecBC>(), ctx->qorder,
                       OutMap{ctx->map_perm, ctx->map_perm ? ctx->map_nactive : (int)n}, m,
                       ctx->va.as<double>(), ctx->vh.as<double>());
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- the ordered scatter of the drag reaction (DragScatter, sphx_internal.h) ------------------------
__global__ __launch_bounds__(256) void drag_count_kernel(int n, int npad, int k, const int* __restrict__ nbr,
                                                         const double* __restrict__ ptype, const int* __restrict__ qorder,
                                                         int* cnt) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;
    for (int kk = 0; kk < k; ++kk) {
        const int j = nbr[(size_t)kk * npad + p];
        if (j >= 0 && j != i && ptype[j] == 2.0) atomicAdd(&cnt[j], 1);
    }
}
// One WAVE per receiving particle (grid-stride): lane q holds entry q of the slice (a particle is a neighbour of a few
// dozen others: one entry per lane nearly always), finds its rank by comparing its key with every other one (broadcast
// lane by lane), parks its value at that rank in LDS, and lane 0 adds the parked values in order.  Slices longer than
// 64 entries are added by lane 0 alone, smallest remaining key first.
#define DRAG_RED_BLOCKS 2048
#define DRAG_RED_U 4                      // entries per lane the wave form holds: slices up to 256 entries
// the slice [s, s + L) added in key order -> component c of the sum in lane c (c = 0, 1, 2)
template <int U>
__device__ __forceinline__ void drag_slice_sum(int s, int L, const DragEntry* __restrict__ ent,
                                               double* pk, int lane, double& acc) {
    u64 key[U];
    double vx[U], vy[U], vz[U];
    int rank[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = lane + 64 * u;
        key[u] = ~0ull; vx[u] = vy[u] = vz[u] = 0.0;
        if (q < L) {
            const double2* e = reinterpret_cast<const double2*>(ent + s + q);
            const double2 e0 = e[0], e1 = e[1];
            key[u] = (u64)__double_as_longlong(e0.x); vx[u] = e0.y; vy[u] = e1.x; vz[u] = e1.y;
        }
        rank[u] = 0;
    }
#pragma unroll
    for (int c = 0; c < U; ++c) {
        const int lc = L - 64 * c < 64 ? L - 64 * c : 64;
        for (int t = 0; t < lc; ++t) {
            const u64 kt = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(key[c] >> 32), t) << 32) |
                           (u64)(unsigned)__builtin_amdgcn_readlane((int)key[c], t);
#pragma unroll
            for (int u = 0; u < U; ++u) rank[u] += kt < key[u] ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = lane + 64 * u;
        if (q < L) { pk[3 * rank[u]] = vx[u]; pk[3 * rank[u] + 1] = vy[u]; pk[3 * rank[u] + 2] = vz[u]; }
    }
    wave_sync();
    if (lane < 3) {
        // (in key order, one addition after the other, a lane per component: eight entries' reads are issued together,
        //  their additions follow)
        int t = 0;
        for (; t + 8 <= L; t += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = pk[3 * (t + q) + lane];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; t < L; ++t) acc += pk[3 * t + lane];
    }
    wave_sync();
}
// A wave takes 64 consecutive receivers at a time: every lane looks up one slice (most are empty: zeros written at once),
// the non-empty ones are then added one after the other by the whole wave.
__global__ __launch_bounds__(256) void drag_reduce_kernel(int n, const int* __restrict__ start,
                                                          const DragEntry* __restrict__ ent, double* react) {
    __shared__ double parked[4][64 * DRAG_RED_U * 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* pk = parked[wave];
    for (int j0 = (blockIdx.x * 4 + wave) * 64; j0 < n; j0 += gridDim.x * 4 * 64) {
        const int jl = j0 + lane;
        int sl = 0, Ll = 0;
        if (jl < n) { sl = start[jl]; Ll = start[jl + 1] - sl; }
        if (jl < n && Ll == 0) { react[3 * (size_t)jl] = 0.0; react[3 * (size_t)jl + 1] = 0.0; react[3 * (size_t)jl + 2] = 0.0; }
        u64 todo = __builtin_amdgcn_ballot_w64(Ll > 0);
        while (todo) {
            const int t = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int s = __builtin_amdgcn_readlane(sl, t), L = __builtin_amdgcn_readlane(Ll, t);
            const int j = j0 + t;
            double ax = 0.0, ay = 0.0, az = 0.0, acc = 0.0;
            const bool by_lane = L <= 64 * DRAG_RED_U;
            if (L <= 64) {
                drag_slice_sum<1>(s, L, ent, pk, lane, acc);
            } else if (by_lane) {
                drag_slice_sum<DRAG_RED_U>(s, L, ent, pk, lane, acc);
            } else if (lane == 0) {                       // a very long slice: smallest remaining key first, one lane
                u64 last = 0;
                bool have = false;
                for (int tt = s; tt < s + L; ++tt) {
                    u64 best = ~0ull;
                    int bi = -1;
                    for (int q = s; q < s + L; ++q) {
                        const u64 kq = ent[q].key;
                        if ((!have || kq > last) && kq < best) { best = kq; bi = q; }
                    }
                    ax += ent[bi].x; ay += ent[bi].y; az += ent[bi].z;
                    last = best;
                    have = true;
                }
            }
            if (by_lane) { if (lane < 3) react[3 * (size_t)j + lane] = acc; }
            else if (lane == 0) { react[3 * (size_t)j] = ax; react[3 * (size_t)j + 1] = ay; react[3 * (size_t)j + 2] = az; }
        }
    }
}
int sphx_drag_scatter_plan(sphx_ctx* ctx, int64_t n, int k, const int* nbr, const double* ptype, const int* qorder,
                           DragScatter* out, bool blob) {
    const size_t cap0 = ctx->ds_cnt.cap;
    SPHX_TRY(sphx_ensure(ctx, ctx->ds_cnt, ((size_t)n + 2) * sizeof(int)));
    SPHX_TRY(sphx_ensure(ctx, ctx->ds_start, ((size_t)n + 2) * sizeof(int)));
    // (every reference could be to a dust particle: n k entries; memory is what this machine has)
    SPHX_TRY(sphx_ensure(ctx, ctx->ds_ent, (size_t)n * k * sizeof(DragEntry)));
    if (ctx->ds_cnt.cap != cap0 || ctx->ds_cnt_zeroed != ctx->ds_cnt.p) {     // counted up here, back down by the fill
        HIPCHK(hipMemsetAsync(ctx->ds_cnt.p, 0, ctx->ds_cnt.cap, ctx->stream));
        ctx->ds_cnt_zeroed = ctx->ds_cnt.p;
    }
    out->cnt = ctx->ds_cnt.as<int>(); out->start = ctx->ds_start.as<int>();
    out->ent = ctx->ds_ent.as<DragEntry>();
    if (blob) {          // the step's own list with its blob lists: counted out of LDS (sphx_blob.hip)
        SPHX_TRY(sphx_blob_drag(ctx, n, k, true, nullptr, ptype, nullptr, nullptr, nullptr, nullptr, *out));
    } else {
        hipLaunchKernelGGL(drag_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                           (int)sphx_pad64(n), k, nbr, ptype, qorder, ctx->ds_cnt.as<int>());
        HIPCHK(hipGetLastError());
    }
    SPHX_TRY(sphx_excl_scan_int(ctx, ctx->ds_cnt.as<int>(), ctx->ds_start.as<int>(), (int)n));
    return SPHX_OK;
}
int sphx_drag_scatter_reduce(sphx_ctx* ctx, int64_t n, const DragScatter& d, double* react) {
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(drag_reduce_kernel, dim3((unsigned)(want < DRAG_RED_BLOCKS ? want : DRAG_RED_BLOCKS)), dim3(256), 0,
                       ctx->stream, (int)n, d.start, d.ent, react);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- dust -> gas drag with scatter-added reaction          nsc:719-742 (net_impulse) --------------
// Loop-form semantics inside the step loop: smoothing length of the dust neighbour (Weigh2_dust,
// nsc:678), deltas relative to the particle itself.  The reaction is an ORDERED scatter (DragScatter): the same bits
// on every run, added in the reference's order (source particle by caller index, then list position).
__global__ __launch_bounds__(256) void pass_drag_kernel(int n, int npad, int k, const int* __restrict__ nbr,
                                                        const RecB* __restrict__ recb,
                                                        const double* __restrict__ m,
                                                        const double* __restrict__ ptype,
                                                        const double* __restrict__ mgm,
                                                        const double* __restrict__ mcs,
                                                        const int* __restrict__ qorder, const int* __restrict__ id,
                                                        double* onto, DragScatter sc) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;
    const double* rq = reinterpret_cast<const double*>(&recb[i]);
    const Q4 r0 = load4(rq), rv = load4(rq + 4);
    // (partial sums over the list positions k mod SPHX_SUM_PARTS, as every sum of the step: the LDS form, sphx_blob.hip
    //  blob_drag_kernel, keeps one per lane)
    double a_x[SPHX_SUM_PARTS] = {}, a_y[SPHX_SUM_PARTS] = {}, a_z[SPHX_SUM_PARTS] = {};
    for (int kk = 0; kk < k; ++kk) {
        const int j = nbr[(size_t)kk * npad + p];
        if (j < 0 || ptype[j] != 2.0) continue;                  // dust neighbours only   nsc:736
        double& ox = a_x[kk & (SPHX_SUM_PARTS - 1)];
        double& oy = a_y[kk & (SPHX_SUM_PARTS - 1)];
        double& oz = a_z[kk & (SPHX_SUM_PARTS - 1)];
        const double* qb = reinterpret_cast<const double*>(&recb[j]);
        const Q4 q0 = load4(qb), qv = load4(qb + 4);
        const double dx = q0.a - r0.a, dy = q0.b - r0.b, dz = q0.c - r0.c;
        const double ds2 = q0.d, ds = sqrt(ds2);
        const double q = ds2 - (dx * dx + dy * dy + dz * dz);
        const double ds4 = ds2 * ds2;
        const double wf = m[j] * 315.0 * (q * q * q) / (201.06192982974676 * (ds4 * ds4 * ds));   // nsc:678-681
        double fx = 0.0, fy = 0.0, fz = 0.0;
        if (wf > 0.0) {
            const double dvx = qv.a - rv.a, dvy = qv.b - rv.b, dvz = qv.c - rv.c;
            const double coef = wf / mgm[j] * mcs[j] * sqrt(dvx * dvx + dvy * dvy + dvz * dvz);
            fx = coef * dvx; fy = coef * dvy; fz = coef * dvz;
            ox += fx; oy += fy; oz += fz;
        }
        if (j != i) {                                              // nsc:741 (a zero where the kernel vanishes: counted too)
            const int slot = sc.start[j] + atomicSub(&sc.cnt[j], 1) - 1;
            sphx_drag_put(sc, slot, ((u64)(unsigned)id[i] << 8) | (u64)kk, -fx, -fy, -fz);
        }
    }
    onto[3 * (size_t)i] = parts_total(a_x); onto[3 * (size_t)i + 1] = parts_total(a_y); onto[3 * (size_t)i + 2] = parts_total(a_z);
}

int sphx_pass_drag(sphx_ctx* ctx, int64_t n, int k, const double* m, const double* ptype, const double* mgm,
                   const double* mcs) {
    SPHX_TRY(sphx_ensure(ctx, ctx->drag_on, (size_t)n * 3 * sizeof(double)));
    SPHX_TRY(sphx_ensure(ctx, ctx->drag_re, (size_t)n * 3 * sizeof(double)));
    DragScatter sc;
    const bool blob = ctx->qorder && ctx->blob_lists && ctx->use_lds && ctx->drag_lds && k <= SPHX_MAX_K;
    SPHX_TRY(sphx_drag_scatter_plan(ctx, n, k, ctx->nbr.as<int>(), ptype, ctx->qorder, &sc, blob));
    const int* ids = ctx->map_perm ? ctx->map_perm : ctx->st.id.as<int>();       // the caller's index of a stored particle
    if (blob) {
        SPHX_TRY(sphx_blob_drag(ctx, n, k, false, m, ptype, mgm, mcs, ids, ctx->drag_on.as<double>(), sc));
        return sphx_drag_scatter_reduce(ctx, n, sc, ctx->drag_re.as<double>());
    }
    hipLaunchKernelGGL(pass_drag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                       (int)sphx_pad64(n), k, ctx->nbr.as<int>(), ctx->recv.as<RecB>(), m, ptype, mgm, mcs, ctx->qorder,
                       ctx->map_perm ? ctx->map_perm : ctx->st.id.as<int>(),       // the caller's index of a stored particle
                       ctx->drag_on.as<double>(), sc);
    HIPCHK(hipGetLastError());
    return sphx_drag_scatter_reduce(ctx, n, sc, ctx->drag_re.as<double>());
}

// ---- species pass: F[s,i] = sum_k Nw_j f_un[j,s] W       nsc:624-627 -------------------------
#define SPEC_CHUNK 8
__global__ __launch_bounds__(256) void pass_species_kernel(int n, int npad, int k, int S,
                                                           const int* __restrict__ nbr,
                                                           const RecA* __restrict__ rec,
                                                           const double* __restrict__ fun,
                                                           const int* __restrict__ qorder, double* F) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;
    const int s0 = blockIdx.y * SPEC_CHUNK;
    int j0 = nbr[p];
    if (j0 < 0) j0 = i;
    const double xr = rec[j0].x, yr = rec[j0].y, zr = rec[j0].z;
    double acc[SPEC_CHUNK];
#pragma unroll
    for (int q = 0; q < SPEC_CHUNK; ++q) acc[q] = 0.0;
#pragma unroll 4
    for (int kk = 0; kk < k; ++kk) {
        int j = nbr[(size_t)kk * npad + p];
        if (j < 0) continue;
        const double* q = reinterpret_cast<const double*>(&rec[j]);
        const Q4 q0 = load4(q);
        const double c1 = q[4], Nw = q[7];
        const double dx = q0.a - xr, dy = q0.b - yr, dz = q0.c - zr;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const double qj = q0.d - r * r;
        double W = c1 * (qj * qj * qj);
        W = (W < 0.0) ? 0.0 : W;
        const double wN = Nw * W;
        const double* f = fun + (size_t)j * S + s0;
#pragma unroll
        for (int t = 0; t < SPEC_CHUNK; ++t)
            if (s0 + t < S) acc[t] += wN * f[t];
    }
#pragma unroll
    for (int t = 0; t < SPEC_CHUNK; ++t)
        if (s0 + t < S) F[(size_t)(s0 + t) * n + i] = acc[t];
}

int sphx_pass_species(sphx_ctx* ctx, int64_t n, int k, int s, const double* fun, double* F) {
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)((s + SPEC_CHUNK - 1) / SPEC_CHUNK));
    hipLaunchKernelGGL(pass_species_kernel, grid, dim3(256), 0, ctx->stream, (int)n,
                       (int)sphx_pad64(n), k, s, ctx->nbr.as<int>(), ctx->rec1.as<RecA>(), fun, ctx->qorder, F);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}


// ---- the species pass of the STEP (nsc:624-627 on the step's own list) with the composition's consumers fused in ----
// One thread per particle keeps all S species sums (the array-API kernel above splits them over blockIdx.y), so that
// the epilogue can form, without another pass over memory:
//   Z_i   = sum_{s >= 6} F[s,i] mu_s / sum_s F[s,i] mu_s      the metallicity expression of drv:663 on the SPH-smoothed
//                                                             composition at the particle;
//   agb_i = the AGB dust yields of config_helper.py:183-189 at (Z_i, m_i)        (BASELINE configs[4]: "metallicity
//           lookup fused into the density pass"; the table lives in the context, sphx_state_set_agb).
template <int SMAX>
__global__ __launch_bounds__(256) void step_species_kernel(int n, int npad, int k, int S, int SP, const int* __restrict__ nbr,
                                                           const RecA* __restrict__ rec, const double* __restrict__ fun,
                                                           const int* __restrict__ row_of,
                                                           const int* __restrict__ qorder, const double* __restrict__ m,
                                                           AgbTable agb, int agb_on, double* F, double* Zout,
                                                           double* agb_out) {
    const int p = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = qorder ? qorder[p] : p;
    int j0 = nbr[p];
    if (j0 < 0) j0 = i;
    const double xr = rec[j0].x, yr = rec[j0].y, zr = rec[j0].z;
    double acc[SMAX];
#pragma unroll
    for (int q = 0; q < SMAX; ++q) acc[q] = 0.0;
    // four list positions at a time: indices, then records, then the composition rows, then the arithmetic
    for (int kk0 = 0; kk0 < k; kk0 += 4) {
        int jb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) jb[u] = (kk0 + u < k) ? nbr[(size_t)(kk0 + u) * npad + p] : -1;
        double wN[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = jb[u] < 0 ? i : jb[u];
            const double* q = reinterpret_cast<const double*>(&rec[jj]);
            const Q4 q0 = load4(q);
            const double c1 = q[4], Nw = q[7];
            const double dx = q0.a - xr, dy = q0.b - yr, dz = q0.c - zr;
            const double r = sqrt(dx * dx + dy * dy + dz * dz);
            const double qj = q0.d - r * r;
            double W = c1 * (qj * qj * qj);
            W = (W < 0.0) ? 0.0 : W;
            wN[u] = jb[u] < 0 ? 0.0 : Nw * W;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (jb[u] < 0) continue;
            // (rows are padded with zeros to SP doubles, whole 128-B lines: no bound test, 16-B loads)
            const double2* f = reinterpret_cast<const double2*>(fun + (size_t)(row_of ? row_of[jb[u]] : jb[u]) * SP);
#pragma unroll
            for (int t = 0; t < SMAX / 2; ++t) {
                if (2 * t < SP) {
                    const double2 v = f[t];
                    acc[2 * t] += wN[u] * v.x;
                    acc[2 * t + 1] += wN[u] * v.y;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < SMAX; ++t)
        if (t < S) F[(size_t)t * n + i] = acc[t];
    if (!agb_on) return;
    double heavy = 0.0, all = 0.0;
#pragma unroll
    for (int t = 0; t < SMAX; ++t) {
        if (t < S) {
            const double w = acc[t] * agb.mu[t];
            all += w;
            if (t >= 6) heavy += w;
        }
    }
    const double Z = heavy / all;                            // drv:663 (0/0 -> NaN for a particle without gas neighbours)
    Zout[i] = Z;
    double dust[AGB_MAX_SPEC];
    agb_dust_yields(agb, m[i], Z, dust);
    for (int t = 0; t < S; ++t) agb_out[(size_t)i * S + t] = dust[t];
}

// the species pass on any set of sorted arrays (the step: the resident state; the device API: gathered copies)
int sphx_species_on(sphx_ctx* ctx, int64_t n, int k, int S, int SP, const double* fun, const int* row_of, const double* m_sorted,
                    double* F, double* Z, double* agb) {
    int agb_on = (ctx->agb_on && Z && agb) ? 1 : 0;
#ifdef SPHX_EXPERIMENTS
    if (ctx->exp_no_agb) agb_on = 0;          // timing experiment (SPHX_EXP_NO_AGB): the species sums alone
#endif
    // out of LDS when the step's blob lists are at hand (sphx_blob.hip); the gather form below otherwise
    if (ctx->use_lds && ctx->blob_lists && ctx->qorder && SP == 16 && S <= 16 && k <= SPHX_MAX_K && ctx->species_lds)
        return sphx_blob_species(ctx, n, k, S, fun, row_of, m_sorted, F, Z, agb, agb_on);
    if (S <= 16)       // the reference's 15 species: sums in 16 registers
        hipLaunchKernelGGL(step_species_kernel<16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n,
                           (int)sphx_pad64(n), k, S, SP, ctx->nbr.as<int>(), ctx->rec1.as<RecA>(), fun, row_of, ctx->qorder,
                           m_sorted, ctx->agb, agb_on, F, Z, agb);
    else
        hipLaunchKernelGGL(step_species_kernel<SPHX_MAX_SPECIES>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (int)n, (int)sphx_pad64(n), k, S, SP, ctx->nbr.as<int>(), ctx->rec1.as<RecA>(), fun, row_of, ctx->qorder,
                           m_sorted, ctx->agb, agb_on, F, Z, agb);
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

int sphx_step_species(sphx_ctx* ctx, int64_t n, int k) {
    const int S = ctx->s;
    SPHX_TRY(sphx_ensure(ctx, ctx->F, (size_t)n * S * sizeof(double)));
    if (ctx->agb_on) {
        SPHX_TRY(sphx_ensure(ctx, ctx->Zmet, (size_t)n * sizeof(double)));
        SPHX_TRY(sphx_ensure(ctx, ctx->agb_dust, (size_t)n * S * sizeof(double)));
    }
    // (the rows stay where the upload put them: sorted particle j's row is the one of its id)
    return sphx_species_on(ctx, n, k, S, ctx->sp, ctx->fun_id.as<double>(), ctx->st.id.as<int>(), ctx->st.m.as<double>(), ctx->F.as<double>(),
                           ctx->Zmet.as<double>(), ctx->agb_dust.as<double>());
}
