// sphx_refresh.hip - incremental exact kNN between cell-list rebuilds (Verlet candidate lists).
//
// The full search (sphx_knn.hip) leaves, for every particle i, the 64 nearest particles inside
// its final search radius (list64[i]) and a radius dref[i] such that every particle NOT in the
// list was farther than dref[i] at that time.  While the largest displacement D of any particle
// since then is small, a later kNN of i can be taken from the list alone:
//     an unlisted particle is now at least dref[i] - 2D away, so if the K-th smallest refreshed
//     distance d_K satisfies  d_K < dref[i] - 2D  no unlisted particle can be among the K nearest
//     and the refreshed answer IS the exact kNN (same (d2, index) order as the full search).
// A particle that fails the test is counted; if any fails, the step falls back to a full
// rebuild (cell sort + search), so results never depend on this shortcut.
// One wave per particle: 64 list entries = 64 lanes; one 32-B gather per lane; one 32-bit
// bitonic sort.  No cell list, no candidate streaming, no state permutation on these steps.
#include "sphx_wave.h"

#define RF_BLOCK 256
#define RF_PPB 64

// pos4[i] = {x,y,z,0}; max squared displacement since the list was built -> atomicMax on the bits
__global__ __launch_bounds__(256) void pack_disp_kernel(int n, const double* x, const double* y,
                                                        const double* z, const double* p0, double* pos4,
                                                        u64* disp2_bits) {
    __shared__ u64 sm[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    u64 mine = 0;
    if (i < n) {
        const double xi = x[i], yi = y[i], zi = z[i];
        double4 v; v.x = xi; v.y = yi; v.z = zi; v.w = 0.0;
        *reinterpret_cast<double4*>(&pos4[4 * (size_t)i]) = v;
        const double dx = xi - p0[i], dy = yi - p0[(size_t)n + i], dz = zi - p0[2 * (size_t)n + i];
        const double d2 = dx * dx + dy * dy + dz * dz;
        mine = (d2 == d2) ? (u64)__double_as_longlong(d2) : 0x7FF0000000000000ull;   // NaN -> inf
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u64 p = __shfl_xor(mine, o, 64);
        mine = p > mine ? p : mine;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 r = sm[0];
        for (int w = 1; w < 4; ++w) r = sm[w] > r ? sm[w] : r;
        atomicMax(disp2_bits, r);
    }
}

struct RefreshArgs {
    int n, k, npad;
    const int* list64;
    const double* dref;
    const double* pos4;
    const u64* disp2_bits;
    int* nbr;
    double* h_sorted;
    u64* nfail;
};

__global__ __launch_bounds__(RF_BLOCK) void knn_refresh_kernel(RefreshArgs a) {
    __shared__ int tile[SPHX_MAX_K][RF_PPB + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int base = xcd_block(blockIdx.x, gridDim.x) * RF_PPB;
    const int K = a.k;
    const double D = sqrt(__longlong_as_double((long long)*a.disp2_bits));
    int nfail = 0;

    // the wave's 16 particles: position + dref in one coalesced round trip
    double qx = 0.0, qy = 0.0, qz = 0.0, qd = 0.0;
    {
        const int ip = base + wave * (RF_PPB / 4) + (lane & 15);
        if (lane < RF_PPB / 4 && ip < a.n) {
            const double4 p = *reinterpret_cast<const double4*>(&a.pos4[4 * (size_t)ip]);
            qx = p.x; qy = p.y; qz = p.z;
            qd = a.dref[ip];
        }
    }
    for (int t16 = 0; t16 < RF_PPB / 4; ++t16) {
        const int li = wave * (RF_PPB / 4) + t16;
        const int i = base + li;
        if (i >= a.n) {
            if (lane < K) tile[lane][li] = -1;
            continue;
        }
        const double xi = bcast_f64(qx, t16), yi = bcast_f64(qy, t16), zi = bcast_f64(qz, t16);
        const double dref = bcast_f64(qd, t16);
        const int j = a.list64[(size_t)i * 64 + lane];
        u64 key = KNN_INF;
        if (j >= 0) {
            const double4 p = *reinterpret_cast<const double4*>(&a.pos4[4 * (size_t)j]);
            const double d2 = dist2_nofma(p.x - xi, p.y - yi, p.z - zi);
            key = (d2 == d2) ? (u64)__double_as_longlong(d2) : KNN_INF;
        }
        // order: unique 32-bit key = 26-bit quantisation of d2 against (dref + 2D)^2, then lane
        const double rb = dref < 1e150 ? (dref + 2.0 * D) : 1e150;
        const double r2 = rb * rb * 1.0000001;
        const double qv = fmin(__longlong_as_double((long long)key) * (67108862.0 / r2), 67108862.0);
        const bool inrange = (key != KNN_INF) && (__longlong_as_double((long long)key) <= r2);
        u32 k32 = inrange ? (((u32)qv << 6) | (u32)lane) : 0xFFFFFFFFu;
        sort32_sizes<64>(k32, lane);
        const bool v = (k32 != 0xFFFFFFFFu);
        const int src = (int)(k32 & 63u);
        // fetch the exact (key, index) of the element now at this rank from its original lane
        u64 ck = __shfl(key, src, 64);
        u32 cv = (u32)__shfl(j, src, 64);
        if (!v) { ck = KNN_INF; cv = 0xFFFFFFFFu; }
        const u32 nextq = (u32)__shfl_down((int)(k32 >> 6), 1, 64);
        if (__ballot(v && lane < 63 && nextq == (k32 >> 6))) {
            for (int it = 0; it < 64; ++it) {              // same-bin neighbours: exact order
                const u64 k0 = ck; const u32 v0 = cv;
                cmpx<1>(ck, cv, (lane & 1) == 0);
                const int partner = (lane & 1) ? lane + 1 : lane - 1;
                const int pc = partner < 0 ? 0 : (partner > 63 ? 63 : partner);
                const u64 pk = __shfl(ck, pc, 64);
                const u32 pv = __shfl(cv, pc, 64);
                if (partner >= 0 && partner <= 63) {
                    const bool p_lt = kv_less(pk, pv, ck, cv);
                    if (p_lt == ((lane & 1) != 0)) { ck = pk; cv = pv; }
                }
                if (!__ballot(k0 != ck || v0 != cv)) break;
            }
        }
        // exactness test on the K-th distance
        const u64 kth = __shfl(ck, K - 1, 64);
        const double dK = sqrt(__longlong_as_double((long long)kth));
        const bool ok = (kth != KNN_INF) && (dK < dref - 2.0 * D);
        if (!ok) ++nfail;
        if (lane < K) tile[lane][li] = (ck != KNN_INF) ? (int)cv : -1;
        if (lane == 0) a.h_sorted[i] = (kth != KNN_INF) ? dK : 0.0;
    }
    __syncthreads();
    for (int kk = wave; kk < K; kk += RF_BLOCK / 64) {
        const int i = base + lane;
        if (i < a.npad) a.nbr[(long long)kk * a.npad + i] = tile[kk][lane];
    }
    if (lane == 0 && nfail) atomicAdd(a.nfail, (u64)nfail);
}

// Refresh the K-major neighbour list and h from the Verlet lists.  *nfail_out = number of
// particles whose result could not be proven exact (host value, after a stream sync).
int sphx_knn_refresh(sphx_ctx* ctx, int64_t n, int k, const double* x, const double* y, const double* z,
                     int32_t* nbr, double* h_sorted, int64_t* nfail_out) {
    u64* sc = ctx->scal.as<u64>();
    SPHX_TRY(sphx_ensure(ctx, ctx->pos4, (size_t)n * 4 * sizeof(double)));
    HIPCHK(hipMemsetAsync(sc + SC_DISP2, 0, 2 * sizeof(u64), ctx->stream));      // DISP2 and NFAIL
    hipLaunchKernelGGL(pack_disp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, x, y,
                       z, ctx->pos0.as<double>(), ctx->pos4.as<double>(), sc + SC_DISP2);
    RefreshArgs a;
    a.n = (int)n; a.k = k; a.npad = (int)sphx_pad64(n);
    a.list64 = ctx->list64.as<int>();
    a.dref = ctx->dref.as<double>();
    a.pos4 = ctx->pos4.as<double>();
    a.disp2_bits = sc + SC_DISP2;
    a.nbr = nbr;
    a.h_sorted = h_sorted;
    a.nfail = sc + SC_NFAIL;
    hipLaunchKernelGGL(knn_refresh_kernel, dim3((unsigned)(sphx_pad64(n) / RF_PPB)), dim3(RF_BLOCK), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->pinned, sc + SC_NFAIL, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *nfail_out = (int64_t)(*(u64*)ctx->pinned);
    return SPHX_OK;
}

// remember the positions the Verlet lists were built from (SoA copy: x | y | z)
int sphx_save_list_positions(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z) {
    const size_t nb = (size_t)n * sizeof(double);
    SPHX_TRY(sphx_ensure(ctx, ctx->pos0, 3 * nb));
    double* p = ctx->pos0.as<double>();
    HIPCHK(hipMemcpyAsync(p, x, nb, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(p + n, y, nb, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(p + 2 * n, z, nb, hipMemcpyDeviceToDevice, ctx->stream));
    return SPHX_OK;
}
