// sphx_grid.hip - spatial hash: bounding box, cell ids, counting sort (cell list in HBM).
//
// Cell index is x-fastest: cell = (cz*ny + cy)*nx + cx, so a run of cells along x at fixed
// (cy,cz) is one contiguous range of the sorted particle arrays (what sphx_knn.hip streams).
// Coordinates outside the box are clamped into the boundary cells by the same monotone map
// the search uses for its range ends, so clamping never loses a neighbour.
#include "sphx_internal.h"
#include <rocprim/rocprim.hpp>
#include <float.h>

#define RED_BLOCK 256
#define RED_MAXBLOCKS 1024
#define FUSED_MAXBLOCKS 4096        // grid_count_fused: one particle per thread up to 1e6 (histogram atomics want threads in flight)

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// partial[block][13] = {min xyz, max xyz, sum xyz, sum of squares xyz, count}; non-finite
// coordinates are ignored
#define BB_W 13
struct ClipBox { double lo[3], hi[3]; int on; };
__global__ __launch_bounds__(RED_BLOCK) void bbox_partial(int n, const double* x, const double* y,
                                                          const double* z, ClipBox clip, double* partial) {
    __shared__ double sm[RED_BLOCK / 64][BB_W];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double su[3] = {0.0, 0.0, 0.0}, sq[3] = {0.0, 0.0, 0.0}, cnt = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double v[3] = {x[i], y[i], z[i]};
        if (isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2])) {
            // min/max over everything finite; mean and variance only over the clip box (the region
            // the cloud occupied last time, doubled) so that escapers cannot drag the grid with them
            bool in = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mn[c] = fmin(mn[c], v[c]); mx[c] = fmax(mx[c], v[c]);
                if (clip.on && (v[c] < clip.lo[c] || v[c] > clip.hi[c])) in = false;
            }
            if (in) {
                cnt += 1.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) { su[c] += v[c]; sq[c] += v[c] * v[c]; }
            }
        }
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); su[c] = wave_sum(su[c]); sq[c] = wave_sum(sq[c]);
    }
    cnt = wave_sum(cnt);
    if (lane == 0) {
        for (int c = 0; c < 3; ++c) {
            sm[wave][c] = mn[c]; sm[wave][3 + c] = mx[c]; sm[wave][6 + c] = su[c]; sm[wave][9 + c] = sq[c];
        }
        sm[wave][12] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < BB_W) {
        const int c = threadIdx.x;
        double v = sm[0][c];
        for (int w = 1; w < RED_BLOCK / 64; ++w)
            v = c < 3 ? fmin(v, sm[w][c]) : (c < 6 ? fmax(v, sm[w][c]) : v + sm[w][c]);
        partial[blockIdx.x * BB_W + c] = v;
    }
}
// one block per component: block c reduces partial[:, c] (four waves, then their four results in a fixed order)
__global__ __launch_bounds__(256) void bbox_final(int nblocks, const double* partial, double* out) {
    __shared__ double sw[4];
    const int c = blockIdx.x;
    double v = c < 3 ? INFINITY : (c < 6 ? -INFINITY : 0.0);
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        const double p = partial[b * BB_W + c];
        v = c < 3 ? fmin(v, p) : (c < 6 ? fmax(v, p) : v + p);
    }
    v = c < 3 ? wave_min(v) : (c < 6 ? wave_max(v) : wave_sum(v));
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a0 = sw[0], a1 = sw[1], a2 = sw[2], a3 = sw[3];
        out[c] = c < 3 ? fmin(fmin(a0, a1), fmin(a2, a3)) : (c < 6 ? fmax(fmax(a0, a1), fmax(a2, a3)) : (a0 + a1) + (a2 + a3));
    }
}

// out_minmax[0..5] = bounding box of all finite points; [6..8] mean, [9..11] standard deviation,
// [12] count - the last three over the context's clip box when it is valid (use_clip)
// launch the reduction over the current positions and its copy to `host_dst` (pinned); no wait
static int bbox_launch(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z, bool use_clip,
                       void* host_dst) {
    ClipBox clip;
    clip.on = (use_clip && ctx->clip_valid) ? 1 : 0;
    for (int c = 0; c < 3; ++c) { clip.lo[c] = ctx->clip_lo[c]; clip.hi[c] = ctx->clip_hi[c]; }
    int blocks = (int)((n + RED_BLOCK - 1) / RED_BLOCK);
    if (blocks > RED_MAXBLOCKS) blocks = RED_MAXBLOCKS;
    SPHX_TRY(sphx_ensure(ctx, ctx->bbox_tmp, (size_t)(RED_MAXBLOCKS + 2 + FUSED_MAXBLOCKS) * BB_W * sizeof(double)));
    double* part = ctx->bbox_tmp.as<double>();
    double* fin = part + (size_t)RED_MAXBLOCKS * BB_W;
    hipLaunchKernelGGL(bbox_partial, dim3(blocks), dim3(RED_BLOCK), 0, ctx->stream, (int)n, x, y, z, clip, part);
    hipLaunchKernelGGL(bbox_final, dim3(BB_W), dim3(256), 0, ctx->stream, blocks, part, fin);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_dst, fin, BB_W * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return SPHX_OK;
}
// sums -> mean and standard deviation
static void bbox_finish(const void* host_src, double out_minmax[13]) {
    memcpy(out_minmax, host_src, BB_W * sizeof(double));
    const double cnt = out_minmax[12] > 0.0 ? out_minmax[12] : 1.0;
    for (int c = 0; c < 3; ++c) {
        const double mean = out_minmax[6 + c] / cnt;
        double var = out_minmax[9 + c] / cnt - mean * mean;
        out_minmax[6 + c] = mean;
        out_minmax[9 + c] = var > 0.0 ? sqrt(var) : 0.0;
    }
}
int sphx_bbox(sphx_ctx* ctx, int64_t n, const double* x, const double* y, const double* z,
              double out_minmax[13], bool use_clip) {
    SPHX_TRY(bbox_launch(ctx, n, x, y, z, use_clip, ctx->pinned));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    bbox_finish(ctx->pinned, out_minmax);
    return SPHX_OK;
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_coord_g(double v, double vmin, double inv_cell, int nmax1) {
    double t = (v - vmin) * inv_cell;
    t = fmin(fmax(t, 0.0), (double)nmax1);
    return (int)t;
}

__global__ __launch_bounds__(256) void cell_count(int n, const double* x, const double* y,
                                                  const double* z, GridParams g, int* cell_of,
                                                  int* hist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cx = cell_coord_g(x[i], g.xmin, g.inv_cell, g.nx - 1);
    int cy = cell_coord_g(y[i], g.ymin, g.inv_cell, g.ny - 1);
    int cz = cell_coord_g(z[i], g.zmin, g.inv_cell, g.nz - 1);
    int c = (cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    atomicAdd(&hist[c], 1);
}

// ---- the fused loop's first pass over the particles: drv:233-238 clamp (optional) + bounding-box statistics +
// cell ids and histogram in ONE kernel (was: clamp_kernel, bbox_partial, bbox_final, cell_count).  Possible
// because the fused loop sizes its grid from the PREVIOUS step's statistics: g is known before this step's
// positions have been looked at.  bbox_final then reduces the block partials as before (a last-block-done ticket
// inside this kernel was tried: its release fence writes back every dirty line the histogram atomics and the
// clamped positions left in the L2 - 441 us instead of 60).
struct FusedCountArgs {
    int n;
    double *x, *y, *z, *vx, *vy, *vz;      // vx == nullptr: no clamp
    double lim;
    ClipBox clip;
    GridParams g;
    int* cell_of;
    int* hist;
    double* partial;
    unsigned* ticket;
    double* fin;
    u64* ct_reset;                          // "no crossing-time vote yet" for this step's pass 2 (nullable)
    int* zero_int;                          // the search's fail-list counter, zeroed here instead of by a launch of its own (nullable)
    int* rank;                              // (nullable) the particle's arrival number in its cell: cell_scatter then needs no atomic
};
__device__ __forceinline__ double nan_to_num_g(double v) {
    if (v != v) return 0.0;
    if (v > DBL_MAX) return DBL_MAX;
    if (v < -DBL_MAX) return -DBL_MAX;
    return v;
}
__global__ __launch_bounds__(RED_BLOCK) void grid_count_fused(FusedCountArgs a) {
    __shared__ double sm[RED_BLOCK / 64][BB_W];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double su[3] = {0.0, 0.0, 0.0}, sq[3] = {0.0, 0.0, 0.0}, cnt = 0.0;
    const GridParams g = a.g;
    if (a.ct_reset && blockIdx.x == 0 && threadIdx.x == 0) *a.ct_reset = SPHX_CT_NONE;
    if (a.zero_int && blockIdx.x == 0 && threadIdx.x == 0) { a.zero_int[0] = 0; a.zero_int[1] = 0; }   // fail count, tie count
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
        double v[3] = {a.x[i], a.y[i], a.z[i]};
        if (a.vx) {                                            // drv:233-238
            const double o0 = v[0], o1 = v[1], o2 = v[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double q = v[c];
                q = (q > a.lim) ? a.lim : q;
                q = (q < -a.lim) ? -a.lim : q;
                v[c] = nan_to_num_g(q);
            }
            // (written back only where the guard changed something - bit patterns compared, NaN included: in a sane state
            //  that is nowhere, and 48 MB of stores per step at 1e6 particles stay undone)
            if (__double_as_longlong(v[0]) != __double_as_longlong(o0)) a.x[i] = v[0];
            if (__double_as_longlong(v[1]) != __double_as_longlong(o1)) a.y[i] = v[1];
            if (__double_as_longlong(v[2]) != __double_as_longlong(o2)) a.z[i] = v[2];
            const double w0 = a.vx[i], w1 = a.vy[i], w2 = a.vz[i];
            const double u0 = nan_to_num_g(w0), u1 = nan_to_num_g(w1), u2 = nan_to_num_g(w2);
            if (__double_as_longlong(u0) != __double_as_longlong(w0)) a.vx[i] = u0;
            if (__double_as_longlong(u1) != __double_as_longlong(w1)) a.vy[i] = u1;
            if (__double_as_longlong(u2) != __double_as_longlong(w2)) a.vz[i] = u2;
        }
        if (isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2])) {
            bool in = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                mn[c] = fmin(mn[c], v[c]); mx[c] = fmax(mx[c], v[c]);
                if (a.clip.on && (v[c] < a.clip.lo[c] || v[c] > a.clip.hi[c])) in = false;
            }
            if (in) {
                cnt += 1.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) { su[c] += v[c]; sq[c] += v[c] * v[c]; }
            }
        }
        const int cx = cell_coord_g(v[0], g.xmin, g.inv_cell, g.nx - 1);
        const int cy = cell_coord_g(v[1], g.ymin, g.inv_cell, g.ny - 1);
        const int cz = cell_coord_g(v[2], g.zmin, g.inv_cell, g.nz - 1);
        const int c = (cz * g.ny + cy) * g.nx + cx;
        a.cell_of[i] = c;
        if (a.rank) a.rank[i] = atomicAdd(&a.hist[c], 1);
        else atomicAdd(&a.hist[c], 1);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); su[c] = wave_sum(su[c]); sq[c] = wave_sum(sq[c]);
    }
    cnt = wave_sum(cnt);
    if (lane == 0) {
        for (int c = 0; c < 3; ++c) {
            sm[wave][c] = mn[c]; sm[wave][3 + c] = mx[c]; sm[wave][6 + c] = su[c]; sm[wave][9 + c] = sq[c];
        }
        sm[wave][12] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < BB_W) {
        const int c = threadIdx.x;
        double v = sm[0][c];
        for (int w = 1; w < RED_BLOCK / 64; ++w)
            v = c < 3 ? fmin(v, sm[w][c]) : (c < 6 ? fmax(v, sm[w][c]) : v + sm[w][c]);
        a.partial[blockIdx.x * BB_W + c] = v;
    }
}


// three-phase exclusive scan of int32: 2048 items per block
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

__device__ __forceinline__ int block_exclusive_scan(int v, int* total_out) {
    // returns exclusive prefix of v across the 256-thread block; *total_out = block total
    __shared__ int wsum[SCAN_BLOCK / 64];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        int s = wsum[w];
        if (w < wave) woff += s;
        tot += s;
    }
    __syncthreads();
    *total_out = tot;
    return woff + incl - v;
}

// ---- exclusive scan of n ints, out[n] = total (in[n] must be 0): ONE launch ------------------------------------------
// Single-pass scan.  A tile of 8192 items publishes its sum, adds up the sums of all earlier tiles (a few hundred words, a
// few per thread) and writes its items: a tile waits for sums only, which no tile waits to publish, and the launcher
// keeps the grid small enough for every tile to be resident at once (<= LBS_MAXTILES; larger scans take rocPRIM's) - so
// there is no chain from tile to tile, no ticket and no order to keep.  A word is {epoch x 4 + 1, sum} in 64 bits, read
// and written whole at agent scope: words of earlier launches carry another epoch and read as "not yet", so nothing is
// reset between launches (rocPRIM's look-back scan orders its tiles by a ticket on one address and resets its tile states
// with a launch of its own: 6 us + 13 us per scan at 2.4e6 cells, twice per step).
#define LBS_ITEMS 32
#define LBS_TILE (LBS_ITEMS * 256)
#define LBS_MAXTILES 512           // 256 CUs x 8 workgroups of 256 threads fit at once: a quarter of that, so that four
                                   // processes sharing a device (rehearsals of several ranks on one GPU) still fit together
// ZERO: the items are put back to zero behind the read (the cell histogram: counted up by the count kernel, all zero
// between builds)
template <int ZERO>
__global__ __launch_bounds__(256) void lookback_scan_kernel(int n_items, int* __restrict__ in, int* __restrict__ out,
                                                            u64* state, int* ctr, unsigned epoch, int ntiles) {
    __shared__ int s_prefix;
    const int tile = blockIdx.x;          // (every tile of a launch is resident at once - the launcher sees to it - so any order will do)
    const int base = tile * LBS_TILE + threadIdx.x * LBS_ITEMS;
    int v[LBS_ITEMS];
    int s = 0;
    if (base + LBS_ITEMS <= n_items) {
#pragma unroll
        for (int q = 0; q < LBS_ITEMS; q += 4) {
            const int4 t = *reinterpret_cast<const int4*>(in + base + q);
            v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w;
            if (ZERO) *reinterpret_cast<int4*>(in + base + q) = make_int4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int q = 0; q < LBS_ITEMS; ++q) {
            v[q] = (base + q < n_items) ? in[base + q] : 0;
            if (ZERO && base + q < n_items) in[base + q] = 0;
        }
    }
#pragma unroll
    for (int q = 0; q < LBS_ITEMS; ++q) s += v[q];
    int tot;
    int ex = block_exclusive_scan(s, &tot);
    // publish the tile's sum, then add up the sums of ALL earlier tiles (each thread a few of them: a tile waits for sums
    // only, which no tile waits to publish - no chain from tile to tile)
    const u64 tag = ((u64)epoch << 2) | 1ull;
    if (threadIdx.x == 0) __hip_atomic_store(&state[tile], (tag << 32) | (u32)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int part = 0;
    for (int j = threadIdx.x; j < tile; j += 256) {
        for (unsigned spins = 0;;) {
            const u64 w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 32) == tag) { part += (int)(u32)w; break; }
            // (tile j took its ticket before this one: it is running and publishes without waiting for anybody; the bound
            //  is there so that a broken invariant shows as wrong numbers and a raised flag, ctr[2], not as a hung device)
            if (++spins > (1u << 22)) { ctr[2] = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    part = wave_sum(part);
    __shared__ int s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) s_prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    ex += s_prefix;
    if (base + LBS_ITEMS <= n_items) {
#pragma unroll
        for (int q = 0; q < LBS_ITEMS; q += 4) {
            int4 t;
            t.x = ex; ex += v[q]; t.y = ex; ex += v[q + 1]; t.z = ex; ex += v[q + 2]; t.w = ex; ex += v[q + 3];
            *reinterpret_cast<int4*>(out + base + q) = t;
        }
    } else {
#pragma unroll
        for (int q = 0; q < LBS_ITEMS; ++q) { if (base + q < n_items) out[base + q] = ex; ex += v[q]; }
    }
}
// zero_in: the items are put back to zero once read (*zeroed says whether that happened: rocPRIM's scan does not)
static int excl_scan_plus_total(sphx_ctx* ctx, const int* in, int* out, int n, bool zero_in = false, bool* zeroed = nullptr) {
    if (zeroed) *zeroed = false;
    if (ctx->scan_rocprim || (((uintptr_t)in | (uintptr_t)out) & 15)) {
        size_t bytes = 0;
        HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)n + 1, rocprim::plus<int>(), ctx->stream));
        SPHX_TRY(sphx_ensure(ctx, ctx->scan_tmp, bytes + 64));
        HIPCHK(rocprim::exclusive_scan(ctx->scan_tmp.p, bytes, in, out, 0, (size_t)n + 1, rocprim::plus<int>(), ctx->stream));
        return SPHX_OK;
    }
    const int n_items = n + 1;
    const int ntiles = (n_items + LBS_TILE - 1) / LBS_TILE;
    if (ntiles > LBS_MAXTILES) {
        size_t bytes = 0;
        HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)n + 1, rocprim::plus<int>(), ctx->stream));
        SPHX_TRY(sphx_ensure(ctx, ctx->scan_tmp, bytes + 64));
        HIPCHK(rocprim::exclusive_scan(ctx->scan_tmp.p, bytes, in, out, 0, (size_t)n + 1, rocprim::plus<int>(), ctx->stream));
        return SPHX_OK;
    }
    // (one set of tile words per stream the scans are launched on: launches on ONE stream are ordered, two streams are not)
    const int which = (ctx->stream == ctx->own_stream) ? 0 : 1;
    DevBuf& st = ctx->lbs_state[which];
    const size_t need = ((size_t)LBS_MAXTILES + 8) * sizeof(u64);          // (one size for every scan: nothing to allocate later, e.g. while a step graph is recorded)
    if (st.cap < need) {
        SPHX_TRY(sphx_ensure(ctx, st, need));
        HIPCHK(hipMemsetAsync(st.p, 0, st.cap, ctx->stream));
    }
    // (a recorded launch is replayed with the epoch it was recorded with: its words are cleared by a node of the graph)
    if (ctx->capturing) HIPCHK(hipMemsetAsync(st.as<u64>() + 2, 0, (size_t)ntiles * sizeof(u64), ctx->stream));
    unsigned& ep = ctx->lbs_epoch[which];
    ep = (ep + 1u) & 0x3FFFFFFFu;
    if (ep == 0u) ep = 1u;
    if (zero_in) {
        hipLaunchKernelGGL(lookback_scan_kernel<1>, dim3(ntiles), dim3(256), 0, ctx->stream, n_items, const_cast<int*>(in), out,
                           st.as<u64>() + 2, st.as<int>(), ep, ntiles);
        if (zeroed) *zeroed = true;
    } else {
        hipLaunchKernelGGL(lookback_scan_kernel<0>, dim3(ntiles), dim3(256), 0, ctx->stream, n_items, const_cast<int*>(in), out,
                           st.as<u64>() + 2, st.as<int>(), ep, ntiles);
    }
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

int sphx_excl_scan_int(sphx_ctx* ctx, const int* in, int* out, int n) { return excl_scan_plus_total(ctx, in, out, n); }

// ---- self-test of the single-launch scan against rocPRIM's on the device (tests/test_gpu_parity.py) ----
__global__ void scan_selftest_fill(int n, unsigned seed, int* in) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        unsigned v = (unsigned)i * 2654435761u + seed * 40503u;
        v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
        in[i] = (i < n) ? (int)(v & 7u) : 0;                  // (in[n] = 0: the scan's contract)
    }
}
__global__ void scan_selftest_cmp(int n, const int* a, const int* b, unsigned long long* bad) {
    unsigned long long c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) c += (a[i] != b[i]) ? 1ull : 0ull;
    if (c) atomicAdd(bad, c);
}
extern "C" int sphx_selftest_scan(sphx_ctx* ctx, int n, unsigned seed, long long* mismatches, int* single_launch) {
    if (!ctx || !mismatches || n < 1) return sphx_set_err(ctx, SPHX_E_ARG, "sphx_selftest_scan: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf in, o1, o2, bad;
    int rc = SPHX_OK;
    do {
        if ((rc = sphx_ensure(ctx, in, ((size_t)n + 8) * sizeof(int))) != SPHX_OK) break;
        if ((rc = sphx_ensure(ctx, o1, ((size_t)n + 8) * sizeof(int))) != SPHX_OK) break;
        if ((rc = sphx_ensure(ctx, o2, ((size_t)n + 8) * sizeof(int))) != SPHX_OK) break;
        if ((rc = sphx_ensure(ctx, bad, 64)) != SPHX_OK) break;
        hipLaunchKernelGGL(scan_selftest_fill, dim3(1024), dim3(256), 0, ctx->stream, n, seed, in.as<int>());
        if (hipMemsetAsync(bad.p, 0, 8, ctx->stream) != hipSuccess) { rc = SPHX_E_HIP; break; }
        const bool keep = ctx->scan_rocprim;
        ctx->scan_rocprim = false;
        rc = excl_scan_plus_total(ctx, in.as<int>(), o1.as<int>(), n);
        ctx->scan_rocprim = true;
        if (rc == SPHX_OK) rc = excl_scan_plus_total(ctx, in.as<int>(), o2.as<int>(), n);
        ctx->scan_rocprim = keep;
        if (rc != SPHX_OK) break;
        hipLaunchKernelGGL(scan_selftest_cmp, dim3(1024), dim3(256), 0, ctx->stream, n, o1.as<int>(), o2.as<int>(),
                           bad.as<unsigned long long>());
        unsigned long long h = 0;
        if (hipMemcpyAsync(&h, bad.p, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = SPHX_E_HIP; break; }
        *mismatches = (long long)h;
        if (single_launch) *single_launch = ((n + 1 + LBS_TILE - 1) / LBS_TILE <= LBS_MAXTILES) ? 1 : 0;
    } while (false);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (in.p) (void)hipFree(in.p);
    if (o1.p) (void)hipFree(o1.p);
    if (o2.p) (void)hipFree(o2.p);
    if (bad.p) (void)hipFree(bad.p);
    return rc;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_phase1(int n, const int* in, int* block_sums) {
    int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) s += (base + q < n) ? in[base + q] : 0;
    int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// single block: exclusive scan of up to 4096 block sums, in place
__global__ __launch_bounds__(SCAN_BLOCK) void scan_phase2(int nb, int* block_sums, int* grand_total) {
    int carry = 0;
    for (int base = 0; base < nb; base += SCAN_BLOCK) {
        int i = base + threadIdx.x;
        int v = i < nb ? block_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
// out[i] = exclusive prefix; also out[n] = total (written by the last block)
__global__ __launch_bounds__(SCAN_BLOCK) void scan_phase3(int n, const int* in, const int* block_sums,
                                                          int* out) {
    int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) { v[q] = (base + q < n) ? in[base + q] : 0; s += v[q]; }
    int tot;
    int ex = block_exclusive_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; ++q) {
        if (base + q < n) out[base + q] = ex;
        ex += v[q];
        if (base + q == n - 1) out[n] = ex;
    }
}

// bb_part != nullptr (the fused loop): the first BB_W blocks also fold one column each of the box statistics' block
// partials (what bbox_final does: same order of operations) - they are wanted by the search and by the next step's host code,
// not by anything before this kernel, so they need no launch of their own on the way (early blocks: off the kernel's tail).
__global__ __launch_bounds__(256) void cell_scatter(int n, const int* cell_of, const int* cell_start,
                                                    int* fill, int* perm, int bb_nblocks, const double* bb_part,
                                                    double* bb_out, double* bb_host, const int* __restrict__ rank) {
    __shared__ double sw[4];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && rank) {
        // (the count kernel kept every particle's arrival number: no second atomic per particle; the histogram was put
        //  back to zero by the scan that read it)
        perm[cell_start[cell_of[i]] + rank[i]] = i;
    } else if (i < n) {
        int c = cell_of[i];
        // counts the histogram back down to zero; slots are handed out upwards (arrival order is mostly
        // the previous cell order already, which the per-cell insertion sort then finds nearly sorted)
        int slot = cell_start[c + 1] - atomicSub(&fill[c], 1);
        perm[slot] = i;
    }
    if (bb_part && (int)blockIdx.x < BB_W && (int)gridDim.x >= BB_W) {
        const int c = blockIdx.x;
        double v = c < 3 ? INFINITY : (c < 6 ? -INFINITY : 0.0);
        for (int b = threadIdx.x; b < bb_nblocks; b += 256) {
            const double p = bb_part[b * BB_W + c];
            v = c < 3 ? fmin(v, p) : (c < 6 ? fmax(v, p) : v + p);
        }
        v = c < 3 ? wave_min(v) : (c < 6 ? wave_max(v) : wave_sum(v));
        if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double a0 = sw[0], a1 = sw[1], a2 = sw[2], a3 = sw[3];
            const double r_ = c < 3 ? fmin(fmin(a0, a1), fmin(a2, a3)) : (c < 6 ? fmax(fmax(a0, a1), fmax(a2, a3)) : (a0 + a1) + (a2 + a3));
            bb_out[c] = r_;
            if (bb_host) bb_host[c] = r_;          // (pinned host memory, read by the next step's host code behind an event: no copy launch)
        }
    }
}

// The atomic scatter fills a cell in arrival order; sorting each cell's slice by the particles'
// previous index makes the cell-sorted order (and everything derived from it: tie-breaks in the
// search, summation order) identical from run to run.  Cells hold a few particles each.
// A cell's members into ascending order, one lane per cell - for the few particles a cell normally holds.  A diverging
// run piles hundreds of escapers into single boundary cells (the corners of the drv:233 clamp cube: identical positions);
// one lane's insertion sort then costs cnt^2 steps and the grid build grew from 0.2 to 4.5 ms over 300 steps of such a run.
// Cells of 17 .. 512 members are therefore sorted by the whole WAVE: every lane holds up to eight members, finds each
// one's rank by comparing it with all of them (broadcast lane by lane: members are distinct indices) and writes it to
// its place - all reads before any write.  Beyond 512 the arrival order stays (a valid cell list; only run-to-run
// tie-breaking is lost).  Every lane of the wave must call this (cnt = 0: nothing to do).
#define CELL_SORT_SERIAL 16
#define CELL_SORT_WAVE 512
#define DENSE_CELL 48              // members from which a cell counts as dense (27 of them: 1300, a tile holds 1408)
#define CROWDED_LIST_MIN 2048      // crowded cells at the last build the host knows of, from which they get a launch of their own
// one crowded cell (m members from perm[bs]) by the whole wave: U = members per lane
template <int U>
__device__ __forceinline__ void sort_crowded_cell(int* perm, int bs, int m, int lane) {
    int v[U], rank[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = lane + 64 * u;
        v[u] = q < m ? perm[bs + q] : 0x7FFFFFFF;
        rank[u] = 0;
    }
#pragma unroll
    for (int c2 = 0; c2 < U; ++c2) {
        const int lc = m - 64 * c2 < 64 ? m - 64 * c2 : 64;
        for (int tt = 0; tt < lc; ++tt) {
            const int kt = __builtin_amdgcn_readlane(v[c2], tt);
#pragma unroll
            for (int u = 0; u < U; ++u) rank[u] += kt < v[u] ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (lane + 64 * u < m) perm[bs + rank[u]] = v[u];
}
// (the comparisons go as members x members-per-lane: a cell of 40 is not charged for a cell of 512)
__device__ __forceinline__ void sort_crowded_cell_any(int* perm, int bs, int m, int lane) {
    if (m <= 64) sort_crowded_cell<1>(perm, bs, m, lane);
    else if (m <= 128) sort_crowded_cell<2>(perm, bs, m, lane);
    else if (m <= 256) sort_crowded_cell<4>(perm, bs, m, lane);
    else sort_crowded_cell<8>(perm, bs, m, lane);
}
__device__ __forceinline__ void sort_cell_members(int* perm, int s, int cnt, bool on) {
    if (on && cnt > 1 && cnt <= CELL_SORT_SERIAL) {
        for (int i = s + 1; i < s + cnt; ++i) {
            const int v = perm[i];
            int j = i - 1;
            while (j >= s && perm[j] > v) { perm[j + 1] = perm[j]; --j; }
            perm[j + 1] = v;
        }
    }
    u64 big = __builtin_amdgcn_ballot_w64(on && cnt > CELL_SORT_SERIAL && cnt <= CELL_SORT_WAVE);
    const int lane = threadIdx.x & 63;
    while (big) {
        const int t = __builtin_ctzll(big);
        big &= big - 1;
        const int bs = __builtin_amdgcn_readlane(s, t), m = __builtin_amdgcn_readlane(cnt, t);
        sort_crowded_cell_any(perm, bs, m, lane);
    }
}
__global__ __launch_bounds__(256) void cell_sort_members(int ncells, const int* cell_start, int* perm) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int s = 0, cnt = 0;
    if (c < ncells) { s = cell_start[c]; cnt = cell_start[c + 1] - s; }
    sort_cell_members(perm, s, cnt, true);
}

#define SPHX_MAX_CELLS (SCAN_TILE * 4096)

int sphx_build_grid(sphx_ctx* ctx, int64_t n, int k, const double* x, const double* y,
                    const double* z, double cell_hint) {
    double bb[13];
    // the statistics window of the robust box as the previous build left it (this build moves it)
    ClipBox clip0;
    clip0.on = ctx->clip_valid ? 1 : 0;
    for (int c = 0; c < 3; ++c) { clip0.lo[c] = ctx->clip_lo[c]; clip0.hi[c] = ctx->clip_hi[c]; }
    bool fused = false;           // this step's statistics come out of the cell-count kernel itself
    int lag_cur = 0;
    ctx->grid_fused = false;
    if (!ctx->in_fused_step) sphx_graph_drop(ctx);      // (an array call or the device API between the loop's steps: its grid, its box statistics)
    if (ctx->capturing) {
        // recorded into a step graph: the box statistics the last real step sized its grid from, no host wait; this step's
        // own statistics still come out of the count kernel (tbox, on the device), but are not copied out
        if (!ctx->lag_on || !ctx->fuse_count) return sphx_set_err(ctx, SPHX_E_STATE, "only the fused loop's grid build can be captured");
        for (int q = 0; q < 13; ++q) bb[q] = ctx->cap_bb[q];
        fused = true;
        ctx->grid_fused = true;
    } else if (ctx->lag_on) {
        // Fused step loop: this step's statistics are launched and copied out, the grid is sized from
        // the previous step's (already on the host) - the loop never waits for the step it launches.
        // One step of motion makes the box slightly stale, which only steers performance: particles
        // outside it are clamped into the boundary cells.  (tbox, the device-side true bounding box
        // the search reads, is this step's.)
        const int prev = ctx->lag_bslot, cur = prev ^ 1;
        char* slot = (char*)ctx->pinned + LAG_OFF;
        // (the previous statistics describe this cloud if they were taken over about as many particles: the
        //  fused loop's n is constant, the decomposed driver's owned + ghost count wobbles by a few per cent)
        const int64_t pn = ctx->lag_bn[prev];
        const int use = (ctx->lag_bvalid[prev] && pn > 0 && (n > pn ? n - pn : pn - n) * 4 <= n) ? prev : cur;
        lag_cur = cur;
        fused = (use == prev) && ctx->fuse_count;
        if (!fused) {
            if (ctx->clamp_vx) { SPHX_TRY(sphx_clamp(ctx, n, ctx->st)); ctx->clamp_vx = nullptr; }
            SPHX_TRY(bbox_launch(ctx, n, x, y, z, true, slot + 512 * cur));
            HIPCHK(hipEventRecord(ctx->lag_bev[cur], ctx->stream));
            ctx->lag_balias[cur] = nullptr;
        }
        HIPCHK(hipEventSynchronize(ctx->lag_balias[use] ? ctx->lag_balias[use] : ctx->lag_bev[use]));
        bbox_finish(slot + 512 * use, bb);
        ctx->lag_bvalid[cur] = true;
        ctx->lag_bn[cur] = n;
        ctx->lag_bslot = cur;
        ctx->grid_fused = fused;
    } else {
        if (ctx->clamp_vx) { SPHX_TRY(sphx_clamp(ctx, n, ctx->st)); ctx->clamp_vx = nullptr; }
        SPHX_TRY(sphx_bbox(ctx, n, x, y, z, bb, true));
    }
    if (!ctx->capturing && ctx->clip_valid && bb[12] < 0.5 * (double)n) {     // the clip box lost the cloud: re-anchor
        if (ctx->clamp_vx) { SPHX_TRY(sphx_clamp(ctx, n, ctx->st)); ctx->clamp_vx = nullptr; }
        SPHX_TRY(sphx_bbox(ctx, n, x, y, z, bb, false));
        ctx->grid_fused = false;                // (a host wait: not a step to replay)
    }
    if (!ctx->capturing)
        for (int q = 0; q < 13; ++q) ctx->cap_bb[q] = bb[q];     // (what a captured step sizes its grid from)
    double tmin[3], tmax[3];
    for (int c = 0; c < 3; ++c) {
        if (!(bb[3 + c] >= bb[c])) { bb[c] = 0.0; bb[3 + c] = 0.0; }   // no finite coordinate
        tmin[c] = bb[c]; tmax[c] = bb[3 + c];
        // Robust box: a few escaped particles (the reference lets them reach 1e11 AU, drv:233) must
        // not stretch the grid over empty space.  The grid covers mean +- 3 sigma (box_sigmas: a
        // uniform sphere ends at 2.2 sigma, a uniform cube at 1.7); anything beyond -- the diffuse
        // halo of an expanding cloud -- is clamped into the boundary cells, which the search handles
        // exactly (half-infinite boundary cells).  With 8 sigma a freely expanding 1e6 polytrope
        // went from 2.25 to 3.74 ms/step in 400 steps (cells at the cap, 100-cell-wide halo
        // searches); with 3 it goes from 2.10 to 2.42.
        const double lo = bb[6 + c] - ctx->box_sigmas * bb[9 + c], hi = bb[6 + c] + ctx->box_sigmas * bb[9 + c];
        if (bb[9 + c] > 0.0 && hi > lo) {
            if (bb[c] < lo) bb[c] = lo;
            if (bb[3 + c] > hi) bb[3 + c] = hi;
        }
    }
    double L[3];
    double Lmax = 0.0;
    for (int c = 0; c < 3; ++c) { L[c] = bb[3 + c] - bb[c]; if (L[c] > Lmax) Lmax = L[c]; }
    // next time, statistics are taken over this box doubled about its centre
    for (int c = 0; c < 3; ++c) {
        const double mid = 0.5 * (bb[c] + bb[3 + c]), half = (L[c] > 0.0 ? L[c] : Lmax);
        ctx->clip_lo[c] = mid - half;
        ctx->clip_hi[c] = mid + half;
    }
    ctx->clip_valid = (Lmax > 0.0);
    if (!(Lmax > 0.0)) Lmax = 1.0;
    double cell = cell_hint;
    if (!(cell > 0.0) || !isfinite(cell)) {
        // mean kNN radius of a uniform fill of the box, times cell_factor
        double V = 1.0;
        for (int c = 0; c < 3; ++c) V *= (L[c] > 1e-6 * Lmax ? L[c] : 1e-6 * Lmax);
        cell = ctx->cell_factor * cbrt(V * (double)k / ((double)n * 4.1887902047863905));
    }
    if (cell < Lmax * 1e-4) cell = Lmax * 1e-4;
    int64_t cap = 32 * n + 1024;      // cells per particle the grid may use when a diffuse halo stretches the box
    if (cap > SPHX_MAX_CELLS) cap = SPHX_MAX_CELLS;
    if (ctx->max_cells > 0 && cap > ctx->max_cells) cap = ctx->max_cells;
    if (ctx->max_cells < 0) { cap = 32 * n + 1024; if (cap > -ctx->max_cells) cap = -ctx->max_cells; }   // experiment: beyond the default limit
    int nx, ny, nz;
    for (;;) {
        nx = (int)fmin(floor(L[0] / cell) + 1.0, 2047.0);
        ny = (int)fmin(floor(L[1] / cell) + 1.0, 2047.0);
        nz = (int)fmin(floor(L[2] / cell) + 1.0, 2047.0);
        int64_t tot = (int64_t)nx * ny * nz;
        if (tot <= cap) break;
        cell *= 1.02 * cbrt((double)tot / (double)cap);
    }
    GridParams& g = ctx->grid;
    for (int c = 0; c < 3; ++c) { ctx->tbox_h[c] = tmin[c]; ctx->tbox_h[3 + c] = tmax[c]; }
    ctx->olev.L = 0;                   // (outlier levels belong to one grid: rebuilt on demand, sphx_build_outlier_levels)
    ctx->tbox = ctx->bbox_tmp.as<double>() + (size_t)RED_MAXBLOCKS * BB_W;   // bbox_final's min/max
    g.xmin = bb[0]; g.ymin = bb[1]; g.zmin = bb[2];
    g.cell = cell; g.inv_cell = 1.0 / cell;
    g.nx = nx; g.ny = ny; g.nz = nz;
    g.ncells = nx * ny * nz;
    g.fnx1 = (float)(nx - 1); g.fny1 = (float)(ny - 1); g.fnz1 = (float)(nz - 1);
    ctx->stats.cells = g.ncells;
    ctx->stats.cell_size = cell;

    const int nc = g.ncells;
    SPHX_TRY(sphx_ensure(ctx, ctx->cell_of, (size_t)n * sizeof(int)));
    SPHX_TRY(sphx_ensure(ctx, ctx->perm, (size_t)n * sizeof(int)));
    SPHX_TRY(sphx_ensure(ctx, ctx->cell_start, ((size_t)nc + 2) * sizeof(int)));
    const size_t fill_cap0 = ctx->cell_fill.cap;
    SPHX_TRY(sphx_ensure(ctx, ctx->cell_fill, ((size_t)nc + 2) * sizeof(int)));
    if (ctx->cell_fill.cap != fill_cap0) ctx->cell_fill_zeroed = nullptr;       // a new allocation
    int nblk = (nc + SCAN_TILE - 1) / SCAN_TILE;
    SPHX_TRY(sphx_ensure(ctx, ctx->scan_tmp, ((size_t)nblk + 2) * sizeof(int)));
    int* fill = ctx->cell_fill.as<int>();
    int* start = ctx->cell_start.as<int>();
    int* bsum = ctx->scan_tmp.as<int>();
    // The histogram array is all zero between builds: cell_count counts it up, cell_scatter counts it
    // back down while handing out slots (a cell's members are sorted afterwards anyway) - no memsets,
    // except once for a new allocation.
    if (ctx->cell_fill_zeroed != ctx->cell_fill.p) {
        HIPCHK(hipMemsetAsync(fill, 0, ctx->cell_fill.cap, ctx->stream));
        ctx->cell_fill_zeroed = ctx->cell_fill.p;
    }
    int pb = (int)((n + 255) / 256);
    (void)bsum;
    int bb_fold_blocks = 0;
    const int* rank_dev = nullptr;            // set: the count kernel kept the particles' arrival numbers
    const double* bb_fold_part = nullptr;
    double* bb_fold_out = nullptr;
    if (fused) {
        // clamp (when the step asked for it) + this step's box statistics + cell ids + histogram: one pass
        const bool fresh = ctx->bbox_tmp.p == nullptr;
        SPHX_TRY(sphx_ensure(ctx, ctx->bbox_tmp, (size_t)(RED_MAXBLOCKS + 2 + FUSED_MAXBLOCKS) * BB_W * sizeof(double)));
        double* fin = ctx->bbox_tmp.as<double>() + (size_t)RED_MAXBLOCKS * BB_W;     // (where bbox_final writes: ctx->tbox)
        unsigned* ticket = reinterpret_cast<unsigned*>(fin + BB_W);
        double* part = fin + 2 * BB_W;
        if (fresh || !ctx->bbox_ticket_zeroed) {
            HIPCHK(hipMemsetAsync(ticket, 0, sizeof(double), ctx->stream));
            ctx->bbox_ticket_zeroed = true;
        }
        FusedCountArgs fa;
        fa.n = (int)n;
        fa.x = const_cast<double*>(x); fa.y = const_cast<double*>(y); fa.z = const_cast<double*>(z);
        fa.vx = ctx->clamp_vx; fa.vy = ctx->clamp_vy; fa.vz = ctx->clamp_vz;
        fa.lim = ctx->cst.pos_clamp;
        fa.clip = clip0;
        fa.g = g;
        fa.cell_of = ctx->cell_of.as<int>();
        fa.hist = fill;
        fa.partial = part; fa.ticket = ticket; fa.fin = fin;
        fa.ct_reset = ctx->scal.as<u64>() + SC_CT_BITS;      // (read by the previous step's update, long done on this stream)
        fa.zero_int = nullptr;
        fa.rank = nullptr;
        if (ctx->scatter_by_rank) {
            SPHX_TRY(sphx_ensure(ctx, ctx->cell_rank, (size_t)n * sizeof(int)));
            fa.rank = ctx->cell_rank.as<int>();
        }
        rank_dev = fa.rank;
        if (ctx->fail_list.p && ctx->fail_list.cap >= ((size_t)sphx_pad64(n) + 64) * sizeof(int)) {
            fa.zero_int = ctx->fail_list.as<int>() + sphx_pad64(n);          // (where sphx_knn keeps the counter for this n)
            ctx->fcount_zeroed = ctx->fail_list.p;
            ctx->fcount_zeroed_n = n;
        }
        ctx->ct_primed = true;
        int fb = pb < FUSED_MAXBLOCKS ? pb : FUSED_MAXBLOCKS;
        hipLaunchKernelGGL(grid_count_fused, dim3(fb), dim3(RED_BLOCK), 0, ctx->stream, fa);
        HIPCHK(hipGetLastError());
        ctx->clamp_vx = nullptr;
        if (pb >= BB_W) {
            bb_fold_blocks = fb; bb_fold_part = part; bb_fold_out = fin;   // folded by cell_scatter's first blocks, copied out there
        } else {
            hipLaunchKernelGGL(bbox_final, dim3(BB_W), dim3(256), 0, ctx->stream, fb, part, fin);
            bb_fold_out = fin;                                             // (a handful of particles: its own launch, copied out below)
        }
    } else {
        hipLaunchKernelGGL(cell_count, dim3(pb), dim3(256), 0, ctx->stream, (int)n, x, y, z, g,
                           ctx->cell_of.as<int>(), fill);
    }
    bool hist_zeroed = false;
    SPHX_TRY(excl_scan_plus_total(ctx, fill, start, nc, rank_dev != nullptr, &hist_zeroed));
    // (the statistics folded by the scatter's first blocks go straight to the pinned slot the next step's host code reads:
    //  the 104-byte copy that followed was a launch of its own on the step's stream, 5 us + its gaps)
    const bool to_host = bb_fold_part != nullptr && bb_fold_out && !ctx->capturing && pb >= BB_W && ctx->bb_direct;
    double* bb_host = to_host ? reinterpret_cast<double*>((char*)ctx->pinned + LAG_OFF + 512 * lag_cur) : nullptr;
    hipLaunchKernelGGL(cell_scatter, dim3(pb), dim3(256), 0, ctx->stream, (int)n,
                       ctx->cell_of.as<int>(), start, fill, ctx->perm.as<int>(), bb_fold_blocks, bb_fold_part, bb_fold_out, bb_host, rank_dev);
    if (rank_dev && !hist_zeroed) HIPCHK(hipMemsetAsync(fill, 0, ((size_t)nc + 1) * sizeof(int), ctx->stream));     // (rocPRIM's scan left the counts in place)
    if (bb_fold_out && !ctx->capturing) {
        char* slot = (char*)ctx->pinned + LAG_OFF;
        if (!to_host)
            HIPCHK(hipMemcpyAsync(slot + 512 * lag_cur, bb_fold_out, BB_W * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (ctx->step_ev1) {
            ctx->lag_balias[lag_cur] = ctx->step_ev1;          // recorded by the caller a few launches on, before the search
        } else {
            HIPCHK(hipEventRecord(ctx->lag_bev[lag_cur], ctx->stream));
            ctx->lag_balias[lag_cur] = nullptr;
        }
    }
    ctx->cells_unsorted = false;
    if (ctx->defer_cell_sort) {
        ctx->cells_unsorted = true;          // sphx_build_blob_order's per-cell pass sorts the members too
    } else {
        hipLaunchKernelGGL(cell_sort_members, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, nc, start,
                           ctx->perm.as<int>());
    }
    HIPCHK(hipGetLastError());
    return SPHX_OK;
}

// ---- outlier levels (OutLevels, sphx_internal.h) ---------------------------------------------------
// level and cell of a position outside the grid box: the lowest level whose cube holds it
__device__ __forceinline__ int olev_key_of(const OutLevels& o, double x, double y, double z) {
    const double ax = fabs(x - o.cx), ay = fabs(y - o.cy), az = fabs(z - o.cz);
    const double m = fmax(fmax(ax, ay), az) / o.hmax;      // Chebyshev distance from the centre in units of hmax
    int e = 0;
    (void)frexp(m, &e);                                    // m = f 2^e, f in [0.5, 1): m <= 2^e
    int lev = (m == m && m <= 1.7e308) ? e : o.L;
    lev = lev < 1 ? 1 : (lev > o.L ? o.L : lev);
    const double W = ldexp(o.hmax, lev);                   // half-width of the level's cube
    const double ic = (double)OLEV_N / (2.0 * W);
    const double nm1 = (double)(OLEV_N - 1);
    const int cx = (int)fmin(fmax((x - (o.cx - W)) * ic, 0.0), nm1);
    const int cy = (int)fmin(fmax((y - (o.cy - W)) * ic, 0.0), nm1);
    const int cz = (int)fmin(fmax((z - (o.cz - W)) * ic, 0.0), nm1);
    return ((lev - 1) * OLEV_N + cz) * OLEV_N * OLEV_N + cy * OLEV_N + cx;
}
__global__ __launch_bounds__(256) void olev_count(int n, const double* x, const double* y, const double* z, GridParams g,
                                                  OutLevels o, int* key, int* hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double xi = x[i], yi = y[i], zi = z[i];
    int kx = -1;
    if (sphx_outside_box(g, xi, yi, zi)) {
        kx = olev_key_of(o, xi, yi, zi);
        atomicAdd(&hist[kx], 1);
    }
    key[i] = kx;
}
__global__ __launch_bounds__(256) void olev_scatter(int n, const int* key, const int* start, int* hist, int* list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int kx = key[i];
    if (kx < 0) return;
    list[start[kx] + atomicSub(&hist[kx], 1) - 1] = i;     // (arrival order: the search ranks candidates by (d^2, index))
}

// Levels for the current grid over the cell-sorted positions.  The number of levels follows the true bounding box the
// grid build knew (a step old in the fused loop: whatever has moved beyond the top cube since sits in its boundary cells).
int sphx_build_outlier_levels(sphx_ctx* ctx, int64_t n, const double* xs, const double* ys, const double* zs) {
    const GridParams& g = ctx->grid;
    OutLevels& o = ctx->olev;
    o.L = 0;
    const double ex = 0.5 * g.nx * g.cell, ey = 0.5 * g.ny * g.cell, ez = 0.5 * g.nz * g.cell;
    o.cx = g.xmin + ex; o.cy = g.ymin + ey; o.cz = g.zmin + ez;
    o.hmax = fmax(fmax(ex, ey), ez);
    if (!(o.hmax > 0.0) || !isfinite(o.hmax)) return SPHX_OK;
    double ext = 0.0;
    const double c3[3] = {o.cx, o.cy, o.cz};
    for (int c = 0; c < 3; ++c) {
        const double a = fabs(ctx->tbox_h[c] - c3[c]), b = fabs(ctx->tbox_h[3 + c] - c3[c]);
        if (isfinite(a) && a > ext) ext = a;
        if (isfinite(b) && b > ext) ext = b;
    }
    int L = 1;
    while (L < OLEV_MAX && ldexp(o.hmax, L) < 2.0 * ext) ++L;    // (one level of head-room for this step's motion)
    const size_t ncell = (size_t)L * OLEV_N * OLEV_N * OLEV_N;
    SPHX_TRY(sphx_ensure(ctx, ctx->olev_start, (ncell + 2) * sizeof(int)));
    const size_t cap0 = ctx->olev_fill.cap;
    SPHX_TRY(sphx_ensure(ctx, ctx->olev_fill, ((size_t)OLEV_MAX * OLEV_N * OLEV_N * OLEV_N + 2) * sizeof(int)));
    if (ctx->olev_fill.cap != cap0) ctx->olev_fill_zeroed = nullptr;
    SPHX_TRY(sphx_ensure(ctx, ctx->olev_list, (size_t)n * sizeof(int)));
    SPHX_TRY(sphx_ensure(ctx, ctx->olev_key, (size_t)n * sizeof(int)));
    int* fill = ctx->olev_fill.as<int>();
    if (ctx->olev_fill_zeroed != ctx->olev_fill.p) {          // counted up by olev_count, back down by olev_scatter
        HIPCHK(hipMemsetAsync(fill, 0, ctx->olev_fill.cap, ctx->stream));
        ctx->olev_fill_zeroed = ctx->olev_fill.p;
    }
    OutLevels ok = o;
    ok.L = L;
    const int pb = (int)((n + 255) / 256);
    hipLaunchKernelGGL(olev_count, dim3(pb), dim3(256), 0, ctx->stream, (int)n, xs, ys, zs, g, ok, ctx->olev_key.as<int>(), fill);
    SPHX_TRY(excl_scan_plus_total(ctx, fill, ctx->olev_start.as<int>(), (int)ncell));
    hipLaunchKernelGGL(olev_scatter, dim3(pb), dim3(256), 0, ctx->stream, (int)n, ctx->olev_key.as<int>(),
                       ctx->olev_start.as<int>(), fill, ctx->olev_list.as<int>());
    HIPCHK(hipGetLastError());
    o.L = L;
    o.start = ctx->olev_start.as<int>();
    o.list = ctx->olev_list.as<int>();
    return SPHX_OK;
}

// ---- blob order: particles listed cell by cell along a space-filling curve -----------------------
// Hilbert curve over the cube of side 2^b that holds the grid (b = bits of the longest axis) when its
// code space fits (3b <= 27 bits): consecutive cells are face neighbours, so a run of consecutive
// particles - a workgroup's blob, a wave's query group (sphx_knn_group.hip) - is one connected lump:
// the box of 64 consecutive particles plus their search radius holds ~1300 candidates against ~2000 for
// the Morton runs of the first version, which jump at every octant boundary (measured on the 1e6
// polytrope).  Otherwise (a very elongated or very fine grid): Morton code with the bits of (cx, cy, cz)
// interleaved from the least significant end, each axis contributing only the bits it has, so the code
// space is at most 8x the cell count whatever the grid's aspect ratio.
// one thread per cell; sort_perm != nullptr: also sorts the cell's members (cell_sort_members, deferred to here)
// crowded != nullptr: cells of 17 .. 512 members are not sorted here but listed (crowded[0] = how many, then their
// indices) for cell_sort_crowded - a wave sorts its crowded cells one after the other, and the crowded cells of a cloud
// with a dense core sit together: 64 in one wave and none in the next was 0.24 ms of a launch that takes 0.03
__global__ __launch_bounds__(256) void blob_count(GridParams g, BlobBits b, const int* cell_start, int* mcount,
                                                  int* sort_perm, int* crowded, u64* counters) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    int s0 = 0, cnt = 0;
    if (c < g.ncells) { s0 = cell_start[c]; cnt = cell_start[c + 1] - s0; }
    if (sort_perm) {          // how many crowded cells there are: the host's cue (a step late) for the list form
        const u64 bm0 = __builtin_amdgcn_ballot_w64(cnt > CELL_SORT_SERIAL && cnt <= CELL_SORT_WAVE);
        if (bm0 && counters && (threadIdx.x & 63) == 0) atomicAdd(&counters[SC_CROWDED], (u64)__popcll(bm0));
        if (bm0 && counters) {            // particles in dense cells: the host's cue for finer cells (sphx_api.hip)
            int dp = cnt >= DENSE_CELL ? cnt : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dp += __shfl_xor(dp, o, 64);
            if (dp && (threadIdx.x & 63) == 0) atomicAdd(&counters[SC_DENSEP], (u64)dp);
        }
    }
    if (sort_perm && crowded) {
        sort_cell_members(sort_perm, s0, cnt <= CELL_SORT_SERIAL ? cnt : 0, true);
        const bool big = cnt > CELL_SORT_SERIAL && cnt <= CELL_SORT_WAVE;
        const u64 bm = __builtin_amdgcn_ballot_w64(big);
        if (bm) {
            int base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(crowded, __popcll(bm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (big) crowded[1 + base + __popcll(bm & ((1ull << (threadIdx.x & 63)) - 1ull))] = c;
        }
    } else {
        sort_cell_members(sort_perm, s0, cnt, sort_perm != nullptr);   // (the whole wave: crowded cells are sorted together)
    }
    if (cnt == 0) return;
    const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
    mcount[blob_rank(cx, cy, cz, b)] = cnt;
}
// one wave per listed cell
__global__ __launch_bounds__(256) void cell_sort_crowded(const int* __restrict__ crowded, const int* __restrict__ cell_start,
                                                         int* perm) {
    const int total = crowded[0];
    const int lane = threadIdx.x & 63;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < total; e += gridDim.x * 4) {
        const int c = crowded[1 + e];
        const int s = cell_start[c];
        sort_crowded_cell_any(perm, s, cell_start[c + 1] - s, lane);
    }
}
__global__ __launch_bounds__(256) void blob_scatter(int n, GridParams g, BlobBits b, const int* cell_of,
                                                    const int* perm, const int* cell_start, const int* mstart,
                                                    int* porder) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;       // storage (cell-sorted) index
    if (t >= n) return;
    const int c = cell_of[perm[t]];
    const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
    porder[mstart[blob_rank(cx, cy, cz, b)] + (t - cell_start[c])] = t;
}

// porder[p] = storage index of the p-th particle in blob order; ctx->qorder points at it on success
// (left nullptr = identity when the code space would be unreasonably large).
int sphx_build_blob_order(sphx_ctx* ctx, int64_t n) {
    const GridParams g = ctx->grid;
    BlobBits b{0, 0, 0, 0};
    while ((1 << b.bx) < g.nx) ++b.bx;
    while ((1 << b.by) < g.ny) ++b.by;
    while ((1 << b.bz) < g.nz) ++b.bz;
    int bits = b.bx + b.by + b.bz;
    int hb = b.bx > b.by ? b.bx : b.by;
    if (b.bz > hb) hb = b.bz;
    if (hb < 1) hb = 1;
    // the cube's code space may be up to 8x the Morton one: taken while it stays within 2^24 codes (a
    // 64 MB count + scan) or within 4x the tight code space
    if (ctx->blob_curve != 1 && 3 * hb <= 27 && (3 * hb <= 24 || 3 * hb <= bits + 2)) { b.hilbert = hb; bits = 3 * hb; }
    ctx->qorder = nullptr;
    if (bits > 27) {
        if (ctx->cells_unsorted) {
            hipLaunchKernelGGL(cell_sort_members, dim3((unsigned)((g.ncells + 255) / 256)), dim3(256), 0, ctx->stream,
                               g.ncells, ctx->cell_start.as<int>(), ctx->perm.as<int>());
            ctx->cells_unsorted = false;
        }
        return SPHX_OK;
    }
    const int M = 1 << bits;
    SPHX_TRY(sphx_ensure(ctx, ctx->porder, (size_t)n * sizeof(int)));
    SPHX_TRY(sphx_ensure(ctx, ctx->mcount, ((size_t)M + 2) * sizeof(int)));
    SPHX_TRY(sphx_ensure(ctx, ctx->mstart, ((size_t)M + 2) * sizeof(int)));
    int* mc = ctx->mcount.as<int>();
    int* ms = ctx->mstart.as<int>();
    // (all zero between builds when the deferred scatter below cleans up after the scan: no memset then, except once per
    //  allocation or after a build that did not)
    if (!(ctx->mcount_zeroed == ctx->mcount.p && ctx->mcount_zeroed_M == M)) {
        HIPCHK(hipMemsetAsync(mc, 0, ctx->mcount.cap, ctx->stream));
    }
    ctx->mcount_zeroed = nullptr;
    int* crowded = nullptr;
    // many crowded cells (a dense core: they sit together, 64 to a wave) are sorted by a launch of their own, one wave
    // per cell; a few are sorted where they are found (two launches less: the headline's case)
    if (ctx->cells_unsorted && ctx->crowded_last >= CROWDED_LIST_MIN) {          // (a crowded cell has >= 17 members: at most n / 17 of them)
        SPHX_TRY(sphx_ensure(ctx, ctx->crowded, ((size_t)n / 17 + 2) * sizeof(int)));
        crowded = ctx->crowded.as<int>();
        HIPCHK(hipMemsetAsync(crowded, 0, sizeof(int), ctx->stream));
    }
    hipLaunchKernelGGL(blob_count, dim3((unsigned)((g.ncells + 255) / 256)), dim3(256), 0, ctx->stream, g, b,
                       ctx->cell_start.as<int>(), mc, ctx->cells_unsorted ? ctx->perm.as<int>() : nullptr, crowded,
                       ctx->scal.as<u64>());
    if (crowded)
        hipLaunchKernelGGL(cell_sort_crowded, dim3(1024), dim3(256), 0, ctx->stream, crowded, ctx->cell_start.as<int>(),
                           ctx->perm.as<int>());
    ctx->cells_unsorted = false;
    SPHX_TRY(excl_scan_plus_total(ctx, mc, ms, M));
    if (ctx->defer_blob_scatter) {
        // the fused loop: the scatter rides in the kernel that permutes the state next (sphx_permute_state), one
        // thread per stored particle there as here
        ctx->blob_scatter_pending = true;
        ctx->blob_scatter_bits = b;
        ctx->blob_scatter_mstart = ms;
        ctx->mcount_zeroed = ctx->mcount.p;        // (the scatter puts the counts it used back to zero)
        ctx->mcount_zeroed_M = M;
    } else {
        hipLaunchKernelGGL(blob_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, g, b,
                           ctx->cell_of.as<int>(), ctx->perm.as<int>(), ctx->cell_start.as<int>(), ms,
                           ctx->porder.as<int>());
    }
    HIPCHK(hipGetLastError());
    ctx->qorder = ctx->porder.as<int>();
    return SPHX_OK;
}
