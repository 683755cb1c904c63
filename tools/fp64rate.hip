// fp64 VALU issue rates on gfx950: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/fp64rate.hip -o fp64rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * 1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] * b;
                else if (OP == 1) a[i] = a[i] + c;
                else if (OP == 2) a[i] = __builtin_fma(a[i], b, c);
                else if (OP == 3) a[i] = __builtin_amdgcn_rsq(a[i]) ;
                else if (OP == 4) a[i] = fmax(a[i], b);
                else if (OP == 5) a[i] = sqrt(a[i]) + 1.0;
                else if (OP == 6) a[i] = b / a[i] + 2.0;
            }
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double* d) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 64;          // lane-ops
    double waveinstr = ops / 64;
    // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / waveinstr;
    printf("%-10s %8.3f ms  %.2f Tlane-op/s  ~%.2f clk per wave-instr per SIMD\n", name, ms, ops / ms / 1e9, cyc);
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("mul_f64", d); run<1>("add_f64", d); run<2>("fma_f64", d); run<3>("rsq_f64", d); run<4>("max_f64", d);
    run<5>("sqrt+add", d); run<6>("div+add", d);
    return 0;
}
