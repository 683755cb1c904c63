// 32-bit VALU issue rates on gfx950 by waves per SIMD (what "VALU-bound" means for the search kernels).
//   hipcc --offload-arch=gfx950 -O3 tools/valurate.hip -o valurate && ./valurate
// Each wave runs a long stream of independent 32-bit VALU instructions of one kind (8 accumulators); the grid puts
// W waves on every SIMD (256 CUs x 4 SIMDs).  Printed: wave-instructions per second and shader cycles per
// wave-instruction per SIMD at the clock measured in the kernel (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned a0, unsigned long long* clk) {
    unsigned a[8];
    f32x2 p[8];
    double q[8];
    const double db = 1.0000001, dc = 1e-9;
    for (int i = 0; i < 8; ++i) { a[i] = a0 + threadIdx.x * 7u + i; p[i] = f32x2{1.0f + i, 2.0f + threadIdx.x}; q[i] = 1.0 + i; }
    const unsigned b = 0x9E3779B9u;
    const float fb = 1.0000001f, fc = 1e-9f;
    const f32x2 pb = {fb, fb}, pc = {fc, fc};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // (inline asm: the compiler would otherwise fold the chains or re-pack them)
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc));
                else if (OP == 1) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
                else if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                else if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                else if (OP == 6) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fb));
                else if (OP == 7) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 8) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 9) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 10) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                else if (OP == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fc));
                else if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fb));
                else if (OP == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                else if (OP == 15) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[i]));
                else if (OP == 16) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                else if (OP == 17) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(fb), "v"(fc) : "vcc");
                else if (OP == 18) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(q[i]) : "v"(db), "v"(dc));
                else if (OP == 19) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(b));
                else if (OP == 20) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(q[i]) : "v"(db));
                else if (OP == 21) asm volatile("v_add_f64 %0, %0, %1" : "+v"(q[i]) : "v"(dc));
                else if (OP == 22) asm volatile("v_max_f64 %0, %0, %1" : "+v"(q[i]) : "v"(dc));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + __float_as_uint(p[i].x + p[i].y) + (unsigned)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int OP> void run(const char* name, int per, unsigned* d, unsigned long long* clk, int wps) {
    const int blocks = 256 * wps, iters = 4000;               // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 5u, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 5u, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);           // s_memrealtime ticks at 100 MHz
    const double waveinstr = (double)blocks * 4 * iters * 64 * per;
    printf("%-14s %d waves/SIMD %8.3f ms  %7.1f G wave-instr/s  clock %.2f GHz  %.2f cycles per wave-instr per SIMD\n", name, wps, ms,
           waveinstr / ms / 1e6, ghz, ms * 1e-3 * ghz * 1e9 * 1024 / waveinstr);
}
int main() {
    unsigned* d; hipMalloc(&d, (size_t)256 * 8 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    for (int wps : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 1, d, clk, wps); run<1>("v_min_u32", 1, d, clk, wps); run<2>("v_alignbit", 1, d, clk, wps);
        run<3>("v_pk_fma_f32", 1, d, clk, wps); run<4>("v_add_u32", 1, d, clk, wps); run<5>("v_pk_add_f32", 1, d, clk, wps);
    }
    for (int wps : {2, 5}) {
        run<6>("v_min_f32", 1, d, clk, wps); run<7>("v_max_u32", 1, d, clk, wps); run<8>("v_min_i32", 1, d, clk, wps);
        run<9>("v_and_b32", 1, d, clk, wps); run<10>("v_lshlrev_b32", 1, d, clk, wps); run<11>("v_sub_f32", 1, d, clk, wps);
        run<12>("v_mul_f32", 1, d, clk, wps); run<13>("v_lshl_or_b32", 1, d, clk, wps); run<14>("v_cndmask_b32", 1, d, clk, wps);
        run<15>("v_bfe_u32", 1, d, clk, wps); run<16>("v_mov_dpp", 1, d, clk, wps); run<17>("v_cmp+v_addc", 2, d, clk, wps);
        run<18>("v_fma_f64", 1, d, clk, wps); run<19>("v_min3_u32", 1, d, clk, wps); run<20>("v_mul_f64", 1, d, clk, wps);
        run<21>("v_add_f64", 1, d, clk, wps); run<22>("v_max_f64", 1, d, clk, wps);
    }
    return 0;
}
