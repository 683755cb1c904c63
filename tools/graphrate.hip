// How much does a hipGraph buy for a chain of tiny dependent kernels on this runtime?  (the 1e4-particle step)
//   hipcc --offload-arch=gfx950 -O3 -w tools/graphrate.hip -o graphrate && ./graphrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void tiny(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1e-9; }
int main() {
    const int n = 10000, nk = 25, reps = 400;
    double* d; hipMalloc(&d, n * sizeof(double)); hipMemset(d, 0, n * sizeof(double));
    hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t f, j; hipEventCreateWithFlags(&f, hipEventDisableTiming); hipEventCreateWithFlags(&j, hipEventDisableTiming);
    auto chain = [&](bool fork) {
        for (int k = 0; k < nk; ++k) {
            if (fork && k == 10) {              // one fork / join onto a second stream, as the step has
                hipEventRecord(f, s); hipStreamWaitEvent(s2, f, 0);
                hipLaunchKernelGGL(tiny, dim3((n + 255) / 256), dim3(256), 0, s2, d, n);
                hipEventRecord(j, s2); hipStreamWaitEvent(s, j, 0);
            }
            hipLaunchKernelGGL(tiny, dim3((n + 255) / 256), dim3(256), 0, s, d, n);
        }
    };
    for (int w = 0; w < 20; ++w) chain(true);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) chain(true);
    hipStreamSynchronize(s);
    double us_direct = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    hipGraph_t g; hipGraphExec_t ge;
    auto c0 = std::chrono::steady_clock::now();
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    chain(true);
    hipError_t e = hipStreamEndCapture(s, &g);
    hipError_t e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    double us_cap = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
    printf("capture %s, instantiate %s: %.0f us\n", hipGetErrorString(e), hipGetErrorString(e2), us_cap);
    for (int w = 0; w < 20; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%d dependent kernels of %d elements (+1 forked): direct launches %.1f us per chain, graph replay %.1f us per chain\n", nk, n, us_direct, us_graph);
    return 0;
}
