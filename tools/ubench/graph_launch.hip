// graph_launch.hip -- is a hipGraph worth it for the per-block launch sequence of small blocks?
// Five short dependent kernels + an 8-byte read-back into page-locked memory + a stream synchronisation per
// iteration (the shape of find_carrier at N = 2^15), launched one by one and as one instantiated graph.
//   hipcc --offload-arch=gfx950 -O3 -o graph_launch tools/ubench/graph_launch.hip && ./graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e));                          \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void k_work(float *p, int n, int rounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int r = 0; r < rounds; ++r) v = v * 1.0001f + 0.5f;
    p[i] = v;
}

int main() {
    const int n = 1 << 16, iters = 2000, nk = 5;
    float *d, *h;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    CK(hipHostMalloc(&h, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto enqueue = [&]() {
        for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n, 20);
        return hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s);
    };
    for (int rounds = 0; rounds < 2; ++rounds) {       // second round: everything warm
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) {
            CK(enqueue());
            CK(hipStreamSynchronize(s));
        }
        auto t1 = std::chrono::steady_clock::now();
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(enqueue());
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t2 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) {
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
        }
        auto t3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        printf("round %d: %d kernels + read-back + sync: direct %.1f us / iteration, graph %.1f us / iteration\n", rounds, nk,
               us(t0, t1) / iters, us(t2, t3) / iters);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
