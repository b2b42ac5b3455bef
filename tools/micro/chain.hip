// Latency floors of tiny dependent-load kernels on MI355X (diagnostics, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void k0(int n, int* out) {}
__global__ void k1(int n, const int* a, int* out) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = a[i]; }
__global__ void k2(int n, const int* a, const int* b, int* out) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = b[a[i]]; }
__global__ void k3(int n, const int* a, const int* b, const int* c, int* out) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = c[b[a[i]]]; }
__global__ void k4(int n, const int* a, const int* b, const int* c, const int* d, int* out) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = d[c[b[a[i]]]]; }
template <class F> float timeit(F f, int reps = 200) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
template <class F> float timegraph(F f, int reps = 200) {
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < reps; ++i) f(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    for (int n : {16384, 122880, 614400, 2457600}) {
        std::vector<int> id(n); std::iota(id.begin(), id.end(), 0);
        std::vector<int> near(n);
        std::mt19937 rng(1);
        for (int i = 0; i < n; ++i) near[i] = std::min(n - 1, std::max(0, i + (int)(rng() % 129) - 64));
        int *a, *b, *c, *d, *o;
        for (int** p : {&a, &b, &c, &d, &o}) hipMalloc(p, n * sizeof(int));
        for (int* p : {a, b, c, d}) hipMemcpy(p, near.data(), n * 4, hipMemcpyHostToDevice);
        int g = (n + 255) / 256;
        printf("n %8d grid %6d | stream: empty %.2f  1-load %.2f  2-chain %.2f  3-chain %.2f  4-chain %.2f us", n, g,
               timeit([&] { hipLaunchKernelGGL(k0, dim3(g), dim3(256), 0, 0, n, o); }),
               timeit([&] { hipLaunchKernelGGL(k1, dim3(g), dim3(256), 0, 0, n, a, o); }),
               timeit([&] { hipLaunchKernelGGL(k2, dim3(g), dim3(256), 0, 0, n, a, b, o); }),
               timeit([&] { hipLaunchKernelGGL(k3, dim3(g), dim3(256), 0, 0, n, a, b, c, o); }),
               timeit([&] { hipLaunchKernelGGL(k4, dim3(g), dim3(256), 0, 0, n, a, b, c, d, o); }));
        printf(" | graph: empty %.2f  1-load %.2f  2-chain %.2f  3-chain %.2f  4-chain %.2f us\n",
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k0, dim3(g), dim3(256), 0, s, n, o); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k1, dim3(g), dim3(256), 0, s, n, a, o); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k2, dim3(g), dim3(256), 0, s, n, a, b, o); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k3, dim3(g), dim3(256), 0, s, n, a, b, c, o); }),
               timegraph([&](hipStream_t s) { hipLaunchKernelGGL(k4, dim3(g), dim3(256), 0, s, n, a, b, c, d, o); }));
        for (int* p : {a, b, c, d, o}) hipFree(p);
    }
    return 0;
}
