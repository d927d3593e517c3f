// launchbench.hip — how many dependent kernel launches per microsecond does the device take, over 1..8 streams,
// each fed by its own host thread? (GPU box)   hipcc --offload-arch=gfx950 -O3 -o /tmp/lb tools/launchbench.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(int* p, int n_blocks_work) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
// a kernel that dirties `bytes` of memory first (what the end-of-kernel write-back has to flush)
__global__ void dirty(uint32_t* buf, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) buf[i] = (uint32_t)i;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 20000;
    for (int wgs : {1, 256, 4096}) {
        for (int ns : {1, 2, 4, 8}) {
            std::vector<hipStream_t> st(ns);
            std::vector<int*> d(ns);
            for (int i = 0; i < ns; i++) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipMalloc(&d[i], 64); hipMemset(d[i], 0, 64); }
            hipDeviceSynchronize();
            const double t0 = now();
            std::vector<std::thread> th;
            for (int i = 0; i < ns; i++) th.emplace_back([&, i] {
                hipSetDevice(0);
                for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(128), 0, st[i], d[i], 0);
            });
            for (auto& t : th) t.join();
            const double t1 = now();
            hipDeviceSynchronize();
            const double t2 = now();
            std::printf("grid %5d x128  streams %d: %.2f us per launch per stream, %.2f launches/us in all (host enqueue %.2f us per launch per thread)\n",
                        wgs, ns, (t2 - t0) / N * 1e6, ns * N / ((t2 - t0) * 1e6), (t1 - t0) / N * 1e6);
            for (int i = 0; i < ns; i++) { hipStreamDestroy(st[i]); hipFree(d[i]); }
        }
    }
    // with 16 MB dirtied per kernel on ONE of the streams (a fill running beside): what do the others' launches cost?
    {
        uint32_t* big; hipMalloc(&big, 256u << 20);
        const int ns = 4;
        std::vector<hipStream_t> st(ns); std::vector<int*> d(ns);
        for (int i = 0; i < ns; i++) { hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); hipMalloc(&d[i], 64); hipMemset(d[i], 0, 64); }
        hipDeviceSynchronize();
        const double t0 = now();
        std::vector<std::thread> th;
        for (int i = 0; i < ns; i++) th.emplace_back([&, i] {
            hipSetDevice(0);
            for (int k = 0; k < N; k++) {
                if (i == 0) hipLaunchKernelGGL(dirty, dim3(1024), dim3(256), 0, st[i], big, (size_t)(16u << 20) / 4);
                else hipLaunchKernelGGL(tiny, dim3(256), dim3(128), 0, st[i], d[i], 0);
            }
        });
        for (auto& t : th) t.join();
        hipDeviceSynchronize();
        const double t2 = now();
        std::printf("3 streams of tiny kernels beside 1 stream writing 16 MB per kernel: %.2f us per launch per stream\n", (t2 - t0) / N * 1e6);
    }
    return 0;
}
