// fillbench.hip — how fast can the background fill (64 MiB RGBA8 + 16 MiB palette index at 4096^2) be written?
// Sweeps store shape, cache policy and grid size of a streaming fill on its own (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fillbench tools/fillbench.hip && /tmp/fillbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x)                                                                \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ void st(T v, T* p) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// PX pixels per wave-iteration (512, 1024, 2048): fb as PX/256 dwordx4 per lane, 1 KiB contiguous per instruction;
// palette index as 8 B (PX = 512) or 16 B per lane.
template <int PX, bool NT>
__global__ __launch_bounds__(256) void fill_px(uint32_t* fb, uint8_t* pal, long long npix, uint32_t rgba) {
    const int lane = threadIdx.x & 63;
    const long long n_chunks = npix / PX;
    const int wpb = blockDim.x >> 6;
    const u32x4 v = {rgba, rgba, rgba, rgba};
    for (long long c = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); c < n_chunks; c += (long long)gridDim.x * wpb) {
        const long long p0 = c * PX;
#pragma unroll
        for (int h = 0; h < PX / 256; h++) st<NT>(v, reinterpret_cast<u32x4*>(fb + p0 + h * 256 + lane * 4));
        if (PX == 512) {
            const u32x2 w = {0xFFFFFFFFu, 0xFFFFFFFFu};
            st<NT>(w, reinterpret_cast<u32x2*>(pal + p0 + lane * 8));
        } else {
            const u32x4 w = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
            for (int h = 0; h < PX / 1024; h++) st<NT>(w, reinterpret_cast<u32x4*>(pal + p0 + h * 1024 + lane * 16));
        }
    }
}

// each workgroup owns one contiguous span of the frame (instead of chunks interleaved over the grid)
template <bool NT>
__global__ __launch_bounds__(256) void fill_span(uint32_t* fb, uint8_t* pal, long long npix, uint32_t rgba) {
    const long long per = npix / gridDim.x;  // multiple of 1024 by construction
    const long long b0 = per * blockIdx.x;
    const u32x4 v = {rgba, rgba, rgba, rgba};
    const u32x4 w = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (long long p = b0 + threadIdx.x * 4; p < b0 + per; p += 1024) st<NT>(v, reinterpret_cast<u32x4*>(fb + p));
    for (long long p = b0 + threadIdx.x * 16; p < b0 + per; p += 4096) st<NT>(w, reinterpret_cast<u32x4*>(pal + p));
}

int main() {
    const long long W = 4096, H = 4096, npix = W * H;
    uint32_t* fb;
    uint8_t* pal;
    HIP_OK(hipMalloc(&fb, npix * 4));
    HIP_OK(hipMalloc(&pal, npix));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    hipStream_t s;
    HIP_OK(hipStreamCreate(&s));
    const double bytes = 5.0 * npix;
    auto report = [&](const char* name, int grid, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        HIP_OK(hipStreamSynchronize(s));
        const int n = 100;
        HIP_OK(hipEventRecord(e0, s));
        for (int i = 0; i < n; i++) launch();
        HIP_OK(hipEventRecord(e1, s));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-28s grid %5d  %7.2f us  %6.2f TB/s\n", name, grid, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
    };
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        report("px512 nt", grid, [&] { hipLaunchKernelGGL((fill_px<512, true>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("px512 plain", grid, [&] { hipLaunchKernelGGL((fill_px<512, false>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("px1024 nt", grid, [&] { hipLaunchKernelGGL((fill_px<1024, true>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("px1024 plain", grid, [&] { hipLaunchKernelGGL((fill_px<1024, false>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("px2048 nt", grid, [&] { hipLaunchKernelGGL((fill_px<2048, true>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("px2048 plain", grid, [&] { hipLaunchKernelGGL((fill_px<2048, false>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("span nt", grid, [&] { hipLaunchKernelGGL((fill_span<true>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
        report("span plain", grid, [&] { hipLaunchKernelGGL((fill_span<false>), dim3(grid), dim3(256), 0, s, fb, pal, npix, 0x1F1F1Fu); });
    }
    report("hipMemsetD32Async x2", 0, [&] {
        (void)hipMemsetD32Async((hipDeviceptr_t)fb, 0x1F1F1Fu, npix, s);
        (void)hipMemsetD32Async((hipDeviceptr_t)pal, 0xFFFFFFFFu, npix / 4, s);
    });
    // one long launch (no launch boundaries inside the measurement): 8 frames' worth in one grid-stride pass
    {
        uint32_t* big;
        uint8_t* bigp;
        const long long n8 = npix * 8;
        HIP_OK(hipMalloc(&big, n8 * 4));
        HIP_OK(hipMalloc(&bigp, n8));
        for (int grid : {1024, 2048}) {
            for (int i = 0; i < 3; i++) hipLaunchKernelGGL((fill_px<1024, true>), dim3(grid), dim3(256), 0, s, big, bigp, n8, 0x1F1F1Fu);
            HIP_OK(hipEventRecord(e0, s));
            hipLaunchKernelGGL((fill_px<1024, true>), dim3(grid), dim3(256), 0, s, big, bigp, n8, 0x1F1F1Fu);
            HIP_OK(hipEventRecord(e1, s));
            HIP_OK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("px1024 nt, 8 frames in one launch, grid %d: %.2f us per frame, %.2f TB/s\n", grid, ms / 8 * 1e3,
                        bytes * 8 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
