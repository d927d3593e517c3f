// fillshape.hip — which SHAPE of the background fill (64 MiB RGBA8 + 16 MiB palette index per 4096^2 frame) does HBM
// take fastest? Eight frames' buffers in a ring (640 MiB: beyond the Infinity Cache), one launch per frame.
//   hipcc --offload-arch=gfx950 -O3 -o build/tools/fillshape tools/fillshape.hip && build/tools/fillshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// the product's shape: a wavefront per 512-pixel chunk: two 1-KiB frame stores + one 512-B palette store
__global__ __launch_bounds__(256) void shape_chunk512(uint32_t* fb, uint8_t* pal, long long npix) {
    const int lane = threadIdx.x & 63;
    const long long n_chunks = npix / 512;
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    const u32x2 w = {~0u, ~0u};
    for (long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); c < n_chunks; c += (long long)gridDim.x * 4) {
        const long long p0 = c * 512;
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(fb + p0 + lane * 4));
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(fb + p0 + 256 + lane * 4));
        __builtin_nontemporal_store(w, reinterpret_cast<u32x2*>(pal + p0 + lane * 8));
    }
}
// 2048-pixel chunks: eight 1-KiB frame stores + two 1-KiB palette stores
__global__ __launch_bounds__(256) void shape_chunk2048(uint32_t* fb, uint8_t* pal, long long npix) {
    const int lane = threadIdx.x & 63;
    const long long n_chunks = npix / 2048;
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    const u32x4 w = {~0u, ~0u, ~0u, ~0u};
    for (long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); c < n_chunks; c += (long long)gridDim.x * 4) {
        const long long p0 = c * 2048;
#pragma unroll
        for (int h = 0; h < 8; h++) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(fb + p0 + h * 256 + lane * 4));
#pragma unroll
        for (int h = 0; h < 2; h++) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(pal + p0 + h * 1024 + lane * 16));
    }
}
// one plane after the other, each swept linearly by the whole grid (16 B per lane, consecutive threads consecutive)
__global__ __launch_bounds__(256) void shape_planes(uint32_t* fb, uint8_t* pal, long long npix) {
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    const u32x4 w = {~0u, ~0u, ~0u, ~0u};
    const long long stride = (long long)gridDim.x * blockDim.x, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = t; i < npix / 4; i += stride) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(fb) + i);
    for (long long i = t; i < npix / 16; i += stride) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(pal) + i);
}
// the same with plain stores
__global__ __launch_bounds__(256) void shape_planes_plain(uint32_t* fb, uint8_t* pal, long long npix) {
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    const u32x4 w = {~0u, ~0u, ~0u, ~0u};
    const long long stride = (long long)gridDim.x * blockDim.x, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = t; i < npix / 4; i += stride) reinterpret_cast<u32x4*>(fb)[i] = v;
    for (long long i = t; i < npix / 16; i += stride) reinterpret_cast<u32x4*>(pal)[i] = w;
}
// both planes swept linearly at once: per step four frame stores and one palette store of the same pixels
__global__ __launch_bounds__(256) void shape_both(uint32_t* fb, uint8_t* pal, long long npix) {
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    const u32x4 w = {~0u, ~0u, ~0u, ~0u};
    const long long stride = (long long)gridDim.x * blockDim.x, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (long long i = t; i < npix / 16; i += stride) {
        const long long wave0 = (i - lane) * 16;  // first pixel of this wavefront's 1024
#pragma unroll
        for (int h = 0; h < 4; h++) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(fb + wave0 + h * 256 + lane * 4));
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(pal) + i);
    }
}

int main() {
    const long long npix = 4096LL * 4096;
    const int R = 8;
    std::vector<uint32_t*> fb(R);
    std::vector<uint8_t*> pal(R);
    for (int r = 0; r < R; r++) { HIP_OK(hipMalloc(&fb[r], npix * 4)); HIP_OK(hipMalloc(&pal[r], npix)); }
    hipStream_t s; HIP_OK(hipStreamCreate(&s));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    auto report = [&](const char* name, int grid, auto kern) {
        for (int i = 0; i < R; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, fb[i % R], pal[i % R], npix);
        HIP_OK(hipStreamSynchronize(s));
        const int n = 64;
        HIP_OK(hipEventRecord(e0, s));
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, fb[i % R], pal[i % R], npix);
        HIP_OK(hipEventRecord(e1, s));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-22s grid %5d  %7.2f us per frame  %5.2f TB/s\n", name, grid, ms / n * 1e3, 5.0 * npix / (ms / n * 1e-3) / 1e12);
    };
    for (int grid : {64, 128, 256, 512, 1024, 2048}) {
        report("chunk512 (product)", grid, shape_chunk512);
        report("chunk2048", grid, shape_chunk2048);
        report("planes in turn", grid, shape_planes);
        report("planes in turn, plain", grid, shape_planes_plain);
        report("both planes, linear", grid, shape_both);
    }
    return 0;
}
