// fillstream.hip — K frames in flight WITHOUT their own fill (flag bit 28) beside ONE continuous, paced streaming
// fill on a stream of its own: what frame rate do the latency-bound chains reach next to a fill that never
// oversubscribes the memory system? (GPU box; GPU_MAX_HW_QUEUES must give every stream a queue of its own)
//   build/tools/fillstream [frames]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "par_raytracer.h"

#define HIP_OK(x)                                                                \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int SLEEP>
__global__ __launch_bounds__(256) void fill_timed(u32x4* p, size_t n16, unsigned long long ticks, unsigned long long* written) {
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long n = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            __builtin_nontemporal_store(v, p + i);
            if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
            i += (size_t)gridDim.x * blockDim.x;
            if (i >= n16) i -= n16;
        }
        n += 4;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(written, n);
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int n_frames = argc > 1 ? std::atoi(argv[1]) : 80;
    const int W = 4096, KMAX = 8;
    par_params params;
    par_default_params(&params);
    params.width = params.height = params.length = W;
    std::vector<par_aabb> aabbs(1024);
    par_light light;
    par_scene_synthetic(1024, W, W, W, 12345, aabbs.data(), &light);
    par_sprite sprite;
    par_sprite_tile_floor(&sprite);
    std::vector<par_context*> ctx(KMAX, nullptr);
    std::vector<par_outputs> out(KMAX);
    std::vector<hipStream_t> st(KMAX);
    for (int k = 0; k < KMAX; k++) {
        if (par_create(&params, 0, &ctx[k]) != PAR_OK) return 1;
        par_set_sprites(ctx[k], &sprite, 1);
        par_set_entities(ctx[k], aabbs.data(), nullptr, 1024);
        par_set_light(ctx[k], &light);
        out[k] = par_outputs{};
        HIP_OK(hipMalloc(&out[k].fb, (size_t)W * W * 4));
        HIP_OK(hipMalloc(&out[k].palidx, (size_t)W * W));
        HIP_OK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    hipStream_t sx;
    HIP_OK(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking));
    const size_t big = (size_t)1 << 30;
    u32x4* d_big;
    HIP_OK(hipMalloc(&d_big, big));
    unsigned long long* d_written;
    HIP_OK(hipMalloc(&d_written, 8));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));

    struct Fill { int wgs, sleep; };
    const std::vector<Fill> fills = {{0, 0}, {48, 0}, {64, 0}, {96, 1}, {128, 2}};
    for (unsigned own : {1u, 0u}) {
        std::printf("--- frames %s\n", own ? "with their own riding fill (the product as it is), nothing beside them"
                                            : "WITHOUT their own fill (bit 28), a continuous fill beside them");
        for (int K : {2, 3, 4, 6, 8}) {
            for (const Fill& f : fills) {
                if (own && f.wgs) continue;
                const unsigned flags = (own ? 0u : (1u << 28)) | (K > 1 ? (unsigned)PAR_RENDER_PIPELINED : 0u);
                for (int i = 0; i < 4 * K; i++) par_render_device(ctx[i % K], st[i % K], 0, W, &out[i % K], flags);
                HIP_OK(hipDeviceSynchronize());
                if (f.wgs) {
                    HIP_OK(hipMemsetAsync(d_written, 0, 8, sx));
                    HIP_OK(hipEventRecord(e0, sx));
                    const unsigned long long ticks = (unsigned long long)(n_frames * 30 + 400) * 100;  // outlasts the frames
                    switch (f.sleep) {
                        case 0: hipLaunchKernelGGL(fill_timed<0>, dim3(f.wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                        case 1: hipLaunchKernelGGL(fill_timed<1>, dim3(f.wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                        default: hipLaunchKernelGGL(fill_timed<2>, dim3(f.wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                    }
                    HIP_OK(hipEventRecord(e1, sx));
                }
                const double t0 = now_s();
                for (int i = 0; i < n_frames; i++) par_render_device(ctx[i % K], st[i % K], 0, W, &out[i % K], flags);
                const double t_enq = now_s() - t0;
                for (int k = 0; k < K; k++) HIP_OK(hipStreamSynchronize(st[k]));
                const double per = (now_s() - t0) / n_frames * 1e6;
                double tbs = 0.0;
                bool still = true;
                if (f.wgs) {
                    still = hipEventQuery(e1) == hipErrorNotReady;
                    HIP_OK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned long long wr = 0;
                    HIP_OK(hipMemcpy(&wr, d_written, 8, hipMemcpyDeviceToHost));
                    tbs = (double)wr * 1024.0 / (ms * 1e-3) / 1e12;
                }
                std::printf("K %d  fill wgs %3d sleep %d: %6.2f us per frame (enqueue %5.2f)  fill %5.2f TB/s = %5.1f us per 84 MB%s\n", K,
                            f.wgs, f.sleep, per, t_enq / n_frames * 1e6, tbs, tbs > 0 ? 83.886 / tbs : 0.0,
                            still ? "" : "  (fill ended early)");
                HIP_OK(hipDeviceSynchronize());
            }
        }
    }
    for (int k = 0; k < KMAX; k++) par_destroy(ctx[k]);
    return 0;
}
