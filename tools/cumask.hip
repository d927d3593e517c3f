// cumask.hip — does a CU-masked stream keep a streaming fill off the CUs the latency-bound kernels of a frame run on,
// and what does that buy them? (GPU box)
//   hipcc --offload-arch=gfx950 -O2 -Iinclude -o /tmp/cumask tools/cumask.hip -Lpixel-art-raytracer_amd/lib -lpar_raytracer \
//         -Wl,-rpath,$PWD/pixel-art-raytracer_amd/lib && /tmp/cumask
// 1. which CUs does a stream created with hipExtStreamCreateWithCUMask run on (HW_ID / XCC_ID of every workgroup)?
// 2. the headline frame without its own fill (flag bit 28), one at a time, while a long streaming fill runs on another
//    stream: fill on every CU / fill masked to a share of the CUs / both masked to complementary shares.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
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

__global__ __launch_bounds__(256) void where_kernel(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
        out[blockIdx.x] = ((xcc & 0xFu) << 16) | (hw & 0xFF00u);         // cu_id [11:8], sh_id [12], se_id [15:13]
    }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(8);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// `passes` passes over n16 16-byte pieces, streaming stores, 1 KiB contiguous per wave instruction
__global__ __launch_bounds__(256) void fill_long(u32x4* p, size_t n16, int passes) {
    const u32x4 v = {0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu, 0x1F1F1Fu};
    for (int k = 0; k < passes; k++) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
            __builtin_nontemporal_store(v, p + i);
        }
    }
}

// streaming stores for `ticks` of the 100 MHz clock, `sleep` x 64 cycles of s_sleep after every store instruction
// (pacing); every wavefront adds the 1-KiB pieces it wrote to `written`
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

static hipStream_t masked_stream(const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (mask.empty()) {
        HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
        HIP_OK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    }
    return s;
}

// bits [0, 256): `keep(i)` says whether CU bit i is set
template <class F>
static std::vector<uint32_t> make_mask(F keep) {
    std::vector<uint32_t> m(8, 0u);
    for (int i = 0; i < 256; i++) if (keep(i)) m[(size_t)i / 32] |= 1u << (i % 32);
    return m;
}

static void probe(const char* name, const std::vector<uint32_t>& mask, uint32_t* d_out) {
    hipStream_t s = masked_stream(mask);
    const int n = 8192;
    HIP_OK(hipMemsetAsync(d_out, 0xFF, n * sizeof(uint32_t), s));
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(256), 0, s, d_out, 200);
    HIP_OK(hipStreamSynchronize(s));
    std::vector<uint32_t> h((size_t)n);
    HIP_OK(hipMemcpy(h.data(), d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::set<uint32_t> cus;
    int per_xcc[16] = {0};
    for (uint32_t v : h) cus.insert(v);
    for (uint32_t v : cus) per_xcc[(v >> 16) & 15]++;
    std::printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
    for (int x = 0; x < 8; x++) std::printf(" %d", per_xcc[x]);
    std::printf("\n");
    HIP_OK(hipStreamDestroy(s));
}

int main(int argc, char** argv) {
    const int fill_share = argc > 1 ? std::atoi(argv[1]) : 4;  // the fill gets every fill_share-th CU
    uint32_t* d_out;
    HIP_OK(hipMalloc(&d_out, 8192 * sizeof(uint32_t)));
    probe("no mask", {}, d_out);
    probe("bits 0..63", make_mask([](int i) { return i < 64; }), d_out);
    probe("bits i % 4 == 0", make_mask([](int i) { return i % 4 == 0; }), d_out);
    probe("bits i % 4 != 0", make_mask([](int i) { return i % 4 != 0; }), d_out);
    probe("bits (i / 8) % 4 == 0", make_mask([](int i) { return (i / 8) % 4 == 0; }), d_out);
    probe("bits (i / 32) % 4 == 0", make_mask([](int i) { return (i / 32) % 4 == 0; }), d_out);

    // ---- the headline frame, no fill of its own, while a streaming fill runs beside it --------------------------
    const int W = 4096;
    par_params params;
    par_default_params(&params);
    params.width = params.height = params.length = W;
    std::vector<par_aabb> aabbs(1024);
    par_light light;
    par_scene_synthetic(1024, W, W, W, 12345, aabbs.data(), &light);
    par_sprite sprite;
    par_sprite_tile_floor(&sprite);
    par_context* ctx = nullptr;
    if (par_create(&params, 0, &ctx) != PAR_OK) return 1;
    par_set_sprites(ctx, &sprite, 1);
    par_set_entities(ctx, aabbs.data(), nullptr, 1024);
    par_set_light(ctx, &light);
    par_outputs out{};
    HIP_OK(hipMalloc(&out.fb, (size_t)W * W * 4));
    HIP_OK(hipMalloc(&out.palidx, (size_t)W * W));
    const size_t big = (size_t)1 << 30;  // 1 GiB: far beyond the Infinity Cache
    u32x4* d_big;
    HIP_OK(hipMalloc(&d_big, big));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));

    const auto fill_mask = make_mask([&](int i) { return i % fill_share == 0; });
    const auto frame_mask = make_mask([&](int i) { return i % fill_share != 0; });
    struct Case { const char* name; bool fill; std::vector<uint32_t> fm, xm; int fill_wgs; };
    const int fill_cus = 256 / fill_share;
    std::vector<Case> cases = {
        {"no fill beside it", false, {}, {}, 0},
        {"fill on every CU, 256 wgs", true, {}, {}, 256},
        {"fill on every CU, 1024 wgs", true, {}, {}, 1024},
        {"fill masked, frames anywhere", true, {}, fill_mask, fill_cus * 4},
        {"fill masked, frames on the rest", true, frame_mask, fill_mask, fill_cus * 4},
        {"fill masked x2 wgs, frames rest", true, frame_mask, fill_mask, fill_cus * 8},
        {"no fill, frames on the rest", false, frame_mask, {}, 0},
    };
    // ---- how much of the write bandwidth can a fill take before the frame beside it starves? -------------------
    {
        unsigned long long* d_written;
        HIP_OK(hipMalloc(&d_written, 8));
        hipStream_t sf = masked_stream({}), sx = masked_stream({});
        std::printf("--- paced fill beside frames without their own fill (one at a time): wgs, sleep -> fill TB/s, frame us\n");
        auto run = [&](int wgs, int sleep) {
            for (int i = 0; i < 10; i++) par_render_device(ctx, sf, 0, W, &out, 1u << 28);
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemsetAsync(d_written, 0, 8, sx));
            HIP_OK(hipEventRecord(e0, sx));
            const unsigned long long ticks = 150000;  // 1.5 ms
            switch (sleep) {
                case 0: hipLaunchKernelGGL(fill_timed<0>, dim3(wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                case 1: hipLaunchKernelGGL(fill_timed<1>, dim3(wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                case 2: hipLaunchKernelGGL(fill_timed<2>, dim3(wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                case 4: hipLaunchKernelGGL(fill_timed<4>, dim3(wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                case 8: hipLaunchKernelGGL(fill_timed<8>, dim3(wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
                default: hipLaunchKernelGGL(fill_timed<16>, dim3(wgs), dim3(256), 0, sx, d_big, big / 16, ticks, d_written); break;
            }
            HIP_OK(hipEventRecord(e1, sx));
            const double t0 = now_s();
            int n = 0;
            while (now_s() - t0 < 1.0e-3) {
                par_render_device(ctx, sf, 0, W, &out, 1u << 28);
                HIP_OK(hipStreamSynchronize(sf));
                n++;
            }
            const double per = (now_s() - t0) / n * 1e6;
            HIP_OK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long wr = 0;
            HIP_OK(hipMemcpy(&wr, d_written, 8, hipMemcpyDeviceToHost));
            std::printf("wgs %5d sleep %2d: fill %5.2f TB/s  frame %7.1f us (%d frames)\n", wgs, sleep,
                        (double)wr * 1024.0 / (ms * 1e-3) / 1e12, per, n);
            HIP_OK(hipDeviceSynchronize());
        };
        for (int wgs : {32, 64, 128, 256, 512}) {
            for (int sleep : {0, 1, 2, 4, 8, 16}) run(wgs, sleep);
        }
        HIP_OK(hipStreamDestroy(sf));
        HIP_OK(hipStreamDestroy(sx));
    }
    for (unsigned flags : {1u << 28}) {
        std::printf("--- frames %s, fill share 1/%d of the CUs\n", flags ? "WITHOUT their own fill (bit 28)" : "with their own fill", fill_share);
        for (const Case& c : cases) {
            hipStream_t sf = masked_stream(c.fm), sx = masked_stream(c.xm);
            for (int i = 0; i < 20; i++) par_render_device(ctx, sf, 0, W, &out, flags);
            HIP_OK(hipDeviceSynchronize());
            const int passes = 5;  // 5 GiB: about a millisecond
            if (c.fill) {
                HIP_OK(hipEventRecord(e0, sx));
                hipLaunchKernelGGL(fill_long, dim3(c.fill_wgs), dim3(256), 0, sx, d_big, big / 16, passes);
                HIP_OK(hipEventRecord(e1, sx));
            }
            // frames for about 0.6 ms, one at a time, while the fill runs
            const double t0 = now_s();
            int n = 0;
            while (now_s() - t0 < 0.6e-3) {
                par_render_device(ctx, sf, 0, W, &out, flags);
                HIP_OK(hipStreamSynchronize(sf));
                n++;
            }
            const double per = (now_s() - t0) / n * 1e6;
            float ms = 0.f;
            double tbs = 0.0;
            if (c.fill) {
                const bool still = hipEventQuery(e1) == hipErrorNotReady;
                HIP_OK(hipEventSynchronize(e1));
                HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                tbs = (double)big * passes / (ms * 1e-3) / 1e12;
                std::printf("%-34s frame %6.1f us (%3d frames)  fill %5.2f TB/s%s\n", c.name, per, n, tbs,
                            still ? "" : "  (the fill ended before the frames did)");
            } else {
                std::printf("%-34s frame %6.1f us (%3d frames)\n", c.name, per, n);
            }
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipStreamDestroy(sf));
            HIP_OK(hipStreamDestroy(sx));
        }
    }
    par_destroy(ctx);
    return 0;
}
