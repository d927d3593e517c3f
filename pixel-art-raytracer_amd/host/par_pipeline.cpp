// par_pipeline.cpp — the frames-in-flight render loop in host C++ over the C ABI (what pipeline.py does from Python).
//
// K contexts with the same scene, K HIP streams, K sets of device output buffers, used round-robin like a swap chain;
// optionally every primitive moves by the reference's +-5 steps each frame (alt:643-678), sent to the slot's context
// with par_update_aabbs_async. The reference's host is C++, so this is the loop a maintainer would write; it also
// shows the frame rate without an interpreter in the submit path.
//
//   par_pipeline [--size S] [--prims N] [--frames F] [--inflight K] [--threads T] [--moving] [--check] [--flags X]
//                [--stamps F0] [--block K]
//
// --flags X: render flags for every frame (the timing-experiment bits of par_raytracer.h; the output is then wrong).
// --stamps F0 (with PAR_DEBUG_STAMPS=1 in the environment): the K frames from F0 on note the GPU's 100 MHz clock at
// every workgroup's start and end; afterwards the span of each of their kernels is printed (a timeline of the
// frames in flight without a profiler in the way).
//
// --check renders the last K frames once more through the blocking host path (par_render) and compares.
// HIP streams share a few hardware queues, and two streams on one queue do not overlap: the streams are picked by a
// short measurement from a pool of candidates (as pipeline.py does).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "par_raytracer.h"

// internal profiling aid of the library (not in the public header)
extern "C" int par_debug_read_stamps(par_context* ctx, unsigned long long* out, size_t count);

#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                                    \
        }                                                                                \
    } while (0)
#define PAR_OK_(ctx, x)                                                                  \
    do {                                                                                 \
        int rc_ = (x);                                                                   \
        if (rc_ != PAR_OK) {                                                             \
            std::fprintf(stderr, "%s: %s (%s)\n", #x, par_status_string(rc_), par_last_error(ctx)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

struct Slot {
    par_context* ctx = nullptr;
    hipStream_t stream = nullptr;
    par_color* fb = nullptr;
    uint8_t* pal = nullptr;
    par_outputs out{};
};

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    int block = 0;
    int size = 4096, prims = 1024, frames = 2000, inflight = 4;
    bool moving = false, check = false;
    int stamps_from = -1, threads = 1;
    unsigned all_flags = 0;
    std::string scene = "synthetic";
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](int& v) { if (i + 1 < argc) v = std::atoi(argv[++i]); };
        if (a == "--size") next(size);
        else if (a == "--prims") next(prims);
        else if (a == "--frames") next(frames);
        else if (a == "--inflight") next(inflight);
        else if (a == "--moving") moving = true;
        else if (a == "--check") check = true;
        else if (a == "--stamps") next(stamps_from);
        else if (a == "--threads") next(threads);
        else if (a == "--block") next(block);
        else if (a == "--scene") { if (i + 1 < argc) scene = argv[++i]; }
        else if (a == "--flags") { int v = 0; next(v); all_flags = (unsigned)v; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (inflight < 1 || inflight > 16 || frames < 1) return 2;
    // --scene synthetic (default; --size, --prims) | floor (a full floor of tiles at --size: every pixel covered) |
    //         graybox (the reference's own 480x320 world, alt:517-599)
    int W = size, H = size, L = size;
    par_params params;
    par_default_params(&params);
    std::vector<par_aabb> aabbs;
    par_light light;
    if (scene == "graybox") {
        W = 480; H = 320; L = 320;
        aabbs.resize((size_t)par_scene_graybox(W, L, nullptr, 0));
        par_scene_graybox(W, L, aabbs.data(), (int)aabbs.size());
        light.x = 480; light.y = 160; light.z = 80; light.radius = 10;  // alt:625-626
    } else if (scene == "floor") {
        for (int i = 0; i < W / 20; i++) {
            for (int j = 0; j < L / 20; j++) aabbs.push_back(par_aabb{(int16_t)(i * 20), 0, (int16_t)(j * 20), 20, 20, 20, {0, 0}});
        }
        light.x = (int16_t)(5 * W / 8); light.y = (int16_t)(H / 2); light.z = (int16_t)(L / 4); light.radius = 10;
    } else {
        aabbs.resize((size_t)prims);
        par_scene_synthetic(prims, W, H, L, 12345, aabbs.data(), &light);
    }
    prims = (int)aabbs.size();
    size = W;
    params.width = W; params.height = H; params.length = L;
    par_sprite sprite;
    par_sprite_tile_floor(&sprite);
    // +-5 or 0 per axis and frame, from a fixed little generator
    std::vector<int16_t> vel((size_t)prims * 3);
    uint32_t lcg = 12345u;
    for (auto& v : vel) { lcg = lcg * 1664525u + 1013904223u; v = (int16_t)(((lcg >> 16) % 3) * 5 - 5); }

    const size_t npix = (size_t)W * H;
    std::vector<Slot> slots((size_t)inflight);
    for (auto& s : slots) {
        PAR_OK_(s.ctx, par_create(&params, 0, &s.ctx));
        PAR_OK_(s.ctx, par_set_sprites(s.ctx, &sprite, 1));
        PAR_OK_(s.ctx, par_set_entities(s.ctx, aabbs.data(), nullptr, prims));
        PAR_OK_(s.ctx, par_set_light(s.ctx, &light));
        HIP_OK(hipMalloc(&s.fb, npix * sizeof(par_color)));
        HIP_OK(hipMalloc(&s.pal, npix));
        s.out.fb = s.fb;
        s.out.palidx = s.pal;
    }
    // streams that overlap pairwise
    std::vector<hipStream_t> cand((size_t)(inflight > 1 ? 3 * inflight : 1));
    for (auto& c : cand) HIP_OK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    auto probe = [&](hipStream_t sa, hipStream_t sb) -> double {
        (void)hipDeviceSynchronize();
        const double t0 = now_s();
        for (int i = 0; i < 24; i++) {
            Slot& s = slots[(size_t)(i & 1) % slots.size()];
            par_render_device(s.ctx, (i & 1) ? sb : sa, 0, H, &s.out, 0);
        }
        (void)hipDeviceSynchronize();
        return now_s() - t0;
    };
    std::vector<hipStream_t> chosen{cand[0]};
    if (inflight > 1) {
        probe(cand[0], cand[0]);
        const double serial = std::min(probe(cand[0], cand[0]), probe(cand[0], cand[0]));
        for (size_t c = 1; c < cand.size() && (int)chosen.size() < inflight; c++) {
            bool ok = true;
            for (hipStream_t x : chosen) ok = ok && std::min(probe(x, cand[c]), probe(x, cand[c])) < 0.85 * serial;
            if (ok) chosen.push_back(cand[c]);
        }
        for (size_t c = 0; c < cand.size() && (int)chosen.size() < inflight; c++) {
            if (std::find(chosen.begin(), chosen.end(), cand[c]) == chosen.end()) chosen.push_back(cand[c]);
        }
    }
    for (size_t k = 0; k < slots.size(); k++) slots[k].stream = chosen[k];

    std::vector<par_aabb> cur = aabbs;
    auto scene_of = [&](int f, std::vector<par_aabb>& dst) {
        dst = aabbs;
        if (!moving) return;
        for (int i = 0; i < prims; i++) {
            dst[(size_t)i].px = (int16_t)(dst[(size_t)i].px + vel[(size_t)i * 3 + 0] * f);
            dst[(size_t)i].py = (int16_t)(dst[(size_t)i].py + vel[(size_t)i * 3 + 1] * f);
            dst[(size_t)i].pz = (int16_t)(dst[(size_t)i].pz + vel[(size_t)i * 3 + 2] * f);
        }
    };
    auto submit = [&](int f) -> int {
        Slot& s = slots[(size_t)f % slots.size()];
        if (moving) {
            scene_of(f, cur);
            PAR_OK_(s.ctx, par_update_aabbs_async(s.ctx, cur.data(), 0, prims, s.stream));
        }
        const unsigned fl = all_flags | (inflight > 1 ? (unsigned)PAR_RENDER_PIPELINED : 0u) |
                            ((stamps_from >= 0 && f >= stamps_from && f < stamps_from + inflight) ? (1u << 29) : 0u);
        PAR_OK_(s.ctx, par_render_device(s.ctx, s.stream, 0, H, &s.out, fl));
        return 0;
    };
    const int warm = std::min(frames, 200);
    for (int f = 0; f < warm; f++) if (submit(f)) return 1;
    HIP_OK(hipDeviceSynchronize());
    if (block > 0) {
        // --block K: blocks of K frames, each started on an idle device and waited for (what a bench step loop of
        // K steps between two synchronisations sees), with an event behind every frame: when did frame i of the
        // block complete? Median over 25 blocks.
        const int n_blocks = 25;
        std::vector<hipEvent_t> ev((size_t)block + 1);
        for (auto& e : ev) HIP_OK(hipEventCreate(&e));
        std::vector<std::vector<float>> done((size_t)block, std::vector<float>());
        std::vector<double> wall;
        for (int b = 0; b < n_blocks; b++) {
            HIP_OK(hipDeviceSynchronize());
            const double tb = now_s();
            HIP_OK(hipEventRecord(ev[0], slots[0].stream));
            for (int f = 0; f < block; f++) {
                if (submit(f)) return 1;
                HIP_OK(hipEventRecord(ev[(size_t)f + 1], slots[(size_t)f % slots.size()].stream));
            }
            HIP_OK(hipDeviceSynchronize());
            wall.push_back(1e6 * (now_s() - tb));
            for (int f = 0; f < block; f++) {
                float ms = 0.f;
                HIP_OK(hipEventElapsedTime(&ms, ev[0], ev[(size_t)f + 1]));
                done[(size_t)f].push_back(ms * 1e3f);
            }
        }
        std::sort(wall.begin(), wall.end());
        std::printf("block of %d frames, %d in flight: median wall %.1f us = %.2f us per frame\ncompleted at (us):", block,
                    inflight, wall[wall.size() / 2], wall[wall.size() / 2] / block);
        for (int f = 0; f < block; f++) {
            std::sort(done[(size_t)f].begin(), done[(size_t)f].end());
            std::printf(" %.0f", done[(size_t)f][done[(size_t)f].size() / 2]);
        }
        std::printf("\n");
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    const double t0 = now_s();
    if (threads <= 1 || moving) {
        for (int f = 0; f < frames; f++) if (submit(f)) return 1;
    } else {
        // one submitting thread per group of frame slots (a slot is only ever touched by its own thread)
        std::atomic<int> failed{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) {
            pool.emplace_back([&, t] {
                (void)hipSetDevice(0);
                for (int f = 0; f < frames; f++) {
                    if ((f % inflight) % threads != t) continue;
                    if (submit(f)) { failed = 1; return; }
                }
            });
        }
        for (auto& th : pool) th.join();
        if (failed) return 1;
    }
    const double t_enq = now_s() - t0;  // the host's share: time until the last frame was enqueued
    HIP_OK(hipDeviceSynchronize());
    const double dt = now_s() - t0;
    std::printf("{\"host\": \"C++\", \"size\": %d, \"prims\": %d, \"moving\": %s, \"frames\": %d, \"inflight\": %d, "
                "\"us_per_frame\": %.2f, \"host_enqueue_us_per_frame\": %.2f, \"frames_per_s\": %.0f, "
                "\"mrays_per_s\": %.0f}\n",
                size, prims, moving ? "true" : "false", frames, inflight, 1e6 * dt / frames, 1e6 * t_enq / frames,
                frames / dt, 2.0 * W * H * frames / dt / 1e6);

    if (stamps_from >= 0) {
        // rows: the frame's kernels; per workgroup 8 slots, slot 0 = start, slot 7 = end (100 MHz ticks)
        static const char* names[6] = {"insert+fill", "resolve+fill", "columns+fill", "render_items", "render_overflow", "render_tiles"};
        const size_t rows = 6, wgs = 8192, n = rows * wgs * 8;
        std::vector<std::vector<unsigned long long>> st(slots.size(), std::vector<unsigned long long>(n));
        unsigned long long t_min = ~0ull;
        for (size_t k = 0; k < slots.size(); k++) {
            if (par_debug_read_stamps(slots[k].ctx, st[k].data(), n) != PAR_OK) {
                std::fprintf(stderr, "no stamps: run with PAR_DEBUG_STAMPS=1\n");
                return 1;
            }
            for (size_t i = 0; i < rows * wgs; i++) {
                if (st[k][i * 8]) t_min = std::min(t_min, st[k][i * 8]);
            }
        }
        for (size_t k = 0; k < slots.size(); k++) {
            const int f = stamps_from + (int)((k + slots.size() - (size_t)stamps_from % slots.size()) % slots.size());
            for (size_t r = 0; r < rows; r++) {
                unsigned long long a0 = ~0ull, a1 = 0, last_start = 0, life = 0;
                size_t cnt = 0;
                for (size_t i = 0; i < wgs; i++) {
                    const unsigned long long b = st[k][(r * wgs + i) * 8], e = st[k][(r * wgs + i) * 8 + 7];
                    if (!b) continue;
                    cnt++;
                    a0 = std::min(a0, b);
                    last_start = std::max(last_start, b);
                    a1 = std::max(a1, std::max(b, e));
                    if (e > b) life += e - b;
                }
                if (!cnt) continue;
                static const int waves[6] = {4, 4, 2, 4, 4, 4};  // wavefronts per workgroup of each kernel
                std::printf("stamps frame %d slot %zu %-16s start %8.2f  last-wg-start %8.2f  end %8.2f  (%.2f us, %zu wgs, "
                            "%.0f wavefront-us)\n",
                            f, k, names[r], (a0 - t_min) * 0.01, (last_start - t_min) * 0.01, (a1 - t_min) * 0.01,
                            (a1 - a0) * 0.01, cnt, life * 0.01 * waves[r]);
            }
        }
    }

    int bad = 0;
    if (check) {
        std::vector<par_color> got(npix), exp(npix);
        std::vector<uint8_t> gpal(npix), epal(npix);
        for (int f = std::max(0, frames - inflight); f < frames; f++) {
            Slot& s = slots[(size_t)f % slots.size()];
            HIP_OK(hipMemcpy(got.data(), s.fb, npix * sizeof(par_color), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(gpal.data(), s.pal, npix, hipMemcpyDeviceToHost));
            // the same scene through the blocking host path of another slot's context
            Slot& o = slots[(size_t)(f + 1) % slots.size()];
            std::vector<par_aabb> sc;
            scene_of(f, sc);
            PAR_OK_(o.ctx, par_update_aabbs(o.ctx, sc.data(), 0, prims));
            par_outputs ho{};
            ho.fb = exp.data();
            ho.palidx = epal.data();
            PAR_OK_(o.ctx, par_render(o.ctx, &ho, 0));
            if (std::memcmp(got.data(), exp.data(), npix * sizeof(par_color)) != 0 ||
                std::memcmp(gpal.data(), epal.data(), npix) != 0) {
                std::fprintf(stderr, "frame %d differs from the blocking render of the same scene\n", f);
                bad++;
            }
        }
        std::printf("check: %s\n", bad ? "FAILED" : "ok");
    }
    for (auto& s : slots) {
        par_destroy(s.ctx);
        (void)hipFree(s.fb);
        (void)hipFree(s.pal);
    }
    for (auto& c : cand) (void)hipStreamDestroy(c);
    return bad ? 1 : 0;
}
