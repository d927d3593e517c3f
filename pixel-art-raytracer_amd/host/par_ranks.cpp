// par_ranks.cpp — one frame sharded over the GPUs of a node, host side in C++: one process per GPU, each
// renders its row block (cut at bin rows, par_row_block) with frames in flight, and the frame is assembled on rank 0
// by an exchange enqueued on the frame's own stream right behind its render: no interpreter and no host wait anywhere
// in the per-frame path. torch.distributed (bench.py) stays the launcher/test path; this is the loop a C++ host (the
// reference's language) would run.
//
//   par_ranks --ranks N --rank R --id-file PATH [--device D] [--size S] [--prims P] [--frames F] [--inflight K]
//             [--gather tiles|tiles-copy|blocks|none] [--check]
//
// --gather tiles (default): only the screen tiles that can show a primitive travel (par_scene_tiles: every rank holds
//   the whole scene, hence the same list; sorted by bin row, so a rank's block is one contiguous run of it). Every rank
//   but rank 0 packs its run (par_tiles_pack) and sends it to rank 0 (ncclSend / ncclRecv in one group: the runs
//   differ in length); rank 0 renders its own block straight into its rows of the assembled frame and writes every
//   other row exactly once, a received tile or the background (par_tiles_assemble). At 4096^2 / 1024 primitives
//   2 536 of 10 609 tiles: 16 MB instead of 64 MiB per frame, of which rank 0's own share does not travel.
// --gather tiles-copy: the same exchange with rank 0 treating its own block like everyone's (pack, then background
//   fill and unpack of every run: par_background_fill, par_tiles_unpack) -- what a one-rank run can exercise.
// --gather blocks: ONE ncclGather of the row blocks per frame, padded to the largest block (block q of frame slot k
//   lies at q * max_block_bytes in slot k's gathered buffer on rank 0).
// --gather none: the frame stays sharded (what a consumer that reads the blocks where they are rendered sees): the
//   render's own scaling.
// Start one process per rank with the same --id-file (rank 0 writes the RCCL unique ids there, the others wait for
// them); --device defaults to R modulo the visible devices.
// --check: rank 0 compares the assembled frame with its own render of the whole frame (none: every rank its block).
// With --ranks 1 the exchange is local (tiles-copy: pack, background, unpack on the one GPU; tiles: the frame is
// rendered in place; blocks: RCCL's copy of the only block): that is what a one-GPU box can run of this path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "par_raytracer.h"

#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)
#define NCCL_OK(x)                                                                        \
    do {                                                                                  \
        ncclResult_t r_ = (x);                                                            \
        if (r_ != ncclSuccess) {                                                          \
            std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(r_)); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)
#define PAR_OK_(ctx, x)                                                                  \
    do {                                                                                 \
        int rc_ = (x);                                                                   \
        if (rc_ != PAR_OK) {                                                             \
            std::fprintf(stderr, "rank %d: %s: %s (%s)\n", g_rank, #x, par_status_string(rc_), par_last_error(ctx)); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static int g_rank = 0;

struct Slot {
    par_context* ctx = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    par_color* block = nullptr;     // this rank's rows of the frame (padded to the largest block)
    uint8_t* pal = nullptr;
    par_color* gathered = nullptr;  // rank 0, --gather blocks: ranks x max block
    par_color* packed = nullptr;    // --gather tiles: this rank's run of tiles
    par_color* inbox = nullptr;     // rank 0, --gather tiles: every rank's run, in list order
    par_color* frame = nullptr;     // rank 0, --gather tiles: the assembled frame
    par_outputs out{};
};

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    int ranks = 1, rank = 0, device = -1, size = 4096, prims = 1024, frames = 2000, inflight = 4;
    bool check = false;
    std::string id_file, gather = "tiles";
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](int& v) { if (i + 1 < argc) v = std::atoi(argv[++i]); };
        if (a == "--ranks") next(ranks);
        else if (a == "--rank") next(rank);
        else if (a == "--device") next(device);
        else if (a == "--size") next(size);
        else if (a == "--prims") next(prims);
        else if (a == "--frames") next(frames);
        else if (a == "--inflight") next(inflight);
        else if (a == "--id-file") { if (i + 1 < argc) id_file = argv[++i]; }
        else if (a == "--gather") { if (i + 1 < argc) gather = argv[++i]; }
        else if (a == "--check") check = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    g_rank = rank;
    const bool g_copy = gather == "tiles-copy";  // (the root packs and unpacks its own tiles too; tests with one rank)
    const bool g_tiles = gather == "tiles" || g_copy, g_blocks = gather == "blocks", g_none = gather == "none";
    const bool in_place = g_tiles && !g_copy;
    if (ranks < 1 || rank < 0 || rank >= ranks || inflight < 1 || inflight > 16 || frames < 1 || id_file.empty() ||
        !(g_tiles || g_blocks || g_none)) {
        std::fprintf(stderr, "usage: par_ranks --ranks N --rank R --id-file PATH [--device D] [--size S] [--prims P] "
                             "[--frames F] [--inflight K] [--gather tiles|tiles-copy|blocks|none] [--check]\n");
        return 2;
    }
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    if (device < 0) device = rank % ndev;
    HIP_OK(hipSetDevice(device));

    const int W = size, H = size, L = size;
    par_params params;
    par_default_params(&params);
    params.width = W; params.height = H; params.length = L;
    std::vector<par_aabb> aabbs((size_t)prims);
    par_light light;
    par_scene_synthetic(prims, W, H, L, 12345, aabbs.data(), &light);
    par_sprite sprite;
    par_sprite_tile_floor(&sprite);

    // row blocks: cut at bin rows, the bin rows dealt evenly; gathered with ONE count, so padded to the largest
    std::vector<int> r0((size_t)ranks), r1((size_t)ranks);
    int max_rows = 1;
    for (int q = 0; q < ranks; q++) {
        par_row_block(q, ranks, H, params.bin_size, &r0[(size_t)q], &r1[(size_t)q]);
        max_rows = std::max(max_rows, r1[(size_t)q] - r0[(size_t)q]);
    }
    const int my0 = r0[(size_t)rank], my1 = r1[(size_t)rank];
    const size_t block_px = (size_t)max_rows * W;

    // --gather tiles: the list every rank derives from the scene, and each rank's run of it (tiles of bin rows
    // [r0 / B, ceil(r1 / B)): the list is sorted by bin row)
    const int B = params.bin_size;
    std::vector<int32_t> tiles;
    std::vector<int> t_first((size_t)ranks, 0), t_count((size_t)ranks, 0);
    int32_t* d_tiles = nullptr;
    int32_t* d_map = nullptr;
    const size_t slot_px = (size_t)B * B;
    if (g_tiles) {
        int gx = 0, gy = 0, gz = 0;
        par_grid_dims(&params, &gx, &gy, &gz);
        tiles.resize((size_t)gx * gy);
        const int n = par_scene_tiles(&params, aabbs.data(), prims, tiles.data(), (int)tiles.size());
        if (n < 0) { std::fprintf(stderr, "par_scene_tiles: %s\n", par_status_string(-n)); return 1; }
        tiles.resize((size_t)n);
        for (int q = 0; q < ranks; q++) {
            const int by0 = r0[(size_t)q] / B, by1 = (r1[(size_t)q] + B - 1) / B;
            int a = 0, b = 0;
            while (a < n && (tiles[(size_t)a] >> 16) < by0) a++;
            b = a;
            while (b < n && (tiles[(size_t)b] >> 16) < by1) b++;
            if (r1[(size_t)q] <= r0[(size_t)q]) b = a;
            t_first[(size_t)q] = a;
            t_count[(size_t)q] = b - a;
        }
        HIP_OK(hipMalloc(&d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(int32_t)));
        if (n) HIP_OK(hipMemcpy(d_tiles, tiles.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
        if (rank == 0 && in_place) {  // tile -> slot of the inbox, for the one-pass assembly of the other ranks' rows
            std::vector<int32_t> map((size_t)gx * gy);
            const int rc = par_scene_tile_map(&params, tiles.data(), n, map.data(), (int)map.size());
            if (rc != PAR_OK) { std::fprintf(stderr, "par_scene_tile_map: %s\n", par_status_string(rc)); return 1; }
            HIP_OK(hipMalloc(&d_map, map.size() * sizeof(int32_t)));
            HIP_OK(hipMemcpy(d_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }

    // one communicator per frame slot (each slot's gathers run on the slot's own stream)
    std::vector<ncclUniqueId> ids((size_t)inflight);
    if (rank == 0) {
        for (auto& id : ids) NCCL_OK(ncclGetUniqueId(&id));
        const std::string tmp = id_file + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(ids.data(), sizeof(ncclUniqueId), ids.size(), f) != ids.size()) {
            std::fprintf(stderr, "cannot write %s\n", tmp.c_str());
            return 1;
        }
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) { std::fprintf(stderr, "cannot rename to %s\n", id_file.c_str()); return 1; }
    } else {
        const double t0 = now_s();
        for (;;) {
            FILE* f = std::fopen(id_file.c_str(), "rb");
            if (f) {
                const size_t n = std::fread(ids.data(), sizeof(ncclUniqueId), ids.size(), f);
                std::fclose(f);
                if (n == ids.size()) break;
            }
            if (now_s() - t0 > 120.0) { std::fprintf(stderr, "rank %d: no ids in %s after 120 s\n", rank, id_file.c_str()); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }

    std::vector<Slot> slots((size_t)inflight);
    for (size_t k = 0; k < slots.size(); k++) {
        Slot& s = slots[k];
        NCCL_OK(ncclCommInitRank(&s.comm, ranks, ids[k], rank));
        PAR_OK_(s.ctx, par_create(&params, device, &s.ctx));
        PAR_OK_(s.ctx, par_set_sprites(s.ctx, &sprite, 1));
        PAR_OK_(s.ctx, par_set_entities(s.ctx, aabbs.data(), nullptr, prims));
        PAR_OK_(s.ctx, par_set_light(s.ctx, &light));
        HIP_OK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_OK(hipMalloc(&s.block, block_px * sizeof(par_color)));
        HIP_OK(hipMalloc(&s.pal, block_px));
        if (rank == 0 && g_blocks) HIP_OK(hipMalloc(&s.gathered, block_px * sizeof(par_color) * (size_t)ranks));
        if (g_tiles) {
            HIP_OK(hipMalloc(&s.packed, std::max<size_t>((size_t)t_count[(size_t)rank], 1) * slot_px * sizeof(par_color)));
            if (rank == 0) {
                HIP_OK(hipMalloc(&s.inbox, std::max<size_t>(tiles.size(), 1) * slot_px * sizeof(par_color)));
                HIP_OK(hipMalloc(&s.frame, (size_t)W * H * sizeof(par_color)));
            }
        }
        s.out.fb = s.block;
        if (rank == 0 && in_place) s.out.fb = s.frame + (size_t)my0 * W;  // rank 0 renders into its rows of the frame
        s.out.palidx = s.pal;
    }

    auto submit = [&](int f) -> int {
        Slot& s = slots[(size_t)f % slots.size()];
        if (my1 > my0) {
            PAR_OK_(s.ctx, par_render_device(s.ctx, s.stream, my0, my1, &s.out, inflight > 1 ? PAR_RENDER_PIPELINED : 0u));
        }
        // the frame's only exchange, behind the render on the same stream
        if (g_blocks) {  // its row blocks to rank 0
            NCCL_OK(ncclGather(s.block, s.gathered, block_px * sizeof(par_color), ncclUint8, 0, s.comm, s.stream));
        } else if (g_tiles) {  // its tiles to rank 0, which writes the background itself
            const int mine = t_count[(size_t)rank];
            if (mine && !(rank == 0 && in_place)) {
                PAR_OK_(s.ctx, par_tiles_pack(&params, s.stream, d_tiles + t_first[(size_t)rank], mine, s.block, my0, my1,
                                              rank == 0 ? s.inbox + (size_t)t_first[0] * slot_px : s.packed));
            }
            if (ranks > 1) {
                NCCL_OK(ncclGroupStart());
                if (rank == 0) {
                    for (int q = 1; q < ranks; q++) {
                        if (t_count[(size_t)q]) {
                            NCCL_OK(ncclRecv(s.inbox + (size_t)t_first[(size_t)q] * slot_px,
                                             (size_t)t_count[(size_t)q] * slot_px * sizeof(par_color), ncclUint8, q, s.comm, s.stream));
                        }
                    }
                } else if (mine) {
                    NCCL_OK(ncclSend(s.packed, (size_t)mine * slot_px * sizeof(par_color), ncclUint8, 0, s.comm, s.stream));
                }
                NCCL_OK(ncclGroupEnd());
            }
            if (rank == 0 && in_place) {  // every row rank 0 did not render, written once: tile or background
                if (my0 > 0) PAR_OK_(s.ctx, par_tiles_assemble(&params, s.stream, d_map, s.inbox, s.frame, 0, my0));
                if (my1 < H) PAR_OK_(s.ctx, par_tiles_assemble(&params, s.stream, d_map, s.inbox, s.frame, my1, H));
            } else if (rank == 0) {
                PAR_OK_(s.ctx, par_background_fill(&params, s.stream, s.frame, H));
                PAR_OK_(s.ctx, par_tiles_unpack(&params, s.stream, d_tiles, (int)tiles.size(), s.inbox, s.frame));
            }
        }
        return 0;
    };
    int* d_flag = nullptr;
    HIP_OK(hipMalloc(&d_flag, sizeof(int)));
    HIP_OK(hipMemset(d_flag, 0, sizeof(int)));
    auto barrier = [&]() -> int {  // every rank's work so far is done, on every rank
        HIP_OK(hipDeviceSynchronize());
        NCCL_OK(ncclAllReduce(d_flag, d_flag, 1, ncclInt32, ncclSum, slots[0].comm, slots[0].stream));
        HIP_OK(hipStreamSynchronize(slots[0].stream));
        return 0;
    };

    const int warm = std::min(frames, 200);
    for (int f = 0; f < warm; f++) if (submit(f)) return 1;
    if (barrier()) return 1;
    const double t0 = now_s();
    for (int f = 0; f < frames; f++) if (submit(f)) return 1;
    if (barrier()) return 1;
    const double dt = now_s() - t0;

    int bad = 0;
    if (check && (rank == 0 || g_none)) {
        // the assembled frame of the last frame's slot (none: this rank's block) against this rank's own render of the
        // whole frame
        Slot& s = slots[(size_t)(frames - 1) % slots.size()];
        std::vector<par_color> got((size_t)W * H), exp((size_t)W * H);
        par_outputs ho{};
        ho.fb = exp.data();
        PAR_OK_(s.ctx, par_render(s.ctx, &ho, 0));
        if (g_blocks) {
            for (int q = 0; q < ranks; q++) {
                const size_t n = (size_t)(r1[(size_t)q] - r0[(size_t)q]) * W;
                if (n) HIP_OK(hipMemcpy(got.data() + (size_t)r0[(size_t)q] * W, s.gathered + (size_t)q * block_px, n * sizeof(par_color), hipMemcpyDeviceToHost));
            }
            bad = std::memcmp(got.data(), exp.data(), got.size() * sizeof(par_color)) != 0;
        } else if (g_tiles) {
            HIP_OK(hipMemcpy(got.data(), s.frame, got.size() * sizeof(par_color), hipMemcpyDeviceToHost));
            bad = std::memcmp(got.data(), exp.data(), got.size() * sizeof(par_color)) != 0;
        } else {
            const size_t n = (size_t)(my1 - my0) * W;
            if (n) HIP_OK(hipMemcpy(got.data(), s.block, n * sizeof(par_color), hipMemcpyDeviceToHost));
            bad = n && std::memcmp(got.data(), exp.data() + (size_t)my0 * W, n * sizeof(par_color)) != 0;
        }
        std::printf("check (rank %d, gather %s): %s\n", rank, gather.c_str(), bad ? "FAILED" : "ok");
    }
    if (rank == 0) {
        std::printf("{\"host\": \"C++ ranks\", \"ranks\": %d, \"size\": %d, \"prims\": %d, \"frames\": %d, \"inflight\": %d, "
                    "\"us_per_frame\": %.2f, \"frames_per_s\": %.0f, \"mrays_per_s\": %.0f, \"gather\": \"%s\", "
                    "\"bytes_to_rank0_per_frame\": %zu, \"tiles\": %zu, \"rows_of_rank0\": [%d, %d]}\n",
                    ranks, size, prims, frames, inflight, 1e6 * dt / frames, frames / dt, 2.0 * W * H * frames / dt / 1e6,
                    gather.c_str(),
                    g_blocks ? block_px * sizeof(par_color) * (size_t)(ranks - 1)
                             : (g_tiles ? (tiles.size() - (size_t)t_count[0]) * slot_px * sizeof(par_color) : (size_t)0),
                    tiles.size(), my0, my1);
    }
    for (auto& s : slots) {
        (void)ncclCommDestroy(s.comm);
        par_destroy(s.ctx);
        (void)hipFree(s.block);
        (void)hipFree(s.pal);
        if (s.gathered) (void)hipFree(s.gathered);
        if (s.packed) (void)hipFree(s.packed);
        if (s.inbox) (void)hipFree(s.inbox);
        if (s.frame) (void)hipFree(s.frame);
        (void)hipStreamDestroy(s.stream);
    }
    (void)hipFree(d_flag);
    if (d_tiles) (void)hipFree(d_tiles);
    if (d_map) (void)hipFree(d_map);
    return bad ? 1 : 0;
}
