// par_context.hip — the renderer context and the C ABI (include/par_raytracer.h) over the HIP kernels.
//
// Host-side mirror of what the reference's `main` does around its render call (alt = src/alternative.cpp):
// it owns the work arrays (alt:503-517), takes the scene the caller built (alt:517-599, 619-626), and runs one
// frame = bin + trace + shade (alt:690-760) per render call. There is no CPU rendering path in this library.
#include <algorithm>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "par_internal.h"

// The bins one entity is inserted into (cull and ranges of alt:202-240, computed once per AABB the host is given):
// bin columns [x0, x1) x [y0, y1), nz bins deep; empty when x1 <= x0. Everything the host sizes launches and lists
// with follows from it.
struct par_footprint {
    int16_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    int32_t nz = 0;
    int32_t items = 0;  // render work items (64-pixel chunks) the entity can cause, see footprint_of
    int16_t px = 0, ex = 0;     // its sprite rectangle on screen (alt:310-317): columns [px, px + ex),
    int32_t row0 = 0, rh = 0;   // rows [row0, row0 + rh) = H - (py + ey + pz + ez) .. H - (py + pz)
    int32_t cols() const { return (x1 - x0) * (y1 - y0); }
    int64_t pairs() const { return (int64_t)cols() * nz; }  // (entity, bin) insertions, alt:243-267
};

struct par_context {
    par_params params{};
    int device = 0;
    int gx = 0, gy = 0, gz = 0, volume = 0;
    hipStream_t stream = nullptr;   // used by the synchronous host-buffer entry points

    // host mirrors
    std::vector<par_aabb> h_aabbs;
    std::vector<par_footprint> h_fp;  // per entity: the bins it is inserted into (alt:202-240) and what follows from them
    int64_t total_pairs = 0;
    int64_t total_cols = 0;        // >= the occupied columns of the frame
    int64_t total_items = 0;       // >= the render work items (64-pixel chunks) of the frame, see footprint_of
    // The same three from the entities' EXTENTS alone (bound_of: wherever an entity stands it reaches no more): what
    // pools and lists are sized by, and what a frame's launches are sized by while the exact bookkeeping above is
    // stale -- par_update_aabbs_async does not keep it (a moving scene would pay the cull and range arithmetic of
    // every moved entity on the host, every frame); the next blocking call brings it up to date (refresh_exact).
    int64_t bound_pairs = 0, bound_cols = 0, bound_items = 0;
    bool exact_stale = false;
    // The per-column histograms alone lag behind the footprints (par_graph_stage keeps footprints and totals exact --
    // it has to refuse a frame the captured launches cannot hold -- but not the histograms, which cost a moving scene
    // more host time per frame than everything else the stage does; a captured graph does not read them).
    bool hist_stale = false;
    std::vector<int32_t> h_colpairs;  // (entity, bin) pairs per screen column (>= its occupied bins, >= its entries)
    int64_t cols_over = 0;            // columns with more pairs than a column record is sure to hold
    // 64-pixel chunks of the entities' sprite rectangles per screen column (what the column kernel adds up, over the
    // visible entries only, to choose between visiting a column entry by entry and as a whole tile), and the columns
    // where that reaches the tile's own chunks: only those can be visited as tiles
    std::vector<int32_t> h_colchunks;
    int64_t cols_tileable = 0;
    int n_entities = 0, n_sprites = 0, max_sprite_id = 0;
    bool have_light = false, have_entities = false;
    par_light light{};
    int set = 0;  // head/count/node set the NEXT frame uses
    hipStream_t last_stream = nullptr;  // stream of the most recent asynchronous render (scene updates wait for it)
    bool has_last_stream = false;

    // device
    par_aabb* d_aabbs = nullptr;
    int32_t* d_sprite_ids = nullptr;
    par_sprite* d_sprites = nullptr;
    par_color* d_palette = nullptr;
    par_texel* d_texinfo = nullptr;
    unsigned long long* d_ray_counter = nullptr;
    uint8_t* d_scratch_lit = nullptr;  // lit plane when every ray is traced but the caller wants no lit plane
    size_t scratch_lit_bytes = 0;
    par_grid_dev grid{};
    int aabb_capacity = 0;

    // device output planes for the host-buffer entry points
    void* d_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t d_out_bytes[5] = {0, 0, 0, 0, 0};

    // hipGraph path: one executable graph per grid set, a pinned staging area they copy from
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
    hipGraph_t graph[2] = {nullptr, nullptr};
    // Each graph copies the scene from a pinned staging area of its own (the copy node reads it when the graph RUNS,
    // which may be long after it was launched): the area of set s is rewritten only after the event recorded behind
    // set s's last launch. `stage_lo/hi`: entities changed since the area last matched the host mirror.
    par_aabb* pin_aabbs[2] = {nullptr, nullptr};
    par_frame_dyn* pin_dyn[2] = {nullptr, nullptr};
    hipEvent_t ev_graph[2] = {nullptr, nullptr};
    bool ev_graph_pending[2] = {false, false};
    int stage_lo[2] = {0, 0}, stage_hi[2] = {0, 0};
    par_aabb* pin_update = nullptr;   // staging of par_update_aabbs_async
    int pin_update_capacity = 0;
    hipEvent_t ev_update = nullptr;   // its last copy
    bool ev_update_pending = false;
    hipStream_t update_stream = nullptr;
    par_frame_dyn* d_dyn = nullptr;
    int graph_set = 0;
    int64_t graph_pair_bound = 0;  // (entity, bin) pairs a captured graph's launch grids can take
    int64_t graph_item_bound = 0;  // ... and render work items

    bool timed_tiles = false, timed_overflow = false, timed_both = false;  // the last timed frame launched these kernels
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    par_frame_stats stats{};
    unsigned last_flags = 0;
    std::string err;
};

namespace {

constexpr size_t kPlaneElem[5] = {sizeof(par_color), sizeof(par_pixel), 1, sizeof(float), 1};

int fail(par_context* c, int status, const std::string& msg) {
    if (c) {
        try {
            c->err = msg;
        } catch (...) {  // (the message is a convenience; the status is the contract)
        }
    }
    return status;
}

// No exception crosses the C boundary (par_raytracer.h): every entry point that allocates on the host runs its
// body through this. PAR_TEST_BAD_ALLOC=1 (tests) makes the guarded bodies fail as an exhausted heap would.
void test_alloc_hook() {
    const char* e = std::getenv("PAR_TEST_BAD_ALLOC");
    if (e && e[0] == '1') throw std::bad_alloc();
}

template <class F>
int guarded(par_context* ctx, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(ctx, PAR_ERR_OOM, "host allocation failed");
    } catch (...) {
        return fail(ctx, PAR_ERR_HIP, "unexpected C++ exception");
    }
}

int hip_fail(par_context* c, hipError_t e, const char* what) {
    return fail(c, e == hipErrorOutOfMemory ? PAR_ERR_OOM : PAR_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}

#define PAR_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

// The cull and range math of alt:202-240 for one AABB. `items`: the render work items it can cause: its sprite
// rectangle (ex wide, ey + ez tall, alt:310-317) is cut by the screen columns it reaches into that many pieces, each
// visited in whole 64-pixel chunks: sum of ceil(area_i / 64) <= floor(area / 64) + pieces. (A column switches to
// visiting its whole tile only when that takes fewer chunks.)
par_footprint footprint_of(const par_context* c, const par_aabb& a) {
    par_footprint f;
    const int W = c->params.width, H = c->params.height, L = c->params.length, B = c->params.bin_size;
    const int minx = a.px, miny = a.py, minz = a.pz;
    const int maxx = minx + a.ex, maxy = miny + a.ey, maxz = minz + a.ez;
    if ((maxx < 0) || (minx >= W) || (maxy < 0 - maxz) || (miny >= H - minz + B) || (maxz < -a.ez - B) ||
        (minz > L + B)) {
        return f;
    }
    const int x0 = std::max(0, minx / B), y0 = std::max(0, (H - maxy - maxz) / B), z0 = std::max(0, minz / B);
    const int x1 = std::min(c->gx, (maxx + B - 1) / B), y1 = std::min(c->gy, (H - miny - minz + B - 1) / B);
    const int z1 = std::min(c->gz, (maxz + B - 1) / B);
    if (x1 <= x0 || y1 <= y0 || z1 <= z0) return f;
    f.x0 = (int16_t)x0; f.x1 = (int16_t)x1; f.y0 = (int16_t)y0; f.y1 = (int16_t)y1;
    f.nz = z1 - z0;
    f.items = (int32_t)((int)a.ex * ((int)a.ey + (int)a.ez) / 64 + f.cols());
    f.px = a.px; f.ex = a.ex;
    f.row0 = H - ((int)a.py + a.ey + a.pz + a.ez);
    f.rh = (int)a.ey + (int)a.ez;
    return f;
}

// What an entity of these extents can cause at most, wherever it stands: an interval of length d meets at most
// ceil(d / B) + 1 bins of width B (alt:222-240), its sprite rectangle is ex x (ey + ez) pixels (alt:310-317).
struct par_bound {
    int64_t pairs, cols, items;
};
par_bound bound_of(const par_context* c, const par_aabb& a) {
    const int B = c->params.bin_size;
    const int64_t nx = std::min<int64_t>(c->gx, ((int)a.ex + B - 1) / B + 1);
    const int64_t ny = std::min<int64_t>(c->gy, ((int)a.ey + (int)a.ez + B - 1) / B + 1);
    const int64_t nz = std::min<int64_t>(c->gz, ((int)a.ez + B - 1) / B + 1);
    return par_bound{nx * ny * nz, nx * ny, (int64_t)a.ex * ((int)a.ey + (int)a.ez) / 64 + nx * ny};
}

// Adds (sign = +1) or removes (-1) a footprint's pairs in the per-column histogram. A column whose pairs exceed
// PAR_COL_NB (<= PAR_COL_ENT) may overflow its record; while there is none, no column can, and the frame needs no
// launch for the overflow list.
void col_hist(par_context* c, const par_footprint& f, int sign) {
    constexpr int kSure = PAR_COL_NB < PAR_COL_ENT ? PAR_COL_NB : PAR_COL_ENT;
    const int W = c->params.width, H = c->params.height, B = c->params.bin_size;
    for (int x = f.x0; x < f.x1; x++) {
        const int cx0 = x * B, tw = std::min(B, W - cx0);
        const int w = std::min(f.px + f.ex, cx0 + tw) - std::max((int)f.px, cx0);
        for (int y = f.y0; y < f.y1; y++) {
            int32_t& n = c->h_colpairs[(size_t)x * c->gy + y];
            const bool was = n > kSure;
            n += sign * f.nz;
            c->cols_over += (int)(n > kSure) - (int)was;
            const int ry0 = y * B, th = std::min(B, H - ry0);
            const int h = std::min(f.row0 + f.rh, ry0 + th) - std::max(f.row0, ry0);
            if (w > 0 && h > 0) {
                const int tile_chunks = (tw * th + 63) / 64;
                int32_t& k = c->h_colchunks[(size_t)x * c->gy + y];
                const bool could = k >= tile_chunks;
                k += sign * ((w * h + 63) / 64);
                c->cols_tileable += (int)(k >= tile_chunks) - (int)could;
            }
        }
    }
}

// What an update of aabbs[first, first + n) does to the host's bookkeeping, in two steps: `plan` computes the new
// footprints and totals (so that the caller can grow pools or refuse before anything changes), `commit` applies them.
struct par_update_plan {
    std::vector<par_footprint> fp;
    int64_t pairs = 0, cols = 0, items = 0;  // the new totals
};

void plan_update(const par_context* c, const par_aabb* aabbs, int first, int n, par_update_plan* plan) {
    plan->fp.resize((size_t)n);
    plan->pairs = c->total_pairs;
    plan->cols = c->total_cols;
    plan->items = c->total_items;
    for (int i = 0; i < n; i++) {
        const par_footprint f = footprint_of(c, aabbs[i]);
        const par_footprint& old = c->h_fp[(size_t)(first + i)];
        plan->fp[(size_t)i] = f;
        plan->pairs += f.pairs() - old.pairs();
        plan->cols += f.cols() - old.cols();
        plan->items += f.items - old.items;
    }
}

// The bound totals after aabbs[first, first + n) replace the entities there (extents rarely change: then nothing does).
void bounds_after(const par_context* c, const par_aabb* aabbs, int first, int n, par_bound* total) {
    *total = par_bound{c->bound_pairs, c->bound_cols, c->bound_items};
    for (int i = 0; i < n; i++) {
        const par_aabb& old = c->h_aabbs[(size_t)(first + i)];
        if (old.ex == aabbs[i].ex && old.ey == aabbs[i].ey && old.ez == aabbs[i].ez) continue;
        const par_bound o = bound_of(c, old), b = bound_of(c, aabbs[i]);
        total->pairs += b.pairs - o.pairs;
        total->cols += b.cols - o.cols;
        total->items += b.items - o.items;
    }
}

void commit_update(par_context* c, const par_aabb* aabbs, int first, int n, const par_update_plan& plan) {
    par_bound bt;
    bounds_after(c, aabbs, first, n, &bt);
    c->bound_pairs = bt.pairs; c->bound_cols = bt.cols; c->bound_items = bt.items;
    for (int i = 0; i < n; i++) {
        par_footprint& slot = c->h_fp[(size_t)(first + i)];
        col_hist(c, slot, -1);
        col_hist(c, plan.fp[(size_t)i], +1);
        slot = plan.fp[(size_t)i];
        c->h_aabbs[(size_t)(first + i)] = aabbs[i];
    }
    c->total_pairs = plan.pairs;
    c->total_cols = plan.cols;
    c->total_items = plan.items;
}

// The exact bookkeeping (footprints, totals, per-column histograms) from the host's copy of the AABBs, after
// asynchronous updates left it stale.
void refresh_exact(par_context* c) {
    if (!c->exact_stale && !c->hist_stale) return;
    c->h_colpairs.assign((size_t)c->gx * c->gy, 0);
    c->h_colchunks.assign((size_t)c->gx * c->gy, 0);
    c->cols_over = 0;
    c->cols_tileable = 0;
    c->total_pairs = c->total_cols = c->total_items = 0;
    for (int i = 0; i < c->n_entities; i++) {
        const par_footprint f = footprint_of(c, c->h_aabbs[(size_t)i]);
        c->h_fp[(size_t)i] = f;
        col_hist(c, f, +1);
        c->total_pairs += f.pairs();
        c->total_cols += f.cols();
        c->total_items += f.items;
    }
    c->exact_stale = false;
    c->hist_stale = false;
}

// commit_update without the per-column histograms (they go stale: hist_stale).
void commit_update_totals(par_context* c, const par_aabb* aabbs, int first, int n, const par_update_plan& plan) {
    par_bound bt;
    bounds_after(c, aabbs, first, n, &bt);
    c->bound_pairs = bt.pairs; c->bound_cols = bt.cols; c->bound_items = bt.items;
    for (int i = 0; i < n; i++) {
        c->h_fp[(size_t)(first + i)] = plan.fp[(size_t)i];
        c->h_aabbs[(size_t)(first + i)] = aabbs[i];
    }
    c->total_pairs = plan.pairs;
    c->total_cols = plan.cols;
    c->total_items = plan.items;
    c->hist_stale = true;
}

bool extent_ok(const par_aabb& a) {
    // The sprite is 20 wide and 40 tall (alt:330, spr:67-71): texel row = (ey + ez) - 1 at most, column < ex.
    return a.ex >= 0 && a.ey >= 0 && a.ez >= 0 && a.ex <= PAR_SPRITE_W && (int)a.ey + (int)a.ez <= PAR_SPRITE_H;
}

// What a frame can hold at most of render work items: every column visited as a whole tile.
int64_t max_items(const par_context* c) {
    const int64_t B = c->params.bin_size;
    return (int64_t)c->gx * c->gy * ((B * B + 63) / 64);
}

bool items_fit(const par_context* c, int64_t items, int64_t cols) {
    const int64_t B = c->params.bin_size;
    cols = std::min<int64_t>(cols, (int64_t)c->gx * c->gy);
    return std::min(items, (cols / PAR_ITEM_SHARDS + 1) * ((B * B + 63) / 64)) <= c->grid.item_capacity;
}

void free_pool(par_context* c) {
    for (int s = 0; s < 2; s++) {
        if (c->grid.node_entity[s]) (void)hipFree(c->grid.node_entity[s]);
        if (c->grid.node_next[s]) (void)hipFree(c->grid.node_next[s]);
        if (c->grid.node_bin[s]) (void)hipFree(c->grid.node_bin[s]);
        c->grid.node_entity[s] = c->grid.node_next[s] = c->grid.node_bin[s] = nullptr;
    }
    if (c->grid.colrec) (void)hipFree(c->grid.colrec);
    c->grid.colrec = nullptr;
    c->grid.col_capacity = 0;
    c->grid.capacity = 0;
}

// One shard of the render work-item list holds the items of the columns whose index is congruent to it: at most
// every item of the frame, and at most its share of the occupied columns (<= `cols`), each visited as a whole tile.
int ensure_items(par_context* ctx, int64_t items, int64_t cols) {
    const int64_t B = ctx->params.bin_size;
    cols = std::min<int64_t>(cols, (int64_t)ctx->gx * ctx->gy);
    const int64_t need = std::min(items, (cols / PAR_ITEM_SHARDS + 1) * ((B * B + 63) / 64));
    if (need <= ctx->grid.item_capacity) return PAR_OK;
    if (ctx->graph_exec[0]) return fail(ctx, PAR_ERR_UNSUPPORTED, "work-item list would grow under a captured graph; capture again");
    const int64_t cap = std::max<int64_t>(need + need / 2, 1 << 10);
    if (cap > 0x3FFFFFFF / (PAR_ITEM_LISTS * PAR_ITEM_SHARDS)) return fail(ctx, PAR_ERR_UNSUPPORTED, "too many render work items");
    PAR_HIP(hipDeviceSynchronize());
    if (ctx->grid.items) PAR_HIP(hipFree(ctx->grid.items));
    ctx->grid.items = nullptr;
    ctx->grid.item_capacity = 0;
    PAR_HIP(hipMalloc(&ctx->grid.items, (size_t)cap * PAR_ITEM_LISTS * PAR_ITEM_SHARDS * sizeof(par_item)));
    ctx->grid.item_capacity = (int32_t)cap;
    return PAR_OK;
}

// Wipe both head/count sets and the node counters (context creation, and whenever the node pool is replaced and
// the record of which bins the previous frame touched is lost with it).
int reset_grid(par_context* ctx) {
    for (int s = 0; s < 2; s++) {
        PAR_HIP(hipMemsetAsync(ctx->grid.head[s], 0, (size_t)ctx->volume * sizeof(int32_t), ctx->stream));
        PAR_HIP(hipMemsetAsync(ctx->grid.count[s], 0, (size_t)ctx->volume, ctx->stream));
        PAR_HIP(hipMemsetAsync(ctx->grid.colflag[s], 0, (size_t)ctx->gx * ctx->gy * sizeof(int32_t), ctx->stream));
    }
    PAR_HIP(hipMemsetAsync(ctx->grid.counters, 0, PAR_CNT_TOTAL * sizeof(int32_t), ctx->stream));
    PAR_HIP(hipMemsetAsync(ctx->grid.node_counter, 0, 2 * sizeof(int32_t), ctx->stream));
    PAR_HIP(hipStreamSynchronize(ctx->stream));
    ctx->set = 0;
    return PAR_OK;
}

int ensure_pool(par_context* ctx, int64_t pairs) {
    if (pairs <= ctx->grid.capacity) return PAR_OK;
    if (ctx->graph_exec[0]) return fail(ctx, PAR_ERR_UNSUPPORTED, "node pool would grow under a captured graph; capture again");
    int64_t cap = std::max<int64_t>(pairs + pairs / 2, 1 << 16);
    if (cap > 0x3FFFFFFF) return fail(ctx, PAR_ERR_UNSUPPORTED, "too many (entity, bin) pairs");
    PAR_HIP(hipDeviceSynchronize());
    free_pool(ctx);
    for (int s = 0; s < 2; s++) {
        PAR_HIP(hipMalloc(&ctx->grid.node_entity[s], (size_t)cap * sizeof(int32_t)));
        PAR_HIP(hipMalloc(&ctx->grid.node_next[s], (size_t)cap * sizeof(int32_t)));
        PAR_HIP(hipMalloc(&ctx->grid.node_bin[s], (size_t)cap * sizeof(int32_t)));
    }
    // occupied columns <= (entity, bin) pairs: one column record each
    const int64_t col_cap = std::min<int64_t>((int64_t)ctx->gx * ctx->gy, cap);
    PAR_HIP(hipMalloc(&ctx->grid.colrec, (size_t)col_cap * sizeof(par_colrec)));
    ctx->grid.col_capacity = (int32_t)col_cap;
    ctx->grid.capacity = (int32_t)cap;
    return reset_grid(ctx);
}

void drop_graphs(par_context* c) {
    for (int s = 0; s < 2; s++) {
        if (c->graph_exec[s]) (void)hipGraphExecDestroy(c->graph_exec[s]);
        if (c->graph[s]) (void)hipGraphDestroy(c->graph[s]);
        c->graph_exec[s] = nullptr;
        c->graph[s] = nullptr;
    }
}

// Entities [first, first + n) changed on the host: both graphs' staging areas are stale there.
void mark_staged(par_context* c, int first, int n) {
    if (n <= 0) return;
    for (int s = 0; s < 2; s++) {
        if (c->stage_hi[s] <= c->stage_lo[s]) {
            c->stage_lo[s] = first;
            c->stage_hi[s] = first + n;
        } else {
            c->stage_lo[s] = std::min(c->stage_lo[s], first);
            c->stage_hi[s] = std::max(c->stage_hi[s], first + n);
        }
    }
}

par_frame_dyn make_dyn(const par_context* c, const par_light& l) {
    const int B = c->params.bin_size, H = c->params.height;
    par_frame_dyn d;
    d.lx = l.x; d.ly = l.y; d.lz = l.z;
    d.lbx = l.x / B;                // alt:729
    d.lby = (H - l.y - l.z) / B;    // alt:730-731
    d.lbz = l.z / B;                // alt:732
    return d;
}

int check_rows(par_context* ctx, int row_begin, int row_end) {
    if (row_begin < 0 || row_end > ctx->params.height || row_begin >= row_end) {
        return fail(ctx, PAR_ERR_INVALID_ARG, "rows must satisfy 0 <= row_begin < row_end <= height");
    }
    return PAR_OK;
}

int check_ready(par_context* ctx) {
    if (ctx->n_sprites <= 0) return fail(ctx, PAR_ERR_NOT_READY, "par_set_sprites has not been called");
    if (!ctx->have_entities) return fail(ctx, PAR_ERR_NOT_READY, "par_set_entities has not been called");
    if (!ctx->have_light) return fail(ctx, PAR_ERR_NOT_READY, "par_set_light has not been called");
    if (ctx->max_sprite_id >= ctx->n_sprites) return fail(ctx, PAR_ERR_SPRITE_ID, "an entity names a sprite that was not uploaded");
    return PAR_OK;
}

par_render_args make_render_args(const par_context* c, int set, int row_begin, int row_end, const par_outputs& out,
                                 unsigned flags, bool dyn_from_device) {
    par_render_args a{};
    const int B = c->params.bin_size;
    a.W = c->params.width; a.H = c->params.height; a.B = B;
    a.row_begin = row_begin; a.row_end = row_end;
    a.by_lo = row_begin / B;
    a.by_hi = (row_end - 1) / B;
    a.set = set;
    // every ray traced (as the reference does), or the lit plane requested
    a.trace_bg = ((flags & PAR_RENDER_TRACE_BACKGROUND) || out.lit) ? 1 : 0;
    // PAR_FORCE_GENERIC=1 (testing): every column goes through render_overflow_kernel
    static const bool force_generic = [] { const char* e = std::getenv("PAR_FORCE_GENERIC"); return e && e[0] == '1'; }();
    a.dense = force_generic ? 1 : 0;
    a.magic_b = (uint32_t)((1ull << 32) / (uint64_t)B + 1ull);
    a.ambient = c->params.ambient;
    a.background = c->params.background;
    a.flags = flags;
    // Columns are visited as whole tiles only in DENSE frames, which get a launch for the tile items: enough columns
    // (a 64th of the grid, at least 16) whose entities' rectangles add up to the tile (the column kernel's own
    // criterion, over the visible entries). A frame with fewer visits every column entry by entry (tile_k 0) and keeps
    // its three launches. A captured graph serves later frames too: it always has the launch. Then: how many chunks
    // per tile item -- a frame with many lets a wavefront read what a column's chunks share once for several of them,
    // a frame with few needs every wavefront it can get. PAR_TUNE_TILE_K overrides (tools; 0 is "never").
    static const int tuned_k = [] {
        const char* e = std::getenv("PAR_TUNE_TILE_K");
        const int v = e ? std::atoi(e) : -1;
        return v > 64 ? 64 : v;
    }();
    const int64_t grid_cols = (int64_t)c->gx * c->gy;
    const bool dense_frame = dyn_from_device || c->cols_tileable >= std::max<int64_t>(16, grid_cols / 64);
    // (what the entities' rectangles add up to, but no more than every column of the grid as a whole tile: the entities
    // of a crowded small view overlap many times over)
    const int64_t chunks = std::min(c->total_items, max_items(c));
    a.tile_k = tuned_k >= 0 ? tuned_k : (!dense_frame ? 0 : (chunks >= 65536 ? 5 : (chunks >= 16384 ? 3 : (chunks >= 8192 ? 2 : 1))));
    a.tile_k_magic = a.tile_k > 0 ? (uint32_t)(65536 / a.tile_k + 1) : 65537u;
    a.dyn = make_dyn(c, c->light);
    a.dyn_ptr = dyn_from_device ? c->d_dyn : nullptr;
    a.count = c->grid.count[set];
    a.slots = c->grid.slots;
    a.sprites = c->d_sprites;
    a.texinfo = c->d_texinfo;
    a.sprite_ids = c->d_sprite_ids;
    a.palette = c->d_palette;
    a.out = out;
    a.ray_counter = c->d_ray_counter;
    return a;
}

par_bin_args make_bin_args(const par_context* c, int set, int row_begin, int row_end, unsigned flags) {
    par_bin_args b{};
    b.flags = flags;
    b.by_lo = row_begin / c->params.bin_size;
    b.by_hi = (row_end - 1) / c->params.bin_size;
    b.W = c->params.width; b.H = c->params.height; b.L = c->params.length; b.B = c->params.bin_size;
    b.n = c->n_entities;
    b.set = set;
    b.aabbs = c->d_aabbs;
    b.magic_b = (uint32_t)((1ull << 32) / (uint64_t)c->params.bin_size + 1ull);
    // tests: a build workgroup that never arrives at the one-launch hash build's barrier (PAR_ERR_DEVICE)
    static const bool lose = [] { const char* e = std::getenv("PAR_TEST_LOSE_BUILD_WG"); return e && e[0] == '1'; }();
    b.test_lose_wg = lose ? 1 : 0;
    return b;
}

// The kernels' sticky failure word (PAR_CNT_ERROR), read after a wait for the device: reported once (PAR_ERR_DEVICE)
// and cleared. The caller has synchronised with the frames it is asking about.
int check_device_error(par_context* ctx) {
    int32_t word = 0;
    PAR_HIP(hipMemcpy(&word, ctx->grid.counters + PAR_CNT_ERROR, sizeof(word), hipMemcpyDeviceToHost));
    if (word == 0) return PAR_OK;
    PAR_HIP(hipMemset(ctx->grid.counters + PAR_CNT_ERROR, 0, sizeof(word)));
    std::string what;
    if (word & PAR_DEVERR_BARRIER) {
        what += "the hash build's barrier timed out (a build workgroup never arrived); ";
    }
    if (word & PAR_DEVERR_OVERFLOW) {
        what += "a column overflowed its record in a frame enqueued without a launch for the overflow list; ";
    }
    return fail(ctx, PAR_ERR_DEVICE, what + "a frame rendered since the last check is not valid");
}

// Enqueue one frame (alt:690-760) on `stream` using grid set `set`.
int enqueue_frame(par_context* ctx, hipStream_t stream, int set, int row_begin, int row_end, const par_outputs& out,
                  unsigned flags, bool graph_mode, hipEvent_t* ev) {
    par_outputs outs = out;
    if ((flags & PAR_RENDER_TRACE_BACKGROUND) && !outs.lit) {
        // every ray is to be traced but the caller wants no lit plane: the results still go to memory (a scratch
        // plane of the context), so the work is real and can be inspected
        const size_t need = (size_t)(row_end - row_begin) * ctx->params.width;
        if (ctx->scratch_lit_bytes < need) {
            if (graph_mode) return fail(ctx, PAR_ERR_NOT_READY, "render once with PAR_RENDER_TRACE_BACKGROUND before capturing it");
            if (ctx->d_scratch_lit) PAR_HIP(hipFree(ctx->d_scratch_lit));
            ctx->d_scratch_lit = nullptr;
            ctx->scratch_lit_bytes = 0;
            PAR_HIP(hipMalloc(&ctx->d_scratch_lit, need));
            ctx->scratch_lit_bytes = need;
        }
        outs.lit = ctx->d_scratch_lit;
    }
    // an asynchronous scene update on another stream: this frame comes after it
    if (ctx->ev_update_pending && ctx->update_stream != stream && !graph_mode) {
        PAR_HIP(hipStreamWaitEvent(stream, ctx->ev_update, 0));
    }
    const par_bin_args b = make_bin_args(ctx, set, row_begin, row_end, flags);
    par_render_args r = make_render_args(ctx, set, row_begin, row_end, outs, flags, graph_mode);
    // The overflow list is empty for sure while no column has more pairs than a record holds (a captured graph also
    // serves later frames, whose columns nobody knows yet): then the frame has no launch for it, and the column
    // kernel flags the frame should a column overflow all the same.
    const bool may_overflow = graph_mode || ctx->cols_over > 0 || ctx->exact_stale || ctx->hist_stale || r.dense ||
                              (ev && !(flags & PAR_RENDER_TIMED_AS_LAUNCHED));
    r.overflow_launched = may_overflow ? 1 : 0;
    if ((flags & PAR_RENDER_COUNT_RAYS) && !graph_mode) {
        PAR_HIP(hipMemsetAsync(ctx->d_ray_counter, 0, sizeof(unsigned long long), stream));
    }
    if (ev) PAR_HIP(hipEventRecord(ev[0], stream));
    // The background fill depends on nothing earlier in the frame and the render kernels come after all of it: when
    // it is the plain streaming one it rides along with the first three launches (timed runs keep all kernels apart
    // so that the event pairs bracket single ones).
    par_fill_plan plan;
    const bool no_fill = (flags & (1u << 28)) != 0;  // ablation (timing experiments only): no background fill
    // (a timed frame keeps its kernels apart unless it is asked to time the launches as a production frame makes them)
    const bool apart = ev && !(flags & PAR_RENDER_TIMED_AS_LAUNCHED);
    const bool ride = !apart && !no_fill && par_plan_fill(r, &plan);
    par_render_args rf = r;  // what rides along: the frame and palette-index planes
    rf.out.lit = nullptr;
    // A captured graph must also hold for later frames, whose pair count is unknown at capture time: the bound is
    // what par_graph_stage accepts (graph_pair_bound); beyond it the caller captures again.
    const bool stale = ctx->exact_stale;  // (asynchronous updates since the last blocking call: extents-only bounds)
    const int64_t pair_bound = graph_mode ? ctx->graph_pair_bound : (stale ? ctx->bound_pairs : ctx->total_pairs);
    // small scenes build the hash in one launch, large ones in two (timed runs keep the kernels apart)
    static const bool two_env = [] { const char* e = std::getenv("PAR_BUILD_TWO_LAUNCHES"); return e && e[0] == '1'; }();
    const bool two_launches = two_env || (flags & (1u << 23));  // bit 23 (tests): insert and resolve as two launches
    hipError_t be = (apart || two_launches) ? hipErrorNotSupported
                                         : par_launch_build(ctx->grid, b, pair_bound, &rf, ride ? &plan : nullptr, stream);
    if (be == hipErrorNotSupported) {
        PAR_HIP(par_launch_bin_insert(ctx->grid, b, &rf, ride ? &plan : nullptr, stream));
        PAR_HIP(par_launch_bin_resolve(ctx->grid, b, pair_bound, &rf, ride ? &plan : nullptr, stream));
    } else if (be != hipSuccess) {
        return hip_fail(ctx, be, "par_launch_build");
    }
    if (ev) PAR_HIP(hipEventRecord(ev[5], stream));  // (behind the hash build)
    // occupied columns <= the columns the entities reach one by one (<= their (entity, bin) pairs)
    const int64_t col_bound = graph_mode ? pair_bound : (stale ? ctx->bound_cols : ctx->total_cols);
    if (ride) {
        PAR_HIP(par_launch_columns_fill(ctx->grid, rf, col_bound, plan, stream));
    } else {
        PAR_HIP(par_launch_columns(ctx->grid, r, col_bound, stream));
    }
    if (ev) PAR_HIP(hipEventRecord(ev[1], stream));
    // Otherwise the fill follows on the same stream. (Forking it onto a second stream beside the build was measured
    // slower, alone and with several frames in flight: the cross-stream events cost more than the overlap gains.)
    // It follows the column kernels because, when background rays are traced, it copies their results into the lit
    // plane.
    if (no_fill) {
    } else if (!ride) {
        PAR_HIP(par_launch_fill(ctx->grid, r, stream));
    } else if (r.out.lit) {  // the lit plane of the background: after the background rays
        par_render_args rl = r;
        rl.out.fb = nullptr;
        rl.out.palidx = nullptr;
        PAR_HIP(par_launch_fill(ctx->grid, rl, stream));
    }
    if (ev) PAR_HIP(hipEventRecord(ev[3], stream));
    // work items <= what the entities can cause one by one, and <= every column of the rendered rows as a whole tile
    const int64_t item_cap_rows = max_items(ctx) / ctx->gy * (r.by_hi - r.by_lo + 1);
    const int64_t item_bound = std::min(graph_mode ? ctx->graph_item_bound : (stale ? ctx->bound_items : ctx->total_items),
                                        item_cap_rows);
    bool both = false;
    if (!apart) {  // small frames: one launch for both render kernels
        const hipError_t e = par_launch_render_both(ctx->grid, r, col_bound, item_bound, may_overflow, stream);
        if (e == hipSuccess) {
            both = true;
        } else if (e != hipErrorNotSupported) {
            return hip_fail(ctx, e, "par_launch_render_both");
        }
    }
    if (!both) {
        PAR_HIP(par_launch_render(ctx->grid, r, item_bound, stream));
        if (ev) PAR_HIP(hipEventRecord(ev[6], stream));
        PAR_HIP(par_launch_render_tiles(ctx->grid, r, item_bound, stream));  // (dense frames only: r.tile_k > 0)
    } else if (ev) {
        PAR_HIP(hipEventRecord(ev[6], stream));
    }
    if (ev) PAR_HIP(hipEventRecord(ev[4], stream));
    if (!both && may_overflow) PAR_HIP(par_launch_render_overflow(ctx->grid, r, col_bound, stream));
    if (ev) PAR_HIP(hipEventRecord(ev[2], stream));
    if (ev) {  // which of the optional launches this frame had (par_frame_stats::ms_launch)
        ctx->timed_tiles = !both && r.tile_k > 0;
        ctx->timed_overflow = !both && may_overflow;
        ctx->timed_both = both;
    }
    return PAR_OK;
}

int render_to_host(par_context* ctx, int row_begin, int row_end, const par_outputs* host_out, unsigned flags) {
    if (!ctx || !host_out) return fail(ctx, PAR_ERR_INVALID_ARG, "null argument");
    int rc = check_rows(ctx, row_begin, row_end);
    if (rc != PAR_OK) return rc;
    rc = check_ready(ctx);
    if (rc != PAR_OK) return rc;
    PAR_HIP(hipSetDevice(ctx->device));
    void* host[5] = {host_out->fb, host_out->gbuf, host_out->palidx, host_out->brightness, host_out->lit};
    const size_t n = (size_t)(row_end - row_begin) * ctx->params.width;
    void* dev[5];
    for (int i = 0; i < 5; i++) {
        dev[i] = nullptr;
        if (!host[i]) continue;
        const size_t bytes = n * kPlaneElem[i];
        if (ctx->d_out_bytes[i] < bytes) {
            if (ctx->d_out[i]) PAR_HIP(hipFree(ctx->d_out[i]));
            ctx->d_out[i] = nullptr;
            ctx->d_out_bytes[i] = 0;
            PAR_HIP(hipMalloc(&ctx->d_out[i], bytes));
            ctx->d_out_bytes[i] = bytes;
        }
        dev[i] = ctx->d_out[i];
    }
    par_outputs d{(par_color*)dev[0], (par_pixel*)dev[1], (uint8_t*)dev[2], (float*)dev[3], (uint8_t*)dev[4]};
    rc = enqueue_frame(ctx, ctx->stream, ctx->set, row_begin, row_end, d, flags, false, nullptr);
    if (rc != PAR_OK) return rc;
    ctx->set ^= 1;
    ctx->last_flags = flags;
    for (int i = 0; i < 5; i++) {
        if (host[i]) PAR_HIP(hipMemcpyAsync(host[i], dev[i], n * kPlaneElem[i], hipMemcpyDeviceToHost, ctx->stream));
    }
    PAR_HIP(hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

}  // namespace

extern "C" {

const char* par_last_error(const par_context* ctx) { return ctx ? ctx->err.c_str() : ""; }

int par_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int par_create_impl(const par_params* params, int device, par_context** out) {
    if (!params || !out) return PAR_ERR_INVALID_ARG;
    *out = nullptr;
    const par_params& p = *params;
    if (p.width <= 0 || p.height <= 0 || p.length <= 0 || p.bin_size <= 0 || p.width > 32767 || p.height > 32767 ||
        p.length > 32767 || !(p.ambient >= 0.f && p.ambient <= 1.f) || p.palette_size <= 0 ||
        p.palette_size > PAR_MAX_PALETTE) {
        return PAR_ERR_INVALID_ARG;  // coordinates are `short` in the reference (alt:12-38); Color*ambient must fit u8
    }
    int gx, gy, gz;
    par_grid_dims(&p, &gx, &gy, &gz);
    if (p.bin_size < PAR_MIN_BIN || p.bin_size > PAR_MAX_BIN || gx > PAR_MAX_GRID_DIM || gy > PAR_MAX_GRID_DIM ||
        gz > PAR_MAX_GRID_DIM || (int64_t)gx * gy * gz > 0x3FFFFFFF) {
        return PAR_ERR_UNSUPPORTED;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PAR_ERR_NO_DEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return PAR_ERR_NO_DEVICE;
    }
    if (device >= ndev) return PAR_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PAR_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PAR_ERR_NO_DEVICE;  // kernels are built for gfx950 only

    par_context* ctx = new (std::nothrow) par_context;
    if (!ctx) return PAR_ERR_OOM;
    ctx->params = p;
    ctx->device = device;
    ctx->gx = gx; ctx->gy = gy; ctx->gz = gz; ctx->volume = gx * gy * gz;
    ctx->grid.gx = gx; ctx->grid.gy = gy; ctx->grid.gz = gz; ctx->grid.volume = ctx->volume;
    ctx->stats.shadow_rays = -1; ctx->stats.ms_bin = -1.f; ctx->stats.ms_fill = -1.f; ctx->stats.ms_render = -1.f;
    ctx->stats.ms_overflow = -1.f;
    for (float& v : ctx->stats.ms_launch) v = -1.f;
    auto bail = [&](hipError_t e) {
        int rc = e == hipErrorOutOfMemory ? PAR_ERR_OOM : PAR_ERR_HIP;
        par_destroy(ctx);
        return rc;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
    for (int s = 0; s < 2; s++) {
        if ((e = hipMalloc(&ctx->grid.head[s], (size_t)ctx->volume * sizeof(int32_t))) != hipSuccess) return bail(e);
        if ((e = hipMalloc(&ctx->grid.count[s], (size_t)ctx->volume)) != hipSuccess) return bail(e);
        if ((e = hipMalloc(&ctx->grid.colflag[s], (size_t)gx * gy * sizeof(int32_t))) != hipSuccess) return bail(e);
    }
    if ((e = hipMalloc(&ctx->grid.col_list, (size_t)gx * gy * sizeof(int32_t))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->grid.counters, PAR_CNT_TOTAL * sizeof(int32_t))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->grid.build_sync, 64 * sizeof(int32_t))) != hipSuccess) return bail(e);
    if ((e = hipMemset(ctx->grid.build_sync, 0, 64 * sizeof(int32_t))) != hipSuccess) return bail(e);
    {
        const size_t bytes = (size_t)PAR_ITEM_LISTS * PAR_ITEM_SHARDS * PAR_ITEM_COUNTER_STRIDE * sizeof(int32_t);
        if ((e = hipMalloc(&ctx->grid.item_counters, bytes)) != hipSuccess) return bail(e);
        if ((e = hipMemset(ctx->grid.item_counters, 0, bytes)) != hipSuccess) return bail(e);
    }
    if (const char* dbg = std::getenv("PAR_DEBUG_STAMPS"); dbg && dbg[0] == '1') {
        const size_t bytes = (size_t)PAR_STAMP_ROWS * PAR_STAMP_WGS * PAR_STAMP_SLOTS * sizeof(unsigned long long);
        if ((e = hipMalloc(&ctx->grid.stamps, bytes)) != hipSuccess) return bail(e);
        if ((e = hipMemset(ctx->grid.stamps, 0, bytes)) != hipSuccess) return bail(e);
    }
    if ((e = hipMalloc(&ctx->grid.slow_list, (size_t)gx * gy * sizeof(int32_t))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->grid.bgwalk, (size_t)gx * sizeof(par_bgwalk))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->grid.bglit, (size_t)p.width + 64)) != hipSuccess) return bail(e);
    if ((e = hipMemset(ctx->grid.bglit, 1, (size_t)p.width + 64)) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->grid.slots, (size_t)ctx->volume * PAR_SLOTS * sizeof(par_slot))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->grid.node_counter, 2 * sizeof(int32_t))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->d_palette, PAR_MAX_PALETTE * sizeof(par_color))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->d_ray_counter, sizeof(unsigned long long))) != hipSuccess) return bail(e);
    if ((e = hipMalloc(&ctx->d_dyn, sizeof(par_frame_dyn))) != hipSuccess) return bail(e);
    if ((e = hipMemcpy(ctx->d_palette, p.palette, PAR_MAX_PALETTE * sizeof(par_color), hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
    if ((e = hipMemset(ctx->grid.slots, 0, (size_t)ctx->volume * PAR_SLOTS * sizeof(par_slot))) != hipSuccess) return bail(e);
    for (int i = 0; i < 7; i++) {
        if ((e = hipEventCreate(&ctx->ev[i])) != hipSuccess) return bail(e);
    }
    if (reset_grid(ctx) != PAR_OK) {
        par_destroy(ctx);
        return PAR_ERR_HIP;
    }
    if (ensure_pool(ctx, 1) != PAR_OK || ensure_items(ctx, 1, 1) != PAR_OK) {
        par_destroy(ctx);
        return PAR_ERR_OOM;
    }
    *out = ctx;
    return PAR_OK;
}

void par_destroy(par_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    drop_graphs(ctx);
    free_pool(ctx);
    for (int s = 0; s < 2; s++) {
        if (ctx->grid.head[s]) (void)hipFree(ctx->grid.head[s]);
        if (ctx->grid.count[s]) (void)hipFree(ctx->grid.count[s]);
        if (ctx->grid.colflag[s]) (void)hipFree(ctx->grid.colflag[s]);
    }
    void* lists[] = {ctx->grid.col_list, ctx->grid.counters, ctx->grid.slow_list, ctx->grid.stamps, ctx->grid.bgwalk,
                     ctx->grid.bglit, ctx->d_scratch_lit, ctx->grid.items, ctx->grid.item_counters, ctx->grid.build_sync};
    for (void* p : lists) {
        if (p) (void)hipFree(p);
    }
    void* ptrs[] = {ctx->grid.slots, ctx->grid.node_counter, ctx->d_palette, ctx->d_ray_counter, ctx->d_dyn,
                    ctx->d_aabbs, ctx->d_sprite_ids, ctx->d_sprites, ctx->d_texinfo};
    for (void* p : ptrs) {
        if (p) (void)hipFree(p);
    }
    for (int i = 0; i < 5; i++) {
        if (ctx->d_out[i]) (void)hipFree(ctx->d_out[i]);
    }
    if (ctx->pin_update) (void)hipHostFree(ctx->pin_update);
    if (ctx->ev_update) (void)hipEventDestroy(ctx->ev_update);
    for (int s = 0; s < 2; s++) {
        if (ctx->pin_aabbs[s]) (void)hipHostFree(ctx->pin_aabbs[s]);
        if (ctx->pin_dyn[s]) (void)hipHostFree(ctx->pin_dyn[s]);
        if (ctx->ev_graph[s]) (void)hipEventDestroy(ctx->ev_graph[s]);
    }
    for (int i = 0; i < 7; i++) {
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int par_set_sprites_impl(par_context* ctx, const par_sprite* sprites, int n_sprites) {
    if (!ctx || !sprites || n_sprites <= 0) return fail(ctx, PAR_ERR_INVALID_ARG, "sprites");
    test_alloc_hook();
    for (int s = 0; s < n_sprites; s++) {
        for (int t = 0; t < PAR_SPRITE_TEXELS; t++) {
            const int c = sprites[s].color[t];
            if (c < 0 || c >= ctx->params.palette_size) {  // color_palette[...] out of bounds is UB at alt:353
                return fail(ctx, PAR_ERR_SPRITE_ID, "sprite palette index outside the palette");
            }
        }
    }
    PAR_HIP(hipSetDevice(ctx->device));
    PAR_HIP(hipDeviceSynchronize());
    drop_graphs(ctx);  // (a captured graph bakes the table's pointers)
    if (ctx->d_sprites) PAR_HIP(hipFree(ctx->d_sprites));
    ctx->d_sprites = nullptr;
    ctx->n_sprites = 0;
    PAR_HIP(hipMalloc(&ctx->d_sprites, (size_t)n_sprites * sizeof(par_sprite)));
    PAR_HIP(hipMemcpy(ctx->d_sprites, sprites, (size_t)n_sprites * sizeof(par_sprite), hipMemcpyHostToDevice));
    // per-texel shading record: normal + the palette colour its index resolves to (alt:349-354)
    std::vector<par_texel> tex((size_t)n_sprites * PAR_SPRITE_TEXELS);
    for (int s = 0; s < n_sprites; s++) {
        for (int t = 0; t < PAR_SPRITE_TEXELS; t++) {
            const par_color pc = ctx->params.palette[sprites[s].color[t]];
            par_texel& x = tex[(size_t)s * PAR_SPRITE_TEXELS + t];
            x.nx = sprites[s].normal[t].x; x.ny = sprites[s].normal[t].y; x.nz = sprites[s].normal[t].z;
            x.rgba = (uint32_t)pc.red | ((uint32_t)pc.green << 8) | ((uint32_t)pc.blue << 16) | ((uint32_t)pc.alpha << 24);
        }
    }
    if (ctx->d_texinfo) PAR_HIP(hipFree(ctx->d_texinfo));
    ctx->d_texinfo = nullptr;
    PAR_HIP(hipMalloc(&ctx->d_texinfo, tex.size() * sizeof(par_texel)));
    PAR_HIP(hipMemcpy(ctx->d_texinfo, tex.data(), tex.size() * sizeof(par_texel), hipMemcpyHostToDevice));
    ctx->n_sprites = n_sprites;
    return PAR_OK;
}

static int par_set_entities_impl(par_context* ctx, const par_aabb* aabbs, const int32_t* sprite_ids, int n) {
    if (!ctx || n < 0 || (n > 0 && !aabbs)) return fail(ctx, PAR_ERR_INVALID_ARG, "entities");
    test_alloc_hook();
    int max_id = 0;
    for (int i = 0; i < n; i++) {
        if (!extent_ok(aabbs[i])) {
            return fail(ctx, PAR_ERR_EXTENT, "entity " + std::to_string(i) + ": extent needs 0<=ex<=20, ey,ez>=0, ey+ez<=40");
        }
        if (sprite_ids) {
            if (sprite_ids[i] < 0) return fail(ctx, PAR_ERR_SPRITE_ID, "negative sprite id");
            max_id = std::max(max_id, (int)sprite_ids[i]);
        }
    }
    PAR_HIP(hipSetDevice(ctx->device));
    PAR_HIP(hipDeviceSynchronize());
    drop_graphs(ctx);
    std::vector<par_footprint> fps((size_t)n);
    int64_t total = 0, total_cols = 0, total_items = 0;
    for (int i = 0; i < n; i++) {
        const par_footprint f = footprint_of(ctx, aabbs[i]);
        if (f.pairs() > 0x7FFFFFFF) return fail(ctx, PAR_ERR_UNSUPPORTED, "entity spans too many bins");
        fps[(size_t)i] = f;
        total += f.pairs();
        total_cols += f.cols();
        total_items += f.items;
    }
    // (pools and lists by what the extents allow: they then hold wherever the entities move)
    par_bound bt{0, 0, 0};
    for (int i = 0; i < n; i++) {
        const par_bound b = bound_of(ctx, aabbs[i]);
        bt.pairs += b.pairs; bt.cols += b.cols; bt.items += b.items;
    }
    if (bt.pairs > 0x3FFFFFFF) return fail(ctx, PAR_ERR_UNSUPPORTED, "too many (entity, bin) pairs");
    int rc = ensure_pool(ctx, std::max(total, bt.pairs));
    if (rc != PAR_OK) return rc;
    rc = ensure_items(ctx, std::max(total_items, bt.items), std::max(total_cols, bt.cols));
    if (rc != PAR_OK) return rc;
    if (n > ctx->aabb_capacity) {
        if (ctx->d_aabbs) PAR_HIP(hipFree(ctx->d_aabbs));
        ctx->d_aabbs = nullptr;
        ctx->aabb_capacity = 0;
        PAR_HIP(hipMalloc(&ctx->d_aabbs, (size_t)std::max(n, 1) * sizeof(par_aabb)));
        ctx->aabb_capacity = std::max(n, 1);
    }
    if (ctx->d_sprite_ids) PAR_HIP(hipFree(ctx->d_sprite_ids));
    ctx->d_sprite_ids = nullptr;
    if (n > 0) PAR_HIP(hipMemcpy(ctx->d_aabbs, aabbs, (size_t)n * sizeof(par_aabb), hipMemcpyHostToDevice));
    if (sprite_ids && n > 0 && max_id > 0) {  // all-zero ids need no table
        PAR_HIP(hipMalloc(&ctx->d_sprite_ids, (size_t)n * sizeof(int32_t)));
        PAR_HIP(hipMemcpy(ctx->d_sprite_ids, sprite_ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    ctx->h_aabbs.assign(aabbs, aabbs + n);
    ctx->h_colpairs.assign((size_t)ctx->gx * ctx->gy, 0);
    ctx->cols_over = 0;
    ctx->h_colchunks.assign((size_t)ctx->gx * ctx->gy, 0);
    ctx->cols_tileable = 0;
    for (int i = 0; i < n; i++) col_hist(ctx, fps[(size_t)i], +1);
    ctx->h_fp.swap(fps);
    ctx->total_pairs = total;
    ctx->total_cols = total_cols;
    ctx->total_items = total_items;
    ctx->bound_pairs = bt.pairs; ctx->bound_cols = bt.cols; ctx->bound_items = bt.items;
    ctx->exact_stale = false;
    ctx->n_entities = n;
    ctx->max_sprite_id = max_id;
    ctx->have_entities = true;
    ctx->stats.entities = n;
    return PAR_OK;
}

static int par_set_entities_ref_layout_impl(par_context* ctx, const par_aabb* aabbs, const par_sprite* sprite_per_entity, int n) {
    if (!ctx || n < 0 || (n > 0 && (!aabbs || !sprite_per_entity))) return fail(ctx, PAR_ERR_INVALID_ARG, "entities");
    // The reference stores one 16 000-byte Sprite per entity (alt:95,107). Keep each distinct sprite once: a 64-bit
    // hash of the bytes finds the candidates, memcmp decides.
    test_alloc_hook();
    std::unordered_map<uint64_t, std::vector<int32_t>> seen;
    std::vector<par_sprite> table;
    std::vector<int32_t> ids((size_t)n);
    for (int i = 0; i < n; i++) {
        const unsigned char* bytes = reinterpret_cast<const unsigned char*>(&sprite_per_entity[i]);
        uint64_t h = 1469598103934665603ull;  // FNV-1a over 8-byte words
        for (size_t k = 0; k + 8 <= sizeof(par_sprite); k += 8) {
            uint64_t w;
            std::memcpy(&w, bytes + k, 8);
            h = (h ^ w) * 1099511628211ull;
        }
        std::vector<int32_t>& bucket = seen[h];
        int32_t id = -1;
        for (int32_t cand : bucket) {
            if (std::memcmp(&table[(size_t)cand], bytes, sizeof(par_sprite)) == 0) {
                id = cand;
                break;
            }
        }
        if (id < 0) {
            id = (int32_t)table.size();
            table.push_back(sprite_per_entity[i]);
            bucket.push_back(id);
        }
        ids[(size_t)i] = id;
    }
    if (table.empty()) {
        par_sprite s;
        par_sprite_tile_floor(&s);
        table.push_back(s);
    }
    int rc = par_set_sprites(ctx, table.data(), (int)table.size());
    if (rc != PAR_OK) return rc;
    return par_set_entities(ctx, aabbs, ids.data(), n);
}

static int par_update_aabbs_impl(par_context* ctx, const par_aabb* aabbs, int first, int n) {
    if (!ctx || !aabbs || first < 0 || n < 0 || !ctx->have_entities || first + n > ctx->n_entities) {
        return fail(ctx, PAR_ERR_INVALID_ARG, "update range outside the uploaded entities");
    }
    for (int i = 0; i < n; i++) {
        if (!extent_ok(aabbs[i])) return fail(ctx, PAR_ERR_EXTENT, "extent needs 0<=ex<=20, ey,ez>=0, ey+ez<=40");
    }
    refresh_exact(ctx);  // (asynchronous updates may have left the exact bookkeeping behind)
    par_update_plan plan;
    plan_update(ctx, aabbs, first, n, &plan);
    par_bound bt;
    bounds_after(ctx, aabbs, first, n, &bt);
    PAR_HIP(hipSetDevice(ctx->device));
    int rc = ensure_pool(ctx, std::max(plan.pairs, bt.pairs));
    if (rc != PAR_OK) return rc;
    rc = ensure_items(ctx, std::max(plan.items, bt.items), std::max(plan.cols, bt.cols));
    if (rc != PAR_OK) return rc;
    // a frame enqueued asynchronously by par_render_device may still be reading the AABBs: wait for it
    if (ctx->has_last_stream) PAR_HIP(hipStreamSynchronize(ctx->last_stream));
    if (ctx->ev_update_pending) {  // ... and an asynchronous update may still be writing them
        PAR_HIP(hipEventSynchronize(ctx->ev_update));
        ctx->ev_update_pending = false;
    }
    PAR_HIP(hipMemcpyAsync(ctx->d_aabbs + first, aabbs, (size_t)n * sizeof(par_aabb), hipMemcpyHostToDevice, ctx->stream));
    PAR_HIP(hipStreamSynchronize(ctx->stream));
    commit_update(ctx, aabbs, first, n, plan);
    mark_staged(ctx, first, n);  // (a captured graph uploads the scene from its staging area)
    return PAR_OK;
}

static int par_update_aabbs_async_impl(par_context* ctx, const par_aabb* aabbs, int first, int n, void* stream_v) {
    if (!ctx || !aabbs || first < 0 || n < 0 || !ctx->have_entities || first + n > ctx->n_entities) {
        return fail(ctx, PAR_ERR_INVALID_ARG, "update range outside the uploaded entities");
    }
    hipStream_t stream = (hipStream_t)stream_v;
    for (int i = 0; i < n; i++) {
        if (!extent_ok(aabbs[i])) return fail(ctx, PAR_ERR_EXTENT, "extent needs 0<=ex<=20, ey,ez>=0, ey+ez<=40");
    }
    // No cull and range arithmetic here (it cost a moving scene more host time per frame than its launches): the
    // frame's launches are sized by what the EXTENTS allow (bound_of) until a blocking call refreshes the exact
    // bookkeeping, and the frame gets its launch for the overflow list whatever the columns hold.
    par_bound bt;
    bounds_after(ctx, aabbs, first, n, &bt);
    PAR_HIP(hipSetDevice(ctx->device));
    // the node pool and the item list grow rarely (only when extents grow); that path frees device memory and has to
    // wait for everything in flight
    if (bt.pairs > ctx->grid.capacity || !items_fit(ctx, bt.items, bt.cols)) {
        return par_update_aabbs(ctx, aabbs, first, n);
    }
    // frames enqueued on another stream are not ordered with this copy: wait for them
    if (ctx->has_last_stream && ctx->last_stream != stream) PAR_HIP(hipStreamSynchronize(ctx->last_stream));
    if (ctx->pin_update_capacity < ctx->aabb_capacity) {
        if (ctx->ev_update_pending) PAR_HIP(hipEventSynchronize(ctx->ev_update));
        ctx->ev_update_pending = false;
        if (ctx->pin_update) PAR_HIP(hipHostFree(ctx->pin_update));
        ctx->pin_update = nullptr;
        ctx->pin_update_capacity = 0;
        PAR_HIP(hipHostMalloc(&ctx->pin_update, (size_t)ctx->aabb_capacity * sizeof(par_aabb), hipHostMallocDefault));
        ctx->pin_update_capacity = ctx->aabb_capacity;
    }
    if (!ctx->ev_update) PAR_HIP(hipEventCreateWithFlags(&ctx->ev_update, hipEventDisableTiming));
    // the staging area is free again once the previous update's copy has run (normally long ago)
    if (ctx->ev_update_pending) PAR_HIP(hipEventSynchronize(ctx->ev_update));
    std::memcpy(ctx->pin_update + first, aabbs, (size_t)n * sizeof(par_aabb));
    PAR_HIP(hipMemcpyAsync(ctx->d_aabbs + first, ctx->pin_update + first, (size_t)n * sizeof(par_aabb),
                           hipMemcpyHostToDevice, stream));
    PAR_HIP(hipEventRecord(ctx->ev_update, stream));
    ctx->ev_update_pending = true;
    ctx->update_stream = stream;
    ctx->bound_pairs = bt.pairs; ctx->bound_cols = bt.cols; ctx->bound_items = bt.items;
    std::memcpy(ctx->h_aabbs.data() + first, aabbs, (size_t)n * sizeof(par_aabb));
    ctx->exact_stale = true;
    mark_staged(ctx, first, n);  // (a captured graph uploads the scene from its staging area)
    return PAR_OK;
}

static int par_set_light_impl(par_context* ctx, const par_light* light) {
    if (!ctx || !light) return fail(ctx, PAR_ERR_INVALID_ARG, "light");
    ctx->light = *light;
    ctx->have_light = true;
    return PAR_OK;
}

static int par_render_impl(par_context* ctx, const par_outputs* host_out, unsigned flags) {
    if (!ctx) return PAR_ERR_INVALID_ARG;
    return render_to_host(ctx, 0, ctx->params.height, host_out, flags);
}

static int par_render_rows_impl(par_context* ctx, int row_begin, int row_end, const par_outputs* host_out, unsigned flags) {
    return render_to_host(ctx, row_begin, row_end, host_out, flags);
}

static int par_render_device_impl(par_context* ctx, void* stream, int row_begin, int row_end, const par_outputs* device_out,
                      unsigned flags) {
    if (!ctx || !device_out) return fail(ctx, PAR_ERR_INVALID_ARG, "null argument");
    int rc = check_rows(ctx, row_begin, row_end);
    if (rc != PAR_OK) return rc;
    rc = check_ready(ctx);
    if (rc != PAR_OK) return rc;
    PAR_HIP(hipSetDevice(ctx->device));
    rc = enqueue_frame(ctx, (hipStream_t)stream, ctx->set, row_begin, row_end, *device_out, flags, false, nullptr);
    if (rc != PAR_OK) return rc;
    ctx->set ^= 1;
    ctx->last_flags = flags;
    ctx->last_stream = (hipStream_t)stream;
    ctx->has_last_stream = true;
    return PAR_OK;
}

static int par_render_device_timed_impl(par_context* ctx, void* stream, int row_begin, int row_end,
                            const par_outputs* device_out, unsigned flags, par_frame_stats* stats) {
    if (!ctx || !device_out) return fail(ctx, PAR_ERR_INVALID_ARG, "null argument");
    int rc = check_rows(ctx, row_begin, row_end);
    if (rc != PAR_OK) return rc;
    rc = check_ready(ctx);
    if (rc != PAR_OK) return rc;
    PAR_HIP(hipSetDevice(ctx->device));
    rc = enqueue_frame(ctx, (hipStream_t)stream, ctx->set, row_begin, row_end, *device_out, flags, false, ctx->ev);
    if (rc != PAR_OK) return rc;
    ctx->set ^= 1;
    ctx->last_flags = flags;
    PAR_HIP(hipEventSynchronize(ctx->ev[2]));
    PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_bin, ctx->ev[0], ctx->ev[1]));
    PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_fill, ctx->ev[1], ctx->ev[3]));
    PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_render, ctx->ev[3], ctx->ev[4]));
    PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_overflow, ctx->ev[4], ctx->ev[2]));
    for (float& v : ctx->stats.ms_launch) v = -1.f;
    ctx->stats.render_merged = 0;
    if (flags & PAR_RENDER_TIMED_AS_LAUNCHED) {
        ctx->stats.render_merged = ctx->timed_both ? 1 : 0;
        PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_launch[0], ctx->ev[0], ctx->ev[5]));
        PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_launch[1], ctx->ev[5], ctx->ev[3]));
        PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_launch[2], ctx->ev[3], ctx->ev[6]));
        PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_launch[3], ctx->ev[6], ctx->ev[4]));
        PAR_HIP(hipEventElapsedTime(&ctx->stats.ms_launch[4], ctx->ev[4], ctx->ev[2]));
        if (!ctx->timed_tiles) ctx->stats.ms_launch[3] = 0.f;  // (an empty bracket still measures the events themselves)
        if (!ctx->timed_overflow) ctx->stats.ms_launch[4] = 0.f;
    }
    if (stats) return par_get_stats(ctx, stats);
    return check_device_error(ctx);
}

static int par_graph_capture_impl(par_context* ctx, void* stream_v, int row_begin, int row_end, const par_outputs* device_out,
                      unsigned flags) {
    if (!ctx || !device_out) return fail(ctx, PAR_ERR_INVALID_ARG, "null argument");
    hipStream_t stream = (hipStream_t)stream_v;
    if (!stream) return fail(ctx, PAR_ERR_INVALID_ARG, "graph capture needs a non-default stream");
    int rc = check_rows(ctx, row_begin, row_end);
    if (rc != PAR_OK) return rc;
    rc = check_ready(ctx);
    if (rc != PAR_OK) return rc;
    PAR_HIP(hipSetDevice(ctx->device));
    PAR_HIP(hipDeviceSynchronize());
    drop_graphs(ctx);
    refresh_exact(ctx);
    // Head-room for moving primitives: pair counts of later frames are only bounded by the pool.
    ctx->graph_pair_bound = ctx->total_pairs * 2 + 4096;
    rc = ensure_pool(ctx, ctx->graph_pair_bound);
    if (rc != PAR_OK) return rc;
    {   // wherever the entities move: each reaches at most this many screen columns (cull and ranges of alt:212-240)
        const int B = ctx->params.bin_size;
        const int64_t cols_max = (int64_t)((PAR_SPRITE_W + B - 1) / B + 1) * ((PAR_SPRITE_H + B - 1) / B + 1);
        ctx->graph_item_bound = std::min<int64_t>(
            (int64_t)ctx->n_entities * ((int64_t)PAR_SPRITE_W * PAR_SPRITE_H / 64 + cols_max), max_items(ctx));
    }
    rc = ensure_items(ctx, ctx->graph_item_bound, ctx->graph_pair_bound);
    if (rc != PAR_OK) return rc;
    for (int s = 0; s < 2; s++) {
        if (!ctx->pin_aabbs[s]) {
            PAR_HIP(hipHostMalloc(&ctx->pin_aabbs[s], (size_t)std::max(ctx->aabb_capacity, 1) * sizeof(par_aabb), hipHostMallocDefault));
        }
        if (!ctx->pin_dyn[s]) PAR_HIP(hipHostMalloc(&ctx->pin_dyn[s], sizeof(par_frame_dyn), hipHostMallocDefault));
        if (!ctx->ev_graph[s]) PAR_HIP(hipEventCreateWithFlags(&ctx->ev_graph[s], hipEventDisableTiming));
        ctx->ev_graph_pending[s] = false;
        std::memcpy(ctx->pin_aabbs[s], ctx->h_aabbs.data(), (size_t)ctx->n_entities * sizeof(par_aabb));
        *ctx->pin_dyn[s] = make_dyn(ctx, ctx->light);
        ctx->stage_lo[s] = ctx->stage_hi[s] = 0;
    }
    // The frame alternates between the two grid sets, and a captured kernel node bakes its pointers: one graph
    // per set, launched alternately; each uploads the scene from its own staging area.
    for (int s = 0; s < 2; s++) {
        PAR_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipMemcpyAsync(ctx->d_aabbs, ctx->pin_aabbs[s], (size_t)ctx->n_entities * sizeof(par_aabb),
                                      hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_dyn, ctx->pin_dyn[s], sizeof(par_frame_dyn), hipMemcpyHostToDevice, stream);
        int erc = PAR_OK;
        if (e == hipSuccess) erc = enqueue_frame(ctx, stream, s, row_begin, row_end, *device_out, flags & ~PAR_RENDER_COUNT_RAYS, true, nullptr);
        hipGraph_t g = nullptr;
        hipError_t e2 = hipStreamEndCapture(stream, &g);
        hipError_t e3 = hipSuccess;
        if (e == hipSuccess && erc == PAR_OK && e2 == hipSuccess) {
            ctx->graph[s] = g;
            e3 = hipGraphInstantiate(&ctx->graph_exec[s], g, nullptr, nullptr, 0);
            if (e3 == hipSuccess) continue;
        } else if (g) {
            (void)hipGraphDestroy(g);
        }
        drop_graphs(ctx);  // never leave one graph of the pair behind
        if (e != hipSuccess) return hip_fail(ctx, e, "graph capture memcpy");
        if (erc != PAR_OK) return erc;
        if (e2 != hipSuccess) return hip_fail(ctx, e2, "hipStreamEndCapture");
        return hip_fail(ctx, e3, "hipGraphInstantiate");
    }
    ctx->graph_set = ctx->set;
    return PAR_OK;
}

static int par_graph_stage_impl(par_context* ctx, const par_aabb* aabbs, int first, int n, const par_light* light) {
    if (!ctx || !ctx->graph_exec[0]) return fail(ctx, PAR_ERR_NOT_READY, "no captured graph");
    if (n < 0 || first < 0 || first + n > ctx->n_entities || (n > 0 && !aabbs)) return fail(ctx, PAR_ERR_INVALID_ARG, "stage range");
    for (int i = 0; i < n; i++) {
        if (!extent_ok(aabbs[i])) return fail(ctx, PAR_ERR_EXTENT, "extent needs 0<=ex<=20, ey,ez>=0, ey+ez<=40");
    }
    if (ctx->exact_stale) refresh_exact(ctx);  // (the footprints and totals; the histograms may stay behind)
    par_update_plan plan;
    plan_update(ctx, aabbs, first, n, &plan);

    if (plan.pairs > ctx->graph_pair_bound || plan.pairs > ctx->grid.capacity) {
        return fail(ctx, PAR_ERR_UNSUPPORTED, "staged frame exceeds what the captured graph was sized for; capture again");
    }
    commit_update_totals(ctx, aabbs, first, n, plan);
    mark_staged(ctx, first, n);  // (the staging areas are brought up to date by par_graph_launch)
    if (light) ctx->light = *light;
    return PAR_OK;
}

static int par_graph_launch_impl(par_context* ctx, void* stream) {
    if (!ctx || !ctx->graph_exec[0]) return fail(ctx, PAR_ERR_NOT_READY, "no captured graph");
    // (the scene may also have been changed by par_update_aabbs[_async]: same limit as par_graph_stage)
    if (ctx->exact_stale) refresh_exact(ctx);
    if (ctx->total_pairs > ctx->graph_pair_bound) {
        return fail(ctx, PAR_ERR_UNSUPPORTED, "the scene exceeds what the captured graph was sized for; capture again");
    }
    const int s = ctx->set;
    if (!ctx->graph_exec[s]) return fail(ctx, PAR_ERR_NOT_READY, "no captured graph for this grid set");
    PAR_HIP(hipSetDevice(ctx->device));
    // This graph's staging area: free once its previous launch has run (its copy nodes read the area when they
    // execute, not when the graph is launched), then brought up to date with the host mirror and the light.
    if (ctx->ev_graph_pending[s]) {
        if (hipEventQuery(ctx->ev_graph[s]) != hipSuccess) PAR_HIP(hipEventSynchronize(ctx->ev_graph[s]));
        ctx->ev_graph_pending[s] = false;
    }
    if (ctx->stage_hi[s] > ctx->stage_lo[s]) {
        std::memcpy(ctx->pin_aabbs[s] + ctx->stage_lo[s], ctx->h_aabbs.data() + ctx->stage_lo[s],
                    (size_t)(ctx->stage_hi[s] - ctx->stage_lo[s]) * sizeof(par_aabb));
        ctx->stage_lo[s] = ctx->stage_hi[s] = 0;
    }
    *ctx->pin_dyn[s] = make_dyn(ctx, ctx->light);
    // an asynchronous scene update on another stream: this frame comes after it
    if (ctx->ev_update_pending && ctx->update_stream != (hipStream_t)stream) {
        PAR_HIP(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_update, 0));
    }
    PAR_HIP(hipGraphLaunch(ctx->graph_exec[s], (hipStream_t)stream));
    PAR_HIP(hipEventRecord(ctx->ev_graph[s], (hipStream_t)stream));
    ctx->ev_graph_pending[s] = true;
    ctx->set ^= 1;
    ctx->last_stream = (hipStream_t)stream;
    ctx->has_last_stream = true;
    return PAR_OK;
}

static int par_pick_impl(par_context* ctx, int x, int y, par_pixel* out) {
    if (!ctx || !out || x < 0 || y < 0 || x >= ctx->params.width || y >= ctx->params.height) {
        return fail(ctx, PAR_ERR_INVALID_ARG, "pick outside the view");
    }
    std::vector<par_pixel> row((size_t)ctx->params.width);
    par_outputs o{nullptr, row.data(), nullptr, nullptr, nullptr};
    int rc = render_to_host(ctx, y, y + 1, &o, 0);
    if (rc != PAR_OK) return rc;
    *out = row[(size_t)x];  // `mouse_pixel`, alt:380-382
    return PAR_OK;
}

static int par_get_stats_impl(par_context* ctx, par_frame_stats* stats) {
    if (!ctx || !stats) return fail(ctx, PAR_ERR_INVALID_ARG, "stats");
    PAR_HIP(hipSetDevice(ctx->device));
    PAR_HIP(hipDeviceSynchronize());
    ctx->stats.entities = ctx->n_entities;
    ctx->stats.shadow_rays = -1;
    {
        int32_t nc[2] = {0, 0};
        static_assert(PAR_CNT_COLS == 0 && PAR_CNT_SLOW == 1, "read together");
        PAR_HIP(hipMemcpy(nc, ctx->grid.counters, sizeof(nc), hipMemcpyDeviceToHost));
        ctx->stats.occupied_columns = nc[0];
        ctx->stats.overflow_columns = nc[1];
        // the insert kernel's own count of the last frame's (entity, bin) nodes (the next frame's resolve resets it)
        int32_t nodes = 0;
        PAR_HIP(hipMemcpy(&nodes, ctx->grid.node_counter + (ctx->set ^ 1), sizeof(nodes), hipMemcpyDeviceToHost));
        ctx->stats.bin_insertions = nodes;
        const int rc = check_device_error(ctx);
        if (rc != PAR_OK) return rc;
    }
    if (ctx->last_flags & PAR_RENDER_COUNT_RAYS) {
        unsigned long long v = 0;
        PAR_HIP(hipMemcpy(&v, ctx->d_ray_counter, sizeof(v), hipMemcpyDeviceToHost));
        ctx->stats.shadow_rays = (int64_t)v;
    }
    *stats = ctx->stats;
    return PAR_OK;
}

// Internal profiling aid (not part of the public header): copy the debug stamp buffer out (PAR_DEBUG_STAMPS=1).
int par_debug_read_stamps(par_context* ctx, unsigned long long* out, size_t count) {
    if (!ctx || !out || !ctx->grid.stamps) return PAR_ERR_NOT_READY;
    const size_t n = (size_t)PAR_STAMP_ROWS * PAR_STAMP_WGS * PAR_STAMP_SLOTS;
    if (hipDeviceSynchronize() != hipSuccess) return PAR_ERR_HIP;
    if (hipMemcpy(out, ctx->grid.stamps, (count < n ? count : n) * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return PAR_ERR_HIP;
    return PAR_OK;
}

static int par_read_grid_impl(par_context* ctx, int32_t* count, int32_t* map, par_aabb* bins) {
    if (!ctx || !count || !map || !bins) return fail(ctx, PAR_ERR_INVALID_ARG, "grid buffers");
    PAR_HIP(hipSetDevice(ctx->device));
    PAR_HIP(hipDeviceSynchronize());
    const int last = ctx->set ^ 1;  // the set the last frame used
    std::vector<uint8_t> c((size_t)ctx->volume);
    std::vector<par_slot> s((size_t)ctx->volume * PAR_SLOTS);
    PAR_HIP(hipMemcpy(c.data(), ctx->grid.count[last], c.size(), hipMemcpyDeviceToHost));
    PAR_HIP(hipMemcpy(s.data(), ctx->grid.slots, s.size() * sizeof(par_slot), hipMemcpyDeviceToHost));
    std::memset(map, 0, (size_t)ctx->volume * PAR_SLOTS * sizeof(int32_t));
    std::memset(bins, 0, (size_t)ctx->volume * PAR_SLOTS * sizeof(par_aabb));
    for (int b = 0; b < ctx->volume; b++) {
        count[b] = c[(size_t)b];
        for (int k = 0; k < c[(size_t)b]; k++) {
            const par_slot& r = s[(size_t)b * PAR_SLOTS + k];
            map[(size_t)b * PAR_SLOTS + k] = r.entity;
            bins[(size_t)b * PAR_SLOTS + k] = par_aabb{r.px, r.py, r.pz, r.ex, r.ey, r.ez, {0, 0}};
        }
    }
    return PAR_OK;
}


// Test hook: the reference's three arithmetic units as the DEVICE computes them (slab_hit, color_scale,
// normalize_l1_and_inverse of par_kernels.hip) on host vectors. See par_raytracer.h.
static int par_debug_units_impl(int device, int kind, const void* in_a, const void* in_b, int n, void* out) {
    const bool slab = kind == 0 || kind == 3 || kind == 4;  // (3, 4: the slab test on a walk record, par_kernels.hip)
    if (kind < 0 || kind > 4 || n < 0 || !in_a || !out || (slab && !in_b)) return PAR_ERR_INVALID_ARG;
    par_context* ctx = nullptr;  // (PAR_HIP reports through it)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return PAR_ERR_NO_DEVICE;
    if (device < 0) device = 0;
    if (device >= ndev) return PAR_ERR_INVALID_ARG;
    PAR_HIP(hipSetDevice(device));
    const size_t a_bytes = (size_t)n * (slab ? sizeof(par_aabb) : (kind == 1 ? 5 : 3) * sizeof(float));
    const size_t b_bytes = slab ? (size_t)n * 20 : 0;
    const size_t o_bytes = (size_t)n * (slab ? 1 : (kind == 1 ? 4 : 12));
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc(&da, std::max<size_t>(a_bytes, 16));
    if (e == hipSuccess) e = hipMalloc(&db, std::max<size_t>(b_bytes, 16));
    if (e == hipSuccess) e = hipMalloc(&dout, std::max<size_t>(o_bytes, 16));
    if (e == hipSuccess && a_bytes) e = hipMemcpy(da, in_a, a_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && b_bytes) e = hipMemcpy(db, in_b, b_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = par_launch_units(kind, da, db, n, dout, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && o_bytes) e = hipMemcpy(out, dout, o_bytes, hipMemcpyDeviceToHost);
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return e == hipSuccess ? PAR_OK : (e == hipErrorOutOfMemory ? PAR_ERR_OOM : PAR_ERR_HIP);
}
int par_debug_units(int device, int kind, const void* in_a, const void* in_b, int n, void* out) {
    return guarded(nullptr, [&] { return par_debug_units_impl(device, kind, in_a, in_b, n, out); });
}

static int par_render_device_slots_impl(par_context* const* ctxs, void* const* streams, const par_outputs* device_outs,
                                        int n_slots, int row_begin, int row_end, int first_frame, int n_frames,
                                        unsigned flags) {
    if (!ctxs || !streams || !device_outs || n_slots < 1 || n_frames < 0 || first_frame < 0) return PAR_ERR_INVALID_ARG;
    if (n_slots > 1) flags |= PAR_RENDER_PIPELINED;  // several frames in flight: throughput before latency
    for (int f = first_frame; f < first_frame + n_frames; f++) {
        const int k = f % n_slots;
        const int rc = par_render_device(ctxs[k], streams[k], row_begin, row_end, &device_outs[k], flags);
        if (rc != PAR_OK) return rc;
    }
    return PAR_OK;
}
int par_render_device_slots(par_context* const* ctxs, void* const* streams, const par_outputs* device_outs, int n_slots,
                            int row_begin, int row_end, int first_frame, int n_frames, unsigned flags) {
    return guarded(nullptr, [&] {
        return par_render_device_slots_impl(ctxs, streams, device_outs, n_slots, row_begin, row_end, first_frame,
                                            n_frames, flags);
    });
}

// ---- sharded frames: tiles and background (context-free: the parameters say all that is needed) -------------
static int tiles_args_ok(const par_params* p, int n) {
    return p && n >= 0 && p->width > 0 && p->height > 0 && p->bin_size >= PAR_MIN_BIN && p->bin_size <= PAR_MAX_BIN;
}
int par_tiles_pack(const par_params* p, void* stream, const int32_t* d_tiles, int n, const par_color* fb_block,
                   int row_begin, int row_end, par_color* packed) {
    if (!tiles_args_ok(p, n) || (n > 0 && (!d_tiles || !fb_block || !packed)) || row_begin < 0 || row_end > p->height ||
        row_begin > row_end) {
        return PAR_ERR_INVALID_ARG;
    }
    const hipError_t e = par_launch_tiles_copy(true, d_tiles, n, p->width, p->height, p->bin_size, row_begin, row_end,
                                               fb_block, packed, (hipStream_t)stream);
    return e == hipSuccess ? PAR_OK : PAR_ERR_HIP;
}
int par_tiles_unpack(const par_params* p, void* stream, const int32_t* d_tiles, int n, const par_color* packed,
                     par_color* frame) {
    if (!tiles_args_ok(p, n) || (n > 0 && (!d_tiles || !frame || !packed))) return PAR_ERR_INVALID_ARG;
    const hipError_t e = par_launch_tiles_copy(false, d_tiles, n, p->width, p->height, p->bin_size, 0, p->height, packed,
                                               frame, (hipStream_t)stream);
    return e == hipSuccess ? PAR_OK : PAR_ERR_HIP;
}
int par_background_fill(const par_params* p, void* stream, par_color* rows, int n_rows) {
    if (!p || n_rows < 0 || (n_rows > 0 && !rows) || p->width <= 0 || !(p->ambient >= 0.f && p->ambient <= 1.f)) {
        return PAR_ERR_INVALID_ARG;
    }
    const uint32_t ch = (uint32_t)(uint8_t)((float)p->background * p->ambient);  // Color{127,127,127,0} * ambient
    const hipError_t e = par_launch_background(rows, (size_t)n_rows * (size_t)p->width, ch | (ch << 8) | (ch << 16),
                                               (hipStream_t)stream);
    return e == hipSuccess ? PAR_OK : PAR_ERR_HIP;
}
int par_tiles_assemble(const par_params* p, void* stream, const int32_t* d_map, const par_color* packed, par_color* frame,
                       int row_begin, int row_end) {
    int gx, gy, gz;
    if (!p || par_grid_dims(p, &gx, &gy, &gz) != PAR_OK || !d_map || !frame || !packed || row_begin < 0 ||
        row_end > p->height || row_begin > row_end || !(p->ambient >= 0.f && p->ambient <= 1.f)) {
        return PAR_ERR_INVALID_ARG;
    }
    const uint32_t ch = (uint32_t)(uint8_t)((float)p->background * p->ambient);  // Color{127,127,127,0} * ambient
    const hipError_t e = par_launch_tiles_assemble(d_map, gx, p->width, p->bin_size, row_begin, row_end, packed, frame,
                                                   ch | (ch << 8) | (ch << 16), (hipStream_t)stream);
    return e == hipSuccess ? PAR_OK : PAR_ERR_HIP;
}

// ---- the exported entry points of the bodies above: no exception leaves the library -----------------------
int par_set_sprites(par_context* ctx, const par_sprite* sprites, int n_sprites) {
    return guarded(ctx, [&] { return par_set_sprites_impl(ctx, sprites, n_sprites); });
}
int par_set_entities(par_context* ctx, const par_aabb* aabbs, const int32_t* sprite_ids, int n) {
    return guarded(ctx, [&] { return par_set_entities_impl(ctx, aabbs, sprite_ids, n); });
}
int par_set_entities_ref_layout(par_context* ctx, const par_aabb* aabbs, const par_sprite* sprite_per_entity, int n) {
    return guarded(ctx, [&] { return par_set_entities_ref_layout_impl(ctx, aabbs, sprite_per_entity, n); });
}
int par_update_aabbs(par_context* ctx, const par_aabb* aabbs, int first, int n) {
    return guarded(ctx, [&] { return par_update_aabbs_impl(ctx, aabbs, first, n); });
}
int par_update_aabbs_async(par_context* ctx, const par_aabb* aabbs, int first, int n, void* stream_v) {
    return guarded(ctx, [&] { return par_update_aabbs_async_impl(ctx, aabbs, first, n, stream_v); });
}
int par_pick(par_context* ctx, int x, int y, par_pixel* out) {
    return guarded(ctx, [&] { return par_pick_impl(ctx, x, y, out); });
}
int par_read_grid(par_context* ctx, int32_t* count, int32_t* map, par_aabb* bins) {
    return guarded(ctx, [&] { return par_read_grid_impl(ctx, count, map, bins); });
}
int par_graph_capture(par_context* ctx, void* stream_v, int row_begin, int row_end, const par_outputs* device_out, unsigned flags) {
    return guarded(ctx, [&] { return par_graph_capture_impl(ctx, stream_v, row_begin, row_end, device_out, flags); });
}
int par_graph_stage(par_context* ctx, const par_aabb* aabbs, int first, int n, const par_light* light) {
    return guarded(ctx, [&] { return par_graph_stage_impl(ctx, aabbs, first, n, light); });
}
int par_get_stats(par_context* ctx, par_frame_stats* stats) {
    return guarded(ctx, [&] { return par_get_stats_impl(ctx, stats); });
}
int par_create(const par_params* params, int device, par_context** out) {
    return guarded(nullptr, [&] { return par_create_impl(params, device, out); });
}

int par_set_light(par_context* ctx, const par_light* light) {
    return guarded(ctx, [&] { return par_set_light_impl(ctx, light); });
}
int par_render(par_context* ctx, const par_outputs* host_out, unsigned flags) {
    return guarded(ctx, [&] { return par_render_impl(ctx, host_out, flags); });
}
int par_render_rows(par_context* ctx, int row_begin, int row_end, const par_outputs* host_out, unsigned flags) {
    return guarded(ctx, [&] { return par_render_rows_impl(ctx, row_begin, row_end, host_out, flags); });
}
int par_render_device(par_context* ctx, void* stream, int row_begin, int row_end, const par_outputs* device_out, unsigned flags) {
    return guarded(ctx, [&] { return par_render_device_impl(ctx, stream, row_begin, row_end, device_out, flags); });
}
int par_render_device_timed(par_context* ctx, void* stream, int row_begin, int row_end, const par_outputs* device_out, unsigned flags, par_frame_stats* stats) {
    return guarded(ctx, [&] { return par_render_device_timed_impl(ctx, stream, row_begin, row_end, device_out, flags, stats); });
}
int par_graph_launch(par_context* ctx, void* stream) {
    return guarded(ctx, [&] { return par_graph_launch_impl(ctx, stream); });
}

}  // extern "C"
