/*
 * par_raytracer.h — C ABI of the MI355X-native pixel-art raytracer (libpar_raytracer.so).
 *
 * This is the drop-in boundary for the reference's render call: the three statements plus the inline loop in
 * `main` at src/alternative.cpp:690-760 (memset + count_entities_in_bins + trace_hash_for_pixel + the
 * shading/quantise loop around trace_hash_for_light). The reference has no FFI of its own; each entry
 * point below names the reference interface it replaces (alt = src/alternative.cpp, spr = src/sprites.hpp).
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no exceptions cross this boundary; every call returns a par_status.
 *   - ownership follows the reference (alt:503-517): the caller owns every input and output buffer; the context
 *     owns only its device-side copies and work arrays.
 *   - one context = one GPU = one host thread at a time (the reference is single-threaded).
 *   - the library REQUIRES a gfx950 device: there is no CPU fallback. par_create fails with PAR_ERR_NO_DEVICE.
 */
#ifndef PAR_RAYTRACER_H
#define PAR_RAYTRACER_H

#include "par_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct par_context par_context;

typedef enum par_status {
    PAR_OK = 0,
    PAR_ERR_INVALID_ARG = 1, /* null pointer, negative size, rows out of range, ambient outside [0,1] ... */
    PAR_ERR_NO_DEVICE = 2,   /* no HIP device / not gfx950 */
    PAR_ERR_HIP = 3,         /* a HIP runtime call failed; see par_last_error */
    PAR_ERR_OOM = 4,         /* host or device allocation failed */
    PAR_ERR_UNSUPPORTED = 5, /* grid dimension or bin size outside what the kernels are built for */
    PAR_ERR_EXTENT = 6,      /* an AABB extent the 20x40 sprite cannot express (reference UB: alt:330, SURVEY a-3b) */
    PAR_ERR_SPRITE_ID = 7,   /* sprite id or sprite palette index out of range */
    PAR_ERR_NOT_READY = 8,   /* render before sprites / entities / light were set */
    PAR_ERR_DEVICE = 9       /* a kernel reported a failure since the flag was last read (the hash build's barrier timed
                              * out, or a column overflowed its record in a frame that was sized for none to): a frame
                              * rendered since then is NOT valid. Reported once, by the first call that looks: par_render,
                              * par_render_rows, par_render_device_timed, par_get_stats */
} par_status;

/* Render flags. */
enum {
    /* Cast the shadow ray of uncovered (background) pixels too, as the reference does (alt:703,738). Their colour
     * cannot depend on the result (SURVEY a-6), so by default the ray is skipped; requesting the `lit` plane turns
     * this on implicitly because the result then is observable. */
    PAR_RENDER_TRACE_BACKGROUND = 1u << 0,
    /* Count the shadow rays actually traced into par_frame_stats (one atomic per workgroup). */
    PAR_RENDER_COUNT_RAYS = 1u << 1,
    /* This frame is one of several in flight on the device (a swap chain, par_render_device_slots): favour the
     * device's throughput over the frame's own latency (one wavefront per screen column builds its record and does
     * all its shadow walks, instead of two sharing the walks). Same pixels either way. */
    PAR_RENDER_PIPELINED = 1u << 2,
    /* par_render_device_timed only: time the frame's launches AS A PRODUCTION FRAME MAKES THEM (the background fill
     * riding with the hash-build and column launches) into par_frame_stats.ms_launch, instead of the kernels apart. */
    PAR_RENDER_TIMED_AS_LAUNCHED = 1u << 3
    /* Bit 22 (tests): no self-contained work items, every column is rendered from its record; same pixels.
     * Bit 23 (tests): build the spatial hash with two launches even where one would do; same pixels.
     * Bits 24-28 switch parts of the frame OFF for timing experiments (tools/ablate.py, tools/overlap.py): the output
     * is then wrong by design. Never set them in a render whose pixels are used. Bit 29: debug time stamps. */
};

/* Output planes of one render, each nullable. Every pointer addresses the element of (row_begin, column 0);
 * a plane holds (row_end - row_begin) * width elements, row-major like the reference's buffers. */
typedef struct par_outputs {
    par_color* fb;     /* final RGBA8 frame, `p_texture` alt:515,735,757 */
    par_pixel* gbuf;   /* G-buffer, `p_pixel_buffer` alt:511,379 */
    uint8_t* palidx;   /* sprite palette index per pixel (alt:352-354), PAR_PALIDX_BACKGROUND where uncovered */
    float* brightness; /* pre-quantise brightness factor: ambient or min(1, diffuse + ambient), alt:735,757-758 */
    uint8_t* lit;      /* 1 where trace_hash_for_light returned true (alt:738) */
} par_outputs;

typedef struct par_frame_stats {
    int64_t entities;        /* entities uploaded */
    int64_t bin_insertions;  /* (entity, bin) pairs the device inserted in the last frame (alt:243-267 iterations; the
                              * insert kernel's own node count) */
    int64_t shadow_rays;     /* shadow rays traced (only with PAR_RENDER_COUNT_RAYS), else -1 */
    int64_t occupied_columns; /* screen columns (bin footprints) in the rendered rows that show a primitive */
    int64_t overflow_columns; /* ... of which did not fit a column record (rendered straight from the hash) */
    float ms_bin;            /* device time of the hash build + column kernels of the last timed render, else -1 */
    float ms_fill;           /* device time of the background fill kernel of the last timed render, else -1 */
    float ms_render;         /* device time of the render kernels (render_items_kernel, and render_tiles_kernel in a
                              * dense frame) of the last timed render, else -1 */
    float ms_overflow;       /* device time of the overflow-column kernel of the last timed render, else -1 */
    float ms_launch[5];      /* PAR_RENDER_TIMED_AS_LAUNCHED: device time of the frame's launches as a production frame
                              * makes them: [0] hash build (+ its share of the fill), [1] column records (+ the rest of the
                              * fill), [2] render_items_kernel (entry work items), [3] render_tiles_kernel (whole-tile
                              * items of a dense frame), [4] overflow list; 0 where the frame has no such launch;
                              * else -1 */
    int32_t render_merged;   /* PAR_RENDER_TIMED_AS_LAUNCHED: 1 when the frame rendered its entry items, tile items and
                              * overflow columns in ONE launch (render_both_kernel, small frames; its time is
                              * ms_launch[2]), else 0 */
} par_frame_stats;

const char* par_status_string(int status);
/* Detail of the last failure on this context ("" when none). The pointer stays valid until the next call. */
const char* par_last_error(const par_context* ctx);

/* Reference defaults: 480x320x320, bin 40, ambient 0.25, background 127, 4-entry gray palette
 * (alt:116-131, alt:281, alt:702, spr:60-65). */
void par_default_params(par_params* params);
/* hash_width/height/length (alt:120-122) for these parameters. */
int par_grid_dims(const par_params* params, int* gx, int* gy, int* gz);

/* Number of visible HIP devices (0 when there is none); never fails. */
int par_device_count(void);

/* Create a renderer on HIP device `device` (-1: the current device). Allocates the grid and work arrays. */
int par_create(const par_params* params, int device, par_context** out);
void par_destroy(par_context* ctx);

/* --- scene surface: replaces `Entities<N>{aabbs, sprites}` (alt:92-114) and `lights` (alt:619-626) ----------- */

/* Unique sprite table. Palette indices are validated against params.palette_size. */
int par_set_sprites(par_context* ctx, const par_sprite* sprites, int n_sprites);
/* All entities; entity index == array index (alt:93-97). `sprite_ids` nullable (all entities use sprite 0). */
int par_set_entities(par_context* ctx, const par_aabb* aabbs, const int32_t* sprite_ids, int n);
/* The reference's own layout: one Sprite per entity (`std::vector<Sprite>`, alt:95,107). Identical sprites are
 * stored once on the device; entity order is kept. */
int par_set_entities_ref_layout(par_context* ctx, const par_aabb* aabbs, const par_sprite* sprite_per_entity, int n);
/* Per-frame mutation (alt:643-660 moves aabbs[0]): overwrite aabbs[first, first+n). */
int par_update_aabbs(par_context* ctx, const par_aabb* aabbs, int first, int n);
/* The same without blocking: the new AABBs are copied to the device in `stream` order, i.e. after the frames already
 * enqueued on `stream` and before the next one. `stream` must be the stream this context's frames are rendered on
 * (par_render_device); `aabbs` may be reused as soon as the call returns. For a render loop with frames in flight:
 * the call does no cull or bin-range arithmetic on the host (the frames that follow size their launches by what the
 * entities' extents allow wherever they stand, and always carry the launch for overflowed columns), until a blocking
 * call (par_update_aabbs, par_set_entities, the graph calls) brings the exact bookkeeping up to date. */
int par_update_aabbs_async(par_context* ctx, const par_aabb* aabbs, int first, int n, void* stream);
/* lights[0] (alt:712-714, 729-732: the only light the reference reads). */
int par_set_light(par_context* ctx, const par_light* light);

/* --- render: replaces alt:690-760 ----------------------------------------------------------------------------- */

/* One frame into caller-owned HOST buffers: bin, trace, shade, copy back, synchronise. */
int par_render(par_context* ctx, const par_outputs* host_out, unsigned flags);
/* Rows [row_begin,row_end) only (multi-GPU row-block sharding); host buffers, synchronous. */
int par_render_rows(par_context* ctx, int row_begin, int row_end, const par_outputs* host_out, unsigned flags);
/* Asynchronous: enqueue on `stream` (a hipStream_t, NULL = default stream) writing DEVICE buffers. No sync.
 * Several frames may be in flight at once, each on its own context and stream (a frame alone is a chain of short
 * kernels that leaves most of the chip idle). A scene update (par_update_aabbs) waits for the context's last
 * asynchronous frame first; par_update_aabbs_async does not. */
int par_render_device(par_context* ctx, void* stream, int row_begin, int row_end, const par_outputs* device_out,
                      unsigned flags);
/* The render loop of a swap chain in one call: frames first_frame .. first_frame + n_frames - 1, frame i on slot
 * i % n_slots (its context, its stream, its device outputs), enqueued back to back without a host wait. Each slot is
 * a context of its own; all of them render the same rows. Like par_render_device it never waits for the device, so it
 * cannot see a kernel's failure flag (PAR_ERR_DEVICE): the caller polls par_get_stats on each slot's context when it
 * next waits for that slot anyway (the flag is sticky until read). */
int par_render_device_slots(par_context* const* ctxs, void* const* streams, const par_outputs* device_outs,
                            int n_slots, int row_begin, int row_end, int first_frame, int n_frames, unsigned flags);
/* As par_render_device, bracketing the kernel groups with HIP events on `stream`; blocks until the frame is done
 * and fills stats->ms_bin (hash build + column kernels) / ms_fill / ms_render / ms_overflow. */
int par_render_device_timed(par_context* ctx, void* stream, int row_begin, int row_end,
                            const par_outputs* device_out, unsigned flags, par_frame_stats* stats);

/* hipGraph path (BASELINE config 5): capture {pinned-host AABB/light upload -> build -> fill -> render} once, replay
 * per frame. `par_graph_stage` writes the next frame's AABBs/light into the pinned staging area the graph copies
 * from; it fails with PAR_ERR_UNSUPPORTED when the staged scene needs larger launch grids than were captured (about
 * twice the bin insertions of the captured frame): capture again then. */
int par_graph_capture(par_context* ctx, void* stream, int row_begin, int row_end, const par_outputs* device_out,
                      unsigned flags);
int par_graph_stage(par_context* ctx, const par_aabb* aabbs, int first, int n, const par_light* light);
int par_graph_launch(par_context* ctx, void* stream);

/* Mouse pick (alt:380-382, 698-700): the G-buffer texel under (x, y) of the last frame rendered with a gbuf
 * plane is the caller's to read; this helper renders just that pixel's row. */
int par_pick(par_context* ctx, int x, int y, par_pixel* out);

/* Statistics of the last render (blocks until it finished). PAR_ERR_DEVICE when a kernel flagged a failure since the
 * flag was last read. */
int par_get_stats(par_context* ctx, par_frame_stats* stats);

/* Read back the spatial hash of the last render in the reference's layout (`count[G]`, `map[G*8]`,
 * `bins[G*8]`, alt:503-509). Only slots below count[b] are defined; the others are zero. Parity tooling. */
int par_read_grid(par_context* ctx, int32_t* count, int32_t* map, par_aabb* bins);

/* Test hook: the reference's three arithmetic units as the DEVICE kernels compute them, on `n` host vectors
 * (tests/golden/ref_units.npz holds the reference's own answers). kind 0: AABB::intersect, alt:40-83 — in_a =
 * par_aabb[n], in_b = n x {float inv_x, inv_y, inv_z; int16 origin x, y, z; int16 pad} (`Ray`, alt:30-33), out =
 * uint8_t[n]. kind 1: Color::operator*(float), spr:8-16 — in_a = n x {float r, g, b, a, v}, out = n x uint8_t[4].
 * kind 2: Vector::normalize, spr:28-35 — in_a = n x float[3], out = n x float[3]. kinds 3 and 4: kind 0's test as the
 * render kernel runs it on the records of a shadow walk (planes as floats, packed arithmetic; 3: hardware min/max
 * where the inverse direction is finite, 4: the reference's compare-selects throughout), same arguments. */
int par_debug_units(int device, int kind, const void* in_a, const void* in_b, int n, void* out);

/* --- host-side scene helpers (C++ host code, no GPU needed) ---------------------------------------------------- */

/* `make_tile_floor` (spr:73-364). */
void par_sprite_tile_floor(par_sprite* out);
/* The graybox world of alt:517-599 for a view of width x length (480 x 320 in the reference). Returns the entity
 * count (162 308 for the reference view); writes at most `capacity` AABBs. */
int par_scene_graybox(int view_width, int view_length, par_aabb* out, int capacity);
/* Synthetic benchmark scene: n boxes of extent (20,20,20), positions from splitmix64(seed):
 * x in [-20,width), y in [-20,200), z in [-20,length). Also returns the light (5w/8, h/2, l/4). */
int par_scene_synthetic(int n, int width, int height, int length, uint64_t seed, par_aabb* out, par_light* light);
/* Row block [begin, end) of `rank` when one frame is sharded over `ranks` GPUs: contiguous, disjoint,
 * covering [0, height), cut at multiples of the bin size (a bin row of 40 screen rows never straddles two ranks), the
 * bin rows dealt as evenly as they go. */
void par_row_block(int rank, int ranks, int height, int bin_size, int* begin, int* end);
/* --- sharded frames: assembling only what can differ from the background ----------------------------
 * A frame sharded by row blocks over the GPUs of a node is assembled on one of them. Most of a sparse frame is the
 * constant background, which the assembling rank can write itself: only the screen TILES (bin footprints, bin_size x
 * bin_size pixels) that can show a primitive need to travel. Every rank holds the whole scene, so every rank derives
 * the same tile list (no exchange of metadata), sorted by bin row: a rank's row block is a contiguous run of it. */

/* The tiles that can show a primitive: the screen columns the entities reach by the cull and bin ranges of
 * alt:212-240 (a superset of the columns the spatial hash will mark occupied). Host arithmetic, no GPU needed.
 * tiles[i] = bx | by << 16, sorted by (by, bx). Returns the number of such tiles (writes at most `capacity`), or a
 * negative par_status. */
int par_scene_tiles(const par_params* params, const par_aabb* aabbs, int n, int32_t* tiles, int capacity);
/* Copy tiles d_tiles[0, n) (device array) out of a frame block -- rows [row_begin, row_end) of the frame, stored
 * at `fb_block` -- into `packed`: n slots of bin_size x bin_size pixels, slot i row-major. Pixels of a slot beyond the
 * view's right / bottom edge are not written. Asynchronous on `stream` (a hipStream_t); device pointers. */
int par_tiles_pack(const par_params* params, void* stream, const int32_t* d_tiles, int n, const par_color* fb_block,
                   int row_begin, int row_end, par_color* packed);
/* The inverse on the assembling rank: slots -> their place in the whole frame (`frame` addresses row 0). */
int par_tiles_unpack(const par_params* params, void* stream, const int32_t* d_tiles, int n, const par_color* packed,
                     par_color* frame);
/* The background (Color{background} * ambient, alt:281, 735) for n_rows whole rows starting at `rows`. */
int par_background_fill(const par_params* params, void* stream, par_color* rows, int n_rows);
/* Both of the above in one pass, for the rows the assembling rank did not render itself: rows [row_begin, row_end) of
 * the frame (`frame` addresses row 0) are written exactly once -- a packed tile's pixels where `d_map` (device array
 * of grid-x * grid-y entries: tile column + tile row * grid-x -> slot in `packed`, -1 = background) names one, the
 * background elsewhere. par_scene_tile_map makes the host copy of the map from the tile list (capacity >= grid-x *
 * grid-y entries). */
int par_tiles_assemble(const par_params* params, void* stream, const int32_t* d_map, const par_color* packed,
                       par_color* frame, int row_begin, int row_end);
int par_scene_tile_map(const par_params* params, const int32_t* tiles, int n, int32_t* map, int capacity);

/* Debug overlay of alt:763-772 (Bresenham line from the picked pixel to the light) drawn into a host frame. */
void par_debug_line(const par_params* params, const par_pixel* pick, int mouse_x, const par_light* light,
                    par_color* fb);

#ifdef __cplusplus
}
#endif
#endif /* PAR_RAYTRACER_H */
