// par_internal.h — device-side data layout and kernel launch interface (internal to libpar_raytracer.so).
#ifndef PAR_INTERNAL_H
#define PAR_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "par_raytracer.h"

// One slot of a hash bin on the device: the reference keeps `AABB bins[G*8]` and `int map[G*8]` side by side
// (alt:503-509); the AABB's 4 bytes of tail padding (alt:86-88) carry the entity index here, so a whole bin
// (8 slots) is exactly one 128-byte line.
struct par_slot {
    int16_t px, py, pz;
    int16_t ex, ey, ez;
    int32_t entity;
};
static_assert(sizeof(par_slot) == 16, "slot record must stay 16 bytes");

constexpr int PAR_STAMP_SLOTS = 8;      // time stamps per workgroup in the debug stamp buffer
constexpr int PAR_STAMP_WGS = 8192;     // workgroups per kernel that get a row in it
constexpr int PAR_STAMP_ROWS = 6;       // kernels of a frame (row 5: render_tiles_kernel)

// Kernel geometry (see DESIGN.md "kernels").
constexpr int PAR_MAX_GRID_DIM = 1024;  // per-axis bin count (bin coordinates are kept in int16)
// Per-column record built once per frame by columns_kernel and consumed by every wavefront rendering the column.
constexpr int PAR_COL_NB = 32;          // occupied bins of one column it can describe
constexpr int PAR_COL_ENT = 64;         // slot records of one column (one per lane of the rendering wavefront)
// (experiments, tools/debug/variants.sh: -DPAR_EXP_BIN_WALK=... -DPAR_EXP_COL_WALK=... ; larger records bought nothing
// on the graybox world and cost the big views)
#if !defined(PAR_EXP_BIN_WALK)
#define PAR_EXP_BIN_WALK 64
#endif
#if !defined(PAR_EXP_COL_WALK)
#define PAR_EXP_COL_WALK 160
#endif
constexpr int PAR_BIN_WALK = PAR_EXP_BIN_WALK;  // occluder records of one start bin's shadow walk
constexpr int PAR_COL_WALK = PAR_EXP_COL_WALK;  // occluder records of all walks of one column
constexpr int PAR_MIN_BIN = 8, PAR_MAX_BIN = 160;  // supported bin sizes
constexpr int PAR_TILE_MASKS = 64;      // chunks of a tile visit whose candidate masks the column record carries (one
                                        // lane of the column's wavefront each: bins up to 64 pixels a side)

// What the shading pass needs of a sprite texel besides its depth, in one 16-byte record: the normal (spr:70) and
// the palette colour the texel's index resolves to (spr:68 through color_palette, alt:352-354). Built on the host
// when sprites are uploaded; one load instead of three dependent ones.
struct par_texel {
    float nx, ny, nz;
    uint32_t rgba;
};
static_assert(sizeof(par_texel) == 16, "texel record must stay 16 bytes");

// An occluder record of a shadow walk as the render kernel's slab test wants it: the box's two planes per axis as
// floats (the coordinates are 16-bit integers and sums of two, exact in a float, and so is their difference to a
// ray origin: (float)(p - o) == (float)p - (float)o), a pair per axis so that one packed instruction handles both.
struct par_walkrec {
    float x_lo, x_hi, y_lo, y_hi, z_lo, z_hi;  // p, p + e (alt:44-46, 59-60, 71-72)
    int32_t entity;
    int32_t pad_;
};
static_assert(sizeof(par_walkrec) == 32, "walk record layout");

// An entry of a column as the TILE pass of the render kernel reads it: one scalar load of eight dwords per candidate
// entry, every field in the form the per-pixel test (alt:310-346) consumes, nothing to unpack or add up per chunk.
struct par_xent {
    int32_t px4;     // 4 * px: the pixel's column and the texel offsets are kept in bytes of an int32 table
    int32_t dims;    // 4 * ex | (ey + ez) << 8: width (bytes) and height of the sprite rectangle, alt:310-317
    int32_t top;     // py + ey + pz + ez: sprite row = top - world_j, alt:324-326
    int32_t k;       // ey - top: min(0, ey - sprite row) = min(0, k + world_j), alt:338-340
    int32_t dbase;   // py - pz, alt:336-337
    int32_t pz;      // the pixel's z = pz + texel depth, its y = world_j - z (alt:356-361)
    int32_t entity;  // alt:363
    int32_t bzk;     // bin_z | run << 16; run = empty stretches of the column before the entry's bin (alt:298-300)
};
static_assert(sizeof(par_xent) == 32, "one s_load_dwordx8 per entry");

struct par_colrec_nb {
    int16_t bz;          // bin_z of an occupied bin of the column, ascending
    uint8_t off, cnt;    // its records: entries[off, off+cnt)
    int16_t woff, wcnt;  // the shadow walk that starts in it: walk[woff, woff+wcnt); wcnt -1: not recorded (too long)
};
struct par_colrec {
    int16_t n_nb, n_entries, n_walk;
    int16_t overflow;    // 1: the column does not fit this record; the generic kernel renders it
    int16_t bx, by;      // the column
    int32_t tile_mode;   // 1: visit the column's pixels as whole tiles, 0: entry rectangle by entry rectangle
    int32_t chunks;      // 64-pixel chunks of the column's visit (what its render work is split by)
    uint32_t dup_lo, dup_hi;  // bit e: entry e repeats an earlier entry's entity (same rectangle: it owns no pixel)
    int32_t pad_;
    par_colrec_nb nb[PAR_COL_NB];
    int16_t ebz[PAR_COL_ENT];  // bin_z of each entry (the primary pass walks the entries as one flat list)
    par_slot entries[PAR_COL_ENT];
    par_walkrec walk[PAR_COL_WALK];
    // the tile pass's view of the entries (written for columns visited as whole tiles):
    uint32_t rect[PAR_COL_ENT];  // the entry's sprite rectangle clipped to the tile, relative to the tile's corner
                                 // (bx * B, by * B): row0 | row1 << 8 | col0 << 16 | col1 << 24, ends exclusive
    par_xent xent[PAR_COL_ENT];
    uint64_t cmask[PAR_TILE_MASKS];  // per 64-pixel chunk of the tile visit (when there are at most PAR_TILE_MASKS):
                                     // bit e = entry e can cover a pixel of the chunk (its rectangle meets the chunk's
                                     // box and it repeats no earlier entry's entity)
};
static_assert(sizeof(par_colrec_nb) == 8 && sizeof(par_colrec) % 16 == 0, "column record layout");
static_assert(PAR_COL_ENT <= 64, "one duplicate bit per entry, one entry per lane");
static_assert((PAR_COL_NB & (PAR_COL_NB - 1)) == 0 && PAR_COL_NB <= 64, "one occupied bin per lane");
static_assert(PAR_COL_WALK % 2 == 0, "a walk list takes an even number of records (the tile pass reads them in pairs)");

// The shadow walk of BACKGROUND pixels (every ray traced as the reference does): an uncovered pixel has world
// position (x, 0, 0) (alt:281, 707-709), so its ray starts in bin (x / B, H / B, 0) whatever its row -- one walk per
// bin column bx, one ray per x.
struct par_bgwalk {
    int32_t cnt;  // records, or -1 when they did not fit
    int32_t pad_[3];
    par_slot rec[PAR_BIN_WALK];
};

// Per-frame values that change without the scene being re-uploaded. In the hipGraph path they live in device memory
// (updated by a memcpy node); otherwise they travel as kernel arguments.
struct par_frame_dyn {
    int32_t lx, ly, lz;     // lights[0] position (alt:712-714)
    int32_t lbx, lby, lbz;  // its bin (alt:729-732)
};

// Render flags that make the render launch use its instrumented variant (ray counting, the timing-experiment bits
// 24-26 and the time stamps, bit 29); a production frame has none of them and runs kernels compiled without them.
constexpr uint32_t PAR_DEBUG_FLAGS = PAR_RENDER_COUNT_RAYS | (7u << 24) | (1u << 29);
constexpr int PAR_WAVE_NW = 4;          // wavefronts per render workgroup
// Render work items: columns_kernel lists every 64-pixel chunk of every column with a record as one item (par_item);
// pass = the entry whose rectangle is visited, PAR_ITEM_TILE = the whole tile. An item of a SIMPLE column (all its
// entries are one entity, every shadow walk from its bins met no occupied bin: most columns of a sparse scene)
// carries all the render kernel needs, which then never touches the column's record. The list is kept in PAR_ITEM_SHARDS shards (column index mod shards), each with its own counter, so
// that the column workgroups' appends do not queue up on one address; wavefront w of the render launch takes items
// w / shards, + waves / shards, ... of shard w mod shards.
constexpr int PAR_ITEM_LISTS = 2;     // 0: entry passes (render_items_kernel), 1: whole-tile visits (render_tiles_kernel)
constexpr int PAR_ITEM_SHARD_BITS = 6;
constexpr int PAR_ITEM_SHARDS = 1 << PAR_ITEM_SHARD_BITS;
constexpr int PAR_ITEM_COUNTER_STRIDE = 32;  // int32 words between two shard counters: one 128-byte line each
constexpr uint32_t PAR_ITEM_TILE = 0xFFFFu;
constexpr uint32_t PAR_ITEM_NONE = 0xFFFFFFFFu;  // a reserved item slot whose column went to the overflow list
struct par_item {
    uint32_t ci;        // index of the column (list and record), or PAR_ITEM_NONE
    uint32_t visit;     // (pass << 16) | chunk of the pass
    uint32_t where;     // bx | by << 10 | PAR_ITEM_SIMPLE
    uint32_t bins;      // simple columns: first | last << 16 of the (contiguous) occupied bins; tile items: the
                        // number of consecutive 64-pixel chunks the item covers, from chunk `visit & 0xFFFF` on
    par_slot entry;     // the pass's entry (entry passes)
};
static_assert(sizeof(par_item) == 32, "work item layout");
constexpr uint32_t PAR_ITEM_SIMPLE = 1u << 31;
struct par_grid_dev {
    int32_t gx, gy, gz, volume;
    int32_t* head[2];         // [volume] node index + 1 of the most recent insertion, 0 = none
    uint8_t* count[2];        // [volume] visible count = insertions & 7 (alt:262-264)
    int32_t* colflag[2];      // [gx*gy] 1 when some bin of screen column (bx, by) shows entries this frame
    par_slot* slots;          // [volume * 8]
    int32_t* node_entity[2];  // [capacity]
    int32_t* node_next[2];    // [capacity]
    int32_t* node_bin[2];     // [capacity]
    int32_t* node_counter;    // [2]
    int32_t* col_list;        // [gx*gy] occupied columns (bx*gy + by) inside the rendered row range, unordered
    int32_t* counters;        // [PAR_CNT_TOTAL]: occupied columns, overflowed columns (reset by insert)
    par_colrec* colrec;       // [col_capacity] indexed like col_list
    par_item* items;          // [PAR_ITEM_LISTS * PAR_ITEM_SHARDS * item_capacity] render work items, list by list and
                              // shard by shard: a 64-pixel chunk of an entry pass, or tile_k chunks of a tile visit
    int32_t* item_counters;   // [PAR_ITEM_LISTS * PAR_ITEM_SHARDS * PAR_ITEM_COUNTER_STRIDE] items per shard (reset by insert)
    int32_t* build_sync;      // [64] barrier words of build_fill_kernel (arrived, left; they reset themselves)
    int32_t* slow_list;       // [gx*gy] indices into col_list of the columns that overflowed their record
    par_bgwalk* bgwalk;       // [gx] shadow walks of the background rays (traced only on request)
    uint8_t* bglit;           // [width] result of the background ray of screen column x
    unsigned long long* stamps;  // debug (PAR_DEBUG_STAMPS=1): per workgroup phase time stamps, else nullptr
    int32_t capacity;
    int32_t col_capacity;
    int32_t item_capacity;    // work items one shard of `items` holds
};

struct par_bin_args {
    int32_t W, H, L, B;
    int32_t n;
    int32_t set;             // which head/count/node/colflag set this frame uses
    int32_t by_lo, by_hi;    // bin rows [by_lo, by_hi] the render of this frame touches (column-list filter)
    uint32_t flags;          // render flags (bit 29: debug time stamps)
    int32_t test_lose_wg;    // tests (PAR_TEST_LOSE_BUILD_WG=1): build workgroup 0 never arrives at the barrier
    uint32_t magic_b;        // floor(n / B) == __umulhi(n, magic_b) for n * B < 2^32 (as par_render_args::magic_b)
    const par_aabb* aabbs;
};

struct par_render_args {
    int32_t W, H, B;
    int32_t row_begin, row_end;    // rows rendered by this launch
    int32_t by_lo, by_hi;          // bin rows touched
    int32_t set;                   // grid set of this frame
    int32_t trace_bg;              // 1: background shadow rays are traced too (flag, or lit plane requested)
    int32_t dense;                 // 1 (PAR_FORCE_GENERIC=1, tests): every column is rendered as if it had no record
    uint32_t magic_b;              // floor(n / B) == __umulhi(n, magic_b) for n * B < 2^32
    float ambient;
    uint32_t background;           // gray level (alt:281)
    uint32_t flags;
    int32_t tile_k;                // 64-pixel chunks per work item of a column visited as a whole tile; 0: the frame
                                   // has no launch for tile items (a sparse frame): every column is visited entry by entry
    uint32_t tile_k_magic;         // floor(n / tile_k) == (n * magic) >> 16 for n < 1024
    int32_t overflow_launched;     // 0: the frame has no launch for the overflow list (the host ruled it out)
    par_frame_dyn dyn;             // used when dyn_ptr == nullptr
    const par_frame_dyn* dyn_ptr;  // graph path
    const uint8_t* count;
    const par_slot* slots;
    const par_sprite* sprites;
    const par_texel* texinfo;      // [n_sprites * 800]
    const int32_t* sprite_ids;     // nullable
    const par_color* palette;
    par_outputs out;               // device pointers, addressing (row_begin, 0)
    unsigned long long* ray_counter;
};

// PAR_CNT_ERROR is STICKY: kernels only ever set bits in it (a frame's insert resets the other counters, not this
// one); the host clears it when it reports it (PAR_ERR_DEVICE).
enum { PAR_CNT_COLS = 0, PAR_CNT_SLOW = 1, PAR_CNT_ERROR = 2, PAR_CNT_TOTAL = 8 };
enum { PAR_DEVERR_BARRIER = 1,    // build_fill_kernel: a build workgroup never arrived at the barrier (bounded wait)
       PAR_DEVERR_OVERFLOW = 2 }; // a column went onto the overflow list in a frame without a launch for that list

// The background fill split over the frame's first three launches: 512-pixel chunks [cut[i], cut[i+1]) go with
// launch i (hash insert, hash resolve, column records).
struct par_fill_plan {
    uint32_t out_rgba;  // Color{127,127,127,0} * ambient
    int32_t cut[4];
};

// Launchers (par_kernels.hip). All asynchronous on `stream`.
// True when the fill can ride along with the first three launches (else: par_launch_fill on its own).
bool par_plan_fill(const par_render_args& a, par_fill_plan* plan);
// `fill` (nullable, with the render args `fa`): this launch also carries its share of the fill.
hipError_t par_launch_bin_insert(const par_grid_dev& g, const par_bin_args& a, const par_render_args* fa,
                                 const par_fill_plan* fill, hipStream_t stream);
// Both of the above in one launch for small scenes (else hipErrorNotSupported, nothing launched).
hipError_t par_launch_build(const par_grid_dev& g, const par_bin_args& a, int64_t pair_bound, const par_render_args* fa,
                            const par_fill_plan* fill, hipStream_t stream);
hipError_t par_launch_bin_resolve(const par_grid_dev& g, const par_bin_args& a, int64_t pair_bound,
                                  const par_render_args* fa, const par_fill_plan* fill, hipStream_t stream);
// Per occupied column: compact slot list + the shadow walks of its bins (+ the background walks when a.trace_bg);
// then, when a.trace_bg, the background rays themselves (one per x).
hipError_t par_launch_columns(const par_grid_dev& g, const par_render_args& a, int64_t column_bound, hipStream_t stream);
// Column records + the last share of the fill in one launch.
hipError_t par_launch_columns_fill(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                   const par_fill_plan& fill, hipStream_t stream);
// Background for every pixel of the row range; the render kernel then overwrites the pixels primitives cover.
// Independent of the hash.
hipError_t par_launch_fill(const par_grid_dev& g, const par_render_args& a, hipStream_t stream);
// `item_bound`: an upper bound of the frame's work items (64-pixel chunks of the columns with a record).
hipError_t par_launch_render(const par_grid_dev& g, const par_render_args& a, int64_t item_bound,
                             hipStream_t stream);
// The columns visited as whole tiles (a.tile_k > 0: a dense frame); `item_bound` as above (in chunks).
hipError_t par_launch_render_tiles(const par_grid_dev& g, const par_render_args& a, int64_t item_bound,
                                   hipStream_t stream);
// All render kernels in one launch for small frames (else hipErrorNotSupported, nothing launched).
hipError_t par_launch_render_both(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                  int64_t item_bound, bool may_overflow, hipStream_t stream);
// The columns that overflowed their record (every column when a.dense).
hipError_t par_launch_render_overflow(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                      hipStream_t stream);

// Sharded frames: tiles between a frame block and packed slots, and the background colour for whole rows.
hipError_t par_launch_tiles_copy(bool pack, const int32_t* d_tiles, int n, int W, int H, int B, int row_begin, int row_end,
                                 const void* src, void* dst, hipStream_t stream);
hipError_t par_launch_background(void* dst, size_t n_px, uint32_t rgba, hipStream_t stream);
hipError_t par_launch_tiles_assemble(const int32_t* d_map, int gx, int W, int B, int row_begin, int row_end,
                                     const void* packed, void* frame, uint32_t rgba, hipStream_t stream);

// Test hook: the device functions slab_hit / color_scale / normalize_l1_and_inverse on caller-supplied vectors (device pointers).
hipError_t par_launch_units(int kind, const void* in_a, const void* in_b, int n, void* out, hipStream_t stream);

#endif
