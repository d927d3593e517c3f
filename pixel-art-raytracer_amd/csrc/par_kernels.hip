// par_kernels.hip — hand-written HIP kernels for gfx950 (MI355X / CDNA4): spatial-hash build + the fused
// primary-ray / shadow-ray / shade / quantise kernel.
//
// What each kernel restates (alt = src/alternative.cpp, spr = src/sprites.hpp of the reference):
//   bin_insert_kernel + bin_resolve_kernel   memset alt:690 + count_entities_in_bins alt:195-269
//   render_kernel                            trace_hash_for_pixel alt:271-397, the shading loop alt:702-760,
//                                            trace_hash_for_light alt:399-500, AABB::intersect alt:40-83,
//                                            Vector::normalize spr:28-35, Color::operator* spr:8-16
//
// Float discipline: this file is compiled with -ffp-contract=off; divisions are hipcc's default correctly-rounded
// fp32 division; min/max are written as the ?: forms of std::min/std::max so NaN handling follows the reference
// (first argument wins). With that every float the reference computes is reproduced bit for bit.
#include "par_internal.h"

#include <limits.h>

namespace {

__device__ __forceinline__ int flat_index(int gy, int gz, int x, int y, int z) {
    return x * gy * gz + y * gz + z;  // index_into_view_hash alt:180-182
}

// ------------------------------------------------------------------------------------------------------------
// Spatial hash build.
//
// The reference inserts entities one after the other; a bin's counter wraps at 8 (alt:262-264), so after k
// insertions the bin shows c = k & 7 entries and they are exactly the LAST c insertions, in insertion (= entity
// index) order. That closed form is order-independent, so the build is parallel and still deterministic:
//   insert : every (entity, bin) pair pushes a node on the bin's lock-free list (atomicExch on the head);
//   resolve: the thread owning a bin's head walks the list, counts k and keeps the 7 largest entity indices.
// Two head/count/node sets alternate between frames; `insert` of frame f also wipes the bins frame f-1 touched in
// the other set, so no O(volume) memset is ever issued after context creation.
// ------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void bin_insert_kernel(par_grid_dev g, par_bin_args a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
    const int s = a.set, o = a.set ^ 1;

    // wipe what the previous frame touched in the other set
    const int prev = g.node_counter[o];
    for (int i = tid; i < prev; i += stride) {
        const int b = g.node_bin[o][i];
        g.head[o][b] = 0;
        g.count[o][b] = 0;
    }

    const int W = a.W, H = a.H, L = a.L, B = a.B;
    for (int e = tid; e < a.n; e += stride) {
        const par_aabb box = a.aabbs[e];
        const int minx = box.px, miny = box.py, minz = box.pz;                       // alt:202-204
        const int maxx = minx + box.ex, maxy = miny + box.ey, maxz = minz + box.ez;  // alt:206-208
        if ((maxx < 0) || (minx >= W) || (maxy < 0 - maxz) || (miny >= H - minz + B) || (maxz < -box.ez - B) ||
            (minz > L + B)) {
            continue;  // cull, alt:212-219
        }
        const int x0 = max(0, minx / B);                          // alt:222
        const int y0 = max(0, (H - maxy - maxz) / B);             // alt:223-225
        const int z0 = max(0, minz / B);                          // alt:226
        const int x1 = min(g.gx, (maxx + B - 1) / B);             // alt:228-230
        const int y1 = min(g.gy, (H - miny - minz + B - 1) / B);  // alt:231-236
        const int z1 = min(g.gz, (maxz + B - 1) / B);             // alt:238-240
        for (int bx = x0; bx < x1; bx++) {
            for (int by = y0; by < y1; by++) {
                for (int bz = z0; bz < z1; bz++) {
                    const int b = flat_index(g.gy, g.gz, bx, by, bz);
                    const int node = atomicAdd(&g.node_counter[s], 1);
                    if (node < g.capacity) {  // the host sizes the pool from the exact pair count; belt and braces
                        g.node_entity[s][node] = e;
                        g.node_bin[s][node] = b;
                        g.node_next[s][node] = atomicExch(&g.head[s][b], node + 1);
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void bin_resolve_kernel(par_grid_dev g, par_bin_args a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = a.set;
    // insert (the previous kernel) has consumed the other set's counter: free it for the next frame's inserts
    if (tid == 0) g.node_counter[s ^ 1] = 0;
    const int n_nodes = min(g.node_counter[s], g.capacity);
    if (tid >= n_nodes) return;
    const int b = g.node_bin[s][tid];
    if (g.head[s][b] != tid + 1) return;  // only the most recent insertion resolves its bin

    int top0 = -1, top1 = -1, top2 = -1, top3 = -1, top4 = -1, top5 = -1, top6 = -1;  // 7 largest, descending
    int k = 0;
    for (int cur = tid + 1; cur != 0; cur = g.node_next[s][cur - 1]) {
        int e = g.node_entity[s][cur - 1];
        int t;
        // bubble the new index through the sorted registers (static indexing keeps them out of scratch)
        if (e > top0) { t = top0; top0 = e; e = t; }
        if (e > top1) { t = top1; top1 = e; e = t; }
        if (e > top2) { t = top2; top2 = e; e = t; }
        if (e > top3) { t = top3; top3 = e; e = t; }
        if (e > top4) { t = top4; top4 = e; e = t; }
        if (e > top5) { t = top5; top5 = e; e = t; }
        if (e > top6) { t = top6; top6 = e; e = t; }
        k++;
    }
    const int c = k & (PAR_SLOTS - 1);  // alt:262-264
    const int top[7] = {top0, top1, top2, top3, top4, top5, top6};
#pragma unroll
    for (int i = 0; i < 7; i++) {
        if (i < c) {  // slot c-1-i holds the (i+1)-th largest index: slots ascend in insertion order
            const int e = top[i];
            const par_aabb box = a.aabbs[e];
            par_slot rec;
            rec.px = box.px; rec.py = box.py; rec.pz = box.pz;
            rec.ex = box.ex; rec.ey = box.ey; rec.ez = box.ez;
            rec.entity = e;
            g.slots[(size_t)b * PAR_SLOTS + (c - 1 - i)] = rec;
        }
    }
    g.count[s][b] = (uint8_t)c;
}

// ------------------------------------------------------------------------------------------------------------
// Render kernel: one workgroup (5 wavefronts) per screen tile = one hash-bin column footprint (B x R pixels).
// ------------------------------------------------------------------------------------------------------------

struct RenderShared {
    par_slot entries[PAR_MAX_ENTRIES];  // the column's slot records, ordered by (bin_z, slot)
    par_slot occ[PAR_MAX_OCC];          // shadow-occluder candidates of the current probe chunk
    int16_t nb_bz[PAR_MAX_GRID_DIM];    // non-empty bins of the column: bin_z ...
    int16_t nb_off[PAR_MAX_GRID_DIM];   // ... first entry (low 13 bits) -- count lives in nb_cnt
    uint8_t nb_cnt[PAR_MAX_GRID_DIM];   // ... visible count
    int32_t chain[3][PAR_CHAIN_ITERS + 1];  // truncated bin coordinates of the walk, per axis
    float chain_carry[3];
    int32_t wsum[PAR_NT / 64];
    int32_t gkey[2];
    int32_t n_nb;
    int32_t n_entries;
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}

// Exclusive prefix sum over the workgroup; two barriers. `total` is uniform.
__device__ __forceinline__ int block_excl_scan(int v, int32_t* wsum, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < PAR_NT / 64; i++) {
        const int s = wsum[i];
        if (i < w) base += s;
        total += s;
    }
    __syncthreads();
    return base + incl - v;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}

// std::min / std::max on floats: (b<a)?b:a and (a<b)?b:a -- the first argument survives a NaN (SURVEY a-5).
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

// AABB::intersect, alt:40-83.
__device__ __forceinline__ bool slab_hit(const par_slot& r, int ox, int oy, int oz, float ix, float iy, float iz) {
    const float x1 = (float)(r.px - ox) * ix;
    const float x2 = (float)(r.px + r.ex - ox) * ix;
    float tmin = std_min(x1, x2);
    float tmax = std_max(x1, x2);
    const float y1 = (float)(r.py - oy) * iy;
    const float y2 = (float)(r.py + r.ey - oy) * iy;
    tmin = std_max(tmin, std_min(y1, y2));
    tmax = std_min(tmax, std_max(y1, y2));
    const float z1 = (float)(r.pz - oz) * iz;
    const float z2 = (float)(r.pz + r.ez - oz) * iz;
    tmin = std_max(tmin, std_min(z1, z2));
    tmax = std_min(tmax, std_max(z1, z2));
    return tmax >= tmin;
}

// Color::operator*, spr:8-16: truncating per-channel scale, alpha passed through. `c` is RGBA little-endian.
__device__ __forceinline__ uint32_t color_scale(uint32_t c, float v) {
    const uint32_t r = (uint32_t)(uint8_t)((float)(c & 0xFF) * v);
    const uint32_t g = (uint32_t)(uint8_t)((float)((c >> 8) & 0xFF) * v);
    const uint32_t b = (uint32_t)(uint8_t)((float)((c >> 16) & 0xFF) * v);
    return r | (g << 8) | (b << 16) | (c & 0xFF000000u);
}

// (sy, sz) of a start bin as one sortable word; bin coordinates are far inside +-32768 (positions are `short`).
__device__ __forceinline__ int pack_key(int sy, int sz) {
    return (int)((((uint32_t)(sy + 16384) & 0x7FFFu) << 16) | ((uint32_t)(sz + 32768) & 0xFFFFu));
}

// Trunc-toward-zero division by the bin size through a precomputed reciprocal: exact for |n| * B < 2^32
// (|n| <= 3 * 32767 here and B <= 1600).
__device__ __forceinline__ int div_bin(int n, uint32_t magic) {
    const int q = (int)__umulhi((uint32_t)(n < 0 ? -n : n), magic);
    return n < 0 ? -q : q;
}

// A tile no primitive can cover: every requested plane gets its background constant.
__device__ __forceinline__ void fill_background_tile(const par_render_args& a, int r0, int r1, int c0, int tw,
                                                     uint32_t out_rgba, uint32_t bg_rgba, float ambient) {
    const int tid = threadIdx.x;
    const int W = a.W, rows = r1 - r0;
    const size_t base = (size_t)(r0 - a.row_begin) * W + c0;
    if (a.out.fb) {
        uint32_t* fb = reinterpret_cast<uint32_t*>(a.out.fb);
        if (((W | c0 | tw) & 3) == 0 && (reinterpret_cast<uintptr_t>(fb) & 15) == 0) {
            const int nvec = tw >> 2;  // 16-byte stores: 4 pixels per lane
            const uint4 v = make_uint4(out_rgba, out_rgba, out_rgba, out_rgba);
            for (int i = tid; i < rows * nvec; i += PAR_NT) {
                const int ry = i / nvec, vx = i - ry * nvec;
                *reinterpret_cast<uint4*>(fb + base + (size_t)ry * W + vx * 4) = v;
            }
        } else {
            for (int i = tid; i < rows * tw; i += PAR_NT) {
                const int ry = i / tw, x = i - ry * tw;
                fb[base + (size_t)ry * W + x] = out_rgba;
            }
        }
    }
    if (a.out.palidx) {
        uint8_t* pal = a.out.palidx;
        if (((W | c0 | tw) & 7) == 0 && (reinterpret_cast<uintptr_t>(pal) & 7) == 0) {
            const int nvec = tw >> 3;  // 8-byte stores: 8 pixels per lane
            const uint2 v = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            for (int i = tid; i < rows * nvec; i += PAR_NT) {
                const int ry = i / nvec, vx = i - ry * nvec;
                *reinterpret_cast<uint2*>(pal + base + (size_t)ry * W + vx * 8) = v;
            }
        } else {
            for (int i = tid; i < rows * tw; i += PAR_NT) {
                const int ry = i / tw, x = i - ry * tw;
                pal[base + (size_t)ry * W + x] = PAR_PALIDX_BACKGROUND;
            }
        }
    }
    if (a.out.brightness || a.out.gbuf) {
        par_pixel px;
        px.normal = par_vec3{0.f, 0.f, 0.f};
        px.color.red = px.color.green = px.color.blue = (uint8_t)(bg_rgba & 0xFF);
        px.color.alpha = 0;
        px.y = 0; px.z = 0; px.entity_index = 0;
        for (int i = tid; i < rows * tw; i += PAR_NT) {
            const int ry = i / tw, x = i - ry * tw;
            const size_t o = base + (size_t)ry * W + x;
            if (a.out.brightness) a.out.brightness[o] = ambient;
            if (a.out.gbuf) a.out.gbuf[o] = px;
        }
    }
}

__global__ __launch_bounds__(PAR_NT) void render_kernel(par_grid_dev g, par_render_args a) {
    __shared__ RenderShared sm;
    const int tid = threadIdx.x;
    const int W = a.W, H = a.H, B = a.B;

    // ---- tile decode (uniform) ----------------------------------------------------------------------------
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2. Tiles that are neighbours in x share
    // 128-byte lines of the frame (a 40-pixel tile row is 160 bytes), so consecutive logical tiles are given to
    // ONE XCD: its L2 then merges the partial lines before they are written back. Bijective for any grid size.
    int t_id;
    {
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        const int q = nb >> 3, rem = nb & 7, xcd = b & 7;
        t_id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (b >> 3);
    }
    const int bx = t_id % g.gx;
    const int trow = t_id / g.gx;
    const int by = a.by_begin + trow / a.subs;
    const int sub = trow - (trow / a.subs) * a.subs;
    const int tile_r0 = by * B + sub * a.tile_rows;
    const int r0 = max(tile_r0, a.row_begin);
    const int r1 = min(min(tile_r0 + a.tile_rows, (by + 1) * B), min(H, a.row_end));
    if (r0 >= r1) return;
    const int c0 = bx * B;
    const int tw = min(B, W - c0);
    const bool trace_bg = (a.flags & PAR_RENDER_TRACE_BACKGROUND) != 0 || a.out.lit != nullptr;
    const float ambient = a.ambient;
    const uint32_t bg_rgba = a.background | (a.background << 8) | (a.background << 16);
    const int col_base = flat_index(g.gy, g.gz, bx, by, 0);

    // ---- phase 0: empty column -> the tile is background (alt:281, 735): constant fill, wide coalesced stores ---
    if (!trace_bg) {
        int any = 0;
        for (int t = tid; t < g.gz; t += PAR_NT) any |= a.count[col_base + t];
        if (!__syncthreads_or(any)) {
            fill_background_tile(a, r0, r1, c0, tw, color_scale(bg_rgba, ambient), bg_rgba, ambient);
            return;
        }
    }
    const par_frame_dyn dyn = a.dyn_ptr ? *a.dyn_ptr : a.dyn;

    if (tid == 0) {
        sm.gkey[0] = INT_MAX;
        sm.gkey[1] = INT_MAX;
    }

    // ---- phase 1: the column (bx, by, *) -> ordered list of non-empty bins and their slot records in LDS -----
    int nb_base = 0, ent_base = 0;
    for (int t0 = 0; t0 < g.gz; t0 += PAR_NT) {
        const int t = t0 + tid;
        const int c = (t < g.gz) ? (int)a.count[col_base + t] : 0;  // consecutive bytes: coalesced
        int total;
        const int packed = block_excl_scan(((c != 0) << 16) | c, sm.wsum, total);
        const int nb_i = nb_base + (packed >> 16);
        const int off = ent_base + (packed & 0xFFFF);
        if (c != 0) {
            sm.nb_bz[nb_i] = (int16_t)t;
            sm.nb_off[nb_i] = (int16_t)off;
            sm.nb_cnt[nb_i] = (uint8_t)c;
            const par_slot* src = a.slots + (size_t)(col_base + t) * PAR_SLOTS;
            for (int k = 0; k < c; k++) {
                if (off + k < PAR_MAX_ENTRIES) sm.entries[off + k] = src[k];
            }
        }
        nb_base += total >> 16;
        ent_base += total & 0xFFFF;
    }
    if (tid == 0) {
        sm.n_nb = nb_base;
        sm.n_entries = ent_base;
    }
    __syncthreads();
    const int n_nb = nb_base;

    // ---- per-pixel registers ------------------------------------------------------------------------------
    bool valid[PAR_KPT];
    int col[PAR_KPT], row[PAR_KPT];
    bool hit[PAR_KPT];
    int p_entity[PAR_KPT], p_y[PAR_KPT], p_z[PAR_KPT], p_tex[PAR_KPT];  // p_tex = sprite id * 800 + texel
#pragma unroll
    for (int k = 0; k < PAR_KPT; k++) {
        const uint32_t p = (uint32_t)(tid + k * PAR_NT);
        const int py = (int)__umulhi(p, a.magic_b);
        const int px = (int)p - py * B;
        col[k] = c0 + px;
        row[k] = tile_r0 + py;
        valid[k] = (px < tw) && (row[k] >= r0) && (row[k] < r1);
        hit[k] = false;
        p_entity[k] = 0;  // background texel: normal 0, y = z = 0, entity_index 0 (alt:281)
        p_y[k] = 0;
        p_z[k] = 0;
        p_tex[k] = 0;
    }

    // ---- phase 2: primary ray, alt:271-397 -----------------------------------------------------------------
    if (n_nb > 0 && !(a.flags & (1u << 29))) {  // bit 29: ablation, no primary pass
#pragma unroll
        for (int k = 0; k < PAR_KPT; k++) {
            const int i = col[k];
            const int world_j = (int)(int16_t)(H - row[k]);  // alt:280
            int adjacent = 0;                                // alt:282
            int closest = INT_MIN;                           // alt:289
            int prev_bz = -2;
            bool done = !valid[k];
            int w_ybase = 0, w_pz = 0, w_d = 0;
            for (int n = 0; n < n_nb; n++) {
                if (__all(done)) break;  // the whole wavefront has its two adjacent hit bins (alt:372-374)
                const int bz = sm.nb_bz[n];
                const int cnt = sm.nb_cnt[n];
                const int off = sm.nb_off[n];
                if (!done) {
                    if (bz != prev_bz + 1) adjacent = 0;  // an empty bin lies in between (alt:298-300)
                    prev_bz = bz;
                    bool hit_in_bin = false;
                    for (int s = 0; s < cnt; s++) {
                        const int ei = off + s;
                        const par_slot rec = (ei < PAR_MAX_ENTRIES)
                                                 ? sm.entries[ei]
                                                 : a.slots[(size_t)(col_base + bz) * PAR_SLOTS + s];
                        const int top = rec.py + rec.ey + rec.pz + rec.ez;
                        if (i >= rec.px && i < rec.px + rec.ex && world_j > rec.py + rec.pz && world_j <= top) {
                            const int sprite_row = top - world_j;                       // alt:324-326
                            const int t = sprite_row * PAR_SPRITE_W + (i - rec.px);     // alt:330-332
                            const int sid = a.sprite_ids ? a.sprite_ids[rec.entity] : 0;  // alt:321-322
                            const int d = a.sprites[sid].depth[t];
                            const int depth = rec.py - rec.pz + min(0, rec.ey - sprite_row) - d;  // alt:336-341
                            if (closest >= depth) continue;                             // alt:344-346
                            closest = depth;
                            w_ybase = rec.py + rec.ey + rec.ez - sprite_row;            // alt:356-359
                            w_pz = rec.pz;                                              // alt:360-361
                            w_d = d;
                            p_entity[k] = rec.entity;                                   // alt:363
                            p_tex[k] = sid * PAR_SPRITE_TEXELS + t;
                            hit[k] = true;
                            hit_in_bin = true;                                          // alt:365
                        }
                    }
                    adjacent += hit_in_bin ? 1 : 0;  // alt:368
                    if (adjacent >= 2) done = true;  // alt:372-374
                }
            }
            if (hit[k]) {
                p_y[k] = w_ybase - w_d;
                p_z[k] = w_pz + w_d;
            }
        }
    }

    // ---- phase 3: shading set-up, alt:704-735 --------------------------------------------------------------
    float inv_x[PAR_KPT], inv_y[PAR_KPT], inv_z[PAR_KPT];
    float b_lit[PAR_KPT];       // brightness if the light is reached: min(1, diffuse + ambient), alt:745-758
    uint32_t rgba[PAR_KPT];     // palette colour of the texel (alt:352-354) or the background gray
    int key[PAR_KPT];
    bool pend[PAR_KPT], lit[PAR_KPT];
    int n_traced = 0;
#pragma unroll
    for (int k = 0; k < PAR_KPT; k++) {
        float nx = 0.f, ny = 0.f, nz = 0.f;
        rgba[k] = bg_rgba;
        if (hit[k]) {
            const int sid = p_tex[k] / PAR_SPRITE_TEXELS;
            const int t = p_tex[k] - sid * PAR_SPRITE_TEXELS;
            const par_vec3 n = a.sprites[sid].normal[t];  // alt:349-350
            nx = n.x; ny = n.y; nz = n.z;
            const par_color pc = a.palette[a.sprites[sid].color[t]];
            rgba[k] = (uint32_t)pc.red | ((uint32_t)pc.green << 8) | ((uint32_t)pc.blue << 16) |
                      ((uint32_t)pc.alpha << 24);
        }
        pend[k] = valid[k] && (hit[k] || trace_bg);
        lit[k] = true;
        // Background pixels whose shadow ray is skipped keep brightness = ambient: with a zero normal
        // min(1, max(0, 0 * t) + ambient) is ambient whether or not the light is reached (SURVEY a-6).
        inv_x[k] = inv_y[k] = inv_z[k] = 0.f;
        b_lit[k] = ambient;
        key[k] = INT_MAX;
        if (pend[k]) {
            const int wx = col[k], wy = p_y[k], wz = p_z[k];  // alt:707-709
            // towards_light = normalize_L1(light - world), alt:711-715 + spr:28-35
            const float dx = (float)(dyn.lx - wx), dy = (float)(dyn.ly - wy), dz = (float)(dyn.lz - wz);
            const float len = __builtin_fabsf(dx) + __builtin_fabsf(dy) + __builtin_fabsf(dz);
            const float tx = dx / len, ty = dy / len, tz = dz / len;
            inv_x[k] = 1.f / tx;  // alt:717-719
            inv_y[k] = 1.f / ty;
            inv_z[k] = 1.f / tz;
            const float dot = nx * tx + ny * ty + nz * tz;           // alt:746-747 (no contraction)
            const float diffuse = std_max(0.f, dot);                 // alt:745
            b_lit[k] = std_min(1.f, diffuse + ambient);              // alt:758
            const int sy = div_bin(H - wy - wz, a.magic_b);          // alt:725-726
            const int sz = div_bin(wz, a.magic_b);                   // alt:727
            key[k] = pack_key(sy, sz);
        }
        n_traced += pend[k] ? 1 : 0;
    }
    if ((a.flags & PAR_RENDER_COUNT_RAYS) && a.ray_counter) {
        // one atomic per wavefront
        int s = n_traced;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if ((tid & 63) == 0 && s) atomicAdd(a.ray_counter, (unsigned long long)s);
    }

    // ---- phase 4: shadow rays, alt:738-742 + alt:399-500 ---------------------------------------------------
    // Pixels of the tile whose walk starts in the same bin visit the same bins (the probe sequence depends only on
    // the start and light bins), so the walk is done once per distinct start bin by the whole workgroup; the
    // occupied bins it finds are staged in LDS and every pixel of the group slab-tests that list. The result of
    // trace_hash_for_light is an OR over probes, hence independent of probe order and of duplicate probes.
    for (int iter = 0;; iter++) {
        if (a.flags & (1u << 30)) break;  // ablation: no shadow pass (timing experiments only)
        int mykey = INT_MAX;
#pragma unroll
        for (int k = 0; k < PAR_KPT; k++) {
            if (pend[k]) mykey = min(mykey, key[k]);
        }
        mykey = wave_min(mykey);
        if ((tid & 63) == 0 && mykey != INT_MAX) atomicMin(&sm.gkey[iter & 1], mykey);
        if (tid == 0) sm.gkey[(iter + 1) & 1] = INT_MAX;
        __syncthreads();
        const int cur = sm.gkey[iter & 1];
        if (cur == INT_MAX) break;  // uniform

        const int sx = bx;  // world_x / B, alt:724
        const int sy = (cur >> 16) - 16384;
        const int sz = (cur & 0xFFFF) - 32768;
        bool mine[PAR_KPT];
#pragma unroll
        for (int k = 0; k < PAR_KPT; k++) mine[k] = pend[k] && key[k] == cur;

        // alt:406-430
        const float fsx = (float)sx, fsy = (float)sy, fsz = (float)sz;
        const float ddx = (float)dyn.lbx - fsx, ddy = (float)dyn.lby - fsy, ddz = (float)dyn.lbz - fsz;
        float largest = __builtin_fabsf(ddx);
        if (largest < __builtin_fabsf(ddy)) largest = __builtin_fabsf(ddy);
        if (largest < __builtin_fabsf(ddz)) largest = __builtin_fabsf(ddz);
        const int m = (int)largest;  // alt:432
        const int start_idx = flat_index(g.gy, g.gz, sx, sy, sz);
        if (tid < 3) sm.chain_carry[tid] = (tid == 0) ? fsx : ((tid == 1) ? fsy : fsz);
        const float step_mine = ((tid == 0) ? ddx : ((tid == 1) ? ddy : ddz)) / largest;  // alt:423-425
        __syncthreads();

        for (int it0 = 0; it0 < m; it0 += PAR_CHAIN_ITERS) {
            // the float accumulation of the walk (alt:436-466) is inherently serial: three lanes, one per axis
            if (tid < 3) {
                float v = sm.chain_carry[tid];
                sm.chain[tid][0] = (int)v;  // alt:468: truncation toward zero
#pragma unroll 8
                for (int s = 1; s <= PAR_CHAIN_ITERS; s++) {
                    v = v + step_mine;
                    sm.chain[tid][s] = (int)v;
                }
                // carry = value after PAR_CHAIN_ITERS steps; recompute exactly as the loop did
                sm.chain_carry[tid] = v;
            }
            __syncthreads();

            // 8 lanes per walk iteration, 7 used: the 7 probes of alt:438-466 are the corners of the 2x2x2 block
            // spanned by bin(tmp) and bin(tmp + step), minus bin(tmp).
            const int li = tid >> 3;
            const int mask = (tid & 7) + 1;
            int cnt = 0, idx = 0;
            if (mask < 8 && it0 + li < m) {
                const int ax = sm.chain[0][li], ay = sm.chain[1][li], az = sm.chain[2][li];
                const int nx = sm.chain[0][li + 1], ny = sm.chain[1][li + 1], nz = sm.chain[2][li + 1];
                // a probe whose stepped axes do not all change bin repeats another probe of this iteration (or
                // the previous iteration's last bin): skip it
                const bool canonical = (!(mask & 1) || nx != ax) && (!(mask & 2) || ny != ay) &&
                                       (!(mask & 4) || nz != az);
                if (canonical) {
                    idx = flat_index(g.gy, g.gz, (mask & 1) ? nx : ax, (mask & 2) ? ny : ay, (mask & 4) ? nz : az);
                    // alt:471-473 start bin skipped; out-of-range flat index reads as empty (UB at alt:476)
                    if (idx != start_idx && idx >= 0 && idx < g.volume) cnt = a.count[idx];
                }
            }
            int total;
            const int off = block_excl_scan(cnt, sm.wsum, total);
            for (int base = 0; base < total; base += PAR_MAX_OCC) {
                for (int k = 0; k < cnt; k++) {
                    const int o = off + k - base;
                    if (o >= 0 && o < PAR_MAX_OCC) sm.occ[o] = a.slots[(size_t)idx * PAR_SLOTS + k];
                }
                __syncthreads();
                const int nrec = min(PAR_MAX_OCC, total - base);
#pragma unroll
                for (int k = 0; k < PAR_KPT; k++) {
                    bool live = mine[k] && lit[k];
                    const int ox = (int)(int16_t)col[k], oy = (int)(int16_t)p_y[k], oz = (int)(int16_t)p_z[k];
                    for (int r = 0; r < nrec; r++) {
                        if (!__any(live)) break;  // wavefront early-out: every lane is shadowed or not in the group
                        const par_slot rec = sm.occ[r];
                        if (live && rec.entity != p_entity[k] &&  // alt:484-487
                            slab_hit(rec, ox, oy, oz, inv_x[k], inv_y[k], inv_z[k])) {  // alt:489-491
                            lit[k] = false;
                            live = false;
                        }
                    }
                }
                __syncthreads();
            }
            // stop walking once every pixel of the group is shadowed
            bool still = false;
#pragma unroll
            for (int k = 0; k < PAR_KPT; k++) still |= mine[k] && lit[k];
            if (!__syncthreads_or(still ? 1 : 0)) break;
        }
#pragma unroll
        for (int k = 0; k < PAR_KPT; k++) pend[k] = pend[k] && !mine[k];
        __syncthreads();
    }

    // ---- phase 5: quantise + store, alt:735, 757-758 --------------------------------------------------------
#pragma unroll
    for (int k = 0; k < PAR_KPT; k++) {
        if (!valid[k]) continue;
        const float bright = lit[k] ? b_lit[k] : ambient;
        const size_t o = (size_t)(row[k] - a.row_begin) * W + col[k];
        if (a.out.fb) reinterpret_cast<uint32_t*>(a.out.fb)[o] = color_scale(rgba[k], bright);
        if (a.out.palidx) {
            uint8_t pi = PAR_PALIDX_BACKGROUND;
            if (hit[k]) {
                const int sid = p_tex[k] / PAR_SPRITE_TEXELS;
                pi = (uint8_t)a.sprites[sid].color[p_tex[k] - sid * PAR_SPRITE_TEXELS];
            }
            a.out.palidx[o] = pi;
        }
        if (a.out.brightness) a.out.brightness[o] = bright;
        if (a.out.lit) a.out.lit[o] = lit[k] ? 1 : 0;
        if (a.out.gbuf) {
            par_pixel px;
            px.normal = par_vec3{0.f, 0.f, 0.f};
            if (hit[k]) {
                const int sid = p_tex[k] / PAR_SPRITE_TEXELS;
                px.normal = a.sprites[sid].normal[p_tex[k] - sid * PAR_SPRITE_TEXELS];
            }
            px.color.red = (uint8_t)(rgba[k] & 0xFF);
            px.color.green = (uint8_t)((rgba[k] >> 8) & 0xFF);
            px.color.blue = (uint8_t)((rgba[k] >> 16) & 0xFF);
            px.color.alpha = (uint8_t)(rgba[k] >> 24);
            px.y = p_y[k];
            px.z = p_z[k];
            px.entity_index = p_entity[k];
            a.out.gbuf[o] = px;
        }
    }
}

}  // namespace

hipError_t par_launch_bin_insert(const par_grid_dev& g, const par_bin_args& a, hipStream_t stream) {
    // enough threads for the entities and for wiping the previous frame's nodes (both loops are grid-stride)
    int64_t work = a.n > g.capacity ? a.n : g.capacity;
    int blocks = (int)((work + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bin_insert_kernel, dim3(blocks), dim3(256), 0, stream, g, a);
    return hipGetLastError();
}

hipError_t par_launch_bin_resolve(const par_grid_dev& g, const par_bin_args& a, int64_t pair_bound,
                                  hipStream_t stream) {
    int64_t blocks = (pair_bound + 255) / 256;
    if (blocks < 1) blocks = 1;  // block 0 always runs: it resets the other set's node counter
    hipLaunchKernelGGL(bin_resolve_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, a);
    return hipGetLastError();
}

hipError_t par_launch_render(const par_grid_dev& g, const par_render_args& a, hipStream_t stream) {
    const int by_end = (a.row_end - 1) / a.B;  // last bin row touched
    const int n_bin_rows = by_end - a.by_begin + 1;
    const int64_t blocks = (int64_t)g.gx * n_bin_rows * a.subs;
    if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(PAR_NT), 0, stream, g, a);
    return hipGetLastError();
}
