// par_kernels.hip — hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// One frame of the reference's render call (alt = src/alternative.cpp, spr = src/sprites.hpp) is three launches
// (four for a dense frame), each a latency chain the next depends on:
//   build_fill_kernel       bin_insert_body, a barrier among the build workgroups, bin_resolve_body: memset alt:690 +
//                           count_entities_in_bins alt:195-269, parallel and deterministic (+ which screen columns
//                           show any primitive). Scenes beyond 16 384 entities / 65 536 (entity, bin) pairs take the
//                           same bodies as two launches (insert_fill_kernel, resolve_fill_kernel)
//   columns_fill_kernel     columns_wave: per occupied column its compact slot list and the bin walks of
//                           trace_hash_for_light (alt:399-500; they depend on the start bin only) -> one record,
//                           and the column's work items: one per 64-pixel chunk of an entry-by-entry visit, or one
//                           per tile_k chunks of a whole-tile visit (dense frames). A column's walks are shared by
//                           a team of 1, 2, 4 or 8 wavefronts (the fewer columns a frame has, the more)
//   render_items_kernel     trace_hash_for_pixel alt:271-397, the shading loop alt:702-760, AABB::intersect
//                           alt:40-83, Vector::normalize spr:28-35, Color::operator* spr:8-16 -- one wavefront
//                           per entry-pass work item, from the column records, no workgroup cooperation
//   render_tiles_kernel     the same for the whole-tile items (render_tile_item: scalar-loaded entries and walk
//                           records, lane masks in scalar registers); launched for dense frames only
//   render_overflow_kernel  the columns that overflow a record: straight from the hash, walks in-kernel; launched
//                           only when the host cannot rule an overflow out (PAR_FORCE_GENERIC=1: every column)
//   (render_both_kernel     the render kernels in one launch for small frames, which are bound by their launches and
//                           by their slowest wavefront: a wavefront for every item the fullest shard can hold)
// The background fill (alt:281 -> alt:735, pure streaming) has no launch of its own: the first two launches each
// carry a share of it (extra workgroups running fill_body), sized so that it rides along in their shadow.
// When other planes are asked for (G-buffer, brightness, lit) or the view is not 8-pixel aligned, the hash
// kernels run bare (bin_insert_kernel, bin_resolve_kernel, columns_kernel) and fill_kernel / fill_generic_kernel
// follow; bgline_kernel traces the W distinct background shadow rays when every ray is to be traced.
//
// Float discipline: compiled with -ffp-contract=off; divisions are hipcc's default correctly-rounded fp32
// division; min/max are the ?: forms of std::min/std::max so NaN handling follows the reference (first argument
// wins). With that every float the reference computes is reproduced bit for bit.
#include "par_internal.h"
#include "par_fastdiv.h"
#include "par_strips.h"

#include <limits.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace {

// Debug time stamps (PAR_DEBUG_STAMPS=1): lane 0 of a workgroup notes the 100 MHz wall clock at phase boundaries
// into a buffer of its own; nothing the kernels compute reads it. `k` = kernel row, `i` = stamp slot.
// Rows: 0 insert, 1 resolve, 2 columns (slots 0..4 are its phases), 3 render, 4 overflow; slot 0 = the workgroup's
// start, slot 7 = its end. Only frames rendered with flag bit 29 are stamped.
__device__ __forceinline__ void stamp(const par_grid_dev& g, uint32_t flags, int k, int i) {
    if (g.stamps && (flags & (1u << 29)) && threadIdx.x == 0 && blockIdx.x < PAR_STAMP_WGS) {
        g.stamps[((size_t)k * PAR_STAMP_WGS + blockIdx.x) * PAR_STAMP_SLOTS + i] = __builtin_amdgcn_s_memrealtime();
    }
}

__device__ __forceinline__ int flat_index(int gy, int gz, int x, int y, int z) {
    return x * gy * gz + y * gz + z;  // index_into_view_hash alt:180-182
}

// Device-coherent accesses for data handed from one workgroup to another INSIDE a launch (build_fill_kernel): the
// L2s of the XCDs are not coherent with each other, so such stores write through (sc1) and such loads miss in L2.
// COH = false: plain accesses (the hand-over is a kernel boundary).
template <bool COH, typename T>
__device__ __forceinline__ void st_shared(T* p, T v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool COH, typename T>
__device__ __forceinline__ T ld_shared(const T* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// Wave-uniform data that an EARLIER launch wrote (work items, their counters): read through the scalar cache. The
// compiler only does that by itself for memory it can prove unchanged during the kernel; the constant address
// space says so.
template <typename T>
__device__ __forceinline__ T ld_uniform(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const T __attribute__((address_space(4))) * cptr;
    return *(cptr)(uintptr_t)p;
#else
    return *p;
#endif
}

// A 32-byte work item, fetched with ONE scalar load that is issued here and waited for in item_arrived: the
// compiler would sink an ordinary load below the branch on the item counter, a dependent round trip later.
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x8 item_fetch(const par_item* p) {
    u32x8 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
#else
    v = *reinterpret_cast<const u32x8*>(p);
#endif
    return v;
}
__device__ __forceinline__ void item_arrived(u32x8& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
#endif
}

// Inclusive prefix sum over the 64 lanes of a (whole) wavefront, by data-parallel-primitive adds: within each row of
// 16 lanes by shifts of 1, 2, 4, 8 (zeros shifted in), then the row totals passed on by the two row broadcasts. Six
// vector instructions and no LDS round trip (a shuffle is a ds_bpermute, a hundred cycles each: the six dependent ones
// of the textbook scan were a third of a microsecond per scan of a column's wavefront, which makes four or five).
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
#else
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
#endif
}

// The value of lane `src` (wave-uniform) in every lane: a v_readlane, not a shuffle through the LDS crossbar.
__device__ __forceinline__ int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// Trunc-toward-zero division by the bin size through a precomputed reciprocal: exact for |n| * B < 2^32
// (|n| <= 3 * 32767 here and B <= 320).
__device__ __forceinline__ int div_bin(int n, uint32_t magic) {
    const int q = (int)__umulhi((uint32_t)(n < 0 ? -n : n), magic);
    return n < 0 ? -q : q;
}

// ------------------------------------------------------------------------------------------------------------
// Spatial hash build.
//
// The reference inserts entities one after the other; a bin's counter wraps at 8 (alt:262-264), so after k
// insertions the bin shows c = k & 7 entries and they are exactly the LAST c insertions, in insertion (= entity
// index) order. That closed form is order-independent, so the build is parallel and still deterministic:
//   insert : every (entity, bin) pair pushes a node on the bin's lock-free list (atomicExch on the head);
//   resolve: the thread owning a bin's head walks the list, counts k and keeps the 7 largest entity indices.
// Two head/count/colflag/node sets alternate between frames; `insert` of frame f also wipes what frame f-1
// touched in the other set, so no O(volume) memset is ever issued after context creation.
// ------------------------------------------------------------------------------------------------------------

// ENT entities per wavefront (16 for small scenes so that the pairs of one wavefront fit one or two passes, 64 for
// large ones so that the node counter sees one atomic per 64 entities). Lane l < ENT culls and sizes entity l; the
// wavefront reserves its nodes with ONE atomic; then all 64 lanes walk the (entity, bin) pairs side by side, so the
// list-head exchanges of a wavefront are in flight together instead of one after the other.
template <int ENT, bool COH = false>
__device__ __forceinline__ void bin_insert_body(const par_grid_dev& g, const par_bin_args& a, int block, int n_blocks) {
    const int tid = block * blockDim.x + threadIdx.x;
    const int stride = n_blocks * blockDim.x;
    const int s = a.set, o = a.set ^ 1;

    // wipe what the previous frame touched in the other set
    const int prev = min(g.node_counter[o], g.capacity);
    for (int i = tid; i < prev; i += stride) {
        const int b = g.node_bin[o][i];
        g.head[o][b] = 0;
        g.count[o][b] = 0;
        g.colflag[o][b / g.gz] = 0;
    }
    // (resolve adds to the column counter with atomics: in the one-launch build a plain zero written here could
    // reach memory after them)
    if (tid < PAR_CNT_TOTAL && tid != PAR_CNT_ERROR) st_shared<COH>(&g.counters[tid], 0);  // (the error word is sticky)
    if (tid < PAR_ITEM_LISTS * PAR_ITEM_SHARDS) g.item_counters[tid * PAR_ITEM_COUNTER_STRIDE] = 0;

    const int W = a.W, H = a.H, L = a.L, B = a.B;
    const int lane = threadIdx.x & 63;
    const int n_waves = stride >> 6;
    for (int e0 = (tid >> 6) * ENT; e0 < a.n; e0 += n_waves * ENT) {  // wave-uniform loop
        const int e = e0 + lane;
        int k = 0, org = 0, dim = 0;
        if (lane < ENT && e < a.n) {
            const par_aabb box = a.aabbs[e];
            const int minx = box.px, miny = box.py, minz = box.pz;                       // alt:202-204
            const int maxx = minx + box.ex, maxy = miny + box.ey, maxz = minz + box.ez;  // alt:206-208
            const bool culled = (maxx < 0) || (minx >= W) || (maxy < 0 - maxz) || (miny >= H - minz + B) ||
                                (maxz < -box.ez - B) || (minz > L + B);  // alt:212-219
            if (!culled) {
                // (the reference's divisions truncate towards zero: div_bin, a multiplication by the reciprocal of the
                // bin size; the compiler's sequence for a run-time divisor is some thirty instructions, six times)
                const uint32_t mb = a.magic_b;
                const int x0 = max(0, div_bin(minx, mb));                                       // alt:222
                const int y0 = max(0, div_bin(H - maxy - maxz, mb));                            // alt:223-225
                const int z0 = max(0, div_bin(minz, mb));                                       // alt:226
                const int nx = max(0, min(g.gx, div_bin(maxx + B - 1, mb)) - x0);               // alt:228-230
                const int ny = max(0, min(g.gy, div_bin(H - miny - minz + B - 1, mb)) - y0);    // alt:231-236
                const int nz = max(0, min(g.gz, div_bin(maxz + B - 1, mb)) - z0);               // alt:238-240
                k = nx * ny * nz;
                org = x0 | (y0 << 10) | (z0 << 20);  // grid dimensions are at most 1024 per axis
                dim = ny | (nz << 11);               // box spans are at most 1024 bins per axis
            }
        }
        const int incl = wave_incl_scan_i(k, lane);
        const int excl = incl - k;
        const int total = wave_bcast(incl, 63);
        if (total == 0) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&g.node_counter[s], total);
        base = wave_bcast(base, 0);
        for (int p0 = 0; p0 < total; p0 += 64) {  // wave-uniform: every lane takes part in the shuffles below
            const int p = p0 + lane;
            // owner of pair p: the last lane whose exclusive offset is <= p (lanes with k == 0 share their
            // successor's offset and are skipped by taking the last)
            int lo = 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const int cand = lo + d;
                const int v = __shfl(excl, cand & 63);
                if (cand < 64 && v <= p) lo = cand;
            }
            const int j = p - __shfl(excl, lo);
            const int o3 = __shfl(org, lo), d3 = __shfl(dim, lo);
            if (p < total) {
                const int ny = d3 & 0x7FF, nz = d3 >> 11;
                // j = (jx * ny + jy) * nz + jz. The spans are small (an extent of at most 40 over bins of at least 8: a
                // handful of bins per axis, j below a few hundred): floor(n / d) = (int)((n + 0.5) * rcp(d)) exactly
                // (the quotient is at least 0.5 / d from every integer, the reciprocal is good to an ulp; par_strips.h)
                const int t = (int)(((float)j + 0.5f) * __builtin_amdgcn_rcpf((float)nz));
                const int jz = j - t * nz;
                const int jx = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ny));
                const int jy = t - jx * ny;
                const int b = flat_index(g.gy, g.gz, (o3 & 0x3FF) + jx, ((o3 >> 10) & 0x3FF) + jy, (o3 >> 20) + jz);
                const int node = base + p;
                // the host sizes the pool from the exact pair count and b is in range by construction: belt and braces
                if (node < g.capacity && b >= 0 && b < g.volume) {
                    st_shared<COH>(&g.node_entity[s][node], e0 + lo);
                    st_shared<COH>(&g.node_bin[s][node], b);
                    st_shared<COH>(&g.node_next[s][node], atomicExch(&g.head[s][b], node + 1));
                }
            }
        }
    }
}

template <int ENT>
__global__ __launch_bounds__(256) void bin_insert_kernel(par_grid_dev g, par_bin_args a) {
    stamp(g, a.flags, 0, 0);
    bin_insert_body<ENT>(g, a, (int)blockIdx.x, (int)gridDim.x);
    stamp(g, a.flags, 0, 7);
}

// `tid` of `n_threads` (one thread per node, or a grid-stride loop over the nodes when the launch is smaller).
template <bool COH>
__device__ __forceinline__ void bin_resolve_node(const par_grid_dev& g, const par_bin_args& a, int tid);

template <bool COH = false>
__device__ __forceinline__ void bin_resolve_body(const par_grid_dev& g, const par_bin_args& a, int block, int n_blocks) {
    const int tid0 = block * blockDim.x + threadIdx.x;
    const int s = a.set;
    // insert has consumed the other set's counter: free it for the next frame's inserts
    if (tid0 == 0) g.node_counter[s ^ 1] = 0;
    const int n_nodes = min(ld_shared<COH>(&g.node_counter[s]), g.capacity);
    for (int tid = tid0; tid < n_nodes; tid += n_blocks * blockDim.x) bin_resolve_node<COH>(g, a, tid);
}

template <bool COH>
__device__ __forceinline__ void bin_resolve_node(const par_grid_dev& g, const par_bin_args& a, int tid) {
    const int s = a.set;
    const int b = ld_shared<COH>(&g.node_bin[s][tid]);
    if (ld_shared<COH>(&g.head[s][b]) != tid + 1) return;  // only the most recent insertion resolves its bin

    int top0 = -1, top1 = -1, top2 = -1, top3 = -1, top4 = -1, top5 = -1, top6 = -1;  // 7 largest, descending
    int k = 0;
    for (int cur = tid + 1; cur != 0; cur = ld_shared<COH>(&g.node_next[s][cur - 1])) {
        int e = ld_shared<COH>(&g.node_entity[s][cur - 1]);
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
    if (c > 0) {
        // the screen column (bx, by) of this bin shows something: flag it, and list it once if the render of
        // this frame covers its rows
        const int col = b / g.gz;
        const int by = col % g.gy;
        if (atomicExch(&g.colflag[s][col], 1) == 0 && by >= a.by_lo && by <= a.by_hi) {
            // one atomic per wavefront on the shared counter (the lanes that got here take consecutive slots)
            const unsigned long long m = __ballot(1);
            const int leader = __ffsll((long long)m) - 1;
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == leader) base = atomicAdd(&g.counters[PAR_CNT_COLS], __popcll(m));
            base = __shfl(base, leader);
            g.col_list[base + __popcll(m & ((1ull << lane) - 1ull))] = col;
        }
    }
}

__global__ __launch_bounds__(256) void bin_resolve_kernel(par_grid_dev g, par_bin_args a) {
    stamp(g, a.flags, 1, 0);
    bin_resolve_body(g, a, (int)blockIdx.x, (int)gridDim.x);
    stamp(g, a.flags, 1, 7);
}

// ------------------------------------------------------------------------------------------------------------
// Small device helpers
// ------------------------------------------------------------------------------------------------------------

// std::min / std::max on floats: (b<a)?b:a and (a<b)?b:a -- the first argument survives a NaN (SURVEY a-5).
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

// AABB::intersect, alt:40-83.
// FINITE: the caller knows that ix, iy, iz are finite. The only NaN the test can meet is 0 x inf (an integer
// difference times an infinite inverse), so none arises then, and without NaNs the reference's compare-selects are
// the mathematical min / max up to the sign of a zero, which the final comparison does not see: the hardware's
// min / max / min3 / max3 give the same verdict in a third of the instructions.
template <bool FINITE = false>
__device__ __forceinline__ bool slab_hit(const par_slot& r, int ox, int oy, int oz, float ix, float iy, float iz) {
    const float x1 = (float)(r.px - ox) * ix;
    const float x2 = (float)(r.px + r.ex - ox) * ix;
    const float y1 = (float)(r.py - oy) * iy;
    const float y2 = (float)(r.py + r.ey - oy) * iy;
    const float z1 = (float)(r.pz - oz) * iz;
    const float z2 = (float)(r.pz + r.ez - oz) * iz;
    if (FINITE) {
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2)), __builtin_fminf(z1, z2));
        const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2)), __builtin_fmaxf(z1, z2));
        return tmax >= tmin;
    }
    float tmin = std_min(x1, x2);
    float tmax = std_max(x1, x2);
    tmin = std_max(tmin, std_min(y1, y2));
    tmax = std_min(tmax, std_max(y1, y2));
    tmin = std_max(tmin, std_min(z1, z2));
    tmax = std_min(tmax, std_max(z1, z2));
    return tmax >= tmin;
}

// The same test on a walk record (par_walkrec: the planes as floats, lo / hi pairs): the differences to the origin
// are exact in both forms, so every product is the one above; two values per packed instruction.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ par_walkrec walkrec_of(const par_slot& r) {
    par_walkrec w;
    w.x_lo = (float)r.px; w.x_hi = (float)(r.px + r.ex);
    w.y_lo = (float)r.py; w.y_hi = (float)(r.py + r.ey);
    w.z_lo = (float)r.pz; w.z_hi = (float)(r.pz + r.ez);
    w.entity = r.entity;
    w.pad_ = 0;
    return w;
}
template <bool FINITE>
__device__ __forceinline__ bool slab_hit_rec(const v2f& rx, const v2f& ry, const v2f& rz, float fox, float foy,
                                             float foz, float ix, float iy, float iz) {
    const v2f tx = (rx - fox) * ix, ty = (ry - foy) * iy, tz = (rz - foz) * iz;
    if (FINITE) {
        const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fminf(tz.x, tz.y));
        const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fmaxf(tz.x, tz.y));
        return tmax >= tmin;
    }
    float tmin = std_min(tx.x, tx.y);
    float tmax = std_max(tx.x, tx.y);
    tmin = std_max(tmin, std_min(ty.x, ty.y));
    tmax = std_min(tmax, std_max(ty.x, ty.y));
    tmin = std_max(tmin, std_min(tz.x, tz.y));
    tmax = std_min(tmax, std_max(tz.x, tz.y));
    return tmax >= tmin;
}

// Color::operator*, spr:8-16: truncating per-channel scale, alpha passed through. `c` is RGBA little-endian.
__device__ __forceinline__ uint32_t color_scale(uint32_t c, float v) {
    const uint32_t r = (uint32_t)(uint8_t)((float)(c & 0xFF) * v);
    const uint32_t g = (uint32_t)(uint8_t)((float)((c >> 8) & 0xFF) * v);
    const uint32_t b = (uint32_t)(uint8_t)((float)((c >> 16) & 0xFF) * v);
    return r | (g << 8) | (b << 16) | (c & 0xFF000000u);
}

// Vector<float>::normalize, spr:28-35 — divides by the L1 length (abs(x) + abs(y)) + abs(z) — and the inverse
// direction 1 / n (alt:717-719), through the short sequences of par_fastdiv.h when every
// lane's operands are in the range they are exact on (components that are integers of magnitude <= 65535 — checked:
// the render kernels' are differences of integers, the test hook's may be anything — and a length >= 1), through the
// ordinary divisions otherwise (a light on
// the pixel: 0 / 0; sprite depths that throw a pixel far out). Wave-uniform choice: one branch, no divergence.
// ANY_INPUT: the components may be any floats (the test hook); false: the caller passes converted integers.
template <bool ANY_INPUT = false>
__device__ __forceinline__ void normalize_l1_and_inverse(float x, float y, float z, float& nx, float& ny, float& nz,
                                                         float& ix, float& iy, float& iz) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), az = __builtin_fabsf(z);
    const float len = ax + ay + az;
    // (tools/divcheck.hip proves the short sequences for INTEGER numerators: the render kernels' are differences of
    // integers; the test hook par_debug_units kind 2 takes any floats, which then go through the ordinary divisions)
    const bool integral = !ANY_INPUT || (x == __builtin_truncf(x) && y == __builtin_truncf(y) && z == __builtin_truncf(z));
    const bool in_range = integral && ax <= PAR_FASTDIV_MAX_NUM && ay <= PAR_FASTDIV_MAX_NUM && az <= PAR_FASTDIV_MAX_NUM && len >= 1.0f;
    if (__all(in_range)) {
        const float r = __builtin_amdgcn_rcpf(len);
        nx = par_fast_div(x, len, r);
        ny = par_fast_div(y, len, r);
        nz = par_fast_div(z, len, r);
        ix = par_fast_rcp(nx);
        iy = par_fast_rcp(ny);
        iz = par_fast_rcp(nz);
    } else {
        nx = x / len;
        ny = y / len;
        nz = z / len;
        ix = 1.f / nx;
        iy = 1.f / ny;
        iz = 1.f / nz;
    }
}

// ------------------------------------------------------------------------------------------------------------
// wave_walk: ONE wavefront walks from bin (sx, sy, sz) to the light's bin as trace_hash_for_light does
// (alt:399-500) and stages the slot records of every occupied bin on the way (start bin excluded, alt:471-473) in
// `stage` (LDS, PAR_BIN_WALK records). Returns their number, or -1 when they do not fit. The probed bin sequence
// depends only on the two bins, not on a ray; the reference's result is an OR over the probes, so neither probe order
// nor duplicates matter.
// It runs in rounds of 64 iterations of the reference's loop, lane k taking iteration it0 + k:
//   - the float accumulation cur += step (alt:436-466) is inherently serial, but every lane can run it: lane k needs
//     the position after k additions, so all lanes start from the round's first position and add the step under an
//     EXEC mask that loses its lowest lane at every step (s_lshl_b64 exec, exec, 1): lane k takes part in exactly k
//     additions. Three register-only additions and one scalar shift per step, the same single-precision sums in the
//     same order as the reference's, no memory. (Until round 3 one lane per axis added and left every position in
//     LDS for the others: 3 us per walk, bound by the LDS store path of a CU whose resident wavefronts all did the
//     same at the same time.)
//   - the 7 probes of an iteration (alt:438-466) are the corners of the 2x2x2 block spanned by bin(tmp) and
//     bin(tmp + step), minus bin(tmp) itself; their counts are loaded side by side, the records of the occupied
//     ones appended to the stage.
// ------------------------------------------------------------------------------------------------------------

// 16 steps of the accumulation for a whole (64-lane) wavefront: a lane of `lanes` takes part in the first step, and
// every step the lowest lane still taking part drops out.
__device__ __forceinline__ void walk_chain_block(float& vx, float& vy, float& vz, float sx, float sy, float sz,
                                                 uint64_t lanes) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (x and y in one packed instruction: v_pk_add_f32 is two IEEE single-precision additions, each rounded as
    // v_add_f32 rounds it -- two vector instructions per step instead of three)
    uint64_t saved;
    v2f vxy = {vx, vy};
    const v2f sxy = {sx, sy};
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "s_mov_b64 exec, %[m0]\n"
        ".rept 16\n"
        "v_pk_add_f32 %[xy], %[xy], %[sxy]\n"
        "v_add_f32 %[z], %[z], %[sz]\n"
        "s_lshl_b64 exec, exec, 1\n"
        ".endr\n"
        "s_mov_b64 exec, %[sv]\n"
        : [xy] "+v"(vxy), [z] "+v"(vz), [sv] "=&s"(saved)
        : [sxy] "v"(sxy), [sz] "v"(sz), [m0] "s"(lanes)
        : "scc");
    vx = vxy.x;
    vy = vxy.y;
#else
    (void)vx; (void)vy; (void)vz; (void)sx; (void)sy; (void)sz; (void)lanes;
#endif
}

__device__ int wave_walk(const par_grid_dev& g, const uint8_t* count, const par_slot* slots, const par_frame_dyn& dyn,
                         int sx, int sy, int sz, par_slot* stage, uint32_t sflags = 0) {
    const int lane = threadIdx.x & 63;
    const int b0 = flat_index(g.gy, g.gz, sx, sy, sz);  // alt:430
    // alt:406-430
    const float fsx = (float)sx, fsy = (float)sy, fsz = (float)sz;
    const float ddx = (float)dyn.lbx - fsx, ddy = (float)dyn.lby - fsy, ddz = (float)dyn.lbz - fsz;
    float largest = __builtin_fabsf(ddx);
    if (largest < __builtin_fabsf(ddy)) largest = __builtin_fabsf(ddy);
    if (largest < __builtin_fabsf(ddz)) largest = __builtin_fabsf(ddz);
    const int m = (int)largest;  // alt:432
    const float stx = ddx / largest, sty = ddy / largest, stz = ddz / largest;  // alt:423-425
    float cx = fsx, cy = fsy, cz = fsz;  // the position after it0 iterations (wave-uniform)
    int n_rec = 0;
    for (int it0 = 0; it0 < m; it0 += 64) {
        const int n_it = min(64, m - it0);
        // lane k: the position after it0 + k iterations (alt:436-466: k additions, one after the other)
        float vx = cx, vy = cy, vz = cz;
        for (int t0 = 1; t0 < n_it; t0 += 16) walk_chain_block(vx, vy, vz, stx, sty, stz, ~0ull << t0);
        // ... and after one more: the next lane's position (the same sum), and the next round's start in lane 63
        const float wx = vx + stx, wy = vy + sty, wz = vz + stz;
        cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), 63));
        cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), 63));
        cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), 63));
        stamp(g, sflags, 2, 5);

        // lane l takes walk iteration it0 + l: its 7 probes (alt:438-466) are the corners of the 2x2x2 block
        // spanned by bin(tmp) and bin(tmp + step), minus bin(tmp) itself (alt:468: truncation; bins fit 16 bits)
        int idx[7], cnt[7];
        int mine = 0;
        {
            const int ax = (int)(int16_t)(int)vx, ay = (int)(int16_t)(int)vy, az = (int)(int16_t)(int)vz;
            const int qx = (int)(int16_t)(int)wx, qy = (int)(int16_t)(int)wy, qz = (int)(int16_t)(int)wz;
            bool ok[7];
#pragma unroll
            for (int q = 0; q < 7; q++) {
                const int mask = q + 1;
                // a probe whose stepped axes do not all change bin repeats another probe
                const bool canonical = (!(mask & 1) || qx != ax) && (!(mask & 2) || qy != ay) &&
                                       (!(mask & 4) || qz != az);
                const int b = flat_index(g.gy, g.gz, (mask & 1) ? qx : ax, (mask & 2) ? qy : ay,
                                         (mask & 4) ? qz : az);
                // alt:471-473: the start bin is skipped; an out-of-range flat index reads as an empty bin (alt:476)
                ok[q] = lane < n_it && canonical && b != b0 && b >= 0 && b < g.volume;
                idx[q] = ok[q] ? b : 0;
            }
            // the seven counts are loaded side by side (no branch between the loads)
#pragma unroll
            for (int q = 0; q < 7; q++) cnt[q] = count[idx[q]];
#pragma unroll
            for (int q = 0; q < 7; q++) {
                cnt[q] = ok[q] ? cnt[q] : 0;
                mine += cnt[q];
            }
        }
        const int incl = wave_incl_scan_i(mine, lane);
        const int wave_total = wave_bcast(incl, 63);
        stamp(g, sflags, 2, 6);
        if (n_rec + wave_total > PAR_BIN_WALK) return -1;
        int o = n_rec + incl - mine;
#pragma unroll
        for (int q = 0; q < 7; q++) {
            for (int k = 0; k < cnt[q]; k++) stage[o++] = slots[(size_t)idx[q] * PAR_SLOTS + k];
        }
        n_rec += wave_total;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n_rec;
}

// LDS written and read by the same wavefront: its LDS operations complete in order; keep the compiler from moving
// reads above writes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------------------
// columns_wave (columns_kernel, columns_fill_kernel): one WAVEFRONT per occupied screen column (bx, by); a workgroup
// is PAR_COL_WAVES independent wavefronts (no workgroup barrier anywhere: a column with one occupied bin does not
// keep a second wavefront waiting for it).
//   A. the column's bins, front to back: compact list of the occupied ones and their slot records;
//   -  how the pixels are to be visited (entry rectangles or the whole tile), which entries repeat an earlier
//      entity; the column's work items (one per 64-pixel chunk of the visit) are reserved in the render list;
//   B. from every occupied bin the walk to the light as trace_hash_for_light does it (alt:399-500). The probed bin
//      sequence depends only on the start and light bins, not on the ray, so it is done ONCE per bin and frame: the
//      slot records of every occupied bin on the way (start bin excluded, alt:471-473) are kept. A pixel whose
//      shadow ray starts in that bin only slab-tests the short list; the reference's result is an OR over probes,
//      so neither probe order nor duplicates matter;
//   C. the work items and the record go to HBM/L2 for the render kernel.
// A column that does not fit the record (PAR_COL_*) goes onto the overflow list (render_overflow_kernel).
// ------------------------------------------------------------------------------------------------------------

struct ColWave {  // LDS of one wavefront
    par_colrec_nb nb[PAR_COL_NB];
    par_slot entries[PAR_COL_ENT];
    int16_t ebz[PAR_COL_ENT];
    par_slot stage[PAR_BIN_WALK];  // the records of the walk being done
};

// The box of chunk `c` of a tile visit (strip order, par_strips.h), in tile coordinates: the rows and columns of its
// first and last pixel when both lie in one strip, the whole rectangle otherwise. `row_off`: the visit's first row
// relative to the tile's.
__device__ __forceinline__ void tile_chunk_box(const par_strips& st, int tw, int rh, int row_off, int c, int& r0, int& r1,
                                               int& q0, int& q1) {
    const int area = tw * rh;
    const int pf = min(c * 64, area - 1), pl = min(c * 64 + 63, area - 1);
    int s0, c0, y0, s1, c1, y1;
    par_strip_pixel(st, rh, pf, s0, c0, y0);
    par_strip_pixel(st, rh, pl, s1, c1, y1);
    r0 = row_off; r1 = row_off + rh - 1; q0 = 0; q1 = tw - 1;
    if (s0 == s1) {
        r0 = row_off + y0;
        r1 = row_off + y1;
        q0 = s0 * st.sw;
        q1 = q0 + (s0 == st.n_strips - 1 ? st.lw : st.sw) - 1;
    }
}

// columns_wave, whole-tile columns of at most PAR_TILE_MASKS chunks: which entries can cover a pixel of each chunk
// (par_colrec::cmask). Lane c works out chunk c's box; then, chunk by chunk, every lane holds its entry's rectangle
// against that box (one ballot per chunk). The render kernel reads a chunk's mask with one scalar load.
__device__ __forceinline__ void tile_candidate_masks(const par_render_args& a, const ColWave& sm, par_colrec* rec,
                                                     int n_entries, uint64_t dup_mask, int bx, int by, int lane) {
    const int c0 = bx * a.B, ty = by * a.B;
    const int tw = min(a.B, a.W - c0);
    const int rows_lo = max(ty, a.row_begin), rows_hi = min(min(ty + a.B, a.H), a.row_end);
    const int rh = rows_hi - rows_lo;
    if (tw <= 0 || rh <= 0) return;
    const int tile_chunks = (tw * rh + 63) >> 6;
    const par_strips st = par_strips_of(tw);
    int b_r0, b_r1, b_q0, b_q1;
    tile_chunk_box(st, tw, rh, rows_lo - ty, min(lane, tile_chunks - 1), b_r0, b_r1, b_q0, b_q1);
    int e_r0 = 0, e_r1 = 0, e_q0 = 0, e_q1 = 0;
    if (lane < n_entries && !((dup_mask >> lane) & 1)) {
        const par_slot r = sm.entries[lane];
        const int top = r.py + r.ey + r.pz + r.ez;
        e_r0 = min(max(a.H - top - ty, 0), a.B); e_r1 = min(max(a.H - (r.py + r.pz) - ty, 0), a.B);
        e_q0 = min(max(r.px - c0, 0), a.B);      e_q1 = min(max(r.px + r.ex - c0, 0), a.B);
    }
    const bool eligible = e_r1 > e_r0 && e_q1 > e_q0;
    uint64_t mine = 0;
    for (int c = 0; c < tile_chunks; c++) {
        const int r0 = __builtin_amdgcn_readlane(b_r0, c), r1 = __builtin_amdgcn_readlane(b_r1, c);
        const int q0 = __builtin_amdgcn_readlane(b_q0, c), q1 = __builtin_amdgcn_readlane(b_q1, c);
        const uint64_t m = __ballot(eligible && e_r0 <= r1 && e_r1 > r0 && e_q0 <= q1 && e_q1 > q0);
        if (lane == c) mine = m;
    }
    if (lane < tile_chunks) rec->cmask[lane] = mine;
}

// `ci`: index into the column list, or (when background rays are traced) n_cols_bound + bx for the walk from the
// background start bin of bin column bx. `role`: PAR_COL_ROLES wavefronts share a column's walks (role r takes the
// occupied bins r, r + roles, ... and the r-th part of the record's walk area); role 0 does everything else. The
// others repeat the scan of the column's counts (they need the list of occupied bins) and leave at once when the
// column has no walk for them.
// What the wavefronts of one column share (ROLES > 1: the workgroup is one column's team).
struct ColTeam {
    int32_t found[8];   // per role: did any of its walks meet an occupied bin (or fail to fit)?
    int32_t walk_used;  // records of the column's walk area handed out so far
};
constexpr int PAR_COL_MAX_ROLES = 8;

template <int ROLES>
__device__ __forceinline__ void columns_wave(const par_grid_dev& g, const par_render_args& a, ColWave& sm,
                                             ColTeam* team_sh, int ci, int role, int n_cols_bound) {
    static_assert(ROLES >= 1 && ROLES <= PAR_COL_MAX_ROLES, "ColTeam::found has one word per role");
    const int lane = (int)threadIdx.x & 63;
    if (ci >= n_cols_bound) {
        const int bx = ci - n_cols_bound;
        if (!a.trace_bg || bx >= g.gx || role != 0) return;
        const par_frame_dyn dyn = a.dyn_ptr ? *a.dyn_ptr : a.dyn;
        // world (x, 0, 0): ray_bin = (x / B, (H - 0 - 0) / B, 0), alt:724-727
        const int n_rec = wave_walk(g, a.count, a.slots, dyn, bx, a.H / a.B, 0, sm.stage);
        par_bgwalk* out = g.bgwalk + bx;
        for (int r = lane; r < n_rec; r += 64) out->rec[r] = sm.stage[r];
        if (lane == 0) out->cnt = n_rec;
        return;
    }
    if (role == 0) stamp(g, a.flags, 2, 0);
    // the launch is sized by an upper bound of the occupied columns; both loads are issued together
    const int n_cols = g.counters[PAR_CNT_COLS];
    const int col = g.col_list[ci];
    if (ci >= n_cols) return;
    if (role == 0) stamp(g, a.flags, 2, 1);
    const int bx = col / g.gy, by = col - (col / g.gy) * g.gy;
    const int col_base = flat_index(g.gy, g.gz, bx, by, 0);
    bool overflow = ci >= g.col_capacity;

    // ---- A: ordered compaction of the column, 64 bins at a time (the next 64 counts are fetched meanwhile) -------
    int n_nb = 0, n_entries = 0;
    int c_next = (lane < g.gz) ? (int)a.count[col_base + lane] : 0;
    for (int t0 = 0; t0 < g.gz; t0 += 64) {
        const int t = t0 + lane;
        const int c = c_next;
        c_next = (t + 64 < g.gz) ? (int)a.count[col_base + t + 64] : 0;
        const int incl = wave_incl_scan_i(((c != 0) << 16) | c, lane);
        const int total = wave_bcast(incl, 63);
        const int excl = incl - (((c != 0) << 16) | c);
        const int nb_i = n_nb + (excl >> 16);
        const int off = n_entries + (excl & 0xFFFF);
        bool over = false;
        if (c != 0) {
            if (nb_i < PAR_COL_NB && off + c <= PAR_COL_ENT) {
                const par_slot* src = a.slots + (size_t)(col_base + t) * PAR_SLOTS;
                par_colrec_nb e;
                e.bz = (int16_t)t; e.off = (uint8_t)off; e.cnt = (uint8_t)c; e.woff = 0; e.wcnt = 0;
                sm.nb[nb_i] = e;
                if (role == 0) {
                    for (int k = 0; k < c; k++) {
                        sm.entries[off + k] = src[k];
                        sm.ebz[off + k] = (int16_t)t;
                    }
                }
            } else {
                over = true;
            }
        }
        overflow = overflow || __any(over);
        n_nb += total >> 16;
        n_entries += total & 0xFFFF;
    }
    wave_lds_fence();
    if (role == 0) stamp(g, a.flags, 2, 2);
    // The column's wavefronts work as a team exactly when it has walks to share (every one of them has scanned the same
    // counts and knows): then all of them stay until the barrier behind the walks, with or without a walk of their own.
    const bool team = ROLES > 1 && !overflow && n_nb >= 2;
    if (role != 0 && !team) return;  // no walk for this wavefront

    // ---- how the render kernel should visit the column's pixels: entry rectangle by entry rectangle when they
    // cover little of it (the lanes of a 64-pixel chunk are then nearly all covered pixels), otherwise the whole
    // tile. An entry that repeats an earlier entry's entity (the same AABB in another bin of the column) has the
    // same rectangle and owns no pixel: it is marked and skipped. Every 64-pixel chunk of the visit becomes one
    // work item of the render launch (at most PAR_COL_ENT <= 64 entries: one per lane).
    // The column's share of its shard of the item list is reserved here, BEFORE the walks, with one atomic on one of
    // PAR_ITEM_SHARDS words that lie a cache line apart: its result is needed only after the walks, which hide its
    // latency.
    const int c0 = bx * a.B, tw = min(a.B, a.W - c0);
    const int rows_lo = max(by * a.B, a.row_begin), rows_hi = min(min((by + 1) * a.B, a.H), a.row_end);
    const int tile_chunks = (tw * max(rows_hi - rows_lo, 0) + 63) >> 6;
    const int shard = ci & (PAR_ITEM_SHARDS - 1);
    int my_chunks = 0;
    bool dup = false;
    if (role == 0 && !overflow) {
        // an entry that repeats an earlier entry's entity: entry by entry through a scalar register (no LDS round
        // trips: this sits on every column's chain of dependent steps)
        const int mine = lane < n_entries ? sm.entries[min(lane, PAR_COL_ENT - 1)].entity : -1;
        for (int e = 0; e + 1 < n_entries; e++) dup = dup || (lane > e && lane < n_entries && mine == wave_bcast(mine, e));
    }
    if (role == 0 && !overflow && lane < n_entries) {
        const par_slot r = sm.entries[lane];
        const int w = min(r.px + r.ex, c0 + tw) - max((int)r.px, c0);
        const int h = min(a.H - (r.py + r.pz), rows_hi) - max(a.H - (r.py + r.ey + r.pz + r.ez), rows_lo);
        if (!dup && w > 0 && h > 0) my_chunks = (w * h + 63) >> 6;  // every visit costs whole wavefronts
    }
    const uint64_t dup_mask = __ballot(dup);
    const int chunk_incl = wave_incl_scan_i(my_chunks, lane);
    const int pass_chunks = wave_bcast(chunk_incl, 63);
    const int first_item = chunk_incl - my_chunks;
    // (a.tile_k == 0: a sparse frame, which has no launch for tile items: every column is visited entry by entry)
    const int tile_mode = (a.tile_k > 0 && pass_chunks >= tile_chunks) ? 1 : 0;
    // a whole-tile visit is listed as items of tile_k consecutive chunks each (what a column's chunks share is then
    // read once per item, render_tile_item), in the list of the tile kernel; an entry-by-entry visit as one item per
    // chunk in the list of the entry kernel
    const int tile_k = max(a.tile_k, 1);
    const int tile_items = (int)(((uint32_t)(tile_chunks + tile_k - 1) * a.tile_k_magic) >> 16);
    const int n_items = (overflow || role != 0) ? 0 : (tile_mode ? tile_items : pass_chunks);
    const int list_shard = tile_mode * PAR_ITEM_SHARDS + shard;
    int item_base = 0;
    if (lane == 0 && n_items > 0) item_base = atomicAdd(&g.item_counters[list_shard * PAR_ITEM_COUNTER_STRIDE], n_items);

    // ---- B: the shadow walks of this wavefront's share of the occupied bins, one after the other ---------------
    // The walk area of the record is handed out walk by walk: a team reserves from a counter in LDS (its wavefronts'
    // walks differ in length: fixed shares would turn away walks that fit), a lone wavefront counts for itself.
    int n_walk = 0;
    bool walk_failed = false;
    if (!overflow && !(a.flags & (1u << 27))) {  // bit 27: ablation (timing experiments only), no walks
        const par_frame_dyn dyn = a.dyn_ptr ? *a.dyn_ptr : a.dyn;
        for (int i = role; i < n_nb; i += ROLES) {
            const int sz = sm.nb[i].bz;
            const int n_rec = wave_walk(g, a.count, a.slots, dyn, bx, by, sz, sm.stage, i == 0 ? a.flags : 0u);
            // (a list takes an even number of records: the tile pass reads them in pairs, walk_list_lit)
            const int need = n_rec + (n_rec & 1);
            int at = n_walk;
            if (ROLES > 1 && team && n_rec > 0) {
                if (lane == 0) at = atomicAdd(&team_sh->walk_used, need);
                at = wave_bcast(at, 0);
            }
            if (n_rec < 0 || at + need > PAR_COL_WALK) {
                // more occluders on the way than the record holds: the pixels that start here walk for themselves
                // (lane_shadow_walk in the render kernels), the column keeps its record
                walk_failed = true;
                if (lane == 0) {
                    sm.nb[i].woff = 0;
                    sm.nb[i].wcnt = -1;
                }
            } else {
                par_walkrec* dst = g.colrec[ci].walk + at;
                for (int r = lane; r < n_rec; r += 64) dst[r] = walkrec_of(sm.stage[r]);
                // a list of odd length repeats its last record behind its end (the result is an OR over the records)
                if ((n_rec & 1) && lane == 0) dst[n_rec] = walkrec_of(sm.stage[n_rec - 1]);
                if (lane == 0) {
                    sm.nb[i].woff = (int16_t)at;
                    sm.nb[i].wcnt = (int16_t)n_rec;
                }
                n_walk += need;
            }
            wave_lds_fence();  // (the next walk overwrites the stage)
        }
    }
    if (role == 0) stamp(g, a.flags, 2, 3);
    // Did any walk of the column meet an occupied bin (or fail to fit)? The wavefronts of a column are its whole
    // workgroup, and all of them are still here exactly when the column has a walk for each (the same test in all
    // of them): then, and only then, they meet at a barrier.
    bool walks_empty = n_walk == 0 && !walk_failed;
    if (ROLES > 1 && team) {
        if (lane == 0) team_sh->found[role] = walks_empty ? 0 : 1;
        __syncthreads();
        if (role == 0) {
            int any = 0;
#pragma unroll
            for (int r = 0; r < ROLES; r++) any |= team_sh->found[r];
            walks_empty = any == 0;
        }
    }
    if (role != 0) {  // the other wavefronts' part of the record: the bins they walked from
        if (ci < g.col_capacity && lane < n_nb && lane % ROLES == role) g.colrec[ci].nb[lane] = sm.nb[lane];
        return;
    }

    // ---- C: the column's work items (none of them usable when the shard is full: the host sizes a shard for every
    // item of the frame, so that is belt and braces) and the record -------------------------------------------
    if (n_items > 0) {
        item_base = wave_bcast(item_base, 0);
        const bool usable = item_base + n_items <= g.item_capacity;
        // A SIMPLE column: every entry is the same entity (one distinct entry), its occupied bins are contiguous and
        // no walk from them met anything. Its items carry all the render kernel needs.
        const int distinct = n_entries - __popcll(dup_mask);
        const int bz_first = sm.nb[0].bz, bz_last = sm.nb[max(n_nb, 1) - 1].bz;
        const bool simple = !(a.flags & (1u << 22)) && distinct == 1 && walks_empty && !tile_mode && n_nb >= 1 &&
                            bz_last - bz_first == n_nb - 1;  // bit 22 (tests): no simple items
        par_item it;
        it.ci = usable ? (uint32_t)ci : PAR_ITEM_NONE;
        it.where = (uint32_t)bx | ((uint32_t)by << 10) | (simple ? PAR_ITEM_SIMPLE : 0u);
        it.bins = (uint32_t)bz_first | ((uint32_t)bz_last << 16);
        par_item* dst = g.items + (size_t)list_shard * g.item_capacity + item_base;
        if (tile_mode) {
            it.entry = par_slot{0, 0, 0, 0, 0, 0, 0};
            for (int k = lane; k < n_items; k += 64) {
                it.visit = (PAR_ITEM_TILE << 16) | (uint32_t)(k * tile_k);
                it.bins = (uint32_t)min(tile_k, tile_chunks - k * tile_k);
                if (item_base + k < g.item_capacity) dst[k] = it;
            }
        } else {
            if (my_chunks > 0) it.entry = sm.entries[lane];
            for (int k = 0; k < my_chunks; k++) {
                it.visit = ((uint32_t)lane << 16) | (uint32_t)k;
                if (item_base + first_item + k < g.item_capacity) dst[first_item + k] = it;
            }
        }
        overflow = overflow || !usable;
    }
    if (ci < g.col_capacity) {
        par_colrec* rec = g.colrec + ci;
        if (lane == 0) {
            rec->n_nb = (int16_t)(overflow ? 0 : n_nb);
            rec->n_entries = (int16_t)(overflow ? 0 : n_entries);
            rec->n_walk = (int16_t)n_walk;  // (of this wavefront's share)
            rec->overflow = overflow ? 1 : 0;
            rec->bx = (int16_t)bx;
            rec->by = (int16_t)by;
            rec->tile_mode = tile_mode;
            rec->chunks = n_items;
            rec->dup_lo = (uint32_t)dup_mask;
            rec->dup_hi = (uint32_t)(dup_mask >> 32);
        }
        if (!overflow) {
            if (lane < n_nb && lane % ROLES == 0) rec->nb[lane] = sm.nb[lane];
            if (lane < n_entries) {
                rec->entries[lane] = sm.entries[lane];
                rec->ebz[lane] = sm.ebz[lane];
            }
            if (tile_mode) {
                // the tile pass's view of the entries (par_xent, rect): every sum and difference the per-pixel test
                // needs, and how many EMPTY stretches of the column lie before the entry's bin (an empty bin
                // between two visited bins resets `adjacent`, alt:298-300; occupied bins in between do not)
                const int bz = lane < n_entries ? (int)sm.ebz[lane] : 0;
                const int prev_bz = __shfl_up(bz, 1);
                const uint64_t gap_mask = __ballot(lane > 0 && lane < n_entries && bz != prev_bz && bz != prev_bz + 1);
                if (lane < n_entries) {
                    const par_slot r = sm.entries[lane];
                    const int top = r.py + r.ey + r.pz + r.ez;
                    par_xent x;
                    x.px4 = 4 * r.px;
                    x.dims = (4 * r.ex) | ((r.ey + r.ez) << 8);
                    x.top = top;
                    x.k = r.ey - top;
                    x.dbase = r.py - r.pz;
                    x.pz = r.pz;
                    x.entity = r.entity;
                    x.bzk = bz | (__popcll(gap_mask & ((2ull << lane) - 1ull)) << 16);
                    rec->xent[lane] = x;
                    const int ty = by * a.B;
                    const int r0 = min(max(a.H - top - ty, 0), a.B), r1 = min(max(a.H - (r.py + r.pz) - ty, 0), a.B);
                    const int q0 = min(max(r.px - c0, 0), a.B), q1 = min(max(r.px + r.ex - c0, 0), a.B);
                    rec->rect[lane] = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)q0 << 16) | ((uint32_t)q1 << 24);
                }
                if (tile_chunks <= PAR_TILE_MASKS) tile_candidate_masks(a, sm, rec, n_entries, dup_mask, bx, by, lane);
            }
        }
    }
    if (overflow && lane == 0) {  // the exception
        g.slow_list[atomicAdd(&g.counters[PAR_CNT_SLOW], 1)] = ci;
        // (the host skips the overflow launch when its per-column pair counts rule an overflow out: should a column
        // get here all the same, its pixels would silently keep the background -- say so)
        if (!a.overflow_launched) atomicOr(&g.counters[PAR_CNT_ERROR], (int32_t)PAR_DEVERR_OVERFLOW);
    }
    stamp(g, a.flags, 2, 4);
}

// ROLES = 2 (a frame on its own: latency): a workgroup is the two wavefronts of ONE column (columns_wave's barrier is
// a workgroup barrier). ROLES = 1 (PAR_RENDER_PIPELINED, a frame among several in flight: throughput): a workgroup
// is four wavefronts, each with a column of its own and all of that column's walks; the second wavefront of a
// column is a launch, a scan of the counts and a wavefront slot that most columns of a sparse scene do not need.
template <int ROLES>
constexpr int col_waves() { return ROLES == 1 ? 4 : ROLES; }

// `n_cols_bound` bounds the column list (the wavefronts past it do the background walks when background rays are
// traced).
template <int ROLES>
__device__ __forceinline__ void columns_block(const par_grid_dev& g, const par_render_args& a, ColWave* sm,
                                              ColTeam* team_sh, int block, int n_cols_bound) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    __builtin_amdgcn_s_setprio(3);  // latency-bound wavefronts go before the streaming fill's when both want to issue
    if (ROLES == 1) {
        columns_wave<1>(g, a, sm[wave], team_sh, block * col_waves<1>() + wave, 0, n_cols_bound);
    } else {
        if (threadIdx.x == 0) team_sh->walk_used = 0;
        __syncthreads();  // (every wavefront of the workgroup is still here)
        columns_wave<ROLES>(g, a, sm[wave], team_sh, block, wave, n_cols_bound);
    }
}

template <int ROLES>
__global__ __launch_bounds__(col_waves<ROLES>() * 64) void columns_kernel(par_grid_dev g, par_render_args a,
                                                                           int n_cols_bound) {
    __shared__ ColWave sm[col_waves<ROLES>()];
    __shared__ ColTeam team_sh;
    columns_block<ROLES>(g, a, sm, &team_sh, (int)blockIdx.x, n_cols_bound);
    stamp(g, a.flags, 2, 7);
}

__device__ bool lane_shadow_walk(const par_grid_dev& g, const uint8_t* count, const par_slot* slots, int sx, int sy,
                                 int sz, const par_frame_dyn& dyn, int self, int ox, int oy, int oz, float ix,
                                 float iy, float iz);

// ------------------------------------------------------------------------------------------------------------
// bgline_kernel: the shadow ray of the background pixels of screen column x (alt:704-742 for a texel with normal 0,
// y = z = 0, entity_index 0). It is the same ray for every row, so it is traced once per x; fill_kernel copies the
// result into the `lit` plane. (The colour of a background pixel does not depend on it, SURVEY a-6.)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bgline_kernel(par_grid_dev g, par_render_args a) {
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (x >= a.W) return;
    const par_frame_dyn dyn = a.dyn_ptr ? *a.dyn_ptr : a.dyn;
    // towards_light = normalize_L1(light - (x, 0, 0)), alt:711-715 + spr:28-35
    const float dx = (float)(dyn.lx - x), dy = (float)(dyn.ly - 0), dz = (float)(dyn.lz - 0);
    float tx, ty, tz, inv_x, inv_y, inv_z;
    normalize_l1_and_inverse(dx, dy, dz, tx, ty, tz, inv_x, inv_y, inv_z);  // alt:711-719
    const int ox = (int)(int16_t)x;                                    // alt:720-722
    const int bx = div_bin(x, a.magic_b), sy = a.H / a.B;              // alt:724-727
    bool lit = true;
    const par_bgwalk* w = g.bgwalk + bx;
    const int n = w->cnt;
    if (n >= 0) {
        for (int r = 0; r < n; r++) {
            const par_slot rec = w->rec[r];
            if (rec.entity != 0 && slab_hit(rec, ox, 0, 0, inv_x, inv_y, inv_z)) {  // self = entity 0, alt:484-491
                lit = false;
                break;
            }
        }
    } else {
        lit = lane_shadow_walk(g, a.count, a.slots, bx, sy, 0, dyn, 0, ox, 0, 0, inv_x, inv_y, inv_z);
    }
    g.bglit[x] = lit ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------
// fill_body / fill_kernel: background for every pixel of the row range. A pixel no primitive covers is
// {127,127,127,0} (alt:281) times ambient (alt:735); its palette index is "none". The render kernels overwrite the
// pixels primitives cover afterwards; the fill does not depend on the hash, so it runs beside the build.
// Lane i of a wavefront owns pixels [8i, 8i+8) of a 512-pixel run: two 16-byte frame stores per lane, laid out so
// that each store instruction of the wavefront covers 1 KiB contiguously, plus one 8-byte palette-index store.
// Requires W % 8 == 0 and 16-byte aligned planes; otherwise fill_generic_kernel runs.
// ------------------------------------------------------------------------------------------------------------

// `part` = {first chunk, end chunk} of this launch's share, or {0, -1} for all of them.
__device__ __forceinline__ void fill_body(const par_render_args& a, uint32_t out_rgba, const uint8_t* bglit, int block,
                                          int n_blocks, int2 part) {
    const int W = a.W;
    const int rows = a.row_end - a.row_begin;
    const int cpr = (W + 511) >> 9;  // 512-pixel chunks per row
    const int n_chunks = part.y < 0 ? rows * cpr : part.y;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    uint32_t* fb = reinterpret_cast<uint32_t*>(a.out.fb);
    // one chunk per wavefront per iteration; the chunk index is wave-uniform (kept in scalar registers)
    for (int c = __builtin_amdgcn_readfirstlane(part.x + block * wpb + ((int)threadIdx.x >> 6)); c < n_chunks;
         c += n_blocks * wpb) {
        const int y = c / cpr, x0 = (c - y * cpr) << 9;
        const size_t rowbase = (size_t)y * W;
        if (fb) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int x = x0 + h * 256 + lane * 4;
                if (x < W) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = {out_rgba, out_rgba, out_rgba, out_rgba};
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(fb + rowbase + x));
                }
            }
        }
        if (a.out.palidx) {
            const int x = x0 + lane * 8;
            if (x < W) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 v = {0xFFFFFFFFu, 0xFFFFFFFFu};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(a.out.palidx + rowbase + x));
            }
        }
        if (a.out.lit) {  // the background ray of column x, traced once by bgline_kernel
            const int x = x0 + lane * 8;
            if (x < W) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(bglit + x),
                                            reinterpret_cast<u32x2*>(a.out.lit + rowbase + x));
            }
        }
    }
}

__global__ __launch_bounds__(256) void fill_kernel(par_render_args a, uint32_t out_rgba, const uint8_t* bglit) {
    fill_body(a, out_rgba, bglit, (int)blockIdx.x, (int)gridDim.x, make_int2(0, -1));
}

// The background fill does not depend on the hash, and it takes about as long as the column records of a
// 16 Mpixel frame: one launch for both (the first `n_col` workgroups build column records, the others fill), so the
// fill costs the frame's launch chain neither a link nor its own duration.
template <int ROLES>
__global__ __launch_bounds__(col_waves<ROLES>() * 64) void columns_fill_kernel(par_grid_dev g, par_render_args a,
                                                                                uint32_t out_rgba, int n_col_blocks,
                                                                                int n_cols_bound, int2 part) {
    __shared__ ColWave sm[col_waves<ROLES>()];
    __shared__ ColTeam team_sh;
    if ((int)blockIdx.x < n_col_blocks) {
        columns_block<ROLES>(g, a, sm, &team_sh, (int)blockIdx.x, n_cols_bound);
    } else {
        stamp(g, a.flags, 2, 0);
        fill_body(a, out_rgba, nullptr, (int)blockIdx.x - n_col_blocks, (int)gridDim.x - n_col_blocks, part);
    }
    stamp(g, a.flags, 2, 7);
}

// The same for the two short kernels of the hash build: each carries a smaller share of the fill.
template <int ENT>
__global__ __launch_bounds__(256) void insert_fill_kernel(par_grid_dev g, par_bin_args b, par_render_args a,
                                                           uint32_t out_rgba, int n_insert, int2 part) {
    stamp(g, a.flags, 0, 0);
    if ((int)blockIdx.x < n_insert) {
        bin_insert_body<ENT>(g, b, (int)blockIdx.x, n_insert);
    } else {
        fill_body(a, out_rgba, nullptr, (int)blockIdx.x - n_insert, (int)gridDim.x - n_insert, part);
    }
    stamp(g, a.flags, 0, 7);
}

__global__ __launch_bounds__(256) void resolve_fill_kernel(par_grid_dev g, par_bin_args b, par_render_args a,
                                                            uint32_t out_rgba, int n_resolve, int2 part) {
    stamp(g, a.flags, 1, 0);
    if ((int)blockIdx.x < n_resolve) {
        bin_resolve_body(g, b, (int)blockIdx.x, n_resolve);
    } else {
        fill_body(a, out_rgba, nullptr, (int)blockIdx.x - n_resolve, (int)gridDim.x - n_resolve, part);
    }
    stamp(g, a.flags, 1, 7);
}

// Small scenes: the whole hash build in ONE launch. The first `n_build` workgroups insert, meet at a barrier of their
// own (they are few and the first of the grid, so they are resident together whatever else runs), and resolve; the
// others carry the fill shares of both launches this replaces. A kernel boundary costs the frame's launch chain about
// 2 us on the device and the host a launch (about 4 us); the barrier costs less than either.
// The barrier: what insert hands to resolve is stored write-through and loaded past the L2 (the L2s of the XCDs
// are not coherent with each other; a device-scope release fence instead would write back every dirty line of the
// L2, the fill's included: measured 14 us for this kernel). Every thread waits for its stores, one thread per
// workgroup arrives on a counter and waits for the others; the last workgroup to leave resets the counters for the
// next frame. The wait is bounded (a lost workgroup must not hang the GPU): should it
// ever expire, resolve is skipped (the frame comes out empty) and the sticky g.counters[PAR_CNT_ERROR] says so; the
// host reports it as PAR_ERR_DEVICE.
// Returns false when the wait expired (some build workgroup never arrived): the caller then skips resolve, so that
// the frame comes out visibly incomplete rather than subtly wrong, and the sticky error word says why.
__device__ __forceinline__ bool build_barrier(const par_grid_dev& g, int n_build, bool test_skip_arrival,
                                              int spin_bound) {
    __shared__ int32_t timed_out;
    // every store of this thread has completed (the hand-over stores write through, st_shared)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) timed_out = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t* arrived = g.build_sync;
        int32_t* left = g.build_sync + 32;  // (a cache line apart)
        // (tests: this workgroup is "lost": it never arrives, the others wait until their bound and flag the frame;
        // it leaves the counters as a frame without the fault would: the last to leave resets them)
        if (!test_skip_arrival) __hip_atomic_fetch_add(arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_build) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > spin_bound) {
                atomicOr(&g.counters[PAR_CNT_ERROR], (int32_t)PAR_DEVERR_BARRIER);
                timed_out = 1;
                break;
            }
        }
        if (__hip_atomic_fetch_add(left, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_build - 1) {
            __hip_atomic_store(arrived, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(left, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    // acquire: drop what this XCD's L2 holds of the handed-over lines (two XCDs write neighbouring nodes of one
    // line; each keeps the line with its own part current and the other's stale). An invalidate, no write-back.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return timed_out == 0;
}

template <int ENT>
__global__ __launch_bounds__(256) void build_fill_kernel(par_grid_dev g, par_bin_args b, par_render_args a,
                                                          uint32_t out_rgba, int n_build, int2 part) {
    stamp(g, a.flags, 0, 0);
    if ((int)blockIdx.x < n_build) {
        bin_insert_body<ENT, true>(g, b, (int)blockIdx.x, n_build);
        // (the bound: seconds in production -- a workgroup that is merely late must not be given up on --, tens of
        // milliseconds in the test that loses one on purpose)
        const bool met = build_barrier(g, n_build, b.test_lose_wg != 0 && blockIdx.x == 0,
                                       b.test_lose_wg != 0 ? (1 << 14) : (1 << 22));
        stamp(g, a.flags, 0, 3);
        if (met) bin_resolve_body<true>(g, b, (int)blockIdx.x, n_build);
    } else if (part.y != 0) {
        fill_body(a, out_rgba, nullptr, (int)blockIdx.x - n_build, (int)gridDim.x - n_build, part);
    }
    stamp(g, a.flags, 0, 7);
}

// Any plane, any geometry: one pixel per thread (parity / debugging planes and odd view sizes).
__global__ __launch_bounds__(256) void fill_generic_kernel(par_render_args a, uint32_t out_rgba, int do_fb,
                                                            int do_pal, int do_lit, const uint8_t* bglit) {
    const long long npix = (long long)(a.row_end - a.row_begin) * a.W;
    par_pixel px;
    px.normal = par_vec3{0.f, 0.f, 0.f};
    px.color.red = px.color.green = px.color.blue = (uint8_t)a.background;
    px.color.alpha = 0;
    px.y = 0; px.z = 0; px.entity_index = 0;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
         p += (long long)gridDim.x * blockDim.x) {
        if (do_fb && a.out.fb) reinterpret_cast<uint32_t*>(a.out.fb)[p] = out_rgba;
        if (do_pal && a.out.palidx) a.out.palidx[p] = PAR_PALIDX_BACKGROUND;
        if (a.out.brightness) a.out.brightness[p] = a.ambient;
        if (a.out.gbuf) a.out.gbuf[p] = px;
        if (do_lit && a.out.lit) a.out.lit[p] = bglit[p % a.W];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Per-lane shadow walk: trace_hash_for_light (alt:399-500) exactly as written, for the rare pixel whose shadow ray
// starts in a bin that holds no primitive (negative world z, sprite depths outside the box) or whose walk was too
// long to record, in the render kernels.
// ------------------------------------------------------------------------------------------------------------
__device__ bool lane_shadow_walk(const par_grid_dev& g, const uint8_t* count, const par_slot* slots, int sx, int sy,
                                 int sz, const par_frame_dyn& dyn, int self, int ox, int oy, int oz, float ix,
                                 float iy, float iz) {
    const float bsx = (float)sx, bsy = (float)sy, bsz = (float)sz;             // alt:406-408
    const float dx = (float)dyn.lbx - bsx, dy = (float)dyn.lby - bsy, dz = (float)dyn.lbz - bsz;  // alt:410-416
    float largest = __builtin_fabsf(dx);                                          // alt:419-421
    if (largest < __builtin_fabsf(dy)) largest = __builtin_fabsf(dy);
    if (largest < __builtin_fabsf(dz)) largest = __builtin_fabsf(dz);
    const float stx = dx / largest, sty = dy / largest, stz = dz / largest;       // alt:423-425
    float cx = bsx, cy = bsy, cz = bsz, tx = bsx, ty = bsy, tz = bsz;             // alt:427-428
    int counter = 0;                                                              // alt:429
    const int start = flat_index(g.gy, g.gz, sx, sy, sz);                         // alt:430
    for (int i = 0; i < (int)largest;) {                                          // alt:432
        cx = tx; cy = ty; cz = tz;                                                // alt:436
        if (counter == 0) { cx = tx + stx; counter++; }                           // alt:438-440
        else if (counter == 1) { cy = ty + sty; counter++; }                      // alt:441-443
        else if (counter == 2) { cz = tz + stz; counter++; }                      // alt:444-446
        else if (counter == 3) { cx = tx + stx; cy = ty + sty; counter++; }       // alt:447-450
        else if (counter == 4) { cx = tx + stx; cz = tz + stz; counter++; }       // alt:451-454
        else if (counter == 5) { cy = ty + sty; cz = tz + stz; counter++; }       // alt:455-458
        else {                                                                    // alt:459-466
            cx = cx + stx; cy = cy + sty; cz = cz + stz;
            tx = cx; ty = cy; tz = cz;
            counter = 0;
            i++;
        }
        const int b = flat_index(g.gy, g.gz, (int)cx, (int)cy, (int)cz);          // alt:468-470
        if (b == start) continue;                                                 // alt:471-473
        if (b < 0 || b >= g.volume) continue;  // out-of-range flat index reads as an empty bin (alt:476)
        const int cnt = count[b];
        for (int j = 0; j < cnt; j++) {                                           // alt:480
            const par_slot rec = slots[(size_t)b * PAR_SLOTS + j];
            if (rec.entity == self) continue;                                     // alt:484-487
            if (slab_hit(rec, ox, oy, oz, ix, iy, iz)) return false;              // alt:489-491
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// render_chunk: one pixel per lane, for up to 64 pixels of one column (wavefront level: no workgroup cooperation;
// must be called by whole wavefronts). `rec_` is the column's record; `own` >= 0: the lanes render the pixels entry
// `own` is the first to cover, -1: all pixels. `generic`: the column has no usable record (it overflowed one):
// the primary pass then reads the column's bins straight from the hash, as the reference does, and every shadow
// walk is done here.
// ------------------------------------------------------------------------------------------------------------
struct WaveScratch {  // per wavefront: what wave_walk needs
    par_slot stage[PAR_BIN_WALK];
};

// A column record's wave-uniform tables, one element per lane (loaded once per wavefront and column): an element is
// then read with v_readlane instead of a dependent load per use.
struct ColumnRegs {
    uint4 ent;     // lane e: entries[e]
    int32_t ebz;   // lane e: ebz[e]
    uint2 nb;      // lane n: nb[n]
};

__device__ __forceinline__ par_slot slot_of_lane(const uint4& v, int e) {
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)v.x, e), d1 = (uint32_t)__builtin_amdgcn_readlane((int)v.y, e);
    const uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane((int)v.z, e);
    par_slot r;
    r.px = (int16_t)(d0 & 0xFFFF); r.py = (int16_t)(d0 >> 16);
    r.pz = (int16_t)(d1 & 0xFFFF); r.ex = (int16_t)(d1 >> 16);
    r.ey = (int16_t)(d2 & 0xFFFF); r.ez = (int16_t)(d2 >> 16);
    r.entity = __builtin_amdgcn_readlane((int)v.w, e);
    return r;
}

// What an entry pass fetched ahead of its own entry's texel (tex = -1: nothing).
struct OwnTexel {
    int tex, pal, depth;
    par_texel ti;
};

// DBG: the debug / instrumentation flags of the frame are looked at (ablation bits 24-26, time stamps bit 29, ray
// counting); the production kernels are compiled without them.
// IDS: the scene may have a sprite-id table (looked at at run time); without one (the render launch knows) every
// entity uses sprite 0 and the table arithmetic (two 32-bit multiplies per candidate entry) is compiled out.
// FULL: the parity planes (brightness, lit, G-buffer) may be asked for; a production frame (RGBA + palette index)
// runs kernels compiled without them (their pointers cost scalar registers the hot loops need).
template <bool GENERIC, bool DBG, bool IDS = true, bool FULL = true>
__device__ __forceinline__ void render_chunk(const par_grid_dev& g, const par_render_args& a, const par_colrec& rec_,
                                             const ColumnRegs& cr, uint64_t dup, const par_frame_dyn& dyn,
                                             int n_entries, int n_nb, int bx, int by, int own, int col, int row,
                                             int row_lo, int row_hi, int col_lo, int col_hi, bool valid, int lane,
                                             WaveScratch* ws,
                                             const OwnTexel& pre = OwnTexel{-1, 0, 0, par_texel{0.f, 0.f, 0.f, 0u}}) {
    const int W = a.W, H = a.H;
    const uint32_t fl = DBG ? a.flags : 0u;
    const float ambient = a.ambient;
    const uint32_t bg_rgba = a.background | (a.background << 8) | (a.background << 16);
    const int32_t* depth0 = a.sprites[0].depth;
    const int pre_tex = pre.tex, pre_pal = pre.pal;
    const par_texel pre_ti = pre.ti;
    // ---- primary ray, alt:271-397: the column's slot records front to back ------------------------------------
    bool hit = false;
    int p_entity = 0, p_y = 0, p_z = 0, p_tex = 0;
    {
        const int i = col;
        const int world_j = (int)(int16_t)(H - row);  // alt:280
        int adjacent = 0;                             // alt:282
        int closest = INT_MIN;                        // alt:289
        int cur_bz = -2;
        int first_cover = n_entries;                  // first entry whose test (alt:310-317) passes
        bool hit_in_bin = false;
        bool done = !valid;
        int w_ybase = 0, w_pz = 0, w_d = 0;
        auto next_bin = [&](int bz) {  // the previous occupied bin is complete
            adjacent += hit_in_bin ? 1 : 0;      // alt:368
            if (adjacent >= 2) done = true;      // alt:372-374
            if (bz != cur_bz + 1) adjacent = 0;  // an empty bin lies in between (alt:298-300)
            cur_bz = bz;
            hit_in_bin = false;
        };
        const bool has_ids = IDS && a.sprite_ids != nullptr;
        auto test = [&](const par_slot& rec, int e) {
            const int top = rec.py + rec.ey + rec.pz + rec.ez;
            // alt:310-317 (one predicate: no branch per condition)
            const bool inside = (!done) & (i >= rec.px) & (i < rec.px + rec.ex) & (world_j > rec.py + rec.pz) &
                                (world_j <= top);
            if (inside) {
                first_cover = min(first_cover, e);
                const int sprite_row = top - world_j;                         // alt:324-326
                const int t = sprite_row * PAR_SPRITE_W + (i - rec.px);       // alt:330-332
                const int sid = has_ids ? a.sprite_ids[rec.entity] : 0;       // alt:321-322
                // (an entry pass has its own entry's texel already: same index, same table)
                const int d = (!GENERIC && t == pre_tex && e == own) ? pre.depth
                                                                       : ((sid == 0) ? depth0[t] : a.sprites[sid].depth[t]);
                const int depth = rec.py - rec.pz + min(0, rec.ey - sprite_row) - d;  // alt:336-341
                if (closest < depth) {                                        // alt:344-346
                    closest = depth;
                    w_ybase = rec.py + rec.ey + rec.ez - sprite_row;          // alt:356-359
                    w_pz = rec.pz;                                            // alt:360-361
                    w_d = d;
                    p_entity = rec.entity;                                    // alt:363
                    p_tex = sid * PAR_SPRITE_TEXELS + t;
                    hit = true;
                    hit_in_bin = true;                                        // alt:365
                }
            }
        };
        if (!GENERIC) {
            // The record's entries as one flat list, front to back. Lane e holds entry e: ONE vector comparison says
            // which entries can cover a pixel of this chunk at all (not a repeat of an earlier entry's entity: the
            // same AABB in another bin gives the same depth, so it can neither improve `closest`, strict compare
            // alt:344, nor be the first to cover; and its rectangle meets the chunk's box, rows [row_lo, row_hi] x
            // columns [col_lo, col_hi]); only those
            // are visited. The bins of the entries skipped are occupied bins without a hit: they leave `adjacent`
            // alone (only an EMPTY bin resets it, alt:298-300), so all the visit needs to know of them is whether
            // an empty bin lies between two visited ones: `gaps` counts the empty stretches up to an entry's bin.
            const int16_t my_px = (int16_t)(cr.ent.x & 0xFFFF), my_py = (int16_t)(cr.ent.x >> 16);
            const int16_t my_pz = (int16_t)(cr.ent.y & 0xFFFF);
            const int16_t my_ey = (int16_t)(cr.ent.z & 0xFFFF), my_ez = (int16_t)(cr.ent.z >> 16);
            const int16_t my_ex = (int16_t)(cr.ent.y >> 16);
            const int my_top_row = H - (my_py + my_ey + my_pz + my_ez), my_end_row = H - (my_py + my_pz);
            const bool mine = lane < n_entries && !((dup >> lane) & 1) && !(row_hi < my_top_row || row_lo >= my_end_row) &&
                              !(col_hi < my_px || col_lo >= my_px + my_ex);
            uint64_t todo = __ballot(mine);
            const int prev_bz = __shfl_up(cr.ebz, 1);
            const uint64_t gap_mask = __ballot(lane > 0 && lane < n_entries && cr.ebz != prev_bz && cr.ebz != prev_bz + 1);
            const int my_gaps = __popcll(gap_mask & ((2ull << lane) - 1ull));
            int cur_gaps = -1;
            while (todo) {
                const int e = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int bz = __builtin_amdgcn_readlane(cr.ebz, e);
                if (bz != cur_bz) {  // the previous visited bin is complete (next_bin, across the bins skipped)
                    const int gaps = __builtin_amdgcn_readlane(my_gaps, e);
                    adjacent += hit_in_bin ? 1 : 0;         // alt:368
                    if (adjacent >= 2) done = true;         // alt:372-374
                    if (gaps != cur_gaps) adjacent = 0;     // an empty bin lies in between (alt:298-300)
                    cur_gaps = gaps;
                    cur_bz = bz;
                    hit_in_bin = false;
                }
                // a lane whose pixel an earlier entry owns has nothing to do in this pass
                if (first_cover < own) done = true;  // (never in tile mode: own = -1)
                if (__all(done)) break;  // wavefront early-out
                test(slot_of_lane(cr.ent, e), e);
            }
        } else {  // the column's bins as they lie in the hash, alt:292-376
            const int col_base = flat_index(g.gy, g.gz, bx, by, 0);
            for (int bz = 0; bz < g.gz; bz++) {
                const int c = a.count[col_base + bz];  // (wave-uniform)
                if (c == 0) continue;
                next_bin(bz);
                if (__all(done)) break;
                for (int k = 0; k < c; k++) test(a.slots[(size_t)(col_base + bz) * PAR_SLOTS + k], 0);
            }
        }
        // an entry pass renders the pixels entry `own` is the first to cover; a tile pass renders them all
        valid = valid && (own < 0 || first_cover == own);
        hit = hit && valid;
        if (hit) {
            p_y = w_ybase - w_d;
            p_z = w_pz + w_d;
        }
    }

    if (!GENERIC) stamp(g, fl, 3, 3);
    // ---- shading, alt:704-758 ---------------------------------------------------------------------------------
    float nx = 0.f, ny = 0.f, nz = 0.f;
    uint32_t rgba = bg_rgba;
    int pal_index = PAR_PALIDX_BACKGROUND;
    float bright = ambient;
    bool lit = true;
    const bool shade = hit && !(fl & (1u << 26));  // bit 26: ablation (timing experiments only), no shading
    bool need_walk = false;  // the shadow ray still has to be resolved
    float inv_x = 0.f, inv_y = 0.f, inv_z = 0.f, b_lit = 0.f;
    int sy = 0, sz = 0, ox = 0, oy = 0, oz = 0;
    if (shade) {
        // normal (alt:349-350) + resolved palette colour (alt:352-354)
        par_texel ti = pre_ti;
        pal_index = pre_pal;
        if (p_tex != pre_tex) {  // another entry won (or nothing was fetched ahead)
            ti = a.texinfo[p_tex];
            if (a.out.palidx) {
                const int sid = IDS ? p_tex / PAR_SPRITE_TEXELS : 0;
                pal_index = a.sprites[sid].color[p_tex - sid * PAR_SPRITE_TEXELS];
            }
        }
        nx = ti.nx; ny = ti.ny; nz = ti.nz;
        rgba = ti.rgba;
        const int wx = col, wy = p_y, wz = p_z;  // alt:707-709
        // towards_light = normalize_L1(light - world), alt:711-715 + spr:28-35
        const float dx = (float)(dyn.lx - wx), dy = (float)(dyn.ly - wy), dz = (float)(dyn.lz - wz);
        float tx, ty, tz;
        normalize_l1_and_inverse(dx, dy, dz, tx, ty, tz, inv_x, inv_y, inv_z);  // alt:711-719
        const float dot = nx * tx + ny * ty + nz * tz;                     // alt:746-747 (no contraction)
        const float diffuse = std_max(0.f, dot);                           // alt:745
        b_lit = std_min(1.f, diffuse + ambient);                           // alt:758
        // alt:725-726: the start bin's row is bin(H - wy - wz). wy + wz is world_j whatever the texel's depth (it is
        // subtracted from y and added to z, alt:356-361), and H - world_j is the pixel's screen row (alt:280; no
        // view is taller than a `short` holds, par_create): a row of this column's tile, whose bin row is `by`. The
        // overflow path keeps the division.
        sy = GENERIC ? div_bin(H - wy - wz, a.magic_b) : by;
        sz = div_bin(wz, a.magic_b);                                       // alt:727
        ox = (int)(int16_t)col; oy = (int)(int16_t)p_y; oz = (int)(int16_t)p_z;  // alt:720-722
        // shadow ray, alt:738-742: columns_kernel has walked from every occupied bin of the column
        int woff = 0, wcnt = -1;
        if (!GENERIC) {
            for (int n = 0; n < n_nb; n++) {
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)cr.nb.x, n);
                const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)cr.nb.y, n);
                if ((int)(int16_t)(d0 & 0xFFFF) == sz) {
                    woff = (int)(d1 & 0xFFFF);
                    wcnt = (int)(int16_t)(d1 >> 16);  // -1: the walk was too long to record
                }
            }
        }
        if (wcnt >= 0) {
            // two records (four 16-byte loads) per step: their loads are in flight together
            const char* walk_base = reinterpret_cast<const char*>(rec_.walk);  // (wave-uniform; 32-bit lane offsets)
            const float fox = (float)ox, foy = (float)oy, foz = (float)oz;
            auto test_list = [&](auto finite) {
                for (int r0 = 0; r0 < wcnt && lit; r0 += 2) {
                    uint4 wa[2], wb[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t off = (uint32_t)(woff + min(r0 + u, wcnt - 1)) * (uint32_t)sizeof(par_walkrec);
                        const uint4* q = reinterpret_cast<const uint4*>(walk_base + off);
                        wa[u] = q[0];
                        wb[u] = q[1];
                    }
                    // (all four loads are issued before the first record is looked at: the compiler otherwise sinks
                    // a record's planes below the comparison of its entity, a second round trip)
                    asm volatile("" ::"v"(wa[0].x), "v"(wa[1].x), "v"(wb[0].x), "v"(wb[1].x));
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const v2f rx = {__uint_as_float(wa[u].x), __uint_as_float(wa[u].y)};
                        const v2f ry = {__uint_as_float(wa[u].z), __uint_as_float(wa[u].w)};
                        const v2f rz = {__uint_as_float(wb[u].x), __uint_as_float(wb[u].y)};
                        if (r0 + u < wcnt && (int)wb[u].z != p_entity &&
                            slab_hit_rec<decltype(finite)::value>(rx, ry, rz, fox, foy, foz, inv_x, inv_y, inv_z)) {  // alt:484-491
                            lit = false;
                        }
                    }
                }
            };
            // (wave-uniform choice: an axis-parallel light direction, 1 / 0, takes the reference's own sequence)
            if (__all(__builtin_isfinite(inv_x) && __builtin_isfinite(inv_y) && __builtin_isfinite(inv_z))) {
                test_list(std::true_type{});
            } else {
                test_list(std::false_type{});
            }
        } else if (GENERIC) {
            need_walk = true;
        } else {
            // a start bin that holds no primitive (negative world z, sprite depths outside the box) has no
            // precomputed walk: trace_hash_for_light as written, per lane
            lit = lane_shadow_walk(g, a.count, a.slots, bx, sy, sz, dyn, p_entity, ox, oy, oz, inv_x, inv_y, inv_z);
        }
    }
    if (GENERIC) {
        // No precomputed walks: the wavefront walks once per distinct start bin among its lanes; every lane of that
        // bin then tests the staged records.
        for (unsigned long long pending = __ballot(need_walk); pending; pending = __ballot(need_walk)) {
            const int leader = __ffsll((long long)pending) - 1;
            const int gsy = __shfl(sy, leader), gsz = __shfl(sz, leader);
            const int n_rec = wave_walk(g, a.count, a.slots, dyn, bx, gsy, gsz, ws->stage);
            if (need_walk && sy == gsy && sz == gsz) {
                if (n_rec >= 0) {
                    for (int r = 0; r < n_rec; r++) {
                        const par_slot rec = ws->stage[r];
                        if (rec.entity != p_entity && slab_hit(rec, ox, oy, oz, inv_x, inv_y, inv_z)) {  // alt:484-491
                            lit = false;
                            break;
                        }
                    }
                } else {  // more records on the way than the stage holds: per lane, as the reference writes it
                    lit = lane_shadow_walk(g, a.count, a.slots, bx, sy, sz, dyn, p_entity, ox, oy, oz, inv_x, inv_y,
                                           inv_z);
                }
                need_walk = false;
            }
            // the next walk overwrites the stage: every lane has read it (LDS operations of one wavefront complete
            // in order; keep the compiler from moving them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (shade) bright = lit ? b_lit : ambient;
    const bool lit_px = lit;
    if (!GENERIC) stamp(g, fl, 3, 4);
    if ((fl & PAR_RENDER_COUNT_RAYS) && a.ray_counter) {
        const unsigned long long m = __ballot(valid && hit);
        if (lane == 0 && m) atomicAdd(a.ray_counter, (unsigned long long)__popcll(m));
    }

    // ---- quantise + store, alt:735, 757-758 -------------------------------------------------------------------
    if (fl & (1u << 25)) {  // bit 25: ablation (timing only), no stores; keep the values alive
        asm volatile("" ::"v"(rgba), "v"(bright), "v"(pal_index));
    } else if (valid && hit) {  // (uncovered pixels keep what the fill wrote)
        const size_t o = (size_t)(row - a.row_begin) * W + col;
#if !defined(PAR_EXP_STORE)
        if (a.out.fb) __builtin_nontemporal_store(color_scale(rgba, bright), reinterpret_cast<uint32_t*>(a.out.fb) + o);
        if (a.out.palidx) __builtin_nontemporal_store((uint8_t)pal_index, a.out.palidx + o);
#elif PAR_EXP_STORE == 1  // (experiments, tools/debug/variants.sh: plain stores)
        if (a.out.fb) reinterpret_cast<uint32_t*>(a.out.fb)[o] = color_scale(rgba, bright);
        if (a.out.palidx) a.out.palidx[o] = (uint8_t)pal_index;
#elif PAR_EXP_STORE == 2  // (timing only: no palette-index store)
        if (a.out.fb) __builtin_nontemporal_store(color_scale(rgba, bright), reinterpret_cast<uint32_t*>(a.out.fb) + o);
        asm volatile("" ::"v"(pal_index));
#elif PAR_EXP_STORE == 3  // (timing only: no frame store)
        asm volatile("" ::"v"(rgba), "v"(bright));
        if (a.out.palidx) __builtin_nontemporal_store((uint8_t)pal_index, a.out.palidx + o);
#elif PAR_EXP_STORE == 4  // (plain frame store, streaming palette index)
        if (a.out.fb) reinterpret_cast<uint32_t*>(a.out.fb)[o] = color_scale(rgba, bright);
        if (a.out.palidx) __builtin_nontemporal_store((uint8_t)pal_index, a.out.palidx + o);
#endif
        if (FULL && a.out.brightness) a.out.brightness[o] = bright;
        if (FULL && a.out.lit) a.out.lit[o] = lit_px ? 1 : 0;
        if (FULL && a.out.gbuf) {
            par_pixel pxl;
            pxl.normal = par_vec3{nx, ny, nz};
            pxl.color.red = (uint8_t)(rgba & 0xFF);
            pxl.color.green = (uint8_t)((rgba >> 8) & 0xFF);
            pxl.color.blue = (uint8_t)((rgba >> 16) & 0xFF);
            pxl.color.alpha = (uint8_t)(rgba >> 24);
            pxl.y = p_y;
            pxl.z = p_z;
            pxl.entity_index = p_entity;
            a.out.gbuf[o] = pxl;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// render_column_generic: a column WITHOUT a usable record (it overflowed one, or PAR_FORCE_GENERIC=1): the whole
// tile, 64 pixels per wavefront; chunk k belongs to wavefront k mod (max_parts * PAR_WAVE_NW) of the column's
// workgroups. The primary pass reads the column's bins straight from the hash, as the reference does, and the shadow
// walks are done in the kernel (render_chunk<true>). Only bx, by of the record are read.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void render_column_generic(const par_grid_dev& g, const par_render_args& a, int ci,
                                                      int part, int max_parts, WaveScratch* ws) {
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const par_colrec& rec_ = g.colrec[ci];
    const int bx = rec_.bx, by = rec_.by;
    ColumnRegs cr;
    cr.ent = make_uint4(0, 0, 0, 0);
    cr.ebz = 0;
    cr.nb = make_uint2(0, 0);
    const int n_workers = max_parts * PAR_WAVE_NW, worker = part * PAR_WAVE_NW + wave;
    const int W = a.W, H = a.H, B = a.B;
    const par_frame_dyn dyn = a.dyn_ptr ? *a.dyn_ptr : a.dyn;
    const int c0 = bx * B;
    const int rw = min(B, W - c0);
    const int ry0 = max(by * B, a.row_begin), rows_hi = min(min((by + 1) * B, H), a.row_end);
    const int rh = rows_hi - ry0;
    if (rw <= 0 || rh <= 0) return;
    const int area = rw * rh;
    const int n_chunks = (area + 63) >> 6;
    // floor(p / rw) == __umulhi(p, magic_w) for p * rw < 2^32; a 1-pixel-wide rectangle has no such multiplier
    const uint32_t magic_w = (uint32_t)(0xFFFFFFFFu / (uint32_t)rw) + 1u;
    for (int c = worker; c < n_chunks; c += n_workers) {
        const int pidx = c * 64 + lane;
        const int pyy = (rw == 1) ? pidx : (int)__umulhi((uint32_t)pidx, magic_w);
        const int col = c0 + (pidx - pyy * rw), row = ry0 + pyy;
        // the chunk's first and last row (wave-uniform)
        const int p_first = c * 64, p_last = min(p_first + 63, area - 1);
        const int row_lo = ry0 + ((rw == 1) ? p_first : (int)__umulhi((uint32_t)p_first, magic_w));
        const int row_hi = ry0 + ((rw == 1) ? p_last : (int)__umulhi((uint32_t)p_last, magic_w));
        render_chunk<true, true>(g, a, rec_, cr, 0, dyn, 0, 0, bx, by, -1, col, row, row_lo, row_hi, c0, c0 + rw - 1, pidx < area, lane,
                           ws + wave);
    }
}

// render_item: one work item = one 64-pixel chunk of one column with a record (columns_kernel listed it), one
// wavefront. `pass`: the entry whose rectangle is visited (the lanes render the pixels it is the first to cover), or
// PAR_ITEM_TILE: the whole tile. The record is read where columns_kernel left it: its fields are wave-uniform, so
// they arrive through the scalar cache; the entry and bin tables sit one element per lane in registers
// (v_readlane). No LDS, no barrier, no loop over chunks: every wavefront of the launch is a handful of dependent
// loads long, whatever its column looks like.
// The item as two 16-byte words (par_item: {ci, visit, where, bins}, {entry}).
template <bool DBG, bool IDS, bool FULL>
__device__ __forceinline__ void render_item(const par_grid_dev& g, const par_render_args& a, uint4 ia, uint4 ib,
                                            int lane) {
    const uint32_t fl = DBG ? a.flags : 0u;
    const int ci = (int)ia.x;
    const uint32_t pass = ia.y >> 16;
    const int chunk = (int)(ia.y & 0xFFFFu);
    const bool simple = (ia.z & PAR_ITEM_SIMPLE) != 0;
    constexpr bool tile_mode = false;  // (whole-tile visits are render_tile_item's)
    const par_colrec& rec_ = g.colrec[ci];
    // ---- everything the ITEM says: the column, the rectangle visited, this lane's pixel, and (entry passes) the
    // texel of the pass's own entry, the likeliest winner of the pixel. Its depth, normal, colour and palette index
    // are fetched right away, beside the column's record, instead of one and two round trips after it.
    const int W = a.W, H = a.H, B = a.B;
    const int bx = (int)(ia.z & 0x3FFu), by = (int)((ia.z >> 10) & 0x3FFu);
    const int c0 = bx * B;
    const int tw = min(B, W - c0);
    const int rows_lo = max(by * B, a.row_begin), rows_hi = min(min((by + 1) * B, H), a.row_end);
    par_slot own_rec;  // the pass's entry as the item carries it (par_slot as it lies in memory)
    own_rec.px = (int16_t)(ib.x & 0xFFFFu); own_rec.py = (int16_t)(ib.x >> 16);
    own_rec.pz = (int16_t)(ib.y & 0xFFFFu); own_rec.ex = (int16_t)(ib.y >> 16);
    own_rec.ey = (int16_t)(ib.z & 0xFFFFu); own_rec.ez = (int16_t)(ib.z >> 16);
    own_rec.entity = (int32_t)ib.w;
    int rx0, rw, ry0, rh;
    if (tile_mode) {
        rx0 = c0; rw = tw; ry0 = rows_lo; rh = rows_hi - rows_lo;
    } else {
        rx0 = max((int)own_rec.px, c0);
        rw = min(own_rec.px + own_rec.ex, c0 + tw) - rx0;
        // alt:314-317: world_j in (py+pz, py+ey+pz+ez], and row = H - world_j (alt:280)
        ry0 = max(H - (own_rec.py + own_rec.ey + own_rec.pz + own_rec.ez), rows_lo);
        rh = min(H - (own_rec.py + own_rec.pz), rows_hi) - ry0;
    }
    if (rw <= 0 || rh <= 0) return;
    const int area = rw * rh;
    const int p_first = chunk * 64, p_last = min(p_first + 63, area - 1);
    if (p_first >= area) return;
    // the rectangle is visited in vertical strips of a sprite's width, row by row within a strip (par_strips.h)
    static_assert(PAR_SPRITE_W == PAR_STRIP_W && PAR_MAX_BIN <= PAR_STRIP_MAX_SIDE, "par_strips.h is written for these");
    const par_strips st = par_strips_of(rw);
    const int n_strips = st.n_strips, sw = st.sw, lw = st.lw;
    const int pidx = p_first + lane;
    const bool valid = pidx < area;
    int col, row, strip;
    par_strip_pixel(st, rh, pidx, strip, col, row);
    col += rx0;
    row += ry0;
    // the chunk's box (wave-uniform): the rows and columns of its first and last pixel when both lie in one strip,
    // the whole rectangle otherwise
    const int last_lane = p_last - p_first;
    const int strip_a = __builtin_amdgcn_readfirstlane(strip), strip_b = __builtin_amdgcn_readlane(strip, last_lane);
    int row_lo = ry0, row_hi = ry0 + rh - 1, col_lo = rx0, col_hi = rx0 + rw - 1;
    if (strip_a == strip_b) {
        row_lo = __builtin_amdgcn_readfirstlane(row);
        row_hi = __builtin_amdgcn_readlane(row, last_lane);
        col_lo = rx0 + strip_a * sw;
        col_hi = col_lo + (strip_a == n_strips - 1 ? lw : sw) - 1;
    }
    OwnTexel pre;
    pre.tex = -1; pre.pal = 0; pre.depth = 0;
    pre.ti = par_texel{0.f, 0.f, 0.f, 0u};
    if (!tile_mode && valid && !(IDS && a.sprite_ids)) {  // (with a sprite-id table the texel is not known yet)
        const int sprite_row = own_rec.py + own_rec.ey + own_rec.pz + own_rec.ez - (int)(int16_t)(H - row);  // alt:324-326
        pre.tex = sprite_row * PAR_SPRITE_W + (col - own_rec.px);                                            // alt:330-332
        pre.depth = a.sprites[0].depth[pre.tex];
        pre.ti = a.texinfo[pre.tex];
        if (a.out.palidx) pre.pal = a.sprites[0].color[pre.tex];
    }
    // ---- what the column's RECORD says (not needed for a simple column: its item says it all) ------------------
    ColumnRegs cr;
    int n_entries_rec, n_nb;
    uint64_t dup;
    if (simple) {
        // one entry (every lane holds it; it is entry 0 of the list and the pass), the occupied bins bins.first ..
        // bins.last, each with an empty walk
        const int bz_first = (int)(ia.w & 0xFFFFu), bz_last = (int)(ia.w >> 16);
        cr.ent = ib;
        cr.ebz = bz_first;
        cr.nb = make_uint2((uint32_t)(uint16_t)(bz_first + lane), 0u);  // lane n: bin first + n, walk [0, 0)
        n_entries_rec = 1;
        n_nb = bz_last - bz_first + 1;
        dup = 0;
    } else {
        cr.ent = reinterpret_cast<const uint4*>(rec_.entries)[min(lane, PAR_COL_ENT - 1)];
        cr.ebz = rec_.ebz[min(lane, PAR_COL_ENT - 1)];
        cr.nb = reinterpret_cast<const uint2*>(rec_.nb)[lane & (PAR_COL_NB - 1)];
        // the record's header (32 bytes, wave-uniform, written by the column launch): two scalar loads
        const uint4 h0 = ld_uniform(reinterpret_cast<const uint4*>(&rec_));
        const uint4 h1 = ld_uniform(reinterpret_cast<const uint4*>(&rec_) + 1);
        n_nb = (int)(int16_t)(h0.x & 0xFFFFu);
        n_entries_rec = (int)(int16_t)(h0.x >> 16);
        dup = ((uint64_t)h1.z << 32) | h1.y;
        if ((h0.y >> 16) != 0) return;  // overflow: render_overflow_kernel's
    }
    stamp(g, fl, 3, 2);
    // The shadow test will read the walk list of the pixel's start bin, which is known only after the primary pass
    // and its depth lookups: lane n touches the first line of bin n's list now, so that those reads find it in the
    // cache instead of adding a round trip to memory at the end of the chain.
    uint32_t touched = 0;
    if (!simple && lane < n_nb && (int)(int16_t)(cr.nb.y >> 16) > 0) {
        touched = *reinterpret_cast<const uint32_t*>(rec_.walk + (cr.nb.y & 0xFFFFu));
    }
    const int n_entries = (fl & (1u << 24)) ? 0 : n_entries_rec;  // bit 24: ablation (timing only)
    const par_frame_dyn dyn = a.dyn_ptr ? ld_uniform(a.dyn_ptr) : a.dyn;  // (graph replay: uploaded before the frame)
    // (a simple column's only entry sits in every lane: it is read as entry 0 whatever its index in the record was)
    const int own = tile_mode ? -1 : (simple ? 0 : (int)pass);
    render_chunk<false, DBG, IDS, FULL>(g, a, rec_, cr, dup, dyn, n_entries, n_nb, bx, by, own, col, row, row_lo, row_hi, col_lo, col_hi, valid,
                             lane, nullptr, pre);
    asm volatile("" ::"v"(touched));  // (keeps the touch alive; nothing reads it)
}

// ------------------------------------------------------------------------------------------------------------
// render_tile_item: a work item of a column that is visited as a WHOLE TILE (every pixel of the bin's footprint; the
// dense regime: a floor, a wall of boxes): `n_chunks` consecutive 64-pixel chunks of the strip order, one pixel per
// lane. This path is bound by instruction ISSUE, not by latency (tools/issuebench.hip has the prices: a scalar
// instruction costs a SIMD as much as a vector one, a vector instruction with a scalar operand twice a plain one,
// a vector load of 16 bytes per lane sixteen times), so it is written around the wave-uniformity of everything but
// the pixel:
//   - what the chunks of a column share (record header, entry rectangles, bin table) is read once per item;
//   - the candidate entries of a chunk are read with ONE scalar load each (par_xent: every sum the test needs);
//   - the per-pixel booleans of the primary pass (done, hit in this bin, adjacent, alt:282-374) are lane MASKS in
//     scalar registers: the bin bookkeeping costs no vector instruction;
//   - the shadow test groups the lanes by start bin (nearly always one group) and reads that bin's walk list with
//     scalar loads: no per-lane record loads, no address arithmetic, no unpacking.
// Same arithmetic, in the same order, as render_chunk (alt:310-365, 704-758).
// ------------------------------------------------------------------------------------------------------------
typedef uint64_t lanemask;
__device__ __forceinline__ bool lane_of(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// todo &= ~(1 << e) in ONE scalar instruction (the compiler's todo & (todo - 1) is three)
__device__ __forceinline__ lanemask mask_clear(lanemask m, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(e));
#else
    m &= ~(1ull << e);
#endif
    return m;
}
// The hardware's min / max without the canonicalising v_max x, x the compiler puts in front of fminf / fmaxf when it
// cannot see where their operands come from (they are products of this wavefront: never signalling NaNs).
__device__ __forceinline__ float hw_min(float x, float y) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float hw_max(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float hw_min3(float x, float y, float z) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float hw_max3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

// A wave-uniform 32-byte record at base + off through the scalar cache (s_load_dwordx8 with a scalar offset: no
// address arithmetic). Issued where it stands; uniform_wait() before its first use.
__device__ __forceinline__ u32x8 uniform_fetch8(const void* base, uint32_t off) {
    u32x8 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(base), "s"(off) : "memory");
#else
    v = *reinterpret_cast<const u32x8*>(static_cast<const char*>(base) + off);
#endif
    return v;
}
__device__ __forceinline__ void uniform_wait(u32x8& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a) : : "memory");
#endif
}
__device__ __forceinline__ void uniform_wait(u32x8& a, u32x8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) : : "memory");
#endif
}

// The shadow test of one group of lanes (the lanes of a chunk whose rays start in one bin) against that bin's walk
// list, the records read with scalar loads, two per step (the column kernel pads a list of odd length with a repeat
// of its last record): the lanes of `grp` that are still lit afterwards. Called in uniform control flow (the other
// lanes compute along; their results are masked). AABB::intersect (alt:40-83) per record and lane, the record's own
// entity skipped (alt:484-487); stops as soon as no lane of the group is lit (the reference's result is an OR over
// the records).
// SIGNS: what is known about the group's inverse directions:
//   0..7  every lane's are finite, and negative on exactly the axes whose bit is set (x: 1, y: 2, z: 4). The slab
//         products of an axis are then ordered the same way in every lane -- (lo - o) * inv <= (hi - o) * inv for a
//         positive inverse (lo <= hi, rounding is monotonic), the other way round for a negative one -- so the
//         min / max of alt:55-82 need no instruction: near and far planes are picked at compile time;
//   8     finite, signs differ among the lanes: the hardware's min / max (no NaN can arise, slab_hit);
//   9     some lane's is not finite (an axis-parallel light direction, 1 / 0): the reference's own sequence.
template <int SIGNS>
__device__ __forceinline__ lanemask slab_occluded(const u32x8& w, float fox, float foy, float foz, float inv_x,
                                                  float inv_y, float inv_z, int self) {
    const v2f rx = {__uint_as_float(w[0]), __uint_as_float(w[1])};
    const v2f ry = {__uint_as_float(w[2]), __uint_as_float(w[3])};
    const v2f rz = {__uint_as_float(w[4]), __uint_as_float(w[5])};
    const v2f tx = (rx - fox) * inv_x, ty = (ry - foy) * inv_y, tz = (rz - foz) * inv_z;  // alt:49-72
    float tmin, tmax;
    if (SIGNS < 8) {
        tmin = hw_max3((SIGNS & 1) ? tx.y : tx.x, (SIGNS & 2) ? ty.y : ty.x, (SIGNS & 4) ? tz.y : tz.x);
        tmax = hw_min3((SIGNS & 1) ? tx.x : tx.y, (SIGNS & 2) ? ty.x : ty.y, (SIGNS & 4) ? tz.x : tz.y);
    } else if (SIGNS == 8) {
        tmin = hw_max3(hw_min(tx.x, tx.y), hw_min(ty.x, ty.y), hw_min(tz.x, tz.y));
        tmax = hw_min3(hw_max(tx.x, tx.y), hw_max(ty.x, ty.y), hw_max(tz.x, tz.y));
    } else {  // alt:55-82 as written
        tmin = std_min(tx.x, tx.y);
        tmax = std_max(tx.x, tx.y);
        tmin = std_max(tmin, std_min(ty.x, ty.y));
        tmax = std_min(tmax, std_max(ty.x, ty.y));
        tmin = std_max(tmin, std_min(tz.x, tz.y));
        tmax = std_min(tmax, std_max(tz.x, tz.y));
    }
    return __ballot(tmax >= tmin) & __ballot((int)w[6] != self);                               // alt:484-491
}

template <int SIGNS>
__device__ __forceinline__ lanemask walk_list_lit(const par_walkrec* wr, int wcnt, lanemask grp, float fox, float foy,
                                                 float foz, float inv_x, float inv_y, float inv_z, int self) {
    lanemask alive = grp;
    uint32_t end = (uint32_t)wcnt * (uint32_t)sizeof(par_walkrec);
    for (uint32_t off = 0; off < end; off += 2 * (uint32_t)sizeof(par_walkrec)) {  // (wave-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(off));  // (a scalar counter: the compiler otherwise counts in a vector register)
#endif
        u32x8 w0 = uniform_fetch8(wr, off);
        u32x8 w1 = uniform_fetch8(wr, off + (uint32_t)sizeof(par_walkrec));
        uniform_wait(w0, w1);
        const lanemask o0 = slab_occluded<SIGNS>(w0, fox, foy, foz, inv_x, inv_y, inv_z, self);
        const lanemask o1 = slab_occluded<SIGNS>(w1, fox, foy, foz, inv_x, inv_y, inv_z, self);
        alive &= ~(o0 | o1);
        if (alive == 0) end = 0;  // (no lane of the group is lit: done; folded into the bound, one branch per step)
    }
    return alive;
}

// The sprite depth table of sprite `sid` as a buffer resource: a load through it with an offset outside the table
// returns 0 instead of faulting (the hardware's range check), so the lanes whose pixel lies outside an entry's
// rectangle need no select in front of the load.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t depth_table(const par_sprite* sprites, int sid) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(sprites[sid].depth), 0,
                                             (int)(PAR_SPRITE_TEXELS * sizeof(int32_t)), 0x00020000);
}

// PIPE: the candidate entries' loads overlap (small frames, bound by their slowest wavefront); without it one entry
// at a time, the fewest instructions (big dense frames, bound by instruction issue).
// `depth_lds`: sprite 0's depth table in LDS (tile_depth_to_lds), or nullptr: read it from memory. A depth lookup
// is then a ds_read (some sixty cycles) instead of a trip through the vector memory path (some five hundred even when
// the CU's cache has the line), and every candidate entry of a chunk needs one before the next can be compared.
template <bool DBG, bool IDS, bool FULL, bool PIPE = false>
__device__ __forceinline__ void render_tile_item(const par_grid_dev& g, const par_render_args& a, uint4 ia, int lane,
                                                 const int32_t* depth_lds) {
    const uint32_t fl = DBG ? a.flags : 0u;
    const int ci = (int)ia.x;
    const int chunk0 = (int)(ia.y & 0xFFFFu);
    const int n_chunks = (int)ia.w;
    const par_colrec& rec_ = g.colrec[ci];
    const int W = a.W, H = a.H, B = a.B;
    const int bx = (int)(ia.z & 0x3FFu), by = (int)((ia.z >> 10) & 0x3FFu);
    const int c0 = bx * B, ty = by * B;
    const int tw = min(B, W - c0);
    const int rows_lo = max(ty, a.row_begin), rows_hi = min(min(ty + B, H), a.row_end);
    const int rh = rows_hi - rows_lo;
    if (tw <= 0 || rh <= 0) return;
    const int area = tw * rh;
    const bool have_masks = ((area + 63) >> 6) <= PAR_TILE_MASKS;  // (the column kernel made the same test)
    // ---- once per item: the record's header (two scalar loads) and its per-lane tables ---------------------------
    const uint4 h0 = ld_uniform(reinterpret_cast<const uint4*>(&rec_));
    const uint4 h1 = ld_uniform(reinterpret_cast<const uint4*>(&rec_) + 1);
    if ((h0.y >> 16) != 0) return;  // overflow: render_overflow_kernel's
    const int n_nb = (int)(int16_t)(h0.x & 0xFFFFu);
    const int n_entries = (int)(int16_t)(h0.x >> 16);
    const uint64_t dup = ((uint64_t)h1.z << 32) | h1.y;
    const uint2 nb = reinterpret_cast<const uint2*>(rec_.nb)[lane & (PAR_COL_NB - 1)];  // lane n: occupied bin n
    const int nb_bz = (int)(int16_t)(nb.x & 0xFFFFu);
    // entry rectangles relative to the tile's corner (ends exclusive), for tiles of more chunks than the record
    // carries masks for; rows of the visit start at rows_lo
    int e_r0 = 0, e_r1 = 0, e_q0 = 0, e_q1 = 0;
    lanemask eligible = 0;
    if (!have_masks) {
        const uint32_t rect = rec_.rect[lane];  // lane e: entry e's rectangle
        e_r0 = (int)(rect & 0xFFu); e_r1 = (int)((rect >> 8) & 0xFFu);
        e_q0 = (int)((rect >> 16) & 0xFFu); e_q1 = (int)(rect >> 24);
        eligible = __ballot(lane < n_entries) & ~dup & __ballot(e_r1 > e_r0) & __ballot(e_q1 > e_q0);
    }
    const lanemask nb_lanes = __ballot(lane < n_nb);
    const int row_off = rows_lo - ty;
    const par_frame_dyn dyn = a.dyn_ptr ? ld_uniform(a.dyn_ptr) : a.dyn;  // (graph replay: uploaded before the frame)
    const float ambient = a.ambient;
    const par_xent* xent = rec_.xent;
    const par_walkrec* walk = rec_.walk;
    const par_strips st = par_strips_of(tw);
    const bool has_ids = IDS && a.sprite_ids != nullptr;
    __amdgpu_buffer_rsrc_t dtab = depth_table(a.sprites, 0);
    // the depth of sprite 0's texel at byte offset t4 (anything for an offset outside the table: such a lane is outside
    // the entry's rectangle and its depth is never looked at)
    auto depth0_at = [&](uint32_t t4) -> int {
        if (true) {  // (both kernels that render tile items keep the table in LDS)
            return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(depth_lds) +
                                                     min(t4, (uint32_t)((PAR_SPRITE_TEXELS - 1) * sizeof(int32_t))));
        }
        return __builtin_amdgcn_raw_buffer_load_b32(dtab, (int)t4, 0, 0);
    };
    // the planes, addressed from the visited rectangle's corner with 32-bit pixel offsets
    const size_t corner = (size_t)(rows_lo - a.row_begin) * (size_t)W + (size_t)c0;

    for (int c = chunk0; c < chunk0 + n_chunks; c++) {  // (wave-uniform)
        const int p_first = c * 64;
        if (p_first >= area) break;
        const int pidx = p_first + lane;
        int strip, colr, rowr;
        par_strip_pixel(st, rh, pidx, strip, colr, rowr);
        const int col = c0 + colr, row = rows_lo + rowr;
        const int wj = H - row;  // world_j, alt:280 (1 <= wj <= H <= 32767)
        const int col4 = col << 2;
        const lanemask valid = __ballot(pidx < area);

        // ---- primary ray, alt:271-397: the candidate entries front to back -------------------------------------
        lanemask todo;
        if (have_masks) {
            todo = ld_uniform(rec_.cmask + c);
        } else {
            int box_r0, box_r1, box_q0, box_q1;
            tile_chunk_box(st, tw, rh, row_off, c, box_r0, box_r1, box_q0, box_q1);
            todo = __ballot(e_r0 <= box_r1) & __ballot(e_r1 > box_r0) & __ballot(e_q0 <= box_q1) &
                   __ballot(e_q1 > box_q0) & eligible;
        }
        if (DBG && (fl & (1u << 24))) todo = 0;  // bit 24: ablation (timing only), no primary pass
        lanemask done = ~valid;                  // lanes that look at no further entry (alt:372-374, or no pixel)
        lanemask adj1 = 0;                       // adjacent == 1 (alt:282, 368)
        lanemask hit_bin = 0;                    // hit_in_bin (alt:303, 365)
        int cur_bzk = -1;
        // alt:289; a lane that is done or has no pixel accepts no entry: nothing lies above INT_MAX
        int closest = lane_of(valid) ? INT_MIN : INT_MAX;
        int w_d = 0, w_pz = 0, w_ent = 0, w_t4 = 0;
        // One candidate entry: the bin bookkeeping when its bin differs from the previous candidate's (alt:298-300,
        // 368-374), the containment test (alt:310-317) and the depth comparison (alt:336-346). `x`: the entry's record,
        // `d`: its texel depth at this lane's pixel (anything for a lane outside its rectangle).
        auto candidate = [&](const u32x8& x, int dx4, int srow, uint32_t t4, int d, int sid) {
            const int bzk = (int)x[7];
            if (bzk != cur_bzk) {  // the previous visited bin is complete (alt:368-374), across the bins skipped
                const lanemask fin = adj1 & hit_bin & ~done;   // adjacent reaches 2
                if (fin) {
                    closest = lane_of(fin) ? INT_MAX : closest;
                    done |= fin;
                    if (done == ~0ull) todo = 0;  // wavefront early-out
                }
                adj1 |= hit_bin;
                if (((bzk ^ cur_bzk) >> 16) != 0) adj1 = 0;    // an empty bin lies in between (alt:298-300)
                hit_bin = 0;
                cur_bzk = bzk;
            }
            const uint32_t ex4 = x[1] & 0xFFu, eh = x[1] >> 8;
            // alt:310-317 as two unsigned range tests
            const lanemask in = __ballot((uint32_t)dx4 < ex4) & __ballot((uint32_t)srow < eh);
            const int depth = (int)x[4] + min(0, (int)x[3] + wj) - d;                      // alt:336-341
            const lanemask better = __ballot(closest < depth) & in;                        // alt:344-346
            if (lane_of(better)) {
                closest = depth;
                w_d = d;
                w_pz = (int)x[5];                                                          // alt:360-361
                w_ent = (int)x[6];                                                         // alt:363
                w_t4 = (int)(t4 + (uint32_t)sid * (uint32_t)(4 * PAR_SPRITE_TEXELS));
            }
            hit_bin |= better;                                                             // alt:365
        };
        if (PIPE && !has_ids) {
            // A frame that is alone on the chip (a small view) is as long as its slowest wavefront, and a chunk of a
            // crowded column looks at twenty entries, each a scalar load and then a depth lookup that needs it: two
            // round trips per entry, one after the other. Here they overlap: while entry i is compared, the depth of
            // entry i + 1 is on its way and the record of entry i + 2 behind it. (Ordinary loads: the compiler has to
            // see what is in flight across the loop's back edge.)
            const u32x8* xent8 = reinterpret_cast<const u32x8*>(xent);
            auto texel_of = [&](const u32x8& x, int& dx4, int& srow, uint32_t& t4) {
                dx4 = col4 - (int)x[0];
                srow = (int)x[2] - wj;                                                     // alt:324-326
                t4 = (uint32_t)(srow * (4 * PAR_SPRITE_W) + dx4);                          // alt:330-332, in bytes
            };
            int e1 = todo ? __builtin_ctzll(todo) : 0;
            lanemask rest = todo ? mask_clear(todo, e1) : 0;  // the candidates behind entry e1
            int e2 = rest ? __builtin_ctzll(rest) : e1;
            u32x8 x1 = ld_uniform(xent8 + e1);
            u32x8 x2 = ld_uniform(xent8 + e2);
            int dx4_1, srow_1;
            uint32_t t4_1;
            texel_of(x1, dx4_1, srow_1, t4_1);
            int d1 = depth0_at(t4_1);
            while (todo) {
                // this round's entry: e1 (record x1, depth d1); next: e2 (record x2 in flight or here)
                const u32x8 x = x1;
                const int dx4 = dx4_1, srow = srow_1, d = d1;
                const uint32_t t4 = t4_1;
                todo = rest;
                if (rest) {
                    rest = mask_clear(rest, e2);
                    x1 = x2;
                    e1 = e2;
                    e2 = rest ? __builtin_ctzll(rest) : e2;
                    x2 = ld_uniform(xent8 + e2);
                    texel_of(x1, dx4_1, srow_1, t4_1);
                    d1 = depth0_at(t4_1);
                }
                candidate(x, dx4, srow, t4, d, 0);
            }
        } else {
            while (todo) {
                const int e = __builtin_ctzll(todo);
                todo = mask_clear(todo, e);
                u32x8 x = uniform_fetch8(xent, (uint32_t)e * (uint32_t)sizeof(par_xent));
                uniform_wait(x);
                const int dx4 = col4 - (int)x[0];
                const int srow = (int)x[2] - wj;                                           // alt:324-326
                const uint32_t t4 = (uint32_t)(srow * (4 * PAR_SPRITE_W) + dx4);            // alt:330-332, in bytes
                int sid = 0, d;
                if (has_ids) {                                                             // alt:321-322
                    sid = ld_uniform(a.sprite_ids + (int)x[6]);
                    dtab = depth_table(a.sprites, sid);
                    // (every lane loads: outside the rectangle t4 is any number; past the table the load returns 0)
                    d = __builtin_amdgcn_raw_buffer_load_b32(dtab, (int)t4, 0, 0);
                } else {
                    d = depth0_at(t4);
                }
                candidate(x, dx4, srow, t4, d, sid);
            }
        }
        stamp(g, fl, 3, 2);  // (debug frames: the primary pass of this workgroup's first wavefront is done)
        // (a pixel was hit exactly when `closest` moved: the comparison alt:344 is strict)
        lanemask hit = __ballot(closest != INT_MIN) & valid;
        if (DBG && (fl & (1u << 26))) hit = 0;  // bit 26: ablation (timing experiments only), no shading
        if (!hit) continue;

        // ---- shading, alt:704-758 ------------------------------------------------------------------------------
        const bool is_hit = lane_of(hit);
        const int p_z = w_pz + w_d;      // alt:360-361
        const int p_y = wj - p_z;        // alt:356-359: y + z == world_j
        const int p_tex = w_t4 >> 2;
        par_texel ti = par_texel{0.f, 0.f, 0.f, 0u};
        int pal_index = PAR_PALIDX_BACKGROUND;
        float inv_x = 0.f, inv_y = 0.f, inv_z = 0.f, b_lit = 0.f;
        int sz = 0;
        if (is_hit) {
            ti = a.texinfo[p_tex];                                                          // alt:349-354
            if (a.out.palidx) {
                const int sid = IDS ? p_tex / PAR_SPRITE_TEXELS : 0;
                pal_index = a.sprites[sid].color[p_tex - sid * PAR_SPRITE_TEXELS];
            }
            // towards_light = normalize_L1(light - world), alt:711-715 + spr:28-35
            const float dx = (float)(dyn.lx - col), dy = (float)(dyn.ly - p_y), dz = (float)(dyn.lz - p_z);
            float tx, tyy, tz;
            normalize_l1_and_inverse(dx, dy, dz, tx, tyy, tz, inv_x, inv_y, inv_z);       // alt:711-719
            const float dot = ti.nx * tx + ti.ny * tyy + ti.nz * tz;                        // alt:746-747
            const float diffuse = std_max(0.f, dot);                                        // alt:745
            b_lit = std_min(1.f, diffuse + ambient);                                        // alt:758
            sz = div_bin(p_z, a.magic_b);                                                   // alt:727
        }
        // shadow ray, alt:738-742: the lanes grouped by start bin (bx, by, sz); the bin's walk list (columns kernel)
        // through scalar loads. The start bin's row is the column's own (y + z == world_j, alt:725-726).
        const float fox = (float)(int)(int16_t)col, foy = (float)(int)(int16_t)p_y, foz = (float)(int)(int16_t)p_z;  // alt:720-722
        stamp(g, fl, 3, 3);  // (... shading up to the shadow test)
        lanemask lit = hit;
        lanemask pending = hit;
        // What a group's inverse directions have in common picks its loop (walk_list_lit's SIGNS). Per lane: the sign
        // bits of the three inverses, or 9 when one of them is not finite (their product is finite exactly when all
        // three are: each is 1 / n with |n| <= 1, so none is below 1 in magnitude and the product cannot overflow).
        const float inv_prod = inv_x * inv_y * inv_z;
        int sign_code = (int)((__float_as_uint(inv_x) >> 31) | ((__float_as_uint(inv_y) >> 31) << 1) |
                              ((__float_as_uint(inv_z) >> 31) << 2));
        sign_code = __builtin_isfinite(inv_prod) ? sign_code : 9;
        while (pending) {
            const int s_bin = __builtin_amdgcn_readlane(sz, __builtin_ctzll(pending));
            const lanemask grp = __ballot(sz == s_bin) & pending;
            pending &= ~grp;
            const lanemask m = __ballot(nb_bz == s_bin) & nb_lanes;
            int woff = 0, wcnt = -1;
            if (m) {
                const uint32_t wd = (uint32_t)__builtin_amdgcn_readlane((int)nb.y, __builtin_ctzll(m));
                woff = (int)(wd & 0xFFFFu);
                wcnt = (int)(int16_t)(wd >> 16);  // -1: the walk was too long to record
            }
            if (wcnt == 0) continue;  // nothing on the way to the light
            lanemask alive = grp;
            if (wcnt < 0) {
                // a start bin that holds no primitive (negative world z, sprite depths outside the box) or whose walk
                // was too long to record: trace_hash_for_light as written, per lane
                bool l = true;
                if (lane_of(grp)) {
                    l = lane_shadow_walk(g, a.count, a.slots, bx, by, s_bin, dyn, w_ent, (int)(int16_t)col,
                                         (int)(int16_t)p_y, (int)(int16_t)p_z, inv_x, inv_y, inv_z);
                }
                alive = __ballot(l) & grp;
            } else {
                // what the group's inverse directions have in common picks the loop (wave-uniform); every lane
                // computes (uniform control flow: the loop counter stays a scalar), the group's lanes count
                const par_walkrec* wr = walk + woff;
                int signs = __builtin_amdgcn_readlane(sign_code, __builtin_ctzll(grp));
                const lanemask same = __ballot(sign_code == signs) & grp;
                if (same != grp) signs = (__ballot(sign_code == 9) & grp) ? 9 : 8;
                switch (signs) {
                    case 0: alive = walk_list_lit<0>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 1: alive = walk_list_lit<1>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 2: alive = walk_list_lit<2>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 3: alive = walk_list_lit<3>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 4: alive = walk_list_lit<4>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 5: alive = walk_list_lit<5>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 6: alive = walk_list_lit<6>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 7: alive = walk_list_lit<7>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    case 8: alive = walk_list_lit<8>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                    default: alive = walk_list_lit<9>(wr, wcnt, grp, fox, foy, foz, inv_x, inv_y, inv_z, w_ent); break;
                }
            }
            lit &= ~(grp & ~alive);
        }
        stamp(g, fl, 3, 4);  // (... the shadow test)
        // ---- quantise + store, alt:735, 757-758 ----------------------------------------------------------------
        if (DBG && (fl & PAR_RENDER_COUNT_RAYS) && a.ray_counter) {
            if (lane == 0) atomicAdd(a.ray_counter, (unsigned long long)__popcll(hit));
        }
        const bool lit_px = lane_of(lit);
        const float bright = lit_px ? b_lit : ambient;
        if (DBG && (fl & (1u << 25))) {  // bit 25: ablation (timing only), no stores; keep the values alive
            asm volatile("" ::"v"(ti.rgba), "v"(bright), "v"(pal_index));
        } else if (is_hit) {  // (uncovered pixels keep what the fill wrote)
            const uint32_t o = (uint32_t)(rowr * W + colr);  // (below 2^23: a tile has at most 160 rows)
            if (a.out.fb) __builtin_nontemporal_store(color_scale(ti.rgba, bright), reinterpret_cast<uint32_t*>(a.out.fb) + corner + o);
            if (a.out.palidx) __builtin_nontemporal_store((uint8_t)pal_index, a.out.palidx + corner + o);
            if (FULL && a.out.brightness) a.out.brightness[corner + o] = bright;
            if (FULL && a.out.lit) a.out.lit[corner + o] = lit_px ? 1 : 0;
            if (FULL && a.out.gbuf) {
                par_pixel pxl;
                pxl.normal = par_vec3{ti.nx, ti.ny, ti.nz};
                pxl.color.red = (uint8_t)(ti.rgba & 0xFF);
                pxl.color.green = (uint8_t)((ti.rgba >> 8) & 0xFF);
                pxl.color.blue = (uint8_t)((ti.rgba >> 16) & 0xFF);
                pxl.color.alpha = (uint8_t)(ti.rgba >> 24);
                pxl.y = p_y;
                pxl.z = p_z;
                pxl.entity_index = w_ent;
                a.out.gbuf[corner + o] = pxl;
            }
        }
    }
}

// Wavefront `w` of `n_waves` (a multiple of PAR_ITEM_SHARDS * PAR_WAVE_NW): the PAR_WAVE_NW wavefronts of a
// workgroup take CONSECUTIVE items of one shard -- a column's items follow each other in its shard, so the
// wavefronts that share a CU's scalar cache and L1 mostly read the same column record at the same time (with each
// wavefront on a shard of its own a third of the scalar loads of a dense frame missed: 24 columns' records per CU
// against 16 KB) -- workgroup v takes shard v mod shards, there the items (v / shards) * NW + its wavefront's
// number, then on by n_waves / shards. The launch offers one wavefront per item of the host's bound (or per few),
// so the loop runs once or a few times.
template <bool DBG, bool IDS, bool FULL, bool TILES, bool PIPE = false>
__device__ __forceinline__ void render_items(const par_grid_dev& g, const par_render_args& a, int w, int n_waves,
                                             const int32_t* depth_lds = nullptr) {
    const int lane = (int)threadIdx.x & 63;
    const int wg = w / PAR_WAVE_NW;
    const int shard = wg & (PAR_ITEM_SHARDS - 1);
    const int list_shard = (TILES ? PAR_ITEM_SHARDS : 0) + shard;
    const par_item* list = g.items + (size_t)list_shard * g.item_capacity;
    const int first = (wg >> PAR_ITEM_SHARD_BITS) * PAR_WAVE_NW + (w % PAR_WAVE_NW);
    // the first item is fetched beside the counter (the list is allocated whatever the counter says)
    u32x8 it = item_fetch(list + min(first, g.item_capacity - 1));
    const int n = min(ld_uniform(g.item_counters + list_shard * PAR_ITEM_COUNTER_STRIDE), g.item_capacity);
    item_arrived(it);
    stamp(g, DBG ? a.flags : 0u, 3, 1);
    for (int i = first; i < n;) {
        if (it[0] != PAR_ITEM_NONE) {
            if (TILES) {
                render_tile_item<DBG, IDS, FULL, PIPE>(g, a, make_uint4(it[0], it[1], it[2], it[3]), lane, depth_lds);
            } else {
                render_item<DBG, IDS, FULL>(g, a, make_uint4(it[0], it[1], it[2], it[3]), make_uint4(it[4], it[5], it[6], it[7]), lane);
            }
        }
        i += n_waves >> PAR_ITEM_SHARD_BITS;
        if (i < n) {
            it = item_fetch(list + i);
            item_arrived(it);
        }
    }
}

#if defined(PAR_RENDER_SGPRS)  // (experiments: cap the entry kernel's scalar registers, tools/debug/variants.sh)
#define PAR_RENDER_ATTR __attribute__((amdgpu_num_sgpr(PAR_RENDER_SGPRS)))
#else
#define PAR_RENDER_ATTR
#endif
template <bool DBG, bool IDS, bool FULL>
__global__ __launch_bounds__(PAR_WAVE_NW * 64) PAR_RENDER_ATTR void render_items_kernel(par_grid_dev g, par_render_args a) {
    stamp(g, DBG ? a.flags : 0u, 3, 0);
    const unsigned long long core0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
    __builtin_amdgcn_s_setprio(3);  // latency-bound wavefronts go before the streaming fill's when both want to issue
    const int w = __builtin_amdgcn_readfirstlane((int)blockIdx.x * PAR_WAVE_NW + ((int)threadIdx.x >> 6));
    render_items<DBG, IDS, FULL, false>(g, a, w, (int)gridDim.x * PAR_WAVE_NW);
    stamp(g, DBG ? a.flags : 0u, 3, 7);
    if (DBG && g.stamps && (a.flags & (1u << 29)) && threadIdx.x == 0 && blockIdx.x < PAR_STAMP_WGS) {
        // slot 5: the wavefront's life in shader-clock cycles (s_memtime), beside slots 0 / 7 in 100 MHz ticks
        g.stamps[((size_t)3 * PAR_STAMP_WGS + blockIdx.x) * PAR_STAMP_SLOTS + 5] = __builtin_amdgcn_s_memtime() - core0;
    }
}

// The work items of the columns visited as whole tiles (render_tile_item): a kernel of its own, launched for DENSE
// frames only (the host decides, par_render_args::tile_k). In one kernel with the entry passes either path would pay
// for the other's registers -- the entry passes of a sparse frame are bound by latency and want every wavefront slot,
// the tile pass is bound by instruction issue and wants its scalars in registers.
// Sprite 0's depth table into LDS, by the whole workgroup (a barrier: every thread of it must call).
__device__ __forceinline__ void tile_depth_to_lds(const par_render_args& a, int32_t* depth_lds) {
    static_assert(PAR_SPRITE_TEXELS % 4 == 0, "copied in 16-byte pieces");
    const uint4* src = reinterpret_cast<const uint4*>(a.sprites[0].depth);
    uint4* dst = reinterpret_cast<uint4*>(depth_lds);
    for (int i = (int)threadIdx.x; i < PAR_SPRITE_TEXELS / 4; i += (int)blockDim.x) dst[i] = src[i];
    __syncthreads();
}

template <bool DBG, bool IDS, bool FULL>
__global__ __launch_bounds__(PAR_WAVE_NW * 64) void render_tiles_kernel(par_grid_dev g, par_render_args a) {
    __shared__ __attribute__((aligned(16))) int32_t depth_lds[PAR_SPRITE_TEXELS];
    tile_depth_to_lds(a, depth_lds);
    stamp(g, DBG ? a.flags : 0u, 5, 0);
    const unsigned long long core0 = DBG ? __builtin_amdgcn_s_memtime() : 0ull;
    const int w = __builtin_amdgcn_readfirstlane((int)blockIdx.x * PAR_WAVE_NW + ((int)threadIdx.x >> 6));
    render_items<DBG, IDS, FULL, true>(g, a, w, (int)gridDim.x * PAR_WAVE_NW, depth_lds);
    stamp(g, DBG ? a.flags : 0u, 5, 7);
    if (DBG && g.stamps && (a.flags & (1u << 29)) && threadIdx.x == 0 && blockIdx.x < PAR_STAMP_WGS) {
        g.stamps[((size_t)5 * PAR_STAMP_WGS + blockIdx.x) * PAR_STAMP_SLOTS + 5] = __builtin_amdgcn_s_memtime() - core0;
    }
}

// The columns that overflowed their record (columns_kernel lists them), or every column when a.dense
// (PAR_FORCE_GENERIC=1, tests): the primary pass reads the column's bins straight from the hash, as the reference
// does, and the shadow walks are done here, once per wavefront and distinct start bin. A rare path: the launch is
// small and its workgroups leave at once when the list is empty.
__global__ __launch_bounds__(PAR_WAVE_NW * 64) void render_overflow_kernel(par_grid_dev g, par_render_args a) {
    __shared__ WaveScratch scratch[PAR_WAVE_NW];
    stamp(g, a.flags, 4, 0);
    const int n_cols = min(g.counters[PAR_CNT_COLS], g.col_capacity);
    if (a.dense) {
        for (int ci = (int)blockIdx.x; ci < n_cols; ci += (int)gridDim.x) {
            render_column_generic(g, a, ci, (int)blockIdx.y, (int)gridDim.y, scratch);
        }
    } else {
        const int n_slow = g.counters[PAR_CNT_SLOW];
        for (int s = (int)blockIdx.x; s < n_slow; s += (int)gridDim.x) {
            render_column_generic(g, a, g.slow_list[s], (int)blockIdx.y, (int)gridDim.y, scratch);
        }
    }
    stamp(g, a.flags, 4, 7);
}

// All of the above in one launch, for small frames: there a frame is bound by its launches (the host enqueues one in
// about 3 us, the device needs about 1.5 us between two), not by the registers the paths cost each other.
// Workgroups [0, n_item_wgs) render the entry items, [n_item_wgs, n_item_wgs + n_tile_wgs) the tile items, the rest
// (groups of `over_parts`) the overflow list.
template <bool DBG>
__global__ __launch_bounds__(PAR_WAVE_NW * 64) void render_both_kernel(par_grid_dev g, par_render_args a,
                                                                          int n_item_wgs, int n_tile_wgs, int over_parts) {
    __shared__ WaveScratch scratch[PAR_WAVE_NW];
    __shared__ __attribute__((aligned(16))) int32_t depth_lds[PAR_SPRITE_TEXELS];
    const int b = (int)blockIdx.x;
    // (time stamps of debug frames: row 3, a workgroup's start and end, whichever list it serves)
    if (b < n_item_wgs) {
        stamp(g, DBG ? a.flags : 0u, 3, 0);
        const int w = __builtin_amdgcn_readfirstlane(b * PAR_WAVE_NW + ((int)threadIdx.x >> 6));
        render_items<DBG, true, true, false>(g, a, w, n_item_wgs * PAR_WAVE_NW);
        stamp(g, DBG ? a.flags : 0u, 3, 7);
        return;
    }
    if (b < n_item_wgs + n_tile_wgs) {
        stamp(g, DBG ? a.flags : 0u, 3, 0);
        const int w = __builtin_amdgcn_readfirstlane((b - n_item_wgs) * PAR_WAVE_NW + ((int)threadIdx.x >> 6));
        tile_depth_to_lds(a, depth_lds);
        render_items<DBG, true, true, true, true>(g, a, w, n_tile_wgs * PAR_WAVE_NW, depth_lds);
        stamp(g, DBG ? a.flags : 0u, 3, 7);
        return;
    }
    const int j = b - n_item_wgs - n_tile_wgs;  // (workgroups of this kind exist only when some column may overflow)
    const int stride = ((int)gridDim.x - n_item_wgs - n_tile_wgs) / over_parts;
    const int n_slow = g.counters[PAR_CNT_SLOW];
    for (int s = j / over_parts; s < n_slow; s += stride) {
        render_column_generic(g, a, g.slow_list[s], j % over_parts, over_parts, scratch);
    }
}

// Test hook (par_debug_units): the device functions of the reference's three arithmetic units on caller-supplied
// vectors, one element per thread. kind 0: AABB::intersect alt:40-83 (a: par_aabb, b: {float inv[3]; int16 origin[3]}
// -> u8 hit; 3, 4: the same on a walk record, as the render kernel tests it); 1: Color::operator* spr:8-16 (a: float r, g, b, a, v -> u8[4]); 2: Vector::normalize spr:28-35
// (a: float[3] -> float[3]).
struct unit_ray {
    float inv_x, inv_y, inv_z;
    int16_t ox, oy, oz, pad;
};
__global__ __launch_bounds__(256) void units_kernel(int kind, const void* in_a, const void* in_b, int n, void* out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    if (kind == 0) {
        const par_aabb box = static_cast<const par_aabb*>(in_a)[i];
        const unit_ray ray = static_cast<const unit_ray*>(in_b)[i];
        par_slot rec;
        rec.px = box.px; rec.py = box.py; rec.pz = box.pz;
        rec.ex = box.ex; rec.ey = box.ey; rec.ez = box.ez;
        rec.entity = 0;
        static_cast<uint8_t*>(out)[i] = slab_hit(rec, ray.ox, ray.oy, ray.oz, ray.inv_x, ray.inv_y, ray.inv_z) ? 1 : 0;
    } else if (kind == 3 || kind == 4) {
        // the same test as the render kernel runs it: on a walk record, through the hardware min / max when the
        // inverse direction is finite (kind 3; the kernel's choice, made per element here) or never (kind 4)
        const par_aabb box = static_cast<const par_aabb*>(in_a)[i];
        const unit_ray ray = static_cast<const unit_ray*>(in_b)[i];
        par_slot rec;
        rec.px = box.px; rec.py = box.py; rec.pz = box.pz;
        rec.ex = box.ex; rec.ey = box.ey; rec.ez = box.ez;
        rec.entity = 0;
        const par_walkrec w = walkrec_of(rec);
        const v2f rx = {w.x_lo, w.x_hi}, ry = {w.y_lo, w.y_hi}, rz = {w.z_lo, w.z_hi};
        const float fox = (float)ray.ox, foy = (float)ray.oy, foz = (float)ray.oz;
        const bool finite = kind == 3 && __builtin_isfinite(ray.inv_x) && __builtin_isfinite(ray.inv_y) &&
                            __builtin_isfinite(ray.inv_z);
        const bool hit = finite ? slab_hit_rec<true>(rx, ry, rz, fox, foy, foz, ray.inv_x, ray.inv_y, ray.inv_z)
                                : slab_hit_rec<false>(rx, ry, rz, fox, foy, foz, ray.inv_x, ray.inv_y, ray.inv_z);
        static_cast<uint8_t*>(out)[i] = hit ? 1 : 0;
    } else if (kind == 1) {
        const float* v = static_cast<const float*>(in_a) + (size_t)i * 5;
        const uint32_t c = (uint32_t)(uint8_t)v[0] | ((uint32_t)(uint8_t)v[1] << 8) | ((uint32_t)(uint8_t)v[2] << 16) |
                           ((uint32_t)(uint8_t)v[3] << 24);
        static_cast<uint32_t*>(out)[i] = color_scale(c, v[4]);
    } else {
        const float* v = static_cast<const float*>(in_a) + (size_t)i * 3;
        float* o = static_cast<float*>(out) + (size_t)i * 3;
        float ix, iy, iz;  // (the same function the shading uses: short sequences where they are exact)
        normalize_l1_and_inverse<true>(v[0], v[1], v[2], o[0], o[1], o[2], ix, iy, iz);
    }
}

}  // namespace

// The fill rides along with the first three launches when it needs only the streaming kernel (frame and
// palette-index planes, aligned): shares of its 512-pixel chunks in proportion to what those launches take anyway
// (insert 5.6 us, resolve 5.6 us, column records 17.6 us at 4096^2 / 1024 primitives; the fill writes about 5 MB per
// microsecond). A lit plane is filled afterwards (it needs the background rays, bgline_kernel). Returns false when
// the whole fill has to be launched on its own (par_launch_fill).
bool par_plan_fill(const par_render_args& a, par_fill_plan* plan) {
    const bool fb_fast = a.out.fb && (a.W % 8 == 0) && ((uintptr_t)a.out.fb % 16 == 0);
    const bool pal_fast = !a.out.palidx || ((a.W % 8 == 0) && ((uintptr_t)a.out.palidx % 8 == 0));
    if (!fb_fast || !pal_fast || a.out.brightness || a.out.gbuf) return false;
    const uint32_t ch = (uint32_t)(uint8_t)((float)a.background * a.ambient);  // Color{127,127,127,0} * ambient
    plan->out_rgba = ch | (ch << 8) | (ch << 16);
    const int64_t chunks = (int64_t)(a.row_end - a.row_begin) * ((a.W + 511) / 512);
    if (chunks > 0x7FFFFFFF) return false;
    plan->cut[0] = 0;
    // (Measured at 4096^2, three frames in flight: 0/0/100 % 36.5 us per frame, 10/10/80 35.7, 20/20/60 34.1,
    // 25/25/50 34.5, 33/33/33 34.0.)
    // (PAR_TUNE_FILL_BUILD_PCT: the hash build's share in percent, tools/sweep.py; default 40, half of it per launch
    // when the build takes two)
    static const int build_pct = [] {
        const char* e = std::getenv("PAR_TUNE_FILL_BUILD_PCT");
        const int v = e ? std::atoi(e) : 40;
        return v < 0 ? 0 : (v > 100 ? 100 : v);
    }();
    plan->cut[1] = (int)(chunks * build_pct / 200);
    plan->cut[2] = (int)(chunks * build_pct / 100);
    plan->cut[3] = (int)chunks;
    return true;
}

// fill workgroups (of `waves` wavefronts) for the chunks [cut[i], cut[i+1]): one chunk per wavefront and iteration.
// Few wavefronts are enough to write at full rate, and every resident fill wavefront is a slot another frame's
// kernels cannot use. Measured at 4096^2 (three frames in flight / one), workgroups in the insert and resolve
// launches / in the column launch: 256 / 1024: 34.6 / 59.5 us, 128 / 512: 32.7 / 58.3, 64 / 256: 31.5 / 58.1,
// 32 / 128: 33.1 / 66.8, 16 / 64: 40.7 / 94.7.
// (PAR_TUNE_FILL_WGS overrides the 64, tools/sweep.py)
static const int PAR_FILL_RIDE_WGS = [] {
    const char* e = std::getenv("PAR_TUNE_FILL_WGS");
    const int v = e ? std::atoi(e) : 64;
    return v < 1 ? 1 : (v > 4096 ? 4096 : v);
}();
static int64_t fill_blocks(const par_fill_plan& p, int i, int waves, int64_t cap) {
    const int64_t chunks = p.cut[i + 1] - p.cut[i];
    int64_t n = (chunks + waves - 1) / waves;
    if (n > cap) n = cap;
    return n < 0 ? 0 : n;
}

hipError_t par_launch_bin_insert(const par_grid_dev& g, const par_bin_args& a, const par_render_args* fa,
                                 const par_fill_plan* fill, hipStream_t stream) {
    // ENT entities per wavefront; the wipe of the previous frame's nodes is a grid-stride loop over <= capacity
    const bool small = a.n <= 16384;
    int64_t work = (int64_t)a.n * (small ? 4 : 1);  // threads = waves * 64 = n / ENT * 64
    if (work < 16384) work = 16384;                 // (the wipe loop strides: any grid covers any node count)
    int64_t blocks = (work + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (fill) {
        const int64_t nf = fill_blocks(*fill, 0, 4, PAR_FILL_RIDE_WGS);
        const int2 part = make_int2(fill->cut[0], fill->cut[1]);
        if (small) {
            hipLaunchKernelGGL(insert_fill_kernel<16>, dim3((unsigned)(blocks + nf)), dim3(256), 0, stream, g, a, *fa,
                               fill->out_rgba, (int)blocks, part);
        } else {
            hipLaunchKernelGGL(insert_fill_kernel<64>, dim3((unsigned)(blocks + nf)), dim3(256), 0, stream, g, a, *fa,
                               fill->out_rgba, (int)blocks, part);
        }
    } else if (small) {
        hipLaunchKernelGGL(bin_insert_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, stream, g, a);
    } else {
        hipLaunchKernelGGL(bin_insert_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, stream, g, a);
    }
    return hipGetLastError();
}

// The whole hash build in one launch (see build_fill_kernel) when the scene is small enough for a handful of
// workgroups; hipErrorNotSupported (nothing launched) otherwise. With `fill`: also the fill shares of both launches.
hipError_t par_launch_build(const par_grid_dev& g, const par_bin_args& a, int64_t pair_bound, const par_render_args* fa,
                            const par_fill_plan* fill, hipStream_t stream) {
    if (a.n > 16384 || pair_bound > 65536) return hipErrorNotSupported;
    int64_t work = (int64_t)a.n * 4;  // ENT = 16 entities per wavefront: threads = n / 16 * 64
    if (work < pair_bound) work = pair_bound;
    int64_t blocks = (work + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;  // (insert, the wipe and resolve all stride)
    par_render_args none{};
    int64_t nf = 0;
    int2 part = make_int2(0, 0);
    if (fill) {  // the chunks [cut[0], cut[2]): one chunk per wavefront and iteration, 4 wavefronts per workgroup
        nf = ((int64_t)fill->cut[2] - fill->cut[0] + 3) / 4;
        if (nf > 2 * PAR_FILL_RIDE_WGS) nf = 2 * PAR_FILL_RIDE_WGS;
        part = make_int2(fill->cut[0], fill->cut[2]);
    }
    hipLaunchKernelGGL(build_fill_kernel<16>, dim3((unsigned)(blocks + nf)), dim3(256), 0, stream, g, a,
                       fill ? *fa : none, fill ? fill->out_rgba : 0u, (int)blocks, part);
    return hipGetLastError();
}

hipError_t par_launch_bin_resolve(const par_grid_dev& g, const par_bin_args& a, int64_t pair_bound,
                                  const par_render_args* fa, const par_fill_plan* fill, hipStream_t stream) {
    int64_t blocks = (pair_bound + 255) / 256;
    if (blocks < 1) blocks = 1;  // block 0 always runs: it resets the other set's node counter
    if (fill) {
        const int64_t nf = fill_blocks(*fill, 1, 4, PAR_FILL_RIDE_WGS);
        hipLaunchKernelGGL(resolve_fill_kernel, dim3((unsigned)(blocks + nf)), dim3(256), 0, stream, g, a, *fa,
                           fill->out_rgba, (int)blocks, make_int2(fill->cut[1], fill->cut[2]));
    } else {
        hipLaunchKernelGGL(bin_resolve_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, a);
    }
    return hipGetLastError();
}

// Columns of the launch: the occupied columns (at most `column_bound`, at most those of the rendered rows) and, when
// background rays are traced, one background walk per bin column; `per_block` of them per workgroup.
static int64_t column_blocks(const par_grid_dev& g, const par_render_args& a, int64_t column_bound, int per_block,
                             int64_t* n_cols) {
    const int64_t cols_in_range = (int64_t)g.gx * (a.by_hi - a.by_lo + 1);
    int64_t n = column_bound < cols_in_range ? column_bound : cols_in_range;
    if (n < 0) n = 0;
    *n_cols = n;
    if (a.trace_bg) n += g.gx;  // the background walks
    return (n + per_block - 1) / per_block;
}

// One wavefront per column instead of two: for a frame among several in flight (PAR_RENDER_PIPELINED) that has
// enough columns to fill the chip anyway. (4096^2 / 1024 primitives, four in flight: 23.8 against 24.1 us per frame;
// full floor 330 against 346 us; the 480x320 graybox scene, 94 columns, 13.5 against 12.0 us: few columns need the
// second wavefront for their walks even then.)
static bool one_wave_per_column(const par_grid_dev& g, const par_render_args& a, int64_t column_bound) {
    static const bool off = [] { const char* e = std::getenv("PAR_TUNE_NO_PIPELINED"); return e && e[0] == '1'; }();
    const int64_t cols_in_range = (int64_t)g.gx * (a.by_hi - a.by_lo + 1);
    return !off && (a.flags & PAR_RENDER_PIPELINED) != 0 && std::min(column_bound, cols_in_range) >= 1024;
}

// Wavefronts per column (columns_wave's ROLES): one for a frame among several in flight that fills the chip anyway;
// otherwise the fewer columns a frame has, the more wavefronts share a column's walks -- a frame with a hundred columns
// leaves the chip empty, and its columns' walks, one after the other in one or two wavefronts, are most of its
// column launch (the 480x320 graybox world alone: 13.4 us with two wavefronts per column). PAR_TUNE_COL_ROLES
// overrides (1, 2, 4 or 8; tools).
static int column_roles(const par_grid_dev& g, const par_render_args& a, int64_t column_bound) {
    static const int tuned = [] {
        const char* e = std::getenv("PAR_TUNE_COL_ROLES");
        const int v = e ? std::atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0;
    }();
    if (tuned) return tuned;
    if (one_wave_per_column(g, a, column_bound)) return 1;
    const int64_t cols_in_range = (int64_t)g.gx * (a.by_hi - a.by_lo + 1);
    const int64_t cols = std::min(column_bound, cols_in_range);
    // (measured alone / four in flight, us per frame, 2 -> 4 -> 8 wavefronts per column: graybox 43.4 / 12.3 -> 40.4 /
    // 11.5 -> 40.2 / 11.5; 512^2 with 64 primitives 26.3 / 7.0 -> 23.8 / 6.9 -> 23.7 / 6.9; 1024^2 with 512: 40.8 /
    // 14.5 -> 39.2 / 14.6 -> 36.7 / 16.1; 4096^2 with 1 024: 42.5 / 23.5 -> 47.1 / 26.5 -> 59.1 / 36.5)
    const bool in_flight = (a.flags & PAR_RENDER_PIPELINED) != 0;
    if (cols <= 256) return 8;
    if (cols <= 1024) return in_flight ? 4 : 8;
    return 2;
}

template <int ROLES>
static hipError_t launch_columns_as(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                    const par_fill_plan* fill, hipStream_t stream) {
    constexpr int waves = col_waves<ROLES>();
    int64_t n_cols;
    const int64_t n_col_blocks = column_blocks(g, a, column_bound, ROLES == 1 ? waves : 1, &n_cols);
    if (fill) {
        // (the same number of fill WAVEFRONTS whatever the workgroup size)
        const int64_t n_fill = fill_blocks(*fill, 2, waves, 4 * PAR_FILL_RIDE_WGS * 2 / waves);
        if (n_col_blocks + n_fill <= 0) return hipSuccess;
        hipLaunchKernelGGL(columns_fill_kernel<ROLES>, dim3((unsigned)(n_col_blocks + n_fill)), dim3(waves * 64), 0, stream,
                           g, a, fill->out_rgba, (int)n_col_blocks, (int)n_cols, make_int2(fill->cut[2], fill->cut[3]));
    } else {
        if (n_col_blocks <= 0) return hipSuccess;
        hipLaunchKernelGGL(columns_kernel<ROLES>, dim3((unsigned)n_col_blocks), dim3(waves * 64), 0, stream, g, a, (int)n_cols);
    }
    return hipGetLastError();
}

static hipError_t launch_columns(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                 const par_fill_plan* fill, hipStream_t stream) {
    hipError_t e;
    switch (column_roles(g, a, column_bound)) {
        case 1: e = launch_columns_as<1>(g, a, column_bound, fill, stream); break;
        case 4: e = launch_columns_as<4>(g, a, column_bound, fill, stream); break;
        case 8: e = launch_columns_as<8>(g, a, column_bound, fill, stream); break;
        default: e = launch_columns_as<2>(g, a, column_bound, fill, stream); break;
    }
    if (e != hipSuccess || !a.trace_bg) return e;
    hipLaunchKernelGGL(bgline_kernel, dim3((unsigned)((a.W + 255) / 256)), dim3(256), 0, stream, g, a);
    return hipGetLastError();
}

hipError_t par_launch_columns(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                              hipStream_t stream) {
    return launch_columns(g, a, column_bound, nullptr, stream);
}

// Column records + the last share of the fill in one launch.
hipError_t par_launch_columns_fill(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                   const par_fill_plan& fill, hipStream_t stream) {
    return launch_columns(g, a, column_bound, &fill, stream);
}

hipError_t par_launch_fill(const par_grid_dev& g, const par_render_args& a, hipStream_t stream) {
    // Color{127,127,127,0} * ambient, spr:8-16 (same truncation on the host)
    const uint32_t ch = (uint32_t)(uint8_t)((float)a.background * a.ambient);
    const uint32_t out_rgba = ch | (ch << 8) | (ch << 16);
    const int64_t npix = (int64_t)(a.row_end - a.row_begin) * a.W;
    const bool fb_fast = a.out.fb && (a.W % 8 == 0) && ((uintptr_t)a.out.fb % 16 == 0);
    const bool pal_fast = a.out.palidx && (a.W % 8 == 0) && ((uintptr_t)a.out.palidx % 8 == 0);
    const bool lit_fast = a.out.lit && (a.W % 8 == 0) && ((uintptr_t)a.out.lit % 8 == 0);
    if (fb_fast || pal_fast || lit_fast) {
        par_render_args f = a;
        if (!fb_fast) f.out.fb = nullptr;
        if (!pal_fast) f.out.palidx = nullptr;
        if (!lit_fast) f.out.lit = nullptr;
        const int64_t chunks = (int64_t)(a.row_end - a.row_begin) * ((a.W + 511) / 512);
        int64_t blocks = (chunks + 3) / 4;  // 4 wavefronts per block, one 512-pixel chunk each per iteration
        // 2048 blocks (8 per CU) already run at full bandwidth (measured: 256 do); more only take wave slots from
        // the other frames in flight
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, f, out_rgba, g.bglit);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const bool need_generic = (a.out.fb && !fb_fast) || (a.out.palidx && !pal_fast) || (a.out.lit && !lit_fast) ||
                              a.out.brightness || a.out.gbuf;
    if (need_generic) {
        int64_t blocks = (npix + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(fill_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, out_rgba,
                           fb_fast ? 0 : 1, pal_fast ? 0 : 1, lit_fast ? 0 : 1, g.bglit);
        return hipGetLastError();
    }
    return hipSuccess;
}

// One wavefront per work item of the host's bound, or per few of them (rounded up to whole workgroups and to a
// multiple of the shard count, which render_items' item-to-wavefront mapping needs).
static int64_t item_workgroups(int64_t item_bound) {
    const int64_t unit = (int64_t)PAR_ITEM_SHARDS * PAR_WAVE_NW;
    // Frames with very many items (dense scenes: 260 000 at 4096^2) are rendered by wavefronts that take several items
    // one after the other: launching a wavefront costs the chip more than its loop's extra iteration (full floor:
    // 382 us with one item per wavefront, 353 with 8, 347 with 16, 345 with 32), while a frame with few items needs
    // every wavefront it can get (480x320 graybox, 2 400 items: 11.8 us with one, 16.7 with 4). PAR_TUNE_ITEMS_PER_WAVE
    // overrides (tools).
    static const int tuned = [] {
        const char* e = std::getenv("PAR_TUNE_ITEMS_PER_WAVE");
        const int v = e ? std::atoi(e) : 0;
        return v < 0 ? 0 : (v > 64 ? 64 : v);
    }();
    int64_t per_wave = tuned > 0 ? tuned : item_bound / 16384;
    if (per_wave < 1) per_wave = 1;
    if (per_wave > 16) per_wave = 16;
    item_bound = (item_bound + per_wave - 1) / per_wave;
    // (PAR_TUNE_ITEM_WAVES_PCT, tools: more wavefronts than the bound asks for, in percent: a shard with more items than
    // its share of the wavefronts makes some of them take two)
    static const int pct = [] {
        const char* e = std::getenv("PAR_TUNE_ITEM_WAVES_PCT");
        const int v = e ? std::atoi(e) : 100;
        return v < 50 ? 50 : (v > 400 ? 400 : v);
    }();
    item_bound = item_bound * pct / 100;
    int64_t waves = (item_bound + unit - 1) / unit * unit;
    if (waves < unit) waves = unit;
    if (waves > (int64_t)1 << 24) waves = (int64_t)1 << 24;  // (the wavefronts then loop over their shard)
    return waves / PAR_WAVE_NW;
}

hipError_t par_launch_render(const par_grid_dev& g, const par_render_args& a, int64_t item_bound,
                             hipStream_t stream) {
    if (item_bound <= 0 || a.dense) return hipSuccess;
    // In a dense frame (a.tile_k > 0) most of the bound's chunks become tile items of the other kernel: the entry
    // kernel gets a modest grid whose wavefronts loop over their shards (16 640 wavefronts that found no item cost a
    // full-floor frame 10 us of kernel time).
    int64_t wgs = item_workgroups(item_bound);
    if (a.tile_k > 0 && wgs > 1024) wgs = 1024;
    const dim3 grid((unsigned)wgs), block(PAR_WAVE_NW * 64);
    // (experiments: PAR_EXP_RENDER_LDS bytes of unused LDS per workgroup cap the workgroups per CU, i.e. the wavefront
    // slots the entry kernel can hold: how much of the frame time is wavefront-slot time?)
    static const unsigned lds = [] { const char* e = std::getenv("PAR_EXP_RENDER_LDS"); return e ? (unsigned)std::atoi(e) : 0u; }();
    if (a.flags & PAR_DEBUG_FLAGS) {
        hipLaunchKernelGGL((render_items_kernel<true, true, true>), grid, block, lds, stream, g, a);
    } else if (a.sprite_ids || a.out.brightness || a.out.lit || a.out.gbuf) {
        hipLaunchKernelGGL((render_items_kernel<false, true, true>), grid, block, lds, stream, g, a);
    } else {  // every entity uses sprite 0 (the reference's own scenes), RGBA + palette index only
        hipLaunchKernelGGL((render_items_kernel<false, false, false>), grid, block, lds, stream, g, a);
    }
    return hipGetLastError();
}

// Tile items cover a.tile_k chunks each: `item_bound` chunks are at most that many fewer items (rounded up per column,
// which the bound's slack of one item per column covers).
static int64_t tile_item_bound(const par_render_args& a, int64_t item_bound) {
    const int64_t k = a.tile_k > 0 ? a.tile_k : 1;
    return (item_bound + k - 1) / k + 1;
}

hipError_t par_launch_render_tiles(const par_grid_dev& g, const par_render_args& a, int64_t item_bound,
                                   hipStream_t stream) {
    if (item_bound <= 0 || a.dense || a.tile_k <= 0) return hipSuccess;
    // (several items per wavefront in big frames, item_workgroups: full floor at 4096^2, tile_k 5: 220 us with one
    // item per wavefront, 197 with three)
    const dim3 grid((unsigned)item_workgroups(tile_item_bound(a, item_bound))), block(PAR_WAVE_NW * 64);
    if (a.flags & PAR_DEBUG_FLAGS) {
        hipLaunchKernelGGL((render_tiles_kernel<true, true, true>), grid, block, 0, stream, g, a);
    } else if (a.sprite_ids || a.out.brightness || a.out.lit || a.out.gbuf) {
        hipLaunchKernelGGL((render_tiles_kernel<false, true, true>), grid, block, 0, stream, g, a);
    } else {
        hipLaunchKernelGGL((render_tiles_kernel<false, false, false>), grid, block, 0, stream, g, a);
    }
    return hipGetLastError();
}

// Small frames: work items, tile items and overflowed columns in one launch. hipErrorNotSupported (nothing launched)
// for large frames, where the kernels' different register needs matter.
hipError_t par_launch_render_both(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                  int64_t item_bound, bool may_overflow, hipStream_t stream) {
    const int64_t cols_in_range = (int64_t)g.gx * (a.by_hi - a.by_lo + 1);
    const int64_t bound = column_bound < cols_in_range ? column_bound : cols_in_range;
    if (bound >= 2048 || a.dense) return hipErrorNotSupported;
    if (bound <= 0) return hipSuccess;
    const int over_parts = 8;
    const int64_t over_cols = !may_overflow ? 0 : (bound < 32 ? bound : 32);
    int64_t n_item_wgs = item_workgroups(item_bound);
    int64_t n_tile_wgs = a.tile_k > 0 ? item_workgroups(tile_item_bound(a, item_bound)) : 0;
    if (bound <= 256) {
        // A frame of a few hundred columns is as long as its slowest wavefront, and a column's items all lie in ONE of
        // the 64 shards (the column's index picks it): a launch sized by the frame's item count gives the fuller shards
        // fewer wavefronts than items, and theirs take two or three items one after the other (the 480x320 graybox
        // world, 94 columns of 25 chunks: 14 us for the launch, of which one chunk's work is 3). So: a wavefront for
        // every item the fullest shard can hold -- its columns' share, every one of them a whole tile.
        const int64_t per_col = (((int64_t)a.B * a.B + 63) / 64);
        const int64_t cols_per_shard = (bound + PAR_ITEM_SHARDS - 1) / PAR_ITEM_SHARDS;
        // (whole workgroups per shard: render_items deals a list's workgroups to the shards in turn)
        const int64_t item_wgs_per_shard = (cols_per_shard * per_col + PAR_WAVE_NW - 1) / PAR_WAVE_NW;
        n_item_wgs = std::max(n_item_wgs, (int64_t)PAR_ITEM_SHARDS * item_wgs_per_shard);
        if (a.tile_k > 0) {
            const int64_t tile_wgs_per_shard = (cols_per_shard * ((per_col + a.tile_k - 1) / a.tile_k) + PAR_WAVE_NW - 1) / PAR_WAVE_NW;
            n_tile_wgs = std::max(n_tile_wgs, (int64_t)PAR_ITEM_SHARDS * tile_wgs_per_shard);
        }
    }
    const dim3 grid((unsigned)(n_item_wgs + n_tile_wgs + over_cols * over_parts));
    if (a.flags & PAR_DEBUG_FLAGS) {
        hipLaunchKernelGGL(render_both_kernel<true>, grid, dim3(PAR_WAVE_NW * 64), 0, stream, g, a, (int)n_item_wgs,
                           (int)n_tile_wgs, over_parts);
    } else {
        hipLaunchKernelGGL(render_both_kernel<false>, grid, dim3(PAR_WAVE_NW * 64), 0, stream, g, a, (int)n_item_wgs,
                           (int)n_tile_wgs, over_parts);
    }
    return hipGetLastError();
}

hipError_t par_launch_render_overflow(const par_grid_dev& g, const par_render_args& a, int64_t column_bound,
                                      hipStream_t stream) {
    const int64_t cols_in_range = (int64_t)g.gx * (a.by_hi - a.by_lo + 1);
    const int64_t bound = column_bound < cols_in_range ? column_bound : cols_in_range;
    if (bound <= 0) return hipSuccess;
    // overflowed columns are the exception: a small strided grid (up to 8 workgroups share a column)
    const int64_t oblocks = a.dense ? (bound < 1024 ? bound : 1024) : (bound < 32 ? bound : 32);
    hipLaunchKernelGGL(render_overflow_kernel, dim3((unsigned)oblocks, 8u), dim3(PAR_WAVE_NW * 64), 0, stream, g, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Sharded frames: the tiles that travel. One workgroup per tile; a tile's row is bin_size consecutive
// pixels in the frame and in the slot (coalesced both ways); pixels beyond the view's edge are left alone.
// PACK: frame block -> slots, else slots -> frame.
// ------------------------------------------------------------------------------------------------------------
namespace {
template <bool PACK>
__global__ __launch_bounds__(256) void tiles_copy_kernel(const int32_t* tiles, int n, int W, int H, int B, int row_begin,
                                                          int row_end, const uint32_t* src, uint32_t* dst) {
    const int t = (int)blockIdx.x;
    if (t >= n) return;
    const int tile = tiles[t];
    const int c0 = (tile & 0xFFFF) * B, r0 = (tile >> 16) * B;
    const int tw = min(B, W - c0);
    const int rows_lo = max(r0, row_begin), rows_hi = min(min(r0 + B, H), row_end);
    const uint32_t* s = src;
    uint32_t* d = dst;
    for (int p = (int)threadIdx.x; p < B * B; p += (int)blockDim.x) {
        const int y = p / B, x = p - y * B;
        const int row = r0 + y;
        if (x >= tw || row < rows_lo || row >= rows_hi) continue;
        const size_t in_frame = (size_t)(row - row_begin) * (size_t)W + (size_t)(c0 + x);
        const size_t in_slot = (size_t)t * (size_t)(B * B) + (size_t)p;
        if (PACK) d[in_slot] = s[in_frame];
        else d[in_frame] = s[in_slot];
    }
}

__global__ __launch_bounds__(256) void background_rows_kernel(uint32_t* dst, size_t n_px, uint32_t rgba) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const size_t n4 = n_px >> 2;
    const u32x4 v = {rgba, rgba, rgba, rgba};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst) + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n_px & 3)) dst[(n4 << 2) + threadIdx.x] = rgba;
}

// Rows [row_begin, row_end) of an assembled frame in ONE pass over them: where `map` (tile column + tile row * gx ->
// slot, or -1) names a packed tile, the tile's pixels, else the background. The frame is written exactly once (the
// separate background fill + unpack wrote the covered tiles twice and took two launches). VEC = pixels per thread
// and step: 4 (16-byte pieces; W and B are multiples of 4, so a piece lies within one tile) or 1.
template <int VEC>
__global__ __launch_bounds__(256) void tiles_assemble_kernel(const int32_t* __restrict__ map, int gx, int W, int B,
                                                             int row_begin, int row_end,
                                                             const uint32_t* __restrict__ packed,
                                                             uint32_t* __restrict__ frame, uint32_t rgba,
                                                             uint32_t magic_b) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int x = q * VEC;
    if (x >= W) return;
    // (divisions by the bin size through its reciprocal, as div_bin: an integer division per piece and row would cost
    // this streaming kernel as many issue cycles as its stores take)
    const int bx = (int)__umulhi((uint32_t)x, magic_b), xin = x - bx * B;
#pragma unroll 4
    for (int y = row_begin + (int)blockIdx.y; y < row_end; y += (int)gridDim.y) {
        const int by = (int)__umulhi((uint32_t)y, magic_b);
        const int slot = map[(size_t)by * gx + bx];
        const size_t at = (size_t)y * (size_t)W + (size_t)x;
        if (VEC == 4) {
            u32x4 v = {rgba, rgba, rgba, rgba};
            if (slot >= 0) {
                v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(packed + (size_t)slot * (size_t)(B * B) +
                                                                              (size_t)((y - by * B) * B + xin)));
            }
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(frame + at));
        } else {
            uint32_t v = rgba;
            if (slot >= 0) v = packed[(size_t)slot * (size_t)(B * B) + (size_t)((y - by * B) * B + xin)];
            frame[at] = v;
        }
    }
}
}  // namespace

hipError_t par_launch_tiles_copy(bool pack, const int32_t* d_tiles, int n, int W, int H, int B, int row_begin, int row_end,
                                 const void* src, void* dst, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (pack) {
        hipLaunchKernelGGL(tiles_copy_kernel<true>, dim3((unsigned)n), dim3(256), 0, stream, d_tiles, n, W, H, B, row_begin,
                           row_end, static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst));
    } else {
        hipLaunchKernelGGL(tiles_copy_kernel<false>, dim3((unsigned)n), dim3(256), 0, stream, d_tiles, n, W, H, B, row_begin,
                           row_end, static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst));
    }
    return hipGetLastError();
}

hipError_t par_launch_background(void* dst, size_t n_px, uint32_t rgba, hipStream_t stream) {
    if (n_px == 0) return hipSuccess;
    size_t blocks = (n_px / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(background_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<uint32_t*>(dst), n_px,
                       rgba);
    return hipGetLastError();
}

hipError_t par_launch_tiles_assemble(const int32_t* d_map, int gx, int W, int B, int row_begin, int row_end,
                                     const void* packed, void* frame, uint32_t rgba, hipStream_t stream) {
    const int rows = row_end - row_begin;
    if (rows <= 0 || W <= 0) return hipSuccess;
    const bool wide = (W % 4 == 0) && (B % 4 == 0) && ((uintptr_t)frame % 16 == 0) && ((uintptr_t)packed % 16 == 0);
    const uint32_t magic_b = (uint32_t)((1ull << 32) / (uint64_t)B + 1ull);  // (exact while coordinate * B < 2^32)
    const int per_row = wide ? W / 4 : W;
    const unsigned bx = (unsigned)((per_row + 255) / 256);
    // each workgroup streams a few rows: about 2 048 workgroups write at the rate HBM takes
    static const int target = [] { const char* e = std::getenv("PAR_TUNE_ASSEMBLE_WGS"); return e ? std::atoi(e) : 2048; }();
    unsigned by = (unsigned)std::max(1, std::min(std::min(rows, 65535), (int)((unsigned)std::max(target, 1) / bx)));
    if (wide) {
        hipLaunchKernelGGL(tiles_assemble_kernel<4>, dim3(bx, by), dim3(256), 0, stream, d_map, gx, W, B, row_begin, row_end,
                           static_cast<const uint32_t*>(packed), static_cast<uint32_t*>(frame), rgba, magic_b);
    } else {
        hipLaunchKernelGGL(tiles_assemble_kernel<1>, dim3(bx, by), dim3(256), 0, stream, d_map, gx, W, B, row_begin, row_end,
                           static_cast<const uint32_t*>(packed), static_cast<uint32_t*>(frame), rgba, magic_b);
    }
    return hipGetLastError();
}

hipError_t par_launch_units(int kind, const void* in_a, const void* in_b, int n, void* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(units_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kind, in_a, in_b, n, out);
    return hipGetLastError();
}
