// par_scene.cpp — host-side scene helpers of the C ABI (no GPU needed).
//
// The reference builds its scene inside `main` (src/alternative.cpp:517-599) and owns it in
// `Entities<N>{vector<AABB>, vector<Sprite>}` (alt:92-114). These helpers produce the same data as plain arrays
// the caller owns, ready for par_set_entities().
#include <cstdlib>
#include <cstring>

#include "par_raytracer.h"

extern "C" {

const char* par_status_string(int status) {
    switch (status) {
        case PAR_OK: return "ok";
        case PAR_ERR_INVALID_ARG: return "invalid argument";
        case PAR_ERR_NO_DEVICE: return "no gfx950 HIP device";
        case PAR_ERR_HIP: return "HIP runtime error";
        case PAR_ERR_OOM: return "out of memory";
        case PAR_ERR_UNSUPPORTED: return "unsupported grid or bin size";
        case PAR_ERR_EXTENT: return "AABB extent not expressible by the 20x40 sprite";
        case PAR_ERR_SPRITE_ID: return "sprite id or palette index out of range";
        case PAR_ERR_NOT_READY: return "scene incomplete (sprites, entities and light must be set)";
        case PAR_ERR_DEVICE: return "a kernel reported a failure: a frame rendered since the last check is not valid";
        default: return "unknown status";
    }
}

void par_default_params(par_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->width = 480;           // view_width  alt:117
    p->height = 320;          // view_height alt:118
    p->length = 320;          // view_length alt:119
    p->bin_size = 40;         // single_bin_cubic_size alt:116
    p->ambient = 0.25f;       // ambient_light alt:702
    p->background = 255 / 2;  // alt:281
    p->palette_size = 4;      // color_palette spr:60-65 (alpha is value-initialised to 0 there)
    const uint8_t gray[4] = {100, 140, 200, 240};
    for (int i = 0; i < 4; i++) p->palette[i] = par_color{gray[i], gray[i], gray[i], 0};
}

int par_grid_dims(const par_params* p, int* gx, int* gy, int* gz) {
    if (!p || p->bin_size <= 0 || p->width <= 0 || p->height <= 0 || p->length <= 0) return PAR_ERR_INVALID_ARG;
    // hash_width/height/length alt:120-122. Ceil division: identical for views that are a multiple of the bin
    // size (the reference's 480x320x320 / 40) and keeps every pixel inside the grid otherwise.
    if (gx) *gx = (p->width + p->bin_size - 1) / p->bin_size;
    if (gy) *gy = (p->height + p->bin_size - 1) / p->bin_size;
    if (gz) *gz = (p->length + p->bin_size - 1) / p->bin_size;
    return PAR_OK;
}

// make_tile_floor, spr:73-364, from its structure: rows 0-19 are the top face (normal +y, depth
// 19-row, a 2x2 checker of 6x6 texel blocks in palette 2/3 on a border of 0), rows 20-39 the front face (normal
// -z, depth 0, palette 2 framed by 1).
void par_sprite_tile_floor(par_sprite* out) {
    if (!out) return;
    for (int r = 0; r < PAR_SPRITE_H; r++) {
        for (int c = 0; c < PAR_SPRITE_W; c++) {
            const int t = r * PAR_SPRITE_W + c;
            if (r < 20) {
                const bool inner = r >= 4 && r < 16 && c >= 4 && c < 16;
                out->color[t] = inner ? (((r < 10) == (c < 10)) ? 2 : 3) : 0;
                out->depth[t] = 19 - r;
                out->normal[t] = par_vec3{0.f, 1.f, 0.f};
            } else {
                out->color[t] = (r >= 38 || c < 2 || c >= 18) ? 1 : 2;
                out->depth[t] = 0;
                out->normal[t] = par_vec3{0.f, 0.f, -1.f};
            }
        }
    }
}

namespace {
struct Emitter {
    par_aabb* out;
    int capacity;
    int n = 0;
    void box(int x, int y, int z) {
        if (out && n < capacity) {
            // positions go through `static_cast<short>` in the reference (alt:538-540 and alike)
            out[n] = par_aabb{static_cast<int16_t>(x), static_cast<int16_t>(y), static_cast<int16_t>(z), 20, 20, 20, {0, 0}};
        }
        n++;
    }
};
}  // namespace

// The graybox world, alt:517-599, parameterised on the view the reference hard-codes as 480 (width) x 320 (length).
int par_scene_graybox(int view_width, int view_length, par_aabb* out, int capacity) {
    Emitter e{out, capacity};
    e.box(view_width / 2, 36, view_length / 4);  // player, alt:520-523
    for (int i = 0; i < view_width; i++) {       // floor with a hole, alt:527-547
        for (int j = 0; j < view_length; j++) {
            const int x = i * 20, z = j * 20;
            if (x >= view_width / 2 - 40 && x < view_width / 2 + 40 && z < view_length / 2 + 40 &&
                z > view_length / 2 - 40) {
                continue;
            }
            e.box(x, 0, z);
        }
    }
    for (int i = 0; i < 6; i++) {  // left wall, alt:549-568
        for (int j = 0; j < view_length - 10; j++) {
            for (int k = 1; k < 6; k++) {
                if (i >= 4 && k >= 4) continue;
                e.box(i * 20, k * 20, view_length - j * 20);
            }
        }
    }
    for (int i = 1; i < 3; i++) {  // right strips, alt:570-584
        for (int j = 0; j < view_length; j++) e.box(view_width - i * 20, 20, j * 20);
    }
    for (int i = 1; i < 20; i++) e.box(view_width - 40 - i * 20, 20, view_length - 60);  // back strip, alt:586-598
    return e.n;
}

namespace {
// splitmix64 (Steele, Lea, Flood 2014): language-independent, so C++ and numpy scene builders agree.
struct SplitMix64 {
    uint64_t state;
    uint64_t next() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};
}  // namespace

// Synthetic benchmark scene, SURVEY §8d: extent-(20,20,20) boxes (the only extent the 20x40 sprite supports),
// uniformly placed; the light sits strictly inside the view volume.
int par_scene_synthetic(int n, int width, int height, int length, uint64_t seed, par_aabb* out, par_light* light) {
    if (n < 0 || width <= 0 || height <= 0 || length <= 0) return PAR_ERR_INVALID_ARG;
    SplitMix64 rng{seed};
    for (int i = 0; i < n; i++) {
        const int x = -20 + static_cast<int>(rng.next() % static_cast<uint64_t>(width + 20));
        const int y = -20 + static_cast<int>(rng.next() % 220u);
        const int z = -20 + static_cast<int>(rng.next() % static_cast<uint64_t>(length + 20));
        if (out) {
            out[i] = par_aabb{static_cast<int16_t>(x), static_cast<int16_t>(y), static_cast<int16_t>(z), 20, 20, 20, {0, 0}};
        }
    }
    if (light) {
        *light = par_light{static_cast<int16_t>(5 * width / 8), static_cast<int16_t>(height / 2),
                           static_cast<int16_t>(length / 4), 10};
    }
    return PAR_OK;
}

// draw_line (alt:139-175) as called at alt:763-772: from (mouse_x, H - (pick.y + pick.z)) to the light's screen
// position, colour {255,0,0,255}, bounds-checked by the callback.
void par_debug_line(const par_params* p, const par_pixel* pick, int mouse_x, const par_light* light, par_color* fb) {
    if (!p || !pick || !light || !fb) return;
    const int W = p->width, H = p->height;
    int x = mouse_x, y = H - (pick->y + pick->z);
    const int x_end = light->x, y_end = H - (light->y + light->z);
    const int x_delta = std::abs(x_end - x), y_delta = -std::abs(y_end - y);
    const int x_sign = x < x_end ? 1 : -1, y_sign = y < y_end ? 1 : -1;
    int error = x_delta + y_delta;
    for (;;) {
        if (x >= 0 && y >= 0 && x < W && y < H) fb[x + static_cast<size_t>(y) * W] = par_color{255, 0, 0, 255};
        if (x == x_end && y == y_end) return;
        const int error2 = 2 * error;
        if (error2 >= y_delta) {
            if (x == x_end) return;
            error += y_delta;
            x += x_sign;
        }
        if (error2 <= x_delta) {
            if (y == y_end) return;
            error += x_delta;
            y += y_sign;
        }
    }
}

// Row blocks of a frame sharded over several GPUs: cut at bin rows, the bin rows dealt evenly.
// The screen columns the entities reach: cull and bin ranges of alt:212-240 (x and y only; the z range decides how
// many bins of a column an entity lands in, not whether it reaches the column -- an entity whose z range is empty
// reaches none).
int par_scene_tiles(const par_params* p, const par_aabb* aabbs, int n, int32_t* tiles, int capacity) {
    int gx, gy, gz;
    if (!p || n < 0 || (n > 0 && !aabbs) || capacity < 0 || (capacity > 0 && !tiles) ||
        par_grid_dims(p, &gx, &gy, &gz) != PAR_OK) {
        return -PAR_ERR_INVALID_ARG;
    }
    if (gx > 65535 || gy > 32767) return -PAR_ERR_UNSUPPORTED;
    const int W = p->width, H = p->height, L = p->length, B = p->bin_size;
    unsigned char* reached = static_cast<unsigned char*>(std::calloc((size_t)gx * gy, 1));
    if (!reached) return -PAR_ERR_OOM;
    for (int i = 0; i < n; i++) {
        const par_aabb& a = aabbs[i];
        const int minx = a.px, miny = a.py, minz = a.pz;
        const int maxx = minx + a.ex, maxy = miny + a.ey, maxz = minz + a.ez;
        if ((maxx < 0) || (minx >= W) || (maxy < 0 - maxz) || (miny >= H - minz + B) || (maxz < -a.ez - B) ||
            (minz > L + B)) {
            continue;  // alt:212-219
        }
        const int x0 = minx / B > 0 ? minx / B : 0, y0 = (H - maxy - maxz) / B > 0 ? (H - maxy - maxz) / B : 0;
        const int z0 = minz / B > 0 ? minz / B : 0;
        const int x1 = (maxx + B - 1) / B < gx ? (maxx + B - 1) / B : gx;
        const int y1 = (H - miny - minz + B - 1) / B < gy ? (H - miny - minz + B - 1) / B : gy;
        const int z1 = (maxz + B - 1) / B < gz ? (maxz + B - 1) / B : gz;  // alt:222-240
        if (z1 <= z0) continue;
        for (int y = y0; y < y1; y++) {
            for (int x = x0; x < x1; x++) reached[(size_t)y * gx + x] = 1;
        }
    }
    int count = 0;
    for (int y = 0; y < gy; y++) {
        for (int x = 0; x < gx; x++) {
            if (!reached[(size_t)y * gx + x]) continue;
            if (count < capacity) tiles[count] = x | (y << 16);
            count++;
        }
    }
    std::free(reached);
    return count;
}

int par_scene_tile_map(const par_params* p, const int32_t* tiles, int n, int32_t* map, int capacity) {
    int gx, gy, gz;
    if (!p || n < 0 || (n > 0 && !tiles) || !map || par_grid_dims(p, &gx, &gy, &gz) != PAR_OK) return PAR_ERR_INVALID_ARG;
    if ((long long)capacity < (long long)gx * gy) return PAR_ERR_INVALID_ARG;
    for (long long i = 0; i < (long long)gx * gy; i++) map[i] = -1;
    for (int i = 0; i < n; i++) {
        const int bx = tiles[i] & 0xFFFF, by = tiles[i] >> 16;
        if (bx >= gx || by < 0 || by >= gy) return PAR_ERR_INVALID_ARG;
        map[(size_t)by * gx + bx] = i;
    }
    return PAR_OK;
}

void par_row_block(int rank, int ranks, int height, int bin_size, int* begin, int* end) {
    if (ranks < 1) ranks = 1;
    if (bin_size < 1) bin_size = 1;
    rank = rank < 0 ? 0 : (rank >= ranks ? ranks - 1 : rank);
    const long long bins = ((long long)height + bin_size - 1) / bin_size;  // bin rows, the last one maybe partial
    const long long b0 = bins * rank / ranks, b1 = bins * (rank + 1) / ranks;
    const long long r0 = b0 * bin_size, r1 = b1 * bin_size;
    if (begin) *begin = (int)(r0 < height ? r0 : height);
    if (end) *end = (int)(r1 < height ? r1 : height);
}

}  // extern "C"
