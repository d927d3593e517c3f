// ref_harness.cpp — glue that exposes the REFERENCE's own hot-path functions through a C ABI. TEST INFRASTRUCTURE.
//
// This file is never compiled on its own: oracle/Makefile streams lines 2-500 of
// /root/reference/src/alternative.cpp (everything between the SDL include and `main`: the types, the grid
// constants, count_entities_in_bins, trace_hash_for_pixel, trace_hash_for_light, AABB::intersect — none of which
// touch SDL) into the compiler from where they lie, followed by this file, and writes only
// oracle/_ref/libref_path.so. No reference source is copied into the repository and no stand-in for the absent
// SDL2 headers is written; the reference's `main` (window, input, blit) is therefore NOT part of this build.
// Its inline shading loop (alt:702-760) and its debug-line call (alt:763-772), which use no SDL symbol, ARE: the
// Makefile streams those lines, from where they lie, into the bodies of `ref_shade_own` and `ref_debug_line_own`
// at the two marker lines below; the parameters and locals there carry the names the reference's lines use
// (p_pixel_buffer, p_texture, lights, p_aabb_count_in_bin, p_aabb_bins, p_aabb_index_to_entity_index_map,
// mouse_pixel). `ref_shade` further down is a REPLAY of the same loop around the reference's own callees that also
// hands out the brightness and lit planes, which the reference's loop keeps in locals; the tests hold the two
// against each other. The whole-program known answers of SURVEY Appendix B pin all of it independently.
#include <cstring>

#include "par_types.h"

static_assert(sizeof(par_aabb) == sizeof(AABB) && alignof(AABB) == 16);
static_assert(sizeof(par_pixel) == sizeof(Pixel));
static_assert(sizeof(par_sprite) == sizeof(Sprite));
static_assert(sizeof(par_color) == sizeof(Color));
static_assert(sizeof(par_ray) == sizeof(Ray));

extern "C" {

void ref_consts(int* out) {
    out[0] = single_bin_cubic_size;
    out[1] = view_width;
    out[2] = view_height;
    out[3] = view_length;
    out[4] = hash_width;
    out[5] = hash_height;
    out[6] = hash_length;
    out[7] = sparse_bin_size;
}

void ref_tile_sprite(void* out) { std::memcpy(out, &tile_single, sizeof(Sprite)); }

void ref_palette(void* out) { std::memcpy(out, color_palette, sizeof(color_palette)); }

void ref_color_scale(const par_color* c, float v, par_color* out) {
    Color in{c->red, c->green, c->blue, c->alpha};
    Color r = in * v;
    std::memcpy(out, &r, 4);
}

void ref_normalize(const par_vec3* v, par_vec3* out) {
    Vector<float> r = Vector<float>{v->x, v->y, v->z}.normalize();
    out->x = r.x;
    out->y = r.y;
    out->z = r.z;
}

int ref_intersect(const par_aabb* box, const par_ray* ray) {
    AABB a;
    std::memcpy(&a, box, sizeof(a));
    Ray r;
    std::memcpy(&r, ray, sizeof(r));
    return a.intersect(r) ? 1 : 0;
}

void* ref_scene_create(const par_aabb* aabbs, int n) {
    auto* e = new Entities<entity_count>;
    e->aabbs.reserve(n);
    e->sprites.reserve(n);
    for (int i = 0; i < n; i++) {
        AABB a;
        std::memcpy(&a, &aabbs[i], sizeof(a));
        e->insert({.aabb = a});
    }
    return e;
}

void ref_scene_set_aabb(void* h, int i, const par_aabb* a) {
    std::memcpy(&static_cast<Entities<entity_count>*>(h)->aabbs[i], a, sizeof(AABB));
}

void ref_scene_free(void* h) { delete static_cast<Entities<entity_count>*>(h); }

// alt:690-693
void ref_bin(void* h, int* count, int* map, par_aabb* bins) {
    std::memset(count, 0, hash_volume * sizeof(int));
    count_entities_in_bins(static_cast<Entities<entity_count>*>(h), reinterpret_cast<AABB*>(bins), count, map);
}

// alt:694
void ref_primary(void* h, int* count, int* map, par_aabb* bins, par_pixel* gbuf) {
    mouse_x = -1;
    mouse_y = -1;
    trace_hash_for_pixel(static_cast<Entities<entity_count>*>(h), reinterpret_cast<AABB*>(bins), count, map,
                         reinterpret_cast<Pixel*>(gbuf));
}

int ref_shadow(int* count, int* map, par_aabb* bins, int sx, int sy, int sz, int ex, int ey, int ez,
               int start_entity, const par_ray* ray) {
    Ray r;
    std::memcpy(&r, ray, sizeof(r));
    return trace_hash_for_light(count, reinterpret_cast<AABB*>(bins), map, sx, sy, sz, ex, ey, ez, start_entity, r)
               ? 1
               : 0;
}

// Replay of the inline loop alt:702-760 around the reference's own callee functions. The light must keep every
// probed flat bin index inside [0, hash_volume) (i.e. 0 <= light bin-x < 12 and the y/z bins in range), or the
// reference reads past its arrays (alt:476); the caller guarantees that.
void ref_shade(int* count, int* map, par_aabb* bins, const par_pixel* gbuf_in, const par_light* light,
               par_color* fb, float* brightness, unsigned char* lit_plane) {
    const Pixel* gbuf = reinterpret_cast<const Pixel*>(gbuf_in);
    float ambient_light = 0.25f;
    for (int i = 0; i < view_height * view_width; i++) {
        Pixel px = gbuf[i];
        Vector normal = px.normal;
        int world_x = i % view_width, world_y = px.y, world_z = px.z;
        Vector towards_light = Vector{.x = static_cast<float>(light->x - world_x),
                                      .y = static_cast<float>(light->y - world_y),
                                      .z = static_cast<float>(light->z - world_z)}
                                   .normalize();
        Ray ray = {.direction_inverse = {.x = 1.f / towards_light.x,
                                         .y = 1.f / towards_light.y,
                                         .z = 1.f / towards_light.z},
                   .origin = {static_cast<short>(world_x), static_cast<short>(world_y),
                              static_cast<short>(world_z)}};
        int rbx = world_x / single_bin_cubic_size;
        int rby = (view_height - world_y - world_z) / single_bin_cubic_size;
        int rbz = world_z / single_bin_cubic_size;
        int lbx = light->x / single_bin_cubic_size;
        int lby = (view_height - light->y - light->z) / single_bin_cubic_size;
        int lbz = light->z / single_bin_cubic_size;
        Color out = px.color * ambient_light;
        float b = ambient_light;
        bool lit = trace_hash_for_light(count, reinterpret_cast<AABB*>(bins), map, rbx, rby, rbz, lbx, lby, lbz,
                                        px.entity_index, ray);
        if (lit) {
            float diffuse = std::max<float>(0, normal.x * towards_light.x + normal.y * towards_light.y +
                                                   normal.z * towards_light.z);
            b = std::min<float>(1.f, diffuse + ambient_light);
            out = px.color * b;
        }
        std::memcpy(&fb[i], &out, 4);
        if (brightness) brightness[i] = b;
        if (lit_plane) lit_plane[i] = lit ? 1 : 0;
    }
}

// The reference's OWN shading loop, alt:702-760, compiled from where it lies (see the Makefile): the final RGBA
// frame from a G-buffer. The light must keep every probed flat bin index inside [0, hash_volume), as for ref_shade.
void ref_shade_own(int* p_aabb_count_in_bin, int* p_aabb_index_to_entity_index_map, par_aabb* bins_in,
                   const par_pixel* gbuf_in, const par_light* light_in, par_color* fb) {
    AABB* p_aabb_bins = reinterpret_cast<AABB*>(bins_in);
    // (the loop binds `Pixel& this_pixel` and only reads through it)
    Pixel* p_pixel_buffer = const_cast<Pixel*>(reinterpret_cast<const Pixel*>(gbuf_in));
    Color* p_texture = reinterpret_cast<Color*>(fb);
    const par_light lights[1] = {*light_in};
// @@REFERENCE alt:702-760@@
}

// The reference's OWN debug-line call, alt:763-772: Bresenham from the picked texel's screen position to the
// light's, drawn into the frame (`draw_line` itself, alt:139-175, is part of lines 2-500).
void ref_debug_line_own(const par_pixel* pick, int mouse_x_in, int mouse_y_in, const par_light* light_in,
                        par_color* fb) {
    Color* p_texture = reinterpret_cast<Color*>(fb);
    const par_light lights[1] = {*light_in};
    Pixel picked;
    std::memcpy(&picked, pick, sizeof(picked));
    mouse_pixel = &picked;
    mouse_x = mouse_x_in;
    mouse_y = mouse_y_in;
// @@REFERENCE alt:763-772@@
    mouse_pixel = nullptr;
}

}  // extern "C"
