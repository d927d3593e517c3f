/*
 * par_types.h — plain-C data model shared by the C-ABI (par_raytracer.h), the CPU oracle and the HIP kernels.
 *
 * Every struct here is byte-for-byte layout compatible with the reference type it replaces, so a caller that
 * holds the reference's `Entities`, `Sprite`, `Pixel`, `Color` arrays can hand the same memory to this library.
 * Citations are `file:line` relative to the reference repository (spr = src/sprites.hpp, alt = src/alternative.cpp).
 */
#ifndef PAR_TYPES_H
#define PAR_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sprite geometry is fixed by the reference: 20 texels wide (literal `20` at alt:330), 40 tall (spr:67-71). */
#define PAR_SPRITE_W 20
#define PAR_SPRITE_H 40
#define PAR_SPRITE_TEXELS (PAR_SPRITE_W * PAR_SPRITE_H)
/* Slots per hash bin, wrapping counter (`sparse_bin_size`, alt:131). */
#define PAR_SLOTS 8
/* Largest palette this build accepts (the reference ships 4 entries, spr:60-65). */
#define PAR_MAX_PALETTE 256
/* Palette-index plane value written for pixels no primitive covers. */
#define PAR_PALIDX_BACKGROUND 0xFF

/* `Color`, spr:5-17: RGBA8, channel order red,green,blue,alpha in memory. */
typedef struct par_color {
    uint8_t red, green, blue, alpha;
} par_color;

/* `Vector<float>`, spr:20-51. */
typedef struct par_vec3 {
    float x, y, z;
} par_vec3;

/* `Pixel`, spr:53-58: the 28-byte G-buffer texel. */
typedef struct par_pixel {
    par_vec3 normal;
    par_color color;
    int32_t y, z;
    int32_t entity_index;
} par_pixel;

/* `Sprite`, spr:67-71: 16 000 bytes; texel index = row * 20 + column. */
typedef struct par_sprite {
    int32_t color[PAR_SPRITE_TEXELS]; /* palette index */
    int32_t depth[PAR_SPRITE_TEXELS];
    par_vec3 normal[PAR_SPRITE_TEXELS];
} par_sprite;

/* `AABB`, alt:35-38,88: two `Point<short>` + 4 bytes of tail padding, 16-byte aligned. */
typedef struct par_aabb {
    int16_t px, py, pz; /* position */
    int16_t ex, ey, ez; /* extent   */
    int16_t pad_[2];
} par_aabb;

/* `Light`, alt:619-622. `radius` is carried but, as in the reference, never read. */
typedef struct par_light {
    int16_t x, y, z;
    int16_t radius;
} par_light;

/* `Ray`, alt:30-33 (20 bytes: fp32 inverse direction + short origin, 2 bytes tail padding). */
typedef struct par_ray {
    float inv_x, inv_y, inv_z;
    int16_t ox, oy, oz;
    int16_t pad_;
} par_ray;

/*
 * View / grid parameters. The reference hard-codes these as constexpr (alt:116-131) and literals
 * (alt:281 background, alt:702 ambient); here they are runtime values whose defaults reproduce the reference.
 * Grid dimensions are ceil(width/bin), ceil(height/bin), ceil(length/bin): identical to alt:120-122 whenever the view
 * is a multiple of the bin size (as 480x320x320 / 40 is).
 */
typedef struct par_params {
    int32_t width;      /* view_width  alt:117 (480) */
    int32_t height;     /* view_height alt:118 (320) */
    int32_t length;     /* view_length alt:119 (320) */
    int32_t bin_size;   /* single_bin_cubic_size alt:116 (40) */
    float ambient;      /* ambient_light alt:702 (0.25f); must lie in [0,1] */
    uint8_t background; /* gray level of uncovered pixels, alt:281 (255/2 = 127) */
    uint8_t reserved_[3];
    int32_t palette_size; /* entries of `palette` in use, spr:60-65 (4) */
    par_color palette[PAR_MAX_PALETTE];
} par_params;

#ifdef __cplusplus
}
#endif
#endif /* PAR_TYPES_H */
