/*
 * par_oracle.h — CPU restatement of the reference render hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (pixel-art-raytracer_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED. Every function below is checked (tests/test_oracle_vs_reference.py, run in the build
 * container) against the reference's own functions compiled from /root/reference/src/alternative.cpp lines 2-500
 * (oracle/Makefile target `_ref`), and against the known answers the survey recorded from the unmodified
 * whole program (tests/golden/appendix_b.json).
 *
 * Citations: alt = src/alternative.cpp, spr = src/sprites.hpp of the reference.
 */
#ifndef PAR_ORACLE_H
#define PAR_ORACLE_H

#include "../include/par_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The spatial hash: `p_aabb_count_in_bin`, `p_aabb_index_to_entity_index_map`, `p_aabb_bins` (alt:503-509). */
typedef struct par_oracle_grid {
    int gx, gy, gz;  /* hash_width/height/length alt:120-122 (ceil division) */
    int volume;      /* hash_volume alt:123 */
    int32_t* count;  /* [volume]              */
    int32_t* map;    /* [volume * PAR_SLOTS]  */
    par_aabb* bins;  /* [volume * PAR_SLOTS]  */
} par_oracle_grid;

void par_oracle_default_params(par_params* p);
void par_oracle_grid_dims(const par_params* p, int* gx, int* gy, int* gz);

/* `Color::operator*`, spr:8-16. */
par_color par_oracle_color_scale(par_color c, float v);
/* `Vector::normalize` (L1 length), spr:28-35. */
par_vec3 par_oracle_normalize(par_vec3 v);
/* `AABB::intersect`, alt:40-83. */
int par_oracle_intersect(const par_aabb* box, const par_ray* ray);
/* `index_into_view_hash`, alt:180-182. */
int par_oracle_hash_index(const par_oracle_grid* g, int x, int y, int z);

/* memset (alt:690) + `count_entities_in_bins` (alt:195-269). `map`/`bins` slots not written keep their contents. */
void par_oracle_bin(const par_params* p, const par_aabb* aabbs, int n, par_oracle_grid* g);

/* `trace_hash_for_pixel`, alt:271-397, restricted to rows [row_begin,row_end). `sprite_ids` may be NULL (all 0). */
void par_oracle_primary(const par_params* p, const par_oracle_grid* g, const par_sprite* sprites,
                        const int32_t* sprite_ids, par_pixel* gbuf, uint8_t* palidx, int row_begin, int row_end);

/* `trace_hash_for_light`, alt:399-500. Returns 1 when the light is reached. `probes` (nullable) counts bin probes. */
int par_oracle_shadow(const par_oracle_grid* g, int sx, int sy, int sz, int ex, int ey, int ez, int start_entity,
                      const par_ray* ray, int64_t* probes);

/* The inline shading/quantise loop, alt:702-760, rows [row_begin,row_end). Nullable planes are skipped. */
void par_oracle_shade(const par_params* p, const par_oracle_grid* g, const par_pixel* gbuf, const par_light* light,
                      par_color* fb, float* brightness, uint8_t* lit, int row_begin, int row_end);

/* alt:690-760 for one frame: bin + primary + shade. All output planes nullable except that gbuf is allocated
 * internally when NULL. Returns 0, or -1 on allocation failure / bad parameters. */
int par_oracle_render(const par_params* p, const par_aabb* aabbs, int n, const par_sprite* sprites,
                      const int32_t* sprite_ids, const par_light* light, par_color* fb, par_pixel* gbuf,
                      uint8_t* palidx, float* brightness, uint8_t* lit);

/* Same frame, rows split across `nthreads` host threads (rows are independent). Ours, not the
 * reference's: the reference is single-threaded. */
int par_oracle_render_mt(const par_params* p, const par_aabb* aabbs, int n, const par_sprite* sprites,
                         const int32_t* sprite_ids, const par_light* light, par_color* fb, par_pixel* gbuf,
                         uint8_t* palidx, float* brightness, uint8_t* lit, int nthreads);

/* Debug overlay: `draw_line` (alt:139-175) as called at alt:763-772 with the pick pixel (mouse_x, mouse_y). */
void par_oracle_debug_line(const par_params* p, const par_pixel* gbuf, const par_light* light, int mouse_x,
                           int mouse_y, par_color* fb);

/* `make_tile_floor`, spr:73-364, regenerated procedurally. */
void par_oracle_tile_floor(par_sprite* out);

#ifdef __cplusplus
}
#endif
#endif
