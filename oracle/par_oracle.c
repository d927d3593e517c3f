/*
 * par_oracle.c — CPU restatement of the reference render hot path. TEST INFRASTRUCTURE ONLY (see par_oracle.h).
 *
 * Written from the semantics of the reference (alt = src/alternative.cpp, spr = src/sprites.hpp), with the
 * compile-time view/grid constants (alt:116-131) promoted to run-time parameters. Build with
 * `-O2 -ffp-contract=off` and never `-ffast-math`: the reference's canonical float behaviour is baseline x86-64
 * without FMA contraction.
 *
 * Defined behaviour where the reference has UB: a flat bin index outside [0, volume) reads as an empty bin
 * (heap overflow read at alt:476 when the light's bin-x >= hash_width).
 */
#include "par_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- small pieces ------------------------------------------------------------------------------------------ */

void par_oracle_default_params(par_params* p) {
    memset(p, 0, sizeof(*p));
    p->width = 480;  /* alt:117 */
    p->height = 320; /* alt:118 */
    p->length = 320; /* alt:119 */
    p->bin_size = 40; /* alt:116 */
    p->ambient = 0.25f; /* alt:702 */
    p->background = 255 / 2; /* alt:281 */
    p->palette_size = 4; /* spr:60-65 */
    static const uint8_t gray[4] = {100, 140, 200, 240};
    for (int i = 0; i < 4; i++) {
        p->palette[i].red = p->palette[i].green = p->palette[i].blue = gray[i];
        p->palette[i].alpha = 0;
    }
}

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

void par_oracle_grid_dims(const par_params* p, int* gx, int* gy, int* gz) {
    /* alt:120-122; ceil so that views that are not a multiple of the bin size keep every pixel inside the grid. */
    *gx = ceil_div(p->width, p->bin_size);
    *gy = ceil_div(p->height, p->bin_size);
    *gz = ceil_div(p->length, p->bin_size);
}

/* spr:8-16 — per-channel truncating scale, alpha passed through. */
par_color par_oracle_color_scale(par_color c, float v) {
    par_color r;
    r.red = (uint8_t)((float)c.red * v);
    r.green = (uint8_t)((float)c.green * v);
    r.blue = (uint8_t)((float)c.blue * v);
    r.alpha = c.alpha;
    return r;
}

/* spr:28-35 — divides by the L1 length (abs(x)+abs(y))+abs(z). */
par_vec3 par_oracle_normalize(par_vec3 v) {
    float length = __builtin_fabsf(v.x) + __builtin_fabsf(v.y) + __builtin_fabsf(v.z);
    par_vec3 r = {v.x / length, v.y / length, v.z / length};
    return r;
}

/* std::min(a,b) == (b<a)?b:a ; std::max(a,b) == (a<b)?b:a — with NaN the FIRST argument wins. */
static float std_min(float a, float b) { return (b < a) ? b : a; }
static float std_max(float a, float b) { return (a < b) ? b : a; }

/* alt:40-83 — branchless slab test on a line (no tmax >= 0 test). Argument order is part of the contract. */
int par_oracle_intersect(const par_aabb* box, const par_ray* ray) {
    float x1 = (float)(box->px - ray->ox) * ray->inv_x;                 /* alt:49-50 */
    float x2 = (float)(box->px + box->ex - ray->ox) * ray->inv_x;       /* alt:51-53 */
    float tmin = std_min(x1, x2);                                       /* alt:55 */
    float tmax = std_max(x1, x2);                                       /* alt:56 */
    float y1 = (float)(box->py - ray->oy) * ray->inv_y;                 /* alt:59-60 */
    float y2 = (float)(box->py + box->ey - ray->oy) * ray->inv_y;       /* alt:61-63 */
    tmin = std_max(tmin, std_min(y1, y2));                              /* alt:65-66 */
    tmax = std_min(tmax, std_max(y1, y2));                              /* alt:67-68 */
    float z1 = (float)(box->pz - ray->oz) * ray->inv_z;                 /* alt:71-72 */
    float z2 = (float)(box->pz + box->ez - ray->oz) * ray->inv_z;       /* alt:73-75 */
    tmin = std_max(tmin, std_min(z1, z2));                              /* alt:77-78 */
    tmax = std_min(tmax, std_max(z1, z2));                              /* alt:79-80 */
    return tmax >= tmin;                                                /* alt:82 */
}

/* alt:180-182 */
int par_oracle_hash_index(const par_oracle_grid* g, int x, int y, int z) {
    return (x * g->gy * g->gz) + (y * g->gz) + z;
}

/* ---- binning: alt:690 + alt:195-269 ------------------------------------------------------------------------- */

static int imax(int a, int b) { return a < b ? b : a; }
static int imin(int a, int b) { return b < a ? b : a; }

void par_oracle_bin(const par_params* p, const par_aabb* aabbs, int n, par_oracle_grid* g) {
    const int W = p->width, H = p->height, L = p->length, B = p->bin_size;
    memset(g->count, 0, (size_t)g->volume * sizeof(int32_t)); /* alt:690 */
    for (int i = 0; i < n; i++) {
        const par_aabb* a = &aabbs[i];
        int minx = a->px, miny = a->py, minz = a->pz;                     /* alt:202-204 */
        int maxx = minx + a->ex, maxy = miny + a->ey, maxz = minz + a->ez; /* alt:206-208 */
        /* cull, alt:212-219 */
        if ((maxx < 0) || (minx >= W) || (maxy < 0 - maxz) || (miny >= H - minz + B) || (maxz < -a->ez - B) ||
            (minz > L + B)) {
            continue;
        }
        int x0 = imax(0, minx / B);                      /* alt:222 */
        int y0 = imax(0, (H - maxy - maxz) / B);         /* alt:223-225 */
        int z0 = imax(0, minz / B);                      /* alt:226 */
        int x1 = imin(g->gx, (maxx + B - 1) / B);        /* alt:228-230 */
        int y1 = imin(g->gy, (H - miny - minz + B - 1) / B); /* alt:231-236 */
        int z1 = imin(g->gz, (maxz + B - 1) / B);        /* alt:238-240 */
        for (int bx = x0; bx < x1; bx++) {
            for (int by = y0; by < y1; by++) {
                for (int bz = z0; bz < z1; bz++) {
                    int b = par_oracle_hash_index(g, bx, by, bz);
                    int c = g->count[b];                              /* alt:246-248 */
                    g->map[b * PAR_SLOTS + c] = i;                    /* alt:250-253 */
                    g->bins[b * PAR_SLOTS + c] = *a;                  /* alt:255-257 */
                    g->count[b] = (c + 1) & (PAR_SLOTS - 1);          /* alt:262-264: wraps at 8 */
                }
            }
        }
    }
}

/* ---- primary pass: alt:271-397 ------------------------------------------------------------------------------ */

void par_oracle_primary(const par_params* p, const par_oracle_grid* g, const par_sprite* sprites,
                        const int32_t* sprite_ids, par_pixel* gbuf, uint8_t* palidx, int row_begin, int row_end) {
    const int W = p->width, H = p->height, B = p->bin_size;
    for (int i = 0; i < W; i++) {                    /* alt:277: x outer */
        for (int j = row_begin; j < row_end; j++) {  /* alt:279: rows inner */
            int world_j = (short)(H - j);            /* alt:280 */
            par_pixel px;                            /* alt:281 */
            memset(&px, 0, sizeof(px));
            px.color.red = px.color.green = px.color.blue = p->background;
            int pal = PAR_PALIDX_BACKGROUND;
            int adjacent = 0;                        /* alt:282 */
            int bin_x = i / B;                       /* alt:287 */
            int closest = INT_MIN;                   /* alt:289 */
            for (int bin_z = 0; bin_z < g->gz; bin_z++) { /* alt:292 */
                int hit_in_bin = 0;                  /* alt:293 */
                int bin_y = j / B;                   /* alt:294 */
                int b = par_oracle_hash_index(g, bin_x, bin_y, bin_z); /* alt:296 */
                int cnt = g->count[b];               /* alt:297 */
                if (cnt == 0) adjacent = 0;          /* alt:298-300 */
                for (int k = 0; k < cnt; k++) {      /* alt:303 */
                    int s = b * PAR_SLOTS + k;
                    const par_aabb* a = &g->bins[s];
                    if (i >= a->px && i < a->px + a->ex && world_j > a->py + a->pz &&
                        world_j <= a->py + a->ey + a->pz + a->ez) {            /* alt:310-317 */
                        int e = g->map[s];                                      /* alt:318-319 */
                        const par_sprite* sp = &sprites[sprite_ids ? sprite_ids[e] : 0]; /* alt:321-322 */
                        int row = a->py + a->ey + a->pz + a->ez - world_j;      /* alt:324-326 */
                        int t = row * PAR_SPRITE_W + (i - a->px);               /* alt:330-332 */
                        int depth = a->py - a->pz + imin(0, a->ey - row) - sp->depth[t]; /* alt:336-341 */
                        if (closest >= depth) continue;                         /* alt:344-346 */
                        closest = depth;                                        /* alt:347 */
                        px.normal = sp->normal[t];                              /* alt:349-350 */
                        pal = sp->color[t];
                        px.color = p->palette[pal];                             /* alt:352-354 */
                        px.y = a->py + a->ey + a->ez - row - sp->depth[t];      /* alt:356-359 */
                        px.z = a->pz + sp->depth[t];                            /* alt:360-361 */
                        px.entity_index = e;                                    /* alt:363 */
                        hit_in_bin = 1;                                         /* alt:365 */
                    }
                }
                adjacent += hit_in_bin;              /* alt:368 */
                if (adjacent >= 2) break;            /* alt:372-374 */
            }
            gbuf[(size_t)j * W + i] = px;            /* alt:379 */
            if (palidx) palidx[(size_t)j * W + i] = (uint8_t)pal;
        }
    }
}

/* ---- shadow walk: alt:399-500 ------------------------------------------------------------------------------- */

int par_oracle_shadow(const par_oracle_grid* g, int sx, int sy, int sz, int ex, int ey, int ez, int start_entity,
                      const par_ray* ray, int64_t* probes) {
    float bsx = (float)sx, bsy = (float)sy, bsz = (float)sz;         /* alt:406-408 */
    float bex = (float)ex, bey = (float)ey, bez = (float)ez;         /* alt:410-412 */
    float dx = bex - bsx, dy = bey - bsy, dz = bez - bsz;            /* alt:414-416 */
    /* std::max<float>({|dx|,|dy|,|dz|}), alt:419-421 */
    float largest = __builtin_fabsf(dx);
    if (largest < __builtin_fabsf(dy)) largest = __builtin_fabsf(dy);
    if (largest < __builtin_fabsf(dz)) largest = __builtin_fabsf(dz);
    float stx = dx / largest, sty = dy / largest, stz = dz / largest; /* alt:423-425 */
    float cx = bsx, cy = bsy, cz = bsz;                               /* alt:427 */
    float tx = bsx, ty = bsy, tz = bsz;                               /* alt:428 */
    int counter = 0;                                                  /* alt:429 */
    int start = par_oracle_hash_index(g, sx, sy, sz);                 /* alt:430 */
    int64_t nprobe = 0;
    int lit = 1;
    for (int i = 0; i < (int)largest;) {                              /* alt:432 */
        cx = tx; cy = ty; cz = tz;                                    /* alt:436 */
        if (counter == 0) { cx = tx + stx; counter++; }               /* alt:438-440 */
        else if (counter == 1) { cy = ty + sty; counter++; }          /* alt:441-443 */
        else if (counter == 2) { cz = tz + stz; counter++; }          /* alt:444-446 */
        else if (counter == 3) { cx = tx + stx; cy = ty + sty; counter++; } /* alt:447-450 */
        else if (counter == 4) { cx = tx + stx; cz = tz + stz; counter++; } /* alt:451-454 */
        else if (counter == 5) { cy = ty + sty; cz = tz + stz; counter++; } /* alt:455-458 */
        else {                                                        /* alt:459-466 */
            cx = cx + stx; cy = cy + sty; cz = cz + stz;
            tx = cx; ty = cy; tz = cz;
            counter = 0;
            i++;
        }
        int b = par_oracle_hash_index(g, (int)cx, (int)cy, (int)cz);  /* alt:468-470 */
        nprobe++;
        if (start == b) continue;                                     /* alt:471-473 */
        if (b < 0 || b >= g->volume) continue; /* defined: out-of-range flat index == empty bin (UB at alt:476) */
        if (g->count[b] > 0) {                                        /* alt:476 */
            for (int j = 0; j < g->count[b]; j++) {                   /* alt:480 */
                int s = b * PAR_SLOTS + j;
                if (start_entity == g->map[s]) continue;              /* alt:484-487 */
                if (par_oracle_intersect(&g->bins[s], ray)) {         /* alt:489-491 */
                    lit = 0;
                    goto done;
                }
            }
        }
    }
done:
    if (probes) *probes += nprobe;
    return lit;
}

/* ---- shading + quantise: alt:702-760 ------------------------------------------------------------------------ */

void par_oracle_shade(const par_params* p, const par_oracle_grid* g, const par_pixel* gbuf, const par_light* light,
                      par_color* fb, float* brightness, uint8_t* lit_plane, int row_begin, int row_end) {
    const int W = p->width, H = p->height, B = p->bin_size;
    const float ambient = p->ambient;                                  /* alt:702 */
    for (size_t i = (size_t)row_begin * W; i < (size_t)row_end * W; i++) { /* alt:703 */
        const par_pixel* px = &gbuf[i];
        par_vec3 n = px->normal;                                       /* alt:705 */
        int wx = (int)(i % (size_t)W);                                 /* alt:707 */
        int wy = px->y, wz = px->z;                                    /* alt:708-709 */
        par_vec3 d = {(float)(light->x - wx), (float)(light->y - wy), (float)(light->z - wz)};
        par_vec3 t = par_oracle_normalize(d);                          /* alt:711-715 */
        par_ray ray;
        ray.inv_x = 1.f / t.x; ray.inv_y = 1.f / t.y; ray.inv_z = 1.f / t.z; /* alt:717-719 */
        ray.ox = (short)wx; ray.oy = (short)wy; ray.oz = (short)wz;    /* alt:720-722 */
        ray.pad_ = 0;
        int rbx = wx / B, rby = (H - wy - wz) / B, rbz = wz / B;       /* alt:724-727 */
        int lbx = light->x / B, lby = (H - light->y - light->z) / B, lbz = light->z / B; /* alt:729-732 */
        float bright = ambient;
        par_color out = par_oracle_color_scale(px->color, ambient);    /* alt:735 */
        int lit = par_oracle_shadow(g, rbx, rby, rbz, lbx, lby, lbz, px->entity_index, &ray, NULL); /* alt:738-742 */
        if (lit) {
            float dot = n.x * t.x + n.y * t.y + n.z * t.z;             /* alt:746-747 */
            float diffuse = std_max(0.0f, dot);                        /* alt:745 */
            bright = std_min(1.f, diffuse + ambient);                  /* alt:758 */
            out = par_oracle_color_scale(px->color, bright);           /* alt:757-758 */
        }
        if (fb) fb[i] = out;
        if (brightness) brightness[i] = bright;
        if (lit_plane) lit_plane[i] = (uint8_t)lit;
    }
}

/* ---- whole frame: alt:690-760 ------------------------------------------------------------------------------- */

static int grid_alloc(const par_params* p, par_oracle_grid* g) {
    par_oracle_grid_dims(p, &g->gx, &g->gy, &g->gz);
    g->volume = g->gx * g->gy * g->gz;
    g->count = (int32_t*)calloc((size_t)g->volume, sizeof(int32_t));
    g->map = (int32_t*)calloc((size_t)g->volume * PAR_SLOTS, sizeof(int32_t));
    g->bins = (par_aabb*)calloc((size_t)g->volume * PAR_SLOTS, sizeof(par_aabb));
    return (g->count && g->map && g->bins) ? 0 : -1;
}

static void grid_free(par_oracle_grid* g) {
    free(g->count);
    free(g->map);
    free(g->bins);
}

static int params_ok(const par_params* p) {
    return p && p->width > 0 && p->height > 0 && p->length > 0 && p->bin_size > 0 && p->width <= 32767 &&
           p->height <= 32767 && p->length <= 32767 && p->ambient >= 0.f && p->ambient <= 1.f &&
           p->palette_size > 0 && p->palette_size <= PAR_MAX_PALETTE;
}

typedef struct mt_job {
    const par_params* p;
    const par_oracle_grid* g;
    const par_sprite* sprites;
    const int32_t* sprite_ids;
    const par_light* light;
    par_color* fb;
    par_pixel* gbuf;
    uint8_t* palidx;
    float* brightness;
    uint8_t* lit;
    int row_begin, row_end;
} mt_job;

static void* mt_worker(void* arg) {
    mt_job* j = (mt_job*)arg;
    par_oracle_primary(j->p, j->g, j->sprites, j->sprite_ids, j->gbuf, j->palidx, j->row_begin, j->row_end);
    par_oracle_shade(j->p, j->g, j->gbuf, j->light, j->fb, j->brightness, j->lit, j->row_begin, j->row_end);
    return NULL;
}

int par_oracle_render_mt(const par_params* p, const par_aabb* aabbs, int n, const par_sprite* sprites,
                         const int32_t* sprite_ids, const par_light* light, par_color* fb, par_pixel* gbuf,
                         uint8_t* palidx, float* brightness, uint8_t* lit, int nthreads) {
    if (!params_ok(p) || n < 0 || !light || !sprites) return -1;
    par_oracle_grid g;
    if (grid_alloc(p, &g) != 0) {
        grid_free(&g);
        return -1;
    }
    par_pixel* own_gbuf = NULL;
    if (!gbuf) {
        own_gbuf = (par_pixel*)malloc((size_t)p->width * p->height * sizeof(par_pixel));
        if (!own_gbuf) {
            grid_free(&g);
            return -1;
        }
        gbuf = own_gbuf;
    }
    par_oracle_bin(p, aabbs, n, &g); /* alt:690-693 */
    if (nthreads < 1) nthreads = 1;
    if (nthreads > p->height) nthreads = p->height;
    if (nthreads > 256) nthreads = 256;
    mt_job jobs[256];
    pthread_t tids[256];
    for (int t = 0; t < nthreads; t++) {
        mt_job j = {p, &g, sprites, sprite_ids, light, fb, gbuf, palidx, brightness, lit,
                    (int)((int64_t)p->height * t / nthreads), (int)((int64_t)p->height * (t + 1) / nthreads)};
        jobs[t] = j;
    }
    if (nthreads == 1) {
        mt_worker(&jobs[0]); /* faithful: one thread, as the reference runs */
    } else {
        for (int t = 0; t < nthreads; t++) pthread_create(&tids[t], NULL, mt_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(tids[t], NULL);
    }
    free(own_gbuf);
    grid_free(&g);
    return 0;
}

int par_oracle_render(const par_params* p, const par_aabb* aabbs, int n, const par_sprite* sprites,
                      const int32_t* sprite_ids, const par_light* light, par_color* fb, par_pixel* gbuf,
                      uint8_t* palidx, float* brightness, uint8_t* lit) {
    return par_oracle_render_mt(p, aabbs, n, sprites, sprite_ids, light, fb, gbuf, palidx, brightness, lit, 1);
}

/* ---- debug overlay: alt:139-175 called as alt:763-772 ------------------------------------------------------- */

void par_oracle_debug_line(const par_params* p, const par_pixel* gbuf, const par_light* light, int mouse_x,
                           int mouse_y, par_color* fb) {
    const int W = p->width, H = p->height;
    const par_pixel* pick = &gbuf[(size_t)mouse_y * W + mouse_x];  /* alt:380-382 */
    int x = mouse_x, y = H - (pick->y + pick->z);                  /* alt:764 */
    int x_end = light->x, y_end = H - (light->y + light->z);       /* alt:765 */
    int x_delta = abs(x_end - x), y_delta = -abs(y_end - y);       /* alt:143-144 */
    int x_sign = x < x_end ? 1 : -1, y_sign = y < y_end ? 1 : -1;  /* alt:149-150 */
    int error = x_delta + y_delta;                                 /* alt:152 */
    const par_color red = {255, 0, 0, 255};                        /* alt:772 */
    for (;;) {
        if (x >= 0 && y >= 0 && x < W && y < H) fb[x + (size_t)y * W] = red; /* alt:766-771 */
        if (x == x_end && y == y_end) return;                      /* alt:156-158 */
        int error2 = 2 * error;                                    /* alt:159 */
        if (error2 >= y_delta) {                                   /* alt:160-166 */
            if (x == x_end) return;
            error += y_delta;
            x += x_sign;
        }
        if (error2 <= x_delta) {                                   /* alt:167-173 */
            if (y == y_end) return;
            error += x_delta;
            y += y_sign;
        }
    }
}

/* ---- the tile sprite: spr:73-364 regenerated from its structure --------------------------------------------- */

void par_oracle_tile_floor(par_sprite* out) {
    for (int r = 0; r < PAR_SPRITE_H; r++) {
        for (int c = 0; c < PAR_SPRITE_W; c++) {
            int t = r * PAR_SPRITE_W + c;
            int col;
            if (r < 20) { /* top face, spr:74-94: a 2x2 checker of 6x6 blocks inside a border of 0 */
                int inner_r = (r >= 4 && r < 16), inner_c = (c >= 4 && c < 16);
                if (inner_r && inner_c) {
                    int upper = r < 10, left = c < 10;
                    col = (upper == left) ? 2 : 3;
                } else {
                    col = 0;
                }
                out->depth[t] = 19 - r;                       /* spr:117-137 */
                out->normal[t].x = 0.f; out->normal[t].y = 1.f; out->normal[t].z = 0.f; /* spr:200-280 */
            } else { /* front face, spr:95-114 */
                col = (r >= 38 || c < 2 || c >= 18) ? 1 : 2;
                out->depth[t] = 0;                            /* spr:138-198 */
                out->normal[t].x = 0.f; out->normal[t].y = 0.f; out->normal[t].z = -1.f; /* spr:281-361 */
            }
            out->color[t] = col;
        }
    }
}
