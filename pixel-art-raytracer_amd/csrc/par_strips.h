// par_strips.h — the order in which a render work item's wavefront visits the pixels of a rectangle.
//
// A rectangle of rw x rh pixels (rw, rh <= PAR_STRIP_MAX_SIDE: a whole bin tile, or one entry's footprint clipped to
// it) is visited in vertical STRIPS of a sprite's width (20 pixels; the last strip takes what is left), row by row
// within a strip: pixel p = 0 .. rw * rh - 1 lies in strip p / (20 * rh), at row (p mod (20 * rh)) / w and column
// strip * 20 + (p mod (20 * rh)) mod w, w the strip's width. A 64-pixel chunk is then a few rows of ONE strip instead
// of one or two rows across the whole width, and fewer entries' rectangles meet it (a full floor: 2.4 instead of
// 4.4 candidate entries per chunk).
//
// The divisions run through float reciprocals (an integer division by a run-time value costs some forty
// instructions per wavefront): floor(p / d) = (int)((p + 0.5) * rcp(d)). p < 2^15 and d <= 160^2, so (p + 0.5) / d
// is at least 0.5 / 25 600 away from every integer, far more than the error of the hardware reciprocal (1 ulp) and
// two roundings on a quotient below 2^15. tools/stripcheck.hip — built from THIS header — checks every pixel of
// every rectangle shape against integer arithmetic on the GPU
// (tests/test_gpu_parity.py::test_strip_order_visits_every_pixel_once runs it).
#ifndef PAR_STRIPS_H
#define PAR_STRIPS_H

#include <hip/hip_runtime.h>

constexpr int PAR_STRIP_W = 20;           // a sprite's width (PAR_SPRITE_W)
constexpr int PAR_STRIP_MAX_SIDE = 160;   // the largest bin size (PAR_MAX_BIN)

struct par_strips {
    int n_strips;  // ceil(rw / 20)
    int sw, lw;    // width of a strip, of the last strip
};

// (wave-uniform arguments give wave-uniform results)
__device__ __forceinline__ par_strips par_strips_of(int rw) {
    par_strips s;
    s.n_strips = (int)(((uint32_t)(rw + PAR_STRIP_W - 1) * 3277u) >> 16);  // / 20, exact below 2^13
    s.sw = rw < PAR_STRIP_W ? rw : PAR_STRIP_W;
    s.lw = rw - (s.n_strips - 1) * s.sw;
    return s;
}

// Pixel p of the visit: its strip, and its column / row relative to the rectangle's corner. p may lie past the
// rectangle (the idle lanes of a last chunk, p < rw * rh + 64): the strip is then clamped and the row runs past
// rh; nothing else is promised for such a p.
__device__ __forceinline__ void par_strip_pixel(const par_strips& s, int rh, int p, int& strip, int& col, int& row) {
    int q = p, w = s.sw;
    float inv_w = __builtin_amdgcn_rcpf((float)s.sw);
    strip = 0;
    if (s.n_strips > 1) {
        const int strip_px = s.sw * rh;
        const int k = (int)(((float)p + 0.5f) * __builtin_amdgcn_rcpf((float)strip_px));
        strip = k < s.n_strips - 1 ? k : s.n_strips - 1;
        q = p - __mul24(strip, strip_px);  // (all of these products are below 2^24: full-rate multiplies)
        const bool last = strip == s.n_strips - 1;
        w = last ? s.lw : s.sw;
        inv_w = last ? __builtin_amdgcn_rcpf((float)s.lw) : inv_w;
    }
    row = (int)(((float)q + 0.5f) * inv_w);
    col = __mul24(strip, s.sw) + (q - __mul24(row, w));
}

#endif
