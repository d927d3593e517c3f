// stripcheck.hip — every pixel of every rectangle shape a render work item can visit, through the strip order of
// csrc/par_strips.h (float reciprocals, the hardware's v_rcp_f32) against plain integer arithmetic, on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pixel-art-raytracer_amd/csrc -o /tmp/stripcheck tools/stripcheck.hip && /tmp/stripcheck
// For rw, rh = 1 .. 160 and p = 0 .. rw * rh - 1: strip, column and row equal the integer formulas, the column lies
// in [0, rw) and the row in [0, rh) (the integer order is a bijection onto the rectangle by construction: the
// strips' pixel counts add up to rw * rh); for the 64 values of p past the rectangle (idle lanes of a last chunk)
// the strip stays in range. Prints the counts; exit status 1 on any mismatch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "par_strips.h"

#define HIP_OK(x)                                                        \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(2);                                                \
        }                                                                \
    } while (0)

__global__ __launch_bounds__(256) void check(unsigned long long* counts) {
    const int rw = (int)blockIdx.x + 1, rh = (int)blockIdx.y + 1;
    const int area = rw * rh;
    const int n = (rw + PAR_STRIP_W - 1) / PAR_STRIP_W;
    const int sw = rw < PAR_STRIP_W ? rw : PAR_STRIP_W;
    const int lw = rw - (n - 1) * sw;
    const par_strips s = par_strips_of(rw);
    unsigned long long bad = 0, seen = 0;
    if (s.n_strips != n || s.sw != sw || s.lw != lw || lw < 1 || lw > sw) bad++;
    for (int p = (int)threadIdx.x; p < area + 64; p += (int)blockDim.x) {
        int strip, col, row;
        par_strip_pixel(s, rh, p, strip, col, row);
        if (p >= area) {  // an idle lane: only the strip is promised
            if (strip < 0 || strip >= n) bad++;
            continue;
        }
        int k = p / (sw * rh);
        if (k > n - 1) k = n - 1;
        const int q = p - k * sw * rh;
        const int w = (k == n - 1) ? lw : sw;
        const int row_ref = q / w, col_ref = k * sw + q % w;
        if (strip != k || row != row_ref || col != col_ref || col < 0 || col >= rw || row < 0 || row >= rh) bad++;
        seen++;
    }
    if (bad) atomicAdd(&counts[0], bad);
    atomicAdd(&counts[1], seen);
}

int main() {
    unsigned long long* d = nullptr;
    unsigned long long h[2] = {0, 0};
    HIP_OK(hipMalloc(&d, sizeof(h)));
    HIP_OK(hipMemset(d, 0, sizeof(h)));
    hipLaunchKernelGGL(check, dim3(PAR_STRIP_MAX_SIDE, PAR_STRIP_MAX_SIDE), dim3(256), 0, nullptr, d);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    HIP_OK(hipFree(d));
    const unsigned long long expect = (unsigned long long)(PAR_STRIP_MAX_SIDE * (PAR_STRIP_MAX_SIDE + 1) / 2) *
                                      (unsigned long long)(PAR_STRIP_MAX_SIDE * (PAR_STRIP_MAX_SIDE + 1) / 2);
    std::printf("strip order: %llu pixels of %d shapes checked (expected %llu): %llu mismatches\n", h[1],
                PAR_STRIP_MAX_SIDE * PAR_STRIP_MAX_SIDE, expect, h[0]);
    return (h[0] == 0 && h[1] == expect) ? 0 : 1;
}
