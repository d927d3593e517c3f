// par_fastdiv.h — short division sequences that give EXACTLY the IEEE quotient on the operands of the shading loop.
//
// hipcc's correctly rounded fp32 division is ten to eleven dependent instructions (scale, reciprocal, four fused
// refinements, fix-up) and the shading does six per pixel (Vector::normalize spr:28-35: x / L1 length for three
// components; alt:717-719: 1 / each). On the operands that occur there, a reciprocal and two (three) fused
// refinements round to the same float, which tools/divcheck.hip — built from THIS header — checks exhaustively on
// the GPU: every a / b with integers |a| <= 65535, 1 <= b <= 196605, |a| <= b (differences of `short` coordinates
// over their L1 length), and every 1 / t with 2^-24 <= |t| <= 2^24, t = +-0, +-inf or NaN (t is such a quotient).
// Callers check the operand ranges and take the ordinary division outside them
// (tests/test_gpu_parity.py::test_short_division_sequences_are_exact runs the check).
#ifndef PAR_FASTDIV_H
#define PAR_FASTDIV_H

#include <hip/hip_runtime.h>

constexpr float PAR_FASTDIV_MAX_NUM = 65535.0f;   // |a|
constexpr float PAR_FASTDIV_MAX_DEN = 196605.0f;  // b = sum of three such magnitudes

// a / b, given y = rcp(b) (one reciprocal serves the three components of a normalize).
__device__ __forceinline__ float par_fast_div(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}

// 1 / t.
__device__ __forceinline__ float par_fast_rcp(float t) {
    const float y = __builtin_amdgcn_rcpf(t);
    const float e = __builtin_fmaf(-t, y, 1.0f);
    const float y1 = __builtin_fmaf(e, y, y);
    // zero, infinity and NaN: the hardware reciprocal is the IEEE answer already (and e is NaN there)
    const float at = __builtin_fabsf(t);
    return (at > 0.0f && at < __builtin_inff()) ? y1 : y;
}

#endif
