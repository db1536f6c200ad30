// exact_core.hip.h — correctly rounded fp32 sqrt, reciprocal and division WITHOUT the range and special-case
// instructions that hipcc's own IEEE expansions carry, for operands known to be in range.
//
// The Barnes-Hut node kernel (benchmark.cpp:188-225) is, per accepted (node, target) pair, one sqrt and two divisions
// that must be correctly rounded for the accelerations to equal the oracle's bit for bit; as hipcc expands them they are
// 15 + 11 + 11 of the ~45 VALU instructions of an accepted term.  Those expansions are a short FMA core wrapped in
// instructions that only act on operands outside the normal range:
//   sqrt:  v_sqrt_f32 (1 ulp), then s-1ulp / s+1ulp are tried against the residuals x - s'*s (2 integer adds, 2 FMAs,
//          2 compares, 2 selects).  Around it: input scaling by 2^32 below 2^-96 (5 instructions) and a class test that
//          passes 0 and inf through (2).
//   a / b: v_rcp_f32, one Newton step on the reciprocal (2 FMAs), quotient, two residual corrections (1 multiply,
//          4 FMAs, the last one as v_div_fmas).  Around it: v_div_scale x 2 (scale either operand by 2^+-64 when an
//          exponent is extreme or the quotient would be subnormal) and v_div_fixup (NaN / inf / 0 / overflow cases).
// With VCC = 0 v_div_fmas is a plain FMA, v_div_scale returns its operand unchanged and v_div_fixup returns its first
// operand (CDNA3/4 ISA, V_DIV_SCALE_F32 / V_DIV_FMAS_F32 / V_DIV_FIXUP_F32), so on operands that trigger none of the
// scaling conditions the cores below are the SAME instruction sequences and return the same bits:
//   sqrt:  2^-96 <= x < inf
//   a / b: a, b positive and finite, b normal and < 2^126, -126 < exponent(a) - exponent(b) < 96, a >= 2^-103.
// The callers guarantee these ranges (host-checked masses and radii, a per-wave test of d^2);
// tools/exact_core_check.hip compares the cores with hipcc's expansions on the GPU — every float of the sqrt and
// reciprocal ranges, 2^34 operand pairs and the mantissa corner cases for the division — and runs as a GPU test.
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {

constexpr float kCoreD2Min = 0x1p-96f;     // squared distances the cores accept (per-wave test in the walk kernels)
constexpr float kCoreD2Max = 0x1p80f;
constexpr float kCoreMassMin = 0x1p-40f;   // body masses (0 also allowed) and radii the host checks at upload:
constexpr float kCoreMassMax = 0x1p40f;    // node masses are sums of at most 2^28 of them -> [2^-40, 2^68]

// sqrt(x) correctly rounded, given s = v_sqrt_f32(x) (the caller usually has it already, for the vote)
__device__ __forceinline__ float core_sqrt_rn(float x, float s)
{
#pragma clang fp contract(off)
    const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
    float r = 0.0f >= rdn ? dn : s;
    r = 0.0f < rup ? up : r;
    return r;
}

// The same in four instructions instead of eight: one Newton correction s + (x - s^2) * (0.5 / sqrt(x)) with the exact
// residual and the hardware's 1-ulp reciprocal square root.  Its error before the final rounding is a few 2^-24 of an
// ulp; whether any float's square root lies that close to a rounding boundary is not argued but COUNTED:
// tools/exact_core_check.hip compares it with sqrtf for every float of [2^-96, 2^80].
__device__ __forceinline__ float core_sqrt_rn_newton(float x, float s)
{
#pragma clang fp contract(off)
    const float r = __builtin_fmaf(-s, s, x);
    const float hy = 0.5f * __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(r, hy, s);
}

// a / b correctly rounded
__device__ __forceinline__ float core_div_rn(float a, float b)
{
#pragma clang fp contract(off)
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float t = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(t, r, q);
}

// 1 / b correctly rounded (core_div_rn with a = 1: the quotient estimate is the refined reciprocal itself)
__device__ __forceinline__ float core_rcp_rn(float b)
{
#pragma clang fp contract(off)
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    float t = __builtin_fmaf(-b, q, 1.0f);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, 1.0f);
    return __builtin_fmaf(t, r, q);
}

// 1 / (d + eps) correctly rounded for d = core_sqrt_rn(x, s), starting the Newton step from r0 = v_rcp_f32(s + eps) — the
// reciprocal the walk's vote has already taken of the 1-ulp distance s = v_sqrt_f32(x) — instead of a new
// v_rcp_f32(d + eps).  r0 is within 3 ulp of 1 / (d + eps); the Newton step squares that error away and the two
// residual corrections do the rounding.  Not the compiler's sequence any more, so not identical by construction:
// tools/exact_core_check.hip checks it against 1.0f / (d + eps) for EVERY x of the sqrt range.
__device__ __forceinline__ float core_rcp_rn_seeded(float b, float r0)
{
#pragma clang fp contract(off)
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    float q = r;
    float t = __builtin_fmaf(-b, q, 1.0f);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, 1.0f);
    return __builtin_fmaf(t, r, q);
}

// {a.x / b.x, a.y / b.y}, both correctly rounded, as one packed sequence: core_div_rn on two lanes, each starting from
// the reciprocal estimate r0 it is given — v_rcp_f32(b), or for core_rcp_rn_seeded's case (a.x = 1) the estimate that
// function documents (1 * r is r: the same values as its q = r).
typedef float core_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ core_f2 core_div2_rn(core_f2 a, core_f2 b, core_f2 r0)
{
#pragma clang fp contract(off)
    const core_f2 one = {1.0f, 1.0f};
    const core_f2 e = __builtin_elementwise_fma(-b, r0, one);
    const core_f2 r = __builtin_elementwise_fma(e, r0, r0);
    core_f2 q = a * r;
    core_f2 t = __builtin_elementwise_fma(-b, q, a);
    q = __builtin_elementwise_fma(t, r, q);
    t = __builtin_elementwise_fma(-b, q, a);
    return __builtin_elementwise_fma(t, r, q);
}

}  // namespace nbody
