// Developer tool / GPU test: n-body_amd/csrc/exact_core.hip.h against hipcc's own IEEE expansions of sqrtf, 1.0f / b and
// a / b, on the GPU (the scalar functions and core_div2_rn, the packed pair the walk uses).  Exhaustive where the function has one operand, 2^34 random pairs plus mantissa corner cases for
// the division.  Prints the mismatch counts; exit status 1 if any.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../n-body_amd/csrc/exact_core.hip.h"
using namespace nbody;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void check_sqrt(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first)
{
#pragma clang fp contract(off)
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = __builtin_sqrtf(x);
        const float got = core_sqrt_rn(x, __builtin_amdgcn_sqrtf(x));
        if (__float_as_uint(want) != __float_as_uint(got)) { atomicAdd(bad, 1ull); atomicMin(first, (uint32_t)b); }
    }
}

__global__ void check_sqrt_newton(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first)
{
#pragma clang fp contract(off)
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = __builtin_sqrtf(x);
        const float got = core_sqrt_rn_newton(x, __builtin_amdgcn_sqrtf(x));
        if (__float_as_uint(want) != __float_as_uint(got)) { atomicAdd(bad, 1ull); atomicMin(first, (uint32_t)b); }
    }
}

__global__ void check_rcp(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first)
{
#pragma clang fp contract(off)
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = 1.0f / x;
        const float got = core_rcp_rn(x);
        if (__float_as_uint(want) != __float_as_uint(got)) { atomicAdd(bad, 1ull); atomicMin(first, (uint32_t)b); }
    }
}

// the walk's reciprocal of the distance: x = d^2 over the whole sqrt range, seeded from the vote's approximations
__global__ void check_rcp_seeded(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first)
{
#pragma clang fp contract(off)
    const float eps = 1e-6f;
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float s = __builtin_amdgcn_sqrtf(x);
        const float r0 = __builtin_amdgcn_rcpf(s + eps);
        const float d = core_sqrt_rn(x, s);
        const float want = 1.0f / (d + eps);
        const float got = core_rcp_rn_seeded(d + eps, r0);
        // the packed form the walk uses (lane x; lane y carries an unrelated division)
        const float got2 = core_div2_rn(core_f2{1.0f, 3.0f}, core_f2{d + eps, 7.0f}, core_f2{r0, __builtin_amdgcn_rcpf(7.0f)}).x;
        if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(want) != __float_as_uint(got2)) { atomicAdd(bad, 1ull); atomicMin(first, (uint32_t)b); }
    }
}

__device__ __forceinline__ uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// operands: a in [2^-40, 2^68] (node masses), b in [2^-20, 2^81] (softened squared distances).  mode 0: random mantissas
// and exponents; mode 1: corner mantissas (all ones, all zeros, single bits, alternating) crossed over all exponents
__global__ void check_div(int mode, uint64_t per_thread, unsigned long long* bad, unsigned long long* first)
{
#pragma clang fp contract(off)
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = 0x1234567ull + tid * 0x9E3779B97F4A7C15ull;
    const uint32_t corner[12] = {0u, 0x7FFFFFu, 0x7FFFFEu, 1u, 2u, 0x400000u, 0x400001u, 0x3FFFFFu, 0x555555u, 0x2AAAAAu, 0x7FF000u, 0x000FFFu};
    for (uint64_t k = 0; k < per_thread; ++k) {
        const uint64_t r = splitmix(s);
        uint32_t ma = (uint32_t)r & 0x7FFFFFu, mb = (uint32_t)(r >> 23) & 0x7FFFFFu;
        if (mode == 1) { ma = corner[(r >> 46) % 12] ^ ((r >> 60) & 1u); mb = corner[(r >> 50) % 12] ^ ((r >> 61) & 1u); ma &= 0x7FFFFFu; mb &= 0x7FFFFFu; }
        const uint32_t ea = 127 - 40 + (uint32_t)((r >> 32) % 109);       // 2^-40 .. 2^68
        const uint32_t eb = 127 - 20 + (uint32_t)((r >> 40) % 102);       // 2^-20 .. 2^81
        const float a = __uint_as_float((ea << 23) | ma), b = __uint_as_float((eb << 23) | mb);
        const float want = a / b;
        const float got = core_div_rn(a, b);
        // the packed form the walk uses (lane y; lane x carries an unrelated reciprocal)
        const float got2 = core_div2_rn(core_f2{1.0f, a}, core_f2{3.0f, b}, core_f2{__builtin_amdgcn_rcpf(3.0f), __builtin_amdgcn_rcpf(b)}).y;
        if (__float_as_uint(want) != __float_as_uint(got) || __float_as_uint(want) != __float_as_uint(got2)) {
            atomicAdd(bad, 1ull);
            atomicMin(first, ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(b));
        }
    }
}

int main()
{
    unsigned long long* bad; uint32_t* first32; unsigned long long* first64;
    CHECK(hipMalloc(&bad, 6 * sizeof(*bad)));
    CHECK(hipMalloc(&first32, 4 * sizeof(uint32_t)));
    CHECK(hipMalloc(&first64, 2 * sizeof(*first64)));
    CHECK(hipMemset(bad, 0, 6 * sizeof(*bad)));
    CHECK(hipMemset(first32, 0xFF, 4 * sizeof(uint32_t)));
    CHECK(hipMemset(first64, 0xFF, 2 * sizeof(*first64)));
    union { float f; uint32_t u; } lo, hi;
    lo.f = kCoreD2Min; hi.f = kCoreD2Max;
    check_sqrt<<<4096, 256>>>(lo.u, hi.u, bad + 0, first32 + 0);                      // every float in [2^-96, 2^80]
    const uint64_t n_sqrt = (uint64_t)hi.u - lo.u + 1;
    check_sqrt_newton<<<4096, 256>>>(lo.u, hi.u, bad + 5, first32 + 3);              // the four-instruction form, same range
    check_rcp_seeded<<<4096, 256>>>(lo.u, hi.u, bad + 4, first32 + 2);               // the same x: 1 / (sqrt(x) + 1e-6), seeded
    lo.f = 0x1p-21f; hi.f = 0x1p41f;                                               // d + 1e-6 for d in [2^-48, 2^40], and beyond
    check_rcp<<<4096, 256>>>(lo.u, hi.u, bad + 1, first32 + 1);
    const uint64_t n_rcp = (uint64_t)hi.u - lo.u + 1;
    const uint64_t threads = 4096ull * 256ull, per = (1ull << 34) / threads;
    check_div<<<4096, 256>>>(0, per, bad + 2, first64 + 0);
    check_div<<<4096, 256>>>(1, per / 16, bad + 3, first64 + 1);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[6]; uint32_t f32[4]; unsigned long long f64[2];
    CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(f32, first32, sizeof(f32), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(f64, first64, sizeof(f64), hipMemcpyDeviceToHost));
    std::printf("sqrt: %llu inputs (every float in [2^-96, 2^80]), %llu differ from sqrtf (first bits 0x%08x)\n", (unsigned long long)n_sqrt, h[0], f32[0]);
    std::printf("sqrt, one Newton step: %llu inputs (the same), %llu differ from sqrtf (first bits 0x%08x)\n", (unsigned long long)n_sqrt, h[5], f32[3]);
    std::printf("rcp:  %llu inputs (every float in [2^-21, 2^41]), %llu differ from 1.0f / b (first bits 0x%08x)\n", (unsigned long long)n_rcp, h[1], f32[1]);
    std::printf("rcp seeded: %llu inputs (every d^2 in [2^-96, 2^80]), %llu differ from 1.0f / (sqrtf(d^2) + 1e-6f) (first bits 0x%08x)\n", (unsigned long long)n_sqrt, h[4], f32[2]);
    std::printf("div:  %llu random pairs a in [2^-40, 2^68], b in [2^-20, 2^81]: %llu differ from a / b (first 0x%016llx)\n", threads * per, h[2], f64[0]);
    std::printf("div:  %llu corner-mantissa pairs over the same exponents: %llu differ from a / b (first 0x%016llx)\n", threads * (per / 16), h[3], f64[1]);
    return (h[0] | h[1] | h[2] | h[3] | h[4] | h[5]) ? 1 : 0;
}
