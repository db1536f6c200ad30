// ubench.hip — VALU issue-rate microbenchmarks for gfx950 (MI355X).
// Answers the design questions of DESIGN.md §kernels: what do v_fma_f32 / v_pk_fma_f32 / v_rsq_f32 /
// v_min_f32 cost per wave64 at 1..8 waves per SIMD, and does v_rsq_f32 overlap with full-rate VALU?
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 40000;

// 8 independent accumulators per instruction kind; every asm block = 8 instructions.
#define DECL8 float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7
#define OUT8 out[gid] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7
#define REGS8 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)

#define OP1x8(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define OP2x8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define FMAx8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"

__global__ void k_fma(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(FMAx8 : REGS8 : "v"(a), "v"(b));
    OUT8;
}
__global__ void k_mul(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP2x8("v_mul_f32") : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_min(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP2x8("v_min_f32") : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_rsq(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP1x8("v_rsq_f32") : REGS8);
    OUT8;
}
typedef float float2v __attribute__((ext_vector_type(2)));
#define DECL4P float2v p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7}; float2v pa = {a, a}, pb = {b, b}
#define OUT4P out[gid] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y
#define REGS4P "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
__global__ void k_pkfma(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL4P;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     : REGS4P : "v"(pa), "v"(pb));
    OUT4P;
}
__global__ void k_pkmul(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL4P;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     : REGS4P : "v"(pa), "v"(pb));
    OUT4P;
}
__global__ void k_pkadd(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL4P;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                     "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                     : REGS4P : "v"(pa), "v"(pb));
    OUT4P;
}
// 6 fma + 2 rsq interleaved: if rsq overlapped with fma issue, cost ~ max; else sum.
__global__ void k_mix62(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rsq_f32 %6, %6\n"
                     "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_rsq_f32 %7, %7\n"
                     : REGS8 : "v"(a), "v"(b));
    OUT8;
}
// 7 fma + 1 rsq
__global__ void k_mix71(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %6, %6, %8, %9\n"
                     "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_rsq_f32 %7, %7\n"
                     : REGS8 : "v"(a), "v"(b));
    OUT8;
}
// fma with an SGPR operand (source broadcast from the scalar file)
__global__ void k_fma_sgpr(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : REGS8 : "s"(a), "v"(b));
    OUT8;
}
// dependent chain: latency of fma and rsq
__global__ void k_fma_dep(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float r0 = seed;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(r0) : "v"(a), "v"(b));
    out[gid] = r0;
}
__global__ void k_rsq_dep(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float r0 = seed;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n"
                     "v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n"
                     : "+v"(r0));
    out[gid] = r0;
}


__global__ void k_sub(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP2x8("v_sub_f32") : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_max(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP2x8("v_max_f32") : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_mul_clamp(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_mul_f32_e64 %0, %0, %8 clamp\n v_mul_f32_e64 %1, %1, %8 clamp\n v_mul_f32_e64 %2, %2, %8 clamp\n v_mul_f32_e64 %3, %3, %8 clamp\n"
                     "v_mul_f32_e64 %4, %4, %8 clamp\n v_mul_f32_e64 %5, %5, %8 clamp\n v_mul_f32_e64 %6, %6, %8 clamp\n v_mul_f32_e64 %7, %7, %8 clamp\n"
                     : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_mul_sgpr(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP2x8("v_mul_f32") : REGS8 : "s"(a));
    OUT8;
}
__global__ void k_fmac(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                     "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                     : REGS8 : "v"(a), "v"(b));
    OUT8;
}
__global__ void k_fmaak(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fmaak_f32 %0, %0, %8, 0x3aa2425\n v_fmaak_f32 %1, %1, %8, 0x3aa2425\n v_fmaak_f32 %2, %2, %8, 0x3aa2425\n v_fmaak_f32 %3, %3, %8, 0x3aa2425\n"
                     "v_fmaak_f32 %4, %4, %8, 0x3aa2425\n v_fmaak_f32 %5, %5, %8, 0x3aa2425\n v_fmaak_f32 %6, %6, %8, 0x3aa2425\n v_fmaak_f32 %7, %7, %8, 0x3aa2425\n"
                     : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_fma_inl(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, 1.0\n v_fma_f32 %1, %1, %8, 1.0\n v_fma_f32 %2, %2, %8, 1.0\n v_fma_f32 %3, %3, %8, 1.0\n"
                     "v_fma_f32 %4, %4, %8, 1.0\n v_fma_f32 %5, %5, %8, 1.0\n v_fma_f32 %6, %6, %8, 1.0\n v_fma_f32 %7, %7, %8, 1.0\n"
                     : REGS8 : "v"(a));
    OUT8;
}
__global__ void k_min3(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                     "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                     : REGS8 : "v"(a), "v"(b));
    OUT8;
}
__global__ void k_mov(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i) asm volatile(OP1x8("v_mov_b32") : REGS8);
    OUT8;
}
__global__ void k_cndmask(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    for (int i = 0; i < kIters; ++i)
        asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                     "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                     : REGS8 : "v"(a) : "vcc");
    OUT8;
}
// the planned pair step, 4 independent targets (x0..x3 reuse one source): sub sub fma fma rsq mul-clamp mul fma fma
__global__ void k_pair9(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float ax0 = 0, ay0 = 0, ax1 = 0, ay1 = 0, xi0 = seed, yi0 = seed + 1, xi1 = seed + 2, yi1 = seed + 3;
    float xj = a + threadIdx.x * 1e-3f, yj = b, tiny = 1e-36f;
    for (int i = 0; i < kIters; ++i) {
        float dx0, dy0, d0, r0, w0, dx1, dy1, d1, r1, w1;
        asm volatile(
            "v_sub_f32 %0, %14, %10\n v_sub_f32 %5, %14, %12\n"
            "v_sub_f32 %1, %15, %11\n v_sub_f32 %6, %15, %13\n"
            "v_fma_f32 %2, %1, %1, %16\n v_fma_f32 %7, %6, %6, %16\n"
            "v_fma_f32 %2, %0, %0, %2\n v_fma_f32 %7, %5, %5, %7\n"
            "v_rsq_f32 %3, %2\n v_rsq_f32 %8, %7\n"
            "v_mul_f32_e64 %4, %3, %3 clamp\n v_mul_f32_e64 %9, %8, %8 clamp\n"
            "v_mul_f32 %4, %3, %4\n v_mul_f32 %9, %8, %9\n"
            "v_fmac_f32 %17, %0, %4\n v_fmac_f32 %19, %5, %9\n"
            "v_fmac_f32 %18, %1, %4\n v_fmac_f32 %20, %6, %9\n"
            : "=&v"(dx0), "=&v"(dy0), "=&v"(d0), "=&v"(r0), "=&v"(w0), "=&v"(dx1), "=&v"(dy1), "=&v"(d1), "=&v"(r1), "=&v"(w1)
            : "v"(xi0), "v"(yi0), "v"(xi1), "v"(yi1), "v"(xj), "v"(yj), "v"(tiny), "v"(ax0), "v"(ay0), "v"(ax1), "v"(ay1));
        asm volatile("" : "+v"(ax0), "+v"(ay0), "+v"(ax1), "+v"(ay1));
    }
    out[gid] = ax0 + ay0 + ax1 + ay1;
}
// clock probe: shader-clock ticks vs 100 MHz realtime ticks around a long fma loop
__global__ void k_clock(float* out, float seed, float a, float b)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    DECL8;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < kIters; ++i) asm volatile(FMAx8 : REGS8 : "v"(a), "v"(b));
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    OUT8;
    if (threadIdx.x == 0) { ((unsigned long long*)out)[1024 + 2 * blockIdx.x] = c1 - c0; ((unsigned long long*)out)[1024 + 2 * blockIdx.x + 1] = t1 - t0; }
}

typedef void (*kern_t)(float*, float, float, float);
struct Case { const char* name; kern_t k; double lane_ops_per_instr; };

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    CK(hipMalloc(&out, sizeof(float) * (size_t)cus * 8 * 256 * 4 + 1024 * 8 + 16 * 8 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Case cases[] = {{"v_fma_f32", k_fma, 1},       {"v_fma_f32(sgpr)", k_fma_sgpr, 1}, {"v_mul_f32", k_mul, 1},
                    {"v_min_f32", k_min, 1},       {"v_rsq_f32", k_rsq, 1},            {"v_pk_fma_f32", k_pkfma, 2},
                    {"v_pk_mul_f32", k_pkmul, 2},  {"v_pk_add_f32", k_pkadd, 2},       {"6fma+2rsq", k_mix62, 1},
                    {"7fma+1rsq", k_mix71, 1},     {"fma dep-chain", k_fma_dep, 1},    {"rsq dep-chain", k_rsq_dep, 1},
                    {"v_sub_f32", k_sub, 1}, {"v_max_f32", k_max, 1}, {"v_mul_f32 clamp", k_mul_clamp, 1}, {"v_mul_f32(sgpr)", k_mul_sgpr, 1},
                    {"v_fmac_f32", k_fmac, 1}, {"v_fmaak_f32(lit)", k_fmaak, 1}, {"v_fma_f32(inl 1.0)", k_fma_inl, 1}, {"v_min3_f32", k_min3, 1},
                    {"v_mov_b32", k_mov, 1}, {"v_cndmask_b32", k_cndmask, 1}};
    printf("%-18s %5s %12s %14s %12s\n", "case", "w/SIMD", "ms", "cyc/instr/wave*", "Tlaneop/s");
    for (auto& c : cases) {
        for (int wps : {1, 2, 4, 8}) {
            // blocks of 256 threads = 4 waves = 1 wave per SIMD; wps blocks per CU
            dim3 grid(cus * wps), block(256);
            c.k<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int reps = 2;
            for (int r = 0; r < reps; ++r) c.k<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            const double instr_per_wave = 8.0 * kIters;
            // cycles per instruction per SIMD (all wps waves together), assuming 2.4 GHz
            const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * wps);
            const double rate = instr_per_wave * 64 * c.lane_ops_per_instr * 4.0 * cus * wps / (ms * 1e-3) / 1e12;
            printf("%-18s %5d %12.4f %14.2f %12.2f\n", c.name, wps, ms, cyc, rate);
        }
    }
    for (int wps : {1, 2, 4, 8}) {
        dim3 grid(cus * wps), block(256);
        k_pair9<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k_pair9<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double inter = 2.0 * kIters * 64.0 * 4 * cus * wps;
        printf("pair9 (18 instr / 2 interactions) w/SIMD=%d  %.4f ms  cyc/interaction/SIMD@2.4GHz=%.2f  %.1f Ginter/s\n", wps, ms,
               ms * 1e-3 * 2.4e9 / (2.0 * kIters * wps), inter / (ms * 1e-3) / 1e9);
    }
    for (int wps : {1, 2, 8}) {
        dim3 grid(cus * wps), block(256);
        CK(hipMemset(out, 0, sizeof(float) * (size_t)cus * 8 * 256 * 4));
        k_clock<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
        k_clock<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2 * cus * wps);
        CK(hipMemcpy(h.data(), (unsigned long long*)out + 1024, h.size() * 8, hipMemcpyDeviceToHost));
        double csum = 0, tsum = 0; for (size_t i = 0; i < h.size(); i += 2) { csum += h[i]; tsum += h[i + 1]; }
        printf("clock probe w/SIMD=%d: shader clock = %.1f MHz (memtime/memrealtime*100), cycles per fma/wave = %.2f\n", wps,
               csum / tsum * 100.0, csum / (h.size() / 2) / (8.0 * kIters));
    }
    CK(hipFree(out));
    return 0;
}
