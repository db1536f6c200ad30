// ubench_mfma.hip — does the fp32 MFMA pipe co-issue with packed-fp32 VALU work on gfx950 (MI355X)?
// VERDICT r1 item 4 / DESIGN.md §9(a): the pair kernel's two accumulate v_pk_fma_f32 per 2 interactions could move to
// the matrix pipe as v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4x1 * 1x4 outer product: lane l supplies A = f_ij of its
// own target and B = one of {x_j, y_j, 1, 0}; D accumulates [sum f*x_j, sum f*y_j, sum f]).  That only pays if an MFMA
// costs the wave (much) less VALU issue time than the 4.2-cycle v_pk_fma_f32 it replaces.  Cases:
//   pk / mfma alone; N pk + M mfma interleaved in ONE wave's instruction stream; pk-only waves next to mfma-only waves
//   on the same SIMD; and the pair loop's real mix (6 pk + 2 rsq [+ 2 pk | + 2 mfma]).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 20000;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define PK(r) "v_pk_fma_f32 %" #r ", %" #r ", %[pa], %[pb]\n"
#define MF(r) "v_mfma_f32_4x4x1_16b_f32 %" #r ", %[a], %[b], %" #r "\n"
#define RSQ(r) "v_rsq_f32 %" #r ", %" #r "\n"

#define DECL                                                                                             \
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;                                               \
    f2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7}; \
    f2 p4 = {seed, seed + 2}, p5 = {seed + 3, seed + 3}, p6 = {seed + 5, seed + 5}, p7 = {seed + 7, seed + 7}; \
    f2 pa = {a, a}, pb = {b, b};                                                                         \
    f4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0}, m3 = {0, 0, 0, 0};                       \
    float t0 = seed + 0.5f, t1 = seed + 1.5f;                                                            \
    (void)t0; (void)t1
#define OUT out[gid] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x + m0.x + m1.y + m2.z + m3.w + t0 + t1
#define REGS "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(t0), "+v"(t1)
#define INS [pa] "v"(pa), [pb] "v"(pb), [a] "v"(a), [b] "v"(b)

// each kernel: `instr` VALU-issued instructions per loop iteration
__global__ void k_pk8(float* out, float seed, float a, float b)       // 8 pk
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) : REGS : INS); OUT; }
__global__ void k_pk6(float* out, float seed, float a, float b)       // 6 pk
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) : REGS : INS); OUT; }
__global__ void k_mf4(float* out, float seed, float a, float b)       // 4 mfma
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(MF(8) MF(9) MF(10) MF(11) : REGS : INS); OUT; }
__global__ void k_mf2(float* out, float seed, float a, float b)       // 2 mfma
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(MF(8) MF(9) : REGS : INS); OUT; }
__global__ void k_pk6mf2(float* out, float seed, float a, float b)    // 6 pk + 2 mfma, interleaved
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) MF(8) PK(3) PK(4) PK(5) MF(9) : REGS : INS); OUT; }
__global__ void k_pk6mf4(float* out, float seed, float a, float b)    // 6 pk + 4 mfma
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) MF(8) PK(2) MF(9) PK(3) PK(4) MF(10) PK(5) MF(11) : REGS : INS); OUT; }
__global__ void k_real_valu(float* out, float seed, float a, float b) // today's mix: 8 pk + 2 rsq
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) PK(3) RSQ(12) RSQ(13) PK(4) PK(5) PK(6) PK(7) : REGS : INS); OUT; }
__global__ void k_real_mfma(float* out, float seed, float a, float b) // candidate: 6 pk + 2 rsq + 2 mfma
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) PK(3) RSQ(12) RSQ(13) PK(4) PK(5) MF(8) MF(9) : REGS : INS); OUT; }
__global__ void k_real_6only(float* out, float seed, float a, float b) // lower bound: 6 pk + 2 rsq, accumulate dropped
{ DECL; for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) PK(3) RSQ(12) RSQ(13) PK(4) PK(5) : REGS : INS); OUT; }
// waves of one workgroup alternate: even waves pk-only, odd waves mfma-only (2 workgroups of 8 waves per CU =>
// every SIMD holds both kinds)
__global__ void k_split(float* out, float seed, float a, float b)
{
    DECL;
    if ((threadIdx.x >> 6) & 1) { for (int i = 0; i < kIters; ++i) asm volatile(MF(8) MF(9) MF(10) MF(11) : REGS : INS); }
    else                        { for (int i = 0; i < kIters; ++i) asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) : REGS : INS); }
    OUT;
}

typedef void (*kern_t)(float*, float, float, float);
struct Case { const char* name; kern_t k; int pk, mf, rsq; };

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d\n", prop.name, cus);
    float* out;
    CK(hipMalloc(&out, sizeof(float) * (size_t)cus * 16 * 512));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Case cases[] = {{"8 pk_fma", k_pk8, 8, 0, 0},           {"6 pk_fma", k_pk6, 6, 0, 0},
                    {"4 mfma4x4x1", k_mf4, 0, 4, 0},        {"2 mfma4x4x1", k_mf2, 0, 2, 0},
                    {"6 pk + 2 mfma", k_pk6mf2, 6, 2, 0},   {"6 pk + 4 mfma", k_pk6mf4, 6, 4, 0},
                    {"8 pk + 2 rsq (today)", k_real_valu, 8, 0, 2}, {"6 pk + 2 rsq + 2 mfma", k_real_mfma, 6, 2, 2},
                    {"6 pk + 2 rsq", k_real_6only, 6, 0, 2}};
    printf("%-24s %6s %10s %22s\n", "case (per iteration)", "w/SIMD", "ms", "SIMD cycles/iteration*");
    for (auto& c : cases) {
        for (int wps : {1, 2, 4, 8}) {
            dim3 grid(cus * wps), block(256);          // 256 threads = 4 waves = 1 per SIMD
            c.k<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            c.k<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-24s %6d %10.4f %22.2f\n", c.name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)kIters * wps));
        }
    }
    // split: 512-thread workgroups (8 waves: 2 per SIMD, one pk-only and one mfma-only), 1 and 2 workgroups per CU
    for (int wgs : {1, 2, 4}) {
        dim3 grid(cus * wgs), block(512);
        k_split<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k_split<<<grid, block>>>(out, 1.0f, 0.999f, 0.001f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("split: per SIMD %d wave(s) of 8 pk + %d wave(s) of 4 mfma: %.4f ms = %.2f SIMD cycles per (8 pk + 4 mfma) pair of iterations\n",
               wgs, wgs, ms, ms * 1e-3 * 2.4e9 / ((double)kIters * wgs));
    }
    printf("* at a nominal 2.4 GHz; all waves of a SIMD together\n");
    CK(hipFree(out));
    return 0;
}
