// ubench_mfma32.hip — can the pair kernel's d^2 come from the matrix pipe?  (round 3)
// MI355X_MICROARCH.md: an MFMA holds its SIMD's vector issue for 8 cycles whatever its size; the rest of its passes
// run beside the VALU.  tools/ubench_mfma.hip tried v_mfma_f32_4x4x1 (8 cycles in all: nothing left to overlap).
// v_mfma_f32_32x32x2_f32 is 64 cycles per SIMD: two of them give the 32 x 32 squared distances of 32 targets and 32
// sources, d^2 = [x_i, y_i, |r_i|^2, 1] . [-2 x_j, -2 y_j, 1, |r_j|^2], for 16 cycles of vector issue — against
// 8 x (2 v_pk_add + 2 v_pk_fma) = 134 cycles on the VALU today.  Per 1024 interactions and wave:
//   today      8 x (8 pk + 2 rsq)                       = 64 pk + 16 rsq
//   candidate  2 MFMA + 8 x (2 rsq + 2 pk_mul + 2 pk_fma + 1 pk_add)   (sums  S f x_j, S f y_j, S f;  a = S f x_j - x_i S f)
// Cases: candidate, its VALU part alone, its MFMAs alone, today's mix; 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma32.hip -o tools/ubench_mfma32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int kTiles = 4096;      // 32-source tiles per wave

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// BF: operands with the shapes a 3 x bf16 split of the fp32 coordinates would supply (8 bf16 per lane for A and for B:
// x_i . x_j as six products of pieces, |r_i|^2 and |r_j|^2 as three pieces each), B preformatted in LDS
template <bool BF> __device__ __forceinline__ f16v mfma_pair_step(float a, float b, bf8 av, bf8 bv, f16v c)
{
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    else              return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int MODE, bool BF = false, bool SCHED = false>   // 0 candidate, 1 VALU part alone, 2 MFMAs alone
__global__ __launch_bounds__(256) void k_cand(float* out, const float* __restrict__ src, float a1, float a2)
{
    __shared__ float4 tile[8 * 32];                 // per source: {x, y, B1, B2} (what a lane reads for its column)
    for (int q = threadIdx.x; q < 8 * 32; q += 256) tile[q] = make_float4(src[q & 1023], src[(q + 7) & 1023], src[(q + 3) & 1023], 1.0f);
    __shared__ bf8 btile[2][8 * 32];                // BF: the lane's 8 K-entries of its source for the two MFMAs
    for (int q = threadIdx.x; q < 2 * 8 * 32; q += 256)
        for (int k = 0; k < 8; ++k) (&btile[0][0])[q][k] = (__bf16)(src[(q + k) & 1023]);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf8 av1, av2;
    for (int k = 0; k < 8; ++k) { av1[k] = (__bf16)(a1 + (float)k); av2[k] = (__bf16)(a2 - (float)k); }
    f2 SX[8], SY[8], SF[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) { SX[g] = f2{0, 0}; SY[g] = f2{0, 0}; SF[g] = f2{0, 0}; }
    f16v tiny;
#pragma unroll
    for (int v = 0; v < 16; ++v) tiny[v] = 1e-36f;
    float4 s = tile[lane & 31];
    f16v acc = tiny;
    if (MODE != 1) {
        acc = mfma_pair_step<BF>(a1, s.z, av1, btile[0][lane & 31], tiny);
        acc = mfma_pair_step<BF>(a2, s.w, av2, btile[1][lane & 31], acc);
    }
    for (int t = 1; t <= kTiles; ++t) {
        const float4 sn = tile[((t & 7) << 5) | (lane & 31)];
        f16v nxt = tiny;
        if (MODE != 1) {
            const int bi = ((t & 7) << 5) | (lane & 31);
            nxt = mfma_pair_step<BF>(a1, sn.z, av1, BF ? btile[0][bi] : av1, tiny);
            nxt = mfma_pair_step<BF>(a2, sn.w, av2, BF ? btile[1][bi] : av2, nxt);
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) nxt[v] = acc[v] + sn.z;       // (keeps the chain alive; 16 plain adds, not in the candidate)
        }
        if (MODE != 2) {
            const f2 xj = {s.x, s.x}, yj = {s.y, s.y};
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f2 inv;
                inv.x = __builtin_amdgcn_rsqf(__builtin_fabsf(acc[2 * g]));
                inv.y = __builtin_amdgcn_rsqf(__builtin_fabsf(acc[2 * g + 1]));
                const f2 w = inv * inv;
                const f2 f = inv * w;
                SX[g] = fma2(f, xj, SX[g]);
                SY[g] = fma2(f, yj, SY[g]);
                SF[g] = SF[g] + f;
            }
            if (MODE == 0 && SCHED) {
                // hand placement: one MFMA, then half of the tile's vector work, the other MFMA, the rest
                // (masks: 0x8 MFMA, 0x2 VALU, 0x400 transcendental... the VALU mask covers both here)
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 28, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 28, 0);
            }
        } else {
            SF[0].x += acc[0] + acc[15];
        }
        acc = nxt;
        s = sn;
    }
    float r = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) r += SX[g].x + SX[g].y + SY[g].x + SY[g].y + SF[g].x + SF[g].y;
    out[blockIdx.x * 256 + threadIdx.x] = r + acc[3];
}

// today's mix on the same amount of work: per 32-source "tile" 8 x (2 pk_add + 2 pk_fma + 2 rsq + 2 pk_mul + 2 pk_fma)
__global__ __launch_bounds__(256) void k_today(float* out, const float* __restrict__ src, float a1, float a2)
{
    __shared__ float2 tile[64 * 8];
    for (int q = threadIdx.x; q < 64 * 8; q += 256) tile[q] = make_float2(src[q & 1023], src[(q + 7) & 1023]);
    __syncthreads();
    f2 X[2] = {{a1, a2}, {a2, a1}}, Y[2] = {{a2, a2}, {a1, a1}}, AX[2] = {{0, 0}, {0, 0}}, AY[2] = {{0, 0}, {0, 0}};
    const f2 tiny = {1e-36f, 1e-36f};
    for (int t = 1; t <= kTiles; ++t) {
        // 1024 interactions per wave = 4 targets per lane x 4 sources
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float2 s = tile[((t & 63) << 3) | u];          // wave-uniform: broadcast read
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f2 dx = f2{s.x, s.x} - X[k], dy = f2{s.y, s.y} - Y[k];
                f2 d2 = fma2(dy, dy, tiny);
                d2 = fma2(dx, dx, d2);
                f2 inv;
                inv.x = __builtin_amdgcn_rsqf(d2.x);
                inv.y = __builtin_amdgcn_rsqf(d2.y);
                const f2 w = inv * inv;
                const f2 f = inv * w;
                AX[k] = fma2(dx, f, AX[k]);
                AY[k] = fma2(dy, f, AY[k]);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = AX[0].x + AX[0].y + AX[1].x + AX[1].y + AY[0].x + AY[0].y + AY[1].x + AY[1].y;
}

typedef void (*kern_t)(float*, const float*, float, float);

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d\n", prop.name, cus);
    float *out, *src;
    CK(hipMalloc(&out, sizeof(float) * (size_t)cus * 8 * 256));
    CK(hipMalloc(&src, sizeof(float) * 1024));
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f + 0.37f * (float)(i % 97);
    CK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct { const char* name; kern_t k; } cases[] = {
        {"candidate: 2 mfma32x32x2 + 8x(2 rsq + 5 pk)", k_cand<0>}, {"its VALU part (+16 v_add)", k_cand<1>},
        {"its 2 MFMAs alone", k_cand<2>},
        {"bf16 candidate: 2 mfma32x32x16_bf16 + 8x(2 rsq + 5 pk) ", k_cand<0, true>}, {"its 2 bf16 MFMAs alone ", k_cand<2, true>},
        {"bf16 candidate, MFMAs placed between halves of the vector work", k_cand<0, true, true>},
        {"today: 8x(8 pk + 2 rsq) (4 sources x 4 targets x 64 lanes)", k_today}};
    printf("%-62s %6s %9s %28s\n", "case (per 1024 interactions of a wave)", "w/SIMD", "ms", "SIMD cycles per 1024 int.*");
    for (auto& c : cases) {
        for (int wps : {1, 2, 4}) {
            dim3 grid(cus * wps), block(256);          // 256 threads = 4 waves = 1 per SIMD
            c.k<<<grid, block>>>(out, src, 0.999f, 0.001f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            c.k<<<grid, block>>>(out, src, 0.999f, 0.001f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-62s %6d %9.4f %28.1f\n", c.name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)kTiles * wps));
        }
    }
    printf("* at a nominal 2.4 GHz, per wave, all waves of a SIMD sharing it (k_today handles 1024 interactions per 4 broadcast sources)\n");
    CK(hipFree(out));
    CK(hipFree(src));
    return 0;
}
