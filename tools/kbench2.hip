// kbench2.hip — experiments on the pair-loop formulation (developer tool).
// Pre-scaled coordinates (x/S, S = clamp radius, power of two) + `clamp` output modifier instead of v_min.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;
constexpr float kTiny = 1e-36f;

struct Args {
    const float* __restrict__ px; const float* __restrict__ py;
    float* __restrict__ ax; float* __restrict__ ay;
    int n; int j_per_slice; int i_stride; float inv_s; float out_scale;
};

// MODE 0: scalar ops, asm clamp.  MODE 1: float2-packed (two targets per lane pair), asm pk clamp.
// MODE 2: scalar ops, fmed3 clamp (compiler folds?).
template <int MODE, int K, int TJ, int UNROLL>
__global__ __launch_bounds__(kBlock) void force_v2(const Args a)
{
    __shared__ float2 tile[2][TJ];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * (kBlock * K) + tid;
    const int j_begin = blockIdx.y * a.j_per_slice;
    const int j_end = min(a.n, j_begin + a.j_per_slice);
    float xi[K], yi[K], ax[K], ay[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = min(i0 + k * kBlock, a.n - 1);
        xi[k] = a.px[i] * a.inv_s; yi[k] = a.py[i] * a.inv_s; ax[k] = 0; ay[k] = 0;
    }
    float tiny = kTiny;
    asm volatile("" : "+v"(tiny));
    constexpr int kPer = TJ / kBlock;
    float2 stage[kPer];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int j = j0 + q * kBlock + tid;
            const bool ok = j < j_end;
            const int jc = ok ? j : 0;
            stage[q].x = ok ? a.px[jc] * a.inv_s : 1e18f;
            stage[q].y = ok ? a.py[jc] * a.inv_s : 1e18f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) tile[buf][q * kBlock + tid] = stage[q];
    };
    fetch(j_begin); commit(0); __syncthreads();
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += TJ) {
        const bool more = j0 + TJ < j_end;
        if (more) fetch(j0 + TJ);
        const float2* __restrict__ t = tile[buf];
        if constexpr (MODE == 1) {
            static_assert(K % 2 == 0, "packed mode needs even K");
            f2 tx[K / 2], ty[K / 2], X[K / 2], Y[K / 2];
#pragma unroll
            for (int k = 0; k < K / 2; ++k) { tx[k] = 0; ty[k] = 0; X[k] = f2{xi[2 * k], xi[2 * k + 1]}; Y[k] = f2{yi[2 * k], yi[2 * k + 1]}; }
            const f2 tiny2 = {tiny, tiny};
#pragma unroll UNROLL
            for (int jj = 0; jj < TJ; ++jj) {
                const float2 s = t[jj];
#pragma unroll
                for (int k = 0; k < K / 2; ++k) {
                    const f2 dx = f2{s.x, s.x} - X[k];
                    const f2 dy = f2{s.y, s.y} - Y[k];
                    const f2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, tiny2));
                    f2 inv; inv.x = __builtin_amdgcn_rsqf(d2.x); inv.y = __builtin_amdgcn_rsqf(d2.y);
                    f2 w;
                    asm("v_pk_mul_f32 %0, %1, %1 clamp" : "=v"(w) : "v"(inv));
                    const f2 f = inv * w;
                    tx[k] = __builtin_elementwise_fma(dx, f, tx[k]);
                    ty[k] = __builtin_elementwise_fma(dy, f, ty[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < K / 2; ++k) { ax[2 * k] += tx[k].x; ax[2 * k + 1] += tx[k].y; ay[2 * k] += ty[k].x; ay[2 * k + 1] += ty[k].y; }
        } else {
            float tx[K], ty[K];
#pragma unroll
            for (int k = 0; k < K; ++k) tx[k] = ty[k] = 0;
#pragma unroll UNROLL
            for (int jj = 0; jj < TJ; ++jj) {
                const float2 s = t[jj];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float dx = s.x - xi[k], dy = s.y - yi[k];
                    const float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, tiny));
                    const float inv = __builtin_amdgcn_rsqf(d2);
                    float w;
                    if constexpr (MODE == 0) asm("v_mul_f32_e64 %0, %1, %1 clamp" : "=v"(w) : "v"(inv));
                    else w = __builtin_amdgcn_fmed3f(inv * inv, 0.0f, 1.0f);
                    const float f = inv * w;
                    tx[k] = __builtin_fmaf(dx, f, tx[k]);
                    ty[k] = __builtin_fmaf(dy, f, ty[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) { ax[k] += tx[k]; ay[k] += ty[k]; }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* ox = a.ax + (size_t)blockIdx.y * a.i_stride;
    float* oy = a.ay + (size_t)blockIdx.y * a.i_stride;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k * kBlock;
        if (i < a.n) { ox[i] = ax[k] * a.out_scale; oy[i] = ay[k] * a.out_scale; }
    }
}


// v3: packed, explicit stage order over G = JU*K/2 independent chains, optional sched_barrier between stages
template <int K, int TJ, int JU, int FENCE>
__global__ __launch_bounds__(kBlock) void force_v3(const Args a)
{
    __shared__ float2 tile[2][TJ];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * (kBlock * K) + tid;
    const int j_begin = blockIdx.y * a.j_per_slice;
    const int j_end = min(a.n, j_begin + a.j_per_slice);
    constexpr int KP = K / 2, G = JU * KP;
    f2 X[KP], Y[KP], AX[KP], AY[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int ia = min(i0 + (2 * k) * kBlock, a.n - 1), ib = min(i0 + (2 * k + 1) * kBlock, a.n - 1);
        X[k] = f2{a.px[ia] * a.inv_s, a.px[ib] * a.inv_s};
        Y[k] = f2{a.py[ia] * a.inv_s, a.py[ib] * a.inv_s};
        AX[k] = 0; AY[k] = 0;
    }
    float tiny = kTiny;
    asm volatile("" : "+v"(tiny));
    const f2 tiny2 = {tiny, tiny};
    constexpr int kPer = TJ / kBlock;
    float2 stage[kPer];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int j = j0 + q * kBlock + tid;
            const bool ok = j < j_end;
            const int jc = ok ? j : 0;
            stage[q].x = ok ? a.px[jc] * a.inv_s : 1e18f;
            stage[q].y = ok ? a.py[jc] * a.inv_s : 1e18f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) tile[buf][q * kBlock + tid] = stage[q];
    };
    fetch(j_begin); commit(0); __syncthreads();
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += TJ) {
        const bool more = j0 + TJ < j_end;
        if (more) fetch(j0 + TJ);
        const float2* __restrict__ t = tile[buf];
        f2 tx[KP], ty[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) { tx[k] = 0; ty[k] = 0; }
#pragma unroll 2
        for (int jj = 0; jj < TJ; jj += JU) {
            f2 dx[G], dy[G], d2[G], inv[G], w[G];
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                const float2 s = t[jj + u];
#pragma unroll
                for (int k = 0; k < KP; ++k) { dx[u * KP + k] = f2{s.x, s.x} - X[k]; dy[u * KP + k] = f2{s.y, s.y} - Y[k]; }
            }
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) d2[g] = __builtin_elementwise_fma(dy[g], dy[g], tiny2);
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) d2[g] = __builtin_elementwise_fma(dx[g], dx[g], d2[g]);
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
            {
#pragma unroll
            for (int g = 0; g < G; ++g) { inv[g].x = __builtin_amdgcn_rsqf(d2[g].x); inv[g].y = __builtin_amdgcn_rsqf(d2[g].y); }
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if constexpr (FENCE == 2) { w[g].x = __builtin_amdgcn_fmed3f(inv[g].x * inv[g].x, 0.0f, 1.0f); w[g].y = __builtin_amdgcn_fmed3f(inv[g].y * inv[g].y, 0.0f, 1.0f); }
                else if constexpr (FENCE == 3) asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %1 clamp" : "=v"(w[g]) : "v"(inv[g]));
                else asm("v_pk_mul_f32 %0, %1, %1 clamp" : "=v"(w[g]) : "v"(inv[g]));
            }
            }
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) w[g] = inv[g] * w[g];
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < JU; ++u)
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    tx[k] = __builtin_elementwise_fma(dx[u * KP + k], w[u * KP + k], tx[k]);
                    ty[k] = __builtin_elementwise_fma(dy[u * KP + k], w[u * KP + k], ty[k]);
                }
            if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) { AX[k] += tx[k]; AY[k] += ty[k]; }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* ox = a.ax + (size_t)blockIdx.y * a.i_stride;
    float* oy = a.ay + (size_t)blockIdx.y * a.i_stride;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int ia = i0 + (2 * k) * kBlock, ib = i0 + (2 * k + 1) * kBlock;
        if (ia < a.n) { ox[ia] = AX[k].x * a.out_scale; oy[ia] = AY[k].x * a.out_scale; }
        if (ib < a.n) { ox[ib] = AX[k].y * a.out_scale; oy[ib] = AY[k].y * a.out_scale; }
    }
}

static std::vector<double> rx, ry;
static std::vector<int> chk;

template <int MODE, int K, int TJ, int UNROLL>
static void run(const char* name, Args a, int n, int slices, int reps)
{
    a.j_per_slice = ((n + slices - 1) / slices + TJ - 1) / TJ * TJ;
    slices = (n + a.j_per_slice - 1) / a.j_per_slice;
    dim3 grid((n + kBlock * K - 1) / (kBlock * K), slices), block(kBlock);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    force_v2<MODE, K, TJ, UNROLL><<<grid, block>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) force_v2<MODE, K, TJ, UNROLL><<<grid, block>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double gi = (double)n * n / (ms * 1e-3) / 1e9;
    std::vector<float> hx((size_t)slices * a.i_stride), hy((size_t)slices * a.i_stride);
    CK(hipMemcpy(hx.data(), a.ax, hx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy.data(), a.ay, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t t = 0; t < chk.size(); ++t) {
        float sx = 0, sy = 0;
        for (int s = 0; s < slices; ++s) { sx += hx[(size_t)s * a.i_stride + chk[t]]; sy += hy[(size_t)s * a.i_stride + chk[t]]; }
        double e = hypot(sx - rx[t], sy - ry[t]) / hypot(rx[t], ry[t]);
        if (e > worst) worst = e;
    }
    printf("%-34s N=%d grid=%ux%u %9.3f ms %8.1f Ginter/s %5.1f%%  err %.2e\n", name, n, grid.x, grid.y, ms, gi, gi * 20 / 157.3e3 * 100, worst);
}

template <int K, int TJ, int JU, int FENCE>
static void run3(const char* name, Args a, int n, int slices, int reps)
{
    a.j_per_slice = ((n + slices - 1) / slices + TJ - 1) / TJ * TJ;
    slices = (n + a.j_per_slice - 1) / a.j_per_slice;
    dim3 grid((n + kBlock * K - 1) / (kBlock * K), slices), block(kBlock);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    force_v3<K, TJ, JU, FENCE><<<grid, block>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) force_v3<K, TJ, JU, FENCE><<<grid, block>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double gi = (double)n * n / (ms * 1e-3) / 1e9;
    std::vector<float> hx((size_t)slices * a.i_stride), hy((size_t)slices * a.i_stride);
    CK(hipMemcpy(hx.data(), a.ax, hx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy.data(), a.ay, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t t = 0; t < chk.size(); ++t) {
        float sx = 0, sy = 0;
        for (int s = 0; s < slices; ++s) { sx += hx[(size_t)s * a.i_stride + chk[t]]; sy += hy[(size_t)s * a.i_stride + chk[t]]; }
        double e = hypot(sx - rx[t], sy - ry[t]) / hypot(rx[t], ry[t]);
        if (e > worst) worst = e;
    }
    printf("%-34s N=%d grid=%ux%u %9.3f ms %8.1f Ginter/s %5.1f%%  err %.2e\n", name, n, grid.x, grid.y, ms, gi, gi * 20 / 157.3e3 * 100, worst);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    std::vector<float> x(n), y(n);
    unsigned s = 12345u;
    auto u01 = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < n; ++i) { float ang = u01() * 6.2831853f, r = u01() * 400.0f; x[i] = r * cosf(ang); y[i] = r * sinf(ang); }
    float *dx, *dy, *pax, *pay;
    const int max_slices = 64;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4));
    CK(hipMalloc(&pax, (size_t)n * 4 * max_slices)); CK(hipMalloc(&pay, (size_t)n * 4 * max_slices));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice));
    Args a{dx, dy, pax, pay, n, 0, n, 1.0f / 8.0f, 1000.0f / 64.0f};
    for (int t = 0; t < 64; ++t) {
        int i = (int)((long long)t * (n - 1) / 63) ^ (t & 1 ? 256 : 0); if (i >= n) i = n - 1;
        chk.push_back(i);
        double sx = 0, sy = 0;
        for (int j = 0; j < n; ++j) {
            double ddx = (double)x[j] - x[i], ddy = (double)y[j] - y[i], d2 = ddx * ddx + ddy * ddy;
            if (d2 <= 0) continue;
            double d = sqrt(d2), sd = d > 8 ? d : 8;
            double f = 1000.0 / (sd * sd + 1e-6);
            sx += ddx / d * f; sy += ddy / d * f;
        }
        rx.push_back(sx); ry.push_back(sy);
    }
    const int sl = n >= (1 << 19) ? 2 : (n >= 65536 ? 8 : 16);
    for (int round = 0; round < 3; ++round) {
        run3<4, 2048, 4, 0>("v3 K=4 JU=4 asm-unsafe", a, n, sl, reps);
        run3<4, 2048, 4, 2>("v3 K=4 JU=4 native scalar clamp", a, n, sl, reps);
        run3<4, 2048, 4, 3>("v3 K=4 JU=4 asm nop-prefixed", a, n, sl, reps);
    }
    return 0;
    run<0, 1, 1024, 8>("scalar-asm K=1 u8", a, n, sl, reps);
    run<0, 2, 1024, 8>("scalar-asm K=2 u8", a, n, sl, reps);
    run<0, 4, 1024, 8>("scalar-asm K=4 u8", a, n, sl, reps);
    run<0, 4, 1024, 4>("scalar-asm K=4 u4", a, n, sl, reps);
    run<0, 8, 1024, 4>("scalar-asm K=8 u4", a, n, sl, reps);
    run<2, 2, 1024, 8>("scalar-med3 K=2 u8", a, n, sl, reps);
    run<2, 4, 1024, 8>("scalar-med3 K=4 u8", a, n, sl, reps);
    run<1, 2, 1024, 8>("packed K=2 u8", a, n, sl, reps);
    run<1, 4, 1024, 8>("packed K=4 u8", a, n, sl, reps);
    run<1, 4, 1024, 4>("packed K=4 u4", a, n, sl, reps);
    run<1, 8, 1024, 4>("packed K=8 u4", a, n, sl, reps);
    run<1, 4, 2048, 8>("packed K=4 u8 TJ2048", a, n, sl, reps);
    run<0, 4, 2048, 8>("scalar-asm K=4 u8 TJ2048", a, n, sl, reps);
    return 0;
}
