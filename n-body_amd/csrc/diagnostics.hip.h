// diagnostics.hip.h — conserved-quantity reduction {Ekin, Epot, Px, Py, Lz} on the GPU.
//
// No counterpart in the reference (it computes no diagnostics); defined by this build
// and mirrored by oracle_diagnostics() in oracle/nbody_oracle.c.  Epot uses the potential that is
// consistent with the reference's clamped force (benchmark.cpp:450-457):
//   phi(d) = -m_i m_j / d                       d >= R = max(R_i, R_j)
//   phi(d) = -m_i m_j (1/R + (R - d)/R^2)       d <  R
// Pair terms are evaluated in fp32 (positions are fp32), summed per 1024-source tile in fp32 and
// accumulated across tiles, lanes and workgroups in fp64.  O(N^2), same LDS tiling as the force kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {

constexpr int kDiagBlock = 256;
constexpr int kDiagTile = 1024;

struct DiagArgs {
    const float* __restrict__ m;
    const float* __restrict__ px;
    const float* __restrict__ py;
    const float* __restrict__ vx;
    const float* __restrict__ vy;
    const float* __restrict__ radius;
    int n;
    double* __restrict__ partials;   // [gridDim.x][5]
};

__global__ __launch_bounds__(kDiagBlock) void diagnostics_kernel(const DiagArgs a)
{
    __shared__ float4 tile[kDiagTile];
    __shared__ double red[5][kDiagBlock / 64];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kDiagBlock + tid;
    const bool live = i < a.n;
    const int ic = live ? i : a.n - 1;
    const float xi = a.px[ic], yi = a.py[ic], ri = a.radius[ic];
    const double mi = live ? (double)a.m[ic] : 0.0;
    double pot = 0.0;
    for (int j0 = 0; j0 < a.n; j0 += kDiagTile) {
        __syncthreads();
        for (int q = tid; q < kDiagTile; q += kDiagBlock) {
            const int j = j0 + q;
            const bool ok = j < a.n;
            const int jc = ok ? j : 0;
            tile[q] = make_float4(a.px[jc], a.py[jc], ok ? a.m[jc] : 0.0f, a.radius[jc]);
        }
        __syncthreads();
        float acc = 0.0f;
#pragma unroll 4
        for (int q = 0; q < kDiagTile; ++q) {
            const float4 s = tile[q];
            const float dx = s.x - xi, dy = s.y - yi;
            const float d2 = __builtin_fmaf(dx, dx, dy * dy);
            const float rr = __builtin_fmaxf(ri, s.w);
            const float inv = __builtin_amdgcn_rsqf(__builtin_fmaxf(d2, 1e-36f));
            const float d = d2 * inv;
            const float ir = __builtin_amdgcn_rcpf(rr);
            const float inner = ir + (rr - d) * ir * ir;
            float phi = d >= rr ? inv : inner;
            phi = (j0 + q == i) ? 0.0f : phi;     // self term only; coincident distinct bodies count (d = 0 < R)
            acc = __builtin_fmaf(s.z, phi, acc);
        }
        pot += (double)acc;
    }
    double v[5];
    const double ui = a.vx[ic], wi = a.vy[ic];
    v[0] = 0.5 * mi * (ui * ui + wi * wi);
    v[1] = -0.5 * mi * pot;
    v[2] = mi * ui;
    v[3] = mi * wi;
    v[4] = mi * ((double)xi * wi - (double)yi * ui);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((tid & 63) == 0) red[k][tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 5) {
        double s = 0.0;
        for (int w = 0; w < kDiagBlock / 64; ++w) s += red[tid][w];
        a.partials[(size_t)blockIdx.x * 5 + tid] = s;
    }
}

}  // namespace nbody
