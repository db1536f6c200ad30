// force_naive.hip.h — CDNA4 (gfx950) kernels for the naive O(N^2) pair force and the Euler update.
//
// Replaces benchmark.cpp:415-478 (pair loop) and benchmark.cpp:480-510
// (integrator, row A2).  Not a translation: the reference broadcasts one source over 8 AVX lanes;
// here every lane of a 64-wide wavefront owns K target bodies in registers, sources stream through
// LDS as wave-uniform (broadcast) reads, and the pair formula is re-derived for the VALU:
//
//   reference:  a_i += G m_j (r_j-r_i) * rcp(d) / (max(d, max(R_i,R_j))^2 + 0.001f^2)   [sqrt, rcp, div]
//   here:       inv = rsq(d2);  a_i += (r_j-r_i) * m_j * inv * min(inv^2, c_i, c_j)      [one v_rsq_f32]
//               with c = 1/(R^2 + 0.001f^2) precomputed per body at upload.
//
// 1/(max(d2,R2)+e) == min(1/(d2+e), 1/(R2+e)); replacing 1/(d2+e) by inv^2 = 1/d2 is exact to
// < 1 ulp whenever the un-clamped branch is taken with d2 >= R2 >= 16 (e/d2 <= 6.3e-8), which the
// host checks at upload (every reference configuration has R = 8, benchmark.cpp:338).  Inputs that
// violate it run the EXACT variant (one extra v_rcp_f32).  d2 == 0 (self pair or coincident bodies,
// masked by the reference at benchmark.cpp:468-472) contributes exactly 0 because dx = dy = 0 and
// d2 carries a 1e-36 floor that keeps rsq finite — no compare/select in the loop.
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kTileJ = 1024;         // sources per LDS tile
constexpr float kTiny = 1e-36f;      // d2 floor: rsq(1e-36) = 1e18, still finite after * m * c
constexpr float kFarAway = 1.0e18f;  // padding source: d2 ~ 1e36 -> inv^3 underflows to 0
constexpr float kSoft2 = 0.001f * 0.001f;  // benchmark.cpp:416,453

enum ForceVariant : int { kUniform = 1, kPerBody = 2, kExact = 3 };

struct ForceArgs {
    const float* __restrict__ px;   // [n] source/target positions
    const float* __restrict__ py;
    const float* __restrict__ m;    // [n]
    const float* __restrict__ c;    // [n] kPerBody: 1/(R^2+soft2);  kExact: R^2
    float* __restrict__ pax;        // [slices][i_stride] partial accelerations
    float* __restrict__ pay;
    int n;                          // number of sources
    int i_begin, i_end;             // targets [i_begin, i_end)
    int i_stride;                   // row pitch of pax/pay (>= i_end - i_begin)
    int j_per_slice;                // multiple of kTileJ
    float uni_c;                    // kUniform: the common 1/(R^2+soft2)
    float uni_m;                    // kUniform: the common mass (applied once at the end)
};

template <int V>
struct Src;                          // one source as staged in LDS
template <>
struct Src<kUniform> { float x, y; };
template <>
struct Src<kPerBody> { float x, y, m, c; };
template <>
struct Src<kExact> { float x, y, m, c; };

template <int V>
__device__ __forceinline__ void interact(float xi, float yi, float ci, const Src<V>& s, float uni_c,
                                         float& ax, float& ay)
{
    const float dx = s.x - xi;
    const float dy = s.y - yi;
    const float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, kTiny));
    const float inv = __builtin_amdgcn_rsqf(d2);
    float f;
    if constexpr (V == kUniform) {
        f = inv * __builtin_fminf(inv * inv, uni_c);
    } else if constexpr (V == kPerBody) {
        f = (s.m * inv) * __builtin_fminf(__builtin_fminf(inv * inv, ci), s.c);
    } else {
        const float den = __builtin_fmaxf(d2, __builtin_fmaxf(ci, s.c)) + kSoft2;
        f = (s.m * inv) * __builtin_amdgcn_rcpf(den);
    }
    ax = __builtin_fmaf(dx, f, ax);
    ay = __builtin_fmaf(dy, f, ay);
}

// grid = (ceil(n_targets / (kBlock*K)), j_slices).  Block b handles targets
// i_begin + b*kBlock*K + k*kBlock + tid (lane-contiguous => coalesced) against the sources of slice y.
template <int V, int K>
__global__ __launch_bounds__(kBlock) void force_naive_kernel(const ForceArgs a)
{
    using S = Src<V>;
    __shared__ S tile[2][kTileJ];

    const int tid = threadIdx.x;
    const int i0 = a.i_begin + blockIdx.x * (kBlock * K) + tid;
    const int j_begin = blockIdx.y * a.j_per_slice;
    const int j_end = min(a.n, j_begin + a.j_per_slice);

    float xi[K], yi[K], ci[K], ax[K], ay[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = min(i0 + k * kBlock, a.i_end - 1);   // clamp: out-of-range lanes compute, never store
        xi[k] = a.px[i];
        yi[k] = a.py[i];
        ci[k] = (V == kUniform) ? 0.0f : a.c[i];
        ax[k] = 0.0f;
        ay[k] = 0.0f;
    }

    constexpr int kPer = kTileJ / kBlock;   // sources each thread stages per tile
    S stage[kPer];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int j = j0 + q * kBlock + tid;
            const bool ok = j < j_end;
            const int jc = ok ? j : 0;
            stage[q].x = ok ? a.px[jc] : kFarAway;
            stage[q].y = ok ? a.py[jc] : kFarAway;
            if constexpr (V != kUniform) {
                stage[q].m = ok ? a.m[jc] : 0.0f;
                stage[q].c = a.c[jc];
            }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) tile[buf][q * kBlock + tid] = stage[q];
    };

    fetch(j_begin);
    commit(0);
    __syncthreads();

    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += kTileJ) {
        const bool more = j0 + kTileJ < j_end;
        if (more) fetch(j0 + kTileJ);          // global loads stay in flight under the pair loop

        float tx[K], ty[K];                    // per-tile partial sums: bounds fp32 accumulation error
#pragma unroll
        for (int k = 0; k < K; ++k) tx[k] = ty[k] = 0.0f;

        const S* __restrict__ t = tile[buf];
#pragma unroll 8
        for (int jj = 0; jj < kTileJ; ++jj) {
            const S s = t[jj];                 // wave-uniform address: LDS broadcast read
#pragma unroll
            for (int k = 0; k < K; ++k) interact<V>(xi[k], yi[k], ci[k], s, a.uni_c, tx[k], ty[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { ax[k] += tx[k]; ay[k] += ty[k]; }

        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    float* __restrict__ ox = a.pax + (size_t)blockIdx.y * a.i_stride;
    float* __restrict__ oy = a.pay + (size_t)blockIdx.y * a.i_stride;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = i0 + k * kBlock;
        if (i < a.i_end) {
            const float sc = (V == kUniform) ? a.uni_m : 1.0f;
            ox[i - a.i_begin] = ax[k] * sc;
            oy[i - a.i_begin] = ay[k] * sc;
        }
    }
}

// Euler update, benchmark.cpp:483-510: both updates read OLD values.  Sums the j-slice partials in
// slice order (deterministic), stores the total acceleration (the reference's tmp_accX/Y, :476-477)
// and advances the owned targets.  HBM-bound: (8*slices + 16) B read + 24 B written per body.
struct IntegrateArgs {
    float* __restrict__ px;
    float* __restrict__ py;
    float* __restrict__ vx;
    float* __restrict__ vy;
    const float* __restrict__ pax;
    const float* __restrict__ pay;
    float* __restrict__ ax;     // [n] total acceleration, indexed by body
    float* __restrict__ ay;
    int i_begin, i_end, i_stride, slices;
    float dt;
    int advance;                // 0: only reduce the partials (nbody_compute_accel)
};

__global__ __launch_bounds__(256) void integrate_kernel(const IntegrateArgs a)
{
    const int i = a.i_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.i_end) return;
    const int r = i - a.i_begin;
    float sx = 0.0f, sy = 0.0f;
    for (int s = 0; s < a.slices; ++s) {
        sx += a.pax[(size_t)s * a.i_stride + r];
        sy += a.pay[(size_t)s * a.i_stride + r];
    }
    a.ax[i] = sx;
    a.ay[i] = sy;
    if (a.advance) {
        const float x = a.px[i], y = a.py[i], u = a.vx[i], w = a.vy[i];
        a.px[i] = __builtin_fmaf(u, a.dt, x);       // :498-499
        a.py[i] = __builtin_fmaf(w, a.dt, y);
        a.vx[i] = __builtin_fmaf(sx, a.dt, u);      // :502-503
        a.vy[i] = __builtin_fmaf(sy, a.dt, w);
    }
}

}  // namespace nbody
