// force_naive.hip.h — CDNA4 (gfx950) kernels for the naive O(N^2) pair force and the Euler update.
//
// Replaces benchmark.cpp:415-478 (pair loop) and benchmark.cpp:480-510
// (integrator, row A2).  Not a translation: the reference broadcasts one source over 8 AVX lanes;
// here every lane of a 64-wide wavefront owns K target bodies in registers (as K/2 packed pairs),
// sources stream through LDS as wave-uniform (broadcast) reads, and the pair formula is re-derived
// for the gfx950 VALU, whose measured issue costs per wave64 instruction are (tools/ubench.hip):
//     v_pk_{add,mul,fma}_f32  4 cycles for 2 results      v_{sub,mul,fma}_f32  2-2.5 cycles
//     v_min/v_max/v_min3_f32  4 cycles                    v_rsq_f32 / v_rcp_f32  8 cycles
//     any VALU op with an SGPR operand: 4 cycles          `clamp` output modifier: free
//
//   reference:  a_i += G m_j (r_j-r_i) * rcp(d) / (max(d, max(R_i,R_j))^2 + 0.001f^2)   [sqrt, rcp, div]
//   here:       inv = rsq(d2);  a_i += (r_j-r_i) * m_j * inv * min(inv^2, c_i, c_j),  c = 1/(R^2+0.001f^2)
//
// 1/(max(d2,R2)+e) == min(1/(d2+e), 1/(R2+e)); replacing 1/(d2+e) by inv^2 = 1/d2 is exact to < 1 ulp
// whenever the un-clamped branch is taken with d2 >= R2 >= 16 (e/d2 <= 6.3e-8), which the host checks at
// upload (every reference configuration has R = 8, benchmark.cpp:338).  Inputs that violate it run the
// EXACT variant (one extra v_rcp_f32).  d2 == 0 (self pair or coincident bodies, masked by the reference
// at benchmark.cpp:468-472) contributes exactly 0 because dx = dy = 0 and d2 carries a 1e-36 floor that
// keeps rsq finite — no compare/select in the loop.
//
// Fastest variant (kScaled) — what every reference configuration runs: uniform mass, uniform radius R
// that is a power of two with R^2 + 1e-6 == R^2 in fp32 (R = 8: 64 + 1e-6 == 64.0f).  Coordinates are
// pre-multiplied by 1/R (exact), so min(inv^2, 1/R^2) becomes clamp(inv'^2) to [0,1] — the free `clamp`
// output modifier on the v_pk_mul_f32 that squares inv' — and the common factor m/R^2 is applied once at the end:
//     per 2 interactions: 2 v_pk_add (dx,dy) + 2 v_pk_fma (d2) + 2 v_rsq + v_pk_mul clamp + v_pk_mul
//                         + 2 v_pk_fma (accumulate)  = 8 packed + 2 transcendental = 48 issue cycles,
// i.e. 24 cycles per interaction per wave64 against 20 at the "20 flop = 100 %" convention: the
// instruction mix itself caps this kernel at 83 % of the 157.3 TFLOP/s headline, which assumes 2.4 GHz; at the
// 2.23-2.34 GHz the chip holds under this load at its ~1.4 kW power cap that is 77-81 %.  Measured 0.756-0.789
// (177.2-185.0 ms per N = 2^20 launch at the end of round 3, 0.81 of the peak at the measured clock on every box; the
// spread between boxes is the clock, see bench.py's `clock` field).
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;          // 4 wavefronts
constexpr float kTiny = 1e-36f;      // d2 floor: rsq(1e-36) = 1e18, still finite after * m * c
constexpr float kFarAway = 1.0e18f;  // padding source: d2 ~ 1e36 -> inv^3 underflows to 0
constexpr float kSoft2 = 0.001f * 0.001f;  // benchmark.cpp:416,453

enum ForceVariant : int { kScaled = 1, kUniform = 2, kPerBody = 3, kExact = 4 };

// Sources per LDS tile: the big tile amortises the per-tile barrier at large N; the small one lets a
// small N be cut into enough j-slices to fill 256 CUs (N = 10 000 -> 40 slices of 256 sources).
constexpr int kTileSmall = 512;      // small launches: the tile is swept only as far as the slice goes (kDyn), any multiple of 8
template <int V> struct Tile { static constexpr int kTileBig = (V == kScaled || V == kUniform) ? 2048 : 1024; };

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
    int j_per_slice;                // multiple of the variant's big tile size, or (small tile) of 8
    float pre_scale;                // kScaled: 1/R (power of two, exact); otherwise 1
    float uni_c;                    // kUniform: the common 1/(R^2+soft2)
    float out_scale;                // kScaled: m/R^2, kUniform: m, otherwise 1
};

template <int V> struct Src;         // one source as staged in LDS
template <> struct Src<kScaled>  { float x, y; };
template <> struct Src<kUniform> { float x, y; };
template <> struct Src<kPerBody> { float x, y, m, c; };
template <> struct Src<kExact>   { float x, y, m, c; };

// One source against one packed pair of targets.  dx/dy are kept for the accumulate step.
template <int V>
__device__ __forceinline__ f2 pair_weight(const f2 d2, const f2 ci, const Src<V>& s, const float uni_c)
{
    f2 inv;
    inv.x = __builtin_amdgcn_rsqf(d2.x);
    inv.y = __builtin_amdgcn_rsqf(d2.y);
    if constexpr (V == kScaled) {
        // min(inv'^2, 1) as the VOP3P `clamp` bit of the packed multiply that squares inv'.  hipcc does not
        // fold a clamp into v_pk_mul_f32 by itself (it emits two half-rate v_max ... clamp), hence the asm.
        // The leading s_nop is REQUIRED: a VALU read of a v_rsq_f32 result needs one wait state, and hipcc's
        // hazard recogniser cannot see that this asm is such a read (without it, lanes come back stale).
        f2 w;
        asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %1 clamp" : "=v"(w) : "v"(inv));
        return inv * w;
    } else if constexpr (V == kUniform) {
        f2 w = inv * inv;
        w.x = __builtin_fminf(w.x, uni_c);
        w.y = __builtin_fminf(w.y, uni_c);
        return inv * w;
    } else if constexpr (V == kPerBody) {
        f2 w = inv * inv;
        w.x = __builtin_fminf(__builtin_fminf(w.x, ci.x), s.c);
        w.y = __builtin_fminf(__builtin_fminf(w.y, ci.y), s.c);
        return (inv * w) * f2{s.m, s.m};
    } else {
        f2 den;
        den.x = __builtin_fmaxf(__builtin_fmaxf(d2.x, ci.x), s.c);
        den.y = __builtin_fmaxf(__builtin_fmaxf(d2.y, ci.y), s.c);
        den = den + f2{kSoft2, kSoft2};
        f2 r;
        r.x = __builtin_amdgcn_rcpf(den.x);
        r.y = __builtin_amdgcn_rcpf(den.y);
        return (inv * r) * f2{s.m, s.m};
    }
}

// kScaled, a whole inner iteration at once: w[g] = clamp(inv[g]^2) for G packed pairs in ONE asm statement.  The wait
// state a VALU read of a fresh v_rsq_f32 result needs (see pair_weight) can only concern the FIRST multiply — every
// later one has at least one instruction between it and the last v_rsq_f32, which the compiler can only have placed
// before the statement — so one s_nop serves the block instead of one per multiply (round 2: 8 per iteration, priced at
// 0.7 % by profiles/r1_clamp_variants_ab.txt).  Outputs are early-clobber: w[0] is written while inv[1..] are unread.
#ifndef NBODY_FORCE_UNROLL
#define NBODY_FORCE_UNROLL 2     // inner iterations per loop body (4 measured: see profiles/r3_force_unroll_ab.txt)
#endif
#ifndef NBODY_CLAMP_BLOCK
#define NBODY_CLAMP_BLOCK 1
#endif
#ifndef NBODY_ASM_BROADCAST
#define NBODY_ASM_BROADCAST 1    // 0: leave the source broadcast to the compiler (A/B: profiles/r3_asm_broadcast_ab.txt)
#endif
template <int G> __device__ __forceinline__ void clamp_square_block(const f2 (&inv)[G], f2 (&w)[G]);
template <> __device__ __forceinline__ void clamp_square_block<4>(const f2 (&inv)[4], f2 (&w)[4])
{
    asm("s_nop 0\n\t"
        "v_pk_mul_f32 %0, %4, %4 clamp\n\t"
        "v_pk_mul_f32 %1, %5, %5 clamp\n\t"
        "v_pk_mul_f32 %2, %6, %6 clamp\n\t"
        "v_pk_mul_f32 %3, %7, %7 clamp"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(inv[0]), "v"(inv[1]), "v"(inv[2]), "v"(inv[3]));
}
template <> __device__ __forceinline__ void clamp_square_block<8>(const f2 (&inv)[8], f2 (&w)[8])
{
    asm("s_nop 0\n\t"
        "v_pk_mul_f32 %0, %8, %8 clamp\n\t"
        "v_pk_mul_f32 %1, %9, %9 clamp\n\t"
        "v_pk_mul_f32 %2, %10, %10 clamp\n\t"
        "v_pk_mul_f32 %3, %11, %11 clamp\n\t"
        "v_pk_mul_f32 %4, %12, %12 clamp\n\t"
        "v_pk_mul_f32 %5, %13, %13 clamp\n\t"
        "v_pk_mul_f32 %6, %14, %14 clamp\n\t"
        "v_pk_mul_f32 %7, %15, %15 clamp"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
        : "v"(inv[0]), "v"(inv[1]), "v"(inv[2]), "v"(inv[3]), "v"(inv[4]), "v"(inv[5]), "v"(inv[6]), "v"(inv[7]));
}

// grid = (ceil(n_targets / (kBlock*K)), j_slices).  Block b handles targets
// i_begin + b*kBlock*K + k*kBlock + tid (lane-contiguous => coalesced) against the sources of slice y.
// K targets per lane are held as K/2 packed pairs; JU sources are processed per inner iteration in
// stage order (all dx/dy, all d2, all rsq, ...) so no dependent instructions sit back to back.
// kDyn (the small tile): j_per_slice is any multiple of 2 * JU, and a tile is swept only up to the slice's (padded) end.
template <int V, int K, int JU, int TJ, int MINW = 4, bool kDyn = false>
__global__ __launch_bounds__(kBlock, MINW) void force_naive_kernel(const ForceArgs a)
{
    using S = Src<V>;
    constexpr int KP = K / 2, G = JU * KP;
    static_assert(K % 2 == 0 && TJ % JU == 0 && TJ % kBlock == 0, "bad tiling");
    __shared__ S tile[2][TJ];

    const int tid = threadIdx.x;
    const int i0 = a.i_begin + blockIdx.x * (kBlock * K) + tid;
    const int j_begin = blockIdx.y * a.j_per_slice;
    const int j_end = min(a.n, j_begin + a.j_per_slice);

    f2 X[KP], Y[KP], C[KP], AX[KP], AY[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        // clamp: out-of-range lanes compute on a valid body and never store
        const int ia = min(i0 + (2 * k) * kBlock, a.i_end - 1), ib = min(i0 + (2 * k + 1) * kBlock, a.i_end - 1);
        X[k] = f2{a.px[ia] * a.pre_scale, a.px[ib] * a.pre_scale};
        Y[k] = f2{a.py[ia] * a.pre_scale, a.py[ib] * a.pre_scale};
        if constexpr (V == kPerBody || V == kExact) C[k] = f2{a.c[ia], a.c[ib]};
        else C[k] = f2{0.0f, 0.0f};
        AX[k] = f2{0.0f, 0.0f};
        AY[k] = f2{0.0f, 0.0f};
    }
    float tiny = kTiny;
    asm volatile("" : "+v"(tiny));          // keep the floor in a VGPR (an SGPR operand halves the issue rate)
    const f2 tiny2 = {tiny, tiny};

    constexpr int kPer = TJ / kBlock;       // sources each thread stages per tile
    // fetch() only ISSUES the (index-clamped, unconditional) global loads; every use of the loaded values —
    // pre-scale, padding select, LDS store — is in commit(), after the pair loop, so the loads stay in flight
    // under ~100k cycles of VALU work instead of being waited for one by one.
    float rx[kPer], ry[kPer], rm[kPer], rc[kPer];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int jc = min(j0 + q * kBlock + tid, a.n - 1);
            rx[q] = a.px[jc];
            ry[q] = a.py[jc];
            if constexpr (V == kPerBody || V == kExact) { rm[q] = a.m[jc]; rc[q] = a.c[jc]; }
        }
    };
    auto commit = [&](int buf, int j0) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const bool ok = j0 + q * kBlock + tid < j_end;
            S v;
            v.x = ok ? rx[q] * a.pre_scale : kFarAway;
            v.y = ok ? ry[q] * a.pre_scale : kFarAway;
            if constexpr (V == kPerBody || V == kExact) { v.m = ok ? rm[q] : 0.0f; v.c = rc[q]; }
            tile[buf][q * kBlock + tid] = v;
        }
    };

    fetch(j_begin);
    commit(0, j_begin);
    __syncthreads();

    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += TJ) {
        const bool more = j0 + TJ < j_end;
        if (more) fetch(j0 + TJ);              // global loads stay in flight under the pair loop

        f2 tx[KP], ty[KP];                     // per-tile partial sums: bounds fp32 accumulation error
#pragma unroll
        for (int k = 0; k < KP; ++k) { tx[k] = f2{0.0f, 0.0f}; ty[k] = f2{0.0f, 0.0f}; }

        const S* __restrict__ t = tile[buf];
        int cnt = TJ;
        if constexpr (kDyn) {
            cnt = min(TJ, j_begin + a.j_per_slice - j0);       // padding sources (far away, massless) fill the rest of the slice
        }
        auto iteration = [&](const int jj) {
            S s[JU];
            f2 dx[G], dy[G], d2[G], f[G];
#pragma unroll
            for (int u = 0; u < JU; ++u) {
                s[u] = t[jj + u];              // wave-uniform address: LDS broadcast read
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    if constexpr (NBODY_ASM_BROADCAST && sizeof(S) == sizeof(f2)) {
                        // {x,x} - X and {y,y} - Y with the broadcast in the instruction's op_sel bits.  hipcc finds that
                        // form by itself for three of the four floats of a 16-byte LDS read and copies the fourth (the
                        // odd source's y) with a v_mov_b32 first: 4 of 165 VALU instructions per 8 sources (round 3).
                        f2 sv;
                        __builtin_memcpy(&sv, &s[u], sizeof(f2));
                        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx[u * KP + k]) : "v"(sv), "v"(X[k]));
                        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy[u * KP + k]) : "v"(sv), "v"(Y[k]));
                    } else {
                        dx[u * KP + k] = f2{s[u].x, s[u].x} - X[k];
                        dy[u * KP + k] = f2{s[u].y, s[u].y} - Y[k];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) d2[g] = __builtin_elementwise_fma(dy[g], dy[g], tiny2);
#pragma unroll
            for (int g = 0; g < G; ++g) d2[g] = __builtin_elementwise_fma(dx[g], dx[g], d2[g]);
            if constexpr (V == kScaled && NBODY_CLAMP_BLOCK && (G == 4 || G == 8)) {
                f2 inv[G], w[G];
#pragma unroll
                for (int g = 0; g < G; ++g) { inv[g].x = __builtin_amdgcn_rsqf(d2[g].x); inv[g].y = __builtin_amdgcn_rsqf(d2[g].y); }
                clamp_square_block<G>(inv, w);
#pragma unroll
                for (int g = 0; g < G; ++g) f[g] = inv[g] * w[g];
            } else {
#pragma unroll
                for (int u = 0; u < JU; ++u)
#pragma unroll
                    for (int k = 0; k < KP; ++k) f[u * KP + k] = pair_weight<V>(d2[u * KP + k], C[k], s[u], a.uni_c);
            }
#pragma unroll
            for (int u = 0; u < JU; ++u)
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    tx[k] = __builtin_elementwise_fma(dx[u * KP + k], f[u * KP + k], tx[k]);
                    ty[k] = __builtin_elementwise_fma(dy[u * KP + k], f[u * KP + k], ty[k]);
                }
                };
        if constexpr (kDyn) {
            // dynamic trip count: unrolled by hand (the optimizer will not unroll a loop whose count it cannot divide)
            for (int jj = 0; jj < cnt; jj += 2 * JU) { iteration(jj); iteration(jj + JU); }
        } else {
#pragma unroll NBODY_FORCE_UNROLL
            for (int jj = 0; jj < TJ; jj += JU) iteration(jj);
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) { AX[k] += tx[k]; AY[k] += ty[k]; }

        if (more) commit(buf ^ 1, j0 + TJ);
        __syncthreads();
        buf ^= 1;
    }

    float* __restrict__ ox = a.pax + (size_t)blockIdx.y * a.i_stride;
    float* __restrict__ oy = a.pay + (size_t)blockIdx.y * a.i_stride;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int ia = i0 + (2 * k) * kBlock, ib = i0 + (2 * k + 1) * kBlock;
        if (ia < a.i_end) { ox[ia - a.i_begin] = AX[k].x * a.out_scale; oy[ia - a.i_begin] = AY[k].x * a.out_scale; }
        if (ib < a.i_end) { ox[ib - a.i_begin] = AX[k].y * a.out_scale; oy[ib - a.i_begin] = AY[k].y * a.out_scale; }
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
    // fixed summation order (slice 0, 1, 2, ...) => bit-reproducible; loads are issued 8 at a time so the
    // reduction is not a chain of dependent HBM round trips
    float sx = 0.0f, sy = 0.0f;
    int s = 0;
    for (; s + 8 <= a.slices; s += 8) {
        float tx[8], ty[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            tx[u] = a.pax[(size_t)(s + u) * a.i_stride + r];
            ty[u] = a.pay[(size_t)(s + u) * a.i_stride + r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { sx += tx[u]; sy += ty[u]; }
    }
    for (; s < a.slices; ++s) {
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

// The same update with four consecutive bodies per thread and 16-byte loads and stores (the host picks it when
// i_begin and i_stride are multiples of 4, so that every row of partials and every per-body array is 16-byte
// aligned at a quad).  Per body the sum runs over the slices in the same order as above: identical bits.
// The last, ragged quad of the range goes through the scalar statements.
__global__ __launch_bounds__(256) void integrate4_kernel(const IntegrateArgs a)
{
    const int i = a.i_begin + 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= a.i_end) return;
    const int r = i - a.i_begin;
    if (i + 4 > a.i_end) {
        for (int q = i; q < a.i_end; ++q) {
            float sx = 0.0f, sy = 0.0f;
            for (int s = 0; s < a.slices; ++s) {
                sx += a.pax[(size_t)s * a.i_stride + (q - a.i_begin)];
                sy += a.pay[(size_t)s * a.i_stride + (q - a.i_begin)];
            }
            a.ax[q] = sx;
            a.ay[q] = sy;
            if (a.advance) {
                const float x = a.px[q], y = a.py[q], u = a.vx[q], w = a.vy[q];
                a.px[q] = __builtin_fmaf(u, a.dt, x);
                a.py[q] = __builtin_fmaf(w, a.dt, y);
                a.vx[q] = __builtin_fmaf(sx, a.dt, u);
                a.vy[q] = __builtin_fmaf(sy, a.dt, w);
            }
        }
        return;
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 x, y, u, w;
    if (a.advance) {
        x = *reinterpret_cast<const f4*>(a.px + i); y = *reinterpret_cast<const f4*>(a.py + i);
        u = *reinterpret_cast<const f4*>(a.vx + i); w = *reinterpret_cast<const f4*>(a.vy + i);
    }
    f4 sx = {0, 0, 0, 0}, sy = {0, 0, 0, 0};
    int s = 0;
    for (; s + 4 <= a.slices; s += 4) {
        f4 tx[4], ty[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            tx[v] = *reinterpret_cast<const f4*>(a.pax + (size_t)(s + v) * a.i_stride + r);
            ty[v] = *reinterpret_cast<const f4*>(a.pay + (size_t)(s + v) * a.i_stride + r);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) { sx += tx[v]; sy += ty[v]; }
    }
    for (; s < a.slices; ++s) {
        sx += *reinterpret_cast<const f4*>(a.pax + (size_t)s * a.i_stride + r);
        sy += *reinterpret_cast<const f4*>(a.pay + (size_t)s * a.i_stride + r);
    }
    *reinterpret_cast<f4*>(a.ax + i) = sx;
    *reinterpret_cast<f4*>(a.ay + i) = sy;
    if (a.advance) {
        const f4 dt = {a.dt, a.dt, a.dt, a.dt};
        *reinterpret_cast<f4*>(a.px + i) = __builtin_elementwise_fma(u, dt, x);
        *reinterpret_cast<f4*>(a.py + i) = __builtin_elementwise_fma(w, dt, y);
        *reinterpret_cast<f4*>(a.vx + i) = __builtin_elementwise_fma(sx, dt, u);
        *reinterpret_cast<f4*>(a.vy + i) = __builtin_elementwise_fma(sy, dt, w);
    }
}

// main.cpp:535-544: in its Barnes-Hut branch the N%8 tail bodies (never targets) still drift, with zero force
__global__ void tail_drift_kernel(float* __restrict__ px, float* __restrict__ py, const float* __restrict__ vx,
                                  const float* __restrict__ vy, int first, int n, float dt)
{
    const int i = first + (int)threadIdx.x;
    if (i >= n) return;
    px[i] = __builtin_fmaf(vx[i], dt, px[i]);
    py[i] = __builtin_fmaf(vy[i], dt, py[i]);
}

}  // namespace nbody
