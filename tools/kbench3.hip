// kbench3.hip — same-process A/B of the PRODUCTION pair kernel's template parameters (developer tool).
// Build: hipcc --offload-arch=gfx950 -O3 -I n-body_amd/csrc tools/kbench3.hip -o tools/kbench3
#include "force_naive.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace nbody;

template <int K, int JU, int TJ, int MINW>
static void run(const char* name, ForceArgs a, int n, int slices, int reps)
{
    a.j_per_slice = ((n + slices - 1) / slices + TJ - 1) / TJ * TJ;
    slices = (n + a.j_per_slice - 1) / a.j_per_slice;
    dim3 grid((n + kBlock * K - 1) / (kBlock * K), slices), block(kBlock);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    force_naive_kernel<kScaled, K, JU, TJ, MINW><<<grid, block>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) force_naive_kernel<kScaled, K, JU, TJ, MINW><<<grid, block>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double gi = (double)n * n / (ms * 1e-3) / 1e9;
    printf("%-28s N=%d grid=%ux%u %9.3f ms %8.1f Ginter/s %5.1f%%\n", name, n, grid.x, grid.y, ms, gi, gi * 20 / 157.3e3 * 100);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 1;
    const int sl = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<float> x(n), y(n);
    unsigned s = 12345u;
    auto u01 = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < n; ++i) { float ang = u01() * 6.2831853f, r = u01() * 400.0f; x[i] = r * cosf(ang); y[i] = r * sinf(ang); }
    float *dx, *dy, *pax, *pay;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4));
    CK(hipMalloc(&pax, (size_t)n * 4 * 64)); CK(hipMalloc(&pay, (size_t)n * 4 * 64));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice));
    ForceArgs a{};
    a.px = dx; a.py = dy; a.pax = pax; a.pay = pay; a.n = n; a.i_begin = 0; a.i_end = n; a.i_stride = n;
    a.pre_scale = 0.125f; a.uni_c = 1.0f / 64.0f; a.out_scale = 1000.0f / 64.0f;
    for (int round = 0; round < 2; ++round) {
        run<4, 4, 2048, 4>("K4 JU4 T2048 w4 (shipped)", a, n, sl, reps);
        run<2, 4, 2048, 4>("K2 JU4 T2048 w4", a, n, sl, reps);
        run<2, 8, 2048, 4>("K2 JU8 T2048 w4", a, n, sl, reps);
        run<4, 2, 2048, 4>("K4 JU2 T2048 w4", a, n, sl, reps);
        run<4, 8, 2048, 2>("K4 JU8 T2048 w2", a, n, sl, reps);
        run<4, 4, 2048, 2>("K4 JU4 T2048 w2", a, n, sl, reps);
        run<4, 4, 2048, 5>("K4 JU4 T2048 w5", a, n, sl, reps);
        run<8, 2, 2048, 2>("K8 JU2 T2048 w2", a, n, sl, reps);
        run<8, 4, 2048, 2>("K8 JU4 T2048 w2", a, n, sl, reps);
        run<6, 4, 2048, 3>("K6 JU4 T2048 w3", a, n, sl, reps);
        run<4, 4, 4096, 4>("K4 JU4 T4096 w4", a, n, sl, reps);
        run<4, 4, 1024, 4>("K4 JU4 T1024 w4", a, n, sl, reps);
    }
    return 0;
}
