// kbench.hip — stand-alone timing of the naive force kernel variants (developer tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -I n-body_amd/csrc tools/kbench.hip -o tools/kbench
// Run:   tools/kbench [N] [reps]
#include "force_naive.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace nbody;

template <int V, int K>
static double run(const char* name, ForceArgs a, int n, int slices, int reps, std::vector<float>& hax, std::vector<float>& hay)
{
    a.j_per_slice = ((n + slices - 1) / slices + kTileJ - 1) / kTileJ * kTileJ;
    slices = (n + a.j_per_slice - 1) / a.j_per_slice;
    dim3 grid((n + kBlock * K - 1) / (kBlock * K), slices), block(kBlock);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    force_naive_kernel<V, K><<<grid, block>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) force_naive_kernel<V, K><<<grid, block>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double gi = (double)n * n / (ms * 1e-3) / 1e9;
    printf("%-22s N=%d slices=%d grid=%ux%u  %9.3f ms  %9.1f Ginter/s  %6.1f%% of 157.3TF@20flop\n", name, n, slices,
           grid.x, grid.y, ms, gi, gi * 20 / 157.3e3 * 100);
    // slice-sum on host for the check
    std::vector<float> px((size_t)slices * a.i_stride), py((size_t)slices * a.i_stride);
    CK(hipMemcpy(px.data(), a.pax, px.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(py.data(), a.pay, py.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        float sx = 0, sy = 0;
        for (int s = 0; s < slices; ++s) { sx += px[(size_t)s * a.i_stride + i]; sy += py[(size_t)s * a.i_stride + i]; }
        hax[i] = sx; hay[i] = sy;
    }
    return ms;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 65536;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const int max_slices = 64;
    std::vector<float> x(n), y(n), m(n, 1000.0f), c(n, 1.0f / (64.0f + kSoft2));
    unsigned s = 12345u;
    auto u01 = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < n; ++i) { float ang = u01() * 6.2831853f, r = u01() * 400.0f; x[i] = r * cosf(ang); y[i] = r * sinf(ang); }
    float *dx, *dy, *dm, *dc, *pax, *pay;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&dm, n * 4)); CK(hipMalloc(&dc, n * 4));
    CK(hipMalloc(&pax, (size_t)n * 4 * max_slices)); CK(hipMalloc(&pay, (size_t)n * 4 * max_slices));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, m.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
    ForceArgs a{};
    a.px = dx; a.py = dy; a.m = dm; a.c = dc; a.pax = pax; a.pay = pay; a.n = n; a.i_begin = 0; a.i_end = n; a.i_stride = n;
    a.uni_c = c[0]; a.uni_m = 1000.0f;

    // f64 check on a handful of targets
    const int nchk = 8;
    std::vector<double> rx(nchk), ry(nchk);
    for (int t = 0; t < nchk; ++t) {
        int i = (int)((long long)t * (n - 1) / (nchk - 1));
        double sx = 0, sy = 0;
        for (int j = 0; j < n; ++j) {
            double ddx = (double)x[j] - x[i], ddy = (double)y[j] - y[i], d2 = ddx * ddx + ddy * ddy;
            if (d2 <= 0) continue;
            double d = sqrt(d2), sd = d > 8 ? d : 8;
            double f = 1000.0 / (sd * sd + (double)kSoft2);
            sx += ddx / d * f; sy += ddy / d * f;
        }
        rx[t] = sx; ry[t] = sy;
    }
    std::vector<float> hax(n), hay(n);
    auto check = [&]() {
        double worst = 0;
        for (int t = 0; t < nchk; ++t) {
            int i = (int)((long long)t * (n - 1) / (nchk - 1));
            double e = hypot(hax[i] - rx[t], hay[i] - ry[t]) / hypot(rx[t], ry[t]);
            if (e > worst) worst = e;
        }
        printf("    max rel err vs f64 on %d targets: %.3e\n", nchk, worst);
    };
    const int sl1 = 1;
    const int slA = n >= (1 << 19) ? 1 : (n >= 65536 ? 4 : 16);
    run<kUniform, 1>("uniform K=1", a, n, slA, reps, hax, hay); check();
    run<kUniform, 2>("uniform K=2", a, n, slA, reps, hax, hay); check();
    run<kUniform, 4>("uniform K=4", a, n, slA, reps, hax, hay); check();
    run<kUniform, 8>("uniform K=8", a, n, slA, reps, hax, hay); check();
    run<kUniform, 4>("uniform K=4 s=1", a, n, sl1, reps, hax, hay);
    run<kUniform, 4>("uniform K=4 s=2x", a, n, slA * 2, reps, hax, hay);
    run<kUniform, 4>("uniform K=4 s=4x", a, n, slA * 4, reps, hax, hay);
    run<kPerBody, 2>("perbody K=2", a, n, slA, reps, hax, hay); check();
    run<kPerBody, 4>("perbody K=4", a, n, slA, reps, hax, hay); check();
    for (int i = 0; i < n; ++i) c[i] = 64.0f;
    CK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice));
    run<kExact, 2>("exact K=2", a, n, slA, reps, hax, hay); check();
    run<kExact, 4>("exact K=4", a, n, slA, reps, hax, hay); check();
    return 0;
}
