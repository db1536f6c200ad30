// benchmark_main.cpp — counterpart of the reference's console benchmark (benchmark.cpp:297-566) on the
// MI355X C-ABI.  Same parameter block (benchmark.cpp:300-306, overridable from the command line instead
// of by editing literals), same generator (benchmark.cpp:317-339, fixed seed), same per->=1 s log line
// `FPS: %.2f | Physics: %.2f ms | Total frame: %.2f ms` to stdout and ./log.txt (benchmark.cpp:534-556),
// but a FINITE loop (the reference's while(true) never reaches its footer, benchmark.cpp:349,559-564)
// and a closing summary with G interactions/s and the FP32-roofline fraction.
//
//   nbody_benchmark [--n N] [--naive|--bh] [--steps K] [--warmup W] [--dt x] [--theta x] [--spread x]
//                   [--mass x] [--seed s] [--no-ref-compat] [--no-log]
#include "nbody.h"
#include "config_reader.hpp"
#include "initial_conditions.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

static std::string now_string()   // benchmark.cpp:281-294
{
    std::time_t t = std::time(nullptr);
    std::tm tmv;
    localtime_r(&t, &tmv);
    std::ostringstream os;
    os << std::put_time(&tmv, "%Y-%m-%d %H:%M:%S");
    return os.str();
}

#define NB_CHECK(call)                                                        \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != NBODY_OK) {                                                \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nbody_last_error()); \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main(int argc, char** argv)
{
    nbody_config cfg;
    nbody_default_config(&cfg);        // benchmark.cpp:300-306
    int steps = 100, warmup = 3;
    uint32_t seed = 12345u;
    bool write_log = true;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--n")) cfg.num_bodies = std::atoi(next("--n"));
        else if (!std::strcmp(argv[i], "--naive")) cfg.use_barnes_hut = 0;
        else if (!std::strcmp(argv[i], "--bh")) cfg.use_barnes_hut = 1;
        else if (!std::strcmp(argv[i], "--steps")) steps = std::atoi(next("--steps"));
        else if (!std::strcmp(argv[i], "--warmup")) warmup = std::atoi(next("--warmup"));
        else if (!std::strcmp(argv[i], "--dt")) cfg.dt = (float)std::atof(next("--dt"));
        else if (!std::strcmp(argv[i], "--theta")) cfg.theta = (float)std::atof(next("--theta"));
        else if (!std::strcmp(argv[i], "--spread")) cfg.spread = (float)std::atof(next("--spread"));
        else if (!std::strcmp(argv[i], "--mass")) cfg.mass = (float)std::atof(next("--mass"));
        else if (!std::strcmp(argv[i], "--seed")) seed = (uint32_t)std::strtoul(next("--seed"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--no-ref-compat")) cfg.ref_compat = 0;
        else if (!std::strcmp(argv[i], "--no-log")) write_log = false;
        else { std::fprintf(stderr, "unknown flag %s\n", argv[i]); return 2; }
    }
    const int n = cfg.num_bodies;

    nbody_host::BodyStore data;
    nbody_host::fill_disc(data, n, cfg.spread, cfg.mass, seed, 0.3f);   // benchmark.cpp:317-339

    nbody_ctx* ctx = nullptr;
    NB_CHECK(nbody_create(&cfg, &ctx));
    NB_CHECK(nbody_upload(ctx, data.m.data(), data.posX.data(), data.posY.data(), data.velX.data(),
                          data.velY.data(), data.radius.data()));
    for (int w = 0; w < warmup; ++w) NB_CHECK(nbody_step(ctx, 1));
    NB_CHECK(nbody_sync(ctx));

    std::ofstream logFile;
    if (write_log) logFile.open("log.txt", std::ios::app);   // benchmark.cpp:314

    using Clock = std::chrono::high_resolution_clock;
    int frameCount = 0;
    double elapsed = 0.0, physics_total_ms = 0.0;
    for (int s = 0; s < steps; ++s) {
        const auto frameStart = Clock::now();
        const auto physicsStart = Clock::now();
        NB_CHECK(nbody_step(ctx, 1));      // replaces benchmark.cpp:354-511
        NB_CHECK(nbody_sync(ctx));
        const auto physicsEnd = Clock::now();
        const auto frameEnd = Clock::now();
        const double frameMs = std::chrono::duration<double, std::milli>(frameEnd - frameStart).count();
        const double physicsMs = std::chrono::duration<double, std::milli>(physicsEnd - physicsStart).count();
        physics_total_ms += physicsMs;
        ++frameCount;
        elapsed += frameMs * 1e-3;
        if (elapsed >= 1.0) {              // benchmark.cpp:531-556: FPS is the window average, ms the last frame
            char buf[256];
            std::snprintf(buf, sizeof(buf), "FPS: %.2f | Physics: %.2f ms | Total frame: %.2f ms",
                          frameCount / elapsed, physicsMs, frameMs);
            std::cout << buf << std::endl;
            if (logFile.is_open()) { logFile << buf << std::endl; logFile.flush(); }
            frameCount = 0;
            elapsed = 0.0;
        }
    }
    const double ms_per_step = physics_total_ms / (steps > 0 ? steps : 1);
    char buf[320];
    if (!cfg.use_barnes_hut) {
        const double ginter = (double)n * (double)n / (ms_per_step * 1e-3) / 1e9;
        std::snprintf(buf, sizeof(buf),
                      "SUMMARY | N=%d | Mode: N^2 | steps=%d | %.4f ms/step | %.1f G interactions/s | %.1f%% of 157.3 TFLOP/s (20 flop/interaction)",
                      n, steps, ms_per_step, ginter, ginter * 20.0 / 157.3e3 * 100.0);
    } else {
        std::snprintf(buf, sizeof(buf), "SUMMARY | N=%d | Mode: BarnesHut theta=%.2f | steps=%d | %.4f ms/step", n,
                      cfg.theta, steps, ms_per_step);
    }
    std::cout << buf << std::endl;
    if (logFile.is_open()) {
        logFile << buf << std::endl;
        logFile << "=== Benchmark ended: " << now_string() << " ===" << std::endl << std::endl;   // benchmark.cpp:561
    }
    nbody_destroy(ctx);
    return 0;
}
