// sim_main.cpp — headless counterpart of the reference's windowed program (main.cpp:334-711).
// Reads config.json with the reference's lookup rules (main.cpp:348-402 -> config_reader.hpp), builds the
// same disc (main.cpp:418-441, velocity factor 0.05), and per frame runs the physics block on the GPU
// (replaces main.cpp:485-642) and downloads posX/posY — the only fields the renderer consumes
// (main.cpp:650-653).  SFML is not available in this image, so instead of drawing
// circles each frame's positions can be rasterised into a PGM image (--dump-pgm DIR).  Log line format
// and log.txt append are those of main.cpp:676-700.
//
// --async-frames pipelines the two: frame f's positions travel to pinned host memory (nbody_frame_publish /
// nbody_frame_acquire) and are "rendered" while step f+1 already runs on the GPU; "Physics" then reports the
// time the host waited for the GPU in that frame.
//
//   nbody_sim [--frames K] [--seed s] [--config path] [--dump-pgm DIR] [--async-frames] [--no-log] [--print-config]
#include "nbody.h"
#include "config_reader.hpp"
#include "initial_conditions.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

static void dump_pgm(const std::string& path, const float* x, const float* y, size_t count, int size, float half_extent)
{
    std::vector<unsigned char> img((size_t)size * size, 0);
    for (size_t i = 0; i < count; ++i) {
        const int u = (int)((x[i] / half_extent * 0.5f + 0.5f) * size);
        const int v = (int)((y[i] / half_extent * 0.5f + 0.5f) * size);
        if (u >= 0 && u < size && v >= 0 && v < size) {
            unsigned char& p = img[(size_t)v * size + u];
            p = (unsigned char)std::min(255, p + 64);
        }
    }
    std::ofstream f(path, std::ios::binary);
    f << "P5\n" << size << " " << size << "\n255\n";
    f.write((const char*)img.data(), (std::streamsize)img.size());
}

int main(int argc, char** argv)
{
    int frames = 100;
    uint32_t seed = 12345u;
    std::string config_name = "config.json", dump_dir;
    bool write_log = true, async_frames = false, print_config = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next("--frames"));
        else if (!std::strcmp(argv[i], "--seed")) seed = (uint32_t)std::strtoul(next("--seed"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--config")) config_name = next("--config");
        else if (!std::strcmp(argv[i], "--dump-pgm")) dump_dir = next("--dump-pgm");
        else if (!std::strcmp(argv[i], "--no-log")) write_log = false;
        else if (!std::strcmp(argv[i], "--async-frames")) async_frames = true;
        else if (!std::strcmp(argv[i], "--print-config")) print_config = true;
        else { std::fprintf(stderr, "unknown flag %s\n", argv[i]); return 2; }
    }

    const nbody_host::SimConfig sc = nbody_host::load_config(config_name);
    if (print_config) {          // what the reader made of the file (no GPU needed); one field per line
        std::printf("useBarnesHut=%d\nuseParallel=%d\nnumBodies=%d\nspread=%.9g\nmass=%.9g\ntheta=%.9g\ndt=%.9g\nwindowSize=%d\nwindowTitle=%s\n",
                    (int)sc.useBarnesHut, (int)sc.useParallel, sc.numBodies, sc.spread, sc.mass, sc.theta, sc.dt, sc.windowSize,
                    sc.windowTitle.c_str());
        return 0;
    }
    nbody_config cfg;
    nbody_default_config(&cfg);
    cfg.num_bodies = sc.numBodies;
    cfg.spread = sc.spread;
    cfg.mass = sc.mass;
    cfg.theta = sc.theta;
    cfg.dt = sc.dt;
    cfg.use_barnes_hut = sc.useBarnesHut;
    cfg.use_parallel = sc.useParallel;
    cfg.ref_compat = 2;                      // main.cpp's handling of the N%8 tail (main.cpp:535-544)
    const int n = cfg.num_bodies;

    nbody_host::BodyStore data;
    nbody_host::fill_disc(data, n, cfg.spread, cfg.mass, seed, 0.05f);   // main.cpp:431-432

    nbody_ctx* ctx = nullptr;
    if (nbody_create(&cfg, &ctx) != NBODY_OK) { std::fprintf(stderr, "nbody_create: %s\n", nbody_last_error()); return 1; }
    if (nbody_upload(ctx, data.m.data(), data.posX.data(), data.posY.data(), data.velX.data(), data.velY.data(),
                     data.radius.data()) != NBODY_OK) {
        std::fprintf(stderr, "nbody_upload: %s\n", nbody_last_error());
        return 1;
    }

    std::ofstream logFile;
    if (write_log) logFile.open("log.txt", std::ios::app);
    using Clock = std::chrono::high_resolution_clock;
    int frameCount = 0;
    double elapsed = 0.0;
    for (int f = 0; f < frames; ++f) {
        const auto frameStart = Clock::now();
        auto physicsStart = Clock::now();
        auto physicsEnd = physicsStart;
        const float *fx = nullptr, *fy = nullptr;
        int shown = f;                               // which step's positions this frame draws
        if (!async_frames) {
            if (nbody_step(ctx, 1) != NBODY_OK || nbody_sync(ctx) != NBODY_OK) {
                std::fprintf(stderr, "nbody_step: %s\n", nbody_last_error());
                return 1;
            }
            physicsEnd = Clock::now();
            // "rendering pass": the renderer needs positions only (main.cpp:650-653)
            if (nbody_download(ctx, data.posX.data(), data.posY.data(), nullptr, nullptr) != NBODY_OK) {
                std::fprintf(stderr, "nbody_download: %s\n", nbody_last_error());
                return 1;
            }
            fx = data.posX.data();
            fy = data.posY.data();
        } else {
            // enqueue step f and its snapshot, then draw frame f-1 (frame 0 is drawn together with frame 1's step)
            if (nbody_step(ctx, 1) != NBODY_OK || nbody_frame_publish(ctx) != NBODY_OK) {
                std::fprintf(stderr, "nbody_step/frame_publish: %s\n", nbody_last_error());
                return 1;
            }
            if (f == 0 && frames > 1) continue;
            int64_t step_index = 0;
            if (nbody_frame_acquire(ctx, &fx, &fy, &step_index) != NBODY_OK) {
                std::fprintf(stderr, "nbody_frame_acquire: %s\n", nbody_last_error());
                return 1;
            }
            physicsEnd = Clock::now();
            shown = (int)step_index - 1;
        }
        if (!dump_dir.empty()) {
            char name[64];
            std::snprintf(name, sizeof(name), "/frame_%05d.pgm", shown);
            dump_pgm(dump_dir + name, fx, fy, (size_t)n, std::min(sc.windowSize, 2048), sc.windowSize * 0.5f);
        }
        const auto frameEnd = Clock::now();
        const double frameMs = std::chrono::duration<double, std::milli>(frameEnd - frameStart).count();
        const double physicsMs = std::chrono::duration<double, std::milli>(physicsEnd - physicsStart).count();
        ++frameCount;
        elapsed += frameMs * 1e-3;
        if (elapsed >= 1.0 || f == frames - 1) {   // main.cpp:676-700 (+ one line at the end of a finite run)
            char buf[256];
            std::snprintf(buf, sizeof(buf), "FPS: %.2f | Physics: %.2f ms | Total frame: %.2f ms | N=%d | Mode: %s",
                          frameCount / elapsed, physicsMs, frameMs, n, cfg.use_barnes_hut ? "BarnesHut" : "N^2");
            std::cout << buf << std::endl;
            if (logFile.is_open()) { logFile << buf << std::endl; logFile.flush(); }
            frameCount = 0;
            elapsed = 0.0;
        }
    }
    if (async_frames && frames > 1) {              // the last frame is still in flight
        const float *fx = nullptr, *fy = nullptr;
        int64_t step_index = 0;
        if (nbody_frame_acquire(ctx, &fx, &fy, &step_index) != NBODY_OK) {
            std::fprintf(stderr, "nbody_frame_acquire: %s\n", nbody_last_error());
            return 1;
        }
        if (!dump_dir.empty()) {
            char name[64];
            std::snprintf(name, sizeof(name), "/frame_%05d.pgm", (int)step_index - 1);
            dump_pgm(dump_dir + name, fx, fy, (size_t)n, std::min(sc.windowSize, 2048), sc.windowSize * 0.5f);
        }
    }
    if (nbody_sync(ctx) != NBODY_OK) { std::fprintf(stderr, "nbody_sync: %s\n", nbody_last_error()); return 1; }
    nbody_destroy(ctx);
    return 0;
}
