// benchmark_main.cpp — counterpart of the reference's console benchmark (benchmark.cpp:297-566) on the
// MI355X C-ABI.  Same parameter block (benchmark.cpp:300-306, overridable from the command line instead
// of by editing literals), same generator (benchmark.cpp:317-339, fixed seed), same per->=1 s log line
// `FPS: %.2f | Physics: %.2f ms | Total frame: %.2f ms` to stdout and ./log.txt (benchmark.cpp:534-556),
// but a FINITE loop (the reference's while(true) never reaches its footer, benchmark.cpp:349,559-564)
// and a closing summary with G interactions/s and the FP32-roofline fraction.
//
//   nbody_benchmark [--n N] [--naive|--bh] [--steps K] [--warmup W] [--dt x] [--theta x] [--spread x]
//                   [--mass x] [--seed s] [--no-ref-compat] [--no-log] [--dist]
//
// --dist: one process per GPU.  RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR and MASTER_PORT come from
// the launcher (mpirun-style env or `python -m torch.distributed.run --no-python ... nbody_benchmark --dist`); rank r
// owns the i-shard nbody_shard_range(N, P, r) on GPU LOCAL_RANK, the RCCL id travels from rank 0 over a TCP socket
// on MASTER_PORT + 1 (NBODY_COMM_PORT overrides), and every step is followed by nbody_exchange_positions.
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

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

static std::string now_string()   // benchmark.cpp:281-294
{
    std::time_t t = std::time(nullptr);
    std::tm tmv;
    localtime_r(&t, &tmv);
    std::ostringstream os;
    os << std::put_time(&tmv, "%Y-%m-%d %H:%M:%S");
    return os.str();
}

// Hand rank 0's 128-byte RCCL id to the other ranks: rank 0 listens, the others connect (retrying while rank 0 is
// still starting) and read it.  Returns false on failure.
static bool share_comm_id(int rank, int nranks, const char* addr, int port, char id[NBODY_COMM_ID_BYTES])
{
    if (nranks == 1) return true;
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1) { std::fprintf(stderr, "bad MASTER_ADDR %s (IPv4 literal expected)\n", addr); return false; }
    if (rank == 0) {
        const int ls = socket(AF_INET, SOCK_STREAM, 0);
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sa.sin_addr.s_addr = htonl(INADDR_ANY);
        if (ls < 0 || bind(ls, (sockaddr*)&sa, sizeof(sa)) != 0 || listen(ls, nranks) != 0) { std::perror("rank 0: bind/listen"); return false; }
        for (int k = 1; k < nranks; ++k) {
            const int c = accept(ls, nullptr, nullptr);
            if (c < 0) { std::perror("accept"); close(ls); return false; }
            const ssize_t w = write(c, id, NBODY_COMM_ID_BYTES);
            close(c);
            if (w != NBODY_COMM_ID_BYTES) { close(ls); return false; }
        }
        close(ls);
        return true;
    }
    for (int attempt = 0; attempt < 300; ++attempt) {          // up to ~30 s for rank 0 to come up
        const int c = socket(AF_INET, SOCK_STREAM, 0);
        if (c >= 0 && connect(c, (sockaddr*)&sa, sizeof(sa)) == 0) {
            size_t got = 0;
            while (got < NBODY_COMM_ID_BYTES) {
                const ssize_t r = read(c, id + got, NBODY_COMM_ID_BYTES - got);
                if (r <= 0) break;
                got += (size_t)r;
            }
            close(c);
            return got == NBODY_COMM_ID_BYTES;
        }
        if (c >= 0) close(c);
        usleep(100 * 1000);
    }
    std::fprintf(stderr, "rank %d: could not reach rank 0 at %s:%d\n", rank, addr, port);
    return false;
}

static int env_int(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
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
    bool write_log = true, dist = false;
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
        else if (!std::strcmp(argv[i], "--dist")) dist = true;
        else if (!std::strcmp(argv[i], "--selftest-bootstrap")) {
            // no GPU involved: pass a recognisable 128-byte pattern from rank 0 to every rank and print what arrived
            const int r = env_int("RANK", 0), p = env_int("WORLD_SIZE", 1);
            char id[NBODY_COMM_ID_BYTES];
            for (int k = 0; k < NBODY_COMM_ID_BYTES; ++k) id[k] = r == 0 ? (char)(k * 7 + 3) : 0;
            const char* addr = std::getenv("MASTER_ADDR");
            if (!share_comm_id(r, p, addr && *addr ? addr : "127.0.0.1", env_int("NBODY_COMM_PORT", env_int("MASTER_PORT", 29500) + 1), id))
                return 1;
            unsigned sum = 0;
            for (int k = 0; k < NBODY_COMM_ID_BYTES; ++k) sum = sum * 31u + (unsigned char)id[k];
            std::printf("rank %d of %d: id checksum %u\n", r, p, sum);
            return 0;
        }
        else { std::fprintf(stderr, "unknown flag %s\n", argv[i]); return 2; }
    }
    const int n = cfg.num_bodies;
    const int rank = dist ? env_int("RANK", 0) : 0, nranks = dist ? env_int("WORLD_SIZE", 1) : 1;
    if (dist) {
        cfg.device = env_int("LOCAL_RANK", rank);
        NB_CHECK(nbody_shard_range(n, nranks, rank, &cfg.shard_begin, &cfg.shard_count));
        if (rank != 0) write_log = false;
    }

    nbody_host::BodyStore data;
    nbody_host::fill_disc(data, n, cfg.spread, cfg.mass, seed, 0.3f);   // benchmark.cpp:317-339

    nbody_ctx* ctx = nullptr;
    NB_CHECK(nbody_create(&cfg, &ctx));
    if (dist) {
        char id[NBODY_COMM_ID_BYTES] = {0};
        if (rank == 0) NB_CHECK(nbody_comm_unique_id(id));
        const char* addr = std::getenv("MASTER_ADDR");
        const int port = env_int("NBODY_COMM_PORT", env_int("MASTER_PORT", 29500) + 1);
        if (!share_comm_id(rank, nranks, addr && *addr ? addr : "127.0.0.1", port, id)) return 1;
        NB_CHECK(nbody_comm_init(ctx, rank, nranks, id));
    }
    NB_CHECK(nbody_upload(ctx, data.m.data(), data.posX.data(), data.posY.data(), data.velX.data(),
                          data.velY.data(), data.radius.data()));
    // one step = local force + Euler on the owned shard, then (sharded) the all-gather of the positions
    auto one_step = [&]() -> int {
        int rc = nbody_step(ctx, 1);
        if (rc == NBODY_OK && dist) rc = nbody_exchange_positions(ctx);
        return rc;
    };
    for (int w = 0; w < warmup; ++w) NB_CHECK(one_step());
    NB_CHECK(nbody_sync(ctx));

    std::ofstream logFile;
    if (write_log) logFile.open("log.txt", std::ios::app);   // benchmark.cpp:314

    using Clock = std::chrono::high_resolution_clock;
    int frameCount = 0;
    double elapsed = 0.0, physics_total_ms = 0.0;
    for (int s = 0; s < steps; ++s) {
        const auto frameStart = Clock::now();
        const auto physicsStart = Clock::now();
        NB_CHECK(one_step());              // replaces benchmark.cpp:354-511
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
            if (rank == 0) std::cout << buf << std::endl;
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
    if (rank == 0) std::cout << buf << (dist ? " | ranks=" + std::to_string(nranks) : std::string()) << std::endl;
    if (logFile.is_open()) {
        logFile << buf << std::endl;
        logFile << "=== Benchmark ended: " << now_string() << " ===" << std::endl << std::endl;   // benchmark.cpp:561
    }
    nbody_destroy(ctx);
    return 0;
}
