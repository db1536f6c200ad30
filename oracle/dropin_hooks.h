/* TEST INFRASTRUCTURE — not product code.
 *
 * Hooks for the DROP-IN build of oracle/build_ref.sh: the reference's own benchmark.cpp, streamed from
 * /root/reference through awk into g++ exactly like the plain reference build, but with the edit INTEGRATION.md
 * describes applied — `QuadTree tree; tree.init(n, spread);` (benchmark.cpp:341-342) becomes nbody_create +
 * nbody_upload and the whole physics block (benchmark.cpp:354-511) becomes nbody_step + nbody_sync.  Generator, loop,
 * timing and logging remain the reference's machine code; the physics is include/nbody.h on the GPU.
 * tests/test_gpu_dropin_reference.py runs both builds on the same seed and compares them.
 */
#pragma once
#include "ref_hooks.h"
#include "../include/nbody.h"
#include <cstdio>

#define REF_DROPIN_CREATE(data)                                                                               \
    nbody_config gpu_cfg_;                                                                                    \
    nbody_default_config(&gpu_cfg_);                                                                          \
    gpu_cfg_.num_bodies = n; gpu_cfg_.spread = spread; gpu_cfg_.mass = mass; gpu_cfg_.dt = dt;                \
    gpu_cfg_.theta = theta; gpu_cfg_.use_barnes_hut = useBarnesHut; gpu_cfg_.use_parallel = useParallel;      \
    nbody_ctx *gpu_ = nullptr;                                                                                \
    if (nbody_create(&gpu_cfg_, &gpu_) != NBODY_OK ||                                                         \
        nbody_upload(gpu_, data.m.data(), data.posX.data(), data.posY.data(), data.velX.data(),               \
                     data.velY.data(), data.radius.data()) != NBODY_OK) {                                     \
        std::fprintf(stderr, "drop-in: %s\n", nbody_last_error());                                            \
        return -1;                                                                                            \
    }

#define REF_DROPIN_STEP()                                                          \
    do {                                                                           \
        if (nbody_step(gpu_, 1) != NBODY_OK || nbody_sync(gpu_) != NBODY_OK) {     \
            std::fprintf(stderr, "drop-in: %s\n", nbody_last_error());             \
            return -2;                                                             \
        }                                                                          \
    } while (0)

/* before the reference's footer (benchmark.cpp:559): bring the state back into its vectors */
#define REF_DROPIN_FINISH(data)                                                                                  \
    do {                                                                                                         \
        if (nbody_download(gpu_, data.posX.data(), data.posY.data(), data.velX.data(), data.velY.data()) != NBODY_OK) return -3; \
        if (p->acc_x && p->acc_y && nbody_get_accel(gpu_, p->acc_x, p->acc_y) != NBODY_OK) return -4;            \
        nbody_destroy(gpu_);                                                                                     \
    } while (0)
