// initial_conditions.hpp — host SoA container and the reference's disc generator.
//
// BodyStore is the counterpart of BodyDataSoA (benchmark.cpp:24-49): one std::vector<float> per field,
// whose .data() pointers are what nbody_upload() takes.  fill_disc() reproduces the generator of
// benchmark.cpp:317-339 / main.cpp:418-441 (angle first, then radius, from one std::mt19937 through
// std::uniform_real_distribution<float>(0,1); tangential speed sqrt(G*mass*n/(r+50)) * vel_factor with
// vel_factor 0.3 in benchmark.cpp:330 and 0.05 in main.cpp:431) but takes the seed as a parameter —
// the reference seeds from std::random_device (benchmark.cpp:312), which no test can reproduce.
#pragma once
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

namespace nbody_host {

struct BodyStore {
    std::vector<float> m, posX, posY, velX, velY, radius;
    size_t num_bodies = 0;
    void resize(size_t n)
    {
        num_bodies = n;
        for (auto* v : {&m, &posX, &posY, &velX, &velY, &radius}) v->assign(n, 0.0f);
    }
};

inline void fill_disc(BodyStore& b, int n, float spread, float mass, uint32_t seed, float vel_factor)
{
    b.resize((size_t)n);
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> u01(0.0f, 1.0f);
    const double two_pi = 2.0 * 3.14159265358979323846;
    for (int i = 0; i < n; ++i) {
        const float phi = (float)((double)u01(gen) * two_pi);
        const float r = u01(gen) * spread;
        const float x = (float)((double)r * std::cos((double)phi));
        const float y = (float)((double)r * std::sin((double)phi));
        const float speed = std::sqrt(1.0f * mass * (float)n / (r + 50.0f)) * vel_factor;
        const float len = std::sqrt(x * x + y * y);
        const float tx = len > 0 ? -y / len : -y, ty = len > 0 ? x / len : x;
        b.m[i] = mass;
        b.posX[i] = x;
        b.posY[i] = y;
        b.velX[i] = tx * speed;
        b.velY[i] = ty * speed;
        b.radius[i] = 8.0f;
    }
}

}  // namespace nbody_host
