"""Initial conditions of the reference's harness, Python counterpart of
host/initial_conditions.hpp: the seeded disc of benchmark.cpp:312-339 / main.cpp:418-441.

Per body, in order: angle = U*2*pi, r = U*spread (two mt19937 draws through
uniform_real_distribution<float>(0,1)), pos = (r cos, r sin), v = sqrt(G*mass*n/(r+50)), velocity = unit tangent
(-y, x)/|.| * v * vel_factor (0.3 in benchmark.cpp:330-331, 0.05 in main.cpp:431-432), m = mass, radius = 8.
The reference seeds from std::random_device (benchmark.cpp:312); here the seed is an argument.
numpy's legacy RandomState(seed) is the same init_genrand-seeded MT19937 as std::mt19937(seed), and its full-range
uint32 draws are the raw outputs, so positions agree with the C++ generator to the rounding of cos/sin."""
import numpy as np

G_LOCAL = 1.0          # benchmark.cpp:297


def uniform01(raw):
    """libstdc++ generate_canonical<float, 24> over one 32-bit draw: x / 2^32 in float, 1.0 mapped below 1."""
    r = raw.astype(np.float32) / np.float32(4294967296.0)
    return np.where(r >= np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(0.0)), r).astype(np.float32)


def disc(n, spread=400.0, mass=1000.0, seed=12345, vel_factor=0.3):
    """Returns (m, posX, posY, velX, velY, radius), float32[n] each."""
    raw = np.random.RandomState(int(seed)).randint(0, 2 ** 32, size=2 * n, dtype=np.uint32)
    u = uniform01(raw)
    angle = ((u[0::2] * np.float32(2.0)).astype(np.float64) * np.pi).astype(np.float32)   # :319
    r = u[1::2] * np.float32(spread)                                                       # :320
    x = (r.astype(np.float64) * np.cos(angle.astype(np.float64))).astype(np.float32)       # :321
    y = (r.astype(np.float64) * np.sin(angle.astype(np.float64))).astype(np.float32)       # :322
    v = np.sqrt(np.float32(G_LOCAL) * np.float32(mass) * np.float32(n) / (r + np.float32(50.0)), dtype=np.float32)  # :324
    tx, ty = -y, x                                                                          # :325-326
    ln = np.sqrt(tx * tx + ty * ty, dtype=np.float32)
    il = np.where(ln > 0, np.float32(1.0) / np.where(ln > 0, ln, np.float32(1.0)), np.float32(1.0)).astype(np.float32)
    vf = np.float32(vel_factor)
    return (np.full(n, mass, np.float32), x, y, (tx * il * v * vf).astype(np.float32), (ty * il * v * vf).astype(np.float32),
            np.full(n, 8.0, np.float32))                                                    # :330-338
