/* TEST INFRASTRUCTURE — CPU oracle for the N-Body physics step. NOT product code.
 *
 * Plain-C restatement of the algorithm in /root/reference/benchmark.cpp (fejzo9/N-Body).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product (n-body_amd/csrc) never links it.
 *
 * Parity pinning: the reference ships no golden vectors or tests, so this
 * restatement is pinned against the reference ITSELF, compiled from its own source by
 * oracle/build_ref.sh into oracle/_ref/libnbody_ref.so and run in the build container;
 * the resulting vectors are committed under tests/golden/ (generator:
 * tests/golden/make_golden.py) and tests/test_oracle.py checks this file against them.
 *
 * Two flavours of every formula:
 *   *_f64 : double-precision evaluation ("truth" for tolerance gates; the reference's
 *           -ffast-math build uses a 12-bit vrcpps, SURVEY.md §8c, so IT is the inaccurate party)
 *   *_f32 : float evaluation in the reference's operation order with exact sqrt/div.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define G_LOCAL 1.0f            /* benchmark.cpp:16 */
#define VEC_STRIDE 8            /* benchmark.cpp:21 */

/* ------------------------------------------------------------------------------------ */
/* A8: initial conditions, benchmark.cpp:312-339 (vel_factor 0.3) / main.cpp:411-441     */
/* (vel_factor 0.05).  std::mt19937 + libstdc++ uniform_real_distribution<float>(0,1):   */
/* one 32-bit draw per variate, value = float(u32) / 2^32, clamped below 1.              */
/* ------------------------------------------------------------------------------------ */
typedef struct { uint32_t s[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t *g, uint32_t seed)
{
    g->s[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->s[i] = 1812433253u * (g->s[i - 1] ^ (g->s[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->s[i] & 0x80000000u) | (g->s[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = g->s[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            g->s[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->s[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static float mt_uniform01(mt19937_t *g)
{
    float r = (float)mt_next(g) / 4294967296.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r;
}

/* exposed so the test can pin the generator on its own (C++11 [rand.predef]: the 10000th
 * consecutive output of a default-seeded (5489) mt19937 is 4123659995) */
void oracle_mt19937_raw(uint32_t seed, int count, uint32_t *out)
{
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < count; ++i) out[i] = mt_next(&g);
}

void oracle_mt19937_uniform01(uint32_t seed, int count, float *out)
{
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < count; ++i) out[i] = mt_uniform01(&g);
}

void oracle_init_disc(int n, float spread, float mass, uint32_t seed, float vel_factor,
                      float *m, float *px, float *py, float *vx, float *vy, float *radius)
{
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < n; ++i) {
        /* benchmark.cpp:319 `dist01(rng) * 2 * M_PI`: float*int, then * double M_PI */
        float angle = (float)((double)(mt_uniform01(&g) * 2.0f) * 3.14159265358979323846);
        float r = mt_uniform01(&g) * spread;                  /* :320 */
        /* :321-322 unqualified cos/sin on a float resolve to the double overloads */
        float x = (float)((double)r * cos((double)angle));
        float y = (float)((double)r * sin((double)angle));
        float v = sqrtf(G_LOCAL * mass * (float)n / (r + 50.0f)); /* :324 */
        float tx = -y, ty = x;                                 /* :325-326 */
        float len = sqrtf(tx * tx + ty * ty);
        if (len > 0) { float il = 1.0f / len; tx *= il; ty *= il; }  /* :328; -ffast-math => one reciprocal */
        m[i] = mass;
        px[i] = x;
        py[i] = y;
        vx[i] = tx * v * vel_factor;                           /* :330-331 */
        vy[i] = ty * v * vel_factor;
        radius[i] = 8.0f;                                      /* :338 */
    }
}

/* ------------------------------------------------------------------------------------ */
/* A1: naive pair force, benchmark.cpp:434-473.                                           */
/*   a_i = sum_j G m_j (r_j - r_i)/d / (max(d, max(R_i,R_j))^2 + 0.001f^2), d2 == 0 masked */
/* ref_compat != 0 reproduces the tail quirk (:423): targets i >= n - n%8 get no force.   */
/* ------------------------------------------------------------------------------------ */
static int target_limit(int n, int ref_compat) { return ref_compat ? n - (n % VEC_STRIDE) : n; }

void oracle_naive_accel_f64_range(int n, const float *m, const float *px, const float *py,
                                  const float *radius, int i0, int i1, double *ax, double *ay)
{
    const double soft2 = (double)(0.001f * 0.001f);            /* :453 */
#pragma omp parallel for schedule(static)
    for (int i = i0; i < i1; ++i) {
        double xi = px[i], yi = py[i], ri = radius[i], sx = 0.0, sy = 0.0;
        for (int j = 0; j < n; ++j) {
            double dx = (double)px[j] - xi, dy = (double)py[j] - yi;
            double d2 = dx * dx + dy * dy;
            if (!(d2 > 0.0)) continue;                          /* :468 */
            double d = sqrt(d2);
            double rr = ri > (double)radius[j] ? ri : (double)radius[j];
            double sd = d > rr ? d : rr;
            double f = ((double)G_LOCAL * (double)m[j]) / (sd * sd + soft2);
            sx += dx / d * f;
            sy += dy / d * f;
        }
        ax[i - i0] = sx;
        ay[i - i0] = sy;
    }
}

void oracle_naive_accel_f64(int n, const float *m, const float *px, const float *py,
                            const float *radius, int ref_compat, double *ax, double *ay)
{
    int lim = target_limit(n, ref_compat);
    oracle_naive_accel_f64_range(n, m, px, py, radius, 0, lim, ax, ay);
    for (int i = lim; i < n; ++i) ax[i] = ay[i] = 0.0;
}

/* arbitrary subset of targets (used for sampled checks at N = 2^20) */
void oracle_naive_accel_f64_subset(int n, const float *m, const float *px, const float *py,
                                   const float *radius, const int *idx, int nidx,
                                   double *ax, double *ay)
{
    const double soft2 = (double)(0.001f * 0.001f);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nidx; ++k) {
        int i = idx[k];
        double xi = px[i], yi = py[i], ri = radius[i], sx = 0.0, sy = 0.0;
        for (int j = 0; j < n; ++j) {
            double dx = (double)px[j] - xi, dy = (double)py[j] - yi;
            double d2 = dx * dx + dy * dy;
            if (!(d2 > 0.0)) continue;
            double d = sqrt(d2);
            double rr = ri > (double)radius[j] ? ri : (double)radius[j];
            double sd = d > rr ? d : rr;
            double f = ((double)G_LOCAL * (double)m[j]) / (sd * sd + soft2);
            sx += dx / d * f;
            sy += dy / d * f;
        }
        ax[k] = sx;
        ay[k] = sy;
    }
}

void oracle_naive_accel_f32(int n, const float *m, const float *px, const float *py,
                            const float *radius, int ref_compat, float *ax, float *ay)
{
    int lim = target_limit(n, ref_compat);
    const float soft2 = 0.001f * 0.001f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < lim; ++i) {
        float xi = px[i], yi = py[i], ri = radius[i], sx = 0.0f, sy = 0.0f;
        for (int j = 0; j < n; ++j) {                           /* j ascending, :434 */
            float dx = px[j] - xi, dy = py[j] - yi;             /* :442-443 */
            float d2 = fmaf(dx, dx, dy * dy);                   /* :446 */
            if (!(d2 > 0.0f)) continue;                         /* :468 */
            float d = sqrtf(d2);                                /* :447 */
            float md = ri > radius[j] ? ri : radius[j];         /* :450 */
            float sd = d > md ? d : md;                         /* :451 */
            float den = sd * sd + soft2;                        /* :452-453 */
            float f = (G_LOCAL * m[j]) / den;                   /* :456-457 */
            float inv = 1.0f / d;                               /* :460 (exact, not vrcpps) */
            sx += (dx * inv) * f;                               /* :461-471 */
            sy += (dy * inv) * f;
        }
        ax[i] = sx;
        ay[i] = sy;
    }
    for (int i = lim; i < n; ++i) ax[i] = ay[i] = 0.0f;
}

/* ------------------------------------------------------------------------------------ */
/* A2: forward Euler, benchmark.cpp:483-510.  Both updates use OLD values.                */
/* ------------------------------------------------------------------------------------ */
void oracle_euler_f32(int n, float *px, float *py, float *vx, float *vy,
                      const float *ax, const float *ay, float dt, int ref_compat)
{
    int lim = target_limit(n, ref_compat);                      /* :486 */
    for (int i = 0; i < lim; ++i) {
        /* add(x, mul(v, dt)) in the source; the reference's -O3 -ffast-math build contracts it into one FMA
         * (checked against the golden step-1 state: bit-exact with fmaf, 1-ulp misses without) */
        px[i] = fmaf(vx[i], dt, px[i]);                         /* :498-499 */
        py[i] = fmaf(vy[i], dt, py[i]);
        vx[i] = fmaf(ax[i], dt, vx[i]);                         /* :502-503 */
        vy[i] = fmaf(ay[i], dt, vy[i]);
    }
}

/* nsteps of (A1 in f64 rounded to f32) + A2: the short-horizon trajectory truth */
void oracle_naive_steps(int n, const float *m, float *px, float *py, float *vx, float *vy,
                        const float *radius, float dt, int nsteps, int ref_compat)
{
    double *ax = (double *)malloc(sizeof(double) * (size_t)n);
    double *ay = (double *)malloc(sizeof(double) * (size_t)n);
    float *fx = (float *)malloc(sizeof(float) * (size_t)n);
    float *fy = (float *)malloc(sizeof(float) * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        oracle_naive_accel_f64(n, m, px, py, radius, ref_compat, ax, ay);
        for (int i = 0; i < n; ++i) { fx[i] = (float)ax[i]; fy[i] = (float)ay[i]; }
        oracle_euler_f32(n, px, py, vx, vy, fx, fy, dt, ref_compat);
    }
    free(ax); free(ay); free(fx); free(fy);
}

/* ------------------------------------------------------------------------------------ */
/* Diagnostics (defined by this build; the reference computes none).         */
/* Potential consistent with the hard clamp: force magnitude m_i m_j / max(d,R)^2          */
/*   phi(d) = -m_i m_j / d                      d >= R                                     */
/*   phi(d) = -m_i m_j (1/R + (R - d)/R^2)      d <  R                                     */
/* out = {Ekin, Epot, Px, Py, Lz}                                                          */
/* ------------------------------------------------------------------------------------ */
void oracle_diagnostics(int n, const float *m, const float *px, const float *py,
                        const float *vx, const float *vy, const float *radius, double out[5])
{
    double ek = 0, ep = 0, mx = 0, my = 0, lz = 0;
#pragma omp parallel for schedule(static) reduction(+ : ek, ep, mx, my, lz)
    for (int i = 0; i < n; ++i) {
        double mi = m[i], xi = px[i], yi = py[i], ui = vx[i], wi = vy[i];
        ek += 0.5 * mi * (ui * ui + wi * wi);
        mx += mi * ui;
        my += mi * wi;
        lz += mi * (xi * wi - yi * ui);
        double e = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            double dx = px[j] - xi, dy = py[j] - yi;
            double d = sqrt(dx * dx + dy * dy);
            double rr = radius[i] > radius[j] ? radius[i] : radius[j];
            double phi = d >= rr ? 1.0 / d : (1.0 / rr + (rr - d) / (rr * rr));
            e -= (double)G_LOCAL * mi * (double)m[j] * phi;
        }
        ep += 0.5 * e;
    }
    out[0] = ek; out[1] = ep; out[2] = mx; out[3] = my; out[4] = lz;
}

/* The two reference programs differ in ONE arithmetic detail of the tree: main.cpp normalises a cell's centre of
 * mass with a reciprocal and two products (main.cpp:173-175), benchmark.cpp with two divisions (benchmark.cpp:139-140).
 * 0 = benchmark.cpp (default), 1 = main.cpp. */
static int g_main_cpp_flavour = 0;
void oracle_set_main_cpp_flavour(int on) { g_main_cpp_flavour = on != 0; }


/* ------------------------------------------------------------------------------------ */
/* A4: quadtree, benchmark.cpp:52-173.  48-byte nodes, fixed root box, capacity-1 leaves, */
/* inclusive bounds on all four sides, children TL,TR,BL,BR, COM refreshed on unwind.     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    float mass, comX, comY;
    float tlX, tlY, brX, brY;
    int32_t bodyIndex;
    int32_t children[4];
} oracle_node;                                                  /* = QuadNode, :52-67 */

typedef struct {
    oracle_node *pool;
    int cap;       /* pool capacity in nodes (reference: 32 n, :148) */
    int used;      /* = lastPoolIndex */
} oracle_tree;

static void node_make(oracle_node *nd, float tlX, float tlY, float brX, float brY)
{
    nd->mass = 0; nd->comX = 0; nd->comY = 0;
    nd->tlX = tlX; nd->tlY = tlY; nd->brX = brX; nd->brY = brY;
    nd->bodyIndex = -1;
    nd->children[0] = nd->children[1] = nd->children[2] = nd->children[3] = -1;
}

static void tree_insert(oracle_tree *t, int ni, int b, const float *m, const float *px, const float *py)
{
    oracle_node *nd = &t->pool[ni];
    float x = px[b], y = py[b];
    if (x < nd->tlX || x > nd->brX || y < nd->tlY || y > nd->brY) return;   /* :83-85 */
    int leaf = nd->children[0] == -1;
    if (nd->bodyIndex == -1 && leaf) {                                        /* :87-94 */
        nd->bodyIndex = b; nd->mass = m[b]; nd->comX = x; nd->comY = y;
        return;
    }
    if (leaf) {                                                               /* :96-115 */
        float midX = (nd->tlX + nd->brX) / 2.0f, midY = (nd->tlY + nd->brY) / 2.0f;
        if ((size_t)t->used + 4 >= (size_t)t->cap) return;                    /* :102 */
        int c0 = t->used; t->used += 4;
        node_make(&t->pool[c0 + 0], nd->tlX, nd->tlY, midX, midY);
        node_make(&t->pool[c0 + 1], midX, nd->tlY, nd->brX, midY);
        node_make(&t->pool[c0 + 2], nd->tlX, midY, midX, nd->brY);
        node_make(&t->pool[c0 + 3], midX, midY, nd->brX, nd->brY);
        for (int k = 0; k < 4; ++k) nd->children[k] = c0 + k;
        int old = nd->bodyIndex;
        nd->bodyIndex = -1;
        for (int k = 0; k < 4; ++k) tree_insert(t, nd->children[k], old, m, px, py);
    }
    for (int k = 0; k < 4; ++k)                                               /* :117-119 */
        if (nd->children[k] != -1) tree_insert(t, nd->children[k], b, m, px, py);
    float ms = 0.0f, sx = 0.0f, sy = 0.0f;                                    /* :122-135 */
    for (int k = 0; k < 4; ++k) {
        if (nd->children[k] != -1) {
            const oracle_node *c = &t->pool[nd->children[k]];
            /* the reference's -O3 -ffast-math build contracts these into FMAs (checked: with fmaf the
             * root COM of the golden N=1024 tree is reproduced to the last bit, without it it is not) */
            ms += c->mass; sx = fmaf(c->comX, c->mass, sx); sy = fmaf(c->comY, c->mass, sy);
        }
    }
    nd->mass = ms;
    if (ms > 0.0f) {
        if (g_main_cpp_flavour) {                          /* main.cpp:173-175: one reciprocal, two products */
            float inv = 1.0f / ms;
            nd->comX = sx * inv; nd->comY = sy * inv;
        } else {                                           /* benchmark.cpp:137-141: two divisions */
            nd->comX = sx / ms; nd->comY = sy / ms;
        }
    }
}

oracle_tree *oracle_tree_create(int n)
{
    oracle_tree *t = (oracle_tree *)malloc(sizeof(oracle_tree));
    t->cap = n * 32;                                                          /* :148 */
    if (t->cap < 8) t->cap = 8;
    t->pool = (oracle_node *)malloc(sizeof(oracle_node) * (size_t)t->cap);
    t->used = 1;
    return t;
}

void oracle_tree_destroy(oracle_tree *t) { if (t) { free(t->pool); free(t); } }

void oracle_tree_build(oracle_tree *t, int n, float spread, const float *m, const float *px, const float *py)
{
    t->used = 1;                                                              /* :168-169 */
    node_make(&t->pool[0], -spread, -spread, spread, spread);
    for (int i = 0; i < n; ++i) tree_insert(t, 0, i, m, px, py);              /* :171-172 */
}

int oracle_tree_nodes_used(const oracle_tree *t) { return t->used; }
const void *oracle_tree_pool(const oracle_tree *t) { return t->pool; }

/* depth / leaf-body count / duplicate statistics for tree invariants */
void oracle_tree_stats(const oracle_tree *t, int out[4])
{
    int maxd = 0, bodies = 0, leaves = 0;
    int cap = 1 << 16, top = 0;
    int *st = (int *)malloc(sizeof(int) * 2 * (size_t)cap);
    st[0] = 0; st[1] = 0; top = 1;
    while (top) {
        --top;
        int ni = st[2 * top], d = st[2 * top + 1];
        const oracle_node *nd = &t->pool[ni];
        if (d > maxd) maxd = d;
        if (nd->children[0] == -1) { ++leaves; if (nd->bodyIndex != -1) ++bodies; continue; }
        for (int k = 0; k < 4; ++k) {
            if (top + 1 >= cap) { cap *= 2; st = (int *)realloc(st, sizeof(int) * 2 * (size_t)cap); }
            st[2 * top] = nd->children[k]; st[2 * top + 1] = d + 1; ++top;
        }
    }
    free(st);
    out[0] = t->used; out[1] = maxd; out[2] = leaves; out[3] = bodies;
}

/* ------------------------------------------------------------------------------------ */
/* A5 + A6: group-of-8 walk and node kernel, benchmark.cpp:176-277.                       */
/* mode f64: same accepted-node set as the f32 vote (the vote is evaluated in float, as    */
/* the reference does), contributions summed in double.                                   */
/* ------------------------------------------------------------------------------------ */
static void bh_group(const oracle_tree *t, const int *ids, int lanes, const float *px, const float *py,
                     const float *radius, float theta, int use_f64, double *ax, double *ay,
                     long long *visits, long long *accepts, unsigned long long *accept_hash)
{
    const float eps = 1e-6f, soft = 0.001f;
    double accx[VEC_STRIDE] = {0}, accy[VEC_STRIDE] = {0};
    float faccx[VEC_STRIDE] = {0}, faccy[VEC_STRIDE] = {0};
    int cap = 256, top = 0;
    int *st = (int *)malloc(sizeof(int) * (size_t)cap);
    st[top++] = 0;                                                            /* :238 */
    while (top) {
        int ni = st[--top];                                                   /* :244-245 */
        const oracle_node *nd = &t->pool[ni];
        if (visits) ++*visits;
        if (nd->mass == 0.0f) continue;                                       /* :248 */
        float width = nd->brX - nd->tlX;                                      /* :251 */
        int all_far = 1;
        for (int k = 0; k < lanes; ++k) {
            float dx = nd->comX - px[ids[k]], dy = nd->comY - py[ids[k]];
            float d = sqrtf(fmaf(dx, dx, dy * dy));                           /* :256-257 (contracted) */
            float ratio = width / (d + eps);                                  /* :259-260 */
            if (!(ratio < theta)) all_far = 0;                                /* :262-264 */
        }
        int leaf = nd->children[0] == -1;
        if (leaf || all_far) {                                                /* :266 */
            if (accepts) ++*accepts;
            if (accept_hash) *accept_hash = (*accept_hash ^ (unsigned long long)(unsigned)ni) * 1099511628211ull;
            for (int k = 0; k < lanes; ++k) {                                 /* A6, :176-225 */
                if ((float)nd->bodyIndex == (float)ids[k]) continue;     /* :197-198 */
                if (use_f64) {
                    double dx = (double)nd->comX - px[ids[k]], dy = (double)nd->comY - py[ids[k]];
                    double d = sqrt(dx * dx + dy * dy);
                    double sd = d > (double)radius[ids[k]] ? d : (double)radius[ids[k]];
                    double f = ((double)G_LOCAL * nd->mass) / (sd * sd + (double)(soft * soft));
                    double inv = 1.0 / (d + (double)eps);
                    accx[k] += dx * inv * f;
                    accy[k] += dy * inv * f;
                } else {
                    float dx = nd->comX - px[ids[k]], dy = nd->comY - py[ids[k]];
                    float d = sqrtf(fmaf(dx, dx, dy * dy));                   /* :193-194 */
                    float sd = d > radius[ids[k]] ? d : radius[ids[k]]; /* :201-202 */
                    float f = (G_LOCAL * nd->mass) / fmaf(sd, sd, soft * soft); /* :203-209 (contracted) */
                    float inv = 1.0f / (d + eps);                             /* :212-213 */
                    faccx[k] += (dx * inv) * f;                               /* :215-225 */
                    faccy[k] += (dy * inv) * f;
                }
            }
        } else {
            for (int k = 0; k < 4; ++k) {                                     /* :272-274 */
                if (nd->children[k] != -1) {
                    if (top >= cap) { cap *= 2; st = (int *)realloc(st, sizeof(int) * (size_t)cap); }
                    st[top++] = nd->children[k];
                }
            }
        }
    }
    free(st);
    for (int k = 0; k < lanes; ++k) {
        ax[ids[k]] = use_f64 ? accx[k] : (double)faccx[k];
        ay[ids[k]] = use_f64 ? accy[k] : (double)faccy[k];
    }
}

/* A7 force part: accelerations for every group of 8 (tail bodies get 0, :368).
 * group_size lets tests evaluate the conventional per-body walk (group_size = 1). */
void oracle_bh_accel(const oracle_tree *t, int n, const float *px, const float *py, const float *radius,
                     float theta, int use_f64, int group_size, double *ax, double *ay, long long counters[2])
{
    long long visits = 0, accepts = 0;
    if (group_size < 1 || group_size > VEC_STRIDE) group_size = VEC_STRIDE;
    int lim = n - (n % group_size);
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : visits, accepts)
    for (int i = 0; i < lim; i += group_size) {
        int ids[VEC_STRIDE];
        for (int k = 0; k < group_size; ++k) ids[k] = i + k;
        bh_group(t, ids, group_size, px, py, radius, theta, use_f64, ax, ay, &visits, &accepts, NULL);
    }
    for (int i = lim; i < n; ++i) ax[i] = ay[i] = 0.0;
    if (counters) { counters[0] = visits; counters[1] = accepts; }
}

/* The same walk with its counters kept PER GROUP of 8 (index = start / 8): nodes popped (:244-245), nodes accepted
 * (:266-268) and an order-dependent hash of the accepted node indices.  The compiled reference exports the same three
 * through oracle/ref_hooks.h (REF_VISIT / REF_ACCEPT), so its vote can be compared group by group. */
void oracle_bh_group_counters(const oracle_tree *t, int n, const float *px, const float *py, const float *radius,
                              float theta, int *grp_visits, int *grp_accepts, unsigned long long *grp_hash)
{
    int lim = n - (n % VEC_STRIDE);
    double *ax = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *ay = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < lim; i += VEC_STRIDE) {
        long long v = 0, a = 0;
        unsigned long long h = 0;
        int ids[VEC_STRIDE];
        for (int k = 0; k < VEC_STRIDE; ++k) ids[k] = i + k;
        bh_group(t, ids, VEC_STRIDE, px, py, radius, theta, 0, ax, ay, &v, &a, &h);
        grp_visits[i / VEC_STRIDE] = (int)v;
        grp_accepts[i / VEC_STRIDE] = (int)a;
        if (grp_hash) grp_hash[i / VEC_STRIDE] = h;
    }
    free(ax); free(ay);
}

/* ---- bh_sorted (NO reference counterpart: nbody_config.bh_sorted) ------------------------------ */
/* Quadrant-path key of every body: the digits TL,TR,BL,BR = 0..3 (:105-108) of its descent through the fixed box with
 * the reference's comparisons and float midpoints (:83-85, :98-99), 31 levels, 2 bits each; a body outside the box
 * gets bit 63 | index.  Sorting bodies by (key, index) is the order in which the GPU build lists them. */
void oracle_path_keys(int n, float spread, const float *px, const float *py, unsigned long long *keys)
{
    for (int i = 0; i < n; ++i) {
        float x = px[i], y = py[i], tlx = -spread, tly = -spread, brx = spread, bry = spread;
        if (x < tlx || x > brx || y < tly || y > bry) { keys[i] = (1ull << 63) | (unsigned long long)(unsigned)i; continue; }
        unsigned long long key = 0;
        for (int l = 0; l < 31; ++l) {
            float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;
            int qx = x > mx, qy = y > my;
            key = (key << 2) | (unsigned long long)(qy * 2 + qx);
            if (qx) tlx = mx; else brx = mx;
            if (qy) tly = my; else bry = my;
        }
        keys[i] = key;
    }
}

/* The reference's walk and node kernel (:176-277) over groups of 8 consecutive entries of `order` instead of 8
 * consecutive body indices; a last, shorter group votes with the lanes it has.  Counters per group position. */
void oracle_bh_accel_ordered(const oracle_tree *t, int n, const float *px, const float *py, const float *radius,
                             float theta, const int *order, double *ax, double *ay, long long counters[2],
                             int *grp_visits, int *grp_accepts)
{
    long long visits = 0, accepts = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : visits, accepts)
    for (int p = 0; p < n; p += VEC_STRIDE) {
        int lanes = n - p < VEC_STRIDE ? n - p : VEC_STRIDE;
        long long v = 0, a = 0;
        bh_group(t, order + p, lanes, px, py, radius, theta, 0, ax, ay, &v, &a, NULL);
        if (grp_visits) grp_visits[p / VEC_STRIDE] = (int)v;
        if (grp_accepts) grp_accepts[p / VEC_STRIDE] = (int)a;
        visits += v; accepts += a;
    }
    if (counters) { counters[0] = visits; counters[1] = accepts; }
}

/* A7: nsteps of build -> walk -> Euler, benchmark.cpp:354-403 (use_parallel == 0: build only, :359) */
void oracle_bh_steps_counted(int n, const float *m, float *px, float *py, float *vx, float *vy, const float *radius,
                             float spread, float theta, float dt, int nsteps, int use_parallel,
                             long long *step_visits, long long *step_accepts)
{
    oracle_tree *t = oracle_tree_create(n);
    double *ax = (double *)malloc(sizeof(double) * (size_t)n);
    double *ay = (double *)malloc(sizeof(double) * (size_t)n);
    float *fx = (float *)malloc(sizeof(float) * (size_t)n);
    float *fy = (float *)malloc(sizeof(float) * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        long long cnt[2] = {0, 0};
        oracle_tree_build(t, n, spread, m, px, py);
        if (use_parallel) {
            oracle_bh_accel(t, n, px, py, radius, theta, 0, VEC_STRIDE, ax, ay, cnt);
            for (int i = 0; i < n; ++i) { fx[i] = (float)ax[i]; fy[i] = (float)ay[i]; }
            oracle_euler_f32(n, px, py, vx, vy, fx, fy, dt, 1);
        }
        if (step_visits) step_visits[s] = cnt[0];
        if (step_accepts) step_accepts[s] = cnt[1];
    }
    free(ax); free(ay); free(fx); free(fy);
    oracle_tree_destroy(t);
}

void oracle_bh_steps(int n, const float *m, float *px, float *py, float *vx, float *vy, const float *radius,
                     float spread, float theta, float dt, int nsteps, int use_parallel)
{
    oracle_bh_steps_counted(n, m, px, py, vx, vy, radius, spread, theta, dt, nsteps, use_parallel, NULL, NULL);
}
