/* TEST INFRASTRUCTURE — not product code.
 *
 * Hook definitions for oracle/build_ref.sh.  That script streams the reference's
 * own /root/reference/benchmark.cpp through `awk` into `g++ -x c++ -` (the patched
 * text is never written to disk) and this header is force-included in front of it.
 * The patches only (a) turn the reference's `int main()` into a callable
 * `ref_benchmark_run(RefIO*)`, (b) replace its hard-coded parameters
 * (benchmark.cpp:300-306) and its random seed (benchmark.cpp:312) with RefIO fields,
 * (c) bound its `while(true)` (benchmark.cpp:349) by RefIO::nsteps and (d) insert the
 * REF_* hook macros below, which copy the reference's intermediate results out.
 * All physics executed is the reference's own machine code.
 */
#pragma once
#include <chrono>
#include <cstring>
#include <vector>
#include <immintrin.h>

extern "C" {
typedef struct RefIO {
    /* parameters: benchmark.cpp:300-306 */
    int   use_barnes_hut, use_parallel, n;
    float spread, mass, theta, dt;
    unsigned seed;          /* replaces std::random_device{}() at benchmark.cpp:312 */
    int   nsteps;           /* bound for the while(true) at benchmark.cpp:349       */
    int   log_enabled;      /* 0: silence the stdout/log.txt line (benchmark.cpp:534) */
    /* optional caller state, SoA, each float[n] (NULL => keep generator output) */
    const float *in_m, *in_px, *in_py, *in_vx, *in_vy, *in_radius;
    /* optional outputs (NULL => skipped) */
    float *init_m, *init_px, *init_py, *init_vx, *init_vy, *init_radius; /* generator output */
    float *out_px, *out_py, *out_vx, *out_vy;   /* state after nsteps                  */
    float *acc_x, *acc_y;                       /* accelerations evaluated in the LAST step */
    int   *tree_nodes_used;                     /* QuadTree::lastPoolIndex after last build */
    float *tree_root;                           /* [3] root mass, comX, comY            */
    void  *tree_pool_out;                       /* raw copy of the node pool (48 B/node) */
    int    tree_pool_cap;                       /* capacity of tree_pool_out in nodes   */
    double *step_ms;                            /* [nsteps] physicsEnd-physicsStart      */
    /* Barnes-Hut walk counters of the reference's OWN vote (benchmark.cpp:242-276), per group of 8 bodies
       (index = start_index / 8), accumulated over the steps of the run; NULL => not counted */
    int   *grp_visits;                          /* nodes popped, benchmark.cpp:244-245   */
    int   *grp_accepts;                         /* nodes handed to computeForce_kernel, :266-268 */
    unsigned long long *grp_hash;               /* order-dependent hash of the accepted node indices */
    long long *step_visits, *step_accepts;      /* [nsteps] totals of each step          */
} RefIO;
int ref_benchmark_run(RefIO *p);
}

/* The walk is a const member function of QuadTree with no RefIO in scope: the counter arrays are reached through
   file-scope pointers, set from RefIO after the generator (REF_AFTER_INIT).  One group is walked by one thread, so
   the per-group slots need no atomics. */
static int *g_ref_grp_visits = nullptr, *g_ref_grp_accepts = nullptr;
static unsigned long long *g_ref_grp_hash = nullptr;
static long long g_ref_prev_visits = 0, g_ref_prev_accepts = 0;

/* The two hooks below sit INSIDE the reference's hot loop (benchmark.cpp:242-276).  They exist only in the COUNTED
   build (oracle/_ref/libnbody_ref_counted.so, -DREF_COUNT_VOTES), which the vote-parity tests use; in
   libnbody_ref.so — the library every TIMING of the reference goes through (bench.py cpu_baseline*, tools/dropin_time.py)
   — they expand to nothing, so the walk is the reference's own machine code, unprobed (the per-group
   counters, 16 groups to a cache line under schedule(dynamic), made the Barnes-Hut step 3-7x slower). */
#ifdef REF_COUNT_VOTES
/* after node_stack.pop_back(), benchmark.cpp:245 */
#define REF_VISIT(start) do { if (g_ref_grp_visits) ++g_ref_grp_visits[(start) >> 3]; } while (0)
/* inside the accept branch, benchmark.cpp:267 */
#define REF_ACCEPT(start, node)                                                                        \
    do { if (g_ref_grp_accepts) ++g_ref_grp_accepts[(start) >> 3];                                      \
         if (g_ref_grp_hash) g_ref_grp_hash[(start) >> 3] =                                             \
             (g_ref_grp_hash[(start) >> 3] ^ (unsigned long long)(unsigned)(node)) * 1099511628211ull; } while (0)
#define REF_VOTES_COUNTED 1
#else
#define REF_VISIT(start) do { } while (0)
#define REF_ACCEPT(start, node) do { } while (0)
#define REF_VOTES_COUNTED 0
#endif
/* 1 when this library was built with the counters (the Python side refuses to ask an uncounted build for them) */
extern "C" int ref_votes_counted(void);
int ref_votes_counted(void) { return REF_VOTES_COUNTED; }

#define REF_CP_IN(dst, src)  do { if (src) std::memcpy((dst).data(), (src), sizeof(float) * (size_t)p->n); } while (0)
#define REF_CP_OUT(dst, src) do { if (dst) std::memcpy((dst), (src).data(), sizeof(float) * (size_t)p->n); } while (0)

/* after the generator loop, benchmark.cpp:339 */
#define REF_AFTER_INIT(data)                                   \
    do {                                                       \
        REF_CP_OUT(p->init_m, data.m);   REF_CP_OUT(p->init_px, data.posX); \
        REF_CP_OUT(p->init_py, data.posY); REF_CP_OUT(p->init_vx, data.velX); \
        REF_CP_OUT(p->init_vy, data.velY); REF_CP_OUT(p->init_radius, data.radius); \
        REF_CP_IN(data.m, p->in_m);     REF_CP_IN(data.posX, p->in_px);    \
        REF_CP_IN(data.posY, p->in_py); REF_CP_IN(data.velX, p->in_vx);    \
        REF_CP_IN(data.velY, p->in_vy); REF_CP_IN(data.radius, p->in_radius); \
        g_ref_grp_visits = p->grp_visits; g_ref_grp_accepts = p->grp_accepts; g_ref_grp_hash = p->grp_hash; \
        g_ref_prev_visits = g_ref_prev_accepts = 0;                           \
    } while (0)

/* after tree.build, benchmark.cpp:357 (QuadTree members made public by the recipe) */
#define REF_AFTER_BUILD(tree)                                                    \
    do {                                                                         \
        if (p->tree_nodes_used) *p->tree_nodes_used = tree.lastPoolIndex;         \
        if (p->tree_root) { p->tree_root[0] = tree.pool[0].mass;                  \
            p->tree_root[1] = tree.pool[0].comX; p->tree_root[2] = tree.pool[0].comY; } \
        if (p->tree_pool_out) {                                                   \
            int cnt = tree.lastPoolIndex < p->tree_pool_cap ? tree.lastPoolIndex : p->tree_pool_cap; \
            std::memcpy(p->tree_pool_out, tree.pool.data(), sizeof(tree.pool[0]) * (size_t)cnt); } \
    } while (0)

/* after computeForce_vec, benchmark.cpp:380 */
#define REF_BH_ACC(i, ax, ay)                                  \
    do { if (p->acc_x) { _mm256_storeu_ps(p->acc_x + (i), (ax)); _mm256_storeu_ps(p->acc_y + (i), (ay)); } } while (0)

/* after the naive force loop, benchmark.cpp:478 */
#define REF_NAIVE_ACC(ax, ay)  do { REF_CP_OUT(p->acc_x, ax); REF_CP_OUT(p->acc_y, ay); } while (0)

/* after physicsEnd, benchmark.cpp:513 */
#define REF_STEP_TIME(t0, t1)                                                     \
    do { if (p->step_ms) p->step_ms[ref_step] =                                   \
             std::chrono::duration<double, std::milli>((t1) - (t0)).count();      \
         ref_step_totals(p, ref_step); } while (0)

static inline void ref_step_totals(RefIO *p, int step)
{
    const int groups = p->n / 8;
    if (p->step_visits && p->grp_visits) {
        long long t = 0;
        for (int g = 0; g < groups; ++g) t += p->grp_visits[g];
        p->step_visits[step] = t - g_ref_prev_visits;
        g_ref_prev_visits = t;
    }
    if (p->step_accepts && p->grp_accepts) {
        long long t = 0;
        for (int g = 0; g < groups; ++g) t += p->grp_accepts[g];
        p->step_accepts[step] = t - g_ref_prev_accepts;
        g_ref_prev_accepts = t;
    }
}

/* after the loop, before benchmark.cpp:559 */
#define REF_AFTER_LOOP(data)                                   \
    do { REF_CP_OUT(p->out_px, data.posX); REF_CP_OUT(p->out_py, data.posY); \
         REF_CP_OUT(p->out_vx, data.velX); REF_CP_OUT(p->out_vy, data.velY); } while (0)
