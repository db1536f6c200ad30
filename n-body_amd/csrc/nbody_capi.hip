// nbody_capi.hip — the C-ABI of include/nbody.h on top of the gfx950 kernels.
//
// Host-side step driver: replaces the block between physicsStart and physicsEnd of the reference
// (benchmark.cpp:352-513).  One context = one GPU = one i-shard; everything is enqueued on one HIP
// stream, nothing in the step path allocates or synchronises.
#include "../../include/nbody.h"
#include "force_naive.hip.h"
#include "diagnostics.hip.h"
#include "bh_tree.hip.h"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

using namespace nbody;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(NBODY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

struct EventPair { hipEvent_t a, b; };

struct Profiler {
    bool on = false;
    std::vector<EventPair> pool[3];   // per kernel class
    size_t used[3] = {0, 0, 0};
    double total_ms[3] = {0, 0, 0};
    int64_t launches[3] = {0, 0, 0};
    std::vector<float> samples[3];    // duration of every launch since the last reset (nbody_profile_samples)
};

constexpr int kMaxSlices = 64;
constexpr int kBhComSparseTop = 8, kBhComSparseBottom = 24;   // per-depth centre-of-mass launches: depths with few cells (bh_com_kernel)
constexpr int kBhClimbMax = 1 << 30;         // bodies: up to here the centre-of-mass pass is ONE launch (bh_com_climb_kernel) — every size since the
                                             // climb moves 16-byte records (round 3: 191 vs 274 us at 4 x 10^6, 714 vs 864 at 2^24); the per-depth
                                             // launches stay as the alternative path (NBODY_BH_CLIMB_MAX=0)
constexpr float kBhSortedThetaScale = 0.6f;     // tools/bh_sorted_scale.py, profiles/r2_bh_sorted_scale.txt
constexpr int kBhQuadMaxGroups = 5120;      // four waves per group (bh_walk_quad_kernel) up to 40 960 targets at theta >= 1 ...
constexpr int kBhQuadMinGroups = 1792;      // ... 14 336 at theta <= 0.5, linear in between (measured crossovers; 49 152 and
                                            // 24 576 before bh_walk_kernel's rewrite at the end of round 3)
constexpr size_t kMaxEventsPerClass = 4096;
constexpr size_t kMaxSamplesPerClass = 1 << 16;   // nbody_profile_samples keeps the most recent 32 768 .. 65 536 launches

// a long-running host that leaves profiling on must not grow without bound: drop the older half when full
void push_sample(std::vector<float>& v, float ms)
{
    if (v.size() >= kMaxSamplesPerClass) v.erase(v.begin(), v.begin() + (std::ptrdiff_t)(kMaxSamplesPerClass / 2));
    v.push_back(ms);
}

}  // namespace

struct nbody_ctx {
    nbody_config cfg{};
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // SoA state in HBM: one float[n] per field, 256-B aligned by hipMalloc.
    float *m = nullptr, *px = nullptr, *py = nullptr, *vx = nullptr, *vy = nullptr, *radius = nullptr;
    float* c = nullptr;              // per-body clamp constant, see force_naive.hip.h
    float *ax = nullptr, *ay = nullptr;      // total accelerations [n]
    float *pax = nullptr, *pay = nullptr;    // j-slice partials [slices][i_stride]
    double* diag_partials = nullptr;
    int diag_blocks = 0;
    int i_begin = 0, i_end = 0;      // targets actually advanced (shard ∩ ref_compat limit)
    int i_stride = 0;
    bool bh_exact_core = false;      // masses / radii within exact_core.hip.h's ranges (nbody_upload)
    int variant = 0, slices = 1, k_per_lane = 2, j_per_slice = 0, blocks_x = 0, tile = 0, partial_rows = 0;
    float uni_c = 0, pre_scale = 1, out_scale = 1;
    bool uploaded = false;
    int64_t steps_enqueued = 0;      // since the last upload
    Profiler prof;
    // native exchange (comm_rccl.hip.h): an ncclComm_t, kept as void* so that this struct needs no RCCL header
    void* comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;
    bool comm_equal = false;                        // in-place ncclAllGather (else one ncclBroadcast per owner)
    std::vector<std::pair<int, int>> comm_spans;   // (begin, count) of every rank
    // position streaming (nbody_frame_*): device snapshots + pinned host frames, copied on their own stream
    struct Frames {
        static constexpr int kSlots = 2;
        float* dev[kSlots] = {nullptr, nullptr};      // [2n]: x then y
        float* host[kSlots] = {nullptr, nullptr};     // pinned, same layout
        hipEvent_t snapped[kSlots] = {nullptr, nullptr}, landed[kSlots] = {nullptr, nullptr};
        int64_t step[kSlots] = {0, 0};
        hipStream_t copy_stream = nullptr;
        int64_t published = 0, acquired = 0;
        bool ready = false;
    } frames;
    // one captured step (hipGraph): the Barnes-Hut step is ~150 short kernels, launch-bound when enqueued eagerly
    hipGraphExec_t step_graph = nullptr;
    hipStream_t step_graph_stream = nullptr;
    bool graph_failed = false;
    // Barnes-Hut state (allocated only when cfg.use_barnes_hut)
    struct Bh {
        uint64_t *keys = nullptr, *keys_sorted = nullptr, *gkeys = nullptr, *gkeys_sorted = nullptr, *keys_merged = nullptr;
        int *vals = nullptr, *vals_sorted = nullptr, *gvals = nullptr, *gvals_sorted = nullptr, *vals_merged = nullptr;
        int *cnt = nullptr, *base = nullptr;
        int* owner = nullptr;            // per internal cell: the sorted instance that owns it
        uint16_t* tie = nullptr;
        unsigned *gfx = nullptr, *gfy = nullptr;   // second instances: forced-up levels, expanded leaf depth
        int *gld = nullptr, *pld = nullptr, *glo = nullptr, *ghi = nullptr, *gpos = nullptr, *tie_table = nullptr;
        int tie_table_size = 0;         // per body: first levels at which x / y sit on a split line
        int gcap = 0;                    // capacity for second instances of bodies on split lines
        void* gsort_tmp = nullptr;
        size_t gsort_bytes = 0;
        BhTree tree{};
        BhCounters* ctr = nullptr;
        int *overflow_host = nullptr, *overflow_dev = nullptr;   // pinned, device-mapped word: the last build's overflow flag
        int *grp_visits = nullptr, *grp_accepts = nullptr;   // per group of 8 bodies, written by every walk
        // bh_sorted on a partial shard: the shard's own bodies in key order (a second, shard-sized sort)
        uint64_t* own_keys_sorted = nullptr;
        int* own_order = nullptr;
        void* own_sort_tmp = nullptr;
        size_t own_sort_bytes = 0;
        int *rank = nullptr, *rank_done = nullptr;   // small-N rank sort: partial ranks [n], per-tile arrival counters,
        float4* body4 = nullptr;                     // {x, y, m, 0} per body, written by bh_keys_kernel for the node kernel's gathers
        unsigned char* tie_first_sorted = nullptr;   //   first split-line level of each body, in sorted order [n]
        void *sort_tmp = nullptr, *scan_tmp = nullptr;
        size_t sort_bytes = 0, scan_bytes = 0;
        bool hipcub_ready = false;                  // rocPRIM scratch allocated (context created with NBODY_BH_HIPCUB set)
        // this repository's sort and scan (bh_sort.hip.h); the rocPRIM ones above stay as the alternative path (NBODY_BH_HIPCUB)
        SortBuffers sort_main, sort_own;
        ScanState* scan_state = nullptr;
        unsigned long long* scan_look = nullptr;
        int64_t node_cap = 0;
    } bh;
};

void destroy_comm(nbody_ctx* ctx);   // comm_rccl.hip.h
namespace { void drop_step_graph(nbody_ctx* ctx); }

namespace {

int profile_begin(nbody_ctx* ctx, int cls, EventPair** out)
{
    *out = nullptr;
    Profiler& p = ctx->prof;
    if (!p.on) return NBODY_OK;
    if (p.used[cls] >= kMaxEventsPerClass) {   // drain before reusing the pool
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < p.used[cls]; ++i) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, p.pool[cls][i].a, p.pool[cls][i].b));
            p.total_ms[cls] += ms;
            push_sample(p.samples[cls], ms);
        }
        p.used[cls] = 0;
    }
    if (p.used[cls] == p.pool[cls].size()) {
        EventPair ep;
        HIP_TRY(hipEventCreate(&ep.a));
        HIP_TRY(hipEventCreate(&ep.b));
        p.pool[cls].push_back(ep);
    }
    EventPair* ep = &p.pool[cls][p.used[cls]++];
    HIP_TRY(hipEventRecord(ep->a, ctx->stream));
    *out = ep;
    return NBODY_OK;
}

int profile_end(nbody_ctx* ctx, int cls, EventPair* ep)
{
    if (!ep) return NBODY_OK;
    HIP_TRY(hipEventRecord(ep->b, ctx->stream));
    ctx->prof.launches[cls]++;
    return NBODY_OK;
}

template <int V>
void launch_force_k(int K, int tile, const ForceArgs& a, dim3 grid, hipStream_t s)
{
    constexpr int JU = 4;
    constexpr int kBig = Tile<V>::kTileBig;
    if (tile == kTileSmall) {
        if (K == 4) force_naive_kernel<V, 4, JU, kTileSmall, 4, true><<<grid, dim3(kBlock), 0, s>>>(a);
        else        force_naive_kernel<V, 2, JU, kTileSmall, 4, true><<<grid, dim3(kBlock), 0, s>>>(a);
    } else {
        if (K == 4) force_naive_kernel<V, 4, JU, kBig><<<grid, dim3(kBlock), 0, s>>>(a);
        else        force_naive_kernel<V, 2, JU, kBig><<<grid, dim3(kBlock), 0, s>>>(a);
    }
}

int big_tile_of(int variant) { return (variant == kScaled || variant == kUniform) ? Tile<kScaled>::kTileBig : Tile<kPerBody>::kTileBig; }

int enqueue_force(nbody_ctx* ctx)
{
    if (ctx->i_end <= ctx->i_begin) return NBODY_OK;
    ForceArgs a{};
    a.px = ctx->px; a.py = ctx->py; a.m = ctx->m; a.c = ctx->c;
    a.pax = ctx->pax; a.pay = ctx->pay;
    a.n = ctx->cfg.num_bodies;
    a.i_begin = ctx->i_begin; a.i_end = ctx->i_end; a.i_stride = ctx->i_stride;
    a.j_per_slice = ctx->j_per_slice;
    a.pre_scale = ctx->pre_scale; a.uni_c = ctx->uni_c; a.out_scale = ctx->out_scale;
    dim3 grid(ctx->blocks_x, ctx->slices);
    EventPair* ep;
    int rc = profile_begin(ctx, 0, &ep);
    if (rc) return rc;
    const int K = ctx->k_per_lane;
    switch (ctx->variant) {
        case kScaled:  launch_force_k<kScaled>(K, ctx->tile, a, grid, ctx->stream); break;
        case kUniform: launch_force_k<kUniform>(K, ctx->tile, a, grid, ctx->stream); break;
        case kPerBody: launch_force_k<kPerBody>(K, ctx->tile, a, grid, ctx->stream); break;
        case kExact:   launch_force_k<kExact>(K, ctx->tile, a, grid, ctx->stream); break;
        default: return fail(NBODY_ERR_INVALID, "no force variant selected");
    }
    HIP_TRY(hipGetLastError());
    return profile_end(ctx, 0, ep);
}

int enqueue_integrate(nbody_ctx* ctx, int advance)
{
    // ref_compat 2 (main.cpp:535-544): the context that owns the end of the array lets the N%8 tail drift
    const int n = ctx->cfg.num_bodies, tail = n % 8;
    const int shard_end = ctx->cfg.shard_begin + (ctx->cfg.shard_count == 0 ? n - ctx->cfg.shard_begin : ctx->cfg.shard_count);
    if (advance && tail && ctx->cfg.ref_compat == 2 && ctx->cfg.use_barnes_hut && shard_end == n) {
        tail_drift_kernel<<<1, 8, 0, ctx->stream>>>(ctx->px, ctx->py, ctx->vx, ctx->vy, n - tail, n, ctx->cfg.dt);
        HIP_TRY(hipGetLastError());
    }
    if (ctx->cfg.use_barnes_hut) return NBODY_OK;     // the walk kernel has stored the accelerations and integrated
    if (ctx->i_end <= ctx->i_begin) return NBODY_OK;
    IntegrateArgs a{};
    a.px = ctx->px; a.py = ctx->py; a.vx = ctx->vx; a.vy = ctx->vy;
    a.pax = ctx->pax; a.pay = ctx->pay; a.ax = ctx->ax; a.ay = ctx->ay;
    a.i_begin = ctx->i_begin; a.i_end = ctx->i_end; a.i_stride = ctx->i_stride;
    a.slices = ctx->cfg.use_barnes_hut ? 1 : ctx->slices;
    a.dt = ctx->cfg.dt; a.advance = advance;
    const int nt = ctx->i_end - ctx->i_begin;
    EventPair* ep;
    int rc = profile_begin(ctx, 1, &ep);
    if (rc) return rc;
    // every array is its own hipMalloc and i_stride a multiple of 64: a quad is 16-byte aligned iff i_begin is
    if (a.i_begin % 4 == 0 && a.i_stride % 4 == 0 && !std::getenv("NBODY_INTEGRATE_SCALAR"))   // (developer/test switch)
        integrate4_kernel<<<dim3(((nt + 3) / 4 + 255) / 256), dim3(256), 0, ctx->stream>>>(a);
    else
        integrate_kernel<<<dim3((nt + 255) / 256), dim3(256), 0, ctx->stream>>>(a);
    HIP_TRY(hipGetLastError());
    return profile_end(ctx, 1, ep);
}


// Barnes-Hut step = build (keys -> sort -> cells -> nodes -> centre of mass) + walk, benchmark.cpp:354-403.
int enqueue_bh_build(nbody_ctx* ctx)
{
    auto& b = ctx->bh;
    const int n = ctx->cfg.num_bodies;
    const dim3 blk(kBhBlock), grid_n((n + kBhBlock - 1) / kBhBlock);
    hipStream_t s = ctx->stream;
    EventPair* ep;
    int rc = profile_begin(ctx, 2, &ep);
    if (rc) return rc;
    const int nm = n + b.gcap;                                 // instances after the merge (tail = padding)
    const dim3 grid_m((nm + kBhBlock - 1) / kBhBlock);
    const bool adaptive = ctx->cfg.adaptive_box != 0;
    if (adaptive) {
        bh_reset_kernel<<<1, 64, 0, s>>>(b.ctr, 1e-30f);
        bh_bounds_kernel<<<dim3(grid_n.x < 1024u ? grid_n.x : 1024u), blk, 0, s>>>(ctx->px, ctx->py, n, b.ctr);
    }
    const bool fused = n <= kBhSmallMax && b.gcap <= kBhSmallGhosts && !std::getenv("NBODY_BH_NO_FUSE");
    bh_keys_kernel<<<grid_n, blk, 0, s>>>(ctx->px, ctx->py, n, b.ctr, ctx->cfg.spread, adaptive ? 0 : 1, b.keys, b.vals, b.tie,
                                          b.gkeys, b.gvals, b.gcap, b.tie_table, b.tie_table_size, fused ? 1 : 0, ctx->m, b.body4);
    const bool use_hipcub = b.hipcub_ready && std::getenv("NBODY_BH_HIPCUB") != nullptr;    // developer/test switch: rocPRIM's sort and scan
    const char* rank_env = std::getenv("NBODY_BH_RANKSORT_MAX");         // developer/test switch
    // (the rank buffers exist only up to kBhRankSortMax bodies, nbody_create: the knob can lower the limit, never raise it)
    const int rank_max = rank_env ? (std::atoi(rank_env) < kBhRankSortMax ? std::atoi(rank_env) : kBhRankSortMax) : kBhRankSortMax;
    const bool ranked = fused && b.rank != nullptr && n <= rank_max;
    bool valid_counted = false;
    if (ranked) {
        // small N: rank by brute-force counting, one launch.  ~1024 workgroups: i-tiles of 256 x j-slices
        const int tiles = (int)grid_n.x;
        int slices = 1024 / tiles < 1 ? 1 : (1024 / tiles > 64 ? 64 : 1024 / tiles);
        const int per = ((n + slices - 1) / slices + kBhRankChunk - 1) / kBhRankChunk * kBhRankChunk;
        slices = (n + per - 1) / per;
        bh_ranksort_kernel<<<dim3(grid_n.x, slices), blk, 0, s>>>(b.keys, b.vals, n, per, b.rank, b.rank_done, b.keys_sorted, b.vals_sorted,
                                                                  b.tie, b.tie_first_sorted);
    } else if (use_hipcub) {
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b.sort_tmp, b.sort_bytes, b.keys, b.keys_sorted, b.vals, b.vals_sorted, n,
                                                   0, 64, s));
    } else {
        // TMP = the merged arrays, written later.  The sort's run kernel also leaves the number of in-box bodies (keys with
        // bit 63 clear) in ctr->valid; bh_keys_kernel has reset it to 0, which stands when there are none.
        valid_counted = sort_pairs(b.sort_main, b.keys, b.vals, b.keys_sorted, b.vals_sorted, n, s, true, fused ? nullptr : &b.ctr->valid);
    }
    // bodies exactly on a split line get a second instance (inclusive bounds, benchmark.cpp:83-85)
    BhTies ties{b.gkeys, b.gvals, b.gfx, b.gfy, b.gld, b.glo, b.ghi, b.gpos, b.pld, b.tie_table, b.tie_table_size - 1, b.gcap};
    if (fused) {
        // small N: valid count, split-line instances, their sort, merge, cell count, scan and owners in one launch
        bh_prep_small_kernel<<<1, kBhFixBlock, 0, s>>>(b.keys_sorted, b.vals_sorted, ctx->px, ctx->py, b.tie, b.ctr, ties, n,
                                                       b.gkeys_sorted, b.gvals_sorted, b.keys_merged, b.vals_merged, b.cnt, b.base,
                                                       b.owner, b.tree.cap_internal, ranked ? b.tie_first_sorted : nullptr);
    } else {
        if (!valid_counted) bh_valid_kernel<<<1, 64, 0, s>>>(b.keys_sorted, n, b.ctr);
        bh_ties_kernel<<<grid_n, blk, 0, s>>>(b.keys_sorted, b.vals_sorted, ctx->px, ctx->py, b.tie, b.ctr, ties);
        if (use_hipcub) {
            bh_tie_fixpoint_kernel<<<1, kBhFixBlock, 0, s>>>(b.keys_sorted, b.vals_sorted, ctx->px, ctx->py, b.tie, b.ctr, ties);
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b.gsort_tmp, b.gsort_bytes, b.gkeys, b.gkeys_sorted, b.gvals, b.gvals_sorted,
                                                       b.gcap, 0, 64, s));
        } else {
            // the fixed point and the sort of the few second instances in one launch (in LDS, 8192 at a time)
            bh_tie_fixpoint_sort_kernel<<<1, kBhFixBlock, 0, s>>>(b.keys_sorted, b.vals_sorted, ctx->px, ctx->py, b.tie, b.ctr, ties,
                                                                  b.gkeys_sorted, b.gvals_sorted);
        }
        bh_merge_kernel<<<dim3((nm + kBhBlock * kBhMergePer - 1) / (kBhBlock * kBhMergePer)), blk, 0, s>>>(b.keys_sorted, b.vals_sorted, n, b.gkeys_sorted, b.gvals_sorted, b.gcap,
                                               b.keys_merged, b.vals_merged, b.ctr);
        if (use_hipcub) {
            bh_count_kernel<<<grid_m, blk, 0, s>>>(b.keys_merged, nm, b.ctr, b.cnt);
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(b.scan_tmp, b.scan_bytes, b.cnt, b.base, nm, s));
            bh_owner_kernel<<<grid_m, blk, 0, s>>>(b.cnt, b.base, nm, b.ctr, b.owner, b.tree.cap_internal);
        } else {
            if (nm >= kBhScanLargeFrom) {
                constexpr int kScanTile = kBhBlock * kBhScanRoundsLarge;
                bh_cells_scan_kernel<kBhScanRoundsLarge><<<dim3((nm + kScanTile - 1) / kScanTile), blk, 0, s>>>(
                    b.keys_merged, nm, b.ctr, b.scan_state, b.scan_look, b.cnt, b.base, b.owner, b.tree.cap_internal);
            } else {
                constexpr int kScanTile = kBhBlock * kBhScanRounds;
                bh_cells_scan_kernel<kBhScanRounds><<<dim3((nm + kScanTile - 1) / kScanTile), blk, 0, s>>>(
                    b.keys_merged, nm, b.ctr, b.scan_state, b.scan_look, b.cnt, b.base, b.owner, b.tree.cap_internal);
            }
        }
    }
    BhWaitErrors werr{{nullptr, nullptr, nullptr}};
    if (!use_hipcub && !ranked) werr.p[0] = &b.sort_main.state->error;
    if (!use_hipcub && !fused) werr.p[1] = &b.scan_state->error;
    const dim3 grid_c((b.tree.cap_internal + kBhBlock - 1) / kBhBlock);
    bh_nodes_kernel<<<grid_c, blk, 0, s>>>(b.keys_merged, b.vals_merged, b.cnt, b.base, b.owner, nm, ctx->m, ctx->px, ctx->py,
                                           b.tree, b.ctr, b.overflow_dev, fused ? b.keys_sorted : nullptr,
                                           fused ? b.vals_sorted : nullptr, b.body4, werr);
    // centre of mass: one launch in which the last child to finish evaluates its parent (the per-depth launches of rounds
    // 1-2 — 31 launches — are the alternative path)
    const char* climb_env = std::getenv("NBODY_BH_CLIMB_MAX");          // developer/test switch, read per enqueue
    const int climb_max = climb_env ? std::atoi(climb_env) : kBhClimbMax;
    if (n <= climb_max) {
        bh_com_climb_kernel<<<grid_c, blk, 0, s>>>(b.tree, b.ctr, std::getenv("NBODY_BH_CLIMB_SCATTER") ? 1 : 0);
    } else {
        const dim3 grid1((unsigned)((b.tree.cap_internal + 4 * kBhBlock - 1) / (4 * kBhBlock)));
        const dim3 grid4((unsigned)((b.tree.cap_internal + 16 * kBhBlock - 1) / (16 * kBhBlock)));
        for (int d = kBhLevels - 1; d >= 0; --d) {
            if (d <= kBhComSparseTop || d >= kBhComSparseBottom) bh_com_kernel<4><<<grid4, blk, 0, s>>>(b.tree, b.ctr, d);
            else                                                  bh_com_kernel<1><<<grid1, blk, 0, s>>>(b.tree, b.ctr, d);
        }
    }
    HIP_TRY(hipGetLastError());
    return profile_end(ctx, 2, ep);
}

// walk + (advance) the Euler update of the walked bodies, fused as in the reference (benchmark.cpp:365-401)
int enqueue_bh_walk(nbody_ctx* ctx, int advance)
{
    if (ctx->i_end <= ctx->i_begin) return NBODY_OK;
    BhWalkArgs w{};
    w.A = ctx->bh.tree.A; w.B = ctx->bh.tree.B;
    w.px = ctx->px; w.py = ctx->py; w.radius = ctx->radius;
    w.vx = ctx->vx; w.vy = ctx->vy;
    w.ax = ctx->ax; w.ay = ctx->ay; w.ctr = ctx->bh.ctr;
    w.grp_visits = ctx->bh.grp_visits; w.grp_accepts = ctx->bh.grp_accepts;
    if (ctx->cfg.bh_sorted) {
        // walk order = the build's sorted order.  Whole array: the sorted body list itself.  A shard: its own bodies
        // sorted by the same keys (still compact patches: a shard's bodies are spread evenly over the box)
        auto& b = ctx->bh;
        if (b.own_order) {
            const int sb = ctx->cfg.shard_begin, sc = ctx->i_end - ctx->i_begin;
            if (b.hipcub_ready && std::getenv("NBODY_BH_HIPCUB"))
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b.own_sort_tmp, b.own_sort_bytes, b.keys + sb, b.own_keys_sorted, b.vals + sb,
                                                           b.own_order, sc, 0, 64, ctx->stream));
            else
                sort_pairs(b.sort_own, b.keys + sb, b.vals + sb, b.own_keys_sorted, b.own_order, sc, ctx->stream, true);
            w.order = b.own_order - sb;
        } else {
            w.order = b.vals_sorted;
        }
    }
    w.dt = ctx->cfg.dt; w.advance = advance;
    w.exact_core = ctx->bh_exact_core ? 1 : 0;
    w.i_begin = ctx->i_begin; w.i_end = ctx->i_end;
    // bh_sorted: compact groups accept a node as soon as ONE patch is far from it, where the reference's scattered
    // groups need eight unrelated places to be; at the same threshold that is a larger error (measured 2x).  The
    // threshold is therefore scaled by 0.6, at which the sorted walk's mean, 90th-percentile and maximum error against
    // the direct sum are at or below the reference-order walk's at the nominal theta and the median within 6 % of it
    // (tests/test_gpu_bh_sorted.py) — with 1.6-2.0x fewer node visits.
    float sorted_scale = kBhSortedThetaScale;
    if (const char* e = std::getenv("NBODY_BH_SORTED_SCALE")) sorted_scale = (float)std::atof(e);     // developer knob
    w.theta = ctx->cfg.bh_sorted ? ctx->cfg.theta * sorted_scale : ctx->cfg.theta;
    const int nt = ctx->i_end - ctx->i_begin;
    EventPair* ep;
    int rc = profile_begin(ctx, 0, &ep);
    if (rc) return rc;
    // Few groups: a workgroup of four waves per group (bh_walk_quad_kernel); many: 8 groups share a wave (bh_walk_kernel).
    // Where the second overtakes depends on the work per group, i.e. on theta.  Measured walk us, four-wave /
    // 8-per-wave (tools/walk_crossover_job.sh, end of round 3): theta = 1: N = 16 384 101 / 195, 24 576 139 / 208, 32 768 188 / 214,
    // 49 152 296 / 239, 65 536 375 / 242; theta = 0.5: 12 288 480 / 514, 16 384 560 / 534, 24 576 849 / 572, 32 768 1148 / 603.
    const int groups = (nt + 7) / 8;
    // (the nominal theta also for bh_sorted: its 0.6 theta vote on compact groups costs about as many visits per group
    // as the nominal theta on scattered ones, and the measured crossovers are the same)
    const float tn = ctx->cfg.theta;
    const float tq = tn < 0.5f ? 0.5f : (tn > 1.0f ? 1.0f : tn);
    int quad_max = kBhQuadMinGroups + (int)((tq - 0.5f) * 2.0f * (float)(kBhQuadMaxGroups - kBhQuadMinGroups));
    if (const char* e = std::getenv("NBODY_BH_QUAD_MAX")) quad_max = std::atoi(e);              // developer/test switch
    // benchmark.cpp:197-198 compares body indices as floats: beyond 2^24 that is not the integer comparison any more
    const bool float_self = ctx->cfg.ref_compat != 0 && ctx->cfg.num_bodies > (1 << 24);
    if (groups <= quad_max) {
        w.lanes_used = 8;
        const dim3 grid(groups), blk(256);
        if (float_self) {
            if (ctx->cfg.bh_fast_math) bh_walk_quad_kernel<true, true><<<grid, blk, 0, ctx->stream>>>(w);
            else                       bh_walk_quad_kernel<false, true><<<grid, blk, 0, ctx->stream>>>(w);
        } else {
            if (ctx->cfg.bh_fast_math) bh_walk_quad_kernel<true, false><<<grid, blk, 0, ctx->stream>>>(w);
            else                       bh_walk_quad_kernel<false, false><<<grid, blk, 0, ctx->stream>>>(w);
        }
    } else {
        w.lanes_used = 8 * bh_walk_groups_per_wave(nt);
        if (const char* e = std::getenv("NBODY_BH_WALK_GROUPS")) {           // developer switch: 1, 2, 4 or 8 groups per wave
            const int g = std::atoi(e);
            if (g == 1 || g == 2 || g == 4 || g == 8) w.lanes_used = 8 * g;
        }
        const int per_block = (kBhBlock / 64) * w.lanes_used;      // targets per workgroup
        const dim3 grid((nt + per_block - 1) / per_block), blk(kBhBlock);
        // 32-bit byte offsets into the node records while 16 B x pool capacity fits (see bh_walk_kernel).  That ends
        // beyond 2^24 bodies, where the float self-mask is on: three of the four combinations exist.
        const bool offs32 = (ctx->bh.node_cap + 1) * 16 < (1ll << 32) && !std::getenv("NBODY_BH_WALK_OFFS64");
        w.done_node = (int)ctx->bh.node_cap;               // (<= 2^31 - 255: nbody_create bounds num_bodies by 2^28 - 64)
        const bool fast = ctx->cfg.bh_fast_math != 0;
        if (offs32 && !float_self) {
            if (fast) bh_walk_kernel<true, false, true><<<grid, blk, 0, ctx->stream>>>(w);
            else      bh_walk_kernel<false, false, true><<<grid, blk, 0, ctx->stream>>>(w);
        } else if (offs32) {
            if (fast) bh_walk_kernel<true, true, true><<<grid, blk, 0, ctx->stream>>>(w);
            else      bh_walk_kernel<false, true, true><<<grid, blk, 0, ctx->stream>>>(w);
        } else {         // (the float comparison is the integer one below 2^24 bodies: the developer switch lands here too)
            if (fast) bh_walk_kernel<true, true, false><<<grid, blk, 0, ctx->stream>>>(w);
            else      bh_walk_kernel<false, true, false><<<grid, blk, 0, ctx->stream>>>(w);
        }
    }
    HIP_TRY(hipGetLastError());
    return profile_end(ctx, 0, ep);
}

// Geometry of the force launch: enough workgroups to fill 256 CUs several times over, so that the
// dispatcher can even out the load (one workgroup = 4 waves; 256 CUs x 4 SIMDs want >= 2 waves each).
struct ForcePlan { int k, blocks_x, tile, j_per_slice, slices; };

// n sources, nt targets.  k_req / slices_req / tile_req: 0 = derive (the developer knobs below set them).
ForcePlan make_force_plan(int n, int nt, int variant, int max_slices, int k_req = 0, int slices_req = 0, int tile_req = 0)
{
    ForcePlan p{};
    const int big = big_tile_of(variant ? variant : (int)kScaled);
    p.k = k_req ? k_req : (nt >= (1 << 15) ? 4 : 2);
    p.blocks_x = (nt + kBlock * p.k - 1) / (kBlock * p.k);
    if (p.blocks_x < 1) p.blocks_x = 1;
    int want = slices_req ? slices_req : (8192 + p.blocks_x - 1) / p.blocks_x;   // aim for >= 8192 workgroups (measured: tools/sweep.py)
    if (want > max_slices) want = max_slices;
    // tile: the big one unless that leaves fewer than HALF the slices wanted (N = 65 536: 32 slices of one 2048-source
    // tile run 0.8 % faster than 64 slices of four 256-source tiles and halve the partial rows, profiles/r3_force_plan_sweep.txt)
    p.tile = tile_req ? tile_req : (((n + want - 1) / want >= big / 2) ? big : kTileSmall);
    // granule of a slice: whole big tiles, or (small tile: swept only as far as the slice goes) 8 sources
    const int gran = p.tile == kTileSmall ? 8 : p.tile;
    const int min_per = p.tile == kTileSmall ? 64 : p.tile;
    const int max_slices_by_n = (n + min_per - 1) / min_per;
    if (want > max_slices_by_n) want = max_slices_by_n;
    if (want < 1) want = 1;
    if (p.tile == kTileSmall && !slices_req && (long long)p.blocks_x * want < 4096) {
        // Few workgroups (N = 10 000: 20 x 40 = 800 on 256 CUs): the launch ends with the CU that got the most of them, so
        // what counts is rounds = ceil(workgroups / 256) x the sources of a slice, not their product — 800 workgroups are
        // 4 rounds of 256 sources, 500 are 2 rounds of 400.  Model fitted to tools/naive_10k_slices.py
        // (profiles/r3_naive_small_sizes.txt): cost = rounds x per-slice sources (/ 0.9 when one round leaves a single
        // wave per SIMD) + 2 source-equivalents per slice (partial rows the Euler kernel sums, barriers).
        double best = 1e30;
        int best_s = want;
        for (int sl = 1; sl <= want; ++sl) {
            const int per = (((n + sl - 1) / sl) + gran - 1) / gran * gran;
            if (per < min_per) break;
            const int nsl = (n + per - 1) / per;
            const int rounds = (int)(((long long)p.blocks_x * nsl + 255) / 256);
            const double cost = rounds * per / (rounds == 1 ? 0.9 : 1.0) + 2.0 * nsl;
            if (cost < best) { best = cost; best_s = sl; }
        }
        want = best_s;
    }
    p.j_per_slice = ((n + want - 1) / want + gran - 1) / gran * gran;
    p.slices = (n + p.j_per_slice - 1) / p.j_per_slice;
    return p;
}

// the plan a context of this shape gets, developer knobs included (shared with nbody_accel_noalloc, whose results are
// promised to be the context's bit for bit: same slices, same summation order)
ForcePlan make_force_plan_env(int n, int nt, int variant)
{
    // developer tuning knobs (not part of the ABI): NBODY_FORCE_K = 2|4, NBODY_FORCE_SLICES = 1..64,
    // NBODY_FORCE_TILE = small|big
    int k = 0, want = 0, tile = 0;
    if (const char* e = std::getenv("NBODY_FORCE_K")) {
        const int v = std::atoi(e);
        if (v == 2 || v == 4) k = v;
    }
    if (const char* e = std::getenv("NBODY_FORCE_SLICES")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= kMaxSlices) want = v;
    }
    if (const char* e = std::getenv("NBODY_FORCE_TILE")) tile = (e[0] == 's') ? kTileSmall : big_tile_of(variant);
    return make_force_plan(n, nt, variant, kMaxSlices, k, want, tile);
}

void plan_force(nbody_ctx* ctx)
{
    const int n = ctx->cfg.num_bodies;
    const int nt = ctx->i_end - ctx->i_begin;
    const int variant = ctx->variant ? ctx->variant : (int)kScaled;
    const ForcePlan p = make_force_plan_env(n, nt, variant);
    ctx->k_per_lane = p.k;
    ctx->blocks_x = p.blocks_x;
    ctx->tile = p.tile;
    ctx->j_per_slice = p.j_per_slice;
    ctx->slices = p.slices;
}

}  // namespace

extern "C" {

void nbody_default_config(nbody_config* cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->num_bodies = 10000;      // benchmark.cpp:302
    cfg->spread = 400.0f;         // :303
    cfg->mass = 1000.0f;          // :304
    cfg->dt = 0.05f;              // :306
    cfg->theta = 1.0f;            // :305
    cfg->use_barnes_hut = 1;      // :300
    cfg->use_parallel = 1;        // :301
    cfg->ref_compat = 1;
    cfg->device = -1;
}

int nbody_shard_range(int32_t n, int32_t nranks, int32_t rank, int32_t* begin, int32_t* count)
{
    if (n < 0 || nranks < 1 || rank < 0 || rank >= nranks || !begin || !count)
        return fail(NBODY_ERR_INVALID, "nbody_shard_range: bad arguments (n=%d nranks=%d rank=%d)", n, nranks, rank);
    // groups of 8 bodies (benchmark.cpp:421) are dealt out as evenly as possible; the N%8 tail
    // (never a target in the reference) rides with the last rank.
    const int64_t groups = n / 8;
    const int64_t g0 = groups * rank / nranks, g1 = groups * (rank + 1) / nranks;
    int64_t b = g0 * 8, e = g1 * 8;
    if (rank == nranks - 1) e = n;
    *begin = (int32_t)b;
    *count = (int32_t)(e - b);
    return NBODY_OK;
}

int nbody_create(const nbody_config* cfg, nbody_ctx** out)
{
    if (!cfg || !out) return fail(NBODY_ERR_INVALID, "nbody_create: null argument");
    *out = nullptr;
    const int n = cfg->num_bodies;
    if (n < 1) return fail(NBODY_ERR_INVALID, "nbody_create: num_bodies=%d", n);
    if (!(cfg->dt == cfg->dt)) return fail(NBODY_ERR_INVALID, "nbody_create: dt is NaN");
    if (cfg->shard_begin < 0 || cfg->shard_count < 0 || (int64_t)cfg->shard_begin + cfg->shard_count > n)
        return fail(NBODY_ERR_INVALID, "nbody_create: shard [%d,+%d) outside [0,%d)", cfg->shard_begin, cfg->shard_count, n);
    if (cfg->force_variant < 0 || cfg->force_variant > 4)
        return fail(NBODY_ERR_INVALID, "nbody_create: force_variant=%d", cfg->force_variant);
    if (cfg->use_barnes_hut && !cfg->adaptive_box && !(cfg->spread > 0.0f))
        return fail(NBODY_ERR_INVALID, "nbody_create: Barnes-Hut needs spread > 0 (root box), got %g", cfg->spread);
    // Barnes-Hut indices are 32-bit: node index <= 4 * (2 n + 64) must fit in int32 (the pool still fits 288 GB of HBM
    // around n = 2^28, so without this bound an overflow would be an out-of-bounds write, not an allocation failure)
    constexpr int kBhMaxBodies = (1 << 28) - 64;
    if (cfg->use_barnes_hut && n > kBhMaxBodies)
        return fail(NBODY_ERR_INVALID, "nbody_create: Barnes-Hut supports at most %d bodies (32-bit node indices), got %d", kBhMaxBodies, n);
    if (cfg->bh_sorted && (!cfg->use_barnes_hut || cfg->ref_compat != 0))
        return fail(NBODY_ERR_INVALID, "nbody_create: bh_sorted needs use_barnes_hut = 1 and ref_compat = 0 (it is not the reference's grouping)");
    if (cfg->use_barnes_hut && cfg->shard_begin % 8 != 0)
        return fail(NBODY_ERR_INVALID, "nbody_create: Barnes-Hut shards must start on a multiple of 8 (groups of 8 bodies)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(NBODY_ERR_NO_DEVICE, "nbody_create: no HIP device visible (there is no CPU fallback)");
    int dev = cfg->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(NBODY_ERR_NO_DEVICE, "nbody_create: device %d of %d", dev, ndev);
    HIP_TRY(hipSetDevice(dev));

    nbody_ctx* ctx = new (std::nothrow) nbody_ctx();
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_create: out of host memory");
    ctx->cfg = *cfg;
    ctx->device = dev;
    const int sb = cfg->shard_begin;
    const int sc = cfg->shard_count == 0 ? n - sb : cfg->shard_count;
    const int lim = cfg->ref_compat ? n - (n % 8) : n;   // benchmark.cpp:423
    ctx->i_begin = sb;
    ctx->i_end = (sb + sc < lim) ? sb + sc : lim;
    if (ctx->i_end < ctx->i_begin) ctx->i_end = ctx->i_begin;
    ctx->i_stride = ((sc + 63) / 64) * 64;
    if (ctx->i_stride < 64) ctx->i_stride = 64;
    auto cleanup = [&](int code) { nbody_destroy(ctx); return code; };
#define HIP_TRY_C(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return cleanup(fail(NBODY_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)));            \
    } while (0)
    HIP_TRY_C(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    const size_t nb = sizeof(float) * (size_t)n;
    float** arrs[] = {&ctx->m, &ctx->px, &ctx->py, &ctx->vx, &ctx->vy, &ctx->radius, &ctx->c, &ctx->ax, &ctx->ay};
    for (float** p : arrs) HIP_TRY_C(hipMalloc((void**)p, nb));
    // the j-slice partial rows (pax/pay) are allocated at upload, once the launch plan is known
    ctx->diag_blocks = (n + kDiagBlock - 1) / kDiagBlock;
    HIP_TRY_C(hipMalloc((void**)&ctx->diag_partials, sizeof(double) * 5 * (size_t)ctx->diag_blocks));
    HIP_TRY_C(hipMemset(ctx->ax, 0, nb));
    HIP_TRY_C(hipMemset(ctx->ay, 0, nb));
    if (cfg->use_barnes_hut) {
        auto& b = ctx->bh;
        // the reference reserves 32 N nodes (benchmark.cpp:148) and uses ~2.9 N;
        // here: 2 N + 64 internal cells => 8 N + 257 nodes of 24 B
        b.tree.cap_internal = 2 * n + 64;
        b.tree.com_reciprocal = cfg->ref_compat == 2;   // main.cpp's tree arithmetic (main.cpp:173-175)
        b.node_cap = 1 + 4 * (int64_t)b.tree.cap_internal;
        HIP_TRY_C(hipMalloc((void**)&b.keys, sizeof(uint64_t) * (size_t)n));
        HIP_TRY_C(hipMalloc((void**)&b.keys_sorted, sizeof(uint64_t) * (size_t)n));
        HIP_TRY_C(hipMalloc((void**)&b.vals, sizeof(int) * (size_t)n));
        HIP_TRY_C(hipMalloc((void**)&b.tie, sizeof(uint16_t) * (size_t)n));
        HIP_TRY_C(hipMalloc((void**)&b.vals_sorted, sizeof(int) * (size_t)n));
        b.gcap = n / 256 > 4096 ? n / 256 : 4096;
        const size_t nm = (size_t)n + (size_t)b.gcap;
        HIP_TRY_C(hipMalloc((void**)&b.gkeys, sizeof(uint64_t) * (size_t)b.gcap));
        size_t gpow = 2;                                        // the sorted tie arrays: next power of two (bh_tie_fixpoint_sort_kernel)
        while (gpow < (size_t)b.gcap) gpow <<= 1;
        HIP_TRY_C(hipMalloc((void**)&b.gkeys_sorted, sizeof(uint64_t) * gpow));
        HIP_TRY_C(hipMalloc((void**)&b.gvals, sizeof(int) * (size_t)b.gcap));
        HIP_TRY_C(hipMalloc((void**)&b.gvals_sorted, sizeof(int) * gpow));
        HIP_TRY_C(hipMalloc((void**)&b.gfx, sizeof(unsigned) * (size_t)b.gcap));
        HIP_TRY_C(hipMalloc((void**)&b.gfy, sizeof(unsigned) * (size_t)b.gcap));
        HIP_TRY_C(hipMalloc((void**)&b.gld, sizeof(int) * (size_t)b.gcap));
        HIP_TRY_C(hipMalloc((void**)&b.glo, sizeof(int) * (size_t)b.gcap));
        HIP_TRY_C(hipMalloc((void**)&b.ghi, sizeof(int) * (size_t)b.gcap));
        HIP_TRY_C(hipMalloc((void**)&b.gpos, sizeof(int) * (size_t)b.gcap));
        b.tie_table_size = 1;
        while (b.tie_table_size < 2 * b.gcap) b.tie_table_size <<= 1;
        HIP_TRY_C(hipMalloc((void**)&b.tie_table, sizeof(int) * (size_t)b.tie_table_size));
        HIP_TRY_C(hipMalloc((void**)&b.pld, sizeof(int) * (size_t)n));
        HIP_TRY_C(hipMalloc((void**)&b.keys_merged, sizeof(uint64_t) * nm));
        HIP_TRY_C(hipMalloc((void**)&b.vals_merged, sizeof(int) * nm));
        HIP_TRY_C(hipMalloc((void**)&b.cnt, sizeof(int) * nm));
        HIP_TRY_C(hipMalloc((void**)&b.base, sizeof(int) * nm));
        // (+1: the spare record that bh_walk_kernel's finished groups point at)
        HIP_TRY_C(hipMalloc((void**)&b.tree.A, sizeof(float4) * (size_t)(b.node_cap + 1)));
        HIP_TRY_C(hipMalloc((void**)&b.tree.B, sizeof(int2) * (size_t)(b.node_cap + 1)));
        HIP_TRY_C(hipMalloc((void**)&b.owner, sizeof(int) * (size_t)b.tree.cap_internal));
        HIP_TRY_C(hipMalloc((void**)&b.tree.depth, (size_t)b.tree.cap_internal + 16));   // read 16 bytes at a time
        HIP_TRY_C(hipMalloc((void**)&b.tree.self, sizeof(int) * (size_t)b.tree.cap_internal));
        HIP_TRY_C(hipMalloc((void**)&b.tree.pend, sizeof(int) * (size_t)b.tree.cap_internal));
        HIP_TRY_C(hipMalloc((void**)&b.body4, sizeof(float4) * (size_t)n));
        if (n <= kBhRankSortMax) {                             // both stay zero between builds (the kernel re-zeroes them)
            HIP_TRY_C(hipMalloc((void**)&b.rank, sizeof(int) * (size_t)n));
            HIP_TRY_C(hipMalloc((void**)&b.rank_done, sizeof(int) * ((size_t)n / kBhBlock + 1)));
            HIP_TRY_C(hipMalloc((void**)&b.tie_first_sorted, (size_t)n + 16));
            HIP_TRY_C(hipMemset(b.rank, 0, sizeof(int) * (size_t)n));
            HIP_TRY_C(hipMemset(b.rank_done, 0, sizeof(int) * ((size_t)n / kBhBlock + 1)));
        }
        HIP_TRY_C(hipMalloc((void**)&b.ctr, sizeof(BhCounters)));
        HIP_TRY_C(hipMemset(b.ctr, 0, sizeof(BhCounters)));
        HIP_TRY_C(hipHostMalloc((void**)&b.overflow_host, 4 * sizeof(int), hipHostMallocMapped));
        b.overflow_host[0] = 0;                                // the build's verdict, read by nbody_sync without a copy
        HIP_TRY_C(hipHostGetDevicePointer((void**)&b.overflow_dev, b.overflow_host, 0));
        HIP_TRY_C(hipMalloc((void**)&b.grp_visits, sizeof(int) * ((size_t)n / 8 + 1)));
        HIP_TRY_C(hipMalloc((void**)&b.grp_accepts, sizeof(int) * ((size_t)n / 8 + 1)));
        HIP_TRY_C(hipMemset(b.grp_visits, 0, sizeof(int) * ((size_t)n / 8 + 1)));
        HIP_TRY_C(hipMemset(b.grp_accepts, 0, sizeof(int) * ((size_t)n / 8 + 1)));
        // rocPRIM's sort and scan: the alternative path of rounds 1-2, kept for A/B and cross-checks.  Its scratch (~24 B per
        // body) is only allocated for a context created with NBODY_BH_HIPCUB set; without it the step path is this
        // repository's kernels only and the switch is ignored.
        b.hipcub_ready = std::getenv("NBODY_BH_HIPCUB") != nullptr;
        if (b.hipcub_ready) {
        HIP_TRY_C(hipcub::DeviceRadixSort::SortPairs(nullptr, b.sort_bytes, b.keys, b.keys_sorted, b.vals, b.vals_sorted, n, 0, 64, ctx->stream));
        HIP_TRY_C(hipcub::DeviceRadixSort::SortPairs(nullptr, b.gsort_bytes, b.gkeys, b.gkeys_sorted, b.gvals, b.gvals_sorted, b.gcap, 0, 64, ctx->stream));
        HIP_TRY_C(hipcub::DeviceScan::ExclusiveSum(nullptr, b.scan_bytes, b.cnt, b.base, (int)nm, ctx->stream));
        }
        // own sort / scan state: zeroed once, epochs start at 1 (no tag then equals a zero-initialised look-back word's)
        auto make_sort = [&](SortBuffers& sbuf, int count, uint64_t* tmp_k, int* tmp_v) -> hipError_t {
            hipError_t e = hipMalloc((void**)&sbuf.state, sizeof(SortState));
            if (e != hipSuccess) return e;
            if ((e = hipMemset(sbuf.state, 0, sizeof(SortState))) != hipSuccess) return e;
            const unsigned one = 1;
            if ((e = hipMemcpy(&sbuf.state->epoch, &one, sizeof(one), hipMemcpyHostToDevice)) != hipSuccess) return e;
            sbuf.tiles = (count + kSortBlock * 8 - 1) / (kSortBlock * 8);           // the smaller tile: enough for either
            const size_t lb = sizeof(unsigned long long) * sort_look_words(sbuf.tiles);
            if ((e = hipMalloc((void**)&sbuf.look, lb)) != hipSuccess) return e;
            if ((e = hipMemset(sbuf.look, 0, lb)) != hipSuccess) return e;
            if ((e = hipMalloc((void**)&sbuf.hist_part, sort_hist_part_bytes())) != hipSuccess) return e;
            if ((e = hipMalloc((void**)&sbuf.queue, sizeof(int) * kSortQueueCap)) != hipSuccess) return e;
            if (tmp_k) { sbuf.tmp_k = tmp_k; sbuf.tmp_v = tmp_v; return hipSuccess; }
            if ((e = hipMalloc((void**)&sbuf.tmp_k, sizeof(uint64_t) * (size_t)count)) != hipSuccess) return e;
            return hipMalloc((void**)&sbuf.tmp_v, sizeof(int) * (size_t)count);
        };
        HIP_TRY_C(make_sort(b.sort_main, n, b.keys_merged, b.vals_merged));      // the merged arrays are written after the sort
        {
            HIP_TRY_C(hipMalloc((void**)&b.scan_state, sizeof(ScanState)));
            const ScanState s0{1u, 0, 0, 0};
            HIP_TRY_C(hipMemcpy(b.scan_state, &s0, sizeof(s0), hipMemcpyHostToDevice));
            const size_t tiles = (nm + (size_t)kBhBlock * kBhScanRounds - 1) / ((size_t)kBhBlock * kBhScanRounds);
            HIP_TRY_C(hipMalloc((void**)&b.scan_look, sizeof(unsigned long long) * tiles));
            HIP_TRY_C(hipMemset(b.scan_look, 0, sizeof(unsigned long long) * tiles));
        }
        if (cfg->bh_sorted && sc < n && sc > 0) {
            HIP_TRY_C(hipMalloc((void**)&b.own_keys_sorted, sizeof(uint64_t) * (size_t)sc));
            HIP_TRY_C(hipMalloc((void**)&b.own_order, sizeof(int) * (size_t)sc));
            HIP_TRY_C(make_sort(b.sort_own, sc, nullptr, nullptr));
            if (b.hipcub_ready) {
                HIP_TRY_C(hipcub::DeviceRadixSort::SortPairs(nullptr, b.own_sort_bytes, b.keys, b.own_keys_sorted, b.vals, b.own_order, sc, 0, 64, ctx->stream));
                HIP_TRY_C(hipMalloc(&b.own_sort_tmp, b.own_sort_bytes ? b.own_sort_bytes : 16));
            }
        }
        if (b.hipcub_ready) {
            HIP_TRY_C(hipMalloc(&b.gsort_tmp, b.gsort_bytes ? b.gsort_bytes : 16));
            HIP_TRY_C(hipMalloc(&b.sort_tmp, b.sort_bytes ? b.sort_bytes : 16));
            HIP_TRY_C(hipMalloc(&b.scan_tmp, b.scan_bytes ? b.scan_bytes : 16));
        }
    }
#undef HIP_TRY_C
    *out = ctx;
    return NBODY_OK;
}

void nbody_destroy(nbody_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->step_graph) (void)hipGraphExecDestroy(ctx->step_graph);
    destroy_comm(ctx);
    float* arrs[] = {ctx->m, ctx->px, ctx->py, ctx->vx, ctx->vy, ctx->radius, ctx->c, ctx->ax, ctx->ay, ctx->pax, ctx->pay};
    for (float* p : arrs)
        if (p) (void)hipFree(p);
    if (ctx->diag_partials) (void)hipFree(ctx->diag_partials);
    {
        auto& b = ctx->bh;
        void* ptrs[] = {b.keys, b.keys_sorted, b.vals, b.vals_sorted, b.cnt, b.base, b.tree.A, b.tree.B,
                        b.tree.depth, b.tree.self, b.tree.pend, b.rank, b.rank_done, b.tie_first_sorted, b.body4, b.ctr, b.tie, b.owner, b.gfx, b.gfy, b.gld, b.pld, b.glo, b.ghi, b.gpos, b.tie_table, b.sort_tmp, b.scan_tmp, b.gkeys, b.gkeys_sorted, b.gvals,
                        b.gvals_sorted, b.keys_merged, b.vals_merged, b.gsort_tmp, b.grp_visits, b.grp_accepts,
                        b.own_keys_sorted, b.own_order, b.own_sort_tmp,
                        b.sort_main.state, b.sort_main.look, b.sort_main.hist_part, b.sort_main.queue,
                        b.sort_own.state, b.sort_own.look, b.sort_own.hist_part, b.sort_own.queue, b.sort_own.tmp_k, b.sort_own.tmp_v, b.scan_state, b.scan_look};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (b.overflow_host) (void)hipHostFree(b.overflow_host);
    }
    for (auto& pool : ctx->prof.pool)
        for (auto& ep : pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    {
        auto& f = ctx->frames;
        if (f.copy_stream) { (void)hipStreamSynchronize(f.copy_stream); (void)hipStreamDestroy(f.copy_stream); }
        for (int k = 0; k < nbody_ctx::Frames::kSlots; ++k) {
            if (f.dev[k]) (void)hipFree(f.dev[k]);
            if (f.host[k]) (void)hipHostFree(f.host[k]);
            if (f.snapped[k]) (void)hipEventDestroy(f.snapped[k]);
            if (f.landed[k]) (void)hipEventDestroy(f.landed[k]);
        }
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int nbody_set_stream(nbody_ctx* ctx, void* hip_stream)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_set_stream: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return NBODY_OK;
}

int nbody_upload(nbody_ctx* ctx, const float* m, const float* posX, const float* posY, const float* velX,
                 const float* velY, const float* radius)
{
    if (!ctx || !m || !posX || !posY || !velX || !velY || !radius)
        return fail(NBODY_ERR_INVALID, "nbody_upload: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->cfg.num_bodies;
    // Choose the pair-formula variant from the data (see force_naive.hip.h for the conditions).
    bool uni_m = true, uni_r = true, finite = true, core_ok = true;
    float rmin = radius[0];
    for (int i = 0; i < n; ++i) {
        // Barnes-Hut node kernel: operand ranges of the short correctly rounded sequences (exact_core.hip.h)
        core_ok &= (m[i] == 0.0f || (m[i] >= kCoreMassMin && m[i] <= kCoreMassMax)) && radius[i] >= 0.0f && radius[i] <= kCoreMassMax;
        uni_m &= (m[i] == m[0]);
        uni_r &= (radius[i] == radius[0]);
        rmin = radius[i] < rmin ? radius[i] : rmin;
        finite &= std::isfinite(m[i]) && std::isfinite(radius[i]) && std::isfinite(posX[i]) && std::isfinite(posY[i]);
    }
    if (!finite) return fail(NBODY_ERR_INVALID, "nbody_upload: non-finite mass/radius/position");
    int variant = ctx->cfg.force_variant;
    const bool clamp_ok = rmin >= 4.0f;          // R^2 >= 16: inv^2 stands in for 1/(d2+1e-6) to < 1 ulp
    const float r0 = radius[0];
    int r_exp = 0;
    // pre-scaling by 1/R is exact iff R is a power of two; clamp(inv'^2) == min(inv^2, 1/(R^2+soft2))*R^2
    // iff R^2 + soft2 rounds to R^2 in fp32
    const bool scalable = uni_m && uni_r && clamp_ok && std::frexp(r0, &r_exp) == 0.5f && (r0 * r0 + kSoft2 == r0 * r0);
    if (variant == 0) variant = !clamp_ok ? kExact : (scalable ? kScaled : (uni_m && uni_r ? kUniform : kPerBody));
    if (variant == kScaled && !scalable)
        return fail(NBODY_ERR_INVALID, "nbody_upload: force_variant=scaled needs constant m and a constant power-of-two radius >= 4");
    if (variant == kUniform && !(uni_m && uni_r))
        return fail(NBODY_ERR_INVALID, "nbody_upload: force_variant=uniform but m/radius arrays are not constant");
    if (variant != kExact && !clamp_ok)
        return fail(NBODY_ERR_INVALID, "nbody_upload: force_variant %d needs every radius >= 4 (min %g)", variant, rmin);
    ctx->variant = variant;
    ctx->bh_exact_core = core_ok && !std::getenv("NBODY_BH_NO_CORE");
    ctx->uni_c = 1.0f / (r0 * r0 + kSoft2);
    ctx->pre_scale = variant == kScaled ? 1.0f / r0 : 1.0f;
    ctx->out_scale = variant == kScaled ? m[0] / (r0 * r0) : (variant == kUniform ? m[0] : 1.0f);
    plan_force(ctx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    drop_step_graph(ctx);                        // variant, launch plan and buffers may all change: re-capture on the next step
    const int rows = ctx->cfg.use_barnes_hut ? 1 : ctx->slices;
    if (rows > ctx->partial_rows) {
        if (ctx->pax) HIP_TRY(hipFree(ctx->pax));
        if (ctx->pay) HIP_TRY(hipFree(ctx->pay));
        ctx->pax = ctx->pay = nullptr;
        ctx->partial_rows = 0;
        const size_t pb = sizeof(float) * (size_t)rows * ctx->i_stride;
        HIP_TRY(hipMalloc((void**)&ctx->pax, pb));
        HIP_TRY(hipMalloc((void**)&ctx->pay, pb));
        ctx->partial_rows = rows;
    }
    std::vector<float> c((size_t)n);
    for (int i = 0; i < n; ++i)
        c[i] = variant == kExact ? radius[i] * radius[i] : 1.0f / (radius[i] * radius[i] + kSoft2);

    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(float) * (size_t)n;
    HIP_TRY(hipMemcpy(ctx->m, m, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->px, posX, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->py, posY, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->vx, velX, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->vy, velY, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->radius, radius, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->c, c.data(), nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(ctx->ax, 0, nb));
    HIP_TRY(hipMemset(ctx->ay, 0, nb));
    ctx->uploaded = true;
    ctx->steps_enqueued = 0;
    if (ctx->frames.ready) {                     // frames of the previous state are dropped
        HIP_TRY(hipStreamSynchronize(ctx->frames.copy_stream));
        ctx->frames.acquired = ctx->frames.published;
    }
    return NBODY_OK;
}

int nbody_upload_state(nbody_ctx* ctx, const float* posX, const float* posY, const float* velX, const float* velY)
{
    if (!ctx || !posX || !posY || !velX || !velY) return fail(NBODY_ERR_INVALID, "nbody_upload_state: null argument");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_upload_state: call nbody_upload first (it brings m and radius)");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->cfg.num_bodies;
    bool finite = true;
    for (int i = 0; i < n; ++i) finite &= std::isfinite(posX[i]) && std::isfinite(posY[i]);
    if (!finite) return fail(NBODY_ERR_INVALID, "nbody_upload_state: non-finite position");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(float) * (size_t)n;
    HIP_TRY(hipMemcpy(ctx->px, posX, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->py, posY, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->vx, velX, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->vy, velY, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(ctx->ax, 0, nb));
    HIP_TRY(hipMemset(ctx->ay, 0, nb));
    ctx->steps_enqueued = 0;
    if (ctx->frames.ready) {
        HIP_TRY(hipStreamSynchronize(ctx->frames.copy_stream));
        ctx->frames.acquired = ctx->frames.published;
    }
    return NBODY_OK;
}

namespace {

int enqueue_one_step(nbody_ctx* ctx);

void drop_step_graph(nbody_ctx* ctx);
}  // namespace
namespace {
void drop_step_graph(nbody_ctx* ctx)
{
    if (ctx->step_graph) (void)hipGraphExecDestroy(ctx->step_graph);
    ctx->step_graph = nullptr;
    ctx->step_graph_stream = nullptr;
}

// Capture one step into a hipGraph (Barnes-Hut only; skipped while per-kernel event timing is on, because event pairs
// recorded inside a capture cannot be queried).  Any failure falls back to eager launches for good.  The naive step's
// two launches are NOT captured: measured back to back at N = 10 000 (tools/naive_small_time.py,
// profiles/r3_naive_small_sizes.txt) the graph replays them in 35.9 us per step, the stream issues them in 31.4.
bool ensure_step_graph(nbody_ctx* ctx)
{
    if (!ctx->cfg.use_barnes_hut || ctx->prof.on || ctx->graph_failed || std::getenv("NBODY_NO_GRAPH")) return false;
    if (ctx->step_graph && ctx->step_graph_stream == ctx->stream) return true;
    drop_step_graph(ctx);
    hipGraph_t graph = nullptr;
    const bool dbg = std::getenv("NBODY_DEBUG") != nullptr;
    const hipError_t be = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
    if (be != hipSuccess) {
        if (dbg) std::fprintf(stderr, "[nbody] graph: begin capture failed: %s\n", hipGetErrorString(be));
        ctx->graph_failed = true;
        (void)hipGetLastError();
        return false;
    }
    const int rc = enqueue_one_step(ctx);
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (dbg) std::fprintf(stderr, "[nbody] graph: capture rc=%d (%s) end=%s graph=%p\n", rc, rc ? g_err : "ok", hipGetErrorString(e), (void*)graph);
    if (rc != NBODY_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        ctx->graph_failed = true;
        return false;
    }
    const hipError_t ie = hipGraphInstantiate(&ctx->step_graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) { ctx->step_graph = nullptr; ctx->graph_failed = true; (void)hipGetLastError(); return false; }
    ctx->step_graph_stream = ctx->stream;
    return true;
}

}  // namespace

int nbody_step(nbody_ctx* ctx, int32_t nsteps)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_step: null context");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_step: nbody_upload has not been called");
    if (nsteps < 0) return fail(NBODY_ERR_INVALID, "nbody_step: nsteps=%d", nsteps);
    HIP_TRY(hipSetDevice(ctx->device));
    // A context that owns only part of the bodies advances only those: after ONE step every other body's position is
    // stale until the owners' slices have been exchanged.  Several steps in one call are therefore either refused or,
    // with a communicator attached (nbody_comm_init), run as nsteps x (step + nbody_exchange_positions).
    const int n = ctx->cfg.num_bodies;
    const bool partial = ctx->cfg.shard_begin != 0 || (ctx->cfg.shard_count != 0 && ctx->cfg.shard_count != n);
    if (partial && nsteps > 1 && !ctx->comm)
        return fail(NBODY_ERR_INVALID, "nbody_step: nsteps=%d on a context that owns only bodies [%d,+%d) of %d: the other bodies' "
                                       "positions would be stale after the first step.  Call nbody_step(ctx, 1) and exchange posX/posY "
                                       "between steps, or attach a communicator (nbody_comm_init)",
                    nsteps, ctx->cfg.shard_begin, ctx->cfg.shard_count, n);
    const bool exchange_each = partial && nsteps > 1;       // (a single step leaves the exchange to the caller, as before)
    const bool graph = nsteps > 0 && ensure_step_graph(ctx);
    for (int s = 0; s < nsteps; ++s) {
        if (graph) {
            HIP_TRY(hipGraphLaunch(ctx->step_graph, ctx->stream));
        } else {
            const int rc = enqueue_one_step(ctx);
            if (rc) return rc;
        }
        ++ctx->steps_enqueued;
        if (exchange_each) {
            const int rc = nbody_exchange_positions(ctx);
            if (rc) return rc;
        }
    }
    return NBODY_OK;
}

namespace {
int enqueue_one_step(nbody_ctx* ctx)
{
    {
        int rc;
        if (ctx->cfg.use_barnes_hut) {
            rc = enqueue_bh_build(ctx);                      // benchmark.cpp:357
            if (rc) return rc;
            if (!ctx->cfg.use_parallel) return NBODY_OK;     // benchmark.cpp:359: tree built, nothing else happens
            rc = enqueue_bh_walk(ctx, 1);                    // :365-401
        } else {
            rc = enqueue_force(ctx);                         // :407-478
        }
        if (rc) return rc;
        rc = enqueue_integrate(ctx, 1);                      // :382-401 / :480-510
        if (rc) return rc;
    }
    return NBODY_OK;
}
}  // namespace

int nbody_compute_accel(nbody_ctx* ctx)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_compute_accel: null context");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_compute_accel: nbody_upload has not been called");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    if (ctx->cfg.use_barnes_hut) {
        rc = enqueue_bh_build(ctx);
        if (rc) return rc;
        rc = enqueue_bh_walk(ctx, 0);
    } else {
        rc = enqueue_force(ctx);
    }
    if (rc) return rc;
    return enqueue_integrate(ctx, 0);
}

int nbody_sync(nbody_ctx* ctx)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_sync: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->cfg.use_barnes_hut) {
        // the flag of the LAST build, written by its node kernel into host-mapped memory: no copy, no extra sync
        // (the C++ loops call nbody_sync every frame of a ~0.3 ms step)
        const int overflow = *static_cast<volatile int*>(ctx->bh.overflow_host);
        if (overflow == 1)
            return fail(NBODY_ERR_INVALID, "Barnes-Hut node pool exhausted (%d internal cells); bodies were dropped from the tree "
                                           "as the reference does at benchmark.cpp:102", ctx->bh.tree.cap_internal);
        if (overflow == 3)
            return fail(NBODY_ERR_HIP, "Barnes-Hut build: a bounded look-back wait of the device sort/scan gave up (internal error; "
                                       "the tree of this step is not valid)");
        if (overflow == 2)
            return fail(NBODY_ERR_INVALID, "Barnes-Hut: more than %d bodies sit exactly on split lines; the surplus was inserted once",
                        ctx->bh.gcap);
    }
    return NBODY_OK;
}

int nbody_download(nbody_ctx* ctx, float* posX, float* posY, float* velX, float* velY)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_download: null context");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_download: nothing uploaded");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(float) * (size_t)ctx->cfg.num_bodies;
    if (posX) HIP_TRY(hipMemcpy(posX, ctx->px, nb, hipMemcpyDeviceToHost));
    if (posY) HIP_TRY(hipMemcpy(posY, ctx->py, nb, hipMemcpyDeviceToHost));
    if (velX) HIP_TRY(hipMemcpy(velX, ctx->vx, nb, hipMemcpyDeviceToHost));
    if (velY) HIP_TRY(hipMemcpy(velY, ctx->vy, nb, hipMemcpyDeviceToHost));
    return NBODY_OK;
}

int nbody_frame_publish(nbody_ctx* ctx)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_frame_publish: null context");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_frame_publish: nothing uploaded");
    HIP_TRY(hipSetDevice(ctx->device));
    auto& f = ctx->frames;
    constexpr int kSlots = nbody_ctx::Frames::kSlots;
    const size_t nb = sizeof(float) * (size_t)ctx->cfg.num_bodies;
    if (!f.ready) {
        // (re-entrant after a failed attempt: only what is still missing is created; nbody_destroy frees the rest)
        if (!f.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&f.copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < kSlots; ++k) {
            if (!f.dev[k]) HIP_TRY(hipMalloc((void**)&f.dev[k], 2 * nb));
            if (!f.host[k]) HIP_TRY(hipHostMalloc((void**)&f.host[k], 2 * nb, hipHostMallocDefault));
            if (!f.snapped[k]) HIP_TRY(hipEventCreateWithFlags(&f.snapped[k], hipEventDisableTiming));
            if (!f.landed[k]) HIP_TRY(hipEventCreateWithFlags(&f.landed[k], hipEventDisableTiming));
        }
        f.ready = true;
    }
    if (f.published - f.acquired >= kSlots)
        return fail(NBODY_ERR_INVALID, "nbody_frame_publish: %d frames already in flight; call nbody_frame_acquire first", kSlots);
    const int k = (int)(f.published % kSlots);
    // The slot's previous frame was acquired, so its host copy has landed and the device snapshot is free.
    // Snapshot on the compute stream (8 B/body, device to device), host copy on the copy stream: later steps
    // overwrite posX/posY in place without waiting for PCIe.
    float* snap = f.dev[k];
    HIP_TRY(hipMemcpyAsync(snap, ctx->px, nb, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(snap + ctx->cfg.num_bodies, ctx->py, nb, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipEventRecord(f.snapped[k], ctx->stream));
    HIP_TRY(hipStreamWaitEvent(f.copy_stream, f.snapped[k], 0));
    HIP_TRY(hipMemcpyAsync(f.host[k], snap, 2 * nb, hipMemcpyDeviceToHost, f.copy_stream));
    HIP_TRY(hipEventRecord(f.landed[k], f.copy_stream));
    f.step[k] = ctx->steps_enqueued;
    ++f.published;
    return NBODY_OK;
}

int nbody_frame_acquire(nbody_ctx* ctx, const float** posX, const float** posY, int64_t* step_index)
{
    if (!ctx || !posX || !posY) return fail(NBODY_ERR_INVALID, "nbody_frame_acquire: null argument");
    auto& f = ctx->frames;
    if (!f.ready || f.acquired >= f.published)
        return fail(NBODY_ERR_INVALID, "nbody_frame_acquire: no published frame is outstanding");
    HIP_TRY(hipSetDevice(ctx->device));
    const int k = (int)(f.acquired % nbody_ctx::Frames::kSlots);
    HIP_TRY(hipEventSynchronize(f.landed[k]));
    *posX = f.host[k];
    *posY = f.host[k] + ctx->cfg.num_bodies;
    if (step_index) *step_index = f.step[k];
    ++f.acquired;
    return NBODY_OK;
}

int nbody_get_accel(nbody_ctx* ctx, float* accX, float* accY)
{
    if (!ctx || !accX || !accY) return fail(NBODY_ERR_INVALID, "nbody_get_accel: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(float) * (size_t)ctx->cfg.num_bodies;
    HIP_TRY(hipMemcpy(accX, ctx->ax, nb, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(accY, ctx->ay, nb, hipMemcpyDeviceToHost));
    return NBODY_OK;
}

int nbody_diagnostics(nbody_ctx* ctx, double out[5])
{
    if (!ctx || !out) return fail(NBODY_ERR_INVALID, "nbody_diagnostics: null argument");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_diagnostics: nothing uploaded");
    HIP_TRY(hipSetDevice(ctx->device));
    DiagArgs a{};
    a.m = ctx->m; a.px = ctx->px; a.py = ctx->py; a.vx = ctx->vx; a.vy = ctx->vy; a.radius = ctx->radius;
    a.n = ctx->cfg.num_bodies; a.partials = ctx->diag_partials;
    diagnostics_kernel<<<dim3(ctx->diag_blocks), dim3(kDiagBlock), 0, ctx->stream>>>(a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<double> h((size_t)5 * ctx->diag_blocks);
    HIP_TRY(hipMemcpy(h.data(), ctx->diag_partials, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < 5; ++k) out[k] = 0.0;
    for (int b = 0; b < ctx->diag_blocks; ++b)
        for (int k = 0; k < 5; ++k) out[k] += h[(size_t)b * 5 + k];
    return NBODY_OK;
}

void* nbody_device_ptr(nbody_ctx* ctx, int32_t which)
{
    if (!ctx) { fail(NBODY_ERR_INVALID, "nbody_device_ptr: null context"); return nullptr; }
    float* t[] = {ctx->m, ctx->px, ctx->py, ctx->vx, ctx->vy, ctx->radius, ctx->ax, ctx->ay};
    if (which < 0 || which > 7) { fail(NBODY_ERR_INVALID, "nbody_device_ptr: which=%d", which); return nullptr; }
    return t[which];
}

int nbody_profile_enable(nbody_ctx* ctx, int32_t on)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_profile_enable: null context");
    ctx->prof.on = on != 0;
    return NBODY_OK;
}

int nbody_profile_read(nbody_ctx* ctx, int32_t which, double* total_ms, int64_t* launches)
{
    if (!ctx || which < 0 || which > 2) return fail(NBODY_ERR_INVALID, "nbody_profile_read: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    Profiler& p = ctx->prof;
    for (size_t i = 0; i < p.used[which]; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.pool[which][i].a, p.pool[which][i].b));
        p.total_ms[which] += ms;
        push_sample(p.samples[which], ms);
    }
    p.used[which] = 0;
    if (total_ms) *total_ms = p.total_ms[which];
    if (launches) *launches = p.launches[which];
    return NBODY_OK;
}

int nbody_profile_reset(nbody_ctx* ctx)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_profile_reset: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; ++k) { ctx->prof.used[k] = 0; ctx->prof.total_ms[k] = 0; ctx->prof.launches[k] = 0; ctx->prof.samples[k].clear(); }
    return NBODY_OK;
}

int64_t nbody_profile_samples(nbody_ctx* ctx, int32_t which, float* out_ms, int64_t cap)
{
    double total = 0;
    int64_t launches = 0;
    const int rc = nbody_profile_read(ctx, which, &total, &launches);     // drains the outstanding event pairs
    if (rc) return rc;
    const auto& v = ctx->prof.samples[which];
    const int64_t k = (int64_t)v.size() < cap ? (int64_t)v.size() : cap;
    if (out_ms && k > 0) std::memcpy(out_ms, v.data(), sizeof(float) * (size_t)k);
    return (int64_t)v.size();
}

int nbody_get_info(nbody_ctx* ctx, int32_t* force_variant, int32_t* j_slices, int32_t* blocks)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_get_info: null context");
    if (force_variant) *force_variant = ctx->variant;
    if (j_slices) *j_slices = ctx->slices;
    if (blocks) *blocks = ctx->blocks_x;
    return NBODY_OK;
}

int64_t nbody_bh_tree_nodes(nbody_ctx* ctx, void* out, int64_t cap)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_bh_tree_nodes: null context");
    if (!ctx->cfg.use_barnes_hut) return fail(NBODY_ERR_INVALID, "nbody_bh_tree_nodes: context is not in Barnes-Hut mode");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    BhCounters h{};
    HIP_TRY(hipMemcpy(&h, ctx->bh.ctr, sizeof(h), hipMemcpyDeviceToHost));
    const int64_t nodes = 1 + 4 * (int64_t)h.internal;
    if (!out) return nodes;
    const int64_t k = nodes < cap ? nodes : cap;
    if (k <= 0) return nodes;
    std::vector<float4> A((size_t)k), box((size_t)k);
    std::vector<int2> B((size_t)k);
    HIP_TRY(hipMemcpy(A.data(), ctx->bh.tree.A, sizeof(float4) * (size_t)k, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(B.data(), ctx->bh.tree.B, sizeof(int2) * (size_t)k, hipMemcpyDeviceToHost));
    float4* dbox = nullptr;                                    // boxes are not kept by the build: recompute them
    HIP_TRY(hipMalloc((void**)&dbox, sizeof(float4) * (size_t)k));
    bh_box_kernel<<<dim3((unsigned)((k + kBhBlock - 1) / kBhBlock)), dim3(kBhBlock), 0, ctx->stream>>>(ctx->bh.tree, ctx->bh.ctr, dbox, k);
    hipError_t be = hipStreamSynchronize(ctx->stream);
    if (be == hipSuccess) be = hipMemcpy(box.data(), dbox, sizeof(float4) * (size_t)k, hipMemcpyDeviceToHost);
    (void)hipFree(dbox);
    HIP_TRY(be);
    struct QuadNodeRecord {   // layout of the reference's QuadNode, benchmark.cpp:52-60
        float mass, comX, comY, tlX, tlY, brX, brY;
        int32_t bodyIndex, children[4];
    };
    static_assert(sizeof(QuadNodeRecord) == 48, "QuadNode is 48 bytes");
    QuadNodeRecord* o = (QuadNodeRecord*)out;
    for (int64_t i = 0; i < k; ++i) {
        o[i].mass = A[i].z; o[i].comX = A[i].x; o[i].comY = A[i].y;
        o[i].tlX = box[i].x; o[i].tlY = box[i].y; o[i].brX = box[i].z; o[i].brY = box[i].w;
        o[i].bodyIndex = B[i].y;
        for (int q = 0; q < 4; ++q) o[i].children[q] = B[i].x >= 0 ? B[i].x + q : -1;
    }
    return nodes;
}

int nbody_bh_counters(nbody_ctx* ctx, int64_t out[2])
{
    if (!ctx || !out) return fail(NBODY_ERR_INVALID, "nbody_bh_counters: null argument");
    if (!ctx->cfg.use_barnes_hut) return fail(NBODY_ERR_INVALID, "nbody_bh_counters: context is not in Barnes-Hut mode");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    BhCounters h{};
    HIP_TRY(hipMemcpy(&h, ctx->bh.ctr, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = (int64_t)h.visits;
    out[1] = (int64_t)h.accepts;
#ifdef NBODY_WIDE_TIMING
    if (std::getenv("NBODY_DEBUG")) {
        std::fprintf(stderr, "[nbody] prep phases (us): valid+ties %.2f fixpoint %.2f ghosts/merge %.2f count+scan+owners %.2f\n",
                     (h.dbg[1] - h.dbg[0]) * 0.01, (h.dbg[2] - h.dbg[1]) * 0.01, (h.dbg[3] - h.dbg[2]) * 0.01, (h.dbg[6] - h.dbg[3]) * 0.01);
    }
#endif
    if (std::getenv("NBODY_DEBUG"))
        std::fprintf(stderr, "[nbody] walk: %llu group-visits in %llu wave iterations: %.1f %% of the 8 group slots busy; "
                             "%d second instances (bodies on split lines), %d valid instances, %d internal cells\n",
                     h.visits, h.wave_iters, h.wave_iters ? 100.0 * (double)h.visits / (8.0 * (double)h.wave_iters) : 0.0,
                     h.ghosts, h.valid, h.internal);
    return NBODY_OK;
}

int nbody_bh_group_counters(nbody_ctx* ctx, int32_t* visits, int32_t* accepts)
{
    if (!ctx || !visits || !accepts) return fail(NBODY_ERR_INVALID, "nbody_bh_group_counters: null argument");
    if (!ctx->cfg.use_barnes_hut) return fail(NBODY_ERR_INVALID, "nbody_bh_group_counters: context is not in Barnes-Hut mode");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nb = sizeof(int) * (size_t)((ctx->cfg.num_bodies + 7) / 8);
    HIP_TRY(hipMemcpy(visits, ctx->bh.grp_visits, nb, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(accepts, ctx->bh.grp_accepts, nb, hipMemcpyDeviceToHost));
    return NBODY_OK;
}

// ---- nbody_accel_noalloc: the force phase for callers that cannot allocate (no counterpart in the reference) ----
// The j-slice partial rows and the per-body clamp constants, which a context keeps in hipMalloc'ed buffers, live
// here in a file-scope device array.  `scratch` is a HOST-SIDE STAND-IN for that array: host code never passes the
// name to a kernel or a memset; it resolves the device address once per device (hipGetSymbolAddress) and uses that.
constexpr int kNoAllocMaxBodies = 65536;
constexpr size_t kNoAllocScratchFloats = (size_t)kNoAllocMaxBodies * (2 * kMaxSlices + 1);   // pax + pay rows, c[]: 33.8 MB
constexpr int kNoAllocMaxDevices = 64;

static __device__ __align__(256) float scratch[kNoAllocScratchFloats];

namespace {

// c[i] = R_i^2: the exact-general variant's clamp constant (what nbody_upload derives on the host for a context)
// ... and the N%8 tail of the outputs, which is never a target (benchmark.cpp:423), gets its zeros here: a
// hipMemsetAsync of those 4-28 bytes is not safe under stream capture (ROCm 7.2 replayed the captured 12-byte memset node
// with the byte value 3: tests/test_zz_no_alloc.py::test_captured_into_a_graph_and_replayed)
__global__ __launch_bounds__(256) void noalloc_prepare_kernel(const float* __restrict__ radius, float* __restrict__ c, int n, int lim,
                                                              float* __restrict__ ax, float* __restrict__ ay)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float r = radius[i]; c[i] = r * r; }
    if (i >= lim && i < n) { ax[i] = 0.0f; ay[i] = 0.0f; }
}

float* g_scratch_ptr[kNoAllocMaxDevices] = {};
std::mutex g_scratch_mutex;

// the device address of `scratch` on the CURRENT device (each GPU holds its own copy at its own address)
int noalloc_scratch(int dev, float** out)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(NBODY_ERR_HIP, "nbody_accel_noalloc: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return NBODY_OK;
}

}  // namespace

int nbody_accel_noalloc(int32_t num_bodies, const float* dev_m, const float* dev_posX, const float* dev_posY,
                        const float* dev_radius, float* dev_accX, float* dev_accY, void* hip_stream)
{
    if (!dev_m || !dev_posX || !dev_posY || !dev_radius || !dev_accX || !dev_accY)
        return fail(NBODY_ERR_INVALID, "nbody_accel_noalloc: null argument");
    if (num_bodies < 1 || num_bodies > kNoAllocMaxBodies)
        return fail(NBODY_ERR_INVALID, "nbody_accel_noalloc: num_bodies=%d outside [1, %d] (the static scratch array holds "
                                       "the partial rows of at most %d bodies; use a context for more)",
                    num_bodies, kNoAllocMaxBodies, kNoAllocMaxBodies);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kNoAllocMaxDevices) return fail(NBODY_ERR_INVALID, "nbody_accel_noalloc: device ordinal %d", dev);
    float* sp = nullptr;
    int rc = noalloc_scratch(dev, &sp);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int n = num_bodies;
    const int lim = n - n % 8;                                   // benchmark.cpp:423: the N%8 tail is never a target
    if (lim <= 0) noalloc_prepare_kernel<<<dim3(1), dim3(256), 0, s>>>(dev_radius, sp, n, lim, dev_accX, dev_accY);   // n < 8: zeros only
    if (lim > 0) {
        const int stride = ((lim + 63) / 64) * 64;
        const ForcePlan p = make_force_plan_env(n, lim, kExact);   // = plan_force of the equivalent context
        if ((size_t)n + 2 * (size_t)p.slices * stride > kNoAllocScratchFloats)
            return fail(NBODY_ERR_INVALID, "nbody_accel_noalloc: internal: plan exceeds the scratch array");
        float* c = sp;                                           // [n]
        float* pax = sp + kNoAllocMaxBodies;                     // [slices][stride]
        float* pay = pax + (size_t)kMaxSlices * kNoAllocMaxBodies;
        // nothing in the scratch array has to start at zero: c[0..n) and every partial row [0..slices) x [0..lim)
        // are written in full by this call before they are read, so the previous call's contents never show
        noalloc_prepare_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(dev_radius, c, n, lim, dev_accX, dev_accY);
        ForceArgs a{};
        a.px = dev_posX; a.py = dev_posY; a.m = dev_m; a.c = c;
        a.pax = pax; a.pay = pay;
        a.n = n; a.i_begin = 0; a.i_end = lim; a.i_stride = stride;
        a.j_per_slice = p.j_per_slice;
        a.pre_scale = 1.0f; a.uni_c = 0.0f; a.out_scale = 1.0f;
        launch_force_k<kExact>(p.k, p.tile, a, dim3(p.blocks_x, p.slices), s);
        IntegrateArgs g{};
        g.pax = pax; g.pay = pay; g.ax = dev_accX; g.ay = dev_accY;
        g.i_begin = 0; g.i_end = lim; g.i_stride = stride; g.slices = p.slices;
        g.dt = 0.0f; g.advance = 0;
        integrate_kernel<<<dim3((lim + 255) / 256), dim3(256), 0, s>>>(g);
    }
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

// Parity hook for bh_sort.hip.h: the build's own device sort on host arrays (tests compare with a stable host sort).
int nbody_sort_pairs_u64(const uint64_t* keys, const int32_t* vals, int32_t n, uint64_t* out_keys, int32_t* out_vals, int32_t repeats,
                         int32_t truncated)
{
    if (!keys || !vals || !out_keys || !out_vals || n < 1 || repeats < 1) return fail(NBODY_ERR_INVALID, "nbody_sort_pairs_u64: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(NBODY_ERR_NO_DEVICE, "nbody_sort_pairs_u64: no HIP device visible");
    SortBuffers sb;
    uint64_t *dk = nullptr, *ok = nullptr;
    int *dv = nullptr, *ov = nullptr;
    SortState host_state{};
    auto run = [&]() -> int {
        const size_t kb = sizeof(uint64_t) * (size_t)n, vb = sizeof(int) * (size_t)n;
        HIP_TRY(hipMalloc((void**)&dk, kb)); HIP_TRY(hipMalloc((void**)&ok, kb));
        HIP_TRY(hipMalloc((void**)&dv, vb)); HIP_TRY(hipMalloc((void**)&ov, vb));
        HIP_TRY(hipMalloc((void**)&sb.tmp_k, kb)); HIP_TRY(hipMalloc((void**)&sb.tmp_v, vb));
        HIP_TRY(hipMalloc((void**)&sb.state, sizeof(SortState)));
        HIP_TRY(hipMemset(sb.state, 0, sizeof(SortState)));
        const unsigned one = 1;
        HIP_TRY(hipMemcpy(&sb.state->epoch, &one, sizeof(one), hipMemcpyHostToDevice));
        const size_t lb = sizeof(unsigned long long) * sort_look_words((n + kSortBlock * 8 - 1) / (kSortBlock * 8));
        HIP_TRY(hipMalloc((void**)&sb.look, lb));
        HIP_TRY(hipMemset(sb.look, 0, lb));
        HIP_TRY(hipMalloc((void**)&sb.hist_part, sort_hist_part_bytes()));
        HIP_TRY(hipMalloc((void**)&sb.queue, sizeof(int) * kSortQueueCap));
        HIP_TRY(hipMemcpy(dk, keys, kb, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dv, vals, vb, hipMemcpyHostToDevice));
        for (int r = 0; r < repeats; ++r) {                       // repeats: the state (epochs, tickets, histograms) carries over
            HIP_TRY(hipMemsetAsync(ok, 0xA5, kb, nullptr));
            sort_pairs(sb, dk, dv, ok, ov, n, nullptr, truncated != 0);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out_keys, ok, kb, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_vals, ov, vb, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&host_state, sb.state, sizeof(SortState), hipMemcpyDeviceToHost));
        return NBODY_OK;
    };
    int rc = run();
    void* ptrs[] = {dk, ok, dv, ov, sb.tmp_k, sb.tmp_v, sb.state, sb.look, sb.hist_part, sb.queue};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
#ifdef NBODY_WIDE_TIMING
    if (rc == NBODY_OK && host_state.dbg_cnt) {
        const char* nm[] = {"plan+ticket", "loads", "rank", "bins+scan", "tile tree", "LDS scatter", "global scatter"};
        for (int q = 0; q < 7; ++q)
            std::fprintf(stderr, "[sort] n=%d up to '%s': mean %.2f us  max %.2f us (per live tile, since the workgroup started)\n", n, nm[q],
                         (double)host_state.dbg_sum[q] / (double)host_state.dbg_cnt * 0.01, (double)host_state.dbg_max[q] * 0.01);
    }
#endif
    if (rc == NBODY_OK && host_state.error) rc = fail(NBODY_ERR_HIP, "nbody_sort_pairs_u64: a look-back wait gave up");
    if (rc == NBODY_OK && host_state.epoch != 1u + (unsigned)repeats) rc = fail(NBODY_ERR_HIP, "nbody_sort_pairs_u64: epoch %u after %d sorts", host_state.epoch, repeats);
    return rc;
}

const char* nbody_last_error(void) { return g_err; }

int nbody_abi_version(void) { return NBODY_ABI_VERSION; }

}  // extern "C"

#include "comm_rccl.hip.h"

void destroy_comm(nbody_ctx* ctx)
{
    if (ctx->comm && rccl_api().CommDestroy) (void)rccl_api().CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
}
