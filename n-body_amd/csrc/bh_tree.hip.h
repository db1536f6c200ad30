// bh_tree.hip.h — Barnes-Hut on gfx950: tree build, centre-of-mass pass, theta-criterion walk.
//
// Replaces QuadTree::build / insert_recursive (benchmark.cpp:77-173),
// computeForce_vec (benchmark.cpp:228-277, A5) and computeForce_kernel (benchmark.cpp:176-225, A6).
//
// The reference inserts bodies one by one, recursively and serially.  Its result, however, is canonical:
// a region quadtree over the FIXED box [-spread,spread]^2 (benchmark.cpp:169) in which a cell is split
// (all four children allocated, benchmark.cpp:105-108) iff it holds >= 2 bodies, children ordered
// TL,TR,BL,BR = (-x-y, +x-y, -x+y, +x+y), split lines at the float midpoint (tl+br)/2 (benchmark.cpp:98-99),
// bodies outside the box dropped (benchmark.cpp:83-85).  That tree is rebuilt here without any insertion:
//
//   1. bh_keys_kernel      per body: descend 31 levels with the reference's own float comparisons and
//                          midpoints -> 62-bit path key (2 bits per level); out-of-box -> invalid key
//   2. hipcub radix sort   (key, body) pairs; invalid keys sort last
//  2b. bh_ties_kernel + bh_merge_kernel   second instances of bodies on split lines (see below)
//   3. bh_count_kernel     common-prefix length (in levels) of neighbouring keys: the cell of depth d that
//                          STARTS at sorted body i is internal iff lcp(i-1,i) < d <= lcp(i,i+1), so body i
//                          owns max(0, lcp(i,i+1) - lcp(i-1,i)) internal cells;  exclusive scan -> cell ids
//   4. bh_nodes_kernel     one thread per internal cell c: binary-search the three quadrant boundaries of
//                          its body range, emit its four children at node indices 1+4c .. 4+4c (leaf with
//                          its body / empty leaf / internal child with childBase = 1 + 4*id(child))
//   5. bh_com_climb_kernel bottom-up in one launch (the last child to finish evaluates the parent); above 2^21 bodies
//                          bh_com_kernel, one launch per depth: mass = ((0+m0)+m1)+m2)+m3,
//                          com = sum(com_k*m_k)/mass in child order (benchmark.cpp:122-141)
//
// Up to 32 768 bodies (the reference's default is 10 000) steps 2b-4a are one launch of one workgroup
// (bh_prep_small_kernel), and up to ~50 000 bodies the walk gives every group four waves (bh_walk_quad_kernel); the
// separate full-width kernels serve larger trees.
//
// Everything is deterministic (no atomic decides any index or any value) and nothing synchronises with the host.
// Node order is Morton order, not the reference's insertion order; topology, boxes, body indices, masses
// and centres of mass are identical to oracle_tree_build() (tests compare them by traversal).
// Bodies EXACTLY on a split line: the reference's bounds are inclusive on all four sides
// (benchmark.cpp:83-85), so such a body is inserted into BOTH children of an internal cell and its mass is
// counted twice from there up (97 bodies in 10^6 — root mass 1.0001e9).  Reproduced:
// bh_ties_kernel expands, for every body, the instances its split-line ties spawn against the primaries' tree, and
// bh_tie_fixpoint_kernel closes the set (an instance can turn the leaf it lands in into an internal cell and so switch
// on a tie of the body that lived there: a least fixed point, reproduced in full — see the comment above BhTies); the
// instances are sorted and merged into the key array before the cells are counted.  Bodies closer together than the
// finest cell of the 31-level key (and coincident ones) share one leaf: the reference keeps splitting them until its
// 32N pool is exhausted (benchmark.cpp:102).
//
// Walk: one group of 8 consecutive body indices per 8 lanes (8 groups per wave64), the reference's vote
// ("accept iff leaf or all 8 lanes far", benchmark.cpp:264-266) taken with one wave ballot, DFS order equal
// to the reference's explicit stack (children pushed 0..3, popped 3..0), so every lane accumulates the same
// nodes in the same order as oracle_bh_accel().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "exact_core.hip.h"
#include "bh_sort.hip.h"

namespace nbody {

#ifdef NBODY_WIDE_TIMING
#define BH_DBG_TICK(ctr, k) do { if (threadIdx.x == 0) (ctr)->dbg[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BH_DBG_TICK(ctr, k) do { } while (0)
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and store the
// wave has in flight (its fence covers all address spaces): in a one-workgroup kernel that keeps prefetches and
// result stores in flight across its phases, that turns every barrier into a ~2 us memory round trip.
__device__ __forceinline__ void bh_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kBhLevels = 31;                       // path-key digits (2 bits each)
constexpr uint64_t kBhInvalidKey = 1ull << 63;      // outside the root box
constexpr int kBhBlock = 256;
constexpr int kBhStackDepth = 34;

struct BhCounters {          // device-resident scalars of one build/walk
    int valid;               // M: bodies inside the root box
    int internal;            // I: internal cells
    int overflow;            // node pool (1) or tie-instance buffer (2) too small
    int ghosts;              // G: second instances of bodies on split lines
    unsigned long long visits, accepts;
    unsigned long long wave_iters;   // walk: loop iterations summed over waves (lane utilisation = visits / (8 * wave_iters))
    float root_half;         // half-width h of the root box [-h,h]^2 of this build (= spread unless adaptive)
#ifdef NBODY_WIDE_TIMING
    unsigned long long dbg[24];  // developer build: 100 MHz ticks at phase boundaries of the one-workgroup kernels
#endif
    int valid_part[64];      // small-N build: in-box bodies counted by bh_keys_kernel (one atomic per wave, spread over 64
                             // counters), summed by bh_prep_small_kernel, zeroed by bh_nodes_kernel for the next build
};

// walk record of a node: A = {comX, comY, mass, width}, B = {childBase (-1: leaf), bodyIndex (-1: none)}
struct BhTree {
    float4* __restrict__ A;
    int2* __restrict__ B;
    uint8_t* __restrict__ depth;         // per internal cell
    int* __restrict__ self;              // per internal cell: its node index
    int* pend;                           // per internal cell: internal children not yet finished (-1: it has none), bh_com_climb_kernel
    int cap_internal;
    int com_reciprocal;                  // 1: com = sum * (1/mass) as main.cpp:173-175 does; 0: sum / mass (benchmark.cpp:139-140)
};

__device__ __forceinline__ int bh_lcp_levels(uint64_t a, uint64_t b)
{
    const uint64_t x = a ^ b;
    int l = x ? (__builtin_clzll(x) - 2) / 2 : kBhLevels;
    return l > kBhLevels - 1 ? kBhLevels - 1 : l;          // internal cells live at depth <= 30
}

// 1. path keys ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bh_reset_counters(BhCounters* __restrict__ ctr, float root_half0);

// fixed_box != 0 (the reference's root box): the half-width comes by value and thread 0 also resets the counters of
// this build (a separate 1-thread launch otherwise; every launch of a short kernel costs ~4.5 us here).  Every
// launch also pre-fills the tie-instance buffer with padding that sorts last (bh_ties_kernel overwrites the front).
__global__ __launch_bounds__(kBhBlock) void bh_keys_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                           int n, BhCounters* __restrict__ ctr, float spread_fixed,
                                                           int fixed_box, uint64_t* __restrict__ keys,
                                                           int* __restrict__ vals, uint16_t* __restrict__ tie,
                                                           uint64_t* __restrict__ gkeys, int* __restrict__ gvals, int gcap,
                                                           int* __restrict__ tie_table, int tie_table_size, int count_valid,
                                                           const float* __restrict__ m, float4* __restrict__ body4)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    for (int g = i; g < gcap; g += gridDim.x * kBhBlock) { gkeys[g] = ~0ull; gvals[g] = -1; }
    for (int g = i; g < tie_table_size; g += gridDim.x * kBhBlock) tie_table[g] = -1;
    if (fixed_box && i == 0) bh_reset_counters(ctr, spread_fixed);
    bool inside = false;
    if (i < n) {
        const float spread = fixed_box ? spread_fixed : ctr->root_half;
        const float x = px[i], y = py[i];
        float tlx = -spread, tly = -spread, brx = spread, bry = spread;
        uint64_t key;
        int tx = kBhLevels, ty = kBhLevels;                    // first level at which x (y) sits ON the split line
        if (x < tlx || x > brx || y < tly || y > bry) {         // benchmark.cpp:83-85
            // invalid keys stay distinct and ordered by body; the index sits ABOVE bit 24 (it fits: n < 2^28), so that
            // the sort's truncated sweeps (bits 24-63, bh_sort.hip.h) already order them completely
            key = kBhInvalidKey | ((uint64_t)(uint32_t)i << 24);
        } else {
            inside = true;
            // The kernel is VALU-bound (31 levels for every body), so the level is written for the instruction count
            // (round 3: 24 -> 16 vector instructions per level).  Key bits go in by doubling-and-adding into two 32-bit
            // words (levels 0-14 and 15-30).  The first tie level is COUNTED: a coordinate on a split line takes the lower
            // child, becomes the upper bound of its cell and stays it (br == x from then on), and before that br > x — so
            // the number of levels after which br != x is the first level with x == m, 31 when there is none.  (A
            // coordinate ON the box's upper edge has br == x from the start and is recorded at level 0: a lower bound,
            // which is all the record's readers need — they filter with it and take the exact ties from a descent.)
            unsigned khi = 0, klo = 0;
            tx = 0; ty = 0;
#pragma unroll 1
            for (int l = 0; l < 15; ++l) {
                const float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;    // benchmark.cpp:98-99
                const bool qx = x > mx, qy = y > my;           // on the line -> lower child
                khi = khi + khi + (qy ? 1u : 0u);              // TL, TR, BL, BR (benchmark.cpp:105-108): bits (y, x)
                khi = khi + khi + (qx ? 1u : 0u);
                tlx = qx ? mx : tlx; brx = qx ? brx : mx;
                tly = qy ? my : tly; bry = qy ? bry : my;
                tx += brx != x ? 1 : 0;
                ty += bry != y ? 1 : 0;
            }
#pragma unroll 1
            for (int l = 15; l < kBhLevels; ++l) {
                const float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;
                const bool qx = x > mx, qy = y > my;
                klo = klo + klo + (qy ? 1u : 0u);
                klo = klo + klo + (qx ? 1u : 0u);
                tlx = qx ? mx : tlx; brx = qx ? brx : mx;
                tly = qy ? my : tly; bry = qy ? bry : my;
                tx += brx != x ? 1 : 0;
                ty += bry != y ? 1 : 0;
            }
            key = ((uint64_t)khi << 32) | klo;
        }
        keys[i] = key;
        vals[i] = i;
        tie[i] = (uint16_t)(tx | (ty << 8));
        // {x, y, m} of the body in ONE record: bh_nodes_kernel gathers the leaves' bodies by index, and a random 4-byte
        // read costs a whole memory sector — one sector per body instead of three (0.67 of the 2.0 ms the node kernel
        // took at N = 2^24 was this gather)
        body4[i] = make_float4(x, y, m[i], 0.0f);
    }
    // small-N build (count_valid): number of in-box bodies = where the invalid keys start after the sort, one atomic
    // per wave.  Measured: same-line device-scope atomics cost ~4 ns each and a short kernel cannot end before they
    // are all acknowledged — 15 625 of them turn this 20-us kernel into a 79-us one at N = 10^6, so the large build
    // keeps its one-thread binary search (bh_valid_kernel, ~5 us).
    if (count_valid) {
        const unsigned long long in_mask = __ballot(inside);
        if ((threadIdx.x & 63) == 0 && in_mask)
            atomicAdd(&ctr->valid_part[(blockIdx.x * (kBhBlock / 64) + (threadIdx.x >> 6)) & 63], __popcll(in_mask));
    }
}

__device__ __forceinline__ int bh_valid_total(const BhCounters* __restrict__ ctr)
{
    int m = 0;
#pragma unroll
    for (int k = 0; k < 64; ++k) m += ctr->valid_part[k];
    return m;
}

// 2 (small N). Sort by RANKING: rank(i) = #{j : (key_j, j) < (key_i, i)} counted by brute force, n^2 64-bit compares.  At
// N = 10 000 that is 10^8 compares = 21 us in one launch (~14 SIMD cycles per wave-wide compare: v_cmp_lt_u64, the
// add and a VCC wait state; a 32-bit borrow chain in its place was measured slower), against 35 us for the five
// launches of a general device sort on an input this small (block sort + four merge passes, each launch-latency
// bound).  Break-even at ~16 000 bodies, used up to kBhRankSortMax.  Stable by
// construction (ties — bodies sharing a finest cell — order by index, as a stable radix sort leaves them), no atomic
// decides anything: partial counts of the j-slices are ADDED into rank[i] (integer sums commute), and the workgroup
// that arrives last at its i-tile (count-off as in bh_com_climb_kernel: adds acknowledged, then one atomic on the
// tile's counter) reads the complete ranks with device-scope loads, scatters the tile and leaves rank[] and the
// counter zero for the next build.
__device__ __forceinline__ int bh_first_tie(uint16_t rec);
constexpr int kBhRankChunk = 512;                   // keys staged in LDS per round
constexpr int kBhRankSortMax = 12288;               // bodies (measured: -14 us at 10 000, +-0 at 16 384, +54 us at 32 768)

__global__ __launch_bounds__(kBhBlock) void bh_ranksort_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals, int n,
                                                               int j_per_slice, int* rank, int* done,
                                                               uint64_t* __restrict__ keys_sorted, int* __restrict__ vals_sorted,
                                                               const uint16_t* __restrict__ tie,
                                                               unsigned char* __restrict__ tie_first_sorted)
{
    __shared__ __align__(16) uint64_t tile[kBhRankChunk];
    __shared__ int last;
    const int tid = threadIdx.x;
    const int ilo = blockIdx.x * kBhBlock, i = ilo + tid;
    const bool live = i < n;
    const uint64_t ki = keys[live ? i : n - 1];
    const int j0 = blockIdx.y * j_per_slice, j1 = min(n, j0 + j_per_slice);
    int cnt = 0;
    for (int jb = j0; jb < j1; jb += kBhRankChunk) {
#pragma unroll
        for (int q = tid; q < kBhRankChunk; q += kBhBlock) tile[q] = jb + q < j1 ? keys[jb + q] : ~0ull;   // padding: never counted
        __syncthreads();
        // (key_j, j) < (key_i, i)  <=>  key_j < key_i + (j < i); keys have bit 62 or bit 63 clear below ~0: no overflow
        if (jb + kBhRankChunk <= ilo || jb >= ilo + kBhBlock) {          // the whole chunk is on one side of the tile
            const uint64_t t = ki + (jb < ilo ? 1u : 0u);
#pragma unroll 8
            for (int q = 0; q < kBhRankChunk; ++q) cnt += tile[q] < t ? 1 : 0;
        } else {
#pragma unroll 8
            for (int q = 0; q < kBhRankChunk; ++q) cnt += tile[q] < ki + (jb + q < i ? 1u : 0u) ? 1 : 0;
        }
        __syncthreads();
    }
    if (live && cnt) __hip_atomic_fetch_add(&rank[i], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                  // this wave's adds are acknowledged ...
    __syncthreads();                                // ... and so are the other waves', before the tile is counted off
    if (tid == 0) last = __hip_atomic_fetch_add(&done[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.y - 1;
    __syncthreads();
    if (!last) return;
    if (live) {
        const int r = __hip_atomic_load(&rank[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int body = vals[i];
        if ((unsigned)r < (unsigned)n) {
            keys_sorted[r] = ki;
            vals_sorted[r] = body;
            // first level at which the body sits on a split line, in sorted order: the preparation kernel reads it next
            // to the keys instead of gathering it through vals_sorted
            tie_first_sorted[r] = (unsigned char)bh_first_tie(tie[body]);
        }
        rank[i] = 0;
    }
    if (tid == 0) done[blockIdx.x] = 0;
}

// large build: number of valid (in-box) bodies = first sorted index with the invalid bit set
__global__ void bh_valid_kernel(const uint64_t* __restrict__ keys, int n, BhCounters* __restrict__ ctr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] & kBhInvalidKey) hi = mid; else lo = mid + 1;
    }
    ctr->valid = lo;
}

// root_half0: the fixed box (spread), or a tiny positive floor that bh_bounds_kernel raises (adaptive box)
__device__ __forceinline__ void bh_reset_counters(BhCounters* __restrict__ ctr, float root_half0)
{
    ctr->root_half = root_half0;
    ctr->valid = 0;
    ctr->internal = 0;
    ctr->overflow = 0;
    ctr->ghosts = 0;
    ctr->visits = 0;
    ctr->accepts = 0;
    ctr->wave_iters = 0;
}

__global__ void bh_reset_kernel(BhCounters* __restrict__ ctr, float root_half0)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bh_reset_counters(ctr, root_half0);
}

// Adaptive root box (opt-in, nbody_config.adaptive_box): h = max_i max(|x_i|,|y_i|) over finite positions.
// Non-negative floats order like their bit patterns, so the cross-workgroup maximum is an integer atomicMax.
__global__ __launch_bounds__(kBhBlock) void bh_bounds_kernel(const float* __restrict__ px, const float* __restrict__ py, int n,
                                                             BhCounters* __restrict__ ctr)
{
    float h = 0.0f;
    for (int i = blockIdx.x * kBhBlock + threadIdx.x; i < n; i += gridDim.x * kBhBlock) {
        const float ax = fabsf(px[i]), ay = fabsf(py[i]);
        if (ax < __builtin_huge_valf()) h = fmaxf(h, ax);      // false for inf and NaN
        if (ay < __builtin_huge_valf()) h = fmaxf(h, ay);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h = fmaxf(h, __shfl_xor(h, o));
    __shared__ float part[kBhBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBhBlock / 64; ++w) h = fmaxf(h, part[w]);
        atomicMax(reinterpret_cast<int*>(&ctr->root_half), __float_as_int(h));
    }
}

// One descent with the reference's float arithmetic.  fx/fy: bit l set = take the UPPER child on that axis at level
// l regardless of the comparison (an instance that went "up" at a split line the body sits on).  Returns the key
// and, per axis, the mask of the levels at which the coordinate equals the midpoint WITHOUT being forced there,
// i.e. the split lines this instance sits on and went "down" at.
__device__ __forceinline__ uint64_t bh_descend(float x, float y, float spread, unsigned fx, unsigned fy, unsigned* tie_x,
                                               unsigned* tie_y)
{
    float tlx = -spread, tly = -spread, brx = spread, bry = spread;
    uint64_t key = 0;
    unsigned tx = 0, ty = 0;
#pragma unroll 1
    for (int l = 0; l < kBhLevels; ++l) {
        const float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;    // benchmark.cpp:98-99
        const bool ux = (fx >> l) & 1u, uy = (fy >> l) & 1u;
        if (x == mx && !ux) tx |= 1u << l;
        if (y == my && !uy) ty |= 1u << l;
        const int qx = (x > mx) || ux, qy = (y > my) || uy;               // on the line -> lower child
        key = (key << 2) | (uint64_t)(qy * 2 + qx);                      // TL, TR, BL, BR (benchmark.cpp:105-108)
        tlx = qx ? mx : tlx; brx = qx ? brx : mx;
        tly = qy ? my : tly; bry = qy ? bry : my;
    }
    *tie_x = tx;
    *tie_y = ty;
    return key;
}

// 2b. second instances.  A body exactly ON the split line of an INTERNAL cell is in both children (inclusive bounds,
// benchmark.cpp:83-85), and each of those instances can sit on further split lines below.  The instances of a body
// form a little tree: instance I spawns "I, but up at level t on axis a" for every unforced tie (a, t) of its own
// path whose cell is internal, i.e. t < leaf depth of I = 1 + the longest common prefix of I's key with the key of
// any instance of ANOTHER body.  Forced levels are only ever added in increasing (level, axis) order, so every
// combination has exactly one parent.  Because a second instance can turn the leaf it lands in into an internal
// cell — and so switch on a tie of the body that lived there — the set is a least fixed point:
//   bh_ties_kernel          one thread per sorted primary, depth-first over its instances, leaf depths taken
//                           against the PRIMARIES only (the whole answer for all but contrived inputs);
//   bh_tie_fixpoint_kernel  one workgroup, rounds until nothing is added: every instance added in the previous
//                           round raises the known leaf depth of itself and of its neighbours (atomicMax: whoever
//                           raises it from a to b spawns the ties in [a, b), so nothing is spawned twice).
struct BhTies {
    uint64_t* __restrict__ gkeys;        // [gcap] second instances, unsorted
    int* __restrict__ gvals;             // body of each
    unsigned* __restrict__ gfx;          // forced-up levels of each, per axis
    unsigned* __restrict__ gfy;
    int* __restrict__ gld;               // leaf depth up to which its ties have been expanded
    int* __restrict__ glo;               // its neighbours among the sorted primaries (own primary skipped): indices,
    int* __restrict__ ghi;               //   -1 / M when there is none
    int* __restrict__ gpos;              // its insertion position among the sorted primaries
    int* __restrict__ pld;               // per SORTED PRIMARY (index into keys): leaf depth up to which its ties have been
                                         //   expanded; written for primaries with a recorded tie only
    int* __restrict__ table;             // open-addressing hash: insertion position -> instance index (-1 = empty);
    int table_mask;                      //   cleared by bh_keys_kernel, size = table_mask + 1 >= 2 * gcap
    int gcap;
};

__device__ __forceinline__ int bh_tie_hash(int pos, int mask) { return (int)(((unsigned)pos * 2654435761u) >> 7) & mask; }

__device__ __forceinline__ int bh_lower_bound(const uint64_t* __restrict__ keys, int lo, int hi, uint64_t v);

// ties of instance (x, y, fx, fy) at levels [from, to): append the spawned instances.  first_tie: lowest level at
// which the instance can have a tie at all, when the caller knows it (saves the descent in the common case).
__device__ __forceinline__ void bh_spawn_ties(float x, float y, float spread, unsigned fx, unsigned fy, int from, int to, int body,
                                              int first_tie, const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                              int M, const BhTies& T, BhCounters* __restrict__ ctr)
{
    if (to > kBhLevels) to = kBhLevels;
    if (from >= to || first_tie >= to) return;
    unsigned tx, ty;
    (void)bh_descend(x, y, spread, fx, fy, &tx, &ty);
    const unsigned forced = fx | fy;
    const int last = forced ? 31 - __builtin_clz(forced) : -1;
    const bool last_is_y = last >= 0 && ((fy >> last) & 1u);
    for (int t = from > last ? from : (last < 0 ? 0 : last); t < to; ++t) {
        const bool x_ok = ((tx >> t) & 1u) && t > last;                            // (level, axis) order: x before y
        const bool y_ok = ((ty >> t) & 1u) && (t > last || (t == last && !last_is_y));
        for (int a = 0; a < 2; ++a) {
            if (!(a == 0 ? x_ok : y_ok)) continue;
            const unsigned nfx = a == 0 ? fx | (1u << t) : fx, nfy = a == 1 ? fy | (1u << t) : fy;
            unsigned ux, uy;
            const uint64_t k = bh_descend(x, y, spread, nfx, nfy, &ux, &uy);
            const int pos = bh_lower_bound(keys, 0, M, k);
            int lo = pos - 1, hi = pos;
            if (lo >= 0 && vals[lo] == body) --lo;
            if (hi < M && vals[hi] == body) ++hi;
            const int slot = atomicAdd(&ctr->ghosts, 1);
            if (slot < T.gcap) {
                T.gkeys[slot] = k; T.gvals[slot] = body; T.gfx[slot] = nfx; T.gfy[slot] = nfy; T.gld[slot] = 0;
                T.glo[slot] = lo; T.ghi[slot] = hi; T.gpos[slot] = pos;
                // The record before its index becomes findable — for observers on this CU only (workgroup scope: a wait,
                // no cache-wide operation).  Nobody else can need more: a reader only follows table entries below the
                // instance count it took at the start of its round (bh_tie_fixpoint_body), and those records were written
                // before a kernel boundary or before the workgroup barrier that ended the round.  (Rounds 1-2 had a
                // device-scope __threadfence() here — an L2 write-back and invalidate per instance: 185 of the 232 us of
                // bh_ties_kernel and most of the fixed-point kernel at N = 2^24, 4271 instances.)
#ifdef NBODY_SPAWN_FENCE_DEVICE
                __threadfence();
#else
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
                int hs = bh_tie_hash(pos, T.table_mask);
                while (atomicCAS(&T.table[hs], -1, slot) != -1) hs = (hs + 1) & T.table_mask;
            } else {
                ctr->overflow = 2;
            }
        }
    }
}

__device__ __forceinline__ int bh_first_tie(uint16_t rec)
{
    const int tx = rec & 0xFF, ty = rec >> 8;
    return tx < ty ? tx : ty;                                    // kBhLevels when the primary's path has none
}

// Lowest level at which a primary with path key k can sit on a split line at all, from the key alone (round 3: replaces a
// 2-byte gather through the sorted index for EVERY body — 1 GB of memory sectors at N = 2^24 — by arithmetic on the key
// the thread has already loaded; the exact record is then read by the few bodies that pass).  A coordinate equal to the
// midpoint of level t takes the lower child there (bit 0) and is from then on the upper bound of its cell: it compares
// above every later midpoint (bit 1) until a midpoint ROUNDS onto it, which needs a cell no wider than an ulp of the
// coordinate, i.e. level >= 24 for coordinates inside the root box (cell width 2 h / 2^l against ulp <= h / 2^23).  So a
// tie at level t <= 22 shows in the key as a 0 at level t followed by 1s down to level 22: the only candidate is the 0
// just above the trailing 1s of levels 0..22; levels 23..30 stay possible for everybody.
__device__ __forceinline__ int bh_min_tie_level(uint64_t k)
{
    const uint64_t odd = 0xAAAAAAAAAAAAAAAAull;
    // x bits of levels 22, 21, ... 0 at bits 0, 2, ... 44 of k >> 16 (y: k >> 17); bit 46 is key bit 62 (63): always 0
    const int rx = __builtin_ctzll(~((k >> 16) | odd)) >> 1, ry = __builtin_ctzll(~((k >> 17) | odd)) >> 1;
    // (an axis whose 23 bits are all 1 — a coordinate on the box's upper edge — has no candidate up here: per axis, then min)
    const int tx = rx >= 23 ? 23 : 22 - rx, ty = ry >= 23 ? 23 : 22 - ry;
    return tx < ty ? tx : ty;
}

__device__ __forceinline__ void bh_ties_one(int i, int M, const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                            const float* __restrict__ px, const float* __restrict__ py,
                                            const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr, const BhTies& T)
{
    const uint64_t k = keys[i];
    const int lp = i > 0 ? bh_lcp_levels(keys[i - 1], k) : -1;
    const int ln = i + 1 < M ? bh_lcp_levels(k, keys[i + 1]) : -1;
    const int leaf_depth = 1 + (lp > ln ? lp : ln);              // 0 when the body is alone in the tree
    T.pld[i] = leaf_depth;                                       // every primary: the fixed point raises it from here
    const float spread = ctr->root_half;
    // (the ulp argument needs normal floats: a root box of denormal size takes the exact path for every body)
#ifndef NBODY_TIES_KEY_FILTER
#define NBODY_TIES_KEY_FILTER 1
#endif
    if (NBODY_TIES_KEY_FILTER && spread >= 1e-20f && bh_min_tie_level(k) >= leaf_depth) return;   // no split line above the leaf: nearly everybody
    const int b = vals[i];
    const int first = bh_first_tie(tie[b]);
    if (first >= leaf_depth) return;
    bh_spawn_ties(px[b], py[b], spread, 0u, 0u, 0, leaf_depth, b, first, keys, vals, M, T, ctr);
}

// (Tried, round 3: four primaries per thread — fewer, fatter waves.  Slower, 447 against 232 us at N = 2^24: the kernel is
// not bound by the wave launch rate but by the latency of the few lanes that pass the filter.)
__global__ __launch_bounds__(kBhBlock) void bh_ties_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                           const float* __restrict__ px, const float* __restrict__ py,
                                                           const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                           BhTies T)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    const int M = ctr->valid;
    if (i >= M) return;
    bh_ties_one(i, M, keys, vals, px, py, tie, ctr, T);
}

constexpr int kBhFixBlock = 1024;
__device__ __forceinline__ void bh_tie_fixpoint_body(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                     const float* __restrict__ px, const float* __restrict__ py,
                                                     const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                     const BhTies& T, const int M)
{
    __shared__ int s_end;
    const float spread = ctr->root_half;
    int begin = 0;
    if (threadIdx.x == 0) { const int g = __hip_atomic_load(&ctr->ghosts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_end = g < T.gcap ? g : T.gcap; }
    __syncthreads();
    int end = s_end;
    while (begin < end) {                                        // block-uniform; ends when a round adds nothing
        for (int g = begin + (int)threadIdx.x; g < end; g += kBhFixBlock) {
            const uint64_t k = T.gkeys[g];
            const int b = T.gvals[g];
            const int lo = T.glo[g], hi = T.ghi[g];
            int ld = 0;
            // neighbours among the primaries: this instance deepens their leaves
            for (int side = 0; side < 2; ++side) {
                const int j = side ? hi : lo;
                if (j < 0 || j >= M) continue;
                const int d = 1 + bh_lcp_levels(keys[j], k);
                ld = d > ld ? d : ld;
                const int bj = vals[j];
                const int first = bh_first_tie(tie[bj]);
                if (first < d) {
                    const int old = atomicMax(&T.pld[j], d);
                    if (d > old) bh_spawn_ties(px[bj], py[bj], spread, 0u, 0u, old, d, bj, first, keys, vals, M, T, ctr);
                }
            }
            // other bodies' second instances between the same primaries (lo, hi]: at most three insertion positions,
            // looked up in the hash; instances added during this round (index >= end) wait for the next one
            for (int ph = lo + 1; ph <= hi; ++ph) {
                for (int hs = bh_tie_hash(ph, T.table_mask);; hs = (hs + 1) & T.table_mask) {
                    const int h = __hip_atomic_load(&T.table[hs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (h < 0) break;
                    if (h >= end || T.gpos[h] != ph) continue;
                    const int bh_ = T.gvals[h];
                    if (bh_ == b) continue;
                    const int d = 1 + bh_lcp_levels(T.gkeys[h], k);
                    ld = d > ld ? d : ld;
                    const int old = atomicMax(&T.gld[h], d);
                    if (d > old) bh_spawn_ties(px[bh_], py[bh_], spread, T.gfx[h], T.gfy[h], old, d, bh_, 0, keys, vals, M, T, ctr);
                }
            }
            const int old = atomicMax(&T.gld[g], ld);
            if (ld > old) bh_spawn_ties(px[b], py[b], spread, T.gfx[g], T.gfy[g], old, ld, b, 0, keys, vals, M, T, ctr);
        }
        __syncthreads();
        begin = end;
        if (threadIdx.x == 0) { const int g = __hip_atomic_load(&ctr->ghosts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_end = g < T.gcap ? g : T.gcap; }
        __syncthreads();
        end = s_end;
    }
}

__global__ __launch_bounds__(kBhFixBlock) void bh_tie_fixpoint_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                                      const float* __restrict__ px, const float* __restrict__ py,
                                                                      const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                                      BhTies T)
{
    bh_tie_fixpoint_body(keys, vals, px, py, tie, ctr, T, ctr->valid);
}

// Large builds: the fixed point AND the sort of the second instances in one launch of one workgroup, total order
// (key, body), the rest of the buffer padded — instead of a device-wide sort of a mostly-padding buffer.  Bitonic network;
// up to kBhLdsGhosts instances entirely in LDS (96 KB: one workgroup runs, the CU's 160 KB are its own).  More of them
// (only above 2^21 bodies, and then only on inputs with many thousands of bodies exactly on split lines): chunks of
// kBhLdsGhosts are sorted in LDS, and of every later merge step only the strides >= a chunk run over device memory (loads
// bypass the L1), the rest again chunk by chunk in LDS.  The output arrays are allocated to the next power of two.
constexpr int kBhSmallGhosts = 4096;                // tie buffer up to which the SMALL build (bh_prep_small_kernel) is used
constexpr int kBhLdsGhosts = 8192;                  // second instances the LDS of bh_tie_fixpoint_sort_kernel holds
__global__ __launch_bounds__(kBhFixBlock) void bh_tie_fixpoint_sort_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                                           const float* __restrict__ px, const float* __restrict__ py,
                                                                           const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                                           BhTies T, uint64_t* gkeys_sorted, int* gvals_sorted)
{
    __shared__ uint64_t sk[kBhLdsGhosts];
    __shared__ int sv[kBhLdsGhosts];
    const int tid = threadIdx.x;
    bh_tie_fixpoint_body(keys, vals, px, py, tie, ctr, T, ctr->valid);
    __syncthreads();            // one workgroup = one CU = one L1: a barrier is all the ordering the phases need
    const int ghosts = __hip_atomic_load(&ctr->ghosts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int G = ghosts < T.gcap ? ghosts : T.gcap;
    int P = 2;
    while (P < G) P <<= 1;
    const int C = P < kBhLdsGhosts ? P : kBhLdsGhosts;          // chunk: what one LDS load holds
    unsigned long long* gk = reinterpret_cast<unsigned long long*>(gkeys_sorted);
    // stages `j_from`, j_from / 2, ... 1 of merge step k on the chunk at c0, in LDS (directions by the GLOBAL index)
    auto lds_stages = [&](int c0, int k, int j_from) {
        for (int j = j_from; j > 0; j >>= 1) {
            for (int p = tid; p < (C >> 1); p += kBhFixBlock) {
                const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
                const uint64_t ka = sk[lo], kb = sk[hi];
                const int va = sv[lo], vb = sv[hi];
                const bool gt = ka > kb || (ka == kb && va > vb);          // total order (key, body)
                if (gt == (((c0 + lo) & k) == 0)) { sk[lo] = kb; sk[hi] = ka; sv[lo] = vb; sv[hi] = va; }
            }
            __syncthreads();
        }
    };
    if (G > 0) {
        // every chunk: load the unsorted instances (padding behind them), all merge steps up to the chunk size, store
        for (int c0 = 0; c0 < P; c0 += C) {
            for (int e = tid; e < C; e += kBhFixBlock) { sk[e] = c0 + e < G ? T.gkeys[c0 + e] : ~0ull; sv[e] = c0 + e < G ? T.gvals[c0 + e] : -1; }
            __syncthreads();
            for (int k = 2; k <= C; k <<= 1) lds_stages(c0, k, k >> 1);
            if (P > C)
                for (int e = tid; e < C; e += kBhFixBlock) { gkeys_sorted[c0 + e] = sk[e]; gvals_sorted[c0 + e] = sv[e]; }
            __syncthreads();
        }
        for (int k = 2 * C; k <= P; k <<= 1) {
            for (int j = k >> 1; j >= C; j >>= 1) {             // strides that cross chunks: over device memory
                for (int p = tid; p < (P >> 1); p += kBhFixBlock) {
                    const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
                    const uint64_t ka = __hip_atomic_load(gk + lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t kb = __hip_atomic_load(gk + hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int va = __hip_atomic_load(gvals_sorted + lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int vb = __hip_atomic_load(gvals_sorted + hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool gt = ka > kb || (ka == kb && va > vb);
                    if (gt == ((lo & k) == 0)) { gkeys_sorted[lo] = kb; gkeys_sorted[hi] = ka; gvals_sorted[lo] = vb; gvals_sorted[hi] = va; }
                }
                __syncthreads();
            }
            for (int c0 = 0; c0 < P; c0 += C) {                 // the strides inside a chunk: in LDS
                for (int e = tid; e < C; e += kBhFixBlock) {
                    sk[e] = __hip_atomic_load(gk + c0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sv[e] = __hip_atomic_load(gvals_sorted + c0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                lds_stages(c0, k, C >> 1);
                for (int e = tid; e < C; e += kBhFixBlock) { gkeys_sorted[c0 + e] = sk[e]; gvals_sorted[c0 + e] = sv[e]; }
                __syncthreads();
            }
        }
    }
    if (G > 0 && P <= C) {                                      // sorted in one piece: still in LDS
        for (int e = tid; e < T.gcap; e += kBhFixBlock) { gkeys_sorted[e] = e < G ? sk[e] : ~0ull; gvals_sorted[e] = e < G ? sv[e] : -1; }
    } else {                                                    // (the padding of the last chunk already sorts behind the instances)
        for (int e = (G > 0 ? P : 0) + tid; e < T.gcap; e += kBhFixBlock) { gkeys_sorted[e] = ~0ull; gvals_sorted[e] = -1; }
    }
}

// merge sorted primaries [n] and sorted second instances [gcap] into one sorted array [n + gcap]
__device__ __forceinline__ void bh_merge_one(int t, const uint64_t* __restrict__ keys, const int* __restrict__ vals, int n,
                                             const uint64_t* __restrict__ gkeys, const int* __restrict__ gvals, int gcap,
                                             uint64_t* __restrict__ okeys, int* __restrict__ ovals)
{
    if (t < n) {
        const uint64_t k = keys[t];
        int lo = 0, hi = gcap;                                 // second instances with a smaller key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (gkeys[mid] < k) lo = mid + 1; else hi = mid; }
        okeys[t + lo] = k;
        ovals[t + lo] = vals[t];
    } else if (t < n + gcap) {
        const int g = t - n;
        const uint64_t k = gkeys[g];
        if (k == ~0ull) { okeys[t] = k; ovals[t] = gvals[g]; return; }   // padding behind the instances: stays where it is
        int lo = 0, hi = n;                                    // primaries with a key <= mine
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] <= k) lo = mid + 1; else hi = mid; }
        // Second instances with EQUAL keys (bodies sharing a finest cell that sit on the same split lines) arrive from
        // the device radix sort in the order in which the atomics of step 2b happened to append them.  Which body a
        // shared leaf names — and so whose self term is masked — must not depend on that: within a run of equal keys
        // the instances go out ordered by body, as the small build's LDS sort leaves them.  Runs are a few entries.
        int g0 = g, before = 0;
        const int body = gvals[g];
        if (k != ~0ull) {                                      // (the padding behind the instances is one long run)
            while (g0 > 0 && gkeys[g0 - 1] == k) --g0;
            for (int j = g0; j < gcap && gkeys[j] == k; ++j) before += gvals[j] < body || (gvals[j] == body && j < g);
        }
        okeys[g0 + before + lo] = k;
        ovals[g0 + before + lo] = body;
    }
}

// A workgroup merges kBhMergePer x 256 consecutive entries.  Its primaries are a sorted stretch: the number of second
// instances before the first and before the last of them (two lanes search, once per workgroup) bracket everybody
// else's, and with a few thousand instances among millions of primaries the bracket is almost always empty — the
// 16-step binary search per primary of round 2 (246 us at N = 2^24) is gone.
constexpr int kBhMergePer = 4;
__global__ __launch_bounds__(kBhBlock) void bh_merge_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals, int n,
                                                            const uint64_t* __restrict__ gkeys, const int* __restrict__ gvals,
                                                            int gcap, uint64_t* __restrict__ okeys, int* __restrict__ ovals,
                                                            BhCounters* __restrict__ ctr)
{
    __shared__ int s_lo[2];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * (kBhBlock * kBhMergePer);
    // valid INSTANCES of the merged array = valid primaries + second instances (their keys have bit 63 clear, the
    // padding and the out-of-box bodies have it set and sort behind them); nobody else in this launch reads ctr
    if (t0 == 0 && tid == 0) ctr->valid += ctr->ghosts < gcap ? ctr->ghosts : gcap;
    const int p1 = min(n, t0 + kBhBlock * kBhMergePer);         // primaries of this workgroup: [t0, p1)
    // the workgroup's own pairs are requested first: they are in flight while two lanes walk their dependent probes
    uint64_t kk[kBhMergePer];
    int vv[kBhMergePer];
#pragma unroll
    for (int u = 0; u < kBhMergePer; ++u) {
        const int t = t0 + u * kBhBlock + tid;
        kk[u] = t < n ? keys[t] : 0;
        vv[u] = t < n ? vals[t] : 0;
    }
    if (tid < 2 && t0 < p1) {
        const uint64_t k = keys[tid == 0 ? t0 : p1 - 1];
        const int ghosts = ctr->ghosts;                        // (behind them the buffer is padding, which sorts last)
        int lo = 0, hi = ghosts < gcap ? ghosts : gcap;        // second instances with a smaller key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (gkeys[mid] < k) lo = mid + 1; else hi = mid; }
        s_lo[tid] = lo;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kBhMergePer; ++u) {
        const int t = t0 + u * kBhBlock + tid;
        if (t < n) {
            const uint64_t k = kk[u];
            int lo = s_lo[0], hi = s_lo[1];
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (gkeys[mid] < k) lo = mid + 1; else hi = mid; }
            okeys[t + lo] = k;
            ovals[t + lo] = vv[u];
        } else {
            bh_merge_one(t, keys, vals, n, gkeys, gvals, gcap, okeys, ovals);
        }
    }
}

// 3. internal cells owned by each sorted body --------------------------------------------------------------
__device__ __forceinline__ int bh_count_one(int i, int M, const uint64_t* __restrict__ keys)
{
    if (i >= M) return 0;
    const uint64_t k = keys[i];
    const int lp = i > 0 ? bh_lcp_levels(keys[i - 1], k) : -1;
    const int ln = i + 1 < M ? bh_lcp_levels(k, keys[i + 1]) : -1;
    return ln > lp ? ln - lp : 0;
}

__global__ __launch_bounds__(kBhBlock) void bh_count_kernel(const uint64_t* __restrict__ keys, int n,
                                                            const BhCounters* __restrict__ ctr, int* __restrict__ cnt)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    if (i >= n) return;
    cnt[i] = bh_count_one(i, ctr->valid, keys);
}

__device__ __forceinline__ int bh_lower_bound(const uint64_t* __restrict__ keys, int lo, int hi, uint64_t v)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// first index in [lo, hi) whose key is >= v, galloping from lo (the answer is usually a few bodies away).  The stride
// doubles up to 16 and then grows fourfold: a far answer (the end of a cell near the root) is bracketed in half the
// steps, for two more steps of the binary search that follows.
__device__ __forceinline__ int bh_gallop_lower_bound(const uint64_t* __restrict__ keys, int lo, int hi, uint64_t v)
{
    int a = lo, step = 1;
    while (a + step <= hi && keys[a + step - 1] < v) { a += step; step <<= (step >= 16 ? 2 : 1); }
    const int e = a + step < hi ? a + step : hi;
    return bh_lower_bound(keys, a, e, v);
}

// 4. nodes ---------------------------------------------------------------------------------------------------
// 4a. owner of every internal cell: sorted instance i owns the cnt[i] cells base[i].. (the cells whose first body
// it is), at depths lcp(i-1,i)+1...  Scattering the owner once replaces a 24-step binary search per cell.
__global__ __launch_bounds__(kBhBlock) void bh_owner_kernel(const int* __restrict__ cnt, const int* __restrict__ base, int n,
                                                            const BhCounters* __restrict__ ctr, int* __restrict__ owner,
                                                            int cap_internal)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    if (i >= n || i >= ctr->valid) return;
    const int ci = cnt[i], c0 = base[i];
    for (int k = 0; k < ci; ++k)
        if (c0 + k < cap_internal) owner[c0 + k] = i;
}

// 3 + 4a in ONE launch (large builds; was bh_count_kernel + a library scan + bh_owner_kernel): cells owned by each sorted
// instance, their exclusive scan = cell ids, and the owner scatter.  A workgroup takes the next tile of 256 x R
// instances by ticket; each wave walks R rounds of 64 consecutive instances (neighbours' keys by shuffle, as
// bh_prep_small_kernel does), the tile total is published and the totals of all tiles before it are collected by
// decoupled look-back — one 64-bit word per tile {epoch, flag, count}, device-scope relaxed atomics, 64 predecessor
// tiles examined per step by wave 0.  Waits are bounded as in bh_sort.hip.h.  The last workgroup to finish resets the
// ticket and opens the next epoch.
struct ScanState { unsigned epoch; int ticket; int done; int error; };
constexpr int kBhScanRounds = 8;             // rounds of 64 instances per wave: tile = 256 x R.  Large builds use 16 (half the tiles for the
constexpr int kBhScanRoundsLarge = 16;       // look-back to cross: 2^24 229 -> 181 us, 4 x 10^6 67 -> 54; 10^6 22.2 -> 23.8, 32: slower everywhere)
constexpr int kBhScanLargeFrom = 1 << 21;    // instances from which the large tile is used

template <int R>
__global__ __launch_bounds__(kBhBlock) void bh_cells_scan_kernel(const uint64_t* __restrict__ keys, int nm,
                                                                 const BhCounters* __restrict__ ctr, ScanState* __restrict__ ss,
                                                                 unsigned long long* __restrict__ look, int* __restrict__ cnt,
                                                                 int* __restrict__ base, int* __restrict__ owner, int cap_internal)
{
    constexpr int T = kBhBlock * R;
    __shared__ int s_tile;
    __shared__ int wave_total[kBhBlock / 64];
    __shared__ int s_excl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned epoch = ss->epoch;
    const int M = ctr->valid;
    if (tid == 0) s_tile = atomicAdd(&ss->ticket, 1);
    __syncthreads();
    const int tile = s_tile;
    if (tile == 0 && tid == 0) ss->error = 0;       // the flag describes THIS scan (a wait gives up after ~0.3 s: long after this)
    if ((long long)tile * T < nm) {
        const int w0 = tile * T + wave * (64 * R);
        uint64_t kc[R], ke[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = w0 + r * 64 + lane;
            kc[r] = i < M ? keys[i] : 0;
            const int j = lane == 0 ? w0 + r * 64 - 1 : w0 + r * 64 + 64;      // lane 0: the key before the round, lane 63: after
            ke[r] = (lane == 0 || lane == 63) && j >= 0 && j < M ? keys[j] : 0;
        }
        int c[R], incl[R], running = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = w0 + r * 64 + lane;
            uint64_t kp = __shfl_up(kc[r], 1, 64), kn = __shfl_down(kc[r], 1, 64);
            if (lane == 0) kp = ke[r];
            if (lane == 63) kn = ke[r];
            const int lp = i > 0 && i < M ? bh_lcp_levels(kp, kc[r]) : -1;
            const int ln = i + 1 < M ? bh_lcp_levels(kc[r], kn) : -1;
            c[r] = i < M && ln > lp ? ln - lp : 0;                             // = bh_count_one(i, M, keys)
            int v = c[r];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(v, off, 64);
                if (lane >= off) v += t;
            }
            incl[r] = running + v;
            running += __shfl(v, 63, 64);
        }
        if (lane == 0) wave_total[wave] = running;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kBhBlock / 64; ++w) { const int p = wave_total[w]; if (w < wave) before += p; total += p; }
        if (wave == 0) {
            const unsigned long long tagw = (unsigned long long)epoch << 32;
            if (lane == 0) sort_look_store(look + tile, tagw | (tile == 0 ? kLookPrefix : kLookAgg) | (unsigned)total);
            // (Tried, round 3: 4 x 64 predecessor words per look-back step instead of 64 — no gain, 188 against 181 us at 2^24: the
            // look-back is not what bounds this kernel; fewer, larger tiles are what helped.)
            int excl = 0, at = tile - 1, spins = 0;
            while (at >= 0) {
                const int j = at - lane;
                const unsigned long long w = j >= 0 ? sort_look_load(look + j) : (tagw | kLookPrefix);   // before tile 0: nothing
                const bool ready = (unsigned)(w >> 32) == epoch && (w & (kLookAgg | kLookPrefix));
                const unsigned long long not_ready = __ballot(!ready), prefix = __ballot(ready && (w & kLookPrefix));
                const int first_nr = not_ready ? __builtin_ctzll(not_ready) : 64, first_pf = prefix ? __builtin_ctzll(prefix) : 64;
                const int take = first_pf < first_nr ? first_pf + 1 : first_nr;       // lanes [0, take) are summed
                int v = lane < take ? (int)(w & kLookCount) : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl += v;
                if (first_pf < first_nr) break;
                at -= take;
                if (take == 0) {
                    if (++spins > kSortSpinLimit) { if (lane == 0) ss->error = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (lane == 0) {
                if (tile > 0) sort_look_store(look + tile, tagw | kLookPrefix | (unsigned)(excl + total));
                s_excl = excl;
            }
        }
        __syncthreads();
        const int offset = s_excl + before;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = w0 + r * 64 + lane;
            if (i < nm) {
                const int first = offset + incl[r] - c[r];
                cnt[i] = c[r];
                base[i] = first;
                for (int q = 0; q < c[r]; ++q)
                    if (first + q < cap_internal) owner[first + q] = i;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int before = atomicAdd(&ss->done, 1);
        if (before == (int)gridDim.x - 1) { ss->done = 0; ss->ticket = 0; ss->epoch = epoch + 1; }
    }
}

// Small N: steps 2b-4a in ONE launch of one workgroup (valid count, split-line instances and their fixed point, their
// sort — bitonic in LDS, nothing to do when there are none —, merge, cell counting, exclusive scan, owner scatter).
// At N = 10 000 these were 12 launches of 4-5 us each around a few hundred nanoseconds of work (the reference's own
// default size, benchmark.cpp:302; a captured graph replays launches no faster than the stream issues them).
// Same device functions as the separate kernels, phases separated by workgroup barriers; results are identical.
constexpr int kBhSmallMax = 14336;                  // bodies; above, the phases are separate full-width launches (measured at the end
                                                    // of round 3, small / large build us: 8192 78 / 103, 12 288 101 / 109, 16 384 131 / 112,
                                                    // 24 576 151 / 114, 32 768 217 / 133 — it was 32 768 before the large build's rewrite)

__global__ __launch_bounds__(kBhFixBlock) void bh_prep_small_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                                    const float* __restrict__ px, const float* __restrict__ py,
                                                                    const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                                    BhTies T, int n, uint64_t* __restrict__ gkeys_sorted,
                                                                    int* __restrict__ gvals_sorted, uint64_t* __restrict__ okeys,
                                                                    int* __restrict__ ovals, int* __restrict__ cnt,
                                                                    int* __restrict__ base, int* __restrict__ owner, int cap_internal,
                                                                    const unsigned char* __restrict__ tie_first_sorted)
{
    __shared__ uint64_t sk[kBhSmallGhosts];
    __shared__ int sv[kBhSmallGhosts];
    __shared__ int part[kBhFixBlock / 64];
    const int tid = threadIdx.x;
    BH_DBG_TICK(ctr, 0);
    const int M0 = bh_valid_total(ctr);                            // counted by bh_keys_kernel
    // Every wave takes one contiguous sixteenth of the sorted instances, 64 consecutive ones per round, 10 rounds of
    // loads in flight (the neighbours' keys come by shuffle; the two at the ends of a round are loaded).
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kRounds = 10;
    // one round of a pass: this lane's key and the common-prefix lengths with its neighbours
    auto load_round = [&](const uint64_t* mk, int M, int r0, int w1, uint64_t& kc, uint64_t& kedge) {
        const int i = r0 + lane;
        kc = i < w1 ? mk[i] : 0;
        const int j = lane == 0 ? r0 - 1 : r0 + 64;                // lane 0: the key before the round, lane 63: after
        kedge = (lane == 0 || lane == 63) && j >= 0 && j < M ? mk[j] : 0;
    };
    auto cells_of = [&](int M, int r0, int w1, uint64_t kc, uint64_t kedge, int& lp, int& ln) -> int {
        const int i = r0 + lane;
        uint64_t kp = __shfl_up(kc, 1, 64), kn = __shfl_down(kc, 1, 64);
        if (lane == 0) kp = kedge;
        if (lane == 63) kn = kedge;
        lp = i > 0 && i < M ? bh_lcp_levels(kp, kc) : -1;
        ln = i + 1 < M ? bh_lcp_levels(kc, kn) : -1;
        return i < w1 && ln > lp ? ln - lp : 0;
    };

    // 2b. second instances of bodies on split lines, and — the same keys, the same prefix lengths — the number of cells
    // each wave's primaries own (pass 0 of step 3, valid as it is when there turn out to be no second instances, i.e.
    // nearly always).  A primary spawns instances only where it sits on a split line ABOVE its leaf: first tie level <
    // leaf depth = 1 + max(lcp with either neighbour).  Far down, where cells are narrower than a float ulp, almost
    // every body "sits on a line", so the test against the leaf depth is what makes the expensive path rare.
    const int per0 = (((M0 + kBhFixBlock / 64 - 1) / (kBhFixBlock / 64)) + 63) & ~63;
    const int a0 = wave * per0, a1 = a0 + per0 < M0 ? a0 + per0 : M0;
    int total = 0;
    for (int i0 = a0; i0 < a1; i0 += kRounds * 64) {
        uint64_t kc[kRounds], ke[kRounds];
        int first[kRounds];
#pragma unroll
        for (int u = 0; u < kRounds; ++u) {
            load_round(keys, M0, i0 + u * 64, a1, kc[u], ke[u]);
            const int i = i0 + u * 64 + lane;
            first[u] = i < a1 ? (tie_first_sorted ? (int)tie_first_sorted[i] : vals[i]) : kBhLevels;
        }
        if (tie_first_sorted == nullptr) {
#pragma unroll
            for (int u = 0; u < kRounds; ++u)
                if (i0 + u * 64 + lane < a1) first[u] = bh_first_tie(tie[first[u]]);
        }
#pragma unroll
        for (int u = 0; u < kRounds; ++u) {
            if (i0 + u * 64 < a1) {                                // wave-uniform
                int lp, ln;
                total += cells_of(M0, i0 + u * 64, a1, kc[u], ke[u], lp, ln);
                const int leaf_depth = 1 + (lp > ln ? lp : ln);
                // the fixed-point phase raises T.pld of ANY primary with a tie record when a second instance deepens
                // its leaf: it has to hold the leaf depth for all of them, not only for those that spawn now
                if (first[u] != kBhLevels) T.pld[i0 + u * 64 + lane] = leaf_depth;
                if (first[u] < leaf_depth) bh_ties_one(i0 + u * 64 + lane, M0, keys, vals, px, py, tie, ctr, T);
            }
        }
    }
    __syncthreads();            // one workgroup = one CU = one L1: a barrier is all the ordering the phases need
    BH_DBG_TICK(ctr, 1);
    // (M0 by value: ctr->valid is only written further down, by one thread; reading it here would race with that)
    bh_tie_fixpoint_body(keys, vals, px, py, tie, ctr, T, M0);
    BH_DBG_TICK(ctr, 2);
    const int ghosts = __hip_atomic_load(&ctr->ghosts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int gmax = T.gcap < kBhSmallGhosts ? T.gcap : kBhSmallGhosts;
    const int G = ghosts < gmax ? ghosts : gmax;
    const uint64_t* mk = keys;                                     // the merged arrays ARE the sorted primaries ...
    if (G > 0) {                                                   // ... unless there are second instances (rare):
        __syncthreads();                                           // sort them by key, bitonic, in LDS, and merge
        int P = 2;
        while (P < G) P <<= 1;
        for (int e = tid; e < P; e += kBhFixBlock) { sk[e] = e < G ? T.gkeys[e] : ~0ull; sv[e] = e < G ? T.gvals[e] : -1; }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int p = tid; p < (P >> 1); p += kBhFixBlock) {
                    const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1)), hi = lo | j;
                    const uint64_t ka = sk[lo], kb = sk[hi];
                    const int va = sv[lo], vb = sv[hi];
                    // total order (key, body): equal keys of different bodies come out the same way every time
                    const bool gt = ka > kb || (ka == kb && va > vb);
                    if (gt == ((lo & k) == 0)) { sk[lo] = kb; sk[hi] = ka; sv[lo] = vb; sv[hi] = va; }
                }
                __syncthreads();
            }
        for (int e = tid; e < G; e += kBhFixBlock) { gkeys_sorted[e] = sk[e]; gvals_sorted[e] = sv[e]; }
        __syncthreads();
        for (int t = tid; t < n + G; t += kBhFixBlock) bh_merge_one(t, keys, vals, n, gkeys_sorted, gvals_sorted, G, okeys, ovals);
        __syncthreads();
        mk = okeys;
    }
    const int M = M0 + G;
    BH_DBG_TICK(ctr, 3);
    if (tid == 0) ctr->valid = M;
    // 3. cells owned by each instance, their exclusive scan (= cell ids) and 4a. the owner of every cell: a wave's
    // total first (pass 0: done above unless the merge changed the array), then — one LDS barrier later, when it knows
    // the totals of the waves before it — the scan itself (shuffle scan per round, running offset in a register) and
    // the stores.  (Round 2's first version synchronised all 16 waves twice per 1024 instances.)
    const int per = (((M + kBhFixBlock / 64 - 1) / (kBhFixBlock / 64)) + 63) & ~63;
    const int w0 = wave * per, w1 = w0 + per < M ? w0 + per : M;
    if (G > 0) {
        total = 0;
        for (int i0 = w0; i0 < w1; i0 += kRounds * 64) {
            uint64_t kc[kRounds], ke[kRounds];
#pragma unroll
            for (int u = 0; u < kRounds; ++u) load_round(mk, M, i0 + u * 64, w1, kc[u], ke[u]);
#pragma unroll
            for (int u = 0; u < kRounds; ++u)
                if (i0 + u * 64 < w1) { int lp, ln; total += cells_of(M, i0 + u * 64, w1, kc[u], ke[u], lp, ln); }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0) part[wave] = total;
    __syncthreads();
    int running = 0;
    for (int w2 = 0; w2 < wave; ++w2) running += part[w2];
    for (int i0 = w0; i0 < w1; i0 += kRounds * 64) {
        uint64_t kc[kRounds], ke[kRounds];
#pragma unroll
        for (int u = 0; u < kRounds; ++u) load_round(mk, M, i0 + u * 64, w1, kc[u], ke[u]);
#pragma unroll
        for (int u = 0; u < kRounds; ++u) {
            if (i0 + u * 64 < w1) {                                // wave-uniform
                int lp, ln;
                const int c = cells_of(M, i0 + u * 64, w1, kc[u], ke[u], lp, ln);
                int incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += v;
                }
                const int i = i0 + u * 64 + lane;
                if (i < w1) {
                    const int first = running + incl - c;
                    cnt[i] = c;
                    base[i] = first;
                    for (int k = 0; k < c; ++k)
                        if (first + k < cap_internal) owner[first + k] = i;
                }
                running += __shfl(incl, 63, 64);
            }
        }
    }
    BH_DBG_TICK(ctr, 6);
}

#ifndef NBODY_NODES_LDS_STORES
#define NBODY_NODES_LDS_STORES 1
#endif
constexpr int kBhNodesSmallSpan = 64;         // instances up to which a cell's quadrant boundaries are found by galloping
constexpr int kBhNodesWindow = 8;             // keys behind the owner that bh_nodes_kernel loads at once
struct BhWaitErrors { const int* p[3]; };      // SortState::error / ScanState::error of the kernels before (nullptr: not used)

// 4b. one thread per internal cell: box, body range (galloping from the owner: the range of a typical cell is a
// few bodies long), the four children.
__global__ __launch_bounds__(kBhBlock) void bh_nodes_kernel(const uint64_t* keys, const int* vals,
                                                            const int* __restrict__ cnt, const int* __restrict__ base,
                                                            const int* __restrict__ owner, int n,
                                                            const float* __restrict__ m, const float* __restrict__ px,
                                                            const float* __restrict__ py, BhTree t,
                                                            BhCounters* __restrict__ ctr, int* __restrict__ overflow_host,
                                                            const uint64_t* __restrict__ keys_unmerged,
                                                            const int* __restrict__ vals_unmerged,
                                                            const float4* __restrict__ body4, BhWaitErrors werr)
{
    // small-N build: without second instances the merged arrays were never written, they ARE the sorted primaries
    if (keys_unmerged != nullptr && ctr->ghosts == 0) { keys = keys_unmerged; vals = vals_unmerged; }
    const int c = blockIdx.x * kBhBlock + threadIdx.x;
    const int M = ctr->valid;
    const float spread = ctr->root_half;
    const int I = M >= 2 ? base[M - 1] + cnt[M - 1] : 0;
    if (c == 0) {
        ctr->internal = I < t.cap_internal ? I : t.cap_internal;
        if (I > t.cap_internal) ctr->overflow = 1;
        // the build's verdict goes to a host-mapped word, so that nbody_sync needs no device-to-host copy (the tie
        // buffer's overflow, 2, was set by the kernels before this one)
        int verdict = I > t.cap_internal ? 1 : ctr->overflow;
        for (int k = 0; k < 3; ++k)
            if (werr.p[k] && *werr.p[k]) verdict = 3;          // a bounded look-back wait of the sort / scan gave up
        overflow_host[0] = verdict;
        for (int k = 0; k < 64; ++k) ctr->valid_part[k] = 0;            // everybody has read ctr->valid by now
        if (M < 2) {                                           // degenerate trees: root is a leaf / empty
            const int b = M == 1 ? vals[0] : -1;
            const float4 p = b >= 0 ? body4[b] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            t.A[0] = make_float4(p.x, p.y, p.z, 2.0f * spread);
            t.B[0] = make_int2(-1, b);
        }
    }
    const int limit = I < t.cap_internal ? I : t.cap_internal;
#if NBODY_NODES_LDS_STORES
    // The four child records of a cell are 64 + 32 contiguous bytes, but a lane that stores its own cell's writes 16
    // (8) bytes at a stride of 64 (32): every store instruction touches 64 lines partially.  Staged through LDS, a
    // wave's 64 cells go out as 4 + 4 fully coalesced stores.
    __shared__ float4 sA[kBhBlock / 64][256];                  // 4 KB per wave: first the float4 records, then (reused) the int2 ones
    int2 bq[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c_wave0 = c - lane;
    if (c_wave0 >= limit) return;                              // wave-uniform
    if (c < limit) {
#else
    if (c >= limit) return;
    {
#endif
    // The kernel is a chain of dependent loads under full occupancy, so the chain is kept short: everything a step needs
    // is requested at once, before anything is stored (keys/vals are not __restrict__: a store in between would pin the
    // loads behind it).  Round trips of a typical cell: owner -> {keys[i-1 .. i+8], base[i]} -> {keys, base, vals at the
    // four quadrant starts} -> the leaves' bodies -> stores.  (Round 3; before: ~20 — three galloping searches one after
    // the other, then the children one by one.)
    const int i = owner[c];
    const uint64_t ki = keys[i];
    const uint64_t kprev = i > 0 ? keys[i - 1] : 0;
    const int base_i = base[i];
    uint64_t kw[kBhNodesWindow];                               // the keys behind the owner; past the array: behind every cell
#pragma unroll
    for (int u = 0; u < kBhNodesWindow; ++u) kw[u] = i + 1 + u < M ? keys[i + 1 + u] : ~0ull;
    const int lp = i > 0 ? bh_lcp_levels(kprev, ki) : -1;
    const int d = lp + 1 + (c - base_i);                       // depth of this cell, 0..30
    const int shift = 2 * (kBhLevels - d);
    const uint64_t prefix = d ? ki >> shift : 0;

    // box of the cell: replay the reference's midpoint arithmetic along the prefix
    float tlx = -spread, tly = -spread, brx = spread, bry = spread;
    for (int l = 0; l < d; ++l) {
        const int q = (int)((ki >> (2 * (kBhLevels - 1 - l))) & 3);
        const float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;
        if (q & 1) tlx = mx; else brx = mx;
        if (q & 2) tly = my; else bry = my;
    }

    // body range [i, e) of the cell and its quadrant boundaries
    int e = M;
    if (d) {
        const uint64_t bound = (prefix + 1) << shift;
        int inside = 0;                                        // sorted: the members among kw[] are its first `inside` entries
#pragma unroll
        for (int u = 0; u < kBhNodesWindow; ++u) inside += kw[u] < bound ? 1 : 0;
        e = inside < kBhNodesWindow ? i + 1 + inside : bh_gallop_lower_bound(keys, i + 1 + kBhNodesWindow, M, bound);
    }
    const int cshift = shift - 2;
    int s[5];
    s[0] = i;
    s[4] = e;
    if (e - i <= kBhNodesWindow + 1) {
        // the whole cell is in registers (nearly every cell: a typical one spans a few bodies)
        int below[4] = {0, 0, 0, 0};
        {
            const int dg = (int)((ki >> cshift) & 3);
#pragma unroll
            for (int q = 1; q < 4; ++q) below[q] += dg < q ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < kBhNodesWindow; ++u) {
            const int dg = (int)((kw[u] >> cshift) & 3);
#pragma unroll
            for (int q = 1; q < 4; ++q) below[q] += (u < e - i - 1 && dg < q) ? 1 : 0;
        }
#pragma unroll
        for (int q = 1; q < 4; ++q) s[q] = i + below[q];
    } else if (e - i <= kBhNodesSmallSpan) {
#pragma unroll
        for (int q = 1; q < 4; ++q) s[q] = bh_gallop_lower_bound(keys, s[q - 1], e, ((prefix << 2) | (uint64_t)q) << cshift);
    } else {
        // A cell near the root spans 10^4-10^6 instances; three galloping searches, each starting where the one before
        // ended, are ~100 dependent loads there.  Three binary searches over [i, e) side by side: their loads are
        // independent, the chain is log2(span).
        int lo[4], hi[4];
        uint64_t v[4];
#pragma unroll
        for (int q = 1; q < 4; ++q) { lo[q] = i; hi[q] = e; v[q] = ((prefix << 2) | (uint64_t)q) << cshift; }
        while (lo[1] < hi[1] || lo[2] < hi[2] || lo[3] < hi[3]) {
            uint64_t km[4];
#pragma unroll
            for (int q = 1; q < 4; ++q) km[q] = lo[q] < hi[q] ? keys[(lo[q] + hi[q]) >> 1] : 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (lo[q] < hi[q]) { const int mid = (lo[q] + hi[q]) >> 1; if (km[q] < v[q]) lo[q] = mid + 1; else hi[q] = mid; }
        }
#pragma unroll
        for (int q = 1; q < 4; ++q) s[q] = lo[q];
    }

    // (Tried, round 3: every cell computing its OWN node index — parent = c - 1, or, for an owner's shallowest cell, found
    // through a window of the 8 keys before the owner — and storing it coalesced, instead of the parent's scattered
    // t.self[child] = node below, which the counters suggested as a third of the kernel's traffic.  Correct (all tree tests),
    // and slower: 65 -> 85 us at N = 10^6, 1067 -> 1345 at 2^24 — nine more 64-bit registers and loads per cell cost more
    // occupancy than the scattered stores cost bandwidth.)
    // what the four children need, requested together (an empty quadrant at the end of the array reads entry M - 1)
    uint64_t ka[4], kb[4];
    int bs[4], vl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = s[q] < M - 1 ? s[q] : M - 1;
        ka[q] = j > 0 ? keys[j - 1] : 0;
        kb[q] = keys[j];
        bs[q] = base[j];
        vl[q] = vals[j];
    }
    int child_base[4], body[4];
    int internal_children = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cntq = s[q + 1] - s[q];
        child_base[q] = -1;
        body[q] = -1;
        if (cntq >= 2 && d + 1 < kBhLevels) {
            const int lpq = s[q] > 0 ? bh_lcp_levels(ka[q], kb[q]) : -1;
            const int cc = bs[q] + d - lpq;                    // id of the internal child
            if (cc < t.cap_internal) { child_base[q] = 1 + 4 * cc; ++internal_children; }
            else body[q] = vl[q];                              // pool exhausted: keep one body (benchmark.cpp:102)
        } else if (cntq >= 1) {
            body[q] = vl[q];                                   // leaf
        }
    }
    float4 pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pb[q] = body4[body[q] >= 0 ? body[q] : vl[q]];      // (always a valid index; unused when there is no body)

    if (c == 0) {
        t.A[0] = make_float4(0.0f, 0.0f, 0.0f, brx - tlx);
        t.B[0] = make_int2(1, -1);
        t.self[0] = 0;
    }
    t.depth[c] = (uint8_t)d;
    const float mx = (tlx + brx) / 2.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int node = 1 + 4 * c + q;
        const float ctlx = (q & 1) ? mx : tlx, cbrx = (q & 1) ? brx : mx;
        const float width = cbrx - ctlx;                       // benchmark.cpp:251
        const int cntq = s[q + 1] - s[q];
        float cx = 0.0f, cy = 0.0f, cm = 0.0f;
        if (child_base[q] >= 0) t.self[(child_base[q] - 1) >> 2] = node;
        if (body[q] >= 0) { cx = pb[q].x; cy = pb[q].y; cm = pb[q].z; }   // benchmark.cpp:89-92
        if (child_base[q] < 0 && cntq >= 2) {
            // Several bodies in one cell that is not split further: closer together than the finest cell (the path
            // key has 31 levels: 800/2^31 = 3.7e-7 at spread 400) or the pool is exhausted.  The reference keeps
            // splitting such a cell (to its depth ~40-150, for coincident bodies until ITS pool is exhausted,
            // benchmark.cpp:102); the subtree it gets has this total mass at this centre of mass, and bodies that
            // close are indistinguishable for every other target, so the leaf stands for all of them (the self mask
            // covers the first one only).
            float ms = 0.0f, sx = 0.0f, sy = 0.0f;
            for (int j = s[q]; j < s[q + 1]; ++j) {
                const int bj = vals[j];
                if (j > s[q] && bj == vals[j - 1]) continue;    // two instances of one body
                const float4 p = body4[bj];
                const float mj = p.z;
                ms += mj;
                sx = __builtin_fmaf(p.x, mj, sx);
                sy = __builtin_fmaf(p.y, mj, sy);
            }
            cm = ms;
            if (ms > 0.0f) { cx = sx / ms; cy = sy / ms; }
        }
        const int child_base_q = child_base[q], body_q = body[q];
#if NBODY_NODES_LDS_STORES
        sA[wv][4 * lane + q] = make_float4(cx, cy, cm, width);
        bq[q] = make_int2(child_base_q, body_q);
#else
        t.A[node] = make_float4(cx, cy, cm, width);
        t.B[node] = make_int2(child_base_q, body_q);
#endif
    }
    t.pend[c] = internal_children ? internal_children : -1;
    }
#if NBODY_NODES_LDS_STORES
    // (one wave, its own LDS rows: LDS executes a wave's instructions in order; the fences are for the compiler)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int records = 4 * (limit - c_wave0 < 64 ? limit - c_wave0 : 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = r * 64 + lane;
        if (idx < records) t.A[1 + 4 * c_wave0 + idx] = sA[wv][idx];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int2* sB = reinterpret_cast<int2*>(&sA[wv][0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) sB[4 * lane + q] = bq[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = r * 64 + lane;
        if (idx < records) t.B[1 + 4 * c_wave0 + idx] = sB[idx];
    }
#endif
}

// Read-back only (nbody_bh_tree_nodes): the box {tlX, tlY, brX, brY} of every node, recomputed from the topology —
// climb to the root collecting quadrant digits, then replay the reference's midpoint arithmetic downwards.  The
// step itself never needs boxes (the walk reads the width from A.w), so they are not stored by the build.
__global__ __launch_bounds__(kBhBlock) void bh_box_kernel(BhTree t, const BhCounters* __restrict__ ctr, float4* __restrict__ box,
                                                          int64_t count)
{
    const int64_t node = (int64_t)blockIdx.x * kBhBlock + threadIdx.x;
    if (node >= count) return;
    uint64_t path = 0;
    int depth = 0;
    for (int64_t v = node; v != 0 && depth < kBhLevels + 1; ++depth) {
        path |= (uint64_t)((v - 1) & 3) << (2 * depth);       // digit of the LAST step first
        v = t.self[(v - 1) >> 2];
    }
    const float h = ctr->root_half;
    float tlx = -h, tly = -h, brx = h, bry = h;
    for (int l = depth - 1; l >= 0; --l) {
        const int q = (int)((path >> (2 * l)) & 3);
        const float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;
        if (q & 1) tlx = mx; else brx = mx;
        if (q & 2) tly = my; else bry = my;
    }
    box[node] = make_float4(tlx, tly, brx, bry);
}

// 5. centre of mass -----------------------------------------------------------------------------------------
__device__ __forceinline__ void bh_com_cell(const BhTree& t, int c)
{
    const int self = t.self[c];
    float ms = 0.0f, sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // benchmark.cpp:126-135 in child order; FMAs as the reference's build contracts them
        const float4 a = t.A[1 + 4 * c + q];
        ms = ms + a.z;
        sx = __builtin_fmaf(a.x, a.z, sx);
        sy = __builtin_fmaf(a.y, a.z, sy);
    }
    // benchmark.cpp:137-141 (IEEE division; an all-massless cell keeps the (0,0) bh_nodes_kernel gave it).  Only
    // {comX, comY, mass} are written: the width in .w stays, so the record is not read first (one round trip less
    // per level of the bottom-up chain).
    float* rec = reinterpret_cast<float*>(t.A + self);
    float cx = 0.0f, cy = 0.0f;
    if (ms > 0.0f) {
        if (t.com_reciprocal) { const float inv = 1.0f / ms; cx = sx * inv; cy = sy * inv; }
        else                  { cx = sx / ms; cy = sy / ms; }
    }
    rec[0] = cx; rec[1] = cy; rec[2] = ms;
}

// One launch per depth, deepest first (trees of more than 2^21 bodies).  One thread per 4 W cells (W 32-bit loads of
// their depth bytes, issued together).  Measured at N = 2^24 (12 M cells): a launch that finds nothing costs its waves
// (8.2 us with W = 1, 4.5 with W = 4), a populated one its memory traffic — 3 M cells in 177 us, ~3.4 TB/s of 64-byte
// child blocks read and 12-byte records written — and is SLOWER with fatter threads (W = 4: +8 % over all levels).
// So the host uses W = 4 for the depths that hold few cells on any input (the top of the tree and cells narrower than
// 800 / 2^24) and W = 1 between.
template <int W>
__global__ __launch_bounds__(kBhBlock) void bh_com_kernel(BhTree t, const BhCounters* __restrict__ ctr, int d)
{
    const int I = ctr->internal;
    const int c0 = (blockIdx.x * kBhBlock + threadIdx.x) * 4 * W;
    if (c0 >= I) return;
    unsigned x[W];
#pragma unroll
    for (int w = 0; w < W; ++w) x[w] = reinterpret_cast<const unsigned*>(t.depth + c0)[w] ^ (0x01010101u * (unsigned)d);   // (16 bytes of slack behind the array)
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (((x[w] - 0x01010101u) & ~x[w] & 0x80808080u) == 0) continue;    // no zero byte = no cell of depth d here
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (((x[w] >> (8 * k)) & 0xFFu) == 0 && c0 + 4 * w + k < I) bh_com_cell(t, c0 + 4 * w + k);
    }
}

// All depths in ONE launch, no level barrier: a cell without internal children is evaluated at once; its thread then
// counts itself off at the parent (one atomic on the parent's counter of unfinished internal children, set by
// bh_nodes_kernel), and whoever arrives LAST evaluates the parent and climbs on.  Every cell is evaluated exactly once,
// by one thread, from finished children in the fixed child order: the bits do not depend on who does it.  No thread
// waits for another (nothing to hang on); a thread's chain is at most the tree's depth.  ~2 us per level of the
// critical path instead of one 4.5-us launch per depth: N = 10 000 55 (one workgroup, results in LDS: round 2's first
// version) -> 22 us, N = 10^6 143 (31 launches) -> 109 us; above ~2 x 10^6 bodies the per-depth launches win again
// (their loads hit the L2), so the host switches there.
//
// Visibility across workgroups, which may sit on different XCDs (one L2 each).  The hand-off is, per child:
//   (1) the child's record {comX, comY, mass, width} is written ONLY by `global_store_dwordx4 ... sc1` — the instruction
//       gfx950 emits for an agent-scope relaxed atomic store, in its 16-byte form (rounds 2-3a: an 8- and a 4-byte atomic
//       store): it writes through the XCD's L2 and leaves no line behind in it;
//   (2) the storing wave waits for them (s_waitcnt vmcnt(0)) BEFORE (3) it issues the count-off, an agent-scope
//       atomic read-modify-write on the parent's counter;
//   (4) the lane whose atomic returned 1 — and nobody else — reads the four children's records, AFTER that return,
//       ONLY with agent-scope relaxed atomic loads (`global_load ... sc1`: served below the L1, never from it).
// Why this is enough, on paper:
//   * LLVM AMDGPU memory model (AMDGPUUsage, "Memory Model", gfx940/gfx942 table, which gfx950 shares): an agent-scope
//     RELEASE is `buffer_wbl2 sc1` + `s_waitcnt vmcnt(0)` before the atomic — the write-back exists to push ORDINARY
//     (non-sc1) dirty lines of the XCD's L2 out to where other XCDs can see them; the waitcnt orders earlier stores
//     before the atomic.  An agent-scope ACQUIRE is the atomic + `s_waitcnt vmcnt(0)` + `buffer_inv sc1` — the
//     invalidate exists so that LATER ORDINARY loads do not hit stale L1/L2 lines.  Steps (2)/(3) are the release
//     without the write-back, steps (3)/(4) the acquire without the invalidate; both omissions are sound exactly
//     because no ordinary store (1) or ordinary load (4) ever touches the handed-off bytes inside this launch.  Every
//     other word the climb reads (t.self, child records of leaf-only cells) was written by the PREVIOUS kernel.
//   * /opt/skills/guides/MI355X_MICROARCH.md, "Valid forms besides Guideline 16's R1/R2", Consumer bullet: `sc1` loads
//     to registers may replace the acquire when (1) EVERY load of the handed-off bytes is such a load, (2) the producer
//     stored every one of those bytes `sc1`, (3) every storing wave ran s_waitcnt vmcnt(0) after its stores and the
//     counter add comes after that wait, (4) the hand-off matches the measured table: its first row — "ONE lane ...
//     an agent-scope atomic add ... all those lanes add to ONE unsharded counter, the workgroup whose add came last,
//     told by the value its add returned; the wave that added last loads only after its add has returned; 4-, 8- or
//     16-B sc1 stores and loads" — is this protocol with a lane signalling for its own stores only.  The guide marks the
//     table "measured on gfx950 / ROCm 7.2, not an architectural guarantee"; so the result is also CHECKED: the
//     determinism test forces every parent and its children onto different workgroups (kScatter below: consecutive
//     cells go to consecutive workgroups, which the dispatcher deals round-robin over the 8 XCDs) and compares all
//     nodes bit for bit with the per-depth kernel, whose hand-offs are kernel boundaries
//     (tests/test_gpu_bh.py::test_centre_of_mass_pass_is_deterministic_over_repeated_builds).
// With the two cache-wide operations (a plain __threadfence() on either side of the atomic) the N = 10^6 build takes
// 3.2 ms instead of 0.52: every fence writes back / invalidates the whole L2 for everybody.  The per-depth kernel
// stays in the tree as the fallback (NBODY_BH_CLIMB_MAX=0).
// kAgentLoads = false: the cell's children are all leaves, written by the PREVIOUS kernel — ordinary (cached) loads do.
#ifndef NBODY_CLIMB_WIDE
#define NBODY_CLIMB_WIDE 1
#endif
template <bool kAgentLoads>
__device__ __forceinline__ void bh_com_cell_agent(const BhTree& t, int c, int self)        // self = t.self[c]
{
#if NBODY_CLIMB_WIDE
    // One 16-byte device-scope access per record instead of an 8- and a 4-byte one: the climb beyond the smallest sizes
    // is bound by the NUMBER of such operations (they are served below the XCD's L2).  The compiler has no 16-byte
    // atomic load/store, so the instructions are written out: global_load/store_dwordx4 with sc1 = what it emits for
    // an agent-scope relaxed atomic of 4 or 8 bytes.  Nobody reads a record while it is written (the count-off orders
    // them), so the access needs no single-copy atomicity of its own.  The store is the 12-byte form (dwordx3): the
    // record's fourth float, the cell's width, is the node kernel's and stays (reading it to store it back — the first
    // version — was a scattered 64-byte fetch per cell: 0.8 of the 2.1 GB the climb fetched at N = 2^24).
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f ch[4];
    const float4* kids = t.A + 1 + 4 * c;
    if constexpr (kAgentLoads) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                     "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
                     "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(ch[0]), "=&v"(ch[1]), "=&v"(ch[2]), "=&v"(ch[3]) : "v"(kids) : "memory");
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 a = kids[q]; ch[q] = v4f{a.x, a.y, a.z, a.w}; }
    }
    float ms = 0.0f, sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // as bh_com_cell
        ms = ms + ch[q].z;
        sx = __builtin_fmaf(ch[q].x, ch[q].z, sx);
        sy = __builtin_fmaf(ch[q].y, ch[q].z, sy);
    }
    float cx = 0.0f, cy = 0.0f;
    if (ms > 0.0f) {
        if (t.com_reciprocal) { const float inv = 1.0f / ms; cx = sx * inv; cy = sy * inv; }
        else                  { cx = sx / ms; cy = sy / ms; }
    }
    typedef float v3f __attribute__((ext_vector_type(3)));
    const v3f out = {cx, cy, ms};                  // 12 bytes: the record's fourth float, the cell's width, is the node kernel's
    asm volatile("global_store_dwordx3 %0, %1, off sc1" : : "v"(t.A + self), "v"(out) : "memory");
#else
    unsigned long long xy[4];
    unsigned mz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(t.A + 1 + 4 * c + q);
        if constexpr (kAgentLoads) {
            xy[q] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mz[q] = __hip_atomic_load(reinterpret_cast<const unsigned*>(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float4 a = t.A[1 + 4 * c + q];
            xy[q] = (unsigned long long)__float_as_uint(a.x) | ((unsigned long long)__float_as_uint(a.y) << 32);
            mz[q] = __float_as_uint(a.z);
        }
    }
    float ms = 0.0f, sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // as bh_com_cell
        const float ax = __uint_as_float((unsigned)xy[q]), ay = __uint_as_float((unsigned)(xy[q] >> 32)), az = __uint_as_float(mz[q]);
        ms = ms + az;
        sx = __builtin_fmaf(ax, az, sx);
        sy = __builtin_fmaf(ay, az, sy);
    }
    float cx = 0.0f, cy = 0.0f;
    if (ms > 0.0f) {
        if (t.com_reciprocal) { const float inv = 1.0f / ms; cx = sx * inv; cy = sy * inv; }
        else                  { cx = sx / ms; cy = sy / ms; }
    }
    unsigned long long* rec = reinterpret_cast<unsigned long long*>(t.A + self);
    __hip_atomic_store(rec, (unsigned long long)__float_as_uint(cx) | ((unsigned long long)__float_as_uint(cy) << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned*>(rec + 1), __float_as_uint(ms), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// scatter != 0 (test switch NBODY_BH_CLIMB_SCATTER): cell c = thread * workgroups + workgroup instead of workgroup * 256 +
// thread, so that the consecutive ids of a subtree — parent and children — sit in DIFFERENT workgroups on different XCDs.
// (Tried, round 3: the first hop taken by the wave together — leaf-only children of one parent have consecutive ids and sit
// in adjacent lanes, so each run of lanes with the same parent counted off once, with the run's length: a third of the
// atomics.  No gain: 51.6 -> 50.7 us at N = 10^6, 703 -> 704 at 2^24.  The atomics are not what the climb waits for.)
__global__ __launch_bounds__(kBhBlock) void bh_com_climb_kernel(BhTree t, const BhCounters* __restrict__ ctr, int scatter)
{
    const int I = ctr->internal;
    int c = scatter ? (int)(threadIdx.x * gridDim.x + blockIdx.x) : (int)(blockIdx.x * kBhBlock + threadIdx.x);
    if (c >= I) return;
    if (t.pend[c] != -1) return;                   // has internal children: the last of them to finish takes it
    int self = t.self[c];                          // (the node kernel's: ordinary loads)
    for (int hop = 0; hop <= kBhLevels; ++hop) {
        if (hop == 0) bh_com_cell_agent<false>(t, c, self);    // a cell without internal children: only leaves below
        else          bh_com_cell_agent<true>(t, c, self);
        if (c == 0) break;
        const int p = (self - 1) >> 2;
        const int pself = t.self[p];               // for the next hop: in flight while the stores are acknowledged
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // compiler: nothing moves below
        __builtin_amdgcn_s_waitcnt(0);             // the result stores are acknowledged before the count-off is issued
        const int before = __hip_atomic_fetch_sub(&t.pend[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // compiler: nothing moves above
        if (before != 1) break;
        c = p;
        self = pself;
    }
}

// Walk --------------------------------------------------------------------------------------------------------
struct BhWalkArgs {
    const float4* __restrict__ A;
    const int2* __restrict__ B;
    float* px;                       // read at the start of the walk, written at its end when advance != 0:
    float* py;                       // deliberately neither const nor __restrict__
    const float* __restrict__ radius;
    float* __restrict__ vx;
    float* __restrict__ vy;
    float* __restrict__ ax;          // total accelerations [n], indexed by body
    float* __restrict__ ay;
    float dt;
    int advance;                     // 1: forward Euler on the walked bodies (benchmark.cpp:382-401), 0: forces only
    BhCounters* __restrict__ ctr;
    int i_begin, i_end;              // targets; i_begin is a multiple of 8
    float theta;
    int lanes_used;                  // 8 x groups per wave (64, 32, 16 or 8): see bh_walk_groups_per_wave()
    int* __restrict__ grp_visits;    // [n/8 + 1] per group of 8 bodies (index = first body / 8): nodes popped and
    int* __restrict__ grp_accepts;   //           nodes accepted by this walk (parity hook, nbody_bh_group_counters)
    const int* __restrict__ order;   // bh_sorted: walk position -> body (the build's sorted order); nullptr: identity
    int exact_core;                  // 1: masses and radii are in the range exact_core.hip.h needs (checked at upload)
    int done_node;                   // bh_walk_kernel: index of the pool's spare last record ("this group is done")
};

// Groups per wave.  8 (all 64 lanes) is the throughput shape.  With few groups the walk is a latency chain and the
// chip is mostly idle, so each wave then takes fewer groups: more waves in flight and a wave no longer executes the
// accept AND the open AND the pop path of 8 diverging groups in every iteration.
inline int bh_walk_groups_per_wave(int targets)
{
    const int groups = (targets + 7) / 8;
    int g = 8;
    while (g > 1 && groups / g < 1024) g >>= 1;     // one wave per SIMD (256 CUs x 4); more waves than that only
                                                    // multiply the instruction stream (measured: 2 groups/wave at
                                                    // N = 65 536 is 28 % slower than 8)
    return g;
}

// kFast = false: IEEE sqrt and divisions everywhere; every lane reproduces oracle_bh_accel() bit for bit.
// kFast = true (nbody_config.bh_fast_math): the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32.  The VOTE stays the
// oracle's: the approximate ratio decides only when it is further than 4e-6 (relative; > 10x its error bound of
// ~3.5 ulp) from theta, otherwise the wave redoes that node with the IEEE expressions.  So the visit pattern
// and the counters are identical in both modes and only the accepted terms differ, by a few ulp each.
// Votes.  A predicate that reaches a ballot as a loop-carried bool or as an OR of two comparisons lives in a lane mask, and
// the compiler then materialises it as 0/1 in a VGPR (v_cndmask) and compares that with 0 again (v_cmp_ne) to get the
// ballot — two VALU instructions in a VALU-bound walk.  So the walk votes on FRESH comparisons: the loop's `active` is an
// int (one v_cmp_ne), and "far or not a target" is ballot(far) | ~live_mask.
__device__ __forceinline__ unsigned long long bh_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool bh_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// kFloatSelf: the reference's self-mask as it is written, on FLOAT indices (benchmark.cpp:197-198, 233-235).  Up to index
// 2^24 that is the integer comparison; beyond it a target also skips the leaves of the neighbours whose index rounds to
// the same float (above 2^24 every odd index rounds onto a multiple of 4).  The host picks it for ref_compat >= 1 and
// more than 2^24 bodies; one v_cvt_f32_i32 per accepted node, which is why it is not the only form.
#ifndef NBODY_WALK_SCALAR_VOTE
#define NBODY_WALK_SCALAR_VOTE 1
#endif
// kOffs32: the tree's records are reached as base + 32-bit byte offset (the scalar-base addressing form: one shift per
// record instead of a sign extension, a 64-bit shift-add and the register-pair copies around it).  The host picks it
// while 16 B x (nodes the pool can hold) < 2^32, i.e. up to ~3.3e7 bodies.
template <bool kFast, bool kFloatSelf, bool kOffs32>
__global__ __launch_bounds__(kBhBlock) void bh_walk_kernel(const BhWalkArgs w)
{
#pragma clang fp contract(off)   // every fma below is explicit: lanes reproduce oracle_bh_accel() bit for bit
    // One stack column per GROUP of 8 lanes.  All 8 lanes of a group hold the same (node, sp) and execute the same
    // pushes/pops, so each lane writes the value it will later read back itself (same address, same value): no
    // cross-lane ordering is needed and 4.5 KB of LDS per workgroup leaves room for 8 waves per SIMD.
    __shared__ int stack[kBhStackDepth + 1][kBhBlock / 8];       // row 0 is never an entry: entry k lives in row k
    __shared__ unsigned long long red[3][kBhBlock / 64];
    const int tid = threadIdx.x;
    // wave v of the grid walks groups [v*G, (v+1)*G), G = lanes_used/8, on its first lanes_used lanes
    const int lane = tid & 63;
    // pos: position in walk order (groups are 8 consecutive positions); i: the body there
    const int pos = w.i_begin + (blockIdx.x * (kBhBlock / 64) + (tid >> 6)) * w.lanes_used + lane;
    const bool live = lane < w.lanes_used && pos < w.i_end;
    const int pc = live ? pos : w.i_end - 1;
    const int i = w.order ? w.order[pc] : pc;
    const float xi = w.px[i], yi = w.py[i], ri = __builtin_canonicalizef(w.radius[i]);
    [[maybe_unused]] const float fi = (float)i;
    const int grp_shift = (tid & 63) & ~7;                      // first lane of this group inside the wave
    const float eps = 1e-6f, soft2 = 0.001f * 0.001f;
    const float vote_margin = 4e-6f * w.theta;
    // With kOffs32 the traversal variable `node` is 8 x the node index: the byte offset of the node's B record, half that
    // of its A record (plain indices otherwise).  A finished group holds `done` = the pool's spare last record, never a
    // node: a valid address, so the loads below need no guard.
    constexpr int kSh = kOffs32 ? 3 : 0, kOne = 1 << kSh;
    const int done = w.done_node << kSh;
    auto rec_a = [&](int nd) -> float4 {
        if constexpr (kOffs32) return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(w.A) + ((unsigned)nd << 1));
        else return w.A[nd];
    };
    auto rec_b = [&](int nd) -> int2 {
        if constexpr (kOffs32) return *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(w.B) + (unsigned)nd);
        else return w.B[nd];
    };
    float ax = 0.0f, ay = 0.0f;
    unsigned visits = 0, accepts = 0;

    // is any lane of my group a real target?
    const unsigned long long live_mask = bh_ballot(live);
    int node = ((live_mask >> grp_shift) & 0xFFull) != 0 ? 0 : done, sp = 0;
    stack[0][tid >> 3] = done;                               // (every lane of a group writes its group's column: no barrier)
    unsigned iters = 0;
    // dx, dy and the packed pair res = {1 / (d + eps), f} mean something only in an iteration that takes the node's term.
    // They are declared out here so that on every other path they simply keep what they held (a value of their own per
    // iteration would be given a 0 on every path that does not compute it: a register move per variable and nesting
    // level); the select after the node kernel drops them.
    typedef float f2 __attribute__((ext_vector_type(2)));
    float dx = 0.0f, dy = 0.0f;
    f2 res = {0.0f, 0.0f};
    f2 one_mass = {1.0f, 0.0f};                                  // the dividends of the node kernel's two divisions
    const float theta_lo = w.theta - vote_margin, theta_hi = w.theta + vote_margin;
    // (Tried, round 3: a per-lane `while (active)` loop, finished groups masked off by the hardware instead of by the
    // `if` below.  It removes the ~10 register copies per iteration that the compiler makes around the divergent
    // region, and is SLOWER: walk 2.07 against 1.99 ms at N = 10^6, 0.46 against 0.385 at 65 536 — same-box A/B.)
    //
    // The loop below is bound by the NUMBER of vector instructions: rocprofv3 shows the vector unit busy in every cycle
    // at 4.2 cycles per instruction, whatever their kind (profiles/r3_walk_pmc.txt), while the scalar unit — one
    // instruction per 4 cycles and SIMD as well (tools/ubench_walk.hip) — has room.  Hence:
    //  * decisions are LANE MASKS combined on the scalar unit and handed back to the lanes as conditions (inverse
    //    ballot); written as bools, the compiler turns the same logic into 0/1 registers and compares;
    //  * only the node kernel — long, and skipped by a third of the visits — sits in a branch: every level of divergent
    //    branching costs ~5 scalar instructions of mask bookkeeping and a copy of every value live across it;
    //  * the next node comes out of selects, and its records are requested before this node's term is computed;
    //  * the groups still walking are voted once per iteration and the vote is the next iteration's condition.
    // exact_core.hip.h's sequences need d^2 <= 2^80.  Every centre of mass lies in the root box, so with the box's
    // half-width and this wave's own targets all below 2^38 no d^2 of this wave can be larger (|dx| < 2^39, d^2 < 2^79):
    // one test per wave here instead of one per visit.  (NaN coordinates fail it and take the compiler's expansions.)
    const bool core_ok = w.exact_core && w.ctr->root_half < 0x1p38f &&
                         bh_ballot(!(__builtin_fabsf(xi) < 0x1p38f && __builtin_fabsf(yi) < 0x1p38f)) == 0ull;
    unsigned long long walking = bh_ballot(node != done), not_root = 0ull;
    float4 a = rec_a(0);
    int2 b = rec_b(0);
    while (walking != 0ull) {
        ++iters;
        if (__builtin_amdgcn_inverse_ballot_w64(walking)) {
            ++visits;
            const int top = stack[sp][tid >> 3];                // the sibling to resume at if this subtree ends here (row 0: done)
            dx = a.x - xi; dy = a.y - yi;                       // benchmark.cpp:254-255
            const float d2 = __builtin_fmaf(dx, dx, dy * dy);
            // Below 2^-96 (a target's own leaf: 0) the squared distance changes nothing that follows: d + eps is eps
            // and max(d, r)^2 + soft^2 is soft^2 or r^2 + soft^2 either way.  Clamped there, exact_core.hip.h's
            // sequences cover it, and the wave does not fall back to the compiler's for every own leaf it meets.
            float d2c;
            asm("v_max_f32 %0, %1, %2" : "=v"(d2c) : "v"(d2), "v"(kCoreD2Min));      // (bare: d2 is a number or a quiet NaN)
            // The vote is a boolean, so both modes take it with the hardware's 1-ulp sqrt/rcp (3 instructions instead
            // of the ~30 of a correctly rounded sqrt and division) unless the ratio is within 4e-6 (relative, > 10x
            // its error bound) of theta — it lies on different sides of theta -+ that —; then the wave redoes the node
            // with the IEEE expressions of benchmark.cpp:256-262.  Same decisions as the oracle either way (the
            // counters are compared).
            const float d_hw = __builtin_amdgcn_sqrtf(d2c);
            const float inv_hw = __builtin_amdgcn_rcpf(d_hw + eps);
            // The sums at the end of the iteration multiply dx by res.x even when the term is dropped (f = 0), and res.x
            // is then an OLDER visit's reciprocal distance — up to 1e6, a target's own leaf.  With every |dx| < 2^39 (the
            // range test above) that product is finite.  A wave that failed the test makes res.x this visit's reciprocal
            // here: 1e6 times a dx beyond 3.4e32 would be inf, and inf x 0 NaN.  (Volatile, so that it stays a skipped
            // scalar branch instead of becoming a select in every wave's loop.)
            if (!core_ok) asm volatile("v_mov_b32 %0, %1" : "+v"(res.x) : "v"(inv_hw));
            const float q = a.w * inv_hw;
            unsigned long long fm = bh_ballot(q < theta_lo);
            // (Tried: __builtin_expect here and on the node kernel's range test below, which lays the common path out as
            // straight fall-through — two taken branches fewer per iteration: no gain, and 2.7 % slower with both.)
            if (fm != bh_ballot(q < theta_hi))                                  // rare: ~1e-4 of the waves' visits
                fm = bh_ballot((a.w / (__builtin_sqrtf(d2) + eps)) < w.theta);
            fm |= ~live_mask;                                                   // far, or not a target
#if NBODY_WALK_SCALAR_VOTE
            // "every lane of my group is far (or not a target)" on the scalar unit: s_wqm_b64 turns every 4 bits into
            // "any of them", or-ing the two halves of a byte gives "any of the 8" in its low half, x 17 copies that to
            // the high half (8 scalar instructions)
            asm("s_wqm_b64 %0, %1" : "=s"(fm) : "s"(~fm) : "scc");             // per quad: a target that is not far
            fm = (fm | (fm >> 4)) & 0x0F0F0F0F0F0F0F0Full;
            fm = ~(fm * 0x11ull);
#else
            fm = bh_ballot(((unsigned)(fm >> grp_shift) & 0xFFu) == 0xFFu);     // the same per lane: shift, and, compare
#endif
            const unsigned long long massive_m = bh_ballot(a.z != 0.0f);        // benchmark.cpp:248: an empty node is neither
            const unsigned long long leaf_m = bh_ballot(b.x < 0);               //                    accepted nor opened
            const unsigned long long accept_m = massive_m & (leaf_m | fm);      // :266 leaf or every lane far
            const unsigned long long open_m = massive_m & ~accept_m;
            // :197-198: an accepted node adds its term unless it is the target's own leaf
            const unsigned long long take_m = accept_m & (kFloatSelf ? bh_ballot((float)b.y != fi) : bh_ballot(b.y != i));
            // Next node.  Children are pushed 0..3 and popped 3..0 (:272-274, :244): an opened node continues at its
            // child 3 and leaves its own next sibling on the stack; otherwise the next sibling, else the stack's top,
            // and the top of the empty stack is done: the group is done.  The sibling is written to the free slot
            // above the top in any case and stays there only if sp moves.
            const unsigned long long more_m = bh_ballot(((node - kOne) & (3 * kOne)) != 0) & not_root;    // (the root has no siblings)
            stack[sp + 1][tid >> 3] = node - kOne;
            const int next = __builtin_amdgcn_inverse_ballot_w64(open_m) ? (b.x << kSh) + 3 * kOne : node - kOne;
            node = __builtin_amdgcn_inverse_ballot_w64(open_m | more_m) ? next : top;
            // sp += pushed; sp -= popped: add and subtract with carry-in from the masks
            asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(sp) : "s"(open_m & more_m) : "vcc");
            asm("v_subb_co_u32 %0, vcc, %0, %2, %1" : "+v"(sp) : "s"(~(open_m | more_m)), "v"(0) : "vcc");
            // The next node's records are requested NOW, before this node's term is computed: the node kernel below —
            // the longest dependent stretch of the iteration — runs while they travel.  It needs the mass of this node only.
            // (the copy as a volatile statement: it stays in front of the loads, so these can overwrite the record's own
            // registers; left to the scheduler it follows them, and the new record is then copied at the end of the iteration)
            asm volatile("v_mov_b32 %0, %1" : "=v"(one_mass.y) : "v"(a.z));
            a = rec_a(node);
            b = rec_b(node);
            if (__builtin_amdgcn_inverse_ballot_w64(accept_m)) {
                ++accepts;
                // computeForce_kernel, benchmark.cpp:188-225: the accepted term is computed in IEEE arithmetic
                // (correctly rounded sqrt and divisions) unless bh_fast_math asked for the hardware's
                // (IEEE: the same correctly rounded results from exact_core.hip.h's shorter sequences when the wave's
                // d^2 are in their range — all but waves with a far-escaped target —, else hipcc's)
                if constexpr (kFast) {
                    const float sd = __builtin_fmaxf(d_hw, ri);   // clamp radius is the TARGET's only (:201-202)
                    res = f2{inv_hw, one_mass.y * __builtin_amdgcn_rcpf(__builtin_fmaf(sd, sd, soft2))};
                } else if (core_ok) {
                    const float d = core_sqrt_rn_newton(d2c, d_hw);             // :193-194
                    // fmaxf() would first canonicalize both operands (two more v_max): d is a number here and ri was
                    // canonicalized when it was loaded, so the bare instruction gives fmaxf's result
                    float sd;
                    asm("v_max_f32 %0, %1, %2" : "=v"(sd) : "v"(d), "v"(ri));
                    // {1 / (d + eps), mass / (sd^2 + soft^2)}, :212-213 and :203-209, both correctly rounded, as ONE
                    // packed sequence; the first starts its Newton step from the vote's v_rcp_f32 of the 1-ulp distance
                    const float den = __builtin_fmaf(sd, sd, soft2);
                    res = core_div2_rn(one_mass, f2{d + eps, den}, f2{inv_hw, __builtin_amdgcn_rcpf(den)});
                } else {
                    const float d = __builtin_sqrtf(d2);
                    const float sd = __builtin_fmaxf(d, ri);
                    res = f2{1.0f / (d + eps), one_mass.y / __builtin_fmaf(sd, sd, soft2)};
                }
            }
            // (+0 leaves a sum as it is: the sums start at +0 and x + (-x) = +0, so none of them is ever -0;
            //  |dx * res.x| is finite — see the note at inv_hw above —: times a zero f, a zero)
            const float fs = __builtin_amdgcn_inverse_ballot_w64(take_m) ? res.y : 0.0f;
            ax = ax + (dx * res.x) * fs;
            ay = ay + (dy * res.x) * fs;
        }
        // (Tried: visits = the number of the iteration in which the group ends — a conditional move in those few
        // iterations instead of an add in every one: one vector instruction fewer, no gain.  Tried: both counters in one
        // add-with-carry (2^16 per visit + the accept mask as carry-in), flushed every 2^15 iterations: 1.52 against 1.42 ms.)
        walking = bh_ballot(node != done);
        not_root = ~0ull;
    }
    // The reference integrates each group right after its walk, in place (benchmark.cpp:382-401): that is race-free
    // because the tree holds COPIES of the positions (leaf records), and it is race-free here for the same reason —
    // during the walk a lane reads posX/posY of its own body only, once, above.
    if (live) {
        w.ax[i] = ax;
        w.ay[i] = ay;
        if (w.advance) {
            const float u = w.vx[i], v = w.vy[i];
            w.px[i] = __builtin_fmaf(u, w.dt, xi);          // :389-390 from the OLD velocity
            w.py[i] = __builtin_fmaf(v, w.dt, yi);
            w.vx[i] = __builtin_fmaf(ax, w.dt, u);          // :393-394
            w.vy[i] = __builtin_fmaf(ay, w.dt, v);
        }
    }

    // counters: one representative lane per group
    if ((tid & 7) == 0 && live) { w.grp_visits[pos >> 3] = (int)visits; w.grp_accepts[pos >> 3] = (int)accepts; }
    unsigned long long v = (tid & 7) == 0 ? visits : 0, acc = (tid & 7) == 0 ? accepts : 0;
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off, 64); acc += __shfl_down(acc, off, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = v; red[1][tid >> 6] = acc; red[2][tid >> 6] = iters; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long sv = 0, sa = 0, si = 0;
        for (int k = 0; k < kBhBlock / 64; ++k) { sv += red[0][k]; sa += red[1][k]; si += red[2][k]; }
        atomicAdd(&w.ctr->visits, sv);
        atomicAdd(&w.ctr->accepts, sa);
        atomicAdd(&w.ctr->wave_iters, si);
    }
}

// Parallel walk (up to ~50 000 targets) ------------------------------------------------------------------------------
// Below ~65 000 bodies the walk above is a latency chain: one node record per visit, ~500 dependent visits per group,
// ~0.5 us each, on a chip that is mostly idle (N = 10 000 is 1250 groups for 1024 SIMDs).  Here a group gets a
// workgroup of FOUR waves, and in every wave all 64 lanes work on it: 8 slots x the group's 8 targets, every slot
// testing a different pending node.  When the group opens the root (every wave takes that vote itself), wave k takes the
// subtree of root child 3-k — the reference pops child 3 first (benchmark.cpp:244) — on LDS lists of its own:
//
//   phase A, the four waves in parallel
//     pass 1  traversal, order-free: pop up to 32 pending nodes (4 per slot), load their records together, take the
//             reference's vote per slot (8-bit ballot field), append accepted nodes to a list and push the children of
//             opened ones.  Every pending node carries its DFS position as a path key — digit 3-q per level, because
//             the reference pops child 3 first (benchmark.cpp:244, :272-274) — so the ORDER in which nodes are tested
//             is free;
//     sort    the accepted (key, node) pairs by key (bitonic, in registers): now they are in the reference's visiting
//             order;
//     terms   node kernel (benchmark.cpp:188-225) of the first 96 accepted nodes for the 8 targets, kept in registers;
//   phase B, wave after wave in the reference's order, wave k starting from the sums wave k-1 left in LDS
//             the additions: 8 nodes per round, one per slot, terms through LDS, every lane adds the terms of its
//             target IN ORDER; accepted nodes beyond the first 96 are evaluated here, 16 per round.
//
// Same arithmetic per (node, target) and the same summation order as bh_walk_kernel, so accelerations and the
// visit/accept counters are bit-identical to it and to the oracle.  The lists live in LDS and are bounded (256 accepted,
// 128 pending nodes per wave); a traversal that outgrows them is abandoned and its root split into its four children,
// each walked on its own, in DFS order, when the wave's turn in phase B comes (a node whose lists overflow was opened,
// so this is exactly what the reference does with it; the group's accept count of the previous walk tells a wave to
// split up front instead).  A group that does not open the root (a leaf, an accepted or a massless root) is one visit,
// done by wave 0 alone.  Waves only ever meet at workgroup barriers that all four pass the same number of times.
constexpr int kWidePerSlot = 4;      // nodes tested per slot per iteration of pass 1
constexpr int kWideRound = 2;        // nodes per slot per round of pass 2

// Bitonic sort of 64 * R (key, node) pairs held in registers, element index = lane * R + r: strides below R are
// compare-exchanges between a lane's own registers, strides of R and more go through cross-lane shuffles
// (ds_bpermute, no LDS traffic, no barriers).  An order of magnitude faster than the same network through LDS.
template <int R, int J>
__device__ __forceinline__ void bh_wide_sort_regs(unsigned long long (&k)[R], int (&v)[R], int kk, int lane)
{
    if constexpr (J < R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r & J) == 0) {
                const int q = r | J;
                const bool up = ((lane * R + r) & kk) == 0;
                const bool sw = (k[r] > k[q]) == up;
                const unsigned long long ka = sw ? k[q] : k[r], kb = sw ? k[r] : k[q];
                const int va = sw ? v[q] : v[r], vb = sw ? v[r] : v[q];
                k[r] = ka; k[q] = kb; v[r] = va; v[q] = vb;
            }
        }
    }
}

template <int R>
__device__ __noinline__ void bh_wide_sort(unsigned long long* __restrict__ akey, int* __restrict__ anode, int lane)
{
    unsigned long long k[R];
    int v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { k[r] = akey[lane * R + r]; v[r] = anode[lane * R + r]; }
#pragma unroll 1
    for (int kk = 2; kk <= 64 * R; kk <<= 1) {
#pragma unroll 1
        for (int j = kk >> 1; j >= R; j >>= 1) {           // strides of R and more: the partner is in another lane
            const int lj = j / R;
            const bool want_min = ((lane & lj) == 0) == (((lane * R) & kk) == 0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned long long ok = __shfl_xor(k[r], lj, 64);
                const int ov = __shfl_xor(v[r], lj, 64);
                const bool take = want_min ? ok < k[r] : ok > k[r];
                k[r] = take ? ok : k[r];
                v[r] = take ? ov : v[r];
            }
        }
        if (kk >= 16) bh_wide_sort_regs<R, 8>(k, v, kk, lane);   // strides below R: the lane's own registers
        if (kk >= 8) bh_wide_sort_regs<R, 4>(k, v, kk, lane);
        if (kk >= 4) bh_wide_sort_regs<R, 2>(k, v, kk, lane);
        bh_wide_sort_regs<R, 1>(k, v, kk, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) anode[lane * R + r] = v[r];
}

#ifndef NBODY_QUAD_WAVES
#define NBODY_QUAD_WAVES 5
#endif
struct BhQuadLds {
    unsigned long long pkey[128];
    unsigned long long akey[256];
    int pnode[128];
    int anode[256];
    float termx[8 * kWideRound][8], termy[8 * kWideRound][8];
    int tasks[128];                  // 1 + 3 per level of splitting, 31 levels
    unsigned char tdepth[128];
};
constexpr int kQuadAcc = 256, kQuadStk = 128;

// one wave: LDS traffic of a wave is executed in order, so lanes exchanging data through LDS need no hardware barrier,
// only the compiler must keep the order
__device__ __forceinline__ void bh_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool kFast, bool kFloatSelf>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NBODY_QUAD_WAVES, 8))) void bh_walk_quad_kernel(const BhWalkArgs w)
{
#pragma clang fp contract(off)
    constexpr int kAcc = kQuadAcc, kStk = kQuadStk;
    __shared__ BhQuadLds lds[4];
    __shared__ float accx[8], accy[8];
    __shared__ unsigned cnt_v, cnt_a;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = lane >> 3, t = lane & 7;
    BhQuadLds& S = lds[wv];
    const int pos = w.i_begin + (int)blockIdx.x * 8 + t;            // position in walk order; i: the body there
    const bool live = pos < w.i_end;
    const int pc = live ? pos : w.i_end - 1;
    const int i = w.order ? w.order[pc] : pc;
    const float xi = w.px[i], yi = w.py[i], ri = w.radius[i];
    [[maybe_unused]] const float fi = (float)i;
    const float eps = 1e-6f, soft2 = 0.001f * 0.001f;
    const float vote_margin = 4e-6f * w.theta;
    const unsigned long long below_me = (1ull << lane) - 1ull;
    const unsigned long long below_slot = (1ull << (slot * 8)) - 1ull;
    const unsigned long long not_live_mask = ~bh_ballot(live);
    float ax = 0.0f, ay = 0.0f;
    unsigned visits = 0, accepts = 0;                    // wave-uniform
    int presplit = 0;
    {
        const int hint = w.grp_accepts[(w.i_begin >> 3) + blockIdx.x];
        for (int h = hint; h > kAcc - kAcc / 4 && presplit < 4; h >>= 2) ++presplit;
    }

    // vote of the 8 targets on one node record, per slot (the reference's, benchmark.cpp:248-266)
    // (Tried, end of round 3: the decisions as lane masks on the scalar unit, as bh_walk_kernel takes them — s_wqm_b64 fold
    // per slot, accept / open as and-or of ballots, positions by v_mbcnt.  Same results, and 1-3 % SLOWER at N = 5000 ...
    // 32 768: this kernel is a chain of dependent steps on a mostly idle chip, and the scalar fold is a longer chain than
    // the three vector instructions it replaces.)
    auto vote = [&](const float4& a, const int2& b, bool act, bool& accept, bool& open) {
        const float dx = a.x - xi, dy = a.y - yi;
        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
        const float q = a.w * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(d2) + eps);
        bool far = q < w.theta;
        if (bh_any(__builtin_fabsf(q - w.theta) <= vote_margin))
            far = (a.w / (__builtin_sqrtf(d2) + eps)) < w.theta;
        const unsigned long long fm = bh_ballot(far) | not_live_mask;     // (= ballot(far || !live): see bh_ballot)
        const bool all_far = ((fm >> (slot * 8)) & 0xFFull) == 0xFFull;
        const bool massive = act && a.z != 0.0f;
        accept = massive && (b.x < 0 || all_far);
        open = massive && !accept;
    };

    // pass 1 below `root`: accepted (key, node) pairs into S.akey/S.anode; false when the lists overflowed
    auto pass1 = [&](int root, int rdepth, int& na, unsigned& tv) -> bool {
        bool first_iter = true, overflow = false;
        int sp = 1;
        na = 0; tv = 0;
        if (lane == 0) { S.pnode[0] = root; S.pkey[0] = 1ull; }
        bh_wave_sync();
        while (sp > 0 && !overflow) {
            int node[kWidePerSlot];
            unsigned long long key[kWidePerSlot];
            bool act[kWidePerSlot];
            const int take = sp < 8 * kWidePerSlot ? sp : 8 * kWidePerSlot;
#pragma unroll
            for (int u = 0; u < kWidePerSlot; ++u) {
                const int e = sp - 1 - (u * 8 + slot);
                act[u] = u * 8 + slot < take;
                node[u] = act[u] ? S.pnode[e] : 0;
                key[u] = act[u] ? S.pkey[e] : 1ull;
            }
            sp -= take;
            tv += (unsigned)take;
            float4 a[kWidePerSlot];
            int2 b[kWidePerSlot];
#pragma unroll
            for (int u = 0; u < kWidePerSlot; ++u) { a[u] = w.A[node[u]]; b[u] = w.B[node[u]]; }
            bh_wave_sync();                               // every lane has read its entries before anyone overwrites them
#pragma unroll
            for (int u = 0; u < kWidePerSlot; ++u) {
                bool accept, open;
                vote(a[u], b[u], act[u], accept, open);
                const unsigned long long am = bh_ballot(accept && t == 0);
                const int apos = na + __popcll(am & below_me);
                if (accept && t == 0 && apos < kAcc) {
                    S.akey[apos] = key[u] << (__builtin_clzll(key[u]) - 1);    // left-aligned: sentinel at bit 62
                    S.anode[apos] = node[u];
                }
                na += __popcll(am);
                const unsigned long long om = bh_ballot(open && t == 0);
                const int base = sp + 4 * __popcll(om & below_slot);
                if (open && t < 4 && base + t < kStk) {
                    S.pnode[base + t] = b[u].x + t;                            // children 0..3 (:272-274)
                    S.pkey[base + t] = (key[u] << 2) | (unsigned long long)(3 - t);   // popped 3 first (:244)
                }
                sp += 4 * __popcll(om);
            }
            overflow = sp > kStk || na > kAcc || (first_iter && rdepth < presplit && sp > 0);
            first_iter = false;
            bh_wave_sync();
        }
        return !overflow;
    };

    auto sort_list = [&](int na) {
        if (na <= 1) return;
        int P = 64;
        while (P < na) P <<= 1;
        for (int e = na + lane; e < P; e += 64) S.akey[e] = ~0ull;
        bh_wave_sync();
        if (P <= 64) bh_wide_sort<1>(S.akey, S.anode, lane);
        else if (P <= 128) bh_wide_sort<2>(S.akey, S.anode, lane);
        else bh_wide_sort<4>(S.akey, S.anode, lane);
        bh_wave_sync();
    };

    // node kernel (benchmark.cpp:188-225): the term of one accepted node for this lane's target
    auto term = [&](const float4& a, const int2& b, float& tx, float& ty) {
        const float dx = a.x - xi, dy = a.y - yi;
        const float d2 = __builtin_fmaf(dx, dx, dy * dy);
        float d, inv, f;
        if constexpr (kFast) {
            d = __builtin_amdgcn_sqrtf(d2);
            inv = __builtin_amdgcn_rcpf(d + eps);
            const float sd = __builtin_fmaxf(d, ri);
            f = a.z * __builtin_amdgcn_rcpf(__builtin_fmaf(sd, sd, soft2));
        } else if (w.exact_core && bh_ballot(!(d2 >= kCoreD2Min && d2 <= kCoreD2Max)) == 0ull) {
            // correctly rounded, by exact_core.hip.h's shorter sequences: every lane's d^2 is in their range
            d = core_sqrt_rn_newton(d2, __builtin_amdgcn_sqrtf(d2));
            const float sd = __builtin_fmaxf(d, ri);                   // :201-202
            f = core_div_rn(a.z, __builtin_fmaf(sd, sd, soft2));       // :203-209
            inv = core_rcp_rn(d + eps);                                // :212-213
        } else {
            d = __builtin_sqrtf(d2);
            const float sd = __builtin_fmaxf(d, ri);
            f = a.z / __builtin_fmaf(sd, sd, soft2);
            inv = 1.0f / (d + eps);
        }
        const bool self = kFloatSelf ? (float)b.y == fi : b.y == i;    // :197-198 (see bh_walk_kernel)
        tx = self ? 0.0f : (dx * inv) * f;
        ty = self ? 0.0f : (dy * inv) * f;
    };

    // pass 2 over the sorted list from entry `c_begin` on: every lane adds its target's terms in list order
    auto pass2 = [&](int c_begin, int na) {
        constexpr int kRound = 8 * kWideRound;
        float4 a[kWideRound];
        int2 b[kWideRound];
#pragma unroll
        for (int u = 0; u < kWideRound; ++u) {
            const int e = c_begin + u * 8 + slot;
            const int nd = e < na ? S.anode[e] : 0;
            a[u] = w.A[nd]; b[u] = w.B[nd];
        }
        for (int c0 = c_begin; c0 < na; c0 += kRound) {
            float4 an[kWideRound];
            int2 bn[kWideRound];
#pragma unroll
            for (int u = 0; u < kWideRound; ++u) {
                const int e = c0 + kRound + u * 8 + slot;
                const int nd = e < na ? S.anode[e] : 0;
                an[u] = w.A[nd]; bn[u] = w.B[nd];
            }
#pragma unroll
            for (int u = 0; u < kWideRound; ++u) {
                float tx, ty;
                term(a[u], b[u], tx, ty);
                S.termx[u * 8 + slot][t] = tx;
                S.termy[u * 8 + slot][t] = ty;
            }
            bh_wave_sync();
            const int nv = na - c0 < kRound ? na - c0 : kRound;
            float vx[kRound], vy[kRound];                  // all reads in flight, then the adds in order
#pragma unroll
            for (int u = 0; u < kRound; ++u) { vx[u] = S.termx[u][t]; vy[u] = S.termy[u][t]; }
#pragma unroll
            for (int u = 0; u < kRound; ++u)
                if (u < nv) { ax = ax + vx[u]; ay = ay + vy[u]; }
            bh_wave_sync();
#pragma unroll
            for (int u = 0; u < kWideRound; ++u) { a[u] = an[u]; b[u] = bn[u]; }
        }
    };

    // task loop: a traversal that outgrows the lists is split into its four children, walked one by one
    auto run_tasks = [&](int root, int rdepth) {
        int ntask = 1;
        if (lane == 0) { S.tasks[0] = root; S.tdepth[0] = (unsigned char)rdepth; }
        bh_wave_sync();
        while (ntask > 0) {
            const int r = S.tasks[--ntask];
            const int rd = S.tdepth[ntask];
            int na; unsigned tv;
            if (!pass1(r, rd, na, tv)) {
                visits += 1;                               // `r` was opened: its visit, then its children 3 first
                const int cb = w.B[r].x;
                if (lane < 4) { S.tasks[ntask + lane] = cb + lane; S.tdepth[ntask + lane] = (unsigned char)(rd + 1); }
                ntask += 4;
                bh_wave_sync();
                continue;
            }
            visits += tv;
            accepts += (unsigned)na;
            sort_list(na);
            pass2(0, na);
        }
    };

    // does the group open the root?  Every wave takes the vote itself (same data, same answer: workgroup-uniform).
    const float4 a0 = w.A[0];
    const int2 b0 = w.B[0];
    bool root_accept, root_open;
    vote(a0, b0, true, root_accept, root_open);
    root_open = bh_any(root_open && slot == 0);             // (identical in every slot; made wave-uniform for the compiler)
    if (!root_open) {
        if (wv != 0) return;
        run_tasks(0, 0);                                   // a leaf, an accepted or a massless root: one visit, wave 0 alone
    } else {
        // phase A, all four waves: traversal + sort of root child 3 - wv, and the terms of its first 8 * kHeld accepted
        // nodes, kept in registers (kHeld per lane: node r * 8 + slot for target t) — everything that does not depend
        // on the sums of the subtrees before it
        constexpr int kHeld = 12;
        const int my_root = b0.x + (3 - wv);
        int na; unsigned tv;
#ifdef NBODY_WIDE_TIMING
        const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long q1 = 0, q2 = 0, q4 = 0, q5 = 0;
#endif
        const bool fits = pass1(my_root, 1, na, tv);
#ifdef NBODY_WIDE_TIMING
        q1 = __builtin_amdgcn_s_memrealtime();
#endif
        float hx[kHeld], hy[kHeld];
#pragma unroll
        for (int r = 0; r < kHeld; ++r) { hx[r] = 0.0f; hy[r] = 0.0f; }
        if (fits) {
            sort_list(na);
#pragma unroll
            for (int r0 = 0; r0 < kHeld; r0 += 4) {
                if (r0 * 8 < na) {                         // wave-uniform
                    float4 a[4];
                    int2 b[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = (r0 + u) * 8 + slot;
                        const int nd = e < na ? S.anode[e] : 0;
                        a[u] = w.A[nd]; b[u] = w.B[nd];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) term(a[u], b[u], hx[r0 + u], hy[r0 + u]);
                }
            }
        }
#ifdef NBODY_WIDE_TIMING
        q2 = __builtin_amdgcn_s_memrealtime();
#endif
        if (threadIdx.x == 0) { cnt_v = 1; cnt_a = 0; }    // the root's own visit
        __syncthreads();
        // phase B, one wave after the other in the reference's order: only the additions (and what did not fit above).
        // Every wave passes four barriers: wv of them before its turn, 4 - wv after it.
        for (int k = 0; k < wv; ++k) __syncthreads();
#ifdef NBODY_WIDE_TIMING
        q4 = __builtin_amdgcn_s_memrealtime();
#endif
        if (wv > 0) { ax = accx[t]; ay = accy[t]; }
        if (fits) {
            visits = tv; accepts = (unsigned)na;
#pragma unroll
            for (int r = 0; r < kHeld; ++r) {
                if (r * 8 < na) {                          // wave-uniform
                    S.termx[slot][t] = hx[r]; S.termy[slot][t] = hy[r];
                    bh_wave_sync();
                    const int nv = na - r * 8 < 8 ? na - r * 8 : 8;
                    float vx[8], vy[8];                    // all reads in flight, then the adds in order
#pragma unroll
                    for (int u = 0; u < 8; ++u) { vx[u] = S.termx[u][t]; vy[u] = S.termy[u][t]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < nv) { ax = ax + vx[u]; ay = ay + vy[u]; }
                    bh_wave_sync();
                }
            }
            if (na > 8 * kHeld) pass2(8 * kHeld, na);
        } else {
            run_tasks(my_root, 1);
        }
        if (slot == 0) { accx[t] = ax; accy[t] = ay; }
        if (lane == 0) { cnt_v += visits; cnt_a += accepts; }
#ifdef NBODY_WIDE_TIMING
        q5 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int k = wv; k < 4; ++k) __syncthreads();
#ifdef NBODY_WIDE_TIMING      // developer build: 100 MHz ticks per phase instead of results, wave wv in bodies 2 wv, 2 wv + 1
        if (slot == 0 && (t >> 1) == wv && live) {
            const unsigned long long q6 = __builtin_amdgcn_s_memrealtime();
            w.ax[i] = (t & 1) ? (float)(q5 - q4) : (float)(q1 - q0);
            w.ay[i] = (t & 1) ? (float)(q6 - q0) : (float)(q2 - q1);
        }
        if (wv == 3 && lane == 0) { w.grp_visits[(w.i_begin >> 3) + blockIdx.x] = (int)cnt_v; w.grp_accepts[(w.i_begin >> 3) + blockIdx.x] = (int)cnt_a; }
        return;
#endif
        if (wv != 3) return;
        visits = cnt_v; accepts = cnt_a;
    }
    if (live && slot == 0) {
        w.ax[i] = ax;
        w.ay[i] = ay;
        if (w.advance) {
            const float u = w.vx[i], v = w.vy[i];
            w.px[i] = __builtin_fmaf(u, w.dt, xi);          // benchmark.cpp:389-390 from the OLD velocity
            w.py[i] = __builtin_fmaf(v, w.dt, yi);
            w.vx[i] = __builtin_fmaf(ax, w.dt, u);          // :393-394
            w.vy[i] = __builtin_fmaf(ay, w.dt, v);
        }
    }
    if (lane == 0) {
        w.grp_visits[(w.i_begin >> 3) + blockIdx.x] = (int)visits;
        w.grp_accepts[(w.i_begin >> 3) + blockIdx.x] = (int)accepts;
        atomicAdd(&w.ctr->visits, (unsigned long long)visits);
        atomicAdd(&w.ctr->accepts, (unsigned long long)accepts);
    }
}

}  // namespace nbody
