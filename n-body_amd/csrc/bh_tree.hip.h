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
//   5. bh_com_kernel       bottom-up, one launch per depth: mass = ((0+m0)+m1)+m2)+m3,
//                          com = sum(com_k*m_k)/mass in child order (benchmark.cpp:122-141)
//
// Everything is deterministic (no atomics decide any index) and nothing synchronises with the host.
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

namespace nbody {

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
    int pad_;
};

// walk record of a node: A = {comX, comY, mass, width}, B = {childBase (-1: leaf), bodyIndex (-1: none)}
struct BhTree {
    float4* __restrict__ A;
    int2* __restrict__ B;
    uint8_t* __restrict__ depth;         // per internal cell
    int* __restrict__ self;              // per internal cell: its node index
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
                                                           int* __restrict__ tie_table, int tie_table_size)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    for (int g = i; g < gcap; g += gridDim.x * kBhBlock) { gkeys[g] = ~0ull; gvals[g] = -1; }
    for (int g = i; g < tie_table_size; g += gridDim.x * kBhBlock) tie_table[g] = -1;
    if (fixed_box && i == 0) bh_reset_counters(ctr, spread_fixed);
    if (i >= n) return;
    const float spread = fixed_box ? spread_fixed : ctr->root_half;
    const float x = px[i], y = py[i];
    float tlx = -spread, tly = -spread, brx = spread, bry = spread;
    uint64_t key;
    int tx = kBhLevels, ty = kBhLevels;                    // first level at which x (y) sits ON the split line
    if (x < tlx || x > brx || y < tly || y > bry) {         // benchmark.cpp:83-85
        key = kBhInvalidKey | (uint64_t)(uint32_t)i;       // keep invalid keys distinct and ordered
    } else {
        key = 0;
#pragma unroll 1
        for (int l = 0; l < kBhLevels; ++l) {
            const float mx = (tlx + brx) / 2.0f, my = (tly + bry) / 2.0f;    // benchmark.cpp:98-99
            tx = (x == mx && tx == kBhLevels) ? l : tx;
            ty = (y == my && ty == kBhLevels) ? l : ty;
            const int qx = x > mx, qy = y > my;            // on the line -> lower child
            key = (key << 2) | (uint64_t)(qy * 2 + qx);    // TL, TR, BL, BR (benchmark.cpp:105-108)
            tlx = qx ? mx : tlx; brx = qx ? brx : mx;
            tly = qy ? my : tly; bry = qy ? bry : my;
        }
    }
    keys[i] = key;
    vals[i] = i;
    tie[i] = (uint16_t)(tx | (ty << 8));
}

// number of valid (in-box) bodies = first sorted index with the invalid bit set
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
                __threadfence();                                 // the record before its index becomes findable
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

__global__ __launch_bounds__(kBhBlock) void bh_ties_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                           const float* __restrict__ px, const float* __restrict__ py,
                                                           const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                           BhTies T)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    const int M = ctr->valid;
    if (i >= M) return;
    const int b = vals[i];
    const int first = bh_first_tie(tie[b]);
    if (first == kBhLevels) return;                              // no split line on the primary's path:
                                                                 // all but a few hundred per million
    const uint64_t k = keys[i];
    const int lp = i > 0 ? bh_lcp_levels(keys[i - 1], k) : -1;
    const int ln = i + 1 < M ? bh_lcp_levels(k, keys[i + 1]) : -1;
    const int leaf_depth = 1 + (lp > ln ? lp : ln);              // 0 when the body is alone in the tree
    T.pld[i] = leaf_depth;
    bh_spawn_ties(px[b], py[b], ctr->root_half, 0u, 0u, 0, leaf_depth, b, first, keys, vals, M, T, ctr);
}

constexpr int kBhFixBlock = 1024;
__global__ __launch_bounds__(kBhFixBlock) void bh_tie_fixpoint_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                                      const float* __restrict__ px, const float* __restrict__ py,
                                                                      const uint16_t* __restrict__ tie, BhCounters* __restrict__ ctr,
                                                                      BhTies T)
{
    __shared__ int s_end;
    const int M = ctr->valid;
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

// merge sorted primaries [n] and sorted second instances [gcap] into one sorted array [n + gcap]
__global__ __launch_bounds__(kBhBlock) void bh_merge_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals, int n,
                                                            const uint64_t* __restrict__ gkeys, const int* __restrict__ gvals,
                                                            int gcap, uint64_t* __restrict__ okeys, int* __restrict__ ovals,
                                                            BhCounters* __restrict__ ctr)
{
    const int t = blockIdx.x * kBhBlock + threadIdx.x;
    // valid INSTANCES of the merged array = valid primaries + second instances (their keys have bit 63 clear, the
    // padding and the out-of-box bodies have it set and sort behind them); nobody else in this launch reads ctr
    if (t == 0) ctr->valid += ctr->ghosts < gcap ? ctr->ghosts : gcap;
    if (t < n) {
        const uint64_t k = keys[t];
        int lo = 0, hi = gcap;                                 // second instances with a smaller key
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (gkeys[mid] < k) lo = mid + 1; else hi = mid; }
        okeys[t + lo] = k;
        ovals[t + lo] = vals[t];
    } else if (t < n + gcap) {
        const int g = t - n;
        const uint64_t k = gkeys[g];
        int lo = 0, hi = n;                                    // primaries with a key <= mine
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] <= k) lo = mid + 1; else hi = mid; }
        okeys[g + lo] = k;
        ovals[g + lo] = gvals[g];
    }
}

// 3. internal cells owned by each sorted body --------------------------------------------------------------
__global__ __launch_bounds__(kBhBlock) void bh_count_kernel(const uint64_t* __restrict__ keys, int n,
                                                            const BhCounters* __restrict__ ctr, int* __restrict__ cnt)
{
    const int i = blockIdx.x * kBhBlock + threadIdx.x;
    if (i >= n) return;
    const int M = ctr->valid;
    int c = 0;
    if (i < M) {
        const uint64_t k = keys[i];
        const int lp = i > 0 ? bh_lcp_levels(keys[i - 1], k) : -1;
        const int ln = i + 1 < M ? bh_lcp_levels(k, keys[i + 1]) : -1;
        c = ln > lp ? ln - lp : 0;
    }
    cnt[i] = c;
}

__device__ __forceinline__ int bh_lower_bound(const uint64_t* __restrict__ keys, int lo, int hi, uint64_t v)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// first index in [lo, hi) whose key is >= v, galloping from lo (the answer is usually a few bodies away)
__device__ __forceinline__ int bh_gallop_lower_bound(const uint64_t* __restrict__ keys, int lo, int hi, uint64_t v)
{
    int a = lo, step = 1;
    while (a + step <= hi && keys[a + step - 1] < v) { a += step; step <<= 1; }
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

// 4b. one thread per internal cell: box, body range (galloping from the owner: the range of a typical cell is a
// few bodies long), the four children.
__global__ __launch_bounds__(kBhBlock) void bh_nodes_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ vals,
                                                            const int* __restrict__ cnt, const int* __restrict__ base,
                                                            const int* __restrict__ owner, int n,
                                                            const float* __restrict__ m, const float* __restrict__ px,
                                                            const float* __restrict__ py, BhTree t,
                                                            BhCounters* __restrict__ ctr, int* __restrict__ overflow_host)
{
    const int c = blockIdx.x * kBhBlock + threadIdx.x;
    const int M = ctr->valid;
    const float spread = ctr->root_half;
    const int I = M >= 2 ? base[M - 1] + cnt[M - 1] : 0;
    if (c == 0) {
        ctr->internal = I < t.cap_internal ? I : t.cap_internal;
        if (I > t.cap_internal) ctr->overflow = 1;
        // the build's verdict goes to a host-mapped word, so that nbody_sync needs no device-to-host copy (the tie
        // buffer's overflow, 2, was set by the kernels before this one)
        *overflow_host = I > t.cap_internal ? 1 : ctr->overflow;
        if (M < 2) {                                           // degenerate trees: root is a leaf / empty
            const int b = M == 1 ? vals[0] : -1;
            t.A[0] = make_float4(b >= 0 ? px[b] : 0.0f, b >= 0 ? py[b] : 0.0f, b >= 0 ? m[b] : 0.0f, 2.0f * spread);
            t.B[0] = make_int2(-1, b);
        }
    }
    if (c >= I || c >= t.cap_internal) return;

    const int i = owner[c];
    const uint64_t ki = keys[i];
    const int lp = i > 0 ? bh_lcp_levels(keys[i - 1], ki) : -1;
    const int d = lp + 1 + (c - base[i]);                      // depth of this cell, 0..30
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
    if (c == 0) {
        t.A[0] = make_float4(0.0f, 0.0f, 0.0f, brx - tlx);
        t.B[0] = make_int2(1, -1);
        t.self[0] = 0;
    }
    t.depth[c] = (uint8_t)d;

    // body range [i, e) of the cell and its quadrant boundaries
    const int e = d ? bh_gallop_lower_bound(keys, i + 1, M, (prefix + 1) << shift) : M;
    const int cshift = shift - 2;
    int s[5];
    s[0] = i;
    s[4] = e;
#pragma unroll
    for (int q = 1; q < 4; ++q) s[q] = bh_gallop_lower_bound(keys, s[q - 1], e, ((prefix << 2) | (uint64_t)q) << cshift);

    const float mx = (tlx + brx) / 2.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int node = 1 + 4 * c + q;
        const float ctlx = (q & 1) ? mx : tlx, cbrx = (q & 1) ? brx : mx;
        const float width = cbrx - ctlx;                       // benchmark.cpp:251
        const int cntq = s[q + 1] - s[q];
        int child_base = -1, body = -1;
        float cx = 0.0f, cy = 0.0f, cm = 0.0f;
        if (cntq >= 2 && d + 1 < kBhLevels) {
            const int sq = s[q];
            const int lpq = sq > 0 ? bh_lcp_levels(keys[sq - 1], keys[sq]) : -1;
            const int cc = base[sq] + d - lpq;                 // id of the internal child
            if (cc < t.cap_internal) {
                child_base = 1 + 4 * cc;
                t.self[cc] = node;
            } else {                                           // pool exhausted: keep one body (benchmark.cpp:102)
                body = vals[sq];
            }
        } else if (cntq >= 1) {
            body = vals[s[q]];                                  // leaf
        }
        if (body >= 0) { cx = px[body]; cy = py[body]; cm = m[body]; }   // benchmark.cpp:89-92
        if (child_base < 0 && cntq >= 2) {
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
                const float mj = m[bj];
                ms += mj;
                sx = __builtin_fmaf(px[bj], mj, sx);
                sy = __builtin_fmaf(py[bj], mj, sy);
            }
            cm = ms;
            if (ms > 0.0f) { cx = sx / ms; cy = sy / ms; }
        }
        t.A[node] = make_float4(cx, cy, cm, width);
        t.B[node] = make_int2(child_base, body);
    }
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

// 5. centre of mass, one launch per depth (deepest first) ---------------------------------------------------
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

// One thread per 4 cells (one 32-bit load of their depth bytes): a quarter of the threads of a launch per pool
// slot, short serial tails.
__global__ __launch_bounds__(kBhBlock) void bh_com_kernel(BhTree t, const BhCounters* __restrict__ ctr, int d, int min_internal)
{
    const int I = ctr->internal;
    if (I <= min_internal) return;                              // done by bh_com_small_kernel
    const int c4 = (blockIdx.x * kBhBlock + threadIdx.x) * 4;
    if (c4 >= I) return;
    const unsigned x = *reinterpret_cast<const unsigned*>(t.depth + c4) ^ (0x01010101u * (unsigned)d);
    if (((x - 0x01010101u) & ~x & 0x80808080u) == 0) return;    // no zero byte = no cell of depth d here
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (((x >> (8 * k)) & 0xFFu) == 0 && c4 + k < I) bh_com_cell(t, c4 + k);
}

// Small trees: all depths in ONE launch of one workgroup (31 launches of a few microseconds each are most of a
// Barnes-Hut step at N = 10 000).  Falls through to nothing when the tree is larger than `max_internal`.
constexpr int kBhComBlock = 1024;
__global__ __launch_bounds__(kBhComBlock) void bh_com_small_kernel(BhTree t, const BhCounters* __restrict__ ctr, int max_internal)
{
    const int I = ctr->internal;
    if (I > max_internal) return;
    // each thread keeps the depths of its (at most kPer) cells in registers: the level loop then touches memory
    // only for the cells of that level
    constexpr int kPer = 32;                                    // max_internal <= kPer * kBhComBlock
    uint8_t dl[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int c = threadIdx.x + k * kBhComBlock;
        dl[k] = c < I ? t.depth[c] : (uint8_t)0xFF;
    }
    for (int d = kBhLevels - 1; d >= 0; --d) {                 // a level without cells costs one barrier
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (dl[k] == d) bh_com_cell(t, threadIdx.x + k * kBhComBlock);
        __syncthreads();        // one workgroup: a barrier (with its LDS/global fence) orders level d before d-1
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
template <bool kFast>
__global__ __launch_bounds__(kBhBlock) void bh_walk_kernel(const BhWalkArgs w)
{
#pragma clang fp contract(off)   // every fma below is explicit: lanes reproduce oracle_bh_accel() bit for bit
    // One stack column per GROUP of 8 lanes.  All 8 lanes of a group hold the same (node, sp) and execute the same
    // pushes/pops, so each lane writes the value it will later read back itself (same address, same value): no
    // cross-lane ordering is needed and 4.3 KB of LDS per workgroup leaves room for 8 waves per SIMD.
    __shared__ int stack[kBhStackDepth][kBhBlock / 8];
    __shared__ unsigned long long red[3][kBhBlock / 64];
    const int tid = threadIdx.x;
    // wave v of the grid walks groups [v*G, (v+1)*G), G = lanes_used/8, on its first lanes_used lanes
    const int lane = tid & 63;
    const int i = w.i_begin + (blockIdx.x * (kBhBlock / 64) + (tid >> 6)) * w.lanes_used + lane;
    const bool live = lane < w.lanes_used && i < w.i_end;
    const int ic = live ? i : w.i_end - 1;
    const float xi = w.px[ic], yi = w.py[ic], ri = w.radius[ic];
    const int grp_shift = (tid & 63) & ~7;                      // first lane of this group inside the wave
    const float eps = 1e-6f, soft2 = 0.001f * 0.001f;
    const float vote_margin = 4e-6f * w.theta;
    float ax = 0.0f, ay = 0.0f;
    unsigned visits = 0, accepts = 0;

    // is any lane of my group a real target?
    const unsigned long long live_mask = __ballot(live);
    bool active = ((live_mask >> grp_shift) & 0xFFull) != 0;
    int node = 0, sp = 0;
    float4 a = w.A[0];
    int2 b = w.B[0];
    unsigned iters = 0;
    while (__any(active)) {
        ++iters;
        if (active) {
            // (Tried: requesting the records of both likely successors — next sibling and last child — before this
            // node's arithmetic.  Same-box A/B: no gain at N = 10^4..65 536, 28 % slower at N = 10^6; not kept.
            // Tried: the whole body as straight-line predicated code (selects instead of nested branches): 110 ->
            // 75 instructions per iteration, but the divisions then run for every group in every iteration:
            // +8 % at N = 10^6 (IEEE) / +4 % (fast math), -14 % at N = 10^4; not kept.)
            const bool more_siblings = node != 0 && ((node - 1) & 3) != 0;
            ++visits;
            bool open = false;
            if (a.z != 0.0f) {                                  // benchmark.cpp:248
                const float dx = a.x - xi, dy = a.y - yi;       // :254-255
                const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                float d, inv;
                bool far;
                if constexpr (kFast) {
                    d = __builtin_amdgcn_sqrtf(d2);
                    inv = __builtin_amdgcn_rcpf(d + eps);
                    const float q = a.w * inv;
                    far = q < w.theta;
                    if (__any(__builtin_fabsf(q - w.theta) <= vote_margin))          // rare: ~1e-4 of the waves' visits
                        far = (a.w / (__builtin_sqrtf(d2) + eps)) < w.theta;
                } else {
                    d = __builtin_sqrtf(d2);                                         // :256-257
                    far = (a.w / (d + eps)) < w.theta;                               // :259-262
                }
                const unsigned long long fm = __ballot(far || !live);
                const bool all_far = ((fm >> grp_shift) & 0xFFull) == 0xFFull;
                if (b.x < 0 || all_far) {                       // :266 leaf or every lane far
                    ++accepts;
                    // computeForce_kernel, benchmark.cpp:188-225
                    const float sd = __builtin_fmaxf(d, ri);    // clamp radius is the TARGET's only (:201-202)
                    float f;
                    if constexpr (kFast) {
                        f = a.z * __builtin_amdgcn_rcpf(__builtin_fmaf(sd, sd, soft2));
                    } else {
                        f = a.z / __builtin_fmaf(sd, sd, soft2);   // :203-209
                        inv = 1.0f / (d + eps);                    // :212-213
                    }
                    const bool self = b.y == i;                 // :197-198
                    ax = ax + (self ? 0.0f : (dx * inv) * f);
                    ay = ay + (self ? 0.0f : (dy * inv) * f);
                } else {
                    open = true;
                }
            }
            if (open) {
                if (more_siblings) stack[sp++][tid >> 3] = node - 1;
                node = b.x + 3;                                 // children pushed 0..3, popped 3..0 (:272-274, :244)
                a = w.A[node]; b = w.B[node];
            } else if (more_siblings) {
                node = node - 1;
                a = w.A[node]; b = w.B[node];
            } else if (sp > 0) {
                node = stack[--sp][tid >> 3];
                a = w.A[node];
                b = w.B[node];
            } else {
                active = false;
            }
        }
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
    if ((tid & 7) == 0 && i < w.i_end && lane < w.lanes_used) { w.grp_visits[i >> 3] = (int)visits; w.grp_accepts[i >> 3] = (int)accepts; }
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

}  // namespace nbody
