// bh_sort.hip.h — this repository's own device sort and scan for the Barnes-Hut build (gfx950).
//
// Replaces what the reference does serially at benchmark.cpp:156-173 (insertion order) on the GPU side of the build:
// rounds 1-2 used hipcub::DeviceRadixSort / DeviceScan here (12 rocPRIM launches, 221 of the 485 us of the N = 10^6
// build).  Two pieces:
//
//  * sort_pairs: stable LSD radix sort of (64-bit key, int value) pairs, 8-bit digits, 10 launches:
//      sort_hist_kernel   ONE pass over the keys: every workgroup counts the 8 digit positions of its chunk in LDS and
//                         stores its 8 x 256 counters (plain stores, no global atomics);
//      sort_plan_kernel   8 workgroups, one per digit position: sum the chunks' counters, exclusive scan over the bins
//                         (= where each digit value starts in the output) and whether the position is LIVE — a position
//                         whose histogram is a single bin moves nothing;
//      sort_pass_kernel   one launch per digit position.  Dead positions are skipped ON THE DEVICE (the launch sequence is
//                         static, so that it can sit in a hipGraph): from the live mask every workgroup knows which buffer
//                         this pass reads and which it writes, such that the LAST live pass lands in the output arrays.
//                         A live pass is one sweep: a workgroup takes the next tile (ticket), ranks its 2048 or 4096 pairs
//                         stably in LDS (per wave: 8 ballots give every lane the peers that share its digit; per-wave
//                         counters; a 256-bin scan), publishes the tile's bin counts, collects the counts of all tiles
//                         before it (below) and scatters through LDS in runs of equal digits.
//    Counts of the tiles before: decoupled look-back — {epoch and pass, flag, count} in ONE 64-bit word per (tile, bin),
//    written and read with device-scope relaxed atomics: nothing to fence, nothing to clear between sorts.  (Tried: a
//    16-ary tree over the tiles in which a tile sums the <= 15 siblings before it at every level with independent loads
//    and the last tile under a node publishes the node's total — no chain of prefixes, a tenth of the loads.  Slower: 33
//    against 26 us per pass at N = 10^6, 94 against 52 at 4 x 10^6; every level waits for the slowest of up to 15 tiles.)
//    Everything is deterministic: tickets decide WHICH workgroup sorts a tile, never where an element goes.
//    Every wait is bounded: a workgroup only ever waits for words of tiles with smaller tickets — held by workgroups that
//    are already running and publish before they wait themselves — and gives up after ~0.3 s with SortState::error set.
//
//  * bh_cells_scan_kernel (bh_tree.hip.h): cells owned per sorted instance, their exclusive scan (= cell ids) and the
//    owner scatter in ONE launch (was bh_count_kernel + rocPRIM's scan + bh_owner_kernel): decoupled look-back with one
//    value per tile, examined a wave (64 predecessor tiles) at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

namespace nbody {

constexpr int kSortBlock = 256;             // 4 waves
constexpr int kSortBins = 256;              // 8-bit digits
constexpr int kSortPassBlock = 512;         // threads of a sort_pass_kernel workgroup (256, 512 or 1024: the tile stays the same).
                                            // Measured per sweep, 256 / 512 / 1024 threads: N = 2 x 10^5 13.0 / 11.8 / 12.0 us, 10^6 22.7 /
                                            // 20.3 / 20.2, 4 x 10^6 49.8 / 45.3 / 53.2, 2^24 167 / 155 / 197 (8192-pair tiles at 512: +-0)
constexpr int kSortDigits = 8;
constexpr int kSortHistBlocks = 256;        // chunks of the histogram pass
#ifndef NBODY_SORT_LOOK_W
#define NBODY_SORT_LOOK_W 4
#endif
constexpr int kSortLookW = NBODY_SORT_LOOK_W;   // predecessor tiles a look-back step has in flight
constexpr int kSortSpinLimit = 1 << 21;     // polls (each >= 128 cycles of s_sleep) before a wait gives up
constexpr unsigned long long kLookAgg = 1ull << 30, kLookPrefix = 2ull << 30, kLookCount = (1ull << 30) - 1;

struct SortState {                          // device-resident, one per sort instance; zero-initialised once, then epoch = 1
    unsigned epoch;                         // of the sort in progress (>= 1: no tag ever equals a zero-initialised word's);
                                            // bumped when the sort's last pass has finished
    int tickets[kSortDigits];               // next tile of each pass; cleared by the plan kernel
    int done_last;                          // workgroups of the last pass that have finished
    int error;                              // 1: a wait gave up (results are then garbage; reported, never silent)
    int long_runs;                          // truncated sort: runs handed to sort_long_runs_kernel (cleared by the plan kernel)
    unsigned live_d[kSortDigits];           // plan: 1 = this digit position moves something
    unsigned base[kSortDigits][kSortBins];  // plan: exclusive scan of the digit histograms
#ifdef NBODY_WIDE_TIMING
    unsigned long long dbg_sum[12], dbg_max[12], dbg_cnt;   // developer build: 100 MHz ticks since the workgroup's start, per phase
#endif
};

#ifdef NBODY_WIDE_TIMING
#define SORT_TICK(p) do { if (threadIdx.x == 0 && work) { const unsigned long long d_ = __builtin_amdgcn_s_memrealtime() - t_begin; \
                          atomicAdd(&st->dbg_sum[p], d_); atomicMax(&st->dbg_max[p], d_); } } while (0)
#else
#define SORT_TICK(p) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long sort_look_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sort_look_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive scan of one value per thread over a BLOCK-thread workgroup; `part` is BLOCK / 64 words of LDS.  Ends with a barrier.
template <int BLOCK = kSortBlock>
__device__ __forceinline__ unsigned sort_block_excl_scan(unsigned v, unsigned* part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    unsigned before = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { const unsigned p = part[w]; if (w < wave) before += p; }
    __syncthreads();
    return before + incl - v;
}

// chunk b of the keys -> part[b][digit position][bin].  1024 threads: the kernel is one workgroup per CU (256 chunks), and
// its LDS atomics pile up on a few bins in the high digit positions — four waves per SIMD overlap what one cannot
// (256 -> 1024 threads: 2^24 keys 82 -> 42 us, 4 x 10^6 19.5 -> 10.3, 10^6 6.2 -> 5.0).
constexpr int kSortHistThreads = 1024;
__global__ __launch_bounds__(kSortHistThreads) void sort_hist_kernel(const uint64_t* __restrict__ keys, int n, unsigned* __restrict__ part, int first)
{
    __shared__ unsigned h[kSortDigits][kSortBins];
    const int tid = threadIdx.x;
    for (int q = tid; q < kSortDigits * kSortBins; q += kSortHistThreads) (&h[0][0])[q] = 0;
    __syncthreads();
    const int per = ((n + (int)gridDim.x - 1) / (int)gridDim.x + kSortHistThreads - 1) / kSortHistThreads * kSortHistThreads;
    const int b0 = (int)blockIdx.x * per, b1 = min(n, b0 + per);
    for (int i0 = b0; i0 < b1; i0 += 4 * kSortHistThreads) {
        uint64_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * kSortHistThreads + tid; k[u] = i < b1 ? keys[i] : 0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * kSortHistThreads + tid < b1) {
#pragma unroll
                for (int d = 0; d < kSortDigits; ++d)
                    if (d >= first) atomicAdd(&h[d][(unsigned)(k[u] >> (8 * d)) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    unsigned* g = part + (size_t)blockIdx.x * (kSortDigits * kSortBins);
    for (int q = tid; q < kSortDigits * kSortBins; q += kSortHistThreads) g[q] = (&h[0][0])[q];
}

// workgroup d = digit position d: total histogram (1024 threads: four take a quarter of the chunks of one bin each),
// liveness, exclusive scan over the bins; workgroup 0 also clears the tickets
__global__ __launch_bounds__(1024) void sort_plan_kernel(const unsigned* __restrict__ part, int chunks, int n, SortState* __restrict__ st, int first)
{
    __shared__ unsigned quarter[4][kSortBins];
    __shared__ unsigned wave_total[16];
    const int tid = threadIdx.x, d = blockIdx.x, bin = tid & (kSortBins - 1), q = tid >> 8;
    const int lane = tid & 63, wave = tid >> 6;
    unsigned c = 0;
    for (int b0 = q; b0 < chunks; b0 += 64) {
        unsigned t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = b0 + 4 * u < chunks ? part[(size_t)(b0 + 4 * u) * (kSortDigits * kSortBins) + d * kSortBins + bin] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) c += t[u];
    }
    quarter[q][bin] = c;
    __syncthreads();
    const unsigned tot = tid < kSortBins ? quarter[0][tid] + quarter[1][tid] + quarter[2][tid] + quarter[3][tid] : 0u;
    const int dead = __syncthreads_or(tid < kSortBins && tot == (unsigned)n);      // one bin holds every key
    unsigned incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wave; ++w) before += wave_total[w];
    if (tid < kSortBins) st->base[d][tid] = before + incl - tot;
    if (tid == 0) st->live_d[d] = (dead || d < first) ? 0u : 1u;       // positions below `first` are left to sort_fix_runs_kernel
    if (d == 0 && tid < kSortDigits) st->tickets[tid] = 0;
    if (d == 0 && tid == 0) { st->long_runs = 0; st->error = 0; }     // the flag describes THIS sort (its reader comes after it)
}

inline size_t sort_look_words(int tiles) { return (size_t)tiles * kSortBins; }

// One digit position.  BLOCK threads, E pairs per thread, tile = BLOCK * E.  Buffers: IN (read-only), OUT (final result), TMP.
// The 256 bins belong to the first 256 threads; with more threads per workgroup the same tile has fewer serial ranking
// rounds per wave and more loads in flight per CU.
template <int E, int BLOCK>
__global__ __launch_bounds__(BLOCK) void sort_pass_kernel(SortState* __restrict__ st, int k, int n,
                                                              const uint64_t* __restrict__ in_k, const int* __restrict__ in_v,
                                                              uint64_t* __restrict__ out_k, int* __restrict__ out_v,
                                                              uint64_t* __restrict__ tmp_k, int* __restrict__ tmp_v,
                                                              unsigned long long* __restrict__ look)
{
    constexpr int T = BLOCK * E;
    __shared__ unsigned wave_hist[BLOCK / 64][kSortBins];
    __shared__ unsigned bin_start[kSortBins];
    __shared__ int gbase[kSortBins];
    __shared__ uint64_t skey[T];
    __shared__ int sval[T];
    __shared__ unsigned part[BLOCK / 64];
    __shared__ int s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NBODY_WIDE_TIMING
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned epoch = st->epoch;                      // stable during this kernel: bumped only by the last pass's last workgroup
    unsigned live = 0;
#pragma unroll
    for (int d = 0; d < kSortDigits; ++d) live |= (st->live_d[d] ? 1u : 0u) << d;
    const int P = __popc(live);
    bool work = n > 0 && ((live >> k) & 1u);
    const uint64_t* src_k = in_k;
    const int* src_v = in_v;
    uint64_t* dst_k = out_k;
    int* dst_v = out_v;
    {
        const int idx = __popc(live & ((1u << k) - 1u));   // live passes before this one
        const bool to_out = ((P - 1 - idx) & 1) == 0;      // the last live pass writes OUT, the one before TMP, ...
        dst_k = to_out ? out_k : tmp_k;
        dst_v = to_out ? out_v : tmp_v;
        if (idx > 0) { src_k = to_out ? tmp_k : out_k; src_v = to_out ? tmp_v : out_v; }
    }
    if (work && tid == 0) s_tile = atomicAdd(&st->tickets[k], 1);
    __syncthreads();
    const int tile = work ? s_tile : 0;
    if (!work && k == kSortDigits - 1 && P == 0 && n > 0) {
        // nothing to sort (one element, or all keys equal): the output is the input
        for (int i = (int)blockIdx.x * T + tid; i < min(n, ((int)blockIdx.x + 1) * T); i += BLOCK) { out_k[i] = in_k[i]; out_v[i] = in_v[i]; }
    }
    work = work && (long long)tile * T < n;

    SORT_TICK(0);
    if (work) {
        const int t0 = tile * T;
        const int shift = 8 * k;
        const unsigned tag = epoch * kSortDigits + (unsigned)k;
        for (int q = tid; q < (BLOCK / 64) * kSortBins; q += BLOCK) (&wave_hist[0][0])[q] = 0;
        uint64_t key[E];
        int val[E];
        unsigned rank[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t0 + wave * (64 * E) + e * 64 + lane;       // tile order = memory order (stability)
            key[e] = i < n ? src_k[i] : ~0ull;
            val[e] = i < n ? src_v[i] : 0;
        }
        const bool bin_thread = tid < kSortBins;               // (BLOCK >= 256)
        const unsigned digit_base = bin_thread ? st->base[k][tid] : 0u;
        __syncthreads();
        SORT_TICK(1);
        // stable rank inside the wave's 64 * E pairs: peers = lanes of this round with my digit (8 ballots), the first
        // of them advances the wave's counter of that digit by their number
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool valid = t0 + wave * (64 * E) + e * 64 + lane < n;
            const unsigned dg = (unsigned)(key[e] >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((dg >> b) & 1u);
                peers &= ((dg >> b) & 1u) ? bal : ~bal;
            }
            const unsigned before = __popcll(peers & lt);
            const unsigned old = wave_hist[wave][dg];
            if (valid && before == 0) wave_hist[wave][dg] = old + (unsigned)__popcll(peers);
            rank[e] = old + before;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the counter update before the next round's read
        }
        __syncthreads();
        SORT_TICK(2);
        // bin tid: counts of the four waves -> tile count, exclusive offsets of the waves
        unsigned tile_count = 0;
        if (bin_thread) {
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) { const unsigned c = wave_hist[w][tid]; wave_hist[w][tid] = tile_count; tile_count += c; }
        }
        unsigned long long* mine = look + (size_t)tile * kSortBins + (bin_thread ? tid : 0);
        const unsigned long long tagw = (unsigned long long)tag << 32;
        if (bin_thread) sort_look_store(mine, tagw | (tile == 0 ? kLookPrefix : kLookAgg) | tile_count);
        const unsigned start = sort_block_excl_scan<BLOCK>(tile_count, part);      // (threads beyond the bins add zeros behind them)
        SORT_TICK(3);
        // decoupled look-back over the tiles before this one, 4 words in flight
        unsigned excl = 0;
        if (tile > 0 && bin_thread) {
            int at = tile - 1, spins = 0;
            bool found = false;
            while (!found) {
                unsigned long long w[kSortLookW];
#pragma unroll
                for (int j = 0; j < kSortLookW; ++j) w[j] = at - j >= 0 ? sort_look_load(look + (size_t)(at - j) * kSortBins + tid) : 0ull;
                int used = 0;
#pragma unroll
                for (int j = 0; j < kSortLookW; ++j) {
                    // ready = written by THIS pass of THIS sort (tag) with one of the two flags set
                    if (found || used != j || at - j < 0 || (unsigned)(w[j] >> 32) != tag || !(w[j] & (kLookAgg | kLookPrefix))) continue;
                    excl += (unsigned)(w[j] & kLookCount);
                    ++used;
                    found = (w[j] & kLookPrefix) != 0;
                }
                at -= used;
                if (!found && used == 0) {
                    if (++spins > kSortSpinLimit) { st->error = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            sort_look_store(mine, tagw | kLookPrefix | (excl + tile_count));
        }
        if (bin_thread) {
            bin_start[tid] = start;
            gbase[tid] = (int)(digit_base + excl) - (int)start;
        }
        __syncthreads();
        SORT_TICK(4);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (t0 + wave * (64 * E) + e * 64 + lane < n) {
                const unsigned dg = (unsigned)(key[e] >> shift) & 255u;
                const unsigned p = bin_start[dg] + wave_hist[wave][dg] + rank[e];
                skey[p] = key[e];
                sval[p] = val[e];
            }
        }
        __syncthreads();
        SORT_TICK(5);
        const int nvalid = min(T, n - t0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int p = e * BLOCK + tid;
            if (p < nvalid) {
                const uint64_t kk = skey[p];
                const int at = gbase[(unsigned)(kk >> shift) & 255u] + p;
                if ((unsigned)at < (unsigned)n) { dst_k[at] = kk; dst_v[at] = sval[p]; }     // (always, unless a wait gave up)
            }
        }
    }

    __syncthreads();
    SORT_TICK(6);
#ifdef NBODY_WIDE_TIMING
    if (threadIdx.x == 0 && work) atomicAdd(&st->dbg_cnt, 1ull);
#endif
    if (k == kSortDigits - 1) {
        // the sort is over when every workgroup of the last pass has come this far: the last one opens the next epoch
        if (tid == 0) {
            const int before = atomicAdd(&st->done_last, 1);
            if (before == (int)gridDim.x - 1) { st->done_last = 0; st->epoch = epoch + 1; }
        }
    }
}

// ---- truncated sort -------------------------------------------------------------------------------------------------
// The keys the build sorts are quadrant paths: two bodies share the top 19 levels (bits 24-61; a cell 800 / 2^19 = 1.5e-3
// wide at spread 400) about 15 times in the 10^6-body disc and a few 10^4 times in the 2^24-body Plummer core.  So the
// three low digit positions are NOT swept: five passes order the pairs by bits 24-63 (stable: inside a run of equal
// prefixes the pairs keep their input order), then sort_fix_runs_kernel — one thread per pair, one coalesced read of
// the keys — finishes every run: a run of up to kSortShortRun pairs by a stable insertion sort of its head thread, a
// longer one (bodies crowded into one such cell: adversarial inputs) goes to a queue that sort_long_runs_kernel
// empties, one WAVE per run: three stable counting passes over the low 24 bits (wave-private LDS histogram, the same
// peer ranking as sort_pass_kernel), ping-pong through the sort's scratch arrays.  The result is the full 64-bit stable order.
// (Keys with bit 63 set — bodies outside the root box — carry their body index in bits 24-51, bh_keys_kernel: unique
// prefixes, never a run.)
constexpr int kSortLowDigits = 3;           // digit positions 0..2 = bits 0-23
constexpr int kSortShortRun = 32;
constexpr int kSortQueueCap = 1 << 16;
constexpr int kSortLongBlocks = 64;

// low63: if not null, receives the number of keys with bit 63 clear (= where they end in the sorted order: the caller's
// "valid" count, one 1-thread launch less in the Barnes-Hut build).
__global__ __launch_bounds__(kSortBlock) void sort_fix_runs_kernel(uint64_t* keys, int* vals, int n, SortState* __restrict__ st,
                                                                  int* __restrict__ queue, int* __restrict__ low63)
{
    const int i = blockIdx.x * kSortBlock + threadIdx.x;
    if (i >= n) return;
    constexpr int sh = 8 * kSortLowDigits;
    const uint64_t p = keys[i] >> sh;
    // (a neighbour's key may be mid-move by the thread that sorts its run: the bits above `sh` are the same in every
    // pair of a run and an 8-byte store is one instruction, so what is read here is right either way)
    const uint64_t before = i > 0 ? keys[i - 1] >> sh : 0;
    if (low63) {                                                // exactly one thread sees the boundary
        const bool hi_here = (p >> (63 - sh)) != 0, hi_before = i > 0 && (before >> (63 - sh)) != 0;
        if (hi_here && !hi_before) *low63 = i;
        if (i == n - 1 && !hi_here) *low63 = n;
    }
    if (i > 0 && before == p) return;                           // not the head of a run
    if (i + 1 >= n || (keys[i + 1] >> sh) != p) return;         // a run of one: nearly everybody
    int len = 2;
    while (i + len < n && len <= kSortShortRun && (keys[i + len] >> sh) == p) ++len;
    if (len > kSortShortRun) {
        const int slot = atomicAdd(&st->long_runs, 1);
        if (slot < kSortQueueCap) queue[slot] = i; else st->error = 2;
        return;
    }
    for (int a = 1; a < len; ++a) {                             // stable insertion sort of the run, in place
        const uint64_t kk = keys[i + a];
        const int vv = vals[i + a];
        int b = a;
        while (b > 0 && keys[i + b - 1] > kk) { keys[i + b] = keys[i + b - 1]; vals[i + b] = vals[i + b - 1]; --b; }
        keys[i + b] = kk;
        vals[i + b] = vv;
    }
}

__global__ __launch_bounds__(kSortBlock) void sort_long_runs_kernel(uint64_t* keys, int* vals, uint64_t* tmp_k, int* tmp_v, int n,
                                                                   const SortState* __restrict__ st, const int* __restrict__ queue)
{
    __shared__ unsigned h[kSortBlock / 64][kSortBins];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int runs = min(st->long_runs, kSortQueueCap);
    constexpr int sh = 8 * kSortLowDigits;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int q = blockIdx.x * (kSortBlock / 64) + wave; q < runs; q += gridDim.x * (kSortBlock / 64)) {
        const int s0 = queue[q];
        const uint64_t p = keys[s0] >> sh;
        int len = 0;                                            // wave-uniform: 64 candidates per step
        for (;;) {
            const int j = s0 + len + lane;
            const unsigned long long same = __ballot(j < n && (keys[j] >> sh) == p);
            if (same == ~0ull) { len += 64; continue; }
            len += __builtin_ctzll(~same);
            break;
        }
        uint64_t* sk = keys;  int* sv = vals;
        uint64_t* dk = tmp_k; int* dv = tmp_v;
        for (int pass = 0; pass <= kSortLowDigits; ++pass) {
            // other lanes' stores of the pass before are read now: they have landed (vmcnt) and the loads bypass the L1
            __builtin_amdgcn_s_waitcnt(0);
            if (pass == kSortLowDigits) {                       // three passes end in the scratch arrays: copy back
                for (int i = lane; i < len; i += 64) {
                    keys[s0 + i] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(sk + s0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vals[s0 + i] = __hip_atomic_load(sv + s0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            const int shift = 8 * pass;
            for (int b = lane; b < kSortBins; b += 64) h[wave][b] = 0;
            for (int i = lane; i < len; i += 64) {
                const uint64_t kk = __hip_atomic_load(reinterpret_cast<unsigned long long*>(sk + s0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&h[wave][(unsigned)(kk >> shift) & 255u], 1u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {   // exclusive scan of the 256 bins by the wave: 4 consecutive bins per lane
                unsigned c[4], tot = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) { c[u] = h[wave][4 * lane + u]; tot += c[u]; }
                unsigned incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned t = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += t;
                }
                unsigned run = incl - tot;
#pragma unroll
                for (int u = 0; u < 4; ++u) { h[wave][4 * lane + u] = run; run += c[u]; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i0 = 0; i0 < len; i0 += 64) {
                const int i = i0 + lane;
                const bool valid = i < len;
                const uint64_t kk = valid ? __hip_atomic_load(reinterpret_cast<unsigned long long*>(sk + s0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const int vv = valid ? __hip_atomic_load(sv + s0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                const unsigned dg = (unsigned)(kk >> shift) & 255u;
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const unsigned long long bal = __ballot((dg >> b) & 1u);
                    peers &= ((dg >> b) & 1u) ? bal : ~bal;
                }
                const unsigned before = __popcll(peers & lt);
                const unsigned old = h[wave][dg];
                if (valid && before == 0) h[wave][dg] = old + (unsigned)__popcll(peers);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (valid) { dk[s0 + old + before] = kk; dv[s0 + old + before] = vv; }
            }
            uint64_t* tk = sk; sk = dk; dk = tk;
            int* tv = sv; sv = dv; dv = tv;
        }
    }
}

struct SortBuffers {
    SortState* state = nullptr;
    unsigned long long* look = nullptr;      // sort_look_words(tiles) words, zero-initialised once
    unsigned* hist_part = nullptr;           // [kSortHistBlocks][8][256]
    int* queue = nullptr;                    // [kSortQueueCap] long runs of a truncated sort
    uint64_t* tmp_k = nullptr;
    int* tmp_v = nullptr;
    int tiles = 0;
};

constexpr int kSortTileBig = 4096, kSortTileSmall = 2048;
inline int sort_tile_pairs(int n)
{
    if (const char* e = std::getenv("NBODY_SORT_E")) return std::atoi(e) == 16 ? kSortTileBig : kSortTileSmall;   // developer knob (pairs per thread at 256 threads)
    return n >= (1 << 19) ? kSortTileBig : kSortTileSmall;   // measured at N = 10^6: 23.3 us per pass with 4096-pair tiles, 26.5 with 2048
}
inline int sort_block_threads()
{
    if (const char* e = std::getenv("NBODY_SORT_BLOCK")) { const int b = std::atoi(e); if (b == 256 || b == 512 || b == 1024) return b; }   // developer knob
    return kSortPassBlock;
}
inline int sort_tiles(int n) { const int t = sort_tile_pairs(n); return (n + t - 1) / t; }
inline size_t sort_hist_part_bytes() { return sizeof(unsigned) * (size_t)kSortHistBlocks * kSortDigits * kSortBins; }

// Enqueue the sort: 10 launches, or 9 when `truncated` (5 sweeps + the two run kernels).  tmp_k/tmp_v: n entries each.
// Returns true when *low63 has been (will have been) written: only the truncated sort has the kernel that does it.
inline bool sort_pairs(const SortBuffers& sb, const uint64_t* in_k, const int* in_v, uint64_t* out_k, int* out_v, int n, hipStream_t s,
                       bool truncated = false, int* low63 = nullptr)
{
    if (n <= 0) return false;
    if (std::getenv("NBODY_SORT_FULL")) truncated = false;                                      // developer/test switch
    const int first = truncated ? kSortLowDigits : 0;
    const int tiles = sort_tiles(n);
    int hb = (n + 8 * kSortBlock - 1) / (8 * kSortBlock);
    if (hb > kSortHistBlocks) hb = kSortHistBlocks;
    sort_hist_kernel<<<dim3(hb), dim3(kSortHistThreads), 0, s>>>(in_k, n, sb.hist_part, first);
    sort_plan_kernel<<<dim3(kSortDigits), dim3(1024), 0, s>>>(sb.hist_part, hb, n, sb.state, first);
    const bool big = sort_tile_pairs(n) == kSortTileBig;
    const int bt = sort_block_threads();
    for (int k = first; k < kSortDigits; ++k) {
#define NB_SORT_PASS(E_, B_) sort_pass_kernel<E_, B_><<<dim3(tiles), dim3(B_), 0, s>>>(sb.state, k, n, in_k, in_v, out_k, out_v, sb.tmp_k, sb.tmp_v, sb.look)
        if (bt == 1024)     { if (big) NB_SORT_PASS(4, 1024); else NB_SORT_PASS(2, 1024); }
        else if (bt == 512) { if (big) NB_SORT_PASS(8, 512);  else NB_SORT_PASS(4, 512); }
        else                { if (big) NB_SORT_PASS(16, 256); else NB_SORT_PASS(8, 256); }
#undef NB_SORT_PASS
    }
    if (truncated) {
        sort_fix_runs_kernel<<<dim3((n + kSortBlock - 1) / kSortBlock), dim3(kSortBlock), 0, s>>>(out_k, out_v, n, sb.state, sb.queue, low63);
        sort_long_runs_kernel<<<dim3(kSortLongBlocks), dim3(kSortBlock), 0, s>>>(out_k, out_v, sb.tmp_k, sb.tmp_v, n, sb.state, sb.queue);
    }
    return truncated && low63 != nullptr;
}

}  // namespace nbody
