// bh_sort.hip.h — this repository's own device sort and scan for the Barnes-Hut build (gfx950).
//
// Replaces what the reference does serially at benchmark.cpp:156-173 (insertion order) on the GPU side of the build:
// rounds 1-2 used hipcub::DeviceRadixSort / DeviceScan here (12 rocPRIM launches, 221 of the 485 us of the N = 10^6
// build).  Two pieces:
//
//  * sort_pairs: stable LSD radix sort of (64-bit key, int value) pairs, 8-bit digits.
//      sort_hist_kernel   ONE pass over the keys fills the histograms of all 8 digit positions (LDS atomics per
//                         workgroup, then one global add per non-empty bin);
//      sort_pass_kernel   one launch per digit position.  A position whose histogram is a single bin moves nothing and is
//                         skipped ON THE DEVICE (the launch sequence is static, so that it can sit in a hipGraph): every
//                         workgroup derives from the histograms which positions are live, which buffer this pass reads
//                         and which it writes, so that the LAST live pass lands in the output arrays.  A live pass is
//                         "one sweep": a workgroup takes the next tile (ticket), ranks its 256 x E pairs stably in LDS
//                         (per wave: 8 ballots give every lane the peers that share its digit; per-wave counters; a
//                         256-bin scan), publishes the tile's bin counts, obtains the counts of all tiles before it by
//                         decoupled look-back (flag, count and the epoch of this sort packed in ONE 64-bit word per
//                         (tile, bin), written and read with device-scope relaxed atomics: nothing to fence, nothing to
//                         clear between sorts) and scatters through LDS in runs of equal digits.
//    Everything is deterministic: tickets decide WHICH workgroup sorts a tile, never where an element goes.
//    Every wait is bounded: a workgroup only ever waits for tiles with smaller tickets — held by workgroups that are
//    already running and publish before they wait themselves — and gives up after ~0.3 s with SortState::error set.
//
//  * bh_cells_scan_kernel: cells owned per sorted instance (bh_count_one), their exclusive scan (= cell ids) and the
//    owner scatter in ONE launch (was bh_count_kernel + rocPRIM's scan + bh_owner_kernel), same look-back scheme with
//    one value per tile, examined a wave (64 predecessor tiles) at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

namespace nbody {

constexpr int kSortBlock = 256;             // 4 waves
constexpr int kSortBins = 256;              // 8-bit digits
constexpr int kSortDigits = 8;
constexpr int kSortHistBlocks = 512;
constexpr int kSortSpinLimit = 1 << 21;     // polls (each >= 128 cycles of s_sleep) before a look-back gives up
constexpr unsigned long long kLookAgg = 1ull << 30, kLookPrefix = 2ull << 30, kLookCount = (1ull << 30) - 1;

struct SortState {                          // device-resident, one per sort instance; zero-initialised once, then epoch = 1
    unsigned epoch;                         // of the sort in progress (>= 1: no tag ever equals a zero-initialised word's);
                                            // bumped when the sort's last pass has finished
    int tickets[kSortDigits];               // next tile of each pass; pass k clears the one of pass k-1
    int done_last;                          // workgroups of the last pass that have finished
    int error;                              // 1: a look-back gave up (results are then garbage; reported, never silent)
    unsigned hist[2][kSortDigits][kSortBins];   // [epoch & 1]: pass 0 clears the other half for the next sort
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

// exclusive scan of one value per thread over a 256-thread workgroup; `part` is 4 words of LDS.  Ends with a barrier.
__device__ __forceinline__ unsigned sort_block_excl_scan(unsigned v, unsigned* part, unsigned* total)
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
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kSortBlock / 64; ++w) { const unsigned p = part[w]; if (w < wave) before += p; all += p; }
    if (total) *total = all;
    __syncthreads();
    return before + incl - v;
}

__global__ __launch_bounds__(kSortBlock) void sort_hist_kernel(const uint64_t* __restrict__ keys, int n, SortState* __restrict__ st)
{
    __shared__ unsigned h[kSortDigits][kSortBins];
    const int tid = threadIdx.x;
    for (int q = tid; q < kSortDigits * kSortBins; q += kSortBlock) (&h[0][0])[q] = 0;
    __syncthreads();
    const int per = ((n + (int)gridDim.x - 1) / (int)gridDim.x + kSortBlock - 1) / kSortBlock * kSortBlock;
    const int b0 = (int)blockIdx.x * per, b1 = min(n, b0 + per);
    for (int i0 = b0; i0 < b1; i0 += 4 * kSortBlock) {
        uint64_t k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * kSortBlock + tid; k[u] = i < b1 ? keys[i] : 0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * kSortBlock + tid < b1) {
#pragma unroll
                for (int d = 0; d < kSortDigits; ++d) atomicAdd(&h[d][(unsigned)(k[u] >> (8 * d)) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    unsigned* g = &st->hist[st->epoch & 1][0][0];
#pragma unroll
    for (int d = 0; d < kSortDigits; ++d) {
        const unsigned c = h[d][tid];
        if (c) atomicAdd(&g[d * kSortBins + tid], c);       // read by the NEXT kernels only (kernel boundary)
    }
}

// One digit position.  E pairs per thread, tile = 256 * E.  Buffers: IN (read-only), OUT (final result), TMP.
template <int E, int W = 4>
__global__ __launch_bounds__(kSortBlock) void sort_pass_kernel(SortState* __restrict__ st, int k, int n,
                                                              const uint64_t* __restrict__ in_k, const int* __restrict__ in_v,
                                                              uint64_t* __restrict__ out_k, int* __restrict__ out_v,
                                                              uint64_t* __restrict__ tmp_k, int* __restrict__ tmp_v,
                                                              unsigned long long* __restrict__ look)
{
    constexpr int T = kSortBlock * E;
    __shared__ unsigned wave_hist[kSortBlock / 64][kSortBins];
    __shared__ unsigned bin_start[kSortBins];
    __shared__ int gbase[kSortBins];
    __shared__ uint64_t skey[T];
    __shared__ int sval[T];
    __shared__ unsigned part[kSortBlock / 64];
    __shared__ unsigned s_mask;
    __shared__ int s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef NBODY_WIDE_TIMING
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned epoch = st->epoch;                      // stable during this kernel: bumped only by the last pass's last workgroup
    const unsigned* __restrict__ hist = &st->hist[epoch & 1][0][0];

    // housekeeping for the sorts to come (nobody in this launch reads what is cleared here)
    if (blockIdx.x == 0) {
        if (tid == 0) st->tickets[(k + kSortDigits - 1) % kSortDigits] = 0;
        if (k == 0) {
            unsigned* other = &st->hist[(epoch + 1) & 1][0][0];
            for (int q = tid; q < kSortDigits * kSortBins; q += kSortBlock) other[q] = 0;
        }
    }
    if (tid == 0) s_mask = 0;
    __syncthreads();
    // live digit positions: a position is dead when one bin holds every key
    {
        unsigned m = 0;
#pragma unroll
        for (int d = 0; d < kSortDigits; ++d)
            if (__ballot(hist[d * kSortBins + tid] == (unsigned)n) != 0ull) m |= 1u << d;
        if (lane == 0 && m) atomicOr(&s_mask, m);
    }
    __syncthreads();
    const unsigned live = ~s_mask & 0xFFu;
    const int P = __popc(live);
    bool work = n > 0 && ((live >> k) & 1u);
    int tile = 0;
    const uint64_t* src_k = in_k;
    const int* src_v = in_v;
    uint64_t* dst_k = out_k;
    int* dst_v = out_v;
    if (work) {
        const int idx = __popc(live & ((1u << k) - 1u));   // live passes before this one
        const bool to_out = ((P - 1 - idx) & 1) == 0;      // the last live pass writes OUT, the one before TMP, ...
        dst_k = to_out ? out_k : tmp_k;
        dst_v = to_out ? out_v : tmp_v;
        if (idx > 0) { src_k = to_out ? tmp_k : out_k; src_v = to_out ? tmp_v : out_v; }
        if (tid == 0) s_tile = atomicAdd(&st->tickets[k], 1);
        __syncthreads();
        tile = s_tile;
        work = (long long)tile * T < n;
    } else if (k == kSortDigits - 1 && P == 0 && n > 0) {
        // nothing to sort (one element, or all keys equal): the output is the input
        for (int i = (int)blockIdx.x * T + tid; i < min(n, ((int)blockIdx.x + 1) * T); i += kSortBlock) { out_k[i] = in_k[i]; out_v[i] = in_v[i]; }
    }

    SORT_TICK(0);
    if (work) {
        const int t0 = tile * T;
        const int shift = 8 * k;
        const unsigned tag = epoch * kSortDigits + (unsigned)k;
        for (int q = tid; q < (kSortBlock / 64) * kSortBins; q += kSortBlock) (&wave_hist[0][0])[q] = 0;
        uint64_t key[E];
        int val[E];
        unsigned rank[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = t0 + wave * (64 * E) + e * 64 + lane;       // tile order = memory order (stability)
            key[e] = i < n ? src_k[i] : ~0ull;
            val[e] = i < n ? src_v[i] : 0;
        }
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
#pragma unroll
        for (int w = 0; w < kSortBlock / 64; ++w) { const unsigned c = wave_hist[w][tid]; wave_hist[w][tid] = tile_count; tile_count += c; }
        unsigned long long* mine = look + (size_t)tile * kSortBins + tid;
        const unsigned long long tagw = (unsigned long long)tag << 32;
        sort_look_store(mine, tagw | (tile == 0 ? kLookPrefix : kLookAgg) | tile_count);
        const unsigned start = sort_block_excl_scan(tile_count, part, nullptr);
        const unsigned digit_base = sort_block_excl_scan(hist[k * kSortBins + tid], part, nullptr);
        SORT_TICK(3);
        // decoupled look-back over the tiles before this one, W words in flight
        unsigned excl = 0;
        if (tile > 0) {
            int at = tile - 1, spins = 0;
            bool found = false;
            while (!found) {
                unsigned long long w[W];
#pragma unroll
                for (int j = 0; j < W; ++j) w[j] = at - j >= 0 ? sort_look_load(look + (size_t)(at - j) * kSortBins + tid) : 0ull;
                int used = 0;
#pragma unroll
                for (int j = 0; j < W; ++j) {
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
        bin_start[tid] = start;
        gbase[tid] = (int)(digit_base + excl) - (int)start;
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
            const int p = e * kSortBlock + tid;
            if (p < nvalid) {
                const uint64_t kk = skey[p];
                const int at = gbase[(unsigned)(kk >> shift) & 255u] + p;
                if ((unsigned)at < (unsigned)n) { dst_k[at] = kk; dst_v[at] = sval[p]; }     // (always, unless a look-back gave up)
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
        __syncthreads();
        if (tid == 0) {
            const int before = atomicAdd(&st->done_last, 1);
            if (before == (int)gridDim.x - 1) { st->done_last = 0; st->epoch = epoch + 1; }
        }
    }
}

struct SortBuffers {
    SortState* state = nullptr;
    unsigned long long* look = nullptr;      // [tiles][256]
    uint64_t* tmp_k = nullptr;
    int* tmp_v = nullptr;
    int tiles = 0, per_thread = 8;
};

inline int sort_per_thread(int n)
{
    if (const char* e = std::getenv("NBODY_SORT_E")) return std::atoi(e) == 16 ? 16 : 8;       // developer knob
    return n > (1 << 21) ? 16 : 8;
}
inline int sort_tiles(int n) { const int t = kSortBlock * sort_per_thread(n); return (n + t - 1) / t; }

// Enqueue the 9 launches.  tmp_k/tmp_v: n entries each; look: sort_tiles(n) * 256 words, zero-initialised once.
inline void sort_pairs(const SortBuffers& sb, const uint64_t* in_k, const int* in_v, uint64_t* out_k, int* out_v, int n, hipStream_t s)
{
    if (n <= 0) return;
    const int tiles = sort_tiles(n);
    int hb = (n + 4 * kSortBlock - 1) / (4 * kSortBlock);
    int hb_max = kSortHistBlocks;
    if (const char* e = std::getenv("NBODY_SORT_HB")) hb_max = std::atoi(e) > 0 ? std::atoi(e) : hb_max;     // developer knob
    if (hb > hb_max) hb = hb_max;
    const bool wide = std::getenv("NBODY_SORT_W8") != nullptr;                                   // developer knob
    sort_hist_kernel<<<dim3(hb), dim3(kSortBlock), 0, s>>>(in_k, n, sb.state);
    for (int k = 0; k < kSortDigits; ++k) {
        if (sort_per_thread(n) == 16) {
            if (wide) sort_pass_kernel<16, 8><<<dim3(tiles), dim3(kSortBlock), 0, s>>>(sb.state, k, n, in_k, in_v, out_k, out_v, sb.tmp_k, sb.tmp_v, sb.look);
            else      sort_pass_kernel<16, 4><<<dim3(tiles), dim3(kSortBlock), 0, s>>>(sb.state, k, n, in_k, in_v, out_k, out_v, sb.tmp_k, sb.tmp_v, sb.look);
        } else {
            if (wide) sort_pass_kernel<8, 8><<<dim3(tiles), dim3(kSortBlock), 0, s>>>(sb.state, k, n, in_k, in_v, out_k, out_v, sb.tmp_k, sb.tmp_v, sb.look);
            else      sort_pass_kernel<8, 4><<<dim3(tiles), dim3(kSortBlock), 0, s>>>(sb.state, k, n, in_k, in_v, out_k, out_v, sb.tmp_k, sb.tmp_v, sb.look);
        }
    }
}

}  // namespace nbody
