/* nbody.h — C-ABI of the MI355X-native physics step for fejzo9/N-Body.
 *
 * The reference has no operator / plugin / FFI interface: its physics is inline in main()
 * between `physicsStart` and `physicsEnd` (benchmark.cpp:352-513, main.cpp:483-644) and works
 * in place on the BodyDataSoA vectors (benchmark.cpp:24-49), parameterised by the locals
 * useBarnesHut/useParallel/n/spread/mass/theta/dt (benchmark.cpp:300-306; the config.json keys of
 * main.cpp:374-390).  This header therefore DEFINES the drop-in seam: a host program replaces
 * that block by nbody_upload (once) + nbody_step (per frame) + nbody_download (when it needs the
 * positions, e.g. main.cpp:650-653).  INTEGRATION.md shows the edit.
 *
 * Conventions: plain C, no C++/torch types; every float array is caller-owned host memory of
 * length num_bodies unless the name says "device"; no alignment requirement (the reference uses
 * unaligned loads throughout, benchmark.cpp:375-401); one context is driven from one host thread;
 * errors are negative return codes + nbody_last_error(); nothing throws.  There is NO CPU
 * fallback: without a gfx950 device nbody_create fails with NBODY_ERR_NO_DEVICE.
 */
#ifndef NBODY_H_
#define NBODY_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: + nbody_sort_pairs_u64 (round 3).  2: + nbody_upload_state, nbody_comm_plan, nbody_profile_samples, nbody_bh_group_counters, nbody_accel_noalloc,
      nbody_config.bh_sorted (was reserved[0]); nbody_step(ctx, n > 1) on a partial shard is refused / exchanges (round 3) */
#define NBODY_ABI_VERSION 3

enum {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = -1,    /* bad argument / bad state            */
    NBODY_ERR_NO_DEVICE = -2,  /* no usable HIP device                */
    NBODY_ERR_HIP = -3,        /* a HIP runtime call failed           */
    NBODY_ERR_UNSUPPORTED = -4 /* feature not built for this config   */
};

typedef struct nbody_ctx nbody_ctx; /* opaque; owns all device memory */

/* Replaces the parameter block benchmark.cpp:300-306 / main.cpp:337-343 (+ config.json keys). */
typedef struct nbody_config {
    int32_t num_bodies;      /* numBodies                                                        */
    float   spread;          /* spread: Barnes-Hut root box is [-spread,spread]^2 (benchmark.cpp:169) */
    float   mass;            /* mass: informational (per-body masses come from nbody_upload)      */
    float   dt;              /* dt                                                               */
    float   theta;           /* theta                                                            */
    int32_t use_barnes_hut;  /* useBarnesHut: 0 = naive O(N^2) (benchmark.cpp:405), 1 = BH (:354).  BH limits: num_bodies
                                <= 2^28 - 64 (32-bit node indices; NBODY_ERR_INVALID beyond).  The reference masks a target's
                                own leaf by comparing body indices as FLOATS (benchmark.cpp:197-198, 233-235): the integer
                                comparison up to index 2^24 (every BASELINE configuration); with more bodies than that a
                                target also skips the neighbours whose index rounds to the same float.  ref_compat >= 1
                                reproduces it (a second instantiation of the walk kernels, picked when num_bodies > 2^24;
                                tests/test_gpu_bh_fullsize.py); ref_compat = 0 compares integers                           */
    int32_t use_parallel;    /* useParallel: in the BH branch 0 means "build tree, do nothing"
                                (benchmark.cpp:359); ignored by the naive branch, as in the reference */
    int32_t ref_compat;      /* 1: reproduce benchmark.cpp's quirks — bodies with index >= N-N%8 are never targets
                                nor integrated (benchmark.cpp:423,486,368) and the BH walk votes per group of 8
                                consecutive bodies (benchmark.cpp:264-266).
                                2: main.cpp's flavour of the same — identical, except that its Barnes-Hut branch lets
                                the N%8 tail bodies drift force-free, pos += vel*dt (main.cpp:535-544), and that its
                                tree normalises a centre of mass as sum * (1/mass) (main.cpp:173-175) where
                                benchmark.cpp divides twice (benchmark.cpp:139-140): a last-bit difference.
                                0: every body is a target; BH walk still uses the group-of-8 vote.                 */
    int32_t device;          /* HIP device ordinal; -1 = current device                           */
    /* i-body shard owned by this context (multi-GPU: one context per GPU/process).
       All num_bodies bodies are sources; targets are [shard_begin, shard_begin+shard_count).
       shard_count == 0 means the whole range.  Use nbody_shard_range() to compute it.            */
    int32_t shard_begin;
    int32_t shard_count;
    int32_t force_variant;   /* 0 = auto (chosen at upload from the m/radius arrays); 1 = uniform mass and a
                                uniform power-of-two radius >= 4 (pre-scaled coordinates, the reference's
                                only case: R = 8), 2 = uniform mass+radius >= 4, 3 = per-body mass and
                                radius >= 4, 4 = exact-general (any radius)                              */
    int32_t adaptive_box;    /* Barnes-Hut only.  0 (default, the reference): fixed root box [-spread,spread]^2, bodies
                                outside it exert no force (benchmark.cpp:83-85,169).  1 (opt-in, no reference
                                counterpart): every step the root box is [-h,h]^2 with
                                h = max_i max(|x_i|,|y_i|) over all finite positions, so no body escapes the tree */
    int32_t bh_fast_math;    /* Barnes-Hut only.  0 (default): IEEE sqrt/divisions, bit-identical to the plain-C oracle.
                                1: the GPU's 1-ulp sqrt/reciprocal in the walk; the accept/open VOTE is still decided
                                exactly (borderline ratios are re-evaluated in IEEE arithmetic), so the same nodes
                                are visited and accepted and only the summed terms differ, by a few ulp each      */
    int32_t bh_sorted;       /* Barnes-Hut only, needs ref_compat = 0.  0 (default, the reference): the walk votes over groups of
                                8 CONSECUTIVE BODY INDICES (benchmark.cpp:366-380), which in generator order are scattered
                                all over the box, so a node is accepted only when it is far from eight unrelated places.
                                1 (opt-in, no reference counterpart): the groups are 8 consecutive bodies in the tree's
                                own space-filling (quadrant path) order — the sort the build does anyway — so a group is
                                a compact patch.  A compact group accepts a node as soon as that one patch is far from it
                                (the reference's scattered group needs eight unrelated places to be), so the vote
                                threshold is 0.6 * theta: measured, that keeps the mean, 90th-percentile and maximum
                                error against the direct sum at or below the reference-order walk's at the same
                                nominal theta (the median within 6 %) while opening 1.6-2 x fewer nodes.  Same tree, same node kernel; WHICH bodies vote together differs, so the
                                accelerations are not the reference-order walk's bit for bit       */
    int32_t reserved[1];
} nbody_config;

/* Defaults = benchmark.cpp:300-306 (BH on, n=10000, spread=400, mass=1000, theta=1, dt=0.05). */
void nbody_default_config(nbody_config *cfg);

/* Contiguous i-shard of rank `rank` of `nranks`; boundaries are multiples of 8 so the reference's
   groups of 8 (benchmark.cpp:421) never straddle two GPUs.  Pure host arithmetic.                 */
int nbody_shard_range(int32_t num_bodies, int32_t nranks, int32_t rank, int32_t *begin, int32_t *count);

int nbody_create(const nbody_config *cfg, nbody_ctx **out);
void nbody_destroy(nbody_ctx *ctx);

/* Run all of the context's work on an existing HIP stream (hipStream_t passed as void*), e.g. the
   stream a framework's collectives are ordered against.  NULL restores the context's own stream. */
int nbody_set_stream(nbody_ctx *ctx, void *hip_stream);

/* = BodyDataSoA::m/posX/posY/velX/velY/radius .data() (benchmark.cpp:28-33).  Synchronous. */
int nbody_upload(nbody_ctx *ctx, const float *m, const float *posX, const float *posY,
                 const float *velX, const float *velY, const float *radius);

/* New positions and velocities for the bodies of the last nbody_upload (same m and radius): skips the host-side scan
   of the m/radius arrays, the variant choice and the rebuild of the per-body clamp constants.  Synchronous. */
int nbody_upload_state(nbody_ctx *ctx, const float *posX, const float *posY, const float *velX, const float *velY);

/* Enqueue nsteps iterations of the block benchmark.cpp:354-511 (force + forward Euler).  Returns
   once enqueued.  In a sharded context only the owned i-range is advanced and the position arrays have to be
   all-gathered between steps: nsteps == 1 leaves that to the caller (nbody_device_ptr + its own collective, or
   nbody_exchange_positions); nsteps > 1 runs nsteps x (step + nbody_exchange_positions) when a communicator is
   attached (nbody_comm_init) and is REFUSED with NBODY_ERR_INVALID when none is (the other ranks' positions would
   be stale from the second step on). */
int nbody_step(nbody_ctx *ctx, int32_t nsteps);

/* Only the force phase (accelerations of the current positions), no integration. */
int nbody_compute_accel(nbody_ctx *ctx);

int nbody_sync(nbody_ctx *ctx);

/* Copies the state back (syncs).  Any pointer may be NULL.  main.cpp:650-653 needs posX/posY only. */
int nbody_download(nbody_ctx *ctx, float *posX, float *posY, float *velX, float *velY);

/* Native position exchange for sharded contexts (no reference counterpart): RCCL straight from the
   library, for hosts without a framework.  One process per GPU: rank 0 calls nbody_comm_unique_id and hands the 128
   bytes to the other ranks by any means (the C++ driver uses a TCP socket); every rank calls nbody_comm_init on a
   context whose shard is nbody_shard_range(N, nranks, rank); then, after each nbody_step(ctx, 1),
   nbody_exchange_positions enqueues the all-gather of posX/posY (in-place ncclAllGather; one ncclBroadcast per
   owner when the shards are ragged) on the context's stream.  RCCL is loaded at run time (librccl.so, or the path in
   NBODY_RCCL_LIB); NBODY_ERR_UNSUPPORTED if it cannot be.  Frameworks that own a process group can instead gather
   nbody_device_ptr(ctx, 1|2) themselves, as n-body_amd/sharded.py does with torch.distributed. */
#define NBODY_COMM_ID_BYTES 128
int nbody_comm_unique_id(char out[NBODY_COMM_ID_BYTES]);
int nbody_comm_init(nbody_ctx *ctx, int32_t rank, int32_t nranks, const char id[NBODY_COMM_ID_BYTES]);
int nbody_exchange_positions(nbody_ctx *ctx);
/* The plan nbody_exchange_positions follows, as pure host arithmetic (no GPU, no RCCL): every rank's span
   (nbody_shard_range) and whether the spans are equal and tile the array, i.e. ONE in-place ncclAllGather per
   array (1) or one ncclBroadcast per owner (0, ragged num_bodies).  begins/counts: nranks entries each. */
int nbody_comm_plan(int32_t num_bodies, int32_t nranks, int32_t *begins, int32_t *counts, int32_t *in_place_allgather);

/* Position streaming for a viewer (the consumer is the per-frame draw loop main.cpp:646-655,
   which reads posX/posY only).  nbody_frame_publish enqueues, behind the steps enqueued so far, a device-side
   snapshot of posX/posY and its asynchronous copy into library-owned PINNED host memory on a separate copy
   stream; it returns at once and later steps do not wait for the copy.  nbody_frame_acquire blocks until the
   OLDEST published, not yet acquired frame has landed and returns its arrays (length num_bodies) and the step
   count it shows; the pointers stay valid until two further frames have been published.  At most 2 frames may
   be in flight: a third publish without an acquire fails with NBODY_ERR_INVALID. */
int nbody_frame_publish(nbody_ctx *ctx);
int nbody_frame_acquire(nbody_ctx *ctx, const float **posX, const float **posY, int64_t *step_index);

/* Parity hook: the accelerations evaluated by the last nbody_step / nbody_compute_accel, i.e. the
   reference's tmp_accX/tmp_accY (benchmark.cpp:412-413, 476-477).  Entries outside the owned shard
   (and, in ref_compat, the N%8 tail) are 0.  Syncs. */
int nbody_get_accel(nbody_ctx *ctx, float *accX, float *accY);

/* {Ekin, Epot, Px, Py, Lz} of the current state, f64 accumulation on the GPU.  The reference has
   no counterpart; Epot uses the potential consistent with its clamped force.  Syncs. */
int nbody_diagnostics(nbody_ctx *ctx, double out[5]);

/* Device pointers of the SoA state for zero-copy interop (RCCL all-gather of the position arrays,
   SURVEY.md §8e).  which: 0 m, 1 posX, 2 posY, 3 velX, 4 velY, 5 radius, 6 accX, 7 accY. */
void *nbody_device_ptr(nbody_ctx *ctx, int32_t which);

/* Per-kernel timing with hipEvents on the context's stream.  which: 0 = force (naive pair kernel or
   BH walk), 1 = integrate, 2 = tree build (BH).  Accumulates while enabled. */
int nbody_profile_enable(nbody_ctx *ctx, int32_t on);
int nbody_profile_read(nbody_ctx *ctx, int32_t which, double *total_ms, int64_t *launches);
int nbody_profile_reset(nbody_ctx *ctx);
/* Duration in ms of every launch of that class since the last reset, in launch order: copies at most cap values,
   returns how many there are (< 0: error).  Lets a harness report min/median/max, not only the mean.  The library
   keeps the most recent 32 768 - 65 536 samples per class (older ones are dropped, so a host that leaves profiling on
   for days does not grow); totals and launch counts of nbody_profile_read are not affected. */
int64_t nbody_profile_samples(nbody_ctx *ctx, int32_t which, float *out_ms, int64_t cap);

/* Introspection for tests/bench: which force variant upload selected (1..4), grid geometry. */
int nbody_get_info(nbody_ctx *ctx, int32_t *force_variant, int32_t *j_slices, int32_t *blocks);

/* Barnes-Hut tree read-back (parity hook for QuadTree::pool / lastPoolIndex, benchmark.cpp:73-74).
   Returns the number of nodes of the last build; copies at most cap 48-byte QuadNode records
   (layout benchmark.cpp:52-60) when out != NULL.  Node ORDER differs from the reference's insertion
   order; compare as a set / via canonical traversal. */
int64_t nbody_bh_tree_nodes(nbody_ctx *ctx, void *out, int64_t cap);
/* {node visits, accepted nodes} summed over all groups of the last BH walk. */
int nbody_bh_counters(nbody_ctx *ctx, int64_t out[2]);

/* The same two counters per group of 8 consecutive bodies (index = first body / 8; ceil(num_bodies/8) entries each;
   groups outside the owned shard and the N%8 tail group keep 0): what the reference's vote (benchmark.cpp:259-266)
   decides group by group.  Parity hook against the counters oracle/build_ref.sh exports from the compiled reference
   at benchmark.cpp:245 (node popped) and :267 (node accepted).  Syncs. */
int nbody_bh_group_counters(nbody_ctx *ctx, int32_t *visits, int32_t *accepts);

/* NO COUNTERPART IN THE REFERENCE.  The force phase for callers that cannot allocate device memory: code under HIP
   graph capture, tight serving loops, harnesses that forbid hipMalloc.  It is nbody_compute_accel + nbody_get_accel
   of a context with ref_compat = 1 and force_variant = 4 (exact-general formula: any mass, any radius) — same
   kernels, same launch plan, bit-identical accelerations — but stateless: every array is a DEVICE pointer of length
   num_bodies owned by the caller (inputs m, posX, posY, radius; outputs accX, accY; the N%8 tail gets 0,
   benchmark.cpp:423), work is enqueued on hip_stream (NULL = the null stream) of the CURRENT device and the call
   returns without synchronising.  It calls no hipMalloc/hipFree: the j-slice partial sums and the per-body clamp
   constants live in a 33.8 MB array that the library holds statically on every GPU it runs on.  Limits:
   num_bodies <= 65536 (NBODY_ERR_INVALID beyond); all calls on one GPU share that array, so calls that overlap on
   one GPU (different streams, different threads) race — serialise them.  The FIRST call on a GPU resolves the
   array's address (hipGetSymbolAddress, which may load the code object): make one warm-up call per GPU OUTSIDE any
   stream capture; every later call only enqueues kernels and can be captured (tests/test_zz_no_alloc.py replays it
   from a hipGraph).  The launch plan honours the same developer knobs (NBODY_FORCE_K/SLICES/TILE) as a context's. */
int nbody_accel_noalloc(int32_t num_bodies, const float *dev_m, const float *dev_posX, const float *dev_posY,
                        const float *dev_radius, float *dev_accX, float *dev_accY, void *hip_stream);

/* Parity hook for the Barnes-Hut build's own device sort (csrc/bh_sort.hip.h; it replaces the order the reference gets
   from serial insertion, benchmark.cpp:171-172): sorts n (key, value) pairs by key, STABLY, on the current device —
   host arrays in and out, `repeats` sorts in a row on the same state (epochs, tickets and histograms carry over as
   they do from build to build).  truncated != 0: the form the build uses — five sweeps over bits 24-63, then the runs
   of equal prefixes are finished by the two run kernels; the result is the same full 64-bit stable order.  Not on any
   step path. */
int nbody_sort_pairs_u64(const uint64_t *keys, const int32_t *vals, int32_t n, uint64_t *out_keys, int32_t *out_vals,
                         int32_t repeats, int32_t truncated);

const char *nbody_last_error(void); /* thread-local, never NULL */
int nbody_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_H_ */
