// comm_rccl.hip.h — native position exchange for sharded contexts: RCCL (ncclAllGather over xGMI) called straight
// from the C-ABI, so that a C++ host (n-body_amd/host/benchmark_main.cpp --dist) scales over the GPUs of a node
// without Python.  SURVEY.md §8e; the reference has no counterpart (it is one OpenMP process).
//
// RCCL is bound at run time (dlopen) rather than at link time: a process that never shards does not need the
// library, and a process that already carries an RCCL (PyTorch ships its own copy) keeps using that one.
// Included by nbody_capi.hip after the definition of nbody_ctx.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    bool tried = false;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && AllGather && Broadcast && GroupStart && GroupEnd; }
};

RcclApi& rccl_api()
{
    static RcclApi api;
    if (api.tried) return api;
    api.tried = true;
    const char* names[] = {std::getenv("NBODY_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        if (!nm || !*nm) continue;
        api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) return api;
#define NB_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, #sym))
    NB_SYM(GetUniqueId, ncclGetUniqueId);
    NB_SYM(CommInitRank, ncclCommInitRank);
    NB_SYM(CommDestroy, ncclCommDestroy);
    NB_SYM(AllGather, ncclAllGather);
    NB_SYM(Broadcast, ncclBroadcast);
    NB_SYM(GroupStart, ncclGroupStart);
    NB_SYM(GroupEnd, ncclGroupEnd);
    NB_SYM(GetErrorString, ncclGetErrorString);
#undef NB_SYM
    return api;
}

int rccl_fail(const char* what, ncclResult_t r)
{
    RcclApi& api = rccl_api();
    return fail(NBODY_ERR_HIP, "%s failed: %s", what, api.GetErrorString ? api.GetErrorString(r) : "RCCL error");
}

#define RCCL_TRY(expr)                                      \
    do {                                                    \
        ncclResult_t r_ = (expr);                           \
        if (r_ != ncclSuccess) return rccl_fail(#expr, r_); \
    } while (0)

int rccl_require()
{
    if (!rccl_api().ok()) {
        const char* why = dlerror();        // once: dlerror() clears the message it returns
        return fail(NBODY_ERR_UNSUPPORTED, "RCCL could not be loaded (librccl.so; set NBODY_RCCL_LIB to its path): %s",
                    why ? why : "symbols missing");
    }
    return NBODY_OK;
}

}  // namespace

extern "C" {

int nbody_comm_unique_id(char out[NBODY_COMM_ID_BYTES])
{
    if (!out) return fail(NBODY_ERR_INVALID, "nbody_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == NBODY_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (int rc = rccl_require()) return rc;
    ncclUniqueId id;
    RCCL_TRY(rccl_api().GetUniqueId(&id));
    std::memcpy(out, &id, sizeof(id));
    return NBODY_OK;
}

int nbody_comm_plan(int32_t num_bodies, int32_t nranks, int32_t* begins, int32_t* counts, int32_t* in_place_allgather)
{
    if (nranks < 1 || !begins || !counts) return fail(NBODY_ERR_INVALID, "nbody_comm_plan: bad arguments");
    bool equal = true;
    for (int r = 0; r < nranks; ++r) {
        if (int rc = nbody_shard_range(num_bodies, nranks, r, &begins[r], &counts[r])) return rc;
        equal = equal && counts[r] == counts[0];
    }
    equal = equal && (int64_t)counts[0] * nranks == num_bodies;
    if (in_place_allgather) *in_place_allgather = equal ? 1 : 0;
    return NBODY_OK;
}

int nbody_comm_init(nbody_ctx* ctx, int32_t rank, int32_t nranks, const char id[NBODY_COMM_ID_BYTES])
{
    if (!ctx || !id) return fail(NBODY_ERR_INVALID, "nbody_comm_init: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(NBODY_ERR_INVALID, "nbody_comm_init: rank %d of %d", rank, nranks);
    if (ctx->comm) return fail(NBODY_ERR_INVALID, "nbody_comm_init: the context already has a communicator");
    // the exchange is defined on the shard plan of nbody_shard_range: this context must own exactly its rank's share
    ctx->comm_spans.assign((size_t)nranks, {0, 0});
    {
        std::vector<int32_t> b((size_t)nranks), c((size_t)nranks);
        int32_t eq = 0;
        if (int rc = nbody_comm_plan(ctx->cfg.num_bodies, nranks, b.data(), c.data(), &eq)) return rc;
        for (int r = 0; r < nranks; ++r) ctx->comm_spans[(size_t)r] = {b[(size_t)r], c[(size_t)r]};
        ctx->comm_equal = eq != 0;
    }
    const int my_b = ctx->cfg.shard_begin;
    const int my_c = ctx->cfg.shard_count == 0 ? ctx->cfg.num_bodies - my_b : ctx->cfg.shard_count;
    if (ctx->comm_spans[(size_t)rank].first != my_b || ctx->comm_spans[(size_t)rank].second != my_c)
        return fail(NBODY_ERR_INVALID, "nbody_comm_init: context owns [%d,+%d) but rank %d of %d owns [%d,+%d) (nbody_shard_range)",
                    my_b, my_c, rank, nranks, ctx->comm_spans[(size_t)rank].first, ctx->comm_spans[(size_t)rank].second);
    if (int rc = rccl_require()) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    RCCL_TRY(rccl_api().CommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return NBODY_OK;
}

int nbody_exchange_positions(nbody_ctx* ctx)
{
    if (!ctx) return fail(NBODY_ERR_INVALID, "nbody_exchange_positions: null context");
    if (!ctx->comm) return fail(NBODY_ERR_INVALID, "nbody_exchange_positions: nbody_comm_init has not been called");
    if (!ctx->uploaded) return fail(NBODY_ERR_INVALID, "nbody_exchange_positions: nothing uploaded");
    HIP_TRY(hipSetDevice(ctx->device));
    RcclApi& api = rccl_api();
    ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
    const auto& sp = ctx->comm_spans;
    const bool equal = ctx->comm_equal;                   // nbody_comm_plan: equal shards that tile the array
    float* arrays[2] = {ctx->px, ctx->py};
    RCCL_TRY(api.GroupStart());
    // Inside the group bracket nothing returns early: a failed call is remembered, the bracket is ALWAYS closed (an
    // open group would silently defer every later collective of this thread), and the first failure is reported.
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    for (float* a : arrays) {
        if (equal) {
            // in place: the send buffer is this rank's slot of the receive buffer
            const size_t cnt = (size_t)sp[0].second;
            const ncclResult_t r = api.AllGather(a + (size_t)ctx->comm_rank * cnt, a, cnt, ncclFloat, comm, ctx->stream);
            if (r != ncclSuccess && first == ncclSuccess) { first = r; what = "ncclAllGather"; }
        } else {
            for (int k = 0; k < ctx->comm_nranks; ++k) {      // ragged shards: one broadcast per owner
                if (sp[(size_t)k].second == 0) continue;
                float* p = a + sp[(size_t)k].first;
                const ncclResult_t r = api.Broadcast(p, p, (size_t)sp[(size_t)k].second, ncclFloat, k, comm, ctx->stream);
                if (r != ncclSuccess && first == ncclSuccess) { first = r; what = "ncclBroadcast"; }
            }
        }
    }
    const ncclResult_t end = api.GroupEnd();
    if (first != ncclSuccess) return rccl_fail(what, first);
    if (end != ncclSuccess) return rccl_fail("ncclGroupEnd", end);
    return NBODY_OK;
}

}  // extern "C"
