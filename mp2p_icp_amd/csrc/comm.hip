// comm.hip -- RCCL inside the C boundary (mp2p_hip_comm_init; section 8e (ii)).
//
// One process per GPU.  The local layer is sharded in contiguous ranges of the visiting order, the map
// and its index replicated; per outer ICP iteration the ranks exchange
//   1. all-reduce MAX of 8 doubles {-min xyz, max xyz of the transformed local points, #claim records}
//   2. all-gather of the claim records (skipped when global re-use is allowed)
//   3. per Gauss-Newton inner iteration an all-reduce SUM of the 48 normal-equation sums
// on the context's stream (ncclAllReduce / ncclAllGather are stream-ordered: no host round trip between
// the two phases of the matcher; the record-list length is predicted from the previous iteration and
// checked after the solve).  The reference has no distributed code.
//
// librccl is opened at run time (dlopen) by mp2p_hip_comm_init: a single-GPU user never loads it.
// mp2p_hip_comm_init_hooks installs caller-provided collectives instead (another transport; tests run two
// contexts on one GPU through it, which RCCL itself refuses).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "device_utils.hpp"

namespace mp2p
{
struct RcclApi
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                        = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                 = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                           = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)     = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                       = nullptr;
};
static RcclApi g_rccl;

static int load_rccl(mp2p_hip_ctx* ctx)
{
    if (g_rccl.lib) return MP2P_HIP_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void*       h       = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return set_err(ctx, MP2P_HIP_ERR_HIP, "cannot open librccl: %s", dlerror());
#define MP2P_SYM(field, name)                                                                          \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                           \
    if (!g_rccl.field) return set_err(ctx, MP2P_HIP_ERR_HIP, "librccl lacks %s", name);
    MP2P_SYM(GetUniqueId, "ncclGetUniqueId")
    MP2P_SYM(CommInitRank, "ncclCommInitRank")
    MP2P_SYM(CommDestroy, "ncclCommDestroy")
    MP2P_SYM(AllReduce, "ncclAllReduce")
    MP2P_SYM(AllGather, "ncclAllGather")
    MP2P_SYM(GetErrorString, "ncclGetErrorString")
#undef MP2P_SYM
    g_rccl.lib = h;
    return MP2P_HIP_OK;
}

#define MP2P_TRY_NCCL(ctx, expr)                                                                       \
    do                                                                                                 \
    {                                                                                                  \
        ncclResult_t r__ = (expr);                                                                     \
        if (r__ != ncclSuccess)                                                                        \
            return set_err((ctx), MP2P_HIP_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
    } while (0)

// ---- the three collectives of the path, on the context's stream ----------------------------------
static int comm_allreduce_f64(mp2p_hip_ctx* ctx, double* buf, size_t n, int op /*0 sum, 1 max*/)
{
    Comm& c = ctx->comm;
    if (c.nranks <= 1) return MP2P_HIP_OK;
    if (c.hook_allreduce)
        return c.hook_allreduce(c.hook_user, buf, n, op, (void*)ctx->stream) ? set_err(ctx, MP2P_HIP_ERR_HIP, "all-reduce hook failed") : MP2P_HIP_OK;
    MP2P_TRY_NCCL(ctx, g_rccl.AllReduce(buf, buf, n, ncclFloat64, op ? ncclMax : ncclSum, (ncclComm_t)c.nccl, ctx->stream));
    return MP2P_HIP_OK;
}
static int comm_allgather_u64(mp2p_hip_ctx* ctx, const unsigned long long* send, unsigned long long* recv, size_t n)
{
    Comm& c = ctx->comm;
    if (c.hook_allgather)
        return c.hook_allgather(c.hook_user, send, recv, n, (void*)ctx->stream) ? set_err(ctx, MP2P_HIP_ERR_HIP, "all-gather hook failed") : MP2P_HIP_OK;
    MP2P_TRY_NCCL(ctx, g_rccl.AllGather(send, recv, n, ncclUint64, (ncclComm_t)c.nccl, ctx->stream));
    return MP2P_HIP_OK;
}

constexpr size_t COMM_CAP_QUANTUM = 4096;  // record lists are exchanged in multiples of this many records
static size_t round_cap(double n)
{
    const size_t v = (size_t)(n < 0 ? 0 : n);
    return std::max(COMM_CAP_QUANTUM, (v + COMM_CAP_QUANTUM - 1) / COMM_CAP_QUANTUM * COMM_CAP_QUANTUM);
}

// one sharded matcher call; cap = 0: read the exact record count back (one 64-byte device-to-host wait)
static int sharded_match(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud, const double pose[12],
                         const mp2p_hip_pt2pt_params* prm, mp2p_hip_pairs* out, size_t cap, size_t* cap_used)
{
    Comm& c = ctx->comm;
    if (c.nranks <= 1)
    {  // one GPU: the unsplit matcher (bounding-box reduction folded into the compaction)
        if (cap_used) *cap_used = 0;
        return mp2p_hip_match_pt2pt(ctx, map, cloud, pose, prm, nullptr, out);
    }
    int rc = mp2p_hip_match_pt2pt_phase1(ctx, map, cloud, pose, prm, nullptr);
    if (rc) return rc;
    {
        void * exch = nullptr, *list = nullptr;
        size_t list_len = 0;
        rc = mp2p_hip_exchange_pack(ctx, map, cloud, prm, &exch, &list, &list_len);
        if (rc) return rc;
        rc = comm_allreduce_f64(ctx, (double*)exch, 8, 1);
        if (rc) return rc;
        const bool claims = !prm->allowMatchAlreadyMatchedGlobalPoints;
        size_t     n_rec  = 0;
        if (claims)
        {
            if (cap == 0)
            {
                if (!ctx->pinned) MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 4096, hipHostMallocDefault));
                MP2P_TRY_HIP(ctx, hipMemcpyAsync(ctx->pinned, exch, 64, hipMemcpyDeviceToHost, ctx->stream));
                MP2P_TRY_HIP(ctx, stream_wait(ctx));
                cap = round_cap(((const double*)ctx->pinned)[6]);
            }
            // every rank sends `cap` records; a shard shorter than that pads with ~0 (= no record)
            const unsigned long long* send = (const unsigned long long*)list;
            if (list_len < cap)
            {
                MP2P_TRY_HIP(ctx, c.pad.ensure(cap));
                MP2P_TRY_HIP(ctx, hipMemsetAsync(c.pad.p, 0xFF, cap * 8, ctx->stream));
                if (list_len) MP2P_TRY_HIP(ctx, hipMemcpyAsync(c.pad.p, list, list_len * 8, hipMemcpyDeviceToDevice, ctx->stream));
                send = c.pad.p;
            }
            MP2P_TRY_HIP(ctx, c.gathered.ensure((size_t)c.nranks * cap));
            rc = comm_allgather_u64(ctx, send, c.gathered.p, cap);
            if (rc) return rc;
            n_rec = (size_t)c.nranks * cap;
        }
        rc = mp2p_hip_exchange_unpack(ctx, map, claims ? c.gathered.p : nullptr, n_rec);
        if (rc) return rc;
    }
    if (cap_used) *cap_used = cap;
    return mp2p_hip_match_pt2pt_phase2(ctx, map, cloud, prm, nullptr, out);
}

constexpr size_t COMM_PINNED_EXCH = 3072;  // where the 64 exchanged bytes land in the context's pinned page (gn_end uses [0, 512))

// also_exch: the all-reduced exchange block travels to the host together with the pose read-back that ends the
// step (one wait instead of two)
static int sharded_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose[12],
                         const mp2p_hip_gn_params* gn, mp2p_hip_gn_result* res, bool also_exch = false)
{
    if (ctx->comm.nranks <= 1) return mp2p_hip_gn_solve(ctx, pairs, pose, gn, res);
    int rc = mp2p_hip_gn_begin(ctx, pairs, pose, gn);
    if (rc) return rc;
    for (uint32_t it = 0; it < gn->maxInnerLoopIterations; it++)
    {
        if ((rc = mp2p_hip_gn_accumulate(ctx))) return rc;
        if ((rc = comm_allreduce_f64(ctx, (double*)mp2p_hip_gn_sums_ptr(ctx), MP2P_HIP_GN_NSUMS, 0))) return rc;
        if ((rc = mp2p_hip_gn_step(ctx))) return rc;  // every rank solves the same 6x6: no broadcast
    }
    if (also_exch)
    {
        if (!ctx->pinned) MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 4096, hipHostMallocDefault));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + COMM_PINNED_EXCH, ctx->exch.p, 64, hipMemcpyDeviceToHost, ctx->stream));
    }
    return mp2p_hip_gn_end(ctx, res);  // the pose read-back: its wait covers the copy above
}

// {-min xyz, max xyz} of ctx->local_bbox <-> the first six doubles of ctx->exch (MAX-reducible)
__global__ void bbox_to_exch_kernel(const float* __restrict__ bb, double* __restrict__ exch)
{
    const int d = threadIdx.x;
    if (d < 3) exch[d] = -(double)bb[d];
    else if (d < 6) exch[d] = (double)bb[d];
    else if (d < 8) exch[d] = 0.0;
}
__global__ void exch_to_bbox_kernel(const double* __restrict__ exch, float* __restrict__ bb)
{
    const int d = threadIdx.x;
    if (d < 3) bb[d] = (float)(-exch[d]);
    else if (d < 6) bb[d] = (float)exch[d];
}
}  // namespace mp2p

using namespace mp2p;

extern "C" {

int mp2p_hip_comm_get_unique_id(void* id_out)
{
    if (!id_out) return MP2P_HIP_ERR_INVALID;
    if (const int rc = load_rccl(nullptr)) return rc;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return set_err(nullptr, MP2P_HIP_ERR_HIP, "ncclGetUniqueId failed");
    static_assert(sizeof(id) == MP2P_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return MP2P_HIP_OK;
}

// side-effect-free pre-flight of mp2p_hip_comm_init: librccl loads and resolves, and the context has no communicator yet
// (drawing a throw-away ncclUniqueId for that, as round 4 did, starts a bootstrap listener thread and socket per call)
int mp2p_hip_comm_available(mp2p_hip_ctx* ctx)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    if (const int rc = load_rccl(ctx)) return rc;
    MP2P_REQUIRE(ctx, !ctx->comm.nccl && !ctx->comm.hook_allreduce, "the context already has a communicator");
    return MP2P_HIP_OK;
}

int mp2p_hip_comm_init(mp2p_hip_ctx* ctx, const void* unique_id, int rank, int nranks)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
    MP2P_REQUIRE(ctx, !ctx->comm.nccl && !ctx->comm.hook_allreduce, "the context already has a communicator");
    if (const int rc = load_rccl(ctx)) return rc;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    MP2P_TRY_NCCL(ctx, g_rccl.CommInitRank(&comm, nranks, id, rank));
    ctx->comm.nccl = comm, ctx->comm.rank = rank, ctx->comm.nranks = nranks, ctx->comm.cap_guess = 0;
    return MP2P_HIP_OK;
}

int mp2p_hip_comm_init_hooks(mp2p_hip_ctx* ctx, int rank, int nranks, mp2p_hip_allreduce_fn allreduce,
                             mp2p_hip_allgather_fn allgather, void* user)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, allreduce && allgather && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
    MP2P_REQUIRE(ctx, !ctx->comm.nccl && !ctx->comm.hook_allreduce, "the context already has a communicator");
    ctx->comm.hook_allreduce = allreduce, ctx->comm.hook_allgather = allgather, ctx->comm.hook_user = user;
    ctx->comm.rank = rank, ctx->comm.nranks = nranks, ctx->comm.cap_guess = 0;
    return MP2P_HIP_OK;
}

int mp2p_hip_comm_destroy(mp2p_hip_ctx* ctx)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    if (ctx->comm.nccl)
    {
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm.nccl);
    }
    ctx->comm.pad.release(), ctx->comm.gathered.release();
    ctx->comm = Comm();
    return MP2P_HIP_OK;
}

int mp2p_hip_comm_rank(const mp2p_hip_ctx* ctx) { return ctx ? ctx->comm.rank : -1; }
int mp2p_hip_comm_size(const mp2p_hip_ctx* ctx) { return ctx ? ctx->comm.nranks : 0; }

int mp2p_hip_comm_allreduce_f64(mp2p_hip_ctx* ctx, void* dev_buf, size_t n, int op_max)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, dev_buf || n == 0, "null buffer");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return comm_allreduce_f64(ctx, (double*)dev_buf, n, op_max ? 1 : 0);
}

int mp2p_hip_step_sharded(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                          const double pose[12], const mp2p_hip_pt2pt_params* prm, const mp2p_hip_gn_params* gn,
                          mp2p_hip_pairs* pairs, mp2p_hip_gn_result* out, int32_t* redone)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && cloud && pose && prm && gn && pairs && out, "null argument");
    // no communicator = one rank: the same call is the single-GPU step (one boundary crossing per
    // outer ICP iteration instead of three)
    MP2P_REQUIRE(ctx, prm->pairingsPerPoint == 1, "the sharded step implements pairingsPerPoint == 1");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    Comm& c = ctx->comm;
    if (redone) *redone = 0;
    const bool claims = c.nranks > 1 && !prm->allowMatchAlreadyMatchedGlobalPoints;
    for (int attempt = 0; attempt < 2; attempt++)
    {
        // one rank: the clear is folded into the compaction's scan kernel (its 64-byte memset was a launch of its own at the head of
        // every step); whatever path does not reach that kernel clears here, before it returns
        int rc = MP2P_HIP_OK;
        if (c.nranks <= 1) ctx->clear_deferred = pairs;
        else rc = mp2p_hip_pairs_clear(ctx, pairs);
        if (rc) return rc;
        size_t cap = 0;
        // first attempt: the list length predicted from the previous iteration (no host round trip
        // between the matcher's phases); second attempt: the exact length
        rc = sharded_match(ctx, map, cloud, pose, prm, pairs, attempt == 0 ? c.cap_guess : 0, &cap);
        if (ctx->clear_deferred)
        {   // (not consumed: an error return, or a path without a compaction -- the flag must not outlive this call)
            ctx->clear_deferred = nullptr;
            if (!rc) rc = mp2p_hip_pairs_clear(ctx, pairs);
        }
        if (rc) return rc;
        // ends with the pose read-back; the true longest list of this iteration (all-reduced MAX in exch[6]) rides along
        rc = sharded_solve(ctx, pairs, pose, gn, out, claims);
        if (rc) return rc;
        if (!claims) return MP2P_HIP_OK;
        const double n_max = ((const double*)((const char*)ctx->pinned + COMM_PINNED_EXCH))[6];
        c.cap_guess        = round_cap(n_max * 1.25 + 1024.0);
        if ((double)cap >= n_max) return MP2P_HIP_OK;
        if (redone) *redone = 1;  // rare: the lists did not fit the predicted length
    }
    return set_err(ctx, MP2P_HIP_ERR_HIP, "sharded step: record lists did not fit twice");
}

int mp2p_hip_step_sharded_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                                const double pose[12], const mp2p_hip_pt2pl_params* prm, uint64_t local_index_offset,
                                const mp2p_hip_gn_params* gn, mp2p_hip_pairs* pairs, mp2p_hip_gn_result* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && cloud && pose && prm && gn && pairs && out, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx && cloud->ctx == ctx && pairs->ctx == ctx, "handle belongs to another context");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mp2p_hip_pairs_clear(ctx, pairs);
    if (rc) return rc;
    Comm& c = ctx->comm;
    if (c.nranks <= 1)
    {  // one rank: the plain matcher + solver in one boundary call
        if ((rc = mp2p_hip_match_pt2pl(ctx, map, cloud, pose, prm, nullptr, pairs))) return rc;
        return mp2p_hip_gn_solve(ctx, pairs, pose, gn, out);
    }
    // Matcher_Point2Plane pairs every local point on its own (no global uniqueness, Matcher_Point2Plane.cpp:87-90):
    // the only thing the shards share before the solver is the layer's bounding box (:59-66)
    if (cloud->n && map->n)
    {
        if ((rc = launch_match_pt2pl(ctx, map, cloud, pose, prm, nullptr, pairs, 1, local_index_offset))) return rc;
    }
    else
    {  // an empty shard: a box that loses every MAX
        MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
        const float e[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(ctx->local_bbox.p, e, sizeof(e), hipMemcpyHostToDevice, ctx->stream));
        MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `e` is a stack temporary
    }
    MP2P_TRY_HIP(ctx, ctx->exch.ensure(8));
    hipLaunchKernelGGL(bbox_to_exch_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->local_bbox.p, ctx->exch.p);
    if ((rc = comm_allreduce_f64(ctx, ctx->exch.p, 8, 1))) return rc;
    hipLaunchKernelGGL(exch_to_bbox_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->exch.p, ctx->local_bbox.p);
    if (cloud->n && map->n)
    {
        if ((rc = launch_match_pt2pl(ctx, map, cloud, pose, prm, nullptr, pairs, 2, local_index_offset))) return rc;
    }
    else if ((rc = launch_add_potential(ctx, pairs, (unsigned long long)cloud->n))) return rc;  // :54 before the early-out
    // per inner iteration: accumulate ; all-reduce SUM of the 48 sums ; every rank takes the same step
    return sharded_solve(ctx, pairs, pose, gn, out);
}

}  // extern "C"
