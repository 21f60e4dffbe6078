// api.hip -- extern "C" entry points of libmp2p_hip.so (see include/mp2p_hip.h).
#include <mutex>

#include "device_utils.hpp"

namespace mp2p
{
static std::mutex  g_err_mu;
static std::string g_err;

void set_global_err(const char* fmt, ...)
{
    char    buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = buf;
}

int set_err(mp2p_hip_ctx* ctx, int code, const char* fmt, ...)
{
    char    buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
    {
        std::lock_guard<std::mutex> lk(g_err_mu);
        g_err = buf;
    }
    return code;
}

// MP2P_HIP_TUNE="lane_cells=3,tile_cand_cap=4096,claim_dedup=0": measurement knobs (common.hpp).
// from_env: the string is the environment variable, read when the context is created -- every knob is taken but tile_sol
// (timing-only launches are a run-time request of a profiling session, never a property of the process).
// At run time (mp2p_hip_set_tune) the knobs a context has already consumed are refused.  Returns the number of entries that
// were refused (unknown name, malformed value, not allowed here); `why` gets the first one.
static int parse_tune(Tune& t, const char* e, bool from_env, std::string* why = nullptr)
{
    if (!e) return 0;
    int         bad = 0;
    auto        refuse = [&](const std::string& kv, const char* reason) {
        if (!bad && why) *why = "'" + kv + "': " + reason;
        bad++;
        if (from_env) fprintf(stderr, "[libmp2p_hip] MP2P_HIP_TUNE: '%s' ignored (%s)\n", kv.c_str(), reason);
    };
    std::string s(e);
    size_t      i = 0;
    while (i < s.size())
    {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string kv = s.substr(i, j - i);
        i = j + 1;
        if (kv.empty()) continue;
        const size_t q = kv.find('=');
        if (q == std::string::npos || q == 0 || q + 1 >= kv.size())
        {
            refuse(kv, "not name=value");
            continue;
        }
        const std::string k = kv.substr(0, q);
        char*             endp = nullptr;
        const long        v = strtol(kv.c_str() + q + 1, &endp, 10);
        if (!endp || *endp != '\0')
        {
            refuse(kv, "value is not an integer");
            continue;
        }
        // consumed when the context is created (buffers, events and streams are sized from them)
        const bool creation_only = k == "copy_chunk_kb" || k == "copy_stage_mb" || k == "dir_budget_mb" || k == "spin_us" ||
                                   k == "sync_spin" || k == "pipelines";
        if (creation_only && !from_env)
        {
            refuse(kv, "read when the context is created: set it in MP2P_HIP_TUNE");
            continue;
        }
        if (k == "lane_cells") t.lane_cells = (uint32_t)v;
        else if (k == "tile_cand_cap") t.tile_cand_cap = (uint32_t)v;
        else if (k == "hard_radius_pct") t.hard_radius_pct = (uint32_t)v;
        else if (k == "sync_spin") t.sync_spin = (int)v;
        else if (k == "pl_cert") t.pl_cert = (int)v;
        else if (k == "pl_cert_pad") t.pl_cert_pad = (uint32_t)v;
        else if (k == "pl_cert_step_mm") t.pl_cert_step_mm = (uint32_t)v;
        else if (k == "pl_cert_margin_mm") t.pl_cert_margin_mm = (uint32_t)v;
        else if (k == "pl_hard_cand") t.pl_hard_cand = (uint32_t)v;
        else if (k == "spin_us") t.spin_us = (int)v;
        else if (k == "single_waves") t.single_waves = (uint32_t)v;
        else if (k == "xcd_map") t.xcd_map = (int)v;
        else if (k == "single_blocks_per_cu") t.single_blocks_per_cu = (uint32_t)v;
        else if (k == "pl_q") t.pl_q = (v == 8 || v == 32) ? (uint32_t)v : 0u;
        else if (k == "claim_dedup") t.claim_dedup = (int)v;
        else if (k == "tile_waves") t.tile_waves = (uint32_t)v;
        else if (k == "pipelines") t.pipelines = (uint32_t)v;
        else if (k == "mfma_scan") t.mfma_scan = (int)v;
        else if (k == "dir_budget_mb") t.dir_budget_mb = (uint32_t)v;
        else if (k == "claim_peek") t.claim_peek = (int)v;
        else if (k == "compact_fused") t.compact_fused = (int)v;
        else if (k == "pl_warm") t.pl_warm = (int)v;
        else if (k == "tile_bricks") t.tile_bricks = (int)v;
        else if (k == "tile_brick_budget") t.tile_brick_budget = (uint32_t)v;
        else if (k == "hard_cand") t.hard_cand = (uint32_t)v;
        else if (k == "empty_room") t.empty_room = (int)v;
        else if (k == "far_pass") t.far_pass = (int)v;
        else if (k == "nn_cert") t.nn_cert = (int)v;
        else if (k == "coop_max") t.coop_max = (uint32_t)v;
        else if (k == "nn_cert_step_mm") t.nn_cert_step_mm = (uint32_t)v;
        else if (k == "tile_cand_cap_easy") t.tile_cand_cap_easy = (uint32_t)v;
        else if (k == "copy_chunk_kb") t.copy_chunk_kb = (uint32_t)v;
        else if (k == "copy_stage_mb") t.copy_stage_mb = (uint32_t)v;
        else if (k == "tile_select") t.tile_select = (int)v;
        else if (k == "nn_direct") t.nn_direct = (int)v;
        else if (k == "tile_sol")
        {
            if (from_env) refuse(kv, "timing-only launches are requested through mp2p_hip_set_tune with profiling on");
            else t.tile_sol = (int)v;
        }
        else if (k == "grp_all_bricks") t.grp_all_bricks = (uint32_t)v;
        else if (k == "pl_select") t.pl_select = (int)v;
        else if (k == "pl_waves") t.pl_waves = (uint32_t)v;
        else if (k == "pl_sel_margin_mm") t.pl_sel_margin_mm = (uint32_t)v;
        else if (k == "pl_sel_hard_cand") t.pl_sel_hard_cand = (uint32_t)v;
        else if (k == "pl_sel_hard_large") t.pl_sel_hard_large = (uint32_t)v;
        else if (k == "pl_no_touch") t.pl_no_touch = (int)v;
        else if (k == "pl_sol")
        {
            if (from_env) refuse(kv, "timing-only launches are requested through mp2p_hip_set_tune with profiling on");
            else t.pl_sol = (int)v;
        }
        else refuse(kv, "unknown knob");
    }
    return bad;
}

static int upload3(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z, size_t n,
                   DevBuf<float>& dx, DevBuf<float>& dy, DevBuf<float>& dz)
{
    MP2P_TRY_HIP(ctx, dx.alloc(n ? n : 1));
    MP2P_TRY_HIP(ctx, dy.alloc(n ? n : 1));
    MP2P_TRY_HIP(ctx, dz.alloc(n ? n : 1));
    if (n)
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(dx.p, x, n * 4, hipMemcpyHostToDevice, ctx->stream));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(dy.p, y, n * 4, hipMemcpyHostToDevice, ctx->stream));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(dz.p, z, n * 4, hipMemcpyHostToDevice, ctx->stream));
        MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MP2P_HIP_OK;
}
}  // namespace mp2p

using namespace mp2p;

extern "C" {

int mp2p_hip_abi_version(void) { return MP2P_HIP_ABI_VERSION; }

int mp2p_hip_abi_check(int header_version, size_t sizeof_pt2pt_params, size_t sizeof_pt2pl_params,
                       size_t sizeof_gn_params, size_t sizeof_gn_result, size_t sizeof_stats)
{
    if (header_version != MP2P_HIP_ABI_VERSION)
        return set_err(nullptr, MP2P_HIP_ERR_INVALID, "caller was built against mp2p_hip.h ABI version %d, this library is version %d",
                       header_version, MP2P_HIP_ABI_VERSION);
    if (sizeof_pt2pt_params != sizeof(mp2p_hip_pt2pt_params) || sizeof_pt2pl_params != sizeof(mp2p_hip_pt2pl_params) ||
        sizeof_gn_params != sizeof(mp2p_hip_gn_params) || sizeof_gn_result != sizeof(mp2p_hip_gn_result) ||
        sizeof_stats != sizeof(mp2p_hip_stats))
        return set_err(nullptr, MP2P_HIP_ERR_INVALID,
                       "struct sizes of the caller's mp2p_hip.h (%zu %zu %zu %zu %zu) differ from the library's (%zu %zu %zu %zu %zu)",
                       sizeof_pt2pt_params, sizeof_pt2pl_params, sizeof_gn_params, sizeof_gn_result, sizeof_stats,
                       sizeof(mp2p_hip_pt2pt_params), sizeof(mp2p_hip_pt2pl_params), sizeof(mp2p_hip_gn_params),
                       sizeof(mp2p_hip_gn_result), sizeof(mp2p_hip_stats));
    return MP2P_HIP_OK;
}
unsigned long long mp2p_hip_debug_alloc_count(void) { return mp2p::dev_alloc_counter(); }

int mp2p_hip_device_count(void)
{
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* mp2p_hip_last_error(const mp2p_hip_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

int mp2p_hip_ctx_create(int device_id, void* hip_stream, mp2p_hip_ctx** out)
{
    if (!out) return MP2P_HIP_ERR_INVALID;
    *out = nullptr;
    const int n = mp2p_hip_device_count();
    if (n <= 0)
        return set_err(nullptr, MP2P_HIP_ERR_NO_DEVICE,
                       "no HIP device visible: libmp2p_hip has no CPU fallback");
    if (device_id < 0 || device_id >= n)
        return set_err(nullptr, MP2P_HIP_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess)
        return set_err(nullptr, MP2P_HIP_ERR_HIP, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    auto* ctx   = new mp2p_hip_ctx();
    ctx->device = device_id;
    if (hip_stream)
        ctx->stream = (hipStream_t)hip_stream, ctx->own_stream = false;
    else
    {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
        {
            delete ctx;
            return set_err(nullptr, MP2P_HIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    for (auto& ev : ctx->ev)
    {
        e = hipEventCreate(&ev);
        if (e != hipSuccess)
        {
            delete ctx;
            return set_err(nullptr, MP2P_HIP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    parse_tune(ctx->tune, getenv("MP2P_HIP_TUNE"), /*from_env=*/true);
    *out = ctx;
    return MP2P_HIP_OK;
}

void mp2p_hip_ctx_destroy(mp2p_hip_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // a copy-out the caller never ended: its later rounds still read the device staging area -- finish it before
    // anything is released (an unknown pointer is a HOST pointer to hipMemcpyAsync: a segmentation fault, not an error)
    (void)mp2p_hip_pairs_copy_end(ctx);
    (void)hipStreamSynchronize(ctx->stream);  // (before the staging buffer goes: no DMA into it may be in flight)
    mp2p::stage_destroy(ctx);
    ctx->nn_spos.release(), ctx->nn_d2.release(), ctx->tile_bbox.release();
    ctx->local_bbox.release(), ctx->block_counts.release(), ctx->counters.release(), ctx->compact_flags.release();
    ctx->gn_partials.release(), ctx->gn_sums.release(), ctx->gn_state.release();
    ctx->aos_stage.release(), ctx->pl_slots.release(), ctx->pl_knn.release();
    ctx->work.release(), ctx->work_q.release(), ctx->tile_bbox2.release(), ctx->block_bbox.release(), ctx->exch.release(), ctx->claim_list.release();
    ctx->pend.release(), ctx->pend_q.release(), ctx->q_counters.release(), ctx->nn_rec.release(), ctx->nn_lb2nd.release();
    ctx->pl_kth.release();
    ctx->pl_lb.release(), ctx->pl_cost.release(), ctx->pl_hard.release(), ctx->pl_pend.release(), ctx->pl_pend_cnt.release(), ctx->pl_cert_stat.release();
    for (auto& b : ctx->scratch) b.release();
    for (auto& ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->copy_ev) (void)hipEventDestroy(ctx->copy_ev);
    (void)mp2p_hip_comm_destroy(ctx);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->q1_pairs) mp2p_hip_pairs_free(nullptr, ctx->q1_pairs);
    if (ctx->q1_cloud) mp2p_hip_cloud_free(nullptr, ctx->q1_cloud);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mp2p_hip_sync(mp2p_hip_ctx* ctx)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

void* mp2p_hip_ctx_stream(mp2p_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---- map ------------------------------------------------------------------------------------
int mp2p_hip_map_upload(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z,
                        size_t n, const mp2p_hip_map_params* prm, mp2p_hip_map** out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, n == 0 || (x && y && z), "null coordinate buffer");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<float> dx, dy, dz;
    int           rc = upload3(ctx, x, y, z, n, dx, dy, dz);
    if (!rc) rc = mp2p_hip_map_upload_device(ctx, dx.p, dy.p, dz.p, n, prm, out);
    dx.release(), dy.release(), dz.release();
    return rc;
}

int mp2p_hip_map_upload_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y,
                               const float* d_z, size_t n, const mp2p_hip_map_params* prm,
                               mp2p_hip_map** out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    auto* m  = new mp2p_hip_map();
    int   rc = build_map(ctx, d_x, d_y, d_z, n, prm, m);
    if (rc)
    {
        mp2p_hip_map_free(ctx, m);
        return rc;
    }
    *out = m;
    return MP2P_HIP_OK;
}

void mp2p_hip_map_free(mp2p_hip_ctx* ctx, mp2p_hip_map* map)
{
    if (!map) return;
    if (ctx && ctx->hint_map == map) ctx->hint_map = nullptr;
    if (ctx && ctx->pl_hint_map == map) ctx->pl_hint_map = nullptr;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    map->pts.release(), map->table.release(), map->claims.release();
    delete map;
}

int mp2p_hip_map_get_info(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, mp2p_hip_map_info* out)
{
    if (!ctx || !map || !out) return MP2P_HIP_ERR_INVALID;
    *out = map->info;
    return MP2P_HIP_OK;
}

void*  mp2p_hip_map_claims_ptr(const mp2p_hip_map* map) { return map ? (void*)map->claims.p : nullptr; }
size_t mp2p_hip_map_claims_count(const mp2p_hip_map* map) { return map ? map->n : 0; }

// ---- cloud ----------------------------------------------------------------------------------
int mp2p_hip_cloud_upload(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z,
                          size_t n, mp2p_hip_cloud** out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, n == 0 || (x && y && z), "null coordinate buffer");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<float> dx, dy, dz;
    int           rc = upload3(ctx, x, y, z, n, dx, dy, dz);
    if (!rc) rc = mp2p_hip_cloud_upload_device(ctx, dx.p, dy.p, dz.p, n, out);
    dx.release(), dy.release(), dz.release();
    return rc;
}

int mp2p_hip_cloud_upload_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y,
                                 const float* d_z, size_t n, mp2p_hip_cloud** out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    auto* c  = new mp2p_hip_cloud();
    int   rc = build_cloud(ctx, d_x, d_y, d_z, n, c);
    if (rc)
    {
        mp2p_hip_cloud_free(ctx, c);
        return rc;
    }
    *out = c;
    return MP2P_HIP_OK;
}

void mp2p_hip_cloud_free(mp2p_hip_ctx* ctx, mp2p_hip_cloud* c)
{
    if (!c) return;
    if (ctx && ctx->hint_cloud == c) ctx->hint_cloud = nullptr;
    if (ctx && ctx->pl_hint_cloud == c) ctx->pl_hint_cloud = nullptr;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    c->sorted.release(), c->x.release(), c->y.release(), c->z.release();
    c->order.release(), c->rank.release(), c->pos.release();
    delete c;
}

size_t mp2p_hip_cloud_size(const mp2p_hip_cloud* c) { return c ? c->n : 0; }

int mp2p_hip_cloud_set_visit_order(mp2p_hip_ctx* ctx, mp2p_hip_cloud* c, const uint32_t* order,
                                   size_t n)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, c && c->ctx == ctx, "bad cloud handle");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (!order || n == 0)
    {
        c->n_visit = 0;  // all points, ascending index
        return MP2P_HIP_OK;
    }
    MP2P_REQUIRE(ctx, n <= c->n, "visit order longer than the cloud");
    std::vector<uint32_t> rank(c->n ? c->n : 1, 0xFFFFFFFFu);
    for (size_t r = 0; r < n; r++)
    {
        MP2P_REQUIRE(ctx, order[r] < c->n, "visit order: index out of range");
        MP2P_REQUIRE(ctx, rank[order[r]] == 0xFFFFFFFFu, "visit order: index listed twice");
        rank[order[r]] = (uint32_t)r;
    }
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a previous match may still read the old lists
    MP2P_TRY_HIP(ctx, c->order.ensure(n));
    MP2P_TRY_HIP(ctx, c->rank.ensure(c->n));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(c->order.p, order, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(c->rank.p, rank.data(), c->n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `rank` is a host temporary
    c->n_visit = n;
    return MP2P_HIP_OK;
}

// ---- MatchState -----------------------------------------------------------------------------
int mp2p_hip_mstate_create(mp2p_hip_ctx* ctx, size_t n_global, size_t n_local,
                           mp2p_hip_mstate** out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    auto* ms = new mp2p_hip_mstate();
    ms->ctx  = ctx;
    MP2P_TRY_HIP(ctx, ms->global_taken.alloc(n_global ? n_global : 1));
    MP2P_TRY_HIP(ctx, ms->local_taken.alloc(n_local ? n_local : 1));
    ms->global_taken.n = n_global ? n_global : 1;
    ms->local_taken.n  = n_local ? n_local : 1;
    *out               = ms;
    return mp2p_hip_mstate_reset(ctx, ms);
}

int mp2p_hip_mstate_reset(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms)
{
    if (!ctx || !ms) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipMemsetAsync(ms->global_taken.p, 0, ms->global_taken.n, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(ms->local_taken.p, 0, ms->local_taken.n, ctx->stream));
    return MP2P_HIP_OK;
}

void mp2p_hip_mstate_free(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms)
{
    if (!ms) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    ms->global_taken.release(), ms->local_taken.release();
    delete ms;
}

int mp2p_hip_mstate_download(mp2p_hip_ctx* ctx, const mp2p_hip_mstate* ms, uint8_t* global_taken,
                             uint8_t* local_taken)
{
    if (!ctx || !ms) return MP2P_HIP_ERR_INVALID;
    if (global_taken)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(global_taken, ms->global_taken.p, ms->global_taken.n,
                                         hipMemcpyDeviceToHost, ctx->stream));
    if (local_taken)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(local_taken, ms->local_taken.p, ms->local_taken.n,
                                         hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

int mp2p_hip_mstate_upload(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms, const uint8_t* global_taken,
                           const uint8_t* local_taken)
{
    if (!ctx || !ms) return MP2P_HIP_ERR_INVALID;
    if (global_taken)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(ms->global_taken.p, global_taken, ms->global_taken.n,
                                         hipMemcpyHostToDevice, ctx->stream));
    if (local_taken)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(ms->local_taken.p, local_taken, ms->local_taken.n,
                                         hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

// packed bit-fields <-> one byte per point
__global__ __launch_bounds__(256) void bits_to_bytes_kernel(const unsigned long long* __restrict__ words, size_t n,
                                                            unsigned char* __restrict__ bytes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bytes[i] = (unsigned char)((words[i >> 6] >> (i & 63)) & 1ull);
}
__global__ __launch_bounds__(256) void bytes_to_bits_kernel(const unsigned char* __restrict__ bytes, size_t n,
                                                            unsigned long long* __restrict__ words)
{
    const size_t i    = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one wave = one word
    const bool   set  = i < n && bytes[i] != 0;
    const unsigned long long m = __ballot(set);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) words[i >> 6] = m;
}

int mp2p_hip_mstate_upload_bits(mp2p_hip_ctx* ctx, mp2p_hip_mstate* ms, const uint64_t* global_words,
                                const uint64_t* local_words)
{
    if (!ctx || !ms) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ng = ms->global_taken.n, nl = ms->local_taken.n;
    const size_t wg = (ng + 63) / 64, wl = (nl + 63) / 64;
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure((wg + wl) * 8 + 16));
    auto* dw = reinterpret_cast<unsigned long long*>(ctx->aos_stage.p);
    if (global_words && ng)
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(dw, global_words, wg * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(bits_to_bytes_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, ctx->stream, dw, ng,
                           ms->global_taken.p);
    }
    if (local_words && nl)
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(dw + wg, local_words, wl * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(bits_to_bytes_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, ctx->stream, dw + wg,
                           nl, ms->local_taken.p);
    }
    MP2P_TRY_HIP(ctx, stream_wait(ctx));  // the sources are caller memory
    return MP2P_HIP_OK;
}

int mp2p_hip_mstate_download_bits(mp2p_hip_ctx* ctx, const mp2p_hip_mstate* ms, uint64_t* global_words,
                                  uint64_t* local_words)
{
    if (!ctx || !ms) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ng = ms->global_taken.n, nl = ms->local_taken.n;
    const size_t wg = (ng + 63) / 64, wl = (nl + 63) / 64;
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure((wg + wl) * 8 + 16));
    auto* dw = reinterpret_cast<unsigned long long*>(ctx->aos_stage.p);
    if (global_words)
    {
        hipLaunchKernelGGL(bytes_to_bits_kernel, dim3((unsigned)(wg * 64 / 256 + 1)), dim3(256), 0, ctx->stream,
                           ms->global_taken.p, ng, dw);
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(global_words, dw, wg * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (local_words)
    {
        hipLaunchKernelGGL(bytes_to_bits_kernel, dim3((unsigned)(wl * 64 / 256 + 1)), dim3(256), 0, ctx->stream,
                           ms->local_taken.p, nl, dw + wg);
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(local_words, dw + wg, wl * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    return MP2P_HIP_OK;
}

void* mp2p_hip_host_alloc(mp2p_hip_ctx* ctx, size_t bytes)
{
    if (!ctx || !bytes) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess)
    {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void mp2p_hip_host_free(mp2p_hip_ctx* ctx, void* p)
{
    (void)ctx;
    if (p) (void)hipHostFree(p);
}

// ---- Matcher_Points_DistanceThreshold ----------------------------------------------------------
static int check_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                       const mp2p_hip_pt2pt_params* prm, const mp2p_hip_mstate* ms)
{
    MP2P_REQUIRE(ctx, map && cloud && prm, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx && cloud->ctx == ctx, "handle belongs to another context");
    // ASSERT_(pairingsPerPoint >= 1); ASSERT_GT_(threshold, .0); ASSERT_GE_(thresholdAngularDeg, .0)
    // (Matcher_Points_DistanceThreshold.cpp:57-59)
    MP2P_REQUIRE(ctx, prm->pairingsPerPoint >= 1, "pairingsPerPoint must be >= 1");
    MP2P_REQUIRE(ctx, prm->threshold > 0.0, "threshold must be > 0");
    MP2P_REQUIRE(ctx, prm->thresholdAngularDeg >= 0.0, "thresholdAngularDeg must be >= 0");
    MP2P_REQUIRE(ctx, prm->pairingsPerPoint <= 16, "pairingsPerPoint > 16 is not supported");
    MP2P_REQUIRE(ctx, (prm->local_index_offset + cloud->n) * prm->pairingsPerPoint <= 0xFFFFFFFFull,
                 "whole-layer (local index x pairingsPerPoint) must fit 32 bits");
    if (ms)
    {
        MP2P_REQUIRE(ctx, ms->global_taken.n >= std::max<size_t>(map->n, 1), "MatchState too small (global)");
        MP2P_REQUIRE(ctx, ms->local_taken.n >= std::max<size_t>(cloud->n, 1), "MatchState too small (local)");
    }
    return MP2P_HIP_OK;
}

static int match_pt2pt_phase1(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                              const double pose[12], const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms,
                              bool reduce_bbox)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    int rc = check_pt2pt(ctx, map, cloud, prm, ms);
    if (rc) return rc;
    MP2P_REQUIRE(ctx, pose, "null pose");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (map->n == 0 || cloud->n == 0) return MP2P_HIP_OK;  // :67
    if (prm->pairingsPerPoint > 1) return launch_nn_pt2pt_knn(ctx, map, cloud, pose, prm, ms);  // :242-248
    return launch_nn_pt2pt(ctx, map, cloud, pose, prm, ms, reduce_bbox);
}

static int match_pt2pt_phase2(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                              const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms, mp2p_hip_pairs* out,
                              bool bbox_from_tiles)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    int rc = check_pt2pt(ctx, map, cloud, prm, ms);
    if (rc) return rc;
    MP2P_REQUIRE(ctx, out && out->ctx == ctx, "bad Pairings handle");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->clear_deferred == out && (map->n == 0 || cloud->n == 0 || ctx->sol_no_records || prm->pairingsPerPoint > 1))
    {   // (a clear left to the compaction by mp2p_hip_step_sharded, on a path that does not run the fused compaction)
        ctx->clear_deferred = nullptr;
        MP2P_TRY_HIP(ctx, hipMemsetAsync(out->counts.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    }
    if (map->n == 0 || cloud->n == 0)  // potential_pairings is added BEFORE the early-out (:64-67)
        return launch_add_potential(ctx, out, (unsigned long long)cloud->n * prm->pairingsPerPoint);
    if (ctx->sol_no_records)
    {  // a timing-only search (mp2p_hip_set_tune tile_sol, profiling sessions): the records are an earlier call's or nothing at
       // all -- the list stays as the caller passed it, only potential_pairings grows (:64)
        ctx->sol_no_records = false;
        ctx->q_counters_clean = false;
        rc = launch_add_potential(ctx, out, (unsigned long long)cloud->n * prm->pairingsPerPoint);
    }
    else
        rc = launch_compact_pt2pt(ctx, map, cloud, prm, ms, out, bbox_from_tiles);
    if (!rc && ctx->profiling)
    {
        ctx->pending_match = ctx->profiling == 4 ? 3 : ctx->profiling;  // read back lazily in mp2p_hip_get_stats
        ctx->pending_map_n = map->n;
        ctx->stats.nn_queries = cloud->n;
    }
    return rc;
}

int mp2p_hip_match_pt2pt_phase1(mp2p_hip_ctx* ctx, const mp2p_hip_map* map,
                                const mp2p_hip_cloud* cloud, const double pose[12],
                                const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms)
{
    return match_pt2pt_phase1(ctx, map, cloud, pose, prm, ms, /*reduce_bbox=*/true);
}

int mp2p_hip_match_pt2pt_phase2(mp2p_hip_ctx* ctx, const mp2p_hip_map* map,
                                const mp2p_hip_cloud* cloud, const mp2p_hip_pt2pt_params* prm,
                                mp2p_hip_mstate* ms, mp2p_hip_pairs* out)
{
    return match_pt2pt_phase2(ctx, map, cloud, prm, ms, out, /*bbox_from_tiles=*/false);
}

int mp2p_hip_match_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const double pose[12], const mp2p_hip_pt2pt_params* prm,
                         mp2p_hip_mstate* ms, mp2p_hip_pairs* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    // one GPU: the layer's bounding box is reduced inside the compaction (two launches less)
    const bool fused = ctx->tune.compact_fused && prm && prm->pairingsPerPoint == 1;
    int rc = match_pt2pt_phase1(ctx, map, cloud, pose, prm, ms, !fused);
    if (rc) return rc;
    return match_pt2pt_phase2(ctx, map, cloud, prm, ms, out, fused);
}

int mp2p_hip_exchange_pack(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                           const mp2p_hip_pt2pt_params* prm, void** exch_dev, void** list_dev,
                           size_t* list_len)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && cloud && prm, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx && cloud->ctx == ctx, "handle belongs to another context");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    int rc = launch_exchange_pack(ctx, map, cloud, prm);
    if (rc) return rc;
    if (exch_dev) *exch_dev = ctx->exch.p;
    if (list_dev) *list_dev = ctx->claim_list.p;
    if (list_len) *list_len = (cloud->n_visit ? cloud->n_visit : cloud->n) * prm->pairingsPerPoint;
    return MP2P_HIP_OK;
}

int mp2p_hip_exchange_unpack(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const void* gathered_dev,
                             size_t n_records)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && map->ctx == ctx, "bad map handle");
    MP2P_REQUIRE(ctx, ctx->exch.p, "exchange_unpack without exchange_pack");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return launch_exchange_unpack(ctx, map, (const unsigned long long*)gathered_dev, n_records);
}

void* mp2p_hip_ctx_local_bbox_ptr(mp2p_hip_ctx* ctx)
{
    if (!ctx) return nullptr;
    if (!ctx->local_bbox.p && ctx->local_bbox.ensure(6) != hipSuccess) return nullptr;
    return ctx->local_bbox.p;
}

// ---- Matcher_Points_InlierRatio ---------------------------------------------------------------------
int mp2p_hip_match_inlier_ratio(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                                const double pose[12], const mp2p_hip_inlier_ratio_params* prm,
                                mp2p_hip_mstate* ms, mp2p_hip_pairs* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && cloud && prm && pose && out, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx && cloud->ctx == ctx && out->ctx == ctx, "handle belongs to another context");
    // ASSERT_GT_(inliersRatio, 0.0); ASSERT_LT_(inliersRatio, 1.0);  (:47-48)
    MP2P_REQUIRE(ctx, prm->inliersRatio > 0.0 && prm->inliersRatio < 1.0, "inliersRatio must be in (0,1)");
    if (ms)
    {
        MP2P_REQUIRE(ctx, ms->global_taken.n >= std::max<size_t>(map->n, 1), "MatchState too small (global)");
        MP2P_REQUIRE(ctx, ms->local_taken.n >= std::max<size_t>(cloud->n, 1), "MatchState too small (local)");
    }
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (map->n == 0 || cloud->n == 0)  // :53-56: potential_pairings first, then the early-out
        return launch_add_potential(ctx, out, (unsigned long long)cloud->n);
    return launch_match_inlier_ratio(ctx, map, cloud, pose, prm, ms, out);
}

// ---- Matcher_Adaptive --------------------------------------------------------------------------------
static int adaptive_check(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                          const mp2p_hip_adaptive_params* prm, mp2p_hip_mstate* ms)
{
    MP2P_REQUIRE(ctx, map && cloud && prm, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx && cloud->ctx == ctx, "handle belongs to another context");
    // Matcher_Adaptive.cpp:50-56
    MP2P_REQUIRE(ctx, prm->confidenceInterval > 0.0 && prm->confidenceInterval < 1.0,
                 "confidenceInterval must be in (0,1)");
    MP2P_REQUIRE(ctx, prm->planeSearchPoints >= prm->planeMinimumFoundPoints,
                 "planeSearchPoints must be >= planeMinimumFoundPoints");
    MP2P_REQUIRE(ctx, prm->planeMinimumFoundPoints >= 3, "planeMinimumFoundPoints must be >= 3");
    MP2P_REQUIRE(ctx, prm->planeEigenThreshold > 0.0, "planeEigenThreshold must be > 0");
    const uint32_t nn = prm->enableDetectPlanes ? prm->planeSearchPoints : prm->maxPt2PtCorrespondences;
    MP2P_REQUIRE(ctx, nn >= 1 && nn <= 16, "neighbours per local point must be in [1,16]");
    MP2P_REQUIRE(ctx, prm->maxPt2PtCorrespondences >= 1, "maxPt2PtCorrespondences must be >= 1");
    MP2P_REQUIRE(ctx, prm->absoluteMaxSearchDistance > 0.0, "absoluteMaxSearchDistance must be > 0");
    MP2P_REQUIRE(ctx, cloud->n_visit == 0,
                 "Matcher_Adaptive with maxLocalPointsPerLayer: the reference throws (matchesPerLocal_.at(localIdx))");
    if (ms)
    {
        MP2P_REQUIRE(ctx, ms->global_taken.n >= std::max<size_t>(map->n, 1), "MatchState too small (global)");
        MP2P_REQUIRE(ctx, ms->local_taken.n >= std::max<size_t>(cloud->n, 1), "MatchState too small (local)");
    }
    return MP2P_HIP_OK;
}

int mp2p_hip_adaptive_search(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                             const double pose[12], const mp2p_hip_adaptive_params* prm,
                             mp2p_hip_mstate* ms, mp2p_hip_adaptive_hist* hist)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, pose && hist, "null argument");
    if (const int rc = adaptive_check(ctx, map, cloud, prm, ms)) return rc;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    memset(hist, 0, sizeof(*hist));
    ctx->ad_knn = 0, ctx->ad_cloud = nullptr, ctx->ad_map = nullptr, ctx->ad_apart = false;
    if (map->n == 0 || cloud->n == 0) return MP2P_HIP_OK;  // :72
    return launch_adaptive_search(ctx, map, cloud, pose, prm, ms, hist);
}

// CHistogram::getHistogramNormalized + confidenceIntervalsFromHistogram(xs, ys, lo, hi, 1 - ci)
// (Matcher_Adaptive.cpp:194-198), restated from MRPT's published sources: xs = linspace(min, max,
// nBins); ys = bins * binSizeInv / count; Hc = cumsum(ys) / max(Hc); hi = xs[first Hc > 1 - arg].
double mp2p_hip_adaptive_ci_high(const mp2p_hip_adaptive_hist* hist, double confidenceInterval)
{
    if (!hist || !hist->valid || hist->count == 0) return NAN;
    const int    N  = MP2P_HIP_ADAPTIVE_BINS;
    const double mn = (double)hist->minSqr, mx = (double)hist->maxSqr;
    if (!(mx > mn)) return mx;  // a single value: MRPT's bin width is 0/0; declared = that value
    const double binSizeInv = (double)(N - 1) / (mx - mn);
    const double K          = binSizeInv / (double)hist->count;
    const double step       = (mx - mn) / (double)(N - 1);
    double       Hc[MP2P_HIP_ADAPTIVE_BINS], xs[MP2P_HIP_ADAPTIVE_BINS];
    double       c = mn, run = 0.0, top = 0.0;
    for (int i = 0; i < N; i++)
    {
        xs[i] = c;  // mrpt::math::linspace accumulates: c = first; c += incr
        c += step;
        run += K * (double)hist->bins[i];
        Hc[i] = run;
        top   = std::max(top, run);
    }
    const double inv = 1.0 / top;
    const double arg = 1.0 - confidenceInterval;
    for (int i = 0; i < N; i++)
        if (Hc[i] * inv > 1.0 - arg) return xs[i];
    return NAN;
}

int mp2p_hip_adaptive_select(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                             const mp2p_hip_adaptive_params* prm, double ci_high, mp2p_hip_mstate* ms,
                             mp2p_hip_pairs* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, out && out->ctx == ctx, "pairings handle missing or of another context");
    if (const int rc = adaptive_check(ctx, map, cloud, prm, ms)) return rc;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    // :69 potential_pairings precedes every early-out
    if (const int rc = launch_add_potential(ctx, out, (unsigned long long)cloud->n * prm->maxPt2PtCorrespondences))
        return rc;
    if (map->n == 0 || cloud->n == 0) return MP2P_HIP_OK;
    const uint32_t nn = prm->enableDetectPlanes ? prm->planeSearchPoints : prm->maxPt2PtCorrespondences;
    MP2P_REQUIRE(ctx, ctx->ad_knn == nn && ctx->ad_cloud == cloud && ctx->ad_map == map,
                 "mp2p_hip_adaptive_select without a matching mp2p_hip_adaptive_search");
    if (ctx->ad_apart)
    {  // the search returned before any launch (boxes apart, Matcher_Adaptive.cpp:78-81): no lists exist, nothing is paired
        ctx->ad_apart = false, ctx->ad_knn = 0;
        return MP2P_HIP_OK;
    }
    MP2P_REQUIRE(ctx, ci_high == ci_high, "threshold is NaN");
    const int rc = launch_adaptive_select(ctx, map, cloud, prm, ci_high, ms, out);
    ctx->ad_knn = 0;  // the lists are consumed (select rewrites them)
    return rc;
}

int mp2p_hip_match_adaptive(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                            const double pose[12], const mp2p_hip_adaptive_params* prm,
                            mp2p_hip_mstate* ms, mp2p_hip_pairs* out, double* ci_high_out,
                            mp2p_hip_adaptive_hist* hist_out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, out && out->ctx == ctx, "pairings handle missing or of another context");
    mp2p_hip_adaptive_hist h;
    int rc = mp2p_hip_adaptive_search(ctx, map, cloud, pose, prm, ms, &h);
    if (rc) return rc;
    if (hist_out) *hist_out = h;
    if (ci_high_out) *ci_high_out = NAN;
    if (!h.valid)  // nothing found (or an empty layer): only potential_pairings changes
        return launch_add_potential(ctx, out, (unsigned long long)cloud->n * prm->maxPt2PtCorrespondences);
    const double hi = mp2p_hip_adaptive_ci_high(&h, prm->confidenceInterval);
    if (ci_high_out) *ci_high_out = hi;
    MP2P_REQUIRE(ctx, hi == hi, "Matcher_Adaptive: confidence limit not found in the histogram");
    return mp2p_hip_adaptive_select(ctx, map, cloud, prm, hi, ms, out);
}

// ---- Matcher_Point2Plane -----------------------------------------------------------------------
int mp2p_hip_match_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const double pose[12], const mp2p_hip_pt2pl_params* prm,
                         mp2p_hip_mstate* ms, mp2p_hip_pairs* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && cloud && prm && pose && out, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx && cloud->ctx == ctx && out->ctx == ctx,
                 "handle belongs to another context");
    MP2P_REQUIRE(ctx, prm->distanceThreshold > 0.0, "distanceThreshold must be > 0");
    MP2P_REQUIRE(ctx, prm->searchRadius > 0.0, "searchRadius must be > 0");
    MP2P_REQUIRE(ctx, prm->knn >= 3 && prm->knn <= 16, "knn must be in [3,16]");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (map->n == 0 || cloud->n == 0)  // Matcher_Point2Plane.cpp:54-57
        return launch_add_potential(ctx, out, (unsigned long long)cloud->n);
    const int rc = launch_match_pt2pl(ctx, map, cloud, pose, prm, ms, out);
    if (!rc && ctx->profiling)
    {
        ctx->pending_match = ctx->profiling == 2 ? 1 : (ctx->profiling == 4 ? 3 : ctx->profiling);
        ctx->stats.nn_queries = cloud->n;
        if (ctx->profiling == 2)
        {  // the k-NN kernel's own counters
            unsigned long long c[64];
            MP2P_TRY_HIP(ctx, hipMemcpyAsync(c, ctx->counters.p, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
            MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->stats.nn_tiles = c[0], ctx->stats.nn_passes = c[1], ctx->stats.nn_candidates_tested = c[2];
            ctx->stats.nn_tile_ticks_sum = c[3], ctx->stats.nn_max_passes_one_tile = c[4];
            ctx->stats.nn_max_candidates_one_tile = c[5], ctx->stats.nn_tile_ticks_max = c[6];
            ctx->stats.nn_cells_visited = c[7];
            // round 6 (pt2pl_seltile_kernel): chain iterations of the queue flushes, queued hits, flushes
            ctx->stats.nn_coop_passes = c[9], ctx->stats.nn_single_queries = c[10], ctx->stats.nn_single_passes = c[11];
            ctx->stats.nn_single_cells = c[12], ctx->stats.nn_single_candidates = c[13], ctx->stats.nn_single_ticks_sum = c[14];  // ticks: staging, prefilter, flushes
            ctx->stats.nn_single_ticks_max = c[15], ctx->stats.nn_single_max_passes = c[48];  // ticks: whole passes up to the merge, merges
            for (int i = 0; i < 6; i++) ctx->stats.nn_wave_phase_ticks[i] = c[50 + i];  // the slowest tile's {hits, chain iterations, enqueue iterations, prefilter positives, blocks}; [5] = enqueue iterations of all tiles
            ctx->stats.nn_wave_inserts = c[56], ctx->stats.nn_wave_rounds = c[57];
            for (int i = 0; i < 24; i++) ctx->stats.nn_tile_ticks_hist[i] = c[16 + i];
            ctx->stats.nn_single_max_candidates = c[49];  // the slowest tile: ticks << 40 | passes << 32 | candidates
            std::vector<unsigned char> t(map->n);
            MP2P_TRY_HIP(ctx, hipMemcpy(t.data(), ctx->scratch[15].p, t.size(), hipMemcpyDeviceToHost));
            uint64_t k = 0;
            for (unsigned char b : t) k += b;
            ctx->stats.nn_points_staged = k;  // distinct map points fetched by the search (N_g,touched)
        }
    }
    return rc;
}

// ---- NearestPlaneCapable::nn_search_pt2pl, one query -----------------------------------------------
__global__ void set_point_kernel(float4* sorted, float* x, float* y, float* z, uint32_t* pos, float px, float py,
                                 float pz)
{
    sorted[0] = make_float4(px, py, pz, __uint_as_float(0u));
    x[0] = px, y[0] = py, z[0] = pz, pos[0] = 0u;
}

int mp2p_hip_nn_search_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const float point[3],
                             float max_search_distance, const mp2p_hip_pt2pl_params* prm,
                             mp2p_hip_nearest_plane* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, map && point && prm && out, "null argument");
    MP2P_REQUIRE(ctx, map->ctx == ctx, "handle belongs to another context");
    MP2P_REQUIRE(ctx, max_search_distance > 0.f, "max_search_distance must be > 0");
    MP2P_REQUIRE(ctx, prm->knn >= 3 && prm->knn <= 16, "knn must be in [3,16]");
    memset(out, 0, sizeof(*out));
    if (map->n == 0) return MP2P_HIP_OK;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->q1_cloud)
    {  // a one-point local layer, re-pointed at every query
        auto* c = new mp2p_hip_cloud();
        c->ctx = ctx, c->n = 1;
        MP2P_TRY_HIP(ctx, c->sorted.alloc(1));
        MP2P_TRY_HIP(ctx, c->pos.alloc(1));
        MP2P_TRY_HIP(ctx, c->x.alloc(1));
        MP2P_TRY_HIP(ctx, c->y.alloc(1));
        MP2P_TRY_HIP(ctx, c->z.alloc(1));
        ctx->q1_cloud = c;
        int rc = mp2p_hip_pairs_create(ctx, 1, 1, &ctx->q1_pairs);
        if (rc) return rc;
    }
    mp2p_hip_cloud* c = ctx->q1_cloud;
    hipLaunchKernelGGL(set_point_kernel, dim3(1), dim3(1), 0, ctx->stream, c->sorted.p, c->x.p, c->y.p, c->z.p,
                       c->pos.p, point[0], point[1], point[2]);
    int rc = mp2p_hip_pairs_clear(ctx, ctx->q1_pairs);
    if (rc) return rc;
    mp2p_hip_pt2pl_params q = *prm;
    q.distanceThreshold = max_search_distance;
    if (!(q.searchRadius > 0.0)) q.searchRadius = max_search_distance;
    q.allowMatchAlreadyMatchedPoints = 1;
    const double I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const int    prof  = ctx->profiling;
    ctx->profiling     = 0;
    rc                 = launch_match_pt2pl(ctx, map, c, I, &q, nullptr, ctx->q1_pairs);
    ctx->profiling     = prof;
    if (rc) return rc;
    if (!ctx->pinned) MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 4096, hipHostMallocDefault));
    auto* h = reinterpret_cast<unsigned char*>(ctx->pinned);
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h, ctx->q1_pairs->counts.p, 64, hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h + 64, ctx->q1_pairs->pl_coef.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h + 96, ctx->q1_pairs->pl_cen.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    unsigned long long cnt[8];
    memcpy(cnt, h, 64);
    if (cnt[1] == 0) return MP2P_HIP_OK;
    out->found = 1;
    memcpy(out->plane, h + 64, 32), memcpy(out->centroid, h + 96, 24);
    // TPlane::distance of the (float) query, narrowed like NearestPlaneResult::distance
    out->distance = (float)fabs(out->plane[0] * (double)point[0] + out->plane[1] * (double)point[1] +
                                out->plane[2] * (double)point[2] + out->plane[3]);
    return MP2P_HIP_OK;
}

// ---- Solver_GaussNewton ------------------------------------------------------------------------
int mp2p_hip_gn_begin(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose0[12],
                      const mp2p_hip_gn_params* prm)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return gn_begin(ctx, pairs, pose0, prm, /*lazy_init=*/false);
}
int   mp2p_hip_gn_accumulate(mp2p_hip_ctx* ctx) { return ctx ? gn_accumulate(ctx) : MP2P_HIP_ERR_INVALID; }
void* mp2p_hip_gn_sums_ptr(mp2p_hip_ctx* ctx) { return ctx ? (void*)ctx->gn_sums.p : nullptr; }
int   mp2p_hip_gn_step(mp2p_hip_ctx* ctx) { return ctx ? gn_step(ctx) : MP2P_HIP_ERR_INVALID; }
int   mp2p_hip_gn_end(mp2p_hip_ctx* ctx, mp2p_hip_gn_result* out) { return ctx ? gn_end(ctx, out) : MP2P_HIP_ERR_INVALID; }

int mp2p_hip_gn_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose0[12],
                      const mp2p_hip_gn_params* prm, mp2p_hip_gn_result* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    int rc = gn_begin(ctx, pairs, pose0, prm, /*lazy_init=*/true);
    if (rc) return rc;
    // the whole inner loop is enqueued without a host round trip; iterations after the
    // convergence test (:365) or the cost test (:344) has fired are no-ops on the device
    for (uint32_t it = 0; it < prm->maxInnerLoopIterations; it++)
    {
        rc = gn_iterate_fused(ctx);
        if (rc) return rc;
    }
    rc = gn_end(ctx, out);
    if (rc) return rc;
    return MP2P_HIP_OK;
}

int mp2p_hip_horn_solve_wp(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const mp2p_hip_horn_params* wp,
                           mp2p_hip_horn_result* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, pairs && wp && out, "null argument");
    MP2P_REQUIRE(ctx, pairs->ctx == ctx, "bad Pairings handle");
    MP2P_REQUIRE(ctx, wp->n_weight_blocks == 0 || (wp->weight_block_count && wp->weight_block_w),
                 "weight blocks announced but not given");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return horn_solve(ctx, pairs, wp, out);
}

int mp2p_hip_horn_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, double w_pt2pt,
                        double pose_out[12], int32_t* solved)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, pairs && pose_out && solved, "null argument");
    MP2P_REQUIRE(ctx, w_pt2pt > 0.0, "pair_weights.pt2pt must be > 0");
    mp2p_hip_horn_params wp;
    memset(&wp, 0, sizeof(wp));
    wp.w_pt2pt = w_pt2pt, wp.scale_outlier_threshold = 1.2, wp.robust_kernel_param = 1.0;
    mp2p_hip_horn_result r;
    *solved = 0;
    unsigned long long h_counts[8];
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_counts, pairs->counts.p, sizeof(h_counts), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_counts[0] < 3) return MP2P_HIP_OK;  // optimal_tf_horn.cpp:98 (also the empty list)
    const int rc = mp2p_hip_horn_solve_wp(ctx, pairs, &wp, &r);
    if (rc) return rc;
    memcpy(pose_out, r.pose, sizeof(r.pose));
    *solved = r.solved;
    return MP2P_HIP_OK;
}

int mp2p_hip_horn_outlier_flags(mp2p_hip_ctx* ctx, uint8_t* flags_host, size_t n)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, flags_host || n == 0, "null argument");
    MP2P_REQUIRE(ctx, n <= ctx->horn_n, "more flags requested than the last Horn call had point pairings");
    if (!n) return MP2P_HIP_OK;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(flags_host, ctx->horn_flags.p, n, hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_pt2ln_pl_to_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* in, const double guess[12],
                                     mp2p_hip_pairs* out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, in && out && guess, "null argument");
    MP2P_REQUIRE(ctx, in->ctx == ctx && out->ctx == ctx && in != out, "bad Pairings handles");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return pt2ln_pl_to_pt2pt(ctx, in, guess, out);
}

// ---- covariance -----------------------------------------------------------------------------------
int mp2p_hip_covariance(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose[12],
                        double finDif_xyz, double finDif_angles, double H_out[36], double cov_out[36],
                        int32_t* positive_definite)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, pairs && pairs->ctx == ctx, "bad Pairings handle");
    MP2P_REQUIRE(ctx, pose && cov_out, "null argument");
    MP2P_REQUIRE(ctx, finDif_xyz > 0 && finDif_angles > 0, "finite-difference increments must be > 0");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    double H[36];
    int    pd = 0;
    const int rc = covariance_run(ctx, pairs, pose, finDif_xyz, finDif_angles, H, cov_out, &pd);
    if (rc) return rc;
    if (H_out) memcpy(H_out, H, sizeof(H));
    if (positive_definite) *positive_definite = pd;
    return MP2P_HIP_OK;
}

// ---- FilterDecimateVoxels ------------------------------------------------------------------------
static int check_decimate(mp2p_hip_ctx* ctx, const mp2p_hip_decimate_params* prm)
{
    MP2P_REQUIRE(ctx, prm, "null parameters");
    MP2P_REQUIRE(ctx, prm->voxel_filter_resolution > 0.0f, "voxel_filter_resolution must be > 0");
    MP2P_REQUIRE(ctx, prm->decimate_method >= MP2P_HIP_DECIMATE_FIRST_POINT &&
                          prm->decimate_method <= MP2P_HIP_DECIMATE_VOXEL_AVERAGE,
                 "unknown decimate_method (RandomPoint is not offered)");
    return MP2P_HIP_OK;
}

int mp2p_hip_filter_decimate_voxels_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y,
                                           const float* d_z, size_t n, const mp2p_hip_decimate_params* prm,
                                           float* d_ox, float* d_oy, float* d_oz, uint32_t* d_osrc,
                                           size_t* n_out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    int rc = check_decimate(ctx, prm);
    if (rc) return rc;
    MP2P_REQUIRE(ctx, n_out && (n == 0 || (d_x && d_y && d_z && d_ox && d_oy && d_oz)), "null argument");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    return filter_decimate_device(ctx, d_x, d_y, d_z, n, prm, d_ox, d_oy, d_oz, d_osrc, n_out);
}

int mp2p_hip_filter_decimate_voxels(mp2p_hip_ctx* ctx, const float* x, const float* y, const float* z,
                                    size_t n, const mp2p_hip_decimate_params* prm, float* out_x,
                                    float* out_y, float* out_z, uint32_t* out_src, size_t* n_out)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    int rc = check_decimate(ctx, prm);
    if (rc) return rc;
    MP2P_REQUIRE(ctx, n_out && (n == 0 || (x && y && z && out_x && out_y && out_z)), "null argument");
    *n_out = 0;
    if (n == 0) return MP2P_HIP_OK;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    mp2p::Scratch<float>    in, out;
    mp2p::Scratch<uint32_t> src;
    MP2P_TRY_HIP(ctx, in.take(ctx, 12, 3 * n));
    MP2P_TRY_HIP(ctx, out.take(ctx, 13, 3 * n));
    MP2P_TRY_HIP(ctx, src.take(ctx, 14, n));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(in.p, x, n * 4, hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(in.p + n, y, n * 4, hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(in.p + 2 * n, z, n * 4, hipMemcpyHostToDevice, ctx->stream));
    size_t m = 0;
    rc = filter_decimate_device(ctx, in.p, in.p + n, in.p + 2 * n, n, prm, out.p, out.p + n, out.p + 2 * n, src.p, &m);
    if (rc) return rc;
    if (m)
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_x, out.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_y, out.p + n, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_z, out.p + 2 * n, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (out_src) MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_src, src.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_out = m;
    return MP2P_HIP_OK;
}

// ---- instrumentation ---------------------------------------------------------------------------
int mp2p_hip_set_profiling(mp2p_hip_ctx* ctx, int enable)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    ctx->profiling = enable < 0 ? 0 : (enable > 4 ? 4 : enable);
    return MP2P_HIP_OK;
}

int mp2p_hip_set_tune(mp2p_hip_ctx* ctx, const char* settings)
{
    if (!ctx || !settings) return MP2P_HIP_ERR_INVALID;
    // all or nothing: an A/B probe must be able to tell that a setting did not take effect
    Tune        t = ctx->tune;
    std::string why;
    if (parse_tune(t, settings, /*from_env=*/false, &why) != 0)
        return set_err(ctx, MP2P_HIP_ERR_INVALID, "mp2p_hip_set_tune: %s (nothing applied)", why.c_str());
    if ((t.tile_sol != 0 || t.pl_sol != 0) && ctx->profiling == 0)
        return set_err(ctx, MP2P_HIP_ERR_INVALID, "mp2p_hip_set_tune: tile_sol / pl_sol (timing-only launches) need profiling on");
    ctx->tune = t;
    return MP2P_HIP_OK;
}

int mp2p_hip_get_timeline(mp2p_hip_ctx* ctx, uint64_t* ticks_host, size_t cap_records, size_t* n_tile_records,
                          size_t* n_single_records)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, n_tile_records && n_single_records, "null argument");
    *n_tile_records = ctx->timeline_tiles, *n_single_records = ctx->timeline_singles;
    const size_t n = ctx->timeline_tiles + ctx->timeline_singles;
    if (!ticks_host || !n) return MP2P_HIP_OK;
    MP2P_REQUIRE(ctx, cap_records >= n, "timeline buffer too small");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpy(ticks_host, ctx->timeline.p, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MP2P_HIP_OK;
}

int mp2p_hip_get_stats(mp2p_hip_ctx* ctx, mp2p_hip_stats* out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    if (ctx->pending_match || ctx->pending_gn) MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pending_match)
    {
        float ms_nn = 0, ms_cp = 0;
        MP2P_TRY_HIP(ctx, hipEventElapsedTime(&ms_nn, ctx->ev[0], ctx->ev[1]));
        ctx->stats.ms_nn = ms_nn;
        if (ctx->pending_match != 3)
        {
            float ms_tile = 0, ms_lane = 0;
            MP2P_TRY_HIP(ctx, hipEventElapsedTime(&ms_tile, ctx->ev[0], ctx->ev[6]));
            ctx->stats.ms_nn_single = ms_nn - ms_tile;
            if (ctx->pending_lane)
            {
                MP2P_TRY_HIP(ctx, hipEventElapsedTime(&ms_lane, ctx->ev[0], ctx->ev[7]));
                ms_tile -= ms_lane;
            }
            ctx->stats.ms_nn_tile = ms_tile, ctx->stats.ms_nn_lane = ms_lane;
            MP2P_TRY_HIP(ctx, hipEventElapsedTime(&ms_cp, ctx->ev[2], ctx->ev[3]));
            ctx->stats.ms_compact = ms_cp;
        }
        if (ctx->pending_match == 2)
        {
            unsigned long long c[64];
            MP2P_TRY_HIP(ctx, hipMemcpy(c, ctx->counters.p, sizeof(c), hipMemcpyDeviceToHost));
            ctx->stats.nn_tile_ticks_sum = c[7], ctx->stats.nn_tile_ticks_max = c[8];
            ctx->stats.nn_coop_passes = c[9];  // queries deferred to the one-per-wave kernel
            ctx->stats.nn_single_queries = c[10], ctx->stats.nn_single_passes = c[11];
            ctx->stats.nn_single_cells = c[12], ctx->stats.nn_single_candidates = c[13];
            ctx->stats.nn_single_max_candidates = c[14];
            ctx->stats.nn_single_ticks_sum = c[40], ctx->stats.nn_single_ticks_max = c[15];
            ctx->stats.nn_single_max_passes = c[41], ctx->stats.nn_single_max_cells = c[42];
            ctx->stats.nn_lane_searched = c[44], ctx->stats.nn_lane_candidates = c[45];
            ctx->stats.nn_lane_voxels = c[46], ctx->stats.nn_lane_pending = c[47], ctx->stats.nn_lane_skipped = c[48];
            ctx->stats.nn_sel_voxels_listed = c[49], ctx->stats.nn_sel_voxels_needed = c[50];
            for (int i = 0; i < 24; i++) ctx->stats.nn_tile_ticks_hist[i] = c[16 + i];
            ctx->stats.nn_tiles = c[0], ctx->stats.nn_passes = c[1];
            ctx->stats.nn_cells_visited = c[2], ctx->stats.nn_candidates_tested = c[3];
            ctx->stats.nn_unresolved_after_first_pass = c[4];
            ctx->stats.nn_max_candidates_one_tile = c[5], ctx->stats.nn_max_passes_one_tile = c[6];
            std::vector<unsigned char> t(ctx->pending_map_n);
            MP2P_TRY_HIP(ctx, hipMemcpy(t.data(), ctx->pl_slots.p, t.size(), hipMemcpyDeviceToHost));
            uint64_t k = 0;
            for (unsigned char b : t) k += b;
            ctx->stats.nn_points_staged = k;
        }
        if (ctx->pending_pl && ctx->pl_cert_stat.p)
        {
            unsigned long long c2[64 * 16];
            MP2P_TRY_HIP(ctx, hipMemcpy(c2, ctx->pl_cert_stat.p, sizeof(c2), hipMemcpyDeviceToHost));
            ctx->stats.pl_certified = 0, ctx->stats.pl_searched = 0;
            for (int i = 0; i < 64; i++) ctx->stats.pl_certified += c2[i * 16], ctx->stats.pl_searched += c2[i * 16 + 1];
        }
        ctx->pending_pl    = 0;
        ctx->pending_match = 0;
    }
    if (ctx->pending_gn)
    {
        float ms = 0;
        MP2P_TRY_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
        ctx->stats.ms_gn = ms;
        ctx->pending_gn = 0;
    }
    *out = ctx->stats;
    return MP2P_HIP_OK;
}

}  // extern "C"
