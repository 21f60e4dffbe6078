// pairs.hip -- K4: threshold/claim resolution + ORDER-PRESERVING compaction into the
// device-resident Pairings, and the Pairings container itself.
//
// Sequential meaning reproduced (Matcher_Points_DistanceThreshold.cpp:94-121, 206-266):
//   local points are visited in ascending index; a pair (i,g) is emitted unless g was
//   already paired; emitting marks i and g.  A skipped candidate has no side effect, hence
//   winner(g) = min{ i : NN(i)=g, d2 < thr } -- exactly what the atomicMin claims of
//   nn_query.hip computed -- and the output order is ascending i (stable compaction).
#include <thread>

#include "device_utils.hpp"

namespace mp2p
{
constexpr int CP_THREADS = 256;
constexpr int CP_ITEMS   = 4;
constexpr int CP_TILE    = CP_THREADS * CP_ITEMS;

struct CompactArgs
{
    unsigned char* flags;  // [blocks * CP_THREADS] the CP_ITEMS flags of every thread, left by the count pass for the write pass
    const uint32_t*           nn_spos;  // [n_l][K] in the Morton order of the local layer
    const uint32_t*           pos;      // original local index -> place in that order
    const float*              nn_d2;
    const uint4*              rec;      // K == 1 point-to-point search: packed records instead (nn_query.hip)
    // fused form (point-to-point matcher on one GPU): the bounding box of the transformed local layer
    // is reduced here -- per-wave boxes -> per-block boxes in the count kernel -> the layer's box and
    // the overlap decision (counts[7]) in the scan kernel -- instead of by two launches of its own
    int                       fresh;         // (fused form) the list is cleared first: mp2p_hip_pairs_clear folded in (its memset was a launch of its own at the head of every step)
    const float*              tile_bbox;     // [n_tile_boxes][6] or null (local_bbox is final already)
    uint32_t                  n_tile_boxes;
    float*                    block_bbox;    // [n_blocks][6]
    float*                    local_bbox_out;
    uint32_t*                 q_counters;    // the search's query-list counters, re-zeroed for the next call
    uint32_t                  n_l;      // number of SLOTS = visited local points x K, in visiting order
    uint32_t                  K;        // pairingsPerPoint
    const uint32_t*           order;    // visit position -> original local index (null: identity)
    const uint32_t*           n_slots_dev;  // optional device override of the slot count (<= n_l)
    int                       always_mark;  // leave MatchState marks even when global re-use is allowed
    const unsigned long long* claims;  // null when global re-use is allowed
    unsigned long long        claim_hi, local_offset;
    const float*              local_bbox;  // device [6]
    float                     gbb[6];      // global layer bbox
    float                     margin;      // threshold + epsilon (:73-75)
    const float4*             gpts;
    const float *             lx, *ly, *lz;  // original-order local coordinates
    uint32_t*                 block_counts;
    unsigned long long*       counts;  // pairs->counts
    unsigned long long        cap;
    unsigned long long        potential_add;
    // outputs
    uint32_t *     o_lidx, *o_gidx;
    float *        o_lx, *o_ly, *o_lz, *o_gx, *o_gy, *o_gz, *o_err;
    unsigned char *ms_local, *ms_global;  // MatchState marks (or null)
};

// TBoundingBoxf::intersection(other, eps).has_value()  (Matcher_Points_DistanceThreshold.cpp:
// 73-75; MRPT semantics: no overlap when a box, inflated by eps, is strictly beyond the other)
__device__ __forceinline__ bool bbox_overlap(const float* g, const float* l, float eps)
{
    for (int d = 0; d < 3; d++)
    {
        if (l[d] - eps > g[3 + d]) return false;
        if (l[3 + d] + eps < g[d]) return false;
    }
    return true;
}

// slot t = (visit position r, neighbour k): does it produce a pair?  i = original local index,
// src = its place in nn_spos / nn_d2
__device__ __forceinline__ bool pair_flag(const CompactArgs& a, uint32_t t, uint32_t& spos,
                                          uint32_t& i, size_t& src, float& d2)
{
    const uint32_t r = (a.K == 1) ? t : t / a.K;
    const uint32_t k = (a.K == 1) ? 0u : t - r * a.K;
    i                = a.order ? a.order[r] : r;
    src              = (size_t)a.pos[i] * a.K + k;
    if (a.rec)
    {
        const uint4 q = a.rec[src];
        spos = (q.w & 1u) ? q.x : NONE_U32, d2 = __uint_as_float(q.y);  // bit 1: nn_query.hip's prediction flag
    }
    else
    {
        spos = a.nn_spos[src];
        d2   = spos != NONE_U32 ? a.nn_d2[src] : 0.f;
    }
    if (spos == NONE_U32) return false;
    if (a.claims && a.claims[spos] != (a.claim_hi | ((a.local_offset + r) * a.K + k))) return false;
    return true;
}

__global__ __launch_bounds__(CP_THREADS) void compact_count_kernel(const CompactArgs a)
{
    __shared__ uint32_t s_w[CP_THREADS / 64];
    __shared__ float    s_bb[CP_THREADS / 64][6];
    uint32_t            c = 0, fbits = 0;
    if (a.tile_bbox)
    {
        // this block's slice of the per-wave boxes (NaN-free by construction: nn_query.hip)
        const uint32_t per = (a.n_tile_boxes + gridDim.x - 1) / gridDim.x;
        const uint32_t b0 = blockIdx.x * per, b1 = min(a.n_tile_boxes, b0 + per);
        float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = b0 + threadIdx.x; i < b1; i += CP_THREADS)
        {
            const float* p = a.tile_bbox + (size_t)i * 6;
            for (int d = 0; d < 3; d++) v[d] = fminf(v[d], p[d]), v[3 + d] = fmaxf(v[3 + d], p[3 + d]);
        }
        for (int d = 0; d < 3; d++) v[d] = wave_min(v[d]), v[3 + d] = wave_max(v[3 + d]);
        if ((threadIdx.x & 63) == 0)
            for (int d = 0; d < 6; d++) s_bb[threadIdx.x >> 6][d] = v[d];
    }
    if (a.tile_bbox || bbox_overlap(a.gbb, a.local_bbox, a.margin))
    {
        const uint32_t base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
#pragma unroll
        for (int k = 0; k < CP_ITEMS; k++)
        {
            uint32_t sp, i;
            size_t   src;
            float    d2;
            if (base + k < (a.n_slots_dev ? min(*a.n_slots_dev, a.n_l) : a.n_l) && pair_flag(a, base + k, sp, i, src, d2))
                c++, fbits |= 1u << k;
        }
    }
    // the write pass gathers (record, claim word, global point) only where a pair is due: ~10 % of the slots
    a.flags[(size_t)blockIdx.x * CP_THREADS + threadIdx.x] = (unsigned char)fbits;
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t t = 0;
        for (int w = 0; w < CP_THREADS / 64; w++) t += s_w[w];
        a.block_counts[blockIdx.x] = t;
    }
    if (a.tile_bbox && threadIdx.x < 6)
    {
        const int d = threadIdx.x;
        float     r = s_bb[0][d];
        for (int k = 1; k < CP_THREADS / 64; k++) r = d < 3 ? fminf(r, s_bb[k][d]) : fmaxf(r, s_bb[k][d]);
        a.block_bbox[(size_t)blockIdx.x * 6 + d] = r;
    }
}

// single block: exclusive scan of the block counts (in place), update the Pairings counters
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* block_counts,
                                                            uint32_t n_blocks,
                                                            unsigned long long* counts,
                                                            unsigned long long cap,
                                                            unsigned long long potential_add,
                                                            int which /*0 pt2pt, 1 pt2pl*/)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_run;
    const int           lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024)
    {
        const uint32_t i    = b0 + threadIdx.x;
        const uint32_t v    = i < n_blocks ? block_counts[i] : 0;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; k++) woff += s_w[k];
        const uint32_t run = s_run;
        if (i < n_blocks) block_counts[i] = run + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = run + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const unsigned long long total = s_run;
        const unsigned long long old   = counts[which];
        counts[3]                      = old;  // write base for the scatter kernel
        unsigned long long nw          = old + total;
        if (nw > cap)
        {
            counts[4] = 1;  // overflow: caller-provided capacity too small
            nw        = cap;
        }
        counts[which] = nw;
        counts[2] += potential_add;
    }
}

// fused form of the scan: first the layer's box from the block boxes and the bounding-box early-out
// of Matcher_Points_DistanceThreshold.cpp:73-75 (no overlap = no pairs: the counts are scanned as
// zeros and the write kernel leaves at once), then the scan above; last, the search's query-list
// counters are cleared for the next call (one launch less at its start)
__global__ __launch_bounds__(1024) void compact_scan_bbox_kernel(const CompactArgs a, uint32_t n_blocks)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_run;
    __shared__ float    s_bb[16][6];
    __shared__ int      s_overlap;
    const int           lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = threadIdx.x; i < n_blocks; i += 1024)
        {
            const float* p = a.block_bbox + (size_t)i * 6;
            for (int d = 0; d < 3; d++) v[d] = fminf(v[d], p[d]), v[3 + d] = fmaxf(v[3 + d], p[3 + d]);
        }
        for (int d = 0; d < 3; d++) v[d] = wave_min(v[d]), v[3 + d] = wave_max(v[3 + d]);
        if (lane == 0)
            for (int d = 0; d < 6; d++) s_bb[w][d] = v[d];
    }
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        float l[6];
        for (int d = 0; d < 6; d++)
        {
            float r = s_bb[0][d];
            for (int k = 1; k < 16; k++) r = d < 3 ? fminf(r, s_bb[k][d]) : fmaxf(r, s_bb[k][d]);
            l[d] = r, a.local_bbox_out[d] = r;
        }
        s_overlap = bbox_overlap(a.gbb, l, a.margin) ? 1 : 0;
    }
    __syncthreads();
    const bool overlap = s_overlap != 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024)
    {
        const uint32_t i    = b0 + threadIdx.x;
        const uint32_t v    = (i < n_blocks && overlap) ? a.block_counts[i] : 0;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; k++) woff += s_w[k];
        const uint32_t run = s_run;
        if (i < n_blocks) a.block_counts[i] = run + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = run + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const unsigned long long total = s_run;
        if (a.fresh) a.counts[0] = 0ull, a.counts[1] = 0ull, a.counts[2] = 0ull, a.counts[4] = 0ull, a.counts[5] = 0ull, a.counts[6] = 0ull;  // (the list's deferred clear)
        const unsigned long long old   = a.counts[0];
        a.counts[3]                    = old;  // write base for the scatter kernel
        unsigned long long nw          = old + total;
        if (nw > a.cap)
        {
            a.counts[4] = 1;  // overflow: caller-provided capacity too small
            nw          = a.cap;
        }
        a.counts[0] = nw;
        a.counts[2] += a.potential_add;
        a.counts[7] = overlap ? 1ull : 0ull;
    }
    if (a.q_counters && threadIdx.x < NN_LISTS * NN_MAX_SEG) a.q_counters[(size_t)threadIdx.x * NN_CNT_STRIDE] = 0u;
}

__global__ __launch_bounds__(CP_THREADS) void compact_write_kernel(const CompactArgs a)
{
    __shared__ uint32_t s_w[CP_THREADS / 64];
    if (a.tile_bbox ? (a.counts[7] == 0ull) : !bbox_overlap(a.gbb, a.local_bbox, a.margin)) return;
    const uint32_t base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    uint32_t       sp[CP_ITEMS], li[CP_ITEMS];
    size_t         src[CP_ITEMS];
    float          d2[CP_ITEMS];
    bool           f[CP_ITEMS];
    uint32_t       c = 0;
    const uint32_t fbits = a.flags[(size_t)blockIdx.x * CP_THREADS + threadIdx.x];
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++)
    {
        f[k] = ((fbits >> k) & 1u) && pair_flag(a, base + k, sp[k], li[k], src[k], d2[k]);
        c += f[k] ? 1u : 0u;
    }
    const int      lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; k++) woff += s_w[k];
    unsigned long long dst = a.counts[3] + a.block_counts[blockIdx.x] + woff + incl - c;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; k++)
    {
        if (!f[k]) continue;
        const uint32_t i = li[k];
        if (dst < a.cap)
        {
            const float4   gp = a.gpts[sp[k]];
            const uint32_t gi = __float_as_uint(gp.w);
            a.o_lidx[dst] = (uint32_t)(a.local_offset + i), a.o_gidx[dst] = gi;  // whole-layer index
            a.o_lx[dst] = a.lx[i], a.o_ly[dst] = a.ly[i], a.o_lz[dst] = a.lz[i];  // UNtransformed
            a.o_gx[dst] = gp.x, a.o_gy[dst] = gp.y, a.o_gz[dst] = gp.z;
            a.o_err[dst] = d2[k];
            if (a.claims || a.always_mark)
            {  // marks are only left when global re-use is forbidden (:116-120)
                if (a.ms_local) a.ms_local[i] = 1;
                if (a.ms_global) a.ms_global[gi] = 1;
            }
        }
        dst++;
    }
}

// ordered compaction over `n_slots` slots (slot = visit position x K + k; `order` maps a visit
// position to the original local index)
int launch_compact_slots(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const uint32_t* order, size_t n_slots, const uint32_t* n_slots_dev, uint32_t K,
                         bool use_claims, bool always_mark, unsigned long long local_offset, float margin,
                         unsigned long long potential_add, mp2p_hip_mstate* ms, mp2p_hip_pairs* out,
                         bool mark_global = true, bool from_rec = false, bool bbox_from_tiles = false)
{
    const size_t   n_l      = n_slots;
    const uint32_t n_blocks = (uint32_t)((n_l + CP_TILE - 1) / CP_TILE);
    MP2P_TRY_HIP(ctx, ctx->block_counts.ensure(n_blocks ? n_blocks : 1));
    MP2P_TRY_HIP(ctx, ctx->compact_flags.ensure((size_t)(n_blocks ? n_blocks : 1) * CP_THREADS));
    CompactArgs a;
    memset(&a, 0, sizeof(a));
    a.nn_spos = ctx->nn_spos.p, a.nn_d2 = ctx->nn_d2.p, a.n_l = (uint32_t)n_l;
    a.rec = from_rec ? ctx->nn_rec.p : nullptr;
    a.K = K, a.order = order, a.n_slots_dev = n_slots_dev, a.always_mark = always_mark ? 1 : 0;
    a.pos = cloud->pos.p;
    a.claims       = use_claims ? map->claims.p : nullptr;
    a.claim_hi     = (~(unsigned long long)ctx->epoch) << 32;
    a.local_offset = local_offset;
    a.local_bbox   = ctx->local_bbox.p;
    for (int d = 0; d < 3; d++) a.gbb[d] = map->view.bbmin[d], a.gbb[3 + d] = map->view.bbmax[d];
    a.margin = margin;
    a.gpts   = map->pts.p;
    a.lx = cloud->x.p, a.ly = cloud->y.p, a.lz = cloud->z.p;
    a.block_counts  = ctx->block_counts.p;
    a.flags         = ctx->compact_flags.p;
    a.counts        = out->counts.p;
    a.cap           = out->cap_pt2pt;
    a.potential_add = potential_add;
    a.o_lidx = out->lidx.p, a.o_gidx = out->gidx.p;
    a.o_lx = out->lx.p, a.o_ly = out->ly.p, a.o_lz = out->lz.p;
    a.o_gx = out->gx.p, a.o_gy = out->gy.p, a.o_gz = out->gz.p, a.o_err = out->err.p;
    a.ms_local  = ms ? ms->local_taken.p : nullptr;
    a.ms_global = (ms && mark_global) ? ms->global_taken.p : nullptr;

    const bool fused = bbox_from_tiles && n_blocks > 0 && ctx->last_n_boxes > 0;
    if (ctx->clear_deferred == out)
    {   // mp2p_hip_step_sharded left the list's clear to this call: the fused scan kernel does it, any other form by the memset
        ctx->clear_deferred = nullptr;
        if (fused) a.fresh = 1;
        else MP2P_TRY_HIP(ctx, hipMemsetAsync(out->counts.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    }
    if (fused)
    {
        MP2P_TRY_HIP(ctx, ctx->block_bbox.ensure((size_t)n_blocks * 6));
        a.tile_bbox = ctx->tile_bbox.p, a.n_tile_boxes = ctx->last_n_boxes;
        a.block_bbox = ctx->block_bbox.p, a.local_bbox_out = ctx->local_bbox.p;
        a.q_counters = ctx->q_counters.p;
    }
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (n_blocks)
        hipLaunchKernelGGL(compact_count_kernel, dim3(n_blocks), dim3(CP_THREADS), 0, ctx->stream, a);
    if (fused)
    {
        hipLaunchKernelGGL(compact_scan_bbox_kernel, dim3(1), dim3(1024), 0, ctx->stream, a, n_blocks);
        ctx->q_counters_clean = true;
    }
    else
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                           ctx->block_counts.p, n_blocks, out->counts.p, a.cap, a.potential_add, 0);
    if (n_blocks)
        hipLaunchKernelGGL(compact_write_kernel, dim3(n_blocks), dim3(CP_THREADS), 0, ctx->stream, a);
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int launch_compact_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms,
                         mp2p_hip_pairs* out, bool bbox_from_tiles)
{
    const size_t n_visit = cloud->n_visit ? cloud->n_visit : cloud->n;
    const size_t n_slots = n_visit * prm->pairingsPerPoint;
    // potential_pairings += pcLocal.size() * pairingsPerPoint (:64): the WHOLE layer, also when
    // maxLocalPointsPerLayer visits a subset of it
    return launch_compact_slots(ctx, map, cloud, cloud->n_visit ? cloud->order.p : nullptr, n_slots, nullptr,
                                prm->pairingsPerPoint, !prm->allowMatchAlreadyMatchedGlobalPoints, false,
                                prm->local_index_offset,
                                (float)(prm->threshold + prm->bounding_box_intersection_check_epsilon),
                                (unsigned long long)cloud->n * prm->pairingsPerPoint, ms, out, true,
                                /*from_rec=*/prm->pairingsPerPoint == 1,
                                bbox_from_tiles && prm->pairingsPerPoint == 1);
}

// ---- sharded local layer: what the ranks exchange between phase 1 and phase 2 -----------------
// exch = double[8]: {-min xyz, +max xyz of this rank's transformed local points, number of claim
// records, 0}: ONE all-reduce MAX gives every rank the whole layer's box (negation is exact) and
// the longest record list.  A claim record = (sorted global position << 32 | whole-layer local
// index) of a local point that survived this rank's own unique-global filter: only these can
// win across ranks, and there are at most as many as distinct global points hit (a few per cent
// of the map), so the ranks all-gather records instead of reducing one word per global point.
constexpr int CE_THREADS = 1024;
__global__ __launch_bounds__(CE_THREADS) void claims_export_kernel(const uint32_t* __restrict__ nn_spos,
                                                            const uint4* __restrict__ rec,
                                                            uint32_t n_slots, uint32_t K,
                                                            const uint32_t* order, const uint32_t* pos,
                                                            const unsigned long long* claims,
                                                            unsigned long long claim_hi,
                                                            unsigned long long local_offset,
                                                            unsigned long long* list,
                                                            unsigned long long* counter)
{
    const uint32_t t    = blockIdx.x * blockDim.x + threadIdx.x;
    const int      lane = threadIdx.x & 63;
    uint32_t       spos = NONE_U32;
    unsigned long long id = 0;
    if (t < n_slots)
    {
        const uint32_t r = t / K, k = t - r * K;
        const uint32_t i = order ? order[r] : r;
        if (rec)
        {
            const uint4 q = rec[pos[i]];
            spos          = (q.w & 1u) ? q.x : NONE_U32;
        }
        else
            spos = nn_spos[(size_t)pos[i] * K + k];
        id = (local_offset + r) * K + k;
    }
    const bool mine = spos != NONE_U32 && claims[spos] == (claim_hi | id);
    const unsigned long long m = __ballot(mine);
    // one reservation per BLOCK of 1 024 slots (round 6: one per wave was 15 000 atomics on one address per rank and step of a
    // 1 M-point shard -- serialised at ~12 ns each, a third of a 0.45 ms step; the records' order in the list means nothing:
    // they are imported with atomicMin)
    __shared__ uint32_t           s_cnt[CE_THREADS / 64];
    __shared__ unsigned long long s_base;
    const int w = threadIdx.x >> 6;
    if (lane == 0) s_cnt[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < CE_THREADS / 64; i++)
    {
        if (i < w) off += s_cnt[i];
        tot += s_cnt[i];
    }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(counter, (unsigned long long)tot) : 0ull;
    __syncthreads();
    if (mine)
        list[s_base + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)spos << 32) | (id & 0xFFFFFFFFull);
}

__global__ void exchange_pack_kernel(const float* bbox, int have_bbox, const unsigned long long* counter,
                                     double* exch)
{
    for (int d = 0; d < 3; d++)
    {
        exch[d]     = have_bbox ? -(double)bbox[d] : -(double)INFINITY;
        exch[3 + d] = have_bbox ? (double)bbox[3 + d] : -(double)INFINITY;
    }
    exch[6] = counter ? (double)*counter : 0.0;
    exch[7] = 0.0;
}

__global__ void exchange_unpack_kernel(const double* exch, float* bbox)
{
    for (int d = 0; d < 3; d++) bbox[d] = (float)(-exch[d]), bbox[3 + d] = (float)exch[3 + d];
}

__global__ __launch_bounds__(256) void claims_import_kernel(const unsigned long long* __restrict__ list,
                                                            size_t n, unsigned long long* claims,
                                                            size_t n_g, unsigned long long claim_hi)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long rec = list[i];
    if (rec == ~0ull) return;  // padding
    const unsigned long long spos = rec >> 32;
    if (spos < n_g) atomicMin(&claims[spos], claim_hi | (rec & 0xFFFFFFFFull));
}

int launch_exchange_pack(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                         const mp2p_hip_pt2pt_params* prm)
{
    const uint32_t K      = prm->pairingsPerPoint;
    const size_t n_l      = (cloud->n_visit ? cloud->n_visit : cloud->n) * K;  // slots
    const bool   searched = map->n > 0 && cloud->n > 0;  // phase 1 ran (it returns early otherwise)
    const bool   claims   = !prm->allowMatchAlreadyMatchedGlobalPoints;
    MP2P_TRY_HIP(ctx, ctx->exch.ensure(8));
    MP2P_TRY_HIP(ctx, ctx->claim_list.ensure(n_l + 1));  // [n_l] records + the counter
    unsigned long long* counter = ctx->claim_list.p + n_l;
    if (claims)
    {
        // padding first: the tail of the list beyond this rank's count travels in the all-gather
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->claim_list.p, 0xFF, n_l * sizeof(unsigned long long), ctx->stream));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
        if (searched)
            hipLaunchKernelGGL(claims_export_kernel, dim3((unsigned)((n_l + CE_THREADS - 1) / CE_THREADS)), dim3(CE_THREADS), 0,
                               ctx->stream, ctx->nn_spos.p, K == 1 ? ctx->nn_rec.p : nullptr, (uint32_t)n_l, K,
                               cloud->n_visit ? cloud->order.p : nullptr, cloud->pos.p, map->claims.p,
                               (~(unsigned long long)ctx->epoch) << 32,
                               (unsigned long long)prm->local_index_offset, ctx->claim_list.p, counter);
    }
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    hipLaunchKernelGGL(exchange_pack_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->local_bbox.p,
                       searched ? 1 : 0, claims ? counter : nullptr, ctx->exch.p);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int launch_exchange_unpack(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const unsigned long long* gathered,
                           size_t n_records)
{
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    hipLaunchKernelGGL(exchange_unpack_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->exch.p,
                       ctx->local_bbox.p);
    if (gathered && n_records && map->n)
        hipLaunchKernelGGL(claims_import_kernel, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0,
                           ctx->stream, gathered, n_records, map->claims.p, map->n,
                           (~(unsigned long long)ctx->epoch) << 32);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

// only the potential_pairings bookkeeping (empty map / empty local: :64-67)
__global__ void add_potential_kernel(unsigned long long* counts, unsigned long long add)
{
    counts[2] += add;
}
int launch_add_potential(mp2p_hip_ctx* ctx, mp2p_hip_pairs* out, unsigned long long add)
{
    hipLaunchKernelGGL(add_potential_kernel, dim3(1), dim3(1), 0, ctx->stream, out->counts.p, add);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

// ---- host images <-> SoA -------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_pt2pt_kernel(
    const uint32_t* lidx, const uint32_t* gidx, const float* lx, const float* ly, const float* lz,
    const float* gx, const float* gy, const float* gz, const float* err, uint32_t n,
    mp2p_hip_pair_pt2pt* out, uint32_t first = 0)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = first + k;
    mp2p_hip_pair_pt2pt p;
    p.globalIdx = gidx[i], p.localIdx = lidx[i];
    p.global_xyz[0] = gx[i], p.global_xyz[1] = gy[i], p.global_xyz[2] = gz[i];
    p.local_xyz[0] = lx[i], p.local_xyz[1] = ly[i], p.local_xyz[2] = lz[i];
    p.errorSquareAfterTransformation = err[i];
    out[k]                           = p;
}

// the part of a point pairing the host cannot know: global point + squared error, 16 bytes (mp2p_hip_pairs_copy_pt2pt_begin_soa)
__global__ __launch_bounds__(256) void pack_pt2pt_g16_kernel(const float* gx, const float* gy, const float* gz, const float* err, uint32_t n,
                                                             float4* out, uint32_t first)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = first + k;
    out[k]           = make_float4(gx[i], gy[i], gz[i], err[i]);
}

__global__ __launch_bounds__(256) void unpack_pt2pt_kernel(const mp2p_hip_pair_pt2pt* in,
                                                           uint32_t n, uint32_t* lidx,
                                                           uint32_t* gidx, float* lx, float* ly,
                                                           float* lz, float* gx, float* gy,
                                                           float* gz, float* err)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mp2p_hip_pair_pt2pt p = in[i];
    gidx[i] = p.globalIdx, lidx[i] = p.localIdx;
    gx[i] = p.global_xyz[0], gy[i] = p.global_xyz[1], gz[i] = p.global_xyz[2];
    lx[i] = p.local_xyz[0], ly[i] = p.local_xyz[1], lz[i] = p.local_xyz[2];
    err[i] = p.errorSquareAfterTransformation;
}

__global__ __launch_bounds__(256) void pack_pt2pl_kernel(const double* coef, const double* cen,
                                                         const float* lx, const float* ly,
                                                         const float* lz, uint32_t n,
                                                         mp2p_hip_pair_pt2pl* out, uint32_t first = 0)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = first + j;
    mp2p_hip_pair_pt2pl p;
    for (int k = 0; k < 4; k++) p.plane[k] = coef[(size_t)i * 4 + k];
    for (int k = 0; k < 3; k++) p.centroid[k] = cen[(size_t)i * 3 + k];
    p.pt_local[0] = lx[i], p.pt_local[1] = ly[i], p.pt_local[2] = lz[i];
    p._pad = 0.f;
    out[j] = p;
}

__global__ __launch_bounds__(256) void unpack_pt2pl_kernel(const mp2p_hip_pair_pt2pl* in,
                                                           uint32_t n, double* coef, double* cen,
                                                           float* lx, float* ly, float* lz,
                                                           uint32_t* lidx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mp2p_hip_pair_pt2pl p = in[i];
    for (int k = 0; k < 4; k++) coef[(size_t)i * 4 + k] = p.plane[k];
    for (int k = 0; k < 3; k++) cen[(size_t)i * 3 + k] = p.centroid[k];
    lx[i] = p.pt_local[0], ly[i] = p.pt_local[1], lz[i] = p.pt_local[2];
    lidx[i] = i;
}

template <class T>
static int grow_buf(mp2p_hip_ctx* ctx, DevBuf<T>& b, size_t new_count, size_t keep)
{
    DevBuf<T> nb;
    MP2P_TRY_HIP(ctx, nb.alloc(new_count));
    if (keep) MP2P_TRY_HIP(ctx, hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    b = std::move(nb);
    return MP2P_HIP_OK;
}

}  // namespace mp2p

using namespace mp2p;

// counts through the pinned buffer, waited for by polling (one per matcher call on the host path)
static int read_counts(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, unsigned long long h[8])
{
    if (!ctx->pinned) MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 4096, hipHostMallocDefault));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(ctx->pinned, p->counts.p, 8 * sizeof(unsigned long long),
                                     hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    memcpy(h, ctx->pinned, 8 * sizeof(unsigned long long));
    if (h[4])
        return set_err(ctx, MP2P_HIP_ERR_CAPACITY, "Pairings capacity exceeded (cap_pt2pt=%zu cap_pt2pl=%zu)",
                       p->cap_pt2pt, p->cap_pt2pl);
    return MP2P_HIP_OK;
}

// ---- device -> caller memory (see mp2p::CopyStage, common.hpp) ----------------------------------------------------
// The caller's containers are pageable (std::vector storage) and stay untouched by the runtime: the records are
// DMA'd into the context's own page-locked buffer, chunk by chunk, and copied from there by the host.
namespace mp2p
{
// one claimed chunk: wait for its DMA, copy it into the caller's container
static void stage_chunk(mp2p_hip_ctx* ctx, size_t k)
{
    CopyStage&  s = ctx->cstage;
    hipError_t  e;
    for (unsigned it = 1; (e = hipEventQuery(s.ev[k])) == hipErrorNotReady; ++it)
        if ((it & 1023u) == 0) std::this_thread::yield();
    if (s.soa_n)
    {   // assemble the records of the pairs [i0, i1)
        const size_t i0 = k * s.soa_cp, i1 = std::min(s.soa_n, i0 + s.soa_cp);
        int          bad = 0;
        if (e == hipSuccess)
        {
            const float* G = reinterpret_cast<const float*>(s.host);  // {gx, gy, gz, err} per pair
            auto*        o = reinterpret_cast<mp2p_hip_pair_pt2pt*>(s.out);
            for (size_t i = i0; i < i1; i++)
            {
                mp2p_hip_pair_pt2pt r;
                r.globalIdx = s.soa_gi[i], r.localIdx = s.soa_li[i];
                r.global_xyz[0] = G[4 * i], r.global_xyz[1] = G[4 * i + 1], r.global_xyz[2] = G[4 * i + 2];
                const unsigned long long l = (unsigned long long)r.localIdx - s.soa_base;
                if (l < s.soa_nl) r.local_xyz[0] = s.soa_l[0][l], r.local_xyz[1] = s.soa_l[1][l], r.local_xyz[2] = s.soa_l[2][l];
                else r.local_xyz[0] = r.local_xyz[1] = r.local_xyz[2] = NAN, bad = 1;
                r.errorSquareAfterTransformation = G[4 * i + 3];
                o[i] = r;
            }
        }
        std::lock_guard<std::mutex> lk(s.mu);
        if (e != hipSuccess && !s.err) s.err = (int)e;
        if (bad) s.soa_bad = 1;
        if (++s.done == s.n_chunks) s.cv_done.notify_all();
        return;
    }
    const size_t off = k * s.chunk, len = std::min(s.chunk, s.bytes - off);
    if (e == hipSuccess) memcpy(s.out + off, s.host + off, len);
    std::lock_guard<std::mutex> lk(s.mu);
    if (e != hipSuccess && !s.err) s.err = (int)e;
    if (++s.done == s.n_chunks) s.cv_done.notify_all();
}
static void stage_worker(mp2p_hip_ctx* ctx)
{
    (void)hipSetDevice(ctx->device);
    CopyStage&                   s = ctx->cstage;
    std::unique_lock<std::mutex> lk(s.mu);
    for (;;)
    {
        s.cv_work.wait(lk, [&] { return s.stop || s.next < s.n_chunks; });
        if (s.stop) return;
        const size_t k = s.next++;
        lk.unlock();
        stage_chunk(ctx, k);
        lk.lock();
    }
}
// enqueue one round: the first min(bytes, capacity) bytes of [dev, dev + bytes) -> the staging buffer, an event per chunk
static int stage_round(mp2p_hip_ctx* ctx, const unsigned char* dev, unsigned char* out, size_t bytes)
{
    CopyStage&   s     = ctx->cstage;
    s.chunk     = std::max<size_t>(4096, (size_t)ctx->tune.copy_chunk_kb << 10);
    s.stage_max = std::max<size_t>(s.chunk, (size_t)ctx->tune.copy_stage_mb << 20);
    const size_t round = std::min(bytes, s.stage_max);
    if (s.cap < round)
    {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr, s.cap = 0;
        const size_t want = std::min(s.stage_max, std::max(round + round / 4, (size_t)(8u << 20)));
        MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&s.host, want, hipHostMallocDefault));
        s.cap = want;
    }
    const size_t nc = (round + s.chunk - 1) / s.chunk;
    while (s.ev.size() < nc)
    {
        hipEvent_t e = nullptr;
        MP2P_TRY_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s.ev.push_back(e);
    }
    for (size_t k = 0; k < nc; k++)
    {
        const size_t off = k * s.chunk, len = std::min(s.chunk, round - off);
        hipError_t   e   = hipMemcpyAsync(s.host + off, dev + off, len, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(s.ev[k], ctx->stream);
        if (e != hipSuccess)
        {
            // some chunks are already on the link, into s.host: nobody may grow / free that buffer while they fly
            (void)hipStreamSynchronize(ctx->stream);
            MP2P_TRY_HIP(ctx, e);
        }
    }
    if (nc > 1 && !s.th.joinable() && !s.no_helper)
    {
        // (no exception may cross the C boundary: without a helper the calling thread copies every chunk itself)
        try
        {
            s.th = std::thread(stage_worker, ctx);
        }
        catch (...)
        {
            s.no_helper = true;
        }
    }
    {
        std::lock_guard<std::mutex> lk(s.mu);
        s.out = out, s.bytes = round, s.n_chunks = nc, s.next = 0, s.done = 0, s.soa_n = 0;
        s.dev_rest = dev + round, s.out_rest = out + round, s.rest = bytes - round;
    }
    if (nc > 1) s.cv_work.notify_one();
    return MP2P_HIP_OK;
}
// the SoA form of a round (CopyStage::soa_*): n pairs from `first` on, all in ONE round (the caller checked that 16 n bytes fit)
static int stage_post_soa(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n, mp2p_hip_pair_pt2pt* out,
                          const uint32_t* li, const uint32_t* gi, const float* lx, const float* ly, const float* lz, size_t n_local,
                          unsigned long long base)
{
    CopyStage& s = ctx->cstage;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        s.err = 0, s.soa_bad = 0;
    }
    s.chunk     = std::max<size_t>(4096, (size_t)ctx->tune.copy_chunk_kb << 10);
    s.stage_max = std::max<size_t>(s.chunk, (size_t)ctx->tune.copy_stage_mb << 20);
    const size_t bytes = 16 * n;
    if (s.cap < bytes)
    {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr, s.cap = 0;
        const size_t want = std::min(s.stage_max, std::max(bytes + bytes / 4, (size_t)(8u << 20)));
        MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&s.host, want, hipHostMallocDefault));
        s.cap = want;
    }
    const size_t cp = std::max<size_t>(1024, s.chunk / 16), nc = (n + cp - 1) / cp;
    while (s.ev.size() < nc)
    {
        hipEvent_t e = nullptr;
        MP2P_TRY_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s.ev.push_back(e);
    }
    // (one DMA command per chunk: four -- one per device array -- cost more in command overhead than the bytes they saved)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(bytes));
    float4* const D = reinterpret_cast<float4*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pt_g16_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, p->gx.p, p->gy.p, p->gz.p, p->err.p,
                       (uint32_t)n, D, (uint32_t)first);
    for (size_t k = 0; k < nc; k++)
    {
        const size_t i0 = k * cp, len = std::min(cp, n - i0);
        hipError_t   e  = hipMemcpyAsync(s.host + 16 * i0, D + i0, 16 * len, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(s.ev[k], ctx->stream);
        if (e != hipSuccess)
        {
            (void)hipStreamSynchronize(ctx->stream);  // (chunks already on the link land in s.host: nobody may free it meanwhile)
            MP2P_TRY_HIP(ctx, e);
        }
    }
    if (nc > 1 && !s.th.joinable() && !s.no_helper)
    {
        try
        {
            s.th = std::thread(stage_worker, ctx);
        }
        catch (...)
        {
            s.no_helper = true;
        }
    }
    {
        std::lock_guard<std::mutex> lk(s.mu);
        s.out = reinterpret_cast<unsigned char*>(out), s.bytes = bytes, s.n_chunks = nc, s.next = 0, s.done = 0;
        s.dev_rest = nullptr, s.out_rest = nullptr, s.rest = 0;
        s.soa_n = n, s.soa_cp = cp, s.soa_li = li, s.soa_gi = gi, s.soa_l[0] = lx, s.soa_l[1] = ly, s.soa_l[2] = lz;
        s.soa_nl = n_local, s.soa_base = base;
    }
    if (nc > 1) s.cv_work.notify_one();
    s.open = true;
    return MP2P_HIP_OK;
}
// start a copy; the caller may do other work before stage_finish()
static int stage_post(mp2p_hip_ctx* ctx, const void* dev, void* out, size_t bytes)
{
    CopyStage& s = ctx->cstage;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        s.err = 0;
    }
    const int rc = stage_round(ctx, static_cast<const unsigned char*>(dev), static_cast<unsigned char*>(out), bytes);
    s.open = rc == MP2P_HIP_OK;
    return rc;
}
// the calling thread joins the copy; returns when every byte has reached the caller's container
static int stage_finish(mp2p_hip_ctx* ctx)
{
    CopyStage& s = ctx->cstage;
    if (!s.open) return MP2P_HIP_OK;
    s.open = false;
    for (;;)
    {
        std::unique_lock<std::mutex> lk(s.mu);
        while (s.next < s.n_chunks)
        {
            const size_t k = s.next++;
            lk.unlock();
            stage_chunk(ctx, k);
            lk.lock();
        }
        s.cv_done.wait(lk, [&] { return s.done == s.n_chunks; });
        const int                  err = s.err;
        const unsigned char* const dev = s.dev_rest;
        unsigned char* const       out = s.out_rest;
        const size_t               rest = s.rest;
        const int                  bad  = s.soa_n ? s.soa_bad : 0;
        s.n_chunks = s.next = s.done = 0, s.rest = 0, s.soa_n = 0, s.soa_bad = 0;
        lk.unlock();
        if (err) MP2P_TRY_HIP(ctx, (hipError_t)err);
        if (bad)
            return set_err(ctx, MP2P_HIP_ERR_INVALID, "copy_pt2pt_begin_soa: a pairing's localIdx lies outside the local arrays the caller passed "
                                                     "(local_index_base / n_local do not describe the layer the pairs were matched on)");
        if (!rest) return MP2P_HIP_OK;
        const int rc = stage_round(ctx, dev, out, rest);  // a list beyond the staging buffer's bound: the next round
        if (rc) return rc;
    }
}
void stage_destroy(mp2p_hip_ctx* ctx)
{
    CopyStage& s = ctx->cstage;
    if (s.th.joinable())
    {
        {
            std::lock_guard<std::mutex> lk(s.mu);
            s.stop = true;
        }
        s.cv_work.notify_all();
        s.th.join();
    }
    for (hipEvent_t e : s.ev) (void)hipEventDestroy(e);
    s.ev.clear();
    if (s.host) (void)hipHostFree(s.host);
    s.host = nullptr, s.cap = 0;
}
}  // namespace mp2p

// Small lists take the runtime's own pageable path (its bounce buffers); the large ones are staged.  Round 3's
// hipHostRegister of the destination is gone (mp2p::CopyStage).
static int copy_out(mp2p_hip_ctx* ctx, const void* dev, void* out, size_t bytes)
{
    if (bytes < (256u << 10))
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        MP2P_TRY_HIP(ctx, stream_wait(ctx));
        return MP2P_HIP_OK;
    }
    if (const int rc = stage_post(ctx, dev, out, bytes)) return rc;
    return stage_finish(ctx);
}

extern "C" {

int mp2p_hip_pairs_create(mp2p_hip_ctx* ctx, size_t cap_pt2pt, size_t cap_pt2pl,
                          mp2p_hip_pairs** out)
{
    if (!ctx || !out) return MP2P_HIP_ERR_INVALID;
    auto* p = new mp2p_hip_pairs();
    p->ctx  = ctx;
    p->cap_pt2pt = cap_pt2pt, p->cap_pt2pl = cap_pt2pl;
    const size_t c1 = cap_pt2pt ? cap_pt2pt : 1, c2 = cap_pt2pl ? cap_pt2pl : 1;
    MP2P_TRY_HIP(ctx, p->lidx.alloc(c1));
    MP2P_TRY_HIP(ctx, p->gidx.alloc(c1));
    MP2P_TRY_HIP(ctx, p->lx.alloc(c1));
    MP2P_TRY_HIP(ctx, p->ly.alloc(c1));
    MP2P_TRY_HIP(ctx, p->lz.alloc(c1));
    MP2P_TRY_HIP(ctx, p->gx.alloc(c1));
    MP2P_TRY_HIP(ctx, p->gy.alloc(c1));
    MP2P_TRY_HIP(ctx, p->gz.alloc(c1));
    MP2P_TRY_HIP(ctx, p->err.alloc(c1));
    MP2P_TRY_HIP(ctx, p->pl_lidx.alloc(c2));
    MP2P_TRY_HIP(ctx, p->pl_coef.alloc(c2 * 4));
    MP2P_TRY_HIP(ctx, p->pl_cen.alloc(c2 * 3));
    MP2P_TRY_HIP(ctx, p->pl_lx.alloc(c2));
    MP2P_TRY_HIP(ctx, p->pl_ly.alloc(c2));
    MP2P_TRY_HIP(ctx, p->pl_lz.alloc(c2));
    MP2P_TRY_HIP(ctx, p->counts.alloc(8));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(p->counts.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    *out = p;
    return MP2P_HIP_OK;
}

void mp2p_hip_pairs_free(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p)
{
    if (!p) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    p->lidx.release(), p->gidx.release();
    p->lx.release(), p->ly.release(), p->lz.release();
    p->gx.release(), p->gy.release(), p->gz.release(), p->err.release();
    p->pl_lidx.release(), p->pl_coef.release(), p->pl_cen.release();
    p->pl_lx.release(), p->pl_ly.release(), p->pl_lz.release();
    p->ln.release(), p->pp.release();
    p->counts.release();
    delete p;
}

int mp2p_hip_pairs_reserve(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p, size_t cap_pt2pt, size_t cap_pt2pl)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    if (cap_pt2pt <= p->cap_pt2pt && cap_pt2pl <= p->cap_pt2pl) return MP2P_HIP_OK;
    uint64_t n1 = 0, n2 = 0;
    int      rc = mp2p_hip_pairs_counts(ctx, p, &n1, &n2, nullptr);
    if (rc) return rc;
#define MP2P_GROW(buf, cnt, keep)                         \
    do                                                    \
    {                                                     \
        rc = grow_buf(ctx, p->buf, (cnt), (keep));        \
        if (rc) return rc;                                \
    } while (0)
    if (cap_pt2pt > p->cap_pt2pt)
    {
        MP2P_GROW(lidx, cap_pt2pt, n1);
        MP2P_GROW(gidx, cap_pt2pt, n1);
        MP2P_GROW(lx, cap_pt2pt, n1);
        MP2P_GROW(ly, cap_pt2pt, n1);
        MP2P_GROW(lz, cap_pt2pt, n1);
        MP2P_GROW(gx, cap_pt2pt, n1);
        MP2P_GROW(gy, cap_pt2pt, n1);
        MP2P_GROW(gz, cap_pt2pt, n1);
        MP2P_GROW(err, cap_pt2pt, n1);
        p->cap_pt2pt = cap_pt2pt;
    }
    if (cap_pt2pl > p->cap_pt2pl)
    {
        MP2P_GROW(pl_lidx, cap_pt2pl, n2);
        MP2P_GROW(pl_coef, cap_pt2pl * 4, n2 * 4);
        MP2P_GROW(pl_cen, cap_pt2pl * 3, n2 * 3);
        MP2P_GROW(pl_lx, cap_pt2pl, n2);
        MP2P_GROW(pl_ly, cap_pt2pl, n2);
        MP2P_GROW(pl_lz, cap_pt2pl, n2);
        p->cap_pt2pl = cap_pt2pl;
    }
#undef MP2P_GROW
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_clear(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipMemsetAsync(p->counts.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_counts(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, uint64_t* n_pt2pt,
                          uint64_t* n_pt2pl, uint64_t* potential)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    unsigned long long h[8];
    if (const int rc = read_counts(ctx, p, h)) return rc;
    if (n_pt2pt) *n_pt2pt = h[0];
    if (n_pt2pl) *n_pt2pl = h[1];
    if (potential) *potential = h[2];
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_download_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p,
                                  mp2p_hip_pair_pt2pt* out, size_t capacity, size_t* n_out)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    uint64_t n = 0;
    int      rc = mp2p_hip_pairs_counts(ctx, p, &n, nullptr, nullptr);
    if (rc) return rc;
    if (n_out) *n_out = (size_t)n;
    if (n == 0) return MP2P_HIP_OK;
    if (!out || capacity < n)
        return set_err(ctx, MP2P_HIP_ERR_CAPACITY, "download_pt2pt: capacity %zu < %llu", capacity,
                       (unsigned long long)n);
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(n * sizeof(mp2p_hip_pair_pt2pt)));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pt*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pt_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, p->lidx.p, p->gidx.p, p->lx.p, p->ly.p, p->lz.p, p->gx.p,
                       p->gy.p, p->gz.p, p->err.p, (uint32_t)n, d);
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(out, d, n * sizeof(mp2p_hip_pair_pt2pt), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_download_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p,
                                  mp2p_hip_pair_pt2pl* out, uint32_t* out_local_idx,
                                  size_t capacity, size_t* n_out)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    uint64_t n = 0;
    int      rc = mp2p_hip_pairs_counts(ctx, p, nullptr, &n, nullptr);
    if (rc) return rc;
    if (n_out) *n_out = (size_t)n;
    if (n == 0) return MP2P_HIP_OK;
    if (!out || capacity < n)
        return set_err(ctx, MP2P_HIP_ERR_CAPACITY, "download_pt2pl: capacity %zu < %llu", capacity,
                       (unsigned long long)n);
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(n * sizeof(mp2p_hip_pair_pt2pl)));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pl*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pl_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, p->pl_coef.p, p->pl_cen.p, p->pl_lx.p, p->pl_ly.p, p->pl_lz.p,
                       (uint32_t)n, d);
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(out, d, n * sizeof(mp2p_hip_pair_pt2pl), hipMemcpyDeviceToHost, ctx->stream));
    if (out_local_idx)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_local_idx, p->pl_lidx.p, n * sizeof(uint32_t),
                                         hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_copy_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                              mp2p_hip_pair_pt2pt* out)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    if (n == 0) return MP2P_HIP_OK;
    MP2P_REQUIRE(ctx, out && first + n <= p->cap_pt2pt, "copy_pt2pt: range outside the list");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(n * sizeof(mp2p_hip_pair_pt2pt)));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pt*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pt_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, p->lidx.p,
                       p->gidx.p, p->lx.p, p->ly.p, p->lz.p, p->gx.p, p->gy.p, p->gz.p, p->err.p, (uint32_t)n, d,
                       (uint32_t)first);
    return copy_out(ctx, d, out, n * sizeof(mp2p_hip_pair_pt2pt));
}

// the same copy in three steps, so that the caller's pass over the indices (the MatchState marks) runs
// while the records are still on the link: index arrays first (small), an event, then the records
int mp2p_hip_pairs_copy_pt2pt_begin(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                                    mp2p_hip_pair_pt2pt* out, uint32_t* idx_local, uint32_t* idx_global)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    // a copy the caller never ended (an exception between begin and end on its side): finish it here -- the stream is
    // drained and the stale host range unpinned -- rather than refusing every later copy
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);
    MP2P_REQUIRE(ctx, n > 0 && out && idx_local && idx_global && first + n <= p->cap_pt2pt,
                 "copy_pt2pt_begin: empty range, null destination or range outside the list");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_ev) MP2P_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(idx_local, p->lidx.p + first, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(idx_global, p->gidx.p + first, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipEventRecord(ctx->copy_ev, ctx->stream));
    const size_t bytes = n * sizeof(mp2p_hip_pair_pt2pt);
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(bytes));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pt*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pt_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, p->lidx.p,
                       p->gidx.p, p->lx.p, p->ly.p, p->lz.p, p->gx.p, p->gy.p, p->gz.p, p->err.p, (uint32_t)n, d,
                       (uint32_t)first);
    ctx->copy_open = true;
    int rc = MP2P_HIP_OK;
    if (bytes < (256u << 10))
    {
        const hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = set_err(ctx, MP2P_HIP_ERR_HIP, "copy_pt2pt_begin: hipMemcpyAsync failed: %s", hipGetErrorString(e));
    }
    else
        rc = stage_post(ctx, d, out, bytes);
    if (rc)
    {
        (void)mp2p_hip_pairs_copy_end(ctx);
        return rc;
    }
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_copy_pt2pt_begin_soa(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                                        mp2p_hip_pair_pt2pt* out, uint32_t* idx_local, uint32_t* idx_global,
                                        const float* local_x, const float* local_y, const float* local_z, size_t n_local,
                                        uint64_t local_index_base)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, local_x && local_y && local_z, "copy_pt2pt_begin_soa: null local arrays");
    // small lists and lists beyond one round of the staging buffer: the record form (same result, 44 bytes per pair on the link)
    const size_t stage_max = std::max<size_t>((size_t)ctx->tune.copy_chunk_kb << 10, (size_t)ctx->tune.copy_stage_mb << 20);
    if (n * sizeof(mp2p_hip_pair_pt2pt) < (256u << 10) || 16 * n > stage_max)
        return mp2p_hip_pairs_copy_pt2pt_begin(ctx, p, first, n, out, idx_local, idx_global);
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);
    MP2P_REQUIRE(ctx, n > 0 && out && idx_local && idx_global && first + n <= p->cap_pt2pt,
                 "copy_pt2pt_begin_soa: empty range, null destination or range outside the list");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_ev) MP2P_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(idx_local, p->lidx.p + first, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(idx_global, p->gidx.p + first, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipEventRecord(ctx->copy_ev, ctx->stream));
    ctx->copy_open = true;
    const int rc = stage_post_soa(ctx, p, first, n, out, idx_local, idx_global, local_x, local_y, local_z, n_local, local_index_base);
    if (rc)
    {
        (void)mp2p_hip_pairs_copy_end(ctx);
        return rc;
    }
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_copy_wait_idx(mp2p_hip_ctx* ctx)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, ctx->copy_open, "copy_wait_idx without copy_pt2pt_begin");
    hipError_t e;
    for (unsigned it = 1; (e = hipEventQuery(ctx->copy_ev)) == hipErrorNotReady; ++it)
        if ((it & 4095u) == 0) std::this_thread::yield();
    MP2P_TRY_HIP(ctx, e);
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_copy_end(mp2p_hip_ctx* ctx)
{
    if (!ctx) return MP2P_HIP_ERR_INVALID;
    if (!ctx->copy_open) return MP2P_HIP_OK;
    const int        rc = stage_finish(ctx);  // (nothing posted: returns at once)
    const hipError_t e  = stream_wait(ctx);
    ctx->copy_open = false;
    if (rc) return rc;
    MP2P_TRY_HIP(ctx, e);
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_copy_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first, size_t n,
                              mp2p_hip_pair_pt2pl* out, uint32_t* out_local_idx)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    if (n == 0) return MP2P_HIP_OK;
    MP2P_REQUIRE(ctx, out && first + n <= p->cap_pt2pl, "copy_pt2pl: range outside the list");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(n * sizeof(mp2p_hip_pair_pt2pl)));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pl*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pl_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, p->pl_coef.p,
                       p->pl_cen.p, p->pl_lx.p, p->pl_ly.p, p->pl_lz.p, (uint32_t)n, d, (uint32_t)first);
    if (out_local_idx)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_local_idx, p->pl_lidx.p + first, n * sizeof(uint32_t),
                                         hipMemcpyDeviceToHost, ctx->stream));
    return copy_out(ctx, d, out, n * sizeof(mp2p_hip_pair_pt2pl));
}

int mp2p_hip_pairs_download_pt2pt_from(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first,
                                       mp2p_hip_pair_pt2pt* out, size_t capacity, size_t* n_out,
                                       uint64_t* potential)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long h[8];
    if (const int rc = read_counts(ctx, p, h)) return rc;
    if (potential) *potential = h[2];
    MP2P_REQUIRE(ctx, first <= h[0], "download_pt2pt_from: first is beyond the list");
    const size_t n = (size_t)h[0] - first;
    if (n_out) *n_out = n;
    if (n == 0) return MP2P_HIP_OK;
    if (!out || capacity < n)
        return set_err(ctx, MP2P_HIP_ERR_CAPACITY, "download_pt2pt_from: capacity %zu < %zu", capacity, n);
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(n * sizeof(mp2p_hip_pair_pt2pt)));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pt*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pt_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, p->lidx.p,
                       p->gidx.p, p->lx.p, p->ly.p, p->lz.p, p->gx.p, p->gy.p, p->gz.p, p->err.p, (uint32_t)n, d,
                       (uint32_t)first);
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(out, d, n * sizeof(mp2p_hip_pair_pt2pt), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_download_pt2pl_from(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, size_t first,
                                       mp2p_hip_pair_pt2pl* out, uint32_t* out_local_idx, size_t capacity,
                                       size_t* n_out, uint64_t* potential)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long h[8];
    if (const int rc = read_counts(ctx, p, h)) return rc;
    if (potential) *potential = h[2];
    MP2P_REQUIRE(ctx, first <= h[1], "download_pt2pl_from: first is beyond the list");
    const size_t n = (size_t)h[1] - first;
    if (n_out) *n_out = n;
    if (n == 0) return MP2P_HIP_OK;
    if (!out || capacity < n)
        return set_err(ctx, MP2P_HIP_ERR_CAPACITY, "download_pt2pl_from: capacity %zu < %zu", capacity, n);
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(n * sizeof(mp2p_hip_pair_pt2pl)));
    auto* d = reinterpret_cast<mp2p_hip_pair_pt2pl*>(ctx->aos_stage.p);
    hipLaunchKernelGGL(pack_pt2pl_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, p->pl_coef.p,
                       p->pl_cen.p, p->pl_lx.p, p->pl_ly.p, p->pl_lz.p, (uint32_t)n, d, (uint32_t)first);
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(out, d, n * sizeof(mp2p_hip_pair_pt2pl), hipMemcpyDeviceToHost, ctx->stream));
    if (out_local_idx)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(out_local_idx, p->pl_lidx.p + first, n * sizeof(uint32_t),
                                         hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_upload(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p, const mp2p_hip_pair_pt2pt* pt2pt,
                          size_t n_pt2pt, const mp2p_hip_pair_pt2pl* pt2pl, size_t n_pt2pl)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    if (n_pt2pt > p->cap_pt2pt || n_pt2pl > p->cap_pt2pl)
        return set_err(ctx, MP2P_HIP_ERR_CAPACITY, "pairs_upload: capacity exceeded");
    const size_t bytes = std::max(n_pt2pt * sizeof(mp2p_hip_pair_pt2pt),
                                  n_pt2pl * sizeof(mp2p_hip_pair_pt2pl));
    if (ctx->copy_open) (void)mp2p_hip_pairs_copy_end(ctx);  // (its later rounds read the staging area)
    MP2P_TRY_HIP(ctx, ctx->aos_stage.ensure(bytes ? bytes : 1));
    if (n_pt2pt)
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(ctx->aos_stage.p, pt2pt, n_pt2pt * sizeof(mp2p_hip_pair_pt2pt),
                                         hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(unpack_pt2pt_kernel, dim3((uint32_t)((n_pt2pt + 255) / 256)), dim3(256),
                           0, ctx->stream,
                           reinterpret_cast<const mp2p_hip_pair_pt2pt*>(ctx->aos_stage.p),
                           (uint32_t)n_pt2pt, p->lidx.p, p->gidx.p, p->lx.p, p->ly.p, p->lz.p,
                           p->gx.p, p->gy.p, p->gz.p, p->err.p);
        MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging buffer is reused below
    }
    if (n_pt2pl)
    {
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(ctx->aos_stage.p, pt2pl, n_pt2pl * sizeof(mp2p_hip_pair_pt2pl),
                                         hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(unpack_pt2pl_kernel, dim3((uint32_t)((n_pt2pl + 255) / 256)), dim3(256),
                           0, ctx->stream,
                           reinterpret_cast<const mp2p_hip_pair_pt2pl*>(ctx->aos_stage.p),
                           (uint32_t)n_pt2pl, p->pl_coef.p, p->pl_cen.p, p->pl_lx.p, p->pl_ly.p,
                           p->pl_lz.p, p->pl_lidx.p);
    }
    unsigned long long h[5] = {n_pt2pt, n_pt2pl, 0, 0, 0};  // [5],[6] (lines, planes) are kept
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(p->counts.p, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_upload_lines_planes(mp2p_hip_ctx* ctx, mp2p_hip_pairs* p,
                                       const mp2p_hip_pair_pt2ln* pt2ln, size_t n_pt2ln,
                                       const mp2p_hip_pair_pl2pl* pl2pl, size_t n_pl2pl)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    MP2P_REQUIRE(ctx, p->ctx == ctx, "bad Pairings handle");
    MP2P_REQUIRE(ctx, (pt2ln || !n_pt2ln) && (pl2pl || !n_pl2pl), "null list with a non-zero count");
    MP2P_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (n_pt2ln)
    {
        MP2P_TRY_HIP(ctx, p->ln.ensure(n_pt2ln));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(p->ln.p, pt2ln, n_pt2ln * sizeof(*pt2ln), hipMemcpyHostToDevice, ctx->stream));
    }
    if (n_pl2pl)
    {
        MP2P_TRY_HIP(ctx, p->pp.ensure(n_pl2pl));
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(p->pp.p, pl2pl, n_pl2pl * sizeof(*pl2pl), hipMemcpyHostToDevice, ctx->stream));
    }
    const unsigned long long h[2] = {n_pt2ln, n_pl2pl};
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(p->counts.p + 5, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the sources are caller memory
    return MP2P_HIP_OK;
}

int mp2p_hip_pairs_counts_lines_planes(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* p, uint64_t* n_pt2ln,
                                       uint64_t* n_pl2pl)
{
    if (!ctx || !p) return MP2P_HIP_ERR_INVALID;
    unsigned long long h[2] = {0, 0};
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h, p->counts.p + 5, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_pt2ln) *n_pt2ln = h[0];
    if (n_pl2pl) *n_pl2pl = h[1];
    return MP2P_HIP_OK;
}

}  // extern "C"
