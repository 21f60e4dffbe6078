// common.hpp -- internal declarations shared by the HIP translation units of libmp2p_hip.so
// (gfx950 only; no CUDA shims, no host fallback).
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mp2p_hip.h"

namespace mp2p
{
constexpr uint32_t NONE_U32 = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------------------
// device cell hash entry: one occupied voxel of one grid level
// ---------------------------------------------------------------------------------------
struct Cell
{
    unsigned long long key;  // level<<60 | cz<<40 | cy<<20 | cx ; ~0 = empty
    uint32_t           start, end;  // range in the Morton-sorted point array
};
constexpr unsigned long long CELL_EMPTY = ~0ull;

struct GridView
{
    const float4* pts;  // sorted points {x,y,z,bits(original index)}
    uint32_t      n;
    const Cell*   table;
    uint64_t      mask;            // capacity-1
    float         ox, oy, oz;      // grid origin (= bbox min)
    float         hf, inv_hf;      // FINE cell edge (level "-shift0")
    uint32_t      shift0;          // fine -> level-0 shift
    uint32_t      n_levels;        // level j uses shift0+j
    float         bbmin[3], bbmax[3];
    float         slack;           // fp32 rounding slack for conservative geometric tests [m]
    // dense occupancy bitmaps, one per level, in 4x4x4 bricks (one u64 per brick, x fastest):
    // tested before a hash probe, because most voxels of a search box are empty and an
    // unsuccessful probe is a random 16-byte read of a table that does not fit any cache
    const unsigned long long* occ;          // all levels in one allocation, or null
    uint32_t                  occ_off[16];  // first word of level l, OCC_NONE = no bitmap for it
    uint32_t                  occ_bx[16], occ_by[16], occ_bz[16];  // bricks per axis
    // dense voxel directory: {first, one-past-last} sorted position of every voxel of the level's grid
    // (4 occ_bx x 4 occ_by x 4 occ_bz voxels, x fastest), empty voxels {0, 0}.  ONE load answers "which
    // points does this voxel hold" -- the bitmap test and the hash probe behind it are two dependent round
    // trips, and the search kernels are bound by exactly that chain.  64 x the bitmap's size: kept for the
    // levels that fit the budget (MP2P_HIP_TUNE dir_budget_mb; HBM is 288 GB), the hash table serves the rest.
    const uint2*              dir;
    unsigned long long        dir_off[16];  // first entry of level l, DIR_NONE = no directory for it
};
constexpr uint32_t           OCC_NONE = 0xFFFFFFFFu;
constexpr unsigned long long DIR_NONE = ~0ull;

// ---------------------------------------------------------------------------------------
// host-side objects behind the opaque C handles
// ---------------------------------------------------------------------------------------
// device allocations made through DevBuf since the library was loaded (mp2p_hip_debug_alloc_count: tests assert that a
// steady-state iteration of a solver / matcher makes none)
inline std::atomic<unsigned long long>& dev_alloc_counter()
{
    static std::atomic<unsigned long long> c{0};  // one context per thread / GPU: incremented concurrently
    return c;
}

template <class T>
struct DevBuf
{
    T*     p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) return hipSuccess;
        dev_alloc_counter()++;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
    hipError_t ensure(size_t count)
    {
        if (count <= n) return hipSuccess;
        return alloc(count + count / 8);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
    // owning, move-only: temporaries free themselves on every return path
    DevBuf() = default;
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&)            = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o)
        {
            release();
            p = o.p, n = o.n;
            o.p = nullptr, o.n = 0;
        }
        return *this;
    }
};

// knobs read once per context from the environment variable MP2P_HIP_TUNE ("name=value,..."):
// for measurements only, every setting computes the same results
struct Tune
{
    uint32_t lane_cells    = 0;     // widest cube (level-0 voxels per axis) the one-query-per-lane kernel searches
                                    // itself; 0 = none (measured: per-lane 16-byte gathers cost one L1 line each,
                                    // the tile kernel's coalesced staging serves the same queries 2.5x cheaper)
    uint32_t tile_cand_cap = 0;     // staged candidates after which a tile hands its pending queries on; 0 = 24 576 for a large layer with
                                    // the matrix-pipe selection (round 5: what a tile stages is what its balls reach, the long tiles
                                    // start first, and a tile that gives up costs 32 one-query searches), else 6 144
    uint32_t tile_cand_cap_easy = 0;  // ... for the tiles of the class served LAST (0 = the same)
    int      tile_bricks   = 1;     // tile kernel: wide groups list their voxels from the level-0 occupancy bricks and stay in
                                    // the tile (0 = round 3: a query beyond the deferral radius goes to the one-query kernel)
    uint32_t coop_max      = 0xFFFFFFFFu;  // tile kernel: a group of at most this many queries is handed to the one-query kernel;
                                    // default: 0 for a large layer with the selection (an isolated query stages little: it stays), else 4
    int      nn_cert       = 1;     // point-to-point search: skip the search of a query whose previous neighbour is certainly still
                                    // the nearest (NNArgs::lb2nd); 1 = bounds tracked after small steps only, 2 = always, 0 = off
    uint32_t nn_cert_step_mm = 5;   // ... "small": the farthest local point moved less than this since the previous call
    int      empty_room    = 1;     // one-query kernel: empty-cube bound for queries with nothing in reach (they skip later calls)
    uint32_t tile_brick_budget = 512;  // ... when the group's box spans at most this many bricks
    uint32_t hard_cand     = 1700;  // a query whose tile staged this many candidates at the previous call joins the hard class
                                    // (dispatched first) whatever its radius; 0 = by radius only
    uint32_t hard_radius_pct = 100; // pending queries with a radius above this % of a level-0 voxel are "hard":
                                    // their tiles are dispatched first (nn_query.hip)
    int      xcd_map       = 1;     // tile kernel: one segment of the pending list per XCD (L2 locality)
    uint32_t single_waves  = 0;     // one-query-per-wave kernel: register budget for 4 waves per SIMD (0 = 5: 96 VGPRs; 4 = the compiler's
                                    // own 108 VGPRs: -3..-11 % slower; the 6- and 8-wave builds spilled 68 / 140 bytes per lane, were
                                    // slower and are gone since round 6: the values are accepted and mean the default)
    uint32_t single_blocks_per_cu = 0;  // ... and its grid, in workgroups per CU (0 = 40: two resident rounds at 5 waves)
    uint32_t pl_q          = 0;     // point-to-plane search: queries per wave (0 = by layer size: 8 up to 400 k points, else 32)
    int      sync_spin     = 1;     // wait for the stream by polling hipStreamQuery (lower wake-up latency)
    int      spin_us       = 2000;  // ... tight for this long, then yielding, then (8x) the blocking wait
    int      far_pass      = 1;     // one-query kernel: one pass at 1.5 r_max for a query with nothing within the threshold and no empty room around it:
                                    // its bound then outlives displacements (C5: 300 000 outliers were searched again at every call; 4.49 -> 4.26 ms/step)
    int      claim_dedup   = 1;     // in-wave minimum per global point before the global atomic
    int      claim_peek    = 1;     // plain look at the claim word before the atomic
    int      compact_fused = 1;     // compaction: bounding-box reduction folded in
    uint32_t tile_waves    = 0;     // tile kernel (matrix-pipe variant): register budget for this many waves per SIMD.  0 = by the path:
                                    // nn_seltile_kernel behind the lane kernel 4 (128 VGPRs, 3 spilled dwords since the staging loads are
                                    // issued four at a time: still the fastest, 2 254 vs 2 193 it/s with 3), with the fused prologue 3
                                    // (no spill, as fast as 4 on C2); round 4's nn_tile_kernel: 4 (no spill; 5 and 6 spill / are slower)
    uint32_t pipelines     = 1;     // 2 = independent lane -> tile -> one-query chains over halves of the local layer
                                    // on two streams (one chain's drain filled by the other's kernels): -6 % search
                                    // time on scene B, +4 % on scene A
    int      mfma_scan     = 1;     // tile kernel: distance tests of a tile on the matrix pipe as a prefilter (always, since round 6: the exact-scan
                                    // builds of rounds 1-3 are gone; the knob is accepted without effect)
    uint32_t dir_budget_mb = 8192;  // dense voxel directories of a map: at most this many MB (0 = none)
    int      pl_warm       = 1;     // point-to-plane search: start radius from the previous call's k-th distance (0 = full radius)
    int      pl_cert       = 2;     // pt2pl: skip the search of a query whose previous list is certainly still its k nearest (PlArgs::lb_io)
    uint32_t pl_cert_step_mm = 10;  // ... only when the farthest local point moved less than this since the previous call (0 = always): the gap
                                    // between a query's k-th and (k+1)-th neighbour is millimetres, so after a larger step no list is
                                    // certified, while reading the lists (five gathers per query) and staging the margin cost 7 % of a C3 step
    uint32_t pl_cert_pad   = 2;     // ... searchRadius + this many per mille is what a search that comes up short asks for
    uint32_t pl_cert_margin_mm = 20; // ... voxels up to this far beyond the search radius of a pass are staged as well (the covered region's margin)
    uint32_t copy_chunk_kb = 2048;  // staged copy-out of the pair lists (CopyStage): bytes per DMA + event ...
    uint32_t copy_stage_mb = 256;   // ... and the bound of the page-locked staging buffer (larger lists go in rounds)
    int      tile_select   = 1;     // pt2pt search, round 5: the voxels a tile stages are SELECTED on the matrix pipe (a voxel is staged iff
                                    // some query's ball reaches its circumsphere) instead of by the group's bounding box (nn_seltile.hip)
    int      nn_direct     = -1;    // ... 1 = the per-query prologue runs in the tile itself (no lane kernel, no pending list), 0 = behind the
                                    // lane kernel (whose cost classes start the long tiles first: measured faster on the 1 M-point bench
                                    // chain), -1 = in the tile for small layers (<= 262 144 points) only
    uint32_t grp_all_bricks = 6;    // ... all pending queries of a tile are served by ONE pass while their common box is at most this many
                                    // 4-voxel bricks wide (0 = always the seed's neighbourhood, the rule of rounds 1-4)
    int      tile_sol      = 0;     // speed-of-light decomposition of that kernel (TIMING ONLY, no results): 1 = list + select + stage,
                                    // 2 = + matrix-pipe prefilter; set at run time through mp2p_hip_set_tune
    int      pl_select     = -1;    // point-to-plane / k-NN search, round 6: the voxels of a pass SELECTED per query ball on the matrix pipe and the
                                    // distance tests prefiltered there (nn_pl_seltile.hip); 0 = the box-rule tile kernel of rounds 2-5; -1 = by the
                                    // layer's size: above 524 288 queries (measured: C5, 5 M queries, 6.46 -> 5.41 ms per step; C3, 120 k queries =
                                    // 3 750 tiles of 32 for 3 072 wave slots, is as long as its longest tile and LOSES: 2 117 -> 1 771 it/s)
    uint32_t pl_waves      = 0;     // ... 1 = every tile by ONE wave (no hard class); default: the tiles of the hard class by the four waves of a workgroup
    uint32_t pl_sel_margin_mm = 5;  // ... the certificate's margin of that kernel: the selection's balls and the prefilter's limits are this much wider
                                    // than the search needs, so that what is not evaluated exactly is provably this far beyond the list
    uint32_t pl_sel_hard_cand = 3000;  // ... a query whose 32-query tile staged this many candidates at the previous call is listed in the class dispatched first (0 = one class)
    uint32_t pl_sel_hard_large = 1500; // ... the same for layers above 524 288 queries, where the class is served by single waves and only sets the dispatch order (0 = Morton order)
    int      pl_sol        = 0;     // timing-only cuts of pt2pl_seltile_kernel's instrumented build (1..4, nn_pl_seltile.hip): set_tune only, profiling on, results invalid
    int      pl_no_touch   = 0;     // profiling level 2 of the point-to-plane search without the per-point 'touched' bytes (phase timers undisturbed)
    uint32_t pl_hard_cand  = 3000;  // pt2pl: a query whose tile staged this many candidates (per 4 queries) at the previous call is searched in the hard class, first and in smaller tiles (0 = one class)
};

// multi-GPU communicator of a context (comm.hip): RCCL, or caller-provided collectives
struct Comm
{
    void* nccl = nullptr;  // ncclComm_t
    int   rank = 0, nranks = 0;
    mp2p_hip_allreduce_fn hook_allreduce = nullptr;
    mp2p_hip_allgather_fn hook_allgather = nullptr;
    void*                 hook_user      = nullptr;
    size_t                cap_guess      = 0;  // record-list length predicted for the next iteration (0: ask)
    DevBuf<unsigned long long> pad, gathered;
};

// Device -> caller memory for the large pair lists (pairs.hip): DMA into the context's OWN page-locked buffer in
// chunks, one event per chunk, and plain host copies from there into the caller's (pageable) container while the
// later chunks are still on the link -- by the calling thread and one helper thread of the context.  Round 3
// page-locked the caller's destination itself for the duration of the copy (hipHostRegister / hipHostUnregister of an
// unaligned sub-range of the malloc heap): that made later pageable copies of the runtime from neighbouring heap
// memory fault on the GPU ("Memory access fault by GPU node" -> abort(), GPUTEST_r03).  The
// library no longer registers memory it does not own.
struct CopyStage
{
    unsigned char*          host = nullptr;  // hipHostMalloc, grow-only
    size_t                  cap  = 0;
    std::vector<hipEvent_t> ev;              // one per chunk of a round
    size_t                  chunk = 2u << 20, stage_max = 256u << 20;  // Tune::copy_chunk_kb / copy_stage_mb
    std::mutex              mu;              // guards everything below
    std::condition_variable cv_work, cv_done;
    unsigned char*          out      = nullptr;  // the round in flight: chunk k = [k * chunk, ...) of `bytes`
    size_t                  bytes    = 0, n_chunks = 0, next = 0, done = 0;
    int                     err      = 0;        // first hipError_t a chunk's event reported
    const unsigned char*    dev_rest = nullptr;  // what is left of a copy larger than the staging buffer
    unsigned char*          out_rest = nullptr;
    size_t                  rest     = 0;
    bool                    open     = false;    // a posted copy has not been finished yet
    std::thread             th;                  // helper (started with the first copy of more than one chunk)
    bool                    stop     = false;
    bool                    no_helper = false;   // the helper could not be started: the caller copies alone
    // ---- round 6, the point pairings as 24 instead of 44 bytes per pair on the link (mp2p_hip_pairs_copy_pt2pt_begin_soa): the
    //      staging buffer holds {gx, gy, gz, err} per pair (packed on the device), DMA'd chunk by chunk (soa_cp pairs each); whoever claims a chunk ASSEMBLES its 36-byte records: indices from the caller's two index arrays
    //      (copied first, same stream: there when the chunk's event is), `local` from the caller's own layer arrays
    size_t                  soa_n = 0, soa_cp = 0;  // soa_n != 0: the round in flight is such a copy
    const uint32_t*         soa_li = nullptr;
    const uint32_t*         soa_gi = nullptr;
    const float*            soa_l[3] = {nullptr, nullptr, nullptr};
    size_t                  soa_nl = 0;
    unsigned long long      soa_base = 0;        // localIdx - soa_base indexes soa_l
    int                     soa_bad = 0;         // a localIdx outside the caller's arrays was met (reported by _end)
};

struct GnState
{
    const mp2p_hip_pairs* pairs = nullptr;
    mp2p_hip_gn_params    prm{};
    bool                  active = false;
    double                pose0[12] = {};
    bool                  state_ready = false;  // gn_state holds {pose0, zeros} or a later iterate
};

}  // namespace mp2p

struct mp2p_hip_ctx
{
    int         device     = 0;
    hipStream_t stream     = nullptr;
    bool        own_stream = false;
    std::string err;
    int         profiling = 0;  // 0 off, 1 hipEvent timing of every stage, 2 + device counters,
                                // 3 only the two events around the search kernels, 4 = 3 + a
                                // {start, end} timestamp per workgroup of the search kernels
    bool        prof_all() const { return profiling == 1 || profiling == 2; }
    hipEvent_t  ev[8]     = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int         pending_match = 0, pending_gn = 0, pending_lane = 0, pending_pl = 0;
    size_t      pending_map_n = 0;
    mp2p_hip_stats stats{};
    uint64_t    epoch = 0;  // claim epoch (see nn_query.hip)

    // per-call scratch, grown on demand
    mp2p::DevBuf<uint32_t>           nn_spos;      // [n_local] by ORIGINAL local index
    mp2p::DevBuf<float>              nn_d2;        // [n_local]
    mp2p::DevBuf<float>              tile_bbox;    // [n_tiles][6]
    mp2p::DevBuf<float>              tile_bbox2;   // [64][6] second reduction level
    mp2p::DevBuf<float>              block_bbox;   // [compaction blocks][6] (fused box reduction, pairs.hip)
    uint32_t                         last_n_boxes = 0;       // per-wave boxes the last pt2pt search left in tile_bbox
    bool                             sol_no_records = false;    // the last pt2pt search was a timing-only launch (Tune::tile_sol): nothing to compact
    mp2p_hip_pairs*                  clear_deferred = nullptr;  // a list whose clear the next compaction into it performs (mp2p_hip_step_sharded on one rank)
    bool                             q_counters_clean = false;  // the search's list counters are zero on the stream
    mp2p::DevBuf<float>              local_bbox;   // [6] min xyz, max xyz of transformed local
    mp2p::DevBuf<double>             exch;         // [8] what a sharded layer all-reduces (pairs.hip)
    mp2p::DevBuf<unsigned long long> claim_list;   // [n_l + 1] claim records + their count
    mp2p::DevBuf<uint32_t>           block_counts; // compaction
    mp2p::DevBuf<unsigned long long> counters;     // profiling counters
    mp2p::DevBuf<double>             gn_partials;  // [GN_BLOCKS][NSUMS]
    mp2p::DevBuf<unsigned char>      compact_flags;  // the count pass's flags, one byte per thread
    mp2p::DevBuf<double>             gn_sums;      // [NSUMS]
    mp2p::DevBuf<double>             gn_state;     // pose(12) H(36) g(6) cost(1) iters(1) done(1)
    mp2p::DevBuf<unsigned char>      aos_stage;    // download staging
    mp2p::DevBuf<unsigned char>      pl_slots;     // pt2pl per-query plane slots
    mp2p::DevBuf<uint32_t>           pl_knn;       // pt2pl neighbour lists [n_local][K] (search -> fit kernel)
    mp2p::DevBuf<float>              pl_kth;       // pt2pl warm start: d2 of every query's k-th neighbour at the previous call
    mp2p::DevBuf<uint32_t>           pl_hard;      // pt2pl: the hard class's query list
    const void*                      pl_hard_cnt_at = nullptr;
    bool                             pl_lists_dirty = false;
    mp2p::DevBuf<uint32_t>           pl_cost;      // pt2pl scheduling hint: ticks of the tile that served each query at the previous call
    mp2p::DevBuf<float>              pl_lb;        // pt2pl certificate: lower bound of the distance to every point outside the list
    mp2p::DevBuf<uint32_t>           pl_pend;      // pt2pl: queries left to the search, a segment of 256 per block of the certificate kernel
    mp2p::DevBuf<uint32_t>           pl_pend_cnt;
    mp2p::DevBuf<unsigned long long> pl_cert_stat; // {queries certified, queries searched} since the context was created
    const void*                      pl_hint_map = nullptr;    //   ... and what that call was made on
    const void*                      pl_hint_cloud = nullptr;
    size_t                           pl_hint_n = 0;
    uint32_t                         pl_hint_knn = 0;
    double                           pl_hint_rad = 0.0;
    double                           pl_hint_pose[12] = {};
    mp2p::DevBuf<unsigned long long> timeline;     // profiling level 4: {start, end} ticks per workgroup
    size_t                           timeline_tiles = 0, timeline_singles = 0;
    mp2p::DevBuf<unsigned char>      horn_flags;   // Horn: scale-outlier flag per point pairing
    size_t                           horn_flags_zero = 0;  // bytes of horn_flags known to be zero on the stream (no pass with the scale detector wrote them since)
    mp2p::DevBuf<unsigned long long> horn_bounds;  //   first pair of each point_weights block, [MP2P_HIP_MAX_WEIGHT_BLOCKS] = error
    size_t                           horn_n = 0;   //   pairings the flags belong to
    mp2p::DevBuf<unsigned long long> ad_hist;      // Matcher_Adaptive: 50 bins, count, {min,max} words
    uint32_t                         ad_knn   = 0;       //   lists held in nn_spos / nn_d2: neighbours per point,
    const void*                      ad_cloud = nullptr; //   and the handles they were searched for
    const void*                      ad_map   = nullptr;
    bool                             ad_apart = false;   //   the search returned before any launch: the layers' boxes cannot meet (select emits nothing)
    mp2p::DevBuf<float>              nn_lb2nd;     // [n_local] pt2pt certificate: bound of the distance to every point but the nearest
    bool                             nn_lb2nd_valid = false;  // ... written by the previous pt2pt call (on hint_map / hint_cloud)
    mp2p::DevBuf<uint4>              nn_rec;       // [n_local] result + warm-start records of the pt2pt search
                                                   // (Morton order of the local layer; see nn_query.hip)
    mp2p::DevBuf<uint4>              work, pend;   // deferred / pending queries of the NN search
    mp2p::DevBuf<uint4>              work_q, pend_q;  // {qx, qy, qz, best_spos} of the list entries
    mp2p::DevBuf<uint32_t>           q_counters;   //   {#pending, #deferred}
    // reusable scratch of the per-call temporaries of the solvers / matchers / filters that used to hipMalloc and
    // hipFree 6..15 buffers per call: grown on demand, never shrunk; a slot belongs to one temporary
    // of the function that runs (mp2p::Scratch)
    mp2p::DevBuf<unsigned char>      scratch[16];
    mp2p::Tune                       tune;         // MP2P_HIP_TUNE (experiments; defaults otherwise)
    double                           hint_pose[12] = {};
    const void*                      hint_map   = nullptr;
    const void*                      hint_cloud = nullptr;
    size_t                           hint_n     = 0;
    mp2p::GnState                    gn;
    mp2p::Comm                       comm;
    uint32_t last_n_tiles = 0;
    uint32_t last_q       = 64;
    void*    pinned       = nullptr;  // 4 KB of page-locked host memory for the small read-backs
    hipStream_t stream2    = nullptr;    // second search pipeline (launch_nn_pt2pt)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t copy_ev     = nullptr;    // mp2p_hip_pairs_copy_pt2pt_begin: the index arrays have arrived
    bool       copy_open   = false;
    mp2p::CopyStage cstage;              // staged copy-out of the pair lists (pairs.hip)
    mp2p_hip_cloud* q1_cloud = nullptr;  // mp2p_hip_nn_search_pt2pl: the one-point query layer
    mp2p_hip_pairs* q1_pairs = nullptr;
};

struct mp2p_hip_map
{
    mp2p_hip_ctx*                    ctx = nullptr;
    size_t                           n   = 0;
    mp2p::DevBuf<float4>             pts;     // Morton-sorted {x,y,z,idx}
    mp2p::DevBuf<mp2p::Cell>         table;
    mp2p::DevBuf<unsigned long long> claims;  // [n] indexed by SORTED position
    mp2p::DevBuf<unsigned long long> occ;     // occupancy bitmaps of all levels
    mp2p::DevBuf<uint2>              dir;     // dense voxel directories of the levels that fit
    mp2p::GridView                   view{};
    mp2p_hip_map_info                info{};
};

struct mp2p_hip_cloud
{
    mp2p_hip_ctx*        ctx = nullptr;
    size_t               n   = 0;
    mp2p::DevBuf<float4> sorted;  // Morton-sorted (own frame) {x,y,z,idx}
    mp2p::DevBuf<float>  x, y, z; // original order (pair output)
    mp2p::DevBuf<uint32_t> pos;   // original index -> place in `sorted` (search results are kept in that order)
    float                  radius = 0.f;  // largest |p| over the layer's points (own frame): bounds a rotation's displacement
    // optional visit order (mp2p_hip_cloud_set_visit_order): order[r] = original index visited
    // r-th, rank = its inverse (NONE for points that are not visited); n_visit == 0: all, ascending
    size_t                 n_visit = 0;
    mp2p::DevBuf<uint32_t> order, rank;
};

struct mp2p_hip_mstate
{
    mp2p_hip_ctx*               ctx = nullptr;
    mp2p::DevBuf<unsigned char> global_taken, local_taken;  // by ORIGINAL index
};

struct mp2p_hip_pairs
{
    mp2p_hip_ctx* ctx = nullptr;
    size_t        cap_pt2pt = 0, cap_pt2pl = 0;
    // pt2pt SoA
    mp2p::DevBuf<uint32_t> lidx, gidx;
    mp2p::DevBuf<float>    lx, ly, lz, gx, gy, gz, err;
    // pt2pl SoA
    mp2p::DevBuf<uint32_t> pl_lidx;
    mp2p::DevBuf<double>   pl_coef;  // [cap][4]  a,b,c,d
    mp2p::DevBuf<double>   pl_cen;   // [cap][3]
    mp2p::DevBuf<float>    pl_lx, pl_ly, pl_lz;
    // host-produced lists kept as uploaded (AoS): paired_pt2ln, paired_pl2pl
    mp2p::DevBuf<mp2p_hip_pair_pt2ln> ln;
    mp2p::DevBuf<mp2p_hip_pair_pl2pl> pp;
    // counts: [0]=n_pt2pt [1]=n_pt2pl [2]=potential_pairings [3]=write base scratch
    // [4]=overflow flag [5]=n_pt2ln [6]=n_pl2pl
    mp2p::DevBuf<unsigned long long> counts;
};

namespace mp2p
{
// a typed, non-owning view of one scratch slot of the context
template <class T>
struct Scratch
{
    T* p = nullptr;
    hipError_t take(mp2p_hip_ctx* ctx, int slot, size_t count)
    {
        const hipError_t e = ctx->scratch[slot].ensure((count ? count : 1) * sizeof(T));
        p                  = reinterpret_cast<T*>(ctx->scratch[slot].p);
        return e;
    }
};
// hipCUB / rocPRIM entry points take the item count as int
#define MP2P_REQUIRE_INT_COUNT(ctx, n) MP2P_REQUIRE(ctx, (unsigned long long)(n) <= 2147483647ull, "more items than a hipCUB call takes (2^31 - 1)")
int  set_err(mp2p_hip_ctx* ctx, int code, const char* fmt, ...);
// hipStreamSynchronize, or (tune.sync_spin) a poll of hipStreamQuery: no wake-up latency
inline hipError_t stream_wait(mp2p_hip_ctx* ctx)
{
    if (!ctx->tune.sync_spin) return hipStreamSynchronize(ctx->stream);
    // bounded: a tight poll for spin_us (one ICP step is 0.3..0.8 ms), then polls that yield the core, then the
    // blocking wait -- a long kernel chain or a collective that waits on a slow peer must not burn a host core
    const auto t0 = std::chrono::steady_clock::now();
    const long spin_us = ctx->tune.spin_us;
    for (unsigned it = 0;; ++it)
    {
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e != hipErrorNotReady) return e;
        if ((it & 31u) != 31u) continue;
        const long us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (us > 8 * spin_us) return hipStreamSynchronize(ctx->stream);
        if (us > spin_us) std::this_thread::yield();
    }
}
void set_global_err(const char* fmt, ...);

#define MP2P_TRY_HIP(ctx, expr)                                                              \
    do                                                                                       \
    {                                                                                        \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return mp2p::set_err((ctx), MP2P_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,    \
                                 hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

#define MP2P_REQUIRE(ctx, cond, msg)                                                         \
    do                                                                                       \
    {                                                                                        \
        if (!(cond)) return mp2p::set_err((ctx), MP2P_HIP_ERR_INVALID, "%s (%s)", msg, #cond); \
    } while (0)

// implemented in the .hip units
int build_map(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z, size_t n,
              const mp2p_hip_map_params* prm, mp2p_hip_map* map);
int build_cloud(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z,
                size_t n, mp2p_hip_cloud* cloud);

}  // namespace mp2p
