// nn_query.hip -- K1 (fused) + K3: local->global transform and EXACT nearest-neighbour search.
//
// Replaces, per outer ICP iteration,
//   transform_local_to_global           Matcher_Points_Base.cpp:183-249
//   the nn_single_search loop            Matcher_Points_DistanceThreshold.cpp:214-243
//   the threshold rule                   :252-259
//   the "global point already paired"    :94-121 (claims; resolved in pairs.hip)
//
// Three kernels, all one wave64 per workgroup; a query is finished by the first one that can
// (DESIGN.md section 4 has the measurements behind every choice made here).  Since round 5 the tile kernel that runs is
// nn_seltile_kernel (nn_seltile.hip: voxels selected on the matrix pipe); nn_tile_kernel below serves maps without a
// level-0 occupancy bitmap (beyond the bitmap budget, or built with no_occupancy_bitmap) and the 16 / 64-query tile sizes:
//
//  nn_lane_kernel  -- ONE QUERY PER LANE, no LDS staging, no wave-level coordination.  The lane
//     transforms its point (K1), reads its warm-start record, and -- when the cube its search
//     radius spans is at most 4 level-0 voxels per axis (the normal case from the second ICP
//     iteration on: the radius is the distance to the previous nearest neighbour) -- builds the
//     64-bit occupancy mask of that cube from the <= 8 bitmap bricks it overlaps, then walks the
//     occupied voxels inside the ball: one hash probe per voxel, 16-byte point loads, 4 in flight.
//     Registers stay below 64 -> 8 waves per SIMD hide the dependent loads (bitmap -> probe ->
//     points) that bound the tile kernel.  Queries it cannot conclude (radius too wide, nothing
//     within the first radius of a cold start) are appended to the PENDING list with their state.
//
//  nn_tile_kernel  -- a TILE of Q consecutive PENDING queries per wave.  lane = (query slot,
//     candidate slice).  The wave resolves the voxels overlapping the search box of the current
//     GROUP of queries (dense voxel directory: one 8-byte load per voxel), stages their points
//     into LDS (SoA) with coalesced 16-byte loads, then tests the staged bucket against the 32
//     queries: d2 of 32 candidates x 32 queries per three v_mfma_f32_32x32x2_f32 on box-centred
//     coordinates as a PREFILTER (error bounded by the box), the few candidates within that bound
//     of a query's running best recomputed in the exact FMA-free sequence (Q = 32; Q = 16 / 64 and
//     MP2P_HIP_TUNE=mfma_scan=0 scan exactly, 8 candidates per step on packed fp32).  Queries that
//     are spatially isolated inside their tile, whose radius outgrows the voxels, or whose tile has
//     already staged more than its budget or run longer than its time budget are DEFERRED to
//
//  nn_single_kernel -- one query per wave: the 64 lanes split the candidates (one coalesced
//     16-byte load each, no LDS staging), and a wave arg-min merges them.  Deferred queries
//     are spread over the whole chip instead of serialising inside one tile's wave.
//
// Exactness: argmin is lexicographic on (fp32 d2, original global index) -> independent of the
// traversal order, ties resolve to the lowest index.  A query is final when its best distance
// is below the radius its visited voxels are guaranteed to cover, or when that radius has
// reached r_max = sqrt(threshold rule), beyond which the reference discards the pair anyway
// (so bounding the unbounded nn_single_search there is result-equivalent).  Otherwise the
// radius grows (to the best distance found, else x2) and the search repeats at a coarser level.
//
// Output: one 16-byte record per query {sorted position of the nearest neighbour, d2, lower
// bound^2 for the next call's warm start, accepted flag}, in the Morton order of the local layer
// (one aligned store per query; it doubles as the warm-start record of the next call).
// Claims: atomicMin(claim[spos], epoch | visit rank) -- a device-scope atomic is a 64-byte
// memory-side transaction on this part, and Morton-neighbouring queries mostly fight over the
// same global point, so the wave first resolves its own minimum per distinct global point in an
// LDS table and only the winner (after a plain look at the current claim) issues the atomic.
#include "device_utils.hpp"

namespace mp2p
{
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NN_CAP      = 256;  // staged candidates per round (LDS: 5 x 1 KB)
constexpr int NN_COOP_MAX = 4;    // groups of up to this many queries are deferred
constexpr int NN_CLAIM_SLOTS = 128;  // in-wave claim table (LDS)
constexpr int NN_MAX_SEG     = 256;  // segments of a query list
constexpr int NN_LISTS       = 3;    // 0 = pending/hard, 1 = deferred, 2 = pending/easy
constexpr int NN_ALL_LISTS   = 3;
constexpr int NN_CNT_STRIDE  = 32;   // uint32 words between two segment counters (128 bytes)
constexpr int NN_TVLIST      = 512;  // tile kernel, wide groups: occupied voxels listed per round (LDS)
constexpr uint32_t NN_PASS_COST  = 320; // tile kernel: what a pass costs, in staged candidates (10 us against 30 us per 960)
constexpr uint32_t NN_COST_SHIFT = 8;  // result record, word 3: bit 0 = accepted, bits 8.. = what the query's tile staged

struct NNArgs
{
    GridView      g;
    const float4* lpts;  // Morton-sorted local points {x,y,z,bits(orig idx)}
    uint32_t      n_l;
    PoseRt        pose;
    float         maxDistSq, angSq;  // Matcher_Points_DistanceThreshold.cpp:82-83
    float         r0;                // first search radius [m]
    float         grp_factor;        // group extent in units of the seed's radius
    float         r_defer;           // radius beyond which a query leaves its tile
    uint32_t      cell_budget;       // voxels of a search box per pass
    uint32_t      brick_budget;      // 4x4x4 bricks per pass of the one-query-per-wave kernel
    uint32_t      lane_cells;        // widest cube (level-0 voxels per axis, <= 4) a lane searches itself; 0 = never
    uint32_t      tile_cand_cap;     // staged candidates after which a tile hands its pending queries on (the class served first ...
    uint32_t      tile_cand_cap_easy;  // ... and the other: a long tile there was MISpredicted and starts late -- it is cut short)
    // ---- search-skip certificate of the point-to-point search (round 4; the rule of the point-to-plane search's, nn_pt2pl.hip):
    //      lb2nd[qi] (Morton order) = a lower bound of the distance from the query, at the pose of the call that wrote it, to
    //      every map point OTHER than its recorded nearest neighbour; 0 = none.  Written by every kernel that finishes a query
    //      (the tracking build of the tile kernel derives it from the prefilter's second-smallest value minus its error bound;
    //      everybody else writes 0 or the decayed old value); read by the lane kernel when the previous call wrote it.
    float*        lb2nd;             // null: neither read nor written by this call
    int           cert_read;         // the previous call on this map and layer left bounds: the lane kernel may certify
    int           empty_room;        // one-query kernel: a query with nothing in reach looks for an empty cube beyond its radius
    int           tile_bricks;       // wide groups stay in their tile: voxels of their box listed from the level-0 occupancy bricks
    uint32_t      tile_brick_budget; // ... when the box spans at most this many bricks (else the coarser dense box)
    uint32_t      coop_max;          // a group of at most this many queries leaves its tile for the one-query kernel
    uint32_t      hard_cand;         // a query whose tile staged at least this many candidates at the previous call is hard (0: by radius only)
    int           far_pass;          // one-query kernel: a query with nothing within the threshold gets one pass at 1.5 r_max for its bound
    int           claim_dedup, claim_peek;
    int           mfma_scan;         // tile kernel, Q = 32: distance tests on the matrix pipe as a prefilter
    const unsigned char* local_taken;   // by original local index, or null
    const unsigned char* global_taken;  // by original global index, or null
    unsigned long long*  claims;        // by sorted global position, or null
    unsigned long long   claim_hi;      // (~epoch) << 32
    unsigned long long   local_offset;  // whole-layer index of this rank's first local point
    // result + warm start, [n_l] in the order of lpts: {sorted position of the nearest neighbour
    // (NONE: none found), d2, lower bound on the SQUARED distance to every map point at this call's
    // pose, accepted (passes the threshold and is not pre-taken)}; read at entry when use_hint
    uint4*               rec;
    float*               tile_bbox;  // [n_waves of the lane kernel][6]
    // pending queries (lane kernel -> tile kernel) and deferred queries (-> one-query-per-wave
    // kernel): {sorted local idx, r, best_d2, best_idx} + {qx, qy, qz, best_spos}.  A device-scope atomic is a
    // round trip to the memory side and atomics on ONE address serialise there (~12 ns each: one
    // counter bumped by every wave set the duration of the whole kernel), so each list is cut into
    // n_seg SEGMENTS of seg_cap entries, one counter per segment on its own 128-byte line; the waves
    // of seg_waves consecutive workgroups of the lane kernel share a segment (Morton-consecutive
    // queries stay together).  q_counters[(list * NN_MAX_SEG + seg) * NN_CNT_STRIDE]
    uint4*               pend;
    uint4*               pend_q;  // {qx, qy, qz (the transformed point), best_spos}: a pending query is read in
    uint4*               work;    // ONE round trip (its point would otherwise be a dependent second one)
    uint4*               work_q;
    uint32_t*            q_counters;
    uint32_t             n_seg, seg_waves, seg_cap, tiles_per_seg;
    // one of several independent PIPELINES over the local layer (launch_nn_pt2pt): this launch serves the
    // waves wave_base .. of the lane kernel = the segments seg_base .. seg_base + n_seg - 1
    uint32_t             wave_base, seg_base;
    // the pending list comes in two classes, hard (list 0: radius above r_hard) and easy (list 2, stored
    // behind the hard entries at pend + list_cap): the tile kernel's grid serves the hard class FIRST.
    // Workgroups are dispatched in index order and a hard tile runs 5-8x as long as an easy one; started
    // last it kept a nearly empty chip waiting (45 % of the kernel's span), started first its tail is
    // covered by the short, uniform easy tiles.  It also keeps far queries out of the easy tiles' boxes.
    uint32_t             list_cap;
    float                r_hard;
    int                  xcd_map;
    float                grp_all_bricks;  // nn_seltile_kernel: all pending queries of a tile form one pass while their common box is at most this many bricks wide
    int                  direct;   // nn_seltile_kernel: tile t serves the queries 32 t .. of the layer itself (no lane kernel, no pending list)
    const uint32_t*      rank;  // visit rank per original local index (NONE = not visited) or null
    int                  use_hint;
    PoseRt               prev_pose;
    unsigned long long*  counters;  // profiling, or null
    unsigned char*       touched;   // profiling: [n_g] by sorted position, or null
    // profiling level 4: {start, end} 100 MHz ticks of every workgroup of the tile and one-query
    // kernels ([n_tiles] then [single blocks]); the plain kernels only pay a uniform null test
    unsigned long long*  timeline;
    uint32_t             timeline_single_base;
};

// ---- geometry of one search pass (all values wave-uniform) -----------------------------------
struct PassBox
{
    uint32_t nx, ny, nz, cx0, cy0, cz0, s, lev;
    unsigned long long ncell;
    float hs, inv_nx, inv_ny;
};

__device__ __forceinline__ PassBox choose_level(const GridView& g, float lox, float loy, float loz,
                                                float hix, float hiy, float hiz, uint32_t budget)
{
    PassBox b;
    b.nx = b.ny = b.nz = b.cx0 = b.cy0 = b.cz0 = 0, b.s = g.shift0, b.lev = 0, b.ncell = 0;
    // clip to the layer's bounding box; disjoint -> nothing to visit
    lox = fmaxf(lox, g.bbmin[0]), loy = fmaxf(loy, g.bbmin[1]), loz = fmaxf(loz, g.bbmin[2]);
    hix = fminf(hix, g.bbmax[0]), hiy = fminf(hiy, g.bbmax[1]), hiz = fminf(hiz, g.bbmax[2]);
    if (!((lox > hix) || (loy > hiy) || (loz > hiz)))
    {
        const uint32_t flx = cell_fine(lox, g.ox, g.inv_hf), fhx = cell_fine(hix, g.ox, g.inv_hf);
        const uint32_t fly = cell_fine(loy, g.oy, g.inv_hf), fhy = cell_fine(hiy, g.oy, g.inv_hf);
        const uint32_t flz = cell_fine(loz, g.oz, g.inv_hf), fhz = cell_fine(hiz, g.oz, g.inv_hf);
        for (;;)
        {
            b.cx0 = flx >> b.s, b.cy0 = fly >> b.s, b.cz0 = flz >> b.s;
            b.nx = (fhx >> b.s) - b.cx0 + 1, b.ny = (fhy >> b.s) - b.cy0 + 1,
            b.nz = (fhz >> b.s) - b.cz0 + 1;
            b.ncell = (unsigned long long)b.nx * b.ny * b.nz;
            if (b.ncell <= budget || b.lev + 1 >= g.n_levels) break;
            b.s++, b.lev++;
        }
    }
    b.hs = g.hf * (float)(1u << b.s);  // voxel edge at this level
    b.inv_nx = 1.0f / (float)max(b.nx, 1u), b.inv_ny = 1.0f / (float)max(b.ny, 1u);
    return b;
}

// lane `lane` resolves voxel number cb+lane of the box: occupied range [start, start+cnt)
__device__ __forceinline__ void lookup_voxel(const GridView& g, const PassBox& b,
                                             unsigned long long cid, float qlx, float qly,
                                             float qlz, float qhx, float qhy, float qhz,
                                             float prune2, uint32_t& start, uint32_t& cnt,
                                             float& md2)
{
    start = 0, cnt = 0, md2 = INFINITY;
    if (cid >= b.ncell) return;
    uint32_t ix, iy, iz;
    if (b.ncell <= 65536ull)
    {
        // exact for these sizes: (c + 0.5) / n is at least 0.5/n away from an integer
        const uint32_t c32 = (uint32_t)cid;
        const uint32_t row = (uint32_t)(((float)c32 + 0.5f) * b.inv_nx);
        ix = c32 - row * b.nx;
        iz = (uint32_t)(((float)row + 0.5f) * b.inv_ny);
        iy = row - iz * b.ny;
    }
    else if (b.ncell <= 0xFFFFFFFFull)
    {
        const uint32_t c32 = (uint32_t)cid, row = c32 / b.nx;
        ix = c32 - row * b.nx, iz = row / b.ny, iy = row - iz * b.ny;
    }
    else
    {
        ix = (uint32_t)(cid % b.nx), iy = (uint32_t)((cid / b.nx) % b.ny);
        iz = (uint32_t)(cid / ((unsigned long long)b.nx * b.ny));
    }
    const uint32_t cx = b.cx0 + ix, cy = b.cy0 + iy, cz = b.cz0 + iz;
    // voxel box vs bounding box of the queries served by this pass
    const float vx0 = g.ox + (float)cx * b.hs, vy0 = g.oy + (float)cy * b.hs,
                vz0 = g.oz + (float)cz * b.hs;
    const float dx = fmaxf(0.f, fmaxf(vx0 - qhx, qlx - (vx0 + b.hs)));
    const float dy = fmaxf(0.f, fmaxf(vy0 - qhy, qly - (vy0 + b.hs)));
    const float dz = fmaxf(0.f, fmaxf(vz0 - qhz, qlz - (vz0 + b.hs)));
    md2 = dx * dx + dy * dy + dz * dz;
    if (md2 <= prune2)
    {
        uint32_t e = 0;
        if (voxel_range(g, b.lev, cx, cy, cz, start, e, false)) cnt = e - start;
        else start = 0;
    }
}

// the same for a voxel given by its coordinates (the brick path of the tile kernel: the voxel is known to be occupied)
__device__ __forceinline__ void resolve_voxel(const GridView& g, const PassBox& b, uint32_t cx, uint32_t cy, uint32_t cz,
                                              float qlx, float qly, float qlz, float qhx, float qhy, float qhz,
                                              float prune2, uint32_t& start, uint32_t& cnt)
{
    start = 0, cnt = 0;
    const float vx0 = g.ox + (float)cx * b.hs, vy0 = g.oy + (float)cy * b.hs, vz0 = g.oz + (float)cz * b.hs;
    const float dx = fmaxf(0.f, fmaxf(vx0 - qhx, qlx - (vx0 + b.hs)));
    const float dy = fmaxf(0.f, fmaxf(vy0 - qhy, qly - (vy0 + b.hs)));
    const float dz = fmaxf(0.f, fmaxf(vz0 - qhz, qlz - (vz0 + b.hs)));
    if (dx * dx + dy * dy + dz * dz <= prune2)
    {
        uint32_t e = 0;
        if (voxel_range(g, b.lev, cx, cy, cz, start, e, true)) cnt = e - start;
        else start = 0;
    }
}

// which voxel of the current batch holds candidate number gt (s_coff = exclusive offsets)
__device__ __forceinline__ uint32_t locate_candidate(const uint32_t* s_cstart,
                                                     const uint32_t* s_coff, uint32_t gt)
{
    int lo = 0, hi = 63;
#pragma unroll
    for (int it = 0; it < 6; it++)
    {
        const int mid = (lo + hi + 1) >> 1;
        if (s_coff[mid] <= gt) lo = mid;
        else hi = mid - 1;
    }
    return s_cstart[lo] + (gt - s_coff[lo]);
}

// a voxel whose nearest corner is farther than this cannot hold the answer or a tie of it
// (fp32 slack on the voxel box included)
__device__ __forceinline__ float voxel_limit(float bound, float slack)
{
    return bound * 1.000001f + slack * (2.f * sqrtf(bound) + slack);
}

// next radius of an unresolved query (grows strictly; capped at r_max)
__device__ __forceinline__ float next_radius(float r, float rmax, float best_d2, bool have,
                                             float slack)
{
    // the radius certain to conclude (the best candidate so far), but never more than doubling: a
    // far candidate (a stale warm start) must not blow the box up
    const float rn = have ? fminf(sqrtf(best_d2) * (1.0f + 1.0f / 512.0f) + 4.f * slack, 2.0f * r) : 2.0f * r;
    return fminf(fmaxf(rn, r * 1.0009765625f), rmax);
}
// the visited voxels cover the whole cube of half-edge r around the query
__device__ __forceinline__ bool is_final(float r, float rmax, float best_d2, float slack)
{
    const float gr = r * (1.0f - 1.0f / 1024.0f) - slack;
    return r >= rmax || (gr > 0.f && best_d2 < gr * gr);
}

// one claim: a plain look first (a value of this epoch that is already lower makes the atomic
// pointless; a stale value can only be higher, so nothing is skipped wrongly), then the atomic
__device__ __forceinline__ void claim_global(const NNArgs& a, uint32_t spos, uint32_t rank32)
{
    const unsigned long long val = a.claim_hi | (unsigned long long)rank32;
    if (a.claim_peek && a.claims[spos] <= val) return;
    atomicMin(&a.claims[spos], val);
}

// Result records + claims of one wave (ONE wave per workgroup: the barriers are wave-local).
// do_emit: this lane writes the record of query qi.  s_claim: NN_CLAIM_SLOTS words of LDS.
__device__ __forceinline__ void emit_wave(const NNArgs& a, unsigned long long* s_claim, int lane, bool do_emit,
                                          uint32_t qi, uint32_t orig, bool active, float thr, float best_d2,
                                          uint32_t best_idx, uint32_t best_spos, float lb2_all, uint32_t cost = 0u)
{
    bool acc = do_emit && active && best_idx != NONE_U32 && best_d2 < thr;  // :259
    if (acc && a.global_taken && a.global_taken[best_idx]) acc = false;     // :98-101
    if (do_emit)
    {
        // next call's warm start: the raw nearest neighbour (even if rejected) and what this search
        // proved: no map point is nearer than lb2_all (min(best, threshold): every point that could pass
        // the threshold was examined; or the bound that let the search be skipped)
        const float lb2 = active ? lb2_all : 0.f;
        a.rec[qi] = make_uint4(best_spos, __float_as_uint(best_d2), __float_as_uint(lb2),
                               (acc ? 1u : 0u) | (min(cost, 0xFFFFFFu) << NN_COST_SHIFT));
    }
    if (!a.claims) return;               // uniform
    if (__ballot(acc) == 0ull) return;   // uniform
    const uint32_t vrank  = (acc && a.rank) ? a.rank[orig] : orig;  // the order the sequential loop visits
    const uint32_t rank32 = (uint32_t)(a.local_offset + vrank);
    if (!a.claim_dedup)
    {
        if (acc) claim_global(a, best_spos, rank32);
        return;
    }
    // the wave's own minimum per distinct global point: the table slot of a global point holds the
    // lowest (spos, rank) key hashed to it; a lane whose global point owns its slot claims only if it
    // is that minimum, a lane whose slot went to another global point claims by itself
    s_claim[lane] = ~0ull, s_claim[lane + 64] = ~0ull;
    __syncthreads();
    const unsigned long long key  = ((unsigned long long)best_spos << 32) | rank32;
    const uint32_t           slot = (best_spos * 0x9E3779B1u) >> 25;  // 7 bits
    if (acc) atomicMin(&s_claim[slot], key);
    __syncthreads();
    if (acc)
    {
        const unsigned long long v = s_claim[slot];
        if ((uint32_t)(v >> 32) != best_spos || v == key) claim_global(a, best_spos, rank32);
    }
    __syncthreads();
}

// push the lanes of `push` (at most one per query) onto segment `seg` of a query list
__device__ __forceinline__ uint32_t push_lanes(const NNArgs& a, int list, uint32_t seg, bool mine,
                                               unsigned long long push, int lane, uint32_t qi, float r,
                                               float best_d2, uint32_t best_idx, uint32_t best_spos,
                                               float qx, float qy, float qz)
{
    uint4* l_rec = list == 1 ? a.work : a.pend + (list == 2 ? a.list_cap : 0u);
    uint4* l_q   = list == 1 ? a.work_q : a.pend_q + (list == 2 ? a.list_cap : 0u);
    const int npush     = __popcll(push);
    uint32_t  base_slot = 0;
    if (lane == 0)
        base_slot = atomicAdd(a.q_counters + ((size_t)list * NN_MAX_SEG + seg) * NN_CNT_STRIDE, (uint32_t)npush);
    base_slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_slot);
    if (mine && ((push >> lane) & 1ull))
    {
        // a segment holds every query of its workgroups: base_slot + rank < seg_cap by construction
        const size_t slot = (size_t)seg * a.seg_cap + base_slot + (uint32_t)__popcll(push & ((1ull << lane) - 1ull));
        l_rec[slot] = make_uint4(qi, __float_as_uint(r), __float_as_uint(best_d2), best_idx);
        l_q[slot]   = make_uint4(__float_as_uint(qx), __float_as_uint(qy), __float_as_uint(qz), best_spos);
    }
    return (uint32_t)npush;
}

// push the lanes of `mask` (one entry per query slot) onto the deferred-query list (the segment the
// tile's own queries came from: their number is bounded by that segment's capacity)
template <int Q>
__device__ __forceinline__ uint32_t defer_lanes(const NNArgs& a, uint32_t seg, bool mine, unsigned long long mask,
                                                int lane, int slice, uint32_t qi, float r,
                                                float best_d2, uint32_t best_idx, uint32_t best_spos,
                                                float qx, float qy, float qz)
{
    const unsigned long long slot_mask = (Q < 64) ? ((1ull << (Q & 63)) - 1ull) : ~0ull;
    return push_lanes(a, 1, seg, mine && slice == 0, mask & slot_mask, lane, qi, r, best_d2, best_idx, best_spos,
                      qx, qy, qz);
}

// ---- per-lane search helpers -------------------------------------------------------------------
// squared distance of the axis-aligned box [v0, v0+h]^3 from the point q
__device__ __forceinline__ float box_dist2(float vx0, float vy0, float vz0, float h, float qx,
                                           float qy, float qz)
{
    const float dx = fmaxf(0.f, fmaxf(vx0 - qx, qx - (vx0 + h)));
    const float dy = fmaxf(0.f, fmaxf(vy0 - qy, qy - (vy0 + h)));
    const float dz = fmaxf(0.f, fmaxf(vz0 - qz, qz - (vz0 + h)));
    return dx * dx + dy * dy + dz * dz;
}

// 4-bit mask of the positions lo..hi (clamped to the brick [b*4, b*4+3]) along one axis
__device__ __forceinline__ uint32_t axis_mask(uint32_t b, uint32_t c0, uint32_t c1)
{
    const uint32_t lo = max(c0, b * 4u) - b * 4u, hi = min(c1, b * 4u + 3u) - b * 4u;
    return ((2u << hi) - 1u) & ~((1u << lo) - 1u);
}
// a 4-bit axis mask spread over the 64 voxels of a brick (bit = z*16 + y*4 + x)
__device__ __forceinline__ unsigned long long spread_x(uint32_t x4)
{
    const uint32_t m = x4 * 0x11111111u;
    return ((unsigned long long)m << 32) | m;
}
__device__ __forceinline__ unsigned long long spread_y(uint32_t y4)
{
    const uint32_t t = (y4 | (y4 << 3) | (y4 << 6) | (y4 << 9)) & 0x1111u;  // bit k -> bit 4k
    const uint32_t m = (t * 0xFu) * 0x00010001u;
    return ((unsigned long long)m << 32) | m;
}
__device__ __forceinline__ unsigned long long spread_z(uint32_t z4)
{
    const uint32_t lo = ((z4 & 1u) ? 0x0000FFFFu : 0u) | ((z4 & 2u) ? 0xFFFF0000u : 0u);
    const uint32_t hi = ((z4 & 4u) ? 0x0000FFFFu : 0u) | ((z4 & 8u) ? 0xFFFF0000u : 0u);
    return ((unsigned long long)hi << 32) | lo;
}

// ================================================================================================
// One query per lane (see the file header).  8 waves per SIMD.
template <bool INSTR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void nn_lane_kernel(const NNArgs a)
{
    __shared__ unsigned long long s_claim[NN_CLAIM_SLOTS];
    const GridView& g    = a.g;
    const int       lane = threadIdx.x;
    const uint32_t  wv   = a.wave_base + blockIdx.x;  // wave of the whole layer
    const uint32_t  qi   = wv * 64u + (uint32_t)lane;
    const bool      valid = qi < a.n_l;

    float4 lp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = a.lpts[qi];
    // the warm-start record is in the same (Morton) order: its load is issued together with the
    // point's, not behind it
    uint4 h = make_uint4(NONE_U32, 0u, 0u, 0u);
    if (valid && a.use_hint) h = a.rec[qi];
    const uint32_t orig = __float_as_uint(lp.w);
    // a visit order on the cloud (maxLocalPointsPerLayer, Matcher_Points_Base.cpp:222-246): only
    // the listed points are transformed, boxed and matched
    bool visited = valid;
    if (a.rank && valid) visited = a.rank[orig] != NONE_U32;

    // ---- K1: transform (fp64 compose, one narrowing) ------------------------------------
    float qx, qy, qz;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);

    // bounding box of ALL transformed local points of the wave (Matcher_Points_Base.cpp:186-196)
    {
        const float bx0 = wave_min_nn((visited && qx == qx) ? qx : INFINITY), by0 = wave_min_nn((visited && qy == qy) ? qy : INFINITY),
                    bz0 = wave_min_nn((visited && qz == qz) ? qz : INFINITY);
        const float bx1 = wave_max_nn((visited && qx == qx) ? qx : -INFINITY), by1 = wave_max_nn((visited && qy == qy) ? qy : -INFINITY),
                    bz1 = wave_max_nn((visited && qz == qz) ? qz : -INFINITY);
        if (lane == 0)
        {
            float* o = a.tile_bbox + (size_t)wv * 6;
            o[0] = bx0, o[1] = by0, o[2] = bz0, o[3] = bx1, o[4] = by1, o[5] = bz1;
        }
    }

    // ---- threshold rule (Matcher_Points_DistanceThreshold.cpp:223-225, 256-259) -----------
    const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
    const float thr    = fadd(a.maxDistSq, fmul(a.angSq, normSq));
    // every point with fl(d2) < thr lies within r_max of the query
    const float rmax = sqrtf(thr) * 1.002f + g.slack;

    bool active = visited && (normSq < INFINITY);  // non-finite query: nothing to pair
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // :218-220

    float    r        = fminf(a.r0, rmax);
    bool     done     = !active;
    float    best_d2  = INFINITY;
    uint32_t best_idx = NONE_U32, best_spos = NONE_U32;

    // ---- warm start from the previous call on the same map and cloud (the previous ICP
    //      iteration).  Two facts survive a pose change: (1) the previous nearest neighbour is
    //      still a map point, so its distance now is an upper bound; (2) every map point was at
    //      least lb away then and this query moved by disp, so every map point is at least
    //      lb - disp away now.  (2) proves radii below that bound useless and lets a query with
    //      nothing within the threshold finish without a search; (1) gives the radius that is
    //      certain to conclude.  The ball that decides the result is still searched completely,
    //      so the result is the cold result. ---------------------------------------------------
    float lb2_out = -1.f;  // >= 0: the record's bound when this kernel concludes without a search
    float lb2nd_out = 0.f; // > 0: the previous neighbour was certified; the bound on all others, as of this pose
    if (a.use_hint && active)
    {
        float       ox, oy, oz;
        compose_point_f(a.prev_pose, lp.x, lp.y, lp.z, ox, oy, oz);
        const float disp = sqrtf(dist2(qx, qy, qz, ox, oy, oz));
        float       lb   = sqrtf(__uint_as_float(h.z)) * 0.99999f - disp * 1.00001f - 4.f * g.slack;
        if (!(lb > 0.f)) lb = 0.f;  // also catches NaN
        float hr = 0.f;
        if (h.x < g.n)
        {
            const float4 hp = g.pts[h.x];
            const float  hd = dist2(qx, qy, qz, hp.x, hp.y, hp.z);
            if (hd < INFINITY)
            {
                best_d2 = hd, best_idx = __float_as_uint(hp.w), best_spos = h.x;
                hr = sqrtf(hd) * (1.0f + 1.0f / 512.0f) + 4.f * g.slack;
            }
        }
        if (lb * 0.999f > sqrtf(thr))
        {  // fl(d2) >= thr for every map point: nothing to pair, nothing to search
            done    = true;
            lb2_out = (lb * 0.9999f) * (lb * 0.9999f);
        }
        else if (lb > r * (1.0f - 1.0f / 1024.0f) - g.slack)
            r = fminf(fmaxf(hr > 0.f ? fminf(hr, 2.0f * lb) : 2.0f * lb, r), rmax);
        // ---- certificate: every OTHER map point was at least l2 away at the previous pose, hence at least l2 - disp now
        //      (the triangle inequality between the two fp32 positions of the query, which both calls compute identically);
        //      if the previous neighbour, re-measured, is nearer than that by a margin far above the rounding of a computed
        //      distance (slack / 2 = 2^-21 of the map's extent; a computed d differs from the true one by 2^-23 relative),
        //      it is the unique nearest neighbour the full search would return -- same point, same fp32 d2 -- and the
        //      search is skipped.  The bound kept for the next call shrinks by the displacement.
        if (a.cert_read && !done && hr > 0.f)
        {
            const float l2   = a.lb2nd[qi];
            const float room = l2 - disp * 1.00001f - 0.25f * g.slack;
            if (l2 > 0.f && sqrtf(best_d2) * 1.00001f + 0.5f * g.slack < room) done = true, lb2nd_out = room;
        }
    }

    // ---- can this lane search its cube [q - r, q + r]^3 by itself?  At most lane_cells level-0
    //      voxels per axis (<= 4: the cube then fits a 4x4x4 mask and touches <= 2 bricks per
    //      axis); a cube that misses the layer's bounding box holds nothing ------------------------
    uint32_t cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0;
    bool     fast = false, empty_cube = false;
    if (!done && a.lane_cells != 0u && g.occ_off[0] != OCC_NONE)
    {
        const float lox = fmaxf(qx - r, g.bbmin[0]), loy = fmaxf(qy - r, g.bbmin[1]), loz = fmaxf(qz - r, g.bbmin[2]);
        const float hix = fminf(qx + r, g.bbmax[0]), hiy = fminf(qy + r, g.bbmax[1]), hiz = fminf(qz + r, g.bbmax[2]);
        empty_cube = (lox > hix) || (loy > hiy) || (loz > hiz);
        if (!empty_cube)
        {
            cx0 = cell_fine(lox, g.ox, g.inv_hf) >> g.shift0, cx1 = cell_fine(hix, g.ox, g.inv_hf) >> g.shift0;
            cy0 = cell_fine(loy, g.oy, g.inv_hf) >> g.shift0, cy1 = cell_fine(hiy, g.oy, g.inv_hf) >> g.shift0;
            cz0 = cell_fine(loz, g.oz, g.inv_hf) >> g.shift0, cz1 = cell_fine(hiz, g.oz, g.inv_hf) >> g.shift0;
            fast = (cx1 - cx0) < a.lane_cells && (cy1 - cy0) < a.lane_cells && (cz1 - cz0) < a.lane_cells;
        }
        else
            fast = true;  // nothing to visit at this radius
    }

    uint32_t st_cand = 0, st_vox = 0;
    if (fast && !empty_cube)
    {
        // ---- occupancy mask of the cube, bit = (z - cz0) * 16 + (y - cy0) * 4 + (x - cx0), from the
        //      <= 2 x 2 x 2 bricks it overlaps (all loads independent) ---------------------------
        const uint32_t bx0 = cx0 >> 2, by0 = cy0 >> 2, bz0 = cz0 >> 2;
        const uint32_t nbx = (cx1 >> 2) - bx0, nby = (cy1 >> 2) - by0, nbz = (cz1 >> 2) - bz0;  // 0 or 1
        const uint32_t obx = g.occ_bx[0], oby = g.occ_by[0], obz = g.occ_bz[0];
        const unsigned long long* occ0 = g.occ + g.occ_off[0];
        unsigned long long word[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const uint32_t dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            const uint32_t Bx = bx0 + dx, By = by0 + dy, Bz = bz0 + dz;
            word[k] = 0ull;
            if (dx <= nbx && dy <= nby && dz <= nbz && Bx < obx && By < oby && Bz < obz)
                word[k] = occ0[((size_t)Bz * oby + By) * obx + Bx];
        }
        const unsigned long long mx[2] = {spread_x(axis_mask(bx0, cx0, cx1)), nbx ? spread_x(axis_mask(bx0 + 1, cx0, cx1)) : 0ull};
        const unsigned long long my[2] = {spread_y(axis_mask(by0, cy0, cy1)), nby ? spread_y(axis_mask(by0 + 1, cy0, cy1)) : 0ull};
        const unsigned long long mz[2] = {spread_z(axis_mask(bz0, cz0, cz1)), nbz ? spread_z(axis_mask(bz0 + 1, cz0, cz1)) : 0ull};
        unsigned long long m = 0ull;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            const unsigned long long w = word[k] & mx[dx] & my[dy] & mz[dz];
            // voxel (x, y, z) of brick B sits at bit + delta of the cube mask
            const int delta = ((int)((bx0 + dx) * 4u) - (int)cx0) + 4 * ((int)((by0 + dy) * 4u) - (int)cy0) +
                              16 * ((int)((bz0 + dz) * 4u) - (int)cz0);
            m |= delta >= 0 ? (w << (delta & 63)) : (w >> ((-delta) & 63));
        }

        // ---- walk the occupied voxels of the cube the ball reaches: probe, then the points, 4 loads
        //      in flight.  One flat loop per lane (refill / work), so lanes with more voxels do not
        //      make the others wait at every nesting level ---------------------------------------
        const float hs     = g.hf * (float)(1u << g.shift0);
        const float prune  = r + 4.f * g.slack;
        const float prune2 = prune * prune;
        uint32_t    p = 0, pe = 0;
        for (;;)
        {
            while (p >= pe && m != 0ull)
            {
                const uint32_t bit = (uint32_t)__ffsll((long long)m) - 1u;
                m &= m - 1ull;
                const uint32_t cx = cx0 + (bit & 3u), cy = cy0 + ((bit >> 2) & 3u), cz = cz0 + (bit >> 4);
                const float    md2 = box_dist2(g.ox + (float)cx * hs, g.oy + (float)cy * hs, g.oz + (float)cz * hs, hs,
                                               qx, qy, qz);
                if (md2 <= fminf(prune2, voxel_limit(best_d2, g.slack)))
                {
                    uint32_t s0 = 0, e0 = 0;
                    if (voxel_range(g, 0u, cx, cy, cz, s0, e0, true)) p = s0, pe = e0;
                    if (INSTR) st_vox++;
                }
            }
            if (p >= pe) break;
            float4 c4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) c4[k] = g.pts[(p + k < pe) ? p + k : p];  // (round 6: four loads in flight; see nn_seltile.hip)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (!(p + k < pe)) c4[k] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const float    d  = dist2(qx, qy, qz, c4[k].x, c4[k].y, c4[k].z);
                const uint32_t ci = __float_as_uint(c4[k].w);
                if (p + k < pe && (d < best_d2 || (d == best_d2 && ci < best_idx)))
                    best_d2 = d, best_idx = ci, best_spos = p + k;
                if (INSTR && p + k < pe) a.touched[p + k] = 1, st_cand++;
            }
            p += 4u;
        }
    }

    // ---- conclude, or hand the query on with its state ------------------------------------------
    bool pending = !done && !fast;
    if (!done && fast)
    {
        if (is_final(r, rmax, best_d2, g.slack)) done = true;
        else
        {
            r       = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
            pending = true;
        }
    }
    const unsigned long long pmask = __ballot(pending);
    {
        // the hard class (served first by the tile kernel's grid): a wide radius, or -- from the previous call on this
        // map and layer -- a tile that staged many candidates (dense neighbourhoods: the tile's duration follows what
        // it stages, whatever the radius)
        // (the radius is all a first call has to go by; on a scene whose pose is half a metre off nearly every radius is
        //  "wide" and the class told nothing -- round 3's hard-first order was no order at all on scene B)
        // By cost the WAVE is classed, not the query (its costliest query decides): the two classes are separate lists,
        // and a wave split between them leaves two lists of fragments -- tiles made of the remains of several waves are
        // spatially loose, stage more and run more passes (measured: 1 382 -> 1 182 it/s with per-query classes).
        const bool               by_cost = a.use_hint && a.hard_cand != 0u;
        const bool               costly  = __ballot(pending && (h.w >> NN_COST_SHIFT) >= a.hard_cand) != 0ull;
        const bool               hard  = pending && (by_cost ? costly : r > a.r_hard);
        const unsigned long long hmask = __ballot(hard), emask = pmask & ~hmask;
        if (hmask)
            push_lanes(a, 0, wv / a.seg_waves, hard, hmask, lane, qi, r, best_d2, best_idx, best_spos, qx, qy,
                       qz);
        if (emask)
            push_lanes(a, 2, wv / a.seg_waves, pending && !hard, emask, lane, qi, r, best_d2, best_idx,
                       best_spos, qx, qy, qz);
    }

    // a query this lane searched itself: all that could pass the threshold was examined
    emit_wave(a, s_claim, lane, valid && !pending, qi, orig, active, thr, best_d2, best_idx, best_spos,
              lb2_out >= 0.f ? lb2_out : fminf(best_d2, thr));
    if (a.lb2nd && valid && !pending) a.lb2nd[qi] = lb2nd_out;

    if (INSTR)
    {
        const uint32_t n_fast = (uint32_t)__popcll(__ballot(fast && !empty_cube));
        const uint32_t n_skip = (uint32_t)__popcll(__ballot(active && (lb2_out >= 0.f || lb2nd_out > 0.f)));
        const uint32_t cand   = wave_sum_u32(st_cand), vox = wave_sum_u32(st_vox);
        if (lane == 0)
        {
            atomicAdd(&a.counters[44], (unsigned long long)n_fast);
            atomicAdd(&a.counters[45], (unsigned long long)cand);
            atomicAdd(&a.counters[46], (unsigned long long)vox);
            atomicAdd(&a.counters[47], (unsigned long long)__popcll(pmask));
            atomicAdd(&a.counters[48], (unsigned long long)n_skip);
        }
    }
}

// ================================================================================================
// Tiles of Q consecutive PENDING queries.
// 5 waves per SIMD: 96 VGPRs with 4 spilled dwords; measured +3.5 % over the compiler's own 108
// VGPRs / 4 waves, while 6 waves (80 VGPRs, 22 spilled dwords) give the gain back
template <int Q, bool INSTR, bool MFMA = false, int WAVES = 5, bool CERT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void nn_tile_kernel(const NNArgs a)
{
    constexpr int S = 64 / Q;
    __shared__ __attribute__((aligned(16))) float s_x[NN_CAP];
    __shared__ __attribute__((aligned(16))) float s_y[NN_CAP];
    __shared__ __attribute__((aligned(16))) float s_z[NN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_idx[NN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_spos[NN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_owner[NN_CAP];
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[64];
    __shared__ uint32_t s_vox[NN_TVLIST];  // wide groups: occupied voxels of the box, 10 bits per axis relative to its corner
    // the claim table is used after the search only: it lives in the staging area (6.5 KB per tile instead
    // of 7.5: 24 tiles per CU instead of 21)
    static_assert(NN_CLAIM_SLOTS * sizeof(unsigned long long) <= NN_CAP * sizeof(float), "claim table fits s_x");
    unsigned long long* s_claim = reinterpret_cast<unsigned long long*>(s_x);

    const GridView& g     = a.g;
    const int       lane  = threadIdx.x;
    const uint32_t  tile  = blockIdx.x;
    // the grid covers the worst case (every query pending): tiles_per_seg tiles for each segment of
    // the pending list; those beyond their segment's count leave at once
    // workgroup b runs on XCD b % 8 (observed placement; each XCD has its own L2): a segment -- ~3 900
    // Morton-consecutive queries, i.e. one neighbourhood of the map -- is served by ONE XCD, and
    // consecutive segments go round the XCDs (fine-grained enough to stay balanced)
    const uint32_t  segs8          = (a.n_seg + 7u) / 8u;               // segments per XCD
    const uint32_t  tiles_per_list = segs8 * 8u * a.tiles_per_seg;
    const uint32_t  cls    = tile >= tiles_per_list ? 1u : 0u;  // 0 = hard (first), 1 = easy
    const uint32_t  tl     = tile - cls * tiles_per_list;
    uint32_t        seg, tk;
    if (a.xcd_map)
    {
        const uint32_t x = tl & 7u, j = tl >> 3;
        const uint32_t sl = j / a.tiles_per_seg;
        tk = j - sl * a.tiles_per_seg, seg = sl * 8u + x;
    }
    else
        seg = tl / a.tiles_per_seg, tk = tl - seg * a.tiles_per_seg;
    if (seg >= a.n_seg) return;
    seg += a.seg_base;  // segment of the whole layer
    const uint32_t  n_pend = a.q_counters[((size_t)(cls ? 2 : 0) * NN_MAX_SEG + seg) * NN_CNT_STRIDE];
    if (tk * (uint32_t)Q >= n_pend) return;
    const unsigned long long tl0 = wall_clock64();
    const uint32_t  cand_cap = cls ? a.tile_cand_cap_easy : a.tile_cand_cap;
    const int       qslot = lane & (Q - 1);
    const int       slice = (Q == 64) ? 0 : lane / Q;
    const bool      valid = tk * Q + qslot < n_pend;
    const size_t    pslot = (size_t)cls * a.list_cap + (size_t)seg * a.seg_cap + tk * Q + qslot;

    // the lane kernel did the per-query set-up (visit list, MatchState, warm start); a pending
    // query arrives with its radius and the best candidate so far
    uint4 w  = make_uint4(0u, 0u, __float_as_uint(INFINITY), NONE_U32);
    uint4 wq = make_uint4(0u, 0u, 0u, NONE_U32);
    if (valid) w = a.pend[pslot], wq = a.pend_q[pslot];
    const uint32_t qi = w.x;
    const uint32_t ws = wq.w;
    // the original index is needed only for the claim at the very end: its load (dependent on the entry)
    // is in flight during the search
    uint32_t orig = 0u;
    if (valid) orig = __float_as_uint(a.lpts[qi].w);
    const float qx = __uint_as_float(wq.x), qy = __uint_as_float(wq.y), qz = __uint_as_float(wq.z);
    const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
    const float thr    = fadd(a.maxDistSq, fmul(a.angSq, normSq));
    const float rmax   = sqrtf(thr) * 1.002f + g.slack;

    const bool active   = valid;
    float      r        = valid ? __uint_as_float(w.y) : 0.f;
    bool       done     = !valid;
    bool       deferred = false;
    float      best_d2  = __uint_as_float(w.z);
    uint32_t   best_idx = w.w, best_spos = ws;

    // a query whose radius already exceeds what a tile should carry goes straight to the
    // one-query-per-wave kernel
    {
        // (with the brick path a wide radius stays -- unless NO candidate is known yet: then nothing bounds the ball but the
        //  radius itself, and the one-query kernel's nearest-voxel-first order is what finds a bound cheaply)
        const bool               wide  = !done && r > a.r_defer && (!a.tile_bricks || best_idx == NONE_U32);
        const unsigned long long wmask = __ballot(wide);
        if (wmask)
        {
            defer_lanes<Q>(a, seg, wide, wmask, lane, slice, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (wide) done = true, deferred = true;
        }
    }

    uint32_t        st_pass = 0, st_cells = 0, st_cand = 0, st_defer = 0;
    const long long t_start = INSTR ? (long long)wall_clock64() : 0;
    // certificate tracking (CERT): the two smallest prefilter values this lane has seen in the running pass, as bit patterns
    // (NNArgs::lb2nd), and the bound of the query once it is final
    constexpr bool track = CERT && MFMA && Q == 32;
    int             t1 = 0x7FFFFFFF, t2 = 0x7FFFFFFF;
    float           lbq = 0.f;
    auto            ins = [&](int x) __attribute__((always_inline)) { t2 = min(t2, max(t1, x)), t1 = min(t1, x); };

    while (true)
    {
        const unsigned long long pend = __ballot(!done);
        if (pend == 0ull) break;

        // ---- this pass serves a GROUP of pending queries: those within grp_factor radii of
        //      the first pending one (and of comparable radius).  A Morton-consecutive tile is
        //      normally one group.  The other pending lanes still test the staged points (every
        //      candidate is a valid upper bound) but only group members may conclude. ----------
        const int   seed = __ffsll((long long)pend) - 1;
        const float sx = readlane_f(qx, seed), sy = readlane_f(qy, seed), sz = readlane_f(qz, seed);
        const float sr = readlane_f(r, seed);
        const float G  = a.grp_factor * sr;
        const bool  grp = !done && fabsf(qx - sx) <= G && fabsf(qy - sy) <= G &&
                         fabsf(qz - sz) <= G && r <= 2.0f * sr;
        const unsigned long long gmask = __ballot(grp);

        // ---- a group of a few isolated queries goes to the one-query-per-wave kernel -----------
        if (__popcll(gmask) <= (int)a.coop_max * S)
        {
            st_defer += defer_lanes<Q>(a, seg, grp, gmask, lane, slice, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (grp) done = true, deferred = true;
            continue;
        }
        st_pass++;
        st_cand += NN_PASS_COST;  // a pass is priced like this many staged candidates (budget and next call's class)
        if (track) t1 = t2 = 0x7FFFFFFF;

        // ---- search box = union of the group's cubes ---------------------------------------
        // (group members are finite: the NaN-free reductions apply)
        const float lox = wave_min_nn(grp ? qx - r : INFINITY), loy = wave_min_nn(grp ? qy - r : INFINITY),
                    loz = wave_min_nn(grp ? qz - r : INFINITY);
        const float hix = wave_max_nn(grp ? qx + r : -INFINITY), hiy = wave_max_nn(grp ? qy + r : -INFINITY),
                    hiz = wave_max_nn(grp ? qz + r : -INFINITY);
        const float rmin_t = wave_min_pos(grp ? r : INFINITY);
        const float rmax_t = wave_max_pos(grp ? r : 0.f);
        // conservative bounding box of the group's queries themselves
        const float qlx = lox + rmin_t, qly = loy + rmin_t, qlz = loz + rmin_t;
        const float qhx = hix - rmin_t, qhy = hiy - rmin_t, qhz = hiz - rmin_t;
        PassBox       box    = choose_level(g, lox, loy, loz, hix, hiy, hiz, a.cell_budget);
        const float   prune  = rmax_t + 4.f * g.slack;
        const float   prune2 = prune * prune;
        // ---- a WIDE group (its level-0 box holds more voxels than a pass resolves one by one): instead of a coarser
        //      level -- whose voxels hold 8x the points, most of them outside every ball -- the level-0 occupancy BRICKS
        //      of the box are read (one u64 per 4x4x4 voxels: empty space costs one load per 64 voxels), the occupied
        //      voxels near the group are listed in LDS and only they are resolved.  This is what the one-query kernel does
        //      for ONE query; a tile of Morton neighbours 1 m from a wall shares the whole list (round 4: such tiles
        //      used to hand all their queries to that kernel, 19 % of the layer on scene B, each fetching its own copy).
        bool     use_bricks = false;
        uint32_t nbx = 0, nby = 0, nb = 0;
        if (a.tile_bricks && box.lev > 0)
        {
            const PassBox b0 = choose_level(g, lox, loy, loz, hix, hiy, hiz, 0xFFFFFFFFu);
            if (b0.lev == 0 && b0.ncell > 0 && b0.nx <= 1024u && b0.ny <= 1024u && b0.nz <= 1024u)
            {
                nbx = ((b0.cx0 + b0.nx - 1u) >> 2) - (b0.cx0 >> 2) + 1u, nby = ((b0.cy0 + b0.ny - 1u) >> 2) - (b0.cy0 >> 2) + 1u;
                const uint32_t nbz = ((b0.cz0 + b0.nz - 1u) >> 2) - (b0.cz0 >> 2) + 1u;
                const unsigned long long nbl = (unsigned long long)nbx * nby * nbz;
                if (nbl <= a.tile_brick_budget) use_bricks = true, nb = (uint32_t)nbl, box = b0;
            }
        }

        const v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        // ---- matrix-pipe prefilter (Q = 32): d2 of 32 staged candidates x 32 queries by three
        //      v_mfma_f32_32x32x2_f32 on coordinates centred on the box:
        //        S = -2 c'.q' + |c'|^2 + |q'|^2,   A = [c'x c'y | c'z |c'|^2 | 1 0],  B = [-2q'x -2q'y | -2q'z 1 | |q'|^2 0]
        //      (rows = candidates, columns = queries: lane l gets column l & 31 and 16 of the 32 rows -- the
        //      (query, slice) lanes of the exact scan).  |S - exact d2| <= tol for every query of the group
        //      (they lie in the box, the candidates in the box grown by one voxel: every term is bounded by
        //      the box), so a candidate with S > best + tol cannot beat or tie the best; the others are
        //      recomputed in the exact FMA-free sequence.  Lanes outside the group only collect upper bounds.
        constexpr bool use_mfma = MFMA && Q == 32;
        const float ocx = 0.5f * (lox + hix), ocy = 0.5f * (loy + hiy), ocz = 0.5f * (loz + hiz);
        float       mtol = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, o0 = 0.f, o1 = 0.f;
        if (use_mfma)
        {
            const float hx = 0.5f * (hix - lox) + box.hs, hy = 0.5f * (hiy - loy) + box.hs, hz = 0.5f * (hiz - loz) + box.hs;
            // the sum's terms are bounded by 4 (hx^2 + hy^2 + hz^2); ~30 roundings of 2^-24 relative each
            mtol = (hx * hx + hy * hy + hz * hz) * (1.0f / 32768.0f);
            const float cqx = qx - ocx, cqy = qy - ocy, cqz = qz - ocz;
            const bool  hi  = lane >= 32;
            b0 = -2.0f * (hi ? cqy : cqx);
            b1 = hi ? 1.0f : -2.0f * cqz;
            b2 = hi ? 0.0f : (cqx * cqx + cqy * cqy + cqz * cqz);
            o0 = hi ? ocy : ocx, o1 = hi ? 0.0f : ocz;
        }
        bool over = false;
        // one batch of <= 64 resolved voxels (lane = voxel: start, cnt): staged in rounds of NN_CAP points and tested
        // against the tile's queries
        auto batch = [&](uint32_t start, uint32_t cnt) __attribute__((always_inline)) {
            const uint32_t incl  = wave_incl_scan(cnt, lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0) return;  // uniform: no occupied voxel in this batch
            const uint32_t off = incl - cnt;
            s_cstart[lane] = start;
            s_coff[lane]   = off;
            st_cand += total;

            for (uint32_t base = 0; base < total && !over; base += NN_CAP)
            {
                const uint32_t m     = min((uint32_t)NN_CAP, total - base);
                over = st_cand - total + base + m > cand_cap;  // (this round is still scanned)
                const uint32_t m_pad = (m + 31u) & ~31u;
                // ---- stage.  Lane l fills slots 4l..4l+3 of the round.  Which voxel a slot
                //      belongs to comes from a segmented broadcast: every occupied voxel drops
                //      its id at its first slot, a prefix-max carries it to the following slots.
                *reinterpret_cast<uint4*>(&s_owner[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
                __syncthreads();
                if (cnt > 0)
                {
                    if (off >= base && off < base + NN_CAP) s_owner[off - base] = (uint32_t)lane + 1u;
                    else if (off < base && off + cnt > base) s_owner[0] = (uint32_t)lane + 1u;
                }
                __syncthreads();
                {
                    const uint4    o4 = *reinterpret_cast<const uint4*>(&s_owner[4 * lane]);
                    const uint32_t p0 = o4.x, p1 = max(p0, o4.y), p2 = max(p1, o4.z), p3 = max(p2, o4.w);
                    const uint32_t in = wave_incl_max(p3, lane);
                    uint32_t       ex = __shfl_up(in, 1, 64);
                    if (lane == 0) ex = 0u;
                    const uint32_t ow[4] = {(uint32_t)max(ex, p0), (uint32_t)max(ex, p1), (uint32_t)max(ex, p2), (uint32_t)max(ex, p3)};
                    const uint32_t t0    = 4u * (uint32_t)lane;
                    uint32_t       src[4];
                    float4         c4[4];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        src[k] = NONE_U32;
                        if (t0 + k < m)
                        {
                            const uint32_t v = ow[k] - 1u;
                            src[k]           = s_cstart[v] + (base + t0 + k - s_coff[v]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        // padding: FAR but finite.  An infinite coordinate makes the prefilter's S an inf - inf = NaN, and the
                        // integer minimum over the accumulators' bit patterns (below) would pick a NaN with the sign bit set
                        // ahead of every real candidate of the block (found by the parity suite: 897 of 904 pairs)
                        // (round 6: the load is unconditional -- an out-of-range slot reads point 0 -- so that all four are in flight
                        //  together: behind `if (ok)` each one got an s_waitcnt vmcnt(0) of its own; nn_seltile.hip)
                        c4[k] = g.pts[(t0 + k < m) ? src[k] : 0u];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (!(t0 + k < m)) c4[k] = make_float4(1e18f, 0.f, 0.f, __uint_as_float(NONE_U32));
                    *reinterpret_cast<float4*>(&s_x[t0]) = make_float4(c4[0].x, c4[1].x, c4[2].x, c4[3].x);
                    *reinterpret_cast<float4*>(&s_y[t0]) = make_float4(c4[0].y, c4[1].y, c4[2].y, c4[3].y);
                    *reinterpret_cast<float4*>(&s_z[t0]) = make_float4(c4[0].z, c4[1].z, c4[2].z, c4[3].z);
                    *reinterpret_cast<uint4*>(&s_idx[t0]) =
                        make_uint4(__float_as_uint(c4[0].w), __float_as_uint(c4[1].w),
                                   __float_as_uint(c4[2].w), __float_as_uint(c4[3].w));
                    *reinterpret_cast<uint4*>(&s_spos[t0]) = make_uint4(src[0], src[1], src[2], src[3]);
                    if (use_mfma)
                    {  // |c - centre|^2 (the lane's own four owner slots are free again: it has read them)
                        float n4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            const float ex = c4[k].x - ocx, ey = c4[k].y - ocy, ez = c4[k].z - ocz;
                            n4[k] = ex * ex + ey * ey + ez * ez;
                        }
                        *reinterpret_cast<float4*>(&s_owner[t0]) = make_float4(n4[0], n4[1], n4[2], n4[3]);
                    }
                    if (INSTR)
                    {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (t0 + k < m) a.touched[src[k]] = 1;
                    }
                }
                __syncthreads();
                if (use_mfma)
                {
                    const float* s_n   = reinterpret_cast<const float*>(s_owner);
                    const bool   hi    = lane >= 32;
                    const float* s_a0  = hi ? s_y : s_x;
                    const float* s_a1  = hi ? s_n : s_z;
                    const float  a2    = hi ? 0.0f : 1.0f;
                    float        lim   = best_d2 * 1.000001f + mtol;
                    for (uint32_t blk = 0; blk < m_pad; blk += 32u)
                    {
                        const uint32_t c   = blk + ((uint32_t)lane & 31u);
                        const float    a0v = s_a0[c] - o0, a1v = s_a1[c] - o1;
                        f32x16         acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
                        // "is any of the 16 within the limit": the minimum of the BIT PATTERNS as signed integers -- eight
                        // v_min3_i32 instead of fifteen fminf, each of which the compiler wraps in two canonicalising
                        // v_max x,x (27 instructions per block, a quarter of the kernel's vector instructions).  Among
                        // non-negative floats the integer order is the float order; a negative S (rounding of a distance
                        // near zero) is a negative integer, hence the minimum, and is below any limit anyway; the padding
                        // slots hold far, FINITE coordinates (a NaN with the sign bit set would win the integer minimum).
                        // ... in four groups of four, so that the recomputation path tests only the rows of a group whose
                        // minimum is within the limit (the sixteen sequential row tests -- a compare, an exec mask and a branch
                        // each -- were most of what a block cost: nearly every block has SOME lane with a hit)
                        int g4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            g4[k] = min(min(__float_as_int(acc[4 * k]), __float_as_int(acc[4 * k + 1])),
                                        min(__float_as_int(acc[4 * k + 2]), __float_as_int(acc[4 * k + 3])));
                        const int   mni = min(min(g4[0], g4[1]), min(g4[2], g4[3]));
                        const float mn  = __int_as_float(mni);
                        // (tracking: a block none of whose values is within the limit cannot hold the nearest neighbour --
                        //  its minimum stands for all of them; in the block that does, a group outside the limit is stood for
                        //  by its minimum and a group within it contributes its four values one by one: the second smallest of
                        //  what was inserted is then a lower bound of the second smallest of ALL values)
                        if (track && !(!done && mn <= lim)) ins(mni);
                        if (!done && mn <= lim)
                        {
#pragma unroll
                            for (int k = 0; k < 4; k++)
                            {
                                if (track && !(__int_as_float(g4[k]) <= lim)) ins(g4[k]);
                                if (__int_as_float(g4[k]) <= lim)
                                {
#pragma unroll
                                    for (int r = 4 * k; r < 4 * k + 4; r++)
                                    {
                                        if (track) ins(__float_as_int(acc[r]));
                                        if (acc[r] <= lim)
                                        {
                                            // row of register r (C/D layout of the 32x32 MFMAs)
                                            const uint32_t j  = blk + (uint32_t)((r & 3) + 8 * (r >> 2)) + (hi ? 4u : 0u);
                                            const float    dd = dist2(qx, qy, qz, s_x[j], s_y[j], s_z[j]);
                                            if (dd <= best_d2)
                                            {
                                                const uint32_t ci = s_idx[j];
                                                if (dd < best_d2 || ci < best_idx)
                                                {
                                                    best_d2   = dd;
                                                    best_idx  = ci;
                                                    best_spos = s_spos[j];
                                                    lim       = best_d2 * 1.000001f + mtol;
                                                }
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                else
                // ---- scan: every lane tests (its slice of) the bucket against its query, 8
                //      candidates per step on the packed-fp32 path; the update is rare
                for (uint32_t jb = (uint32_t)slice * 8u; jb < m_pad; jb += 8u * S)
                {
                    const float4 xa = *reinterpret_cast<const float4*>(&s_x[jb]);
                    const float4 xb = *reinterpret_cast<const float4*>(&s_x[jb + 4]);
                    const float4 ya = *reinterpret_cast<const float4*>(&s_y[jb]);
                    const float4 yb = *reinterpret_cast<const float4*>(&s_y[jb + 4]);
                    const float4 za = *reinterpret_cast<const float4*>(&s_z[jb]);
                    const float4 zb = *reinterpret_cast<const float4*>(&s_z[jb + 4]);
                    const v2f d01 = dist2_pk(qx2, qy2, qz2, v2f{xa.x, xa.y}, v2f{ya.x, ya.y}, v2f{za.x, za.y});
                    const v2f d23 = dist2_pk(qx2, qy2, qz2, v2f{xa.z, xa.w}, v2f{ya.z, ya.w}, v2f{za.z, za.w});
                    const v2f d45 = dist2_pk(qx2, qy2, qz2, v2f{xb.x, xb.y}, v2f{yb.x, yb.y}, v2f{zb.x, zb.y});
                    const v2f d67 = dist2_pk(qx2, qy2, qz2, v2f{xb.z, xb.w}, v2f{yb.z, yb.w}, v2f{zb.z, zb.w});
                    const float d[8] = {d01.x, d01.y, d23.x, d23.y, d45.x, d45.y, d67.x, d67.y};
                    const float mn = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])),
                                           fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
                    if (!done && mn <= best_d2)
                    {
#pragma unroll
                        for (int k = 0; k < 8; k++)
                        {
                            if (d[k] <= best_d2)
                            {
                                const uint32_t ci = s_idx[jb + k];
                                if (d[k] < best_d2 || ci < best_idx)
                                {
                                    best_d2   = d[k];
                                    best_idx  = ci;
                                    best_spos = s_spos[jb + k];
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
        };
        // the tile's budget of staged candidates also ends a pass IN FLIGHT (`over`): a group in a dense place
        // (vegetation within a wide ball: up to 10^5 points) would otherwise run for a millisecond and be the
        // kernel's duration; what it has not finished goes to the one-query kernel with the bounds found so far
        const uint32_t n_outer = use_bricks ? (nb + 63u) / 64u : 1u;
        for (uint32_t ob = 0; ob < n_outer && !over; ob++)
        {
        // one round of 64 bricks (lane = brick): the occupied voxels of the brick that lie in the box
        unsigned long long bm = 0ull, vtotal = box.ncell;
        uint32_t           bpk = 0, bincl = 0;  // bpk: the brick's position in the box's brick grid, 10 bits per axis
        if (use_bricks)
        {
            const uint32_t id = ob * 64u + (uint32_t)lane;
            if (id < nb)
            {
                const uint32_t row = id / nbx, ix = id - row * nbx, iz = row / nby, iy = row - iz * nby;
                const uint32_t Bx = (box.cx0 >> 2) + ix, By = (box.cy0 >> 2) + iy, Bz = (box.cz0 >> 2) + iz;
                bpk = iz << 20 | iy << 10 | ix;
                const float h4 = 4.f * box.hs;
                const float x0 = g.ox + (float)(Bx * 4u) * box.hs, y0 = g.oy + (float)(By * 4u) * box.hs, z0 = g.oz + (float)(Bz * 4u) * box.hs;
                const float dx = fmaxf(0.f, fmaxf(x0 - qhx, qlx - (x0 + h4)));
                const float dy = fmaxf(0.f, fmaxf(y0 - qhy, qly - (y0 + h4)));
                const float dz = fmaxf(0.f, fmaxf(z0 - qhz, qlz - (z0 + h4)));
                if (dx * dx + dy * dy + dz * dz <= prune2 && Bx < g.occ_bx[0] && By < g.occ_by[0] && Bz < g.occ_bz[0])
                {
                    const unsigned long long word = g.occ[(size_t)g.occ_off[0] + ((size_t)Bz * g.occ_by[0] + By) * g.occ_bx[0] + Bx];
                    bm = word & spread_x(axis_mask(Bx, box.cx0, box.cx0 + box.nx - 1u)) & spread_y(axis_mask(By, box.cy0, box.cy0 + box.ny - 1u)) &
                         spread_z(axis_mask(Bz, box.cz0, box.cz0 + box.nz - 1u));
                }
            }
            bincl  = wave_incl_scan((uint32_t)__popcll(bm), lane);
            vtotal = (uint32_t)__builtin_amdgcn_readlane((int)bincl, 63);
            st_cells += min(64u, nb - ob * 64u);
        }
        for (unsigned long long r0 = 0; r0 < vtotal && !over; r0 += use_bricks ? (unsigned long long)NN_TVLIST : vtotal)
        {
        unsigned long long nv = vtotal;
        if (use_bricks)
        {
            uint32_t           rank = bincl - (uint32_t)__popcll(bm);
            unsigned long long mm   = bm;
            // first voxel of the brick, relative to the box's corner (may be negative: the brick grid starts at or before it)
            const int bvx = (int)(((box.cx0 >> 2) + (bpk & 1023u)) * 4u) - (int)box.cx0, bvy = (int)(((box.cy0 >> 2) + ((bpk >> 10) & 1023u)) * 4u) - (int)box.cy0,
                      bvz = (int)(((box.cz0 >> 2) + (bpk >> 20)) * 4u) - (int)box.cz0;
            while (mm)
            {
                const uint32_t bit = (uint32_t)__ffsll((long long)mm) - 1u;
                mm &= mm - 1ull;
                if (rank >= r0 && rank < r0 + NN_TVLIST)
                    s_vox[rank - (uint32_t)r0] = (uint32_t)(bvz + (int)(bit >> 4)) << 20 | (uint32_t)(bvy + (int)((bit >> 2) & 3u)) << 10 |
                                                 (uint32_t)(bvx + (int)(bit & 3u));
                rank++;
            }
            __syncthreads();
            nv = min((unsigned long long)NN_TVLIST, vtotal - r0);
        }
        for (unsigned long long cb = 0; cb < nv && !over; cb += 64)
        {
            uint32_t cnt = 0, start = 0;
            if (use_bricks)
            {
                if (cb + lane < nv)
                {
                    const uint32_t pk = s_vox[(uint32_t)cb + lane];
                    resolve_voxel(g, box, box.cx0 + (pk & 1023u), box.cy0 + ((pk >> 10) & 1023u), box.cz0 + (pk >> 20), qlx, qly,
                                  qlz, qhx, qhy, qhz, prune2, start, cnt);
                }
            }
            else
            {
                float md2_unused;
                lookup_voxel(g, box, cb + lane, qlx, qly, qlz, qhx, qhy, qhz, prune2, start, cnt, md2_unused);
                st_cells += (uint32_t)min((unsigned long long)64, box.ncell - cb);
            }
            batch(start, cnt);
        }
        if (use_bricks) __syncthreads();  // the list is rewritten by the next round
        }
        }

        // ---- merge the S slices of each query slot ------------------------------------------
        if (S > 1)
        {
#pragma unroll
            for (int off = Q; off < 64; off <<= 1)
            {
                const float    od = __shfl_xor(best_d2, off, 64);
                const uint32_t oi = __shfl_xor(best_idx, off, 64);
                const uint32_t os = __shfl_xor(best_spos, off, 64);
                if (od < best_d2 || (od == best_d2 && oi < best_idx))
                    best_d2 = od, best_idx = oi, best_spos = os;
            }
        }

        if (track)
        {  // the two smallest over both slices of the query slot
            const int o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
            t2 = min(max(t1, o1), min(t2, o2)), t1 = min(t1, o1);
        }
        bool too_wide = false;
        if (grp && !over)  // (a pass cut short has not covered its balls: nobody concludes)
        {
            if (track && is_final(r, rmax, best_d2, g.slack))
            {
                // every staged point but the nearest has S >= t2, hence d2 >= t2 - mtol (the prefilter's proven bound; a
                // negative pattern means a second point within rounding of the query: no bound); every point NOT staged
                // lies beyond the radius this pass covered
                const float cover = r * (1.0f - 1.0f / 1024.0f) - g.slack;
                const float s2    = t2 < 0 ? 0.f : (t2 == 0x7FFFFFFF ? INFINITY : fmaxf(__int_as_float(t2) - mtol, 0.f));
                lbq = fmaxf(fminf(sqrtf(s2) * 0.99999f - g.slack, cover), 0.f);
            }
            if (is_final(r, rmax, best_d2, g.slack)) done = true;
            else
            {
                r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
                // a query whose radius outgrows the voxels would drag the shared box with it:
                // it continues alone, with the whole wave on its own candidates
                too_wide = r > a.r_defer && (!a.tile_bricks || best_idx == NONE_U32);
            }
        }
        // a tile that has already staged more than its budget hands ALL its unfinished queries on
        // (a bound on WORK: round 3 also had a wall-clock bound, which made the kernel that finishes a query depend on
        // timing -- gone with the cost-ordered dispatch below): tiles are dispatched in order, a few of them run 5-8x
        // the mean (many passes over small groups), and one of those starting late keeps a nearly empty
        // chip waiting -- measured: the chip is full for the first 45 % of the kernel's span only
        // (the one-query kernel spreads the same work evenly; results do not depend on who finishes a
        // query)
        if (!done && st_cand > cand_cap) too_wide = true;
        const unsigned long long wmask = __ballot(too_wide);
        if (wmask)
        {
            // (the sign of the entry's radius tells the one-query kernel that the query comes from a tile over its budget:
            //  it records a cost that puts the query's tile first again at the next call, whatever it stages alone)
            st_defer += defer_lanes<Q>(a, seg, too_wide, wmask, lane, slice, qi, st_cand > cand_cap ? -r : r, best_d2, best_idx,
                                       best_spos, qx, qy, qz);
            if (too_wide) done = true, deferred = true;
        }
    }

    // ---- output (Morton order of the local layer) + claim of the global point -----------------
    // every point that could pass the threshold was examined: no map point is nearer than min(best, threshold)
    emit_wave(a, s_claim, lane, valid && slice == 0 && !deferred, qi, orig, active, thr, best_d2, best_idx,
              best_spos, fminf(best_d2, thr), st_cand);
    if (a.lb2nd && valid && slice == 0 && !deferred) a.lb2nd[qi] = track ? lbq : 0.f;

    if (a.timeline && lane == 0)
        a.timeline[2 * (size_t)tile] = tl0, a.timeline[2 * (size_t)tile + 1] = wall_clock64();
    if (INSTR && lane == 0)
    {
        atomicAdd(&a.counters[0], 1ull);
        atomicAdd(&a.counters[1], (unsigned long long)st_pass);
        atomicAdd(&a.counters[2], (unsigned long long)st_cells);
        atomicAdd(&a.counters[3], (unsigned long long)(st_cand - NN_PASS_COST * st_pass));
        if (st_pass > 1) atomicAdd(&a.counters[4], 1ull);
        atomicMax(&a.counters[5], (unsigned long long)(st_cand - NN_PASS_COST * st_pass));
        atomicMax(&a.counters[6], (unsigned long long)st_pass);
        const unsigned long long dt = (unsigned long long)((long long)wall_clock64() - t_start);
        atomicAdd(&a.counters[7], dt);
        atomicMax(&a.counters[8], dt);
        atomicAdd(&a.counters[9], (unsigned long long)st_defer);
        int b = 63 - __clzll((long long)(dt | 1ull));  // log2 bins of 100 MHz ticks
        if (b > 23) b = 23;
        atomicAdd(&a.counters[16 + b], 1ull);
    }
}

// ================================================================================================
// One deferred query per wave.  Such a query is far from its tile mates or far from the map: its
// box holds hundreds of voxels, most of them empty.
//
// scan_batch: one batch of <= 64 resolved voxels (lane = voxel: start, cnt, squared distance of
// the voxel box from the query).  The closest voxel first when no bound is known yet (its points
// give one), then every voxel the bound cannot exclude as one flat candidate list, 4 loads in
// flight per lane.  pd/pi/ps = this lane's partial best, bound = wave-uniform upper bound.
template <bool INSTR>
__device__ __forceinline__ void scan_batch(const NNArgs& a, const GridView& g, int lane, float qx,
                                           float qy, float qz, uint32_t start, uint32_t cnt, float md2,
                                           uint32_t* s_cstart, uint32_t* s_coff, float& pd,
                                           uint32_t& pi, uint32_t& ps, float& bound, uint32_t& st_cand)
{
    const unsigned long long occ = __ballot(cnt > 0);
    if (occ == 0ull) return;
    {
        const float kmin = wave_min_pos(cnt > 0 ? md2 : INFINITY);  // squared distances: >= +0
        const float lim  = bound * 1.000001f + g.slack * (2.f * sqrtf(bound) + g.slack);
        if (!(kmin <= lim)) return;  // nothing in this batch can matter
        if (!(bound < INFINITY))
        {
            const int      lc = __ffsll((long long)__ballot(cnt > 0 && md2 == kmin)) - 1;
            const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)start, lc);
            const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)cnt, lc);
            st_cand += cc;
            for (uint32_t j = lane; j < cc; j += 64)
            {
                const uint32_t sa = cs + j;
                const float4   ca = g.pts[sa];
                const float    da = dist2(qx, qy, qz, ca.x, ca.y, ca.z);
                const uint32_t ia = __float_as_uint(ca.w);
                if (da < pd || (da == pd && ia < pi)) pd = da, pi = ia, ps = sa;
                if (INSTR) a.touched[sa] = 1;
            }
            bound = fminf(bound, wave_min_pos(pd));
            if (lane == lc) cnt = 0;  // done
        }
    }
    // conservative: a voxel is skipped only if even its nearest corner is farther than the bound
    // (fp32 slack on the voxel box included); ties must be seen
    const float    lim   = bound * 1.000001f + g.slack * (2.f * sqrtf(bound) + g.slack);
    const uint32_t c2    = (cnt > 0 && md2 <= lim) ? cnt : 0u;
    const uint32_t incl  = wave_incl_scan(c2, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0) return;
    s_cstart[lane] = start;
    s_coff[lane]   = incl - c2;
    __syncthreads();
    st_cand += total;
    for (uint32_t t0 = 0; t0 < total; t0 += 256)
    {
        uint32_t sa[4];
        float4   ca[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint32_t t = t0 + 64u * k + lane;
            sa[k] = (t < total) ? locate_candidate(s_cstart, s_coff, t) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            ca[k] = g.pts[sa[k]];  // (round 6: unconditional -- sa is 0 beyond the list -- so that the four loads are in flight together)
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint32_t t = t0 + 64u * k + lane;
            if (!(t < total)) ca[k] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const uint32_t t  = t0 + 64u * k + lane;
            const float    da = dist2(qx, qy, qz, ca[k].x, ca[k].y, ca[k].z);
            const uint32_t ia = __float_as_uint(ca[k].w);
            if (t < total && (da < pd || (da == pd && ia < pi))) pd = da, pi = ia, ps = sa[k];
            if (INSTR && t < total) a.touched[sa[k]] = 1;
        }
    }
    __syncthreads();
    bound = fminf(bound, wave_min_pos(pd));
}

// A query with NOTHING within r_max (nn_single_kernel): is a cube of half-edge R = 2 r_max (else 1.5 r_max) around it empty?
// A handful of coarse voxels' occupancy bits tell.  Returns the square of a lower bound of the distance to every map point,
// or -1.
__device__ __forceinline__ float empty_room_bound(const GridView& g, int lane, float qx, float qy, float qz, float rmax)
{
    for (float grow = 2.0f; grow >= 1.49f; grow -= 0.5f)
    {
        const float R = rmax * grow;
        uint32_t    lev = 0;
        while (lev + 1 < g.n_levels && g.hf * (float)(1u << (g.shift0 + lev)) < 0.5f * R) lev++;
        if (g.occ_off[lev] == OCC_NONE) return -1.f;
        const uint32_t sh = g.shift0 + lev;
        const float lox = fmaxf(qx - R, g.bbmin[0]), loy = fmaxf(qy - R, g.bbmin[1]), loz = fmaxf(qz - R, g.bbmin[2]);
        const float hix = fminf(qx + R, g.bbmax[0]), hiy = fminf(qy + R, g.bbmax[1]), hiz = fminf(qz + R, g.bbmax[2]);
        bool occupied = false;
        if (!((lox > hix) || (loy > hiy) || (loz > hiz)))  // (a cube off the layer's bounding box holds nothing)
        {
            const uint32_t cx0 = cell_fine(lox, g.ox, g.inv_hf) >> sh, cx1 = cell_fine(hix, g.ox, g.inv_hf) >> sh;
            const uint32_t cy0 = cell_fine(loy, g.oy, g.inv_hf) >> sh, cy1 = cell_fine(hiy, g.oy, g.inv_hf) >> sh;
            const uint32_t cz0 = cell_fine(loz, g.oz, g.inv_hf) >> sh, cz1 = cell_fine(hiz, g.oz, g.inv_hf) >> sh;
            const uint32_t nx = cx1 - cx0 + 1u, ny = cy1 - cy0 + 1u, nz = cz1 - cz0 + 1u, n = nx * ny * nz;
            if (nx > 8u || ny > 8u || nz > 8u) return -1.f;
            for (uint32_t c0 = 0; c0 < n && !occupied; c0 += 64u)
            {
                const uint32_t c = c0 + (uint32_t)lane;
                bool           o = false;
                if (c < n)
                {
                    const uint32_t row = c / nx, ix = c - row * nx, iz = row / ny, iy = row - iz * ny;
                    o = occ_maybe(g, lev, cx0 + ix, cy0 + iy, cz0 + iz);
                }
                occupied = __ballot(o) != 0ull;
            }
        }
        // every map point lies outside the cube [q - R, q + R]^3 (cell_fine is monotone: a coordinate inside the interval
        // maps into the cell range that was tested), hence farther than R
        if (!occupied) return (R * 0.9999f - 4.f * g.slack) * (R * 0.9999f - 4.f * g.slack);
    }
    return -1.f;
}

constexpr int NN_VLIST = 1024;  // occupied voxels listed per round (LDS)

// W = waves per SIMD the register allocation aims at (__launch_bounds__' second argument; 1 = the
// compiler's own choice, 96-98 VGPRs = 5 waves): the kernel is a chain of dependent loads per query, so
// queries in flight per CU is what it is bound by (measured variants: MP2P_HIP_TUNE single_waves)
template <bool INSTR, int W>
__global__ __launch_bounds__(64, W) void nn_single_kernel(const NNArgs a)
{
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[64];
    __shared__ uint32_t s_vox[NN_VLIST];
    __shared__ uint32_t s_segoff[NN_MAX_SEG + 1];
    const GridView& g      = a.g;
    const int       lane   = threadIdx.x;
    const unsigned long long tl0 = a.timeline ? wall_clock64() : 0ull;
    // the deferred list is segmented (see NNArgs): exclusive prefix of the segment counts, so that
    // item k of the whole list is entry k - off[s] of its segment s
    uint32_t n_work = 0;
    {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < NN_MAX_SEG / 64; k++)
        {
            const uint32_t sg  = (uint32_t)(k * 64 + lane);
            const uint32_t c   = sg < a.n_seg ? a.q_counters[((size_t)1 * NN_MAX_SEG + a.seg_base + sg) * NN_CNT_STRIDE] : 0u;
            const uint32_t inc = wave_incl_scan(c, lane);
            s_segoff[sg]       = run + inc - c;
            run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        n_work = run;
        if (lane == 0) s_segoff[NN_MAX_SEG] = run;
        __syncthreads();
    }

    for (uint32_t k_item = blockIdx.x; k_item < n_work; k_item += gridDim.x)
    {
        // segment of this item: the last one whose offset is <= k_item (empty segments share an offset
        // with their successor and are skipped by taking the last)
        int lo = 0, hi = NN_MAX_SEG - 1;
#pragma unroll
        for (int it = 0; it < 8; it++)
        {
            const int mid = (lo + hi + 1) >> 1;
            if (s_segoff[mid] <= k_item) lo = mid;
            else hi = mid - 1;
        }
        const size_t   item = (size_t)(a.seg_base + lo) * a.seg_cap + (k_item - s_segoff[lo]);
        uint32_t qi, orig, best_idx, best_spos;
        float    qx, qy, qz, thr, rmax, r, best_d2;
        bool     search = true, active = true, heavy = false;
        float    lb2_skip = -1.f;
        {
            const uint4 w  = a.work[item];
            const uint4 wq = a.work_q[item];
            qi   = w.x;
            orig = __float_as_uint(a.lpts[qi].w);  // used by the claim at the end only
            qx = __uint_as_float(wq.x), qy = __uint_as_float(wq.y), qz = __uint_as_float(wq.z);
            const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
            thr  = fadd(a.maxDistSq, fmul(a.angSq, normSq));
            rmax = sqrtf(thr) * 1.002f + g.slack;
            r     = fabsf(__uint_as_float(w.y));
            heavy = (w.y >> 31) != 0u;  // handed on by a tile that had spent its budget
            // wave-uniform running best (carried over from the kernel that handed the query on)
            best_d2 = __uint_as_float(w.z), best_idx = w.w, best_spos = wq.w;
        }
        uint32_t st_pass = 0, st_cand = 0, st_cells = 0;
        const long long t_start = INSTR ? (long long)wall_clock64() : 0;
        bool            far_done = false;

        for (; search;)
        {
            st_pass++;
            const float prune  = r + 4.f * g.slack;
            const float prune2 = prune * prune;
            // per-lane partial best over the candidates this lane tests
            float    pd = INFINITY;
            uint32_t pi = NONE_U32, ps = NONE_U32;
            float    bound = best_d2;  // wave-uniform upper bound of the answer (prunes voxels)

            // ---- voxel enumeration through the occupancy bitmaps: one lane = one 4x4x4 brick,
            //      an empty brick dismisses 64 voxels with one 8-byte load; the occupied voxels
            //      are listed in LDS and only they are probed (always successfully) -------------
            float lox = fmaxf(qx - r, g.bbmin[0]), loy = fmaxf(qy - r, g.bbmin[1]), loz = fmaxf(qz - r, g.bbmin[2]);
            float hix = fminf(qx + r, g.bbmax[0]), hiy = fminf(qy + r, g.bbmax[1]), hiz = fminf(qz + r, g.bbmax[2]);
            const bool empty_box = (lox > hix) || (loy > hiy) || (loz > hiz);
            uint32_t   lev = 0, s = g.shift0, cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0;
            uint32_t   nbx = 0, nby = 0, nbz = 0;
            bool       bricks = !empty_box;
            if (bricks)
            {
                const uint32_t flx = cell_fine(lox, g.ox, g.inv_hf), fhx = cell_fine(hix, g.ox, g.inv_hf);
                const uint32_t fly = cell_fine(loy, g.oy, g.inv_hf), fhy = cell_fine(hiy, g.oy, g.inv_hf);
                const uint32_t flz = cell_fine(loz, g.oz, g.inv_hf), fhz = cell_fine(hiz, g.oz, g.inv_hf);
                for (;;)
                {
                    cx0 = flx >> s, cy0 = fly >> s, cz0 = flz >> s;
                    cx1 = fhx >> s, cy1 = fhy >> s, cz1 = fhz >> s;
                    nbx = (cx1 >> 2) - (cx0 >> 2) + 1, nby = (cy1 >> 2) - (cy0 >> 2) + 1,
                    nbz = (cz1 >> 2) - (cz0 >> 2) + 1;
                    if ((unsigned long long)nbx * nby * nbz <= a.brick_budget || lev + 1 >= g.n_levels) break;
                    s++, lev++;
                }
                bricks = g.occ_off[lev] != OCC_NONE && (unsigned long long)nbx * nby * nbz <= 65536ull &&
                         (cx1 - cx0) < 1024u && (cy1 - cy0) < 1024u && (cz1 - cz0) < 1024u;
            }
            if (bricks)
            {
                const float    hs   = g.hf * (float)(1u << s);
                const uint32_t nb   = nbx * nby * nbz;
                const float    inbx = 1.0f / (float)nbx, inby = 1.0f / (float)nby;
                for (uint32_t b0 = 0; b0 < nb; b0 += 64)
                {
                    const uint32_t id = b0 + lane;
                    st_cells += min(64u, nb - b0);
                    unsigned long long m = 0ull;
                    uint32_t Bx = 0, By = 0, Bz = 0;
                    if (id < nb)
                    {
                        // exact for these sizes: (c + 0.5) / n is at least 0.5/n away from an integer
                        const uint32_t row = (uint32_t)(((float)id + 0.5f) * inbx);
                        const uint32_t ix = id - row * nbx, iz = (uint32_t)(((float)row + 0.5f) * inby),
                                       iy = row - iz * nby;
                        Bx = (cx0 >> 2) + ix, By = (cy0 >> 2) + iy, Bz = (cz0 >> 2) + iz;
                        const float lim0 = fminf(prune2, voxel_limit(bound, g.slack));
                        const float bd2  = box_dist2(g.ox + (float)(Bx * 4u) * hs, g.oy + (float)(By * 4u) * hs,
                                                     g.oz + (float)(Bz * 4u) * hs, 4.f * hs, qx, qy, qz);
                        if (bd2 <= lim0 && Bx < g.occ_bx[lev] && By < g.occ_by[lev] && Bz < g.occ_bz[lev])
                        {
                            const unsigned long long word =
                                g.occ[(size_t)g.occ_off[lev] + ((size_t)Bz * g.occ_by[lev] + By) * g.occ_bx[lev] + Bx];
                            const unsigned long long mx = axis_mask(Bx, cx0, cx1) * 0x1111111111111111ull;
                            const uint32_t           y4 = axis_mask(By, cy0, cy1), z4 = axis_mask(Bz, cz0, cz1);
                            const unsigned long long my16 = ((y4 & 1u) ? 0x000Full : 0) | ((y4 & 2u) ? 0x00F0ull : 0) |
                                                            ((y4 & 4u) ? 0x0F00ull : 0) | ((y4 & 8u) ? 0xF000ull : 0);
                            const unsigned long long my = my16 * 0x0001000100010001ull;
                            const unsigned long long mz = ((z4 & 1u) ? 0x000000000000FFFFull : 0) |
                                                          ((z4 & 2u) ? 0x00000000FFFF0000ull : 0) |
                                                          ((z4 & 4u) ? 0x0000FFFF00000000ull : 0) |
                                                          ((z4 & 8u) ? 0xFFFF000000000000ull : 0);
                            m = word & mx & my & mz;
                        }
                    }
                    const uint32_t c     = (uint32_t)__popcll(m);
                    const uint32_t incl  = wave_incl_scan(c, lane);
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    for (uint32_t r0 = 0; r0 < total; r0 += NN_VLIST)
                    {
                        uint32_t           rank = incl - c;
                        unsigned long long mm   = m;
                        while (mm)
                        {
                            const uint32_t bit = (uint32_t)__ffsll((long long)mm) - 1u;
                            mm &= mm - 1ull;
                            if (rank >= r0 && rank < r0 + NN_VLIST)
                                s_vox[rank - r0] = ((Bz * 4u + (bit >> 4)) - cz0) << 20 |
                                                   ((By * 4u + ((bit >> 2) & 3u)) - cy0) << 10 |
                                                   ((Bx * 4u + (bit & 3u)) - cx0);
                            rank++;
                        }
                        __syncthreads();
                        const uint32_t nlist = min((uint32_t)NN_VLIST, total - r0);
                        for (uint32_t v0 = 0; v0 < nlist; v0 += 64)
                        {
                            uint32_t cnt = 0, start = 0;
                            float    md2 = INFINITY;
                            if (v0 + lane < nlist)
                            {
                                const uint32_t pk = s_vox[v0 + lane];
                                const uint32_t cx = cx0 + (pk & 1023u), cy = cy0 + ((pk >> 10) & 1023u),
                                               cz = cz0 + (pk >> 20);
                                md2 = box_dist2(g.ox + (float)cx * hs, g.oy + (float)cy * hs, g.oz + (float)cz * hs,
                                                hs, qx, qy, qz);
                                if (md2 <= fminf(prune2, voxel_limit(bound, g.slack)))
                                {
                                    uint32_t e = 0;
                                    if (voxel_range(g, lev, cx, cy, cz, start, e, true)) cnt = e - start;
                                    else start = 0;
                                }
                            }
                            scan_batch<INSTR>(a, g, lane, qx, qy, qz, start, cnt, md2, s_cstart, s_coff, pd, pi,
                                              ps, bound, st_cand);
                        }
                        __syncthreads();
                    }
                }
            }
            else
            {
                // no bitmap at a usable level: every voxel of the box is probed
                const PassBox box = choose_level(g, qx - r, qy - r, qz - r, qx + r, qy + r, qz + r,
                                                 a.cell_budget);
                for (unsigned long long cb = 0; cb < box.ncell; cb += 64)
                {
                    uint32_t cnt, start;
                    float    md2;
                    lookup_voxel(g, box, cb + lane, qx, qy, qz, qx, qy, qz, prune2, start, cnt, md2);
                    st_cells += (uint32_t)min((unsigned long long)64, box.ncell - cb);
                    scan_batch<INSTR>(a, g, lane, qx, qy, qz, start, cnt, md2, s_cstart, s_coff, pd, pi, ps, bound,
                                      st_cand);
                }
            }
            wave_argmin(pd, pi, ps);
            if (pd < best_d2 || (pd == best_d2 && pi < best_idx)) best_d2 = pd, best_idx = pi, best_spos = ps;
            if (is_final(r, rmax, best_d2, g.slack))
            {
                // round 6 (NNArgs::far_pass): nothing within the threshold, and no empty room around the query either -- ONE wider
                // pass (1.5 r_max) finds its true nearest distance, or that nothing lies within that ball: the bound the record
                // then carries lets the warm start skip the query until it has moved by the difference, instead of searching the
                // r_max ball again at every call (C5: 300 000 outliers per call)
                if (a.far_pass && !far_done && !(best_d2 < thr) && r < 1.5f * rmax)
                {
                    far_done = true;
                    if (best_idx == NONE_U32 && a.empty_room) lb2_skip = empty_room_bound(g, lane, qx, qy, qz, rmax);
                    if (lb2_skip < 0.f)
                    {
                        r = 1.5f * rmax;
                        continue;
                    }
                }
                break;
            }
            r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
        }
        if (far_done && lb2_skip < 0.f)
        {   // the ball of radius cover was searched completely: every map point is at least min(nearest found, cover) away
            const float cover = 1.5f * rmax * (1.0f - 1.0f / 1024.0f) - g.slack;
            lb2_skip = fminf(best_d2, cover * cover);
        }
        // ---- NOTHING within reach (an outlier of the local layer, metres from every surface).  The record's bound would be
        //      the radius just covered, so the next call -- any displacement at all -- would search the whole ball again,
        //      and the next (round 3: 3 of the 8 ms of configuration C5 were such searches).  Room is cheap for such a
        //      query: a cube of half-edge R = 1.5 r_max .. 2 r_max around it is a handful of COARSE voxels whose occupancy
        //      bits tell whether it is empty; if so every map point is farther than R and the warm start lets the query
        //      skip its search until it has moved by R - r_max.
        // ---- NOTHING within reach (an outlier of the local layer, metres from every surface).  The record's bound would be
        //      the radius just covered, so the next call -- any displacement at all -- would search the whole ball again,
        //      and the next (round 3: 3 of the 8 ms of configuration C5 were such searches).  Room is cheap for such a
        //      query: if the cube of half-edge 2 r_max (else 1.5 r_max) around it is empty, every map point is farther
        //      than that and the warm start lets the query skip its search until it has moved by the difference.
        if (search && !far_done && best_idx == NONE_U32 && a.empty_room) lb2_skip = empty_room_bound(g, lane, qx, qy, qz, rmax);
        if (lane == 0)
        {
            bool acc = active && best_idx != NONE_U32 && best_d2 < thr;         // :259
            if (acc && a.global_taken && a.global_taken[best_idx]) acc = false;  // :98-101
            const float lb2 = !active ? 0.f : (lb2_skip >= 0.f ? lb2_skip : fminf(best_d2, thr));
            // (cost: what this query alone staged -- it was handed on for being isolated, or by a tile over its budget)
            a.rec[qi] = make_uint4(best_spos, __float_as_uint(best_d2), __float_as_uint(lb2),
                                   (acc ? 1u : 0u) | ((heavy ? 0xFFFFFFu : min(st_cand, 0xFFFFFFu)) << NN_COST_SHIFT));
            if (acc && a.claims)
                claim_global(a, best_spos, (uint32_t)(a.local_offset + (a.rank ? a.rank[orig] : orig)));
            if (a.lb2nd) a.lb2nd[qi] = 0.f;  // (no certificate from this kernel)
        }
        if (INSTR && lane == 0)
        {
            atomicAdd(&a.counters[10], 1ull);
            atomicAdd(&a.counters[11], (unsigned long long)st_pass);
            atomicAdd(&a.counters[12], (unsigned long long)st_cells);
            atomicAdd(&a.counters[13], (unsigned long long)st_cand);
            atomicMax(&a.counters[14], (unsigned long long)st_cand);
            const unsigned long long dt = (unsigned long long)((long long)wall_clock64() - t_start);
            atomicAdd(&a.counters[40], dt);
            atomicMax(&a.counters[15], dt);
            atomicMax(&a.counters[41], (unsigned long long)st_pass);
            atomicMax(&a.counters[42], (unsigned long long)st_cells);
        }
    }
    if (a.timeline && lane == 0)
    {
        const size_t k = (size_t)a.timeline_single_base + blockIdx.x;
        a.timeline[2 * k] = tl0, a.timeline[2 * k + 1] = wall_clock64();
    }
}

}  // namespace mp2p
// round 5: the tile kernel with matrix-pipe voxel selection and the fused prologue (the default search path)
#include "nn_seltile.hip"
namespace mp2p
{
// resets the segment counters of the two query lists
__global__ __launch_bounds__(NN_MAX_SEG) void nn_reset_kernel(uint32_t* q_counters)
{
    q_counters[((size_t)blockIdx.x * NN_MAX_SEG + threadIdx.x) * NN_CNT_STRIDE] = 0u;  // one block per list
}

// the [n_l][1] result arrays the other matchers' kernels read, from the packed records
__global__ __launch_bounds__(256) void nn_unpack_rec_kernel(const uint4* __restrict__ rec, uint32_t n,
                                                            uint32_t* __restrict__ spos, float* __restrict__ d2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = rec[i];
    spos[i] = (r.w & 1u) ? r.x : NONE_U32;
    d2[i]   = __uint_as_float(r.y);
}
int launch_unpack_rec(mp2p_hip_ctx* ctx, size_t n_l)
{
    MP2P_TRY_HIP(ctx, ctx->nn_spos.ensure(n_l ? n_l : 1));
    MP2P_TRY_HIP(ctx, ctx->nn_d2.ensure(n_l ? n_l : 1));
    if (n_l)
        hipLaunchKernelGGL(nn_unpack_rec_kernel, dim3((uint32_t)((n_l + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->nn_rec.p, (uint32_t)n_l, ctx->nn_spos.p, ctx->nn_d2.p);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

// reduce the per-tile boxes to the layer box {min xyz, max xyz}: [n_in][6] -> [gridDim.x][6]
__global__ __launch_bounds__(256) void tile_bbox_reduce_kernel(const float* __restrict__ tb,
                                                               uint32_t n_in,
                                                               float* __restrict__ out)
{
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += gridDim.x * blockDim.x)
    {
        const float* p = tb + (size_t)i * 6;
        for (int d = 0; d < 3; d++) v[d] = fminf(v[d], p[d]), v[3 + d] = fmaxf(v[3 + d], p[3 + d]);
    }
    __shared__ float s[4][6];
    const int        lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = 0; d < 3; d++) v[d] = wave_min(v[d]), v[3 + d] = wave_max(v[3 + d]);
    if (lane == 0)
        for (int d = 0; d < 6; d++) s[w][d] = v[d];
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const int d = threadIdx.x;
        float     r = s[0][d];
        for (int k = 1; k < 4; k++) r = d < 3 ? fminf(r, s[k][d]) : fmaxf(r, s[k][d]);
        out[(size_t)blockIdx.x * 6 + d] = r;
    }
}

int launch_bbox_reduce(mp2p_hip_ctx* ctx, uint32_t n_tiles)
{
    constexpr uint32_t NB = 64;
    MP2P_TRY_HIP(ctx, ctx->tile_bbox2.ensure(NB * 6));
    hipLaunchKernelGGL(tile_bbox_reduce_kernel, dim3(NB), dim3(256), 0, ctx->stream,
                       ctx->tile_bbox.p, n_tiles, ctx->tile_bbox2.p);
    hipLaunchKernelGGL(tile_bbox_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream,
                       ctx->tile_bbox2.p, NB, ctx->local_bbox.p);
    return MP2P_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// reduce_bbox: leave the layer's bounding box in ctx->local_bbox (two small launches); false when
// the fused compaction of pairs.hip follows and reduces the per-wave boxes itself
int launch_nn_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                    const double pose[12], const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms,
                    bool reduce_bbox = true)
{
    const size_t n_l = cloud->n;
    uint32_t     Q   = prm->queries_per_wave ? prm->queries_per_wave : 32;
    MP2P_REQUIRE(ctx, Q == 64 || Q == 32 || Q == 16, "queries_per_wave must be 64, 32 or 16");
    // round 6: every tile is 32 queries -- the shape of the matrix pipe's prefilter.  The 16- and 64-query tiles and the exact
    // (non-matrix) scan of rounds 1-3 computed the same lists more slowly (2 202 / 2 390 / 1 945 it/s for 16 / 32 / 64)
    // and cost eight instantiations of nn_tile_kernel, two of them with 92 bytes of scratch per lane; the parameter is still
    // accepted (a tuning value: results never depended on it), as are the knobs mfma_scan and tile_waves = 5 / 6, without effect
    Q = 32;
    const uint32_t n_waves = (uint32_t)((n_l + 63) / 64);     // lane kernel

    // round 5 (nn_seltile.hip): voxels selected on the matrix pipe; needs the level-0 occupancy bricks and the 32-query tile.
    // DIRECT: the per-query prologue runs in the tile itself (no lane kernel, no pending list)
    const bool sel    = ctx->tune.tile_select && map->view.occ != nullptr && map->view.occ_off[0] != OCC_NONE;
    // a SMALL layer (at most two rounds of resident tiles: 2 x 4 096 x 32 queries) is one wave of tiles -- the kernel is as
    // long as its longest tile, and what a long tile hands on is spread over the idle CUs by the one-query kernel: the budgets of
    // rounds 1-4, and the prologue in the tile (no hard-first order to lose, one launch less).  Measured on configuration C2
    // (120 k x 2 M): 3 390 it/s against 3 030 with the large-layer policy, 3 280 with round 4's kernels.
    const bool small_layer = n_l <= 262144;
    const bool direct = sel && (ctx->tune.nn_direct < 0 ? small_layer : ctx->tune.nn_direct != 0);
    // speed-of-light decomposition (timing only: no results, no state): a run-time request of a profiling session (mp2p_hip_set_tune
    // refuses it without profiling, the environment variable cannot set it); the matcher call then appends NOTHING to the caller's list
    const int  sol    = (direct && ctx->profiling != 0 && ctx->tune.tile_sol >= 1 && ctx->tune.tile_sol <= 5) ? ctx->tune.tile_sol : 0;
    const uint32_t n_boxes = direct ? (uint32_t)((n_l + 31) / 32) : n_waves;  // per-tile / per-wave bounding boxes
    MP2P_TRY_HIP(ctx, ctx->nn_rec.ensure(n_l));
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)std::max(n_boxes, 1u) * 6));
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    // query lists in segments (NNArgs): seg_waves consecutive workgroups of the lane kernel share one
    const uint32_t seg_waves = std::max<uint32_t>(1u, (n_waves + NN_MAX_SEG - 1) / NN_MAX_SEG);
    const uint32_t n_seg     = std::max<uint32_t>(1u, (n_waves + seg_waves - 1) / seg_waves);
    const uint32_t seg_cap   = seg_waves * 64u;
    const size_t   list_cap  = (size_t)n_seg * seg_cap;
    MP2P_TRY_HIP(ctx, ctx->work.ensure(list_cap));
    MP2P_TRY_HIP(ctx, ctx->work_q.ensure(list_cap));
    MP2P_TRY_HIP(ctx, ctx->pend.ensure(2 * list_cap));  // hard class, then easy class
    MP2P_TRY_HIP(ctx, ctx->pend_q.ensure(2 * list_cap));
    if (ctx->q_counters.n < (size_t)NN_ALL_LISTS * NN_MAX_SEG * NN_CNT_STRIDE)
    {
        MP2P_TRY_HIP(ctx, ctx->q_counters.ensure((size_t)NN_ALL_LISTS * NN_MAX_SEG * NN_CNT_STRIDE));
        ctx->q_counters_clean = false;
    }
    ctx->last_q       = Q;
    ctx->sol_no_records = false;

    NNArgs a;
    memset(&a, 0, sizeof(a));
    a.g    = map->view;
    a.lpts = cloud->sorted.p;
    a.n_l  = (uint32_t)n_l;
    for (int i = 0; i < 9; i++) a.pose.r[i] = pose[i];
    for (int i = 0; i < 3; i++) a.pose.t[i] = pose[9 + i];
    // mrpt::square(double) narrowed to float (Matcher_Points_DistanceThreshold.cpp:82-83)
    a.maxDistSq         = (float)(prm->threshold * prm->threshold);
    const double angRad = prm->thresholdAngularDeg * 3.14159265358979323846 / 180.0;
    a.angSq             = (float)(angRad * angRad);
    const float cell0   = map->view.hf * (float)(1u << map->view.shift0);
    a.r0          = cell0 * (prm->initial_radius_cells > 0 ? prm->initial_radius_cells : 1.0f);
    a.grp_factor  = prm->group_radius_factor > 0 ? prm->group_radius_factor : 2.5f;
    a.cell_budget = prm->cell_budget > 0 ? prm->cell_budget : 512u;
    a.brick_budget = prm->brick_budget > 0 ? prm->brick_budget : 128u;
    // Deferral radius.  Measured on the street scene at two map densities (voxel 0.25 and 0.5 m) and
    // thresholds 1, 2 and 4 m: the optimum sits near 1 m in every case, i.e. 4 voxels of the dense
    // map and 2 of the sparse one (-8..-16 % at 3 / 4 voxels respectively).  Default: 1 m, kept
    // between 2 and 4 voxels so that it still scales with maps of another size.
    a.r_defer = prm->defer_radius_cells > 0 ? cell0 * prm->defer_radius_cells
                                             : fminf(fmaxf(1.0f, 2.0f * cell0), 4.0f * cell0);
    a.lane_cells    = std::min<uint32_t>(ctx->tune.lane_cells, 4u);
    a.tile_cand_cap = ctx->tune.tile_cand_cap ? ctx->tune.tile_cand_cap : ((sel && !small_layer) ? 24576u : 6144u);
    a.tile_cand_cap_easy = (ctx->tune.hard_cand && ctx->tune.tile_cand_cap_easy) ? ctx->tune.tile_cand_cap_easy : a.tile_cand_cap;
    a.tile_bricks       = (ctx->tune.tile_bricks && map->view.occ != nullptr && map->view.occ_off[0] != OCC_NONE) ? 1 : 0;
    a.tile_brick_budget = ctx->tune.tile_brick_budget;
    a.hard_cand         = ctx->tune.hard_cand;
    a.coop_max          = ctx->tune.coop_max != 0xFFFFFFFFu ? ctx->tune.coop_max : ((sel && !small_layer) ? 0u : 4u);
    a.empty_room        = ctx->tune.empty_room;
    a.far_pass          = ctx->tune.far_pass;
    a.claim_dedup   = ctx->tune.claim_dedup;
    a.claim_peek    = ctx->tune.claim_peek;
    a.mfma_scan     = 1;
    a.local_taken =
        (ms && !prm->allowMatchAlreadyMatchedPoints) ? ms->local_taken.p : nullptr;
    a.global_taken =
        (ms && !prm->allowMatchAlreadyMatchedGlobalPoints) ? ms->global_taken.p : nullptr;
    a.claims = prm->allowMatchAlreadyMatchedGlobalPoints ? nullptr : map->claims.p;
    ctx->epoch++;
    a.claim_hi     = (~(unsigned long long)ctx->epoch) << 32;
    a.local_offset = prm->local_index_offset;
    a.rank         = cloud->n_visit ? cloud->rank.p : nullptr;
    a.rec          = ctx->nn_rec.p;
    a.tile_bbox    = ctx->tile_bbox.p;
    a.work         = ctx->work.p;
    a.work_q       = ctx->work_q.p;
    a.pend         = ctx->pend.p;
    a.pend_q       = ctx->pend_q.p;
    a.q_counters   = ctx->q_counters.p;
    a.n_seg = n_seg, a.seg_waves = seg_waves, a.seg_cap = seg_cap, a.tiles_per_seg = seg_cap / Q;
    a.list_cap = (uint32_t)list_cap;
    a.r_hard   = cell0 * 0.01f * (float)ctx->tune.hard_radius_pct;
    a.xcd_map  = ctx->tune.xcd_map;
    a.direct   = direct ? 1 : 0;
    a.grp_all_bricks = (float)ctx->tune.grp_all_bricks;
    // worst case for each class: every query in it (DIRECT: one class, every tile a fixed slice of the layer)
    const uint32_t n_tiles = (direct ? 1u : 2u) * ((n_seg + 7u) / 8u) * 8u * a.tiles_per_seg;
    ctx->last_n_tiles = n_tiles;
    for (int i = 0; i < 9; i++) a.prev_pose.r[i] = ctx->hint_pose[i];
    for (int i = 0; i < 3; i++) a.prev_pose.t[i] = ctx->hint_pose[9 + i];
    a.use_hint = (ctx->hint_map == map && ctx->hint_cloud == cloud && ctx->hint_n == n_l && !prm->disable_warm_start) ? 1 : 0;
    // ---- search-skip certificate (NNArgs::lb2nd).  The bounds are tracked by a build of the tile kernel that costs ~25 more
    //      instructions per block, so it runs only when the step from the previous call is small enough for a certificate to
    //      have a chance at the next one (nn_cert = 1: the displacement of the farthest local point below nn_cert_step_mm;
    //      2: always; 0: never); the lane kernel reads the bounds whenever the previous call left them.
    bool cert_track = false;
    {
        const bool cert_read = a.use_hint && ctx->nn_lb2nd_valid && ctx->tune.nn_cert != 0;
        if (ctx->tune.nn_cert == 2) cert_track = true;
        else if (ctx->tune.nn_cert == 1 && a.use_hint)
        {
            // |T p - T' p| <= |t - t'| + |R - R'|_F |p| for every local point p
            double dt = 0, dr = 0;
            for (int i = 0; i < 3; i++) dt += (pose[9 + i] - ctx->hint_pose[9 + i]) * (pose[9 + i] - ctx->hint_pose[9 + i]);
            for (int i = 0; i < 9; i++) dr += (pose[i] - ctx->hint_pose[i]) * (pose[i] - ctx->hint_pose[i]);
            cert_track = std::sqrt(dt) + std::sqrt(dr) * (double)cloud->radius <= 1e-3 * (double)ctx->tune.nn_cert_step_mm;
        }
        cert_track = cert_track && ctx->profiling != 2 && sol == 0;
        a.lb2nd = nullptr, a.cert_read = 0;
        if (cert_track || cert_read)
        {
            MP2P_TRY_HIP(ctx, ctx->nn_lb2nd.ensure(n_l));
            a.lb2nd = ctx->nn_lb2nd.p, a.cert_read = cert_read ? 1 : 0;
        }
        ctx->nn_lb2nd_valid = false;  // set again once the launches are on the stream
    }
    const void* const keep_hint_map = ctx->hint_map;
    ctx->hint_map = nullptr;  // committed after the launches: an error return in between leaves no warm start
    a.counters     = nullptr;
    a.touched      = nullptr;
    if (ctx->profiling == 2)
    {
        MP2P_TRY_HIP(ctx, ctx->counters.ensure(64));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64 * sizeof(unsigned long long), ctx->stream));
        a.counters = ctx->counters.p;
        MP2P_TRY_HIP(ctx, ctx->pl_slots.ensure(map->n));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->pl_slots.p, 0, map->n, ctx->stream));
        a.touched = ctx->pl_slots.p;
    }
    a.timeline = nullptr, a.timeline_single_base = n_tiles;
    ctx->timeline_tiles = ctx->timeline_singles = 0;
    if (ctx->profiling == 4)
    {
        // (upper bound of the one-query kernel's grid, see below)
        const size_t single_blocks = 256u * (size_t)(ctx->tune.single_blocks_per_cu ? ctx->tune.single_blocks_per_cu : 40u);
        MP2P_TRY_HIP(ctx, ctx->timeline.ensure(2 * (std::max<size_t>(n_tiles, n_waves) + single_blocks)));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->timeline.p, 0, 2 * (std::max<size_t>(n_tiles, n_waves) + single_blocks) * sizeof(unsigned long long),
                                         ctx->stream));
        a.timeline = ctx->timeline.p;
        ctx->timeline_tiles = n_tiles, ctx->timeline_singles = single_blocks;
    }
    ctx->pending_lane = direct ? 0 : 1;
    ctx->last_n_boxes = n_boxes;
    // the list counters: cleared by the previous call's fused compaction, or here
    if (!ctx->q_counters_clean)
        hipLaunchKernelGGL(nn_reset_kernel, dim3(NN_ALL_LISTS), dim3(NN_MAX_SEG), 0, ctx->stream, a.q_counters);
    ctx->q_counters_clean = false;
    // ev[0]..ev[1] brackets exactly the search kernels (the roofline kernels of bench.py)
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (n_tiles)
    {
        const bool instr = a.counters != nullptr;
        // ---- PIPELINES.  lane -> tile -> one-query kernel is a strict chain, and both big kernels end with
        //      a drain (the last round of tiles / queries runs on an emptying chip: 25-30 % of the tile
        //      kernel's span).  The local layer is therefore cut into independent halves (disjoint segments,
        //      records and list ranges; the claim words take atomics from both), each with its own chain on
        //      its own stream: one chain's drain is filled by the other's kernels.  Joined before the
        //      compaction.  Per-stage events (profiling 1), counters (2) and the timeline (4) keep one chain.
        //      Measured (MP2P_HIP_TUNE=pipelines=2): scene B -6 % search time, scene A +4 % (two lane kernels, two
        //      more launches and the fork / join events cost more than its shorter drains give back): off by
        //      default.  Never on the legacy null stream (event record / wait on it crashed the runtime).
        uint32_t P = (ctx->tune.pipelines >= 2 && n_seg >= 32 && (ctx->profiling == 0 || ctx->profiling == 3) &&
                      ctx->stream != nullptr && ctx->stream != hipStreamLegacy) ? 2u : 1u;
        if (P > 1)
        {
            if (!ctx->stream2) MP2P_TRY_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            if (!ctx->ev_fork) MP2P_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            if (!ctx->ev_join) MP2P_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            MP2P_TRY_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        }
        NNArgs     ap[2];
        uint32_t   wn[2], tn[2], sbn[2];
        hipStream_t st[2] = {ctx->stream, ctx->stream2};
        for (uint32_t p = 0; p < P; p++)
        {
            const uint32_t s0 = (uint32_t)((unsigned long long)n_seg * p / P), s1 = (uint32_t)((unsigned long long)n_seg * (p + 1) / P);
            const uint32_t w0 = s0 * seg_waves, w1 = std::min<uint32_t>(n_waves, s1 * seg_waves);
            ap[p]           = a;
            ap[p].seg_base  = s0, ap[p].n_seg = s1 - s0;
            ap[p].wave_base = w0;
            wn[p]           = w1 - w0;
            tn[p]           = (direct ? 1u : 2u) * ((ap[p].n_seg + 7u) / 8u) * 8u * a.tiles_per_seg;
            const uint32_t all = 256u * (ctx->tune.single_blocks_per_cu ? ctx->tune.single_blocks_per_cu : 40u);
            sbn[p]          = (uint32_t)std::min<size_t>((size_t)wn[p] * 64u, all / P);
        }
        for (uint32_t p = 0; p < P && !direct; p++)
        {
            if (instr) hipLaunchKernelGGL(nn_lane_kernel<true>, dim3(wn[p]), dim3(64), 0, st[p], ap[p]);
            else hipLaunchKernelGGL(nn_lane_kernel<false>, dim3(wn[p]), dim3(64), 0, st[p], ap[p]);
        }
        if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
#define MP2P_LAUNCH_SEL(INSTR_, CERT_, DIRECT_, SOL_) \
    hipLaunchKernelGGL((nn_seltile_kernel<INSTR_, CERT_, DIRECT_, 4, SOL_>), dim3(tn[p]), dim3(64), 0, st[p], ap[p])
        for (uint32_t p = 0; p < P; p++)
        {
            if (sel)
            {
                if (sol == 1) MP2P_LAUNCH_SEL(false, false, true, 1);
                else if (sol == 2) MP2P_LAUNCH_SEL(false, false, true, 2);
                else if (sol == 3) MP2P_LAUNCH_SEL(false, false, true, 3);
                else if (sol == 4) MP2P_LAUNCH_SEL(false, false, true, 4);
                else if (sol == 5) MP2P_LAUNCH_SEL(false, false, true, 5);
                else if (instr) { if (direct) MP2P_LAUNCH_SEL(true, false, true, 0); else MP2P_LAUNCH_SEL(true, false, false, 0); }
                else if (ctx->tune.tile_waves == 3 || (ctx->tune.tile_waves == 0 && direct))
                {   // round 6: the register budget of 3 waves per SIMD (168): with the four staging loads of a round in flight together
                    // the 128-register build spills 3 dwords per lane behind the lane kernel and 11 with the prologue fused in
                    // (tools/kernel_resources.py).  Measured: behind the lane kernel (1 M queries: ten rounds of resident tiles, the
                    // occupancy counts) 4 waves with the spill win, 2 254 vs 2 193 it/s; with the fused prologue (small layers:
                    // one round of tiles, C2) the two are equal, 3 615 vs 3 625 it/s -- so small layers take the spill-free build
#define MP2P_LAUNCH_SEL3(CERT_, DIRECT_) hipLaunchKernelGGL((nn_seltile_kernel<false, CERT_, DIRECT_, 3, 0>), dim3(tn[p]), dim3(64), 0, st[p], ap[p])
                    if (cert_track) { if (direct) MP2P_LAUNCH_SEL3(true, true); else MP2P_LAUNCH_SEL3(true, false); }
                    else { if (direct) MP2P_LAUNCH_SEL3(false, true); else MP2P_LAUNCH_SEL3(false, false); }
#undef MP2P_LAUNCH_SEL3
                }
                else if (cert_track) { if (direct) MP2P_LAUNCH_SEL(false, true, true, 0); else MP2P_LAUNCH_SEL(false, true, false, 0); }
                else { if (direct) MP2P_LAUNCH_SEL(false, false, true, 0); else MP2P_LAUNCH_SEL(false, false, false, 0); }
            }
            else
            {   // a map without the level-0 occupancy bricks (or tile_select = 0): round 4's box-rule tile kernel, matrix-pipe prefilter
                if (instr) hipLaunchKernelGGL((nn_tile_kernel<32, true, true>), dim3(tn[p]), dim3(64), 0, st[p], ap[p]);
                else if (cert_track) hipLaunchKernelGGL((nn_tile_kernel<32, false, true, 4, true>), dim3(tn[p]), dim3(64), 0, st[p], ap[p]);
                else hipLaunchKernelGGL((nn_tile_kernel<32, false, true, 4>), dim3(tn[p]), dim3(64), 0, st[p], ap[p]);
            }
        }
#undef MP2P_LAUNCH_SEL
        if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        // deferred queries: the count lives on the device; a fixed grid strides over it
        for (uint32_t p = 0; p < P; p++)
        {
            if (instr) hipLaunchKernelGGL((nn_single_kernel<true, 1>), dim3(sbn[p]), dim3(64), 0, st[p], ap[p]);
            else if (ctx->tune.single_waves == 4) hipLaunchKernelGGL((nn_single_kernel<false, 1>), dim3(sbn[p]), dim3(64), 0, st[p], ap[p]);
            else hipLaunchKernelGGL((nn_single_kernel<false, 5>), dim3(sbn[p]), dim3(64), 0, st[p], ap[p]);
        }
        if (P > 1)
        {
            MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            MP2P_TRY_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    }
    else if (ctx->prof_all())
    {
        MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
        MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    }
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (reduce_bbox)
    {
        int rc = launch_bbox_reduce(ctx, n_boxes);
        if (rc) return rc;
    }
    MP2P_TRY_HIP(ctx, hipGetLastError());
    if (sol != 0)
    {  // a timing-only launch wrote no record: the warm start of the previous call stands
        ctx->hint_map = keep_hint_map;
        ctx->nn_lb2nd_valid = a.cert_read != 0;
        ctx->sol_no_records = true;  // phase 2 must not compact records this call did not write
        return MP2P_HIP_OK;
    }
    ctx->hint_map = map, ctx->hint_cloud = cloud, ctx->hint_n = n_l;
    for (int i = 0; i < 12; i++) ctx->hint_pose[i] = pose[i];
    ctx->nn_lb2nd_valid = a.lb2nd != nullptr;  // every finished query's entry was written by this call
    return MP2P_HIP_OK;
}

}  // namespace mp2p
