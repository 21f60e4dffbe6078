// nn_query.hip -- K1 (fused) + K3: local->global transform and EXACT nearest-neighbour search.
//
// Replaces, per outer ICP iteration,
//   transform_local_to_global           Matcher_Points_Base.cpp:183-249
//   the nn_single_search loop            Matcher_Points_DistanceThreshold.cpp:214-243
//   the threshold rule                   :252-259
//   the "global point already paired"    :94-121 (claims; resolved in pairs.hip)
//
// Two kernels, both one wave64 per workgroup:
//
//  nn_tile_kernel  -- a TILE of Q Morton-consecutive queries per wave.  lane = (query slot,
//     candidate slice).  Every lane keeps its query in registers; the wave stages the points of
//     all voxels overlapping the search box of the current GROUP of queries into LDS (SoA) with
//     coalesced 16-byte loads, then every lane scans the staged bucket against its own query,
//     8 candidates per step (ds_read_b128 broadcasts).  Each HBM/L2 byte is fetched once per
//     tile and reused by up to 64 queries.  Queries that are spatially isolated inside their
//     tile (sparse far-range returns, tiles straddling a jump of the Morton curve) would make
//     the shared box dwarf their search balls; they are DEFERRED to
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
#include "device_utils.hpp"

namespace mp2p
{
constexpr int NN_CAP      = 256;  // staged candidates per round (LDS: 5 x 1 KB)
constexpr int NN_COOP_MAX = 4;    // groups of up to this many queries are deferred

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
    const unsigned char* local_taken;   // by original local index, or null
    const unsigned char* global_taken;  // by original global index, or null
    unsigned long long*  claims;        // by sorted global position, or null
    unsigned long long   claim_hi;      // (~epoch) << 32
    unsigned long long   local_offset;  // whole-layer index of this rank's first local point
    uint32_t*            out_spos;      // [n_l] in the order of lpts (coalesced; pairs.hip maps back)
    float*               out_d2;
    float*               tile_bbox;  // [n_tiles][6]
    uint4*               work;       // deferred queries {sorted idx, r, best_d2, best_idx}
    uint32_t*            work_spos;  //   + best_spos
    uint32_t*            work_count;
    // warm start, [n_l] by original local index: {sorted position of the nearest neighbour found by
    // the previous call on the same (map, cloud) pair or NONE, lower bound on the SQUARED distance
    // to every map point at that call's pose}, in the order of lpts; read at entry, rewritten at exit
    uint2*               hint;
    const uint32_t*      rank;  // visit rank per original local index (NONE = not visited) or null
    int                  use_hint;
    PoseRt               prev_pose;
    unsigned long long*  counters;  // profiling, or null
    unsigned char*       touched;   // profiling: [n_g] by sorted position, or null
    // profiling level 4: {start, end} 100 MHz ticks of every workgroup of the two search kernels
    // ([n_tiles] then [single blocks]); the plain kernels only pay a uniform null test for it
    unsigned long long*  timeline;
    uint32_t             timeline_single_base;
    // launch order: tiles are dispatched in blockIdx order and a late straggler leaves the chip
    // idle behind it, so the tiles that took longest in the previous call (same map, same cloud)
    // go first.  tile_cost[tile] = this call's duration in 100 MHz ticks (written at exit);
    // tile_order[blockIdx.x] = tile, or null (cold call: identity)
    uint32_t*            tile_cost;
    const uint32_t*      tile_order;
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
    if (md2 <= prune2 && occ_maybe(g, b.lev, cx, cy, cz))
    {
        uint32_t e = 0;
        if (cell_lookup(g, cell_key(b.lev, cx, cy, cz), start, e)) cnt = e - start;
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

__device__ __forceinline__ void emit_result(const NNArgs& a, uint32_t qi, uint32_t orig, bool active, float thr,
                                            float best_d2, uint32_t best_idx, uint32_t best_spos,
                                            float lb2_keep = 0.f)
{
    bool acc = active && best_idx != NONE_U32 && best_d2 < thr;  // :259
    if (acc && a.global_taken && a.global_taken[best_idx]) acc = false;  // :98-101
    a.out_spos[qi] = acc ? best_spos : NONE_U32;
    a.out_d2[qi]   = best_d2;
    // next call's warm start: the raw nearest neighbour (even if rejected) and what this search
    // proved: no map point is nearer than min(best, threshold) (every point that could pass the
    // threshold was examined), or than the bound that let the search be skipped
    const float lb2 = active ? fmaxf(fminf(best_d2, thr), lb2_keep) : 0.f;
    a.hint[qi]      = make_uint2(best_spos, __float_as_uint(lb2));
    if (acc && a.claims)
    {
        const uint32_t vrank = a.rank ? a.rank[orig] : orig;  // the order the sequential loop visits
        atomicMin(&a.claims[best_spos], a.claim_hi | (a.local_offset + vrank));
    }
}

// push the lanes of `mask` (one entry per query slot) onto the deferred-query list
template <int Q>
__device__ __forceinline__ uint32_t defer_lanes(const NNArgs& a, bool mine, unsigned long long mask,
                                                int lane, int slice, uint32_t qi, float r,
                                                float best_d2, uint32_t best_idx, uint32_t best_spos)
{
    const unsigned long long slot_mask = (Q < 64) ? ((1ull << (Q & 63)) - 1ull) : ~0ull;
    const unsigned long long push      = mask & slot_mask;
    const int                npush     = __popcll(push);
    uint32_t                 base_slot = 0;
    if (lane == 0) base_slot = atomicAdd(a.work_count, (uint32_t)npush);
    base_slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_slot);
    if (mine && slice == 0)
    {
        const uint32_t slot = base_slot + (uint32_t)__popcll(push & ((1ull << lane) - 1ull));
        a.work[slot]      = make_uint4(qi, __float_as_uint(r), __float_as_uint(best_d2), best_idx);
        a.work_spos[slot] = best_spos;
    }
    return (uint32_t)npush;
}

// ================================================================================================
// 5 waves per SIMD: 96 VGPRs with 4 spilled dwords; measured +3.5 % over the compiler's own 108
// VGPRs / 4 waves, while 6 waves (80 VGPRs, 22 spilled dwords) give the gain back
template <int Q, bool INSTR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void nn_tile_kernel(const NNArgs a)
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

    const GridView& g     = a.g;
    const int       lane  = threadIdx.x;
    const unsigned long long tl0 = wall_clock64();
    const int       qslot = lane & (Q - 1);
    const int       slice = (Q == 64) ? 0 : lane / Q;
    const uint32_t  tile  = a.tile_order ? a.tile_order[blockIdx.x] : blockIdx.x;
    const uint32_t  qi    = tile * Q + qslot;
    const bool      valid = qi < a.n_l;

    float4 lp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = a.lpts[qi];
    // the warm-start record is in the same (Morton) order: its load is issued together with the
    // point's, not behind it
    uint2 h = make_uint2(NONE_U32, 0u);
    if (valid && a.use_hint) h = a.hint[qi];
    const uint32_t orig = __float_as_uint(lp.w);
    // a visit order on the cloud (maxLocalPointsPerLayer, Matcher_Points_Base.cpp:222-246): only
    // the listed points are transformed, boxed and matched
    bool visited = valid;
    if (a.rank && valid) visited = a.rank[orig] != NONE_U32;

    // ---- K1: transform (fp64 compose, one narrowing) ------------------------------------
    float qx, qy, qz;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);

    // bounding box of ALL transformed local points of the tile (Matcher_Points_Base.cpp:186-196)
    {
        const float bx0 = wave_min_nn((visited && qx == qx) ? qx : INFINITY), by0 = wave_min_nn((visited && qy == qy) ? qy : INFINITY),
                    bz0 = wave_min_nn((visited && qz == qz) ? qz : INFINITY);
        const float bx1 = wave_max_nn((visited && qx == qx) ? qx : -INFINITY), by1 = wave_max_nn((visited && qy == qy) ? qy : -INFINITY),
                    bz1 = wave_max_nn((visited && qz == qz) ? qz : -INFINITY);
        if (lane == 0)
        {
            float* o = a.tile_bbox + (size_t)tile * 6;
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
    bool     deferred = false;
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
    float lb2_keep = 0.f;
    if (a.use_hint && active)
    {
        float       ox, oy, oz;
        compose_point_f(a.prev_pose, lp.x, lp.y, lp.z, ox, oy, oz);
        const float disp = sqrtf(dist2(qx, qy, qz, ox, oy, oz));
        float       lb   = sqrtf(__uint_as_float(h.y)) * 0.99999f - disp * 1.00001f - 4.f * g.slack;
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
            done     = true;
            lb2_keep = (lb * 0.9999f) * (lb * 0.9999f);
        }
        else if (lb > r * (1.0f - 1.0f / 1024.0f) - g.slack)
            r = fminf(fmaxf(hr > 0.f ? fminf(hr, 2.0f * lb) : 2.0f * lb, r), rmax);
    }
    // a query whose radius already exceeds what a tile should carry goes straight to the
    // one-query-per-wave kernel
    {
        const bool               wide  = !done && r > a.r_defer;
        const unsigned long long wmask = __ballot(wide);
        if (wmask)
        {
            defer_lanes<Q>(a, wide, wmask, lane, slice, qi, r, best_d2, best_idx, best_spos);
            if (wide) done = true, deferred = true;
        }
    }

    uint32_t        st_pass = 0, st_cells = 0, st_cand = 0, st_defer = 0;
    const long long t_start = INSTR ? (long long)wall_clock64() : 0;

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
        if (__popcll(gmask) <= NN_COOP_MAX * S)
        {
            st_defer += defer_lanes<Q>(a, grp, gmask, lane, slice, qi, r, best_d2, best_idx, best_spos);
            if (grp) done = true, deferred = true;
            continue;
        }
        st_pass++;

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
        const PassBox box    = choose_level(g, lox, loy, loz, hix, hiy, hiz, a.cell_budget);
        const float   prune  = rmax_t + 4.f * g.slack;
        const float   prune2 = prune * prune;

        const v2f qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        for (unsigned long long cb = 0; cb < box.ncell; cb += 64)
        {
            uint32_t cnt, start;
            float    md2_unused;
            lookup_voxel(g, box, cb + lane, qlx, qly, qlz, qhx, qhy, qhz, prune2, start, cnt, md2_unused);
            const uint32_t incl  = wave_incl_scan(cnt, lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            st_cells += (uint32_t)min((unsigned long long)64, box.ncell - cb);
            if (total == 0) continue;  // uniform: no occupied voxel in this batch
            const uint32_t off = incl - cnt;
            s_cstart[lane] = start;
            s_coff[lane]   = off;
            st_cand += total;

            for (uint32_t base = 0; base < total; base += NN_CAP)
            {
                const uint32_t m     = min((uint32_t)NN_CAP, total - base);
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
                    const uint32_t ow[4] = {max(ex, p0), max(ex, p1), max(ex, p2), max(ex, p3)};
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
                        c4[k] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
                        if (t0 + k < m) c4[k] = g.pts[src[k]];
                    }
                    *reinterpret_cast<float4*>(&s_x[t0]) = make_float4(c4[0].x, c4[1].x, c4[2].x, c4[3].x);
                    *reinterpret_cast<float4*>(&s_y[t0]) = make_float4(c4[0].y, c4[1].y, c4[2].y, c4[3].y);
                    *reinterpret_cast<float4*>(&s_z[t0]) = make_float4(c4[0].z, c4[1].z, c4[2].z, c4[3].z);
                    *reinterpret_cast<uint4*>(&s_idx[t0]) =
                        make_uint4(__float_as_uint(c4[0].w), __float_as_uint(c4[1].w),
                                   __float_as_uint(c4[2].w), __float_as_uint(c4[3].w));
                    *reinterpret_cast<uint4*>(&s_spos[t0]) = make_uint4(src[0], src[1], src[2], src[3]);
                    if (INSTR)
                    {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (t0 + k < m) a.touched[src[k]] = 1;
                    }
                }
                __syncthreads();
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

        bool too_wide = false;
        if (grp)
        {
            if (is_final(r, rmax, best_d2, g.slack)) done = true;
            else
            {
                r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
                // a query whose radius outgrows the voxels would drag the shared box with it:
                // it continues alone, with the whole wave on its own candidates
                too_wide = r > a.r_defer;
            }
        }
        const unsigned long long wmask = __ballot(too_wide);
        if (wmask)
        {
            st_defer += defer_lanes<Q>(a, too_wide, wmask, lane, slice, qi, r, best_d2, best_idx, best_spos);
            if (too_wide) done = true, deferred = true;
        }
    }

    // ---- output (original local order) + claim of the global point --------------------------
    if (valid && slice == 0 && !deferred)
        emit_result(a, qi, orig, active, thr, best_d2, best_idx, best_spos, lb2_keep);

    if (lane == 0)
    {
        const unsigned long long tl1 = wall_clock64();
        a.tile_cost[tile] = (uint32_t)min(tl1 - tl0, 0xFFFFFFFFull);
        if (a.timeline) a.timeline[2 * (size_t)tile] = tl0, a.timeline[2 * (size_t)tile + 1] = tl1;
    }
    if (INSTR && lane == 0)
    {
        atomicAdd(&a.counters[0], 1ull);
        atomicAdd(&a.counters[1], (unsigned long long)st_pass);
        atomicAdd(&a.counters[2], (unsigned long long)st_cells);
        atomicAdd(&a.counters[3], (unsigned long long)st_cand);
        if (st_pass > 1) atomicAdd(&a.counters[4], 1ull);
        atomicMax(&a.counters[5], (unsigned long long)st_cand);
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
            const uint32_t t = t0 + 64u * k + lane;
            ca[k] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
            if (t < total) ca[k] = g.pts[sa[k]];
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

constexpr int NN_VLIST = 1024;  // occupied voxels listed per round (LDS)

template <bool INSTR>
__global__ __launch_bounds__(64) void nn_single_kernel(const NNArgs a)
{
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[64];
    __shared__ uint32_t s_vox[NN_VLIST];
    const GridView& g      = a.g;
    const int       lane   = threadIdx.x;
    const uint32_t  n_work = *a.work_count;
    const unsigned long long tl0 = a.timeline ? wall_clock64() : 0ull;

    for (uint32_t item = blockIdx.x; item < n_work; item += gridDim.x)
    {
        const uint4    w    = a.work[item];
        const uint32_t qi   = w.x;
        const float4   lp   = a.lpts[qi];
        const uint32_t orig = __float_as_uint(lp.w);
        float          qx, qy, qz;
        compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);
        const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
        const float thr    = fadd(a.maxDistSq, fmul(a.angSq, normSq));
        const float rmax   = sqrtf(thr) * 1.002f + g.slack;
        float       r      = __uint_as_float(w.y);
        // wave-uniform running best (carried over from the tile kernel)
        float    best_d2  = __uint_as_float(w.z);
        uint32_t best_idx = w.w, best_spos = a.work_spos[item];
        uint32_t st_pass = 0, st_cand = 0, st_cells = 0;
        const long long t_start = INSTR ? (long long)wall_clock64() : 0;

        for (;;)
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
                                    if (cell_lookup(g, cell_key(lev, cx, cy, cz), start, e)) cnt = e - start;
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
            if (is_final(r, rmax, best_d2, g.slack)) break;
            r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
        }
        if (lane == 0) emit_result(a, qi, orig, true, thr, best_d2, best_idx, best_spos);
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

__global__ void zero_u32_kernel(uint32_t* p) { *p = 0; }

// Resets the work-queue counter and, for a warm call, lists the tiles by decreasing duration of
// the previous call: a counting sort on the log2 of the tick count (only "the long ones first"
// matters).  One workgroup of 16 waves; a wave adds each distinct bucket of its 64 tiles with ONE
// LDS atomic (the durations cluster in a handful of buckets: per-tile atomics would serialise).
// Measured on the bench scene: the tile kernel drops from 0.222 to 0.187 ms, but this kernel's two
// passes over the costs are 62 dependent global loads in a row = 47 us, more than the gain, and
// letting the tiles count themselves with a global atomic doubles THEIR time (a few hot
// addresses) -- hence opt-in (mp2p_hip_pt2pt_params::tile_order) until the sort is cheap.
__device__ __forceinline__ uint32_t wave_bucket_slot(uint32_t* s_ctr, uint32_t bucket, bool have, int lane)
{
    // returns, for every lane with `have`, a distinct slot of its bucket's counter range
    uint32_t           slot = 0;
    unsigned long long todo = __ballot(have);
    while (todo)
    {
        const int                leader = __ffsll((long long)todo) - 1;
        const uint32_t           b      = (uint32_t)__builtin_amdgcn_readlane((int)bucket, leader);
        const unsigned long long same   = __ballot(have && bucket == b) & todo;
        uint32_t                 base   = 0;
        if (lane == leader) base = atomicAdd(&s_ctr[b], (uint32_t)__popcll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if ((same >> lane) & 1ull) slot = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t n_tiles,
                                                          uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ work_count)
{
    __shared__ uint32_t s_cnt[33], s_off[33];
    if (threadIdx.x == 0) *work_count = 0;
    if (!order) return;  // uniform
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 33) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // bucket 0: cost 0, bucket b: 2^(b-1) <= cost < 2^b
    const uint32_t rounds = (n_tiles + 1023u) / 1024u;
    for (uint32_t k = 0; k < rounds; k++)
    {
        const uint32_t t    = k * 1024u + threadIdx.x;
        const bool     have = t < n_tiles;
        const uint32_t b    = have ? (uint32_t)(32 - __clz((int)cost[t])) : 0u;
        (void)wave_bucket_slot(s_cnt, b, have, lane);
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t run = 0;
        for (int b = 32; b >= 0; b--) s_off[b] = run, run += s_cnt[b];  // long tiles first
    }
    __syncthreads();
    for (uint32_t k = 0; k < rounds; k++)
    {
        const uint32_t t    = k * 1024u + threadIdx.x;
        const bool     have = t < n_tiles;
        const uint32_t b    = have ? (uint32_t)(32 - __clz((int)cost[t])) : 0u;
        const uint32_t slot = wave_bucket_slot(s_off, b, have, lane);
        if (have) order[slot] = t;
    }
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
int launch_nn_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                    const double pose[12], const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms)
{
    const size_t n_l = cloud->n;
    uint32_t     Q   = prm->queries_per_wave ? prm->queries_per_wave : 32;
    MP2P_REQUIRE(ctx, Q == 64 || Q == 32 || Q == 16, "queries_per_wave must be 64, 32 or 16");
    const uint32_t n_tiles = (uint32_t)((n_l + Q - 1) / Q);

    MP2P_TRY_HIP(ctx, ctx->nn_spos.ensure(n_l));
    MP2P_TRY_HIP(ctx, ctx->nn_d2.ensure(n_l));
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)n_tiles * 6));
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    MP2P_TRY_HIP(ctx, ctx->work.ensure(n_l));
    MP2P_TRY_HIP(ctx, ctx->work_spos.ensure(n_l + 1));  // last word = counter
    ctx->last_n_tiles = n_tiles;
    ctx->last_q       = Q;

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
    a.local_taken =
        (ms && !prm->allowMatchAlreadyMatchedPoints) ? ms->local_taken.p : nullptr;
    a.global_taken =
        (ms && !prm->allowMatchAlreadyMatchedGlobalPoints) ? ms->global_taken.p : nullptr;
    a.claims = prm->allowMatchAlreadyMatchedGlobalPoints ? nullptr : map->claims.p;
    ctx->epoch++;
    a.claim_hi     = (~(unsigned long long)ctx->epoch) << 32;
    a.local_offset = prm->local_index_offset;
    a.rank         = cloud->n_visit ? cloud->rank.p : nullptr;
    a.out_spos     = ctx->nn_spos.p;
    a.out_d2       = ctx->nn_d2.p;
    a.tile_bbox    = ctx->tile_bbox.p;
    a.work         = ctx->work.p;
    a.work_spos    = ctx->work_spos.p;
    a.work_count   = ctx->work_spos.p + n_l;
    MP2P_TRY_HIP(ctx, ctx->hint.ensure(n_l));
    a.hint     = ctx->hint.p;
    for (int i = 0; i < 9; i++) a.prev_pose.r[i] = ctx->hint_pose[i];
    for (int i = 0; i < 3; i++) a.prev_pose.t[i] = ctx->hint_pose[9 + i];
    a.use_hint = (ctx->hint_map == map && ctx->hint_cloud == cloud && ctx->hint_n == n_l && !prm->disable_warm_start) ? 1 : 0;
    ctx->hint_map = map, ctx->hint_cloud = cloud, ctx->hint_n = n_l;
    for (int i = 0; i < 12; i++) ctx->hint_pose[i] = pose[i];
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
        const size_t single_blocks = std::min<size_t>(n_l, 256u * 32u);
        MP2P_TRY_HIP(ctx, ctx->timeline.ensure(2 * (n_tiles + single_blocks)));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->timeline.p, 0, 2 * (n_tiles + single_blocks) * sizeof(unsigned long long),
                                         ctx->stream));
        a.timeline = ctx->timeline.p;
        ctx->timeline_tiles = n_tiles, ctx->timeline_singles = single_blocks;
    }
    // tile order from the previous call's durations (same condition as the warm start: the same
    // local layer in the same tiling against the same map)
    MP2P_TRY_HIP(ctx, ctx->tile_cost.ensure(n_tiles ? n_tiles : 1));
    MP2P_TRY_HIP(ctx, ctx->tile_order.ensure(n_tiles ? n_tiles : 1));
    const bool ordered = a.use_hint && ctx->tile_cost_tiles == n_tiles && ctx->tile_cost_q == Q && n_tiles > 1 &&
                         prm->tile_order != 0;
    a.tile_cost  = ctx->tile_cost.p;
    a.tile_order = ordered ? ctx->tile_order.p : nullptr;
    ctx->tile_cost_tiles = n_tiles, ctx->tile_cost_q = Q;
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->tile_cost.p, n_tiles,
                       ordered ? ctx->tile_order.p : nullptr, a.work_count);
    // ev[0]..ev[1] brackets exactly the two search kernels (the roofline kernels of bench.py)
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (n_tiles)
    {
        const bool     instr         = a.counters != nullptr;
        const uint32_t single_blocks = (uint32_t)std::min<size_t>(n_l, 256u * 32u);
#define MP2P_LAUNCH_TILE(QQ)                                                                       \
    do                                                                                             \
    {                                                                                              \
        if (instr) hipLaunchKernelGGL((nn_tile_kernel<QQ, true>), dim3(n_tiles), dim3(64), 0, ctx->stream, a);  \
        else hipLaunchKernelGGL((nn_tile_kernel<QQ, false>), dim3(n_tiles), dim3(64), 0, ctx->stream, a);      \
    } while (0)
        if (Q == 64) MP2P_LAUNCH_TILE(64);
        else if (Q == 32) MP2P_LAUNCH_TILE(32);
        else MP2P_LAUNCH_TILE(16);
#undef MP2P_LAUNCH_TILE
        if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        // deferred queries: the count lives on the device; a fixed grid strides over it
        if (instr) hipLaunchKernelGGL(nn_single_kernel<true>, dim3(single_blocks), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL(nn_single_kernel<false>, dim3(single_blocks), dim3(64), 0, ctx->stream, a);
    }
    else if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    int rc = launch_bbox_reduce(ctx, n_tiles);
    if (rc) return rc;
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

}  // namespace mp2p
