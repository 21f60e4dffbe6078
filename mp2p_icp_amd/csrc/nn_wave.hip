// nn_wave.hip -- K1 + K3, round 3: ONE kernel for transform, warm start and the exact nearest-neighbour
// search of (nearly) every query; replaces nn_lane_kernel + nn_tile_kernel of nn_query.hip on maps
// that carry a level-0 occupancy bitmap (the default build).
//
// Reference loop replaced: Matcher_Points_DistanceThreshold.cpp:214-265 (+ transform_local_to_global,
// Matcher_Points_Base.cpp:183-249).
//
// Why another structure (DESIGN.md section 4 has the numbers): the tile kernel staged ONE box per 32
// queries -- sized by the worst radius and the tile's extent -- and paid a chain of ~6 dependent round
// trips per 32 queries.  On the bench chains 80 % of the queries have a previous nearest neighbour
// 2..9 cm away while a few per tile are 0.3..1 m off.  Here a wave owns 64 Morton-consecutive queries
// (one per lane); every step below is wave-uniform control flow:
//   A. each lane's ball gives a cube of level-0 voxels; the wave looks at a WINDOW of up to 4 x 4 x 4
//      occupancy bricks (4x4x4 voxels, one 8-byte word each, all words fetched by one load) around its
//      first open query; per brick every lane ANDs the word with the mask of its own cube and a wave-wide OR
//      gives the set of occupied voxels ANY lane needs -- the union of the cubes, not their bounding box;
//   B. lane b resolves voxel b of each brick through the directory (one 8-byte load per voxel, all
//      independent), the voxels holding points are listed and their counts prefixed;
//   C. the listed points are staged into LDS (SoA) with coalesced 16-byte loads, 256 per round;
//   D. every lane tests every staged point against its query in the exact FMA-free sequence, 8 per step
//      on packed fp32 (the update is rare).
// The radius of a warm query is the distance to its previous nearest neighbour (certain to conclude,
// however small), so one pass finishes it.  Lanes whose cube does not fit the window wait for the next
// pass (new window); queries whose cube exceeds 8 voxels per axis or whose radius outgrows r_defer go to
// nn_single_kernel (a whole wave per query) as before.  Radii above one voxel are served in groups of 16 queries
// (16 queries x 4 slices of the staged points) after the narrow ones.
// Results are bit-identical to the other kernels' (same arg-min rule, same finality rule).
#include "device_utils.hpp"

namespace mp2p
{
constexpr int NW_NL      = 256;  // listed voxels per chunk
constexpr int NW_CAP     = 256;  // staged points per round
constexpr int NW_MAXPASS = 10;  // (a wave of 64 wide cubes takes 4 group passes)
constexpr uint32_t NW_WIN = 4;     // window: at most this many bricks per axis
constexpr uint32_t NW_MAXSPAN = 8;  // widest cube (level-0 voxels per axis) the window can hold at any alignment

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t k)
{
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR1>((int)k));
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR2>((int)k));
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR4>((int)k));
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR8>((int)k));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)k, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)k, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)k, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t k)
{
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR1>((int)k));
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR2>((int)k));
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR4>((int)k));
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR8>((int)k));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)k, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)k, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)k, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
    return max(max(a, b), max(c, d));
}

// 64-bit occupancy mask of the level-0 voxel cube [cx0..cx1] x [cy0..cy1] x [cz0..cz1] (at most 4 voxels
// per axis), bit = (z - cz0) * 16 + (y - cy0) * 4 + (x - cx0), from the <= 2 x 2 x 2 bricks it overlaps
// (all loads independent).  Same construction as nn_lane_kernel's.
__device__ __forceinline__ unsigned long long cube_occ_mask(const GridView& g, uint32_t cx0, uint32_t cy0, uint32_t cz0,
                                                            uint32_t cx1, uint32_t cy1, uint32_t cz1)
{
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
    return m;
}

// counters of the instrumented variant (ctx->counters): 50.. = this kernel's own
enum
{
    NWC_LANE_TESTS = 50,  // distance tests (staged points x open lanes)
    NWC_MAXLANE    = 51,  // staged points scanned, summed over waves (what a wave's scan costs)
    NWC_INSERTS    = 52,  // bricks of all windows that held a needed voxel
    NWC_OVF        = 53,  // lane-passes spent waiting for a window (cube outside the current one)
    NWC_ROUNDS     = 54,  // staging rounds
    NWC_LISTED     = 55,  // voxels listed
    NWC_TOOBIG     = 56,  // queries handed on for their cube's width (too wide, or wide and alone)
    NWC_T_PRO      = 57,  // 100 MHz ticks per phase, summed over waves: prologue (loads, transform, warm start)
    NWC_T_INS      = 58,  //   window, brick words, masks
    NWC_T_DIR      = 59,  //   directory resolve + list
    NWC_T_STAGE    = 60,  //   staging
    NWC_T_SCAN     = 61,  //   distance tests
    NWC_T_EMIT     = 62,  //   hand-over + records + claims
};

__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo |= (uint32_t)dpp_i<DPP_ROW_ROR1>((int)lo), hi |= (uint32_t)dpp_i<DPP_ROW_ROR1>((int)hi);
    lo |= (uint32_t)dpp_i<DPP_ROW_ROR2>((int)lo), hi |= (uint32_t)dpp_i<DPP_ROW_ROR2>((int)hi);
    lo |= (uint32_t)dpp_i<DPP_ROW_ROR4>((int)lo), hi |= (uint32_t)dpp_i<DPP_ROW_ROR4>((int)hi);
    lo |= (uint32_t)dpp_i<DPP_ROW_ROR8>((int)lo), hi |= (uint32_t)dpp_i<DPP_ROW_ROR8>((int)hi);
    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lo, 0) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 16) |
                       (uint32_t)__builtin_amdgcn_readlane((int)lo, 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 48);
    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hi, 0) | (uint32_t)__builtin_amdgcn_readlane((int)hi, 16) |
                       (uint32_t)__builtin_amdgcn_readlane((int)hi, 32) | (uint32_t)__builtin_amdgcn_readlane((int)hi, 48);
    return ((unsigned long long)h << 32) | l;
}

// bit b of an 8-bit value -> the nibble at bit 4b (all four bits)
__device__ __forceinline__ uint32_t nibble_spread8(uint32_t x)
{
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x * 0xFu;
}
// 4-bit mask of the voxels 4b..4b+3 of brick b that lie in [c0, c1]; 0 when they do not overlap
__device__ __forceinline__ uint32_t axis_mask_in(uint32_t b, uint32_t c0, uint32_t c1)
{
    const uint32_t lo = max(c0, b * 4u), hi = min(c1, b * 4u + 3u);
    const uint32_t m  = ((2u << ((hi - b * 4u) & 3u)) - 1u) & ~((1u << ((lo - b * 4u) & 3u)) - 1u);
    return lo <= hi ? m : 0u;
}

// WPE = waves per SIMD the register allocation aims at (MP2P_HIP_TUNE wave_waves)
template <bool INSTR, int WPE, bool MF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void nn_wave_kernel(const NNArgs a)
{
    __shared__ __attribute__((aligned(16))) uint2    s_list[NW_NL];  // listed voxels (all hold points): {first sorted position, first staged slot}
    __shared__ __attribute__((aligned(16))) float    s_x[NW_CAP];
    __shared__ __attribute__((aligned(16))) float    s_y[NW_CAP];
    __shared__ __attribute__((aligned(16))) float    s_z[NW_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_idx[NW_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_spos[NW_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_owner[NW_CAP];  // staging: slot -> listed voxel; before: the voxel codes
    __shared__ uint32_t s_grp[16];  // group pass: the lanes of the group's members
    static_assert(NW_NL == NW_CAP, "the voxel codes of a chunk live in s_owner");
    static_assert(NN_CLAIM_SLOTS * sizeof(unsigned long long) <= 3 * NW_CAP * sizeof(float), "claim table fits s_x..s_z");
    unsigned long long* s_claim = reinterpret_cast<unsigned long long*>(s_x);
    uint32_t*           s_vox   = s_owner;

    const GridView& g    = a.g;
    const int       lane = threadIdx.x;
    const uint32_t  wv   = a.wave_base + blockIdx.x;  // wave of the whole layer
    const uint32_t  qi   = wv * 64u + (uint32_t)lane;
    const bool      valid = qi < a.n_l;
    const unsigned long long tl0 = (a.timeline || INSTR) ? wall_clock64() : 0ull;
    unsigned long long       tph = tl0;  // phase clock (INSTR)
    unsigned long long       t_ins = 0, t_dir = 0, t_stage = 0, t_scan = 0;

    // ================= prologue: as nn_lane_kernel (K1 transform, box, threshold rule, MatchState, warm start)
    float4 lp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = a.lpts[qi];
    uint4 h = make_uint4(NONE_U32, 0u, 0u, 0u);
    if (valid && a.use_hint) h = a.rec[qi];
    const uint32_t orig = __float_as_uint(lp.w);
    bool visited = valid;
    if (a.rank && valid) visited = a.rank[orig] != NONE_U32;

    float qx, qy, qz;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);
    // (stored at the very end: a store in flight holds up every later wait for a load -- vmcnt counts both)
    const float wbx0 = wave_min_nn((visited && qx == qx) ? qx : INFINITY), wby0 = wave_min_nn((visited && qy == qy) ? qy : INFINITY),
                wbz0 = wave_min_nn((visited && qz == qz) ? qz : INFINITY);
    const float wbx1 = wave_max_nn((visited && qx == qx) ? qx : -INFINITY), wby1 = wave_max_nn((visited && qy == qy) ? qy : -INFINITY),
                wbz1 = wave_max_nn((visited && qz == qz) ? qz : -INFINITY);
    const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
    const float thr    = fadd(a.maxDistSq, fmul(a.angSq, normSq));
    const float rmax   = sqrtf(thr) * 1.002f + g.slack;

    bool active = visited && (normSq < INFINITY);
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // :218-220

    float    r        = fminf(a.r0, rmax);
    bool     done     = !active;
    float    best_d2  = INFINITY;
    uint32_t best_idx = NONE_U32, best_spos = NONE_U32;
    float    lb2_out  = -1.f;
    if (a.use_hint && active)
    {
        float ox, oy, oz;
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
        else
        {
            // the previous neighbour's distance is the radius certain to conclude -- taken as it is,
            // however small (the tile kernel never went below one voxel); a far one (stale warm start)
            // must not blow the ball up: never more than twice the proven lower bound / the first radius
            const float cap = fmaxf(2.0f * lb, r);
            r = fminf(hr > 0.f ? fminf(hr, cap) : cap, rmax);
        }
    }
    bool deferred = false;
    // a query the previous call predicted to be far: nn_single_kernel<PRED> serves it on the other stream (NNArgs::pred)
    if (a.pred != nullptr && valid && (h.w & 2u)) done = true, deferred = true;
    const float hs = g.hf * (float)(1u << g.shift0);  // level-0 voxel edge
    uint32_t dbg_nu = 0, dbg_rounds = 0, dbg_flags = 0;  // profiling level 4: packed into the timeline record
    uint32_t st_pass = 0, st_T = 0, st_tests = 0, st_bricks = 0, st_wait = 0, st_rounds = 0, st_listed = 0, st_toobig = 0,
             st_defer = 0;
    const bool clocks = INSTR || a.timeline != nullptr;  // uniform
    unsigned long long t_pro = 0;
    if (clocks)
    {
        const unsigned long long t = wall_clock64();
        t_pro = t - tph, tph = t;
    }
    const uint32_t n_search = (uint32_t)__popcll(__ballot(!done));
    const uint32_t n_skip   = (uint32_t)__popcll(__ballot(active && lb2_out >= 0.f));
    const v2f      qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // the coarsest level a pass may use: it needs an occupancy bitmap (levels are built finest first)
    // (MP2P_HIP_TUNE wave_levels, default 0 = level 0 only: coarser voxels for the wide groups were measured WORSE --
    //  scene A 0.230 -> 0.315 ms, scene B 0.429 -> 0.794 ms: twice the staged points, the rounds wait for their loads)
    uint32_t lev_max = 0;
    while (lev_max + 1u < g.n_levels && lev_max < a.wave_lev_max && g.occ_off[lev_max + 1u] != OCC_NONE) lev_max++;
    const float hs0   = hs;
    // a cube of this half-width spans <= 8 voxels of level 0 (any alignment fits the 4-brick window), <= 5 of a coarser one
    const float r_fit = (lev_max == 0u ? 3.4f : 1.7f * (float)(1u << lev_max)) * hs0;

    // ================= passes
    for (int pass = 0; pass < NW_MAXPASS; pass++)
    {
        bool        part = !done && !deferred;
        const float lox = fmaxf(qx - r, g.bbmin[0]), loy = fmaxf(qy - r, g.bbmin[1]), loz = fmaxf(qz - r, g.bbmin[2]);
        const float hix = fminf(qx + r, g.bbmax[0]), hiy = fminf(qy + r, g.bbmax[1]), hiz = fminf(qz + r, g.bbmax[2]);
        const bool  empty = !part || (lox > hix) || (loy > hiy) || (loz > hiz);  // nothing to visit at this radius
        // ---- who leaves for the one-query-per-wave kernel (a whole wave per query): a radius beyond r_defer, or
        //      one whose cube no usable level holds in a window ---------------------------------------------
        {
            const bool leave = part && !empty && (r > a.r_defer || r > r_fit);
            const unsigned long long lmask = __ballot(leave);
            if (lmask)
            {
                st_defer += push_lanes(a, 1, wv / a.seg_waves, leave, lmask, lane, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
                if (leave) deferred = true, part = false;
                if (INSTR) st_toobig += (uint32_t)__popcll(__ballot(leave && r <= a.r_defer));
                dbg_flags |= 2u;
            }
        }
        if (__ballot(part) == 0ull) break;  // uniform
        st_pass++;
        // ---- the kind of pass.  NARROW cubes (radius up to one level-0 voxel: a few dozen points each, shared by
        //      the neighbours) are served 64 at a time at level 0, every lane testing every staged point.  WIDE
        //      cubes hold hundreds of points per query and the union over 64 queries thousands -- all of which
        //      every lane would test: they are served in GROUPS of 16 Morton-adjacent queries (a quarter of the
        //      union), the 64 lanes being 16 queries x 4 slices of the staged points, and at the LEVEL whose voxels
        //      their radius spans in <= 5 (a 0.9 m ball is 8 level-0 voxels wide -- a 64-brick window and hundreds
        //      of directory entries -- but 4 of level 1).  Narrow ones first. ------------------------------------
        const bool               wide  = part && !empty && r > hs0;
        const unsigned long long nmask = __ballot(part && !empty && !wide);
        const bool               grp   = nmask == 0ull;  // uniform: only wide cubes are left
        bool                     want  = part && !empty && (grp || !wide);
        if (grp) want = want && (uint32_t)__popcll(__ballot(want) & lane_lt) < 16u;  // the first 16 of them
        uint32_t lev = 0;
        if (grp)
        {
            const float rg = wave_max_pos(want ? r : 0.f);
            while (lev < lev_max && rg > 1.7f * hs0 * (float)(1u << lev)) lev++;
        }
        const uint32_t sh  = g.shift0 + lev;
        const float    hs  = hs0 * (float)(1u << lev);  // voxel edge of this pass (shadows the level-0 edge)
        const uint32_t obx = g.occ_bx[lev], oby = g.occ_by[lev], obz = g.occ_bz[lev];
        const unsigned long long* occ0 = g.occ + g.occ_off[lev];
        // ---- the lane's cube, in voxels of that level, clipped to the layer's box ----------------------
        uint32_t cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0;
        if (want)
        {
            cx0 = cell_fine(lox, g.ox, g.inv_hf) >> sh, cx1 = cell_fine(hix, g.ox, g.inv_hf) >> sh;
            cy0 = cell_fine(loy, g.oy, g.inv_hf) >> sh, cy1 = cell_fine(hiy, g.oy, g.inv_hf) >> sh;
            cz0 = cell_fine(loz, g.oz, g.inv_hf) >> sh, cz1 = cell_fine(hiz, g.oz, g.inv_hf) >> sh;
            // (r <= 1.7 voxels of the level -> at most 5 voxels per axis; a clipped or oddly aligned one is cut off
            //  by the window test below and waits / leaves like any lane outside the window)
        }
        const unsigned long long wmask = __ballot(want);
        bool go = false;  // this lane's cube lies inside the window of this pass
        if (wmask)        // uniform
        {
            // ---- A: the window: up to 3 x 3 x 3 bricks holding the first open lane's cube, started at the
            //      lowest brick any open lane needs where that still holds the first one ------------------
            const int      seed = __ffsll((long long)wmask) - 1;
            const uint32_t b0x = cx0 >> 2, b0y = cy0 >> 2, b0z = cz0 >> 2, b1x = cx1 >> 2, b1y = cy1 >> 2, b1z = cz1 >> 2;
            const uint32_t s0x = (uint32_t)__builtin_amdgcn_readlane((int)b0x, seed), s1x = (uint32_t)__builtin_amdgcn_readlane((int)b1x, seed);
            const uint32_t s0y = (uint32_t)__builtin_amdgcn_readlane((int)b0y, seed), s1y = (uint32_t)__builtin_amdgcn_readlane((int)b1y, seed);
            const uint32_t s0z = (uint32_t)__builtin_amdgcn_readlane((int)b0z, seed), s1z = (uint32_t)__builtin_amdgcn_readlane((int)b1z, seed);
            constexpr uint32_t NB = NW_WIN;  // bricks per axis at most (NB^3 <= 64: one occupancy word per lane)
            const uint32_t wx = min(max(wave_min_u32(want ? b0x : 0xFFFFFFFFu), s1x >= NB - 1u ? s1x - (NB - 1u) : 0u), s0x);
            const uint32_t wy = min(max(wave_min_u32(want ? b0y : 0xFFFFFFFFu), s1y >= NB - 1u ? s1y - (NB - 1u) : 0u), s0y);
            const uint32_t wz = min(max(wave_min_u32(want ? b0z : 0xFFFFFFFFu), s1z >= NB - 1u ? s1z - (NB - 1u) : 0u), s0z);
            const uint32_t nbx = min(NB, max(wave_max_u32(want ? b1x : 0u), s1x) - wx + 1u);
            const uint32_t nby = min(NB, max(wave_max_u32(want ? b1y : 0u), s1y) - wy + 1u);
            const uint32_t nbz = min(NB, max(wave_max_u32(want ? b1z : 0u), s1z) - wz + 1u);
            const uint32_t nbt = nbx * nby * nbz;  // <= 64
            go = want && b0x >= wx && b1x < wx + nbx && b0y >= wy && b1y < wy + nby && b0z >= wz && b1z < wz + nbz;
            // ---- a SPREAD wave (sparse far field: 64 Morton-consecutive queries metres apart, every one with
            //      voxels of its own): some open lane lies outside the window.  Staging shares nothing there and
            //      windows would take one pass per cluster -- every lane with a cube of at most 4 voxels per axis
            //      searches it by itself instead (occupancy mask of its cube from <= 8 brick words, then voxel by
            //      voxel: directory, points 4 loads in flight; nn_lane_kernel's own path).  Wider cubes keep to
            //      the windows of the following passes. ----------------------------------------------------------
            if (!grp && __ballot(want && !go))  // uniform
            {
                dbg_flags |= 1u;
                const bool lite = want && (cx1 - cx0) < 4u && (cy1 - cy0) < 4u && (cz1 - cz0) < 4u;
                if (INSTR) st_wait += (uint32_t)__popcll(__ballot(lite));
                if (lite)
                {
                    unsigned long long m = cube_occ_mask(g, cx0, cy0, cz0, cx1, cy1, cz1);
                    uint32_t           p = 0, pe = 0;
                    for (;;)
                    {
                        while (p >= pe && m != 0ull)
                        {
                            const uint32_t bit = (uint32_t)__ffsll((long long)m) - 1u;
                            m &= m - 1ull;
                            uint32_t s0 = 0, e0 = 0;
                            if (voxel_range(g, 0u, cx0 + (bit & 3u), cy0 + ((bit >> 2) & 3u), cz0 + (bit >> 4), s0, e0, true)) p = s0, pe = e0;
                        }
                        if (p >= pe) break;
                        float4 c4[4];
#pragma unroll
                        for (int j = 0; j < 4; j++)
                        {
                            c4[j] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
                            if (p + j < pe) c4[j] = g.pts[p + j];
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++)
                        {
                            const float    d  = dist2(qx, qy, qz, c4[j].x, c4[j].y, c4[j].z);
                            const uint32_t ci = __float_as_uint(c4[j].w);
                            if (p + j < pe && (d < best_d2 || (d == best_d2 && ci < best_idx)))
                                best_d2 = d, best_idx = ci, best_spos = p + j;
                            if (INSTR && p + j < pe) a.touched[p + j] = 1, st_tests++;
                        }
                        p += 4u;
                    }
                    // its whole cube was examined
                    if (is_final(r, rmax, best_d2, g.slack)) done = true;
                    else r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
                }
                if (clocks)
                {
                    const unsigned long long t = wall_clock64();
                    t_scan += t - tph, tph = t;
                }
                go = go && !lite;  // the window of this pass serves the wider cubes inside it
            }

            // the window's occupancy words: lane k fetches brick k (x fastest)
            unsigned long long word = 0ull;
            if ((uint32_t)lane < nbt)
            {
                const uint32_t kz = (uint32_t)lane / (nbx * nby), kr = (uint32_t)lane - kz * nbx * nby;
                const uint32_t ky = kr / nbx, kx = kr - ky * nbx;
                const uint32_t Bx = wx + kx, By = wy + ky, Bz = wz + kz;
                if (Bx < obx && By < oby && Bz < obz) word = occ0[((size_t)Bz * oby + By) * obx + Bx];
            }
            // the lane's cube as 4-bit masks per brick column of the window, spread over a brick's 64 voxels
            // (kept packed -- 4 bits per brick column and axis -- and spread per brick: 2 registers instead of 24)
            uint32_t ax = 0u, ay = 0u, az = 0u;
            if (go)
            {
#pragma unroll
                for (int i = 0; i < (int)NB; i++)
                {
                    ax |= axis_mask_in(wx + (uint32_t)i, cx0, cx1) << (4 * i);
                    ay |= axis_mask_in(wy + (uint32_t)i, cy0, cy1) << (4 * i);
                    az |= axis_mask_in(wz + (uint32_t)i, cz0, cz1) << (4 * i);
                }
            }
            const uint32_t axy = ax | (ay << 16);  // 2 x 16 bits; az on its own
            const uint32_t wlo = (uint32_t)word, whi = (uint32_t)(word >> 32);
            // (the same conservative limit as the other kernels' voxel tests: radius + rounding slack, and the running best)
            const float prune = r + 4.f * g.slack;
            const float lim2  = fminf(prune * prune, voxel_limit(best_d2, g.slack));
            const bool  refine = __ballot(go && r > hs0) != 0ull;
            // ---- matrix-pipe prefilter (MP2P_HIP_TUNE mfma_scan, default on): d2 of 32 staged points x 32 queries by
            //      three v_mfma_f32_32x32x2_f32 on coordinates centred on the box of the window's open cubes,
            //        S = -2 c'.q' + |c'|^2 + |q'|^2,  A = [c'x c'y | c'z |c'|^2 | 1 0],  B = [-2q'x -2q'y | -2q'z 1 | |q'|^2 0]
            //      (rows = points, columns = queries; lane l holds column l & 31 and 16 of the 32 rows), exactly as
            //      nn_tile_kernel's: |S - exact d2| <= mtol for every query in the box and every point in the box grown
            //      by one voxel (tests/test_prefilter_bound.py), so a point with S > best + mtol cannot beat or tie the
            //      best; the others are recomputed in the exact FMA-free sequence.  The wave's 64 queries are two
            //      such tiles: in tile t the lanes of half t work for their own query, the other half for its partner's
            //      (lane ^ 32) -- a FOREIGN running best that is merged back after the chunk.
            const bool  hb = lane >= 32;
            float       ocx = 0.f, ocy = 0.f, ocz = 0.f, mtol = 0.f;
            float       bq0[2] = {0.f, 0.f}, bq1[2] = {0.f, 0.f}, bq2[2] = {0.f, 0.f};  // B operands of the two tiles
            float       fqx = 0.f, fqy = 0.f, fqz = 0.f;                                  // the partner's query
            bool        fgo = false;
            if (MF)
            {
                const float blx = wave_min_nn(go ? qx - r : INFINITY), bly = wave_min_nn(go ? qy - r : INFINITY), blz = wave_min_nn(go ? qz - r : INFINITY);
                const float bhx = wave_max_nn(go ? qx + r : -INFINITY), bhy = wave_max_nn(go ? qy + r : -INFINITY), bhz = wave_max_nn(go ? qz + r : -INFINITY);
                ocx = 0.5f * (blx + bhx), ocy = 0.5f * (bly + bhy), ocz = 0.5f * (blz + bhz);
                const float hx = 0.5f * (bhx - blx) + hs, hy = 0.5f * (bhy - bly) + hs, hz = 0.5f * (bhz - blz) + hs;
                mtol = (hx * hx + hy * hy + hz * hz) * (1.0f / 32768.0f);
                fqx = __shfl_xor(qx, 32, 64), fqy = __shfl_xor(qy, 32, 64), fqz = __shfl_xor(qz, 32, 64);
                fgo = __shfl_xor(go ? 1 : 0, 32, 64) != 0;
#pragma unroll
                for (int t = 0; t < 2; t++)
                {
                    const bool  own = hb == (t == 1);  // this lane's own query sits in tile t
                    const float cx = (own ? qx : fqx) - ocx, cy = (own ? qy : fqy) - ocy, cz = (own ? qz : fqz) - ocz;
                    bq0[t] = -2.0f * (hb ? cy : cx);
                    bq1[t] = hb ? 1.0f : -2.0f * cz;
                    bq2[t] = hb ? 0.0f : (cx * cx + cy * cy + cz * cz);
                }
            }
            if (clocks)
            {
                const unsigned long long t = wall_clock64();
                t_ins += t - tph, tph = t;
            }

            // ---- bricks -> voxel list, in chunks of at most NW_NL voxels; each chunk is resolved, staged
            //      and scanned before the next one is listed ------------------------------------------
            uint32_t n_u = 0;
            for (uint32_t k = 0; k <= nbt; k++)
            {
                if (k < nbt)
                {
                    const unsigned long long W = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)whi, (int)k) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readlane((int)wlo, (int)k);
                    if (W == 0ull) continue;  // uniform: an empty brick
                    const uint32_t kz = k / (nbx * nby), kr = k - kz * nbx * nby;
                    const uint32_t ky = kr / nbx, kx = kr - ky * nbx;
                    const unsigned long long sx = spread_x((axy >> (4u * kx)) & 15u);
                    const unsigned long long sy = spread_y((axy >> (16u + 4u * ky)) & 15u);
                    const unsigned long long sz = spread_z((az >> (4u * kz)) & 15u);
                    unsigned long long N = W & sx & sy & sz;  // occupied voxels of this brick inside the lane's cube
                    // a wide cube holds ~2-3x the points its ball reaches (a wall 0.9 m away: the cube cuts 1.8 x 1.8 m out
                    // of it, the ball touches a 0.6 m disc): the voxels the BALL cannot reach are dropped, row by row
                    // (x-interval of every (y, z) row from the per-axis squared distances).  Only when some lane's radius
                    // exceeds a voxel: narrow cubes gain nothing
                    if (refine)  // uniform
                    {
                        unsigned long long keep = 0ull;
                        if (N != 0ull)
                        {
                            // the brick's x-extent as ONE slab (its nearest face), the 4 y- and 4 z-slabs exactly: a (y, z)
                            // row is kept when the ball reaches it at all (the per-voxel x test costs 4x the instructions
                            // and the all-pairs scan it saves is cheaper than that)
                            const float bx0 = g.ox + (float)((wx + kx) * 4u) * hs;
                            const float dxb = fmaxf(0.f, fmaxf(bx0 - qx, qx - (bx0 + 4.f * hs)));
                            const float lim_yz = lim2 - dxb * dxb * 0.999999f;
                            float ddy[4], ddz[4];
#pragma unroll
                            for (int i = 0; i < 4; i++)
                            {
                                const float y0 = g.oy + (float)((wy + ky) * 4u + (uint32_t)i) * hs, z0 = g.oz + (float)((wz + kz) * 4u + (uint32_t)i) * hs;
                                const float dy = fmaxf(0.f, fmaxf(y0 - qy, qy - (y0 + hs))), dz = fmaxf(0.f, fmaxf(z0 - qz, qz - (z0 + hs)));
                                ddy[i] = dy * dy, ddz[i] = dz * dz;
                            }
                            uint32_t rows = 0u;  // bit zz * 4 + yy
#pragma unroll
                            for (int zz = 0; zz < 4; zz++)
#pragma unroll
                                for (int yy = 0; yy < 4; yy++) rows |= (ddy[yy] + ddz[zz] <= lim_yz) ? (1u << (zz * 4 + yy)) : 0u;
                            // 16 row bits -> 64 voxel bits (every row bit times 0xF at bit 4 * row)
                            const uint32_t r_lo = rows & 0xFFu, r_hi = rows >> 8;
                            const uint32_t e_lo = nibble_spread8(r_lo), e_hi = nibble_spread8(r_hi);
                            keep = ((unsigned long long)e_hi << 32) | e_lo;
                        }
                        N &= keep;
                    }
                    const unsigned long long U = wave_or_u64(N);  // voxels of this brick some lane needs
                    if (U == 0ull) continue;  // uniform
                    if (INSTR) st_bricks++;
                    if ((U >> lane) & 1ull)
                    {
                        // voxel code: window-relative voxel coordinates, 4 bits each (bit = z * 16 + y * 4 + x)
                        const uint32_t vx = kx * 4u + ((uint32_t)lane & 3u), vy = ky * 4u + (((uint32_t)lane >> 2) & 3u),
                                       vz = kz * 4u + ((uint32_t)lane >> 4);
                        s_vox[n_u + (uint32_t)__popcll(U & lane_lt)] = 1u | (vx << 4) | (vy << 8) | (vz << 12);
                    }
                    n_u += (uint32_t)__popcll(U);
                    if (n_u <= (uint32_t)(NW_NL - 64) && k + 1u < nbt) continue;  // room for another brick
                }
                if (n_u == 0u) continue;
                // ---- B: resolve the chunk (4 voxels per lane, loads independent), drop the voxels without
                //      points (the bitmap said otherwise only for a stale map), prefix the counts --------
                __syncthreads();
                uint32_t T = 0, n_l = 0;
                {
                    const uint4 k4 = *reinterpret_cast<const uint4*>(&s_vox[4 * lane]);
                    uint32_t    kk[4] = {k4.x, k4.y, k4.z, k4.w};
                    uint32_t    st[4] = {0u, 0u, 0u, 0u}, cn[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (4u * (uint32_t)lane + (uint32_t)j >= n_u) kk[j] = 0u;  // stale codes of an earlier chunk
                    const unsigned long long doff = g.dir_off[lev];
                    if (doff != DIR_NONE)  // uniform
                    {
                        const uint32_t nx = obx * 4u, ny = oby * 4u, nz = obz * 4u;
                        size_t         ix[4];
                        bool           in[4];
#pragma unroll
                        for (int j = 0; j < 4; j++)
                        {
                            const uint32_t cx = wx * 4u + ((kk[j] >> 4) & 15u), cy = wy * 4u + ((kk[j] >> 8) & 15u),
                                           cz = wz * 4u + ((kk[j] >> 12) & 15u);
                            in[j] = kk[j] != 0u && cx < nx && cy < ny && cz < nz;
                            ix[j] = in[j] ? ((size_t)cz * ny + cy) * nx + cx : 0;
                        }
                        uint2 e[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) e[j] = g.dir[doff + ix[j]];
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (in[j] && e[j].y > e[j].x) st[j] = e[j].x, cn[j] = e[j].y - e[j].x;
                    }
                    else
                    {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (kk[j] != 0u)
                            {
                                uint32_t s0 = 0, e0 = 0;
                                if (voxel_range(g, lev, wx * 4u + ((kk[j] >> 4) & 15u), wy * 4u + ((kk[j] >> 8) & 15u),
                                                wz * 4u + ((kk[j] >> 12) & 15u), s0, e0, true))
                                    st[j] = s0, cn[j] = e0 - s0;
                            }
                    }
                    uint32_t used = 0, tot = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) used += cn[j] ? 1u : 0u, tot += cn[j];
                    const uint32_t iu = wave_incl_scan(used, lane), it = wave_incl_scan(tot, lane);
                    n_l = (uint32_t)__builtin_amdgcn_readlane((int)iu, 63);
                    T   = (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
                    uint32_t li = iu - used, off = it - tot;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (cn[j])
                        {
                            s_list[li] = make_uint2(st[j], off);
                            li++, off += cn[j];
                        }
                }
                n_u = 0;
                __syncthreads();
                // a wave with much to scan (wide radii) is on the kernel's critical path -- the other ~4 000 resident
                // waves finish in a tenth of its time: it takes the SIMD's issue slots first from here on
                if (T > 2u * NW_CAP) __builtin_amdgcn_s_setprio(3);
                dbg_nu = max(dbg_nu, n_l), dbg_rounds += (T + NW_CAP - 1) / NW_CAP;
                if (INSTR) st_listed += n_l, st_T += T;
                if (clocks)
                {
                    const unsigned long long t = wall_clock64();
                    t_dir += t - tph, tph = t;
                }

                // ---- C + D per round of NW_CAP staged points.  The loads of round n + 1 are issued before round n
                //      is scanned (a chunk of a wide-radius wave takes up to 20+ rounds: their round trips were
                //      most of its time).  Every lane tests every staged point: per-lane WALKS over the round's voxels
                //      (only those the lane's ball reaches: half the tests) were measured 3.5x slower -- divergent refill
                //      loops and dependent LDS reads (commit "wave kernel v2"). ---------------------------------
                uint32_t   src[4];
                float4     c4[4];
                // which listed voxel does a staged slot belong to: every voxel drops its id at its first slot of the
                // round, a prefix-max carries it to the following slots (ids ascend with the offsets); then the loads
                auto stage_issue = [&] __device__(uint32_t base) {
                    const uint32_t m = min((uint32_t)NW_CAP, T - base);
                    *reinterpret_cast<uint4*>(&s_owner[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
                    __syncthreads();
                    for (uint32_t e = (uint32_t)lane; e < n_l; e += 64u)
                    {
                        const uint32_t off = s_list[e].y, nxt = e + 1u < n_l ? s_list[e + 1u].y : T;  // its points: [off, nxt)
                        if (off >= base && off < base + NW_CAP) s_owner[off - base] = e + 1u;
                        else if (off < base && nxt > base) s_owner[0] = e + 1u;
                    }
                    __syncthreads();
                    const uint4    o4 = *reinterpret_cast<const uint4*>(&s_owner[4 * lane]);
                    const uint32_t p0 = o4.x, p1 = max(p0, o4.y), p2 = max(p1, o4.z), p3 = max(p2, o4.w);
                    const uint32_t in = wave_incl_max(p3, lane);
                    uint32_t       ex = __shfl_up(in, 1, 64);
                    if (lane == 0) ex = 0u;
                    const uint32_t ow[4] = {max(ex, p0), max(ex, p1), max(ex, p2), max(ex, p3)};
                    const uint32_t t0    = 4u * (uint32_t)lane;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        src[j] = NONE_U32;
                        if (t0 + j < m)
                        {
                            const uint2 L = s_list[ow[j] - 1u];
                            src[j]        = L.x + (base + t0 + j - L.y);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        c4[j] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
                        if (t0 + j < m) c4[j] = g.pts[src[j]];
                    }
                };
                if (T) stage_issue(0u);
                // group pass: lane l works for the (l & 15)-th member of the group on slice l >> 4 of the staged points
                float    gqx = qx, gqy = qy, gqz = qz, gb_d2 = best_d2;
                uint32_t gb_idx = best_idx, gb_spos = best_spos;
                bool     gok = false;
                if (grp)
                {
                    const unsigned long long gomask = __ballot(go);
                    const uint32_t           rank   = (uint32_t)__popcll(gomask & lane_lt);
                    if (go) s_grp[rank] = (uint32_t)lane;
                    __syncthreads();
                    const uint32_t mem = ((uint32_t)lane & 15u) < (uint32_t)__popcll(gomask) ? s_grp[lane & 15] : 0u;
                    gok  = ((uint32_t)lane & 15u) < (uint32_t)__popcll(gomask);
                    gqx = __shfl(qx, (int)mem, 64), gqy = __shfl(qy, (int)mem, 64), gqz = __shfl(qz, (int)mem, 64);
                    gb_d2 = __shfl(best_d2, (int)mem, 64), gb_idx = __shfl(best_idx, (int)mem, 64), gb_spos = __shfl(best_spos, (int)mem, 64);
                }
                // the partner's running best (the query this lane works for in the other tile)
                float    fb_d2 = INFINITY;
                uint32_t fb_idx = NONE_U32, fb_spos = NONE_U32;
                if (MF) fb_d2 = __shfl_xor(best_d2, 32, 64), fb_idx = __shfl_xor(best_idx, 32, 64), fb_spos = __shfl_xor(best_spos, 32, 64);
                for (uint32_t base = 0; base < T; base += NW_CAP)
                {
                    const uint32_t m = min((uint32_t)NW_CAP, T - base);
                    if (INSTR) st_rounds++;
                    {
                        const uint32_t t0 = 4u * (uint32_t)lane;
                        *reinterpret_cast<float4*>(&s_x[t0]) = make_float4(c4[0].x, c4[1].x, c4[2].x, c4[3].x);
                        *reinterpret_cast<float4*>(&s_y[t0]) = make_float4(c4[0].y, c4[1].y, c4[2].y, c4[3].y);
                        *reinterpret_cast<float4*>(&s_z[t0]) = make_float4(c4[0].z, c4[1].z, c4[2].z, c4[3].z);
                        *reinterpret_cast<uint4*>(&s_idx[t0]) = make_uint4(__float_as_uint(c4[0].w), __float_as_uint(c4[1].w),
                                                                           __float_as_uint(c4[2].w), __float_as_uint(c4[3].w));
                        *reinterpret_cast<uint4*>(&s_spos[t0]) = make_uint4(src[0], src[1], src[2], src[3]);
                        if (INSTR)
                        {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (t0 + j < m) a.touched[src[j]] = 1;
                        }
                    }
                    __syncthreads();
                    if (base + NW_CAP < T) stage_issue(base + NW_CAP);  // in flight while this round is scanned
                    if (clocks)
                    {
                        const unsigned long long t = wall_clock64();
                        t_stage += t - tph, tph = t;
                    }
                    if (grp)  // uniform
                    {
                        // ---- D (groups): 16 queries x 4 slices; the lane tests the 8-point blocks of its slice ----
                        const v2f      gx2 = {gqx, gqx}, gy2 = {gqy, gqy}, gz2 = {gqz, gqz};
                        const uint32_t m_pad = (m + 7u) & ~7u;
                        for (uint32_t jb = ((uint32_t)lane >> 4) * 8u; jb < m_pad; jb += 32u)
                        {
                            const float4 xa = *reinterpret_cast<const float4*>(&s_x[jb]);
                            const float4 xb = *reinterpret_cast<const float4*>(&s_x[jb + 4]);
                            const float4 ya = *reinterpret_cast<const float4*>(&s_y[jb]);
                            const float4 yb = *reinterpret_cast<const float4*>(&s_y[jb + 4]);
                            const float4 za = *reinterpret_cast<const float4*>(&s_z[jb]);
                            const float4 zb = *reinterpret_cast<const float4*>(&s_z[jb + 4]);
                            const v2f d01 = dist2_pk(gx2, gy2, gz2, v2f{xa.x, xa.y}, v2f{ya.x, ya.y}, v2f{za.x, za.y});
                            const v2f d23 = dist2_pk(gx2, gy2, gz2, v2f{xa.z, xa.w}, v2f{ya.z, ya.w}, v2f{za.z, za.w});
                            const v2f d45 = dist2_pk(gx2, gy2, gz2, v2f{xb.x, xb.y}, v2f{yb.x, yb.y}, v2f{zb.x, zb.y});
                            const v2f d67 = dist2_pk(gx2, gy2, gz2, v2f{xb.z, xb.w}, v2f{yb.z, yb.w}, v2f{zb.z, zb.w});
                            const float d[8] = {d01.x, d01.y, d23.x, d23.y, d45.x, d45.y, d67.x, d67.y};
                            const float mn = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])),
                                                   fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
                            if (gok && mn <= gb_d2)
                            {
#pragma unroll
                                for (int j = 0; j < 8; j++)
                                {
                                    if (d[j] <= gb_d2)
                                    {
                                        const uint32_t ci = s_idx[jb + j];
                                        if (d[j] < gb_d2 || ci < gb_idx) gb_d2 = d[j], gb_idx = ci, gb_spos = s_spos[jb + j];
                                    }
                                }
                            }
                        }
                        if (INSTR) st_tests += m * 16u;
                    }
                    else if (MF)
                    {
                        // ---- D (matrix pipe): 32 staged points x 2 tiles of 32 queries per step ------------------
                        const uint32_t m_pad = (m + 31u) & ~31u;  // slots beyond m hold x = inf: S = inf or NaN, never <= lim
                        for (uint32_t blk = 0; blk < m_pad; blk += 32u)
                        {
                            const uint32_t c  = blk + ((uint32_t)lane & 31u);
                            const float    px = s_x[c], py = s_y[c], pz = s_z[c];
                            const float    ex = px - ocx, ey = py - ocy, ez = pz - ocz;
                            const float    a0v = hb ? ey : ex;
                            const float    a1v = hb ? (ex * ex + ey * ey + ez * ez) : ez;
                            const float    a2v = hb ? 0.0f : 1.0f;
#pragma unroll
                            for (int t = 0; t < 2; t++)
                            {
                                const bool own = hb == (t == 1);
                                f32x16     acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, bq0[t], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, bq1[t], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2v, bq2[t], acc, 0, 0, 0);
                                const float m0 = fminf(fminf(acc[0], acc[1]), fminf(acc[2], acc[3]));
                                const float m1 = fminf(fminf(acc[4], acc[5]), fminf(acc[6], acc[7]));
                                const float m2 = fminf(fminf(acc[8], acc[9]), fminf(acc[10], acc[11]));
                                const float m3 = fminf(fminf(acc[12], acc[13]), fminf(acc[14], acc[15]));
                                const float mn = fminf(fminf(m0, m1), fminf(m2, m3));
                                // the query this lane works for in tile t, and its running best
                                const float tqx = own ? qx : fqx, tqy = own ? qy : fqy, tqz = own ? qz : fqz;
                                float       bd  = own ? best_d2 : fb_d2;
                                uint32_t    bi = own ? best_idx : fb_idx, bs = own ? best_spos : fb_spos;
                                float       lim = bd * 1.000001f + mtol;
                                if ((own ? go : fgo) && mn <= lim)
                                {
#pragma unroll
                                    for (int rr = 0; rr < 16; rr++)
                                    {
                                        if (acc[rr] <= lim)
                                        {
                                            // row of register rr (C/D layout of the 32x32 MFMAs)
                                            const uint32_t j  = blk + (uint32_t)((rr & 3) + 8 * (rr >> 2)) + (hb ? 4u : 0u);
                                            const float    dd = dist2(tqx, tqy, tqz, s_x[j], s_y[j], s_z[j]);
                                            if (dd <= bd)
                                            {
                                                const uint32_t ci = s_idx[j];
                                                if (dd < bd || ci < bi) bd = dd, bi = ci, bs = s_spos[j], lim = bd * 1.000001f + mtol;
                                            }
                                        }
                                    }
                                    if (own) best_d2 = bd, best_idx = bi, best_spos = bs;
                                    else fb_d2 = bd, fb_idx = bi, fb_spos = bs;
                                }
                            }
                        }
                        if (INSTR) st_tests += m * (uint32_t)__popcll(__ballot(go));
                    }
                    else
                    {
                        // ---- D (all pairs): every lane tests every staged point, 8 per step on the packed-fp32 path
                        //      (same roundings as dist2()); the update is rare.  Every staged point is a map point,
                        //      so open lanes outside the window collect upper bounds too -------------------------
                        const uint32_t m_pad = (m + 7u) & ~7u;
                        for (uint32_t jb = 0; jb < m_pad; jb += 8u)
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
                            if (part && mn <= best_d2)
                            {
#pragma unroll
                                for (int j = 0; j < 8; j++)
                                {
                                    if (d[j] <= best_d2)
                                    {
                                        const uint32_t ci = s_idx[jb + j];
                                        if (d[j] < best_d2 || ci < best_idx)
                                        {
                                            best_d2   = d[j];
                                            best_idx  = ci;
                                            best_spos = s_spos[jb + j];
                                        }
                                    }
                                }
                            }
                        }
                        if (INSTR) st_tests += m * (uint32_t)__popcll(__ballot(part));
                    }
                    __syncthreads();  // the next round overwrites the staged points
                    if (clocks)
                    {
                        const unsigned long long t = wall_clock64();
                        t_scan += t - tph, tph = t;
                    }
                }
                if (grp)
                {
                    // the 4 slices of every member, then back to the member's own lane
#pragma unroll
                    for (int off = 16; off < 64; off <<= 1)
                    {
                        const float    od = __shfl_xor(gb_d2, off, 64);
                        const uint32_t oi = __shfl_xor(gb_idx, off, 64), os = __shfl_xor(gb_spos, off, 64);
                        if (od < gb_d2 || (od == gb_d2 && oi < gb_idx)) gb_d2 = od, gb_idx = oi, gb_spos = os;
                    }
                    const uint32_t rank = (uint32_t)__popcll(__ballot(go) & lane_lt);  // this member's workers: lanes rank, rank + 16, ..
                    const float    md = __shfl(gb_d2, (int)rank, 64);
                    const uint32_t mi = __shfl(gb_idx, (int)rank, 64), ms = __shfl(gb_spos, (int)rank, 64);
                    if (go && (md < best_d2 || (md == best_d2 && mi < best_idx))) best_d2 = md, best_idx = mi, best_spos = ms;
                }
                else if (MF)
                {  // what the partner found for this lane's query
                    const float    od = __shfl_xor(fb_d2, 32, 64);
                    const uint32_t oi = __shfl_xor(fb_idx, 32, 64), os = __shfl_xor(fb_spos, 32, 64);
                    if (od < best_d2 || (od == best_d2 && oi < best_idx)) best_d2 = od, best_idx = oi, best_spos = os;
                }
            }
        }
        // ---- conclude, or grow: lanes whose whole cube was inside the window (or empty) --------------
        if (part && (go || empty))
        {
            if (is_final(r, rmax, best_d2, g.slack)) done = true;
            else r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
        }
    }
    // whatever is still open (many windows, many growth steps) is handed on with its state
    {
        const bool               left  = !done && !deferred;
        const unsigned long long lmask = __ballot(left);
        if (lmask)
        {
            st_defer += push_lanes(a, 1, wv / a.seg_waves, left, lmask, lane, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (left) deferred = true;
        }
    }

    // ================= records + claims (Morton order of the local layer)
    if (lane == 0)
    {
        float* o = a.tile_bbox + (size_t)wv * 6;
        o[0] = wbx0, o[1] = wby0, o[2] = wbz0, o[3] = wbx1, o[4] = wby1, o[5] = wbz1;
    }
    __syncthreads();  // the claim table lives in the staging area
    emit_wave(a, s_claim, lane, valid && !deferred, qi, orig, active, thr, best_d2, best_idx, best_spos,
              lb2_out >= 0.f ? lb2_out : fminf(best_d2, thr));

    if (a.timeline && lane == 0)
    {
        // 8 words per wave: start tick (low 40 bits; above: widest voxel list (8 bits), staging rounds (8), passes (4),
        // flags), end tick, ticks of {prologue, window, directory, staging, tests, records + claims}
        const unsigned long long info = (unsigned long long)min(dbg_nu, 255u) | ((unsigned long long)min(dbg_rounds, 255u) << 8) |
                                        ((unsigned long long)min(st_pass, 15u) << 16) | ((unsigned long long)dbg_flags << 20);
        const unsigned long long t_end = wall_clock64();
        unsigned long long* o = a.timeline + 8 * (size_t)blockIdx.x;
        o[0] = (tl0 & 0xFFFFFFFFFFull) | (info << 40), o[1] = t_end & 0xFFFFFFFFFFull;
        o[2] = t_pro, o[3] = t_ins, o[4] = t_dir, o[5] = t_stage, o[6] = t_scan, o[7] = t_end - tph;
    }
    if (INSTR)
    {
        const unsigned long long t_end = wall_clock64();
        const uint32_t waits = st_wait;
        if (lane == 0)
        {
            atomicAdd(&a.counters[0], 1ull);
            atomicAdd(&a.counters[1], (unsigned long long)st_pass);
            atomicAdd(&a.counters[2], (unsigned long long)st_listed);
            atomicAdd(&a.counters[3], (unsigned long long)st_T);
            if (st_pass > 1) atomicAdd(&a.counters[4], 1ull);
            atomicMax(&a.counters[5], (unsigned long long)st_T);
            atomicMax(&a.counters[6], (unsigned long long)st_pass);
            const unsigned long long dt = t_end - tl0;
            atomicAdd(&a.counters[7], dt);
            atomicMax(&a.counters[8], dt);
            atomicAdd(&a.counters[9], (unsigned long long)st_defer);
            int b = 63 - __clzll((long long)(dt | 1ull));  // log2 bins of 100 MHz ticks
            if (b > 23) b = 23;
            atomicAdd(&a.counters[16 + b], 1ull);
            atomicAdd(&a.counters[47], (unsigned long long)n_search);
            atomicAdd(&a.counters[48], (unsigned long long)n_skip);
            atomicAdd(&a.counters[NWC_LANE_TESTS], (unsigned long long)st_tests);
            atomicAdd(&a.counters[NWC_MAXLANE], (unsigned long long)st_T);
            atomicAdd(&a.counters[NWC_INSERTS], (unsigned long long)st_bricks);
            atomicAdd(&a.counters[NWC_OVF], (unsigned long long)waits);
            atomicAdd(&a.counters[NWC_ROUNDS], (unsigned long long)st_rounds);
            atomicAdd(&a.counters[NWC_LISTED], (unsigned long long)st_listed);
            atomicAdd(&a.counters[NWC_TOOBIG], (unsigned long long)st_toobig);
            atomicAdd(&a.counters[NWC_T_PRO], t_pro);
            atomicAdd(&a.counters[NWC_T_INS], t_ins);
            atomicAdd(&a.counters[NWC_T_DIR], t_dir);
            atomicAdd(&a.counters[NWC_T_STAGE], t_stage);
            atomicAdd(&a.counters[NWC_T_SCAN], t_scan);
            atomicAdd(&a.counters[NWC_T_EMIT], t_end - tph);
        }
    }
}

}  // namespace mp2p
