// nn_wave.hip -- K1 + K3, round 3: ONE kernel for transform, warm start and the exact nearest-neighbour
// search of (nearly) every query; replaces nn_lane_kernel + nn_tile_kernel of nn_query.hip on maps
// that carry a level-0 occupancy bitmap (the default build).
//
// Reference loop replaced: Matcher_Points_DistanceThreshold.cpp:214-265 (+ transform_local_to_global,
// Matcher_Points_Base.cpp:183-249).
//
// Why another structure (DESIGN.md section 4 has the numbers): the tile kernel shared ONE box per 32
// queries -- its candidates are the box's points, whatever each query's own radius -- and paid a
// chain of ~6 dependent round trips per 32 queries.  On the bench chains 80 % of the queries have a
// previous nearest neighbour 2..9 cm away while a few per tile are 0.3..1 m off: the box is sized by the
// worst of them.  Here a wave owns 64 Morton-consecutive queries (one per lane) and
//   A. every lane enumerates the OCCUPIED level-0 voxels its own ball [q, r] touches -- from the 4x4x4
//      occupancy bricks (one 8-byte word per brick, <= 8 words per 4x4x4 sub-cube of the ball's cube) --
//      and inserts them into a hash SET in LDS (ds_cmpst: duplicates of neighbouring queries collapse);
//   B. the set's voxels are resolved through the directory (one 8-byte load per voxel, 4 per lane, all
//      independent) and listed with their point ranges;
//   C. the points of the listed voxels are staged into LDS with coalesced 16-byte loads;
//   D. every lane walks the list, keeps the voxels its ball (shrinking with its running best) reaches,
//      and tests their staged points in the exact FMA-free sequence.
// The radius of a warm query is the distance to its previous nearest neighbour (certain to conclude,
// however small), so one pass finishes it.  Queries whose cube exceeds 8 voxels per axis, or whose radius
// outgrows r_defer, go to nn_single_kernel (a whole wave per query) as before.
// Results are bit-identical to the other kernels' (same arg-min rule, same finality rule).
#include "device_utils.hpp"

namespace mp2p
{
constexpr int NW_HS       = 256;  // slots of the voxel set = capacity of the voxel list
constexpr int NW_CAP      = 256;  // staged points per round
constexpr int NW_MAXPROBE = 16;
constexpr int NW_MAXPASS  = 4;
constexpr uint32_t NW_MAXSPAN = 8;  // widest cube (level-0 voxels per axis) a lane enumerates

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
    NWC_LANE_TESTS = 50,  // candidates tested, summed over lanes
    NWC_MAXLANE    = 51,  // ... and what the waves waited for: the longest lane of each walk, in candidates
    NWC_INSERTS    = 52,  // voxel insertions attempted (occupied voxels inside a ball)
    NWC_OVF        = 53,  // lanes that could not place a voxel (set full) or sat too far from the wave's origin
    NWC_ROUNDS     = 54,  // staging rounds
    NWC_LISTED     = 55,  // voxels listed (sum over passes)
    NWC_TOOBIG     = 56,  // queries handed on because their cube exceeds NW_MAXSPAN voxels per axis
    NWC_T_PRO      = 57,  // 100 MHz ticks per phase, summed over waves: prologue (loads, transform, warm start)
    NWC_T_INS      = 58,  //   voxel enumeration + set insertion
    NWC_T_DIR      = 59,  //   directory resolve + list
    NWC_T_STAGE    = 60,  //   staging
    NWC_T_SCAN     = 61,  //   walks
    NWC_T_EMIT     = 62,  //   hand-over + records + claims
};

template <bool INSTR>
__global__ __launch_bounds__(64) void nn_wave_kernel(const NNArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_key[NW_HS];
    __shared__ __attribute__((aligned(16))) uint4    s_list[NW_HS];  // {key, first sorted position, first staged slot, count}
    __shared__ __attribute__((aligned(16))) float4   s_pts[NW_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_owner[NW_CAP];
    static_assert(NN_CLAIM_SLOTS * sizeof(unsigned long long) <= NW_CAP * sizeof(float4), "claim table fits s_pts");
    unsigned long long* s_claim = reinterpret_cast<unsigned long long*>(s_pts);

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
    const float hs = g.hf * (float)(1u << g.shift0);  // level-0 voxel edge
    uint32_t dbg_nu = 0, dbg_rounds = 0, dbg_flags = 0;  // profiling level 4: packed into the timeline record
    uint32_t st_pass = 0, st_T = 0, st_tests = 0, st_maxlane = 0, st_ins = 0, st_ovf = 0, st_rounds = 0, st_listed = 0,
             st_toobig = 0, st_defer = 0;
    if (INSTR)
    {
        const unsigned long long t = wall_clock64();
        if (lane == 0) atomicAdd(&a.counters[NWC_T_PRO], t - tph);
        tph = t;
    }
    const uint32_t n_search = (uint32_t)__popcll(__ballot(!done));
    const uint32_t n_skip   = (uint32_t)__popcll(__ballot(active && lb2_out >= 0.f));

    // ================= passes
    for (int pass = 0; pass < NW_MAXPASS; pass++)
    {
        // a radius beyond r_defer leaves for the one-query-per-wave kernel (coarser levels, bricks)
        {
            const bool               wide  = !done && !deferred && r > a.r_defer;
            const unsigned long long wmask = __ballot(wide);
            if (wmask)
            {
                st_defer += push_lanes(a, 1, wv / a.seg_waves, wide, wmask, lane, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
                if (wide) deferred = true;
            }
        }
        const bool part = !done && !deferred;
        if (__ballot(part) == 0ull) break;  // uniform
        st_pass++;

        // ---- the lane's cube, in level-0 voxels, clipped to the layer's box ----------------------
        const float lox = fmaxf(qx - r, g.bbmin[0]), loy = fmaxf(qy - r, g.bbmin[1]), loz = fmaxf(qz - r, g.bbmin[2]);
        const float hix = fminf(qx + r, g.bbmax[0]), hiy = fminf(qy + r, g.bbmax[1]), hiz = fminf(qz + r, g.bbmax[2]);
        const bool  empty = !part || (lox > hix) || (loy > hiy) || (loz > hiz);  // nothing to visit at this radius
        uint32_t cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0;
        if (!empty)
        {
            cx0 = cell_fine(lox, g.ox, g.inv_hf) >> g.shift0, cx1 = cell_fine(hix, g.ox, g.inv_hf) >> g.shift0;
            cy0 = cell_fine(loy, g.oy, g.inv_hf) >> g.shift0, cy1 = cell_fine(hiy, g.oy, g.inv_hf) >> g.shift0;
            cz0 = cell_fine(loz, g.oz, g.inv_hf) >> g.shift0, cz1 = cell_fine(hiz, g.oz, g.inv_hf) >> g.shift0;
        }
        const bool toobig = !empty && ((cx1 - cx0) >= NW_MAXSPAN || (cy1 - cy0) >= NW_MAXSPAN || (cz1 - cz0) >= NW_MAXSPAN);
        // voxels are keyed relative to the wave's lowest corner, 10 bits per axis: a lane too far from it
        // (a Morton jump inside the wave) waits for a later pass
        const bool     want = !empty && !toobig;
        const uint32_t wx0 = wave_min_u32(want ? cx0 : 0xFFFFFFFFu), wy0 = wave_min_u32(want ? cy0 : 0xFFFFFFFFu),
                       wz0 = wave_min_u32(want ? cz0 : 0xFFFFFFFFu);
        const bool far = want && ((cx1 - wx0) > 1022u || (cy1 - wy0) > 1022u || (cz1 - wz0) > 1022u);
        bool       ovf = far;  // this lane cannot conclude in this pass
        const bool go  = want && !far;

        // ---- A: the occupied voxels inside the ball go into the set ------------------------------
        *reinterpret_cast<uint4*>(&s_key[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        const float prune  = r + 4.f * g.slack;
        const float prune2 = prune * prune;
        if (go)
        {
            for (uint32_t sz = cz0; sz <= cz1; sz += 4u)
                for (uint32_t sy = cy0; sy <= cy1; sy += 4u)
                    for (uint32_t sx = cx0; sx <= cx1; sx += 4u)
                    {
                        unsigned long long m = cube_occ_mask(g, sx, sy, sz, min(sx + 3u, cx1), min(sy + 3u, cy1), min(sz + 3u, cz1));
                        while (m != 0ull)
                        {
                            const uint32_t bit = (uint32_t)__ffsll((long long)m) - 1u;
                            m &= m - 1ull;
                            const uint32_t cx = sx + (bit & 3u), cy = sy + ((bit >> 2) & 3u), cz = sz + (bit >> 4);
                            const float    md2 = box_dist2(g.ox + (float)cx * hs, g.oy + (float)cy * hs, g.oz + (float)cz * hs, hs,
                                                           qx, qy, qz);
                            if (md2 <= fminf(prune2, voxel_limit(best_d2, g.slack)))
                            {
                                const uint32_t key  = 1u + (((cz - wz0) << 20) | ((cy - wy0) << 10) | (cx - wx0));
                                uint32_t       slot = (key * 0x9E3779B1u) >> 24;  // log2(NW_HS) = 8 bits
                                bool           ok   = false;
                                for (int pr = 0; pr < NW_MAXPROBE; pr++)
                                {
                                    const uint32_t old = atomicCAS(&s_key[slot], 0u, key);
                                    if (old == 0u || old == key)
                                    {
                                        ok = true;
                                        break;
                                    }
                                    slot = (slot + 1u) & (uint32_t)(NW_HS - 1);
                                }
                                if (!ok) ovf = true;
                                if (INSTR) st_ins++;
                            }
                        }
                    }
        }
        __syncthreads();
        if (INSTR)
        {
            const unsigned long long t = wall_clock64();
            t_ins += t - tph, tph = t;
        }

        // ---- B: resolve the set (4 slots per lane, loads independent), list the voxels that hold
        //      points, prefix their point counts --------------------------------------------------
        uint32_t n_u = 0, T = 0;
        {
            const uint4    k4    = *reinterpret_cast<const uint4*>(&s_key[4 * lane]);
            const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
            uint32_t       st[4] = {0u, 0u, 0u, 0u}, cn[4] = {0u, 0u, 0u, 0u};
            const unsigned long long doff = g.dir_off[0];
            if (doff != DIR_NONE)  // uniform
            {
                const uint32_t nx = g.occ_bx[0] * 4u, ny = g.occ_by[0] * 4u, nz = g.occ_bz[0] * 4u;
                size_t         ix[4];
                bool           in[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const uint32_t v  = kk[k] - 1u;
                    const uint32_t cx = wx0 + (v & 1023u), cy = wy0 + ((v >> 10) & 1023u), cz = wz0 + (v >> 20);
                    in[k] = kk[k] != 0u && cx < nx && cy < ny && cz < nz;
                    ix[k] = in[k] ? ((size_t)cz * ny + cy) * nx + cx : 0;
                }
                uint2 e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) e[k] = g.dir[doff + ix[k]];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (in[k] && e[k].y > e[k].x) st[k] = e[k].x, cn[k] = e[k].y - e[k].x;
            }
            else
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (kk[k] != 0u)
                    {
                        const uint32_t v = kk[k] - 1u;
                        uint32_t       s0 = 0, e0 = 0;
                        if (voxel_range(g, 0u, wx0 + (v & 1023u), wy0 + ((v >> 10) & 1023u), wz0 + (v >> 20), s0, e0, true))
                            st[k] = s0, cn[k] = e0 - s0;
                    }
            }
            uint32_t used = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) used += cn[k] ? 1u : 0u, tot += cn[k];
            const uint32_t iu = wave_incl_scan(used, lane), it = wave_incl_scan(tot, lane);
            n_u = (uint32_t)__builtin_amdgcn_readlane((int)iu, 63);
            T   = (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
            uint32_t li = iu - used, off = it - tot;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (cn[k])
                {
                    s_list[li] = make_uint4(kk[k], st[k], off, cn[k]);
                    li++, off += cn[k];
                }
        }
        __syncthreads();
        dbg_nu = max(dbg_nu, n_u), dbg_rounds += (T + NW_CAP - 1) / NW_CAP;
        dbg_flags |= (__ballot(part && ovf) ? 1u : 0u) | (__ballot(part && toobig) ? 2u : 0u);
        if (INSTR)
        {
            st_listed += n_u, st_T += T;
            const unsigned long long t = wall_clock64();
            t_dir += t - tph, tph = t;
        }

        // ---- C + D per round of NW_CAP staged points ---------------------------------------------
        for (uint32_t base = 0; base < T; base += NW_CAP)
        {
            const uint32_t m = min((uint32_t)NW_CAP, T - base);
            if (INSTR) st_rounds++;
            // which listed voxel does a staged slot belong to: every voxel drops its id at its first slot of
            // the round, a prefix-max carries it to the following slots (ids ascend with the offsets)
            *reinterpret_cast<uint4*>(&s_owner[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            for (uint32_t e = (uint32_t)lane; e < n_u; e += 64u)
            {
                const uint4 L = s_list[e];
                if (L.z >= base && L.z < base + NW_CAP) s_owner[L.z - base] = e + 1u;
                else if (L.z < base && L.z + L.w > base) s_owner[0] = e + 1u;
            }
            __syncthreads();
            uint32_t e_first = 0, e_end = 0;  // the listed voxels that own slots of this round
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
                uint32_t       last = 0u;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    src[k] = 0u;
                    if (t0 + k < m)
                    {
                        const uint4 L = s_list[ow[k] - 1u];
                        src[k]        = L.y + (base + t0 + k - L.z);
                        last          = ow[k];
                    }
                }
                e_first = (uint32_t)__builtin_amdgcn_readlane((int)ow[0], 0) - 1u;  // slot 0 always has an owner
                e_end   = wave_max_u32(last);                                       // ids ascend with the slots
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    c4[k] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
                    if (t0 + k < m) c4[k] = g.pts[src[k]];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) s_pts[t0 + k] = c4[k];
                if (INSTR)
                {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (t0 + k < m) a.touched[src[k]] = 1;
                }
            }
            __syncthreads();
            if (INSTR)
            {
                const unsigned long long t = wall_clock64();
                t_stage += t - tph, tph = t;
            }

            // ---- D: the lane's walk over the listed voxels: those its ball still reaches, their staged
            //      points 4 at a time.  One flat loop (refill / work) so that lanes that skip a voxel do
            //      not wait at every nesting level ---------------------------------------------------
            uint32_t my_tests = 0;
            if (go)
            {
                uint32_t e = e_first, k = 0, kend = 0, sb = 0;
                for (;;)
                {
                    while (k >= kend && e < e_end)
                    {
                        const uint4    L  = s_list[e];
                        const uint32_t v  = L.x - 1u;
                        const uint32_t cx = wx0 + (v & 1023u), cy = wy0 + ((v >> 10) & 1023u), cz = wz0 + (v >> 20);
                        e++;
                        const float md2 = box_dist2(g.ox + (float)cx * hs, g.oy + (float)cy * hs, g.oz + (float)cz * hs, hs, qx, qy, qz);
                        if (md2 <= fminf(prune2, voxel_limit(best_d2, g.slack)))
                        {
                            const uint32_t k0 = max(L.z, base), k1 = min(L.z + L.w, base + (uint32_t)NW_CAP);
                            if (k0 < k1) k = k0 - base, kend = k1 - base, sb = L.y + (k0 - L.z) - k;  // sorted position of slot j = sb + j
                        }
                    }
                    if (k >= kend) break;
                    float4 c4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        c4[j] = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
                        if (k + j < kend) c4[j] = s_pts[k + j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        const float    d  = dist2(qx, qy, qz, c4[j].x, c4[j].y, c4[j].z);
                        const uint32_t ci = __float_as_uint(c4[j].w);
                        if (k + j < kend && (d < best_d2 || (d == best_d2 && ci < best_idx)))
                            best_d2 = d, best_idx = ci, best_spos = sb + k + j;
                        if (INSTR && k + j < kend) my_tests++;
                    }
                    k += 4u;
                }
            }
            __syncthreads();  // the next round overwrites the staged points
            if (INSTR)
            {
                st_tests += my_tests;
                st_maxlane += wave_max_u32(my_tests);
                const unsigned long long t = wall_clock64();
                t_scan += t - tph, tph = t;
            }
        }

        // ---- conclude, or grow ---------------------------------------------------------------------
        if (part)
        {
            if (toobig)
            {
                if (INSTR) st_toobig++;
            }
            else if (ovf)
            {
                if (INSTR) st_ovf++;
            }
            else if (is_final(r, rmax, best_d2, g.slack)) done = true;
            else r = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
        }
        // a cube too wide for a lane: the one-query-per-wave kernel enumerates bricks at a coarser level
        {
            const unsigned long long bmask = __ballot(part && toobig);
            if (bmask)
            {
                st_defer += push_lanes(a, 1, wv / a.seg_waves, part && toobig, bmask, lane, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
                if (part && toobig) deferred = true;
            }
        }
    }
    // whatever is still open (set overflow, many growth steps) is handed on with its state
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
    __syncthreads();  // the claim table lives in the staging area
    emit_wave(a, s_claim, lane, valid && !deferred, qi, orig, active, thr, best_d2, best_idx, best_spos,
              lb2_out >= 0.f ? lb2_out : fminf(best_d2, thr));

    if (a.timeline && lane == 0)
    {
        // start tick in the low 40 bits; above: widest voxel list (8 bits), staging rounds (8), passes (4), flags
        const unsigned long long info = (unsigned long long)min(dbg_nu, 255u) | ((unsigned long long)min(dbg_rounds, 255u) << 8) |
                                        ((unsigned long long)min(st_pass, 15u) << 16) | ((unsigned long long)dbg_flags << 20);
        a.timeline[2 * (size_t)blockIdx.x]     = (tl0 & 0xFFFFFFFFFFull) | (info << 40);
        a.timeline[2 * (size_t)blockIdx.x + 1] = wall_clock64() & 0xFFFFFFFFFFull;
    }
    if (INSTR)
    {
        const unsigned long long t_end = wall_clock64();
        const uint32_t tests = wave_sum_u32(st_tests), ins = wave_sum_u32(st_ins), ovfs = wave_sum_u32(st_ovf),
                       big = wave_sum_u32(st_toobig);
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
            atomicAdd(&a.counters[NWC_LANE_TESTS], (unsigned long long)tests);
            atomicAdd(&a.counters[NWC_MAXLANE], (unsigned long long)st_maxlane);
            atomicAdd(&a.counters[NWC_INSERTS], (unsigned long long)ins);
            atomicAdd(&a.counters[NWC_OVF], (unsigned long long)ovfs);
            atomicAdd(&a.counters[NWC_ROUNDS], (unsigned long long)st_rounds);
            atomicAdd(&a.counters[NWC_LISTED], (unsigned long long)st_listed);
            atomicAdd(&a.counters[NWC_TOOBIG], (unsigned long long)big);
            atomicAdd(&a.counters[NWC_T_INS], t_ins);
            atomicAdd(&a.counters[NWC_T_DIR], t_dir);
            atomicAdd(&a.counters[NWC_T_STAGE], t_stage);
            atomicAdd(&a.counters[NWC_T_SCAN], t_scan);
            atomicAdd(&a.counters[NWC_T_EMIT], t_end - tph);
        }
    }
}

}  // namespace mp2p
