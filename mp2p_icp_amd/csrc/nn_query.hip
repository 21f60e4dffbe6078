// nn_query.hip -- K1 (fused) + K3: local->global transform and EXACT nearest-neighbour search.
//
// Replaces, per outer ICP iteration,
//   transform_local_to_global           Matcher_Points_Base.cpp:183-249
//   the nn_single_search loop            Matcher_Points_DistanceThreshold.cpp:214-243
//   the threshold rule                   :252-259
//   the "global point already paired"    :94-121 (claims; resolved in pairs.hip)
//
// Mapping onto CDNA4 (one wave64 = one workgroup = one TILE of Q Morton-consecutive queries):
//   * lane = (query slot, candidate slice): Q query slots x S=64/Q slices.  Every lane keeps
//     its query in registers; the wave stages the candidate points of all voxels overlapping
//     the tile's search box into LDS with coalesced 16-byte loads, then every lane scans the
//     staged bucket (its slice of it) with broadcast ds_read_b128 -- each HBM/L2 byte is
//     fetched once per tile and reused by Q queries.
//   * argmin is lexicographic on (fp32 d2, original global index), so the result does not
//     depend on traversal order and ties resolve to the lowest index (the repo's policy).
//   * exactness without an unbounded search: a query is final when its best distance is
//     below the radius its visited voxels are guaranteed to cover, or when that radius has
//     reached r_max = sqrt(threshold rule) beyond which the reference discards the pair
//     anyway.  Otherwise the radius grows (to the best distance found, else x2) and the tile
//     repeats at a coarser voxel level.
#include "device_utils.hpp"

namespace mp2p
{
constexpr int      NN_CAP         = 512;   // staged candidates per round (LDS: 5 x 2 KB)
constexpr uint32_t NN_CELL_BUDGET = 2048;  // voxels of the search box per pass (<=32 lookups/lane)
constexpr int      NN_COOP_MAX    = 4;     // group size up to which the scan is cooperative

struct NNArgs
{
    GridView      g;
    const float4* lpts;  // Morton-sorted local points {x,y,z,bits(orig idx)}
    uint32_t      n_l;
    PoseRt        pose;
    float         maxDistSq, angSq;  // Matcher_Points_DistanceThreshold.cpp:82-83
    float         r0;                // first search radius [m]
    float         grp_factor;        // group extent in units of the seed's radius
    const unsigned char* local_taken;   // by original local index, or null
    const unsigned char* global_taken;  // by original global index, or null
    unsigned long long*  claims;        // by sorted global position, or null
    unsigned long long   claim_hi;      // (~epoch) << 32
    unsigned long long   local_offset;  // whole-layer index of this rank's first local point
    uint32_t*            out_spos;      // [n_l] by original local index
    float*               out_d2;
    float*               tile_bbox;  // [n_tiles][6]
    unsigned long long*  counters;   // profiling, or null
    unsigned char*       touched;    // profiling: [n_g] by sorted position, or null
};

template <int Q>
__global__ __launch_bounds__(64) void nn_tile_kernel(const NNArgs a)
{
    constexpr int S = 64 / Q;
    __shared__ __attribute__((aligned(16))) float s_x[NN_CAP];
    __shared__ __attribute__((aligned(16))) float s_y[NN_CAP];
    __shared__ __attribute__((aligned(16))) float s_z[NN_CAP];
    __shared__ uint32_t s_idx[NN_CAP];
    __shared__ uint32_t s_spos[NN_CAP];
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[65];

    const GridView& g     = a.g;
    const int       lane  = threadIdx.x;
    const int       qslot = lane & (Q - 1);
    const int       slice = (Q == 64) ? 0 : lane / Q;
    const uint32_t  tile  = blockIdx.x;
    const uint32_t  qi    = tile * Q + qslot;
    const bool      valid = qi < a.n_l;

    float4 lp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = a.lpts[qi];
    const uint32_t orig = __float_as_uint(lp.w);

    // ---- K1: transform (fp64 compose, one narrowing) ------------------------------------
    float qx, qy, qz;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);

    // bounding box of ALL transformed local points of the tile (Matcher_Points_Base.cpp:186-196)
    {
        const float bx0 = wave_min(valid ? qx : INFINITY), by0 = wave_min(valid ? qy : INFINITY),
                    bz0 = wave_min(valid ? qz : INFINITY);
        const float bx1 = wave_max(valid ? qx : -INFINITY), by1 = wave_max(valid ? qy : -INFINITY),
                    bz1 = wave_max(valid ? qz : -INFINITY);
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

    bool active = valid && (normSq < INFINITY);  // non-finite query: nothing to pair
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // :218-220

    float    r        = fminf(a.r0, rmax);
    bool     done     = !active;
    float    best_d2  = INFINITY;
    uint32_t best_idx = NONE_U32, best_spos = NONE_U32;

    uint32_t st_pass = 0, st_cells = 0, st_cand = 0, st_maxcand = 0, st_coop = 0;
    const long long t_start = a.counters ? (long long)wall_clock64() : 0;

    while (true)
    {
        const unsigned long long pend = __ballot(!done);
        if (pend == 0ull) break;
        st_pass++;

        // ---- this pass serves a GROUP of pending queries: those within grp_factor radii of
        //      the first pending one (and of comparable radius).  A Morton-consecutive tile is
        //      normally one group; a tile straddling a jump of the curve (or holding far-range
        //      returns) is split so that the search box never dwarfs the search balls.  The
        //      other pending lanes still test the staged points (every candidate is a valid
        //      upper bound) but only group members may conclude. -------------------------------
        const int   seed = __ffsll((long long)pend) - 1;
        const float sx = __shfl(qx, seed, 64), sy = __shfl(qy, seed, 64), sz = __shfl(qz, seed, 64);
        const float sr = __shfl(r, seed, 64);
        const float G  = a.grp_factor * sr;
        const bool  grp = !done && fabsf(qx - sx) <= G && fabsf(qy - sy) <= G &&
                         fabsf(qz - sz) <= G && r <= 2.0f * sr;

        // ---- search box = union of the group's cubes ---------------------------------------
        float lox = wave_min(grp ? qx - r : INFINITY), loy = wave_min(grp ? qy - r : INFINITY),
              loz = wave_min(grp ? qz - r : INFINITY);
        float hix = wave_max(grp ? qx + r : -INFINITY), hiy = wave_max(grp ? qy + r : -INFINITY),
              hiz = wave_max(grp ? qz + r : -INFINITY);
        const float rmin_t = wave_min(grp ? r : INFINITY);
        const float rmax_t = wave_max(grp ? r : 0.f);
        // conservative bounding box of the group's queries themselves
        const float qlx = lox + rmin_t, qly = loy + rmin_t, qlz = loz + rmin_t;
        const float qhx = hix - rmin_t, qhy = hiy - rmin_t, qhz = hiz - rmin_t;

        // clip to the layer's bounding box; disjoint -> nothing to visit
        lox = fmaxf(lox, g.bbmin[0]), loy = fmaxf(loy, g.bbmin[1]), loz = fmaxf(loz, g.bbmin[2]);
        hix = fminf(hix, g.bbmax[0]), hiy = fminf(hiy, g.bbmax[1]), hiz = fminf(hiz, g.bbmax[2]);
        const bool empty_box = (lox > hix) || (loy > hiy) || (loz > hiz);

        uint32_t           nx = 0, ny = 0, nz = 0, cx0 = 0, cy0 = 0, cz0 = 0, s = g.shift0, lev = 0;
        unsigned long long ncell = 0;
        if (!empty_box)
        {
            const uint32_t flx = cell_fine(lox, g.ox, g.inv_hf), fhx = cell_fine(hix, g.ox, g.inv_hf);
            const uint32_t fly = cell_fine(loy, g.oy, g.inv_hf), fhy = cell_fine(hiy, g.oy, g.inv_hf);
            const uint32_t flz = cell_fine(loz, g.oz, g.inv_hf), fhz = cell_fine(hiz, g.oz, g.inv_hf);
            for (;;)
            {
                cx0 = flx >> s, cy0 = fly >> s, cz0 = flz >> s;
                nx = (fhx >> s) - cx0 + 1, ny = (fhy >> s) - cy0 + 1, nz = (fhz >> s) - cz0 + 1;
                ncell = (unsigned long long)nx * ny * nz;
                if (ncell <= NN_CELL_BUDGET || lev + 1 >= g.n_levels) break;
                s++, lev++;
            }
        }
        const float hs     = g.hf * (float)(1u << s);  // voxel edge at this level
        const float prune  = rmax_t + 4.f * g.slack;
        const float prune2 = prune * prune;

        // ---- small group: switch to the COOPERATIVE scan (lanes split the candidates of up
        //      to 4 member queries and a wave arg-min merges them): a sparse far-range return
        //      then costs 1/64 of the per-lane scan ------------------------------------------
        const unsigned long long gmask = __ballot(grp);
        const int                k_grp = __popcll(gmask);
        const bool               coop  = (S == 1) && (k_grp <= NN_COOP_MAX);
        st_coop += coop ? 1u : 0u;
        float    mqx[NN_COOP_MAX], mqy[NN_COOP_MAX], mqz[NN_COOP_MAX];
        float    pb_d2[NN_COOP_MAX];
        uint32_t pb_idx[NN_COOP_MAX], pb_spos[NN_COOP_MAX];
        int      mlane[NN_COOP_MAX];
        {
            unsigned long long tmp = gmask;
#pragma unroll
            for (int t = 0; t < NN_COOP_MAX; t++)
            {
                mlane[t] = tmp ? (__ffsll((long long)tmp) - 1) : 0;
                tmp &= tmp - 1;
                mqx[t] = __shfl(qx, mlane[t], 64), mqy[t] = __shfl(qy, mlane[t], 64),
                mqz[t] = __shfl(qz, mlane[t], 64);
                pb_d2[t] = INFINITY, pb_idx[t] = NONE_U32, pb_spos[t] = NONE_U32;
            }
        }

        const bool small_grid = ncell <= 0xFFFFFFFFull;
        for (unsigned long long cb = 0; cb < ncell; cb += 64)
        {
            const unsigned long long cid = cb + lane;
            uint32_t                 cnt = 0, start = 0;
            if (cid < ncell)
            {
                uint32_t ix, iy, iz;
                if (small_grid)
                {
                    const uint32_t c32 = (uint32_t)cid, row = c32 / nx;
                    ix = c32 - row * nx, iz = row / ny, iy = row - iz * ny;
                }
                else
                {
                    ix = (uint32_t)(cid % nx), iy = (uint32_t)((cid / nx) % ny);
                    iz = (uint32_t)(cid / ((unsigned long long)nx * ny));
                }
                const uint32_t cx = cx0 + ix, cy = cy0 + iy, cz = cz0 + iz;
                // voxel box vs bounding box of the group's queries
                const float vx0 = g.ox + (float)cx * hs, vy0 = g.oy + (float)cy * hs,
                            vz0 = g.oz + (float)cz * hs;
                const float dx = fmaxf(0.f, fmaxf(vx0 - qhx, qlx - (vx0 + hs)));
                const float dy = fmaxf(0.f, fmaxf(vy0 - qhy, qly - (vy0 + hs)));
                const float dz = fmaxf(0.f, fmaxf(vz0 - qhz, qlz - (vz0 + hs)));
                if (dx * dx + dy * dy + dz * dz <= prune2)
                {
                    uint32_t e = 0;
                    if (cell_lookup(g, cell_key(lev, cx, cy, cz), start, e)) cnt = e - start;
                }
            }
            const uint32_t incl  = wave_incl_scan(cnt, lane);
            const uint32_t total = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));
            st_cells += (uint32_t)min((unsigned long long)64, ncell - cb);
            if (total == 0) continue;  // uniform: no occupied voxel in this batch
            s_cstart[lane] = start;
            s_coff[lane]   = incl - cnt;
            if (lane == 63) s_coff[64] = total;
            __syncthreads();
            st_cand += total;
            st_maxcand = max(st_maxcand, total);

            for (uint32_t base = 0; base < total; base += NN_CAP)
            {
                const uint32_t m     = min((uint32_t)NN_CAP, total - base);
                const uint32_t m_pad = (m + 31u) & ~31u;
                // ---- stage: coalesced 16-byte loads, lane t <- t-th candidate of the round,
                //      stored as SoA so that the scan reads 4 candidates per ds_read_b128
                for (uint32_t t = lane; t < m_pad; t += 64)
                {
                    float4   c   = make_float4(INFINITY, 0.f, 0.f, __uint_as_float(NONE_U32));
                    uint32_t src = NONE_U32;
                    if (t < m)
                    {
                        const uint32_t gt = base + t;
                        int            lo = 0, hi = 63;
                        while (lo < hi)
                        {
                            const int mid = (lo + hi + 1) >> 1;
                            if (s_coff[mid] <= gt) lo = mid;
                            else hi = mid - 1;
                        }
                        src = s_cstart[lo] + (gt - s_coff[lo]);
                        c   = g.pts[src];
                        if (a.touched) a.touched[src] = 1;
                    }
                    s_x[t] = c.x, s_y[t] = c.y, s_z[t] = c.z;
                    s_idx[t]  = __float_as_uint(c.w);
                    s_spos[t] = src;
                }
                __syncthreads();
                if (coop)
                {
                    for (uint32_t j = lane; j < m; j += 64)
                    {
                        const float    cx = s_x[j], cy = s_y[j], cz = s_z[j];
                        const uint32_t ci = s_idx[j];
#pragma unroll
                        for (int t = 0; t < NN_COOP_MAX; t++)
                        {
                            if (t < k_grp)
                            {
                                const float d2 = dist2(mqx[t], mqy[t], mqz[t], cx, cy, cz);
                                if (d2 < pb_d2[t] || (d2 == pb_d2[t] && ci < pb_idx[t]))
                                    pb_d2[t] = d2, pb_idx[t] = ci, pb_spos[t] = s_spos[j];
                            }
                        }
                    }
                }
                else
                {
                    // ---- scan: every lane tests (its slice of) the bucket against its query,
                    //      8 candidates per step; the update path is rare after the first few
                    for (uint32_t jb = (uint32_t)slice * 8u; jb < m_pad; jb += 8u * S)
                    {
                        const float4 xa = *reinterpret_cast<const float4*>(&s_x[jb]);
                        const float4 xb = *reinterpret_cast<const float4*>(&s_x[jb + 4]);
                        const float4 ya = *reinterpret_cast<const float4*>(&s_y[jb]);
                        const float4 yb = *reinterpret_cast<const float4*>(&s_y[jb + 4]);
                        const float4 za = *reinterpret_cast<const float4*>(&s_z[jb]);
                        const float4 zb = *reinterpret_cast<const float4*>(&s_z[jb + 4]);
                        float        d[8];
                        d[0] = dist2(qx, qy, qz, xa.x, ya.x, za.x);
                        d[1] = dist2(qx, qy, qz, xa.y, ya.y, za.y);
                        d[2] = dist2(qx, qy, qz, xa.z, ya.z, za.z);
                        d[3] = dist2(qx, qy, qz, xa.w, ya.w, za.w);
                        d[4] = dist2(qx, qy, qz, xb.x, yb.x, zb.x);
                        d[5] = dist2(qx, qy, qz, xb.y, yb.y, zb.y);
                        d[6] = dist2(qx, qy, qz, xb.z, yb.z, zb.z);
                        d[7] = dist2(qx, qy, qz, xb.w, yb.w, zb.w);
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
                }
                __syncthreads();
            }
        }

        if (coop)
        {
            // wave arg-min per member, merged into the member's own lane
#pragma unroll
            for (int t = 0; t < NN_COOP_MAX; t++)
            {
                if (t < k_grp)
                {
                    float    bd = pb_d2[t];
                    uint32_t bi = pb_idx[t], bs = pb_spos[t];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1)
                    {
                        const float    od = __shfl_xor(bd, off, 64);
                        const uint32_t oi = __shfl_xor(bi, off, 64);
                        const uint32_t os = __shfl_xor(bs, off, 64);
                        if (od < bd || (od == bd && oi < bi)) bd = od, bi = oi, bs = os;
                    }
                    if (lane == mlane[t] && (bd < best_d2 || (bd == best_d2 && bi < best_idx)))
                        best_d2 = bd, best_idx = bi, best_spos = bs;
                }
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

        // ---- final?  (visited voxels cover the whole cube of half-edge r around the query)
        if (grp)
        {
            const float gr = r * (1.0f - 1.0f / 1024.0f) - g.slack;
            if (r >= rmax || (gr > 0.f && best_d2 < gr * gr))
                done = true;
            else
            {
                const float rn = (best_idx != NONE_U32)
                                     ? sqrtf(best_d2) * (1.0f + 1.0f / 512.0f) + 4.f * g.slack
                                     : 2.0f * r;
                r = fminf(fmaxf(rn, r * 1.0009765625f), rmax);
            }
        }
    }

    // ---- output (original local order) + claim of the global point --------------------------
    if (valid && slice == 0)
    {
        bool acc = active && best_idx != NONE_U32 && best_d2 < thr;  // :259
        if (acc && a.global_taken && a.global_taken[best_idx]) acc = false;  // :98-101
        a.out_spos[orig] = acc ? best_spos : NONE_U32;
        a.out_d2[orig]   = best_d2;
        if (acc && a.claims)
            atomicMin(&a.claims[best_spos], a.claim_hi | (a.local_offset + orig));
    }
    if (a.counters && lane == 0)
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
        atomicAdd(&a.counters[9], (unsigned long long)st_coop);
        // histogram of per-tile wall time, log2 bins of 100 MHz ticks
        int b = 63 - __clzll((long long)(dt | 1ull));
        if (b > 23) b = 23;
        atomicAdd(&a.counters[16 + b], 1ull);
    }
}

// reduce the per-tile boxes to the layer box {min xyz, max xyz}
__global__ __launch_bounds__(256) void tile_bbox_reduce_kernel(const float* __restrict__ tb,
                                                               uint32_t n_tiles,
                                                               float* __restrict__ out6)
{
    float v[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = threadIdx.x; i < n_tiles; i += blockDim.x)
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
        out6[d] = r;
    }
}

// ------------------------------------------------------------------------------------------
int launch_nn_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                    const double pose[12], const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms)
{
    const size_t n_l = cloud->n;
    uint32_t     Q   = prm->queries_per_wave ? prm->queries_per_wave : 64;
    MP2P_REQUIRE(ctx, Q == 64 || Q == 16, "queries_per_wave must be 64 or 16");
    const uint32_t n_tiles = (uint32_t)((n_l + Q - 1) / Q);

    MP2P_TRY_HIP(ctx, ctx->nn_spos.ensure(n_l));
    MP2P_TRY_HIP(ctx, ctx->nn_d2.ensure(n_l));
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)n_tiles * 6));
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
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
    a.r0 = cell0 * (prm->initial_radius_cells > 0 ? prm->initial_radius_cells : 1.0f);
    a.grp_factor = prm->group_radius_factor > 0 ? prm->group_radius_factor : 4.0f;
    a.local_taken =
        (ms && !prm->allowMatchAlreadyMatchedPoints) ? ms->local_taken.p : nullptr;
    a.global_taken =
        (ms && !prm->allowMatchAlreadyMatchedGlobalPoints) ? ms->global_taken.p : nullptr;
    a.claims = prm->allowMatchAlreadyMatchedGlobalPoints ? nullptr : map->claims.p;
    ctx->epoch++;
    a.claim_hi     = (~(unsigned long long)ctx->epoch) << 32;
    a.local_offset = prm->local_index_offset;
    a.out_spos     = ctx->nn_spos.p;
    a.out_d2       = ctx->nn_d2.p;
    a.tile_bbox    = ctx->tile_bbox.p;
    a.counters     = nullptr;
    a.touched      = nullptr;
    if (ctx->profiling >= 2)
    {
        MP2P_TRY_HIP(ctx, ctx->counters.ensure(64));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64 * sizeof(unsigned long long), ctx->stream));
        a.counters = ctx->counters.p;
        MP2P_TRY_HIP(ctx, ctx->pl_slots.ensure(map->n));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->pl_slots.p, 0, map->n, ctx->stream));
        a.touched = ctx->pl_slots.p;
    }
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (n_tiles)
    {
        switch (Q)
        {
            case 64: hipLaunchKernelGGL(nn_tile_kernel<64>, dim3(n_tiles), dim3(64), 0, ctx->stream, a); break;
            default: hipLaunchKernelGGL(nn_tile_kernel<16>, dim3(n_tiles), dim3(64), 0, ctx->stream, a); break;
        }
    }
    // ev[0]..ev[1] brackets exactly the search kernel (the roofline kernel of bench.py)
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    hipLaunchKernelGGL(tile_bbox_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream,
                       ctx->tile_bbox.p, n_tiles, ctx->local_bbox.p);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

}  // namespace mp2p
