// nn_pt2pl.hip -- K5: Matcher_Point2Plane::implMatchOneLayer (Matcher_Point2Plane.cpp:41-114)
// with NearestPlaneCapable::nn_search_pt2pl (NearestPlaneCapable.h:33-52) implemented as
//   exact k-NN (fp32 metric, (d2,idx) order) restricted to d2 <= searchRadius^2
//   -> estimate_points_eigen (estimate_points_eigen.cpp:27-123: fp32 mean, fp32 products
//      accumulated in fp64, scaled by fp32 1/n) -> symmetric 3x3 eigen (cyclic Jacobi)
//   -> planar iff e0 < thr*e1 && e0 < thr*e2 (Matcher_Adaptive.cpp:241-242)
//   -> plane (centroid, eigvec0), unit normal with its largest |component| positive
//   -> |plane.distance(q)| <= distanceThreshold (Matcher_Point2Plane.cpp:100-101).
// The arithmetic that decides WHICH local points get a pairing is sequenced exactly as in
// oracle/mp2p_oracle.c (no FMA contraction: this unit is built with -ffp-contract=off), so
// the pair lists match bit for bit; upstream has no in-repo implementor of nn_search_pt2pl
//  -> "parity unpinned" beyond the disabled test's expectations.
//
// Same tile machinery as nn_query.hip (wave64 tile, LDS-staged voxel buckets); each lane keeps
// a sorted k-list in registers.
#include "device_utils.hpp"
#include "nn_pl_seltile.hip"  // round 6: knn_sel_search (ball-rule selection + matrix-pipe prefilter, W waves per tile)

namespace mp2p
{
constexpr int      PL_CAP         = 256;
constexpr int      PL_CB          = 1024; // queries per block of pt2pl_cert_kernel = one reservation in the pending list (one same-address atomic: 256 per block were 19 500 serialised atomics, 0.23 ms, per call of a 5 M-query layer)
constexpr int      PL_STAGE       = 4;    // staging loads in flight per lane (x 64 candidates per fetch)
constexpr int      PL_HITQ        = 8;    // queued hits per lane before the insertion chains run
constexpr int      PL_SB          = 4;    // batches of 64 voxels resolved together (round 5): their directory loads are in flight at once and
constexpr int      PL_CELLS       = 64 * PL_SB;  // ... their points staged and scanned as ONE bucket list (a pass of <= 256 voxels: 2-3 dependent
                                          // round trips instead of 8; a pass of a latency-bound 8-query tile cost 19 us)
constexpr uint32_t PL_CELL_BUDGET  = 256;   // voxels per pass; measured insensitive 256..4096
constexpr float    PL_GROUP_FACTOR = 4.0f;  // group extent in search radii; insensitive 1.5..4

struct PlArgs
{
    GridView      g;
    const float4* lpts;
    uint32_t      n_l;
    PoseRt        pose;
    float         radSq, rad, distThr, r0, grp_factor;
    uint32_t      cell_budget;
    unsigned long long* dbg;  // profiling level 2: {tiles, passes, candidates, ticks, max passes, max cand, max ticks, cells, max cells}
    double        eigThr;
    uint32_t      minPts;
    uint32_t      knn;  // <= K (template capacity of the register k-list)
    const unsigned char* local_taken;
    const uint32_t*      rank;      // visit rank per original local index (NONE = not visited) or null
    uint32_t*            out_knn;   // [n_l][K capacity] neighbour lists by original local index (search -> fit)
    unsigned char*       out_flag;  // [n_l] by original local index
    double*              out_rec;   // [n_l][7] plane(4) + centroid(3)
    float*               tile_bbox;
    // warm start (round 3): the matcher is called on the same map and cloud with a slowly changing pose.  Every
    // one of a query's previous k neighbours is still a map point and at most (its old distance + the query's
    // displacement) away, so the previous k-th distance + the displacement is a radius CERTAIN to hold k points
    // again: the search starts there (typically 0.1 m) instead of at the full search radius (0.4 m: 8..30 x the
    // points).  kth_io[n_l] in the Morton order of the local layer: d2 of the knn-th neighbour (inf: fewer in reach)
    float*               kth_io;
    int                  use_hint;
    PoseRt               prev_pose;
    float                grp_min;   // smallest group extent [m]: tight radii must not split a tile into many passes
    // certificate (round 3): lb_io[n_l] (Morton order) = a lower bound of the distance from the query to every map
    // point that is NOT in its list (0: none known) -- the smaller of the nearest candidate a search rejected and the
    // radius it covered completely.  At the next call the query has moved by `disp`: no outsider is nearer than
    // lb - disp, so if the old neighbours, re-measured from the new position, all stay below that, the k nearest are
    // the same points and only their order has to be re-established -- no search (pt2pl_cert_kernel).  A query with
    // fewer than knn points in reach stays as it is while lb - disp clears the search radius (rad_cert = the radius
    // such a search covers: a little more than searchRadius, so that this margin exists).
    float*               lb_io;
    // the queries left to the search, as places in the Morton-ordered copy: two lists for the layer (easy class, hard
    // class).  A block of pt2pl_cert_kernel reserves its part of a list with one atomic, padded with NONE to whole
    // tiles -- a tile never mixes two blocks, the lists have no holes larger than a tile, and the grid holds no empty
    // workgroups except at its very end (per-block segments left 30..50 % of the grid empty once queries were
    // certified, and the dispatcher's time for those showed: 3 150 waves resident instead of 4 400)
    uint32_t*            pend;      // [n_l + blocks * Q]
    uint32_t*            hard_list; // [hard_cap]
    uint32_t*            list_cnt;  // {entries of pend, entries of hard_list}; zeroed by the fit kernel for the next call
    uint32_t             hard_cap, pend_cap;
    unsigned long long*  cert_stat; // 64 lines (16 words apart) of {queries certified, queries searched, of these in the hard class}, accumulated; block b adds to line b % 64
    uint32_t*            cost_io;   // [n_l] (Morton order): candidates the tile that served the query staged at the previous call
    uint32_t             hard_cand;  // a query whose tile staged at least this many goes to the hard class (0: no classes)
    float                rad_cert, cert_margin;
    int                  use_cert;
    unsigned char*       touched;   // profiling level 2: one byte per map point (sorted position), set when the point is fetched
    int                  sol;             // timing-only cuts of pt2pl_seltile_kernel (profiling builds; MP2P_HIP_TUNE pl_sol through mp2p_hip_set_tune)
    float                grp_all_bricks;  // round 6 (pt2pl_seltile_kernel): all pending queries of a tile form one pass while their box is this many bricks wide
    uint32_t             timeline_n; // tiles in the grid; after their {start, end}: one word {passes << 48 | voxels << 32 | candidates} each
    unsigned long long*  timeline;  // profiling level 4: {start, end} 100 MHz ticks per tile of pt2pl_tile_kernel (0 0: an empty tile)
};

// cyclic Jacobi, identical operation order to sym_eig_jacobi(3, ...) of the oracle
__device__ void jacobi3(const double* Ain, double* eval, double* evec0)
{
    double A[9], Q[9];
    for (int i = 0; i < 9; i++) A[i] = Ain[i];
    Q[0] = 1, Q[1] = 0, Q[2] = 0, Q[3] = 0, Q[4] = 1, Q[5] = 0, Q[6] = 0, Q[7] = 0, Q[8] = 1;
    const int n = 3;
    // same convergence rule as the oracle (off-diagonal mass below 1e-18 of the diagonal scale)
    double scale = fabs(A[0]) + fabs(A[4]) + fabs(A[8]);
    scale        = 1e-36 * (scale * scale);
    for (int sweep = 0; sweep < 64; sweep++)
    {
        double off = 0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        if (off <= scale) break;
        for (int p = 0; p < n; p++)
        {
            for (int q = p + 1; q < n; q++)
            {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double app = A[p * n + p], aqq = A[q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++)
                {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++)
                {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++)
                {
                    const double qkp = Q[k * n + p], qkq = Q[k * n + q];
                    Q[k * n + p] = c * qkp - s * qkq;
                    Q[k * n + q] = s * qkp + c * qkq;
                }
            }
        }
    }
    int order[3] = {0, 1, 2};
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]])
            {
                const int t = order[i];
                order[i] = order[j], order[j] = t;
            }
    for (int k = 0; k < 3; k++) eval[k] = A[order[k] * n + order[k]];
    for (int i = 0; i < 3; i++) evec0[i] = Q[i * n + order[0]];
}

// distance of the knn-th neighbour held so far (static indexing only: no scratch)
template <int K>
__device__ __forceinline__ float kth_d2(const float (&kd2)[K], uint32_t knn)
{
    float v = INFINITY;
#pragma unroll
    for (int q = 0; q < K; q++)
        if (q == (int)knn - 1) v = kd2[q];
    return v;
}

// Exact k nearest neighbours (fp32 metric, (d2, idx) order) of Q queries per wave, restricted to
// d2 <= lim2 (STRICT: d2 < lim2); rmax = a radius that covers every such point.  A query is held by
// S = 64/Q lanes (lane, lane+Q, ...), each scanning every S-th staged candidate into its own
// k-list; the lists are merged at the end of a pass.  All lanes scan
// the staged voxel buckets of the wave's common search box; each lane keeps its sorted k-list in
// registers.  Uniform control flow: must be called by the whole wave.
// CERT: *lb_out = a lower bound of the (computed) distance to every point outside the final list: the nearest
// candidate this search tested and did not keep (rej), or the radius it covered completely.
template <int K, bool STRICT, int Q, bool CERT = false>
__device__ __forceinline__ void knn_search(const GridView& g, int lane, float qx, float qy, float qz,
                                           bool active, float lim2, float rmax, float r0, uint32_t knn,
                                           float grp_factor, uint32_t cell_budget,
                                           unsigned long long* dbg, uint32_t* s_hit, float4* s_cand,
                                           uint32_t* s_spos, uint32_t* s_cstart,
                                           uint32_t* s_coff, float (&kd2)[K], uint32_t (&kidx)[K],
                                           uint32_t (&kspos)[K], float grp_min = 0.f, float* lb_out = nullptr,
                                           float cert_margin = 0.f, unsigned long long* tl_info = nullptr,
                                           uint32_t* cand_out = nullptr, unsigned char* touched = nullptr)
{
    float r    = fminf(r0, rmax);
    bool  done = !active;
    float rej  = INFINITY;  // CERT: smallest d2 of a candidate tested and not kept (this pass, this lane's slice); once done: the bound
#pragma unroll
    for (int j = 0; j < K; j++) kd2[j] = INFINITY, kidx[j] = NONE_U32, kspos[j] = NONE_U32;
    unsigned long long dbg_cand = 0, dbg_pass = 0, dbg_cells = 0;
    const long long    dbg_t0 = dbg ? (long long)wall_clock64() : 0;

    while (true)
    {
        const unsigned long long pend = __ballot(!done);
        if (pend == 0ull) break;
        // One pass serves the GROUP of pending queries near the first pending one (a few search
        // radii around it): Morton-consecutive queries are normally one group, an outlier is a
        // group of its own with a box of its own size.  Without this the pass box is the union over
        // the whole tile, and one far-away query makes all 64 lanes scan 1e5 points.
        const int   seed = __ffsll((long long)pend) - 1;
        const float sx = readlane_f(qx, seed), sy = readlane_f(qy, seed), sz = readlane_f(qz, seed);
        const float sr = readlane_f(r, seed);
        const float G  = fmaxf(grp_factor * sr, grp_min);
        const bool  grp = !done && fabsf(qx - sx) <= G && fabsf(qy - sy) <= G && fabsf(qz - sz) <= G &&
                         r <= fmaxf(2.0f * sr, 0.5f * grp_min);
        float lox = wave_min_nn(grp ? qx - r : INFINITY), loy = wave_min_nn(grp ? qy - r : INFINITY),
              loz = wave_min_nn(grp ? qz - r : INFINITY);
        float hix = wave_max_nn(grp ? qx + r : -INFINITY), hiy = wave_max_nn(grp ? qy + r : -INFINITY),
              hiz = wave_max_nn(grp ? qz + r : -INFINITY);
        const float rmin_t = wave_min_pos(grp ? r : INFINITY);
        const float rmax_t = wave_max_pos(grp ? r : 0.f);
        const float qlx = lox + rmin_t, qly = loy + rmin_t, qlz = loz + rmin_t;
        const float qhx = hix - rmin_t, qhy = hiy - rmin_t, qhz = hiz - rmin_t;
        // the staging filter's box: the union of the group's search cubes (+ the certificate's margin); a point outside it
        // is farther than r from every query of the group
        const float fpad = (CERT ? cert_margin : 0.f) + 2.f * g.slack;
        const float flx = lox - fpad, fly = loy - fpad, flz = loz - fpad, fhx = hix + fpad, fhy = hiy + fpad, fhz = hiz + fpad;
        // CERT: a face of the pass box that was cut back to the map's bounding box has nothing beyond it
        const bool open_lx = lox < g.bbmin[0], open_ly = loy < g.bbmin[1], open_lz = loz < g.bbmin[2];
        const bool open_hx = hix > g.bbmax[0], open_hy = hiy > g.bbmax[1], open_hz = hiz > g.bbmax[2];
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
                if (ncell <= cell_budget || lev + 1 >= g.n_levels) break;
                s++, lev++;
            }
        }
        const float hs     = g.hf * (float)(1u << s);
        const float inv_nx = 1.0f / (float)max(nx, 1u), inv_ny = 1.0f / (float)max(ny, 1u);
        // CERT: voxels up to cert_margin beyond the largest radius are staged too, so that the region this pass
        // covers completely reaches that far beyond every query of the group (the corner voxels of the box)
        const float prune  = rmax_t + 4.f * g.slack + (CERT ? cert_margin : 0.f);
        const float prune2 = prune * prune;
        dbg_pass++, dbg_cells += ncell;

        // a repeated pass rescans voxels already seen: restart the k-list so that no neighbour
        // is inserted twice
#pragma unroll
        for (int j = 0; j < K; j++)
            if (grp) kd2[j] = INFINITY, kidx[j] = NONE_U32, kspos[j] = NONE_U32;
        if (CERT && grp) rej = INFINITY;
        float kth = kth_d2(kd2, knn);  // INFINITY for the lanes of this pass
        const float r2 = r * r;

        for (unsigned long long cb = 0; cb < ncell; cb += PL_CELLS)
        {
            // ---- PL_SB batches of 64 voxels: all their directory loads first (independent), then one prefix over the 256 ----------
            uint32_t cst[PL_SB], ccn[PL_SB];
#pragma unroll
            for (int u = 0; u < PL_SB; u++)
            {
                const unsigned long long cid = cb + 64ull * (unsigned long long)u + (unsigned long long)lane;
                uint32_t                 cnt = 0, start = 0;
                if (cid < ncell)
                {
                    uint32_t ix, iy, iz;
                    if (ncell <= 65536ull)
                    {   // (exact for these sizes: (c + 0.5) / n is at least 0.5 / n away from an integer; a 64-bit division is ~100 instructions)
                        const uint32_t c32 = (uint32_t)cid, row = (uint32_t)(((float)c32 + 0.5f) * inv_nx);
                        ix = c32 - row * nx, iz = (uint32_t)(((float)row + 0.5f) * inv_ny), iy = row - iz * ny;
                    }
                    else
                        ix = (uint32_t)(cid % nx), iy = (uint32_t)((cid / nx) % ny), iz = (uint32_t)(cid / ((unsigned long long)nx * ny));
                    const uint32_t cx = cx0 + ix, cy = cy0 + iy, cz = cz0 + iz;
                    const float vx0 = g.ox + (float)cx * hs, vy0 = g.oy + (float)cy * hs,
                                vz0 = g.oz + (float)cz * hs;
                    const float dx = fmaxf(0.f, fmaxf(vx0 - qhx, qlx - (vx0 + hs)));
                    const float dy = fmaxf(0.f, fmaxf(vy0 - qhy, qly - (vy0 + hs)));
                    const float dz = fmaxf(0.f, fmaxf(vz0 - qhz, qlz - (vz0 + hs)));
                    if (dx * dx + dy * dy + dz * dz <= prune2)
                    {
                        uint32_t e = 0;
                        if (voxel_range(g, lev, cx, cy, cz, start, e, false)) cnt = e - start;
                        else start = 0;
                    }
                }
                cst[u] = start, ccn[u] = cnt;
            }
            uint32_t total = 0;
#pragma unroll
            for (int u = 0; u < PL_SB; u++)
            {
                const uint32_t incl = wave_incl_scan(ccn[u], lane);
                s_cstart[64 * u + lane] = cst[u];
                s_coff[64 * u + lane]   = total + incl - ccn[u];
                total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            dbg_cand += total;
            if (lane == 63) s_coff[PL_CELLS] = total;
            __syncthreads();
            // ---- scan of the staged candidates.  With 64 different queries in a wave nearly every candidate
            //      enters SOMEBODY's list, so an insertion chain behind a per-candidate branch runs for all of
            //      them (~100 instructions each).  Instead a lane only notes its hits (candidates within its
            //      current k-th distance) in a small LDS queue; the chains run when a queue is full or the
            //      bucket ends, with most lanes busy.  A queued candidate that no longer qualifies by then falls
            //      through the chain unchanged.
            uint32_t hq = 0;  // hits queued by this lane
            auto     flush = [&]()
            {
                const uint32_t hmax = (uint32_t)wave_max((float)hq);  // hq <= PL_HITQ: exact in fp32
                for (uint32_t e = 0; e < hmax; e++)
                {
                    const bool     mine = e < hq;
                    const uint32_t jj   = mine ? s_hit[e * 64 + lane] : 0u;
                    const float4   c    = s_cand[jj];
                    float          cd   = mine ? dist2(qx, qy, qz, c.x, c.y, c.z) : INFINITY;
                    uint32_t       ci = mine ? __float_as_uint(c.w) : NONE_U32, cs = s_spos[jj];
#pragma unroll
                    for (int q = 0; q < K; q++)
                    {
                        const bool     less = mine && ((cd < kd2[q]) || (cd == kd2[q] && ci < kidx[q]));
                        const float    td   = kd2[q];
                        const uint32_t ti = kidx[q], ts = kspos[q];
                        kd2[q]   = less ? cd : td;
                        kidx[q]  = less ? ci : ti;
                        kspos[q] = less ? cs : ts;
                        cd = less ? td : cd, ci = less ? ti : ci, cs = less ? ts : cs;
                    }
                    if (CERT) rej = fminf(rej, cd);  // what fell off the end: the candidate itself or the old last entry (inf: nothing)
                }
#pragma unroll
                for (int q = 0; q < K; q++)  // only the knn nearest are kept
                    if (q >= (int)knn)
                    {
                        if (CERT) rej = fminf(rej, kd2[q]);
                        kd2[q] = INFINITY, kidx[q] = NONE_U32, kspos[q] = NONE_U32;
                    }
                kth = kth_d2(kd2, knn);
                hq  = 0;
            };
            auto scan = [&](const uint32_t m)
            {
                // (uniform trip count: the flush inside is a wave-wide operation -- DPP / readlane
                //  reductions -- and must be reached by every lane together)
                for (uint32_t j0 = 0; j0 < m; j0 += 64 / Q)
                {
                    const uint32_t j  = j0 + (uint32_t)(lane / Q);
                    const bool     jv = j < m;
                    const float4   c  = s_cand[jv ? j : 0u];
                    const float    d2 = dist2(qx, qy, qz, c.x, c.y, c.z);
                    // (beyond the pass radius a candidate is of no use: the pass ends the search only if the k-th lies
                    //  inside it, and otherwise the list restarts with a larger radius.  The staged cube holds 20..100 x
                    //  the points of the ball; without this bound they fill the lists first and run the insertion chains)
                    const bool     in = (STRICT ? (d2 < lim2) : (d2 <= lim2)) && d2 <= r2;
                    if (jv && grp && in && d2 <= kth) s_hit[hq * 64 + lane] = j, hq++;
                    else if (CERT && jv && grp) rej = fminf(rej, d2);
                    if (__ballot(hq >= (uint32_t)PL_HITQ) != 0ull) flush();
                }
                if (__ballot(hq > 0u) != 0ull) flush();
            };
            // ---- staging.  The voxels' points are fetched 64 x PL_STAGE at a time (PL_STAGE independent loads per lane in flight: the
            //      tiles that set the kernel's span stage thousands of points and wait for every round trip alone), and
            //      only those inside the group's box are kept: the voxel-aligned cube holds 20..100 x the points of the
            //      search cubes, and a point outside the box is beyond the radius of every query of the group.  The
            //      survivors collect in s_cand over as many fetches as fit; the scan runs when the buffer is full.
            uint32_t                 ns    = 0;  // candidates in s_cand (wave-uniform)
            const unsigned long long below = (1ull << lane) - 1ull;
            for (uint32_t base = 0; base < total; base += 64u * PL_STAGE)
            {
                const uint32_t m = min(64u * (uint32_t)PL_STAGE, total - base);
                if (ns + m > (uint32_t)PL_CAP)
                {
                    __syncthreads();
                    scan(ns);
                    __syncthreads();
                    ns = 0;
                }
                uint32_t src[PL_STAGE];
#pragma unroll
                for (int u = 0; u < PL_STAGE; u++)
                {
                    const uint32_t gt = base + min((uint32_t)lane + 64u * (uint32_t)u, m - 1u);
                    int            lo = 0, hi = PL_CELLS - 1;
#pragma unroll
                    for (int it = 0; it < 8; it++)
                    {
                        const int mid = (lo + hi + 1) >> 1;
                        if (s_coff[mid] <= gt) lo = mid;
                        else hi = mid - 1;
                    }
                    src[u] = s_cstart[lo] + (gt - s_coff[lo]);
                }
                float4 c[PL_STAGE];
#pragma unroll
                for (int u = 0; u < PL_STAGE; u++) c[u] = g.pts[src[u]];
                if (touched)
                {
#pragma unroll
                    for (int u = 0; u < PL_STAGE; u++)
                        if ((uint32_t)lane + 64u * (uint32_t)u < m) touched[src[u]] = 1;
                }
#pragma unroll
                for (int u = 0; u < PL_STAGE; u++)
                {
                    const bool keep = ((uint32_t)lane + 64u * (uint32_t)u < m) && c[u].x >= flx && c[u].x <= fhx && c[u].y >= fly &&
                                      c[u].y <= fhy && c[u].z >= flz && c[u].z <= fhz;
                    const unsigned long long bal = __ballot(keep);
                    if (keep)
                    {
                        const uint32_t pos = ns + (uint32_t)__popcll(bal & below);
                        s_cand[pos] = c[u], s_spos[pos] = src[u];
                    }
                    ns += (uint32_t)__popcll(bal);
                }
            }
            __syncthreads();
            if (ns) scan(ns);
            __syncthreads();
        }
        // ---- merge the k-lists of the lanes that hold the same query (partners differ in the bits
        //      >= Q of the lane number; both end up with the merged list) ------------------------
        if (Q < 64)
        {
#pragma unroll
            for (int off = Q; off < 64; off <<= 1)
            {
                float    od[K];
                uint32_t oi[K], os[K];
#pragma unroll
                for (int q = 0; q < K; q++)
                    od[q] = __shfl_xor(kd2[q], off, 64), oi[q] = __shfl_xor(kidx[q], off, 64),
                    os[q] = __shfl_xor(kspos[q], off, 64);
#pragma unroll
                for (int e = 0; e < K; e++)
                {
                    float    cd = od[e];
                    uint32_t ci = oi[e], cs = os[e];
                    if (__ballot(grp && ci != NONE_U32) == 0ull) break;  // lists are sorted: nothing further
#pragma unroll
                    for (int q = 0; q < K; q++)
                    {
                        // only the lanes of this pass: a finished query's list is already the
                        // merged one on both partners (merging again would duplicate it)
                        const bool     less = grp && ((cd < kd2[q]) || (cd == kd2[q] && ci < kidx[q]));
                        const float    td   = kd2[q];
                        const uint32_t ti = kidx[q], ts = kspos[q];
                        kd2[q] = less ? cd : td, kidx[q] = less ? ci : ti, kspos[q] = less ? cs : ts;
                        cd = less ? td : cd, ci = less ? ti : ci, cs = less ? ts : cs;
                    }
                    if (CERT && grp) rej = fminf(rej, cd);
                }
                if (CERT)
                {
                    // entries of the partner's list the early exit above did not visit are all NONE (sorted lists);
                    // the partner's own rejects count for this query too
                    const float orej = __shfl_xor(rej, off, 64);
                    if (grp) rej = fminf(rej, orej);
                }
#pragma unroll
                for (int q = 0; q < K; q++)
                    if (grp && q >= (int)knn)
                    {
                        if (CERT) rej = fminf(rej, kd2[q]);
                        kd2[q] = INFINITY, kidx[q] = NONE_U32, kspos[q] = NONE_U32;
                    }
            }
        }
        if (grp)
        {
            const float gr  = r * (1.0f - 1.0f / 1024.0f) - g.slack;
            kth             = kth_d2(kd2, knn);  // INFINITY while fewer than knn are known
            if (r >= rmax || (gr > 0.f && kth < gr * gr))
            {
                done = true;
                if (CERT)
                {
                    // What this pass staged: every voxel of the aligned box [cx0, cx0 + nx) x .. that lies within `prune`
                    // of the group's queries.  A point that was NOT staged is therefore beyond a face of that box or
                    // farther than prune - 4 slack; a staged one was tested: kept, or in rej.
                    float cd = r;  // (empty box: the cube of half-width r around the query holds no map point)
                    if (!empty_box)
                    {
                        const float bx0 = g.ox + (float)cx0 * hs, bx1 = g.ox + (float)(cx0 + nx) * hs;
                        const float by0 = g.oy + (float)cy0 * hs, by1 = g.oy + (float)(cy0 + ny) * hs;
                        const float bz0 = g.oz + (float)cz0 * hs, bz1 = g.oz + (float)(cz0 + nz) * hs;
                        cd = prune - 4.f * g.slack;
                        cd = fminf(cd, fminf(open_lx ? INFINITY : qx - bx0, open_hx ? INFINITY : bx1 - qx));
                        cd = fminf(cd, fminf(open_ly ? INFINITY : qy - by0, open_hy ? INFINITY : by1 - qy));
                        cd = fminf(cd, fminf(open_lz ? INFINITY : qz - bz0, open_hz ? INFINITY : bz1 - qz));
                        // ... or outside the staging filter's box
                        cd = fminf(cd, fminf(fminf(qx - flx, fhx - qx), fminf(fminf(qy - fly, fhy - qy), fminf(qz - flz, fhz - qz))));
                    }
                    cd  = fmaxf(cd - 2.f * g.slack, gr);  // gr: the radius the search itself relies on
                    rej = fmaxf(0.f, fminf(sqrtf(rej), cd));  // (a finished lane is in no later pass: rej is free)
                }
            }
            else
            {
                const float rn = (kth < INFINITY)
                                     ? sqrtf(kth) * (1.0f + 1.0f / 512.0f) + 4.f * g.slack
                                     : 2.0f * r;
                r = fminf(fmaxf(rn, r * 1.0009765625f), rmax);
            }
        }
    }
    if (CERT && lb_out) *lb_out = rej;
    if (tl_info && lane == 0) *tl_info = (dbg_pass << 48) | ((dbg_cells & 0xFFFFull) << 32) | (dbg_cand & 0xFFFFFFFFull);
    if (cand_out) *cand_out = (uint32_t)min(dbg_cand, 0xFFFFFFFFull);
    if (dbg && lane == 0)
    {
        const unsigned long long dt = (unsigned long long)((long long)wall_clock64() - dbg_t0);
        atomicAdd(&dbg[0], 1ull), atomicAdd(&dbg[1], dbg_pass), atomicAdd(&dbg[2], dbg_cand), atomicAdd(&dbg[3], dt);
        atomicMax(&dbg[4], dbg_pass), atomicMax(&dbg[5], dbg_cand), atomicMax(&dbg[6], dt), atomicAdd(&dbg[7], dbg_cells);
        atomicMax(&dbg[8], dbg_cells);
        // duration histogram (log2 of 100 MHz ticks) and the slowest tile's {ticks, passes, candidates}
        atomicAdd(&dbg[16 + min(23, 63 - (int)__clzll((long long)(dt | 1ull)))], 1ull);
        atomicMax(&dbg[49], (dt << 40) | ((dbg_pass & 0xFFull) << 32) | (dbg_cand & 0xFFFFFFFFull));
    }
}

// transform one local point per lane and publish the tile's bounding box of the visited points
template <int Q>
__device__ __forceinline__ void transform_tile(const PoseRt& pose, const float4* lpts, uint32_t n_l,
                                               const uint32_t* rank, float* tile_bbox, int lane,
                                               bool& valid, bool& visited, uint32_t& orig,
                                               uint32_t& vrank, float& qx, float& qy, float& qz)
{
    const uint32_t tile = blockIdx.x;
    const uint32_t qi   = tile * Q + (uint32_t)(lane % Q);  // = place in the sorted copy
    valid               = qi < n_l;
    float4 lp           = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = lpts[qi];
    orig    = __float_as_uint(lp.w);
    vrank   = orig;
    visited = valid;
    if (rank && valid) vrank = rank[orig], visited = vrank != NONE_U32;
    compose_point_f(pose, lp.x, lp.y, lp.z, qx, qy, qz);
    const float bx0 = wave_min_nn((visited && qx == qx) ? qx : INFINITY), by0 = wave_min_nn((visited && qy == qy) ? qy : INFINITY),
                bz0 = wave_min_nn((visited && qz == qz) ? qz : INFINITY);
    const float bx1 = wave_max_nn((visited && qx == qx) ? qx : -INFINITY), by1 = wave_max_nn((visited && qy == qy) ? qy : -INFINITY),
                bz1 = wave_max_nn((visited && qz == qz) ? qz : -INFINITY);
    if (lane == 0)
    {
        float* o = tile_bbox + (size_t)tile * 6;
        o[0] = bx0, o[1] = by0, o[2] = bz0, o[3] = bx1, o[4] = by1, o[5] = bz1;
    }
}

// estimate_points_eigen (estimate_points_eigen.cpp:27-123) over the first m of K points, in
// order: fp32 mean, fp32 products accumulated in fp64 and scaled by the fp32 1/n; the planarity
// test e0 < thr*e1 && e0 < thr*e2 (Matcher_Adaptive.cpp:237-238); unit normal = eigenvector 0
// with its largest |component| positive.  Operation order = oracle/mp2p_oracle.c.
template <int K>
__device__ __forceinline__ bool plane_of_points(const float (&px)[K], const float (&py)[K],
                                                const float (&pz)[K], int m, double eigThr,
                                                double (&n)[3], float& mx, float& my, float& mz)
{
    mx = 0.f, my = 0.f, mz = 0.f;
#pragma unroll
    for (int j = 0; j < K; j++)
        if (j < m) mx = fadd(mx, px[j]), my = fadd(my, py[j]), mz = fadd(mz, pz[j]);
    const float inv_n = 1.0f / (float)m;
    mx = fmul(mx, inv_n), my = fmul(my, inv_n), mz = fmul(mz, inv_n);
    double a00 = 0, a10 = 0, a20 = 0, a11 = 0, a21 = 0, a22 = 0;
#pragma unroll
    for (int j = 0; j < K; j++)
    {
        if (j < m)
        {
            const float ax = fsub(px[j], mx), ay = fsub(py[j], my), az = fsub(pz[j], mz);
            a00 = dadd(a00, (double)fmul(ax, ax));
            a10 = dadd(a10, (double)fmul(ax, ay));
            a20 = dadd(a20, (double)fmul(ax, az));
            a11 = dadd(a11, (double)fmul(ay, ay));
            a21 = dadd(a21, (double)fmul(ay, az));
            a22 = dadd(a22, (double)fmul(az, az));
        }
    }
    const double sc = (double)inv_n;
    a00 = dmul(a00, sc), a10 = dmul(a10, sc), a20 = dmul(a20, sc);
    a11 = dmul(a11, sc), a21 = dmul(a21, sc), a22 = dmul(a22, sc);
    const double cov[9] = {a00, a10, a20, a10, a11, a21, a20, a21, a22};
    double       ev[3];
    jacobi3(cov, ev, n);
    if (!(ev[0] < eigThr * ev[2] && ev[0] < eigThr * ev[1])) return false;
    const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= nn, n[1] /= nn, n[2] /= nn;
    int big = 0;
    if (fabs(n[1]) > fabs(n[big])) big = 1;
    if (fabs(n[2]) > fabs(n[big])) big = 2;
    if (n[big] < 0) n[0] = -n[0], n[1] = -n[1], n[2] = -n[2];
    return true;
}

constexpr int PL_Q = 32;  // queries per wave (2 candidate slices)

// Search and plane fit are two kernels: the search holds a query on 64 / Q lanes, the fit needs one
// lane per query (fp64 Jacobi, thousands of instructions) -- fused, the fit ran on Q of 64 lanes.
// Q = 32 for large local layers (a staged bucket serves 32 queries); Q = 8 when the layer is too small
// to fill the chip with 32-query tiles (a KITTI scan of 120 k points = 3 750 tiles for 5 120 wave
// slots: the kernel then lasts as long as its slowest tile; 8-query tiles cut that tile's work 4x).
// (register budget: the occupancy the kernel had before the warm start: 5 / 4 / 3 / 2 waves per SIMD for K = 5 / 8 / 12 / 16)
// one tile: Q consecutive entries of a segment of a pending list (pt2pl_cert_kernel)
template <int K, int Q, bool HARD>
__device__ __forceinline__ void pt2pl_tile_body(const PlArgs& a, const uint32_t* __restrict__ pend, uint32_t cnt, uint32_t k,
                                                uint32_t tile_id, float4* s_cand, uint32_t* s_spos, uint32_t* s_hit,
                                                uint32_t* s_cstart, uint32_t* s_coff)
{
    const GridView& g    = a.g;
    const int       lane = threadIdx.x;
    const unsigned long long tl0 = a.timeline ? wall_clock64() : 0ull;
    const uint32_t slot  = k * Q + (uint32_t)(lane % Q);
    const uint32_t ent   = slot < cnt ? pend[slot] : NONE_U32;  // (NONE: the padding of a block's part of the hard list)
    const bool     valid = ent != NONE_U32;
    const uint32_t qi    = valid ? ent : 0u;  // place in the sorted copy
    const float4   lp    = a.lpts[qi];
    const uint32_t orig  = __float_as_uint(lp.w);
    float          qx, qy, qz;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);
    const bool active = valid;  // visited, finite and not taken: checked by pt2pl_cert_kernel

    float r0 = a.r0;
    if (a.use_hint && active)
    {
        float ox, oy, oz;
        compose_point_f(a.prev_pose, lp.x, lp.y, lp.z, ox, oy, oz);
        const float disp = sqrtf(dist2(qx, qy, qz, ox, oy, oz));
        const float kp   = a.kth_io[qi];
        if (kp < INFINITY) r0 = sqrtf(kp) * (1.0f + 1.0f / 512.0f) + disp * 1.00001f + 4.f * g.slack;  // NaN / inf: the full radius
    }
    float    kd2[K];
    uint32_t kidx[K], kspos[K];
    float    lb = 0.f;
    uint32_t ncand = 0;
    knn_search<K, false, Q, true>(g, lane, qx, qy, qz, active, a.radSq, a.rad_cert, r0, a.knn,
                                  a.grp_factor, a.cell_budget, a.dbg, s_hit, s_cand, s_spos, s_cstart, s_coff, kd2, kidx, kspos,
                                  a.grp_min, &lb, a.cert_margin, a.timeline ? a.timeline + 2 * (size_t)a.timeline_n + tile_id : nullptr,
                                  &ncand, a.touched);
    if (a.timeline && lane == 0) a.timeline[2 * (size_t)tile_id] = tl0, a.timeline[2 * (size_t)tile_id + 1] = wall_clock64();
    // the neighbour list (sorted positions, ascending (d2, idx); NONE beyond its end) for the fit kernel
    if (!valid || lane >= Q) return;
    a.kth_io[qi]  = kth_d2(kd2, a.knn);
    a.lb_io[qi]   = lb;
    a.cost_io[qi] = min(ncand, 0x7FFFFFFFu) | (HARD ? 0x80000000u : 0u);  // what this query's tile staged: the next call's scheduling hint
    uint32_t* o   = a.out_knn + (size_t)orig * K;
#pragma unroll
    for (int j = 0; j < K; j++) o[j] = (kidx[j] != NONE_U32) ? kspos[j] : NONE_U32;
}

// The kernel lasts as long as its slowest tiles if they start late (C3, 15 000 tiles of 8 on 5 120 wave slots: a first
// round of 110 us, then a tail to 385 us made of tiles that take 200-300 us -- dense vegetation, several passes -- and
// happened to start in the second round: 37 % of the wave slots busy on average).  The queries whose tile was slow at
// the previous call (PlArgs::cost_io: its staged candidates, which its duration follows with r = 0.94) are therefore listed
// apart (the hard class), cut into tiles of HQ < Q queries -- 64 / HQ lanes per query, half the passes -- and dispatched
// FIRST: the workgroups [0, n_hard_tiles) of the grid.
template <int K, int Q, int HQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 8 ? 4 : (K <= 12 ? 3 : 2), 8))) void pt2pl_tile_kernel(const PlArgs a, const uint32_t n_hard_tiles)
{
    __shared__ float4   s_cand[PL_CAP];
    __shared__ uint32_t s_spos[PL_CAP];
    __shared__ uint32_t s_hit[PL_HITQ * 64];
    __shared__ uint32_t s_cstart[PL_CELLS];
    __shared__ uint32_t s_coff[PL_CELLS + 1];
    if (blockIdx.x < n_hard_tiles)
    {
        const uint32_t cnt = min(a.list_cnt[1], a.hard_cap);
        if (blockIdx.x * HQ >= cnt) return;  // wave-uniform
        pt2pl_tile_body<K, HQ, true>(a, a.hard_list, cnt, blockIdx.x, blockIdx.x, s_cand, s_spos, s_hit, s_cstart, s_coff);
    }
    else
    {
        const uint32_t t   = blockIdx.x - n_hard_tiles;
        const uint32_t cnt = min(a.list_cnt[0], a.pend_cap);
        if (t * Q >= cnt) return;  // wave-uniform
        pt2pl_tile_body<K, Q, false>(a, a.pend, cnt, t, blockIdx.x, s_cand, s_spos, s_hit, s_cstart, s_coff);
    }
}

// ---- round 6: the same tile through knn_sel_search (nn_pl_seltile.hip): 32 queries per tile in both classes.  A workgroup is
//      four waves.  The workgroups [0, n_hard_tiles) serve ONE tile of the hard class each (the queries whose tile staged many
//      candidates at the previous call), the four waves dealing its selected voxels between them (W = 4); every other workgroup
//      serves FOUR tiles of the easy class, a wave each (W = 1).  W is a run-time value of the workgroup, so both share one body
//      (two instantiations would not fit the instruction cache together).  A KITTI scan is 3 750 tiles for 3 072 wave slots: the
//      kernel lasts as long as its longest tile, which four waves cut to a third; four waves for EVERY tile (the first version)
//      quadrupled the waves and the redundant listing with them.
//      Register budget: 3 waves per SIMD up to K = 8 (168 registers; the LDS, 12.9 KB per wave, allows no more), 2 beyond.
template <int K, bool INSTR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K <= 8 ? 3 : 2, K <= 8 ? 3 : 2))) void pt2pl_seltile_kernel(const PlArgs a, const uint32_t n_hard_tiles)
{
    // (sized at the launch: one PsLds per wave of the workgroup -- four with a hard class, one without: a large layer has no use for
    //  the hard class, and four independent tiles per workgroup hold their LDS and wave slots until the slowest of the four is done:
    //  C5 5.3 -> 6.0 ms)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    PsLds* const lds = reinterpret_cast<PsLds*>(lds_raw);
    constexpr int    Q = 32;
    const GridView&  g    = a.g;
    const int        lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool       hard = blockIdx.x < n_hard_tiles;
    // (a hard tile by the four waves of its workgroup -- or, launched with one wave per workgroup (large layers: the hard class is only
    //  the ORDER there, long tiles first), by that wave alone)
    const int        W = (hard && blockDim.x == 256u) ? 4 : 1, wv = W == 4 ? wave : 0;
    const uint32_t*  pend;
    uint32_t         cnt, k;
    if (hard)
    {
        cnt = min(a.list_cnt[1], a.hard_cap), k = blockIdx.x, pend = a.hard_list;
    }
    else
    {
        cnt = min(a.list_cnt[0], a.pend_cap), k = (blockIdx.x - n_hard_tiles) * (blockDim.x >> 6) + (uint32_t)wave, pend = a.pend;
    }
    if (k * Q >= cnt) return;  // hard: the whole workgroup; easy: this wave (its tile runs without workgroup barriers)
    const uint32_t tile_id = hard ? blockIdx.x : n_hard_tiles + k;
    const unsigned long long tl0 = a.timeline ? wall_clock64() : 0ull;  // (profiling level 4; the plain build pays a uniform null test)
    const uint32_t slot  = k * Q + (uint32_t)(lane % Q);
    const uint32_t ent   = slot < cnt ? pend[slot] : NONE_U32;  // (NONE: the padding of a block's part of a list)
    const bool     valid = ent != NONE_U32;
    const uint32_t qi    = valid ? ent : 0u;  // place in the sorted copy
    const float4   lp    = a.lpts[qi];
    const uint32_t orig  = __float_as_uint(lp.w);
    float          qx, qy, qz;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);
    const bool active = valid;  // visited, finite and not taken: checked by pt2pl_cert_kernel

    float r0 = a.r0;
    if (a.use_hint && active)
    {
        float ox, oy, oz;
        compose_point_f(a.prev_pose, lp.x, lp.y, lp.z, ox, oy, oz);
        const float disp = sqrtf(dist2(qx, qy, qz, ox, oy, oz));
        const float kp   = a.kth_io[qi];
        if (kp < INFINITY) r0 = sqrtf(kp) * (1.0f + 1.0f / 512.0f) + disp * 1.00001f + 4.f * g.slack;  // NaN / inf: the full radius
    }
    float    kd2[K];
    uint32_t kspos[K];
    float    lb = 0.f;
    uint32_t ncand = 0;
    knn_sel_search<K, false, true, INSTR>(g, W, lane, wv, &lds[wave], hard ? &lds[0] : &lds[wave], qx, qy, qz, active, a.radSq, a.rad_cert, r0, a.knn,
                                          a.grp_factor, a.grp_min, a.grp_all_bricks, a.cert_margin, kd2, kspos, &lb, &ncand, a.dbg,
                                          a.timeline ? a.timeline + 2 * (size_t)a.timeline_n + tile_id : nullptr, a.touched, a.sol);
    if (a.timeline && lane == 0 && wv == 0) a.timeline[2 * (size_t)tile_id] = tl0, a.timeline[2 * (size_t)tile_id + 1] = wall_clock64();
    if (!valid || lane >= Q || wv != 0) return;
    a.kth_io[qi]  = ps_kth(kd2, a.knn);
    a.lb_io[qi]   = lb;
    a.cost_io[qi] = min(ncand, 0x7FFFFFFFu) | (hard ? 0x80000000u : 0u);  // what this query's tile staged: the next call's scheduling hint
    uint32_t* o   = a.out_knn + (size_t)orig * K;
#pragma unroll
    for (int j = 0; j < K; j++) o[j] = kspos[j];
}

// ---- certificate + query list: one thread per local point (Morton order), see PlArgs::lb_io ------------------------
// Also the per-point checks of the matcher's loop (Matcher_Point2Plane.cpp:76-85: visited, finite, not taken) and the
// bounding boxes of the transformed points (one per wave).  A query that is not certified is appended to its
// block's segment of the pending list, in Morton order (deterministic: block scan, no global atomics).
template <int K, int Q, int HQ>
__global__ __launch_bounds__(PL_CB) void pt2pl_cert_kernel(const PlArgs a)
{
    __shared__ uint32_t s_w[PL_CB / 64], s_h[PL_CB / 64], s_c[PL_CB / 64], s_at;
    const GridView& g    = a.g;
    const int       lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t  qi   = blockIdx.x * PL_CB + threadIdx.x;
    const bool      valid = qi < a.n_l;
    float4          lp    = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = a.lpts[qi];
    const uint32_t orig    = __float_as_uint(lp.w);
    bool           visited = valid;
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
            float* o = a.tile_bbox + ((size_t)blockIdx.x * (PL_CB / 64) + w) * 6;
            o[0] = bx0, o[1] = by0, o[2] = bz0, o[3] = bx1, o[4] = by1, o[5] = bz1;
        }
    }
    const float fin    = fadd(fadd(qx, qy), qz);
    bool        active = visited && (fin - fin == 0.0f);
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // Matcher_Point2Plane.cpp:83-85
    uint32_t* o = a.out_knn + (size_t)orig * K;
    if (valid && !active)
    {
#pragma unroll
        for (int j = 0; j < K; j++) o[j] = NONE_U32;
        a.kth_io[qi] = INFINITY, a.lb_io[qi] = 0.f;
    }
    bool search = active, certified = false;
    if (active && a.use_cert)
    {
        const float lb_old = a.lb_io[qi];
        if (lb_old > 0.f)
        {
            float ox, oy, oz;
            compose_point_f(a.prev_pose, lp.x, lp.y, lp.z, ox, oy, oz);
            const float disp = sqrtf(dist2(qx, qy, qz, ox, oy, oz));
            // bound of the outsiders at the new position.  The triangle inequality holds exactly between the fp32
            // positions the kernels work on; what is rounded are the three computed distances involved (relative 2^-22
            // of decimetre values: 1e-7 m) -- slack / 4 (2^-22 of the map's extent: 25 um on a 100 m map) covers that a
            // hundred times over.  (The first version took 4 slack = 0.4 mm per call, a sixth of the median gap
            // between the 5th and the 6th neighbour on the C3 scene: the bound decayed six times faster than it must.)
            const float lbn = lb_old - disp * 1.00001f - 0.25f * g.slack;
            uint32_t    ks[K], ki[K];
            float       kd[K];
            int         m = 0;
#pragma unroll
            for (int j = 0; j < K; j++)
            {
                ks[j] = o[j];
                kd[j] = INFINITY, ki[j] = NONE_U32;
                if (ks[j] != NONE_U32)
                {
                    m++;
                    const float4 p = g.pts[ks[j]];
                    const float  d = dist2(qx, qy, qz, p.x, p.y, p.z);
                    if (d <= a.radSq) kd[j] = d, ki[j] = __float_as_uint(p.w);  // a member that left the search radius drops out
                    else ks[j] = NONE_U32;
                }
            }
            // ascending (d2, idx), the order the search keeps its list in (NONE entries, d2 = inf, end up last)
#pragma unroll
            for (int i = 1; i < K; i++)
#pragma unroll
                for (int j = i; j > 0; j--)
                {
                    const bool sw = (kd[j] < kd[j - 1]) || (kd[j] == kd[j - 1] && ki[j] < ki[j - 1]);
                    const float    td = kd[j];
                    const uint32_t ti = ki[j], ts = ks[j];
                    kd[j] = sw ? kd[j - 1] : td, ki[j] = sw ? ki[j - 1] : ti, ks[j] = sw ? ks[j - 1] : ts;
                    kd[j - 1] = sw ? td : kd[j - 1], ki[j - 1] = sw ? ti : ki[j - 1], ks[j - 1] = sw ? ts : ks[j - 1];
                }
            int   m2   = 0;
            float kmax = 0.f;
#pragma unroll
            for (int j = 0; j < K; j++)
                if (ki[j] != NONE_U32) m2++, kmax = kd[j];
            bool ok;
            if (m == (int)a.knn) ok = (m2 == m) && lbn > 0.f && sqrtf(kmax) < lbn;  // the same knn points are the nearest
            // still nobody else in reach -- and nobody left: a member that drops out becomes an outsider NEARER than the
            // bound (it may come back within reach at the next call), so such a query goes through the search
            else ok = a.use_cert >= 2 && (m2 == m) && lbn > a.rad * 1.000001f + g.slack;
            if (ok)
            {
#pragma unroll
                for (int j = 0; j < K; j++) o[j] = ks[j];
                a.kth_io[qi] = (m2 == (int)a.knn) ? kmax : INFINITY;
                a.lb_io[qi]  = lbn;
                search = false, certified = true;
            }
        }
    }
    // two classes (pt2pl_tile_kernel): the queries whose tile staged many candidates at the previous call go to the layer's
    // hard list (a block reserves its part, padded to whole tiles of HQ, with one atomic; a part that does not fit any more
    // stays in the easy class), the rest to the block's own segment in Morton order
    bool hard = false;
    if (search && a.use_hint && a.hard_cand)
    {
        // cost: candidates its tile staged at the previous call; bit 31: that was a (smaller) tile of the hard class --
        // it stays there down to half the threshold, so that a query does not change class at every call
        const uint32_t c = a.cost_io[qi];
        hard = (c & 0x80000000u) ? (c & 0x7FFFFFFFu) >= a.hard_cand / 2u : c >= a.hard_cand;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long balh = __ballot(hard), balc = __ballot(certified);
    if (lane == 0) s_h[w] = (uint32_t)__popcll(balh), s_c[w] = (uint32_t)__popcll(balc);
    __syncthreads();
    uint32_t baseh = 0, totalh = 0, ncert = 0;
#pragma unroll
    for (int i = 0; i < PL_CB / 64; i++)
    {
        if (i < w) baseh += s_h[i];
        totalh += s_h[i], ncert += s_c[i];
    }
    const uint32_t padded = (totalh + (uint32_t)HQ - 1u) / (uint32_t)HQ * (uint32_t)HQ;
    if (threadIdx.x == 0)
    {
        uint32_t at = NONE_U32;
        if (padded) at = atomicAdd(&a.list_cnt[1], padded);
        s_at = at;
    }
    __syncthreads();
    uint32_t at = s_at;
    if (at != NONE_U32 && at + padded > a.hard_cap)
    {
        // no room: the part of the reservation that lies inside the list is blanked (the tile kernel clamps the count to
        // the capacity and would otherwise read what an earlier call left there); these queries stay in the easy class
        if (threadIdx.x < padded && at + threadIdx.x < a.hard_cap) a.hard_list[at + threadIdx.x] = NONE_U32;
        at = NONE_U32;
    }
    if (at == NONE_U32) hard = false, totalh = 0;
    else
    {
        if (hard) a.hard_list[at + baseh + (uint32_t)__popcll(balh & below)] = qi;
        if (threadIdx.x < padded - totalh) a.hard_list[at + totalh + threadIdx.x] = NONE_U32;
    }
    const bool               easy = search && !hard;
    const unsigned long long bal  = __ballot(easy);
    if (lane == 0) s_w[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < PL_CB / 64; i++)
    {
        if (i < w) base += s_w[i];
        total += s_w[i];
    }
    const uint32_t padded_e = (total + (uint32_t)Q - 1u) / (uint32_t)Q * (uint32_t)Q;
    if (threadIdx.x == 0) s_at = padded_e ? atomicAdd(&a.list_cnt[0], padded_e) : 0u;  // (capacity: every query + a tile per block)
    __syncthreads();
    const uint32_t at_e = s_at;
    if (easy) a.pend[at_e + base + (uint32_t)__popcll(bal & below)] = qi;
    if (threadIdx.x < padded_e - total) a.pend[at_e + total + threadIdx.x] = NONE_U32;
    if (threadIdx.x == 0)
    {
        if (a.cert_stat)
        {   // (64 lines of counters, summed when the statistics are read: on ONE line the three atomics of 19 500 blocks -- a 5 M-query
            //  layer -- were serialised, and WERE this kernel's 0.71 ms)
            unsigned long long* cs = a.cert_stat + (size_t)(blockIdx.x & 63u) * 16u;
            atomicAdd(&cs[0], (unsigned long long)ncert), atomicAdd(&cs[1], (unsigned long long)(total + totalh)), atomicAdd(&cs[2], (unsigned long long)totalh);
        }
    }
}

// ---- plane fit: one thread per local point (original index) -----------------------------------
template <int K>
__global__ __launch_bounds__(256) void pt2pl_fit_kernel(const PlArgs a, const float* __restrict__ lx,
                                                        const float* __restrict__ ly, const float* __restrict__ lz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && a.list_cnt) a.list_cnt[0] = 0u, a.list_cnt[1] = 0u;  // the search is over: both query lists are empty for the next call
    if (i >= a.n_l) return;
    const GridView& g = a.g;
    const uint32_t* o = a.out_knn + (size_t)i * K;
    uint32_t        ks[K];
    int             m = 0;
#pragma unroll
    for (int j = 0; j < K; j++)
    {
        ks[j] = o[j];
        if (ks[j] != NONE_U32) m++;  // all stored entries satisfy d2 <= radSq
    }
    unsigned char flag = 0;
    if (m >= (int)a.minPts && m >= 3)
    {
        float qx, qy, qz;
        compose_point_f(a.pose, lx[i], ly[i], lz[i], qx, qy, qz);
        float px[K], py[K], pz[K];
#pragma unroll
        for (int j = 0; j < K; j++)
        {
            if (j < m)
            {
                const float4 p = g.pts[ks[j]];
                px[j] = p.x, py[j] = p.y, pz[j] = p.z;
            }
        }
        float  mx, my, mz;
        double n[3];
        if (plane_of_points<K>(px, py, pz, m, a.eigThr, n, mx, my, mz))
        {
            const double c0 = (double)mx, c1 = (double)my, c2 = (double)mz;
            const double d  = -(n[0] * c0 + n[1] * c1 + n[2] * c2);
            const float  dist = (float)fabs(n[0] * (double)qx + n[1] * (double)qy + n[2] * (double)qz + d);
            if (!(dist > a.distThr))
            {
                flag       = 1;
                double* r7 = a.out_rec + (size_t)i * 7;
                r7[0] = n[0], r7[1] = n[1], r7[2] = n[2], r7[3] = d;
                r7[4] = c0, r7[5] = c1, r7[6] = c2;
            }
        }
    }
    a.out_flag[i] = flag;
}

// ---- Matcher_Points_DistanceThreshold with pairingsPerPoint > 1 ----------------------------------
// (Matcher_Points_DistanceThreshold.cpp:242-265): the k nearest in ascending d2, cut at the first
// d2 >= thr, a neighbour whose global point is already marked skipped.  Searching only d2 < thr
// gives the same list.  Slot (i, k) claims its global point with the word (visit rank * K + k):
// the sequential loop's "first claimant wins" (pairs.hip).
struct KnnArgs
{
    GridView      g;
    const float4* lpts;
    uint32_t      n_l;
    PoseRt        pose;
    float         maxDistSq, angSq, r0, grp_factor;
    uint32_t      knn, cell_budget;
    const unsigned char *local_taken, *global_taken;
    const uint32_t*      rank;
    unsigned long long*  claims;
    unsigned long long   claim_hi, local_offset;
    uint32_t*            out_spos;  // [n_l][knn] in the order of lpts
    float*               out_d2;
    float*               tile_bbox;
};

template <int K>
__global__ __launch_bounds__(64) void pt2pt_knn_kernel(const KnnArgs a)
{
    __shared__ float4   s_cand[PL_CAP];
    __shared__ uint32_t s_spos[PL_CAP];
    __shared__ uint32_t s_hit[PL_HITQ * 64];
    __shared__ uint32_t s_cstart[PL_CELLS];
    __shared__ uint32_t s_coff[PL_CELLS + 1];

    const GridView& g    = a.g;
    const int       lane = threadIdx.x;
    bool            valid, visited;
    uint32_t        orig, vrank;
    float           qx, qy, qz;
    transform_tile<PL_Q>(a.pose, a.lpts, a.n_l, a.rank, a.tile_bbox, lane, valid, visited, orig, vrank, qx, qy, qz);
    const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));  // :223-225
    const float thr    = fadd(a.maxDistSq, fmul(a.angSq, normSq));              // :256-257
    bool        active = visited && (normSq < INFINITY);
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // :218-220

    float    kd2[K];
    uint32_t kidx[K], kspos[K];
    knn_search<K, true, PL_Q>(g, lane, qx, qy, qz, active, thr, sqrtf(thr) * 1.002f + g.slack, a.r0, a.knn,
                        a.grp_factor, a.cell_budget, nullptr, s_hit, s_cand, s_spos, s_cstart, s_coff, kd2, kidx, kspos);
    if (!valid || lane >= PL_Q) return;
#pragma unroll
    for (int k = 0; k < K; k++)
    {
        if (k >= (int)a.knn) continue;
        bool acc = active && kidx[k] != NONE_U32;                             // d2 < thr by construction
        if (acc && a.global_taken && a.global_taken[kidx[k]]) acc = false;   // :98-101
        const size_t slot = (size_t)(blockIdx.x * (uint32_t)PL_Q + (uint32_t)lane) * a.knn + k;  // sorted order
        a.out_spos[slot]  = acc ? kspos[k] : NONE_U32;
        a.out_d2[slot]    = kd2[k];
        if (acc && a.claims)
            atomicMin(&a.claims[kspos[k]], a.claim_hi | ((a.local_offset + vrank) * a.knn + (unsigned)k));
    }
}

// ---- ordered compaction of the per-query plane slots ---------------------------------------------
constexpr int PC_THREADS = 256, PC_ITEMS = 4, PC_TILE = PC_THREADS * PC_ITEMS;

struct PlCompactArgs
{
    const unsigned char* flag;
    const double*        rec;
    uint32_t             n_l;      // slots = visited local points, in visiting order
    const uint32_t*      order;    // slot -> original local index (null: identity)
    unsigned long long   local_offset;
    const float*         local_bbox;
    float                gbb[6];
    float                margin;
    const float *        lx, *ly, *lz;
    uint32_t*            block_counts;
    unsigned long long*  counts;
    unsigned long long   cap;
    uint32_t*            o_lidx;
    double *             o_coef, *o_cen;
    float *              o_lx, *o_ly, *o_lz;
    unsigned char*       ms_local;
};

__device__ __forceinline__ bool pl_bbox_overlap(const float* g, const float* l, float eps)
{
    for (int d = 0; d < 3; d++)
    {
        if (l[d] - eps > g[3 + d]) return false;
        if (l[3 + d] + eps < g[d]) return false;
    }
    return true;
}

__global__ __launch_bounds__(PC_THREADS) void pl_count_kernel(const PlCompactArgs a)
{
    __shared__ uint32_t s_w[PC_THREADS / 64];
    uint32_t            c = 0;
    if (pl_bbox_overlap(a.gbb, a.local_bbox, a.margin))
    {
        const uint32_t base = blockIdx.x * PC_TILE + threadIdx.x * PC_ITEMS;
#pragma unroll
        for (int k = 0; k < PC_ITEMS; k++)
            if (base + k < a.n_l && a.flag[a.order ? a.order[base + k] : base + k]) c++;
    }
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t t = 0;
        for (int w = 0; w < PC_THREADS / 64; w++) t += s_w[w];
        a.block_counts[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(PC_THREADS) void pl_write_kernel(const PlCompactArgs a)
{
    __shared__ uint32_t s_w[PC_THREADS / 64];
    if (!pl_bbox_overlap(a.gbb, a.local_bbox, a.margin)) return;
    const uint32_t base = blockIdx.x * PC_TILE + threadIdx.x * PC_ITEMS;
    bool           f[PC_ITEMS];
    uint32_t       c = 0;
#pragma unroll
    for (int k = 0; k < PC_ITEMS; k++)
    {
        f[k] = (base + k < a.n_l) && a.flag[a.order ? a.order[base + k] : base + k];
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
    for (int k = 0; k < PC_ITEMS; k++)
    {
        if (!f[k]) continue;
        const uint32_t i = a.order ? a.order[base + k] : base + k;
        if (dst < a.cap)
        {
            const double* r = a.rec + (size_t)i * 7;
            for (int q = 0; q < 4; q++) a.o_coef[dst * 4 + q] = r[q];
            for (int q = 0; q < 3; q++) a.o_cen[dst * 3 + q] = r[4 + q];
            a.o_lidx[dst] = (uint32_t)(a.local_offset + i);
            a.o_lx[dst] = a.lx[i], a.o_ly[dst] = a.ly[i], a.o_lz[dst] = a.lz[i];
            if (a.ms_local) a.ms_local[i] = 1;  // Matcher_Point2Plane.cpp:109
        }
        dst++;
    }
}

template <int K>
static void launch_k(const PlArgs& a, uint32_t q, const mp2p_hip_cloud* cloud, hipStream_t st, uint32_t sel_waves)
{
    const uint32_t n_blocks = (a.n_l + PL_CB - 1) / PL_CB;
    const bool classes = a.use_hint && a.hard_cand;
    if (sel_waves)
    {   // round 6: ball-rule selection + matrix-pipe prefilter, 32 queries per tile in both classes; a hard tile per workgroup of four
        // waves, four easy tiles per workgroup (pt2pl_seltile_kernel)
        const uint32_t nh  = classes ? a.hard_cap / 32u : 0u;
        const uint32_t tpw = (nh && sel_waves == 4u) ? 4u : 1u;  // waves (= easy tiles) per workgroup
        const dim3     grid(nh + (a.pend_cap / 32u + tpw - 1u) / tpw);
        const size_t   lds_bytes = sizeof(PsLds) * tpw;
        hipLaunchKernelGGL((pt2pl_cert_kernel<K, 32, 32>), dim3(n_blocks), dim3(PL_CB), 0, st, a);
        const bool instr = a.dbg != nullptr || a.touched != nullptr;  // profiling level 2 (level 4, the timeline: the plain build stamps {start, end} itself)
        if (instr) hipLaunchKernelGGL((pt2pl_seltile_kernel<K, true>), grid, dim3(64u * tpw), lds_bytes, st, a, nh);
        else hipLaunchKernelGGL((pt2pl_seltile_kernel<K, false>), grid, dim3(64u * tpw), lds_bytes, st, a, nh);
    }
    else if (q == 8)
    {
        const uint32_t nh = classes ? a.hard_cap / 4u : 0u;
        hipLaunchKernelGGL((pt2pl_cert_kernel<K, 8, 4>), dim3(n_blocks), dim3(PL_CB), 0, st, a);
        hipLaunchKernelGGL((pt2pl_tile_kernel<K, 8, 4>), dim3(nh + a.pend_cap / 8u), dim3(64), 0, st, a, nh);
    }
    else
    {
        const uint32_t nh = classes ? a.hard_cap / 8u : 0u;
        hipLaunchKernelGGL((pt2pl_cert_kernel<K, PL_Q, 8>), dim3(n_blocks), dim3(PL_CB), 0, st, a);
        hipLaunchKernelGGL((pt2pl_tile_kernel<K, PL_Q, 8>), dim3(nh + a.pend_cap / (uint32_t)PL_Q), dim3(64), 0, st, a, nh);
    }
    hipLaunchKernelGGL(pt2pl_fit_kernel<K>, dim3((a.n_l + 255) / 256), dim3(256), 0, st, a, cloud->x.p, cloud->y.p,
                       cloud->z.p);
}

// phase: 0 = the whole matcher; 1 = search + plane fit + the shard's bounding box (left in ctx->local_bbox);
// 2 = the compaction (a sharded caller all-reduces the box in between: mp2p_hip_step_sharded_pt2pl).
// local_offset: whole-layer index of this shard's first local point (Pairings carry whole-layer indices)
int launch_match_pt2pl(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                       const double pose[12], const mp2p_hip_pt2pl_params* prm,
                       mp2p_hip_mstate* ms, mp2p_hip_pairs* out, int phase = 0, unsigned long long local_offset = 0)
{
    const size_t   n_l     = cloud->n;
    // tile size: see pt2pl_tile_kernel (MP2P_HIP_TUNE pl_q = 8 / 32 forces one)
    // (round 2 took 32-query tiles above 400 k points; with the insertions bounded by the pass radius and the filtered
    //  staging the 8-query tile also wins at 1 M queries: 1.10..1.47 ms with 32, 0.75..0.88 ms with 8.  At 5 M queries
    //  with 30 % uniform outliers -- BASELINE C5 -- 32 is still 10 % ahead: 7.8 vs 8.7 ms per step)
    // round 6: the ball-rule / matrix-pipe kernel (nn_pl_seltile.hip) whenever the map has its level-0 occupancy bricks and the
    // search radius stays within what one query's voxel list addresses; else the box-rule kernel of rounds 2-5
    const float    cell0_  = map->view.hf * (float)(1u << map->view.shift0);
    const bool     sel     = (ctx->tune.pl_select < 0 ? n_l > 524288 : ctx->tune.pl_select != 0) && map->view.occ != nullptr && map->view.occ_off[0] != OCC_NONE &&
                             (float)prm->searchRadius * 1.01f + 0.05f <= PS_MAX_RADIUS_CELLS * cell0_;
    const uint32_t Q       = sel ? 32u : (ctx->tune.pl_q ? ctx->tune.pl_q : (n_l <= 2000000 ? 8u : (uint32_t)PL_Q));
    // the ball-rule kernel's hard class: small layers (their tiles do not fill the chip several times over) -- a hard tile by the FOUR
    // waves of a workgroup; large layers -- one wave per tile, the class is only the dispatch ORDER (the tiles that staged the most at
    // the previous call first: in Morton order the 150-190 us tiles of a 1 M-query layer started anywhere up to 230 us into a 470 us
    // kernel, the last quarter of which ran below half residency)
    const bool     sel_large = sel && n_l > 524288;
    const uint32_t sel_waves = sel ? ((sel_large || ctx->tune.pl_waves == 1) ? 1u : 4u) : 0u;
    const uint32_t n_cblocks = (uint32_t)((n_l + PL_CB - 1) / PL_CB);
    const uint32_t n_boxes   = n_cblocks * (PL_CB / 64);  // one bounding box per wave of pt2pl_cert_kernel
    const uint32_t Kcap    = prm->knn <= 5 ? 5u : prm->knn <= 8 ? 8u : prm->knn <= 12 ? 12u : 16u;
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)std::max(n_boxes, 1u) * 6));
    const uint32_t pend_cap = (uint32_t)(((n_l + Q - 1) / Q + n_cblocks) * Q);
    MP2P_TRY_HIP(ctx, ctx->pl_pend.ensure(std::max(pend_cap, 64u)));
    MP2P_TRY_HIP(ctx, ctx->pl_pend_cnt.ensure(2));
    // the hard class: at most an eighth of the layer (in whole tiles), and a grid prefix of at most 8192 workgroups
    const uint32_t hard_cap = sel_large ? (uint32_t)(std::max<size_t>(n_l / 16, 64) / 32u * 32u)
                                        : (uint32_t)std::min<size_t>(std::max<size_t>(n_l / (sel ? 4 : 8), 64), 8192u * 4u) / 32u * 32u;
    MP2P_TRY_HIP(ctx, ctx->pl_hard.ensure(hard_cap));
    MP2P_TRY_HIP(ctx, ctx->pl_lb.ensure(n_l ? n_l : 1));
    MP2P_TRY_HIP(ctx, ctx->pl_cost.ensure(n_l ? n_l : 1));
    if (!ctx->pl_cert_stat.p)
    {
        MP2P_TRY_HIP(ctx, ctx->pl_cert_stat.ensure(64 * 16));  // 64 lines of {certified, searched, hard} (pt2pl_cert_kernel)
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->pl_cert_stat.p, 0, 64 * 16 * sizeof(unsigned long long), ctx->stream));
    }
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    MP2P_TRY_HIP(ctx, ctx->pl_slots.ensure(n_l * (7 * sizeof(double) + 1) + 64));
    MP2P_TRY_HIP(ctx, ctx->pl_knn.ensure(n_l * Kcap));
    // layout: [n_l][7] doubles, then [n_l] flags
    double*        rec  = reinterpret_cast<double*>(ctx->pl_slots.p);
    unsigned char* flag = ctx->pl_slots.p + n_l * 7 * sizeof(double);

    PlArgs a;
    memset(&a, 0, sizeof(a));
    a.g = map->view, a.lpts = cloud->sorted.p, a.n_l = (uint32_t)n_l;
    for (int i = 0; i < 9; i++) a.pose.r[i] = pose[i];
    for (int i = 0; i < 3; i++) a.pose.t[i] = pose[9 + i];
    a.radSq   = (float)(prm->searchRadius * prm->searchRadius);
    a.rad     = (float)prm->searchRadius;
    a.distThr = (float)prm->distanceThreshold;
    const float cell0 = map->view.hf * (float)(1u << map->view.shift0);
    a.r0     = cell0 * (prm->initial_radius_cells > 0 ? prm->initial_radius_cells : 2.0f);
    a.eigThr = prm->planeEigenThreshold;
    a.grp_factor = PL_GROUP_FACTOR, a.cell_budget = PL_CELL_BUDGET;
    a.minPts = prm->minimumPlanePoints;
    a.knn    = prm->knn;
    a.local_taken = (ms && !prm->allowMatchAlreadyMatchedPoints) ? ms->local_taken.p : nullptr;
    a.rank     = cloud->n_visit ? cloud->rank.p : nullptr;
    a.out_flag = flag, a.out_rec = rec, a.tile_bbox = ctx->tile_bbox.p;
    a.out_knn  = ctx->pl_knn.p;
    // warm start: same map, cloud, neighbour count and radius as the previous call of this matcher on this context
    MP2P_TRY_HIP(ctx, ctx->pl_kth.ensure(n_l ? n_l : 1));
    a.kth_io   = ctx->pl_kth.p;
    a.use_hint = (ctx->pl_hint_map == map && ctx->pl_hint_cloud == cloud && ctx->pl_hint_n == n_l && ctx->pl_hint_knn == prm->knn &&
                  ctx->pl_hint_rad == prm->searchRadius && ctx->tune.pl_warm) ? 1 : 0;
    for (int i = 0; i < 9; i++) a.prev_pose.r[i] = ctx->pl_hint_pose[i];
    for (int i = 0; i < 3; i++) a.prev_pose.t[i] = ctx->pl_hint_pose[9 + i];
    // the warm-start / certificate state is committed only once the search kernels are enqueued: an error
    // return in between must not leave a hint whose pose belongs to a call that never ran (lb_io / kth_io / pl_knn would
    // still be an older pose's, the displacement understated and stale lists certified)
    if (phase != 2) ctx->pl_hint_map = nullptr;
    a.grp_min = 2.0f * cell0;
    a.lb_io = ctx->pl_lb.p, a.pend = ctx->pl_pend.p, a.pend_cap = pend_cap, a.cert_stat = ctx->pl_cert_stat.p;
    // the certificate is worth reading (and its margin worth staging) only after a small step (Tune::pl_cert_step_mm; the same
    // bound as the point matcher's: |T p - T' p| <= |t - t'| + |R - R'|_F |p| for every local point p).  A call that skips it still
    // leaves valid bounds (without the margin: weaker), and results never depend on it.
    bool cert_on = ctx->tune.pl_cert != 0;
    if (cert_on && ctx->tune.pl_cert_step_mm != 0u && a.use_hint)
    {
        double dt = 0, dr = 0;
        for (int i = 0; i < 3; i++) dt += (pose[9 + i] - ctx->pl_hint_pose[9 + i]) * (pose[9 + i] - ctx->pl_hint_pose[9 + i]);
        for (int i = 0; i < 9; i++) dr += (pose[i] - ctx->pl_hint_pose[i]) * (pose[i] - ctx->pl_hint_pose[i]);
        cert_on = std::sqrt(dt) + std::sqrt(dr) * (double)cloud->radius <= 1e-3 * (double)ctx->tune.pl_cert_step_mm;
    }
    a.use_cert = (a.use_hint && cert_on) ? ctx->tune.pl_cert : 0;
    // (the hard class of the round-6 kernel: for layers whose tiles do not fill the chip several times over -- up to 16 384 tiles)
    a.cost_io = ctx->pl_cost.p, a.hard_cand = sel ? (sel_large ? ctx->tune.pl_sel_hard_large : (ctx->tune.pl_waves == 1 ? 0u : ctx->tune.pl_sel_hard_cand)) : ctx->tune.pl_hard_cand;
    a.grp_all_bricks = (float)ctx->tune.grp_all_bricks;
    a.sol = (ctx->profiling != 0) ? ctx->tune.pl_sol : 0;
    a.hard_list = ctx->pl_hard.p, a.hard_cap = hard_cap, a.list_cnt = ctx->pl_pend_cnt.p;
    if (ctx->pl_hard_cnt_at != (const void*)a.list_cnt || ctx->pl_lists_dirty)
    {  // just allocated (or a call that did not get as far as its fit kernel): zero once; from then on the fit kernel leaves the counters zeroed
        MP2P_TRY_HIP(ctx, hipMemsetAsync(a.list_cnt, 0, 2 * sizeof(uint32_t), ctx->stream));
        ctx->pl_hard_cnt_at = a.list_cnt;
    }
    // the radius a search that finds fewer than knn points has covered: searchRadius + pl_cert_pad per mille, the room
    // the certificate of such a query has before a point outside its list could come into reach (0.2 % without it)
    a.rad_cert = a.rad * (1.0f + 0.001f * (float)(ctx->tune.pl_cert ? std::max(2u, ctx->tune.pl_cert_pad) : 2u)) + map->view.slack;
    a.cert_margin = cert_on ? 0.001f * (float)(sel ? ctx->tune.pl_sel_margin_mm : ctx->tune.pl_cert_margin_mm) : 0.f;
    a.dbg = nullptr, a.touched = nullptr;
    if (ctx->profiling == 2)
    {
        MP2P_TRY_HIP(ctx, ctx->counters.ensure(64));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64 * sizeof(unsigned long long), ctx->stream));
        a.dbg = ctx->counters.p;
        // N_g,touched of SURVEY.md 8d, exactly: a byte per map point, set by the staging loop (scratch slot 15)
        MP2P_TRY_HIP(ctx, ctx->scratch[15].ensure(map->n ? map->n : 1));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->scratch[15].p, 0, map->n, ctx->stream));
        a.touched = ctx->tune.pl_no_touch ? nullptr : ctx->scratch[15].p;
    }

    a.timeline = nullptr;
    ctx->timeline_tiles = ctx->timeline_singles = 0;
    if (ctx->profiling == 4 && phase != 2)
    {
        const size_t n_grid = (size_t)pend_cap / Q + (a.hard_cand ? (size_t)hard_cap / (sel ? 32 : (Q == 8 ? 4 : 8)) : 0);
        // (read back as 2 x 1.5 n_grid words: the probe knows that the last third is the per-tile info)
        const size_t n_rec = n_grid + (n_grid + 1) / 2;
        MP2P_TRY_HIP(ctx, ctx->timeline.ensure(2 * std::max<size_t>(n_rec, 1)));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->timeline.p, 0, 2 * n_rec * sizeof(unsigned long long), ctx->stream));
        a.timeline = ctx->timeline.p, a.timeline_n = (uint32_t)n_grid, ctx->timeline_tiles = n_rec;
    }
    ctx->pending_lane = 0;
    ctx->pending_pl   = (ctx->profiling == 1 || ctx->profiling == 2) ? 1 : 0;
    if (phase != 2)
    {
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    ctx->pl_lists_dirty = true;
    if (Kcap == 5) launch_k<5>(a, Q, cloud, ctx->stream, sel_waves);
    else if (Kcap == 8) launch_k<8>(a, Q, cloud, ctx->stream, sel_waves);
    else if (Kcap == 12) launch_k<12>(a, Q, cloud, ctx->stream, sel_waves);
    else launch_k<16>(a, Q, cloud, ctx->stream, sel_waves);
    if (hipPeekAtLastError() == hipSuccess)
    {
        ctx->pl_lists_dirty = false;
        ctx->pl_hint_map = map, ctx->pl_hint_cloud = cloud, ctx->pl_hint_n = n_l, ctx->pl_hint_knn = prm->knn, ctx->pl_hint_rad = prm->searchRadius;
        for (int i = 0; i < 12; i++) ctx->pl_hint_pose[i] = pose[i];
    }
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    {
        const int rc = launch_bbox_reduce(ctx, n_boxes);
        if (rc) return rc;
    }
    }
    if (phase == 1)
    {
        MP2P_TRY_HIP(ctx, hipGetLastError());
        return MP2P_HIP_OK;
    }

    const size_t   n_slots  = cloud->n_visit ? cloud->n_visit : n_l;
    const uint32_t n_blocks = (uint32_t)((n_slots + PC_TILE - 1) / PC_TILE);
    MP2P_TRY_HIP(ctx, ctx->block_counts.ensure(n_blocks ? n_blocks : 1));
    PlCompactArgs c;
    memset(&c, 0, sizeof(c));
    c.flag = flag, c.rec = rec, c.n_l = (uint32_t)n_slots, c.local_bbox = ctx->local_bbox.p;
    c.local_offset = local_offset;
    c.order = cloud->n_visit ? cloud->order.p : nullptr;
    for (int d = 0; d < 3; d++) c.gbb[d] = map->view.bbmin[d], c.gbb[3 + d] = map->view.bbmax[d];
    c.margin = (float)(prm->distanceThreshold + prm->bounding_box_intersection_check_epsilon);
    c.lx = cloud->x.p, c.ly = cloud->y.p, c.lz = cloud->z.p;
    c.block_counts = ctx->block_counts.p, c.counts = out->counts.p, c.cap = out->cap_pt2pl;
    c.o_lidx = out->pl_lidx.p, c.o_coef = out->pl_coef.p, c.o_cen = out->pl_cen.p;
    c.o_lx = out->pl_lx.p, c.o_ly = out->pl_ly.p, c.o_lz = out->pl_lz.p;
    c.ms_local = ms ? ms->local_taken.p : nullptr;
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    hipLaunchKernelGGL(pl_count_kernel, dim3(n_blocks), dim3(PC_THREADS), 0, ctx->stream, c);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                       ctx->block_counts.p, n_blocks, out->counts.p, c.cap,
                       (unsigned long long)n_l /* :54: pcLocal.size(), the whole layer */, 1);
    hipLaunchKernelGGL(pl_write_kernel, dim3(n_blocks), dim3(PC_THREADS), 0, ctx->stream, c);
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

template <int K>
static void launch_knn_k(const KnnArgs& a, uint32_t n_tiles, hipStream_t st)
{
    hipLaunchKernelGGL(pt2pt_knn_kernel<K>, dim3(n_tiles), dim3(64), 0, st, a);
}

// phase 1 of Matcher_Points_DistanceThreshold for pairingsPerPoint > 1 (<= 16)
int launch_nn_pt2pt_knn(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                        const double pose[12], const mp2p_hip_pt2pt_params* prm, mp2p_hip_mstate* ms)
{
    const size_t   n_l     = cloud->n;
    const uint32_t K       = prm->pairingsPerPoint;
    const uint32_t n_tiles = (uint32_t)((n_l + PL_Q - 1) / PL_Q);
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)n_tiles * 6));
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    MP2P_TRY_HIP(ctx, ctx->nn_spos.ensure(n_l * K));
    MP2P_TRY_HIP(ctx, ctx->nn_d2.ensure(n_l * K));
    KnnArgs a;
    memset(&a, 0, sizeof(a));
    a.g = map->view, a.lpts = cloud->sorted.p, a.n_l = (uint32_t)n_l;
    for (int i = 0; i < 9; i++) a.pose.r[i] = pose[i];
    for (int i = 0; i < 3; i++) a.pose.t[i] = pose[9 + i];
    a.maxDistSq = (float)(prm->threshold * prm->threshold);                 // :82
    const double ang = prm->thresholdAngularDeg * 3.14159265358979323846 / 180.0;
    a.angSq          = (float)(ang * ang);                                  // :83
    // the shipped (TBB) build searches with nn_radius_search(maxDistSq, ..., k) for pairingsPerPoint > 1
    // (:172-177): a neighbour is returned only if d2 < maxDistSq, so the angular term (which can only
    // raise the threshold above maxDistSq) never admits anything further
    if (prm->multi_search_radius_mode) a.angSq = 0.0f;
    const float cell0 = map->view.hf * (float)(1u << map->view.shift0);
    a.r0  = cell0 * (prm->initial_radius_cells > 0 ? prm->initial_radius_cells : 1.5f);
    a.knn = K;
    a.grp_factor = PL_GROUP_FACTOR, a.cell_budget = PL_CELL_BUDGET;
    a.local_taken  = (ms && !prm->allowMatchAlreadyMatchedPoints) ? ms->local_taken.p : nullptr;
    a.global_taken = (ms && !prm->allowMatchAlreadyMatchedGlobalPoints) ? ms->global_taken.p : nullptr;
    a.rank         = cloud->n_visit ? cloud->rank.p : nullptr;
    a.claims       = prm->allowMatchAlreadyMatchedGlobalPoints ? nullptr : map->claims.p;
    ctx->epoch++;
    a.claim_hi     = (~(unsigned long long)ctx->epoch) << 32;
    a.local_offset = prm->local_index_offset;
    a.out_spos = ctx->nn_spos.p, a.out_d2 = ctx->nn_d2.p, a.tile_bbox = ctx->tile_bbox.p;
    ctx->pending_lane = 0;
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (K <= 5) launch_knn_k<5>(a, n_tiles, ctx->stream);
    else if (K <= 8) launch_knn_k<8>(a, n_tiles, ctx->stream);
    else if (K <= 12) launch_knn_k<12>(a, n_tiles, ctx->stream);
    else launch_knn_k<16>(a, n_tiles, ctx->stream);
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    const int rc = launch_bbox_reduce(ctx, n_tiles);
    if (rc) return rc;
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

}  // namespace mp2p
