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

namespace mp2p
{
constexpr int      PL_CAP         = 256;
constexpr int      PL_HITQ        = 8;    // queued hits per lane before the insertion chains run
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
template <int K, bool STRICT, int Q>
__device__ __forceinline__ void knn_search(const GridView& g, int lane, float qx, float qy, float qz,
                                           bool active, float lim2, float rmax, float r0, uint32_t knn,
                                           float grp_factor, uint32_t cell_budget,
                                           unsigned long long* dbg, uint32_t* s_hit, float4* s_cand,
                                           uint32_t* s_spos, uint32_t* s_cstart,
                                           uint32_t* s_coff, float (&kd2)[K], uint32_t (&kidx)[K],
                                           uint32_t (&kspos)[K], float grp_min = 0.f)
{
    float r    = fminf(r0, rmax);
    bool  done = !active;
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
        const float prune  = rmax_t + 4.f * g.slack;
        const float prune2 = prune * prune;
        dbg_pass++, dbg_cells += ncell;

        // a repeated pass rescans voxels already seen: restart the k-list so that no neighbour
        // is inserted twice
#pragma unroll
        for (int j = 0; j < K; j++)
            if (grp) kd2[j] = INFINITY, kidx[j] = NONE_U32, kspos[j] = NONE_U32;
        float kth = kth_d2(kd2, knn);  // INFINITY for the lanes of this pass

        for (unsigned long long cb = 0; cb < ncell; cb += 64)
        {
            const unsigned long long cid = cb + lane;
            uint32_t                 cnt = 0, start = 0;
            if (cid < ncell)
            {
                const uint32_t ix = (uint32_t)(cid % nx), iy = (uint32_t)((cid / nx) % ny),
                               iz = (uint32_t)(cid / ((unsigned long long)nx * ny));
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
            const uint32_t incl  = wave_incl_scan(cnt, lane);
            const uint32_t total = __shfl(incl, 63, 64);
            dbg_cand += total;
            s_cstart[lane] = start;
            s_coff[lane]   = incl - cnt;
            if (lane == 63) s_coff[64] = total;
            __syncthreads();
            for (uint32_t base = 0; base < total; base += PL_CAP)
            {
                const uint32_t m = min((uint32_t)PL_CAP, total - base);
                for (uint32_t t = lane; t < m; t += 64)
                {
                    const uint32_t gt = base + t;
                    int            lo = 0, hi = 63;
                    while (lo < hi)
                    {
                        const int mid = (lo + hi + 1) >> 1;
                        if (s_coff[mid] <= gt) lo = mid;
                        else hi = mid - 1;
                    }
                    const uint32_t src = s_cstart[lo] + (gt - s_coff[lo]);
                    s_cand[t]          = g.pts[src];
                    s_spos[t]          = src;
                }
                __syncthreads();
                // ---- scan.  With 64 different queries in a wave nearly every candidate enters
                //      SOMEBODY's list, so an insertion chain behind a per-candidate branch runs for
                //      all of them (~100 instructions each).  Instead a lane only notes its hits
                //      (candidates within its current k-th distance) in a small LDS queue; the chains
                //      run when a queue is full or the bucket ends, with most lanes busy.  A queued
                //      candidate that no longer qualifies by then falls through the chain unchanged.
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
                    }
#pragma unroll
                    for (int q = 0; q < K; q++)  // only the knn nearest are kept
                        if (q >= (int)knn) kd2[q] = INFINITY, kidx[q] = NONE_U32, kspos[q] = NONE_U32;
                    kth = kth_d2(kd2, knn);
                    hq  = 0;
                };
                // (uniform trip count: the flush inside is a wave-wide operation -- DPP / readlane
                //  reductions -- and must be reached by every lane together)
                for (uint32_t j0 = 0; j0 < m; j0 += 64 / Q)
                {
                    const uint32_t j  = j0 + (uint32_t)(lane / Q);
                    const bool     jv = j < m;
                    const float4   c  = s_cand[jv ? j : 0u];
                    const float    d2 = dist2(qx, qy, qz, c.x, c.y, c.z);
                    const bool     in = STRICT ? (d2 < lim2) : (d2 <= lim2);
                    if (jv && grp && in && d2 <= kth) s_hit[hq * 64 + lane] = j, hq++;
                    if (__ballot(hq >= (uint32_t)PL_HITQ) != 0ull) flush();
                }
                if (__ballot(hq > 0u) != 0ull) flush();
                __syncthreads();
            }
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
                }
#pragma unroll
                for (int q = 0; q < K; q++)
                    if (grp && q >= (int)knn) kd2[q] = INFINITY, kidx[q] = NONE_U32, kspos[q] = NONE_U32;
            }
        }
        if (grp)
        {
            const float gr  = r * (1.0f - 1.0f / 1024.0f) - g.slack;
            kth             = kth_d2(kd2, knn);  // INFINITY while fewer than knn are known
            if (r >= rmax || (gr > 0.f && kth < gr * gr))
                done = true;
            else
            {
                const float rn = (kth < INFINITY)
                                     ? sqrtf(kth) * (1.0f + 1.0f / 512.0f) + 4.f * g.slack
                                     : 2.0f * r;
                r = fminf(fmaxf(rn, r * 1.0009765625f), rmax);
            }
        }
    }
    if (dbg && lane == 0)
    {
        const unsigned long long dt = (unsigned long long)((long long)wall_clock64() - dbg_t0);
        atomicAdd(&dbg[0], 1ull), atomicAdd(&dbg[1], dbg_pass), atomicAdd(&dbg[2], dbg_cand), atomicAdd(&dbg[3], dt);
        atomicMax(&dbg[4], dbg_pass), atomicMax(&dbg[5], dbg_cand), atomicMax(&dbg[6], dt), atomicAdd(&dbg[7], dbg_cells);
        atomicMax(&dbg[8], dbg_cells);
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
template <int K, int Q>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 5 ? 5 : (K <= 8 ? 4 : (K <= 12 ? 3 : 2)), 8))) void pt2pl_tile_kernel(const PlArgs a)
{
    __shared__ float4   s_cand[PL_CAP];
    __shared__ uint32_t s_spos[PL_CAP];
    __shared__ uint32_t s_hit[PL_HITQ * 64];
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[65];

    const GridView& g    = a.g;
    const int       lane = threadIdx.x;
    bool            valid, visited;
    uint32_t        orig, vrank;
    float           qx, qy, qz;
    transform_tile<Q>(a.pose, a.lpts, a.n_l, a.rank, a.tile_bbox, lane, valid, visited, orig, vrank, qx, qy, qz);
    const float fin    = fadd(fadd(qx, qy), qz);
    bool        active = visited && (fin - fin == 0.0f);
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // Matcher_Point2Plane.cpp:83-85

    const uint32_t qi = blockIdx.x * Q + (uint32_t)(lane % Q);  // place in the sorted copy
    float          r0 = a.r0;
    if (a.use_hint && active)
    {
        const float4 lp = a.lpts[qi];
        float        ox, oy, oz;
        compose_point_f(a.prev_pose, lp.x, lp.y, lp.z, ox, oy, oz);
        const float disp = sqrtf(dist2(qx, qy, qz, ox, oy, oz));
        const float kp   = a.kth_io[qi];
        if (kp < INFINITY) r0 = sqrtf(kp) * (1.0f + 1.0f / 512.0f) + disp * 1.00001f + 4.f * g.slack;  // NaN / inf: the full radius
    }
    float    kd2[K];
    uint32_t kidx[K], kspos[K];
    knn_search<K, false, Q>(g, lane, qx, qy, qz, active, a.radSq, a.rad * 1.002f + g.slack, r0, a.knn,
                            a.grp_factor, a.cell_budget, a.dbg, s_hit, s_cand, s_spos, s_cstart, s_coff, kd2, kidx, kspos,
                            a.grp_min);
    // the neighbour list (sorted positions, ascending (d2, idx); NONE beyond its end) for the fit kernel
    if (!valid || lane >= Q) return;
    if (a.kth_io) a.kth_io[qi] = active ? kth_d2(kd2, a.knn) : INFINITY;
    uint32_t* o = a.out_knn + (size_t)orig * K;
#pragma unroll
    for (int j = 0; j < K; j++) o[j] = (active && kidx[j] != NONE_U32) ? kspos[j] : NONE_U32;
}

// ---- plane fit: one thread per local point (original index) -----------------------------------
template <int K>
__global__ __launch_bounds__(256) void pt2pl_fit_kernel(const PlArgs a, const float* __restrict__ lx,
                                                        const float* __restrict__ ly, const float* __restrict__ lz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
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
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[65];

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
static void launch_k(const PlArgs& a, uint32_t q, const mp2p_hip_cloud* cloud, hipStream_t st)
{
    const uint32_t n_tiles = (a.n_l + q - 1) / q;
    if (q == 8) hipLaunchKernelGGL((pt2pl_tile_kernel<K, 8>), dim3(n_tiles), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((pt2pl_tile_kernel<K, PL_Q>), dim3(n_tiles), dim3(64), 0, st, a);
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
    const uint32_t Q       = ctx->tune.pl_q ? ctx->tune.pl_q : (n_l <= 400000 ? 8u : (uint32_t)PL_Q);
    const uint32_t n_tiles = (uint32_t)((n_l + Q - 1) / Q);
    const uint32_t Kcap    = prm->knn <= 5 ? 5u : prm->knn <= 8 ? 8u : prm->knn <= 12 ? 12u : 16u;
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)n_tiles * 6));
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
    if (phase != 2)
    {
        ctx->pl_hint_map = map, ctx->pl_hint_cloud = cloud, ctx->pl_hint_n = n_l, ctx->pl_hint_knn = prm->knn, ctx->pl_hint_rad = prm->searchRadius;
        for (int i = 0; i < 12; i++) ctx->pl_hint_pose[i] = pose[i];
    }
    a.grp_min = 2.0f * cell0;
    a.dbg = nullptr;
    if (ctx->profiling == 2)
    {
        MP2P_TRY_HIP(ctx, ctx->counters.ensure(64));
        MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64 * sizeof(unsigned long long), ctx->stream));
        a.dbg = ctx->counters.p;
    }

    ctx->pending_lane = 0;
    if (phase != 2)
    {
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    if (Kcap == 5) launch_k<5>(a, Q, cloud, ctx->stream);
    else if (Kcap == 8) launch_k<8>(a, Q, cloud, ctx->stream);
    else if (Kcap == 12) launch_k<12>(a, Q, cloud, ctx->stream);
    else launch_k<16>(a, Q, cloud, ctx->stream);
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if (ctx->profiling) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    {
        const int rc = launch_bbox_reduce(ctx, n_tiles);
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
