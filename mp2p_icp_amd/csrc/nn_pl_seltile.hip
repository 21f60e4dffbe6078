// nn_pl_seltile.hip -- round 6: the k-nearest-neighbour tile search (Matcher_Point2Plane, Matcher_Adaptive,
// Matcher_Points_DistanceThreshold with pairingsPerPoint > 1) rebuilt the way round 5 rebuilt the point-to-point search
// (nn_seltile.hip).  Included by nn_pt2pl.hip; DESIGN.md section 4 (K5) has the measurements.
//
// What rounds 2-5 did (knn_search in nn_pt2pl.hip): a pass staged every occupied voxel of the BOUNDING BOX of the group's
// search cubes and every lane scanned every staged candidate exactly (fp32 dist2 + compare per (query, candidate) pair,
// ~190 instructions per 1 024 pairs), with tiles of 8 queries because nothing else filled the chip.  Here, per tile of 32
// Morton-consecutive queries (lane = query slot + 32 x candidate slice, as in nn_seltile_kernel):
//
//   * BALL RULE.  The occupied voxels of the pass's box are listed from the level-0 occupancy bricks and SELECTED on the
//     matrix pipe: voxel v is staged iff |c_v - q_m|^2 <= (r_m + margin + rho)^2 for some query m of the group (c_v = centre,
//     rho = half diagonal): one v_mfma_f32_32x32x2_f32 pair per 32 voxels x 32 queries.  The box rule staged the slab between
//     a wall and the tile; the balls only touch the wall near each query's foot point (tools/cand_model.py: 2-2.5 x fewer
//     points, and the heavy tail gone).
//   * PREFILTER.  The staged candidates are tested against the tile's queries on the matrix pipe in the K = 4 form of
//     nn_seltile.hip (S = d2 - |q'|^2 + beta, |S - exact| <= mtol, proven bound: tests/test_prefilter_bound.py); a lane only
//     NOTES the candidates with S <= its limit (k-th distance so far, the pass radius, the search radius: whichever is
//     smallest, + mtol) in an LDS queue -- the 16 values a lane holds of a block become a 16-bit mask with two instructions each
//     (subtract, v_alignbit of the sign), no branch per value: with 32 different queries in a wave SOMEBODY has a hit in nearly
//     every block, so the nested tests of the point-to-point kernel (whose hits are rare) ran in full every time: 11 600
//     vector + 8 800 scalar instructions per tile in the first build of this file -- and the exact FMA-free dist2 and the
//     k-list insertion chain run over the queues when one is nearly full, with most lanes busy.  Exactness: a candidate within the limit has S <= limit + mtol, so nothing that
//     belongs in a list is dropped; what gets through is recomputed exactly and ordered by (d2, original index).
//   * CERTIFICATE (CERT; PlArgs::lb_io).  The bound of "every map point NOT in the final list" that the next call's
//     pt2pl_cert_kernel relies on is the smallest of three things:
//       (1) candidates that were evaluated exactly and not kept: their exact distance (rej);
//       (2) candidates that were staged and dropped by the prefilter: the lane's limit is widened to
//           (min(sqrt(kth), r) + margin)^2, so such a candidate is farther than min(sqrt(kth_final), r) + margin -- kth only
//           shrinks during a pass, the final value gives the weakest bound;
//       (3) points that were not staged: their voxel's circumsphere misses the ball of radius r + margin around THIS query,
//           so they are farther than r + margin (the box rule's argument was "beyond a face of the box").
//     tests/test_certificate_bound.py models (1)-(3) on the CPU against brute force.
//   * W WAVES PER TILE (W = 1 or 4, a run-time property of the workgroup: see pt2pl_seltile_kernel).  A KITTI scan is 120 k queries = 3 750 tiles for 4 096+ wave slots: the kernel lasts as
//     long as its longest tile.  With W > 1 the workgroup's waves serve the SAME 32 queries: every wave lists and selects
//     (redundant, a quarter of a tile's instructions), the selected voxels are dealt round-robin in runs of 16, each wave
//     stages / filters / inserts its share in its own LDS, and the k-lists are merged through LDS at the end of the pass
//     (workgroup barriers only there: inside a pass a wave orders its own LDS traffic with wave-scope fences).  Round 5
//     split tiles ACROSS workgroups and lost to the agent-scope fences of the hand-off; inside a workgroup there are none.
#include "device_utils.hpp"

namespace mp2p
{
constexpr int PS_CAP   = 256;  // staged candidates per round
constexpr int PS_VLIST = 256;  // occupied voxels listed per round (LDS is what bounds the kernel's occupancy: 12.5 KB per wave = 3 waves per SIMD)
constexpr int PS_HITQ  = 8;    // queued hits per lane (flushed before a block's hits would not fit)
constexpr int PS_HL    = 384;  // hits of one staging round (else of a quarter of a block: <= 256) listed for the wave-wide exact test
constexpr int PS_MARKS = 512;  // level-2 cells (4x4x4 bricks each) of a wide pass's box: one u64 of brick marks each (the staging area's first 4 KB)

// LDS of ONE wave of a tile (12.9 KB: 3 waves per SIMD).  At the end of a pass of a W > 1 tile the first 4 KB carry the wave's k-lists to
// the other waves (32 queries x 16 entries x {d2, sorted position}).
struct __attribute__((aligned(16))) PsLds
{
    float          x[PS_CAP], y[PS_CAP], z[PS_CAP];
    uint32_t       spos[PS_CAP], owner[PS_CAP];
    uint32_t       cstart[64], coff[64];
    uint32_t       vox[PS_VLIST];
    uint2          hitq[PS_HITQ * 64];  // queued hits {exact d2, sorted position} of lane l at [e * 64 + l]: self-contained, a queue outlives the staging round that filled it
    float          qsx[32], qsy[32], qsz[32], tB[32];  // the tile's queries and their search limit (d2 <= / < tB)
    float          tA[64];              // per lane: min(pass radius^2, k-th d2 of the lane's list as of its last flush)
    uint32_t       qcnt[64];            // per lane: entries in its queue
    uint32_t       qrej[64];            // per lane (CERT): smallest exact d2 tested and not queued, as bits (>= +0: they order like the value)
    unsigned short hl[PS_HL];           // the hits of a block: (lane << 4 | row) of every prefilter value within its lane's limit
};

// a wave's own LDS writes made visible to its own later reads (other lanes): LDS executes a wave's instructions in order,
// so only the compiler has to be told (a workgroup barrier here would have to be reached by the tile's other waves as well,
// whose trip counts differ)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int K>
__device__ __forceinline__ float ps_kth(const float (&kd2)[K], uint32_t knn)
{
    float v = INFINITY;
#pragma unroll
    for (int q = 0; q < K; q++)
        if (q == (int)knn - 1) v = kd2[q];
    return v;
}

// Sorted insertion of (cd, cs) into the K-list, ascending (d2, original index); what falls off the end is left in (cd, cs).
// The list holds (d2, sorted position) only: the original index -- the tie-break -- is the w component of the map point and is
// fetched only when a distance EQUALS one in the list (exact lattices and duplicates): one test for the whole chain, a wave-uniform
// branch that real scans never take; the common chain is a compare and four moves per entry (a per-entry tie test made it 36
// instructions per entry: 4 500 of a tile's 15 000).
template <int K>
__device__ __forceinline__ void ps_insert(const float4* __restrict__ pts, float (&kd2)[K], uint32_t (&kspos)[K], bool on, float& cd, uint32_t& cs)
{
    bool tie = false;
#pragma unroll
    for (int q = 0; q < K; q++) tie = tie || (cd == kd2[q]);
    tie = tie && on && cs != NONE_U32 && cd < INFINITY;
    // (once the candidate has taken a place, everything behind it moves one place on: the displaced entry precedes its old
    //  successor by construction, also where their distances are equal -- `sw`)
    bool sw = false;
    if (__ballot(tie) != 0ull)
    {
        const uint32_t ci = tie ? __float_as_uint(pts[cs].w) : 0u;
#pragma unroll
        for (int q = 0; q < K; q++)
        {
            bool less = on && (sw || cd < kd2[q]);
            if (tie && !sw && cd == kd2[q] && kspos[q] != NONE_U32) less = ci < __float_as_uint(pts[kspos[q]].w);
            const float    td = kd2[q];
            const uint32_t ts = kspos[q];
            kd2[q] = less ? cd : td, kspos[q] = less ? cs : ts;
            cd = less ? td : cd, cs = less ? ts : cs;
            sw = sw || less;
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < K; q++)
    {
        const bool     less = on && (sw || cd < kd2[q]);
        const float    td   = kd2[q];
        const uint32_t ts   = kspos[q];
        kd2[q] = less ? cd : td, kspos[q] = less ? cs : ts;
        cd = less ? td : cd, cs = less ? ts : cs;
        sw = sw || less;
    }
}

// Exact k nearest neighbours (fp32 metric, (d2, idx) order) of the tile's 32 queries restricted to d2 <= lim2 (STRICT: <);
// rmax = a radius that covers every such point; r0 = the first pass's radius (warm start).  Must be called by all W waves of
// the tile with identical arguments (lane l and lane l + 32 of every wave hold query l); on return every wave holds the same
// lists.  CERT: *lb_out = a lower bound of the distance to every map point outside the final list (see the file header).
// The map must have a level-0 occupancy bitmap and rmax must not exceed PS_MAX_RADIUS_CELLS level-0 voxels (the host checks).
constexpr float PS_MAX_RADIUS_CELLS = 48.0f;

template <int K, bool STRICT, bool CERT, bool INSTR>
__device__ __forceinline__ void knn_sel_search(const GridView& g, const int W, const int lane, const int wv, PsLds* __restrict__ Lw, PsLds* __restrict__ L0,
                                               const float qx, const float qy, const float qz, const bool active, const float lim2,
                                               const float rmax, const float r0, const uint32_t knn, const float grp_factor,
                                               const float grp_min, const float grp_all_bricks, const float cm, float (&kd2)[K],
                                               uint32_t (&kspos)[K], float* lb_out, uint32_t* cand_out,
                                               unsigned long long* dbg, unsigned long long* tl_info, unsigned char* touched, const int sol = 0)
{
    // sol (timing only, INSTR builds: results are NOT valid): 1 = no exact tests / queues / chains, 2 = + no prefilter blocks, 3 = + no staging,
    // 4 = + no listing / selection

    const bool hi = lane >= 32;
    float      r    = fminf(r0, rmax);
    bool       done = !active;
    float      rej  = INFINITY;  // CERT: smallest exact d2 of a candidate tested and not kept (this pass); once done: the bound
#pragma unroll
    for (int j = 0; j < K; j++) kd2[j] = INFINITY, kspos[j] = NONE_U32;
    uint32_t        st_pass = 0, st_cand = 0, st_listed = 0;  // st_cand: what THIS wave staged (all passes)
    uint32_t        cand_total = 0;                           // W > 1: the tile's sum, redone at every merge
    uint32_t        st_fiter = 0, st_hits = 0, st_flush = 0, st_eiter = 0, st_pos = 0, st_blocks = 0;  // profiling level 2: chain iterations, queued hits, flushes (this wave)
    unsigned long long tk_stage = 0, tk_pref = 0, tk_flush = 0, tk_pass = 0, tk_merge = 0;  // ... and 100 MHz ticks spent staging / in the prefilter blocks / in flushes
    const long long dbg_t0 = (INSTR && dbg) ? (long long)wall_clock64() : 0;

    // bounding box of the tile's queries, once: a pass's box is this box grown by its widest radius
    const float tqx0 = wave_min_nn(!done ? qx : INFINITY), tqy0 = wave_min_nn(!done ? qy : INFINITY), tqz0 = wave_min_nn(!done ? qz : INFINITY);
    const float tqx1 = wave_max_nn(!done ? qx : -INFINITY), tqy1 = wave_max_nn(!done ? qy : -INFINITY), tqz1 = wave_max_nn(!done ? qz : -INFINITY);
    const uint32_t obx = g.occ_bx[0], oby = g.occ_by[0], obz = g.occ_bz[0];
    const unsigned long long* occ0 = g.occ + g.occ_off[0];
    if (!hi) Lw->qsx[lane] = qx, Lw->qsy[lane] = qy, Lw->qsz[lane] = qz, Lw->tB[lane] = lim2;
    wave_lds_sync();
    const float hs  = g.hf * (float)(1u << g.shift0);  // level-0 voxel edge
    const float rho = hs * 0.8660255f;                 // half diagonal (rounded up)
    const float cmx = CERT ? cm : 0.f;

    while (true)
    {
        const unsigned long long pend = __ballot(!done);
        if (pend == 0ull) break;
        // ---- the GROUP of this pass.  Every voxel is tested against every query's OWN ball, so what a pass stages does not depend
        //      on how far apart its queries are; only the LISTING does (the occupied bricks of the group's box).  A compact box
        //      (<= 64 bricks) is listed brick by brick; a wide one -- a tile of far-field scan points is 32 places metres apart: the
        //      seed-neighbourhood rule of the box-rule kernel made 12-14 passes of such tiles, 30 us each -- is entered through the
        //      level-2 occupancy words AND a bitmap of the bricks some query's ball box touches (marks, below): all pending
        //      queries stay ONE pass while that bitmap fits (512 level-2 cells: 32 bricks per axis), else the seed's cluster, else
        //      the seed alone.
        bool  grp    = !done;
        float rmax_t = 0.f;
        float qlx = tqx0, qly = tqy0, qlz = tqz0, qhx = tqx1, qhy = tqy1, qhz = tqz1;
        uint32_t cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0, nbx = 0, nby = 0, nb = 0, nL = 0;
        float    lox, loy, loz, hix, hiy, hiz;
        const bool have_l2 = g.n_levels > 2u && g.occ_off[2] != OCC_NONE;
        for (int attempt = 0;; attempt++)
        {
            if (attempt >= 1)
            {
                const int   seed = __ffsll((long long)pend) - 1;
                const float sx = readlane_f(qx, seed), sy = readlane_f(qy, seed), sz = readlane_f(qz, seed);
                const float sr = readlane_f(r, seed);
                // (the cluster: a few bricks, but no wider than the tolerance rule below admits for the seed's radius: half extent
                //  h with 3 h^2 2^-15 <= 0.5 (r + cm)^2, i.e. h <= 74 (r + cm); a third of that per axis side leaves room for the radii)
                const float G  = fminf(fmaxf(fmaxf(grp_factor * sr, grp_min), grp_all_bricks * 4.f * hs), fmaxf(24.f * (sr + cmx), grp_min));
                grp = !done && fabsf(qx - sx) <= G && fabsf(qy - sy) <= G && fabsf(qz - sz) <= G;
                if (attempt >= 2) grp = !done && (lane & 31) == (seed & 31);
                qlx = wave_min_nn(grp ? qx : INFINITY), qly = wave_min_nn(grp ? qy : INFINITY), qlz = wave_min_nn(grp ? qz : INFINITY);
                qhx = wave_max_nn(grp ? qx : -INFINITY), qhy = wave_max_nn(grp ? qy : -INFINITY), qhz = wave_max_nn(grp ? qz : -INFINITY);
            }
            rmax_t           = wave_max_pos(grp ? r : 0.f);
            const float rext = rmax_t + cmx;
            lox = qlx - rext, loy = qly - rext, loz = qlz - rext, hix = qhx + rext, hiy = qhy + rext, hiz = qhz + rext;
            cx0 = cy0 = cz0 = cx1 = cy1 = cz1 = nbx = nby = nb = nL = 0;
            const float clx = fmaxf(lox, g.bbmin[0]), cly = fmaxf(loy, g.bbmin[1]), clz = fmaxf(loz, g.bbmin[2]);
            const float chx = fminf(hix, g.bbmax[0]), chy = fminf(hiy, g.bbmax[1]), chz = fminf(hiz, g.bbmax[2]);
            if (!((clx > chx) || (cly > chy) || (clz > chz)))
            {
                cx0 = cell_fine(clx, g.ox, g.inv_hf) >> g.shift0, cx1 = cell_fine(chx, g.ox, g.inv_hf) >> g.shift0;
                cy0 = cell_fine(cly, g.oy, g.inv_hf) >> g.shift0, cy1 = cell_fine(chy, g.oy, g.inv_hf) >> g.shift0;
                cz0 = cell_fine(clz, g.oz, g.inv_hf) >> g.shift0, cz1 = cell_fine(chz, g.oz, g.inv_hf) >> g.shift0;
                nbx = (cx1 >> 2) - (cx0 >> 2) + 1u, nby = (cy1 >> 2) - (cy0 >> 2) + 1u;
                const uint32_t           nbz = (cz1 >> 2) - (cz0 >> 2) + 1u;
                const unsigned long long nbl = (unsigned long long)nbx * nby * nbz;
                nb = (uint32_t)min(nbl, 0xFFFFFFFFull);
                const unsigned long long nLl = (unsigned long long)((cx1 >> 4) - (cx0 >> 4) + 1u) * ((cy1 >> 4) - (cy0 >> 4) + 1u) * ((cz1 >> 4) - (cz0 >> 4) + 1u);
                nL = (uint32_t)min(nLl, 0xFFFFFFFFull);
            }
            // the voxel list addresses 10 bits per axis relative to the box's corner; a wide box needs the marks (have_l2, 512 cells)
            // or, on a map without level-2 words, at most 32 768 bricks enumerated one by one
            // ... and the prefilter's tolerance must stay small against the limits it is added to: it grows with the box (2^-15 of the
            // squared half diagonal), and a dense tile whose queries sat 20 m apart let 17 000 candidates through to the exact test
            // where 900 belonged (the slowest tile of the C3 scan: 2.6 x the time of its neighbours).  Half of the smallest limit.
            const float rmin_t = wave_min_pos(grp ? r : INFINITY);
            const float hxb = 0.5f * (hix - lox) + hs, hyb = 0.5f * (hiy - loy) + hs, hzb = 0.5f * (hiz - loz) + hs;
            const bool  tol_ok = attempt >= 2 || (hxb * hxb + hyb * hyb + hzb * hzb) * (1.0f / 32768.0f) <= 0.5f * (rmin_t + cmx) * (rmin_t + cmx);
            const bool fits = !(cx1 - cx0 >= 1024u || cy1 - cy0 >= 1024u || cz1 - cz0 >= 1024u) &&
                              (nb <= 64u || (have_l2 ? nL <= (uint32_t)PS_MARKS : nb <= 32768u)) && tol_ok;
            if (fits) break;
            if (attempt >= 2) __builtin_trap();  // (one query's box: the host admits radii up to PS_MAX_RADIUS_CELLS voxels only)
        }
        st_pass++;
        const unsigned long long tp0 = (INSTR && dbg) ? wall_clock64() : 0ull;

        const float prune  = rmax_t + cmx + 4.f * g.slack;
        const float prune2 = prune * prune;
        const float ocx = 0.5f * (lox + hix), ocy = 0.5f * (loy + hiy), ocz = 0.5f * (loz + hiz);
        const float hx = 0.5f * (hix - lox) + hs, hy = 0.5f * (hiy - loy) + hs, hz = 0.5f * (hiz - loz) + hs;
        const float beta = hx * hx + hy * hy + hz * hz;
        const float mtol = beta * (1.0f / 32768.0f);  // proven error bound of S (nn_query.hip, tests/test_prefilter_bound.py)
        const float cqx = qx - ocx, cqy = qy - ocy, cqz = qz - ocz;
        const float qn2 = cqx * cqx + cqy * cqy + cqz * cqz;
        // K = 4 operands (nn_seltile.hip): T = [c'x c'y | c'z  |c'|^2 + beta] . [-2q'x -2q'y | -2q'z  1] = d2 - |q'|^2 + beta
        const float b0 = -2.0f * (hi ? cqy : cqx);
        const float b1 = hi ? 1.0f : -2.0f * cqz;
        const float o0 = hi ? ocy : ocx, o1 = hi ? -beta : ocz;
        const float loff = beta - qn2;
        // selection: [-2q'x -2q'y | -2q'z  |q'|^2 - R^2 + 4 beta] . [c'x c'y | c'z 1], per column against tol - |c'|^2 + 4 beta
        const float Rq   = r + cmx + rho + 4.f * g.slack;
        const float sel1 = hi ? (grp ? qn2 - Rq * Rq + 4.f * beta : 1e30f) : -2.0f * cqz;
        const float stol = 4.0f * mtol + 1e-12f;

        // a repeated pass rescans voxels already seen: the lists of its queries restart
#pragma unroll
        for (int j = 0; j < K; j++)
            if (grp) kd2[j] = INFINITY, kspos[j] = NONE_U32;
        if (CERT && grp) rej = INFINITY;
        float       kth = ps_kth(kd2, knn);
        const float r2  = r * r;
        // the lane's limit on the prefilter's scale.  CERT: widened by the margin, so that what the prefilter drops is
        // farther than min(sqrt(kth), r) + cm; else exactly what an insertion needs
        auto lane_lim = [&]() __attribute__((always_inline)) -> float {
            float t;
            if (CERT)
            {
                const float a = fminf(sqrtf(kth), r) + cm;
                t             = a * a;
            }
            else
                t = fminf(fminf(kth, r2), lim2);
            return t * 1.000001f + mtol + loff;
        };
        float    lim = lane_lim();
        uint32_t hq  = 0;  // hits queued for this lane (= Lw->qcnt[lane])
        Lw->qcnt[lane] = 0u, Lw->qrej[lane] = __float_as_uint(INFINITY);
        Lw->tA[lane] = grp ? r2 : -1.0f;  // (a lane outside the group takes nothing)
        wave_lds_sync();

        // the queued hits through the insertion chain; wave-wide (uniform trip count).  Runs when a queue could not take a
        // block's hits and at the end of the pass -- NOT per staging round: a query's neighbours sit in one or two voxels, so
        // its hits come in a burst while the other lanes have none.  An entry is {exact d2, sorted position}: it outlives the
        // staging round (a version that kept the position only and fetched the point again here: 1.7 us per iteration, the
        // latency of a dependent load).
        auto flush = [&]() __attribute__((always_inline)) {
            const unsigned long long tf0 = (INSTR && dbg) ? wall_clock64() : 0ull;
            const uint32_t hmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(wave_max_pos(__uint_as_float(hq))));  // (small integers: their bit patterns order like floats)
            if (INSTR && dbg) st_fiter += hmax, st_hits += wave_sum_u32(hq), st_flush++;
            for (uint32_t e = 0; e < hmax; e++)
            {
                const bool  mine = e < hq;
                const uint2 h    = Lw->hitq[e * 64u + (uint32_t)lane];
                float       cd   = mine ? __uint_as_float(h.x) : INFINITY;
                uint32_t    cs   = mine ? h.y : NONE_U32;
                ps_insert<K>(g.pts, kd2, kspos, mine, cd, cs);
                if (CERT) rej = fminf(rej, cd);  // what fell off the end: the candidate itself or the old last entry (inf: nothing)
            }
#pragma unroll
            for (int q = 0; q < K; q++)  // only the knn nearest are kept
                if (q >= (int)knn)
                {
                    if (CERT) rej = fminf(rej, kd2[q]);
                    kd2[q] = INFINITY, kspos[q] = NONE_U32;
                }
            kth = ps_kth(kd2, knn);
            hq  = 0;
            lim = lane_lim();
            Lw->qcnt[lane] = 0u;
            Lw->tA[lane]   = fminf(r2, kth);
            wave_lds_sync();
            if (INSTR && dbg) tk_flush += wall_clock64() - tf0;
        };
        // The prefilter's hits, tested exactly and queued WAVE-WIDE.  With 32 different queries a block of 32 candidates holds ~8
        // hits spread over as many lanes; a lane working through its own (the first two builds of this file) ran ~30 instructions per
        // hit with one or two lanes active: 11 000 vector instructions per tile, 300 000 in a dense tile whose wide box made the
        // prefilter's tolerance generous.  Here the hits of a whole staging round (8 blocks) are listed in LDS -- a code (lane, block,
        // row) per hit at the lane's prefix offset -- and lane e takes the e-th: candidate from the staging arrays, query and limits
        // from the per-query / per-lane tables, the exact FMA-free dist2 of the reference, the tests an insertion needs, a slot in the
        // OWNER's queue by an LDS atomic.
        // emit: this lane's hits in m (bit 15 - rr of the low half = row rr of block b0, the high half: block b0 + 1) at hl[off ...]
        auto emit = [&](uint32_t mm, const uint32_t b0, uint32_t& off) __attribute__((always_inline)) {
            while (mm)
            {
                const int b = 31 - __clz((int)mm);
                mm &= ~(1u << b);
                Lw->hl[off++] = (unsigned short)(((uint32_t)lane << 7) | ((b0 + (uint32_t)(b >> 4)) << 4) | (uint32_t)(15 - (b & 15)));
            }
        };
        // process_list: the `total` (<= PS_HL) listed hits.  A hit whose owner's queue is full stays listed (compacted to the front
        // of the list); the queues are then flushed and the rest is tried again -- against the owners' tighter limits.  (A fixed
        // share of the queue per round does not work: a query's neighbours sit in one or two voxels, i.e. ALL its hits of a pass
        // arrive within one round.)
        auto process_list = [&](uint32_t n) __attribute__((always_inline)) {
            while (n != 0u)
            {
                wave_lds_sync();
                if (INSTR && dbg) st_pos += n, st_eiter += (n + 63u) / 64u;
                uint32_t nfail = 0u;
                for (uint32_t base = 0; base < n; base += 64u)
                {
                    const uint32_t e    = base + (uint32_t)lane;
                    const bool     ok   = e < n;
                    const uint32_t code = Lw->hl[ok ? e : 0u];
                    const uint32_t ql = code >> 7, rr = code & 15u, qq = ql & 31u;
                    const uint32_t j  = ((code >> 4) & 7u) * 32u + (rr & 3u) + 8u * (rr >> 2) + ((ql & 32u) ? 4u : 0u);
                    const float    cd = dist2(Lw->qsx[qq], Lw->qsy[qq], Lw->qsz[qq], Lw->x[j], Lw->y[j], Lw->z[j]);
                    const float    t2 = Lw->tB[qq];
                    // (tA: the owner's k-th distance as of its last flush -- what does not beat it now never will -- and the pass radius)
                    const bool in = ok && (STRICT ? (cd < t2) : (cd <= t2)) && cd <= Lw->tA[ql];
                    bool       fail = false;
                    if (in)
                    {
                        const uint32_t slot = atomicAdd(&Lw->qcnt[ql], 1u);
                        if (slot < (uint32_t)PS_HITQ) Lw->hitq[slot * 64u + ql] = make_uint2(__float_as_uint(cd), Lw->spos[j]);
                        else fail = true;
                    }
                    else if (CERT && ok) atomicMin(&Lw->qrej[ql], __float_as_uint(cd));
                    const unsigned long long fb = __ballot(fail);
                    if (fb != 0ull)
                    {   // (written at or below this chunk's first entry, which every lane has read)
                        if (fail) Lw->hl[nfail + (uint32_t)__popcll(fb & ((1ull << lane) - 1ull))] = (unsigned short)code;
                        nfail += (uint32_t)__popcll(fb);
                    }
                }
                wave_lds_sync();
                hq = min(Lw->qcnt[lane], (uint32_t)PS_HITQ);
                if (nfail != 0u) flush();
                n = nfail;
            }
        };

        // one batch of <= 64 resolved voxels (lane = voxel: start, cnt): staged in rounds of PS_CAP points, filtered, queued
        auto batch = [&](uint32_t start, uint32_t cnt) __attribute__((always_inline)) {
            const uint32_t incl  = wave_incl_scan(cnt, lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0) return;
            const uint32_t off = incl - cnt;
            Lw->cstart[lane] = start;
            Lw->coff[lane]   = off;
            st_cand += total;
            if (INSTR && sol >= 3) return;
            for (uint32_t base = 0; base < total; base += PS_CAP)
            {
                const uint32_t m     = min((uint32_t)PS_CAP, total - base);
                const uint32_t m_pad = (m + 31u) & ~31u;
                const unsigned long long ts0 = (INSTR && dbg) ? wall_clock64() : 0ull;
                // ---- stage: lane l fills slots 4l..4l+3; a segmented broadcast tells which voxel a slot belongs to ----------
                *reinterpret_cast<uint4*>(&Lw->owner[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
                wave_lds_sync();
                if (cnt > 0)
                {
                    if (off >= base && off < base + PS_CAP) Lw->owner[off - base] = (uint32_t)lane + 1u;
                    else if (off < base && off + cnt > base) Lw->owner[0] = (uint32_t)lane + 1u;
                }
                wave_lds_sync();
                {
                    const uint4    o4 = *reinterpret_cast<const uint4*>(&Lw->owner[4 * lane]);
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
                            src[k]           = Lw->cstart[v] + (base + t0 + k - Lw->coff[v]);
                        }
                    }
                    // (all four loads issued before any is waited for: written as `if (ok) c4[k] = pts[src[k]]` the compiler put every
                    //  load in an exec-masked block of its own with its own s_waitcnt vmcnt(0) -- four serial round trips per round,
                    //  8-20 us of a 256-candidate round; an out-of-range slot reads point 0 and is overwritten by the padding)
#pragma unroll
                    for (int k = 0; k < 4; k++) c4[k] = g.pts[(t0 + k < m) ? src[k] : 0u];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (!(t0 + k < m)) c4[k] = make_float4(1e18f, 0.f, 0.f, __uint_as_float(NONE_U32));  // padding: far but FINITE (inf - inf = NaN wins an integer minimum)
                    wave_lds_sync();  // (the owner slots are read: they now take |c'|^2)
                    *reinterpret_cast<float4*>(&Lw->x[t0]) = make_float4(c4[0].x, c4[1].x, c4[2].x, c4[3].x);
                    *reinterpret_cast<float4*>(&Lw->y[t0]) = make_float4(c4[0].y, c4[1].y, c4[2].y, c4[3].y);
                    *reinterpret_cast<float4*>(&Lw->z[t0]) = make_float4(c4[0].z, c4[1].z, c4[2].z, c4[3].z);
                    *reinterpret_cast<uint4*>(&Lw->spos[t0]) = make_uint4(src[0], src[1], src[2], src[3]);
                    {
                        float n4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            const float ex_ = c4[k].x - ocx, ey_ = c4[k].y - ocy, ez_ = c4[k].z - ocz;
                            n4[k] = ex_ * ex_ + ey_ * ey_ + ez_ * ez_;
                        }
                        *reinterpret_cast<float4*>(&Lw->owner[t0]) = make_float4(n4[0], n4[1], n4[2], n4[3]);
                    }
                    if (INSTR && touched)
                    {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (t0 + k < m) touched[src[k]] = 1;
                    }
                }
                wave_lds_sync();
                const unsigned long long ts1 = (INSTR && dbg) ? wall_clock64() : 0ull;
                const unsigned long long tf_before = tk_flush;
                // ---- prefilter on the matrix pipe: 32 candidates x 32 queries per two instructions.  The lane's 16 values of a block
                //      become a 16-bit mask (value rr at bit 15 - rr: the sign of (S - lim) shifted in by one v_alignbit each; S == lim
                //      counts as outside: the limit carries a relative margin of 1e-6 for exactly that); the round's 8 masks are
                //      processed together
                {
                    const float* s_n  = reinterpret_cast<const float*>(Lw->owner);
                    const float* s_a0 = hi ? Lw->y : Lw->x;
                    const float* s_a1 = hi ? s_n : Lw->z;
                    uint32_t     mw[PS_CAP / 64];  // two blocks per word
#pragma unroll
                    for (int b = 0; b < PS_CAP / 32; b++)
                    {
                        uint32_t m16 = 0u;
                        if ((uint32_t)(32 * b) < m_pad && !(INSTR && sol >= 2))  // (wave-uniform)
                        {
                            const uint32_t c   = (uint32_t)(32 * b) + ((uint32_t)lane & 31u);
                            const float    a0v = s_a0[c] - o0, a1v = s_a1[c] - o1;
                            f32x16         acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc, 0, 0, 0);
#pragma unroll
                            for (int rr = 0; rr < 16; rr++) m16 = __builtin_amdgcn_alignbit(m16, __float_as_uint(acc[rr] - lim), 31);
                            m16 &= 0xFFFFu;
                            if (INSTR && dbg) st_blocks++;
                        }
                        if (b & 1) mw[b >> 1] |= m16 << 16;
                        else mw[b >> 1] = m16;
                    }
                    uint32_t cnt = 0u;
#pragma unroll
                    for (int w = 0; w < PS_CAP / 64; w++)
                    {
                        if (!grp) mw[w] = 0u;
                        cnt += (uint32_t)__popc(mw[w]);
                    }
                    if (INSTR && sol >= 1) cnt = (cnt == 0xFFFFFFFFu) ? 1u : 0u;
                    if (__ballot(cnt != 0u) != 0ull)
                    {
                        // The round's hits as ONE item when the list takes them (PS_HL); else 32 items, a quarter of a block each
                        // (4 rows: at most 256 in the wave).  One loop for both: flush, emit and process_list are inlined once -- unrolled over blocks and quarters the
                        // kernel was 9 000 instructions long and ran out of the instruction cache.
                        const uint32_t incl0  = wave_incl_scan(cnt, lane);
                        const uint32_t total0 = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
                        const bool     whole  = total0 <= (uint32_t)PS_HL;
                        const uint32_t n_items = whole ? 1u : (uint32_t)(PS_CAP / 32) * 4u;
#pragma unroll 1
                        for (uint32_t it = 0; it < n_items; it++)
                        {
                            uint32_t m0 = mw[0], m1 = mw[1], m2 = mw[2], m3 = mw[3];
                            uint32_t ci = cnt, ii = incl0, ti = total0;
                            if (!whole)
                            {
                                // item it = block it / 4, rows 4 (it % 4) .. + 3: the bits 15 - rr of that block's half word
                                const uint32_t b = it >> 2, sh = 16u * (b & 1u), qm = (0xF000u >> (4u * (it & 3u))) << sh;
                                const uint32_t wsel = b >> 1;
                                m0 = wsel == 0u ? (mw[0] & qm) : 0u, m1 = wsel == 1u ? (mw[1] & qm) : 0u;
                                m2 = wsel == 2u ? (mw[2] & qm) : 0u, m3 = wsel == 3u ? (mw[3] & qm) : 0u;
                                ci = (uint32_t)__popc(m0 | m1 | m2 | m3);
                                if (__ballot(ci != 0u) == 0ull) continue;
                                ii = wave_incl_scan(ci, lane), ti = (uint32_t)__builtin_amdgcn_readlane((int)ii, 63);
                            }
                            uint32_t off = ii - ci;
                            emit(m0, 0u, off), emit(m1, 2u, off), emit(m2, 4u, off), emit(m3, 6u, off);
                            process_list(ti);
                        }
                    }
                }
                if (INSTR && dbg) tk_stage += ts1 - ts0, tk_pref += (wall_clock64() - ts1) - (tk_flush - tf_before);
                wave_lds_sync();
            }
        };

        // the masked occupancy word of one level-0 brick and its first voxel relative to (cx0, cy0, cz0), 10 bits per axis
        auto brick_word = [&](bool ok, uint32_t Bx, uint32_t By, uint32_t Bz, unsigned long long& bm, int& pb) __attribute__((always_inline)) {
            bm = 0ull;
            pb = ((int)(Bx * 4u) - (int)cx0) + ((int)(By * 4u) - (int)cy0) * 1024 + ((int)(Bz * 4u) - (int)cz0) * 1048576;
            if (!ok) return;
            const float h4 = 4.f * hs;
            const float x0 = g.ox + (float)(Bx * 4u) * hs, y0 = g.oy + (float)(By * 4u) * hs, z0 = g.oz + (float)(Bz * 4u) * hs;
            const float dx = fmaxf(0.f, fmaxf(x0 - qhx, qlx - (x0 + h4)));
            const float dy = fmaxf(0.f, fmaxf(y0 - qhy, qly - (y0 + h4)));
            const float dz = fmaxf(0.f, fmaxf(z0 - qhz, qlz - (z0 + h4)));
            if (dx * dx + dy * dy + dz * dz <= prune2 && Bx < obx && By < oby && Bz < obz)
            {
                const unsigned long long word = occ0[((size_t)Bz * oby + By) * obx + Bx];
                bm = word & spread_x(axis_mask(Bx, cx0, cx1)) & spread_y(axis_mask(By, cy0, cy1)) & spread_z(axis_mask(Bz, cz0, cz1));
            }
        };
        // the set bits of the lanes' words -> list entries base + (x, y, z of the bit) in out[0 .. cap), ranks r0 .. r0 + cap - 1
        auto list_bits = [&](unsigned long long bm, int pb, uint32_t bcnt, uint32_t bincl, uint32_t r0_, uint32_t cap, uint32_t* out) __attribute__((always_inline)) {
            unsigned long long nz = __ballot(bcnt != 0u);
            if (__popcll(nz) <= 24)
            {
                const int      lj  = (lane & 3) + ((lane >> 2) & 3) * 1024 + (lane >> 4) * 1048576;
                const uint32_t blo = (uint32_t)bm, bhi = (uint32_t)(bm >> 32), bbase = bincl - bcnt;
                nz &= __ballot(bbase < r0_ + cap && bincl > r0_);
                while (nz)
                {
                    const int b = __ffsll((long long)nz) - 1;
                    nz &= nz - 1ull;
                    const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)blo, b), whi = (uint32_t)__builtin_amdgcn_readlane((int)bhi, b);
                    const uint32_t rank = (uint32_t)__builtin_amdgcn_readlane((int)bbase, b) + __builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u));
                    const bool     has  = (((hi ? whi : wlo) >> (lane & 31)) & 1u) != 0u;
                    if (has && rank >= r0_ && rank < r0_ + cap) out[rank - r0_] = (uint32_t)(__builtin_amdgcn_readlane(pb, b) + lj);
                }
            }
            else
            {
                uint32_t           rank = bincl - bcnt;
                unsigned long long mm   = (rank < r0_ + cap && bincl > r0_) ? bm : 0ull;
                while (mm)
                {
                    const int bit = __ffsll((long long)mm) - 1;
                    mm &= mm - 1ull;
                    if (rank >= r0_ && rank < r0_ + cap) out[rank - r0_] = (uint32_t)(pb + (bit & 3) + ((bit >> 2) & 3) * 1024 + (bit >> 4) * 1048576);
                    rank++;
                }
            }
        };
        uint32_t nsel_all = 0;  // voxels selected so far in this pass (all waves count alike: the deal of W > 1)
        // one round of <= 64 bricks (lane = brick: bm, pb): list, select, resolve, stage + filter
        auto serve = [&](unsigned long long bm, int pb, uint32_t vcap) __attribute__((always_inline)) {
            const uint32_t bcnt   = (uint32_t)__popcll(bm);
            const uint32_t bincl  = wave_incl_scan(bcnt, lane);
            const uint32_t vtotal = (uint32_t)__builtin_amdgcn_readlane((int)bincl, 63);
            for (uint32_t rr0 = 0; rr0 < vtotal; rr0 += vcap)
            {
                list_bits(bm, pb, bcnt, bincl, rr0, vcap, Lw->vox);
                wave_lds_sync();
                const uint32_t nv = min(vcap, vtotal - rr0);
                if (INSTR) st_listed += nv;
                uint32_t nsel = 0;  // this wave's share, compacted in place
                for (uint32_t vb = 0; vb < nv; vb += 32u)
                {
                    const uint32_t vi = vb + ((uint32_t)lane & 31u);
                    uint32_t       pk = 0u;
                    float          ccx = 1e15f, ccy = 0.f, ccz = 0.f;  // padding: far, finite
                    if (vi < nv)
                    {
                        pk  = Lw->vox[vi];
                        ccx = (g.ox + ((float)(cx0 + (pk & 1023u)) + 0.5f) * hs) - ocx;
                        ccy = (g.oy + ((float)(cy0 + ((pk >> 10) & 1023u)) + 0.5f) * hs) - ocy;
                        ccz = (g.oz + ((float)(cz0 + (pk >> 20)) + 0.5f) * hs) - ocz;
                    }
                    const float v0   = hi ? ccy : ccx;
                    const float v1   = hi ? 1.0f : ccz;
                    const float vlim = stol - (ccx * ccx + ccy * ccy + ccz * ccz) + 4.f * beta;  // this column's limit
                    f32x16      acc  = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, v0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sel1, v1, acc, 0, 0, 0);
                    int mni = min(min(__float_as_int(acc[0]), __float_as_int(acc[1])), min(__float_as_int(acc[2]), __float_as_int(acc[3])));
#pragma unroll
                    for (int k = 1; k < 4; k++)
                        mni = min(mni, min(min(__float_as_int(acc[4 * k]), __float_as_int(acc[4 * k + 1])),
                                           min(__float_as_int(acc[4 * k + 2]), __float_as_int(acc[4 * k + 3]))));
                    const unsigned long long nb64 = __ballot(vi < nv && __int_as_float(mni) <= vlim);
                    const uint32_t           m32  = (uint32_t)nb64 | (uint32_t)(nb64 >> 32);  // both halves hold the same voxel
                    const uint32_t           below = (1u << (lane & 31)) - 1u;
                    if (W == 1)
                    {
                        if (!hi && ((m32 >> lane) & 1u)) Lw->vox[nsel + (uint32_t)__popc(m32 & below)] = pk;  // (writes at or below vb + lane)
                        nsel += (uint32_t)__popc(m32);
                    }
                    else
                    {
                        // this wave's share: runs of 16 selected voxels dealt round-robin (every wave computes the same m32)
                        const uint32_t gi   = nsel_all + (uint32_t)__popc(m32 & below);
                        const bool     keep = !hi && ((m32 >> lane) & 1u) && (((gi >> 4) & (uint32_t)(W - 1)) == (uint32_t)wv);
                        const uint32_t k32  = (uint32_t)__ballot(keep);
                        if (keep) Lw->vox[nsel + (uint32_t)__popc(k32 & below)] = pk;
                        nsel += (uint32_t)__popc(k32);
                    }
                    nsel_all += (uint32_t)__popc(m32);
                }
                wave_lds_sync();
                for (uint32_t cb = 0; cb < nsel; cb += 64u)
                {
                    uint32_t cnt = 0, start = 0;
                    if (cb + (uint32_t)lane < nsel)
                    {
                        const uint32_t pk = Lw->vox[cb + (uint32_t)lane];
                        uint32_t       e  = 0;
                        if (voxel_range(g, 0u, cx0 + (pk & 1023u), cy0 + ((pk >> 10) & 1023u), cz0 + (pk >> 20), start, e, true)) cnt = e - start;
                        else start = 0;
                    }
                    batch(start, cnt);
                }
                wave_lds_sync();  // the list is rewritten by the next round
            }
        };
        // a box of more than one round of bricks is entered through the level-2 occupancy words (one u64 per 4x4x4 BRICKS)
        const bool     two_stage = nb > 64u && have_l2;
        const uint32_t vcap      = two_stage ? (uint32_t)PS_VLIST / 2u : (uint32_t)PS_VLIST;
        uint32_t*      s_bl      = Lw->vox + PS_VLIST / 2;  // listed bricks, relative to the box's first brick (10 bits per axis)
        const uint32_t bx0 = cx0 >> 2, by0 = cy0 >> 2, bz0 = cz0 >> 2;
        const float    inv_nbx = 1.0f / (float)max(nbx, 1u), inv_nby = 1.0f / (float)max(nby, 1u);
        constexpr uint32_t CH = PS_VLIST / 2;  // bricks per chunk
        for (uint32_t chunk_lo = 0; nb != 0u; chunk_lo += CH)
        {
            uint32_t n_all = nb;
            if (two_stage)
            {
                const uint32_t bx1 = cx1 >> 2, by1 = cy1 >> 2, bz1 = cz1 >> 2;
                const uint32_t nLx = (bx1 >> 2) - (bx0 >> 2) + 1u, nLy = (by1 >> 2) - (by0 >> 2) + 1u, nLz = (bz1 >> 2) - (bz0 >> 2) + 1u;
                const uint32_t nLn = nLx * nLy * nLz;  // (= nL: at most PS_MARKS)
                const float    inv_nLx = 1.0f / (float)nLx, inv_nLy = 1.0f / (float)nLy;
                const unsigned long long* occ2 = g.occ + g.occ_off[2];
                const uint32_t o2x = g.occ_bx[2], o2y = g.occ_by[2], o2z = g.occ_bz[2];
                // ---- marks: the bricks some query's ball box touches (the first chunk builds them; they stay for the later ones:
                //      nothing else writes the staging area's head before the pass's first batch -- and a second chunk comes only
                //      after batches, so the marks are rebuilt per chunk) ----------------------------------------------------------
                unsigned long long* marks = reinterpret_cast<unsigned long long*>(Lw);
                bool use_marks = true;
                {
                    for (uint32_t i = (uint32_t)lane; i < nLn; i += 64u) marks[i] = 0ull;
                    wave_lds_sync();
                    // this lane's query: the voxels its ball's bounding box spans, clipped to the pass's box, as bricks
                    const float Rb = r + cmx + 4.f * g.slack;
                    uint32_t    qbx0 = 1u, qbx1 = 0u, qby0 = 1u, qby1 = 0u, qbz0 = 1u, qbz1 = 0u;
                    if (grp)
                    {
                        const float ax = fmaxf(qx - Rb, g.bbmin[0]), ay = fmaxf(qy - Rb, g.bbmin[1]), az = fmaxf(qz - Rb, g.bbmin[2]);
                        const float ex = fminf(qx + Rb, g.bbmax[0]), ey = fminf(qy + Rb, g.bbmax[1]), ez = fminf(qz + Rb, g.bbmax[2]);
                        if (!(ax > ex || ay > ey || az > ez))
                        {
                            qbx0 = max(cell_fine(ax, g.ox, g.inv_hf) >> g.shift0, cx0) >> 2, qbx1 = min(cell_fine(ex, g.ox, g.inv_hf) >> g.shift0, cx1) >> 2;
                            qby0 = max(cell_fine(ay, g.oy, g.inv_hf) >> g.shift0, cy0) >> 2, qby1 = min(cell_fine(ey, g.oy, g.inv_hf) >> g.shift0, cy1) >> 2;
                            qbz0 = max(cell_fine(az, g.oz, g.inv_hf) >> g.shift0, cz0) >> 2, qbz1 = min(cell_fine(ez, g.oz, g.inv_hf) >> g.shift0, cz1) >> 2;
                        }
                    }
                    const uint32_t qnx = qbx1 >= qbx0 ? qbx1 - qbx0 + 1u : 0u, qny = qby1 >= qby0 ? qby1 - qby0 + 1u : 0u,
                                   qnz = qbz1 >= qbz0 ? qbz1 - qbz0 + 1u : 0u;
                    const uint32_t qn  = qnx * qny * qnz;  // (a ball box: a few bricks per axis)
                    const uint32_t qmax = __float_as_uint(wave_max_pos(__uint_as_float(min(qn, 0x7F000000u))));
                    if (qmax > 128u) use_marks = false;  // radii of many bricks: the marks would be most of the box anyway
                    else
                    {
                        // the two lanes of a query share its bricks: lane parity = brick parity
                        for (uint32_t i = hi ? 1u : 0u; i < qmax; i += 2u)
                        {
                            if (i < qn)
                            {
                                const uint32_t iz = i / (qnx * qny), rem = i - iz * (qnx * qny), iy = rem / qnx, ix = rem - iy * qnx;
                                const uint32_t Bx = qbx0 + ix, By = qby0 + iy, Bz = qbz0 + iz;
                                const uint32_t id = (((Bz >> 2) - (bz0 >> 2)) * nLy + ((By >> 2) - (by0 >> 2))) * nLx + ((Bx >> 2) - (bx0 >> 2));
                                atomicOr(&marks[id], 1ull << ((Bx & 3u) | ((By & 3u) << 2) | ((Bz & 3u) << 4)));
                            }
                        }
                        wave_lds_sync();
                    }
                }
                uint32_t T = 0;
                for (uint32_t oL = 0; oL < nLn; oL += 64u)
                {
                    const uint32_t     id = oL + (uint32_t)lane;
                    unsigned long long w2 = 0ull;
                    int                pl = 0;
                    if (id < nLn)
                    {
                        const uint32_t row = (uint32_t)(((float)id + 0.5f) * inv_nLx), ix = id - row * nLx;
                        const uint32_t iz  = (uint32_t)(((float)row + 0.5f) * inv_nLy), iy = row - iz * nLy;
                        const uint32_t Lx = (bx0 >> 2) + ix, Ly = (by0 >> 2) + iy, Lz = (bz0 >> 2) + iz;
                        pl = ((int)(Lx * 4u) - (int)bx0) + ((int)(Ly * 4u) - (int)by0) * 1024 + ((int)(Lz * 4u) - (int)bz0) * 1048576;
                        if (Lx < o2x && Ly < o2y && Lz < o2z)
                            w2 = occ2[((size_t)Lz * o2y + Ly) * o2x + Lx] & spread_x(axis_mask(Lx, bx0, bx1)) & spread_y(axis_mask(Ly, by0, by1)) &
                                 spread_z(axis_mask(Lz, bz0, bz1));
                        if (use_marks) w2 &= marks[id];
                    }
                    const uint32_t c2 = (uint32_t)__popcll(w2), i2 = wave_incl_scan(c2, lane);
                    list_bits(w2, pl, c2, i2 + T, chunk_lo, CH, s_bl);
                    T += (uint32_t)__builtin_amdgcn_readlane((int)i2, 63);
                }
                wave_lds_sync();
                n_all = T;
            }
            if (n_all <= chunk_lo) break;
            const uint32_t n_in = min(CH, n_all - chunk_lo);
            for (uint32_t b0_ = 0; b0_ < n_in; b0_ += 64u)
            {
                const bool ok = b0_ + (uint32_t)lane < n_in;
                uint32_t   Bx, By, Bz;
                if (two_stage)
                {
                    const uint32_t pk = ok ? s_bl[b0_ + (uint32_t)lane] : 0u;
                    Bx = bx0 + (pk & 1023u), By = by0 + ((pk >> 10) & 1023u), Bz = bz0 + (pk >> 20);
                }
                else
                {
                    const uint32_t id  = chunk_lo + b0_ + (uint32_t)lane;
                    const uint32_t row = (uint32_t)(((float)id + 0.5f) * inv_nbx), ix = id - row * nbx;
                    const uint32_t iz  = (uint32_t)(((float)row + 0.5f) * inv_nby), iy = row - iz * nby;
                    Bx = bx0 + ix, By = by0 + iy, Bz = bz0 + iz;
                }
                unsigned long long bm;
                int                pb;
                brick_word(ok, Bx, By, Bz, bm, pb);
                if (INSTR && sol >= 4) st_cand += (uint32_t)(bm & 1ull);
                else serve(bm, pb, vcap);
            }
            if (two_stage) wave_lds_sync();  // the brick list is rewritten by the next chunk
            if (n_all <= chunk_lo + CH) break;
        }

        if (__ballot(hq > 0u) != 0ull) flush();  // what is still queued
        if (CERT) rej = fminf(rej, __uint_as_float(Lw->qrej[lane]));  // the exact tests' rejects
        const unsigned long long tp1 = (INSTR && dbg) ? wall_clock64() : 0ull;
        if (INSTR && dbg) tk_pass += tp1 - tp0;
        // ---- merge the two slices of each query slot (both lanes end up with the merged list).  A run-time loop over the partner's
        //      entries (the list is rotated down one place per step): unrolled, the chain was inlined K times
        {
            float    od[K];
            uint32_t os[K];
#pragma unroll
            for (int q = 0; q < K; q++) od[q] = __shfl_xor(kd2[q], 32, 64), os[q] = __shfl_xor(kspos[q], 32, 64);
#pragma unroll 1
            for (int e = 0; e < K; e++)
            {
                float    cd = od[0];
                uint32_t cs = os[0];
                if (__ballot(grp && cs != NONE_U32) == 0ull) break;  // sorted lists: nothing further
#pragma unroll
                for (int q = 0; q + 1 < K; q++) od[q] = od[q + 1], os[q] = os[q + 1];
                od[K - 1] = INFINITY, os[K - 1] = NONE_U32;
                ps_insert<K>(g.pts, kd2, kspos, grp && cs != NONE_U32, cd, cs);  // (only the lanes of this pass: a finished query's list is merged already)
                if (CERT && grp) rej = fminf(rej, cd);
            }
            if (CERT)
            {
                const float orej = __shfl_xor(rej, 32, 64);
                if (grp) rej = fminf(rej, orej);
            }
#pragma unroll
            for (int q = 0; q < K; q++)
                if (grp && q >= (int)knn)
                {
                    if (CERT) rej = fminf(rej, kd2[q]);
                    kd2[q] = INFINITY, kspos[q] = NONE_U32;
                }
        }
        // ---- W > 1: the waves' lists through LDS (every wave's staging area is free now) --------------------------------------
        if (W > 1)
        {
            __syncthreads();  // every wave is through its share of the pass
            uint32_t* pub = reinterpret_cast<uint32_t*>(Lw);
            if (!hi)
            {
#pragma unroll
                for (int q = 0; q < K; q++)
                {
                    const uint32_t o = ((uint32_t)lane * K + q) * 2u;
                    pub[o] = __float_as_uint(kd2[q]), pub[o + 1] = kspos[q];
                }
                Lw->cstart[lane] = __float_as_uint(rej);
                if (lane == 0) Lw->coff[0] = st_cand;
            }
            __syncthreads();
            uint32_t cand_all = st_cand;
#pragma unroll 1
            for (int ow = 1; ow < W; ow++)
            {
                const PsLds*    Lo = L0 + ((wv + ow) & (W - 1));
                const uint32_t* op = reinterpret_cast<const uint32_t*>(Lo);
                cand_all += Lo->coff[0];
#pragma unroll 1
                for (int e = 0; e < K; e++)
                {
                    const uint32_t o  = ((uint32_t)(lane & 31) * K + e) * 2u;
                    float          cd = __uint_as_float(op[o]);
                    uint32_t       cs = op[o + 1];
                    if (__ballot(grp && cs != NONE_U32) == 0ull) break;
                    ps_insert<K>(g.pts, kd2, kspos, grp && cs != NONE_U32, cd, cs);
                    if (CERT && grp) rej = fminf(rej, cd);
                }
                if (CERT && grp) rej = fminf(rej, __uint_as_float(Lo->cstart[lane & 31]));
            }
#pragma unroll
            for (int q = 0; q < K; q++)
                if (grp && q >= (int)knn)
                {
                    if (CERT) rej = fminf(rej, kd2[q]);
                    kd2[q] = INFINITY, kspos[q] = NONE_U32;
                }
            cand_total = cand_all;
            __syncthreads();     // the published lists are read: the areas may be overwritten
        }
        if (grp)
        {
            const float gr = r * (1.0f - 1.0f / 1024.0f) - g.slack;
            kth            = ps_kth(kd2, knn);  // INFINITY while fewer than knn are known
            if (r >= rmax || (gr > 0.f && kth < gr * gr))
            {
                done = true;
                if (CERT)
                {
                    // (3) not staged: outside the ball of radius r + cm; the search itself relies on gr
                    const float cover = fmaxf((r + cm) * (1.0f - 1.0f / 1024.0f) - 3.f * g.slack, gr);
                    // (2) staged, dropped by the prefilter: farther than min(sqrt(kth), r) + cm (kth: the final one)
                    const float pref = fminf(sqrtf(kth), r) + cm - 2.f * g.slack;
                    // (1) evaluated exactly and not kept: rej
                    rej = fmaxf(0.f, fminf(sqrtf(rej), fminf(cover, pref)));  // (a finished lane is in no later pass: rej is free)
                }
            }
            else
            {
                const float rn = (kth < INFINITY) ? sqrtf(kth) * (1.0f + 1.0f / 512.0f) + 4.f * g.slack : 2.0f * r;
                r              = fminf(fmaxf(rn, r * 1.0009765625f), rmax);
            }
        }
        if (INSTR && sol != 0) done = true;  // (a cut pass finds nothing: one pass only, like the product's typical tile)
        if (INSTR && dbg) tk_merge += wall_clock64() - tp1;
    }
    if (CERT && lb_out) *lb_out = rej;
    if (INSTR && tl_info && lane == 0 && wv == 0)
        *tl_info = ((unsigned long long)min(st_pass, 0xFFFFu) << 48) | ((unsigned long long)(st_listed & 0xFFFFu) << 32) | (unsigned long long)(W > 1 ? cand_total : st_cand);
    if (W > 1) st_cand = cand_total;
    if (cand_out) *cand_out = st_cand;
    if (INSTR && dbg && lane == 0 && wv == 0)
    {
        const unsigned long long dt = (unsigned long long)((long long)wall_clock64() - dbg_t0);
        atomicAdd(&dbg[0], 1ull), atomicAdd(&dbg[1], (unsigned long long)st_pass), atomicAdd(&dbg[2], (unsigned long long)st_cand), atomicAdd(&dbg[3], dt);
        atomicMax(&dbg[4], (unsigned long long)st_pass), atomicMax(&dbg[5], (unsigned long long)st_cand), atomicMax(&dbg[6], dt),
            atomicAdd(&dbg[7], (unsigned long long)st_listed);
        atomicMax(&dbg[8], (unsigned long long)st_listed);
        atomicAdd(&dbg[9], (unsigned long long)st_fiter), atomicAdd(&dbg[10], (unsigned long long)st_hits), atomicAdd(&dbg[11], (unsigned long long)st_flush);
        atomicMax(&dbg[50], (dt << 32) | st_hits), atomicMax(&dbg[51], (dt << 32) | st_fiter), atomicMax(&dbg[52], (dt << 32) | st_eiter), atomicMax(&dbg[53], (dt << 32) | st_pos), atomicMax(&dbg[54], (dt << 32) | st_blocks);
        atomicAdd(&dbg[55], (unsigned long long)st_eiter), atomicAdd(&dbg[56], (unsigned long long)st_pos), atomicAdd(&dbg[57], (unsigned long long)st_blocks);
        atomicAdd(&dbg[12], tk_stage), atomicAdd(&dbg[13], tk_pref), atomicAdd(&dbg[14], tk_flush), atomicAdd(&dbg[15], tk_pass), atomicAdd(&dbg[48], tk_merge);
        atomicAdd(&dbg[16 + min(23, 63 - (int)__clzll((long long)(dt | 1ull)))], 1ull);
        atomicMax(&dbg[49], (dt << 40) | (((unsigned long long)st_pass & 0xFFull) << 32) | ((unsigned long long)st_cand & 0xFFFFFFFFull));
    }
}

}  // namespace mp2p
