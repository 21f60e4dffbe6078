// nn_seltile.hip -- round 5: the tile kernel of the point-to-point search with the voxels of a pass SELECTED on the matrix
// pipe, two matrix instructions per block instead of three, and (optionally) the per-query prologue fused in.  Included by
// nn_query.hip; DESIGN.md section 4 "Round 5" has every measurement quoted here.
//
// What rounds 1-4 staged for a tile of 32 Morton-consecutive queries was every occupied voxel within the widest radius of
// the BOUNDING BOX of the group's queries.  A CPU model of the bench chain (tools/cand_model.py: scene B, 1 M x 10 M, warm
// start from the previous pose) says what that rule costs: 950-2 000 points per tile, p95 3 400-4 800, single tiles 10^5
// (the tiles that ran out of budget and sent 3.6 % of the layer to the one-query kernel = 35 % of the search) against
// 370-790 (p95 1 200-2 800) for the voxels some query's BALL really reaches -- the box rule stages the slab between a
// wall and the tile, the balls only touch the wall near each query's foot point.  The exact rule is a box test per
// (voxel, query) pair -- dearer than the distance tests it saves.  Its sphere relaxation is a bilinear form:
//
//     voxel v is needed by query m  <=  |c_v - q_m|^2 - (r_m + rho)^2 <= 0        (c_v = centre, rho = half diagonal)
//
// i.e. the same v_mfma_f32_32x32x2_f32 as the distance prefilter with the roles swapped: rows = the tile's 32 queries,
// columns = 32 listed voxels, so that a LANE ends up with one voxel's values against 16 queries: an integer min-tree over
// its 16 accumulators + one ballot say which of the 32 voxels anybody needs (~40 instructions per 32 voxels x 32 queries;
// the model: 480-900 points per tile, within 1.15-1.3 x of the exact rule; measured: 1 119 -> 771 staged per tile, the
// longest tile 86 000 -> 9 000, queries handed to the one-query kernel 36 000 -> 12 000 -> 480 once the budget went up).
// Only the needed voxels are resolved through the directory and staged; the rest of a pass (staging rounds, distance
// prefilter, exact recomputation of the survivors, claims) is the tile kernel of round 4.  Exactness is untouched: a
// query is final when its best distance is below the radius of the BALL whose voxels were all staged (a voxel that
// intersects the ball has a point within r of the query, hence its centre within r + rho); the selection only errs
// towards staging more (tolerance = four times the prefilter's proven error bound + the fp32 slack of the addressing).
//
// Other changes of round 5, each measured on the headline chain (tile kernel 0.325 ms in round 4):
//   * K = 4: the constant-per-column term |q'|^2 moves from the matrix product into the lane's limit (with an offset that
//     keeps the integer min-tree's values non-negative): TWO matrix instructions per block of 32 candidates, not three --
//     the matrix pipe (64 cycles per instruction and SIMD) is the prefilter's floor (0.117 -> 0.081 ms of the tile kernel);
//   * every voxel is tested against every query's OWN ball, so all pending queries of a tile form ONE pass while their
//     common box is a few bricks wide (1.36 -> 1.25 passes per tile);
//   * the occupied voxels are listed by a wave-uniform loop over the non-empty bricks (v_mbcnt places), and a box of more
//     than 64 bricks is entered through the level-2 occupancy words (one u64 per 4x4x4 BRICKS): a spatially loose tile of
//     far-field points used to list 10^4 nearly empty bricks 64 per round (150 us for 4 000 candidates);
//   * nothing is handed to the one-query kernel for being heavy or alone any more (budget 24 576, long tiles start first by
//     the cost classes; an isolated query's ball stages little): that kernel went from 0.19 ms to 0.02 ms;
//   * the empty-room bound of a query with nothing in reach is found by the tile itself.
//
// Fused prologue (DIRECT): tile t serves the queries 32 t .. 32 t + 31 of the Morton-sorted local layer itself -- transform,
// bounding box, threshold rule, MatchState / visit list, warm start, skip certificate: what nn_lane_kernel does -- and no
// pending list exists.  Measured SLOWER on the 1 M-point chain (tile 0.358 vs lane 0.020 + tile 0.286 ms: without the lane
// kernel's cost classes the long tiles start late) and faster on small layers (one launch less, nothing to order: C2
// 3 390 vs 3 300 it/s): the launch code takes it for layers of at most 262 144 points.
#include "device_utils.hpp"

namespace mp2p
{
// the warm-started per-query set-up of nn_lane_kernel (see there for the reasoning), for one query per call
__device__ __forceinline__ void query_prologue(const NNArgs& a, const GridView& g, uint32_t qi, bool valid, float& qx, float& qy, float& qz,
                                               float& thr, float& rmax, float& r, bool& visited, bool& active, bool& done, float& best_d2,
                                               uint32_t& best_idx, uint32_t& best_spos, uint32_t& orig, float& lb2_out, float& lb2nd_out)
{
    float4 lp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) lp = a.lpts[qi];
    uint4 h = make_uint4(NONE_U32, 0u, 0u, 0u);
    if (valid && a.use_hint) h = a.rec[qi];
    orig    = __float_as_uint(lp.w);
    visited = valid;
    if (a.rank && valid) visited = a.rank[orig] != NONE_U32;
    compose_point_f(a.pose, lp.x, lp.y, lp.z, qx, qy, qz);  // K1
    const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
    thr  = fadd(a.maxDistSq, fmul(a.angSq, normSq));  // Matcher_Points_DistanceThreshold.cpp:223-225, 256-259
    rmax = sqrtf(thr) * 1.002f + g.slack;
    active = visited && (normSq < INFINITY);
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // :218-220
    r        = fminf(a.r0, rmax);
    done     = !active;
    best_d2  = INFINITY;
    best_idx = NONE_U32, best_spos = NONE_U32;
    lb2_out = -1.f, lb2nd_out = 0.f;
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
        else if (lb > r * (1.0f - 1.0f / 1024.0f) - g.slack)
            r = fminf(fmaxf(hr > 0.f ? fminf(hr, 2.0f * lb) : 2.0f * lb, r), rmax);
        if (a.cert_read && !done && hr > 0.f)
        {  // the skip certificate (nn_lane_kernel)
            const float l2   = a.lb2nd[qi];
            const float room = l2 - disp * 1.00001f - 0.25f * g.slack;
            if (l2 > 0.f && sqrtf(best_d2) * 1.00001f + 0.5f * g.slack < room) done = true, lb2nd_out = room;
        }
    }
}

// SOL (speed-of-light decomposition, profiles/r05_tile_sol.txt; results are NOT valid, nothing is written): the kernel cut after
// 3 = the prologue, 4 = + pass set-up and the voxel list, 5 = + selection and directory look-up, 1 = + staging,
// 2 = + the matrix-pipe prefilter and its min-tree (no recomputation); 0 = the product
template <bool INSTR, bool CERT, bool DIRECT, int WAVES, int SOL = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void nn_seltile_kernel(const NNArgs a)
{
    constexpr int Q = 32;
    __shared__ __attribute__((aligned(16))) float s_x[NN_CAP];
    __shared__ __attribute__((aligned(16))) float s_y[NN_CAP];
    __shared__ __attribute__((aligned(16))) float s_z[NN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_idx[NN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_spos[NN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_owner[NN_CAP];
    __shared__ uint32_t s_cstart[64];
    __shared__ uint32_t s_coff[64];
    __shared__ uint32_t s_vox[NN_TVLIST];  // occupied voxels of the box (listed), then those some query needs (compacted in place)
    static_assert(NN_CLAIM_SLOTS * sizeof(unsigned long long) <= NN_CAP * sizeof(float), "claim table fits s_x");
    unsigned long long* s_claim = reinterpret_cast<unsigned long long*>(s_x);

    const GridView& g    = a.g;
    const int       lane = threadIdx.x;
    const uint32_t  tile = blockIdx.x;
    // one class in DIRECT mode (every tile is a fixed slice of the layer), two (hard first) behind the lane kernel
    const uint32_t segs8          = (a.n_seg + 7u) / 8u;
    const uint32_t tiles_per_list = segs8 * 8u * a.tiles_per_seg;
    const uint32_t cls            = (!DIRECT && tile >= tiles_per_list) ? 1u : 0u;  // 0 = hard (first), 1 = easy
    const uint32_t tl             = tile - cls * tiles_per_list;
    uint32_t       seg, tk;
    if (a.xcd_map)
    {
        const uint32_t x = tl & 7u, j = tl >> 3;
        const uint32_t sl = j / a.tiles_per_seg;
        tk = j - sl * a.tiles_per_seg, seg = sl * 8u + x;
    }
    else
        seg = tl / a.tiles_per_seg, tk = tl - seg * a.tiles_per_seg;
    if (seg >= a.n_seg) return;
    seg += a.seg_base;
    uint32_t n_pend;
    if (DIRECT)
    {
        const unsigned long long first = (unsigned long long)seg * a.seg_cap;
        n_pend = first >= a.n_l ? 0u : (uint32_t)min((unsigned long long)a.seg_cap, (unsigned long long)a.n_l - first);
    }
    else
        n_pend = a.q_counters[((size_t)(cls ? 2 : 0) * NN_MAX_SEG + seg) * NN_CNT_STRIDE];
    if (tk * (uint32_t)Q >= n_pend) return;
    const unsigned long long tl0 = wall_clock64();
    const uint32_t cand_cap = cls ? a.tile_cand_cap_easy : a.tile_cand_cap;
    const int      qslot = lane & (Q - 1);
    const int      slice = lane / Q;
    const bool     hi    = lane >= 32;
    const bool     valid = tk * Q + qslot < n_pend;

    uint32_t qi = 0, orig = 0, best_idx = NONE_U32, best_spos = NONE_U32;
    float    qx = 0.f, qy = 0.f, qz = 0.f, thr = 0.f, rmax = 0.f, r = 0.f, best_d2 = INFINITY;
    bool     active = true, done = false;
    float    lb2_out = -1.f, lb2nd_out = 0.f;  // (DIRECT) bounds of a query the prologue finished without a search
    if (DIRECT)
    {
        qi = seg * a.seg_cap + tk * Q + (uint32_t)qslot;
        bool visited;
        query_prologue(a, g, qi, valid, qx, qy, qz, thr, rmax, r, visited, active, done, best_d2, best_idx, best_spos, orig, lb2_out, lb2nd_out);
        // bounding box of ALL transformed local points of the tile (Matcher_Points_Base.cpp:186-196); box index = tile of the layer
        const float bx0 = wave_min_nn((visited && qx == qx) ? qx : INFINITY), by0 = wave_min_nn((visited && qy == qy) ? qy : INFINITY),
                    bz0 = wave_min_nn((visited && qz == qz) ? qz : INFINITY);
        const float bx1 = wave_max_nn((visited && qx == qx) ? qx : -INFINITY), by1 = wave_max_nn((visited && qy == qy) ? qy : -INFINITY),
                    bz1 = wave_max_nn((visited && qz == qz) ? qz : -INFINITY);
        if (lane == 0)
        {
            float* o = a.tile_bbox + (size_t)(qi / Q) * 6;
            o[0] = bx0, o[1] = by0, o[2] = bz0, o[3] = bx1, o[4] = by1, o[5] = bz1;
        }
        if (INSTR)
        {
            const uint32_t n_pend0 = (uint32_t)__popcll(__ballot(!done && slice == 0));
            const uint32_t n_skip  = (uint32_t)__popcll(__ballot(active && slice == 0 && (lb2_out >= 0.f || lb2nd_out > 0.f)));
            if (lane == 0) atomicAdd(&a.counters[47], (unsigned long long)n_pend0), atomicAdd(&a.counters[48], (unsigned long long)n_skip);
        }
    }
    else
    {
        const size_t pslot = (size_t)cls * a.list_cap + (size_t)seg * a.seg_cap + tk * Q + qslot;
        uint4        w = make_uint4(0u, 0u, __float_as_uint(INFINITY), NONE_U32), wq = make_uint4(0u, 0u, 0u, NONE_U32);
        if (valid) w = a.pend[pslot], wq = a.pend_q[pslot];
        qi = w.x;
        if (valid) orig = __float_as_uint(a.lpts[qi].w);
        qx = __uint_as_float(wq.x), qy = __uint_as_float(wq.y), qz = __uint_as_float(wq.z);
        const float normSq = fadd(fadd(fmul(qx, qx), fmul(qy, qy)), fmul(qz, qz));
        thr  = fadd(a.maxDistSq, fmul(a.angSq, normSq));
        rmax = sqrtf(thr) * 1.002f + g.slack;
        r    = valid ? __uint_as_float(w.y) : 0.f;
        best_d2 = __uint_as_float(w.z), best_idx = w.w, best_spos = wq.w;
        active = valid, done = !valid;
    }
    const bool pro_done = DIRECT && valid && done;  // finished (or inactive) before any search
    bool       deferred = false;
    // bounding box of the tile's pending queries, once per tile: a pass's box is this box grown by its widest radius (the
    // queries do not move between passes; the six wave reductions per pass were a tenth of the pass's instructions)
    const float tqx0 = wave_min_nn(!done ? qx : INFINITY), tqy0 = wave_min_nn(!done ? qy : INFINITY), tqz0 = wave_min_nn(!done ? qz : INFINITY);
    const float tqx1 = wave_max_nn(!done ? qx : -INFINITY), tqy1 = wave_max_nn(!done ? qy : -INFINITY), tqz1 = wave_max_nn(!done ? qz : -INFINITY);

    // a query with no candidate at all and a radius beyond what a tile should carry: the one-query kernel's
    // nearest-voxel-first order finds a bound cheaply
    {
        const bool               wide  = !done && r > a.r_defer && best_idx == NONE_U32;
        const unsigned long long wmask = __ballot(wide);
        if (wmask)
        {
            if (SOL == 0) defer_lanes<Q>(a, seg, wide, wmask, lane, slice, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (wide) done = true, deferred = true;
        }
    }

    if (SOL == 3) done = true;
    uint32_t        st_pass = 0, st_cells = 0, st_cand = 0, st_defer = 0, st_listed = 0, st_needed = 0;
    const long long t_start = INSTR ? (long long)wall_clock64() : 0;
    constexpr bool  track = CERT;
    int             t1 = 0x7FFFFFFF, t2 = 0x7FFFFFFF;
    float           lbq = 0.f;
    auto            ins = [&](int x) __attribute__((always_inline)) { t2 = min(t2, max(t1, x)), t1 = min(t1, x); };
    const uint32_t  obx = g.occ_bx[0], oby = g.occ_by[0], obz = g.occ_bz[0];
    const unsigned long long* occ0 = g.occ + g.occ_off[0];
    const float     hs  = g.hf * (float)(1u << g.shift0);  // level-0 voxel edge
    const float     rho = hs * 0.8660255f;                 // half diagonal (rounded up)

    while (true)
    {
        const unsigned long long pend = __ballot(!done);
        if (pend == 0ull) break;
        // ---- the GROUP this pass serves: the pending queries near the first pending one, of comparable radius ----------
        // (round 5: what a pass stages no longer depends on the group's box -- every voxel is tested against every query's own
        //  ball -- so ALL pending queries form one group as long as their common box is a few bricks wide: a pass per radius
        //  class re-lists and re-stages the same neighbourhood; only a box too wide to list cheaply is cut by the old rule)
        bool  grp = !done;
        float rmax_t = wave_max_pos(grp ? r : 0.f);
        // box of the group's queries, and the box their balls reach (= the queries' box grown by the widest radius)
        float qlx = tqx0, qly = tqy0, qlz = tqz0, qhx = tqx1, qhy = tqy1, qhz = tqz1;
        {
            const float wide = a.grp_all_bricks * 4.f * hs;  // (edge of the box in bricks, about)
            const float w2   = 2.f * rmax_t;
            if (qhx - qlx + w2 > wide || qhy - qly + w2 > wide || qhz - qlz + w2 > wide)
            {

                const int   seed = __ffsll((long long)pend) - 1;
                const float sx = readlane_f(qx, seed), sy = readlane_f(qy, seed), sz = readlane_f(qz, seed);
                const float sr = readlane_f(r, seed);
                const float G  = a.grp_factor * sr;
                grp = !done && fabsf(qx - sx) <= G && fabsf(qy - sy) <= G && fabsf(qz - sz) <= G && r <= 2.0f * sr;
                rmax_t = wave_max_pos(grp ? r : 0.f);
                qlx = wave_min_nn(grp ? qx : INFINITY), qly = wave_min_nn(grp ? qy : INFINITY), qlz = wave_min_nn(grp ? qz : INFINITY);
                qhx = wave_max_nn(grp ? qx : -INFINITY), qhy = wave_max_nn(grp ? qy : -INFINITY), qhz = wave_max_nn(grp ? qz : -INFINITY);
            }
        }
        const float lox = qlx - rmax_t, loy = qly - rmax_t, loz = qlz - rmax_t, hix = qhx + rmax_t, hiy = qhy + rmax_t, hiz = qhz + rmax_t;
        const unsigned long long gmask = __ballot(grp);
        if (__popcll(gmask) <= (int)a.coop_max * 2)
        {  // a few isolated queries: the one-query kernel
            if (SOL == 0) st_defer += defer_lanes<Q>(a, seg, grp, gmask, lane, slice, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (grp) done = true, deferred = true;
            continue;
        }
        st_pass++;
        st_cand += NN_PASS_COST;
        if (track) t1 = t2 = 0x7FFFFFFF;

        // ---- the box clipped to the layer; level-0 voxels and their 4x4x4 bricks ---------------
        const float prune  = rmax_t + 4.f * g.slack;
        const float prune2 = prune * prune;
        const float ocx = 0.5f * (lox + hix), ocy = 0.5f * (loy + hiy), ocz = 0.5f * (loz + hiz);
        // prefilter / selection operands on coordinates centred on the box (nn_tile_kernel: the proven bound mtol)
        const float hx = 0.5f * (hix - lox) + hs, hy = 0.5f * (hiy - loy) + hs, hz = 0.5f * (hiz - loz) + hs;
        const float mtol = (hx * hx + hy * hy + hz * hz) * (1.0f / 32768.0f);
        const float cqx = qx - ocx, cqy = qy - ocy, cqz = qz - ocz;
        // ---- K = 4: TWO v_mfma_f32_32x32x2_f32 per block instead of round 4's three (the matrix pipe is the prefilter's floor:
        //      64 cycles per instruction and SIMD).  The term |q'|^2 is constant per column (= per lane) and moves into the lane's
        //      limit; so that the integer min-tree still sees non-negative values the candidate side carries the offset
        //      beta = hx^2 + hy^2 + hz^2 >= |q'|^2 (group members lie in the box):
        //        T = [c'x c'y | c'z  |c'|^2 + beta] . [-2q'x -2q'y | -2q'z  1] = d2 - |q'|^2 + beta >= 0,   tested against lim - |q'|^2 + beta
        //      (a lane outside the group may see negative values: it only collects upper bounds, and a missed one costs nothing but
        //       a later update).  Every term is still bounded by 4 (hx^2 + hy^2 + hz^2): the error bound mtol stands.
        const float beta = hx * hx + hy * hy + hz * hz;
        const float qn2  = cqx * cqx + cqy * cqy + cqz * cqz;
        const float b0 = -2.0f * (hi ? cqy : cqx);
        const float b1 = hi ? 1.0f : -2.0f * cqz;
        const float o0 = hi ? ocy : ocx, o1 = hi ? -beta : ocz;  // (hi lanes read |c'|^2 and ADD beta: s_n - (-beta))
        const float loff = beta - qn2;                           // lane's limit in the shifted scale: lim + loff
        // selection (rows = queries, columns = voxels): [-2q'x -2q'y | -2q'z  |q'|^2 - R^2 + 4 beta] . [c'x c'y | c'z 1] >= 0 for group
        // members (|2 q'.c'| <= 2 beta, R^2 <= beta), tested per column against tol - |c'|^2 + 4 beta; R = r + rho (+ slack);
        // a lane outside the group never needs a voxel
        const float Rq   = r + rho + 4.f * g.slack;
        const float sel1 = hi ? (grp ? qn2 - Rq * Rq + 4.f * beta : 1e30f) : -2.0f * cqz;
        const float stol = 4.0f * mtol + 1e-12f;
        // clipped box in level-0 voxels
        uint32_t cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0, nbx = 0, nby = 0, nb = 0;
        {
            const float clx = fmaxf(lox, g.bbmin[0]), cly = fmaxf(loy, g.bbmin[1]), clz = fmaxf(loz, g.bbmin[2]);
            const float chx = fminf(hix, g.bbmax[0]), chy = fminf(hiy, g.bbmax[1]), chz = fminf(hiz, g.bbmax[2]);
            if (!((clx > chx) || (cly > chy) || (clz > chz)))
            {
                cx0 = cell_fine(clx, g.ox, g.inv_hf) >> g.shift0, cx1 = cell_fine(chx, g.ox, g.inv_hf) >> g.shift0;
                cy0 = cell_fine(cly, g.oy, g.inv_hf) >> g.shift0, cy1 = cell_fine(chy, g.oy, g.inv_hf) >> g.shift0;
                cz0 = cell_fine(clz, g.oz, g.inv_hf) >> g.shift0, cz1 = cell_fine(chz, g.oz, g.inv_hf) >> g.shift0;
                nbx = (cx1 >> 2) - (cx0 >> 2) + 1u, nby = (cy1 >> 2) - (cy0 >> 2) + 1u;
                const uint32_t nbz = (cz1 >> 2) - (cz0 >> 2) + 1u;
                // (a box of 2^20 voxels per axis is 2^18 bricks per axis: the product fits 64 bits; a group's box is a few metres)
                const unsigned long long nbl = (unsigned long long)nbx * nby * nbz;
                nb = (uint32_t)min(nbl, 0xFFFFFFFFull);
            }
        }
        // a box beyond what the voxel list can address (10 bits per axis relative to its corner) or an absurd number of bricks
        // (thresholds of tens of metres): the one-query kernel picks its own level for such a radius
        if (cx1 - cx0 >= 1024u || cy1 - cy0 >= 1024u || cz1 - cz0 >= 1024u || nb > 32768u)
        {
            if (SOL == 0) st_defer += defer_lanes<Q>(a, seg, grp, gmask, lane, slice, qi, r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (grp) done = true, deferred = true;
            continue;
        }
        bool over = false;

        // one batch of <= 64 resolved voxels (lane = voxel: start, cnt): staged in rounds of NN_CAP points, tested against the tile's queries
        auto batch = [&](uint32_t start, uint32_t cnt) __attribute__((always_inline)) {
            const uint32_t incl  = wave_incl_scan(cnt, lane);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0) return;
            const uint32_t off = incl - cnt;
            s_cstart[lane] = start;
            s_coff[lane]   = off;
            st_cand += total;
            for (uint32_t base = 0; base < total && !over; base += NN_CAP)
            {
                const uint32_t m = min((uint32_t)NN_CAP, total - base);
                over = st_cand - total + base + m > cand_cap;  // (this round is still scanned)
                const uint32_t m_pad = (m + 31u) & ~31u;
                // ---- stage: lane l fills slots 4l..4l+3; a segmented broadcast tells which voxel a slot belongs to ----------
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
                    // round 6: all four loads issued before any is waited for.  Written as `if (ok) c4[k] = pts[src[k]]` (rounds 1-5)
                    // the compiler gave every load an exec-masked block of its own with its own s_waitcnt vmcnt(0): four SERIAL round
                    // trips per staging round -- the "4 us per round of 256 that no prefetch moved" of DESIGN.md section 4.  An
                    // out-of-range slot reads point 0 and is overwritten by the padding.
                    // (the sorted positions go to LDS BEFORE the loads are issued: four registers less across the wait -- 28 -> 12 bytes of
                    //  scratch per lane in the 128-register build)
                    *reinterpret_cast<uint4*>(&s_spos[t0]) = make_uint4(src[0], src[1], src[2], src[3]);
#pragma unroll
                    for (int k = 0; k < 4; k++) c4[k] = g.pts[(t0 + k < m) ? src[k] : 0u];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (!(t0 + k < m)) c4[k] = make_float4(1e18f, 0.f, 0.f, __uint_as_float(NONE_U32));  // padding: far but FINITE (nn_tile_kernel)
                    *reinterpret_cast<float4*>(&s_x[t0]) = make_float4(c4[0].x, c4[1].x, c4[2].x, c4[3].x);
                    *reinterpret_cast<float4*>(&s_y[t0]) = make_float4(c4[0].y, c4[1].y, c4[2].y, c4[3].y);
                    *reinterpret_cast<float4*>(&s_z[t0]) = make_float4(c4[0].z, c4[1].z, c4[2].z, c4[3].z);
                    *reinterpret_cast<uint4*>(&s_idx[t0]) =
                        make_uint4(__float_as_uint(c4[0].w), __float_as_uint(c4[1].w), __float_as_uint(c4[2].w), __float_as_uint(c4[3].w));
                    {  // |c - centre|^2 over the lane's own four owner slots (read already)
                        float n4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            const float ex_ = c4[k].x - ocx, ey_ = c4[k].y - ocy, ez_ = c4[k].z - ocz;
                            n4[k] = ex_ * ex_ + ey_ * ey_ + ez_ * ez_;
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
                if (SOL != 1)
                {
                    const float* s_n  = reinterpret_cast<const float*>(s_owner);
                    const float* s_a0 = hi ? s_y : s_x;
                    const float* s_a1 = hi ? s_n : s_z;
                    float        lim  = best_d2 * 1.000001f + mtol + loff;
                    for (uint32_t blk = 0; blk < m_pad; blk += 32u)
                    {
                        const uint32_t c   = blk + ((uint32_t)lane & 31u);
                        const float    a0v = s_a0[c] - o0, a1v = s_a1[c] - o1;
                        f32x16         acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1, acc, 0, 0, 0);
                        int g4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            g4[k] = min(min(__float_as_int(acc[4 * k]), __float_as_int(acc[4 * k + 1])),
                                        min(__float_as_int(acc[4 * k + 2]), __float_as_int(acc[4 * k + 3])));
                        const int   mni = min(min(g4[0], g4[1]), min(g4[2], g4[3]));
                        const float mn  = __int_as_float(mni);
                        if (SOL == 2)
                        {  // (keeps the min-tree alive without the recomputation)
                            if (mn <= lim && mn < -1e30f) best_d2 = mn;
                            continue;
                        }
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
                                    for (int rr = 4 * k; rr < 4 * k + 4; rr++)
                                    {
                                        if (track) ins(__float_as_int(acc[rr]));
                                        if (acc[rr] <= lim)
                                        {
                                            const uint32_t j  = blk + (uint32_t)((rr & 3) + 8 * (rr >> 2)) + (hi ? 4u : 0u);
                                            const float    dd = dist2(qx, qy, qz, s_x[j], s_y[j], s_z[j]);
                                            if (dd <= best_d2)
                                            {
                                                const uint32_t ci = s_idx[j];
                                                if (dd < best_d2 || ci < best_idx)
                                                {
                                                    best_d2   = dd;
                                                    best_idx  = ci;
                                                    best_spos = s_spos[j];
                                                    lim       = best_d2 * 1.000001f + mtol + loff;
                                                }
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
        };

        // ---- bricks of the box (lane = brick) -> occupied voxels listed in LDS -> SELECTED on the matrix pipe -> resolved -> staged ----
        // the masked occupancy word of one level-0 brick (0 when it is beyond the layer or farther from the group's queries than their
        // widest radius) and its first voxel relative to (cx0, cy0, cz0), packed 10 bits per axis as one integer sum
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
        // the set bits of the lanes' words -> list entries base + (x, y, z of the bit) in out[0 .. cap), those of rank r0 .. r0 + cap - 1.
        // Few non-empty words (the normal pass: ~5): a wave-uniform loop over them, lane j = bit j, its place from the bits below
        // (v_mbcnt) -- ~25 instructions per word; the per-lane bit loop costs ~20 per iteration and runs as long as the fullest word
        auto list_bits = [&](unsigned long long bm, int pb, uint32_t bcnt, uint32_t bincl, uint32_t r0, uint32_t cap, uint32_t* out) __attribute__((always_inline)) {
            unsigned long long nz = __ballot(bcnt != 0u);
            if (__popcll(nz) <= 24)
            {
                const int      lj   = (lane & 3) + ((lane >> 2) & 3) * 1024 + (lane >> 4) * 1048576;
                const uint32_t blo  = (uint32_t)bm, bhi = (uint32_t)(bm >> 32), bbase = bincl - bcnt;
                // (only the words with entries in this round: a dense box is listed in several rounds, each walking all its words
                //  made the listing quadratic -- 230 us for one tile)
                nz &= __ballot(bbase < r0 + cap && bincl > r0);
                while (nz)
                {
                    const int b = __ffsll((long long)nz) - 1;
                    nz &= nz - 1ull;
                    const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)blo, b), whi = (uint32_t)__builtin_amdgcn_readlane((int)bhi, b);
                    const uint32_t rank = (uint32_t)__builtin_amdgcn_readlane((int)bbase, b) + __builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u));
                    const bool     has  = (((hi ? whi : wlo) >> (lane & 31)) & 1u) != 0u;
                    if (has && rank >= r0 && rank < r0 + cap) out[rank - r0] = (uint32_t)(__builtin_amdgcn_readlane(pb, b) + lj);
                }
            }
            else
            {
                uint32_t           rank = bincl - bcnt;
                unsigned long long mm   = (rank < r0 + cap && bincl > r0) ? bm : 0ull;
                while (mm)
                {
                    const int bit = __ffsll((long long)mm) - 1;
                    mm &= mm - 1ull;
                    if (rank >= r0 && rank < r0 + cap) out[rank - r0] = (uint32_t)(pb + (bit & 3) + ((bit >> 2) & 3) * 1024 + (bit >> 4) * 1048576);
                    rank++;
                }
            }
        };
        // one round of <= 64 bricks (lane = brick: bm, pb): list, select, resolve, stage + scan
        auto serve = [&](unsigned long long bm, int pb, uint32_t vcap) __attribute__((always_inline)) {
            const uint32_t bcnt   = (uint32_t)__popcll(bm);
            const uint32_t bincl  = wave_incl_scan(bcnt, lane);
            const uint32_t vtotal = (uint32_t)__builtin_amdgcn_readlane((int)bincl, 63);
            if (INSTR || SOL != 0) st_cells += 64u;
            for (uint32_t r0 = 0; r0 < vtotal && !over; r0 += vcap)
            {
                list_bits(bm, pb, bcnt, bincl, r0, vcap, s_vox);
                __syncthreads();
                const uint32_t nv = min(vcap, vtotal - r0);
                if (INSTR || SOL != 0) st_listed += nv;
                if (SOL == 4)
                {
                    st_cand += s_vox[lane] & 1u;
                    __syncthreads();
                    continue;
                }
                // ---- selection: 32 listed voxels (columns) x the tile's 32 queries (rows) per three MFMAs ----------------------
                uint32_t nsel = 0;
                for (uint32_t vb = 0; vb < nv; vb += 32u)
                {
                    const uint32_t vi = vb + ((uint32_t)lane & 31u);
                    uint32_t       pk = 0u;
                    float          ccx = 1e15f, ccy = 0.f, ccz = 0.f;  // padding: far, finite
                    if (vi < nv)
                    {
                        pk  = s_vox[vi];
                        ccx = (g.ox + ((float)(cx0 + (pk & 1023u)) + 0.5f) * hs) - ocx;
                        ccy = (g.oy + ((float)(cy0 + ((pk >> 10) & 1023u)) + 0.5f) * hs) - ocy;
                        ccz = (g.oz + ((float)(cz0 + (pk >> 20)) + 0.5f) * hs) - ocz;
                    }
                    const float v0 = hi ? ccy : ccx;
                    const float v1 = hi ? 1.0f : ccz;
                    const float vlim = stol - (ccx * ccx + ccy * ccy + ccz * ccz) + 4.f * beta;  // this column's limit
                    f32x16      acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, v0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sel1, v1, acc, 0, 0, 0);
                    int mni = min(min(__float_as_int(acc[0]), __float_as_int(acc[1])), min(__float_as_int(acc[2]), __float_as_int(acc[3])));
#pragma unroll
                    for (int k = 1; k < 4; k++)
                        mni = min(mni, min(min(__float_as_int(acc[4 * k]), __float_as_int(acc[4 * k + 1])),
                                           min(__float_as_int(acc[4 * k + 2]), __float_as_int(acc[4 * k + 3]))));
                    // (a negative value is a negative integer: the minimum is then some value <= 0 <= stol; among the non-negative
                    //  ones the integer order is the float order)
                    const unsigned long long nb64 = __ballot(vi < nv && __int_as_float(mni) <= vlim);
                    const uint32_t           m32  = (uint32_t)nb64 | (uint32_t)(nb64 >> 32);  // both halves hold the same voxel
                    if (!hi && ((m32 >> lane) & 1u)) s_vox[nsel + (uint32_t)__popc(m32 & ((1u << lane) - 1u))] = pk;  // (writes at or below vb + lane)
                    nsel += (uint32_t)__popc(m32);
                }
                __syncthreads();
                if (INSTR || SOL != 0) st_needed += nsel;
                for (uint32_t cb = 0; cb < nsel && !over; cb += 64u)
                {
                    uint32_t cnt = 0, start = 0;
                    if (cb + (uint32_t)lane < nsel)
                    {
                        const uint32_t pk = s_vox[cb + (uint32_t)lane];
                        uint32_t       e  = 0;
                        if (voxel_range(g, 0u, cx0 + (pk & 1023u), cy0 + ((pk >> 10) & 1023u), cz0 + (pk >> 20), start, e, true)) cnt = e - start;
                        else start = 0;
                    }
                    if (SOL == 5)
                    {
                        st_cand += (cnt + start) & 1u;
                        continue;
                    }
                    batch(start, cnt);
                }
                __syncthreads();  // the list is rewritten by the next round
            }
        };
        // A box of more than one round of bricks (wide balls, or a spatially loose tile of far-field points: the box of 32 points
        // 10 m apart holds 10^4 bricks, nearly all empty, and listing them 64 per round WAS such a tile: 150 us for 4 000
        // candidates) is entered one level up: the level-2 occupancy word of 4x4x4 BRICKS says which of them hold anything; only
        // those are visited.  (The lists share s_vox: voxels in its first half, bricks in the second, 256 each.)
        const bool     two_stage = nb > 64u && g.n_levels > 2u && g.occ_off[2] != OCC_NONE;
        const uint32_t vcap      = two_stage ? (uint32_t)NN_TVLIST / 2u : (uint32_t)NN_TVLIST;
        uint32_t*      s_bl      = s_vox + NN_TVLIST / 2;  // listed bricks, relative to the box's first brick (10 bits per axis)
        const uint32_t bx0 = cx0 >> 2, by0 = cy0 >> 2, bz0 = cz0 >> 2;
        const float    inv_nbx = 1.0f / (float)max(nbx, 1u), inv_nby = 1.0f / (float)max(nby, 1u);
        constexpr uint32_t CH = NN_TVLIST / 2;  // bricks per chunk
        for (uint32_t chunk_lo = 0; !over; chunk_lo += CH)
        {
            uint32_t n_all = nb;  // bricks to visit in all
            if (two_stage)
            {
                // phase A: the level-2 words of the box -> the occupied bricks of rank chunk_lo .. chunk_lo + CH - 1
                const uint32_t bx1 = cx1 >> 2, by1 = cy1 >> 2, bz1 = cz1 >> 2;
                const uint32_t nLx = (bx1 >> 2) - (bx0 >> 2) + 1u, nLy = (by1 >> 2) - (by0 >> 2) + 1u, nLz = (bz1 >> 2) - (bz0 >> 2) + 1u;
                const uint32_t nL  = nLx * nLy * nLz;  // (<= nb: at most 32 768)
                const float    inv_nLx = 1.0f / (float)nLx, inv_nLy = 1.0f / (float)nLy;
                const unsigned long long* occ2 = g.occ + g.occ_off[2];
                const uint32_t o2x = g.occ_bx[2], o2y = g.occ_by[2], o2z = g.occ_bz[2];
                uint32_t T = 0;
                for (uint32_t oL = 0; oL < nL; oL += 64u)
                {
                    const uint32_t id = oL + (uint32_t)lane;
                    unsigned long long w2 = 0ull;
                    int                pl = 0;
                    if (id < nL)
                    {
                        const uint32_t row = (uint32_t)(((float)id + 0.5f) * inv_nLx), ix = id - row * nLx;
                        const uint32_t iz  = (uint32_t)(((float)row + 0.5f) * inv_nLy), iy = row - iz * nLy;
                        const uint32_t Lx = (bx0 >> 2) + ix, Ly = (by0 >> 2) + iy, Lz = (bz0 >> 2) + iz;
                        pl = ((int)(Lx * 4u) - (int)bx0) + ((int)(Ly * 4u) - (int)by0) * 1024 + ((int)(Lz * 4u) - (int)bz0) * 1048576;
                        if (Lx < o2x && Ly < o2y && Lz < o2z)
                            w2 = occ2[((size_t)Lz * o2y + Ly) * o2x + Lx] & spread_x(axis_mask(Lx, bx0, bx1)) & spread_y(axis_mask(Ly, by0, by1)) &
                                 spread_z(axis_mask(Lz, bz0, bz1));
                    }
                    const uint32_t c2 = (uint32_t)__popcll(w2), i2 = wave_incl_scan(c2, lane);
                    list_bits(w2, pl, c2, i2 + T, chunk_lo, CH, s_bl);
                    T += (uint32_t)__builtin_amdgcn_readlane((int)i2, 63);
                }
                __syncthreads();
                n_all = T;
            }
            if (n_all <= chunk_lo) break;
            const uint32_t n_in = min(CH, n_all - chunk_lo);
            for (uint32_t b0 = 0; b0 < n_in && !over; b0 += 64u)
            {
                const bool ok = b0 + (uint32_t)lane < n_in;
                uint32_t   Bx, By, Bz;
                if (two_stage)
                {
                    const uint32_t pk = ok ? s_bl[b0 + (uint32_t)lane] : 0u;
                    Bx = bx0 + (pk & 1023u), By = by0 + ((pk >> 10) & 1023u), Bz = bz0 + (pk >> 20);
                }
                else
                {
                    // (exact for nb <= 32 768: (c + 0.5) / n is at least 0.5 / n away from an integer; an integer division costs ~25 instructions)
                    const uint32_t id  = chunk_lo + b0 + (uint32_t)lane;
                    const uint32_t row = (uint32_t)(((float)id + 0.5f) * inv_nbx), ix = id - row * nbx;
                    const uint32_t iz  = (uint32_t)(((float)row + 0.5f) * inv_nby), iy = row - iz * nby;
                    Bx = bx0 + ix, By = by0 + iy, Bz = bz0 + iz;
                }
                unsigned long long bm;
                int                pb;
                brick_word(ok, Bx, By, Bz, bm, pb);
                serve(bm, pb, vcap);
            }
            if (two_stage) __syncthreads();  // the brick list is rewritten by the next chunk
            if (n_all <= chunk_lo + CH) break;
        }

        // ---- merge the two slices of each query slot ----------------------------------------
        {
            const float    od = __shfl_xor(best_d2, 32, 64);
            const uint32_t oi = __shfl_xor(best_idx, 32, 64);
            const uint32_t os = __shfl_xor(best_spos, 32, 64);
            if (od < best_d2 || (od == best_d2 && oi < best_idx)) best_d2 = od, best_idx = oi, best_spos = os;
        }
        if (track)
        {
            const int o1_ = __shfl_xor(t1, 32, 64), o2_ = __shfl_xor(t2, 32, 64);
            t2 = min(max(t1, o1_), min(t2, o2_)), t1 = min(t1, o1_);
        }
        bool too_wide = false;
        if (grp && !over)  // (a pass cut short has not covered its balls: nobody concludes)
        {
            const bool fin = SOL != 0 || is_final(r, rmax, best_d2, g.slack);
            if (track && fin)
            {
                // every staged point but the nearest has S >= t2, hence d2 >= t2 - mtol; every point NOT staged lies outside
                // the ball this pass covered
                const float cover = r * (1.0f - 1.0f / 1024.0f) - g.slack;
                // (the prefilter's values are d2 - |q'|^2 + beta: back to d2, with the rounding of that offset inside the margin)
                const float s2    = t2 < 0 ? 0.f : (t2 == 0x7FFFFFFF ? INFINITY : fmaxf(__int_as_float(t2) - loff - 1.5f * mtol, 0.f));
                lbq = fmaxf(fminf(sqrtf(s2) * 0.99999f - g.slack, cover), 0.f);
            }
            if (fin) done = true;
            else
            {
                r        = next_radius(r, rmax, best_d2, best_idx != NONE_U32, g.slack);
                too_wide = r > a.r_defer && best_idx == NONE_U32;
            }
        }
        if (!done && st_cand > cand_cap) too_wide = true;
        const unsigned long long wmask = __ballot(too_wide);
        if (wmask)
        {
            if (SOL == 0)
                st_defer += defer_lanes<Q>(a, seg, too_wide, wmask, lane, slice, qi, st_cand > cand_cap ? -r : r, best_d2, best_idx, best_spos, qx, qy, qz);
            if (too_wide) done = true, deferred = true;
        }
    }

    // ---- a query with NOTHING within reach (an outlier of the local layer): is the cube of half-edge 2 r_max (else 1.5 r_max)
    //      around it empty?  Then that is its bound and the warm start skips it until it has moved by the difference
    //      (nn_single_kernel has the reasoning; isolated queries used to be handed to that kernel, with the selection they stay).
    //      The whole wave serves one such query at a time: a handful of coarse occupancy bits each.
    float lb2_room = -1.f;
    if (SOL == 0 && a.empty_room)
    {
        unsigned long long em = __ballot(valid && slice == 0 && !deferred && active && !pro_done && best_idx == NONE_U32);
        while (em)
        {
            const int l = __ffsll((long long)em) - 1;
            em &= em - 1ull;
            const float b = empty_room_bound(g, lane, readlane_f(qx, l), readlane_f(qy, l), readlane_f(qz, l), readlane_f(rmax, l));
            if (lane == l) lb2_room = b;
        }
    }
    // ---- output (Morton order of the local layer) + claim of the global point -----------------
    if (SOL == 0)
    {
        emit_wave(a, s_claim, lane, valid && slice == 0 && !deferred, qi, orig, active, thr, best_d2, best_idx, best_spos,
                  (pro_done && lb2_out >= 0.f) ? lb2_out : (lb2_room >= 0.f ? lb2_room : fminf(best_d2, thr)), st_cand);
        if (a.lb2nd && valid && slice == 0 && !deferred) a.lb2nd[qi] = pro_done ? lb2nd_out : (track ? lbq : 0.f);
    }
    else if ((best_d2 == -1.f || st_cand == 0xFFFFFFFFu) && lane == 0) a.rec[0].x = best_idx + st_listed + st_needed;  // (never true: keeps the timing-only build's work alive)

    // (profiling level 4: the spare high bits carry what the tile did -- staged candidates / 32, passes, rounds of 64 bricks)
    if (a.timeline && lane == 0)
        a.timeline[2 * (size_t)tile]     = (tl0 & 0xFFFFFFFFFFull) | ((unsigned long long)min(st_cand >> 5, 0xFFFFFFu) << 40),
        a.timeline[2 * (size_t)tile + 1] = (wall_clock64() & 0xFFFFFFFFFFull) | ((unsigned long long)min(st_pass, 255u) << 40) |
                                           ((unsigned long long)min(st_cells >> 6, 0xFFFFu) << 48);
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
        int b = 63 - __clzll((long long)(dt | 1ull));
        if (b > 23) b = 23;
        atomicAdd(&a.counters[16 + b], 1ull);
        atomicAdd(&a.counters[49], (unsigned long long)st_listed);
        atomicAdd(&a.counters[50], (unsigned long long)st_needed);
    }
}

}  // namespace mp2p
