// matcher_adaptive.hip -- Matcher_Adaptive::implMatchOneLayer (mp2p_icp/src/Matcher_Adaptive.cpp:59-314;
// demos/icp-settings-kitti.yaml from iteration 6 on).
//
// Reference, per local point: up to nn = (enableDetectPlanes ? planeSearchPoints :
// maxPt2PtCorrespondences) nearest global points within absoluteMaxSearchDistance, ascending; the
// first MAX_CORRS_PER_LOCAL = 10 are kept (Matcher_Adaptive.h:83).  The squared distances of
// everybody's first two go into a 50-bin mrpt::math::CHistogram between their min and max; the
// upper confidence limit of that histogram (or minimumCorrDist^2 if larger) is the pairing
// threshold.  Second loop: a local point whose kept neighbours are planar and whose (UNtransformed,
// :113,245) coordinates lie within planeMinimumDistance of that plane yields a pt2pl pairing;
// otherwise its first maxPt2PtCorrespondences neighbours below the threshold yield pt2pt pairings
// until one is farther than firstToSecondDistanceMax^2 times the first.  Global marks are read,
// never written; local marks as :260,289-293.
//
// Here: three steps on the device around one small host decision.
//   1. adaptive_knn_kernel: the k-NN tile search of nn_pt2pl.hip, lists to ctx->nn_spos / nn_d2,
//      min / max of the first two by atomics on the fp32 bit patterns (d2 >= 0);
//   2. adaptive_hist_kernel: the 50 bins (LDS, then global);  the host turns them into the
//      threshold (mp2p_hip_adaptive_ci_high, api.hip) -- CHistogram and
//      confidenceIntervalsFromHistogram are MRPT's (un-vendored): restated, PARITY UNPINNED; a
//      caller that links MRPT passes its own threshold to the select step instead;
//   3. adaptive_select_kernel (one thread per local point): plane test, pair selection; then the
//      ordered compactions of nn_pt2pl.hip (planes) and pairs.hip (point pairs, nn slots per point).
// maxLocalPointsPerLayer is refused: the reference indexes matchesPerLocal_ (sized by the subset)
// with the ORIGINAL local index (:103,123) and throws std::out_of_range for it.
#include "device_utils.hpp"

namespace mp2p
{
constexpr int AD_BINS = MP2P_HIP_ADAPTIVE_BINS;  // Matcher_Adaptive.cpp:189
constexpr int AD_KEEP = 10;                      // MAX_CORRS_PER_LOCAL

struct AdSearchArgs
{
    GridView             g;
    const float4*        lpts;
    uint32_t             n_l;
    PoseRt               pose;
    float                absMaxSq, rad, r0;
    uint32_t             knn;
    const unsigned char* local_taken;
    uint32_t*            out_spos;  // [n_l][knn] in the order of lpts
    float*               out_d2;
    float*               tile_bbox;
    uint32_t*            minmax;  // fp32 bits: [0] min (init ~0), [1] max (init 0)
};

template <int K, bool STRICT>
__global__ __launch_bounds__(64) void adaptive_knn_kernel(const AdSearchArgs a)
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
    transform_tile<PL_Q>(a.pose, a.lpts, a.n_l, nullptr, a.tile_bbox, lane, valid, visited, orig, vrank, qx, qy, qz);
    const float fin    = fadd(fadd(qx, qy), qz);
    bool        active = visited && (fin - fin == 0.0f);
    if (active && a.local_taken && a.local_taken[orig]) active = false;  // :126-132

    float    kd2[K];
    uint32_t kidx[K], kspos[K];
    // :155-158 nn_radius_search keeps d2 < r^2 (STRICT); :138-152 + :165 nn_single_search, d2 <= r^2
    knn_search<K, STRICT, PL_Q>(g, lane, qx, qy, qz, active, a.absMaxSq, a.rad, a.r0, a.knn, PL_GROUP_FACTOR,
                                PL_CELL_BUDGET, nullptr, s_hit, s_cand, s_spos, s_cstart, s_coff, kd2, kidx, kspos);
    float mn = INFINITY, mx = -1.0f;
    if (valid && lane < PL_Q)
    {
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            if (k >= (int)a.knn) continue;
            const bool   have = active && kidx[k] != NONE_U32;
            const size_t slot = (size_t)(blockIdx.x * (uint32_t)PL_Q + (uint32_t)lane) * a.knn + k;
            a.out_spos[slot]  = have ? kspos[k] : NONE_U32;
            a.out_d2[slot]    = kd2[k];
            if (have && k <= 1) mn = fminf(mn, kd2[k]), mx = fmaxf(mx, kd2[k]);  // :167-181
        }
    }
    mn = wave_min(mn), mx = wave_max(mx);
    if (lane == 0 && mx >= 0.0f)
    {
        atomicMin(&a.minmax[0], __float_as_uint(mn));
        atomicMax(&a.minmax[1], __float_as_uint(mx));
    }
}

// CHistogram(min, max, 50)::add of the first two squared distances of every local point (:189-193):
// bin = (size_t)((nBins-1)/(max-min) * (x - min)); bins[AD_BINS] = number of samples
__global__ __launch_bounds__(256) void adaptive_hist_kernel(const uint32_t* __restrict__ spos,
                                                            const float* __restrict__ d2, uint32_t n_l,
                                                            uint32_t knn, const uint32_t* __restrict__ minmax,
                                                            unsigned long long* __restrict__ bins)
{
    __shared__ uint32_t s_b[AD_BINS + 1];
    for (int i = threadIdx.x; i <= AD_BINS; i += blockDim.x) s_b[i] = 0;
    __syncthreads();
    const double mn = (double)__uint_as_float(minmax[0]), mx = (double)__uint_as_float(minmax[1]);
    const bool   flat       = !(mx > mn);  // one value only: everything in bin 0 (MRPT: 0 * inf)
    const double binSizeInv = (double)(AD_BINS - 1) / (mx - mn);
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_l; q += (size_t)gridDim.x * blockDim.x)
        for (uint32_t k = 0; k < knn && k < 2; k++)
        {
            if (spos[q * knn + k] == NONE_U32) continue;
            const double x = (double)d2[q * knn + k];
            size_t       b = flat ? 0 : (size_t)(binSizeInv * (x - mn));
            if (b >= (size_t)AD_BINS) b = AD_BINS - 1;
            atomicAdd(&s_b[b], 1u);
            atomicAdd(&s_b[AD_BINS], 1u);
        }
    __syncthreads();
    for (int i = threadIdx.x; i <= AD_BINS; i += blockDim.x)
        if (s_b[i]) atomicAdd(&bins[i], (unsigned long long)s_b[i]);
}

struct AdSelectArgs
{
    const float4*        gpts;
    const float4*        lpts;
    uint32_t             n_l, knn;
    uint32_t*            spos;  // in: the lists; out: NONE wherever no point pair is produced
    const float*         d2;
    const float *        lx, *ly, *lz;  // original-order local coordinates
    int                  detect;
    uint32_t             minFound, maxPt2Pt;
    double               planeMinDist, eigThr, maxCorrDistSqr;
    float                maxSqr1to2;
    const unsigned char* global_taken;
    unsigned char*       out_flag;  // [n_l] by original local index
    double*              out_rec;   // [n_l][7]
};

__global__ __launch_bounds__(256) void adaptive_select_kernel(const AdSelectArgs a)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n_l) return;
    const uint32_t orig = __float_as_uint(a.lpts[q].w);
    uint32_t*      sp   = a.spos + (size_t)q * a.knn;
    const float*   dd   = a.d2 + (size_t)q * a.knn;
    int            m    = 0;  // kept neighbours: ascending, NONE only at the tail
    for (uint32_t k = 0; k < a.knn && k < (uint32_t)AD_KEEP; k++)
    {
        if (sp[k] == NONE_U32) break;
        m++;
    }
    unsigned char flag = 0;
    if (a.detect && m >= (int)a.minFound && m >= 1)  // :221
    {
        float px[AD_KEEP], py[AD_KEEP], pz[AD_KEEP];
#pragma unroll
        for (int j = 0; j < AD_KEEP; j++)
        {
            px[j] = py[j] = pz[j] = 0.f;
            if (j < m)
            {
                const float4 p = a.gpts[sp[j]];
                px[j] = p.x, py[j] = p.y, pz[j] = p.z;
            }
        }
        float  mx, my, mz;
        double n[3];
        if (plane_of_points<AD_KEEP>(px, py, pz, m, a.eigThr, n, mx, my, mz))  // :233-238
        {
            const double c0 = (double)mx, c1 = (double)my, c2 = (double)mz;
            const double d  = -(n[0] * c0 + n[1] * c1 + n[2] * c2);
            // :245-246 the distance of mspl[0].local: the local point as stored, not transformed
            const double dist =
                fabs(n[0] * (double)a.lx[orig] + n[1] * (double)a.ly[orig] + n[2] * (double)a.lz[orig] + d);
            if (dist < a.planeMinDist)  // :248
            {
                flag      = 1;
                double* o = a.out_rec + (size_t)orig * 7;
                o[0] = n[0], o[1] = n[1], o[2] = n[2], o[3] = d;
                o[4] = c0, o[5] = c1, o[6] = c2;
            }
        }
    }
    a.out_flag[orig] = flag;
    const float d0   = dd[0];
    bool        stop = flag != 0;  // :263 a plane pairing ends this local point
    for (uint32_t k = 0; k < a.knn; k++)
    {
        bool keep = false;
        if (!stop && (int)k < m && k < a.maxPt2Pt)  // :268
        {
            const uint32_t gi = __float_as_uint(a.gpts[sp[k]].w);
            if (a.global_taken && a.global_taken[gi]) {}                // :273-275
            else if ((double)dd[k] >= a.maxCorrDistSqr) {}              // :278
            else if (k != 0 && dd[k] > fmul(d0, a.maxSqr1to2)) stop = true;  // :280-284
            else keep = true;
        }
        if (!keep) sp[k] = NONE_U32;
    }
}

static uint32_t adaptive_nn(const mp2p_hip_adaptive_params* prm)
{
    return prm->enableDetectPlanes ? prm->planeSearchPoints : prm->maxPt2PtCorrespondences;  // :120
}

template <int K, bool STRICT>
static void launch_ad_k(const AdSearchArgs& a, uint32_t n_tiles, hipStream_t st)
{
    hipLaunchKernelGGL((adaptive_knn_kernel<K, STRICT>), dim3(n_tiles), dim3(64), 0, st, a);
}

// steps 1 + 2; synchronises the stream twice (min/max, then the bins)
int launch_adaptive_search(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                           const double pose[12], const mp2p_hip_adaptive_params* prm, mp2p_hip_mstate* ms,
                           mp2p_hip_adaptive_hist* hist)
{
    const size_t   n_l     = cloud->n;
    const uint32_t K       = adaptive_nn(prm);
    const uint32_t n_tiles = (uint32_t)((n_l + PL_Q - 1) / PL_Q);
    // Matcher_Adaptive.cpp:78-81 returns BEFORE any search when the two boxes (inflated by the epsilon) do not meet.  Every transformed
    // local point lies within cloud->radius of the pose's translation: when that ball (a superset of the layer's box, also of a
    // visited subset's) clears the map's box along some axis, the boxes are certainly apart and nothing is launched (ADVICE r5;
    // disjoint layers used to pay a full search first).  The exact test on the reduced box, below, stays for everything else.
    ctx->ad_apart = false;
    if (cloud->radius == cloud->radius && cloud->radius < INFINITY)
    {
        // |R p| <= |p| for a rotation (checked: R^T R = I to 1e-6), else <= ||R||_F |p|
        double dev = 0.0, fro = 0.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
            {
                double dot = 0.0;
                for (int k = 0; k < 3; k++) dot += pose[3 * k + i] * pose[3 * k + j];
                dev = std::max(dev, std::fabs(dot - (i == j ? 1.0 : 0.0)));
                fro += pose[3 * i + j] * pose[3 * i + j];
            }
        const double gain = dev <= 1e-6 ? 1.00001 : std::sqrt(fro) * 1.00001;
        const double eps = prm->bounding_box_intersection_check_epsilon, R = (double)cloud->radius * gain + 1e-6 * (1.0 + (double)cloud->radius);
        for (int d = 0; d < 3; d++)
            if (pose[9 + d] - R - eps > (double)map->view.bbmax[d] || pose[9 + d] + R + eps < (double)map->view.bbmin[d])
            {
                memset(hist, 0, sizeof(*hist));
                ctx->ad_knn = K, ctx->ad_cloud = cloud, ctx->ad_map = map, ctx->ad_apart = true;
                return MP2P_HIP_OK;
            }
    }
    MP2P_TRY_HIP(ctx, ctx->tile_bbox.ensure((size_t)n_tiles * 6));
    MP2P_TRY_HIP(ctx, ctx->local_bbox.ensure(6));
    MP2P_TRY_HIP(ctx, ctx->nn_spos.ensure(n_l * K));
    MP2P_TRY_HIP(ctx, ctx->nn_d2.ensure(n_l * K));
    MP2P_TRY_HIP(ctx, ctx->ad_hist.ensure(AD_BINS + 2));  // bins, count, {min,max} words
    MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->ad_hist.p, 0, (AD_BINS + 2) * sizeof(unsigned long long), ctx->stream));
    uint32_t* minmax = reinterpret_cast<uint32_t*>(ctx->ad_hist.p + AD_BINS + 1);
    MP2P_TRY_HIP(ctx, hipMemsetAsync(minmax, 0xFF, sizeof(uint32_t), ctx->stream));

    AdSearchArgs a;
    memset(&a, 0, sizeof(a));
    a.g = map->view, a.lpts = cloud->sorted.p, a.n_l = (uint32_t)n_l;
    for (int i = 0; i < 9; i++) a.pose.r[i] = pose[i];
    for (int i = 0; i < 3; i++) a.pose.t[i] = pose[9 + i];
    a.absMaxSq = (float)(prm->absoluteMaxSearchDistance * prm->absoluteMaxSearchDistance);  // :88
    a.rad      = (float)prm->absoluteMaxSearchDistance * 1.002f + map->view.slack;
    const float cell0 = map->view.hf * (float)(1u << map->view.shift0);
    a.r0  = cell0 * 2.0f;
    a.knn = K;
    a.local_taken = (ms && !prm->allowMatchAlreadyMatchedPoints) ? ms->local_taken.p : nullptr;
    a.out_spos = ctx->nn_spos.p, a.out_d2 = ctx->nn_d2.p, a.tile_bbox = ctx->tile_bbox.p;
    a.minmax = minmax;
    if (K == 1) launch_ad_k<5, false>(a, n_tiles, ctx->stream);
    else if (K <= 5) launch_ad_k<5, true>(a, n_tiles, ctx->stream);
    else if (K <= 8) launch_ad_k<8, true>(a, n_tiles, ctx->stream);
    else if (K <= 12) launch_ad_k<12, true>(a, n_tiles, ctx->stream);
    else launch_ad_k<16, true>(a, n_tiles, ctx->stream);
    const int rc = launch_bbox_reduce(ctx, n_tiles);
    if (rc) return rc;

    uint32_t h_mm[2] = {0, 0};
    float    h_bb[6];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_mm, minmax, sizeof(h_mm), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_bb, ctx->local_bbox.p, sizeof(h_bb), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    memset(hist, 0, sizeof(*hist));
    ctx->ad_knn = K, ctx->ad_cloud = cloud, ctx->ad_map = map;
    if (h_mm[0] == 0xFFFFFFFFu) return MP2P_HIP_OK;  // nobody found a neighbour: hist->valid = 0
    // Matcher_Adaptive.cpp:78-81 returns before any search when the two boxes, inflated by the epsilon only, do not meet: no
    // histogram exists then, although queries may well have neighbours within absoluteMaxSearchDistance (the selection step
    // applies the same test and leaves no pairing)
    {
        const float eps = (float)prm->bounding_box_intersection_check_epsilon;
        for (int d = 0; d < 3; d++)
            if (h_bb[d] - eps > map->view.bbmax[d] || h_bb[3 + d] + eps < map->view.bbmin[d]) return MP2P_HIP_OK;
    }
    hist->valid = 1;
    memcpy(&hist->minSqr, &h_mm[0], 4), memcpy(&hist->maxSqr, &h_mm[1], 4);
    const uint32_t nb = (uint32_t)std::min<size_t>((n_l + 255) / 256, 2048);
    hipLaunchKernelGGL(adaptive_hist_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->nn_spos.p, ctx->nn_d2.p,
                       (uint32_t)n_l, K, minmax, ctx->ad_hist.p);
    unsigned long long h_b[AD_BINS + 1];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_b, ctx->ad_hist.p, sizeof(h_b), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    for (int i = 0; i < AD_BINS; i++) hist->bins[i] = h_b[i];
    hist->count = h_b[AD_BINS];
    return MP2P_HIP_OK;
}

// step 3 for the lists of the last launch_adaptive_search on this context
int launch_adaptive_select(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                           const mp2p_hip_adaptive_params* prm, double ci_high, mp2p_hip_mstate* ms,
                           mp2p_hip_pairs* out)
{
    const size_t   n_l = cloud->n;
    const uint32_t K   = adaptive_nn(prm);
    MP2P_TRY_HIP(ctx, ctx->pl_slots.ensure(n_l * (7 * sizeof(double) + 1) + 64));
    double*        rec  = reinterpret_cast<double*>(ctx->pl_slots.p);
    unsigned char* flag = ctx->pl_slots.p + n_l * 7 * sizeof(double);

    AdSelectArgs s;
    memset(&s, 0, sizeof(s));
    s.gpts = map->pts.p, s.lpts = cloud->sorted.p, s.n_l = (uint32_t)n_l, s.knn = K;
    s.spos = ctx->nn_spos.p, s.d2 = ctx->nn_d2.p;
    s.lx = cloud->x.p, s.ly = cloud->y.p, s.lz = cloud->z.p;
    s.detect   = prm->enableDetectPlanes ? 1 : 0;
    s.minFound = prm->planeMinimumFoundPoints, s.maxPt2Pt = prm->maxPt2PtCorrespondences;
    s.planeMinDist = prm->planeMinimumDistance, s.eigThr = prm->planeEigenThreshold;
    s.maxCorrDistSqr = std::max(prm->minimumCorrDist * prm->minimumCorrDist, ci_high);               // :212
    s.maxSqr1to2     = (float)(prm->firstToSecondDistanceMax * prm->firstToSecondDistanceMax);        // :214
    s.global_taken   = (ms && !prm->allowMatchAlreadyMatchedGlobalPoints) ? ms->global_taken.p : nullptr;
    s.out_flag = flag, s.out_rec = rec;
    const uint32_t nb = (uint32_t)((n_l + 255) / 256);
    hipLaunchKernelGGL(adaptive_select_kernel, dim3(nb), dim3(256), 0, ctx->stream, s);

    const float margin = (float)prm->bounding_box_intersection_check_epsilon;  // :78-81: epsilon only
    {  // planes, in the order of the local points (:250-260)
        const uint32_t n_blocks = (uint32_t)((n_l + PC_TILE - 1) / PC_TILE);
        MP2P_TRY_HIP(ctx, ctx->block_counts.ensure(n_blocks ? n_blocks : 1));
        PlCompactArgs c;
        memset(&c, 0, sizeof(c));
        c.flag = flag, c.rec = rec, c.n_l = (uint32_t)n_l, c.local_bbox = ctx->local_bbox.p;
        for (int d = 0; d < 3; d++) c.gbb[d] = map->view.bbmin[d], c.gbb[3 + d] = map->view.bbmax[d];
        c.margin = margin;
        c.lx = cloud->x.p, c.ly = cloud->y.p, c.lz = cloud->z.p;
        c.block_counts = ctx->block_counts.p, c.counts = out->counts.p, c.cap = out->cap_pt2pl;
        c.o_lidx = out->pl_lidx.p, c.o_coef = out->pl_coef.p, c.o_cen = out->pl_cen.p;
        c.o_lx = out->pl_lx.p, c.o_ly = out->pl_ly.p, c.o_lz = out->pl_lz.p;
        c.ms_local = ms ? ms->local_taken.p : nullptr;
        hipLaunchKernelGGL(pl_count_kernel, dim3(n_blocks), dim3(PC_THREADS), 0, ctx->stream, c);
        hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->block_counts.p,
                           n_blocks, out->counts.p, c.cap, 0ull, 1);
        hipLaunchKernelGGL(pl_write_kernel, dim3(n_blocks), dim3(PC_THREADS), 0, ctx->stream, c);
    }
    // point pairs: slot (local point, k); the local mark only when global re-use is forbidden
    // (:289-293), global marks never
    return launch_compact_slots(ctx, map, cloud, nullptr, n_l * K, nullptr, K, /*use_claims=*/false,
                                /*always_mark=*/!prm->allowMatchAlreadyMatchedGlobalPoints, 0ull, margin,
                                0ull /* potential_pairings: added by the caller (:69) */, ms, out,
                                /*mark_global=*/false);
}

}  // namespace mp2p
