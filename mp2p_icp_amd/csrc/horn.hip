// horn.hip -- "next #1": optimal_tf_horn (optimal_tf_horn.cpp:77-252) with WeightParameters
// (WeightParameters.h:34-72): weighted centroids without the current outliers
// (eval_centroids_robust, Pairings.cpp:68-110), then visit_correspondences
// (visit_correspondences.h:38-212) as one streaming fp64 reduction over the device-resident point
// pairs and the uploaded plane-to-plane pairs: attitude weights, point_weights blocks, the
// "(almost) on the centroid" skip, the scale outlier detector (flags, second pass with new
// centroids), the robust kernel against currentEstimateForRobust; 3x3 S and the weight sum go to
// the host for the 4x4 symmetric eigenproblem (:158) and the translation (:238-247).
// Also pt2ln_pl_to_pt2pt (pt2ln_pl_to_pt2pt.cpp:47-113), the conversion Solver_Horn applies when
// the pairings hold point-to-plane / point-to-line entries (Solver_Horn.cpp:51-55).
// paired_ln2ln has no container on this path.
#include <hipcub/hipcub.hpp>

#include "device_utils.hpp"

namespace mp2p
{
constexpr int HORN_BLOCKS = 512;

struct HornKernelPrm
{
    int                use_scale;
    double             scale_thr;
    double             waPoints, waPlanes;  // (computed by horn_cov_kernel from the list's own counts: no host round trip before the pass)
    double             w_pt2pt, w_ln2ln, w_pl2pl;
    GnKernelPrm        rk;  // robust kernel (kernel, c, c2)
    double             est[12];
    uint32_t           n_blocks;  // point_weights blocks (0: one block of weight 1)
    unsigned long long blk_count[MP2P_HIP_MAX_WEIGHT_BLOCKS];
    double             blk_w[MP2P_HIP_MAX_WEIGHT_BLOCKS];
};

// sums of the non-flagged pairs: global xyz, local xyz, their number
__global__ __launch_bounds__(GN_THREADS) void horn_centroid_kernel(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned long long* __restrict__ counts, const unsigned char* __restrict__ outlier,
    double* __restrict__ partials)
{
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long n = counts[0];
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n;
         i += (unsigned long long)HORN_BLOCKS * GN_THREADS)
    {
        if (outlier[i]) continue;  // Pairings.cpp:91-95
        acc[0] += (double)gx[i], acc[1] += (double)gy[i], acc[2] += (double)gz[i];
        acc[3] += (double)lx[i], acc[4] += (double)ly[i], acc[5] += (double)lz[i];
        acc[6] += 1.0;
    }
    block_reduce_store<7>(acc, partials + (size_t)blockIdx.x * 16);
}

// The weight-block cursor of visit_correspondences.h:113-119 only moves on VISITED pairs (flagged
// ones are skipped before it), by one block per pair: bounds[b] = first pair of block b as that
// loop would see it.  One wave; bounds[MP2P_HIP_MAX_WEIGHT_BLOCKS] = 1 when pairs remain after the last block (the
// reference then reads past point_weights.end()).
__global__ __launch_bounds__(64) void horn_block_bounds_kernel(const unsigned char* __restrict__ outlier,
                                                               const unsigned long long* __restrict__ counts,
                                                               HornKernelPrm prm,
                                                               unsigned long long* __restrict__ bounds)
{
    const int                lane = threadIdx.x;
    const unsigned long long n    = counts[0];
    unsigned long long       start = 0;
    uint32_t                 b     = 0;
    int                      bad   = 0;
    if (lane == 0) bounds[0] = 0;
    while (true)
    {
        unsigned long long s = start + prm.blk_count[b];
        if (b > 0 && s <= start) s = start + 1;  // the pair that opened block b belongs to it
        // first non-flagged pair >= s
        unsigned long long found = n;
        for (unsigned long long base = s; base < n; base += 64)
        {
            const unsigned long long i  = base + lane;
            const bool               ok = i < n && !outlier[i];
            const unsigned long long m  = __ballot(ok);
            if (m)
            {
                found = base + (unsigned long long)(__ffsll((long long)m) - 1);
                break;
            }
        }
        if (found >= n) break;
        if (b + 1 >= prm.n_blocks)
        {
            bad = 1;
            break;
        }
        b++, start = found;
        if (lane == 0) bounds[b] = found;
    }
    if (lane == 0)
    {
        for (uint32_t k = b + 1; k < MP2P_HIP_MAX_WEIGHT_BLOCKS; k++) bounds[k] = n;
        bounds[MP2P_HIP_MAX_WEIGHT_BLOCKS] = (unsigned long long)bad;
    }
}

// acc: S (9), sum of weights, pairs with a weight <= 0, pairs flagged after this pass
__global__ __launch_bounds__(GN_THREADS) void horn_cov_kernel(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const mp2p_hip_pair_pl2pl* __restrict__ pp, const unsigned long long* __restrict__ counts,
    const double* __restrict__ cent /*cg(3) cl(3)*/, const unsigned long long* __restrict__ bounds,
    const HornKernelPrm prm, unsigned char* __restrict__ outlier, double* __restrict__ partials)
{
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = 0;
    const unsigned long long nPt = counts[0], nPl = counts[6];
    // visit_correspondences.h:76-86 (no line pairs on this path); the host's expression of rounds 1-5, evaluated here
    const double kk = 1.0 / (prm.w_pt2pt * (double)nPt + prm.w_ln2ln * 0.0 + prm.w_pl2pl * (double)nPl);
    const double waPoints = prm.w_pt2pt * kk, waPlanes = prm.w_pl2pl * kk;
    const double cg0 = cent[0], cg1 = cent[1], cg2 = cent[2], cl0 = cent[3], cl1 = cent[4], cl2 = cent[5];
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < nPt + nPl;
         i += (unsigned long long)HORN_BLOCKS * GN_THREADS)
    {
        double b0, b1, b2, r0, r1, r2, wi;
        if (i < nPt)
        {
            if (outlier[i])
            {  // visit_correspondences.h:97-103: stays an outlier
                acc[11] += 1.0;
                continue;
            }
            wi = waPoints;
            if (prm.n_blocks)
            {
                uint32_t b = 0;
                while (b + 1 < prm.n_blocks && i >= bounds[b + 1]) b++;
                wi *= prm.blk_w[b];  // :120
            }
            b0 = (double)gx[i] - cg0, b1 = (double)gy[i] - cg1, b2 = (double)gz[i] - cg2;
            r0 = (double)lx[i] - cl0, r1 = (double)ly[i] - cl1, r2 = (double)lz[i] - cl2;
            const double bn = sqrt(b0 * b0 + b1 * b1 + b2 * b2), rn = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
            if (bn < 1e-4 || rn < 1e-4) continue;  // :135-140
            if (prm.use_scale)                     // :153-164
            {
                const double mism = fmax(bn, rn) / fmin(bn, rn);
                if (mism > prm.scale_thr)
                {
                    outlier[i] = 1;
                    acc[11] += 1.0;
                    continue;
                }
            }
        }
        else
        {  // :181-192 plane normals as stored (getNormalVector)
            const mp2p_hip_pair_pl2pl& q = pp[i - nPt];
            wi = waPlanes;
            b0 = q.pl_global[0], b1 = q.pl_global[1], b2 = q.pl_global[2];
            r0 = q.pl_local[0], r1 = q.pl_local[1], r2 = q.pl_local[2];
        }
        if (prm.rk.kernel != MP2P_HIP_KERNEL_NONE)  // :194-205
        {
            const double* T = prm.est;
            const double  x = T[0] * r0 + T[1] * r1 + T[2] * r2 + T[9];
            const double  y = T[3] * r0 + T[4] * r1 + T[5] * r2 + T[10];
            const double  z = T[6] * r0 + T[7] * r1 + T[8] * r2 + T[11];
            const double  e2 = (x - b0) * (x - b0) + (y - b1) * (y - b1) + (z - b2) * (z - b2);
            wi *= robust_w(prm.rk, e2);
        }
        if (!(wi > 0.0))
        {  // :207 ASSERT_(wi > .0)
            acc[10] += 1.0;
            continue;
        }
        acc[0] += wi * r0 * b0, acc[1] += wi * r0 * b1, acc[2] += wi * r0 * b2;
        acc[3] += wi * r1 * b0, acc[4] += wi * r1 * b1, acc[5] += wi * r1 * b2;
        acc[6] += wi * r2 * b0, acc[7] += wi * r2 * b1, acc[8] += wi * r2 * b2;
        acc[9] += wi;
    }
    block_reduce_store<12>(acc, partials + (size_t)blockIdx.x * 16);
}

// fixed-order sum of [HORN_BLOCKS][16] partials; centroid mode: the first 6 scaled by 1/count
__global__ __launch_bounds__(256) void horn_sum_kernel(const double* __restrict__ partials, int nq,
                                                       int centroid_mode, double* __restrict__ out,
                                                       const unsigned long long* __restrict__ counts = nullptr)
{
    // (counts: the list's eight counters ride behind the sums -- out[12 .. 19] as raw bits -- so that ONE copy brings both back)
    if (counts && threadIdx.x >= 248) reinterpret_cast<unsigned long long*>(out + 12)[threadIdx.x - 248] = counts[threadIdx.x - 248];
    // (round 6: 16 parts x 16 quantities, every thread's HORN_BLOCKS / 16 loads independent, the parts combined as a fixed tree --
    //  deterministic run to run like the Gauss-Newton sums; one thread per quantity walking all 512 rows was 12.6 us per launch, two
    //  launches per solve = 9 % of a C2 step)
    static_assert(HORN_BLOCKS % 16 == 0, "16 parts");
    __shared__ double s[16][16];
    __shared__ double s_t[16];
    const int         q = threadIdx.x & 15, part = threadIdx.x >> 4;
    double            v[HORN_BLOCKS / 16];
#pragma unroll
    for (int k = 0; k < HORN_BLOCKS / 16; k++) v[k] = q < nq ? partials[(size_t)(part + 16 * k) * 16 + q] : 0.0;
    double t = 0;
#pragma unroll
    for (int k = 0; k < HORN_BLOCKS / 16; k++) t += v[k];
    s[part][q] = t;
    __syncthreads();
    if (threadIdx.x < 16)
    {
        const int i = threadIdx.x;
        double    r[8];
        for (int k = 0; k < 8; k++) r[k] = s[2 * k][i] + s[2 * k + 1][i];
        s_t[i] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
    __syncthreads();
    if (threadIdx.x >= 16) return;
    t = s_t[q];
    if (q >= nq) return;
    if (centroid_mode && q < 6)
    {
        const double cnt = s_t[6];
        t *= (cnt > 0.0 ? 1.0 / cnt : 0.0);  // wcPoints (Pairings.cpp:80)
    }
    out[q] = t;
}

static void host_jacobi4(const double* Ain, double* eval, double* V)
{
    const int n = 4;
    double    A[16], Q[16];
    for (int i = 0; i < 16; i++) A[i] = Ain[i], Q[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; sweep++)
    {
        double off = 0;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++)
            {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double th = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t  = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++)
                {
                    const double x = A[k * n + p], y = A[k * n + q];
                    A[k * n + p] = c * x - s * y, A[k * n + q] = s * x + c * y;
                }
                for (int k = 0; k < n; k++)
                {
                    const double x = A[p * n + k], y = A[q * n + k];
                    A[p * n + k] = c * x - s * y, A[q * n + k] = s * x + c * y;
                }
                for (int k = 0; k < n; k++)
                {
                    const double x = Q[k * n + p], y = Q[k * n + q];
                    Q[k * n + p] = c * x - s * y, Q[k * n + q] = s * x + c * y;
                }
            }
    }
    for (int k = 0; k < n; k++)
    {
        eval[k] = A[k * n + k];
        for (int i = 0; i < n; i++) V[k * n + i] = Q[i * n + k];
    }
}

// one pass of se3_l2_internal (optimal_tf_horn.cpp:77-196) for the current flags; h = {cg, cl,
// count, -, S(9), w_sum, bad, n_outliers}
static int horn_pass(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* P, const HornKernelPrm& k, double h[20], unsigned long long* bad_blocks,
                     unsigned long long* h_counts /* [8] or null */)
{
    double* part = ctx->gn_partials.p;
    double* sums = ctx->gn_sums.p;
    unsigned char* fl = ctx->horn_flags.p;
    hipLaunchKernelGGL(horn_centroid_kernel, dim3(HORN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                       P->lx.p, P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p, P->counts.p, fl, part);
    hipLaunchKernelGGL(horn_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, part, 7, 1, sums);
    if (k.n_blocks)
        hipLaunchKernelGGL(horn_block_bounds_kernel, dim3(1), dim3(64), 0, ctx->stream, fl, P->counts.p, k,
                           ctx->horn_bounds.p);
    hipLaunchKernelGGL(horn_cov_kernel, dim3(HORN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream, P->lx.p,
                       P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p, P->pp.p, P->counts.p, sums,
                       ctx->horn_bounds.p, k, fl, part);
    hipLaunchKernelGGL(horn_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, part, 12, 0, sums + 8, (const unsigned long long*)(h_counts ? P->counts.p : nullptr));
    // (round 6: into the context's page-locked page and waited for by polling, like the Gauss-Newton read-back -- a copy to the
    //  caller's stack is staged by the runtime, and the blocking wait adds its wake-up latency; the list's counts ride along with
    //  the first pass: they used to be a round trip of their own in front of it)
    if (!ctx->pinned) MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 4096, hipHostMallocDefault));
    double* const             ph = reinterpret_cast<double*>((char*)ctx->pinned + 1024);
    unsigned long long* const pb = reinterpret_cast<unsigned long long*>((char*)ctx->pinned + 1024 + 28 * sizeof(double));
    *pb = 0ull;
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(ph, sums, (h_counts ? 28 : 20) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (k.n_blocks)
        MP2P_TRY_HIP(ctx, hipMemcpyAsync(pb, ctx->horn_bounds.p + MP2P_HIP_MAX_WEIGHT_BLOCKS, sizeof(unsigned long long),
                                         hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    memcpy(h, ph, 20 * sizeof(double));
    *bad_blocks = *pb;
    if (h_counts) memcpy(h_counts, ph + 20, 8 * sizeof(unsigned long long));
    return MP2P_HIP_OK;
}

int horn_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* P, const mp2p_hip_horn_params* w,
               mp2p_hip_horn_result* res)
{
    memset(res, 0, sizeof(*res));
    // optimal_tf_horn.cpp:207-209
    MP2P_REQUIRE(ctx, w->w_pt2pt >= 0.0 && w->w_ln2ln >= 0.0 && w->w_pl2pl >= 0.0, "pair weights must be >= 0");
    MP2P_REQUIRE(ctx, w->n_weight_blocks <= MP2P_HIP_MAX_WEIGHT_BLOCKS, "at most 32 point_weights blocks are supported");
    MP2P_REQUIRE(ctx, w->robust_kernel == MP2P_HIP_KERNEL_NONE || w->has_current_estimate,
                 "robust kernel needs currentEstimateForRobust (visit_correspondences.h:197)");
    MP2P_TRY_HIP(ctx, ctx->gn_partials.ensure((size_t)GN_BLOCKS * NS));  // >= HORN_BLOCKS*16
    MP2P_TRY_HIP(ctx, ctx->gn_sums.ensure(NS));
    MP2P_TRY_HIP(ctx, ctx->horn_bounds.ensure(MP2P_HIP_MAX_WEIGHT_BLOCKS + 8));
    // visit_correspondences.h:76-86 (no line pairs on this path)
    MP2P_REQUIRE(ctx, w->w_pt2pt + w->w_ln2ln + w->w_pl2pl > 0.0, "all attitude weights are <= 0");
    HornKernelPrm k;
    memset(&k, 0, sizeof(k));
    k.use_scale = w->use_scale_outlier_detector ? 1 : 0, k.scale_thr = w->scale_outlier_threshold;
    k.w_pt2pt = w->w_pt2pt, k.w_ln2ln = w->w_ln2ln, k.w_pl2pl = w->w_pl2pl;  // (the attitude weights: horn_cov_kernel, from the list's counts)
    k.rk.kernel = w->robust_kernel, k.rk.c = w->robust_kernel_param;
    k.rk.c2 = w->robust_kernel_param * w->robust_kernel_param;
    for (int i = 0; i < 12; i++) k.est[i] = w->current_estimate[i];
    k.n_blocks = w->n_weight_blocks;
    for (uint32_t b = 0; b < w->n_weight_blocks; b++)
        k.blk_count[b] = w->weight_block_count[b], k.blk_w[b] = w->weight_block_w[b];

    // the outlier flags: one byte per point pairing the list CAN hold (its length is on the device; it comes back with the first pass)
    const size_t n_cap = P->cap_pt2pt ? P->cap_pt2pt : 1;
    {   // (the flags are only ever SET by a pass with the scale-outlier detector on: without it the previous call's zeros stand -- one
        //  launch less per solve, C2 is a dozen launches long)
        const size_t before = ctx->horn_flags.n;
        MP2P_TRY_HIP(ctx, ctx->horn_flags.ensure(n_cap));
        if (ctx->horn_flags.n != before) ctx->horn_flags_zero = 0;  // (a new allocation: contents unknown)
        if (ctx->horn_flags_zero < n_cap)
        {
            MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->horn_flags.p, 0, n_cap, ctx->stream));
            ctx->horn_flags_zero = n_cap;
        }
        if (w->use_scale_outlier_detector) ctx->horn_flags_zero = 0;  // this call may set some
    }
    double             h[20];
    unsigned long long h_counts[8], bad_blocks = 0;
    int                rc = horn_pass(ctx, P, k, h, &bad_blocks, h_counts);
    if (rc) return rc;
    const unsigned long long n = h_counts[0], nPl = h_counts[6];
    ctx->horn_n = n;
    // (the checks in the order of rounds 1-5, when the counts were read before the pass)
    MP2P_REQUIRE(ctx, h_counts[1] == 0 && h_counts[5] == 0,
                 "This solver cannot handle point-to-plane / point-to-line pairings (convert them first)");
    // eval_centroids_robust: ASSERT_GT_(nPt2Pt, outliers.size())  (Pairings.cpp:76)
    MP2P_REQUIRE(ctx, n > 0, "Horn: needs more point pairings than outliers (none given)");
    if (n + nPl < 3) return MP2P_HIP_OK;  // :98 needs >= 3 references
    MP2P_REQUIRE(ctx, bad_blocks == 0, "Pairings::point_weights blocks cover fewer pairs than paired_pt2pt");
    MP2P_REQUIRE(ctx, h[8 + 10] == 0.0, "Horn: a visited pairing has weight <= 0 (ASSERT_(wi > .0))");
    if (w->use_scale_outlier_detector && h[8 + 11] > 0.0)  // :224-236
    {
        MP2P_REQUIRE(ctx, (double)n > h[8 + 11], "Horn: every point pairing is a scale outlier");
        rc = horn_pass(ctx, P, k, h, &bad_blocks, nullptr);
        if (rc) return rc;
        MP2P_REQUIRE(ctx, bad_blocks == 0, "Pairings::point_weights blocks cover fewer pairs than paired_pt2pt");
        MP2P_REQUIRE(ctx, h[8 + 10] == 0.0, "Horn: a visited pairing has weight <= 0 (ASSERT_(wi > .0))");
    }
    res->n_outliers = (uint64_t)h[8 + 11];
    const double* cg = h;
    const double* cl = h + 3;
    double        S[9];
    const double  w_sum = h[8 + 9];
    for (int i = 0; i < 9; i++) S[i] = (w_sum > 0) ? h[8 + i] * (1.0 / w_sum) : h[8 + i];  // :128
    double N[16];
    N[0] = S[0] + S[4] + S[8], N[1] = S[5] - S[7], N[2] = S[6] - S[2], N[3] = S[1] - S[3];
    N[4] = N[1], N[5] = S[0] - S[4] - S[8], N[6] = S[1] + S[3], N[7] = S[6] + S[2];
    N[8] = N[2], N[9] = N[6], N[10] = -S[0] + S[4] - S[8], N[11] = S[5] + S[7];
    N[12] = N[3], N[13] = N[7], N[14] = N[11], N[15] = -S[0] - S[4] + S[8];
    double ev[4], V[16];
    host_jacobi4(N, ev, V);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (ev[i] > ev[best]) best = i;
    double q[4] = {V[best * 4], V[best * 4 + 1], V[best * 4 + 2], V[best * 4 + 3]};
    if (q[0] < 0)
        for (double& v : q) v = -v;  // :165-171
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= qn;
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    double*      T = res->pose;
    T[0] = r * r + x * x - y * y - z * z, T[1] = 2 * (x * y - r * z), T[2] = 2 * (z * x + r * y);
    T[3] = 2 * (x * y + r * z), T[4] = r * r - x * x + y * y - z * z, T[5] = 2 * (y * z - r * x);
    T[6] = 2 * (z * x - r * y), T[7] = 2 * (y * z + r * x), T[8] = r * r - x * x - y * y + z * z;
    for (int i = 0; i < 3; i++)  // :238-247
        T[9 + i] = cg[i] - (T[i * 3] * cl[0] + T[i * 3 + 1] * cl[1] + T[i * 3 + 2] * cl[2]);
    res->solved = 1;
    return MP2P_HIP_OK;
}

// ---- pt2ln_pl_to_pt2pt (pt2ln_pl_to_pt2pt.cpp:47-113) ---------------------------------------------
// key = |distance| as a double (bit pattern orders like the value), payload = the new pair
struct CvPair
{
    float gx, gy, gz, lx, ly, lz;
};

__global__ __launch_bounds__(256) void cv_planes_kernel(const double* __restrict__ coef, const float* __restrict__ lx,
                                                        const float* __restrict__ ly, const float* __restrict__ lz,
                                                        unsigned long long n, PoseRt pose,
                                                        unsigned long long* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, CvPair* __restrict__ pairs)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = (double)lx[i], y = (double)ly[i], z = (double)lz[i];
    const double g0 = pose.r[0] * x + pose.r[1] * y + pose.r[2] * z + pose.t[0];  // :61 composePoint
    const double g1 = pose.r[3] * x + pose.r[4] * y + pose.r[5] * z + pose.t[1];
    const double g2 = pose.r[6] * x + pose.r[7] * y + pose.r[8] * z + pose.t[2];
    const double* c = coef + i * 4;
    const double  d = c[0] * g0 + c[1] * g1 + c[2] * g2 + c[3];  // :69 evaluatePoint
    keys[i] = (unsigned long long)__double_as_longlong(fabs(d));
    vals[i] = (uint32_t)i;
    pairs[i] = {(float)(g0 - c[0] * d), (float)(g1 - c[1] * d), (float)(g2 - c[2] * d), lx[i], ly[i], lz[i]};  // :71-79
}

__global__ __launch_bounds__(256) void cv_lines_kernel(const mp2p_hip_pair_pt2ln* __restrict__ ln,
                                                       unsigned long long n, PoseRt pose,
                                                       unsigned long long* __restrict__ keys,
                                                       uint32_t* __restrict__ vals, CvPair* __restrict__ pairs)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mp2p_hip_pair_pt2ln& q = ln[i];
    const double x = q.pt_local[0], y = q.pt_local[1], z = q.pt_local[2];
    const double g0 = pose.r[0] * x + pose.r[1] * y + pose.r[2] * z + pose.t[0];
    const double g1 = pose.r[3] * x + pose.r[4] * y + pose.r[5] * z + pose.t[1];
    const double g2 = pose.r[6] * x + pose.r[7] * y + pose.r[8] * z + pose.t[2];
    // TLine3D::closestPointTo (:97)
    const double* b = q.ln_base;
    const double* u = q.ln_director;
    const double  t = ((g0 - b[0]) * u[0] + (g1 - b[1]) * u[1] + (g2 - b[2]) * u[2]) /
                     (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double c0 = b[0] + t * u[0], c1 = b[1] + t * u[1], c2 = b[2] + t * u[2];
    const double e0 = c0 - g0, e1 = c1 - g1, e2 = c2 - g2;
    keys[i] = (unsigned long long)__double_as_longlong(fabs(sqrt(e0 * e0 + e1 * e1 + e2 * e2)));  // :98,108
    vals[i] = (uint32_t)i;
    pairs[i] = {(float)c0, (float)c1, (float)c2, (float)x, (float)y, (float)z};
}

// append_from_sorted (:25-45): from the largest error down to 25 % of it, but at least until the
// output holds 3 pairs.  keys ascending -> take = max(#(key >= thr), min(n, 3 - already))
__global__ void cv_take_kernel(const unsigned long long* __restrict__ keys, unsigned long long n,
                               unsigned long long* __restrict__ counts, unsigned long long cap,
                               unsigned long long* __restrict__ take_out)
{
    const double       largest = __longlong_as_double((long long)keys[n - 1]);
    const double       thr     = largest * 0.25;
    unsigned long long lo = 0, hi = n;  // first index with key >= thr
    while (lo < hi)
    {
        const unsigned long long mid = (lo + hi) / 2;
        if (__longlong_as_double((long long)keys[mid]) < thr) lo = mid + 1;
        else hi = mid;
    }
    unsigned long long       take    = n - lo;
    const unsigned long long already = counts[0];
    const unsigned long long need    = already < 3 ? 3 - already : 0;
    if (take < need) take = need < n ? need : n;
    if (already + take > cap) take = cap > already ? cap - already : 0, counts[4] = 1;
    take_out[0] = take, take_out[1] = already;
    counts[0]   = already + take;
}

__global__ __launch_bounds__(256) void cv_write_kernel(const uint32_t* __restrict__ sorted_vals,
                                                       const CvPair* __restrict__ pairs, unsigned long long n,
                                                       const unsigned long long* __restrict__ take,
                                                       uint32_t* o_lidx, uint32_t* o_gidx, float* o_lx, float* o_ly,
                                                       float* o_lz, float* o_gx, float* o_gy, float* o_gz,
                                                       float* o_err)
{
    const unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= take[0]) return;
    // descending walk; among equal keys the later insertion first = the stable ascending sort read backwards
    const CvPair             p   = pairs[sorted_vals[n - 1 - j]];
    const unsigned long long dst = take[1] + j;
    o_lidx[dst] = 0, o_gidx[dst] = 0;  // :73-74 dummy indices
    o_lx[dst] = p.lx, o_ly[dst] = p.ly, o_lz[dst] = p.lz;
    o_gx[dst] = p.gx, o_gy[dst] = p.gy, o_gz[dst] = p.gz;
    o_err[dst] = 0.f;
}

static int cv_append(mp2p_hip_ctx* ctx, size_t n, Scratch<unsigned long long>& k0, Scratch<unsigned long long>& k1,
                     Scratch<uint32_t>& v0, Scratch<uint32_t>& v1, Scratch<CvPair>& pairs,
                     Scratch<unsigned long long>& take, mp2p_hip_pairs* out)
{
    if (!n) return MP2P_HIP_OK;  // :33
    size_t                tmp_bytes = 0;
    MP2P_REQUIRE_INT_COUNT(ctx, n);
    Scratch<unsigned char> tmp;
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, 64,
                                                         ctx->stream));
    MP2P_TRY_HIP(ctx, tmp.take(ctx, 6, tmp_bytes));
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, 64,
                                                         ctx->stream));
    hipLaunchKernelGGL(cv_take_kernel, dim3(1), dim3(1), 0, ctx->stream, k1.p, (unsigned long long)n, out->counts.p,
                       (unsigned long long)out->cap_pt2pt, take.p);
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(cv_write_kernel, dim3(nb), dim3(256), 0, ctx->stream, v1.p, pairs.p, (unsigned long long)n,
                       take.p, out->lidx.p, out->gidx.p, out->lx.p, out->ly.p, out->lz.p, out->gx.p, out->gy.p,
                       out->gz.p, out->err.p);
    MP2P_TRY_HIP(ctx, stream_wait(ctx));  // tmp is a temporary
    return MP2P_HIP_OK;
}

// out must be empty (the reference starts from a fresh Pairings: the input's point pairs are NOT kept)
int pt2ln_pl_to_pt2pt(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* in, const double pose[12], mp2p_hip_pairs* out)
{
    unsigned long long h_counts[8];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_counts, in->counts.p, sizeof(h_counts), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    const size_t n_pl = h_counts[1], n_ln = h_counts[5];
    MP2P_REQUIRE(ctx, out->cap_pt2pt >= n_pl + n_ln, "output Pairings too small for the converted pairs");
    const size_t m = std::max<size_t>(std::max(n_pl, n_ln), 1);
    Scratch<unsigned long long> k0, k1, take;
    Scratch<uint32_t>           v0, v1;
    Scratch<CvPair>             pairs;
    MP2P_TRY_HIP(ctx, k0.take(ctx, 0, m));
    MP2P_TRY_HIP(ctx, k1.take(ctx, 1, m));
    MP2P_TRY_HIP(ctx, v0.take(ctx, 2, m));
    MP2P_TRY_HIP(ctx, v1.take(ctx, 3, m));
    MP2P_TRY_HIP(ctx, pairs.take(ctx, 4, m));
    MP2P_TRY_HIP(ctx, take.take(ctx, 5, 2));
    PoseRt T;
    for (int i = 0; i < 9; i++) T.r[i] = pose[i];
    for (int i = 0; i < 3; i++) T.t[i] = pose[9 + i];
    if (n_pl)
        hipLaunchKernelGGL(cv_planes_kernel, dim3((uint32_t)((n_pl + 255) / 256)), dim3(256), 0, ctx->stream,
                           in->pl_coef.p, in->pl_lx.p, in->pl_ly.p, in->pl_lz.p, (unsigned long long)n_pl, T, k0.p,
                           v0.p, pairs.p);
    int rc = cv_append(ctx, n_pl, k0, k1, v0, v1, pairs, take, out);
    if (rc) return rc;
    if (n_ln)
        hipLaunchKernelGGL(cv_lines_kernel, dim3((uint32_t)((n_ln + 255) / 256)), dim3(256), 0, ctx->stream,
                           in->ln.p, (unsigned long long)n_ln, T, k0.p, v0.p, pairs.p);
    rc = cv_append(ctx, n_ln, k0, k1, v0, v1, pairs, take, out);
    if (rc) return rc;
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

}  // namespace mp2p
