// horn.hip -- "next #1": optimal_tf_horn (optimal_tf_horn.cpp:77-252) for point pairs:
// weighted centroids (eval_centroids_robust, Pairings.cpp:68-110) + 3x3 cross-covariance S
// (visit_correspondences.h:100-212, lambda_each_pair optimal_tf_horn.cpp:108-123) as two
// streaming fp64 reductions over the device-resident pairs; the 4x4 symmetric eigenproblem
// (:158) and the translation from the centroids (:238-247) are O(1) host work.
// Not covered: scale outlier detector, robust kernel, per-block weights.
#include "device_utils.hpp"

namespace mp2p
{
constexpr int HORN_BLOCKS = 512;

__global__ __launch_bounds__(GN_THREADS) void horn_centroid_kernel(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned long long* __restrict__ counts, double* __restrict__ partials)
{
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long n = counts[0];
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n;
         i += (unsigned long long)HORN_BLOCKS * GN_THREADS)
    {
        acc[0] += (double)gx[i], acc[1] += (double)gy[i], acc[2] += (double)gz[i];
        acc[3] += (double)lx[i], acc[4] += (double)ly[i], acc[5] += (double)lz[i];
    }
    block_reduce_store<6>(acc, partials + (size_t)blockIdx.x * 16);
}

__global__ __launch_bounds__(GN_THREADS) void horn_cov_kernel(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned long long* __restrict__ counts, const double* __restrict__ cent /*cg(3) cl(3)*/,
    double wi, double* __restrict__ partials)
{
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = 0;
    const unsigned long long n = counts[0];
    const double cg0 = cent[0], cg1 = cent[1], cg2 = cent[2], cl0 = cent[3], cl1 = cent[4], cl2 = cent[5];
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n;
         i += (unsigned long long)HORN_BLOCKS * GN_THREADS)
    {
        const double b0 = (double)gx[i] - cg0, b1 = (double)gy[i] - cg1, b2 = (double)gz[i] - cg2;
        const double r0 = (double)lx[i] - cl0, r1 = (double)ly[i] - cl1, r2 = (double)lz[i] - cl2;
        const double bn = sqrt(b0 * b0 + b1 * b1 + b2 * b2), rn = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
        if (bn < 1e-4 || rn < 1e-4) continue;  // visit_correspondences.h:135-140
        acc[0] += wi * r0 * b0, acc[1] += wi * r0 * b1, acc[2] += wi * r0 * b2;
        acc[3] += wi * r1 * b0, acc[4] += wi * r1 * b1, acc[5] += wi * r1 * b2;
        acc[6] += wi * r2 * b0, acc[7] += wi * r2 * b1, acc[8] += wi * r2 * b2;
        acc[9] += wi;
    }
    block_reduce_store<10>(acc, partials + (size_t)blockIdx.x * 16);
}

// fixed-order sum of [HORN_BLOCKS][16] partials; optionally scale the first 6 by 1/n (centroids)
__global__ __launch_bounds__(64) void horn_sum_kernel(const double* __restrict__ partials, int nq,
                                                      const unsigned long long* __restrict__ counts,
                                                      int centroid_mode, double* __restrict__ out)
{
    const int q = threadIdx.x;
    if (q >= nq) return;
    double t = 0;
    for (int b = 0; b < HORN_BLOCKS; b++) t += partials[(size_t)b * 16 + q];
    if (centroid_mode)
    {
        const unsigned long long n = counts[0];
        t *= (n ? 1.0 / (double)n : 0.0);  // wcPoints (Pairings.cpp:80)
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

int horn_solve(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* P, double w_pt2pt, double* pose_out,
               int32_t* solved)
{
    *solved = 0;
    MP2P_REQUIRE(ctx, w_pt2pt > 0.0, "pair_weights.pt2pt must be > 0");
    MP2P_TRY_HIP(ctx, ctx->gn_partials.ensure((size_t)GN_BLOCKS * NS));  // >= HORN_BLOCKS*16
    MP2P_TRY_HIP(ctx, ctx->gn_sums.ensure(NS));
    unsigned long long h_counts[8];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_counts, P->counts.p, sizeof(h_counts), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long n = h_counts[0];
    if (n < 3) return MP2P_HIP_OK;  // optimal_tf_horn.cpp:98: needs >= 3 references
    double* part = ctx->gn_partials.p;
    double* sums = ctx->gn_sums.p;
    hipLaunchKernelGGL(horn_centroid_kernel, dim3(HORN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                       P->lx.p, P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p, P->counts.p, part);
    hipLaunchKernelGGL(horn_sum_kernel, dim3(1), dim3(64), 0, ctx->stream, part, 6, P->counts.p, 1, sums);
    // waPoints = wPt / (wPt * nPt2Pt)   (visit_correspondences.h:76-86)
    const double wa = w_pt2pt * (1.0 / (w_pt2pt * (double)n));
    hipLaunchKernelGGL(horn_cov_kernel, dim3(HORN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream, P->lx.p,
                       P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p, P->counts.p, sums, wa, part);
    hipLaunchKernelGGL(horn_sum_kernel, dim3(1), dim3(64), 0, ctx->stream, part, 10, P->counts.p, 0, sums + 8);
    double h[18];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h, sums, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    for (int k = 1; k < 4; k++)
        if (ev[k] > ev[best]) best = k;
    double q[4] = {V[best * 4], V[best * 4 + 1], V[best * 4 + 2], V[best * 4 + 3]};
    if (q[0] < 0)
        for (double& v : q) v = -v;  // :165-171
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= qn;
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    double*      T = pose_out;
    T[0] = r * r + x * x - y * y - z * z, T[1] = 2 * (x * y - r * z), T[2] = 2 * (z * x + r * y);
    T[3] = 2 * (x * y + r * z), T[4] = r * r - x * x + y * y - z * z, T[5] = 2 * (y * z - r * x);
    T[6] = 2 * (z * x - r * y), T[7] = 2 * (y * z + r * x), T[8] = r * r - x * x - y * y + z * z;
    for (int i = 0; i < 3; i++)  // :238-247
        T[9 + i] = cg[i] - (T[i * 3] * cl[0] + T[i * 3 + 1] * cl[1] + T[i * 3 + 2] * cl[2]);
    *solved = 1;
    return MP2P_HIP_OK;
}

}  // namespace mp2p
