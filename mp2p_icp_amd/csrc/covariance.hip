// covariance.hip -- mp2p_icp::covariance (mp2p_icp/src/covariance.cpp:29-141, called once per
// ICP::align at ICP.cpp:334-337): the Hessian J^T J of the stacked error vector with respect to
// (x, y, z, yaw, pitch, roll), J by central finite differences as mrpt::math::estimateJacobian
// does, cov = H^-1.
//
// The reference evaluates the whole error vector 12 times (2 per parameter) and multiplies a
// (3 N x 6)^T (3 N x 6) product.  Here one streaming pass over the device-resident Pairings: the 12
// perturbed poses travel as kernel arguments, each pair forms its 3x6 block of J (same
// arithmetic as the oracle's error terms, no FMA) and adds its 21 products; deterministic
// lane -> wave -> block -> fixed-order final sum.  The 6x6 inverse (Cholesky, inverse_LLt) runs
// on the host.  paired_ln2ln is not supported.
#include "device_utils.hpp"

namespace mp2p
{
constexpr int CV_BLOCKS = 256, CV_THREADS = 256, CV_N = 21;

struct CovPoses
{
    double T[12][12];  // [2 j]: x + h_j e_j ; [2 j + 1]: x - h_j e_j   (R row-major, t)
    double s[6];       // 0.5 / h_j
};

__device__ __forceinline__ void cv_compose(const double* T, double lx, double ly, double lz, double (&g)[3])
{
    g[0] = T[0] * lx + T[1] * ly + T[2] * lz + T[9];
    g[1] = T[3] * lx + T[4] * ly + T[5] * lz + T[10];
    g[2] = T[6] * lx + T[7] * ly + T[8] * lz + T[11];
}

__device__ __forceinline__ void cv_accum(double (&acc)[CV_N], const double (&J)[18])
{
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
        for (int q = p; q < 6; q++) acc[k++] += J[p] * J[q] + J[6 + p] * J[6 + q] + J[12 + p] * J[12 + q];
}

__global__ __launch_bounds__(CV_THREADS) void cov_accum_kernel(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const double* __restrict__ coef, const float* __restrict__ plx, const float* __restrict__ ply,
    const float* __restrict__ plz, const mp2p_hip_pair_pt2ln* __restrict__ lines,
    const mp2p_hip_pair_pl2pl* __restrict__ planes, const unsigned long long* __restrict__ counts,
    const CovPoses P, double* __restrict__ partials)
{
    __shared__ double s_red[CV_THREADS / 64][CV_N];
    double            acc[CV_N];
#pragma unroll
    for (int k = 0; k < CV_N; k++) acc[k] = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * CV_THREADS;
    const unsigned long long first  = (unsigned long long)blockIdx.x * CV_THREADS + threadIdx.x;

    // point-to-point: e = T l - g   (errorTerms.cpp:36-66)
    const unsigned long long n_pt = lx ? counts[0] : 0ull;
    for (unsigned long long i = first; i < n_pt; i += stride)
    {
        const double l0 = lx[i], l1 = ly[i], l2 = lz[i], g0 = gx[i], g1 = gy[i], g2 = gz[i];
        double       J[18];
        for (int j = 0; j < 6; j++)
        {
            double a[3], b[3];
            cv_compose(P.T[2 * j], l0, l1, l2, a);
            cv_compose(P.T[2 * j + 1], l0, l1, l2, b);
            J[j]      = P.s[j] * ((a[0] - g0) - (b[0] - g0));
            J[6 + j]  = P.s[j] * ((a[1] - g1) - (b[1] - g1));
            J[12 + j] = P.s[j] * ((a[2] - g2) - (b[2] - g2));
        }
        cv_accum(acc, J);
    }
    // point-to-line: e = q - u (u.q), q = T l - pBase   (errorTerms.cpp:68-90)
    const unsigned long long n_ln = lines ? counts[5] : 0ull;
    for (unsigned long long i = first; i < n_ln; i += stride)
    {
        const mp2p_hip_pair_pt2ln Q = lines[i];
        const double*             u = Q.ln_director;
        double                    J[18];
        for (int j = 0; j < 6; j++)
        {
            double e[2][3];
            for (int sgn = 0; sgn < 2; sgn++)
            {
                double g[3];
                cv_compose(P.T[2 * j + sgn], Q.pt_local[0], Q.pt_local[1], Q.pt_local[2], g);
                const double q[3] = {g[0] - Q.ln_base[0], g[1] - Q.ln_base[1], g[2] - Q.ln_base[2]};
                const double uq   = u[0] * q[0] + u[1] * q[1] + u[2] * q[2];
                e[sgn][0] = q[0] - u[0] * uq, e[sgn][1] = q[1] - u[1] * uq, e[sgn][2] = q[2] - u[2] * uq;
            }
            for (int r = 0; r < 3; r++) J[r * 6 + j] = P.s[j] * (e[0][r] - e[1][r]);
        }
        cv_accum(acc, J);
    }
    // point-to-plane: e = -(n / |n|^2) (n.(T l) + d)   (errorTerms.cpp:115-134)
    const unsigned long long n_pl = coef ? counts[1] : 0ull;
    for (unsigned long long i = first; i < n_pl; i += stride)
    {
        const double c0 = coef[i * 4], c1 = coef[i * 4 + 1], c2 = coef[i * 4 + 2], c3 = coef[i * 4 + 3];
        const double l0 = plx[i], l1 = ply[i], l2 = plz[i];
        const double mod_n = c0 * c0 + c1 * c1 + c2 * c2;
        double       J[18];
        for (int j = 0; j < 6; j++)
        {
            double e[2][3];
            for (int sgn = 0; sgn < 2; sgn++)
            {
                double g[3];
                cv_compose(P.T[2 * j + sgn], l0, l1, l2, g);
                const double s = c0 * g[0] + c1 * g[1] + c2 * g[2] + c3;
                e[sgn][0] = -(c0 / mod_n) * s, e[sgn][1] = -(c1 / mod_n) * s, e[sgn][2] = -(c2 / mod_n) * s;
            }
            for (int r = 0; r < 3; r++) J[r * 6 + j] = P.s[j] * (e[0][r] - e[1][r]);
        }
        cv_accum(acc, J);
    }
    // plane-to-plane: e = R n_local - n_global   (errorTerms.cpp:325-340)
    const unsigned long long n_pp = planes ? counts[6] : 0ull;
    for (unsigned long long i = first; i < n_pp; i += stride)
    {
        const mp2p_hip_pair_pl2pl Q  = planes[i];
        const double*             nl = Q.pl_local;
        double                    J[18];
        for (int j = 0; j < 6; j++)
        {
            double e[2][3];
            for (int sgn = 0; sgn < 2; sgn++)
            {
                const double* T = P.T[2 * j + sgn];
                for (int r = 0; r < 3; r++)
                    e[sgn][r] = T[r * 3 + 0] * nl[0] + T[r * 3 + 1] * nl[1] + T[r * 3 + 2] * nl[2] - Q.pl_global[r];
            }
            for (int r = 0; r < 3; r++) J[r * 6 + j] = P.s[j] * (e[0][r] - e[1][r]);
        }
        cv_accum(acc, J);
    }

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < CV_N; k++) acc[k] = wave_sum_f64(acc[k]);
    if (lane == 0)
    {
#pragma unroll
        for (int k = 0; k < CV_N; k++) s_red[w][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < CV_N)
    {
        double t = 0;
        for (int k = 0; k < CV_THREADS / 64; k++) t += s_red[k][threadIdx.x];
        partials[(size_t)blockIdx.x * CV_N + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void cov_final_kernel(const double* __restrict__ partials, double* __restrict__ H21)
{
    if (threadIdx.x < CV_N)
    {
        double t = 0;
        for (int b = 0; b < CV_BLOCKS; b++) t += partials[(size_t)b * CV_N + threadIdx.x];  // fixed order
        H21[threadIdx.x] = t;
    }
}

// CPose3D::setFromValues / getYawPitchRoll conventions: R = Rz(yaw) Ry(pitch) Rx(roll)
static void cv_pose_from_xyzypr(const double x[6], double T[12])
{
    const double cy = cos(x[3]), sy = sin(x[3]), cp = cos(x[4]), sp = sin(x[4]), cr = cos(x[5]), sr = sin(x[5]);
    T[0] = cy * cp, T[1] = cy * sp * sr - sy * cr, T[2] = cy * sp * cr + sy * sr;
    T[3] = sy * cp, T[4] = sy * sp * sr + cy * cr, T[5] = sy * sp * cr - cy * sr;
    T[6] = -sp, T[7] = cp * sr, T[8] = cp * cr;
    T[9] = x[0], T[10] = x[1], T[11] = x[2];
}

static void cv_pose_to_ypr(const double T[12], double& yaw, double& pitch, double& roll)
{
    const double sp = -T[6];
    if (fabs(sp) > 1.0 - 1e-12)
    {  // gimbal lock: roll := 0
        pitch = sp > 0 ? M_PI / 2 : -M_PI / 2;
        yaw   = atan2(-T[1], T[4]);
        roll  = 0;
        return;
    }
    pitch = asin(sp);
    yaw   = atan2(T[3], T[0]);
    roll  = atan2(T[7], T[8]);
}

int covariance_run(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* P, const double pose[12], double h_xyz,
                   double h_ang, double H_out[36], double cov_out[36], int* positive_definite)
{
    unsigned long long c[8];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(c, P->counts.p, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    const unsigned long long n_terms = c[0] + c[1] + (P->ln.p ? c[5] : 0) + (P->pp.p ? c[6] : 0);
    *positive_definite = 0;
    if (n_terms == 0)
    {  // covariance.cpp:33-39
        for (int i = 0; i < 36; i++) cov_out[i] = 0, H_out[i] = 0;
        for (int i = 0; i < 6; i++) cov_out[i * 6 + i] = 1e6;
        return MP2P_HIP_OK;
    }
    CovPoses cp;
    double   x0[6] = {pose[9], pose[10], 0.0 /* :41-43: z is never assigned */, 0, 0, 0};
    cv_pose_to_ypr(pose, x0[3], x0[4], x0[5]);
    for (int j = 0; j < 6; j++)
    {
        const double h = j < 3 ? h_xyz : h_ang;
        double       x[6];
        memcpy(x, x0, sizeof(x));
        x[j] = x0[j] + h;
        cv_pose_from_xyzypr(x, cp.T[2 * j]);
        x[j] = x0[j] - h;
        cv_pose_from_xyzypr(x, cp.T[2 * j + 1]);
        cp.s[j] = 0.5 / h;
    }
    Scratch<double> partials, h21;
    MP2P_TRY_HIP(ctx, partials.take(ctx, 0, (size_t)CV_BLOCKS * CV_N));
    MP2P_TRY_HIP(ctx, h21.take(ctx, 1, CV_N));
    hipLaunchKernelGGL(cov_accum_kernel, dim3(CV_BLOCKS), dim3(CV_THREADS), 0, ctx->stream,
                       P->cap_pt2pt > 0 ? P->lx.p : nullptr, P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p,
                       P->cap_pt2pl > 0 ? P->pl_coef.p : nullptr, P->pl_lx.p, P->pl_ly.p, P->pl_lz.p, P->ln.p,
                       P->pp.p, P->counts.p, cp, partials.p);
    hipLaunchKernelGGL(cov_final_kernel, dim3(1), dim3(64), 0, ctx->stream, partials.p, h21.p);
    double s[CV_N];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(s, h21.p, sizeof(s), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    MP2P_TRY_HIP(ctx, hipGetLastError());
    double H[36];
    int    k = 0;
    for (int p = 0; p < 6; p++)
        for (int q = p; q < 6; q++) H[p * 6 + q] = H[q * 6 + p] = s[k++];
    memcpy(H_out, H, sizeof(H));
    // hessian.inverse_LLt()
    double L[36];
    memset(L, 0, sizeof(L));
    bool ok = true;
    for (int i = 0; i < 6 && ok; i++)
        for (int j = 0; j <= i; j++)
        {
            double v = H[i * 6 + j];
            for (int q = 0; q < j; q++) v -= L[i * 6 + q] * L[j * 6 + q];
            if (i == j)
            {
                if (!(v > 0)) { ok = false; break; }
                L[i * 6 + i] = sqrt(v);
            }
            else
                L[i * 6 + j] = v / L[j * 6 + j];
        }
    for (int col = 0; col < 6 && ok; col++)
    {
        double y[6], z[6];
        for (int i = 0; i < 6; i++)
        {
            double v = (i == col) ? 1.0 : 0.0;
            for (int q = 0; q < i; q++) v -= L[i * 6 + q] * y[q];
            y[i] = v / L[i * 6 + i];
        }
        for (int i = 5; i >= 0; i--)
        {
            double v = y[i];
            for (int q = i + 1; q < 6; q++) v -= L[q * 6 + i] * z[q];
            z[i] = v / L[i * 6 + i];
        }
        for (int i = 0; i < 6; i++) cov_out[i * 6 + col] = z[i];
    }
    if (!ok)
        for (int i = 0; i < 36; i++) cov_out[i] = NAN;
    *positive_definite = ok ? 1 : 0;
    return MP2P_HIP_OK;
}

}  // namespace mp2p
