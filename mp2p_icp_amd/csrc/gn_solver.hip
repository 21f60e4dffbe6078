// gn_solver.hip -- K6/K7/K8: Gauss-Newton normal equations and SE(3) update.
//
// Replaces optimal_tf_gauss_newton (optimal_tf_gauss_newton.cpp:36-372).  Per inner
// iteration the reference forms, for every pair, J1 (3x12) * dDexpe_de (12x6), then
// H += w Ji^T Ji and g += w Ji^T e through dense Eigen products (~900 flop / pair).
// With Ji = [R | -R [l]x] (pt2pt) and Ji = -n n^T [R | -R [l]x] (pt2pl) those sums collapse to
// (SURVEY.md Appendix B):
//   pt2pt : 17 fp64 sums  {Sw, Sw*l (3), Sw*l l^T (6), Sw*e' (3), Sw*(l x e') (3), Sw*|e|^2}
//           with e' = R^T e, independent of R in the Hessian part;
//   pt2pl : 28 fp64 sums  {Sw*a a^T (21 upper), Sw*a*r (6), Sw*r^2}, a = [n' ; l x n'],
//           n' = R^T n/|n|, r = signed point-plane distance;
//   pt2ln / pl2pl (host-produced lists, small): the same 28 sums {H upper, g, cost} with
//           Ji = (I - u u^T)[R | -R [l]x]  resp.  Ji = [0 | -R [n_l]x], added by the same kernel.
// The kernels stream the pair arrays once per inner iteration (HBM-bound: 24 B / pt2pt pair,
// 44 B / pt2pl pair), reduce per lane -> wave (shuffles) -> block (LDS) -> fixed-order final
// sum (deterministic, no fp64 atomics).  The 6x6 solve, the prior term and the retraction
// run in a single-thread kernel so that the whole inner loop needs no host round trip;
// between "sums" and "step" a multi-GPU caller all-reduces the 48 doubles.
// H and g are rebuilt every inner iteration (TBB-build meaning, :145-146).
#include "device_utils.hpp"

namespace mp2p
{
constexpr int GN_BLOCKS  = 256;
constexpr int GN_THREADS = 256;
constexpr int NS         = MP2P_HIP_GN_NSUMS;  // 48
constexpr int NS_PT      = 17;
constexpr int NS_PL      = 28;

// gn_state layout (doubles): pose[12] H[36] g[6] cost iters done
constexpr int ST_POSE = 0, ST_H = 12, ST_G = 48, ST_COST = 54, ST_ITERS = 55, ST_DONE = 56,
              ST_SIZE = 64;

struct GnKernelPrm
{
    int    kernel;
    double c, c2;
    double w_pt2pt, w_pt2pl, w_pt2ln, w_pl2pl;
    uint32_t           n_blocks;  // weight blocks
    unsigned long long blk_end[8];
    double             blk_w[8];
};

// robust_kernels.h:57-94 (weight on the SQUARED error)
__device__ __forceinline__ double robust_w(const GnKernelPrm& p, double esq)
{
    if (p.kernel == MP2P_HIP_KERNEL_GEMANMCCLURE)
    {
        const double d = esq + p.c;
        return p.c2 / (d * d);
    }
    if (p.kernel == MP2P_HIP_KERNEL_CAUCHY) return p.c2 / (esq + p.c2);
    return 1.0;
}

// AGENT: the partials are handed to another workgroup inside the same launch (gn_iter_kernel, mode 2):
// agent-scope atomic stores (write-through, no L2 write-back fence needed on this side)
template <int N, bool AGENT = false>
__device__ __forceinline__ void block_reduce_store(double (&acc)[N], double* __restrict__ out)
{
    __shared__ double s[GN_THREADS / 64][N];
    const int         lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = wave_sum_f64(acc[k]);
    if (lane == 0)
    {
#pragma unroll
        for (int k = 0; k < N; k++) s[w][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < N)
    {
        double t = 0;
        for (int k = 0; k < GN_THREADS / 64; k++) t += s[k][threadIdx.x];
        if (AGENT)
        {
            __hip_atomic_store(out + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        else
            out[threadIdx.x] = t;
    }
}

// ---- K6: point-to-point (errorTerms.cpp:36-66 + optimal_tf_gauss_newton.cpp:149-180) --------
template <bool AGENT = false>
__device__ __forceinline__ void accum_pt2pt_body(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    unsigned long long n, const double (&R)[9], const double (&t)[3], const GnKernelPrm& prm,
    double* __restrict__ partials)
{
    double acc[NS_PT];
#pragma unroll
    for (int k = 0; k < NS_PT; k++) acc[k] = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n;
         i += (unsigned long long)GN_BLOCKS * GN_THREADS)
    {
        const double l0 = lx[i], l1 = ly[i], l2 = lz[i];
        // e = R l + t - g   (composePoint then subtract, errorTerms.cpp:44-48)
        const double e0 = R[0] * l0 + R[1] * l1 + R[2] * l2 + t[0] - (double)gx[i];
        const double e1 = R[3] * l0 + R[4] * l1 + R[5] * l2 + t[1] - (double)gy[i];
        const double e2 = R[6] * l0 + R[7] * l1 + R[8] * l2 + t[2] - (double)gz[i];
        const double esq = e0 * e0 + e1 * e1 + e2 * e2;
        double       w   = prm.w_pt2pt;
        if (prm.n_blocks)
        {  // Pairings::point_weights run-length blocks (:159-167)
            uint32_t b = 0;
            while (b + 1 < prm.n_blocks && i >= prm.blk_end[b]) b++;
            w = prm.blk_w[b];
        }
        w *= robust_w(prm, esq);
        // e' = R^T e
        const double p0 = R[0] * e0 + R[3] * e1 + R[6] * e2;
        const double p1 = R[1] * e0 + R[4] * e1 + R[7] * e2;
        const double p2 = R[2] * e0 + R[5] * e1 + R[8] * e2;
        acc[0] += w;
        acc[1] += w * l0, acc[2] += w * l1, acc[3] += w * l2;
        acc[4] += w * l0 * l0, acc[5] += w * l0 * l1, acc[6] += w * l0 * l2;
        acc[7] += w * l1 * l1, acc[8] += w * l1 * l2, acc[9] += w * l2 * l2;
        acc[10] += w * p0, acc[11] += w * p1, acc[12] += w * p2;
        acc[13] += w * (l1 * p2 - l2 * p1);
        acc[14] += w * (l2 * p0 - l0 * p2);
        acc[15] += w * (l0 * p1 - l1 * p0);
        acc[16] += w * esq;
    }
    block_reduce_store<NS_PT, AGENT>(acc, partials + (size_t)blockIdx.x * NS);
}

__global__ __launch_bounds__(GN_THREADS) void gn_accum_pt2pt_kernel(
    const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ lz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const unsigned long long* __restrict__ counts, const double* __restrict__ state,
    const GnKernelPrm prm, double* __restrict__ partials)
{
    const bool done = state[ST_DONE] != 0.0;
    const unsigned long long n = done ? 0ull : counts[0];
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = state[ST_POSE + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = state[ST_POSE + 9 + k];
    accum_pt2pt_body(lx, ly, lz, gx, gy, gz, n, R, t, prm, partials);
}

// ---- K7: point-to-plane (errorTerms.cpp:115-161 + optimal_tf_gauss_newton.cpp:229-259) ------
// one 3-row term: H(upper) += w Ji^T Ji, g += w Ji^T e     (Ji row-major 3x6)
__device__ __forceinline__ void accum_rows3(double (&acc)[28], const double (&J)[18], const double (&e)[3],
                                            double w)
{
    int k = 0;
#pragma unroll
    for (int p = 0; p < 6; p++)
#pragma unroll
        for (int q = p; q < 6; q++)
            acc[k++] += w * (J[p] * J[q] + J[6 + p] * J[6 + q] + J[12 + p] * J[12 + q]);
#pragma unroll
    for (int p = 0; p < 6; p++) acc[21 + p] += w * (J[p] * e[0] + J[6 + p] * e[1] + J[12 + p] * e[2]);
}

template <bool AGENT = false>
__device__ __forceinline__ void accum_pt2pl_body(
    const double* __restrict__ coef, const float* __restrict__ lx, const float* __restrict__ ly,
    const float* __restrict__ lz, const mp2p_hip_pair_pt2ln* __restrict__ lines,
    const mp2p_hip_pair_pl2pl* __restrict__ planes, const unsigned long long* __restrict__ counts,
    bool done, const double (&R)[9], const double (&t)[3], const GnKernelPrm& prm,
    double* __restrict__ partials)
{
    double acc[NS_PL];
#pragma unroll
    for (int k = 0; k < NS_PL; k++) acc[k] = 0;
    const unsigned long long n = (done || !coef) ? 0ull : counts[1];

    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n;
         i += (unsigned long long)GN_BLOCKS * GN_THREADS)
    {
        const double2 ab = *reinterpret_cast<const double2*>(coef + i * 4);
        const double2 cd = *reinterpret_cast<const double2*>(coef + i * 4 + 2);
        const double  l0 = lx[i], l1 = ly[i], l2 = lz[i];
        const double  g0 = R[0] * l0 + R[1] * l1 + R[2] * l2 + t[0];
        const double  g1 = R[3] * l0 + R[4] * l1 + R[5] * l2 + t[1];
        const double  g2 = R[6] * l0 + R[7] * l1 + R[8] * l2 + t[2];
        const double  m    = ab.x * ab.x + ab.y * ab.y + cd.x * cd.x;  // |n|^2 (not assumed 1)
        const double  inv  = 1.0 / sqrt(m);
        const double  nx = ab.x * inv, ny = ab.y * inv, nz = cd.x * inv;
        const double  r  = (ab.x * g0 + ab.y * g1 + cd.x * g2 + cd.y) * inv;  // |e|^2 = r^2
        const double  w  = prm.w_pt2pl * robust_w(prm, r * r);
        // n' = R^T n ;  a = [n' ; l x n']
        double a[6];
        a[0] = R[0] * nx + R[3] * ny + R[6] * nz;
        a[1] = R[1] * nx + R[4] * ny + R[7] * nz;
        a[2] = R[2] * nx + R[5] * ny + R[8] * nz;
        a[3] = l1 * a[2] - l2 * a[1];
        a[4] = l2 * a[0] - l0 * a[2];
        a[5] = l0 * a[1] - l1 * a[0];
        int k = 0;
#pragma unroll
        for (int p = 0; p < 6; p++)
        {
            const double wa = w * a[p];
#pragma unroll
            for (int q = p; q < 6; q++) acc[k++] += wa * a[q];
        }
#pragma unroll
        for (int p = 0; p < 6; p++) acc[21 + p] += w * a[p] * r;
        acc[27] += w * r * r;
    }

    // ---- point-to-line (errorTerms.cpp:68-112 + optimal_tf_gauss_newton.cpp:184-202) ---------
    const unsigned long long n_ln = (done || !lines) ? 0ull : counts[5];
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n_ln;
         i += (unsigned long long)GN_BLOCKS * GN_THREADS)
    {
        const mp2p_hip_pair_pt2ln P = lines[i];
        const double l[3] = {P.pt_local[0], P.pt_local[1], P.pt_local[2]};
        const double* u   = P.ln_director;  // not assumed unit (:77-84)
        double q[3];
        for (int r = 0; r < 3; r++)
            q[r] = R[r * 3] * l[0] + R[r * 3 + 1] * l[1] + R[r * 3 + 2] * l[2] + t[r] - P.ln_base[r];
        const double uq   = u[0] * q[0] + u[1] * q[1] + u[2] * q[2];
        const double e[3] = {q[0] - u[0] * uq, q[1] - u[1] * uq, q[2] - u[2] * uq};
        const double esq  = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        const double w    = prm.w_pt2ln * robust_w(prm, esq);
        // A = [R | -R [l]x]: column 3+j = R (e_j x l);  Ji = (I - u u^T) A
        double A[18];
        const double c[3][3] = {{0, -l[2], l[1]}, {l[2], 0, -l[0]}, {-l[1], l[0], 0}};  // e_j x l
        for (int r = 0; r < 3; r++)
        {
            for (int j = 0; j < 3; j++) A[r * 6 + j] = R[r * 3 + j];
            for (int j = 0; j < 3; j++)
                A[r * 6 + 3 + j] = R[r * 3] * c[j][0] + R[r * 3 + 1] * c[j][1] + R[r * 3 + 2] * c[j][2];
        }
        double J[18];
        for (int col = 0; col < 6; col++)
        {
            const double uA = u[0] * A[col] + u[1] * A[6 + col] + u[2] * A[12 + col];
            for (int r = 0; r < 3; r++) J[r * 6 + col] = A[r * 6 + col] - u[r] * uA;
        }
        accum_rows3(acc, J, e, w);
        acc[27] += w * w * esq;  // :198 (the weight enters squared here, unlike :175)
    }

    // ---- plane-to-plane, normals only (errorTerms.cpp:325-363 + ...gauss_newton.cpp:289-308) --
    const unsigned long long n_pp = (done || !planes) ? 0ull : counts[6];
    for (unsigned long long i = (unsigned long long)blockIdx.x * GN_THREADS + threadIdx.x; i < n_pp;
         i += (unsigned long long)GN_BLOCKS * GN_THREADS)
    {
        const mp2p_hip_pair_pl2pl P = planes[i];
        const double* nl = P.pl_local;
        double e[3];
        for (int r = 0; r < 3; r++)
            e[r] = R[r * 3] * nl[0] + R[r * 3 + 1] * nl[1] + R[r * 3 + 2] * nl[2] - P.pl_global[r];
        const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        const double w   = prm.w_pl2pl * robust_w(prm, esq);
        const double c[3][3] = {{0, -nl[2], nl[1]}, {nl[2], 0, -nl[0]}, {-nl[1], nl[0], 0}};  // e_j x n_l
        double J[18];
        for (int r = 0; r < 3; r++)
        {
            for (int j = 0; j < 3; j++) J[r * 6 + j] = 0.0;  // insensitive to translation
            for (int j = 0; j < 3; j++)
                J[r * 6 + 3 + j] = R[r * 3] * c[j][0] + R[r * 3 + 1] * c[j][1] + R[r * 3 + 2] * c[j][2];
        }
        accum_rows3(acc, J, e, w);
        acc[27] += w * w * esq;  // :303
    }
    block_reduce_store<NS_PL, AGENT>(acc, partials + (size_t)blockIdx.x * NS + NS_PT);
}

__global__ __launch_bounds__(GN_THREADS) void gn_accum_pt2pl_kernel(
    const double* __restrict__ coef, const float* __restrict__ lx, const float* __restrict__ ly,
    const float* __restrict__ lz, const mp2p_hip_pair_pt2ln* __restrict__ lines,
    const mp2p_hip_pair_pl2pl* __restrict__ planes, const unsigned long long* __restrict__ counts,
    const double* __restrict__ state, const GnKernelPrm prm, double* __restrict__ partials)
{
    const bool done = state[ST_DONE] != 0.0;
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = state[ST_POSE + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = state[ST_POSE + 9 + k];
    accum_pt2pl_body(coef, lx, ly, lz, lines, planes, counts, done, R, t, prm, partials);
}

// fixed-order sum of the block partials -> sums[48] (16 interleaved partial sums per quantity,
// combined as a fixed tree: deterministic run to run)
__device__ __forceinline__ void gn_sums_body(const double* __restrict__ partials, bool done,
                                             int use_pt, int use_pl, double* __restrict__ sums)
{
    __shared__ double s[16][NS];
    const int         q = threadIdx.x & 63, part = threadIdx.x >> 6;
    double            t = 0;
    const bool mine = q < NS && ((q < NS_PT && use_pt) || (q >= NS_PT && q < NS_PT + NS_PL && use_pl));
    if (mine && !done)
        for (int b = part; b < GN_BLOCKS; b += 16) t += partials[(size_t)b * NS + q];
    if (q < NS) s[part][q] = t;
    __syncthreads();
    if (threadIdx.x < NS)
    {
        const int i = threadIdx.x;
        double    r[8];
        for (int k = 0; k < 8; k++) r[k] = s[2 * k][i] + s[2 * k + 1][i];
        sums[i] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
}

__global__ __launch_bounds__(1024) void gn_sums_kernel(const double* __restrict__ partials,
                                                       const double* __restrict__ state,
                                                       int use_pt, int use_pl,
                                                       double* __restrict__ sums)
{
    gn_sums_body(partials, state[ST_DONE] != 0.0, use_pt, use_pl, sums);
}

// ---- small dense helpers (single thread) -----------------------------------------------------
__device__ void d_mat3_mul(const double* A, const double* B, double* C)
{
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = r[i];
}
__device__ void d_skew(const double* w, double* S)
{
    S[0] = 0, S[1] = -w[2], S[2] = w[1];
    S[3] = w[2], S[4] = 0, S[5] = -w[0];
    S[6] = -w[1], S[7] = w[0], S[8] = 0;
}
__device__ void d_pose_compose(const double* A, const double* B, double* out)
{
    double r[12];
    d_mat3_mul(A, B, r);
    for (int i = 0; i < 3; i++)
        r[9 + i] = A[i * 3] * B[9] + A[i * 3 + 1] * B[10] + A[i * 3 + 2] * B[11] + A[9 + i];
    for (int i = 0; i < 12; i++) out[i] = r[i];
}
__device__ void d_pose_inverse(const double* A, double* out)
{
    double r[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 3 + j] = A[j * 3 + i];
    for (int i = 0; i < 3; i++) r[9 + i] = -(r[i * 3] * A[9] + r[i * 3 + 1] * A[10] + r[i * 3 + 2] * A[11]);
    for (int i = 0; i < 12; i++) out[i] = r[i];
}
// Lie::SE<3>::exp, xi = [v; w] (true exponential)
__device__ void d_se3_exp(const double* xi, double* T)
{
    const double* v   = xi;
    const double* w   = xi + 3;
    const double  th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double  th  = sqrt(th2);
    double        a, b, c;
    if (th < 1e-6)
        a = 1.0 - th2 / 6.0, b = 0.5 - th2 / 24.0, c = 1.0 / 6.0 - th2 / 120.0;
    else
        a = sin(th) / th, b = (1.0 - cos(th)) / th2, c = (th - sin(th)) / (th2 * th);
    double W[9], W2[9];
    d_skew(w, W);
    d_mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) T[i] = a * W[i] + b * W2[i];
    T[0] += 1.0, T[4] += 1.0, T[8] += 1.0;
    double V[9];
    for (int i = 0; i < 9; i++) V[i] = b * W[i] + c * W2[i];
    V[0] += 1.0, V[4] += 1.0, V[8] += 1.0;
    for (int i = 0; i < 3; i++) T[9 + i] = V[i * 3] * v[0] + V[i * 3 + 1] * v[1] + V[i * 3 + 2] * v[2];
}
__device__ void d_so3_log(const double* R, double* w)
{
    const double tr = R[0] + R[4] + R[8];
    double       c  = 0.5 * (tr - 1.0);
    c               = fmin(1.0, fmax(-1.0, c));
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    const double s  = 0.5 * sqrt(vx * vx + vy * vy + vz * vz);
    const double th = atan2(s, c);
    if (th < 1e-6)
    {
        const double k = 0.5 * (1.0 + th * th / 6.0);
        w[0] = k * vx, w[1] = k * vy, w[2] = k * vz;
        return;
    }
    if (3.14159265358979323846 - th < 1e-6)
    {
        double ax[3] = {sqrt(fmax(0.0, 0.5 * (R[0] + 1.0))), sqrt(fmax(0.0, 0.5 * (R[4] + 1.0))),
                        sqrt(fmax(0.0, 0.5 * (R[8] + 1.0)))};
        int    k     = 0;
        if (ax[1] > ax[k]) k = 1;
        if (ax[2] > ax[k]) k = 2;
        for (int i = 0; i < 3; i++)
            if (i != k && (R[k * 3 + i] + R[i * 3 + k]) < 0) ax[i] = -ax[i];
        if (vx * ax[0] + vy * ax[1] + vz * ax[2] < 0)
            for (int i = 0; i < 3; i++) ax[i] = -ax[i];
        w[0] = th * ax[0], w[1] = th * ax[1], w[2] = th * ax[2];
        return;
    }
    const double k = th / (2.0 * s);
    w[0] = k * vx, w[1] = k * vy, w[2] = k * vz;
}
__device__ void d_se3_log(const double* T, double* xi)
{
    double w[3];
    d_so3_log(T, w);
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th  = sqrt(th2);
    double       k;
    if (th < 1e-6)
        k = 1.0 / 12.0 + th2 / 720.0;
    else
        k = (1.0 - (th * sin(th)) / (2.0 * (1.0 - cos(th)))) / th2;
    double W[9], W2[9], Vi[9];
    d_skew(w, W);
    d_mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) Vi[i] = -0.5 * W[i] + k * W2[i];
    Vi[0] += 1.0, Vi[4] += 1.0, Vi[8] += 1.0;
    for (int i = 0; i < 3; i++) xi[i] = Vi[i * 3] * T[9] + Vi[i * 3 + 1] * T[10] + Vi[i * 3 + 2] * T[11];
    xi[3] = w[0], xi[4] = w[1], xi[5] = w[2];
}

// LDL^T with diagonal pivoting (role of Eigen's H.ldlt().solve(g), :351)
// ws: >= 78 doubles of workspace (the callers pass LDS: indexed private arrays would live in
// scratch memory, i.e. a global-memory round trip per element -- this solve used to cost 12 us)
__device__ void d_ldlt6_solve(const double* H, const double* g, double* x, double* ws)
{
    const int n = 6;
    double *  A = ws, *L = ws + 36, *D = ws + 72;
    int       perm[6];
    for (int i = 0; i < 36; i++) A[i] = H[i], L[i] = 0;
    for (int i = 0; i < n; i++) perm[i] = i, D[i] = 0;
    for (int k = 0; k < n; k++)
    {
        int    piv  = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + i]) > best) best = fabs(A[i * n + i]), piv = i;
        if (piv != k)
        {
            for (int j = 0; j < n; j++)
            {
                const double t = A[k * n + j];
                A[k * n + j] = A[piv * n + j], A[piv * n + j] = t;
            }
            for (int i = 0; i < n; i++)
            {
                const double t = A[i * n + k];
                A[i * n + k] = A[i * n + piv], A[i * n + piv] = t;
            }
            for (int j = 0; j < k; j++)
            {
                const double t = L[k * n + j];
                L[k * n + j] = L[piv * n + j], L[piv * n + j] = t;
            }
            const int t = perm[k];
            perm[k] = perm[piv], perm[piv] = t;
        }
        D[k]         = A[k * n + k];
        L[k * n + k] = 1.0;
        if (D[k] == 0.0) continue;
        for (int i = k + 1; i < n; i++) L[i * n + k] = A[i * n + k] / D[k];
        for (int i = k + 1; i < n; i++)
            for (int j = k + 1; j < n; j++) A[i * n + j] -= L[i * n + k] * D[k] * L[j * n + k];
    }
    double b[6], y[6], z[6];
    for (int i = 0; i < n; i++) b[i] = g[perm[i]];
    for (int i = 0; i < n; i++)
    {
        double s = b[i];
        for (int j = 0; j < i; j++) s -= L[i * n + j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < n; i++) z[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
    for (int i = n - 1; i >= 0; i--)
    {
        double s = z[i];
        for (int j = i + 1; j < n; j++) s -= L[j * n + i] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < n; i++) x[perm[i]] = y[i];
}

struct GnStepPrm
{
    double minDelta, maxCost;
    int    has_prior;
    double prior_mean[12];
    double prior_cov_inv[36];
    int    use_pt, use_pl;
};

// ---- K8: assemble H,g from the sums, prior, solve, retract (one thread) -----------------------
constexpr int GN_STEP_WS = 36 + 6 + 78 + 2;  // H, g, LDLT workspace

__device__ void gn_step_body(const double* __restrict__ sums, double* __restrict__ state,
                             const GnStepPrm& prm, double* ws)
{
    if (state[ST_DONE] != 0.0) return;
    double T[12];
    for (int i = 0; i < 12; i++) T[i] = state[ST_POSE + i];
    double *H = ws, *g = ws + 36;  // LDS workspace (see d_ldlt6_solve)
    for (int i = 0; i < 36; i++) H[i] = 0;
    for (int i = 0; i < 6; i++) g[i] = 0;
    double cost = 0;
    if (prm.use_pt)
    {
        const double* s  = sums;
        const double  sw = s[0];
        const double  sl[3] = {s[1], s[2], s[3]};
        const double  xx = s[4], xy = s[5], xz = s[6], yy = s[7], yz = s[8], zz = s[9];
        // H_vv = Sw I
        H[0 * 6 + 0] += sw, H[1 * 6 + 1] += sw, H[2 * 6 + 2] += sw;
        // H_vw = -[Sw l]x ; H_wv = transpose
        double K[9];
        d_skew(sl, K);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
            {
                H[i * 6 + 3 + j] += -K[i * 3 + j];
                H[(3 + j) * 6 + i] += -K[i * 3 + j];
            }
        // H_ww = tr(Sll) I - Sll
        const double tr = xx + yy + zz;
        H[3 * 6 + 3] += tr - xx, H[3 * 6 + 4] += -xy, H[3 * 6 + 5] += -xz;
        H[4 * 6 + 3] += -xy, H[4 * 6 + 4] += tr - yy, H[4 * 6 + 5] += -yz;
        H[5 * 6 + 3] += -xz, H[5 * 6 + 4] += -yz, H[5 * 6 + 5] += tr - zz;
        for (int i = 0; i < 6; i++) g[i] += s[10 + i];
        cost += s[16];
    }
    if (prm.use_pl)
    {
        const double* s = sums + NS_PT;
        int           k = 0;
        for (int p = 0; p < 6; p++)
            for (int q = p; q < 6; q++)
            {
                H[p * 6 + q] += s[k];
                if (q != p) H[q * 6 + p] += s[k];
                k++;
            }
        for (int p = 0; p < 6; p++) g[p] += s[21 + p];
        cost += s[27];
    }
    if (prm.has_prior)
    {
        // :311-341  err = log(prior^-1 * pose); J = d log(A exp(eps))/d eps (central differences)
        double Pinv[12], A[12], err[6], J[36];
        d_pose_inverse(prm.prior_mean, Pinv);
        d_pose_compose(Pinv, T, A);
        d_se3_log(A, err);
        const double h = 1e-6;
        for (int j = 0; j < 6; j++)
        {
            double xi[6] = {0, 0, 0, 0, 0, 0}, E[12], Ap[12], Am[12], lp[6], lm[6];
            xi[j] = h;
            d_se3_exp(xi, E);
            d_pose_compose(A, E, Ap);
            xi[j] = -h;
            d_se3_exp(xi, E);
            d_pose_compose(A, E, Am);
            d_se3_log(Ap, lp);
            d_se3_log(Am, lm);
            for (int i = 0; i < 6; i++) J[i * 6 + j] = (lp[i] - lm[i]) / (2 * h);
        }
        double JtL[36];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++)
            {
                double s = 0;
                for (int k = 0; k < 6; k++) s += J[k * 6 + i] * prm.prior_cov_inv[k * 6 + j];
                JtL[i * 6 + j] = s;
            }
        for (int i = 0; i < 6; i++)
        {
            double s = 0;
            for (int k = 0; k < 6; k++) s += JtL[i * 6 + k] * err[k];
            g[i] += s;
            for (int j = 0; j < 6; j++)
            {
                double q = 0;
                for (int k = 0; k < 6; k++) q += JtL[i * 6 + k] * J[k * 6 + j];
                H[i * 6 + j] += q;
            }
        }
    }
    for (int i = 0; i < 36; i++) state[ST_H + i] = H[i];
    for (int i = 0; i < 6; i++) state[ST_G + i] = g[i];
    state[ST_COST] = cost;
    state[ST_ITERS] += 1.0;
    if (sqrt(cost) <= prm.maxCost)  // :344-346
    {
        state[ST_DONE] = 1.0;
        return;
    }
    double delta[6];
    d_ldlt6_solve(H, g, delta, ws + 42);
    for (int i = 0; i < 6; i++) delta[i] = -delta[i];  // :351
    double dE[12], Tn[12];
    d_se3_exp(delta, dE);       // :354
    d_pose_compose(T, dE, Tn);  // :356
    for (int i = 0; i < 12; i++) state[ST_POSE + i] = Tn[i];
    double nrm = 0;
    for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
    if (sqrt(nrm) < prm.minDelta) state[ST_DONE] = 1.0;  // :365
}

__global__ void gn_step_kernel(const double* __restrict__ sums, double* __restrict__ state,
                               const GnStepPrm prm)
{
    __shared__ double ws[GN_STEP_WS];
    if (threadIdx.x == 0 && blockIdx.x == 0) gn_step_body(sums, state, prm, ws);
}

// single-GPU form: final reduction and the 6x6 step in one launch (no all-reduce in between)
__global__ __launch_bounds__(1024) void gn_sums_step_kernel(const double* __restrict__ partials,
                                                            double* __restrict__ state, int use_pt,
                                                            int use_pl, double* __restrict__ sums,
                                                            const GnStepPrm prm)
{
    __shared__ double ws[GN_STEP_WS];
    gn_sums_body(partials, state[ST_DONE] != 0.0, use_pt, use_pl, sums);
    __syncthreads();  // sums[] written by this block are visible to its thread 0
    if (threadIdx.x == 0) gn_step_body(sums, state, prm, ws);
}

struct GnInit
{
    double pose[12];
};
// the linearisation point travels as a kernel argument: no H2D copy, no host synchronisation
__global__ void gn_init_kernel(double* __restrict__ state, const GnInit init)
{
    const int i = threadIdx.x;
    if (i < ST_SIZE) state[i] = (i < 12) ? init.pose[i] : 0.0;
}

// ---- one launch per inner iteration (single GPU): every block accumulates its share of the pairs
//      and takes a ticket; the LAST block to arrive adds the block partials in a fixed order and runs
//      the 6x6 step.  The first iteration takes the linearisation point from the kernel arguments
//      (no init launch).  Hand-off between the blocks = the agent-scope release / acquire pair of
//      MI355X_MICROARCH.md ("inter-workgroup visibility"): plain stores, __syncthreads, one lane's
//      release fence + s_waitcnt, relaxed ticket atomic; the last block: one acquire fence,
//      __syncthreads, plain loads.  ~2 us per side against ~8 us per saved launch boundary.
struct GnIterArgs
{
    const float *lx, *ly, *lz, *gx, *gy, *gz;        // pt2pt SoA
    const double* coef;                              // pt2pl
    const float * pl_lx, *pl_ly, *pl_lz;
    const mp2p_hip_pair_pt2ln* lines;
    const mp2p_hip_pair_pl2pl* planes;
    const unsigned long long*  counts;
    double *     state, *partials, *sums;
    unsigned int* ticket;
    GnKernelPrm  kprm;
    GnStepPrm    sprm;
    GnInit       init;
    int          first, use_pt, use_pl;
};

template <bool AGENT>
__global__ __launch_bounds__(GN_THREADS) void gn_iter_kernel(const GnIterArgs a)
{
    __shared__ int    s_last;
    __shared__ double s_part[GN_THREADS / 64][NS];
    __shared__ double s_ws[GN_STEP_WS];
    const bool done = a.first ? false : (a.state[ST_DONE] != 0.0);
    double     R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = a.first ? a.init.pose[k] : a.state[ST_POSE + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = a.first ? a.init.pose[9 + k] : a.state[ST_POSE + 9 + k];

    if (a.use_pt)
        accum_pt2pt_body<AGENT>(a.lx, a.ly, a.lz, a.gx, a.gy, a.gz, done ? 0ull : a.counts[0], R, t, a.kprm,
                                a.partials);
    if (a.use_pl)
        accum_pt2pl_body<AGENT>(a.coef, a.pl_lx, a.pl_ly, a.pl_lz, a.lines, a.planes, a.counts, done, R, t,
                                a.kprm, a.partials);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        if (!AGENT)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned int tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == gridDim.x - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0)
    {
        if (!AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *a.ticket = 0u;  // for the next launch (ordered by the kernel boundary)
    }
    __syncthreads();
    // fixed-order sum of the block partials: 4 interleaved partial sums per quantity, then a fixed tree
    {
        const int  q = threadIdx.x & 63, part = threadIdx.x >> 6;
        const bool mine = q < NS && ((q < NS_PT && a.use_pt) || (q >= NS_PT && q < NS_PT + NS_PL && a.use_pl));
        double     v = 0;
        if (mine && !done)
            for (int b = part; b < GN_BLOCKS; b += GN_THREADS / 64)
                v += AGENT ? __hip_atomic_load(a.partials + (size_t)b * NS + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : a.partials[(size_t)b * NS + q];
        if (q < NS) s_part[part][q] = v;
        __syncthreads();
        if (threadIdx.x < NS)
        {
            const int i = threadIdx.x;
            a.sums[i]   = (s_part[0][i] + s_part[1][i]) + (s_part[2][i] + s_part[3][i]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        if (a.first)
            for (int i = 0; i < ST_SIZE; i++) a.state[i] = (i < 12) ? a.init.pose[i] : 0.0;
        gn_step_body(a.sums, a.state, a.sprm, s_ws);
    }
}

static GnKernelPrm make_kernel_prm(const mp2p_hip_gn_params& p)
{
    GnKernelPrm k;
    memset(&k, 0, sizeof(k));
    k.kernel  = p.kernel;
    k.c       = p.kernelParam;
    k.c2      = p.kernelParam * p.kernelParam;
    k.w_pt2pt = p.w_pt2pt, k.w_pt2pl = p.w_pt2pl;
    k.w_pt2ln = p.w_pt2ln, k.w_pl2pl = p.w_pl2pl;
    k.n_blocks = p.n_weight_blocks;
    unsigned long long end = 0;
    for (uint32_t b = 0; b < p.n_weight_blocks && b < 8; b++)
    {
        end += p.weight_block_count[b];
        k.blk_end[b] = end;
        k.blk_w[b]   = p.weight_block_w[b];
    }
    return k;
}

// the state vector starts as {pose0, zeros}: written by a launch of its own (split form), or by the
// first fused iteration
static int gn_ensure_state(mp2p_hip_ctx* ctx)
{
    if (ctx->gn.state_ready) return MP2P_HIP_OK;
    GnInit init;
    for (int i = 0; i < 12; i++) init.pose[i] = ctx->gn.pose0[i];
    hipLaunchKernelGGL(gn_init_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->gn_state.p, init);
    ctx->gn.state_ready = true;
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int gn_begin(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose0[12],
             const mp2p_hip_gn_params* prm, bool lazy_init)
{
    MP2P_REQUIRE(ctx, pairs && pose0 && prm, "null argument");
    MP2P_REQUIRE(ctx, prm->n_weight_blocks <= 8, "at most 8 point_weights blocks are supported");
    // Pairings::point_weights semantics of this build: block b covers the next
    // weight_block_count[b] point pairings, at EVERY inner iteration.  The reference's cursor
    // (optimal_tf_gauss_newton.cpp:66-68, 159-167) is not reset between inner iterations (from the
    // second one on it reads past its list) and steps over an empty block one point late; an empty
    // block is therefore refused rather than given either meaning.
    for (uint32_t b = 0; b < prm->n_weight_blocks; b++)
        MP2P_REQUIRE(ctx, prm->weight_block_count[b] > 0, "point_weights: a block of zero pairings");
    MP2P_TRY_HIP(ctx, ctx->gn_partials.ensure((size_t)GN_BLOCKS * NS));
    MP2P_TRY_HIP(ctx, ctx->gn_sums.ensure(NS));
    MP2P_TRY_HIP(ctx, ctx->gn_state.ensure(ST_SIZE));
    ctx->gn.pairs  = pairs;
    ctx->gn.prm    = *prm;
    ctx->gn.active = true;
    for (int i = 0; i < 12; i++) ctx->gn.pose0[i] = pose0[i];
    ctx->gn.state_ready = false;
    if (ctx->prof_all()) MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    if (!lazy_init) return gn_ensure_state(ctx);
    return MP2P_HIP_OK;
}

static GnStepPrm make_step_prm(mp2p_hip_ctx* ctx)
{
    const mp2p_hip_gn_params& p = ctx->gn.prm;
    GnStepPrm                 s;
    memset(&s, 0, sizeof(s));
    s.minDelta = p.minDelta, s.maxCost = p.maxCost, s.has_prior = p.has_prior;
    memcpy(s.prior_mean, p.prior_mean, sizeof(s.prior_mean));
    memcpy(s.prior_cov_inv, p.prior_cov_inv, sizeof(s.prior_cov_inv));
    s.use_pt = ctx->gn.pairs->cap_pt2pt > 0;
    s.use_pl = ctx->gn.pairs->cap_pt2pl > 0 || ctx->gn.pairs->ln.p || ctx->gn.pairs->pp.p;
    return s;
}

static void launch_partials(mp2p_hip_ctx* ctx, int& use_pt, int& use_pl)
{
    const mp2p_hip_pairs* P = ctx->gn.pairs;
    const GnKernelPrm     k = make_kernel_prm(ctx->gn.prm);
    use_pt = P->cap_pt2pt > 0, use_pl = P->cap_pt2pl > 0 || P->ln.p || P->pp.p;
    if (use_pt)
        hipLaunchKernelGGL(gn_accum_pt2pt_kernel, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                           P->lx.p, P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p, P->counts.p,
                           ctx->gn_state.p, k, ctx->gn_partials.p);
    if (use_pl)
        hipLaunchKernelGGL(gn_accum_pt2pl_kernel, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                           P->cap_pt2pl > 0 ? P->pl_coef.p : nullptr, P->pl_lx.p, P->pl_ly.p, P->pl_lz.p,
                           P->ln.p, P->pp.p, P->counts.p, ctx->gn_state.p, k, ctx->gn_partials.p);
}

int gn_accumulate(mp2p_hip_ctx* ctx)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_accumulate without gn_begin");
    int use_pt, use_pl;
    launch_partials(ctx, use_pt, use_pl);
    hipLaunchKernelGGL(gn_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->gn_partials.p,
                       ctx->gn_state.p, use_pt, use_pl, ctx->gn_sums.p);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int gn_step(mp2p_hip_ctx* ctx)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_step without gn_begin");
    hipLaunchKernelGGL(gn_step_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->gn_sums.p,
                       ctx->gn_state.p, make_step_prm(ctx));
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

// accumulate ; reduce ; step  without the all-reduce seam (one GPU)
int gn_iterate_fused(mp2p_hip_ctx* ctx)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_iterate_fused without gn_begin");
    if (ctx->tune.gn_ticket)
    {
        const mp2p_hip_pairs* P = ctx->gn.pairs;
        MP2P_TRY_HIP(ctx, ctx->gn_ticket.ensure(1));
        if (!ctx->gn_ticket_zeroed)
        {
            MP2P_TRY_HIP(ctx, hipMemsetAsync(ctx->gn_ticket.p, 0, sizeof(unsigned int), ctx->stream));
            ctx->gn_ticket_zeroed = true;
        }
        GnIterArgs a;
        memset(&a, 0, sizeof(a));
        a.lx = P->lx.p, a.ly = P->ly.p, a.lz = P->lz.p, a.gx = P->gx.p, a.gy = P->gy.p, a.gz = P->gz.p;
        a.coef  = P->cap_pt2pl > 0 ? P->pl_coef.p : nullptr;
        a.pl_lx = P->pl_lx.p, a.pl_ly = P->pl_ly.p, a.pl_lz = P->pl_lz.p;
        a.lines = P->ln.p, a.planes = P->pp.p, a.counts = P->counts.p;
        a.state = ctx->gn_state.p, a.partials = ctx->gn_partials.p, a.sums = ctx->gn_sums.p;
        a.ticket = ctx->gn_ticket.p;
        a.kprm = make_kernel_prm(ctx->gn.prm), a.sprm = make_step_prm(ctx);
        a.use_pt = a.sprm.use_pt, a.use_pl = a.sprm.use_pl;
        a.first = ctx->gn.state_ready ? 0 : 1;
        for (int i = 0; i < 12; i++) a.init.pose[i] = ctx->gn.pose0[i];
        if (ctx->tune.gn_ticket == 2)
            hipLaunchKernelGGL(gn_iter_kernel<true>, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(gn_iter_kernel<false>, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream, a);
        ctx->gn.state_ready = true;
        MP2P_TRY_HIP(ctx, hipGetLastError());
        return MP2P_HIP_OK;
    }
    if (const int rc = gn_ensure_state(ctx)) return rc;
    int use_pt, use_pl;
    launch_partials(ctx, use_pt, use_pl);
    hipLaunchKernelGGL(gn_sums_step_kernel, dim3(1), dim3(1024), 0, ctx->stream,
                       ctx->gn_partials.p, ctx->gn_state.p, use_pt, use_pl, ctx->gn_sums.p,
                       make_step_prm(ctx));
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int gn_end(mp2p_hip_ctx* ctx, mp2p_hip_gn_result* out)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_end without gn_begin");
    if (const int rc = gn_ensure_state(ctx)) return rc;  // maxInnerLoopIterations == 0
    double st[ST_SIZE];
    if (ctx->prof_all())
    {
        MP2P_TRY_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
        ctx->pending_gn = 1;
    }
    // 512 bytes to a pinned buffer, then wait for the stream by polling (the step ends here; a
    // blocking wait adds its wake-up latency to every outer iteration)
    if (!ctx->pinned)
        MP2P_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->pinned, 4096, hipHostMallocDefault));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->gn_state.p, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));
    memcpy(st, ctx->pinned, sizeof(st));
    ctx->gn.active = false;
    if (out)
    {
        memcpy(out->pose, st + ST_POSE, 12 * sizeof(double));
        memcpy(out->H, st + ST_H, 36 * sizeof(double));
        memcpy(out->g, st + ST_G, 6 * sizeof(double));
        out->cost       = st[ST_COST];
        out->iterations = (uint32_t)st[ST_ITERS];
    }
    return MP2P_HIP_OK;
}

}  // namespace mp2p
