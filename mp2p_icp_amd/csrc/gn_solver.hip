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
    unsigned long long blk_end[MP2P_HIP_MAX_WEIGHT_BLOCKS];
    double             blk_w[MP2P_HIP_MAX_WEIGHT_BLOCKS];
};

// the linearisation point of the first inner iteration, as a kernel argument
struct GnInit
{
    double pose[12];
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
    const GnKernelPrm prm, double* __restrict__ partials, const GnInit init, const int first)
{
    const bool done = first ? false : state[ST_DONE] != 0.0;
    const unsigned long long n = done ? 0ull : counts[0];
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = first ? init.pose[k] : state[ST_POSE + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = first ? init.pose[9 + k] : state[ST_POSE + 9 + k];
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
    const double* __restrict__ state, const GnKernelPrm prm, double* __restrict__ partials,
    const GnInit init, const int first)
{
    const bool done = first ? false : state[ST_DONE] != 0.0;
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = first ? init.pose[k] : state[ST_POSE + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = first ? init.pose[9 + k] : state[ST_POSE + 9 + k];
    accum_pt2pl_body(coef, lx, ly, lz, lines, planes, counts, done, R, t, prm, partials);
}

// both kinds of pairings in ONE launch (round 6): a list that CAN hold point pairings next to the plane ones (its capacity says so; the
// counts live on the device) used to cost the point kernel's launch at every inner iteration even when it holds none -- C3: three
// empty launches per step, 6 % of it.  The same two bodies one after the other, the same partial rows: sums bit-identical.
__global__ __launch_bounds__(GN_THREADS) void gn_accum_both_kernel(
    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
    const float* __restrict__ gx, const float* __restrict__ gy, const float* __restrict__ gz,
    const double* __restrict__ coef, const float* __restrict__ lx, const float* __restrict__ ly,
    const float* __restrict__ lz, const mp2p_hip_pair_pt2ln* __restrict__ lines,
    const mp2p_hip_pair_pl2pl* __restrict__ planes, const unsigned long long* __restrict__ counts,
    const double* __restrict__ state, const GnKernelPrm prm, double* __restrict__ partials,
    const GnInit init, const int first)
{
    const bool done = first ? false : state[ST_DONE] != 0.0;
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = first ? init.pose[k] : state[ST_POSE + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = first ? init.pose[9 + k] : state[ST_POSE + 9 + k];
    accum_pt2pt_body(px, py, pz, gx, gy, gz, done ? 0ull : counts[0], R, t, prm, partials);
    accum_pt2pl_body(coef, lx, ly, lz, lines, planes, counts, done, R, t, prm, partials);
}

// fixed-order sum of the block partials -> sums[48] (16 interleaved partial sums per quantity,
// combined as a fixed tree: deterministic run to run).  The 16 loads of a thread are independent (one
// round trip, not 16); the result goes to global memory (the all-reduce buffer) and to s_sums (LDS).
__device__ __forceinline__ void gn_sums_body(const double* __restrict__ partials, bool done,
                                             int use_pt, int use_pl, double* __restrict__ sums,
                                             double* s_sums)
{
    __shared__ double s[16][NS];
    const int         q = threadIdx.x & 63, part = threadIdx.x >> 6;
    const bool mine = q < NS && ((q < NS_PT && use_pt) || (q >= NS_PT && q < NS_PT + NS_PL && use_pl));
    double     v[GN_BLOCKS / 16];
#pragma unroll
    for (int k = 0; k < GN_BLOCKS / 16; k++)
        v[k] = (mine && !done) ? partials[(size_t)(part + 16 * k) * NS + q] : 0.0;
    double t = 0;
#pragma unroll
    for (int k = 0; k < GN_BLOCKS / 16; k++) t += v[k];
    if (q < NS) s[part][q] = t;
    __syncthreads();
    if (threadIdx.x < NS)
    {
        const int i = threadIdx.x;
        double    r[8];
        for (int k = 0; k < 8; k++) r[k] = s[2 * k][i] + s[2 * k + 1][i];
        const double tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        sums[i] = tot;
        if (s_sums) s_sums[i] = tot;
    }
}

__global__ __launch_bounds__(1024) void gn_sums_kernel(const double* __restrict__ partials,
                                                       const double* __restrict__ state,
                                                       int use_pt, int use_pl,
                                                       double* __restrict__ sums)
{
    gn_sums_body(partials, state[ST_DONE] != 0.0, use_pt, use_pl, sums, nullptr);
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

// LDL^T with diagonal pivoting (role of Eigen's H.ldlt().solve(g), :351), run by ONE WAVE: lane
// 6 i + j holds A(i,j) and L(i,j); pivots, D and the permutation are wave-uniform (v_readlane); the row /
// column exchanges are two ds_bpermute.  Same operations per element and in the same order as the
// textbook serial loop, which it replaced: that loop, run by one thread on an LDS copy of the matrix
// (indexed private arrays would live in scratch memory), was a chain of ~300 dependent LDS round trips,
// ~10 us of the 15 us gn_sums_step_kernel used to take.
__device__ __forceinline__ double readlane_d(double v, int l)
{
    const unsigned long long u  = __builtin_bit_cast(unsigned long long, v);
    const unsigned int       lo = (unsigned int)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), l);
    const unsigned int       hi = (unsigned int)__builtin_amdgcn_readlane((int)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// H (6x6 row-major) and g in LDS; x is returned in every lane
__device__ __forceinline__ void wave_ldlt6_solve(const double* H, const double* g, double (&x)[6])
{
    const int lane = threadIdx.x & 63;
    const int l    = lane < 36 ? lane : 35;
    const int i = l / 6, j = l % 6;
    double    a = H[l], Lv = 0.0;
    const double gv = g[j];            // lanes 0..5 hold g[0..5]
    int          pv = lane < 6 ? lane : 0;  // lanes 0..5 hold perm[0..5]
    double       D[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
    {
        int    piv  = k;
        double best = fabs(readlane_d(a, k * 7));
#pragma unroll
        for (int r = k + 1; r < 6; r++)
        {
            const double d = fabs(readlane_d(a, r * 7));
            if (d > best) best = d, piv = r;
        }
        if (piv != k)  // wave-uniform
        {
            const int pi = (i == k) ? piv : (i == piv) ? k : i;
            const int pj = (j == k) ? piv : (j == piv) ? k : j;
            a            = __shfl(a, pi * 6 + pj, 64);
            Lv           = __shfl(Lv, pi * 6 + j, 64);  // columns >= k of both rows are still zero
            const int pk = __builtin_amdgcn_readlane(pv, k), pp = __builtin_amdgcn_readlane(pv, piv);
            pv           = (lane == k) ? pp : (lane == piv) ? pk : pv;
        }
        const double Dk = readlane_d(a, k * 7);
        D[k]            = Dk;
        if (i == k && j == k) Lv = 1.0;
        if (Dk == 0.0) continue;
        const double lik = __shfl(a, i * 6 + k, 64) / Dk;
        const double ljk = __shfl(a, j * 6 + k, 64) / Dk;
        if (j == k && i > k) Lv = lik;
        if (i > k && j > k) a -= lik * Dk * ljk;
    }
    double Lr[6][6];
#pragma unroll
    for (int r = 1; r < 6; r++)
#pragma unroll
        for (int c = 0; c < r; c++) Lr[r][c] = readlane_d(Lv, r * 6 + c);
    int    perm[6];
    double y[6], z[6];
#pragma unroll
    for (int r = 0; r < 6; r++)
    {
        perm[r]  = __builtin_amdgcn_readlane(pv, r);
        double q = readlane_d(gv, perm[r]);  // b[r] = g[perm[r]]
#pragma unroll
        for (int c = 0; c < r; c++) q -= Lr[r][c] * y[c];
        y[r] = q;
    }
#pragma unroll
    for (int r = 0; r < 6; r++) z[r] = (D[r] != 0.0) ? y[r] / D[r] : 0.0;
#pragma unroll
    for (int r = 5; r >= 0; r--)
    {
        double q = z[r];
#pragma unroll
        for (int c = r + 1; c < 6; c++) q -= Lr[c][r] * y[c];
        y[r] = q;
    }
    double xt = 0.0;  // lane t < 6: x[t]
#pragma unroll
    for (int r = 0; r < 6; r++)
        if (perm[r] == lane) xt = y[r];
#pragma unroll
    for (int r = 0; r < 6; r++) x[r] = readlane_d(xt, r);
}

struct GnStepPrm
{
    double minDelta, maxCost;
    int    has_prior;
    double prior_mean[12];
    double prior_cov_inv[36];
    int    use_pt, use_pl;
};

// ---- K8: assemble H,g from the sums and the prior (one thread), solve and retract (one wave) ----
constexpr int GN_STEP_WS = 36 + 6 + 2;  // H, g, cost
static_assert(ST_SIZE == 64, "the state vector is stored by one wave");

template <bool PRIOR>
__device__ void gn_assemble(const double* sums, const double (&T)[12], const GnStepPrm& prm, double* ws)
{
    double *H = ws, *g = ws + 36;
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
    if (PRIOR)  // a kernel variant of its own: its 6x6 temporaries cost registers and scratch memory
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
    ws[42] = cost;
}

// Called by every thread of the block (it contains a barrier).  s_sums: the 48 sums in LDS; T, iters: the
// current iterate (from the kernel argument at the first iteration: the state vector needs no launch of
// its own); ws: GN_STEP_WS + ST_SIZE doubles of LDS.  The whole state vector leaves as one 512-byte store.
template <bool PRIOR>
__device__ void gn_step_block(const double* s_sums, double* __restrict__ state, const GnStepPrm& prm,
                              double* ws, const double (&T)[12], double iters)
{
    if (threadIdx.x == 0) gn_assemble<PRIOR>(s_sums, T, prm, ws);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int    lane  = threadIdx.x;
    double*      s_out = ws + GN_STEP_WS;
    const double cost  = ws[42];
    bool         done  = sqrt(cost) <= prm.maxCost;  // :344-346
    double       Tn[12];
    for (int i = 0; i < 12; i++) Tn[i] = T[i];
    if (!done)
    {
        double delta[6], dE[12];
        wave_ldlt6_solve(ws, ws + 36, delta);
        for (int i = 0; i < 6; i++) delta[i] = -delta[i];  // :351
        d_se3_exp(delta, dE);                              // :354
        d_pose_compose(T, dE, Tn);                         // :356
        double nrm = 0;
        for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
        done = sqrt(nrm) < prm.minDelta;  // :365
    }
    if (lane < 36) s_out[ST_H + lane] = ws[lane];
    if (lane < 6) s_out[ST_G + lane] = ws[36 + lane];
    if (lane >= ST_DONE + 1) s_out[lane] = 0.0;
    if (lane == 0)
    {
#pragma unroll
        for (int i = 0; i < 12; i++) s_out[ST_POSE + i] = Tn[i];
        s_out[ST_COST] = cost, s_out[ST_ITERS] = iters + 1.0, s_out[ST_DONE] = done ? 1.0 : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    state[lane] = s_out[lane];  // ST_SIZE == 64
}

// split form (an all-reduce sits between the sums and the step)
template <bool PRIOR>
__global__ __launch_bounds__(64) void gn_step_kernel(const double* __restrict__ sums, double* __restrict__ state,
                                                     const GnStepPrm prm)
{
    __shared__ double ws[GN_STEP_WS + ST_SIZE];
    __shared__ double s_sums[NS];
    if (state[ST_DONE] != 0.0) return;
    double T[12];
    for (int i = 0; i < 12; i++) T[i] = state[ST_POSE + i];
    const double iters = state[ST_ITERS];
    if (threadIdx.x < NS) s_sums[threadIdx.x] = sums[threadIdx.x];
    __syncthreads();
    gn_step_block<PRIOR>(s_sums, state, prm, ws, T, iters);
}

// single-GPU form: final reduction and the 6x6 step in one launch (no all-reduce in between)
template <bool PRIOR>
__global__ __launch_bounds__(1024) void gn_sums_step_kernel(const double* __restrict__ partials,
                                                            double* __restrict__ state, int use_pt,
                                                            int use_pl, double* __restrict__ sums,
                                                            const GnStepPrm prm, const GnInit init,
                                                            const int first)
{
    __shared__ double ws[GN_STEP_WS + ST_SIZE];
    __shared__ double s_sums[NS];
    // the iterate is read while the partials are on their way
    double T[12], iters = 0.0;
    bool   done = false;
    if (first)
        for (int i = 0; i < 12; i++) T[i] = init.pose[i];
    else
    {
        done = state[ST_DONE] != 0.0;
        for (int i = 0; i < 12; i++) T[i] = state[ST_POSE + i];
        iters = state[ST_ITERS];
    }
    if (done) return;  // block-uniform
    gn_sums_body(partials, false, use_pt, use_pl, sums, s_sums);
    __syncthreads();
    gn_step_block<PRIOR>(s_sums, state, prm, ws, T, iters);
}

__global__ void gn_init_kernel(double* __restrict__ state, const GnInit init)
{
    const int i = threadIdx.x;
    if (i < ST_SIZE) state[i] = (i < 12) ? init.pose[i] : 0.0;
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
    for (uint32_t b = 0; b < p.n_weight_blocks && b < MP2P_HIP_MAX_WEIGHT_BLOCKS; b++)
    {
        end += p.weight_block_count[b];
        k.blk_end[b] = end;
        k.blk_w[b]   = p.weight_block_w[b];
    }
    return k;
}

static GnInit make_init(const mp2p_hip_ctx* ctx)
{
    GnInit init;
    for (int i = 0; i < 12; i++) init.pose[i] = ctx->gn.pose0[i];
    return init;
}

// the state vector starts as {pose0, zeros}: written by a launch of its own (split form), or by the
// first fused iteration
static int gn_ensure_state(mp2p_hip_ctx* ctx)
{
    if (ctx->gn.state_ready) return MP2P_HIP_OK;
    hipLaunchKernelGGL(gn_init_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->gn_state.p, make_init(ctx));
    ctx->gn.state_ready = true;
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int gn_begin(mp2p_hip_ctx* ctx, const mp2p_hip_pairs* pairs, const double pose0[12],
             const mp2p_hip_gn_params* prm, bool lazy_init)
{
    MP2P_REQUIRE(ctx, pairs && pose0 && prm, "null argument");
    MP2P_REQUIRE(ctx, prm->n_weight_blocks <= MP2P_HIP_MAX_WEIGHT_BLOCKS, "at most 32 point_weights blocks are supported");
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

// first: the state vector is not written yet; the kernels take the linearisation point from `init`
static void launch_partials(mp2p_hip_ctx* ctx, int& use_pt, int& use_pl, int first)
{
    const GnInit init = make_init(ctx);
    const mp2p_hip_pairs* P = ctx->gn.pairs;
    const GnKernelPrm     k = make_kernel_prm(ctx->gn.prm);
    use_pt = P->cap_pt2pt > 0, use_pl = P->cap_pt2pl > 0 || P->ln.p || P->pp.p;
    if (use_pt && use_pl)
    {
        hipLaunchKernelGGL(gn_accum_both_kernel, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                           P->lx.p, P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p,
                           P->cap_pt2pl > 0 ? P->pl_coef.p : nullptr, P->pl_lx.p, P->pl_ly.p, P->pl_lz.p,
                           P->ln.p, P->pp.p, P->counts.p, ctx->gn_state.p, k, ctx->gn_partials.p, init, first);
        return;
    }
    if (use_pt)
        hipLaunchKernelGGL(gn_accum_pt2pt_kernel, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                           P->lx.p, P->ly.p, P->lz.p, P->gx.p, P->gy.p, P->gz.p, P->counts.p,
                           ctx->gn_state.p, k, ctx->gn_partials.p, init, first);
    if (use_pl)
        hipLaunchKernelGGL(gn_accum_pt2pl_kernel, dim3(GN_BLOCKS), dim3(GN_THREADS), 0, ctx->stream,
                           P->cap_pt2pl > 0 ? P->pl_coef.p : nullptr, P->pl_lx.p, P->pl_ly.p, P->pl_lz.p,
                           P->ln.p, P->pp.p, P->counts.p, ctx->gn_state.p, k, ctx->gn_partials.p, init, first);
}

int gn_accumulate(mp2p_hip_ctx* ctx)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_accumulate without gn_begin");
    if (const int rc = gn_ensure_state(ctx)) return rc;
    int use_pt, use_pl;
    launch_partials(ctx, use_pt, use_pl, 0);
    hipLaunchKernelGGL(gn_sums_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->gn_partials.p,
                       ctx->gn_state.p, use_pt, use_pl, ctx->gn_sums.p);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

int gn_step(mp2p_hip_ctx* ctx)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_step without gn_begin");
    const GnStepPrm sp = make_step_prm(ctx);
    if (sp.has_prior)
        hipLaunchKernelGGL(gn_step_kernel<true>, dim3(1), dim3(64), 0, ctx->stream, ctx->gn_sums.p, ctx->gn_state.p, sp);
    else
        hipLaunchKernelGGL(gn_step_kernel<false>, dim3(1), dim3(64), 0, ctx->stream, ctx->gn_sums.p, ctx->gn_state.p, sp);
    MP2P_TRY_HIP(ctx, hipGetLastError());
    return MP2P_HIP_OK;
}

// accumulate ; reduce ; step  without the all-reduce seam (one GPU)
int gn_iterate_fused(mp2p_hip_ctx* ctx)
{
    MP2P_REQUIRE(ctx, ctx->gn.active, "gn_iterate_fused without gn_begin");
    const int first = ctx->gn.state_ready ? 0 : 1;
    int       use_pt, use_pl;
    launch_partials(ctx, use_pt, use_pl, first);
    const GnStepPrm sp = make_step_prm(ctx);
    if (sp.has_prior)
        hipLaunchKernelGGL(gn_sums_step_kernel<true>, dim3(1), dim3(1024), 0, ctx->stream, ctx->gn_partials.p,
                           ctx->gn_state.p, use_pt, use_pl, ctx->gn_sums.p, sp, make_init(ctx), first);
    else
        hipLaunchKernelGGL(gn_sums_step_kernel<false>, dim3(1), dim3(1024), 0, ctx->stream, ctx->gn_partials.p,
                           ctx->gn_state.p, use_pt, use_pl, ctx->gn_sums.p, sp, make_init(ctx), first);
    ctx->gn.state_ready = true;
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
