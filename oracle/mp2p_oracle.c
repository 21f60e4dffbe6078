/*
 * mp2p_oracle.c -- CPU restatement of the mp2p_icp per-iteration hot path.
 * TEST INFRASTRUCTURE ONLY: see mp2p_oracle.h for the rules and the pinning status.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so the fp32
 * distance / threshold expressions round exactly like the reference's x86-64 build and like
 * the HIP kernels, which use __fmul_rn/__fadd_rn).
 */
#include "mp2p_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

const char* orc_version(void) { return "mp2p_oracle 0.1 (restates MOLAorg/mp2p_icp v1.8.0)"; }

/* ======================================================================================
 *  Small dense helpers
 * ====================================================================================== */
static void mat3_mul(const double* A, const double* B, double* C)
{
    double r[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] +
                           A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, r, sizeof(r));
}

static void skew(const double w[3], double S[9])
{
    S[0] = 0, S[1] = -w[2], S[2] = w[1];
    S[3] = w[2], S[4] = 0, S[5] = -w[0];
    S[6] = -w[1], S[7] = w[0], S[8] = 0;
}

/* cyclic Jacobi eigen-solver for symmetric n x n (n<=4). Eigenvalues ascending,
 * eigenvectors returned as ROWS of V (V[k*n + :] is the k-th eigenvector).
 * Stands in for mrpt::math::CMatrixFixed::eig_symmetric (estimate_points_eigen.cpp:111,
 * optimal_tf_horn.cpp:158); SURVEY.md Appendix C.7 (ascending order). */
static void sym_eig_jacobi(int n, const double* Ain, double* eval, double* V)
{
    double A[16], Q[16];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Q[i * n + j] = (i == j) ? 1.0 : 0.0;

    /* converged when the off-diagonal mass is below 1e-18 of the diagonal scale: a further
     * rotation would change nothing at double precision.  (Iterating until it is exactly 0 takes
     * tens of extra sweeps through the denormal range; the HIP plane fit uses the same rule.) */
    double scale = 0;
    for (int i = 0; i < n; i++) scale += fabs(A[i * n + i]);
    scale = 1e-36 * (scale * scale);
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
                const double t =
                    (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
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
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]])
            {
                int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    for (int k = 0; k < n; k++)
    {
        eval[k] = A[order[k] * n + order[k]];
        for (int i = 0; i < n; i++) V[k * n + i] = Q[i * n + order[k]];
    }
}

/* LDL^T with diagonal pivoting for a symmetric 6x6 (stands in for Eigen's
 * H.ldlt().solve(g), optimal_tf_gauss_newton.cpp:351). */
static void ldlt6_solve(const double* H, const double* g, double* x)
{
    const int n = 6;
    double    A[36];
    int       perm[6];
    memcpy(A, H, sizeof(A));
    for (int i = 0; i < n; i++) perm[i] = i;
    double L[36] = {0}, D[6] = {0};
    for (int k = 0; k < n; k++)
    {
        /* pivot: largest |diag| of the remaining Schur complement */
        int    piv  = k;
        double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + i]) > best) best = fabs(A[i * n + i]), piv = i;
        if (piv != k)
        {
            for (int j = 0; j < n; j++)
            {
                double t = A[k * n + j];
                A[k * n + j] = A[piv * n + j];
                A[piv * n + j] = t;
            }
            for (int i = 0; i < n; i++)
            {
                double t = A[i * n + k];
                A[i * n + k] = A[i * n + piv];
                A[i * n + piv] = t;
            }
            for (int j = 0; j < k; j++)
            {
                double t = L[k * n + j];
                L[k * n + j] = L[piv * n + j];
                L[piv * n + j] = t;
            }
            int t = perm[k];
            perm[k] = perm[piv];
            perm[piv] = t;
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

/* ======================================================================================
 *  SE(3)   (closed forms of the un-vendored MRPT calls)
 * ====================================================================================== */
void orc_pose_identity(double T[12])
{
    memset(T, 0, 12 * sizeof(double));
    T[0] = T[4] = T[8] = 1.0;
}

/* CPose3D(x,y,z,yaw,pitch,roll): R = Rz(yaw) Ry(pitch) Rx(roll)   (Appendix C.1) */
void orc_pose_from_xyzypr(double x, double y, double z, double yaw, double pitch, double roll,
                          double T[12])
{
    const double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll),
                 sr = sin(roll);
    T[0] = cy * cp, T[1] = cy * sp * sr - sy * cr, T[2] = cy * sp * cr + sy * sr;
    T[3] = sy * cp, T[4] = sy * sp * sr + cy * cr, T[5] = sy * sp * cr - cy * sr;
    T[6] = -sp, T[7] = cp * sr, T[8] = cp * cr;
    T[9] = x, T[10] = y, T[11] = z;
}

void orc_pose_to_xyzypr(const double T[12], double o[6])
{
    o[0] = T[9], o[1] = T[10], o[2] = T[11];
    const double sp = -T[6];
    double       pitch;
    if (sp >= 1.0)
        pitch = M_PI / 2;
    else if (sp <= -1.0)
        pitch = -M_PI / 2;
    else
        pitch = asin(sp);
    o[4] = pitch;
    if (fabs(fabs(sp) - 1.0) < 1e-12)
    { /* gimbal lock: roll := 0 */
        o[5] = 0;
        o[3] = atan2(-T[1], T[4]);
    }
    else
    {
        o[3] = atan2(T[3], T[0]);
        o[5] = atan2(T[7], T[8]);
    }
}

void orc_pose_compose(const double A[12], const double B[12], double out[12])
{
    double r[12];
    mat3_mul(A, B, r);
    for (int i = 0; i < 3; i++)
        r[9 + i] = A[i * 3 + 0] * B[9] + A[i * 3 + 1] * B[10] + A[i * 3 + 2] * B[11] + A[9 + i];
    memcpy(out, r, sizeof(r));
}

void orc_pose_inverse(const double A[12], double out[12])
{
    double r[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 3 + j] = A[j * 3 + i];
    for (int i = 0; i < 3; i++)
        r[9 + i] = -(r[i * 3 + 0] * A[9] + r[i * 3 + 1] * A[10] + r[i * 3 + 2] * A[11]);
    memcpy(out, r, sizeof(r));
}

/* CPose3D::composePoint in fp64, left-to-right (Appendix C.2) */
void orc_pose_compose_point(const double T[12], double lx, double ly, double lz, double g[3])
{
    g[0] = T[0] * lx + T[1] * ly + T[2] * lz + T[9];
    g[1] = T[3] * lx + T[4] * ly + T[5] * lz + T[10];
    g[2] = T[6] * lx + T[7] * ly + T[8] * lz + T[11];
}

void orc_pose_inverse_compose_point(const double T[12], double gx, double gy, double gz,
                                    double l[3])
{
    const double dx = gx - T[9], dy = gy - T[10], dz = gz - T[11];
    l[0] = T[0] * dx + T[3] * dy + T[6] * dz;
    l[1] = T[1] * dx + T[4] * dy + T[7] * dz;
    l[2] = T[2] * dx + T[5] * dy + T[8] * dz;
}

static void so3_exp(const double w[3], double R[9])
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th  = sqrt(th2);
    double       a, b; /* a = sin(th)/th, b = (1-cos(th))/th^2 */
    if (th < 1e-6)
        a = 1.0 - th2 / 6.0, b = 0.5 - th2 / 24.0;
    else
        a = sin(th) / th, b = (1.0 - cos(th)) / th2;
    double W[9], W2[9];
    skew(w, W);
    mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) R[i] = a * W[i] + b * W2[i];
    R[0] += 1.0, R[4] += 1.0, R[8] += 1.0;
}

static void so3_log(const double R[9], double w[3])
{
    const double tr = R[0] + R[4] + R[8];
    double       c  = 0.5 * (tr - 1.0);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
    const double s  = 0.5 * sqrt(vx * vx + vy * vy + vz * vz); /* = sin(theta) */
    const double th = atan2(s, c);
    if (th < 1e-6)
    {
        const double k = 0.5 * (1.0 + th * th / 6.0);
        w[0] = k * vx, w[1] = k * vy, w[2] = k * vz;
        return;
    }
    if (M_PI - th < 1e-6)
    { /* near pi: axis from the diagonal of (R+I)/2 */
        double ax[3] = {sqrt(fmax(0.0, 0.5 * (R[0] + 1.0))), sqrt(fmax(0.0, 0.5 * (R[4] + 1.0))),
                        sqrt(fmax(0.0, 0.5 * (R[8] + 1.0)))};
        int    k     = 0;
        if (ax[1] > ax[k]) k = 1;
        if (ax[2] > ax[k]) k = 2;
        /* signs from the symmetric part relative to the largest component */
        for (int i = 0; i < 3; i++)
            if (i != k && (R[k * 3 + i] + R[i * 3 + k]) < 0) ax[i] = -ax[i];
        /* overall sign from the antisymmetric part when it is not exactly zero */
        if (vx * ax[0] + vy * ax[1] + vz * ax[2] < 0)
            for (int i = 0; i < 3; i++) ax[i] = -ax[i];
        w[0] = th * ax[0], w[1] = th * ax[1], w[2] = th * ax[2];
        return;
    }
    const double k = th / (2.0 * s);
    w[0] = k * vx, w[1] = k * vy, w[2] = k * vz;
}

/* Lie::SE<3>::exp, true exponential, xi = [v; w]   (Appendix C.3) */
void orc_se3_exp(const double xi[6], double T[12])
{
    const double* v = xi;
    const double* w = xi + 3;
    so3_exp(w, T);
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th  = sqrt(th2);
    double       b, c; /* b = (1-cos)/th^2, c = (th - sin)/th^3 */
    if (th < 1e-6)
        b = 0.5 - th2 / 24.0, c = 1.0 / 6.0 - th2 / 120.0;
    else
        b = (1.0 - cos(th)) / th2, c = (th - sin(th)) / (th2 * th);
    double W[9], W2[9], V[9];
    skew(w, W);
    mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) V[i] = b * W[i] + c * W2[i];
    V[0] += 1.0, V[4] += 1.0, V[8] += 1.0;
    for (int i = 0; i < 3; i++) T[9 + i] = V[i * 3] * v[0] + V[i * 3 + 1] * v[1] + V[i * 3 + 2] * v[2];
}

/* Lie::SE<3>::log -> [v; w] (ICP.cpp:194-196 splits it that way) */
void orc_se3_log(const double T[12], double xi[6])
{
    double w[3];
    so3_log(T, w);
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th  = sqrt(th2);
    double       k; /* V^-1 = I - W/2 + k W^2 */
    if (th < 1e-6)
        k = 1.0 / 12.0 + th2 / 720.0;
    else
        k = (1.0 - (th * sin(th)) / (2.0 * (1.0 - cos(th)))) / th2;
    double W[9], W2[9], Vi[9];
    skew(w, W);
    mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) Vi[i] = -0.5 * W[i] + k * W2[i];
    Vi[0] += 1.0, Vi[4] += 1.0, Vi[8] += 1.0;
    for (int i = 0; i < 3; i++)
        xi[i] = Vi[i * 3] * T[9] + Vi[i * 3 + 1] * T[10] + Vi[i * 3 + 2] * T[11];
    xi[3] = w[0], xi[4] = w[1], xi[5] = w[2];
}

/* Lie::SE<3>::jacob_dDexpe_de(D): d vec([R t] (+) exp(eps)) / d eps at 0, 12x6,
 * rows = column-major [c1;c2;c3;t], eps=[v;w]   (optimal_tf_gauss_newton.cpp:73,
 * closed form SURVEY.md Appendix B) */
void orc_jacob_dDexpe_de(const double T[12], double J[72])
{
    memset(J, 0, 72 * sizeof(double));
    /* rows 0..8, cols 3..5:  -R [e_k]x  for column k */
    for (int k = 0; k < 3; k++)
    {
        double e[3] = {0, 0, 0}, E[9], M[9];
        e[k] = 1.0;
        skew(e, E);
        mat3_mul(T, E, M);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) J[(3 * k + i) * 6 + 3 + j] = -M[i * 3 + j];
    }
    /* rows 9..11, cols 0..2: R */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) J[(9 + i) * 6 + j] = T[i * 3 + j];
}

/* ======================================================================================
 *  Error terms (errorTerms.cpp) and robust kernels (robust_kernels.h)
 * ====================================================================================== */
static void fill_J2(double lx, double ly, double lz, double J[36])
{
    /* [lx*I ly*I lz*I I]   errorTerms.cpp:54-59 */
    memset(J, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
    {
        J[i * 12 + 0 + i] = lx;
        J[i * 12 + 3 + i] = ly;
        J[i * 12 + 6 + i] = lz;
        J[i * 12 + 9 + i] = 1.0;
    }
}

/* errorTerms.cpp:36-66 */
void orc_error_point2point(const orc_pair_pt2pt* p, const double T[12], double e[3],
                           double J1[36])
{
    double g[3];
    orc_pose_compose_point(T, p->lx, p->ly, p->lz, g);
    e[0] = g[0] - (double)p->gx;
    e[1] = g[1] - (double)p->gy;
    e[2] = g[2] - (double)p->gz;
    if (J1) fill_J2(p->lx, p->ly, p->lz, J1);
}

/* errorTerms.cpp:115-161 */
void orc_error_point2plane(const orc_pair_pt2pl* p, const double T[12], double e[3],
                           double J1[36])
{
    double g[3];
    orc_pose_compose_point(T, p->lx, p->ly, p->lz, g);
    const double* c     = p->plane;
    const double  mod_n = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    const double  s     = c[0] * g[0] + c[1] * g[1] + c[2] * g[2] + c[3];
    e[0] = -(c[0] / mod_n) * s;
    e[1] = -(c[1] / mod_n) * s;
    e[2] = -(c[2] / mod_n) * s;
    if (J1)
    {
        double A[9], J2[36];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) A[i * 3 + j] = -c[i] * c[j] / mod_n;
        fill_J2(p->lx, p->ly, p->lz, J2);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 12; j++)
                J1[i * 12 + j] =
                    A[i * 3 + 0] * J2[0 * 12 + j] + A[i * 3 + 1] * J2[1 * 12 + j] + A[i * 3 + 2] * J2[2 * 12 + j];
    }
}

/* errorTerms.cpp:68-113 */
void orc_error_point2line(const orc_pair_pt2ln* p, const double T[12], double e[3],
                          double J1[36])
{
    double g[3];
    orc_pose_compose_point(T, p->lx, p->ly, p->lz, g);
    const double* u  = p->director;
    const double  q[3] = {g[0] - p->pbase[0], g[1] - p->pbase[1], g[2] - p->pbase[2]};
    const double  uq   = u[0] * q[0] + u[1] * q[1] + u[2] * q[2];
    e[0] = q[0] - u[0] * uq;
    e[1] = q[1] - u[1] * uq;
    e[2] = q[2] - u[2] * uq;
    if (J1)
    {
        double A[9], J2[36];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) A[i * 3 + j] = (i == j ? 1.0 : 0.0) - u[i] * u[j];
        fill_J2(p->lx, p->ly, p->lz, J2);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 12; j++)
                J1[i * 12 + j] =
                    A[i * 3 + 0] * J2[0 * 12 + j] + A[i * 3 + 1] * J2[1 * 12 + j] + A[i * 3 + 2] * J2[2 * 12 + j];
    }
}

/* errorTerms.cpp:325-363: e = R*n_local - n_global (normals = the first three TPlane coefs, not
 * re-normalised); J1 = d(R n_l)/d[R t] with the translation block zero */
void orc_error_plane2plane(const orc_pair_pl2pl* p, const double T[12], double e[3], double J1[36])
{
    const double* nl = p->pl_local;
    const double* ng = p->pl_global;
    for (int i = 0; i < 3; i++)
        e[i] = T[i * 3 + 0] * nl[0] + T[i * 3 + 1] * nl[1] + T[i * 3 + 2] * nl[2] - ng[i]; /* rotateVector */
    if (J1)
    {
        memset(J1, 0, 36 * sizeof(double));
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < 3; c++) J1[i * 12 + c * 3 + i] = nl[c]; /* :350-354 */
    }
}

/* robust_kernels.h:57-94: weight functor on the SQUARED error */
double orc_robust_weight(int32_t kernel, double c, double errSq)
{
    const double c2 = c * c;
    switch (kernel)
    {
        case ORC_KERNEL_GEMANMCCLURE:
            return c2 / ((errSq + c) * (errSq + c)); /* :76-77 (denominator uses c, not c^2) */
        case ORC_KERNEL_CAUCHY:
            return c2 / (errSq + c2); /* :88-89 */
        default:
            return 1.0;
    }
}

/* ======================================================================================
 *  a3: transform_local_to_global  (Matcher_Points_Base.cpp:183-249, all-points branch)
 * ====================================================================================== */
void orc_transform_local_to_global(const float* lx, const float* ly, const float* lz, size_t n,
                                   const double T[12], float* ox, float* oy, float* oz,
                                   float bmin[3], float bmax[3])
{
    /* TransformedLocalPointCloud initialises localMin=+max, localMax=-max
     * (Matcher_Points_Base.h:98-112) */
    bmin[0] = bmin[1] = bmin[2] = FLT_MAX;
    bmax[0] = bmax[1] = bmax[2] = -FLT_MAX;
    for (size_t i = 0; i < n; i++)
    {
        double g[3];
        orc_pose_compose_point(T, lx[i], ly[i], lz[i], g); /* fp64, :216 */
        const float x = (float)g[0], y = (float)g[1], z = (float)g[2]; /* narrowed once, :217 */
        ox[i] = x, oy[i] = y, oz[i] = z;
        if (x > bmax[0]) bmax[0] = x;
        if (y > bmax[1]) bmax[1] = y;
        if (z > bmax[2]) bmax[2] = z;
        if (x < bmin[0]) bmin[0] = x;
        if (y < bmin[1]) bmin[1] = y;
        if (z < bmin[2]) bmin[2] = z;
    }
}

/* ======================================================================================
 *  a5: exact k nearest neighbours.  fp32 metric, accumulation order x,y,z
 *  (SURVEY.md Appendix C.4); tie winner = lowest index (this repo's stated policy).
 * ====================================================================================== */
static inline float dist2f(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

/* sorted insertion into an ascending (d2, idx) list of capacity k */
static inline void knn_insert(uint32_t* idx, float* d2, int* count, int k, uint32_t i, float d)
{
    int pos = *count;
    if (pos == k)
    {
        /* full: reject if not better than the worst */
        if (d > d2[k - 1] || (d == d2[k - 1] && i > idx[k - 1])) return;
        pos = k - 1;
    }
    else
        (*count)++;
    while (pos > 0 && (d2[pos - 1] > d || (d2[pos - 1] == d && idx[pos - 1] > i)))
    {
        d2[pos]  = d2[pos - 1];
        idx[pos] = idx[pos - 1];
        pos--;
    }
    d2[pos]  = d;
    idx[pos] = i;
}

int orc_brute_knn(const float* x, const float* y, const float* z, size_t n, float qx, float qy,
                  float qz, int k, float max_d2, uint32_t* out_idx, float* out_d2)
{
    int count = 0;
    for (size_t i = 0; i < n; i++)
    {
        const float d = dist2f(qx, qy, qz, x[i], y[i], z[i]);
        if (max_d2 >= 0 && !(d < max_d2)) continue;
        knn_insert(out_idx, out_d2, &count, k, (uint32_t)i, d);
    }
    return count;
}

/* ---- KD-tree (nanoflann-style: widest-dimension midpoint split, leaf buckets) ---------- */
typedef struct
{
    int32_t  left, right; /* children (node ids) or -1 for leaf */
    int32_t  dim;
    float    split_lo, split_hi; /* max of left side / min of right side along dim */
    uint32_t begin, end;         /* leaf: range in perm */
} kd_node;

struct orc_kdtree
{
    const float *x, *y, *z;
    size_t       n;
    uint32_t*    perm;
    kd_node*     nodes;
    size_t       n_nodes, cap_nodes;
    int          leaf_max;
    float        bmin[3], bmax[3];
    /* leaf-ordered copies for cache-friendly scans */
    float *      px, *py, *pz;
    /* multi-thread baseline (orc_match_pt2pt_mt_ms): the layer's bounding box, computed once with the tree (the reference's
     * point maps cache theirs), and the "global point taken" bytes of the unique-global filter, kept all-zero between calls
     * (a call clears the entries it set: no 10 MB calloc per iteration) */
    float        gmin[3], gmax[3];
    uint32_t*    mt_owner; /* [n]: lowest claiming local index per global point, 0xFFFFFFFF between calls */
};

static int32_t kd_new_node(orc_kdtree* t)
{
    if (t->n_nodes == t->cap_nodes)
    {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes     = (kd_node*)realloc(t->nodes, t->cap_nodes * sizeof(kd_node));
    }
    return (int32_t)t->n_nodes++;
}

static inline float kd_coord(const orc_kdtree* t, uint32_t i, int d)
{
    return d == 0 ? t->x[i] : (d == 1 ? t->y[i] : t->z[i]);
}

static int32_t kd_build_rec(orc_kdtree* t, uint32_t b, uint32_t e)
{
    const int32_t id = kd_new_node(t);
    if ((int)(e - b) <= t->leaf_max)
    {
        kd_node* nd = &t->nodes[id];
        nd->left = nd->right = -1;
        nd->begin = b, nd->end = e;
        nd->dim = 0, nd->split_lo = nd->split_hi = 0;
        return id;
    }
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = b; i < e; i++)
        for (int d = 0; d < 3; d++)
        {
            const float v = kd_coord(t, t->perm[i], d);
            if (v < lo[d]) lo[d] = v;
            if (v > hi[d]) hi[d] = v;
        }
    int dim = 0;
    if (hi[1] - lo[1] > hi[dim] - lo[dim]) dim = 1;
    if (hi[2] - lo[2] > hi[dim] - lo[dim]) dim = 2;
    const float mid = 0.5f * (lo[dim] + hi[dim]);
    /* partition: < mid left, >= mid right; balance degenerate cases */
    uint32_t i = b, j = e;
    while (i < j)
    {
        if (kd_coord(t, t->perm[i], dim) < mid)
            i++;
        else
        {
            j--;
            uint32_t tmp = t->perm[i];
            t->perm[i]   = t->perm[j];
            t->perm[j]   = tmp;
        }
    }
    uint32_t m = i;
    if (m == b || m == e)
    { /* all equal along dim (or all on one side): split by count */
        m = b + (e - b) / 2;
    }
    float slo = -FLT_MAX, shi = FLT_MAX;
    for (uint32_t q = b; q < m; q++)
    {
        const float v = kd_coord(t, t->perm[q], dim);
        if (v > slo) slo = v;
    }
    for (uint32_t q = m; q < e; q++)
    {
        const float v = kd_coord(t, t->perm[q], dim);
        if (v < shi) shi = v;
    }
    const int32_t l = kd_build_rec(t, b, m);
    const int32_t r = kd_build_rec(t, m, e);
    kd_node*      nd = &t->nodes[id];
    nd->left = l, nd->right = r, nd->dim = dim;
    nd->split_lo = slo, nd->split_hi = shi;
    nd->begin = b, nd->end = e;
    return id;
}

orc_kdtree* orc_kdtree_build(const float* x, const float* y, const float* z, size_t n,
                             int leaf_max)
{
    orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof(orc_kdtree));
    t->x = x, t->y = y, t->z = z, t->n = n;
    t->leaf_max = leaf_max > 0 ? leaf_max : 10;
    t->perm     = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) t->perm[i] = (uint32_t)i;
    if (n) kd_build_rec(t, 0, (uint32_t)n);
    t->px = (float*)malloc((n ? n : 1) * sizeof(float));
    t->py = (float*)malloc((n ? n : 1) * sizeof(float));
    t->pz = (float*)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; i++)
    {
        t->px[i] = x[t->perm[i]];
        t->py[i] = y[t->perm[i]];
        t->pz[i] = z[t->perm[i]];
    }
    t->gmin[0] = t->gmin[1] = t->gmin[2] = FLT_MAX;
    t->gmax[0] = t->gmax[1] = t->gmax[2] = -FLT_MAX;
    for (size_t i = 0; i < n; i++)
    { /* (the comparisons of bbox_of, in its order) */
        if (x[i] < t->gmin[0]) t->gmin[0] = x[i];
        if (y[i] < t->gmin[1]) t->gmin[1] = y[i];
        if (z[i] < t->gmin[2]) t->gmin[2] = z[i];
        if (x[i] > t->gmax[0]) t->gmax[0] = x[i];
        if (y[i] > t->gmax[1]) t->gmax[1] = y[i];
        if (z[i] > t->gmax[2]) t->gmax[2] = z[i];
    }
    t->mt_owner = NULL;
    return t;
}

void orc_kdtree_free(orc_kdtree* t)
{
    if (!t) return;
    free(t->perm);
    free(t->nodes);
    free(t->px);
    free(t->py);
    free(t->pz);
    free(t->mt_owner);
    free(t);
}

typedef struct
{
    const orc_kdtree* t;
    float             q[3];
    int               k, count;
    float             max_d2;
    uint32_t*         idx;
    float*            d2;
} kd_query;

static inline double kd_worst(const kd_query* s)
{
    if (s->count == s->k) return (double)s->d2[s->k - 1];
    return s->max_d2 >= 0 ? (double)s->max_d2 : DBL_MAX;
}

static void kd_search_rec(kd_query* s, int32_t id, double mind2, double off[3])
{
    const kd_node* nd = &s->t->nodes[id];
    if (nd->left < 0)
    {
        const orc_kdtree* t = s->t;
        for (uint32_t i = nd->begin; i < nd->end; i++)
        {
            const float d = dist2f(s->q[0], s->q[1], s->q[2], t->px[i], t->py[i], t->pz[i]);
            if (s->max_d2 >= 0 && !(d < s->max_d2)) continue;
            knn_insert(s->idx, s->d2, &s->count, s->k, t->perm[i], d);
        }
        return;
    }
    const int    dim = nd->dim;
    const double v   = (double)s->q[dim];
    /* distance (along dim) to each child's slab */
    const double dl = v > (double)nd->split_lo ? v - (double)nd->split_lo : 0.0;
    const double dr = v < (double)nd->split_hi ? (double)nd->split_hi - v : 0.0;
    int32_t      first, second;
    double       dsecond;
    if (dl <= dr)
        first = nd->left, second = nd->right, dsecond = dr;
    else
        first = nd->right, second = nd->left, dsecond = dl;
    kd_search_rec(s, first, mind2, off);
    const double old   = off[dim];
    const double mind2b = mind2 - old * old + dsecond * dsecond;
    /* conservative, non-strict pruning: computed fp32 distances may be a few ulp below the
     * real ones, and ties must still be visited for the lowest-index rule */
    if (mind2b * (1.0 - 1e-6) <= kd_worst(s))
    {
        off[dim] = dsecond > old ? dsecond : old;
        const double m2 = mind2 - old * old + off[dim] * off[dim];
        kd_search_rec(s, second, m2, off);
        off[dim] = old;
    }
}

int orc_kdtree_knn(const orc_kdtree* t, float qx, float qy, float qz, int k, float max_d2,
                   uint32_t* out_idx, float* out_d2)
{
    if (!t || t->n == 0 || k <= 0) return 0;
    kd_query s;
    s.t = t, s.q[0] = qx, s.q[1] = qy, s.q[2] = qz;
    s.k = k, s.count = 0, s.max_d2 = max_d2, s.idx = out_idx, s.d2 = out_d2;
    double off[3] = {0, 0, 0};
    kd_search_rec(&s, 0, 0.0, off);
    return s.count;
}

static int nn_search(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                     size_t n_g, float qx, float qy, float qz, int k, float max_d2,
                     uint32_t* idx, float* d2)
{
    if (tree) return orc_kdtree_knn(tree, qx, qy, qz, k, max_d2, idx, d2);
    return orc_brute_knn(gx, gy, gz, n_g, qx, qy, qz, k, max_d2, idx, d2);
}

static void bbox_of(const float* x, const float* y, const float* z, size_t n, float mn[3],
                    float mx[3])
{
    mn[0] = mn[1] = mn[2] = FLT_MAX;
    mx[0] = mx[1] = mx[2] = -FLT_MAX;
    for (size_t i = 0; i < n; i++)
    {
        if (x[i] < mn[0]) mn[0] = x[i];
        if (y[i] < mn[1]) mn[1] = y[i];
        if (z[i] < mn[2]) mn[2] = z[i];
        if (x[i] > mx[0]) mx[0] = x[i];
        if (y[i] > mx[1]) mx[1] = y[i];
        if (z[i] > mx[2]) mx[2] = z[i];
    }
}

/* TBoundingBoxf::intersection(other, eps).has_value() (Matcher_Points_DistanceThreshold.cpp:
 * 73-75).  MRPT (un-vendored) returns no intersection when `other`, inflated by eps, lies
 * strictly beyond `this` on some axis:
 *     b.min - eps > max  ||  b.max + eps < min
 * (a = global layer box = "this", b = transformed local box).  With thresholdAngularDeg == 0
 * the early-out cannot change the result: eps = threshold + 0.2 exceeds any accepted pair
 * distance.  SURVEY.md Appendix C.6 lists this as an assumption to re-check against MRPT. */
static int bbox_intersects(const float amin[3], const float amax[3], const float bmin[3],
                           const float bmax[3], float eps)
{
    for (int d = 0; d < 3; d++)
    {
        if (bmin[d] - eps > amax[d]) return 0;
        if (bmax[d] + eps < amin[d]) return 0;
    }
    return 1;
}

/* ======================================================================================
 *  a4: Matcher_Points_DistanceThreshold::implMatchOneLayer
 *      (Matcher_Points_DistanceThreshold.cpp:48-121, sequential branch :206-266)
 * ====================================================================================== */
#define ORC_MAX_K 64

size_t orc_match_pt2pt(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                       const double T[12], const orc_pt2pt_params* prm, uint8_t* local_taken,
                       uint8_t* global_taken, orc_pair_pt2pt* out, uint64_t* potential_pairings)
{
    const int K = (int)prm->pairingsPerPoint;
    if (potential_pairings) *potential_pairings += (uint64_t)n_l * prm->pairingsPerPoint; /* :64 */
    if (n_g == 0 || n_l == 0 || K < 1 || K > ORC_MAX_K) return 0;                         /* :67 */

    float* tx = (float*)malloc(n_l * sizeof(float));
    float* ty = (float*)malloc(n_l * sizeof(float));
    float* tz = (float*)malloc(n_l * sizeof(float));
    float  lmin[3], lmax[3], gmin[3], gmax[3];
    orc_transform_local_to_global(lx, ly, lz, n_l, T, tx, ty, tz, lmin, lmax); /* :69-70 */
    bbox_of(gx, gy, gz, n_g, gmin, gmax);

    size_t   n_out     = 0;
    uint8_t* own_taken = NULL; /* MatchState always exists in the reference (Matcher.h:44-70) */
    if (!global_taken && !prm->allowMatchAlreadyMatchedGlobalPoints)
        global_taken = own_taken = (uint8_t*)calloc(n_g, 1);
    /* :73-75 */
    if (bbox_intersects(gmin, gmax, lmin, lmax, (float)(prm->threshold + prm->bbox_eps)))
    {
        /* :82-83  mrpt::square(double) narrowed to float */
        const float  maxDistSq = (float)(prm->threshold * prm->threshold);
        const double angRad    = prm->thresholdAngularDeg * M_PI / 180.0;
        const float  angSq     = (float)(angRad * angRad);

        uint32_t nidx[ORC_MAX_K];
        float    nd2[ORC_MAX_K];
        for (size_t i = 0; i < n_l; i++) /* :214 */
        {
            if (!prm->allowMatchAlreadyMatchedPoints && local_taken && local_taken[i])
                continue; /* :218-220 */
            const float x = tx[i], y = ty[i], z = tz[i];
            const float normSq = (x * x + y * y) + z * z; /* :223-225 */
            int         found;
            if (K == 1)
                found = nn_search(tree, gx, gy, gz, n_g, x, y, z, 1, -1.0f, nidx, nd2); /* :235 */
            else if (prm->multi_search_radius_mode)
                found = nn_search(tree, gx, gy, gz, n_g, x, y, z, K, maxDistSq, nidx, nd2); /* :174 */
            else
                found = nn_search(tree, gx, gy, gz, n_g, x, y, z, K, -1.0f, nidx, nd2); /* :246 */

            for (int k = 0; k < found; k++) /* :252 */
            {
                const float thr = maxDistSq + angSq * normSq; /* :256-257 */
                if (nd2[k] >= thr) break;                     /* :259 */
                /* lambdaAddPair :94-121 */
                const uint32_t g = nidx[k];
                if (!prm->allowMatchAlreadyMatchedGlobalPoints && global_taken &&
                    global_taken[g])
                    continue;
                orc_pair_pt2pt* p = &out[n_out++];
                p->globalIdx = g, p->localIdx = (uint32_t)i;
                p->gx = gx[g], p->gy = gy[g], p->gz = gz[g];
                p->lx = lx[i], p->ly = ly[i], p->lz = lz[i]; /* untransformed */
                p->errSq = nd2[k];
                if (!prm->allowMatchAlreadyMatchedGlobalPoints)
                {
                    if (local_taken) local_taken[i] = 1;
                    if (global_taken) global_taken[g] = 1;
                }
            }
        }
    }
    free(tx), free(ty), free(tz), free(own_taken);
    return n_out;
}

/* ---- multi-threaded CPU baseline of the same contract (K==1) --------------------------------
 * Round 5 (the baseline must be a competently parallel port, not a flattering one): a PERSISTENT thread pool
 * (round 4 created and joined 2 x n_threads threads per call), the global layer's bounding box cached with the tree, the
 * unique-global filter resolved by all threads (atomic minimum of the claiming local index), the pair records gathered in
 * parallel, no per-call allocation that scales with the map.  Same lists as the sequential loop. */
typedef void (*pool_fn)(void* jobs, int t);
static struct
{
    pthread_t       th[1024];
    int             n;          /* threads alive */
    pthread_mutex_t mu;
    pthread_cond_t  cv_go, cv_done;
    unsigned long   gen;        /* incremented per run */
    int             n_run, n_left;
    pool_fn         fn;
    void*           jobs;
    int             inited;
} g_pool;

static void* pool_main(void* arg)
{
    const int     me   = (int)(intptr_t)arg;
    unsigned long seen = 0;
    for (;;)
    {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.cv_go, &g_pool.mu);
        seen             = g_pool.gen;
        const int     n  = g_pool.n_run;
        const pool_fn fn = g_pool.fn;
        void*         jb = g_pool.jobs;
        pthread_mutex_unlock(&g_pool.mu);
        if (me < n) fn(jb, me);
        pthread_mutex_lock(&g_pool.mu);
        if (me < n && --g_pool.n_left == 0) pthread_cond_signal(&g_pool.cv_done);
        pthread_mutex_unlock(&g_pool.mu);
    }
    return NULL;
}

/* runs fn(jobs, t) for t = 0 .. n - 1 on the pool's threads and waits (one caller at a time: the baseline's use) */
static void pool_run(pool_fn fn, void* jobs, int n)
{
    if (n <= 1)
    {
        fn(jobs, 0);
        return;
    }
    if (!g_pool.inited)
    {
        pthread_mutex_init(&g_pool.mu, NULL);
        pthread_cond_init(&g_pool.cv_go, NULL);
        pthread_cond_init(&g_pool.cv_done, NULL);
        g_pool.inited = 1;
    }
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.n < n && g_pool.n < 1024)
    {
        pthread_create(&g_pool.th[g_pool.n], NULL, pool_main, (void*)(intptr_t)g_pool.n);
        g_pool.n++;
    }
    g_pool.fn = fn, g_pool.jobs = jobs, g_pool.n_run = n, g_pool.n_left = n;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.cv_go);
    while (g_pool.n_left > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}

typedef struct
{
    const orc_kdtree* tree;
    const float *     lx, *ly, *lz;
    const double*     T;
    float *           tx, *ty, *tz;
    size_t            n_l;
    int               n_threads;
    float             maxDistSq, angSq;
    uint32_t*         nn_idx;
    float*            nn_d2;
    const uint8_t*    skip; /* local points not to search, or NULL */
    float (*bb)[6];         /* per-thread bounding box of the transformed points */
    /* claims + gather */
    const float *     gx, *gy, *gz;
    uint8_t*          win;
    size_t*           chunk_off;  /* [n_threads + 1]: winners per chunk, then their exclusive prefix */
    orc_pair_pt2pt*   out;
    uint32_t*         owner;      /* NULL: global points may be paired again (no filter) */
    const uint8_t*    gtaken;     /* pre-marked global points (lose), or NULL */
    uint8_t*          gmark;      /* marks to leave on the global / local layer, or NULL */
    uint8_t*          lmark;
} mt_ctx;

static void mt_transform(void* jobs, int t)
{
    mt_ctx*      j = (mt_ctx*)jobs;
    const size_t b = j->n_l * (size_t)t / j->n_threads, e = j->n_l * (size_t)(t + 1) / j->n_threads;
    float        mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = b; i < e; i++)
    {
        double g[3];
        orc_pose_compose_point(j->T, j->lx[i], j->ly[i], j->lz[i], g);
        const float x = (float)g[0], y = (float)g[1], z = (float)g[2];
        j->tx[i] = x, j->ty[i] = y, j->tz[i] = z;
        if (x < mn[0]) mn[0] = x;
        if (y < mn[1]) mn[1] = y;
        if (z < mn[2]) mn[2] = z;
        if (x > mx[0]) mx[0] = x;
        if (y > mx[1]) mx[1] = y;
        if (z > mx[2]) mx[2] = z;
    }
    for (int d = 0; d < 3; d++) j->bb[t][d] = mn[d], j->bb[t][3 + d] = mx[d];
}

static void mt_search(void* jobs, int t)
{
    mt_ctx*      j = (mt_ctx*)jobs;
    const size_t b = j->n_l * (size_t)t / j->n_threads, e = j->n_l * (size_t)(t + 1) / j->n_threads;
    for (size_t i = b; i < e; i++)
    {
        if (j->skip && j->skip[i])
        {
            j->nn_idx[i] = 0xFFFFFFFFu, j->nn_d2[i] = 0;
            continue;
        }
        const float x = j->tx[i], y = j->ty[i], z = j->tz[i];
        const float normSq = (x * x + y * y) + z * z;
        uint32_t    id;
        float       d2;
        const int   f   = orc_kdtree_knn(j->tree, x, y, z, 1, -1.0f, &id, &d2);
        const float thr = j->maxDistSq + j->angSq * normSq;
        if (f && d2 < thr)
            j->nn_idx[i] = id, j->nn_d2[i] = d2;
        else
            j->nn_idx[i] = 0xFFFFFFFFu, j->nn_d2[i] = 0;
    }
}

/* The unique-global filter in parallel.  The sequential loop gives a contested global point to the FIRST local point that
 * reaches it (a claimant that loses has no side effects, Matcher_Points_DistanceThreshold.cpp:94-121): winner(g) = the lowest
 * claiming local index.  Pass 1: atomic minimum of the local index per global point; pass 2: a query wins iff it is that
 * minimum; pass 3 (after the gather): the touched entries go back to "none".  Random accesses to a 40 MB array from every
 * thread instead of 2 x n_l of them from one (which WAS the multi-thread baseline's floor: 30 of its 34 ms). */
static void mt_claim(void* jobs, int t)
{
    mt_ctx*      j = (mt_ctx*)jobs;
    const size_t b = j->n_l * (size_t)t / j->n_threads, e = j->n_l * (size_t)(t + 1) / j->n_threads;
    for (size_t i = b; i < e; i++)
    {
        const uint32_t g = j->nn_idx[i];
        if (g == 0xFFFFFFFFu || (j->gtaken && j->gtaken[g])) continue;
        uint32_t cur = __atomic_load_n(&j->owner[g], __ATOMIC_RELAXED);
        while ((uint32_t)i < cur && !__atomic_compare_exchange_n(&j->owner[g], &cur, (uint32_t)i, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
}

static void mt_winners(void* jobs, int t)
{
    mt_ctx*      j = (mt_ctx*)jobs;
    const size_t b = j->n_l * (size_t)t / j->n_threads, e = j->n_l * (size_t)(t + 1) / j->n_threads;
    size_t       n = 0;
    for (size_t i = b; i < e; i++)
    {
        const uint32_t g = j->nn_idx[i];
        uint8_t        w = 0;
        if (g != 0xFFFFFFFFu)
        {
            if (!j->owner) w = 1;
            else if (!(j->gtaken && j->gtaken[g]) && __atomic_load_n(&j->owner[g], __ATOMIC_RELAXED) == (uint32_t)i) w = 1;
        }
        j->win[i] = w;
        n += w;
    }
    j->chunk_off[t + 1] = n;
}

static void mt_release(void* jobs, int t)
{
    mt_ctx*      j = (mt_ctx*)jobs;
    const size_t b = j->n_l * (size_t)t / j->n_threads, e = j->n_l * (size_t)(t + 1) / j->n_threads;
    for (size_t i = b; i < e; i++)
    {
        if (!j->win[i]) continue;
        const uint32_t g = j->nn_idx[i];
        j->owner[g] = 0xFFFFFFFFu;
        if (j->gmark) j->gmark[g] = 1;
        if (j->lmark) j->lmark[i] = 1;
    }
}

static void mt_gather(void* jobs, int t)
{
    mt_ctx*      j = (mt_ctx*)jobs;
    const size_t b = j->n_l * (size_t)t / j->n_threads, e = j->n_l * (size_t)(t + 1) / j->n_threads;
    size_t       o = j->chunk_off[t];
    for (size_t i = b; i < e; i++)
    {
        if (!j->win[i]) continue;
        const uint32_t  g = j->nn_idx[i];
        orc_pair_pt2pt* p = &j->out[o++];
        p->globalIdx = g, p->localIdx = (uint32_t)i;
        p->gx = j->gx[g], p->gy = j->gy[g], p->gz = j->gz[g];
        p->lx = j->lx[i], p->ly = j->ly[i], p->lz = j->lz[i];
        p->errSq = j->nn_d2[i];
    }
}

size_t orc_match_pt2pt_mt(const orc_kdtree* tree, const float* gx, const float* gy,
                          const float* gz, size_t n_g, const float* lx, const float* ly,
                          const float* lz, size_t n_l, const double T[12],
                          const orc_pt2pt_params* prm, orc_pair_pt2pt* out, int n_threads)
{
    return orc_match_pt2pt_mt_ms(tree, gx, gy, gz, n_g, lx, ly, lz, n_l, T, prm, NULL, NULL, out, n_threads);
}

/* ... with a MatchState: local points already paired are not searched (:218-220), pre-marked global
 * points lose (:98-101), and the marks of the emitted pairs are left (:116-120) */
size_t orc_match_pt2pt_mt_ms(const orc_kdtree* tree, const float* gx, const float* gy,
                             const float* gz, size_t n_g, const float* lx, const float* ly,
                             const float* lz, size_t n_l, const double T[12],
                             const orc_pt2pt_params* prm, uint8_t* local_taken, uint8_t* global_taken,
                             orc_pair_pt2pt* out, int n_threads)
{
    if (!tree || n_g == 0 || n_l == 0 || prm->pairingsPerPoint != 1) return 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    if ((size_t)n_threads > n_l) n_threads = (int)n_l;
    float*    tx  = (float*)malloc(n_l * sizeof(float));
    float*    ty  = (float*)malloc(n_l * sizeof(float));
    float*    tz  = (float*)malloc(n_l * sizeof(float));
    uint32_t* nn  = (uint32_t*)malloc(n_l * sizeof(uint32_t));
    float*    nd  = (float*)malloc(n_l * sizeof(float));
    uint8_t*  win = (uint8_t*)malloc(n_l);
    float (*bb)[6] = (float (*)[6])malloc(sizeof(float[6]) * (size_t)n_threads);
    size_t*   coff = (size_t*)malloc(sizeof(size_t) * ((size_t)n_threads + 1));
    mt_ctx    c;
    memset(&c, 0, sizeof(c));
    c.tree = tree, c.lx = lx, c.ly = ly, c.lz = lz, c.T = T, c.tx = tx, c.ty = ty, c.tz = tz, c.n_l = n_l, c.n_threads = n_threads;
    c.nn_idx = nn, c.nn_d2 = nd, c.bb = bb, c.gx = gx, c.gy = gy, c.gz = gz, c.win = win, c.chunk_off = coff, c.out = out;
    pool_run(mt_transform, &c, n_threads);
    float lmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int t = 0; t < n_threads; t++)
        for (int d = 0; d < 3; d++)
        {
            if (bb[t][d] < lmin[d]) lmin[d] = bb[t][d];
            if (bb[t][3 + d] > lmax[d]) lmax[d] = bb[t][3 + d];
        }
    size_t n_out = 0;
    /* (the tree's cached box is the box of the points it was built on: the layer the caller passes) */
    if (bbox_intersects(tree->gmin, tree->gmax, lmin, lmax, (float)(prm->threshold + prm->bbox_eps)))
    {
        const double angRad = prm->thresholdAngularDeg * M_PI / 180.0;
        c.maxDistSq = (float)(prm->threshold * prm->threshold);
        c.angSq     = (float)(angRad * angRad);
        c.skip      = prm->allowMatchAlreadyMatchedPoints ? NULL : local_taken;
        pool_run(mt_search, &c, n_threads);
        if (!prm->allowMatchAlreadyMatchedGlobalPoints)
        {
            orc_kdtree* tw = (orc_kdtree*)tree; /* (scratch of the handle, not part of its value) */
            if (!tw->mt_owner)
            {
                tw->mt_owner = (uint32_t*)malloc(n_g * sizeof(uint32_t));
                memset(tw->mt_owner, 0xFF, n_g * sizeof(uint32_t));
            }
            c.owner = tw->mt_owner, c.gtaken = global_taken, c.gmark = global_taken, c.lmark = local_taken;
            pool_run(mt_claim, &c, n_threads);
        }
        coff[0] = 0;
        pool_run(mt_winners, &c, n_threads);
        for (int t = 0; t < n_threads; t++) coff[t + 1] += coff[t];
        n_out = coff[n_threads];
        pool_run(mt_gather, &c, n_threads);
        if (c.owner) pool_run(mt_release, &c, n_threads); /* the scratch back to "none"; the MatchState marks of the emitted pairs */
    }
    free(tx), free(ty), free(tz), free(nn), free(nd), free(win), free(bb), free(coff);
    return n_out;
}

/* ======================================================================================
 *  estimate_points_eigen (estimate_points_eigen.cpp:27-123, totalCount branch)
 * ====================================================================================== */
void orc_estimate_points_eigen(const float* xs, const float* ys, const float* zs, size_t n,
                               float mean[3], double cov[9], double eval[3], double evec[9])
{
    float       mx = 0, my = 0, mz = 0;
    const float inv_n = 1.0f / (float)n; /* :45 */
    for (size_t i = 0; i < n; i++) mx += xs[i], my += ys[i], mz += zs[i]; /* :46-51, fp32 */
    mx *= inv_n, my *= inv_n, mz *= inv_n;                                 /* :52 */
    double a00 = 0, a10 = 0, a20 = 0, a11 = 0, a21 = 0, a22 = 0;
    for (size_t i = 0; i < n; i++)
    {
        /* TPoint3Df - TPoint3Df: fp32 differences; products formed in fp32, then
         * accumulated into the double matrix (:55-61) */
        const float ax = xs[i] - mx, ay = ys[i] - my, az = zs[i] - mz;
        a00 += (double)(ax * ax);
        a10 += (double)(ax * ay);
        a20 += (double)(ax * az);
        a11 += (double)(ay * ay);
        a21 += (double)(ay * az);
        a22 += (double)(az * az);
    }
    const double s = (double)inv_n; /* mat_a *= inv_n (:63) */
    a00 *= s, a10 *= s, a20 *= s, a11 *= s, a21 *= s, a22 *= s;
    cov[0] = a00, cov[1] = a10, cov[2] = a20;
    cov[3] = a10, cov[4] = a11, cov[5] = a21;
    cov[6] = a20, cov[7] = a21, cov[8] = a22;
    mean[0] = mx, mean[1] = my, mean[2] = mz;
    sym_eig_jacobi(3, cov, eval, evec); /* :108-117 */
}

/* ======================================================================================
 *  a6: Matcher_Point2Plane::implMatchOneLayer (Matcher_Point2Plane.cpp:41-114) with this
 *  repo's DECLARED nn_search_pt2pl (parity unpinned):
 *    k nearest global points with d2 <= searchRadius^2 (filter as Matcher_Point2Line.cpp:
 *    113-129); need >= minimumPlanePoints; estimate_points_eigen over them in ascending
 *    (d2, idx) order; planar iff e0 < thr*e1 && e0 < thr*e2 (Matcher_Adaptive.cpp:241-242);
 *    plane = TPlane(centroid, eigvec0) normalised so that its largest |component| is
 *    positive; distance = |plane.distance(q)| as float.
 * ====================================================================================== */
/* one query of the loop :79-110: 1 = a pairing was produced (plane, centroid written to p) */
static int pt2pl_query(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const orc_pt2pl_params* prm, float x, float y, float z,
                       orc_pair_pt2pl* p)
{
    const int   K       = (int)prm->knn;
    const float radSq   = (float)(prm->searchRadius * prm->searchRadius);
    const float distThr = (float)prm->distanceThreshold;
    uint32_t    nidx[ORC_MAX_K];
    float       nd2[ORC_MAX_K], kx[ORC_MAX_K], ky[ORC_MAX_K], kz[ORC_MAX_K];
    int found = nn_search(tree, gx, gy, gz, n_g, x, y, z, K, -1.0f, nidx, nd2);
    int m     = 0;
    while (m < found && !(nd2[m] > radSq)) m++; /* keep d2 <= radius^2 */
    if (m < (int)prm->minimumPlanePoints || m < 3) return 0;
    for (int k = 0; k < m; k++) kx[k] = gx[nidx[k]], ky[k] = gy[nidx[k]], kz[k] = gz[nidx[k]];
    float  mean[3];
    double cov[9], ev[3], evec[9];
    orc_estimate_points_eigen(kx, ky, kz, (size_t)m, mean, cov, ev, evec);
    if (!(ev[0] < prm->planeEigenThreshold * ev[2] && ev[0] < prm->planeEigenThreshold * ev[1]))
        return 0;
    /* TPlane(point, normal): unit normal, d = -n.c */
    double       n[3] = {evec[0], evec[1], evec[2]};
    const double nn   = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= nn, n[1] /= nn, n[2] /= nn;
    int big = 0;
    if (fabs(n[1]) > fabs(n[big])) big = 1;
    if (fabs(n[2]) > fabs(n[big])) big = 2;
    if (n[big] < 0) n[0] = -n[0], n[1] = -n[1], n[2] = -n[2];
    const double c[3] = {(double)mean[0], (double)mean[1], (double)mean[2]};
    const double d    = -(n[0] * c[0] + n[1] * c[1] + n[2] * c[2]);
    const float  dist = (float)fabs(n[0] * (double)x + n[1] * (double)y + n[2] * (double)z + d);
    if (dist > distThr) return 0; /* :100-101 */
    p->plane[0] = n[0], p->plane[1] = n[1], p->plane[2] = n[2], p->plane[3] = d;
    p->centroid[0] = c[0], p->centroid[1] = c[1], p->centroid[2] = c[2];
    return 1;
}

size_t orc_match_pt2pl(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                       const double T[12], const orc_pt2pl_params* prm, uint8_t* local_taken,
                       orc_pair_pt2pl* out, uint32_t* out_local_idx,
                       uint64_t* potential_pairings)
{
    const int K = (int)prm->knn;
    if (potential_pairings) *potential_pairings += (uint64_t)n_l; /* :54 */
    if (n_g == 0 || n_l == 0 || K < 3 || K > ORC_MAX_K) return 0;

    float* tx = (float*)malloc(n_l * sizeof(float));
    float* ty = (float*)malloc(n_l * sizeof(float));
    float* tz = (float*)malloc(n_l * sizeof(float));
    float  lmin[3], lmax[3], gmin[3], gmax[3];
    orc_transform_local_to_global(lx, ly, lz, n_l, T, tx, ty, tz, lmin, lmax); /* :59-60 */
    bbox_of(gx, gy, gz, n_g, gmin, gmax);
    size_t n_out = 0;
    if (bbox_intersects(gmin, gmax, lmin, lmax,
                        (float)(prm->distanceThreshold + prm->bbox_eps))) /* :63-66 */
    {
        for (size_t i = 0; i < n_l; i++) /* :79 */
        {
            if (!prm->allowMatchAlreadyMatchedPoints && local_taken && local_taken[i])
                continue; /* :83-85 */
            orc_pair_pt2pl* p = &out[n_out];
            if (!pt2pl_query(tree, gx, gy, gz, n_g, prm, tx[i], ty[i], tz[i], p)) continue;
            p->lx = lx[i], p->ly = ly[i], p->lz = lz[i]; /* :104-106 untransformed */
            p->_pad = 0;
            if (out_local_idx) out_local_idx[n_out] = (uint32_t)i;
            n_out++;
            if (local_taken) local_taken[i] = 1; /* :109 */
        }
    }
    free(tx), free(ty), free(tz);
    return n_out;
}

/* ---- the same contract with the per-query work spread over threads (the queries of this matcher
 *      do not interact: no global uniqueness, :87-90); results are gathered in ascending local
 *      index, i.e. the sequential loop's order.  For the full-size configurations of the tests. */
typedef struct
{
    const orc_kdtree*       tree;
    const float *           gx, *gy, *gz;
    size_t                  n_g;
    const orc_pt2pl_params* prm;
    const float *           tx, *ty, *tz;
    const uint8_t*          local_taken;
    size_t                  b, e;
    uint8_t*                flag;
    orc_pair_pt2pl*         rec;
} pl_job;

static void* pl_worker(void* arg)
{
    pl_job* j = (pl_job*)arg;
    for (size_t i = j->b; i < j->e; i++)
    {
        j->flag[i] = 0;
        if (!j->prm->allowMatchAlreadyMatchedPoints && j->local_taken && j->local_taken[i]) continue;
        j->flag[i] = (uint8_t)pt2pl_query(j->tree, j->gx, j->gy, j->gz, j->n_g, j->prm, j->tx[i], j->ty[i],
                                          j->tz[i], &j->rec[i]);
    }
    return NULL;
}

size_t orc_match_pt2pl_mt(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                          size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                          const double T[12], const orc_pt2pl_params* prm, uint8_t* local_taken,
                          orc_pair_pt2pl* out, uint32_t* out_local_idx,
                          uint64_t* potential_pairings, int n_threads)
{
    const int K = (int)prm->knn;
    if (potential_pairings) *potential_pairings += (uint64_t)n_l; /* :54 */
    if (!tree || n_g == 0 || n_l == 0 || K < 3 || K > ORC_MAX_K) return 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    float* tx = (float*)malloc(n_l * sizeof(float));
    float* ty = (float*)malloc(n_l * sizeof(float));
    float* tz = (float*)malloc(n_l * sizeof(float));
    float  lmin[3], lmax[3], gmin[3], gmax[3];
    orc_transform_local_to_global(lx, ly, lz, n_l, T, tx, ty, tz, lmin, lmax);
    bbox_of(gx, gy, gz, n_g, gmin, gmax);
    size_t n_out = 0;
    if (bbox_intersects(gmin, gmax, lmin, lmax, (float)(prm->distanceThreshold + prm->bbox_eps)))
    {
        uint8_t*        flag = (uint8_t*)malloc(n_l);
        orc_pair_pt2pl* rec  = (orc_pair_pt2pl*)malloc(n_l * sizeof(orc_pair_pt2pl));
        pthread_t       th[1024];
        pl_job          jobs[1024];
        for (int t = 0; t < n_threads; t++)
        {
            jobs[t] = (pl_job){tree, gx, gy, gz, n_g, prm, tx, ty, tz, local_taken, n_l * t / n_threads,
                               n_l * (t + 1) / n_threads, flag, rec};
            pthread_create(&th[t], NULL, pl_worker, &jobs[t]);
        }
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
        for (size_t i = 0; i < n_l; i++)
        {
            if (!flag[i]) continue;
            orc_pair_pt2pl* p = &out[n_out];
            *p    = rec[i];
            p->lx = lx[i], p->ly = ly[i], p->lz = lz[i];
            p->_pad = 0;
            if (out_local_idx) out_local_idx[n_out] = (uint32_t)i;
            n_out++;
            if (local_taken) local_taken[i] = 1;
        }
        free(flag), free(rec);
    }
    free(tx), free(ty), free(tz);
    return n_out;
}

/* ---- maxLocalPointsPerLayer: the matchers visit a LIST of local points ------------------------
 * Matcher_Points_Base.cpp:222-246 transforms only x_locals[ri] = T * l[idxs[ri]] (the bounding
 * box is theirs alone) and both matchers then run their loop over ri with
 * localIdx = idxs[ri] (Matcher_Points_DistanceThreshold.cpp:216, Matcher_Point2Plane.cpp:81).
 * That is the plain matcher on the gathered cloud with indices and MatchState marks mapped back.
 * The list itself comes from mrpt::random::partial_shuffle (un-vendored MRPT): an input here. */
size_t orc_match_pt2pt_subset(const orc_kdtree* tree, const float* gx, const float* gy,
                              const float* gz, size_t n_g, const float* lx, const float* ly,
                              const float* lz, size_t n_l, const uint32_t* idxs, size_t n_idxs,
                              const double T[12], const orc_pt2pt_params* prm,
                              uint8_t* local_taken, uint8_t* global_taken, orc_pair_pt2pt* out,
                              uint64_t* potential_pairings)
{
    if (!idxs)
        return orc_match_pt2pt(tree, gx, gy, gz, n_g, lx, ly, lz, n_l, T, prm, local_taken,
                               global_taken, out, potential_pairings);
    float*   sx = (float*)malloc((n_idxs + 1) * sizeof(float));
    float*   sy = (float*)malloc((n_idxs + 1) * sizeof(float));
    float*   sz = (float*)malloc((n_idxs + 1) * sizeof(float));
    uint8_t* st = local_taken ? (uint8_t*)malloc(n_idxs + 1) : NULL;
    for (size_t r = 0; r < n_idxs; r++)
    {
        const uint32_t i = idxs[r];
        sx[r] = lx[i], sy[r] = ly[i], sz[r] = lz[i];
        if (st) st[r] = local_taken[i];
    }
    /* :64 adds pcLocal.size() * pairingsPerPoint -- the WHOLE layer, not the visited subset */
    if (potential_pairings) *potential_pairings += (uint64_t)n_l * prm->pairingsPerPoint;
    const size_t n = orc_match_pt2pt(tree, gx, gy, gz, n_g, sx, sy, sz, n_idxs, T, prm, st,
                                     global_taken, out, NULL);
    for (size_t k = 0; k < n; k++) out[k].localIdx = idxs[out[k].localIdx]; /* :216 */
    if (st)
        for (size_t r = 0; r < n_idxs; r++)
            if (st[r]) local_taken[idxs[r]] = 1;
    free(sx), free(sy), free(sz), free(st);
    return n;
}

size_t orc_match_pt2pl_subset(const orc_kdtree* tree, const float* gx, const float* gy,
                              const float* gz, size_t n_g, const float* lx, const float* ly,
                              const float* lz, size_t n_l, const uint32_t* idxs, size_t n_idxs,
                              const double T[12], const orc_pt2pl_params* prm,
                              uint8_t* local_taken, orc_pair_pt2pl* out, uint32_t* out_local_idx,
                              uint64_t* potential_pairings)
{
    if (!idxs)
        return orc_match_pt2pl(tree, gx, gy, gz, n_g, lx, ly, lz, n_l, T, prm, local_taken, out,
                               out_local_idx, potential_pairings);
    float*    sx = (float*)malloc((n_idxs + 1) * sizeof(float));
    float*    sy = (float*)malloc((n_idxs + 1) * sizeof(float));
    float*    sz = (float*)malloc((n_idxs + 1) * sizeof(float));
    uint8_t*  st = local_taken ? (uint8_t*)malloc(n_idxs + 1) : NULL;
    uint32_t* oi = (uint32_t*)malloc((n_idxs + 1) * sizeof(uint32_t));
    for (size_t r = 0; r < n_idxs; r++)
    {
        const uint32_t i = idxs[r];
        sx[r] = lx[i], sy[r] = ly[i], sz[r] = lz[i];
        if (st) st[r] = local_taken[i];
    }
    /* :54 adds pcLocal.size() -- the WHOLE layer, not the visited subset */
    if (potential_pairings) *potential_pairings += (uint64_t)n_l;
    const size_t n = orc_match_pt2pl(tree, gx, gy, gz, n_g, sx, sy, sz, n_idxs, T, prm, st, out, oi,
                                     NULL);
    if (out_local_idx)
        for (size_t k = 0; k < n; k++) out_local_idx[k] = idxs[oi[k]]; /* :81 */
    if (st)
        for (size_t r = 0; r < n_idxs; r++)
            if (st[r]) local_taken[idxs[r]] = 1;
    free(sx), free(sy), free(sz), free(st), free(oi);
    (void)n_l;
    return n;
}

/* ======================================================================================
 *  f3 (part): Matcher_Points_InlierRatio::implMatchOneLayer
 *  (mp2p_icp/src/Matcher_Points_InlierRatio.cpp:40-143).
 *  std::multimap<double, pair> filled with emplace_hint(begin()): among equal keys the later
 *  insertion is visited first.  Returns the number of pairs, or (size_t)-1 for ASSERT_(nTotal > 0).
 * ====================================================================================== */
typedef struct
{
    double   key;
    uint32_t seq; /* insertion order */
    orc_pair_pt2pt p;
} ir_item;

static int ir_cmp(const void* a, const void* b)
{
    const ir_item* p = (const ir_item*)a;
    const ir_item* q = (const ir_item*)b;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    return p->seq > q->seq ? -1 : (p->seq < q->seq ? 1 : 0); /* later insertion first */
}

size_t orc_match_inlier_ratio(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                              size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                              const uint32_t* idxs, size_t n_idxs, const double T[12], double inliersRatio,
                              int allowMatchAlreadyMatchedPoints, int allowMatchAlreadyMatchedGlobalPoints,
                              double bbox_eps, uint8_t* local_taken, uint8_t* global_taken,
                              orc_pair_pt2pt* out, uint64_t* potential_pairings)
{
    if (potential_pairings) *potential_pairings += (uint64_t)n_l; /* :53 */
    if (n_g == 0 || n_l == 0) return 0;                           /* :56 */
    const size_t nv = idxs ? n_idxs : n_l;
    float* sx = (float*)malloc((nv + 1) * sizeof(float));
    float* sy = (float*)malloc((nv + 1) * sizeof(float));
    float* sz = (float*)malloc((nv + 1) * sizeof(float));
    for (size_t r = 0; r < nv; r++)
    {
        const size_t i = idxs ? idxs[r] : r;
        sx[r] = lx[i], sy[r] = ly[i], sz[r] = lz[i];
    }
    float* tx = (float*)malloc((nv + 1) * sizeof(float));
    float* ty = (float*)malloc((nv + 1) * sizeof(float));
    float* tz = (float*)malloc((nv + 1) * sizeof(float));
    float  lmin[3], lmax[3], gmin[3], gmax[3];
    orc_transform_local_to_global(sx, sy, sz, nv, T, tx, ty, tz, lmin, lmax); /* :58-59 */
    bbox_of(gx, gy, gz, n_g, gmin, gmax);
    size_t n_out = 0;
    uint8_t* own_g = NULL;
    if (!global_taken) global_taken = own_g = (uint8_t*)calloc(n_g, 1);
    if (bbox_intersects(gmin, gmax, lmin, lmax, (float)bbox_eps)) /* :63-66 */
    {
        ir_item* items = (ir_item*)malloc((nv + 1) * sizeof(ir_item));
        size_t   nTotal = 0;
        for (size_t r = 0; r < nv; r++) /* :78-106 */
        {
            const size_t i = idxs ? idxs[r] : r;
            if (!allowMatchAlreadyMatchedPoints && local_taken && local_taken[i]) continue;
            uint32_t id;
            float    d2;
            if (!nn_search(tree, gx, gy, gz, n_g, tx[r], ty[r], tz[r], 1, -1.0f, &id, &d2)) continue;
            ir_item* it = &items[nTotal];
            it->key = (double)d2, it->seq = (uint32_t)nTotal;
            it->p.globalIdx = id, it->p.localIdx = (uint32_t)i;
            it->p.gx = gx[id], it->p.gy = gy[id], it->p.gz = gz[id];
            it->p.lx = lx[i], it->p.ly = ly[i], it->p.lz = lz[i];
            it->p.errSq = d2;
            nTotal++;
        }
        if (nTotal == 0) /* :117 ASSERT_(nTotal > 0) */
        {
            free(items), free(sx), free(sy), free(sz), free(tx), free(ty), free(tz), free(own_g);
            return (size_t)-1;
        }
        qsort(items, nTotal, sizeof(ir_item), ir_cmp);
        const size_t nKeep = (size_t)nearbyint((double)nTotal * inliersRatio); /* :119 mrpt::round */
        for (size_t k = 0; k < nKeep && k < nTotal; k++) /* :125-139 */
        {
            const orc_pair_pt2pt* p = &items[k].p;
            if (!allowMatchAlreadyMatchedGlobalPoints && global_taken[p->globalIdx]) continue;
            out[n_out++] = *p;
            if (local_taken) local_taken[p->localIdx] = 1;
            global_taken[p->globalIdx] = 1;
        }
        free(items);
    }
    free(sx), free(sy), free(sz), free(tx), free(ty), free(tz), free(own_g);
    return n_out;
}

/* ======================================================================================
 *  f3: Matcher_Adaptive::implMatchOneLayer (mp2p_icp/src/Matcher_Adaptive.cpp:59-314).
 *  The neighbour lists, the plane test and the pair selection follow the reference's code.
 *  The adaptive threshold goes through mrpt::math::CHistogram and
 *  mrpt::math::confidenceIntervalsFromHistogram (MRPT >= 2.11.5, un-vendored): restated below
 *  from MRPT's published sources -- PARITY UNPINNED for these two helpers; callers can pass the
 *  threshold in instead (ci_high_given) to pin everything else.
 *    nn_radius_search(q, r2, ..., maxPoints) = the maxPoints nearest with d2 < r2, ascending
 *    (nanoflann RKNN result set); nn_single_search has no radius, the caller keeps d2 <= r2.
 * ====================================================================================== */
/* CHistogram(min, max, nBins): binSizeInv = (nBins-1)/(max-min); add(x): bin = (size_t)(binSizeInv *
 * (x-min)); getHistogramNormalized: x = linspace(min, max, nBins), y = bins * binSizeInv / count.
 * confidenceIntervalsFromHistogram(x, y, lo, hi, ci): Hc = cumsum(y) / max(Hc);
 * hi = x[first index with Hc > 1 - ci] (std::upper_bound). */
double orc_adaptive_ci_high(double minSq, double maxSq, const uint64_t* bins, int n_bins,
                            uint64_t count, double confidenceInterval)
{
    const double binSizeInv = (double)(n_bins - 1) / (maxSq - minSq);
    const double K          = binSizeInv / (double)count;
    double       Hc[ORC_ADAPTIVE_BINS], xs[ORC_ADAPTIVE_BINS];
    const double step = (maxSq - minSq) / (double)(n_bins - 1);
    double       c = minSq, run = 0, mx = 0;
    for (int i = 0; i < n_bins; i++)
    {
        xs[i] = c; /* mrpt::math::linspace: c = first; c += incr */
        c += step;
        run += K * (double)bins[i];
        Hc[i] = run;
        if (run > mx) mx = run;
    }
    const double inv = 1.0 / mx;
    const double ci  = 1.0 - confidenceInterval; /* the call passes 1.0 - confidenceInterval (:198) */
    for (int i = 0; i < n_bins; i++)
        if (Hc[i] * inv > 1.0 - ci) return xs[i];
    return NAN; /* ASSERT_(it_high != Hc.end()) */
}

int orc_match_adaptive(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                       const double T[12], const orc_adaptive_params* prm, uint8_t* local_taken,
                       const uint8_t* global_taken, int ci_high_given, double* ci_high,
                       orc_adaptive_hist* hist_out, orc_pair_pt2pt* out_pt2pt, size_t* n_pt2pt,
                       orc_pair_pt2pl* out_pt2pl, uint32_t* out_pl_local_idx, size_t* n_pt2pl,
                       uint64_t* potential_pairings)
{
    *n_pt2pt = *n_pt2pl = 0;
    if (hist_out) memset(hist_out, 0, sizeof(*hist_out));
    if (potential_pairings) *potential_pairings += (uint64_t)n_l * prm->maxPt2PtCorrespondences; /* :69 */
    if (n_g == 0 || n_l == 0) return 0;                                                           /* :72 */
    const uint32_t K = prm->enableDetectPlanes ? prm->planeSearchPoints : prm->maxPt2PtCorrespondences; /* :120 */
    if (K < 1 || K > ORC_MAX_K) return -1;

    float* tx = (float*)malloc(n_l * sizeof(float));
    float* ty = (float*)malloc(n_l * sizeof(float));
    float* tz = (float*)malloc(n_l * sizeof(float));
    float  lmin[3], lmax[3], gmin[3], gmax[3];
    orc_transform_local_to_global(lx, ly, lz, n_l, T, tx, ty, tz, lmin, lmax); /* :74-75 */
    bbox_of(gx, gy, gz, n_g, gmin, gmax);
    if (!bbox_intersects(gmin, gmax, lmin, lmax, (float)prm->bbox_eps)) /* :78-81 */
    {
        free(tx), free(ty), free(tz);
        return 0;
    }
    const float absMaxSq = (float)(prm->absoluteMaxSearchDistance * prm->absoluteMaxSearchDistance); /* :88 */
    const int   KEEP     = ORC_ADAPTIVE_MAX_CORRS; /* MAX_CORRS_PER_LOCAL (Matcher_Adaptive.h:83) */
    uint32_t*   m_idx    = (uint32_t*)malloc(n_l * KEEP * sizeof(uint32_t));
    float*      m_d2     = (float*)malloc(n_l * KEEP * sizeof(float));
    uint8_t*    m_n      = (uint8_t*)calloc(n_l, 1);
    int         have = 0;
    float       mn = 0, mx = 0;
    uint32_t    nidx[ORC_MAX_K];
    float       nd2[ORC_MAX_K];
    for (size_t i = 0; i < n_l; i++) /* :122-185 */
    {
        if (!prm->allowMatchAlreadyMatchedPoints && local_taken && local_taken[i]) continue; /* :126-132 */
        int found;
        if (K == 1)
            found = nn_search(tree, gx, gy, gz, n_g, tx[i], ty[i], tz[i], 1, -1.0f, nidx, nd2); /* :138-152 */
        else
            found = nn_search(tree, gx, gy, gz, n_g, tx[i], ty[i], tz[i], (int)K, absMaxSq, nidx, nd2); /* :155-158 */
        for (int k = 0; k < found; k++)
        {
            const float e = nd2[k];
            if (e > absMaxSq) continue; /* :165 */
            if (k <= 1)                 /* :167-181 */
            {
                if (have)
                {
                    if (e > mx) mx = e;
                    if (e < mn) mn = e;
                }
                else
                    mn = mx = e, have = 1;
            }
            if (m_n[i] >= KEEP) continue; /* lambdaAddPair :105 */
            m_idx[i * KEEP + m_n[i]] = nidx[k], m_d2[i * KEEP + m_n[i]] = e;
            m_n[i]++;
        }
    }
    int rc = 0;
    if (!have)
    { /* the reference dereferences an empty std::optional here (:189); nothing to pair */
        rc = 1;
        goto done;
    }
    {
        /* :189-198 */
        orc_adaptive_hist h;
        memset(&h, 0, sizeof(h));
        h.valid = 1, h.minSq = mn, h.maxSq = mx;
        const double binSizeInv = (double)(ORC_ADAPTIVE_BINS - 1) / ((double)mx - (double)mn);
        for (size_t i = 0; i < n_l; i++)
            for (int k = 0; k < m_n[i] && k < 2; k++)
            {
                const double x = (double)m_d2[i * KEEP + k];
                if (x < (double)mn || x > (double)mx) continue;
                /* min == max: MRPT's index is (size_t)(inf * 0); declared: bin 0 */
                const size_t b = (mx > mn) ? (size_t)(binSizeInv * (x - (double)mn)) : 0;
                h.bins[b < ORC_ADAPTIVE_BINS ? b : ORC_ADAPTIVE_BINS - 1]++;
                h.count++;
            }
        if (hist_out) *hist_out = h;
        double hi = *ci_high;
        if (!ci_high_given)
        {
            if (!(mx > mn))
            { /* binSizeInv = inf: MRPT's bin index is undefined; declared: threshold = max */
                hi = (double)mx;
            }
            else
                hi = orc_adaptive_ci_high((double)mn, (double)mx, h.bins, ORC_ADAPTIVE_BINS, h.count,
                                          prm->confidenceInterval);
            *ci_high = hi;
        }
        const double m2 = prm->minimumCorrDist * prm->minimumCorrDist;
        const double maxCorrDistSqr = m2 > hi ? m2 : hi;                                                    /* :212 */
        const float  maxSqr1to2 = (float)(prm->firstToSecondDistanceMax * prm->firstToSecondDistanceMax); /* :214 */
        float kx[ORC_ADAPTIVE_MAX_CORRS], ky[ORC_ADAPTIVE_MAX_CORRS], kz[ORC_ADAPTIVE_MAX_CORRS];
        for (size_t i = 0; i < n_l; i++) /* :217-295 */
        {
            const int       m   = m_n[i];
            const uint32_t* idx = &m_idx[i * KEEP];
            const float*    d2  = &m_d2[i * KEEP];
            if (prm->enableDetectPlanes && m >= (int)prm->planeMinimumFoundPoints && m >= 1) /* :221 */
            {
                for (int k = 0; k < m; k++) kx[k] = gx[idx[k]], ky[k] = gy[idx[k]], kz[k] = gz[idx[k]];
                float  mean[3];
                double cov[9], ev[3], evec[9];
                orc_estimate_points_eigen(kx, ky, kz, (size_t)m, mean, cov, ev, evec); /* :233-234 */
                if (ev[0] < prm->planeEigenThreshold * ev[2] && ev[0] < prm->planeEigenThreshold * ev[1]) /* :237-238 */
                {
                    double       n[3] = {evec[0], evec[1], evec[2]};
                    const double nn   = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    n[0] /= nn, n[1] /= nn, n[2] /= nn;
                    int big = 0; /* eigenvector sign: as orc_match_pt2pl (declared) */
                    if (fabs(n[1]) > fabs(n[big])) big = 1;
                    if (fabs(n[2]) > fabs(n[big])) big = 2;
                    if (n[big] < 0) n[0] = -n[0], n[1] = -n[1], n[2] = -n[2];
                    const double c[3] = {(double)mean[0], (double)mean[1], (double)mean[2]};
                    const double d    = -(n[0] * c[0] + n[1] * c[1] + n[2] * c[2]);
                    /* :245-246: distance of mspl[0].local = the UNtransformed local point (:113) */
                    const double dist = fabs(n[0] * (double)lx[i] + n[1] * (double)ly[i] + n[2] * (double)lz[i] + d);
                    if (dist < prm->planeMinimumDistance) /* :248 */
                    {
                        orc_pair_pt2pl* p = &out_pt2pl[*n_pt2pl];
                        p->plane[0] = n[0], p->plane[1] = n[1], p->plane[2] = n[2], p->plane[3] = d;
                        p->centroid[0] = c[0], p->centroid[1] = c[1], p->centroid[2] = c[2];
                        p->lx = lx[i], p->ly = ly[i], p->lz = lz[i];
                        p->_pad = 0;
                        if (out_pl_local_idx) out_pl_local_idx[*n_pt2pl] = (uint32_t)i;
                        (*n_pt2pl)++;
                        if (local_taken) local_taken[i] = 1; /* :260 */
                        continue;                            /* :263 */
                    }
                }
            }
            for (int k = 0; k < m && k < (int)prm->maxPt2PtCorrespondences; k++) /* :268 */
            {
                const uint32_t g = idx[k];
                if (!prm->allowMatchAlreadyMatchedGlobalPoints && global_taken && global_taken[g]) continue; /* :273-275 */
                if ((double)d2[k] >= maxCorrDistSqr) continue;                                              /* :278 */
                if (k != 0 && d2[k] > d2[0] * maxSqr1to2) break;                                             /* :280-284 */
                orc_pair_pt2pt* p = &out_pt2pt[*n_pt2pt];
                p->globalIdx = g, p->localIdx = (uint32_t)i;
                p->gx = gx[g], p->gy = gy[g], p->gz = gz[g];
                p->lx = lx[i], p->ly = ly[i], p->lz = lz[i];
                p->errSq = d2[k];
                (*n_pt2pt)++;
                if (!prm->allowMatchAlreadyMatchedGlobalPoints && local_taken) local_taken[i] = 1; /* :289-293 */
            }
        }
    }
done:
    free(tx), free(ty), free(tz), free(m_idx), free(m_d2), free(m_n);
    return rc;
}

/* ======================================================================================
 *  f4: covariance() (mp2p_icp/src/covariance.cpp:29-141): Hessian J^T J of the stacked error
 *  vector w.r.t. (x, y, z, yaw, pitch, roll) by central finite differences
 *  (mrpt::math::estimateJacobian: J(:,j) = (f(x + h_j e_j) - f(x - h_j e_j)) * (0.5 / h_j)),
 *  cov = H^-1.  Reproduced literally, including the linearisation point's z staying 0
 *  (:41-43 assign x twice and never z; harmless: every supported error term is affine in t).
 *  paired_ln2ln is not supported.
 * ====================================================================================== */
static void cov_errors(const orc_pair_pt2pt* pt, size_t n_pt, const orc_pair_pt2ln* ln, size_t n_ln,
                       const orc_pair_pt2pl* pl, size_t n_pl, const orc_pair_pl2pl* pp, size_t n_pp,
                       const double T[12], double* err)
{
    size_t k = 0;
    for (size_t i = 0; i < n_pt; i++, k += 3) orc_error_point2point(&pt[i], T, err + k, NULL); /* :74-80 */
    for (size_t i = 0; i < n_ln; i++, k += 3) orc_error_point2line(&ln[i], T, err + k, NULL);  /* :84-90 */
    for (size_t i = 0; i < n_pl; i++, k += 3) orc_error_point2plane(&pl[i], T, err + k, NULL); /* :104-110 */
    for (size_t i = 0; i < n_pp; i++, k += 3) orc_error_plane2plane(&pp[i], T, err + k, NULL); /* :114-120 */
}

int orc_covariance(const orc_pair_pt2pt* pt, size_t n_pt, const orc_pair_pt2pl* pl, size_t n_pl,
                   const orc_pair_pt2ln* ln, size_t n_ln, const orc_pair_pl2pl* pp, size_t n_pp,
                   const double T[12], double finDif_xyz, double finDif_angles, double H_out[36],
                   double cov_out[36])
{
    const size_t m = 3 * (n_pt + n_pl + n_ln + n_pp);
    if (m == 0) /* :33-39 */
    {
        for (int i = 0; i < 36; i++) cov_out[i] = 0, H_out[i] = 0;
        for (int i = 0; i < 6; i++) cov_out[i * 6 + i] = 1e6;
        return 0;
    }
    double x0[6];
    orc_pose_to_xyzypr(T, x0);
    x0[2] = 0.0; /* :41-43 */
    double* J  = (double*)calloc(m * 6, sizeof(double));
    double* fp = (double*)malloc(m * sizeof(double));
    double* fm = (double*)malloc(m * sizeof(double));
    for (int j = 0; j < 6; j++)
    {
        const double h = j < 3 ? finDif_xyz : finDif_angles;
        double       x[6], Tp[12];
        memcpy(x, x0, sizeof(x));
        x[j] = x0[j] + h;
        orc_pose_from_xyzypr(x[0], x[1], x[2], x[3], x[4], x[5], Tp);
        cov_errors(pt, n_pt, ln, n_ln, pl, n_pl, pp, n_pp, Tp, fp);
        x[j] = x0[j] - h;
        orc_pose_from_xyzypr(x[0], x[1], x[2], x[3], x[4], x[5], Tp);
        cov_errors(pt, n_pt, ln, n_ln, pl, n_pl, pp, n_pp, Tp, fm);
        const double s = 0.5 / h;
        for (size_t t = 0; t < m; t++) J[t * 6 + j] = s * (fp[t] - fm[t]);
    }
    double H[36];
    memset(H, 0, sizeof(H));
    for (size_t t = 0; t < m; t++)
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) H[a * 6 + b] += J[t * 6 + a] * J[t * 6 + b];
    memcpy(H_out, H, sizeof(H));
    /* inverse_LLt: Cholesky, then solve for the identity */
    double L[36];
    memset(L, 0, sizeof(L));
    int ok = 1;
    for (int i = 0; i < 6 && ok; i++)
        for (int j = 0; j <= i; j++)
        {
            double v = H[i * 6 + j];
            for (int k = 0; k < j; k++) v -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j)
            {
                if (!(v > 0)) { ok = 0; break; }
                L[i * 6 + i] = sqrt(v);
            }
            else
                L[i * 6 + j] = v / L[j * 6 + j];
        }
    for (int c = 0; c < 6 && ok; c++)
    {
        double y[6], z[6];
        for (int i = 0; i < 6; i++)
        {
            double v = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) v -= L[i * 6 + k] * y[k];
            y[i] = v / L[i * 6 + i];
        }
        for (int i = 5; i >= 0; i--)
        {
            double v = y[i];
            for (int k = i + 1; k < 6; k++) v -= L[k * 6 + i] * z[k];
            z[i] = v / L[i * 6 + i];
        }
        for (int i = 0; i < 6; i++) cov_out[i * 6 + c] = z[i];
    }
    if (!ok)
        for (int i = 0; i < 36; i++) cov_out[i] = NAN;
    free(J), free(fp), free(fm);
    return ok;
}

/* ======================================================================================
 *  f2: FilterDecimateVoxels (mp2p_icp_filters/src/FilterDecimateVoxels.cpp:107-381) with
 *  PointCloudToVoxelGrid[Single] (PointCloudToVoxelGrid.cpp:57-92, ...Single.cpp:50-92).
 *  One input layer.  The voxel container is the reference's std::map mode: voxels visited in
 *  ascending (cx, cy, cz) (IndicesHash::operator() as the comparator,
 *  PointCloudToVoxelGrid.h:100-111); the default tsl::robin_map mode visits the same voxels in
 *  a table-dependent order (un-vendored) -> the SET is pinned, the sequence only for std::map.
 * ====================================================================================== */
typedef struct
{
    int32_t  cx, cy, cz;
    uint32_t idx;
} dv_item;

static int dv_cmp(const void* a, const void* b)
{
    const dv_item* p = (const dv_item*)a;
    const dv_item* q = (const dv_item*)b;
    if (p->cx != q->cx) return p->cx < q->cx ? -1 : 1;
    if (p->cy != q->cy) return p->cy < q->cy ? -1 : 1;
    if (p->cz != q->cz) return p->cz < q->cz ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0); /* vxl.indices: ascending point index */
}

size_t orc_filter_decimate_voxels(const float* x, const float* y, const float* z, size_t n,
                                  float resolution, int method, int has_flatten_to,
                                  float flatten_to, float* ox, float* oy, float* oz,
                                  uint32_t* osrc)
{
    if (n == 0) return 0;
    dv_item* it = (dv_item*)malloc(n * sizeof(dv_item));
    for (size_t i = 0; i < n; i++)
    {
        /* coord2idx: static_cast<int32_t>(xyz / resolution_)  (PointCloudToVoxelGrid.h:110) */
        it[i].cx = (int32_t)(x[i] / resolution), it[i].cy = (int32_t)(y[i] / resolution),
        it[i].cz = (int32_t)(z[i] / resolution), it[i].idx = (uint32_t)i;
    }
    qsort(it, n, sizeof(dv_item), dv_cmp);
    size_t m = 0;
    for (size_t b = 0; b < n;)
    {
        size_t e = b + 1;
        while (e < n && it[e].cx == it[b].cx && it[e].cy == it[b].cy && it[e].cz == it[b].cz) e++;
        /* flatten: only the first voxel visited of each (cx, cy) column (:232-246, :346-360) */
        const int skip = has_flatten_to && b > 0 && it[b - 1].cx == it[b].cx && it[b - 1].cy == it[b].cy;
        if (!skip)
        {
            uint32_t src = it[b].idx; /* FirstPoint / idxInVoxel = 0 (:327-334) */
            float    px = x[src], py = y[src], pz = z[src];
            if (method != 0)
            {
                float mx = 0, my = 0, mz = 0; /* :291-300 */
                const float inv_n = 1.0f / (float)(e - b);
                for (size_t j = b; j < e; j++) mx += x[it[j].idx], my += y[it[j].idx], mz += z[it[j].idx];
                mx *= inv_n, my *= inv_n, mz *= inv_n;
                if (method == 2)
                    px = mx, py = my, pz = mz, src = 0xFFFFFFFFu; /* VoxelAverage :323-326 */
                else
                {
                    float best = 0;
                    int   have = 0;
                    for (size_t j = b; j < e; j++) /* ClosestToAverage :302-321 */
                    {
                        const uint32_t p  = it[j].idx;
                        const float    dx = x[p] - mx, dy = y[p] - my, dz = z[p] - mz;
                        const float    s  = (dx * dx + dy * dy) + dz * dz;
                        if (!have || s < best) best = s, src = p, have = 1;
                    }
                    px = x[src], py = y[src], pz = z[src];
                }
            }
            ox[m] = px, oy[m] = py, oz[m] = has_flatten_to ? flatten_to : pz;
            if (osrc) osrc[m] = src;
            m++;
        }
        b = e;
    }
    free(it);
    return m;
}

/* ======================================================================================
 *  a10: optimal_tf_gauss_newton (optimal_tf_gauss_newton.cpp:36-372)
 *  Sequential summation order; H and g reset at the top of every inner iteration (the
 *  TBB-build meaning :145-146).
 * ====================================================================================== */
static void accum_term(const double e[3], const double J1[36], const double dD[72], double w,
                       double* H, double* g)
{
    /* Ji = J1 (3x12) * dDexpe_de (12x6)   :176 */
    double Ji[18];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++)
        {
            double s = 0;
            for (int k = 0; k < 12; k++) s += J1[i * 12 + k] * dD[k * 6 + j];
            Ji[i * 6 + j] = s;
        }
    for (int a = 0; a < 6; a++)
    {
        g[a] += w * (Ji[0 * 6 + a] * e[0] + Ji[1 * 6 + a] * e[1] + Ji[2 * 6 + a] * e[2]); /* :177 */
        for (int b = 0; b < 6; b++)
            H[a * 6 + b] += w * (Ji[0 * 6 + a] * Ji[0 * 6 + b] + Ji[1 * 6 + a] * Ji[1 * 6 + b] +
                                 Ji[2 * 6 + a] * Ji[2 * 6 + b]); /* :178 */
    }
}

/* prior term :311-341.  df_de2 = d log(P1^-1 P2 exp(eps)) / d eps evaluated by central
 * differences (MRPT's jacob_dDinvP1invP2_de1e2 is un-vendored). */
static void prior_term(const double T[12], const orc_gn_params* prm, double* H, double* g)
{
    double Pinv[12], A[12], err[6];
    orc_pose_inverse(prm->prior_mean, Pinv);
    orc_pose_compose(Pinv, T, A); /* result.optimalPose - priorMean  :321 */
    orc_se3_log(A, err);          /* :322 */
    double       J[36];
    const double h = 1e-6;
    for (int j = 0; j < 6; j++)
    {
        double xi[6] = {0, 0, 0, 0, 0, 0}, E[12], Ap[12], Am[12], lp[6], lm[6];
        xi[j] = h;
        orc_se3_exp(xi, E);
        orc_pose_compose(A, E, Ap);
        xi[j] = -h;
        orc_se3_exp(xi, E);
        orc_pose_compose(A, E, Am);
        orc_se3_log(Ap, lp);
        orc_se3_log(Am, lm);
        for (int i = 0; i < 6; i++) J[i * 6 + j] = (lp[i] - lm[i]) / (2 * h);
    }
    /* g += (J^T Lambda) err ; H += (J^T Lambda) J   :338-340 */
    double JtL[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++)
        {
            double s = 0;
            for (int k = 0; k < 6; k++) s += J[k * 6 + i] * prm->prior_cov_inv[k * 6 + j];
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

int orc_optimal_tf_gauss_newton(const orc_pair_pt2pt* pt2pt, size_t n_pt2pt,
                                const orc_pair_pt2pl* pt2pl, size_t n_pt2pl,
                                const orc_pair_pt2ln* pt2ln, size_t n_pt2ln,
                                const orc_pair_pl2pl* pl2pl, size_t n_pl2pl,
                                const double T0[12], const orc_gn_params* prm, double T_out[12],
                                double* H_out, double* g_out)
{
    double T[12];
    memcpy(T, T0, sizeof(T)); /* :48 */
    double H[36], g[6];
    memset(H, 0, sizeof(H));
    memset(g, 0, sizeof(g));

    const int has_w     = prm->n_weight_blocks > 0; /* :66 */
    uint32_t  cur_block = 0;                        /* :67 */
    size_t    cur_start = 0;                        /* :68 */
    double    w_pt2pt   = prm->w_pt2pt;

    int iters = 0;
    for (uint32_t iter = 0; iter < prm->maxInnerLoopIterations; iter++) /* :70 */
    {
        iters++;
        double dD[72];
        orc_jacob_dDexpe_de(T, dD); /* :73 */
        double errNormSqr = 0;      /* :75 */
        memset(H, 0, sizeof(H));    /* TBB-branch meaning :145-146 */
        memset(g, 0, sizeof(g));
        if (prm->reset_weight_cursor_each_iter) cur_block = 0, cur_start = 0;

        for (size_t i = 0; i < n_pt2pt; i++) /* :149-180 */
        {
            double e[3], J1[36];
            orc_error_point2point(&pt2pt[i], T, e, J1);
            if (has_w)
            {
                if (i >= cur_start + prm->weight_block_count[cur_block] &&
                    cur_block + 1 < prm->n_weight_blocks)
                {
                    cur_block++;
                    cur_start = i;
                }
                w_pt2pt = prm->weight_block_w[cur_block];
            }
            double       w   = w_pt2pt;
            const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            if (prm->kernel != ORC_KERNEL_NONE) w *= orc_robust_weight(prm->kernel, prm->kernelParam, esq);
            errNormSqr += w * esq; /* :175 */
            accum_term(e, J1, dD, w, H, g);
        }
        for (size_t i = 0; i < n_pt2ln; i++) /* :184-202 */
        {
            double e[3], J1[36];
            orc_error_point2line(&pt2ln[i], T, e, J1);
            double       w   = prm->w_pt2ln;
            const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            if (prm->kernel != ORC_KERNEL_NONE) w *= orc_robust_weight(prm->kernel, prm->kernelParam, esq);
            errNormSqr += w * w * esq; /* :198 */
            accum_term(e, J1, dD, w, H, g);
        }
        for (size_t i = 0; i < n_pt2pl; i++) /* TBB branch :229-259 */
        {
            double e[3], J1[36];
            orc_error_point2plane(&pt2pl[i], T, e, J1);
            double       w   = prm->w_pt2pl;
            const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            if (prm->kernel != ORC_KERNEL_NONE) w *= orc_robust_weight(prm->kernel, prm->kernelParam, esq);
            errNormSqr += w * esq; /* :249 */
            accum_term(e, J1, dD, w, H, g);
        }
        for (size_t i = 0; i < n_pl2pl; i++) /* :289-308 */
        {
            double e[3], J1[36];
            orc_error_plane2plane(&pl2pl[i], T, e, J1);
            double       w   = prm->w_pl2pl;
            const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            if (prm->kernel != ORC_KERNEL_NONE) w *= orc_robust_weight(prm->kernel, prm->kernelParam, esq);
            errNormSqr += w * w * esq; /* :303 */
            accum_term(e, J1, dD, w, H, g);
        }
        if (prm->has_prior) prior_term(T, prm, H, g); /* :311-341 */

        if (sqrt(errNormSqr) <= prm->maxCost) break; /* :344-346 */

        double delta[6], rhs[6];
        for (int i = 0; i < 6; i++) rhs[i] = g[i];
        ldlt6_solve(H, rhs, delta);
        for (int i = 0; i < 6; i++) delta[i] = -delta[i]; /* :351 */
        double dE[12];
        orc_se3_exp(delta, dE);      /* :354 */
        orc_pose_compose(T, dE, T);  /* :356 */
        double nrm = 0;
        for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
        if (sqrt(nrm) < prm->minDelta) break; /* :365 */
    }
    memcpy(T_out, T, sizeof(T));
    if (H_out) memcpy(H_out, H, sizeof(H));
    if (g_out) memcpy(g_out, g, sizeof(g));
    return iters;
}

/* ---- multi-threaded accumulation (CPU baseline only; same math, per-thread partials) -- */
typedef struct
{
    const orc_pair_pt2pt* pt2pt;
    const orc_pair_pt2pl* pt2pl;
    size_t                b1, e1, b2, e2;
    const double *        T, *dD;
    const orc_gn_params*  prm;
    double                H[36], g[6], err;
} gn_job;

static void gn_worker_body(gn_job* j);
static void gn_pool_fn(void* jobs, int t) { gn_worker_body(&((gn_job*)jobs)[t]); }
static void gn_worker_body(gn_job* j)
{
    memset(j->H, 0, sizeof(j->H));
    memset(j->g, 0, sizeof(j->g));
    j->err = 0;
    for (size_t i = j->b1; i < j->e1; i++)
    {
        double e[3], J1[36];
        orc_error_point2point(&j->pt2pt[i], j->T, e, J1);
        double       w   = j->prm->w_pt2pt;
        const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        if (j->prm->kernel != ORC_KERNEL_NONE)
            w *= orc_robust_weight(j->prm->kernel, j->prm->kernelParam, esq);
        j->err += w * esq;
        accum_term(e, J1, j->dD, w, j->H, j->g);
    }
    for (size_t i = j->b2; i < j->e2; i++)
    {
        double e[3], J1[36];
        orc_error_point2plane(&j->pt2pl[i], j->T, e, J1);
        double       w   = j->prm->w_pt2pl;
        const double esq = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        if (j->prm->kernel != ORC_KERNEL_NONE)
            w *= orc_robust_weight(j->prm->kernel, j->prm->kernelParam, esq);
        j->err += w * esq;
        accum_term(e, J1, j->dD, w, j->H, j->g);
    }
}

int orc_optimal_tf_gauss_newton_mt(const orc_pair_pt2pt* pt2pt, size_t n_pt2pt,
                                   const orc_pair_pt2pl* pt2pl, size_t n_pt2pl,
                                   const double T0[12], const orc_gn_params* prm,
                                   double T_out[12], int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    double T[12];
    memcpy(T, T0, sizeof(T));
    gn_job*   jobs = (gn_job*)malloc(sizeof(gn_job) * n_threads);
    int       iters = 0;
    for (uint32_t iter = 0; iter < prm->maxInnerLoopIterations; iter++)
    {
        iters++;
        double dD[72], H[36], g[6], err = 0;
        orc_jacob_dDexpe_de(T, dD);
        memset(H, 0, sizeof(H));
        memset(g, 0, sizeof(g));
        for (int t = 0; t < n_threads; t++)
        {
            jobs[t].pt2pt = pt2pt, jobs[t].pt2pl = pt2pl;
            jobs[t].b1 = n_pt2pt * t / n_threads, jobs[t].e1 = n_pt2pt * (t + 1) / n_threads;
            jobs[t].b2 = n_pt2pl * t / n_threads, jobs[t].e2 = n_pt2pl * (t + 1) / n_threads;
            jobs[t].T = T, jobs[t].dD = dD, jobs[t].prm = prm;
        }
        pool_run(gn_pool_fn, jobs, n_threads);
        for (int t = 0; t < n_threads; t++)
        {
            for (int i = 0; i < 36; i++) H[i] += jobs[t].H[i];
            for (int i = 0; i < 6; i++) g[i] += jobs[t].g[i];
            err += jobs[t].err;
        }
        if (prm->has_prior) prior_term(T, prm, H, g);
        if (sqrt(err) <= prm->maxCost) break;
        double delta[6];
        ldlt6_solve(H, g, delta);
        for (int i = 0; i < 6; i++) delta[i] = -delta[i];
        double dE[12];
        orc_se3_exp(delta, dE);
        orc_pose_compose(T, dE, T);
        double nrm = 0;
        for (int i = 0; i < 6; i++) nrm += delta[i] * delta[i];
        if (sqrt(nrm) < prm->minDelta) break;
    }
    free(jobs);
    memcpy(T_out, T, sizeof(T));
    return iters;
}

/* ======================================================================================
 *  f1: optimal_tf_horn (optimal_tf_horn.cpp:77-252) with visit_correspondences
 *  (visit_correspondences.h:38-212), eval_centroids_robust (Pairings.cpp:68-110) and
 *  WeightParameters (WeightParameters.h:34-72): point pairs + plane-to-plane normals, pair
 *  weights, point_weights blocks, scale outlier detector (two passes), robust kernel against
 *  currentEstimateForRobust.  paired_ln2ln is not supported (no container on this path).
 *  Returns 1 solved, 0 not enough pairings (:98), -1 a reference ASSERT would throw.
 * ====================================================================================== */
static int horn_pass(const orc_pair_pt2pt* p, size_t nPt, const orc_pair_pl2pl* pp, size_t nPl,
                     const orc_horn_params* w, const double cl[3], const double cg[3],
                     uint8_t* outlier /* in/out, [nPt] */, double q_out[4])
{
    if (nPt + nPl < 3) return 0; /* :98 */
    const double wPt = w->w_pt2pt, wLi = w->w_ln2ln, wPl = w->w_pl2pl;
    if (!(wPt + wLi + wPl > 0.0)) return -1; /* visit_correspondences.h:79 */
    const double k = 1.0 / (wPt * (double)nPt + wLi * 0.0 + wPl * (double)nPl); /* :81 */
    const double waPoints = wPt * k, waPlanes = wPl * k;
    /* point_weights blocks (:60-69); the cursor only moves on visited pairs (:113-119) */
    size_t       blk = 0, blk_start = 0;
    const size_t n_blk = w->n_weight_blocks;
    double       S[9] = {0}, w_sum = 0;
    for (size_t i = 0; i < nPt + nPl; i++)
    {
        if (i < nPt && outlier[i]) continue; /* :97-103 (stays an outlier) */
        double bi[3], ri[3], wi;
        if (i < nPt)
        {
            wi = waPoints;
            if (n_blk)
            {
                if (i >= blk_start + w->weight_block_count[blk])
                {
                    if (blk + 1 >= n_blk) return -1; /* ASSERT_(cur != end) then dereference */
                    blk++, blk_start = i;
                }
                wi *= w->weight_block_w[blk];
            }
            bi[0] = p[i].gx - cg[0], bi[1] = p[i].gy - cg[1], bi[2] = p[i].gz - cg[2];
            ri[0] = p[i].lx - cl[0], ri[1] = p[i].ly - cl[1], ri[2] = p[i].lz - cl[2];
            const double bn = sqrt(bi[0] * bi[0] + bi[1] * bi[1] + bi[2] * bi[2]);
            const double rn = sqrt(ri[0] * ri[0] + ri[1] * ri[1] + ri[2] * ri[2]);
            if (bn < 1e-4 || rn < 1e-4) continue; /* :135-140 */
            if (w->use_scale_outlier_detector)    /* :153-164 */
            {
                const double mism = (bn > rn ? bn : rn) / (bn < rn ? bn : rn);
                if (mism > w->scale_outlier_threshold)
                {
                    outlier[i] = 1;
                    continue;
                }
            }
        }
        else
        { /* :181-192 getNormalVector(): the coefficients as stored */
            const orc_pair_pl2pl* q = &pp[i - nPt];
            wi = waPlanes;
            for (int d = 0; d < 3; d++) bi[d] = q->pl_global[d], ri[d] = q->pl_local[d];
        }
        if (w->robust_kernel != ORC_KERNEL_NONE) /* :194-205 */
        {
            if (!w->has_current_estimate) return -1;
            double r2[3];
            orc_pose_compose_point(w->current_estimate, ri[0], ri[1], ri[2], r2);
            const double e2 = (r2[0] - bi[0]) * (r2[0] - bi[0]) + (r2[1] - bi[1]) * (r2[1] - bi[1]) +
                              (r2[2] - bi[2]) * (r2[2] - bi[2]);
            wi *= orc_robust_weight(w->robust_kernel, w->robust_kernel_param, e2);
        }
        if (!(wi > 0.0)) return -1; /* :207 */
        w_sum += wi;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) S[a * 3 + b] += wi * ri[a] * bi[b]; /* optimal_tf_horn.cpp:112-122 */
    }
    if (w_sum > 0)
        for (int i = 0; i < 9; i++) S[i] *= 1.0 / w_sum; /* :128 */
    double N[16];
    N[0]  = S[0] + S[4] + S[8];
    N[1]  = S[5] - S[7];
    N[2]  = S[6] - S[2];
    N[3]  = S[1] - S[3];
    N[4]  = N[1];
    N[5]  = S[0] - S[4] - S[8];
    N[6]  = S[1] + S[3];
    N[7]  = S[6] + S[2];
    N[8]  = N[2];
    N[9]  = N[6];
    N[10] = -S[0] + S[4] - S[8];
    N[11] = S[5] + S[7];
    N[12] = N[3];
    N[13] = N[7];
    N[14] = N[11];
    N[15] = -S[0] - S[4] + S[8];
    double ev[4], V[16];
    sym_eig_jacobi(4, N, ev, V);
    double q[4] = {V[12], V[13], V[14], V[15]}; /* largest eigenvalue :160 */
    if (q[0] < 0)
        for (int i = 0; i < 4; i++) q[i] = -q[i]; /* :165-171 */
    memcpy(q_out, q, sizeof(q));
    return 1;
}

/* eval_centroids_robust (Pairings.cpp:68-110); returns 0 for ASSERT_GT_(nPt2Pt, outliers) */
static int horn_centroids(const orc_pair_pt2pt* p, size_t n, const uint8_t* outlier, double cl[3],
                          double cg[3])
{
    size_t n_out = 0;
    for (size_t i = 0; i < n; i++) n_out += outlier[i] ? 1 : 0;
    if (!(n > n_out)) return 0;
    const double wc = 1.0 / (double)(n - n_out);
    for (int d = 0; d < 3; d++) cl[d] = cg[d] = 0;
    for (size_t i = 0; i < n; i++)
    {
        if (outlier[i]) continue;
        cg[0] += p[i].gx, cg[1] += p[i].gy, cg[2] += p[i].gz;
        cl[0] += p[i].lx, cl[1] += p[i].ly, cl[2] += p[i].lz;
    }
    for (int d = 0; d < 3; d++) cl[d] *= wc, cg[d] *= wc;
    return 1;
}

int orc_optimal_tf_horn_wp(const orc_pair_pt2pt* p, size_t n, const orc_pair_pl2pl* pp, size_t n_pl,
                           const orc_horn_params* w, double T_out[12], uint8_t* outlier_out)
{
    if (w->w_pt2pt < 0 || w->w_ln2ln < 0 || w->w_pl2pl < 0) return -1; /* :207-209 */
    uint8_t* outlier = (uint8_t*)calloc(n ? n : 1, 1);
    double   cl[3], cg[3], q[4];
    int      rc = horn_centroids(p, n, outlier, cl, cg) ? 1 : -1; /* :212 */
    if (rc == 1) rc = horn_pass(p, n, pp, n_pl, w, cl, cg, outlier, q); /* :217 */
    size_t n_out = 0;
    for (size_t i = 0; i < n; i++) n_out += outlier[i];
    if (rc == 1 && w->use_scale_outlier_detector && n_out) /* :224-236 */
    {
        rc = horn_centroids(p, n, outlier, cl, cg) ? 1 : -1;
        if (rc == 1) rc = horn_pass(p, n, pp, n_pl, w, cl, cg, outlier, q);
    }
    if (rc == 1)
    {
        const double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int i = 0; i < 4; i++) q[i] /= qn;
        const double r = q[0], x = q[1], y = q[2], z = q[3];
        double       T[12];
        T[0] = r * r + x * x - y * y - z * z, T[1] = 2 * (x * y - r * z), T[2] = 2 * (z * x + r * y);
        T[3] = 2 * (x * y + r * z), T[4] = r * r - x * x + y * y - z * z, T[5] = 2 * (y * z - r * x);
        T[6] = 2 * (z * x - r * y), T[7] = 2 * (y * z + r * x), T[8] = r * r - x * x - y * y + z * z;
        T[9] = T[10] = T[11] = 0;
        double t[3];
        orc_pose_compose_point(T, cl[0], cl[1], cl[2], t);              /* :241-242 */
        T[9] = cg[0] - t[0], T[10] = cg[1] - t[1], T[11] = cg[2] - t[2]; /* :245-247 */
        memcpy(T_out, T, sizeof(T));
    }
    if (outlier_out) memcpy(outlier_out, outlier, n);
    free(outlier);
    return rc;
}

int orc_optimal_tf_horn(const orc_pair_pt2pt* p, size_t n, double w_pt2pt, double T_out[12])
{
    orc_horn_params w;
    memset(&w, 0, sizeof(w));
    w.w_pt2pt = w_pt2pt, w.scale_outlier_threshold = 1.2, w.robust_kernel_param = 1.0;
    const int rc = orc_optimal_tf_horn_wp(p, n, NULL, 0, &w, T_out, NULL);
    return rc == 1;
}

/* ======================================================================================
 *  pt2ln_pl_to_pt2pt (pt2ln_pl_to_pt2pt.cpp:47-113), what Solver_Horn feeds optimal_tf_horn when
 *  the pairings hold point-to-plane / point-to-line entries (Solver_Horn.cpp:51-55): every such
 *  pairing becomes (closest point on the plane / line, local point); they are taken from the
 *  largest distance down to 25 % of the largest (at least 3 in total); the input's own
 *  paired_pt2pt are NOT carried over (out starts empty, :49).  std::multimap::insert keeps equal
 *  keys in insertion order; the walk is in reverse.  Returns the number of pairs written.
 * ====================================================================================== */
typedef struct
{
    double         key;
    uint32_t       seq;
    orc_pair_pt2pt p;
} cv_item;
static int cv_cmp(const void* a, const void* b)
{
    const cv_item* x = (const cv_item*)a;
    const cv_item* y = (const cv_item*)b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1; /* descending */
    return x->seq > y->seq ? -1 : (x->seq < y->seq ? 1 : 0); /* later insertion first */
}
static size_t cv_append(cv_item* it, size_t n, orc_pair_pt2pt* out, size_t n_out)
{
    if (!n) return n_out; /* :33 */
    qsort(it, n, sizeof(cv_item), cv_cmp);
    const double thr = it[0].key * 0.25; /* :35-36 */
    for (size_t k = 0; k < n; k++)
    {
        if (it[k].key < thr && n_out >= 3) break; /* :42 */
        out[n_out++] = it[k].p;
    }
    return n_out;
}
size_t orc_pt2ln_pl_to_pt2pt(const orc_pair_pt2pl* pl, size_t n_pl, const orc_pair_pt2ln* ln,
                             size_t n_ln, const double T[12], orc_pair_pt2pt* out)
{
    const size_t m  = n_pl > n_ln ? n_pl : n_ln;
    cv_item*     it = (cv_item*)malloc((m + 1) * sizeof(cv_item));
    size_t       n_out = 0;
    for (size_t i = 0; i < n_pl; i++) /* :59-84 */
    {
        double g[3];
        orc_pose_compose_point(T, (double)pl[i].lx, (double)pl[i].ly, (double)pl[i].lz, g);
        const double* c = pl[i].plane;
        const double  d = c[0] * g[0] + c[1] * g[1] + c[2] * g[2] + c[3]; /* evaluatePoint */
        it[i].key = fabs(d), it[i].seq = (uint32_t)i;
        orc_pair_pt2pt* q = &it[i].p;
        q->globalIdx = q->localIdx = 0;
        q->gx = (float)(g[0] - c[0] * d), q->gy = (float)(g[1] - c[1] * d), q->gz = (float)(g[2] - c[2] * d);
        q->lx = pl[i].lx, q->ly = pl[i].ly, q->lz = pl[i].lz;
        q->errSq = 0; /* TMatchingPair default */
    }
    n_out = cv_append(it, n_pl, out, n_out);
    for (size_t i = 0; i < n_ln; i++) /* :92-109 */
    {
        double g[3];
        orc_pose_compose_point(T, ln[i].lx, ln[i].ly, ln[i].lz, g);
        /* TLine3D::closestPointTo: pBase + director * ((p - pBase).director / |director|^2) */
        const double* b = ln[i].pbase;
        const double* u = ln[i].director;
        const double  t = ((g[0] - b[0]) * u[0] + (g[1] - b[1]) * u[1] + (g[2] - b[2]) * u[2]) /
                         (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const double c[3] = {b[0] + t * u[0], b[1] + t * u[1], b[2] + t * u[2]};
        const double e[3] = {c[0] - g[0], c[1] - g[1], c[2] - g[2]};
        it[i].key = fabs(sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2])), it[i].seq = (uint32_t)i;
        orc_pair_pt2pt* q = &it[i].p;
        q->globalIdx = q->localIdx = 0;
        q->gx = (float)c[0], q->gy = (float)c[1], q->gz = (float)c[2];
        q->lx = (float)ln[i].lx, q->ly = (float)ln[i].ly, q->lz = (float)ln[i].lz;
        q->errSq = 0;
    }
    n_out = cv_append(it, n_ln, out, n_out);
    free(it);
    return n_out;
}
