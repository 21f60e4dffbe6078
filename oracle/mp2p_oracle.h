/*
 * mp2p_oracle.h -- CPU restatement ("oracle") of the mp2p_icp per-iteration hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libmp2p_hip.so) never links,
 * loads or calls anything in this directory.
 *
 * It restates, in dependency-free C, the *sequential* branches of the reference
 * (MOLAorg/mp2p_icp v1.8.0), with H and g reset at every Gauss-Newton inner iteration
 * (the shipped TBB-build meaning).  Every function cites the reference
 * file:line it follows (paths relative to /root/reference).
 *
 * Pinning status: the reference itself cannot be compiled here (MRPT, Eigen, TBB and
 * mola_common are absent) so oracle/_ref does not exist.  The oracle is
 * pinned against the reference's own known-answer tests, restated in tests/:
 *   tests/test-mp2p_matcher_pt2pt.cpp:56-107        (exact indices, 4 poses)
 *   tests/test-mp2p_optimize_pt2pl.cpp:36-128       (15 poses, 1e-3)
 *   tests/test-mp2p_optimize_with_prior.cpp:36-108  (3 cases)
 *   tests/test-mp2p_optimize_pt2ln.cpp:25-76        (1e-3)
 *   tests/test-mp2p_error_terms_jacobians.cpp:45-252 (J1*dDexpe_de vs finite differences)
 * Matcher_Point2Plane's neighbour search / plane fit has no arithmetic inside the
 * reference: its semantics are declared by this repo from
 * Matcher_Adaptive.cpp:227-270 + estimate_points_eigen.cpp:27-123 and are PARITY UNPINNED
 * beyond the (disabled upstream) tests/test-mp2p_matcher_pt2pl.cpp expectations.
 *
 * Conventions
 *   pose  : double T[12] = R (row-major 3x3) followed by t (3)      [CPose3D]
 *   xi    : double[6]    = [v; w]  translation part first           [Lie::SE<3>]
 *   points: SoA float arrays (CPointsMap::getPointsBufferRef_{x,y,z})
 */
#ifndef MP2P_ORACLE_H
#define MP2P_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data contracts (byte-compatible with the reference containers) ---------------- */

/* mrpt::tfest::TMatchingPair as used at Matcher_Points_DistanceThreshold.cpp:106-113 */
typedef struct
{
    uint32_t globalIdx, localIdx;
    float    gx, gy, gz; /* global (NN map point)            */
    float    lx, ly, lz; /* local  (UNtransformed)           */
    float    errSq;      /* errorSquareAfterTransformation   */
} orc_pair_pt2pt; /* 36 bytes */

/* mp2p_icp::point_plane_pair_t (point_plane_pair_t.h:34-46, plane_patch.h:30-40) */
typedef struct
{
    double plane[4];    /* TPlane coefs a,b,c,d   */
    double centroid[3]; /* plane_patch_t::centroid */
    float  lx, ly, lz;  /* pt_local (untransformed) */
    float  _pad;
} orc_pair_pt2pl; /* 72 bytes */

/* mp2p_icp::point_line_pair_t (Pairings.h:61-73): TLine3D{pBase,director} + TPoint3D pt_local
 * (a DOUBLE point, unlike the float point of point_plane_pair_t) */
typedef struct
{
    double pbase[3];
    double director[3];
    double lx, ly, lz;
} orc_pair_pt2ln; /* 72 bytes */

/* mp2p_icp::matched_plane_t (Pairings.h:37-48): two plane_patch_t {TPlane coefs, centroid} */
typedef struct
{
    double pl_global[4], c_global[3];
    double pl_local[4], c_local[3];
} orc_pair_pl2pl; /* 112 bytes */

enum
{
    ORC_KERNEL_NONE         = 0, /* robust_kernels.h:33-43 */
    ORC_KERNEL_GEMANMCCLURE = 1,
    ORC_KERNEL_CAUCHY       = 2
};

typedef struct
{
    double   threshold;           /* Matcher_Points_DistanceThreshold.cpp:43 */
    double   thresholdAngularDeg; /* :44 */
    uint32_t pairingsPerPoint;    /* :45 */
    int32_t  allowMatchAlreadyMatchedPoints;       /* Matcher_Points_Base.cpp:168-169 */
    int32_t  allowMatchAlreadyMatchedGlobalPoints; /* :171-172 */
    double   bbox_eps; /* bounding_box_intersection_check_epsilon, default 0.20 (:179-180) */
    int32_t  multi_search_radius_mode; /* K>1 only: 0 = nn_multiple_search (sequential
                                          branch :246-248), 1 = nn_radius_search(maxDistSq)
                                          (TBB branch :174-177).  SURVEY.md F9. */
} orc_pt2pt_params;

typedef struct
{
    double   distanceThreshold; /* Matcher_Point2Plane.cpp:38 */
    /* parameters of the NearestPlaneCapable map (declared semantics, see header note) */
    double   searchRadius;
    uint32_t knn;
    uint32_t minimumPlanePoints;
    double   planeEigenThreshold;
    int32_t  allowMatchAlreadyMatchedPoints;
    double   bbox_eps;
} orc_pt2pl_params;

typedef struct
{
    uint32_t maxInnerLoopIterations; /* optimal_tf_gauss_newton.h:32-61 */
    double   minDelta;               /* 1e-7 */
    double   maxCost;                /* 0    */
    int32_t  kernel;                 /* ORC_KERNEL_*  */
    double   kernelParam;
    double   w_pt2pt, w_pt2pl, w_pt2ln, w_pl2pl; /* PairWeights */
    int32_t  has_prior;
    double   prior_mean[12];    /* pose                               */
    double   prior_cov_inv[36]; /* 6x6 row-major information matrix   */
    /* Pairings::point_weights run-length blocks (count, weight); n_blocks==0 -> none */
    uint32_t      n_weight_blocks;
    const size_t* weight_block_count;
    const double* weight_block_w;
    int32_t  reset_weight_cursor_each_iter; /* 0 = literal reference (cursor not reset,
                                               optimal_tf_gauss_newton.cpp:66-68), 1 = intended */
} orc_gn_params;

/* ---- SE(3) helpers (closed forms of un-vendored MRPT) ------- */
void orc_pose_from_xyzypr(double x, double y, double z, double yaw, double pitch, double roll,
                          double T[12]);
void orc_pose_to_xyzypr(const double T[12], double out6[6]);
void orc_pose_identity(double T[12]);
void orc_pose_compose(const double A[12], const double B[12], double out[12]);
void orc_pose_inverse(const double A[12], double out[12]);
void orc_pose_compose_point(const double T[12], double lx, double ly, double lz, double g[3]);
void orc_pose_inverse_compose_point(const double T[12], double gx, double gy, double gz,
                                    double l[3]);
void orc_se3_exp(const double xi[6], double T[12]);
void orc_se3_log(const double T[12], double xi[6]);
void orc_jacob_dDexpe_de(const double T[12], double J[72]); /* 12x6 row-major */

/* ---- error terms (errorTerms.cpp) ---------------------------------------------------- */
void orc_error_point2point(const orc_pair_pt2pt* p, const double T[12], double e[3],
                           double J1[36] /* 3x12 row-major, may be NULL */);
void orc_error_point2plane(const orc_pair_pt2pl* p, const double T[12], double e[3],
                           double J1[36]);
void orc_error_plane2plane(const orc_pair_pl2pl* p, const double T[12], double e[3],
                           double J1[36]);
void orc_error_point2line(const orc_pair_pt2ln* p, const double T[12], double e[3],
                          double J1[36]);
double orc_robust_weight(int32_t kernel, double kernelParam, double errSq);

/* ---- a3: transform_local_to_global (Matcher_Points_Base.cpp:183-249) --------------- */
void orc_transform_local_to_global(const float* lx, const float* ly, const float* lz, size_t n,
                                   const double T[12], float* ox, float* oy, float* oz,
                                   float bbox_min[3], float bbox_max[3]);

/* ---- a5: exact nearest neighbours ---------------------------------------------------- */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree* orc_kdtree_build(const float* x, const float* y, const float* z, size_t n,
                             int leaf_max);
void        orc_kdtree_free(orc_kdtree* t);
/* k nearest, ascending (d2, idx); returns number found (<=k).  max_d2<0 => unbounded;
 * otherwise only neighbours with d2 < max_d2 (strict, nanoflann RadiusResultSet). */
int orc_kdtree_knn(const orc_kdtree* t, float qx, float qy, float qz, int k, float max_d2,
                   uint32_t* out_idx, float* out_d2);
/* brute-force version of the same contract: THE definition of the expected answer
 * (fp32 d2 = ((dx*dx)+(dy*dy))+(dz*dz), ties -> lowest index). */
int orc_brute_knn(const float* x, const float* y, const float* z, size_t n, float qx, float qy,
                  float qz, int k, float max_d2, uint32_t* out_idx, float* out_d2);

/* ---- a4: Matcher_Points_DistanceThreshold::implMatchOneLayer ------------------------- */
/* tree may be NULL -> brute force.  local_taken / global_taken: byte per point (0/1), may be
 * NULL (= all clear); updated like MatchState.  Returns number of pairs written
 * (capacity must be >= n_l * pairingsPerPoint). */
size_t orc_match_pt2pt(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                       const double T[12], const orc_pt2pt_params* prm, uint8_t* local_taken,
                       uint8_t* global_taken, orc_pair_pt2pt* out, uint64_t* potential_pairings);

/* multi-threaded variant used ONLY as the CPU baseline (range split of the query loop,
 * unique-global filter resolved afterwards by "lowest local index wins" -- identical
 * output to the sequential function). */
size_t orc_match_pt2pt_mt(const orc_kdtree* tree, const float* gx, const float* gy,
                          const float* gz, size_t n_g, const float* lx, const float* ly,
                          const float* lz, size_t n_l, const double T[12],
                          const orc_pt2pt_params* prm, orc_pair_pt2pt* out, int n_threads);
size_t orc_match_pt2pt_mt_ms(const orc_kdtree* tree, const float* gx, const float* gy,
                             const float* gz, size_t n_g, const float* lx, const float* ly,
                             const float* lz, size_t n_l, const double T[12],
                             const orc_pt2pt_params* prm, uint8_t* local_taken, uint8_t* global_taken,
                             orc_pair_pt2pt* out, int n_threads);
size_t orc_match_pt2pl_mt(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                          size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                          const double T[12], const orc_pt2pl_params* prm, uint8_t* local_taken,
                          orc_pair_pt2pl* out, uint32_t* out_local_idx,
                          uint64_t* potential_pairings, int n_threads);

/* ---- a6: Matcher_Point2Plane::implMatchOneLayer + declared nn_search_pt2pl ------------ */
size_t orc_match_pt2pl(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                       const double T[12], const orc_pt2pl_params* prm, uint8_t* local_taken,
                       orc_pair_pt2pl* out, uint32_t* out_local_idx,
                       uint64_t* potential_pairings);

/* maxLocalPointsPerLayer (Matcher_Points_Base.cpp:222-246): visit only idxs[0..n_idxs) in that
 * order; idxs == NULL -> the plain functions above.  Capacity of out >= n_idxs * K. */
size_t orc_match_pt2pt_subset(const orc_kdtree* tree, const float* gx, const float* gy,
                              const float* gz, size_t n_g, const float* lx, const float* ly,
                              const float* lz, size_t n_l, const uint32_t* idxs, size_t n_idxs,
                              const double T[12], const orc_pt2pt_params* prm,
                              uint8_t* local_taken, uint8_t* global_taken, orc_pair_pt2pt* out,
                              uint64_t* potential_pairings);
size_t orc_match_pt2pl_subset(const orc_kdtree* tree, const float* gx, const float* gy,
                              const float* gz, size_t n_g, const float* lx, const float* ly,
                              const float* lz, size_t n_l, const uint32_t* idxs, size_t n_idxs,
                              const double T[12], const orc_pt2pl_params* prm,
                              uint8_t* local_taken, orc_pair_pt2pl* out, uint32_t* out_local_idx,
                              uint64_t* potential_pairings);

/* estimate_points_eigen (estimate_points_eigen.cpp:27-123): fp32 mean, fp64 covariance,
 * eigenvalues ascending with eigenvectors as rows of evec[3][3]. */
void orc_estimate_points_eigen(const float* xs, const float* ys, const float* zs, size_t n,
                               float mean[3], double cov[9], double eval[3], double evec[9]);

/* ---- f3 (part): Matcher_Points_InlierRatio (Matcher_Points_InlierRatio.cpp:40-143).  idxs may be
 *      NULL.  Returns the number of pairs written, (size_t)-1 when no local point has a candidate
 *      (the reference's ASSERT_(nTotal > 0)). ---- */
size_t orc_match_inlier_ratio(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                              size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                              const uint32_t* idxs, size_t n_idxs, const double T[12], double inliersRatio,
                              int allowMatchAlreadyMatchedPoints, int allowMatchAlreadyMatchedGlobalPoints,
                              double bbox_eps, uint8_t* local_taken, uint8_t* global_taken,
                              orc_pair_pt2pt* out, uint64_t* potential_pairings);

/* ---- f3: Matcher_Adaptive (Matcher_Adaptive.cpp:59-314).  The histogram / confidence-interval
 *      helpers are MRPT's (un-vendored): restated, PARITY UNPINNED; pass ci_high_given = 1 and
 *      *ci_high to pin the rest.  Returns 0, 1 when no local point found any neighbour (the
 *      reference dereferences an empty optional), -1 on unsupported parameters. ---- */
#define ORC_ADAPTIVE_BINS 50      /* Matcher_Adaptive.cpp:189 */
#define ORC_ADAPTIVE_MAX_CORRS 10 /* Matcher_Adaptive.h:83 */
typedef struct
{
    double   confidenceInterval, firstToSecondDistanceMax, absoluteMaxSearchDistance, minimumCorrDist;
    int32_t  enableDetectPlanes;
    uint32_t maxPt2PtCorrespondences, planeSearchPoints, planeMinimumFoundPoints;
    double   planeMinimumDistance, planeEigenThreshold;
    int32_t  allowMatchAlreadyMatchedPoints, allowMatchAlreadyMatchedGlobalPoints;
    double   bbox_eps;
} orc_adaptive_params;
typedef struct
{
    int32_t  valid;
    float    minSq, maxSq;
    uint64_t count;
    uint64_t bins[ORC_ADAPTIVE_BINS];
} orc_adaptive_hist;
double orc_adaptive_ci_high(double minSq, double maxSq, const uint64_t* bins, int n_bins,
                            uint64_t count, double confidenceInterval);
int orc_match_adaptive(const orc_kdtree* tree, const float* gx, const float* gy, const float* gz,
                       size_t n_g, const float* lx, const float* ly, const float* lz, size_t n_l,
                       const double T[12], const orc_adaptive_params* prm, uint8_t* local_taken,
                       const uint8_t* global_taken, int ci_high_given, double* ci_high,
                       orc_adaptive_hist* hist_out, orc_pair_pt2pt* out_pt2pt, size_t* n_pt2pt,
                       orc_pair_pt2pl* out_pt2pl, uint32_t* out_pl_local_idx, size_t* n_pt2pl,
                       uint64_t* potential_pairings);

/* ---- f4: covariance() (covariance.cpp:29-141): H = J^T J by central differences over
 *      (x,y,z,yaw,pitch,roll), cov = H^-1 (Cholesky).  Returns 1, or 0 when H is not positive
 *      definite (cov = NaN) / there are no pairings (cov = 1e6 * I). ---- */
int orc_covariance(const orc_pair_pt2pt* pt, size_t n_pt, const orc_pair_pt2pl* pl, size_t n_pl,
                   const orc_pair_pt2ln* ln, size_t n_ln, const orc_pair_pl2pl* pp, size_t n_pp,
                   const double T[12], double finDif_xyz, double finDif_angles, double H_out[36],
                   double cov_out[36]);

/* ---- f2: FilterDecimateVoxels (one input layer; method 0 FirstPoint, 1 ClosestToAverage,
 *      2 VoxelAverage; std::map visiting order).  Returns the number of output points. ---- */
size_t orc_filter_decimate_voxels(const float* x, const float* y, const float* z, size_t n,
                                  float resolution, int method, int has_flatten_to,
                                  float flatten_to, float* ox, float* oy, float* oz,
                                  uint32_t* osrc);

/* ---- a10: optimal_tf_gauss_newton ----------------------------------------------------- */
/* Returns number of inner iterations executed.  H_out(36)/g_out(6) = last assembled
 * normal equations (may be NULL). */
int orc_optimal_tf_gauss_newton(const orc_pair_pt2pt* pt2pt, size_t n_pt2pt,
                                const orc_pair_pt2pl* pt2pl, size_t n_pt2pl,
                                const orc_pair_pt2ln* pt2ln, size_t n_pt2ln,
                                const orc_pair_pl2pl* pl2pl, size_t n_pl2pl,
                                const double T0[12], const orc_gn_params* prm, double T_out[12],
                                double* H_out, double* g_out);
/* multi-threaded accumulation, CPU baseline only */
int orc_optimal_tf_gauss_newton_mt(const orc_pair_pt2pt* pt2pt, size_t n_pt2pt,
                                   const orc_pair_pt2pl* pt2pl, size_t n_pt2pl,
                                   const double T0[12], const orc_gn_params* prm,
                                   double T_out[12], int n_threads);

/* ---- f1: Horn closed form (optimal_tf_horn.cpp:77-252 + visit_correspondences.h:38-212 +
 *      Pairings.cpp:68-110).  Returns 1 solved, 0 fewer than 3 pairings, -1 where the reference
 *      throws (all weights 0, a weight <= 0 on a visited pair, robust kernel without
 *      currentEstimateForRobust, no more points than outliers, weight blocks exhausted). ---- */
typedef struct
{
    int32_t use_scale_outlier_detector; /* WeightParameters.h:41 */
    double  scale_outlier_threshold;    /* 1.20 */
    double  w_pt2pt, w_ln2ln, w_pl2pl;  /* PairWeights used by Horn */
    int32_t robust_kernel;              /* ORC_KERNEL_* */
    double  robust_kernel_param;
    int32_t has_current_estimate;
    double  current_estimate[12]; /* currentEstimateForRobust */
    /* Pairings::point_weights blocks; 0 -> one block of weight 1 */
    uint32_t      n_weight_blocks;
    const size_t* weight_block_count;
    const double* weight_block_w;
} orc_horn_params;
int orc_optimal_tf_horn_wp(const orc_pair_pt2pt* pt2pt, size_t n, const orc_pair_pl2pl* pl2pl,
                           size_t n_pl2pl, const orc_horn_params* wp, double T_out[12],
                           uint8_t* outlier_flags /* [n] or NULL */);
/* point pairs, default WeightParameters; returns 1/0 */
int orc_optimal_tf_horn(const orc_pair_pt2pt* pt2pt, size_t n, double w_pt2pt, double T_out[12]);
/* pt2ln_pl_to_pt2pt (pt2ln_pl_to_pt2pt.cpp:47-113); capacity of out >= n_pl + n_ln */
size_t orc_pt2ln_pl_to_pt2pt(const orc_pair_pt2pl* pl, size_t n_pl, const orc_pair_pt2ln* ln,
                             size_t n_ln, const double T[12], orc_pair_pt2pt* out);

const char* orc_version(void);

#ifdef __cplusplus
}
#endif
#endif
