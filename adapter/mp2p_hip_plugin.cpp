// mp2p_hip_plugin.cpp -- reference-side binding of libmp2p_hip.so.
//
// A plugin for the UNMODIFIED mp2p_icp (MOLAorg/mp2p_icp v1.8.0): classes deriving the
// reference's own mp2p_icp::Matcher_Points_Base / mp2p_icp::Solver, registered with
// MRPT's class factory so that a pipeline YAML selects them with
//
//     matchers:
//       - class: mp2p_icp_hip::Matcher_Points_DistanceThreshold
//         plugin: libmp2p_icp_hip_plugin.so            # ICP.cpp:540-547, load_plugin.cpp:70-134
//         params: { threshold: 2.0, thresholdAngularDeg: 0 }
//     solvers:
//       - class: mp2p_icp_hip::Solver_GaussNewton
//         plugin: libmp2p_icp_hip_plugin.so            # ICP.cpp:501-508
//         params: { maxIterations: 3, robustKernel: 'RobustKernel::GemanMcClure', robustKernelParam: 0.15 }
//
// It needs MRPT >= 2.11.5 and mp2p_icp to build (neither exists in the development image of
// this repository, so this file is NOT compiled by __graft_entry__.build(); see
// adapter/CMakeLists.txt and INTEGRATION.md).  All numerics happen behind the C ABI of
// include/mp2p_hip.h; this file only converts containers.
//
// Interfaces implemented (reference file:line):
//   Matcher_Points_Base::implMatchOneLayer     Matcher_Points_Base.h:125-128 (private virtual)
//   Solver::impl_optimal_pose                  Solver.h:100-101   (Gauss-Newton, Horn)
//   Matcher::initialize / Solver::initialize   Matcher.h:88, Solver.h:80
//   registration                               register.cpp:43-69
#include <mp2p_icp/Matcher_Points_Base.h>
#include <mp2p_icp/Solver.h>
#include <mp2p_icp/PairWeights.h>
#include <mp2p_icp/WeightParameters.h>
#include <mp2p_icp/robust_kernels.h>
#include <mp2p_icp/metricmap.h>
#include <mrpt/core/initializer.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/random/random_shuffle.h>
#include <mrpt/rtti/CObject.h>

#include <chrono>
#include <cstring>
#include <numeric>
#include <random>
#include <map>
#include <memory>
#include <stdexcept>

#include "mp2p_hip.h"

static_assert(sizeof(mrpt::tfest::TMatchingPair) == sizeof(mp2p_hip_pair_pt2pt),
              "TMatchingPair layout changed: update mp2p_hip_pair_pt2pt");

namespace mp2p_icp_hip
{
// ---- one context + handle caches per thread (ICP::align is single-threaded per object) --------
struct Runtime
{
    mp2p_hip_ctx* ctx = nullptr;
    struct MapEntry
    {
        mp2p_hip_map* h = nullptr;
        size_t        n = 0;
        float         probe[6] = {0, 0, 0, 0, 0, 0};  // first/last point: cheap change detector
    };
    struct CloudEntry
    {
        mp2p_hip_cloud* h = nullptr;
        size_t          n = 0;
        float           probe[6] = {0, 0, 0, 0, 0, 0};
    };
    std::map<const void*, MapEntry>   maps;
    std::map<const void*, CloudEntry> clouds;
    // the Pairings the last matcher call left in HBM, so that the solver of the same ICP
    // iteration need not upload them again (run_matchers copies Pairings by value,
    // Matcher.cpp:74-77, so a derived container type would not survive)
    mp2p_hip_pairs* dev_pairs = nullptr;
    mp2p_hip_pairs* conv_pairs = nullptr;  // Solver_Horn: output of pt2ln_pl_to_pt2pt
    size_t          dev_cap_pt = 0, dev_cap_pl = 0;
    size_t          token_n_pt = 0, token_n_pl = 0;
    bool            has_lines_planes = false;
    uint32_t        token_first = 0, token_last = 0;

    static Runtime& get()
    {
        static thread_local Runtime r;
        if (!r.ctx)
        {
            const int rc = mp2p_hip_ctx_create(0, nullptr, &r.ctx);
            if (rc) throw std::runtime_error(std::string("mp2p_hip_ctx_create: ") + mp2p_hip_last_error(nullptr));
        }
        return r;
    }
    void check(int rc) const
    {
        if (rc) throw std::runtime_error(std::string("libmp2p_hip: ") + mp2p_hip_last_error(ctx));
    }
    static void fill_probe(const mrpt::maps::CPointsMap& m, float p[6])
    {
        const auto &x = m.getPointsBufferRef_x(), &y = m.getPointsBufferRef_y(), &z = m.getPointsBufferRef_z();
        const size_t n = x.size();
        if (!n) return;
        p[0] = x[0], p[1] = y[0], p[2] = z[0], p[3] = x[n - 1], p[4] = y[n - 1], p[5] = z[n - 1];
    }
    mp2p_hip_map* global_layer(const mrpt::maps::CPointsMap& m)
    {
        auto& e = maps[&m];
        float p[6] = {0, 0, 0, 0, 0, 0};
        fill_probe(m, p);
        if (!e.h || e.n != m.size() || std::memcmp(p, e.probe, sizeof(p)) != 0)
        {  // (re)build the index: the role of nn_prepare_for_3d_queries() after mark_as_modified()
            if (e.h) mp2p_hip_map_free(ctx, e.h);
            e.h = nullptr;
            check(mp2p_hip_map_upload(ctx, m.getPointsBufferRef_x().data(), m.getPointsBufferRef_y().data(),
                                      m.getPointsBufferRef_z().data(), m.size(), nullptr, &e.h));
            e.n = m.size();
            std::memcpy(e.probe, p, sizeof(p));
        }
        return e.h;
    }
    mp2p_hip_cloud* local_layer(const mrpt::maps::CPointsMap& m)
    {
        auto& e = clouds[&m];
        float p[6] = {0, 0, 0, 0, 0, 0};
        fill_probe(m, p);
        if (!e.h || e.n != m.size() || std::memcmp(p, e.probe, sizeof(p)) != 0)
        {
            if (e.h) mp2p_hip_cloud_free(ctx, e.h);
            e.h = nullptr;
            check(mp2p_hip_cloud_upload(ctx, m.getPointsBufferRef_x().data(), m.getPointsBufferRef_y().data(),
                                        m.getPointsBufferRef_z().data(), m.size(), &e.h));
            e.n = m.size();
            std::memcpy(e.probe, p, sizeof(p));
        }
        return e.h;
    }
    mp2p_hip_pairs* pairs(size_t cap_pt, size_t cap_pl)
    {
        if (!dev_pairs) check(mp2p_hip_pairs_create(ctx, cap_pt, cap_pl, &dev_pairs));
        else check(mp2p_hip_pairs_reserve(ctx, dev_pairs, cap_pt, cap_pl));
        dev_cap_pt = std::max(dev_cap_pt, cap_pt), dev_cap_pl = std::max(dev_cap_pl, cap_pl);
        return dev_pairs;
    }
};

// MatchState bit-fields <-> one byte per point (mp2p_hip_mstate)
static void bits_to_bytes(const mp2p_icp::pointcloud_bitfield_t::DenseOrSparseBitField& bf, size_t n,
                          std::vector<uint8_t>& out, bool& any)
{
    out.assign(n ? n : 1, 0);
    any = false;
    for (size_t i = 0; i < n; i++)
        if (bf[i]) out[i] = 1, any = true;
}

// ================================================================================================
class Matcher_Points_DistanceThreshold : public mp2p_icp::Matcher_Points_Base
{
    DEFINE_MRPT_OBJECT(Matcher_Points_DistanceThreshold, mp2p_icp_hip)
   public:
    void initialize(const mrpt::containers::yaml& params) override
    {
        Matcher_Points_Base::initialize(params);
        DECLARE_PARAMETER_REQ(params, threshold);            // Matcher_Points_DistanceThreshold.cpp:43
        DECLARE_PARAMETER_REQ(params, thresholdAngularDeg);  // :44
        DECLARE_PARAMETER_OPT(params, pairingsPerPoint);     // :45
    }
    double   threshold           = 0.50;
    double   thresholdAngularDeg = 0.50;
    uint32_t pairingsPerPoint    = 1;

   private:
    void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                           const mrpt::poses::CPose3D& localPose, mp2p_icp::MatchState& ms,
                           const mp2p_icp::layer_name_t& globalName, const mp2p_icp::layer_name_t& localName,
                           mp2p_icp::Pairings& out) const override
    {
        checkAllParametersAreRealized();
        ASSERT_(pairingsPerPoint >= 1);
        ASSERT_GT_(threshold, .0);
        ASSERT_GE_(thresholdAngularDeg, .0);
        ASSERT_LE_(pairingsPerPoint, 16u);
        const auto* gl = mp2p_icp::MapToPointsMap(pcGlobal);
        if (!gl) THROW_EXCEPTION("HIP matcher: the global layer must be a CPointsMap");

        auto& rt = Runtime::get();
        // maxLocalPointsPerLayer: the SAME list the reference would visit (Matcher_Points_Base.cpp:
        // 222-232), drawn here with MRPT's own shuffle and handed to the library
        std::vector<uint32_t> visit;
        if (maxLocalPointsPerLayer_ != 0 && pcLocal.size() > maxLocalPointsPerLayer_)
        {
            std::vector<std::size_t> idxs(maxLocalPointsPerLayer_);
            std::iota(idxs.begin(), idxs.end(), 0);
            const unsigned int seed = localPointsSampleSeed_ != 0
                                          ? localPointsSampleSeed_
                                          : std::chrono::system_clock::now().time_since_epoch().count();
            mrpt::random::partial_shuffle(idxs.begin(), idxs.end(), std::default_random_engine(seed),
                                          maxLocalPointsPerLayer_);
            visit.assign(idxs.begin(), idxs.end());
        }
        const size_t nVisited = visit.empty() ? pcLocal.size() : visit.size();
        out.potential_pairings += nVisited * pairingsPerPoint;  // :64 (the library adds the
        if (pcGlobal.isEmpty() || pcLocal.empty()) return;     //  same amount on its side)

        mp2p_hip_pt2pt_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.threshold = threshold, prm.thresholdAngularDeg = thresholdAngularDeg;
        prm.pairingsPerPoint = pairingsPerPoint;
        prm.allowMatchAlreadyMatchedPoints       = allowMatchAlreadyMatchedPoints_;
        prm.allowMatchAlreadyMatchedGlobalPoints = allowMatchAlreadyMatchedGlobalPoints_;
        prm.bounding_box_intersection_check_epsilon = bounding_box_intersection_check_epsilon_;

        // pose: R row-major + t
        double T[12];
        const auto& R = localPose.getRotationMatrix();
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) T[i * 3 + j] = R(i, j);
        T[9] = localPose.x(), T[10] = localPose.y(), T[11] = localPose.z();

        // MatchState -> device (only when something is already marked)
        mp2p_hip_mstate*     dms = nullptr;
        std::vector<uint8_t> gtaken, ltaken;
        bool                 anyG = false, anyL = false;
        auto& gbf = ms.globalPairedBitField.point_layers.at(globalName);
        auto& lbf = ms.localPairedBitField.point_layers.at(localName);
        bits_to_bytes(gbf, gl->size(), gtaken, anyG);
        bits_to_bytes(lbf, pcLocal.size(), ltaken, anyL);
        rt.check(mp2p_hip_mstate_create(rt.ctx, gl->size(), pcLocal.size(), &dms));
        if (anyG || anyL) rt.check(mp2p_hip_mstate_upload(rt.ctx, dms, gtaken.data(), ltaken.data()));

        mp2p_hip_pairs* dp = rt.pairs(out.paired_pt2pt.size() + pcLocal.size() * pairingsPerPoint, rt.dev_cap_pl);
        if (out.paired_pt2pt.empty() && out.paired_pt2pl.empty()) rt.check(mp2p_hip_pairs_clear(rt.ctx, dp));
        mp2p_hip_cloud* dl = rt.local_layer(pcLocal);
        rt.check(mp2p_hip_cloud_set_visit_order(rt.ctx, dl, visit.empty() ? nullptr : visit.data(), visit.size()));
        const int rc = mp2p_hip_match_pt2pt(rt.ctx, rt.global_layer(*gl), dl, T, &prm, dms, dp);
        if (rc)
        {
            mp2p_hip_mstate_free(rt.ctx, dms);
            rt.check(rc);
        }
        // host containers (ICP::align needs them: .empty(), quality, covariance, logs)
        uint64_t n64 = 0;
        rt.check(mp2p_hip_pairs_counts(rt.ctx, dp, &n64, nullptr, nullptr));
        size_t n = static_cast<size_t>(n64);
        out.paired_pt2pt.resize(n);  // the device list already holds what `out` held before
        if (n)
            rt.check(mp2p_hip_pairs_download_pt2pt(
                rt.ctx, dp, reinterpret_cast<mp2p_hip_pair_pt2pt*>(out.paired_pt2pt.data()), n, &n));
        // marks back to the host MatchState (only left when global re-use is forbidden, :116-120)
        if (!allowMatchAlreadyMatchedGlobalPoints_)
        {
            rt.check(mp2p_hip_mstate_download(rt.ctx, dms, gtaken.data(), ltaken.data()));
            for (size_t i = 0; i < gl->size(); i++)
                if (gtaken[i]) gbf.mark_as_set(i);
            for (size_t i = 0; i < pcLocal.size(); i++)
                if (ltaken[i]) lbf.mark_as_set(i);
        }
        mp2p_hip_mstate_free(rt.ctx, dms);
        rt.token_n_pt  = n;
        rt.token_first = n ? out.paired_pt2pt.front().localIdx : 0;
        rt.token_last  = n ? out.paired_pt2pt.back().localIdx : 0;
    }
};

// Pairings -> the device handle the solvers read.  Point pairings produced by this plugin's matcher
// in the same ICP iteration are still in HBM (Runtime::token_*): only lists from other matchers
// are uploaded.
static mp2p_hip_pairs* pairings_to_device(Runtime& rt, const mp2p_icp::Pairings& pairings)
{
    const size_t n1 = pairings.paired_pt2pt.size(), n2 = pairings.paired_pt2pl.size();
    mp2p_hip_pairs* dp = rt.pairs(std::max<size_t>(n1, 1), n2);
    const bool resident = n2 == 0 && n1 == rt.token_n_pt && n1 > 0 &&
                          pairings.paired_pt2pt.front().localIdx == rt.token_first &&
                          pairings.paired_pt2pt.back().localIdx == rt.token_last;
    if (!resident)
    {  // pairings this plugin did not produce (or pt2pl from another matcher): upload them
        std::vector<mp2p_hip_pair_pt2pl> pl(n2);
        for (size_t i = 0; i < n2; i++)
        {
            const auto& p = pairings.paired_pt2pl[i];
            for (int k = 0; k < 4; k++) pl[i].plane[k] = p.pl_global.plane.coefs[k];
            pl[i].centroid[0] = p.pl_global.centroid.x, pl[i].centroid[1] = p.pl_global.centroid.y,
            pl[i].centroid[2] = p.pl_global.centroid.z;
            pl[i].pt_local[0] = p.pt_local.x, pl[i].pt_local[1] = p.pt_local.y, pl[i].pt_local[2] = p.pt_local.z;
            pl[i]._pad = 0;
        }
        rt.check(mp2p_hip_pairs_upload(
            rt.ctx, dp, reinterpret_cast<const mp2p_hip_pair_pt2pt*>(pairings.paired_pt2pt.data()), n1,
            pl.data(), n2));
        rt.token_n_pt = 0;
    }

    {  // paired_pt2ln / paired_pl2pl always come from host matchers: (re)upload, also when empty
        std::vector<mp2p_hip_pair_pt2ln> ln(pairings.paired_pt2ln.size());
        for (size_t i = 0; i < ln.size(); i++)
        {
            const auto& q = pairings.paired_pt2ln[i];
            for (int k = 0; k < 3; k++)
                ln[i].ln_base[k] = q.ln_global.pBase[k], ln[i].ln_director[k] = q.ln_global.director[k],
                ln[i].pt_local[k] = q.pt_local[k];
        }
        std::vector<mp2p_hip_pair_pl2pl> pp(pairings.paired_pl2pl.size());
        for (size_t i = 0; i < pp.size(); i++)
        {
            const auto& q = pairings.paired_pl2pl[i];
            for (int k = 0; k < 4; k++)
                pp[i].pl_global[k] = q.p_global.plane.coefs[k], pp[i].pl_local[k] = q.p_local.plane.coefs[k];
            for (int k = 0; k < 3; k++)
                pp[i].c_global[k] = q.p_global.centroid[k], pp[i].c_local[k] = q.p_local.centroid[k];
        }
        if (!ln.empty() || !pp.empty() || rt.has_lines_planes)
            rt.check(mp2p_hip_pairs_upload_lines_planes(rt.ctx, dp, ln.data(), ln.size(), pp.data(), pp.size()));
        rt.has_lines_planes = !ln.empty() || !pp.empty();
    }

    return dp;
}

// ================================================================================================
class Solver_GaussNewton : public mp2p_icp::Solver
{
    DEFINE_MRPT_OBJECT(Solver_GaussNewton, mp2p_icp_hip)
   public:
    uint32_t               maxIterations = 5;
    mp2p_icp::PairWeights  pairWeights;
    mp2p_icp::RobustKernel robustKernel      = mp2p_icp::RobustKernel::None;
    double                 robustKernelParam = 1.0;

    void initialize(const mrpt::containers::yaml& params) override
    {
        Solver::initialize(params);
        MCP_LOAD_REQ(params, maxIterations);  // Solver_GaussNewton.cpp:33
        MCP_LOAD_OPT(params, robustKernel);   // :35
        DECLARE_PARAMETER_OPT(params, robustKernelParam);
        if (params.has("pair_weights")) pairWeights.load_from(params["pair_weights"]);
    }

   protected:
    bool impl_optimal_pose(const mp2p_icp::Pairings& pairings, mp2p_icp::OptimalTF_Result& out,
                           const mp2p_icp::SolverContext& sc) const override
    {
        checkAllParametersAreRealized();
        out = mp2p_icp::OptimalTF_Result();
        ASSERT_(sc.guessRelativePose.has_value());
        if (!pairings.paired_ln2ln.empty())
            THROW_EXCEPTION("HIP Gauss-Newton: paired_ln2ln is not supported");
        auto& rt = Runtime::get();

        mp2p_hip_pairs* dp = pairings_to_device(rt, pairings);

        mp2p_hip_gn_params p;
        std::memset(&p, 0, sizeof(p));
        p.maxInnerLoopIterations = maxIterations;
        p.minDelta = 1e-7, p.maxCost = 0;  // optimal_tf_gauss_newton.h:46-58
        p.kernel      = static_cast<int32_t>(robustKernel);
        p.kernelParam = robustKernelParam;
        p.w_pt2pt = pairWeights.pt2pt, p.w_pt2pl = pairWeights.pt2pl;
        p.w_pt2ln = pairWeights.pt2ln, p.w_pl2pl = pairWeights.pl2pl;
        ASSERT_(pairings.point_weights.size() <= 8);
        p.n_weight_blocks = static_cast<uint32_t>(pairings.point_weights.size());
        for (size_t i = 0; i < pairings.point_weights.size(); i++)
            p.weight_block_count[i] = pairings.point_weights[i].first, p.weight_block_w[i] = pairings.point_weights[i].second;
        auto fill_pose = [](const mrpt::poses::CPose3D& P, double T[12])
        {
            const auto& R = P.getRotationMatrix();
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) T[i * 3 + j] = R(i, j);
            T[9] = P.x(), T[10] = P.y(), T[11] = P.z();
        };
        if (sc.prior.has_value())
        {
            p.has_prior = 1;
            fill_pose(sc.prior->mean, p.prior_mean);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) p.prior_cov_inv[i * 6 + j] = sc.prior->cov_inv(i, j);
        }
        double T0[12];
        fill_pose(mrpt::poses::CPose3D(sc.guessRelativePose.value()), T0);
        mp2p_hip_gn_result res;
        rt.check(mp2p_hip_gn_solve(rt.ctx, dp, T0, &p, &res));
        mrpt::math::CMatrixDouble33 R;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R(i, j) = res.pose[i * 3 + j];
        out.optimalPose = mrpt::poses::CPose3D(R, mrpt::math::TPoint3D(res.pose[9], res.pose[10], res.pose[11]));
        return true;  // optimal_tf_gauss_newton.cpp:369
    }
};

// ================================================================================================
// Solver_Horn (Solver_Horn.cpp:33-61): WeightParameters from `pairingsWeightParameters`; pairings
// holding point-to-plane / point-to-line entries go through pt2ln_pl_to_pt2pt first.
class Solver_Horn : public mp2p_icp::Solver
{
    DEFINE_MRPT_OBJECT(Solver_Horn, mp2p_icp_hip)
   public:
    mp2p_icp::WeightParameters pairingsWeightParameters;

    void initialize(const mrpt::containers::yaml& params) override
    {
        Solver::initialize(params);
        if (params.has("pairingsWeightParameters"))
            pairingsWeightParameters.load_from(params["pairingsWeightParameters"]);  // Solver_Horn.cpp:37-38
    }

   protected:
    bool impl_optimal_pose(const mp2p_icp::Pairings& pairings, mp2p_icp::OptimalTF_Result& out,
                           const mp2p_icp::SolverContext& sc) const override
    {
        out = mp2p_icp::OptimalTF_Result();
        if (!pairings.paired_ln2ln.empty()) THROW_EXCEPTION("HIP Horn: paired_ln2ln is not supported");
        auto&                 rt  = Runtime::get();
        const mp2p_hip_pairs* eff = pairings_to_device(rt, pairings);
        const auto&           wp  = pairingsWeightParameters;
        auto fill_pose = [](const mrpt::poses::CPose3D& P, double T[12])
        {
            const auto& R = P.getRotationMatrix();
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) T[i * 3 + j] = R(i, j);
            T[9] = P.x(), T[10] = P.y(), T[11] = P.z();
        };
        const bool converted = !pairings.paired_pt2ln.empty() || !pairings.paired_pt2pl.empty();
        if (converted)
        {  // Solver_Horn.cpp:51-55: a fresh Pairings with the converted point pairings only
            ASSERT_(sc.guessRelativePose.has_value());
            const size_t cap = pairings.paired_pt2ln.size() + pairings.paired_pt2pl.size();
            if (!rt.conv_pairs) rt.check(mp2p_hip_pairs_create(rt.ctx, cap, 0, &rt.conv_pairs));
            else rt.check(mp2p_hip_pairs_reserve(rt.ctx, rt.conv_pairs, cap, 0));
            rt.check(mp2p_hip_pairs_clear(rt.ctx, rt.conv_pairs));
            double T[12];
            fill_pose(mrpt::poses::CPose3D(sc.guessRelativePose.value()), T);
            rt.check(mp2p_hip_pairs_pt2ln_pl_to_pt2pt(rt.ctx, eff, T, rt.conv_pairs));
            eff = rt.conv_pairs;
        }
        mp2p_hip_horn_params w;
        std::memset(&w, 0, sizeof(w));
        w.use_scale_outlier_detector = wp.use_scale_outlier_detector;
        w.scale_outlier_threshold    = wp.scale_outlier_threshold;
        w.w_pt2pt = wp.pair_weights.pt2pt, w.w_ln2ln = wp.pair_weights.ln2ln, w.w_pl2pl = wp.pair_weights.pl2pl;
        w.robust_kernel       = static_cast<int32_t>(wp.robust_kernel);
        w.robust_kernel_param = wp.robust_kernel_param;
        if (wp.currentEstimateForRobust.has_value())
        {
            w.has_current_estimate = 1;
            fill_pose(*wp.currentEstimateForRobust, w.current_estimate);
        }
        std::vector<size_t> blk_n;
        std::vector<double> blk_w;
        if (!converted)  // the converted list carries no point_weights (pt2ln_pl_to_pt2pt.cpp:49)
            for (const auto& b : pairings.point_weights) blk_n.push_back(b.first), blk_w.push_back(b.second);
        w.n_weight_blocks    = static_cast<uint32_t>(blk_n.size());
        w.weight_block_count = blk_n.data(), w.weight_block_w = blk_w.data();
        mp2p_hip_horn_result res;
        rt.check(mp2p_hip_horn_solve_wp(rt.ctx, eff, &w, &res));
        if (!res.solved) return false;  // optimal_tf_horn.cpp:98
        mrpt::math::CMatrixDouble33 R;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R(i, j) = res.pose[i * 3 + j];
        out.optimalPose = mrpt::poses::CPose3D(R, mrpt::math::TPoint3D(res.pose[9], res.pose[10], res.pose[11]));
        if (res.n_outliers)
        {  // OptimalTF_Result::outliers (OutlierIndices::point2point)
            uint64_t n64 = 0;
            rt.check(mp2p_hip_pairs_counts(rt.ctx, eff, &n64, nullptr, nullptr));
            std::vector<uint8_t> flags(static_cast<size_t>(n64));
            rt.check(mp2p_hip_horn_outlier_flags(rt.ctx, flags.data(), flags.size()));
            for (size_t i = 0; i < flags.size(); i++)
                if (flags[i]) out.outliers.point2point.push_back(i);
        }
        return true;
    }
};

IMPLEMENTS_MRPT_OBJECT(Matcher_Points_DistanceThreshold, mp2p_icp::Matcher, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Solver_Horn, mp2p_icp::Solver, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Solver_GaussNewton, mp2p_icp::Solver, mp2p_icp_hip)

}  // namespace mp2p_icp_hip

MRPT_INITIALIZER(register_mp2p_icp_hip)
{
    using mrpt::rtti::registerClass;
    registerClass(CLASS_ID(mp2p_icp_hip::Matcher_Points_DistanceThreshold));
    registerClass(CLASS_ID(mp2p_icp_hip::Solver_GaussNewton));
    registerClass(CLASS_ID(mp2p_icp_hip::Solver_Horn));
}
