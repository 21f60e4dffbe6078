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
//       - class: mp2p_icp_hip::Matcher_Point2Plane
//         plugin: libmp2p_icp_hip_plugin.so
//         params: { distanceThreshold: 0.4, searchRadius: 0.4, knn: 5, minimumPlanePoints: 5, planeEigenThreshold: 0.05 }
//     solvers:
//       - class: mp2p_icp_hip::Solver_GaussNewton
//         plugin: libmp2p_icp_hip_plugin.so            # ICP.cpp:501-508
//         params: { maxIterations: 3, robustKernel: 'RobustKernel::GemanMcClure', robustKernelParam: 0.15 }
//
// It needs MRPT >= 2.11.5 and mp2p_icp to build (neither exists in the development image of this
// repository, so this file is NOT compiled by __graft_entry__.build(); adapter/CMakeLists.txt and
// INTEGRATION.md).  tests/test_adapter_syntax.py compiles it with g++ -fsyntax-only against
// declaration-only stand-ins of the MRPT / mp2p_icp types it touches (tests/adapter_stubs/).
// All numerics happen behind the C ABI of include/mp2p_hip.h; all per-call host logic (what is
// transferred when) lives in mp2p_hip_host.hpp, which IS compiled, tested and timed here.  This file
// only converts MRPT containers into the plain views that header works on.
//
// Interfaces implemented (reference file:line):
//   Matcher::match (virtual)                    Matcher.h:93-96    -> captures MatchContext::icpIteration
//   Matcher_Points_Base::implMatchOneLayer      Matcher_Points_Base.h:125-128 (private virtual)
//   Solver::impl_optimal_pose                   Solver.h:100-101   (Gauss-Newton, Horn)
//   Matcher::initialize / Solver::initialize    Matcher.h:88, Solver.h:80
//   NearestPlaneCapable::nn_search_pt2pl        NearestPlaneCapable.h:49-50  (PointsMapPlanes layer)
//   Matcher_Adaptive / Matcher_Points_InlierRatio  Matcher_Adaptive.cpp:59-314, Matcher_Points_InlierRatio.cpp:41-143
//   FilterBase::filter (FilterDecimateVoxels)   FilterBase.h:62, FilterDecimateVoxels.cpp:107-381
//   -> with these, icp-run on demos/icp-settings-kitti.yaml (Matcher_Adaptive from iteration 6, every matcher on the
//      `decimated` layer) stays on the device path for the whole alignment
//   QualityEvaluator::evaluate (PairedRatio)    QualityEvaluator.h:49-51, QualityEvaluator_PairedRatio.cpp:45-73
//   registration                                register.cpp:43-69
#include <mp2p_icp/Matcher_Points_Base.h>
#include <mp2p_icp/NearestPlaneCapable.h>
#include <mp2p_icp/PairWeights.h>
#include <mp2p_icp/QualityEvaluator.h>
#include <mp2p_icp/Solver.h>
#include <mp2p_icp/WeightParameters.h>
#include <mp2p_icp/metricmap.h>
#include <mp2p_icp/pointcloud_bitfield.h>
#include <mp2p_icp/robust_kernels.h>
#include <mp2p_icp_filters/FilterBase.h>
#include <mp2p_icp_filters/FilterDecimateVoxels.h>
#include <mp2p_icp_filters/GetOrCreatePointLayer.h>
#include <mrpt/core/initializer.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/maps/CSimplePointsMap.h>
#include <mrpt/math/distributions.h>
#include <mrpt/random/random_shuffle.h>
#include <mrpt/rtti/CObject.h>

#include <chrono>
#include <cstring>
#include <numeric>
#include <optional>
#include <random>
#include <vector>

#include "mp2p_hip.h"
#include "mp2p_hip_host.hpp"

static_assert(sizeof(mrpt::tfest::TMatchingPair) == sizeof(mp2p_hip_pair_pt2pt),
              "TMatchingPair layout changed: update mp2p_hip_pair_pt2pt");

namespace mp2p_icp_hip
{
using mp2p_hip_host::BitView;
using mp2p_hip_host::MatchCall;
using mp2p_hip_host::Runtime;

// ---- the packed words behind a DenseOrSparseBitField -------------------------------------------------
// pointcloud_bitfield_t::DenseOrSparseBitField (pointcloud_bitfield.h:46-92) offers operator[] and
// mark_as_set only; its dense form is a private std::optional<std::vector<bool>>.  Reading 10 M bits
// through operator[] costs ~10 ms per matcher call, the words cost 30 us -- so the dense member is
// reached through the explicit-instantiation idiom (access checks do not apply to the arguments of an
// explicit template instantiation: standard C++, no reinterpret_cast of the object), and the word
// pointer through libstdc++'s public iterator member.  Any other standard library, or the sparse form,
// takes the generic route (a temporary packed copy filled through operator[]).
namespace detail
{
using BitField = mp2p_icp::pointcloud_bitfield_t::DenseOrSparseBitField;
template <typename Tag, typename Tag::type M>
struct Rob
{
    friend typename Tag::type get(Tag) { return M; }
};
struct DenseTag
{
    using type = std::optional<std::vector<bool>> BitField::*;
    friend type get(DenseTag);
};
template struct Rob<DenseTag, &BitField::dense_>;

inline std::optional<std::vector<bool>>& dense_of(BitField& bf) { return bf.*get(DenseTag{}); }
}  // namespace detail

// a BitView over one bit-field for the duration of a matcher call; commit() carries new marks back
// when the view is a temporary copy
class BitAccess
{
   public:
    BitAccess(detail::BitField& bf, size_t n) : bf_(bf), n_(n)
    {
#if defined(__GLIBCXX__)
        auto& d = detail::dense_of(bf);
        if (d.has_value() && d->size() >= n)
        {
            direct_ = true;
            view_   = BitView{reinterpret_cast<uint64_t*>(d->begin()._M_p), n};
            static_assert(sizeof(*d->begin()._M_p) == sizeof(uint64_t), "std::vector<bool> word size");
            return;
        }
#endif
        tmp_.assign((n + 63) / 64 + 1, 0);
        for (size_t i = 0; i < n; i++)
            if (bf[i]) tmp_[i >> 6] |= 1ull << (i & 63);
        before_ = tmp_;
        view_   = BitView{tmp_.data(), n};
    }
    BitView view() const { return view_; }
    void    commit()
    {
        if (direct_) return;
        for (size_t w = 0; w < tmp_.size(); w++)
        {
            uint64_t m = tmp_[w] & ~before_[w];
            while (m)
            {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                bf_.mark_as_set(w * 64 + (size_t)b);
            }
        }
    }

   private:
    detail::BitField&     bf_;
    size_t                n_;
    bool                  direct_ = false;
    BitView               view_;
    std::vector<uint64_t> tmp_, before_;
};

static void fill_pose(const mrpt::poses::CPose3D& P, double T[12])
{
    const auto& R = P.getRotationMatrix();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i * 3 + j] = R(i, j);
    T[9] = P.x(), T[10] = P.y(), T[11] = P.z();
}
static mrpt::poses::CPose3D make_pose(const double T[12])
{
    mrpt::math::CMatrixDouble33 R;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R(i, j) = T[i * 3 + j];
    return mrpt::poses::CPose3D(R, mrpt::math::TPoint3D(T[9], T[10], T[11]));
}

// maxLocalPointsPerLayer: the SAME list the reference would visit (Matcher_Points_Base.cpp:222-232),
// drawn with MRPT's own shuffle and handed to the library
static std::vector<uint32_t> visit_list(size_t n_local, uint64_t maxLocalPoints, uint64_t seed0)
{
    std::vector<uint32_t> visit;
    if (maxLocalPoints != 0 && n_local > maxLocalPoints)
    {
        std::vector<std::size_t> idxs(maxLocalPoints);
        std::iota(idxs.begin(), idxs.end(), 0);
        const unsigned int seed =
            seed0 != 0 ? (unsigned int)seed0 : (unsigned int)std::chrono::system_clock::now().time_since_epoch().count();
        mrpt::random::partial_shuffle(idxs.begin(), idxs.end(), std::default_random_engine(seed), maxLocalPoints);
        visit.assign(idxs.begin(), idxs.end());
    }
    return visit;
}

// what both matchers share: the ICP iteration and MatchState of the running run_matchers call
class MatcherCallContext
{
   protected:
    mutable uint32_t    cur_iteration_ = 0;
    mutable const void* cur_ms_        = nullptr;
    void note_call(const mp2p_icp::MatchContext& mc, const mp2p_icp::MatchState& ms) const
    {
        cur_iteration_ = mc.icpIteration, cur_ms_ = &ms;
    }
    // handles of the two layers: verified in full at ICP iteration 0 (ICP::align holds its maps const
    // afterwards), by size + buffer addresses + 1024 sampled points otherwise
    void layers(Runtime& rt, const mrpt::maps::CPointsMap& gl, const mrpt::maps::CPointsMap& lc, mp2p_hip_map*& m,
                mp2p_hip_cloud*& c) const
    {
        const bool full = cur_iteration_ == 0;
        m = rt.global_layer(&gl, gl.getPointsBufferRef_x().data(), gl.getPointsBufferRef_y().data(),
                            gl.getPointsBufferRef_z().data(), gl.size(), full);
        c = rt.local_layer(&lc, lc.getPointsBufferRef_x().data(), lc.getPointsBufferRef_y().data(),
                           lc.getPointsBufferRef_z().data(), lc.size(), full);
    }
    // the local layer's own arrays travel with the call: the point pairings then come back as 24 bytes per pair and their
    // `local` member is read from here (mp2p_hip_host::copy_pt2pt_begin)
    static void local_arrays(MatchCall& call, const mrpt::maps::CPointsMap& lc)
    {
        call.lx = lc.getPointsBufferRef_x().data(), call.ly = lc.getPointsBufferRef_y().data();
        call.lz = lc.getPointsBufferRef_z().data(), call.n_local = lc.size();
    }
};

// ================================================================================================
class Matcher_Points_DistanceThreshold : public mp2p_icp::Matcher_Points_Base, protected MatcherCallContext
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
    bool match(const mp2p_icp::metric_map_t& pcGlobal, const mp2p_icp::metric_map_t& pcLocal,
               const mrpt::poses::CPose3D& localPose, const mp2p_icp::MatchContext& mc, mp2p_icp::MatchState& ms,
               mp2p_icp::Pairings& out) const override
    {
        mp2p_hip_host::RoctxRange roctx_range("align.3.1_matchers");  // the reference's profiler section (ICP.cpp:141)
        note_call(mc, ms);
        return mp2p_icp::Matcher::match(pcGlobal, pcLocal, localPose, mc, ms, out);
    }
    double   threshold           = 0.50;
    double   thresholdAngularDeg = 0.50;
    uint32_t pairingsPerPoint    = 1;

    // Count-only calls (mp2p_icp_hip::QualityEvaluator_PairedRatio): with allowMatchAlreadyMatchedGlobalPoints the
    // matcher neither reads global marks nor leaves any mark (Matcher_Points_DistanceThreshold.cpp:98-101, 116-120), so
    // on the evaluator's fresh MatchState the layer pairs are independent and only the NUMBER of pairs is wanted:
    // nothing but two counters leaves the device.  Returns false when the flags do not allow that shortcut.
    bool count_pairs(const mp2p_icp::metric_map_t& pcGlobal, const mp2p_icp::metric_map_t& pcLocal,
                     const mrpt::poses::CPose3D& localPose, mp2p_icp::MatchState& ms, uint64_t& n_pairs,
                     uint64_t& potential) const
    {
        if (!allowMatchAlreadyMatchedGlobalPoints_) return false;
        mp2p_icp::Pairings tmp;
        // The request travels in thread-local state scoped to THIS call (mutable members made concurrent or
        // re-entrant evaluate() calls on one evaluator mix counts or silently return no pairs): the reference's
        // QualityEvaluator_PairedRatio::evaluate is a stateless const method, and so is this one per thread.
        struct Scope
        {
            CountRequest* prev;
            explicit Scope(CountRequest* r) : prev(count_request()) { count_request() = r; }
            ~Scope() { count_request() = prev; }
        };
        CountRequest req{this, 0};
        {
            Scope scope(&req);
            match(pcGlobal, pcLocal, localPose, {}, ms, tmp);
        }
        n_pairs = req.counted, potential = tmp.potential_pairings;
        return true;
    }

   private:
    struct CountRequest
    {
        const void* owner;  // the matcher the request is addressed to (a nested matcher of another object ignores it)
        uint64_t    counted;
    };
    static CountRequest*& count_request()
    {
        static thread_local CountRequest* r = nullptr;
        return r;
    }
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

        out.potential_pairings += pcLocal.size() * pairingsPerPoint;  // :64 (the whole layer)
        if (pcGlobal.isEmpty() || pcLocal.empty()) return;            // :67

        mp2p_hip_pt2pt_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.threshold = threshold, prm.thresholdAngularDeg = thresholdAngularDeg;
        prm.pairingsPerPoint                     = pairingsPerPoint;
        prm.allowMatchAlreadyMatchedPoints       = allowMatchAlreadyMatchedPoints_;
        prm.allowMatchAlreadyMatchedGlobalPoints = allowMatchAlreadyMatchedGlobalPoints_;
        prm.bounding_box_intersection_check_epsilon = bounding_box_intersection_check_epsilon_;
        prm.multi_search_radius_mode = 1;  // pairingsPerPoint > 1: the shipped (TBB) build's nn_radius_search (:172-177)
        double T[12];
        fill_pose(localPose, T);

        auto&                       rt    = Runtime::get();
        const std::vector<uint32_t> visit = visit_list(pcLocal.size(), maxLocalPointsPerLayer_, localPointsSampleSeed_);
        mp2p_hip_map*               m;
        mp2p_hip_cloud*             c;
        layers(rt, *gl, pcLocal, m, c);
        if (CountRequest* req = count_request(); req && req->owner == this)
        {
            req->counted += mp2p_hip_host::count_pt2pt_layer(rt, m, c, T, prm, visit.data(), visit.size());
            return;
        }
        BitAccess gbits(ms.globalPairedBitField.point_layers.at(globalName), gl->size());
        BitAccess lbits(ms.localPairedBitField.point_layers.at(localName), pcLocal.size());
        MatchCall call;
        call.ms_key = cur_ms_ ? cur_ms_ : &ms, call.iteration = cur_iteration_;
        call.gbits = gbits.view(), call.lbits = lbits.view();
        local_arrays(call, pcLocal);
        // mrpt::tfest::TMatchingPairList derives std::vector<TMatchingPair>: appended in place
        mp2p_hip_host::match_pt2pt_layer(rt, call, m, c, T, prm, visit.data(), visit.size(), out.paired_pt2pt);
        gbits.commit(), lbits.commit();
    }
};

// ================================================================================================
// Matcher_Point2Plane (Matcher_Point2Plane.cpp:41-114) with the plane search batched over the whole
// local layer on the device.  The reference delegates the plane construction to the global layer's
// nn_search_pt2pl (no implementor in its own tree); here it is the k-NN + covariance + eigen rule
// declared in oracle/mp2p_oracle.c, with its parameters exposed in the YAML (defaults = those of
// the PointsMapPlanes layer below, so both routes pair the same points).
class Matcher_Point2Plane : public mp2p_icp::Matcher_Points_Base, protected MatcherCallContext
{
    DEFINE_MRPT_OBJECT(Matcher_Point2Plane, mp2p_icp_hip)
   public:
    void initialize(const mrpt::containers::yaml& params) override
    {
        Matcher_Points_Base::initialize(params);
        DECLARE_PARAMETER_REQ(params, distanceThreshold);  // Matcher_Point2Plane.cpp:38
        DECLARE_PARAMETER_OPT(params, searchRadius);
        MCP_LOAD_OPT(params, knn);
        MCP_LOAD_OPT(params, minimumPlanePoints);
        MCP_LOAD_OPT(params, planeEigenThreshold);
    }
    bool match(const mp2p_icp::metric_map_t& pcGlobal, const mp2p_icp::metric_map_t& pcLocal,
               const mrpt::poses::CPose3D& localPose, const mp2p_icp::MatchContext& mc, mp2p_icp::MatchState& ms,
               mp2p_icp::Pairings& out) const override
    {
        mp2p_hip_host::RoctxRange roctx_range("align.3.1_matchers");  // the reference's profiler section (ICP.cpp:141)
        note_call(mc, ms);
        return mp2p_icp::Matcher::match(pcGlobal, pcLocal, localPose, mc, ms, out);
    }
    double   distanceThreshold   = 0.50;
    double   searchRadius        = 0.0;  // 0: distanceThreshold
    uint32_t knn                 = 5;
    uint32_t minimumPlanePoints  = 5;
    double   planeEigenThreshold = 0.05;

   private:
    void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                           const mrpt::poses::CPose3D& localPose, mp2p_icp::MatchState& ms,
                           const mp2p_icp::layer_name_t& globalName, const mp2p_icp::layer_name_t& localName,
                           mp2p_icp::Pairings& out) const override
    {
        checkAllParametersAreRealized();
        ASSERT_GT_(distanceThreshold, .0);
        ASSERT_(knn >= 3 && knn <= 16);
        const auto* gl = mp2p_icp::MapToPointsMap(pcGlobal);
        if (!gl) THROW_EXCEPTION("HIP point-to-plane matcher: the global layer must be a CPointsMap");
        out.potential_pairings += pcLocal.size();           // :54
        if (pcGlobal.isEmpty() || pcLocal.empty()) return;  // :57

        mp2p_hip_pt2pl_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.distanceThreshold = distanceThreshold;
        prm.searchRadius      = searchRadius > 0 ? searchRadius : distanceThreshold;
        prm.knn = knn, prm.minimumPlanePoints = minimumPlanePoints, prm.planeEigenThreshold = planeEigenThreshold;
        prm.allowMatchAlreadyMatchedPoints          = allowMatchAlreadyMatchedPoints_;
        prm.bounding_box_intersection_check_epsilon = bounding_box_intersection_check_epsilon_;
        double T[12];
        fill_pose(localPose, T);

        auto&                       rt    = Runtime::get();
        const std::vector<uint32_t> visit = visit_list(pcLocal.size(), maxLocalPointsPerLayer_, localPointsSampleSeed_);
        mp2p_hip_map*               m;
        mp2p_hip_cloud*             c;
        layers(rt, *gl, pcLocal, m, c);
        BitAccess lbits(ms.localPairedBitField.point_layers.at(localName), pcLocal.size());
        MatchCall call;
        call.ms_key = cur_ms_ ? cur_ms_ : &ms, call.iteration = cur_iteration_;
        call.gbits = BitView{nullptr, gl->size()}, call.lbits = lbits.view();  // global marks: not read, not set (:87-90)
        (void)globalName;
        mp2p_hip_host::match_pt2pl_layer(
            rt, call, m, c, T, prm, visit.data(), visit.size(),
            [&](const mp2p_hip_pair_pt2pl& r)
            {
                auto& p = out.paired_pt2pl.emplace_back();
                p.pt_local = {r.pt_local[0], r.pt_local[1], r.pt_local[2]};
                for (int k = 0; k < 4; k++) p.pl_global.plane.coefs[k] = r.plane[k];
                p.pl_global.centroid = {r.centroid[0], r.centroid[1], r.centroid[2]};
            });
        lbits.commit();
    }
};

// ================================================================================================
// A point layer type that also offers NearestPlaneCapable (NearestPlaneCapable.h:33-52), so that the
// reference's OWN Matcher_Point2Plane runs on it (MapToNP is a dynamic_cast, metricmap.cpp:804-822).
// One device call per query (~0.1 ms): the contract, not the fast path -- a pipeline that wants speed
// names mp2p_icp_hip::Matcher_Point2Plane, which batches the whole layer.
class PointsMapPlanes : public mrpt::maps::CSimplePointsMap, public mp2p_icp::NearestPlaneCapable
{
    DEFINE_SERIALIZABLE(PointsMapPlanes, mp2p_icp_hip)
   public:
    uint32_t knn                 = 5;
    uint32_t minimumPlanePoints  = 5;
    double   planeEigenThreshold = 0.05;
    double   searchRadius        = 0.0;  // 0: the max_search_distance of the query

    NearestPlaneResult nn_search_pt2pl(const mrpt::math::TPoint3Df& point, const float max_search_distance) const override
    {
        NearestPlaneResult res;
        if (this->empty()) return res;
        auto&         rt = Runtime::get();
        mp2p_hip_map* m  = rt.global_layer(this, getPointsBufferRef_x().data(), getPointsBufferRef_y().data(),
                                           getPointsBufferRef_z().data(), size(), /*full_check=*/false);
        mp2p_hip_pt2pl_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.searchRadius = searchRadius, prm.knn = knn, prm.minimumPlanePoints = minimumPlanePoints;
        prm.planeEigenThreshold                     = planeEigenThreshold;
        prm.bounding_box_intersection_check_epsilon = 0.20;
        const float            q[3] = {point.x, point.y, point.z};
        mp2p_hip_nearest_plane np;
        rt.check(mp2p_hip_nn_search_pt2pl(rt.ctx, m, q, max_search_distance, &prm, &np));
        if (!np.found) return res;
        mp2p_icp::point_plane_pair_t pr;
        for (int k = 0; k < 4; k++) pr.pl_global.plane.coefs[k] = np.plane[k];
        pr.pl_global.centroid = {np.centroid[0], np.centroid[1], np.centroid[2]};
        pr.pt_local           = point;
        res.pairing           = pr;
        res.distance          = np.distance;
        return res;
    }
};

uint8_t PointsMapPlanes::serializeGetVersion() const { return 0; }
void    PointsMapPlanes::serializeTo(mrpt::serialization::CArchive& out) const
{
    out << knn << minimumPlanePoints << planeEigenThreshold << searchRadius;
    mrpt::maps::CSimplePointsMap::serializeTo(out);
}
void PointsMapPlanes::serializeFrom(mrpt::serialization::CArchive& in, uint8_t version)
{
    if (version != 0) MRPT_THROW_UNKNOWN_SERIALIZATION_VERSION(version);
    in >> knn >> minimumPlanePoints >> planeEigenThreshold >> searchRadius;
    uint8_t base_version = 0;
    in >> base_version;
    mrpt::maps::CSimplePointsMap::serializeFrom(in, base_version);
}

// ================================================================================================
// Pairings -> the device handle the solvers read (mp2p_hip_host::pairings_to_device decides whether
// the list this plugin's matchers left in HBM is the list handed over)
static mp2p_hip_pairs* pairings_to_device(Runtime& rt, const mp2p_icp::Pairings& pairings)
{
    const size_t                     n2 = pairings.paired_pt2pl.size();
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
        for (int k = 0; k < 3; k++) pp[i].c_global[k] = q.p_global.centroid[k], pp[i].c_local[k] = q.p_local.centroid[k];
    }
    return mp2p_hip_host::pairings_to_device(
        rt, reinterpret_cast<const mp2p_hip_pair_pt2pt*>(pairings.paired_pt2pt.data()), pairings.paired_pt2pt.size(),
        pl.data(), pl.size(), ln.data(), ln.size(), pp.data(), pp.size());
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
        mp2p_hip_host::RoctxRange roctx_range("align.3.2_solvers");  // the reference's profiler section (ICP.cpp:162)
        checkAllParametersAreRealized();
        out = mp2p_icp::OptimalTF_Result();
        ASSERT_(sc.guessRelativePose.has_value());
        if (!pairings.paired_ln2ln.empty())
            THROW_EXCEPTION("HIP Gauss-Newton: paired_ln2ln is not supported");
        if (pairings.point_weights.size() > MP2P_HIP_MAX_WEIGHT_BLOCKS)
            THROW_EXCEPTION("HIP Gauss-Newton: more than 32 point_weights blocks are not supported");
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
        p.n_weight_blocks = static_cast<uint32_t>(pairings.point_weights.size());
        for (size_t i = 0; i < pairings.point_weights.size(); i++)
            p.weight_block_count[i] = pairings.point_weights[i].first, p.weight_block_w[i] = pairings.point_weights[i].second;
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
        out.optimalPose = make_pose(res.pose);
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
        mp2p_hip_host::RoctxRange roctx_range("align.3.2_solvers");  // the reference's profiler section (ICP.cpp:162)
        out = mp2p_icp::OptimalTF_Result();
        if (!pairings.paired_ln2ln.empty()) THROW_EXCEPTION("HIP Horn: paired_ln2ln is not supported");
        if (pairings.point_weights.size() > MP2P_HIP_MAX_WEIGHT_BLOCKS) THROW_EXCEPTION("HIP Horn: more than 32 point_weights blocks are not supported");
        auto&                 rt  = Runtime::get();
        const mp2p_hip_pairs* eff = pairings_to_device(rt, pairings);
        const auto&           wp  = pairingsWeightParameters;
        const bool converted = !pairings.paired_pt2ln.empty() || !pairings.paired_pt2pl.empty();
        if (converted)
        {  // Solver_Horn.cpp:51-55: a fresh Pairings with the converted point pairings only
            ASSERT_(sc.guessRelativePose.has_value());
            mp2p_hip_pairs* conv = rt.conv_pairs(pairings.paired_pt2ln.size() + pairings.paired_pt2pl.size());
            double          T[12];
            fill_pose(mrpt::poses::CPose3D(sc.guessRelativePose.value()), T);
            rt.check(mp2p_hip_pairs_pt2ln_pl_to_pt2pt(rt.ctx, eff, T, conv));
            eff = conv;
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
        out.optimalPose = make_pose(res.pose);
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


// ================================================================================================
// Matcher_Points_InlierRatio (Matcher_Points_InlierRatio.cpp:41-143): same YAML key, same asserts
class Matcher_Points_InlierRatio : public mp2p_icp::Matcher_Points_Base, protected MatcherCallContext
{
    DEFINE_MRPT_OBJECT(Matcher_Points_InlierRatio, mp2p_icp_hip)
   public:
    void initialize(const mrpt::containers::yaml& params) override
    {
        Matcher_Points_Base::initialize(params);
        MCP_LOAD_REQ(params, inliersRatio);  // :38
    }
    bool match(const mp2p_icp::metric_map_t& pcGlobal, const mp2p_icp::metric_map_t& pcLocal,
               const mrpt::poses::CPose3D& localPose, const mp2p_icp::MatchContext& mc, mp2p_icp::MatchState& ms,
               mp2p_icp::Pairings& out) const override
    {
        mp2p_hip_host::RoctxRange roctx_range("align.3.1_matchers");  // the reference's profiler section (ICP.cpp:141)
        note_call(mc, ms);
        return mp2p_icp::Matcher::match(pcGlobal, pcLocal, localPose, mc, ms, out);
    }
    double inliersRatio = 0.80;

   private:
    void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                           const mrpt::poses::CPose3D& localPose, mp2p_icp::MatchState& ms,
                           const mp2p_icp::layer_name_t& globalName, const mp2p_icp::layer_name_t& localName,
                           mp2p_icp::Pairings& out) const override
    {
        ASSERT_GT_(inliersRatio, 0.0);  // :47-48
        ASSERT_LT_(inliersRatio, 1.0);
        const auto* gl = mp2p_icp::MapToPointsMap(pcGlobal);
        if (!gl) THROW_EXCEPTION("HIP matcher: the global layer must be a CPointsMap");
        out.potential_pairings += pcLocal.size();           // :53
        if (pcGlobal.isEmpty() || pcLocal.empty()) return;  // :56
        mp2p_hip_inlier_ratio_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.inliersRatio                            = inliersRatio;
        prm.allowMatchAlreadyMatchedPoints          = allowMatchAlreadyMatchedPoints_;
        prm.allowMatchAlreadyMatchedGlobalPoints    = allowMatchAlreadyMatchedGlobalPoints_;
        prm.bounding_box_intersection_check_epsilon = bounding_box_intersection_check_epsilon_;
        double T[12];
        fill_pose(localPose, T);
        auto&                       rt    = Runtime::get();
        const std::vector<uint32_t> visit = visit_list(pcLocal.size(), maxLocalPointsPerLayer_, localPointsSampleSeed_);
        mp2p_hip_map*               m;
        mp2p_hip_cloud*             c;
        layers(rt, *gl, pcLocal, m, c);
        BitAccess gbits(ms.globalPairedBitField.point_layers.at(globalName), gl->size());
        BitAccess lbits(ms.localPairedBitField.point_layers.at(localName), pcLocal.size());
        MatchCall call;
        call.ms_key = cur_ms_ ? cur_ms_ : &ms, call.iteration = cur_iteration_;
        call.gbits = gbits.view(), call.lbits = lbits.view();
        local_arrays(call, pcLocal);
        mp2p_hip_host::match_inlier_ratio_layer(rt, call, m, c, T, prm, visit.data(), visit.size(), out.paired_pt2pt);
        gbits.commit(), lbits.commit();
    }
};

// ================================================================================================
// Matcher_Adaptive (Matcher_Adaptive.cpp:59-314): neighbour search + 50-bin histogram and the selection on the
// device; the threshold between them by MRPT's OWN confidenceIntervalsFromHistogram (:203-205) from the 50 bins --
// the one step whose arithmetic this repository cannot pin is thereby the reference's.  (The binning rule and
// getHistogramNormalized -- linspace of the bin positions, counts / (count x bin width) -- are restated: two lines.)
class Matcher_Adaptive : public mp2p_icp::Matcher_Points_Base, protected MatcherCallContext
{
    DEFINE_MRPT_OBJECT(Matcher_Adaptive, mp2p_icp_hip)
   public:
    void initialize(const mrpt::containers::yaml& params) override
    {
        Matcher_Points_Base::initialize(params);
        MCP_LOAD_REQ(params, confidenceInterval);         // Matcher_Adaptive.cpp:36-49
        MCP_LOAD_REQ(params, firstToSecondDistanceMax);
        MCP_LOAD_REQ(params, absoluteMaxSearchDistance);
        MCP_LOAD_OPT(params, minimumCorrDist);
        MCP_LOAD_REQ(params, enableDetectPlanes);
        MCP_LOAD_OPT(params, planeSearchPoints);
        MCP_LOAD_OPT(params, planeMinimumFoundPoints);
        MCP_LOAD_OPT(params, planeEigenThreshold);
        MCP_LOAD_OPT(params, maxPt2PtCorrespondences);
        MCP_LOAD_OPT(params, planeMinimumDistance);
        ASSERT_LT_(confidenceInterval, 1.0);  // :51-57
        ASSERT_GT_(confidenceInterval, 0.0);
        ASSERT_GE_(planeSearchPoints, planeMinimumFoundPoints);
        ASSERT_GE_(planeMinimumFoundPoints, 3u);
        ASSERT_GT_(planeEigenThreshold, 0.0);
    }
    bool match(const mp2p_icp::metric_map_t& pcGlobal, const mp2p_icp::metric_map_t& pcLocal,
               const mrpt::poses::CPose3D& localPose, const mp2p_icp::MatchContext& mc, mp2p_icp::MatchState& ms,
               mp2p_icp::Pairings& out) const override
    {
        mp2p_hip_host::RoctxRange roctx_range("align.3.1_matchers");  // the reference's profiler section (ICP.cpp:141)
        note_call(mc, ms);
        return mp2p_icp::Matcher::match(pcGlobal, pcLocal, localPose, mc, ms, out);
    }
    double   confidenceInterval        = 0.80;
    double   firstToSecondDistanceMax  = 1.2;
    double   absoluteMaxSearchDistance = 5.0;
    bool     enableDetectPlanes        = false;
    uint32_t maxPt2PtCorrespondences   = 1;
    uint32_t planeSearchPoints         = 8;
    uint32_t planeMinimumFoundPoints   = 4;
    double   planeMinimumDistance      = 0.10;
    double   planeEigenThreshold       = 0.01;
    double   minimumCorrDist           = 0.1;

   private:
    void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                           const mrpt::poses::CPose3D& localPose, mp2p_icp::MatchState& ms,
                           const mp2p_icp::layer_name_t& globalName, const mp2p_icp::layer_name_t& localName,
                           mp2p_icp::Pairings& out) const override
    {
        const auto* gl = mp2p_icp::MapToPointsMap(pcGlobal);
        if (!gl) THROW_EXCEPTION("HIP matcher: the global layer must be a CPointsMap");
        out.potential_pairings += pcLocal.size() * maxPt2PtCorrespondences;  // :75
        if (pcGlobal.isEmpty() || pcLocal.empty()) return;
        if (maxLocalPointsPerLayer_ != 0 && pcLocal.size() > maxLocalPointsPerLayer_)
            THROW_EXCEPTION("Matcher_Adaptive does not support maxLocalPointsPerLayer (the reference indexes out of range there)");
        mp2p_hip_adaptive_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.confidenceInterval = confidenceInterval, prm.firstToSecondDistanceMax = firstToSecondDistanceMax;
        prm.absoluteMaxSearchDistance = absoluteMaxSearchDistance, prm.minimumCorrDist = minimumCorrDist;
        prm.enableDetectPlanes = enableDetectPlanes ? 1 : 0, prm.maxPt2PtCorrespondences = maxPt2PtCorrespondences;
        prm.planeSearchPoints = planeSearchPoints, prm.planeMinimumFoundPoints = planeMinimumFoundPoints;
        prm.planeMinimumDistance = planeMinimumDistance, prm.planeEigenThreshold = planeEigenThreshold;
        prm.allowMatchAlreadyMatchedPoints          = allowMatchAlreadyMatchedPoints_;
        prm.allowMatchAlreadyMatchedGlobalPoints    = allowMatchAlreadyMatchedGlobalPoints_;
        prm.bounding_box_intersection_check_epsilon = bounding_box_intersection_check_epsilon_;
        double T[12];
        fill_pose(localPose, T);
        auto&           rt = Runtime::get();
        mp2p_hip_map*   m;
        mp2p_hip_cloud* c;
        layers(rt, *gl, pcLocal, m, c);
        BitAccess gbits(ms.globalPairedBitField.point_layers.at(globalName), gl->size());
        BitAccess lbits(ms.localPairedBitField.point_layers.at(localName), pcLocal.size());
        MatchCall call;
        call.ms_key = cur_ms_ ? cur_ms_ : &ms, call.iteration = cur_iteration_;
        call.gbits = gbits.view(), call.lbits = lbits.view();
        local_arrays(call, pcLocal);
        const double ci = 1.0 - confidenceInterval;
        mp2p_hip_host::match_adaptive_layer(
            rt, call, m, c, T, prm,
            [&](const mp2p_hip_adaptive_hist& h)
            {
                // CHistogram::getHistogramNormalized (bin positions = linspace(min, max, 50), hits = count_i /
                // (total x bin width)), then MRPT's own confidence limits (:197-205)
                const size_t        nb = MP2P_HIP_ADAPTIVE_BINS;
                std::vector<double> xs(nb), vs(nb);
                const double        lo = h.minSqr, hi = h.maxSqr, bw = (hi - lo) / nb;
                for (size_t i = 0; i < nb; i++)
                {
                    xs[i] = lo + (hi - lo) * (double)i / (double)(nb - 1);
                    vs[i] = (h.count && bw > 0) ? (double)h.bins[i] / ((double)h.count * bw) : 0.0;
                }
                double ci_low = 0, ci_high = 0;
                mrpt::math::confidenceIntervalsFromHistogram(xs, vs, ci_low, ci_high, ci);
                return ci_high;
            },
            out.paired_pt2pt,
            [&](const mp2p_hip_pair_pt2pl& r)
            {
                auto& p = out.paired_pt2pl.emplace_back();
                p.pt_local = {r.pt_local[0], r.pt_local[1], r.pt_local[2]};
                for (int k = 0; k < 4; k++) p.pl_global.plane.coefs[k] = r.plane[k];
                p.pl_global.centroid = {r.centroid[0], r.centroid[1], r.centroid[2]};
            });
        gbits.commit(), lbits.commit();
    }
};

// ================================================================================================
// FilterDecimateVoxels (FilterDecimateVoxels.cpp:41-381) deriving the reference's FilterBase: same YAML keys; the
// voxel decimation on the device.  RandomPoint (mrpt::random stream) is handed to the reference's own class.
class FilterDecimateVoxels : public mp2p_icp_filters::FilterBase
{
    DEFINE_MRPT_OBJECT(FilterDecimateVoxels, mp2p_icp_hip)
   public:
    void initialize(const mrpt::containers::yaml& c) override
    {
        ASSERTMSG_(c.has("input_pointcloud_layer"), "YAML configuration must have an entry `input_pointcloud_layer` with a scalar or sequence.");  // :43-60
        input_pointcloud_layer.clear();
        auto cfgIn = c["input_pointcloud_layer"];
        if (cfgIn.isScalar()) input_pointcloud_layer.push_back(cfgIn.as<std::string>());
        else
        {
            ASSERTMSG_(cfgIn.isSequence(), "YAML configuration must have an entry `input_pointcloud_layer` with a scalar or sequence.");
            for (int i = 0; i < (int)cfgIn.size(); i++) input_pointcloud_layer.push_back(cfgIn(i).as<std::string>());
        }
        ASSERT_(!input_pointcloud_layer.empty());
        MCP_LOAD_OPT(c, error_on_missing_input_layer);  // :65-78
        std::string decimate_method = c["decimate_method"].as<std::string>();
        method = decimate_method.find("ClosestToAverage") != std::string::npos ? mp2p_icp_filters::DecimateMethod::ClosestToAverage
                 : decimate_method.find("VoxelAverage") != std::string::npos   ? mp2p_icp_filters::DecimateMethod::VoxelAverage
                 : decimate_method.find("RandomPoint") != std::string::npos    ? mp2p_icp_filters::DecimateMethod::RandomPoint
                                                                               : mp2p_icp_filters::DecimateMethod::FirstPoint;
        MCP_LOAD_REQ(c, output_pointcloud_layer);
        MCP_LOAD_OPT(c, minimum_input_points_to_filter);
        DECLARE_PARAMETER_IN_REQ(c, voxel_filter_resolution, *this);
        if (c.has("flatten_to")) flatten_to = c["flatten_to"].as<double>();
        if (method == mp2p_icp_filters::DecimateMethod::RandomPoint)
        {
            reference_ = std::make_shared<mp2p_icp_filters::FilterDecimateVoxels>();
            reference_->initialize(c);
        }
    }

    void filter(mp2p_icp::metric_map_t& inOut) const override
    {
        if (reference_) return reference_->filter(inOut);
        checkAllParametersAreRealized();
        std::vector<const mrpt::maps::CPointsMap*> in;  // :112-139
        for (const auto& name : input_pointcloud_layer)
        {
            auto it = inOut.layers.find(name);
            if (it == inOut.layers.end())
            {
                if (error_on_missing_input_layer) THROW_EXCEPTION_FMT("Input layer '%s' not found on input map.", name.c_str());
                continue;
            }
            const auto* pc = mp2p_icp::MapToPointsMap(*it->second);
            if (!pc) THROW_EXCEPTION_FMT("Layer '%s' must be of point cloud type.", name.c_str());
            in.push_back(pc);
        }
        ASSERT_(!in.empty());
        ASSERT_(!output_pointcloud_layer.empty());
        auto outPc = mp2p_icp_filters::GetOrCreatePointLayer(inOut, output_pointcloud_layer, false,
                                                             in.at(0)->GetRuntimeClass()->className);  // :145-151
        // layers below minimum_input_points_to_filter pass through undecimated (:155-189)
        std::vector<const mrpt::maps::CPointsMap*> todo;
        for (const auto* pc : in)
        {
            if (minimum_input_points_to_filter == 0 || pc->size() > minimum_input_points_to_filter)
            {
                todo.push_back(pc);
                continue;
            }
            const auto& xs = pc->getPointsBufferRef_x();
            const auto& ys = pc->getPointsBufferRef_y();
            for (size_t i = 0; i < xs.size(); i++)
            {
                if (flatten_to.has_value()) outPc->insertPointFast(xs[i], ys[i], (float)*flatten_to);
                else outPc->insertPointFrom(*pc, i);
            }
        }
        if (todo.empty()) return;
        if (todo.size() > 1 && method != mp2p_icp_filters::DecimateMethod::FirstPoint)
            THROW_EXCEPTION("HIP FilterDecimateVoxels: several input layers are decimated together for FirstPoint only");
        // the input layers in order, as one array triple (one layer: its own buffers)
        std::vector<float> cx, cy, cz;
        const float *      px = nullptr, *py = nullptr, *pz = nullptr;
        size_t             n = 0;
        if (todo.size() == 1)
        {
            px = todo[0]->getPointsBufferRef_x().data(), py = todo[0]->getPointsBufferRef_y().data(),
            pz = todo[0]->getPointsBufferRef_z().data(), n = todo[0]->size();
        }
        else
        {
            for (const auto* pc : todo)
            {
                const auto &xs = pc->getPointsBufferRef_x(), &ys = pc->getPointsBufferRef_y(), &zs = pc->getPointsBufferRef_z();
                cx.insert(cx.end(), xs.begin(), xs.end()), cy.insert(cy.end(), ys.begin(), ys.end()), cz.insert(cz.end(), zs.begin(), zs.end());
            }
            px = cx.data(), py = cy.data(), pz = cz.data(), n = cx.size();
        }
        mp2p_hip_decimate_params prm;
        std::memset(&prm, 0, sizeof(prm));
        prm.voxel_filter_resolution = voxel_filter_resolution;
        prm.decimate_method         = method == mp2p_icp_filters::DecimateMethod::ClosestToAverage ? MP2P_HIP_DECIMATE_CLOSEST_TO_AVERAGE
                                      : method == mp2p_icp_filters::DecimateMethod::VoxelAverage   ? MP2P_HIP_DECIMATE_VOXEL_AVERAGE
                                                                                                   : MP2P_HIP_DECIMATE_FIRST_POINT;
        prm.has_flatten_to = flatten_to.has_value() ? 1 : 0, prm.flatten_to = flatten_to.has_value() ? (float)*flatten_to : 0.f;
        std::vector<float>    ox, oy, oz;
        std::vector<uint32_t> src;
        const size_t          m = mp2p_hip_host::filter_decimate(Runtime::get(), px, py, pz, n, prm, ox, oy, oz, src);
        outPc->reserve(outPc->size() + m);
        for (size_t k = 0; k < m; k++)
        {
            // a picked input point keeps its extra fields (ring, intensity, timestamp: insertPointFrom, :236-244); an
            // average or a flattened point is coordinates only
            if (src[k] != 0xFFFFFFFFu && !flatten_to.has_value())
            {
                size_t i = src[k], li = 0;
                while (li + 1 < todo.size() && i >= todo[li]->size()) i -= todo[li]->size(), li++;
                outPc->insertPointFrom(*todo[li], i);
            }
            else
                outPc->insertPointFast(ox[k], oy[k], oz[k]);
        }
        outPc->mark_as_modified();
    }

    std::vector<std::string>         input_pointcloud_layer{"raw"};
    bool                             error_on_missing_input_layer = true;
    std::string                      output_pointcloud_layer;
    float                            voxel_filter_resolution       = 1.0f;
    uint32_t                         minimum_input_points_to_filter = 0;
    std::optional<double>            flatten_to;
    mp2p_icp_filters::DecimateMethod method = mp2p_icp_filters::DecimateMethod::FirstPoint;

   private:
    std::shared_ptr<mp2p_icp_filters::FilterDecimateVoxels> reference_;
};

// ================================================================================================
// QualityEvaluator_PairedRatio (QualityEvaluator_PairedRatio.cpp:22-75) with the HIP matcher as its private matcher.
// The reference's class holds a concrete mp2p_icp::Matcher_Points_DistanceThreshold BY VALUE (QualityEvaluator_PairedRatio.h:60)
// and calls it, when reuse_icp_pairings is false, at every quality checkpoint and at the end of ICP::align (ICP.cpp:259-283,
// 322-324): with the stock evaluator in the YAML the KD-tree of the 10 M-point map would be built after all, through that
// side door.  Quality evaluators are created by class name with an optional `plugin:` exactly like matchers
// (ICP.cpp:559-601):
//     quality:
//       - class: mp2p_icp_hip::QualityEvaluator_PairedRatio
//         plugin: libmp2p_icp_hip_plugin.so
//         params: { reuse_icp_pairings: false, thresholdDistance: 0.10, thresholdAngularDeg: 0 }   # + the matcher's keys
// Same YAML keys, same defaults (reuse_icp_pairings = true, absolute_minimum_pairing_ratio = 0.20, and -- for the
// private matcher -- allowMatchAlreadyMatchedGlobalPoints = true unless the YAML says otherwise, :33-38).
class QualityEvaluator_PairedRatio : public mp2p_icp::QualityEvaluator
{
    DEFINE_MRPT_OBJECT(QualityEvaluator_PairedRatio, mp2p_icp_hip)
   public:
    void initialize(const mrpt::containers::yaml& params) override
    {
        MCP_LOAD_OPT(params, reuse_icp_pairings);              // :28
        MCP_LOAD_OPT(params, absolute_minimum_pairing_ratio);  // :29
        if (!reuse_icp_pairings)
        {
            // "in quality assesment, it DOES make sense to count several times the same global point" (:33-38)
            mrpt::containers::yaml p = params;
            if (!p.has("allowMatchAlreadyMatchedGlobalPoints")) p["allowMatchAlreadyMatchedGlobalPoints"] = true;
            matcher_.initialize(p);
        }
    }
    Result evaluate(const mp2p_icp::metric_map_t& pcGlobal, const mp2p_icp::metric_map_t& pcLocal,
                    const mrpt::poses::CPose3D& localPose, const mp2p_icp::Pairings& pairingsFromICP) const override
    {
        uint64_t n_pairs = 0, potential = 0;
        if (reuse_icp_pairings)  // :51-55
            n_pairs = pairingsFromICP.size(), potential = pairingsFromICP.potential_pairings;
        else
        {
            mp2p_icp::MatchState ms(pcGlobal, pcLocal);  // :59
            if (!matcher_.count_pairs(pcGlobal, pcLocal, localPose, ms, n_pairs, potential))
            {
                // global re-use forbidden in the YAML: the layer pairs are coupled through the marks -> the whole match
                mp2p_icp::Pairings newPairings;
                matcher_.match(pcGlobal, pcLocal, localPose, {}, ms, newPairings);  // :60
                n_pairs = newPairings.size(), potential = newPairings.potential_pairings;
            }
        }
        Result r;
        r.quality      = potential ? (double)n_pairs / (double)potential : .0;  // :67-70
        r.hard_discard = r.quality < absolute_minimum_pairing_ratio;           // :72
        return r;
    }
    void attachToParameterSource(mp2p_icp::ParameterSource& source) override  // QualityEvaluator_PairedRatio.h:53-57
    {
        source.attach(*this);
        source.attach(matcher_);
    }

   private:
    Matcher_Points_DistanceThreshold matcher_;
    bool                             reuse_icp_pairings             = true;
    double                           absolute_minimum_pairing_ratio = 0.20;
};

// a caller that edits a layer in place between the iterations of its OWN loop (ICP::align never does)
// ... and one that is done with a layer (a scan's maps at the end of ICP::align): frees its device copy now; the
// cache is bounded anyway (mp2p_hip_host::Runtime::max_layers / byte_budget, least recently used first)
void release_layer(const void* layer) { Runtime::get().release_layer(layer); }

void invalidate_layers() { Runtime::get().invalidate_layers(); }

IMPLEMENTS_MRPT_OBJECT(Matcher_Points_DistanceThreshold, mp2p_icp::Matcher, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Matcher_Point2Plane, mp2p_icp::Matcher, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Matcher_Points_InlierRatio, mp2p_icp::Matcher, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Matcher_Adaptive, mp2p_icp::Matcher, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(FilterDecimateVoxels, mp2p_icp_filters::FilterBase, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(QualityEvaluator_PairedRatio, mp2p_icp::QualityEvaluator, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Solver_Horn, mp2p_icp::Solver, mp2p_icp_hip)
IMPLEMENTS_MRPT_OBJECT(Solver_GaussNewton, mp2p_icp::Solver, mp2p_icp_hip)
IMPLEMENTS_SERIALIZABLE(PointsMapPlanes, CSimplePointsMap, mp2p_icp_hip)

}  // namespace mp2p_icp_hip

MRPT_INITIALIZER(register_mp2p_icp_hip)
{
    using mrpt::rtti::registerClass;
    registerClass(CLASS_ID(mp2p_icp_hip::Matcher_Points_DistanceThreshold));
    registerClass(CLASS_ID(mp2p_icp_hip::Matcher_Point2Plane));
    registerClass(CLASS_ID(mp2p_icp_hip::Matcher_Points_InlierRatio));
    registerClass(CLASS_ID(mp2p_icp_hip::Matcher_Adaptive));
    registerClass(CLASS_ID(mp2p_icp_hip::FilterDecimateVoxels));
    registerClass(CLASS_ID(mp2p_icp_hip::QualityEvaluator_PairedRatio));
    registerClass(CLASS_ID(mp2p_icp_hip::Solver_GaussNewton));
    registerClass(CLASS_ID(mp2p_icp_hip::Solver_Horn));
    registerClass(CLASS_ID(mp2p_icp_hip::PointsMapPlanes));
}
