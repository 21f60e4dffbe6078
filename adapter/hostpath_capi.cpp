// hostpath_capi.cpp -- a C ABI over adapter/mp2p_hip_host.hpp for tests/ and bench.py.
//
// The plugin (mp2p_hip_plugin.cpp) needs MRPT and mp2p_icp to build; its per-call logic lives in
// mp2p_hip_host.hpp, free of MRPT types.  This file instantiates that logic on plain host containers
// -- exactly the containers the plugin would view: SoA float buffers, std::vector<bool>-style packed
// bit-fields, a vector of 36-byte pair records -- so that the host path of the drop-in boundary is
// compiled, parity-tested (tests/test_gpu_boundary_hostpath.py) and timed (bench.py, "host_boundary")
// in this repository.  A session plays the caller's part of ICP::align (ICP.cpp:123-308):
//   run_matchers : fresh MatchState (Matcher.cpp:57-66), then Matcher::match per matcher
//   run_solvers  : the solver is handed the host Pairings
// Built by __graft_entry__.build() with g++ (no HIP code here): libmp2p_hip_hostpath.so.
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "mp2p_hip_host.hpp"

using namespace mp2p_hip_host;

namespace
{
thread_local std::string g_err;

struct Session
{
    const float *gx, *gy, *gz, *lx, *ly, *lz;  // caller-owned (CPointsMap buffers)
    size_t       ng, nl;
    std::vector<uint64_t>            gbits, lbits;  // MatchState::{global,local}PairedBitField of the layer pair
    std::vector<mp2p_hip_pair_pt2pt> pt2pt;         // Pairings::paired_pt2pt
    std::vector<mp2p_hip_pair_pt2pl> pt2pl;         // Pairings::paired_pt2pl (as 72-byte records)
    uint64_t                         potential = 0;
    int                              dummy_ms  = 0;  // its address stands for the MatchState object
    double                           last_ms[4] = {0, 0, 0, 0};  // wall time of the last calls [match, solve]
};

template <class F>
int guarded(F&& f)
{
    try
    {
        f();
        return 0;
    }
    catch (const std::exception& e)
    {
        g_err = e.what();
        return -1;
    }
}
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

const char* mp2p_hostpath_last_error(void) { return g_err.c_str(); }

void* mp2p_hostpath_open(const float* gx, const float* gy, const float* gz, size_t ng, const float* lx,
                         const float* ly, const float* lz, size_t nl)
{
    auto* s = new Session();
    s->gx = gx, s->gy = gy, s->gz = gz, s->ng = ng;
    s->lx = lx, s->ly = ly, s->lz = lz, s->nl = nl;
    s->gbits.assign((ng + 63) / 64 + 1, 0), s->lbits.assign((nl + 63) / 64 + 1, 0);
    return s;
}
void mp2p_hostpath_close(void* h) { delete static_cast<Session*>(h); }

// the caller's part of run_matchers: a fresh MatchState and an empty aggregate Pairings
int mp2p_hostpath_begin_iteration(void* h)
{
    auto* s = static_cast<Session*>(h);
    std::fill(s->gbits.begin(), s->gbits.end(), 0);  // pointcloud_bitfield_t::initialize_from: assign(n, false)
    std::fill(s->lbits.begin(), s->lbits.end(), 0);
    s->pt2pt.clear(), s->pt2pl.clear(), s->potential = 0;
    return 0;
}

int mp2p_hostpath_match_pt2pt(void* h, const double pose[12], const mp2p_hip_pt2pt_params* prm,
                              uint32_t icp_iteration, const uint32_t* visit, size_t n_visit, size_t* n_added)
{
    mp2p_hip_host::RoctxRange range("align.3.1_matchers");
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            const double t0 = now_ms();
            Runtime&     rt = Runtime::get();
            MatchCall    c;
            c.ms_key = &s->dummy_ms, c.iteration = icp_iteration;
            c.gbits = BitView{s->gbits.data(), s->ng}, c.lbits = BitView{s->lbits.data(), s->nl};
            c.lx = s->lx, c.ly = s->ly, c.lz = s->lz, c.n_local = s->nl;
            s->potential += (uint64_t)s->nl * prm->pairingsPerPoint;  // :64
            size_t n = 0;
            if (s->ng && s->nl)
            {
                mp2p_hip_map*   m = rt.global_layer(s->gx, s->gx, s->gy, s->gz, s->ng, icp_iteration == 0);
                mp2p_hip_cloud* l = rt.local_layer(s->lx, s->lx, s->ly, s->lz, s->nl, icp_iteration == 0);
                n = match_pt2pt_layer(rt, c, m, l, pose, *prm, visit, n_visit, s->pt2pt);
            }
            if (n_added) *n_added = n;
            s->last_ms[0] = now_ms() - t0;
        });
}

int mp2p_hostpath_match_pt2pl(void* h, const double pose[12], const mp2p_hip_pt2pl_params* prm,
                              uint32_t icp_iteration, size_t* n_added)
{
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            const double t0 = now_ms();
            Runtime&     rt = Runtime::get();
            MatchCall    c;
            c.ms_key = &s->dummy_ms, c.iteration = icp_iteration;
            c.gbits = BitView{s->gbits.data(), s->ng}, c.lbits = BitView{s->lbits.data(), s->nl};
            c.lx = s->lx, c.ly = s->ly, c.lz = s->lz, c.n_local = s->nl;
            s->potential += (uint64_t)s->nl;  // Matcher_Point2Plane.cpp:54
            size_t n = 0;
            if (s->ng && s->nl)
            {
                mp2p_hip_map*   m = rt.global_layer(s->gx, s->gx, s->gy, s->gz, s->ng, icp_iteration == 0);
                mp2p_hip_cloud* l = rt.local_layer(s->lx, s->lx, s->ly, s->lz, s->nl, icp_iteration == 0);
                n = match_pt2pl_layer(rt, c, m, l, pose, *prm, nullptr, 0,
                                      [&](const mp2p_hip_pair_pt2pl& r) { s->pt2pl.push_back(r); });
            }
            if (n_added) *n_added = n;
            s->last_ms[0] = now_ms() - t0;
        });
}

// QualityEvaluator_PairedRatio::evaluate, reuse_icp_pairings = false (QualityEvaluator_PairedRatio.cpp:45-73), as the plugin's
// mp2p_icp_hip::QualityEvaluator_PairedRatio does it: the private matcher on a fresh MatchState, only the two counts come
// back.  *quality = pairs / potential_pairings (0 when nothing could pair), *hard_discard = quality < absolute_minimum
int mp2p_hostpath_quality_paired_ratio(void* h, const double pose[12], const mp2p_hip_pt2pt_params* prm,
                                       double absolute_minimum_pairing_ratio, double* quality, int* hard_discard,
                                       size_t* n_pairs)
{
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            const double   t0        = now_ms();
            Runtime&       rt        = Runtime::get();
            const uint64_t potential = (uint64_t)s->nl * prm->pairingsPerPoint;  // Matcher_Points_DistanceThreshold.cpp:64
            size_t         n         = 0;
            if (s->ng && s->nl)
            {
                // matcher_.match(pcGlobal, pcLocal, localPose, {}, ...): MatchContext{} = ICP iteration 0 (:60)
                mp2p_hip_map*   m = rt.global_layer(s->gx, s->gx, s->gy, s->gz, s->ng, true);
                mp2p_hip_cloud* l = rt.local_layer(s->lx, s->lx, s->ly, s->lz, s->nl, true);
                n = count_pt2pt_layer(rt, m, l, pose, *prm, nullptr, 0);
            }
            const double q = potential ? (double)n / (double)potential : 0.0;  // :67-70
            if (quality) *quality = q;
            if (hard_discard) *hard_discard = q < absolute_minimum_pairing_ratio;  // :72
            if (n_pairs) *n_pairs = n;
            s->last_ms[0] = now_ms() - t0;
        });
}

// Matcher_Points_InlierRatio (Matcher_Points_InlierRatio.cpp:41-143) through the host layer
int mp2p_hostpath_match_inlier_ratio(void* h, const double pose[12], const mp2p_hip_inlier_ratio_params* prm,
                                     uint32_t icp_iteration, const uint32_t* visit, size_t n_visit, size_t* n_added)
{
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            const double t0 = now_ms();
            Runtime&     rt = Runtime::get();
            MatchCall    c;
            c.ms_key = &s->dummy_ms, c.iteration = icp_iteration;
            c.gbits = BitView{s->gbits.data(), s->ng}, c.lbits = BitView{s->lbits.data(), s->nl};
            c.lx = s->lx, c.ly = s->ly, c.lz = s->lz, c.n_local = s->nl;
            s->potential += (uint64_t)s->nl;  // :53
            size_t n = 0;
            if (s->ng && s->nl)
            {
                mp2p_hip_map*   m = rt.global_layer(s->gx, s->gx, s->gy, s->gz, s->ng, icp_iteration == 0);
                mp2p_hip_cloud* l = rt.local_layer(s->lx, s->lx, s->ly, s->lz, s->nl, icp_iteration == 0);
                n = match_inlier_ratio_layer(rt, c, m, l, pose, *prm, visit, n_visit, s->pt2pt);
            }
            if (n_added) *n_added = n;
            s->last_ms[0] = now_ms() - t0;
        });
}

// Matcher_Adaptive (Matcher_Adaptive.cpp:59-314) through the host layer; the threshold step is this library's
// restatement of MRPT's histogram helpers here (the plugin calls MRPT's own); *ci_high_out: what it returned
int mp2p_hostpath_match_adaptive(void* h, const double pose[12], const mp2p_hip_adaptive_params* prm,
                                 uint32_t icp_iteration, size_t* n_pt_added, size_t* n_pl_added, double* ci_high_out)
{
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            const double t0 = now_ms();
            Runtime&     rt = Runtime::get();
            MatchCall    c;
            c.ms_key = &s->dummy_ms, c.iteration = icp_iteration;
            c.gbits = BitView{s->gbits.data(), s->ng}, c.lbits = BitView{s->lbits.data(), s->nl};
            c.lx = s->lx, c.ly = s->ly, c.lz = s->lz, c.n_local = s->nl;
            s->potential += (uint64_t)s->nl * prm->maxPt2PtCorrespondences;  // Matcher_Adaptive.cpp:75
            std::pair<size_t, size_t> n{0, 0};
            if (s->ng && s->nl)
            {
                mp2p_hip_map*   m = rt.global_layer(s->gx, s->gx, s->gy, s->gz, s->ng, icp_iteration == 0);
                mp2p_hip_cloud* l = rt.local_layer(s->lx, s->lx, s->ly, s->lz, s->nl, icp_iteration == 0);
                n = match_adaptive_layer(
                    rt, c, m, l, pose, *prm,
                    [&](const mp2p_hip_adaptive_hist& hist) { return mp2p_hip_adaptive_ci_high(&hist, prm->confidenceInterval); },
                    s->pt2pt, [&](const mp2p_hip_pair_pt2pl& r) { s->pt2pl.push_back(r); }, ci_high_out);
            }
            if (n_pt_added) *n_pt_added = n.first;
            if (n_pl_added) *n_pl_added = n.second;
            s->last_ms[0] = now_ms() - t0;
        });
}

// FilterDecimateVoxels (FilterDecimateVoxels.cpp:107-381) on the session's LOCAL layer through the host layer:
// the decimated points into caller-provided arrays of capacity n_local; returns their number in *n_out
int mp2p_hostpath_filter_decimate_local(void* h, const mp2p_hip_decimate_params* prm, float* ox, float* oy, float* oz,
                                        uint32_t* src, size_t* n_out)
{
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            Runtime&              rt = Runtime::get();
            std::vector<float>    x, y, z;
            std::vector<uint32_t> si;
            const size_t          m = filter_decimate(rt, s->lx, s->ly, s->lz, s->nl, *prm, x, y, z, si);
            std::memcpy(ox, x.data(), m * 4), std::memcpy(oy, y.data(), m * 4), std::memcpy(oz, z.data(), m * 4);
            if (src) std::memcpy(src, si.data(), m * 4);
            *n_out = m;
        });
}

// the layer cache of this thread's runtime: [0] cached layers, [1] cached device bytes, [2] evictions,
// [3] full-content checks, [4] re-seen (strided) checks; set max_layers (> 0) first when given
int mp2p_hostpath_cache(size_t max_layers, size_t out[5])
{
    return guarded(
        [&]()
        {
            Runtime& rt = Runtime::get();
            if (max_layers) rt.max_layers = max_layers;
            out[0] = rt.cached_layers(), out[1] = rt.cached_bytes(), out[2] = rt.n_evictions, out[3] = rt.n_full_checks,
            out[4] = rt.n_reseen_checks;
        });
}
int mp2p_hostpath_release_layers(void* h)
{
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            Runtime& rt = Runtime::get();
            rt.release_layer(s->gx), rt.release_layer(s->lx);
        });
}

// Solver_GaussNewton::impl_optimal_pose on the session's host Pairings
int mp2p_hostpath_solve_gn(void* h, const double pose0[12], const mp2p_hip_gn_params* prm, mp2p_hip_gn_result* out)
{
    mp2p_hip_host::RoctxRange range("align.3.2_solvers");
    auto* s = static_cast<Session*>(h);
    return guarded(
        [&]()
        {
            const double    t0 = now_ms();
            Runtime&        rt = Runtime::get();
            mp2p_hip_pairs* dp = pairings_to_device(rt, s->pt2pt.data(), s->pt2pt.size(), s->pt2pl.data(),
                                                    s->pt2pl.size(), nullptr, 0, nullptr, 0);
            rt.check(mp2p_hip_gn_solve(rt.ctx, dp, pose0, prm, out));
            s->last_ms[1] = now_ms() - t0;
        });
}

// a Pairings produced elsewhere (another matcher, a test): replaces the session's lists
int mp2p_hostpath_set_pairings(void* h, const mp2p_hip_pair_pt2pt* pt, size_t n_pt, const mp2p_hip_pair_pt2pl* pl,
                               size_t n_pl)
{
    auto* s = static_cast<Session*>(h);
    s->pt2pt.assign(pt, pt + n_pt), s->pt2pl.assign(pl, pl + n_pl);
    return 0;
}

const mp2p_hip_pair_pt2pt* mp2p_hostpath_pairs_pt2pt(void* h, size_t* n)
{
    auto* s = static_cast<Session*>(h);
    if (n) *n = s->pt2pt.size();
    return s->pt2pt.data();
}
const mp2p_hip_pair_pt2pl* mp2p_hostpath_pairs_pt2pl(void* h, size_t* n)
{
    auto* s = static_cast<Session*>(h);
    if (n) *n = s->pt2pl.size();
    return s->pt2pl.data();
}
uint64_t* mp2p_hostpath_bits(void* h, int which, size_t* nbits)
{
    auto* s = static_cast<Session*>(h);
    if (nbits) *nbits = which ? s->nl : s->ng;
    return which ? s->lbits.data() : s->gbits.data();
}
uint64_t mp2p_hostpath_potential(void* h) { return static_cast<Session*>(h)->potential; }

// [0] map uploads, [1] cloud uploads, [2] MatchState uploads, [3] Pairings uploads (of this thread's runtime)
int mp2p_hostpath_counters(size_t out[4])
{
    return guarded(
        [&]()
        {
            Runtime& rt = Runtime::get();
            out[0] = rt.n_map_uploads, out[1] = rt.n_cloud_uploads, out[2] = rt.n_mstate_uploads,
            out[3] = rt.n_pair_uploads;
        });
}
void mp2p_hostpath_last_ms(void* h, double out[2])
{
    auto* s = static_cast<Session*>(h);
    out[0] = s->last_ms[0], out[1] = s->last_ms[1];
}
// stages of the last matcher call of this thread's runtime (mp2p_hip_host::Runtime::stage_ms):
// {state in, device, resize, copy-out window, marks (inside that window), list fingerprint}
void mp2p_hostpath_stage_ms(double out[6])
{
    try
    {
        Runtime& rt = Runtime::get();
        for (int i = 0; i < 6; i++) out[i] = rt.stage_ms[i];
    }
    catch (...)
    {
    }
}
// Runtime::strict of this thread's runtime (MP2P_HIP_HOST_STRICT): every solver call uploads the host Pairings
void mp2p_hostpath_set_strict(int on)
{
    try
    {
        Runtime::get().strict = on != 0;
    }
    catch (...)
    {
    }
}
// Runtime::trust_reseen (MP2P_HIP_HOST_TRUST_RESEEN): re-seen layers are re-verified on a stride at ICP iteration 0
void mp2p_hostpath_set_trust_reseen(int on)
{
    try
    {
        Runtime::get().trust_reseen = on != 0;
    }
    catch (...)
    {
    }
}
void mp2p_hostpath_invalidate_layers(void)
{
    try
    {
        Runtime::get().invalidate_layers();
    }
    catch (...)
    {
    }
}

}  // extern "C"
