// matcher_inlier_ratio.hip -- Matcher_Points_InlierRatio::implMatchOneLayer
// (mp2p_icp/src/Matcher_Points_InlierRatio.cpp:40-143; used by tests/test-mp2p_icp_algos.cpp).
//
// Reference: the UNBOUNDED nearest neighbour of every visited local point goes into a
// std::multimap keyed by d2 (emplace_hint(begin()): among equal keys the LATER insertion comes
// first, :100-101); the nKeep = round(nTotal * inliersRatio) smallest are walked in that order, a
// pair is dropped if its global point is already marked (unless re-use is allowed), and both marks
// are set for every emitted pair (:123-139, unconditionally).
//
// Here: the search kernels of nn_query.hip with an infinite threshold (they expand until the
// nearest point is found), a 64-bit key (fp32 bits of d2 << 32 | ~visit rank) per found point, one
// radix sort, nKeep on the device, "first in sorted order wins" as an atomicMin of the sorted
// position, and the ordered compaction of pairs.hip over the sorted list.
#include "device_utils.hpp"

namespace mp2p
{
__global__ __launch_bounds__(256) void ir_keys_kernel(const uint32_t* __restrict__ nn_spos,
                                                      const float* __restrict__ nn_d2,
                                                      const uint32_t* __restrict__ pos,
                                                      const uint32_t* __restrict__ order, uint32_t n_visit,
                                                      unsigned long long* __restrict__ keys,
                                                      uint32_t* __restrict__ vals, uint32_t* __restrict__ n_found)
{
    const uint32_t r    = blockIdx.x * blockDim.x + threadIdx.x;
    const int      lane = threadIdx.x & 63;
    bool           found = false;
    if (r < n_visit)
    {
        const uint32_t i = order ? order[r] : r;
        const uint32_t q = pos[i];
        found            = nn_spos[q] != NONE_U32;
        // d2 >= 0: its bit pattern orders like the value; ~r: the later insertion first
        keys[r] = found ? (((unsigned long long)__float_as_uint(nn_d2[q]) << 32) | (unsigned long long)(~r)) : ~0ull;
        vals[r] = i;
    }
    // (one atomic per block, not per wave: 15 000 same-address atomics per 1 M-point layer are serialised at ~12 ns each)
    const unsigned long long m = __ballot(found);
    __shared__ uint32_t s_n[4];
    if (lane == 0) s_n[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const uint32_t tot = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        if (tot) atomicAdd(n_found, tot);
    }
}

// nKeep = mrpt::round(double(nTotal) * inliersRatio)   (:119; ties to even, as lrint does)
__global__ void ir_keep_kernel(const uint32_t* n_found, double ratio, uint32_t* n_keep)
{
    const double v = (double)*n_found * ratio;
    double       k = nearbyint(v);
    if (k < 0) k = 0;
    if (k > (double)*n_found) k = (double)*n_found;
    *n_keep = (uint32_t)k;
}

// first in sorted order wins the global point (:126-132); pre-marked global points stay unclaimed
__global__ __launch_bounds__(256) void ir_claim_kernel(const uint32_t* __restrict__ sorted_orig,
                                                       const uint32_t* __restrict__ n_keep,
                                                       const uint32_t* __restrict__ nn_spos,
                                                       const uint32_t* __restrict__ pos,
                                                       const float4* __restrict__ gpts,
                                                       const unsigned char* __restrict__ global_taken,
                                                       unsigned long long* claims, unsigned long long claim_hi)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *n_keep) return;
    const uint32_t spos = nn_spos[pos[sorted_orig[r]]];
    if (spos == NONE_U32) return;
    if (global_taken && global_taken[__float_as_uint(gpts[spos].w)]) return;
    atomicMin(&claims[spos], claim_hi | (unsigned long long)r);
}

int launch_match_inlier_ratio(mp2p_hip_ctx* ctx, const mp2p_hip_map* map, const mp2p_hip_cloud* cloud,
                              const double pose[12], const mp2p_hip_inlier_ratio_params* prm,
                              mp2p_hip_mstate* ms, mp2p_hip_pairs* out)
{
    // 1. unbounded nearest neighbour of every visited point (no claims, no global marks yet)
    mp2p_hip_pt2pt_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.threshold = INFINITY, sp.thresholdAngularDeg = 0.0, sp.pairingsPerPoint = 1;
    sp.allowMatchAlreadyMatchedPoints       = prm->allowMatchAlreadyMatchedPoints;
    sp.allowMatchAlreadyMatchedGlobalPoints = 1;
    sp.bounding_box_intersection_check_epsilon = prm->bounding_box_intersection_check_epsilon;
    int rc = launch_nn_pt2pt(ctx, map, cloud, pose, &sp, ms);
    if (rc) return rc;
    rc = launch_unpack_rec(ctx, cloud->n);  // the keys below read the plain [n_l] arrays
    if (rc) return rc;

    // 2. sort the found points by (d2, later insertion first)
    const size_t n_visit = cloud->n_visit ? cloud->n_visit : cloud->n;
    MP2P_REQUIRE_INT_COUNT(ctx, n_visit);
    Scratch<unsigned long long> k0, k1;
    Scratch<uint32_t>           v0, v1, cnt;
    Scratch<unsigned char>      tmp;
    MP2P_TRY_HIP(ctx, k0.take(ctx, 0, n_visit));
    MP2P_TRY_HIP(ctx, k1.take(ctx, 1, n_visit));
    MP2P_TRY_HIP(ctx, v0.take(ctx, 2, n_visit));
    MP2P_TRY_HIP(ctx, v1.take(ctx, 3, n_visit));
    MP2P_TRY_HIP(ctx, cnt.take(ctx, 4, 2));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(cnt.p, 0, 2 * sizeof(uint32_t), ctx->stream));
    const uint32_t nb = (uint32_t)((n_visit + 255) / 256);
    hipLaunchKernelGGL(ir_keys_kernel, dim3(nb), dim3(256), 0, ctx->stream, ctx->nn_spos.p, ctx->nn_d2.p,
                       cloud->pos.p, cloud->n_visit ? cloud->order.p : nullptr, (uint32_t)n_visit, k0.p, v0.p,
                       cnt.p);
    size_t tmp_bytes = 0;
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n_visit,
                                                         0, 64, ctx->stream));
    MP2P_TRY_HIP(ctx, tmp.take(ctx, 5, tmp_bytes));
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n_visit, 0,
                                                         64, ctx->stream));
    hipLaunchKernelGGL(ir_keep_kernel, dim3(1), dim3(1), 0, ctx->stream, cnt.p, prm->inliersRatio, cnt.p + 1);

    // 3. unique-global filter in sorted order + ordered compaction over the first nKeep entries
    const bool use_claims = !prm->allowMatchAlreadyMatchedGlobalPoints;
    if (use_claims)
    {
        ctx->epoch++;  // the search above made no claims; a fresh epoch for this list
        hipLaunchKernelGGL(ir_claim_kernel, dim3(nb), dim3(256), 0, ctx->stream, v1.p, cnt.p + 1, ctx->nn_spos.p,
                           cloud->pos.p, map->pts.p, ms ? ms->global_taken.p : nullptr, map->claims.p,
                           (~(unsigned long long)ctx->epoch) << 32);
    }
    // the bounding-box test uses +epsilon only (:63-66): no threshold in this matcher
    rc = launch_compact_slots(ctx, map, cloud, v1.p, n_visit, cnt.p + 1, 1, use_claims, /*always_mark=*/true, 0ull,
                              (float)prm->bounding_box_intersection_check_epsilon,
                              (unsigned long long)cloud->n /* :53: the whole layer's size */, ms, out);
    if (rc) return rc;
    uint32_t h_cnt[2] = {0, 0};
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h_cnt, cnt.p, sizeof(h_cnt), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, stream_wait(ctx));  // also: the sorted list is a temporary
    // ASSERT_(nTotal > 0)  (:117) -- only reached when the bounding boxes overlap (:63-66), which the
    // device decided; a layer whose every point is already paired is the caller's error there too
    MP2P_REQUIRE(ctx, h_cnt[0] > 0, "Matcher_Points_InlierRatio: no local point has a candidate (nTotal == 0)");
    return MP2P_HIP_OK;
}

}  // namespace mp2p
