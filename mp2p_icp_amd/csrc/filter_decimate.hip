// filter_decimate.hip -- FilterDecimateVoxels (mp2p_icp_filters/src/FilterDecimateVoxels.cpp:107-381),
// the step right before the matchers in every demo pipeline ("decimated" layer).
//
// Reference semantics reproduced:
//   voxel of a point  = (int32)(x / resolution) per axis, truncation toward zero
//                       (PointCloudToVoxelGrid.h:110 coord2idx -- the cells touching 0 are double)
//   FirstPoint        = the lowest point index of the voxel (PointCloudToVoxelGridSingle.cpp:58-84)
//   VoxelAverage      = fp32 mean: coordinates summed in ascending point index, then * (1.0f / n)
//                       (FilterDecimateVoxels.cpp:290-300)
//   ClosestToAverage  = first point (ascending index) with the smallest
//                       ((x-mx)^2 + (y-my)^2) + (z-mz)^2 in fp32 (:302-322)
//   flatten_to        = only the first voxel visited of each (cx, cy) column emits, with z replaced
//                       (:232-246, :346-360)
//   RandomPoint       = not offered: it draws from mrpt::random (un-vendored), nothing to pin.
// Output ORDER: the reference visits a tsl::robin_map (un-vendored; its order depends on the table's
// growth history) or a std::map ordered by (cx, cy, cz).  This implementation always emits in the
// std::map order -- the same SET of points as the reference in either mode, the same sequence as
// its std::map mode.
//
// Plan: key = offset-binary (cx, cy, cz) in 3 x 21 bits -> stable radix sort of (key, index) ->
// a voxel is a run of equal keys whose indices ascend -> one thread per voxel walks its run in the
// reference's order -> flags + exclusive scan -> ordered write.
#include "device_utils.hpp"

namespace mp2p
{
constexpr int DV_OFFSET = 1 << 20;  // voxel coordinates in [-2^20, 2^20)

__device__ __forceinline__ int coord2idx(float v, float res)
{
    return (int)__fdiv_rn(v, res);  // static_cast<int32_t>(xyz / resolution_)
}

__global__ __launch_bounds__(256) void dv_keys_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ z, uint32_t n, float res,
                                                      unsigned long long* __restrict__ keys,
                                                      uint32_t* __restrict__ idx, uint32_t* __restrict__ bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float fx = __fdiv_rn(x[i], res), fy = __fdiv_rn(y[i], res), fz = __fdiv_rn(z[i], res);
    // out of the 21-bit range (or not finite): reported, never silently wrapped
    const float lim = (float)(DV_OFFSET - 1);
    if (!(fabsf(fx) < lim && fabsf(fy) < lim && fabsf(fz) < lim)) atomicAdd(bad, 1u);
    const long long cx = (long long)(int)fx + DV_OFFSET, cy = (long long)(int)fy + DV_OFFSET,
                    cz = (long long)(int)fz + DV_OFFSET;
    keys[i] = ((unsigned long long)(cx & 0x1FFFFF) << 42) | ((unsigned long long)(cy & 0x1FFFFF) << 21) |
              (unsigned long long)(cz & 0x1FFFFF);
    idx[i] = i;
}

// one thread per sorted position that starts a voxel: walk the run
__global__ __launch_bounds__(256) void dv_voxels_kernel(
    const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ idx, uint32_t n,
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int method,
    int flatten, unsigned char* __restrict__ flag, float* __restrict__ vx, float* __restrict__ vy,
    float* __restrict__ vz, uint32_t* __restrict__ vsrc)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k    = keys[i];
    const bool               head = (i == 0) || keys[i - 1] != k;
    unsigned char            f    = 0;
    if (head)
    {
        // flatten_to: only the first voxel of the (cx, cy) column in (cx, cy, cz) order emits
        f = 1;
        if (flatten && i > 0 && (keys[i - 1] >> 21) == (k >> 21)) f = 0;
        if (f)
        {
            const uint32_t first = idx[i];
            float          ox = x[first], oy = y[first], oz = z[first];
            uint32_t       src = first;
            if (method != MP2P_HIP_DECIMATE_FIRST_POINT)
            {
                uint32_t e = i + 1;
                while (e < n && keys[e] == k) e++;
                float mx = 0.f, my = 0.f, mz = 0.f;
                for (uint32_t j = i; j < e; j++)
                {
                    const uint32_t p = idx[j];
                    mx = fadd(mx, x[p]), my = fadd(my, y[p]), mz = fadd(mz, z[p]);
                }
                const float inv_n = __fdiv_rn(1.0f, (float)(e - i));
                mx = fmul(mx, inv_n), my = fmul(my, inv_n), mz = fmul(mz, inv_n);
                if (method == MP2P_HIP_DECIMATE_VOXEL_AVERAGE)
                    ox = mx, oy = my, oz = mz, src = NONE_U32;
                else
                {
                    float best = INFINITY;
                    bool  have = false;
                    for (uint32_t j = i; j < e; j++)
                    {
                        const uint32_t p  = idx[j];
                        const float    dx = fsub(x[p], mx), dy = fsub(y[p], my), dz = fsub(z[p], mz);
                        const float    s  = fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz));
                        if (!have || s < best) best = s, src = p, have = true;
                    }
                    ox = x[src], oy = y[src], oz = z[src];
                }
            }
            vx[i] = ox, vy[i] = oy, vz[i] = oz, vsrc[i] = src;
        }
    }
    flag[i] = f;
}

constexpr int DV_THREADS = 256, DV_ITEMS = 4, DV_TILE = DV_THREADS * DV_ITEMS;

__global__ __launch_bounds__(DV_THREADS) void dv_count_kernel(const unsigned char* __restrict__ flag, uint32_t n,
                                                              uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t s_w[DV_THREADS / 64];
    const uint32_t      base = blockIdx.x * DV_TILE + threadIdx.x * DV_ITEMS;
    uint32_t            c    = 0;
#pragma unroll
    for (int k = 0; k < DV_ITEMS; k++)
        if (base + k < n && flag[base + k]) c++;
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t t = 0;
        for (int w = 0; w < DV_THREADS / 64; w++) t += s_w[w];
        block_counts[blockIdx.x] = t;
    }
}

// single block: exclusive scan in place, total in block_counts[n_blocks]
__global__ __launch_bounds__(1024) void dv_scan_kernel(uint32_t* block_counts, uint32_t n_blocks)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_run;
    const int           lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024)
    {
        const uint32_t i    = b0 + threadIdx.x;
        const uint32_t v    = i < n_blocks ? block_counts[i] : 0;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; k++) woff += s_w[k];
        const uint32_t run = s_run;
        if (i < n_blocks) block_counts[i] = run + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = run + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_counts[n_blocks] = s_run;
}

__global__ __launch_bounds__(DV_THREADS) void dv_write_kernel(
    const unsigned char* __restrict__ flag, uint32_t n, const uint32_t* __restrict__ block_counts,
    const float* __restrict__ vx, const float* __restrict__ vy, const float* __restrict__ vz,
    const uint32_t* __restrict__ vsrc, int flatten, float flatten_to, float* __restrict__ ox,
    float* __restrict__ oy, float* __restrict__ oz, uint32_t* __restrict__ osrc)
{
    __shared__ uint32_t s_w[DV_THREADS / 64];
    const uint32_t      base = blockIdx.x * DV_TILE + threadIdx.x * DV_ITEMS;
    bool                f[DV_ITEMS];
    uint32_t            c = 0;
#pragma unroll
    for (int k = 0; k < DV_ITEMS; k++)
    {
        f[k] = (base + k < n) && flag[base + k];
        c += f[k] ? 1u : 0u;
    }
    const int      lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; k++) woff += s_w[k];
    uint32_t dst = block_counts[blockIdx.x] + woff + incl - c;
#pragma unroll
    for (int k = 0; k < DV_ITEMS; k++)
    {
        if (!f[k]) continue;
        const uint32_t i = base + k;
        ox[dst] = vx[i], oy[dst] = vy[i], oz[dst] = flatten ? flatten_to : vz[i];
        if (osrc) osrc[dst] = vsrc[i];
        dst++;
    }
}

static inline uint32_t dv_nblk(size_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

// device arrays in, device arrays out (capacity n each); *n_out = number of output points
int filter_decimate_device(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z, size_t n,
                           const mp2p_hip_decimate_params* prm, float* d_ox, float* d_oy, float* d_oz,
                           uint32_t* d_osrc, size_t* n_out)
{
    *n_out = 0;
    if (n == 0) return MP2P_HIP_OK;
    MP2P_REQUIRE(ctx, n < 0xFFFFFFF0ull, "layer too large for 32-bit indices");
    MP2P_REQUIRE_INT_COUNT(ctx, n);
    Scratch<unsigned long long> k0, k1;
    Scratch<uint32_t>           i0, i1, bad, vsrc, blocks;
    Scratch<unsigned char>      flag, tmp;
    Scratch<float>              vx, vy, vz;
    MP2P_TRY_HIP(ctx, k0.take(ctx, 0, n));
    MP2P_TRY_HIP(ctx, k1.take(ctx, 1, n));
    MP2P_TRY_HIP(ctx, i0.take(ctx, 2, n));
    MP2P_TRY_HIP(ctx, i1.take(ctx, 3, n));
    MP2P_TRY_HIP(ctx, bad.take(ctx, 4, 1));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(bad.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(dv_keys_kernel, dim3(dv_nblk(n, 256)), dim3(256), 0, ctx->stream, d_x, d_y, d_z,
                       (uint32_t)n, prm->voxel_filter_resolution, k0.p, i0.p, bad.p);
    size_t tmp_bytes = 0;
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, i0.p, i1.p, (int)n, 0,
                                                         63, ctx->stream));
    MP2P_TRY_HIP(ctx, tmp.take(ctx, 5, tmp_bytes));
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, i0.p, i1.p, (int)n, 0, 63,
                                                         ctx->stream));
    MP2P_TRY_HIP(ctx, flag.take(ctx, 6, n));
    MP2P_TRY_HIP(ctx, vx.take(ctx, 7, n));
    MP2P_TRY_HIP(ctx, vy.take(ctx, 8, n));
    MP2P_TRY_HIP(ctx, vz.take(ctx, 9, n));
    MP2P_TRY_HIP(ctx, vsrc.take(ctx, 10, n));
    hipLaunchKernelGGL(dv_voxels_kernel, dim3(dv_nblk(n, 256)), dim3(256), 0, ctx->stream, k1.p, i1.p,
                       (uint32_t)n, d_x, d_y, d_z, (int)prm->decimate_method, (int)prm->has_flatten_to, flag.p,
                       vx.p, vy.p, vz.p, vsrc.p);
    const uint32_t n_blocks = dv_nblk(n, DV_TILE);
    MP2P_TRY_HIP(ctx, blocks.take(ctx, 11, n_blocks + 1));
    hipLaunchKernelGGL(dv_count_kernel, dim3(n_blocks), dim3(DV_THREADS), 0, ctx->stream, flag.p, (uint32_t)n,
                       blocks.p);
    hipLaunchKernelGGL(dv_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, blocks.p, n_blocks);
    hipLaunchKernelGGL(dv_write_kernel, dim3(n_blocks), dim3(DV_THREADS), 0, ctx->stream, flag.p, (uint32_t)n,
                       blocks.p, vx.p, vy.p, vz.p, vsrc.p, (int)prm->has_flatten_to, prm->flatten_to, d_ox, d_oy,
                       d_oz, d_osrc);
    uint32_t h_total = 0, h_bad = 0;
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(&h_total, blocks.p + n_blocks, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                     ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(&h_bad, bad.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MP2P_TRY_HIP(ctx, hipGetLastError());
    MP2P_REQUIRE(ctx, h_bad == 0, "decimate: a coordinate / resolution is not finite or exceeds 2^20 voxels");
    *n_out = h_total;
    return MP2P_HIP_OK;
}

}  // namespace mp2p
