// index_build.hip -- K2: build the NN index of a global point layer.
//
// Replaces nn_prepare_for_3d_queries() (Matcher_Points_DistanceThreshold.cpp:92), i.e. MRPT's
// nanoflann KD-tree build, with a structure shaped for wave64 / HBM:
//   * points sorted by a 60-bit Morton code of their FINE voxel coordinates, stored as
//     float4 {x,y,z,bits(original index)} so that one 16-byte coalesced load per lane
//     brings a whole candidate;
//   * one open-addressing hash table holding the occupied voxels of EVERY level
//     (voxel edge = fine edge << (shift0 + level)); in Morton order a voxel of any level
//     is one contiguous range [start,end) of the sorted array.
// The radix sort itself is the rocPRIM/hipCUB device primitive (SURVEY.md section 7 allows it
// for the amortised build; it is not on the per-iteration path).
#include <hipcub/hipcub.hpp>

#include <chrono>

#include "device_utils.hpp"

namespace mp2p
{
// ---- bbox reduction ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ y,
                                                   const float* __restrict__ z, uint32_t n,
                                                   uint32_t* __restrict__ out /*6 ordered*/)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const float a = x[i], b = y[i], c = z[i];
        mn[0] = fminf(mn[0], a), mn[1] = fminf(mn[1], b), mn[2] = fminf(mn[2], c);
        mx[0] = fmaxf(mx[0], a), mx[1] = fmaxf(mx[1], b), mx[2] = fmaxf(mx[2], c);
    }
    __shared__ float s[4][6];
    const int        lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = 0; d < 3; d++)
    {
        mn[d] = wave_min(mn[d]);
        mx[d] = wave_max(mx[d]);
    }
    if (lane == 0)
        for (int d = 0; d < 3; d++) s[w][d] = mn[d], s[w][3 + d] = mx[d];
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const int d = threadIdx.x;
        float     v = s[0][d];
        for (int k = 1; k < 4; k++) v = d < 3 ? fminf(v, s[k][d]) : fmaxf(v, s[k][d]);
        if (d < 3)
            atomicMin(&out[d], f2ord(v));
        else
            atomicMax(&out[d], f2ord(v));
    }
}

__global__ void bbox_init_kernel(uint32_t* out)
{
    if (threadIdx.x < 3) out[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) out[threadIdx.x] = 0u;
}

// ---- Morton keys of the fine voxel ----------------------------------------------------------
__global__ __launch_bounds__(256) void morton_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z, uint32_t n,
                                                     float ox, float oy, float oz, float inv_hf,
                                                     unsigned long long* __restrict__ keys,
                                                     uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = morton60(cell_fine(x[i], ox, inv_hf), cell_fine(y[i], oy, inv_hf),
                       cell_fine(z[i], oz, inv_hf));
    idx[i]  = i;
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z,
                                                     const uint32_t* __restrict__ idx, uint32_t n,
                                                     float4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx[i];
    out[i]           = make_float4(x[j], y[j], z[j], __uint_as_float(j));
}

// g(i) = index of the highest 3-bit group in which sorted keys i-1 and i differ (-1: equal).
// Element i starts a new voxel at every level l <= g(i).
__device__ __forceinline__ int diff_group(unsigned long long a, unsigned long long b)
{
    const unsigned long long x = a ^ b;
    if (x == 0) return -1;
    return (63 - __clzll((long long)x)) / 3;
}

// hist[l] = #{ i>=1 : g(i) == l }, l in 0..19
__global__ __launch_bounds__(256) void level_hist_kernel(
    const unsigned long long* __restrict__ keys, uint32_t n, unsigned long long* __restrict__ hist)
{
    __shared__ uint32_t h[20];
    if (threadIdx.x < 20) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1; i < n;
         i += gridDim.x * blockDim.x)
    {
        const int g = diff_group(keys[i - 1], keys[i]);
        if (g >= 0) atomicAdd(&h[g], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 20 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__device__ __forceinline__ void hash_insert(Cell* table, uint64_t mask, unsigned long long key,
                                            uint32_t start)
{
    uint64_t slot = hash_key(key) & mask;
    for (;;)
    {
        const unsigned long long prev = atomicCAS(&table[slot].key, CELL_EMPTY, key);
        if (prev == CELL_EMPTY || prev == key)
        {
            table[slot].start = start;
            return;
        }
        slot = (slot + 1) & mask;
    }
}

__device__ __forceinline__ void hash_set_end(Cell* table, uint64_t mask, unsigned long long key,
                                             uint32_t end)
{
    uint64_t slot = hash_key(key) & mask;
    for (;;)
    {
        const unsigned long long k = table[slot].key;
        if (k == key)
        {
            table[slot].end = end;
            return;
        }
        if (k == CELL_EMPTY) return;  // cannot happen
        slot = (slot + 1) & mask;
    }
}

// pass 0: voxel heads -> insert {key,start}; pass 1: voxel tails -> set end
struct OccBuild
{
    unsigned long long* words;
    uint32_t            off[16], bx[16], by[16];
    uint2*              dir;  // dense voxel directory (GridView::dir)
    unsigned long long  dir_off[16];
};

template <int PASS>
__global__ __launch_bounds__(256) void cells_kernel(const unsigned long long* __restrict__ keys,
                                                    const float4* __restrict__ pts, uint32_t n,
                                                    float ox, float oy, float oz, float inv_hf,
                                                    uint32_t shift0, uint32_t n_levels,
                                                    Cell* __restrict__ table, uint64_t mask,
                                                    const OccBuild occ)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (PASS == 0)
    {
        if (i >= n) return;
        const int g = (i == 0) ? 100 : diff_group(keys[i - 1], keys[i]);
        if (g < (int)shift0) return;  // not a head at any stored level
        const float4   p  = pts[i];
        const uint32_t fx = cell_fine(p.x, ox, inv_hf), fy = cell_fine(p.y, oy, inv_hf),
                       fz = cell_fine(p.z, oz, inv_hf);
        for (uint32_t l = 0; l < n_levels && (int)(shift0 + l) <= g; l++)
        {
            const uint32_t s  = shift0 + l;
            const uint32_t cx = fx >> s, cy = fy >> s, cz = fz >> s;
            hash_insert(table, mask, cell_key(l, cx, cy, cz), i);
            if (occ.words && occ.off[l] != OCC_NONE)  // one bit per occupied voxel (this is its head)
                atomicOr(&occ.words[(size_t)occ.off[l] +
                                    ((size_t)(cz >> 2) * occ.by[l] + (cy >> 2)) * occ.bx[l] + (cx >> 2)],
                         1ull << (((cz & 3u) << 4) | ((cy & 3u) << 2) | (cx & 3u)));
            if (occ.dir_off[l] != DIR_NONE)
                occ.dir[occ.dir_off[l] + ((size_t)cz * (occ.by[l] * 4u) + cy) * (occ.bx[l] * 4u) + cx].x = i;
        }
    }
    else
    {
        // boundary between element i and i+1 (i+1 == n: end of everything)
        if (i >= n) return;
        const int g = (i + 1 == n) ? 100 : diff_group(keys[i], keys[i + 1]);
        if (g < (int)shift0) return;
        const float4   p  = pts[i];
        const uint32_t fx = cell_fine(p.x, ox, inv_hf), fy = cell_fine(p.y, oy, inv_hf),
                       fz = cell_fine(p.z, oz, inv_hf);
        for (uint32_t l = 0; l < n_levels && (int)(shift0 + l) <= g; l++)
        {
            const uint32_t s = shift0 + l;
            hash_set_end(table, mask, cell_key(l, fx >> s, fy >> s, fz >> s), i + 1);
            if (occ.dir_off[l] != DIR_NONE)
                occ.dir[occ.dir_off[l] + ((size_t)(fz >> s) * (occ.by[l] * 4u) + (fy >> s)) * (occ.bx[l] * 4u) + (fx >> s)].y = i + 1;
        }
    }
}

static inline uint32_t nblk(size_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

static int device_bbox(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z,
                       size_t n, float mn[3], float mx[3])
{
    DevBuf<uint32_t> d_bb;
    MP2P_TRY_HIP(ctx, d_bb.alloc(6));
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, ctx->stream, d_bb.p);
    const uint32_t blocks = std::min<uint32_t>(nblk(n, 256), 2048);
    hipLaunchKernelGGL(bbox_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_x, d_y, d_z,
                       (uint32_t)n, d_bb.p);
    uint32_t h[6];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(h, d_bb.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int d = 0; d < 3; d++) mn[d] = ord2f(h[d]), mx[d] = ord2f(h[3 + d]);
    d_bb.release();
    return MP2P_HIP_OK;
}

// sorts the points by fine-voxel Morton code; returns sorted keys (optional) and float4 points
static int morton_sort(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z,
                       size_t n, float ox, float oy, float oz, float inv_hf,
                       DevBuf<unsigned long long>* keys_out, float4* d_pts_out)
{
    MP2P_REQUIRE_INT_COUNT(ctx, n);
    DevBuf<unsigned long long> k0, k1;
    DevBuf<uint32_t>           i0, i1;
    MP2P_TRY_HIP(ctx, k0.alloc(n));
    MP2P_TRY_HIP(ctx, k1.alloc(n));
    MP2P_TRY_HIP(ctx, i0.alloc(n));
    MP2P_TRY_HIP(ctx, i1.alloc(n));
    hipLaunchKernelGGL(morton_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_x, d_y,
                       d_z, (uint32_t)n, ox, oy, oz, inv_hf, k0.p, i0.p);
    size_t tmp_bytes = 0;
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, i0.p,
                                                         i1.p, (int)n, 0, 60, ctx->stream));
    DevBuf<unsigned char> tmp;
    MP2P_TRY_HIP(ctx, tmp.alloc(tmp_bytes ? tmp_bytes : 1));
    MP2P_TRY_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, i0.p, i1.p,
                                                         (int)n, 0, 60, ctx->stream));
    hipLaunchKernelGGL(gather_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, d_x, d_y,
                       d_z, i1.p, (uint32_t)n, d_pts_out);
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (keys_out)
    {
        *keys_out = std::move(k1);
    }
    k0.release(), k1.release(), i0.release(), i1.release(), tmp.release();
    return MP2P_HIP_OK;
}

int build_map(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z, size_t n,
              const mp2p_hip_map_params* prm, mp2p_hip_map* map)
{
    const auto t0 = std::chrono::steady_clock::now();
    map->ctx      = ctx;
    map->n        = n;
    GridView& g   = map->view;
    memset(&g, 0, sizeof(g));
    memset(&map->info, 0, sizeof(map->info));
    map->info.n_points = n;
    if (n == 0) return MP2P_HIP_OK;  // isEmpty() map: matchers early-out
    MP2P_REQUIRE(ctx, n < 0xFFFFFFF0ull, "global layer too large for 32-bit indices");

    float mn[3], mx[3];
    int   rc = device_bbox(ctx, d_x, d_y, d_z, n, mn, mx);
    if (rc) return rc;
    for (int d = 0; d < 3; d++)
        MP2P_REQUIRE(ctx, std::isfinite(mn[d]) && std::isfinite(mx[d]),
                     "global layer has non-finite coordinates");

    const float ext    = std::max(std::max(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    float       maxabs = 0;
    for (int d = 0; d < 3; d++) maxabs = std::max(maxabs, std::max(std::fabs(mn[d]), std::fabs(mx[d])));

    const bool  user_cell = prm && prm->cell_size > 0;
    const float target    = (prm && prm->target_per_cell > 0) ? std::max(prm->target_per_cell, 1.25f) : 6.0f;
    // fine voxel: the user's cell, or the finest 20-bit subdivision of the bounding cube
    float hf = user_cell ? prm->cell_size : std::max(ext * (1.0f / 1048000.0f), 1e-6f);
    if (user_cell && ext / hf > 1048000.0f) hf = ext / 1048000.0f;  // keep 20-bit coordinates
    const float inv_hf = 1.0f / hf;

    MP2P_TRY_HIP(ctx, map->pts.alloc(n));
    DevBuf<unsigned long long> keys;
    rc = morton_sort(ctx, d_x, d_y, d_z, n, mn[0], mn[1], mn[2], inv_hf, &keys, map->pts.p);
    if (rc) return rc;

    // occupied-voxel counts per fine level: cells(l) = 1 + #{i : g(i) >= l}
    DevBuf<unsigned long long> d_hist;
    MP2P_TRY_HIP(ctx, d_hist.alloc(20));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(d_hist.p, 0, 20 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(level_hist_kernel, dim3(std::min<uint32_t>(nblk(n, 256), 4096)), dim3(256),
                       0, ctx->stream, keys.p, (uint32_t)n, d_hist.p);
    unsigned long long hist[20];
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(hist, d_hist.p, sizeof(hist), hipMemcpyDeviceToHost, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    d_hist.release();
    unsigned long long cells[21];
    {
        unsigned long long acc = 1;
        for (int l = 20; l >= 0; l--)
        {
            cells[l] = acc;
            if (l > 0) acc += hist[l - 1];
        }
        // cells[l] = 1 + sum_{k>=l} hist[k]
    }
    uint32_t shift0 = 0;
    if (!user_cell)
    {
        // finest level whose occupied voxels hold >= target points on average
        while (shift0 < 19 && (double)n / (double)cells[shift0] < target) shift0++;
    }
    // levels up to the one where the whole layer is a handful of voxels
    uint32_t top = shift0;
    while (top < 20 && cells[top] > 8) top++;
    uint32_t n_levels = top - shift0 + 1;
    const uint32_t max_levels = (prm && prm->max_levels) ? prm->max_levels : 16;
    if (n_levels > max_levels) n_levels = max_levels;
    if (n_levels > 15) n_levels = 15;  // 4 key bits
    unsigned long long total_cells = 0;
    for (uint32_t l = 0; l < n_levels; l++) total_cells += cells[shift0 + l];

    uint64_t cap = 1024;
    while (cap < 2 * total_cells) cap <<= 1;
    MP2P_TRY_HIP(ctx, map->table.alloc(cap));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(map->table.p, 0xFF, cap * sizeof(Cell), ctx->stream));
    // occupancy bitmaps: level l has ((max fine cell >> s) + 1) voxels per axis, in 4x4x4 bricks;
    // a level whose bitmap would not fit the budget simply has none (the probe decides alone)
    OccBuild ob;
    memset(&ob, 0, sizeof(ob));
    {
        const unsigned long long budget = (prm && (prm->no_occupancy_bitmap & 1u)) ? 0ull : (1ull << 27);  // words = 1 GB
        unsigned long long       total  = 0;
        uint32_t nf[3];
        for (int d = 0; d < 3; d++)
            nf[d] = (uint32_t)std::min<double>(1048575.0, std::floor((double)(mx[d] - mn[d]) * (double)inv_hf) + 2.0) + 1u;
        for (int l = (int)n_levels - 1; l >= 0; l--)  // coarse levels first: they are tiny
        {
            const uint32_t s = shift0 + (uint32_t)l;
            const unsigned long long bx = (((nf[0] - 1) >> s) + 4) / 4, by = (((nf[1] - 1) >> s) + 4) / 4,
                                     bz = (((nf[2] - 1) >> s) + 4) / 4;
            const unsigned long long w = bx * by * bz;
            g.occ_bx[l] = (uint32_t)bx, g.occ_by[l] = (uint32_t)by, g.occ_bz[l] = (uint32_t)bz;
            if (total + w > budget || total + w >= OCC_NONE) { g.occ_off[l] = OCC_NONE; continue; }
            g.occ_off[l] = (uint32_t)total;
            total += w;
        }
        for (uint32_t l = n_levels; l < 16; l++) g.occ_off[l] = OCC_NONE, g.occ_bx[l] = g.occ_by[l] = g.occ_bz[l] = 0;
        if (total)
        {
            MP2P_TRY_HIP(ctx, map->occ.alloc(total));
            MP2P_TRY_HIP(ctx, hipMemsetAsync(map->occ.p, 0, total * sizeof(unsigned long long), ctx->stream));
        }
        ob.words = map->occ.p;
        for (int l = 0; l < 16; l++) ob.off[l] = g.occ_off[l], ob.bx[l] = g.occ_bx[l], ob.by[l] = g.occ_by[l];
        // dense voxel directories, finest level first (it serves nearly every lookup of a warm ICP
        // iteration; each coarser level is 1/8 of the one below)
        const uint32_t     variant  = prm ? prm->no_occupancy_bitmap : 0u;
        unsigned long long dir_left = (variant & 3u) ? 0ull : (unsigned long long)ctx->tune.dir_budget_mb * (1ull << 20) / sizeof(uint2);
        {
            // never more than a quarter of what is free now: the directory is an accelerator, the hash table serves
            // every level it does not cover (a host that keeps many maps resident must not run the device dry on it)
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) dir_left = std::min<unsigned long long>(dir_left, free_b / 4 / sizeof(uint2));
            else (void)hipGetLastError();
        }
        unsigned long long dir_total = 0;
        for (int l = 0; l < 16; l++)
        {
            g.dir_off[l] = DIR_NONE;
            if (l >= (int)n_levels || (l == 0 && (variant & 4u))) continue;
            const unsigned long long v = 64ull * g.occ_bx[l] * g.occ_by[l] * g.occ_bz[l];
            if (v > dir_left) continue;
            g.dir_off[l] = dir_total, dir_total += v, dir_left -= v;
        }
        if (dir_total)
        {
            const hipError_t ea = map->dir.alloc(dir_total);
            if (ea == hipErrorOutOfMemory)
            {
                // degrade instead of failing the upload: no directory, every lookup probes the hash table
                (void)hipGetLastError();
                for (int l = 0; l < 16; l++) g.dir_off[l] = DIR_NONE;
                dir_total = 0;
            }
            else
            {
                MP2P_TRY_HIP(ctx, ea);
                MP2P_TRY_HIP(ctx, hipMemsetAsync(map->dir.p, 0, dir_total * sizeof(uint2), ctx->stream));
            }
        }
        ob.dir = map->dir.p;
        for (int l = 0; l < 16; l++) ob.dir_off[l] = g.dir_off[l];
    }
    hipLaunchKernelGGL(cells_kernel<0>, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, keys.p,
                       map->pts.p, (uint32_t)n, mn[0], mn[1], mn[2], inv_hf, shift0, n_levels,
                       map->table.p, cap - 1, ob);
    hipLaunchKernelGGL(cells_kernel<1>, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, keys.p,
                       map->pts.p, (uint32_t)n, mn[0], mn[1], mn[2], inv_hf, shift0, n_levels,
                       map->table.p, cap - 1, ob);
    MP2P_TRY_HIP(ctx, map->claims.alloc(n));
    MP2P_TRY_HIP(ctx, hipMemsetAsync(map->claims.p, 0xFF, n * sizeof(unsigned long long), ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    keys.release();

    g.pts = map->pts.p, g.n = (uint32_t)n;
    g.table = map->table.p, g.mask = cap - 1;
    g.occ   = map->occ.p;
    g.dir   = map->dir.p;
    g.ox = mn[0], g.oy = mn[1], g.oz = mn[2];
    g.hf = hf, g.inv_hf = inv_hf;
    g.shift0 = shift0, g.n_levels = n_levels;
    for (int d = 0; d < 3; d++) g.bbmin[d] = mn[d], g.bbmax[d] = mx[d];
    // 8 ulp of the largest coordinate magnitude (see nn_query.hip for how it is used)
    g.slack = std::max(std::max(maxabs, ext) * (1.0f / 1048576.0f), 1e-30f);

    mp2p_hip_map_info& info = map->info;
    for (int d = 0; d < 3; d++) info.bbox_min[d] = mn[d], info.bbox_max[d] = mx[d];
    info.cell_size      = hf * (float)(1u << shift0);
    info.n_levels       = n_levels;
    info.n_cells_total  = total_cells;
    info.n_cells_level0 = cells[shift0];
    info.hash_capacity  = cap;
    info.device_bytes   = map->pts.bytes() + map->table.bytes() + map->claims.bytes() + map->occ.bytes() + map->dir.bytes();
    info.build_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MP2P_HIP_OK;
}

// pos[original index] = place in the Morton-sorted copy
__global__ __launch_bounds__(256) void inverse_order_kernel(const float4* __restrict__ sorted, uint32_t n,
                                                            uint32_t* __restrict__ pos)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pos[__float_as_uint(sorted[i].w)] = i;
}

int build_cloud(mp2p_hip_ctx* ctx, const float* d_x, const float* d_y, const float* d_z,
                size_t n, mp2p_hip_cloud* cloud)
{
    cloud->ctx = ctx;
    cloud->n   = n;
    if (n == 0) return MP2P_HIP_OK;
    MP2P_REQUIRE(ctx, n < 0xFFFFFFF0ull, "local layer too large for 32-bit indices");
    float mn[3], mx[3];
    int   rc = device_bbox(ctx, d_x, d_y, d_z, n, mn, mx);
    if (rc) return rc;
    const float ext = std::max(std::max(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    const float hf  = std::max(ext * (1.0f / 1048000.0f), 1e-9f);
    {
        double r2 = 0;
        for (int d = 0; d < 3; d++) r2 += (double)std::max(std::fabs(mn[d]), std::fabs(mx[d])) * std::max(std::fabs(mn[d]), std::fabs(mx[d]));
        cloud->radius = (float)std::sqrt(r2) * 1.000001f;  // (of the bounding box's farthest corner: >= every |p|)
    }
    MP2P_TRY_HIP(ctx, cloud->sorted.alloc(n));
    rc = morton_sort(ctx, d_x, d_y, d_z, n, mn[0], mn[1], mn[2], 1.0f / hf, nullptr,
                     cloud->sorted.p);
    if (rc) return rc;
    MP2P_TRY_HIP(ctx, cloud->pos.alloc(n));
    hipLaunchKernelGGL(inverse_order_kernel, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream,
                       cloud->sorted.p, (uint32_t)n, cloud->pos.p);
    MP2P_TRY_HIP(ctx, cloud->x.alloc(n));
    MP2P_TRY_HIP(ctx, cloud->y.alloc(n));
    MP2P_TRY_HIP(ctx, cloud->z.alloc(n));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(cloud->x.p, d_x, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(cloud->y.p, d_y, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipMemcpyAsync(cloud->z.p, d_z, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    MP2P_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MP2P_HIP_OK;
}

}  // namespace mp2p
