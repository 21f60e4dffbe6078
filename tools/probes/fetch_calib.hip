// What does FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md "HBM") mean for the access patterns of the search kernels?
// Known byte counts, 16-byte loads from a 2 GB float4 array (far beyond the 8 x 4 MB of L2):
//   stream    : every float4 once, coalesced                      -> requested = touched = 16 N
//   gather_1  : one float4 per 128-byte line at a random slot (sorted positions: the warm-start re-measure of
//               nn_lane_kernel, one 16-byte load per query at its previous neighbour)
//   gather_runs: runs of RUN consecutive float4 starting at sorted random places (a voxel's range staged by the tile
//               kernel: RUN = 6..24), 4 loads in flight per lane
// For each kernel the host prints requested bytes and the bytes of the distinct 64-B and 128-B blocks it touches; the
// counter (rocprofv3 --pmc FETCH_SIZE, tools/gpu_calib.sh) is set against them in profiles/r04_fetch_size_calibration.txt.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/fetch_calib.hip -o tools/probes/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void stream_read(const float4* __restrict__ a, size_t n, float* out)
{
    float        acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += a[i].x;
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void gather_1(const float4* __restrict__ a, const uint32_t* __restrict__ pos, size_t m, float* out)
{
    float        acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) acc += a[pos[i]].x;
    if (acc == 12345.678f) out[0] = acc;
}
// one wave per group of 64 runs; lane l of the wave walks the points of the runs flattened (as the staging loop does)
__global__ void gather_runs(const float4* __restrict__ a, const uint32_t* __restrict__ start, uint32_t run, size_t n_runs, float* out)
{
    float        acc  = 0.f;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    for (size_t g = wave; g * 64 < n_runs; g += n_waves)
    {
        const size_t   r0    = g * 64;
        const uint32_t total = (uint32_t)std::min<size_t>(64, n_runs - r0) * run;
        for (uint32_t t0 = 0; t0 < total; t0 += 256)
        {
            float4 c[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t t = t0 + 64u * k + (uint32_t)lane;
                c[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < total) c[k] = a[start[r0 + t / run] + t % run];
            }
            acc += c[0].x + c[1].x + c[2].x + c[3].x;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

static void blocks(const std::vector<uint64_t>& idx, double& b64, double& b128)
{
    std::vector<uint64_t> s(idx.size());
    for (size_t i = 0; i < idx.size(); i++) s[i] = idx[i] / 4;  // 64-byte block of a 16-byte element
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    b64 = 64.0 * (double)s.size();
    for (auto& v : s) v /= 2;
    s.erase(std::unique(s.begin(), s.end()), s.end());
    b128 = 128.0 * (double)s.size();
}

int main()
{
    const size_t N = (size_t)128 << 20;  // 2 GB of float4
    float4*      a;
    float*       out;
    CK(hipMalloc(&a, N * sizeof(float4)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, N * sizeof(float4)));
    std::mt19937_64 rng(7);
    // ---- stream
    hipLaunchKernelGGL(stream_read, dim3(8192), dim3(256), 0, 0, a, N, out);
    CK(hipDeviceSynchronize());
    printf("stream_read     requested %.1f MB  distinct64 %.1f MB  distinct128 %.1f MB\n", N * 16 / 1e6, N * 16 / 1e6, N * 16 / 1e6);
    // ---- one element per 128-byte line
    {
        const size_t          M = N / 8;
        std::vector<uint32_t> pos(M);
        std::vector<uint64_t> idx(M);
        for (size_t i = 0; i < M; i++) pos[i] = (uint32_t)(i * 8 + (rng() & 7)), idx[i] = pos[i];
        uint32_t* d;
        CK(hipMalloc(&d, M * 4));
        CK(hipMemcpy(d, pos.data(), M * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(gather_1, dim3(8192), dim3(256), 0, 0, a, d, M, out);
        CK(hipDeviceSynchronize());
        double b64, b128;
        blocks(idx, b64, b128);
        printf("gather_1        requested %.1f MB  distinct64 %.1f MB  distinct128 %.1f MB  (+ %.1f MB of positions, streamed)\n", M * 16 / 1e6, b64 / 1e6, b128 / 1e6, M * 4 / 1e6);
        CK(hipFree(d));
    }
    // ---- runs
    for (uint32_t run : {6u, 24u})
    {
        const size_t          R = N / (4 * run);  // a quarter of the array is fetched
        std::vector<uint32_t> st(R);
        std::vector<uint64_t> idx;
        idx.reserve(R * run);
        for (size_t i = 0; i < R; i++)
        {
            st[i] = (uint32_t)(i * 4 * run + rng() % (3 * run));  // sorted, non-overlapping, unaligned
            for (uint32_t k = 0; k < run; k++) idx.push_back(st[i] + k);
        }
        uint32_t* d;
        CK(hipMalloc(&d, R * 4));
        CK(hipMemcpy(d, st.data(), R * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(gather_runs, dim3(16384), dim3(64), 0, 0, a, d, run, R, out);
        CK(hipDeviceSynchronize());
        double b64, b128;
        blocks(idx, b64, b128);
        printf("gather_runs<%2u> requested %.1f MB  distinct64 %.1f MB  distinct128 %.1f MB  (+ %.1f MB of run starts)\n", run, R * run * 16 / 1e6, b64 / 1e6, b128 / 1e6,
               R * 4 / 1e6);
        CK(hipFree(d));
    }
    return 0;
}
