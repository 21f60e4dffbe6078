// device_utils.hpp -- device-side helpers (gfx950, wave64).
#pragma once
#include "common.hpp"

namespace mp2p
{
// ---- rounding-exact fp32 / fp64 arithmetic: never contracted into FMA ---------------------
// The reference evaluates d2 = ((dx*dx)+(dy*dy))+(dz*dz), the thresholds and the pose
// composition with separately rounded mul/add on x86-64; bit-exact
// correspondence indices need the same roundings here.
// NOTE: HIP's __fmul_rn/__fadd_rn are plain `a*b` / `a+b` (see __clang_hip_math.h) and hipcc
// defaults to -ffp-contract=fast, so they do NOT stop FMA fusion.  Contraction is disabled
// here per function (the IR ops carry no `contract` flag) and, belt and braces, the build
// passes -ffp-contract=off; tests/test_build.py greps the ISA of the NN kernels for fma.
__device__ __forceinline__ float fmul(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float fadd(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float fsub(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double dmul(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double dadd(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float px, float py, float pz)
{
    const float dx = fsub(qx, px), dy = fsub(qy, py), dz = fsub(qz, pz);
    return fadd(fadd(fmul(dx, dx), fmul(dy, dy)), fmul(dz, dz));
}

// CPose3D::composePoint in fp64, left to right, narrowed once (Matcher_Points_Base.cpp:216-217)
struct PoseRt
{
    double r[9];
    double t[3];
};
__device__ __forceinline__ void compose_point_f(const PoseRt& P, float lx, float ly, float lz,
                                                float& gx, float& gy, float& gz)
{
    const double X = lx, Y = ly, Z = lz;
    gx = (float)dadd(dadd(dadd(dmul(P.r[0], X), dmul(P.r[1], Y)), dmul(P.r[2], Z)), P.t[0]);
    gy = (float)dadd(dadd(dadd(dmul(P.r[3], X), dmul(P.r[4], Y)), dmul(P.r[5], Z)), P.t[1]);
    gz = (float)dadd(dadd(dadd(dmul(P.r[6], X), dmul(P.r[7], Y)), dmul(P.r[8], Z)), P.t[2]);
}

// ---- order-preserving float <-> uint encoding (for atomicMin/Max on floats) ---------------
__device__ __host__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __host__ __forceinline__ float ord2f(uint32_t u)
{
    uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float    f;
    memcpy(&f, &b, 4);
    return f;
}

// ---- wave64 collectives --------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---- voxel addressing: THE cell function, used identically by the index build and by the
//      queries (monotone non-decreasing in x, so an interval of coordinates maps onto an
//      interval of cells whatever the rounding) ------------------------------------------
__device__ __forceinline__ uint32_t cell_fine(float x, float o, float inv_hf)
{
    float f = fmul(fsub(x, o), inv_hf);
    f       = fminf(fmaxf(f, 0.0f), 1048575.0f);  // 20 bits; NaN -> 0
    return (uint32_t)f;
}

__device__ __host__ __forceinline__ unsigned long long spread20(uint32_t v)
{
    unsigned long long x = v & 0xFFFFFu;
    x                    = (x | x << 32) & 0x1f00000000ffffull;
    x                    = (x | x << 16) & 0x1f0000ff0000ffull;
    x                    = (x | x << 8) & 0x100f00f00f00f00full;
    x                    = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x                    = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __host__ __forceinline__ unsigned long long morton60(uint32_t cx, uint32_t cy,
                                                                uint32_t cz)
{
    return spread20(cx) | (spread20(cy) << 1) | (spread20(cz) << 2);
}

__device__ __host__ __forceinline__ unsigned long long cell_key(uint32_t level, uint32_t cx,
                                                                uint32_t cy, uint32_t cz)
{
    return ((unsigned long long)level << 60) | ((unsigned long long)cz << 40) |
           ((unsigned long long)cy << 20) | (unsigned long long)cx;
}
__device__ __host__ __forceinline__ uint64_t hash_key(unsigned long long k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// returns true and [start,end) when the voxel is occupied
__device__ __forceinline__ bool cell_lookup(const GridView& g, unsigned long long key,
                                            uint32_t& start, uint32_t& end)
{
    uint64_t slot = hash_key(key) & g.mask;
    for (;;)
    {
        // 16-byte entry, one dwordx4 load
        const uint4 e = *reinterpret_cast<const uint4*>(&g.table[slot]);
        const unsigned long long k = ((unsigned long long)e.y << 32) | e.x;
        if (k == key)
        {
            start = e.z;
            end   = e.w;
            return true;
        }
        if (k == CELL_EMPTY) return false;
        slot = (slot + 1) & g.mask;
    }
}

}  // namespace mp2p
