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

// two candidates per instruction (v_pk_add_f32 / v_pk_mul_f32), same roundings as dist2()
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f dist2_pk(v2f qx, v2f qy, v2f qz, v2f px, v2f py, v2f pz)
{
#pragma clang fp contract(off)
    const v2f dx = qx - px, dy = qy - py, dz = qz - pz;
    const v2f xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const v2f s  = xx + yy;
    return s + zz;
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

// ---- wave64 collectives ---------------------------------------------------------------------
// Reductions run on the DPP cross-lane path (row rotations inside each 16-lane row, then four
// v_readlane to combine the rows): ~12 issue slots and no LDS round trip, instead of the
// 6 x ds_bpermute (~100 cycles each) a __shfl_xor butterfly costs.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u0(uint32_t v)  // out-of-row source lanes read 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;

__device__ __forceinline__ float wave_min(float v)
{
    v = fminf(v, dpp_f<DPP_ROW_ROR1>(v));
    v = fminf(v, dpp_f<DPP_ROW_ROR2>(v));
    v = fminf(v, dpp_f<DPP_ROW_ROR4>(v));
    v = fminf(v, dpp_f<DPP_ROW_ROR8>(v));
    return fminf(fminf(readlane_f(v, 0), readlane_f(v, 16)), fminf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f<DPP_ROW_ROR1>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_ROR2>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_ROR4>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_ROR8>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// The same reductions for values known not to be NaN, on the order-preserving integer image of
// the float: v_min_i32 / v_max_i32 need no sNaN canonicalisation of their inputs (one v_max x,x per
// operand of every fminf on values that come out of DPP / readlane), the DPP move folds into
// the integer min, and the four rows are combined on the scalar unit.
__device__ __forceinline__ int f2key(float v)
{
    const int b = __builtin_bit_cast(int, v);
    return b ^ ((b >> 31) & 0x7FFFFFFF);
}
__device__ __forceinline__ float key2f(int k)
{
    return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7FFFFFFF));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_min_nn(float v)
{
    int k = f2key(v);
    k = min(k, dpp_i<DPP_ROW_ROR1>(k));
    k = min(k, dpp_i<DPP_ROW_ROR2>(k));
    k = min(k, dpp_i<DPP_ROW_ROR4>(k));
    k = min(k, dpp_i<DPP_ROW_ROR8>(k));
    const int a = __builtin_amdgcn_readlane(k, 0), b = __builtin_amdgcn_readlane(k, 16),
              c = __builtin_amdgcn_readlane(k, 32), d = __builtin_amdgcn_readlane(k, 48);
    return key2f(min(min(a, b), min(c, d)));
}
__device__ __forceinline__ float wave_max_nn(float v)
{
    int k = f2key(v);
    k = max(k, dpp_i<DPP_ROW_ROR1>(k));
    k = max(k, dpp_i<DPP_ROW_ROR2>(k));
    k = max(k, dpp_i<DPP_ROW_ROR4>(k));
    k = max(k, dpp_i<DPP_ROW_ROR8>(k));
    const int a = __builtin_amdgcn_readlane(k, 0), b = __builtin_amdgcn_readlane(k, 16),
              c = __builtin_amdgcn_readlane(k, 32), d = __builtin_amdgcn_readlane(k, 48);
    return key2f(max(max(a, b), max(c, d)));
}

// ... and for values that are >= +0 (squared distances, radii, +inf): the bit pattern itself
// orders like the value
__device__ __forceinline__ float wave_min_pos(float v)
{
    uint32_t k = __builtin_bit_cast(uint32_t, v);
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR1>((int)k));
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR2>((int)k));
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR4>((int)k));
    k = min(k, (uint32_t)dpp_i<DPP_ROW_ROR8>((int)k));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)k, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)k, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)k, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
    return __builtin_bit_cast(float, min(min(a, b), min(c, d)));
}
__device__ __forceinline__ float wave_max_pos(float v)
{
    uint32_t k = __builtin_bit_cast(uint32_t, v);
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR1>((int)k));
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR2>((int)k));
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR4>((int)k));
    k = max(k, (uint32_t)dpp_i<DPP_ROW_ROR8>((int)k));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)k, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)k, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)k, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)k, 48);
    return __builtin_bit_cast(float, max(max(a, b), max(c, d)));
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
// inclusive prefix sum across the 64 lanes: Hillis-Steele inside each 16-lane row on DPP
// (row_shr with zero fill), then the three row carries through v_readlane
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    v += dpp_u0<DPP_ROW_SHR1>(v);
    v += dpp_u0<DPP_ROW_SHR2>(v);
    v += dpp_u0<DPP_ROW_SHR4>(v);
    v += dpp_u0<DPP_ROW_SHR8>(v);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const int      row = lane >> 4;
    return v + (row > 0 ? t0 : 0u) + (row > 1 ? t1 : 0u) + (row > 2 ? t2 : 0u);
}
// inclusive prefix MAX across the 64 lanes (values >= 0; 0 = "nothing yet")
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, int lane)
{
    v = max(v, dpp_u0<DPP_ROW_SHR1>(v));
    v = max(v, dpp_u0<DPP_ROW_SHR2>(v));
    v = max(v, dpp_u0<DPP_ROW_SHR4>(v));
    v = max(v, dpp_u0<DPP_ROW_SHR8>(v));
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const int      row = lane >> 4;
    uint32_t       c   = 0;
    if (row > 0) c = t0;
    if (row > 1) c = max(c, t1);
    if (row > 2) c = max(c, t2);
    return max(v, c);
}
// lexicographic arg-min of (d2, idx) over the wave; the winner's triple is returned uniformly
__device__ __forceinline__ void wave_argmin(float& d, uint32_t& i, uint32_t& s)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const float    od = __shfl_xor(d, off, 64);
        const uint32_t oi = __shfl_xor(i, off, 64);
        const uint32_t os = __shfl_xor(s, off, 64);
        if (od < d || (od == d && oi < i)) d = od, i = oi, s = os;
    }
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
    // 32-bit multiplicative mix of the two key halves (a handful of VALU ops per lookup)
    uint32_t h = (uint32_t)k * 0x9E3779B1u ^ ((uint32_t)(k >> 32) * 0x85EBCA77u);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}

// false = the voxel is certainly empty (no probe needed); true = occupied, or no bitmap to tell
__device__ __forceinline__ bool occ_maybe(const GridView& g, uint32_t lev, uint32_t cx, uint32_t cy,
                                          uint32_t cz)
{
    const uint32_t off = g.occ_off[lev];
    if (off == OCC_NONE) return true;
    const uint32_t bx = cx >> 2, by = cy >> 2, bz = cz >> 2;
    if (bx >= g.occ_bx[lev] || by >= g.occ_by[lev] || bz >= g.occ_bz[lev]) return false;
    const unsigned long long w = g.occ[(size_t)off + ((size_t)bz * g.occ_by[lev] + by) * g.occ_bx[lev] + bx];
    return (w >> (((cz & 3u) << 4) | ((cy & 3u) << 2) | (cx & 3u))) & 1ull;
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

// true and [start, end) when voxel (cx, cy, cz) of level lev holds points.  occ_checked: the caller has
// already seen the voxel's occupancy bit set (the hash path then skips that test).
__device__ __forceinline__ bool voxel_range(const GridView& g, uint32_t lev, uint32_t cx, uint32_t cy,
                                            uint32_t cz, uint32_t& start, uint32_t& end, bool occ_checked)
{
    const unsigned long long off = g.dir_off[lev];
    if (off != DIR_NONE)
    {
        const uint32_t nx = g.occ_bx[lev] * 4u, ny = g.occ_by[lev] * 4u, nz = g.occ_bz[lev] * 4u;
        if (cx >= nx || cy >= ny || cz >= nz) return false;
        const uint2 e = g.dir[off + ((size_t)cz * ny + cy) * nx + cx];
        start = e.x, end = e.y;
        return e.y > e.x;
    }
    if (!occ_checked && !occ_maybe(g, lev, cx, cy, cz)) return false;
    return cell_lookup(g, cell_key(lev, cx, cy, cz), start, end);
}

}  // namespace mp2p
