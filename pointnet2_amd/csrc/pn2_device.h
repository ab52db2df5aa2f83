// pn2_device.h -- shared device helpers for the gfx950 kernels.
//
// Arithmetic contract (DESIGN.md "Exactness"): every distance is the fp32
// value ((dx*dx)+(dy*dy))+(dz*dz) with one rounding per operation -- what the
// reference CPU build computes. The translation units are compiled with
// -ffp-contract=off; sqdist() additionally spells the roundings out with
// __fmul_rn/__fadd_rn so a stray contraction can never fuse them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pn2ops.h"

#define PN2_WAVE 64

namespace pn2 {

__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = __fsub_rn(ax, bx);
    const float dy = __fsub_rn(ay, by);
    const float dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- DPP cross-lane moves (wave64, gfx9 DPP controls) ---------------------
// quad_perm[1,0,3,2]=0xB1  quad_perm[2,3,0,1]=0x4E  row_half_mirror=0x141  row_mirror=0x140
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

// Max of a signed int over each 16-lane DPP row; every lane of the row ends
// up holding the row's maximum (4 butterfly steps, no LDS, no SALU).
__device__ __forceinline__ int row16_max_i32(int v)
{
    v = max(v, dpp_mov<0xB1>(v));
    v = max(v, dpp_mov<0x4E>(v));
    v = max(v, dpp_mov<0x141>(v));
    v = max(v, dpp_mov<0x140>(v));
    return v;
}

// Wave-wide max, returned as a wave-uniform value (lives in an SGPR):
// 4 DPP steps + 4 v_readlane + 3 s_max.
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = row16_max_i32(v);
    const int r0 = __builtin_amdgcn_readlane(v, 0);
    const int r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32);
    const int r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

__device__ __forceinline__ int row16_min_i32(int v)
{
    v = min(v, dpp_mov<0xB1>(v));
    v = min(v, dpp_mov<0x4E>(v));
    v = min(v, dpp_mov<0x141>(v));
    v = min(v, dpp_mov<0x140>(v));
    return v;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// number of set bits of `mask` below this lane (v_mbcnt_lo + v_mbcnt_hi)
__device__ __forceinline__ int mbcnt(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// launch status -> C ABI return (positive hipError_t)
inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace pn2
