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

#include <mutex>
#include <tuple>
#include <unordered_map>
#include <utility>

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

// The same value for use as the high word of a (distance : index) ordering key read as fp64 (three_nn,
// kNN): the last addition takes |operands| (free VOP3 source modifiers; the operands are squares, so
// nothing changes for numbers), which clears the sign of a propagated NaN. A NaN with its sign bit set --
// x86 produces such NaNs, and NaN coordinates propagate them -- would otherwise read as a NEGATIVE finite
// double and win every v_min_f64; a positive NaN pattern sorts above +inf and is never admitted, like
// `d < best` being false in the reference (tf_interpolate.cpp:74-93).
__device__ __forceinline__ float sqdist_key(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = __fsub_rn(ax, bx);
    const float dy = __fsub_rn(ay, by);
    const float dz = __fsub_rn(az, bz);
    return __fadd_rn(__builtin_fabsf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))), __builtin_fabsf(__fmul_rn(dz, dz)));
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

// Block number -> (cloud, part) for kernels that launch `parts` workgroups per cloud, all of which read
// the same cloud (ball query: the staged points; group / interpolate: the gathered source rows). Workgroup i runs on XCD i % 8 (observed dispatch rule; used for speed only), and every XCD has
// its own L2: with the plain map i -> (i / parts, i % parts) the workgroups of one cloud land on different
// XCDs and each of them misses its L2 (rocprofv3, round 1: 6.7x the algorithmic read traffic). Here the
// workgroups of a cloud share i % 8, so the first one's fetch serves the others from that XCD's L2.
// Clouds beyond the last multiple of eight use the plain map. Bijective for every (parts, b).
__device__ __forceinline__ void decode_cloud_block(unsigned i, int parts, int b, int &cloud, int &part)
{
    const unsigned full = (unsigned)(b & ~7) * (unsigned)parts;
    if (i < full) {
        const unsigned x = i & 7u, s = i >> 3;
        cloud = (int)((s / (unsigned)parts) * 8u + x);
        part = (int)(s % (unsigned)parts);
    } else {
        cloud = (int)(i / (unsigned)parts);
        part = (int)(i % (unsigned)parts);
    }
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- host side: launching ---------------------------------------------------------------------
// pn2::launch enqueues `kern` with hipLaunchKernel and returns THAT call's hipError_t as the C ABI's
// positive return code. (hipGetLastError() after a <<<>>> launch would also report -- and clear -- a
// stale error left behind by some earlier, unrelated runtime call of the thread.) Arguments are
// converted to the kernel's parameter types first, like a direct call would.
template <typename... KArgs, size_t... I>
inline int launch_packed(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st,
                         std::tuple<KArgs...> &vals, std::index_sequence<I...>)
{
    void *ptrs[] = {const_cast<void *>(static_cast<const void *>(&std::get<I>(vals)))..., nullptr};
    return (int)hipLaunchKernel(reinterpret_cast<const void *>(kern), grid, block, ptrs, lds, st);
}

template <typename... KArgs, typename... Args>
inline int launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args &&...args)
{
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count must match the kernel's parameter list");
    std::tuple<KArgs...> vals{static_cast<KArgs>(std::forward<Args>(args))...};
    return launch_packed(kern, grid, block, lds, st, vals, std::index_sequence_for<KArgs...>{});
}

// ---- zero-fill: a kernel of the library's own, NEVER hipMemsetAsync --------------------------------------------------------
// Every entry point is meant to be capturable into a HIP graph, and a hipMemsetAsync captured into a graph is a memset NODE.
// On this runtime (ROCm 7.2) a replayed memset node does not reliably clear: round 5's serving loop and round 6's reproducer
// (scripts/memset_node_repro.hip / .py, scripts/stale_granule_repro.py; profiles/r06/stale_granules.md) read the buffer
// right behind the node and found it filled with a 16-byte pattern made of ANOTHER kernel's launch arguments (an element
// count and a pointer of an eager torch kernel launched between the replays), or not touched at all -- the node's fill
// parameters are not kept for the life of the executable graph. A kernel node's arguments are. So every clear the library
// needs (accumulation targets of the gradient scatters, counters, the overlapped launch's granules in its eager form) is
// this kernel: 16-byte stores over the aligned middle, bytes at the ragged ends.
static __global__ void pn2_clear_kernel(unsigned char *p, size_t head, size_t n16, size_t tail)
{
    const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    uint4 *mid = reinterpret_cast<uint4 *>(p + head);
    for (size_t i = i0; i < n16; i += step) mid[i] = make_uint4(0u, 0u, 0u, 0u);
    if (i0 < head) p[i0] = 0;
    if (i0 < tail) p[head + n16 * 16 + i0] = 0;
}

inline int clear_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return 0;
    unsigned char *p = static_cast<unsigned char *>(ptr);
    size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
    if (head > bytes) head = bytes;
    const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
    size_t blocks = (n16 + 256 * 4 - 1) / (256 * 4);              // four stores per thread, at most eight workgroups per CU
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    return launch(pn2_clear_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, head, n16, tail);
}

// Kernels that ask for more than 48 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize
// raised. The attribute call costs microseconds, so it is made once per (device, kernel) and the granted
// size remembered -- the whole 160 KiB of a gfx950 CU when the kernel has no static LDS, else exactly what
// is asked for (and again only if a later launch asks for more). This is the library's only process
// state, and it is idempotent (every launch passes its exact size).
template <typename K>
inline int allow_dynamic_lds(K kern, size_t bytes)
{
    if (bytes <= 48 * 1024) return 0;
    if (bytes > 160 * 1024) return PN2_E_TOO_LARGE;
    static std::mutex mu;
    static std::unordered_map<unsigned long long, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const void *fn = reinterpret_cast<const void *>(kern);
    const unsigned long long key = (unsigned long long)reinterpret_cast<uintptr_t>(fn) ^ ((unsigned long long)dev << 56);
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return 0;
    size_t want = 160 * 1024;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
    if (e != hipSuccess) {                                         // static LDS in the kernel: ask for the exact size
        (void)hipGetLastError();
        want = bytes;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != hipSuccess) return (int)e;
    }
    granted[key] = want;
    return 0;
}

// Workgroups of `kern` (threads, dynamic LDS bytes) the current device can hold at once: occupancy query x CU
// count, cached per (device, kernel, LDS size). Negative = -hipError_t.
template <typename K>
inline int resident_workgroups(K kern, int threads, size_t lds)
{
    static std::mutex mu;
    static std::unordered_map<unsigned long long, int> cache;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    const void *fn = reinterpret_cast<const void *>(kern);
    const unsigned long long key = ((unsigned long long)reinterpret_cast<uintptr_t>(fn) * 1315423911ull) ^
                                   ((unsigned long long)dev << 56) ^ (unsigned long long)lds;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds);
    if (e != hipSuccess) return -(int)e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return -(int)e;
    return cache[key] = per_cu * cus;
}

}  // namespace pn2
