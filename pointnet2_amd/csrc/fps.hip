// fps.hip -- farthest point sampling for gfx950 (MI355X).
//
// Replaces farthestpointsamplingKernel/Launcher (reference
// tf_ops/sampling/tf_sampling_g.cu:105-170, :203-205). Results are index-exact
// with that kernel run in CPU arithmetic (oracle/pn2_oracle.c), including its
// tie rule: among equal running-min distances the point with the smallest
// (k mod 512, k) wins (512 = the reference's blockDim).
//
// Design (DESIGN.md "FPS"). FPS is a chain of m-1 dependent block-wide
// arg-max steps; it is latency bound, not HBM bound. One workgroup owns one
// cloud for the whole chain:
//   * every point's x,y,z and running min-distance live in VGPRs for all m
//     rounds (no per-round memory traffic at all; the reference round-trips
//     `temp` through global memory every round);
//   * points are dealt to threads in TIE-RANK order: thread t holds ranks
//     t*P .. t*P+P-1 where rank(k) = (k mod 512)*ceil(n/512) + k/512, so the
//     reference's tie rule becomes "smaller rank wins";
//   * a candidate is the 64-bit key (value bits : T*P - 1 - rank). Read as
//     an fp64 number that pattern is positive, finite and ordered exactly like
//     (larger value, then smaller rank), so EVERY level of the arg-max is a
//     plain v_max_f64: P-1 per lane, six DPP steps per wave (two v_mov_b32_dpp
//     + one v_max_f64 each, no scalar unit), W-1 per wave after the barrier;
//   * one s_barrier per round: lane 63 of each wave publishes the wave key in
//     a parity double-buffered LDS slot array, every wave max-reduces the W
//     keys from broadcast reads and broadcast-reads the winner's (x,y,z,k) from
//     an LDS mirror of the cloud kept in rank order: two dependent LDS trips.
// Round time on MI355X at B=32, N=4096 (512 threads x 8 points), scripts/fps_prod_lab.hip:
//   650 ns  first version (int compares + v_cndmask, DPP + v_readlane + ballot, 3 LDS trips)
//   557 ns  asm DPP ladder, broadcast select, -fno-slp-vectorize (hipcc's v_pk_* packing cost 11 %)
//   506 ns  64-bit rank keys (2 LDS trips), key low word formed in VALU
//   425 ns  v_max_f64 on the keys at all three levels
// Rejected after measurement: ds_max_u64 LDS atomics (~30 cycles per same-address op);
// scalar-unit resolution of lane/slot (SALU ops cost a 4-cycle issue slot); box-pruned
// updates over Morton-sorted 64-point regions (exact, but 784 ns: the per-slot scalar
// branches and a non-rank-ordered lane arg-max cost more than the skipped distances).
// Clouds too large for the register tiers fall back to a global-memory tier
// that keeps the running distances in the caller's `temp` buffer.
#include "fps_body.h"
#include "fps_pruned_body.h"

#include <limits.h>

namespace pn2 {

template <int T, int P, bool LDSXYZ>
__global__ __launch_bounds__(T) void fps_reg_kernel(int n, int m, int Q, const float *__restrict__ xyz,
                                                    int *__restrict__ out, float *__restrict__ out_xyz)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fps_reg_body<T, P, LDSXYZ, false>(n, m, Q, blockIdx.x, xyz, out, out_xyz, nullptr, smem);
}

// Pruned tier (fps_pruned_body.h): kd-grouped slots, per-round box tests, only the groups the new sample can reach are
// updated. 2049..8192 rank slots, chains long enough to pay for the kd build.
template <int P, int GS>
__global__ __launch_bounds__(kPrT) void fps_pruned_kernel(int n, int m, int Q, const float *__restrict__ xyz,
                                                          int *__restrict__ out, float *__restrict__ out_xyz)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fps_pruned_body<P, GS, false>(n, m, Q, blockIdx.x, xyz, out, out_xyz, nullptr, smem);
}

// ---------------------------------------------------------------------------
// Generic tier: any n. Running distances in global `temp` (b*n floats), cloud
// re-read from global/L2 each round, 64-bit (value, tie-key) reduction.
// Slow path for clouds beyond the register tiers (n > 16384).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m, 64);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void fps_generic_kernel(int n, int m, const float *__restrict__ xyz,
                                                           float *__restrict__ temp, int *__restrict__ out,
                                                           float *__restrict__ out_xyz)
{
    constexpr int T = 1024, W = T / PN2_WAVE;
    __shared__ unsigned long long part[2][W];
    const int cloud = blockIdx.x;
    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    float *__restrict__ mind = temp + (size_t)cloud * n;
    int *__restrict__ dst = out + (size_t)cloud * m;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int k = t; k < n; k += T) mind[k] = 1e38f;
    float *__restrict__ dxyz = out_xyz ? out_xyz + (size_t)cloud * m * 3 : nullptr;
    int cur = 0;
    if (t == 0) {
        dst[0] = 0;
        if (dxyz) { dxyz[0] = src[0]; dxyz[1] = src[1]; dxyz[2] = src[2]; }
    }
    // each thread only ever touches its own mind[k] (k = t mod T): no barrier needed for temp
    for (int j = 1; j < m; ++j) {
        const float sx = src[(size_t)cur * 3 + 0], sy = src[(size_t)cur * 3 + 1], sz = src[(size_t)cur * 3 + 2];
        unsigned long long best = 0ull;  // (value bits << 32) | ~tiekey ; real candidates are > 0
        for (int k = t; k < n; k += T) {
            const float d = sqdist(src[(size_t)k * 3 + 0], src[(size_t)k * 3 + 1], src[(size_t)k * 3 + 2], sx, sy, sz);
            const float d2 = __builtin_fminf(d, mind[k]);
            mind[k] = d2;
            const unsigned tiekey = ((unsigned)(k & (kRefThreads - 1)) << 22) | (unsigned)(k >> 9);
            const unsigned long long key =
                ((unsigned long long)(unsigned)__float_as_int(d2) << 32) | (unsigned long long)(0xFFFFFFFFu - tiekey);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const unsigned long long o = shfl_xor_u64(best, s);
            best = o > best ? o : best;
        }
        if (lane == 0) part[j & 1][w] = best;
        __syncthreads();
        unsigned long long v = (lane < W) ? part[j & 1][lane] : 0ull;
#pragma unroll
        for (int s = 1; s < W; s <<= 1) {
            const unsigned long long o = shfl_xor_u64(v, s);
            v = o > v ? o : v;
        }
        const unsigned tiekey = 0xFFFFFFFFu - (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        cur = (int)(((tiekey & 0x3FFFFFu) << 9) | (tiekey >> 22));
        if (t == 0) {
            dst[j] = cur;
            if (dxyz) {
                dxyz[j * 3 + 0] = src[(size_t)cur * 3 + 0];
                dxyz[j * 3 + 1] = src[(size_t)cur * 3 + 1];
                dxyz[j * 3 + 2] = src[(size_t)cur * 3 + 2];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int T, int P, bool LDSXYZ>
static int launch_reg(int b, int n, int m, int Q, const float *inp, int *out, float *oxyz, hipStream_t st)
{
    const size_t lds = 256 + (LDSXYZ ? sizeof(float4) : sizeof(int)) * (size_t)T * P;
    auto kern = fps_reg_kernel<T, P, LDSXYZ>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    if (int rc = launch(kern, dim3(b), dim3(T), lds, st, n, m, Q, inp, out, oxyz)) return rc;
    return PN2_OK;
}

template <int P, int GS>
static int launch_pruned(int b, int n, int m, int Q, const float *inp, int *out, float *oxyz, hipStream_t st)
{
    const size_t lds = fps_pruned_lds_bytes(P);
    auto kern = fps_pruned_kernel<P, GS>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    return launch(kern, dim3(b), dim3(kPrT), lds, st, n, m, Q, inp, out, oxyz);
}

static bool pruned_covers(int ranks) { return ranks > 2048 && ranks <= 8192; }

static int fps_launch_pruned(int b, int n, int m, const float *inp, int *out, float *oxyz, hipStream_t st)
{
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    const int ranks = kRefThreads * Q;
    if (!pruned_covers(ranks)) return PN2_E_ARG;
    // 32 groups either way: 16 slots per thread in groups of 2, 32 in groups of 4
    if (ranks <= 4096) return launch_pruned<16, 2>(b, n, m, Q, inp, out, oxyz, st);
    return launch_pruned<32, 4>(b, n, m, Q, inp, out, oxyz, st);
}

constexpr int kMaxLdsSlots = 8192;     // 256 B + 16 B per rank slot <= 160 KiB
constexpr int kMaxRegPoints = 16384;

template <int T, bool LDSXYZ>
static int dispatch_p(int P, int b, int n, int m, int Q, const float *inp, int *out, float *oxyz, hipStream_t st)
{
    switch (P) {
    case 1: return launch_reg<T, 1, LDSXYZ>(b, n, m, Q, inp, out, oxyz, st);
    case 2: return launch_reg<T, 2, LDSXYZ>(b, n, m, Q, inp, out, oxyz, st);
    case 4: return launch_reg<T, 4, LDSXYZ>(b, n, m, Q, inp, out, oxyz, st);
    case 8: return launch_reg<T, 8, LDSXYZ>(b, n, m, Q, inp, out, oxyz, st);
    case 16: return launch_reg<T, 16, LDSXYZ>(b, n, m, Q, inp, out, oxyz, st);
    case 32:
        if constexpr (T <= 512) return launch_reg<T, 32, LDSXYZ>(b, n, m, Q, inp, out, oxyz, st);
        break;
    default: break;
    }
    return PN2_E_ARG;
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// force a (T,P) configuration: used by the tuning harness (bench.py --fps-sweep)
static int fps_launch_config(int T, int P, int b, int n, int m, const float *inp, int *out, float *oxyz,
                             hipStream_t st)
{
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    if ((long long)T * P < (long long)kRefThreads * Q) return PN2_E_ARG;
    const bool lds = (long long)T * P <= kMaxLdsSlots;
    switch (T) {
    case 256: return lds ? dispatch_p<256, true>(P, b, n, m, Q, inp, out, oxyz, st) : dispatch_p<256, false>(P, b, n, m, Q, inp, out, oxyz, st);
    case 512: return lds ? dispatch_p<512, true>(P, b, n, m, Q, inp, out, oxyz, st) : dispatch_p<512, false>(P, b, n, m, Q, inp, out, oxyz, st);
    case 1024: return lds ? dispatch_p<1024, true>(P, b, n, m, Q, inp, out, oxyz, st) : dispatch_p<1024, false>(P, b, n, m, Q, inp, out, oxyz, st);
    default: return PN2_E_ARG;
    }
}

}  // namespace pn2

extern "C" long long pn2_fps_temp_floats(int b, int n)
{
    if (b <= 0 || n <= 0) return 0;
    return n > pn2::kMaxRegPoints ? (long long)b * n : 0;
}

static int fps_entry(int b, int n, int m, const float *inp, float *temp, int *out, float *out_xyz, void *stream,
                     int variant = PN2_FPS_AUTO)
{
    using namespace pn2;
    if (m <= 0 || b == 0) return PN2_OK;          // tf_sampling_g.cu:106
    if (b < 0 || n <= 0) return PN2_E_SHAPE;
    if (!inp || !out) return PN2_E_NULL;
    if ((long long)b * n * 3 > INT_MAX || (long long)b * m * 3 > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (n > kMaxRegPoints) {
        if (!temp) return PN2_E_NULL;
        if (int rc = launch(fps_generic_kernel, dim3(b), dim3(1024), 0, st, n, m, inp, temp, out, out_xyz)) return rc;
        return PN2_OK;
    }
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    const int ranks = kRefThreads * Q;
    if (variant == PN2_FPS_PRUNED || (variant == PN2_FPS_AUTO && fps_pruned_pays(ranks, m)))
        return fps_launch_pruned(b, n, m, inp, out, out_xyz, st);
    // default geometry (measured, scripts/fps_prod_lab.hip, ns per round at n = 1024/2048/4096/8192):
    // 256 threads with the packed distance update 273/312/402/585, 512 threads (scalar) 288/326/393/552
    const int T = ranks <= 2048 ? 256 : 512;
    const int P = next_pow2((ranks + T - 1) / T);
    return fps_launch_config(T, P, b, n, m, inp, out, out_xyz, st);
}

extern "C" int pn2_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream)
{
    return fps_entry(b, n, m, inp, temp, out, nullptr, stream);
}

extern "C" int pn2_farthest_point_sample_gather(int b, int n, int m, const float *inp, float *temp, int *out,
                                                float *out_xyz, void *stream)
{
    if (m > 0 && b > 0 && !out_xyz) return PN2_E_NULL;
    return fps_entry(b, n, m, inp, temp, out, out_xyz, stream);
}

// Same operator with the tier chosen by the caller (tests, A/B timing; results never depend on it): PN2_FPS_AUTO = the
// size rule of pn2_farthest_point_sample, PN2_FPS_FULL = every point updated every round (fps_reg_body), PN2_FPS_PRUNED =
// the kd-grouped tier (PN2_E_ARG outside 2049..8192 rank slots). out_xyz may be NULL.
extern "C" int pn2_farthest_point_sample_variant(int variant, int b, int n, int m, const float *inp, float *temp, int *out,
                                                 float *out_xyz, void *stream)
{
    if (variant < PN2_FPS_AUTO || variant > PN2_FPS_PRUNED) return PN2_E_ARG;
    return fps_entry(b, n, m, inp, temp, out, out_xyz, stream, variant);
}

// tuning / test hook: run the register tier with an explicit geometry
extern "C" int pn2_farthest_point_sample_ex(int T, int P, int b, int n, int m, const float *inp, int *out, void *stream)
{
    if (m <= 0 || b <= 0 || n <= 0 || !inp || !out) return PN2_E_ARG;
    if (n > pn2::kMaxRegPoints) return PN2_E_TOO_LARGE;
    return pn2::fps_launch_config(T, P, b, n, m, inp, out, nullptr, pn2::as_stream(stream));
}
