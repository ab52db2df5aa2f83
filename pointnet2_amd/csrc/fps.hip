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
#include "fps_batch_body.h"

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

// Batched tier (fps_batch_body.h): the pruned tier's slots and boxes, several samples per arg-max exchange.
// P = slots per updater thread (kBtUT of them), GS slots per group
template <int P, int GS>
__global__ __launch_bounds__(kBtT) void fps_batch_kernel(int n, int m, int Q, const float *__restrict__ xyz,
                                                         int *__restrict__ out, float *__restrict__ out_xyz)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fps_batch_body<P, GS, false>(n, m, Q, blockIdx.x, xyz, out, out_xyz, nullptr, smem);
}

// ---------------------------------------------------------------------------
// Input that is ALREADY in farthest-point order (round 5; pn2_farthest_point_sample_ordered).
// The second and later levels of every network sample from the previous level's samples
// (pointnet2_sem_seg.py:28-31: l2 = sa(l1_xyz, ...)): the first m of them, in the order level 1 picked them, are a
// farthest-point order of the subset with the same fp32 running distances, so farthest_point_sample(m, new_xyz_1) is
// 0, 1, ..., m-1 -- unless the reference's tie rule (smallest (k mod 512, k), tf_sampling_g.cu:146,153-163), applied to the
// RENUMBERED points, breaks an exact tie the other way, or the cloud has run out of distinct points (every further sample
// is index 0). Whether it is the identity is decidable without the chain: sample i is selected at step i iff no other
// point j beats it there, i.e. iff for every j != i the running distance r_i(j) = min_{s<i} d(j, s) and the selection
// value v_i = r_i(i) satisfy (r_i(j), tie key j) < (v_i, tie key i). All r and v are prefix minima of the n x m distance
// matrix: n independent rows, no reduction. fps_verify_kernel evaluates them (parts x b workgroups) and flags a cloud on
// the first violation; the chain kernel behind it writes 0..m-1 for unflagged clouds and runs the real chain for
// flagged ones, so the result never depends on the guess.
// ---------------------------------------------------------------------------
constexpr int kVerT = 1024;
// a < b ? x : y as a bit select on a mask the compiler cannot see through (it turned the plain select into an exec-mask
// branch around the computation of x: one exposed LDS latency per step; a volatile barrier on x serialised the reads)
__device__ __forceinline__ float pick_lt(int a, int b, float x, float y)
{
    unsigned sel = (unsigned)((a - b) >> 31);                  // all ones when a < b (|a - b| < 2^31 here)
    asm("" : "+v"(sel));
    return __uint_as_float((__float_as_uint(x) & sel) | (__float_as_uint(y) & ~sel));
}
// Work decomposition: n x (m - 1) prefix-minimum entries per cloud would be m - 1 DEPENDENT steps per thread if a thread
// owned a row (60 us at m = 256: one LDS read + ten dependent vector instructions per step and nothing to overlap them with).
// A row is therefore cut into C chunks of L source samples (C a power of two <= 16, L ~ 32): chunk minima first, an exclusive
// minimum over the earlier chunks as the carry, then the chunk's own steps against the selection values -- 2 L dependent
// steps. A workgroup is C waves = C chunks x 64 points (sample reads are LDS broadcasts; small workgroups, because the work of
// a cloud is vector-issue bound on the CUs it lands on: 1024-thread workgroups measured 10.8 us at m = 128).
// The selection values v_i = r_i(i) are rows of the same matrix; every workgroup first rebuilds them for its cloud
// (triangular, the same chunks, LDS atomic minimum of the fp32 bit patterns: the values are >= 0) -- m^2 / 2 distance tests
// next to the workgroup's own 64 x 2 m, cheaper than a launch boundary.
__global__ __launch_bounds__(kVerT) void fps_verify_kernel(int n, int m, int C, int L, const float *__restrict__ xyz, int *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *sm = reinterpret_cast<float4 *>(smem);             // [m]: x, y, z of sample e and the value sample e + 1 was selected with
    float *part = reinterpret_cast<float *>(sm + m);           // [C][64] chunk minima
    const int cloud = blockIdx.y, t = threadIdx.x;
    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    const int T = C * 64;                                      // one wave per chunk, 64 points per workgroup
    for (int i = t; i < m; i += T)
        sm[i] = make_float4(src[(size_t)i * 3 + 0], src[(size_t)i * 3 + 1], src[(size_t)i * 3 + 2], 1e38f);   // 1e38: tf_sampling_g.cu:118
    __syncthreads();
    const int E = m - 1;                                       // source samples 0 .. m - 2 (the last sample updates nothing)
    // selection values: v_i = min_{e < i} d(i, e), i = 1 .. m - 1 (tf_sampling_g.cu:144)
    const int mpad = (m + 63) & ~63;
    for (int item = t; item < mpad * C; item += T) {
        // c is wave-uniform (mpad and the wave's first item are multiples of 64); the compiler is told so, the loops below are
        // scalar-controlled and hand-blocked by 8 (a runtime-bound loop is not unrolled for this target: one LDS read and ten
        // dependent instructions per step otherwise)
        const int c = __builtin_amdgcn_readfirstlane(item / mpad), i = item - c * mpad;
        const int e0 = c * L;
        const int eend = min(min(e0 + L, E), __builtin_amdgcn_readfirstlane(i | 63));   // no lane of the wave has a source at or beyond it
        if (e0 >= eend) continue;
        const float4 p = sm[min(i, m - 1)];
        float r = 1e38f;
        int eb = e0;
        for (; eb + 8 <= eend; eb += 8) {                        // whole blocks: eight independent LDS reads in flight
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 q = sm[eb + u];
                r = vmin_f32(pick_lt(eb + u, i, sqdist(p.x, p.y, p.z, q.x, q.y, q.z), 1e38f), r);
            }
        }
        for (; eb < eend; ++eb) {
            const float4 q = sm[eb];
            r = vmin_f32(pick_lt(eb, i, sqdist(p.x, p.y, p.z, q.x, q.y, q.z), 1e38f), r);
        }
        if (i >= 1 && i < m) atomicMin(reinterpret_cast<unsigned *>(&sm[i - 1].w), __float_as_uint(r));
    }
    __syncthreads();
    // every point against every step
    constexpr int JB = 64;
    const int c = __builtin_amdgcn_readfirstlane(t / JB), jl = t - c * JB;
    const int j = blockIdx.x * JB + jl;
    const bool live = j < n;
    const int jc = live ? j : 0;
    const float px = src[(size_t)jc * 3 + 0], py = src[(size_t)jc * 3 + 1], pz = src[(size_t)jc * 3 + 2];
    const int e0 = c * L, e1 = min(e0 + L, E);
    float r = 1e38f;
    if (C > 1) {                                               // kernel-uniform
        float a = 1e38f;
        int eb = e0;
        for (; eb + 8 <= e1; eb += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 q = sm[eb + u];
                a = vmin_f32(sqdist(px, py, pz, q.x, q.y, q.z), a);
            }
        }
        for (; eb < e1; ++eb) {
            const float4 q = sm[eb];
            a = vmin_f32(sqdist(px, py, pz, q.x, q.y, q.z), a);
        }
        part[c * JB + jl] = a;
        __syncthreads();
        for (int c2 = 0; c2 < c; ++c2) r = vmin_f32(part[c2 * JB + jl], r);
    }
    const unsigned tkj = ((unsigned)(j & (kRefThreads - 1)) << 22) | (unsigned)(j >> 9);
    // a coordinate that is not a finite number: v_min drops NaNs where the chain's arg-max sees them -- leave it to the chain
    int bad = !(__builtin_fabsf(px) <= 3.402823466e38f && __builtin_fabsf(py) <= 3.402823466e38f && __builtin_fabsf(pz) <= 3.402823466e38f);
    auto step = [&](int e) {
        const float4 q = sm[e];                                     // sample e and the value sample e + 1 was selected with
        r = vmin_f32(sqdist(px, py, pz, q.x, q.y, q.z), r);         // r_i(j), i = e + 1
        const int i = e + 1;
        const unsigned tki = ((unsigned)(i & (kRefThreads - 1)) << 22) | (unsigned)(i >> 9);
        bad |= (int)(j != i) & ~((int)(r < q.w) | ((int)(r == q.w) & (int)(tkj >= tki)));   // (value, key) order of tf_sampling_g.cu:146,153-163
    };
    int eb = e0;
    for (; eb + 8 <= e1; eb += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) step(eb + u);
    }
    for (; eb < e1; ++eb) step(eb);
    if (live && (bad & 1)) flags[cloud] = 1;
}

// chunks of ~32 source samples, at most 16 of them
static void fps_verify_shape(int m, int &C, int &L)
{
    const int E = m - 1;
    C = 1;
    while (C < 16 && C * 32 < E) C <<= 1;
    L = (E + C - 1) / C;
}

static int launch_verify(int b, int n, int m, const float *inp, int *flags, hipStream_t st)
{
    int C, L;
    fps_verify_shape(m, C, L);
    return launch(fps_verify_kernel, dim3((n + 63) / 64, b), dim3(C * 64), sizeof(float4) * (size_t)m + sizeof(float) * C * 64, st, n, m, C,
                  L, inp, flags);
}

// the chain behind the verifier: identity for unflagged clouds, the real chain (and a cleared flag) for the others
template <int P>
__global__ __launch_bounds__(256) void fps_reg_cond_kernel(int n, int m, int Q, const float *__restrict__ xyz, int *__restrict__ out,
                                                           float *__restrict__ out_xyz, int *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cloud = blockIdx.x, t = threadIdx.x;
    if (flags[cloud] == 0) {                                     // block-uniform
        int *__restrict__ dst = out + (size_t)cloud * m;
        for (int j = t; j < m; j += 256) dst[j] = j;
        if (out_xyz) {
            const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
            float *__restrict__ d = out_xyz + (size_t)cloud * m * 3;
            for (int e = t; e < m * 3; e += 256) d[e] = src[e];
        }
        return;
    }
    fps_reg_body<256, P, true, false>(n, m, Q, cloud, xyz, out, out_xyz, nullptr, smem);
    if (t == 0) flags[cloud] = 0;                                // the workspace is left as it was found
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

template <int P, int GS>
static int launch_batch(int b, int n, int m, int Q, const float *inp, int *out, float *oxyz, hipStream_t st)
{
    const size_t lds = fps_batch_lds_bytes(P);
    auto kern = fps_batch_kernel<P, GS>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    return launch(kern, dim3(b), dim3(kBtT), lds, st, n, m, Q, inp, out, oxyz);
}

static int fps_launch_batch(int b, int n, int m, const float *inp, int *out, float *oxyz, hipStream_t st)
{
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    const int ranks = kRefThreads * Q;
    if (!fps_batch_covers(ranks)) return PN2_E_ARG;
    if constexpr (kBtUT == 512) {
        if (ranks <= 1024) return launch_batch<2, 2>(b, n, m, Q, inp, out, oxyz, st);   // 8 groups, one per updater wave
        if (ranks <= 2048) return launch_batch<4, 2>(b, n, m, Q, inp, out, oxyz, st);   // 16 groups
    } else if (ranks <= 2048) {
        return PN2_E_ARG;
    }
    // 32 groups either way: at 512 updater threads 8 slots per thread in groups of 2 or 16 in groups of 4
    if (ranks <= 4096) return launch_batch<4096 / kBtUT, 4096 / kBtUT / (32 / (kBtUT / 64))>(b, n, m, Q, inp, out, oxyz, st);
    return launch_batch<8192 / kBtUT, 8192 / kBtUT / (32 / (kBtUT / 64))>(b, n, m, Q, inp, out, oxyz, st);
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

template <int P>
static int launch_cond(int b, int n, int m, int Q, const float *inp, int *out, float *oxyz, int *flags, hipStream_t st)
{
    const size_t lds = 256 + sizeof(float4) * (size_t)256 * P;
    auto kern = fps_reg_cond_kernel<P>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    return launch(kern, dim3(b), dim3(256), lds, st, n, m, Q, inp, out, oxyz, flags);
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
    if (variant == PN2_FPS_BATCH || (variant == PN2_FPS_AUTO && fps_batch_pays(ranks, m))) return fps_launch_batch(b, n, m, inp, out, out_xyz, st);
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
    if (variant < PN2_FPS_AUTO || variant > PN2_FPS_BATCH) return PN2_E_ARG;
    return fps_entry(b, n, m, inp, temp, out, out_xyz, stream, variant);
}

// farthest_point_sample for input the caller BELIEVES to be in farthest-point order already (the previous level's samples:
// pointnet2_sem_seg.py:28-31, pointnet2_cls_ssg.py:27-29): verified in parallel, the chain runs only for clouds where the
// belief is wrong -- the result is pn2_farthest_point_sample's whatever the input is. ws: pn2_fps_ordered_ws_bytes(b) bytes,
// zeroed by the caller ONCE (every call leaves it zeroed). Outside 2 <= m <= min(n, 1024), n <= 2048 there is nothing to
// gain and the call is pn2_farthest_point_sample_gather / pn2_farthest_point_sample (PN2_E_TOO_LARGE beyond the register
// tiers, where that operator needs its temp buffer). out_xyz may be NULL.
extern "C" long long pn2_fps_ordered_ws_bytes(int b) { return b > 0 ? (long long)b * 4 : 0; }

extern "C" int pn2_farthest_point_sample_ordered(int b, int n, int m, const float *inp, int *out, float *out_xyz, void *ws,
                                                 void *stream)
{
    using namespace pn2;
    if (m <= 0 || b == 0) return PN2_OK;
    if (b < 0 || n <= 0) return PN2_E_SHAPE;
    if (!inp || !out) return PN2_E_NULL;
    if (n > kMaxRegPoints) return PN2_E_TOO_LARGE;
    if (m < 2 || m > n || m > 1024 || n > 2048) return fps_entry(b, n, m, inp, nullptr, out, out_xyz, stream);
    if (b > 65535) return fps_entry(b, n, m, inp, nullptr, out, out_xyz, stream);   // beyond the check launch's grid.y: the plain operator (same result)
    if (!ws) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    int *flags = static_cast<int *>(ws);
    if (int rc = launch_verify(b, n, m, inp, flags, st)) return rc;
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    const int P = next_pow2((kRefThreads * Q + 255) / 256);
    switch (P) {
    case 2: return launch_cond<2>(b, n, m, Q, inp, out, out_xyz, flags, st);
    case 4: return launch_cond<4>(b, n, m, Q, inp, out, out_xyz, flags, st);
    case 8: return launch_cond<8>(b, n, m, Q, inp, out, out_xyz, flags, st);
    default: return PN2_E_ARG;
    }
}

// test hook: the verifier of pn2_farthest_point_sample_ordered alone. flags (b ints, zeroed by the caller) comes back 1 for the
// clouds whose farthest-point sampling is NOT 0 .. m-1 (exactly those: the first violated step is where the chain leaves the
// identity). Same envelope as the short cut: 2 <= m <= min(n, 1024), n <= 2048.
extern "C" int pn2_fps_ordered_check(int b, int n, int m, const float *inp, int *flags, void *stream)
{
    using namespace pn2;
    if (b <= 0 || !inp || !flags) return PN2_E_ARG;
    if (m < 2 || m > n || m > 1024 || n > 2048 || b > 65535) return PN2_E_ARG;
    return launch_verify(b, n, m, inp, flags, as_stream(stream));
}

// tuning / test hook: run the register tier with an explicit geometry
extern "C" int pn2_farthest_point_sample_ex(int T, int P, int b, int n, int m, const float *inp, int *out, void *stream)
{
    if (m <= 0 || b <= 0 || n <= 0 || !inp || !out) return PN2_E_ARG;
    if (n > pn2::kMaxRegPoints) return PN2_E_TOO_LARGE;
    return pn2::fps_launch_config(T, P, b, n, m, inp, out, nullptr, pn2::as_stream(stream));
}
