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
//     t*P .. t*P+P-1 where rank(k) = (k mod 512)*ceil(n/512) + k/512. The
//     reference's tie rule then degenerates to "first slot, lowest lane,
//     lowest wave", which costs nothing in the reduction;
//   * values are compared as signed-int bit patterns (they are >= 0, or -1
//     for padding slots), so the wave arg-max is 6 single-instruction
//     v_max_i32_dpp steps (quad_perm x2, row_half_mirror, row_mirror,
//     row_bcast:15, row_bcast:31), one v_readlane, one v_cmp_eq (the ballot)
//     and an s_ff1;
//   * one s_barrier per round: per-wave partials go through a parity
//     double-buffered LDS slot array; every wave redundantly picks the winner
//     from broadcast reads of the <=8 partials (a short VALU select chain on
//     wave-uniform data; the 16-wave geometry reduces them with DPP instead),
//     then broadcast-reads the winner's xyz from an LDS copy of the cloud.
// Measured on MI355X (scripts/fps_lab.hip, B=32 N=4096): the round costs
// ~540 ns at 512 threads x 8 points, of which ~290 ns is the synchronisation
// chain and ~250 ns the distance update; 256x16 and 1024x4 are slower
// (a lone wave per SIMD issues VALU at ~3 cycles/op, 16 waves pay for 16
// redundant reductions). LDS atomics (ds_max_u64) were tried and rejected:
// same-address LDS atomics serialise at ~30 cycles each.
// Clouds too large for the register tiers fall back to a global-memory tier
// that keeps the running distances in the caller's `temp` buffer.
#include "pn2_device.h"

#include <limits.h>

namespace pn2 {

constexpr int kRefThreads = 512;  // tie rule modulus: reference blockDim (tf_sampling_g.cu:204)

// min(d, td) of tf_sampling_g.cu:144 as ONE v_min_f32 (the builtin adds a canonicalising v_max per
// operand). v_min_f32 returns the non-NaN operand, like CUDA's min(float,float).
#ifndef PN2_FPS_WINNER_WRITES
#define PN2_FPS_WINNER_WRITES 0
#endif
#ifndef PN2_FPS_VMIN_ASM
#define PN2_FPS_VMIN_ASM 1
#endif
#ifndef PN2_FPS_TREE_ARGMAX
#define PN2_FPS_TREE_ARGMAX 0
#endif
__device__ __forceinline__ float vmin_f32(float a, float b)
{
#if PN2_FPS_VMIN_ASM
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fminf(a, b);
#endif
}

// Wave-wide signed max as a wave-uniform scalar: six single-instruction DPP steps (the two wait
// states a DPP source needs after a VALU write are spelled as s_nop 1), then one v_readlane.
#define PN2_DPP_MAX(v, ctrl) \
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ int wave_max_i32_fast(int v)
{
    PN2_DPP_MAX(v, "quad_perm:[1,0,3,2] row_mask:0xf");
    PN2_DPP_MAX(v, "quad_perm:[2,3,0,1] row_mask:0xf");
    PN2_DPP_MAX(v, "row_half_mirror row_mask:0xf");
    PN2_DPP_MAX(v, "row_mirror row_mask:0xf");
    PN2_DPP_MAX(v, "row_bcast:15 row_mask:0xa");
    PN2_DPP_MAX(v, "row_bcast:31 row_mask:0xc");
    return __builtin_amdgcn_readlane(v, 63);
}

// Fused gather_point: new_xyz[j] = inp[idx[j]], written once after the last round by the whole
// workgroup (coalesced). Doing it inside the round loop costs: the extra live scalars made hipcc
// switch the arg-max compares from SGPR-pair to VCC encodings, which serialised the selects
// (+27 % per round, measured).
template <int T>
__device__ __forceinline__ void fps_gather_epilogue(int m, const float *__restrict__ src, const int *dst,
                                                    float *__restrict__ dxyz)
{
    if (!dxyz) return;                             // uniform
    __syncthreads();                               // thread 0's index stores are visible to the workgroup
    for (int j = threadIdx.x; j < m; j += T) {
        const int k = __builtin_nontemporal_load(dst + j);
        dxyz[j * 3 + 0] = src[(size_t)k * 3 + 0];
        dxyz[j * 3 + 1] = src[(size_t)k * 3 + 1];
        dxyz[j * 3 + 2] = src[(size_t)k * 3 + 2];
    }
}

// ---------------------------------------------------------------------------
// Register-resident tier.  T threads, P points per thread, n <= T*P.
//
// Every slot r = t*P+p is a tie RANK. The cloud is mirrored in LDS in rank order as
// (x, y, z, bits(k)) so the winner's coordinates AND its original index come back in
// one broadcast ds_read_b128 (LDSXYZ). Without the LDS mirror (clouds of 8193..16384
// points) only a rank -> k table lives in LDS and the winner is re-read from L2.
//
// A wave's candidate is published as ONE 64-bit key
//     key = (value bits << 32) | (0xFFFFFFFF - rank)
// whose unsigned order is exactly "larger value, then smaller rank". After the
// barrier every wave max-reduces the W keys from broadcast reads (a depth-log2(W)
// tree of v_cmp_gt_u64 + 2 v_cndmask on wave-uniform data) and reads the winner's
// point: two dependent LDS trips per round. (The first version selected a byte
// offset, then fetched the index, then the coordinates: three trips; the s_memtime
// profile in scripts/fps_prof.hip showed that chain at ~540 of ~1340 cycles.)
// Padding slots carry value +0.0: they can only tie with real zero-distance
// points, and rank 0 (k = 0, always real) then wins, as in the reference.
// ---------------------------------------------------------------------------
template <int T, int P, bool LDSXYZ>
__global__ __launch_bounds__(T) void fps_reg_kernel(int n, int m, int Q, const float *__restrict__ xyz,
                                                    int *__restrict__ out, float *__restrict__ out_xyz)
{
    constexpr int W = T / PN2_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *partial = reinterpret_cast<unsigned long long *>(smem);   // [2][W] (256 B reserved)
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);                    // [T*P] when LDSXYZ
    int *lds_k = reinterpret_cast<int *>(smem + 256);                             // [T*P] otherwise

    const int cloud = blockIdx.x;
    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    int *__restrict__ dst = out + (size_t)cloud * m;
    float *__restrict__ dxyz = out_xyz ? out_xyz + (size_t)cloud * m * 3 : nullptr;   // fused gather_point
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    float x[P], y[P], z[P], md[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = t * P + p;                       // tie rank of this slot
        const int k = (r % Q) * kRefThreads + r / Q;   // original point index
        const bool valid = (r < kRefThreads * Q) && (k < n);
        const int kk = valid ? k : 0;
        x[p] = valid ? src[(size_t)kk * 3 + 0] : 0.0f;
        y[p] = valid ? src[(size_t)kk * 3 + 1] : 0.0f;
        z[p] = valid ? src[(size_t)kk * 3 + 2] : 0.0f;
        md[p] = valid ? 1e38f : 0.0f;                  // tf_sampling_g.cu:118; padding: see header
        if (LDSXYZ) lds_rank[r] = make_float4(x[p], y[p], z[p], __int_as_float(kk));
        else lds_k[r] = kk;
    }
    __syncthreads();

    float sx, sy, sz;                                  // the point selected last (starts at k = 0 = rank 0)
    if (LDSXYZ) {
        const float4 s = lds_rank[0];
        sx = s.x; sy = s.y; sz = s.z;
    } else {
        sx = src[0]; sy = src[1]; sz = src[2];
    }
    if (t == 0) dst[0] = 0;                            // tf_sampling_g.cu:114-116

    const unsigned low0 = 0xFFFFFFFFu - (unsigned)(t * P);   // key low word of this thread's slot 0
    // one round; `par` (the partial buffer parity) is a literal at both call sites so the slot
    // addresses fold to constants (scalar address arithmetic costs 4-cycle issue slots)
    auto round = [&](const int j, const int par) __attribute__((always_inline)) {
#if PN2_FPS_TREE_ARGMAX
        // distances first, then a pairwise tournament over the P slots: independent compares at
        // each level (strict >, the lower slot survives a tie, :146: slots ascend in rank)
        int tv[P], tp[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float d = sqdist(x[p], y[p], z[p], sx, sy, sz);
            md[p] = vmin_f32(d, md[p]);                // min(d,td), :144
            tv[p] = __float_as_int(md[p]);             // >= 0: int order == float order
            tp[p] = p;
        }
#pragma unroll
        for (int st = 1; st < P; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < P; i += 2 * st) {
                const bool c = tv[i + st] > tv[i];
                tp[i] = c ? tp[i + st] : tp[i];
                tv[i] = c ? tv[i + st] : tv[i];
            }
        const int bv = tv[0], bp = tp[0];
#else
        int bv = -1, bp = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float d = sqdist(x[p], y[p], z[p], sx, sy, sz);
            md[p] = vmin_f32(d, md[p]);                // min(d,td), :144
            const int iv = __float_as_int(md[p]);      // >= 0: int order == float order
            if (iv > bv) { bv = iv; bp = p; }          // strict >, :146 (slots ascend in rank)
        }
#endif
        // wave arg-max, ties -> lowest lane (= lowest rank). The low key word of this lane's
        // candidate is formed in VALU before the reduction; the winning lane stores its own key
        // (no v_readlane round trip through the scalar unit for the payload).
        unsigned mylow = low0 - (unsigned)bp;
#if PN2_FPS_WINNER_WRITES
        asm volatile("" : "+v"(mylow));                // keep the slot select out of the store branch
#endif
        const int wm = wave_max_i32_fast(bv);
        const int wl = __builtin_ctzll(__ballot(bv == wm));   // the max is held by some lane
        unsigned long long *slot = partial + par * W;
#if PN2_FPS_WINNER_WRITES
        if (lane == wl) slot[w] = ((unsigned long long)(unsigned)bv << 32) | (unsigned long long)mylow;
#else
        const unsigned wlow = (unsigned)__builtin_amdgcn_readlane((int)mylow, wl);
        if (lane == 0) slot[w] = ((unsigned long long)(unsigned)wm << 32) | (unsigned long long)wlow;
#endif
        __syncthreads();
        // block arg-max: unsigned max of the W keys, every wave redundantly, on wave-uniform data
        unsigned long long key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = slot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st) key[i] = key[i + st] > key[i] ? key[i + st] : key[i];
        const unsigned rank = 0xFFFFFFFFu - (unsigned)key[0];
        int k;
        if (LDSXYZ) {
            const float4 s = lds_rank[rank];           // same address in every lane: LDS broadcast
            sx = s.x; sy = s.y; sz = s.z;
            k = __float_as_int(s.w);
        } else {
            k = lds_k[rank];
            sx = src[(size_t)k * 3 + 0]; sy = src[(size_t)k * 3 + 1]; sz = src[(size_t)k * 3 + 2];
        }
        if (t == 0) dst[j] = k;
    };
    int j = 1;
    for (; j + 1 < m; j += 2) {
        round(j, 1);
        round(j + 1, 0);
    }
    if (j < m) round(j, 1);
    fps_gather_epilogue<T>(m, src, dst, dxyz);
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
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, st, n, m, Q, inp, out, oxyz);
    return launch_status();
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

static int fps_entry(int b, int n, int m, const float *inp, float *temp, int *out, float *out_xyz, void *stream)
{
    using namespace pn2;
    if (m <= 0 || b == 0) return PN2_OK;          // tf_sampling_g.cu:106
    if (b < 0 || n <= 0) return PN2_E_SHAPE;
    if (!inp || !out) return PN2_E_NULL;
    if ((long long)b * n * 3 > INT_MAX || (long long)b * m * 3 > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (n > kMaxRegPoints) {
        if (!temp) return PN2_E_NULL;
        hipLaunchKernelGGL(fps_generic_kernel, dim3(b), dim3(1024), 0, st, n, m, inp, temp, out, out_xyz);
        return launch_status();
    }
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    const int ranks = kRefThreads * Q;
    // default geometry (measured, scripts/fps_lab.hip): 256 threads up to 1024 ranks, else 512
    const int T = ranks <= 1024 ? 256 : 512;
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

// tuning / test hook: run the register tier with an explicit geometry
extern "C" int pn2_debug_fps_config(int T, int P, int b, int n, int m, const float *inp, int *out, void *stream)
{
    if (m <= 0 || b <= 0 || n <= 0 || !inp || !out) return PN2_E_ARG;
    if (n > pn2::kMaxRegPoints) return PN2_E_TOO_LARGE;
    return pn2::fps_launch_config(T, P, b, n, m, inp, out, nullptr, pn2::as_stream(stream));
}
