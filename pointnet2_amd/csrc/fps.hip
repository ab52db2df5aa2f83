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
#include "pn2_device.h"

#include <limits.h>

namespace pn2 {

constexpr int kRefThreads = 512;  // tie rule modulus: reference blockDim (tf_sampling_g.cu:204)

// min(d, td) of tf_sampling_g.cu:144 as ONE v_min_f32 (the builtin adds a canonicalising v_max per
// operand). v_min_f32 returns the non-NaN operand, like CUDA's min(float,float).
#ifndef PN2_FPS_VMIN_ASM
#define PN2_FPS_VMIN_ASM 1
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

// Wave-wide max of a positive finite double (a (value:low) key, see fps_reg_kernel) WITHOUT the scalar
// unit: per DPP step two v_mov_b32_dpp fetch the partner lane's halves and one v_max_f64 combines.
// After the six steps lane 63 holds the wave maximum (rows 1/3 after row_bcast:15, rows 2/3 after
// row_bcast:31). For the two broadcast steps the unwritten rows keep `old` = the lane's own value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_f64_step(double v)
{
    const int hi = __double2hiint(v), lo = __double2loint(v);
    int ohi, olo;
    if (ROW_MASK == 0xf) {
        // every lane has a valid source: the destination needs no initial value (saves two v_mov)
        ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
        olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    } else {
        ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
        olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    }
    const double o = __hiloint2double(ohi, olo);
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));
    return r;
}
__device__ __forceinline__ double wave_max_f64_lane63(double v)
{
    v = dpp_max_f64_step<0xB1, 0xf>(v);    // quad_perm:[1,0,3,2]
    v = dpp_max_f64_step<0x4E, 0xf>(v);    // quad_perm:[2,3,0,1]
    v = dpp_max_f64_step<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_max_f64_step<0x140, 0xf>(v);   // row_mirror
    v = dpp_max_f64_step<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v = dpp_max_f64_step<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
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
// Keys: (value bits << 32) | (T*P - 1 - rank), compared as fp64 (header). The value
// is <= 1e38f < 0x7FF00000, so the pattern is never an fp64 Inf/NaN; small values give
// fp64 denormals, which gfx9 never flushes for v_max_f64 operands.
// Padding slots carry value +0.0: they can only tie with real zero-distance
// points, and rank 0 (k = 0, always real) then wins, as in the reference.
// ---------------------------------------------------------------------------
template <int T, int P, bool LDSXYZ>
__global__ __launch_bounds__(T) void fps_reg_kernel(int n, int m, int Q, const float *__restrict__ xyz,
                                                    int *__restrict__ out, float *__restrict__ out_xyz)
{
    constexpr int W = T / PN2_WAVE;
    constexpr int NS = T * P;                      // rank slots
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
        // mirrors are indexed by the key's low word (kMaxLow - rank): one shift-add to the address
        if (LDSXYZ) lds_rank[NS - 1 - r] = make_float4(x[p], y[p], z[p], __int_as_float(kk));
        else lds_k[NS - 1 - r] = kk;
    }
    __syncthreads();

    float sx, sy, sz;                                  // the point selected last (starts at k = 0 = rank 0)
    if (LDSXYZ) {
        const float4 s = lds_rank[NS - 1];
        sx = s.x; sy = s.y; sz = s.z;
    } else {
        sx = src[0]; sy = src[1]; sz = src[2];
    }
    if (t == 0) dst[0] = 0;                            // tf_sampling_g.cu:114-116

    const unsigned low0 = (unsigned)(NS - 1 - t * P);   // key low word of this thread's slot 0: larger = smaller rank
    // one round; `par` (the partial buffer parity) is a literal at both call sites so the slot
    // addresses fold to constants (scalar address arithmetic costs 4-cycle issue slots)
    auto round = [&](const int j, const int par) __attribute__((always_inline)) {
        // Lane arg-max as ONE v_max_f64 per slot: the 64-bit pattern (value bits : low key word) of a
        // slot, read as a double, is positive, finite (value <= 1e38f < 0x7FF00000) and ordered exactly
        // like the pair (value, smaller rank first); fp64 denormals are never flushed on gfx9.
        double kd[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float d = sqdist(x[p], y[p], z[p], sx, sy, sz);
            md[p] = vmin_f32(d, md[p]);                // min(d,td), :144
            kd[p] = __hiloint2double(__float_as_int(md[p]), (int)(low0 - (unsigned)p));
        }
#pragma unroll
        for (int st = 1; st < P; st <<= 1)             // tournament: depth log2(P), independent v_max_f64 per level
#pragma unroll
            for (int i = 0; i + st < P; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(kd[i]) : "v"(kd[i]), "v"(kd[i + st]));
        const double bestd = kd[0];
        // whole-wave key max in VALU only; lane 63 ends up with it and publishes it
        unsigned long long *slot = partial + par * W;
        {
            const double wd = wave_max_f64_lane63(bestd);
            if (lane == 63) reinterpret_cast<double *>(slot)[w] = wd;
        }
        __syncthreads();
        // block arg-max: v_max_f64 tournament over the W keys, every wave redundantly (wave-uniform data)
        const double *dslot = reinterpret_cast<const double *>(slot);
        double key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = dslot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
        const unsigned win = (unsigned)__double2loint(key[0]);   // low word of the winning key = mirror index
        int k;
        if (LDSXYZ) {
            const float4 s = lds_rank[win];            // same address in every lane: LDS broadcast
            sx = s.x; sy = s.y; sz = s.z;
            k = __float_as_int(s.w);
        } else {
            k = lds_k[win];
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
