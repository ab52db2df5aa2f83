// fps_body.h -- device body of the register-resident FPS tier (shared by fps.hip and sa_fused.hip).
// See fps.hip for the design notes and the measurement history.
#pragma once
#include "pn2_device.h"

namespace pn2 {

typedef unsigned long long __attribute__((address_space(1))) pn2_gu64;   // global-memory u64 for agent-scope atomics

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
// After the six steps lane 63 holds the wave maximum. Every step uses row_mask 0xf with bound_ctrl: the
// destination then needs no initial value. For the two row_bcast steps that means the rows WITHOUT a
// source (row 0 for row_bcast:15, rows 0-1 for row_bcast:31) combine their key with 0 -- or with whatever
// the register held -- which is harmless: lane 63's result depends only on rows 1 and 3 after step five
// and on row 3 after step six, and those rows have sources. (The row_mask 0xa / 0xc form of round 1 had to
// pre-load the destination with the lane's own halves: two v_mov and an s_nop more per step on the
// serial path, 408 -> 393 ns per round at 512 x 8.)
template <int CTRL>
__device__ __forceinline__ double dpp_max_f64_step(double v)
{
    const int hi = __double2hiint(v), lo = __double2loint(v);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const double o = __hiloint2double(ohi, olo);
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));
    return r;
}
__device__ __forceinline__ double wave_max_f64_lane63(double v)
{
    v = dpp_max_f64_step<0xB1>(v);    // quad_perm:[1,0,3,2]
    v = dpp_max_f64_step<0x4E>(v);    // quad_perm:[2,3,0,1]
    v = dpp_max_f64_step<0x141>(v);   // row_half_mirror
    v = dpp_max_f64_step<0x140>(v);   // row_mirror
    v = dpp_max_f64_step<0x142>(v);   // row_bcast:15: row r += row r-1 (rows 1, 3 matter)
    v = dpp_max_f64_step<0x143>(v);   // row_bcast:31: rows 2, 3 += lane 31 (row 3 matters)
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
// Packed fp32 (VOP3P v_pk_*_f32: two IEEE fp32 operations per lane and instruction, each rounded
// exactly like its scalar twin). The distance update is the VALU-throughput part of a round and packs
// perfectly: two slots per instruction, the selected point broadcast from the LOW half of a register pair
// (op_sel_hi:[1,0]), its negation folded into the add (neg_lo/neg_hi; a + (-s) == a - s bit for bit).
// Measured (scripts/ubench_pk.hip, 2 waves per SIMD): 3.76 cycles per v_pk op vs 3.26 per scalar op,
// i.e. 1.7x the fp32 rate. hipcc's own SLP packing of the scalar code was slower (-fno-slp-vectorize):
// it assembles the pairs with extra moves on the critical path; here the slots LIVE as pairs.
typedef float pn2_f2 __attribute__((ext_vector_type(2)));
#ifndef PN2_FPS_PACK_512
#define PN2_FPS_PACK_512 0
#endif

__device__ __forceinline__ pn2_f2 pk_sub_bcast_lo(pn2_f2 a, pn2_f2 s)   // a - s.x in both halves
{
    pn2_f2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(s));
    return r;
}

// There is deliberately NO "broadcast the HIGH half" twin (op_sel:[0,1] op_sel_hi:[1,1]). Round 5 found that form returning
// wrong values whenever a wave of ANOTHER kernel issues MFMAs on the same SIMD (scripts/pk_hazard_lab.hip: 2e5 of 1.3e9
// lanes disagree with the scalar evaluation beside a v_mfma_f32_32x32x16_bf16 loop, none alone, none beside a VALU loop, no
// number of wait states helps; the low-half form, with or without the neg modifiers, and hipcc's own packed code -- which
// only ever emits the low-half form -- are exact in all three situations). Rounds 1-4 used it for the y coordinate: every
// packed FPS variant (256 and 1024 threads) picked different samples beside the fused MLP kernels -- the "flaky result
// under two-stream use" of profiles/r04/geometry_prefetch_experiment.txt; tests/test_multistream_gpu.py now covers it.
// The selected point therefore lives in THREE register pairs with the coordinate in the low half.

__device__ __forceinline__ pn2_f2 pk_mul(pn2_f2 a, pn2_f2 b)
{
    pn2_f2 r;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ pn2_f2 pk_add(pn2_f2 a, pn2_f2 b)
{
    pn2_f2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// PUBLISH: thread 0 additionally stores every selected index as an 8-byte {tag, index} granule
// with ONE write-through (sc1, agent-scope relaxed atomic) store, so other workgroups of the same
// launch can consume the samples while the chain is still running (sa_fused.hip; hand-off form R2 of
// the CDNA guide: the data is the flag, no fences).
template <int T, int P, bool LDSXYZ, bool PUBLISH>
__device__ __forceinline__ void fps_reg_body(int n, int m, int Q, int cloud, const float *__restrict__ xyz,
                                             int *__restrict__ out, float *__restrict__ out_xyz,
                                             unsigned long long *__restrict__ tagged, char *smem,
                                             unsigned tag = 1u)
{
    // A chain is one dependent instruction after another on b CUs; beside another stream's kernels (the layer stacks of the
    // previous batch, pointnet2_amd/geometry.py) its waves would queue behind theirs at every issue. Highest wave priority: the
    // neighbours lose a few issue slots on b of 256 CUs, the chain keeps its pace (sem_seg training step with the geometry one
    // step ahead: 10.5 ms without this line, profiles/r05/geometry_ahead.txt).
#ifndef PN2_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int W = T / PN2_WAVE;
    constexpr int NS = T * P;                      // rank slots
    unsigned long long *partial = reinterpret_cast<unsigned long long *>(smem);   // [2][W] (256 B reserved)
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);                    // [T*P] when LDSXYZ
    int *lds_k = reinterpret_cast<int *>(smem + 256);                             // [T*P] otherwise

    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    int *__restrict__ dst = out + (size_t)cloud * m;
    float *__restrict__ dxyz = out_xyz ? out_xyz + (size_t)cloud * m * 3 : nullptr;   // fused gather_point
    pn2_gu64 *gtag = PUBLISH ? (pn2_gu64 *)(tagged + (size_t)cloud * m) : nullptr;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    // slots live as register pairs (see pk_* above) -- except at 512 threads, where the packed form
    // measured slower (2 waves per SIMD: 438 vs 411 ns per round at 512x8; PN2_FPS_PACK_512 is the lab switch)
    constexpr bool PACKED = (P % 2 == 0) && (T != 512 || PN2_FPS_PACK_512);
    constexpr int PH = PACKED ? P / 2 : 1;
    float x[P], y[P], z[P], md[P];
    pn2_f2 xx[PH], yy[PH], zz[PH];
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
        if (PACKED) {
            if (p & 1) { xx[p / 2].y = x[p]; yy[p / 2].y = y[p]; zz[p / 2].y = z[p]; }
            else { xx[p / 2].x = x[p]; yy[p / 2].x = y[p]; zz[p / 2].x = z[p]; }
        }
    }
    __syncthreads();

    pn2_f2 sxy = {0.f, 0.f}, syy = {0.f, 0.f}, szk = {0.f, 0.f};   // the selected point, each coordinate in the LOW half of a pair (packed path)
    float sx, sy, sz;                                  // the point selected last (starts at k = 0 = rank 0)
    if (LDSXYZ) {
        const float4 s = lds_rank[NS - 1];
        sx = s.x; sy = s.y; sz = s.z;
    } else {
        sx = src[0]; sy = src[1]; sz = src[2];
    }
    sxy.x = sx; sxy.y = sy; syy.x = sy; szk.x = sz;
    if (t == 0) {
        dst[0] = 0;                                    // tf_sampling_g.cu:114-116
        if (PUBLISH) __hip_atomic_store(gtag, (unsigned long long)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const unsigned low0 = (unsigned)(NS - 1 - t * P);   // key low word of this thread's slot 0: larger = smaller rank
    int kprev = 0;                                     // index selected by the previous round (stored one round late)
    // one round; `par` (the partial buffer parity) is a literal at both call sites so the slot
    // addresses fold to constants (scalar address arithmetic costs 4-cycle issue slots)
    auto round = [&](const int j, const int par) __attribute__((always_inline)) {
        // Lane arg-max as ONE v_max_f64 per slot: the 64-bit pattern (value bits : low key word) of a
        // slot, read as a double, is positive, finite (value <= 1e38f < 0x7FF00000) and ordered exactly
        // like the pair (value, smaller rank first); fp64 denormals are never flushed on gfx9.
        double kd[P];
        if (PACKED) {
            // breadth first (the asm statements are volatile, so this IS the issue order): every
            // operation is >= PH instructions away from its producer
            pn2_f2 dx[PH], dy[PH], dz[PH];
#pragma unroll
            for (int h = 0; h < PH; ++h) dx[h] = pk_sub_bcast_lo(xx[h], sxy);
#pragma unroll
            for (int h = 0; h < PH; ++h) dy[h] = pk_sub_bcast_lo(yy[h], syy);
#pragma unroll
            for (int h = 0; h < PH; ++h) dz[h] = pk_sub_bcast_lo(zz[h], szk);
#pragma unroll
            for (int h = 0; h < PH; ++h) dx[h] = pk_mul(dx[h], dx[h]);
#pragma unroll
            for (int h = 0; h < PH; ++h) dy[h] = pk_mul(dy[h], dy[h]);
#pragma unroll
            for (int h = 0; h < PH; ++h) dz[h] = pk_mul(dz[h], dz[h]);
#pragma unroll
            for (int h = 0; h < PH; ++h) dx[h] = pk_add(dx[h], dy[h]);
#pragma unroll
            for (int h = 0; h < PH; ++h) dx[h] = pk_add(dx[h], dz[h]);
#pragma unroll
            for (int h = 0; h < PH; ++h) {
                md[2 * h] = vmin_f32(dx[h].x, md[2 * h]);          // min(d,td), :144
                md[2 * h + 1] = vmin_f32(dx[h].y, md[2 * h + 1]);
                kd[2 * h] = __hiloint2double(__float_as_int(md[2 * h]), (int)(low0 - (unsigned)(2 * h)));
                kd[2 * h + 1] = __hiloint2double(__float_as_int(md[2 * h + 1]), (int)(low0 - (unsigned)(2 * h + 1)));
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float d = sqdist(x[p], y[p], z[p], sx, sy, sz);
                md[p] = vmin_f32(d, md[p]);                // min(d,td), :144
                kd[p] = __hiloint2double(__float_as_int(md[p]), (int)(low0 - (unsigned)p));
            }
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
        unsigned win;                                  // low word of the winning key = mirror index
        // the PREVIOUS round's index leaves under the latency of the key reads (behind the winner read its address
        // arithmetic and exec juggling sat on the serial path: 14 cycles per round, scripts/ubench_xchg.hip kinds 14 / 15)
        auto store_prev = [&]() __attribute__((always_inline)) {
            if (t == 0 && j > 1) {
                dst[j - 1] = kprev;
                if (PUBLISH)
                    __hip_atomic_store(gtag + (j - 1), ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)kprev,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        if (W >= 16) {
        // lane i reads key i mod W (ONE LDS read per wave instead of W/2 broadcast reads) and the W
        // keys are combined across lanes with log2(W) butterfly DPP steps: every lane ends with the max
        double kq = dslot[lane & (W - 1)];
        store_prev();
        if (W >= 2) kq = dpp_max_f64_step<0xB1>(kq);    // lane ^ 1
        if (W >= 4) kq = dpp_max_f64_step<0x4E>(kq);    // lane ^ 2
        if (W >= 8) kq = dpp_max_f64_step<0x141>(kq);   // other quad of the half row
        if (W >= 16) kq = dpp_max_f64_step<0x140>(kq);  // other half row
        win = (unsigned)__double2loint(kq);
        } else {
        // W <= 8: broadcast-read all keys, tournament on wave-uniform data (measured faster: 408 vs 441 ns)
        double key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = dslot[i];
        store_prev();
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
        win = (unsigned)__double2loint(key[0]);
        }
        int k;
        if (LDSXYZ) {
            const float4 s = lds_rank[win];            // same address in every lane: LDS broadcast
            sx = s.x; sy = s.y; sz = s.z;
            sxy.x = s.x; sxy.y = s.y; syy.x = s.y; szk.x = s.z; szk.y = s.w;
            k = __float_as_int(s.w);
        } else {
            k = lds_k[win];
            sx = src[(size_t)k * 3 + 0]; sy = src[(size_t)k * 3 + 1]; sz = src[(size_t)k * 3 + 2];
            sxy.x = sx; sxy.y = sy; syy.x = sy; szk.x = sz;
        }
        kprev = k;
    };
    int j = 1;
    for (; j + 1 < m; j += 2) {
        round(j, 1);
        round(j + 1, 0);
    }
    if (j < m) round(j, 1);
    if (t == 0 && m > 1) {                             // the last round's index
        dst[m - 1] = kprev;
        if (PUBLISH)
            __hip_atomic_store(gtag + (m - 1), ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)kprev, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    fps_gather_epilogue<T>(m, src, dst, dxyz);
}


}  // namespace pn2
