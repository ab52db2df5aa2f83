// ball_query_body.h -- device body of the ball-query sweep (shared by ball_query.hip and sa_fused.hip).
// See ball_query.hip for the design notes and the measurement history.
#pragma once
#include "pn2_device.h"

#include <math.h>

namespace pn2 {

typedef unsigned long long __attribute__((address_space(1))) pn2_gu64b;

constexpr int kBqThreads = 512;
constexpr int kBqWaves = kBqThreads / PN2_WAVE;
constexpr int kBqQpw = 2;              // queries swept together by one wave (each LDS read serves both)
constexpr int kBqMaxLdsPoints = 9600;  // 16 B/point + row buffers must stay under 160 KiB

// popcount(mask) + acc as two VALU ops on a VGPR accumulator. On gfx9-class CUs a scalar op costs a
// SIMD issue slot of 4 cycles (a VALU op 2), and the first version of this kernel was bound by its
// ~60 scalar ops per trip (rocprofv3: 64 us at the metric shape); the hit counters therefore live in
// VGPRs (wave-uniform values) and only the exec-mask updates of the predicated stores stay scalar.
__device__ __forceinline__ int vbcnt_acc(unsigned long long mask, int acc)
{
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2\n\tv_bcnt_u32_b32 %0, %3, %0"
        : "=&v"(r)
        : "s"((unsigned)mask), "v"(acc), "s"((unsigned)(mask >> 32)));
    return r;
}

// A query's hits are kept as a BITMAP while the cloud is swept: the ballot of chunk c (64 candidates)
// is the c-th 64-bit word, parked in lane c of two VGPRs (one v_cndmask per half word, no
// LDS traffic, no exec-mask juggling). One "window" is 64 chunks = 4096 candidates; bq_flush turns
// the window's bitmap into the ordered index list: a wave-wide prefix sum of the word popcounts gives
// every lane its first output slot, then each lane peels its set bits in ascending order.
// The ordered-compaction-per-chunk version of this kernel spent more issue slots on appending hits
// (7 VALU + 3 SALU per chunk and query) than on the distances themselves.
// Park the two queries' ballots of chunk c in lane c of their bitmap registers (`sel` = lane id == c,
// one v_cmp per chunk shared by the four half words). v_writelane_b32 would need two scalar operands
// (value + lane select), which gfx9's single constant-bus port only allows through M0.
__device__ __forceinline__ void bq_park(unsigned &a_lo, unsigned &a_hi, unsigned &b_lo, unsigned &b_hi,
                                        unsigned long long ma, unsigned long long mb, bool sel)
{
    a_lo = sel ? (unsigned)ma : a_lo;
    a_hi = sel ? (unsigned)(ma >> 32) : a_hi;
    b_lo = sel ? (unsigned)mb : b_lo;
    b_hi = sel ? (unsigned)(mb >> 32) : b_hi;
}

__device__ __forceinline__ int wave_prefix_sum_incl(int v)
{
    // Hillis-Steele over DPP row shifts, then the two cross-row broadcasts
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += t;    // row_shr:1
    t = __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); v += t;    // row_shr:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += t;    // row_shr:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); v += t;    // row_shr:8
    t = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); v += t;   // row_bcast:15 -> rows 1,3
    t = __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); v += t;   // row_bcast:31 -> rows 2,3
    return v;
}

// Append the window's hits (bitmap words mlo/mhi, lane c = chunk c) to rowbuf in ascending index order.
// `done` = hits already in rowbuf (wave-uniform); returns the new total (may exceed nsample).
__device__ __forceinline__ int bq_flush(unsigned mlo, unsigned mhi, int window_base, int done, int nsample,
                                        int *rowbuf, int lane)
{
    const int pc = __popc(mlo) + __popc(mhi);
    const int incl = wave_prefix_sum_incl(pc);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int pos = done + incl - pc;                       // first output slot of this lane's word
    unsigned long long word = ((unsigned long long)mhi << 32) | mlo;
    const int kbase = window_base + lane * 64;
    while (__any(word != 0ull && pos < nsample)) {
        if (word != 0ull && pos < nsample) rowbuf[pos] = kbase + __builtin_ctzll(word);
        word &= word - 1ull;
        ++pos;
    }
    return done + total;
}

template <bool LDS_CLOUD, bool FUSE>
__device__ __forceinline__ void bq_emit(size_t row, int nsample, int cnt, const int *rowbuf,
                                        const float4 *cloud, const float *__restrict__ data, float qx, float qy,
                                        float qz, int *__restrict__ idx, int *__restrict__ pts_cnt,
                                        float *__restrict__ grouped, int subtract, int lane)
{
    const int first = cnt > 0 ? rowbuf[0] : 0;   // the first hit pads the row; zeros when the ball is empty
    for (int l = lane; l < nsample; l += 64) {
        const int v = (l < cnt) ? rowbuf[l] : first;
        if (idx) idx[row * nsample + l] = v;
        if (FUSE) {
            float gx, gy, gz;
            if (LDS_CLOUD) {
                const float4 p = cloud[v];
                gx = p.x; gy = p.y; gz = p.z;
            } else {
                gx = data[(size_t)v * 3 + 0]; gy = data[(size_t)v * 3 + 1]; gz = data[(size_t)v * 3 + 2];
            }
            if (subtract) { gx = __fsub_rn(gx, qx); gy = __fsub_rn(gy, qy); gz = __fsub_rn(gz, qz); }
            float *o = grouped + (row * nsample + l) * 3;
            o[0] = gx; o[1] = gy; o[2] = gz;
        }
    }
    if (lane == 0 && pts_cnt) pts_cnt[row] = cnt;
}

// Wait until the FPS workgroup of the same launch has published sample j of this cloud and return its
// index (consumer side of the R2 granule hand-off: ONE 8-byte agent-scope relaxed load per poll, the
// tag travels with the data, no fence). The spin is bounded: a launch whose producers are not
// resident would otherwise hang the GPU; it traps instead.
__device__ __forceinline__ int bq_poll_sample(const unsigned long long *tagged)
{
    const pn2_gu64b *g = (const pn2_gu64b *)tagged;
    for (unsigned it = 0;; ++it) {
        const unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == 1u) return (int)(unsigned)v;
        __builtin_amdgcn_s_sleep(16);
        if (it > (1u << 23)) __builtin_trap();   // ~10 s: the longest legal chain (n = m = 8192) takes < 10 ms
    }
}

// One workgroup's share of the ball queries of cloud `bi`: queries [q0, q1).
// POLL: the query points are not read from xyz2; they are the FPS samples of the same launch, taken
// from the tagged index stream as soon as they exist (and written to new_xyz on the way).
template <bool LDS_CLOUD, bool FUSE, bool POLL>
__device__ __forceinline__ void bq_block_body(int n, int m, int nsample, float thr, int bi, int q0, int q1,
                                              const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                              const unsigned long long *__restrict__ tagged,
                                              float *__restrict__ new_xyz, int *__restrict__ idx,
                                              int *__restrict__ pts_cnt, float *__restrict__ grouped, int subtract,
                                              char *smem)
{
    float4 *cloud = reinterpret_cast<float4 *>(smem);                                   // [n] when LDS_CLOUD
    int *rowbuf_all = reinterpret_cast<int *>(smem + (LDS_CLOUD ? sizeof(float4) * (size_t)((n + 127) & ~127) : 0));

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *__restrict__ data = xyz1 + (size_t)bi * n * 3;
    int *rowbuf0 = rowbuf_all + (w * kBqQpw + 0) * nsample;
    int *rowbuf1 = rowbuf_all + (w * kBqQpw + 1) * nsample;

    // LDS copy of the cloud, padded to a multiple of 128 with points at +inf: a padded candidate's
    // distance is +inf (or NaN), never < thr, so the sweep needs no bounds predicate.
    const int npad = (n + 127) & ~127;
    if (LDS_CLOUD) {
        for (int k = t; k < npad; k += kBqThreads) {
            if (k < n) {
                const float *p = data + (size_t)k * 3;
                cloud[k] = make_float4(p[0], p[1], p[2], 0.0f);
            } else {
                cloud[k] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
            }
        }
        __syncthreads();
    }

    for (int j = q0 + w * kBqQpw; j < q1; j += kBqWaves * kBqQpw) {
        const bool two = (j + 1) < q1;                       // wave-uniform
        const size_t row0 = (size_t)bi * m + j;
        const size_t row1 = row0 + (two ? 1 : 0);
        float ax, ay, az, bx, by, bz;
        if (POLL) {
            const int ka = bq_poll_sample(tagged + row0);
            const int kb = bq_poll_sample(tagged + row1);
            const float4 pa = cloud[ka], pb = cloud[kb];          // POLL implies LDS_CLOUD
            ax = pa.x; ay = pa.y; az = pa.z; bx = pb.x; by = pb.y; bz = pb.z;
            if (lane == 0) {
                float *o = new_xyz + row0 * 3;
                o[0] = ax; o[1] = ay; o[2] = az;
                if (two) { o[3] = bx; o[4] = by; o[5] = bz; }
            }
        } else {
            ax = xyz2[row0 * 3 + 0]; ay = xyz2[row0 * 3 + 1]; az = xyz2[row0 * 3 + 2];
            bx = xyz2[row1 * 3 + 0]; by = xyz2[row1 * 3 + 1]; bz = xyz2[row1 * 3 + 2];
        }
        // hit counters: wave-uniform values kept in VGPRs (see vbcnt_acc); bitmap words: lane c = chunk c
        int cnt0 = 0, cnt1 = two ? 0 : nsample;
        asm volatile("v_mov_b32 %0, %0" : "+v"(cnt0));
        asm volatile("v_mov_b32 %0, %0" : "+v"(cnt1));
        int done0 = 0, done1 = 0;                            // hits already flushed to the row buffers
        for (int wbase = 0; wbase < n; wbase += 4096) {      // one window = 64 chunks of 64 candidates
            unsigned m0lo = 0u, m0hi = 0u, m1lo = 0u, m1hi = 0u;
            const int wend = min(n, wbase + 4096);
            // 128 candidates per trip: two chunks x two queries = four independent distance chains per lane
            for (int base = wbase; base < wend; base += 128) {
                const int kA = base + lane, kB = base + 64 + lane;
                float pax, pay, paz, pbx, pby, pbz;
                if (LDS_CLOUD) {
                    const float4 pa = cloud[kA];
                    const float4 pb = cloud[kB];
                    pax = pa.x; pay = pa.y; paz = pa.z; pbx = pb.x; pby = pb.y; pbz = pb.z;
                } else {
                    const float inf = INFINITY;
                    const float *pa = data + (size_t)min(kA, n - 1) * 3;
                    const float *pb = data + (size_t)min(kB, n - 1) * 3;
                    pax = kA < n ? pa[0] : inf; pay = pa[1]; paz = pa[2];
                    pbx = kB < n ? pb[0] : inf; pby = pb[1]; pbz = pb[2];
                }
                // reference operand order: (x2-x1) with x2 the query (query_ball_point.cpp:26-32)
                const float sA0 = sqdist(ax, ay, az, pax, pay, paz);
                const float sB0 = sqdist(ax, ay, az, pbx, pby, pbz);
                const float sA1 = sqdist(bx, by, bz, pax, pay, paz);
                const float sB1 = sqdist(bx, by, bz, pbx, pby, pbz);
                const unsigned long long mA0 = __ballot(sA0 < thr), mB0 = __ballot(sB0 < thr);
                const unsigned long long mA1 = __ballot(sA1 < thr), mB1 = __ballot(sB1 < thr);
                const int cA = (base - wbase) >> 6;          // chunk number inside the window (scalar)
                bq_park(m0lo, m0hi, m1lo, m1hi, mA0, mA1, lane == cA);
                bq_park(m0lo, m0hi, m1lo, m1hi, mB0, mB1, lane == cA + 1);
                cnt0 = vbcnt_acc(mB0, vbcnt_acc(mA0, cnt0));
                cnt1 = vbcnt_acc(mB1, vbcnt_acc(mA1, cnt1));
                // stop as soon as both rows are full (reference: break at cnt == nsample, :23-24)
                if (__builtin_amdgcn_readfirstlane(min(cnt0, cnt1)) >= nsample) break;
            }
            if (done0 < nsample) done0 = bq_flush(m0lo, m0hi, wbase, done0, nsample, rowbuf0, lane);
            if (done1 < nsample && two) done1 = bq_flush(m1lo, m1hi, wbase, done1, nsample, rowbuf1, lane);
            if (__builtin_amdgcn_readfirstlane(min(cnt0, cnt1)) >= nsample) break;
        }
        cnt0 = __builtin_amdgcn_readfirstlane(min(cnt0, nsample));
        cnt1 = __builtin_amdgcn_readfirstlane(min(cnt1, nsample));
        // LDS ops of one wave execute in order; only the compiler must not reorder
        asm volatile("" ::: "memory");
        bq_emit<LDS_CLOUD, FUSE>(row0, nsample, cnt0, rowbuf0, cloud, data, ax, ay, az, idx, pts_cnt, grouped,
                                 subtract, lane);
        if (two)
            bq_emit<LDS_CLOUD, FUSE>(row1, nsample, cnt1, rowbuf1, cloud, data, bx, by, bz, idx, pts_cnt,
                                     grouped, subtract, lane);
        asm volatile("" ::: "memory");
    }
}


}  // namespace pn2
