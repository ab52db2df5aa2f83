// fps_body_r2_experiments.h -- NOT product code. The FPS round body of pointnet2_amd/csrc/fps_body.h with the
// round-2 experiments behind compile-time switches, kept so that profiles/r02/fps_experiments.txt can be
// reproduced (scripts/build_labs.sh + scripts/fps_prod_lab.hip). Measured on MI355X, ns per round at
// n = 4096, 256x16 / 512x8 (baseline 404 / 408):
//   PN2_FPS_BCAST_FULL   row_mask 0xf + bound_ctrl on the two row_bcast steps          402 / 393  ADOPTED
//   PN2_FPS_WAVE32       wave arg-max in 32-bit DPP ops (value all-reduce, masked low)  429 / 436  (385 without the
//                        hazard s_nops, but then wrong indices at 512 and 1024 threads: the hazards are real)
//   PN2_FPS_POLL         key exchange by tagged keys + polling instead of s_barrier     418 / 424
//   PN2_FPS_LATE_STORE   thread 0's index store moved under the winner read             409 / 405
//   PN2_FPS_DIAG         staggered update order with the lane arg-max folded in         406 / 407
//   PN2_FPS_PACK_512     packed update at 512 threads                                   -   / 423-467
//   (not in this file) wave key stored by all 64 lanes, lane 63 to the slot, the rest to a scratch strip (no exec-mask
//   juggling around the store)                                                         403 / 421   (267 vs 273 at n = 1024)
#pragma once
#include "../pointnet2_amd/csrc/pn2_device.h"

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
// After the six steps lane 63 holds the wave maximum (rows 1/3 after row_bcast:15, rows 2/3 after
// row_bcast:31). For the two broadcast steps the unwritten rows keep `old` = the lane's own value.
#ifndef PN2_FPS_BCAST_FULL
#define PN2_FPS_BCAST_FULL 0
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_f64_step(double v)
{
    const int hi = __double2hiint(v), lo = __double2loint(v);
    int ohi, olo;
    if (ROW_MASK == 0xf || PN2_FPS_BCAST_FULL) {
        // every lane has a valid source: the destination needs no initial value (saves two v_mov).
        // PN2_FPS_BCAST_FULL applies the same form to the two row_bcast steps: the rows without a source
        // (row 0 for row_bcast:15, rows 0-1 for row_bcast:31) read 0 -- or keep whatever the register held --
        // and a positive key max-combined with that is only ever looked at in rows that DO have a source:
        // lane 63's result depends on row 1 and row 3 after step five and on row 3 after step six.
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

// The same wave arg-max in 32-bit operations (PN2_FPS_WAVE32). A key is the pair (hi = value bits, lo = low
// word, larger lo = smaller rank). hi is a non-negative fp32, so its bit pattern orders like an unsigned
// integer and the lexicographic max splits into
//   1. an ALL-REDUCE of hi: four in-row butterfly steps, ONE v_max_u32 with a DPP operand each, then the
//      xor-16 and xor-32 exchanges with gfx950's v_permlane16/32_swap (copy, swap, max);
//   2. lo' = (hi == wave max) ? lo : 0 -- only the lanes that hold the maximum value stay in the race;
//   3. a plain max ladder of lo' that ends in lane 63 (four in-row steps + row_bcast:15 + row_bcast:31).
// 18 single-issue 32-bit instructions with ~10 dependent DPP hops, against six steps of
// (2 x v_mov_b32_dpp + v_max_f64) = 18 instructions whose fp64 op has twice the latency.
// A VGPR written by a VALU instruction may be read through DPP two wait states later at the earliest;
// inside inline asm the compiler's hazard recogniser does not see the instructions, hence the s_nop 1.
#ifndef PN2_FPS_WAVE32
#define PN2_FPS_WAVE32 0
#endif
#ifndef PN2_FPS_LATE_STORE
#define PN2_FPS_LATE_STORE 0
#endif
// PN2_FPS_POLL: the cross-wave key exchange WITHOUT s_barrier. Every wave key carries a freshness tag in
// its sign bit (the value is a non-negative fp32, so bit 63 is free): TAG = (round >> 1) & 1. The slot array
// is double-buffered by round parity, so what a slot still holds from round j - 2 has the opposite tag.
// Lane 63 stores its wave's key with one ds_write_b64 and every wave then re-reads the W slots until all W
// carry this round's tag: no s_waitcnt on the store, no barrier -- the wave that arrives last sees every
// key with its first read. LDS operations of a CU execute in issue order, and a wave cannot be two rounds
// ahead of another one (it needs that wave's key of the round in between), so two buffers suffice.
// The tournament ignores the tag with |abs| source modifiers.
#ifndef PN2_FPS_POLL
#define PN2_FPS_POLL 0
#endif
#ifndef PN2_FPS_DIAG
#define PN2_FPS_DIAG 0
#endif
#ifndef PN2_FPS_W32_NOP
#define PN2_FPS_W32_NOP 1
#endif
#if PN2_FPS_W32_NOP
#define PN2_NOP1 "s_nop 1\n\t"
#else
#define PN2_NOP1 ""
#endif
__device__ __forceinline__ void wave_max_key32_lane63(unsigned hi, unsigned lo, unsigned &whi, unsigned &wlo)
{
    unsigned m, t, l;
    asm volatile(
        PN2_NOP1
        "v_max_u32_dpp %[m], %[hi], %[hi] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[m], %[m], %[m] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[m], %[m], %[m] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[m], %[m], %[m] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32 %[t], %[m]\n\t"
        PN2_NOP1
        "v_permlane16_swap_b32 %[t], %[m]\n\t"
        PN2_NOP1
        "v_max_u32 %[m], %[m], %[t]\n\t"
        "v_mov_b32 %[t], %[m]\n\t"
        PN2_NOP1
        "v_permlane32_swap_b32 %[t], %[m]\n\t"
        PN2_NOP1
        "v_max_u32 %[m], %[m], %[t]\n\t"
        "v_cmp_eq_u32 vcc, %[m], %[hi]\n\t"
        "v_cndmask_b32 %[l], 0, %[lo], vcc\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[l], %[l], %[l] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[l], %[l], %[l] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[l], %[l], %[l] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[l], %[l], %[l] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[l], %[l], %[l] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        PN2_NOP1
        "v_max_u32_dpp %[l], %[l], %[l] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        : [m] "=&v"(m), [t] "=&v"(t), [l] "=&v"(l)
        : [hi] "v"(hi), [lo] "v"(lo)
        : "vcc");
    whi = m;
    wlo = l;
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
// perfectly: two slots per instruction, the selected point broadcast from one half of a register pair
// (op_sel), its negation folded into the add (neg_lo/neg_hi; a + (-s) == a - s bit for bit).
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

__device__ __forceinline__ pn2_f2 pk_sub_bcast_hi(pn2_f2 a, pn2_f2 s)   // a - s.y in both halves
{
    pn2_f2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(s));
    return r;
}

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
    if (PN2_FPS_POLL && t < 2 * W)                     // stale tags: buffer 1 is first used with tag 0, buffer 0 with tag 1
        partial[t] = t >= W ? 0x8000000000000000ull : 0ull;
    __syncthreads();

    pn2_f2 sxy = {0.f, 0.f}, szk = {0.f, 0.f};         // the selected point as two register pairs (packed path)
    float sx, sy, sz;                                  // the point selected last (starts at k = 0 = rank 0)
    if (LDSXYZ) {
        const float4 s = lds_rank[NS - 1];
        sx = s.x; sy = s.y; sz = s.z;
    } else {
        sx = src[0]; sy = src[1]; sz = src[2];
    }
    sxy.x = sx; sxy.y = sy; szk.x = sz;
    if (t == 0) {
        dst[0] = 0;                                    // tf_sampling_g.cu:114-116
        if (PUBLISH) __hip_atomic_store(gtag, (unsigned long long)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const unsigned low0 = (unsigned)(NS - 1 - t * P);   // key low word of this thread's slot 0: larger = smaller rank
    // PN2_FPS_LATE_STORE: thread 0's store of sample j sits between the winner read and the distance update
    // (exec-mask juggling + a branch on the critical path); with the switch on it is issued one round later,
    // right after the barrier, while the key reads are in flight.
    int kprev = 0;
    // one round; `par` (the partial buffer parity) is a literal at both call sites so the slot
    // addresses fold to constants (scalar address arithmetic costs 4-cycle issue slots)
    auto round = [&](const int j, const int par, const int tagbit = 0) __attribute__((always_inline)) {
        // Lane arg-max as ONE v_max_f64 per slot: the 64-bit pattern (value bits : low key word) of a
        // slot, read as a double, is positive, finite (value <= 1e38f < 0x7FF00000) and ordered exactly
        // like the pair (value, smaller rank first); fp64 denormals are never flushed on gfx9.
        double kd[P];
        if (PACKED && PN2_FPS_DIAG) {
            // Staggered ("diagonal") order: slot pair h runs one stage behind pair h - 1, and the lane arg-max
            // is folded in as a chain acc = max(acc, pair key) that consumes each pair's key as it appears. The
            // breadth-first order below finishes every slot at the same moment and then runs a four-level
            // fp64 tournament as a dependent tail; here the tail after the last v_min is two v_max_f64.
            pn2_f2 dx[PH], dy[PH], dz[PH];
            double pk[PH], acc = 0.0;
#pragma unroll
            for (int d = 0; d < 12 + PH - 1; ++d) {
#pragma unroll
                for (int h = 0; h < PH; ++h) {
                    const int st = d - h;
                    if (st == 0) dx[h] = pk_sub_bcast_lo(xx[h], sxy);
                    if (st == 1) dy[h] = pk_sub_bcast_hi(yy[h], sxy);
                    if (st == 2) dz[h] = pk_sub_bcast_lo(zz[h], szk);
                    if (st == 3) dx[h] = pk_mul(dx[h], dx[h]);
                    if (st == 4) dy[h] = pk_mul(dy[h], dy[h]);
                    if (st == 5) dz[h] = pk_mul(dz[h], dz[h]);
                    if (st == 6) dx[h] = pk_add(dx[h], dy[h]);
                    if (st == 7) dx[h] = pk_add(dx[h], dz[h]);
                    if (st == 8) asm volatile("v_min_f32 %0, %1, %0" : "+v"(md[2 * h]) : "v"(dx[h].x));
                    if (st == 9) asm volatile("v_min_f32 %0, %1, %0" : "+v"(md[2 * h + 1]) : "v"(dx[h].y));
                    if (st == 10) {
                        const double k0 = __hiloint2double(__float_as_int(md[2 * h]), (int)(low0 - (unsigned)(2 * h)));
                        const double k1 = __hiloint2double(__float_as_int(md[2 * h + 1]), (int)(low0 - (unsigned)(2 * h + 1)));
                        asm volatile("v_max_f64 %0, %1, %2" : "=v"(pk[h]) : "v"(k0), "v"(k1));
                    }
                    if (st == 11) {
                        if (h == 0) acc = pk[0];
                        else asm volatile("v_max_f64 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(pk[h]));
                    }
                }
            }
            kd[0] = acc;
        } else {
        if (PACKED) {
            // breadth first (the asm statements are volatile, so this IS the issue order): every
            // operation is >= PH instructions away from its producer
            pn2_f2 dx[PH], dy[PH], dz[PH];
#pragma unroll
            for (int h = 0; h < PH; ++h) dx[h] = pk_sub_bcast_lo(xx[h], sxy);
#pragma unroll
            for (int h = 0; h < PH; ++h) dy[h] = pk_sub_bcast_hi(yy[h], sxy);
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
        }
        const double bestd = kd[0];
        // whole-wave key max in VALU only; lane 63 ends up with it and publishes it
        unsigned long long *slot = partial + par * W;
        constexpr bool POLL = PN2_FPS_POLL && (W == 4 || W == 8);
        if (PN2_FPS_WAVE32) {
            unsigned whi, wlo;
            wave_max_key32_lane63((unsigned)__double2hiint(bestd), (unsigned)__double2loint(bestd), whi, wlo);
            if (POLL && tagbit) whi |= 0x80000000u;
            if (lane == 63) slot[w] = ((unsigned long long)whi << 32) | wlo;
        } else {
            const double wd = wave_max_f64_lane63(bestd);
            if (lane == 63) {
                if (POLL && tagbit) slot[w] = (unsigned long long)__double_as_longlong(wd) | 0x8000000000000000ull;
                else reinterpret_cast<double *>(slot)[w] = wd;
            }
        }
        if (!POLL) __syncthreads();
        // block arg-max: v_max_f64 tournament over the W keys, every wave redundantly (wave-uniform data)
        const double *dslot = reinterpret_cast<const double *>(slot);
        unsigned win;                                  // low word of the winning key = mirror index
        if (POLL) {
            typedef unsigned pn2_u4 __attribute__((ext_vector_type(4)));
            pn2_u4 q[W / 2];
            const unsigned saddr = (unsigned)(par * W * 8);                    // LDS byte address of slot[par][0]
            for (;;) {
                if (W == 4) {
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q[0]), "=&v"(q[1]) : "v"(saddr) : "memory");
                } else {
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
                                 "ds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[W / 2 - 2]), "=&v"(q[W / 2 - 1]) : "v"(saddr) : "memory");
                }
                unsigned all_or = 0u, all_and = 0xffffffffu;                      // over the hi words (tag = bit 31)
#pragma unroll
                for (int i = 0; i < W / 2; ++i) {
                    all_or |= q[i].y | q[i].w;
                    all_and &= q[i].y & q[i].w;
                }
                if (tagbit ? ((int)all_and < 0) : ((int)all_or >= 0)) break;      // every key carries this round's tag
            }
            double key[W];
#pragma unroll
            for (int i = 0; i < W / 2; ++i) {
                key[2 * i] = __hiloint2double((int)q[i].y, (int)q[i].x);
                key[2 * i + 1] = __hiloint2double((int)q[i].w, (int)q[i].z);
            }
#pragma unroll
            for (int st = 1; st < W; st <<= 1)
#pragma unroll
                for (int i = 0; i + st < W; i += 2 * st)
                    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
            win = (unsigned)__double2loint(key[0]);
        } else if (W >= 16) {
        // lane i reads key i mod W (ONE LDS read per wave instead of W/2 broadcast reads) and the W
        // keys are combined across lanes with log2(W) butterfly DPP steps: every lane ends with the max
        double kq = dslot[lane & (W - 1)];
        if (W >= 2) kq = dpp_max_f64_step<0xB1, 0xf>(kq);    // lane ^ 1
        if (W >= 4) kq = dpp_max_f64_step<0x4E, 0xf>(kq);    // lane ^ 2
        if (W >= 8) kq = dpp_max_f64_step<0x141, 0xf>(kq);   // other quad of the half row
        if (W >= 16) kq = dpp_max_f64_step<0x140, 0xf>(kq);  // other half row
        win = (unsigned)__double2loint(kq);
        } else {
        // W <= 8: broadcast-read all keys, tournament on wave-uniform data (measured faster: 408 vs 441 ns)
        double key[W];
#pragma unroll
        for (int i = 0; i < W; ++i) key[i] = dslot[i];
#pragma unroll
        for (int st = 1; st < W; st <<= 1)
#pragma unroll
            for (int i = 0; i + st < W; i += 2 * st)
                asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
        win = (unsigned)__double2loint(key[0]);
        }
        int k;
        if (LDSXYZ) {
            typedef float pn2_f4 __attribute__((ext_vector_type(4)));
            pn2_f4 s;                                  // same address in every lane: LDS broadcast
            if (PN2_FPS_LATE_STORE && !PUBLISH) {
                // issue the read, do thread 0's store of the PREVIOUS sample under its latency, then wait
                asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(s) : "v"(win << 4) : "memory");
                if (t == 0) dst[j - 1] = kprev;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s) : : "memory");
            } else {
                const float4 q = lds_rank[win];
                s.x = q.x; s.y = q.y; s.z = q.z; s.w = q.w;
            }
            sx = s.x; sy = s.y; sz = s.z;
            sxy.x = s.x; sxy.y = s.y; szk.x = s.z; szk.y = s.w;
            k = __float_as_int(s.w);
        } else {
            k = lds_k[win];
            if (PN2_FPS_LATE_STORE && !PUBLISH && t == 0) dst[j - 1] = kprev;
            sx = src[(size_t)k * 3 + 0]; sy = src[(size_t)k * 3 + 1]; sz = src[(size_t)k * 3 + 2];
            sxy.x = sx; sxy.y = sy; szk.x = sz;
        }
        if (PN2_FPS_LATE_STORE && !PUBLISH) {
            kprev = k;                                 // stored by the NEXT round (or after the loop), see above
        } else if (t == 0) {
            dst[j] = k;
            if (PUBLISH)
                __hip_atomic_store(gtag + j, ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)k, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    int j = 1;
    if (PN2_FPS_POLL) {
        // (parity, tag) = (j & 1, (j >> 1) & 1) as literals: four round bodies
        for (; j + 3 < m; j += 4) {
            round(j, 1, 0);
            round(j + 1, 0, 1);
            round(j + 2, 1, 1);
            round(j + 3, 0, 0);
        }
        if (j < m) { round(j, 1, 0); ++j; }
        if (j < m) { round(j, 0, 1); ++j; }
        if (j < m) { round(j, 1, 1); ++j; }
    } else {
        for (; j + 1 < m; j += 2) {
            round(j, 1);
            round(j + 1, 0);
        }
        if (j < m) { round(j, 1); ++j; }
    }
    if (PN2_FPS_LATE_STORE && !PUBLISH && t == 0 && m > 1) dst[m - 1] = kprev;
    fps_gather_epilogue<T>(m, src, dst, dxyz);
}


}  // namespace pn2
