// fps_batch_body.h -- farthest point sampling with SEVERAL samples per arg-max exchange and a wave of its own for the
// arg-max chain (round 6).
//
// Same results as fps_pruned_body / fps_reg_body / the reference kernel (tf_sampling_g.cu:105-170), bit for bit. What changes is
// the shape of the chain. In every other tier a sample costs one workgroup-wide arg-max: a six-step wave ladder, an LDS write, a
// barrier, an LDS read, a tournament, a mirror read -- ~700 dependent cycles of which the distance update is ~120. Here the
// workgroup exchanges a short LIST of candidates once per BATCH; one more wave, the PICKER, then works through the list on its
// own -- one wave ladder and nine vector instructions per sample, no barrier, no LDS trip on its path -- for as many samples as
// the list provably determines (~20 on the bench clouds once the chain is past its first quarter), while the UPDATER waves
// apply the samples to the slots as they appear. Measured (profiles/r06/fps_batch.txt): 4096 -> 1024 220 us against the pruned
// tier's 366, 8192 -> 1024 247 against 416.
//
// Why a list determines several samples. Let td[] be the running distances after j samples, ordered by the reference's key
// (value, then smaller tie rank). Fix a threshold theta below the current maximum. A point whose value is below theta can never
// become a sample while the sample values stay >= theta, because running distances only fall. So as long as the next sample's
// value is >= theta, the next sample is the best of the points that are >= theta NOW, after lowering their values by the samples
// taken since -- and those are few: the points near the covering radius. The batch ends when the best listed candidate falls
// below the bound; everything taken until then is exactly the reference's sequence (scripts/fps_batch_sim.py checks the rule
// against the oracle; tests/test_fps_batch_model.py is the CPU model of this file's arithmetic).
//
// Organisation. 576 threads: waves 0..7 are the updaters -- they hold the slots as the pruned tier deals them (fps_pruned_prologue
// on 512 threads: 32 spatial groups, four per wave, rank-ordered LDS mirror, group boxes) --, wave 8 is the picker. EIGHT updater
// waves, two per SIMD, because a lone wave issues one instruction per ~5-8 cycles whatever its kind (vector, scalar, branch) and
// the apply loop is half scalar work: two waves share a SIMD's vector and scalar issue (four updater waves of twice the slots:
// +14 % on the whole chain).
//   * EARLY. The first PN2_BT_EARLY samples are taken one per exchange (the full tier's round on the updaters; the picker sits
//     them out at the barriers): while the farthest-point distance still halves every few samples a list yields three or four.
//   * COLLECT (updaters, once per batch). Every lane computes the best key and the second-best VALUE of its P slots. Lanes whose
//     best value is >= theta are candidate lanes: at most 64 / W per wave -- a wave with more raises its own threshold by
//     bisection on the value bits, a wave with none (or with too many equal values) falls back to its exact best lane (one
//     64-bit wave ladder). A candidate lane writes its best key to the wave's part of the list; the wave's BOUND -- the smallest
//     value bits a sample must have for the list to be complete -- is the maximum of its threshold and of (second-best value +
//     1 ulp) of its candidate lanes (an LDS atomic maximum on the fp32 bit patterns: the values are >= 0; issued as a plain
//     ds_max_u32 -- hipcc turns atomicMax into a readlane loop over the active lanes). One barrier, the only one of the batch.
//   * PICK (picker). Lane i takes candidate i (key from the list, x, y, z, k from the mirror). Per sample, hand-scheduled (the
//     asm block below): the winner lane alone (exec = one lane) stores its (x, y, z, k) row and then the new count to LDS and
//     leaves the contest; three v_readlane of its coordinates; the reference's distance (tf_sampling_g.cu:141-144) from the
//     sample to the other candidates, nine vector instructions; and, interleaved with those, the 32-bit wave ladder
//     (v_max_i32 with the DPP operand folded in) over the values as they were BEFORE the update -- values only fall, so a lane
//     that still holds that maximum afterwards is the arg-max if it is the only one (99 % of the samples); otherwise the exact
//     arg-max (value ladder, 64-bit keys among equal values, the pruned tier's ladder). A sample is accepted while its value
//     bits are >= the bound; the first sample of a batch always is: the list holds every updater wave's best lane, so its
//     maximum is the global one.
//   * APPLY (updaters, behind the picker). A wave polls the count, takes up to 64 / GW new samples at a time -- lane l tests
//     sample l / GW against the box of the wave's group l % GW: one distance-to-box computation for 64 (sample, group) pairs --
//     and updates the touched groups, group by group (packed fp32, as in the pruned tier; no key work: keys are only needed at
//     COLLECT). The skip test uses the value of the previous batch's LAST sample as v* (no running distance is above it).
//   * theta = (1 - g) * (value of the last sample); g adapts so that the list stays about half full. The picker decides
//     and publishes theta with the end-of-batch flag.
//
// Once a sample's value is 0 every running distance is 0 and the reference keeps selecting point 0 (its tie rule): the rest of
// the output is filled directly.
//
// Exactness of the acceptance rule in fp32 bit patterns: values are non-negative floats, so integer comparison of the bits is
// the value order; "value > second-best" is "bits >= second-best bits + 1". A sample with value bits >= the maximum of the waves'
// bounds beats (by value alone, no tie rule needed) every point that is not a listed candidate's best point; among the listed
// ones the 64-bit keys decide, ties included. Degenerate clouds with many EQUAL values at the top (lattices) make short batches
// (the bound is strict), never different samples.
//
// Hand-offs inside the workgroup: LDS only. The picker's row store and count store come from the same lane (LDS executes a
// wave's operations in order), counts are release stores / acquire loads at workgroup scope. Double-buffered by batch parity:
// list, counts, bounds and the header; the sample ring is single (a batch's samples are consumed before the next barrier).
// Chains of different clouds now differ in length (the lists are the data's): 248-261 us over the 32 clouds of a bench batch.
#pragma once
#include "fps_pruned_body.h"

namespace pn2 {

constexpr int kBtCand = PN2_WAVE;              // 64 candidates = the picker's lanes: 64 / W candidate lanes per updater wave
constexpr int kBtMaxW = 8;                     // updater waves: 4 (one per SIMD) or 8 (two per SIMD)
#ifndef PN2_BT_UT
#define PN2_BT_UT 512                          // updater threads of the product
#endif
constexpr int kBtUT = PN2_BT_UT;
constexpr int kBtT = kBtUT + PN2_WAVE;         // + the picker
constexpr unsigned kBtEnd = 0x100u, kBtFill = 0x200u, kBtCountMask = 0xffu;

struct BtXchg {
    double list[kBtCand];                      // wave w's candidates at [w * cap, w * cap + cnt[w]), cap = 64 / W
    unsigned cnt[kBtMaxW];
    unsigned bound[kBtMaxW];
    unsigned count;                            // samples of this batch published so far | kBtEnd | kBtFill
    unsigned single;                           // after this batch: that many samples one per exchange (SLOW BATCHES below)
    unsigned theta, vlast;                     // for the NEXT collect: threshold bits, value bits of the batch's last sample
    unsigned pad[4];
};
static_assert(sizeof(BtXchg) % 16 == 0, "16-byte rows");

// P = slots per updater thread, UT = updater threads
__host__ __device__ constexpr size_t fps_batch_xchg_offset(int P, int UT = kBtUT) { return (fps_pruned_lds_bytes(P, UT) + 15) & ~(size_t)15; }
__host__ __device__ constexpr size_t fps_batch_lds_bytes(int P, int UT = kBtUT) { return fps_batch_xchg_offset(P, UT) + 2 * sizeof(BtXchg) + 16 * kBtCand; }

__device__ __forceinline__ float vmax_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmed3_f32(float a, float b, float c)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// wave-wide signed maximum, result in lane 63: the DPP operand rides on the v_max itself (VOP2), one instruction per step.
// Lanes without a source (bound_ctrl:0 reads 0) combine with 0 -- harmless for a maximum of values of which at least one is >= 0.
__device__ __forceinline__ int bt_wave_max_i32_lane63(int v)
{
    // written with the builtin so that the compiler folds the v_mov_dpp into the v_max (GCNDPPCombine), keeps track of the DPP
    // hazards itself and schedules independent work into the wait states
#define PN2_BT_STEP(ctrl) v = max(v, __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true))
    PN2_BT_STEP(0xB1);    // quad_perm:[1,0,3,2]
    PN2_BT_STEP(0x4E);    // quad_perm:[2,3,0,1]
    PN2_BT_STEP(0x141);   // row_half_mirror
    PN2_BT_STEP(0x140);   // row_mirror
    PN2_BT_STEP(0x142);   // row_bcast:15
    PN2_BT_STEP(0x143);   // row_bcast:31
#undef PN2_BT_STEP
    return v;
}

// LDS atomic maximum without the compiler's wave-level pre-reduction (a readlane loop over the active lanes: ~50 cycles per
// lane on a lone wave; the LDS unit serialises same-address atomics in a few cycles each)
__device__ __forceinline__ void lds_max_u32(unsigned *p, unsigned v)
{
    asm volatile("ds_max_u32 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
}

// Where the batched tier exists and pays (measured, profiles/r06/fps_batch.txt): 513..8192 rank slots (1024 / 2048: 8 / 16 groups,
// 4096 / 8192: 32). Its grouping prologue costs 2-10 us more than the full tier's staging, its first PN2_BT_EARLY samples cost the
// full tier's round, from ~200 samples on a sample is 160-175 ns against 250-390. npoint = 256: 61 us against 67 (full) at 1024
// rank slots, 68 / 81 at 2048, 84 / 102 at 4096, 109 / 134 (pruned) at 8192; npoint = 128: no gain anywhere.
inline bool fps_batch_covers(int ranks) { return ranks > 512 && ranks <= 8192; }
#ifdef PN2_BT_LAB_NEVER_AUTO      // A/B library: the size rule never chooses this tier (scripts/: what did the tier change in a model?)
inline bool fps_batch_pays(int, int) { return false; }
#else
inline bool fps_batch_pays(int ranks, int m) { return fps_batch_covers(ranks) && m >= 256; }
#endif

#ifdef PN2_BT_STATS
// lab: [0] batches, [1] samples, [2] exact fallbacks, [3] bisection steps, [4] sum of list sizes, [5] (group, sample) updates,
// [6] picker cycles in READ, [7] picker cycles in PICK, [8] picker cycles waiting at the barrier, [9] updater 0 cycles in COLLECT,
// [10] updater 0 cycles from the barrier to the end flag, [11] tie resolutions, [12] speculation misses, [13] samples taken one per
// exchange after slow batches, [14] such runs
__device__ unsigned long long g_bt_stats[16];
#ifndef PN2_BT_STATS_FROM
#define PN2_BT_STATS_FROM 0
#endif
#define PN2_BT_STAT(i, v) do { if (lane == 0 && cloud == 0 && j >= PN2_BT_STATS_FROM) atomicAdd(&g_bt_stats[i], (unsigned long long)(v)); } while (0)
#define PN2_BT_CLOCK() __builtin_readcyclecounter()
#else
#define PN2_BT_STAT(i, v) do { } while (0)
#define PN2_BT_CLOCK() 0ll
#pragma clang diagnostic ignored "-Wunused-variable"
#endif

#ifndef PN2_BT_LIST_HI
#define PN2_BT_LIST_HI 36      // measured: 44 / 28 +1.7 %, 54 / 40 +6 %, 28 / 14 +2 %, 20 / 10 +7 % (profiles/r06/fps_batch.txt)
#define PN2_BT_LIST_LO 20
#endif
#ifndef PN2_BT_EARLY
#define PN2_BT_EARLY 48            // the first samples of a cloud are taken one per exchange (below; 32 / 64 / 96 measured: 255.7 / 255.5 / 261.9 us, none: 280.4)
#endif
#ifndef PN2_BT_ASM_LOOP
#define PN2_BT_ASM_LOOP 1
#endif
#ifndef PN2_BT_G0
#define PN2_BT_G0 0.10f               // initial 1 - theta / (last sample value)
#endif
// SLOW BATCHES. A batch costs a COLLECT + a barrier + a list read whatever it yields (~0.6 us), and a sample whose arg-max needs
// the 64-bit keys costs three times the speculative one; the classic round is 0.39 us at 4096 rank slots. Clouds with many EQUAL
// values at the top (lattices, coordinates quantised to a coarse grid, duplicated points) end most batches after a sample or two
// -- the bound is strict -- or resolve ties at every sample, and the tier then costs 1.2-2 x the full tier's chain
// (profiles/r06/fps_tier_by_cloud.txt: 493 / 694 / 778 us against 395 at 4096 -> 1024). The picker therefore CLOCKS both forms:
// the EARLY rounds give the cost of a one-per-exchange round on this cloud and this device, the interval between two list
// barriers the cost of a batch. When the batches of a run (decayed sums) cost more per sample than the round -- 1.5 x the round
// while fewer than PN2_BT_SLOW_MIN batches have been clocked --, the workgroup takes the next R samples one per exchange (the
// EARLY round), then tries batches again; R doubles (16 .. 512) while the batches stay slow and starts over after eight batches
// in a row that pay. Same samples either way -- both
// forms are the reference's arg-max; only the schedule depends on the clock.
#ifndef PN2_BT_PACKED_FROM
#define PN2_BT_PACKED_FROM 16       // the one-per-exchange round on register PAIRS from this many slots per thread (all sizes packed: 1024 -> 1024 182 -> 190 us)
#endif
#ifndef PN2_BT_SLOW_MIN
#define PN2_BT_SLOW_MIN 3
#define PN2_BT_QUICK 1
#define PN2_BT_SINGLE0 16
#define PN2_BT_SINGLE1 512
#endif

template <int P, int GS, bool PUBLISH, int UT = kBtUT>
__device__ __forceinline__ void fps_batch_body(int n, int m, int Q, int cloud, const float *__restrict__ xyz,
                                               int *__restrict__ out, float *__restrict__ out_xyz,
                                               unsigned long long *__restrict__ tagged, char *smem, unsigned tag = 1u)
{
    constexpr int W = UT / PN2_WAVE, NS = UT * P;      // updater waves, rank slots
    constexpr int GW = P / GS;                         // groups per updater wave
    constexpr int CAP = kBtCand / W;                   // candidate lanes per updater wave
    constexpr int BT = UT + PN2_WAVE;                  // threads of the workgroup
    static_assert((W * GW == 32 && (W == 4 || W == 8)) || (W == 8 && (GW == 2 || GW == 1)),
                  "32 groups on four or eight updater waves; 16 / 8 groups (2048 / 1024 rank slots) on eight");
    float4 *lds_rank = reinterpret_cast<float4 *>(smem + 256);
    BtXchg *xch = reinterpret_cast<BtXchg *>(smem + fps_batch_xchg_offset(P, UT));
    float4 *ring = reinterpret_cast<float4 *>(smem + fps_batch_xchg_offset(P, UT) + 2 * sizeof(BtXchg));   // [kBtCand] samples of the batch: x, y, z, k

    const float *__restrict__ src = xyz + (size_t)cloud * n * 3;
    int *__restrict__ dst = out + (size_t)cloud * m;
    float *__restrict__ dxyz = out_xyz ? out_xyz + (size_t)cloud * m * 3 : nullptr;
    pn2_gu64 *gtag = PUBLISH ? (pn2_gu64 *)(tagged + (size_t)cloud * m) : nullptr;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);

    int j = 1;                                   // samples written so far (wave-uniform, the same in every wave)
    const int jE = min(PN2_BT_EARLY, m);         // samples 1 .. jE - 1 are taken one per exchange
    int fill_k = -1;                             // >= 0: the cloud ran out of distinct points at sample j, the rest is this index

    if (w == W) {
        // ================================================ the picker ===========================================================
#ifndef PN2_NO_SETPRIO
        __builtin_amdgcn_s_setprio(3);           // it shares SIMD 0 with updater wave 0 and is the chain
#endif
        for (int i = lane; i < (int)(2 * sizeof(BtXchg) / 4); i += PN2_WAVE) reinterpret_cast<unsigned *>(xch)[i] = 0u;   // counts, bounds, flags
        for (int i = 0; i < kPrPrologueBarriers; ++i) __syncthreads();
        long long e0 = 0;
        for (; j < jE; ++j) {                    // the updaters' early rounds: one barrier each
            if (j == 8) e0 = (long long)__builtin_readcyclecounter();
            __syncthreads();
        }
        // SLOW BATCHES: cycles of a one-per-exchange round (none measured: batches always pay), decayed cycles / samples of the
        // batches since the last such run, batches counted, length of the next run, clock at the previous idle point
        const float round_cyc = jE >= 24 ? (float)((long long)__builtin_readcyclecounter() - e0) / (float)(jE - 8) : 1e30f;
        // The bookkeeping runs in the picker's idle time ahead of a list barrier (the updaters are collecting), not between the last
        // pick and the end flag: a decision is published with the end flag of the batch that follows it.
        float bt_cyc = 0.f, bt_smp = 0.f;
        int bt_n = 0, bt_paid = 0, run = PN2_BT_SINGLE0, a_prev = 0, single_next = 0;
        long long qb = 0;
        float g = PN2_BT_G0;
        int par = 0;
        if (j < m)
        for (;;) {
            const long long q0 = PN2_BT_CLOCK();
            // SLOW BATCHES, in the idle time (the clock's latency too: s_memtime shares the LDS counter, behind the barrier it sat
            // on the list read): the batch that just ended into the decayed sums, then the decision
            {
                const long long qn = (long long)__builtin_readcyclecounter();
                if (bt_n > 0) {
                    bt_cyc = 0.75f * bt_cyc + (float)(qn - qb); bt_smp = 0.75f * bt_smp + (float)a_prev;
                    if (bt_n > PN2_BT_QUICK) {       // at least two whole batches of this run have been clocked
                        if (bt_cyc > bt_smp * round_cyc * (bt_n > PN2_BT_SLOW_MIN ? 1.0f : 1.5f)) {
                            single_next = run;
                            run = min(2 * run, PN2_BT_SINGLE1);
                            bt_n = -1; bt_paid = 0; bt_cyc = 0.f; bt_smp = 0.f;   // the next interval holds the run: not folded
                        } else if (++bt_paid >= 8) {
                            run = PN2_BT_SINGLE0;
                        }
                    }
                }
                qb = qn;
                ++bt_n;
            }
            __syncthreads();                     // the list of this batch is complete
            const long long q1 = PN2_BT_CLOCK();
            BtXchg &X = xch[par];
            if (lane == 0) __hip_atomic_store(&xch[par ^ 1].count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // everybody is past the batch that used it
            const uint4 c4 = *reinterpret_cast<const uint4 *>(X.cnt), c5 = *reinterpret_cast<const uint4 *>(X.cnt + 4);      // waves beyond W: 0
            const uint4 b4 = *reinterpret_cast<const uint4 *>(X.bound), b5 = *reinterpret_cast<const uint4 *>(X.bound + 4);
            const double key = X.list[lane];
            const unsigned cw = X.cnt[lane / CAP];
            const bool valid = (unsigned)(lane & (CAP - 1)) < cw;
            const unsigned lowc = valid ? (unsigned)__double2loint(key) : 0u;
            typedef float bt_f4 __attribute__((ext_vector_type(4)));
            const bt_f4 cand = *reinterpret_cast<const bt_f4 *>(&lds_rank[lowc]);   // x, y, z, bits of k
            const int boundb = __builtin_amdgcn_readfirstlane((int)max(max(max(b4.x, b4.y), max(b4.z, b4.w)), max(max(b5.x, b5.y), max(b5.z, b5.w))));
            const int total = __builtin_amdgcn_readfirstlane((int)(c4.x + c4.y + c4.z + c4.w + c5.x + c5.y + c5.z + c5.w));
            int cval = valid ? __double2hiint(key) : (int)0xBF800000;             // -1.0f: below every value as an integer, kept by v_min_f32
            const int clo = (int)lowc;
            int a = 0;
            int vlastb = 0;
            bool fill = false;
            const unsigned ring_base = (unsigned)(size_t)ring, count_addr = (unsigned)(size_t)&X.count;   // LDS byte addresses
            const long long q2 = PN2_BT_CLOCK();
            int bh;
            unsigned long long eq;                                               // one bit: the winner's lane
            // the exact arg-max of the current values: value ladder, then the 64-bit keys among equal values
            auto argmax = [&]() __attribute__((always_inline)) {
                bh = __builtin_amdgcn_readlane(bt_wave_max_i32_lane63(cval), 63);
                eq = __ballot(cval == bh);
                if (__popcll(eq) != 1) {
                    PN2_BT_STAT(11, 1);
                    const double kq = cval == bh ? __hiloint2double(cval, clo) : -1.0;
                    const double km = wave_max_f64_lane63(kq);
                    const int ml = __builtin_amdgcn_readlane(__double2loint(km), 63);
                    eq = __ballot(cval == bh && clo == ml);
                    eq &= 0ull - eq;                                             // equal keys exist only among padding slots
                }
            };
            argmax();                                                            // the batch's first sample: always (the global arg-max)
            const int amax = min(m - j, kBtCand);
            unsigned raddr = ring_base, na = 1u;
#if PN2_BT_ASM_LOOP
            // The sample loop, hand-scheduled (the compiler's version of the same loop -- #else below -- takes three taken
            // branches and six more scalar moves per sample; on a lone wave every instruction is an issue slot of ~8 cycles):
            //   publish: the winner lane alone (exec = eq) stores its row and then the new count -- two LDS writes of one wave
            //     execute in order, so a reader that sees the count sees the row; no wait in between, none behind -- and leaves
            //     the contest (-1.0f);
            //   SPECULATION: the maximum of the values as they are BEFORE this sample's update is computed in the shadow of the
            //     update (the DPP steps need two independent instructions between them anyway). Values only fall: a lane that
            //     still holds that maximum afterwards is the arg-max -- if it is the only one. A sample of value 0 ends the batch by
            //     itself: nothing is above the bound afterwards.
            // reason: 0 = the batch is full / the cloud is done, 1 = nothing above the bound is left, 2 = the speculative maximum
            // is gone or not unique (the exact arg-max below decides, then the loop resumes).
            const float cx = cand.x, cy = cand.y, cz = cand.z;
            for (;;) {
                int reason, t_wl, t_sx, t_sy, t_sz, t_ms, t_cnt;
                int v_t;
                float v_dx, v_dy, v_dz;
                asm volatile(
                    "s_branch 1f\n\t"
                    "0:\n\t"
                    "s_mov_b32 %[bh], %[ms]\n\t"                  // the speculative maximum stood: it is the next sample's value
                    "1:\n\t"
                    "s_mov_b64 exec, %[eq]\n\t"
                    "ds_write_b128 %[raddr], %[cand]\n\t"
                    "ds_write_b32 %[caddr], %[na]\n\t"
                    "v_mov_b32 %[cval], 0xbf800000\n\t"
                    "s_mov_b64 exec, -1\n\t"
                    "s_ff1_i32_b64 %[wl], %[eq]\n\t"
                    "s_add_i32 %[a], %[a], 1\n\t"
                    "v_add_u32 %[raddr], 16, %[raddr]\n\t"
                    "v_add_u32 %[na], 1, %[na]\n\t"
                    "s_cmp_ge_i32 %[a], %[amax]\n\t"
                    "s_cbranch_scc1 2f\n\t"
                    "v_readlane_b32 %[sx], %[cx], %[wl]\n\t"
                    "v_readlane_b32 %[sy], %[cy], %[wl]\n\t"
                    "v_max_i32_dpp %[t], %[cval], %[cval] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_readlane_b32 %[sz], %[cz], %[wl]\n\t"
                    "v_subrev_f32 %[dx], %[sx], %[cx]\n\t"
                    "v_max_i32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_subrev_f32 %[dy], %[sy], %[cy]\n\t"
                    "v_subrev_f32 %[dz], %[sz], %[cz]\n\t"
                    "v_max_i32_dpp %[t], %[t], %[t] row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_mul_f32 %[dx], %[dx], %[dx]\n\t"
                    "v_mul_f32 %[dy], %[dy], %[dy]\n\t"
                    "v_max_i32_dpp %[t], %[t], %[t] row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_mul_f32 %[dz], %[dz], %[dz]\n\t"
                    "v_add_f32 %[dx], %[dx], %[dy]\n\t"
                    "v_max_i32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_add_f32 %[dx], %[dx], %[dz]\n\t"
                    "s_nop 0\n\t"
                    "v_max_i32_dpp %[t], %[t], %[t] row_bcast:31 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                    "v_min_f32 %[cval], %[dx], %[cval]\n\t"
                    "s_nop 0\n\t"
                    "v_readlane_b32 %[ms], %[t], 63\n\t"
                    "s_nop 0\n\t"
                    "s_cmp_lt_i32 %[ms], %[bound]\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "v_cmp_eq_u32_e64 %[eq], %[ms], %[cval]\n\t"
                    "s_bcnt1_i32_b64 %[cnt], %[eq]\n\t"
                    "s_cmp_eq_u32 %[cnt], 1\n\t"
                    "s_cbranch_scc1 0b\n\t"
                    "s_mov_b32 %[reason], 2\n\t"
                    "s_branch 4f\n\t"
                    "2:\n\t"
                    "s_mov_b32 %[reason], 0\n\t"
                    "s_branch 4f\n\t"
                    "3:\n\t"
                    "s_mov_b32 %[reason], 1\n\t"
                    "4:"
                    : [cval] "+v"(cval), [raddr] "+v"(raddr), [na] "+v"(na), [eq] "+s"(eq), [a] "+s"(a), [bh] "+s"(bh),
                      [reason] "=&s"(reason), [wl] "=&s"(t_wl), [sx] "=&s"(t_sx), [sy] "=&s"(t_sy), [sz] "=&s"(t_sz), [ms] "=&s"(t_ms), [cnt] "=&s"(t_cnt),
                      [t] "=&v"(v_t), [dx] "=&v"(v_dx), [dy] "=&v"(v_dy), [dz] "=&v"(v_dz)
                    : [cand] "v"(cand), [caddr] "v"(count_addr), [cx] "v"(cx), [cy] "v"(cy), [cz] "v"(cz), [amax] "s"(amax), [bound] "s"(boundb)
                    : "memory", "scc");
                vlastb = bh;                                                     // the value of the sample published last
                if (reason != 2) break;
                PN2_BT_STAT(12, 1);
                argmax();
                if (bh < boundb) break;
            }
#else
            for (;;) {
                // the winner lane alone stores its row and then the new count: two LDS writes of one wave execute in order, so a
                // reader that sees the count sees the row (no wait in between, none behind). It also leaves the contest (-1.0f).
                asm volatile("s_mov_b64 exec, %1\n\t"
                             "ds_write_b128 %2, %3\n\t"
                             "ds_write_b32 %4, %5\n\t"
                             "v_mov_b32 %0, 0xbf800000\n\t"
                             "s_mov_b64 exec, -1"
                             : "+v"(cval) : "s"(eq), "v"(raddr), "v"(cand), "v"(count_addr), "v"(na) : "memory");
                const int wl = (int)__builtin_ctzll(eq);
                raddr += 16u; na += 1u;
                a = __builtin_amdgcn_readfirstlane(a + 1);                       // scalar loop control
                vlastb = bh;
                if (a >= amax) break;
                // SPECULATION: the maximum of the values as they are BEFORE this sample's update is computed in the shadow of the
                // update. Values only fall: a lane that still holds that maximum afterwards is the arg-max -- if it is the only one.
                // (A sample of value 0 ends the batch by itself: nothing is above the bound afterwards.)
                const int ms = __builtin_amdgcn_readlane(bt_wave_max_i32_lane63(cval), 63);
                const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand.x), wl));
                const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand.y), wl));
                const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand.z), wl));
                const float d = sqdist(cand.x, cand.y, cand.z, sx, sy, sz);      // tf_sampling_g.cu:141-143
                cval = __float_as_int(vmin_f32(d, __int_as_float(cval)));        // :144
                if (ms < boundb) break;                                          // nothing above the bound is left, whatever the update did
                eq = __ballot(cval == ms);
                bh = ms;
                if (__popcll(eq) != 1) {
                    PN2_BT_STAT(12, 1);
                    argmax();
                    if (bh < boundb) break;
                }
            }
#endif
            if (vlastb == 0) { fill = true; fill_k = __builtin_amdgcn_readlane(__float_as_int(cand.w), (int)__builtin_ctzll(eq)); }   // every running distance is 0 from here on
            const long long q3 = PN2_BT_CLOCK();
            // the list about half full
            if (total > PN2_BT_LIST_HI) g = fmaxf(g * 0.8f, 1.0f / 128.0f);
            else if (total < PN2_BT_LIST_LO) g = fminf(g * 1.25f, 0.5f);
            X.theta = __float_as_uint(__fmul_rn(__int_as_float(vlastb), 1.0f - g));
            X.vlast = (unsigned)vlastb;
            a_prev = a;
            int single = min(single_next, m - (j + a));
            if (single_next) { PN2_BT_STAT(13, single); PN2_BT_STAT(14, 1); }
            single_next = 0;
            single = __builtin_amdgcn_readfirstlane(single);
            X.single = (unsigned)single;
            if (lane == 0)
                __hip_atomic_store(&X.count, (unsigned)a | kBtEnd | (fill ? kBtFill : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            PN2_BT_STAT(0, 1); PN2_BT_STAT(1, a); PN2_BT_STAT(4, total); PN2_BT_STAT(6, q2 - q1); PN2_BT_STAT(7, q3 - q2); PN2_BT_STAT(8, q1 - q0);
            j = __builtin_amdgcn_readfirstlane(j + a);
            if (fill || j >= m) break;
            for (int i = 0; i < single; ++i) __syncthreads();     // the updaters' one-per-exchange rounds: one barrier each
            j += single;
            if (j >= m) break;
            par ^= 1;
        }
    } else {
        // ================================================ the updaters =========================================================
        PrSlots<P> S;
        fps_pruned_prologue<P, GS, UT>(n, Q, src, smem, S);
        pn2_f2 (&xx)[P / 2] = S.xx;
        pn2_f2 (&yy)[P / 2] = S.yy;
        pn2_f2 (&zz)[P / 2] = S.zz;
        float (&md)[P] = S.md;
        unsigned (&low)[P] = S.low;
        // (only wave w ever raises bound[.][w]; the picker has zeroed the exchange area before the prologue's barriers)
        // lane l tests samples against the box of THIS wave's group l % GW
        float blx, bly, blz, bhx, bhy, bhz;
        {
            const float *gbox = reinterpret_cast<const float *>(smem + 256 + (size_t)16 * NS + (size_t)kPrHistRows * kPrBins * 4);
            const float *o = gbox + (w * GW + (lane & (GW - 1))) * 8;
            blx = o[0]; bly = o[1]; blz = o[2]; bhx = o[4]; bhy = o[5]; bhz = o[6];
        }
        pn2_f2 sxy = {0.f, 0.f}, syy = {0.f, 0.f}, szk = {0.f, 0.f};   // the sample in the LOW halves (fps_body.h: the high-half broadcast form is not safe)
        auto update_group = [&](auto gic) __attribute__((always_inline)) {
            constexpr int gi = decltype(gic)::value;
            constexpr int H = GS / 2;
            pn2_f2 dx[H], dy[H], dz[H];
#pragma unroll
            for (int h = 0; h < H; ++h) dx[h] = pk_sub_bcast_lo(xx[gi * H + h], sxy);
#pragma unroll
            for (int h = 0; h < H; ++h) dy[h] = pk_sub_bcast_lo(yy[gi * H + h], syy);
#pragma unroll
            for (int h = 0; h < H; ++h) dz[h] = pk_sub_bcast_lo(zz[gi * H + h], szk);
#pragma unroll
            for (int h = 0; h < H; ++h) dx[h] = pk_mul(dx[h], dx[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) dy[h] = pk_mul(dy[h], dy[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) dz[h] = pk_mul(dz[h], dz[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) dx[h] = pk_add(dx[h], dy[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) dx[h] = pk_add(dx[h], dz[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int p0 = gi * GS + 2 * h;
                md[p0] = vmin_f32(dx[h].x, md[p0]);              // min(d,td), tf_sampling_g.cu:144
                md[p0 + 1] = vmin_f32(dx[h].y, md[p0 + 1]);
            }
        };
        auto update_all = [&]() __attribute__((always_inline)) {
            update_group(std::integral_constant<int, 0>());
            if constexpr (GW >= 2) update_group(std::integral_constant<int, 1>());
            if constexpr (GW >= 4) { update_group(std::integral_constant<int, 2>()); update_group(std::integral_constant<int, 3>()); }
            if constexpr (GW == 8) {
                update_group(std::integral_constant<int, 4>()); update_group(std::integral_constant<int, 5>());
                update_group(std::integral_constant<int, 6>()); update_group(std::integral_constant<int, 7>());
            }
        };
        // sample 0 is point 0 (tf_sampling_g.cu:114-116): every slot against it
        {
            const float4 s = lds_rank[NS - 1];
            sxy.x = s.x; syy.x = s.y; szk.x = s.z;
            update_all();
        }
        if (t == 0) {
            dst[0] = 0;
            if (PUBLISH) __hip_atomic_store(gtag, (unsigned long long)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned vlastb = __float_as_uint(1e38f);    // value bits of the last sample: no running distance is above it
        unsigned thetab = __float_as_uint(1e38f);    // first batch: nobody reaches it, every wave sends its exact best lane
        // ---- EARLY: the first samples, one per exchange (the full tier's round, fps_body.h) --------------------------------------
        // While the farthest-point distance still halves every few samples a list determines three or four samples and every
        // sample reaches most groups: a batch then costs more than the classic round (measured: the first 64 samples 44 us in
        // batches, 25 us like this; the whole chain at 4096 -> 1024 280 -> 255 us). The picker and its barriers: one per round, see its branch.
        // The update works on single registers, not pairs, as the full tier does at 512 threads (fps_body.h: two waves per SIMD, 411
        // against 438 ns per round there; 395 against 440 here). The index is stored at once: behind the next round's key reads
        // (the full tier's place for it) the overlapped launch lost 6 us. Also what the workgroup falls back to when batches do not
        // pay (SLOW BATCHES above).
        auto single_rounds = [&](const int jend) __attribute__((always_inline)) {
            double *partial = reinterpret_cast<double *>(smem);                  // [2][W]: the pruned layout's wave keys
            for (; j < jend; ++j) {
                double kd[P];
#pragma unroll
                for (int p = 0; p < P; ++p) kd[p] = __hiloint2double(__float_as_int(md[p]), (int)low[p]);
#pragma unroll
                for (int st = 1; st < P; st <<= 1)
#pragma unroll
                    for (int i = 0; i + st < P; i += 2 * st)
                        asm("v_max_f64 %0, %1, %2" : "=v"(kd[i]) : "v"(kd[i]), "v"(kd[i + st]));
                const double wd = wave_max_f64_lane63(kd[0]);
                double *slot = partial + (j & 1) * W;
                if (lane == 63) slot[w] = wd;
                __syncthreads();
                double key[W];
#pragma unroll
                for (int i = 0; i < W; ++i) key[i] = slot[i];
#pragma unroll
                for (int st = 1; st < W; st <<= 1)
#pragma unroll
                    for (int i = 0; i + st < W; i += 2 * st)
                        asm("v_max_f64 %0, %1, %2" : "=v"(key[i]) : "v"(key[i]), "v"(key[i + st]));
                vlastb = (unsigned)__double2hiint(key[0]);
                const float4 s = lds_rank[(unsigned)__double2loint(key[0])];    // same address in every lane: LDS broadcast
                if (t == 0) {
                    const int k = __float_as_int(s.w);
                    dst[j] = k;
                    if (PUBLISH)
                        __hip_atomic_store(gtag + j, ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)k, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                if constexpr (P >= PN2_BT_PACKED_FROM) {                         // (16 slots on single registers: spills at 576 threads)
                    sxy.x = s.x; syy.x = s.y; szk.x = s.z;
                    update_all();
                } else {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float d = p & 1 ? sqdist(xx[p / 2].y, yy[p / 2].y, zz[p / 2].y, s.x, s.y, s.z)
                                              : sqdist(xx[p / 2].x, yy[p / 2].x, zz[p / 2].x, s.x, s.y, s.z);   // tf_sampling_g.cu:141-143
                        md[p] = vmin_f32(d, md[p]);                              // :144
                    }
                }
            }
            thetab = __float_as_uint(__fmul_rn(__uint_as_float(vlastb), 1.0f - PN2_BT_G0));
        };
        if (jE > 1) single_rounds(jE);
        int par = 0;
        if (j < m)
        for (;;) {
            BtXchg &X = xch[par];
            const long long u0 = PN2_BT_CLOCK();
            // ---- COLLECT ----------------------------------------------------------------------------------------------------------
            double kl;
            float sec;
            {
                // best key and the two largest values of the lane's P slots (straight-line: 2 P - 1 tournament nodes)
                double kd[P];
                float hv[P / 2], lv[P / 2];
#pragma unroll
                for (int p = 0; p < P; ++p) kd[p] = __hiloint2double(__float_as_int(md[p]), (int)low[p]);
#pragma unroll
                for (int i = 0; i < P / 2; ++i) { hv[i] = vmax_f32(md[2 * i], md[2 * i + 1]); lv[i] = vmin_f32(md[2 * i], md[2 * i + 1]); }
#pragma unroll
                for (int st = 1; st < P; st <<= 1)
#pragma unroll
                    for (int i = 0; i + st < P; i += 2 * st)
                        asm("v_max_f64 %0, %1, %2" : "=v"(kd[i]) : "v"(kd[i]), "v"(kd[i + st]));
#pragma unroll
                for (int st = 1; st < P / 2; st <<= 1)
#pragma unroll
                    for (int i = 0; i + st < P / 2; i += 2 * st) {
                        const float ll = vmax_f32(lv[i], lv[i + st]);
                        lv[i] = vmed3_f32(hv[i], hv[i + st], ll);     // second largest of (h1 >= l1, h2 >= l2)
                        hv[i] = vmax_f32(hv[i], hv[i + st]);
                    }
                kl = kd[0];
                sec = lv[0];
            }
            const unsigned vb = (unsigned)__double2hiint(kl);
            unsigned long long mask = __ballot(vb >= thetab);
            int cnt = __popcll(mask);
            unsigned thb = thetab;                   // this wave's threshold: its points outside the list are below it
            bool exact = false;
            if (cnt > CAP) {
                unsigned lob = thetab, hib = vlastb + 1u;       // more than CAP lanes at lob, none at hib
                for (int it = 0; it < 16; ++it) {
                    const unsigned mid = lob + ((hib - lob) >> 1);
                    if (mid == lob) break;
                    const unsigned long long mk = __ballot(vb >= mid);
                    const int c = __popcll(mk);
                    PN2_BT_STAT(3, 1);
                    if (c > CAP) lob = mid;
                    else if (c == 0) hib = mid;
                    else { mask = mk; cnt = c; thb = mid; break; }
                }
                exact = cnt > CAP;
            } else if (cnt == 0) {
                exact = true;
            }
            if (exact) {                              // the wave's best lane alone
                PN2_BT_STAT(2, 1);
                const double wk = wave_max_f64_lane63(kl);
                const int bh = __builtin_amdgcn_readlane(__double2hiint(wk), 63), bl = __builtin_amdgcn_readlane(__double2loint(wk), 63);
                unsigned long long mk = __ballot(__double2hiint(kl) == bh && __double2loint(kl) == bl);
                mk &= 0ull - mk;                      // equal keys exist only among padding slots
                if (cnt != 0) thb = (unsigned)bh + 1u;   // too many equal values: everything else of the wave is <= this lane's value, with a lower key
                mask = mk;
                cnt = 1;
            }
            if ((mask >> lane) & 1ull) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                X.list[w * CAP + pos] = kl;
                lds_max_u32(&X.bound[w], __float_as_uint(sec) + 1u);
            }
            if (lane == 0) {
                X.cnt[w] = (unsigned)cnt;
                lds_max_u32(&X.bound[w], thb);
            }
            const long long u1 = PN2_BT_CLOCK();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm LDS operations above are invisible to the compiler's counters
            __syncthreads();
            const long long u2 = PN2_BT_CLOCK();
            if (lane == 0) xch[par ^ 1].bound[w] = 0u;            // next batch's word of this wave (nobody reads it before the next barrier)
            // ---- APPLY: the picker's samples as they appear ----------------------------------------------------------------------------
            int done = 0;
            unsigned c;
            // v* of the skip test: the previous batch's last sample value bounds every running distance (fps_pruned_body: the exact skip)
            const float thr = __fadd_rn(__fmul_rn(__uint_as_float(vlastb), 1.00001f), 1e-30f);
            for (;;) {
                c = __hip_atomic_load(&X.count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                const int avail = (int)(c & kBtCountMask);
                if (avail == done) {
                    if (c & kBtEnd) break;
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                constexpr int CH = PN2_WAVE / GW;                                 // samples per chunk: lane l tests sample l / GW against group l % GW
                constexpr unsigned long long kStride = GW == 8 ? 0x0101010101010101ull : GW == 4 ? 0x1111111111111111ull
                                                       : GW == 2 ? 0x5555555555555555ull : ~0ull;   // one bit per sample
                const int np = min(avail - done, CH);
                const int pi = lane / GW;                                        // this lane's sample of the chunk
                const float4 s = ring[done + (pi < np ? pi : 0)];
                const float ax = __fsub_rn(s.x, __builtin_amdgcn_fmed3f(s.x, blx, bhx));
                const float ay = __fsub_rn(s.y, __builtin_amdgcn_fmed3f(s.y, bly, bhy));
                const float az = __fsub_rn(s.z, __builtin_amdgcn_fmed3f(s.z, blz, bhz));
                const float bd = __fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az));
                unsigned long long touched = ~__ballot(bd >= thr);               // NaN -> not far -> updated
                if (np * GW < 64) touched &= (1ull << (np * GW)) - 1ull;
                // group by group (static): the samples that reach group g, straight into that group's update -- no dispatch on a
                // group number (three compare-and-branch pairs per (group, sample) otherwise)
                if (touched) {
                    // the first samples of a cloud reach most groups: then every sample of the chunk updates all of the wave's
                    // groups in straight-line code (an update of an untouched group changes nothing)
                    if (2 * __popcll(touched) >= GW * np) {
                        for (int p = 0; p < np; ++p) {
                            sxy.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.x), p * GW));
                            syy.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.y), p * GW));
                            szk.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.z), p * GW));
                            update_all();
                        }
                    } else {
                        auto one_group = [&](auto gic) __attribute__((always_inline)) {
                            constexpr int g8 = decltype(gic)::value;
                            unsigned long long mg = touched & (kStride << g8);
                            while (mg) {
                                const int sl = (int)__builtin_ctzll(mg) & ~(GW - 1);
                                mg &= mg - 1ull;
                                sxy.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.x), sl));
                                syy.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.y), sl));
                                szk.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.z), sl));
                                PN2_BT_STAT(5, 1);
                                update_group(gic);
                            }
                        };
                        one_group(std::integral_constant<int, 0>());
                        if constexpr (GW >= 2) one_group(std::integral_constant<int, 1>());
                        if constexpr (GW >= 4) { one_group(std::integral_constant<int, 2>()); one_group(std::integral_constant<int, 3>()); }
                        if constexpr (GW == 8) {
                            one_group(std::integral_constant<int, 4>()); one_group(std::integral_constant<int, 5>());
                            one_group(std::integral_constant<int, 6>()); one_group(std::integral_constant<int, 7>());
                        }
                    }
                }
                done += np;
            }
            const long long u3 = PN2_BT_CLOCK();
            if (w == 0) { PN2_BT_STAT(9, u1 - u0); PN2_BT_STAT(10, u3 - u2); }
            const int a = (int)(c & kBtCountMask);
            // ---- the batch's samples leave in one store ------------------------------------------------------------------------------
            if (w == 0 && lane < a) {
                const int k = __float_as_int(ring[lane].w);
                dst[j + lane] = k;
                if (PUBLISH)
                    __hip_atomic_store(gtag + (j + lane), ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)k, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            if (c & kBtFill) fill_k = __float_as_int(ring[a - 1].w);
            j += a;
            if ((c & kBtFill) || j >= m) break;
            thetab = X.theta;
            vlastb = X.vlast;
            const int single = __builtin_amdgcn_readfirstlane((int)X.single);    // SLOW BATCHES (the picker's decision, published ahead of the end flag)
            if (single) {
                single_rounds(min(j + single, m));                               // (the picker has clipped it already: a run never passes the output row)
                if (j >= m) break;
            }
            par ^= 1;
        }
    }
    if (fill_k >= 0) {
        for (int i = j + t; i < m; i += BT) {
            dst[i] = fill_k;
            if (PUBLISH)
                __hip_atomic_store(gtag + i, ((unsigned long long)tag << 32) | (unsigned long long)(unsigned)fill_k, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    fps_gather_epilogue<BT>(m, src, dst, dxyz);
}

}  // namespace pn2
