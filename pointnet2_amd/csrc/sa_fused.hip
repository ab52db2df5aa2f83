// sa_fused.hip -- the xyz half of sample_and_group in ONE launch, with the ball queries overlapped
// under the farthest-point-sampling chain.
//
// What it computes is exactly reference utils/pointnet_util.py:40-46:
//     fps_idx = farthest_point_sample(npoint, xyz);  new_xyz = gather_point(xyz, fps_idx)
//     idx, pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)
//     grouped_xyz = group_point(xyz, idx) [- new_xyz]
// with the same kernels' device bodies (fps_body.h, ball_query_body.h), so every output is
// bit-identical to the separate operators (tests/test_parity_gpu.py::test_sample_and_group_overlapped).
//
// Why: FPS is a serial chain that keeps ONE CU per cloud busy for its whole duration (32 of 256 CUs at
// B=32) while the ball queries (a whole-GPU kernel, 56 us at the metric shape) can only start after it
// in a stream. Query j, however, needs nothing but sample j. Here the grid is heterogeneous:
//     blocks [0, b)          FPS producers, one per cloud; thread 0 publishes every selected index as
//                            an 8-byte {tag, index} granule with one write-through store;
//     blocks [b, b + c*b)    PERSISTENT ball-query consumers, c (= 1) per cloud (round 5): a consumer stages and bins its
//                            cloud once and walks the 64-query ranges in publish order, each wave polling the granules
//                            of its queries. (Rounds 2-4: one workgroup per range, 16 per cloud at the metric shape --
//                            544 workgroups, 224 of them spinning on every idle CU for the whole chain, each re-staging
//                            the cloud. Now 2 b = 64 workgroups; the step time is the same, 192 CUs stay free.)
// Hand-off: form R2 of the CDNA programming guide (Guideline 16) -- the data is the flag, agent-scope
// relaxed 8-byte atomics on both sides, no fences. What makes a granule THIS launch's is its tag, and the tag comes from
// one of three places (sample_and_group_common):
//   generation 1 .. 0xfffffffe   the caller numbers its launches on a workspace it zeroed once (pn2_..._gen);
//   PN2_GENERATION_DEVICE        the launch numbers ITSELF (round 6): every workgroup of a cloud -- its producer and its
//                                consumers -- arrives at the cloud's counter word behind the granules with ONE returning
//                                atomic add (a read-modify-write is served by the memory side, never by a cached copy),
//                                the count gives all of them the same launch ordinal, and the last arriver moves the word
//                                to the next ordinal. No clear, no host-supplied number: the form for CAPTURED graphs,
//                                whose arguments are frozen (fused_arrive below says what went wrong before);
//   generation 0                 the workspace is cleared on the stream in front of the launch (a kernel, never a memset
//                                node: pn2_device.h) and the tag is 1 -- eager launches only: inside a capture the call
//                                takes the two launches instead.
// Every workgroup asks for more than half of the CU's LDS, so producers never share
// a CU with consumers (a co-resident consumer would steal issue slots from the latency-bound chain).
//
// Forward progress. Consumers spin until their producer has advanced; that cannot deadlock while the b
// producer workgroups are running, which holds because (1) workgroups start in block-index order on this
// hardware (producers are blocks [0, b); MI355X_MICROARCH.md: a 1024-block grid starts first -> last within
// 0.34-0.69 us) and (2) the launch is refused unless the device can hold all b producer workgroups at once
// (occupancy query at launch, cached): b <= 256 = the CUs, and every workgroup asks for more than half of a
// CU's LDS, so a producer never shares its CU. HIP does not PROMISE (1), so the consumers' spin is bounded
// (~10 s) and a consumer that gives up records it in the status word of `ws`
// (pn2_sample_and_group_status_offset) and exits: an error the host can read, not a trap that would take
// the process down. Shapes outside the envelope are refused (callers use the two-launch path:
// pn2_farthest_point_sample_gather + pn2_query_ball_group_xyz; the Python layer also has a switch).
// Tried and measured (round 2), neither shipped: (1) the last 12 queries of a cloud answered by its PRODUCER workgroup
// after the final round (candidate-parallel over the LDS mirror, bitmaps, bq_flush): correct, but 421.8-424.6 us per
// step against 419.8 -- scripts/fused_tail_probe.py shows why: the launch ends 5-7 us after the SLOWEST of the 32
// chains (they spread over 3.5 us), and the epilogue lengthens exactly that workgroup. (2) Roles taken by ARRIVAL TICKET (one returning atomicAdd per workgroup; the
// first b arrivals become producers), which needs no assumption about dispatch order -- 452-470 us per
// launch against 419 with roles by block index (the ticket winners are the workgroups closest to the
// counter's memory channel, i.e. the producers end up bunched on one XCD). Not shipped.
#include "ball_query_body.h"
#include "fps_body.h"
#include "fps_pruned_body.h"
#include "fps_batch_body.h"

#include <limits.h>

namespace pn2 {

constexpr int kFusedThreads = 512;
constexpr int kFusedMaxClouds = 256;            // one producer per CU at most (round 6: beyond 128 clouds the consumers take the CUs the producers leave, the rest
                                                // of them when chains end -- the `batched` leg of bench.py)
constexpr size_t kFusedMinLds = 82 * 1024;      // > 160 KiB / 2: one workgroup per CU
constexpr int kFusedConsumers = 1;              // persistent consumer workgroups per cloud: 1 / 2 / 4 / 16 measured 397.6 / 398.0 / 398.5 /
                                                // 398.4 us per step at the metric shape (profiles/r05/fused_consumers.txt)


// ---- the launch numbers itself (PN2_GENERATION_DEVICE) -------------------------------------------------------------------
// History (profiles/r05/geometry_ahead.txt, profiles/r06/stale_granules.md): rounds 2-4 captured this launch behind a
// hipMemsetAsync with the constant tag 1, and a serving loop's soak found replays whose consumers accepted granules that were
// not the replay's. Round 6 found why: a replayed memset NODE of this runtime does not clear -- it fills the buffer with a
// 16-byte pattern taken from wherever its fill pattern was staged at capture time, which by then holds the launch arguments of
// some later eager kernel. In the serving loop that was the soak's `output != expected` compare: (numel 1280, 1, a pointer),
// so every EVEN granule read {tag 1, index 1280} -- published, as far as a consumer can tell -- and half of new_xyz's rows
// became point 1280. (With other neighbours the garbage carries other tag words and nothing shows: most processes "worked".)
// The library no longer issues hipMemsetAsync anywhere (pn2_device.h: clear_async). Independently of that, a constant tag
// makes every word that survives -- a late or skipped clear, memory that was somebody else's in between -- look published;
// here no launch can mistake anything older for its own: consecutive launches on a workspace carry different tags, and the
// tag is agreed by read-modify-writes on one word per cloud.
//
// word (8 bytes per cloud, behind the status word; zero when the workspace is new):  [63:16] ordinal | [15:0] arrivals
// Every participant of a cloud (1 producer + cpc consumers; npc <= 129: one consumer per 64-query range at most) adds 1 and
// reads the old word: the ordinal field is the launch's ordinal for all of them, because the word leaves this ordinal only
// when the LAST of them has arrived -- that one adds 65536 - npc (arrivals back to 0, ordinal + 1). Launches on one workspace
// are ordered by the stream, so the next launch's first arrival finds arrivals = 0. npc may differ from launch to launch.
// tag = ordinal mod (2^32 - 1) + 1: never 0, and different for consecutive launches, which is all that is needed -- every
// launch rewrites every granule of (b, m).
__device__ __forceinline__ unsigned fused_arrive(unsigned long long *word, unsigned npc)
{
    pn2_gu64 *w = (pn2_gu64 *)word;
    const unsigned long long old = __hip_atomic_fetch_add(w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(old & 0xffffull) + 1u == npc)
        __hip_atomic_fetch_add(w, 65536ull - (unsigned long long)npc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)((old >> 16) % 0xffffffffull) + 1u;
}

#ifdef PN2_LAB_STALE
// Lab build only (make lab_stale -> build_lab/libpn2ops_stalelab.so, scripts/stale_granule_repro.py): what does a workgroup
// find in its cloud's LAST granule -- published by the chain's final round, so any tag there at entry is an older launch's --
// when it starts, (1) by the agent-scope load the consumers poll with and (2) by a read-modify-write, which the memory side
// serves? [0] workgroups, [1] load saw this launch's tag, [2] RMW saw it, [3] load saw any non-zero word, [4] RMW saw any.
__device__ unsigned long long pn2_lab_stale[8];
__device__ unsigned long long pn2_lab_samples[64 * 4];    // first 64 non-zero finds of launches with tag 1: load, RMW, block | m << 32, first granule
static int g_lab_clear_with_kernel = 0;
#endif

// LPQ: lanes per query of the cell-list consumers; 0 = sweep consumers (clouds whose cell list does not fit
// beside the position table: n > ~6000).
// TIER of the producers: 0 = fps_reg_body; 1 = the kd-grouped chain of fps_pruned_body.h (4096 / 8192 rank slots: P = 8, 16) on four
// waves; 2 = the same groups with several samples per exchange (fps_batch_body.h: eight updater waves and the picker, 576 threads).
constexpr int fused_pruned_gs(int P) { return P <= 8 ? 2 : 4; }       // slots per group (pruned tier: 16 / 32 slots per thread; batched: 2 .. 16)
template <int P, int LPQ, int TIER = 0>
__global__ __launch_bounds__(TIER == 2 ? kBtT : kFusedThreads) void sa_fused_kernel(int b, int n, int m, int Q, int nsample, float thr,
                                                                 float radius, int qpb, int cpc, unsigned tag,
                                                                 const float *__restrict__ xyz,
                                                                 unsigned long long *__restrict__ tagged,
                                                                 int *__restrict__ fps_idx,
                                                                 float *__restrict__ new_xyz, int *__restrict__ idx,
                                                                 int *__restrict__ pts_cnt,
                                                                 float *__restrict__ grouped, int subtract)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *status = reinterpret_cast<unsigned *>(tagged + (size_t)b * m);   // launch status word behind the granules
    const int blk = blockIdx.x;
    if (tag == PN2_GENERATION_DEVICE) {
        // one arrival per workgroup at its cloud's counter word; the whole workgroup takes the tag from LDS (the bodies below
        // own the dynamic LDS from the second barrier on)
        if (threadIdx.x == 0)
            *reinterpret_cast<volatile unsigned *>(smem) = fused_arrive(tagged + (size_t)b * m + 2 + (blk < b ? blk : (blk - b) % b), 1u + (unsigned)cpc);
        __syncthreads();
        tag = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned *>(smem));
        __syncthreads();
    }
#ifdef PN2_LAB_STALE
    if (threadIdx.x == 0) {
        pn2_gu64 *last = (pn2_gu64 *)(tagged + (size_t)(blk < b ? blk : (blk - b) % b) * m + (m - 1));
        const unsigned long long v1 = __hip_atomic_load(last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long v2 = __hip_atomic_fetch_or(last, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&pn2_lab_stale[0], 1ull);
        if ((unsigned)(v1 >> 32) == tag) atomicAdd(&pn2_lab_stale[1], 1ull);
        if ((unsigned)(v2 >> 32) == tag) atomicAdd(&pn2_lab_stale[2], 1ull);
        if (v1) atomicAdd(&pn2_lab_stale[3], 1ull);
        if (v2) atomicAdd(&pn2_lab_stale[4], 1ull);
        if (tag == 1u && (v1 | v2)) {
            const unsigned long long k = atomicAdd(&pn2_lab_stale[5], 1ull);
            if (k < 64) {
                pn2_lab_samples[4 * k + 0] = v1;
                pn2_lab_samples[4 * k + 1] = v2;
                pn2_lab_samples[4 * k + 2] = (unsigned long long)blk | ((unsigned long long)m << 32) | ((unsigned long long)b << 48);
                pn2_lab_samples[4 * k + 3] = __hip_atomic_load((pn2_gu64 *)(tagged + (size_t)(blk < b ? blk : (blk - b) % b) * m + (m - 2)),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#endif
#ifndef PN2_FUSED_LAB_PUBLISH                     // lab switches (scripts/: where does the launch's time go?)
#define PN2_FUSED_LAB_PUBLISH true
#endif
    if (blk < b) {
#ifdef PN2_FUSED_LAB_TIMES
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
#endif
        // Up to 2048 rank slots the chain is shorter with FOUR waves of twice the points each (273 vs 288 ns per round
        // at n = 1024, 312 vs 326 at 2048: a smaller block-wide arg-max; profiles/r02/fps_experiments.txt), which is what
        // pn2_farthest_point_sample launches at these sizes: the upper four waves of a producer retire at once
        // (a retired wave no longer counts at the workgroup's barriers).
        if constexpr (TIER == 2) {
            // eight updater waves of P slots per thread + the picker: the launch's workgroups are kBtT = 576 threads for this tier
            static_assert(kBtUT == kFusedThreads, "the batched tier's updaters are the workgroup's first 512 threads");
            fps_batch_body<P, fused_pruned_gs(P), PN2_FUSED_LAB_PUBLISH>(n, m, Q, blk, xyz, fps_idx, nullptr, tagged, smem, tag);
        } else if constexpr (TIER == 1) {
            if (threadIdx.x >= kPrT) return;
            fps_pruned_body<2 * P, fused_pruned_gs(P), PN2_FUSED_LAB_PUBLISH>(n, m, Q, blk, xyz, fps_idx, nullptr, tagged, smem, tag);
        } else if constexpr (kFusedThreads * P <= 2048) {
            if (threadIdx.x >= kFusedThreads / 2) return;
            fps_reg_body<kFusedThreads / 2, 2 * P, true, PN2_FUSED_LAB_PUBLISH>(n, m, Q, blk, xyz, fps_idx, nullptr, tagged, smem, tag);
        } else {
            fps_reg_body<kFusedThreads, P, true, PN2_FUSED_LAB_PUBLISH>(n, m, Q, blk, xyz, fps_idx, nullptr, tagged, smem, tag);
        }
#ifdef PN2_FUSED_LAB_TIMES
        if (threadIdx.x == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            if (blk == 0) { status[1] = (unsigned)(t1 - t0); status[2] = (unsigned)t1; }   // chain (10 ns ticks), its absolute end
            status[8 + 2 * blk] = (unsigned)t0;                              // the probe allocates ws with room for these
            status[9 + 2 * blk] = (unsigned)t1;
        }
#endif
#ifdef PN2_FUSED_LAB_TIMES
        if (threadIdx.x == 0) atomicMax(status + 3, (unsigned)__builtin_amdgcn_s_memrealtime());   // last workgroup to finish
#endif
    } else {
#ifdef PN2_FUSED_LAB_NO_CONSUMERS
        return;
#endif
        // PERSISTENT consumers (round 5): cpc workgroups per cloud; consumer c stages (and bins) its cloud ONCE and walks the
        // query ranges c, c + cpc, c + 2 cpc, ... in the order the producer publishes them. (Rounds 2-4 launched one workgroup per
        // RANGE, m / 64 = 16 per cloud at the metric shape: 512 workgroups spinning on 224 CUs, each re-staging the cloud.)
        if (TIER == 2 && threadIdx.x >= kFusedThreads) return;      // the picker's wave has no part in a consumer workgroup
        const int id = blk - b;
        const int cloud = id % b;                // consumer number first, cloud second: the consumers that can
        const int q0 = (id / b) * qpb;           // start earliest are dispatched first
        const int stride = cpc * qpb;
        if (LPQ == 0) {
            if (bq_block_body<true, true, true>(n, m, nsample, thr, cloud, q0, min(q0 + qpb, m), xyz, nullptr, tagged,
                                                new_xyz, idx, pts_cnt, grouped, subtract, smem, tag, 0, status))
                for (int a = q0 + stride; a < m; a += stride)
                    if (!bq_block_body<true, true, true, kBqThreads, true>(n, m, nsample, thr, cloud, a, min(a + qpb, m), xyz, nullptr, tagged,
                                                                           new_xyz, idx, pts_cnt, grouped, subtract, smem, tag, 0, status)) break;
        } else {
            bq_cells_block_body<kFusedThreads, (LPQ ? LPQ : 8), true, true>(n, m, nsample, thr, radius, cloud, q0,
                                                                          min(q0 + qpb, m), xyz, nullptr, tagged, new_xyz,
                                                                          idx, pts_cnt, grouped, subtract, smem, tag, status, stride);
        }
#ifdef PN2_FUSED_LAB_TIMES
        if (threadIdx.x == 0) atomicMax(status + 3, (unsigned)__builtin_amdgcn_s_memrealtime());
#endif
    }
}

// LDS of a cell-list consumer workgroup (512 threads) incl. the position table and the sweep fallback's layout
static size_t fused_cells_lds(int n, int nsample, int lpq)
{
    const size_t cells = bq_cells_lds_bytes(n, nsample, lpq, kFusedThreads) + sizeof(unsigned short) * (size_t)n;
    const size_t sweep = sizeof(float4) * (size_t)((n + 127) & ~127) + sizeof(int) * (size_t)nsample * kBqWaves * kBqQpw;
    return cells > sweep ? cells : sweep;
}

template <int P, int LPQ, int TIER = 0>
static int launch_fused(int b, int n, int m, int Q, int nsample, float thr, float radius, unsigned tag, const float *xyz,
                        unsigned long long *ws, int *fps_idx, float *new_xyz, int *idx, int *pts_cnt,
                        float *grouped, int subtract, hipStream_t st, int consumers)
{
    constexpr int kGran = kBqWaves * kBqQpw;
    // a query range is 64 queries (a multiple of 16); `consumers` persistent workgroups per cloud share the ranges
    int qpb = 64;
    if (qpb > m) qpb = ((m + kGran - 1) / kGran) * kGran;
    const int nranges = (m + qpb - 1) / qpb;
    const int nq = consumers <= 0 ? (nranges < kFusedConsumers ? nranges : kFusedConsumers) : consumers < nranges ? consumers : nranges;
    size_t lds_f = TIER == 2 ? fps_batch_lds_bytes(P) : TIER == 1 ? fps_pruned_lds_bytes(2 * P)
                                                                       : 256 + sizeof(float4) * (size_t)kFusedThreads * P;
    size_t lds_q = LPQ ? fused_cells_lds(n, nsample, LPQ)
                       : sizeof(float4) * (size_t)((n + 127) & ~127) + sizeof(int) * (size_t)nsample * kGran;
    size_t lds = lds_f > lds_q ? lds_f : lds_q;
    if (lds < kFusedMinLds) lds = kFusedMinLds;
    if (lds > 160 * 1024) return PN2_E_TOO_LARGE;
    auto kern = sa_fused_kernel<P, LPQ, TIER>;
    if (int rc = allow_dynamic_lds(kern, lds)) return rc;
    {
        // residency (header): room for every producer at the same time. Consumers only ever wait for producers and producers wait
        // for nobody, so consumers that find no CU free simply start when a workgroup ends (b > CUs / 2: the last of them after
        // the chains, where they find every sample published)
        const int room = resident_workgroups(kern, TIER == 2 ? kBtT : kFusedThreads, lds);
        if (room < 0) return -room;
        if (room < b) return PN2_E_TOO_LARGE;
    }
    if (tag == 0) {                                               // caller did not manage generations: clear, use tag 1
        // (never inside a capture: sample_and_group_common has sent captured calls to the two launches.) The counter words of
        // the device-numbered form are left alone: a workspace serves one form for its life.
#ifdef PN2_LAB_STALE
        if (!g_lab_clear_with_kernel) {                           // the lab build's default: the memset NODE of rounds 2-4
            hipError_t e = hipMemsetAsync(ws, 0, sizeof(unsigned long long) * (size_t)b * m + 16, st);
            if (e != hipSuccess) return (int)e;
        } else
#endif
        if (int rc = clear_async(ws, sizeof(unsigned long long) * (size_t)b * m + 16, st)) return rc;
        tag = 1u;
    }
    if (int rc = launch(kern, dim3(b + nq * b), dim3(TIER == 2 ? kBtT : kFusedThreads), lds, st, b, n, m, Q, nsample, thr, radius, qpb, nq, tag, xyz, ws,
                       fps_idx, new_xyz, idx, pts_cnt, grouped, subtract)) return rc;
    return PN2_OK;
}

}  // namespace pn2

extern "C" long long pn2_sample_and_group_ws_bytes(int b, int m)
{
    if (b <= 0 || m <= 0) return 0;
    // b*m sample granules + the launch status word (padded to 16 bytes) + one counter word per cloud (PN2_GENERATION_DEVICE)
    return (long long)sizeof(unsigned long long) * b * m + 16 + (long long)sizeof(unsigned long long) * b;
}

static int sample_and_group_common(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws, unsigned tag,
                                   int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz,
                                   int subtract_centroid, void *stream, int fps_variant = PN2_FPS_AUTO, int consumers = 0)
{
    using namespace pn2;
    if (!(radius > 0.0f) || nsample <= 0 || m <= 0) return PN2_E_ARG;
    if (b <= 0 || n <= 0) return PN2_E_SHAPE;
    if (!xyz || !ws || !fps_idx || !new_xyz || !idx || !pts_cnt || !grouped_xyz) return PN2_E_NULL;
    // envelope of the overlapped launch (outside it the caller uses the two-launch path)
    if (b > kFusedMaxClouds || n > 8192 || n < 64 || nsample > 256) return PN2_E_TOO_LARGE;
    if ((long long)b * m * nsample * 3 > INT_MAX) return PN2_E_TOO_LARGE;
    hipStream_t st = as_stream(stream);
    if (tag == 0u) {
        // The cleared form (clear + the constant tag 1) is for eager launches. A captured one would be replayed with the same
        // tag behind a clear it has to trust -- and the clear of rounds 2-4, a memset node, turned out not to clear when
        // replayed (profiles/r06/stale_granules.md). The clear is a kernel now, but a captured call still enqueues the two
        // launches instead (same outputs, `ws` untouched): callers that want the overlapped launch inside a graph pass
        // PN2_GENERATION_DEVICE, whose correctness rests on no clear at all.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        hipError_t e = hipStreamIsCapturing(st, &cs);
        if (e != hipSuccess) return (int)e;
#ifdef PN2_LAB_STALE
        cs = hipStreamCaptureStatusNone;                          // the lab build captures the cleared form, as rounds 2-4 did
#endif
        if (cs != hipStreamCaptureStatusNone) {
            if (fps_variant < PN2_FPS_AUTO || fps_variant > PN2_FPS_BATCH) return PN2_E_ARG;
            if (int rc = pn2_farthest_point_sample_variant(fps_variant, b, n, m, xyz, nullptr, fps_idx, new_xyz, stream)) return rc;
            return pn2_query_ball_group_xyz(b, n, m, radius, nsample, xyz, new_xyz, subtract_centroid, idx, pts_cnt, grouped_xyz, stream);
        }
    }
    const int Q = (n + kRefThreads - 1) / kRefThreads;
    const int ranks = kRefThreads * Q;
    int P = 1;
    while (kFusedThreads * P < ranks) P <<= 1;
    const float thr = pn2_ball_threshold(radius);
    unsigned long long *w = reinterpret_cast<unsigned long long *>(ws);
    // consumers: the cell-list body with as many queries per wave as fit beside the sorted cloud, the cell table
    // and the position table; the sweep body when nothing fits (large clouds) or the cloud is tiny
    int lpq = 0;
    if (n >= 1024)
        for (int cand : {8, 16, 32})
            if (fused_cells_lds(n, nsample, cand) <= 160 * 1024) { lpq = cand; break; }
    // Clouds too large for a cell list beside their sorted copy (n > ~7000) leave the consumers the full sweep, and a sweep of
    // 8192 points per query does not hide under the chain: one persistent consumer per cloud 968 us, sixteen 478 us, the
    // two launches 459 us (sem_seg SA1, b = 8, 8192 -> 1024; profiles/r05/fused_sweep_consumers.txt). The library's own choice
    // there is the two launches -- same outputs, `ws` untouched; an explicit consumer count still gets the overlapped launch.
    if (lpq == 0 && n >= 1024 && consumers == 0) {
        if (fps_variant < PN2_FPS_AUTO || fps_variant > PN2_FPS_BATCH) return PN2_E_ARG;
        if (int rc = pn2_farthest_point_sample_variant(fps_variant, b, n, m, xyz, nullptr, fps_idx, new_xyz, stream)) return rc;
        return pn2_query_ball_group_xyz(b, n, m, radius, nsample, xyz, new_xyz, subtract_centroid, idx, pts_cnt, grouped_xyz, stream);
    }
    // producers: the kd-grouped chain where it exists (4096 / 8192 rank slots) and the chain is long enough to pay for the
    // kd build (the rule of pn2_farthest_point_sample, fps.hip)
    if (fps_variant < PN2_FPS_AUTO || fps_variant > PN2_FPS_BATCH) return PN2_E_ARG;
    if (fps_variant == PN2_FPS_PRUNED && P != 8 && P != 16) return PN2_E_ARG;
    if (fps_variant == PN2_FPS_BATCH && !fps_batch_covers(ranks)) return PN2_E_ARG;
    const int tier = fps_variant == PN2_FPS_BATCH || (fps_variant == PN2_FPS_AUTO && fps_batch_pays(ranks, m))     ? 2
                     : fps_variant == PN2_FPS_PRUNED || (fps_variant == PN2_FPS_AUTO && fps_pruned_pays(ranks, m)) ? 1
                                                                                                                    : 0;
#define PN2_FUSED_CASE(PP, LL, PR)                                                                                     \
    if (P == PP && lpq == LL && tier == PR)                                                                            \
        return launch_fused<PP, LL, PR>(b, n, m, Q, nsample, thr, radius, tag, xyz, w, fps_idx, new_xyz, idx, pts_cnt,   \
                                        grouped_xyz, subtract_centroid, st, consumers)
#define PN2_FUSED_P(PP, PR) PN2_FUSED_CASE(PP, 0, PR); PN2_FUSED_CASE(PP, 8, PR); PN2_FUSED_CASE(PP, 16, PR); PN2_FUSED_CASE(PP, 32, PR)
    PN2_FUSED_P(1, 0); PN2_FUSED_P(2, 0); PN2_FUSED_P(4, 0); PN2_FUSED_P(8, 0); PN2_FUSED_P(16, 0);
    PN2_FUSED_P(8, 1); PN2_FUSED_P(16, 1);
    PN2_FUSED_P(2, 2); PN2_FUSED_P(4, 2); PN2_FUSED_P(8, 2); PN2_FUSED_P(16, 2);
#undef PN2_FUSED_P
#undef PN2_FUSED_CASE
    return PN2_E_TOO_LARGE;
}

extern "C" int pn2_sample_and_group_xyz(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws,
                                        int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz,
                                        int subtract_centroid, void *stream)
{
    return sample_and_group_common(b, n, m, radius, nsample, xyz, ws, 0u, fps_idx, new_xyz, idx, pts_cnt, grouped_xyz,
                                   subtract_centroid, stream);
}

// Same launch without the per-call clear of `ws`: the caller manages GENERATIONS. `ws` must hold no granule
// whose tag word equals `generation` (zero it once when it is allocated, then pass 1, 2, 3, ... -- a
// granule left behind by an earlier generation can never be mistaken for a published sample). One
// workspace per stream; generation 0 is not allowed. PN2_GENERATION_DEVICE: the launch numbers itself (header of this
// file) -- `ws` zeroed once when it is allocated, used by this form only, never by two launches at the same time.
extern "C" int pn2_sample_and_group_xyz_gen(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws,
                                            unsigned generation, int *fps_idx, float *new_xyz, int *idx, int *pts_cnt,
                                            float *grouped_xyz, int subtract_centroid, void *stream)
{
    if (generation == 0u) return PN2_E_ARG;
    return sample_and_group_common(b, n, m, radius, nsample, xyz, ws, generation, fps_idx, new_xyz, idx, pts_cnt,
                                   grouped_xyz, subtract_centroid, stream);
}

// The same launch with the FPS tier of the producers (PN2_FPS_AUTO / _FULL / _PRUNED / _BATCH, see pn2_farthest_point_sample_variant;
// PN2_E_ARG for _PRUNED / _BATCH outside 2049..8192 rank slots) and the number of persistent consumer workgroups per cloud (0 = the
// library's choice; more than one per 64 queries is clamped) chosen by the caller. generation 0 = clear `ws` first.
extern "C" int pn2_sample_and_group_xyz_ex(int b, int n, int m, float radius, int nsample, const float *xyz, void *ws,
                                           unsigned generation, int fps_variant, int consumers, int *fps_idx, float *new_xyz,
                                           int *idx, int *pts_cnt, float *grouped_xyz, int subtract_centroid, void *stream)
{
    if (consumers < 0) return PN2_E_ARG;
    return sample_and_group_common(b, n, m, radius, nsample, xyz, ws, generation, fps_idx, new_xyz, idx, pts_cnt, grouped_xyz,
                                   subtract_centroid, stream, fps_variant, consumers);
}

// Offset (bytes) of the launch status word inside `ws`: 0 = ok, 1 = a consumer gave up waiting for its
// producer (the outputs of that launch are incomplete). Written by the device only in that case; a caller
// that wants to assert forward progress reads it after synchronising the stream.
extern "C" long long pn2_sample_and_group_status_offset(int b, int m)
{
    if (b <= 0 || m <= 0) return -1;
    return (long long)sizeof(unsigned long long) * b * m;
}

#ifdef PN2_LAB_STALE
// lab build only: counters of sa_fused_kernel's entry check -> host[8] (synchronises the device); reset != 0 zeroes them after
extern "C" int pn2_lab_stale_counters(unsigned long long *host8, int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(pn2::pn2_lab_stale), sizeof(unsigned long long) * 8);
    if (e != hipSuccess) return (int)e;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(pn2::pn2_lab_stale), z, sizeof(z));
    }
    return (int)e;
}
extern "C" int pn2_lab_stale_samples(unsigned long long *host256)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(host256, HIP_SYMBOL(pn2::pn2_lab_samples), sizeof(unsigned long long) * 256);
}
// lab build only: generation 0 clears the workspace with a kernel of the library's own instead of a memset node
extern "C" void pn2_lab_clear_with_kernel(int on) { pn2::g_lab_clear_with_kernel = on; }
#endif
