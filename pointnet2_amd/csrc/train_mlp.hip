// train_mlp.hip -- TRAINING mode of the shared MLPs of a set-abstraction / feature-propagation level on the matrix
// cores (second half). gfx950.
//
// Reference: utils/pointnet_util.py:113-127 (SA: 3 x [conv2d 1x1 + batch_norm + ReLU], reduce_max over nsample),
// :222-226 (FP: the same stack without the pool), tf_util.py:512-531 (batch moments, eps 1e-3, moving averages),
// train.py:96-104 (bn_decay), train.py:188 (is_training = True: the reference's main mode).
//
// Why this is not the inference kernel with one more flag. Batch statistics couple ALL rows of a level: layer l's
// normalisation needs the moments of z_l over every row before layer l+1 can start, and the backward pass has the
// same coupling in the other direction (dz = s dy - c0 - c1 z with c0, c1 from two per-channel sums over all rows).
// So a level is a SEQUENCE of passes with a per-channel reduction between them, and each pass is one GEMM over
// the rows with everything elementwise folded into its prologue and epilogue. The pre-norm tensors z_l are the only
// activations kept in HBM (plain row-major (rows, C) fp32, the caller's buffers); h_l = relu(a z + c), the
// grouped input, dz, the ReLU masks and the normalised tensors never exist in memory.
//
// One GEMM core serves every pass (tl_gemm_kernel): a wave owns 32 rows and NS 32-column output tiles,
//     D (32 rows x 32 cols) += A (32 rows x 16 k) . B (16 k x 32 cols)      v_mfma_f32_32x32x16_bf16
//   A operand = the wave's rows: lane l supplies row l & 31, contraction indices 32u + 16e + 8(l >> 5) + j, j = 0..7,
//               i.e. 32 contiguous bytes of a row-major row -- read straight from HBM (two dwordx4), transformed by
//               the pass's prologue, split into three bf16 levels (sa_mlp_common.h: fp32 products as six bf16 terms);
//   B operand = weights, packed ON THE DEVICE each step (they change with every optimiser step) into the same
//               three-level operand layout, staged through LDS (all k tiles resident when they fit, else a
//               double-buffered stream of one k tile per stage, one s_barrier per stage);
//   C/D: lane l holds column l & 31 for the 16 rows 8(v >> 2) + 4(l >> 5) + (v & 3): per-channel work of the epilogue
//        (bias, batch moments, max / min pool, ReLU mask of the layer below, batch-norm backward sums) is lane-local,
//        and a row's 32 columns leave as one 128-byte store.
// The weight gradient contracts over ROWS (tl_wgrad_kernel): both operands are read with lane = channel, slots = rows
// (coalesced 128-byte row segments), each wave accumulates its share of the rows into a register-resident dW slab,
// and a two-stage reduction sums the waves' slabs (fp64 in the last stage) in a fixed order: the result does not
// depend on timing.
#include "sa_mlp_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace pn2 {

constexpr int kTlThreads = 512;          // GEMM workgroup: 8 waves, one 32-row item each per round
constexpr int kTlWaves = kTlThreads / 64;
constexpr int kPairVec = kPairWords / 4; // 16-byte vectors of one 32x32 weight tile pair
constexpr int kMaxParts = 256;           // rows of a per-channel partial-sum array (one per row workgroup)

enum { A_PLAIN = 0, A_GATHER = 1, A_RELU = 2, A_DZ = 3, A_DZ_POOL = 4, A_FILL = 5 };
enum { E_STORE = 0, E_POOL = 1, E_MASK = 2, E_PLAIN = 3 };

// ---- per-channel finalisations, folded into the launch that produces their partial sums --------------------------------------
// Between two passes of a level stands a reduction over ALL rows: the workgroups of a pass leave one partial row each, and a
// 5 us launch (tl_bn_finalize_kernel / tl_bn_backward_finalize_kernel) turns the rows into the next pass's coefficients. A
// level of a few thousand rows is ~25 launches of which a third are such 5 us finalisations (profiles/r04: sem_seg SA4 forward
// 89 us in 9 launches, 24 of them in pack / finalise launches). TlFin folds the finalisation into the producer: every workgroup
// publishes its partial row (device-scope release), takes a ticket, and the workgroup that draws the LAST ticket -- all rows
// are then visible to it (device-scope acquire) -- does the finalisation before it exits. No workgroup ever waits for another:
// nothing can hang. The sums are added in a fixed order (J contiguous chunks of the partial rows, each in ascending order, the
// chunk sums in ascending order), so results do not depend on which workgroup comes last. Tickets live in the caller's workspace
// and are zeroed by the direction's first launch (tl_pack_kernel).
struct TlFin {
    unsigned *ticket;           // nullptr: not folded (the caller launches the finalisation kernel)
    unsigned total;             // workgroups that publish a partial row (all of them take a ticket)
    int mode;                   // 1: batch moments -> (mean, invstd, a, c) + running statistics; 2: (sum dy, sum dy z) -> grad_gamma, grad_beta, dz coefficients
    int nparts, N;
    const double *stats;        // (nparts, 2, N) partial rows -- written by THIS launch, read after the ticket
    double count;
    const float *gamma, *beta, *bias;
    float *running_mean, *running_var;
    float momentum, eps;
    int var_biased;
    float *save;                // mode 1: written (4, N); mode 2: read
    float *grad_gamma, *grad_beta, *coef;
    int accumulate;
};

// batch moments of one channel -> (mean, invstd, a, c), running statistics (torch.nn.BatchNorm semantics: unbiased variance in
// the average unless var_biased -- tf.contrib.layers.batch_norm, tf_util.py:512-531, averages the biased one)
__device__ __forceinline__ void tl_bn_finalize_channel(int c, int N, double s1, double s2, double count, const float *gamma,
                                                       const float *beta, float *running_mean, float *running_var, float momentum,
                                                       float eps, float *save, const float *bias, int var_biased)
{
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * invstd;
    save[c] = (float)mean;
    save[N + c] = (float)invstd;
    save[2 * N + c] = (float)a;
    save[3 * N + c] = (float)((double)beta[c] - a * mean);
    // the stored pre-norm tensor is h W WITHOUT the conv bias (see pn2_mlp_train_forward): the layer's batch mean is mean + b
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (mean + (bias ? (double)bias[c] : 0.0)));
    if (running_var) {
        const double bv = (var_biased || count <= 1.0) ? var : var * count / (count - 1.0);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * bv);
    }
}

// (sum dy, sum dy z) of one channel -> grad_gamma, grad_beta and the coefficients of dz = s dy - c0 - c1 z
__device__ __forceinline__ void tl_bn_backward_finalize_channel(int c, int N, double s1, double s2, double count, const float *gamma,
                                                                const float *save, float *grad_gamma, float *grad_beta, float *coef,
                                                                int accumulate)
{
    const double mean = save[c], invstd = save[N + c];
    const double dbeta = s1, dgamma = (s2 - mean * s1) * invstd;
    const double s = (double)gamma[c] * invstd;
    const double c1 = s * dgamma * invstd / count;
    const double c0 = s * dbeta / count - c1 * mean;
    if (grad_gamma) grad_gamma[c] = accumulate ? __fadd_rn(grad_gamma[c], (float)dgamma) : (float)dgamma;
    if (grad_beta) grad_beta[c] = accumulate ? __fadd_rn(grad_beta[c], (float)dbeta) : (float)dbeta;
    coef[c] = (float)s;
    coef[N + c] = (float)c0;
    coef[2 * N + c] = (float)c1;
}

constexpr size_t kFinLds = 16 * 512 + 64;          // LDS the tail needs at 512 threads: two doubles per thread + the flag
constexpr int kFinTickets = 16;                    // one counter per layer of a direction

// Device-scope traffic of the hand-off WITHOUT cache-wide fences. A release fence at agent scope is a write-back of the whole
// L2 of the XCD (buffer_wbl2) and an acquire fence invalidates it: with one such pair per workgroup the folded form measured
// 20-35 us SLOWER per pass than the separate launch (the passes' own outputs are tens of megabytes of dirty lines, and the
// invalidate costs the workgroups still running their weights). Instead the partial rows are written with write-through
// stores (agent-scope relaxed atomic stores: sc1), the storing threads wait for the write acknowledgements (s_waitcnt
// vmcnt(0)) before the workgroup takes its ticket (agent-scope relaxed atomic add, performed at the memory side), and the last
// workgroup reads the rows with agent-scope relaxed atomic loads, which bypass the non-coherent copies of its own L2 -- the
// hand-off form of sa_fused.hip's sample granules, with the ticket in place of the tag.
typedef unsigned long long __attribute__((address_space(1))) tl_gu64;
typedef unsigned __attribute__((address_space(1))) tl_gu32;

__device__ __forceinline__ void tl_fin_store(double *p, double v)     // a partial sum another workgroup of this launch will read
{
    __hip_atomic_store((tl_gu64 *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double tl_fin_load(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const tl_gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The tail of a producing workgroup of NT threads (every thread of every workgroup of the launch calls it, after its
// tl_fin_store()s of the partial row). `lds`: >= 16 NT + 16 bytes of the workgroup's LDS that nobody uses any more.
template <int NT>
__device__ __forceinline__ void tl_fin_tail(const TlFin &f, char *lds)
{
    const int tid = threadIdx.x;
    unsigned *flag = reinterpret_cast<unsigned *>(lds);
    double *sh = reinterpret_cast<double *>(lds + 16);              // [2][NT]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this thread's write-through stores are acknowledged ...
    __syncthreads();                                                // ... and so are every other thread's of this workgroup
    if (tid == 0)
        *flag = (__hip_atomic_fetch_add((tl_gu32 *)f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.total - 1u) ? 1u : 0u;
    __syncthreads();
    if (*flag == 0u) return;                                        // workgroup-uniform
    int cb = 32;                                                    // channels per sweep (a power of two), J = NT / cb threads each
    while (cb < f.N && cb < NT) cb <<= 1;
    const int J = NT / cb, c = tid & (cb - 1), j = tid / cb;
    const int chunk = (f.nparts + J - 1) / J;
    const int N = f.N;
    const double *st = f.stats;
    for (int c0 = 0; c0 < N; c0 += cb) {
        const int ch = c0 + c;
        double s1 = 0.0, s2 = 0.0;
        if (ch < N) {
            int q = j * chunk;
            const int q1 = min(f.nparts, q + chunk);
            const double *src = st + (size_t)q * 2 * N + ch;
            for (; q + 8 <= q1; q += 8, src += (size_t)16 * N) {    // sixteen independent loads in flight, the sums in order
                double a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a[u] = tl_fin_load(src + (size_t)(2 * u) * N); b[u] = tl_fin_load(src + (size_t)(2 * u + 1) * N); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += b[u]; }
            }
            for (; q < q1; ++q, src += (size_t)2 * N) { s1 += tl_fin_load(src); s2 += tl_fin_load(src + N); }
        }
        sh[tid] = s1;
        sh[NT + tid] = s2;
        __syncthreads();
        if (j == 0 && ch < N) {
            double t1 = sh[c], t2 = sh[NT + c];
            for (int jj = 1; jj < J; ++jj) { t1 += sh[jj * cb + c]; t2 += sh[NT + jj * cb + c]; }
            if (f.mode == 1)
                tl_bn_finalize_channel(ch, N, t1, t2, f.count, f.gamma, f.beta, f.running_mean, f.running_var, f.momentum, f.eps, f.save,
                                       f.bias, f.var_biased);
            else
                tl_bn_backward_finalize_channel(ch, N, t1, t2, f.count, f.gamma, f.save, f.grad_gamma, f.grad_beta, f.coef, f.accumulate);
        }
        __syncthreads();
    }
}

struct TlGather {
    int n, m, nsample, cfeat, xyz_off, feat_off;
    const float *xyz, *new_xyz, *points;
    const int *idx;
};

struct TlGemm {
    long long rows;
    int K, N;                   // contraction width = pitch of A; output width = pitch of out / zprev
    int tk;                     // 32-wide k tiles
    int resident;               // every k tile's weights stay in LDS
    // A operand
    const float *A;             // A_PLAIN: x; A_RELU: z of the layer below; A_DZ / A_DZ_POOL: z of this layer
    const float *G;             // A_DZ: dy (rows, K); A_DZ_POOL: gq (groups, K)
    const int *argsel;          // A_DZ_POOL: (groups, K)
    const float *p0, *p1, *p2;  // A_RELU: a, c; A_DZ*: s, c0, c1   (K floats each)
    int group_rows;             // A_DZ_POOL, A_FILL
    // A_FILL (pooled top layer without its pre-norm tensor, see pn2_mlp_train_backward): k tiles [0, tk0) are the routed
    // gradient s dy -- (argsel == sample) ? p0[k] * G[group][k] : 0, K0 channels -- and k tiles [tk0, tk) are
    // h = relu(q0 * A2 + q1) of the layer below (K1 channels, pitch K1)
    int tk0, K0, K1;
    const float *A2, *q0, *q1;
    TlGather g;                 // A_GATHER
    const u32x4 *wpacked;       // [slab][k tile][NS] tile pairs
    const float *bias;          // (N) or nullptr
    // epilogue
    int emode;
    float *out;                 // E_STORE / E_POOL: z (rows, N); E_MASK: dy of the layer below (rows, N); E_PLAIN: see col0
    int out_pitch, col0, col1;  // E_PLAIN: columns [col0, col1) go to out[row * out_pitch + col - col0]
    double *stats;              // (2, N): E_STORE / E_POOL: sum z, sum z^2; E_MASK: sum dy, sum dy * zprev
    const float *zprev, *ea, *ec;   // E_MASK: pre-norm tensor of the layer below (rows, N) and its (a, c)
    float *pmax;                // E_POOL partials (rows / prow, N): the extremum the pool will select -- the max of z where
    int *pamax;                 // gamma >= 0, the min where gamma < 0 (batch norm + ReLU are monotone per channel) -- and its row
    const float *pool_gamma;    // E_POOL: (N) batch-norm scale of this layer (its sign picks max or min)
    int prow;                   // 32 or 16
    int nt;                     // streaming (non-temporal) stores: outputs that do not fit the 256 MB Infinity Cache anyway
    int lab;                    // lab builds of the timing study only (PN2_TL_LAB): 1 = no stores, 2 = no statistics; 0 in production
    TlFin fin;                  // the per-channel finalisation of `stats`, by the workgroup that finishes last (fin.ticket != nullptr)
};

// ---- A operand: load + prologue. Register v = 8e + j of lane (row s, half hl) <-> channel 32u + 16e + 8hl + j ------------
struct ARaw { f32x16 a, g; int4 sel[4]; };
struct RowCtx { long long grp; int sample, pt; long long cloud; };      // of the lane's row (gather / pooled passes)

template <int AMODE>
__device__ __forceinline__ RowCtx tl_row_ctx(const TlGemm &p, long long row, bool active)
{
    RowCtx c = {0, 0, 0, 0};
    if (!active) return c;
    if (AMODE == A_GATHER) {
        c.grp = row / p.g.nsample;
        c.sample = (int)(row - c.grp * p.g.nsample);
        c.cloud = c.grp / p.g.m;
        c.pt = p.g.idx ? p.g.idx[row] : c.sample;
    } else if (AMODE == A_DZ_POOL || AMODE == A_FILL) {
        c.grp = row / p.group_rows;
        c.sample = (int)(row - c.grp * p.group_rows);
    }
    return c;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Buffer addressing (SRSRC): a wave-uniform 128-bit descriptor in SGPRs + ONE per-lane byte offset in a VGPR + a uniform
// byte offset in an SGPR per instruction. A lane's eight row loads then share one address register (flat addressing
// needs a 64-bit VGPR pair per load in flight: 100+ registers of addresses in the kernels below).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    // The descriptor's inputs go through readfirstlane: they ARE wave-uniform (kernel arguments, block and wave numbers),
    // but hipcc cannot always prove it -- anything that met a value derived from threadIdx in a select or a phi is
    // "divergent" to it -- and an unproven descriptor gets a waterfall loop (4 x v_readfirstlane, compare, saveexec,
    // branch) around EVERY buffer instruction: 487 readfirstlanes per two blocks in the weight-gradient kernel.
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((uintptr_t)(((unsigned long long)hi << 32) | lo)), 0, (int)nb,
                                             0x00020000);
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }       // a value known to be wave-uniform
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 bload4(rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ int4 bload4i(rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_int4((int)v[0], (int)v[1], (int)v[2], (int)v[3]);
}
template <bool NT>
__device__ __forceinline__ void bstore(float x, rsrc_t r, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, NT ? 2 : 0);     // aux bit 1 = nt (streaming store)
}

template <int AMODE>
__device__ __forceinline__ void tl_load_raw(const TlGemm &p, long long row0, long long row, const RowCtx &rc, int u, int hl,
                                            bool active, ARaw &r)
{
#pragma unroll
    for (int v = 0; v < 16; ++v) { r.a[v] = 0.0f; r.g[v] = 0.0f; }
    if (!active) return;
    if (AMODE == A_GATHER) {
        const TlGather &g = p.g;
        const float *px = g.xyz + ((size_t)rc.cloud * g.n + rc.pt) * 3;
        const float *pf = g.points ? g.points + ((size_t)rc.cloud * g.n + rc.pt) * g.cfeat : nullptr;
        const float *pc = g.new_xyz ? g.new_xyz + rc.grp * 3 : nullptr;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * u + 16 * e + 8 * hl + j;
                float val = 0.0f;
                const int kx = k - g.xyz_off, kf = k - g.feat_off;
                if (kx >= 0 && kx < 3) val = pc ? __fsub_rn(px[kx], pc[kx]) : px[kx];      // pointnet_util.py:46
                else if (kf >= 0 && kf < g.cfeat) val = pf[kf];
                r.a[8 * e + j] = val;
            }
        return;
    }
    const int s = (int)(row - row0);
    if (AMODE == A_FILL) {
        if (u < p.tk0) {                                           // (groups, K0) routed gradient + its sample numbers
            const float *pg = p.G + (size_t)rc.grp * p.K0;
            const int *ps = p.argsel + (size_t)rc.grp * p.K0;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = 32 * u + 16 * e + 8 * hl + 4 * q;
                    int4 s4 = {-1, -1, -1, -1};
                    if (k < p.K0) {
                        const float4 t = ld4(pg + k);
                        r.a[8 * e + 4 * q] = t.x; r.a[8 * e + 4 * q + 1] = t.y; r.a[8 * e + 4 * q + 2] = t.z; r.a[8 * e + 4 * q + 3] = t.w;
                        s4 = *reinterpret_cast<const int4 *>(ps + k);
                    }
                    r.sel[2 * e + q] = s4;
                }
        } else {                                                   // rows of the layer below
            const rsrc_t r2 = make_rsrc(p.A2 + (size_t)row0 * p.K1, 32u * (unsigned)p.K1 * 4u);
            const int voff2 = (s * p.K1 + 8 * hl) * 4, soff2 = (u - p.tk0) * 128;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = 32 * (u - p.tk0) + 16 * e + 8 * hl + 4 * q;
                    if (k < p.K1) {
                        const float4 t = bload4(r2, voff2 + (16 * e + 4 * q) * 4, soff2);
                        r.a[8 * e + 4 * q] = t.x; r.a[8 * e + 4 * q + 1] = t.y; r.a[8 * e + 4 * q + 2] = t.z; r.a[8 * e + 4 * q + 3] = t.w;
                    }
                }
        }
        return;
    }
    // rows of the item: descriptor at the item's first row, lane offset = its row and half, uniform offset = the k tile
    const rsrc_t ra = make_rsrc(p.A + (size_t)row0 * p.K, 32u * (unsigned)p.K * 4u);
    const int voff = (s * p.K + 8 * hl) * 4, soff = u * 128;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = 32 * u + 16 * e + 8 * hl + 4 * q;
            if (k < p.K) {
                const float4 t = bload4(ra, voff + (16 * e + 4 * q) * 4, soff);
                r.a[8 * e + 4 * q] = t.x; r.a[8 * e + 4 * q + 1] = t.y; r.a[8 * e + 4 * q + 2] = t.z; r.a[8 * e + 4 * q + 3] = t.w;
            }
        }
    if (AMODE == A_DZ) {
        const rsrc_t rg = make_rsrc(p.G + (size_t)row0 * p.K, 32u * (unsigned)p.K * 4u);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = 32 * u + 16 * e + 8 * hl + 4 * q;
                if (k < p.K) {
                    const float4 t = bload4(rg, voff + (16 * e + 4 * q) * 4, soff);
                    r.g[8 * e + 4 * q] = t.x; r.g[8 * e + 4 * q + 1] = t.y; r.g[8 * e + 4 * q + 2] = t.z; r.g[8 * e + 4 * q + 3] = t.w;
                }
            }
    }
    if (AMODE == A_DZ_POOL) {
        const float *pg = p.G + (size_t)rc.grp * p.K;
        const int *ps = p.argsel + (size_t)rc.grp * p.K;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = 32 * u + 16 * e + 8 * hl + 4 * q;
                int4 s4 = {-1, -1, -1, -1};
                if (k < p.K) {
                    const float4 t = ld4(pg + k);
                    r.g[8 * e + 4 * q] = t.x; r.g[8 * e + 4 * q + 1] = t.y; r.g[8 * e + 4 * q + 2] = t.z; r.g[8 * e + 4 * q + 3] = t.w;
                    s4 = *reinterpret_cast<const int4 *>(ps + k);
                }
                r.sel[2 * e + q] = s4;
            }
    }
}

// the pass's prologue on the 16 values of one k tile; lp*: the per-channel parameters in LDS (zero beyond K)
template <int AMODE>
__device__ __forceinline__ f32x16 tl_finish(const ARaw &r, const RowCtx &rc, int u, int tk0, int hl, const float *lp0,
                                            const float *lp1, const float *lp2)
{
    if (AMODE == A_PLAIN || AMODE == A_GATHER) return r.a;
    f32x16 x;
    const int sample = rc.sample;
    if (AMODE == A_FILL) {                                         // lp0: s (fill tiles) / a (rows of the layer below); lp1: c
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = 32 * u + 16 * e + 8 * hl + 4 * q;
                const float4 c0 = ld4(lp0 + k), c1 = ld4(lp1 + k);
                const float a0[4] = {c0.x, c0.y, c0.z, c0.w}, a1[4] = {c1.x, c1.y, c1.z, c1.w};
                const int4 s4 = r.sel[2 * e + q];
                const int sl[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = 8 * e + 4 * q + i;
                    if (u < tk0) x[v] = sl[i] == sample ? __fmul_rn(a0[i], r.a[v]) : 0.0f;        // the pool routes dy to ONE sample
                    else x[v] = vmax(__fadd_rn(__fmul_rn(a0[i], r.a[v]), a1[i]), 0.0f);
                }
            }
        return x;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = 32 * u + 16 * e + 8 * hl + 4 * q;
            const float4 c0 = ld4(lp0 + k), c1 = ld4(lp1 + k);
            const float a0[4] = {c0.x, c0.y, c0.z, c0.w}, a1[4] = {c1.x, c1.y, c1.z, c1.w};
            if (AMODE == A_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = 8 * e + 4 * q + i;
                    x[v] = vmax(__fadd_rn(__fmul_rn(a0[i], r.a[v]), a1[i]), 0.0f);        // h = relu(a z + c)
                }
            } else {
                const float4 c2 = ld4(lp2 + k);
                const float a2[4] = {c2.x, c2.y, c2.z, c2.w};
                int sl[4] = {0, 0, 0, 0};
                if (AMODE == A_DZ_POOL) {
                    const int4 s4 = r.sel[2 * e + q];
                    sl[0] = s4.x; sl[1] = s4.y; sl[2] = s4.z; sl[3] = s4.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = 8 * e + 4 * q + i;
                    float dy = r.g[v];
                    if (AMODE == A_DZ_POOL) dy = sl[i] == sample ? dy : 0.0f;             // the pool routes dy to ONE sample
                    x[v] = __fsub_rn(__fsub_rn(__fmul_rn(a0[i], dy), a1[i]), __fmul_rn(a2[i], r.a[v]));   // s dy - c0 - c1 z
                }
            }
        }
    return x;
}

template <int NS, int AMODE>
__global__ __launch_bounds__(kTlThreads) void tl_gemm_kernel(const TlGemm p)
{
#define PN2_BX blockIdx.x
#define PN2_BY blockIdx.y
#define PN2_GX gridDim.x
#include "tl_gemm_body.inc"
#undef PN2_BX
#undef PN2_BY
#undef PN2_GX
}

// ---- weights -> three-level bf16 operand tiles, on the device ---------------------------------------------------------
// value for K16 step e, level, lane l, slot j of pair (slab, u, t) = level of W[32u + 16e + 8(l >> 5) + j][32(slab NS + t) + (l & 31)]
struct TlPackJob { const float *w; long long sk, sn; int K, N, tk, ns, slabs; u32x4 *out; };
struct TlPackJobs {                                               // one launch packs every layer of a level (blockIdx.y = layer)
    TlPackJob j[8];
    float *ident; int ident_c;                                    // ... and writes the identity coefficients (1, 0, 0) x ident_c, if wanted
    unsigned *tickets;                                            // ... and zeroes the tickets of the direction's folded finalisations (TlFin)
};

__global__ __launch_bounds__(256) void tl_pack_kernel(const TlPackJobs jobs)
{
    if (jobs.ident && blockIdx.x == 0 && blockIdx.y == 0)         // dz = 1 * g - 0 - 0 * z (layer 1 per point: S enters as it is)
        for (int i = threadIdx.x; i < 3 * jobs.ident_c; i += 256) jobs.ident[i] = i < jobs.ident_c ? 1.0f : 0.0f;
    if (jobs.tickets && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < kFinTickets) jobs.tickets[threadIdx.x] = 0u;
    const TlPackJob &q = jobs.j[blockIdx.y];
    const long long total = (long long)q.slabs * q.tk * q.ns * 128;     // one thread per (pair, e, lane)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63), e = (int)((i >> 6) & 1);
        const long long pair = i >> 7;
        const int t = (int)(pair % q.ns), u = (int)((pair / q.ns) % q.tk), slab = (int)(pair / ((long long)q.ns * q.tk));
        const int n = (slab * q.ns + t) * 32 + (lane & 31);
        f32x16 x;
#pragma unroll
        for (int v = 0; v < 16; ++v) x[v] = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * u + 16 * e + 8 * (lane >> 5) + j;
            x[j] = (k < q.K && n < q.N) ? q.w[k * q.sk + n * q.sn] : 0.0f;
        }
        const ActSplit sp = split_act(x);
        u32x4 *o = q.out + pair * kPairVec + (size_t)e * 192 + lane;
        o[0] = sp.p[0][0];
        o[64] = sp.p[0][1];
        o[128] = sp.p[0][2];
    }
}

// ---- per-channel finalisation kernels (one thread per channel) -----------------------------------------------------------
// batch moments -> (mean, invstd, a, c), running statistics (torch.nn.BatchNorm semantics: unbiased variance in the average)
// the per-channel sums arrive as `nparts` partial rows; a block of 256 threads owns 8 channels and adds the rows 32 at a time
__device__ __forceinline__ void tl_sum_parts(const double *__restrict__ stats, int nparts, int N, double &s1, double &s2)
{
    __shared__ double sh[2][32][8];
    const int g = threadIdx.x >> 3, cl = threadIdx.x & 7, c = blockIdx.x * 8 + cl;
    double a = 0.0, b = 0.0;
    if (c < N)
        for (int q = g; q < nparts; q += 32) { a += stats[((size_t)q * 2) * N + c]; b += stats[((size_t)q * 2 + 1) * N + c]; }
    sh[0][g][cl] = a;
    sh[1][g][cl] = b;
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { s1 += sh[0][i][cl]; s2 += sh[1][i][cl]; }
    }
}

__global__ __launch_bounds__(256) void tl_bn_finalize_kernel(const double *__restrict__ stats, int nparts, int N, double count,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *running_mean, float *running_var, float momentum, float eps,
                                                             float *__restrict__ save, const float *__restrict__ bias, int var_biased)
{
    double s1, s2;
    tl_sum_parts(stats, nparts, N, s1, s2);
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    if (threadIdx.x >= 8 || c >= N) return;
    tl_bn_finalize_channel(c, N, s1, s2, count, gamma, beta, running_mean, running_var, momentum, eps, save, bias, var_biased);
}

// (sum dy, sum dy z) -> grad_gamma, grad_beta and the coefficients of dz = s dy - c0 - c1 z
__global__ __launch_bounds__(256) void tl_bn_backward_finalize_kernel(const double *__restrict__ stats, int nparts, int N,
                                                                      double count, const float *__restrict__ gamma,
                                                                      const float *__restrict__ save, float *__restrict__ grad_gamma,
                                                                      float *__restrict__ grad_beta, float *__restrict__ coef, int accumulate)
{
    double s1, s2;
    tl_sum_parts(stats, nparts, N, s1, s2);
    const int c = blockIdx.x * 8 + (threadIdx.x & 7);
    if (threadIdx.x >= 8 || c >= N) return;
    tl_bn_backward_finalize_channel(c, N, s1, s2, count, gamma, save, grad_gamma, grad_beta, coef, accumulate);
}

// pool: partial extrema of z (the max where gamma >= 0, else the min) -> out = relu(a zsel + c), the sample the gradient flows to, zsel
__global__ void tl_pool_finalize_kernel(long long groups, int N, int parts, int prow, const float *__restrict__ pmax,
                                        const int *__restrict__ pamax, const float *__restrict__ gamma,
                                        const float *__restrict__ save,
                                        float *__restrict__ out, int *__restrict__ argsel, float *__restrict__ zsel)
{
    const long long total = groups * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long g = i / N;
        const int c = (int)(i - g * N);
        const float a = save[2 * N + c], cc = save[3 * N + c];
        const bool up = gamma[c] >= 0.0f;                          // the rule of the GEMM epilogue that wrote the partials
        float best = 0.0f;
        int arg = 0;
        for (int q = 0; q < parts; ++q) {
            const size_t o = (size_t)(g * parts + q) * N + c;
            const float v = pmax[o];
            const int r = pamax[o] + q * prow;
            if (q == 0 || (up ? v > best : v < best)) { best = v; arg = r; }
        }
        out[i] = vmax(__fadd_rn(__fmul_rn(a, best), cc), 0.0f);
        argsel[i] = arg;
        zsel[i] = best;
    }
}

// pooled top layer: gq = grad_out . [out > 0]; sums of dy and dy z over all rows = over the selected entries
__global__ __launch_bounds__(256) void tl_pool_grad_kernel(long long groups, int N, const float *__restrict__ out,
                                                           const float *__restrict__ gout, const float *__restrict__ zsel,
                                                           float *__restrict__ gq, double *__restrict__ stats)
{
    // thread (x = column within a 64-column strip, y = row lane): column sums over a strided set of groups
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    double s1 = 0.0, s2 = 0.0;
    if (c < N) {
        // four groups per trip: the three loads of each are independent of the sums, and one group at a time left the loop a
        // chain of memory latencies (36 us for 64 MB at the metric shape)
        const long long gstep = (long long)gridDim.y * 4;
        long long g = (long long)blockIdx.y * 4 + ry;
        for (; g + 3 * gstep < groups; g += 4 * gstep) {
            float o4[4], g4[4], z4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = (size_t)(g + u * gstep) * N + c;
                o4[u] = out[o]; g4[u] = gout[o]; z4[u] = zsel[o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float q = o4[u] > 0.0f ? g4[u] : 0.0f;
                gq[(size_t)(g + u * gstep) * N + c] = q;
                s1 += (double)q;
                s2 += (double)q * (double)z4[u];
            }
        }
        for (; g < groups; g += gstep) {
            const size_t o = (size_t)g * N + c;
            const float q = out[o] > 0.0f ? gout[o] : 0.0f;
            gq[o] = q;
            s1 += (double)q;
            s2 += (double)q * (double)zsel[o];
        }
    }
    __shared__ double sh[2][4][64];
    sh[0][ry][threadIdx.x & 63] = s1;
    sh[1][ry][threadIdx.x & 63] = s2;
    __syncthreads();
    if (ry == 0 && c < N) {
        const int x = threadIdx.x & 63;
        stats[((size_t)blockIdx.y * 2) * N + c] = sh[0][0][x] + sh[0][1][x] + sh[0][2][x] + sh[0][3][x];
        stats[((size_t)blockIdx.y * 2 + 1) * N + c] = sh[1][0][x] + sh[1][1][x] + sh[1][2][x] + sh[1][3][x];
    }
}

// unpooled top layer (FP levels): out = relu(a z + c)
__global__ __launch_bounds__(256) void tl_apply_kernel(long long total4, int N, const float *__restrict__ z,
                                                       const float *__restrict__ save, float *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = (int)((i * 4) % N);
        const float4 zz = ld4(z + i * 4), a = ld4(save + 2 * N + c), cc = ld4(save + 3 * N + c);
        float4 o;
        o.x = vmax(__fadd_rn(__fmul_rn(a.x, zz.x), cc.x), 0.0f);
        o.y = vmax(__fadd_rn(__fmul_rn(a.y, zz.y), cc.y), 0.0f);
        o.z = vmax(__fadd_rn(__fmul_rn(a.z, zz.z), cc.z), 0.0f);
        o.w = vmax(__fadd_rn(__fmul_rn(a.w, zz.w), cc.w), 0.0f);
        *reinterpret_cast<float4 *>(out + i * 4) = o;
    }
}

// unpooled top layer backward: dy = grad_out . [out > 0] (rows, N) + its two column sums
__global__ __launch_bounds__(256) void tl_top_grad_kernel(long long rows, int N, const float *__restrict__ out,
                                                          const float *__restrict__ gout, const float *__restrict__ z,
                                                          float *__restrict__ dy, double *__restrict__ stats)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    double s1 = 0.0, s2 = 0.0;
    if (c < N) {
        // four rows per trip (independent loads in flight; one row at a time was a chain of memory latencies, as in
        // tl_pool_grad_kernel): the sums keep their order
        const long long rstep = (long long)gridDim.y * 4;
        long long r = (long long)blockIdx.y * 4 + ry;
        for (; r + 3 * rstep < rows; r += 4 * rstep) {
            float o4[4], g4[4], z4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = (size_t)(r + u * rstep) * N + c;
                o4[u] = out[o]; g4[u] = gout[o]; z4[u] = z[o];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float q = o4[u] > 0.0f ? g4[u] : 0.0f;
                dy[(size_t)(r + u * rstep) * N + c] = q;
                s1 += (double)q;
                s2 += (double)q * (double)z4[u];
            }
        }
        for (; r < rows; r += rstep) {
            const size_t o = (size_t)r * N + c;
            const float q = out[o] > 0.0f ? gout[o] : 0.0f;
            dy[o] = q;
            s1 += (double)q;
            s2 += (double)q * (double)z[o];
        }
    }
    __shared__ double sh[2][4][64];
    sh[0][ry][threadIdx.x & 63] = s1;
    sh[1][ry][threadIdx.x & 63] = s2;
    __syncthreads();
    if (ry == 0 && c < N) {
        const int x = threadIdx.x & 63;
        stats[((size_t)blockIdx.y * 2) * N + c] = sh[0][0][x] + sh[0][1][x] + sh[0][2][x] + sh[0][3][x];
        stats[((size_t)blockIdx.y * 2 + 1) * N + c] = sh[1][0][x] + sh[1][1][x] + sh[1][2][x] + sh[1][3][x];
    }
}

// ---- weight gradient: dW (KI x NO) = h^T dz, contraction over the rows -----------------------------------------------------
struct TlWgrad {
    long long rows;
    int amode;                  // A_PLAIN / A_GATHER / A_RELU: how h (rows, KI) is formed
    int KI;
    const float *A, *pa, *pc;
    TlGather g;
    int dmode;                  // A_DZ / A_DZ_POOL / A_FILL
    int NO;                     // A_FILL: tf * 32 + tx * 32 + 32 columns: [routed gradient (NF) | h again (KI) | ones], see below
    int tf, NF;                 // A_FILL: tiles / channels of the routed-gradient block
    int xshare;                 // A_FILL: every tile of h is in the slab's image, the "h again" tiles are read from there
    const float *Z, *G;
    const int *argsel;
    const float *coef;          // (3, NO): s, c0, c1
    int group_rows;
    float *partial;             // [slab][workgroup][tus * tts tiles][1024]
    size_t partial_cap;         // host side: bytes planned for `partial` (0 = unchecked); a launch whose slabs need more is refused
    int tus, tts, tslabs;       // tiles of h / of dz per slab; slabs along dz
    // ---- the layer's DATA gradient in the same pass (template flag DY; one slab only): dy_{l-1} = (second operand) . Wt,
    // contraction over the k tiles the block image already holds, see "One pass per layer" below
    const u32x4 *dy_w;          // packed operand tiles [k tile][dy_nt] of Wt (tl_pack_kernel, ns = dy_nt, one slab)
    int dy_tk, dy_nt;           // k tiles (32 channels) of the contraction / 32-column tiles of the output
    int dy_tf;                  // D_TOP with shared h tiles: k tiles >= dy_tf are image tiles (u - dy_tf), the others tus + u
    int dy_cols, dy_pitch;      // output columns / floats per row of dy_out and dy_zprev
    float *dy_out;              // (rows, dy_pitch)
    const float *dy_zprev, *dy_ea, *dy_ec;   // ReLU mask of the layer below: its pre-norm tensor and (a, c); nullptr: plain store
    const float *dy_bias;       // constant row added to every output row (D_TOP: -r) or nullptr
    double *dy_stats;           // (gridDim.x, 2, dy_pitch): sum dy, sum dy * zprev of this workgroup's rows, or nullptr
    int dy_nt_store;            // streaming stores
    int single;                 // ONE block image in LDS (two barriers per block) instead of two
    int dy_acopy;               // the dense second-operand units also write their fragments in the data gradient's own layout
    // Layer 1 of a level WITHOUT features below this layer (its input is the three centred coordinates x of a row): the
    // data gradient produced here is dy_1, and all that is wanted from it is dW_1 = x^T dz_1. With dz_1 = s dy_1 - c0 - c1 z_1
    // and z_1 = x W_1:   dW_1 = s (x^T dy_1) - c0 (x^T 1) - c1 ((x^T x) W_1)   -- the last two from nine moments of x, the
    // first accumulated HERE from the epilogue's registers. dy_1 is then never written and the pass over (dy_1, z_1) that
    // formed dW_1 (tl_l1_dz_kernel) disappears.
    const float4 *l1x;          // (rows) centred coordinates of every row, w = 0 (tl_l1_xrows_kernel) or nullptr
    double *l1a;                // (gridDim.x, 3, dy_pitch): sum over this workgroup's rows of x[k] * dy[.][col]
    int xr_off;                 // byte offset of the coordinate rows in LDS
    unsigned long long *timing; // lab builds (PN2_WG_TIMING): per-wave cycle counts of the block loop's phases, workgroup 0
};

// One UNIT of operand data = what one wave holds as the MFMA fragment of K16 step e of a 32-channel tile: lane (c, hl)
// <-> channel 32 tile + c, rows 16e + 8hl + j (j = 0..7) of the 32-row block. A wave loads a unit with dword loads whose
// 32 lanes cover 128 contiguous bytes of a row, applies the pass's prologue, splits into the three bf16 levels and writes
// three 16-byte fragments into the block's LDS image, from where EVERY wave of the workgroup reads the fragments of the
// output tiles it owns: operands cross the vector memory path once per workgroup.
//
// Two rules shaped this code (both measured):
//  * BRANCH-FREE loads. The units of a workgroup differ in kind (rows of h, rows of z and dy, routed gradient, nothing),
//    and a wait shared by paths with different numbers of loads in flight can only be vmcnt(0) -- with branches around the
//    loads the two-block prefetch drained at every block. Every unit of every wave therefore issues the same sequence of
//    buffer loads, and what a unit does not need points at an empty descriptor (out-of-range: returns 0, no memory access).
//  * Everything that does not depend on the block is computed ONCE per unit (WgUnit, before the block loop): these
//    kernels issue ~2500 instructions per 32-row block and wave, and were bound by that, not by memory.
enum { K_NONE = 0, K_H = 1, K_HGATHER = 2, K_DZ = 3, K_DZPOOL = 4, K_FILL = 5, K_ONES = 6 };
enum { D_DZ = 0, D_DZPOOL = 1, D_TOP = 2 };                       // second-operand class of the launch (template parameter)

struct WgRaw { float z[8], g[8]; float gq; int sel, off; };      // off: first row of the unit inside its group
struct WgUnit {
    int kind;                   // uniform
    int relu;                   // uniform: K_H rows go through relu(p0 z + p1) (else taken as they are)
    int e, tile;                // uniform: K16 step, tile of the block image
    const float *b1, *b2;       // uniform: row streams (nullptr: none)
    int pitch1;                 // uniform: floats per row of the row streams
    int voff;                   // lane: byte offset of its first row inside the block's rows (kWgOob: no such channel)
    const float *bq;            // uniform: per-group values (routed gradient / centroid)
    const int *bs;              // uniform: per-group sample numbers
    int nq, chq;                // uniform pitch / lane channel (-1: none) of the per-group streams
    // gather (layer 1 of an SA level): a lane's channel is a coordinate (kx) or a feature (kf) of the row's point; the two
    // tensors are read through two UNIFORM descriptors, the lane that needs neither / only one reads out of range there
    // (a per-lane descriptor would put a waterfall loop around every load)
    const float *bgx, *bgf;     // uniform: xyz, points
    int kx, kf, pitchf;         // lane: coordinate / feature number (-1: none); uniform: feature channels per point
    float p0, p1, p2;           // lane: per-channel parameters of the prologue
};
constexpr int kWgOob = (int)0xfffffff0u;                         // beyond every descriptor's num_records (<= 0x7fffffff)

__device__ __forceinline__ int bloadi(rsrc_t r, int voff, int soff)
{
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}

__device__ __forceinline__ WgUnit wg_plan_unit(const TlWgrad &p, int unit, int nunits, int us, int ts, int lane)
{
    WgUnit w;
    const int hl = lane >> 5, c = lane & 31;
    w.kind = K_NONE; w.relu = 0; w.e = unit & 1; w.tile = unit >> 1;
    w.b1 = nullptr; w.b2 = nullptr; w.bq = nullptr; w.bs = nullptr; w.bgx = nullptr; w.bgf = nullptr;
    w.pitch1 = 0; w.voff = kWgOob; w.nq = 0; w.chq = -1; w.kx = -1; w.kf = -1; w.pitchf = 0;
    w.p0 = 1.0f; w.p1 = 0.0f; w.p2 = 0.0f;
    if (unit >= nunits) return w;
    const int rin = 16 * w.e + 8 * hl, tx = (p.KI + 31) / 32;
    if (w.tile < p.tus) {                                          // the layer's input h
        const int ch = (us * p.tus + w.tile) * 32 + c;
        if (p.amode == A_GATHER) {
            const TlGather &g = p.g;
            const int kx = ch - g.xyz_off, kf = ch - g.feat_off;
            w.kind = K_HGATHER;
            w.bgx = g.xyz; w.bgf = g.points; w.pitchf = g.cfeat; w.bq = g.new_xyz; w.nq = 3;     // pointers: uniform choices only
            if (ch < p.KI && kx >= 0 && kx < 3) { w.kx = kx; w.chq = g.new_xyz ? kx : -1; }
            else if (ch < p.KI && kf >= 0 && kf < g.cfeat) w.kf = kf;
        } else {
            w.kind = K_H; w.b1 = p.A; w.pitch1 = p.KI; w.relu = p.amode == A_RELU;
            if (ch < p.KI) {
                w.voff = (rin * p.KI + ch) * 4;
                if (p.amode == A_RELU) { w.p0 = p.pa[ch]; w.p1 = p.pc[ch]; }
            }
        }
        return w;
    }
    const int tg = ts * p.tts + w.tile - p.tus;                    // tile of the second operand
    if (p.dmode == A_FILL) {                                       // [s dy routed to the pooled samples | h itself | ones]
        if (tg < p.tf) {
            const int ch = tg * 32 + c;
            w.kind = K_FILL; w.bq = p.G; w.bs = p.argsel; w.nq = p.NF;
            if (ch < p.NF) { w.chq = ch; w.p0 = p.coef[ch]; }
        } else if (tg < p.tf + tx) {
            if (!p.xshare) {
                const int ch = (tg - p.tf) * 32 + c;
                w.kind = K_H; w.b1 = p.A; w.pitch1 = p.KI; w.relu = 1;
                if (ch < p.KI) { w.voff = (rin * p.KI + ch) * 4; w.p0 = p.pa[ch]; w.p1 = p.pc[ch]; }
            }
        } else if (tg == p.tf + tx) {
            w.kind = K_ONES;
        }
        return w;
    }
    const int ch = tg * 32 + c;                                    // dz = s dy - c0 - c1 z
    w.kind = p.dmode == A_DZ_POOL ? K_DZPOOL : K_DZ;
    w.b1 = p.Z; w.pitch1 = p.NO;
    if (p.dmode == A_DZ_POOL) { w.bq = p.G; w.bs = p.argsel; w.nq = p.NO; } else w.b2 = p.G;
    if (ch < p.NO) {
        w.voff = (rin * p.NO + ch) * 4;
        if (p.dmode == A_DZ_POOL) w.chq = ch;
        w.p0 = p.coef[ch]; w.p1 = p.coef[p.NO + ch]; w.p2 = p.coef[2 * p.NO + ch];
    }
    return w;
}

// the loads of one unit for the block whose first row is row0 (live = false: no such block -- everything out of range).
// grp_u / off_u: group of the block and its first row inside it when a group is a multiple of 32 rows (uniform).
template <bool GATHER, int DCLS>
__device__ __forceinline__ void wg_load_unit(const TlWgrad &p, const WgUnit &w, long long row0, int grp_u, int off_u, int lane,
                                             bool live, WgRaw &r)
{
    const int hl = lane >> 5, rin = 16 * w.e + 8 * hl, step = uni(w.pitch1 * 4);
    const unsigned bytes1 = live ? 32u * (unsigned)w.pitch1 * 4u : 0u;
    rsrc_t r1 = make_rsrc(w.b1 ? w.b1 + (size_t)row0 * w.pitch1 : nullptr, w.b1 ? bytes1 : 0u);
    // per-group values: groups are 16 rows or a multiple of 32 (one group per block, uniform)
    const bool g16 = p.group_rows == 16;
    const int grp = g16 ? (((int)row0 + rin) >> 4) : grp_u;
    r.off = g16 ? ((rin & 8)) : off_u + rin;
    rsrc_t rq = make_rsrc(w.bq, (w.bq && live) ? 0x7fffffffu : 0u);
    const rsrc_t rs = make_rsrc(w.bs, (w.bs && live) ? 0x7fffffffu : 0u);
    int voffq = w.chq >= 0 ? (grp * w.nq + w.chq) * 4 : kWgOob;
    if (GATHER) {
        // rows of the grouped input: the point numbers of the step's 16 rows come through wave-uniform (scalar) loads --
        // counted by lgkmcnt, they do not disturb the vector loads in flight -- and each lane then picks its half
        const TlGather &g = p.g;
        const bool gat = w.kind == K_HGATHER;
        const int ggrp = (int)((unsigned)((int)row0 + rin) / (unsigned)g.nsample);      // group sizes are multiples of 8
        const int s0 = (int)row0 + rin - ggrp * g.nsample, cloud = ggrp / g.m;
        int pts[16];
        const int *ip = (g.idx && live) ? g.idx + row0 + 16 * w.e : nullptr;
#pragma unroll
        for (int j = 0; j < 16; ++j) pts[j] = ip ? ip[j] : 0;
        const rsrc_t r2 = make_rsrc(w.b2 ? w.b2 + (size_t)row0 * w.pitch1 : nullptr, w.b2 ? bytes1 : 0u);
        const rsrc_t rx = make_rsrc(w.bgx, (w.bgx && live) ? 0x7fffffffu : 0u), rf = make_rsrc(w.bgf, (w.bgf && live) ? 0x7fffffffu : 0u);
        if (gat) voffq = w.chq >= 0 ? (ggrp * 3 + w.chq) * 4 : kWgOob;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pt = g.idx ? (hl ? pts[8 + j] : pts[j]) : s0 + j;
            const int vx = w.kx >= 0 ? ((cloud * g.n + pt) * 3 + w.kx) * 4 : kWgOob;
            const int vf = w.kf >= 0 ? ((cloud * g.n + pt) * w.pitchf + w.kf) * 4 : kWgOob;
            const int vr = w.voff == kWgOob ? kWgOob : w.voff + j * step;
            r.z[j] = bload(gat ? rx : r1, gat ? vx : vr, 0);         // uniform choice of the descriptor
            r.g[j] = bload(gat ? rf : r2, gat ? vf : vr, 0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r.z[j] = bload(r1, w.voff, j * step);
        if (DCLS == D_DZ) {
            const rsrc_t r2 = make_rsrc(w.b2 ? w.b2 + (size_t)row0 * w.pitch1 : nullptr, w.b2 ? bytes1 : 0u);
#pragma unroll
            for (int j = 0; j < 8; ++j) r.g[j] = bload(r2, w.voff, j * step);
        }
    }
    if (DCLS != D_DZ || GATHER) {
        r.gq = bload(rq, voffq, 0);
        r.sel = bloadi(rs, voffq, 0);
    }
}

// Position of lane's 16-byte fragment inside a (tile, level, e) row of the block image. Plain lane order serves the weight
// gradient (every reader takes lane's own fragment); the data gradient fused into the pass reads the image TRANSPOSED --
// lane = row, eight 2-byte reads from eight channels' fragments -- and in plain order the 64 lanes of such a read hit
// four banks (rows 8 apart are 128-byte multiples apart). The XOR spreads the eight (e, row half, channel half) classes
// over the eight 16-byte bank groups; it permutes fragments inside aligned groups of eight, so the 16-byte accesses
// stay conflict-free.
__device__ __forceinline__ int wg_swz(int lane, int e) { return lane ^ (((lane >> 3) & 1) | (((lane >> 5) & 1) << 1) | (e << 2)); }

// prologue + split of a loaded unit -> its three fragments in the block image ([tile][level][e][lane] 16-byte vectors)
// zr != nullptr (data gradient in the same pass): the RAW rows of the first operand (the pre-norm tensor of the layer below)
// also go to LDS as fp32 [row][channel], pitch zpitch floats -- the data gradient's epilogue needs them for the ReLU mask
// and the batch-norm backward sums, and a global load there, however close in L2, could only return after every older
// prefetch load (in-order return counting): it cost the two-block prefetch
//
// imgA != nullptr: a dense second-operand unit (dz) also leaves its three levels in the layout the DATA gradient's MFMA reads
// as its A operand -- lane = row, eight consecutive channels per 16-byte fragment: [tile][level][K16 step q][row + 32 g] --
// as 2-byte stores (each lane holds ONE channel of eight rows; the transposition has to happen somewhere, and here it is
// spread over the eight producer waves instead of eight 2-byte reads per fragment in the two consumer waves). The 16-byte
// slot index is XOR-ed with (g | hl << 1 | q << 2): without it the 64 lanes of one store hit four banks.
template <int DCLS>
__device__ __forceinline__ void wg_store_unit(const WgUnit &w, const WgRaw &r, int lane, u32x4 *img, float *zr = nullptr, int zpitch = 0,
                                              u32x4 *imgA = nullptr, int tus = 0)
{
    if (w.kind == K_NONE || w.kind == K_ONES) return;             // nothing / written once before the loop
    if (zr && w.kind == K_H && w.relu && w.tile * 32 + 32 <= zpitch) {
        float *zo = zr + (16 * w.e + 8 * (lane >> 5)) * zpitch + w.tile * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) zo[j * zpitch] = r.z[j];
    }
    u32x4 *o = img + ((size_t)w.tile * 3 * 2 + w.e) * 64 + wg_swz(lane, w.e);
    if (DCLS == D_TOP && w.kind == K_FILL) {
        // one non-zero per lane (the pool routes dy to ONE row): split it once and drop its three bf16 levels into slot rel
        const int rel = r.sel - r.off;                             // the pooled sample's row inside this unit, if it is here
        const float v = (rel >= 0 && rel < 8 && w.chq >= 0) ? __fmul_rn(w.p0, r.gq) : 0.0f;
        const unsigned b1 = pack_bf16(v, 0.0f) & 0xffffu;
        const float r1 = __fsub_rn(v, __uint_as_float(b1 << 16));
        const unsigned b2 = pack_bf16(r1, 0.0f) & 0xffffu;
        const float r2 = __fsub_rn(r1, __uint_as_float(b2 << 16));
        const unsigned b3 = pack_bf16(r2, 0.0f) & 0xffffu;
        const int d = rel >> 1, sh = (rel & 1) * 16;
        u32x4 l1, l2, l3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            l1[q] = d == q ? b1 << sh : 0u;
            l2[q] = d == q ? b2 << sh : 0u;
            l3[q] = d == q ? b3 << sh : 0u;
        }
        o[0] = l1; o[128] = l2; o[256] = l3;
        return;
    }
    f32x16 x;
#pragma unroll
    for (int v = 0; v < 16; ++v) x[v] = 0.0f;
    if (w.kind == K_H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = vmax(__fadd_rn(__fmul_rn(w.p0, r.z[j]), w.p1), 0.0f);
            x[j] = w.voff == kWgOob ? 0.0f : w.relu ? y : r.z[j];
        }
    } else if (w.kind == K_HGATHER) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __fadd_rn(__fsub_rn(r.z[j], r.gq), r.g[j]);   // pointnet_util.py:46: coordinate - centroid (z, gq) or feature (g); the other stream read 0
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dy = DCLS == D_DZPOOL ? (r.sel - r.off == j ? r.gq : 0.0f) : r.g[j];
            x[j] = w.voff == kWgOob ? 0.0f : __fsub_rn(__fsub_rn(__fmul_rn(w.p0, dy), w.p1), __fmul_rn(w.p2, r.z[j]));
        }
    }
    const ActSplit sp = split_act(x);                             // registers 0..7 -> p[0][level]
    o[0] = sp.p[0][0];
    o[128] = sp.p[0][1];
    o[256] = sp.p[0][2];
    if (imgA && (w.kind == K_DZ || w.kind == K_DZPOOL)) {
        const int c = lane & 31, hl = lane >> 5, q = c >> 4, g = (c >> 3) & 1, sg = g | (hl << 1) | (q << 2);
        char *ba = reinterpret_cast<char *>(imgA) + ((size_t)(w.tile - tus) * 6 + q) * 1024 + (16 * w.e + 8 * hl + 32 * g) * 16 + (c & 7) * 2;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                char *pa = ba + (((2 * d + half) ^ sg) << 4);
#pragma unroll
                for (int lv = 0; lv < 3; ++lv)
                    *reinterpret_cast<unsigned short *>(pa + lv * 2048) = (unsigned short)(half ? sp.p[0][lv][d] >> 16 : sp.p[0][lv][d] & 0xffffu);
            }
    }
}

// Every wave keeps the rows of the NEXT TWO blocks in flight in registers (two raw sets, the block loop is unrolled by
// two), the block image in LDS is double buffered, one s_barrier per block.
// TPW: output tiles per wave; UPW: operand units a wave loads per 32-row block
//
// One pass per layer (DY): the weight gradient dW_l = h^T dz and the data gradient dy_{l-1} = (dz W_l^T) . [h > 0] consume
// the SAME two tiles of a row block -- dz_l from (dy_l, z_l) and h_{l-1} from z_{l-1} -- so as two kernels the layer's
// activations crossed HBM twice per direction. With DY the block image serves both: the waves that own no (or the fewest)
// dW tiles take one 32-column tile of dy_{l-1} each. Its A operand is the image read TRANSPOSED (lane = row: eight
// ds_read_u16 per 16-byte fragment instead of one ds_read_b128, no vector instruction but four packs -- the operand was
// formed, split and stored once, by the unit loads), its B operand the packed W^T, LDS-resident for the whole launch;
// the epilogue is the data-gradient GEMM's (mask of the layer below from its pre-norm tensor, the batch-norm backward
// sums, 128-byte row stores). The dy waves run their own copy of the block loop (template ROLE): vector-memory returns
// are counted in order, and a wait shared with waves that issue no mask loads / stores between two prefetches could
// only be the smaller count, i.e. the dy waves would wait for half of the prefetch they just issued.
template <int TPW, int UPW, bool GATHER, int DCLS, bool DY, bool L1X = false>
__global__ __launch_bounds__(kTlThreads) void tl_wgrad_kernel(const TlWgrad p)
{
#define PN2_BX blockIdx.x
#define PN2_BY blockIdx.y
#define PN2_GX gridDim.x
#include "tl_wgrad_body.inc"
#undef PN2_BX
#undef PN2_BY
#undef PN2_GX
}

// ---- a layer's data gradient AND its weight gradient in one launch, side by side (small levels) ---------------------------------
// dy_{l-1} = dz_l W_l^T (tl_gemm_body) and dW_l = h_{l-1}^T dz_l (tl_wgrad_body) read the same tensors and write disjoint ones. On
// a level of a few thousand rows each is a launch of 10-50 us that occupies a fraction of the chip for a few dependent
// round trips to memory, and one after the other they were half of such a level's backward time (profiles/r04: sem_seg SA4
// 25 + 29 and 40 + 30 us for its two upper layers). Two streams cost more than they gave (~10 us per cross-queue
// dependency, SideStream above). Here the two passes are two RANGES OF WORKGROUPS of one grid -- blocks [0, ga * gsl) run the
// GEMM on a (ga, gsl) grid, the rest the weight gradient on a (gw, slabs) grid -- like the producers and consumers of
// sa_fused_kernel, but with nothing to exchange. Registers and LDS are the larger of the two bodies'. One instantiation per
// shape pair that occurs at the reference networks' levels (launch_pair's table); any other pair takes the two launches.
template <int NS, int AMODE, int TPW, int UPW, int DCLS, bool GATHER = false>
__global__ __launch_bounds__(kTlThreads) void tl_pair_kernel(const TlGemm pg, const TlWgrad pw, const unsigned ga, const unsigned gsl,
                                                             const unsigned gw)
{
    const unsigned na = ga * gsl;
    if (blockIdx.x < na) {
        const unsigned sbx = blockIdx.x % ga, sby = blockIdx.x / ga;
        const TlGemm &p = pg;
#define PN2_BX sbx
#define PN2_BY sby
#define PN2_GX ga
#include "tl_gemm_body.inc"
#undef PN2_BX
#undef PN2_BY
#undef PN2_GX
    } else {
        const unsigned sb = blockIdx.x - na, sbx = sb % gw, sby = sb / gw;
        constexpr bool DY = false, L1X = false;
        const TlWgrad &p = pw;
#define PN2_BX sbx
#define PN2_BY sby
#define PN2_GX gw
#include "tl_wgrad_body.inc"
#undef PN2_BX
#undef PN2_BY
#undef PN2_GX
    }
}

// The sum of the workgroups' slabs ([slab][workgroup][E floats]) -> the caller's weight gradient, ONE launch: a block owns 32
// consecutive floats of the slab layout ([tile][v >> 2][lane][v & 3]: what the workgroups dumped, so every partial is read as
// contiguous 128-byte pieces) and its eight groups of 32 threads each add an eighth of the workgroups, in order, in fp64;
// the eight sums meet in LDS and are added in order. A fixed order whatever the timing; two stages in two launches (fp32
// sums of 32 workgroups, then fp64) were 17-20 us per weight gradient on levels whose whole backward is 200 us, one thread
// per OUTPUT element read 4 of every 16 bytes it touched.
// (tl_wgrad_reduce_a_kernel below still serves tl_top_s_kernel's partials.)
__global__ __launch_bounds__(256) void tl_wgrad_reduce_kernel(const float *__restrict__ in, long long nw, int TU, int TT, int tslabs,
                                                              int KI, int NO, float *__restrict__ gw, long long sk, long long sn,
                                                              double *__restrict__ plain, int accumulate)
{
    __shared__ double sh[8][32];
    const int ox = threadIdx.x & 31, ck = threadIdx.x >> 5;
    const int tu = (KI + 31) / 32, tt = (NO + 31) / 32, uslabs = (tu + TU - 1) / TU;
    const long long e = (long long)TU * TT * 1024, total = (long long)uslabs * tslabs * e;
    const long long chunk = (nw + 7) / 8;
    for (long long base = (long long)blockIdx.x * 32; base < total; base += (long long)gridDim.x * 32) {      // uniform trip count
        const long long i = base + ox, slab = i / e;
        const int r = (int)(i - slab * e), tile = r >> 10, q = r & 1023;
        const int v = ((q >> 8) << 2) | (q & 3), lane = (q >> 2) & 63;
        const int us = (int)(slab / tslabs), ts = (int)(slab - (long long)us * tslabs), ul = tile / TT, tl = tile - ul * TT;
        const int k = (us * TU + ul) * 32 + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), n = (ts * TT + tl) * 32 + (lane & 31);
        const bool live = us * TU + ul < tu && ts * TT + tl < tt && k < KI && n < NO;
        double sum = 0.0;
        if (live) {
            const long long w0 = ck * chunk, w1 = w0 + chunk < nw ? w0 + chunk : nw;
            const float *src = in + (size_t)(slab * nw + w0) * e + r;
#pragma unroll 8
            for (long long w = w0; w < w1; ++w, src += e) sum += (double)*src;
        }
        sh[ck][ox] = sum;
        __syncthreads();
        if (ck == 0 && live) {
            double d = sh[0][ox];
#pragma unroll
            for (int c = 1; c < 8; ++c) d += sh[c][ox];
            if (plain) plain[(size_t)k * NO + n] = d;              // (KI, NO) row-major fp64, for the pooled top layer's fix-up
            else gw[k * sk + n * sn] = accumulate ? __fadd_rn(gw[k * sk + n * sn], (float)d) : (float)d;
        }
        __syncthreads();
    }
}

// sums of `chunk` consecutive workgroups' partials (layout unchanged): in [slab][nw][E] -> out [slab][nchunks][E]
__global__ __launch_bounds__(256) void tl_wgrad_reduce_a_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                                long long nw, int chunk, long long nchunks, long long e4)
{
    const long long slab = blockIdx.z, ck = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < e4; i += (long long)gridDim.x * 256) {
        float4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
        const long long w0 = ck * chunk, w1 = w0 + chunk < nw ? w0 + chunk : nw;
        for (long long w = w0; w < w1; ++w) {
            const float4 v = in[(slab * nw + w) * e4 + i];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        out[(slab * nchunks + ck) * e4 + i] = sum;
    }
}

// ---- pooled top layer WITHOUT its pre-norm tensor ------------------------------------------------------------------------
// z_L (rows, C_L) is the largest tensor of an SA level and is only ever needed in dz_L = s dy_L - c0 - c1 z_L. With
// z_L = h W + b (h = the layer's input) both products that consume dz_L split into a part through the ROUTED gradient
// (one entry per group and channel) and a part through h:
//     dz_L W^T   = (s dy_L) W^T - h M - 1 r^T           M = W diag(c1) W^T  (K x K),   r = W (c0 + b c1)
//     h^T dz_L   = h^T (s dy_L) - (h^T h) W diag(c1) - (h^T 1) (c0 + b c1)^T
// so backward reads h (K channels) where it would read z_L (C_L = 2K channels in the reference stacks), forward never
// writes z_L, and the extra matrix work is K / C_L of the layer's -- the passes are memory-bound, it is free.
// tl_top_mats_kernel: the stacked fp32 weight of the data-gradient GEMM, rows [0, NFp) = W^T (c -> k), rows [NFp, NFp + K)
// = -M, and its constant row -r. One thread per element, fp64 accumulation.
__global__ __launch_bounds__(256) void tl_top_mats_kernel(const float *__restrict__ w, long long sk, long long sn, int K, int NF,
                                                          int NFp, const float *__restrict__ coef, const float *__restrict__ bias,
                                                          float *__restrict__ wp, float *__restrict__ rowc)
{
    // (W staged in LDS -- 64 x 128 at the metric shape -- was measured in the last session of round 6: 13.2 -> 24.5 us; the walk
    // over a row's columns hits the same cache lines trip after trip, the staging loop does not. Not kept.)
    auto W = [&](int n, int c) __attribute__((always_inline)) -> float { return w[n * sk + c * sn]; };
    const long long total = (long long)(NFp + K + 1) * K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / K), n = (int)(i - (long long)r * K);
        if (r < NFp) {
            wp[i] = r < NF ? W(n, r) : 0.0f;
        } else if (r < NFp + K) {
            const int j = r - NFp;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};                  // four chains: the loop is load latency, not arithmetic
            int c = 0;
            for (; c + 4 <= NF; c += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] += (double)W(j, c + u) * (double)coef[2 * NF + c + u] * (double)W(n, c + u);
            }
            for (; c < NF; ++c) acc[0] += (double)W(j, c) * (double)coef[2 * NF + c] * (double)W(n, c);
            wp[i] = (float)(-((acc[0] + acc[1]) + (acc[2] + acc[3])));
        } else {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            int c = 0;
            for (; c + 4 <= NF; c += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] += ((double)coef[NF + c + u] + (bias ? (double)bias[c + u] : 0.0) * (double)coef[2 * NF + c + u]) *
                              (double)W(n, c + u);
            }
            for (; c < NF; ++c)
                acc[0] += ((double)coef[NF + c] + (bias ? (double)bias[c] : 0.0) * (double)coef[2 * NF + c]) * (double)W(n, c);
            rowc[n] = (float)(-((acc[0] + acc[1]) + (acc[2] + acc[3])));
        }
    }
}

// dW[k][n] = S[k][n] - c1[n] sum_j G[k][j] W[j][n] - sumh[k] (c0[n] + b[n] c1[n]); sf = [S | G | sumh ..] (K, ld) fp64
__global__ __launch_bounds__(256) void tl_top_wgrad_fix_kernel(const double *__restrict__ sf, int ld, int K, int NF, int goff, int hoff,
                                                               const float *__restrict__ w, long long sk, long long sn,
                                                               const float *__restrict__ coef, const float *__restrict__ bias,
                                                               float *__restrict__ gw, const double *__restrict__ S, int accumulate)
{
    const long long total = (long long)K * NF;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i / NF), n = (int)(i - (long long)k * NF);
        const double *row = sf + (size_t)k * ld;
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
        int j = 0;
        for (; j + 4 <= K; j += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a4[u] += row[goff + j + u] * (double)w[(j + u) * sk + n * sn];
        }
        for (; j < K; ++j) a4[0] += row[goff + j] * (double)w[j * sk + n * sn];
        const double acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        const double c0 = coef[NF + n], c1 = coef[2 * NF + n], b = bias ? (double)bias[n] : 0.0;
        const float gv = (float)((S ? S[i] : row[n]) - c1 * acc - row[hoff] * (c0 + b * c1));
        gw[k * sk + n * sn] = accumulate ? __fadd_rn(gw[k * sk + n * sn], gv) : gv;      // S: (K, NF) of tl_top_s_kernel
    }
}

// ---- the ROUTED part of that weight gradient on the vector units ----------------------------------------------------------
// S = h^T (s dy_L), and dy_L has one non-zero per group and channel (the pooled sample):
//     S[k][c] = sum over the groups g of   h[row(g, argsel[g][c])][k] * s_c gq[g][c]
// -- C_L K multiply-adds per GROUP instead of per row. As tiles of the dense kernel (tl_wgrad_kernel, K_FILL units) the
// routed gradient was 2/3 of its operand tiles and matrix-core work (312 us at the metric shape, instruction-bound).
// A workgroup of eight waves stages the h rows of GB groups in LDS (relu(a z + c) applied on the way in; the next
// batch's rows are in flight in registers meanwhile); a wave owns 64 channels x KC inputs of S in registers, a lane = a
// channel: it reads the KC values of ITS pooled sample's row (ds_read_b128) and scales them. One (K, C_L) partial per
// workgroup, summed in fp64 afterwards (tl_wgrad_reduce_a_kernel + tl_top_s_reduce_kernel).
struct TlTopS {
    long long groups;
    int ns, K, NF, GB, ld;                  // rows per group, input / output channels, groups per batch, LDS row pitch (floats)
    const float *z, *pa, *pc;               // z_{L-1} (rows, K) and the coefficients of h = relu(pa z + pc)
    const float *gq;                        // (groups, NF) routed gradient
    const int *argsel;                      // (groups, NF) pooled sample of the group
    const float *coef;                      // s (NF)
    float *partial;                         // (gridDim.x, K, NF)
};
constexpr int kTopSThreads = 512, kTopSGroups = 8;

template <int KC, int NLD>
__global__ __launch_bounds__(kTopSThreads, (KC <= 16 && NLD <= 4) ? 4 : 2) void tl_top_s_kernel(const TlTopS p)
{
    extern __shared__ __attribute__((aligned(16))) float tops_lds[];
    float *coefs = tops_lds, *tile = tops_lds + 2 * p.K;            // [pa | pc], then GB x ns rows of ld floats
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kchunks = p.K / KC, nitems = ((p.NF + 63) / 64) * kchunks, item = blockIdx.y * 8 + wave;
    const bool work = item < nitems;
    const int cc = work ? item / kchunks : 0, kc = work ? item - cc * kchunks : 0, c = cc * 64 + lane;
    const bool cok = work && c < p.NF;
    const float sc = cok ? p.coef[c] : 0.0f;
    for (int i = threadIdx.x; i < p.K; i += kTopSThreads) { coefs[i] = p.pa[i]; coefs[p.K + i] = p.pc[i]; }
    float acc[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) acc[k] = 0.0f;
    const int k4row = p.K / 4;
    const long long nb = (p.groups + p.GB - 1) / p.GB;
    float4 raw[NLD];
    int an[kTopSGroups];
    float vn[kTopSGroups];
    auto fetch = [&](long long batch) {                          // rows of a batch are one contiguous range of z
        const long long g0 = batch * p.GB;
        const int ng = batch < nb ? (int)(p.groups - g0 < p.GB ? p.groups - g0 : p.GB) : 0;
        const int total4 = ng * p.ns * k4row;
        const float *src = p.z + (size_t)g0 * p.ns * p.K;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = threadIdx.x + i * kTopSThreads;
            raw[i] = e < total4 ? ld4(src + (size_t)e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int gi = 0; gi < kTopSGroups; ++gi) {
            const bool ok = cok && gi < ng;
            const size_t o = (size_t)(g0 + gi) * p.NF + c;
            an[gi] = ok ? p.argsel[o] : 0;
            vn[gi] = ok ? __fmul_rn(sc, p.gq[o]) : 0.0f;
        }
    };
    fetch(blockIdx.x);
    __syncthreads();                                               // coefs
    for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
        const int total4 = p.GB * p.ns * k4row;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = threadIdx.x + i * kTopSThreads;
            if (e < total4) {
                const int row = e / k4row, k = (e - row * k4row) * 4;
                const float4 a = *reinterpret_cast<const float4 *>(coefs + k), b = *reinterpret_cast<const float4 *>(coefs + p.K + k);
                float4 h;
                h.x = vmax(__fadd_rn(__fmul_rn(a.x, raw[i].x), b.x), 0.0f);
                h.y = vmax(__fadd_rn(__fmul_rn(a.y, raw[i].y), b.y), 0.0f);
                h.z = vmax(__fadd_rn(__fmul_rn(a.z, raw[i].z), b.z), 0.0f);
                h.w = vmax(__fadd_rn(__fmul_rn(a.w, raw[i].w), b.w), 0.0f);
                *reinterpret_cast<float4 *>(tile + (size_t)row * p.ld + k) = h;
            }
        }
        int ac[kTopSGroups];
        float vc[kTopSGroups];
#pragma unroll
        for (int gi = 0; gi < kTopSGroups; ++gi) { ac[gi] = an[gi]; vc[gi] = vn[gi]; }
        __syncthreads();
        fetch(batch + gridDim.x);                                  // in flight under the multiply-adds
        if (work) {
#pragma unroll
            for (int gi = 0; gi < kTopSGroups; ++gi) {
                if (gi < p.GB) {
                    const float *hr = tile + (size_t)(gi * p.ns + ac[gi]) * p.ld + kc * KC;
#pragma unroll
                    for (int k4 = 0; k4 < KC / 4; ++k4) {
                        const float4 h = *reinterpret_cast<const float4 *>(hr + 4 * k4);
                        acc[4 * k4] = fmaf(vc[gi], h.x, acc[4 * k4]);
                        acc[4 * k4 + 1] = fmaf(vc[gi], h.y, acc[4 * k4 + 1]);
                        acc[4 * k4 + 2] = fmaf(vc[gi], h.z, acc[4 * k4 + 2]);
                        acc[4 * k4 + 3] = fmaf(vc[gi], h.w, acc[4 * k4 + 3]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (cok) {
        float *dst = p.partial + ((size_t)blockIdx.x * p.K + (size_t)kc * KC) * p.NF + c;
#pragma unroll
        for (int k = 0; k < KC; ++k) dst[(size_t)k * p.NF] = acc[k];
    }
}

// S (K, NF) fp64 = sum of `nparts` fp32 partials
__global__ __launch_bounds__(256) void tl_top_s_reduce_kernel(const float *__restrict__ in, int nparts, long long total,
                                                              double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        int q = 0;
        for (; q + 4 <= nparts; q += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += (double)in[(size_t)(q + u) * total + i];
        }
        for (; q < nparts; ++q) a[0] += (double)in[(size_t)q * total + i];
        out[i] = (a[0] + a[1]) + (a[2] + a[3]);
    }
}

// ---- layer 1 of a grouped level ONCE PER POINT -----------------------------------------------------------------------------
// Layer 1 reads [xyz_j - c, f_j] (utils/pointnet_util.py:44-50): z_1 = W1f^T f_j + W1x^T (xyz_j - c) + b. The feature term
// depends on the POINT only, and a point is a sample of nsample m / n (16-64) groups: P = points . W1f is one GEMM over the
// b n points (the generic kernel, plain rows), and the pass over the b m nsample rows only gathers a row of P (cout_1 floats
// where the features were up to 320) and adds the three coordinate terms -- no matrix pipe needed for K = 3. Backward:
//     dW1x = (xyz - c)^T dz_1 and dz_1 itself      one pass over the rows (tl_l1_dz_kernel; dz_1 overwrites dy_1)
//     S    = scatter-add of dz_1 onto the points   pn2_group_point_grad_seg (the level's ordinary segmented reduction)
//     dW1f = points^T S,   dPoints = S W1f^T       two GEMMs over the b n points
// instead of a weight-gradient and a data-gradient GEMM over all rows with the gathered 131-323 channel input, and a
// segmented reduction of a (rows, cfeat) tensor. (The inference kernels do the same: csrc/sa_mlp_stream.hip.)
struct TlL1 {
    long long rows;
    int n, m, nsample, C;                   // points per cloud, groups per cloud, rows per group, cout_1
    const float *xyz, *new_xyz;             // (b,n,3), (b,m,3) or nullptr
    const int *idx;                         // (rows)
    const float *P;                         // forward: (b n, C)
    const float *wx;                        // forward: W1x, 3 rows of the weight: wx[k * skx + col * sn]
    long long skx, sn;
    const float *bias;                      // forward: (C) or nullptr
    float *z;                               // forward: out (rows, C);  backward: z_1 (rows, C)
    double *stats;                          // forward: (workgroups, 2, C) partial sums
    float *g;                               // backward: dy_1 in, dz_1 out (rows, C)
    const float *coef;                      // backward: (3, C): s, c0, c1
    float *part;                            // backward: (workgroups, 3 + cf, C) partial dW1 rows: coordinates, then features
    // backward, a FEW feature channels beside the coordinates whose gradient nobody wants (the input normals of cls_msg /
    // part_seg level 1): gathered per row and handled like three more coordinates -- the weight gradient of a layer of six
    // inputs is no more a matrix-core job than one of three (forward keeps the gathered GEMM: measured faster there)
    const float *points;                    // (b n, cf) or nullptr
    int cf;                                 // 0..kL1MaxFeat
};
constexpr int kL1MaxFeat = 5;               // 3 + 5 = 8 inputs at most on the vector units

// thread <-> (row lane, 4 columns): a block of kL1Threads covers kL1Threads / (C / 4) rows at a time, columns fixed per
// thread, and every thread keeps kL1U rows in flight (all loads of a batch are issued before the first is used: the
// point number -> coordinates / row of P chain is two dependent latencies, one row at a time ran at 1-2 TB/s).
// P == nullptr: a level WITHOUT features (the first level of every network): z_1 = b + (xyz - c) W1 on the vector units,
// three multiply-adds per output -- the generic gathered GEMM spent a matrix-core pass on a contraction of three.
constexpr int kL1Threads = 512, kL1U = 4;

struct L1Rows {                                   // the batch's rows: number, point, group (clamped to a valid row when !ok)
    unsigned row[kL1U];
    size_t pt[kL1U];                              // cloud * n + point
    unsigned grp[kL1U];
    bool ok[kL1U];
};

__device__ __forceinline__ L1Rows l1_rows(const TlL1 &p, unsigned base, unsigned stride, unsigned rows)
{
    L1Rows r;
#pragma unroll
    for (int u = 0; u < kL1U; ++u) {
        const unsigned rr = base + (unsigned)u * stride;
        r.ok[u] = rr < rows;
        r.row[u] = r.ok[u] ? rr : base;
        r.grp[u] = r.row[u] / (unsigned)p.nsample;
    }
#pragma unroll
    for (int u = 0; u < kL1U; ++u) r.pt[u] = (size_t)(r.grp[u] / (unsigned)p.m) * p.n + p.idx[r.row[u]];
    return r;
}

__device__ __forceinline__ void l1_coords(const TlL1 &p, const L1Rows &r, float (&x)[kL1U][3])
{
#pragma unroll
    for (int u = 0; u < kL1U; ++u) {
        const float *px = p.xyz + r.pt[u] * 3;
        x[u][0] = px[0]; x[u][1] = px[1]; x[u][2] = px[2];
    }
    if (p.new_xyz) {
#pragma unroll
        for (int u = 0; u < kL1U; ++u) {
            const float *pc = p.new_xyz + (size_t)r.grp[u] * 3;
            const float c0 = pc[0], c1 = pc[1], c2 = pc[2];
            x[u][0] = __fsub_rn(x[u][0], c0); x[u][1] = __fsub_rn(x[u][1], c1); x[u][2] = __fsub_rn(x[u][2], c2);   // pointnet_util.py:46
        }
    }
}

__global__ __launch_bounds__(kL1Threads) void tl_l1_forward_kernel(const TlL1 p)
{
    const int qpr = p.C / 4, q = threadIdx.x % qpr, rl = threadIdx.x / qpr, rpb = kL1Threads / qpr, col = 4 * q;
    float a0[4], a1[4], a2[4], b4[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float *w = p.wx + (size_t)col * p.sn;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0[i] = w[i * p.sn]; a1[i] = w[p.skx + i * p.sn]; a2[i] = w[2 * p.skx + i * p.sn]; }
        if (p.bias) { const float4 b = ld4(p.bias + col); b4[0] = b.x; b4[1] = b.y; b4[2] = b.z; b4[3] = b.w; }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // a workgroup walks ONE contiguous range of rows, a batch = kL1U consecutive slices of rpb rows (whole 32 KB runs of z)
    const unsigned rows = (unsigned)p.rows, stride = (unsigned)rpb, span = kL1U * stride;
    const unsigned chunk = (rows + gridDim.x * span - 1) / (gridDim.x * span) * span;
    const unsigned first = blockIdx.x * chunk, stop = first + chunk < rows ? first + chunk : rows;
    for (unsigned base = first + rl; base < stop; base += span) {
        const L1Rows r = l1_rows(p, base, stride, rows);
        float x[kL1U][3];
        float4 pp[kL1U];
        l1_coords(p, r, x);
#pragma unroll
        for (int u = 0; u < kL1U; ++u) pp[u] = p.P ? ld4(p.P + r.pt[u] * p.C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kL1U; ++u) {
            float z[4] = {pp[u].x + b4[0], pp[u].y + b4[1], pp[u].z + b4[2], pp[u].w + b4[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                z[i] = fmaf(x[u][0], a0[i], z[i]);
                z[i] = fmaf(x[u][1], a1[i], z[i]);
                z[i] = fmaf(x[u][2], a2[i], z[i]);
            }

            if (r.ok[u]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { s1[i] += z[i]; s2[i] = fmaf(z[i], z[i], s2[i]); }
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f zo = {z[0], z[1], z[2], z[3]};
                __builtin_nontemporal_store(zo, reinterpret_cast<v4f *>(p.z + (size_t)r.row[u] * p.C + col));
            }
        }
    }
    __shared__ double red[2][kL1Threads][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[0][threadIdx.x][i] = (double)s1[i]; red[1][threadIdx.x][i] = (double)s2[i]; }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < rpb; ++k) { a += red[0][k * qpr + q][i]; b += red[1][k * qpr + q][i]; }
            p.stats[((size_t)blockIdx.x * 2) * p.C + col + i] = a;
            p.stats[((size_t)blockIdx.x * 2 + 1) * p.C + col + i] = b;
        }
    }
}

// dz_1 = s dy_1 - c0 - c1 z_1 and the coordinate rows of the weight gradient, dW1x = (xyz - c)^T dz_1, in one pass over the
// rows. STORE: dz_1 overwrites dy_1 (the per-point path scatters it onto the points next); a level without features needs
// only dW1x = its whole first-layer weight gradient.
template <bool STORE, bool FEAT>
__global__ __launch_bounds__(kL1Threads) void tl_l1_dz_kernel(const TlL1 p)
{
    constexpr int NIN = FEAT ? 3 + kL1MaxFeat : 3;               // rows of the weight gradient a thread accumulates
    const int qpr = p.C / 4, q = threadIdx.x % qpr, rl = threadIdx.x / qpr, rpb = kL1Threads / qpr, col = 4 * q;
    const int nin = FEAT ? 3 + p.cf : 3;
    const float4 s4 = ld4(p.coef + col), c04 = ld4(p.coef + p.C + col), c14 = ld4(p.coef + 2 * p.C + col);
    const float s[4] = {s4.x, s4.y, s4.z, s4.w}, c0[4] = {c04.x, c04.y, c04.z, c04.w}, c1[4] = {c14.x, c14.y, c14.z, c14.w};
    float acc[NIN][4];
#pragma unroll
    for (int k = 0; k < NIN; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = 0.0f;
    // a workgroup walks ONE contiguous range of rows, a batch = kL1U consecutive slices of rpb rows (whole 32 KB runs of z)
    const unsigned rows = (unsigned)p.rows, stride = (unsigned)rpb, span = kL1U * stride;
    const unsigned chunk = (rows + gridDim.x * span - 1) / (gridDim.x * span) * span;
    const unsigned first = blockIdx.x * chunk, stop = first + chunk < rows ? first + chunk : rows;
    for (unsigned base = first + rl; base < stop; base += span) {
        const L1Rows r = l1_rows(p, base, stride, rows);
        float x[kL1U][NIN];
        float4 g4[kL1U], z4[kL1U];
#pragma unroll
        for (int u = 0; u < kL1U; ++u) {
            const size_t o = (size_t)r.row[u] * p.C + col;
            g4[u] = ld4(p.g + o);
            z4[u] = ld4(p.z + o);
        }
        {
            float xc[kL1U][3];
            l1_coords(p, r, xc);
#pragma unroll
            for (int u = 0; u < kL1U; ++u) {
                x[u][0] = xc[u][0]; x[u][1] = xc[u][1]; x[u][2] = xc[u][2];
                if (FEAT) {
#pragma unroll
                    for (int k = 0; k < kL1MaxFeat; ++k) x[u][3 + k] = k < p.cf ? p.points[r.pt[u] * p.cf + k] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kL1U; ++u) {
            const float gg[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w}, zz[4] = {z4[u].x, z4[u].y, z4[u].z, z4[u].w};
            float dz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dz[i] = __fsub_rn(__fsub_rn(__fmul_rn(s[i], gg[i]), c0[i]), __fmul_rn(c1[i], zz[i]));      // s dy - c0 - c1 z
                const float dv = r.ok[u] ? dz[i] : 0.0f;
#pragma unroll
                for (int k = 0; k < NIN; ++k) acc[k][i] = fmaf(x[u][k], dv, acc[k][i]);
            }
            if (STORE && r.ok[u])
                *reinterpret_cast<float4 *>(p.g + (size_t)r.row[u] * p.C + col) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        }
    }
    __shared__ float red[3][kL1Threads][4];
#pragma unroll
    for (int k0 = 0; k0 < NIN; k0 += 3) {                          // three gradient rows per trip through the 24 KB buffer
        if (k0) __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[k][threadIdx.x][i] = k0 + k < NIN ? acc[k0 + k < NIN ? k0 + k : 0][i] : 0.0f;
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (k0 + k < nin) {
                        double a = 0.0;
                        for (int j = 0; j < rpb; ++j) a += (double)red[k][j * qpr + q][i];
                        p.part[((size_t)blockIdx.x * nin + k0 + k) * p.C + col + i] = (float)a;
                    }
                }
        }
    }
}

// dW1x[k][col] = sum over the workgroups' partials (fp64), written to rows [xyz_off, xyz_off + 3) of grad_weight.
// A block of 256 threads owns 8 of the 3 C sums and adds the partial rows 32 at a time (a thread per sum walking all
// 256 rows was 60 us of dependent L2 latencies).
// (nin = 3 + cf rows: the coordinate rows go to gw, the feature rows to gwf -- the two blocks of the layer's weight gradient)
__global__ __launch_bounds__(256) void tl_l1_wx_reduce_kernel(const float *__restrict__ part, int nparts, int C, int nin,
                                                              float *__restrict__ gw, float *__restrict__ gwf,
                                                              long long sk, long long sn, int accumulate)
{
    __shared__ double sh[32][8];
    const int g = threadIdx.x >> 3, cl = threadIdx.x & 7, i = blockIdx.x * 8 + cl;
    double a = 0.0;
    if (i < nin * C)
        for (int q = g; q < nparts; q += 32) a += (double)part[(size_t)q * nin * C + i];
    sh[g][cl] = a;
    __syncthreads();
    if (g != 0 || i >= nin * C) return;
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < 32; ++r) sum += sh[r][cl];
    const int k = i / C, col = i - k * C;
    float *dst = k < 3 ? gw + k * sk + col * sn : gwf + (k - 3) * sk + col * sn;
    *dst = accumulate ? __fadd_rn(*dst, (float)sum) : (float)sum;
}

// ---- layer 1 of a level without features, weight gradient from moments (TlWgrad::l1x) ----------------------------------------
// the centred coordinates of every row as (x, y, z, 0) -- the layer above's one-pass backward reads them 16 bytes per row
// instead of gathering through idx -- and the nine moments sum x, sum x x^T of this workgroup's rows (fp64)
constexpr int kL1XrowsThreads = 1024;
__global__ __launch_bounds__(kL1XrowsThreads) void tl_l1_xrows_kernel(const TlL1 p, float4 *__restrict__ xg, double *__restrict__ mom)
{
    // (last session of round 6. The launch is at most 256 workgroups -- one partial row of moments each, summed in fp64 by the
    // consumer; with 256 threads a thread walked 16 rows one at a time, a chain of idx -> coordinates round trips, and nine
    // threads then added 256 LDS values each, serially: 20.8 us at the metric shape. Now 1024 threads, four rows in flight per
    // thread -- one trip at the metric shape -- and the workgroup's sums meet through a shuffle tree + one sum per wave, a fixed order.)
    double s[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = 0.0;
    const unsigned rows = (unsigned)p.rows, stride = gridDim.x * (unsigned)kL1XrowsThreads;
    auto row_in = [&](unsigned r, float (&x)[3], float (&c)[3]) __attribute__((always_inline)) {
        const unsigned grp = r / (unsigned)p.nsample;
        const size_t pt = (size_t)(grp / (unsigned)p.m) * p.n + p.idx[r];
        const float *px = p.xyz + pt * 3;
        x[0] = px[0]; x[1] = px[1]; x[2] = px[2];
        c[0] = c[1] = c[2] = 0.0f;
        if (p.new_xyz) {
            const float *pc = p.new_xyz + (size_t)grp * 3;
            c[0] = pc[0]; c[1] = pc[1]; c[2] = pc[2];
        }
    };
    auto row_out = [&](unsigned r, const float (&x)[3], const float (&c)[3]) __attribute__((always_inline)) {
        float x0 = x[0], x1 = x[1], x2 = x[2];
        if (p.new_xyz) { x0 = __fsub_rn(x0, c[0]); x1 = __fsub_rn(x1, c[1]); x2 = __fsub_rn(x2, c[2]); }      // pointnet_util.py:46
        xg[r] = make_float4(x0, x1, x2, 0.0f);
        const double d0 = x0, d1 = x1, d2 = x2;
        s[0] += d0; s[1] += d1; s[2] += d2;
        s[3] += d0 * d0; s[4] += d0 * d1; s[5] += d0 * d2; s[6] += d1 * d1; s[7] += d1 * d2; s[8] += d2 * d2;
    };
    unsigned r = blockIdx.x * (unsigned)kL1XrowsThreads + threadIdx.x;
    for (; (unsigned long long)r + 3ull * stride < rows; r += 4u * stride) {
        float x[4][3], c[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) row_in(r + u * stride, x[u], c[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) row_out(r + u * stride, x[u], c[u]);
    }
    for (; r < rows; r += stride) {
        float x[3], c[3];
        row_in(r, x, c);
        row_out(r, x, c);
    }
    __shared__ double red[kL1XrowsThreads / 64][9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double v = s[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < kL1XrowsThreads / 64; ++w) a += red[w][threadIdx.x];
        mom[(size_t)blockIdx.x * 9 + threadIdx.x] = a;
    }
}

// dW_1[k][c] = s_c A[k][c] - c0_c (sum x_k) - c1_c ((sum x x^T) W_1)[k][c] in fp64. A block of 256 threads owns 8 of the 3 C
// entries and adds the partial rows 32 at a time, in a fixed order (a thread per entry walking 256 rows was 160 us of
// dependent L2 latencies); every block sums the nine moments itself the same way.
__global__ __launch_bounds__(256) void tl_l1_wx_combine_kernel(const double *__restrict__ mom, int nmom, const double *__restrict__ l1a,
                                                               int nparts, int C, int pitch, const float *__restrict__ coef,
                                                               const float *__restrict__ wx, long long skx, long long sn,
                                                               float *__restrict__ gw, int accumulate)
{
    __shared__ double sh[32][9];
    __shared__ double m9[9];
    const int g = threadIdx.x >> 3, cl = threadIdx.x & 7;
    // moments: thread (g, j) for j < 9 (cl + 8 * (g & 1) covers 0..15) -- simpler: 32 groups x 9 values via two passes
    for (int j = cl; j < 9; j += 8) {
        double a = 0.0;
        for (int q = g; q < nmom; q += 32) a += mom[(size_t)q * 9 + j];
        sh[g][j] = a;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 32; ++r) a += sh[r][threadIdx.x];
        m9[threadIdx.x] = a;
    }
    __syncthreads();
    const int i = blockIdx.x * 8 + cl;                             // entry k * C + c
    const bool ok = i < 3 * C;
    const int k = ok ? i / C : 0, c = ok ? i - k * C : 0;
    double a = 0.0;
    if (ok)
        for (int q = g; q < nparts; q += 32) a += l1a[((size_t)q * 3 + k) * pitch + c];
    __syncthreads();
    sh[g][cl] = a;
    __syncthreads();
    if (g != 0 || !ok) return;
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < 32; ++r) sum += sh[r][cl];
    const double xx[3][3] = {{m9[3], m9[4], m9[5]}, {m9[4], m9[6], m9[7]}, {m9[5], m9[7], m9[8]}};
    double mw = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) mw += xx[k][j] * (double)wx[j * skx + c * sn];
    const double gr = (double)coef[c] * sum - (double)coef[C + c] * m9[k] - (double)coef[2 * C + c] * mw;
    gw[k * skx + c * sn] = accumulate ? __fadd_rn(gw[k * skx + c * sn], (float)gr) : (float)gr;
}

__global__ void tl_identity_coef_kernel(int C, float *__restrict__ coef)       // dz = 1 * g - 0 - 0 * z
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * C) coef[i] = i < C ? 1.0f : 0.0f;
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int tiles(int c) { return (c + 31) / 32; }
// The organisation overrides of a call (include/pn2ops.h: pn2_train_opts; NULL = every rule automatic). They travel as an
// argument through every rule below: the library reads no environment variable and keeps no mode.
typedef pn2_train_opts Opts;
static inline Opts opts_of(const pn2_train_opts *o)
{
    Opts d;
    memset(&d, 0, sizeof(d));
    return o ? *o : d;
}
static inline int pick_ns(int tn, const Opts &o)
{
    const int cap = (o.max_ns == 1 || o.max_ns == 2) ? o.max_ns : 4;      // results never depend on it
    const int ns = tn >= 3 ? 4 : tn == 2 ? 2 : 1;
    return ns > cap ? cap : ns;
}
// CUs of the current device (partition modes expose fewer than 256); 256 when there is no device (host-side queries)
static inline int device_cus()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
        (void)hipGetLastError();
        return 256;
    }
    return cus > 256 ? 256 : cus;
}

struct GemmShape { int K, N, tk, tn, ns, slabs, resident; size_t lds, pack_bytes; };

static GemmShape gemm_shape(long long rows, int K, int N, const Opts &o)
{
    GemmShape g;
    g.K = K; g.N = N;
    g.tk = tiles(K); g.tn = tiles(N);
    g.ns = pick_ns(g.tn, o);
    g.slabs = (g.tn + g.ns - 1) / g.ns;
    // few rows (group_all, the deep levels of the segmentation nets): a workgroup covers 256 rows of one column slab, so
    // narrower slabs are what spreads the pass over the chip -- the rows are re-read per slab, which is nothing here
    const long long rounds = (rows / 32 + kTlWaves - 1) / kTlWaves;
    while (g.ns > 1 && rounds * g.slabs < 128) {
        g.ns /= 2;
        g.slabs = (g.tn + g.ns - 1) / g.ns;
    }
    const size_t params = (size_t)3 * g.tk * 32 * sizeof(float), stage = (size_t)g.ns * kPairWords * 4;
    g.resident = params + stage * g.tk <= (size_t)144 * 1024 && o.force_stream != PN2_OPT_ON;
    g.lds = params + stage * (g.resident ? g.tk : 2);
    g.pack_bytes = (size_t)g.slabs * g.tk * g.ns * kPairWords * 4;
    return g;
}

struct WgradShape { int tus, tts, uslabs, tslabs, tpw, upw, two; long long gridx, nw, nchunks; size_t e, lds, lds_dy, partial_bytes, partial2_bytes; };

static WgradShape wgrad_shape(long long rows, int KI, int NO, bool gather = false, int cus = 256, int two_opt = PN2_OPT_AUTO)
{
    WgradShape w;
    const int tu = tiles(KI), tt = tiles(NO);
    w.tus = tu < 4 ? tu : 4;
    w.tts = tt < 8 ? tt : 8;
    if (gather && tu > 4 && tu <= 6) {
        // layer 1 of an SA level with features (131 / 134 input channels = five tiles): the grouped input is gathered from
        // an L2-resident tensor and cheap to read again, dz is not -- so ALL input tiles sit in one slab and dz is cut into
        // slabs of two tiles, each read once (with the 4 + 8 shape dz crossed HBM twice). Two units and two tiles per wave.
        w.tus = tu;
        w.tts = tt < 2 ? tt : 2;
    }
    // few rows: more, smaller slabs spread the pass over the chip (each slab re-reads the rows, which is nothing here)
    while ((long long)((tu + w.tus - 1) / w.tus) * ((tt + w.tts - 1) / w.tts) * ((rows / 32 + 3) / 4) < 64 && (w.tus > 1 || w.tts > 1)) {
        if (w.tts >= w.tus && w.tts > 1) w.tts = (w.tts + 1) / 2; else w.tus = (w.tus + 1) / 2;
    }
    w.uslabs = (tu + w.tus - 1) / w.tus;
    w.tslabs = (tt + w.tts - 1) / w.tts;
    const int nout = w.tus * w.tts, per = (nout + 7) / 8;
    w.tpw = per <= 1 ? 1 : per <= 2 ? 2 : 4;
    w.upw = (2 * (w.tus + w.tts) + 7) / 8;
    const size_t slabs = (size_t)w.uslabs * w.tslabs;
    const long long blocks = rows / 32;
    long long gx = cus / (long long)slabs;                         // one workgroup per CU over all slabs ...
    // ... or TWO where a second one fits beside the first (<= 128 registers: one output tile and at most two units per wave,
    // no gather; half the LDS): these passes run one barrier per 32-row block with little work behind it, and a second
    // workgroup fills the waits of the first
    // (measured, scripts/lab_ab.sh wgrad_two_per_cu: Gram pass 173 -> 124 us, layer-2 weight gradient 171 -> 140 us at the metric
    // shape; nothing to gain below ~16 row blocks per workgroup, where the passes are a few microseconds of launch latency)
    w.two = two_opt != PN2_OPT_OFF && w.tpw == 1 && w.upw <= 2 && !gather && (size_t)2 * (w.tus + w.tts) * 6144 <= (size_t)76 * 1024 &&
            (two_opt == PN2_OPT_ON || blocks >= 16ll * cus);
    if (w.two) gx *= 2;
    if (gx < 1) gx = 1;
    if (gx > (blocks + 3) / 4) gx = (blocks + 3) / 4;               // at least four row blocks per workgroup
    w.gridx = gx;
    w.nw = gx;
    w.e = (size_t)nout * 1024;
    w.lds = (size_t)2 * (w.tus + w.tts) * 6144;
    w.lds_dy = 0;
    w.nchunks = w.nw > 32 ? (w.nw + 31) / 32 : 0;
    w.partial_bytes = slabs * w.nw * w.e * sizeof(float);
    w.partial2_bytes = slabs * (size_t)w.nchunks * w.e * sizeof(float);
    return w;
}

// Can the layer's data gradient ride in its weight-gradient pass (tl_wgrad_kernel<.., DY>)? One slab (every tile of both
// operands in the block image), at most four output tiles and two dW tiles per wave, and image(s) + resident W^T within the
// CU's LDS -- with ONE image and two barriers per block when two do not fit. kc: channels of the contraction (= the second
// operand's tiles that enter the product), ki: output columns.
struct FuseShape { bool ok; int tk, nt, single, acopy, upw; size_t lds, xr_off, pack_bytes; };
static FuseShape fuse_shape(long long rows, const WgradShape &w, int kc, int ki, const Opts &o, bool dense = true)
{
    FuseShape f;
    memset(&f, 0, sizeof(f));
    if (o.fuse_wgrad == PN2_OPT_OFF || w.uslabs != 1 || w.tslabs != 1 || w.tpw > 2) return f;
    // measured (scripts/lab_ab.sh fuse_wgrad): -7 % of a level's backward at 1 M rows (-25 % of the layer's two passes), -4 % at
    // 0.5 M, nothing below -- there the passes are latency-bound and the resident W^T costs its load per workgroup
    if (o.fuse_wgrad == PN2_OPT_AUTO && rows < (1ll << 19)) return f;
    f.tk = tiles(kc); f.nt = tiles(ki);
    if (f.nt > 4 || f.tk > 8) return f;
    f.upw = w.upw;
    f.acopy = dense ? 1 : 0;                                       // dense second operand: its fragments also in the data gradient's layout
    const size_t img = (size_t)(w.tus + w.tts + (f.acopy ? f.tk : 0)) * 6144 + (size_t)32 * (w.tus * 32 + 2) * 4;      // operand fragments + the raw rows of the layer below
    const size_t wb = (size_t)f.tk * f.nt * kPairWords * 4;
    const size_t cap = (size_t)156 * 1024;
    if (2 * img + wb <= cap) f.single = 0;
    else if (img + wb <= cap) f.single = 1;
    else return f;
    f.xr_off = (f.single ? img : 2 * img) + wb;                   // 2 x 32 coordinate rows behind everything else
    f.lds = f.xr_off + 1024;
    f.pack_bytes = wb;
    f.ok = true;
    return f;
}

struct TlPlan {
    size_t pack[8];             // forward: W_l; backward: W_l^T
    size_t stats[8];            // (2, cout_l) doubles
    size_t coef[8];             // backward: (3, cout_l) floats
    size_t pool;                // forward: pmax, pamax (2 arrays of parts_total x cout_L)
    size_t gq;                  // backward, pooled: (groups, cout_L)
    size_t ga, gb;              // backward: dy ping-pong (rows, max width)
    size_t partial, partial2;   // backward: weight-gradient partial sums
    size_t partial_cap;         // ... and the bytes planned for `partial` (launch_wgrad / launch_pair refuse a larger shape; `partial2` is the
                                // second-stage buffer of the former two-launch reduction: still planned, no longer written)
    size_t topw, topsf;         // backward, pooled top layer without z_L: stacked fp32 weight + constant row; [S | G | sumh] fp64
    size_t tops_part, tops_part2, tops64;   // ... its routed part on the vector units: partials (two stages), S (K, C_L) fp64
    size_t l1p;                 // layer 1 per point: forward P (b n, cout_1); backward S (b n, cout_1)
    size_t l1seg, l1part, l1coef;   // backward: scratch of the segmented reduction, dW1x partials (256, 3, cout_1), identity coefficients
    size_t l1xg, l1mom, l1a;        // backward, level without features: centred coordinates of every row, their moments, x^T dy_1 partials
    size_t tickets;             // kFinTickets counters of the folded finalisations (TlFin), zeroed by the direction's pack launch
    size_t total;
};

// Is the pooled top layer's pre-norm tensor z_L kept (small levels), or are the passes that would read it rewritten in
// terms of the layer's input (tl_top_mats_kernel)? One rule for forward, backward and the caller's allocation.
static bool top_stored(long long rows, int nlayers, const int *widths, int pool_rows, const Opts &o)
{
    if (!pool_rows || nlayers < 2) return true;
    if (o.top_stored != PN2_OPT_AUTO) return o.top_stored == PN2_OPT_ON;
    return (size_t)rows * widths[nlayers] * sizeof(float) < ((size_t)32 << 20);
}
static inline int top_cols(int kin, int cl) { return tiles(cl) * 32 + tiles(kin) * 32 + 32; }

// group dims as the C ABI passes them to the workspace query: b, n, m, nsample, cfeat, has_idx
struct GroupDims { int b, n, m, nsample, cfeat, has_idx; };

// Is layer 1 of this grouped level evaluated once per POINT (tl_l1_forward_kernel)? One rule for forward, backward, the
// workspace sizes and the caller's allocation of the feature gradient.
static bool l1_per_point(int nlayers, const int *widths, const GroupDims *g, const Opts &o)
{
    if (!g || !g->has_idx || nlayers < 2 || o.l1_per_point == PN2_OPT_OFF) return false;
    const int c1 = widths[1];
    if (g->cfeat < 8 || g->cfeat % 4 || ((long long)g->b * g->n) % 32 || c1 % 4 || c1 / 4 > 256 || 256 % (c1 / 4)) return false;
    return widths[0] == 3 + g->cfeat;
}

// launch shape of tl_top_s_kernel (ok = false: the dense kernel takes the routed gradient as operand tiles)
struct TopSShape { bool ok; int GB, KC, NLD, ld, gridx, gridy, nchunks; size_t lds, part_bytes, part2_bytes; };

static TopSShape top_s_shape(long long rows, int pool_rows, int K, int NF, const Opts &o)
{
    TopSShape t;
    memset(&t, 0, sizeof(t));
    if (!pool_rows || K % 4) return t;
    {
        // measured (scripts/lab_ab.sh PN2_TL_TOP_SPARSE): 3-12 % of a level's backward from 0.26 M rows x 128 inputs up,
        // nothing or a few microseconds lost below (sem_seg's levels) -- there the extra launches cost what the tiles saved
        if (o.top_sparse == PN2_OPT_OFF || (o.top_sparse == PN2_OPT_AUTO && rows * K < (1ll << 24))) return t;
    }
    const long long per_group = (long long)pool_rows * K;           // floats of one group's input rows
    if (per_group > 16384) return t;                               // eight 16-byte loads per thread at most
    t.ld = K + 4;                                                  // rows 16 bytes apart in the banks
    const size_t group_lds = (size_t)pool_rows * t.ld * 4;
    int gb = (int)(8192 / per_group);                              // four loads per thread when a group allows (registers: two
    if (gb < 1) gb = 1;                                            // workgroups per CU), eight for the largest groups
    if (gb > kTopSGroups) gb = kTopSGroups;
    while (gb > 1 && (size_t)gb * group_lds > ((size_t)64 << 10)) --gb;
    t.GB = gb;
    t.lds = (size_t)2 * K * 4 + (size_t)gb * group_lds;
    if (t.lds > ((size_t)144 << 10)) return t;
    const int per_thread = (int)((gb * per_group / 4 + kTopSThreads - 1) / kTopSThreads);
    t.NLD = per_thread <= 1 ? 1 : per_thread <= 2 ? 2 : per_thread <= 4 ? 4 : 8;
    const int nchunk = (NF + 63) / 64;
    t.KC = 4;
    for (int kc = 64; kc >= 4; kc /= 2)                            // the widest strip of inputs that still gives eight waves work
        if (K % kc == 0 && nchunk * (K / kc) >= 6) { t.KC = kc; break; }
    const int items = nchunk * (K / t.KC);
    t.gridy = (items + 7) / 8;
    const long long groups = rows / pool_rows, nb = (groups + gb - 1) / gb;
    long long gx = ((long long)32 << 20) / ((long long)K * NF * 4);  // at most 32 MB of partial sums ...
    if (gx < 128) gx = 128;                                        // ... but half the chip at least
    if (gx > 512) gx = 512;
    if (gx > nb) gx = nb;
    t.gridx = (int)gx;
    t.nchunks = gx > 32 ? (int)((gx + 31) / 32) : 0;
    t.part_bytes = (size_t)gx * K * NF * 4;
    t.part2_bytes = (size_t)t.nchunks * K * NF * 4;
    t.ok = true;
    return t;
}

// Is this a grouped level WITHOUT features whose first layer (a contraction of the three coordinates) runs on the vector
// units (tl_l1_forward_kernel with P == nullptr, tl_l1_dz_kernel<false>)? The first level of every reference network.
static bool l1_coords_only(int nlayers, const int *widths, const GroupDims *g, const Opts &o)
{
    // (round 4: also with up to kL1MaxFeat feature channels beside the coordinates -- normals -- gathered per row)
    if (!g || !g->has_idx || g->cfeat > kL1MaxFeat || widths[0] != 3 + g->cfeat || nlayers < 2 || o.l1_coords == PN2_OPT_OFF) return false;
    const int c1 = widths[1];
    return c1 % 4 == 0 && c1 / 4 <= kL1Threads && kL1Threads % (c1 / 4) == 0;
}

// The weight-gradient launch's slab shape AS PLANNED: the launches call wgrad_shape with the device's CU count (<= 256,
// device_cus) and the caller's wgrad_two_per_cu, both of which can only make the grid -- hence the `partial` buffers --
// SMALLER than 256 CUs with two workgroups per CU wherever the shape allows them (planned with the
// defaults, an opted-in second workgroup or a 129..255-CU device wrote its slabs past the planned buffer).
static WgradShape wgrad_plan_shape(long long rows, int KI, int NO, bool gather, const Opts &o)
{
    return wgrad_shape(rows, KI, NO, gather, 256, o.wgrad_two_per_cu == PN2_OPT_OFF ? PN2_OPT_OFF : PN2_OPT_ON);
}

static bool tl_plan(long long rows, int nlayers, const int *widths, int pool_rows, int backward, TlPlan &pl,
                    const GroupDims *gd, const Opts &o)
{
    if (rows <= 0 || rows % 32 || rows >= (1ll << 31) || nlayers < 1 || nlayers > 8) return false;
    if (pool_rows && pool_rows != 16 && pool_rows % 32) return false;
    memset(&pl, 0, sizeof(pl));
    size_t off = 0;
    int maxw = 0;
    for (int l = 0; l < nlayers; ++l) {
        const int cin = widths[l], cout = widths[l + 1];
        if (cin <= 0 || cout <= 0 || cout % 4) return false;
        const bool ztop = backward && l == nlayers - 1 && !top_stored(rows, nlayers, widths, pool_rows, o);
        const GemmShape g = ztop ? gemm_shape(rows, tiles(cout) * 32 + cin, cin, o) : backward ? gemm_shape(rows, cout, cin, o) : gemm_shape(rows, cin, cout, o);
        size_t pb = g.pack_bytes;
        if (l == 0 && l1_per_point(nlayers, widths, gd, o)) {           // the per-point GEMMs' operand tiles instead
            const long long bn = (long long)gd->b * gd->n;
            const GemmShape gp = backward ? gemm_shape(bn, cout, gd->cfeat, o) : gemm_shape(bn, gd->cfeat, cout, o);
            if (gp.pack_bytes > pb) pb = gp.pack_bytes;
        }
        pl.pack[l] = off; off = align_up(off + pb);
        pl.stats[l] = off; off = align_up(off + sizeof(double) * 2 * cout * kMaxParts);
        if (backward) { pl.coef[l] = off; off = align_up(off + sizeof(float) * 3 * cout); }
        if (cout > maxw) maxw = cout;
    }
    const int cl = widths[nlayers];
    if (!backward) {
        if (pool_rows) {
            const long long parts = rows / (pool_rows == 16 ? 16 : 32);
            pl.pool = off; off = align_up(off + (size_t)2 * parts * cl * 4);
        }
    } else {
        if (pool_rows) { pl.gq = off; off = align_up(off + (size_t)(rows / pool_rows) * cl * 4); }
        pl.ga = off; off = align_up(off + (size_t)rows * maxw * 4);
        pl.gb = off; off = align_up(off + (size_t)rows * maxw * 4);
        size_t p1 = 0, p2 = 0;
        const bool ztop = !top_stored(rows, nlayers, widths, pool_rows, o);
        for (int l = 0; l < nlayers; ++l) {
            const bool zt = ztop && l == nlayers - 1;
            for (int gat = 0; gat < (l == 0 ? 2 : 1); ++gat) {      // layer 1 may be a gathered input (other slab shape)
                const WgradShape w = wgrad_plan_shape(rows, widths[l], zt ? top_cols(widths[l], widths[l + 1]) : widths[l + 1], gat != 0, o);
                if (w.partial_bytes > p1) p1 = w.partial_bytes;
                if (w.partial2_bytes > p2) p2 = w.partial2_bytes;
            }
            if (zt) {                                              // without the routed tiles (tl_top_s_kernel takes them)
                const WgradShape w = wgrad_plan_shape(rows, widths[l], top_cols(widths[l], 0), false, o);
                if (w.partial_bytes > p1) p1 = w.partial_bytes;
                if (w.partial2_bytes > p2) p2 = w.partial2_bytes;
            }
        }
        if (l1_per_point(nlayers, widths, gd, o)) {                  // dW1f = points^T S over the b n points
            const WgradShape w = wgrad_plan_shape((long long)gd->b * gd->n, gd->cfeat, widths[1], false, o);
            if (w.partial_bytes > p1) p1 = w.partial_bytes;
            if (w.partial2_bytes > p2) p2 = w.partial2_bytes;
        }
        pl.partial = off; off = align_up(off + p1);
        pl.partial2 = off; off = align_up(off + p2);
        pl.partial_cap = p1;
        if (ztop) {
            const int kin = widths[nlayers - 1];
            pl.topw = off; off = align_up(off + sizeof(float) * (size_t)(tiles(cl) * 32 + kin + 1) * kin);
            pl.topsf = off; off = align_up(off + sizeof(double) * (size_t)kin * top_cols(kin, cl));
            const TopSShape ts = top_s_shape(rows, pool_rows, kin, cl, o);
            if (ts.ok) {
                pl.tops_part = off; off = align_up(off + ts.part_bytes);
                pl.tops_part2 = off; off = align_up(off + ts.part2_bytes);
                pl.tops64 = off; off = align_up(off + sizeof(double) * (size_t)kin * cl);
            }
        }
    }
    if (l1_per_point(nlayers, widths, gd, o)) {
        const long long bn = (long long)gd->b * gd->n;
        pl.l1p = off; off = align_up(off + (size_t)bn * widths[1] * 4);
        if (backward) {
            pl.l1seg = off; off = align_up(off + (size_t)pn2_seg_grad_ws_bytes(gd->b, gd->n, (long long)gd->m * gd->nsample));
            pl.l1part = off; off = align_up(off + (size_t)kMaxParts * 3 * widths[1] * 4);
            pl.l1coef = off; off = align_up(off + (size_t)3 * widths[1] * 4);
        }
    } else if (backward && l1_coords_only(nlayers, widths, gd, o)) {
        pl.l1part = off; off = align_up(off + (size_t)kMaxParts * (3 + kL1MaxFeat) * widths[1] * 4);
        pl.l1xg = off; off = align_up(off + (size_t)rows * 16);
        pl.l1mom = off; off = align_up(off + (size_t)kMaxParts * 9 * sizeof(double));
        pl.l1a = off; off = align_up(off + (size_t)kMaxParts * 3 * widths[1] * sizeof(double));
    }
    pl.tickets = off; off = align_up(off + sizeof(unsigned) * kFinTickets);
    pl.total = off;
    return true;
}

static GroupDims group_dims(const pn2_group_src *g)
{
    GroupDims d = {g->b, g->n, g->m, g->nsample, g->points ? g->cfeat : 0, g->idx ? 1 : 0};
    return d;
}

static TlGather make_gather(const pn2_group_src *g)
{
    TlGather t;
    t.n = g->n; t.m = g->m; t.nsample = g->nsample; t.cfeat = g->points ? g->cfeat : 0;
    t.xyz_off = g->xyz_first ? 0 : t.cfeat;
    t.feat_off = g->xyz_first ? 3 : 0;
    t.xyz = g->xyz; t.new_xyz = g->new_xyz; t.points = g->points; t.idx = g->idx;
    return t;
}

// (Measured and not kept, round 4: a workgroup splitting its own resident slab from the fp32 weight instead of copying the
// packed tiles -- one launch of 7-9 us fewer per direction, but +5-7 us in EVERY GEMM of the level (4-6 trips of eight strided
// loads and a split per thread ahead of the first MFMA; with all loads issued up front the 48 live registers cost more than
// the latency they hid: sem_seg SA4 backward 245 -> 269 us, the metric level's data gradient 266 -> 292 us).
static void add_pack_job(TlPackJobs &jobs, int &n, const float *w, long long sk, long long sn, const GemmShape &g, void *out)
{
    TlPackJob &q = jobs.j[n++];
    q.w = w; q.sk = sk; q.sn = sn; q.K = g.K; q.N = g.N; q.tk = g.tk; q.ns = g.ns; q.slabs = g.slabs;
    q.out = reinterpret_cast<u32x4 *>(out);
}

static int launch_pack_jobs(const TlPackJobs &jobs, int n, hipStream_t st)
{
    if (n == 0) return PN2_OK;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        const long long total = (long long)jobs.j[i].slabs * jobs.j[i].tk * jobs.j[i].ns * 128;
        if (total > most) most = total;
    }
    long long blocks = (most + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    return launch(tl_pack_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, st, jobs);
}

static int launch_pack(const float *w, long long sk, long long sn, const GemmShape &g, void *out, hipStream_t st)
{
    TlPackJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    int n = 0;
    add_pack_job(jobs, n, w, sk, sn, g, out);
    return launch_pack_jobs(jobs, n, st);
}

template <int NS>
static int launch_gemm_ns(int amode, const TlGemm &p, const GemmShape &g, dim3 grid, hipStream_t st)
{
#define PN2_TL_CASE(M)                                                                   \
    case M: {                                                                            \
        auto kern = tl_gemm_kernel<NS, M>;                                               \
        if (int rc = allow_dynamic_lds(kern, lds)) return rc;                            \
        return launch(kern, grid, dim3(kTlThreads), lds, st, p);                         \
    }
    const size_t lds = (p.fin.ticket && g.lds < kFinLds) ? kFinLds : g.lds;             // the folded finalisation's scratch (tl_fin_tail)
    switch (amode) {
        PN2_TL_CASE(A_PLAIN)
        PN2_TL_CASE(A_GATHER)
        PN2_TL_CASE(A_RELU)
        PN2_TL_CASE(A_DZ)
        PN2_TL_CASE(A_DZ_POOL)
        PN2_TL_CASE(A_FILL)
    }
#undef PN2_TL_CASE
    return PN2_E_ARG;
}

static dim3 prep_gemm(TlGemm &p, const GemmShape &g, const Opts &o)
{
    p.K = g.K; p.N = g.N; p.tk = g.tk; p.resident = g.resident;
    {
        const size_t obytes = (size_t)p.rows * g.N * sizeof(float);
        p.nt = o.nt == PN2_OPT_OFF ? 0 : o.nt == PN2_OPT_ON ? 1 : obytes >= ((size_t)128 << 20);
#ifdef PN2_TL_LAB_BUILD            /* timing-study builds only (scripts/build_mlp_labs.sh): 1 = no stores, 2 = no statistics */
        p.lab = getenv("PN2_TL_LAB") ? atoi(getenv("PN2_TL_LAB")) : 0;
#else
        p.lab = 0;
#endif
    }
    const long long rounds = (p.rows / 32 + kTlWaves - 1) / kTlWaves;
    long long gx = kMaxParts / g.slabs;                        // persistent: one 8-wave workgroup per CU over all slabs
    if (gx < 1) gx = 1;
    if (gx > rounds) gx = rounds;
    return dim3((unsigned)gx, (unsigned)g.slabs);
}

static int launch_gemm(int amode, TlGemm &p, const GemmShape &g, hipStream_t st, const Opts &o, int *nparts = nullptr)
{
    const dim3 grid = prep_gemm(p, g, o);
    if (nparts) *nparts = (int)grid.x;
    if (p.fin.ticket) { p.fin.total = grid.x * grid.y; p.fin.nparts = (int)grid.x; }
    if (g.ns == 4) return launch_gemm_ns<4>(amode, p, g, grid, st);
    if (g.ns == 2) return launch_gemm_ns<2>(amode, p, g, grid, st);
    return launch_gemm_ns<1>(amode, p, g, grid, st);
}

template <int TPW, bool GATHER, int DCLS, bool DY = false, bool L1X = false>
static int launch_wgrad_kern(const TlWgrad &p, const WgradShape &w, dim3 grid, hipStream_t st)
{
    const size_t lds = DY ? w.lds_dy : w.lds;
#define PN2_WG_CASE(U)                                                          \
    if (w.upw == U) {                                                           \
        auto kern = tl_wgrad_kernel<TPW, U, GATHER, DCLS, DY, L1X>;             \
        if (int rc = allow_dynamic_lds(kern, lds)) return rc;                   \
        return launch(kern, grid, dim3(kTlThreads), lds, st, p);                \
    }
    PN2_WG_CASE(1) PN2_WG_CASE(2) PN2_WG_CASE(3)
#undef PN2_WG_CASE
    return PN2_E_ARG;
}

template <int TPW>
static int launch_wgrad_tpw(const TlWgrad &p, const WgradShape &w, dim3 grid, hipStream_t st)
{
    if (p.dy_w) {                                                  // the data gradient in the same pass (fuse_shape: TPW <= 2, no gather)
        if (TPW > 2 || p.amode == A_GATHER) return PN2_E_ARG;
        constexpr int T = TPW > 2 ? 2 : TPW;
        if (p.dmode == A_FILL) return launch_wgrad_kern<T, false, D_TOP, true>(p, w, grid, st);
        if (p.l1x) return p.dmode == A_DZ ? launch_wgrad_kern<T, false, D_DZ, true, true>(p, w, grid, st) : PN2_E_ARG;
        return p.dmode == A_DZ_POOL ? launch_wgrad_kern<T, false, D_DZPOOL, true>(p, w, grid, st)
                                    : launch_wgrad_kern<T, false, D_DZ, true>(p, w, grid, st);
    }
    if (p.dmode == A_FILL) return launch_wgrad_kern<TPW, false, D_TOP>(p, w, grid, st);
    if (p.amode == A_GATHER)
        return p.dmode == A_DZ_POOL ? launch_wgrad_kern<TPW, true, D_DZPOOL>(p, w, grid, st)
                                    : launch_wgrad_kern<TPW, true, D_DZ>(p, w, grid, st);
    return p.dmode == A_DZ_POOL ? launch_wgrad_kern<TPW, false, D_DZPOOL>(p, w, grid, st)
                                : launch_wgrad_kern<TPW, false, D_DZ>(p, w, grid, st);
}

static int launch_wgrad_reduce(const TlWgrad &p, const WgradShape &w, float *partial2, const pn2_bn_layer &L, hipStream_t st, double *plain);

static int launch_wgrad(TlWgrad &p, const WgradShape &w, float *partial2, const pn2_bn_layer &L, hipStream_t st,
                        double *plain = nullptr)
{
    if (p.partial_cap && w.partial_bytes > p.partial_cap) return PN2_E_ARG;   // never write past the planned buffer
    p.tus = w.tus; p.tts = w.tts; p.tslabs = w.tslabs;
    const dim3 grid((unsigned)w.gridx, (unsigned)(w.uslabs * w.tslabs));
#ifdef PN2_WG_TIMING               /* lab build (scripts/build_mlp_labs.sh wgtime): cycles per phase and wave of workgroup 0, printed per launch */
    static unsigned long long *tbuf = nullptr;
    if (!tbuf) (void)hipMalloc(&tbuf, 48 * sizeof(unsigned long long));
    (void)clear_async(tbuf, 48 * sizeof(unsigned long long), st);
    p.timing = tbuf;
#endif
    int rc = w.tpw == 1 ? launch_wgrad_tpw<1>(p, w, grid, st) : w.tpw == 2 ? launch_wgrad_tpw<2>(p, w, grid, st)
                                                                             : launch_wgrad_tpw<4>(p, w, grid, st);
    if (rc) return rc;
#ifdef PN2_WG_TIMING
    {
        unsigned long long h[48];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, tbuf, sizeof(h), hipMemcpyDeviceToHost);
        const double nb = (double)((p.rows / 32 + w.gridx - 1) / w.gridx);
        fprintf(stderr, "wgtime KI %d NO %d dy %d tus %d tts %d blocks/wg %.0f (cycles per block: units | barrier | loads+dW | dy mfma | dy epilogue | loop)\n",
                p.KI, p.NO, p.dy_w ? 1 : 0, w.tus, w.tts, nb);
        for (int wv = 0; wv < 8; ++wv)
            fprintf(stderr, "  wave %d: %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f\n", wv, h[wv * 6] / nb, h[wv * 6 + 1] / nb, h[wv * 6 + 2] / nb,
                    h[wv * 6 + 3] / nb, h[wv * 6 + 4] / nb, h[wv * 6 + 5] / nb);
    }
#endif
    return launch_wgrad_reduce(p, w, partial2, L, st, plain);
}

// the sum of the workgroups' slabs -> the caller's weight gradient (or `plain`, fp64, for the pooled top layer's fix-up)
static int launch_wgrad_reduce(const TlWgrad &p, const WgradShape &w, float *partial2, const pn2_bn_layer &L, hipStream_t st, double *plain)
{
    (void)partial2;                                                // (the second stage's buffer of the two-launch reduction)
    const long long total = (long long)w.uslabs * w.tslabs * (long long)w.e;     // floats of the slab layout
    long long blocks = (total + 31) / 32;
    if (blocks > 4096) blocks = 4096;
    return launch(tl_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)p.partial, w.nw, w.tus, w.tts, w.tslabs,
                  p.KI, p.NO, L.grad_weight, L.w_stride_k, L.w_stride_n, plain, L.grad_accumulate);
}

// One launch for a layer's data-gradient GEMM (its workgroups first) and its weight-gradient pass (tl_pair_kernel), then the
// weight gradient's reduction. kNoPair: no kernel for this pair of shapes -- the caller launches the two passes one after the
// other. The table = the pairs the size rules produce at the levels of the four reference networks below 0.5 M rows
// (scripts/train_pairs.py lists them); a level of other widths simply takes the two launches.
constexpr int kNoPair = -12345;
static int launch_pair(int amode, TlGemm &pg, const GemmShape &g, TlWgrad &pw, const WgradShape &w, float *partial2, const pn2_bn_layer &L,
                       hipStream_t st, const Opts &o, int *nparts, double *plain = nullptr)
{
    if (pw.dy_w) return kNoPair;
    if (pw.partial_cap && w.partial_bytes > pw.partial_cap) return PN2_E_ARG;   // never write past the planned buffer
    const bool gather = pw.amode == A_GATHER;
    const int dcls = pw.dmode == A_FILL ? D_TOP : pw.dmode == A_DZ_POOL ? D_DZPOOL : D_DZ;
    size_t lds = g.lds > w.lds ? g.lds : w.lds;
    if (pg.fin.ticket && lds < kFinLds) lds = kFinLds;
#define PN2_PAIR(AM, NS_, DC, TP, UP) PN2_PAIR_G(AM, NS_, DC, TP, UP, false)
#define PN2_PAIR_G(AM, NS_, DC, TP, UP, GA)                                                                              \
    if (amode == AM && g.ns == NS_ && dcls == DC && w.tpw == TP && w.upw == UP && gather == GA) {                        \
        auto kern = tl_pair_kernel<NS_, AM, TP, UP, DC, GA>;                                                             \
        const dim3 ga = prep_gemm(pg, g, o);                                                                             \
        if (pg.fin.ticket) { pg.fin.total = ga.x * ga.y; pg.fin.nparts = (int)ga.x; }                                    \
        pw.tus = w.tus; pw.tts = w.tts; pw.tslabs = w.tslabs;                                                            \
        if (int rc = allow_dynamic_lds(kern, lds)) return rc;                                                            \
        const unsigned total = ga.x * ga.y + (unsigned)w.gridx * (unsigned)(w.uslabs * w.tslabs);                        \
        if (int rc = launch(kern, dim3(total), dim3(kTlThreads), lds, st, pg, pw, ga.x, ga.y, (unsigned)w.gridx)) return rc; \
        if (nparts) *nparts = (int)ga.x;                                                                                 \
        return launch_wgrad_reduce(pw, w, partial2, L, st, plain);                                                       \
    }
    PN2_PAIR(A_DZ, 1, D_DZ, 1, 1)
    PN2_PAIR(A_DZ, 1, D_DZ, 1, 2)
    PN2_PAIR(A_DZ, 1, D_DZ, 2, 2)
    PN2_PAIR(A_DZ, 1, D_DZ, 4, 3)
    PN2_PAIR(A_DZ, 2, D_DZ, 1, 1)
    PN2_PAIR(A_DZ, 2, D_DZ, 2, 2)
    PN2_PAIR(A_DZ, 2, D_DZ, 4, 3)
    PN2_PAIR(A_DZ, 4, D_DZ, 2, 2)
    PN2_PAIR(A_DZ_POOL, 1, D_DZPOOL, 4, 3)
    PN2_PAIR(A_DZ_POOL, 2, D_DZPOOL, 4, 3)
    PN2_PAIR(A_FILL, 1, D_TOP, 1, 2)
    PN2_PAIR(A_FILL, 2, D_TOP, 2, 3)
    PN2_PAIR(A_FILL, 4, D_TOP, 4, 3)
    PN2_PAIR(A_PLAIN, 1, D_DZ, 1, 1)                               // layer 1 per point: dPoints = S W1f^T beside dW1f = points^T S
    PN2_PAIR(A_PLAIN, 1, D_DZ, 2, 2)
    PN2_PAIR(A_PLAIN, 2, D_DZ, 2, 2)
    PN2_PAIR(A_PLAIN, 4, D_DZ, 1, 2)
    PN2_PAIR(A_PLAIN, 4, D_DZ, 2, 2)
    PN2_PAIR_G(A_DZ, 1, D_DZ, 4, 3, true)                          // layer 1 of a group_all level (gathered input) with a feature gradient
    PN2_PAIR_G(A_DZ, 1, D_DZ, 2, 2, true)
    PN2_PAIR_G(A_DZ, 2, D_DZ, 4, 3, true)
#undef PN2_PAIR
#undef PN2_PAIR_G
#ifdef PN2_PAIR_TRACE              /* lab build: which pairs a run asks for that the table does not hold */
    fprintf(stderr, "no pair kernel: amode %d ns %d dcls %d tpw %d upw %d gather %d (rows %lld)\n", amode, g.ns, dcls, w.tpw, w.upw, (int)gather, pg.rows);
#endif
    return kNoPair;
}

// Is the pair launch wanted for this layer? (pn2_train_opts.pair_launch; the helper stream, when asked for, keeps the two launches)
static inline bool pair_wanted(long long rows, const Opts &o)
{
    if (o.pair_launch == PN2_OPT_OFF || o.side_stream == PN2_OPT_ON) return false;
    return o.pair_launch == PN2_OPT_ON || rows < (1ll << 19);
}

template <int KC>
static int launch_top_s_kc(const TlTopS &p, const TopSShape &t, hipStream_t st)
{
    const dim3 grid((unsigned)t.gridx, (unsigned)t.gridy);
#define PN2_TS_CASE(N)                                                              \
    if (t.NLD == N) {                                                               \
        auto kern = tl_top_s_kernel<KC, N>;                                         \
        if (int rc = allow_dynamic_lds(kern, t.lds)) return rc;                     \
        return launch(kern, grid, dim3(kTopSThreads), t.lds, st, p);                \
    }
    PN2_TS_CASE(1) PN2_TS_CASE(2) PN2_TS_CASE(4) PN2_TS_CASE(8)
#undef PN2_TS_CASE
    return PN2_E_ARG;
}

// S (K, NF) fp64 = h^T (s dy_L) through the pooled samples (tl_top_s_kernel + the two reduction stages)
static int launch_top_s(TlTopS &p, const TopSShape &t, float *part2, double *s64, hipStream_t st)
{
    p.GB = t.GB; p.ld = t.ld;
    int rc = t.KC == 64 ? launch_top_s_kc<64>(p, t, st) : t.KC == 32 ? launch_top_s_kc<32>(p, t, st)
           : t.KC == 16 ? launch_top_s_kc<16>(p, t, st) : t.KC == 8 ? launch_top_s_kc<8>(p, t, st) : launch_top_s_kc<4>(p, t, st);
    if (rc) return rc;
    const long long total = (long long)p.K * p.NF;
    const float *src = p.partial;
    int nparts = t.gridx;
    if (t.nchunks) {
        const long long e4 = total / 4;
        long long bx = (e4 + 255) / 256;
        if (bx > 64) bx = 64;
        rc = launch(tl_wgrad_reduce_a_kernel, dim3((unsigned)bx, (unsigned)t.nchunks, 1u), dim3(256), 0, st,
                    reinterpret_cast<const float4 *>(p.partial), reinterpret_cast<float4 *>(part2), (long long)t.gridx, 32,
                    (long long)t.nchunks, e4);
        if (rc) return rc;
        src = part2;
        nparts = t.nchunks;
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    return launch(tl_top_s_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, nparts, total, s64);
}

// ---- a helper stream for the backward pass of SMALL levels --------------------------------------------------------------------
// A layer's weight gradient and its data gradient both start from (dy_l, z_l, z_{l-1}) and neither needs the other. On levels
// of a few thousand rows each is a launch of 10-50 us that covers a fraction of the chip (the deep levels of the segmentation
// networks: ~25 such launches per level, one after the other on the caller's stream, were 260 us per level whatever its
// size). With a helper stream the weight-gradient launches (and their reductions / fix-ups) CAN run beside the data-gradient
// chain (opt-in, pn2_train_opts.side_stream -- the measurement at its use site says why it is not the default): fork after the layer's batch-norm coefficients exist, join before the next layer's data gradient overwrites the
// dy buffer the helper reads. One helper stream and two events per (device, caller stream), created on first use and kept
// (with allow_dynamic_lds's table the library's only process state); event record / wait are legal under HIP graph capture
// (the helper joins the capture at the fork and returns at the join), so a captured training step keeps working.
struct SideKey { int dev; hipStream_t main; bool operator==(const SideKey &o) const { return dev == o.dev && main == o.main; } };
struct SideKeyHash { size_t operator()(const SideKey &k) const { return std::hash<unsigned long long>()((unsigned long long)(uintptr_t)k.main * 31ull + (unsigned long long)k.dev); } };
struct SideRes { hipStream_t side; hipEvent_t fork_ev, join_ev; };

struct SideStream {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    bool forked = false;
    int init(hipStream_t st, bool enable)
    {
        main = st;
        if (!enable) return PN2_OK;
        static std::mutex mu;
        static std::unordered_map<SideKey, SideRes, SideKeyHash> table;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        std::lock_guard<std::mutex> lock(mu);
        const SideKey key = {dev, st};
        auto it = table.find(key);
        if (it == table.end()) {
            SideRes r = {nullptr, nullptr, nullptr};
            if ((e = hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking)) == hipSuccess &&
                (e = hipEventCreateWithFlags(&r.fork_ev, hipEventDisableTiming)) == hipSuccess)
                e = hipEventCreateWithFlags(&r.join_ev, hipEventDisableTiming);
            if (e != hipSuccess) {                                 // nothing half-built stays behind
                if (r.join_ev) (void)hipEventDestroy(r.join_ev);
                if (r.fork_ev) (void)hipEventDestroy(r.fork_ev);
                if (r.side) (void)hipStreamDestroy(r.side);
                return (int)e;
            }
            it = table.emplace(key, r).first;
        }
        side = it->second.side; fork_ev = it->second.fork_ev; join_ev = it->second.join_ev;
        return PN2_OK;
    }
    // Scope guard: an error return between fork() and join() must not leave the helper stream forked -- under HIP graph
    // capture that would be an unjoined capture, which invalidates it. join() is idempotent.
    ~SideStream() { (void)join(); }
    SideStream() = default;
    SideStream(const SideStream &) = delete;
    SideStream &operator=(const SideStream &) = delete;
    hipStream_t get() const { return side ? side : main; }      // where the forked work goes
    int fork()                                                   // the helper continues from here
    {
        if (!side || forked) return PN2_OK;
        hipError_t e = hipEventRecord(fork_ev, main);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, fork_ev, 0);
        forked = e == hipSuccess;
        return (int)e;
    }
    int join()                                                   // the caller's stream waits for everything forked so far
    {
        if (!side || !forked) return PN2_OK;
        hipError_t e = hipEventRecord(join_ev, side);
        if (e == hipSuccess) e = hipStreamWaitEvent(main, join_ev, 0);
        forked = false;
        return (int)e;
    }
};

static bool layers_ok(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group, int *widths)
{
    if (!layers || nlayers < 1 || nlayers > 8) return false;
    for (int l = 0; l < nlayers; ++l) {
        const pn2_bn_layer &L = layers[l];
        if (L.cin <= 0 || L.cout <= 0 || L.cout % 4) return false;
        if (l > 0 && L.cin != layers[l - 1].cout) return false;
        if (!L.weight || !L.gamma || !L.beta || !L.save) return false;
        widths[l] = L.cin;
        widths[l + 1] = L.cout;
    }
    if (group) {
        if (group->b <= 0 || group->n <= 0 || group->m <= 0 || group->nsample <= 0 || !group->xyz) return false;
        if ((long long)group->b * group->m * group->nsample != rows) return false;
        if (layers[0].cin != 3 + (group->points ? group->cfeat : 0)) return false;
        if (!group->idx && group->nsample != group->n) return false;
    } else if (layers[0].cin % 4) {
        return false;
    }
    return true;
}

}  // namespace pn2

extern "C" long long pn2_mlp_train_ws_bytes_ex(long long rows, int nlayers, const int *widths, int pool_rows, int backward,
                                               const int *group_dims, const pn2_train_opts *opts)
{
    pn2::TlPlan pl;
    pn2::GroupDims gd;
    if (group_dims) gd = {group_dims[0], group_dims[1], group_dims[2], group_dims[3], group_dims[4], group_dims[5]};
    if (!widths || !pn2::tl_plan(rows, nlayers, widths, pool_rows, backward, pl, group_dims ? &gd : nullptr, pn2::opts_of(opts))) return -1;
    return (long long)pl.total;
}
extern "C" long long pn2_mlp_train_ws_bytes(long long rows, int nlayers, const int *widths, int pool_rows, int backward,
                                            const int *group_dims)
{
    return pn2_mlp_train_ws_bytes_ex(rows, nlayers, widths, pool_rows, backward, group_dims, nullptr);
}

// 1: layer 1 of this grouped level runs once per point; backward then writes the gradient of `points` itself
// (grad_points (b, n, cfeat)) instead of the per-row gradient grad_feat_rows
extern "C" int pn2_mlp_train_layer1_per_point_ex(int nlayers, const int *widths, const int *group_dims, const pn2_train_opts *opts)
{
    if (!widths || !group_dims || nlayers < 1 || nlayers > 8) return 0;
    const pn2::GroupDims gd = {group_dims[0], group_dims[1], group_dims[2], group_dims[3], group_dims[4], group_dims[5]};
    return pn2::l1_per_point(nlayers, widths, &gd, pn2::opts_of(opts)) ? 1 : 0;
}
extern "C" int pn2_mlp_train_layer1_per_point(int nlayers, const int *widths, const int *group_dims)
{
    return pn2_mlp_train_layer1_per_point_ex(nlayers, widths, group_dims, nullptr);
}

// 1: the top layer's pre-norm tensor z_L is written by forward and read by backward (the caller allocates layers[L-1].z);
// 0: it is not (pooled stacks of two or more layers on large levels: layers[L-1].z may be NULL)
extern "C" int pn2_mlp_train_top_stored_ex(long long rows, int nlayers, const int *widths, int pool_rows, const pn2_train_opts *opts)
{
    if (!widths || nlayers < 1 || nlayers > 8) return 1;
    return pn2::top_stored(rows, nlayers, widths, pool_rows, pn2::opts_of(opts)) ? 1 : 0;
}
extern "C" int pn2_mlp_train_top_stored(long long rows, int nlayers, const int *widths, int pool_rows)
{
    return pn2_mlp_train_top_stored_ex(rows, nlayers, widths, pool_rows, nullptr);
}

// byte offsets of the backward workspace's dy ping-pong buffers and per-layer sums (diagnostics: scripts/train_mlp_check.py)
extern "C" int pn2_mlp_train_ws_layout(long long rows, int nlayers, const int *widths, int pool_rows, long long *ga, long long *gb,
                                       long long *stats, long long *coef)
{
    pn2::TlPlan pl;
    if (!widths || !pn2::tl_plan(rows, nlayers, widths, pool_rows, 1, pl, nullptr, pn2::opts_of(nullptr))) return PN2_E_ARG;
    if (ga) *ga = (long long)pl.ga;
    if (gb) *gb = (long long)pl.gb;
    for (int l = 0; l < nlayers; ++l) {
        if (stats) stats[l] = (long long)pl.stats[l];
        if (coef) coef[l] = (long long)pl.coef[l];
    }
    return PN2_OK;
}

namespace pn2 {
// layer 1 on the vector units: the pass over the rows (P: the per-point products, or nullptr for a level without features)
static int launch_l1_forward(long long rows, const GroupDims &gd, const pn2_group_src *group, const pn2_bn_layer &L, const float *P,
                             double *stats, hipStream_t st, int *nparts)
{
    const TlGather gt = make_gather(group);
    TlL1 q;
    memset(&q, 0, sizeof(q));
    q.rows = rows; q.n = gd.n; q.m = gd.m; q.nsample = gd.nsample; q.C = L.cout;
    q.xyz = group->xyz; q.new_xyz = group->new_xyz; q.idx = group->idx; q.P = P;
    q.wx = L.weight + gt.xyz_off * L.w_stride_k; q.skx = L.w_stride_k; q.sn = L.w_stride_n;

    q.bias = nullptr; q.z = L.z;                  // no conv bias in the stored tensor (pn2_mlp_train_forward)
    q.stats = stats;
    const int rpb = kL1Threads / (L.cout / 4);
    long long blocks = (rows + (long long)rpb * kL1U - 1) / ((long long)rpb * kL1U);
    if (blocks > kMaxParts) blocks = kMaxParts;
    *nparts = (int)blocks;
    return launch(tl_l1_forward_kernel, dim3((unsigned)blocks), dim3(kL1Threads), 0, st, q);
}

// dz_1 (in place when `store`) and dW1x -> rows [xyz_off, xyz_off + 3) of the layer's weight gradient
static int launch_l1_dz(long long rows, const GroupDims &gd, const pn2_group_src *group, const pn2_bn_layer &L, float *dy,
                        const float *coef, float *part, bool store, hipStream_t st)
{
    const TlGather gt = make_gather(group);
    TlL1 q;
    memset(&q, 0, sizeof(q));
    q.rows = rows; q.n = gd.n; q.m = gd.m; q.nsample = gd.nsample; q.C = L.cout;
    q.xyz = group->xyz; q.new_xyz = group->new_xyz; q.idx = group->idx;
    q.z = L.z; q.g = dy; q.coef = coef; q.part = part;
    const bool feat = !store && gt.cfeat > 0;                     // (store = the per-point path: its features went through P)
    if (feat) { q.points = group->points; q.cf = gt.cfeat; }
    const int nin = 3 + q.cf;
    const int rpb = kL1Threads / (L.cout / 4);
    long long blocks = (rows + (long long)rpb * kL1U - 1) / ((long long)rpb * kL1U);
    if (blocks > kMaxParts) blocks = kMaxParts;
    if (int rc = store ? launch(tl_l1_dz_kernel<true, false>, dim3((unsigned)blocks), dim3(kL1Threads), 0, st, q)
                 : feat ? launch(tl_l1_dz_kernel<false, true>, dim3((unsigned)blocks), dim3(kL1Threads), 0, st, q)
                        : launch(tl_l1_dz_kernel<false, false>, dim3((unsigned)blocks), dim3(kL1Threads), 0, st, q)) return rc;
    return launch(tl_l1_wx_reduce_kernel, dim3((unsigned)((nin * L.cout + 7) / 8)), dim3(256), 0, st, (const float *)part, (int)blocks,
                  L.cout, nin, L.grad_weight + gt.xyz_off * L.w_stride_k, L.grad_weight + gt.feat_off * L.w_stride_k, L.w_stride_k,
                  L.w_stride_n, L.grad_accumulate);
}
}  // namespace pn2

namespace pn2 {
// Per-channel finalisations inside the launches that produce their sums (TlFin): OPT-IN. Measured (rocprofv3 per-dispatch
// traces, profiles/r04/fold_finalize_experiment.txt): the folded pass is 5-7 us longer on a level of 4,096 rows (sem_seg SA4:
// 17.1 / 25.8 / 31.6 against 15.2 / 18.6 / 26.2 us) and 13 us longer with 256 partial rows of 128 channels (sem_seg FP4), for a
// finalisation launch of 4.5-6 us saved -- the write-through stores, the ticket and the last workgroup's reads are four or
// more dependent trips to memory, because the L2s of the eight XCDs are not coherent with each other. With release / acquire
// fences instead (a write-back and an invalidate of the whole L2 per workgroup) it was 20-35 us longer.
static inline bool fold_wanted(const Opts &o) { return o.fold_finalize == PN2_OPT_ON; }

static TlFin fin_forward(const pn2_bn_layer &L, long long rows, double *stats, unsigned *ticket)
{
    TlFin f;
    memset(&f, 0, sizeof(f));
    f.ticket = ticket; f.mode = 1; f.N = L.cout; f.stats = stats; f.count = (double)rows;
    f.gamma = L.gamma; f.beta = L.beta; f.bias = L.bias; f.running_mean = L.running_mean; f.running_var = L.running_var;
    f.momentum = L.momentum; f.eps = L.eps; f.var_biased = L.running_var_biased; f.save = L.save;
    return f;
}

static TlFin fin_backward(const pn2_bn_layer &L, long long rows, double *stats, float *coef, unsigned *ticket)
{
    TlFin f;
    memset(&f, 0, sizeof(f));
    f.ticket = ticket; f.mode = 2; f.N = L.cout; f.stats = stats; f.count = (double)rows;
    f.gamma = L.gamma; f.save = L.save; f.grad_gamma = L.grad_gamma; f.grad_beta = L.grad_beta; f.coef = coef;
    f.accumulate = L.grad_accumulate;
    return f;
}
}  // namespace pn2

extern "C" int pn2_mlp_train_forward(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                                     const float *x, int pool_rows, float *out, int *argsel, float *zsel, void *ws, void *stream)
{
    return pn2_mlp_train_forward_ex(rows, nlayers, layers, group, x, pool_rows, out, argsel, zsel, ws, nullptr, stream);
}

extern "C" int pn2_mlp_train_forward_ex(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                                        const float *x, int pool_rows, float *out, int *argsel, float *zsel, void *ws,
                                        const pn2_train_opts *opts, void *stream)
{
    using namespace pn2;
    const Opts o = opts_of(opts);
    int widths[9];
    if (!layers_ok(rows, nlayers, layers, group, widths)) return PN2_E_ARG;
    if ((!group && !x) || !out || !ws || (pool_rows && (!argsel || !zsel))) return PN2_E_NULL;
    if (pool_rows && (rows % pool_rows || (group && pool_rows != group->nsample))) return PN2_E_ARG;
    TlPlan pl;
    GroupDims gd;
    if (group) gd = group_dims(group);
    if (!tl_plan(rows, nlayers, widths, pool_rows, 0, pl, group ? &gd : nullptr, o)) return PN2_E_ARG;
    const bool per_point = group && l1_per_point(nlayers, widths, &gd, o);
    // forward: layer 1 on the vector units only WITHOUT features (with the input normals gathered per row the vector kernel
    // measured slower than the gathered GEMM: cls_msg forward 2.51 -> 2.57 ms; its backward counterpart is the one that pays)
    const bool coords_only = group && gd.cfeat == 0 && l1_coords_only(nlayers, widths, &gd, o);
    const bool keep_top = top_stored(rows, nlayers, widths, pool_rows, o);
    for (int l = 0; l < nlayers; ++l)
        if (!layers[l].z && (keep_top || l < nlayers - 1)) return PN2_E_NULL;
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(ws);
    unsigned *tickets = reinterpret_cast<unsigned *>(base + pl.tickets);
    bool fold = false;                                            // the finalisation of a layer's batch moments inside the pass that sums them
    {
        TlPackJobs jobs;                                          // every layer's weights -> operand tiles, one launch
        memset(&jobs, 0, sizeof(jobs));
        int nj = 0;
        for (int l = 0; l < nlayers; ++l) {
            const pn2_bn_layer &L = layers[l];
            if (l == 0 && coords_only) continue;                 // no matrix-core pass at all
            if (l == 0 && per_point) {                            // only the feature rows of W_1: P = points . W1f
                const TlGather gt = make_gather(group);
                add_pack_job(jobs, nj, L.weight + gt.feat_off * L.w_stride_k, L.w_stride_k, L.w_stride_n,
                             gemm_shape((long long)gd.b * gd.n, gt.cfeat, L.cout, o), base + pl.pack[l]);
            } else {
                add_pack_job(jobs, nj, L.weight, L.w_stride_k, L.w_stride_n, gemm_shape(rows, L.cin, L.cout, o), base + pl.pack[l]);
            }
        }
        fold = fold_wanted(o) && nj > 0;                          // (the pack launch zeroes the tickets)
        if (fold) jobs.tickets = tickets;
        if (int rc = launch_pack_jobs(jobs, nj, st)) return rc;
    }
    // The conv bias is NOT added to the pre-norm tensors: batch normalisation removes any per-channel constant, so
    // z_l := h W_l gives the same output, the same gradients (the bias gradient is zero) and the same batch variance; only
    // the batch MEAN that enters the running average is mean(z_l) + b_l (tl_bn_finalize_kernel). It is more than a saved
    // add: the folded form a z + c loses accuracy with |mean| / std of a channel, and a bias is pure mean.
    for (int l = 0; l < nlayers; ++l) {
        const pn2_bn_layer &L = layers[l];
        const GemmShape g = gemm_shape(rows, L.cin, L.cout, o);
        const bool last = l == nlayers - 1;
        if (l == 0 && per_point) {
            // layer 1 once per point (tl_l1_forward_kernel): P = points . W1f over the b n points, then one pass over the rows
            const TlGather gt = make_gather(group);
            const long long bn = (long long)gd.b * gd.n;
            float *P = reinterpret_cast<float *>(base + pl.l1p);
            {
                const GemmShape gp = gemm_shape(bn, gt.cfeat, L.cout, o);
                TlGemm q;
                memset(&q, 0, sizeof(q));
                q.rows = bn;
                q.A = group->points;
                q.wpacked = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
                q.emode = E_STORE;
                q.out = P;
                if (int rc = launch_gemm(A_PLAIN, q, gp, st, o)) return rc;
            }
            int np = 0;
            if (int rc = launch_l1_forward(rows, gd, group, L, P, reinterpret_cast<double *>(base + pl.stats[l]), st, &np)) return rc;
            if (int rc = launch(tl_bn_finalize_kernel, dim3((unsigned)((L.cout + 7) / 8)), dim3(256), 0, st,
                                reinterpret_cast<const double *>(base + pl.stats[l]), np, L.cout, (double)rows, L.gamma,
                                L.beta, L.running_mean, L.running_var, L.momentum, L.eps, L.save, L.bias, L.running_var_biased)) return rc;
            continue;
        }
        if (l == 0 && coords_only) {
            // a level without features: z_1 = b + (xyz - c) W1 in one pass on the vector units (tl_l1_forward_kernel, no P)
            int np = 0;
            if (int rc = launch_l1_forward(rows, gd, group, L, nullptr, reinterpret_cast<double *>(base + pl.stats[l]), st, &np)) return rc;
            if (int rc = launch(tl_bn_finalize_kernel, dim3((unsigned)((L.cout + 7) / 8)), dim3(256), 0, st,
                                reinterpret_cast<const double *>(base + pl.stats[l]), np, L.cout, (double)rows, L.gamma,
                                L.beta, L.running_mean, L.running_var, L.momentum, L.eps, L.save, L.bias, L.running_var_biased)) return rc;
            continue;
        }
        TlGemm p;
        memset(&p, 0, sizeof(p));
        p.rows = rows;
        int amode;
        if (l == 0 && group) { amode = A_GATHER; p.g = make_gather(group); }
        else if (l == 0) { amode = A_PLAIN; p.A = x; }
        else { amode = A_RELU; p.A = layers[l - 1].z; p.p0 = layers[l - 1].save + 2 * layers[l - 1].cout; p.p1 = layers[l - 1].save + 3 * layers[l - 1].cout; }
        p.wpacked = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
        p.bias = nullptr;                                        // see the comment above the loop
        p.emode = (last && pool_rows) ? E_POOL : E_STORE;
        p.out = (last && !keep_top) ? nullptr : L.z;              // the pooled top layer of a large level is never written
        p.stats = reinterpret_cast<double *>(base + pl.stats[l]);
        if (p.emode == E_POOL) {
            const long long parts = rows / (pool_rows == 16 ? 16 : 32);
            float *pp = reinterpret_cast<float *>(base + pl.pool);
            p.pmax = pp;
            p.pamax = reinterpret_cast<int *>(pp + parts * L.cout);
            p.pool_gamma = L.gamma;
            p.prow = pool_rows == 16 ? 16 : 32;
        }
        int nparts = 0;
        if (fold) p.fin = fin_forward(L, rows, p.stats, tickets + l);
        if (int rc = launch_gemm(amode, p, g, st, o, &nparts)) return rc;
        if (!fold)
            if (int rc = launch(tl_bn_finalize_kernel, dim3((unsigned)((L.cout + 7) / 8)), dim3(256), 0, st,
                                reinterpret_cast<const double *>(base + pl.stats[l]), nparts, L.cout, (double)rows, L.gamma, L.beta,
                                L.running_mean, L.running_var, L.momentum, L.eps, L.save, L.bias, L.running_var_biased)) return rc;
        if (last && pool_rows) {
            const long long groups = rows / pool_rows;
            const int prow = pool_rows == 16 ? 16 : 32;
            long long blocks = (groups * L.cout + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            if (int rc = launch(tl_pool_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, groups, L.cout, pool_rows / prow,
                                prow, (const float *)p.pmax, (const int *)p.pamax, (const float *)L.gamma,
                                (const float *)L.save, out, argsel, zsel)) return rc;
        } else if (last) {
            const long long total4 = rows * L.cout / 4;
            long long blocks = (total4 + 255) / 256;
            if (blocks > 8192) blocks = 8192;
            if (int rc = launch(tl_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, total4, L.cout, (const float *)L.z,
                                (const float *)L.save, out)) return rc;
        }
    }
    return PN2_OK;
}

extern "C" int pn2_mlp_train_backward(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                                      const float *x, int pool_rows, const float *out, const int *argsel, const float *zsel,
                                      const float *grad_out, float *grad_x, float *grad_feat_rows, float *grad_points,
                                      int reproducible, void *ws, void *stream)
{
    return pn2_mlp_train_backward_ex(rows, nlayers, layers, group, x, pool_rows, out, argsel, zsel, grad_out, grad_x, grad_feat_rows,
                                     grad_points, reproducible, ws, nullptr, stream);
}

extern "C" int pn2_mlp_train_backward_ex(long long rows, int nlayers, const pn2_bn_layer *layers, const pn2_group_src *group,
                                         const float *x, int pool_rows, const float *out, const int *argsel, const float *zsel,
                                         const float *grad_out, float *grad_x, float *grad_feat_rows, float *grad_points,
                                         int reproducible, void *ws, const pn2_train_opts *opts, void *stream)
{
    using namespace pn2;
    const Opts o = opts_of(opts);
    const int cus = device_cus();
    int widths[9];
    if (!layers_ok(rows, nlayers, layers, group, widths)) return PN2_E_ARG;
    if ((!group && !x) || !out || !grad_out || !ws || (pool_rows && (!argsel || !zsel))) return PN2_E_NULL;
    for (int l = 0; l < nlayers; ++l)
        if (!layers[l].grad_weight || !layers[l].grad_gamma || !layers[l].grad_beta) return PN2_E_NULL;
    TlPlan pl;
    GroupDims gd;
    if (group) gd = group_dims(group);
    if (!tl_plan(rows, nlayers, widths, pool_rows, 1, pl, group ? &gd : nullptr, o)) return PN2_E_ARG;
    hipStream_t st = as_stream(stream);
    char *base = static_cast<char *>(ws);
    const bool per_point = group && l1_per_point(nlayers, widths, &gd, o);
    const bool coords_only = group && l1_coords_only(nlayers, widths, &gd, o);
    const bool want_dx = group ? ((per_point ? grad_points : grad_feat_rows) && group->points && group->cfeat > 0) : grad_x != nullptr;
    const bool ztop = !top_stored(rows, nlayers, widths, pool_rows, o);     // pooled top layer without z_L (tl_top_mats_kernel)
    for (int l = 0; l < nlayers; ++l)
        if (!layers[l].z && !(ztop && l == nlayers - 1)) return PN2_E_NULL;
    // Which layers run their data gradient inside the weight-gradient pass (one pass over the layer's activations instead of
    // two, tl_wgrad_kernel<.., DY>): decided here, once, for the packing below and the launches
    FuseShape fz[8];
    WgradShape wz[8];
    memset(fz, 0, sizeof(fz));
    for (int l = 0; l < nlayers; ++l) {
        const pn2_bn_layer &L = layers[l];
        if (ztop && l == nlayers - 1) {
            // (the z-free pooled top layer in one pass is correct and tested, but not yet faster than its three kernels -- 690 vs
            // 590 us at the metric shape: two waves carry the whole data gradient, 72 MFMAs on transposed reads each -- so the
            // size rule leaves it off; fuse_wgrad = PN2_OPT_ON forces it)
            wz[l] = wgrad_shape(rows, L.cin, top_cols(L.cin, L.cout), false, cus, PN2_OPT_OFF);
            if (o.fuse_wgrad == PN2_OPT_ON) fz[l] = fuse_shape(rows, wz[l], tiles(L.cout) * 32 + L.cin, L.cin, o, false);
        } else if (!(l == 0 && (group || !want_dx))) {
            wz[l] = wgrad_shape(rows, L.cin, L.cout, false, cus, PN2_OPT_OFF);
            fz[l] = fuse_shape(rows, wz[l], L.cout, L.cin, o);
        }
        if (fz[l].ok) { wz[l].lds_dy = fz[l].lds; wz[l].upw = fz[l].upw; }
    }
    bool ident_written = false;
    unsigned *tickets = reinterpret_cast<unsigned *>(base + pl.tickets);
    bool fold = false;
    {
        TlPackJobs jobs;                                          // W_l^T of every data-gradient GEMM, one launch
        memset(&jobs, 0, sizeof(jobs));
        int nj = 0;
        for (int l = 0; l < nlayers; ++l) {
            const pn2_bn_layer &L = layers[l];
            if ((l > 0 || want_dx) && !(ztop && l == nlayers - 1)) {      // dy_{l-1} = dz_l . W_l^T
                if (fz[l].ok) {                                   // every output tile in ONE slab (the fused pass keeps W^T resident)
                    GemmShape gg;
                    memset(&gg, 0, sizeof(gg));
                    gg.K = L.cout; gg.N = L.cin; gg.tk = fz[l].tk; gg.tn = fz[l].nt; gg.ns = fz[l].nt; gg.slabs = 1;
                    add_pack_job(jobs, nj, L.weight, L.w_stride_n, L.w_stride_k, gg, base + pl.pack[l]);
                } else if (l == 0 && group) {
                    // layer 1 of a grouped level: only the FEATURE rows of W_1 (the grouped xyz takes no gradient here);
                    // per point: the same tiles, for the GEMM over the b n points
                    const TlGather gt = make_gather(group);
                    add_pack_job(jobs, nj, L.weight + gt.feat_off * L.w_stride_k, L.w_stride_n, L.w_stride_k,
                                 gemm_shape(per_point ? (long long)gd.b * gd.n : rows, L.cout, gt.cfeat, o), base + pl.pack[l]);
                } else {
                    add_pack_job(jobs, nj, L.weight, L.w_stride_n, L.w_stride_k, gemm_shape(rows, L.cout, L.cin, o), base + pl.pack[l]);
                }
            }
        }
        if (per_point && nj > 0) {                                // the identity coefficients of layer 1's per-point weight gradient
            jobs.ident = reinterpret_cast<float *>(base + pl.l1coef);
            jobs.ident_c = layers[0].cout;
        }
        ident_written = jobs.ident != nullptr;
        fold = fold_wanted(o) && nj > 0;                          // (the pack launch zeroes the tickets)
        if (fold) jobs.tickets = tickets;
        if (int rc = launch_pack_jobs(jobs, nj, st)) return rc;
    }
    bool folded[8] = {false, false, false, false, false, false, false, false};     // layers whose backward finalisation ran inside the pass above
    // the data-gradient GEMM that sums (dy, dy z) of layer l - 1 also turns the sums into that layer's gradients and coefficients
    auto fold_below = [&](TlGemm &q, int l) {
        if (!fold || l < 1) return;
        q.fin = fin_backward(layers[l - 1], rows, q.stats, reinterpret_cast<float *>(base + pl.coef[l - 1]), tickets + (l - 1));
        folded[l - 1] = true;
    };
    float *ga = reinterpret_cast<float *>(base + pl.ga), *gb = reinterpret_cast<float *>(base + pl.gb);
    float *gq = reinterpret_cast<float *>(base + pl.gq);
    const pn2_bn_layer &T = layers[nlayers - 1];
    int nparts[8];                                      // rows of each layer's partial-sum array
    // top of the stack: dy_L and its two column sums
    if (pool_rows) {
        const long long groups = rows / pool_rows;
        long long gy = (groups + 63) / 64;
        if (gy > kMaxParts) gy = kMaxParts;
        nparts[nlayers - 1] = (int)gy;
        if (int rc = launch(tl_pool_grad_kernel, dim3((unsigned)((T.cout + 63) / 64), (unsigned)gy), dim3(256), 0, st, groups, T.cout, out,
                            grad_out, zsel, gq, reinterpret_cast<double *>(base + pl.stats[nlayers - 1]))) return rc;
    } else {
        long long gy = (rows + 255) / 256;
        if (gy > kMaxParts) gy = kMaxParts;
        nparts[nlayers - 1] = (int)gy;
        if (int rc = launch(tl_top_grad_kernel, dim3((unsigned)((T.cout + 63) / 64), (unsigned)gy), dim3(256), 0, st, rows, T.cout, out,
                            grad_out, (const float *)T.z, ga, reinterpret_cast<double *>(base + pl.stats[nlayers - 1]))) return rc;
    }
    float *gcur = ga, *gnext = gb;                      // dy of the current layer (dense case) / of the layer below
    int l1_moment_parts = 0;                            // > 0: layer 1's weight gradient comes from moments (TlWgrad::l1x)
    // weight-gradient launches on a helper stream beside the data-gradient chain (SideStream above): OPT-IN. Measured
    // (scripts/lab_ab.sh side_stream, profiles/r04/README.md): every fork / join is a cross-queue dependency of ~10 us on this
    // runtime, three pairs per level -- sem_seg SA2 280 -> 330 us, SA4 250 -> 305, FP4 335 -> 380; only the group_all level
    // (three wide layers over 4,096 rows) gains, 431 -> 379 us. The size rule therefore never switches it on.
    SideStream sd;
    if (int rc = sd.init(st, o.side_stream == PN2_OPT_ON)) return rc;
    hipStream_t ss = sd.get();
    for (int l = nlayers - 1; l >= 0; --l) {
        const pn2_bn_layer &L = layers[l];
        float *coef = reinterpret_cast<float *>(base + pl.coef[l]);
        if (!folded[l])
            if (int rc = launch(tl_bn_backward_finalize_kernel, dim3((unsigned)((L.cout + 7) / 8)), dim3(256), 0, st,
                                reinterpret_cast<const double *>(base + pl.stats[l]), nparts[l], L.cout, (double)rows, L.gamma,
                                (const float *)L.save, L.grad_gamma, L.grad_beta, coef, L.grad_accumulate)) return rc;
        if (int rc = sd.join()) return rc;                          // the helper's reads of the dy buffer this layer's data gradient overwrites
        const bool pooled_top = pool_rows && l == nlayers - 1;
        if (pooled_top && ztop) {
            // ---- the pooled top layer in terms of its input h = relu(a z_{l-1} + c): see tl_top_mats_kernel
            const pn2_bn_layer &D = layers[l - 1];
            const int K = L.cin, NF = L.cout, tf = tiles(NF), NFp = tf * 32, ld = top_cols(K, NF);
            float *wp = reinterpret_cast<float *>(base + pl.topw), *rowc = wp + (size_t)(NFp + K) * K;
            double *sf = reinterpret_cast<double *>(base + pl.topsf);
            {
                long long blocks = ((long long)(NFp + K + 1) * K + 255) / 256;
                if (blocks > 4096) blocks = 4096;
                if (int rc = launch(tl_top_mats_kernel, dim3((unsigned)blocks), dim3(256), 0, st, L.weight, L.w_stride_k, L.w_stride_n,
                                    K, NF, NFp, (const float *)coef, (const float *)nullptr, wp, rowc)) return rc;    // z_L = h W: no bias term
            }
            if (fz[l].ok) {
                // ---- ONE pass over z_{l-1}: the routed gradient and h as operand tiles of the block image serve the weight
                // gradient (S, Gram matrix, column sums: tl_top_wgrad_fix_kernel combines them) AND the data gradient
                // dy_{l-1} = [s dy routed | h] . [W^T ; -M] - r, masked by the layer below (as separate kernels z_{l-1} crossed
                // HBM in the data-gradient GEMM, in tl_top_s_kernel and in the Gram pass)
                GemmShape gg;
                memset(&gg, 0, sizeof(gg));
                gg.K = NFp + K; gg.N = K; gg.tk = fz[l].tk; gg.tn = fz[l].nt; gg.ns = fz[l].nt; gg.slabs = 1;
                if (int rc = launch_pack(wp, K, 1, gg, base + pl.pack[l], st)) return rc;
                TlWgrad w;
                memset(&w, 0, sizeof(w));
                w.rows = rows;
                w.KI = K;
                w.amode = A_RELU; w.A = D.z; w.pa = D.save + 2 * D.cout; w.pc = D.save + 3 * D.cout;
                w.dmode = A_FILL;
                w.NO = ld; w.tf = tf; w.NF = NF;
                w.G = gq; w.argsel = argsel; w.coef = coef; w.group_rows = pool_rows;
                w.partial = reinterpret_cast<float *>(base + pl.partial); w.partial_cap = pl.partial_cap;
                w.xshare = 1;                                     // one slab: the "h again" tiles are the first operand's
                w.dy_w = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
                w.xr_off = (int)fz[l].xr_off;
                w.dy_tk = fz[l].tk; w.dy_nt = fz[l].nt; w.dy_tf = tf; w.single = fz[l].single;
                w.dy_cols = K; w.dy_pitch = K;
                w.dy_out = gnext;
                w.dy_zprev = D.z; w.dy_ea = D.save + 2 * D.cout; w.dy_ec = D.save + 3 * D.cout;
                w.dy_bias = rowc;
                w.dy_stats = reinterpret_cast<double *>(base + pl.stats[l - 1]);
                w.dy_nt_store = o.nt == PN2_OPT_OFF ? 0 : o.nt == PN2_OPT_ON ? 1 : (size_t)rows * K * sizeof(float) >= ((size_t)128 << 20);
                if (int rc = launch_wgrad(w, wz[l], reinterpret_cast<float *>(base + pl.partial2), L, st, sf)) return rc;
                long long blocks = ((long long)K * NF + 255) / 256;
                if (blocks > 4096) blocks = 4096;
                if (int rc = launch(tl_top_wgrad_fix_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const double *)sf, ld, K, NF,
                                    tf * 32, tf * 32 + tiles(K) * 32, L.weight, L.w_stride_k, L.w_stride_n, (const float *)coef,
                                    (const float *)nullptr, L.grad_weight, (const double *)nullptr, L.grad_accumulate)) return rc;
                nparts[l - 1] = (int)wz[l].gridx;
                float *tmp = gcur; gcur = gnext; gnext = tmp;
                continue;
            }
            {
                // weight gradient: the routed part S on the vector units (tl_top_s_kernel) when its shape allows, the Gram
                // matrix h^T h and the column sums of h from the dense kernel, combined by tl_top_wgrad_fix_kernel;
                // data gradient: the GEMM over [routed gradient | h]. The dense kernel and the GEMM are independent passes over
                // z_{l-1}: one launch for both where the pair has a kernel (tl_pair_kernel)
                const TopSShape ts = top_s_shape(rows, pool_rows, K, NF, o);
                double *s64 = ts.ok ? reinterpret_cast<double *>(base + pl.tops64) : nullptr;
                if (int rc = sd.fork()) return rc;                 // the weight gradient's kernels beside the data gradient below
                if (ts.ok) {
                    TlTopS q;
                    memset(&q, 0, sizeof(q));
                    q.groups = rows / pool_rows; q.ns = pool_rows; q.K = K; q.NF = NF;
                    q.z = D.z; q.pa = D.save + 2 * D.cout; q.pc = D.save + 3 * D.cout;
                    q.gq = gq; q.argsel = argsel; q.coef = coef;
                    q.partial = reinterpret_cast<float *>(base + pl.tops_part);
                    if (int rc = launch_top_s(q, ts, reinterpret_cast<float *>(base + pl.tops_part2), s64, ss)) return rc;
                }
                const int tfw = ts.ok ? 0 : tf, ldw = ts.ok ? top_cols(K, 0) : ld;       // operand tiles of the dense kernel
                TlWgrad w;
                memset(&w, 0, sizeof(w));
                w.rows = rows;
                w.KI = K;
                w.amode = A_RELU; w.A = D.z; w.pa = D.save + 2 * D.cout; w.pc = D.save + 3 * D.cout;
                w.dmode = A_FILL;
                w.NO = ldw; w.tf = tfw; w.NF = ts.ok ? 0 : NF;
                w.G = gq; w.argsel = argsel; w.coef = coef; w.group_rows = pool_rows;
                w.partial = reinterpret_cast<float *>(base + pl.partial); w.partial_cap = pl.partial_cap;
                const WgradShape ws_ = wgrad_shape(rows, K, ldw, false, cus, o.wgrad_two_per_cu);
                w.xshare = ws_.uslabs == 1;
                const GemmShape g = gemm_shape(rows, NFp + K, K, o);
                if (int rc = launch_pack(wp, K, 1, g, base + pl.pack[l], st)) return rc;
                TlGemm p;
                memset(&p, 0, sizeof(p));
                p.rows = rows;
                p.tk0 = tf; p.K0 = NF; p.K1 = K;
                p.G = gq; p.argsel = argsel; p.p0 = coef; p.group_rows = pool_rows;
                p.A2 = D.z; p.q0 = D.save + 2 * D.cout; p.q1 = D.save + 3 * D.cout;
                p.wpacked = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
                p.bias = rowc;
                p.emode = E_MASK;
                p.out = gnext;
                p.zprev = D.z; p.ea = D.save + 2 * D.cout; p.ec = D.save + 3 * D.cout;
                p.stats = reinterpret_cast<double *>(base + pl.stats[l - 1]);
                fold_below(p, l);
                int np = 0, rc = kNoPair;
                if (pair_wanted(rows, o)) rc = launch_pair(A_FILL, p, g, w, ws_, reinterpret_cast<float *>(base + pl.partial2), L, st, o, &np, sf);
                if (rc == kNoPair) {
                    if ((rc = launch_wgrad(w, ws_, reinterpret_cast<float *>(base + pl.partial2), L, ss, sf))) return rc;
                    rc = launch_gemm(A_FILL, p, g, st, o, &np);
                }
                if (rc) return rc;
                nparts[l - 1] = np;
                long long blocks = ((long long)K * NF + 255) / 256;
                if (blocks > 4096) blocks = 4096;
                if (int rc2 = launch(tl_top_wgrad_fix_kernel, dim3((unsigned)blocks), dim3(256), 0, ss, (const double *)sf, ldw, K, NF,
                                     tfw * 32, tfw * 32 + tiles(K) * 32, L.weight, L.w_stride_k, L.w_stride_n, (const float *)coef,
                                     (const float *)nullptr, L.grad_weight, (const double *)s64, L.grad_accumulate)) return rc2;
            }
            float *tmp = gcur; gcur = gnext; gnext = tmp;
            continue;
        }
        if (l == 0 && coords_only && !(gd.cfeat > 0 && want_dx)) {
            // ---- a level without features (or with a few whose gradient nobody wants -- the network's input normals): the
            // first layer's weight gradient on the vector units
            if (l1_moment_parts) {                                // ... from x^T dy_1 of the pass above and the moments of x
                const TlGather gt = make_gather(group);
                if (int rc = launch(tl_l1_wx_combine_kernel, dim3((unsigned)((3 * L.cout + 7) / 8)), dim3(256), 0, st,
                                    reinterpret_cast<const double *>(base + pl.l1mom), l1_moment_parts,
                                    reinterpret_cast<const double *>(base + pl.l1a), nparts[0], L.cout, L.cout, (const float *)coef,
                                    L.weight + gt.xyz_off * L.w_stride_k, L.w_stride_k, L.w_stride_n,
                                    L.grad_weight + gt.xyz_off * L.w_stride_k, L.grad_accumulate)) return rc;
                break;
            }
            // ... in one pass over dy_1 and z_1
            if (int rc = launch_l1_dz(rows, gd, group, L, gcur, coef, reinterpret_cast<float *>(base + pl.l1part), false, st)) return rc;
            break;
        }
        if (l == 0 && per_point) {
            // ---- layer 1 once per point (see tl_l1_forward_kernel): dz_1 and dW1x in one pass over the rows, the scatter of
            // dz_1 onto the points, then two GEMMs over the b n points
            const TlGather gt = make_gather(group);
            const long long bn = (long long)gd.b * gd.n;
            float *S = reinterpret_cast<float *>(base + pl.l1p), *part = reinterpret_cast<float *>(base + pl.l1part);
            float *ident = reinterpret_cast<float *>(base + pl.l1coef);
            if (int rc = launch_l1_dz(rows, gd, group, L, gcur, coef, part, true, st)) return rc;
            if (int rc = pn2_group_point_grad_seg(gd.b, gd.n, L.cout, gd.m, gd.nsample, gcur, group->idx, S, base + pl.l1seg,
                                                  reproducible, stream)) return rc;
            if (!ident_written)
                if (int rc = launch(tl_identity_coef_kernel, dim3((unsigned)((3 * L.cout + 127) / 128)), dim3(128), 0, st, L.cout, ident)) return rc;
            {
                TlWgrad w;                                        // dW1f = points^T S
                memset(&w, 0, sizeof(w));
                w.rows = bn; w.KI = gt.cfeat; w.amode = A_PLAIN; w.A = group->points;
                w.dmode = A_DZ; w.NO = L.cout; w.Z = S; w.G = S; w.coef = ident;
                w.partial = reinterpret_cast<float *>(base + pl.partial); w.partial_cap = pl.partial_cap;
                pn2_bn_layer Lf = L;
                Lf.grad_weight = L.grad_weight + gt.feat_off * L.w_stride_k;
                const WgradShape ws_ = wgrad_shape(bn, gt.cfeat, L.cout, false, cus, o.wgrad_two_per_cu);
                if (!want_dx) {
                    if (int rc = launch_wgrad(w, ws_, reinterpret_cast<float *>(base + pl.partial2), Lf, st)) return rc;
                    break;
                }
                const GemmShape g = gemm_shape(bn, L.cout, gt.cfeat, o);      // dPoints = S W1f^T: beside it, in one launch where the pair has a kernel
                TlGemm p;
                memset(&p, 0, sizeof(p));
                p.rows = bn;
                p.A = S;
                p.wpacked = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
                p.emode = E_PLAIN;
                p.out = grad_points; p.out_pitch = gt.cfeat; p.col0 = 0; p.col1 = gt.cfeat;
                int rc = kNoPair;
                if (pair_wanted(rows, o)) rc = launch_pair(A_PLAIN, p, g, w, ws_, reinterpret_cast<float *>(base + pl.partial2), Lf, st, o, nullptr);
                if (rc == kNoPair) {
                    if ((rc = sd.fork())) return rc;
                    if ((rc = launch_wgrad(w, ws_, reinterpret_cast<float *>(base + pl.partial2), Lf, ss))) return rc;
                    rc = launch_gemm(A_PLAIN, p, g, st, o);
                }
                if (rc) return rc;
            }
            break;
        }
        // weight gradient
        {
            TlWgrad w;
            memset(&w, 0, sizeof(w));
            w.rows = rows;
            w.KI = L.cin;
            if (l == 0 && group) { w.amode = A_GATHER; w.g = make_gather(group); }
            else if (l == 0) { w.amode = A_PLAIN; w.A = x; }
            else { w.amode = A_RELU; w.A = layers[l - 1].z; w.pa = layers[l - 1].save + 2 * layers[l - 1].cout; w.pc = layers[l - 1].save + 3 * layers[l - 1].cout; }
            w.dmode = pooled_top ? A_DZ_POOL : A_DZ;
            w.NO = L.cout;
            w.Z = L.z;
            w.G = pooled_top ? gq : gcur;
            w.argsel = argsel;
            w.coef = coef;
            w.group_rows = pool_rows;
            w.partial = reinterpret_cast<float *>(base + pl.partial); w.partial_cap = pl.partial_cap;
            if (fz[l].ok) {
                // ... and the data gradient in the same pass over (dy_l, z_l, z_{l-1}), see tl_wgrad_kernel
                w.dy_w = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
                w.dy_tk = fz[l].tk; w.dy_nt = fz[l].nt; w.dy_tf = 0; w.single = fz[l].single; w.dy_acopy = fz[l].acopy;
                w.dy_cols = L.cin; w.dy_pitch = L.cin;
                if (l > 0) {
                    const pn2_bn_layer &D = layers[l - 1];
                    w.dy_out = gnext;
                    w.dy_zprev = D.z; w.dy_ea = D.save + 2 * D.cout; w.dy_ec = D.save + 3 * D.cout;
                    w.dy_stats = reinterpret_cast<double *>(base + pl.stats[l - 1]);
                } else {
                    w.dy_out = grad_x;
                }
                w.dy_nt_store = o.nt == PN2_OPT_OFF ? 0 : o.nt == PN2_OPT_ON ? 1 : (size_t)rows * L.cin * sizeof(float) >= ((size_t)128 << 20);
                w.xr_off = (int)fz[l].xr_off;
                if (l == 1 && coords_only && gd.cfeat == 0 && !pooled_top) {            // (a pooled two-layer stack keeps the pass over dy_1: its dz comes from the routed gradient)
                    // the layer below takes the three centred coordinates: dy_1 is wanted only as x^T dy_1 (TlWgrad::l1x) -- never
                    // written, and tl_l1_dz_kernel's pass over (dy_1, z_1) is replaced by nine moments of x
                    const pn2_bn_layer &D = layers[0];
                    TlL1 q;
                    memset(&q, 0, sizeof(q));
                    q.rows = rows; q.n = gd.n; q.m = gd.m; q.nsample = gd.nsample; q.C = D.cout;
                    q.xyz = group->xyz; q.new_xyz = group->new_xyz; q.idx = group->idx;
                    long long xb = (rows + kL1XrowsThreads - 1) / kL1XrowsThreads;
                    if (xb > kMaxParts) xb = kMaxParts;
                    l1_moment_parts = (int)xb;
                    if (int rc = launch(tl_l1_xrows_kernel, dim3((unsigned)xb), dim3(kL1XrowsThreads), 0, st, q, reinterpret_cast<float4 *>(base + pl.l1xg),
                                        reinterpret_cast<double *>(base + pl.l1mom))) return rc;
                    w.l1x = reinterpret_cast<const float4 *>(base + pl.l1xg);
                    w.l1a = reinterpret_cast<double *>(base + pl.l1a);
                    w.dy_out = nullptr;
                }
                if (int rc = launch_wgrad(w, wz[l], reinterpret_cast<float *>(base + pl.partial2), L, st)) return rc;
                if (l > 0) nparts[l - 1] = (int)wz[l].gridx;
                float *tmp = gcur; gcur = gnext; gnext = tmp;
                continue;
            }
            const WgradShape ws_ = wgrad_shape(rows, L.cin, L.cout, w.amode == A_GATHER, cus, o.wgrad_two_per_cu);
            if (!(l > 0 || want_dx)) {                             // no data gradient below this layer
                if (int rc = launch_wgrad(w, ws_, reinterpret_cast<float *>(base + pl.partial2), L, st)) return rc;
            } else {
                // data gradient: independent of the weight gradient -- ONE launch for both where the pair has a kernel
                // (tl_pair_kernel), else the weight gradient first (on the helper stream when that is asked for)
                const GemmShape g = (l == 0 && group) ? gemm_shape(rows, L.cout, make_gather(group).cfeat, o) : gemm_shape(rows, L.cout, L.cin, o);
                TlGemm p;
                memset(&p, 0, sizeof(p));
                p.rows = rows;
                p.A = L.z;
                p.G = pooled_top ? gq : gcur;
                p.argsel = argsel;
                p.p0 = coef; p.p1 = coef + L.cout; p.p2 = coef + 2 * L.cout;
                p.group_rows = pool_rows;
                p.wpacked = reinterpret_cast<const u32x4 *>(base + pl.pack[l]);
                if (l > 0) {
                    const pn2_bn_layer &D = layers[l - 1];
                    p.emode = E_MASK;
                    p.out = gnext;
                    p.zprev = D.z;
                    p.ea = D.save + 2 * D.cout;
                    p.ec = D.save + 3 * D.cout;
                    p.stats = reinterpret_cast<double *>(base + pl.stats[l - 1]);
                    fold_below(p, l);
                } else {
                    p.emode = E_PLAIN;
                    if (group) {
                        const TlGather gt = make_gather(group);
                        p.out = grad_feat_rows; p.out_pitch = gt.cfeat; p.col0 = 0; p.col1 = gt.cfeat;
                    } else {
                        p.out = grad_x; p.out_pitch = L.cin; p.col0 = 0; p.col1 = L.cin;
                    }
                }
                const int amode = pooled_top ? A_DZ_POOL : A_DZ;
                int np = 0, rc = kNoPair;
                if (pair_wanted(rows, o)) rc = launch_pair(amode, p, g, w, ws_, reinterpret_cast<float *>(base + pl.partial2), L, st, o, &np);
                if (rc == kNoPair) {
                    if ((rc = sd.fork())) return rc;               // beside the data-gradient GEMM
                    if ((rc = launch_wgrad(w, ws_, reinterpret_cast<float *>(base + pl.partial2), L, ss))) return rc;
                    rc = launch_gemm(amode, p, g, st, o, &np);
                }
                if (rc) return rc;
                if (l > 0) nparts[l - 1] = np;
            }
        }
        float *tmp = gcur; gcur = gnext; gnext = tmp;
    }
    return sd.join();                                              // everything of this call is ordered before what the caller enqueues next
}
