// ubench_mfma_valu.hip -- do vector-ALU instructions hide under v_mfma_f32_32x32x16_bf16 on gfx950, and at what clock?
// The fused MLP kernels (csrc/sa_mlp*.hip) execute ~4.5 VALU instructions (the bf16 level split, ReLU, pooling) per MFMA and
// run at the SUM of their MFMA time and their non-MFMA time (profiles/r03/DESIGN_round3.md section 4.6a). This probe issues
// the same mix in controlled orders on every SIMD of the chip:
//   mfma      6 dependent MFMAs per trip (the x6 product chain), nothing else
//   valu      K VALU instructions per MFMA slot, no MFMA (the split's instruction kinds: cvt_pk, shift, and, sub)
//   inter     the same wave alternates: MFMA, K VALU, MFMA, K VALU ... (independent registers; the order is pinned)
//   burst     the same wave: 6 MFMAs, then 6 K VALU (what the kernels do today: a layer's MFMAs, then its split)
//   spec      wave-specialised: the first wave of each SIMD only MFMAs, the second only VALU (2 waves per SIMD)
//   inter_acc like inter, but the VALU instructions READ registers that an MFMA wrote long ago (a second accumulator, idle
//             during the loop) -- what a level split of a finished layer does
//   mfma_lds / inter_lds: the MFMAs' A operands come from LDS, three ds_read_b128 per trip issued one trip ahead (the weight
//             reads of the kernels: 3 KB per wave and six MFMAs), without / with the K VALU instructions per MFMA
//   inter_wop like inter, but the VALU instructions WRITE registers that a LATER MFMA reads as its B operand (the split's results)
// Times are whole-launch HIP-event times per trip with all 256 CUs busy (the clock under load is part of the answer) and
// s_memtime ticks (100 MHz) of wave 0.  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_mfma_valu.hip -o build_lab/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { MFMA = 0, VALU = 1, INTER = 2, BURST = 3, SPEC = 4, INTER_ACC = 5, INTER_WOP = 6, MFMA_LDS = 7, INTER_LDS = 8 };

// K VALU instructions of the split's kinds on registers the MFMAs do not touch; asm volatile: a scheduling barrier,
// so the MFMA builtins between two calls stay where they are written
template <int K>
__device__ __forceinline__ void valu(float (&a)[4], float &b, unsigned (&p)[4], unsigned (&q)[4], int &phase)
{
    // four independent register sets, consecutive instructions on different sets (the split of 16 values has that much
    // parallelism); the kind of instruction advances every four
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int set = (phase + i) & 3, kind = ((phase + i) >> 2) & 3;
        switch (kind) {
        case 0: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[set]) : "v"(a[set]), "v"(b)); break;
        case 1: asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(q[set]) : "v"(p[set])); break;
        case 2: asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(p[set]) : "v"(p[set])); break;
        default: asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[set]) : "v"(a[set]), "v"(__uint_as_float(q[set]))); break;
        }
    }
    phase = (phase + K) & 15;
}

template <int MODE, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *sink, unsigned long long *ticks, int iters)
{
    const int t = threadIdx.x;
    __shared__ u32x4 lds_w[26 * 3 * 64];                           // 78 KB: the resident kernel's weights of one item (26 steps x 3 levels)
    if (MODE == MFMA_LDS || MODE == INTER_LDS)
        for (int i = t; i < 26 * 3 * 64; i += 64 * WAVES) lds_w[i] = u32x4{0x3f803f80u + (unsigned)i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    f32x16 acc = {0};
    u32x4 w[3], x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { w[i] = u32x4{0x3f803f80u + t, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i}; x[i] = u32x4{0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u + t}; }
    float a[4] = {1.0f + t, 2.0f + t, 3.0f + t, 4.0f + t}, b = 2.0f + t;
    unsigned p[4] = {(unsigned)t, 1u, 2u, 3u}, q[4] = {(unsigned)t + 1u, 5u, 6u, 7u};
    f32x16 acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[0]), __builtin_bit_cast(bf16x8, x[0]), acc2, 0, 0, 0);     // MFMA-written, then idle
    const bool mfma_wave = MODE != SPEC || (t >> 6) < WAVES / 2;      // SPEC: waves 0..3 (one per SIMD) run the MFMAs, 4..7 the VALU work
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        int phase = 0;                                             // compile-time through the unrolled trip (6 K is a multiple of 4 for even K)
#define M(WL, XL) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[WL]), __builtin_bit_cast(bf16x8, x[XL]), acc, 0, 0, 0)
        if (MODE == MFMA) { M(0, 2); M(1, 1); M(2, 0); M(0, 1); M(1, 0); M(0, 0); }
        if (MODE == VALU) { valu<6 * K>(a, b, p, q, phase); }
        if (MODE == INTER) {
            M(0, 2); valu<K>(a, b, p, q, phase); M(1, 1); valu<K>(a, b, p, q, phase); M(2, 0); valu<K>(a, b, p, q, phase);
            M(0, 1); valu<K>(a, b, p, q, phase); M(1, 0); valu<K>(a, b, p, q, phase); M(0, 0); valu<K>(a, b, p, q, phase);
        }
        if (MODE == INTER_ACC) {                                   // the cvt of every fourth VALU instruction reads an MFMA-written register
#define VA(i0) { a[0] = acc2[(i0) & 15]; a[1] = acc2[((i0) + 1) & 15]; a[2] = acc2[((i0) + 2) & 15]; a[3] = acc2[((i0) + 3) & 15]; valu<K>(a, b, p, q, phase); }
            M(0, 2); VA(0) M(1, 1); VA(4) M(2, 0); VA(8) M(0, 1); VA(12) M(1, 0); VA(2) M(0, 0); VA(6)
#undef VA
        }
        if (MODE == INTER_WOP) {                                   // the VALU results become the B operand of the MFMA two slots later
#define VW(XL) { valu<K>(a, b, p, q, phase); x[XL][1] = p[0]; x[XL][2] = q[1]; }
            M(0, 2); VW(0) M(1, 1); VW(2) M(2, 0); VW(1) M(0, 1); VW(0) M(1, 0); VW(1) M(0, 0); VW(2)
#undef VW
        }
        if (MODE == MFMA_LDS || MODE == INTER_LDS) {
            const u32x4 *w4 = lds_w + (t & 63);
            const int step = (it + 1) % 26;
            u32x4 nw[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) nw[l] = w4[(step * 3 + l) * 64];      // next trip's weights
            asm volatile("" ::: "memory");
            if (MODE == MFMA_LDS) { M(0, 2); M(1, 1); M(2, 0); M(0, 1); M(1, 0); M(0, 0); }
            else {
                M(0, 2); valu<K>(a, b, p, q, phase); M(1, 1); valu<K>(a, b, p, q, phase); M(2, 0); valu<K>(a, b, p, q, phase);
                M(0, 1); valu<K>(a, b, p, q, phase); M(1, 0); valu<K>(a, b, p, q, phase); M(0, 0); valu<K>(a, b, p, q, phase);
            }
#pragma unroll
            for (int l = 0; l < 3; ++l) w[l] = nw[l];
        }
        if (MODE == BURST) { M(0, 2); M(1, 1); M(2, 0); M(0, 1); M(1, 0); M(0, 0); valu<6 * K>(a, b, p, q, phase); }
        if (MODE == SPEC) {
            if (mfma_wave) { M(0, 2); M(1, 1); M(2, 0); M(0, 1); M(1, 0); M(0, 0); }
            else { valu<6 * K>(a, b, p, q, phase); }
        }
#undef M
        asm volatile("" : "+v"(w[0]), "+v"(x[0]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    float s = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += a[i] + __uint_as_float(p[i]) + __uint_as_float(q[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    sink[blockIdx.x * 64 * WAVES + t] = s;
}

template <int MODE, int K, int WAVES> static void run(const char *name)
{
    float *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 256 * 64 * WAVES * 4)); CK(hipMalloc(&ticks, 16));
    const int iters = 8192;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE, K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, sink, ticks, iters); CK(hipDeviceSynchronize()); }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, sink, ticks, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    // one trip = 6 MFMA slots (+ 6 K VALU); ns per trip of the whole launch, and the clock the wave saw: a dependent MFMA is 32 cycles
    printf("%-58s %7.1f ns per trip (launch)  %7.1f ns per trip (wave 0, s_memtime)\n", name, ms * 1e6 / iters, (double)h * 10.0 / iters);
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main()
{
    printf("one trip = 6 MFMA slots of 32 cycles (80 ns at 2.4 GHz, 97 ns at 1.97 GHz) and / or 6 K VALU instructions\n");
    run<MFMA, 0, 4>("mfma   1 wave/SIMD");
    run<MFMA, 0, 8>("mfma   2 waves/SIMD");
    run<VALU, 4, 4>("valu   K=4 1 wave/SIMD");
    run<VALU, 4, 8>("valu   K=4 2 waves/SIMD");
    run<INTER, 2, 4>("inter  K=2 1 wave/SIMD");
    run<INTER, 4, 4>("inter  K=4 1 wave/SIMD");
    run<INTER, 6, 4>("inter  K=6 1 wave/SIMD");
    run<INTER, 4, 8>("inter  K=4 2 waves/SIMD");
    run<INTER, 6, 8>("inter  K=6 2 waves/SIMD");
    run<INTER_ACC, 4, 4>("inter_acc K=4 1 wave/SIMD (VALU reads MFMA-written regs)");
    run<INTER_ACC, 4, 8>("inter_acc K=4 2 waves/SIMD");
    run<INTER_WOP, 4, 4>("inter_wop K=4 1 wave/SIMD (VALU writes later MFMA operands)");
    run<INTER_WOP, 4, 8>("inter_wop K=4 2 waves/SIMD");
    run<MFMA_LDS, 0, 4>("mfma_lds  1 wave/SIMD (3 ds_read_b128 per 6 MFMAs)");
    run<MFMA_LDS, 0, 8>("mfma_lds  2 waves/SIMD");
    run<INTER_LDS, 4, 4>("inter_lds K=4 1 wave/SIMD");
    run<INTER_LDS, 4, 8>("inter_lds K=4 2 waves/SIMD");
    run<BURST, 4, 4>("burst  K=4 1 wave/SIMD");
    run<BURST, 4, 8>("burst  K=4 2 waves/SIMD");
    run<BURST, 6, 8>("burst  K=6 2 waves/SIMD");
    run<SPEC, 4, 8>("spec   K=4 (MFMA wave + VALU wave per SIMD)");
    run<SPEC, 8, 8>("spec   K=8 (MFMA wave + VALU wave per SIMD)");
    return 0;
}
