// ubench_xchg.hip -- what the pieces of an FPS round's arg-max exchange cost on gfx950 with one wave per SIMD (development
// aid, round 5). Every kernel runs a DEPENDENT chain of one piece (32 workgroups of 256 threads, as the FPS chain does) and
// reports ns and shader cycles per iteration. hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench_xchg.hip -o build_lab/ubench_xchg
#include "../pointnet2_amd/csrc/fps_body.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using namespace pn2;

constexpr int ITERS = 8192;

template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32_step(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
    return v > o ? v : o;      // hipcc folds this into one v_max_u32_dpp (+ s_nop 1 for the DPP read-after-write hazard)
}
__device__ __forceinline__ unsigned wave_max_u32_lane63(unsigned v)    // same step sequence as wave_max_f64_lane63
{
    v = dpp_max_u32_step<0xB1>(v);
    v = dpp_max_u32_step<0x4E>(v);
    v = dpp_max_u32_step<0x141>(v);
    v = dpp_max_u32_step<0x140>(v);
    v = dpp_max_u32_step<0x142>(v);
    v = dpp_max_u32_step<0x143>(v);
    return v;
}

__device__ __forceinline__ double max64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int KIND>
__global__ __launch_bounds__(256) void k(float *sink, unsigned long long *ticks, int iters)
{
    __shared__ float4 mirror[1024];
    __shared__ double keys[2][4];
    __shared__ float4 recs[2][4];
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 1024; i += 256) mirror[i] = make_float4(i * 0.5f, i * 0.25f, i, __int_as_float((i * 7 + 1) & 1023));
    if (t < 8) (&keys[0][0])[t] = 1.0 + t;
    __syncthreads();
    double kd = __hiloint2double(0x3f800000 + t, 1023 - t);
    unsigned u = 0x3f800000u + t;
    unsigned cur = t & 1023;
    float acc = 0.f;
    // pruned-tier state of kinds 12-15: one box per lane (lane l = group l; most are far from any sample), one group's slots
    const float blx = (lane & 3) * 300.f, bly = (lane >> 2 & 3) * 300.f, blz = (lane >> 4) * 300.f, bhx = blx + 200.f, bhy = bly + 200.f, bhz = blz + 1100.f;
    pn2_f2 gx = {t * 1.f, t * 2.f}, gy = {t * .5f, t * .25f}, gz = {1.f * lane, 2.f * lane};
    float md0 = 1e30f, md1 = 1e30f, sx = 1.f, sy = 2.f, sz = 3.f, vstar = 100.f;
    const double c1 = 1.0 + t, c2 = 2.0 + t, c3 = 3.0 + t;
    int kprev = 0;
    int *gout = reinterpret_cast<int *>(sink) + 32 * 256 + blockIdx.x * 1024;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        const int par = i & 1;
        if (KIND == 0) {                       // 64-bit DPP ladder
            kd = wave_max_f64_lane63(kd);
        } else if (KIND == 1) {                // 32-bit DPP ladder (v_max_u32_dpp)
            u = wave_max_u32_lane63(u) + 1u;
        } else if (KIND == 2) {                // dependent broadcast ds_read_b128
            const float4 s = mirror[cur];
            cur = (unsigned)__float_as_int(s.w) & 1023u;
            acc += s.x;
        } else if (KIND == 3) {                // dependent broadcast ds_read_b64
            const double s = reinterpret_cast<const double *>(mirror)[cur * 2 + 1];
            cur = (unsigned)__double2hiint(s) & 1023u;
        } else if (KIND == 4) {                // dependent broadcast ds_read_b32
            cur = (unsigned)__float_as_int(mirror[cur].w) & 1023u;
        } else if (KIND == 5 || KIND == 6 || KIND == 8) {    // the product's exchange (5), without the mirror read (6), write + barrier only (8)
            if (lane == 63) keys[par][w] = kd;
            __syncthreads();
            if (KIND != 8) {
                const double k0 = keys[par][0], k1 = keys[par][1], k2 = keys[par][2], k3 = keys[par][3];
                const double m = max64(max64(k0, k1), max64(k2, k3));
                if (KIND == 5) {
                    const float4 s = mirror[(unsigned)__double2loint(m) & 1023u];
                    kd = __hiloint2double(__double2hiint(kd), __float_as_int(s.w) + lane);
                } else {
                    kd = __hiloint2double(__double2hiint(kd), (__double2loint(m) + lane) & 1023);
                }
            } else {
                kd = __hiloint2double(__double2hiint(kd), (__double2loint(kd) + 1) & 1023);
            }
        } else if (KIND == 7) {                // barrier only
            __syncthreads();
        } else if (KIND == 9) {                // the whole serial part of a round at P = 4 slots without the update: lane max, ladder, exchange
            kd = wave_max_f64_lane63(kd);
            if (lane == 63) keys[par][w] = kd;
            __syncthreads();
            const double k0 = keys[par][0], k1 = keys[par][1], k2 = keys[par][2], k3 = keys[par][3];
            const double m = max64(max64(k0, k1), max64(k2, k3));
            const float4 s = mirror[(unsigned)__double2loint(m) & 1023u];
            kd = __hiloint2double(0x3f800000 + ((__float_as_int(s.w) * (lane + 3)) & 0xffff), (__float_as_int(s.w) + t) & 1023);
        } else if (KIND >= 12 && KIND <= 15) { // kind 9 + the pruned tier's other pieces, one at a time
            // 12: + box test of 32 groups and the wave's touched bits; 13: + one group's update and the lane key; 14: + thread 0's
            // index store; 15: 13 with the store issued under the key reads' latency
            const float thr = __fadd_rn(__fmul_rn(vstar, 1.00001f), 1e-30f);
            const float ax = __fsub_rn(sx, __builtin_amdgcn_fmed3f(sx, blx, bhx));
            const float ay = __fsub_rn(sy, __builtin_amdgcn_fmed3f(sy, bly, bhy));
            const float az = __fsub_rn(sz, __builtin_amdgcn_fmed3f(sz, blz, bhz));
            const float bd = __fadd_rn(__fadd_rn(__fmul_rn(ax, ax), __fmul_rn(ay, ay)), __fmul_rn(az, az));
            const unsigned long long far_mask = __ballot(bd >= thr);
            const unsigned mybits = (unsigned)(~far_mask >> (w * 8)) & 0xffu;
            if (mybits != 0u) {
                if (KIND >= 13) {
                    pn2_f2 sxy = {sx, sy}, syy = {sy, 0.f}, szk = {sz, 0.f};
                    pn2_f2 dx = pk_sub_bcast_lo(gx, sxy), dy = pk_sub_bcast_lo(gy, syy), dz = pk_sub_bcast_lo(gz, szk);
                    dx = pk_mul(dx, dx); dy = pk_mul(dy, dy); dz = pk_mul(dz, dz);
                    dx = pk_add(dx, dy); dx = pk_add(dx, dz);
                    md0 = vmin_f32(dx.x, md0); md1 = vmin_f32(dx.y, md1);
                    double k0 = __hiloint2double(__float_as_int(md0), 2 * t), k1 = __hiloint2double(__float_as_int(md1), 2 * t + 1);
                    const double g0 = max64(k0, k1);
                    const double p0 = max64(g0, c1), p1 = max64(p0, c2);
                    kd = max64(p1, c3);
                    md0 += 1.0f; md1 += 0.5f;
                }
                kd = wave_max_f64_lane63(kd);
            }
            if (lane == 63) keys[par][w] = kd;
            __syncthreads();
            const double k0 = keys[par][0], k1 = keys[par][1], k2 = keys[par][2], k3 = keys[par][3];
            if (KIND == 15 && t == 0) gout[i & 1023] = kprev;
            const double m = max64(max64(k0, k1), max64(k2, k3));
            const float4 s = mirror[(unsigned)__double2loint(m) & 1023u];
            sx = s.x; sy = s.y; sz = s.z; vstar = __int_as_float(__double2hiint(m));
            kprev = __float_as_int(s.w);
            if (KIND == 14 && t == 0) gout[i & 1023] = kprev;
            kd = __hiloint2double(0x3f800000 + ((__float_as_int(s.w) * (lane + 3)) & 0xffff), (__float_as_int(s.w) + t) & 1023);
        } else if (KIND == 10) {               // readlane -> compare -> exec-masked LDS write
            const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)u, 63);
            if (u == v + (unsigned)lane - 63u) keys[par][w] = kd;
            u += 1u;
        } else if (KIND == 11) {               // six broadcast b128 reads + register select (the rejected one-trip exchange's read side)
            const double k0 = keys[par][0], k1 = keys[par][1], k2 = keys[par][2], k3 = keys[par][3];
            const float4 r0 = recs[par][0], r1 = recs[par][1], r2 = recs[par][2], r3 = recs[par][3];
            const double m01 = max64(k0, k1), m23 = max64(k2, k3), mm = max64(m01, m23);
            const bool c01 = __double2loint(k1) == __double2loint(m01), c23 = __double2loint(k3) == __double2loint(m23);
            const bool cc = __double2loint(m23) == __double2loint(mm);
            const float4 a = c01 ? r1 : r0, b = c23 ? r3 : r2;
            const float4 s = cc ? b : a;
            acc += s.x + s.y + s.z;
            if (acc == 12345.f) keys[par][w] = kd;     // keeps the loads inside the loop
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0 && blockIdx.x == 0) *ticks = t1 - t0;
    sink[blockIdx.x * 256 + t] = acc + (float)kd + u + cur;
}

template <int KIND>
static void bench(const char *name)
{
    float *sink; unsigned long long *ticks;
    CK(hipMalloc(&sink, 32 * 256 * 4 + 32 * 1024 * 4)); CK(hipMalloc(&ticks, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(32), dim3(256), 0, 0, sink, ticks, ITERS);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(32), dim3(256), 0, 0, sink, ticks, ITERS * 8);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    const double ns = ms * 1e6 / (ITERS * 8.0);
    printf("%-78s : %7.1f ns  %7.1f shader cycles per iteration\n", name, ns, (double)h / (ITERS * 8.0));
    CK(hipFree(sink)); CK(hipFree(ticks));
}

int main()
{
    bench<0>("64-bit DPP ladder (6 x [2 v_mov_dpp + v_max_f64])");
    bench<1>("32-bit DPP ladder (6 x v_max_u32_dpp) + 1 add");
    bench<2>("dependent broadcast ds_read_b128");
    bench<3>("dependent broadcast ds_read_b64");
    bench<4>("dependent broadcast ds_read_b32");
    bench<7>("s_barrier only");
    bench<8>("lane-63 ds_write_b64 + waitcnt + s_barrier");
    bench<6>("... + 2 broadcast ds_read_b128 of the keys + 3 v_max_f64");
    bench<5>("... + dependent ds_read_b128 of the mirror (the product's exchange)");
    bench<9>("ladder + the product's exchange (a round without the update)");
    bench<12>("a round without the update + box test of the groups, touched bits");
    bench<13>("... + one group's update (GS = 2), lane key");
    bench<14>("... + thread 0 stores the index after the mirror read");
    bench<15>("... the same store issued under the key reads instead");
    bench<10>("v_readlane -> v_cmp -> exec-masked ds_write_b64");
    bench<11>("6 broadcast ds_read_b128 + 3 v_max_f64 + 3 v_cmp + 12 v_cndmask");
    return 0;
}
