// fps_spec_lab.hip -- the speculative FPS tier against the plain one: same output, ns per SAMPLE.
// Development aid. hipcc <product flags> scripts/fps_spec_lab.hip -o build_lab/fps_spec_lab
#include "../pointnet2_amd/csrc/fps.hip"
#include "fps_spec_body.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
namespace pn2 {
template <int T, int P>
__global__ __launch_bounds__(T) void fps_spec_kernel(int n, int m, int Q, const float *__restrict__ xyz,
                                                     int *__restrict__ out, float *__restrict__ out_xyz)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fps_spec_body<T, P, false>(n, m, Q, blockIdx.x, xyz, out, out_xyz, nullptr, smem);
}
template <int T, int P>
static int launch_spec(int b, int n, int m, int Q, const float *inp, int *out, hipStream_t st)
{
    const size_t lds = 256 + sizeof(float4) * (size_t)T * P;
    auto kern = fps_spec_kernel<T, P>;
    if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), lds, st, n, m, Q, inp, out, nullptr);
    return launch_status();
}
}  // namespace pn2
static int pn2_debug_fps_spec(int T, int P, int b, int n, int m, const float *inp, int *out, void *)
{
    const int Q = (n + pn2::kRefThreads - 1) / pn2::kRefThreads;
#define PN2_SPEC_CASE(TT, PP) if (T == TT && P == PP) return pn2::launch_spec<TT, PP>(b, n, m, Q, inp, out, nullptr)
    PN2_SPEC_CASE(256, 4); PN2_SPEC_CASE(256, 8); PN2_SPEC_CASE(256, 16); PN2_SPEC_CASE(256, 32);
    PN2_SPEC_CASE(512, 2); PN2_SPEC_CASE(512, 4); PN2_SPEC_CASE(512, 8); PN2_SPEC_CASE(512, 16);
#undef PN2_SPEC_CASE
    return -1;
}

int main()
{
    const int b = 32;
    for (int n : {1024, 2048, 4096, 8192}) {
        const int m = n / 4;
        std::vector<float> h((size_t)b * n * 3);
        uint32_t s = 12345u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
        // make the last cloud tie-heavy: coordinates on a coarse lattice, with duplicates
        for (size_t i = (size_t)(b - 1) * n * 3; i < h.size(); ++i) h[i] = floorf(h[i] * 6.0f) * 0.125f;
        float *d_xyz; int *d_out;
        CK(hipMalloc(&d_xyz, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)b * m * 4));
        CK(hipMemcpy(d_xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<int> ref((size_t)b * m), got((size_t)b * m);
        for (int T : {256, 512}) {
            const int P = n / T;
            if (P < 1 || P > 32 || (long long)T * P > 8192) continue;
            for (int spec = 0; spec < 2; ++spec) {
                auto run = [&]() { return spec ? pn2_debug_fps_spec(T, P, b, n, m, d_xyz, d_out, nullptr)
                                               : pn2_farthest_point_sample_ex(T, P, b, n, m, d_xyz, d_out, nullptr); };
                CK(hipMemset(d_out, 0xff, (size_t)b * m * 4));
                if (run()) { printf("n=%d T=%d P=%d spec=%d: launch refused\n", n, T, P, spec); continue; }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
                if (!spec) ref = got;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                for (int r = 0; r < 5; ++r) run();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("n=%5d T=%4d P=%2d %s : %7.1f ns/sample  %s\n", n, T, P, spec ? "speculative" : "plain      ",
                       ms * 1e6f / 5 / (m - 1), got == ref ? "same" : "DIFF");
            }
        }
        CK(hipFree(d_xyz)); CK(hipFree(d_out));
    }
    return 0;
}
