// fps_prod_lab.hip -- times the PRODUCT FPS kernel (fps.hip included verbatim) under compile-time
// tunables (-DPN2_FPS_*), so micro-variants can be A/B'd in one gpurun call. Development aid.
#include "../pointnet2_amd/csrc/fps.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const char *tag = argc > 1 ? argv[1] : "";
    const int b = 32;
    for (int n : {1024, 2048, 4096, 8192, 16384}) {
        const int m = n / 4;
        std::vector<float> h((size_t)b * n * 3);
        uint32_t s = 12345u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
        float *d_xyz; int *d_out;
        CK(hipMalloc(&d_xyz, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)b * m * 4));
        CK(hipMemcpy(d_xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<int> ref;
        for (int T : {256, 512, 1024}) {
            const int P = n / T;
            if (P < 1 || P > 32 || (T == 1024 && P > 16)) continue;
            if (pn2_farthest_point_sample_ex(T, P, b, n, m, d_xyz, d_out, nullptr)) { printf("launch failed T=%d P=%d\n", T, P); continue; }
            CK(hipDeviceSynchronize());
            std::vector<int> got((size_t)b * m);
            CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
            if (ref.empty()) ref = got;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) pn2_farthest_point_sample_ex(T, P, b, n, m, d_xyz, d_out, nullptr);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-16s n=%5d T=%4d P=%2d : %7.1f ns/round %s\n", tag, n, T, P, ms * 1e6f / 5 / (m - 1), got == ref ? "same" : "DIFF");
        }
        CK(hipFree(d_xyz)); CK(hipFree(d_out));
    }
    return 0;
}
