// fps_pruned_lab.hip -- A/B of the pruned FPS tier (csrc/fps_pruned_body.h) against the full tiers, product code included
// verbatim. Prints ns per round, the kd build's cost (a launch with m = 1 runs the build and no round) and whether the
// indices agree. Development aid (scripts/build_labs.sh builds it into build_lab/fps_pruned_lab).
#include "../pointnet2_amd/csrc/fps.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static float timed(int reps, const std::function<void()> &fn)
{
    fn();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;   // us
}

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int b = 32;
    // LAB_KINDS = 5: also the clouds whose batches are slow (2 = drawn with replacement from n / 4 sphere points, 3 = cube quantised
    // to 1 / 64, 4 = 6 x 6 x 6 lattice)
    const int kinds = getenv("LAB_KINDS") ? atoi(getenv("LAB_KINDS")) : 2;
    static const char *kind_name[] = {"cube  ", "sphere", "dupl  ", "q 1/64", "lattic"};
    for (int n : {4096, 8192, 1024, 2048}) {
        for (int kind = 0; kind < kinds; ++kind) {
            const int m = getenv("LAB_M") ? atoi(getenv("LAB_M")) : n == 3000 ? 750 : 1024;
            std::vector<float> h((size_t)b * n * 3);
            uint32_t s = 12345u + kind;
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
            for (size_t i = 0; i < h.size(); i += 3) {
                float x = rnd(), y = rnd(), z = rnd();
                if (kind == 1 || kind == 2) {   // sphere surface, radius 0.9 .. 1.0
                    x = 2 * x - 1; y = 2 * y - 1; z = 2 * z - 1;
                    const float r = sqrtf(x * x + y * y + z * z) + 1e-9f, q = (0.9f + 0.1f * rnd()) / r;
                    x *= q; y *= q; z *= q;
                }
                if (kind == 3) { x = roundf(x * 64.f) / 64.f; y = roundf(y * 64.f) / 64.f; z = roundf(z * 64.f) / 64.f; }
                if (kind == 4) { x = floorf(x * 6.f) * 0.125f; y = floorf(y * 6.f) * 0.125f; z = floorf(z * 6.f) * 0.125f; }
                h[i] = x; h[i + 1] = y; h[i + 2] = z;
            }
            if (kind == 2)
                for (int c = 0; c < b; ++c)
                    for (int i = n / 4; i < n; ++i) {
                        const int src = (int)(rnd() * (n / 4)) % (n / 4);
                        for (int a = 0; a < 3; ++a) h[((size_t)c * n + i) * 3 + a] = h[((size_t)c * n + src) * 3 + a];
                    }
            float *d_xyz; int *d_out;
            CK(hipMalloc(&d_xyz, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)b * m * 4));
            CK(hipMemcpy(d_xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            std::vector<int> ref((size_t)b * m), got((size_t)b * m);
            const int P512 = n <= 1024 ? 2 : n <= 2048 ? 4 : n <= 4096 ? 8 : 16, P256 = 2 * P512;
            struct V { const char *name; std::function<int(int)> run; };
            std::vector<V> vs = {
                {"full 512", [&](int mm) { return pn2_farthest_point_sample_ex(512, P512, b, n, mm, d_xyz, d_out, nullptr); }},
                {"full 256", [&](int mm) { return pn2_farthest_point_sample_ex(256, P256, b, n, mm, d_xyz, d_out, nullptr); }},
                {"pruned", [&](int mm) { return pn2_farthest_point_sample_variant(PN2_FPS_PRUNED, b, n, mm, d_xyz, nullptr, d_out, nullptr, nullptr); }},
                {"batch", [&](int mm) { return pn2_farthest_point_sample_variant(PN2_FPS_BATCH, b, n, mm, d_xyz, nullptr, d_out, nullptr, nullptr); }},
            };
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                CK(hipMemset(d_out, 0xff, (size_t)b * m * 4));
                if (int rc = vs[vi].run(m)) { printf("%-10s n=%5d %s : launch refused (%d)\n", vs[vi].name, n, kind_name[kind], rc); continue; }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
                if (vi == 0) ref = got;
                const float us = timed(5, [&]() { vs[vi].run(m); });
                const float us1 = timed(5, [&]() { vs[vi].run(1); });
                const float us_half = timed(5, [&]() { vs[vi].run(m / 2); });
#ifdef PN2_BT_STATS
                if (vi == 3) {
                    unsigned long long z[16] = {0}, st[16];
                    CK(hipMemcpyToSymbol(HIP_SYMBOL(pn2::g_bt_stats), z, sizeof(z)));
                    vs[vi].run(m); CK(hipDeviceSynchronize());
                    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(pn2::g_bt_stats), sizeof(st)));
                    printf("   cloud 0: batches %llu, samples %llu (%.2f per batch), exact fallbacks %llu, bisection steps %llu, list size %.1f, (group, sample) updates %llu, ties %llu, speculation misses %llu\n"
                           "   one-per-exchange runs %llu, samples in them %llu\n"
                           "   cycles per batch: picker read %.0f, pick %.0f (%.0f per sample), picker at the barrier %.0f | updater 0: collect %.0f, barrier -> end flag %.0f\n",
                           st[0], st[1], (double)st[1] / st[0], st[2], st[3], (double)st[4] / st[0], st[5], st[11], st[12], st[14], st[13], (double)st[6] / st[0], (double)st[7] / st[0],
                           (double)st[7] / st[1], (double)st[8] / st[0], (double)st[9] / st[0], (double)st[10] / st[0]);
                }
#endif
                printf("%-10s n=%5d %s m=%4d : %7.1f us, prologue (m=1) %6.1f us, %6.1f ns/round overall, %6.1f ns/round in the second half  %s\n",
                       vs[vi].name, n, kind_name[kind], m, us, us1, (us - us1) * 1e3f / (m - 1), (us - us_half) * 1e3f / (m - m / 2),
                       got == ref ? "same" : "DIFF");
            }
            CK(hipFree(d_xyz)); CK(hipFree(d_out));
        }
    }
    return 0;
}
