// memset_node_repro.hip -- minimal reproducer for the defect behind round 5's "stale granules" (profiles/r06/stale_granules.md):
// does a hipMemsetAsync captured into a HIP graph still clear its buffer when the graph is replayed after OTHER work has been
// launched? The graph is [memset(buf, 0, bytes); copy buf -> out]; between replays the buffer is poisoned and eager "noise"
// kernels with recognisable arguments run on the same and on other streams (the serving loop's pattern: eager elementwise
// kernels between graph replays). After every replay `out` must be all zero.
//   hipcc --offload-arch=gfx950 -O2 scripts/memset_node_repro.hip -o build_lab/memset_node_repro && build_lab/memset_node_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void poison(unsigned long long *p, size_t words, unsigned long long v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void copy_words(unsigned long long *dst, const unsigned long long *src, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// noise: a kernel whose arguments are easy to recognise if they show up as a fill pattern
__global__ void noise(unsigned long long *sink, unsigned long long a, unsigned long long b, unsigned long long c, int n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && n < 0) sink[0] = a + b + c;
}

struct Big { unsigned long long v[32]; };
__global__ void noise_big(unsigned long long *sink, Big big, int n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && n < 0) sink[0] = big.v[3];
}

int main(int argc, char **argv)
{
    const int replays = argc > 1 ? atoi(argv[1]) : 3000;
    const int noise_per_replay = argc > 2 ? atoi(argv[2]) : 8;
    const int ngraphs = argc > 3 ? atoi(argv[3]) : 4;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;     // (bit 3: stream kinds, below; bit 4: capture in thread-local mode) bit 0: instantiate with hipGraphInstantiateFlagAutoFreeOnLaunch (what PyTorch does)
                                                          // bit 1: the noise kernels go to the NULL stream; bit 2: noise with a large argument block
    const size_t bytes = 8 * 16 * 512 + 16, words = bytes / 8;       // part_seg's level-1 workspace
    hipStream_t s, s2;
    if (variant & 8) {                                    // bit 3: non-blocking streams, the graph's one of high priority (what PyTorch creates)
        CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1));
        CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, 0));
    } else {
        CK(hipStreamCreate(&s));
        CK(hipStreamCreateWithPriority(&s2, hipStreamDefault, -1));
    }
    std::vector<unsigned long long *> bufs(ngraphs), outs(ngraphs);
    std::vector<hipGraphExec_t> execs(ngraphs);
    unsigned long long *sink;
    CK(hipMalloc(&sink, 64));
    for (int g = 0; g < ngraphs; ++g) {
        CK(hipMalloc(&bufs[g], bytes));
        CK(hipMalloc(&outs[g], bytes));
        // a few eager launches between captures, like a framework's warm-up
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(noise, dim3(1), dim3(64), 0, s, sink, 0xAAAA0000ull + k, 0xBBBBull, 0xCCCCull, 1);
        hipGraph_t graph;
        CK(hipStreamBeginCapture(s, (variant & 16) ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(bufs[g], 0, bytes, s));
        hipLaunchKernelGGL(copy_words, dim3(16), dim3(256), 0, s, outs[g], bufs[g], words);
        CK(hipStreamEndCapture(s, &graph));
        if (variant & 1) CK(hipGraphInstantiateWithFlags(&execs[g], graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
        else CK(hipGraphInstantiate(&execs[g], graph, nullptr, nullptr, 0));
        CK(hipGraphDestroy(graph));
    }
    std::vector<unsigned long long> host(words);
    long bad_replays = 0;
    int first_bad = -1;
    for (int it = 0; it < replays; ++it) {
        const int g = it % ngraphs;
        hipLaunchKernelGGL(poison, dim3(16), dim3(256), 0, s, bufs[g], words, 0x0000000100000000ull | (unsigned)it);   // "tag 1" granules
        for (int k = 0; k < noise_per_replay; ++k) {
            hipStream_t ns = (variant & 2) ? (hipStream_t)0 : (k & 1) ? s2 : s;
            if (variant & 4) {
                Big big;
                for (int q = 0; q < 32; ++q) big.v[q] = 0x2222000000000000ull + q;
                hipLaunchKernelGGL(noise_big, dim3(1), dim3(64), 0, ns, sink, big, 1);
            } else
                hipLaunchKernelGGL(noise, dim3(1), dim3(64), 0, ns, sink, 0x1111000000000000ull + it, 0x500ull, (unsigned long long)(uintptr_t)sink, 1);
        }
        CK(hipGraphLaunch(execs[g], s));
        if (it % 7 == 0 || it > replays - 50) {
            CK(hipMemcpyAsync(host.data(), outs[g], bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            size_t nz = 0, firstw = 0;
            for (size_t i = 0; i < words; ++i) if (host[i]) { if (!nz) firstw = i; ++nz; }
            if (nz) {
                if (first_bad < 0) {
                    first_bad = it;
                    printf("replay %d (graph %d): %zu of %zu words NOT zero after the memset node; first at word %zu: %#018llx %#018llx %#018llx %#018llx\n",
                           it, g, nz, words, firstw, host[firstw], host[firstw + 1 < words ? firstw + 1 : firstw],
                           host[firstw + 2 < words ? firstw + 2 : firstw], host[firstw + 3 < words ? firstw + 3 : firstw]);
                }
                ++bad_replays;
            }
        }
    }
    CK(hipDeviceSynchronize());
    printf("memset node inside a replayed graph: %ld checked replays left non-zero words behind (first at replay %d) -- %s\n", bad_replays,
           first_bad, bad_replays ? "DEFECT REPRODUCED" : "clean in this run");
    return 0;
}
