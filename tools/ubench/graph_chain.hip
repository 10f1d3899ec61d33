// Micro-benchmark (GPU box): what a DEPENDENT KERNEL BOUNDARY costs when the chain is replayed from a hipGraph instead of
// being enqueued launch by launch on a stream - the one launch-gap mechanism never measured (VERDICT r4 missing 3 / next 2;
// SURVEY 7 step 6 "whole loop under hipGraph").  Same method as grid_sync.hip (a)/(b): N dependent launches of a trivial
// 256 x 256-thread kernel, eight kernels in rotation; here three ways -
//   (a) launch by launch on one stream (the product's AR step chains today);
//   (b) the same N launches captured once (hipStreamBeginCapture) and replayed with hipGraphLaunch;
//   (c) a graph built node by node (hipGraphAddKernelNode, a linear dependency chain).
// and with kernels that WORK for 2.5 ... 43 us (a dependent FMA chain), where every kernel also stamps s_memrealtime (100 MHz,
// constant rate) at entry and exit: gap[i] = entry[i + 1] - exit[i] is the boundary itself, free of event overheads.
// The one-utterance AR step is ~86 dependent launches: graphs of 86 nodes are measured too (replayed back to back).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/graph_chain.hip -o variants/ubench/graph_chain && variants/ubench/graph_chain
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ID> __global__ void tiny(float* p, int n, unsigned long long* stamp, int slot, int iters) {
    unsigned long long t0 = 0;
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int k = 0; k < iters; ++k) v = __builtin_fmaf(v, 1.0001f, (float)ID);      // dependent chain: ~4 cycles per iteration
    if (i < n) p[i] = v;
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) {
        stamp[2 * slot] = t0;
        stamp[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

static void launch_i(int i, float* buf, int n, unsigned long long* stamp, int iters, hipStream_t s) {
    const dim3 g(256), b(256);
    switch (i & 7) {
        case 0: hipLaunchKernelGGL(tiny<0>, g, b, 0, s, buf, n, stamp, i, iters); break;
        case 1: hipLaunchKernelGGL(tiny<1>, g, b, 0, s, buf, n, stamp, i, iters); break;
        case 2: hipLaunchKernelGGL(tiny<2>, g, b, 0, s, buf, n, stamp, i, iters); break;
        case 3: hipLaunchKernelGGL(tiny<3>, g, b, 0, s, buf, n, stamp, i, iters); break;
        case 4: hipLaunchKernelGGL(tiny<4>, g, b, 0, s, buf, n, stamp, i, iters); break;
        case 5: hipLaunchKernelGGL(tiny<5>, g, b, 0, s, buf, n, stamp, i, iters); break;
        case 6: hipLaunchKernelGGL(tiny<6>, g, b, 0, s, buf, n, stamp, i, iters); break;
        default: hipLaunchKernelGGL(tiny<7>, g, b, 0, s, buf, n, stamp, i, iters); break;
    }
}
static void* fn_i(int i) {
    switch (i & 7) {
        case 0: return (void*)tiny<0>; case 1: return (void*)tiny<1>; case 2: return (void*)tiny<2>; case 3: return (void*)tiny<3>;
        case 4: return (void*)tiny<4>; case 5: return (void*)tiny<5>; case 6: return (void*)tiny<6>; default: return (void*)tiny<7>;
    }
}

int main() {
    const int NWG = 256, NEL = NWG * 256;
    float* buf;
    unsigned long long* stamp;
    CK(hipMalloc(&buf, sizeof(float) * NEL));
    CK(hipMemset(buf, 0, sizeof(float) * NEL));
    CK(hipMalloc(&stamp, sizeof(unsigned long long) * 2 * 512));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f;
    std::vector<unsigned long long> h(2 * 512);
    auto gaps = [&](int N, double& med_gap_us, double& med_kernel_us) {
        if (hipMemcpy(h.data(), stamp, sizeof(unsigned long long) * 2 * N, hipMemcpyDeviceToHost) != hipSuccess) return;
        std::vector<double> g, k;
        for (int i = 0; i + 1 < N; ++i) g.push_back((double)(long long)(h[2 * (i + 1)] - h[2 * i + 1]) * 0.01);
        for (int i = 0; i < N; ++i) k.push_back((double)(long long)(h[2 * i + 1] - h[2 * i]) * 0.01);
        std::sort(g.begin(), g.end());
        std::sort(k.begin(), k.end());
        med_gap_us = g[g.size() / 2];
        med_kernel_us = k[k.size() / 2];
    };
    for (int N : {400, 86}) {
        for (int iters : {0, 150, 300, 600, 2600}) {   // 0: the trivial kernel of grid_sync.hip; a dependent FMA costs ~16.5 ns here: 2.5 / 5 / 10 / 43 us
            // ---- (a) stream
            for (int i = 0; i < 16; ++i) launch_i(i, buf, NEL, stamp, iters, s);
            CK(hipStreamSynchronize(s));
            double best_a = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < N; ++i) launch_i(i, buf, NEL, stamp, iters, s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_a = std::min(best_a, (double)ms * 1e3 / N);
            }
            double ga = 0, ka = 0;
            gaps(N, ga, ka);
            // ---- (b) captured graph
            hipGraph_t graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < N; ++i) launch_i(i, buf, NEL, stamp, iters, s);
            CK(hipStreamEndCapture(s, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(exec, s));
            CK(hipStreamSynchronize(s));
            double best_b = 1e9, best_b2 = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(exec, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_b = std::min(best_b, (double)ms * 1e3 / N);
            }
            double gb = 0, kb = 0;
            gaps(N, gb, kb);
            for (int rep = 0; rep < 3; ++rep) {            // four replays back to back: the graph-to-graph seam is inside
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(exec, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_b2 = std::min(best_b2, (double)ms * 1e3 / (4 * N));
            }
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
            // ---- (c) explicit nodes
            hipGraph_t g2;
            CK(hipGraphCreate(&g2, 0));
            std::vector<hipGraphNode_t> nodes(N);
            int n_el = NEL;
            std::vector<int> slots(N);
            for (int i = 0; i < N; ++i) {
                slots[i] = i;
                void* args[5] = {&buf, &n_el, &stamp, &slots[i], &iters};
                hipKernelNodeParams kp{};
                kp.func = fn_i(i);
                kp.gridDim = dim3(256);
                kp.blockDim = dim3(256);
                kp.sharedMemBytes = 0;
                kp.kernelParams = args;
                kp.extra = nullptr;
                CK(hipGraphAddKernelNode(&nodes[i], g2, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &kp));
            }
            hipGraphExec_t ex2;
            CK(hipGraphInstantiate(&ex2, g2, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ex2, s));
            CK(hipStreamSynchronize(s));
            double best_c = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(ex2, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_c = std::min(best_c, (double)ms * 1e3 / N);
            }
            double gc = 0, kc = 0;
            gaps(N, gc, kc);
            CK(hipGraphExecDestroy(ex2));
            CK(hipGraphDestroy(g2));
            std::printf("N = %3d dependent launches, kernel body %s (median in-kernel time %.2f us):\n", N,
                        iters ? "a chain of dependent FMAs" : "trivial", ka);
            std::printf("  (a) stream, launch by launch : %.2f us per launch   median gap exit -> next entry %.2f us\n", best_a, ga);
            std::printf("  (b) captured graph, replayed : %.2f us per launch   median gap %.2f us   (4 replays back to back: %.2f us per launch)\n",
                        best_b, gb, best_b2);
            std::printf("  (c) explicit kernel nodes    : %.2f us per launch   median gap %.2f us\n", best_c, gc);
        }
    }
    return 0;
}
