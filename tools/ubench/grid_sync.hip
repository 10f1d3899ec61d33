// Micro-benchmark (GPU box): what a PHASE BOUNDARY costs on this chip, three ways - the question behind "one cooperative
// kernel per AR step" (VERDICT r3 item 4):
//   (a) a dependent kernel boundary: a chain of N launches on one stream, all the same trivial kernel;
//   (b) the same with EIGHT different kernels in rotation (instruction cache cold at every launch, as in a real layer);
//   (c) a grid-wide barrier INSIDE one launch: XCD-hierarchical (per-XCC arrival counter -> top counter -> per-XCC generation
//       word; one agent-scope release per XCC leader, one agent-scope acquire per workgroup), 256 workgroups, one per CU,
//       with every workgroup touching a private 4-KiB record between barriers so that the fences have something to publish;
//   (d) the flat form: one monotonic counter all workgroups arrive on.
// Every spin is bounded (a stuck barrier sets a timeout word and the kernel ends): a hung GPU would cost the round a strike.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_sync.hip -o variants/ubench/grid_sync && variants/ubench/grid_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ID> __global__ void tiny(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + (float)ID;
}

struct Sync {
    unsigned* xcc_cnt;    // [8 * 32] one arrival counter per XCC (128-B apart)
    unsigned* top;        // [32]
    unsigned* gen;        // [8 * 32] generation word per XCC
    unsigned* xcc_pop;    // [8 * 32] workgroups resident per XCC (census, filled by the kernel itself)
    unsigned* timeout;
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7; }    // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned want, unsigned* timeout) {
    for (int it = 0; it < (1 << 22); ++it) {
        if ((int)(ld_relaxed(p) - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    atomicExch(timeout, 1u);
    return false;
}

// one barrier episode number `ep` (1, 2, ...): returns false on timeout
__device__ bool barrier_xcd(const Sync& s, unsigned ep, unsigned xcc, unsigned pop, unsigned nxcc) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(&s.xcc_cnt[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == ep * pop - 1) {      // last arriver of this XCC: up one level
            const unsigned t = __hip_atomic_fetch_add(s.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t != ep * nxcc - 1) ok = spin_until(s.top, ep * nxcc, s.timeout);
            __hip_atomic_store(&s.gen[xcc * 32], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ok = spin_until(&s.gen[xcc * 32], ep, s.timeout);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__device__ bool barrier_flat(const Sync& s, unsigned ep, unsigned nwg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(s.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = spin_until(s.top, ep * nwg, s.timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <bool XCD>
__global__ __launch_bounds__(256) void phases(Sync s, float* rec, int nphase, unsigned nxcc_expected) {
    __shared__ unsigned sh[4];
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) {     // census: how many workgroups of this grid sit on my XCC (everyone is resident: grid <= CUs)
        __hip_atomic_fetch_add(&s.xcc_pop[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // phase 0 is a flat barrier: after it the census is complete
    if (!barrier_flat(Sync{s.xcc_cnt, s.top + 16, s.gen, s.xcc_pop, s.timeout}, 1, gridDim.x)) return;
    if (threadIdx.x == 0) {
        sh[0] = ld_relaxed(&s.xcc_pop[xcc * 32]);
        unsigned n = 0;
        for (int x = 0; x < 8; ++x) n += ld_relaxed(&s.xcc_pop[x * 32]) ? 1u : 0u;
        sh[1] = n;
    }
    __syncthreads();
    const unsigned pop = sh[0], nxcc = sh[1];
    float* mine = rec + (size_t)blockIdx.x * 1024;
    float acc = 0.f;
    for (int ph = 1; ph <= nphase; ++ph) {
        // "work": rewrite my 4-KiB record from my right neighbour's (a cross-workgroup dependency through memory)
        const float* other = rec + (size_t)((blockIdx.x + 1) % gridDim.x) * 1024;
        float4 v = reinterpret_cast<const float4*>(other)[threadIdx.x];
        acc += v.x;
        v.x += 1.0f;
        if (!(XCD ? barrier_xcd(s, 2 * ph - 1, xcc, pop, nxcc) : barrier_flat(s, 2 * ph - 1, gridDim.x))) return;
        reinterpret_cast<float4*>(mine)[threadIdx.x] = v;
        if (!(XCD ? barrier_xcd(s, 2 * ph, xcc, pop, nxcc) : barrier_flat(s, 2 * ph, gridDim.x))) return;
    }
    if (threadIdx.x == 0 && acc == 123.456f) rec[0] = acc;
    (void)nxcc_expected;
}

int main() {
    const int NWG = 256, N = 400;
    float* buf;
    CK(hipMalloc(&buf, sizeof(float) * NWG * 1024));
    CK(hipMemset(buf, 0, sizeof(float) * NWG * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f;
    auto launch_same = [&](int i) { (void)i; hipLaunchKernelGGL(tiny<0>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); };
    auto launch_mixed = [&](int i) {
        switch (i & 7) {
            case 0: hipLaunchKernelGGL(tiny<0>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            case 1: hipLaunchKernelGGL(tiny<1>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            case 2: hipLaunchKernelGGL(tiny<2>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            case 3: hipLaunchKernelGGL(tiny<3>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            case 4: hipLaunchKernelGGL(tiny<4>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            case 5: hipLaunchKernelGGL(tiny<5>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            case 6: hipLaunchKernelGGL(tiny<6>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
            default: hipLaunchKernelGGL(tiny<7>, dim3(NWG), dim3(256), 0, 0, buf, NWG * 256); break;
        }
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 16; ++i) which ? launch_mixed(i) : launch_same(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < N; ++i) which ? launch_mixed(i) : launch_same(i);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) std::printf("%s: %d dependent launches of a trivial 256 x 256-thread kernel: %.2f us per launch\n",
                                 which ? "(b) eight kernels in rotation" : "(a) one kernel", N, ms * 1e3 / N);
        }
    }
    unsigned* words;
    CK(hipMalloc(&words, sizeof(unsigned) * (8 * 32 * 3 + 64)));
    for (int xcd = 1; xcd >= 0; --xcd) {
        for (int nphase : {50, 200}) {
            CK(hipMemset(words, 0, sizeof(unsigned) * (8 * 32 * 3 + 64)));
            Sync s{words, words + 8 * 32 * 3, words + 8 * 32, words + 8 * 32 * 2, words + 8 * 32 * 3 + 48};
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (xcd) hipLaunchKernelGGL(phases<true>, dim3(NWG), dim3(256), 0, 0, s, buf, nphase, 8u);
            else hipLaunchKernelGGL(phases<false>, dim3(NWG), dim3(256), 0, 0, s, buf, nphase, 8u);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned> h(8 * 32 * 3 + 64);
            CK(hipMemcpy(h.data(), words, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            std::printf("(%s) %s barrier, 256 workgroups, %d phases x 2 barriers in one launch: %.1f us total -> %.2f us per barrier "
                        "(incl. a 4-KiB record exchanged per workgroup and phase)%s; workgroups per XCC:", xcd ? "c" : "d",
                        xcd ? "XCD-hierarchical" : "flat-counter", nphase, ms * 1e3, ms * 1e3 / (2 * nphase + 1),
                        h[8 * 32 * 3 + 48] ? "  ** TIMEOUT **" : "");
            for (int x = 0; x < 8; ++x) std::printf(" %u", h[8 * 32 * 2 + x * 32]);
            std::printf("\n");
        }
    }
    return 0;
}
