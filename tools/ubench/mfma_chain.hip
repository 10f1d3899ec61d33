// Micro-benchmark (GPU box): f32 MFMA issue vs dependent-accumulator latency, and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, long long* cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ void k16(float* out, int iters, long long* cyc) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 4; ++e) acc[a][e] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 4; ++e) s += acc[a][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <class F> void run(const char* name, F kern, int blocks, int threads, int iters, double flop_per_mfma) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double mf = 16.0 * iters;
    double waves = (double)blocks * threads / 64;
    printf("%-22s blocks %4d thr %4d: %8.1f us  memtime %10lld ticks (%.1f ticks/mfma, tick rate %.0f MHz)  %.1f TF/s\n",
           name, blocks, threads, ms * 1e3, c, c / mf, c / (ms * 1e3), waves * mf * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    const double f32 = 2.0 * 32 * 32 * 2, f16 = 2.0 * 16 * 16 * 4;
    for (int blocks : {12, 256}) {
        run("32x32x2 1 acc", k32<1>, blocks, 256, it, f32);
        run("32x32x2 2 acc", k32<2>, blocks, 256, it, f32);
        run("32x32x2 4 acc", k32<4>, blocks, 256, it, f32);
        run("16x16x4 1 acc", k16<1>, blocks, 256, it, f16);
        run("16x16x4 2 acc", k16<2>, blocks, 256, it, f16);
        run("16x16x4 4 acc", k16<4>, blocks, 256, it, f16);
    }
    run("32x32x2 1 acc 2w/SIMD", k32<1>, 256, 512, it, f32);
    run("32x32x2 4 acc 2w/SIMD", k32<4>, 256, 512, it, f32);
    return 0;
}
