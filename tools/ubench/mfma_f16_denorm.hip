// Micro-benchmark (GPU box): does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs, does v_cvt_pk_f16_f32 produce them, and
// at what rate does the fp16 MFMA issue compared with the bf16 one?  (gemm_x3h.hip relies on the first two for its accuracy at
// the small end: |a| < 2^-14.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f16_denorm.hip -o variants/ubench/mfma_f16_denorm && variants/ubench/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// every lane: A[row = lane & 31][k = 8 (lane >> 5) + i] = a, B likewise = b -> C[r][c] = 16 a b
__global__ void probe(float a, float b, float* out, unsigned* bits) {
    const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = h[0]; B[i] = h[1]; }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; bits[0] = __builtin_bit_cast(unsigned, h); }
}
template <bool F16>
__global__ void rate(float* out, int iters, long long* cyc) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    f16x8 x; bf16x8 y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 1e-3f); y[i] = (__bf16)(threadIdx.x * 1e-3f); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (F16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, acc[a], 0, 0, 0);
            }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; unsigned* bits; long long* cyc;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&bits, 16); hipMalloc(&cyc, 8);
    struct { float a, b; const char* what; } cases[] = {
        {1.0f, 1.0f, "normal x normal"},
        {ldexpf(1.f, -20), 1.0f, "subnormal (2^-20) x 1"},
        {ldexpf(1.f, -24), 1024.0f, "smallest subnormal (2^-24) x 1024"},
        {ldexpf(1.5f, -16), ldexpf(1.25f, -15), "subnormal x subnormal (1.5 2^-16 x 1.25 2^-15)"},
        {ldexpf(1.f, -25) * 1.01f, 1.0f, "just above half the smallest subnormal x 1 (rounds UP to 2^-24)"},
        {70000.0f, 1.0f, "70000 (beyond fp16) x 1"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c.a, c.b, out, bits);
        float r; unsigned hb;
        hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bits, 4, hipMemcpyDeviceToHost);
        printf("%-66s cvt_pk bits a=0x%04x b=0x%04x  mfma C = %.9g  (16 a b in f32 = %.9g)\n", c.what, hb & 0xffff, hb >> 16, r,
               16.0 * (double)c.a * (double)c.b);
    }
    for (int f16 = 0; f16 < 2; ++f16)
        for (int wpc : {1, 2}) {
            const int blocks = 256, threads = 256 * wpc, iters = 4096;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0);
                if (f16) hipLaunchKernelGGL(rate<true>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
                else hipLaunchKernelGGL(rate<false>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double mf = 16.0 * iters, waves = (double)blocks * threads / 64;
            printf("%s 32x32x16, %d wave(s) per SIMD on 256 CUs: %.1f cycles per MFMA and wave, %.0f TF/s\n", f16 ? "f16 " : "bf16", wpc,
                   (double)c / mf, waves * mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
