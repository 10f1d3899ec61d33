// Micro-benchmark (GPU box): how fast can the waves of ONE workgroup per CU move L2-resident data into LDS with LDS-DMA?
//   * NL = 1, 2, 4, 8 issuing waves per workgroup (one workgroup per CU, 256 workgroups)
//   * global_load_lds_dwordx4 (64-bit per-lane addresses) vs buffer_load_dwordx4 ... lds (SGPR resource + 32-bit offsets)
//   * rows of 128 B (the f32 A pieces: 8 rows per instruction) vs rows of 64 B (the 16-bit weight planes: 16 rows per instruction)
// Each wave issues PIECES 1-KiB pieces per "chunk", waits for them (vmcnt(0)), meets the others at s_barrier - the loader's life
// in gemm_x3h_ldr_kernel without the compute waves.  Output: bytes per clock and CU, ns per 32 KiB chunk.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsdma_issue.hip -o variants/ubench/ldsdma_issue && variants/ubench/ldsdma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int NL, int ROWB, bool BUF, int DEPTH>
__global__ __launch_bounds__(NL * 64) void k(const char* src, long long rows, int chunks, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int PIECES = 32 / NL;                       // 1-KiB pieces per wave and chunk (32 KiB per chunk)
    constexpr int LPR = ROWB / 16, RPP = 64 / LPR;        // lanes per row, rows per piece
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a tile's rows: piece p of this wave covers rows (wave * PIECES + p) * RPP .. + RPP - 1 of the workgroup's row block; row stride 4 KiB
    const long long row0 = ((long long)blockIdx.x * 7 % (rows / 512)) * 512;
    long long base[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) base[p] = (row0 + (wave * PIECES + p) * RPP + lane / LPR) * 4096 + (lane % LPR) * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
    const long long t0 = __builtin_amdgcn_s_memtime();
    auto issue = [&](int c, int st) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            char* dst = lds + st * 32768 + (wave * PIECES + p) * 1024;
            const long long off = base[p] + (long long)c * ROWB;
            if (BUF)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)off, 0, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
    int st = 0;
    for (int c = 0; c < chunks; ++c) {
        if (DEPTH == 3 && c + 1 < chunks) __builtin_amdgcn_s_waitcnt((PIECES & 0xF) | ((PIECES >> 4) << 14) | (7 << 4) | (15 << 8));
        else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        __builtin_amdgcn_s_barrier();
        if (c + DEPTH - 1 < chunks) issue(c + DEPTH - 1, st == 0 ? DEPTH - 1 : st - 1);
        st = st + 1 == DEPTH ? 0 : st + 1;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (chunks < 0) sink[threadIdx.x] = lds[threadIdx.x];
}

template <int NL, int ROWB, bool BUF, int DEPTH> void run(const char* src, long long rows, float* sink, long long* cyc) {
    const int chunks = 128 / (ROWB / 16) * 2;             // stay inside the 4-KiB row: ROWB * chunks <= 4096
    auto fn = k<NL, ROWB, BUF, DEPTH>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(256), dim3(NL * 64), DEPTH * 32768, 0, src, rows, chunks, cyc, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s rows of %3d B, %d issuing wave(s), ring %d: %7.1f cycles per 32-KiB chunk = %5.1f B/clk/CU; launch %.1f us -> %.2f TB/s over 256 CUs\n",
           BUF ? "buffer_load lds" : "global_load_lds", ROWB, NL, DEPTH, (double)c / chunks, 32768.0 * chunks / (double)c, ms * 1e3,
           256.0 * 32768.0 * chunks / (ms * 1e-3) / 1e12);
}
int main() {
    const long long rows = 8192;                          // 8192 rows x 4 KiB = 32 MiB: L2 / MALL resident after the warm-up
    char* src; float* sink; long long* cyc;
    (void)hipMalloc(&src, rows * 4096); (void)hipMemset(src, 1, rows * 4096); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 8);
    run<1, 128, false, 3>(src, rows, sink, cyc); run<2, 128, false, 3>(src, rows, sink, cyc); run<4, 128, false, 3>(src, rows, sink, cyc);
    run<8, 128, false, 3>(src, rows, sink, cyc); run<16, 128, false, 3>(src, rows, sink, cyc);
    run<4, 64, false, 3>(src, rows, sink, cyc); run<8, 64, false, 3>(src, rows, sink, cyc);
    run<4, 128, true, 3>(src, rows, sink, cyc); run<8, 128, true, 3>(src, rows, sink, cyc); run<4, 64, true, 3>(src, rows, sink, cyc);
    run<4, 128, false, 2>(src, rows, sink, cyc); run<4, 256, false, 3>(src, rows, sink, cyc); run<4, 1024, false, 3>(src, rows, sink, cyc);
    return 0;
}
