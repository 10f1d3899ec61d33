// Micro-benchmark (GPU box): what one SIMD of gfx950 sustains when v_mfma_f32_32x32x16_bf16 shares its waves' instruction
// streams with the fillers of the x6 GEMM's inner loop (split arithmetic: v_perm / v_and / v_sub, fragment fetch:
// ds_read_b128), for 1, 2 and 3 waves per SIMD.  Prints shader cycles per MFMA per SIMD (32 = the matrix pipe's pace).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/x6_issue.hip -o variants/ubench/x6_issue && variants/ubench/x6_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the split of the x6 GEMM (gemm_f32.hip, split3_bf16): 44 VALU per call
__device__ __forceinline__ void split3(const f32x4& lo, const f32x4& hi, u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = i < 2 ? lo[2 * i] : hi[2 * i - 4], y = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
        const unsigned xb = __float_as_uint(x), yb = __float_as_uint(y);
        p1[i] = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
        const float xr = x - __uint_as_float(xb & 0xffff0000u), yr = y - __uint_as_float(yb & 0xffff0000u);
        const unsigned xc = __float_as_uint(xr), yc = __float_as_uint(yr);
        p2[i] = __builtin_amdgcn_perm(yc, xc, 0x07060302u);
        const float xs = xr - __uint_as_float(xc & 0xffff0000u), ys = yr - __uint_as_float(yc & 0xffff0000u);
        p3[i] = __builtin_amdgcn_perm(__float_as_uint(ys), __float_as_uint(xs), 0x07060302u);
    }
}

// per 12 MFMAs (two accumulators, as one wave of the 128x128 tile has): NS splits (44 VALU each, interleaved with the
// MFMAs by the same sched_group_barrier pattern the GEMM uses) and NLDS ds_read_b128 (waited for once per 12 MFMAs)
template <int NS, int NLDS, int PH = 0, int BAR = 0>     // BAR: s_barrier every BAR iterations (0: the waves run free)
__global__ void k(float* out, int iters, long long* cyc, long long* real) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 1e-3f + e); b[e] = (__bf16)(1.0f + e * 0.25f); }
    f32x4 lo[3], hi[3];
    for (int q = 0; q < 3; ++q)
        for (int e = 0; e < 4; ++e) { lo[q][e] = threadIdx.x * 0.37f + e + q; hi[q][e] = threadIdx.x * 0.11f - e - q; }
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16 +
                          (threadIdx.x >> 6) * 1024;
    f32x4 r[8];
    for (int e = 0; e < 8; ++e) r[e] = f32x4{0, 0, 0, 0};
    constexpr int VPM = (44 * NS + 11) / 12;
    const long long t0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < NLDS; ++l) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[l & 7]) : "v"(base), "i"(l * 4096 % 32768));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NS; ++q) {     // loop-carried: the planes of this round are the inputs of the next
            u32x4 p1, p2, p3;
            split3(lo[q], hi[q], p1, p2, p3);
            lo[q] = __builtin_bit_cast(f32x4, p1 ^ p3);
            hi[q] = __builtin_bit_cast(f32x4, p2);
        }
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
        if (PH == 0) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (VPM) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
        } else {        // serial phases, as the non-MP GEMM loop: all the VALU first, then the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x002, 44 * NS + 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (BAR && (it % BAR) == BAR - 1) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    for (int q = 0; q < 3; ++q) s += lo[q][0] + hi[q][1];
    for (int e = 0; e < 8; ++e) s += r[e][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // the LAST wave to finish counts (issue arbitration favours the oldest wave: it finishes early, the others late)
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
        atomicMax((unsigned long long*)real, (unsigned long long)(w1 - w0));
        atomicMin((unsigned long long*)(cyc + 1), (unsigned long long)(t1 - t0));
    }
}

// LDS throughput as the x6 tile uses it: NC "compute" waves each issue 8 ds_read_b128 (1 KiB per wave-instruction) and wait,
// while NL "loader" waves stream 1-KiB LDS-DMA pieces (global_load_lds_dwordx4) from an L2-resident buffer into the same LDS.
template <int NC, int NL, int PIECES>
__global__ __launch_bounds__((NC + NL) * 64) void lds_bw(const float* src, float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[32768];      // 128 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (wave < NC) {
        const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + lane * 16 + wave * 8192;
        f32x4 r[8];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 8; ++l) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[l]) : "v"(base), "i"(l * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) :: "memory");
            s += r[0][0] + r[7][3];
        }
    } else {
        const float* g = src + (size_t)blockIdx.x * 65536 + (wave - NC) * 16384 + lane * 4;
        float* dst = lds + 16384 + (wave - NC) * 256 * PIECES;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + pc * 256 + (it & 7) * 2048),
                                                 (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));        // vmcnt(0)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == NC * 64 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
// LDS-DMA ingest of one CU with DEPTH rounds in flight per loader wave (counted vmcnt, as the GEMM's loader waves do):
// NL waves x PIECES KiB per round from a per-workgroup region of FOOT KiB (L2-resident when small)
template <int NL, int PIECES, int DEPTH, int FOOT>
__global__ __launch_bounds__(NL * 64) void dma_bw(const float* src, float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[NL * PIECES * 256 * DEPTH];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float* g = src + (size_t)blockIdx.x * (FOOT * 256) + lane * 4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float* dst = lds + ((it % DEPTH) * NL + wave) * PIECES * 256;
        const int off = ((it * NL + wave) * PIECES * 256) % (FOOT * 256);
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (off + pc * 256) % (FOOT * 256)),
                                             (__attribute__((address_space(3))) void*)(dst + pc * 256), 16, 0, 0);
        constexpr int LEFT = (DEPTH - 1) * PIECES;
        __builtin_amdgcn_s_waitcnt((LEFT & 0xF) | ((LEFT >> 4) << 14) | (7 << 4) | (15 << 8));
    }
    __builtin_amdgcn_s_waitcnt(0x0070 | (15 << 8));
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NL, int PIECES, int DEPTH, int FOOT> void run_dma(int iters, int blocks) {
    float *src, *out; long long* cyc;
    hipMalloc(&src, (size_t)256 * FOOT * 1024 + 1048576); hipMemset(src, 0, (size_t)256 * FOOT * 1024 + 1048576);
    hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dma_bw<NL, PIECES, DEPTH, FOOT>), dim3(blocks), dim3(NL * 64), 0, 0, src, out, iters, cyc);
    hipDeviceSynchronize();
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("LDS-DMA ingest: %d workgroups, %d loader waves x %d KiB per round, %d rounds in flight, %d KiB per workgroup region: "
           "%.1f B/clk/CU (%.0f cycles per round)\n", blocks, NL, PIECES, DEPTH, FOOT, (double)NL * PIECES * 1024.0 * iters / c, (double)c / iters);
    hipFree(src); hipFree(out); hipFree(cyc);
}

template <int NC, int NL, int PIECES> void run_lds(int iters) {
    float *src, *out; long long* cyc;
    hipMalloc(&src, 256ull * 65536 * 4 + 1048576); hipMemset(src, 0, 256ull * 65536 * 4 + 1048576);
    hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((lds_bw<NC, NL, PIECES>), dim3(256), dim3((NC + NL) * 64), 0, 0, src, out, iters, cyc);
    hipDeviceSynchronize();
    long long c[2] = {0, 0};
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("LDS: %d reader waves x 8 ds_read_b128 / wait, %d loader waves x %d KiB LDS-DMA / wait: readers %.0f B/clk/CU", NC, NL, PIECES,
           NC ? (double)NC * 8192.0 * iters / c[0] : 0.0);
    if (NL) printf(", loaders %.1f B/clk/CU (%.0f cycles per round)", (double)NL * PIECES * 1024.0 * iters / c[1], (double)c[1] / iters);
    printf("\n");
    hipFree(src); hipFree(out); hipFree(cyc);
}

template <class F> void run(const char* name, F kern, int waves_per_simd, int iters) {
    const int blocks = 256, threads = 256 * waves_per_simd;
    float* out; long long *cyc, *real;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, 16); hipMalloc(&real, 8);
    long long c2[2] = {0, 0}, r = 0;
    for (int w = 0; w < 2; ++w) {
        const long long init[2] = {0, 0x7fffffffffffffffll};
        hipMemcpy(cyc, init, 16, hipMemcpyHostToDevice); hipMemset(real, 0, 8);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc, real);
        hipDeviceSynchronize();
    }
    hipMemcpy(c2, cyc, 16, hipMemcpyDeviceToHost); hipMemcpy(&r, real, 8, hipMemcpyDeviceToHost);
    const long long c = c2[0];
    const double mf = 12.0 * iters * waves_per_simd;        // MFMAs issued on one SIMD
    printf("%-22s waves/SIMD %d: %7.1f cycles per MFMA per SIMD (pipe pace 32; first wave done at %.0f %% of the last), clock %.2f GHz\n",
           name, waves_per_simd, (double)c / mf, 100.0 * c2[1] / c, (double)c / ((double)r / 100e6) / 1e9);
    hipFree(out); hipFree(cyc); hipFree(real);
}

int main() {
    const int iters = 4000;
    for (int w = 1; w <= 3; ++w) {
        run("0 splits (0 VALU)", k<0, 0>, w, iters);
        run("1 split / 12 (3.7 V/M)", k<1, 0>, w, iters);
        run("2 splits / 12 (7.3)", k<2, 0>, w, iters);
        run("3 splits / 12 (11)", k<3, 0>, w, iters);
        run("0 splits + 8 LDS", k<0, 8>, w, iters);
        run("1 split + 8 LDS", k<1, 8>, w, iters);
        run("2 splits + 8 LDS", k<2, 8>, w, iters);
        run("1sp+8LDS, barrier/24", k<1, 8, 0, 2>, w, iters);
        run("1sp+8LDS ser, bar/24", k<1, 8, 1, 2>, w, iters);
        run("1sp+8LDS, barrier/12", k<1, 8, 0, 1>, w, iters);
        run("1sp+8LDS ser, bar/12", k<1, 8, 1, 1>, w, iters);
        run("1sp+8LDS, barrier/96", k<1, 8, 0, 8>, w, iters);
        run("1 split, serial phases", k<1, 0, 1>, w, iters);
        run("1 split+8 LDS, serial", k<1, 8, 1>, w, iters);
        run("2 splits+8 LDS, serial", k<2, 8, 1>, w, iters);
    }
    run_dma<4, 10, 1, 64>(4000, 256);  run_dma<4, 10, 2, 64>(4000, 256);  run_dma<4, 10, 3, 64>(4000, 256);
    run_dma<4, 16, 2, 64>(4000, 256);  run_dma<4, 8, 4, 64>(4000, 256);   run_dma<8, 8, 2, 64>(4000, 256);
    run_dma<4, 10, 2, 4096>(4000, 256); run_dma<4, 10, 3, 4096>(4000, 256); run_dma<4, 16, 2, 4096>(4000, 256);
    run_dma<4, 10, 3, 64>(4000, 96);   run_dma<4, 10, 3, 4096>(4000, 96);
    run_lds<8, 0, 1>(4000);
    run_lds<4, 0, 1>(4000);
    run_lds<8, 4, 10>(4000);
    run_lds<0, 4, 10>(4000);
    run_lds<0, 4, 4>(4000);
    run_lds<0, 8, 5>(4000);
    run_lds<8, 4, 4>(4000);
    return 0;
}
