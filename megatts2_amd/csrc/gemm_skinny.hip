// Weight-streaming linear layer for a handful of rows (gfx950): C[g][m, n] = epi( sum_k pro(X[g][src(m), k]) W[g][n, k] )
// with M <= 64 - the regime of the reference's own inference call (one utterance: t+1 rows at AR step t,
// models/megatts2.py:165-181,257-275) and of the "last row only" launches of every batched AR step.
//
// Such a launch moves N*K weights once and does almost no arithmetic: its floor is the weight stream.  The tiled engine
// (gemm_f32.hip) stages operands through an LDS ring with a barrier per 32-wide chunk; at one or two chunks per wave the
// ring never fills and a launch costs 8-16 us whatever its size (profiles/r03_c1_kernel_stats.csv).  Here nothing is
// staged: both MFMA operands of v_mfma_f32_32x32x2_f32 have the SAME register layout (lane = (row & 31) + 32 * k-half),
// so a lane loads 64 contiguous bytes of "its" weight row and of "its" activation row straight from memory - 16 dwords
// that feed 16 MFMAs - with every load of the wave's whole K share in flight before the first MFMA.
//   workgroup = one 32-column block of W and one K slice (blockIdx.z: the GemmP group = split-K slab), its NW waves
//   split that K range; the partial 32x32 accumulators meet in LDS and are added in wave order (deterministic);
//   each wave then finishes 16/NW accumulator elements: bias, activation, residual, row mask - the engine's epilogue.
// Arithmetic: exact f32 products, f32 accumulation in a fixed order (k ascending inside a wave, waves ascending) - a
// DIFFERENT order than the tiled engine's, so a layer evaluated here (M <= 64) and there (M > 64) agrees to f32 round-off,
// not bit for bit (batch-1 and batched results of the same utterance differ in the last bits; every discrete output of
// the path is checked on both, tests/test_gpu_stages.py::test_prod_plm_batched_vs_alone_decisions).
#include "mt2_kernels.h"

namespace mt2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sk_act(int act, float v, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_LRELU: return v >= 0.0f ? v : v * slope;
        case ACT_TANH: return tanhf(v);
        case ACT_LOGCLAMP: return logf(fmaxf(v, slope));
        default: return v;
    }
}

// RT: row tiles of 32; NW: waves (K shares); U: 32-wide K steps whose loads are issued together
template <int RT, int NW, int U, bool NT>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_f32_kernel(GemmP p) {
    extern __shared__ float sk_red[];                       // [NW][RT][16][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.z, n0 = blockIdx.x * 32;
    const int col = lane & 31, h = lane >> 5;
    const int Kw = p.K / NW;                                // multiple of 32 (checked by the launcher)
    const int koff = wave * Kw + h * 16;
    // weight row of this lane (columns past N: clamped, never stored)
    const int nrow = min(n0 + col, p.N - 1);
    const float* __restrict__ wp = p.W + (long long)g * p.strideW + (long long)nrow * p.ldw + koff;
    const float* __restrict__ xp[RT];
    bool xok[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = i * 32 + col;
        const int src = m * p.a_mul + p.shift0;
        xok[i] = m < p.M && src >= 0 && src < p.Rx;
        xp[i] = p.X + (long long)g * p.strideX + (long long)(xok[i] ? src : 0) * p.ldx + koff;
    }
    // prologue activation, branch-free and identical to the tiled engine's apply_act on every input, non-finite ones
    // included: max(v, relu ? 0 : v * ns) with ns = 1 (none: max(v, v) = v) or the leaky slope (0 < slope < 1)
    const float slope = p.pro_slope;
    const bool relu = p.pro_act == ACT_RELU;
    const float ns = p.pro_act == ACT_NONE ? 1.0f : slope;

    f32x16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    const int NS = Kw >> 5;
    for (int sb = 0; sb < NS; sb += U) {
        f32x4 w[U][4], a[U][RT][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (sb + u < NS) {
                const int k = (sb + u) * 32;
                if (NT) {     // weights are streamed once per launch: keep them out of the way of the activations in L2
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[u][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + k + 4 * j));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[u][j] = *reinterpret_cast<const f32x4*>(wp + k + 4 * j);
                }
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][i][j] = *reinterpret_cast<const f32x4*>(xp[i] + k + 4 * j);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (sb + u < NS) {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = a[u][i][j];
                        if (!xok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        v.x = fmaxf(v.x, relu ? 0.0f : v.x * ns); v.y = fmaxf(v.y, relu ? 0.0f : v.y * ns);
                        v.z = fmaxf(v.z, relu ? 0.0f : v.z * ns); v.w = fmaxf(v.w, relu ? 0.0f : v.w * ns);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, w[u][j].x, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, w[u][j].y, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, w[u][j].z, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, w[u][j].w, acc[i], 0, 0, 0);
                    }
            }
        }
    }

    // K shares meet in LDS; wave w finishes accumulator elements [w * NE, (w + 1) * NE) of every row tile
    constexpr int NE = 16 / NW;
    if (NW > 1) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) sk_red[((wave * RT + i) * 16 + e) * 64 + lane] = acc[i][e];
        __syncthreads();
    }
    const float* __restrict__ bias = p.bias ? p.bias + (long long)g * p.strideB : nullptr;
    const float* __restrict__ R = p.R ? p.R + (long long)g * p.strideR : nullptr;
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int n = n0 + col;
    const bool nok = n < p.N;
    const float bv = (bias && nok) ? bias[n] : 0.0f;
    const int epi = p.epi_act;
    const float osc = p.out_scale;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = NW > 1 ? wave * NE + q : q;
            float s;
            if (NW > 1) {
                s = sk_red[((0 * RT + i) * 16 + e) * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) s += sk_red[((w2 * RT + i) * 16 + e) * 64 + lane];
            } else {
                s = acc[i][q];
            }
            const int m = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (nok && m < p.M) {
                float v = sk_act(epi, s + bv, slope) * osc;
                if (R) v += R[(long long)m * p.ldr + n];
                if (p.valid && p.valid[m] == 0) v = 0.0f;
                C[(long long)m * p.ldc + n] = v;
            }
        }
}

template <int RT, int NW, int U>
hipError_t sk_launch(const GemmP& p, hipStream_t s) {
    static bool attr_done = false;      // idempotent: a race only repeats the call
    void (*fn)(GemmP) = p.w_nt ? gemm_skinny_f32_kernel<RT, NW, U, true> : gemm_skinny_f32_kernel<RT, NW, U, false>;
    const size_t lds = NW > 1 ? (size_t)NW * RT * 16 * 64 * sizeof(float) : 0;
    if (!attr_done && lds > 48 * 1024) {
        for (int v = 0; v < 2; ++v) {
            void (*f2)(GemmP) = v ? gemm_skinny_f32_kernel<RT, NW, U, true> : gemm_skinny_f32_kernel<RT, NW, U, false>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(f2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(fn, dim3((p.N + 31) / 32, 1, p.groups), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

}  // namespace

bool gemm_skinny_eligible(const GemmP& p, int max_rows) {
    return p.taps == 1 && !p.rowbase && p.M >= 1 && p.M <= max_rows && p.M <= 64 && p.K >= 32 && (p.K & 31) == 0 &&
           (p.ldx & 3) == 0 && (p.ldw & 3) == 0 && (p.strideX & 3) == 0 && (p.strideW & 3) == 0 &&
           p.pro_act >= ACT_NONE && p.pro_act <= ACT_LRELU && p.a_mul >= 1 &&
           (reinterpret_cast<uintptr_t>(p.X) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0;
}

const char* gemm_skinny_name(const GemmP& p) { return p.M <= 32 ? "skinny32_f32" : "skinny64_f32"; }

hipError_t launch_gemm_skinny(const GemmP& p, hipStream_t s) {
    if (!gemm_skinny_eligible(p, 64)) return hipErrorInvalidValue;
    const int nw = (p.K % 256) == 0 ? 8 : ((p.K % 128) == 0 ? 4 : ((p.K % 64) == 0 ? 2 : 1));
    if (p.M <= 32) {
        switch (nw) {
            case 8: return sk_launch<1, 8, 4>(p, s);
            case 4: return sk_launch<1, 4, 4>(p, s);
            case 2: return sk_launch<1, 2, 4>(p, s);
            default: return sk_launch<1, 1, 4>(p, s);
        }
    }
    switch (nw) {
        case 8: return sk_launch<2, 8, 2>(p, s);
        case 4: return sk_launch<2, 4, 2>(p, s);
        case 2: return sk_launch<2, 2, 2>(p, s);
        default: return sk_launch<2, 1, 2>(p, s);
    }
}

}  // namespace mt2
