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
template <int RT, int NW, int U>
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
                // (non-temporal weight loads measured SLOWER here - C1 65.2 vs 56.8 ms, round 3 - and were retired in round 6)
#pragma unroll
                for (int j = 0; j < 4; ++j) w[u][j] = *reinterpret_cast<const f32x4*>(wp + k + 4 * j);
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
    static std::atomic<unsigned long long> attr_done;         // per device (dyn_lds_once; zero-initialised)
    void (*fn)(GemmP) = gemm_skinny_f32_kernel<RT, NW, U>;
    const size_t lds = NW > 1 ? (size_t)NW * RT * 16 * 64 * sizeof(float) : 0;
    if (lds > 48 * 1024) {
        hipError_t e = dyn_lds_once(attr_done, reinterpret_cast<const void*>(fn), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3((p.N + 31) / 32, 1, p.groups), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}


// ---- round 4: tile-major weights, 16-column blocks, optional LayerNorm prologue ---------------------------------------------
//
// What the row-major kernel above leaves on the table at a handful of rows (profiles/r04_c1_kernel_stats_*.csv): (1) a
// wave-wide 16-byte load touches 64 separate pieces of 32 weight rows; (2) the f32 MFMA pipe runs at the VECTOR rate, so a
// 32 x 32 x K block costs K/2 x 64 cycles whatever M is - 3.9 us for K = 1024 on the two waves per SIMD of a workgroup, with
// only N/32 = 24...96 of the 256 CUs at work.  This form halves the block to 16 columns and uses v_mfma_f32_16x16x4_f32 on
// 16-row tiles: twice the workgroups, a quarter of the matrix-pipe time per workgroup at M <= 16 (the last-row launches of the
// batched AR steps, the first 16 steps of a lone utterance), 3/8 at M = 42.
//
// Weight layout (model_load.hip `TmRange`; kernel tests: launch_tile_major): 16-column x 64-k blocks of 1024 floats, block
// (nb, kb) at (nb * K/64 + kb) * 1024, inside a block
//     [j = 0..3][lane = 0..63][i = 0..3]  =  W[nb*16 + (lane & 15)][kb*64 + j*16 + (lane >> 4)*4 + i]
// so load instruction j of a wave reads 1 KiB of CONTIGUOUS memory and leaves lane (column, k-quarter) with the B values of
// four consecutive MFMAs; the activation side follows the same k order (lane (row, k-quarter) loads x[row][kb*64 + j*16 +
// quarter*4 .. +4]).  Exact f32 products, f32 accumulation in a fixed order (k-quarters inside an MFMA, MFMAs ascending,
// waves ascending in the LDS reduction).
//
// PRO_LN (GemmP::pro_act == 3): C = LN(X; ln_g, ln_b, ln_eps) W^T ... without a LayerNorm launch in front.  The workgroup
// first issues its first weight / activation loads (they do not depend on the statistics), then computes mean / rstd of ALL
// M <= 64 rows (wave w: rows w, w + NW, ..., four at a time; two passes in registers like layernorm_kernel; the rows come from
// L2 - the other column blocks read the same ones), and normalises the A values on the fly: a' = (a - mean) * rstd * gamma_k
// + beta_k.
template <int RT, int NW, int U, bool LNP>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_tm_kernel(GemmP p) {
    extern __shared__ float sk_red[];                       // [NW][RT][4][64] | stat [RT * 16][2]
    float* stat = sk_red + NW * RT * 4 * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.z, nb = blockIdx.x, n0 = nb * 16;
    const int col = lane & 15, kq = lane >> 4;
    // 64-wide K chunks of this wave: [c0, c0 + nc)
    const int kch = p.K >> 6, base = kch / NW, rem = kch % NW;
    const int nc = base + (wave < rem ? 1 : 0), c0 = wave * base + min(wave, rem);
    // block coordinates of this group's sub-matrix inside the whole tile-major matrix (16-row / 64-column units)
    const long long eo = (long long)g * p.strideW + (long long)p.tm_k0 * 64;
    const long long nbg = p.tm_n0 + ((eo / p.ldw) >> 4) + nb;
    const int kbg = (int)((eo % p.ldw) >> 6);
    const float* __restrict__ wp = p.Wtm + (nbg * p.tm_kb + kbg + c0) * 1024 + lane * 4;
    const float* __restrict__ xp[RT];
    bool xok[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = i * 16 + col;
        const int src = m * p.a_mul + p.shift0;
        xok[i] = m < p.M && src >= 0 && src < p.Rx;
        xp[i] = p.X + (long long)g * p.strideX + (long long)(xok[i] ? src : 0) * p.ldx + c0 * 64 + kq * 4;
    }
    const float slope = p.pro_slope;
    const bool relu = p.pro_act == ACT_RELU;
    const float ns = (LNP || p.pro_act == ACT_NONE) ? 1.0f : slope;

    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 w[U][4], a[U][RT][4];
    auto load_batch = [&](int sb) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (sb + u < nc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[u][j] = *reinterpret_cast<const f32x4*>(wp + (long long)(sb + u) * 1024 + j * 256);
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[u][i][j] = *reinterpret_cast<const f32x4*>(xp[i] + (sb + u) * 64 + j * 16);
            }
    };
    load_batch(0);

    float mu[RT], rs[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) { mu[i] = 0.0f; rs[i] = 1.0f; }
    if (LNP && p.ln_stat) {
        // Round 5: the rows came with their statistics - (mean, M2) pairs per 16-column block, written by the epilogue of the
        // GEMM that produced them (GemmP::stat_out, below).  Sixteen lanes per row merge its <= 64 pairs (Chan, fixed order:
        // two float4 = four pairs per lane at most) instead of every workgroup re-reading all M x K activations twice.
        const int nq = p.ln_nt >> 1, part = lane & 15;
        const float wc = (float)p.ln_w, inv_nt = 1.0f / (float)p.ln_nt;
        for (int r0 = wave * 4; r0 < p.M; r0 += NW * 4) {
            const int r = r0 + (lane >> 4);
            const int src = r * p.a_mul + p.shift0;
            const bool ok = r < p.M && src >= 0 && src < p.Rx;
            const f32x4* __restrict__ pr = reinterpret_cast<const f32x4*>(p.ln_stat) + (long long)(ok ? src : 0) * nq;
            f32x4 q0 = f32x4{0.f, 0.f, 0.f, 0.f}, q1 = q0;
            if (ok && part < nq) q0 = pr[part];
            if (ok && part + 16 < nq) q1 = pr[part + 16];
            float sm = (q0.x + q0.z) + (q1.x + q1.z);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o);
            const float mean = sm * inv_nt;
            float m2 = 0.0f;
            if (part < nq) { const float d0 = q0.x - mean, d1 = q0.z - mean; m2 += (q0.y + wc * d0 * d0) + (q0.w + wc * d1 * d1); }
            if (part + 16 < nq) { const float d0 = q1.x - mean, d1 = q1.z - mean; m2 += (q1.y + wc * d0 * d0) + (q1.w + wc * d1 * d1); }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m2 += __shfl_xor(m2, o);
            if (part == 0 && r < p.M) {
                stat[2 * r] = ok ? mean : 0.0f;
                stat[2 * r + 1] = ok ? rsqrtf(m2 * inv_nt / wc + p.ln_eps) : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = i * 16 + col;
            if (m < p.M) { mu[i] = stat[2 * m]; rs[i] = stat[2 * m + 1]; }
        }
    } else if (LNP) {
        const int Kf = p.K;
        const float inv_k = 1.0f / (float)Kf;
        const float* __restrict__ Xg = p.X + (long long)g * p.strideX;
        constexpr int RF = NW >= 12 ? 2 : 4;      // rows of this wave in flight: ONE memory round trip for up to 32 rows per workgroup
        for (int r0 = wave; r0 < p.M; r0 += RF * NW) {
            f32x4 xv[RF][4];
#pragma unroll
            for (int q = 0; q < RF; ++q) {
                const int r = r0 + q * NW;
                const int src = r * p.a_mul + p.shift0;
                const bool ok = r < p.M && src >= 0 && src < p.Rx;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = (v * 64 + lane) * 4;
                    xv[q][v] = (ok && c < Kf) ? *reinterpret_cast<const f32x4*>(Xg + (long long)src * p.ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            float sum[RF], q2[RF];
#pragma unroll
            for (int q = 0; q < RF; ++q) {
                sum[q] = 0.0f;
#pragma unroll
                for (int v = 0; v < 4; ++v) sum[q] += (xv[q][v].x + xv[q][v].y) + (xv[q][v].z + xv[q][v].w);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int q = 0; q < RF; ++q) sum[q] += __shfl_xor(sum[q], o);
#pragma unroll
            for (int q = 0; q < RF; ++q) {
                const float mean = sum[q] * inv_k;
                sum[q] = mean;
                q2[q] = 0.0f;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = (v * 64 + lane) * 4;
                    if (c < Kf) {
                        const float d0 = xv[q][v].x - mean, d1 = xv[q][v].y - mean, d2 = xv[q][v].z - mean, d3 = xv[q][v].w - mean;
                        q2[q] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int q = 0; q < RF; ++q) q2[q] += __shfl_xor(q2[q], o);
#pragma unroll
            for (int q = 0; q < RF; ++q) {
                const int r = r0 + q * NW;
                if (lane == 0 && r < p.M) {
                    stat[2 * r] = sum[q];
                    stat[2 * r + 1] = rsqrtf(q2[q] * inv_k + p.ln_eps);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = i * 16 + col;
            if (m < p.M) { mu[i] = stat[2 * m]; rs[i] = stat[2 * m + 1]; }
        }
    }

    for (int sb = 0; sb < nc; sb += U) {
        if (sb) load_batch(sb);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (sb + u < nc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 gam, bet;
                    if (LNP) {
                        const int k = (c0 + sb + u) * 64 + j * 16 + kq * 4;
                        gam = *reinterpret_cast<const f32x4*>(p.ln_g + k);
                        bet = *reinterpret_cast<const f32x4*>(p.ln_b + k);
                    }
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        f32x4 v = a[u][i][j];
                        if (LNP) {
                            v.x = (v.x - mu[i]) * rs[i] * gam.x + bet.x; v.y = (v.y - mu[i]) * rs[i] * gam.y + bet.y;
                            v.z = (v.z - mu[i]) * rs[i] * gam.z + bet.z; v.w = (v.w - mu[i]) * rs[i] * gam.w + bet.w;
                        } else {
                            v.x = fmaxf(v.x, relu ? 0.0f : v.x * ns); v.y = fmaxf(v.y, relu ? 0.0f : v.y * ns);
                            v.z = fmaxf(v.z, relu ? 0.0f : v.z * ns); v.w = fmaxf(v.w, relu ? 0.0f : v.w * ns);
                        }
                        if (!xok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, w[u][j].x, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, w[u][j].y, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, w[u][j].z, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, w[u][j].w, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // K shares meet in LDS; thread t finishes accumulator element (tile i, e, lane) = t of the RT * 256 outputs
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sk_red[((wave * RT + i) * 4 + e) * 64 + lane] = acc[i][e];
    __syncthreads();
    const float* __restrict__ bias = p.bias ? p.bias + (long long)g * p.strideB : nullptr;
    const float* __restrict__ R = p.R ? p.R + (long long)g * p.strideR : nullptr;
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int epi = p.epi_act;
    const float osc = p.out_scale;
    for (int t = threadIdx.x; t < RT * 256; t += NW * 64) {
        const int i = t >> 8, e = (t >> 6) & 3, ln = t & 63;
        float sacc = sk_red[((0 * RT + i) * 4 + e) * 64 + ln];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) sacc += sk_red[((w2 * RT + i) * 4 + e) * 64 + ln];
        const int m = i * 16 + (ln >> 4) * 4 + e, n = n0 + (ln & 15);
        float v = 0.0f;
        if (m < p.M && n < p.N) {
            v = sk_act(epi, sacc + (bias ? bias[n] : 0.0f), slope) * osc;
            if (R) v += R[(long long)m * p.ldr + n];
            if (p.valid && p.valid[m] == 0) v = 0.0f;
            C[(long long)m * p.ldc + n] = v;
        }
        if (p.stat_out) {       // (mean, M2) of this row's 16 columns: the 16 lanes that hold them are neighbours (t is wave-uniform in range)
            float sv = v, qv = v * v;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o); qv += __shfl_xor(qv, o); }
            if ((ln & 15) == 0 && m < p.M) {
                const float mean = sv * (1.0f / 16.0f);
                float2 pr2;
                pr2.x = mean;
                pr2.y = fmaxf(qv - sv * mean, 0.0f);
                *reinterpret_cast<float2*>(p.stat_out + ((long long)m * p.stat_nt + nb) * 2) = pr2;
            }
        }
    }
}

template <int RT, int U, bool LNP, int NW = 8>
hipError_t sk_tm_launch(const GemmP& p, hipStream_t s) {
    void (*fn)(GemmP) = gemm_skinny_tm_kernel<RT, NW, U, LNP>;
    const size_t lds = ((size_t)NW * RT * 4 * 64 + (size_t)RT * 16 * 2) * sizeof(float);      // <= 33 KiB with 8 waves; the sixteen-wave
    if (lds > 48 * 1024) {                                                                    // form with RT = 3 (M in 33..48) is 48.4 KiB
        static std::atomic<unsigned long long> done{0};                                       // (one per instantiation)
        const hipError_t e = dyn_lds_once(done, reinterpret_cast<const void*>(fn), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3(p.N / 16, 1, p.groups), dim3(NW * 64), lds, s, p);
    return hipGetLastError();
}

}  // namespace

bool gemm_skinny_eligible(const GemmP& p, int max_rows) {
    return p.taps == 1 && !p.rowbase && p.M >= 1 && p.M <= max_rows && p.M <= 64 && p.K >= 32 && (p.K & 31) == 0 &&
           (p.ldx & 3) == 0 && (p.ldw & 3) == 0 && (p.strideX & 3) == 0 && (p.strideW & 3) == 0 &&
           p.pro_act >= ACT_NONE && p.pro_act <= ACT_LRELU && p.a_mul >= 1 &&
           (reinterpret_cast<uintptr_t>(p.X) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0;
}

// tile-major form: whole 16-column x 64-k blocks only; the LayerNorm prologue exists in this form only
bool gemm_skinny_tm_eligible(const GemmP& p, int max_rows) {
    if (!p.Wtm || p.tm_kb <= 0) return false;
    const bool pairs_ok = !p.ln_stat || (p.ln_w == 16 && p.ln_nt * 16 == p.K && (p.ln_nt & 1) == 0 && p.ln_nt <= 64 &&
                                         (reinterpret_cast<uintptr_t>(p.ln_stat) & 15) == 0);
    const bool pro_ok = (p.pro_act >= ACT_NONE && p.pro_act <= ACT_LRELU) ||
                        (p.pro_act == 3 && p.ln_g && p.ln_b && p.K <= 1024 && p.groups == 1 && pairs_ok);
    return p.taps == 1 && !p.rowbase && p.M >= 1 && p.M <= max_rows && p.M <= 64 && p.K >= 64 && (p.K & 63) == 0 &&
           (p.N & 15) == 0 && (p.ldx & 3) == 0 && (p.ldw & 63) == 0 && (p.strideX & 3) == 0 && (p.strideW & 63) == 0 &&
           (p.groups == 1 || ((p.strideW % p.ldw) == 0 && ((p.strideW / p.ldw) & 15) == 0) ||
            (long long)(p.groups - 1) * p.strideW + (long long)p.tm_k0 * 64 + p.K <= p.ldw) &&
           pro_ok && p.a_mul >= 1 && (reinterpret_cast<uintptr_t>(p.X) & 15) == 0;
}

hipError_t launch_gemm_skinny_tm(const GemmP& p, hipStream_t s) {
    if (!gemm_skinny_tm_eligible(p, 64)) return hipErrorInvalidValue;
    const bool ln = p.pro_act == 3;
    const int per_wave = ((p.K >> 6) + 7) / 8;          // K chunks of the busiest wave
    // GemmP::sk_nw == 16 (option skinny_nw, default since round 5): SIXTEEN waves split K - four per SIMD, one 64-wide chunk each at
    // K = 1024 (a wave's dependent MFMA chain and its share of the loads halve; the LDS reduction sums 16 slabs in wave order) -
    // or TWELVE when K has only twelve chunks (the ADM's d = 768).  M <= 32 only (registers: 128 / 168 per wave).  Interleaved
    // A/B on the one-utterance path: C1 42.5 -> 41.75 ms (profiles/r05_opts_ab.txt).
    if (p.sk_nw == 16 && p.M <= 32 && (p.K >> 6) >= 16) {
        if (p.M <= 16) return ln ? sk_tm_launch<1, 1, true, 16>(p, s) : (per_wave > 2 ? sk_tm_launch<1, 2, false, 16>(p, s) : sk_tm_launch<1, 1, false, 16>(p, s));
        return ln ? sk_tm_launch<2, 1, true, 16>(p, s) : (per_wave > 2 ? sk_tm_launch<2, 2, false, 16>(p, s) : sk_tm_launch<2, 1, false, 16>(p, s));
    }
    if (p.sk_nw == 16 && p.M <= 32 && (p.K >> 6) == 12) {
        if (p.M <= 16) return ln ? sk_tm_launch<1, 1, true, 12>(p, s) : sk_tm_launch<1, 1, false, 12>(p, s);
        return ln ? sk_tm_launch<2, 1, true, 12>(p, s) : sk_tm_launch<2, 1, false, 12>(p, s);
    }
    // three row tiles (the ADM's steps 33 .. 48 of a lone utterance): twelve waves at K = 768 (168 registers per wave); sixteen at
    // K = 1024 for the plain prologue only (128 registers: the LayerNorm form and a second chunk in flight would spill)
    if (p.sk_nw == 16 && p.M > 32 && p.M <= 48 && (p.K >> 6) == 12)
        return ln ? sk_tm_launch<3, 1, true, 12>(p, s) : sk_tm_launch<3, 1, false, 12>(p, s);
    if (p.sk_nw == 16 && p.M > 32 && p.M <= 48 && (p.K >> 6) >= 16 && (p.K >> 6) <= 16 && !ln) return sk_tm_launch<3, 1, false, 16>(p, s);
    switch ((p.M + 15) / 16) {
        case 1:
            if (ln) return sk_tm_launch<1, 2, true>(p, s);
            return per_wave > 2 ? sk_tm_launch<1, 4, false>(p, s) : sk_tm_launch<1, 2, false>(p, s);
        case 2:
            if (ln) return sk_tm_launch<2, 2, true>(p, s);
            return per_wave > 2 ? sk_tm_launch<2, 4, false>(p, s) : sk_tm_launch<2, 2, false>(p, s);
        case 3:
            if (ln) return sk_tm_launch<3, 2, true>(p, s);
            return sk_tm_launch<3, 2, false>(p, s);       // U = 4 would spill (3 row tiles x 4 chunks of A in flight)
        default:
            if (ln) return sk_tm_launch<4, 2, true>(p, s);
            return sk_tm_launch<4, 2, false>(p, s);
    }
}

// row-major [N, K] (N a multiple of 16, K of 64) -> tile-major blocks (kernel tests and micro-benchmarks; the model loader
// repacks on the host)
__global__ void tile_major_kernel(const float* __restrict__ W, int K, float* __restrict__ out, long long n4) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of the output
    if (t >= n4) return;
    const long long blk = t >> 8;
    const int r = (int)(t & 255), j = r >> 6, lane = r & 63, c = lane & 15, kq = lane >> 4;
    const int KB = K >> 6;
    const long long nb = blk / KB;
    const int kb = (int)(blk % KB);
    *reinterpret_cast<f32x4*>(out + t * 4) = *reinterpret_cast<const f32x4*>(W + (nb * 16 + c) * K + kb * 64 + j * 16 + kq * 4);
}
hipError_t launch_tile_major(const float* W, int N, int K, float* out, hipStream_t s) {
    if ((N & 15) || (K & 63) || N <= 0 || K <= 0) return hipErrorInvalidValue;
    const long long n4 = (long long)N * K / 4;
    hipLaunchKernelGGL(tile_major_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, W, K, out, n4);
    return hipGetLastError();
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
