// Device-side pieces shared by the GEMM translation units (gemm_f32.hip: the f32-MFMA and bf16-pipe "x6" kernels and the tile
// table; gemm_x3h.hip: the fp16-pipe "x3h" kernels): activation helpers, the fused epilogues (plain, prefetched, 16-byte stores,
// row-statistics pairs), the LayerNorm pair merge, LDS read / counted-wait / loader-priority primitives.
#pragma once
#include "mt2_kernels.h"
#include "planes_store.h"
#include <type_traits>
#include <utility>

namespace mt2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float apply_act(float v, float slope) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LRELU) return fmaxf(v, v * slope);   // 0 < slope < 1: identical to v >= 0 ? v : v*slope, one VALU op less
    if (ACT == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float act_rt(int act, float v, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_LRELU: return v >= 0.0f ? v : v * slope;
        case ACT_TANH: return tanhf(v);
        case ACT_LOGCLAMP: return logf(fmaxf(v, slope));
        default: return v;
    }
}

// ---- LayerNorm statistics handed from GEMM to GEMM (GemmP::stat_out / ln_stat; round 5).
// rows_sum32: v[i] = the value of row i (i < NE, a power of two <= 16) in this lane's column (column = lane & 31; the two
// halves of the wave hold different rows and never mix: every xor mask is < 32).  Returns, in EVERY lane, the sum over the 32
// columns of row (lane & (NE - 1)).  A reduce-scatter butterfly: at each step a lane keeps half of its values and sends the
// other half to the partner that keeps those - NE - 1 exchanges instead of 5 NE - then plain butterflies over the remaining
// lane bits.  Fixed order: deterministic.
template <int NE>
__device__ __forceinline__ float rows_sum32(float (&v)[NE], int lane) {
#pragma unroll
    for (int w = NE / 2; w >= 1; w >>= 1) {
        const bool up = (lane & w) != 0;                  // lanes with this bit set keep the upper half of the index range
#pragma unroll
        for (int k = 0; k < w; ++k) {
            const float mine = up ? v[k + w] : v[k], send = up ? v[k] : v[k + w];
            v[k] = mine + __shfl_xor(send, w);
        }
    }
    float sum = v[0];
#pragma unroll
    for (int m = NE; m < 32; m <<= 1) sum += __shfl_xor(sum, m);
    return sum;
}
// Producer: the pair (mean_t, M2_t) of the W = 32 * NJ columns of this wave tile for each of the NE rows e0 .. e0 + NE - 1
// (accumulator element numbering) of the 32-row block at mw.  sv / qv: per-lane sum and sum of squares over the lane's NJ
// columns of the FINAL output values (0 outside M x N).  One single-pass (sum, sum of squares) per tile - its cancellation
// error is eps * (1 + mean_t^2 / var_t), harmless for a residual stream whose channel mean is of the order of its spread
// (measured: |mean| / std <= 2.2 per tile on the production models) - then Chan's merge across tiles on the consumer side.
template <int NE>
__device__ __forceinline__ void emit_row_pairs(const GemmP& p, float (&sv)[NE], float (&qv)[NE], int mw, int t, float w_cols,
                                               int lane, int e0) {
    const float S = rows_sum32<NE>(sv, lane), Q = rows_sum32<NE>(qv, lane);
    const int i = lane & 31;
    if (i < NE) {
        const int e = e0 + i;
        const int m = mw + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < p.M) {
            const float mean = S / w_cols;
            float2 pr;
            pr.x = mean;
            pr.y = fmaxf(Q - S * mean, 0.0f);
            *reinterpret_cast<float2*>(p.stat_out + ((long long)m * p.stat_nt + t) * 2) = pr;
        }
    }
}
// Consumer (pro_act == PRO_LNX): mean / rstd of source row `src` from its ln_nt pairs of ln_w columns each.  LPR lanes work
// on the SAME row (`part` = 0 .. LPR - 1 of the row's lane group, `xm` = the distance between the group's lanes: 1 for
// adjacent lanes, 32 when lane l and l ^ 32 pair up) and each takes every LPR-th float4 (two pairs) - all loads of a lane
// go out together (one memory round trip, overlapped with the ring prologue), the parts meet in a butterfly, the second
// pass (Chan: M2 = sum M2_t + w * sum (mean_t - mean)^2) runs on the registers.  ln_nt even, <= 32.
template <int LPR>
__device__ __forceinline__ void lnx_row_stats(const GemmP& p, int m, int part, int xm, float& mu, float& rs) {
    constexpr int NQ = 16 / LPR;                          // float4 loads per lane: 16 pairs-of-pairs at most per row
    int src = -1;
    if (m < p.M) src = p.rowbase ? p.rowbase[m] : m * p.a_mul + p.shift0;
    const bool ok = (unsigned)src < (unsigned)p.Rx;
    const int nq = p.ln_nt >> 1;
    const float4* __restrict__ pr = reinterpret_cast<const float4*>(p.ln_stat) + (long long)(ok ? src : 0) * nq;
    float4 q[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int qi = j * LPR + part;
        q[j] = (ok && qi < nq) ? pr[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float sm = 0.0f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) sm += q[j].x + q[j].z;
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) sm += __shfl_xor(sm, o * xm);
    const float mean = sm / (float)p.ln_nt, w = (float)p.ln_w;
    float m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int qi = j * LPR + part;
        if (qi < nq) {
            const float d0 = q[j].x - mean, d1 = q[j].z - mean;
            m2 += (q[j].y + w * d0 * d0) + (q[j].w + w * d1 * d1);
        }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) m2 += __shfl_xor(m2, o * xm);
    mu = ok ? mean : 0.0f;
    rs = ok ? 1.0f / sqrtf(m2 / ((float)p.ln_nt * w) + p.ln_eps) : 0.0f;
}

// Fused epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// EPG < 16: only accumulator elements e with e / EPG == esel are written (the K-split reduction shares the
// 16 elements of a 32x32 tile among the K groups; esel is wave-uniform).
template <int TM, int TN, int EPG = 16>
__device__ __forceinline__ void epilogue(const GemmP& p, f32x16 (&acc)[TM][TN], int g, int mw, int nw, int lane,
                                         int esel = 0) {
    const float* __restrict__ bias = p.bias ? p.bias + (long long)g * p.strideB : nullptr;
    const float* __restrict__ R = p.R ? p.R + (long long)g * p.strideR : nullptr;
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int epi_act = p.epi_act;
    const float epi_par = p.pro_slope;     // parameter of the epilogue activation (ACT_LOGCLAMP: the clip value)
    const float out_scale = p.out_scale;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + (lane & 31);
        const bool nok = n < p.N;
        const float bv = (bias && nok) ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (EPG < 16 && e / EPG != esel) continue;
                const int m = mw + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (nok && m < p.M) {
                    float v = acc[i][j][e] + bv;
                    v = act_rt(epi_act, v, epi_par) * out_scale;
                    if (R) v += R[(long long)m * p.ldr + n];
                    if (p.valid && p.valid[m] == 0) v = 0.0f;
                    C[(long long)m * p.ldc + n] = v;
                }
            }
        }
    }
}

// Epilogue operands fetched BEFORE the K loop (one 32x32 tile per wave only: 33 registers).  A launch of the
// autoregressive steps has at most one workgroup per CU, so the bias / residual / row-mask loads at the end
// of the kernel were ~1.5 us of exposed latency per launch; here they fly during the K loop.
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};
static __device__ int g_one_i[4] = {1, 1, 1, 1};
// A wave writes accumulator elements e = e0 .. e0+NE-1 of its 32x32 tile (all 16, or its 16/KS share after a
// K-split reduction; e0 is wave-uniform but a RUN-TIME value, so everything is indexed by i = e - e0).
template <int NE> struct EpiPre { float r[NE]; int v[NE]; float b; };
template <int NE>
__device__ __forceinline__ void epi_prefetch(const GemmP& p, EpiPre<NE>& q, int g, int mw, int nw, int lane, int e0) {
    // every load is unconditional from a safe address (absent operands point at constants): a conditional load
    // becomes a phi and hipcc folds the epilogue's compare into it, i.e. waits for each load where it is issued
    const int n = nw + (lane & 31);
    const bool nok = n < p.N;
    const float* __restrict__ bias = p.bias ? p.bias + (long long)g * p.strideB + (nok ? n : 0) : g_zero16;
    const float* __restrict__ R = p.R ? p.R + (long long)g * p.strideR + (nok ? n : 0) : g_zero16;
    const long long ldr = p.R ? p.ldr : 0;
    const int* __restrict__ vp = p.valid ? p.valid : g_one_i;
    const int vs = p.valid ? 1 : 0;
    q.b = *bias;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = e0 + i;
        int m = mw + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        m = m < p.M ? m : 0;
        q.r[i] = R[m * ldr];
        q.v[i] = vp[m * vs];
    }
}
template <int NE, bool STAT = false>
__device__ __forceinline__ void epilogue_pre(const GemmP& p, const float (&acc)[NE], const EpiPre<NE>& q, int g, int mw,
                                             int nw, int lane, int e0) {
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int epi_act = p.epi_act;
    const float epi_par = p.pro_slope;     // parameter of the epilogue activation (ACT_LOGCLAMP: the clip value)
    const float out_scale = p.out_scale;
    const bool hasR = p.R != nullptr;
    const int n = nw + (lane & 31);
    const bool nok = n < p.N;
    float sv[STAT ? NE : 1], qv[STAT ? NE : 1];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = e0 + i;
        const int m = mw + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        float v = 0.0f;
        if (nok && m < p.M) {
            v = acc[i] + q.b;
            v = act_rt(epi_act, v, epi_par) * out_scale;
            if (hasR) v += q.r[i];
            if (q.v[i] == 0) v = 0.0f;
            C[(long long)m * p.ldc + n] = v;
        }
        if constexpr (STAT) { sv[i] = v; qv[i] = v * v; }
    }
    if constexpr (STAT) emit_row_pairs<NE>(p, sv, qv, mw, nw >> 5, 32.0f, lane, e0);      // one pair per 32-column wave tile
}

// The same for TM x TN tiles per wave (no K split): residual, row mask and bias of the whole wave tile are
// requested before the K loop.  On the vocoder's residual convolutions the un-prefetched epilogue cost 11-25 %
// of the launch (1 workgroup per CU: nothing else covers the residual read).
template <int TM, int TN> struct EpiPreT { float r[TM][TN][16]; int v[TM][16]; float b[TN]; };
template <int TM, int TN>
__device__ __forceinline__ void epi_prefetch_t(const GemmP& p, EpiPreT<TM, TN>& q, int g, int mw, int nw, int lane) {
    const float* __restrict__ bias0 = p.bias ? p.bias + (long long)g * p.strideB : g_zero16;
    const float* __restrict__ R0 = p.R ? p.R + (long long)g * p.strideR : g_zero16;
    const long long ldr = p.R ? p.ldr : 0;
    const int* __restrict__ vp = p.valid ? p.valid : g_one_i;
    const int vs = p.valid ? 1 : 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + (lane & 31);
        const int nc = (n < p.N && (p.bias || p.R)) ? n : 0;
        q.b[j] = bias0[p.bias ? nc : 0];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int m = mw + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                m = m < p.M ? m : 0;
                q.r[i][j][e] = R0[m * ldr + (p.R ? nc : 0)];
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int m = mw + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            m = m < p.M ? m : 0;
            q.v[i][e] = vp[m * vs];
        }
}
template <int TM, int TN, bool STAT = false>
__device__ __forceinline__ void epilogue_pre_t(const GemmP& p, f32x16 (&acc)[TM][TN], const EpiPreT<TM, TN>& q, int g,
                                               int mw, int nw, int lane) {
    static_assert(!STAT || TM == 1, "row statistics: one 32-row block per wave");
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int epi_act = p.epi_act;
    const float epi_par = p.pro_slope;
    const float out_scale = p.out_scale;
    const bool hasR = p.R != nullptr;
    float sv[STAT ? 16 : 1], qv[STAT ? 16 : 1];
    if constexpr (STAT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { sv[e] = 0.0f; qv[e] = 0.0f; }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + (lane & 31);
        const bool nok = n < p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mw + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                float v = 0.0f;
                if (nok && m < p.M) {
                    v = acc[i][j][e] + q.b[j];
                    v = act_rt(epi_act, v, epi_par) * out_scale;
                    if (hasR) v += q.r[i][j][e];
                    if (q.v[i][e] == 0) v = 0.0f;
                    C[(long long)m * p.ldc + n] = v;
                }
                if constexpr (STAT) { sv[e] += v; qv[e] += v * v; }
            }
    }
    if constexpr (STAT) emit_row_pairs<16>(p, sv, qv, mw, nw / (32 * TN), 32.0f * TN, lane, 0);   // one pair per wave tile of 32 TN columns
}

// Epilogue with 16-byte stores.  The 32x32 MFMA leaves a lane with ONE column and 16 scattered rows, i.e. 16 dword stores per
// tile - the store tail of a wave that owns four tiles (64x64: the 256x128 tile, the one-wave-per-SIMD 128x128 tile) measured
// 11.6-17.3 us per tile (tools/x6_overheads.py: store ISSUE, not bandwidth).  Every group of four accumulator registers
// (rows R0..R0+3 of the lane's column) is a 4x4 block across the four lanes of a quad (columns 4k..4k+3): a two-stage DPP
// butterfly (lane xor 1 on register pairs (0,1) (2,3), lane xor 2 on (0,2) (1,3); 8 v_mov_dpp + 8 v_cndmask per block)
// transposes it, after which lane r of the quad holds row R0+r, columns 4k..4k+3 contiguously: one dwordx4 store (and one
// float4 load of bias / residual) instead of four dword stores.  Needs N, ldc, ldr multiples of 4 and 16-byte aligned
// bases (the caller checks, wave-uniformly, and falls back to `epilogue`).
__device__ __forceinline__ float dpp_quad(float v, const int ctrl) {
    return __builtin_bit_cast(float, ctrl == 0xB1 ? __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)
                                                  : __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ bool epilogue_t4_ok(const GemmP& p) {
    return ((p.N | p.ldc | (p.R ? p.ldr : 0)) & 3) == 0 && (((unsigned long long)p.C | (unsigned long long)p.R | (unsigned long long)p.bias) & 15) == 0 &&
           ((p.strideC | p.strideR | p.strideB) & 3) == 0;
}
// STAT (GemmP::stat_out, one 32-row block per wave): the (mean, M2) pair of every output row over the wave tile's 32 TN columns - after
// the transposition a lane holds 4 consecutive columns of row 8 q + rr, the row's other columns sit in the 7 lanes that differ in lane
// bits 2-4: three butterfly steps per quantity, fixed order.
// PLN: the kernel may be asked for fp16 planes instead of f32 (GemmP::c_planes; x3h loader tile).
template <int TM, int TN, bool STAT = false, bool PLN = false>
__device__ __forceinline__ void epilogue_t4(const GemmP& p, f32x16 (&acc)[TM][TN], int g, int mw, int nw, int lane) {
    static_assert(!STAT || TM == 1, "row statistics: one 32-row block per wave");
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int epi_act = p.epi_act;
    const float epi_par = p.pro_slope, out_scale = p.out_scale;
    const bool b0 = lane & 1, b1 = lane & 2;
    const int rr = (lane & 3) + 4 * (lane >> 5), c4 = ((lane & 31) >> 2) * 4;
    // Every operand load of the epilogue goes out FIRST, unconditionally and from a safe address (absent operands and rows / columns
    // beyond the matrix point at constants), so that the bias vectors, the row masks and the TM x TN x 4 residual vectors are ONE
    // memory round trip - a load behind a branch per row block (round 3's form) serialised them: 8.8 us instead of 4.4 for a
    // 128x128 tile with residual and row mask (profiles/r06_x3h_overheads.txt).  The K loop's operand registers are dead by now.
    const bool hasR = p.R != nullptr, hasB = p.bias != nullptr;
    const float* __restrict__ bias0 = hasB ? p.bias + (long long)g * p.strideB : g_zero16;
    const float* __restrict__ R0 = hasR ? p.R + (long long)g * p.strideR : g_zero16;
    const long long ldr = hasR ? p.ldr : 0;
    const int* __restrict__ vp = p.valid ? p.valid : g_one_i;
    const int vs = p.valid ? 1 : 0;
    f32x4 bv[TN], rv[TM][TN][4];
    int vm[TM][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + c4;
        bv[j] = *reinterpret_cast<const f32x4*>(bias0 + ((hasB && n < p.N) ? n : 0));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int m = mw + i * 32 + 8 * q + rr;
            m = m < p.M ? m : 0;
            vm[i][q] = vp[m * vs];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nw + j * 32 + c4;
                rv[i][j][q] = *reinterpret_cast<const f32x4*>(R0 + m * ldr + ((hasR && n < p.N) ? n : 0));
            }
        }
    float sv[STAT ? 4 : 1], qv[STAT ? 4 : 1];
    if constexpr (STAT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sv[q] = 0.0f; qv[q] = 0.0f; }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 32 + c4;
        const bool nok = n < p.N;                       // N is a multiple of 4: the four columns are in or out together
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = acc[i][j][4 * q], x1 = acc[i][j][4 * q + 1], x2 = acc[i][j][4 * q + 2], x3 = acc[i][j][4 * q + 3];
                const float p0 = dpp_quad(x0, 0xB1), p1 = dpp_quad(x1, 0xB1), p2 = dpp_quad(x2, 0xB1), p3 = dpp_quad(x3, 0xB1);
                const float y0 = b0 ? p1 : x0, y1 = b0 ? x1 : p0, y2 = b0 ? p3 : x2, y3 = b0 ? x3 : p2;
                const float q0 = dpp_quad(y0, 0x4E), q1 = dpp_quad(y1, 0x4E), q2 = dpp_quad(y2, 0x4E), q3 = dpp_quad(y3, 0x4E);
                f32x4 v = {b1 ? q2 : y0, b1 ? q3 : y1, b1 ? y2 : q0, b1 ? y3 : q1};
                const int m = mw + i * 32 + 8 * q + rr;
                const bool in = nok && m < p.M;
                if (in) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_rt(epi_act, v[e] + bv[j][e], epi_par) * out_scale;
                    if (hasR) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[i][j][q][e];
                    }
                    if (vm[i][q] == 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (PLN && p.c_planes) {
                        if (store_planes4(C + (long long)m * p.ldc, n, v[0], v[1], v[2], v[3]) >= 65504.0f && p.x3h_flag) atomicOr(p.x3h_flag, 1);
                    } else
                    *reinterpret_cast<f32x4*>(C + (long long)m * p.ldc + n) = v;
                }
                if constexpr (STAT) {                   // of the FINAL values (0 outside M x N)
                    if (!in) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    sv[q] += (v[0] + v[1]) + (v[2] + v[3]);
                    qv[q] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
        }
    }
    if constexpr (STAT) {
        const float w_cols = 32.0f * TN;
        const int t = nw / (32 * TN);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float S = sv[q], Q = qv[q];
#pragma unroll
            for (int o = 4; o < 32; o <<= 1) { S += __shfl_xor(S, o); Q += __shfl_xor(Q, o); }
            const int m = mw + 8 * q + rr;
            if ((lane & 28) == 0 && m < p.M) {
                const float mean = S / w_cols;
                float2 pr;
                pr.x = mean;
                pr.y = fmaxf(Q - S * mean, 0.0f);
                *reinterpret_cast<float2*>(p.stat_out + ((long long)m * p.stat_nt + t) * 2) = pr;
            }
        }
    }
}

constexpr int BK = 32;   // K chunk (floats)
constexpr int PRO_LN = 3;   // prologue kind: LayerNorm of the A rows (value of GemmP::pro_act; the <= 64-row weight-streaming kernel only)
constexpr int PRO_APL = 3;  // variant index of the x3h tiles whose A operand ARRIVES as fp16 planes (GemmP::a_planes; no prologue)
constexpr int PRO_LNA = 4;  // LayerNorm of the A rows, ALGEBRAIC form: statistics in the prologue, correction in the epilogue
constexpr int PRO_LNX = 5;  // ... ALGEBRAIC form on PAIR statistics written by the producer GEMM's epilogue (GemmP::ln_stat): no pass over K

__device__ __forceinline__ f32x4 lds_read_b128(unsigned byte_addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr) : "memory");
    return v;
}

// ... with a constant byte offset in the instruction's 16-bit offset field: the wave tile's second 32-row block, the weight planes
// and the column tiles are constants of the tile shape - folding them into the immediate takes a v_add (and its register) per
// read out of the K loop (round 5: the 256x128 loader tile sat at its 168-VGPR cap and reloaded a spilled address every chunk)
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128_imm(unsigned byte_addr) {
    static_assert(OFF >= 0 && OFF < 65536 && (OFF & 15) == 0, "ds_read_b128 immediate offset");
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF) : "memory");
    return v;
}
template <int... Is, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// issue priority of a LOADER wave (GemmP::ldr_prio, 0..3; the immediate of s_setprio must be a constant).  Round 4
// (tools/x6_prio.py, profiles/r04_x6_setprio.txt): a loader's few instructions per chunk - address update, LDS-DMA issue,
// counted wait, barrier - compete for issue slots with two compute waves' MFMA / VALU streams on the same SIMD; at priority 3
// the refill of a freed ring stage starts sooner: the AR shapes on the 128x128 loader tile -3.6 ... -5.5 % per launch.
__device__ __forceinline__ void loader_priority(int prio) {
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field on gfx9");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));   // expcnt/lgkmcnt: no wait
}

}  // namespace mt2
