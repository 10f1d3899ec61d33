// f32 implicit-GEMM engine for gfx950 (CDNA4): every Linear / Conv1d / ConvTranspose1d of the
// Mega-TTS 2 synthesis path (A2-A9, A13, A15-A17, A19) runs through this kernel.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-exact k-ordered fma chain) - the
// reference computes in fp32 and its discrete decisions (VQ argmin, PLM argmax, ADM rounding) must
// be reproduced bit-for-bit, so reduced-precision MFMA is not used.  Roofline: 157.3 TFLOP/s.
//
// Tiling (wave64): a workgroup of WGM x WGN waves owns a BM x BN output tile; each wave owns
// (BM/WGM) x (BN/WGN) as TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each).  K is walked in
// chunks of 32: both operands are K-contiguous in HBM (activations [rows, C] time-major, weights
// [N, taps*Cin]), loaded as coalesced float4 (one 128-B line per 8 lanes), staged in LDS with a row
// stride of 36 floats so that the MFMA operand fetch - one ds_read_b128 per lane giving 4 k-values
// of one row - is bank-conflict free (36*r mod 64 is distinct for 16 consecutive rows).  The MFMA
// k index is a free permutation (lanes 0-31 take k = 8j+e, lanes 32-63 take k = 8j+4+e), so one
// b128 read feeds four MFMAs.  Global loads of chunk c+1 are issued before the MFMAs of chunk c
// and written to the other LDS buffer afterwards: one barrier per chunk.
//
// Conv1d is the same GEMM with a virtual A: A[m, tap*Cin + c] = X[src(m) + tap*dil, c]; gap rows
// between utterances supply the zero padding (mt2_kernels.h).  Epilogue: bias, activation, scale,
// residual add, gap-row mask, all fused.
#include "mt2_kernels.h"

#include <vector>

namespace mt2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACT>
__device__ __forceinline__ float apply_act(float v, float slope) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LRELU) return v >= 0.0f ? v : v * slope;
    if (ACT == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float act_rt(int act, float v, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_LRELU: return v >= 0.0f ? v : v * slope;
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

constexpr int BK = 32;   // K chunk (floats)
constexpr int LS = 36;   // LDS row stride (floats): 144 B = 9 x 16 B -> conflict-free ds_read_b128

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_f32_kernel(GemmP p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;   // float4 loads per thread per chunk
    constexpr int ROWS_PER_IT = NT / 8;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/thread mismatch");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be 32x32 MFMA tiles");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.z;

    // ---- tile id, XCD-aware: block b runs on XCD b%8 (observed); give each XCD a contiguous run of
    // tiles with the n-tiles of one m-tile adjacent, so the A panel is fetched into ONE XCD's L2.
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const float* __restrict__ X = p.X + (long long)g * p.strideX;
    const float* __restrict__ W = p.W + (long long)g * p.strideW;

    // ---- per-thread load coordinates: 8 lanes cover the 32 floats (128 B) of one row chunk
    const int col4 = tid & 7, row_in_it = tid >> 3;
    int abase[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int m = m0 + row_in_it + j * ROWS_PER_IT;
        int b = kInvalidRow;
        if (m < p.M) b = p.rowbase ? p.rowbase[m] : m * p.a_mul + p.shift0;
        abase[j] = b;
    }
    const int nk = (p.K + BK - 1) / BK;
    const int pro_act = p.pro_act;
    const float pro_slope = p.pro_slope;

    float4 ra[A_IT], rb[B_IT];
    auto load_chunk = [&](int kc) {
        const int k = kc * BK + col4 * 4;
        const bool kok = k < p.K;
        int tap = 0, c = k;
        if (p.taps > 1) { tap = k / p.Cin; c = k - tap * p.Cin; }
        const int shift = tap * p.dil;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int src = abase[j] + shift;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok && src >= 0 && src < p.Rx)
                v = *reinterpret_cast<const float4*>(X + (long long)src * p.ldx + c);
            if (pro_act != ACT_NONE) {
                v.x = act_rt(pro_act, v.x, pro_slope);
                v.y = act_rt(pro_act, v.y, pro_slope);
                v.z = act_rt(pro_act, v.z, pro_slope);
                v.w = act_rt(pro_act, v.w, pro_slope);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int n = n0 + row_in_it + j * ROWS_PER_IT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kok && n < p.N) v = *reinterpret_cast<const float4*>(W + (long long)n * p.ldw + k);
            rb[j] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float* As = smem + buf * (BM + BN) * LS;
        float* Bs = As + BM * LS;
#pragma unroll
        for (int j = 0; j < A_IT; ++j)
            *reinterpret_cast<float4*>(As + (row_in_it + j * ROWS_PER_IT) * LS + col4 * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            *reinterpret_cast<float4*>(Bs + (row_in_it + j * ROWS_PER_IT) * LS + col4 * 4) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int frag_off = (lane & 31) * LS + 4 * (lane >> 5);
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_chunk(kc + 1);
        const float* As = smem + buf * (BM + BN) * LS + (wm * WTM) * LS + frag_off;
        const float* Bs = smem + buf * (BM + BN) * LS + BM * LS + (wn * WTN) * LS + frag_off;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * LS + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LS + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (kc + 1 < nk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const float* __restrict__ bias = p.bias ? p.bias + (long long)g * p.strideB : nullptr;
    const float* __restrict__ R = p.R ? p.R + (long long)g * p.strideR : nullptr;
    float* __restrict__ C = p.C + (long long)g * p.strideC;
    const int epi_act = p.epi_act;
    const float out_scale = p.out_scale;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + (lane & 31);
        const bool nok = n < p.N;
        const float bv = (bias && nok) ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (nok && m < p.M) {
                    float v = acc[i][j][e] + bv;
                    v = act_rt(epi_act, v, 0.0f) * out_scale;
                    if (R) v += R[(long long)m * p.ldr + n];
                    if (p.valid && p.valid[m] == 0) v = 0.0f;
                    C[(long long)m * p.ldc + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side: tile-configuration choice and launch

struct TileCfg {
    int bm, bn, threads;
    const char* name;
    void (*fn)(GemmP);
};

#define MT2_CFG(BM_, BN_, WM_, WN_) \
    { BM_, BN_, WM_* WN_ * 64, #BM_ "x" #BN_ "_" #WM_ "x" #WN_, gemm_f32_kernel<BM_, BN_, WM_, WN_> }

static const TileCfg kCfgs[] = {
    MT2_CFG(128, 128, 2, 2),   // 64x64 per wave: the MFMA-bound workhorse (large M, N >= 128)
    MT2_CFG(64, 128, 2, 2),    // 32x64 per wave
    MT2_CFG(128, 64, 2, 2),    // 64x32 per wave
    MT2_CFG(64, 64, 2, 2),     // 32x32 per wave: small-M AR steps, fills the chip sooner
    MT2_CFG(32, 128, 1, 4),    // very small M (first AR steps, per-utterance heads)
    MT2_CFG(32, 64, 1, 2),
    MT2_CFG(128, 32, 4, 1),    // narrow N (HiFi-GAN 32-channel stage, conv_post)
    MT2_CFG(64, 32, 2, 1),
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

static thread_local const char* g_last_cfg = "";
const char* gemm_last_config() { return g_last_cfg; }

static int g_force_cfg = -1;
extern "C" void mt2_debug_force_gemm_config(int idx) { g_force_cfg = idx; }

static bool g_attr_done[kNumCfgs] = {};

// ---- launch trace (measurement only): HIP events around every GEMM launch, on the launch stream
struct TraceRec { int cfg; double flops; hipEvent_t e0, e1; };
static bool g_trace_on = false;
static std::vector<TraceRec> g_trace;

extern "C" int mt2_gemm_trace_begin(void) {
    for (auto& r : g_trace) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_trace.clear();
    g_trace_on = true;
    return 0;
}
// Per tile configuration: launches, executed FLOPs (2*M*N*K*groups) and summed kernel time (ms).
extern "C" int mt2_gemm_trace_end(int cap, const char** names, int64_t* launches, double* flops, double* ms) {
    g_trace_on = false;
    int n = 0;
    for (int i = 0; i < kNumCfgs && n < cap; ++i) {
        int64_t cnt = 0;
        double fl = 0.0, t = 0.0;
        for (auto& r : g_trace) {
            if (r.cfg != i) continue;
            if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
            float dt = 0.f;
            if (hipEventElapsedTime(&dt, r.e0, r.e1) != hipSuccess) return -1;
            ++cnt; fl += r.flops; t += dt;
        }
        if (cnt == 0) continue;
        names[n] = kCfgs[i].name; launches[n] = cnt; flops[n] = fl; ms[n] = t;
        ++n;
    }
    for (auto& r : g_trace) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_trace.clear();
    return n;
}

// Cost model (cycles per CU-slot): a tile costs max(MFMA time, operand-fetch time) per unit of K plus
// a fixed prologue/epilogue; the grid runs in ceil(tiles / 256) rounds (one tile per CU per round).
static const TileCfg* choose_cfg(const GemmP& p, int* idx_out) {
    double best = 1e300;
    int bi = 0;
    for (int i = 0; i < kNumCfgs; ++i) {
        const TileCfg& c = kCfgs[i];
        const long long tiles = (long long)((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn) * p.groups;
        const double rounds = (double)((tiles + 255) / 256);
        const double mfma = (double)c.bm * c.bn / 128.0;            // 128 MAC/clk/CU on f32 MFMA
        const double fetch = (double)(c.bm + c.bn) / 3.0;            // ~12 B/clk/CU of L2->LDS staging
        const double per_k = mfma > fetch ? mfma : fetch;
        const double cost = rounds * (per_k * p.K + 3000.0);
        if (cost < best) { best = cost; bi = i; }
    }
    if (g_force_cfg >= 0 && g_force_cfg < kNumCfgs) bi = g_force_cfg;
    *idx_out = bi;
    return &kCfgs[bi];
}

hipError_t launch_gemm(const GemmP& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.groups <= 0) return hipSuccess;
    if ((p.Cin & 3) || (p.ldx & 3) || (p.ldw & 3) || p.K != p.taps * p.Cin) return hipErrorInvalidValue;
    int idx = 0;
    const TileCfg* c = choose_cfg(p, &idx);
    const size_t lds = 2ull * (c->bm + c->bn) * LS * sizeof(float);
    if (!g_attr_done[idx]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c->fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        g_attr_done[idx] = true;
    }
    const int tiles = ((p.M + c->bm - 1) / c->bm) * ((p.N + c->bn - 1) / c->bn);
    dim3 grid(tiles, 1, p.groups), block(c->threads);
    g_last_cfg = c->name;
    if (g_trace_on) {
        TraceRec r;
        r.cfg = idx;
        r.flops = 2.0 * p.M * p.N * p.K * p.groups;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return hipErrorUnknown;
        (void)hipEventRecord(r.e0, s);
        hipLaunchKernelGGL(c->fn, grid, block, lds, s, p);
        (void)hipEventRecord(r.e1, s);
        g_trace.push_back(r);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(c->fn, grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace mt2
