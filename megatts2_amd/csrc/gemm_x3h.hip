// "x3h": the f32-equivalent GEMM engine on the fp16 matrix pipe of gfx950 (CDNA4) - THREE v_mfma_f32_32x32x16_f16 per
// 16-deep k block where the bf16 form ("x6", gemm_f32.hip) issues six.  Built as its own translation unit: the split
// arithmetic below wants scalar f32 VALU (v_mul_f32, v_fma_mix_f32), and this unit is compiled with -fno-slp-vectorize so that
// hipcc does not pack it into v_pk_mul_f32 / v_pk_fma_f32 (packed f32 VALU beside MFMAs is an anti-lever on this chip).
//
// Arithmetic (error-corrected tensor-core SGEMM: Markidis et al. 2018; Ootomo & Yokota 2022).  fp16 has an 11-bit significand.
//   a = a_hi + 2^-11 a_lo + da,   a_hi = fp16_rn(a),  a_lo = fp16_rn((a - a_hi) * 2^11),  |da| <= 2^-23 |a|   (2^-24 typical)
//   b likewise (weights: split once at load, x3h_planes.h, after an exact power-of-two scale per weight row)
//   a b = a_hi b_hi + 2^-11 (a_hi b_lo + a_lo b_hi) + [ 2^-22 a_lo b_lo + a db + b da ]      the bracket is dropped: <= 2^-22 |a b|
// The residual a - a_hi of a round-to-nearest conversion is exact in f32; scaled by 2^11 it has the magnitude of a again, so
// the low plane keeps 11 bits whatever the size of a (an unscaled low plane would sink into fp16's subnormals for |a| < 2^-3).
// A product of two fp16 is exact in f32 and the MFMA accumulates in f32.  The two scales live in TWO accumulators:
//   acc_hi += a_hi b_hi                       (1 MFMA per column tile)
//   acc_lo += a_hi b_lo + a_lo b_hi           (2 MFMAs per column tile)
//   C = (acc_hi + 2^-11 acc_lo) * inv_n       (one fma and one exact multiply per output element, before the fused epilogue)
// i.e. the result differs from an f32 fma chain by the summation order and by an operand representation error of <= 2^-23 per
// factor - the same class as x6 (tests/test_gpu_kernels.py::test_gemm_x3h_is_f32_equivalent holds it to the bar of the x6 test).
//
// Range.  fp16 ends at 65504.  Weights cannot leave it (row scale).  Activations are split at run time: every compute wave
// keeps the running max |a| of what it converts (one v_max3_f32 per element pair) and raises GemmP::x3h_flag when it reaches
// 65504 - the host then repeats the call on the x6 path (capi.inc), so an out-of-range activation costs time, never accuracy.
// NaN / inf inputs give NaN outputs (as the f32 chain gives NaN / inf) and inf raises the flag as well.  At the small end:
// |a| < 2^-14 makes a_hi subnormal, a_lo still holds the residual down to 2^-36 ABSOLUTE (fp16 subnormals are honoured by
// v_cvt_pk_f16_f32 and by the MFMA: tools/ubench/mfma_f16_denorm.hip) - such elements lose relative, not normwise accuracy.
//
// Operand path, tile map, loader waves, LDS swizzles: gemm_x6_ldr_kernel's (gemm_f32.hip) - the weights arrive as one 128-byte
// [hi | lo] block per row and chunk (x3h_planes.h), a stage of the 128x128 tile is 32 KiB instead of 40.  Differences that matter
// (round 6): the loaders issue BUFFER loads (32-bit lane offset computed once, the K walk in the scalar
// offset, out of range = zero rows); the compute waves of the loader tile and of the window convolution run their fragment
// pipeline ACROSS the chunk barrier; the epilogue is the 16-byte-store one with all operand loads at its start.
#include "gemm_common.h"

#include <atomic>

// measurement builds (tools/x3h_ablate.py): MT2_X3H_ABLATE = 2 no ingest inside the K loop, 3 no split arithmetic, 4 no
// matrix instructions, 5 = 2 + 3, 6 = 2 + 3 + no barrier inside the K loop (cross-chunk form), 7 = 2 + no barrier
#ifndef MT2_X3H_ABLATE
#define MT2_X3H_ABLATE 0
#endif
#define MT2_ABL_NOINGEST (MT2_X3H_ABLATE == 2 || MT2_X3H_ABLATE == 5 || MT2_X3H_ABLATE == 6 || MT2_X3H_ABLATE == 7)
#define MT2_ABL_NOSPLIT (MT2_X3H_ABLATE == 3 || MT2_X3H_ABLATE == 5 || MT2_X3H_ABLATE == 6)
#define MT2_ABL_NOBARRIER (MT2_X3H_ABLATE == 6 || MT2_X3H_ABLATE == 7)
// measurement build -DMT2_X3H_NO_BUFFER_LOADS: the loaders' 64-bit global_load_lds form everywhere (the A/B of the buffer-load form)
#ifdef MT2_X3H_NO_BUFFER_LOADS
#define MT2_BUFFER_LOADS false
#else
#define MT2_BUFFER_LOADS true
#endif

namespace mt2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kX3hLoScale = 2048.0f, kX3hLoInv = 1.0f / 2048.0f, kX3hMaxIn = 65504.0f;

// (lo, hi) = 8 consecutive f32 of one A row -> two planes of 8 fp16 (element 2i / 2i+1 in dword i).  6 VALU per element pair:
// v_cvt_pk_f16_f32, 2 v_mul_f32, 2 v_fma_mix_f32 (reads the fp16 halves in place), v_cvt_pk_f16_f32 - and one v_max3_f32 for
// the range guard; the bf16 split of x6 takes 11 per pair.
// Prologue activation of a value that is about to be SPLIT (its consumers are VALU instructions).  One v_max_f32: fmaxf() costs a
// second one in front of it, the canonicalisation IEEE mode asks of an operand that may be a signalling NaN - 8 of the 44 VALU
// instructions of a leaky-ReLU fragment split.  v_max_f32 returns the other operand for one NaN and NaN for two: fmaxf's result on
// every input (NaN * slope is NaN).  NOT for values that feed a matrix instruction directly: the compiler does not see the VALU write
// inside the asm statement and leaves out the wait states a VALU -> MFMA operand dependency needs (the f32-MFMA kernels keep fmaxf).
template <int ACT>
__device__ __forceinline__ float apply_act_split(float v, float slope) {
    if (ACT == ACT_RELU) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v)); return r; }
    if (ACT == ACT_LRELU) {            // 0 < slope < 1: identical to v >= 0 ? v : v*slope
        float r;
        const float w = v * slope;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(w));
        return r;
    }
    return apply_act<ACT>(v, slope);
}

template <int PRO>
__device__ __forceinline__ void split2_f16(const f32x4& lo, const f32x4& hi, float slope, u32x4& ph, u32x4& pl, float& amax) {
#if MT2_ABL_NOSPLIT     // ablation: no split arithmetic (wrong numbers, same MFMA / LDS / DMA work)
    ph = __builtin_bit_cast(u32x4, lo); pl = __builtin_bit_cast(u32x4, hi);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = apply_act_split<PRO>(i < 2 ? lo[2 * i] : hi[2 * i - 4], slope);
        const float y = apply_act_split<PRO>(i < 2 ? lo[2 * i + 1] : hi[2 * i - 3], slope);
        amax = fmaxf(fmaxf(fabsf(x), fabsf(y)), amax);
        const f16x2 h = __builtin_convertvector((f32x2){x, y}, f16x2);            // round to nearest even
        const float rx = __builtin_fmaf((float)h[0], -kX3hLoScale, x * kX3hLoScale);   // (x - h) * 2^11, exact
        const float ry = __builtin_fmaf((float)h[1], -kX3hLoScale, y * kX3hLoScale);
        const f16x2 l = __builtin_convertvector((f32x2){rx, ry}, f16x2);
        ph[i] = __builtin_bit_cast(unsigned, h);
        pl[i] = __builtin_bit_cast(unsigned, l);
    }
}

// NL loader waves refill the ring (f32 A rows + two fp16 weight planes), the WGM x WGN compute waves never issue a vector-memory
// instruction inside the K loop.  PRO: prologue activation of the A values (ACT_*), or PRO_LNX: the pair-fed algebraic LayerNorm /
// row-statistics epilogue form (K loop of ACT_NONE).
// One s_barrier per 32-deep chunk, and the compute waves' fragment pipeline runs ACROSS it (the K loop below).  (A form with one barrier
// per 64-deep super-chunk was built in round 6, parity-green, +1 % isolated and +2.4 % SLOWER in the model:
// profiles/r06_experiment_x3h_superchunk.patch.)
template <int BM, int BN, int WGM, int WGN, int NL, int NST, int PRO>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void gemm_x3h_ldr_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 8;                            // f32 A pieces per chunk: 8 rows x 128 B
    constexpr int PB = BN / 8;                            // weight pieces per chunk: 8 rows x 128 B ([hi | lo] blocks, x3h_planes.h)
    constexpr int A_IT = (PA + NL - 1) / NL, B_IT = (PB + NL - 1) / NL;   // per loader wave
    constexpr int L = A_IT + B_IT;
    constexpr int STAGE_A = BM * BK * 4, STAGE_B = PB * 1024, STAGE = STAGE_A + STAGE_B;   // bytes
    static_assert(PA % NL == 0 && PB % NL == 0 && WTM % 32 == 0 && WTN % 32 == 0 && NST >= 2 && (NST - 2) * L < 64, "config");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const unsigned long long t_entry = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;      // clock probe only

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const bool nmajor = p.M < p.N;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;
    const int Kt = p.K;
    const int nk = (Kt + BK - 1) / BK;

    if (wave_all >= NW) {
        // ------------------------------------------------------------------ loader wave lw: pieces lw, lw + NL, ...
        loader_priority(p.ldr_prio);
        const int lw = wave_all - NW;
        const float* __restrict__ X = p.X + (long long)g * p.strideX;
        const char* __restrict__ Wh = reinterpret_cast<const char*>(p.Wh) + (long long)g * p.wh_gstride;
        const long long zoff_x = (const float*)g_zero16 - X;
        const long long zoff_w = (const char*)g_zero16 - Wh;
        const int lrow = lane >> 3;
        const int ldx = p.ldx, Rx = p.Rx, Cin = p.Cin, dil = p.dil;
        const bool multi_tap = p.taps > 1;
        int abase[A_IT], akl[A_IT];
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int pc = j * NL + lw;                      // A piece: rows pc*8 .. pc*8+7
            const int m = m0 + pc * 8 + lrow;
            int b = kInvalidRow;
            if (m < p.M) b = p.rowbase ? p.rowbase[m] : m * p.a_mul + p.shift0;
            abase[j] = b;
            akl[j] = ((lane & 7) ^ ((pc * 4 + (lane >> 4)) & 7)) * 4;       // k offset of this lane's 16-byte slot
        }
        // B piece pc: weight rows pc*8 .. +7 of the tile, one 128-byte [hi | lo] block per row and chunk; the lane's 16-byte slot is
        // swizzled like an A row's (logical slot = physical ^ ((row >> 1) & 7); logical slots 0-3: hi of k = 8 s .., 4-7: lo)
        long long wofs[B_IT];        // BYTE offset of the lane's slot in chunk 0, or < 0 when the row is beyond N
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int pc = j * NL + lw;
            const int nl = pc * 8 + lrow, n = n0 + nl;
            const int sl = (lane & 7) ^ ((nl >> 1) & 7);
            wofs[j] = n < p.N ? (long long)n * p.wh_ldb + sl * 16 : -1;
        }
        wait_vmcnt<0>();                                     // the rowbase loads
        const bool fast = (Kt % BK == 0) && (!multi_tap || Cin % BK == 0);
        int s_tap = 0, s_cc = 0;
        // Whole chunks and operands below 2 GiB: BUFFER loads - the lane's byte offset of a piece is a 32-bit register computed once
        // (per tap), the walk along K is the instruction's SCALAR offset, a row outside the operand is the offset 2^31 = out of range
        // = zeros.  A piece then costs the loader an s_add (M0) and the load: the 64-bit per-lane address arithmetic of the
        // global_load form below (v_mad_i64, compares, selects: ~10 VALU instructions per piece, ~950 cycles per chunk and loader
        // wave on the SIMDs the compute waves split their fragments on) was the longest phase of a loader's chunk
        // (profiles/r06_x3h_phase_timing_v3_loader.txt).
        const bool fast32 = MT2_BUFFER_LOADS && !p.ldr64 && fast && (long long)Rx * ldx * 4 + (long long)Kt * 4 < 0x7fffffffll && (long long)p.N * p.wh_ldb < 0x7fffffffll;
        constexpr unsigned kOut = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)kOut, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wh), 0, (int)kOut, 0x00020000);
        unsigned voa[A_IT], vob[B_IT];
        auto row_offsets = [&](int dsrc) __attribute__((always_inline)) {     // the tap's rows: abase + tap * dil
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                const unsigned src = (unsigned)(abase[j] + dsrc);
                voa[j] = src < (unsigned)Rx ? (src * (unsigned)ldx + (unsigned)akl[j]) * 4u : kOut;
            }
        };
        row_offsets(0);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) vob[j] = wofs[j] >= 0 ? (unsigned)wofs[j] : kOut;
        // (always_inline: with more than two call sites hipcc emits the body as a FUNCTION - the closure, address state included, then
        // travels through scratch memory: measured 8x slower)
        auto issue = [&](int c, int st) __attribute__((always_inline)) {
            const int kchunk = c * BK;
            float* As = reinterpret_cast<float*>(ring + st * STAGE) + lw * 256;
            char* Bs = ring + st * STAGE + STAGE_A + lw * 1024;
            if (fast32) {
                const int so_x = s_cc * 4, so_w = kchunk * 4;
#pragma unroll
                for (int j = 0; j < A_IT; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(As + j * NL * 256), 16,
                                                             (int)voa[j], so_x, 0, 0);
#pragma unroll
                for (int j = 0; j < B_IT; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Bs + j * NL * 1024), 16,
                                                             (int)vob[j], so_w, 0, 0);
                s_cc += BK;
                if (multi_tap && s_cc == Cin) {          // next tap: the rows move by dil, their validity with them
                    s_cc = 0; ++s_tap;
                    row_offsets(s_tap * dil);
                }
                return;
            }
            if (fast) {
                const int dsrc = s_tap * dil;
#pragma unroll
                for (int j = 0; j < A_IT; ++j) {
                    const int src = abase[j] + dsrc;
                    const long long off = (unsigned)src < (unsigned)Rx ? (long long)src * ldx + (s_cc + akl[j]) : zoff_x;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + off),
                                                     (__attribute__((address_space(3))) void*)(As + j * NL * 256), 16, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < B_IT; ++j) {
                    const long long off = wofs[j] >= 0 ? wofs[j] + (long long)kchunk * 4 : zoff_w;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wh + off),
                                                     (__attribute__((address_space(3))) void*)(Bs + j * NL * 1024), 16, 0, 0);
                }
                s_cc += BK;
                if (multi_tap && s_cc == Cin) { s_cc = 0; ++s_tap; }
                return;
            }
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                const int k = kchunk + akl[j];
                int tap = 0, cc = k;
                if (multi_tap) { tap = k / Cin; cc = k - tap * Cin; }
                const int src = abase[j] + tap * dil;
                const bool ok = (k < Kt) & ((unsigned)src < (unsigned)Rx);
                const long long off = ok ? (long long)src * ldx + cc : zoff_x;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + off),
                                                 (__attribute__((address_space(3))) void*)(As + j * NL * 256), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < B_IT; ++j) {             // (the planes are zero-padded to whole chunks: no K tail on this side)
                const bool ok = (kchunk < Kt) & (wofs[j] >= 0);
                const long long off = ok ? wofs[j] + (long long)kchunk * 4 : zoff_w;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wh + off),
                                                 (__attribute__((address_space(3))) void*)(Bs + j * NL * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nk) issue(st, st);
#ifdef MT2_PHASE_TIMING
        // measurement build: where loader wave 0 of one workgroup spends its cycles (p.dbg[11..13]: vmcnt wait, barrier, issue)
        const bool lprobe = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && lw == 0;
        unsigned long long lacc[3] = {0, 0, 0}, lprev = lprobe ? __builtin_readcyclecounter() : 0ull;
#define MT2_LT(i_) do { if (lprobe) { const unsigned long long t_ = __builtin_readcyclecounter(); lacc[i_] += t_ - lprev; lprev = t_; } } while (0)
#define MT2_LT_END() do { if (lprobe && lane == 0) { p.dbg[11] = lacc[0]; p.dbg[12] = lacc[1]; p.dbg[13] = lacc[2]; } } while (0)
#else
#define MT2_LT(i_) do { } while (0)
#define MT2_LT_END() do { } while (0)
#endif
        // cross-chunk form: at barrier c the compute waves still have reads of chunk c-1 in flight (they run one chunk ahead of
        // their products) - the stage that is free is chunk c-2's, and chunk c + NST - 2 goes there
        static_assert(NST >= 3, "cross-chunk form: chunk c-1 is still being read at barrier c");
        for (int c = 0; c < nk; ++c) {
            if (c == 0) { if (NST - 2 < nk) wait_vmcnt<(NST - 2) * L>(); else wait_vmcnt<0>(); }
            else if (c + NST - 3 < nk) wait_vmcnt<(NST - 3) * L>();
            else wait_vmcnt<0>();
            MT2_LT(0);
            if (!MT2_ABL_NOBARRIER || c == 0)
            __builtin_amdgcn_s_barrier();                        // chunk c complete; chunk c-2's stage is free
            MT2_LT(1);
#if MT2_ABL_NOINGEST                   // ablation: no operand ingest inside the K loop
            if (c >= 1 && c + NST - 2 < nk && c < 2) {
#else
            if (c >= 1 && c + NST - 2 < nk) {
#endif
                const int cs = (c + NST - 2) % NST;
                issue(c + NST - 2, cs);
            }
            MT2_LT(2);
        }
        MT2_LT_END();
#undef MT2_LT
#undef MT2_LT_END
        return;
    }

    // ---------------------------------------------------------------------- compute wave
    const int wave = wave_all;
    const int wm = wave / WGN, wn = wave % WGN;
    // No epilogue operand is prefetched during the K loop: every variant takes the 16-byte-store epilogue (DPP-transposed 4 x 4 blocks,
    // float4 bias / residual loads issued together at its start, row statistics by butterflies over the transposed values): 4 stores
    // per 32x32 tile and lane instead of 16 - the dword-store tail of this tile measured 4.6 us of a 36-us AR launch; +5..7 % per
    // launch (profiles/r06_gemm_sweep_x3h_v4_t4_epilogue.txt) - and the 49 registers the prefetch spilled to scratch are free.
    // the inverse row scales of this lane's output columns (exact powers of two)
    float inv_s[TN];
    {
        const float* __restrict__ inv = p.wh_inv + (long long)g * p.wh_inv_stride;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + (lane & 31);
            inv_s[j] = inv[n < p.N ? n : 0];
        }
    }
    // pair-fed algebraic LayerNorm and row-statistics epilogue (GemmP::ln_stat / stat_out), as in gemm_x6_ldr_kernel
    constexpr bool LNXOK = TM == 1 && TM * TN <= 2 && PRO == PRO_LNX;
    const bool lnx = LNXOK && p.pro_act == PRO_LNX;
    [[maybe_unused]] float* lnstat = reinterpret_cast<float*>(ring + NST * STAGE);
    if constexpr (LNXOK) {
        if (lnx) {
            constexpr int LPR = NW * 64 / BM;
            static_assert(LPR >= 1 && LPR <= 16 && (LPR & (LPR - 1)) == 0, "lanes per row");
            const int tc = wave * 64 + lane, row = tc / LPR;
            float mu, rs;
            lnx_row_stats<LPR>(p, m0 + row, tc % LPR, 1, mu, rs);
            if (tc % LPR == 0) { lnstat[2 * row] = mu; lnstat[2 * row + 1] = rs; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the write is in LDS before this wave's first s_barrier
        }
    }

    f32x16 acc[TM][TN], acl[TM][TN];                      // a_hi b_hi | a_hi b_lo + a_lo b_hi (scaled 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; acl[i][j][e] = 0.0f; }

    const float pro_slope = p.pro_slope;
    const int half = lane >> 5;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    const int swza = (lane >> 1) & 7;
    const unsigned a_lane = lds0 + ((wm * WTM + (lane & 31)) * BK) * 4;
    const int nrow = wn * WTN + (lane & 31);
    const int swzb = (nrow >> 1) & 7;                     // (row + 32 j has the same swizzle: the column tiles ride in the offset field)
    const unsigned b_lane = lds0 + STAGE_A + nrow * 128;
    unsigned koffa[2][2], koffb[2][2];                    // koffb[b][plane]: logical slot plane * 4 + b * 2 + half of the row's 128-byte block
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        koffa[b][0] = (unsigned)(((b * 4 + half * 2) ^ swza) * 16);
        koffa[b][1] = (unsigned)(((b * 4 + half * 2 + 1) ^ swza) * 16);
        koffb[b][0] = (unsigned)(((b * 2 + half) ^ swzb) * 16);
        koffb[b][1] = (unsigned)(((4 + b * 2 + half) ^ swzb) * 16);
    }

    float amax = 0.0f;
    // clock probe (two scalar reads when requested): shader cycles (s_memtime) and constant-rate ticks (s_memrealtime) across
    // the whole K loop of one wave -> the clock the matrix pipe actually sustained (bench.py)
    const bool probe = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && wave == 0;
    unsigned long long treal0 = 0, tcyc0 = 0;
    if (probe) {
        treal0 = __builtin_amdgcn_s_memrealtime();
        tcyc0 = __builtin_readcyclecounter();
        if (lane == 0) p.dbg[9] = treal0 - t_entry;
    }
    // A operand as fp16 planes (GemmP::a_planes: the producer - a LayerNorm kernel - wrote [32 hi | 32 lo] per 32 k of a row, a weight
    // row's layout; same bytes, same strides as f32): the loaders move them unchanged, the compute waves read BOTH operands as ready
    // fragments - no split, no guard (the producer's), nothing but LDS reads and matrix instructions in the K loop.  Cross-chunk
    // pipeline without a split stage: k block q = 2 c + b lives in register set b; step q: lgkmcnt <= 2 TM + 2 TN (set b has landed),
    // the 3 TM TN products, [q even: s_barrier - chunk c + 1 has landed - behind the products], then A(q+2), B(q+2) into set b.  At
    // barrier c + 1 chunk c is still being read: the free stage is chunk c - 1's (the loader's protocol above).
    if constexpr (PRO == PRO_APL) {
        constexpr int N1 = 2 * TM + 2 * TN;
        static_assert(N1 <= 15, "lgkmcnt is four bits wide");
        u32x4 ra[2][TM][2];
        u32x4 rb[2][2][TN];
        unsigned koffah[2][2];                            // A fragment of k block b: hi = logical slot b * 2 + half, lo = 4 + b * 2 + half
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            koffah[b][0] = (unsigned)(((b * 2 + half) ^ swza) * 16);
            koffah[b][1] = (unsigned)(((4 + b * 2 + half) ^ swza) * 16);
        }
        auto fetch_a = [&](int b, unsigned sa) __attribute__((always_inline)) {
            const unsigned vh = sa + koffah[b][0], vl = sa + koffah[b][1];
            static_for(std::make_integer_sequence<int, TM>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ra[b][i][0] = __builtin_bit_cast(u32x4, lds_read_b128_imm<i * 32 * BK * 4>(vh));
                ra[b][i][1] = __builtin_bit_cast(u32x4, lds_read_b128_imm<i * 32 * BK * 4>(vl));
            });
        };
        auto fetch_b = [&](int b, unsigned sb) __attribute__((always_inline)) {
            const unsigned vb0 = sb + koffb[b][0], vb1 = sb + koffb[b][1];
            static_for(std::make_integer_sequence<int, TN>{}, [&](auto ic) {
                constexpr int j = decltype(ic)::value;
                rb[b][0][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<j * 32 * 128>(vb0));
                rb[b][1][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<j * 32 * 128>(vb1));
            });
        };
        auto tie_set = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(ra[b][i][0]), "+v"(ra[b][i][1]));
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rb[b][pl][j]));
        };
        auto products = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f16x8 Ah = __builtin_bit_cast(f16x8, ra[b][i][0]), Al = __builtin_bit_cast(f16x8, ra[b][i][1]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][1][j]), acl[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][0][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(f16x8, rb[b][0][j]), acl[i][j], 0, 0, 0);
            }
        };
        __builtin_amdgcn_s_barrier();                     // chunk 0 has landed
        asm volatile("" ::: "memory");
        fetch_a(0, a_lane); fetch_b(0, b_lane); fetch_a(1, a_lane); fetch_b(1, b_lane);
        __builtin_amdgcn_sched_barrier(0);
        int stn = 1 % NST;                                // stage of chunk c + 1
        auto chunk = [&](auto last_c) __attribute__((always_inline)) {
            constexpr bool last = decltype(last_c)::value;
            const unsigned sa = a_lane + (unsigned)stn * STAGE, sb = b_lane + (unsigned)stn * STAGE;    // chunk c + 1
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (last && b == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N1) : "memory");
                tie_set(b);
                __builtin_amdgcn_sched_barrier(0);
                products(b);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!last) {
                    if (b == 0) {
                        __builtin_amdgcn_s_barrier();     // chunk c + 1 has landed
                        asm volatile("" ::: "memory");
                    }
                    fetch_a(b, sa);                       // A(q + 2), B(q + 2)
                    fetch_b(b, sb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            stn = stn + 1 == NST ? 0 : stn + 1;
        };
        for (int c = 0; c + 1 < nk; ++c) chunk(std::false_type{});
        chunk(std::true_type{});
    } else {
    f32x4 ra[2][TM][2];
    u32x4 rb[2][2][TN];
    u32x4 pln[2][2];
    // VALU per split: 4 pairs x (6 + 1 guard) = 28 (+ the prologue activation); 3 TN MFMAs per fragment carry one split
    constexpr int NMF = 3 * TN, SPLIT_VALU = 28 + (PRO == ACT_RELU || PRO == ACT_LRELU ? 8 : 0);
    constexpr int VPM = (SPLIT_VALU + NMF - 1) / NMF;
    // the three products of fragment (b, i) with the column tiles of k-block b: cross terms first (into the low accumulator),
    // column tiles innermost so that consecutive MFMAs never wait on each other's accumulator
    auto products = [&](int b, int i, const u32x4* pp) {
        const f16x8 Ah = __builtin_bit_cast(f16x8, pp[0]), Al = __builtin_bit_cast(f16x8, pp[1]);
#if MT2_X3H_ABLATE == 4     // ablation: fetch + split, no matrix instructions (operands kept live)
        asm volatile("" :: "v"(Ah), "v"(Al));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(rb[b][0][j]), "v"(rb[b][1][j]));
        if (Kt >= 0) return;
#endif
#pragma unroll
        for (int j = 0; j < TN; ++j)
            acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][1][j]), acl[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][0][j]), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(f16x8, rb[b][0][j]), acl[i][j], 0, 0, 0);
    };
    auto tie = [&](int b, int i) { asm volatile("" : "+v"(ra[b][i][0]), "+v"(ra[b][i][1])); };
    auto pattern = [&]() {
#pragma unroll
        for (int k = 0; k < NMF; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
    };
    constexpr int FS = 2 * TM;                            // fragments per chunk (two 16-deep k blocks)
    {
        // Cross-chunk form.  The one-barrier-per-chunk loop below starts every chunk with all eight compute waves waiting on the
        // barrier, then on the first fragment's LDS latency (all waves fetch at once), then on its split - ~750 cycles in which no wave
        // has a matrix instruction to issue, against ~780 cycles of matrix work per chunk and SIMD (profiles/r06_x3h_phase_timing_v1.txt).
        // Here the stream of k blocks q = 2 c + b is pipelined without regard to chunks:
        //   step (q, i):  split fragment (q, i) + 1  ||  products of fragment (q, i)
        //   at the start of step (q, TM-1):  [q even: s_barrier - chunk c+1 has landed]  A(q+2) -> ra[q & 1]  (its last split was a step ago)
        //   at the end   of step (q, TM-1):  B(q+2) -> rb[q & 1]  (the last product of k block q has been issued)
        // so that A values are in flight for TM steps before their split and B fragments for TM steps before their first product;
        // LDS returns in order, so "A(q+1) landed" is lgkmcnt <= |B(q+1)| + |A(q+2)| = 2 TN + 2 TM, and so is "B(q+1) landed" at step
        // (q+1, 0) (behind it: A(q+2), B(q+2)).  No scalar memory instruction may sit in this loop (SMEM shares the counter and returns
        // out of order) - tools/asm_audit.py checks.  In the last chunk nothing is fetched and every wait is lgkmcnt(0).
        constexpr int N1 = 2 * TM + 2 * TN;
        static_assert(N1 <= 15, "lgkmcnt is four bits wide");
        auto fetch_a = [&](int b, unsigned sa) __attribute__((always_inline)) {
            const unsigned va0 = sa + koffa[b][0], va1 = sa + koffa[b][1];
            static_for(std::make_integer_sequence<int, TM>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ra[b][i][0] = lds_read_b128_imm<i * 32 * BK * 4>(va0);
                ra[b][i][1] = lds_read_b128_imm<i * 32 * BK * 4>(va1);
            });
        };
        auto fetch_b = [&](int b, unsigned sb) __attribute__((always_inline)) {
            const unsigned vb0 = sb + koffb[b][0], vb1 = sb + koffb[b][1];
            static_for(std::make_integer_sequence<int, TN>{}, [&](auto ic) {
                constexpr int j = decltype(ic)::value;
                rb[b][0][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<j * 32 * 128>(vb0));
                rb[b][1][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<j * 32 * 128>(vb1));
            });
        };
        auto tie_a = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) tie(b, i);
        };
        auto tie_b = [&](int b) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rb[b][pl][j]));
        };
        __builtin_amdgcn_s_barrier();                     // chunk 0 has landed
        asm volatile("" ::: "memory");
        fetch_a(0, a_lane); fetch_b(0, b_lane); fetch_a(1, a_lane); fetch_b(1, b_lane);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N1) : "memory");
        tie_a(0); tie_b(0);
        split2_f16<PRO>(ra[0][0][0], ra[0][0][1], pro_slope, pln[0][0], pln[0][1], amax);
        __builtin_amdgcn_sched_barrier(0);
        int stn = 1 % NST;                                // stage of chunk c + 1
        // one chunk of the stream; LAST: the launch's last chunk (nothing left to fetch) - peeled so that the steady-state body is
        // straight-line code
        auto chunk = [&](auto last_c) __attribute__((always_inline)) {
            constexpr bool last = decltype(last_c)::value;
            const unsigned sa = a_lane + (unsigned)stn * STAGE, sb = b_lane + (unsigned)stn * STAGE;    // chunk c + 1
#pragma unroll
            for (int s = 0; s < FS; ++s) {
                const int b = s / TM, i = s % TM;         // fragment (q = 2 c + b, i); its split halves are pln[s & 1]
                if (i == 0 && TM > 1) {                   // B(q) must have landed
                    if constexpr (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N1) : "memory");
                    tie_b(b);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i == TM - 1) {
                    if constexpr (!last) {
                        if (b == 0 && !MT2_ABL_NOBARRIER) {
                            __builtin_amdgcn_s_barrier();                 // chunk c + 1 has landed
                            asm volatile("" ::: "memory");
                        }
                        fetch_a(b, sa);                                   // A(q + 2)
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N1) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    tie_a(b ^ 1);                                         // A(q + 1)
                    if (TM == 1) tie_b(b);                                // B(q): issued before A(q + 1)
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bool more = s + 1 < FS || !last;                    // a fragment follows
                if (more) {
                    const int s2 = (s + 1) % FS, b2 = s2 / TM, i2 = s2 % TM;
                    if (b2 == b) tie(b2, i2);
                    split2_f16<PRO>(ra[b2][i2][0], ra[b2][i2][1], pro_slope, pln[(s + 1) & 1][0], pln[(s + 1) & 1][1], amax);
                    products(b, i, pln[s & 1]);
                    pattern();
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    products(b, i, pln[s & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!last) {
                    if (i == TM - 1) {
                        fetch_b(b, sb);                                   // B(q + 2)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            stn = stn + 1 == NST ? 0 : stn + 1;
        };
        for (int c = 0; c + 1 < nk; ++c) chunk(std::false_type{});
        chunk(std::true_type{});
    }
    }
    unsigned long long t_loop_end = 0;
    if (probe) {
        t_loop_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            p.dbg[6] = (unsigned long long)nk;
            p.dbg[7] = t_loop_end - treal0;
            p.dbg[8] = __builtin_readcyclecounter() - tcyc0;
        }
    }
    // range guard: an activation at or beyond the largest finite fp16 was converted somewhere in this wave's rows
    if (p.x3h_flag != nullptr && __builtin_amdgcn_ballot_w64(amax >= kX3hMaxIn) != 0ull && lane == 0) atomicOr(p.x3h_flag, 1);
    // C = (acc_hi + 2^-11 acc_lo) * inv_n
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_fmaf(acl[i][j][e], kX3hLoInv, acc[i][j][e]) * inv_s[j];

    if constexpr (LNXOK) {
        if (lnx) {                                        // rstd_r * (acc - mean_r * s_n); the bias operand is c
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + (lane & 31);
                const float s_n = n < p.N ? p.ln_g[n] : 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float2 st2 = reinterpret_cast<const float2*>(lnstat)[wm * WTM + (e & 3) + 8 * (e >> 2) + 4 * half];
                    acc[0][j][e] = st2.y * (acc[0][j][e] - st2.x * s_n);
                }
            }
        }
        // (launch_gemm admits the PRO_LNX variant only with N, ldc, ldr multiples of 4 and 16-byte aligned bases: epilogue_t4_ok)
        if (p.stat_out) epilogue_t4<TM, TN, true>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
        else epilogue_t4<TM, TN, false, true>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    } else if (p.epi_t4 && epilogue_t4_ok(p)) epilogue_t4<TM, TN, false, true>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    else epilogue<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    if (probe) {                                  // ticks spent in the epilogue (stores issued, not necessarily retired)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) p.dbg[10] = t_end - t_loop_end;
    }
}

// ===================================================================================================
// x3h arithmetic on the K-SPLIT tiles of the autoregressive steps (gemm_x6_ks_kernel's decomposition, gemm_f32.hip): KS groups of
// WGM x WGN waves, one 32x32 tile per wave, group kg walking chunks kg, kg + KS, ... through its OWN ring, NL loader waves own
// the whole refill, one barrier per round; the partial tiles (acc_hi + 2^-11 acc_lo of each group) are summed through LDS in a
// fixed order and each group finishes 16 / KS accumulator elements in the fused epilogue.  Per wave and k block: 3 MFMAs and one
// 28-VALU split instead of 6 and 44.  Linear layers only (taps = 1, K a multiple of 32 KS).
template <int BM, int BN, int WGM, int WGN, int KS, int NL, int NST, int PRO>
__global__ __launch_bounds__((WGM * WGN * KS + NL) * 64) void gemm_x3h_ks_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;                         // waves of one K group
    constexpr int NWC = NW * KS;                          // compute waves
    constexpr int PA = BM / 8, PB = BN / 8;               // 1-KiB pieces of one group's chunk: f32 A rows, [hi | lo] weight blocks (8 rows x 128 B)
    constexpr int PG = PA + PB, PT = PG * KS;             // pieces per round (all groups)
    constexpr int LW = PT / NL;                           // pieces per loader wave and round
    constexpr int STAGE_A = BM * BK * 4, STAGE_B = PB * 1024, STAGE = STAGE_A + STAGE_B;   // bytes, one group's stage
    constexpr int EPG = 16 / KS;
    static_assert(BM == 32 * WGM && BN == 32 * WGN, "one 32x32 tile per wave");
    static_assert(PT % NL == 0 && 16 % KS == 0 && NST >= 2 && (NST - 2) * LW < 64 && NWC + NL <= 16, "config");
    static_assert(KS * NST * STAGE >= NWC * 16 * 64 * 4 || KS == 1, "the K-group reduction reuses the ring");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const unsigned long long t_entry = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;      // overhead probe only (tools/x3h_overheads.py)
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const bool nmajor = p.M < p.N;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;
    const int nr = p.K / (BK * KS);                       // rounds (launch_gemm guarantees K % (32 KS) == 0)

    if (wave_all >= NWC) {
        // ------------------------------------------------------------------ loader wave lw: pieces lw, lw + NL, ... of a round
        loader_priority(p.ldr_prio);
        const int lw = wave_all - NWC;
        const char* __restrict__ Xb = reinterpret_cast<const char*>(p.X + (long long)g * p.strideX);
        const char* __restrict__ Wb = reinterpret_cast<const char*>(p.Wh) + (long long)g * p.wh_gstride;
        const char* zero = reinterpret_cast<const char*>(g_zero16);
        const char* base[LW];            // operand base of the piece (X or Wh)
        long long rowb[LW];              // byte offset of the lane's row + its 16-byte slot inside a chunk, < 0: zero row
        int ldsoff[LW];                  // LDS byte offset of the piece inside its group's stage (both operands: 128 bytes per row and chunk)
        int grp[LW];
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int pi = j * NL + lw, kg = pi / PG, w = pi - kg * PG;
            grp[j] = kg;
            if (w < PA) {                                  // A piece: rows w*8 .. +7, 8 lanes per 128-byte row
                const int m = m0 + w * 8 + (lane >> 3);
                int src = kInvalidRow;
                if (m < p.M) src = p.rowbase ? p.rowbase[m] : m * p.a_mul + p.shift0;
                const int kl = ((lane & 7) ^ ((w * 4 + (lane >> 4)) & 7)) * 4;
                base[j] = Xb; ldsoff[j] = w * 1024;
                rowb[j] = (unsigned)src < (unsigned)p.Rx ? ((long long)src * p.ldx + kl) * 4 : -1;
            } else {                                       // B piece: weight rows bp*8 .. +7, one 128-byte [hi | lo] block per row
                const int bp = w - PA;
                const int nl = bp * 8 + (lane >> 3), n = n0 + nl;
                const int sl = (lane & 7) ^ ((nl >> 1) & 7);
                base[j] = Wb; ldsoff[j] = STAGE_A + bp * 1024;
                rowb[j] = n < p.N ? (long long)n * p.wh_ldb + sl * 16 : -1;
            }
        }
        wait_vmcnt<0>();                                   // the rowbase loads
        // operands below 2 GiB: buffer loads - per piece a 32-bit lane offset (row + slot + the K group's chunk), the round as the
        // scalar offset, zero rows as the out-of-range offset 2^31 (gemm_x3h_ldr_kernel's loader, above)
        constexpr unsigned kOut = 0x80000000u;
        const bool fast32 = MT2_BUFFER_LOADS && !p.ldr64 && (long long)p.Rx * p.ldx * 4 + (long long)p.K * 4 < 0x7fffffffll && (long long)p.N * p.wh_ldb < 0x7fffffffll;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Xb), 0, (int)kOut, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wb), 0, (int)kOut, 0x00020000);
        unsigned vo[LW];
#pragma unroll
        for (int j = 0; j < LW; ++j) vo[j] = rowb[j] >= 0 ? (unsigned)rowb[j] + (unsigned)grp[j] * (BK * 4) : kOut;
        auto issue = [&](int rd, int st) __attribute__((always_inline)) {
            if (fast32) {
                const int so = rd * KS * BK * 4;
#pragma unroll
                for (int j = 0; j < LW; ++j) {
                    const bool is_a = base[j] == Xb;               // (wave-uniform: the piece index depends on the wave only)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rs_x : rs_w,
                                                             (__attribute__((address_space(3))) void*)(ring + (grp[j] * NST + st) * STAGE + ldsoff[j]),
                                                             16, (int)vo[j], so, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < LW; ++j) {
                const long long kb = (long long)(rd * KS + grp[j]) * BK * 4;
                const char* src = rowb[j] >= 0 ? base[j] + rowb[j] + kb : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ring + (grp[j] * NST + st) * STAGE + ldsoff[j]),
                                                 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nr) issue(st, st);
#ifdef MT2_PHASE_TIMING
        // measurement build: where loader wave 0 of one workgroup spends its cycles (p.dbg[11..13]: vmcnt wait, barrier, issue)
        const bool lprobe = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && lw == 0;
        unsigned long long lacc[3] = {0, 0, 0}, lprev = lprobe ? __builtin_readcyclecounter() : 0ull;
#define MT2_LT(i_) do { if (lprobe) { const unsigned long long t_ = __builtin_readcyclecounter(); lacc[i_] += t_ - lprev; lprev = t_; } } while (0)
#else
#define MT2_LT(i_) do { } while (0)
#endif
        int st = 0;
        for (int rd = 0; rd < nr; ++rd) {
            if (rd + NST - 2 < nr) wait_vmcnt<(NST - 2) * LW>();
            else wait_vmcnt<0>();
            MT2_LT(0);
            __builtin_amdgcn_s_barrier();                  // round rd complete in LDS; round rd-1's stages are free
            MT2_LT(1);
            if (rd + NST - 1 < nr) issue(rd + NST - 1, st == 0 ? NST - 1 : st - 1);
            MT2_LT(2);
            st = st + 1 == NST ? 0 : st + 1;
        }
#ifdef MT2_PHASE_TIMING
        if (lprobe && lane == 0) { p.dbg[11] = lacc[0]; p.dbg[12] = lacc[1]; p.dbg[13] = lacc[2]; }
#endif
#undef MT2_LT
        return;
    }

    // ---------------------------------------------------------------------- compute wave: K group kg, tile (wm, wn)
    const int kg = wave_all / NW, wave = wave_all % NW;
    const int wm = wave / WGN, wn = wave % WGN;
    EpiPre<EPG> pre;
    epi_prefetch<EPG>(p, pre, g, m0 + wm * 32, n0 + wn * 32, lane, kg * EPG);
    float inv_s;
    {
        const int n = n0 + wn * 32 + (lane & 31);
        inv_s = (p.wh_inv + (long long)g * p.wh_inv_stride)[n < p.N ? n : 0];
    }
    float ln_mu = 0.0f, ln_rs = 0.0f;
    const bool lnx = PRO == PRO_LNX && p.pro_act == PRO_LNX;
    if (lnx) lnx_row_stats<2>(p, m0 + wm * 32 + (lane & 31), lane >> 5, 32, ln_mu, ln_rs);      // lanes l and l ^ 32 share a row
    f32x16 acc, acl;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.0f; acl[e] = 0.0f; }
    const float pro_slope = p.pro_slope;
    const int half = lane >> 5;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring + (unsigned)(kg * NST) * STAGE;
    const int swza = (lane >> 1) & 7;
    const unsigned a_lane = lds0 + ((wm * 32 + (lane & 31)) * BK) * 4;
    const int nrow = wn * 32 + (lane & 31);
    const int swzb = (nrow >> 1) & 7;
    const unsigned b_lane = lds0 + STAGE_A + nrow * 128;
    unsigned koffa[2][2], koffb[2][2];                    // koffb[b][plane]: logical slot plane * 4 + b * 2 + half
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        koffa[b][0] = (unsigned)(((b * 4 + half * 2) ^ swza) * 16);
        koffa[b][1] = (unsigned)(((b * 4 + half * 2 + 1) ^ swza) * 16);
        koffb[b][0] = (unsigned)(((b * 2 + half) ^ swzb) * 16);
        koffb[b][1] = (unsigned)(((4 + b * 2 + half) ^ swzb) * 16);
    }
    float amax = 0.0f;
    f32x4 ra[2][2];
    u32x4 rb[2][2], pln[2];
    // (PRO_APL - GemmP::a_planes: the A rows arrive as [32 hi | 32 lo] fp16 per 32 k, a weight row's layout: the fragment of k block
    // b is the hi slot b * 2 + half and the lo slot 4 + b * 2 + half, and there is nothing to split)
    auto fetch = [&](int b, unsigned sa, unsigned sb) {
        ra[b][0] = lds_read_b128(sa + (PRO == PRO_APL ? (unsigned)(((b * 2 + half) ^ swza) * 16) : koffa[b][0]));
        ra[b][1] = lds_read_b128(sa + (PRO == PRO_APL ? (unsigned)(((4 + b * 2 + half) ^ swza) * 16) : koffa[b][1]));
        rb[b][0] = __builtin_bit_cast(u32x4, lds_read_b128(sb + koffb[b][0]));
        rb[b][1] = __builtin_bit_cast(u32x4, lds_read_b128(sb + koffb[b][1]));
    };
    auto wait_block = [&](int b) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(ra[b][0]), "+v"(ra[b][1]), "+v"(rb[b][0]), "+v"(rb[b][1]));
    };
    auto products = [&](int b) {
        const f16x8 Ah = __builtin_bit_cast(f16x8, pln[0]), Al = __builtin_bit_cast(f16x8, pln[1]);
        acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][1]), acl, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][0]), acc, 0, 0, 0);
        acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(f16x8, rb[b][0]), acl, 0, 0, 0);
    };
    const bool probe = p.dbg != nullptr && blockIdx.x == gridDim.x / 2 && wave_all == 0;
    unsigned long long treal0 = 0, tcyc0 = 0;
    if (probe) {
        treal0 = __builtin_amdgcn_s_memrealtime();
        tcyc0 = __builtin_readcyclecounter();
        if (lane == 0) p.dbg[9] = treal0 - t_entry;
    }
    int st = 0;
    for (int rd = 0; rd < nr; ++rd) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned sa = a_lane + (unsigned)st * STAGE, sb = b_lane + (unsigned)st * STAGE;
        fetch(0, sa, sb);
        fetch(1, sa, sb);
        __builtin_amdgcn_sched_barrier(0);
        // (a form with the fragment pipeline running across the rounds - the cross-chunk form of gemm_x3h_ldr_kernel - measured EQUAL,
        // +-2 %: these tiles are bound by the ingest, KS x 8..16 KiB per round at the 64 B/clk of a CU's LDS-DMA path = 512..1 024
        // cycles against 384 of matrix work; profiles/r06_experiment_x3h_ks_cross_round.patch, r06_x3h_phase_timing_v4_*.txt)
        // one tile per wave: a split does not fit in the shadow of its three MFMAs - split and multiply in turn, the other
        // waves of the SIMD (another K group) fill the gaps
        wait_block(0);
        wait_block(1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRO == PRO_APL) { pln[0] = __builtin_bit_cast(u32x4, ra[0][0]); pln[1] = __builtin_bit_cast(u32x4, ra[0][1]); }
        else split2_f16<PRO>(ra[0][0], ra[0][1], pro_slope, pln[0], pln[1], amax);
        products(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRO == PRO_APL) { pln[0] = __builtin_bit_cast(u32x4, ra[1][0]); pln[1] = __builtin_bit_cast(u32x4, ra[1][1]); }
        else split2_f16<PRO>(ra[1][0], ra[1][1], pro_slope, pln[0], pln[1], amax);
        products(1);
        __builtin_amdgcn_sched_barrier(0);
        st = st + 1 == NST ? 0 : st + 1;
    }
    unsigned long long t_loop_end = 0;
    if (probe) {
        t_loop_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            p.dbg[6] = (unsigned long long)(nr * KS);
            p.dbg[7] = t_loop_end - treal0;
            p.dbg[8] = __builtin_readcyclecounter() - tcyc0;
        }
    }
    if (p.x3h_flag != nullptr && __builtin_amdgcn_ballot_w64(amax >= kX3hMaxIn) != 0ull && lane == 0) atomicOr(p.x3h_flag, 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = __builtin_fmaf(acl[e], kX3hLoInv, acc[e]);
    // ---- sum the KS partial tiles through LDS (every DMA has been waited for; the barrier orders the last operand reads),
    // fixed order kg = 0..KS-1; group kg finishes elements e = kg*EPG .. kg*EPG+EPG-1
    if constexpr (KS > 1) {
        __syncthreads();                                  // (the loader waves have left: the barrier counts live waves only)
        float* red = smem + ((kg * NW + wave) * 16) * 64 + lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) red[e * 64] = acc[e];
        __syncthreads();
        float out[EPG];
#pragma unroll
        for (int i = 0; i < EPG; ++i) {
            const int e = kg * EPG + i;
            float v = 0.0f;
#pragma unroll
            for (int g2 = 0; g2 < KS; ++g2) v += smem[(((g2 * NW + wave) * 16) + e) * 64 + lane];
            out[i] = v * inv_s;
        }
        if (lnx) {                                        // rstd_r * (acc - mean_r * s_n); the bias operand is c
            const int n = n0 + wn * 32 + (lane & 31);
            const float s_n = n < p.N ? p.ln_g[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < EPG; ++i) {
                const int e = kg * EPG + i, rr = (e & 3) + 8 * (e >> 2) + 4 * half;
                out[i] = __shfl(ln_rs, rr) * (out[i] - __shfl(ln_mu, rr) * s_n);
            }
        }
        if (PRO == PRO_LNX && p.stat_out) epilogue_pre<EPG, PRO == PRO_LNX>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, kg * EPG);
        else epilogue_pre<EPG>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, kg * EPG);
    } else {
        float out[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) out[e] = acc[e] * inv_s;
        if (lnx) {
            const int n = n0 + wn * 32 + (lane & 31);
            const float s_n = n < p.N ? p.ln_g[n] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * half;
                out[e] = __shfl(ln_rs, rr) * (out[e] - __shfl(ln_mu, rr) * s_n);
            }
        }
        if (PRO == PRO_LNX && p.stat_out) epilogue_pre<16, PRO == PRO_LNX>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, 0);
        else epilogue_pre<16>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, 0);
    }
    if (probe) {                                  // ticks from the end of the K loop to the last store issued (reduction + epilogue)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) p.dbg[10] = t_end - t_loop_end;
    }
}

// ===================================================================================================
// The WINDOW convolution (conv_win_x6_kernel, gemm_f32.hip: narrow square "same" convolutions of the vocoder's resblocks, Cin =
// Cout = 32 QS; the f32 input window of the row tile sits in LDS once, the weights stream through the ring) in the x3h form: the
// weights arrive as two fp16 planes (2 x BN x 64 B per chunk), every A fragment is split into two fp16 planes in registers.
// NL loader waves refill the weight ring (and help load the window); the compute waves' fragment pipeline runs across the chunk
// barrier (gemm_x3h_ldr_kernel's cross-chunk form).
template <int QS, int BM, int BN, int WGM, int WGN, int NST, int PRO, int NL>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void conv_win_x3h_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BPIECES = BN / 8;                       // 1-KiB pieces of one weight chunk: 8 rows x 128 B ([hi | lo] blocks)
    constexpr int NI = NL;                                // waves that issue the ring refill: the NL loader waves
    constexpr int B_IT = (BPIECES + NI - 1) / NI;
    constexpr int STAGE_B = B_IT * NI * 1024;             // BYTES per ring stage (dummy slots included)
    static_assert(NL > 0 && WTM % 32 == 0 && WTN % 32 == 0 && NST >= 3 && (NST - 2) * B_IT < 64 && BN == 32 * QS, "config");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave_all >= NW;
    const int wave = loader ? wave_all - NW : wave_all;   // index among the loaders / among the compute waves
    const int wm = wave / WGN, wn = wave % WGN;
    const int taps = p.taps, dil = p.dil;
    const int WR = BM + (taps - 1) * dil, WRp = (WR + 7) & ~7;
    char* ring = reinterpret_cast<char*>(smem);
    float* win = smem + NST * STAGE_B / 4;                          // [QS][WRp][32] f32, slot-swizzled rows

    const int ntm = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qq = ntm >> 3, rr = ntm & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    const int m0 = tile * BM;

    const float* __restrict__ X = p.X;
    const char* __restrict__ Wh = reinterpret_cast<const char*>(p.Wh);
    const long long zoff_x = (const float*)g_zero16 - X;
    const long long zoff_w = (const char*)g_zero16 - Wh;
    const int ldx = p.ldx, Rx = p.Rx, Kt = p.K;

    {   // ---- the f32 input window, once
        const int lrow = lane >> 3;
        const int ppq = WRp >> 3, pieces = QS * ppq;
        const int row_first = m0 + p.shift0;
        for (int pc = wave_all; pc < pieces; pc += NW + NL) {
            const int q = pc / ppq, r8 = pc - q * ppq;
            const int row = r8 * 8 + lrow;
            const int grow = row_first + row;
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            const bool ok = (row < WR) & ((unsigned)grow < (unsigned)Rx);
            const long long off = ok ? (long long)grow * ldx + q * 32 + slot * 4 : zoff_x;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + off),
                                             (__attribute__((address_space(3))) void*)(win + (q * WRp + r8 * 8) * 32), 16, 0, 0);
        }
    }
    // ---- weight chunks (BN rows x one 128-byte [hi | lo] block) through the ring; piece = 8 rows; lane -> row lane >> 3, physical
    // 16-B slot lane & 7, logical slot = phys ^ ((row >> 1) & 7) (0-3: hi of k = 8 s .., 4-7: lo)
    const int nk = Kt / 32;
    long long wofs[B_IT];        // BYTE offset of the lane's slot in chunk 0
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int pc = j * NI + wave;                    // piece = row block
        const int n = pc * 8 + (lane >> 3);
        const int sl = (lane & 7) ^ ((n >> 1) & 7);
        wofs[j] = (pc < BPIECES && n < p.N) ? (long long)n * p.wh_ldb + sl * 16 : -1;
    }
    // weights below 2 GiB: buffer loads (32-bit lane offset, the chunk as the scalar offset, rows beyond N out of range = zeros)
    constexpr unsigned kOut = 0x80000000u;
    const bool fast32 = MT2_BUFFER_LOADS && !p.ldr64 && (long long)p.N * p.wh_ldb < 0x7fffffffll;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wh), 0, (int)kOut, 0x00020000);
    unsigned vob[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) vob[j] = wofs[j] >= 0 ? (unsigned)wofs[j] : kOut;
    auto issue = [&](int c, int st) __attribute__((always_inline)) {
        char* Bs = ring + st * STAGE_B + wave * 1024;
        if (fast32) {
#pragma unroll
            for (int j = 0; j < B_IT; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Bs + j * NI * 1024), 16, (int)vob[j], c * 128, 0, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const long long off = wofs[j] >= 0 ? wofs[j] + (long long)c * 128 : zoff_w;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wh + off),
                                             (__attribute__((address_space(3))) void*)(Bs + j * NI * 1024), 16, 0, 0);
        }
    };
    if (loader) {                  // ---- loader wave: its window pieces above, then the first ring stages
        loader_priority(p.ldr_prio);
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nk) issue(st, st);
        if (nk >= NST - 1) wait_vmcnt<(NST - 1) * B_IT>();      // the window pieces (older than the ring's) have landed
        else wait_vmcnt<0>();
    } else {
        wait_vmcnt<0>();               // this compute wave's window pieces
    }
    // ---- the window becomes fp16 planes IN PLACE, once: every 128-byte row block (32 channels of one window row) turns from 32 f32
    // into [32 hi | 32 lo] (logical 16-byte slots 0-3: hi of k = 8 s .., 4-7: lo - the layout of a weight row's block), prologue
    // activation and range guard included.  The K loop then reads BOTH operands as ready fp16 fragments: no split arithmetic, no
    // activation, no guard in it - a window element used to be fetched and split taps x WGN times (7 x 2 on a 7-tap 128-channel tile),
    // ~2 000 VALU instructions per wave and tile beside the matrix instructions; the pass costs ~120 per wave.  All waves take part
    // (read everything, barrier, write: slot j and 4 + j of a row block are other tasks' inputs).
    float amax = 0.0f;
    {
        constexpr int NT = (NW + NL) * 64;
        constexpr int MAXT = (QS * (BM + 64) * 4 + NT - 1) / NT;          // tasks per thread at the widest window (launch_gemm: span <= 64)
        const unsigned lds_w = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)win;
        const int ntask = QS * WRp * 4;                                   // (row block, j): f32 slots 2 j, 2 j + 1 -> hi slot j, lo slot 4 + j
        f32x4 in[MAXT][2];
        __builtin_amdgcn_s_barrier();                                     // every wave's window pieces have landed
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int t = tid + u * NT;
            const int tt = t < ntask ? t : 0;
            const int rbk = tt >> 2, j = tt & 3;
            const int row = rbk % WRp;
            const int swz = (row >> 1) & 7;
            const unsigned base = lds_w + (unsigned)rbk * 128;
            in[u][0] = lds_read_b128(base + (unsigned)(((2 * j) ^ swz) * 16));
            in[u][1] = lds_read_b128(base + (unsigned)(((2 * j + 1) ^ swz) * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < MAXT; ++u) asm volatile("" : "+v"(in[u][0]), "+v"(in[u][1]));
        __builtin_amdgcn_s_barrier();                                     // everything has been read
        asm volatile("" ::: "memory");
        const float slope0 = p.pro_slope;
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int t = tid + u * NT;
            if (t < ntask) {
                const int rbk = t >> 2, j = t & 3;
                const int row = rbk % WRp;
                const int swz = (row >> 1) & 7;
                u32x4 ph, pl;
                split2_f16<PRO>(in[u][0], in[u][1], slope0, ph, pl, amax);
                char* dst = reinterpret_cast<char*>(win) + (size_t)rbk * 128;
                *reinterpret_cast<u32x4*>(dst + ((j ^ swz) * 16)) = ph;
                *reinterpret_cast<u32x4*>(dst + (((4 + j) ^ swz) * 16)) = pl;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the planes are in LDS before this wave's next barrier
    }
    if (p.x3h_flag != nullptr && __builtin_amdgcn_ballot_w64(amax >= kX3hMaxIn) != 0ull && lane == 0) atomicOr(p.x3h_flag, 1);
    if (loader) {
        // cross-chunk form (the compute waves' K loop below): at barrier c the weights of chunk c-1 are still being read - the stage
        // that is free is chunk c-2's, and chunk c + NST - 2 goes there
        static_assert(NST >= 3, "cross-chunk form: chunk c-1 is still being read at barrier c");
        for (int c = 0; c < nk; ++c) {
            if (c == 0) { if (NST - 2 < nk) wait_vmcnt<(NST - 2) * B_IT>(); else wait_vmcnt<0>(); }
            else if (c + NST - 3 < nk) wait_vmcnt<(NST - 3) * B_IT>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (c >= 1 && c + NST - 2 < nk) issue(c + NST - 2, (c + NST - 2) % NST);
        }
        return;
    }
    // (No epilogue operand is requested before or during the K loop.  The one-barrier-per-chunk loop this kernel had until round 6
    // prefetched bias / residual / row masks there - 50 registers that did not fit beside the fragment pipeline: hipcc spilled them
    // one by one, each load behind its own vmcnt(0), and reloaded them one by one in the epilogue, ~10 us per tile once the pipeline
    // took the registers.  The 16-byte-store epilogue issues all its operand loads together at its start instead.)
    float inv_s[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * WTN + j * 32 + (lane & 31);
        inv_s[j] = p.wh_inv[n < p.N ? n : 0];
    }

    f32x16 acc[TM][TN], acl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; acl[i][j][e] = 0.0f; }

    const int half = lane >> 5;
    const unsigned lds_ring = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    const unsigned lds_win = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)win;
    // fragment of k block b (16 k) of a 128-byte block: hi = logical slot b * 2 + half, lo = 4 + b * 2 + half - window rows and weight rows alike
    const int nrow = wn * WTN + (lane & 31);
    const unsigned b_lane = lds_ring + nrow * 128;
    const int swzb = (nrow >> 1) & 7;
    unsigned koffb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        koffb[b][0] = (unsigned)(((b * 2 + half) ^ swzb) * 16);
        koffb[b][1] = (unsigned)(((4 + b * 2 + half) ^ swzb) * 16);
    }
    const int arow0 = wm * WTM + (lane & 31);

    // Cross-chunk fragment pipeline without a split stage: k block q = 2 c + b lives in register set b (A hi / lo of TM row tiles, B
    // hi / lo of TN column tiles).  Step q: lgkmcnt <= 2 TM + 2 TN (set b has landed, set b ^ 1 may be in flight), the 3 TM TN products
    // of the block, [q even: s_barrier - the weights of chunk c + 1 have landed - behind the products], then A(q+2) from the window and
    // B(q+2) from the ring into set b: every LDS read is in flight for a whole step.  At barrier c + 1 the weights of chunk c are still
    // being read: the free stage is chunk c - 1's (the loader's protocol above).
    constexpr int N1 = 2 * TM + 2 * TN;
    static_assert(N1 <= 15, "lgkmcnt is four bits wide");
    u32x4 ra[2][TM][2];
    u32x4 rb[2][2][TN];
    auto fetch_a = [&](int b, unsigned sa, int swza) __attribute__((always_inline)) {
        const unsigned vh = sa + (unsigned)(((b * 2 + half) ^ swza) * 16), vl = sa + (unsigned)(((4 + b * 2 + half) ^ swza) * 16);
        static_for(std::make_integer_sequence<int, TM>{}, [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ra[b][i][0] = __builtin_bit_cast(u32x4, lds_read_b128_imm<i * 32 * BK * 4>(vh));
            ra[b][i][1] = __builtin_bit_cast(u32x4, lds_read_b128_imm<i * 32 * BK * 4>(vl));
        });
    };
    auto fetch_b = [&](int b, unsigned sb) __attribute__((always_inline)) {
        const unsigned vb0 = sb + koffb[b][0], vb1 = sb + koffb[b][1];
        static_for(std::make_integer_sequence<int, TN>{}, [&](auto ic) {
            constexpr int j = decltype(ic)::value;
            rb[b][0][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<j * 32 * 128>(vb0));
            rb[b][1][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<j * 32 * 128>(vb1));
        });
    };
    auto tie_set = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(ra[b][i][0]), "+v"(ra[b][i][1]));
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rb[b][pl][j]));
    };
    // cross terms first (into the low accumulators), column tiles innermost: consecutive MFMAs never wait on each other's accumulator
    auto products = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const f16x8 Ah = __builtin_bit_cast(f16x8, ra[b][i][0]), Al = __builtin_bit_cast(f16x8, ra[b][i][1]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][1][j]), acl[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, __builtin_bit_cast(f16x8, rb[b][0][j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, __builtin_bit_cast(f16x8, rb[b][0][j]), acl[i][j], 0, 0, 0);
        }
    };
    __builtin_amdgcn_s_barrier();                         // the window's planes are written and the weights of chunk 0 have landed
    asm volatile("" ::: "memory");
    {
        const unsigned sa0 = lds_win + (unsigned)(arow0 * BK) * 4;
        const int swz0 = (arow0 >> 1) & 7;
        fetch_a(0, sa0, swz0); fetch_b(0, b_lane); fetch_a(1, sa0, swz0); fetch_b(1, b_lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    int stn = 1 % NST, tapn = 0, qn = 1;                  // chunk c + 1: its ring stage, tap and 32-channel slice
    if (qn == QS) { qn = 0; tapn = 1; }
    auto chunk = [&](auto last_c) __attribute__((always_inline)) {
        constexpr bool last = decltype(last_c)::value;
        const int arow = arow0 + tapn * dil;
        const int swza = (arow >> 1) & 7;
        const unsigned sa = lds_win + (unsigned)((qn * WRp + arow) * BK) * 4;
        const unsigned sb = b_lane + (unsigned)stn * STAGE_B;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (last && b == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N1) : "memory");
            tie_set(b);
            __builtin_amdgcn_sched_barrier(0);
            products(b);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!last) {
                if (b == 0) {
                    __builtin_amdgcn_s_barrier();         // the weights of chunk c + 1 have landed
                    asm volatile("" ::: "memory");
                }
                fetch_a(b, sa, swza);                     // A(q + 2), B(q + 2)
                fetch_b(b, sb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stn = stn + 1 == NST ? 0 : stn + 1;
        if (++qn == QS) { qn = 0; ++tapn; }
    };
    for (int c = 0; c + 1 < nk; ++c) chunk(std::false_type{});
    chunk(std::true_type{});
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_fmaf(acl[i][j][e], kX3hLoInv, acc[i][j][e]) * inv_s[j];
    if (epilogue_t4_ok(p)) epilogue_t4<TM, TN>(p, acc, 0, m0 + wm * WTM, wn * WTN, lane);
    else epilogue<TM, TN>(p, acc, 0, m0 + wm * WTM, wn * WTN, lane);
}

// ---------------------------------------------------------------------------------------------------
// host side: the kernels of this unit by tile id and prologue (the tile table lives in gemm_f32.hip)
#define MT2_X3H_LDR(BM_, BN_, WM_, WN_, NL_, NST_)                                                                          \
    { gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_NONE>, gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_RELU>, \
      gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_LRELU>, gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, PRO_APL>, nullptr, \
      gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, PRO_LNX> }
#define MT2_X3H_KS(BM_, BN_, WM_, WN_, KS_, NL_, NST_)                                                                          \
    { gemm_x3h_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, ACT_NONE>, gemm_x3h_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, ACT_RELU>, \
      gemm_x3h_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, ACT_LRELU>, gemm_x3h_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, PRO_APL>, nullptr, \
      gemm_x3h_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, PRO_LNX> }

#define MT2_X3H_WIN(QS_, BM_, BN_, WM_, WN_, NST_, NL_)                                                                         \
    { conv_win_x3h_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_NONE, NL_>, conv_win_x3h_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_RELU, NL_>, \
      conv_win_x3h_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_LRELU, NL_>, nullptr, nullptr, nullptr }

X3hKernel x3h_kernel(int tile, int variant) {
    static void (*const kTable[kX3hTiles][6])(GemmP) = {
        // X3H_LDR_128x128: 8 compute + 4 loader waves, 4 x 32 KiB (+ PRO_LNX).  Retired beside it in round 6, each measured slower on
        // every shape of the model (profiles/r06_gemm_sweep_x3hxc_v2_buffer_loads.txt): the one-barrier-per-chunk loop of the same
        // tile (3 x 32 KiB), the cross-chunk form with a 3-deep ring (one chunk-time of DMA latency), and the 64x64-per-wave forms
        // (one compute wave per SIMD + 4 loaders, 211..227 registers) in both loops
        MT2_X3H_LDR(128, 128, 4, 2, 4, 4),
        MT2_X3H_KS(32, 64, 1, 2, 4, 8, 2),          // X3H_KS_32x64_K4: the 84 tile (8 compute + 8 loader waves)
        MT2_X3H_KS(64, 64, 2, 2, 2, 8, 3),          // X3H_KS_64x64_K2: the 85 tile
        MT2_X3H_KS(32, 32, 1, 1, 8, 8, 2),          // X3H_KS_32x32_K8: the 86 tile
        MT2_X3H_WIN(2, 256, 64, 8, 1, 4, 4),        // X3H_WIN_256x64: the 58 tile (8 compute + 4 loader waves), cross-chunk form: one more stage
        MT2_X3H_WIN(4, 128, 128, 4, 2, 3, 4),       // X3H_WIN_128x128: the 59 tile
        MT2_X3H_WIN(1, 256, 32, 8, 1, 4, 4),        // X3H_WIN_256x32: the 34 tile with loader waves
        // (measured and not kept, round 6: the 32-channel window convolution - 3..19 % slower than its x6 form)
    };
    if (tile < 0 || tile >= kX3hTiles || variant < 0 || variant >= 6) return nullptr;
    return kTable[tile][variant];
}

}  // namespace mt2
