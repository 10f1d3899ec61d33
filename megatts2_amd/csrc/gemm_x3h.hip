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
// Operand path, tile map, loader waves, LDS swizzles, epilogues: gemm_x6_ldr_kernel's (gemm_f32.hip) - the weights arrive as
// 2 x BN x 64 B per chunk instead of 3 x, a stage of the 128x128 tile is 32 KiB instead of 40.
#include "gemm_common.h"

#include <atomic>

namespace mt2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kX3hLoScale = 2048.0f, kX3hLoInv = 1.0f / 2048.0f, kX3hMaxIn = 65504.0f;

// (lo, hi) = 8 consecutive f32 of one A row -> two planes of 8 fp16 (element 2i / 2i+1 in dword i).  6 VALU per element pair:
// v_cvt_pk_f16_f32, 2 v_mul_f32, 2 v_fma_mix_f32 (reads the fp16 halves in place), v_cvt_pk_f16_f32 - and one v_max3_f32 for
// the range guard; the bf16 split of x6 takes 11 per pair.
template <int PRO>
__device__ __forceinline__ void split2_f16(const f32x4& lo, const f32x4& hi, float slope, u32x4& ph, u32x4& pl, float& amax) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = apply_act<PRO>(i < 2 ? lo[2 * i] : hi[2 * i - 4], slope);
        const float y = apply_act<PRO>(i < 2 ? lo[2 * i + 1] : hi[2 * i - 3], slope);
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
// instruction inside the K loop; one s_barrier per 32-deep chunk.  PRO: prologue activation of the A values (ACT_*), or PRO_LNX:
// the pair-fed algebraic LayerNorm / row-statistics epilogue form (K loop of ACT_NONE).
template <int BM, int BN, int WGM, int WGN, int NL, int NST, int PRO>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void gemm_x3h_ldr_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 8;                            // f32 A pieces per chunk: 8 rows x 128 B
    constexpr int PB = 2 * BN / 16;                       // fp16 plane pieces per chunk: 16 rows x 64 B
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
        const unsigned short* __restrict__ Wh = reinterpret_cast<const unsigned short*>(p.Wh) + (long long)g * p.strideW;
        const long long zoff_x = (const float*)g_zero16 - X;
        const long long zoff_w = (const unsigned short*)g_zero16 - Wh;
        const long long plane = p.wh_plane;
        const int lrow = lane >> 3;
        const int ldx = p.ldx, Rx = p.Rx, Cin = p.Cin, dil = p.dil, ldw = p.ldw;
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
        long long wofs[B_IT];
        int wk[B_IT];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int pc = j * NL + lw;                      // B piece = plane * (BN / 16) + row block
            const int pl = pc / (BN / 16), rb = pc - pl * (BN / 16);
            const int nl = rb * 16 + (lane >> 2);
            const int n = n0 + nl;
            wk[j] = ((lane & 3) ^ ((nl >> 2) & 3)) * 8;
            wofs[j] = n < p.N ? pl * plane + (long long)n * ldw : -1;
        }
        wait_vmcnt<0>();                                     // the rowbase loads
        const bool fast = (Kt % BK == 0) && (!multi_tap || Cin % BK == 0);
        int s_tap = 0, s_cc = 0;
        auto issue = [&](int c, int st) {
            const int kchunk = c * BK;
            float* As = reinterpret_cast<float*>(ring + st * STAGE) + lw * 256;
            char* Bs = ring + st * STAGE + STAGE_A + lw * 1024;
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
                    const long long off = wofs[j] >= 0 ? wofs[j] + (kchunk + wk[j]) : zoff_w;
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
            for (int j = 0; j < B_IT; ++j) {
                const int k = kchunk + wk[j];
                const bool ok = (k < Kt) & (wofs[j] >= 0);
                const long long off = ok ? wofs[j] + k : zoff_w;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wh + off),
                                                 (__attribute__((address_space(3))) void*)(Bs + j * NL * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nk) issue(st, st);
        int st = 0;
        for (int c = 0; c < nk; ++c) {
            if (c + NST - 2 < nk) wait_vmcnt<(NST - 2) * L>();       // this wave's pieces of chunk c have landed
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                            // chunk c complete; chunk c-1's stage is free
            if (c + NST - 1 < nk) issue(c + NST - 1, st == 0 ? NST - 1 : st - 1);
            st = st + 1 == NST ? 0 : st + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute wave
    const int wave = wave_all;
    const int wm = wave / WGN, wn = wave % WGN;
    constexpr bool PRET = TM * TN <= 2;                   // epilogue operands in flight during the K loop
    EpiPreT<PRET ? TM : 1, PRET ? TN : 1> pret;
    if constexpr (PRET) epi_prefetch_t<TM, TN>(p, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);
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
    constexpr bool LNXOK = PRET && TM == 1 && PRO == PRO_LNX;
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
    const int swzb = (nrow >> 2) & 3;
    const unsigned b_lane = lds0 + STAGE_A + nrow * 64;
    unsigned koffa[2][2], koffb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        koffa[b][0] = (unsigned)(((b * 4 + half * 2) ^ swza) * 16);
        koffa[b][1] = (unsigned)(((b * 4 + half * 2 + 1) ^ swza) * 16);
        koffb[b] = (unsigned)(((b * 2 + half) ^ swzb) * 16);
    }

    int st = 0;
    float amax = 0.0f;
    f32x4 ra[2][TM][2];
    u32x4 rb[2][2][TN];
    u32x4 pln[2][2];
    // VALU per split: 4 pairs x (6 + 1 guard) = 28 (+ the prologue activation); 3 TN MFMAs per fragment carry one split
    constexpr int F = 2 * TM, NMF = 3 * TN, SPLIT_VALU = 28 + (PRO == ACT_RELU || PRO == ACT_LRELU ? 8 : 0);
    constexpr int VPM = (SPLIT_VALU + NMF - 1) / NMF;
    auto fetch = [&](int b, unsigned sa, unsigned sb) {
        const unsigned va0 = sa + koffa[b][0], va1 = sa + koffa[b][1], vb = sb + koffb[b];
        static_for(std::make_integer_sequence<int, TM>{}, [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ra[b][i][0] = lds_read_b128_imm<i * 32 * BK * 4>(va0);
            ra[b][i][1] = lds_read_b128_imm<i * 32 * BK * 4>(va1);
        });
        static_for(std::make_integer_sequence<int, 2 * TN>{}, [&](auto ic) {
            constexpr int pl = decltype(ic)::value / TN, j = decltype(ic)::value % TN;
            rb[b][pl][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<(pl * BN + j * 32) * 64>(vb));
        });
    };
    // the three products of fragment (b, i) with the column tiles of k-block b: cross terms first (into the low accumulator),
    // column tiles innermost so that consecutive MFMAs never wait on each other's accumulator
    auto products = [&](int b, int i, const u32x4* pp) {
        const f16x8 Ah = __builtin_bit_cast(f16x8, pp[0]), Al = __builtin_bit_cast(f16x8, pp[1]);
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
    auto wait_block = [&](int b) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) tie(b, i);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rb[b][pl][j]));
    };
    auto pattern = [&]() {
#pragma unroll
        for (int k = 0; k < NMF; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
    };
    // clock probe (two scalar reads when requested): shader cycles (s_memtime) and constant-rate ticks (s_memrealtime) across
    // the whole K loop of one wave -> the clock the matrix pipe actually sustained (bench.py)
    const bool probe = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && wave == 0;
    unsigned long long treal0 = 0, tcyc0 = 0;
    if (probe) {
        treal0 = __builtin_amdgcn_s_memrealtime();
        tcyc0 = __builtin_readcyclecounter();
        if (lane == 0) p.dbg[9] = treal0 - t_entry;
    }
    for (int c = 0; c < nk; ++c) {
        const unsigned sa = a_lane + (unsigned)st * STAGE, sb = b_lane + (unsigned)st * STAGE;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fetch(0, sa, sb);
        __builtin_amdgcn_sched_barrier(0);
        wait_block(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, sa, sb);
        split2_f16<PRO>(ra[0][0][0], ra[0][0][1], pro_slope, pln[0][0], pln[0][1], amax);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < F; ++s) {
            const int b = s / TM, i = s % TM;
            if (s + 1 < F) {
                const int b2 = (s + 1) / TM, i2 = (s + 1) % TM;
                if (b2 != b) {
                    wait_block(b2);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    tie(b2, i2);
                }
                split2_f16<PRO>(ra[b2][i2][0], ra[b2][i2][1], pro_slope, pln[(s + 1) & 1][0], pln[(s + 1) & 1][1], amax);
                products(b, i, pln[s & 1]);
                pattern();
                __builtin_amdgcn_sched_barrier(0);
            } else {
                products(b, i, pln[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        st = st + 1 == NST ? 0 : st + 1;
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
        if (p.stat_out) epilogue_pre_t<TM, TN, true>(p, acc, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);
        else epilogue_pre_t<TM, TN>(p, acc, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    } else if constexpr (PRET) epilogue_pre_t<TM, TN>(p, acc, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    else if (NW + NL <= 8 && p.epi_t4 && epilogue_t4_ok(p)) epilogue_t4<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    else epilogue<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    if (probe) {                                  // ticks spent in the epilogue (stores issued, not necessarily retired)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) p.dbg[10] = t_end - t_loop_end;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side: the kernels of this unit by tile id and prologue (the tile table lives in gemm_f32.hip)
#define MT2_X3H_LDR(BM_, BN_, WM_, WN_, NL_, NST_)                                                                          \
    { gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_NONE>, gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_RELU>, \
      gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_LRELU>, nullptr, nullptr,                                         \
      gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, PRO_LNX> }
#define MT2_X3H_LDR_PLAIN(BM_, BN_, WM_, WN_, NL_, NST_)                                                                    \
    { gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_NONE>, gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_RELU>, \
      gemm_x3h_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_LRELU>, nullptr, nullptr, nullptr }

X3hKernel x3h_kernel(int tile, int variant) {
    static void (*const kTable[kX3hTiles][6])(GemmP) = {
        MT2_X3H_LDR(128, 128, 4, 2, 4, 3),          // X3H_LDR_128x128: 8 compute + 4 loader waves, 3 x 32 KiB
        MT2_X3H_LDR(128, 128, 4, 2, 4, 4),          // X3H_LDR_128x128_S4: the same with a 4-deep ring (128 KiB)
        // 64x64 per wave needs 2 x 64 accumulator registers: more than the 168 of a 12-wave workgroup (a 256x128 tile of 8 + 4 waves
        // spills inside the K loop) - ONE compute wave per SIMD + 4 loaders = 8 waves, 211 registers, no spill
        MT2_X3H_LDR_PLAIN(128, 128, 2, 2, 4, 3),    // X3H_LDR_128x128_W4
        MT2_X3H_LDR_PLAIN(128, 128, 2, 2, 4, 4),    // X3H_LDR_128x128_W4_S4: 4-deep ring (128 KiB)
    };
    if (tile < 0 || tile >= kX3hTiles || variant < 0 || variant >= 6) return nullptr;
    return kTable[tile][variant];
}

}  // namespace mt2
