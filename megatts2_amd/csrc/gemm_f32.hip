// f32 implicit-GEMM engine for gfx950 (CDNA4): every Linear / Conv1d / ConvTranspose1d of the
// Mega-TTS 2 synthesis path (A2-A9, A13, A15-A17, A19) runs through this kernel.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bit-exact k-ordered fma chain) - the
// reference computes in fp32 and its discrete decisions (VQ argmin, PLM argmax, ADM rounding) must
// be reproduced bit-for-bit, so reduced-precision MFMA is not used.  Roofline: 157.3 TFLOP/s.
//
// Tiling (wave64): a workgroup of WGM x WGN (x KS) waves owns a BM x BN output tile; each wave owns
// (BM/WGM) x (BN/WGN) as TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each).  K is walked in
// chunks of 32: both operands are K-contiguous in HBM (activations [rows, C] time-major, weights
// [N, taps*Cin]).  The MFMA k index is a free permutation (lanes 0-31 take k = 8j+e, lanes 32-63 take
// k = 8j+4+e), so one ds_read_b128 per lane feeds four MFMAs.
//
// Two kernels share the epilogue and the tile map:
//   gemm_f32_kernel      (v1, tile configurations 0-7) stages operands through registers into a padded,
//                        double-buffered LDS tile.  Kept as the simple reference implementation for A/B
//                        runs and kernel tests; never chosen by choose_cfg.
//   gemm_f32_dma_kernel  (v2, configurations 8+, what the model runs) moves operands global -> LDS with
//                        LDS-DMA into a swizzled ring, optionally splits K over wave groups of the
//                        workgroup, and prefetches the epilogue operands (description at the kernel).
//
// Conv1d is the same GEMM with a virtual A: A[m, tap*Cin + c] = X[src(m) + tap*dil, c]; gap rows
// between utterances supply the zero padding (mt2_kernels.h).  Epilogue: bias, activation, scale,
// residual add, gap-row mask, all fused.
#include "gemm_common.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <utility>
#include <vector>

namespace mt2 {

// (typedefs, activations, epilogues, LDS / wait primitives: gemm_common.h)
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
    // an XCD's contiguous run of tiles walks the LARGER operand's tiles slowest, so that each of the 8 L2s
    // streams only its slice of it and re-reads the smaller one (AR steps: M rows < N weight rows -> n-major)
    const bool nmajor = p.M < p.N;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;

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

    epilogue<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
}

// ===================================================================================================
// v2: LDS-DMA pipelined variant.  Same math, same epilogue; the operand path differs:
//   * global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass), NST-deep ring,
//     NST-1 chunks in flight per workgroup - the v1 kernel exposed one full L2/HBM latency per chunk
//     whenever fewer than ~4 workgroups shared a CU (every AR-step GEMM, measured 1-60 TF/s);
//   * the DMA writes LDS linearly (wave-uniform base + lane*16 B: one instruction = 8 rows x 128 B), so
//     padding is impossible; bank conflicts of the MFMA operand fetch are removed with an XOR swizzle of
//     the 16-B slot index, slot' = slot ^ ((row >> 1) & 7), applied on the SOURCE address of the DMA and on
//     the ds_read_b128 address (the same involution on both sides);
//   * zero fill (conv halo rows beyond the buffer, gap sentinel rows, K tail) comes from pointing the
//     lane's source at a 16-byte zero constant; the input activation (ReLU / leaky ReLU prologue) moves
//     to the fragment registers after the ds_read;
//   * one raw s_barrier per chunk, preceded by a COUNTED s_waitcnt vmcnt((NST-2)*L) so younger chunks stay
//     in flight across the barrier (hipcc's __syncthreads would drain them with vmcnt(0)).

// KS > 1: in-workgroup K split.  The workgroup holds KS groups of WGM x WGN waves; group kg walks the chunks
// kg, kg+KS, kg+2KS, ... through its OWN ring and all groups meet at the one barrier per round.  This puts
// KS waves on every SIMD even when the launch has fewer tiles than the chip has CUs (every GEMM of the
// autoregressive steps): one wave's barrier / DMA-issue / ds_read bubble is covered by its neighbours' MFMAs
// and the serial MFMA chain of a tile shrinks KS-fold.  The partial tiles are summed through LDS in a FIXED
// order (deterministic) and each group writes 16/KS of the accumulator elements in the fused epilogue.
template <int BM, int BN, int WGM, int WGN, int KS, int NST, int PRO>
__global__ __launch_bounds__(WGM* WGN * KS * 64) void gemm_f32_dma_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;                               // waves of one K group
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW);   // 1-KiB DMA pieces per wave per chunk
    constexpr int L = A_IT + B_IT;
    constexpr int STAGE = (BM + BN) * BK;                       // floats per ring stage (of one K group)
    static_assert(PRO == ACT_NONE || PRO == ACT_RELU || PRO == ACT_LRELU, "prologue activations only (the LayerNorm-prologue forms "
                  "PRO_LN / PRO_LNA of rounds 1-2 were retired in round 6: profiles/r06_retired_kernel_forms_and_options.patch)");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "piece/wave mismatch");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && NST >= 2 && (NST - 2) * L < 64, "config");
    static_assert(KS == 1 || (TM == 1 && TN == 1 && 16 % KS == 0), "K split: one 32x32 tile per wave");
    static_assert(KS * NW <= 16, "at most 16 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave_all / NW, wave = wave_all % NW;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.z;

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nt >> 3, r = nt & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // an XCD's contiguous run of tiles walks the LARGER operand's tiles slowest, so that each of the 8 L2s
    // streams only its slice of it and re-reads the smaller one (AR steps: M rows < N weight rows -> n-major)
    const bool nmajor = p.M < p.N;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;

    const float* __restrict__ X = p.X + (long long)g * p.strideX;
    const float* __restrict__ W = p.W + (long long)g * p.strideW;
    // invalid lanes (halo beyond the buffer, gap sentinel rows, M/N/K tails) read 16 zero bytes: their
    // element offset relative to X / W is redirected to the zero constant, branch-free
    const long long zoff_x = (const float*)g_zero16 - X;
    const long long zoff_w = (const float*)g_zero16 - W;

    // DMA coordinates of this lane: piece row lane>>3, physical 16-B slot lane&7, logical slot = phys ^ swz(row)
    const int lrow = lane >> 3;
    // k offset of this lane inside a chunk for piece j: row = (j*NW + wave)*8 + lrow, swz = (row >> 1) & 7
    auto kslot_of = [&](int j) { return ((lane & 7) ^ (((j * NW + wave) * 4 + (lane >> 4)) & 7)) * 4; };
    int abase[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int m = m0 + (j * NW + wave) * 8 + lrow;
        int b = kInvalidRow;
        if (m < p.M) b = p.rowbase ? p.rowbase[m] : m * p.a_mul + p.shift0;
        abase[j] = b;
    }
    long long wofs[B_IT];    // element offset of the lane's weight row, or < 0 when the row is beyond N
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int n = n0 + (j * NW + wave) * 8 + lrow;
        wofs[j] = n < p.N ? (long long)n * p.ldw : -1;
    }
    const int nk = (p.K + BK - 1) / BK;
    const int nr = (nk + KS - 1) / KS;        // rounds; group kg owns chunk r*KS + kg of round r (zeros past K)
    const int ldx = p.ldx, Rx = p.Rx, Kt = p.K, Cin = p.Cin, dil = p.dil;
    const bool multi_tap = p.taps > 1;
    // retire the rowbase loads HERE: once DMAs are in flight hipcc can only wait for an ordinary load
    // with vmcnt(0), which would drain the ring in the prologue
    wait_vmcnt<0>();
    constexpr bool PRE = TM * TN == 1;          // epilogue operands in flight during the K loop
    constexpr int EPGK = 16 / KS;
    EpiPre<PRE ? EPGK : 1> pre;
    constexpr bool PRET = !PRE && TM * TN <= 2;  // 2 tiles per wave: 49 registers; 4 tiles would spill (64x64 per wave)
    EpiPreT<PRET ? TM : 1, PRET ? TN : 1> pret;
    if constexpr (PRE) epi_prefetch<EPGK>(p, pre, g, m0 + wm * WTM, n0 + wn * WTN, lane, kg * EPGK);
    else if constexpr (PRET) epi_prefetch_t<TM, TN>(p, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);

    float* ring = smem + kg * (NST * STAGE);
    auto issue = [&](int rd, int st) {
        const int kchunk = (rd * KS + kg) * BK;
        float* As = ring + st * STAGE + wave * 256;            // + j*NW*256 floats per piece
        float* Bs = As + BM * BK;
        // with an even number of waves per K group the slot does not depend on the piece (j*NW*4 = 0 mod 8)
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const int k = kchunk + kslot_of(NW % 2 == 0 ? 0 : j);
            int tap = 0, c = k;
            if (multi_tap) { tap = k / Cin; c = k - tap * Cin; }
            const int src = abase[j] + tap * dil;
            const bool ok = (k < Kt) & ((unsigned)src < (unsigned)Rx);
            const long long off = ok ? (long long)src * ldx + c : zoff_x;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + off),
                                             (__attribute__((address_space(3))) void*)(As + j * NW * 256), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int k = kchunk + kslot_of(NW % 2 == 0 ? 0 : j);
            const bool ok = (k < Kt) & (wofs[j] >= 0);
            const long long off = ok ? wofs[j] + k : zoff_w;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + off),
                                             (__attribute__((address_space(3))) void*)(Bs + j * NW * 256), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nr) issue(st, st);

    const float pro_slope = p.pro_slope;
    // MFMA operand fetch: inline-asm ds_read_b128 (a compiler-visible LDS load would make hipcc drain the
    // DMA queue with s_waitcnt vmcnt(0) in front of it), software-pipelined one k-group ahead.
    const int swz = (lane >> 1) & 7;                    // ((row >> 1) & 7), row = lane & 31 (+ multiples of 32)
    const int half = lane >> 5;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)ring;
    const unsigned a_lane = lds0 + ((wm * WTM + (lane & 31)) * BK) * 4;
    const unsigned b_lane = lds0 + ((BM + wn * WTN + (lane & 31)) * BK) * 4;
    unsigned koff[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) koff[kk] = (unsigned)(((2 * kk + half) ^ swz) * 16);

#ifdef MT2_PHASE_TIMING
    const bool timing = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && wave_all == 0;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, treal0 = 0, tcyc0 = 0;
#define MT2_T(i_) do { if (timing) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i_] += t_ - tprev; tprev = t_; } } while (0)
    if (timing) {
        treal0 = __builtin_amdgcn_s_memrealtime();
        tcyc0 = tprev = __builtin_readcyclecounter();
    }
#else
#define MT2_T(i_) do { } while (0)
#endif
    int st = 0;
    for (int rd = 0; rd < nr; ++rd) {
        MT2_T(5);                                   // MFMA groups of the previous round
        // round rd has landed once at most NST-2 younger rounds of this wave are still in flight
        if (rd + NST - 2 < nr) wait_vmcnt<(NST - 2) * L>();
        else wait_vmcnt<0>();
        MT2_T(0);                                   // DMA wait
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MT2_T(1);                                   // barrier
        // first operand fetch of this round goes out BEFORE the address arithmetic of the next DMA issue, so
        // its LDS latency is covered by that VALU work instead of adding to it
        const unsigned sa = a_lane + (unsigned)st * (STAGE * 4), sb = b_lane + (unsigned)st * (STAGE * 4);
        f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = lds_read_b128(sa + koff[0] + i * 32 * BK * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = lds_read_b128(sb + koff[0] + j * 32 * BK * 4);
        if (rd + NST - 1 < nr) issue(rd + NST - 1, st == 0 ? NST - 1 : st - 1);
        MT2_T(2);                                   // first fragment reads issued + refill issue (drains those reads)
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            // fences on both sides: hipcc otherwise hoists this wait into the previous group's MFMAs, right behind
            // the reads it waits for
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = lds_read_b128(sa + koff[kk + 1] + i * 32 * BK * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = lds_read_b128(sb + koff[kk + 1] + j * 32 * BK * 4);
            }
            // pin the fetch of group kk+1 ABOVE the MFMAs of group kk (hipcc otherwise sinks the asm reads below
            // them and the s_waitcnt of the next group then exposes the whole LDS latency, every group)
            __builtin_amdgcn_sched_barrier(0);
            if (PRO != ACT_NONE) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[cur][i][e] = apply_act<PRO>(fa[cur][i][e], pro_slope);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][e], fb[cur][j][e], acc[i][j], 0, 0, 0);
        }
        st = st + 1 == NST ? 0 : st + 1;
    }
#ifdef MT2_PHASE_TIMING
    MT2_T(5);
    if (timing && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) p.dbg[i] = tacc[i];
        p.dbg[6] = (unsigned long long)nr;
        p.dbg[7] = __builtin_amdgcn_s_memrealtime() - treal0;
        p.dbg[8] = __builtin_readcyclecounter() - tcyc0;
    }
#endif
#undef MT2_T
    if constexpr (KS > 1) {
        // sum the KS partial tiles through LDS (ring memory is free: every DMA has been waited for and the
        // barrier below orders the last operand reads), fixed order kg = 0..KS-1; group kg finishes elements
        // e = kg*EPG .. kg*EPG+EPG-1 of every tile
        constexpr int EPG = 16 / KS;
        __syncthreads();
        float* red = smem + ((kg * NW + wave) * 16) * 64 + lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) red[e * 64] = acc[0][0][e];
        __syncthreads();
        float out[EPG];
#pragma unroll
        for (int i = 0; i < EPG; ++i) {
            const int e = kg * EPG + i;
            float v = 0.0f;
#pragma unroll
            for (int g2 = 0; g2 < KS; ++g2) v += smem[(((g2 * NW + wave) * 16) + e) * 64 + lane];
            out[i] = v;
        }
        epilogue_pre<EPG>(p, out, pre, g, m0 + wm * WTM, n0 + wn * WTN, lane, kg * EPG);
    } else {
        if constexpr (PRE) {
            float out[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) out[e] = acc[0][0][e];
            epilogue_pre<16>(p, out, pre, g, m0 + wm * WTM, n0 + wn * WTN, lane, 0);
        } else if constexpr (PRET) {
            epilogue_pre_t<TM, TN>(p, acc, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);
        } else {
            epilogue<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
        }
    }
}

// ===================================================================================================
// v3: WINDOW convolution.  Conv1d("same", k taps, dilation d) over contiguous time-major rows with Cin = 32*QS
// in {32, 64, 128} and Cout = Cin: the HiFi-GAN resblock convolutions of the narrow stages (A19; 75 % of the
// vocoder's FLOPs) and every k-tap conv stack of that width.
//
// The implicit-GEMM kernels above treat every tap as its own K chunk, i.e. they pull the SAME activation rows
// through the global->LDS DMA k times: 13-21 FLOP per ingested byte, ingest-bound at 20-60 TF/s for N = 32 / 64.
// Here a workgroup owns BM output rows x ALL output channels and loads its input WINDOW - rows
// [m0 + shift0, m0 + shift0 + BM + (k-1)*d) x Cin - into LDS ONCE (one linear DMA, XOR-swizzled like the ring
// slots); the K loop then walks (tap, 32-channel slice) chunks whose A fragments are ds_read from the window at a
// row offset of tap*d, and only the weight chunks (Cout x 32 floats each, L2-resident, shared by all workgroups)
// stream through the ring.  Ingest per output row drops from k*Cin*4 to Cin*4 bytes (+ weights): the loop is
// MFMA-bound.  Same arithmetic as the implicit GEMM: the k index order is tap-major, 32-wide chunks, identical
// fma chain per output element.
template <int QS, int BM, int BN, int WGM, int WGN, int NST, int PRO>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_win_f32_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BPIECES = BN / 8;                       // 1-KiB pieces of one weight chunk (BN rows x 32 floats)
    constexpr int B_IT = (BPIECES + NW - 1) / NW;         // per wave; waves without a real piece issue a dummy one
    constexpr int STAGE = B_IT * NW * 256;                // floats per ring stage (dummy slots included)
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && NST >= 2 && (NST - 2) * B_IT < 64 && BN == 32 * QS, "config");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int taps = p.taps, dil = p.dil;
    const int WR = BM + (taps - 1) * dil, WRp = (WR + 7) & ~7;     // window rows (padded to whole DMA pieces)
    float* ring = smem;
    float* win = smem + NST * STAGE;                                // [QS][WRp][32], slot-swizzled rows

    const int ntm = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qq = ntm >> 3, rr = ntm & 7;
    const int tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);   // contiguous run per XCD
    const int m0 = tile * BM;

    const float* __restrict__ X = p.X;
    const float* __restrict__ W = p.W;
    const long long zoff_x = (const float*)g_zero16 - X;
    const long long zoff_w = (const float*)g_zero16 - W;
    const int ldx = p.ldx, Rx = p.Rx, Kt = p.K, ldw = p.ldw;
    const int lrow = lane >> 3;

    // ---- the input window, once: piece pc = (slice q, 8 rows r8*8..+7); lane -> (row r8*8 + lrow, physical slot lane & 7)
    {
        const int ppq = WRp >> 3, pieces = QS * ppq;
        const int row_first = m0 + p.shift0;
        for (int pc = wave; pc < pieces; pc += NW) {
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
    // ---- weight chunks through the ring: chunk c = K columns [32c, 32c + 32) = (tap c / QS, slice c % QS)
    const int nk = (Kt + BK - 1) / BK;
    long long wofs[B_IT];
    int wslot[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int pc = j * NW + wave;
        const int n = pc * 8 + lrow;
        wofs[j] = (pc < BPIECES && n < p.N) ? (long long)n * ldw : -1;
        wslot[j] = ((lane & 7) ^ ((n >> 1) & 7)) * 4;
    }
    auto issue = [&](int c, int st) {
        float* Bs = ring + st * STAGE + wave * 256;
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int k = c * BK + wslot[j];
            const bool ok = (k < Kt) & (wofs[j] >= 0);
            const long long off = ok ? wofs[j] + k : zoff_w;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + off),
                                             (__attribute__((address_space(3))) void*)(Bs + j * NW * 256), 16, 0, 0);
        }
    };
    constexpr bool PRET = TM * TN <= 2;
    EpiPreT<PRET ? TM : 1, PRET ? TN : 1> pret;
    if constexpr (PRET) epi_prefetch_t<TM, TN>(p, pret, 0, m0 + wm * WTM, wn * WTN, lane);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nk) issue(st, st);

    const float pro_slope = p.pro_slope;
    const int half = lane >> 5;
    const unsigned lds_ring = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)ring;
    const unsigned lds_win = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)win;
    const unsigned b_lane = lds_ring + ((wn * WTN + (lane & 31)) * BK) * 4;
    const int swzb = (lane >> 1) & 7;
    unsigned koffb[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) koffb[kk] = (unsigned)(((2 * kk + half) ^ swzb) * 16);
    const int arow0 = wm * WTM + (lane & 31);

    int st = 0, tap = 0, q = 0;
    for (int c = 0; c < nk; ++c) {
        if (c + NST - 2 < nk) wait_vmcnt<(NST - 2) * B_IT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int arow = arow0 + tap * dil;                        // window row of this lane's output row, this tap
        const int swza = (arow >> 1) & 7;                          // (+ i*32 rows leave the swizzle unchanged)
        const unsigned sa = lds_win + (unsigned)((q * WRp + arow) * BK) * 4;
        const unsigned sb = b_lane + (unsigned)st * (STAGE * 4);
        unsigned koffa[BK / 8];
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) koffa[kk] = (unsigned)(((2 * kk + half) ^ swza) * 16);
        f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = lds_read_b128(sa + koffa[0] + i * 32 * BK * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = lds_read_b128(sb + koffb[0] + j * 32 * BK * 4);
        if (c + NST - 1 < nk) issue(c + NST - 1, st == 0 ? NST - 1 : st - 1);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = lds_read_b128(sa + koffa[kk + 1] + i * 32 * BK * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = lds_read_b128(sb + koffb[kk + 1] + j * 32 * BK * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PRO != ACT_NONE) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[cur][i][e] = apply_act<PRO>(fa[cur][i][e], pro_slope);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][e], fb[cur][j][e], acc[i][j], 0, 0, 0);
        }
        st = st + 1 == NST ? 0 : st + 1;
        if (++q == QS) { q = 0; ++tap; }
    }
    if constexpr (PRET) epilogue_pre_t<TM, TN>(p, acc, pret, 0, m0 + wm * WTM, wn * WTN, lane);
    else epilogue<TM, TN>(p, acc, 0, m0 + wm * WTM, wn * WTN, lane);
}


// ===================================================================================================
// v3b: the WINDOW convolution on the bf16 matrix pipe, f32-equivalent ("x6").
//
// The f32 MFMA runs at 1/16 of the bf16 rate.  An f32 number is EXACTLY the sum of three bf16 numbers (truncation
// split: a1 = top 16 bits of a, a2 = top 16 bits of a - a1, a3 = a - a1 - a2: 3 x 8 = 24 significant bits), a product
// of two bf16 is exact in f32, and v_mfma_f32_32x32x16_bf16 accumulates in f32.  So
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2)  +  terms below 2^-24 |a||b| (dropped: a2b3, a3b2, a3b3)
// gives the f32 product to f32 accuracy with SIX bf16 MFMAs (each 32x32x16 in 32 cycles) instead of eight f32 MFMAs
// (32x32x2, 64 cycles each) for the same 16-deep k block: 2.67x the throughput at the accuracy of an f32 GEMM (the
// result differs from an f32 fma chain only by summation order - the same class of difference as between any two
// f32 GEMM implementations; tests/test_gpu_kernels.py::test_window_conv_x6_is_f32_equivalent measures it).
//
// Operands: the WEIGHTS are split once at load into three bf16 planes [3][Cout][K] and stream through the ring as
// (3 x Cout x 32) bf16 chunks; the ACTIVATIONS stay f32 everywhere (same HBM formats as the f32 kernels): the f32
// window is loaded into LDS exactly as in conv_win_f32_kernel and each A fragment (8 consecutive k of one row) is
// split into its three planes in REGISTERS (11 VALU per 2 elements, on the vector pipe beside the matrix pipe) -
// amortised over the TN column tiles a wave owns.  Epilogue and arithmetic order over taps unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PRO>
__device__ __forceinline__ void split3_bf16(const f32x4& lo, const f32x4& hi, float slope, u32x4& p1, u32x4& p2, u32x4& p3) {
#if defined(MT2_ABLATE) && MT2_ABLATE == 3     // ablation: no split arithmetic (wrong numbers, same MFMA / LDS / DMA work)
    p1 = __builtin_bit_cast(u32x4, lo); p2 = __builtin_bit_cast(u32x4, hi); p3 = p1;
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = apply_act<PRO>(i < 2 ? lo[2 * i] : hi[2 * i - 4], slope);
        const float y = apply_act<PRO>(i < 2 ? lo[2 * i + 1] : hi[2 * i - 3], slope);
        const unsigned xb = __float_as_uint(x), yb = __float_as_uint(y);
        p1[i] = __builtin_amdgcn_perm(yb, xb, 0x07060302u);                     // (hi16(x), hi16(y)): truncation to bf16
        const float xr = x - __uint_as_float(xb & 0xffff0000u), yr = y - __uint_as_float(yb & 0xffff0000u);   // exact
        const unsigned xc = __float_as_uint(xr), yc = __float_as_uint(yr);
        p2[i] = __builtin_amdgcn_perm(yc, xc, 0x07060302u);
        const float xs = xr - __uint_as_float(xc & 0xffff0000u), ys = yr - __uint_as_float(yc & 0xffff0000u);  // <= 8 bits left
        p3[i] = __builtin_amdgcn_perm(__float_as_uint(ys), __float_as_uint(xs), 0x07060302u);
    }
}

// NL > 0: NL loader waves refill the weight ring (and help load the window); the compute waves issue no LDS-DMA in the
// K loop (see gemm_x6_ldr_kernel).
template <int QS, int BM, int BN, int WGM, int WGN, int NST, int PRO, int NL = 0>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void conv_win_x6_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BPIECES = 3 * BN / 16;                  // 1-KiB pieces of one weight chunk: 3 planes x BN rows x 64 B
    constexpr int NI = NL > 0 ? NL : NW;                  // waves that issue the ring refill
    constexpr int B_IT = (BPIECES + NI - 1) / NI;
    constexpr int STAGE_B = B_IT * NI * 1024;             // BYTES per ring stage (dummy slots included)
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && NST >= 2 && (NST - 2) * B_IT < 64 && BN == 32 * QS, "config");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = NL > 0 && wave_all >= NW;
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
    const unsigned short* __restrict__ W3 = reinterpret_cast<const unsigned short*>(p.W3);
    const long long zoff_x = (const float*)g_zero16 - X;
    const long long zoff_w = (const unsigned short*)g_zero16 - W3;
    const int ldx = p.ldx, Rx = p.Rx, Kt = p.K;
    const long long plane = p.w3_plane;                             // elements between the weight planes

    {   // ---- the f32 input window, once (as conv_win_f32_kernel)
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
    // ---- weight chunks (3 bf16 planes x BN rows x 32 k = 64-byte rows) through the ring; piece = 16 rows of one plane;
    // lane -> row lane >> 2, physical 16-B slot lane & 3, logical slot = phys ^ ((row >> 2) & 3)
    const int nk = Kt / 32;
    long long wofs[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int pc = j * NI + wave;                    // piece = plane * (BN / 16) + row block
        const int pl = pc / (BN / 16), rb = pc - pl * (BN / 16);
        const int n = rb * 16 + (lane >> 2);
        const int sl = (lane & 3) ^ ((n >> 2) & 3);
        wofs[j] = (pc < BPIECES && n < p.N) ? pl * plane + (long long)n * p.ldw + sl * 8 : -1;
    }
    auto issue = [&](int c, int st) {
        char* Bs = ring + st * STAGE_B + wave * 1024;
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const long long off = wofs[j] >= 0 ? wofs[j] + c * 32 : zoff_w;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W3 + off),
                                             (__attribute__((address_space(3))) void*)(Bs + j * NI * 1024), 16, 0, 0);
        }
    };
    const int nk_l = Kt / 32;
    if (NL > 0 && loader) {        // ---- loader wave: window pieces above, then nothing but the ring
        loader_priority(p.ldr_prio);
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nk_l) issue(st, st);
        int st = 0;
        for (int c = 0; c < nk_l; ++c) {
            if (c + NST - 2 < nk_l) wait_vmcnt<(NST - 2) * B_IT>();   // window pieces (older) and chunk c have landed
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (c + NST - 1 < nk_l) issue(c + NST - 1, st == 0 ? NST - 1 : st - 1);
            st = st + 1 == NST ? 0 : st + 1;
        }
        return;
    }
    constexpr bool PRET = TM * TN <= 2;
    EpiPreT<PRET ? TM : 1, PRET ? TN : 1> pret;
    if constexpr (PRET) epi_prefetch_t<TM, TN>(p, pret, 0, m0 + wm * WTM, wn * WTN, lane);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    if constexpr (NL == 0) {
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nk) issue(st, st);
    } else {
        wait_vmcnt<0>();               // this compute wave's window pieces, before the first barrier
    }

    const float pro_slope = p.pro_slope;
    const int half = lane >> 5;
    const unsigned lds_ring = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    const unsigned lds_win = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)win;
    // B fragment of k block b (16 k): logical slot b*2 + half of row n = wn*WTN + j*32 + (lane & 31), plane pl
    const int nrow = wn * WTN + (lane & 31);
    const unsigned b_lane = lds_ring + nrow * 64;
    const int swzb = (nrow >> 2) & 3;
    unsigned koffb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) koffb[b] = (unsigned)(((b * 2 + half) ^ swzb) * 16);
    const int arow0 = wm * WTM + (lane & 31);

    int st = 0, tap = 0, q = 0;
    for (int c = 0; c < nk; ++c) {
        if constexpr (NL == 0) {
            if (c + NST - 2 < nk) wait_vmcnt<(NST - 2) * B_IT>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int arow = arow0 + tap * dil;
        const int swza = (arow >> 1) & 7;
        const unsigned sa = lds_win + (unsigned)((q * WRp + arow) * BK) * 4;
        const unsigned sb = b_lane + (unsigned)st * STAGE_B;
        // the stage consumed in the previous iteration is free once everyone has passed the barrier: refill it first
        if (NL == 0 && c + NST - 1 < nk) issue(c + NST - 1, st == 0 ? NST - 1 : st - 1);
        f32x4 ra[2][TM][2];
        u32x4 rb[2][3][TN];
        auto fetch = [&](int b) {
            // constants of the tile shape (second 32-row block, planes, column tiles) ride in the ds_read offset field
            // (round 5: +3..7 % on the loader-wave GEMM tiles from the same change, profiles/r05_gemm_sweep_x6_imm_offsets.txt)
            const unsigned va0 = sa + (unsigned)(((b * 4 + half * 2) ^ swza) * 16), va1 = sa + (unsigned)(((b * 4 + half * 2 + 1) ^ swza) * 16);
            const unsigned vb = sb + koffb[b];
            static_for(std::make_integer_sequence<int, TM>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ra[b][i][0] = lds_read_b128_imm<i * 32 * BK * 4>(va0);
                ra[b][i][1] = lds_read_b128_imm<i * 32 * BK * 4>(va1);
            });
            static_for(std::make_integer_sequence<int, 3 * TN>{}, [&](auto ic) {
                constexpr int pl = decltype(ic)::value / TN, j = decltype(ic)::value % TN;
                rb[b][pl][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<(pl * BN + j * 32) * 64>(vb));
            });
        };
        // Fragment-level software pipeline: the three planes of fragment s+1 are split on the vector pipe WHILE the
        // matrix pipe works through the 6*TN products of fragment s (sched_group_barrier pins the interleave: hipcc
        // otherwise issues the 44 VALU of a split as one block in front of its MFMAs and the matrix pipe idles).
        constexpr int F = 2 * TM, NMF = 6 * TN, VPM = (44 + NMF - 1) / NMF;
        u32x4 pln[2][3];
        auto products = [&](int b, int i, const u32x4* pp) {
            const bf16x8 A1 = __builtin_bit_cast(bf16x8, pp[0]), A2 = __builtin_bit_cast(bf16x8, pp[1]),
                         A3 = __builtin_bit_cast(bf16x8, pp[2]);
            // smallest terms first (they meet an accumulator increment of their own size before the big one lands);
            // column tiles innermost, so that consecutive MFMAs never wait on each other's accumulator
            constexpr int PA[6] = {3, 1, 2, 2, 1, 1}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const bf16x8 At = PA[t] == 1 ? A1 : (PA[t] == 2 ? A2 : A3);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bf16x8 Bt = __builtin_bit_cast(bf16x8, rb[b][PB[t]][j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At, Bt, acc[i][j], 0, 0, 0);
                }
            }
        };
        // The LDS reads are inline asm, so the compiler sees no dependency between "s_waitcnt" and the consumers of the
        // fragments: instruction selection may linearise a split in front of the wait that makes its input valid
        // (sched_barrier only binds the machine scheduler).  Passing the registers through an empty asm after the wait
        // gives every consumer a data dependency on it.
        auto tie = [&](int b, int i) { asm volatile("" : "+v"(ra[b][i][0]), "+v"(ra[b][i][1])); };
        auto wait_block = [&](int b) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) tie(b, i);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rb[b][pl][j]));
        };
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        wait_block(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1);
        if constexpr (TM * TN == 1) {
            // one tile per wave: 44 VALU per 6 MFMAs is more than fits in the MFMA shadow (measured: interleaving costs
            // 20 %); split and multiply in turn and let the other wave of the SIMD fill the gaps
            __builtin_amdgcn_sched_barrier(0);
            split3_bf16<PRO>(ra[0][0][0], ra[0][0][1], pro_slope, pln[0][0], pln[0][1], pln[0][2]);
            products(0, 0, pln[0]);
            __builtin_amdgcn_sched_barrier(0);
            wait_block(1);
            __builtin_amdgcn_sched_barrier(0);
            split3_bf16<PRO>(ra[1][0][0], ra[1][0][1], pro_slope, pln[1][0], pln[1][1], pln[1][2]);
            products(1, 0, pln[1]);
        } else {
        split3_bf16<PRO>(ra[0][0][0], ra[0][0][1], pro_slope, pln[0][0], pln[0][1], pln[0][2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < F; ++s) {
            const int b = s / TM, i = s % TM;
            if (s + 1 < F) {
                const int b2 = (s + 1) / TM, i2 = (s + 1) % TM;
                if (b2 != b) {                   // block 1's fragments were requested a whole step ago
                    wait_block(b2);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    tie(b2, i2);                 // keeps this split inside this step's scheduling region
                }
                split3_bf16<PRO>(ra[b2][i2][0], ra[b2][i2][1], pro_slope, pln[(s + 1) & 1][0], pln[(s + 1) & 1][1], pln[(s + 1) & 1][2]);
            }
            products(b, i, pln[s & 1]);
            if (s + 1 < F) {
#pragma unroll
                for (int k = 0; k < NMF; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        st = st + 1 == NST ? 0 : st + 1;
        if (++q == QS) { q = 0; ++tap; }
    }
    if constexpr (PRET) epilogue_pre_t<TM, TN>(p, acc, pret, 0, m0 + wm * WTM, wn * WTN, lane);
    else epilogue<TM, TN>(p, acc, 0, m0 + wm * WTM, wn * WTN, lane);
}


// ===================================================================================================
// v2d: the x6 engine with LOADER WAVES.  Measured with s_memtime (profiles/r02_x6_phase_timing.txt): an LDS-DMA
// instruction costs the wave that issues it 140-230 cycles wherever it is placed - 970 of a 256x128 chunk's 5270 cycles
// when the 8 compute waves issue the refill themselves, with the matrix pipe idle meanwhile.  Here NL extra waves do
// nothing but the ring refill (all address state lives in them) and the NW compute waves never issue a vector-memory
// instruction inside the K loop: per chunk everybody meets at ONE s_barrier - a loader arrives once its pieces of the
// chunk have landed (counted vmcnt), a compute wave once it has finished the previous chunk - then the loaders refill
// the stage that barrier freed while the compute waves fetch, split and multiply.  Arithmetic, LDS layout and fragment
// pipeline are gemm_x6_dma_kernel's.
// (The mid-chunk-barrier, cross-chunk-prefetch and free-running pipeline forms of rounds 2-3 - measured equal or slower, DESIGN 4.5 -
// were retired in round 6: profiles/r06_retired_kernel_forms_and_options.patch.)
template <int BM, int BN, int WGM, int WGN, int NL, int NST, int PRO>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void gemm_x6_ldr_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int PA = BM / 8;                            // f32 A pieces per chunk: 8 rows x 128 B
    constexpr int PB = 3 * BN / 16;                       // bf16 plane pieces per chunk: 16 rows x 64 B
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
        const unsigned short* __restrict__ W3 = reinterpret_cast<const unsigned short*>(p.W3) + (long long)g * p.strideW;
        const long long zoff_x = (const float*)g_zero16 - X;
        const long long zoff_w = (const unsigned short*)g_zero16 - W3;
        const long long plane = p.w3_plane;
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
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W3 + off),
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
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W3 + off),
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
#if defined(MT2_ABLATE) && MT2_ABLATE == 2                           // ablation: no operand ingest inside the K loop
            if (c + NST - 1 < nk && c < 1) issue(c + NST - 1, st == 0 ? NST - 1 : st - 1);
#else
            if (c + NST - 1 < nk) issue(c + NST - 1, st == 0 ? NST - 1 : st - 1);
#endif
            st = st + 1 == NST ? 0 : st + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute wave
    const int wave = wave_all;
    const int wm = wave / WGN, wn = wave % WGN;
    // epilogue operands in flight during the K loop (49 registers)
    constexpr bool PRET = TM * TN <= 2;
    EpiPreT<PRET ? TM : 1, PRET ? TN : 1> pret;
    if constexpr (PRET) epi_prefetch_t<TM, TN>(p, pret, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    // pair-fed algebraic LayerNorm and row-statistics epilogue (GemmP::ln_stat / stat_out): the plain 128x128 tile only
    // (one 32-row block per wave, epilogue operands prefetched) in its PRO_LNX instantiation - the K loop of ACT_NONE, a
    // kernel of its own so that the plain launches keep their register allocation (the tile sits at its 168-VGPR cap)
    constexpr bool LNXOK = PRET && TM == 1 && PRO == PRO_LNX;
    const bool lnx = LNXOK && p.pro_act == PRO_LNX;
    // [BM][2] (mean, rstd) behind the ring: the NW compute waves share the merge - 64 NW / BM adjacent lanes per row, <= 16 / LPR
    // float4 loads each (this tile sits at its 168-register cap with the epilogue operands in flight) - and every wave reads
    // its 32 rows back in the epilogue (the K loop's barriers order the two)
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

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

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
    f32x4 ra[2][TM][2];
    u32x4 rb[2][3][TN];
    u32x4 pln[2][3];
    constexpr int F = 2 * TM, NMF = 6 * TN, VPM = (44 + NMF - 1) / NMF;
    auto fetch = [&](int b, unsigned sa, unsigned sb) {
        // three address registers per k-block; everything that is a constant of the tile shape rides in the offset field
        const unsigned va0 = sa + koffa[b][0], va1 = sa + koffa[b][1], vb = sb + koffb[b];
        static_for(std::make_integer_sequence<int, TM>{}, [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ra[b][i][0] = lds_read_b128_imm<i * 32 * BK * 4>(va0);
            ra[b][i][1] = lds_read_b128_imm<i * 32 * BK * 4>(va1);
        });
        static_for(std::make_integer_sequence<int, 3 * TN>{}, [&](auto ic) {
            constexpr int pl = decltype(ic)::value / TN, j = decltype(ic)::value % TN;
            rb[b][pl][j] = __builtin_bit_cast(u32x4, lds_read_b128_imm<(pl * BN + j * 32) * 64>(vb));
        });
    };
    // products t0 <= t < t1 of fragment (b, i) with the column tiles of k-block b
    auto products = [&](int b, int i, const u32x4* pp, int t0, int t1) {
        const bf16x8 A1 = __builtin_bit_cast(bf16x8, pp[0]), A2 = __builtin_bit_cast(bf16x8, pp[1]),
                     A3 = __builtin_bit_cast(bf16x8, pp[2]);
        constexpr int PA_[6] = {3, 1, 2, 2, 1, 1}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t < t0 || t >= t1) continue;
            const bf16x8 At = PA_[t] == 1 ? A1 : (PA_[t] == 2 ? A2 : A3);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const bf16x8 Bt = __builtin_bit_cast(bf16x8, rb[b][PB_[t]][j]);
#if defined(MT2_ABLATE) && MT2_ABLATE == 4     // ablation: fetch + split, no matrix instructions (operands kept live)
                asm volatile("" :: "v"(At), "v"(Bt));
                if (Kt < 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At, Bt, acc[i][j], 0, 0, 0);
                continue;
#endif
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At, Bt, acc[i][j], 0, 0, 0);
            }
        }
    };
    auto tie = [&](int b, int i) { asm volatile("" : "+v"(ra[b][i][0]), "+v"(ra[b][i][1])); };
    auto wait_block = [&](int b) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) tie(b, i);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(rb[b][pl][j]));
    };
    auto pattern = [&](int nmf) {
#pragma unroll
        for (int k = 0; k < NMF; ++k) {
            if (k >= nmf) break;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        }
    };
#ifdef MT2_PHASE_TIMING
    const bool timing = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && wave == 0;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0, treal0 = 0, tcyc0 = 0;
#define MT2_T(i_) do { if (timing) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i_] += t_ - tprev; tprev = t_; } } while (0)
    if (timing) {
        treal0 = __builtin_amdgcn_s_memrealtime();      // constant-rate counter (hipDeviceAttributeWallClockRate): the
        tcyc0 = tprev = __builtin_readcyclecounter();   // ratio to s_memtime is the shader clock the K loop ran at
    }
#else
#define MT2_T(i_) do { } while (0)
    // clock probe (always built, two scalar reads when requested): shader cycles (s_memtime) and constant-rate ticks
    // (s_memrealtime) across the whole K loop of one wave -> the clock the matrix pipe actually sustained (bench.py)
    const bool probe = p.dbg != nullptr && bid == (int)(gridDim.x / 2) && wave == 0;
    unsigned long long treal0 = 0, tcyc0 = 0;
    if (probe) {
        treal0 = __builtin_amdgcn_s_memrealtime();
        tcyc0 = __builtin_readcyclecounter();
        if (lane == 0) p.dbg[9] = treal0 - t_entry;        // ticks from kernel entry to the start of the K loop
    }
#endif
    for (int c = 0; c < nk; ++c) {
        const unsigned sa = a_lane + (unsigned)st * STAGE, sb = b_lane + (unsigned)st * STAGE;
        const int stn = st + 1 == NST ? 0 : st + 1;
#if defined(MT2_ABLATE) && MT2_ABLATE == 1                                   // ablation: ingest only - the compute waves just
        if (c + 1 < nk) {                                                    // keep the barrier cadence (the last chunk runs the
            __builtin_amdgcn_s_barrier();                                    // real body so that the accumulators stay live)
            st = stn;
            continue;
        }
#endif
        MT2_T(5);                               // MFMA steps of the previous chunk
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MT2_T(1);                               // barrier (loaders' landing wait included)
        fetch(0, sa, sb);
        __builtin_amdgcn_sched_barrier(0);
        wait_block(0);
        __builtin_amdgcn_sched_barrier(0);
        MT2_T(3);                               // first fragment fetch
        fetch(1, sa, sb);
        split3_bf16<PRO>(ra[0][0][0], ra[0][0][1], pro_slope, pln[0][0], pln[0][1], pln[0][2]);
        __builtin_amdgcn_sched_barrier(0);
        MT2_T(4);                               // second fetch + first split
#pragma unroll
        for (int s = 0; s < F; ++s) {
            const int b = s / TM, i = s % TM;
            const bool last = s + 1 == F;
            if (!last) {
                const int b2 = (s + 1) / TM, i2 = (s + 1) % TM;
                if (b2 != b) {
                    wait_block(b2);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    tie(b2, i2);
                }
                split3_bf16<PRO>(ra[b2][i2][0], ra[b2][i2][1], pro_slope, pln[(s + 1) & 1][0], pln[(s + 1) & 1][1], pln[(s + 1) & 1][2]);
                products(b, i, pln[s & 1], 0, 6);
                pattern(NMF);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                products(b, i, pln[s & 1], 0, 6);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        st = stn;
    }
#ifdef MT2_PHASE_TIMING
    MT2_T(5);
    if (timing && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) p.dbg[i] = tacc[i];
        p.dbg[6] = (unsigned long long)nk;
        p.dbg[7] = __builtin_amdgcn_s_memrealtime() - treal0;
        p.dbg[8] = __builtin_readcyclecounter() - tcyc0;
    }
#else
    unsigned long long t_loop_end = 0;
    if (probe) {
        t_loop_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            p.dbg[6] = (unsigned long long)nk;
            p.dbg[7] = t_loop_end - treal0;
            p.dbg[8] = __builtin_readcyclecounter() - tcyc0;
        }
    }
#endif
#undef MT2_T
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
    // the 16-byte-store epilogue only where the variant has registers to spare (<= 8 waves: 256 VGPRs):
    // in the 12-wave 256x128 tile, which sits at its 168-VGPR cap, the extra code made the allocator spill an in-flight
    // ds_read destination inside the K loop (tools/asm_audit.py; NaNs at production size) - that tile keeps `epilogue`
    else if (NW + NL <= 8 && p.epi_t4 && epilogue_t4_ok(p))
        epilogue_t4<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
    else epilogue<TM, TN>(p, acc, g, m0 + wm * WTM, n0 + wn * WTN, lane);
#ifndef MT2_PHASE_TIMING
    if (probe) {                                  // ticks spent in the epilogue (stores issued, not necessarily retired)
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) p.dbg[10] = t_end - t_loop_end;
    }
#endif
}

// ===================================================================================================
// v2h: x6 arithmetic for the K-SPLIT tiles of the autoregressive steps.  The mid-size AR launches
// (M = 16 t rows per stream group, t < ~30: 3 900 launches, 86 ms of traced kernel time per C3 step) run on
// gemm_f32_dma_kernel's K-split tiles because one tile's serial K chain is their latency - and those tiles do their
// products on the f32 MFMA (16 x 64 cycles per wave and 32-deep chunk: 65-74 % of a round, profiles/r02_x6_phase_timing.txt).
// Here the same decomposition - KS groups of WGM x WGN waves, one 32x32 tile per wave, group kg walking chunks kg, kg+KS, ...
// through its OWN ring, one barrier per round, partial tiles summed through LDS in a fixed order, each group finishing
// 16 / KS accumulator elements in the fused epilogue - carries the x6 arithmetic (12 bf16 MFMAs of 32 cycles instead of 16
// f32 MFMAs of 64) with the weights arriving as three bf16 planes, and NL loader waves own the whole refill (a K-split tile
// issues KS times the LDS-DMA instructions of a plain one: 770 cycles of a 3 150-cycle round when the compute waves do it).
// Linear layers only (taps = 1, K a multiple of 32 KS); same arithmetic order per output element as the other x6 kernels
// within a K group, groups summed kg = 0..KS-1.
template <int BM, int BN, int WGM, int WGN, int KS, int NL, int NST, int PRO>
__global__ __launch_bounds__((WGM * WGN * KS + NL) * 64) void gemm_x6_ks_kernel(GemmP p) {
    constexpr int NW = WGM * WGN;                         // waves of one K group
    constexpr int NWC = NW * KS;                          // compute waves
    constexpr int PA = BM / 8, PB = 3 * BN / 16;          // 1-KiB pieces of one group's chunk: f32 A rows, bf16 plane rows
    constexpr int PG = PA + PB, PT = PG * KS;             // pieces per round (all groups)
    constexpr int LW = PT / NL;                           // pieces per loader wave and round
    constexpr int STAGE_A = BM * BK * 4, STAGE_B = PB * 1024, STAGE = STAGE_A + STAGE_B;   // bytes, one group's stage
    constexpr int EPG = 16 / KS;
    static_assert(BM == 32 * WGM && BN == 32 * WGN, "one 32x32 tile per wave");
    static_assert(PT % NL == 0 && 16 % KS == 0 && NST >= 2 && (NST - 2) * LW < 64 && NWC + NL <= 16, "config");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
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
        const char* __restrict__ Wb = reinterpret_cast<const char*>(reinterpret_cast<const unsigned short*>(p.W3) + (long long)g * p.strideW);
        const char* zero = reinterpret_cast<const char*>(g_zero16);
        const char* base[LW];            // operand base of the piece (X or W3), nullptr-free
        long long rowb[LW];              // byte offset of the lane's row + its k slot inside a chunk, < 0: zero row
        int kmul[LW], ldsoff[LW];        // bytes per k element (4: A, 2: B); LDS byte offset of the piece inside its group's stage
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
                base[j] = Xb; kmul[j] = 4; ldsoff[j] = w * 1024;
                rowb[j] = (unsigned)src < (unsigned)p.Rx ? ((long long)src * p.ldx + kl) * 4 : -1;
            } else {                                       // B piece: plane pl, rows rb*16 .. +15, 4 lanes per 64-byte row
                const int bp = w - PA, pl = bp / (BN / 16), rb = bp - pl * (BN / 16);
                const int nl = rb * 16 + (lane >> 2), n = n0 + nl;
                const int kl = ((lane & 3) ^ ((nl >> 2) & 3)) * 8;
                base[j] = Wb; kmul[j] = 2; ldsoff[j] = STAGE_A + bp * 1024;
                rowb[j] = n < p.N ? (pl * p.w3_plane + (long long)n * p.ldw + kl) * 2 : -1;
            }
        }
        wait_vmcnt<0>();                                   // the rowbase loads
        auto issue = [&](int rd, int st) {
#pragma unroll
            for (int j = 0; j < LW; ++j) {
                const long long kb = (long long)(rd * KS + grp[j]) * BK * kmul[j];
                const char* src = rowb[j] >= 0 ? base[j] + rowb[j] + kb : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ring + (grp[j] * NST + st) * STAGE + ldsoff[j]),
                                                 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nr) issue(st, st);
        int st = 0;
        for (int rd = 0; rd < nr; ++rd) {
            if (rd + NST - 2 < nr) wait_vmcnt<(NST - 2) * LW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                  // round rd complete in LDS; round rd-1's stages are free
            if (rd + NST - 1 < nr) issue(rd + NST - 1, st == 0 ? NST - 1 : st - 1);
            st = st + 1 == NST ? 0 : st + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute wave: K group kg, tile (wm, wn)
    const int kg = wave_all / NW, wave = wave_all % NW;
    const int wm = wave / WGN, wn = wave % WGN;
    EpiPre<EPG> pre;
    epi_prefetch<EPG>(p, pre, g, m0 + wm * 32, n0 + wn * 32, lane, kg * EPG);
    // pair-fed algebraic LayerNorm and row-statistics epilogue: the PRO_LNX instantiation (K loop of ACT_NONE; its own kernel so
    // that the plain launches keep their register allocation).  Lane l holds mean / rstd of row l & 31 of the wave's 32-row
    // block, merged from the producer's pairs while the first ring stages are in flight
    float ln_mu = 0.0f, ln_rs = 0.0f;
    const bool lnx = PRO == PRO_LNX && p.pro_act == PRO_LNX;
    if (lnx) lnx_row_stats<2>(p, m0 + wm * 32 + (lane & 31), lane >> 5, 32, ln_mu, ln_rs);      // lanes l and l ^ 32 share a row
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const float pro_slope = p.pro_slope;
    const int half = lane >> 5;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring + (unsigned)(kg * NST) * STAGE;
    const int swza = (lane >> 1) & 7;
    const unsigned a_lane = lds0 + ((wm * 32 + (lane & 31)) * BK) * 4;
    const int nrow = wn * 32 + (lane & 31);
    const int swzb = (nrow >> 2) & 3;
    const unsigned b_lane = lds0 + STAGE_A + nrow * 64;
    unsigned koffa[2][2], koffb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        koffa[b][0] = (unsigned)(((b * 4 + half * 2) ^ swza) * 16);
        koffa[b][1] = (unsigned)(((b * 4 + half * 2 + 1) ^ swza) * 16);
        koffb[b] = (unsigned)(((b * 2 + half) ^ swzb) * 16);
    }
    f32x4 ra[2][2];
    u32x4 rb[2][3], pln[3];
    auto fetch = [&](int b, unsigned sa, unsigned sb) {
        ra[b][0] = lds_read_b128(sa + koffa[b][0]);
        ra[b][1] = lds_read_b128(sa + koffa[b][1]);
        const unsigned vb = sb + koffb[b];                 // the plane offset is a constant of the tile: offset field
        rb[b][0] = __builtin_bit_cast(u32x4, lds_read_b128_imm<0>(vb));
        rb[b][1] = __builtin_bit_cast(u32x4, lds_read_b128_imm<BN * 64>(vb));
        rb[b][2] = __builtin_bit_cast(u32x4, lds_read_b128_imm<2 * BN * 64>(vb));
    };
    auto wait_block = [&](int b) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(ra[b][0]), "+v"(ra[b][1]), "+v"(rb[b][0]), "+v"(rb[b][1]), "+v"(rb[b][2]));
    };
    auto products = [&](int b) {
        const bf16x8 A1 = __builtin_bit_cast(bf16x8, pln[0]), A2 = __builtin_bit_cast(bf16x8, pln[1]), A3 = __builtin_bit_cast(bf16x8, pln[2]);
        constexpr int PA_[6] = {3, 1, 2, 2, 1, 1}, PB_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const bf16x8 At = PA_[t] == 1 ? A1 : (PA_[t] == 2 ? A2 : A3);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(At, __builtin_bit_cast(bf16x8, rb[b][PB_[t]]), acc, 0, 0, 0);
        }
    };
    int st = 0;
    for (int rd = 0; rd < nr; ++rd) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned sa = a_lane + (unsigned)st * STAGE, sb = b_lane + (unsigned)st * STAGE;
        fetch(0, sa, sb);
        fetch(1, sa, sb);
        __builtin_amdgcn_sched_barrier(0);
        // one tile per wave: 44 VALU per 6 MFMAs do not fit in the MFMA shadow - split and multiply in turn, the other
        // waves of the SIMD (another K group) fill the gaps
        wait_block(0);
        wait_block(1);
        __builtin_amdgcn_sched_barrier(0);
        split3_bf16<PRO>(ra[0][0], ra[0][1], pro_slope, pln[0], pln[1], pln[2]);
        products(0);
        __builtin_amdgcn_sched_barrier(0);
        split3_bf16<PRO>(ra[1][0], ra[1][1], pro_slope, pln[0], pln[1], pln[2]);
        products(1);
        __builtin_amdgcn_sched_barrier(0);
        st = st + 1 == NST ? 0 : st + 1;
    }
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
            out[i] = v;
        }
        if (lnx) {                                        // rstd_r * (acc - mean_r * s_n); the bias operand is c
            const int n = n0 + wn * 32 + (lane & 31);
            const float s_n = n < p.N ? p.ln_g[n] : 0.0f;
#pragma unroll
            for (int i = 0; i < EPG; ++i) {
                const int e = kg * EPG + i, r = (e & 3) + 8 * (e >> 2) + 4 * half;
                out[i] = __shfl(ln_rs, r) * (out[i] - __shfl(ln_mu, r) * s_n);
            }
        }
        if (PRO == PRO_LNX && p.stat_out) epilogue_pre<EPG, PRO == PRO_LNX>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, kg * EPG);
        else epilogue_pre<EPG>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, kg * EPG);
    } else {
        float out[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) out[e] = acc[e];
        if (lnx) {
            const int n = n0 + wn * 32 + (lane & 31);
            const float s_n = n < p.N ? p.ln_g[n] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = (e & 3) + 8 * (e >> 2) + 4 * half;
                out[e] = __shfl(ln_rs, r) * (out[e] - __shfl(ln_mu, r) * s_n);
            }
        }
        if (PRO == PRO_LNX && p.stat_out) epilogue_pre<16, PRO == PRO_LNX>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, 0);
        else epilogue_pre<16>(p, out, pre, g, m0 + wm * 32, n0 + wn * 32, lane, 0);
    }
}


// ---------------------------------------------------------------------------------------------------
// host side: tile-configuration choice and launch

struct TileCfg {
    int bm, bn, threads;
    size_t lds;               // window configurations: the ring part only (the window depends on taps and dilation)
    const char* name;
    void (*fn[6])(GemmP);     // indexed by the prologue: none / relu / leaky relu / LayerNorm / algebraic LayerNorm / none + pair statistics
                              // (PRO_LNX: pair-fed algebraic LayerNorm and the row-statistics epilogue) - nullptr: no variant
    int win_qs = 0;           // > 0: window convolution for Cin = Cout = 32 * win_qs
    bool x6 = false;          // window convolution on the bf16 pipe (3-way split, 6 products): needs GemmP::W3
    int x6_ks = 0;            // > 0: x6 K-split tile (gemm_x6_ks_kernel): linear layers with K a multiple of 32 * x6_ks
    int stat_w = 0;           // > 0: the tile has the row-statistics epilogue (GemmP::stat_out: one pair per stat_w columns) and
                              // the pair-fed algebraic-LayerNorm form (pro_act == PRO_LNX)
    int x3h = -1;             // >= 0: the tile runs on the fp16 pipe (3 products, gemm_x3h.hip): needs GemmP::Wh / wh_inv; the kernels
                              // come from x3h_kernel(x3h, variant), fn[] only says which variants exist
};

// A configuration that was measured, documented (profiles/) and is no longer built: the index keeps its
// meaning in the profiles of earlier rounds, launch_gemm answers hipErrorNotSupported
#define MT2_RETIRED(NAME_) { 0, 0, 0, 0, "retired:" NAME_, { nullptr, nullptr, nullptr, nullptr, nullptr } }
#define MT2_CFG(BM_, BN_, WM_, WN_)                                                                    \
    { BM_, BN_, WM_* WN_ * 64, 2ull * (BM_ + BN_) * LS * sizeof(float), #BM_ "x" #BN_ "_" #WM_ "x" #WN_, \
      { gemm_f32_kernel<BM_, BN_, WM_, WN_>, gemm_f32_kernel<BM_, BN_, WM_, WN_>,                        \
        gemm_f32_kernel<BM_, BN_, WM_, WN_>, nullptr, nullptr } }
#define MT2_DMA(BM_, BN_, WM_, WN_, NST_)                                              \
    { BM_, BN_, WM_* WN_ * 64, (size_t)NST_ * (BM_ + BN_) * BK * sizeof(float),          \
      "dma" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_s" #NST_,                                   \
      { gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, 1, NST_, ACT_NONE>, gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, 1, NST_, ACT_RELU>, \
        gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, 1, NST_, ACT_LRELU>, nullptr, nullptr } }
#define MT2_DMAK(BM_, BN_, WM_, WN_, KS_, NST_)                                                        \
    { BM_, BN_, WM_* WN_ * KS_ * 64, (size_t)KS_ * NST_ * (BM_ + BN_) * BK * sizeof(float),              \
      "dma" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_k" #KS_ "_s" #NST_,                                        \
      { gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, KS_, NST_, ACT_NONE>, gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, KS_, NST_, ACT_RELU>, \
        gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, KS_, NST_, ACT_LRELU>, nullptr, nullptr } }
// 2-deep ring, one 32x32 tile per wave
#define MT2_DMAL(BM_, BN_, WM_, WN_, KS_)                                                               \
    { BM_, BN_, WM_* WN_ * KS_ * 64, (size_t)KS_ * 2 * (BM_ + BN_) * BK * sizeof(float),                 \
      "dma" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_k" #KS_ "_s2",                                             \
      { gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, KS_, 2, ACT_NONE>, gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, KS_, 2, ACT_RELU>, \
        gemm_f32_dma_kernel<BM_, BN_, WM_, WN_, KS_, 2, ACT_LRELU>, nullptr, nullptr } }

#define MT2_WIN(QS_, BM_, BN_, WM_, WN_, NST_)                                                               \
    { BM_, BN_, WM_* WN_ * 64, (size_t)NST_ * (((BN_ / 8 + WM_ * WN_ - 1) / (WM_ * WN_)) * WM_ * WN_ * 256) * sizeof(float), \
      "win" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_s" #NST_,                                                       \
      { conv_win_f32_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_NONE>, conv_win_f32_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_RELU>, \
        conv_win_f32_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_LRELU>, nullptr, nullptr }, QS_ }

#define MT2_GX6L(BM_, BN_, WM_, WN_, NL_, NST_)                                                                \
    { BM_, BN_, (WM_* WN_ + NL_) * 64, (size_t)NST_ * ((size_t)BM_ * BK * 4 + (size_t)(3 * BN_ / 16) * 1024),       \
      "x6ldr" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "+" #NL_ "_s" #NST_,                                                  \
      { gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_NONE>, gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_RELU>, \
        gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_LRELU>, nullptr, nullptr }, 0, true }
// ... with the PRO_LNX variant (LayerNorm statistics handed from GEMM to GEMM): the production AR tiles only
#define MT2_GX6L_S(BM_, BN_, WM_, WN_, NL_, NST_)                                                              \
    { BM_, BN_, (WM_* WN_ + NL_) * 64, (size_t)NST_ * ((size_t)BM_ * BK * 4 + (size_t)(3 * BN_ / 16) * 1024),       \
      "x6ldr" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "+" #NL_ "_s" #NST_,                                                  \
      { gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_NONE>, gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_RELU>, \
        gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, ACT_LRELU>, nullptr, nullptr,                               \
        gemm_x6_ldr_kernel<BM_, BN_, WM_, WN_, NL_, NST_, PRO_LNX> }, 0, true, 0, (BN_) / (WN_) }
#define MT2_GX6K_S(BM_, BN_, WM_, WN_, KS_, NL_, NST_)                                                         \
    { BM_, BN_, (WM_* WN_ * KS_ + NL_) * 64, (size_t)KS_ * NST_ * ((size_t)BM_ * BK * 4 + (size_t)(3 * BN_ / 16) * 1024), \
      "x6ks" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_k" #KS_ "+" #NL_ "_s" #NST_,                                         \
      { gemm_x6_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, ACT_NONE>, gemm_x6_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, ACT_RELU>, \
        gemm_x6_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, ACT_LRELU>, nullptr, nullptr,                           \
        gemm_x6_ks_kernel<BM_, BN_, WM_, WN_, KS_, NL_, NST_, PRO_LNX> }, 0, true, KS_, 32 }
#define MT2_WX6L(QS_, BM_, BN_, WM_, WN_, NST_, NL_)                                                         \
    { BM_, BN_, (WM_* WN_ + NL_) * 64, (size_t)NST_ * (((3 * BN_ / 16 + NL_ - 1) / NL_) * NL_ * 1024),                 \
      "x6winl" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "+" #NL_ "_s" #NST_,                                           \
      { conv_win_x6_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_NONE, NL_>, conv_win_x6_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_RELU, NL_>, \
        conv_win_x6_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_LRELU, NL_>, nullptr, nullptr }, QS_, true }
#define MT2_WX6(QS_, BM_, BN_, WM_, WN_, NST_)                                                               \
    { BM_, BN_, WM_* WN_ * 64, (size_t)NST_ * (((3 * BN_ / 16 + WM_ * WN_ - 1) / (WM_ * WN_)) * WM_ * WN_ * 1024),   \
      "x6win" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_s" #NST_,                                                     \
      { conv_win_x6_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_NONE>, conv_win_x6_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_RELU>, \
        conv_win_x6_kernel<QS_, BM_, BN_, WM_, WN_, NST_, ACT_LRELU>, nullptr, nullptr }, QS_, true }

// x3h loader-wave tiles (gemm_x3h.hip): stage = BM x 128 B (A, f32) + 2 x BN x 64 B (fp16 planes); HAS_LNX_: the PRO_LNX variant exists
static void x3h_variant_exists(GemmP) {}
#define MT2_X3HL(ID_, BM_, BN_, WM_, WN_, NL_, NST_, HAS_LNX_)                                                     \
    { BM_, BN_, (WM_* WN_ + NL_) * 64, (size_t)NST_ * ((size_t)BM_ * BK * 4 + (size_t)(2 * BN_ / 16) * 1024),       \
      "x3hldr" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "+" #NL_ "_s" #NST_ "xc",                                            \
      { x3h_variant_exists, x3h_variant_exists, x3h_variant_exists, x3h_variant_exists, nullptr,                    \
        (HAS_LNX_) ? x3h_variant_exists : nullptr }, 0, true, 0, (HAS_LNX_) ? (BN_) / (WN_) : 0, ID_ }

#define MT2_X3HK(ID_, BM_, BN_, WM_, WN_, KS_, NL_, NST_)                                                           \
    { BM_, BN_, (WM_* WN_ * KS_ + NL_) * 64, (size_t)KS_ * NST_ * ((size_t)BM_ * BK * 4 + (size_t)(2 * BN_ / 16) * 1024), \
      "x3hks" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "_k" #KS_ "+" #NL_ "_s" #NST_,                                        \
      { x3h_variant_exists, x3h_variant_exists, x3h_variant_exists, x3h_variant_exists, nullptr, x3h_variant_exists }, 0, true, KS_, 32, ID_ }

#define MT2_X3HW(ID_, QS_, BM_, BN_, WM_, WN_, NST_, NL_)                                                           \
    { BM_, BN_, (WM_* WN_ + NL_) * 64,                                                                               \
      (size_t)NST_ * (((2 * BN_ / 16 + ((NL_) > 0 ? (NL_) : WM_ * WN_) - 1) / ((NL_) > 0 ? (NL_) : WM_ * WN_)) * ((NL_) > 0 ? (NL_) : WM_ * WN_) * 1024), \
      "x3hwin" #BM_ "x" #BN_ "_" #WM_ "x" #WN_ "+" #NL_ "_s" #NST_,                                                  \
      { x3h_variant_exists, x3h_variant_exists, x3h_variant_exists, nullptr, nullptr, nullptr }, QS_, true, 0, 0, ID_ }

static const TileCfg kCfgs[] = {
    // v1: register-staged double buffer (kept for A/B runs and as the reference implementation)
    MT2_RETIRED("128x128_2x2"),                             // 0
    MT2_RETIRED("64x128_2x2"),                              // 1
    MT2_RETIRED("128x64_2x2"),                              // 2
    MT2_CFG(64, 64, 2, 2),     // 3
    MT2_RETIRED("32x128_1x4"),                              // 4
    MT2_RETIRED("32x64_1x2"),                               // 5
    MT2_RETIRED("128x32_4x1"),                              // 6
    MT2_RETIRED("64x32_2x1"),                               // 7
    // v2: LDS-DMA ring
    MT2_RETIRED("dma128x128_2x2_s3"),                       // 8
    MT2_RETIRED("dma128x128_2x2_s4"),                       // 9
    MT2_RETIRED("dma64x128_2x2_s4"),                        // 10
    MT2_RETIRED("dma64x64_2x2_s4"),                         // 11
    MT2_DMA(64, 64, 2, 2, 3),     // 12
    MT2_RETIRED("dma32x128_1x4_s4"),                        // 13
    MT2_RETIRED("dma32x64_1x2_s4"),                         // 14
    MT2_DMA(128, 32, 4, 1, 4),    // 15
    MT2_DMA(256, 128, 4, 2, 3),   // 16: 8 waves (2 per SIMD), 64x64 per wave, 43 FLOP per operand byte
    MT2_DMA(128, 128, 4, 2, 4),   // 17: 8 waves, 32x64 per wave
    // v2 + in-workgroup K split (one 32x32 tile per wave, KS waves per SIMD)
    MT2_DMAL(64, 64, 2, 2, 2),      // 18:  8 waves,  64 KiB LDS (2 workgroups per CU)
    MT2_RETIRED("dma64x64_2x2_k2_s4"),                      // 19:  8 waves, 128 KiB
    MT2_DMAL(64, 64, 2, 2, 4),      // 20: 16 waves, 128 KiB
    MT2_RETIRED("dma32x64_1x2_k4_s3"),                      // 21:  8 waves, 144 KiB, 32-row tiles for the first AR steps
    MT2_DMAL(32, 64, 1, 2, 4),      // 22:  8 waves,  96 KiB
    // 64-wide outputs (HiFi-GAN stage 3): a 128-wide tile would idle half of its MFMAs
    MT2_DMA(256, 64, 4, 2, 3),      // 23: 8 waves, 64x32 per wave, 120 KiB
    MT2_RETIRED("dma128x64_4x2_s4"),                        // 24: 8 waves, 32x32 per wave,  96 KiB
    MT2_RETIRED("dma128x64_4x2_s2"),                        // 25: the same with a 2-deep ring: 48 KiB -> 3 workgroups per CU
    MT2_RETIRED("dma64x64_2x2_k1_s2"),                      // 26: 4 waves, 2-deep ring: 32 KiB -> 5 workgroups per CU
    MT2_RETIRED("dma128x64_4x2_k2_s2"),                     // 27: 16 waves (2 K groups of 4x2), 96 KiB
    MT2_DMAL(32, 32, 1, 1, 8),      // 28: 8 waves = 8 K groups of one wave, 128 KiB: the shortest K chain (M*N <= 256 tiles)
    MT2_RETIRED("dma32x32_1x1_k4_s3"),                      // 29: 4 waves, 96 KiB
    // v3: window convolutions (Cin = Cout in {32, 64, 128}); ring = 3 stages of max(BN / 8, waves) KiB
    MT2_WIN(1, 256, 32, 8, 1, 3),    // 30: 8 waves, one 32x32 tile each; 24 + 40 KiB -> 2 workgroups per CU
    MT2_WIN(2, 256, 64, 8, 1, 3),    // 31: 8 waves, 32x64 each; 24 + 80 KiB
    MT2_WIN(4, 128, 128, 4, 2, 3),   // 32: 8 waves, 32x64 each; 48 + 92 KiB
    MT2_RETIRED("win128x64_4x2_s3"),                        // 33: 8 waves, 32x32 each; 24 + 46 KiB -> 2 workgroups per CU
    // v3b: window convolutions on the bf16 pipe, f32-equivalent (6 products); ring stage = 3 planes x BN x 64 B
    MT2_WX6(1, 256, 32, 8, 1, 3),    // 34: 8 waves, 32x32 each; 24 + 40 KiB
    MT2_RETIRED("x6win256x64_8x1_s3"),                      // 35: 8 waves, 32x64 each; 48 + 80 KiB
    MT2_RETIRED("x6win128x128_4x2_s2"),                     // 36: 8 waves, 32x64 each; 48 + 92 KiB
    // v2b: implicit GEMM on the bf16 pipe, f32-equivalent; stage = BM x 128 B (A, f32) + 3 x BN x 64 B (B planes)
    MT2_RETIRED("x6dma256x128_4x2_s2"),                     // 37: 8 waves, 64x64 each; 2 x 56 KiB
    MT2_RETIRED("x6dma128x128_4x2_s3"),                     // 38: 8 waves, 32x64 each; 3 x 40 KiB
    MT2_RETIRED("x6dma128x128_4x2_s2"),                     // 39: the same with a 2-deep ring: 80 KiB (LDS would admit two workgroups per CU, its 197 VGPRs one)
    MT2_RETIRED("x6dma128x256_2x4_s2"),      // 40: 8 waves, 64x64 each; 2 x 64 KiB (wide N: the AR feed-forward / QKV)
    MT2_RETIRED("x6dma256x128_8x2_s2"),      // 41: 16 waves, 32x64 each; 2 x 56 KiB (4 waves per SIMD)
    MT2_RETIRED("x6dma256x128_8x1_s2"),      // 42: 8 waves, 32x128 each: every A fragment is split ONCE per workgroup, 24 MFMAs per split
    MT2_RETIRED("x6dma128x128_4x1_s2"),      // 43: 4 waves, 32x128 each; 80 KiB -> 2 workgroups per CU
    MT2_RETIRED("x6dma128x256_4x1_s2"),      // 44: 4 waves, 32x256 each (48 MFMAs per split); 2 x 64 KiB
    // v2c: x6 with the A operand through registers (plain vector loads), weight planes through the ring
    MT2_RETIRED("x6areg256x128_8x1_s3"),     // 45: 8 waves, 32x128 each; ring 3 x 24 KiB
    MT2_RETIRED("x6areg256x128_4x2_s3"),     // 46: 8 waves, 64x64 each
    MT2_RETIRED("x6areg128x128_4x2_s3"),     // 47: 8 waves, 32x64 each; 72 KiB -> 2 workgroups per CU
    MT2_RETIRED("x6areg128x128_4x1_s3"),     // 48: 4 waves, 32x128 each
    MT2_RETIRED("x6areg64x128_2x2_s3"),      // 49: 4 waves, 32x64 each; 72 KiB -> 2 workgroups per CU: mid-size AR launches
    MT2_RETIRED("x6areg128x256_4x2_s2"),     // 50: 8 waves, 32x128 each; ring 2 x 48 KiB
    // v2d: x6 with loader waves (the compute waves issue no vector-memory instruction inside the K loop)
    MT2_GX6L(256, 128, 4, 2, 4, 2),  // 51: 8 compute + 4 loader waves
    MT2_RETIRED("x6ldr256x128_4x2+2_s2"),                   // 52: 8 + 2
    MT2_RETIRED("x6ldr128x128_4x2+4_s2"),  // 53: 8 + 4
    MT2_RETIRED("x6ldr128x128_4x2+2_s2"),  // 54: 8 + 2
    MT2_GX6L_S(128, 128, 4, 2, 4, 3),  // 55: 8 + 4, 3-deep ring (120 KiB); + the PRO_LNX variant
    MT2_RETIRED("x6ldrx128x128_4x2+4_s3"),    // 56: the same with cross-chunk prefetch of the first fragments
    MT2_RETIRED("x6ldrx128x128_4x2+2_s3"),    // 57: 8 + 2 loader waves
    // v3c: x6 window convolutions with loader waves
    MT2_WX6L(2, 256, 64, 8, 1, 3, 4),   // 58: 8 compute + 4 loader waves
    MT2_WX6L(4, 128, 128, 4, 2, 2, 4),  // 59
    MT2_RETIRED("x6winl128x128_4x2+2_s2"),  // 60: 8 + 2
    MT2_RETIRED("x6winl256x64_8x1+2_s3"),   // 61: 8 + 2
    // v2e: loader waves + de-phased compute groups
    MT2_RETIRED("x6ldrd128x128_4x2+4_s3"),       // 62
    // v2d, small tiles for launches that cannot fill the chip with 128x128 tiles (the AR steps' mid-size GEMMs): one
    // 32x32 tile per compute wave, 3-deep ring
    MT2_RETIRED("x6ldr64x128_2x4+4_s3"),                    // 63: 8 + 4 waves, 96 KiB
    MT2_RETIRED("x6ldr128x64_4x2+4_s3"),                    // 64: 8 + 4 waves, 84 KiB (less operand ingest per FLOP than 63: the A panel is the cheap one)
    MT2_RETIRED("x6ldr64x128_2x4+2_s3"),      // 65: 8 + 2 waves
    MT2_RETIRED("x6ldr128x64_4x2+2_s3"),      // 66: 8 + 2 waves
    // v2f: loader waves + mid-chunk barrier (MP): fragment fetch and split never wait with an empty matrix pipe
    MT2_RETIRED("x6ldm128x128_4x2+4_s3"),                   // 67: the 55 tile
    MT2_RETIRED("x6ldm256x128_4x2+4_s2"),                   // 68: the 51 tile
    MT2_RETIRED("x6ldm128x64_4x2+4_s3"),     // 69: the 64 tile
    MT2_RETIRED("x6ldm64x128_2x4+4_s3"),     // 70: the 63 tile
    MT2_RETIRED("x6ldm128x128_4x2+4_s2"),    // 71: 128x128 with a 2-deep ring (80 KiB)
    // ONE compute wave per SIMD (64x64 per wave) + 4 loader waves: two barrier-synchronised waves on a SIMD run one after
    // the other (the matrix pipe's arbiter serves the older wave first, profiles/r03_ubench_x6_issue_v2.txt), each at a
    // lone wave's efficiency and each with its own exposed head; one wave with twice the tile has the same MFMA count per
    // SIMD, 37 % less LDS traffic, half the splits per MFMA, and 256 registers for the MP pipeline
    MT2_RETIRED("x6ldm128x128_2x2+4_s3"),                   // 72: 4 + 4 waves, 120 KiB
    MT2_RETIRED("x6ldm128x128_2x2+2_s3"),    // 73: 4 + 2 waves
    MT2_RETIRED("x6ldr128x128_2x2+4_s3"),     // 74: the same tile without the MP pipeline (A/B)
    // v2g: loader waves + FREE-RUNNING compute waves (LDS counters instead of s_barrier in the K loop)
    MT2_RETIRED("x6ldf128x128_4x2+4_s3"),                   // 75: the 55 tile
    MT2_RETIRED("x6ldf256x128_4x2+4_s2"),    // 76: the 51 tile
    MT2_RETIRED("x6ldf128x64_4x2+4_s3"),     // 77: the 64 tile
    MT2_RETIRED("x6ldf128x128_4x2+2_s3"),    // 78: 55 with 2 loader waves
    // v2h: x6 arithmetic on the K-split tiles of the AR steps, loader waves own the refill
    MT2_RETIRED("x6ks32x64_1x2_k4+4_s2"),                   // 79: 8 compute (4 K groups of 1x2) + 4 loader waves, 128 KiB: the 22 tile
    MT2_RETIRED("x6ks64x64_2x2_k2+4_s3"),                   // 80: 8 compute (2 K groups of 2x2) + 4 loader waves, 120 KiB: the 18 / 20 tile
    MT2_RETIRED("x6ks32x64_1x2_k4+2_s2"),    // 81: 79 with 2 loader waves
    MT2_RETIRED("x6ks32x32_1x1_k8+4_s2"),                   // 82: 8 K groups of one wave + 4 loader waves, 112 KiB: the 28 tile
    MT2_RETIRED("x6ks64x64_2x2_k2+4_s2"),    // 83: 80 with a 2-deep ring (80 KiB)
    MT2_GX6K_S(32, 64, 1, 2, 4, 8, 2),  // 84 (+ PRO_LNX): 79 with EIGHT loader waves (16 waves: a round's 64 pieces are 8 per loader)
    MT2_GX6K_S(64, 64, 2, 2, 2, 8, 3),  // 85 (+ PRO_LNX): 80 with eight loader waves (5 pieces per loader and round)
    MT2_GX6K_S(32, 32, 1, 1, 8, 8, 2),  // 86 (+ PRO_LNX): 82 with eight loader waves
    // names only: the weight-streaming kernel for M <= 64 rows lives in gemm_skinny.hip (launch_gemm routes to it)
    { 32, 32, 512, 0, "skinny32_f32", { nullptr, nullptr, nullptr, nullptr, nullptr } },    // 87
    { 64, 32, 512, 0, "skinny64_f32", { nullptr, nullptr, nullptr, nullptr, nullptr } },    // 88
    { 32, 32, 512, 0, "skinnytm32_f32", { nullptr, nullptr, nullptr, nullptr, nullptr } },  // 89: the same on tile-major weights
    { 64, 32, 512, 0, "skinnytm64_f32", { nullptr, nullptr, nullptr, nullptr, nullptr } },  // 90   (+ LayerNorm prologue)
    // v4: the loader-wave tiles on the fp16 pipe, f32-equivalent THREE-product form (gemm_x3h.hip)
    MT2_RETIRED("x3hldr128x128_4x2+4_s3"),                  // 91: the 55 tile, one-barrier-per-chunk loop, 3 x 32 KiB (-> 103)
    MT2_RETIRED("x3hldr128x128_4x2+4_s4"),                  // 92: ... with a 4-deep ring (128 KiB)
    MT2_RETIRED("x3hldr128x128_2x2+4_s3"),                  // 93: one compute wave per SIMD (64x64 per wave) + 4 loaders
    MT2_RETIRED("x3hldr128x128_2x2+4_s4"),                  // 94: ... with a 4-deep ring
    // ... and the K-split tiles of the AR steps (gemm_x3h_ks_kernel; + PRO_LNX)
    MT2_X3HK(X3H_KS_32x64_K4, 32, 64, 1, 2, 4, 8, 2),             // 95: the 84 tile, 96 KiB
    MT2_X3HK(X3H_KS_64x64_K2, 64, 64, 2, 2, 2, 8, 3),             // 96: the 85 tile, 96 KiB
    MT2_X3HK(X3H_KS_32x32_K8, 32, 32, 1, 1, 8, 8, 2),             // 97: the 86 tile, 128 KiB
    // ... and the window convolutions of the vocoder's resblocks (conv_win_x3h_kernel)
    MT2_X3HW(X3H_WIN_256x32, 1, 256, 32, 8, 1, 4, 4),             // 98: the 34 tile (32 channels)
    MT2_X3HW(X3H_WIN_256x64, 2, 256, 64, 8, 1, 4, 4),             // 99: the 58 tile
    MT2_X3HW(X3H_WIN_128x128, 4, 128, 128, 4, 2, 3, 4),           // 100: the 59 tile
    // ... the loader tiles with ONE barrier per 64-deep super-chunk (4 stages = 2 super-stages, 128 KiB): +1 % isolated, +2.4 % SLOWER
    // in the model (profiles/r06_experiment_x3h_superchunk.patch)
    MT2_RETIRED("x3hldr128x128_4x2+4_s4c2"),                // 101: the 91 tile
    MT2_RETIRED("x3hldr128x128_2x2+4_s4c2"),                // 102: the 94 tile
    // ... the loader tile with the fragment pipeline running across the chunk boundary (gemm_x3h_ldr_kernel; "xc" in the name)
    MT2_X3HL(X3H_LDR_128x128, 128, 128, 4, 2, 4, 4, true),  // 103: the 55 tile (+ PRO_LNX), 4 x 32 KiB: THE x3h loader tile
    MT2_RETIRED("x3hldr128x128_4x2+4_s3xc"),                // 104: ... 3 x 32 KiB (one chunk-time of DMA latency: slower)
    MT2_RETIRED("x3hldr128x128_2x2+4_s4xc"),                // 105: the 94 tile in this form (slower than 103 on every shape)
};
constexpr int kSkinny32 = 87, kSkinny64 = 88, kSkinnyTm32 = 89, kSkinnyTm64 = 90;
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

int gemm_trace_shapes(EngineOpts& o, char* buf, int cap, int top) {
    struct Agg { int cfg, M, N, K, g; long long n; double ms, fl; };
    std::vector<Agg> v;
    for (auto& r : o.trace) {
        float dt = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&dt, r.e0, r.e1) != hipSuccess) return -1;
        bool found = false;
        for (auto& a : v)
            if (a.cfg == r.cfg && a.M == r.M && a.N == r.N && a.K == r.K && a.g == r.groups) {
                ++a.n; a.ms += dt; a.fl += r.flops; found = true;
                break;
            }
        if (!found) v.push_back({r.cfg, r.M, r.N, r.K, r.groups, 1, (double)dt, r.flops});
    }
    std::sort(v.begin(), v.end(), [](const Agg& a, const Agg& b) { return a.ms > b.ms; });
    int off = 0;
    for (int i = 0; i < (int)v.size() && i < top; ++i) {
        const Agg& a = v[i];
        const int w = snprintf(buf + off, cap - off, "%s %d %d %d %d %lld %.3f %.2f\n", kCfgs[a.cfg].name, a.M, a.N, a.K, a.g,
                               a.n, a.ms, a.fl / (a.ms > 0 ? a.ms : 1e-9) / 1e9);
        if (w < 0 || w >= cap - off) break;
        off += w;
    }
    return off;
}

int gemm_num_configs() { return kNumCfgs; }
const char* gemm_config_name(int idx) { return idx >= 0 && idx < kNumCfgs ? kCfgs[idx].name : ""; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per-DEVICE state of the code object: a "done" mask per (configuration,
// prologue) with one bit per device only saves the call; the benign race (two threads setting the same value) is harmless.
static std::atomic<unsigned long long> g_attr_done[kNumCfgs][6];   // bit per device (dyn_lds_once)

// ---- launch trace (measurement only): HIP events around every GEMM launch, on the launch stream; the records
// live in the EngineOpts of whoever asked for the trace (the model handle).
// Per tile configuration: launches, executed FLOPs (2*M*N*K*groups) and summed kernel time (ms).  When the
// launches ran on several streams (AR stream groups) their intervals overlap; a last pseudo-entry named
// "union" carries the length of the UNION of all launch intervals (= time during which at least one engine
// kernel was running), the right denominator for a whole-engine throughput.
int gemm_trace_collect(EngineOpts& o, int cap, const char** names, int64_t* launches, double* flops, double* ms) {
    o.trace_on = false;
    auto& tr = o.trace;
    int n = 0;
    std::vector<std::pair<double, double>> iv;
    double fl_all = 0.0;
    bool ok = true;
    for (int i = 0; i < kNumCfgs && n < cap && ok; ++i) {
        int64_t cnt = 0;
        double fl = 0.0, t = 0.0;
        for (auto& r : tr) {
            if (r.cfg != i) continue;
            float dt = 0.f, t0 = 0.f;
            if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&dt, r.e0, r.e1) != hipSuccess ||
                hipEventElapsedTime(&t0, tr.front().e0, r.e0) != hipSuccess) { ok = false; break; }
            iv.emplace_back((double)t0, (double)t0 + dt);
            ++cnt; fl += r.flops; t += dt;
        }
        if (cnt == 0) continue;
        names[n] = kCfgs[i].name; launches[n] = cnt; flops[n] = fl; ms[n] = t;
        fl_all += fl;
        ++n;
    }
    if (ok && n < cap && !iv.empty()) {
        std::sort(iv.begin(), iv.end());
        double uni = 0.0, lo = iv[0].first, hi = iv[0].second;
        for (auto& x : iv) {
            if (x.first > hi) { uni += hi - lo; lo = x.first; hi = x.second; }
            else if (x.second > hi) hi = x.second;
        }
        uni += hi - lo;
        names[n] = "union"; launches[n] = (int64_t)iv.size(); flops[n] = fl_all; ms[n] = uni;
        ++n;
    }
    for (auto& r : tr) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    tr.clear();
    return ok ? n : -1;
}

// Tile choice, from tools/gemm_sweep.py on MI355X (profiles/r01_gemm_sweep_v3.txt).  Two regimes:
//   * operand ingest: a CU sustains ~20 GB/s of global->LDS DMA however many workgroups it hosts, so a
//     tile costs about (bm + bn) * K * 4 B / 20 GB/s; 64x64 tiles (16 FLOP per operand byte) are ingest-bound
//     at ~85 TFLOP/s but fill the chip earliest - they win for every GEMM of the autoregressive steps
//     (M <= ~2200 rows);
//   * matrix issue: big tiles (32-43 FLOP/B) need >= 2 waves per SIMD to keep the MFMA pipe busy across the
//     per-chunk barrier: the 8-wave 256x128 / 128x128 tiles reach 95-106 TFLOP/s once there are enough of
//     them to load every CU (conv stacks, vocoder).
//   * chain / fill (AR steps): a launch with fewer 64x64 tiles than the chip has CUs leaves one workgroup per
//     CU at best, and a 4-wave tile then runs at the pace of ONE wave per SIMD: serial chain K/2 x 64 cycles
//     plus an exposed barrier + DMA-issue + ds_read bubble per chunk (measured ~1 us per 32-wide chunk vs
//     0.43 us of MFMA).  The K-split tiles put 2-4 waves on each SIMD of the same CU instead.
// (the thresholds, in tiles, are EngineOpts::t_ks4 / t_ks2 / t32 / t32x32)

// A CU retires one 64x64 tile of K=768 in ~12 us whatever the launch looks like, so for the AR-step shapes the
// choice is about how many CUs get a tile and how many tiles the busiest CU gets (profiles/r01_gemm_sweep_ar_*):
//   32x64 K-split tiles while they fit one per CU; 64x64 K-split tiles while THEY fit one (k4) / two (k2) per CU;
//   a big 8-wave tile when its tile count just fills the chip once (200..256); otherwise plain 64x64 tiles
//   (three workgroups per CU, de-phased) and the 8-wave tiles for the conv stacks and the vocoder.
// window convolution: plain "same" conv over contiguous rows, square, narrow (see conv_win_f32_kernel)
static bool win_eligible(const GemmP& p) {
    return p.taps >= 2 && !p.rowbase && p.a_mul == 1 && p.groups == 1 && p.N == p.Cin &&
           (p.Cin == 32 || p.Cin == 64 || p.Cin == 128) && (p.taps - 1) * p.dil <= 64 && p.pro_act < PRO_LN;
}
static const TileCfg* choose_cfg(const GemmP& p, const EngineOpts& o, int* idx_out) {
    int bi = 12;                                                        // dma64x64_2x2_s3
    const bool x3h_ok = p.Wh && p.wh_inv;
    if (o.win_conv && win_eligible(p) && !(o.force_cfg >= 0 && o.force_cfg < kNumCfgs)) {
        bi = p.Cin == 32 ? 30 : (p.Cin == 64 ? 31 : 32);
        if (o.x6_conv && p.W3) {
            // the bf16-pipe forms: 34 (32 channels), and with loader waves 58 / 59 (64 / 128 channels: +5..14 % / +2..6 % over the
            // self-refilling forms 35 / 36, retired in round 6)
            bi = bi == 30 ? 34 : (bi == 31 ? 58 : 59);
            // the fp16-pipe forms of the 64- and 128-channel tiles (profiles/r06_gemm_sweep_x3hwin_v1.txt: +19..37 % and +35..40 %, and
            // +11..18 % more with the cross-chunk pipeline, _v3_cross_chunk.txt, +10..35 % more with the window converted to fp16
            // planes once per tile, _v4_planes_in_lds.txt).  The 32-channel convolutions too since then: 94 vs 72 TF/s with 3 taps,
            // 221 vs 123 with 11 (the first x3h build of that tile was 3..19 % SLOWER than x6)
            if ((o.x3h & 4) && x3h_ok) bi = bi == 58 ? 99 : (bi == 59 ? 100 : (bi == 34 ? 98 : bi));
        }
        *idx_out = bi;
        return &kCfgs[bi];
    }
    const long long t32 = (long long)((p.M + 31) / 32) * ((p.N + 63) / 64) * p.groups;
    const long long t32x32 = (long long)((p.M + 31) / 32) * ((p.N + 31) / 32) * p.groups;
    const long long t64 = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * p.groups;
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.groups;
    const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 127) / 128) * p.groups;
    if (p.N <= 32) bi = 15;                                             // dma128x32_4x1_s4
    else if (p.N <= 64 && t128 >= 400) bi = 23;                         // dma256x64_4x2_s3
    else if (p.N <= 256 && t128 >= 400) bi = 17;                        // two n-tiles: 128x128 beats 256x128 (vocoder)
    else if (t256 >= 400 || (t256 >= 200 && t256 <= 256)) bi = 16;      // dma256x128_4x2_s3
    else if (t128 >= 400 || (t128 >= 200 && t128 <= 256)) bi = 17;      // dma128x128_4x2_s4
    else if (t32x32 <= o.t32x32 && p.K >= 512) bi = 28;                 // dma32x32_1x1_k8_s2
    else if (t32 <= o.t32) bi = 22;                                     // dma32x64_1x2_k4_s2
    else if (t64 <= o.t_ks4) bi = 20;                                   // dma64x64_2x2_k4_s2
    else if (t64 <= o.t_ks2) bi = 18;                                   // dma64x64_2x2_k2_s2
    // Launches whose weights come with planes leave the f32 MFMA for the 16-bit matrix pipe in an f32-equivalent form: the
    // loader-wave tiles (256x128 from t_x6_256 tiles on, 128x128 from t_x6_128 / t_x3h_128 on - conv stacks, vocoder stage 1, the
    // PLM / ADM QKV and ff.0 at full batch) and, below that, the K-split tiles of the AR steps (out-projection, ff.3, early steps).
    if (o.x6_gemm && p.W3 && (p.K & 7) == 0 && (p.ldw & 7) == 0 && (p.pro_act < PRO_LN || p.pro_act == PRO_LNX) && p.N > 64) {
        const bool h1 = (o.x3h & 1) && x3h_ok;      // the 128x128 tile will run in its x3h form: its own crossover against the K-split tiles
        if (t256 >= o.t_x6_256) bi = 51;
        else if (t128 >= (h1 ? o.t_x3h_128 : o.t_x6_128)) bi = 55;
        // K-split tiles on the bf16 pipe, eight loader waves (x6_ks: 0 off; 1, 3: the 32x64 k4 and 64x64 k2 tiles; 2, 4: + the 32x32
        // k8 tile; 5: the 64x64 tile only)
        if (o.x6_ks && p.taps == 1) {
            if (bi == 22 && p.K % (BK * 4) == 0 && o.x6_ks != 5) bi = 84;
            else if ((bi == 20 || bi == 18) && p.K % (BK * 2) == 0) bi = 85;
            else if ((o.x6_ks == 2 || o.x6_ks == 4) && bi == 28 && p.K % (BK * 8) == 0) bi = 86;
        }
    }
    // the fp16-pipe form of the tile (three products instead of six) where one exists and the weights come with fp16 planes
    // (profiles/r06_gemm_sweep_x3h_v1_gate.txt: the 128x128 x3h tile beats BOTH x6 loader tiles on every shape of the model - 199 vs
    // 146 TF/s at 864x4096x1024, 245 vs 188 at 4096^3 on the first build; 243 and 297 with the cross-chunk pipeline and the loaders
    // on buffer loads, profiles/r06_gemm_sweep_x3hxc_v2_buffer_loads.txt)
    if ((o.x3h & 1) && x3h_ok && (bi == 55 || bi == 51)) bi = 103;
    // K-split tiles 84 / 85 / 86 -> 95 / 96 / 97 (profiles/r06_gemm_sweep_x3hk_v1.txt: +13..20 %, +25..50 %, +20..30 % per launch)
    if ((o.x3h & 2) && x3h_ok && bi >= 84 && bi <= 86) bi += 11;
    if (o.force_cfg >= 0 && o.force_cfg < kNumCfgs) bi = o.force_cfg;
    *idx_out = bi;
    return &kCfgs[bi];
}

bool gemm_takes_planes(const GemmP& p_in, const EngineOpts& o) {
    GemmP p = p_in;
    if (p.taps <= 0) p.taps = 1;
    if (p.groups <= 0) p.groups = 1;
    if (p.a_mul == 0) p.a_mul = 1;
    p.K = p.taps * p.Cin;
    if (p.ldw == 0) p.ldw = p.K;
    if (!(o.x3h & 3) || o.force_cfg >= 0 || p.M <= 64 || p.pro_act != ACT_NONE || p.stat_out || (p.Cin % BK) != 0 || (p.ldx % BK) != 0 ||
        (p.groups > 1 && (p.strideX % BK) != 0) || p.a_mul != 1 || p.rowbase || !p.Wh || !p.wh_inv || (((unsigned long long)p.X) & 127))
        return false;
    int idx = -1;
    const TileCfg* c = choose_cfg(p, o, &idx);
    return c && c->x3h >= 0 && !c->win_qs && c->fn[PRO_APL] != nullptr;
}

// C as fp16 planes (GemmP::c_planes): the x3h loader tile's 16-byte-store epilogue, whole 128-byte blocks per row, one group
static bool c_planes_ok(const GemmP& p, const TileCfg* c) {
    return c && c->x3h >= 0 && !c->x6_ks && !c->win_qs && p.groups == 1 && !p.R && !p.stat_out && (p.N & 31) == 0 && (p.ldc & 31) == 0 &&
           (((unsigned long long)p.C) & 127) == 0 && (((unsigned long long)p.bias) & 15) == 0 && ((p.strideC | p.strideB) & 3) == 0 && p.Wh && p.wh_inv && p.M > 64;
}
bool gemm_writes_planes(const GemmP& p_in, const EngineOpts& o) {
    GemmP p = p_in;
    if (p.taps <= 0) p.taps = 1;
    if (p.groups <= 0) p.groups = 1;
    if (p.a_mul == 0) p.a_mul = 1;
    p.K = p.taps * p.Cin;
    if (p.ldw == 0) p.ldw = p.K;
    if (!(o.x3h & 1) || o.force_cfg >= 0 || !o.epi_t4 || p.M <= 64) return false;
    int idx = -1;
    const TileCfg* c = choose_cfg(p, o, &idx);
    return c_planes_ok(p, c);
}

hipError_t launch_gemm(const GemmP& p_in, hipStream_t s, EngineOpts* opts) {
    static const EngineOpts kDefaults;
    const EngineOpts& o = opts ? *opts : kDefaults;
    GemmP p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.groups <= 0) return hipSuccess;
    if ((p.Cin & 3) || (p.ldx & 3) || (p.ldw & 3) || p.K != p.taps * p.Cin) return hipErrorInvalidValue;
    if (p.pro_act < 0 || p.pro_act > PRO_LNX) return hipErrorInvalidValue;
    if (opts) opts->last_stat_nt = opts->last_stat_w = 0;
    if (p.c_planes && p.M <= 64) return hipErrorNotSupported;      // (the <= 64-row kernels write f32)
    if (p.pro_act == PRO_LNX && (p.taps != 1 || p.groups != 1 || !p.ln_g || !p.ln_stat || p.ln_nt < 2 || p.ln_nt > 32 ||
                                 (p.ln_nt & 1) || p.ln_w <= 0 || p.ln_nt * p.ln_w != p.K || (((unsigned long long)p.ln_stat) & 15)))
        return hipErrorInvalidValue;
    // a handful of rows: the weight-streaming kernel (gemm_skinny.hip) instead of a tile configuration
    const bool sk_forced = o.force_cfg == kSkinny32 || o.force_cfg == kSkinny64;
    const bool tm_forced = o.force_cfg == kSkinnyTm32 || o.force_cfg == kSkinnyTm64;
    if (tm_forced || (!p.a_planes && o.skinny_tm && o.skinny_rows > 0 && o.force_cfg < 0 && p.groups >= o.skinny_groups &&
                      gemm_skinny_tm_eligible(p, o.skinny_rows))) {
        if (!gemm_skinny_tm_eligible(p, 64)) return hipErrorInvalidValue;
        p.sk_nw = o.skinny_nw;
        if (p.stat_out) {       // row statistics as pairs per 16-column block (gemm_skinny_tm_kernel's epilogue): N / 16 <= 64 pairs per row
            const int nt = p.N / 16;
            if (p.groups == 1 && (nt & 1) == 0 && nt <= 64 && ((((unsigned long long)p.stat_out) & 15) == 0)) { p.stat_nt = nt; p.stat_w = 16; }
            else { p.stat_out = nullptr; p.stat_nt = p.stat_w = 0; }
        }
        if (opts) { opts->last_stat_nt = p.stat_nt; opts->last_stat_w = p.stat_w; }
        const int sidx = p.M <= 32 ? kSkinnyTm32 : kSkinnyTm64;
        if (opts) opts->last_cfg = kCfgs[sidx].name;
        if (opts && opts->trace_on) {
            TraceRec r;
            r.cfg = sidx;
            r.flops = 2.0 * p.M * p.N * p.K * p.groups;
            r.M = p.M; r.N = p.N; r.K = p.K; r.groups = p.groups;
            if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return hipErrorUnknown;
            (void)hipEventRecord(r.e0, s);
            const hipError_t e = launch_gemm_skinny_tm(p, s);
            (void)hipEventRecord(r.e1, s);
            opts->trace.push_back(r);
            return e;
        }
        return launch_gemm_skinny_tm(p, s);
    }
    if (sk_forced || (o.skinny_rows > 0 && o.force_cfg < 0 && p.groups >= o.skinny_groups && gemm_skinny_eligible(p, o.skinny_rows))) {
        if (!gemm_skinny_eligible(p, 64)) return hipErrorInvalidValue;
        const int sidx = p.M <= 32 ? kSkinny32 : kSkinny64;
        if (opts) opts->last_cfg = kCfgs[sidx].name;
        if (opts && opts->trace_on) {
            TraceRec r;
            r.cfg = sidx;
            r.flops = 2.0 * p.M * p.N * p.K * p.groups;
            r.M = p.M; r.N = p.N; r.K = p.K; r.groups = p.groups;
            if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return hipErrorUnknown;
            (void)hipEventRecord(r.e0, s);
            const hipError_t e = launch_gemm_skinny(p, s);
            (void)hipEventRecord(r.e1, s);
            opts->trace.push_back(r);
            return e;
        }
        return launch_gemm_skinny(p, s);
    }
    int idx = 0;
    const TileCfg* c = choose_cfg(p, o, &idx);
    if (p.pro_act == PRO_LNX && !(c->stat_w && c->fn[PRO_LNX])) return hipErrorNotSupported;      // callers fall back to LayerNorm + GEMM
    // the x3h loader tile writes through the 16-byte-store epilogue only in its pair-statistics variant: columns in fours, 16-byte bases
    if (c->x3h >= 0 && !c->x6_ks && !c->win_qs && (p.pro_act == PRO_LNX || p.stat_out)) {
        const bool t4 = ((p.N | p.ldc | (p.R ? p.ldr : 0)) & 3) == 0 && ((p.strideC | p.strideR | p.strideB) & 3) == 0 &&
                        (((unsigned long long)p.C | (unsigned long long)p.R | (unsigned long long)p.bias) & 15) == 0;
        if (!t4) {
            if (p.pro_act == PRO_LNX) return hipErrorNotSupported;
            p.stat_out = nullptr;
        }
    }
    if (p.stat_out) {       // row-statistics epilogue where the chosen tile has one; otherwise the launch simply writes none
        const int nt = c->stat_w ? p.N / c->stat_w : 0;
        if (c->stat_w && c->fn[PRO_LNX] && (p.pro_act == ACT_NONE || p.pro_act == PRO_LNX) && p.groups == 1 &&
            p.N % c->stat_w == 0 && nt >= 2 && nt <= 32 && (nt & 1) == 0 && ((((unsigned long long)p.stat_out) & 15) == 0)) {
            p.stat_w = c->stat_w; p.stat_nt = nt;
        } else {
            p.stat_out = nullptr; p.stat_w = p.stat_nt = 0;
        }
    }
    // variant index: pair statistics (consumer and / or producer side) run the PRO_LNX instantiation - the K loop of ACT_NONE
    // ... an A operand that arrives as fp16 planes (a_planes) runs the PRO_APL instantiation: x3h loader / K-split tiles only
    if (p.a_planes && (c->x3h < 0 || c->win_qs || p.pro_act != ACT_NONE || p.stat_out || (p.Cin % BK) != 0 || (p.ldx % BK) != 0 ||
                       (p.groups > 1 && (p.strideX % BK) != 0) || p.a_mul != 1 || p.rowbase || (((unsigned long long)p.X) & 127)))
        return hipErrorNotSupported;
    if (p.c_planes && !(c_planes_ok(p, c) && o.epi_t4)) return hipErrorNotSupported;
    const int fi = p.a_planes ? PRO_APL : ((p.pro_act == PRO_LNX || p.stat_out) ? PRO_LNX : p.pro_act);
    // LayerNorm as a prologue of the f32 tiles (pro_act 3 / 4: rounds 1-2, measured slower than LayerNorm + GEMM) is retired: callers
    // fall back on NotSupported; the <= 64-row weight-streaming kernel (above) keeps its own LayerNorm prologue
    if (p.pro_act == PRO_LN || p.pro_act == PRO_LNA) return hipErrorNotSupported;
    size_t lds = c->lds, lds_attr = 0;
    // pair-fed LayerNorm on the loader-wave tiles: + row statistics [BM][2] behind the ring.  The PRO_LNX instantiation also serves
    // stat_out-only producers: the same size for both, so that the cached MaxDynamicSharedMemorySize attribute covers either use
    if (fi == PRO_LNX && !c->x6_ks) lds = c->lds + (size_t)c->bm * 2 * sizeof(float);
    if (c->x3h >= 0) {
        if (!p.Wh || !p.wh_inv || (p.K & 7) || (p.ldw & 7) || (p.pro_act >= PRO_LN && p.pro_act != PRO_LNX)) return hipErrorInvalidValue;
        if (p.wh_ldb == 0) p.wh_ldb = 4ll * ((p.ldw + 31) / 32 * 32);      // chunk-interleaved rows, K padded to whole chunks
        if (p.groups > 1 && p.wh_gstride == 0) p.wh_gstride = p.strideW * 4;   // (exact for whole-chunk rows; attach_planes sets it otherwise)
        if ((((unsigned long long)p.Wh) & 127) || (p.wh_ldb & 127) || (p.wh_gstride & 127)) return hipErrorInvalidValue;
        p.x3h_flag = o.x3h_flag;
    } else
    if (c->x6 && (!p.W3 || (p.K & 7) || (p.ldw & 7) || (p.pro_act >= PRO_LN && p.pro_act != PRO_LNX))) return hipErrorInvalidValue;
    if (c->x6_ks && (p.taps != 1 || p.K % (BK * c->x6_ks) != 0)) return hipErrorInvalidValue;
    if (c->x6 && c->x3h < 0 && p.w3_plane == 0) p.w3_plane = (long long)p.N * p.ldw;
    if (c->win_qs) {
        if (!win_eligible(p) || p.Cin != 32 * c->win_qs) return hipErrorInvalidValue;
        const int wrp = (c->bm + (p.taps - 1) * p.dil + 7) & ~7;
        lds = c->lds + (size_t)c->win_qs * wrp * BK * sizeof(float);
    }
    void (*fn)(GemmP) = c->fn[fi];
    if (fn && c->x3h >= 0) fn = x3h_kernel(c->x3h, fi);
    if (!fn) return hipErrorNotSupported;           // retired configuration / no variant for this prologue
    {
        if (c->win_qs) lds_attr = c->lds + (size_t)c->win_qs * ((c->bm + 64 + 7) & ~7) * BK * sizeof(float);
        hipError_t e = dyn_lds_once(g_attr_done[idx][fi], reinterpret_cast<const void*>(fn), c->win_qs ? lds_attr : lds);
        if (e != hipSuccess) return e;
    }
    if (opts) { opts->last_stat_nt = p.stat_nt; opts->last_stat_w = p.stat_w; }
    const int tiles = ((p.M + c->bm - 1) / c->bm) * ((p.N + c->bn - 1) / c->bn);
    p.epi_t4 = o.epi_t4 ? 1 : 0;
    p.ldr_prio = o.ldr_prio;
    p.ldr64 = o.ldr64 ? 1 : 0;
    dim3 grid(tiles, 1, p.groups), block(c->threads);
    if (opts) opts->last_cfg = c->name;
    if (opts && opts->trace_on) {
        TraceRec r;
        r.cfg = idx;
        r.flops = 2.0 * p.M * p.N * p.K * p.groups;
        r.M = p.M; r.N = p.N; r.K = p.K; r.groups = p.groups;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return hipErrorUnknown;
        (void)hipEventRecord(r.e0, s);
        hipLaunchKernelGGL(fn, grid, block, lds, s, p);
        (void)hipEventRecord(r.e1, s);
        opts->trace.push_back(r);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(fn, grid, block, lds, s, p);
    return hipGetLastError();
}


}  // namespace mt2
