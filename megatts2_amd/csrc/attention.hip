// Non-causal multi-head attention over per-utterance row ranges, f32, flash-style, gfx950.
// Replaces F.scaled_dot_product_attention at reference modules/transformer.py:52-53 for the four
// shapes on the synthesis path (2x256, 1x512, 8x96, 16x64) with mask=None:
// every query of an utterance attends to every key of the SAME utterance (the reference's AR
// inference is non-causal) and to nothing else (batch-1 semantics, N1).
//
// One wave64 owns a 32-query tile of one (utterance, head) and walks the keys 32 at a time, with the
// whole online-softmax state in registers and no LDS:
//   S^T = K . Q^T   on v_mfma_f32_32x32x2_f32 (A = K rows, B = Q rows; both operands are read as one
//                   float4 per lane along the head dim, the MFMA k index being a free permutation)
//          -> lane (q = lane&31) holds 16 of the 32 scores of ITS query; the other 16 live in lane^32,
//             so row max / row sum need one 32-lane-apart shuffle each and nothing else;
//   O^T += V^T . P^T  (A = V[kv(e), d0 + lane&31] - a coalesced 128-B row segment, B = p[e] straight
//                   from the score registers) -> lane holds O^T[:, q] and the softmax rescale factor
//                   alpha[q] is lane-local.
// Wide heads are split across the waves of a workgroup along d (<= 128 columns = 64 accumulator
// VGPRs per wave); each wave recomputes S (cheap: these heads only occur in the tiny cross-attention).
//
// Two kernels: attn_f32_reg_kernel<D> for head dims <= 128 (every autoregressive step of the ADM / PLM:
// 8x96, 16x64) keeps the Q fragments in registers for the whole kernel, issues ALL loads of a key tile
// (K fragments, V columns) before the first MFMA and prefetches the next tile's K fragments while the
// current tile is on the matrix pipe - these launches are latency-bound (<= 3 key tiles, one wave per
// (utterance, head, 32 queries)), so what matters is that a tile costs one memory round trip, not D/8.
// attn_f32_kernel<DT> is the streaming form for the wide heads (2x256 phone encoder, 1x512 cross).
#include "mt2_kernels.h"
#include "planes_store.h"
#include <math.h>

namespace mt2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// four consecutive head-dim columns of an output row: f32, or fp16 planes for an out-projection that takes them (AttnP::o_planes)
__device__ __forceinline__ void attn_store4(const AttnP& p, float* __restrict__ row, int c, const float4& v) {
    if (p.o_planes) {
        if (store_planes4(row, c, v) >= 65504.0f && p.x3h_flag) atomicOr(p.x3h_flag, 1);
    } else
        *reinterpret_cast<float4*>(row + c) = v;
}

template <int DT>   // DT 32-column tiles of the head dim per wave
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
    int qs, ql, ks, kl;
    if (p.q_start) {
        qs = p.q_start[b]; ql = p.q_len[b]; ks = p.kv_start[b]; kl = p.kv_len[b];
    } else {
        qs = b * p.u_qstride; ql = p.u_qlen; ks = b * p.u_kvstride; kl = p.u_kvlen;
    }
    if (qt * 32 >= ql || kl <= 0) return;
    const int D = p.D;
    const int qrow = qt * 32 + l31;
    const bool qok = qrow < ql;
    const float* __restrict__ qptr =
        p.Q + (long long)(qs + (qok ? qrow : qt * 32)) * p.ldq + h * D + 4 * half;
    const int os = p.o_start ? p.o_start[b] : (p.q_start ? qs : b * (p.u_ostride ? p.u_ostride : p.u_qstride));
    const int d0 = wave * DT * 32;
    const float scale = p.scale;

    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.0f;

    for (int kv0 = 0; kv0 < kl; kv0 += 32) {
        const int kvrow = kv0 + l31;
        const float* __restrict__ kptr =
            p.K + (long long)(ks + (kvrow < kl ? kvrow : kv0)) * p.ldk + h * D + 4 * half;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.0f;
        for (int d8 = 0; d8 < D; d8 += 8) {
            const float4 a = *reinterpret_cast<const float4*>(kptr + d8);
            const float4 bq = *reinterpret_cast<const float4*>(qptr + d8);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, s, 0, 0, 0);
        }
        // s[e] = S^T[kv0 + (e&3) + 8*(e>>2) + 4*half][q = l31]
        float mloc = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
            s[e] = kvr < kl ? s[e] * scale : -INFINITY;
            mloc = fmaxf(mloc, s[e]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);          // finite: key kv0 is always in range
        const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
        float lsum = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = expf(s[e] - m_new);
            lsum += s[e];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
            const float* __restrict__ vcol = p.V + h * D + d0 + t * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const float v = kvr < kl ? vcol[(long long)(ks + kvr) * p.ldv] : 0.0f;
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[e], o[t], 0, 0, 0);
            }
        }
    }
    if (!qok) return;
    const float inv = 1.0f / l_run;
    float* __restrict__ orow = p.O + (long long)(os + qrow) * p.ldo;
    const int oc = h * D + d0 + 4 * half;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float4 v;
            v.x = o[t][4 * e4 + 0] * inv;
            v.y = o[t][4 * e4 + 1] * inv;
            v.z = o[t][4 * e4 + 2] * inv;
            v.w = o[t][4 * e4 + 3] * inv;
            attn_store4(p, orow, oc + t * 32 + 8 * e4, v);
        }
}

// Register-resident form for D <= 128 (see the file header).  The key tiles of one (utterance, head, 32
// queries) are dealt round-robin to the 1..4 waves of the block (split-KV: a sequence of 70 positions is
// three tiles - three waves finish in the time of one); the partial (m, l, O) states are merged through LDS
// by wave 0 with the usual log-sum-exp rescale.
template <int D>
__global__ __launch_bounds__(256) void attn_f32_reg_kernel(AttnP p) {
    constexpr int NF = D / 8;      // float4 fragments per row along the head dim
    constexpr int DT = D / 32;     // 32-column output tiles
    extern __shared__ __attribute__((aligned(16))) float amem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
    int qs, ql, ks, kl;
    if (p.q_start) {
        qs = p.q_start[b]; ql = p.q_len[b]; ks = p.kv_start[b]; kl = p.kv_len[b];
    } else {
        qs = b * p.u_qstride; ql = p.u_qlen; ks = b * p.u_kvstride; kl = p.u_kvlen;
    }
    if (qt * 32 >= ql || kl <= 0) return;          // block-uniform
    const int os = p.o_start ? p.o_start[b] : (p.q_start ? qs : b * (p.u_ostride ? p.u_ostride : p.u_qstride));
    const int qrow = qt * 32 + l31;
    const bool qok = qrow < ql;
    const float scale = p.scale;
    const int kstep = 32 * nwv;

    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.0f;

    if (wave * 32 < kl) {
        float4 qf[NF], kf[NF], kn[NF];
        {
            const float* __restrict__ qptr =
                p.Q + (long long)(qs + (qok ? qrow : qt * 32)) * p.ldq + h * D + 4 * half;
#pragma unroll
            for (int f = 0; f < NF; ++f) qf[f] = *reinterpret_cast<const float4*>(qptr + 8 * f);
        }
        auto load_k = [&](int kv0, float4 (&dst)[NF]) {
            const int kvrow = kv0 + l31;
            const float* __restrict__ kptr =
                p.K + (long long)(ks + (kvrow < kl ? kvrow : kv0)) * p.ldk + h * D + 4 * half;
#pragma unroll
            for (int f = 0; f < NF; ++f) dst[f] = *reinterpret_cast<const float4*>(kptr + 8 * f);
        };
        load_k(wave * 32, kf);
        for (int kv0 = wave * 32; kv0 < kl; kv0 += kstep) {
            // every load of this tile (V columns) and this wave's next K fragments go out before the first MFMA
            float vr[DT][16];
            const float* __restrict__ vcol = p.V + h * D + l31;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    vr[t][e] = kvr < kl ? vcol[(long long)(ks + kvr) * p.ldv + t * 32] : 0.0f;
                }
            const bool more = kv0 + kstep < kl;
            if (more) load_k(kv0 + kstep, kn);

            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].x, qf[f].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].y, qf[f].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].z, qf[f].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].w, qf[f].w, s, 0, 0, 0);
            }
            // s[e] = S^T[kv0 + (e&3) + 8*(e>>2) + 4*half][q = l31]
            float mloc = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                s[e] = kvr < kl ? s[e] * scale : -INFINITY;
                mloc = fmaxf(mloc, s[e]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run, mloc);          // finite: key kv0 is always in range
            const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
            float lsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = expf(s[e] - m_new);
                lsum += s[e];
            }
            lsum += __shfl_xor(lsum, 32);
            l_run = l_run * alpha + lsum;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[t][e], s[e], o[t], 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int f = 0; f < NF; ++f) kf[f] = kn[f];
            }
        }
    }
    if (nwv > 1) {                                 // merge the per-wave partial states (block-uniform branch)
        constexpr int WS = (DT * 16 + 2) * 64;     // floats per wave: O^T tiles, m, l; [.][lane] = conflict-free
        if (wave > 0) {
            float* mine = amem + (wave - 1) * WS + lane;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) mine[(t * 16 + e) * 64] = o[t][e];
            mine[DT * 16 * 64] = m_run;
            mine[(DT * 16 + 1) * 64] = l_run;
        }
        __syncthreads();
        if (wave > 0) return;
        for (int w = 1; w < nwv; ++w) {
            const float* oth = amem + (w - 1) * WS + lane;
            const float m_o = oth[DT * 16 * 64], l_o = oth[(DT * 16 + 1) * 64];
            const float m_new = fmaxf(m_run, m_o);           // wave 0 always has key tile 0: finite
            const float a0 = expf(m_run - m_new), a1 = expf(m_o - m_new);   // exp(-inf) = 0 for an idle wave
            l_run = l_run * a0 + l_o * a1;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] = o[t][e] * a0 + oth[(t * 16 + e) * 64] * a1;
        }
    }
    if (!qok) return;
    const float inv = 1.0f / l_run;
    float* __restrict__ orow = p.O + (long long)(os + qrow) * p.ldo;
    const int oc = h * D + 4 * half;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float4 v;
            v.x = o[t][4 * e4 + 0] * inv;
            v.y = o[t][4 * e4 + 1] * inv;
            v.z = o[t][4 * e4 + 2] * inv;
            v.w = o[t][4 * e4 + 3] * inv;
            attn_store4(p, orow, oc + t * 32 + 8 * e4, v);
        }
}

// Short sequences on the AR heads (D = 64 / 96, at most 128 keys: every step of C1 - C3), round 4: the head dim is SPLIT over
// the waves as well.  In the register kernel above a wave multiplies its key tile against the whole head dim: D/2 + D/2
// v_mfma_f32_32x32x2_f32 of 64 cycles - 2.0 / 3.1 us of matrix time per launch, most of a 6.4 / 7.9-us launch that is
// otherwise one memory round trip (profiles/r04_c1_kernel_stats_tm16.csv: attention = 27 % of the one-utterance path).  Here a
// workgroup = (key tiles) x (D / 32 slices of 32 channels) waves: wave (kt, ds) computes the partial S^T of its tile over
// its 32 channels (16 MFMAs), the D/32 partials of a tile meet in LDS and are summed in slice order by each of the tile's
// waves, which then runs the tile's softmax (redundantly: it is 32 x 32) and P V for ITS 32 output channels (16 MFMAs); the
// tiles' (m, l, O-slice) states are merged by the waves of tile 0 with the log-sum-exp rescale, in tile order.  32 MFMAs per
// wave instead of 64 / 96.  f32 throughout, fixed summation order (slices ascending, keys ascending, tiles ascending).
template <int D>
__global__ __launch_bounds__(768) void attn_f32_ds_kernel(AttnP p) {
    constexpr int NS = D / 32;
    extern __shared__ __attribute__((aligned(16))) float amem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nkv = (blockDim.x >> 6) / NS;
    const int kt = wave / NS, ds = wave - kt * NS;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
    int qs, ql, ks, kl;
    if (p.q_start) {
        qs = p.q_start[b]; ql = p.q_len[b]; ks = p.kv_start[b]; kl = p.kv_len[b];
    } else {
        qs = b * p.u_qstride; ql = p.u_qlen; ks = b * p.u_kvstride; kl = p.u_kvlen;
    }
    if (qt * 32 >= ql || kl <= 0) return;          // block-uniform
    const int os = p.o_start ? p.o_start[b] : (p.q_start ? qs : b * (p.u_ostride ? p.u_ostride : p.u_qstride));
    const int qrow = qt * 32 + l31;
    const bool qok = qrow < ql;
    const int kv0 = kt * 32;
    const bool tile_ok = kv0 < kl;                 // wave-uniform (a ragged launch is sized by its longest key range)
    float* sp = amem;                              // [nkv * NS][16][64] partial scores
    float* st = amem + nkv * NS * 16 * 64;         // [nkv * NS][18][64]: O^T slice (16), m, l

    f32x16 sc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = 0.0f;
    float vr[16];
    if (tile_ok) {
        // every load of this wave goes out before its first MFMA: Q and K fragments of its 32 channels, its V column
        float4 qf[4], kf[4];
        const float* __restrict__ qptr = p.Q + (long long)(qs + (qok ? qrow : qt * 32)) * p.ldq + h * D + ds * 32 + 4 * half;
        const int kvrow = kv0 + l31;
        const float* __restrict__ kptr = p.K + (long long)(ks + (kvrow < kl ? kvrow : kv0)) * p.ldk + h * D + ds * 32 + 4 * half;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            qf[f] = *reinterpret_cast<const float4*>(qptr + 8 * f);
            kf[f] = *reinterpret_cast<const float4*>(kptr + 8 * f);
        }
        const float* __restrict__ vcol = p.V + h * D + ds * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
            vr[e] = kvr < kl ? vcol[(long long)(ks + kvr) * p.ldv] : 0.0f;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].x, qf[f].x, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].y, qf[f].y, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].z, qf[f].z, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[f].w, qf[f].w, sc, 0, 0, 0);
        }
    }
    {
        float* mine = sp + (size_t)wave * 16 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[e * 64] = sc[e];
    }
    __syncthreads();
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.0f;
    if (tile_ok) {
        // the tile's scores: its NS partials in slice order; sc[e] = S^T[kv0 + (e&3) + 8*(e>>2) + 4*half][q = l31]
        const float* part = sp + (size_t)(kt * NS) * 16 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = part[e * 64];
#pragma unroll
            for (int x = 1; x < NS; ++x) v += part[(x * 16 + e) * 64];
            const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
            sc[e] = kvr < kl ? v * p.scale : -INFINITY;
            m_run = fmaxf(m_run, sc[e]);
        }
        m_run = fmaxf(m_run, __shfl_xor(m_run, 32));      // finite: key kv0 is in range
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sc[e] = expf(sc[e] - m_run);
            l_run += sc[e];
        }
        l_run += __shfl_xor(l_run, 32);
#pragma unroll
        for (int e = 0; e < 16; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[e], sc[e], o, 0, 0, 0);
    }
    if (nkv > 1) {                                  // block-uniform
        if (kt > 0) {
            float* mine = st + (size_t)wave * 18 * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[e * 64] = o[e];
            mine[16 * 64] = m_run;
            mine[17 * 64] = l_run;
        }
        __syncthreads();
        if (kt > 0) return;
        for (int t = 1; t < nkv; ++t) {             // tile 0 always holds key 0: m_run is finite
            const float* oth = st + (size_t)(t * NS + ds) * 18 * 64 + lane;
            const float m_o = oth[16 * 64], l_o = oth[17 * 64];
            const float m_new = fmaxf(m_run, m_o);
            const float a0 = expf(m_run - m_new), a1 = expf(m_o - m_new);       // exp(-inf) = 0 for an empty tile
            l_run = l_run * a0 + l_o * a1;
            m_run = m_new;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = o[e] * a0 + oth[e * 64] * a1;
        }
    }
    if (!qok) return;
    const float inv = 1.0f / l_run;
    float* __restrict__ orow = p.O + (long long)(os + qrow) * p.ldo;
    const int oc = h * D + ds * 32 + 4 * half;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        float4 v;
        v.x = o[4 * e4 + 0] * inv;
        v.y = o[4 * e4 + 1] * inv;
        v.z = o[4 * e4 + 2] * inv;
        v.w = o[4 * e4 + 3] * inv;
        attn_store4(p, orow, oc + 8 * e4, v);
    }
}
template <int D>
static hipError_t launch_attn_ds(const AttnP& p, int nkv, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};      // per device (dyn_lds_once)
    constexpr int NS = D / 32;
    const size_t lds = (size_t)nkv * NS * (16 + 18) * 64 * sizeof(float);
    {                       // up to 4 x 3 waves: 102 KiB
        hipError_t e = dyn_lds_once(attr_done, reinterpret_cast<const void*>(attn_f32_ds_kernel<D>),
                                    (size_t)4 * NS * (16 + 18) * 64 * sizeof(float));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(attn_f32_ds_kernel<D>, dim3((p.max_qlen + 31) / 32, p.H, p.B), dim3(64 * nkv * NS), lds, s, p);
    return hipGetLastError();
}

// Very long sequences (default: >= 640 queries, AttnP::lds_min_qlen): the LDS-tiled form.  The register kernel above gives every 32-query tile its own
// workgroup and lets each of its waves fetch "its" K / V tiles from memory - right for the <= 3 key tiles of a 70-position
// step (one round trip per launch), wrong for 834 positions: 27 query tiles re-read the whole K / V of the head, a wave
// holds ~270 registers (one wave per SIMD) and the matrix pipe idles through every softmax and every load
// (profiles/r03_c5_kernel_stats.csv: 47 TF/s, 18 % of a C5 step).  Here a workgroup of 4 or 8 waves owns as many consecutive query
// tiles of one (utterance, head); the K and V tiles (32 keys x D, f32) are loaded ONCE per workgroup, coalesced, one tile
// ahead, into a double-buffered LDS stage (rows padded to D + 4 floats) and every wave reads its MFMA operands from
// there: 4-8x fewer global loads, two waves per SIMD (one wave's softmax under the other's MFMAs).  The per-tile
// arithmetic is the register kernel's (same MFMA, same online-softmax update, keys in ascending order).  Measured
// (tools/attn_bench.py, profiles/r03_attn_bench.txt): +25 % at 834 positions (60 vs 45 TF/s), nothing below ~600 - both
// kernels sit at ~0.65 of the f32 MFMA rate at the sustained clock; what is left is the arithmetic (an x6 form), not the loads.
template <int D, int NWQ>     // NWQ waves = NWQ consecutive query tiles per workgroup
__global__ __launch_bounds__(64 * NWQ) void attn_f32_lds_kernel(AttnP p) {
    constexpr int NT = 64 * NWQ;
    constexpr int NF = D / 8, DT = D / 32, LDK = D + 4, NLD = 16 * D / NT;   // NLD float4 per thread and tile (K and V together)
    static_assert((16 * D) % NT == 0, "tile size must be a multiple of the workgroup size");
    constexpr int TILE = 32 * LDK;                                      // floats per K (or V) tile in LDS
    extern __shared__ __attribute__((aligned(16))) float amem[];        // [2 stages][K | V][32][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    int qs, ql, ks, kl;
    if (p.q_start) {
        qs = p.q_start[b]; ql = p.q_len[b]; ks = p.kv_start[b]; kl = p.kv_len[b];
    } else {
        qs = b * p.u_qstride; ql = p.u_qlen; ks = b * p.u_kvstride; kl = p.u_kvlen;
    }
    if (blockIdx.x * 32 * NWQ >= ql || kl <= 0) return;                 // block-uniform
    const int os = p.o_start ? p.o_start[b] : (p.q_start ? qs : b * (p.u_ostride ? p.u_ostride : p.u_qstride));
    const int qt = blockIdx.x * NWQ + wave;
    const bool active = qt * 32 < ql;                                   // wave-uniform: idle waves still load and synchronise
    const int qrow = qt * 32 + l31;
    const bool qok = qrow < ql;
    const float scale = p.scale;

    // cooperative tile load: float4 index f = tid + j * NT over [K rows | V rows] (32 * D / 4 float4 each)
    float4 pre[NLD];
    auto gload = [&](int kv0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = tid + j * NT;
            const int isv = f >= 8 * D ? 1 : 0;
            const int g = f - isv * 8 * D, row = g / (D / 4), c4 = g - row * (D / 4);
            const int kvr = kv0 + row;
            const float* src = isv ? p.V + (long long)(ks + kvr) * p.ldv + h * D + c4 * 4
                                   : p.K + (long long)(ks + kvr) * p.ldk + h * D + c4 * 4;
            pre[j] = kvr < kl ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int st) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = tid + j * NT;
            const int isv = f >= 8 * D ? 1 : 0;
            const int g = f - isv * 8 * D, row = g / (D / 4), c4 = g - row * (D / 4);
            *reinterpret_cast<float4*>(amem + (st * 2 + isv) * TILE + row * LDK + c4 * 4) = pre[j];
        }
    };

    float4 qf[NF];
    {
        const float* __restrict__ qptr = p.Q + (long long)(qs + (qok ? qrow : 0)) * p.ldq + h * D + 4 * half;
#pragma unroll
        for (int f = 0; f < NF; ++f) qf[f] = active ? *reinterpret_cast<const float4*>(qptr + 8 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.0f;

    gload(0);
    sstore(0);
    __syncthreads();
    int st = 0;
    for (int kv0 = 0; kv0 < kl; kv0 += 32) {
        const bool more = kv0 + 32 < kl;                                // block-uniform
        if (more) gload(kv0 + 32);                                      // in flight during this tile's MFMAs
        if (active) {
            const float* Ks = amem + (st * 2) * TILE + l31 * LDK + 4 * half;
            const float* Vs = amem + (st * 2 + 1) * TILE + l31;
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float4 kf = *reinterpret_cast<const float4*>(Ks + 8 * f);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[f].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[f].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[f].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[f].w, s, 0, 0, 0);
            }
            // s[e] = S^T[kv0 + (e&3) + 8*(e>>2) + 4*half][q = l31]
            float mloc = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                s[e] = kvr < kl ? s[e] * scale : -INFINITY;
                mloc = fmaxf(mloc, s[e]);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run, mloc);          // finite: key kv0 is always in range
            const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
            float lsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] = expf(s[e] - m_new);
                lsum += s[e];
            }
            lsum += __shfl_xor(lsum, 32);
            l_run = l_run * alpha + lsum;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = Vs[((e & 3) + 8 * (e >> 2) + 4 * half) * LDK + t * 32];   // rows past kl are zero in LDS
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[e], o[t], 0, 0, 0);
                }
            }
        }
        if (more) sstore(st ^ 1);      // the other stage was last read before the barrier that ended the previous tile
        __syncthreads();
        st ^= 1;
    }
    if (!active || !qok) return;
    const float inv = 1.0f / l_run;
    float* __restrict__ orow = p.O + (long long)(os + qrow) * p.ldo;
    const int oc = h * D + 4 * half;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float4 v;
            v.x = o[t][4 * e4 + 0] * inv;
            v.y = o[t][4 * e4 + 1] * inv;
            v.z = o[t][4 * e4 + 2] * inv;
            v.w = o[t][4 * e4 + 3] * inv;
            attn_store4(p, orow, oc + t * 32 + 8 * e4, v);
        }
}

template <int D, int NWQ>
static hipError_t launch_attn_lds(const AttnP& p, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};      // per device (dyn_lds_once)
    void (*fn)(AttnP) = attn_f32_lds_kernel<D, NWQ>;
    const size_t lds = (size_t)4 * 32 * (D + 4) * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = dyn_lds_once(attr_done, reinterpret_cast<const void*>(fn), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3((p.max_qlen + 32 * NWQ - 1) / (32 * NWQ), p.H, p.B), dim3(64 * NWQ), lds, s, p);
    return hipGetLastError();
}
template <int D>
static hipError_t launch_attn_lds_d(const AttnP& p, hipStream_t s) {
    // 4 query tiles per workgroup (two workgroups per CU) unless 8 is asked for: profiles/r03_attn_bench.txt - 8x96 at
    // 834 positions 189 us (register kernel) / 141 (4) / 213 (8); 16x64 at 646: 137 / 142 / 133
    if (p.lds_waves == 8) return launch_attn_lds<D, 8>(p, s);
    return launch_attn_lds<D, 4>(p, s);
}

// ---------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix pipe, f32-equivalent ("x6", see gemm_f32.hip: an f32 number is exactly the sum of
// three bf16 numbers, a product of two bf16 is exact in f32, and six of the nine plane products carry everything above
// 2^-24 |a||b|).  Both contractions take activations on both sides, so every operand is split at run time:
//   K tile  [32 keys][D]      -> three bf16 planes in LDS, row-major           (split ONCE per workgroup and tile)
//   V tile  [32 keys][D]      -> three bf16 planes in LDS, TRANSPOSED [D][32]  (so that a lane reads 8 keys of one channel)
//   Q rows  (one per lane)    -> three planes in registers                      (once per workgroup)
//   P = exp(S - m)            -> three planes in registers, per tile and wave: the accumulator layout of S^T (lane = query,
//                                16 of the 32 keys) IS the B-operand layout of v_mfma_f32_32x32x16_bf16 once the keys of a
//                                16-key block are permuted (a free choice, the same permutation is applied to V^T in LDS).
// Per 32x32 tile pair: 6 * D/16 + 6 * 2 * D/32 MFMAs of 32 cycles instead of 2 * D/2 ... of 64: 2.67x fewer pipe cycles.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
typedef float af32x4 __attribute__((ext_vector_type(4)));

// (lo, hi) = 8 consecutive f32 -> three planes of 8 bf16 (truncation split, exact sum), element 2i / 2i+1 in dword i
__device__ __forceinline__ void attn_split3(const af32x4& lo, const af32x4& hi, au32x4& p1, au32x4& p2, au32x4& p3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = i < 2 ? lo[2 * i] : hi[2 * i - 4];
        const float y = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
        const unsigned xb = __float_as_uint(x), yb = __float_as_uint(y);
        p1[i] = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
        const float xr = x - __uint_as_float(xb & 0xffff0000u), yr = y - __uint_as_float(yb & 0xffff0000u);
        const unsigned xc = __float_as_uint(xr), yc = __float_as_uint(yr);
        p2[i] = __builtin_amdgcn_perm(yc, xc, 0x07060302u);
        const float xs = xr - __uint_as_float(xc & 0xffff0000u), ys = yr - __uint_as_float(yc & 0xffff0000u);
        p3[i] = __builtin_amdgcn_perm(__float_as_uint(ys), __float_as_uint(xs), 0x07060302u);
    }
}
// acc += A B with the six plane products, smallest terms first
__device__ __forceinline__ void attn_x6(f32x16& acc, const au32x4 (&a)[3], const au32x4 (&b)[3]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, a[PA[t]]), __builtin_bit_cast(abf16x8, b[PB[t]]), acc, 0, 0, 0);
}

// The fp16-pipe form of the same kernel ("x3h", round 6; gemm_x3h.hip has the derivation): every operand as TWO fp16 planes,
// a = hi + 2^-11 lo with hi = fp16_rn(a), lo = fp16_rn((a - hi) * 2^11), THREE products per k block - hi hi into the main
// accumulator, hi lo + lo hi into a second one that is merged with the weight 2^-11 (the dropped lo lo term is 2^-22 relative).
// Half the matrix instructions of x6, two planes instead of three in LDS and in the split arithmetic.  fp16 has no f32 exponent
// range: Q / K / V values at or beyond 65504 raise the handle's range-guard word (AttnP::x3h_flag) and the caller repeats the
// call with x3h = 0, i.e. on the x6 kernel; P = exp(s - m) lies in [0, 1] (values below 2^-35 vanish: 2^-36 of the row maximum).
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x2 __attribute__((ext_vector_type(2)));
typedef float af32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void attn_split2(const af32x4& lo, const af32x4& hi, au32x4& ph, au32x4& pl, float& amax) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = i < 2 ? lo[2 * i] : hi[2 * i - 4];
        const float y = i < 2 ? lo[2 * i + 1] : hi[2 * i - 3];
        amax = fmaxf(fmaxf(fabsf(x), fabsf(y)), amax);
        const af16x2 h = __builtin_convertvector((af32x2){x, y}, af16x2);            // round to nearest even
        const float rx = __builtin_fmaf((float)h[0], -2048.0f, x * 2048.0f);        // (x - h) * 2^11, exact
        const float ry = __builtin_fmaf((float)h[1], -2048.0f, y * 2048.0f);
        const af16x2 l = __builtin_convertvector((af32x2){rx, ry}, af16x2);
        ph[i] = __builtin_bit_cast(unsigned, h);
        pl[i] = __builtin_bit_cast(unsigned, l);
    }
}
// acc += A_hi B_hi;  accl += A_hi B_lo + A_lo B_hi  (accl carries the weight 2^-11)
__device__ __forceinline__ void attn_x3h(f32x16& acc, f32x16& accl, const au32x4 (&a)[2], const au32x4 (&b)[2]) {
    const af16x8 ah = __builtin_bit_cast(af16x8, a[0]), al = __builtin_bit_cast(af16x8, a[1]);
    const af16x8 bh = __builtin_bit_cast(af16x8, b[0]), bl = __builtin_bit_cast(af16x8, b[1]);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accl, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accl, 0, 0, 0);
}

// Round 4: a DOUBLE-BUFFERED plane stage (split of tile t+1 between the S products and the softmax of tile t, one barrier per
// tile) was built and measured: parity-green, 0...-5 % on the isolated launch, +0.4 % on the C5 step (twice the LDS per
// workgroup) - not kept (profiles/r04_attn_bench.txt, r04_experiment_attn_double_buffer.patch).  A tile's cost is its VALU work -
// the softmax (16 expf per lane), two run-time splits of P and the rescale of O - not the barrier count: what WAS kept are
// the exact savings there (no key mask on full tiles, no rescale when no query's running maximum moved).
template <int D, int NWQ, bool H3 = false>     // D = 64 or 96; workgroup = NWQ waves = NWQ consecutive query tiles of one (utterance, head);
                                               // H3: the fp16-pipe form (two planes, three products) instead of bf16 (three planes, six)
__global__ __launch_bounds__(64 * NWQ) __attribute__((amdgpu_waves_per_eu(2))) void attn_x6_kernel(AttnP p) {
    constexpr int NT = 64 * NWQ;
    constexpr int NP = H3 ? 2 : 3;                  // planes per operand
    float amax = 0.0f;                              // H3: largest |Q|, |K|, |V| value this thread split (range guard)
    constexpr int NB = D / 16, DT = D / 32;
    constexpr int RSK = D * 2 + 16;                 // bytes per key row of a K plane (padded)
    constexpr int RSV = 64 + 16;                    // bytes per channel row of a V^T plane: 32 keys (permuted) + pad
    constexpr int KPL = 32 * RSK, VPL = D * RSV;    // bytes per plane
    constexpr int KI = (4 * D + NT - 1) / NT;       // staging items per thread: K (32 rows x D/8 groups of 8 channels) ...
    constexpr int VI = (4 * D + NT - 1) / NT;       // ... and V (D channels x 4 groups of 8 keys)
    extern __shared__ __attribute__((aligned(16))) char xmem[];          // [NP K planes][NP V^T planes]
    char* kp = xmem;
    char* vp = xmem + NP * KPL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    int qs, ql, ks, kl;
    if (p.q_start) {
        qs = p.q_start[b]; ql = p.q_len[b]; ks = p.kv_start[b]; kl = p.kv_len[b];
    } else {
        qs = b * p.u_qstride; ql = p.u_qlen; ks = b * p.u_kvstride; kl = p.u_kvlen;
    }
    if (blockIdx.x * 32 * NWQ >= ql || kl <= 0) return;                      // block-uniform
    const int os = p.o_start ? p.o_start[b] : (p.q_start ? qs : b * (p.u_ostride ? p.u_ostride : p.u_qstride));
    const int qt = blockIdx.x * NWQ + wave;
    const bool active = qt * 32 < ql;                                   // wave-uniform
    const int qrow = qt * 32 + l31;
    const bool qok = qrow < ql;
    const float scale = p.scale;

    // ---- staging: global -> registers (one tile ahead) -> split -> LDS planes
    af32x4 kreg[KI][2];
    float vreg[VI][8];
    auto prefetch = [&](int kv0) {
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int i = tid + j * NT;
            if (i < 4 * D) {
                const int row = i / (D / 8), grp = i - row * (D / 8);
                const bool ok = kv0 + row < kl;
                const float* src = p.K + (long long)(ks + (ok ? kv0 + row : 0)) * p.ldk + h * D + grp * 8;
                const af32x4 z = {0.f, 0.f, 0.f, 0.f};
                kreg[j][0] = ok ? *reinterpret_cast<const af32x4*>(src) : z;
                kreg[j][1] = ok ? *reinterpret_cast<const af32x4*>(src + 4) : z;
            }
        }
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int i = tid + j * NT;
            if (i < 4 * D) {
                const int G = i / D, d = i - G * D;                      // G = 2 * key block + operand half
                const float* src = p.V + (long long)ks * p.ldv + h * D + d;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int kvr = kv0 + 16 * (G >> 1) + 4 * (G & 1) + (q & 3) + 8 * (q >> 2);
                    vreg[j][q] = kvr < kl ? src[(long long)kvr * p.ldv] : 0.0f;
                }
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int i = tid + j * NT;
            if (i < 4 * D) {
                const int row = i / (D / 8), grp = i - row * (D / 8);
                au32x4 a, bq, c;
                char* dst = kp + row * RSK + grp * 16;
                if constexpr (H3) attn_split2(kreg[j][0], kreg[j][1], a, bq, amax);
                else attn_split3(kreg[j][0], kreg[j][1], a, bq, c);
                *reinterpret_cast<au32x4*>(dst) = a;
                *reinterpret_cast<au32x4*>(dst + KPL) = bq;
                if constexpr (!H3) *reinterpret_cast<au32x4*>(dst + 2 * KPL) = c;
            }
        }
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int i = tid + j * NT;
            if (i < 4 * D) {
                const int G = i / D, d = i - G * D;
                const af32x4 lo = {vreg[j][0], vreg[j][1], vreg[j][2], vreg[j][3]};
                const af32x4 hi = {vreg[j][4], vreg[j][5], vreg[j][6], vreg[j][7]};
                au32x4 a, bq, c;
                char* dst = vp + d * RSV + G * 16;
                if constexpr (H3) attn_split2(lo, hi, a, bq, amax);
                else attn_split3(lo, hi, a, bq, c);
                *reinterpret_cast<au32x4*>(dst) = a;
                *reinterpret_cast<au32x4*>(dst + VPL) = bq;
                if constexpr (!H3) *reinterpret_cast<au32x4*>(dst + 2 * VPL) = c;
            }
        }
    };

    // ---- Q planes: lane (query l31, half) holds channels 16 f + 8 half + 0..7 of its row
    au32x4 qp[NB][NP];
    {
        const float* __restrict__ qptr = p.Q + (long long)(qs + (qok ? qrow : 0)) * p.ldq + h * D + 8 * half;
#pragma unroll
        for (int f = 0; f < NB; ++f) {
            af32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
            if (active) {
                lo = *reinterpret_cast<const af32x4*>(qptr + 16 * f);
                hi = *reinterpret_cast<const af32x4*>(qptr + 16 * f + 4);
            }
            if constexpr (H3) attn_split2(lo, hi, qp[f][0], qp[f][1], amax);
            else attn_split3(lo, hi, qp[f][0], qp[f][1], qp[f][NP - 1]);
        }
    }
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.0f;

    f32x16 sc;
    auto scores = [&]() {                 // S^T = K Q^T of the staged tile
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = 0.0f;
        const char* krow = kp + l31 * RSK + half * 16;
        if constexpr (H3) {
            f32x16 scl;
#pragma unroll
            for (int e = 0; e < 16; ++e) scl[e] = 0.0f;
#pragma unroll
            for (int f = 0; f < NB; ++f) {
                au32x4 ka[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) ka[pl] = *reinterpret_cast<const au32x4*>(krow + pl * KPL + f * 32);
                attn_x3h(sc, scl, ka, qp[f]);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[e] = __builtin_fmaf(scl[e], 1.0f / 2048.0f, sc[e]);
        } else {
#pragma unroll
            for (int f = 0; f < NB; ++f) {
                au32x4 ka[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ka[pl] = *reinterpret_cast<const au32x4*>(krow + pl * KPL + f * 32);
                attn_x6(sc, ka, qp[f]);
            }
        }
    };
    auto softmax_pv = [&](int kv0) {
        f32x16& s = sc;
        // s[e] = S^T[kv0 + (e&3) + 8*(e>>2) + 4*half][q = l31]
        float mloc = -INFINITY;
        if (kv0 + 32 <= kl) {              // a full tile (block-uniform): no key mask
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[e] *= scale;
                mloc = fmaxf(mloc, s[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kvr = kv0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                s[e] = kvr < kl ? s[e] * scale : -INFINITY;
                mloc = fmaxf(mloc, s[e]);
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = expf(m_run - m_new);
        float lsum = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = expf(s[e] - m_new);
            lsum += s[e];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {        // the running maximum moved for some query of this wave (rare after the first tiles):
#pragma unroll                             // x * 1.0f == x, so skipping the rescale changes no bit
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
        }
        // P planes of key block kb: the lane's s[8 kb .. 8 kb + 7] (keys 16 kb + 4 half + {0..3, 8..11}: the permuted order
        // in which stage() laid out V^T)
        if constexpr (H3) {
            // both key blocks' planes first, then per output tile: the cross terms of the tile in a temporary accumulator, merged
            // into o[t] once per key tile (16 fma) - one accumulator set stays resident, not two
            au32x4 pp[2][2];
            float pmax = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const af32x4 lo = {s[8 * kb], s[8 * kb + 1], s[8 * kb + 2], s[8 * kb + 3]};
                const af32x4 hi = {s[8 * kb + 4], s[8 * kb + 5], s[8 * kb + 6], s[8 * kb + 7]};
                attn_split2(lo, hi, pp[kb][0], pp[kb][1], pmax);
            }
            (void)pmax;                    // P lies in [0, 1]: no guard
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                f32x16 ol;
#pragma unroll
                for (int e = 0; e < 16; ++e) ol[e] = 0.0f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const char* vrow = vp + (t * 32 + l31) * RSV + (2 * kb + half) * 16;
                    au32x4 va[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) va[pl] = *reinterpret_cast<const au32x4*>(vrow + pl * VPL);
                    attn_x3h(o[t], ol, va, pp[kb]);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] = __builtin_fmaf(ol[e], 1.0f / 2048.0f, o[t][e]);
            }
        } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const af32x4 lo = {s[8 * kb], s[8 * kb + 1], s[8 * kb + 2], s[8 * kb + 3]};
            const af32x4 hi = {s[8 * kb + 4], s[8 * kb + 5], s[8 * kb + 6], s[8 * kb + 7]};
            au32x4 pp[NP];
            attn_split3(lo, hi, pp[0], pp[1], pp[NP - 1]);
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const char* vrow = vp + (t * 32 + l31) * RSV + (2 * kb + half) * 16;
                au32x4 va[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) va[pl] = *reinterpret_cast<const au32x4*>(vrow + pl * VPL);
                attn_x6(o[t], va, pp);
            }
        }
        }
    };

    prefetch(0);
    for (int kv0 = 0; kv0 < kl; kv0 += 32) {
        stage();
        __syncthreads();
        if (kv0 + 32 < kl) prefetch(kv0 + 32);                          // block-uniform; in flight during the MFMAs
        if (active) {
            scores();
            softmax_pv(kv0);
        }
        __syncthreads();               // every wave is done with this tile's planes before the next stage() overwrites them
    }
    if constexpr (H3) {
        if (__any(amax >= 65504.0f) && lane == 0 && p.x3h_flag) atomicOr(p.x3h_flag, 1);
    }
    if (!active || !qok) return;
    const float inv = 1.0f / l_run;
    float* __restrict__ orow = p.O + (long long)(os + qrow) * p.ldo;
    const int oc = h * D + 4 * half;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float4 v;
            v.x = o[t][4 * e4 + 0] * inv;
            v.y = o[t][4 * e4 + 1] * inv;
            v.z = o[t][4 * e4 + 2] * inv;
            v.w = o[t][4 * e4 + 3] * inv;
            attn_store4(p, orow, oc + t * 32 + 8 * e4, v);
        }
}

template <int D, int NWQ, bool H3 = false>
static hipError_t launch_attn_x6(const AttnP& p, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};      // per device (dyn_lds_once)
    void (*fn)(AttnP) = attn_x6_kernel<D, NWQ, H3>;
    const size_t lds = (size_t)(H3 ? 2 : 3) * (32 * (D * 2 + 16) + D * 80);
    if (lds > 48 * 1024) {
        hipError_t e = dyn_lds_once(attr_done, reinterpret_cast<const void*>(fn), lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3((p.max_qlen + 32 * NWQ - 1) / (32 * NWQ), p.H, p.B), dim3(64 * NWQ), lds, s, p);
    return hipGetLastError();
}
template <int D>
static hipError_t launch_attn_x6_d(const AttnP& p, hipStream_t s) {
    // the K / V split of a tile is paid once per workgroup: 8 query tiles per workgroup once there are enough of them
    const bool w8 = p.lds_waves == 8 || (p.lds_waves != 4 && p.max_qlen >= 600);      // profiles/r03_attn_bench.txt
    if (p.x3h) return w8 ? launch_attn_x6<D, 8, true>(p, s) : launch_attn_x6<D, 4, true>(p, s);
    return w8 ? launch_attn_x6<D, 8>(p, s) : launch_attn_x6<D, 4>(p, s);
}

hipError_t launch_attention(const AttnP& p, hipStream_t s) {
    if (p.B <= 0 || p.H <= 0 || p.max_qlen <= 0) return hipSuccess;
    if (p.D % 32 != 0 || (p.ldq & 3) || (p.ldk & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
    if (p.o_planes && ((p.ldo & 31) || (((unsigned long long)p.O) & 127))) return hipErrorInvalidValue;      // whole 128-byte blocks per row
    if ((p.D == 64 || p.D == 96) && p.x6_min_qlen > 0 && p.max_qlen >= p.x6_min_qlen && ((p.ldq | p.ldk) & 3) == 0)
        return p.D == 64 ? launch_attn_x6_d<64>(p, s) : launch_attn_x6_d<96>(p, s);
    if (p.D <= 128 && p.max_qlen >= p.lds_min_qlen && p.lds_min_qlen > 0 && (p.ldv & 3) == 0) {
        switch (p.D) {        // long sequences: K / V tiles shared by 8 query tiles through LDS
            case 32: return launch_attn_lds_d<32>(p, s);
            case 64: return launch_attn_lds_d<64>(p, s);
            case 96: return launch_attn_lds_d<96>(p, s);
            default: return launch_attn_lds_d<128>(p, s);
        }
    }
    if (p.D <= 128) {
        // split-KV width: the longest key range of the launch (uniform geometry knows it exactly; ragged
        // launches pass max_kvlen, 0 = unknown -> assume as long as the queries)
        int kvmax = p.q_start ? (p.max_kvlen > 0 ? p.max_kvlen : p.max_qlen) : p.u_kvlen;
        if (p.ds_short && (p.D == 64 || p.D == 96) && kvmax <= 128 && (!p.q_start || p.max_kvlen > 0)) {      // kvmax must be a true bound
            // the AR steps' short sequences: key tiles x head-dim slices (attn_f32_ds_kernel)
            const int nkv = (kvmax + 31) / 32;
            return p.D == 64 ? launch_attn_ds<64>(p, nkv < 1 ? 1 : nkv, s) : launch_attn_ds<96>(p, nkv < 1 ? 1 : nkv, s);
        }
        int nwv = (kvmax + 31) / 32;
        nwv = nwv < 1 ? 1 : (nwv > 4 ? 4 : nwv);
        const int DT = p.D / 32;
        const size_t lds = nwv > 1 ? (size_t)(nwv - 1) * (DT * 16 + 2) * 64 * sizeof(float) : 0;
        dim3 grid((p.max_qlen + 31) / 32, p.H, p.B), block(64 * nwv);
        switch (p.D) {
            case 32: hipLaunchKernelGGL(attn_f32_reg_kernel<32>, grid, block, lds, s, p); break;
            case 64: hipLaunchKernelGGL(attn_f32_reg_kernel<64>, grid, block, lds, s, p); break;
            case 96: hipLaunchKernelGGL(attn_f32_reg_kernel<96>, grid, block, lds, s, p); break;
            default: hipLaunchKernelGGL(attn_f32_reg_kernel<128>, grid, block, lds, s, p); break;
        }
        return hipGetLastError();
    }
    const int tiles = p.D / 32;
    int nw = (p.D + 127) / 128;
    while (nw <= 4 && tiles % nw != 0) ++nw;
    if (nw > 4 || tiles / nw > 4) return hipErrorInvalidValue;
    const int dt = tiles / nw;
    dim3 grid((p.max_qlen + 31) / 32, p.H, p.B), block(64 * nw);
    switch (dt) {
        case 1: hipLaunchKernelGGL(attn_f32_kernel<1>, grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL(attn_f32_kernel<2>, grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL(attn_f32_kernel<3>, grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL(attn_f32_kernel<4>, grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

}  // namespace mt2
