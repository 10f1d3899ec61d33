// HBM-bound row kernels of the Mega-TTS 2 synthesis path (gfx950): LayerNorm, embedding + positional
// table, length-regulator gather, ceil-mode max-pool, VQ L2-argmin / decode, AR step assembly and heads.
// All of them move each byte once with float4 (16 B/lane) coalesced accesses; row reductions use wave64
// shuffles (one wave per row), never LDS.
#include "mt2_kernels.h"
#include "planes_store.h"
#include <math.h>

namespace mt2 {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float act_rt2(int act, float v) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm: F.layer_norm(x, (C,), gamma, beta, eps=1e-5) at reference modules/convnet.py:29,
// modules/transformer.py:94-99, modules/mrte.py:168.  Two-pass (mean, then centred variance) in
// registers: a lane holds up to 4 float4 of its row (C <= 1024).
__global__ __launch_bounds__(256) void layernorm_kernel(LnP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= p.M) return;
    const int C = p.C;
    float* __restrict__ out = p.out + (long long)m * p.ldo;
    const int vm = p.valid_rows > 0 ? m % p.valid_rows : m;
    const bool live = !(p.valid && p.valid[vm] == 0);
    float4 x[4];
    float s = 0.0f;
    if (live) {
        const float* __restrict__ xr = p.x + (long long)m * p.ldx;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = (v * 64 + lane) * 4;
            x[v] = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (x[v].x + x[v].y) + (x[v].z + x[v].w);
        }
    }
    if (!live) {   // gap row: stays all-zero
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = (v * 64 + lane) * 4;
            if (c < C) *reinterpret_cast<float4*>(out + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float inv_c = 1.0f / (float)C;
    const float mean = wave_sum(s) * inv_c;
    float q = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < C) {
            x[v].x -= mean; x[v].y -= mean; x[v].z -= mean; x[v].w -= mean;
            q += (x[v].x * x[v].x + x[v].y * x[v].y) + (x[v].z * x[v].z + x[v].w * x[v].w);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_c + p.eps);
    const int g = p.rows_per_group > 0 ? m / p.rows_per_group : 0;
    const float* __restrict__ gam = p.gamma + (long long)g * C;
    const float* __restrict__ bet = p.beta + (long long)g * C;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < C) {
            const float4 gv = *reinterpret_cast<const float4*>(gam + c);
            const float4 bv = *reinterpret_cast<const float4*>(bet + c);
            float4 y;
            y.x = act_rt2(p.act, x[v].x * rstd * gv.x + bv.x);
            y.y = act_rt2(p.act, x[v].y * rstd * gv.y + bv.y);
            y.z = act_rt2(p.act, x[v].z * rstd * gv.z + bv.z);
            y.w = act_rt2(p.act, x[v].w * rstd * gv.w + bv.w);
            if (p.R1) {
                const int rm = p.r1_rows > 0 ? m % p.r1_rows : m;
                const float4 r = *reinterpret_cast<const float4*>(p.R1 + (long long)rm * p.ldr1 + c);
                y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
            }
            if (p.R2) {
                const float4 r = *reinterpret_cast<const float4*>(p.R2 + (long long)m * p.ldr2 + c);
                y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
            }
            if (p.out_planes) { if (store_planes4(out, c, y) >= 65504.0f && p.x3h_flag) atomicOr(p.x3h_flag, 1); }
            else *reinterpret_cast<float4*>(out + c) = y;
        }
    }
}

// Split-K consumer: x_new[m, :] = R[m, :] + bias + sum_{g < S} parts[g][m, :]  (fixed order g = 0..S-1),
// xout = x_new (the residual stream), hout = LayerNorm(x_new).  The GEMM that produced `parts` ran as S
// independent K slices (GemmP groups) and skipped its epilogue; the kernel boundary is the only
// synchronisation, the reduction is deterministic, and no launch is added: this IS the layer's LayerNorm.
__global__ __launch_bounds__(256) void ln_reduce_kernel(LnReduceP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= p.M) return;
    const int C = p.C;
    float4 x[4];
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = (v * 64 + lane) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            for (int g = 0; g < p.S; ++g) {
                const float4 q = *reinterpret_cast<const float4*>(p.parts + (long long)g * p.pstride + (long long)m * C + c);
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            }
            if (p.bias) {
                const float4 q = *reinterpret_cast<const float4*>(p.bias + c);
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            }
            if (p.R) {   // residual last, as the fused GEMM epilogue does: (acc + bias) + R
                const float4 q = *reinterpret_cast<const float4*>(p.R + (long long)m * p.ldr + c);
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            }
            if (p.xout) *reinterpret_cast<float4*>(p.xout + (long long)m * p.ldx + c) = a;
        }
        x[v] = a;
        s += (a.x + a.y) + (a.z + a.w);
    }
    const float inv_c = 1.0f / (float)C;
    const float mean = wave_sum(s) * inv_c;
    float q2 = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < C) {
            x[v].x -= mean; x[v].y -= mean; x[v].z -= mean; x[v].w -= mean;
            q2 += (x[v].x * x[v].x + x[v].y * x[v].y) + (x[v].z * x[v].z + x[v].w * x[v].w);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q2) * inv_c + p.eps);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = (v * 64 + lane) * 4;
        if (c < C) {
            const float4 gv = *reinterpret_cast<const float4*>(p.gamma + c);
            const float4 bv = *reinterpret_cast<const float4*>(p.beta + c);
            float4 y;
            y.x = x[v].x * rstd * gv.x + bv.x;
            y.y = x[v].y * rstd * gv.y + bv.y;
            y.z = x[v].z * rstd * gv.z + bv.z;
            y.w = x[v].w * rstd * gv.w + bv.w;
            *reinterpret_cast<float4*>(p.hout + (long long)m * p.ldh + c) = y;
        }
    }
}
// The same reduction with ONE ROW PER WORKGROUP (thread = one float4 of the row): every slab load of a thread is
// independent of the others and they are issued eight at a time, so a row costs one or two memory round trips instead
// of 4 * S serial ones (a wave per row: 16 us at S = 16 whatever M is - the AR steps' most expensive small launch,
// profiles/r03_c1_kernel_stats.csv).  Same summation order per element: slabs ascending, then bias, then the residual.
__global__ __launch_bounds__(256) void ln_reduce_row_kernel(LnReduceP p) {
    __shared__ float red[2][4];
    const int m = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int C = p.C, c = t * 4;
    const bool ok = c < C;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), gv = a, bv = a;
    if (ok) {
        const float* __restrict__ base = p.parts + (long long)m * C + c;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), rq = bq;
        if (p.bias) bq = *reinterpret_cast<const float4*>(p.bias + c);
        if (p.R) rq = *reinterpret_cast<const float4*>(p.R + (long long)m * p.ldr + c);
        gv = *reinterpret_cast<const float4*>(p.gamma + c);
        bv = *reinterpret_cast<const float4*>(p.beta + c);
        for (int g0 = 0; g0 < p.S; g0 += 8) {
            float4 q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (g0 + j < p.S) q[j] = *reinterpret_cast<const float4*>(base + (long long)(g0 + j) * p.pstride);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (g0 + j < p.S) { a.x += q[j].x; a.y += q[j].y; a.z += q[j].z; a.w += q[j].w; }
        }
        if (p.bias) { a.x += bq.x; a.y += bq.y; a.z += bq.z; a.w += bq.w; }
        if (p.R) { a.x += rq.x; a.y += rq.y; a.z += rq.z; a.w += rq.w; }
        if (p.xout) *reinterpret_cast<float4*>(p.xout + (long long)m * p.ldx + c) = a;
    }
    const float inv_c = 1.0f / (float)C;
    float s = wave_sum(ok ? (a.x + a.y) + (a.z + a.w) : 0.0f);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv_c;
    a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean;
    float q2 = wave_sum(ok ? (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w) : 0.0f);
    if (lane == 0) red[1][wave] = q2;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv_c + p.eps);
    if (ok) {
        float4 y;
        y.x = a.x * rstd * gv.x + bv.x;
        y.y = a.y * rstd * gv.y + bv.y;
        y.z = a.z * rstd * gv.z + bv.z;
        y.w = a.w * rstd * gv.w + bv.w;
        if (p.h_planes) { if (store_planes4(p.hout + (long long)m * p.ldh, c, y) >= 65504.0f && p.x3h_flag) atomicOr(p.x3h_flag, 1); }
        else *reinterpret_cast<float4*>(p.hout + (long long)m * p.ldh + c) = y;
    }
}
hipError_t launch_ln_reduce(const LnReduceP& p, hipStream_t s) {
    if (p.M <= 0) return hipSuccess;
    if (p.C > 1024 || (p.C & 3) || (p.ldx & 3) || (p.ldh & 3) || (p.ldr & 3) || p.S < 1) return hipErrorInvalidValue;
    if (p.h_planes && (p.M > 4096 || (p.C & 31))) return hipErrorInvalidValue;
    if (p.M <= 4096)      // every AR step: one row per workgroup
        hipLaunchKernelGGL(ln_reduce_row_kernel, dim3(p.M), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(ln_reduce_kernel, dim3((p.M + 3) / 4), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_layernorm(const LnP& p, hipStream_t s) {
    if (p.M <= 0) return hipSuccess;
    if (p.C > 1024 || (p.C & 3) || (p.ldx & 3) || (p.ldo & 3)) return hipErrorInvalidValue;
    if (p.out_planes && ((p.C & 31) || p.R1 || p.R2)) return hipErrorInvalidValue;      // (planes of act(LN(x)): the activation is applied first)
    hipLaunchKernelGGL(layernorm_kernel, dim3((p.M + 3) / 4), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// generic "one float4 per thread" row kernels: thread -> (row r, column chunk c4)

#define MT2_ROW_LOOP(R_, C_)                                                              \
    const long long total_ = (long long)(R_) * ((C_) >> 2);                               \
    for (long long i_ = (long long)blockIdx.x * blockDim.x + threadIdx.x; i_ < total_;    \
         i_ += (long long)gridDim.x * blockDim.x)

static inline dim3 row_grid(long long work) {
    long long blocks = (work + 255) / 256;
    if (blocks > 2048 * 8) blocks = 2048 * 8;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

// Caller-supplied ids are clamped into the table by every gather (the range check that turns a bad id into an error
// runs beside the call, model_stages.hip ids_check): an out-of-range id never becomes an out-of-bounds read.
__device__ __forceinline__ long long clamp_id(long long v, int hi) { return v < 0 ? 0 : (v >= hi ? hi - 1 : v); }

// TokenEmbedding + SinePositionalEmbedding (modules/embedding.py:43-47,94-98; mrte.py:159-160)
__global__ void embed_pe_kernel(const float* table, int C, const int64_t* ids, const int* idmap, const int* pos,
                                const float* pe, float* out, int ldo, int R, int vocab) {
    const int c4n = C >> 2;
    MT2_ROW_LOOP(R, C) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int im = idmap[r];
        if (im >= 0) {
            const float4 e = *reinterpret_cast<const float4*>(table + clamp_id(ids[im], vocab) * C + c);
            const float4 q = *reinterpret_cast<const float4*>(pe + (long long)pos[r] * C + c);
            v.x = e.x * 1.0f + q.x; v.y = e.y * 1.0f + q.y; v.z = e.z * 1.0f + q.z; v.w = e.w * 1.0f + q.w;
        }
        *reinterpret_cast<float4*>(out + (long long)r * ldo + c) = v;
    }
}
hipError_t launch_embed_pe(const float* table, int C, const int64_t* ids, const int* idmap, const int* pos,
                           const float* pe, float* out, int ldo, int R, int vocab, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_pe_kernel, row_grid((long long)R * (C >> 2)), dim3(256), 0, s, table, C, ids, idmap,
                       pos, pe, out, ldo, R, vocab);
    return hipGetLastError();
}

// LengthRegulator (modules/mrte.py:42-60) as a gather: alignment @ x with a 0/1 alignment matrix is
// exactly "copy row map[r]" (SURVEY M4), so no matmul and no host round trip.
__global__ void gather_rows_kernel(const float* src, int lds_, const int* map, float* out, int ldo, int C, int R) {
    const int c4n = C >> 2;
    MT2_ROW_LOOP(R, C) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        const int sr = map[r];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sr >= 0) v = *reinterpret_cast<const float4*>(src + (long long)sr * lds_ + c);
        *reinterpret_cast<float4*>(out + (long long)r * ldo + c) = v;
    }
}
__global__ void gather_rows_scalar_kernel(const float* src, int lds_, const int* map, float* out, int ldo, int C,
                                          int R) {
    const long long total = (long long)R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        const int sr = map[r];
        out[(long long)r * ldo + c] = sr >= 0 ? src[(long long)sr * lds_ + c] : 0.0f;
    }
}
hipError_t launch_gather_rows(const float* src, int lds_, const int* map, float* out, int ldo, int C, int R,
                              hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if ((C & 3) || (lds_ & 3) || (ldo & 3)) {   // odd widths: scalar path
        hipLaunchKernelGGL(gather_rows_scalar_kernel, row_grid((long long)R * C), dim3(256), 0, s, src, lds_, map, out,
                           ldo, C, R);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gather_rows_kernel, row_grid((long long)R * (C >> 2)), dim3(256), 0, s, src, lds_, map, out,
                       ldo, C, R);
    return hipGetLastError();
}

// F.max_pool1d(k, stride k, ceil_mode=True) along time (models/megatts2.py:357-358, vqpe.py:38):
// window r covers source rows [first[r], first[r] + cnt[r]) - the last one of an utterance is partial.
__global__ void pool_max_kernel(const float* src, int lds_, const int* first, const int* cnt, float* out, int ldo,
                                int C, int R) {
    const int c4n = C >> 2;
    MT2_ROW_LOOP(R, C) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        const int n = cnt[r];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n > 0) {
            const float* sp = src + (long long)first[r] * lds_ + c;
            v = *reinterpret_cast<const float4*>(sp);
            for (int i = 1; i < n; ++i) {
                const float4 w = *reinterpret_cast<const float4*>(sp + (long long)i * lds_);
                v.x = fmaxf(v.x, w.x); v.y = fmaxf(v.y, w.y); v.z = fmaxf(v.z, w.z); v.w = fmaxf(v.w, w.w);
            }
        }
        *reinterpret_cast<float4*>(out + (long long)r * ldo + c) = v;
    }
}
hipError_t launch_pool_max(const float* src, int lds_, const int* first, const int* cnt, float* out, int ldo,
                           int C, int R, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(pool_max_kernel, row_grid((long long)R * (C >> 2)), dim3(256), 0, s, src, lds_, first, cnt,
                       out, ldo, C, R);
    return hipGetLastError();
}

// ConvNetDouble.forward branch sum (modules/convnet.py:205-207), left-to-right like the reference
__global__ void sum_groups_kernel(const float* x, long long strideG, int groups, int ld, float* out, int ldo,
                                  int C, int R) {
    const int c4n = C >> 2;
    MT2_ROW_LOOP(R, C) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        const float* xp = x + (long long)r * ld + c;
        float4 v = *reinterpret_cast<const float4*>(xp);
        for (int g = 1; g < groups; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(xp + g * strideG);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *reinterpret_cast<float4*>(out + (long long)r * ldo + c) = v;
    }
}
hipError_t launch_sum_groups(const float* x, long long strideG, int groups, int ld, float* out, int ldo, int C,
                             int R, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(sum_groups_kernel, row_grid((long long)R * (C >> 2)), dim3(256), 0, s, x, strideG, groups,
                       ld, out, ldo, C, R);
    return hipGetLastError();
}

// HiFi-GAN multi-receptive-field fusion: (rb0 + rb1 + rb2) / 3
__global__ void avg3_kernel(const float* a, const float* b, const float* d, float scale, float* out, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4*>(a)[i];
        const float4 y = reinterpret_cast<const float4*>(b)[i];
        const float4 z = reinterpret_cast<const float4*>(d)[i];
        float4 v;
        v.x = ((x.x + y.x) + z.x) * scale; v.y = ((x.y + y.y) + z.y) * scale;
        v.z = ((x.z + y.z) + z.z) * scale; v.w = ((x.w + y.w) + z.w) * scale;
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
hipError_t launch_avg3(const float* a, const float* b, const float* d, float scale, float* out, long long n,
                       hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avg3_kernel, row_grid(n >> 2), dim3(256), 0, s, a, b, d, scale, out, n >> 2);
    return hipGetLastError();
}

// HiFi-GAN output layer: the last stage's MRF mean, F.leaky_relu, conv_post (Cout = 1, k taps) and tanh in ONE pass
// (speechbrain HifiganGenerator.forward tail; models/megatts2.py:370).  One output channel is a dot product per row -
// VALU work bound by the read of the three resblock outputs, not a GEMM: a 256-row block stages its
// (256 + k - 1) x ch window of lrelu(mean) in LDS (row stride ch + 1: conflict-free), every thread owns one row.
// The mean keeps avg3_kernel's operation order, so the rows the convolution sees are the same values as before.
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                         const float* __restrict__ x2, float scale, long long R, int ch,
                                                         int k, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float slope, const int* __restrict__ valid, float* __restrict__ out) {
    extern __shared__ float sm_post[];
    const int ld = ch + 1, half = (k - 1) / 2, nrow = 256 + k - 1, c4 = ch >> 2;
    float* xs = sm_post;
    float* ws = sm_post + nrow * ld;
    const long long r0 = (long long)blockIdx.x * 256 - half;
    for (int i = threadIdx.x; i < nrow * c4; i += 256) {
        const int rr = i / c4, cq = i - rr * c4;
        const long long g = r0 + rr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g >= 0 && g < R) {
            v = *reinterpret_cast<const float4*>(x0 + g * ch + cq * 4);
            if (x1) {
                const float4 y = *reinterpret_cast<const float4*>(x1 + g * ch + cq * 4);
                const float4 z = *reinterpret_cast<const float4*>(x2 + g * ch + cq * 4);
                v.x = ((v.x + y.x) + z.x) * scale; v.y = ((v.y + y.y) + z.y) * scale;
                v.z = ((v.z + y.z) + z.z) * scale; v.w = ((v.w + y.w) + z.w) * scale;
            }
        }
        float* d = xs + rr * ld + cq * 4;
        d[0] = v.x > 0.f ? v.x : v.x * slope; d[1] = v.y > 0.f ? v.y : v.y * slope;
        d[2] = v.z > 0.f ? v.z : v.z * slope; d[3] = v.w > 0.f ? v.w : v.w * slope;
    }
    for (int i = threadIdx.x; i < k * ch; i += 256) ws[i] = w[i];
    __syncthreads();
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= R) return;
    float acc = 0.0f;
    for (int t = 0; t < k; ++t) {
        const float* xr = xs + (threadIdx.x + t) * ld;
        const float* wr = ws + t * ch;
#pragma unroll 8
        for (int cidx = 0; cidx < ch; ++cidx) acc = fmaf(xr[cidx], wr[cidx], acc);
    }
    acc += bias[0];
    out[m] = (valid && !valid[m]) ? 0.0f : tanhf(acc);
}
hipError_t launch_conv_post(const float* x0, const float* x1, const float* x2, float scale, long long R, int ch, int k,
                            const float* w, const float* bias, float slope, const int* valid, float* out, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if ((ch & 3) || ch > 128 || k > 15 || !(k & 1)) return hipErrorInvalidValue;
    const size_t lds = ((size_t)(256 + k - 1) * (ch + 1) + (size_t)k * ch) * sizeof(float);
    // the default dynamic-LDS limit is 64 KiB and this kernel does not raise it: wider last stages (a checkpoint with
    // upsample_initial_channel 1024 / 2048 -> ch 64 / 128) take the avg3 + conv_same path instead (hifigan_rows)
    if (lds > 64 * 1024) return hipErrorNotSupported;
    hipLaunchKernelGGL(conv_post_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), lds, s, x0, x1, x2, scale, R, ch, k, w,
                       bias, slope, valid, out);
    return hipGetLastError();
}

// Reflect halo of every utterance (torch F.pad(mode="reflect"), what speechbrain's Conv1d(padding="same") applies):
//   x[off - j] = x[off + j],  x[off + L - 1 + j] = x[off + L - 1 - j],  j = 1..G   (rows; off = start * scale, L = len * scale)
// written into the gap rows in front of / behind the utterance (the gap is >= 2 G, checked by the caller).  A conv kernel
// that follows reads them as its padding; its own epilogue zeroes the gap rows of its output again (row mask).
__global__ void fill_reflect_kernel(float* x, int ld, int C, const int* start, const int* len, int B, long long scale, int G) {
    const int c4n = C >> 2;
    const long long total = (long long)B * 2 * G * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long long r = i / c4n;
        const int j = (int)(r % G) + 1, side = (int)((r / G) & 1), b = (int)(r / (2 * G));
        const long long off = (long long)start[b] * scale, L = (long long)len[b] * scale;
        if (j >= L) continue;                                         // torch requires pad < length; nothing to mirror
        const long long dst = side == 0 ? off - j : off + L - 1 + j, src = side == 0 ? off + j : off + L - 1 - j;
        *reinterpret_cast<float4*>(x + dst * ld + c) = *reinterpret_cast<const float4*>(x + src * ld + c);
    }
}
hipError_t launch_fill_reflect(float* x, int ld, int C, const int* start, const int* len, int B, long long scale, int G,
                               hipStream_t s) {
    if (B <= 0 || G <= 0) return hipSuccess;
    if (C & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fill_reflect_kernel, row_grid((long long)B * 2 * G * (C >> 2)), dim3(256), 0, s, x, ld, C, start, len,
                       B, scale, G);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// boundary layout conversion: reference tensors are padded batch-first [B, Tmax, C] (or [B, C, Tmax]
// for the mel decoder / vocoder, "B D T"); internal rows are packed with gap rows.

__global__ void pack_rows_kernel(const float* src, int C, int Tmax, int cmajor, const int* rowmap, float* dst,
                                 int ldd, int R) {
    const long long total = (long long)R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        const int bt = rowmap[r];
        float v = 0.0f;
        if (bt >= 0) {
            if (cmajor) {
                const int b = bt / Tmax, t = bt % Tmax;
                v = src[((long long)b * C + c) * Tmax + t];
            } else {
                v = src[(long long)bt * C + c];
            }
        }
        dst[(long long)r * ldd + c] = v;
    }
}
hipError_t launch_pack_rows(const float* src, int C, int Tmax, int cmajor, const int* rowmap, float* dst, int ldd,
                            int R, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_rows_kernel, row_grid((long long)R * C), dim3(256), 0, s, src, C, Tmax, cmajor, rowmap,
                       dst, ldd, R);
    return hipGetLastError();
}

__global__ void unpack_rows_kernel(const float* src, int lds_, int C, int Tmax, int cmajor, const int* rowmap,
                                   float* dst, int R) {
    const long long total = (long long)R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int r, c;
        if (cmajor) { c = (int)(i / R); r = (int)(i % R); }   // consecutive threads -> consecutive t
        else { r = (int)(i / C); c = (int)(i % C); }
        const int bt = rowmap[r];
        if (bt < 0) continue;
        const float v = src[(long long)r * lds_ + c];
        if (cmajor) {
            const int b = bt / Tmax, t = bt % Tmax;
            dst[((long long)b * C + c) * Tmax + t] = v;
        } else {
            dst[(long long)bt * C + c] = v;
        }
    }
}
hipError_t launch_unpack_rows(const float* src, int lds_, int C, int Tmax, int cmajor, const int* rowmap,
                              float* dst, int R, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_rows_kernel, row_grid((long long)R * C), dim3(256), 0, s, src, lds_, C, Tmax, cmajor,
                       rowmap, dst, R);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// autoregressive step assembly (all active sequences have the same length n = t + 1 at step t)

// MegaADM.infer step input (models/megatts2.py:265-269): cat([tc_emb, dt_linear_emb(p)]) + pe
__global__ void adm_step_input_kernel(const float* tc_emb, int ld_tc, const int* tc_row, const float* w_dt,
                                      const float* p, int pstride, const float* pe, float* x, int Dc, int De,
                                      int n, int A) {
    const int D = Dc + De, c4n = D >> 2;
    MT2_ROW_LOOP(A * n, D) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        const int j = r / n, i = r % n;
        float4 v;
        if (c < Dc) {
            v = *reinterpret_cast<const float4*>(tc_emb + (long long)(tc_row[j] + i) * ld_tc + c);
        } else {
            const float pv = p[(long long)j * pstride + i];
            const float4 w = *reinterpret_cast<const float4*>(w_dt + (c - Dc));
            v.x = w.x * pv; v.y = w.y * pv; v.z = w.z * pv; v.w = w.w * pv;
        }
        const float4 q = *reinterpret_cast<const float4*>(pe + (long long)i * D + c);
        v.x = v.x * 1.0f + q.x; v.y = v.y * 1.0f + q.y; v.z = v.z * 1.0f + q.z; v.w = v.w * 1.0f + q.w;
        *reinterpret_cast<float4*>(x + (long long)r * D + c) = v;
    }
}
hipError_t launch_adm_step_input(const float* tc_emb, int ld_tc, const int* tc_row, const float* w_dt,
                                 const float* p, int pstride, const float* pe, float* x, int Dc, int De,
                                 int n, int A, hipStream_t s) {
    if (A <= 0) return hipSuccess;
    hipLaunchKernelGGL(adm_step_input_kernel, row_grid((long long)A * n * ((Dc + De) >> 2)), dim3(256), 0, s,
                       tc_emb, ld_tc, tc_row, w_dt, p, pstride, pe, x, Dc, De, n, A);
    return hipGetLastError();
}

// MegaPLM.infer step input (models/megatts2.py:173-175): cat([cond[:t+1], pc_embedding(codes)]) + pe
__global__ void plm_step_input_kernel(const float* cond, int ld_c, const int* cond_row, const float* emb,
                                      const int64_t* codes, int cstride, const float* pe, float* x, int Dc, int De,
                                      int n, int A, int emb_rows) {
    const int D = Dc + De, c4n = D >> 2;
    MT2_ROW_LOOP(A * n, D) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        const int j = r / n, i = r % n;
        float4 v;
        if (c < Dc) {
            v = *reinterpret_cast<const float4*>(cond + (long long)(cond_row[j] + i) * ld_c + c);
        } else {
            v = *reinterpret_cast<const float4*>(emb + clamp_id(codes[(long long)j * cstride + i], emb_rows) * De + (c - Dc));
        }
        const float4 q = *reinterpret_cast<const float4*>(pe + (long long)i * D + c);
        v.x = v.x * 1.0f + q.x; v.y = v.y * 1.0f + q.y; v.z = v.z * 1.0f + q.z; v.w = v.w * 1.0f + q.w;
        *reinterpret_cast<float4*>(x + (long long)r * D + c) = v;
    }
}
hipError_t launch_plm_step_input(const float* cond, int ld_c, const int* cond_row, const float* emb,
                                 const int64_t* codes, int cstride, const float* pe, float* x, int Dc, int De,
                                 int n, int A, int emb_rows, hipStream_t s) {
    if (A <= 0) return hipSuccess;
    hipLaunchKernelGGL(plm_step_input_kernel, row_grid((long long)A * n * ((Dc + De) >> 2)), dim3(256), 0, s, cond,
                       ld_c, cond_row, emb, codes, cstride, pe, x, Dc, De, n, A, emb_rows);
    return hipGetLastError();
}

// MegaADM predict_layer on the last position only (models/megatts2.py:272): one wave per sequence
__global__ __launch_bounds__(256) void adm_predict_kernel(const float* x, int D, const float* w, float* p,
                                                          int pstride, int n, int xn, int A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= A) return;
    const float* xr = x + ((long long)j * xn + (xn - 1)) * D;
    float s = 0.0f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(xr + c);
        const float4 b = *reinterpret_cast<const float4*>(w + c);
        s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    s = wave_sum(s);
    if (lane == 0) p[(long long)j * pstride + n] = s;
}
hipError_t launch_adm_predict(const float* x, int D, const float* w, float* p, int pstride, int n, int xn, int A,
                              hipStream_t s) {
    if (A <= 0) return hipSuccess;
    hipLaunchKernelGGL(adm_predict_kernel, dim3((A + 3) / 4), dim3(256), 0, s, x, D, w, p, pstride, n, xn, A);
    return hipGetLastError();
}

// (p_code[:, 1:] + 0.5).to(int32).clamp(1, 128)  (models/megatts2.py:275); .to(int32) truncates
// slot j (length-sorted order) is written back to utterance slot_b[j].
__global__ void adm_finalize_kernel(const float* p, int pstride, const int* lens, const int* slot_b, int32_t* dur,
                                    float* flt, int dstride, int A, int nmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * nmax) return;
    const int j = i / nmax, t = i % nmax;
    int d = 0;
    float f = 0.0f;
    if (t < lens[j]) {
        f = p[(long long)j * pstride + t + 1];
        d = (int)(f + 0.5f);
        d = d < 1 ? 1 : (d > 128 ? 128 : d);
    }
    const int b = slot_b ? slot_b[j] : j;
    dur[(long long)b * dstride + t] = d;
    if (flt) flt[(long long)b * dstride + t] = f;
}
hipError_t launch_adm_finalize(const float* p, int pstride, const int* lens, const int* slot_b, int32_t* dur,
                               float* flt, int dstride, int A, int nmax, hipStream_t s) {
    if (A * nmax <= 0) return hipSuccess;
    hipLaunchKernelGGL(adm_finalize_kernel, dim3((A * nmax + 255) / 256), dim3(256), 0, s, p, pstride, lens, slot_b,
                       dur, flt, dstride, A, nmax);
    return hipGetLastError();
}

// codes_out[slot_b[j]*ostride + t] = codes[j*cstride + 1 + skip + t] for t < lens[j] - skip, else 0
__global__ void plm_finalize_kernel(const int64_t* codes, int cstride, const int* lens, const int* slot_b,
                                    int64_t* out, int ostride, int A, int nmax, int skip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * nmax) return;
    const int j = i / nmax, t = i % nmax;
    const int b = slot_b ? slot_b[j] : j;
    out[(long long)b * ostride + t] = t < lens[j] - skip ? codes[(long long)j * cstride + 1 + skip + t] : 0;
}
hipError_t launch_plm_finalize(const int64_t* codes, int cstride, const int* lens, const int* slot_b, int64_t* out,
                               int ostride, int A, int nmax, int skip, hipStream_t s) {
    if (A * nmax <= 0) return hipSuccess;
    hipLaunchKernelGGL(plm_finalize_kernel, dim3((A * nmax + 255) / 256), dim3(256), 0, s, codes, cstride, lens,
                       slot_b, out, ostride, A, nmax, skip);
    return hipGetLastError();
}

// AR history initialisation (see mt2_kernels.h)
__global__ void adm_init_hist_kernel(float* p, int pstride, const float* prefix, int P, const int* slot_b, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * pstride) return;
    const int j = i / pstride, t = i % pstride;
    float v = 0.0f;
    if (t >= 1 && t <= P) v = prefix[(long long)(slot_b ? slot_b[j] : j) * P + (t - 1)];
    p[i] = v;
}
hipError_t launch_adm_init_hist(float* p, int pstride, const float* prefix, int P, const int* slot_b, int A,
                                hipStream_t s) {
    if (A * pstride <= 0) return hipSuccess;
    hipLaunchKernelGGL(adm_init_hist_kernel, dim3((A * pstride + 255) / 256), dim3(256), 0, s, p, pstride, prefix, P,
                       slot_b, A);
    return hipGetLastError();
}
__global__ void plm_init_hist_kernel(int64_t* codes, int cstride, int64_t bos, const int64_t* prefix, int P, int pstride,
                                     const int* slot_b, int A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * cstride) return;
    const int j = i / cstride, t = i % cstride;
    int64_t v = 0;
    if (t == 0) v = bos;
    else if (t <= P) v = prefix[(long long)(slot_b ? slot_b[j] : j) * pstride + (t - 1)];
    codes[i] = v;
}
hipError_t launch_plm_init_hist(int64_t* codes, int cstride, int64_t bos, const int64_t* prefix, int P, int pstride,
                                const int* slot_b, int A, hipStream_t s) {
    if (A * cstride <= 0) return hipSuccess;
    hipLaunchKernelGGL(plm_init_hist_kernel, dim3((A * cstride + 255) / 256), dim3(256), 0, s, codes, cstride, bos,
                       prefix, P, pstride, slot_b, A);
    return hipGetLastError();
}

// Range check of caller-supplied ids before they are used as gather indices (nn.Embedding / F.embedding raise
// IndexError in the reference, modules/embedding.py:43-47, core_vq.py:188-190): flag |= bit on any id outside [0, hi).
__global__ void check_ids_kernel(const int64_t* ids, const int* map, int R, long long hi, int* flag, int bit) {
    bool bad = false;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (long long)gridDim.x * blockDim.x) {
        const int src = map ? map[r] : (int)r;
        if (src < 0) continue;
        const long long v = ids[src];
        bad |= v < 0 || v >= hi;
    }
    if (bad) atomicOr(flag, bit);
}
hipError_t launch_check_ids(const int64_t* ids, const int* map, int R, long long hi, int* flag, int bit,
                            hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(check_ids_kernel, row_grid(R), dim3(256), 0, s, ids, map, R, hi, flag, bit);
    return hipGetLastError();
}

__global__ void copy_2d_kernel(const float* src, long long spitch, float* dst, long long dpitch, long long w4,
                               int rows) {
    const long long total = w4 * rows;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long j = i / w4, c = (i % w4) * 4;
        *reinterpret_cast<float4*>(dst + j * dpitch + c) = *reinterpret_cast<const float4*>(src + j * spitch + c);
    }
}
hipError_t launch_copy_2d(const float* src, long long spitch, float* dst, long long dpitch, long long width, int rows,
                          hipStream_t s) {
    if (rows <= 0 || width <= 0) return hipSuccess;
    if ((width & 3) || (spitch & 3) || (dpitch & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_2d_kernel, row_grid((width >> 2) * rows), dim3(256), 0, s, src, spitch, dst, dpitch,
                       width >> 2, rows);
    return hipGetLastError();
}

// out[map[r]] = src[r] for map[r] >= 0 (int64 row scatter: VQ indices back to the padded layout)
__global__ void scatter_i64_kernel(const int64_t* src, const int* map, int64_t* out, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R && map[r] >= 0) out[map[r]] = src[r];
}
hipError_t launch_scatter_i64(const int64_t* src, const int* map, int64_t* out, int R, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_i64_kernel, dim3((R + 255) / 256), dim3(256), 0, s, src, map, out, R);
    return hipGetLastError();
}

// gap-row mask of an up-sampled row set: out[r] = in[r / factor]
__global__ void expand_mask_kernel(const int* in, int factor, int* out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = in[i / factor];
}
hipError_t launch_expand_mask(const int* in, int factor, int* out, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(expand_mask_kernel, row_grid(n), dim3(256), 0, s, in, factor, out, n);
    return hipGetLastError();
}

// waveform rows (1 channel, already tanh'ed) -> padded [B, hop*T_max]; utterance b = rows [start[b], start[b]+len[b])
__global__ void unpack_wav_kernel(const float* src, const long long* start, const long long* len, float* out,
                                  long long out_stride) {
    const int b = blockIdx.y;
    const long long n = len[b], s0 = start[b];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[b * out_stride + i] = src[s0 + i];
}
hipError_t launch_unpack_wav(const float* src, const long long* start, const long long* len, float* out,
                             long long out_stride, long long max_len, int B, hipStream_t s) {
    if (B <= 0 || max_len <= 0) return hipSuccess;
    long long bx = (max_len + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(unpack_wav_kernel, dim3((unsigned)bx, B), dim3(256), 0, s, src, start, len, out, out_stride);
    return hipGetLastError();
}

// torch.argmax semantics: first index of the maximum.  One wave per row, (value, index) reduction.
__device__ __forceinline__ void argmax_combine(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, int ldx, int N, int64_t* out,
                                                          int ostride, int ooff, int A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= A) return;
    const float* xr = x + (long long)j * ldx;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (((N | ldx) & 3) == 0) {      // four float4 loads in flight per lane (a serial scalar loop: one round trip per 64 logits)
        for (int n0 = lane * 4; n0 < N; n0 += 1024) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (n0 + u * 256 < N) q[u] = *reinterpret_cast<const float4*>(xr + n0 + u * 256);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (n0 + u * 256 < N) {
                    const int n = n0 + u * 256;
                    argmax_combine(bv, bi, q[u].x, n); argmax_combine(bv, bi, q[u].y, n + 1);
                    argmax_combine(bv, bi, q[u].z, n + 2); argmax_combine(bv, bi, q[u].w, n + 3);
                }
        }
    } else {
        for (int n = lane; n < N; n += 64) argmax_combine(bv, bi, xr[n], n);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        argmax_combine(bv, bi, ov, oi);
    }
    if (lane == 0) out[(long long)j * ostride + ooff] = bi;
}
hipError_t launch_argmax_rows(const float* x, int ldx, int N, int64_t* out, int ostride, int ooff, int A,
                              hipStream_t s) {
    if (A <= 0) return hipSuccess;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((A + 3) / 4), dim3(256), 0, s, x, ldx, N, out, ostride, ooff, A);
    return hipGetLastError();
}

// EuclideanCodebook.quantize (modules/quantization/core_vq.py:175-183):
//   dist = -(x.pow(2).sum(1) - 2 * x @ embed + embed.pow(2).sum(0)); ind = dist.max(-1).indices
// evaluated in that operand order, ((xx - 2*xe) + ee), lowest index on ties.  xe comes from the GEMM.
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* x, int ldx, int D, const float* xe, int ldxe,
                                                        const float* ee, int N, const int* valid, int64_t* idx,
                                                        int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    if (valid && valid[m] == 0) { if (lane == 0) idx[m] = 0; return; }
    const float* xr = x + (long long)m * ldx;
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
    const float xx = wave_sum(s);
    const float* er = xe + (long long)m * ldxe;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = lane; n < N; n += 64) {
        const float dist = -((xx - 2.0f * er[n]) + ee[n]);
        argmax_combine(bv, bi, dist, n);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        argmax_combine(bv, bi, ov, oi);
    }
    if (lane == 0) idx[m] = bi;
}
hipError_t launch_vq_argmin(const float* x, int ldx, int D, const float* xe, int ldxe, const float* ee, int N,
                            const int* valid, int64_t* idx, int M, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, D, xe, ldxe, ee, N, valid, idx,
                       M);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* E, int D, float* ee, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float s = 0.0f;
    for (int c = lane; c < D; c += 64) s += E[(long long)n * D + c] * E[(long long)n * D + c];
    s = wave_sum(s);
    if (lane == 0) ee[n] = s;
}
hipError_t launch_row_sqnorm(const float* E, int D, float* ee, int N, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((N + 3) / 4), dim3(256), 0, s, E, D, ee, N);
    return hipGetLastError();
}

// decoder input rows (models/megatts2.py:361-366): [tc_latent_expand (gather), zq (codebook row, x8 repeat)]
__global__ void decoder_input_kernel(const float* tc, int ld_tc, const int* tcmap, const float* E,
                                     const int64_t* codes, const int* codemap, float* out, int Dc, int Dq, int R,
                                     int bins) {
    const int D = Dc + Dq, c4n = D >> 2;
    MT2_ROW_LOOP(R, D) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int tr = tcmap[r];
        if (tr >= 0) {
            if (c < Dc) v = *reinterpret_cast<const float4*>(tc + (long long)tr * ld_tc + c);
            else v = *reinterpret_cast<const float4*>(E + clamp_id(codes[codemap[r]], bins) * Dq + (c - Dc));
        }
        *reinterpret_cast<float4*>(out + (long long)r * D + c) = v;
    }
}
hipError_t launch_decoder_input(const float* tc, int ld_tc, const int* tcmap, const float* E, const int64_t* codes,
                                const int* codemap, float* out, int Dc, int Dq, int R, int bins, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(decoder_input_kernel, row_grid((long long)R * ((Dc + Dq) >> 2)), dim3(256), 0, s, tc, ld_tc,
                       tcmap, E, codes, codemap, out, Dc, Dq, R, bins);
    return hipGetLastError();
}

// EuclideanCodebook.dequantize (core_vq.py:188-190) + the x8 repeat of vqpe.py:59-61 via codemap
__global__ void codebook_rows_kernel(const float* E, const int64_t* codes, const int* codemap, float* out, int ldo,
                                     int Dq, int R, int bins) {
    const int c4n = Dq >> 2;
    MT2_ROW_LOOP(R, Dq) {
        const int r = (int)(i_ / c4n), c = (int)(i_ % c4n) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int cm = codemap[r];
        if (cm >= 0) v = *reinterpret_cast<const float4*>(E + clamp_id(codes[cm], bins) * Dq + c);
        *reinterpret_cast<float4*>(out + (long long)r * ldo + c) = v;
    }
}
hipError_t launch_codebook_rows(const float* E, const int64_t* codes, const int* codemap, float* out, int ldo,
                                int Dq, int R, int bins, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(codebook_rows_kernel, row_grid((long long)R * (Dq >> 2)), dim3(256), 0, s, E, codes, codemap,
                       out, ldo, Dq, R, bins);
    return hipGetLastError();
}

__global__ void tanh_col_kernel(const float* x, int ldx, float* out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = tanhf(x[i * ldx]);
}
hipError_t launch_tanh_col(const float* x, int ldx, float* out, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tanh_col_kernel, row_grid(n), dim3(256), 0, s, x, ldx, out, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// mel front-end (reference modules/tokenizer.py:107-125 -> torchaudio Spectrogram, center=True, reflect)

// Reflect-padded waveform laid out as hop-sized blocks: row r belongs to utterance blk_b[r], block blk_t[r];
// sample (r, c) = wav[b, reflect(blk_t*hop + c - pad)], zero beyond the padded signal.
__global__ void reflect_pad_blocks_kernel(const float* wav, long long wstride, const int* blk_b, const int* blk_t,
                                          const int* len, int hop, int pad, float* out, int R) {
    const long long total = (long long)R * hop;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / hop), c = (int)(i % hop);
        const int b = blk_b[r], L = len[b];
        long long j = (long long)blk_t[r] * hop + c - pad;
        float v = 0.0f;
        if (j < (long long)L + pad) {
            if (j < 0) j = -j;
            else if (j >= L) j = 2ll * (L - 1) - j;
            v = wav[(long long)b * wstride + j];
        }
        out[i] = v;
    }
}
hipError_t launch_reflect_pad_blocks(const float* wav, long long wstride, const int* blk_b, const int* blk_t,
                                     const int* len, int hop, int pad, float* out, int R, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    hipLaunchKernelGGL(reflect_pad_blocks_kernel, row_grid((long long)R * hop), dim3(256), 0, s, wav, wstride, blk_b,
                       blk_t, len, hop, pad, out, R);
    return hipGetLastError();
}

// |X|: spec rows hold [re(0..F-1) | im(0..F-1)]; out[m, f] = sqrt(re^2 + im^2), columns F..ldo-1 zeroed
__global__ void magnitude_kernel(const float* spec, int lds_, int F, float* out, int ldo, int M) {
    const long long total = (long long)M * ldo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / ldo), f = (int)(i % ldo);
        float v = 0.0f;
        if (f < F) {
            const float re = spec[(long long)m * lds_ + f], im = spec[(long long)m * lds_ + F + f];
            v = sqrtf(re * re + im * im);
        }
        out[i] = v;
    }
}
hipError_t launch_magnitude(const float* spec, int lds_, int F, float* out, int ldo, int M, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(magnitude_kernel, row_grid((long long)M * ldo), dim3(256), 0, s, spec, lds_, F, out, ldo, M);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// stage boundary markers for kernel traces (measurement only; EngineOpts::markers)
template <int ID> __global__ void stage_marker_kernel() {}
hipError_t launch_stage_marker(int id, hipStream_t s) {
    switch (id & 15) {
#define MT2_MARK(I) case I: hipLaunchKernelGGL(stage_marker_kernel<I>, dim3(1), dim3(64), 0, s); break;
        MT2_MARK(0) MT2_MARK(1) MT2_MARK(2) MT2_MARK(3) MT2_MARK(4) MT2_MARK(5) MT2_MARK(6) MT2_MARK(7)
        MT2_MARK(8) MT2_MARK(9) MT2_MARK(10) MT2_MARK(11) MT2_MARK(12) MT2_MARK(13) MT2_MARK(14) MT2_MARK(15)
#undef MT2_MARK
    }
    return hipGetLastError();
}

}  // namespace mt2
