// Stage drivers of libmegatts2_hip: they plan row sets on the host, upload the (small) integer plans
// with one copy per call and enqueue the gfx950 kernels of mt2_kernels.h on the caller's stream.
// Each driver cites the reference function it replaces; batch semantics are "B independent batch-1
// runs" (SURVEY.md N1): per-utterance zero gaps for convolutions, per-utterance attention ranges,
// positional indices restarting at 0.
#include "mt2_model.h"
#include "x3h_planes.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace mt2 {

// Tuning switches (split-K through the LayerNorm, vocoder stream count, tile thresholds, ...) live in the handle:
// mt2_model::opts (mt2_kernels.h EngineOpts), set through mt2_set_option.

// ---------------------------------------------------------------------------------------------------
// planning helpers

static RowSet make_rows(const int* lens, int B, int G) {
    RowSet rs;
    rs.B = B; rs.G = G;
    rs.len.assign(lens, lens + B);
    rs.off.resize(B);
    int r = G;
    for (int b = 0; b < B; ++b) {
        MT2_REQUIRE(lens[b] >= 0, "negative utterance length");
        rs.off[b] = r;
        r += lens[b] + G;
        rs.maxlen = std::max(rs.maxlen, lens[b]);
    }
    rs.R = r;
    return rs;
}

struct RowPlanOffsets { int valid, start, len; };
static RowPlanOffsets plan_rows(IntPlan& ip, const RowSet& rs) {
    std::vector<int> valid(rs.R, 0);
    for (int b = 0; b < rs.B; ++b) std::fill(valid.begin() + rs.off[b], valid.begin() + rs.off[b] + rs.len[b], 1);
    RowPlanOffsets o;
    o.valid = ip.add(valid);
    o.start = ip.add(rs.off);
    o.len = ip.add(rs.len);
    return o;
}
static void bind_rows(const IntPlan& ip, const RowPlanOffsets& o, RowSet& rs) {
    rs.d_valid = ip.dev(o.valid);
    rs.d_start = ip.dev(o.start);
    rs.d_len = ip.dev(o.len);
}
// rowmap[r] = b * stride + t for real rows, -1 for gap rows (pack / unpack of padded tensors)
static int plan_rowmap(IntPlan& ip, const RowSet& rs, int stride) {
    std::vector<int> map(rs.R, -1);
    for (int b = 0; b < rs.B; ++b)
        for (int t = 0; t < rs.len[b]; ++t) map[rs.off[b] + t] = b * stride + t;
    return ip.add(map);
}

// ---------------------------------------------------------------------------------------------------
// kernel-call helpers

struct Ctx {
    mt2_model& m;
    hipStream_t s;
    Arena& ws;
};

// the bf16 planes of a weight pointer, if its buffer has them (mt2_model::planes, sorted by base)
static void attach_planes(const mt2_model& m, GemmP& p) {
    if (p.W3 || m.planes.empty()) return;
    auto it = std::upper_bound(m.planes.begin(), m.planes.end(), p.W,
                               [](const float* w, const PlaneRange& r) { return w < r.base; });
    if (it == m.planes.begin()) return;
    --it;
    if (p.W < it->base + it->n) {
        p.W3 = it->p3 + (p.W - it->base);
        p.w3_plane = (long long)it->n;
        // the fp16 planes carry one scale per weight row: usable when this launch walks the buffer with the rows it was split by
        // (K slices of a row - split-K - share the row's scale)
        const int ldw = p.ldw ? p.ldw : (p.taps > 0 ? p.taps : 1) * p.Cin;
        const long long rl = (long long)it->row_len, groups = p.groups > 0 ? p.groups : 1;
        const long long row0 = (long long)((size_t)(p.W - it->base) / it->row_len), col0 = (long long)((size_t)(p.W - it->base) % it->row_len);
        const long long ldb = 4ll * (long long)x3h_padded_k(it->row_len);      // bytes per chunk-interleaved row (x3h_planes.h)
        // groups either step through whole matrices (the parallel branches of a conv stack: strideW a multiple of the row length)
        // or through K slices of the SAME rows (split-K: every slice inside one row, shared scales); a launch starts on a chunk
        const bool whole = p.strideW % rl == 0, slices = !whole && col0 + p.strideW * groups <= rl && p.strideW % 32 == 0;
        if (it->ph && ldw == (int)it->row_len && col0 % 32 == 0 && (whole || slices || groups == 1)) {
            p.Wh = reinterpret_cast<const char*>(it->ph) + row0 * ldb + (col0 / 32) * 128;
            p.wh_ldb = ldb;
            p.wh_gstride = whole ? (p.strideW / rl) * ldb : (p.strideW / 32) * 128;
            p.wh_inv = it->inv + row0;
            p.wh_inv_stride = whole ? p.strideW / rl : 0;
        }
    }
}
// the tile-major copy of the matrix a weight pointer lies in, if it has one (mt2_model::tm, sorted by base) and the
// pointer starts on a block boundary (row multiple of 16, column multiple of 64)
static void attach_tm(const mt2_model& m, GemmP& p) {
    if (p.Wtm || m.tm.empty() || p.M > 64) return;
    auto it = std::upper_bound(m.tm.begin(), m.tm.end(), p.W, [](const float* w, const TmRange& r) { return w < r.base; });
    if (it == m.tm.begin()) return;
    --it;
    if (p.W >= it->base + it->n) return;
    const size_t off = (size_t)(p.W - it->base);
    const int n0 = (int)(off / it->K), k0 = (int)(off % it->K);
    if ((n0 & 15) || (k0 & 63) || p.ldw != it->K) return;
    p.Wtm = it->tm; p.tm_n0 = n0 / 16; p.tm_k0 = k0 / 64; p.tm_kb = it->K / 64;
}
static void gemm(const Ctx& c, GemmP p) {
    attach_planes(c.m, p);
    if (p.ldw == 0) p.ldw = p.taps > 0 ? p.taps * p.Cin : p.Cin;
    attach_tm(c.m, p);
    if (p.taps <= 0) p.taps = 1;
    if (p.dil <= 0) p.dil = 1;
    if (p.a_mul == 0) p.a_mul = 1;
    if (p.groups <= 0) p.groups = 1;
    if (p.out_scale == 0.0f) p.out_scale = 1.0f;
    p.K = p.taps * p.Cin;
    if (p.ldw == 0) p.ldw = p.K;
    MT2_HIP(launch_gemm(p, c.s, &c.m.opts));
}

// y[M, N] = x[M, K] @ W^T + b  (F.linear)
static void linear(const Ctx& c, const float* x, int ldx, int M, const float* W, const float* b, int N, int K,
                   float* y, int ldy, const float* R = nullptr, int ldr = 0, const int* valid = nullptr,
                   int epi_act = ACT_NONE, bool a_planes = false, bool c_planes = false) {
    GemmP p{};
    p.X = x; p.ldx = ldx; p.Rx = M; p.Cin = K; p.W = W; p.bias = b; p.R = R; p.ldr = ldr; p.valid = valid;
    p.C = y; p.ldc = ldy; p.M = M; p.N = N; p.epi_act = epi_act; p.a_planes = a_planes ? 1 : 0; p.c_planes = c_planes ? 1 : 0;
    gemm(c, p);
}
// the fields gemm() fills in before launch_gemm, for the queries gemm_takes_planes / gemm_writes_planes on a launch not yet made
static GemmP planned(const Ctx& c, GemmP p) {
    if (p.taps <= 0) p.taps = 1;
    if (p.dil <= 0) p.dil = 1;
    if (p.a_mul == 0) p.a_mul = 1;
    if (p.groups <= 0) p.groups = 1;
    if (p.out_scale == 0.0f) p.out_scale = 1.0f;
    p.K = p.taps * p.Cin;
    if (p.ldw == 0) p.ldw = p.K;
    attach_planes(c.m, p);
    return p;
}
// Will linear(x -> y) run on an x3h tile that takes x as fp16 planes (GemmP::a_planes)?  Then the LayerNorm that produces x writes
// planes (LnP::out_planes / LnReduceP::h_planes: same bytes, same row stride) and the GEMM's K loop holds no split arithmetic - in
// the loader tile that arithmetic is 8..18 % of a launch (profiles/r06_x3h_ablate_v3_split.txt).
static bool linear_takes_planes(const Ctx& c, const float* x, int ldx, int M, const float* W, const float* b, int N, int K,
                                float* y, int ldy, int epi_act) {
    if (!(c.m.opts.a_planes & 1)) return false;
    GemmP p{};
    p.X = x; p.ldx = ldx; p.Rx = M; p.Cin = K; p.W = W; p.bias = b; p.C = y; p.ldc = ldy; p.M = M; p.N = N; p.epi_act = epi_act;
    p.taps = 1; p.dil = 1; p.a_mul = 1; p.groups = 1; p.out_scale = 1.0f; p.K = K; p.ldw = K;
    attach_planes(c.m, p);
    return gemm_takes_planes(p, c.m.opts);
}

// y[M, N] = act(LN(x_rows)[M, K] @ W^T + b), x_rows[m] = x + (m * a_mul + shift0) * ldx.  Three forms, same result
// up to fp32 round-off:
//   at most 64 rows: ONE launch of the weight-streaming kernel, which normalises its A rows itself (from the (mean, M2) pairs
//     of the producer's epilogue where they exist);
//   pair-fed (ln_pairs; needs the folded operands Wl / s / c of EncLayerW and the producer's pairs): ONE launch,
//     rstd * (acc - mean * s) + c in the epilogue of an x6 / x3h tile, no pass over K;
//   plain: launch_layernorm into `h_scratch` + a GEMM.
// (LayerNorm as a prologue of the f32 tiles - options lnfuse / lnalg of rounds 1-2 - measured slower at every size and was
// retired in round 6: profiles/r06_retired_kernel_forms_and_options.patch.)
// What a layer hands to the next LayerNorm about the residual stream x (AR step forms below):
//   S > 0    - a residual update not yet applied: x += bias + sum_g parts[g] (split-K through the LayerNorm);
//   stat     - x is final and the GEMM that wrote it left its row statistics as (mean, M2) pairs per wave tile
//              (GemmP::stat_out): the consuming LN -> Linear pair runs as ONE pair-fed algebraic-LayerNorm GEMM.
struct Pending {
    const float* parts = nullptr; long long pstride = 0; int S = 0; const float* bias = nullptr;
    const float* stat = nullptr; int stat_nt = 0, stat_w = 0;
};
struct LnOps {                 // operands of one LN -> Linear pair, rows [n0, n0 + N) of the weight matrix
    const float *g, *b, *W, *bias;        // LayerNorm affine, Linear weight / bias
    const float *Wl, *s, *c;              // folded operands (nullptr: not available)
};
static LnOps ln_ops(const float* g, const float* b, const float* W, const float* bias, const float* Wl, const float* s,
                    const float* c, int n0, int K) {
    return LnOps{g, b, W + (size_t)n0 * K, bias + n0, Wl ? Wl + (size_t)n0 * K : nullptr, s ? s + n0 : nullptr,
                 c ? c + n0 : nullptr};
}
static void ln_linear(const Ctx& c, const float* x, int ldx, int Rx, int a_mul, int shift0, int M, const LnOps& w, int N,
                      int K, float* y, int ldy, float* h_scratch, int epi_act = ACT_NONE, const Pending* st = nullptr,
                      bool c_planes = false) {
    GemmP p{};
    p.c_planes = c_planes ? 1 : 0;      // y as fp16 planes (the caller asked gemm_writes_planes; more than 64 rows)
    p.X = x; p.ldx = ldx; p.Rx = Rx; p.a_mul = a_mul ? a_mul : 1; p.shift0 = shift0; p.taps = 1; p.dil = 1; p.Cin = K;
    p.K = K; p.W = w.W; p.ldw = K; p.bias = w.bias; p.C = y; p.ldc = ldy; p.M = M; p.N = N; p.groups = 1; p.out_scale = 1.0f;
    p.epi_act = epi_act; p.ln_eps = 1e-5f;
    if (c.m.opts.skinny_tm && M <= c.m.opts.skinny_rows && c.m.opts.force_cfg < 0) {
        // a handful of rows: the weight-streaming kernel normalises its A rows itself (gemm_skinny_tm_kernel, PRO_LN) -
        // one launch, statistics computed while the first weight pieces are in flight
        GemmP q = p;
        q.pro_act = 3; q.ln_g = w.g; q.ln_b = w.b;
        if (st && st->stat && st->stat_w == 16 && c.m.opts.skinny_pairs) {      // the rows' statistics came with them (16-column pairs)
            q.ln_stat = st->stat; q.ln_nt = st->stat_nt; q.ln_w = 16;
        }
        attach_tm(c.m, q);
        if (q.Wtm && gemm_skinny_tm_eligible(q, c.m.opts.skinny_rows)) {
            // launch_gemm routes to the tile-major kernel only when its own conditions hold too (skinny_groups); with another
            // routing the LayerNorm-prologue form may not exist for the tile it picks: fall through like the branches below
            const hipError_t e = launch_gemm(q, c.s, &c.m.opts);
            if (e == hipSuccess) return;
            if (e != hipErrorNotSupported) MT2_HIP(e);
        }
    }
    // (pairs of the tile-major kernel - 16 columns each, up to 64 per row - are not what the x6 / x3h tiles merge: launch_gemm would
    // answer InvalidValue, not NotSupported)
    if (st && st->stat && st->stat_nt > 0 && st->stat_w != 16 && st->stat_nt <= 32 && (st->stat_nt & 1) == 0 && c.m.opts.ln_pairs && w.Wl &&
        c.m.opts.force_cfg < 0 && M <= c.m.opts.ln_pairs_maxm) {
        // the rows' statistics came with them (pairs from the producer GEMM's epilogue): algebraic LayerNorm with NO pass
        // over K - rstd * (x W'^T - mean * s) + c on the x6 tiles; a tile without that form answers NotSupported
        GemmP q = p;
        q.pro_act = 5; q.W = w.Wl; q.bias = w.c; q.ln_g = w.s; q.ln_stat = st->stat; q.ln_nt = st->stat_nt; q.ln_w = st->stat_w;
        attach_planes(c.m, q);
        const hipError_t e = launch_gemm(q, c.s, &c.m.opts);
        if (e == hipSuccess) return;
        if (e != hipErrorNotSupported) MT2_HIP(e);
    }
    const bool planes = linear_takes_planes(c, h_scratch, K, M, w.W, w.bias, N, K, y, ldy, epi_act);
    LnP q{};
    q.x = x + (long long)shift0 * ldx; q.ldx = ldx * p.a_mul; q.gamma = w.g; q.beta = w.b; q.out = h_scratch; q.ldo = K;
    q.M = M; q.C = K; q.eps = 1e-5f; q.act = ACT_NONE; q.out_planes = planes ? 1 : 0; q.x3h_flag = c.m.opts.x3h_flag;
    MT2_HIP(launch_layernorm(q, c.s));
    linear(c, h_scratch, K, M, w.W, w.bias, N, K, y, ldy, nullptr, 0, nullptr, epi_act, planes, c_planes);
}
static LnOps ln1_qkv(const EncLayerW& w, int d, int n0 = 0) {
    return ln_ops(w.ln1g, w.ln1b, w.wqkv, w.bqkv, w.wqkv_l, w.sqkv, w.cqkv, n0, d);
}
static LnOps ln2_ff0(const EncLayerW& w, int d) {
    return ln_ops(w.ln2g, w.ln2b, w.ff0w, w.ff0b, w.ff0_l, w.sff0, w.cff0, 0, d);
}

// nn.Conv1d(k, stride 1, padding (k-1)/2 * dil, dilation dil) over gap-padded rows
static void conv_same(const Ctx& c, const float* x, int ldx, int R, const ConvW& w, float* y, int ldy,
                      const int* valid, int pro_act = ACT_NONE, float slope = 0.f, int epi_act = ACT_NONE,
                      const float* Rsd = nullptr, int ldr = 0, int dil = 1) {
    GemmP p{};
    p.X = x; p.ldx = ldx; p.Rx = R; p.taps = w.k; p.dil = dil; p.shift0 = -((w.k - 1) / 2) * dil; p.Cin = w.cin;
    p.W = w.w; p.bias = w.b; p.R = Rsd; p.ldr = ldr; p.valid = valid; p.C = y; p.ldc = ldy; p.M = R; p.N = w.cout;
    p.pro_act = pro_act; p.pro_slope = slope; p.epi_act = epi_act;
    gemm(c, p);
}

static void layernorm(const Ctx& c, const float* x, int ldx, const float* g, const float* b, int M, int C,
                      float* out, int ldo, const int* valid = nullptr, int valid_rows = 0,
                      const float* R1 = nullptr, int ldr1 = 0, int r1_rows = 0, int rows_per_group = 0,
                      int act = ACT_NONE, bool out_planes = false) {
    LnP p{};
    p.x = x; p.ldx = ldx; p.gamma = g; p.beta = b; p.rows_per_group = rows_per_group;
    p.R1 = R1; p.ldr1 = ldr1; p.r1_rows = r1_rows; p.valid = valid; p.valid_rows = valid_rows;
    p.out = out; p.ldo = ldo; p.M = M; p.C = C; p.eps = 1e-5f; p.act = act;
    p.out_planes = out_planes ? 1 : 0; p.x3h_flag = c.m.opts.x3h_flag;
    MT2_HIP(launch_layernorm(p, c.s));
}

// ResidualBlockStack.forward (modules/convnet.py:69-72) for `groups` parallel branches at once:
// x = x + ConvStack(x), ConvBlock = ReLU -> Conv1d -> LayerNorm(C) (convnet.py:23-31).
// x_in: [R, C] shared by all groups (shared_in) or [groups][R, C].  Returns [groups][R, C].
static float* run_stack(const Ctx& c, const StackW& w, const float* x_in, bool shared_in, int R,
                        const int* valid) {
    const int C = w.C, G = w.groups;
    const size_t per = (size_t)R * C;
    float* T = c.ws.get<float>(per * G);
    float* Y = c.ws.get<float>(per * G);
    float* XA = c.ws.get<float>(per * G);
    float* XB = c.ws.get<float>(per * G);
    const float* cur = x_in;
    bool cur_shared = shared_in;
    float* nxt = XA;
    const size_t wsz = (size_t)C * w.k * C;
    for (int st = 0; st < w.nstack; ++st) {
        const float* bin = cur;
        bool bin_shared = cur_shared;
        bool bin_relued = false;    // a block output with ONE consumer (the next block's conv) is stored ReLU'd
        bool bin_planes = false;    // ... and as fp16 planes when that conv runs on an x3h tile that takes them (GemmP::a_planes)
        auto block_conv = [&](int blk, const float* in, bool in_shared, bool relued, bool planes) {
            const size_t e = w.idx(st, blk);
            GemmP p{};
            p.X = in; p.strideX = in_shared ? 0 : (long long)per; p.ldx = C; p.Rx = R;
            p.taps = w.k; p.shift0 = -((w.k - 1) / 2); p.Cin = C;
            p.W = w.w + e * wsz; p.strideW = (long long)wsz;
            p.bias = w.b + e * C; p.strideB = C;
            p.valid = valid; p.C = T; p.strideC = (long long)per; p.ldc = C; p.M = R; p.N = C; p.groups = G;
            p.pro_act = relued ? ACT_NONE : ACT_RELU;
            p.a_planes = planes ? 1 : 0;
            return p;
        };
        for (int blk = 0; blk < w.nblock; ++blk) {
            const size_t e = w.idx(st, blk);
            gemm(c, block_conv(blk, bin, bin_shared, bin_relued, bin_planes));
            const bool last = blk == w.nblock - 1;
            bool planes = false;
            if (!last && (c.m.opts.a_planes & 1)) {       // will the next block's conv take Y as planes?
                GemmP q = block_conv(blk + 1, Y, false, true, false);
                q.dil = 1; q.a_mul = 1; q.out_scale = 1.0f; q.K = q.taps * q.Cin; q.ldw = q.K;
                attach_planes(c.m, q);
                planes = gemm_takes_planes(q, c.m.opts);
            }
            if (last)
                layernorm(c, T, C, w.g + e * C, w.be + e * C, R * G, C, nxt, C, valid, R, cur, C,
                          cur_shared ? R : 0, R);
            else   // next ConvBlock starts with ReLU (convnet.py:24): fold it into this LayerNorm's store
                layernorm(c, T, C, w.g + e * C, w.be + e * C, R * G, C, Y, C, valid, R, nullptr, 0, 0, R, ACT_RELU, planes);
            bin = Y;
            bin_shared = false;
            bin_relued = !last;
            bin_planes = planes;
        }
        cur = nxt;
        cur_shared = false;
        nxt = (nxt == XA) ? XB : XA;
    }
    return const_cast<float*>(cur);
}

// TransformerEncoderLayer.forward (modules/transformer.py:88-102) over packed rows, in place on x.
//   attention geometry: per-utterance (starts/lens arrays) or uniform (AR steps).
struct AttnGeom {
    const int* start = nullptr; const int* len = nullptr;
    int u_stride = 0, u_len = 0, B = 0, max_len = 0;
};
struct EncScratch { float *h, *qkv, *att, *f, *parts, *stat; };
static EncScratch enc_scratch(const Ctx& c, const EncW& e, int M) {
    EncScratch s;
    s.h = c.ws.get<float>((size_t)M * e.d);
    s.qkv = c.ws.get<float>((size_t)M * 3 * e.d);
    s.att = c.ws.get<float>((size_t)M * e.d);
    s.f = c.ws.get<float>((size_t)M * e.ff);
    // split-K slabs [S][M, d]: the f32 rule of choose_split keeps tiles(32x64) * S <= 256 (S * M * d <= 256 * 32 * 64 *
    // d / 64); the x6 rule splits at most 8 ways at any M
    s.parts = c.ws.get<float>(std::max((size_t)256 * 32 * 64 + (size_t)16 * 32 * e.d, (size_t)8 * M * e.d));
    s.stat = c.ws.get<float>((size_t)M * 128);     // row statistics handed from GEMM to GEMM: <= 64 (mean, M2) pairs per row
    return s;
}
static void attention_self(const Ctx& c, const EncW& e, const AttnGeom& g, const float* qkv, float* att, bool o_planes = false) {
    AttnP a{};
    a.o_planes = o_planes ? 1 : 0; a.x3h_flag = c.m.opts.x3h_flag;
    const int d = e.d, D = d / e.heads;
    a.Q = qkv; a.ldq = 3 * d; a.K = qkv + d; a.ldk = 3 * d; a.V = qkv + 2 * d; a.ldv = 3 * d;
    a.O = att; a.ldo = d;
    a.q_start = g.start; a.q_len = g.len; a.kv_start = g.start; a.kv_len = g.len;
    a.u_qstride = g.u_stride; a.u_qlen = g.u_len; a.u_kvstride = g.u_stride; a.u_kvlen = g.u_len;
    a.B = g.B; a.H = e.heads; a.D = D; a.max_qlen = g.max_len; a.max_kvlen = g.max_len;
    a.scale = 1.0f / std::sqrt((float)D);
    a.lds_min_qlen = c.m.opts.attn_lds_min; a.lds_waves = c.m.opts.attn_lds_waves; a.x6_min_qlen = c.m.opts.attn_x6_min;
    a.x3h = (c.m.opts.x3h & 8) ? 1 : 0; a.x3h_flag = c.m.opts.x3h_flag;
    a.ds_short = c.m.opts.attn_ds;
    MT2_HIP(launch_attention(a, c.s));
}
static void encoder_layer(const Ctx& c, const EncW& e, const EncLayerW& w, float* x, int M, const AttnGeom& g,
                          const int* valid, const EncScratch& s) {
    const int d = e.d;
    layernorm(c, x, d, w.ln1g, w.ln1b, M, d, s.h, d, valid);
    linear(c, s.h, d, M, w.wqkv, w.bqkv, 3 * d, d, s.qkv, 3 * d);
    attention_self(c, e, g, s.qkv, s.att);
    linear(c, s.att, d, M, w.wo, w.bo, d, d, x, d, x, d, valid);            // x = x + out_proj(att)
    if (e.conv_ff) {
        // x = norm2(x); x = x + conv2(relu(conv1(x)))   (transformer.py:95-99; residual from the NORMED x)
        layernorm(c, x, d, w.ln2g, w.ln2b, M, d, s.h, d, valid);
        ConvW c1{w.ff0w, w.ff0b, e.ff, d, 5}, c2{w.ff1w, w.ff1b, d, e.ff, 5};
        conv_same(c, s.h, d, M, c1, s.f, e.ff, valid, ACT_NONE, 0.f, ACT_RELU);
        conv_same(c, s.f, e.ff, M, c2, x, d, valid, ACT_NONE, 0.f, ACT_NONE, s.h, d);
    } else {
        layernorm(c, x, d, w.ln2g, w.ln2b, M, d, s.h, d, valid);
        linear(c, s.h, d, M, w.ff0w, w.ff0b, e.ff, d, s.f, e.ff, nullptr, 0, nullptr, ACT_RELU);
        linear(c, s.f, e.ff, M, w.ff1w, w.ff1b, d, e.ff, x, d, x, d, valid);
    }
}

// ---- autoregressive-step forms of the encoder layer (Linear-FF encoders only).  All of them are exact
// identities of TransformerEncoderLayer.forward (modules/transformer.py:88-102), not approximations.
//
// Split-K through the LayerNorm.  A GEMM of the early / middle steps has too few tiles for the chip, and a
// tile's serial K chain is its latency floor (K = 4096 in the PLM's second feed-forward: 45 us at any M).
// Such a GEMM runs as S independent K slices (GemmP groups: slice g reads columns [g*K/S, (g+1)*K/S) of both
// operands and writes a raw partial slab) and the NEXT LayerNorm - a launch the layer has anyway - sums the
// slabs in fixed order, adds bias and residual, writes the residual stream and its normalisation
// (launch_ln_reduce).  Deterministic, no extra launch, no inter-workgroup hand-off.
static int choose_split(const Ctx& c, int M, int N, int K) {
    if (!c.m.opts.splitk) return 1;
    // a handful of rows on the tile-major weight-streaming kernel (round 4): that kernel adds bias + residual itself and the
    // NEXT GEMM normalises its own input rows (LayerNorm prologue), so a split - whose reduction needs a LayerNorm launch of
    // its own - only pays where one column block's K range is too long for one workgroup: the PLM's ff.3 (K = 4096: 64
    // blocks x 256 KiB); there the K slices are 1024 wide (256 workgroups x 64 KiB) and the reduction rides on LN1 as before
    if (c.m.opts.skinny_tm && c.m.opts.skinny_rows > 0 && M <= c.m.opts.skinny_rows && M <= 64 && c.m.opts.force_cfg < 0 &&
        (N & 15) == 0 && (K & 63) == 0) {
        if (K <= 1536) return 1;
        int S = 1;
        const int kmin = 1024;      // narrowest K slice (two slices of 2048: C1 +2.6 %, eight of 512: +0.4 %; un-split: +7.3 %)
        while (S < 16 && K % (S * 2) == 0 && K / (S * 2) >= kmin && (K / (S * 2)) % 64 == 0) S *= 2;
        return S;
    }
    // x6 form (large M): the N = d GEMMs of a layer (out-projection, ff.3) have too few 128x128 tiles for the bf16 pipe
    // (N = 768 / 1024: 6 / 8 column tiles); K slices as GEMM groups bring them to >= t_x6_128 tiles, and the reduction
    // rides on the next LayerNorm as before (profiles/r02_gemm_sweep_x6.txt: ff.3 at M = 864 does 85 TF/s on the f32
    // K-split tile; four x6 slices of K = 1024 have the shape of ff.0, 125 TF/s)
    if (c.m.opts.x6_gemm && c.m.opts.x6_splitk && (N & 127) == 0 && (K & 31) == 0) {
        const long long t128 = (long long)((M + 127) / 128) * (N / 128);
        if (t128 < c.m.opts.t_x6_128) {
            int S = 1;
            while (S < 8 && t128 * S < 2 * c.m.opts.t_x6_128 && K % (S * 2) == 0 && K / (S * 2) >= 512 && (K / (S * 2)) % 32 == 0)
                S *= 2;
            if (S > 1 && t128 * S >= c.m.opts.t_x6_128) return S;
        }
    }
    const long long tiles = (long long)((M + 31) / 32) * ((N + 63) / 64);
    int S = 1;
    while (S < 16 && tiles * (S * 2) <= 256 && K % (S * 2) == 0 && K / (S * 2) >= 256 && (K / (S * 2)) % 32 == 0) S *= 2;
    return S;
}
// y_parts[g] = x[:, gK/S:(g+1)K/S] @ W[:, gK/S:(g+1)K/S]^T, g < S  (raw partial slabs [S][M, N])
static GemmP splitk_params(const float* x, int ldx, int M, const float* W, int N, int K, int S, float* parts) {
    GemmP p{};
    p.X = x; p.strideX = K / S; p.ldx = ldx; p.Rx = M; p.Cin = K / S; p.W = W; p.strideW = K / S; p.ldw = K;
    p.C = parts; p.strideC = (long long)M * N; p.ldc = N; p.M = M; p.N = N; p.groups = S;
    return p;
}
static void linear_splitk(const Ctx& c, const float* x, int ldx, int M, const float* W, int N, int K, int S,
                          float* parts, bool a_planes = false) {
    GemmP p = splitk_params(x, ldx, M, W, N, K, S, parts);
    p.a_planes = a_planes ? 1 : 0;
    gemm(c, p);
}
// h = LayerNorm(x) after applying a pending update to x (in place)
static void ln_pending(const Ctx& c, float* x, int d, int M, const Pending& in, const float* g, const float* b,
                       float* h, bool h_planes = false) {
    if (in.S == 0) {
        LnP q{};
        q.x = x; q.ldx = d; q.gamma = g; q.beta = b; q.out = h; q.ldo = d; q.M = M; q.C = d; q.eps = 1e-5f; q.act = ACT_NONE;
        q.out_planes = h_planes ? 1 : 0; q.x3h_flag = c.m.opts.x3h_flag;
        MT2_HIP(launch_layernorm(q, c.s));
        return;
    }
    LnReduceP p{};
    p.parts = in.parts; p.pstride = in.pstride; p.S = in.S; p.bias = in.bias; p.R = x; p.ldr = d;
    p.gamma = g; p.beta = b; p.xout = x; p.ldx = d; p.hout = h; p.ldh = d; p.M = M; p.C = d; p.eps = 1e-5f;
    p.h_planes = h_planes ? 1 : 0; p.x3h_flag = c.m.opts.x3h_flag;
    MT2_HIP(launch_ln_reduce(p, c.s));
}
// h = LN(x (+ the pending split-K update)), y = act(h W^T + b): h travels as fp16 planes when the GEMM takes them
static void ln_pending_linear(const Ctx& c, float* x, int d, int M, const Pending& in, const float* g, const float* b, float* h,
                              const float* W, const float* bias, int N, float* y, int ldy, int epi_act = ACT_NONE,
                              bool c_planes = false) {
    const bool planes = M <= 4096 && linear_takes_planes(c, h, d, M, W, bias, N, d, y, ldy, epi_act);
    ln_pending(c, x, d, M, in, g, b, h, planes);
    linear(c, h, d, M, W, bias, N, d, y, ldy, nullptr, 0, nullptr, epi_act, planes, c_planes);
}
// everything after attention for M full rows: x += out_proj(att); h = LN2(x); f = relu(ff0(h));
// x += ff1(f) - the last update is returned as pending when it was split
// x += a @ W^T + b (residual update in the GEMM's epilogue); with ln_pairs the epilogue also leaves the row statistics of the
// new x as pairs in `stat` where the chosen tile can - the returned Pending says whether it did
static GemmP residual_params(const Ctx& c, const float* a, int lda, int M, const float* W, const float* b, int N, int K,
                             float* x, float* stat, const float* res, int ldr) {
    GemmP p{};
    p.X = a; p.ldx = lda; p.Rx = M; p.Cin = K; p.W = W; p.bias = b; p.R = res ? res : x; p.ldr = res ? ldr : N; p.C = x; p.ldc = N;
    p.M = M; p.N = N;
    const bool small = M <= 64 && M <= c.m.opts.skinny_rows && c.m.opts.skinny_tm && c.m.opts.skinny_pairs;     // tile-major kernel: pairs per 16 columns
    const bool want = stat != nullptr && c.m.opts.force_cfg < 0 &&
                      (small || (c.m.opts.ln_pairs > 0 && M > 64 && M <= c.m.opts.ln_pairs_maxm));
    if (want) p.stat_out = stat;
    return p;
}
static Pending linear_residual(const Ctx& c, const float* a, int lda, int M, const float* W, const float* b, int N, int K,
                               float* x, float* stat, const float* res = nullptr, int ldr = 0, bool a_planes = false) {
    GemmP p = residual_params(c, a, lda, M, W, b, N, K, x, stat, res, ldr);
    const bool want = p.stat_out != nullptr;
    p.a_planes = a_planes ? 1 : 0;
    gemm(c, p);
    Pending r{};
    if (want && c.m.opts.last_stat_nt > 0) { r.stat = stat; r.stat_nt = c.m.opts.last_stat_nt; r.stat_w = c.m.opts.last_stat_w; }
    return r;
}
// K slices of the two residual GEMMs of an AR layer (out-projection K = d, ff.3 K = ff)
static void ar_tail_splits(const Ctx& c, const EncW& e, int M, int& S1, int& S2) {
    const int d = e.d;
    // ln_pairs = 2: the residual GEMMs with a short K chain (<= 1024) are not K-split any more - a split hands its reduction
    // to a LayerNorm launch, the un-split GEMM hands the statistics to the next GEMM instead
    const bool unsplit = c.m.opts.ln_pairs >= 2 && M > 64 && M <= c.m.opts.ln_pairs_maxm && c.m.opts.force_cfg < 0;
    S1 = choose_split(c, M, d, d);
    if (unsplit && d <= c.m.opts.ln_pairs_maxk) S1 = 1;
    S2 = choose_split(c, M, d, e.ff);
    if (unsplit && e.ff <= c.m.opts.ln_pairs_maxk) S2 = 1;
}
// attention -> out-projection: will the out-projection of ar_layer_tail take `att` as fp16 planes (then the attention kernel stores
// them: AttnP::o_planes)?  The same hand-over as ff.0 -> ff.3 below, with the attention kernels as producers.
static bool ar_outproj_takes_planes(const Ctx& c, const EncW& e, const EncLayerW& w, float* x, int M, const float* att,
                                    const EncScratch& s) {
    const int d = e.d;
    if (!(c.m.opts.a_planes & 4) || M <= 64 || M > 4096 || (d & 31) || ((d / e.heads) & 31)) return false;
    int S1, S2;
    ar_tail_splits(c, e, M, S1, S2);
    const GemmP q = S1 > 1 ? splitk_params(att, d, M, w.wo, d, d, S1, s.parts)
                           : residual_params(c, att, d, M, w.wo, w.bo, d, d, x, s.stat, nullptr, 0);
    return gemm_takes_planes(planned(c, q), c.m.opts);
}
static Pending ar_layer_tail(const Ctx& c, const EncW& e, const EncLayerW& w, float* x, int M, const float* att,
                             const EncScratch& s, bool att_planes = false) {
    const int d = e.d;
    int S1, S2;
    ar_tail_splits(c, e, M, S1, S2);
    // ff.0 -> ff.3: f = relu(ff.0(h)) has ONE consumer.  Where ff.0 runs on the x3h loader tile and ff.3 on an x3h tile that takes its A
    // operand as fp16 planes, ff.0's epilogue stores f as planes (GemmP::c_planes: the split once per element, in the producer, instead of
    // once per element and column tile in the consumer's K loop) - same values, bit-identical results
    bool fpl = false;
    if ((c.m.opts.a_planes & 2) && M > 64 && M <= 4096) {
        GemmP q0{};
        q0.X = s.h; q0.ldx = d; q0.Rx = M; q0.Cin = d; q0.W = w.ff0w; q0.bias = w.ff0b; q0.C = s.f; q0.ldc = e.ff; q0.M = M; q0.N = e.ff;
        q0.epi_act = ACT_RELU;
        const GemmP q3 = S2 > 1 ? splitk_params(s.f, e.ff, M, w.ff1w, d, e.ff, S2, s.parts)
                                : residual_params(c, s.f, e.ff, M, w.ff1w, w.ff1b, d, e.ff, x, s.stat, nullptr, 0);
        fpl = gemm_writes_planes(planned(c, q0), c.m.opts) && gemm_takes_planes(planned(c, q3), c.m.opts);
    }
    if (S1 > 1) {
        linear_splitk(c, att, d, M, w.wo, d, d, S1, s.parts, att_planes);
        Pending p1{s.parts, (long long)M * d, S1, w.bo};
        ln_pending_linear(c, x, d, M, p1, w.ln2g, w.ln2b, s.h, w.ff0w, w.ff0b, e.ff, s.f, e.ff, ACT_RELU, fpl);
    } else {
        const Pending p1 = linear_residual(c, att, d, M, w.wo, w.bo, d, d, x, s.stat, nullptr, 0, att_planes);   // x += out_proj(att)
        ln_linear(c, x, d, M, 1, 0, M, ln2_ff0(w, d), e.ff, d, s.f, e.ff, s.h, ACT_RELU, &p1, fpl);    // LN2 -> ff.0
    }
    if (S2 > 1) {
        linear_splitk(c, s.f, e.ff, M, w.ff1w, d, e.ff, S2, s.parts, fpl);
        return Pending{s.parts, (long long)M * d, S2, w.ff1b};
    }
    return linear_residual(c, s.f, e.ff, M, w.ff1w, w.ff1b, d, e.ff, x, s.stat, nullptr, 0, fpl);     // x += ff.3(f)
}

// MIDDLE layer over M = A*n compact rows
static Pending encoder_layer_ar(const Ctx& c, const EncW& e, const EncLayerW& w, float* x, int M, const AttnGeom& g,
                                const EncScratch& s, const Pending& in) {
    MT2_REQUIRE(!e.conv_ff, "AR step layers use the Linear feed-forward");
    const int d = e.d;
    if (in.S) {
        ln_pending_linear(c, x, d, M, in, w.ln1g, w.ln1b, s.h, w.wqkv, w.bqkv, 3 * d, s.qkv, 3 * d);
    } else {
        ln_linear(c, x, d, M, 1, 0, M, ln1_qkv(w, d), 3 * d, d, s.qkv, 3 * d, s.h, ACT_NONE, &in);     // LN1 -> QKV
    }
    const bool opl = !g.start && ar_outproj_takes_planes(c, e, w, x, M, s.att, s);      // (uniform geometry: every row of att is written)
    attention_self(c, e, g, s.qkv, s.att, opl);
    return ar_layer_tail(c, e, w, x, M, s.att, s, opl);
}

// LAST layer: only row n-1 of each sequence is consumed downstream (models/megatts2.py:178,272).  LayerNorm
// and the K/V projections are row-wise and needed for all rows; attention row i depends on query row i only,
// and out-projection, norm2 and the feed-forward are row-wise - so Q, attention, out-proj, LN2 and FF are
// evaluated for the A last rows only.  x: [A*n, d] compact step rows; y: [A, d] result rows.
static void encoder_layer_last(const Ctx& c, const EncW& e, const EncLayerW& w, float* x, int n, int A,
                               const EncScratch& s, float* y, const Pending& in) {
    MT2_REQUIRE(!e.conv_ff, "AR step layers use the Linear feed-forward");
    const int d = e.d, M = A * n, D = d / e.heads;
    float* kv = s.qkv;                                   // [M, 2d]: K | V
    float* q = s.att;                                    // [A, d]
    float* att = s.att + (size_t)A * d;                  // [A, d]
    AttnP a{};
    if (c.m.opts.skinny_tm && M <= c.m.opts.skinny_rows && M <= 64 && c.m.opts.force_cfg < 0) {
        // a handful of rows (one utterance's steps): Q | K | V of ALL rows in ONE weight-streaming launch - the Q rows that
        // are not the last of their sequence cost nothing measurable at M <= 64, a second launch costs ~8 us
        float* qkv = s.qkv;                              // [M, 3d]
        if (in.S) {
            ln_pending(c, x, d, M, in, w.ln1g, w.ln1b, s.h);
            linear(c, s.h, d, M, w.wqkv, w.bqkv, 3 * d, d, qkv, 3 * d);
        } else {
            ln_linear(c, x, d, M, 1, 0, M, ln1_qkv(w, d), 3 * d, d, qkv, 3 * d, s.h, ACT_NONE, &in);
        }
        a.Q = qkv + (size_t)(n - 1) * 3 * d; a.ldq = 3 * d; a.u_qstride = n;      // query of sequence j = row j * n + n - 1
        a.K = qkv + d; a.ldk = 3 * d; a.V = qkv + 2 * d; a.ldv = 3 * d;
        a.u_ostride = 1;
    } else if (in.S) {
        ln_pending(c, x, d, M, in, w.ln1g, w.ln1b, s.h);
        linear(c, s.h, d, M, w.wqkv + (size_t)d * d, w.bqkv + d, 2 * d, d, kv, 2 * d);
        GemmP p{};
        p.X = s.h; p.ldx = d; p.Rx = M; p.a_mul = n; p.shift0 = n - 1; p.Cin = d; p.W = w.wqkv; p.bias = w.bqkv;
        p.C = q; p.ldc = d; p.M = A; p.N = d;
        gemm(c, p);
    } else {   // LN1 fused into both consumers: K|V of all rows, Q of the last row of each sequence
        ln_linear(c, x, d, M, 1, 0, M, ln1_qkv(w, d, d), 2 * d, d, kv, 2 * d, s.h, ACT_NONE, &in);
        ln_linear(c, x, d, M, n, n - 1, A, ln1_qkv(w, d), d, d, q, d, s.f, ACT_NONE, &in);
    }
    if (!a.Q) {
        a.Q = q; a.ldq = d; a.K = kv; a.ldk = 2 * d; a.V = kv + d; a.ldv = 2 * d;
        a.u_qstride = 1;
    }
    a.O = att; a.ldo = d;
    a.u_qlen = 1; a.u_kvstride = n; a.u_kvlen = n; a.B = A; a.H = e.heads; a.D = D; a.max_qlen = 1;
    a.scale = 1.0f / std::sqrt((float)D);
    a.lds_min_qlen = c.m.opts.attn_lds_min; a.lds_waves = c.m.opts.attn_lds_waves; a.x6_min_qlen = c.m.opts.attn_x6_min;
    a.x3h = (c.m.opts.x3h & 8) ? 1 : 0; a.x3h_flag = c.m.opts.x3h_flag;
    a.ds_short = c.m.opts.attn_ds;
    MT2_HIP(launch_attention(a, c.s));
    // y = x[last rows] + out_proj(att): the residual rows sit n*d floats apart starting at row n-1
    const Pending py = linear_residual(c, att, d, A, w.wo, w.bo, d, d, y, s.stat, x + (size_t)(n - 1) * d, n * d);
    ln_linear(c, y, d, A, 1, 0, A, ln2_ff0(w, d), e.ff, d, s.f, e.ff, s.h, ACT_RELU, &py);   // LN2 -> ff.0
    linear(c, s.f, e.ff, A, w.ff1w, w.ff1b, d, e.ff, y, d, y, d);
}

// FIRST layer: its input rows (embeddings + positional table of positions < t) never change between steps,
// so LN1 -> QKV of position i is computed once, at step i, into a per-sequence cache with a FIXED row
// stride (slot j owns rows [j*cs, j*cs + n)); every later step only adds row n-1.  Attention reads the
// cache and writes compact rows; the rest of the layer is the ordinary full-row form.
static Pending encoder_layer_first_cached(const Ctx& c, const EncW& e, const EncLayerW& w, float* x, int n, int A,
                                          float* qkv_cache, int cs, const EncScratch& s, bool fill_all) {
    MT2_REQUIRE(!e.conv_ff, "AR step layers use the Linear feed-forward");
    const int d = e.d, M = A * n, D = d / e.heads;
    if (fill_all && n > 1) {
        // first step of a run that starts from a forced history (prompt prefix, teacher-forced tests): the cache
        // has no rows yet - LN1 -> QKV of ALL n rows, compact, then one strided copy into the cache layout
        ln_linear(c, x, d, M, 1, 0, M, ln1_qkv(w, d), 3 * d, d, s.qkv, 3 * d, s.h);
        MT2_HIP(launch_copy_2d(s.qkv, (long long)n * 3 * d, qkv_cache, (long long)cs * 3 * d, (long long)n * 3 * d, A,
                               c.s));
    } else {
        // LN1 -> QKV of the newest row of every active sequence, written into the cache at row stride cs
        ln_linear(c, x, d, M, n, n - 1, A, ln1_qkv(w, d), 3 * d, d, qkv_cache + (size_t)(n - 1) * 3 * d, cs * 3 * d, s.h);
    }
    AttnP a{};
    a.Q = qkv_cache; a.ldq = 3 * d; a.K = qkv_cache + d; a.ldk = 3 * d; a.V = qkv_cache + 2 * d; a.ldv = 3 * d;
    a.O = s.att; a.ldo = d;
    a.u_qstride = cs; a.u_qlen = n; a.u_kvstride = cs; a.u_kvlen = n; a.u_ostride = n;
    a.B = A; a.H = e.heads; a.D = D; a.max_qlen = n; a.scale = 1.0f / std::sqrt((float)D);
    a.lds_min_qlen = c.m.opts.attn_lds_min; a.lds_waves = c.m.opts.attn_lds_waves; a.x6_min_qlen = c.m.opts.attn_x6_min;
    a.x3h = (c.m.opts.x3h & 8) ? 1 : 0; a.x3h_flag = c.m.opts.x3h_flag;
    a.ds_short = c.m.opts.attn_ds;
    const bool opl = ar_outproj_takes_planes(c, e, w, x, M, s.att, s);
    a.o_planes = opl ? 1 : 0; a.x3h_flag = c.m.opts.x3h_flag;
    MT2_HIP(launch_attention(a, c.s));
    return ar_layer_tail(c, e, w, x, M, s.att, s, opl);
}

// One AR step of an encoder over A sequences of n positions (x: [A*n, d], overwritten); the rows the head
// needs (position n-1 of each sequence) are returned as a [A, d] matrix.
static const float* ar_step_layers(const Ctx& c, const EncW& e, float* x, int n, int A, float* qkv_cache, int cs,
                                   const EncScratch& sc, float* ylast, bool fill_cache = false) {
    const int L = (int)e.layers.size();
    AttnGeom g;
    g.u_stride = n; g.u_len = n; g.B = A; g.max_len = n;
    Pending pend;
    for (int l = 0; l < L; ++l) {
        if (l == L - 1) encoder_layer_last(c, e, e.layers[l], x, n, A, sc, ylast, pend);
        else if (l == 0 && qkv_cache) pend = encoder_layer_first_cached(c, e, e.layers[l], x, n, A, qkv_cache, cs, sc, fill_cache);
        else pend = encoder_layer_ar(c, e, e.layers[l], x, A * n, g, sc, pend);
    }
    return ylast;
}

// ---------------------------------------------------------------------------------------------------
// stage timers (HIP events on the caller's stream)

struct Stages {
    mt2_model& m;
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    std::vector<std::string> names;
    explicit Stages(mt2_model& mm, hipStream_t ss) : m(mm), s(ss) {}
    // marker ids (tools/pmc_stage_summary.py): 0 start, 1 mrte, 2 adm, 3 regulate, 4 plm, 5 decoder, 6 vocoder,
    // 8 / 9 around mt2_vqpe_forward
    int nmark = 0;
    void mark(const char* name) {
        if (m.opts.markers) MT2_HIP(launch_stage_marker(nmark, s));
        ++nmark;
        if (!m.profiling) return;
        hipEvent_t e;
        MT2_HIP(hipEventCreate(&e));
        MT2_HIP(hipEventRecord(e, s));
        ev.push_back(e);
        names.push_back(name);
    }
    void finish() {
        if (!m.profiling || ev.empty()) return;
        MT2_HIP(hipEventSynchronize(ev.back()));
        m.stage_names.clear();
        m.stage_ms.clear();
        for (size_t i = 1; i < ev.size(); ++i) {
            float ms = 0.f;
            MT2_HIP(hipEventElapsedTime(&ms, ev[i - 1], ev[i]));
            m.stage_names.push_back(names[i]);
            m.stage_ms.push_back(ms);
        }
        for (auto e : ev) (void)hipEventDestroy(e);
        ev.clear();
    }
};

// ---------------------------------------------------------------------------------------------------
// ConvNetDouble (modules/convnet.py:156-210): first conv, N parallel branches (stack1 -> middle ->
// stack2) summed, last conv.  `middle` turns [G][Rin, C] into [G][Rout, C].

// MRTE mel encoder: middle = ONE shared Conv1d(C, C, stride+1, stride, pad stride/2) (mrte.py:101-107)
static float* mel_context_rows(const Ctx& c, const float* xmel, int ld_mel, RowSet& F, RowSet& X,
                               const int* d_rowbase) {
    mt2_model& m = c.m;
    const int C = m.cfg.mrte_hidden, G = m.mel_s1.groups;
    float* h0 = c.ws.get<float>((size_t)F.R * C);
    {
        GemmP p{};
        p.X = xmel; p.ldx = ld_mel; p.Rx = F.R; p.taps = m.mel_first.k; p.shift0 = -((m.mel_first.k - 1) / 2);
        p.Cin = m.mel_first.cin; p.W = m.mel_first.w; p.bias = m.mel_first.b; p.valid = F.d_valid;
        p.C = h0; p.ldc = C; p.M = F.R; p.N = C;
        gemm(c, p);
    }
    float* s1 = run_stack(c, m.mel_s1, h0, true, F.R, F.d_valid);
    float* mid = c.ws.get<float>((size_t)G * X.R * C);
    {
        GemmP p{};
        p.X = s1; p.strideX = (long long)F.R * C; p.ldx = C; p.Rx = F.R; p.rowbase = d_rowbase;
        p.taps = m.mel_mid.k; p.Cin = C; p.W = m.mel_mid.w; p.strideW = 0; p.bias = m.mel_mid.b; p.strideB = 0;
        p.valid = X.d_valid; p.C = mid; p.strideC = (long long)X.R * C; p.ldc = C; p.M = X.R; p.N = C; p.groups = G;
        gemm(c, p);
    }
    float* s2 = run_stack(c, m.mel_s2, mid, false, X.R, X.d_valid);
    float* sum = c.ws.get<float>((size_t)X.R * C);
    MT2_HIP(launch_sum_groups(s2, (long long)X.R * C, G, C, sum, C, C, X.R, c.s));
    float* ctx = c.ws.get<float>((size_t)X.R * C);
    conv_same(c, sum, C, X.R, m.mel_last, ctx, C, X.d_valid);
    return ctx;
}

struct MelPlan {
    RowSet F, X;
    int o_rowmapF, o_rowbase, o_rowmapX;
    RowPlanOffsets oF, oX;
};
static MelPlan plan_mel(IntPlan& ip, const mt2_config& cfg, const int* mel_lens, int B, int Tp_max, int Tc_max) {
    MelPlan mp;
    const int s = cfg.mrte_stride;
    mp.F = make_rows(mel_lens, B, std::max(s / 2, 2));
    std::vector<int> xl(B);
    for (int b = 0; b < B; ++b) {
        MT2_REQUIRE(mel_lens[b] >= 1 && mel_lens[b] <= Tp_max, "prompt mel length out of range");
        xl[b] = (mel_lens[b] - 1) / s + 1;      // Conv1d(k = s+1, stride s, pad s/2) output length
    }
    mp.X = make_rows(xl.data(), B, 2);
    mp.oF = plan_rows(ip, mp.F);
    mp.oX = plan_rows(ip, mp.X);
    mp.o_rowmapF = plan_rowmap(ip, mp.F, Tp_max);
    mp.o_rowmapX = plan_rowmap(ip, mp.X, Tc_max > 0 ? Tc_max : 1);
    std::vector<int> base(mp.X.R, kInvalidRow);
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < xl[b]; ++j) base[mp.X.off[b] + j] = mp.F.off[b] + j * s - s / 2;
    mp.o_rowbase = ip.add(base);
    return mp;
}

// internal side streams (created once per handle) + fork / join events
static void ensure_aux(mt2_model& m, int n_streams) {
    while ((int)m.aux_streams.size() < n_streams) {
        hipStream_t s;
        MT2_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        m.aux_streams.push_back(s);
    }
    if (!m.ev_fork) MT2_HIP(hipEventCreateWithFlags(&m.ev_fork, hipEventDisableTiming));
    while ((int)m.ev_join.size() < n_streams) {
        hipEvent_t e;
        MT2_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m.ev_join.push_back(e);
    }
}

// Range check of caller-supplied gather indices: the reference raises IndexError from nn.Embedding / F.embedding
// (modules/embedding.py:43-47, core_vq.py:188-190).  Here every gather kernel CLAMPS the id into its table (never an
// out-of-bounds read), and a tiny kernel per id tensor ORs a bit into a device flag.  The checks depend on the call's
// INPUTS only, so they run on the handle's own id stream, forked from the caller's stream at the first check of the
// call; the verdict (ids_verdict) is read at the END of the API call - a host wait for the id stream only, i.e. for
// what was queued on the caller's stream BEFORE this call, never for the call's own kernels.  A server can therefore
// enqueue batch k+1 behind batch k; the error still comes back from the call that carried the bad id.
enum IdBit { ID_PHONE = 1, ID_CODE = 2, ID_PREFIX = 4 };
static void ids_begin(const Ctx& c) {
    mt2_model& m = c.m;
    if (m.id_open) return;
    if (!m.id_stream) {     // all or nothing: a half-built set (stream without its events / flags) must never be used
        hipStream_t st = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        int *fd = nullptr, *fh = nullptr;
        const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
                        hipMalloc((void**)&fd, 4 * sizeof(int)) == hipSuccess &&
                        hipHostMalloc((void**)&fh, 4 * sizeof(int), hipHostMallocDefault) == hipSuccess;
        if (!ok) {
            if (fh) (void)hipHostFree(fh);
            if (fd) (void)hipFree(fd);
            if (e1) (void)hipEventDestroy(e1);
            if (e0) (void)hipEventDestroy(e0);
            if (st) (void)hipStreamDestroy(st);
            (void)hipGetLastError();
            throw Error("could not create the id-check stream, events or flags");
        }
        m.id_stream = st; m.ev_id_fork = e0; m.ev_id_done = e1; m.id_flag_dev = fd; m.id_flag_host = fh;
    }
    MT2_HIP(hipEventRecord(m.ev_id_fork, c.s));                  // the ids may be produced by earlier work on c.s
    MT2_HIP(hipStreamWaitEvent(m.id_stream, m.ev_id_fork, 0));
    MT2_HIP(hipMemsetAsync(m.id_flag_dev, 0, 4 * sizeof(int), m.id_stream));
    m.id_open = true;
}
// ids[map[r]] for r < R with map[r] >= 0 (map == nullptr: ids[r]); `map` is a HOST array, uploaded on the id stream
static void ids_check(const Ctx& c, const int64_t* ids, const std::vector<int>* map, long long R, long long hi, int bit) {
    MT2_REQUIRE(R < (1ll << 31), "id tensor too large");
    if (R <= 0) return;
    ids_begin(c);
    const int* dmap = nullptr;
    if (map) {
        MT2_REQUIRE((long long)map->size() >= R, "id map shorter than the id range");
        int* hm = static_cast<int*>(c.m.pinned().alloc((size_t)R * sizeof(int)));
        std::memcpy(hm, map->data(), (size_t)R * sizeof(int));
        int* dm = c.ws.get<int>((size_t)R);
        MT2_HIP(hipMemcpyAsync(dm, hm, (size_t)R * sizeof(int), hipMemcpyHostToDevice, c.m.id_stream));
        dmap = dm;
    }
    MT2_HIP(launch_check_ids(ids, dmap, (int)R, hi, c.m.id_flag_dev, bit, c.m.id_stream));
}
// end of the API call (also on its error paths, without throwing: CallScope): wait for the id stream, read the flag
static int ids_wait(mt2_model& m) {
    if (!m.id_open) return 0;
    m.id_open = false;
    *m.id_flag_host = 0;
    if (hipMemcpyAsync(m.id_flag_host, m.id_flag_dev, sizeof(int), hipMemcpyDeviceToHost, m.id_stream) != hipSuccess) return -1;
    if (hipEventRecord(m.ev_id_done, m.id_stream) != hipSuccess) return -1;
    if (hipEventSynchronize(m.ev_id_done) != hipSuccess) return -1;
    return *m.id_flag_host;
}
static void ids_verdict(const Ctx& c) {
    const int f = ids_wait(c.m);
    MT2_REQUIRE(f >= 0, "id range check could not be read back");
    if (f == 0) return;
    std::string what = "index out of range in embedding lookup:";
    if (f & ID_PHONE) what += " phone id >= phone_vocab_size (or negative);";
    if (f & ID_CODE) what += " prosody code >= vq_bins (or negative);";
    if (f & ID_PREFIX) what += " prompt prosody code >= vq_bins + 2 (or negative);";
    throw Error(what);
}

// MRTE.tc_latent (modules/mrte.py:154-171) -> packed rows [P.R, hidden] (gap rows zero)
// Several phone sequences per utterance may share ONE mel context (prompt-conditioned synthesis: the target's phones and the
// prompt's own phones against the same prompt mel, modules/datamodule.py:161-177): sequence s * B + b = set s, utterance b.
struct TcResult { float* rows; RowSet P; };
struct PhoneSet { const int64_t* ids; const int* lens; int Np_max; };
static TcResult tc_latent_rows(const Ctx& c, const std::vector<PhoneSet>& sets, const float* mel, const int* mel_lens,
                               int Tp_max, int B) {
    mt2_model& m = c.m;
    const mt2_config& cfg = m.cfg;
    const int H = cfg.mrte_hidden, S = (int)sets.size(), BS = B * S;
    IntPlan ip;
    MelPlan mp = plan_mel(ip, cfg, mel_lens, B, Tp_max, 0);
    std::vector<int> all_lens(BS);
    for (int s_ = 0; s_ < S; ++s_)
        for (int b = 0; b < B; ++b) {
            MT2_REQUIRE(sets[s_].lens[b] >= 1 && sets[s_].lens[b] <= sets[s_].Np_max, "phone length out of range");
            all_lens[s_ * B + b] = sets[s_].lens[b];
        }
    RowSet P = make_rows(all_lens.data(), BS, 2);
    MT2_REQUIRE(P.maxlen <= cfg.max_positions, "phone sequence longer than the positional table");
    RowPlanOffsets oP = plan_rows(ip, P);
    std::vector<int> idmap(P.R, -1);                 // row -> index into ITS SET's padded [B, Np_max] id tensor (gap rows: -1)
    std::vector<int> pos(P.R, 0);
    for (int s_ = 0; s_ < S; ++s_)
        for (int b = 0; b < B; ++b) {
            const int q = s_ * B + b;
            for (int t = 0; t < P.len[q]; ++t) {
                idmap[P.off[q] + t] = b * sets[s_].Np_max + t;
                pos[P.off[q] + t] = t;
            }
        }
    const int o_idmap = ip.add(idmap);
    const int o_pos = ip.add(pos);
    // cross attention: sequence q reads the mel context of utterance q % B
    std::vector<int> kvs(BS), kvl(BS);
    int o_kvs = -1, o_kvl = -1;
    ip.upload(c.ws, c.m.pinned(), c.s);
    bind_rows(ip, mp.oF, mp.F);
    bind_rows(ip, mp.oX, mp.X);
    bind_rows(ip, oP, P);
    IntPlan ip2;                                      // kv ranges need mp.X's offsets: a second (tiny) plan when S > 1
    if (S > 1) {
        for (int q = 0; q < BS; ++q) { kvs[q] = mp.X.off[q % B]; kvl[q] = mp.X.len[q % B]; }
        o_kvs = ip2.add(kvs); o_kvl = ip2.add(kvl);
        ip2.upload(c.ws, c.m.pinned(), c.s);
    }
    // phone ids index the embedding table: range check on the id stream (verdict at the end of the API call)
    for (int s_ = 0; s_ < S; ++s_) {
        const int r0 = P.off[s_ * B] - P.G, r1 = s_ + 1 < S ? P.off[(s_ + 1) * B] - P.G : P.R;
        std::vector<int> sub(idmap.begin() + r0, idmap.begin() + r1);
        ids_check(c, sets[s_].ids, &sub, (long long)sub.size(), cfg.phone_vocab, ID_PHONE);
    }

    // The phone branch (embedding, conv-FF transformer, query projection: ~60 small launches) does not depend on
    // the mel encoder: it runs on a side stream and fills the CUs the mel stack's big launches leave idle
    // (220 tiles on 256 CUs); the two meet at the cross attention.
    ensure_aux(m, 1);
    hipStream_t side = m.aux_streams[0];
    MT2_HIP(hipEventRecord(m.ev_fork, c.s));
    m.aux_forked = true;
    MT2_HIP(hipStreamWaitEvent(side, m.ev_fork, 0));
    const Ctx cp{m, side, c.ws};

    // prompt mel -> rows, mel encoder
    float* xmel = c.ws.get<float>((size_t)mp.F.R * cfg.mel_bins);
    MT2_HIP(launch_pack_rows(mel, cfg.mel_bins, Tp_max, 0, ip.dev(mp.o_rowmapF), xmel, cfg.mel_bins, mp.F.R, c.s));
    float* ctx = mel_context_rows(c, xmel, cfg.mel_bins, mp.F, mp.X, ip.dev(mp.o_rowbase));

    // phone embedding + PE, conv-FF transformer (mrte.py:159-160,165)
    float* x = c.ws.get<float>((size_t)P.R * H);
    for (int s_ = 0; s_ < S; ++s_) {                  // one embedding launch per id tensor, over that set's rows
        const int r0 = P.off[s_ * B] - P.G, r1 = s_ + 1 < S ? P.off[(s_ + 1) * B] - P.G : P.R;
        MT2_HIP(launch_embed_pe(m.phone_emb, H, sets[s_].ids, ip.dev(o_idmap) + r0, ip.dev(o_pos) + r0, m.pe_mrte,
                                x + (size_t)r0 * H, H, r1 - r0, cfg.phone_vocab, side));
    }
    EncScratch sc = enc_scratch(c, m.phone_enc, P.R);
    AttnGeom g;
    g.start = P.d_start; g.len = P.d_len; g.B = BS; g.max_len = P.maxlen;
    for (auto& lw : m.phone_enc.layers) encoder_layer(cp, m.phone_enc, lw, x, P.R, g, P.d_valid, sc);

    // cross attention, ONE head of width H (mrte.py:131-135,167), LayerNorm, ReLU (:168-169)
    float* q = sc.h;
    linear(cp, x, H, P.R, m.x_wq, m.x_bq, H, H, q, H);
    MT2_HIP(hipEventRecord(m.ev_join[0], side));
    MT2_HIP(hipStreamWaitEvent(c.s, m.ev_join[0], 0));
    float* kv = c.ws.get<float>((size_t)mp.X.R * 2 * H);
    linear(c, ctx, H, mp.X.R, m.x_wkv, m.x_bkv, 2 * H, H, kv, 2 * H);
    AttnP a{};
    a.Q = q; a.ldq = H; a.K = kv; a.ldk = 2 * H; a.V = kv + H; a.ldv = 2 * H; a.O = sc.att; a.ldo = H;
    a.q_start = P.d_start; a.q_len = P.d_len;
    a.kv_start = S > 1 ? ip2.dev(o_kvs) : mp.X.d_start; a.kv_len = S > 1 ? ip2.dev(o_kvl) : mp.X.d_len;
    a.B = BS; a.H = 1; a.D = H; a.max_qlen = P.maxlen; a.max_kvlen = mp.X.maxlen; a.scale = 1.0f / std::sqrt((float)H);
    a.lds_min_qlen = c.m.opts.attn_lds_min; a.lds_waves = c.m.opts.attn_lds_waves; a.x6_min_qlen = c.m.opts.attn_x6_min;
    a.x3h = (c.m.opts.x3h & 8) ? 1 : 0; a.x3h_flag = c.m.opts.x3h_flag;
    a.ds_short = c.m.opts.attn_ds;
    MT2_HIP(launch_attention(a, c.s));
    float* o = c.ws.get<float>((size_t)P.R * H);
    linear(c, sc.att, H, P.R, m.x_wo, m.x_bo, H, H, o, H);
    float* tc = c.ws.get<float>((size_t)P.R * H);
    layernorm(c, o, H, m.x_ng, m.x_nb, P.R, H, tc, H, P.d_valid, 0, nullptr, 0, 0, 0, ACT_RELU);
    return {tc, P};
}
static TcResult tc_latent_rows(const Ctx& c, const int64_t* phone, const int* phone_lens, int Np_max,
                               const float* mel, const int* mel_lens, int Tp_max, int B) {
    return tc_latent_rows(c, std::vector<PhoneSet>{{phone, phone_lens, Np_max}}, mel, mel_lens, Tp_max, B);
}

// ---------------------------------------------------------------------------------------------------
// autoregressive models.  Utterances are visited in order of decreasing length so that the sequences
// still active at step t are always slots [0, A_t); at step t every active sequence has exactly
// n = t+1 positions, stored compactly as rows j*n + i.  ALL positions are re-encoded every step,
// non-causally, exactly as the reference does (models/megatts2.py:172-179,264-273) -
// a KV cache would change the result.

// per-stage value of an engine option for the duration of a stage driver (restored on every exit path)
struct OptGuard {
    int& slot; int saved;
    OptGuard(int& s_, int v) : slot(s_), saved(s_) { if (v >= 0) slot = v; }
    ~OptGuard() { slot = saved; }
};
struct ArOrder {
    std::vector<int> slot_b, len;   // slot j -> utterance, length
    int nmax = 0;
};
static ArOrder ar_order(const int* lens, int B) {
    ArOrder o;
    o.slot_b.resize(B);
    std::iota(o.slot_b.begin(), o.slot_b.end(), 0);
    std::stable_sort(o.slot_b.begin(), o.slot_b.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    o.len.resize(B);
    for (int j = 0; j < B; ++j) o.len[j] = lens[o.slot_b[j]];
    o.nmax = B ? o.len[0] : 0;
    return o;
}

// ---- stream groups.  The sequences of an AR run are dealt round-robin (in length order) into G groups and
// every group runs its own step loop on its own HIP stream.  Nothing changes arithmetically - sequences are
// independent (batch-1 semantics) - but two independent kernel chains are in flight: one chain's launch
// boundary, workgroup-tail and prologue bubbles (~5 us fixed per launch, and the idle CUs of a launch whose
// tile count is not a multiple of the chip) are filled by the other chain's kernels.
struct ArGroups {
    int G = 1;
    std::vector<std::vector<int>> slots;   // group -> positions in the length-sorted order
    std::vector<hipStream_t> stream;
};
static ArGroups ar_groups(mt2_model& m, hipStream_t main, const ArOrder& ord, int B, int stage_groups = 0) {
    ArGroups g;
    g.G = std::max(1, std::min(stage_groups > 0 ? stage_groups : m.ar_groups, B));
    g.slots.resize(g.G);
    for (int j = 0; j < B; ++j) g.slots[j % g.G].push_back(j);
    ensure_aux(m, g.G - 1);
    g.stream.push_back(main);
    for (int i = 1; i < g.G; ++i) g.stream.push_back(m.aux_streams[i - 1]);
    return g;
}
static void ar_fork(mt2_model& m, const ArGroups& g) {
    if (g.G <= 1) return;
    MT2_HIP(hipEventRecord(m.ev_fork, g.stream[0]));
    m.aux_forked = true;
    for (int i = 1; i < g.G; ++i) MT2_HIP(hipStreamWaitEvent(g.stream[i], m.ev_fork, 0));
}
static void ar_join(mt2_model& m, const ArGroups& g) {
    for (int i = 1; i < g.G; ++i) {
        MT2_HIP(hipEventRecord(m.ev_join[i - 1], g.stream[i]));
        MT2_HIP(hipStreamWaitEvent(g.stream[0], m.ev_join[i - 1], 0));
    }
}

// A run may start from a FORCED history of P positions per sequence (uniform P): the PLM's prompt prefix (SURVEY 8f
// row f1: the layout the PLM is trained on, modules/datamodule.py:201-212) and the teacher-forced single steps of
// the long-shape parity tests.  The loop then starts at t = P and runs `max_steps` steps (0 = to the end).
struct ArPrefix {
    int P = 0;
    const void* data = nullptr;   // ADM: float [B, P] un-rounded predictions; PLM: int64 [B, P] prosody codes
    int max_steps = 0;
    int stride = 0;               // PLM: row stride of `data` in elements (0: P)
};

// MegaADM.infer (models/megatts2.py:257-275).  tc: rows buffer (ld), utterance b's first row row0[b].
static void adm_run(const Ctx& c, const float* tc, int ld_tc, int tc_rows, const std::vector<int>& row0,
                    const int* lens, int B, int32_t* dur_out, float* flt_out, int dstride,
                    const ArPrefix& pre = ArPrefix()) {
    mt2_model& m = c.m;
    const mt2_config& cfg = m.cfg;
    const OptGuard pairs_guard(m.opts.ln_pairs, m.opts.ln_pairs_adm);
    const EncW& e = m.adm_enc;
    const int d = e.d, Dc = cfg.adm_tc_emb_dim, De = cfg.adm_emb_dim;
    ArOrder ord = ar_order(lens, B);
    MT2_REQUIRE(ord.nmax <= cfg.max_positions, "ADM sequence longer than the positional table");
    for (int b = 0; b < B; ++b) MT2_REQUIRE(lens[b] > pre.P, "forced history is not shorter than the sequence");
    ArGroups grp = ar_groups(m, c.s, ord, B, m.adm_groups);
    struct Grp {
        int B, nmax, A; int o_tcrow, o_len, o_slot;
        std::vector<int> len;
        float *p, *x, *ylast, *qkv0; EncScratch sc;
    };
    std::vector<Grp> gs(grp.G);
    IntPlan ip;
    for (int g = 0; g < grp.G; ++g) {
        Grp& q = gs[g];
        q.B = (int)grp.slots[g].size();
        std::vector<int> tcrow, slot;
        for (int j : grp.slots[g]) {
            tcrow.push_back(row0[ord.slot_b[j]]);
            q.len.push_back(ord.len[j]);
            slot.push_back(ord.slot_b[j]);
        }
        q.nmax = q.len[0];
        q.A = q.B;
        q.o_tcrow = ip.add(tcrow); q.o_len = ip.add(q.len); q.o_slot = ip.add(slot);
    }
    ip.upload(c.ws, c.m.pinned(), c.s);

    float* tcemb = c.ws.get<float>((size_t)tc_rows * Dc);
    linear(c, tc, ld_tc, tc_rows, m.adm_wtc, nullptr, Dc, cfg.adm_tc_dim, tcemb, Dc);   // tc_linear_emb (no bias)
    const int pstride = ord.nmax + 1;
    float* p_all = c.ws.get<float>((size_t)B * pstride);
    int pofs = 0;
    for (Grp& q : gs) {
        const int Mmax = q.B * q.nmax;
        q.p = p_all + (size_t)pofs * pstride;
        pofs += q.B;
        // p_code starts at 0.0 (:262), followed by the forced history if any
        MT2_HIP(launch_adm_init_hist(q.p, pstride, static_cast<const float*>(pre.data), pre.P, ip.dev(q.o_slot), q.B,
                                     c.s));
        q.x = c.ws.get<float>((size_t)Mmax * d);
        q.sc = enc_scratch(c, e, std::max(Mmax, 2 * q.B));   // last layer: q | att rows of A sequences
        q.ylast = c.ws.get<float>((size_t)q.B * d);
        q.qkv0 = e.layers.size() >= 2 ? c.ws.get<float>((size_t)Mmax * 3 * d) : nullptr;   // layer-0 QKV cache
    }
    ar_fork(m, grp);
    const int t_end = pre.max_steps > 0 ? std::min(ord.nmax, pre.P + pre.max_steps) : ord.nmax;
    auto step = [&](int g, int t) {
        const int n = t + 1;
        Grp& q = gs[g];
        while (q.A > 0 && q.len[q.A - 1] <= t) --q.A;
        if (q.A == 0) return;
        Ctx cg{m, grp.stream[g], c.ws};
        MT2_HIP(launch_adm_step_input(tcemb, Dc, ip.dev(q.o_tcrow), m.adm_wdt, q.p, pstride, m.pe_adm, q.x, Dc, De,
                                      n, q.A, cg.s));
        const float* y = ar_step_layers(cg, e, q.x, n, q.A, q.qkv0, q.nmax, q.sc, q.ylast, t == pre.P && pre.P > 0);
        MT2_HIP(launch_adm_predict(y, d, m.adm_wpred, q.p, pstride, n, 1, q.A, cg.s));
    };
    for (int t = pre.P; t < t_end; ++t)        // interleaved: both chains' queues are fed step by step
        for (int g = 0; g < grp.G; ++g) step(g, t);
    for (int g = 0; g < grp.G; ++g) {
        Grp& q = gs[g];
        MT2_HIP(launch_adm_finalize(q.p, pstride, ip.dev(q.o_len), ip.dev(q.o_slot), dur_out, flt_out, dstride, q.B,
                                    dstride < q.nmax ? dstride : q.nmax, grp.stream[g]));
    }
    ar_join(m, grp);
}

// MegaPLM.infer (models/megatts2.py:165-181).  cond rows buffer (ld), utterance b's first row row0[b]; lens[b] =
// ALL positions of the sequence (prompt prefix + target); codes_out / last_logits receive the target positions.
static void plm_run(const Ctx& c, const float* cond, int ld_c, const std::vector<int>& row0, const int* lens, int B,
                    int64_t* codes_out, int ostride, float* last_logits, int logit_tmax,
                    const ArPrefix& pre = ArPrefix()) {
    mt2_model& m = c.m;
    const mt2_config& cfg = m.cfg;
    const OptGuard pairs_guard(m.opts.ln_pairs, 0);      // the hand-off pays in the ADM only (profiles/r05_opts_ab.txt)
    const EncW& e = m.plm_enc;
    const int d = e.d, Dc = cfg.plm_tc_dim, De = cfg.plm_vq_dim, NB = cfg.plm_bins;
    ArOrder ord = ar_order(lens, B);
    MT2_REQUIRE(ord.nmax <= cfg.max_positions, "PLM sequence longer than the positional table");
    MT2_REQUIRE(1024 < cfg.plm_bins + 2, "pc_embedding too small for the BOS id 1024");
    for (int b = 0; b < B; ++b) MT2_REQUIRE(lens[b] > pre.P, "prompt prefix is not shorter than the sequence");
    ArGroups grp = ar_groups(m, c.s, ord, B, m.plm_groups);
    struct Grp {
        int B, nmax, A; int o_crow, o_len, o_slot;
        std::vector<int> len, slot;
        int64_t* codes; float *x, *ylast, *qkv0, *logits; EncScratch sc;
    };
    std::vector<Grp> gs(grp.G);
    IntPlan ip;
    for (int g = 0; g < grp.G; ++g) {
        Grp& q = gs[g];
        q.B = (int)grp.slots[g].size();
        std::vector<int> crow;
        for (int j : grp.slots[g]) {
            crow.push_back(row0[ord.slot_b[j]]);
            q.len.push_back(ord.len[j]);
            q.slot.push_back(ord.slot_b[j]);
        }
        q.nmax = q.len[0];
        q.A = q.B;
        q.o_crow = ip.add(crow); q.o_len = ip.add(q.len); q.o_slot = ip.add(q.slot);
    }
    ip.upload(c.ws, c.m.pinned(), c.s);

    const int cstride = ord.nmax + 1;
    int64_t* codes_all = c.ws.get<int64_t>((size_t)B * cstride);
    int cofs = 0;
    for (Grp& q : gs) {
        const int Mmax = q.B * q.nmax;
        q.codes = codes_all + (size_t)cofs * cstride;
        cofs += q.B;
        // BOS literal 1024 (models/megatts2.py:170), then the prompt's codes if any - on the device, no host staging
        MT2_HIP(launch_plm_init_hist(q.codes, cstride, 1024, static_cast<const int64_t*>(pre.data), pre.P,
                                     pre.stride > 0 ? pre.stride : pre.P, ip.dev(q.o_slot), q.B, c.s));
        q.x = c.ws.get<float>((size_t)Mmax * d);
        q.logits = c.ws.get<float>((size_t)q.B * NB);
        q.sc = enc_scratch(c, e, std::max(Mmax, 2 * q.B));   // last layer: q | att rows of A sequences
        q.ylast = c.ws.get<float>((size_t)q.B * d);
        q.qkv0 = e.layers.size() >= 2 ? c.ws.get<float>((size_t)Mmax * 3 * d) : nullptr;   // layer-0 QKV cache
    }
    ar_fork(m, grp);
    const int t_end = pre.max_steps > 0 ? std::min(ord.nmax, pre.P + pre.max_steps) : ord.nmax;
    auto step = [&](int g, int t) {
        const int n = t + 1;
        Grp& q = gs[g];
        while (q.A > 0 && q.len[q.A - 1] <= t) --q.A;
        if (q.A == 0) return;
        Ctx cg{m, grp.stream[g], c.ws};
        MT2_HIP(launch_plm_step_input(cond, ld_c, ip.dev(q.o_crow), m.plm_emb, q.codes, cstride, m.pe_plm, q.x, Dc,
                                      De, n, q.A, NB + 2, cg.s));
        const float* y = ar_step_layers(cg, e, q.x, n, q.A, q.qkv0, q.nmax, q.sc, q.ylast, t == pre.P && pre.P > 0);
        // predict_layer on the last position of each sequence only (:178 takes [:, -1:]), then argmax
        GemmP p{};
        p.X = y; p.ldx = d; p.Rx = q.A; p.Cin = d; p.W = m.plm_wpred;
        p.C = q.logits; p.ldc = NB; p.M = q.A; p.N = NB;
        gemm(cg, p);
        MT2_HIP(launch_argmax_rows(q.logits, NB, NB, q.codes, cstride, n, q.A, cg.s));
        if (last_logits && t - pre.P < logit_tmax)
            for (int j = 0; j < q.A; ++j)
                MT2_HIP(hipMemcpyAsync(last_logits + ((size_t)q.slot[j] * logit_tmax + (t - pre.P)) * NB,
                                       q.logits + (size_t)j * NB, sizeof(float) * NB, hipMemcpyDeviceToDevice,
                                       cg.s));
    };
    for (int t = pre.P; t < t_end; ++t)
        for (int g = 0; g < grp.G; ++g) step(g, t);
    for (int g = 0; g < grp.G; ++g) {
        Grp& q = gs[g];
        const int nt = q.nmax - pre.P;
        MT2_HIP(launch_plm_finalize(q.codes, cstride, ip.dev(q.o_len), ip.dev(q.o_slot), codes_out, ostride, q.B,
                                    ostride < nt ? ostride : nt, pre.P, grp.stream[g]));
    }
    ar_join(m, grp);
}

// ---------------------------------------------------------------------------------------------------
// length regulation, PLM conditioning, decoder

struct FramePlan {
    RowSet D;                 // mel frames, gap 2
    std::vector<int> tq;      // prosody tokens per utterance
    std::vector<int> q_row0;  // first cond row of each utterance (compact, no gaps)
    int Qrows = 0;
    int o_tcmap, o_codemap, o_first, o_cnt, o_rowmapD;
    RowPlanOffsets oD;
};
// dur: host [B, dstride]; tc rows of utterance b start at tc_row0[b]; codes are addressed as b*code_stride + q
static FramePlan plan_frames(IntPlan& ip, const int* dur, int dstride, const int* lens, int B,
                             const std::vector<int>& tc_row0, int pool, int code_stride, int out_stride) {
    FramePlan fp;
    std::vector<int> tm(B);
    for (int b = 0; b < B; ++b) {
        long long s = 0;
        for (int i = 0; i < lens[b]; ++i) {
            MT2_REQUIRE(dur[(size_t)b * dstride + i] >= 0, "negative duration");
            s += dur[(size_t)b * dstride + i];
        }
        MT2_REQUIRE(s < (1 << 24), "utterance too long");
        tm[b] = (int)s;
    }
    fp.D = make_rows(tm.data(), B, 2);
    fp.oD = plan_rows(ip, fp.D);
    fp.o_rowmapD = plan_rowmap(ip, fp.D, out_stride > 0 ? out_stride : 1);
    std::vector<int> tcmap(fp.D.R, -1), codemap(fp.D.R, -1);
    fp.tq.resize(B);
    fp.q_row0.resize(B);
    int qr = 0;
    for (int b = 0; b < B; ++b) {
        int f = 0;
        for (int i = 0; i < lens[b]; ++i)                       // create_alignment, mrte.py:23-31
            for (int k = 0; k < dur[(size_t)b * dstride + i]; ++k, ++f) {
                tcmap[fp.D.off[b] + f] = tc_row0[b] + i;
                codemap[fp.D.off[b] + f] = b * code_stride + f / pool;
            }
        fp.tq[b] = (tm[b] + pool - 1) / pool;
        fp.q_row0[b] = qr;
        qr += fp.tq[b];
    }
    fp.Qrows = qr;
    std::vector<int> first(qr > 0 ? qr : 1, 0), cnt(qr > 0 ? qr : 1, 0);
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < fp.tq[b]; ++q) {
            first[fp.q_row0[b] + q] = fp.D.off[b] + q * pool;
            cnt[fp.q_row0[b] + q] = std::min(pool, tm[b] - q * pool);    // ceil_mode: partial last window
        }
    fp.o_tcmap = ip.add(tcmap);
    fp.o_codemap = ip.add(codemap);
    fp.o_first = ip.add(first);
    fp.o_cnt = ip.add(cnt);
    return fp;
}

// ConvNet.forward (modules/convnet.py:115-119) on rows xdec [D.R, decoder_in] -> mel rows [D.R, mel_bins]
static float* decoder_rows(const Ctx& c, const float* xdec, const RowSet& D) {
    mt2_model& m = c.m;
    const int H = m.cfg.dec_hidden, DIN = m.dec_first.cin;
    float* h0 = c.ws.get<float>((size_t)D.R * H);
    conv_same(c, xdec, DIN, D.R, m.dec_first, h0, H, D.d_valid);
    float* s = run_stack(c, m.dec_stack, h0, true, D.R, D.d_valid);
    float* mel = c.ws.get<float>((size_t)D.R * m.cfg.mel_bins);
    conv_same(c, s, H, D.R, m.dec_last, mel, m.cfg.mel_bins, D.d_valid);
    return mel;
}

// ---------------------------------------------------------------------------------------------------
// VQ prosody encoder (modules/vqpe.py:50-62)

struct VqpeResult { float* ze; int64_t* idx; RowSet F, Q; int o_rowmapQ, o_rowmapF, o_codemapF; };
static VqpeResult vqpe_rows(const Ctx& c, IntPlan& ip, const float* mel, int mel_ld, const int* lens, int T_max,
                            int Tq_max, int B) {
    mt2_model& m = c.m;
    const mt2_config& cfg = m.cfg;
    const int C = cfg.vq_hidden, G = m.vq_s1.groups, st = cfg.vq_stride, Dq = cfg.vq_dim;
    VqpeResult r;
    r.F = make_rows(lens, B, 2);
    std::vector<int> ql(B);
    for (int b = 0; b < B; ++b) {
        MT2_REQUIRE(lens[b] >= 1 && lens[b] <= T_max, "mel length out of range");
        ql[b] = (lens[b] + st - 1) / st;
    }
    r.Q = make_rows(ql.data(), B, 2);
    RowPlanOffsets oF = plan_rows(ip, r.F), oQ = plan_rows(ip, r.Q);
    r.o_rowmapF = plan_rowmap(ip, r.F, T_max);
    r.o_rowmapQ = plan_rowmap(ip, r.Q, Tq_max);
    std::vector<int> first(r.Q.R, 0), cnt(r.Q.R, 0), codemap(r.F.R, -1);
    for (int b = 0; b < B; ++b) {
        for (int q = 0; q < ql[b]; ++q) {
            first[r.Q.off[b] + q] = r.F.off[b] + q * st;
            cnt[r.Q.off[b] + q] = std::min(st, lens[b] - q * st);
        }
        for (int t = 0; t < lens[b]; ++t) codemap[r.F.off[b] + t] = r.Q.off[b] + t / st;
    }
    const int o_first = ip.add(first), o_cnt = ip.add(cnt);
    r.o_codemapF = ip.add(codemap);
    ip.upload(c.ws, c.m.pinned(), c.s);
    bind_rows(ip, oF, r.F);
    bind_rows(ip, oQ, r.Q);

    float* xmel = c.ws.get<float>((size_t)r.F.R * mel_ld);
    MT2_HIP(launch_pack_rows(mel, mel_ld, T_max, 0, ip.dev(r.o_rowmapF), xmel, mel_ld, r.F.R, c.s));
    float* h0 = c.ws.get<float>((size_t)r.F.R * C);
    {   // first_layer reads only the first vq_mel_bins columns (vqpe.py:55)
        GemmP p{};
        p.X = xmel; p.ldx = mel_ld; p.Rx = r.F.R; p.taps = m.vq_first.k; p.shift0 = -((m.vq_first.k - 1) / 2);
        p.Cin = m.vq_first.cin; p.W = m.vq_first.w; p.bias = m.vq_first.b; p.valid = r.F.d_valid;
        p.C = h0; p.ldc = C; p.M = r.F.R; p.N = C;
        gemm(c, p);
    }
    float* s1 = run_stack(c, m.vq_s1, h0, true, r.F.R, r.F.d_valid);
    float* mid = c.ws.get<float>((size_t)G * r.Q.R * C);     // MaxPool1d(stride, ceil_mode=True), vqpe.py:38
    for (int g = 0; g < G; ++g)
        MT2_HIP(launch_pool_max(s1 + (size_t)g * r.F.R * C, C, ip.dev(o_first), ip.dev(o_cnt),
                                mid + (size_t)g * r.Q.R * C, C, C, r.Q.R, c.s));
    float* s2 = run_stack(c, m.vq_s2, mid, false, r.Q.R, r.Q.d_valid);
    float* sum = c.ws.get<float>((size_t)r.Q.R * C);
    MT2_HIP(launch_sum_groups(s2, (long long)r.Q.R * C, G, C, sum, C, C, r.Q.R, c.s));
    r.ze = c.ws.get<float>((size_t)r.Q.R * Dq);
    conv_same(c, sum, C, r.Q.R, m.vq_last, r.ze, Dq, r.Q.d_valid);
    // EuclideanCodebook.quantize: distance GEMM + ordered argmax
    float* xe = c.ws.get<float>((size_t)r.Q.R * cfg.vq_bins);
    linear(c, r.ze, Dq, r.Q.R, m.codebook, nullptr, cfg.vq_bins, Dq, xe, cfg.vq_bins);
    r.idx = c.ws.get<int64_t>(r.Q.R);
    MT2_HIP(launch_vq_argmin(r.ze, Dq, Dq, xe, cfg.vq_bins, m.codebook_sq, cfg.vq_bins, r.Q.d_valid, r.idx, r.Q.R,
                             c.s));
    return r;
}

// ---------------------------------------------------------------------------------------------------
// HiFi-GAN V1 generator on mel rows [M0.R, in_dim] (gap 4) -> waveform rows [M0.R * hop] (1 channel, tanh'ed)

static float* hifigan_rows(const Ctx& c, const float* xmel, const RowSet& M0) {
    mt2_model& m = c.m;
    const mt2_config& cfg = m.cfg;
    const float slope = cfg.hg_slope;
    long long R = M0.R;
    int ch = cfg.hg_init_channels;
    // reflect edge mode (speechbrain): the halo rows of every utterance are mirrored into the gap in front of each
    // "same" convolution; `scale` = output rows per mel frame at the current stage
    const bool reflect = cfg.hg_reflect_pad != 0;
    long long scale = 1;
    int min_len = M0.B ? M0.len[0] : 0;
    for (int b = 1; b < M0.B; ++b) min_len = std::min(min_len, M0.len[b]);
    auto halo = [&](const Ctx& cc, float* buf, int width, int G) {
        if (!reflect || G <= 0) return;
        MT2_REQUIRE(2ll * G <= (long long)M0.G * scale, "reflect padding needs a gap of two halos between utterances");
        // torch F.pad(mode="reflect") / speechbrain raise when the padding is not smaller than the input: so do we
        MT2_REQUIRE((long long)min_len * scale > G, "reflect padding must be smaller than the utterance (as F.pad raises): "
                                                    "utterance too short for the vocoder's reflect-padded convolutions");
        MT2_HIP(launch_fill_reflect(buf, width, width, M0.d_start, M0.d_len, M0.B, scale, G, cc.s));
    };
    float* x = c.ws.get<float>((size_t)R * ch);
    if (reflect) {      // conv_pre reads its input in place: mirror the mel rows first (the rows are ours: a packed copy)
        MT2_REQUIRE((cfg.hg_in_dim & 3) == 0, "reflect padding needs a mel width that is a multiple of 4");
        halo(c, const_cast<float*>(xmel), cfg.hg_in_dim, (m.hg_pre.k - 1) / 2);
    }
    conv_same(c, xmel, cfg.hg_in_dim, (int)R, m.hg_pre, x, ch, M0.d_valid);
    const int* valid = M0.d_valid;
    for (int i = 0; i < cfg.hg_n_up; ++i) {
        const UpW& u = m.hg_up[i];
        const int s = u.stride, co = u.cout, hs = s / 2;
        MT2_REQUIRE(R * s < (1ll << 31), "waveform row count exceeds int32");
        // ConvTranspose1d as two phase GEMMs over the input rows; output [R, s*co] IS [R*s, co] time-major
        float* up = c.ws.get<float>((size_t)R * s * co);
        for (int half = 0; half < 2; ++half) {
            GemmP p{};
            p.X = x; p.ldx = ch; p.Rx = (int)R; p.taps = 2; p.shift0 = half == 0 ? -1 : 0; p.Cin = ch;
            p.W = half == 0 ? u.wlo : u.whi; p.bias = u.bias; p.valid = valid;
            p.C = up + (half == 0 ? 0 : hs * co); p.ldc = s * co; p.M = (int)R; p.N = hs * co;
            p.pro_act = ACT_LRELU; p.pro_slope = slope;
            gemm(c, p);
        }
        int* v2 = c.ws.get<int>((size_t)R * s);
        MT2_HIP(launch_expand_mask(valid, s, v2, R * s, c.s));
        valid = v2;
        R *= s;
        scale *= s;
        ch = co;
        // multi-receptive-field fusion: mean of the resblocks (ResBlock1: x += conv2(lrelu(conv1(lrelu(x)))) x3)
        // The three resblocks of a stage only share their input: each runs on its own stream (the caller's + two
        // side streams), so that one chain's launch tails (868 tiles on 256 CUs at stage 1) are filled by the others.
        const size_t per = (size_t)R * ch;
        float* rb[3] = {nullptr, nullptr, nullptr};
        MT2_REQUIRE(cfg.hg_n_res == 3, "HiFi-GAN V1 uses three resblocks per stage");
        const int nside = m.opts.voc_streams > 1 ? 2 : 0;
        ensure_aux(m, nside);
        if (reflect) {     // the three chains share `up`: one halo wide enough for the widest first convolution, before the fork
            int g0 = 0;
            for (int j = 0; j < 3; ++j) g0 = std::max(g0, (m.hg_res[i * 3 + j].k - 1) / 2 * m.hg_res[i * 3 + j].dil[0]);
            halo(c, up, ch, g0);
        }
        if (nside) {
            MT2_HIP(hipEventRecord(m.ev_fork, c.s));
            m.aux_forked = true;
            for (int j = 0; j < nside; ++j) MT2_HIP(hipStreamWaitEvent(m.aux_streams[j], m.ev_fork, 0));
        }
        for (int j = 0; j < 3; ++j) {
            const ResW& r = m.hg_res[i * 3 + j];
            const Ctx cj{m, (nside && j > 0) ? m.aux_streams[j - 1] : c.s, c.ws};
            float* t1 = c.ws.get<float>(per);
            float* ha = c.ws.get<float>(per);
            float* hb = c.ws.get<float>(per);
            rb[j] = c.ws.get<float>(per);
            const float* h = up;
            for (int n = 0; n < 3; ++n) {
                float* out = n == 2 ? rb[j] : (n == 0 ? ha : hb);
                // x + conv2(lrelu(conv1(lrelu(x)))): the inner leaky ReLU has ONE consumer, so it is applied once in
                // conv1's epilogue instead of on every operand fragment of conv2 (same values, no VALU in that loop)
                if (n > 0) halo(cj, const_cast<float*>(h), ch, (r.k - 1) / 2 * r.dil[n]);
                conv_same(cj, h, ch, (int)R, r.c1[n], t1, ch, valid, ACT_LRELU, slope, ACT_LRELU, nullptr, 0, r.dil[n]);
                halo(cj, t1, ch, (r.k - 1) / 2);
                conv_same(cj, t1, ch, (int)R, r.c2[n], out, ch, valid, ACT_NONE, slope, ACT_NONE, h, ch, 1);
                h = out;
            }
        }
        for (int j = 0; j < nside; ++j) {
            MT2_HIP(hipEventRecord(m.ev_join[j], m.aux_streams[j]));
            MT2_HIP(hipStreamWaitEvent(c.s, m.ev_join[j], 0));
        }
        if (!reflect && i + 1 == cfg.hg_n_up && m.hg_post.cout == 1 && (ch & 3) == 0 && ch <= 128 && m.hg_post.k <= 15) {
            // last stage: the mean goes straight into the output layer (one pass over the three resblock outputs)
            // F.leaky_relu default slope 0.01, conv_post k7, tanh
            float* wav = c.ws.get<float>((size_t)R);
            const hipError_t e = launch_conv_post(rb[0], rb[1], rb[2], 1.0f / 3.0f, R, ch, m.hg_post.k, m.hg_post.w,
                                                  m.hg_post.b, 0.01f, valid, wav, c.s);
            if (e == hipSuccess) return wav;
            if (e != hipErrorNotSupported) MT2_HIP(e);      // not supported (LDS): the two-launch form below
        }
        x = c.ws.get<float>(per);
        MT2_HIP(launch_avg3(rb[0], rb[1], rb[2], 1.0f / 3.0f, x, (long long)per, c.s));
    }
    float* wav = c.ws.get<float>((size_t)R);
    halo(c, x, ch, (m.hg_post.k - 1) / 2);
    conv_same(c, x, ch, (int)R, m.hg_post, wav, 1, valid, ACT_LRELU, 0.01f, ACT_TANH);
    return wav;
}

// ---------------------------------------------------------------------------------------------------
// mel front-end: extract_mel_spec (modules/tokenizer.py:107-125).  Restates torchaudio's Spectrogram +
// MelScale (neither is in the reference tree; parity unpinned): periodic Hann window,
// center=True with reflect padding, one-sided magnitude (power 1), slaney mel scale and slaney area norm,
// log(clamp(., clip)).  Constants are built once per audio configuration in double precision.

static double hz_to_mel_slaney(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz_slaney(double mel) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return mel >= min_log_mel ? min_log_hz * std::exp(logstep * (mel - min_log_mel)) : f_sp * mel;
}
static void frontend_prepare(mt2_model& m, const mt2_audio_config& ac) {
    if (m.fe_basis && std::memcmp(&m.fe_cfg, &ac, sizeof(ac)) == 0) return;
    MT2_REQUIRE(ac.n_fft >= 8 && ac.hop_length >= 4 && ac.hop_length % 4 == 0 && ac.n_fft % ac.hop_length == 0,
                "n_fft must be a multiple of hop_length, hop_length a multiple of 4");
    MT2_REQUIRE(ac.win_length >= 1 && ac.win_length <= ac.n_fft && ac.n_mels >= 1 && ac.n_mels % 4 == 0,
                "win_length <= n_fft, n_mels a multiple of 4");
    MT2_REQUIRE(ac.f_max > ac.f_min && ac.sample_rate > 0 && ac.clip > 0.0f, "bad audio configuration");
    const int N = ac.n_fft, F = N / 2 + 1, Fp = (F + 3) & ~3;
    const double PI = 3.14159265358979323846;
    std::vector<double> win(N, 0.0);
    const int left = (N - ac.win_length) / 2;       // torch.stft centres a short window inside n_fft
    for (int k = 0; k < ac.win_length; ++k) win[left + k] = 0.5 - 0.5 * std::cos(2.0 * PI * k / ac.win_length);
    std::vector<float> basis((size_t)2 * F * N);
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < N; ++k) {
            const double ang = 2.0 * PI * (double)(((long long)f * k) % N) / N;
            basis[(size_t)f * N + k] = (float)(win[k] * std::cos(ang));
            basis[(size_t)(F + f) * N + k] = (float)(-win[k] * std::sin(ang));
        }
    // torchaudio.functional.melscale_fbanks(n_freqs=F, f_min, f_max, n_mels, sample_rate, "slaney", "slaney")
    std::vector<float> fb((size_t)ac.n_mels * Fp, 0.0f);
    const double m_min = hz_to_mel_slaney(ac.f_min), m_max = hz_to_mel_slaney(ac.f_max);
    std::vector<double> fpts(ac.n_mels + 2);
    for (int i = 0; i < ac.n_mels + 2; ++i) fpts[i] = mel_to_hz_slaney(m_min + (m_max - m_min) * i / (ac.n_mels + 1));
    for (int j = 0; j < ac.n_mels; ++j) {
        const double enorm = 2.0 / (fpts[j + 2] - fpts[j]);
        for (int f = 0; f < F; ++f) {
            const double freq = (double)(ac.sample_rate / 2) * f / (F - 1);
            const double down = (freq - fpts[j]) / (fpts[j + 1] - fpts[j]);
            const double up = (fpts[j + 2] - freq) / (fpts[j + 2] - fpts[j + 1]);
            fb[(size_t)j * Fp + f] = (float)(std::max(0.0, std::min(down, up)) * enorm);
        }
    }
    auto up = [&](const std::vector<float>& v) {
        float* d = nullptr;
        MT2_HIP(hipMalloc(reinterpret_cast<void**>(&d), v.size() * sizeof(float)));
        MT2_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        m.dev_allocs.push_back(d);
        return d;
    };
    m.fe_basis = up(basis);
    m.fe_fb = up(fb);
    m.fe_nfreq = F; m.fe_nfreq_pad = Fp;
    m.fe_cfg = ac;
}

static void mel_spectrogram_run(const Ctx& c, const mt2_audio_config& ac, const float* wav, const int* lens, int L_max,
                                int B, float* mel, int T_max) {
    mt2_model& m = c.m;
    frontend_prepare(m, ac);
    const int hop = ac.hop_length, taps = ac.n_fft / hop, pad = ac.n_fft / 2, F = m.fe_nfreq, Fp = m.fe_nfreq_pad;
    std::vector<int> blk_b, blk_t, rowbase, rowmap, len(lens, lens + B);
    for (int b = 0; b < B; ++b) {
        MT2_REQUIRE(lens[b] > pad && lens[b] <= L_max, "waveform shorter than n_fft/2 + 1 samples (reflect padding) or > L_max");
        const int T = 1 + lens[b] / hop, row0 = (int)blk_b.size();
        MT2_REQUIRE(T <= T_max, "T_max smaller than 1 + L / hop_length");
        for (int t = 0; t < T - 1 + taps; ++t) { blk_b.push_back(b); blk_t.push_back(t); }
        for (int t = 0; t < T; ++t) { rowbase.push_back(row0 + t); rowmap.push_back(b * T_max + t); }
    }
    const int Rb = (int)blk_b.size(), Fr = (int)rowbase.size();
    IntPlan ip;
    const int o_b = ip.add(blk_b), o_t = ip.add(blk_t), o_len = ip.add(len), o_base = ip.add(rowbase),
              o_map = ip.add(rowmap);
    ip.upload(c.ws, c.m.pinned(), c.s);
    float* xp = c.ws.get<float>((size_t)Rb * hop);
    MT2_HIP(launch_reflect_pad_blocks(wav, L_max, ip.dev(o_b), ip.dev(o_t), ip.dev(o_len), hop, pad, xp, Rb, c.s));
    const int lds = (2 * F + 3) & ~3;
    float* spec = c.ws.get<float>((size_t)Fr * lds);
    {   // STFT = Conv1d over hop-sized blocks: frame t reads blocks t .. t+taps-1
        GemmP p{};
        p.X = xp; p.ldx = hop; p.Rx = Rb; p.rowbase = ip.dev(o_base); p.taps = taps; p.Cin = hop;
        p.W = m.fe_basis; p.C = spec; p.ldc = lds; p.M = Fr; p.N = 2 * F;
        gemm(c, p);
    }
    float* mag = c.ws.get<float>((size_t)Fr * Fp);
    MT2_HIP(launch_magnitude(spec, lds, F, mag, Fp, Fr, c.s));
    float* mrows = c.ws.get<float>((size_t)Fr * ac.n_mels);
    {   // MelScale + dynamic range compression
        GemmP p{};
        p.X = mag; p.ldx = Fp; p.Rx = Fr; p.Cin = Fp; p.W = m.fe_fb; p.C = mrows; p.ldc = ac.n_mels; p.M = Fr;
        p.N = ac.n_mels; p.epi_act = ACT_LOGCLAMP; p.pro_slope = ac.clip;
        gemm(c, p);
    }
    MT2_HIP(hipMemsetAsync(mel, 0, sizeof(float) * (size_t)B * T_max * ac.n_mels, c.s));
    MT2_HIP(launch_unpack_rows(mrows, ac.n_mels, ac.n_mels, T_max, 0, ip.dev(o_map), mel, Fr, c.s));
}

}  // namespace mt2

// the C ABI lives in capi.hip and includes this translation unit's helpers
#include "capi.inc"
